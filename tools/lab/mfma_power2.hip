// The power price of what surrounds the MFMAs in a bf16x6 main loop (companion of mfma_power.hip): a GEMM-shaped loop without a
// GEMM's addressing.  A workgroup of four waves (2 x 2) owns a (64 MT) x (64 NT) tile; per iteration every wave reads its
// (MT + NT) x 3 plane fragments from LDS (static random operands), issues MT x NT x 6 MFMAs and, optionally, PIECES LDS-DMA
// loads of 1 KiB (buffer_load ... lds) from a global buffer of `span` bytes per workgroup stream, behind a vmcnt(0) + barrier
// per iteration as in conv_mainloop_xp.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power2 tools/lab/mfma_power2.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// LDS operand image: [plane][half][row][16 B]; rows 0 .. 64 MT - 1 are A, then 64 NT rows of B
template <int MT, int NT, int PIECES, bool BARRIER, int PAD>      // PAD: waves per SIMD the register budget is set for
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PAD, PAD))) void loop_kernel(const u32x4* __restrict__ opnd, const char* __restrict__ stream, unsigned span,
                                                   float* __restrict__ out, long iters, unsigned long long* __restrict__ clk) {
    constexpr int ROWS = 64 * (MT + NT);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dump = lds + 3 * 2 * ROWS * 16;                       // two stages of PIECES x 4 KiB for the DMA stream
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 3 * 2 * ROWS; i += 256) reinterpret_cast<u32x4*>(lds)[i] = opnd[(i + 97 * blockIdx.x) % (3 * 2 * 64 * 12)];
    __syncthreads();
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(stream), 0, (int)span, 0x00020000);
    // this workgroup's window into the stream: PIECES x 4 KiB per iteration, walking through `span` bytes
    unsigned pos = (unsigned)(((unsigned long long)blockIdx.x * 40503ull * 4096ull) % span);
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && tid == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    int buf = 0;
    for (long it = 0; it < iters; ++it) {
        if (PIECES > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BARRIER) __syncthreads();
        if (PIECES > 0) {
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                unsigned off = pos + (unsigned)(p * 4096 + wave * 1024 + lane * 16);
                if (off >= span) off -= span;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dump + buf * PIECES * 4096 + p * 4096 + wave * 1024), 16, (int)off, 0, 0, 0);
            }
            pos += PIECES * 4096;
            if (pos >= span) pos -= span;
            buf ^= 1;
        }
        bf16x8 a[MT][3], b[NT][3];
        constexpr int PA[3] = {2, 0, 1}, PB[3] = {0, 2, 1};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
                a[t][PA[q]] = *reinterpret_cast<const bf16x8*>(lds + ((PA[q] * 2 + h) * ROWS + (wr * MT + t) * 32 + r) * 16);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                b[t][PB[q]] = *reinterpret_cast<const bf16x8*>(lds + ((PB[q] * 2 + h) * ROWS + 64 * MT + (wc * NT + t) * 32 + r) * 16);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][TA[t]], b[nt][TB[t]], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (PIECES > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0 && tid == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    if (s == 123.456f) out[tid] = s + dump[tid];
}

static uint16_t bf16_rn(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

struct Case {
    const char* name;
    int mt, nt, pieces, wgs;
    bool barrier;
    size_t span;
    void (*fn)(const u32x4*, const char*, unsigned, float*, long, unsigned long long*);
};

int main(int argc, char** argv) {
    const double target_ms = argc > 1 ? atof(argv[1]) : 400.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // operands: [plane][half][row][8 bf16] with the three planes of N(0, 1) values; 768 rows
    const int rows = 64 * 12;
    std::vector<uint16_t> host((size_t)3 * 2 * rows * 8);
    srand(99);
    for (int row = 0; row < rows; ++row)
        for (int hh = 0; hh < 2; ++hh)
            for (int e = 0; e < 8; ++e) {
                const float u1 = (rand() + 1.f) / 2147483649.f, u2 = (rand() + 1.f) / 2147483649.f;
                const float x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
                const uint16_t hi = bf16_rn(x);
                const float r1 = x - bf16_f(hi);
                const uint16_t mi = bf16_rn(r1), lo = bf16_rn(r1 - bf16_f(mi));
                host[((size_t)(0 * 2 + hh) * rows + row) * 8 + e] = hi;
                host[((size_t)(1 * 2 + hh) * rows + row) * 8 + e] = mi;
                host[((size_t)(2 * 2 + hh) * rows + row) * 8 + e] = lo;
            }
    uint16_t* dop;
    CK(hipMalloc(&dop, host.size() * 2));
    CK(hipMemcpy(dop, host.data(), host.size() * 2, hipMemcpyHostToDevice));
    const size_t big = (size_t)2 << 30;                    // 2 GiB stream: past L2 and the 256 MB Infinity Cache
    char* dstream;
    CK(hipMalloc(&dstream, big));
    {   // random bits in the stream too
        std::vector<uint32_t> rb(1 << 22);
        for (auto& v : rb) v = (uint32_t)rand() * 2654435761u;
        for (size_t o = 0; o < big; o += rb.size() * 4) CK(hipMemcpy(dstream + o, rb.data(), rb.size() * 4, hipMemcpyHostToDevice));
    }
    float* dout;
    unsigned long long* dclk;
    CK(hipMalloc(&dout, 4096));
    CK(hipMalloc(&dclk, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t L2 = (size_t)8 << 20;       // 8 MiB: fits the 8 x 4 MiB of L2 with every XCD holding its own copy of what it touches
    const size_t MALL = (size_t)128 << 20;   // fits the Infinity Cache, not the L2s
    Case cases[] = {
        {"2x2 tile: LDS reads + MFMA, no barrier", 2, 2, 0, 3, false, L2, loop_kernel<2, 2, 0, false, 3>},
        {"2x2 tile: + barrier per k-tile", 2, 2, 0, 3, true, L2, loop_kernel<2, 2, 0, true, 3>},
        {"2x2 tile: + 6 DMA pieces/wave (24 KB per k-tile) from L2", 2, 2, 6, 3, true, L2, loop_kernel<2, 2, 6, true, 3>},
        {"2x2 tile: + 6 DMA pieces/wave from the Infinity Cache", 2, 2, 6, 3, true, MALL, loop_kernel<2, 2, 6, true, 3>},
        {"2x2 tile: + 6 DMA pieces/wave from HBM", 2, 2, 6, 3, true, big, loop_kernel<2, 2, 6, true, 3>},
        {"2x2 tile: + 4 DMA pieces/wave from L2 (halo-shared A)", 2, 2, 4, 3, true, L2, loop_kernel<2, 2, 4, true, 3>},
        {"2x2 tile: + 3 DMA pieces/wave from L2", 2, 2, 3, 3, true, L2, loop_kernel<2, 2, 3, true, 3>},
        {"2x4 tile: LDS reads + MFMA, no barrier, 2 WG/CU", 2, 4, 0, 2, false, L2, loop_kernel<2, 4, 0, false, 2>},
        {"2x4 tile: + barrier + 9 DMA pieces/wave from L2", 2, 4, 9, 2, true, L2, loop_kernel<2, 4, 9, true, 2>},
        {"2x4 tile: + barrier + 7 DMA pieces/wave from L2 (halo-shared A)", 2, 4, 7, 2, true, L2, loop_kernel<2, 4, 7, true, 2>},
        {"4x2 tile: + barrier + 9 DMA pieces/wave from L2", 4, 2, 9, 2, true, L2, loop_kernel<4, 2, 9, true, 2>},
        {"2x2 tile, 2 WG/CU: + barrier + 6 DMA pieces from L2", 2, 2, 6, 2, true, L2, loop_kernel<2, 2, 6, true, 3>},
        {"2x2 tile, 1 WG/CU: LDS reads + MFMA", 2, 2, 0, 1, false, L2, loop_kernel<2, 2, 0, false, 3>},
    };
    printf("%s: %d CUs, ~%.0f ms per case, N(0,1) operands split h/m/l\n", prop.name, cus, target_ms);
    for (const Case& c : cases) {
        const int rowsc = 64 * (c.mt + c.nt);
        size_t lds = (size_t)3 * 2 * rowsc * 16 + (size_t)2 * c.pieces * 4096;
        const size_t want = (size_t)160 * 1024 / c.wgs;                       // occupancy through the LDS size
        if (lds < want - 8192 && c.wgs < 3) lds = want - 8192;
        if (c.wgs == 3 && lds < 48 * 1024) lds = 48 * 1024;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(c.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int grid = cus * c.wgs;
        const int nm = c.mt * c.nt * 6;
        long iters = (long)(target_ms * 1e-3 * 1.8e9 / (32.0 * nm * c.wgs));
        auto launch = [&](long n) { hipLaunchKernelGGL(c.fn, dim3(grid), dim3(256), lds, 0, reinterpret_cast<const u32x4*>(dop), dstream, (unsigned)(c.span > 0xfffff000u ? 0xfffff000u : c.span), dout, n, dclk); };
        launch(iters / 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        launch(iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long clk[2];
        CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
        const double flops = (double)iters * nm * 32768.0 * 4 * grid;
        const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
        const double tf = flops / (ms * 1e-3) / 1e12;
        const double gbs = (double)iters * c.pieces * 4096.0 * grid / (ms * 1e-3) / 1e9;
        printf("%-66s %7.1f ms %7.1f TFLOP/s  clock %.3f GHz  busy %.3f  busy x GHz %.3f   DMA %6.0f GB/s\n", c.name, ms, tf, ghz,
               tf * 1e12 / (cus * 4.0 * 1024.0 * ghz * 1e9), tf / (cus * 4.0 * 1024.0) * 1e3, gbs);
        fflush(stdout);
    }
    return 0;
}
