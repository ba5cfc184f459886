// What the power budget buys on the bf16 matrix cores, without any kernel around them: every wave holds the three bf16 planes
// of 2 A and 2 B fragments (32 x 16 each) in registers and issues the 24 MFMAs of a 64 x 64 x 16 bf16x6 step in a loop, in
// different orders and on different data.  Prints TFLOP/s (bf16 MFMA flops), the shader clock and flops per clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/lab/mfma_power.hip && /tmp/mfma_power
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

// ORDER 0: product-major, tiles inside (what x6_mfma_tile does)     1: accumulator-stationary (six products back to back)
//       2: A-fragment-stationary                                    3: product-major with the tile order reversed every product
//       4: only h*h, six times (same MFMA count)                    5 / 6: order 0 with the accumulators pinned in VGPRs / AGPRs
template <int ORDER>
__global__ __launch_bounds__(256) void mfma32_kernel(const bf16x8* __restrict__ frag, float* __restrict__ out, long iters,
                                                     unsigned long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 a[2][3], b[2][3];
    const bf16x8* f = frag + ((size_t)(blockIdx.x & 63) * 4 + wave) * 12 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[t][p] = f[(t * 3 + p) * 64];
            b[t][p] = f[(6 + t * 3 + p) * 64];
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
    for (long it = 0; it < iters; ++it) {
        if (ORDER == 5 || ORDER == 6) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ORDER == 5)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q >> 1][q & 1]) : "v"(a[q >> 1][TA[t]]), "v"(b[q & 1][TB[t]]));
                    else
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[q >> 1][q & 1]) : "v"(a[q >> 1][TA[t]]), "v"(b[q & 1][TB[t]]));
                }
        } else if (ORDER == 0 || ORDER == 3 || ORDER == 4) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int qq = (ORDER == 3 && (t & 1)) ? 3 - q : q;
                    const int mt = qq >> 1, nt = qq & 1;
                    const int pa = ORDER == 4 ? 0 : TA[t], pb = ORDER == 4 ? 0 : TB[t];
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][pa], b[nt][pb], acc[mt][nt], 0, 0, 0);
                }
        } else if (ORDER == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    acc[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q >> 1][TA[t]], b[q & 1][TB[t]], acc[q >> 1][q & 1], 0, 0, 0);
        } else {
            // A-stationary: a[mt][l] x b[.][h]; a[mt][m] x b[.][m], b[.][h]; a[mt][h] x b[.][l], b[.][m], b[.][h]
            constexpr int NA[3] = {3, 2, 1};      // plane h pairs with l, m, h; m with m, h; l with h
            constexpr int PB0[3][3] = {{2, 1, 0}, {1, 0, 0}, {0, 0, 0}};
#pragma unroll
            for (int pa = 2; pa >= 0; --pa)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int k = 0; k < NA[pa]; ++k)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][pa], b[nt][PB0[pa][k]], acc[mt][nt], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// the 16 x 16 x 32 form: two planes side by side in one MFMA, 4 x 4 tiles x 3 MFMAs (xp16_mfma_ktile)
__global__ __launch_bounds__(256) void mfma16_kernel(const bf16x8* __restrict__ frag, float* __restrict__ out, long iters,
                                                     unsigned long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bf16x8* f = frag + ((size_t)(blockIdx.x & 63) * 4 + wave) * 12 * 64 + lane;
    bf16x8 ahl[4], ahm[4], blh[4], bmh[4], bhm[4];
    // lanes 0-31 carry the first plane, 32-63 the second: build from the plane fragments
    const bool second = lane >= 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x8 ah = f[((i & 1) * 3 + 0) * 64], am = f[((i & 1) * 3 + 1) * 64], al = f[((i & 1) * 3 + 2) * 64];
        const bf16x8 bh = f[(6 + (i & 1) * 3 + 0) * 64], bm = f[(6 + (i & 1) * 3 + 1) * 64], bl = f[(6 + (i & 1) * 3 + 2) * 64];
        ahl[i] = second ? al : ah;
        ahm[i] = second ? am : ah;
        blh[i] = second ? bh : bl;
        bmh[i] = second ? bh : bm;
        bhm[i] = second ? bm : bh;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahl[rb], blh[cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[rb], bmh[cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[rb], bhm[cb], acc[rb][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) out[threadIdx.x] = s;
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

// data 0: zeros   1: N(0, 1) fp32 split in three planes   2: random bits with sane exponents in every plane
// 3: N(0, 1) * 0.05 split (score-like magnitudes)   4: h plane of N(0, 1), m and l planes zero
static void fill(std::vector<uint16_t>& v, int kind) {
    const size_t nfrag = v.size() / (12 * 64 * 8);      // [block & 63][wave] sets of 12 fragments x 64 lanes x 8 bf16
    srand(1234);
    auto nrm = []() {
        const float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = (rand() + 1.f) / (RAND_MAX + 2.f);
        return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
    };
    for (size_t s = 0; s < nfrag; ++s)
        for (int fr = 0; fr < 4; ++fr)                      // a0, a1, b0, b1
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    uint16_t h = 0, m = 0, l = 0;
                    if (kind == 1 || kind == 3 || kind == 4) {
                        const float x = nrm() * (kind == 3 ? 0.05f : 1.f);
                        h = bf16_rn(x);
                        const float r = x - bf16_f(h);
                        m = bf16_rn(r);
                        l = bf16_rn(r - bf16_f(m));
                        if (kind == 4) m = l = 0;
                    } else if (kind == 2) {
                        h = (uint16_t)((rand() & 0x80ff) | (0x3f00 + ((rand() & 3) << 7)));
                        m = (uint16_t)((rand() & 0x80ff) | (0x3b00 + ((rand() & 3) << 7)));
                        l = (uint16_t)((rand() & 0x80ff) | (0x3700 + ((rand() & 3) << 7)));
                    }
                    const size_t base = ((s * 12 + fr * 3) * 64 + lane) * 8 + e;
                    v[base] = h;
                    v[base + 64 * 8] = m;
                    v[base + 2 * 64 * 8] = l;
                }
}

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
    const double target_ms = argc > 2 ? atof(argv[2]) : 150.0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs; %d waves per SIMD\n", prop.name, cus, waves_per_simd);
    const size_t nelem = (size_t)64 * 4 * 12 * 64 * 8;
    std::vector<uint16_t> host(nelem);
    uint16_t* dfrag;
    float* dout;
    unsigned long long* dclk;
    CK(hipMalloc(&dfrag, nelem * 2));
    CK(hipMalloc(&dout, 4096));
    CK(hipMalloc(&dclk, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const char* dname[5] = {"zeros", "N(0,1) split h/m/l", "random bits", "0.05 N(0,1) split", "h plane only"};
    const char* oname[8] = {"product-major (x6_mfma_tile)", "accumulator-stationary", "A-stationary", "product-major, zigzag tiles",
                            "h*h six times", "product-major, VGPR acc", "product-major, AGPR acc", "16x16x32, planes paired"};
    const int grid = cus * waves_per_simd;
    for (int kind = 0; kind < 5; ++kind) {
        fill(host, kind);
        CK(hipMemcpy(dfrag, host.data(), nelem * 2, hipMemcpyHostToDevice));
        for (int order = 0; order < 8; ++order) {
            auto launch = [&](long iters) {
                const bf16x8* fp = reinterpret_cast<const bf16x8*>(dfrag);
                switch (order) {
                    case 0: hipLaunchKernelGGL(mfma32_kernel<0>, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                    case 1: hipLaunchKernelGGL(mfma32_kernel<1>, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                    case 2: hipLaunchKernelGGL(mfma32_kernel<2>, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                    case 3: hipLaunchKernelGGL(mfma32_kernel<3>, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                    case 4: hipLaunchKernelGGL(mfma32_kernel<4>, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                    case 5: hipLaunchKernelGGL(mfma32_kernel<5>, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                    case 6: hipLaunchKernelGGL(mfma32_kernel<6>, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                    default: hipLaunchKernelGGL(mfma16_kernel, dim3(grid), dim3(256), 0, 0, fp, dout, iters, dclk); break;
                }
            };
            // 24 MFMAs of 32 cycles (or 48 of 16) per iteration and wave: ~768 cycles x waves per SIMD
            long iters = (long)(target_ms * 1e-3 * 2.0e9 / (768.0 * waves_per_simd));
            launch(iters / 4);                             // warm-up: lets the clock settle under this load
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            launch(iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long clk[2];
            CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
            const double flops = (double)iters * 24.0 * 32 * 32 * 16 * 2 * 4 * grid;
            const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);         // wall clock ticks at 100 MHz
            const double tf = flops / (ms * 1e-3) / 1e12;
            printf("data %-20s order %-30s %8.2f ms  %7.1f TFLOP/s  clock %.3f GHz  busy %.3f  busy x GHz %.3f\n", dname[kind], oname[order], ms,
                   tf, ghz, tf * 1e12 / (cus * 4.0 * 1024.0 * ghz * 1e9), tf / (cus * 4.0 * 1024.0) * 1e3);
        }
    }
    return 0;
}
