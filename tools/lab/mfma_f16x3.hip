// Would fp32 = TWO fp16 planes (22 significant bits + the fp16 subnormal tail) with THREE products
//     a b = a_h b_h + (a_h b_l + a_l b_h) + O(2^-22 |a||b|) worst case, ~2^-25 typical (the dropped a_l b_l)
// do for the contractions of this path what bf16x6 does with six?  v_mfma_f32_16x16x32_f16 runs at the bf16 rate, so the
// matrix-core work per fp32 product halves — IF (1) the matrix cores take fp16 SUBNORMAL inputs exactly (a_l of an element
// below 2^-2 is subnormal: flushed, the scheme is a 12-bit one for small elements), (2) the measured error of a long
// contraction is fp32-class, (3) the power-limited rate is the bf16 one.  This lab answers the three on the chip.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16x3 tools/lab/mfma_f16x3.hip && /tmp/mfma_f16x3
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

// ---- (1) + (2): one wave, out[16][16] = A[16][K] * B[16][K]^T, three ways ------------------------------------------------
// mode 0: f16x3 (two planes, scaled by `scale` on the way in, 2^-2s on the way out); 1: bf16x6; 2: f16 single plane
__global__ __launch_bounds__(64) void gemm16_kernel(const float* __restrict__ A, const float* __restrict__ B, int K, int mode,
                                                    float scale, float* __restrict__ out) {
    const int lane = threadIdx.x, row = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        if (mode == 0 || mode == 2) {
            // h x h over 32 k: lane group g holds k0 + 8 g .. + 7
            f16x8 ah, bh;
            for (int e = 0; e < 8; ++e) {
                ah[e] = (_Float16)(A[row * K + k0 + 8 * g + e] * scale);
                bh[e] = (_Float16)(B[row * K + k0 + 8 * g + e] * scale);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
            if (mode == 0)
                for (int half = 0; half < 2; ++half) {       // [a_h | a_l] x [b_l | b_h] over 16 k
                    f16x8 a, b;
                    for (int e = 0; e < 8; ++e) {
                        const int k = k0 + 16 * half + 8 * (g & 1) + e;
                        const float av = A[row * K + k] * scale, bv = B[row * K + k] * scale;
                        const _Float16 a_h = (_Float16)av, b_h = (_Float16)bv;
                        const _Float16 a_l = (_Float16)(av - (float)a_h), b_l = (_Float16)(bv - (float)b_h);
                        a[e] = g < 2 ? a_h : a_l;
                        b[e] = g < 2 ? b_l : b_h;
                    }
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
                }
        } else {
            // bf16x6 in the 16x16x32 pairing of gx_conv.hip: [h|l]x[l|h], [h|m]x[m|h], [h|m]x[h|m] per 16 k
            for (int half = 0; half < 2; ++half) {
                bf16x8 p[3][2];                              // [product][a / b]
                for (int e = 0; e < 8; ++e) {
                    const int k = k0 + 16 * half + 8 * (g & 1) + e;
                    const float av = A[row * K + k], bv = B[row * K + k];
                    const __bf16 a0 = (__bf16)av, b0 = (__bf16)bv;
                    const float ar = av - (float)a0, br = bv - (float)b0;
                    const __bf16 a1 = (__bf16)ar, b1 = (__bf16)br;
                    const __bf16 a2 = (__bf16)(ar - (float)a1), b2 = (__bf16)(br - (float)b1);
                    p[0][0][e] = g < 2 ? a0 : a2; p[0][1][e] = g < 2 ? b2 : b0;
                    p[1][0][e] = g < 2 ? a0 : a1; p[1][1][e] = g < 2 ? b1 : b0;
                    p[2][0][e] = g < 2 ? a0 : a1; p[2][1][e] = g < 2 ? b0 : b1;
                }
                for (int q = 0; q < 3; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p[q][0], p[q][1], acc, 0, 0, 0);
            }
        }
    }
    const float inv = (mode == 1) ? 1.f : 1.f / (scale * scale);
    // D layout of 16x16: column = lane & 15 (B row), row = 4 (lane >> 4) + r (A row)
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + row] = acc[r] * inv;
}

// ---- (3): sustained rate, operands in registers ---------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(256) void rate_kernel(const uint4* __restrict__ frag, float* __restrict__ out, long iters,
                                                   unsigned long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 a[4][2], b[4][2];
    const uint4* f = frag + ((size_t)(blockIdx.x & 63) * 4 + wave) * 16 * 64 + lane;
    for (int t = 0; t < 4; ++t)
        for (int p = 0; p < 2; ++p) { a[t][p] = f[(t * 2 + p) * 64]; b[t][p] = f[(8 + t * 2 + p) * 64]; }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[i][p & 1]), __builtin_bit_cast(f16x8, b[j][p >> 1]), acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][p & 1]), __builtin_bit_cast(bf16x8, b[j][p >> 1]), acc[i][j], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// the same with the 32x32x16 shape (what the f16x3 kernels issue): 2 x 2 tiles, three products per step
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <bool F16>
__global__ __launch_bounds__(256) void rate32_kernel(const uint4* __restrict__ frag, float* __restrict__ out, long iters,
                                                     unsigned long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 a[2][2], b[2][2];
    const uint4* f = frag + ((size_t)(blockIdx.x & 63) * 4 + wave) * 16 * 64 + lane;
    for (int t = 0; t < 2; ++t)
        for (int p = 0; p < 2; ++p) { a[t][p] = f[(t * 2 + p) * 64]; b[t][p] = f[(8 + t * 2 + p) * 64]; }
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    unsigned long long t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][p & 1]), __builtin_bit_cast(f16x8, b[j][p >> 1]), acc[i][j], 0, 0, 0);
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][p & 1]), __builtin_bit_cast(bf16x8, b[j][p >> 1]), acc[i][j], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][7];
    if (s == 123.456f) out[threadIdx.x] = s;
}

static uint16_t bf16_of(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float gauss() {
    float u1 = (rand() + 1.f) / (RAND_MAX + 2.f), u2 = rand() / (float)RAND_MAX;
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    float *dA, *dB, *dO;
    const int KMAX = 8192;
    CK(hipMalloc(&dA, 16 * KMAX * 4));
    CK(hipMalloc(&dB, 16 * KMAX * 4));
    CK(hipMalloc(&dO, 256 * 4));
    std::vector<float> A(16 * KMAX), B(16 * KMAX), O(256);
    // (1) subnormal inputs: A = 2^-20 (an fp16 subnormal: 16 units of 2^-24), B = 2^10; K = 32: exact answer 32 * 2^-10
    {
        for (auto& v : A) v = ldexpf(1.f, -20);
        for (auto& v : B) v = ldexpf(1.f, 10);
        CK(hipMemcpy(dA, A.data(), 16 * 32 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 16 * 32 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(gemm16_kernel, dim3(1), dim3(64), 0, 0, dA, dB, 32, 2, 1.0f, dO);
        CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
        printf("(1) subnormal fp16 A (2^-20) x B (2^10), K = 32: got %.9g, exact %.9g -> %s\n", O[0], 32.0 * ldexp(1.0, -10),
               O[0] == (float)(32.0 * ldexp(1.0, -10)) ? "subnormal inputs are NOT flushed" : "FLUSHED / inexact");
        // the smallest subnormal, 2^-24, times 1
        for (auto& v : A) v = ldexpf(1.f, -24);
        for (auto& v : B) v = 1.f;
        CK(hipMemcpy(dA, A.data(), 16 * 32 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 16 * 32 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(gemm16_kernel, dim3(1), dim3(64), 0, 0, dA, dB, 32, 2, 1.0f, dO);
        CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
        printf("    2^-24 x 1, K = 32: got %.9g, exact %.9g\n", O[0], 32.0 * ldexp(1.0, -24));
    }
    // (2) error of long contractions against fp64, by element-magnitude profile and scale
    srand(11);
    struct Case { const char* name; float sa, sb; int heavy; };
    const Case cases[] = {{"N(0,1) x N(0,1)", 1.f, 1.f, 0}, {"N(0,1) x N(0,0.02)", 1.f, 0.02f, 0}, {"N(0,30) x N(0,1)", 30.f, 1.f, 0},
                          {"heavy-tailed (x1000 outliers 0.1 %) x N(0,0.05)", 1.f, 0.05f, 1}, {"N(0,0.001) x N(0,0.001)", 1e-3f, 1e-3f, 0}};
    for (const Case& c : cases)
        for (int K : {512, 4608}) {
            for (int i = 0; i < 16 * K; ++i) {
                A[i] = gauss() * c.sa * ((c.heavy && rand() % 1000 == 0) ? 1000.f : 1.f);
                B[i] = gauss() * c.sb;
            }
            CK(hipMemcpy(dA, A.data(), 16 * K * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, B.data(), 16 * K * 4, hipMemcpyHostToDevice));
            std::vector<double> ref(256), nrm(256);
            std::vector<float> f32(256);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double s = 0, n = 0;
                    float fs = 0.f;
                    for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[j * K + k]; n += fabs((double)A[i * K + k] * B[j * K + k]); fs = fmaf(A[i * K + k], B[j * K + k], fs); }
                    ref[i * 16 + j] = s; nrm[i * 16 + j] = n; f32[i * 16 + j] = fs;
                }
            auto err = [&](const float* o, double& nw, double& cw) {           // norm-wise (Frobenius) and worst |err| / sum |a b|
                double se = 0, sr = 0; cw = 0;
                for (int i = 0; i < 256; ++i) { const double e = o[i] - ref[i]; se += e * e; sr += ref[i] * ref[i]; cw = fmax(cw, fabs(e) / nrm[i]); }
                nw = sqrt(se / sr);
            };
            double nw, cw;
            err(f32.data(), nw, cw);
            printf("(2) %-48s K %4d  fp32 fma chain: %.2e (%.1e)", c.name, K, nw, cw);
            hipLaunchKernelGGL(gemm16_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, 1, 1.f, dO);
            CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
            err(O.data(), nw, cw);
            printf("  bf16x6: %.2e (%.1e)", nw, cw);
            for (float sc : {1.f, 16.f, 256.f}) {
                hipLaunchKernelGGL(gemm16_kernel, dim3(1), dim3(64), 0, 0, dA, dB, K, 0, sc, dO);
                CK(hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost));
                err(O.data(), nw, cw);
                printf("  f16x3 x%g: %.2e (%.1e)", sc, nw, cw);
            }
            printf("\n");
        }
    // (3) sustained rate: 48 x 16x16x32 per step and wave, 2 waves per SIMD
    const int grid = prop.multiProcessorCount * 2;
    const size_t nfrag = (size_t)64 * 4 * 16 * 64;
    std::vector<uint16_t> hf(nfrag * 8), hb(nfrag * 8);
    for (size_t i = 0; i < nfrag * 8; ++i) {
        const float x = gauss();
        _Float16 h = (_Float16)x;
        memcpy(&hf[i], &h, 2);
        hb[i] = bf16_of(x);
    }
    void *df, *db;
    unsigned long long* clk;
    CK(hipMalloc(&df, nfrag * 16));
    CK(hipMalloc(&db, nfrag * 16));
    CK(hipMalloc(&clk, 16));
    CK(hipMemcpy(df, hf.data(), nfrag * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), nfrag * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long iters = 60000;
    for (int rep = 0; rep < 3; ++rep)
        for (int eng = 0; eng < 4; ++eng) {
            for (int warm = 0; warm < 2; ++warm) {
                CK(hipEventRecord(e0));
                if (eng == 0) hipLaunchKernelGGL(rate_kernel<false>, dim3(grid), dim3(256), 0, 0, (const uint4*)db, dO, iters, clk);
                else if (eng == 1) hipLaunchKernelGGL(rate_kernel<true>, dim3(grid), dim3(256), 0, 0, (const uint4*)df, dO, iters, clk);
                else if (eng == 2) hipLaunchKernelGGL(rate32_kernel<false>, dim3(grid), dim3(256), 0, 0, (const uint4*)db, dO, 2 * iters, clk);
                else hipLaunchKernelGGL(rate32_kernel<true>, dim3(grid), dim3(256), 0, 0, (const uint4*)df, dO, 2 * iters, clk);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long hc[2];
            CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
            const double ghz = (double)hc[0] / ((double)hc[1] / 100e6) / 1e9;
            // 48 x 16x16x32 (16384 FLOP each) per step, or 12 x 32x32x16 (32768 FLOP each) over twice the steps: the same FLOPs
            const double flops = 2.0 * 16 * 16 * 32 * 48.0 * iters * 4.0 * grid;
            const double busy = 2.0 * 48.0 * 8.0 * iters / (double)hc[0];        // 8 cycles per 16x16x32 (16 per 32x32x16); two waves share a SIMD
            const char* names[4] = {"16x16x32 bf16", "16x16x32 f16", "32x32x16 bf16", "32x32x16 f16"};
            printf("(3) %-22s %8.2f ms  %8.1f TFLOP/s  clock %.3f GHz  busy %.3f  busy x GHz %.3f\n",
                   names[eng], ms, flops / ms / 1e9, ghz, busy, busy * ghz);
        }
    return 0;
}
