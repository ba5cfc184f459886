// Dense fp32 MFMA GEMM for the SwAV contractions:
//   C[m][n] = alpha * row_scale[m] * sum_k opA[m][k] * opB[k][n] + col_bias[n] + beta * C[m][n]
// used as
//   S3 projection      z      = X  Wp^T          (NT, K = C = 5376)
//   S3 prototypes      scores = rnorm .* (z Wc^T) + b   (NT, K = D = 512; normalise fused as row scale)
//   S6 dWc             = dS^T z_hat              (TN, long reduction over pixels -> split-K)
//   S6 dz_hat          = dS Wc                   (NN)
//   S6 dWp             = dz^T X                  (TN, split-K)
// Reference op chain: hfc_with_swav/swav_clustering.py:171-175 and autograd of it (:459).
#include "gx_mfma_x6.h"
#include <cstdlib>

#ifndef GX_DEFAULT_GEMM_MODE
#define GX_DEFAULT_GEMM_MODE 1      // bf16x6; 0 = exact-fp32 MFMA (env GX_GEMM_MODE / gx_set_gemm_mode)
#endif

namespace gx {

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long lda, ldb, ldc;
    float alpha, beta;
    const float* row_scale; const float* col_bias;
    int k_per_split;      // multiple of BK
    float* slab;          // split-K partials [splits][M][N] (raw sums) or nullptr
    int vec_ok;
    int tiles_m, tiles_n;
    // optional second A source, concatenated along K after the first K1 columns (A = [A | A2]);
    // K1 is a multiple of BK (K1 == K when unused).  Non-transposed A only.
    const float* A2; long lda2; int K1;
};

// Accumulators of a wave's (32 MT) x (32 NT) sub-tile whose first row / column in C is (row0, col0).
template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue_t(const GemmArgs& g, f32x16 (&acc)[MT][NT], int row0, int col0, int lane) {
    const bool to_slab = g.slab != nullptr;
    float* out = to_slab ? g.slab + (long)blockIdx.y * g.M * g.N : g.C;
    const long ldo = to_slab ? g.N : g.ldc;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = col0 + nt * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M && col < g.N) {
                    float v = acc[mt][nt][r];
                    if (!to_slab) {
                        v *= g.alpha;
                        if (g.row_scale) v *= g.row_scale[row];
                        if (g.col_bias) v += g.col_bias[col];
                        if (g.beta != 0.f) v += g.beta * out[(long)row * ldo + col];
                    }
                    out[(long)row * ldo + col] = v;
                }
            }
        }
}
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], int m0, int n0, int wr, int wc,
                                              int lane) {
    gemm_epilogue_t<2, 2>(g, acc, m0 + wr * 64, n0 + wc * 64, lane);
}

// TA: A given as [K][M] (m contiguous).  TB: B given as [N][K] (k contiguous).
// VEC: every operand is 16-B aligned with ld % 4 == 0 and (MC operands) M/N % 4 == 0, K % 4 == 0.
template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(GemmArgs g) {
    constexpr bool A_KC = !TA, B_KC = TB;
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE_FLOATS];   // A0 B0 A1 B1
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int ntile = g.tiles_m * g.tiles_n;
    const int logical = xcd_remap(blockIdx.x, ntile);
    const int tm = logical / g.tiles_n, tn = logical - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    f32x16 acc[2][2];
    zero_acc(acc);
    f32x4 ra[4], rb[4];

    auto load_a = [&](int k0) {       // scalar (wave-uniform) selects only: no divergent control flow
        const bool second = !TA && k0 >= g.K1;
        load_tile<A_KC, VEC>(second ? g.A2 : g.A, second ? g.lda2 : g.lda, m0, g.M, second ? k0 - g.K1 : k0,
                             second ? kend - g.K1 : (TA ? kend : min(kend, g.K1)), ra, tid);
    };
    const int nkt = (kend - kbeg + BK - 1) / BK;
    if (nkt > 0) {
        load_a(kbeg);
        load_tile<B_KC, VEC>(g.B, g.ldb, n0, g.N, kbeg, kend, rb, tid);
        store_tile<A_KC>(lds, ra, tid);
        store_tile<B_KC>(lds + TILE_FLOATS, rb, tid);
    }
    __syncthreads();
    for (int t = 0; t < nkt; ++t) {
        float* As = lds + (t & 1) * 2 * TILE_FLOATS;
        float* Bs = As + TILE_FLOATS;
        const bool more = (t + 1 < nkt);
        if (more) {   // issue next tile's global loads before the MFMA phase (latency hides under it)
            load_a(kbeg + (t + 1) * BK);
            load_tile<B_KC, VEC>(g.B, g.ldb, n0, g.N, kbeg + (t + 1) * BK, kend, rb, tid);
        }
        mfma_ktile<A_KC, B_KC>(As, Bs, acc, wr, wc, lane);
        if (more) {
            float* An = lds + ((t + 1) & 1) * 2 * TILE_FLOATS;
            store_tile<A_KC>(An, ra, tid);
            store_tile<B_KC>(An + TILE_FLOATS, rb, tid);
        }
        __syncthreads();
    }

    gemm_epilogue(g, acc, m0, n0, wr, wc, lane);
}

// The same contraction on the bf16 matrix cores (gx_mfma_x6.h): 128 x 128 x 16 tiles, every operand
// element split into three bf16 on its way into LDS, six MFMA products per fp32 product.
template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(GEMM_THREADS, 3) void gemm_x6_kernel(GemmArgs g) {
    constexpr bool A_KC = !TA, B_KC = TB;
    __shared__ __attribute__((aligned(16))) char lds[4 * X_TILE_BYTES];   // A0 B0 A1 B1
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int ntile = g.tiles_m * g.tiles_n;
    const int logical = xcd_remap(blockIdx.x, ntile);
    const int tm = logical / g.tiles_n, tn = logical - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);

    f32x16 acc[2][2];
    zero_acc(acc);
    f32x4 ra[2], rb[2];

    auto load_a = [&](int k0) {
        const bool second = !TA && k0 >= g.K1;
        x6_load_tile<A_KC, VEC>(second ? g.A2 : g.A, second ? g.lda2 : g.lda, m0, g.M, second ? k0 - g.K1 : k0,
                                second ? kend - g.K1 : (TA ? kend : min(kend, g.K1)), ra, tid);
    };
    const int nkt = (kend - kbeg + XK - 1) / XK;
    if (nkt > 0) {
        load_a(kbeg);
        x6_load_tile<B_KC, VEC>(g.B, g.ldb, n0, g.N, kbeg, kend, rb, tid);
        x6_store_tile<A_KC>(lds, ra, tid);
        x6_store_tile<B_KC>(lds + X_TILE_BYTES, rb, tid);
    }
    __syncthreads();
    for (int t = 0; t < nkt; ++t) {
        char* As = lds + (t & 1) * 2 * X_TILE_BYTES;
        char* Bs = As + X_TILE_BYTES;
        const bool more = (t + 1 < nkt);
        if (more) {   // next tile's global loads fly under this tile's MFMAs (a deeper, two-tile prefetch was slower)
            load_a(kbeg + (t + 1) * XK);
            x6_load_tile<B_KC, VEC>(g.B, g.ldb, n0, g.N, kbeg + (t + 1) * XK, kend, rb, tid);
        }
        x6_mfma_ktile<A_KC, B_KC>(As, Bs, acc, wr, wc, lane);
        if (more) {
            char* An = lds + ((t + 1) & 1) * 2 * X_TILE_BYTES;
            x6_store_tile<A_KC>(An, ra, tid);
            x6_store_tile<B_KC>(An + X_TILE_BYTES, rb, tid);
        }
        __syncthreads();
    }
    gemm_epilogue(g, acc, m0, n0, wr, wc, lane);
}

// Wave-specialised bf16x6 GEMM.  A 16-deep k-tile is only 768 MFMA cycles per wave — less than the
// split arithmetic, the LDS round trip and a barrier that the 4-wave kernel above strings in series
// with it (its MFMA pipe is busy 55 % of the time).  Here a workgroup of 512 threads owns a 256 x 128
// tile and its waves have fixed roles:
//   waves 0-3  consumers, 2 x 2, each a 128 x 64 sub-tile: 12 fragment reads + 48 MFMAs per k-tile,
//              nothing else;
//   waves 4-7  producers: global loads two k-tiles ahead (two register sets), exact three-way bf16
//              split, LDS stores into the slot the consumers left one barrier ago.
// One barrier per k-tile, two LDS slots: between barriers t and t+1 the consumers read slot t&1
// while the producers fill slot (t+1)&1.  One producer and one consumer wave share each SIMD, so
// the split arithmetic issues in the gaps of the other wave's MFMAs.
constexpr int WS_BM = 256, WS_BN = 128, WS_THREADS = 512;
constexpr int WS_SLOT = XImg<WS_BM>::BYTES + XImg<WS_BN>::BYTES;          // 25344 + 13056 B

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(WS_THREADS, 1) void gemm_x6ws_kernel(GemmArgs g) {
    constexpr bool A_KC = !TA, B_KC = TB;
    __shared__ __attribute__((aligned(16))) char lds[2 * WS_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = g.tiles_m * g.tiles_n;
    const int logical = xcd_remap(blockIdx.x, ntile);
    const int tm = logical / g.tiles_n, tn = logical - tm * g.tiles_n;
    const int m0 = tm * WS_BM, n0 = tn * WS_BN;
    const int kbeg = blockIdx.y * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nkt = (kend - kbeg + XK - 1) / XK;
    char* slot0 = lds;
    char* slot1 = lds + WS_SLOT;

    if (wid >= 4) {
        // ------------------------------------------------------------------ producers
        const int p = tid - 256;
        f32x4 ra0[4], rb0[2], ra1[4], rb1[2];
        auto load_tiles = [&](int t, f32x4 (&ra)[4], f32x4 (&rb)[2]) {     // tiles past kend load as zeros
            const int k0 = kbeg + t * XK;
            const bool second = !TA && k0 >= g.K1;
            x6_load_tile<A_KC, VEC, WS_BM>(second ? g.A2 : g.A, second ? g.lda2 : g.lda, m0, g.M,
                                           second ? k0 - g.K1 : k0,
                                           second ? kend - g.K1 : (TA ? kend : min(kend, g.K1)), ra, p);
            x6_load_tile<B_KC, VEC, WS_BN>(g.B, g.ldb, n0, g.N, k0, kend, rb, p);
        };
        auto stage = [&](char* slot, const f32x4 (&ra)[4], const f32x4 (&rb)[2]) {
            x6_store_tile<A_KC, WS_BM>(slot, ra, p);
            x6_store_tile<B_KC, WS_BN>(slot + XImg<WS_BM>::BYTES, rb, p);
        };
        load_tiles(0, ra0, rb0);
        load_tiles(1, ra1, rb1);
        if (nkt > 0) stage(slot0, ra0, rb0);
        __syncthreads();                                             // barrier 0: tile 0 visible
        for (int t = 0; t < nkt; t += 2) {
            load_tiles(t + 2, ra0, rb0);
            if (t + 1 < nkt) stage(slot1, ra1, rb1);
            __syncthreads();                                         // barrier t+1
            if (t + 1 >= nkt) break;
            load_tiles(t + 3, ra1, rb1);
            if (t + 2 < nkt) stage(slot0, ra0, rb0);
            __syncthreads();                                         // barrier t+2
        }
        return;
    }
    // ---------------------------------------------------------------------- consumers
    const int wr = wid >> 1, wc = wid & 1;
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    __syncthreads();                                                 // barrier 0
    for (int t = 0; t < nkt; ++t) {
        const char* slot = (t & 1) ? slot1 : slot0;
        x6_mfma_tile<A_KC, B_KC, 4, 2, WS_BM, WS_BN>(slot, slot + XImg<WS_BM>::BYTES, acc, wr * 128, wc * 64, lane);
        __syncthreads();                                             // barrier t+1
    }
    gemm_epilogue_t<4, 2>(g, acc, m0 + wr * 128, n0 + wc * 64, lane);
}

// Sum split-K slabs in a fixed order (bitwise reproducible) and apply the epilogue.
__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ slab, int splits, float* __restrict__ C,
                                          int M, int N, long ldc, float alpha, float beta,
                                          const float* __restrict__ row_scale,
                                          const float* __restrict__ col_bias) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += slab[(long)z * total + i];
        const int row = (int)(i / N), col = (int)(i - (long)row * N);
        float v = s * alpha;
        if (row_scale) v *= row_scale[row];
        if (col_bias) v += col_bias[col];
        float* c = C + (long)row * ldc + col;
        if (beta != 0.f) v += beta * *c;
        *c = v;
    }
}

// 0: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain); 1: bf16x6 on the bf16 MFMA.
static int g_gemm_mode = -1;
static int gemm_mode() {
    if (g_gemm_mode < 0) {
        const char* e = getenv("GX_GEMM_MODE");
        g_gemm_mode = (e && e[0] == '0') ? 0 : (e && e[0] == '1') ? 1 : GX_DEFAULT_GEMM_MODE;
    }
    return g_gemm_mode;
}

static int launch_gemm(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B,
                       long ldb, float* C, long ldc, float alpha, float beta, const float* row_scale,
                       const float* col_bias, int split_k, float* workspace, size_t workspace_bytes,
                       hipStream_t stream, const float* A2 = nullptr, long lda2 = 0, int K1 = -1) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N) return GX_ERR_BAD_ARG;
    GemmArgs g;
    g.A2 = A2; g.lda2 = lda2; g.K1 = (A2 ? K1 : K);
    if (A2 && (transA || K1 <= 0 || K1 >= K || (K1 % BK) != 0 || lda2 <= 0)) return GX_ERR_BAD_ARG;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta; g.row_scale = row_scale; g.col_bias = col_bias;
    g.vec_ok = (((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2) & 15) == 0 && (lda % 4 == 0) && (ldb % 4 == 0) &&
               (lda2 % 4 == 0) && lda < (1 << 22) && ldb < (1 << 22) && lda2 < (1 << 22) && (K % 4 == 0) && (g.K1 % 4 == 0) && (!transA || M % 4 == 0) &&
               (transB || N % 4 == 0);
    const bool x6 = gemm_mode() == 1;
    // wave-specialised 256 x 128 tiles (one workgroup per CU) pay off once the k loop is long enough
    // to amortise a tile's prologue / epilogue, which nothing overlaps there
    const bool ws = x6 && M > BM && (split_k > 1 ? K / split_k : K) >= 768;
    g.tiles_m = gx_cdiv(M, ws ? WS_BM : BM); g.tiles_n = gx_cdiv(N, BN);
    const int ktiles = gx_cdiv(K, BK);
    if (split_k < 1) split_k = 1;
    if (split_k > ktiles) split_k = ktiles;
    g.k_per_split = gx_cdiv(ktiles, split_k) * BK;
    split_k = gx_cdiv(K, g.k_per_split);
    g.slab = nullptr;
    if (split_k > 1) {
        if (!workspace || workspace_bytes < (size_t)split_k * M * N * sizeof(float)) return GX_ERR_WORKSPACE;
        g.slab = workspace;
    }
    dim3 grid(g.tiles_m * g.tiles_n, split_k), block(GEMM_THREADS);
#define GX_LAUNCH_GEMM(TA_, TB_)                                                                             \
    do {                                                                                                       \
        if (ws && g.vec_ok) hipLaunchKernelGGL((gemm_x6ws_kernel<TA_, TB_, true>), grid, dim3(WS_THREADS), 0, stream, g); \
        else if (ws) hipLaunchKernelGGL((gemm_x6ws_kernel<TA_, TB_, false>), grid, dim3(WS_THREADS), 0, stream, g); \
        else if (x6 && g.vec_ok) hipLaunchKernelGGL((gemm_x6_kernel<TA_, TB_, true>), grid, block, 0, stream, g);  \
        else if (x6) hipLaunchKernelGGL((gemm_x6_kernel<TA_, TB_, false>), grid, block, 0, stream, g);        \
        else if (g.vec_ok) hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, true>), grid, block, 0, stream, g); \
        else hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, false>), grid, block, 0, stream, g);              \
    } while (0)
    if (!transA && transB) GX_LAUNCH_GEMM(false, true);
    else if (!transA && !transB) GX_LAUNCH_GEMM(false, false);
    else if (transA && !transB) GX_LAUNCH_GEMM(true, false);
    else GX_LAUNCH_GEMM(true, true);
#undef GX_LAUNCH_GEMM
    GX_CHECK_LAUNCH();
    if (split_k > 1) {
        const long total = (long)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, g.slab, split_k, C, M,
                           N, ldc, alpha, beta, row_scale, col_bias);
        GX_CHECK_LAUNCH();
    }
    return GX_OK;
}

}  // namespace gx

extern "C" {

// Contraction engine of gx_gemm_f32 / gx_gemm_f32_cat: 0 = exact-fp32 MFMA, 1 = bf16x6 (fp32-level
// error on the 16x faster bf16 matrix cores).  Returns the previous mode.  Env GX_GEMM_MODE=0|1 sets
// the initial value.
int gx_set_gemm_mode(int mode) {
    const int prev = gx::gemm_mode();
    if (mode == 0 || mode == 1) gx::g_gemm_mode = mode;
    return prev;
}
int gx_get_gemm_mode(void) { return gx::gemm_mode(); }

size_t gx_gemm_f32_workspace_bytes(int M, int N, int split_k) {
    return split_k > 1 ? (size_t)split_k * M * N * sizeof(float) : 0;
}

int gx_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                float* C, long ldc, float alpha, float beta, const float* row_scale, const float* col_bias,
                int split_k, float* workspace, size_t workspace_bytes, void* stream) {
    return gx::launch_gemm(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, row_scale, col_bias,
                           split_k, workspace, workspace_bytes, (hipStream_t)stream);
}

// C = alpha * [A1 | A2] opB + ... : A1 is [M][K1], A2 is [M][K2]; K1 must be a multiple of 32.
// (two same-resolution feature taps contracted against adjacent column blocks of the
// projection weight in one launch)
int gx_gemm_f32_cat(int transB, int M, int N, int K1, const float* A1, long lda1, int K2, const float* A2, long lda2,
                    const float* B, long ldb, float* C, long ldc, float alpha, float beta, const float* col_bias,
                    void* stream) {
    if (!A2 || K2 <= 0) return GX_ERR_BAD_ARG;
    return gx::launch_gemm(0, transB, M, N, K1 + K2, A1, lda1, B, ldb, C, ldc, alpha, beta, nullptr, col_bias, 1,
                           nullptr, 0, (hipStream_t)stream, A2, lda2, K1);
}

}  // extern "C"
