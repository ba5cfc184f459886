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
#include <cmath>
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
    // optional second B source, concatenated along N after the first N1 columns (B = [B1 | B2], both [K][.], n
    // contiguous); N1 is a multiple of BN, so a tile's columns come from one source (N1 == N when unused)
    const float* B2; long ldb2; int N1;
    // optional epilogue term C[m][n] += up_src[coarse(m)][n]: C's rows are the pixels (b, y, x) of up_rf x up_rf
    // images (up_rf = 1 << up_shift), up_src's the pixels (b, y / 2, x / 2) of the half-resolution table [.][N]
    // (the upsample-add of the pyramid projection, fused).  Not with split-K.
    const float* up_src; int up_shift;
    const unsigned short* Bp; unsigned Bp_bytes;   // pre-split B planes (gemm_xw_kernel), gx_gemm_weight_planes_bytes
    int n_group;          // gemm_xw_kernel: N tiles per group of the tile order (xw_tile_of)
    float* colmax_part;   // gemm_xw_kernel: [4 tiles_m][N] maxima of the written values over each wave's 32 rows, or nullptr
    float* usum_part;     // ... and (with colmax_part) [4 tiles_m][N] sums of exp((value - that maximum) * sk_inv_eps) over the
    float sk_inv_eps;     //     same rows: the first Sinkhorn pass, left behind by the GEMM that writes the scores
    int ep_vec;           // C / slab / up_src rows can be written and read as aligned float4 (N % 4 == 0, ...)
    // gemm_xw_kernel<.., FIN>: the low-rank path's dense finish in the epilogue (gx_gemm_f32_nt_planes_finish).  C's rows are
    // the pixels of (1 << fin_rlog)^2 images; a tile is 2 image rows x 64 pixels, a wave 2 x 16 of them:
    //   C[q] = acc[q] + sum_k fin_C[q][k] fin_Wc[k] + fin_C[q][fin_K] fin_Z[q];  fin_pool[q / 2x2] = the 2x2 sums of C;
    //   fin_part[tile_m][k][n] = sum over the tile's 128 pixels of fin_C[q][k] fin_Z[q][n]
    const float* fin_C; const float* fin_Z; const float* fin_Wc;
    float* fin_pool; float* fin_part;
    int fin_K, fin_ls, fin_rlog;
    // optional device scalar: the rows of the LONG dimension that carry data — the M rows of gemm_xw_kernel (tiles at or past
    // it return at once), the k rows of a TN contraction (its k range is cut there).  The many-prototype tables hold the drawn
    // pixels first and their count is only known on the device: no host sync, the launch covers the capacity (DESIGN par. 4.15)
    const int* live;
    // f16x3 form (gx_mfma_x6.h): the A elements are multiplied by a_scale (a power of two) before their two-plane fp16 split;
    // the B planes are two fp16 planes of B * xh_scale_of(*b_amax) (the trailer gx_split_gemm_weights_h2 leaves behind the planes)
    float a_scale; const float* b_amax;
    float b_scale;        // x6ws: B's host-given power-of-two scale
    // x6ws: A's bound on the DEVICE (max |A| left behind by A's producer — fin_amax below —, times a host factor); NULL: a_scale
    // (gemm_xw_kernel<.., H2> reads the same pair: A = a table some kernel left its maximum of — Zc, Tt)
    const float* a_amax; float a_amax_mul;
    const float* b_amax_ws; float b_amax_mul;   // x6ws: the same for B (NULL: b_scale)
    unsigned* fin_amax;   // gemm_xw_kernel<.., FIN>: optional, max |G| over everything the launch writes (float bits, atomicMax)
};

// (tile, K-split) of this workgroup.  The grid is (tiles, splits) and workgroups are dealt to the 8 XCDs in dispatch
// order (x fastest), so `blockIdx.x -> tile, blockIdx.y -> split` put the tiles of ONE split — which read the same
// K-slice of A and of B — on different XCDs: every XCD's L2 then fetched every slice itself (the pyramid-backward TN
// GEMMs, 512 x C_l outputs = 4-8 tiles x ~100 splits, moved 2.3 x their operands).  Here the dispatch index is
// remapped as a whole, split-major / tile-minor: an XCD owns a contiguous run of (split, tile) pairs, so the tiles
// of a split run side by side on one L2.
__device__ __forceinline__ void gemm_tile_of_block(int ntile, int& logical, int& split) {
    const int L = xcd_remap((int)(blockIdx.y * gridDim.x + blockIdx.x), ntile * (int)gridDim.y);
    split = L / ntile;
    logical = L - split * ntile;
}

// The k range of one split.  With a device-side bound (g.live: only the first *g.live of the K rows hold anything) the LIVE
// range is what the launched splits share — cutting only the tail splits short left their workgroups idle while the others
// ran as long as before (TN dWc over the score tables, 0.84 live: 31.2 -> 29.2 ms instead of -> 26).
__device__ __forceinline__ void gemm_k_range(const GemmArgs& g, int split, int& kbeg, int& kend) {
    int K = g.K, kps = g.k_per_split;
    if (g.live) {
        K = min(g.K, max(0, *g.live));
        const int kt = (K + BK - 1) / BK, ns = (int)gridDim.y;
        kps = (kt + ns - 1) / ns * BK;
    }
    kbeg = split * kps;
    kend = min(K, kbeg + kps);
}

// Accumulators of a wave's (32 MT) x (32 NT) sub-tile whose first row / column in C is (row0, col0).
template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue_t(const GemmArgs& g, f32x16 (&acc)[MT][NT], int row0, int col0, int lane,
                                                int split) {
#pragma clang fp contract(off)      // one rounding per written operation: every epilogue form gives the same bits
    const bool to_slab = g.slab != nullptr;
    float* out = to_slab ? g.slab + (long)split * g.M * g.N : g.C;
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
                        if (g.up_src) {
                            const int s = g.up_shift, rc = 1 << (s - 1);
                            const int b = row >> (2 * s), y = (row >> s) & ((1 << s) - 1), x = row & ((1 << s) - 1);
                            v += g.up_src[(((long)b * rc + (y >> 1)) * rc + (x >> 1)) * g.N + col];
                        }
                    }
                    out[(long)row * ldo + col] = v;
                }
            }
        }
}
// The same epilogue through LDS: a wave parks 32 rows of its sub-tile in its own staging area (the operand
// tiles are dead by now), reads them back row-contiguous and issues 16-B loads / stores — 256 contiguous bytes per
// row segment instead of 64 dword stores (and as many dword loads for beta / up_src) per lane.
constexpr int EP_STAGE_FLOATS = 32 * (64 + 4);           // per wave: 32 rows x (32 NT + 4) floats, NT <= 2
// The epilogue-only fields of GemmArgs, re-read from the kernel-argument segment when the epilogue starts: read at
// kernel entry they would sit in registers through the whole k loop (the compiler spilled them to scratch and
// re-loaded them, with a full vmcnt(0) wait, once per store pass).  GemmArgs is the kernels' only parameter.
__device__ __forceinline__ const GemmArgs* gemm_late_args() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const GemmArgs*)p;
}
template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs& g, f32x16 (&acc)[MT][NT], int row0, int col0, int lane,
                                                     float* stage, int split) {
#pragma clang fp contract(off)      // one rounding per written operation: every epilogue form gives the same bits
    constexpr int SW = 32 * NT + 4, LPR = 8 * NT, RPP = 64 / LPR;
    const GemmArgs& q = *gemm_late_args();
    const float alpha = q.alpha, beta = q.beta;
    const float* row_scale = q.row_scale;
    const float* up_src = q.up_src;
    const int up_s = q.up_shift;
    const bool to_slab = g.slab != nullptr;
    float* out = to_slab ? g.slab + (long)split * g.M * g.N : g.C;
    const long ldo = to_slab ? g.N : g.ldc;
    const int rl = lane / LPR, c4 = (lane % LPR) * 4;
    const int col = col0 + c4;
    f32x4 cb = zero4();
    if (!to_slab && q.col_bias && col < g.N) cb = f32x4{q.col_bias[col], q.col_bias[col + 1], q.col_bias[col + 2], q.col_bias[col + 3]};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SW + nt * 32 + (lane & 31)] = acc[mt][nt][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int row = row0 + mt * 32 + ps * RPP + rl;
            if (row < g.M && col < g.N) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + (ps * RPP + rl) * SW + c4);
                float* dst = out + (long)row * ldo + col;
                if (!to_slab) {
                    v *= alpha;
                    if (row_scale) v *= row_scale[row];
                    v += cb;
                    if (beta != 0.f) v += beta * *reinterpret_cast<const f32x4*>(dst);
                    if (up_src) {
                        const int s = up_s, rc = 1 << (s - 1);
                        const int b = row >> (2 * s), y = (row >> s) & ((1 << s) - 1), x = row & ((1 << s) - 1);
                        v += *reinterpret_cast<const f32x4*>(up_src + (((long)b * rc + (y >> 1)) * rc + (x >> 1)) * g.N + col);
                    }
                }
                *reinterpret_cast<f32x4*>(dst) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], int m0, int n0, int wr, int wc,
                                              int lane, float* stage, int split) {
    if (g.ep_vec) gemm_epilogue_staged<2, 2>(g, acc, m0 + wr * 64, n0 + wc * 64, lane, stage, split);
    else gemm_epilogue_t<2, 2>(g, acc, m0 + wr * 64, n0 + wc * 64, lane, split);
}

// TA: A given as [K][M] (m contiguous).  TB: B given as [N][K] (k contiguous).
// VEC: every operand is 16-B aligned with ld % 4 == 0 and (MC operands) M/N % 4 == 0, K % 4 == 0.
template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(GemmArgs g) {
    constexpr bool A_KC = !TA, B_KC = TB;
    __shared__ __attribute__((aligned(16))) float lds[4 * TILE_FLOATS];   // A0 B0 A1 B1
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int ntile = g.tiles_m * g.tiles_n;
    int logical, split;
    gemm_tile_of_block(ntile, logical, split);
    const int tm = logical / g.tiles_n, tn = logical - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const bool bsec = n0 >= g.N1;                                    // [B1 | B2]: this tile's columns are B2's
    const float* Bsrc = bsec ? g.B2 : g.B;
    const long ldbs = bsec ? g.ldb2 : g.ldb;
    const int nb0 = bsec ? n0 - g.N1 : n0, nbN = bsec ? g.N - g.N1 : g.N1;
    int kbeg, kend;
    gemm_k_range(g, split, kbeg, kend);

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
        load_tile<B_KC, VEC>(Bsrc, ldbs, nb0, nbN, kbeg, kend, rb, tid);
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
            load_tile<B_KC, VEC>(Bsrc, ldbs, nb0, nbN, kbeg + (t + 1) * BK, kend, rb, tid);
        }
        mfma_ktile<A_KC, B_KC>(As, Bs, acc, wr, wc, lane);
        if (more) {
            float* An = lds + ((t + 1) & 1) * 2 * TILE_FLOATS;
            store_tile<A_KC>(An, ra, tid);
            store_tile<B_KC>(An + TILE_FLOATS, rb, tid);
        }
        __syncthreads();
    }

    gemm_epilogue(g, acc, m0, n0, wr, wc, lane, lds + wid * EP_STAGE_FLOATS, split);
}

// The same contraction on the bf16 matrix cores (gx_mfma_x6.h): 128 x 128 x 16 tiles, every operand
// element split into three bf16 on its way into LDS, six MFMA products per fp32 product.
template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(GEMM_THREADS, (!TA && TB) ? 3 : 2) void gemm_x6_kernel(GemmArgs g) {   // (an m-contiguous operand image is 15 KB: two workgroups per CU)
    constexpr bool A_KC = !TA, B_KC = TB;
    constexpr int X_TILE_BYTES = (A_KC && B_KC) ? gx::X_TILE_BYTES : X_TILE_BYTES_MC;     // (shadows the KC-only size)
    __shared__ __attribute__((aligned(16))) char lds[4 * X_TILE_BYTES];   // A0 B0 A1 B1
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    const int ntile = g.tiles_m * g.tiles_n;
    int logical, split;
    gemm_tile_of_block(ntile, logical, split);
    const int tm = logical / g.tiles_n, tn = logical - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const bool bsec = n0 >= g.N1;                                    // [B1 | B2]: this tile's columns are B2's
    const float* Bsrc = bsec ? g.B2 : g.B;
    const long ldbs = bsec ? g.ldb2 : g.ldb;
    const int nb0 = bsec ? n0 - g.N1 : n0, nbN = bsec ? g.N - g.N1 : g.N1;
    int kbeg, kend;
    gemm_k_range(g, split, kbeg, kend);

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
        x6_load_tile<B_KC, VEC>(Bsrc, ldbs, nb0, nbN, kbeg, kend, rb, tid);
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
            x6_load_tile<B_KC, VEC>(Bsrc, ldbs, nb0, nbN, kbeg + (t + 1) * XK, kend, rb, tid);
        }
        x6_mfma_ktile<A_KC, B_KC>(As, Bs, acc, wr, wc, lane);
        if (more) {
            char* An = lds + ((t + 1) & 1) * 2 * X_TILE_BYTES;
            x6_store_tile<A_KC>(An, ra, tid);
            x6_store_tile<B_KC>(An + X_TILE_BYTES, rb, tid);
        }
        __syncthreads();
    }
    gemm_epilogue(g, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(lds) + wid * EP_STAGE_FLOATS, split);
}

// Epilogue of gemm_xw_kernel for beta == 0: a wave's 32 rows x 64 columns through its LDS staging area, as
// gemm_epilogue_staged<1, 2>, but with every global LOAD the 8 store passes need — the coarse rows of the fused
// upsample-add, the row scales — issued up front: issued pass by pass behind the wave barriers each one cost a full
// memory latency before its store (16 dependent round trips per tile).
template <bool USUM>
__device__ __forceinline__ void xw_epilogue(const GemmArgs& g, f32x16 (&acc)[1][2], int row0, int col0, int lane, float* stage,
                                            long part_row) {
#pragma clang fp contract(off)      // one rounding per written operation: every epilogue form gives the same bits
    constexpr int SW = 68;
    const GemmArgs& q = *gemm_late_args();
    const float alpha = q.alpha;
    const float* row_scale = q.row_scale;
    const float* up_src = q.up_src;
    const int rl = lane >> 4, c4 = (lane & 15) * 4;
    const int col = col0 + c4;
    const bool colok = col < g.N;
    const int colc = colok ? col : 0;
    f32x4 cb = zero4();
    if (q.col_bias && colok) cb = *reinterpret_cast<const f32x4*>(q.col_bias + col);
    f32x4 up[8];
    float rs[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) { up[ps] = zero4(); rs[ps] = 1.f; }
    if (up_src) {
        const int s = q.up_shift, rc = 1 << (s - 1);
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int row = min(row0 + ps * 4 + rl, g.M - 1);
            const int b = row >> (2 * s), y = (row >> s) & ((1 << s) - 1), x = row & ((1 << s) - 1);
            up[ps] = *reinterpret_cast<const f32x4*>(up_src + (((long)b * rc + (y >> 1)) * rc + (x >> 1)) * g.N + colc);
        }
    }
    if (row_scale) {
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) rs[ps] = row_scale[min(row0 + ps * 4 + rl, g.M - 1)];
    }
#ifdef XW_ABL_DIRECT
    // lab (WRONG values, the right store pattern): what an epilogue that stores the accumulators straight from registers — lane =
    // row, four consecutive columns per register quad, no LDS staging — would cost; tools/lab/sc_lab.py
    if (!up_src && !row_scale && !q.colmax_part) {
        const int row = row0 + (lane & 31);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cc = col0 + nt * 32 + 8 * g4 + 4 * (lane >> 5);
                f32x4 v = {acc[0][nt][4 * g4], acc[0][nt][4 * g4 + 1], acc[0][nt][4 * g4 + 2], acc[0][nt][4 * g4 + 3]};
                v *= alpha;
                v += cb;
                if (row < g.M && cc < g.N) *reinterpret_cast<f32x4*>(g.C + (long)row * g.ldc + cc) = v;
            }
        return;
    }
#endif
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SW + nt * 32 + (lane & 31)] = acc[0][nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 cm = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int row = row0 + ps * 4 + rl;
        f32x4 v = *reinterpret_cast<const f32x4*>(stage + (ps * 4 + rl) * SW + c4);
        v *= alpha;
        v *= rs[ps];
        v += cb;
        v += up[ps];
        if (row < g.M && colok) {
            *reinterpret_cast<f32x4*>(g.C + (long)row * g.ldc + col) = v;
#pragma unroll
            for (int e = 0; e < 4; ++e) cm[e] = fmaxf(cm[e], v[e]);
        }
    }
    if (q.colmax_part) {      // column maxima of what this wave wrote (the Sinkhorn stabiliser: no pass of its own over C)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cm[e] = fmaxf(cm[e], __shfl_xor(cm[e], 16));
            cm[e] = fmaxf(cm[e], __shfl_xor(cm[e], 32));
        }
        if (lane < 16 && colok) *reinterpret_cast<f32x4*>(q.colmax_part + part_row * g.N + col) = cm;
        if (USUM) {           // ... and the first Sinkhorn pass over these 32 rows, relative to their own maxima (merged later
                              // by log-sum-exp: gx_sinkhorn_first_from_parts): one sweep of the score matrix fewer.  The
                              // values are rebuilt from the staging area (same operations, same bits) instead of being
                              // held in registers: the kernel keeps its four workgroups per CU
            float* const usum_part = q.usum_part;
            const float ie = q.sk_inv_eps * 1.44269504088896341f;
            f32x4 us = zero4();
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                f32x4 v = *reinterpret_cast<const f32x4*>(stage + (ps * 4 + rl) * SW + c4);
                v *= alpha;
                v *= rs[ps];
                v += cb;
                v += up[ps];
                if (row0 + ps * 4 + rl < g.M && colok)
#pragma unroll
                    for (int e = 0; e < 4; ++e) us[e] += __builtin_amdgcn_exp2f((v[e] - cm[e]) * ie);      // v_exp_f32: arguments <= 0, no range fix-up
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                us[e] += __shfl_xor(us[e], 16);
                us[e] += __shfl_xor(us[e], 32);
            }
            if (lane < 16 && colok) *reinterpret_cast<f32x4*>(usum_part + part_row * g.N + col) = us;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Epilogue of gemm_xw_kernel<false, true>: the dense finish of the low-rank prototype path (what proto_lowrank_finish_kernel
// does in a pass of its own over G: read G, Z, write G, Gpool — 3.4 GB for the 128^2 level of the bench step) while the tile
// is on the chip.  The wave's 32 rows are 2 image rows x 16 pixels (`pw` = its first pixel), so the 2x2 sums close inside
// the wave; its coefficient rows sit in LDS (`cst`, [32][ls]); the prototype-gradient partial of the workgroup's 128 pixels
// is summed across the four waves through the staging areas in a fixed order.
// pixel of store pass ps of a lane whose first pixel is pl: image row ps >> 2 of the pair, x = (ps & 3) * 4 + (lane >> 4)
#define FIN_PIX(ps) (pl + ((long)((ps) >> 2) << rlog) + ((ps) & 3) * 4)
__device__ __forceinline__ void xw_fin_load_z(const GemmArgs& q, long N, long pl, int col, f32x4 (&z)[8]) {
    const int rlog = q.fin_rlog;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) z[ps] = *reinterpret_cast<const f32x4*>(q.fin_Z + FIN_PIX(ps) * N + col);
}
__device__ __forceinline__ void xw_epilogue_fin(const GemmArgs& g, f32x16 (&acc)[1][2], const f32x4 (&z)[8], int tm, long pw,
                                                int col0, int lane, float* stage, const float* cst, const float* stage_all) {
    constexpr int SW = 68;
    const GemmArgs& q = *gemm_late_args();
    const int K = q.fin_K, ls = q.fin_ls, rlog = q.fin_rlog;
    const int rl = lane >> 4, c4 = (lane & 15) * 4, col = col0 + c4;          // N % 128 == 0: every column is live
    const long N = g.N;
    const long pl = pw + rl;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SW + nt * 32 + (lane & 31)] = acc[0][nt][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 v[8];
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        v[ps] = *reinterpret_cast<const f32x4*>(stage + (ps * 4 + rl) * SW + c4);
        v[ps] += cst[(ps * 4 + rl) * ls + K] * z[ps];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                     // the staging area is free: it takes the wave's gWc partial [K][64]
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const float* const Wc = q.fin_Wc;
    f32x4 w = *reinterpret_cast<const f32x4*>(Wc + col);
#ifndef FIN_ABL
#define FIN_ABL 0       // lab (tools/lab/fin_abl.sh): 1 no gWc partial, 2 no pooling, 3 no prototype term, 5 plain store only
#endif
    for (int k = 0; k < K && FIN_ABL != 5; ++k) {
        const f32x4 wn = *reinterpret_cast<const f32x4*>(Wc + (long)min(k + 1, K - 1) * N + col);
        f32x4 gw = zero4();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const float c = cst[(ps * 4 + rl) * ls + k];
            if (FIN_ABL != 3) v[ps] += c * w;
            if (FIN_ABL != 1) gw += c * z[ps];
        }
        if (FIN_ABL != 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) gw[e] = gx_xor32_sum(gx_xor16_sum(gw[e]));      // over the four row lanes
            if (lane < 16) *reinterpret_cast<f32x4*>(stage + k * 64 + c4) = gw;
        }
        w = wn;
    }
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) *reinterpret_cast<f32x4*>(g.C + FIN_PIX(ps) * g.ldc + col) = v[ps];
    if (q.fin_amax) {                                     // the bound the f16x3 gradient GEMMs of the levels below scale G by
        float m = 0.f;
#pragma unroll
        for (int ps = 0; ps < 8; ++ps)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[ps][e]));
        m = gx_wave_max(m);
        // 131 072 waves, one word: an unconditional atomic serialises in L2 (0.98 -> 1.67 ms per launch, measured); the plain
        // read first lets all but the few record-setting waves skip it (a stale read only costs a redundant atomic)
        if (lane == 0 && __float_as_uint(m) > *reinterpret_cast<volatile unsigned*>(q.fin_amax)) atomicMax(q.fin_amax, __float_as_uint(m));
    }
    if (q.fin_pool && FIN_ABL != 2 && FIN_ABL != 5) {
        // pooled pixel of (image row pair, x / 2): the wave's 8 of them, lanes with an even row lane write
        const long rhalf = 1L << (rlog - 1);
        const long b = pw >> (2 * rlog), rem = pw & ((1L << (2 * rlog)) - 1);
        const long pool0 = (b * rhalf + (rem >> (rlog + 1))) * rhalf + ((rem & ((1L << rlog) - 1)) >> 1);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            f32x4 p = v[ps] + v[ps + 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = gx_xor16_sum(p[e]);
            if (!(rl & 1)) *reinterpret_cast<f32x4*>(q.fin_pool + (pool0 + ps * 2 + (rl >> 1)) * N + col) = p;
        }
    }
    if (FIN_ABL == 1 || FIN_ABL == 5) return;
    __syncthreads();                                     // the four waves' partials are in their staging areas
    const int tid = threadIdx.x;
    if (tid < K * 16) {
        const int k = tid >> 4, c = (tid & 15) * 4;
        const float* sp = stage_all + k * 64 + c;
        const f32x4 s = (*reinterpret_cast<const f32x4*>(sp) + *reinterpret_cast<const f32x4*>(sp + EP_STAGE_FLOATS)) +
                        (*reinterpret_cast<const f32x4*>(sp + 2 * EP_STAGE_FLOATS) + *reinterpret_cast<const f32x4*>(sp + 3 * EP_STAGE_FLOATS));
        *reinterpret_cast<f32x4*>(q.fin_part + ((long)tm * K + k) * N + col0 + c) = s;
    }
    __syncthreads();
}

#undef FIN_PIX

// (tm, tn) of logical tile `logical` for the weight-plane kernels: the N tiles go in groups of g.n_group, and within a
// group the pixel tiles are swept with the group's N tiles innermost.  An XCD's contiguous run of logical tiles then
// reads an A tile from HBM once per GROUP (the group's other tiles find it in L2) and keeps only the group's weight
// planes hot.  With tn simply fastest, a 5000-column score GEMM kept all 40 weight slices (15 MB against 4 MB of L2) in
// flight per XCD and fetched 23 GB per launch for 0.65 GB of activations (profiles/r02_k5000_traffic.json history).
__device__ __forceinline__ void xw_tile_of(const GemmArgs& g, int logical, int& tm, int& tn) {
    const int per_group = g.tiles_m * g.n_group;
    const int grp = logical / per_group, rem = logical - grp * per_group;
    const int width = min(g.n_group, g.tiles_n - grp * g.n_group);      // the last group may be narrower
    tm = rem / width;
    tn = grp * g.n_group + (rem - tm * width);
    if (g.live) {
        // the rows past *g.live are the TAIL of the M tiles, and an XCD owns a contiguous run of logical tiles: the dead tail would
        // sit on the last XCDs of a group and the others would carry a full load.  Deal the M tiles to eight interleaved
        // sub-ranges instead (a bijection of [0, tiles_m)): every run then holds its share of the tail.
        const int q = g.tiles_m >> 3, r = g.tiles_m & 7, j = tm & 7, i = tm >> 3;
        tm = j * q + min(j, r) + i;
    }
}

// NT contraction with the weight operand PRE-SPLIT (gx_split_gemm_weights: the projection / prototype weights change
// once per step, their planes are made once per step) — "xw".  What the kernel above strings in series per 16-deep
// k-tile (global loads -> registers -> split -> six LDS plane stores -> barrier) becomes, as in modconv_xp_kernel,
// LDS-DMA for both operands: the B planes land in their LDS image as they are stored, and A lands as RAW fp32
// (128 rows x 16 floats = 8 KB, 16-B units swizzled so the fragment reads are conflict-free).  The activation split
// moves to the fragment read: the four waves are stacked along M (each 32 rows x all 128 columns), so a lane's 8 k
// values of its row are read (2 ds_read_b128) and split by exactly one lane — half the split arithmetic per k-tile of
// the kernel above (its threads split 8 A + 8 B elements), no plane stores, no staging registers, NS-deep prefetch.
// Products and their order equal x6_mfma_tile's, so results are bitwise those of gemm_x6_kernel.
//   LDS unit p (16 B) of the A stage holds row p >> 2, k-quad (p & 3) ^ ((p >> 4) & 3)
// H2: the f16x3 form — B planes are two fp16 planes (8 KB per k-tile), a lane splits its 8 scaled A values into two fp16
// vectors, three products; the accumulators are un-scaled once after the k loop.
// TALL (f16x3 only): 256-row tiles, a wave 64 rows x all 128 columns.  With three products per k-tile the 128-row form is bound
// by LDS traffic, not the matrix cores — every wave reads the whole 8 KB B tile for its 12 MFMAs: 56 KB through LDS per 48
// MFMAs (448 cycles of the 128 B/clk port against 384 MFMA cycles per wave); here 72 KB per 96.
template <bool USUM, bool FIN = false, bool H2 = false, bool TALL = false>   // USUM: the epilogue also leaves the first Sinkhorn pass behind (usum_part); FIN: the
                                         // low-rank path's dense finish in the epilogue, rows in 2 x 64-pixel tiles (xw_epilogue_fin)
__global__ __launch_bounds__(GEMM_THREADS, TALL ? 2 : 3) void gemm_xw_kernel(GemmArgs g) {
    static_assert(!TALL || (H2 && !FIN), "the 256-row tile: f16x3, plain epilogues");
#ifndef XW_TALL_NS
#define XW_TALL_NS 2            // lab (tools/lab/xw_ring.sh): A ring depth of the f16x3 256-row form, with a raw barrier past 2
#endif
    constexpr int NS = TALL ? XW_TALL_NS : 2;       // A ring depth (128-row bf16x6 form: 3 and 4 measured 0 / -8 %: fewer workgroups per CU)
    constexpr int NPL = H2 ? 2 : 3, XW_B = NPL * 4096;               // one B tile (planes)
    constexpr int NRT = TALL ? 2 : 1, TBM = 128 * NRT, XW_A = 8192 * NRT, NU = 2 * NRT;   // 32-row tiles per wave; rows, bytes, DMAs per thread of an A tile
    // A ring of NS tiles, B ring of 2; FIN: + the four waves' coefficient rows [32][ls <= 20], DMA'd in before the k loop;
    // never less than the epilogues' staging (4 waves x EP_STAGE_FLOATS)
    constexpr int LDS_RING = NS * XW_A + 2 * XW_B, LDS_EPI = 4 * EP_STAGE_FLOATS * 4;
    constexpr int CST_OFF = LDS_RING > LDS_EPI ? LDS_RING : LDS_EPI;  // the coefficient rows outlive the ring: past the staging too
    __shared__ __attribute__((aligned(16))) char lds[CST_OFF + (FIN ? 4 * 32 * 20 * 4 : 0)];
    char* const ldsB = lds + NS * XW_A;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = g.tiles_m * g.tiles_n;
    const int logical = xcd_remap((int)blockIdx.x, ntile);
    int tm, tn;
    xw_tile_of(g, logical, tm, tn);
    const int m0 = tm * TBM, n0 = tn * BN;
    if (g.live && m0 >= *g.live) return;                             // rows past the live count: nothing reads the outputs
    const int nkt = g.K >> 4, nkt1 = g.K1 >> 4;                      // K, K1 are multiples of 16
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // A: units p = tid and 256 + tid; tail rows re-read the last valid row (their outputs are dropped)
    unsigned aoff1[NU], aoff2[NU];
    long pix0 = m0;                      // first pixel (row of A / C) of the tile
    if (FIN) {                           // tile tm = (image, row pair, 64-pixel block): local row r = (wave, image row, x)
        const int rlog = g.fin_rlog;
        const int xb = tm & ((1 << (rlog - 6)) - 1), yp = (tm >> (rlog - 6)) & ((1 << (rlog - 1)) - 1);
        pix0 = ((((long)(tm >> (2 * rlog - 7)) << rlog) + 2 * yp) << rlog) + xb * 64;
    }
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int p = i * 256 + tid;
        int row = p >> 2;
        const int q = (p & 3) ^ ((p >> 4) & 3);
        if (FIN) row = (((row & 31) >> 4) << g.fin_rlog) + (row >> 5) * 16 + (row & 15);
        else row = min(row, g.M - 1 - m0);
        aoff1[i] = ((unsigned)row * (unsigned)g.lda + (unsigned)q * 4u) * 4u;
        aoff2[i] = ((unsigned)row * (unsigned)g.lda2 + (unsigned)q * 4u) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rs_a1 = make_rsrc(g.A + pix0 * g.lda, 0x7FFFFFF0u);
    const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(g.A2 ? g.A2 + pix0 * g.lda2 : g.A, 0x7FFFFFF0u);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(reinterpret_cast<const float*>(g.Bp), g.Bp_bytes);
    const unsigned boff = (unsigned)tid * 16u;
    const unsigned bblk0 = (unsigned)tn * (unsigned)nkt * (unsigned)NPL * 4096u;

    auto issue_b = [&](int kt, char* dst) {
        const unsigned blk = bblk0 + (unsigned)kt * (unsigned)NPL * 4096u;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(dst + p * 4096 + wid * 1024), 16, (int)boff,
                                                     (int)(blk + p * 4096u), 0, 0);
    };
    auto issue_a = [&](int kt, char* dst) {
        const bool second = kt >= nkt1;                              // wave-uniform
        const int ks = (second ? kt - nkt1 : kt) * 64;               // byte offset of the k-tile inside a row
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(dst + i * 4096 + wid * 1024), 16, (int)aoff2[i], ks, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_ptr_t)(dst + i * 4096 + wid * 1024), 16, (int)aoff1[i], ks, 0, 0);
        }
    };

    if (FIN) {      // the wave's 32 coefficient rows (2 image rows x 16 pixels = two runs of 16 * ls floats) -> LDS, 4 B per lane
        const int ls = g.fin_ls;
        const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(g.fin_C + (pix0 + wid * 16) * ls, 0x7FFFFFF0u);
        char* const cdst = lds + CST_OFF + wid * (32 * 20 * 4);
        for (int dy = 0; dy < 2; ++dy)
            for (int i = 0; i < (ls >> 2); ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_c, (lds_ptr_t)(cdst + (dy * 16 * ls + i * 64) * 4), 4, lane * 4,
                                                         (int)((((long)dy << g.fin_rlog) * ls + i * 64) * 4), 0, 0);
    }
    // accX[rt]: the wave's rt-th 32-row tile; acc0 = columns 0 .. 63, acc1 = 64 .. 127 (the epilogues take [1][2])
    f32x16 acc0a[NRT][1][2], acc1a[NRT][1][2];
    f32x16 (&acc0)[1][2] = acc0a[0];
    f32x16 (&acc1)[1][2] = acc1a[0];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0a[rt][0][j][r] = 0.f; acc1a[rt][0][j][r] = 0.f; }

    const int r32 = lane & 31, hh = lane >> 5;
    const int arow = wid * (32 * NRT) + r32;                         // the wave's first row tile (TALL: the second is 32 rows down)
    const int af = (arow >> 2) & 3;
    const int au0 = (arow * 4 + ((2 * hh) ^ af)) * 16, au1 = (arow * 4 + ((2 * hh + 1) ^ af)) * 16;
    float a_scale = g.a_scale;
    if constexpr (H2) { if (g.a_amax) a_scale = xh_scale_of(*g.a_amax * g.a_amax_mul); }
    auto compute = [&](const char* st, const char* Bs) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(st + au0);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(st + au1);
        if constexpr (H2) {
            f16x8 b[4][2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int t = 0; t < 4; ++t) b[t][q] = __builtin_bit_cast(f16x8, x6_frag<true, 128>(Bs, q, t * 32 + r32, hh));
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
                // (row + 32: the unit's swizzle term ((p >> 4) & 3 = (row >> 2) & 3) is unchanged, the address moves 32 rows x 64 B)
                const f32x4 y0 = rt ? *reinterpret_cast<const f32x4*>(st + au0 + 2048) : x0;
                const f32x4 y1 = rt ? *reinterpret_cast<const f32x4*>(st + au1 + 2048) : x1;
                unsigned h[4], l[4];
                split2h(y0[0], y0[1], a_scale, h[0], l[0]);
                split2h(y0[2], y0[3], a_scale, h[1], l[1]);
                split2h(y1[0], y1[1], a_scale, h[2], l[2]);
                split2h(y1[2], y1[3], a_scale, h[3], l[3]);
                const f16x8 a[2] = {__builtin_bit_cast(f16x8, (u32x4){h[0], h[1], h[2], h[3]}),
                                    __builtin_bit_cast(f16x8, (u32x4){l[0], l[1], l[2], l[3]})};
                constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};                  // l*h, h*l, h*h
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    acc0a[rt][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[TA[t]], b[0][TB[t]], acc0a[rt][0][0], 0, 0, 0);
                    acc0a[rt][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[TA[t]], b[1][TB[t]], acc0a[rt][0][1], 0, 0, 0);
                    acc1a[rt][0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[TA[t]], b[2][TB[t]], acc1a[rt][0][0], 0, 0, 0);
                    acc1a[rt][0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[TA[t]], b[3][TB[t]], acc1a[rt][0][1], 0, 0, 0);
                }
            }
            return;
        }
        bf16x8 b[4][3];
        constexpr int PB[3] = {0, 2, 1};
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t][PB[q]] = x6_frag<true, 128>(Bs, PB[q], t * 32 + r32, hh);
        unsigned h[4], m[4], l[4];
        split3(x0[0], x0[1], h[0], m[0], l[0]);
        split3(x0[2], x0[3], h[1], m[1], l[1]);
        split3(x1[0], x1[1], h[2], m[2], l[2]);
        split3(x1[2], x1[3], h[3], m[3], l[3]);
        const bf16x8 a[3] = {__builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]}),
                             __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]}),
                             __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]})};
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // l*h, h*l, m*m, m*h, h*m, h*h
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            acc0[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], b[0][TB[t]], acc0[0][0], 0, 0, 0);
            acc0[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], b[1][TB[t]], acc0[0][1], 0, 0, 0);
            acc1[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], b[2][TB[t]], acc1[0][0], 0, 0, 0);
            acc1[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], b[3][TB[t]], acc1[0][1], 0, 0, 0);
        }
    };

    // Rings: A (streamed from HBM once, 64 B per row and k-tile) NS tiles deep, B (L2-resident weights) 2 deep.  Behind
    // the barrier that retires tile t - 1 a wave issues B(t + 1), then A(t + NS - 1), into the slots tile t - 1 used;
    // before tile t it waits for everything up to B(t) — all but the 2 DMAs of the newest A tile.
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) {
        if (t == NS - 2 && nkt > 0) issue_b(0, ldsB);
        if (t < nkt) issue_a(t, lds + t * XW_A);
    }
    int rd = 0, wr = NS - 1;
    for (int t = 0; t < nkt; ++t) {
        // (the newest A tile's NU DMAs per thread may stay in flight; a __syncthreads() would drain them — hipcc puts a
        // vmcnt(0) in front of it —, so a ring deeper than 2 takes the raw barrier)
        if (NS > 2 && t + NS - 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NU) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (NS > 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
        if (t + 1 < nkt) issue_b(t + 1, ldsB + ((t + 1) & 1) * XW_B);
        if (t + NS - 1 < nkt) issue_a(t + NS - 1, lds + wr * XW_A);
        compute(lds + rd * XW_A, ldsB + (t & 1) * XW_B);
        rd = (rd + 1 == NS) ? 0 : rd + 1;
        wr = (wr + 1 == NS) ? 0 : wr + 1;
    }
    __syncthreads();                                                 // every wave has read its last fragments
    if constexpr (H2) {                                              // the scales leave: both are powers of two, the product is exact
        const float un = 1.f / (a_scale * xh_scale_of(*g.b_amax));
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0a[rt][0][j][r] *= un; acc1a[rt][0][j][r] *= un; }
    }
    float* stage = reinterpret_cast<float*>(lds) + wid * EP_STAGE_FLOATS;
    if (FIN) {
        const GemmArgs& q = *gemm_late_args();
        const long pw = pix0 + wid * 16;
        const float* cst = reinterpret_cast<const float*>(lds + CST_OFF) + wid * (32 * 20);
        {       // (both halves' Z rows in flight up front measured 4 % slower: 16 more live registers, spills)
            f32x4 z[8];
            xw_fin_load_z(q, g.N, pw + (lane >> 4), n0 + (lane & 15) * 4, z);
            xw_epilogue_fin(g, acc0, z, tm, pw, n0, lane, stage, cst, reinterpret_cast<const float*>(lds));
        }
        {
            f32x4 z[8];
            xw_fin_load_z(q, g.N, pw + (lane >> 4), n0 + 64 + (lane & 15) * 4, z);
            xw_epilogue_fin(g, acc1, z, tm, pw, n0 + 64, lane, stage, cst, reinterpret_cast<const float*>(lds));
        }
        return;
    }
    if constexpr (!TALL) {
        if (g.beta != 0.f) {
            gemm_epilogue_staged<1, 2>(g, acc0, m0 + wid * 32, n0, lane, stage, 0);
            gemm_epilogue_staged<1, 2>(g, acc1, m0 + wid * 32, n0 + 64, lane, stage, 0);
        } else {
            xw_epilogue<USUM>(g, acc0, m0 + wid * 32, n0, lane, stage, (long)tm * 4 + wid);
            xw_epilogue<USUM>(g, acc1, m0 + wid * 32, n0 + 64, lane, stage, (long)tm * 4 + wid);
        }
    } else {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int row0 = m0 + wid * (32 * NRT) + rt * 32;
            // (256-row tiles: a row tile wholly past the last 128-row tile has no rows and no slot in the per-slice tables)
            if (row0 >= ((g.M + 127) & ~127)) break;
            if (g.beta != 0.f) {
                gemm_epilogue_staged<1, 2>(g, acc0a[rt], row0, n0, lane, stage, 0);
                gemm_epilogue_staged<1, 2>(g, acc1a[rt], row0, n0 + 64, lane, stage, 0);
            } else {
                xw_epilogue<USUM>(g, acc0a[rt], row0, n0, lane, stage, (long)(row0 >> 5));
                xw_epilogue<USUM>(g, acc1a[rt], row0, n0 + 64, lane, stage, (long)(row0 >> 5));
            }
        }
    }
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
constexpr int WS_SLOT = XImg<WS_BM>::BYTES + XImg<WS_BN>::BYTES;          // 27648 + 15360 B

// H2: the f16x3 form — the producers scale each operand by its power of two (g.a_scale / g.b_scale, from the caller's bounds)
// and write two fp16 planes, the consumers issue three products per tile pair and un-scale their accumulators once.
template <bool TA, bool TB, bool VEC, bool H2 = false>
__global__ __launch_bounds__(WS_THREADS, 1) void gemm_x6ws_kernel(GemmArgs g) {
    constexpr bool A_KC = !TA, B_KC = TB;
    __shared__ __attribute__((aligned(16))) char lds[2 * WS_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntile = g.tiles_m * g.tiles_n;
    int logical, split;
    gemm_tile_of_block(ntile, logical, split);
    const int tm = logical / g.tiles_n, tn = logical - tm * g.tiles_n;
    const int m0 = tm * WS_BM, n0 = tn * WS_BN;
    const bool bsec = n0 >= g.N1;                                    // [B1 | B2]: this tile's columns are B2's
    const float* Bsrc = bsec ? g.B2 : g.B;
    const long ldbs = bsec ? g.ldb2 : g.ldb;
    const int nb0 = bsec ? n0 - g.N1 : n0, nbN = bsec ? g.N - g.N1 : g.N1;
    int kbeg, kend;
    gemm_k_range(g, split, kbeg, kend);
    const int nkt = (kend - kbeg + XK - 1) / XK;
    char* slot0 = lds;
    char* slot1 = lds + WS_SLOT;
    float a_scale = 1.f;
    float b_scale = 1.f;
    if constexpr (H2) {
        a_scale = g.a_amax ? xh_scale_of(*g.a_amax * g.a_amax_mul) : g.a_scale;
        b_scale = g.b_amax_ws ? xh_scale_of(*g.b_amax_ws * g.b_amax_mul) : g.b_scale;
    }

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
            x6_load_tile<B_KC, VEC, WS_BN>(Bsrc, ldbs, nb0, nbN, k0, kend, rb, p);
        };
        auto stage = [&](char* slot, const f32x4 (&ra)[4], const f32x4 (&rb)[2]) {
            if constexpr (H2) {
                xh_store_tile<A_KC, WS_BM>(slot, ra, p, a_scale);
                xh_store_tile<B_KC, WS_BN>(slot + XImg<WS_BM>::BYTES, rb, p, b_scale);
            } else {
                x6_store_tile<A_KC, WS_BM>(slot, ra, p);
                x6_store_tile<B_KC, WS_BN>(slot + XImg<WS_BM>::BYTES, rb, p);
            }
        };
#ifndef WS_DEPTH
#define WS_DEPTH 2      // lab (tools/lab/ws_depth.sh): register sets of k-tiles in flight per producer thread
#endif
        if constexpr (WS_DEPTH == 4) {
            // LAB ONLY.  Four k-tiles in flight per producer thread (four register sets, 96 VGPRs): a tile's loads issued four
            // barrier periods before it is staged, against one with two sets.  Measured same-box, alternating (round 5,
            // profiles/r05_lab_ws_depth.txt): SLOWER — the step's TN / NT GEMMs 5.26 -> 5.46 ms at K = 8, 95.2 -> 97.4 ms at
            // K = 5000: the producers are not waiting for their loads (45 % MFMA busy is the power budget shared with the
            // operand streams, DESIGN par. 4.21), and 72 more live registers cost the consumers' schedule.
            f32x4 ra2[4], rb2[2], ra3[4], rb3[2];
            load_tiles(0, ra0, rb0);
            load_tiles(1, ra1, rb1);
            load_tiles(2, ra2, rb2);
            load_tiles(3, ra3, rb3);
            if (nkt > 0) stage(slot0, ra0, rb0);
            load_tiles(4, ra0, rb0);
            __syncthreads();                                         // barrier 0: tile 0 visible
            for (int t = 0; t < nkt; t += 4) {
                if (t + 1 < nkt) stage(slot1, ra1, rb1);
                load_tiles(t + 5, ra1, rb1);
                __syncthreads();                                     // barrier t+1
                if (t + 1 >= nkt) break;
                if (t + 2 < nkt) stage(slot0, ra2, rb2);
                load_tiles(t + 6, ra2, rb2);
                __syncthreads();                                     // barrier t+2
                if (t + 2 >= nkt) break;
                if (t + 3 < nkt) stage(slot1, ra3, rb3);
                load_tiles(t + 7, ra3, rb3);
                __syncthreads();                                     // barrier t+3
                if (t + 3 >= nkt) break;
                if (t + 4 < nkt) stage(slot0, ra0, rb0);
                load_tiles(t + 8, ra0, rb0);
                __syncthreads();                                     // barrier t+4
            }
            return;
        }
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
        if constexpr (H2) xh_mfma_tile<4, 2, WS_BM, WS_BN, A_KC, B_KC>(slot, slot + XImg<WS_BM>::BYTES, acc, wr * 128, wc * 64, lane);
        else x6_mfma_tile<A_KC, B_KC, 4, 2, WS_BM, WS_BN>(slot, slot + XImg<WS_BM>::BYTES, acc, wr * 128, wc * 64, lane);
        __syncthreads();                                             // barrier t+1
    }
    if constexpr (H2) {
        const float un = 1.f / (a_scale * b_scale);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= un;
    }
    if (g.ep_vec)
        gemm_epilogue_staged<4, 2>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, reinterpret_cast<float*>(lds) + wid * EP_STAGE_FLOATS, split);
    else
        gemm_epilogue_t<4, 2>(g, acc, m0 + wr * 128, n0 + wc * 64, lane, split);
}

// Sum split-K slabs in a fixed order (bitwise reproducible) and apply the epilogue.
__global__ void gemm_splitk_reduce_kernel(const float* __restrict__ slab, int splits, float* __restrict__ C,
                                          int M, int N, long ldc, float alpha, float beta,
                                          const float* __restrict__ row_scale,
                                          const float* __restrict__ col_bias) {
#pragma clang fp contract(off)
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

// max |x| (or max sqrt |x|) of a strided matrix as float bits (non-negative floats order like unsigned integers)
__global__ void absmax_strided_kernel(const float* __restrict__ x, long n, int cols, long ld, int sqrt_of,
                                      unsigned* __restrict__ out) {
    __shared__ float red[16];
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long r = cols == 1 ? i : i / cols;
        const float v = fabsf(x[r * ld + (i - r * cols)]);
        m = fmaxf(m, sqrt_of ? sqrtf(v) : v);
    }
    m = gx_block_max(m, red);
    if (threadIdx.x == 0 && __float_as_uint(m) > *reinterpret_cast<volatile unsigned*>(out)) atomicMax(out, __float_as_uint(m));
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

constexpr int WS_MIN_K = 768;      // shortest k extent (per split) that takes the wave-specialised kernel (256: 7.4 -> 9.0 ms)

static int launch_gemm(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B,
                       long ldb, float* C, long ldc, float alpha, float beta, const float* row_scale,
                       const float* col_bias, int split_k, float* workspace, size_t workspace_bytes,
                       hipStream_t stream, const float* A2 = nullptr, long lda2 = 0, int K1 = -1,
                       const float* up_src = nullptr, int up_rf = 0, const float* B2 = nullptr, long ldb2 = 0,
                       int N1 = -1, const int* k_live = nullptr, float a_range = 0.f, float b_range = 0.f,
                       const float* a_amax_dev = nullptr, const float* b_amax_dev = nullptr) {
    if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N) return GX_ERR_BAD_ARG;
    GemmArgs g;
    g.live = k_live;
    g.a_scale = g.b_scale = 1.f; g.b_amax = nullptr; g.a_amax = nullptr; g.a_amax_mul = 1.f; g.fin_amax = nullptr;
    g.b_amax_ws = nullptr; g.b_amax_mul = 1.f;
    g.Bp = nullptr; g.Bp_bytes = 0; g.colmax_part = nullptr; g.usum_part = nullptr; g.sk_inv_eps = 0.f; g.n_group = 1;
    g.up_src = up_src; g.up_shift = 0;
    if (up_src) {
        if (split_k > 1 || up_rf < 2 || (up_rf & (up_rf - 1)) || (long)M % ((long)up_rf * up_rf)) return GX_ERR_BAD_ARG;
        while ((1 << g.up_shift) < up_rf) ++g.up_shift;
    }
    g.A2 = A2; g.lda2 = lda2; g.K1 = (A2 ? K1 : K);
    g.B2 = B2; g.ldb2 = B2 ? ldb2 : ldb; g.N1 = (B2 ? N1 : N);
    if (B2 && (transB || N1 <= 0 || N1 >= N || (N1 % BN) != 0 || ldb2 < N - N1)) return GX_ERR_BAD_ARG;
    if (A2 && (transA || K1 <= 0 || K1 >= K || (K1 % BK) != 0 || lda2 <= 0)) return GX_ERR_BAD_ARG;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta; g.row_scale = row_scale; g.col_bias = col_bias;
    g.vec_ok = (((uintptr_t)A | (uintptr_t)B | (uintptr_t)A2 | (uintptr_t)B2) & 15) == 0 && (lda % 4 == 0) && (ldb % 4 == 0) &&
               (lda2 % 4 == 0) && (g.ldb2 % 4 == 0) && g.ldb2 < (1 << 22) && lda < (1 << 22) && ldb < (1 << 22) && lda2 < (1 << 22) && (K % 4 == 0) && (g.K1 % 4 == 0) && (!transA || M % 4 == 0) &&
               (transB || N % 4 == 0);
    const bool x6 = gemm_mode() == 1;
    // wave-specialised 256 x 128 tiles (one workgroup per CU) pay off once the k loop is long enough
    // to amortise a tile's prologue / epilogue, which nothing overlaps there
    const bool ws = x6 && M > BM && (split_k > 1 ? K / split_k : K) >= WS_MIN_K;
    // f16x3 form: both operands' bounds given, on the wave-specialised kernel (the 4-wave kernels keep bf16x6: short k loops)
    const bool h2 = ws && a_range > 0.f && b_range > 0.f;
    if (h2) {
        g.a_scale = xh_scale_of(a_range); g.b_scale = xh_scale_of(b_range);
        if (a_amax_dev) { g.a_amax = a_amax_dev; g.a_amax_mul = a_range; }     // the bound is *a_amax_dev x a_range, read on the device
        if (b_amax_dev) { g.b_amax_ws = b_amax_dev; g.b_amax_mul = b_range; }
    }
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
    {   // float4 epilogue: every row segment it touches must be 16-B aligned
        const bool direct = (((uintptr_t)C & 15) == 0) && (ldc % 4 == 0);
        const bool slab_ok = (((uintptr_t)workspace & 15) == 0);
        const bool up_ok = !up_src || (((uintptr_t)up_src & 15) == 0);
        g.ep_vec = (N % 4 == 0) && (g.slab ? slab_ok : (direct && up_ok));
    }
    dim3 grid(g.tiles_m * g.tiles_n, split_k), block(GEMM_THREADS);
#define GX_LAUNCH_GEMM(TA_, TB_)                                                                             \
    do {                                                                                                       \
        if (h2 && g.vec_ok) hipLaunchKernelGGL((gemm_x6ws_kernel<TA_, TB_, true, true>), grid, dim3(WS_THREADS), 0, stream, g); \
        else if (h2) hipLaunchKernelGGL((gemm_x6ws_kernel<TA_, TB_, false, true>), grid, dim3(WS_THREADS), 0, stream, g); \
        else if (ws && g.vec_ok) hipLaunchKernelGGL((gemm_x6ws_kernel<TA_, TB_, true>), grid, dim3(WS_THREADS), 0, stream, g); \
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

// C = alpha * row_scale .* ([A1 | A2] B^T) + col_bias + beta * C (+ upsample2(coarse)) with B given as pre-split planes
// (gx_split_gemm_weights of the [N][K1 + K2] weight).  bf16x6 engine only; K1, K2 multiples of 16, rows 16-B aligned.
static int g_xw_tall = 1;                 // lab / tests: gx_set_gemm_tall(0) keeps the 128-row tile of the f16x3 weight-plane GEMM
static int g_xw_ngroup = 0;               // lab: gx_set_gemm_ngroup(n > 0) overrides the N tiles per group of the tile order
static int launch_gemm_xw(int M, int N, int K1, const float* A1, long lda1, int K2, const float* A2, long lda2,
                          const unsigned short* Bp, float* C, long ldc, float alpha, float beta, const float* row_scale,
                          const float* col_bias, const float* up_src, int up_rf, float* colmax_part, hipStream_t stream,
                          float* usum_part = nullptr, float sk_eps = 0.f, const GemmArgs* fin = nullptr,
                          const int* m_live = nullptr, float a_range = 0.f, const float* a_amax_dev = nullptr) {
    if (!A1 || !Bp || !C || M <= 0 || N <= 0 || K1 <= 0 || K2 < 0 || (K2 > 0 && !A2) || ldc < N) return GX_ERR_BAD_ARG;
    if (gemm_mode() != 1) return GX_ERR_BAD_ARG;
    const int K = K1 + K2;
    if ((K1 & 15) || (K2 & 15) || (lda1 & 3) || (K2 > 0 && (lda2 & 3)) || lda1 < K1 || (K2 > 0 && lda2 < K2) ||
        lda1 >= (1 << 22) || lda2 >= (1 << 22) || (N & 3) || (ldc & 3) ||
        127L * lda1 * 4 + 4L * K1 >= 0x7FFFFFF0L || 127L * lda2 * 4 + 4L * K2 >= 0x7FFFFFF0L ||      // 32-bit buffer offsets of a tile
        (((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)Bp | (uintptr_t)C | (uintptr_t)up_src) & 15))
        return GX_ERR_BAD_ARG;
    GemmArgs g;
    g.live = m_live;
    if (m_live && (colmax_part || fin)) return GX_ERR_BAD_ARG;        // per-slice outputs / the finish need every tile
    g.up_src = up_src; g.up_shift = 0;
    if (up_src) {
        if (up_rf < 2 || (up_rf & (up_rf - 1)) || (long)M % ((long)up_rf * up_rf)) return GX_ERR_BAD_ARG;
        while ((1 << g.up_shift) < up_rf) ++g.up_shift;
    }
    g.A = A1; g.lda = lda1; g.A2 = K2 > 0 ? A2 : nullptr; g.lda2 = K2 > 0 ? lda2 : lda1; g.K1 = K1;
    g.B = nullptr; g.ldb = 0; g.B2 = nullptr; g.ldb2 = 0; g.N1 = N; g.C = C; g.M = M; g.N = N; g.K = K; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta; g.row_scale = row_scale; g.col_bias = col_bias;
    g.k_per_split = K; g.slab = nullptr; g.vec_ok = 1; g.ep_vec = 1;
    const bool h2 = a_range > 0.f;            // the planes are two fp16 planes + the amax trailer (gx_split_gemm_weights_h2)
    g.tiles_n = gx_cdiv(N, BN);
    // f16x3: 256-row tiles where they still give every CU two workgroups (the 128-row form is LDS-bound, see the kernel)
    const bool tall = h2 && !fin && g_xw_tall && (long)gx_cdiv(M, 2 * BM) * g.tiles_n >= 512;
    g.tiles_m = gx_cdiv(M, tall ? 2 * BM : BM);
    const size_t bpb = (size_t)g.tiles_n * (K >> 4) * (h2 ? 2 : 3) * 4096;
    if (bpb >= 0xFFFFFFF0ull || (long)g.tiles_m * g.tiles_n >= (1L << 31)) return GX_ERR_BAD_ARG;
    g.Bp = Bp; g.Bp_bytes = (unsigned)bpb;
    g.a_scale = h2 ? xh_scale_of(a_range) : 1.f;
    g.b_scale = 1.f; g.a_amax = nullptr; g.a_amax_mul = 1.f; g.fin_amax = nullptr;
    g.b_amax_ws = nullptr; g.b_amax_mul = 1.f;
    if (h2 && a_amax_dev) { g.a_amax = a_amax_dev; g.a_amax_mul = a_range; }   // the bound is *a_amax_dev x a_range, read on the device
    g.b_amax = reinterpret_cast<const float*>(reinterpret_cast<const char*>(Bp) + (size_t)g.tiles_n * (K >> 4) * 3 * 4096);
    {   // N tiles per group: their weight planes (K x 128 x 6 B each) should sit in an XCD's 4 MB L2 next to the A stream
        const long per_tile = (long)K * 128 * 6;
        long gn = (2L << 20) / per_tile;
        // ... but never a group of one: with N tiles of more than 2 MB of planes each (the dz GEMM at K = 5000: 3.85 MB) every N
        // tile streamed the whole A matrix from HBM by itself — 26 GB per launch for 6.4 GB of dS (profiles/r03_traffic_k5000.json);
        // pairs share it through L2 and the planes come from the Infinity Cache: 7.72 -> 7.46 ms, same-process A/B (dz_lab.py)
        gn = gn < 2 ? 2 : gn;
        if (g_xw_ngroup > 0) gn = g_xw_ngroup;
        gn = gn > g.tiles_n ? g.tiles_n : gn;
        g.n_group = (int)gn;
    }
    if (colmax_part && (beta != 0.f || ((uintptr_t)colmax_part & 15))) return GX_ERR_BAD_ARG;
    if (usum_part && (!colmax_part || !(sk_eps > 0.f) || ((uintptr_t)usum_part & 15))) return GX_ERR_BAD_ARG;
    g.colmax_part = colmax_part;
    g.usum_part = usum_part; g.sk_inv_eps = usum_part ? 1.f / sk_eps : 0.f;
    g.fin_C = g.fin_Z = g.fin_Wc = nullptr; g.fin_pool = g.fin_part = nullptr; g.fin_K = g.fin_ls = g.fin_rlog = 0;
    const dim3 grid(g.tiles_m * g.tiles_n), block(GEMM_THREADS);
    if (fin) {
        g.fin_C = fin->fin_C; g.fin_Z = fin->fin_Z; g.fin_Wc = fin->fin_Wc; g.fin_pool = fin->fin_pool; g.fin_part = fin->fin_part;
        g.fin_K = fin->fin_K; g.fin_ls = fin->fin_ls; g.fin_rlog = fin->fin_rlog; g.fin_amax = fin->fin_amax;
        if (h2) hipLaunchKernelGGL((gemm_xw_kernel<false, true, true>), grid, block, 0, stream, g);
        else hipLaunchKernelGGL((gemm_xw_kernel<false, true>), grid, block, 0, stream, g);
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    if (tall && usum_part) hipLaunchKernelGGL((gemm_xw_kernel<true, false, true, true>), grid, block, 0, stream, g);
    else if (tall) hipLaunchKernelGGL((gemm_xw_kernel<false, false, true, true>), grid, block, 0, stream, g);
    else if (h2 && usum_part) hipLaunchKernelGGL((gemm_xw_kernel<true, false, true>), grid, block, 0, stream, g);
    else if (h2) hipLaunchKernelGGL((gemm_xw_kernel<false, false, true>), grid, block, 0, stream, g);
    else if (usum_part) hipLaunchKernelGGL(gemm_xw_kernel<true>, grid, block, 0, stream, g);
    else hipLaunchKernelGGL(gemm_xw_kernel<false>, grid, block, 0, stream, g);
    GX_CHECK_LAUNCH();
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
int gx_set_gemm_tall(int on) { const int prev = gx::g_xw_tall; gx::g_xw_tall = on ? 1 : 0; return prev; }
int gx_set_gemm_ngroup(int n) { const int prev = gx::g_xw_ngroup; gx::g_xw_ngroup = n > 0 ? n : 0; return prev; }

size_t gx_gemm_f32_workspace_bytes(int M, int N, int split_k) {
    return split_k > 1 ? (size_t)split_k * M * N * sizeof(float) : 0;
}

int gx_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                float* C, long ldc, float alpha, float beta, const float* row_scale, const float* col_bias,
                int split_k, float* workspace, size_t workspace_bytes, void* stream) {
    return gx::launch_gemm(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, row_scale, col_bias,
                           split_k, workspace, workspace_bytes, (hipStream_t)stream);
}

// gx_gemm_f32 for a TN contraction whose k rows carry data only below *k_live (a device int; rows at or past it are not
// read): transA must be 1.  The launch covers K; the count never visits the host.
int gx_gemm_f32_klive(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                      float* C, long ldc, float alpha, float beta, const float* row_scale, const float* col_bias,
                      int split_k, float* workspace, size_t workspace_bytes, const int* k_live, void* stream) {
    if (!transA || !k_live) return GX_ERR_BAD_ARG;
    return gx::launch_gemm(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, row_scale, col_bias,
                           split_k, workspace, workspace_bytes, (hipStream_t)stream, nullptr, 0, -1, nullptr, 0, nullptr, 0, -1,
                           k_live);
}

// gx_gemm_f32 / gx_gemm_f32_klive (k_live may be NULL) with the caller's bounds on |A| and |B|: where the wave-specialised
// kernel takes the launch (long k loops: the gradient GEMMs over pixels) it runs the f16x3 form — operands scaled by the
// powers of two that bring the bounds to [2^13, 2^14), two fp16 planes, three products (csrc/gx_mfma_x6.h); elsewhere the
// bounds are ignored (bf16x6).  An element above 4 x its bound comes out NaN; every element carries an absolute error of
// bound x 2^-38, so a bound should sit within ~2^14 of its operand's RMS.  a_amax_dev / b_amax_dev (optional device floats: the
// operand's maximum magnitude as its producer left it): the bound is then *x_amax_dev x x_range, read on the device.
int gx_gemm_f32_ranged(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                       float* C, long ldc, float alpha, float beta, const float* row_scale, const float* col_bias,
                       int split_k, float* workspace, size_t workspace_bytes, const int* k_live, float a_range, float b_range,
                       const float* a_amax_dev, const float* b_amax_dev, void* stream) {
    if ((k_live && !transA) || !(a_range > 0.f) || !(b_range > 0.f)) return GX_ERR_BAD_ARG;
    return gx::launch_gemm(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, alpha, beta, row_scale, col_bias,
                           split_k, workspace, workspace_bytes, (hipStream_t)stream, nullptr, 0, -1, nullptr, 0, nullptr, 0, -1,
                           k_live, a_range, b_range, a_amax_dev, b_amax_dev);
}
// gx_gemm_f32_tn_cat with the bounds of gx_gemm_f32_ranged (A's on the device: *a_amax_dev x a_mul; |B1|, |B2| <= b_range)
int gx_gemm_f32_tn_cat_ranged(int M, int N1, int N2, int K, const float* A, long lda, const float* B1, long ldb1, const float* B2,
                              long ldb2, float* C, long ldc, float alpha, float beta, int split_k, float* workspace,
                              size_t workspace_bytes, const float* a_amax_dev, float a_mul, float b_range, void* stream) {
    if (!B2 || N2 <= 0 || !a_amax_dev || !(a_mul > 0.f) || !(b_range > 0.f)) return GX_ERR_BAD_ARG;
    return gx::launch_gemm(1, 0, M, N1 + N2, K, A, lda, B1, ldb1, C, ldc, alpha, beta, nullptr, nullptr, split_k, workspace,
                           workspace_bytes, (hipStream_t)stream, nullptr, 0, -1, nullptr, 0, B2, ldb2, N1, nullptr, a_mul, b_range,
                           a_amax_dev);
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

// C = [A1 | A2] opB + upsample2(coarse): C's M rows are the pixels of (fine_res x fine_res) images, coarse
// [M / 4][N] the finished table of the level below (K2 = 0: a single A).  One level of the pyramid projection.
int gx_gemm_f32_upadd(int transB, int M, int N, int K1, const float* A1, long lda1, int K2, const float* A2, long lda2,
                      const float* B, long ldb, float* C, long ldc, float beta, const float* coarse, int fine_res,
                      void* stream) {
    if (!coarse || K2 < 0 || (K2 > 0 && !A2)) return GX_ERR_BAD_ARG;
    return gx::launch_gemm(0, transB, M, N, K1 + K2, A1, lda1, B, ldb, C, ldc, 1.f, beta, nullptr, nullptr, 1, nullptr, 0,
                           (hipStream_t)stream, K2 > 0 ? A2 : nullptr, K2 > 0 ? lda2 : 0, K2 > 0 ? K1 : -1, coarse, fine_res);
}

// The NT contraction with pre-split weight planes (gemm_xw_kernel): C = alpha * row_scale .* ([A1 | A2] B^T) + col_bias
// + beta * C + upsample2(coarse).  b_planes = gx_split_gemm_weights(B [N][K1 + K2]); K2 = 0 / coarse = NULL / row_scale
// = NULL / col_bias = NULL switch the optional parts off.  colmax_part (optional, beta == 0): gx_gemm_colmax_part_bytes(M, N) bytes = [4 ceil(M / 128)][N]
// partial column maxima of C (rows past M: -inf) — max over its rows = the Sinkhorn column maxima, without a pass over C.  Results are bitwise those of gx_gemm_f32 / _cat / _upadd on
// the bf16x6 engine.
// (+ 16 B: the trailer of the f16x3 form — max |B| as a float, written by gx_split_gemm_weights_h2)
size_t gx_gemm_weight_planes_bytes(int N, int K) { return (size_t)gx_cdiv(N, 128) * gx_cdiv(K, 16) * 3 * 4096 + 16; }
// a_range: 0 = b_planes are the three bf16 planes of gx_split_gemm_weights (bf16x6); > 0 = an upper bound of |A| (rounded up
// to a power of two inside), b_planes the two fp16 planes of gx_split_gemm_weights_h2 (f16x3: half the matrix-core work).
// An A element above a_range x 4 overflows fp16 and comes out NaN; every element carries an absolute error of a_range x 2^-38
// (the fp16 subnormal step): the bound should sit within ~2^14 of A's RMS.  a_amax_dev (optional, with a_range > 0): a device
// float holding max |A| (or a bound of it) as A's producer left it — the bound is then *a_amax_dev x a_range, read by the kernel.
int gx_gemm_f32_nt_planes(int M, int N, int K1, const float* A1, long lda1, int K2, const float* A2, long lda2,
                          const unsigned short* b_planes, float* C, long ldc, float alpha, float beta,
                          const float* row_scale, const float* col_bias, const float* coarse, int fine_res,
                          float* colmax_part, float* usum_part, float sk_eps, float a_range, const float* a_amax_dev,
                          void* stream) {
    if (a_range < 0.f || (a_amax_dev && !(a_range > 0.f))) return GX_ERR_BAD_ARG;
    return gx::launch_gemm_xw(M, N, K1, A1, lda1, K2, A2, lda2, b_planes, C, ldc, alpha, beta, row_scale, col_bias, coarse,
                              fine_res, colmax_part, (hipStream_t)stream, usum_part, sk_eps, nullptr, nullptr, a_range, a_amax_dev);
}
// The gradient of the level below the finest one of the low-rank prototype path (K <= 16, pyramid.backward_lowrank) in ONE
// kernel: G = A B^T (A = Tt (M, K1) from gx_fineq_bwd, B = Wf as planes) and, in the epilogue, the dense finish
//   G[q] += coeffs[q][:K] Wc + coeffs[q][K] Z[q];   gWc = beta gWc + coeffs[:, :K]^T Z;   Gpool = 2x2 sums of G (may be NULL)
// — what gx_gemm_f32_nt_planes + gx_proto_lowrank_finish(accumulate) do in two passes over G.  The M rows are the pixels of
// res x res images (res a power of two >= 64), coeffs (M, ls = gx_lowrank_stride(K)), Z and G (M, N) dense, N % 128 == 0.
// a_range / a_amax_dev: as in gx_gemm_f32_nt_planes (b_planes then from gx_split_gemm_weights_h2; gx_fineq_bwd leaves max |Tt|).
int gx_gemm_finish_ok(int N, int K1, int res, int K) {
    return gx::gemm_mode() == 1 && N > 0 && (N & 127) == 0 && K1 > 0 && (K1 & 15) == 0 && res >= 64 && !(res & (res - 1)) &&
           K >= 1 && K <= 16;
}
size_t gx_gemm_finish_workspace_bytes(long M, int N, int K) { return ((size_t)(M / 128) + 8) * K * N * sizeof(float); }
int gx_gemm_f32_nt_planes_finish(long M, int N, int K1, const float* A1, long lda1, const unsigned short* b_planes, float* G,
                                 int res, const float* coeffs, const float* Z, const float* Wc, int K, float* Gpool,
                                 float* gWc, float beta, float* workspace, size_t workspace_bytes, float* g_amax, float a_range,
                                 const float* a_amax_dev, void* stream) {
    if (a_range < 0.f || (a_amax_dev && !(a_range > 0.f))) return GX_ERR_BAD_ARG;
    if (!gx_gemm_finish_ok(N, K1, res, K) || !coeffs || !Z || !Wc || !gWc || M <= 0 || M % ((long)res * res) ||
        M >= (1L << 31) || (((uintptr_t)coeffs | (uintptr_t)Z | (uintptr_t)Wc | (uintptr_t)Gpool) & 15))
        return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_gemm_finish_workspace_bytes(M, N, K)) return GX_ERR_WORKSPACE;
    gx::GemmArgs fin{};
    fin.fin_C = coeffs; fin.fin_Z = Z; fin.fin_Wc = Wc; fin.fin_pool = Gpool; fin.fin_part = workspace;
    fin.fin_K = K; fin.fin_ls = (K + 1 + 3) & ~3; fin.fin_rlog = 0;
    fin.fin_amax = reinterpret_cast<unsigned*>(g_amax);            // optional: max |G| (the caller zeroes it)
    while ((1 << fin.fin_rlog) < res) ++fin.fin_rlog;
    const int rc = gx::launch_gemm_xw((int)M, N, K1, A1, lda1, 0, nullptr, 0, b_planes, G, N, 1.f, 0.f, nullptr, nullptr, nullptr,
                                      0, nullptr, (hipStream_t)stream, nullptr, 0.f, &fin, nullptr, a_range, a_amax_dev);
    if (rc) return rc;
    const int tiles_m = (int)(M / 128);
    gx::launch_partial_reduce(workspace, tiles_m, K * N, 1.f, beta, gWc, (hipStream_t)stream, workspace + (size_t)tiles_m * K * N);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// gx_gemm_f32_nt_planes (one A source, no coarse table, no column maxima) for an A whose rows carry data only below *m_live
// (a device int): 128-row tiles at or past it are skipped and their rows of C left untouched.
int gx_gemm_f32_nt_planes_mlive(int M, int N, int K1, const float* A1, long lda1, const unsigned short* b_planes, float* C,
                                long ldc, float alpha, float beta, const float* col_bias, const int* m_live, float a_range,
                                void* stream) {
    if (!m_live || a_range < 0.f) return GX_ERR_BAD_ARG;
    return gx::launch_gemm_xw(M, N, K1, A1, lda1, 0, nullptr, 0, b_planes, C, ldc, alpha, beta, nullptr, col_bias, nullptr, 0,
                              nullptr, (hipStream_t)stream, nullptr, 0.f, nullptr, m_live, a_range);
}
// *out = max |x| over an [rows][cols] matrix of row stride ld (sqrt_of: max sqrt(|x|) — e.g. a column of squared row norms
// turns into a bound of the rows' elements); zero for an all-zero matrix.  The device-side bound gx_gemm_f32_nt_planes /
// gx_gemm_f32_ranged take as a_amax_dev / b_amax_dev when no producer left one behind.
int gx_absmax(const float* x, long rows, int cols, long ld, int sqrt_of, float* out, void* stream) {
    if (!x || !out || rows <= 0 || cols <= 0 || ld < cols) return GX_ERR_BAD_ARG;
    if (hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return GX_ERR_LAUNCH;
    const long n = rows * cols;
    long want = (n + 1023) / 1024;
    hipLaunchKernelGGL(gx::absmax_strided_kernel, dim3((int)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, x, n,
                       cols, ld, sqrt_of, reinterpret_cast<unsigned*>(out));
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// bytes of the colmax_part table of gx_gemm_f32_nt_planes: N floats per wave (32 rows of C) of every 128-row tile
size_t gx_gemm_colmax_part_bytes(int M, int N) { return (size_t)4 * gx_cdiv(M, 128) * N * sizeof(float); }

// C = alpha * A^T [B1 | B2] + beta * C: A is [K][M], B1 [K][N1], B2 [K][N2] (N1 a multiple of 128).  The weight gradient
// of one pyramid level against both of its taps in ONE pass over the (large) A = dZ table instead of two.
int gx_gemm_f32_tn_cat(int M, int N1, int N2, int K, const float* A, long lda, const float* B1, long ldb1, const float* B2,
                       long ldb2, float* C, long ldc, float alpha, float beta, int split_k, float* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!B2 || N2 <= 0) return GX_ERR_BAD_ARG;
    return gx::launch_gemm(1, 0, M, N1 + N2, K, A, lda, B1, ldb1, C, ldc, alpha, beta, nullptr, nullptr, split_k, workspace,
                           workspace_bytes, (hipStream_t)stream, nullptr, 0, -1, nullptr, 0, B2, ldb2, N1);
}

}  // extern "C"
