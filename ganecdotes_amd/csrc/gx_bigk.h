// Row kernels for MANY prototypes (64 < K <= 6144, e.g. the reference's K = 4000 / 5000; the column passes up to 8192): one
// wavefront owns a score row, lane l holds columns l + 64 j in registers (NJ = ceil(K/64) rounded
// up), so a Sinkhorn sweep reads the row ONCE from HBM, evaluates ONE exp per score, reduces
// with wavefront shuffles and accumulates the column sums in per-lane registers (no LDS atomics
// in the row loop).  The generic team kernels of gx_swav.hip re-read each row three times and
// pay an LDS atomic per score; at K = 5000, 320 000 rows they ran at 0.4 TB/s.
#pragma once
#include "gx_common.h"
#include "gx_bigk_io.h"

namespace gx {

__device__ __forceinline__ unsigned bk_f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bk_ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

constexpr float BK_LOG2E = 1.44269504088896340736f, BK_LN2 = 0.693147180559945309417f;
enum { BK_COLMAX = 0, BK_COLSUM = 1, BK_FIRST = 2, BK_SWEEP = 3, BK_FIRSTLSE = 4 };

// exp(x) for x <= ~0 (every call site subtracts a maximum first) as v_exp_f32(x * log2 e) with the product carried
// in two floats: y = fl(x L) and its error fma(x, L, -y) + x (log2 e - L), folded back as exp2(y) (1 + r ln 2).
// Five VALU instructions instead of the ~12 of expf (range reduction + ldexp + overflow selects), same accuracy
// class (<= 2 ulp); arguments below -126 / ln 2 return 0 like expf's underflow.
__device__ __forceinline__ float bk_exp(float x) {
    constexpr float L = 1.44269502162933349609375f, L_LO = 1.925963033500011e-08f, LN2 = 0.693147182464599609375f;
    x = fmaxf(x, -200.f);                                            // exp(-inf) = exp(-200) = 0 (no inf - inf below)
    const float y = x * L;
    const float r = fmaf(x, L, -y) + x * L_LO;
    const float e = __builtin_amdgcn_exp2f(y);
    return fmaf(e * LN2, r, e);
}

// ... and with the product x log2 e rounded ONCE: a relative error of |x| 2^-24 on a term e^x <= 1, i.e. at most 0.37 x 2^-24 of
// the largest term of the sum it enters — the error the reference's own `scores / eps` and `scores / T` carry into their exp
// (swav_clustering.py:519, :553).  Two VALU instructions; what the cross-entropy kernels — VALU-bound next to their row
// traffic — evaluate four times per score.
#ifndef CE_PIX_ABL
#define CE_PIX_ABL 0        // lab (tools/lab/ce_tab_abl.sh): 1 no scaling-vector loads, 2 no d(scores) stores, 3 a multiply for every exponential
#endif
#if CE_PIX_ABL == 3
__device__ __forceinline__ float bk_exp1(float x) { return x * BK_LOG2E; }
#else
__device__ __forceinline__ float bk_exp1(float x) { return __builtin_amdgcn_exp2f(x * BK_LOG2E); }
#endif

// Lane l owns columns 4*(l + 64*c) + q, q < 4, c < NJ/4 (one float4 per chunk: a wave-instruction
// covers 1 KiB of the row).  Needs K % 4 == 0, ld % 4 == 0, 16-B aligned base.
#define BK_COL(lane, c) (4 * ((lane) + 64 * (c)))

// Column pass over a score matrix, BLOCK per group of BK_R rows: thread t owns columns 4 (t + 256 c) + q, c < NJ/16,
// for every row the block visits, so its column accumulators are NJ/4 registers (with a wave per row they were NJ:
// 275 VGPRs with the row itself, one wave per SIMD, 4.2 TB/s) and the block's partial row is written straight from
// registers, no LDS atomics.  The row reductions of the sweep (max, then sum of exp, over all K columns) cross the four
// waves through LDS; BK_R rows travel together (BK_R x NJ/16 float4 loads in flight per thread, two barriers per group).
// MODE BK_FIRSTLSE: the column maxima AND the first pass in one sweep, with no reference known beforehand: per block and column
//                   the running pair (m, s = sum exp((S_nk - m) / eps)) of an online log-sum-exp; part[b][k] = s and
//                   lb_out reused as the float table [gridDim.x][K] of the m (merged exactly by sk_first_from_parts_kernel)
//      BK_COLMAX : cmax[k] = max_n S_nk                     (ordered-uint atomicMax into `cmax_out`)
//      BK_COLSUM : part[b][k] = sum_{rows of block} S_nk
//      BK_FIRST  : part[b][k] = sum exp((S_nk - cmax_k)/eps)
//      BK_SWEEP  : lb_n = log c_n - logsumexp_k(S_nk/eps + la_k); part[b][k] = sum exp(S_nk/eps + la_k + lb_n)
constexpr int BK_R = 4;
#ifndef BK_PASS_MINB
#define BK_PASS_MINB 3       // lab: workgroups per CU the column pass is compiled for (tools/lab/bigk_aux.sh)
#endif
template <int NJ, int MODE>
__global__ __launch_bounds__(256, BK_PASS_MINB) void bigk_pass_kernel(const float* __restrict__ S, long rows, int K, long ld,
                                                           float inv_eps, const unsigned* __restrict__ cmax,
                                                           const float* __restrict__ la, const float* __restrict__ c,
                                                           float c_uniform, float* __restrict__ lb_out,
                                                           float* __restrict__ part, unsigned* __restrict__ cmax_out,
                                                           long s_stride = 0, long c_stride = 0,
                                                           const int* __restrict__ rowmap = nullptr) {
    // rowmap (optional, [problem][rows] int32): row n of the problem is row rowmap[n] of S — a Sinkhorn problem over the
    // SAMPLED rows of a per-pixel score table (DESIGN par. 4.15) without a gathered copy of them
    constexpr int NC = (NJ + 15) / 16;                                   // float4 chunks per thread and row
    constexpr int R = NJ > 80 ? 2 : BK_R;                                // rows per group: two past 5120 columns (four spilled 29 VGPRs at 6 chunks per row)
    int bx = blockIdx.x;
    {   // blockIdx.y = problem of a batch of equally shaped ones: score matrices s_stride apart, K-vectors (cmax, la,
        // cmax_out) K apart, row marginals c_stride apart, partials [problem][block][K]  (gridDim.y == 1: one problem)
        long prob = blockIdx.y;
        if (rowmap && gridDim.y > 1) {
            // problems that are row selections of ONE table (the patches of a view, DESIGN par. 4.15): their row maps are sorted,
            // so block b of every problem walks the same stretch of the table — consecutive dispatch indices go to the SAME
            // block of consecutive problems, all of them resident together, and a table row drawn by several patches is
            // fetched from HBM once and found in the Infinity Cache by the others
            const int L = blockIdx.y * gridDim.x + blockIdx.x;
            prob = L % (int)gridDim.y;
            bx = L / (int)gridDim.y;
        }
        S += prob * s_stride;
        if (cmax) cmax += prob * K;
        if (la) la += prob * K;
        if (c) c += prob * c_stride;
        if (part) part += prob * gridDim.x * K;
        if (cmax_out) cmax_out += prob * K;
        if (lb_out) lb_out += prob * rows;
        if (rowmap) rowmap += prob * rows;
    }
    extern __shared__ __attribute__((aligned(16))) float sm[];           // vec[K] | red[2][R][4]
    float* vec = sm;
    float* red = sm + K;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (MODE == BK_FIRST) {
        const float neg = -inv_eps;       // (a float negation of the decoded value was folded into the decode's
        for (int k = tid; k < K; k += 256) vec[k] = bk_ord2f(cmax[k]) * neg;      //  integer selects and lost: hipcc 7.2)
        __syncthreads();
    } else if (MODE == BK_SWEEP) {
        // log2 units: exp(S/eps + la - max) = 2^(S (log2 e / eps) + la log2 e - max2), one v_exp_f32 per score on an argument
        // <= 0 whose product with log2 e / eps is rounded once (a relative error of |arg| 2^-24 on a term <= 1: what the
        // reference's own `scores / eps` carries into its exp, swav_clustering.py:519)
        for (int k = tid; k < K; k += 256) vec[k] = la[k] * BK_LOG2E;
        __syncthreads();
    }
    const float sc = MODE == BK_SWEEP ? inv_eps * BK_LOG2E : inv_eps;
    f32x4 u[NC];
    f32x4 um[MODE == BK_FIRSTLSE ? NC : 1];                              // BK_FIRSTLSE: the running column maxima
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) {
        const float init = (MODE == BK_COLMAX) ? -INFINITY : 0.f;
        u[cidx] = f32x4{init, init, init, init};
        if (MODE == BK_FIRSTLSE) um[cidx] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    const long ngroups = (rows + R - 1) / R;
    for (long grp = bx; grp < ngroups; grp += gridDim.x) {
        const long r0 = grp * R;
        f32x4 w[R][NC];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // rows past the end read as zeros through an EMPTY descriptor (their contributions are masked below)
            const bool live = r0 + r < rows;
            // (rowmap: a row's index is read right in front of its loads — all R indices in one vector load up front, or a group
            //  ahead, measured 9 % slower / spilled: profiles/r04_lab_sktab.txt)
            const long srow = !live ? 0 : (rowmap ? (long)rowmap[r0 + r] : r0 + r);
            const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(S, srow, ld, live ? K : 0);
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) w[r][cidx] = bk_load4(rs, 4 * (tid + 256 * cidx));
        }
        if (MODE == BK_FIRSTLSE) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r0 + r < rows)
#pragma unroll
                    for (int cidx = 0; cidx < NC; ++cidx)
                        if (4 * (tid + 256 * cidx) < K)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float x = w[r][cidx][q], d = (x - um[cidx][q]) * inv_eps;     // first row: d = +inf, e = 0
                                const float e = bk_exp(-fabsf(d));
                                u[cidx][q] = d > 0.f ? fmaf(u[cidx][q], e, 1.f) : u[cidx][q] + e;
                                um[cidx][q] = fmaxf(um[cidx][q], x);
                            }
        } else if (MODE == BK_COLMAX) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r0 + r < rows)
#pragma unroll
                    for (int cidx = 0; cidx < NC; ++cidx)
#pragma unroll
                        for (int q = 0; q < 4; ++q) u[cidx][q] = fmaxf(u[cidx][q], w[r][cidx][q]);
        } else if (MODE == BK_COLSUM) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int cidx = 0; cidx < NC; ++cidx) u[cidx] += w[r][cidx];          // dead rows / columns read 0
        } else if (MODE == BK_FIRST) {
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const int k = 4 * (tid + 256 * cidx);
                if (k < K) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(vec + k);
#pragma unroll
                    for (int r = 0; r < R; ++r)
                        if (r0 + r < rows)
#pragma unroll
                            for (int q = 0; q < 4; ++q) u[cidx][q] += bk_exp(fmaf(w[r][cidx][q], inv_eps, v[q]));
                }
            }
        } else {
            float mx[R];
#pragma unroll
            for (int r = 0; r < R; ++r) mx[r] = -INFINITY;
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const int k = 4 * (tid + 256 * cidx);
                const bool ok = k < K;
                const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(vec + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w[r][cidx][q] = ok ? fmaf(w[r][cidx][q], sc, v[q]) : -INFINITY;
                        mx[r] = fmaxf(mx[r], w[r][cidx][q]);
                    }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mx[r] = gx_wave_max_valu(mx[r]);
                if (lane == 0) red[r * 4 + wid] = mx[r];
            }
            __syncthreads();
            float sum[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                mx[r] = fmaxf(fmaxf(red[r * 4], red[r * 4 + 1]), fmaxf(red[r * 4 + 2], red[r * 4 + 3]));
                sum[r] = 0.f;
#pragma unroll
                for (int cidx = 0; cidx < NC; ++cidx)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { w[r][cidx][q] = __builtin_amdgcn_exp2f(w[r][cidx][q] - mx[r]); sum[r] += w[r][cidx][q]; }
                sum[r] = gx_wave_sum_valu(sum[r]);
                if (lane == 0) red[R * 4 + r * 4 + wid] = sum[r];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float* rs = red + R * 4 + r * 4;
                const float tot = (rs[0] + rs[1]) + (rs[2] + rs[3]);
                const bool live = r0 + r < rows;
                const float cn = c ? (live ? c[r0 + r] : 0.f) : c_uniform;
                const float f = live ? cn / tot : 0.f;                  // exp(lb + mx)
#pragma unroll
                for (int cidx = 0; cidx < NC; ++cidx) u[cidx] += w[r][cidx] * f;
                if (lb_out && live && tid == 0) lb_out[r0 + r] = logf(cn) - mx[r] * BK_LN2 - logf(tot);
            }
            // (the next group's first barrier orders this group's reads of `red` before its writes: every wave
            //  passes through it only after finishing here)
        }
    }
    // the block's partial row straight from registers (or one global atomicMax per column)
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) {
        const int k = 4 * (tid + 256 * cidx);
        if (k < K) {
            if (MODE == BK_COLMAX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicMax(cmax_out + k + q, bk_f2ord(u[cidx][q]));
            } else {
                *reinterpret_cast<f32x4*>(part + (long)bx * K + k) = u[cidx];
                if (MODE == BK_FIRSTLSE) *reinterpret_cast<f32x4*>(lb_out + (long)bx * K + k) = um[cidx];
            }
        }
    }
}

// Swapped-prediction CE forward + d(scores) for many prototypes, as two streaming kernels (the one-kernel form
// kept a row of codes in registers next to three re-reads of the other view's row: 468 VGPRs, one wave per SIMD,
// 2.4 TB/s):
//   ce_rowstats : per row and view X in {s, t}: wm_X = max_k(X_k/eps + la_X,k), v_X = sum_k exp(. - wm_X)   (codes)
//                                               m_X = max_k X_k,               z_X = sum_k exp((X_k - m_X)/T) (prediction)
//                 one row in registers at a time (NJ VGPRs), both views one after the other;
//   ce_apply    : elementwise over (row, k): q_X = exp(X/eps + la_X - wm_X)/v_X, p_X = exp((X - m_X)/T)/z_X,
//                 dS_t = f (q_s - p_t), dS_s = f (q_t - p_s), loss partial sum_k q_s (x_t - log z_t) + q_t (x_s - log z_s);
//                 a handful of registers per lane, row statistics wave-uniform.
// stats [rows][8] = {wm_s, 1/v_s, m_s, z_s, wm_t, 1/v_t, m_t, z_t}.
template <int NJ>
__global__ __launch_bounds__(256, NJ > 96 ? 2 : 4) void ce_rowstats_bigk_kernel(const float* __restrict__ Ss, const float* __restrict__ St,
                                                                  long rows, int K, long ld, float inv_eps, float inv_T,
                                                                  const float* __restrict__ la_s,
                                                                  const float* __restrict__ la_t,
                                                                  float* __restrict__ stats) {
    constexpr int NC = NJ / 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];           // las[K] | lat[K]
    float* las = sm;
    float* lat = sm + K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) { las[k] = la_s[k]; lat[k] = la_t[k]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    for (long row = wave0; row < rows; row += nwave) {
#pragma unroll
        for (int view = 0; view < 2; ++view) {
            const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(view ? St : Ss, row, ld, K);
            const float* la = view ? lat : las;
            f32x4 w[NC];
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) w[cidx] = bk_load4(rs, BK_COL(lane, cidx));
            float wm = -INFINITY, m = -INFINITY;
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const int k = BK_COL(lane, cidx);
                if (k < K) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(la + k);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        m = fmaxf(m, w[cidx][q]);
                        wm = fmaxf(wm, fmaf(w[cidx][q], inv_eps, v[q]));
                    }
                }
            }
            wm = gx_wave_max(wm); m = gx_wave_max(m);
            float vs = 0.f, z = 0.f;
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const int k = BK_COL(lane, cidx);
                if (k < K) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(la + k);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        vs += bk_exp1(fmaf(w[cidx][q], inv_eps, v[q]) - wm);
                        z += bk_exp1((w[cidx][q] - m) * inv_T);
                    }
                }
            }
            vs = gx_wave_sum(vs); z = gx_wave_sum(z);
            if (lane == 0) {
                float* o = stats + row * 8 + view * 4;
                o[0] = wm; o[1] = 1.f / vs; o[2] = m; o[3] = z;
            }
        }
    }
}

__global__ __launch_bounds__(256, 4) void ce_apply_bigk_kernel(const float* __restrict__ Ss, const float* __restrict__ St,
                                                               long rows, int K, long ld, float inv_eps, float inv_T,
                                                               const float* __restrict__ la_s,
                                                               const float* __restrict__ la_t,
                                                               const float* __restrict__ stats, float gscale,
                                                               float* __restrict__ dSs, float* __restrict__ dSt, long ldd,
                                                               float* __restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];           // las[K] | lat[K]
    float* las = sm;
    float* lat = sm + K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) { las[k] = la_s[k]; lat[k] = la_t[k]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    const float f = -0.5f * gscale * inv_T;
    const bool write = dSs != nullptr;
    float lacc = 0.f;
    for (long row = wave0; row < rows; row += nwave) {
        const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(Ss, row, ld, K), rt = bk_row_rsrc(St, row, ld, K);
        const __amdgpu_buffer_rsrc_t ds = bk_row_rsrc(write ? dSs : Ss, row, write ? ldd : ld, K);
        const __amdgpu_buffer_rsrc_t dt = bk_row_rsrc(write ? dSt : St, row, write ? ldd : ld, K);
        const float* st = stats + row * 8;                               // wave-uniform
        const float wm_s = st[0], iv_s = st[1], m_s = st[2], z_s = st[3];
        const float wm_t = st[4], iv_t = st[5], m_t = st[6], z_t = st[7];
        const float iz_s = 1.f / z_s, iz_t = 1.f / z_t, lz_s = logf(z_s), lz_t = logf(z_t);
#pragma unroll 4
        for (int k = 4 * lane; k < K; k += 256) {
            const f32x4 a = bk_load4(rs, k), b = bk_load4(rt, k);
            const f32x4 ls = *reinterpret_cast<const f32x4*>(las + k), lt = *reinterpret_cast<const f32x4*>(lat + k);
            f32x4 d_s, d_t;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float qs = bk_exp1(fmaf(a[q], inv_eps, ls[q]) - wm_s) * iv_s;
                const float qt = bk_exp1(fmaf(b[q], inv_eps, lt[q]) - wm_t) * iv_t;
                const float xs = (a[q] - m_s) * inv_T, xt = (b[q] - m_t) * inv_T;
                lacc += qs * (xt - lz_t) + qt * (xs - lz_s);
                d_t[q] = f * (qs - bk_exp1(xt) * iz_t);
                d_s[q] = f * (qt - bk_exp1(xs) * iz_s);
            }
            if (write) { bk_store4(dt, k, d_t); bk_store4(ds, k, d_s); }
        }
    }
    __shared__ float red[16];
    const float tot = gx_block_sum(lacc, red);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = -0.5f * tot;
}

// The swapped CE as ONE pass (block per pair of rows, columns spread over the 256 threads as in bigk_pass_kernel):
// both views' rows stay in registers (2 rows x 2 views x NJ/4 floats per thread) between the row statistics — two
// block reductions through LDS — and the element-wise part, so the score matrices are read once instead of twice,
// and the column sums of dS (the prototype-bias gradient, until now a pass of its own over dS) leave as the block's
// partial row.  4 passes over a (rows, K) matrix per view pair (read 2, write 2) instead of 7.
//   part_s / part_t [gridDim.x][K]: column sums of dSs / dSt over the block's rows
constexpr int CE_R = 2;
template <int NJ>
__global__ __launch_bounds__(256, NJ > 80 ? 2 : 3) void ce_fused_bigk_kernel(const float* __restrict__ Ss, const float* __restrict__ St,
                                                               long rows, int K, long ld, float inv_eps, float inv_T,
                                                               const float* __restrict__ la_s, const float* __restrict__ la_t,
                                                               float gscale, float* __restrict__ dSs, float* __restrict__ dSt,
                                                               long ldd, float* __restrict__ loss_part,
                                                               float* __restrict__ part_s, float* __restrict__ part_t) {
    constexpr int NC = (NJ + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];           // las[K] | lat[K] | red[2][CE_R][4 stats][4 waves]
    float* las = sm;
    float* lat = sm + K;
    float* red = sm + 2 * K;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int k = tid; k < K; k += 256) { las[k] = la_s[k]; lat[k] = la_t[k]; }
    __syncthreads();
    const float f = -0.5f * gscale * inv_T;
    f32x4 cs[NC], ct[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { cs[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ct[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float lacc = 0.f;
    const long ngroups = (rows + CE_R - 1) / CE_R;
    for (long grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const long r0 = grp * CE_R;
        f32x4 a[CE_R][NC], b[CE_R][NC];
#pragma unroll
        for (int r = 0; r < CE_R; ++r) {
            const bool live = r0 + r < rows;
            const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(Ss, live ? r0 + r : 0, ld, live ? K : 0);
            const __amdgpu_buffer_rsrc_t rt = bk_row_rsrc(St, live ? r0 + r : 0, ld, live ? K : 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) { a[r][c] = bk_load4(rs, 4 * (tid + 256 * c)); b[r][c] = bk_load4(rt, 4 * (tid + 256 * c)); }
        }
        // maxima: wm_X = max_k (X_k / eps + la_X,k), m_X = max_k X_k
        float mx[CE_R][4];
#pragma unroll
        for (int r = 0; r < CE_R; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) mx[r][j] = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = 4 * (tid + 256 * c);
            if (k < K) {
                const f32x4 ls = *reinterpret_cast<const f32x4*>(las + k), lt = *reinterpret_cast<const f32x4*>(lat + k);
#pragma unroll
                for (int r = 0; r < CE_R; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        mx[r][0] = fmaxf(mx[r][0], fmaf(a[r][c][q], inv_eps, ls[q]));
                        mx[r][1] = fmaxf(mx[r][1], a[r][c][q]);
                        mx[r][2] = fmaxf(mx[r][2], fmaf(b[r][c][q], inv_eps, lt[q]));
                        mx[r][3] = fmaxf(mx[r][3], b[r][c][q]);
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < CE_R; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = gx_wave_max(mx[r][j]);
                if (lane == 0) red[(r * 4 + j) * 4 + wid] = v;
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CE_R; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* p = red + (r * 4 + j) * 4;
                mx[r][j] = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
            }
        // sums: v_X = sum_k exp(X_k / eps + la_X,k - wm_X), z_X = sum_k exp((X_k - m_X) / T)
        float sv[CE_R][4];
#pragma unroll
        for (int r = 0; r < CE_R; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[r][j] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = 4 * (tid + 256 * c);
            if (k < K) {
                const f32x4 ls = *reinterpret_cast<const f32x4*>(las + k), lt = *reinterpret_cast<const f32x4*>(lat + k);
#pragma unroll
                for (int r = 0; r < CE_R; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sv[r][0] += bk_exp1(fmaf(a[r][c][q], inv_eps, ls[q]) - mx[r][0]);
                        sv[r][1] += bk_exp1((a[r][c][q] - mx[r][1]) * inv_T);
                        sv[r][2] += bk_exp1(fmaf(b[r][c][q], inv_eps, lt[q]) - mx[r][2]);
                        sv[r][3] += bk_exp1((b[r][c][q] - mx[r][3]) * inv_T);
                    }
            }
        }
        float* red2 = red + CE_R * 16;
#pragma unroll
        for (int r = 0; r < CE_R; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = gx_wave_sum(sv[r][j]);
                if (lane == 0) red2[(r * 4 + j) * 4 + wid] = v;
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CE_R; ++r) {
            if (r0 + r >= rows) continue;                                // block-uniform
            float t4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* p = red2 + (r * 4 + j) * 4;
                t4[j] = (p[0] + p[1]) + (p[2] + p[3]);
            }
            const float wm_s = mx[r][0], m_s = mx[r][1], wm_t = mx[r][2], m_t = mx[r][3];
            const float iv_s = 1.f / t4[0], z_s = t4[1], iv_t = 1.f / t4[2], z_t = t4[3];
            const float iz_s = 1.f / z_s, iz_t = 1.f / z_t, lz_s = logf(z_s), lz_t = logf(z_t);
            const __amdgpu_buffer_rsrc_t ds = bk_row_rsrc(dSs, r0 + r, ldd, K), dt = bk_row_rsrc(dSt, r0 + r, ldd, K);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = 4 * (tid + 256 * c);
                if (k < K) {
                    const f32x4 ls = *reinterpret_cast<const f32x4*>(las + k), lt = *reinterpret_cast<const f32x4*>(lat + k);
                    f32x4 d_s, d_t;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float qs = bk_exp1(fmaf(a[r][c][q], inv_eps, ls[q]) - wm_s) * iv_s;
                        const float qt = bk_exp1(fmaf(b[r][c][q], inv_eps, lt[q]) - wm_t) * iv_t;
                        const float xs = (a[r][c][q] - m_s) * inv_T, xt = (b[r][c][q] - m_t) * inv_T;
                        lacc += qs * (xt - lz_t) + qt * (xs - lz_s);
                        d_t[q] = f * (qs - bk_exp1(xt) * iz_t);
                        d_s[q] = f * (qt - bk_exp1(xs) * iz_s);
                    }
                    bk_store4(dt, k, d_t); bk_store4(ds, k, d_s);
                    cs[c] += d_s; ct[c] += d_t;
                }
            }
        }
        // (the next group's first barrier orders this group's reads of `red2` before its writes, and its second
        //  barrier those of `red`)
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = 4 * (tid + 256 * c);
        if (k < K) {
            *reinterpret_cast<f32x4*>(part_s + (long)blockIdx.x * K + k) = cs[c];
            *reinterpret_cast<f32x4*>(part_t + (long)blockIdx.x * K + k) = ct[c];
        }
    }
    __syncthreads();
    const float tot = gx_block_sum(lacc, red);
    if (tid == 0) loss_part[blockIdx.x] = -0.5f * tot;
}

// The swapped CE of ALL patches of a step, pixel-major (DESIGN par. 4.15): a block owns a pixel of the (2 views x B images)
// grid; for every patch that drew the pixel (inv[p][pixel] = its row in that patch's score matrices, -1 = not drawn) it
// does what ce_fused_bigk_kernel does for one row pair — statistics, loss share, d(scores) — and ADDS the two d(scores)
// rows up in registers; the pixel's rows of the per-view tables dSs / dSt [npix][ldd] are written ONCE (zeros for a
// pixel no patch drew; columns up to kpad zeroed), so the backward GEMMs run over pixels instead of samples with no
// read-modify-write of the tables (as a row-indexed accumulate the CE moved 2.3x the bytes and ran 3x longer).
//   S_all [P][2][rows][ld]; la_all [(p * 2 + view) * ngroups + group][K], group = pixel / pix_per_group
constexpr int CE_PMAX = 8;                  // patches per step the pixel-major CE takes
#ifndef CE_PIX_NJ2
#define CE_PIX_NJ2 80       // widest rows (64-column chunks per lane) compiled for two workgroups per CU (lab: tools/lab/bigk_minb.sh)
#endif
// (two workgroups of 256 threads per CU up to 5120 columns; past that the rows and their exponentials need more than 256 registers per
//  thread — 188 spilled at 6144 columns —: ONE workgroup of 512 threads per CU, half the columns per thread; round 3 ran 256
//  threads there with the AGPRs as the second half of the register file: four waves per CU, 43.6 ms at K = 8000)
#ifndef CE_PIX_BT_WIDE
#define CE_PIX_BT_WIDE 512     // threads per workgroup past CE_PIX_NJ2 (lab: 512 / 1024)
#endif
template <int NJ, int BT = (NJ > CE_PIX_NJ2 ? CE_PIX_BT_WIDE : 256)>
__global__ __launch_bounds__(BT, BT == 256 ? 2 : 1) void ce_pixels_bigk_kernel(const float* __restrict__ S_all, int P, long rows, int K, long ld,
                                                                const int* __restrict__ inv, long npix, int ngroups,
                                                                long pix_per_group, const float* __restrict__ la_all,
                                                                float inv_eps, float inv_T, float gscale,
                                                                float* __restrict__ dSs, float* __restrict__ dSt, long ldd,
                                                                int kpad, float* __restrict__ loss_part,
                                                                float* __restrict__ part_s, float* __restrict__ part_t,
                                                                const float* __restrict__ St_tab = nullptr,
                                                                const int* __restrict__ live = nullptr,
                                                                const int* __restrict__ row_group = nullptr) {
    // St_tab != NULL: the scores are per-pixel TABLES too (S_all = the first view's [npix][ld], St_tab the second's): a
    // pixel's two score rows are read once for all the patches that drew it.  live != NULL: the drawn pixels are the first
    // *live rows of the tables (DESIGN §4.15 d) — the d(scores) rows past them are neither read by the GEMMs nor written here
    const long nlive = live ? min(npix, (long)max(0, *live)) : npix;      // (npix stays the row length of `inv`)
    constexpr int NC = (NJ * 64 + 4 * BT - 1) / (4 * BT), NW = BT / 64;      // 16-byte chunks per thread; waves
    __shared__ __attribute__((aligned(16))) float red[2 * 4 * NW];
    // the block's column sums of both tables (the bias gradient) live in LDS, every thread adding to its own columns: in
    // registers (40 more) the kernel spilled 77 dwords at three workgroups per CU and ran at 1.8 TB/s
    extern __shared__ __attribute__((aligned(16))) float csum[];         // [2][K]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float f = -0.5f * gscale * inv_T;
    for (int k = tid; k < 2 * K; k += BT) csum[k] = 0.f;
    __syncthreads();          // a block without a pixel (or whose first pixel nobody drew) reaches its csum reads with no other barrier
    float lacc = 0.f;
    for (long pix = blockIdx.x; pix < nlive; pix += gridDim.x) {
        f32x4 as[NC], at[NC];                                            // the pixel's accumulated d(scores), both views
#pragma unroll
        for (int c = 0; c < NC; ++c) { as[c] = f32x4{0.f, 0.f, 0.f, 0.f}; at[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // the Sinkhorn problem (image) of the pixel: equal runs of pixels, or — tables with the drawn pixels first — looked up per row
        const long grp = ngroups > 1 ? (row_group ? (long)__builtin_amdgcn_readfirstlane(row_group[pix]) : pix / pix_per_group) : 0;
        int nrow[CE_PMAX];                                               // the pixel's row in every patch, fetched together
#pragma unroll
        for (int p = 0; p < CE_PMAX; ++p) nrow[p] = p < P ? __builtin_amdgcn_readfirstlane(inv[(long)p * npix + pix]) : -1;
        f32x4 a[NC], b[NC];
        if (St_tab) {
            bool any = false;
#pragma unroll
            for (int p = 0; p < CE_PMAX; ++p) any = any || nrow[p] >= 0;
            const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(S_all, pix, ld, any ? K : 0), rt = bk_row_rsrc(St_tab, pix, ld, any ? K : 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) { a[c] = bk_load4(rs, 4 * (tid + BT * c)); b[c] = bk_load4(rt, 4 * (tid + BT * c)); }
        }
#pragma unroll
        for (int p = 0; p < CE_PMAX; ++p) {
            const int n = nrow[p];
            if (n < 0) continue;                                         // block-uniform
            const float* las = la_all + ((long)(p * 2) * ngroups + grp) * K;
            const float* lat = la_all + ((long)(p * 2 + 1) * ngroups + grp) * K;
            if (!St_tab) {
                const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(S_all + (long)(p * 2) * rows * ld, n, ld, K);
                const __amdgpu_buffer_rsrc_t rt = bk_row_rsrc(S_all + (long)(p * 2 + 1) * rows * ld, n, ld, K);
#pragma unroll
                for (int c = 0; c < NC; ++c) { a[c] = bk_load4(rs, 4 * (tid + BT * c)); b[c] = bk_load4(rt, 4 * (tid + BT * c)); }
            }
            // the patch's la vectors (L2-resident: 10 x 20 KB; LDS has no room for all of them) are read ONCE per row pair into
            // registers, not once per use: 10 instead of 30 loads per thread — at two workgroups per CU, where the registers
            // for it exist (38.9 ms per step against 45.4 with three workgroups and the vectors re-read; same-box A/B)
            f32x4 lsr[NC], ltr[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = min(4 * (tid + BT * c), K - 4);
                lsr[c] = *reinterpret_cast<const f32x4*>(las + k); ltr[c] = *reinterpret_cast<const f32x4*>(lat + k);
#if CE_PIX_ABL == 1
                lsr[c] = f32x4{0.1f, 0.2f, 0.3f, 0.4f}; ltr[c] = f32x4{0.1f, 0.2f, 0.3f, 0.4f};      // lab: no scaling-vector loads
#endif
            }
#define CE_LS(c, k) lsr[c]
#define CE_LT(c, k) ltr[c]
            float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = 4 * (tid + BT * c);
                if (k < K) {
                    const f32x4 ls = CE_LS(c, k), lt = CE_LT(c, k);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        mx[0] = fmaxf(mx[0], fmaf(a[c][q], inv_eps, ls[q]));
                        mx[1] = fmaxf(mx[1], a[c][q]);
                        mx[2] = fmaxf(mx[2], fmaf(b[c][q], inv_eps, lt[q]));
                        mx[3] = fmaxf(mx[3], b[c][q]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = gx_wave_max(mx[j]);
                if (lane == 0) red[j * NW + wid] = v;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < NW; i += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(red + j * NW + i);
                    m = fmaxf(m, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
                }
                mx[j] = m;
            }
            // the four exponentials of an element are evaluated ONCE: the sums pass leaves them in registers (the code ones
            // over the la registers, dead by then) and the element-wise pass only scales them — v_exp_f32 is a quarter-rate
            // instruction and the pass is VALU-bound
            float sv[4] = {0.f, 0.f, 0.f, 0.f};
            f32x4 es[NC], et[NC];                                        // exp((x - m) / T) of both views
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = 4 * (tid + BT * c);
                es[c] = f32x4{0.f, 0.f, 0.f, 0.f}; et[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (k < K) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        lsr[c][q] = bk_exp1(fmaf(a[c][q], inv_eps, lsr[c][q]) - mx[0]);      // lsr / ltr now hold the code numerators
                        es[c][q] = bk_exp1((a[c][q] - mx[1]) * inv_T);
                        ltr[c][q] = bk_exp1(fmaf(b[c][q], inv_eps, ltr[c][q]) - mx[2]);
                        et[c][q] = bk_exp1((b[c][q] - mx[3]) * inv_T);
                        sv[0] += lsr[c][q]; sv[1] += es[c][q]; sv[2] += ltr[c][q]; sv[3] += et[c][q];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = gx_wave_sum(sv[j]);
                if (lane == 0) red[4 * NW + j * NW + wid] = v;
            }
            __syncthreads();
            float t4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NW; i += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(red + 4 * NW + j * NW + i);
                    t += (v[0] + v[1]) + (v[2] + v[3]);
                }
                t4[j] = t;
            }
            const float m_s = mx[1], m_t = mx[3];
            const float iv_s = 1.f / t4[0], z_s = t4[1], iv_t = 1.f / t4[2], z_t = t4[3];
            const float iz_s = 1.f / z_s, iz_t = 1.f / z_t, lz_s = logf(z_s), lz_t = logf(z_t);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int k = 4 * (tid + BT * c);
                if (k < K) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float qs = lsr[c][q] * iv_s, qt = ltr[c][q] * iv_t;
                        const float xs = (a[c][q] - m_s) * inv_T, xt = (b[c][q] - m_t) * inv_T;
                        lacc += qs * (xt - lz_t) + qt * (xs - lz_s);
                        at[c][q] += f * (qs - et[c][q] * iz_t);
                        as[c][q] += f * (qt - es[c][q] * iz_s);
                    }
                }
            }
            // (the next patch's first barrier orders these reads of the sums' half of `red` before its writes, its second those of the maxima's)
        }
        const __amdgpu_buffer_rsrc_t ds = bk_row_rsrc(dSs, pix, ldd, kpad), dt = bk_row_rsrc(dSt, pix, ldd, kpad);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = 4 * (tid + BT * c);
#if CE_PIX_ABL == 2
            if (k < kpad && as[c][0] == 123.f) {                         // lab: no d(scores) stores
#else
            if (k < kpad) {                                              // columns K .. kpad - 1 hold zeros: the dz GEMM reads them
#endif
                bk_store4(ds, k, as[c]); bk_store4(dt, k, at[c]);
            }
            if (k < K) {
                *reinterpret_cast<f32x4*>(csum + k) += as[c];
                *reinterpret_cast<f32x4*>(csum + K + k) += at[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int k = 4 * (tid + BT * c);
        if (k < K) {
            *reinterpret_cast<f32x4*>(part_s + (long)blockIdx.x * K + k) = *reinterpret_cast<const f32x4*>(csum + k);
            *reinterpret_cast<f32x4*>(part_t + (long)blockIdx.x * K + k) = *reinterpret_cast<const f32x4*>(csum + K + k);
        }
    }
    __syncthreads();
    __shared__ float red2[16];
    const float tot = gx_block_sum(lacc, red2);
    if (tid == 0) loss_part[blockIdx.x] = -0.5f * tot;
}
#undef CE_LS
#undef CE_LT

static inline int bigk_nj(int K) {          // registers per lane for a row of K scores; 0 = not supported
    const int need = (K + 63) / 64;
    const int opts[] = {16, 32, 48, 64, 80, 96, 128};
    for (int o : opts)
        if (need <= o) return o;
    return 0;
}
// the column passes (Sinkhorn sweeps, column maxima / sums) take up to 8192 columns — the generic configuration's 8000 prototypes —,
// the cross-entropy kernels, which hold several rows and their exponentials in registers, up to 6144
static inline bool bigk_ok(int K) { return K > 64 && (K % 4) == 0 && bigk_nj(K) != 0; }
static inline bool bigk_ce_ok(int K) { return bigk_ok(K) && bigk_nj(K) <= 96; }
static inline bool bigk_aligned(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld % 4) == 0; }

#define GX_BIGK_DISPATCH(NJV, CALL)            \
    switch (NJV) {                             \
        case 16: { constexpr int NJ = 16; CALL; } break; \
        case 32: { constexpr int NJ = 32; CALL; } break; \
        case 48: { constexpr int NJ = 48; CALL; } break; \
        case 64: { constexpr int NJ = 64; CALL; } break; \
        case 80: { constexpr int NJ = 80; CALL; } break; \
        default: { constexpr int NJ = 96; CALL; } break; \
    }
#define GX_BIGK_DISPATCH_PASS(NJV, CALL)       \
    switch (NJV) {                             \
        case 16: { constexpr int NJ = 16; CALL; } break; \
        case 32: { constexpr int NJ = 32; CALL; } break; \
        case 48: { constexpr int NJ = 48; CALL; } break; \
        case 64: { constexpr int NJ = 64; CALL; } break; \
        case 80: { constexpr int NJ = 80; CALL; } break; \
        case 96: { constexpr int NJ = 96; CALL; } break; \
        default: { constexpr int NJ = 128; CALL; } break; \
    }

static inline int bigk_blocks(long rows, int cap) {
    long nb = (rows + 3) / 4;
    return (int)(nb < cap ? nb : cap);
}
static inline size_t bigk_pass_lds(int K) { return ((size_t)K + 2 * BK_R * 4) * sizeof(float); }

}  // namespace gx
