// Sinkhorn-Knopp on the per-pixel score TABLES of a many-prototype step (DESIGN par. 4.15 / 4.16; swav_clustering.py:509-544
// once per view and patch).  A step's problems are row selections of two tables (one per view): problem (p, v) = the rows
// of table v that patch p drew.  The per-problem kernels (gx_bigk.h, gx_sinkhorn_rows_batch) sweep every problem's rows on
// their own — a table row that several patches drew (1.8 draws per drawn pixel at 5 x 20 000 of 65 536) is read once per
// draw.  Here ONE pass over the live rows of a table serves all its patches: a row is loaded once, and for every patch that
// drew it (inv[p][row] >= 0) the pass does what bigk_pass_kernel<., BK_SWEEP> does for a row of that problem.
//
// Work split: a workgroup of T = 64 ceil(K / 512) threads (640 at K = 5000) owns groups of 4 consecutive table rows; thread t
// holds columns 4 (t + T c) + q, c < 2, of every row it visits, so a problem's column accumulators are 8 registers per
// thread (P problems: 8 P) and leave as the block's partial row; the patches' log scaling vectors sit in LDS (P x 8T
// floats, in log2 units: one workgroup per CU), the next group's rows fly into a second set of row registers meanwhile.
// The table rows are ordered by WHICH patches drew them (swav_clustering.py _score_tables), so the four rows of a group go
// through a patch together — four independent reduction chains, no branch per row — and a patch none of them belongs to
// costs nothing.  The row reductions cross the waves through LDS: a wave's four partials are reduced together (7 cross-lane
// steps, skt_reduce4), lane i < 16 of every wave then reads wave i's and the same butterfly joins them.
// History (bench shape, both views, ms per pass; tools/lab/sktab_lab.py, profiles/r04_lab_sktab.txt): 22.9 with the membership
// read row by row through scalar-looking loads that hipcc made vector loads + vmcnt(0) — each also waiting for the row loads
// before it; 20.0 without the compiler's NW-long reduction loops; 13.6 with the patches software-pipelined (one barrier per
// patch, a branch per row and patch: the rows of a group rarely share their patches in pixel order); 10.9 with the rows
// grouped by membership and a group's rows going through a patch together; 8.5 with the four-value butterfly and one
// v_max_f32 per step (fmaxf behind a DPP move costs three instructions); 7.3 in the shifted form (below).  The per-problem
// sweeps it replaces: 10.0 (gx_sinkhorn_rows_batch with the same one-instruction exp), bound by the 12.6 B/clk a CU can load —
// 7.1 TB/s of row bytes whether they come from HBM or the Infinity Cache — so the way on was fewer bytes INTO the CU.
// Tried and dropped: pulling the next group towards L2 with one 4-byte load (or LDS-DMA) per thread and 128-byte line: a touch
// moves its whole line to the CU — twice the line requests and fabric bytes per row, sweeps 1.7x slower.
// exp: 2^(t log2 e / eps ...) by v_exp_f32 on arguments <= 0 whose product with log2 e / eps is rounded once: a relative
// error of |arg| 2^-24 on a term e^arg <= 1, i.e. at most 0.37 x 2^-24 of the row's largest term — the error the reference's
// own `scores / eps` carries into its exp (swav_clustering.py:519).
#include "gx_bigk_io.h"

namespace gx {

#ifndef SKT_ABL
#define SKT_ABL 0                   // lab ablation: 1 = the sweeps load their rows and do nothing with them
#endif
constexpr int SKT_R = 4;            // table rows per group
constexpr int SKT_MAXT = 768;       // threads per workgroup at most (K <= 6144)
constexpr int SKT_MAXW = SKT_MAXT / 64;
constexpr int SKT_LA_CAP = 32768;   // floats of LDS for the patches' scaling vectors (P x 8T)
constexpr int SKT_LA_CAP4 = 40000;  // ... of the wide form (16 columns per thread, K <= 8192: P x K floats)
constexpr int SKT_T4 = 512;         // threads per workgroup at most in the wide form
constexpr float SKT_LOG2E = 1.44269504088896340736f;

struct SkTabArgs {
    const float* S[2];       // score tables of the two views, [npix][ld]
    long ld;
    int K;
    const float* wt;         // [nv][P][npix] row weights: the row marginal c_n of table row q in patch p's problem, 0 = not in it
    long wt_view;            // stride between the views' weights (0: both views share them — uniform marginals)
    long npix;
    const int* live;         // device scalar: the drawn pixels are table rows [0, *live) (NULL: npix)
    const int* seg;          // per-image scope: [ngrp + 1] device ints, image b's live rows are table rows [seg[b], seg[b + 1])
    int ngrp;                // Sinkhorn problems per (patch, view): 1, or the images (blockIdx.z = image); problem index
                             // = (p * 2 + view) * ngrp + image in every per-problem array
    float sc;                // log2(e) / eps
    const float* la;         // [(p * 2 + view)][K] log scaling vectors (sweep)
    const float* stats;      // [(p * 2 + view)][2] = {largest step of la since the last sweep (log2 units), 1 if the shifted form is safe}
                             // from the update before this sweep, or NULL: every row's maximum is taken exactly
    float* ltab;             // [2][P][npix] log2-sum-exp of every (view, patch, table row) under the la of the sweep that wrote it
    float* part;             // [(p * 2 + view)][nb][K] column sums of the block (first pass: relative to partm)
    float* partm;            // first pass: [(p * 2 + view)][nb][K] running column maxima of the block
    int nb;
};

__device__ __forceinline__ float skt_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// workgroup barrier that retires this wave's LDS writes but leaves its vector-memory loads in flight (__syncthreads() may
// drain them: csrc/gx_fineq.hip)
__device__ __forceinline__ void skt_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Row weights of a pass, made once per call from the index tables (a few tens of microseconds): wt[v][p][q] = the row marginal
// of table row q in problem (p, v) — c_uniform, or c[(p * 2 + v)][inv[p][q]] — and 0 when patch p did not draw it or q >= live.
// The sweeps then learn everything about a group of rows from ONE vector load (lane p * R + r), issued with the rows.
__global__ void sk_table_weights_kernel(const int* __restrict__ inv, long npix, int P, const int* __restrict__ live,
                                        const float* __restrict__ c, long rows, float c_uniform, float* __restrict__ wt) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y, v = blockIdx.z;
    if (q >= npix) return;
    const long lv = live ? (long)*live : npix;
    const int iv = q < lv ? inv[(long)p * npix + q] : -1;
    float w = 0.f;
    if (iv >= 0) w = c ? c[(long)(p * 2 + v) * rows + iv] : c_uniform;
    wt[((long)v * P + p) * npix + q] = w;
}

// the weights of the R rows of group g under the P patches: lane p * R + r of every wave holds wt[p][g R + r]
template <int P, int R = SKT_R>
__device__ __forceinline__ float skt_group_weights(const float* __restrict__ wt, long npix, long lo, long hi, long g, int lane) {
    const int p = lane / R, r = lane % R;
    const long row = lo + g * R + r;
    float w = 0.f;
    if (lane < P * R && row < hi) w = wt[(long)p * npix + row];
    return w;
}
// the rows [lo, hi) this workgroup's problems live on: all live rows, or those of image blockIdx.z
__device__ __forceinline__ void skt_window(const SkTabArgs& a, long& lo, long& hi) {
    lo = 0;
    hi = a.live ? (long)__builtin_amdgcn_readfirstlane(*a.live) : a.npix;
    if (a.seg) {
        lo = (long)__builtin_amdgcn_readfirstlane(a.seg[blockIdx.z]);
        hi = (long)__builtin_amdgcn_readfirstlane(a.seg[blockIdx.z + 1]);
    }
}
__device__ __forceinline__ float skt_lane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// ---- the column maxima AND the first pass of all problems in one sweep of the tables (online log-sum-exp per block and
//      column, as bigk_pass_kernel<., BK_FIRSTLSE>; the (max, sum) pairs are merged exactly by sk_first_from_parts_kernel)
template <int P, int NC = 2>       // NC: 16-byte column chunks per thread (2; 4 = the wide form for 6144 < K <= 8192, 512 threads)
__global__ __launch_bounds__(NC == 2 ? SKT_MAXT : SKT_T4) void sk_table_first_kernel(SkTabArgs a) {
    // the next group's rows are in flight under this one's arithmetic; four rows per group while the registers last (the
    // running (max, sum) pairs are 8 NC P of them), two past three patches, one in the wide form
    constexpr bool PF = true;
    constexpr int R = NC == 2 ? (P <= 3 ? SKT_R : 2) : 1;
    const int T = blockDim.x, tid = threadIdx.x;
    const int view = blockIdx.y, K = a.K;
    const float* S = a.S[view];
    long lo, live;
    skt_window(a, lo, live);
    const long ngroups = (live - lo + R - 1) / R;
    const int img = blockIdx.z;
    const float inv_eps2 = a.sc;                                       // log2(e) / eps
    const float* wt = a.wt + view * a.wt_view;
    const int lane = tid & 63;
    f32x4 m[P][NC], s[P][NC];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            m[p][c] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            s[p][c] = zero4();
        }
    f32x4 w[R][NC];
    float gw;
    auto load = [&](long g, f32x4 (&dst)[R][NC], float& gwd) {
        gwd = skt_group_weights<P, R>(wt, a.npix, lo, live, g, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = lo + g * R + r;
            const bool ok = row < live;
            const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(S, ok ? row : 0, a.ld, ok ? K : 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[r][c] = bk_load4(rs, 4 * (tid + T * c));
        }
    };
    if (PF) load(blockIdx.x, w, gw);
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        f32x4 wn[PF ? R : 1][NC];
        float gwn = 0.f;
        if constexpr (PF) load(g + gridDim.x, wn, gwn);
        else load(g, w, gw);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (skt_lane(gw, p * R + r) > 0.f) {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float x = w[r][c][q], d = (x - m[p][c][q]) * inv_eps2;     // first row: d = +inf, e = 0
                            const float e = skt_exp2(-fabsf(d));
                            s[p][c][q] = d > 0.f ? fmaf(s[p][c][q], e, 1.f) : s[p][c][q] + e;
                            m[p][c][q] = fmaxf(m[p][c][q], x);
                        }
                }
        if constexpr (PF) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int c = 0; c < NC; ++c) w[r][c] = wn[r][c];
            }
            gw = gwn;
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const long o = (((long)(p * 2 + view) * a.ngrp + img) * a.nb + blockIdx.x) * K;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = 4 * (tid + T * c);
            if (k < K) {
                *reinterpret_cast<f32x4*>(a.part + o + k) = s[p][c];
                *reinterpret_cast<f32x4*>(a.partm + o + k) = m[p][c];
            }
        }
    }
}

// ---- one Sinkhorn sweep of all problems: part[(p, v)][b][k] = sum over the block's rows n of problem (p, v) of
//      c_n exp(S_nk / eps + la_k) / sum_j exp(S_nj / eps + la_j)
// A group's R rows go through a patch TOGETHER (four independent reduction chains, no branch per row): maxima -> barrier ->
// exponentials and sums -> barrier -> accumulate, and a patch none of them belongs to costs nothing, not even a barrier.  The
// table rows are ordered by their membership mask (ganecdotes_amd/swav_clustering.py _score_tables), so the rows of a group
// almost always belong to the same patches — 1.8 of 5 on average; a row of a mixed group that is not in the patch rides along
// with weight 0.
// Four values per lane (the group's four rows) reduced over the wave TOGETHER: two butterfly steps hand half of the values to
// the partner lane each (lane ^ 1 keeps rows {0, 1} or {2, 3}, lane ^ 2 one of its two), after which a lane holds ONE row's
// sum over its quad — row 2 (lane & 1) + ((lane >> 1) & 1) — and the quads are joined by two rotations inside the row of 16
// lanes and the two permlane swaps: 7 cross-lane steps instead of 4 x 6.  Inline asm: one instruction per step (fmaxf behind a
// DPP move costs hipcc a v_mov_dpp and two canonicalising v_max), the s_nop in front of every DPP step is the wait its
// freshly written source needs.
#define SKT_DPP_OP(name, op, ctrl)                                                                  \
    __device__ __forceinline__ float name(float give, float keep) {                                 \
        float r;                                                                                    \
        asm volatile("s_nop 1\n\t" op "_dpp %0, %1, %2 " ctrl " row_mask:0xf bank_mask:0xf"         \
                     : "=v"(r) : "v"(give), "v"(keep));                                           \
        return r;                                                                                   \
    }
SKT_DPP_OP(skt_add_x1, "v_add_f32", "quad_perm:[1,0,3,2]")
SKT_DPP_OP(skt_add_x2, "v_add_f32", "quad_perm:[2,3,0,1]")
SKT_DPP_OP(skt_add_r4, "v_add_f32", "row_ror:4")
SKT_DPP_OP(skt_add_r8, "v_add_f32", "row_ror:8")
SKT_DPP_OP(skt_max_x1, "v_max_f32", "quad_perm:[1,0,3,2]")
SKT_DPP_OP(skt_max_x2, "v_max_f32", "quad_perm:[2,3,0,1]")
SKT_DPP_OP(skt_max_r4, "v_max_f32", "row_ror:4")
SKT_DPP_OP(skt_max_r8, "v_max_f32", "row_ror:8")
#undef SKT_DPP_OP
__device__ __forceinline__ float skt_vmax(float a, float b) {        // v_max_f32 with no canonicalising twin (no NaN can reach it)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float skt_swap16(float x, bool is_max) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return is_max ? skt_vmax(a, b) : a + b;
}
__device__ __forceinline__ float skt_swap32(float x, bool is_max) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return is_max ? skt_vmax(a, b) : a + b;
}
// row held by `lane` after the butterfly, and the lane (0 .. 3) that holds row r
__device__ __forceinline__ int skt_row_of_lane(int lane) { return 2 * (lane & 1) + ((lane >> 1) & 1); }
__device__ __forceinline__ int skt_lane_of_row(int r) { return 2 * (r & 1) + (r >> 1); }
// WIDE: over all 64 lanes (a wave's partial of each row); otherwise over each row of 16 lanes (the waves' partials, one per lane)
template <bool MAX, bool WIDE>
__device__ __forceinline__ float skt_reduce4(f32x4 v, int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const float k0 = b0 ? v[2] : v[0], k1 = b0 ? v[3] : v[1], g0 = b0 ? v[0] : v[2], g1 = b0 ? v[1] : v[3];
    const float s0 = MAX ? skt_max_x1(g0, k0) : skt_add_x1(g0, k0);
    const float s1 = MAX ? skt_max_x1(g1, k1) : skt_add_x1(g1, k1);
    const float kk = b1 ? s1 : s0, gg = b1 ? s0 : s1;
    float s = MAX ? skt_max_x2(gg, kk) : skt_add_x2(gg, kk);
    s = MAX ? skt_max_r4(s, s) : skt_add_r4(s, s);
    s = MAX ? skt_max_r8(s, s) : skt_add_r8(s, s);
    if (WIDE) s = skt_swap32(skt_swap16(s, MAX), MAX);
    return s;                                            // the total of row skt_row_of_lane(lane)
}
static_assert(SKT_R == 4, "a wave's partials of a group are one float4");

template <int P, bool SHIFT, int NC = 2, int SMALL = 0>     // SMALL: 0, or the (small) workgroup size 64 / 256
__global__ __launch_bounds__(SMALL ? 256 : (NC == 2 ? SKT_MAXT : SKT_T4)) void sk_table_sweep_kernel(SkTabArgs a) {
    // NC = 4, the wide form (6144 < K <= 8192: the generic configuration's 8000 prototypes): 16 columns per thread, 512 threads,
    // two rows per group (the reductions still carry four values, two of them idle), the scaling vectors K apart in LDS.
    // SMALL (K <= 2048: workgroups of 256 threads, K <= 512: of ONE wave): the scaling vectors take P x 8 T floats, not the
    // 128 KB of the general form, so twelve waves share a CU whatever the workgroup size — with one workgroup per CU a sweep
    // cost 3.6 ms however short the rows were (the 64 < K < 2000 cliff of round 4).
    constexpr int R = NC == 2 ? SKT_R : 2;
    __shared__ __attribute__((aligned(16))) float la2[SMALL ? P * 8 * SMALL : (NC == 2 ? SKT_LA_CAP : SKT_LA_CAP4)];      // [P][8T] (wide: [P][K]), log2 units, -inf past K
    // the waves' partials, [wave][row]; the sums alternate between two buffers from one patch block to the next (a shifted
    // block has ONE barrier: a fast wave's next sums must not land in the slots a slow wave is still reading)
    __shared__ __attribute__((aligned(16))) float redm[16][SKT_R], reds[2][16][SKT_R];
    const int T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int myrow = skt_row_of_lane(lane);
    if (tid < 16 * SKT_R) {
        (&redm[0][0])[tid] = -INFINITY;
        (&reds[0][0][0])[tid] = 0.f;
        (&reds[1][0][0])[tid] = 0.f;
    }
    int par = 0;
    const int view = blockIdx.y, img = blockIdx.z, K = a.K, W8 = NC == 2 ? 8 * T : K;
    const float* S = a.S[view];
    for (int p = 0; p < P; ++p)
        for (int k = tid; k < W8; k += T)
            la2[p * W8 + k] = k < K ? a.la[((long)(p * 2 + view) * a.ngrp + img) * K + k] * SKT_LOG2E : -INFINITY;
    __syncthreads();
    long lo, live;
    skt_window(a, lo, live);
    const long ngroups = (live - lo + R - 1) / R;
    const float sc = a.sc;
    const float* wt = a.wt + view * a.wt_view;
    float* ltab = a.ltab + (long)view * P * a.npix;
    // Two forms, two instantiations.  SHIFT (every sweep after the first): a row's scores are exponentiated relative to
    // sigma = its log2-sum-exp under the PREVIOUS sweep's scaling vector + the largest step any la_k took since — an upper
    // bound of its new maximum that is known before the row is touched, so the maxima pass, its barrier and its LDS round
    // trip drop out.  Nothing overflows (every term <= 1); the largest term is >= 2^-(log2 K + spread of the steps), and the
    // update that made la vouches for a spread below 64 (stats[1]).  The problems it does not vouch for — and all of them in
    // the first sweep, stats == NULL — go through the exact form, which takes the row maxima first; a launch works on its own
    // problems only and returns at once when it has none.
    float dmx[P];
    bool sel[P];
    bool some = false;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const long pr = (long)(p * 2 + view) * a.ngrp + img;
        const int flag = a.stats ? __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(a.stats)[pr * 2 + 1]) : 0;
        dmx[p] = SHIFT ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.stats[pr * 2]))) : 0.f;
        sel[p] = SHIFT ? flag != 0 : flag == 0;
        some = some || sel[p];
    }
    if (!some) return;
    f32x4 u[P][NC];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int c = 0; c < NC; ++c) u[p][c] = zero4();
    f32x4 w[R][NC];
    float gw, gl;
    auto load = [&](long g, f32x4 (&dst)[R][NC], float& gwd, float& gld) {
        gwd = skt_group_weights<P, R>(wt, a.npix, lo, live, g, lane);
        gld = SHIFT ? skt_group_weights<P, R>(ltab, a.npix, lo, live, g, lane) : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = lo + g * R + r;
            const bool ok = row < live;
            const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(S, ok ? row : 0, a.ld, ok ? K : 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) dst[r][c] = bk_load4(rs, 4 * (tid + T * c));
        }
    };
    load(blockIdx.x, w, gw, gl);
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        f32x4 wn[R][NC];                                 // the next group's rows fly under this group's arithmetic
        float gwn, gln;
        load(g + gridDim.x, wn, gwn, gln);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float cw[R];
            bool any = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                cw[r] = skt_lane(gw, p * R + r);
                any = any || cw[r] > 0.f;
            }
            if (!any || !sel[p]) continue;                  // (uniform over the workgroup: every wave holds the same weights)
#if SKT_ABL == 1
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) asm volatile("" ::"v"(w[r][c][0]));
            continue;                                       // lab: the loads alone
#endif
            f32x4 l[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int kc = 4 * (tid + T * c);
                l[c] = (NC == 2 || kc < K) ? *reinterpret_cast<const f32x4*>(la2 + p * W8 + kc)
                                           : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
            f32x4 e[R][NC];
            f32x4 mx4 = SHIFT ? zero4() : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if constexpr (SHIFT) {
#pragma unroll
                for (int r = 0; r < R; ++r) mx4[r] = cw[r] > 0.f ? skt_lane(gl, p * R + r) + dmx[p] : 0.f;
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float mx = -INFINITY;
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            e[r][c][q] = fmaf(w[r][c][q], sc, l[c][q]);
                            mx = skt_vmax(mx, e[r][c][q]);
                        }
                    mx4[r] = mx;
                }
                // the wave's maximum of each row -> its slot (lanes 0 .. 3 hold one row each) ; all waves' -> every lane
                const float m = skt_reduce4<true, true>(mx4, lane);
                if (lane < SKT_R) redm[wid][myrow] = m;
                skt_barrier();
                const float ma = skt_reduce4<true, false>(*reinterpret_cast<const f32x4*>(&redm[lane & 15][0]), lane);
#pragma unroll
                for (int r = 0; r < R; ++r) mx4[r] = skt_lane(ma, skt_lane_of_row(r));
            }
            f32x4 sm4 = zero4();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float sum = 0.f;
                if constexpr (NC == 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        e[r][0][q] = skt_exp2((SHIFT ? fmaf(w[r][0][q], sc, l[0][q]) : e[r][0][q]) - mx4[r]);
                        e[r][1][q] = skt_exp2((SHIFT ? fmaf(w[r][1][q], sc, l[1][q]) : e[r][1][q]) - mx4[r]);
                        sum += e[r][0][q] + e[r][1][q];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            e[r][c][q] = skt_exp2((SHIFT ? fmaf(w[r][c][q], sc, l[c][q]) : e[r][c][q]) - mx4[r]);
                            sum += e[r][c][q];
                        }
                }
                sm4[r] = sum;
            }
            {
                const float t = skt_reduce4<false, true>(sm4, lane);
                if (lane < SKT_R) reds[par][wid][myrow] = t;
                skt_barrier();
                const float ta = skt_reduce4<false, false>(*reinterpret_cast<const f32x4*>(&reds[par][lane & 15][0]), lane);
                par ^= 1;
#pragma unroll
                for (int r = 0; r < R; ++r) sm4[r] = skt_lane(ta, skt_lane_of_row(r));
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (cw[r] > 0.f) {                          // (a row of a mixed group that is not in this patch: nothing of it counts)
                    const float f = cw[r] * __builtin_amdgcn_rcpf(sm4[r]);          // (1 ulp: the sums it scales carry more)
#pragma unroll
                    for (int c = 0; c < NC; ++c) u[p][c] += e[r][c] * f;
                    if (tid == 0) ltab[(long)p * a.npix + lo + g * R + r] = mx4[r] + __builtin_amdgcn_logf(sm4[r]);
                }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int c = 0; c < NC; ++c) w[r][c] = wn[r][c];
        }
        gw = gwn;
        gl = gln;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (!sel[p]) continue;
        const long o = (((long)(p * 2 + view) * a.ngrp + img) * a.nb + blockIdx.x) * K;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int k = 4 * (tid + T * c);
            if (k < K) *reinterpret_cast<f32x4*>(a.part + o + k) = u[p][c];
        }
    }
}

// la_k = (first ? -cmax_k / eps : la_k) + log r_k - log u_k for all problems (cmax as plain floats); blockIdx.x = problem.
// stats (optional) [problem][2]: the largest step any la_k took, in log2 units, and 1 (as an int) if the steps of the problem
// lie within 64 log2 units of each other and are finite — what lets the next sweep run in its shifted form
__global__ __launch_bounds__(256) void sk_table_update_kernel(const float* __restrict__ u, const float* __restrict__ r, int K, int first,
                                                              const float* __restrict__ cmax, float inv_eps, float* __restrict__ la,
                                                              float* __restrict__ stats) {
    __shared__ float red[16];
    const long o0 = (long)blockIdx.x * K;
    float dmax = -INFINITY, dmin = INFINITY;
    for (int k = threadIdx.x; k < K; k += 256) {
        const long o = o0 + k;
        const float d = logf(r ? r[o] : 1.f / (float)K) - logf(u[o]);
        la[o] = (first ? -cmax[o] * inv_eps : la[o]) + d;
        dmax = fmaxf(dmax, d);                  // (a NaN step is dropped by fmaxf / fminf and caught by the spread test below
        dmin = fminf(dmin, d == d ? d : INFINITY);            //  only if every step is one; an infinite one fails the test)
        if (!(d == d)) dmax = INFINITY;
    }
    dmax = gx_block_max(dmax, red);
    dmin = -gx_block_max(-dmin, red);
    if (stats && threadIdx.x == 0) {
        const float spread = (dmax - dmin) * SKT_LOG2E;
        stats[blockIdx.x * 2] = dmax * SKT_LOG2E;
        reinterpret_cast<int*>(stats)[blockIdx.x * 2 + 1] = (!first && spread < 64.f && fabsf(dmax) < 1e30f) ? 1 : 0;
    }
}
// u *= exp((cmax_local - cmax_global) / eps): a rank's first-pass sums re-based on the maxima over all ranks
__global__ void sk_table_rebase_kernel(float* __restrict__ u, const float* __restrict__ cm_local, const float* __restrict__ cm_global,
                                       long n, float inv_eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) u[i] *= expf((cm_local[i] - cm_global[i]) * inv_eps);
}

// chunks of 4 columns per thread: 2 while the scaling vectors of all patches fit LDS 8 T apart at T <= 768, else the wide form
static inline int skt_nc(int npatch, int K) {
    const int t = 64 * ((K + 511) / 512);
#ifdef SKT_FORCE_WIDE
    if ((long)npatch * K <= SKT_LA_CAP4 && K <= 16 * SKT_T4) return 4;          // lab: the wide form wherever it fits
#endif
    return (t <= SKT_MAXT && (long)npatch * 8 * (t < 256 ? 256 : t) <= SKT_LA_CAP) ? 2 : 4;
}
static inline int skt_threads(int K, int nc = 2) {
    int t = 64 * ((K + 256 * nc - 1) / (256 * nc));
    if (nc == 2 && t <= 256) return t == 192 ? 256 : t;          // the small forms: 64 / 128 / 256 threads = up to 512 / 1024 / 2048 columns
    return t < 256 ? 256 : t;
}
// K <= 2048 in the 8-columns-per-thread form: 256-thread workgroups whose scaling vectors leave room for three per CU
static inline int skt_small(int npatch, int K) { return (skt_nc(npatch, K) == 2 && skt_threads(K) <= 256) ? skt_threads(K) : 0; }
static inline int skt_blocks(long npix, int ngrp, int small) {   // workgroups per view and image: one per CU in all (LDS); small: 12 waves per CU
    const long groups = (npix / ngrp + SKT_R - 1) / SKT_R;
    const long nb = ((small ? 128 * 768 / small : 256) + ngrp - 1) / ngrp;
    return (int)(groups < nb ? (groups < 1 ? 1 : groups) : nb);
}

#define SKT_DISPATCH(PV, CALL)                      \
    switch (PV) {                                   \
        case 1: { constexpr int P = 1; CALL; } break; \
        case 2: { constexpr int P = 2; CALL; } break; \
        case 3: { constexpr int P = 3; CALL; } break; \
        case 4: { constexpr int P = 4; CALL; } break; \
        default: { constexpr int P = 5; CALL; } break; \
    }

}  // namespace gx

using namespace gx;

extern "C" {

int gx_sinkhorn_table_ok(int npatch, int K) {
    if (npatch < 1 || npatch > 5 || K <= 64 || K % 4 != 0) return 0;
    if (skt_nc(npatch, K) == 2) return 1;
    return K <= 16 * SKT_T4 && (long)npatch * K <= SKT_LA_CAP4;                 // the wide form: 16 columns per thread
}
// workspace: partial sums and maxima [2 P][nb][K] each | row weights [2][P][npix] | log2-sum-exp of the rows [2][P][npix]
// (the last part carries a sweep's row normalisers to the next one: the calls of one solve share their workspace)
static inline size_t skt_parts_floats(int npatch, long npix, int K, int ngrp) {
    return (size_t)2 * 2 * npatch * ngrp * skt_blocks(npix, ngrp, skt_small(npatch, K)) * K;
}
size_t gx_sinkhorn_table_workspace_bytes(int npatch, long npix, int K, int ngroups) {
    if (ngroups < 1) ngroups = 1;
    return (skt_parts_floats(npatch, npix, K, ngroups) + (size_t)4 * npatch * npix) * sizeof(float);
}
// the row weights of this call into the workspace; returns their view stride (0 = both views share them)
static long skt_weights(const int* inv, long npix, int npatch, const int* live, const float* c_all, long rows, float c_uniform,
                        float* wt, hipStream_t st) {
    const int nv = c_all ? 2 : 1;
    hipLaunchKernelGGL(sk_table_weights_kernel, dim3(gx_cdiv(npix, 256), npatch, nv), dim3(256), 0, st, inv, npix, npatch, live,
                       c_all, rows, c_uniform, wt);
    return c_all ? (long)npatch * npix : 0L;
}

static int skt_check(const float* S0, const float* S1, long ld, int K, const int* inv, long npix, int npatch, float eps,
                     const int* seg, int ngroups) {
    if (!S0 || !S1 || !inv || npix <= 0 || ld < K || !(eps > 0.f) || ngroups < 1 || ngroups > 65535 || (ngroups > 1 && !seg))
        return GX_ERR_BAD_ARG;
    if (!gx_sinkhorn_table_ok(npatch, K)) return GX_ERR_BAD_ARG;
    if ((((uintptr_t)S0 | (uintptr_t)S1) & 15) != 0 || (ld % 4) != 0) return GX_ERR_ALIGN;
    return GX_OK;
}

// cmax [2 P][K], u [2 P][K] (problem index p * 2 + view): column maxima over the problem's rows and
// u_k = sum_n exp((S_nk - cmax_k) / eps) — ONE sweep of the live rows of the two tables
int gx_sinkhorn_table_first(const float* S0, const float* S1, long ld, int K, const int* inv, long npix, int npatch,
                            const int* live, const int* seg, int ngroups, float eps, float* cmax, float* u, float* workspace,
                            size_t workspace_bytes, void* stream) {
    int rc = skt_check(S0, S1, ld, K, inv, npix, npatch, eps, seg, ngroups);
    if (rc) return rc;
    if (!cmax || !u) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_sinkhorn_table_workspace_bytes(npatch, npix, K, ngroups)) return GX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = skt_blocks(npix, ngroups, skt_small(npatch, K)), nprob = 2 * npatch * ngroups;
    SkTabArgs a{};
    a.S[0] = S0; a.S[1] = S1; a.ld = ld; a.K = K; a.npix = npix; a.live = live; a.seg = ngroups > 1 ? seg : nullptr; a.ngrp = ngroups;
    a.sc = SKT_LOG2E / eps; a.part = workspace; a.partm = workspace + (size_t)nprob * nb * K; a.nb = nb;
    float* wt = workspace + skt_parts_floats(npatch, npix, K, ngroups);
    a.wt = wt;
    a.wt_view = skt_weights(inv, npix, npatch, live, nullptr, 0, 1.f, wt, st);          // membership only
    GX_CHECK_LAUNCH();
    if (skt_nc(npatch, K) == 2) {
        SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_first_kernel<P>), dim3(nb, 2, ngroups), dim3(skt_threads(K)), 0, st, a));
    } else {
        SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_first_kernel<P, 4>), dim3(nb, 2, ngroups), dim3(skt_threads(K, 4)), 0, st, a));
    }
    GX_CHECK_LAUNCH();
    // a block that met no row of a problem leaves (-inf, 0): sk_first_from_parts_kernel skips such parts
    launch_first_from_parts(a.partm, a.part, nb, nprob, K, 1.f / eps, cmax, u, st);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// u [2 P][K]: this rank's column sums of one sweep under the scaling vectors la [2 P][K]
int gx_sinkhorn_table_sweep(const float* S0, const float* S1, long ld, int K, const int* inv, long npix, int npatch,
                            const int* live, const int* seg, int ngroups, float eps, const float* la, const float* stats,
                            const float* c_all, long rows, float c_uniform, float* u, float* workspace, size_t workspace_bytes,
                            void* stream) {
    int rc = skt_check(S0, S1, ld, K, inv, npix, npatch, eps, seg, ngroups);
    if (rc) return rc;
    if (!la || !u || (c_all && rows <= 0) || (!c_all && !(c_uniform > 0.f))) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_sinkhorn_table_workspace_bytes(npatch, npix, K, ngroups)) return GX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = skt_blocks(npix, ngroups, skt_small(npatch, K)), nprob = 2 * npatch * ngroups;
    SkTabArgs a{};
    a.S[0] = S0; a.S[1] = S1; a.ld = ld; a.K = K; a.npix = npix; a.live = live; a.seg = ngroups > 1 ? seg : nullptr; a.ngrp = ngroups;
    a.sc = SKT_LOG2E / eps; a.la = la; a.stats = stats; a.part = workspace; a.nb = nb;
    float* wt = workspace + skt_parts_floats(npatch, npix, K, ngroups);
    a.wt = wt;
    a.ltab = wt + (size_t)2 * npatch * npix;
    a.wt_view = skt_weights(inv, npix, npatch, live, c_all, rows, c_uniform, wt, st);
    GX_CHECK_LAUNCH();
    // the exact form for the problems the update does not vouch for (all of them without stats), the shifted form for the others
    const bool wide = skt_nc(npatch, K) == 4;
    const int small = skt_small(npatch, K);
    if (small == 64) {
        SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, false, 2, 64>), dim3(nb, 2, ngroups), dim3(64), 0, st, a));
    } else if (small == 128) {
        SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, false, 2, 128>), dim3(nb, 2, ngroups), dim3(128), 0, st, a));
    } else if (small) {
        SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, false, 2, 256>), dim3(nb, 2, ngroups), dim3(256), 0, st, a));
    } else if (!wide) {
        SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, false>), dim3(nb, 2, ngroups), dim3(skt_threads(K)), 0, st, a));
    } else {
        SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, false, 4>), dim3(nb, 2, ngroups), dim3(skt_threads(K, 4)), 0, st, a));
    }
    GX_CHECK_LAUNCH();
    if (stats) {
        if (small == 64) {
            SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, true, 2, 64>), dim3(nb, 2, ngroups), dim3(64), 0, st, a));
        } else if (small == 128) {
            SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, true, 2, 128>), dim3(nb, 2, ngroups), dim3(128), 0, st, a));
        } else if (small) {
            SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, true, 2, 256>), dim3(nb, 2, ngroups), dim3(256), 0, st, a));
        } else if (!wide) {
            SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, true>), dim3(nb, 2, ngroups), dim3(skt_threads(K)), 0, st, a));
        } else {
            SKT_DISPATCH(npatch, hipLaunchKernelGGL((sk_table_sweep_kernel<P, true, 4>), dim3(nb, 2, ngroups), dim3(skt_threads(K, 4)), 0, st, a));
        }
        GX_CHECK_LAUNCH();
    }
    launch_partial_reduce_batch(workspace, nb, K, nprob, u, st);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// la = (first ? -cmax / eps : la) + log r - log u for the 2 P problems (r_all [2 P][K] or NULL = uniform 1 / K)
int gx_sinkhorn_table_update(const float* u, const float* r_all, int nprob, int K, int first, const float* cmax, float eps,
                             float* la, float* stats, void* stream) {
    if (!u || !la || nprob <= 0 || K <= 0 || (first && !cmax) || !(eps > 0.f)) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(sk_table_update_kernel, dim3(nprob), dim3(256), 0, (hipStream_t)stream, u, r_all, K, first, cmax, 1.f / eps, la,
                       stats);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// sharded first pass: u (relative to this rank's maxima cm_local) re-based on the all-reduced maxima cm_global
int gx_sinkhorn_table_rebase(float* u, const float* cm_local, const float* cm_global, long n, float eps, void* stream) {
    if (!u || !cm_local || !cm_global || n <= 0 || !(eps > 0.f)) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(sk_table_rebase_kernel, dim3(gx_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, u, cm_local, cm_global, n,
                       1.f / eps);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// The whole single-rank solve: la [2 P][K] (problem p * 2 + view).  cmax / u: scratch [2 P][K] each.
int gx_sinkhorn_table(const float* S0, const float* S1, long ld, int K, const int* inv, long npix, int npatch, const int* live,
                      const int* seg, int ngroups, float eps, int niters, const float* r_all, const float* c_all, long rows, float c_uniform, float* la,
                      float* cmax, float* u, float* stats, float* workspace, size_t workspace_bytes, void* stream) {
    if (niters < 1 || !la || !stats) return GX_ERR_BAD_ARG;
    const int nprob = 2 * npatch * (ngroups < 1 ? 1 : ngroups);
    int rc = gx_sinkhorn_table_first(S0, S1, ld, K, inv, npix, npatch, live, seg, ngroups, eps, cmax, u, workspace, workspace_bytes,
                                     stream);
    if (rc) return rc;
    rc = gx_sinkhorn_table_update(u, r_all, nprob, K, 1, cmax, eps, la, nullptr, stream);
    if (rc) return rc;
    for (int it = 1; it < niters; ++it) {      // the first sweep takes the row maxima exactly and leaves the row normalisers behind
        rc = gx_sinkhorn_table_sweep(S0, S1, ld, K, inv, npix, npatch, live, seg, ngroups, eps, la, it == 1 ? nullptr : stats, c_all,
                                     rows, c_uniform, u, workspace, workspace_bytes, stream);
        if (rc) return rc;
        rc = gx_sinkhorn_table_update(u, r_all, nprob, K, 0, nullptr, eps, la, stats, stream);
        if (rc) return rc;
    }
    return GX_OK;
}

}  // extern "C"
