// The finest pyramid level of the low-rank (K <= 16 prototypes) path WITHOUT its projected table.
//
// Reference maths (hfc_with_swav/swav_clustering.py:120-128 upsample + concat, :171-175 projection, normalise,
// prototypes): a pixel q of the finest level with tap vector t(q) (CF channels) and parent p = (y/2, x/2) has
//     z(q) = Wf t(q) + Zc[p]          Wf = Wp[:, finest columns] (D x CF), Zc = the finished table of the level below
// and the loss only ever sees   <Wc_k, z(q)>   and   |z(q)|^2.   Both are CF-wide forms of t(q):
//     <Wc_k, z> = (Wc Wf)_k . t  +  Ac[p][k]                      Ac = Zc Wc^T            (rows/4 x K)
//     |z|^2     = t^T (Wf^T Wf) t  +  t . Y2[p]  +  nc[p]         Y2 = 2 Zc Wf (rows/4 x CF),  nc = |Zc[p]|^2
// so the level is ONE contraction  U = t . Bq^T  with Bq = [ tri(Wf^T Wf) ; Wc Wf ]  ((CF + 16) x CF instead of the
// D = 512-wide projection), whose epilogue reduces a row to its K scores and its 1 / |z| — the (rows x D) table Z of
// the finest level (4.3 GB at the bench batch) is never written, read back for the score tables, or read by the dense
// finish.  tri(M) = 2 strict-lower + diagonal (t^T M t is unchanged for symmetric M): k-tile kt only meets the
// column blocks nb >= kt / 2, 56 % of the MFMA products of the quadratic form.
//
// Backward (autograd of the same lines): with the coefficient grid C of the low-rank path (gx_swav.hip:
// G[q] = C[q][:K] Wc + e[q] z(q), e = C[q][K]) the level's share of every gradient is again CF-wide:
//     dWf  = Wc^T P + Wf S + Zc^T Tt        P = C[:, :K]^T t (K x CF),  S = sum_q e[q] t(q) t(q)^T (CF x CF),
//                                           Tt[p] = sum_{q in p} e[q] t(q)               (rows/4 x CF)
//     Gc   = Tt Wf^T + Cpool[:, :K] Wc + epool .* Zc        (the gradient that reaches the level below)
//     gWc  = P Wf^T + Cpool[:, :K]^T Zc
// gx_fineq_bwd makes S, P, Tt and Cpool in ONE wave-specialised pass over the taps.
#include "gx_mfma_x6.h"
#include <type_traits>

namespace gx {

struct FineqArgs {
    const float* A1; const float* A2;      // the level's taps [rows][K1], [rows][K2] (K2 = 0: one tap)
    long lda1, lda2;
    int K1, K2;
    const unsigned short* Bp;              // gx_split_gemm_weights planes of Bq [CF + 16][CF]
    unsigned Bp_bytes;
    const float* Yx;                       // [rows / 4][FQ_YLD]: Y2 | Ac | nc (gx_fineq_ytable_ld())
    const float* bias;                     // [K] or null
    float* rn_table;                       // [rows]
    float* sc_table;                       // [rows][K]
    int M, K, shift;                       // rows, prototypes, log2(fine resolution)
    int ntile;                             // 128-row tiles
    float a_scale; const float* b_amax;    // f16x3 form (gx_mfma_x6.h): the taps' power-of-two scale, max |Bq| behind the planes
    unsigned* t_amax;                      // optional: max |tap| over the launch (float bits, atomicMax; zeroed by the launcher)
};

#ifndef FQ_ABL
#define FQ_ABL 0                           // lab ablations (tools/lab/fineq_abl.sh): 2 no MFMAs, 11 / 12 the backward kernel's
#endif
#ifndef FQ_PRIO
#define FQ_PRIO 0                          // lab (tools/lab/fineq_flags.sh -DFQ_PRIO=n): wave priority around the MFMA bursts / of the consumer waves
#endif
// The coarse tables travel as ONE table of FQ_YLD floats per parent pixel: columns 0 .. 255 = Y2 = 2 Zc Wf, 256 .. 271 =
// Ac = Zc Wc^T (K of them live), 272 = nc = |Zc|^2 — so that the score block's share is a ninth "column block" of it.
constexpr int FQ_YLD = 288;
constexpr int FQ_A = 8192;                 // one A tile: 128 rows x 16 raw fp32
constexpr int FQ_NSA = 4;                  // A ring: prefetch distance 2, and the tile before the current one stays readable
constexpr int FQ_NSB = 3;                  // B ring: prefetch distance 2
constexpr int FQ_PST = 9 * 1024;           // one plane of a B stage: 9 column blocks x 32 rows x 32 B
constexpr int FQ_B = 3 * FQ_PST;           // 27 KB
constexpr int FQ_Y = 8192;                 // one Y block: 4 row groups x 16 parents x 32 columns
constexpr int FQ_LDS = FQ_NSA * FQ_A + FQ_NSB * FQ_B + 3 * FQ_Y + 512;

// per-wave DMA instruction counts of k-tile kt (compile-time): B = the 1 KB chunks (plane, column block >= kt / 2) dealt to
// 8 waves; Y = one per wave at odd k-tiles (block (kt + 1) / 2 of this / the next tile) and at kt = 12 (the score block)
__host__ __device__ constexpr int fq_nblk(int kt) { return 9 - ((kt & 15) >> 1); }
__host__ __device__ constexpr int fq_nb_dma(int kt, int npl = 3) { return (npl * fq_nblk(kt) + 7) / 8; }
__host__ __device__ constexpr int fq_ny_dma(int kt) { return ((kt & 1) || (kt & 15) == 12) ? 1 : 0; }
// outstanding vector-memory operations allowed at the top of k-tile t: everything issued in k-tile t - 1 (the DMAs of
// k-tile t itself were issued in t - 2, before them, and complete in order)
__host__ __device__ constexpr int fq_vmcnt(int t, int npl = 3) { return fq_ny_dma(t + 15) + fq_nb_dma(t + 1, npl) + 1; }
__device__ __forceinline__ void fq_wait_vm(int n) {        // n is a constant once the k loop is unrolled: one case survives
    switch (n) {
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Workgroup barrier WITHOUT the fences of __syncthreads(): those make hipcc drain every outstanding vector-memory operation
// (s_waitcnt vmcnt(0)) — i.e. the DMAs of the NEXT k-tiles — in front of every barrier.  LDS writes of this wave are
// retired (lgkmcnt(0)), the DMAs it must see are covered by the counted vmcnt wait before it, and the "memory" clobber
// keeps the compiler from moving LDS accesses across.
__device__ __forceinline__ void fq_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Persistent: one 512-thread workgroup per CU walks 128-row tiles; 16 k-tiles per tile, ONE barrier per k-tile, and the
// rings run on across tile boundaries (k-tile t issues the DMAs of k-tile t + 2, possibly of the next tile), so no
// workgroup ever ramps up again after its first tile.  Roles: waves 0-3 ("L") own column blocks 0 .. 5 of their 32 rows,
// waves 4-7 ("R") blocks 6, 7 and the score block — 42 against 46 block-k-tiles, a pair (w, w + 4) shares a SIMD.
// Bq is lower triangular, so column block nb is FINAL after k-tile 2 nb + 1 — the two k-tiles that hold the block's own tap
// columns, both still in the A ring: its owner folds it into the row sums
//     sp[r] += t[row_r][j] * (U[row_r][j] + Y2[parent(row_r)][j])          (accumulator layout: lane = column j, 16 rows)
// right there (t and Y2 from LDS).  Nothing of U is staged or stored and the taps leave HBM exactly once.  History: U parked
// in LDS after the loop with t / Y2 re-read from global 2.4 ms (4.2 GB of L2 misses with little in flight); folding in the
// loop, 4 waves x all 9 blocks, 2 workgroups per CU 1.7 ms (256 VGPRs: the Y2 loads spilled one by one behind vmcnt(0)).
// H2: the f16x3 form — Bq as two fp16 planes scaled on the device (gx_split_gemm_weights_h2), the taps scaled by g.a_scale
// (their range is the convs' contract), three products per block-k-tile; a block's sums are un-scaled where they are folded.
template <bool H2>
__global__ __launch_bounds__(512, 1) void fineq_fwd_kernel(FineqArgs g) {
    constexpr int CF = 256, NKT = 16, NPL = H2 ? 2 : 3;
    const float un = H2 ? 1.f / (g.a_scale * xh_scale_of(*g.b_amax)) : 1.f;
    __shared__ __attribute__((aligned(16))) char lds[FQ_LDS];
    char* const ldsB = lds + FQ_NSA * FQ_A;
    char* const ldsY = ldsB + FQ_NSB * FQ_B;
    float* const xs = reinterpret_cast<float*>(ldsY + 3 * FQ_Y);           // [4 row groups][32 rows]: L's share of the row sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wid & 3;
    const bool isL = wid < 4;
    const int nkt1 = g.K1 >> 4;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    // Tiles to workgroups: consecutive tiles share parents (the two image rows above one parent row are 2 x res / 128 tiles
    // apart at most), and dispatch order deals consecutive workgroups to different XCDs — with tile = blockIdx.x + i gridDim.x
    // every XCD's L2 fetched the parent table for itself (the table was read twice from HBM).  Each XCD takes a contiguous
    // run of tiles instead, its workgroups walking it side by side.
    int tile0 = (int)blockIdx.x, tstride = (int)gridDim.x, tend = g.ntile;
    if ((gridDim.x & 7) == 0 && FQ_ABL != 31) {      // (FQ_ABL 31: the strided assignment, for the A/B)
        const int xcd = blockIdx.x & 7, q = g.ntile >> 3, r = g.ntile & 7;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        tile0 = base + (int)(blockIdx.x >> 3);
        tstride = (int)(gridDim.x >> 3);
        tend = base + q + (xcd < r ? 1 : 0);
    }
    const int my_tiles = tile0 < tend ? (tend - 1 - tile0) / tstride + 1 : 0;
    if (my_tiles == 0) return;

    // ---- DMA sources.  A: unit p = tid holds row p >> 2, k-quad (p & 3) ^ ((p >> 4) & 3) (the layout of gemm_xw_kernel)
    const unsigned arow = (unsigned)(tid >> 2), aq = (unsigned)((tid & 3) ^ ((tid >> 4) & 3));
    const unsigned aoff1 = (arow * (unsigned)g.lda1 + aq * 4u) * 4u, aoff2 = (arow * (unsigned)g.lda2 + aq * 4u) * 4u;
    const __amdgpu_buffer_rsrc_t rs_a1 = make_rsrc(g.A1, 0xFFFFFFF0u);
    const __amdgpu_buffer_rsrc_t rs_a2 = make_rsrc(g.A2 ? g.A2 : g.A1, 0xFFFFFFF0u);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(reinterpret_cast<const float*>(g.Bp), g.Bp_bytes);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(g.Yx, 0xFFFFFFF0u);
    const int s = g.shift, rc = 1 << (s - 1), msk = (1 << s) - 1;
    // first parent of row group rg of tile tm (a row group = 32 consecutive x of one image row, x even: host checks res >= 32)
    auto pbase_of = [&](int tm) -> unsigned {
        const int row0 = tm * 128 + rg * 32;
        return (unsigned)(((row0 >> (2 * s)) * rc + (((row0 >> s) & msk) >> 1)) * rc + ((row0 & msk) >> 1));
    };
    auto issue_a = [&](int tm, int kt, int slot) {
        if (FQ_ABL == 3) return;
        const bool second = kt >= nkt1;
        const unsigned so = (unsigned)tm * 128u * (unsigned)(second ? g.lda2 : g.lda1) * 4u + (unsigned)((second ? kt - nkt1 : kt) * 64);
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (lds_ptr_t)(lds + slot * FQ_A + wid * 1024), 16, (int)aoff2, (int)so, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a1, (lds_ptr_t)(lds + slot * FQ_A + wid * 1024), 16, (int)aoff1, (int)so, 0, 0);
    };
    // B: the 1 KB chunks (plane p, column block nb >= kt / 2) of k-tile kt, chunk c = wave + 8 i (clamped: a few duplicates)
    auto issue_b = [&](int kt, int slot) {
        if (FQ_ABL == 4) return;
        const int nblk = fq_nblk(kt), nch = NPL * nblk, nbmin = 9 - nblk;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= fq_nb_dma(kt, NPL)) continue;
            const int c = min(wid + 8 * i, nch - 1);
            const int p = c / nblk, nb = nbmin + c - p * nblk;
            const unsigned src = (unsigned)(((nb >> 2) * NKT + kt) * NPL + p) * 4096u + (unsigned)(nb & 3) * 1024u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(ldsB + slot * FQ_B + p * FQ_PST + nb * 1024), 16,
                                                     (int)(lane * 16), (int)src, 0, 0);
        }
    };
    // Y: block nb (8 = the Ac | nc columns) of tile tm for this wave's row group: parents 8 (wid >> 2) .. + 7, 128 B each
    const unsigned yoff = ((unsigned)(8 * (wid >> 2) + (lane >> 3)) * FQ_YLD) * 4u + (unsigned)(lane & 7) * 16u;
    auto issue_y = [&](int tm, int nb, int slot) {
        if (FQ_ABL == 5) return;
        const unsigned so = (pbase_of(tm) * FQ_YLD + (unsigned)nb * 32u) * 4u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lds_ptr_t)(ldsY + slot * FQ_Y + rg * 2048 + (wid >> 2) * 1024), 16, (int)yoff,
                                                 (int)so, 0, 0);
    };

    // ---- per-lane constants
    const int r32 = lane & 31, hh = lane >> 5;
    const int afr = rg * 32 + r32, af = (afr >> 2) & 3;                     // A fragment: row r32 of the row group, k = 8 hh ..
    const int au0 = (afr * 4 + ((2 * hh) ^ af)) * 16, au1 = (afr * 4 + ((2 * hh + 1) ^ af)) * 16;
    // the fold of a finished block reads, for the lane's row, the 16 columns its accumulator registers hold: quads hh and
    // 2 + hh of the row in the two k-tiles of the block (A ring, 16-B units swizzled like the fragment reads) and the same
    // columns of the parent's Y2 row
    const int tq0 = afr * 64 + ((hh ^ af) << 4), tq1 = afr * 64 + (((2 + hh) ^ af) << 4);
    const int ybase = rg * 2048 + (r32 >> 1) * 128 + hh * 16;               // Y[parent r32 / 2][4 hh ..] (+ 32 B per group of 8 columns)
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // l*h, h*l, m*m, m*h, h*m, h*h
    f32x4 myb[2];                                                         // bias of the prototypes 4 hh + 8 g2 .. + 3 this lane writes
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * hh + 8 * g2 + q;
            myb[g2][q] = (k < g.K && g.bias) ? g.bias[k] : 0.f;
        }

    // ---- prologue: what k-tiles -2 and -1 of a previous tile would have issued
    int tm = tile0;
    int sb = 0;                                                          // B ring slot of the current k-tile (runtime: 16 % 3 != 0)
    issue_b(0, 0); issue_a(tm, 0, 0);
    issue_y(tm, 0, 0); issue_b(1, 1); issue_a(tm, 1, 1);

    f32x16 acc[6];
    float sp;                                                            // this lane's share of its row's t^T M t + t . Y2
    float tmax = 0.f;                                                    // R waves: max |tap| of the fragments they read (every k-tile)
    for (int it = 0; it < my_tiles; ++it) {
        const int tn = (it + 1 < my_tiles) ? tm + tstride : tm;        // next tile (the last one prefetches itself: unused)
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        sp = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            if (FQ_ABL >= 3 && FQ_ABL <= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else fq_wait_vm(fq_vmcnt(t, NPL));
            if (FQ_ABL != 7) fq_barrier();
            // issue: [Y] B(t + 2) A(t + 2) — Y first: its consumer (two k-tiles on) only waits for the A / B issued after it
            if (t & 1) issue_y(t == 15 ? tn : tm, ((t + 1) >> 1) & 7, ((t + 1) >> 1) & 1);
            if (t == 12) issue_y(tm, 8, 2);
            {
                const int sb2 = sb + 2 >= FQ_NSB ? sb + 2 - FQ_NSB : sb + 2;
                issue_b((t + 2) & 15, sb2);
                issue_a(t + 2 >= NKT ? tn : tm, (t + 2) & 15, (t + 2) & 3);
            }
            const int nb0 = t >> 1;
            const bool active = isL ? (nb0 <= 5) : true;
            if (active) {
                const char* st = lds + (t & 3) * FQ_A;
                const char* Bs = ldsB + sb * FQ_B;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(st + au0);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(st + au1);
                unsigned h[4], m[4], l[4];
                if (!isL)       // (an R wave meets all 16 k-tiles of its 32 rows: the four of them see every tap value of the tile)
                    tmax = fmaxf(fmaxf(fmaxf(tmax, fmaxf(fabsf(x0[0]), fabsf(x0[1]))), fmaxf(fabsf(x0[2]), fabsf(x0[3]))),
                                 fmaxf(fmaxf(fabsf(x1[0]), fabsf(x1[1])), fmaxf(fabsf(x1[2]), fabsf(x1[3]))));
                if constexpr (H2) {
                    split2h(x0[0], x0[1], g.a_scale, h[0], l[0]);
                    split2h(x0[2], x0[3], g.a_scale, h[1], l[1]);
                    split2h(x1[0], x1[1], g.a_scale, h[2], l[2]);
                    split2h(x1[2], x1[3], g.a_scale, h[3], l[3]);
                    m[0] = m[1] = m[2] = m[3] = 0u;
                } else {
                    split3(x0[0], x0[1], h[0], m[0], l[0]);
                    split3(x0[2], x0[3], h[1], m[1], l[1]);
                    split3(x1[0], x1[1], h[2], m[2], l[2]);
                    split3(x1[2], x1[3], h[3], m[3], l[3]);
                }
                // (f16x3: planes 0 = h, 1 = l of both operands — a[1] holds l, a[2] is unused)
                const bf16x8 a[3] = {__builtin_bit_cast(bf16x8, (u32x4){h[0], h[1], h[2], h[3]}),
                                     H2 ? __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]})
                                        : __builtin_bit_cast(bf16x8, (u32x4){m[0], m[1], m[2], m[3]}),
                                     __builtin_bit_cast(bf16x8, (u32x4){l[0], l[1], l[2], l[3]})};
                // this wave's column blocks that k-tile t still meets: L: nb0 .. 5 -> acc[nb]; R: max(nb0, 6) .. 8 -> acc[nb - 6]
                auto run = [&](auto role) {
                    constexpr bool L = decltype(role)::value;
                    constexpr int lo = L ? 0 : 6, hi = L ? 5 : 8;
#pragma unroll
                    for (int gb = lo; gb <= hi; gb += 2) {
                        bf16x8 b[2][3];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int nb = gb + j;
                            if (nb > hi || nb < nb0) continue;
#pragma unroll
                            for (int q = 0; q < NPL; ++q) {
                                if (FQ_ABL == 8 && (gb != lo || j != 0)) { b[j][q] = b[0][q]; continue; }   // lab: one block's fragments for all
                                b[j][q] = *reinterpret_cast<const bf16x8*>(Bs + q * FQ_PST + x_kc_off(nb * 32 + r32, hh));
                            }
                        }
#pragma unroll
                        for (int u = 0; u < (H2 ? 3 : 6); ++u)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const int nb = gb + j;
                                if (nb > hi || nb < nb0 || FQ_ABL == 2) continue;
                                // operands swapped: the accumulator is the block of U TRANSPOSED — lane = the row of t (r32),
                                // register r = column (r & 3) + 8 (r >> 2) + 4 hh of the block — so that a row's fold is lane-local
                                if constexpr (H2) {
                                    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};          // l*h, h*l, h*h
                                    acc[nb - lo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[j][HB[u]]),
                                                                                           __builtin_bit_cast(f16x8, a[HA[u]]), acc[nb - lo], 0, 0, 0);
                                } else {
                                    acc[nb - lo] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][TB[u]], a[TA[u]], acc[nb - lo], 0, 0, 0);
                                }
                            }
                    }
                };
                if (FQ_PRIO) __builtin_amdgcn_s_setprio(FQ_PRIO);
                if (isL) run(std::integral_constant<bool, true>{});
                else run(std::integral_constant<bool, false>{});
                if (FQ_PRIO) __builtin_amdgcn_s_setprio(0);
            }
            if ((t & 1) && FQ_ABL != 6 && (isL ? nb0 <= 5 : (nb0 >= 6))) {                 // block nb0 is final and this wave owns it: fold it
                // sp += sum over the lane's 16 columns j of t[row][j] (U[row][j] + Y2[parent][j]): four 16-byte reads of t
                // (k-tiles t - 1 and t, both still in the A ring) and four of Y2 — the accumulator being U^T, everything
                // else is in this lane's registers.  (Round 3 held U with lane = column: 32 ds_read_b32 per fold and sixteen
                // 32-lane reductions per tile; 1.41 -> 1.22 ms.  FQ_ABL 6 — no fold — is NOT what the fold costs: without it
                // the products of eight of the nine column blocks are dead code and hipcc drops them: 0.70 ms = the kernel
                // without its MFMAs, 0.63.  What is left is that sum itself: 0.63 ms of waits, fragment reads and split +
                // 0.59 ms of MFMAs that one barrier domain per CU runs one after the other.)
                const char* ta = lds + ((t - 1) & 3) * FQ_A;
                const char* tb = lds + (t & 3) * FQ_A;
                const char* ysrc = ldsY + (nb0 & 1) * FQ_Y + ybase;
                const f32x4 tv[4] = {*reinterpret_cast<const f32x4*>(ta + tq0), *reinterpret_cast<const f32x4*>(ta + tq1),
                                     *reinterpret_cast<const f32x4*>(tb + tq0), *reinterpret_cast<const f32x4*>(tb + tq1)};
                const f32x4 yv[4] = {*reinterpret_cast<const f32x4*>(ysrc), *reinterpret_cast<const f32x4*>(ysrc + 32),
                                     *reinterpret_cast<const f32x4*>(ysrc + 64), *reinterpret_cast<const f32x4*>(ysrc + 96)};
                auto fold = [&](const f32x16& u) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) sp = fmaf(tv[r >> 2][r & 3], (H2 ? u[r] * un : u[r]) + yv[r >> 2][r & 3], sp);
                };
                if (isL) fold(acc[nb0 <= 5 ? nb0 : 5]);
                else fold(acc[nb0 >= 6 ? nb0 - 6 : 0]);
                if (isL && nb0 == 5) {                                      // L's share of the row sums is complete
                    const float v = gx_xor32_sum(sp);
                    if (hh == 0) xs[rg * 32 + r32] = v;
                }
            }
            sb = sb + 1 >= FQ_NSB ? 0 : sb + 1;
        }
        if (!isL) {
            // ---- rows: |z|^2 = L's share + R's + nc[parent]; scores from the score block + Ac[parent]
            // (the score block of Y arrived at k-tile 12 + 2; xs was written at k-tile 11: four barriers ago)
            // lane = row r32 (both halves of the wave): the half hh holds prototypes 4 hh + 8 g2 .. + 3 in registers 4 g2 .. + 3
            const char* yp = ldsY + 2 * FQ_Y + rg * 2048 + (r32 >> 1) * 128;
            const long row = (long)tm * 128 + rg * 32 + r32;
            const float nsq = gx_xor32_sum(sp) + xs[rg * 32 + r32] + *reinterpret_cast<const float*>(yp + 16 * 4);
            const float rn = 1.f / fmaxf(sqrtf(fmaxf(nsq, 0.f)), 1e-12f);
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int k0 = 4 * hh + 8 * g2;
                if (k0 >= g.K) continue;
                const f32x4 ac = *reinterpret_cast<const f32x4*>(yp + k0 * 4);
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = fmaf((H2 ? acc[2][4 * g2 + q] * un : acc[2][4 * g2 + q]) + ac[q], rn, myb[g2][q]);
                float* dst = g.sc_table + row * g.K + k0;
                if (k0 + 3 < g.K && (g.K & 3) == 0) *reinterpret_cast<f32x4*>(dst) = o;
                else
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (k0 + q < g.K) dst[q] = o[q];
            }
            if (hh == 0) g.rn_table[row] = rn;
        }
        tm = tn;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // the last tile's look-ahead DMAs still target this LDS
    __syncthreads();
    if (g.t_amax && !isL) {                  // the taps' true maximum: the bound fineq_bwd_kernel<true> scales its images by
        tmax = gx_wave_max(tmax);
        if (lane == 0 && __float_as_uint(tmax) > *reinterpret_cast<volatile unsigned*>(g.t_amax)) atomicMax(g.t_amax, __float_as_uint(tmax));
    }
}

// Bq [CF + 16][CF] (row stride CF) for fineq_fwd_kernel: rows j < CF = the lower-triangular form of the symmetric
// M = Wf^T Wf (2 M[j][i] for i < j, M[j][j], 0 above: sum_j t_j (Bq t)_j = t^T M t), rows CF .. CF + K - 1 = Wcf = Wc Wf,
// the rest zero.
__global__ void fineq_pack_kernel(const float* __restrict__ Mg, const float* __restrict__ Wcf, int CF, int K,
                                  float* __restrict__ Bq) {
    const int total = (CF + 16) * CF;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int j = i / CF, c = i - j * CF;
        float v = 0.f;
        if (j < CF) {
            const float mv = Mg[(long)j * CF + c];
            v = c < j ? 2.f * mv : (c == j ? mv : 0.f);
        } else if (j - CF < K) {
            v = Wcf[(long)(j - CF) * CF + c];
        }
        Bq[i] = v;
    }
}

// ---- the backward of the finest level in ONE pass over its taps -------------------------------------------------------------
// S = sum_q e[q] t(q) t(q)^T (CF x CF, symmetric), P = C[:, :K]^T t (K x CF), Tt[p] = sum over the 2x2 children of e[q] t(q),
// Cpool[p] = the children's coefficient sums.  A wave-specialised TN contraction (the shape of gemm_x6ws_kernel): a
// workgroup owns a run of k-tiles, a k-tile = one 2 x 8 pixel block (4 parents x 4 children), so that a parent's four
// children meet in one producer wave:
//   producer wave w (threads 512 + 64 w ..): lanes 0-31 hold the pixel pair (y, x0 + 2 w), (y, x0 + 2 w + 1), lanes 32-63
//     the pair below it — the same parent; a thread owns tap columns 4 (l & 31) .. + 3 (+ 128).  It scales its taps by e,
//     splits t and e t into bf16 planes (B and A images of the k-tile), sums e t over its pair, adds the partner lane's
//     (lane ^ 32) and writes Tt; lanes l & 31 < stride do the same for the coefficients (Cpool) and put C[q][:K] into rows
//     256 .. 271 (columns of its [k][m] image) of the A operand, so that P comes out of the same MFMA loop;
//   consumer waves 0-7: the ten 64 x 64 blocks on or above the diagonal of S + two 32 x 128 blocks of P = 48 MFMA tiles,
//     six per wave, two consumer waves per SIMD (each hides the other's LDS latency) — 75 % of the products of the full
//     Gram matrix, and none of a plain TN GEMM's extra passes over the taps (round 3's first cut, a k-scaled TN GEMM + a
//     separate pool pass, read them 2.5 times: 2.6 ms against 1.3).
// Each workgroup leaves its partial [272][256] in the slab; fineq_bwd_reduce_kernel sums the slabs in a fixed order and
// mirrors the upper blocks of S.
// LDS images of a k-tile, one per operand and plane: plain [16 k rows][m columns] bf16, exactly as the m-contiguous taps
// arrive, read back TRANSPOSED by ds_read_b64_tr_b16 (guide T10): per 16-lane group a block of 4 k rows x 16 m columns, lane
// i receiving column i — two of them are one 32x32x16 operand fragment (m = lane & 31, k = 8 (lane >> 5) .. + 7).  The
// k-pair-packed image they replace needed four ds_read_b32 per fragment, at half the LDS rate (128 B/clk/CU, and far less at
// two waves per SIMD: MI355X_MICROARCH.md, LDS) — it was what the producers and consumers of this kernel fought over.
// Row stride 576 B = 64 (mod 256): the four rows of a block and the two blocks of a half-wave cover all 64 banks once.
// A: 256 tap columns + the 16 coefficient columns + 16 zero columns = 288 columns = 576 B; B: 256 columns + 64 B of padding.
constexpr int FB_RS = 576, FB_PL = 16 * FB_RS, FB_IMG = 3 * FB_PL;        // row stride, one plane, one operand image (27 KB)
constexpr int FB_SLOT = 2 * FB_IMG;
struct FineqBwdArgs {
    const float* C; int ls;                // coefficient grid [rows][ls]
    const float* T1; const float* T2;      // tap columns 0..127 and 128..255 (T2 = T1 + 128 for a single 256-channel tap)
    long ld1, ld2;
    int K, shift;                          // prototypes, log2(fine resolution)
    long nkt;                              // k-tiles in all = rows / 16
    int kt_per;                            // k-tiles per workgroup
    float* slab;                           // [gridDim][272][256]
    float* Tt;                             // [rows / 4][256]
    float* Cpool;                          // [rows / 4][ls]
    unsigned* tt_amax;                     // optional: max |Tt| over the launch (float bits, atomicMax; zeroed by the launcher)
    // fineq_bwd_kernel<true> (f16x3): device floats max |tap| (gx_fineq_fwd's t_amax) and {max |C[:, :K]|, max |C[:, K]|}
    // (gx_lowrank_coeff_amax): the images are two fp16 planes of t sB, e t sA, c sC with the powers of two these bounds give
    const float* t_amax; const float* c_amax;
};

// Consumer wave W of fineq_bwd_kernel.  Tasks (row block of the A image, column block of the B image), four MFMA tiles each:
//   W = 0: (0,0) (0,64) (0,128)    W = 1: (0,192) (64,192) (128,192)    W = 2: (64,64) (64,128) (128,128)   — 64 x 64 blocks of S
//   W = 3: (192,192), then the coefficient rows 256.. against columns 128..255 and 0..127 (32 x 128 each: P)
// The twelve MFMA tiles of a k-tile are walked in an order in which only ONE operand fragment changes from tile to tile
// (FB_GA / FB_GB); the changing fragment of tile n + 1 is read from LDS before tile n's six products issue, so its
// latency hides under them.  Each list is cut in two: eight consumer waves of six tiles (96 accumulators + 36 fragment
// registers; with sched_barrier pinning the order nothing spills — hoisted freely, the four-wave form spilled ~90
// registers per k-tile: 2.4 ms instead of 1.6).
constexpr int FB_TA0[4][3] = {{0, 0, 0}, {0, 64, 128}, {64, 64, 128}, {192, 256, 256}};
constexpr int FB_TB0[4][3] = {{0, 64, 128}, {192, 192, 192}, {64, 128, 128}, {192, 0, 128}};
constexpr int FB_GA[4][12] = {{0, 32, 32, 0, 0, 32, 32, 0, 0, 32, 32, 0},
                              {0, 32, 32, 0, 64, 96, 96, 64, 128, 160, 160, 128},
                              {64, 96, 96, 64, 64, 96, 96, 64, 128, 160, 160, 128},
                              {192, 224, 224, 192, 256, 256, 256, 256, 256, 256, 256, 256}};
constexpr int FB_GB[4][12] = {{0, 0, 32, 32, 64, 64, 96, 96, 128, 128, 160, 160},
                              {192, 192, 224, 224, 224, 224, 192, 192, 192, 192, 224, 224},
                              {64, 64, 96, 96, 128, 128, 160, 160, 160, 160, 128, 128},
                              {192, 192, 224, 224, 224, 192, 160, 128, 96, 64, 32, 0}};
struct FbFrag { bf16x8 p[3]; };
// fragment (32 rows starting at column `col0` of the image, plane `pl`) for lane `lane`: toff = this lane's byte offset
// inside a 4 x 16 block = (8 h + q) * FB_RS + (16 (g & 1) + 4 p) * 2 with g = lane >> 4, h = g >> 1, q = (lane & 15) >> 2, p = lane & 3
__device__ __forceinline__ bf16x8 fb_frag(const char* img, int pl, int col0, int toff) {
    typedef __attribute__((address_space(3))) s16x4* lp;
    const char* a = img + pl * FB_PL + col0 * 2 + toff;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a + 4 * FB_RS));
    return __builtin_bit_cast(bf16x8, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}
// x (lanes 0-31) + x (lanes 32-63) in every lane: v_permlane32_swap (swaps a's upper half-wave with b's lower one, in place),
// one VALU instruction instead of a ds_bpermute.  Inline asm: hipcc 7.2 returns the FIRST register for both elements of
// __builtin_amdgcn_permlane32_swap's result (tools/lab history: r[0] + r[1] came out as 2 r[0]).
__device__ __forceinline__ float fb_half_sum(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return a + b;
}
template <int W8, bool H2>
__device__ __forceinline__ void fineq_bwd_consume(const FineqBwdArgs& g, char* lds, int nkt, int lane, float un_s, float un_p) {
    constexpr int W = W8 >> 1, N0 = 6 * (W8 & 1);                     // this wave: tiles N0 .. N0 + 5 of list W
    constexpr int AIMG = FB_IMG;
    const int toff = (8 * (lane >> 5) + ((lane & 15) >> 2)) * FB_RS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int r32 = lane & 31, hh = lane >> 5;
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // l*h, h*l, m*m, m*h, h*m, h*h
    constexpr int NPL = H2 ? 2 : 3;                                   // f16x3: planes 0 = h, 1 = l
    constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};               // l*h, h*l, h*h
    auto load_a = [&](const char* As, int row) {
        FbFrag f;
#pragma unroll
        for (int q = 0; q < NPL; ++q) f.p[q] = fb_frag(As, q, row, toff);
        return f;
    };
    auto load_b = [&](const char* Bs, int row) {
        FbFrag f;
#pragma unroll
        for (int q = 0; q < NPL; ++q) f.p[q] = fb_frag(Bs, q, row, toff);
        return f;
    };
    __syncthreads();                                                 // barrier 0
    if (FQ_PRIO) __builtin_amdgcn_s_setprio(FQ_PRIO);
    for (int t = 0; t < nkt; ++t) {
        const char* As = lds + (t & 1) * FB_SLOT;
        const char* Bs = As + AIMG;
        FbFrag a = load_a(As, FB_GA[W][N0]), b = load_b(Bs, FB_GB[W][N0]), nx;
#pragma unroll
        for (int n = N0; n < N0 + 6; ++n) {
            const bool last = n == N0 + 5;
            const bool a_changes = !last && FB_GA[W][n + 1 > 11 ? 11 : n + 1] != FB_GA[W][n];
            if (!last) nx = a_changes ? load_a(As, FB_GA[W][n + 1 > 11 ? 11 : n + 1]) : load_b(Bs, FB_GB[W][n + 1 > 11 ? 11 : n + 1]);
            const int ai = n - N0;
            if constexpr (H2) {
#pragma unroll
                for (int u = 0; u < (FQ_ABL == 11 ? 1 : 3); ++u)
                    acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.p[HA[u]]),
                                                                     __builtin_bit_cast(f16x8, b.p[HB[u]]), acc[ai], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < (FQ_ABL == 11 ? 1 : 6); ++u)
                    acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[TA[u]], b.p[TB[u]], acc[ai], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                        // the order above is the schedule
            if (!last) { if (a_changes) a = nx; else b = nx; }
        }
        __syncthreads();                                             // barrier t + 1
    }
    float* out = g.slab + (long)blockIdx.x * (272 * 256);
#pragma unroll
    for (int n = N0; n < N0 + 6; ++n) {
        const int ai = n - N0;
        const int col = FB_GB[W][n] + r32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = FB_GA[W][n] + (r & 3) + 8 * (r >> 2) + 4 * hh;
            // (f16x3: the scales leave — powers of two; the coefficient rows carry their own)
            if (row < 272) out[row * 256 + col] = H2 ? acc[ai][r] * (FB_GA[W][n] >= 256 ? un_p : un_s) : acc[ai][r];
        }
    }
}

// H2: the f16x3 form — the producers scale t, e t and the coefficient columns by the powers of two of their device-side
// bounds and write two fp16 planes per image, the consumers issue three products per tile and un-scale what they store.
template <bool H2>
__global__ __launch_bounds__(768, 1) void fineq_bwd_kernel(FineqBwdArgs g) {
    __shared__ __attribute__((aligned(16))) char lds[2 * FB_SLOT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long kt0 = (long)blockIdx.x * g.kt_per;
    const int nkt = (int)max(0L, min((long)g.kt_per, g.nkt - kt0));
    char* slot0 = lds;
    char* slot1 = lds + FB_SLOT;
    constexpr int AIMG = FB_IMG;
    float sA = 1.f, sB = 1.f, sC = 1.f;
    if constexpr (H2) {
        const float tb = *g.t_amax;
        sB = xh_scale_of(tb); sA = xh_scale_of(tb * g.c_amax[1]); sC = xh_scale_of(g.c_amax[0]);
    }
    // columns 272 .. 287 of the A image are never written by the producers: zero them once (both slots, all planes, 16 k rows
    // x 32 B = 8 words each)
    for (int i = tid; i < 2 * 3 * 16 * 8; i += 768) {
        const int sl = i / (3 * 16 * 8), r = i % (3 * 16 * 8);
        const int pl = r / (16 * 8), k = (r / 8) & 15, w = r & 7;
        *reinterpret_cast<unsigned*>(lds + sl * FB_SLOT + pl * FB_PL + k * FB_RS + 272 * 2 + w * 4) = 0u;
    }

    if (wid >= 8) {
        // ------------------------------------------------------------------ producers
        const int p = tid - 512, pw = p >> 6, half = (p >> 5) & 1, c4 = (p & 31) * 4, mm = p & 31;
        const int s = g.shift, res = 1 << s, rc = res >> 1;
        const int xbs = s - 3;                                              // x-blocks per image row = res / 8
        struct Tile { f32x4 v[4]; float c0, c1, e0, e1; long q0, par; bool live; };
        float ttm = 0.f;                                                   // max |Tt| of what this thread wrote
        auto load_tile = [&](int t, Tile& T) {
            T.live = t < nkt;
            const long bid = kt0 + min(t, max(nkt - 1, 0));
            const long per_img = (long)1 << (2 * s - 4);
            const long img = bid >> (2 * s - 4), rem = bid & (per_img - 1);
            const int yp = (int)(rem >> xbs), xb = (int)(rem & ((1 << xbs) - 1));
            T.q0 = (img * res + 2 * yp + half) * res + 8 * xb + 2 * pw;    // the pair q0, q0 + 1
            T.par = (img * rc + yp) * rc + 4 * xb + pw;
            if (T.live) {
                T.v[0] = *reinterpret_cast<const f32x4*>(g.T1 + T.q0 * g.ld1 + c4);
                T.v[1] = *reinterpret_cast<const f32x4*>(g.T1 + (T.q0 + 1) * g.ld1 + c4);
                T.v[2] = *reinterpret_cast<const f32x4*>(g.T2 + T.q0 * g.ld2 + c4);
                T.v[3] = *reinterpret_cast<const f32x4*>(g.T2 + (T.q0 + 1) * g.ld2 + c4);
                T.c0 = mm < g.ls ? g.C[T.q0 * g.ls + mm] : 0.f;
                T.c1 = mm < g.ls ? g.C[(T.q0 + 1) * g.ls + mm] : 0.f;
                T.e0 = g.C[T.q0 * g.ls + g.K];                           // one address per half-wave: a broadcast load
                T.e1 = g.C[(T.q0 + 1) * g.ls + g.K];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) T.v[i] = zero4();
                T.c0 = T.c1 = T.e0 = T.e1 = 0.f;
            }
        };
        auto stage = [&](char* slot, const Tile& T) {
            if (FQ_ABL == 12) return;
            const int kp = p >> 5;                                          // k-pair of this thread: pixels q0, q0 + 1
            const float e0 = T.e0, e1 = T.e1;
            char* A = slot;
            char* B = slot + AIMG;
            // k rows 2 kp (pixel q0) and 2 kp + 1 (pixel q0 + 1) of the images; four columns = 8 B per row and plane
            auto put = [&](char* img, int col, const f32x4& v0, const f32x4& v1, float scale) {
                unsigned h[4], m[4], l[4];
                char* d = img + (2 * kp) * FB_RS + col * 2;
                if constexpr (H2) {
                    split2h(v0[0], v0[1], scale, h[0], l[0]);
                    split2h(v0[2], v0[3], scale, h[1], l[1]);
                    split2h(v1[0], v1[1], scale, h[2], l[2]);
                    split2h(v1[2], v1[3], scale, h[3], l[3]);
                    *reinterpret_cast<u32x2*>(d) = (u32x2){h[0], h[1]};
                    *reinterpret_cast<u32x2*>(d + FB_PL) = (u32x2){l[0], l[1]};
                    *reinterpret_cast<u32x2*>(d + FB_RS) = (u32x2){h[2], h[3]};
                    *reinterpret_cast<u32x2*>(d + FB_RS + FB_PL) = (u32x2){l[2], l[3]};
                    return;
                }
                split3(v0[0], v0[1], h[0], m[0], l[0]);
                split3(v0[2], v0[3], h[1], m[1], l[1]);
                split3(v1[0], v1[1], h[2], m[2], l[2]);
                split3(v1[2], v1[3], h[3], m[3], l[3]);
                *reinterpret_cast<u32x2*>(d) = (u32x2){h[0], h[1]};
                *reinterpret_cast<u32x2*>(d + FB_PL) = (u32x2){m[0], m[1]};
                *reinterpret_cast<u32x2*>(d + 2 * FB_PL) = (u32x2){l[0], l[1]};
                *reinterpret_cast<u32x2*>(d + FB_RS) = (u32x2){h[2], h[3]};
                *reinterpret_cast<u32x2*>(d + FB_RS + FB_PL) = (u32x2){m[2], m[3]};
                *reinterpret_cast<u32x2*>(d + FB_RS + 2 * FB_PL) = (u32x2){l[2], l[3]};
            };
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 t0 = T.v[2 * j], t1 = T.v[2 * j + 1];
                const f32x4 s0 = e0 * t0, s1 = e1 * t1;
                put(B, c4 + 128 * j, t0, t1, sB);
                put(A, c4 + 128 * j, s0, s1, sA);
                // Tt: this pair + the pair below / above it (the partner lane holds the same columns of the same parent)
                f32x4 tt = s0 + s1;
#pragma unroll
                for (int e = 0; e < 4; ++e) tt[e] = fb_half_sum(tt[e]);
                if (half == 0 && T.live) *reinterpret_cast<f32x4*>(g.Tt + T.par * 256 + c4 + 128 * j) = tt;
                ttm = fmaxf(fmaxf(ttm, fmaxf(fabsf(tt[0]), fabsf(tt[1]))), fmaxf(fabsf(tt[2]), fabsf(tt[3])));
            }
            if (mm < 16) {                                                   // coefficient columns 256 .. 271 of the A image
                const bool kok = mm < g.K;
                unsigned h, m, l;                                            // low half: pixel q0's, high half: pixel q0 + 1's
                char* d = A + (2 * kp) * FB_RS + (256 + mm) * 2;
                if constexpr (H2) {
                    split2h(kok ? T.c0 : 0.f, kok ? T.c1 : 0.f, sC, h, l);
                    *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h & 0xffffu);
                    *reinterpret_cast<unsigned short*>(d + FB_PL) = (unsigned short)(l & 0xffffu);
                    *reinterpret_cast<unsigned short*>(d + FB_RS) = (unsigned short)(h >> 16);
                    *reinterpret_cast<unsigned short*>(d + FB_RS + FB_PL) = (unsigned short)(l >> 16);
                } else {
                split3(kok ? T.c0 : 0.f, kok ? T.c1 : 0.f, h, m, l);
                *reinterpret_cast<unsigned short*>(d) = (unsigned short)(h & 0xffffu);
                *reinterpret_cast<unsigned short*>(d + FB_PL) = (unsigned short)(m & 0xffffu);
                *reinterpret_cast<unsigned short*>(d + 2 * FB_PL) = (unsigned short)(l & 0xffffu);
                *reinterpret_cast<unsigned short*>(d + FB_RS) = (unsigned short)(h >> 16);
                *reinterpret_cast<unsigned short*>(d + FB_RS + FB_PL) = (unsigned short)(m >> 16);
                *reinterpret_cast<unsigned short*>(d + FB_RS + 2 * FB_PL) = (unsigned short)(l >> 16);
                }
            }
            const float cp = fb_half_sum(T.c0 + T.c1);
            if (half == 0 && mm < g.ls && T.live) g.Cpool[T.par * g.ls + mm] = cp;
        };
        // Four k-tiles (64 KB per workgroup = per CU) stay in flight: the taps stream from HBM exactly once and nothing else
        // hides their latency (two tiles in flight ran the kernel at 2.4 ms: 1.1 TB/s)
        Tile T0, T1, T2, T3;
        load_tile(0, T0);
        load_tile(1, T1);
        load_tile(2, T2);
        load_tile(3, T3);
        __syncthreads();                                             // the zero rows are in place
        if (nkt > 0) stage(slot0, T0);
        load_tile(4, T0);
        __syncthreads();                                             // barrier 0: tile 0 visible
        for (int t = 0; t < nkt; t += 4) {
            if (t + 1 < nkt) stage(slot1, T1);
            load_tile(t + 5, T1);
            __syncthreads();                                         // barrier t + 1
            if (t + 1 >= nkt) break;
            if (t + 2 < nkt) stage(slot0, T2);
            load_tile(t + 6, T2);
            __syncthreads();                                         // barrier t + 2
            if (t + 2 >= nkt) break;
            if (t + 3 < nkt) stage(slot1, T3);
            load_tile(t + 7, T3);
            __syncthreads();                                         // barrier t + 3
            if (t + 3 >= nkt) break;
            if (t + 4 < nkt) stage(slot0, T0);
            load_tile(t + 8, T0);
            __syncthreads();                                         // barrier t + 4
        }
        if (g.tt_amax) {                     // the bound the f16x3 GEMMs on Tt scale it by (gx_gemm_f32_nt_planes_finish, Zc^T Tt)
            ttm = gx_wave_max(ttm);
            if (lane == 0 && __float_as_uint(ttm) > *reinterpret_cast<volatile unsigned*>(g.tt_amax)) atomicMax(g.tt_amax, __float_as_uint(ttm));
        }
        return;
    }
    // ---------------------------------------------------------------------- consumers
    __syncthreads();                                                 // (zero rows)
    const float un_s = 1.f / (sA * sB), un_p = 1.f / (sC * sB);
    switch (wid) {
        case 0: fineq_bwd_consume<0, H2>(g, lds, nkt, lane, un_s, un_p); break;
        case 1: fineq_bwd_consume<1, H2>(g, lds, nkt, lane, un_s, un_p); break;
        case 2: fineq_bwd_consume<2, H2>(g, lds, nkt, lane, un_s, un_p); break;
        case 3: fineq_bwd_consume<3, H2>(g, lds, nkt, lane, un_s, un_p); break;
        case 4: fineq_bwd_consume<4, H2>(g, lds, nkt, lane, un_s, un_p); break;
        case 5: fineq_bwd_consume<5, H2>(g, lds, nkt, lane, un_s, un_p); break;
        case 6: fineq_bwd_consume<6, H2>(g, lds, nkt, lane, un_s, un_p); break;
        default: fineq_bwd_consume<7, H2>(g, lds, nkt, lane, un_s, un_p); break;
    }
}

// out[0] = max |C[q][k]| over k < K, out[1] = max |C[q][K]| over the coefficient grid C [rows][ls] (float bits, atomicMax; the
// launcher zeroes them): the bounds fineq_bwd_kernel<true> scales its coefficient columns and its e t image by
__global__ void lowrank_coeff_amax_kernel(const float* __restrict__ C, long rows, int ls, int K, unsigned* __restrict__ out) {
    __shared__ float red[16];
    // flat 16-byte units: unit u holds columns 4 (u % (ls / 4)) .. + 3 of row u / (ls / 4); four units in flight per thread
    const int upr = ls >> 2;
    const long nu = rows * upr, stride = (long)gridDim.x * blockDim.x;
    float mc = 0.f, me = 0.f;
    auto take = [&](long u, const f32x4& v) {
        const int c0 = (int)(u % upr) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = fabsf(v[e]);
            if (c0 + e < K) mc = fmaxf(mc, a);
            else if (c0 + e == K) me = fmaxf(me, a);
        }
    };
    long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; u + 3 * stride < nu; u += 4 * stride) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(C + u * 4), v1 = *reinterpret_cast<const f32x4*>(C + (u + stride) * 4);
        const f32x4 v2 = *reinterpret_cast<const f32x4*>(C + (u + 2 * stride) * 4), v3 = *reinterpret_cast<const f32x4*>(C + (u + 3 * stride) * 4);
        take(u, v0); take(u + stride, v1); take(u + 2 * stride, v2); take(u + 3 * stride, v3);
    }
    for (; u < nu; u += stride) take(u, *reinterpret_cast<const f32x4*>(C + u * 4));
    mc = gx_block_max(mc, red);
    me = gx_block_max(me, red);
    if (threadIdx.x == 0) {      // (read first: a thousand unconditional atomics on one word serialise in L2)
        if (__float_as_uint(mc) > *reinterpret_cast<volatile unsigned*>(out)) atomicMax(out, __float_as_uint(mc));
        if (__float_as_uint(me) > *reinterpret_cast<volatile unsigned*>(out + 1)) atomicMax(out + 1, __float_as_uint(me));
    }
}

// S [256][256] (row stride lds_) and P [K][256] from the workgroups' partials: fixed order; the blocks of S below the
// diagonal (64 x 64 granularity) are the mirror images of the ones above it
__global__ void fineq_bwd_reduce_kernel(const float* __restrict__ slab, int nslab, int K, float* __restrict__ S, float* __restrict__ P) {
#pragma clang fp contract(off)
    const int total = (256 + K) * 256;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int row = i >> 8, col = i & 255;
        int sr = row, sc = col;
        if (row < 256 && (row >> 6) > (col >> 6)) { sr = col; sc = row; }
        const float* src = slab + sr * 256 + sc;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = 0;
        for (; b + 3 < nslab; b += 4) {
            s0 += src[(long)b * (272 * 256)]; s1 += src[(long)(b + 1) * (272 * 256)];
            s2 += src[(long)(b + 2) * (272 * 256)]; s3 += src[(long)(b + 3) * (272 * 256)];
        }
        for (; b < nslab; ++b) s0 += src[(long)b * (272 * 256)];
        const float v = (s0 + s1) + (s2 + s3);
        if (row < 256) S[i] = v;
        else P[(row - 256) * 256 + col] = v;
    }
}

}  // namespace gx

extern "C" {

// the taps a finest level must have for this path: CF = K1 + K2 = 256 channels as one tap or as two of 128 (the Gram GEMM
// of the backward takes its 256-row tile from at most two sources split at 128), K <= 16 prototypes
int gx_fineq_ok(int K1, int K2, int K) {
    return K1 + K2 == 256 && (K2 == 0 || K1 == 128) && K >= 1 && K <= 16;
}

int gx_fineq_pack_weights(const float* Mg, const float* Wcf, int CF, int K, float* Bq, void* stream) {
    if (!Mg || !Wcf || !Bq || CF <= 0 || K < 1 || K > 16) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(gx::fineq_pack_kernel, dim3(gx_cdiv((long)(CF + 16) * CF, 256)), dim3(256), 0, (hipStream_t)stream, Mg, Wcf,
                       CF, K, Bq);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_fineq_ytable_ld(void) { return gx::FQ_YLD; }
static int fineq_num_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}
int gx_fineq_fwd(int rows, int fine_res, int K1, const float* t1, long ld1, int K2, const float* t2, long ld2,
                 const unsigned short* bq_planes, const float* ytable, const float* bias, int K, float* rn_table,
                 float* sc_table, float a_range, float* t_amax, void* stream) {
    if (!t1 || !bq_planes || !ytable || !rn_table || !sc_table || rows <= 0 || (K2 > 0 && !t2) || a_range < 0.f) return GX_ERR_BAD_ARG;
    if (!gx_fineq_ok(K1, K2, K)) return GX_ERR_BAD_ARG;
    if (gx_get_gemm_mode() != 1) return GX_ERR_BAD_ARG;                   // a bf16x6 kernel
    // >= 32: a wave's 32 rows are consecutive pixels of one image row (fineq_fwd_kernel's parent arithmetic); rows is then
    // a multiple of 1024: whole 128-row tiles
    if (fine_res < 32 || (fine_res & (fine_res - 1)) || (long)rows % ((long)fine_res * fine_res)) return GX_ERR_BAD_ARG;
    if ((ld1 & 3) || (K2 > 0 && (ld2 & 3)) || ld1 < K1 || (K2 > 0 && ld2 < K2) ||
        (long)rows * ld1 * 4 >= 0xFFFFFFF0L || (long)rows * ld2 * 4 >= 0xFFFFFFF0L ||           // 32-bit buffer offsets
        (long)(rows / 4) * gx::FQ_YLD * 4 >= 0xFFFFFFF0L ||
        (((uintptr_t)t1 | (uintptr_t)t2 | (uintptr_t)bq_planes | (uintptr_t)ytable) & 15))
        return GX_ERR_ALIGN;
    const int CF = K1 + K2;
    gx::FineqArgs g;
    g.A1 = t1; g.A2 = K2 > 0 ? t2 : nullptr; g.lda1 = ld1; g.lda2 = K2 > 0 ? ld2 : ld1; g.K1 = K1; g.K2 = K2;
    g.Bp = bq_planes; g.Bp_bytes = (unsigned)gx_gemm_weight_planes_bytes(CF + 16, CF);
    g.Yx = ytable; g.bias = bias; g.rn_table = rn_table; g.sc_table = sc_table;
    g.M = rows; g.K = K; g.shift = 0;
    while ((1 << g.shift) < fine_res) ++g.shift;
    g.ntile = rows / 128;
    const int nwg = g.ntile < fineq_num_cus() ? g.ntile : fineq_num_cus();
    // a_range > 0: bq_planes are gx_split_gemm_weights_h2's (two fp16 planes + the max-|Bq| trailer), the taps bounded by a_range
    g.a_scale = a_range > 0.f ? gx::xh_scale_of(a_range) : 1.f;
    g.b_amax = reinterpret_cast<const float*>(reinterpret_cast<const char*>(bq_planes) + gx_gemm_weight_planes_bytes(CF + 16, CF) - 16);
    g.t_amax = reinterpret_cast<unsigned*>(t_amax);
    if (t_amax && hipMemsetAsync(t_amax, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return GX_ERR_LAUNCH;
    if (a_range > 0.f) hipLaunchKernelGGL(gx::fineq_fwd_kernel<true>, dim3(nwg), dim3(512), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(gx::fineq_fwd_kernel<false>, dim3(nwg), dim3(512), 0, (hipStream_t)stream, g);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// One pass over the finest taps for the whole backward of the level (fineq_bwd_kernel): S [256][256] = sum_q e[q] t t^T,
// P [K][256], Tt [rows / 4][256], Cpool [rows / 4][stride].  bf16x6 engine; fine_res a power of two >= 16.
// tt_amax (optional device float): max |Tt|, the a_amax_dev / b_amax_dev of the f16x3 GEMMs that contract Tt.
// t_amax + c_amax (both or neither; device floats: max |tap| as gx_fineq_fwd left it, the two maxima of gx_lowrank_coeff_amax):
// the f16x3 form of the pass — two fp16 planes per image, three products per tile, half the matrix-core work.
int gx_lowrank_coeff_amax(const float* C, long rows, int K, float* out2, void* stream) {
    if (!C || !out2 || rows <= 0 || K < 1 || K > 16 || ((uintptr_t)C & 15)) return GX_ERR_BAD_ARG;
    if (hipMemsetAsync(out2, 0, 2 * sizeof(float), (hipStream_t)stream) != hipSuccess) return GX_ERR_LAUNCH;
    const int ls = (K + 1 + 3) & ~3;
    long want = (rows * (ls >> 2) + 1023) / 1024;
    hipLaunchKernelGGL(gx::lowrank_coeff_amax_kernel, dim3((int)(want < 1024 ? want : 1024)), dim3(256), 0, (hipStream_t)stream, C,
                       rows, ls, K, reinterpret_cast<unsigned*>(out2));
    GX_CHECK_LAUNCH();
    return GX_OK;
}
#define GX_FINEQ_BWD_WGS 256
size_t gx_fineq_bwd_workspace_bytes(void) { return (size_t)GX_FINEQ_BWD_WGS * 272 * 256 * sizeof(float); }
int gx_fineq_bwd(const float* C, int rows, int fine_res, int K1, const float* t1, long ld1, int K2, const float* t2, long ld2,
                 int K, float* S, float* P, float* Tt, float* Cpool, float* workspace, size_t workspace_bytes, float* tt_amax,
                 const float* t_amax, const float* c_amax, void* stream) {
    if ((t_amax == nullptr) != (c_amax == nullptr)) return GX_ERR_BAD_ARG;
    if (!C || !t1 || !S || !P || !Tt || !Cpool || rows <= 0 || (K2 > 0 && !t2)) return GX_ERR_BAD_ARG;
    if (!gx_fineq_ok(K1, K2, K) || gx_get_gemm_mode() != 1) return GX_ERR_BAD_ARG;
    if (fine_res < 16 || (fine_res & (fine_res - 1)) || (long)rows % ((long)fine_res * fine_res)) return GX_ERR_BAD_ARG;
    if ((ld1 & 3) || (K2 > 0 && (ld2 & 3)) || (((uintptr_t)t1 | (uintptr_t)t2 | (uintptr_t)Tt) & 15)) return GX_ERR_ALIGN;
    if (!workspace || workspace_bytes < gx_fineq_bwd_workspace_bytes()) return GX_ERR_WORKSPACE;
    gx::FineqBwdArgs g;
    g.C = C; g.ls = (K + 1 + 3) & ~3;
    g.T1 = t1; g.ld1 = ld1;
    g.T2 = K2 > 0 ? t2 : t1 + 128; g.ld2 = K2 > 0 ? ld2 : ld1;
    g.K = K; g.shift = 0;
    while ((1 << g.shift) < fine_res) ++g.shift;
    g.nkt = rows / 16;
    int nwg = (int)(g.nkt < GX_FINEQ_BWD_WGS ? g.nkt : GX_FINEQ_BWD_WGS);
    g.kt_per = (int)((g.nkt + nwg - 1) / nwg);
    nwg = (int)((g.nkt + g.kt_per - 1) / g.kt_per);
    g.slab = workspace; g.Tt = Tt; g.Cpool = Cpool;
    hipStream_t st = (hipStream_t)stream;
    g.tt_amax = reinterpret_cast<unsigned*>(tt_amax);
    if (tt_amax && hipMemsetAsync(tt_amax, 0, sizeof(float), st) != hipSuccess) return GX_ERR_LAUNCH;
    g.t_amax = t_amax; g.c_amax = c_amax;
    if (t_amax) hipLaunchKernelGGL(gx::fineq_bwd_kernel<true>, dim3(nwg), dim3(768), 0, st, g);
    else hipLaunchKernelGGL(gx::fineq_bwd_kernel<false>, dim3(nwg), dim3(768), 0, st, g);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(gx::fineq_bwd_reduce_kernel, dim3(gx_cdiv((256 + K) * 256, 256)), dim3(256), 0, st, workspace, nwg, K, S, P);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"
