// fp32-in / fp32-accumulate MFMA tile machinery for gfx950 (v_mfma_f32_32x32x2_f32).
//
// Why fp32 MFMA: the Sinkhorn stage multiplies score errors by 1/eps (200-333x), so the
// projection / prototype / conv contractions must stay exact-fp32.  The
// 32x32x2 f32 MFMA is a k-ordered fmaf chain at 64 FLOP/clk/SIMD (157 TFLOP/s chip peak).
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128 x 32.  Each wave owns a
// 64 x 64 sub-tile = 2 x 2 MFMA tiles (4 x 16 accumulator VGPRs).  Operands are staged
// through LDS in one of two images:
//   KC ("k-contiguous source"): [128 rows][32 k + 4 pad] — fragments read as ds_read_b128;
//        row stride 144 B makes every 16-lane ds_read_b128 group hit 16 distinct slots.
//   MC ("m-contiguous source"): [32 k][128 m + 4 pad]   — fragments read as ds_read_b32,
//        lanes 0..31 consecutive m -> conflict-free.
// k-order inside a 32-deep tile is permuted so that a KC lane consumes its float4 in four
// consecutive MFMAs: MFMA step (j,t) contracts k = 8j + t (lanes 0-31) and k = 8j + 4 + t
// (lanes 32-63); both operands use the same permutation, so the sum is over all 32 k.
#pragma once
#include "gx_common.h"

namespace gx {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK = BK + 4;           // KC image row stride (floats)
constexpr int LDM = 128 + 4;          // MC image row stride (floats)
constexpr int TILE_FLOATS = 128 * LDK;  // 4608 >= BK*LDM (4224): one operand, one buffer
constexpr int GEMM_THREADS = 256;

// Load up to 4 consecutive floats; `nvalid` = how many of them exist.
__device__ __forceinline__ f32x4 load4(const float* p, int nvalid, bool vec_ok) {
    if (nvalid >= 4 && vec_ok) return *reinterpret_cast<const f32x4*>(p);
    f32x4 v = zero4();
    if (nvalid > 0) v[0] = p[0];
    if (nvalid > 1) v[1] = p[1];
    if (nvalid > 2) v[2] = p[2];
    if (nvalid > 3) v[3] = p[3];
    return v;
}

// ---- buffer (SRD) loads: hardware range check instead of branches ------------------------------
// A 128 x 32 operand tile is addressed through a buffer descriptor whose base is the tile's first
// element and whose extent ends at the tile's last valid element, so rows past the matrix edge
// read as zero with no branch and no select; the (thread-constant) column validity is folded
// into the offset.  All descriptor inputs are wave-uniform (kernel arguments, blockIdx, loop
// counters), so hipcc keeps them in SGPRs (no waterfall loops).
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
// 0 for a valid lane, 0xFFFFFFF0 (OR-ed into the byte offset -> out of range -> zeros) otherwise.
// The empty asm keeps the value opaque so hipcc cannot turn the select into divergent branches
// around the loads (it does otherwise, serialising them behind s_waitcnt vmcnt(0)).
__device__ __forceinline__ unsigned oob_mask(bool ok) {
    unsigned m = ok ? 0u : 0xFFFFFFF0u;
    asm volatile("" : "+v"(m));
    return m;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)nbytes, 0x00020000);
}

// Global -> registers for one 128 x 32 operand tile (4 float4 per thread).
//  KC: element (row, k) at P[row*ld + k];  MC: element (row, k) at P[k*ld + row].
//  VEC (fast path): 16-B aligned base, ld % 4 == 0, K % 4 == 0 (and rows % 4 == 0 for MC), ld < 2^22.
template <bool KC, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, long ld, int row0, int nrows,
                                          int k0, int kend, f32x4 (&v)[4], int tid) {
    if (VEC) {
        const unsigned uld = (unsigned)ld;
        if (KC) {
            const int nv = min(128, nrows - row0), kv = min(BK, kend - k0);
            const unsigned nbytes = (nv > 0 && kv > 0) ? ((unsigned)(nv - 1) * uld + (unsigned)kv) * 4u : 0u;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(P + (long)row0 * ld + k0, nbytes);
            const unsigned kk = (tid & 7) * 4;
            const unsigned km = oob_mask((int)kk < kv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned r = (tid >> 3) + 32 * i;
                v[i] = buf_load4(rs, ((r * uld + kk) * 4u) | km);
            }
        } else {
            const int kv = min(BK, kend - k0), mv = min(128, nrows - row0);
            const unsigned nbytes = (kv > 0 && mv > 0) ? ((unsigned)(kv - 1) * uld + (unsigned)mv) * 4u : 0u;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(P + (long)k0 * ld + row0, nbytes);
            const unsigned mm = (tid & 31) * 4;
            const unsigned km = oob_mask((int)mm < mv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned kr = (tid >> 5) + 8 * i;
                v[i] = buf_load4(rs, ((kr * uld + mm) * 4u) | km);
            }
        }
        return;
    }
    if (KC) {
        const int k = k0 + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + (tid >> 3) + 32 * i;
            v[i] = (row < nrows && k < kend) ? load4(P + (long)row * ld + k, kend - k, false) : zero4();
        }
    } else {
        const int m = row0 + (tid & 31) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = k0 + (tid >> 5) + 8 * i;
            v[i] = (kr < kend && m < nrows) ? load4(P + (long)kr * ld + m, nrows - m, false) : zero4();
        }
    }
}

template <bool KC>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const f32x4 (&v)[4], int tid) {
    if (KC) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(&S[((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4]) = v[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f32x4*>(&S[((tid >> 5) + 8 * i) * LDM + (tid & 31) * 4]) = v[i];
    }
}

// One 32-deep k-tile of MFMAs for this wave's 64 x 64 sub-tile.
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void mfma_ktile(const float* __restrict__ As, const float* __restrict__ Bs,
                                           f32x16 (&acc)[2][2], int wr, int wc, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (A_KC) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&As[(wr * 64 + mt * 32 + r) * LDK + 8 * j + 4 * h]);
                a[mt][0] = t[0]; a[mt][1] = t[1]; a[mt][2] = t[2]; a[mt][3] = t[3];
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) a[mt][t] = As[(8 * j + 4 * h + t) * LDM + wr * 64 + mt * 32 + r];
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            if (B_KC) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(&Bs[(wc * 64 + nt * 32 + r) * LDK + 8 * j + 4 * h]);
                b[nt][0] = t[0]; b[nt][1] = t[1]; b[nt][2] = t[2]; b[nt][3] = t[3];
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) b[nt][t] = Bs[(8 * j + 4 * h + t) * LDM + wc * 64 + nt * 32 + r];
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt][t], b[nt][t], acc[mt][nt], 0, 0, 0);
    }
}

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
}

// C/D fragment coordinates of accumulator register `reg` (0..15) of MFMA tile (mt, nt).
__device__ __forceinline__ int acc_row(int wr, int mt, int reg, int lane) {
    return wr * 64 + mt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int wc, int nt, int lane) { return wc * 64 + nt * 32 + (lane & 31); }

// XCD-aware bijective remap of a linear block id: blocks b and b+8 share an XCD (L2), so
// hand each XCD a contiguous run of logical tile ids (tiles that share operand panels).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, slot = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

}  // namespace gx
