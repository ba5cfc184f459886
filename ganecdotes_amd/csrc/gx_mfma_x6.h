// fp32 contractions on the bf16 matrix cores ("bf16x6"): gfx950 runs v_mfma_f32_32x32x16_bf16 at
// 16x the rate of v_mfma_f32_32x32x2_f32 (2.5 PFLOP/s vs 157 TFLOP/s), so an fp32 product costs less
// as SIX bf16 products than as one fp32 product:
//     x = x_h + x_m + x_l  exactly  (three bf16 = 3 x 8 significant bits = the 24 bits of an fp32)
//     a*b = a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_m b_m + a_l b_h) + O(2^-24 |a||b|)
// Every bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the result carries
// fp32-level error (measured against fp64 next to the exact-fp32 MFMA kernel: tests/test_hip_kernels.py
// ::test_gemm_x6_accuracy) — the dropped terms a_m b_l + a_l b_m + a_l b_l are below one fp32 ulp of
// the product.  The split is done once per operand element on its way from HBM into LDS.
//
// Tile: 128 x 128 x 16 per workgroup (4 waves, 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles x 6 products),
// LDS double-buffered, three bf16 planes (h, m, l) per operand:
//   KC image (k-contiguous source):  plane [128 rows][16 bf16], unpadded    — fragment = one ds_read_b128
//        (32-B rows; the two 16-B halves of rows 8..15 (mod 16) are swapped, so the 16 lanes of a
//        ds_read_b128 group fall in 16 distinct 4-bank slots)
//   MC image (m-contiguous source):  plane [16 k rows][R bf16 + 64 B]      — the source's own layout, read back TRANSPOSED:
//        fragment = two ds_read_b64_tr_b16 (per 16-lane group a block of 4 k rows x 16 m columns, lane i receiving column
//        i: cdna_hip_programming.md T10).  Row stride 2 R + 64 B = 64 (mod 256): the four rows of a block and the two
//        blocks of a half-wave cover all 64 banks once.  (Rounds 1-2 packed k-pairs into words and read four ds_read_b32
//        per fragment: half the LDS rate, MI355X_MICROARCH.md "LDS", and what the TN kernels stalled on.)
// Operand lane map of v_mfma_f32_32x32x16_bf16: lane (r = l & 31, h = l >> 5) holds row r, k = 8h..8h+7.
#pragma once
#include "gx_mfma.h"

namespace gx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XK = 16;                              // fp32 k per tile = one 32x32x16 MFMA step
constexpr int X_KC_ROW = 32;                        // bytes per row of a KC plane
// Images of R rows (R = 128: one operand of the 128 x 128 tile; R = 256: the A operand of the
// wave-specialised 256 x 128 tile), filled by 256 threads.
template <int R> struct XImg {
    static constexpr int KC_PLANE = R * X_KC_ROW;             // R = 128: 4096 B
    static constexpr int MC_RS = 2 * R + 64;                  // bytes per k row of an MC plane (R = 128: 320, 256: 576)
    static constexpr int MC_PLANE = 16 * MC_RS;               // R = 128: 5120 B
    static constexpr int BYTES = 3 * MC_PLANE;                // one image (the KC form is smaller)
    static constexpr int NV = R / 64;                         // float4 per thread per tile
};
constexpr int X_TILE_BYTES = 3 * XImg<128>::KC_PLANE;          // one KC image of a 128-row operand: 12288 B
constexpr int X_TILE_BYTES_MC = XImg<128>::BYTES;              // ... when an operand may be m-contiguous: 15360 B
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// byte offset of the 16-B half `half` of row `row` inside a KC plane
__device__ __forceinline__ int x_kc_off(int row, int half) { return row * X_KC_ROW + ((half ^ ((row >> 3) & 1)) << 4); }

// (x0, x1) -> packed bf16 pairs h, m, l (element 0 in the low half) with x = h + m + l exactly.
// Each remainder is exact in fp32 (it is a multiple of the ulp of x below the rounding point).
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){x0, x1}, bf16x2));
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, bf16x2));
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){s0, s1}, bf16x2));
}

// Global -> registers for one R x 16 operand tile (R/64 float4 per thread, 256 threads, t = 0..255).
//  KC: element (row, k) at P[row*ld + k]: v[i] = row (t>>2) + 64 i, k = 4 (t&3) .. +3.
//  MC: element (row, k) at P[k*ld + row]: v[2j+i] = rows 4 (t&31) + 128 j .. +3, k = 2 (t>>5) + i.
//  VEC as in load_tile.
template <bool KC, bool VEC, int R = 128>
__device__ __forceinline__ void x6_load_tile(const float* __restrict__ P, long ld, int row0, int nrows, int k0,
                                             int kend, f32x4 (&v)[R / 64], int t) {
    constexpr int NV = R / 64;
    if (VEC) {
        const unsigned uld = (unsigned)ld;
        if (KC) {
            const int nv = min(R, nrows - row0), kv = min(XK, kend - k0);
            const unsigned nbytes = (nv > 0 && kv > 0) ? ((unsigned)(nv - 1) * uld + (unsigned)kv) * 4u : 0u;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(P + (long)row0 * ld + k0, nbytes);
            const unsigned kk = (t & 3) * 4;
            const unsigned km = oob_mask((int)kk < kv);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const unsigned r = (t >> 2) + 64 * i;
                v[i] = buf_load4(rs, ((r * uld + kk) * 4u) | km);
            }
        } else {
            const int kv = min(XK, kend - k0), mv = min(R, nrows - row0);
            const unsigned nbytes = (kv > 0 && mv > 0) ? ((unsigned)(kv - 1) * uld + (unsigned)mv) * 4u : 0u;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(P + (long)k0 * ld + row0, nbytes);
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) {
                const unsigned mm = (t & 31) * 4 + 128 * j;
                const unsigned km = oob_mask((int)mm < mv);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned kr = 2 * (t >> 5) + i;
                    v[2 * j + i] = buf_load4(rs, ((kr * uld + mm) * 4u) | km);
                }
            }
        }
        return;
    }
    if (KC) {
        const int k = k0 + (t & 3) * 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int row = row0 + (t >> 2) + 64 * i;
            v[i] = (row < nrows && k < kend) ? load4(P + (long)row * ld + k, kend - k, false) : zero4();
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV / 2; ++j) {
            const int m = row0 + (t & 31) * 4 + 128 * j;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kr = k0 + 2 * (t >> 5) + i;
                v[2 * j + i] = (kr < kend && m < nrows) ? load4(P + (long)kr * ld + m, nrows - m, false) : zero4();
            }
        }
    }
}

// registers -> the three bf16 planes of one operand image
template <bool KC, int R = 128>
__device__ __forceinline__ void x6_store_tile(char* __restrict__ S, const f32x4 (&v)[R / 64], int t) {
    constexpr int NV = R / 64;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned h[2], m[2], l[2];
            split3(v[i][0], v[i][1], h[0], m[0], l[0]);
            split3(v[i][2], v[i][3], h[1], m[1], l[1]);
            char* p = S + x_kc_off((t >> 2) + 64 * i, (t >> 1) & 1) + (t & 1) * 8;
            *reinterpret_cast<u32x2*>(p) = (u32x2){h[0], h[1]};
            *reinterpret_cast<u32x2*>(p + XImg<R>::KC_PLANE) = (u32x2){m[0], m[1]};
            *reinterpret_cast<u32x2*>(p + 2 * XImg<R>::KC_PLANE) = (u32x2){l[0], l[1]};
        }
    } else {
        // v[2 j + i] = rows (m) 4 (t & 31) + 128 j .. + 3 at k = 2 (t >> 5) + i: four bf16 = 8 B of k row 2 (t >> 5) + i
#pragma unroll
        for (int j = 0; j < NV / 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned h[2], m[2], l[2];
                split3(v[2 * j + i][0], v[2 * j + i][1], h[0], m[0], l[0]);
                split3(v[2 * j + i][2], v[2 * j + i][3], h[1], m[1], l[1]);
                char* p = S + (2 * (t >> 5) + i) * XImg<R>::MC_RS + ((t & 31) * 4 + 128 * j) * 2;
                *reinterpret_cast<u32x2*>(p) = (u32x2){h[0], h[1]};
                *reinterpret_cast<u32x2*>(p + XImg<R>::MC_PLANE) = (u32x2){m[0], m[1]};
                *reinterpret_cast<u32x2*>(p + 2 * XImg<R>::MC_PLANE) = (u32x2){l[0], l[1]};
            }
    }
}

template <bool KC, int R = 128>
__device__ __forceinline__ bf16x8 x6_frag(const char* __restrict__ S, int plane, int row, int h) {
    if (KC) return *reinterpret_cast<const bf16x8*>(S + plane * XImg<R>::KC_PLANE + x_kc_off(row, h));
    // transposed read: `row` = the fragment's first row (a multiple of 32) + (lane & 31).  Lane 4 q + p of a 16-lane group
    // supplies the address of k row q, columns 4 p .. 4 p + 3 of its block and receives the block's column (lane & 15);
    // groups 0 / 1 of a half-wave are the fragment's rows 0-15 / 16-31, the two reads its k = 8 h .. + 3 and 8 h + 4 .. + 7.
    // Every lane of the wave must be active (EXEC all ones: the gather crosses lanes).
    typedef __attribute__((address_space(3))) s16x4* lp;
    const int r = row & 31, q = (r & 15) >> 2, p = r & 3;
    const char* a = S + plane * XImg<R>::MC_PLANE + (8 * h + q) * XImg<R>::MC_RS + ((row - r) + 16 * (r >> 4) + 4 * p) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(a + 4 * XImg<R>::MC_RS));
    return __builtin_bit_cast(bf16x8, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
}

// One 16-deep k-tile for a wave's (32 MT) x (32 NT) sub-tile at rows a_row0 / b_row0 of the images:
// MT x NT MFMA tiles x 6 products, smallest terms first.
template <bool A_KC, bool B_KC, int MT, int NT, int RA, int RB>
__device__ __forceinline__ void x6_mfma_tile(const char* __restrict__ As, const char* __restrict__ Bs,
                                             f32x16 (&acc)[MT][NT], int a_row0, int b_row0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    bf16x8 a[MT][3], b[NT][3];
    // fragment reads in the order the products consume them (l*h first), so the first MFMAs can
    // issue when a third of the reads have landed
    constexpr int PA[3] = {2, 0, 1}, PB[3] = {0, 2, 1};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t][PA[q]] = x6_frag<A_KC, RA>(As, PA[q], a_row0 + t * 32 + r, h);
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t][PB[q]] = x6_frag<B_KC, RB>(Bs, PB[q], b_row0 + t * 32 + r, h);
    }
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};     // l*h, h*l, m*m, m*h, h*m, h*h
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt][TA[t]], b[nt][TB[t]], acc[mt][nt], 0, 0, 0);
}
// ---- "f16x3": fp32 contractions as THREE fp16 products (the conv engine of round 5; tools/lab/mfma_f16x3.hip) ------------
//     s x = x_h + x_l + O(2^-23 |s x|)        two fp16 planes of the SCALED element (11 + 11 significant bits; below the fp16
//                                             normal range x_l keeps an absolute 2^-25: the matrix cores do not flush fp16
//                                             subnormals, measured)
//     a b = (a_h b_h + a_h b_l + a_l b_h) / (s_a s_b) + the dropped a_l b_l: <= 2^-22 |a b|, ~ 2^-25 in the mean
// Measured against fp64 beside bf16x6 and an fp32 fma chain (K = 512 / 4608, N(0,1) x N(0, 0.02 .. 30), heavy tails): the
// same 3e-7 / 7e-7 norm-wise — while the matrix cores do HALF the work and each operand is 4 B, not 6, in HBM and LDS.
// The price is RANGE: |s x| must stay below 65504 (fp16 max), and elements below 2^-14 / s lose relative (not absolute)
// precision.  The scales are powers of two, fixed per operand kind (activations x 2: |x| < 32752, absolute floor 1.5e-8;
// weights x 16: |w| < 4094), and leave the
// accumulator exactly in the epilogue; an element out of range becomes inf - inf = NaN in its low plane — loud, not wrong.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
constexpr float XH_SCALE_X = 2.f, XH_SCALE_W = 16.f, XH_UNSCALE = 1.f / (XH_SCALE_X * XH_SCALE_W);
// The power of two that brings a tensor of maximum magnitude `amax` to [2^13, 2^14): two binades below the fp16 maximum (a
// bound that is exceeded by less than 4 x still converts); elements down to amax 2^-16 keep their full 22 bits, below that
// the low plane is an fp16 subnormal: an absolute error of amax 2^-38.
__host__ __device__ __forceinline__ float xh_scale_of(float amax) {
    unsigned bits;
    __builtin_memcpy(&bits, &amax, 4);
    int be = (int)((bits >> 23) & 0xffu);                              // biased exponent (0: zero / subnormal -> no scaling)
    if (be == 0 || be == 255) return 1.f;
    int se = 267 - be;                                                  // 127 + 13 - (be - 127)
    se = se > 254 ? 254 : (se < 1 ? 1 : se);
    const unsigned sb = (unsigned)se << 23;
    float sc;
    __builtin_memcpy(&sc, &sb, 4);
    return sc;
}
// (x0, x1) * scale -> packed fp16 pairs h, l (element 0 in the low half); the remainder is exact in fp32
__device__ __forceinline__ void split2h(float x0, float x1, float scale, unsigned& h, unsigned& l) {
    const f32x2 v = {x0 * scale, x1 * scale};
    const f16x2 hh = __builtin_convertvector(v, f16x2);                 // v_cvt_f16_f32: round to nearest even
    h = __builtin_bit_cast(unsigned, hh);
    const f32x2 r = v - __builtin_convertvector(hh, f32x2);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// registers -> the two fp16 planes (0 = h, 1 = l) of one operand image, in the layouts of x6_store_tile
template <bool KC, int R = 128>
__device__ __forceinline__ void xh_store_tile(char* __restrict__ S, const f32x4 (&v)[R / 64], int t, float scale) {
    constexpr int NV = R / 64;
    if (KC) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            unsigned h[2], l[2];
            split2h(v[i][0], v[i][1], scale, h[0], l[0]);
            split2h(v[i][2], v[i][3], scale, h[1], l[1]);
            char* p = S + x_kc_off((t >> 2) + 64 * i, (t >> 1) & 1) + (t & 1) * 8;
            *reinterpret_cast<u32x2*>(p) = (u32x2){h[0], h[1]};
            *reinterpret_cast<u32x2*>(p + XImg<R>::KC_PLANE) = (u32x2){l[0], l[1]};
        }
    } else {
#pragma unroll
        for (int j = 0; j < NV / 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned h[2], l[2];
                split2h(v[2 * j + i][0], v[2 * j + i][1], scale, h[0], l[0]);
                split2h(v[2 * j + i][2], v[2 * j + i][3], scale, h[1], l[1]);
                char* p = S + (2 * (t >> 5) + i) * XImg<R>::MC_RS + ((t & 31) * 4 + 128 * j) * 2;
                *reinterpret_cast<u32x2*>(p) = (u32x2){h[0], h[1]};
                *reinterpret_cast<u32x2*>(p + XImg<R>::MC_PLANE) = (u32x2){l[0], l[1]};
            }
    }
}
// One 16-deep k-tile of the two-plane images (plane 0 = h, 1 = l; the KC layout of the bf16 images), smallest terms first
template <int MT, int NT, int RA, int RB, bool A_KC = true, bool B_KC = true>
__device__ __forceinline__ void xh_mfma_tile(const char* __restrict__ As, const char* __restrict__ Bs, f32x16 (&acc)[MT][NT],
                                             int a_row0, int b_row0, int lane) {
    const int r = lane & 31, h = lane >> 5;
    f16x8 a[MT][2], b[NT][2];
    constexpr int PA[2] = {1, 0}, PB[2] = {0, 1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t][PA[q]] = __builtin_bit_cast(f16x8, x6_frag<A_KC, RA>(As, PA[q], a_row0 + t * 32 + r, h));
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t][PB[q]] = __builtin_bit_cast(f16x8, x6_frag<B_KC, RB>(Bs, PB[q], b_row0 + t * 32 + r, h));
    }
    constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};                  // l*h, h*l, h*h
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mt][TA[t]], b[nt][TB[t]], acc[mt][nt], 0, 0, 0);
}

template <bool A_KC, bool B_KC>
__device__ __forceinline__ void x6_mfma_ktile(const char* __restrict__ As, const char* __restrict__ Bs,
                                              f32x16 (&acc)[2][2], int wr, int wc, int lane) {
    x6_mfma_tile<A_KC, B_KC, 2, 2, 128, 128>(As, Bs, acc, wr * 64, wc * 64, lane);
}

}  // namespace gx
