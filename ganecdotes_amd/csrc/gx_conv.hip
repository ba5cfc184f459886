// StyleGAN2 generator forward kernels for gfx950.
//
// Activations are NHWC ([b][y][x][c]): the implicit-GEMM A operand (pixel x (tap, cin)) is
// then k-contiguous, and every hidden-feature tap is written once in exactly the layout the
// SwAV gather wants (one contiguous C*4-byte row per pixel).
//
// Modulated conv (models/stylegan2/model.py:327-368) is re-associated so the weight is
// shared by the whole batch:   out[b,co] = demod'[b,co] * sum_{tap,ci} (x[b,ci]*s[b,ci]) * W[co,ci,tap]
// with demod'[b,co] = scale * rsqrt(scale^2 * sum_ci s[b,ci]^2 * Wsq[co,ci] + 1e-8), scale =
// 1/sqrt(Cin k^2), Wsq = sum_taps W^2 (ToRGB: demod' = scale).  The contraction runs on the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) as an implicit GEMM, M = B*OH*OW, N = Cout,
// K = ntaps*Cin; noise + bias + leaky-ReLU*sqrt2 (fused_bias_act_kernel.cu:18-85 semantics)
// are the epilogue.  The stride-2 transposed conv of the upsampling layers is four such
// GEMMs (one per output parity class, 4+2+2+1 taps) writing the (2H+1)^2 pre-blur image,
// followed by one fused FIR-blur + demod + noise + bias + act pass (upfirdn2d mode 1,
// upfirdn2d_kernel.cu:114-215 semantics).
#include "gx_mfma_x6.h"
#include <cstdlib>

// (prototypes of the C ABI: ganecdotes_hip.h via gx_common.h)

namespace gx {

struct ConvSub {                               // one implicit-GEMM sub-problem of a launch
    int ntaps; int dy[9], dx[9], tw[9];        // input offsets and weight-tap index per tap
    int OH, OW;                                // output sub-grid (M = B*OH*OW)
    int yadd, xadd;                            // out y = oy*ymul + yadd
    int tile_base, tiles_m;                    // logical tile ids [tile_base, tile_base + tiles_m*tiles_n)
};
struct ConvArgs {
    const float* in; int B, IH, IW, Cin;      // NHWC input
    const float* wt;                           // [9 or 1][Cout][Cin]
    const unsigned short* wtp;                 // optional pre-split weights (gx_split_conv_weights), bf16x6 engine
    int wtp_ncc; unsigned wtp_bytes;           // 16-channel chunks per (tap, Cout tile); extent for the buffer descriptor
    const float* style;                        // [B][Cin] modulation or nullptr
    int nsub; ConvSub sub[4];                  // 1 (plain conv) or 4 (parity classes of the stride-2 transposed conv)
    float* out; int outH, outW, Cout;          // NHWC output tensor
    int ymul, xmul;
    const float* demod;                        // [B][Cout] or nullptr
    const float* noise; long noise_bstride;    // [.][outH][outW] or nullptr
    const float* noise_w;                      // device scalar
    const float* bias;                         // [Cout] or nullptr
    int act;                                   // 1: lrelu(0.2)*sqrt2
    int vec_ok;
    int tiles_n, ntile;
    unsigned in_bytes, wt_bytes, style_bytes;  // extents for the buffer descriptors (VEC path)
    int splits; float* slab;                   // split-K over k-tiles: raw partial sums [split][B*outH*outW*Cout]
    const unsigned short* xp;                  // optional pre-split MODULATED input (gx_modsplit): see modconv_xp_kernel
    unsigned xp_plane_bytes;                   // bytes of one (chunk, plane) image = B*(IH+2)*(IW+2)*32
    const float* s_next;                       // [B][Cout] modulation of the NEXT conv, or nullptr
    unsigned short* xp_out;                    // x planes of the output for the next conv (out * s_next), or nullptr
    int h2;                                    // plane format of xp / wtp / xp_out: 0 = three bf16 planes (bf16x6), 1 = two fp16 planes (f16x3)
    float acc_scale;                           // what the accumulators are multiplied by on their way out (f16x3: 1 / (scale_x scale_w), else 1)
};

__device__ __forceinline__ float lrelu_sqrt2(float v) { return (v > 0.f ? v : 0.2f * v) * 1.41421356237309515f; }

// Epilogue of a wave's (32 MT) x 64 accumulator sub-tile whose first row (output pixel index within
// the sub-problem) is row0 and first output channel col0.  A lane's 16 rows of an MFMA tile are
// m = base + (r&3) + 8*(r>>2): decode the base once, then walk the rows with carries instead of 64
// integer divisions per lane.
template <int MT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& g, const ConvSub& sb, f32x16 (&acc)[MT][2], unsigned row0,
                                              int col0, int lane) {
    const int OH = sb.OH, OW = sb.OW;
    const unsigned ohw = (unsigned)OH * (unsigned)OW;
    const unsigned Mtot = (unsigned)g.B * ohw;
    const bool to_slab = g.slab != nullptr;
    float* outp = to_slab ? g.slab + (long)blockIdx.y * g.B * g.outH * g.outW * g.Cout : g.out;
    const float nw = (!to_slab && g.noise && g.noise_w) ? *g.noise_w : 0.f;
    int co[2]; bool cok[2]; float bias_c[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        co[nt] = col0 + nt * 32 + (lane & 31);
        cok[nt] = co[nt] < g.Cout;
        bias_c[nt] = (!to_slab && g.bias && cok[nt]) ? g.bias[co[nt]] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const unsigned mbase = row0 + mt * 32 + 4 * (lane >> 5);
        unsigned b = mbase / ohw, rem = mbase - b * ohw;
        int oy = (int)(rem / (unsigned)OW), ox = (int)(rem - (unsigned)oy * OW);
        int prev = 0, bcur = -1;
        float dm[2] = {1.f, 1.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int delta = (r & 3) + 8 * (r >> 2);
            ox += delta - prev; prev = delta;
            while (ox >= OW) { ox -= OW; ++oy; }
            while (oy >= OH) { oy -= OH; ++b; }
            if (mbase + delta >= Mtot) continue;
            const int y = oy * g.ymul + sb.yadd, x = ox * g.xmul + sb.xadd;
            const long pix = ((long)b * g.outH + y) * g.outW + x;
            float nz = 0.f;
            if (!to_slab) {
                if (g.noise) nz = nw * g.noise[(long)b * g.noise_bstride + (long)y * g.outW + x];
                if (g.demod && (int)b != bcur) {
                    bcur = (int)b;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) dm[nt] = cok[nt] ? g.demod[(long)b * g.Cout + co[nt]] : 1.f;
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                if (!cok[nt]) continue;
                float v = acc[mt][nt][r] * g.acc_scale;
                if (!to_slab) {
                    v = fmaf(v, dm[nt], nz + bias_c[nt]);
                    if (g.act) v = lrelu_sqrt2(v);
                }
                outp[pix * g.Cout + co[nt]] = v;
            }
        }
    }
}

// Main loop of one output tile for a sub-problem with NT taps (compile-time: the tap loop is
// unrolled, so per-tap offsets live in SGPRs and nothing is fetched from the argument block inside
// the loop).  k order = channel chunk (32 ch) outer, tap inner: the NT shifted reads of one
// 32-channel slab of the input rows are consecutive (L1/L2 hits), and the modulation vector of
// the chunk is loaded once per chunk, not per tap.
template <bool VEC, int NT>
__device__ __forceinline__ void conv_mainloop(const ConvArgs& g, const ConvSub& sb, float* lds, f32x16 (&acc)[2][2],
                                              const long (&pix_off)[4], const int (&soff)[4],
                                              const unsigned (&tapmask)[4], int n0, int cc0, int ncc, int tid, int wr,
                                              int wc, int lane) {
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(g.in, VEC ? g.in_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_st = make_rsrc(g.style ? g.style : g.in, (VEC && g.style) ? g.style_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(g.wt, VEC ? g.wt_bytes : 0u);
    const int kq = (tid & 7) * 4;
    long toff[NT], wb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        toff[t] = ((long)sb.dy[t] * g.IW + sb.dx[t]) * g.Cin;
        wb[t] = (long)sb.tw[t] * g.Cout * g.Cin;
    }
    unsigned rowok[4];                                            // weight rows (co) this thread stages
#pragma unroll
    for (int i = 0; i < 4; ++i) rowok[i] = (n0 + (tid >> 3) + 32 * i) < g.Cout;
    f32x4 ra[4], rw[4], st[4];

    auto load_style = [&](int cc) {                               // once per channel chunk
        const int ci = cc * BK + kq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!g.style) { st[i] = f32x4{1.f, 1.f, 1.f, 1.f}; continue; }
            if (VEC) st[i] = buf_load4(rs_st, (unsigned)((soff[i] + ci) * 4) | oob_mask(ci < g.Cin));
            else st[i] = ci < g.Cin ? load4(g.style + soff[i] + ci, g.Cin - ci, false) : zero4();
        }
    };
    auto load_ab = [&](int cc, int tap, long to, long wbt) {      // tap / to / wbt are compile-time-indexed
        const int ci = cc * BK + kq;
        const bool cok = ci < g.Cin;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = ((tapmask[i] >> tap) & 1u) && cok;
            if (VEC) ra[i] = buf_load4(rs_in, (unsigned)((pix_off[i] + to + ci) * 4) | oob_mask(ok));
            else ra[i] = ok ? load4(g.in + pix_off[i] + to + ci, g.Cin - ci, false) : zero4();
            const long wo = wbt + (long)(n0 + (tid >> 3) + 32 * i) * g.Cin + ci;
            const bool wok = rowok[i] && cok;
            if (VEC) rw[i] = buf_load4(rs_wt, (unsigned)(wo * 4) | oob_mask(wok));
            else rw[i] = wok ? load4(g.wt + wo, g.Cin - ci, false) : zero4();
        }
    };
    auto stage = [&](float* dst) {                                // modulate, then registers -> LDS
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] *= st[i];
        store_tile<true>(dst, ra, tid);
        store_tile<true>(dst + TILE_FLOATS, rw, tid);
    };

    if (ncc > 0) {
        load_style(cc0);
        load_ab(cc0, 0, toff[0], wb[0]);
        stage(lds);
    }
    __syncthreads();
    int buf = 0;
    for (int cc = cc0; cc < cc0 + ncc; ++cc) {
        const bool last_cc = (cc + 1 == cc0 + ncc);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            float* As = lds + buf * 2 * TILE_FLOATS;
            const bool wrap = (tap + 1 == NT);
            const bool more = !(wrap && last_cc);
            if (more) load_ab(wrap ? cc + 1 : cc, wrap ? 0 : tap + 1, toff[wrap ? 0 : tap + 1], wb[wrap ? 0 : tap + 1]);
            mfma_ktile<true, true>(As, As + TILE_FLOATS, acc, wr, wc, lane);
            if (more) {
                if (wrap) load_style(cc + 1);                     // next chunk's modulation (used by stage below)
                stage(lds + (buf ^ 1) * 2 * TILE_FLOATS);
            }
            __syncthreads();
            buf ^= 1;
        }
    }
}

// The same main loop on the bf16 matrix cores (gx_mfma_x6.h): 16-channel chunks, the thread stages
// rows (tid >> 2) + 64 i (i < 2) of the pixel tile and of the weight tile, channels 4 (tid & 3) .. +3;
// the modulated activation and the weight are split into three bf16 planes on their way into LDS.
// WP: the weight tile comes pre-split (three bf16 planes in HBM, made once per model by gx_split_bf16x3):
// a thread copies 16 B per plane straight into the LDS image (row tid >> 1, half tid & 1), no arithmetic.
template <bool VEC, int NT, bool WP>
__device__ __forceinline__ void conv_mainloop_x6(const ConvArgs& g, const ConvSub& sb, char* lds, f32x16 (&acc)[2][2],
                                                 const long (&pix_off)[4], const int (&soff)[4],
                                                 const unsigned (&tapmask)[4], int n0, int cc0, int ncc, int tid,
                                                 int wr, int wc, int lane) {
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(g.in, VEC ? g.in_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_st = make_rsrc(g.style ? g.style : g.in, (VEC && g.style) ? g.style_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(g.wt, VEC ? g.wt_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_wp = make_rsrc(reinterpret_cast<const float*>(g.wtp ? (const void*)g.wtp : (const void*)g.wt),
                                                   WP ? g.wtp_bytes : 0u);
    const int wp_ntn = (g.Cout + 127) >> 7;
    u32x4 rwp[3];
    const int kq = (tid & 3) * 4;
    long toff[NT], wb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        toff[t] = ((long)sb.dy[t] * g.IW + sb.dx[t]) * g.Cin;
        wb[t] = (long)sb.tw[t] * g.Cout * g.Cin;
    }
    unsigned rowok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) rowok[i] = (n0 + (tid >> 2) + 64 * i) < g.Cout;
    f32x4 ra[2], rw[2], st[2];

    auto load_style = [&](int cc) {
        const int ci = cc * XK + kq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (!g.style) { st[i] = f32x4{1.f, 1.f, 1.f, 1.f}; continue; }
            if (VEC) st[i] = buf_load4(rs_st, (unsigned)((soff[i] + ci) * 4) | oob_mask(ci < g.Cin));
            else st[i] = ci < g.Cin ? load4(g.style + soff[i] + ci, g.Cin - ci, false) : zero4();
        }
    };
    auto load_ab = [&](int cc, int tap, long to, long wbt) {
        const int ci = cc * XK + kq;
        const bool cok = ci < g.Cin;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = ((tapmask[i] >> tap) & 1u) && cok;
            if (VEC) ra[i] = buf_load4(rs_in, (unsigned)((pix_off[i] + to + ci) * 4) | oob_mask(ok));
            else ra[i] = ok ? load4(g.in + pix_off[i] + to + ci, g.Cin - ci, false) : zero4();
            if (WP) continue;
            const long wo = wbt + (long)(n0 + (tid >> 2) + 64 * i) * g.Cin + ci;
            const bool wok = rowok[i] && cok;
            if (VEC) rw[i] = buf_load4(rs_wt, (unsigned)(wo * 4) | oob_mask(wok));
            else rw[i] = wok ? load4(g.wt + wo, g.Cin - ci, false) : zero4();
        }
        if (WP) {      // block (tap, Cout tile, chunk): 3 planes x 4 KB, copied linearly
            const unsigned blk = (unsigned)(((sb.tw[tap] * wp_ntn + (n0 >> 7)) * g.wtp_ncc + cc) * 3) * 4096u;
            const unsigned m = oob_mask(cc < g.wtp_ncc);
#pragma unroll
            for (int p = 0; p < 3; ++p)
                rwp[p] = __builtin_bit_cast(u32x4, buf_load4(rs_wp, (blk + p * 4096u + (unsigned)tid * 16u) | m));
        }
    };
    auto stage = [&](char* dst) {
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] *= st[i];
        x6_store_tile<true>(dst, ra, tid);
        if (WP) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
                *reinterpret_cast<u32x4*>(dst + X_TILE_BYTES + p * XImg<128>::KC_PLANE + tid * 16) = rwp[p];
        } else {
            x6_store_tile<true>(dst + X_TILE_BYTES, rw, tid);
        }
    };

    if (ncc > 0) {
        load_style(cc0);
        load_ab(cc0, 0, toff[0], wb[0]);
        stage(lds);
    }
    __syncthreads();
    int buf = 0;
    for (int cc = cc0; cc < cc0 + ncc; ++cc) {
        const bool last_cc = (cc + 1 == cc0 + ncc);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            char* As = lds + buf * 2 * X_TILE_BYTES;
            const bool wrap = (tap + 1 == NT);
            const bool more = !(wrap && last_cc);
            if (more) load_ab(wrap ? cc + 1 : cc, wrap ? 0 : tap + 1, toff[wrap ? 0 : tap + 1], wb[wrap ? 0 : tap + 1]);
            x6_mfma_ktile<true, true>(As, As + X_TILE_BYTES, acc, wr, wc, lane);
            if (more) {
                if (wrap) load_style(cc + 1);
                stage(lds + (buf ^ 1) * 2 * X_TILE_BYTES);
            }
            __syncthreads();
            buf ^= 1;
        }
    }
}

// VEC: in / wt / style 16-B aligned and Cin % 4 == 0 (every real layer); otherwise the scalar path.
// X6: contraction on the bf16 matrix cores (three-way split, gx_mfma_x6.h) instead of the fp32 MFMA.
template <bool VEC, bool X6>
__global__ __launch_bounds__(GEMM_THREADS, 2) void modconv_kernel(ConvArgs g) {
    constexpr int CK = X6 ? XK : BK;                              // channels per k-tile
    constexpr int NR = X6 ? 2 : 4;                                // pixel rows a thread stages
    __shared__ __attribute__((aligned(16))) char lds_raw[X6 ? 4 * X_TILE_BYTES : 4 * TILE_FLOATS * 4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1;
    // sub-problem from the block id range; XCD-aware remap INSIDE the sub-problem so that every
    // XCD gets an equal share of each class (classes differ 4x in work per tile)
    int si = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.nsub && (int)blockIdx.x >= g.sub[i].tile_base) si = i;
    const ConvSub& sb = g.sub[si];
    const int local = xcd_remap(blockIdx.x - sb.tile_base, sb.tiles_m * g.tiles_n);
    const int tm = local / g.tiles_n, tn = local - tm * g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int OH = sb.OH, OW = sb.OW, ntaps = sb.ntaps;
    const unsigned ohw = (unsigned)OH * (unsigned)OW;
    const unsigned Mtot = (unsigned)g.B * ohw;                    // host guarantees < 2^31

    // per-thread constants of the 4 pixel rows it stages: input offset of the (un-shifted)
    // pixel, style row offset, and one validity bit per tap (zero padding / row >= M)
    long pix_off[4];
    int soff[4];
    unsigned tapmask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned m = (unsigned)m0 + (X6 ? (tid >> 2) + 64 * i : (tid >> 3) + 32 * i);
        pix_off[i] = 0; soff[i] = 0; tapmask[i] = 0;
        if (i < NR && m < Mtot) {
            const unsigned b = m / ohw, rem = m - b * ohw;
            const int oy = (int)(rem / (unsigned)OW), ox = (int)(rem - (unsigned)oy * OW);
            pix_off[i] = (((long)b * g.IH + oy) * g.IW + ox) * g.Cin;
            soff[i] = (int)b * g.Cin;
            for (int t = 0; t < ntaps; ++t) {
                const int iy = oy + sb.dy[t], ix = ox + sb.dx[t];
                if (iy >= 0 && iy < g.IH && ix >= 0 && ix < g.IW) tapmask[i] |= 1u << t;
            }
        }
    }
    // split-K over channel chunks
    const int ncc_all = (g.Cin + CK - 1) / CK;
    const int cps = (ncc_all + g.splits - 1) / g.splits;
    const int cc0 = blockIdx.y * cps;
    const int ncc = max(0, min(ncc_all, cc0 + cps) - cc0);

    f32x16 acc[2][2];
    zero_acc(acc);
#define GX_CONV_LOOP(NT_)                                                                                       \
    do {                                                                                                      \
        if (X6 && VEC && g.wtp) conv_mainloop_x6<VEC, NT_, true>(g, sb, lds_raw, acc, pix_off, soff, tapmask, n0, cc0, ncc, tid, wr, wc, lane); \
        else if (X6) conv_mainloop_x6<VEC, NT_, false>(g, sb, lds_raw, acc, pix_off, soff, tapmask, n0, cc0, ncc, tid, wr, wc, lane); \
        else conv_mainloop<VEC, NT_>(g, sb, reinterpret_cast<float*>(lds_raw), acc, pix_off, soff, tapmask, n0, cc0,  \
                                     ncc, tid, wr, wc, lane);                                                 \
    } while (0)
    switch (ntaps) {
        case 9: GX_CONV_LOOP(9); break;
        case 4: GX_CONV_LOOP(4); break;
        case 2: GX_CONV_LOOP(2); break;
        default: GX_CONV_LOOP(1); break;
    }
#undef GX_CONV_LOOP

    conv_epilogue<2>(g, sb, acc, (unsigned)m0 + wr * 64, n0 + wc * 64, lane);
}

// ---- the modulated input, split once ("x planes") -------------------------------------------------------
// modconv_kernel<.., X6> re-splits every activation element (x * s into three bf16) once per tap and per Cout
// tile: 9-36 x redundant VALU + LDS-write work next to the MFMAs.  Here the split happens ONCE, in the kernel that
// produces the activation (or in modsplit_kernel), into
//     xp [ncc = ceil(Cin/16)][3 planes h, m, l][B][IH+2][IW+2][16] bf16        (zero border, zero channel padding)
// i.e. chunk-major, pixel-contiguous: the 128-pixel x 16-channel A tile of tap (dy, dx) is, per plane, the
// 32-B records of 128 padded pixel indices that differ from the un-shifted ones by the CONSTANT dy*(IW+2)+dx — no
// masks (the border is real zeros), and for tiles inside one image row one contiguous 4 KB run.  Both operands
// then go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no VALU, no ds_write; the
// layout of the LDS image (half-major for the default 16x16x32 product form, the half-swap swizzle x_kc_off for the
// 32x32x16 form) is applied on the per-lane SOURCE address, the LDS side of a DMA is linear.
// Main loop per wave and k-tile: 6 DMA + 20 ds_read_b128 + 48 MFMA 16x16x32 (or 12 reads + 24 MFMA 32x32x16).
constexpr int XP_STAGE_BYTES = 2 * 3 * XImg<128>::KC_PLANE;       // A planes + B planes of one k-tile: 24 KB
#ifndef XH_STAGES
#define XH_STAGES 3                                                  // ring depth of the f16x3 form (16 KB per stage); lab: tools/lab/conv_stages.sh
#endif

// the same four channels into an LDS staging row whose planes are 512 B apart (blur_act_xp_kernel / blur_act_dma_kernel)
template <bool H2>
__device__ __forceinline__ void stage_xplanes4(const f32x4& v, char* st) {
    if (H2) {
        unsigned h[2], l[2];
        split2h(v[0], v[1], XH_SCALE_X, h[0], l[0]);
        split2h(v[2], v[3], XH_SCALE_X, h[1], l[1]);
        *reinterpret_cast<u32x2*>(st) = (u32x2){h[0], h[1]};
        *reinterpret_cast<u32x2*>(st + 512) = (u32x2){l[0], l[1]};
    } else {
        unsigned h[2], m[2], l[2];
        split3(v[0], v[1], h[0], m[0], l[0]);
        split3(v[2], v[3], h[1], m[1], l[1]);
        *reinterpret_cast<u32x2*>(st) = (u32x2){h[0], h[1]};
        *reinterpret_cast<u32x2*>(st + 512) = (u32x2){m[0], m[1]};
        *reinterpret_cast<u32x2*>(st + 1024) = (u32x2){l[0], l[1]};
    }
}

// Epilogue of modconv_xp_kernel, through LDS: the wave parks one 32 x 64 accumulator block at a time in its own
// 8.5 KB of (now idle) stage memory and reads it back as 16 pixels x one 16-channel chunk, 4 channels per lane — 16-B
// fp32 stores instead of 64 dword stores per lane, the demod / noise / bias / activation on float4, and,
// when the next conv reads x planes, that conv's modulated input split RIGHT HERE (out * s_next -> three bf16
// planes, 8 B per lane and plane): the activation is split once, by its producer, not per tap and Cout tile.
constexpr int EPI_LDW = 68;
// four consecutive channels of one pixel into the x planes of chunk `chunk` at element offset `off` of a (chunk, plane) image
// of `oplane` elements: three bf16 planes, or (h2) two fp16 planes of 8 v — 8 B per plane
__device__ __forceinline__ void store_xplanes4(const f32x4& v, int h2, unsigned short* __restrict__ xp, long chunk, long oplane,
                                               long off) {
    if (h2) {
        unsigned h[2], l[2];
        split2h(v[0], v[1], XH_SCALE_X, h[0], l[0]);
        split2h(v[2], v[3], XH_SCALE_X, h[1], l[1]);
        unsigned short* d = xp + chunk * 2 * oplane + off;
        *reinterpret_cast<u32x2*>(d) = (u32x2){h[0], h[1]};
        *reinterpret_cast<u32x2*>(d + oplane) = (u32x2){l[0], l[1]};
    } else {
        unsigned h[2], m[2], l[2];
        split3(v[0], v[1], h[0], m[0], l[0]);
        split3(v[2], v[3], h[1], m[1], l[1]);
        unsigned short* d = xp + chunk * 3 * oplane + off;
        *reinterpret_cast<u32x2*>(d) = (u32x2){h[0], h[1]};
        *reinterpret_cast<u32x2*>(d + oplane) = (u32x2){m[0], m[1]};
        *reinterpret_cast<u32x2*>(d + 2 * oplane) = (u32x2){l[0], l[1]};
    }
}
// staging writers: one 32-row half (mt) of a wave's 64 x 64 accumulators -> S [32][EPI_LDW]
template <int MT = 2>
struct StageAcc32 {          // MT x 2 tiles of v_mfma_f32_32x32x16_*: lane = column, 16 registers = rows
    static constexpr int NMT = MT;
    const f32x16 (&acc)[MT][2];
    __device__ __forceinline__ void operator()(int mt, float* S, int lane) const {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                S[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LDW + nt * 32 + (lane & 31)] = acc[mt][nt][r];
    }
};
struct StageAcc16 {          // 4 x 4 tiles of v_mfma_f32_16x16x32_bf16: column = lane & 15, row = 4 (lane >> 4) + register
    static constexpr int NMT = 2;
    const f32x4 (&acc)[4][4];
    __device__ __forceinline__ void operator()(int mt, float* S, int lane) const {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    S[(rb * 16 + 4 * (lane >> 4) + j) * EPI_LDW + cb * 16 + (lane & 15)] = acc[2 * mt + rb][cb][j];
    }
};
template <class STAGE>
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& g, const ConvSub& sb, const STAGE& stage_write, char* lds,
                                                  unsigned row0, int col0, int wid, int lane) {
    constexpr int LDW = EPI_LDW;                                          // floats per LDS row (64 + 4: the 16-row column reads
    float* S = reinterpret_cast<float*>(lds) + wid * 32 * LDW;       //  below then fall in distinct banks)
    const int OH = sb.OH, OW = sb.OW;
    const unsigned ohw = (unsigned)OH * (unsigned)OW, Mtot = (unsigned)g.B * ohw;
    const float nw = (g.noise && g.noise_w) ? *g.noise_w : 0.f;
    const long oplane = (long)g.B * (g.outH + 2) * (g.outW + 2) * 16;   // bf16 elements of one output (chunk, plane)
    // a pass = 16 pixel rows x ONE 16-channel chunk: lane -> row (lane >> 2), channel quad (lane & 3); its store into a
    // (chunk, plane) image is one contiguous 512-B run when the 16 pixels are neighbours in an image row
    const int q4 = (lane & 3) * 4, lrow = lane >> 2;
#pragma unroll
    for (int mt = 0; mt < STAGE::NMT; ++mt) {
        stage_write(mt, S, lane);
        // (same wave wrote and reads: program order + the LDS counter, no barrier)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) {
            const unsigned mrow = row0 + mt * 32 + rg * 16 + lrow;
            const bool rok = mrow < Mtot;
            const unsigned mm = rok ? mrow : 0u;
            const unsigned b = mm / ohw, rem = mm - b * ohw;
            const int oy = (int)(rem / (unsigned)OW), ox = (int)(rem - (unsigned)oy * OW);
            const int y = oy * g.ymul + sb.yadd, x = ox * g.xmul + sb.xadd;
            const long pix = ((long)b * g.outH + y) * g.outW + x;
            const float nz = (g.noise && rok) ? nw * g.noise[(long)b * g.noise_bstride + (long)y * g.outW + x] : 0.f;
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const int c4 = col0 + ch * 16 + q4;
                const f32x4 a = *reinterpret_cast<const f32x4*>(S + (rg * 16 + lrow) * LDW + ch * 16 + q4);
                if (!rok || c4 >= g.Cout) continue;                  // host: Cout % 4 == 0
                f32x4 dm4 = {1.f, 1.f, 1.f, 1.f}, bias4 = zero4();
                if (g.demod) dm4 = *reinterpret_cast<const f32x4*>(g.demod + (long)b * g.Cout + c4);
                dm4 *= g.acc_scale;
                if (g.bias) bias4 = *reinterpret_cast<const f32x4*>(g.bias + c4);
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = fmaf(a[q], dm4[q], nz + bias4[q]);
                    v[q] = g.act ? lrelu_sqrt2(t) : t;
                }
                *reinterpret_cast<f32x4*>(g.out + pix * g.Cout + c4) = v;
                if (g.xp_out) {
                    if (g.s_next) v *= *reinterpret_cast<const f32x4*>(g.s_next + (long)b * g.Cout + c4);
                    const long ppix = ((long)b * (g.outH + 2) + y + 1) * (g.outW + 2) + x + 1;
                    store_xplanes4(v, g.h2, g.xp_out, (long)(c4 >> 4), oplane, ppix * 16 + (c4 & 15));
                }
            }
        }
    }
}

// 16x16x32 form of the six products: two planes side by side fill the K = 32 of one v_mfma_f32_16x16x32_bf16, so
//   [a_h | a_l] x [b_l | b_h] = a_h b_l + a_l b_h,   [a_h | a_m] x [b_m | b_h] = a_h b_m + a_m b_h,
//   [a_h | a_m] x [b_h | b_m] = a_h b_h + a_m b_m
// are three MFMAs (16 cycles each: the same pipe time as six 32x32x16 products per 32 x 32 outputs).  Lane l holds row /
// column l & 15 and the 8 k of group l >> 4: groups 0, 1 come from the first plane, 2, 3 from the second.  The LDS image
// is half-major here ([2 halves][128 rows][16 B] per plane): any 16 rows of a fragment read fall in 16 distinct slots.
template <int ROWS = 128>          // rows of the (half-major) image: [2 halves][ROWS][16 B] per plane
__device__ __forceinline__ bf16x8 xp16_frag(const char* S, int px, int py, int row, int lane) {
    const int kq = lane >> 4;
    return *reinterpret_cast<const bf16x8*>(S + (kq < 2 ? px : py) * (ROWS * 32) + (kq & 1) * (ROWS * 16) + row * 16);
}
template <int AROWS = 128>         // 256: the tall tile of narrow layers (four waves stacked along M, all on the same 64 columns)
__device__ __forceinline__ void xp16_mfma_ktile(const char* As, const char* Bs, f32x4 (&acc)[4][4], int wr, int wc, int lane) {
    const int ar = wr * 64 + (lane & 15), br = wc * 64 + (lane & 15);
    bf16x8 ahl[4], ahm[4], blh[4], bmh[4], bhm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { ahl[i] = xp16_frag<AROWS>(As, 0, 2, ar + 16 * i, lane); blh[i] = xp16_frag(Bs, 2, 0, br + 16 * i, lane); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { ahm[i] = xp16_frag<AROWS>(As, 0, 1, ar + 16 * i, lane); bmh[i] = xp16_frag(Bs, 1, 0, br + 16 * i, lane); }
#pragma unroll
    for (int i = 0; i < 4; ++i) bhm[i] = xp16_frag(Bs, 0, 1, br + 16 * i, lane);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahl[rb], blh[cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[rb], bmh[cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
            acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ahm[rb], bhm[cb], acc[rb][cb], 0, 0, 0);
}

// TALL (narrow layers, Cout <= 64: the 512^2 layers of a 512^2 generator): the workgroup's tile is 256 pixels x 64 channels —
// four waves stacked along M — instead of 128 x 128 with the two right-hand waves multiplying padding.  The A image of a
// k-tile then has 256 rows (two DMA instructions per plane and thread: rows tid & 127 and + 128), the B image is the same
// 128-row weight block (its upper 64 rows are the Cout padding, never read).
template <int NT, bool M16, bool TALL, bool H2, class ACC>
__device__ __forceinline__ void conv_mainloop_xp(const ConvArgs& g, const ConvSub& sb, char* lds, ACC& acc,
                                                 unsigned pixbyte, unsigned pixbyte2, int n0, int tid, int wr, int wc, int lane) {
    constexpr int NPL = H2 ? 2 : 3;                                  // planes per operand
    constexpr int APL = TALL ? 8192 : 4096;                          // bytes of one A plane of a k-tile
    constexpr int BOFF = NPL * APL;                                  // the B planes follow the A planes
    constexpr int STAGE = BOFF + NPL * 4096;
    const int ncc = g.wtp_ncc;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rs_wp = make_rsrc(reinterpret_cast<const float*>(g.wtp), g.wtp_bytes);
    const int wp_ntn = (g.Cout + 127) >> 7;
    const int IWp = g.IW + 2;
    int tapoff[NT];                                                  // byte offset of the tap's padded pixel
    unsigned wblk[NT];                                               // weight block of (tap, Cout tile, chunk 0)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        tapoff[t] = (sb.dy[t] * IWp + sb.dx[t]) * 32;
        wblk[t] = (unsigned)(((sb.tw[t] * wp_ntn + (n0 >> 7)) * ncc) * NPL) * 4096u;
    }
    // this thread's 16 bytes of a pre-split weight plane (stored in the half-swap layout): linear for the 32x32 image,
    // (row tid & 127, half tid >> 7) for the half-major image
    const unsigned wlane = M16 ? (unsigned)x_kc_off(tid & 127, tid >> 7) : (unsigned)tid * 16u;
    const char* xp = reinterpret_cast<const char*>(g.xp);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;

    // 6 LDS-DMA instructions per thread.  with_a: lab only (tools/lab/conv_abl.sh: -DXP_ABL_A2 / -DXP_ABL_B2 fetch the x planes /
    // the weight planes for 4 of the 9 taps — what sharing a halo tile among the taps would save: 3 % / 4 %)
    auto issue = [&](int cc, int to, unsigned wb, char* dst, bool with_a = true) {
        const unsigned va = pixbyte + (unsigned)to;                  // signed add: the padded pixel is never negative
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#if defined(XP_ABL_A2) || defined(XP_ABL_A3)
            if (!with_a) break;
#endif
            const char* base = xp + (long)(cc * NPL + p) * (long)g.xp_plane_bytes;
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)g.xp_plane_bytes, 0x00020000);
            if constexpr (TALL && H2) {          // KC image of 256 rows: rows tid >> 1 and + 128, 4 KB apart
                char* d = dst + p * APL + wid * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)d, 16, (int)va, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(d + 4096), 16, (int)(pixbyte2 + (unsigned)to), 0, 0, 0);
            } else if constexpr (TALL) {
                char* d = dst + p * APL + (wid >> 1) * 4096 + (wid & 1) * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)d, 16, (int)va, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(d + 2048), 16, (int)(pixbyte2 + (unsigned)to), 0, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + p * 4096 + wid * 1024), 16, (int)va, 0, 0, 0);
            }
        }
        const unsigned blk = wb + (unsigned)cc * (unsigned)NPL * 4096u;
#ifdef XP_ABL_B2
        if (!with_a) return;
#endif
#pragma unroll
        for (int p = 0; p < NPL; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wp, (lds_ptr_t)(dst + BOFF + p * 4096 + wid * 1024), 16,
                                                     (int)wlane, (int)(blk + p * 4096u), 0, 0);
    };

    if constexpr (H2) {
        // f16x3: a k-tile is 12 MFMAs per wave — half the bf16x6 tile — and 16 KB of LDS, so the ring is XH_STAGES deep:
        // the DMAs of tiles k+1 .. k+NS-2 stay in flight across the barrier that retires tile k (counted vmcnt: a thread
        // issues PER = 4 DMAs per tile, in tile order), tile k+NS-1 is issued into the slot tile k-1 left.  Raw barrier: a
        // __syncthreads() would drain the DMAs of the tiles ahead.
        constexpr int NS = XH_STAGES, PER = (TALL ? 3 : 2) * NPL;      // DMAs per thread and tile: A (x 2 rows when tall) + B
        static_assert(NS >= 2 && NS <= 4, "ring depth");
        const int total = ncc * NT;
#pragma unroll
        for (int t = 0; t < NS - 1; ++t)
            if (t < total) issue(t / NT, tapoff[t % NT], wblk[t % NT], lds + t * STAGE);
        int rd = 0, ws = NS - 1, k = 0;                                  // ring slots being read / written
        for (int cc = 0; cc < ncc; ++cc) {
#pragma unroll
            for (int tap = 0; tap < NT; ++tap, ++k) {
                const int ahead = min(NS - 2, total - 1 - k);           // newer tiles whose DMAs may still be in flight
#ifdef XP_ABL_A3
                // lab (tools/lab/conv_abl.sh -DXP_ABL_A3; WRONG values): the x planes fetched for ONE of every three taps — the
                // upper bound of what a halo row shared by the dx = -1, 0, 1 taps would save; the counted waits follow
                static_assert(NS == 3, "the ablation's wait counts are written for a ring of three");
                constexpr int PB = NPL;                                 // a B-only tile's DMAs
                const int c1 = (NT < 9 || ((tap + 1) % NT) % 3 == 0) ? PER : PB;
                if (ahead >= 1) { if (c1 == PER) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB) : "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
                else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const int nt_ = (tap + NS - 1) % NT, dc_ = (tap + NS - 1) / NT;      // (constants once the tap loop is unrolled)
#ifdef XP_ABL_A3
                if (k + NS - 1 < total) issue(cc + dc_, tapoff[nt_], wblk[nt_], lds + ws * STAGE, NT < 9 || nt_ % 3 == 0);
#else
                if (k + NS - 1 < total) issue(cc + dc_, tapoff[nt_], wblk[nt_], lds + ws * STAGE);
#endif
                const char* As = lds + rd * STAGE;
                xh_mfma_tile<TALL ? 4 : 2, 2, TALL ? 256 : 128, 128>(As, As + BOFF, acc, wr * (TALL ? 128 : 64), wc * 64, lane);
                rd = (rd + 1 == NS) ? 0 : rd + 1;
                ws = (ws + 1 == NS) ? 0 : ws + 1;
            }
        }
        return;
    }
    // Two stages: the DMAs of tile k+1 are issued right after the barrier that retires tile k and have the 24 MFMAs
    // of tile k to land; three workgroups per CU (48 KB of LDS, 118 VGPRs) cover what that does not.  (A three-stage
    // ring with a counted vmcnt(6) and a raw s_barrier measured 3.5 % SLOWER: 72 KB of LDS = two workgroups per CU.)
    if (ncc > 0) issue(0, tapoff[0], wblk[0], lds);
    int buf = 0;
    for (int cc = 0; cc < ncc; ++cc) {
        const bool last_cc = (cc + 1 == ncc);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            // tile k has landed (this wave's DMAs: vmcnt; the other waves': the barrier), and every wave is done
            // reading the other stage (its MFMAs of tile k-1 were issued before it arrived here)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            char* As = lds + buf * STAGE;
            const bool wrap = (tap + 1 == NT);
            if (!(wrap && last_cc))
                issue(wrap ? cc + 1 : cc, tapoff[wrap ? 0 : tap + 1], wblk[wrap ? 0 : tap + 1],
                      lds + (buf ^ 1) * STAGE, NT < 9 || wrap || tap < 1 || (tap + 1) % 3 == 0);
            if constexpr (M16) xp16_mfma_ktile<TALL ? 256 : 128>(As, As + BOFF, acc, wr, wc, lane);
            else if constexpr (!H2) x6_mfma_ktile<true, true>(As, As + BOFF, acc, wr, wc, lane);
            buf ^= 1;
        }
    }
}

template <bool M16, bool TALL = false, bool H2 = false>
__global__ __launch_bounds__(GEMM_THREADS, 2) void modconv_xp_kernel(ConvArgs g) {
    static_assert(!TALL || M16 || H2, "the tall tile: the 16x16x32 product form of bf16x6 (256 x 64), or f16x3 (256 x 128)");
    static_assert(!H2 || !M16, "the two-plane fp16 engine uses the 32x32x16 products");
    constexpr int TBM = TALL ? 2 * BM : BM;                          // pixels per tile
    // two stages of (A planes + B planes); never less than the epilogue's 4 x 32 x EPI_LDW floats
    constexpr int LDS_MAIN = (H2 ? XH_STAGES : 2) * ((H2 ? 2 : 3) * ((TALL ? 8192 : 4096) + 4096));   // f16x3 tall: 3 x 24 KB
    constexpr int LDS_EPI = 4 * 32 * EPI_LDW * 4;
    __shared__ __attribute__((aligned(16))) char lds_raw[LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI];
    // waves: 2 x 2 over the tile (f16x3 tall: each 128 x 64 of 256 x 128); bf16x6 tall: four stacked along M on the same 64 columns
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = (TALL && !H2) ? wid : (wid >> 1), wc = (TALL && !H2) ? 0 : (wid & 1);
    int si = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < g.nsub && (int)blockIdx.x >= g.sub[i].tile_base) si = i;
    const ConvSub& sb = g.sub[si];
    const int local = xcd_remap(blockIdx.x - sb.tile_base, sb.tiles_m * g.tiles_n);
    const int tm = local / g.tiles_n, tn = local - tm * g.tiles_n;
    const int m0 = tm * TBM, n0 = tn * BN;
    const unsigned ohw = (unsigned)sb.OH * (unsigned)sb.OW;
    const unsigned Mtot = (unsigned)g.B * ohw;
    // this thread DMAs bytes [16 tid, 16 tid + 16) of every plane of the A image: row R = tid >> 1, 16-B slot tid & 1,
    // which under the half-swap swizzle holds channel half (tid & 1) ^ ((R >> 3) & 1) of the row's 16 channels
    // (half-major image of the 16x16x32 form: row tid & 127, channel half tid >> 7, no swizzle; tall tile: also row + 128)
    const int R = M16 ? (tid & 127) : (tid >> 1);
    auto pixbyte_of = [&](unsigned m) -> unsigned {
        if (m >= Mtot) m = Mtot - 1;                                 // tail rows: any valid pixel (outputs are dropped)
        const unsigned b = m / ohw, rem = m - b * ohw;
        const unsigned oy = rem / (unsigned)sb.OW, ox = rem - oy * (unsigned)sb.OW;
        const unsigned pixpad = (b * (unsigned)(g.IH + 2) + oy + 1u) * (unsigned)(g.IW + 2) + ox + 1u;
        return pixpad * 32u + (unsigned)((M16 ? (tid >> 7) : ((tid & 1) ^ ((R >> 3) & 1))) << 4);
    };
    const unsigned pixbyte = pixbyte_of((unsigned)m0 + (unsigned)R);
    const unsigned pixbyte2 = TALL ? pixbyte_of((unsigned)m0 + 128u + (unsigned)R) : 0u;

    if constexpr (M16) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = zero4();
        switch (sb.ntaps) {
            case 9: conv_mainloop_xp<9, true, TALL, false>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
            case 4: conv_mainloop_xp<4, true, TALL, false>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
            case 2: conv_mainloop_xp<2, true, TALL, false>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
            default: conv_mainloop_xp<1, true, TALL, false>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
        }
        __syncthreads();                                             // every wave has read its last fragments
        conv_epilogue_lds(g, sb, StageAcc16{acc}, lds_raw, (unsigned)m0 + wr * 64, n0 + wc * 64, wid, lane);
    } else {
        constexpr int MT = TALL ? 4 : 2;                             // 32-row MFMA tiles of a wave along M
        f32x16 acc[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        switch (sb.ntaps) {
            case 9: conv_mainloop_xp<9, false, TALL, H2>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
            case 4: conv_mainloop_xp<4, false, TALL, H2>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
            case 2: conv_mainloop_xp<2, false, TALL, H2>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
            default: conv_mainloop_xp<1, false, TALL, H2>(g, sb, lds_raw, acc, pixbyte, pixbyte2, n0, tid, wr, wc, lane); break;
        }
        if ((g.Cout & 3) == 0) {                                     // epilogue through LDS (+ the next layer's planes)
            __syncthreads();                                         // every wave has read its last fragments
            conv_epilogue_lds(g, sb, StageAcc32<MT>{acc}, lds_raw, (unsigned)m0 + wr * (32 * MT), n0 + wc * 64, wid, lane);
        } else {
            conv_epilogue<MT>(g, sb, acc, (unsigned)m0 + wr * (32 * MT), n0 + wc * 64, lane);
        }
    }
}

// x [B][H][W][Cin] fp32 NHWC times s [B][Cin] (or s = NULL) -> xp (layout above).  A thread owns 4 channels of a
// pixel (float4 load, 8 B per plane); the channel padding up to ncc*16 is written as zeros, the spatial border is not
// touched (the host zeroes the buffer once).
__global__ void modsplit_kernel(const float* __restrict__ x, const float* __restrict__ s, int B, int H, int W, int Cin,
                                int ncc, int h2, unsigned short* __restrict__ xp) {
    const int q4 = ncc * 4;
    const long total = (long)B * H * W * q4;
    const long plane = (long)B * (H + 2) * (W + 2) * 16;            // bf16 elements of one (chunk, plane) image
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % q4) * 4;
        long pix = i / q4;
        const int xx = (int)(pix % W); pix /= W;
        const int y = (int)(pix % H);
        const int b = (int)(pix / H);
        f32x4 v = zero4();
        if (c < Cin) {
            const long src = (((long)b * H + y) * W + xx) * Cin + c;
            if (c + 4 <= Cin && (Cin & 3) == 0) {
                v = *reinterpret_cast<const f32x4*>(x + src);
                if (s) v *= *reinterpret_cast<const f32x4*>(s + (long)b * Cin + c);
            } else {
                for (int e = 0; e < 4 && c + e < Cin; ++e) v[e] = x[src + e] * (s ? s[(long)b * Cin + c + e] : 1.f);
            }
        }
        const long ppix = ((long)b * (H + 2) + y + 1) * (W + 2) + xx + 1;
        store_xplanes4(v, h2, xp, (long)(c >> 4), plane, ppix * 16 + (c & 15));
    }
}

// Split-K tail: out = epilogue(sum_s slab[s]) over the whole NHWC output (fixed order).
__global__ void conv_splitk_epilogue_kernel(const float* __restrict__ slab, int splits, long total, int B, int outH,
                                            int outW, int Cout, const float* __restrict__ demod,
                                            const float* __restrict__ noise, long noise_bstride,
                                            const float* __restrict__ noise_w, const float* __restrict__ bias, int act,
                                            float* __restrict__ out) {
    const float nw = (noise && noise_w) ? *noise_w : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += slab[(long)z * total + i];
        const int co = (int)(i % Cout);
        const long pix = i / Cout;
        const int b = (int)(pix / ((long)outH * outW));
        const long yx = pix - (long)b * outH * outW;
        if (demod) v *= demod[(long)b * Cout + co];
        if (noise) v += nw * noise[(long)b * noise_bstride + yx];
        if (bias) v += bias[co];
        out[i] = act ? lrelu_sqrt2(v) : v;
    }
}

// demod'[b][co] = scale * rsqrt(scale^2 * sum_ci s[b][ci]^2 * wsq[co][ci] + 1e-8)   (one wave each)
__global__ void demod_kernel(const float* __restrict__ s, const float* __restrict__ wsq, int B, int Cin, int Cout,
                             float scale, float* __restrict__ demod) {
    const long idx = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= (long)B * Cout) return;
    const int b = (int)(idx / Cout), co = (int)(idx - (long)b * Cout), lane = threadIdx.x & 63;
    float a = 0.f;
    for (int ci = lane; ci < Cin; ci += 64) { const float sv = s[(long)b * Cin + ci]; a = fmaf(sv * sv, wsq[(long)co * Cin + ci], a); }
    a = gx_wave_sum(a);
    if (lane == 0) demod[idx] = scale * rsqrtf(scale * scale * a + 1e-8f);
}

// max |x| over an [N][K] matrix (row stride ld) into *out as float bits (non-negative floats order like unsigned integers;
// *out zeroed by the caller): the per-tensor scale of the f16x3 weight planes
__global__ void absmax_kernel(const float* __restrict__ x, int N, int K, long ld, unsigned* __restrict__ out) {
    __shared__ float red[16];
    const long n = (long)N * K;
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(x[(i / K) * ld + (i % K)]));
    m = gx_block_max(m, red);
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(m));
}

// Pre-split conv weights for the bf16x6 engine, stored in the order the kernel consumes them: for every
// (tap, 128-row Cout tile, 16-channel chunk) one 12 KB block = the three bf16 planes (h, m, l with
// w = h + m + l exactly) of that weight tile, each plane already in the LDS image layout (32-B rows with
// the half-swap swizzle).  A workgroup copies a block linearly: thread t moves bytes [16 t, 16 t + 16) of each plane.
__global__ void split_conv_weights_kernel(const float* __restrict__ wt, int taps, int Cout, int Cin, long ld, int ntn,
                                          int ncc, int h2, const float* __restrict__ amax, unsigned short* __restrict__ planes) {
    // h2: two fp16 planes of scale * w per block (8 KB blocks) for the f16x3 forms; scale = 16 (conv weights) or, with `amax`
    // (device: the maximum magnitude of wt), xh_scale_of(*amax) (gx_split_gemm_weights_h2)
    const float wscale = amax ? xh_scale_of(*amax) : XH_SCALE_W;
    const long total = (long)taps * ntn * ncc * 128 * 16;          // padded element count
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i & 15), row = (int)((i >> 4) & 127);
        long blk = i >> 11;
        const int cc = (int)(blk % ncc); blk /= ncc;
        const int tn = (int)(blk % ntn); const int tap = (int)(blk / ntn);
        const int co = tn * 128 + row, ci = cc * 16 + k;
        const float x = (co < Cout && ci < Cin) ? wt[((long)tap * Cout + co) * ld + ci] : 0.f;
        const int off = (x_kc_off(row, k >> 3) >> 1) + (k & 7);
        if (h2) {
            unsigned h, l;
            split2h(x, 0.f, wscale, h, l);
            unsigned short* base = planes + ((((long)tap * ntn + tn) * ncc + cc) * 2) * 2048;
            base[off] = (unsigned short)(h & 0xffffu);
            base[2048 + off] = (unsigned short)(l & 0xffffu);
            continue;
        }
        unsigned h, m, l;
        split3(x, 0.f, h, m, l);
        unsigned short* base = planes + ((((long)tap * ntn + tn) * ncc + cc) * 3) * 2048;      // 2048 bf16 per plane
        base[off] = (unsigned short)(h & 0xffffu);
        base[2048 + off] = (unsigned short)(m & 0xffffu);
        base[4096 + off] = (unsigned short)(l & 0xffffu);
    }
}

// Latents of one view of a batch (swav_clustering.py:599-640 + image_augmentor.py:42-53 + the truncation of
// model.py:594-602): lat[b][l][:] = keep * (m + psi (w_b - m)) + mix[b][l][0] nw[b][0][:] + mix[b][l][1] nw[b][1][:]
// with keep = 1 - mix[b][l][0] - mix[b][l][1]  (mix is sigma at the two perturbed latents of image b, else 0).
__global__ void view_latents_kernel(const float* __restrict__ w, const float* __restrict__ mean, float psi,
                                    const float* __restrict__ nw, const float* __restrict__ mix, int B, int NL, int S,
                                    float* __restrict__ lat) {
    const long total = (long)B * NL * S;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % S);
        const long bl = i / S;
        const int b = (int)(bl / NL);
        const float m0 = mix[bl * 2], m1 = mix[bl * 2 + 1];
        const float mu = mean[c];
        const float base = fmaf(psi, w[(long)b * S + c] - mu, mu);
        lat[i] = (1.f - (m0 + m1)) * base + m0 * nw[((long)b * 2) * S + c] + m1 * nw[((long)b * 2 + 1) * S + c];
    }
}

// ---- all layers of a generator forward at once -----------------------------------------------
// style_pick: S_all [B*n_latent][total] (every latent through every layer's affine, one GEMM) ->
// packed per-layer styles: layer j occupies out[B*off_j .. B*(off_j+cin_j)) as [B][cin_j].
// table rows: {latent index, column offset, cin}.
__global__ void style_pick_kernel(const float* __restrict__ S_all, int B, int n_latent, int total,
                                  const int* __restrict__ table, int nlayer, float* __restrict__ out) {
    const long n = (long)B * total;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        // i enumerates the packed output: find the layer whose slab [B*off, B*(off+cin)) holds i
        int j = 0;
        while (j + 1 < nlayer && i >= (long)B * table[3 * (j + 1) + 1]) ++j;
        const int idx = table[3 * j], off = table[3 * j + 1], cin = table[3 * j + 2];
        const long r = i - (long)B * off;
        const int b = (int)(r / cin), c = (int)(r - (long)b * cin);
        out[i] = S_all[((long)b * n_latent + idx) * total + off + c];
    }
}
// demod_batch: one wave per (layer, b, co).  table rows: {style offset (= column offset), cin, cout,
// wsq offset, demod offset (in units of B rows: out index = B*doff + b*cout + co)}; scale[j] floats.
__global__ void demod_batch_kernel(const float* __restrict__ s_packed, const float* __restrict__ wsq_all, int B,
                                   const int* __restrict__ table, const float* __restrict__ scale, int nconv,
                                   long total_waves, float* __restrict__ demod) {
    const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= total_waves) return;
    int j = 0;
    while (j + 1 < nconv && w >= (long)B * table[5 * (j + 1) + 4]) ++j;
    const int soff = table[5 * j], cin = table[5 * j + 1], cout = table[5 * j + 2], woff = table[5 * j + 3],
              doff = table[5 * j + 4];
    const long r = w - (long)B * doff;
    const int b = (int)(r / cout), co = (int)(r - (long)b * cout), lane = threadIdx.x & 63;
    const float* s = s_packed + (long)B * soff + (long)b * cin;
    const float* wq = wsq_all + woff + (long)co * cin;
    float a = 0.f;
    for (int ci = lane; ci < cin; ci += 64) { const float sv = s[ci]; a = fmaf(sv * sv, wq[ci], a); }
    a = gx_wave_sum(a);
    const float sc = scale[j];
    if (lane == 0) demod[w] = sc * rsqrtf(sc * sc * a + 1e-8f);
}

// style_norm: one workgroup per (styled conv j, sample b) of demod_batch's table.  The conv is linear in the style, so
//   demod[b][co] * conv(x * s[b]) = (demod[b][co] * c) * conv(x * (s[b] / c))   for any c;
// with c = the power of two that brings max_ci |s[b][ci]| into [2, 4) both scalings are exact and what the convs see of a
// style is a vector of magnitude < 4 whatever the checkpoint's affine layers produce (trained StyleGAN2 styles reach the
// hundreds): the f16x3 planes' fixed range then bounds the ACTIVATIONS alone (|x| < 4094).  s and demod are scaled in place.
__global__ __launch_bounds__(256) void style_norm_kernel(float* __restrict__ s_packed, float* __restrict__ demod, int B,
                                                         const int* __restrict__ table) {
    __shared__ float red[16];
    const int j = blockIdx.x / B, b = blockIdx.x - j * B;
    const int soff = table[5 * j], cin = table[5 * j + 1], cout = table[5 * j + 2], doff = table[5 * j + 4];
    float* s = s_packed + (long)B * soff + (long)b * cin;
    float m = 0.f;
    for (int ci = threadIdx.x; ci < cin; ci += 256) m = fmaxf(m, fabsf(s[ci]));
    m = gx_block_max(m, red);
    const int be = (int)((__float_as_uint(m) >> 23) & 0xffu);          // m in [2^(be - 127), 2^(be - 126))
    if (be == 0 || be == 255) return;                                   // zero / subnormal / not finite: left alone
    int e = be - 128;                                                   // m 2^-e in [2, 4)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    if (e == 0) return;
    const float down = __uint_as_float((unsigned)(127 - e) << 23), up = __uint_as_float((unsigned)(127 + e) << 23);
    for (int ci = threadIdx.x; ci < cin; ci += 256) s[ci] *= down;
    float* d = demod + (long)B * doff + (long)b * cout;
    for (int co = threadIdx.x; co < cout; co += 256) d[co] *= up;
}

// Blur of the (2H+1)^2 transposed-conv image with pad (1,1) + StyledConv epilogue:
//   out[b,y,x,c] = act( demod[b,c] * sum_{fy,fx} kf[fy][fx] * T[b, y+fy-1, x+fx-1, c] + nw*noise + bias[c] )
// kf is the FLIPPED 4x4 FIR (upfirdn2d correlates with the flipped kernel, model.py:88).
// A thread owns a (x, 4-channel) column strip of BLUR_ROWS output rows and slides down it: each
// input row is loaded once (4 float4) and scattered into the 4 outputs in flight, instead of 16
// loads per output.  Accumulator slots are compile-time ((j - fy) & 3) so they stay in registers.
constexpr int BLUR_ROWS = 16;
__global__ void blur_act_kernel(const float* __restrict__ T, int B, int TH, int TW, int C, const float* __restrict__ kf,
                                const float* __restrict__ demod, const float* __restrict__ noise, long noise_bstride,
                                const float* __restrict__ noise_w, const float* __restrict__ bias, int act,
                                float* __restrict__ out, int OH, int OW, const float* __restrict__ s_next,
                                unsigned short* __restrict__ xp_out, int h2) {
    // s_next / xp_out (optional): also write the NEXT conv's modulated input out * s_next as x planes (modconv_xp_kernel)
    const long oplane = (long)B * (OH + 2) * (OW + 2) * 16;
    const int c4n = C >> 2;
    const int nseg = (OH + BLUR_ROWS - 1) / BLUR_ROWS;
    const long total = (long)B * nseg * OW * c4n;
    float k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = kf[i];
    const float nw = (noise && noise_w) ? *noise_w : 0.f;
    // Thread -> (pixel column x, 4 channels).  Default: channels fastest (a wave reads / writes whole pixels).  When the x
    // planes are written too (C % 16 == 0), a wave instead owns 16 consecutive pixels x ONE 16-channel chunk, so that its
    // store into a (chunk, plane) image is 16 x 32 B = one contiguous 512-B run instead of 16 scattered 32-B pieces;
    // consecutive waves take the other chunks of the same pixels (the fp32 side then moves in 64-B pieces).
    const bool pmap = xp_out && (C & 15) == 0;
    const int ncc = C >> 4, nxb = (OW + 15) >> 4;
    const long total_p = (long)B * nseg * nxb * ncc * 64;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (pmap ? total_p : total);
         i += (long)gridDim.x * blockDim.x) {
        int c, x, seg, b;
        if (pmap) {
            long rest = i >> 6;
            const int cc = (int)(rest % ncc); rest /= ncc;
            const int xb = (int)(rest % nxb); rest /= nxb;
            seg = (int)(rest % nseg);
            b = (int)(rest / nseg);
            c = cc * 16 + ((int)i & 3) * 4;
            x = xb * 16 + (((int)i >> 2) & 15);
            if (x >= OW) continue;
        } else {
            c = (int)(i % c4n) * 4;
            long rest = i / c4n;
            x = (int)(rest % OW); rest /= OW;
            seg = (int)(rest % nseg);
            b = (int)(rest / nseg);
        }
        const int y0 = seg * BLUR_ROWS;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, dm4 = {1.f, 1.f, 1.f, 1.f};
        if (bias) bias4 = *reinterpret_cast<const f32x4*>(bias + c);
        if (demod) dm4 = *reinterpret_cast<const f32x4*>(demod + (long)b * C + c);
        f32x4 sn4 = {1.f, 1.f, 1.f, 1.f};
        if (xp_out && s_next) sn4 = *reinterpret_cast<const f32x4*>(s_next + (long)b * C + c);
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* Tb = T + (long)b * TH * TW * C + c;
#pragma unroll 1
        for (int g4 = 0; g4 < BLUR_ROWS / 4 + 1; ++g4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g4 + j;                 // input row y0 - 1 + r
                if (r > BLUR_ROWS + 2) break;
                const int ty = y0 - 1 + r;
                f32x4 v[4];
#pragma unroll
                for (int fx = 0; fx < 4; ++fx) {
                    const int tx = x + fx - 1;
                    v[fx] = (ty >= 0 && ty < TH && tx >= 0 && tx < TW)
                                ? *reinterpret_cast<const f32x4*>(Tb + ((long)ty * TW + tx) * C)
                                : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int fy = 0; fy < 4; ++fy) {          // contributes to output rel row o = r - fy
                    const f32x4 h = v[0] * k[fy * 4 + 0] + v[1] * k[fy * 4 + 1] + v[2] * k[fy * 4 + 2] +
                                    v[3] * k[fy * 4 + 3];
                    acc[(j - fy) & 3] += h;
                }
                const int o = r - 3;                      // this output row is now complete
                const int y = y0 + o;
                if (o >= 0 && o < BLUR_ROWS && y < OH) {
                    const f32x4 a = acc[(j + 1) & 3];
                    const float nz = noise ? nw * noise[(long)b * noise_bstride + (long)y * OW + x] : 0.f;
                    f32x4 ov;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = a[q] * dm4[q] + nz + bias4[q];
                        ov[q] = act ? lrelu_sqrt2(t) : t;
                    }
                    *reinterpret_cast<f32x4*>(out + (((long)b * OH + y) * OW + x) * C + c) = ov;
                    if (xp_out) {
                        ov *= sn4;
                        const long ppix = ((long)b * (OH + 2) + y + 1) * (OW + 2) + x + 1;
                        store_xplanes4(ov, h2, xp_out, (long)(c >> 4), oplane, ppix * 16 + (c & 15));
                    }
                }
                acc[(j + 1) & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// blur_act_kernel for layers that also write the next conv's x planes (C % 64 == 0, OW % 16 == 0): a WORKGROUP owns
// 16 consecutive pixels x 64 channels of a ROWS-row strip, thread t = (pixel t >> 4, channel quad t & 15).  The T
// reads and the fp32 stores are then 256-B runs per pixel (the per-wave "16 pixels x one chunk" map of blur_act_kernel
// moves them in 64-B pieces), and the three bf16 planes of an output row go through LDS ([4 chunks][3 planes][16 px][32 B],
// double-buffered, one barrier per row) so that every (chunk, plane) image still receives one contiguous 512-B run.
#ifndef BLUR_MINB
#define BLUR_MINB 1      // lab: workgroups per CU the register allocation is held to (tools/lab/blur_lb.sh)
#endif
// H2: the planes are the two fp16 planes of the f16x3 conv engine ([4 chunks][2 planes] = 8 runs: one 16-B store per thread)
template <int ROWS, bool H2>   // output rows per strip: 32 on the large layers (vertical halo 35 / 32 instead of 19 / 16), 16 on the small
__global__ __launch_bounds__(256, BLUR_MINB) void blur_act_xp_kernel(const float* __restrict__ T, int B, int TH, int TW, int C,
                                                          const float* __restrict__ kf, const float* __restrict__ demod,
                                                          const float* __restrict__ noise, long noise_bstride,
                                                          const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                          int act, float* __restrict__ out, int OH, int OW,
                                                          const float* __restrict__ s_next, unsigned short* __restrict__ xp_out) {
    constexpr int NPL = H2 ? 2 : 3;
    __shared__ __attribute__((aligned(16))) char stage[2][12 * 512];
    const long oplane = (long)B * (OH + 2) * (OW + 2) * 16;
    const int nseg = (OH + ROWS - 1) / ROWS, nxb = OW >> 4, ncg = C >> 6;
    const int tid = threadIdx.x, px = tid >> 4, c4 = tid & 15;
    float k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = kf[i];
    const float nw = (noise && noise_w) ? *noise_w : 0.f;
    const long nblk = (long)B * nseg * nxb * ncg;
    int par = 0;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        // neighbouring tiles share their 3-pixel halos of T: an XCD takes a contiguous run of tiles, so the halo a
        // neighbour already fetched is found in that XCD's L2 (dispatch order deals consecutive blocks to different XCDs)
        long rest = (nblk == (long)gridDim.x) ? (long)xcd_remap((int)blk, (int)nblk) : blk;
        const int cg = (int)(rest % ncg); rest /= ncg;
        const int xb = (int)(rest % nxb); rest /= nxb;
        const int seg = (int)(rest % nseg);
        const int b = (int)(rest / nseg);
        const int c = cg * 64 + c4 * 4, x = xb * 16 + px, y0 = seg * ROWS;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, dm4 = {1.f, 1.f, 1.f, 1.f}, sn4 = {1.f, 1.f, 1.f, 1.f};
        if (bias) bias4 = *reinterpret_cast<const f32x4*>(bias + c);
        if (demod) dm4 = *reinterpret_cast<const f32x4*>(demod + (long)b * C + c);
        if (s_next) sn4 = *reinterpret_cast<const f32x4*>(s_next + (long)b * C + c);
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* Tb = T + (long)b * TH * TW * C + c;
        auto load_row = [&](int r, f32x4 (&v)[4]) {          // input row y0 - 1 + r, columns x - 1 .. x + 2 (zero outside T)
            const int ty = y0 - 1 + r;
#pragma unroll
            for (int fx = 0; fx < 4; ++fx) {
                const int tx = x + fx - 1;
                v[fx] = (ty >= 0 && ty < TH && tx >= 0 && tx < TW)
                            ? *reinterpret_cast<const f32x4*>(Tb + ((long)ty * TW + tx) * C)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
#pragma unroll 1
        for (int g4 = 0; g4 < ROWS / 4 + 1; ++g4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g4 + j;                 // input row y0 - 1 + r
                if (r > ROWS + 2) break;
                f32x4 v[4];
                load_row(r, v);       // (loading row r + 1 ahead of this row's barrier measured 10 % slower: 16 more registers)
#pragma unroll
                for (int fy = 0; fy < 4; ++fy) {
                    const f32x4 h = v[0] * k[fy * 4 + 0] + v[1] * k[fy * 4 + 1] + v[2] * k[fy * 4 + 2] + v[3] * k[fy * 4 + 3];
                    acc[(j - fy) & 3] += h;
                }
                const int o = r - 3;                      // this output row is now complete
                const int y = y0 + o;
                if (o >= 0 && o < ROWS && y < OH) {  // block-uniform
                    const f32x4 a = acc[(j + 1) & 3];
                    const float nz = noise ? nw * noise[(long)b * noise_bstride + (long)y * OW + x] : 0.f;
                    f32x4 ov;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = a[q] * dm4[q] + nz + bias4[q];
                        ov[q] = act ? lrelu_sqrt2(t) : t;
                    }
                    *reinterpret_cast<f32x4*>(out + (((long)b * OH + y) * OW + x) * C + c) = ov;
                    ov *= sn4;
                    stage_xplanes4<H2>(ov, stage[par] + ((c4 >> 2) * NPL * 16 + px) * 32 + (c4 & 3) * 8);
                    __syncthreads();
                    // 4 NPL runs (chunk, plane) of 512 B = 384 (256) 16-B units
                    const long ppix = ((long)b * (OH + 2) + y + 1) * (OW + 2) + xb * 16 + 1;
#pragma unroll
                    for (int it = 0; it < (H2 ? 1 : 2); ++it) {
                        const int u = it * 256 + tid;
                        if (u < NPL * 128) {
                            const int run = u >> 5, chunk = run / NPL, plane = run - chunk * NPL;
                            const u32x4 d = *reinterpret_cast<const u32x4*>(stage[par] + u * 16);
                            unsigned short* dst = xp_out + ((long)(cg * 4 + chunk) * NPL + plane) * oplane + ppix * 16 + (u & 31) * 8;
                            *reinterpret_cast<u32x4*>(dst) = d;
                        }
                    }
                    par ^= 1;        // the next row writes the other buffer; its barrier orders these reads before the row after
                }
                acc[(j + 1) & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// blur_act_xp_kernel with the T rows arriving by LDS-DMA several rows AHEAD (round 5).  The register version keeps ONE row of
// loads in flight per wave (four 16-byte loads per thread, each T element fetched by four neighbouring threads) and waits for
// it before it can issue the next; a prefetch in registers cost occupancy (16 more VGPRs: 10 % slower).  Here a workgroup's
// 19 (+1) pixel x 64 channel slice of a T row goes into a ring of LDS rows by buffer_load ... lds — no registers, BD_AHEAD rows
// in flight per workgroup — and every thread reads its four neighbours from LDS.  Waits are COUNTED (the stores of the
// finished rows sit between the DMAs in the wave's vector-memory queue: loads, stores and LDS-DMA retire in issue order;
// every wave issues the same two DMAs per T row and three stores per output row, so the counts are uniform), the barriers
// are raw (a __syncthreads() would drain the DMAs of the rows ahead), and the ring is read by inline asm: hipcc cannot prove
// that a read of ring slot r % 7 does not alias the DMA it has just issued into slot (r + 5) % 7 and puts s_waitcnt vmcnt(0)
// in front of every ds_read it sees.  Host: OH % ROWS == 0 (whole strips: every iteration of a strip stores or none does).
#ifndef BD_AHEAD
#define BD_AHEAD 5                             // T rows in flight beyond the one being consumed
#endif
#ifndef BD_ROWS
#define BD_ROWS 32                             // output rows per strip
#endif
constexpr int BD_RING = BD_AHEAD + 2;          // ring rows: the row being read + BD_AHEAD in flight + one the slow waves may still read
constexpr int BD_ROWB = 20 * 64 * 4;           // one ring row: 20 pixels (19 used) x 64 channels, fp32
__device__ __forceinline__ void bd_wait_vm(int n) {            // s_waitcnt takes an immediate; n = 2 (DMAs) + 3 (stores) in flight
    switch (n) {
#define BD_CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
        BD_CASE(1) BD_CASE(2) BD_CASE(3) BD_CASE(4) BD_CASE(5) BD_CASE(6) BD_CASE(7) BD_CASE(8) BD_CASE(9) BD_CASE(10)
        BD_CASE(11) BD_CASE(12) BD_CASE(13) BD_CASE(14) BD_CASE(15) BD_CASE(16) BD_CASE(17) BD_CASE(18) BD_CASE(19) BD_CASE(20)
        BD_CASE(21) BD_CASE(22) BD_CASE(23) BD_CASE(24) BD_CASE(25) BD_CASE(26) BD_CASE(27) BD_CASE(28) BD_CASE(29) BD_CASE(30)
        BD_CASE(31) BD_CASE(32) BD_CASE(33) BD_CASE(34) BD_CASE(35) BD_CASE(36) BD_CASE(37) BD_CASE(38) BD_CASE(39) BD_CASE(40)
#undef BD_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}
__device__ __forceinline__ void bd_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int ROWS, bool H2>
__global__ __launch_bounds__(256, 3) void blur_act_dma_kernel(const float* __restrict__ T, int B, int TH, int TW, int C,
                                                              const float* __restrict__ kf, const float* __restrict__ demod,
                                                              const float* __restrict__ noise, long noise_bstride,
                                                              const float* __restrict__ noise_w, const float* __restrict__ bias,
                                                              int act, float* __restrict__ out, int OH, int OW,
                                                              const float* __restrict__ s_next, unsigned short* __restrict__ xp_out) {
    constexpr int NPL = H2 ? 2 : 3;                   // planes per chunk
    constexpr int NST = H2 ? 2 : 3;                   // stores per wave and output row: the fp32 row + 1 (2) of plane units
    __shared__ __attribute__((aligned(16))) char ring[BD_RING * BD_ROWB];
    __shared__ __attribute__((aligned(16))) char stage[2][12 * 512];
    __shared__ float nzs[ROWS * 16];
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    typedef __attribute__((address_space(3))) char* lds_char_t;
    const long oplane = (long)B * (OH + 2) * (OW + 2) * 16;
    const int nseg = OH / ROWS, nxb = OW >> 4, ncg = C >> 6;
    const int tid = threadIdx.x, px = tid >> 4, c4 = tid & 15, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float k[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) k[i] = kf[i];
    const float nw = (noise && noise_w) ? *noise_w : 0.f;
    const long nblk = (long)B * nseg * nxb * ncg;
    const unsigned rowbytes = (unsigned)TW * (unsigned)C * 4u;
    const unsigned ring_lds = (unsigned)(size_t)(lds_char_t)ring;          // LDS byte address of the ring
    const unsigned rd_off = (unsigned)((px * 16 + c4) * 16);               // this thread's first neighbour (pixel x - 1) in a ring row
    int par = 0;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        long rest = (nblk == (long)gridDim.x) ? (long)xcd_remap((int)blk, (int)nblk) : blk;
        const int cg = (int)(rest % ncg); rest /= ncg;
        const int xb = (int)(rest % nxb); rest /= nxb;
        const int seg = (int)(rest % nseg);
        const int b = (int)(rest / nseg);
        const int c = cg * 64 + c4 * 4, x0 = xb * 16, x = x0 + px, y0 = seg * ROWS;
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, dm4 = {1.f, 1.f, 1.f, 1.f}, sn4 = {1.f, 1.f, 1.f, 1.f};
        if (bias) bias4 = *reinterpret_cast<const f32x4*>(bias + c);
        if (demod) dm4 = *reinterpret_cast<const f32x4*>(demod + (long)b * C + c);
        if (s_next) sn4 = *reinterpret_cast<const f32x4*>(s_next + (long)b * C + c);
        // the strip's noise values [row][pixel] into LDS (an ordinary load inside the loop would make hipcc drain the DMAs)
        for (int i = tid; i < ROWS * 16; i += 256)
            nzs[i] = noise ? nw * noise[(long)b * noise_bstride + (long)(y0 + (i >> 4)) * OW + x0 + (i & 15)] : 0.f;
        // every ordinary load of this strip has returned before the first DMA is issued: hipcc then has no reason to wait
        // for "all vector memory" (vmcnt(0)) at the first use of one of them inside the pipeline
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(bias4), "v"(dm4), "v"(sn4) : "memory");
        // this wave's 80 of the row's 320 16-byte units: lanes 0 .. 63 and (second instruction) 0 .. 15.  Unit u = pixel
        // x0 - 1 + (u >> 4), channel quad u & 15 of the slice; a pixel left of the image gives a negative = huge unsigned
        // offset: out of range, the DMA writes zeros (so does a row above / below T: zero-byte resource)
        const int u1 = wid * 80 + lane, u2 = wid * 80 + 64 + (lane & 15);
        const unsigned off1 = (unsigned)(((x0 - 1 + (u1 >> 4)) * C + cg * 64 + (u1 & 15) * 4) * 4);
        const unsigned off2 = (unsigned)(((x0 - 1 + (u2 >> 4)) * C + cg * 64 + (u2 & 15) * 4) * 4);
        const float* Tb = T + (long)b * TH * TW * C;
        auto issue = [&](int r) {                             // T row y0 - 1 + r into ring slot r % BD_RING: 2 DMAs per wave
            const int ty = y0 - 1 + r;
            const bool ok = ty >= 0 && ty < TH;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(Tb + (long)(ok ? ty : 0) * TW * C, ok ? rowbytes : 0u);
            char* dst = ring + (r % BD_RING) * BD_ROWB + wid * 1280;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, (int)off1, 0, 0, 0);
            if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + 1024), 16, (int)off2, 0, 0, 0);
        };
        constexpr int NIT = ROWS + 3;                         // input rows of a strip
#pragma unroll
        for (int r = 0; r < BD_AHEAD; ++r) issue(r);
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int g4 = 0; g4 < (NIT + 3) / 4; ++g4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g4 + j;                     // input row y0 - 1 + r
                if (r >= NIT) break;
                if (r + BD_AHEAD < NIT) issue(r + BD_AHEAD);
                // Row r has landed once all but the operations issued after its DMA have retired: the two DMAs of each of the
                // up to BD_AHEAD later rows issued since, and the three stores of each output row finished in the BD_AHEAD
                // iterations before this one (output rows exist from iteration 3 on).
                {
                    const int later = min(BD_AHEAD, NIT - 1 - r);                 // rows r + 1 .. issued after row r
                    const int outs = min(BD_AHEAD, max(0, r - 3));              // iterations r - 5 .. r - 1 that stored
                    bd_wait_vm(2 * later + NST * outs);
                }
                bd_barrier();
                f32x4 v[4];
                {
                    const unsigned ra = ring_lds + (unsigned)((r % BD_RING) * BD_ROWB) + rd_off;
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:512\n\t"
                                 "ds_read_b128 %3, %4 offset:768\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(ra) : "memory");
                }
#pragma unroll
                for (int fy = 0; fy < 4; ++fy) {
                    const f32x4 h = v[0] * k[fy * 4 + 0] + v[1] * k[fy * 4 + 1] + v[2] * k[fy * 4 + 2] + v[3] * k[fy * 4 + 3];
                    acc[(j - fy) & 3] += h;
                }
                const int o = r - 3;                          // this output row is now complete
                const int y = y0 + o;
                if (o >= 0) {                                 // (o < ROWS and y < OH always: whole strips)
                    const f32x4 a = acc[(j + 1) & 3];
                    const float nz = nzs[o * 16 + px];
                    f32x4 ov;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float t = a[q] * dm4[q] + nz + bias4[q];
                        ov[q] = act ? lrelu_sqrt2(t) : t;
                    }
                    *reinterpret_cast<f32x4*>(out + (((long)b * OH + y) * OW + x) * C + c) = ov;          // store 1
                    ov *= sn4;
                    stage_xplanes4<H2>(ov, stage[par] + ((c4 >> 2) * NPL * 16 + px) * 32 + (c4 & 3) * 8);
                    bd_barrier();
                    // 12 runs (chunk, plane) of 512 B = 384 16-B units: this wave's 96 — lanes 0 .. 63, then 0 .. 31 (stores 2, 3);
                    // two fp16 planes: 8 runs = 256 units, this wave's 64 (store 2)
                    const long ppix = ((long)b * (OH + 2) + y + 1) * (OW + 2) + xb * 16 + 1;
#pragma unroll
                    for (int it = 0; it < NST - 1; ++it) {
                        const int u = H2 ? wid * 64 + lane : wid * 96 + it * 64 + (it ? (lane & 31) : lane);
                        const int run = u >> 5, chunk = run / NPL, plane = run - chunk * NPL;
                        const u32x4 d = *reinterpret_cast<const u32x4*>(stage[par] + u * 16);
                        unsigned short* dst = xp_out + ((long)(cg * 4 + chunk) * NPL + plane) * oplane + ppix * 16 + (u & 31) * 8;
                        if (it == 0 || lane < 32) *reinterpret_cast<u32x4*>(dst) = d;
                    }
                    par ^= 1;
                }
                acc[(j + 1) & 3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        bd_barrier();                                         // every wave is done with the ring and the noise of this strip
    }
}

// ToRGB (model.py:445-454): 1x1 modulated conv without demod + bias + FIR-upsampled skip.
//   img[b][c][y][x] = scale * sum_ci x[b,y,x,ci] s[b,ci] W[c][ci] + bias[c] + up2(prev)[b][c][y][x]
// up2 = upfirdn2d(up=2, pad=(2,1)) with the flipped 4x4 FIR kf (already x4).  Image is planar
// NCHW (3 channels).  A team of `lp` lanes (lp = min(64, Cin/4), power of two) owns a pixel and
// reads its Cin floats as float4: one fully coalesced pass over the tap.
// ToRGB for Cin % 4 == 0, Cin <= 4 * 4 * lp: the lane's slice of the 3 x Cin weight stays in registers and a
// wave keeps the rows of TG_R pixel groups in flight (the one-group-at-a-time form below reaches 1.8 TB/s).
constexpr int TG_R = 4;
// One output of upfirdn2d(up=2, pad=(2,1)): the zero-inserted image has samples at even (zy, zx) only, so of the
// 4x4 FIR window exactly the 2x2 taps fy = (y & 1) + {0, 2}, fx = (x & 1) + {0, 2} meet a sample.  Branch-free:
// out-of-range taps read a clamped address with weight 0.
__device__ __forceinline__ float torgb_skip(const float* __restrict__ prev, const float* __restrict__ kf, int b, int c,
                                            int y, int xx, int H, int Wd) {
    const int ph = H >> 1, pw = Wd >> 1;
    const float* pp = prev + ((long)b * 3 + c) * ph * pw;
    const int fy0 = y & 1, fx0 = xx & 1;
    const int py0 = ((y + fy0) >> 1) - 1, px0 = ((xx + fx0) >> 1) - 1;       // (zy >> 1) of the first tap
    float u = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int py = py0 + a;
        const bool yok = py >= 0 && py < ph;
        const int pyc = py < 0 ? 0 : (py >= ph ? ph - 1 : py);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int px = px0 + e;
            const bool ok = yok && px >= 0 && px < pw;
            const int pxc = px < 0 ? 0 : (px >= pw ? pw - 1 : px);
            const float wgt = ok ? kf[(fy0 + 2 * a) * 4 + fx0 + 2 * e] : 0.f;
            u = fmaf(wgt, pp[pyc * pw + pxc], u);
        }
    }
    return u;
}
template <int TG_NI>                 // float4 chunks of a pixel's channels per lane: ceil(Cin / (4 lp))
__global__ __launch_bounds__(256) void torgb_stream_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                           const float* __restrict__ W, float scale,
                                                           const float* __restrict__ bias, const float* __restrict__ prev,
                                                           const float* __restrict__ kf, int B, int H, int Wd, int Cin,
                                                           int lp, float* __restrict__ img) {
    const int ppw = 64 / lp, lane = threadIdx.x & 63, tl = lane % lp, sub = lane / lp;
    constexpr int ni = TG_NI;
    const long npix = (long)B * H * Wd, hw = (long)H * Wd;
    const long ngroups = (npix + ppw - 1) / ppw;
    f32x4 w0[TG_NI], w1[TG_NI], w2[TG_NI];
#pragma unroll
    for (int it = 0; it < TG_NI; ++it) {
        const int ci = (it * lp + tl) * 4;
        const bool ok = it < ni && ci < Cin;
        w0[it] = ok ? *reinterpret_cast<const f32x4*>(W + ci) : zero4();
        w1[it] = ok ? *reinterpret_cast<const f32x4*>(W + Cin + ci) : zero4();
        w2[it] = ok ? *reinterpret_cast<const f32x4*>(W + 2 * Cin + ci) : zero4();
    }
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    for (long g0 = wave0 * TG_R; g0 < ngroups; g0 += nwave * TG_R) {
        f32x4 xv[TG_R][TG_NI];
        long pix[TG_R];
#pragma unroll
        for (int r = 0; r < TG_R; ++r) {
            pix[r] = (g0 + r) * ppw + sub;
            const bool live = pix[r] < npix;
#pragma unroll
            for (int it = 0; it < TG_NI; ++it) {
                const int ci = (it * lp + tl) * 4;
                xv[r][it] = (live && it < ni && ci < Cin) ? *reinterpret_cast<const f32x4*>(x + pix[r] * Cin + ci) : zero4();
            }
        }
        // the skip connection's taps (tiny, cached image of the previous resolution) are fetched now, next to
        // the row loads, not after the reductions
        float skipv[TG_R];
        int oy[TG_R], ox[TG_R], ob[TG_R];
#pragma unroll
        for (int r = 0; r < TG_R; ++r) {
            const bool live = pix[r] < npix;
            const unsigned p32 = live ? (unsigned)pix[r] : 0u;       // the host checks npix < 2^31: 32-bit divisions
            ob[r] = (int)(p32 / (unsigned)hw);
            const unsigned rem = p32 - (unsigned)ob[r] * (unsigned)hw;
            oy[r] = (int)(rem / (unsigned)Wd); ox[r] = (int)(rem - (unsigned)oy[r] * (unsigned)Wd);
            skipv[r] = (prev && live && tl < 3) ? torgb_skip(prev, kf, ob[r], tl, oy[r], ox[r], H, Wd) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < TG_R; ++r) {
            const bool live = pix[r] < npix;
            const int b = ob[r];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int it = 0; it < TG_NI; ++it) {
                const int ci = (it * lp + tl) * 4;
                if (it < ni && ci < Cin) {
                    const f32x4 v = xv[r][it] * *reinterpret_cast<const f32x4*>(s + (long)b * Cin + ci);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a0 = fmaf(v[q], w0[it][q], a0); a1 = fmaf(v[q], w1[it][q], a1); a2 = fmaf(v[q], w2[it][q], a2);
                    }
                }
            }
            a0 = gx_team_sum(a0, lp); a1 = gx_team_sum(a1, lp); a2 = gx_team_sum(a2, lp);
            if (live && tl < 3)
                img[(((long)b * 3 + tl) * H + oy[r]) * Wd + ox[r]] =
                    (tl == 0 ? a0 : (tl == 1 ? a1 : a2)) * scale + bias[tl] + skipv[r];
        }
    }
}

__global__ void torgb_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ W,
                             float scale, const float* __restrict__ bias, const float* __restrict__ prev,
                             const float* __restrict__ kf, int B, int H, int Wd, int Cin, int lp,
                             float* __restrict__ img) {
    const int ppw = 64 / lp;                                       // pixels per wave
    const int lane = threadIdx.x & 63, tl = lane % lp;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long pix = wave * ppw + lane / lp;
    const bool live = pix < (long)B * H * Wd;
    const int b = live ? (int)(pix / ((long)H * Wd)) : 0;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (live) {
        const float* px = x + pix * Cin;
        const float* ps = s + (long)b * Cin;
        if ((Cin & 3) == 0) {
            for (int ci = tl * 4; ci < Cin; ci += lp * 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(px + ci) * *reinterpret_cast<const f32x4*>(ps + ci);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(W + ci);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(W + Cin + ci);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(W + 2 * Cin + ci);
#pragma unroll
                for (int q = 0; q < 4; ++q) { a0 = fmaf(v[q], w0[q], a0); a1 = fmaf(v[q], w1[q], a1); a2 = fmaf(v[q], w2[q], a2); }
            }
        } else {
            for (int ci = tl; ci < Cin; ci += lp) {
                const float v = px[ci] * ps[ci];
                a0 = fmaf(v, W[ci], a0); a1 = fmaf(v, W[Cin + ci], a1); a2 = fmaf(v, W[2 * Cin + ci], a2);
            }
        }
    }
    a0 = gx_team_sum(a0, lp); a1 = gx_team_sum(a1, lp); a2 = gx_team_sum(a2, lp);
    if (live && tl < 3) {
        const int rem = (int)(pix - (long)b * H * Wd);
        const int y = rem / Wd, xx = rem - y * Wd;
        float v = (tl == 0 ? a0 : (tl == 1 ? a1 : a2)) * scale + bias[tl];
        if (prev) {
            const int ph = H >> 1, pw = Wd >> 1;
            const float* pp = prev + ((long)b * 3 + tl) * ph * pw;
            float u = 0.f;
#pragma unroll
            for (int fy = 0; fy < 4; ++fy) {
                const int zy = y + fy - 2;
                if (zy < 0 || zy >= H || (zy & 1)) continue;
#pragma unroll
                for (int fx = 0; fx < 4; ++fx) {
                    const int zx = xx + fx - 2;
                    if (zx < 0 || zx >= Wd || (zx & 1)) continue;
                    u = fmaf(kf[fy * 4 + fx], pp[(zy >> 1) * pw + (zx >> 1)], u);
                }
            }
            v += u;
        }
        img[(((long)b * 3 + tl) * H + y) * Wd + xx] = v;
    }
}

// PixelNorm (model.py:105-110): x * rsqrt(mean(x^2) + 1e-8), one wave per row.
__global__ void pixel_norm_kernel(const float* __restrict__ x, long rows, int D, float* __restrict__ y) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = x[row * D + c]; s = fmaf(v, v, s); }
    s = gx_wave_sum(s);
    const float r = rsqrtf(s / (float)D + 1e-8f);
    for (int c = lane; c < D; c += 64) y[row * D + c] = x[row * D + c] * r;
}

// y = act(x * alpha + bias[col] * bscale)   (EqualLinear epilogue, model.py:242-252)
__global__ void bias_act_kernel(float* __restrict__ y, long rows, int N, float alpha, const float* __restrict__ bias,
                                float bscale, int act) {
    const long total = rows * N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = y[i] * alpha;
        if (bias) v += bias[i % N] * bscale;
        y[i] = act ? lrelu_sqrt2(v) : v;
    }
}

// EqualLinear + fused_leaky_relu for a FEW rows (the mapping network on a batch of latents, model.py:242-252, :420-435):
//   out[r][n] = act(alpha * sum_k x[r][k] W[n][k] + bias[n] * bscale),  W given transposed (wt [K][N], made once per model).
// 48 latents x 512 x 512 is 25 MFLOP: on the MFMA tiles that was 4 output tiles split 8 ways over k, a slab reduce and a
// bias pass — three launches of 4-9 us per layer, eight layers in a row (0.24 ms of the bench step, all latency).  Here a
// workgroup owns 64 columns x RG rows, its four waves a quarter of k each (weights: one coalesced 256-B row per k and
// wave; activations: wave-uniform addresses, i.e. scalar loads), plain fp32 FMAs, fixed-order sum of the four quarters.
template <int RG>
__global__ __launch_bounds__(256) void equal_linear_skinny_kernel(const float* __restrict__ x, int rows, int K,
                                                                  const float* __restrict__ wt, int N, float alpha,
                                                                  const float* __restrict__ bias, float bscale, int act,
                                                                  float* __restrict__ out) {
    __shared__ float red[3][RG][64];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.x * 64 + lane, nc = n < N ? n : N - 1;
    const int r0 = blockIdx.y * RG, kq = K >> 2, k0 = w * kq;
    float acc[RG];
#pragma unroll
    for (int r = 0; r < RG; ++r) acc[r] = 0.f;
    for (int k = k0; k < k0 + kq; k += 4) {
        float wv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wv[j] = wt[(long)(k + j) * N + nc];
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (long)min(r0 + r, rows - 1) * K + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[r] = fmaf(xv[j], wv[j], acc[r]);
        }
    }
    if (w > 0) {
#pragma unroll
        for (int r = 0; r < RG; ++r) red[w - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (w == 0 && n < N) {
        const float b = bias ? bias[n] * bscale : 0.f;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            if (r0 + r >= rows) break;
            const float v = ((acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane])) * alpha + b;
            out[(long)(r0 + r) * N + n] = act ? lrelu_sqrt2(v) : v;
        }
    }
}

// out = base + psi * (w - base), base broadcast over rows of length D  (truncation, model.py:594-602)
__global__ void lerp_rows_kernel(const float* __restrict__ w, const float* __restrict__ base, long n, int D, float psi,
                                 float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float m = base[i % D];
        out[i] = m + psi * (w[i] - m);
    }
}

// General up-FIR-down (upfirdn2d_kernel.cu:53-215 semantics) over [major][in_h][in_w][minor]:
//   out[m,oy,ox,c] = sum_{ky,kx} K[kh-1-ky][kw-1-kx] * U[m, oy*down_y + ky - pad_y0, ox*down_x + kx - pad_x0, c]
// where U is the zero-inserted (up_y, up_x) image.  One thread per output element, the minor index fastest:
// writes and (for minor > 1, or up_x = down_x = 1) reads are coalesced; the taps that meet a sample are
// enumerated directly (first ky with (oy*down_y + ky - pad_y0) % up_y == 0, then steps of up_y).
__global__ void upfirdn2d_kernel(const float* __restrict__ in, const float* __restrict__ K, int major, int in_h,
                                 int in_w, int minor, int kh, int kw, int up_x, int up_y, int down_x, int down_y,
                                 int pad_x0, int pad_y0, int out_h, int out_w, float* __restrict__ out) {
    const long total = (long)major * out_h * out_w * minor;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % minor);
        long r = i / minor;
        const int ox = (int)(r % out_w); r /= out_w;
        const int oy = (int)(r % out_h);
        const long m = r / out_h;
        const int by = oy * down_y - pad_y0, bx = ox * down_x - pad_x0;       // U coordinate of tap (0, 0)
        int ky0 = ((-by) % up_y + up_y) % up_y, kx0 = ((-bx) % up_x + up_x) % up_x;
        const float* src = in + m * in_h * (long)in_w * minor + c;
        float v = 0.f;
        for (int ky = ky0; ky < kh; ky += up_y) {
            const int uy = by + ky;
            if (uy < 0) continue;
            const int iy = uy / up_y;
            if (iy >= in_h) break;
            for (int kx = kx0; kx < kw; kx += up_x) {
                const int ux = bx + kx;
                if (ux < 0) continue;
                const int ix = ux / up_x;
                if (ix >= in_w) break;
                v = fmaf(src[((long)iy * in_w + ix) * minor], K[(kh - 1 - ky) * kw + (kw - 1 - kx)], v);
            }
        }
        out[i] = v;
    }
}

// fused_bias_act_kernel.cu:18-85: y = act'(x + b[(i / step_b) % size_b]) * scale
__global__ void fused_bias_act_kernel(const float* __restrict__ x, long n, const float* __restrict__ b, int size_b,
                                      long step_b, const float* __restrict__ ref, int act, int grad, float alpha,
                                      float scale, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = x[i];
        if (b) v += b[(i / step_b) % size_b];
        const float rf = ref ? ref[i] : 0.f;
        float y;
        switch (act * 10 + grad) {
            default:
            case 10: case 11: y = v; break;
            case 12: case 32: y = 0.f; break;
            case 30: y = v > 0.f ? v : v * alpha; break;
            case 31: y = rf > 0.f ? v : v * alpha; break;
        }
        out[i] = y * scale;
    }
}

// NCHW <-> NHWC (API boundary only: the reference hands taps around as NCHW)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, int B, int C, int HW, float* __restrict__ out) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long r = i / C; const int p = (int)(r % HW); const int b = (int)(r / HW);
        out[i] = in[((long)b * C + c) * HW + p];
    }
}

// Small layers (fewer tiles than CUs) are split over their k-tiles so that the chip is filled.
size_t conv_slab_bytes(int B, int outH, int outW, int Cout, int splits) {
    return splits > 1 ? (size_t)splits * B * outH * outW * Cout * sizeof(float) : 0;
}
static int conv_pick_splits(long max_m, int Cout, int nsub_tiles_total, int min_ktiles) {
    (void)max_m; (void)Cout;
    if (nsub_tiles_total >= 256) return 1;
    int s = (512 + nsub_tiles_total - 1) / nsub_tiles_total;
    if (s > min_ktiles) s = min_ktiles;
    if (s > 16) s = 16;
    return s < 1 ? 1 : s;
}

// Plane format of the x-planes conv path: 0 = three bf16 planes, six products (bf16x6); 1 = two fp16 planes, three products
// (f16x3, gx_mfma_x6.h).  Read at launch time by the producers (gx_split_conv_weights, gx_modsplit, the conv / blur epilogues
// that write the next conv's planes) and by the conv: planes made under one format must be consumed under the same.
#define GX_DEFAULT_CONV_PLANES 1
static int g_conv_planes = -1;
static int g_xh_tall = 1;                      // lab: gx_set_conv_tall(0) keeps the 128 x 128 tile everywhere
constexpr long XH_TALL_MIN_TILES = 512;
static int conv_planes() {
    if (g_conv_planes < 0) {
        const char* e = getenv("GX_CONV_PLANES");
        g_conv_planes = (e && e[0] == '0') ? 0 : (e && e[0] == '1') ? 1 : GX_DEFAULT_CONV_PLANES;
    }
    return g_conv_planes;
}

static int launch_conv(ConvArgs& g, float* slab, size_t slab_bytes, hipStream_t stream) {
    int wtaps = 1;
    for (int i = 0; i < g.nsub; ++i)
        for (int t = 0; t < g.sub[i].ntaps; ++t) wtaps = g.sub[i].tw[t] + 1 > wtaps ? g.sub[i].tw[t] + 1 : wtaps;
    const size_t inb = (size_t)g.B * g.IH * g.IW * g.Cin * 4, wtb = (size_t)wtaps * g.Cout * g.Cin * 4;
    g.vec_ok = ((((uintptr_t)g.in | (uintptr_t)g.wt | (uintptr_t)g.style) & 15) == 0) && (g.Cin % 4 == 0) &&
               inb < 0xFFFFFFF0ull && wtb < 0xFFFFFFF0ull;
    g.in_bytes = (unsigned)inb;
    g.wt_bytes = (unsigned)wtb;
    g.wtp_ncc = gx_cdiv(g.Cin, 16);
    const size_t wpb = gx_conv_weight_planes_bytes(wtaps, g.Cout, g.Cin);
    if (g.wtp && (((uintptr_t)g.wtp & 15) || wpb >= 0xFFFFFFF0ull)) g.wtp = nullptr;
    g.wtp_bytes = (unsigned)wpb;
    g.style_bytes = (unsigned)((size_t)g.B * g.Cin * 4);
    g.tiles_n = gx_cdiv(g.Cout, BN);
    const bool x6 = gx_get_gemm_mode() == 1;
    const bool h2 = x6 && conv_planes() == 1;
    g.h2 = (h2 && g.xp) ? 1 : 0;
    g.acc_scale = g.h2 ? XH_UNSCALE : 1.f;
    if (h2 && !g.xp) g.wtp = nullptr;      // the kernels that split x in flight (bf16x6) cannot read fp16 weight planes: they split w too
    int base = 0, min_kt = 1 << 30;
    // narrow layers on the x-planes path: 256-pixel x 64-channel tiles (modconv_xp_kernel<true, true>)
    bool tall = g.xp && x6 && !h2 && (g.Cout & 3) == 0 && g.Cout <= 64;
    if (g.xp && h2) {       // f16x3: 256 x 128 tiles (a wave 128 x 64: 12 fragment reads per 24 MFMAs instead of 8 per 12 — the
                            // 128 x 128 tile is LDS-bound) where that still leaves every CU two workgroups
        long t256 = 0;
        for (int i = 0; i < g.nsub; ++i) t256 += (long)gx_cdiv((long)g.B * g.sub[i].OH * g.sub[i].OW, 2 * BM) * g.tiles_n;
        tall = g_xh_tall && t256 >= XH_TALL_MIN_TILES;
    }
    for (int i = 0; i < g.nsub; ++i) {
        g.sub[i].tile_base = base;
        g.sub[i].tiles_m = gx_cdiv((long)g.B * g.sub[i].OH * g.sub[i].OW, tall ? 2 * BM : BM);
        base += g.sub[i].tiles_m * g.tiles_n;
        const int kt = gx_cdiv(g.Cin, BK);                     // split-K granularity = channel chunks (>= this many in either engine)
        min_kt = kt < min_kt ? kt : min_kt;
        if ((long)g.B * g.sub[i].OH * g.sub[i].OW >= (1L << 31)) return GX_ERR_BAD_ARG;
    }
    g.ntile = base;
    g.splits = g.xp ? 1 : conv_pick_splits(0, g.Cout, base, min_kt);
    g.slab = nullptr;
    if (g.splits > 1) {
        if (!slab || slab_bytes < conv_slab_bytes(g.B, g.outH, g.outW, g.Cout, g.splits)) g.splits = 1;   // no scratch: unsplit
        else g.slab = slab;
    }
    dim3 grid(g.ntile, g.splits);
    if (g.xp) {       // pre-split modulated input: bf16x6 engine, pre-split weights, no split-K
        if (!x6 || !g.wtp || g.splits != 1 || ((uintptr_t)g.xp & 15)) return GX_ERR_BAD_ARG;
        const size_t pb = (size_t)g.B * (g.IH + 2) * (g.IW + 2) * 32;
        if (pb >= 0x7FFFFFF0ull) return GX_ERR_BAD_ARG;
        g.xp_plane_bytes = (unsigned)pb;
        // MFMA shape of the six products: 16x16x32 (plane pairs along K) where Cout allows its epilogue, else 32x32x16.  Equal
        // pipe cycles per FLOP; on random data the power-limited chip holds a higher clock on the 16x16 shape: plain convs
        // 243-246 -> 256-260 TFLOP/s, same-process A/B (round 2, tools/bench_conv_xp.py)
        if (h2 && tall) hipLaunchKernelGGL((modconv_xp_kernel<false, true, true>), grid, dim3(GEMM_THREADS), 0, stream, g);
        else if (h2) hipLaunchKernelGGL((modconv_xp_kernel<false, false, true>), grid, dim3(GEMM_THREADS), 0, stream, g);
        else if (tall) hipLaunchKernelGGL((modconv_xp_kernel<true, true>), grid, dim3(GEMM_THREADS), 0, stream, g);
        else if ((g.Cout & 3) == 0) hipLaunchKernelGGL(modconv_xp_kernel<true>, grid, dim3(GEMM_THREADS), 0, stream, g);
        else hipLaunchKernelGGL(modconv_xp_kernel<false>, grid, dim3(GEMM_THREADS), 0, stream, g);
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    if (x6 && g.vec_ok) hipLaunchKernelGGL((modconv_kernel<true, true>), grid, dim3(GEMM_THREADS), 0, stream, g);
    else if (x6) hipLaunchKernelGGL((modconv_kernel<false, true>), grid, dim3(GEMM_THREADS), 0, stream, g);
    else if (g.vec_ok) hipLaunchKernelGGL((modconv_kernel<true, false>), grid, dim3(GEMM_THREADS), 0, stream, g);
    else hipLaunchKernelGGL((modconv_kernel<false, false>), grid, dim3(GEMM_THREADS), 0, stream, g);
    GX_CHECK_LAUNCH();
    if (g.splits > 1) {
        const long total = (long)g.B * g.outH * g.outW * g.Cout;
        long want = (total + 255) / 256;
        hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, stream,
                           g.slab, g.splits, total, g.B, g.outH, g.outW, g.Cout, g.demod, g.noise, g.noise_bstride,
                           g.noise_w, g.bias, g.act, g.out);
        GX_CHECK_LAUNCH();
    }
    return GX_OK;
}

}  // namespace gx

using namespace gx;

extern "C" {

int gx_demod(const float* s, const float* wsq, int B, int Cin, int Cout, float scale, float* demod, void* stream) {
    if (!s || !wsq || !demod || B <= 0 || Cin <= 0 || Cout <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(demod_kernel, dim3(gx_cdiv((long)B * Cout, 4)), dim3(256), 0, (hipStream_t)stream, s, wsq, B,
                       Cin, Cout, scale, demod);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// Pre-split (frozen) conv weights wt [taps][Cout][Cin] for the bf16x6 engine; `planes` holds
// gx_conv_weight_planes_bytes(taps, Cout, Cin) bytes (tiled, see split_conv_weights_kernel).
size_t gx_conv_weight_planes_bytes(int taps, int Cout, int Cin) {
    return (size_t)taps * gx_cdiv(Cout, 128) * gx_cdiv(Cin, 16) * 3 * 4096;
}
int gx_split_conv_weights(const float* wt, int taps, int Cout, int Cin, unsigned short* planes, void* stream) {
    if (!wt || !planes || taps <= 0 || Cout <= 0 || Cin <= 0) return GX_ERR_BAD_ARG;
    const int ntn = gx_cdiv(Cout, 128), ncc = gx_cdiv(Cin, 16);
    const long total = (long)taps * ntn * ncc * 2048;
    long want = (total + 255) / 256;
    hipLaunchKernelGGL(split_conv_weights_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream,
                       wt, taps, Cout, Cin, (long)Cin, ntn, ncc, (gx_get_gemm_mode() == 1 && conv_planes() == 1) ? 1 : 0, nullptr, planes);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// The same planes for the weight operand B [N][K] (row stride ldb) of gx_gemm_f32_nt_planes: one "tap";
// gx_gemm_weight_planes_bytes(N, K) bytes.
int gx_split_gemm_weights(const float* B, int N, int K, long ldb, unsigned short* planes, void* stream) {
    if (!B || !planes || N <= 0 || K <= 0 || ldb < K) return GX_ERR_BAD_ARG;
    const int ntn = gx_cdiv(N, 128), ncc = gx_cdiv(K, 16);
    const long total = (long)ntn * ncc * 2048;
    long want = (total + 255) / 256;
    hipLaunchKernelGGL(split_conv_weights_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream,
                       B, 1, N, K, ldb, ntn, ncc, 0, nullptr, planes);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// ... as the two fp16 planes of the f16x3 form: B scaled by the power of two that brings its largest magnitude to [2^13, 2^14)
// (found here, on the device: one pass over B) and that magnitude left in the 16-byte trailer of `planes`, where
// gx_gemm_f32_nt_planes(..., a_range > 0) reads it back.
int gx_split_gemm_weights_h2(const float* B, int N, int K, long ldb, unsigned short* planes, void* stream) {
    if (!B || !planes || N <= 0 || K <= 0 || ldb < K) return GX_ERR_BAD_ARG;
    const int ntn = gx_cdiv(N, 128), ncc = gx_cdiv(K, 16);
    float* trailer = reinterpret_cast<float*>(reinterpret_cast<char*>(planes) + (size_t)ntn * ncc * 3 * 4096);
    if (hipMemsetAsync(trailer, 0, 16, (hipStream_t)stream) != hipSuccess) return GX_ERR_LAUNCH;
    const long n = (long)N * K;
    long wantm = (n + 1023) / 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3((int)(wantm < 1024 ? wantm : 1024)), dim3(256), 0, (hipStream_t)stream, B, N, K, ldb,
                       reinterpret_cast<unsigned*>(trailer));
    GX_CHECK_LAUNCH();
    const long total = (long)ntn * ncc * 2048;
    long want = (total + 255) / 256;
    hipLaunchKernelGGL(split_conv_weights_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream,
                       B, 1, N, K, ldb, ntn, ncc, 1, trailer, planes);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// lat [B][n_latent][S] of one view: truncated w, with the two perturbed latents of every image mixed
// towards its two mapped noises (w [B][S], mean [S], nw [B][2][S], mix [B][n_latent][2]).
int gx_view_latents(const float* w, const float* mean, float psi, const float* nw, const float* mix, int B, int n_latent,
                    int S, float* lat, void* stream) {
    if (!w || !mean || !nw || !mix || !lat || B <= 0 || n_latent <= 0 || S <= 0) return GX_ERR_BAD_ARG;
    long want = ((long)B * n_latent * S + 255) / 256;
    hipLaunchKernelGGL(view_latents_kernel, dim3((int)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, w,
                       mean, psi, nw, mix, B, n_latent, S, lat);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// Per-layer styles of one generator forward, gathered from the all-latents x all-layers affine GEMM.
int gx_style_pick(const float* S_all, int B, int n_latent, int total, const int* table, int nlayer, float* out,
                  void* stream) {
    if (!S_all || !table || !out || B <= 0 || n_latent <= 0 || total <= 0 || nlayer <= 0) return GX_ERR_BAD_ARG;
    const long n = (long)B * total;
    long want = (n + 255) / 256;
    hipLaunchKernelGGL(style_pick_kernel, dim3((int)(want < 2048 ? want : 2048)), dim3(256), 0, (hipStream_t)stream, S_all,
                       B, n_latent, total, table, nlayer, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// demod' of every styled conv of one generator forward (gx_demod per layer, one launch).
// total_rows = sum_j cout_j; demod holds B*total_rows floats, layer j at B*doff_j as [B][cout_j].
int gx_demod_batch(const float* s_packed, const float* wsq_all, int B, const int* table, const float* scale, int nconv,
                   int total_rows, float* demod, void* stream) {
    if (!s_packed || !wsq_all || !table || !scale || !demod || B <= 0 || nconv <= 0 || total_rows <= 0)
        return GX_ERR_BAD_ARG;
    const long waves = (long)B * total_rows;
    hipLaunchKernelGGL(demod_batch_kernel, dim3(gx_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, s_packed, wsq_all,
                       B, table, scale, nconv, waves, demod);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// Per (styled conv, sample) of gx_demod_batch's table: s /= c, demod *= c in place, c = the power of two that brings the
// sample's largest |s| into [2, 4) (style_norm_kernel: exact, the convs' results are unchanged).  After gx_demod_batch.
int gx_style_norm(float* s_packed, float* demod, int B, const int* table, int nconv, void* stream) {
    if (!s_packed || !demod || !table || B <= 0 || nconv <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(style_norm_kernel, dim3(B * nconv), dim3(256), 0, (hipStream_t)stream, s_packed, demod, B, table);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// Plain 3x3 (pad 1) or 1x1 modulated conv with the StyledConv epilogue.
//   x [B][H][W][Cin] NHWC; wt [k*k][Cout][Cin]; s [B][Cin]; demod [B][Cout] or null;
//   noise [.][H][W] (batch stride noise_bstride, 0 = shared) or null; noise_w device scalar;
//   bias [Cout] or null; act 1 = lrelu*sqrt2; out [B][H][W][Cout] NHWC.
// Split-K scratch for the small (low-resolution) layers: pass at least this many bytes as `slab`
// (0 bytes / NULL is allowed: the layer then runs unsplit).
size_t gx_modconv_slab_bytes(int B, int outH, int outW, int Cout) { return conv_slab_bytes(B, outH, outW, Cout, 16); }

// wt_planes (optional, may be NULL): gx_split_bf16x3(wt) — used by the bf16x6 engine instead of splitting
// the weight tile in every workgroup.
int gx_modconv_fwd_p(const float* x, int B, int H, int W, int Cin, const float* wt, const unsigned short* wt_planes,
                     int ksize, int Cout, const float* s, const float* demod, const float* noise, long noise_bstride,
                     const float* noise_w, const float* bias, int act, float* out, float* slab, size_t slab_bytes,
                     void* stream) {
    if (!x || !wt || !out || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (ksize != 1 && ksize != 3))
        return GX_ERR_BAD_ARG;
    ConvArgs g{};
    g.wtp = wt_planes;
    g.in = x; g.B = B; g.IH = H; g.IW = W; g.Cin = Cin; g.wt = wt; g.style = s;
    g.nsub = 1;
    ConvSub& sb = g.sub[0];
    sb.ntaps = ksize * ksize;
    for (int ky = 0; ky < ksize; ++ky)
        for (int kx = 0; kx < ksize; ++kx) {
            const int t = ky * ksize + kx;
            sb.dy[t] = ky - ksize / 2; sb.dx[t] = kx - ksize / 2; sb.tw[t] = t;
        }
    sb.OH = H; sb.OW = W; sb.yadd = 0; sb.xadd = 0;
    g.out = out; g.outH = H; g.outW = W; g.Cout = Cout;
    g.ymul = 1; g.xmul = 1;
    g.demod = demod; g.noise = noise; g.noise_bstride = noise_bstride; g.noise_w = noise_w; g.bias = bias; g.act = act;
    return launch_conv(g, slab, slab_bytes, (hipStream_t)stream);
}
int gx_modconv_fwd(const float* x, int B, int H, int W, int Cin, const float* wt, int ksize, int Cout, const float* s,
                   const float* demod, const float* noise, long noise_bstride, const float* noise_w,
                   const float* bias, int act, float* out, float* slab, size_t slab_bytes, void* stream) {
    return gx_modconv_fwd_p(x, B, H, W, Cin, wt, nullptr, ksize, Cout, s, demod, noise, noise_bstride, noise_w, bias, act,
                            out, slab, slab_bytes, stream);
}

// The modulated input x * s split once into bf16 planes (layout: modconv_xp_kernel).  `planes` holds
// gx_xplanes_bytes(B, H, W, Cin) bytes whose spatial border the CALLER has zeroed once (kernels never write it).
size_t gx_xplanes_bytes(int B, int H, int W, int Cin) {
    return (size_t)gx_cdiv(Cin, 16) * 3 * B * (H + 2) * (W + 2) * 32;
}
int gx_modsplit(const float* x, int B, int H, int W, int Cin, const float* s, unsigned short* planes, void* stream) {
    if (!x || !planes || B <= 0 || H <= 0 || W <= 0 || Cin <= 0) return GX_ERR_BAD_ARG;
    if ((((uintptr_t)x | (uintptr_t)s | (uintptr_t)planes) & 15) != 0) return GX_ERR_ALIGN;
    const int ncc = gx_cdiv(Cin, 16);
    const long total = (long)B * H * W * ncc * 4;
    long want = (total + 255) / 256;
    hipLaunchKernelGGL(modsplit_kernel, dim3((int)(want < 32768 ? want : 32768)), dim3(256), 0, (hipStream_t)stream, x, s, B,
                       H, W, Cin, ncc, (gx_get_gemm_mode() == 1 && conv_planes() == 1) ? 1 : 0, planes);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// gx_modconv_fwd_p / gx_modconv_up_convt_p with the input given as x planes (x and s are then unused by the
// contraction; bf16x6 engine and wt_planes required, layers with >= 256 output tiles only).
int gx_modconv_fwd_xp(const unsigned short* x_planes, int B, int H, int W, int Cin, const float* wt,
                      const unsigned short* wt_planes, int ksize, int Cout, const float* demod, const float* noise,
                      long noise_bstride, const float* noise_w, const float* bias, int act, float* out,
                      const float* s_next, unsigned short* planes_out, void* stream) {
    if (planes_out && (Cout & 3)) return GX_ERR_BAD_ARG;
    if (!x_planes || !wt || !wt_planes || !out || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 ||
        (ksize != 1 && ksize != 3))
        return GX_ERR_BAD_ARG;
    ConvArgs g{};
    g.xp = x_planes; g.wtp = wt_planes;
    g.in = wt; g.B = B; g.IH = H; g.IW = W; g.Cin = Cin; g.wt = wt; g.style = nullptr;
    g.nsub = 1;
    ConvSub& sb = g.sub[0];
    sb.ntaps = ksize * ksize;
    for (int ky = 0; ky < ksize; ++ky)
        for (int kx = 0; kx < ksize; ++kx) {
            const int t = ky * ksize + kx;
            sb.dy[t] = ky - ksize / 2; sb.dx[t] = kx - ksize / 2; sb.tw[t] = t;
        }
    sb.OH = H; sb.OW = W; sb.yadd = 0; sb.xadd = 0;
    g.out = out; g.outH = H; g.outW = W; g.Cout = Cout;
    g.ymul = 1; g.xmul = 1;
    g.demod = demod; g.noise = noise; g.noise_bstride = noise_bstride; g.noise_w = noise_w; g.bias = bias; g.act = act;
    g.s_next = s_next; g.xp_out = planes_out;
    return launch_conv(g, nullptr, 0, (hipStream_t)stream);
}
int gx_modconv_up_convt_xp(const unsigned short* x_planes, int B, int H, int W, int Cin, const float* wt,
                           const unsigned short* wt_planes, int Cout, float* T, void* stream) {
    if (!x_planes || !wt || !wt_planes || !T || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return GX_ERR_BAD_ARG;
    ConvArgs g{};
    g.xp = x_planes; g.wtp = wt_planes;
    g.in = wt; g.B = B; g.IH = H; g.IW = W; g.Cin = Cin; g.wt = wt; g.style = nullptr;
    g.nsub = 4;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            ConvSub& sb = g.sub[py * 2 + px];
            int n = 0;
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2) {
                    sb.dy[n] = -(ky >> 1); sb.dx[n] = -(kx >> 1); sb.tw[n] = ky * 3 + kx; ++n;
                }
            sb.ntaps = n;
            sb.OH = py ? H : H + 1; sb.OW = px ? W : W + 1;
            sb.yadd = py; sb.xadd = px;
        }
    g.out = T; g.outH = 2 * H + 1; g.outW = 2 * W + 1; g.Cout = Cout;
    g.ymul = 2; g.xmul = 2;
    return launch_conv(g, nullptr, 0, (hipStream_t)stream);
}

// Upsampling StyledConv: conv_transpose2d(stride 2, pad 0) -> (2H+1)^2 scratch T, then blur
// pad (1,1) + demod + noise + bias + act -> out [B][2H][2W][Cout].
//   T: caller-allocated scratch of B*(2H+1)*(2W+1)*Cout floats; kf: flipped 4x4 FIR (device, 16 floats)
size_t gx_modconv_up_scratch_bytes(int B, int H, int W, int Cout) {
    return (size_t)B * (2 * H + 1) * (2 * W + 1) * Cout * sizeof(float);
}
// Stage 2 of an upsampling StyledConv: blur (pad (1,1)) of the (TH x TW) transposed-conv image +
// demod + noise + bias + act -> out [B][TH-1][TW-1][C]   (upfirdn2d mode 1 + fused_bias_act).
int gx_set_conv_planes(int fmt) {
    const int prev = gx::conv_planes();
    if (fmt == 0 || fmt == 1) gx::g_conv_planes = fmt;
    return prev;
}
int gx_get_conv_planes(void) { return gx::conv_planes(); }
int gx_set_conv_tall(int on) { const int prev = gx::g_xh_tall; gx::g_xh_tall = on ? 1 : 0; return prev; }
static int g_blur_dma = 1;
int gx_set_blur_dma(int on) { const int prev = g_blur_dma; g_blur_dma = on ? 1 : 0; return prev; }
int gx_blur_act_fwd(const float* T, int B, int TH, int TW, int C, const float* kf, const float* demod,
                    const float* noise, long noise_bstride, const float* noise_w, const float* bias, int act,
                    float* out, void* stream) {
    return gx_blur_act_fwd_xp(T, B, TH, TW, C, kf, demod, noise, noise_bstride, noise_w, bias, act, out, nullptr, nullptr,
                              stream);
}
// one of the strip kernels in the plane format of the moment (h2: two fp16 planes)
#define GX_BLUR_LAUNCH(KERNEL, ROWS_, NBLK)                                                                                  \
    do {                                                                                                                     \
        const dim3 grid_((int)((NBLK) < 65536 ? (NBLK) : 65536));                                                            \
        if (h2) hipLaunchKernelGGL((KERNEL<ROWS_, true>), grid_, dim3(256), 0, (hipStream_t)stream, T, B, TH, TW, C, kf, demod, noise, \
                                   noise_bstride, noise_w, bias, act, out, OH, OW, s_next, planes_out);                      \
        else hipLaunchKernelGGL((KERNEL<ROWS_, false>), grid_, dim3(256), 0, (hipStream_t)stream, T, B, TH, TW, C, kf, demod, noise,  \
                                noise_bstride, noise_w, bias, act, out, OH, OW, s_next, planes_out);                         \
    } while (0)
// ... and, optionally, the x planes of out * s_next for the next conv (planes_out: gx_xplanes_bytes(B, TH-1, TW-1, C))
int gx_blur_act_fwd_xp(const float* T, int B, int TH, int TW, int C, const float* kf, const float* demod,
                       const float* noise, long noise_bstride, const float* noise_w, const float* bias, int act,
                       float* out, const float* s_next, unsigned short* planes_out, void* stream) {
    if (!T || !kf || !out || B <= 0 || TH < 2 || TW < 2 || C <= 0 || (C & 3)) return GX_ERR_BAD_ARG;
    const int OH = TH - 1, OW = TW - 1;
    const bool h2 = gx_get_gemm_mode() == 1 && conv_planes() == 1;               // format of planes_out
    long total = (long)B * ((OH + BLUR_ROWS - 1) / BLUR_ROWS) * OW * (C >> 2);
    if (planes_out && (C & 15) == 0) total = (long)B * ((OH + BLUR_ROWS - 1) / BLUR_ROWS) * ((OW + 15) / 16) * (C >> 4) * 64;
    long want = (total + 255) / 256;
    if (planes_out && (C & 63) == 0 && (OW & 15) == 0 &&
        ((((uintptr_t)T | (uintptr_t)out | (uintptr_t)planes_out) & 15) == 0)) {
        // strips of 32 rows where that still leaves every CU several workgroups (the 64^2 outputs and up at the bench batch);
        // same-process A/B (round 3): the XCD-contiguous tile order 0.467 -> 0.435 ms on the 128^2 output, 0.790 -> 0.782 on 256^2,
        // 32-row strips another 1 %
        const long nblk32 = (long)B * ((OH + 31) / 32) * (OW >> 4) * (C >> 6);
        if (OH >= 64 && nblk32 >= 2048) {
            const size_t rowb = (size_t)TW * C * 4;
            if (g_blur_dma && rowb < 0x7FFFFFF0ull && OH % BD_ROWS == 0)   // T rows by LDS-DMA, several ahead (gx_set_blur_dma(0): the register form)
                GX_BLUR_LAUNCH(blur_act_dma_kernel, BD_ROWS, nblk32 * 32 / BD_ROWS);
            else
                GX_BLUR_LAUNCH(blur_act_xp_kernel, 32, nblk32);
            GX_CHECK_LAUNCH();
            return GX_OK;
        }
        const long nblk = (long)B * ((OH + BLUR_ROWS - 1) / BLUR_ROWS) * (OW >> 4) * (C >> 6);
        GX_BLUR_LAUNCH(blur_act_xp_kernel, BLUR_ROWS, nblk);
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    const int nb = (int)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(blur_act_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, T, B, TH, TW, C, kf, demod,
                       noise, noise_bstride, noise_w, bias, act, out, OH, OW, s_next, planes_out, h2 ? 1 : 0);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// Stage 1: conv_transpose2d(stride 2, pad 0) of the modulated input as four implicit GEMMs (one
// per output parity class) -> T [B][2H+1][2W+1][Cout] (raw sums, no demod).
int gx_modconv_up_convt_p(const float* x, int B, int H, int W, int Cin, const float* wt, const unsigned short* wt_planes,
                          int Cout, const float* s, float* T, float* slab, size_t slab_bytes, void* stream) {
    if (!x || !wt || !T || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return GX_ERR_BAD_ARG;
    ConvArgs g{};
    g.wtp = wt_planes;
    g.in = x; g.B = B; g.IH = H; g.IW = W; g.Cin = Cin; g.wt = wt; g.style = s;
    g.nsub = 4;
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            ConvSub& sb = g.sub[py * 2 + px];
            // T[u,v] = sum over (ky,kx) with u-ky, v-kx even: x[(u-ky)/2, (v-kx)/2] * W[.,.,ky,kx]
            int n = 0;
            for (int ky = py; ky < 3; ky += 2)
                for (int kx = px; kx < 3; kx += 2) {
                    sb.dy[n] = -(ky >> 1); sb.dx[n] = -(kx >> 1); sb.tw[n] = ky * 3 + kx; ++n;
                }
            sb.ntaps = n;
            sb.OH = py ? H : H + 1; sb.OW = px ? W : W + 1;
            sb.yadd = py; sb.xadd = px;
        }
    g.out = T; g.outH = 2 * H + 1; g.outW = 2 * W + 1; g.Cout = Cout;
    g.ymul = 2; g.xmul = 2;
    g.demod = nullptr; g.noise = nullptr; g.noise_w = nullptr; g.bias = nullptr; g.act = 0;
    return launch_conv(g, slab, slab_bytes, (hipStream_t)stream);
}

int gx_modconv_up_convt(const float* x, int B, int H, int W, int Cin, const float* wt, int Cout, const float* s,
                        float* T, float* slab, size_t slab_bytes, void* stream) {
    return gx_modconv_up_convt_p(x, B, H, W, Cin, wt, nullptr, Cout, s, T, slab, slab_bytes, stream);
}

int gx_modconv_up_fwd(const float* x, int B, int H, int W, int Cin, const float* wt, int Cout, const float* s,
                      const float* demod, const float* kf, const float* noise, long noise_bstride,
                      const float* noise_w, const float* bias, int act, float* T, float* out, float* slab,
                      size_t slab_bytes, void* stream) {
    if (!out || !kf || (Cout & 3)) return GX_ERR_BAD_ARG;
    const int rc = gx_modconv_up_convt(x, B, H, W, Cin, wt, Cout, s, T, slab, slab_bytes, stream);
    if (rc) return rc;
    return gx_blur_act_fwd(T, B, 2 * H + 1, 2 * W + 1, Cout, kf, demod, noise, noise_bstride, noise_w, bias, act, out,
                           stream);
}

int gx_torgb_fwd(const float* x, int B, int H, int W, int Cin, const float* w3, const float* s, float scale,
                 const float* bias3, const float* prev_img, const float* kf, float* img, void* stream) {
    if (!x || !w3 || !s || !bias3 || !img || B <= 0 || H <= 0 || W <= 0 || Cin <= 0) return GX_ERR_BAD_ARG;
    if (prev_img && (!kf || (H & 1) || (W & 1))) return GX_ERR_BAD_ARG;
    int lp = 4;                                   // lanes per pixel: >= 4 (three output lanes), <= 64
    while (lp < 64 && lp * 4 < Cin) lp <<= 1;
    const long waves = ((long)B * H * W + (64 / lp) - 1) / (64 / lp);
    const int ni = (Cin + lp * 4 - 1) / (lp * 4);
    if ((Cin & 3) == 0 && ni <= 4 && (((uintptr_t)x | (uintptr_t)s | (uintptr_t)w3) & 15) == 0 &&
        (long)B * H * W < (1L << 31)) {
        long nb = gx_cdiv(gx_cdiv(waves, TG_R), 4);
        if (nb > 8192) nb = 8192;
#define GX_TORGB(NI_) hipLaunchKernelGGL((torgb_stream_kernel<NI_>), dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, \
                                         s, w3, scale, bias3, prev_img, kf, B, H, W, Cin, lp, img)
        if (ni == 1) GX_TORGB(1); else if (ni == 2) GX_TORGB(2); else GX_TORGB(4);
#undef GX_TORGB
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    hipLaunchKernelGGL(torgb_kernel, dim3(gx_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, x, s, w3, scale,
                       bias3, prev_img, kf, B, H, W, Cin, lp, img);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_upfirdn2d_out_shape(int in_h, int in_w, int kernel_h, int kernel_w, int up_x, int up_y, int down_x, int down_y,
                           int pad_x0, int pad_x1, int pad_y0, int pad_y1, int* out_h, int* out_w) {
    if (!out_h || !out_w || in_h <= 0 || in_w <= 0 || kernel_h <= 0 || kernel_w <= 0 || up_x <= 0 || up_y <= 0 ||
        down_x <= 0 || down_y <= 0)
        return GX_ERR_BAD_ARG;
    *out_h = (in_h * up_y + pad_y0 + pad_y1 - kernel_h + down_y) / down_y;
    *out_w = (in_w * up_x + pad_x0 + pad_x1 - kernel_w + down_x) / down_x;
    return (*out_h > 0 && *out_w > 0) ? GX_OK : GX_ERR_BAD_ARG;
}

int gx_upfirdn2d(const float* input, int major, int in_h, int in_w, int minor, const float* kernel, int kernel_h,
                 int kernel_w, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0,
                 int pad_y1, float* out, void* stream) {
    int oh = 0, ow = 0;
    if (!input || !kernel || !out || major <= 0 || minor <= 0) return GX_ERR_BAD_ARG;
    const int rc = gx_upfirdn2d_out_shape(in_h, in_w, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
                                          pad_y0, pad_y1, &oh, &ow);
    if (rc) return rc;
    const long total = (long)major * oh * ow * minor;
    long want = (total + 255) / 256;
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3((int)(want < 16384 ? want : 16384)), dim3(256), 0, (hipStream_t)stream,
                       input, kernel, major, in_h, in_w, minor, kernel_h, kernel_w, up_x, up_y, down_x, down_y, pad_x0,
                       pad_y0, oh, ow, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_fused_bias_act(const float* input, long n, const float* bias, int size_b, long step_b, const float* refer,
                      int act, int grad, float alpha, float scale, float* out, void* stream) {
    if (!input || !out || n <= 0 || (act != 1 && act != 3) || grad < 0 || grad > 2) return GX_ERR_BAD_ARG;
    if (bias && (size_b <= 0 || step_b <= 0)) return GX_ERR_BAD_ARG;
    long want = (n + 255) / 256;
    hipLaunchKernelGGL(fused_bias_act_kernel, dim3((int)(want < 16384 ? want : 16384)), dim3(256), 0,
                       (hipStream_t)stream, input, n, bias, size_b, step_b, refer, act, grad, alpha, scale, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_pixel_norm(const float* x, long rows, int D, float* y, void* stream) {
    if (!x || !y || rows <= 0 || D <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(pixel_norm_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, D, y);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_bias_act(float* y, long rows, int N, float alpha, const float* bias, float bscale, int act, void* stream) {
    if (!y || rows <= 0 || N <= 0) return GX_ERR_BAD_ARG;
    long want = (rows * N + 255) / 256;
    const int nb = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(bias_act_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, y, rows, N, alpha, bias, bscale, act);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// out (rows, N) = act(alpha * x (rows, K) wt (K, N) + bias * bscale): EqualLinear (+ fused_leaky_relu when act) for few rows,
// wt = the layer's weight TRANSPOSED.  K % 16 == 0, every pointer 16-B aligned; x / out dense.
int gx_equal_linear_skinny(const float* x, int rows, int K, const float* wt, int N, float alpha, const float* bias,
                           float bscale, int act, float* out, void* stream) {
    if (!x || !wt || !out || rows <= 0 || K <= 0 || N <= 0 || (K & 15) || (((uintptr_t)x | (uintptr_t)wt) & 15))
        return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(equal_linear_skinny_kernel<8>, dim3(gx_cdiv(N, 64), gx_cdiv(rows, 8)), dim3(256), 0, (hipStream_t)stream,
                       x, rows, K, wt, N, alpha, bias, bscale, act, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_lerp_rows(const float* w, const float* base, long n, int D, float psi, float* out, void* stream) {
    if (!w || !base || !out || n <= 0 || D <= 0) return GX_ERR_BAD_ARG;
    long want = (n + 255) / 256;
    const int nb = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(lerp_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, base, n, D, psi, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_nchw_to_nhwc(const float* in, int B, int C, int HW, float* out, void* stream) {
    if (!in || !out || B <= 0 || C <= 0 || HW <= 0) return GX_ERR_BAD_ARG;
    long want = ((long)B * C * HW + 255) / 256;
    const int nb = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, in, B, C, HW, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

const char* gx_version(void) { return "ganecdotes_amd-hip 0.1 (gfx950)"; }

}  // extern "C"
