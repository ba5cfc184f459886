// HBM-bound kernels of the per-pixel SwAV path (SURVEY.md §8a rows S1, S1b, S1c, S2, S4, S5,
// S6 (elementwise part), S7, S8, P1-argmax).  Layout: feature taps are NHWC ("pixel-major",
// [b][y][x][c]) so that a sampled pixel is ONE contiguous C_l*4-byte read per tap; the sampled
// panel X is (rows, hlen) row-major; scores are (rows, K) row-major.
#include "gx_common.h"
#include "gx_bigk.h"

namespace gx {

// ===================================================================== S1b: rotate+flip map
// Source-pixel map of RandomRotation(nearest, centre, zero fill) followed by
// RandomHorizontalFlip (call sites hfc_with_swav/swav_clustering.py:98-102, :358-359),
// restated from torchvision's public tensor path: inverse affine for -angle about the image
// centre on a pixel-centre grid, grid_sample(nearest, zeros, align_corners=False).
__global__ void rotflip_map_kernel(int* __restrict__ map, int H, int W, float cs, float sn, int flip) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    const int xr = flip ? (W - 1 - x) : x;              // flip is applied AFTER the rotation
    const float bx = (float)xr - 0.5f * (float)W + 0.5f;
    const float by = (float)y - 0.5f * (float)H + 0.5f;
    // normalised grid coordinates, then grid_sample's un-normalisation
    const float gxn = __fdiv_rn(__fadd_rn(__fmul_rn(bx, cs), __fmul_rn(by, sn)), 0.5f * (float)W);
    const float gyn = __fdiv_rn(__fadd_rn(__fmul_rn(bx, -sn), __fmul_rn(by, cs)), 0.5f * (float)H);
    const float ix = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(gxn, 1.f), (float)W), -1.f), 0.5f);
    const float iy = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(gyn, 1.f), (float)H), -1.f), 0.5f);
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    int src = -1;
    if (rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1)) src = (int)ry * W + (int)rx;
    map[p] = src;
}

// The maps of a whole view batch in one launch: blockIdx.y = image, per-image (cos, sin, flip) by value.
constexpr int RF_MAX = 64;
struct RotFlipBatch { float cs[RF_MAX], sn[RF_MAX]; int flip[RF_MAX]; };
__global__ void rotflip_map_batch_kernel(int* __restrict__ maps, int H, int W, RotFlipBatch rf) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (p >= H * W) return;
    const float cs = rf.cs[b], sn = rf.sn[b];
    const int y = p / W, x = p - y * W;
    const int xr = rf.flip[b] ? (W - 1 - x) : x;
    const float bx = (float)xr - 0.5f * (float)W + 0.5f;
    const float by = (float)y - 0.5f * (float)H + 0.5f;
    const float gxn = __fdiv_rn(__fadd_rn(__fmul_rn(bx, cs), __fmul_rn(by, sn)), 0.5f * (float)W);
    const float gyn = __fdiv_rn(__fadd_rn(__fmul_rn(bx, -sn), __fmul_rn(by, cs)), 0.5f * (float)H);
    const float ix = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(gxn, 1.f), (float)W), -1.f), 0.5f);
    const float iy = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(gyn, 1.f), (float)H), -1.f), 0.5f);
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    int src = -1;
    if (rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1)) src = (int)ry * W + (int)rx;
    maps[(long)b * H * W + p] = src;
}

// ===================================================================== S2: random pixel subsets
// picks[row][0..N) = the indices of the N smallest of HW i.i.d. uniform keys = a uniform random N-subset of
// the pixels (the set `randperm(HW)[:N]` draws, swav_clustering.py:388-390; its order is irrelevant to the
// loss) — by a 3-level radix SELECT (11 + 11 + 10 bits of the float bit pattern) and an order-preserving
// compaction instead of a full sort: 4 reads of the 256 KB key row, indices come out ascending
// (deterministic, and the later row gathers walk memory forwards).  One workgroup of 1024 threads per row.
__device__ __forceinline__ void rs_find_bin(const unsigned* hist, int nbins, unsigned need, unsigned* s_bin,
                                            unsigned* s_below) {
    // wave 0: lane l owns bins [l*per, (l+1)*per); first bin where the running count reaches `need`
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, per = nbins / 64;
        unsigned mine = 0;
        for (int i = 0; i < per; ++i) mine += hist[lane * per + i];
        unsigned incl = mine;
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        const unsigned excl = incl - mine;
        if (excl < need && need <= incl) {
            unsigned run = excl;
            for (int i = 0; i < per; ++i) {
                const unsigned c = hist[lane * per + i];
                if (run + c >= need) { *s_bin = lane * per + i; *s_below = run; break; }
                run += c;
            }
        }
    }
}

__global__ __launch_bounds__(1024) void random_subset_kernel(const unsigned* __restrict__ keys, int HW, int N,
                                                             int* __restrict__ picks) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned s_bin, s_below, s_w[2][16];
    const unsigned* k = keys + (long)blockIdx.x * HW;
    int* out = picks + (long)blockIdx.x * N;
    const int tid = threadIdx.x;
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)k) & 15) == 0);     // 16-byte key loads
    unsigned prefix = 0, need = (unsigned)N, below = 0;
    // three radix levels: shifts 21, 10, 0 with 11, 11, 10 bits
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
#pragma unroll
    for (int lv = 0; lv < 3; ++lv) {
        const int nb = 1 << bits[lv];
        for (int i = tid; i < 2048; i += 1024) hist[i] = 0u;
        __syncthreads();
        const int hs = shifts[lv] + bits[lv];                    // bits above this level
        auto count = [&](unsigned v) {
            if (lv == 0 || (v >> hs) == prefix) atomicAdd(&hist[(v >> shifts[lv]) & (nb - 1)], 1u);
        };
        if (vec) {
            for (int i = tid; i < (HW >> 2); i += 1024) {
                const u32x4_t v = reinterpret_cast<const u32x4_t*>(k)[i];
                count(v[0]); count(v[1]); count(v[2]); count(v[3]);
            }
        } else {
            for (int i = tid; i < HW; i += 1024) count(k[i]);
        }
        __syncthreads();
        rs_find_bin(hist, nb, need, &s_bin, &s_below);
        __syncthreads();
        prefix = (prefix << bits[lv]) | s_bin;
        below += s_below;
        need -= s_below;                                          // still to take inside the chosen bin
        __syncthreads();
    }
    const unsigned T = prefix;                                    // the N-th smallest key; `need` of the keys == T are taken
    // Order-preserving compaction, a tile of 1024 consecutive keys at a time (coalesced reads; a thread owning a contiguous
    // range of the row instead read 64 cache lines per wave instruction, twice: 1.8 of the 2.0 ms the kernel took on the car
    // configuration's 262 144-pixel rows): positions from wave ballots + the 16 waves' counts; the keys < T come out ascending
    // at out[0 .. below), the first `need` of the keys == T behind them.
    const int lane = tid & 63, wv = tid >> 6;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    unsigned base_less = 0, base_tie = 0;
    for (int t0 = 0; t0 < HW; t0 += 1024) {
        const int i = t0 + tid;
        const unsigned v = i < HW ? k[i] : 0xFFFFFFFFu;           // (keys are fp32 in [0, 1): never 0xFFFFFFFF)
        const bool less = v < T, tie = v == T && i < HW;
        const unsigned long long bl = __ballot(less), bt = __ballot(tie);
        if (lane == 0) { s_w[0][wv] = (unsigned)__popcll(bl); s_w[1][wv] = (unsigned)__popcll(bt); }
        __syncthreads();
        unsigned ol = base_less + (unsigned)__popcll(bl & lt_mask), ot = base_tie + (unsigned)__popcll(bt & lt_mask);
        unsigned tl = 0, tt = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned cl = s_w[0][w], ct = s_w[1][w];
            if (w < wv) { ol += cl; ot += ct; }
            tl += cl; tt += ct;
        }
        if (less) out[ol] = i;
        else if (tie && ot < need) out[below + ot] = i;
        base_less += tl;
        base_tie += tt;
        __syncthreads();                                          // s_w is rewritten by the next tile
    }
}

// ===================================================================== S1 + S1b + S2: gather
constexpr int MAX_TAPS = 16;
struct TapList {
    const float* ptr[MAX_TAPS];   // [B][res][res][C] NHWC
    int C[MAX_TAPS];              // channels of the tap
    int res[MAX_TAPS];            // spatial size (square)
    int off[MAX_TAPS];            // channel offset in the concatenated descriptor
    int n;
};

// X[row][:] = concat_l tap_l[b][(sy*res_l)/H][(sx*res_l)/W][:]  (first hlen channels), where
// (sy,sx) = map_b[pick] (or the pick itself), zeros where the map says "filled".
// Never materialises the (1, hlen, H, W) stack of swav_clustering.py:108-130 / :160-167.
// F.interpolate(mode='bilinear', align_corners=False) source of output coordinate d on an axis of `in` samples upsampled to
// `out` (ATen area_pixel_compute_source_index + the lambdas of upsample_bilinear2d): src = in / out * (d + 0.5) - 0.5 clamped
// at 0, i0 = floor(src), i1 = i0 + (i0 < in - 1), l1 = src - i0.
__device__ __forceinline__ void gx_bilinear_axis(int d, int in, int out, int& i0, int& i1, float& l0, float& l1) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void gather_taps_kernel(TapList taps, const int* __restrict__ picks,
                                                          const int* __restrict__ maps, int rows_per_img,
                                                          int H, int W, int hlen, int interp, float* __restrict__ X,
                                                          long ldx, int total_rows) {
    const int row = blockIdx.x;
    if (row >= total_rows) return;
    const int b = row / rows_per_img;
    int pix = picks[row];
    if (maps) pix = maps[(long)b * H * W + pix];
    float* out = X + (long)row * ldx;
    const bool vec = ((ldx & 3) == 0) && ((((uintptr_t)X) & 15) == 0);
    const int sy = pix / W, sx = pix - sy * W;
    for (int l = 0; l < taps.n; ++l) {
        const int off = taps.off[l];
        if (off >= hlen) break;
        const int cl = min(taps.C[l], hlen - off);
        const int r = taps.res[l];
        if (interp == 1 && pix >= 0 && (r != H || r != W)) {
            // hf_interp = 'bilinear' (swav_clustering.py:123-124): h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11), ATen's order
            int y0, y1, x0, x1;
            float h0, h1, w0, w1;
            gx_bilinear_axis(sy, r, H, y0, y1, h0, h1);
            gx_bilinear_axis(sx, r, W, x0, x1, w0, w1);
            const float* base = taps.ptr[l] + (long)b * r * r * taps.C[l];
            const float* p00 = base + ((long)y0 * r + x0) * taps.C[l];
            const float* p01 = base + ((long)y0 * r + x1) * taps.C[l];
            const float* p10 = base + ((long)y1 * r + x0) * taps.C[l];
            const float* p11 = base + ((long)y1 * r + x1) * taps.C[l];
            for (int c = threadIdx.x; c < cl; c += blockDim.x)
                out[off + c] = h0 * (w0 * p00[c] + w1 * p01[c]) + h1 * (w0 * p10[c] + w1 * p11[c]);
            continue;
        }
        const float* src = nullptr;
        if (pix >= 0) {
            const int ty = (sy * r) / H, tx = (sx * r) / W;
            src = taps.ptr[l] + (((long)b * r + ty) * r + tx) * taps.C[l];
        }
        if (vec && (cl & 3) == 0 && (off & 3) == 0 && (taps.C[l] & 3) == 0) {
            for (int c = threadIdx.x * 4; c < cl; c += blockDim.x * 4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (src) v = *reinterpret_cast<const f32x4*>(src + c);
                *reinterpret_cast<f32x4*>(out + off + c) = v;
            }
        } else {
            for (int c = threadIdx.x; c < cl; c += blockDim.x) out[off + c] = src ? src[c] : 0.f;
        }
    }
}

// img[b][p] = ||descriptor of (rotated / flipped) pixel p||_2 with the taps BILINEARLY upsampled (hf_interp = 'bilinear'): the
// per-tap sums of squares of norm_map_kernel do not apply — an interpolated value is not a tap value.  One wave per pixel.
__global__ __launch_bounds__(256) void stack_norm_bilinear_kernel(TapList taps, const int* __restrict__ maps, int B, int H, int W,
                                                                  int hlen, float* __restrict__ img) {
    const long i = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= (long)B * H * W) return;
    const int lane = threadIdx.x & 63;
    const int b = (int)(i / (H * W)), p = (int)(i - (long)b * H * W);
    const int pix = maps ? maps[i] : p;
    float s = 0.f;
    if (pix >= 0) {
        const int sy = pix / W, sx = pix - sy * W;
        for (int l = 0; l < taps.n; ++l) {
            const int off = taps.off[l];
            if (off >= hlen) break;
            const int cl = min(taps.C[l], hlen - off), r = taps.res[l];
            int y0, y1, x0, x1;
            float h0, h1, w0, w1;
            gx_bilinear_axis(sy, r, H, y0, y1, h0, h1);
            gx_bilinear_axis(sx, r, W, x0, x1, w0, w1);
            const float* base = taps.ptr[l] + (long)b * r * r * taps.C[l];
            const float* p00 = base + ((long)y0 * r + x0) * taps.C[l];
            const float* p01 = base + ((long)y0 * r + x1) * taps.C[l];
            const float* p10 = base + ((long)y1 * r + x0) * taps.C[l];
            const float* p11 = base + ((long)y1 * r + x1) * taps.C[l];
            for (int c = lane; c < cl; c += 64) {
                const float v = (r == H && r == W) ? p00[c] : h0 * (w0 * p00[c] + w1 * p01[c]) + h1 * (w0 * p10[c] + w1 * p11[c]);
                s = fmaf(v, v, s);
            }
        }
    }
    s = gx_wave_sum(s);
    if (lane == 0) img[i] = sqrtf(s);
}

// ===================================================================== row reductions
// out[row] = sum_c x[row][c]^2 over the first `cused` of `ld` columns.  One wave per row.
__global__ void row_sumsq_kernel(const float* __restrict__ x, long rows, int cused, long ld,
                                 float* __restrict__ out) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = x + row * ld;
    float s = 0.f;
    for (int c = lane; c < cused; c += 64) { const float v = p[c]; s = fmaf(v, v, s); }
    s = gx_wave_sum(s);
    if (lane == 0) out[row] = s;
}
// ... for 16-byte aligned rows of a multiple of 4 columns: a TEAM of lanes per row (64 / TEAM rows per wave), 16 bytes per lane
// and load.  The norm maps of the 'image' marginals sum every tap of both views (3.9 GB per bench step): with a wave per row
// and dword loads a 128-channel row of the finest taps kept a wave busy for 512 bytes — 2.7 TB/s
template <int TEAM>
__global__ void row_sumsq_vec_kernel(const float* __restrict__ x, long rows, int cused, long ld, float* __restrict__ out) {
    const int tl = threadIdx.x % TEAM;
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) / TEAM;
    float s = 0.f;
    if (row < rows) {
        const float* p = x + row * ld;
        for (int c = 4 * tl; c < cused; c += 4 * TEAM) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + c);
            s = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], s))));
        }
    }
    s = gx_team_sum(s, TEAM);
    if (row < rows && tl == 0) out[row] = s;
}

// S1c: img[b][p] = sqrt(sum_l ss_l[b][src(p) -> tap pixel])  (torch.norm(stack, dim=1) after the
// rotate/flip, swav_clustering.py:361-362), from per-tap native-resolution sums of squares.
struct SsList { const float* ptr[MAX_TAPS]; int res[MAX_TAPS]; int n; };
__global__ void norm_map_kernel(SsList ss, const int* __restrict__ maps, int B, int H, int W,
                                float* __restrict__ img) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * H * W) return;
    const int b = (int)(i / (H * W)), p = (int)(i - (long)b * H * W);
    const int pix = maps ? maps[i] : p;
    float s = 0.f;
    if (pix >= 0) {
        const int sy = pix / W, sx = pix - sy * W;
        for (int l = 0; l < ss.n; ++l) {
            const int r = ss.res[l];
            s += ss.ptr[l][((long)b * r + (sy * r) / H) * r + (sx * r) / W];
        }
    }
    img[i] = sqrtf(s);
}

// S3 normalise: z[row] <- z[row] / max(||z[row]||, 1e-12) in place; rnorm[row] = the factor.
// (F.normalize(p=2, dim=1), swav_clustering.py:174).  One wave per row.
__global__ void row_l2_normalize_kernel(float* __restrict__ z, long rows, int D, long ld,
                                        float* __restrict__ rnorm) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* p = z + row * ld;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = p[c]; s = fmaf(v, v, s); }
    s = gx_wave_sum(s);
    const float rn = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int c = lane; c < D; c += 64) p[c] *= rn;
    if (rnorm && lane == 0) rnorm[row] = rn;
}

// S6 normalise-backward, in place on g (= d z_hat -> d z):
//   dz = rnorm * (dzh - zh * <zh, dzh>)      (norm > 1e-12 branch of F.normalize)
__global__ void l2_normalize_bwd_kernel(float* __restrict__ g, const float* __restrict__ zh,
                                        const float* __restrict__ rnorm, long rows, int D, long ld) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float* gp = g + row * ld;
    const float* zp = zh + row * ld;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot = fmaf(gp[c], zp[c], dot);
    dot = gx_wave_sum(dot);
    const float rn = rnorm[row];
    const bool clamped = rn >= 1e12f;       // ||z|| <= 1e-12: normalize divided by the constant eps
    for (int c = lane; c < D; c += 64) gp[c] = clamped ? gp[c] * rn : rn * (gp[c] - zp[c] * dot);
}

// P1: labels[row] = argmax_c z[row][c] (first index on ties), swav_clustering.py:691.
__global__ void row_argmax_kernel(const float* __restrict__ z, long rows, int D, long ld,
                                  long long* __restrict__ labels) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* p = z + row * ld;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < D; c += 64) { const float v = p[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) labels[row] = bi;
}

// column sums over rows: part[block][k] = sum_{rows of block} x[row][k]; then reduce.
__global__ void colsum_partial_kernel(const float* __restrict__ x, long rows, int K, long ld,
                                      float* __restrict__ part) {
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float s = 0.f;
        for (long r = r0; r < r1; ++r) s += x[r * ld + k];
        part[(long)blockIdx.x * K + k] = s;
    }
}
// out[k] = alpha * sum_b part[b][k] + beta * out[k].  Block = kx columns x (256/kx) partial
// groups: coalesced across k, fixed-order tree over the groups (bitwise reproducible).
__global__ void partial_reduce_kernel(const float* __restrict__ part, int nblk, int K, float alpha, float beta,
                                      float* __restrict__ out, int kx) {
    __shared__ float red[256];
    part += (long)blockIdx.y * nblk * K;                          // blockIdx.y = problem (batched callers)
    out += (long)blockIdx.y * K;
    const int tx = threadIdx.x % kx, ty = threadIdx.x / kx, ny = blockDim.x / kx;
    const int k = blockIdx.x * kx + tx;
    float s = 0.f;
    if (k < K) for (int b = ty; b < nblk; b += ny) s += part[(long)b * K + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = ny >> 1; h > 0; h >>= 1) {
        if (ty < h) red[threadIdx.x] += red[threadIdx.x + h * kx];
        __syncthreads();
    }
    if (ty == 0 && k < K) out[k] = alpha * red[tx] + (beta != 0.f ? beta * out[k] : 0.f);
}
// out2[g][k] = sum of the g-th slice of partial rows (fixed order); blockIdx.y = slice
__global__ void partial_reduce_slices_kernel(const float* __restrict__ part, int nblk, int K, int per,
                                             float* __restrict__ out2) {
    __shared__ float red[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + tx;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float s = 0.f;
    if (k < K) for (int b = b0 + ty; b < b1; b += 4) s += part[(long)b * K + k];
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && k < K) out2[(long)blockIdx.y * K + k] = (red[tx] + red[tx + 64]) + (red[tx + 128] + red[tx + 192]);
}
// `spare` (optional): >= 8*K floats of scratch; with it, many partial rows of many columns are reduced in two stages
// (8 row slices x K/64 column blocks fill the chip: one stage ran 79 blocks over 40 MB at 0.25 TB/s for K = 5000)
void launch_partial_reduce(const float* part, int nblk, int K, float alpha, float beta, float* out, hipStream_t stream,
                           float* spare) {
    int kx = 1;
    while (kx < 64 && kx < K) kx <<= 1;
    if (spare && K >= 1024 && nblk >= 64) {
        const int per = (nblk + 7) / 8;
        hipLaunchKernelGGL(partial_reduce_slices_kernel, dim3((K + 63) / 64, 8), dim3(256), 0, stream, part, nblk, K, per,
                           spare);
        hipLaunchKernelGGL(partial_reduce_kernel, dim3((K + kx - 1) / kx), dim3(256), 0, stream, spare, 8, K, alpha, beta,
                           out, kx);
        return;
    }
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((K + kx - 1) / kx), dim3(256), 0, stream, part, nblk, K, alpha,
                       beta, out, kx);
}

// ===================================================================== S4: Sinkhorn-Knopp
// Log-domain scaling-vector form of sinkhorn_knopp (swav_clustering.py:509-544).  With
// E_kn = exp(S_nk/eps) the reference's iterate is Q = diag(a) E diag(b): each iteration sets
// a_k = r_k / sum_n E_kn b_n (row normalisation) then b_n = c_n / sum_k a_k E_kn (column
// normalisation), and the returned code is q[n][k] = a_k E_kn / sum_k' a_k' E_k'n — a softmax
// over k of  w_nk = S_nk/eps + la_k  with la = log a (b and the initial 1/sum(Q) cancel).
// We never form Q and carry la_k / lb_n = log b_n instead of a, b, so nothing overflows or
// underflows where the reference's un-shifted fp32 exp() would:
//   colmax : cmax_k = max_n S_nk                                  (la_k := -cmax_k/eps)
//   first  : u_k = sum_n exp((S_nk - cmax_k)/eps)        [= row sums of Q0 up to exp(la)]
//   sweep  : per sample n: lb_n = log c_n - logsumexp_k(w_nk);  u_k = sum_n exp(w_nk + lb_n)
//   update : la_k += log r_k - log u_k      (after an all-reduce of u over ranks when sharded)
// Every term of every sum is <= 1.  Each pass is one read of the score matrix; a row is
// handled by a team of T lanes (T = 8..64), column sums go through LDS then fixed-order
// per-block partials.

__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void sk_colmax_kernel(const float* __restrict__ S, long rows, int K, long ld,
                                 unsigned* __restrict__ cmax) {
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
    if (r0 >= r1) return;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        float v = -INFINITY;
        for (long r = r0; r < r1; ++r) v = fmaxf(v, S[r * ld + k]);
        atomicMax(&cmax[k], f2ord(v));
    }
}
__global__ void sk_ord_decode_kernel(const unsigned* __restrict__ in, int K, float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = ord2f(in[k]);
}
__global__ void sk_ord_encode_kernel(const float* __restrict__ in, int K, unsigned* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = f2ord(in[k]);
}

// One Sinkhorn pass (see above).  LDS: K floats of la (FIRST: -cmax/eps) + K floats of
// block-local u accumulators.
template <bool FIRST>
__global__ void sk_pass_kernel(const float* __restrict__ S, long rows, int K, long ld, int team,
                               float inv_eps, const unsigned* __restrict__ cmax,
                               const float* __restrict__ la, const float* __restrict__ c, float c_uniform,
                               float* __restrict__ lb_out, float* __restrict__ upart) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* uacc = sm;
    float* las = sm + K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        uacc[k] = 0.f;
        las[k] = FIRST ? -ord2f(cmax[k]) * inv_eps : la[k];
    }
    __syncthreads();
    const int tpb = blockDim.x / team;
    const int ti = threadIdx.x / team, tl = threadIdx.x % team;
    const long nrow_iters = (rows + (long)gridDim.x * tpb - 1) / ((long)gridDim.x * tpb);
    for (long it = 0; it < nrow_iters; ++it) {
        const long row = (it * gridDim.x + blockIdx.x) * tpb + ti;
        const bool live = row < rows;
        const float* p = S + (live ? row : 0) * ld;
        float lb = 0.f;
        if (!FIRST) {
            float mx = -INFINITY;
            if (live) for (int k = tl; k < K; k += team) mx = fmaxf(mx, fmaf(p[k], inv_eps, las[k]));
            mx = gx_team_max(mx, team);
            float sum = 0.f;
            if (live) for (int k = tl; k < K; k += team) sum += expf(fmaf(p[k], inv_eps, las[k]) - mx);
            sum = gx_team_sum(sum, team);
            lb = live ? logf(c ? c[row] : c_uniform) - mx - logf(sum) : 0.f;
            if (live && tl == 0 && lb_out) lb_out[row] = lb;
        }
        // column accumulation: lanes of different teams holding the same k are summed by
        // shuffles (teams inside one wave) before one LDS atomic per (wave, k).
        for (int k0 = 0; k0 < K; k0 += team) {
            const int k = k0 + tl;
            float e = 0.f;
            if (live && k < K) e = expf(fmaf(p[k], inv_eps, las[k]) + lb);
            for (int o = team; o < 64; o <<= 1) e += __shfl_xor(e, o, 64);
            if ((threadIdx.x & 63) < team && k < K) atomicAdd(&uacc[k], e);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) upart[(long)blockIdx.x * K + k] = uacc[k];
}

// la_k = (first ? -cmax_k/eps : la_k) + log r_k - log u_k     (r uniform = 1/K when r == nullptr)
__global__ void sk_update_la_kernel(const float* __restrict__ u, const float* __restrict__ r, int K, int first,
                                    const unsigned* __restrict__ cmax, float inv_eps, float* __restrict__ la,
                                    long r_stride = 0) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    {   // blockIdx.y = problem of a batch: K-vectors K apart, prototype marginals r_stride apart
        const long prob = blockIdx.y;
        u += prob * K; la += prob * K;
        if (cmax) cmax += prob * K;
        if (r) r += prob * r_stride;
    }
    const float base = first ? -ord2f(cmax[k]) * inv_eps : la[k];
    la[k] = base + logf(r ? r[k] : 1.f / (float)K) - logf(u[k]);
}

// codes q[n][k] = softmax_k(S_nk/eps + la_k)   (the matrix sinkhorn_knopp returns)
__global__ void sk_codes_kernel(const float* __restrict__ S, long rows, int K, long ld, int team,
                                float inv_eps, const float* __restrict__ la, float* __restrict__ q, long ldq) {
    const int tpb = blockDim.x / team;
    const int ti = threadIdx.x / team, tl = threadIdx.x % team;
    const long nrow_iters = (rows + (long)gridDim.x * tpb - 1) / ((long)gridDim.x * tpb);
    for (long it = 0; it < nrow_iters; ++it) {
        const long row = (it * gridDim.x + blockIdx.x) * tpb + ti;
        const bool live = row < rows;
        const float* p = S + (live ? row : 0) * ld;
        float mx = -INFINITY;
        if (live) for (int k = tl; k < K; k += team) mx = fmaxf(mx, fmaf(p[k], inv_eps, la[k]));
        mx = gx_team_max(mx, team);
        float sum = 0.f;
        if (live) for (int k = tl; k < K; k += team) sum += expf(fmaf(p[k], inv_eps, la[k]) - mx);
        sum = gx_team_sum(sum, team);
        if (live) for (int k = tl; k < K; k += team) q[row * ldq + k] = expf(fmaf(p[k], inv_eps, la[k]) - mx) / sum;
    }
}

// ===================================================================== S5 + S6: swapped CE
// Per row n (p = S/T, q = softmax_k(S/eps + la) from the Sinkhorn log-scaling vectors):
//   loss_n = -1/2 [ sum_k q_s (p_t - lse_t) + sum_k q_t (p_s - lse_s) ]
//   dS_t   = -gscale/(2T) (q_s - softmax(p_t) * sum_k q_s),  dS_s likewise with q_t
// (calculate_swapped_prediction_loss, swav_clustering.py:547-570 with :408-409, and its
// autograd).  gscale = 1/(N_total * num_patches [* num_images]); loss partial sums are per
// block (fixed order) and are NOT yet scaled.
// blockIdx.y = pair (batched form): pair y = (patch y / ngroups, row group y % ngroups); see gx_swapped_ce_batch.
struct CePairs { int ngroups; long patch_stride, group_stride, la_patch_stride, la_group_stride; };
__global__ void swapped_ce_kernel(const float* __restrict__ Ss, const float* __restrict__ St, long rows, int K,
                                  long ld, int team, float inv_eps, float inv_T,
                                  const float* __restrict__ las, const float* __restrict__ lat, float gscale,
                                  float* __restrict__ dSs, float* __restrict__ dSt, long ldd,
                                  float* __restrict__ loss_part, CePairs pr) {
    {
        const long pa = blockIdx.y / pr.ngroups, gr = blockIdx.y % pr.ngroups;
        const long so = pa * pr.patch_stride + gr * pr.group_stride, lo = pa * pr.la_patch_stride + gr * pr.la_group_stride;
        Ss += so; St += so; las += lo; lat += lo;
        if (dSs) { dSs += so; dSt += so; }
        loss_part += (long)blockIdx.y * gridDim.x;
    }
    const int tpb = blockDim.x / team;
    const int ti = threadIdx.x / team, tl = threadIdx.x % team;
    float lacc = 0.f;
    const long nrow_iters = (rows + (long)gridDim.x * tpb - 1) / ((long)gridDim.x * tpb);
    for (long it = 0; it < nrow_iters; ++it) {
        const long row = (it * gridDim.x + blockIdx.x) * tpb + ti;
        const bool live = row < rows;
        const float* ps = Ss + (live ? row : 0) * ld;
        const float* pt = St + (live ? row : 0) * ld;
        // row maxima of the Sinkhorn logits w = S/eps + la and of the scores themselves
        float wms = -INFINITY, wmt = -INFINITY, m_s = -INFINITY, m_t = -INFINITY;
        if (live) for (int k = tl; k < K; k += team) {
            wms = fmaxf(wms, fmaf(ps[k], inv_eps, las[k])); wmt = fmaxf(wmt, fmaf(pt[k], inv_eps, lat[k]));
            m_s = fmaxf(m_s, ps[k]); m_t = fmaxf(m_t, pt[k]);
        }
        wms = gx_team_max(wms, team); wmt = gx_team_max(wmt, team);
        m_s = gx_team_max(m_s, team); m_t = gx_team_max(m_t, team);
        float vs = 0.f, vt = 0.f, zs = 0.f, zt = 0.f;
        if (live) for (int k = tl; k < K; k += team) {
            vs += expf(fmaf(ps[k], inv_eps, las[k]) - wms);
            vt += expf(fmaf(pt[k], inv_eps, lat[k]) - wmt);
            zs += expf((ps[k] - m_s) * inv_T);
            zt += expf((pt[k] - m_t) * inv_T);
        }
        vs = gx_team_sum(vs, team); vt = gx_team_sum(vt, team);
        zs = gx_team_sum(zs, team); zt = gx_team_sum(zt, team);
        const float lzs = logf(zs), lzt = logf(zt);
        float qsum_s = 0.f, qsum_t = 0.f, l = 0.f;
        if (live) for (int k = tl; k < K; k += team) {
            const float qs = expf(fmaf(ps[k], inv_eps, las[k]) - wms) / vs;
            const float qt = expf(fmaf(pt[k], inv_eps, lat[k]) - wmt) / vt;
            qsum_s += qs; qsum_t += qt;
            // log_softmax(p)[k] = (S_k - m)/T - log Z
            l += qs * ((pt[k] - m_t) * inv_T - lzt) + qt * ((ps[k] - m_s) * inv_T - lzs);
        }
        qsum_s = gx_team_sum(qsum_s, team); qsum_t = gx_team_sum(qsum_t, team);
        if (live && dSs) {
            const float f = -0.5f * gscale * inv_T;
            for (int k = tl; k < K; k += team) {
                const float qs = expf(fmaf(ps[k], inv_eps, las[k]) - wms) / vs;
                const float qt = expf(fmaf(pt[k], inv_eps, lat[k]) - wmt) / vt;
                dSt[row * ldd + k] = f * (qs - expf((pt[k] - m_t) * inv_T) / zt * qsum_s);
                dSs[row * ldd + k] = f * (qt - expf((ps[k] - m_s) * inv_T) / zs * qsum_t);
            }
        }
        lacc += l;
    }
    __shared__ float red[16];
    const float tot = gx_block_sum(lacc, red);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = -0.5f * tot;
}

// swapped_ce_kernel with a ROW PER LANE for K = 4 KQ <= 16 (16-B aligned rows): both views' rows and their exponentials
// stay in registers — 4 K exponentials per row pair instead of 12 K, no team shuffles.  Same arguments and partials.
template <int KQ>
__global__ __launch_bounds__(256) void swapped_ce_rowlane_kernel(const float* __restrict__ Ss, const float* __restrict__ St,
                                                                 long rows, long ld, float inv_eps, float inv_T,
                                                                 const float* __restrict__ las, const float* __restrict__ lat,
                                                                 float gscale, float* __restrict__ dSs, float* __restrict__ dSt,
                                                                 long ldd, float* __restrict__ loss_part, CePairs pr) {
    constexpr int K = 4 * KQ;
    {
        const long pa = blockIdx.y / pr.ngroups, gr = blockIdx.y % pr.ngroups;
        const long so = pa * pr.patch_stride + gr * pr.group_stride, lo = pa * pr.la_patch_stride + gr * pr.la_group_stride;
        Ss += so; St += so; las += lo; lat += lo;
        if (dSs) { dSs += so; dSt += so; }
        loss_part += (long)blockIdx.y * gridDim.x;
    }
    float ls[K], lt[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { ls[k] = las[k]; lt[k] = lat[k]; }
    const float f = -0.5f * gscale * inv_T;
    float lacc = 0.f;
    for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long)gridDim.x * 256) {
        float a[K], b[K];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(Ss + row * ld + 4 * q);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(St + row * ld + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[4 * q + e] = va[e]; b[4 * q + e] = vb[e]; }
        }
        float wms = -INFINITY, wmt = -INFINITY, m_s = -INFINITY, m_t = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            wms = fmaxf(wms, fmaf(a[k], inv_eps, ls[k])); wmt = fmaxf(wmt, fmaf(b[k], inv_eps, lt[k]));
            m_s = fmaxf(m_s, a[k]); m_t = fmaxf(m_t, b[k]);
        }
        float qs[K], qt[K], es[K], et[K];
        float vs = 0.f, vt = 0.f, zs = 0.f, zt = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            qs[k] = bk_exp(fmaf(a[k], inv_eps, ls[k]) - wms); vs += qs[k];
            qt[k] = bk_exp(fmaf(b[k], inv_eps, lt[k]) - wmt); vt += qt[k];
            es[k] = bk_exp((a[k] - m_s) * inv_T); zs += es[k];
            et[k] = bk_exp((b[k] - m_t) * inv_T); zt += et[k];
        }
        const float ivs = 1.f / vs, ivt = 1.f / vt, izs = 1.f / zs, izt = 1.f / zt, lzs = logf(zs), lzt = logf(zt);
        float qsum_s = 0.f, qsum_t = 0.f, l = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            qs[k] *= ivs; qt[k] *= ivt;
            qsum_s += qs[k]; qsum_t += qt[k];
            l += qs[k] * ((b[k] - m_t) * inv_T - lzt) + qt[k] * ((a[k] - m_s) * inv_T - lzs);
        }
        lacc += l;
        if (dSs) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                f32x4 d_t, d_s;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * q + e;
                    d_t[e] = f * (qs[k] - et[k] * izt * qsum_s);
                    d_s[e] = f * (qt[k] - es[k] * izs * qsum_t);
                }
                *reinterpret_cast<f32x4*>(dSt + row * ldd + 4 * q) = d_t;
                *reinterpret_cast<f32x4*>(dSs + row * ldd + 4 * q) = d_s;
            }
        }
    }
    __shared__ float red[16];
    const float tot = gx_block_sum(lacc, red);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = -0.5f * tot;
}

// ===================================================================== S7 / S8
// LARC(clip=False) around momentum SGD (swav_clustering.py:286-292; apex LARC.step restated):
//   g' = g * trust*|p| / (|g| + 1e-8)  if |p| != 0 and |g| != 0 ;  buf = first ? g' : mom*buf + g'
//   p -= lr * buf.       norms2 = {sum p^2, sum g^2} on the device (no host sync).
__global__ void larc_sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                long n, const float* __restrict__ norms2, float lr, float momentum,
                                float trust, float eps, int first, int use_larc) {
    float scale = 1.f;
    if (use_larc) {
        const float pn = sqrtf(norms2[0]), gn = sqrtf(norms2[1]);
        if (pn != 0.f && gn != 0.f) scale = trust * pn / (gn + eps);
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gg = g[i] * scale;
        const float b = first ? gg : fmaf(momentum, buf[i], gg);
        buf[i] = b;
        p[i] = fmaf(-lr, b, p[i]);
    }
}

__global__ void sumsq_partial_kernel(const float* __restrict__ x, long n, float* __restrict__ part) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i]; s = fmaf(v, v, s);
    }
    __shared__ float red[16];
    s = gx_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ===================================================================== histc ('image' pdf)
__global__ void minmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ mm) {
    float lo = INFINITY, hi = -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        lo = fminf(lo, x[i]); hi = fmaxf(hi, x[i]);
    }
    __shared__ float red[16];
    hi = gx_block_max(hi, red);
    lo = -gx_block_max(-lo, red);
    if (threadIdx.x == 0) { atomicMax(&mm[1], f2ord(hi)); atomicMin(&mm[0], f2ord(lo)); }
}
// torch.histc(x, bins) with the data range: pos = int((x-lo)/(hi-lo)*bins), last bin closed.
// LDSB > 0 (bins <= LDSB): the block counts in LDS (integers) and adds its counts to the global histogram once — with the K
// bins of a few-prototype 'image' marginal (16 for the cat configuration) a million float atomics on 16 addresses took 5 - 10 ms
// per call, 21 ms of a 52 ms step; LDSB = 0: straight global atomics (the per-sample histogram's hundreds of thousands of bins).
constexpr int HISTC_LDS_BINS = 4096;
template <int LDSB>
__global__ void histc_kernel(const float* __restrict__ x, long n, const unsigned* __restrict__ mm, int bins,
                             float* __restrict__ hist) {
    __shared__ unsigned cnt[LDSB > 0 ? LDSB : 1];
    const float lo = ord2f(mm[0]), hi = ord2f(mm[1]);
    if (LDSB > 0) {
        for (int i = threadIdx.x; i < bins; i += blockDim.x) cnt[i] = 0u;
        __syncthreads();
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int pos = (hi > lo) ? (int)((x[i] - lo) / (hi - lo) * (float)bins) : 0;
        pos = min(max(pos, 0), bins - 1);
        if (LDSB > 0) atomicAdd(&cnt[pos], 1u);
        else atomicAdd(&hist[pos], 1.0f);
    }
    if (LDSB > 0) {
        __syncthreads();
        for (int i = threadIdx.x; i < bins; i += blockDim.x)
            if (cnt[i]) atomicAdd(&hist[i], (float)cnt[i]);          // (counts below 2^24: the same sum as one by one)
    }
}
// Many bins (the per-sample marginal: one bin per sample, 320 000 at the bench batch): scattered global float atomics run at about
// 2 G/s on this chip (every lane its own cache line, served beyond the XCD's L2) — 0.5 ms for a million values.  Instead a block
// OWNS a range of HISTC_RANGE bins, counts in LDS while it scans ALL the values (4 MB from L2 / the Infinity Cache, once per range)
// and writes its bins with plain stores: no global atomic, no memset of the histogram.
constexpr int HISTC_RANGE = 8192;
__global__ __launch_bounds__(1024) void histc_ranges_kernel(const float* __restrict__ x, long n, const unsigned* __restrict__ mm, int bins,
                                                            float* __restrict__ hist) {
    __shared__ unsigned cnt[HISTC_RANGE];
    const float lo = ord2f(mm[0]), hi = ord2f(mm[1]);
    const int b0 = blockIdx.x * HISTC_RANGE, nb = min(HISTC_RANGE, bins - b0);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    const bool vec = (((uintptr_t)x) & 15) == 0;
    const long n4 = vec ? n >> 2 : 0;
    auto one = [&](float v) {
        int pos = (hi > lo) ? (int)((v - lo) / (hi - lo) * (float)bins) : 0;
        pos = min(max(pos, 0), bins - 1) - b0;
        if ((unsigned)pos < (unsigned)nb) atomicAdd(&cnt[pos], 1u);
    };
    for (long i = threadIdx.x; i < n4; i += blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        one(v[0]); one(v[1]); one(v[2]); one(v[3]);
    }
    for (long i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) one(x[i]);
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += blockDim.x) hist[b0 + i] = (float)cnt[i];
}
static inline void launch_histc(const float* x, long n, const unsigned* mm, int bins, float* hist, int nb, hipStream_t s) {
    if (bins <= HISTC_LDS_BINS) hipLaunchKernelGGL(histc_kernel<HISTC_LDS_BINS>, dim3(nb), dim3(256), 0, s, x, n, mm, bins, hist);
    else if ((long)gx_cdiv(bins, HISTC_RANGE) * n <= (1L << 31))      // every range scans all values: while that stays cheap
        hipLaunchKernelGGL(histc_ranges_kernel, dim3(gx_cdiv(bins, HISTC_RANGE)), dim3(1024), 0, s, x, n, mm, bins, hist);
    else hipLaunchKernelGGL(histc_kernel<0>, dim3(nb), dim3(256), 0, s, x, n, mm, bins, hist);
}
// h += 1e-9 ; h[0] = h[1] ; h /= sum(h)      (swav_clustering.py:524-530).  Single block.
__global__ void hist_marginal_kernel(float* __restrict__ h, int bins) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < bins; i += blockDim.x) {
        const float v = ((i == 0 && bins > 1) ? h[1] : h[i]) + 1e-9f;
        s += v;
    }
    s = gx_block_sum(s, red);
    const float h1 = (bins > 1 ? h[1] : h[0]) + 1e-9f;
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += blockDim.x) {
        const float v = (i == 0) ? h1 : h[i] + 1e-9f;
        h[i] = v / s;
    }
}

// the same in two launches over many bins (the per-sample marginal of the 'image' pdf has one bin per sample: 320 000 at the bench
// batch, 0.15 ms in one block): block partials of sum(h + 1e-9) in a fixed order, then every block adds the partials up the same
// way and scales its own bins.  part: HIST_PARTS floats.
constexpr int HIST_PARTS = 256;
__device__ __forceinline__ float hist_term(const float* __restrict__ h, int i, int bins) {
    return ((i == 0 && bins > 1) ? h[1] : h[i]) + 1e-9f;
}
__global__ __launch_bounds__(256) void hist_sum_parts_kernel(const float* __restrict__ h, int bins, float* __restrict__ part) {
    __shared__ float red[16];
    const int per = (bins + HIST_PARTS - 1) / HIST_PARTS, i0 = blockIdx.x * per, i1 = min(bins, i0 + per);
    float s = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) s += hist_term(h, i, bins);
    s = gx_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void hist_scale_kernel(float* __restrict__ h, int bins, const float* __restrict__ part) {
    __shared__ float red[16];
    const float tot = gx_block_sum(part[threadIdx.x], red);              // HIST_PARTS == blockDim.x: the same sum in every block
    const int per = (bins + HIST_PARTS - 1) / HIST_PARTS, i0 = blockIdx.x * per, i1 = min(bins, i0 + per);
    const float h1 = (bins > 1 ? h[1] : h[0]) + 1e-9f;                   // (block 0 owns bins 0 and 1: read before its writes)
    __syncthreads();
    for (int i = i0 + threadIdx.x; i < i1; i += 256) h[i] = ((i == 0) ? h1 : h[i] + 1e-9f) / tot;
}

static inline int pick_team(int K) {
    int t = 8;
    while (t < 64 && t < K) t <<= 1;
    return t;
}

// cmax [groups][N], u [groups][N] from the per-32-row-slice tables of gx_gemm_f32_nt_planes (colmax_part, usum_part):
// group gi (blockIdx.y) = slices [gi * per_group, ...) — the last group may hold fewer (`total` slices in all).  A block =
// 64 columns x 4 slice lanes (a lane takes every fourth slice of the group), merged through LDS in a fixed order:
//     (m, s) + (m', s') = (max, s exp((m - max) / eps) + s' exp((m' - max) / eps))      — associative, so groups of
// many slices are merged in stages (gx_sinkhorn_first_from_parts).
__global__ __launch_bounds__(256) void sk_first_from_parts_kernel(const float* __restrict__ cmp, const float* __restrict__ usp,
                                                                  int per_group, int total, int N, float inv_eps,
                                                                  float* __restrict__ cmax, float* __restrict__ u) {
    __shared__ float sm[4][64], ss[4][64];
    const int cx = threadIdx.x & 63, py = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + cx, gi = blockIdx.y;
    const int p0 = gi * per_group, p1 = min(total, p0 + per_group);
    float m = -INFINITY, acc = 0.f;
    if (col < N) {
        for (int p = p0 + py; p < p1; p += 4) m = fmaxf(m, cmp[(long)p * N + col]);
        for (int p = p0 + py; p < p1; p += 4) {
            const float cp = cmp[(long)p * N + col];
            if (cp > -INFINITY) acc += usp[(long)p * N + col] * expf((cp - m) * inv_eps);
        }
    }
    sm[py][cx] = m; ss[py][cx] = acc;
    __syncthreads();
    if (py == 0 && col < N) {
        float mm = fmaxf(fmaxf(sm[0][cx], sm[1][cx]), fmaxf(sm[2][cx], sm[3][cx])), a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (sm[j][cx] > -INFINITY) a += ss[j][cx] * expf((sm[j][cx] - mm) * inv_eps);
        cmax[(long)gi * N + col] = mm;
        u[(long)gi * N + col] = a;
    }
}
void launch_partial_reduce_batch(const float* part, int nblk, int K, int nprob, float* out, hipStream_t stream) {
    int kx = 1;
    while (kx < 64 && kx < K) kx <<= 1;
    hipLaunchKernelGGL(partial_reduce_kernel, dim3((K + kx - 1) / kx, nprob), dim3(256), 0, stream, part, nblk, K, 1.f, 0.f, out, kx);
}
void launch_first_from_parts(const float* cmp, const float* usp, int per, int nprob, int K, float inv_eps, float* cmax, float* u,
                             hipStream_t stream) {
    hipLaunchKernelGGL(sk_first_from_parts_kernel, dim3(gx_cdiv(K, 64), nprob), dim3(256), 0, stream, cmp, usp, per, nprob * per, K,
                       inv_eps, cmax, u);
}
__global__ void sk_first_preset_kernel(const float* __restrict__ u_local, const float* __restrict__ cmax_local, int K, float inv_eps,
                                       const unsigned* __restrict__ cmax, float* __restrict__ u) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) u[k] = u_local[k] * expf((cmax_local[k] - ord2f(cmax[k])) * inv_eps);
}
}  // namespace gx

using namespace gx;

extern "C" {

int gx_rotflip_map(float angle_deg, int flip, int H, int W, int* map, void* stream) {
    if (!map || H <= 0 || W <= 0) return GX_ERR_BAD_ARG;
    const double rot = -(double)angle_deg * 3.14159265358979323846 / 180.0;
    const float cs = (float)cos(rot), sn = (float)sin(rot);
    hipLaunchKernelGGL(rotflip_map_kernel, dim3(gx_cdiv((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       map, H, W, cs, sn, flip);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// maps [n][H*W]: image i rotated by angles_deg[i] (host array) then flipped if flips[i] (host array)
int gx_rotflip_map_batch(const float* angles_deg, const int* flips, int n, int H, int W, int* maps, void* stream) {
    if (!angles_deg || !flips || !maps || n <= 0 || H <= 0 || W <= 0) return GX_ERR_BAD_ARG;
    for (int i0 = 0; i0 < n; i0 += RF_MAX) {
        const int m = n - i0 < RF_MAX ? n - i0 : RF_MAX;
        RotFlipBatch rf;
        for (int i = 0; i < m; ++i) {
            const double rot = -(double)angles_deg[i0 + i] * 3.14159265358979323846 / 180.0;
            rf.cs[i] = (float)cos(rot); rf.sn[i] = (float)sin(rot); rf.flip[i] = flips[i0 + i] ? 1 : 0;
        }
        hipLaunchKernelGGL(rotflip_map_batch_kernel, dim3(gx_cdiv((long)H * W, 256), m), dim3(256), 0, (hipStream_t)stream,
                           maps + (long)i0 * H * W, H, W, rf);
        GX_CHECK_LAUNCH();
    }
    return GX_OK;
}

// picks [rows][N] int32 = indices of the N smallest of the HW uniform fp32 keys (>= 0) of every row:
// a uniform random N-subset per row (ascending; pixels whose key ties the N-th smallest come last).  HW <= 2^22, 1 <= N <= HW.
int gx_random_subset_picks(const float* keys, int rows, int HW, int N, int* picks, void* stream) {
    if (!keys || !picks || rows <= 0 || HW <= 0 || HW > (1 << 22) || N <= 0 || N > HW) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(random_subset_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned*>(keys), HW, N, picks);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

static int gx_tap_list(TapList& t, int ntaps, const float* const* tap_ptrs, const int* tap_channels, const int* tap_res, int hlen) {
    if (ntaps <= 0 || ntaps > MAX_TAPS || !tap_ptrs || !tap_channels || !tap_res || hlen <= 0) return GX_ERR_BAD_ARG;
    t.n = ntaps;
    int off = 0;
    for (int l = 0; l < ntaps; ++l) {
        if (!tap_ptrs[l] || tap_channels[l] <= 0 || tap_res[l] <= 0) return GX_ERR_BAD_ARG;
        t.ptr[l] = tap_ptrs[l]; t.C[l] = tap_channels[l]; t.res[l] = tap_res[l]; t.off[l] = off;
        off += tap_channels[l];
    }
    return off < hlen ? GX_ERR_BAD_ARG : GX_OK;
}
int gx_stack_norm_map_bilinear(int ntaps, const float* const* tap_ptrs, const int* tap_channels, const int* tap_res, int hlen,
                               const int* maps, int B, int H, int W, float* img, void* stream) {
    if (!img || B <= 0 || H <= 0 || W <= 0) return GX_ERR_BAD_ARG;
    TapList t;
    const int rc = gx_tap_list(t, ntaps, tap_ptrs, tap_channels, tap_res, hlen);
    if (rc) return rc;
    hipLaunchKernelGGL(stack_norm_bilinear_kernel, dim3(gx_cdiv((long)B * H * W, 4)), dim3(256), 0, (hipStream_t)stream, t, maps, B,
                       H, W, hlen, img);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_gather_taps(int ntaps, const float* const* tap_ptrs, const int* tap_channels, const int* tap_res,
                   int B, int H, int W, int hlen, const int* picks, int rows_per_img, const int* maps, int interp,
                   float* X, long ldx, void* stream) {
    if (ntaps <= 0 || ntaps > MAX_TAPS || !tap_ptrs || !tap_channels || !tap_res || !picks || !X || B <= 0 ||
        rows_per_img <= 0 || hlen <= 0 || ldx < hlen || interp < 0 || interp > 1)
        return GX_ERR_BAD_ARG;
    TapList t;
    t.n = ntaps;
    int off = 0;
    for (int l = 0; l < ntaps; ++l) {
        if (!tap_ptrs[l] || tap_channels[l] <= 0 || tap_res[l] <= 0) return GX_ERR_BAD_ARG;
        t.ptr[l] = tap_ptrs[l]; t.C[l] = tap_channels[l]; t.res[l] = tap_res[l]; t.off[l] = off;
        off += tap_channels[l];
    }
    if (off < hlen) return GX_ERR_BAD_ARG;
    const int total = B * rows_per_img;
    hipLaunchKernelGGL(gather_taps_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, t, picks, maps,
                       rows_per_img, H, W, hlen, interp, X, ldx, total);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_row_sumsq(const float* x, long rows, int cused, long ld, float* out, void* stream) {
    if (!x || !out || rows <= 0 || cused <= 0 || ld < cused) return GX_ERR_BAD_ARG;
    if ((cused & 3) == 0 && (ld & 3) == 0 && (((uintptr_t)x) & 15) == 0) {
        const int q = cused >> 2;                               // 16-byte units per row
#define GX_RSS(T_) hipLaunchKernelGGL(row_sumsq_vec_kernel<T_>, dim3(gx_cdiv(rows * T_, 256)), dim3(256), 0, (hipStream_t)stream, x, rows, cused, ld, out)
        if (q <= 8) GX_RSS(8);
        else if (q <= 16) GX_RSS(16);
        else if (q <= 32) GX_RSS(32);
        else GX_RSS(64);
#undef GX_RSS
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    hipLaunchKernelGGL(row_sumsq_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, cused,
                       ld, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_norm_map(int ntaps, const float* const* ss_ptrs, const int* tap_res, const int* maps, int B, int H, int W,
                float* img, void* stream) {
    if (ntaps <= 0 || ntaps > MAX_TAPS || !ss_ptrs || !tap_res || !img || B <= 0) return GX_ERR_BAD_ARG;
    SsList s; s.n = ntaps;
    for (int l = 0; l < ntaps; ++l) { s.ptr[l] = ss_ptrs[l]; s.res[l] = tap_res[l]; }
    hipLaunchKernelGGL(norm_map_kernel, dim3(gx_cdiv((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, s,
                       maps, B, H, W, img);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_row_l2_normalize(float* z, long rows, int D, long ld, float* rnorm, void* stream) {
    if (!z || rows <= 0 || D <= 0 || ld < D) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(row_l2_normalize_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, z, rows,
                       D, ld, rnorm);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_l2_normalize_bwd(float* g, const float* zhat, const float* rnorm, long rows, int D, long ld, void* stream) {
    if (!g || !zhat || !rnorm || rows <= 0 || D <= 0 || ld < D) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(l2_normalize_bwd_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, g, zhat,
                       rnorm, rows, D, ld);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_row_argmax(const float* z, long rows, int D, long ld, long long* labels, void* stream) {
    if (!z || !labels || rows <= 0 || D <= 0 || ld < D) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(row_argmax_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, z, rows, D, ld,
                       labels);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// ---- column sums (db of the prototype bias): out = alpha*colsum(x) + beta*out ----
#define GX_RED_BLOCKS 256
#define GX_COLSUM_BLOCKS 1024
size_t gx_colsum_workspace_bytes(int K) { return (size_t)GX_COLSUM_BLOCKS * K * sizeof(float); }
int gx_colsum(const float* x, long rows, int K, long ld, float alpha, float beta, float* out, float* workspace,
              size_t workspace_bytes, void* stream) {
    if (!x || !out || rows <= 0 || K <= 0 || ld < K) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_colsum_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    if (bigk_ok(K) && bigk_aligned(x, ld)) {   // many columns: wave-per-row streaming kernel
        const int nbk = bigk_blocks(rows, GX_COLSUM_BLOCKS);
        const size_t lds = bigk_pass_lds(K);
        GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_COLSUM>), dim3(nbk), dim3(256), lds,
                                                        (hipStream_t)stream, x, rows, K, ld, 0.f, nullptr, nullptr,
                                                        nullptr, 0.f, nullptr, workspace, nullptr));
        GX_CHECK_LAUNCH();
        launch_partial_reduce(workspace, nbk, K, alpha, beta, out, (hipStream_t)stream);
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    const int nb = (int)((rows < GX_RED_BLOCKS) ? rows : GX_RED_BLOCKS);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, rows, K, ld, workspace);
    GX_CHECK_LAUNCH();
    launch_partial_reduce(workspace, nb, K, alpha, beta, out, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// ---- Sinkhorn ----------------------------------------------------------------------------
#define GX_SK_BLOCKS 512
// workspace layout (4-byte words): [0, K) cmax (ordered-uint), [K, 2K) u, [2K, 2K + BLOCKS*K) partials
#define GX_SK_SMALL_BLOCKS 1024   // single-rank small-K path: two partial buffers of this many blocks
size_t gx_sinkhorn_workspace_bytes(int K) {
    return (size_t)(4 * (size_t)K + (size_t)2 * GX_SK_SMALL_BLOCKS * K) * sizeof(float);
}
float* gx_sinkhorn_u(float* workspace, int K) { return workspace + K; }

static int sk_blocks(long rows, int team) {
    const long tpb = 256 / team;
    long nb = (rows + tpb - 1) / tpb;
    return (int)(nb < GX_SK_BLOCKS ? nb : GX_SK_BLOCKS);
}

// cmax_k = max_n S_nk (kept in the workspace, ordered-uint encoded; reset here)
static int sk_colmax_impl(const float* S, const int* rowmap, long rows, int K, long ld, float* workspace,
                          size_t workspace_bytes, void* stream) {
    if (!S || !workspace || rows <= 0 || K <= 0 || ld < K) return GX_ERR_BAD_ARG;
    if (workspace_bytes < gx_sinkhorn_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    if (rowmap && !(bigk_ok(K) && bigk_aligned(S, ld))) return GX_ERR_BAD_ARG;       // row maps: the many-prototype kernels only
    if (hipMemsetAsync(workspace, 0, (size_t)K * 4, (hipStream_t)stream) != hipSuccess) return GX_ERR_LAUNCH;
    if (bigk_ok(K) && bigk_aligned(S, ld)) {
        const int nbk = bigk_blocks(rows, 2040);
        const size_t lds = bigk_pass_lds(K);
        GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_COLMAX>), dim3(nbk), dim3(256), lds,
                                                        (hipStream_t)stream, S, rows, K, ld, 0.f, nullptr, nullptr,
                                                        nullptr, 0.f, nullptr, nullptr, (unsigned*)workspace, 0L, 0L, rowmap));
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    const int nb = (int)(rows < 256 ? rows : 256);
    hipLaunchKernelGGL(sk_colmax_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, S, rows, K, ld,
                       (unsigned*)workspace);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_sinkhorn_colmax(const float* S, long rows, int K, long ld, float* workspace, size_t workspace_bytes,
                       void* stream) {
    return sk_colmax_impl(S, nullptr, rows, K, ld, workspace, workspace_bytes, stream);
}
// ... over the rows rowmap[0 .. rows) of a table S (int32 row indices; many prototypes only): the Sinkhorn problem of one
// patch and view on the per-pixel score table of the view (DESIGN par. 4.15)
int gx_sinkhorn_colmax_rows(const float* S, const int* rowmap, long rows, int K, long ld, float* workspace,
                            size_t workspace_bytes, void* stream) {
    if (!rowmap) return GX_ERR_BAD_ARG;
    return sk_colmax_impl(S, rowmap, rows, K, ld, workspace, workspace_bytes, stream);
}
// Column-max exchange hooks for the sharded (multi-GPU) form: decode to / encode from floats.
int gx_sinkhorn_cmax_get(const float* workspace, int K, float* out, void* stream) {
    if (!workspace || !out || K <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(sk_ord_decode_kernel, dim3(gx_cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned*)workspace, K, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_sinkhorn_cmax_set(float* workspace, int K, const float* in, void* stream) {
    if (!workspace || !in || K <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(sk_ord_encode_kernel, dim3(gx_cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, in, K,
                       (unsigned*)workspace);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// One pass: first != 0 -> u_k = sum_n exp((S_nk - cmax_k)/eps); else the sweep with the log
// scaling vector la (lb_out optional).  Result: local column sums u (K floats) at
// gx_sinkhorn_u(workspace, K).
static int sk_pass_impl(const float* S, const int* rowmap, long rows, int K, long ld, float eps, int first, const float* la,
                        const float* c, float c_uniform, float* lb_out, float* workspace, size_t workspace_bytes,
                        void* stream) {
    if (!S || rows <= 0 || K <= 0 || ld < K || eps <= 0.f || (!first && !la)) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_sinkhorn_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    if (rowmap && !(bigk_ok(K) && bigk_aligned(S, ld))) return GX_ERR_BAD_ARG;
    const int team = pick_team(K);
    const int nb = sk_blocks(rows, team);
    float* u = workspace + K;
    float* part = u + K;
    const size_t lds = (size_t)2 * K * sizeof(float);
    if (lds > 160 * 1024) return GX_ERR_BAD_ARG;
    if (bigk_ok(K) && bigk_aligned(S, ld)) {
        const int nbk = bigk_blocks(rows, 2040);           // partial area holds 2048*K floats
        const size_t lds = bigk_pass_lds(K);
        if (first) {
            GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_FIRST>), dim3(nbk), dim3(256), lds,
                                                            (hipStream_t)stream, S, rows, K, ld, 1.f / eps,
                                                            (const unsigned*)workspace, la, c, c_uniform, lb_out, part,
                                                            nullptr, 0L, 0L, rowmap));
        } else {
            GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_SWEEP>), dim3(nbk), dim3(256), lds,
                                                            (hipStream_t)stream, S, rows, K, ld, 1.f / eps,
                                                            (const unsigned*)workspace, la, c, c_uniform, lb_out, part,
                                                            nullptr, 0L, 0L, rowmap));
        }
        GX_CHECK_LAUNCH();
        launch_partial_reduce(part, nbk, K, 1.f, 0.f, u, (hipStream_t)stream, part + (size_t)2040 * K);   // rows 2040..2047 are spare
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    if (first)
        hipLaunchKernelGGL((sk_pass_kernel<true>), dim3(nb), dim3(256), lds, (hipStream_t)stream, S, rows, K, ld, team,
                           1.f / eps, (const unsigned*)workspace, la, c, c_uniform, lb_out, part);
    else
        hipLaunchKernelGGL((sk_pass_kernel<false>), dim3(nb), dim3(256), lds, (hipStream_t)stream, S, rows, K, ld,
                           team, 1.f / eps, (const unsigned*)workspace, la, c, c_uniform, lb_out, part);
    GX_CHECK_LAUNCH();
    launch_partial_reduce(part, nb, K, 1.f, 0.f, u, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_sinkhorn_pass(const float* S, long rows, int K, long ld, float eps, int first, const float* la,
                     const float* c, float c_uniform, float* lb_out, float* workspace, size_t workspace_bytes,
                     void* stream) {
    return sk_pass_impl(S, nullptr, rows, K, ld, eps, first, la, c, c_uniform, lb_out, workspace, workspace_bytes, stream);
}
int gx_sinkhorn_pass_rows(const float* S, const int* rowmap, long rows, int K, long ld, float eps, int first, const float* la,
                          const float* c, float c_uniform, float* lb_out, float* workspace, size_t workspace_bytes,
                          void* stream) {
    if (!rowmap) return GX_ERR_BAD_ARG;
    return sk_pass_impl(S, rowmap, rows, K, ld, eps, first, la, c, c_uniform, lb_out, workspace, workspace_bytes, stream);
}
// (a group with many slices is merged in stages — chunks of 64 slices, then the chunks — so that the 10 000 slices of a
// 320 000-row score matrix are not one thread's serial loop; workspace: gx_sinkhorn_first_from_parts_workspace_bytes)
size_t gx_sinkhorn_first_from_parts_workspace_bytes(int M, int N) {
    const long n1 = gx_cdiv(4L * gx_cdiv(M, 128), 64);
    return (size_t)2 * (n1 + gx_cdiv(n1, 64)) * N * sizeof(float);
}
int gx_sinkhorn_first_from_parts(const float* colmax_part, const float* usum_part, int M, int N, int groups, float eps,
                                 float* cmax, float* u, float* workspace, size_t workspace_bytes, void* stream) {
    if (!colmax_part || !usum_part || !cmax || !u || M <= 0 || N <= 0 || groups < 1 || !(eps > 0.f)) return GX_ERR_BAD_ARG;
    if (groups > 1 && (M % (32 * groups))) return GX_ERR_BAD_ARG;          // whole 32-row slices per group
    const int per_group = groups > 1 ? M / (32 * groups) : 4 * gx_cdiv(M, 128);
    hipStream_t st = (hipStream_t)stream;
    const dim3 blk(256);
    if (groups > 1 || per_group <= 256) {
        hipLaunchKernelGGL(sk_first_from_parts_kernel, dim3(gx_cdiv(N, 64), groups), blk, 0, st, colmax_part, usum_part, per_group,
                           per_group * groups, N, 1.f / eps, cmax, u);
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    if (!workspace || workspace_bytes < gx_sinkhorn_first_from_parts_workspace_bytes(M, N)) return GX_ERR_WORKSPACE;
    const float* cin = colmax_part;
    const float* uin = usum_part;
    int parts = per_group;
    float* w = workspace;
    while (parts > 256) {                                                // stage: chunks of 64 -> one pair of rows per chunk
        const int nch = gx_cdiv(parts, 64);
        float* cm1 = w;
        float* u1 = w + (size_t)nch * N;
        hipLaunchKernelGGL(sk_first_from_parts_kernel, dim3(gx_cdiv(N, 64), nch), blk, 0, st, cin, uin, 64, parts, N, 1.f / eps, cm1, u1);
        GX_CHECK_LAUNCH();
        cin = cm1; uin = u1; parts = nch;
        w += (size_t)2 * nch * N;
    }
    hipLaunchKernelGGL(sk_first_from_parts_kernel, dim3(gx_cdiv(N, 64), 1), blk, 0, st, cin, uin, parts, parts, N, 1.f / eps, cmax, u);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_sinkhorn_first_preset(const float* u_local, const float* cmax_local, int K, float eps, float* workspace, void* stream) {
    if (!u_local || !cmax_local || !workspace || K <= 0 || !(eps > 0.f)) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(sk_first_preset_kernel, dim3(gx_cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, u_local, cmax_local,
                       K, 1.f / eps, (const unsigned*)workspace, workspace + K);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// la update from the (all-reduced) column sums u; `first` uses -cmax/eps from the workspace as base.
int gx_sinkhorn_update(const float* u, const float* r, int K, int first, const float* workspace, float eps,
                       float* la, void* stream) {
    if (!u || !la || K <= 0 || (first && !workspace)) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(sk_update_la_kernel, dim3(gx_cdiv(K, 256)), dim3(256), 0, (hipStream_t)stream, u, r, K, first,
                       (const unsigned*)workspace, 1.f / eps, la);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// Whole single-device Sinkhorn: leaves the log scaling vector la[K].
int gx_sinkhorn(const float* S, long rows, int K, long ld, float eps, int niters, const float* r, const float* c,
                float* la, float* workspace, size_t workspace_bytes, void* stream) {
    if (niters < 1 || !la) return GX_ERR_BAD_ARG;
    int rc = gx_sinkhorn_colmax(S, rows, K, ld, workspace, workspace_bytes, stream);
    if (rc) return rc;
    for (int it = 0; it < niters; ++it) {
        rc = gx_sinkhorn_pass(S, rows, K, ld, eps, it == 0, la, c, 1.f / (float)rows, nullptr, workspace,
                              workspace_bytes, stream);
        if (rc) return rc;
        rc = gx_sinkhorn_update(workspace + K, r, K, it == 0, workspace, eps, la, stream);
        if (rc) return rc;
    }
    return GX_OK;
}
// The column maxima AND the first Sinkhorn pass of the problem "rows rowmap[0 .. rows) of table S" (rowmap NULL: rows 0 .. rows)
// in ONE sweep: cmax[k] = max_n S_nk, u[k] = sum_n exp((S_nk - cmax[k]) / eps) (what gx_sinkhorn_colmax + gx_sinkhorn_pass(first)
// leave in two) — every block keeps a running (max, sum) pair per column (online log-sum-exp, one exponential per score)
// and the pairs are merged exactly.  Many prototypes only.  workspace >= gx_sinkhorn_first_rows_workspace_bytes(K).
size_t gx_sinkhorn_first_rows_workspace_bytes(int K) { return (size_t)(2 * 2040 + 2 * 32 + 8) * K * sizeof(float); }
int gx_sinkhorn_first_rows(const float* S, const int* rowmap, long rows, int K, long ld, float eps, float* cmax, float* u,
                           float* workspace, size_t workspace_bytes, void* stream) {
    if (!S || !cmax || !u || rows <= 0 || K <= 0 || ld < K || !(eps > 0.f)) return GX_ERR_BAD_ARG;
    if (!(bigk_ok(K) && bigk_aligned(S, ld))) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_sinkhorn_first_rows_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nbk = bigk_blocks(rows, 2040);
    float* ps = workspace;                        // [nbk][K] sums
    float* pm = ps + (size_t)2040 * K;            // [nbk][K] maxima
    float* w = pm + (size_t)2040 * K;             // merge stage: 2 x 32 x K
    const size_t lds = bigk_pass_lds(K);
    GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_FIRSTLSE>), dim3(nbk), dim3(256), lds, st, S, rows, K,
                                                    ld, 1.f / eps, nullptr, nullptr, nullptr, 0.f, pm, ps, nullptr, 0L, 0L,
                                                    rowmap));
    GX_CHECK_LAUNCH();
    const float* cin = pm;
    const float* uin = ps;
    int parts = nbk;
    if (parts > 256) {                            // chunks of 64 blocks first (at most 32 of them)
        const int nch = gx_cdiv(parts, 64);
        hipLaunchKernelGGL(sk_first_from_parts_kernel, dim3(gx_cdiv(K, 64), nch), dim3(256), 0, st, cin, uin, 64, parts, K, 1.f / eps,
                           w, w + (size_t)nch * K);
        GX_CHECK_LAUNCH();
        cin = w; uin = w + (size_t)nch * K; parts = nch;
    }
    hipLaunchKernelGGL(sk_first_from_parts_kernel, dim3(gx_cdiv(K, 64), 1), dim3(256), 0, st, cin, uin, parts, parts, K, 1.f / eps, cmax, u);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// gx_sinkhorn over the rows rowmap[0 .. rows) of a table (many prototypes only); c_uniform = the uniform row marginal
// (1 / rows on one rank)
int gx_sinkhorn_rows(const float* S, const int* rowmap, long rows, int K, long ld, float eps, int niters, const float* r,
                     const float* c, float c_uniform, float* la, float* workspace, size_t workspace_bytes, void* stream) {
    if (niters < 1 || !la || !rowmap) return GX_ERR_BAD_ARG;
    int rc = sk_colmax_impl(S, rowmap, rows, K, ld, workspace, workspace_bytes, stream);
    if (rc) return rc;
    for (int it = 0; it < niters; ++it) {
        rc = sk_pass_impl(S, rowmap, rows, K, ld, eps, it == 0, la, c, c_uniform, nullptr, workspace, workspace_bytes, stream);
        if (rc) return rc;
        rc = gx_sinkhorn_update(workspace + K, r, K, it == 0, workspace, eps, la, stream);
        if (rc) return rc;
    }
    return GX_OK;
}
// Many equally shaped many-prototype problems at once (the per-image Sinkhorn scope at K = 5000: 2 views x 5 patches x 16
// images = 160 problems of 20 000 x 5000 per step — one by one they are 1 760 passes + as many reductions of a few
// tens of microseconds each, i.e. launch-bound): every pass of the iteration runs over all problems in one launch
// (blockIdx.y = problem), like gx_sinkhorn_small_batch for K <= 64.
//   S_all: problem i at S_all + i * prob_stride, [rows][K] with row stride ld; r_all / c_all optional (NULL = uniform);
//   la_all [nprob][K]; cmax_all (optional) [nprob][K]: the problems' column maxima when their producer has them (no
//   column-max pass then).  workspace: cmax [nprob][K] | u [nprob][K] | partials [nprob][NB][K].
static int bigk_batch_blocks(int nprob, long rows) {
    long nb = (4096 + nprob - 1) / nprob;
    if (nb < 8) nb = 8;
    if (nb > 512) nb = 512;
    const long groups = (rows + BK_R - 1) / BK_R;
    return (int)(nb < groups ? nb : groups);
}
size_t gx_sinkhorn_bigk_batch_workspace_bytes(int nprob, long rows, int K) {
    return ((size_t)2 * nprob * K + (size_t)nprob * bigk_batch_blocks(nprob, rows) * K) * sizeof(float);
}
static int bigk_batch_impl(const float* S_all, const int* rowmap_all, int nprob, long prob_stride, long rows, int K, long ld,
                           float eps, int niters, const float* r_all, long r_stride, const float* c_all, long c_stride,
                           float c_uniform, const float* cmax_all, const float* u_first_all, float* la_all, float* workspace,
                           size_t workspace_bytes, void* stream);
int gx_sinkhorn_bigk_batch(const float* S_all, int nprob, long prob_stride, long rows, int K, long ld, float eps, int niters,
                           const float* r_all, long r_stride, const float* c_all, long c_stride, const float* cmax_all,
                           const float* u_first_all, float* la_all, float* workspace, size_t workspace_bytes, void* stream) {
    if (rows <= 0) return GX_ERR_BAD_ARG;
    return bigk_batch_impl(S_all, nullptr, nprob, prob_stride, rows, K, ld, eps, niters, r_all, r_stride, c_all, c_stride,
                           1.f / (float)rows, cmax_all, u_first_all, la_all, workspace, workspace_bytes, stream);
}
// ... and several problems that are ROW SELECTIONS of one table (the patches of a view on its per-pixel score table, DESIGN
// par. 4.15): rowmap_all [nprob][rows] int32, each SORTED ascending (a problem does not care about the order of its rows), so
// that the problems' blocks — all resident together, dispatched interleaved — walk the table side by side and a row that
// several patches drew comes from HBM once.  r_all [nprob][K] / c_all [nprob][rows] (c in the order of the SORTED rows) or NULL:
// uniform marginals, c_uniform = the row marginal (1 / rows on one rank).
static int rows_batch_blocks(int nprob, long rows) {
    // a view's few patches: every problem's blocks resident at once (3 workgroups per CU), walking the table side by side; the many
    // small problems of the per-image scope: four rounds of blocks (nine blocks per problem left the sweeps 14 % short)
    long nb = (nprob <= 8 ? 3 * 256 : 12 * 256) / nprob;
    if (nb < 8) nb = 8;
    const long groups = (rows + BK_R - 1) / BK_R;
    return (int)(nb < groups ? nb : groups);
}
size_t gx_sinkhorn_rows_batch_workspace_bytes(int nprob, long rows, int K) {
    return ((size_t)2 * nprob * K + (size_t)nprob * rows_batch_blocks(nprob, rows) * K) * sizeof(float);
}
int gx_sinkhorn_rows_batch(const float* S, const int* rowmap_all, int nprob, long rows, int K, long ld, float eps, int niters,
                           const float* r_all, const float* c_all, float c_uniform, const float* cmax_all,
                           const float* u_first_all, float* la_all, float* workspace, size_t workspace_bytes, void* stream) {
    if (!rowmap_all || !(c_uniform > 0.f) || nprob > 96) return GX_ERR_BAD_ARG;
    return bigk_batch_impl(S, rowmap_all, nprob, 0, rows, K, ld, eps, niters, r_all, K, c_all, rows, c_uniform, cmax_all,
                           u_first_all, la_all, workspace, workspace_bytes, stream);
}
static int bigk_batch_impl(const float* S_all, const int* rowmap_all, int nprob, long prob_stride, long rows, int K, long ld,
                           float eps, int niters, const float* r_all, long r_stride, const float* c_all, long c_stride,
                           float c_uniform, const float* cmax_all, const float* u_first_all, float* la_all, float* workspace,
                           size_t workspace_bytes, void* stream) {
    if (!S_all || !la_all || nprob <= 0 || nprob > 65535 || rows <= 0 || K <= 0 || ld < K || eps <= 0.f || niters < 1)
        return GX_ERR_BAD_ARG;
    if (u_first_all && !cmax_all) return GX_ERR_BAD_ARG;
    if (!bigk_ok(K) || !bigk_aligned(S_all, ld) || (prob_stride & 3)) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < (rowmap_all ? gx_sinkhorn_rows_batch_workspace_bytes(nprob, rows, K)
                                                    : gx_sinkhorn_bigk_batch_workspace_bytes(nprob, rows, K)))
        return GX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int nb = rowmap_all ? rows_batch_blocks(nprob, rows) : bigk_batch_blocks(nprob, rows);
    unsigned* cmax = (unsigned*)workspace;
    float* u = workspace + (size_t)nprob * K;
    float* part = u + (size_t)nprob * K;
    const size_t lds = bigk_pass_lds(K);
    const dim3 grid(nb, nprob);
    if (cmax_all) {       // the producer of the scores already has the column maxima [nprob][K] (gx_gemm_f32_nt_planes)
        if ((long)nprob * K >= (1L << 31)) return GX_ERR_BAD_ARG;
        hipLaunchKernelGGL(sk_ord_encode_kernel, dim3(gx_cdiv((long)nprob * K, 256)), dim3(256), 0, st, cmax_all, nprob * K, cmax);
    } else {
        if (hipMemsetAsync(cmax, 0, (size_t)nprob * K * 4, st) != hipSuccess) return GX_ERR_LAUNCH;
        GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_COLMAX>), grid, dim3(256), lds, st, S_all, rows, K,
                                                        ld, 0.f, nullptr, nullptr, nullptr, 0.f, nullptr, nullptr, cmax,
                                                        prob_stride, 0L, rowmap_all));
    }
    GX_CHECK_LAUNCH();
    int kx = 1;
    while (kx < 64 && kx < K) kx <<= 1;
    for (int it = 0; it < niters; ++it) {
        if (it == 0 && u_first_all) {      // the score GEMM left the first pass behind (gx_sinkhorn_first_from_parts)
            hipLaunchKernelGGL(sk_update_la_kernel, dim3(gx_cdiv(K, 256), nprob), dim3(256), 0, st, u_first_all, r_all, K, 1, cmax,
                               1.f / eps, la_all, r_stride);
            GX_CHECK_LAUNCH();
            continue;
        }
        if (it == 0) {
            GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_FIRST>), grid, dim3(256), lds, st, S_all, rows, K,
                                                            ld, 1.f / eps, cmax, la_all, c_all, c_uniform, nullptr, part,
                                                            nullptr, prob_stride, c_stride, rowmap_all));
        } else {
            GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((bigk_pass_kernel<NJ, BK_SWEEP>), grid, dim3(256), lds, st, S_all, rows, K,
                                                            ld, 1.f / eps, cmax, la_all, c_all, c_uniform, nullptr, part,
                                                            nullptr, prob_stride, c_stride, rowmap_all));
        }
        GX_CHECK_LAUNCH();
        hipLaunchKernelGGL(partial_reduce_kernel, dim3((K + kx - 1) / kx, nprob), dim3(256), 0, st, part, nb, K, 1.f, 0.f, u, kx);
        GX_CHECK_LAUNCH();
        hipLaunchKernelGGL(sk_update_la_kernel, dim3(gx_cdiv(K, 256), nprob), dim3(256), 0, st, u, r_all, K, it == 0 ? 1 : 0, cmax,
                           1.f / eps, la_all, r_stride);
        GX_CHECK_LAUNCH();
    }
    return GX_OK;
}
int gx_sinkhorn_codes(const float* S, long rows, int K, long ld, float eps, const float* la, float* q, long ldq,
                      void* stream) {
    if (!S || !la || !q || rows <= 0 || K <= 0 || ld < K || ldq < K || eps <= 0.f) return GX_ERR_BAD_ARG;
    const int team = pick_team(K);
    hipLaunchKernelGGL(sk_codes_kernel, dim3(sk_blocks(rows, team)), dim3(256), 0, (hipStream_t)stream, S, rows, K,
                       ld, team, 1.f / eps, la, q, ldq);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// ---- swapped-prediction CE forward + d(scores) ---------------------------------------------
// loss_out[0] = loss_scale * sum_n loss_n + loss_beta * loss_out[0]
// 2048 loss partials, then (many-prototype path) 8 row statistics per row
size_t gx_swapped_ce_workspace_bytes(long rows, int K) {
    return ((size_t)2048 + (bigk_ok(K) ? (size_t)rows * 8 : 0)) * sizeof(float);
}
int gx_swapped_ce(const float* Ss, const float* St, long rows, int K, long ld, float eps, float temperature,
                  const float* la_s, const float* la_t, float grad_scale, float* dSs, float* dSt, long ldd,
                  float loss_scale, float loss_beta, float* loss_out, float* workspace, size_t workspace_bytes,
                  void* stream) {
    if (!Ss || !St || !la_s || !la_t || !loss_out || rows <= 0 || K <= 0 || ld < K) return GX_ERR_BAD_ARG;
    if ((dSs == nullptr) != (dSt == nullptr) || (dSs && ldd < K)) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_swapped_ce_workspace_bytes(rows, K)) return GX_ERR_WORKSPACE;
    if (bigk_ok(K) && bigk_aligned(Ss, ld) && bigk_aligned(St, ld) && (!dSs || (bigk_aligned(dSs, ldd) && bigk_aligned(dSt, ldd)))) {
        const int nbk = bigk_blocks(rows, 2040);
        const size_t lds = (size_t)2 * K * sizeof(float);
        float* stats = workspace + 2048;
        GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((ce_rowstats_bigk_kernel<NJ>), dim3(nbk), dim3(256), lds,
                                                        (hipStream_t)stream, Ss, St, rows, K, ld, 1.f / eps,
                                                        1.f / temperature, la_s, la_t, stats));
        GX_CHECK_LAUNCH();
        hipLaunchKernelGGL(ce_apply_bigk_kernel, dim3(nbk), dim3(256), lds, (hipStream_t)stream, Ss, St, rows, K, ld,
                           1.f / eps, 1.f / temperature, la_s, la_t, stats, grad_scale, dSs, dSt, ldd, workspace);
        GX_CHECK_LAUNCH();
        launch_partial_reduce(workspace, nbk, 1, loss_scale, loss_beta, loss_out, (hipStream_t)stream);
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    const int team = pick_team(K);
    const int nb = sk_blocks(rows, team);
    hipLaunchKernelGGL(swapped_ce_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, Ss, St, rows, K, ld, team,
                       1.f / eps, 1.f / temperature, la_s, la_t, grad_scale, dSs, dSt, ldd, workspace,
                       CePairs{1, 0, 0, 0, 0});
    GX_CHECK_LAUNCH();
    launch_partial_reduce(workspace, nb, 1, loss_scale, loss_beta, loss_out, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// The same loss and d(scores) as gx_swapped_ce for many prototypes (64 < K <= 6144, 16-B aligned rows) in ONE pass
// over the two score matrices (ce_fused_bigk_kernel), plus gb = gb_beta * gb + column sums of (dSs + dSt) — the
// prototype-bias gradient, otherwise a gx_colsum pass over each dS.  GX_ERR_BAD_ARG for shapes the fused kernel
// does not take (the caller then uses gx_swapped_ce + gx_colsum).
size_t gx_swapped_ce_fused_workspace_bytes(int K) { return ((size_t)2048 + (size_t)2 * 2048 * K) * sizeof(float); }
int gx_swapped_ce_fused(const float* Ss, const float* St, long rows, int K, long ld, float eps, float temperature,
                        const float* la_s, const float* la_t, float grad_scale, float* dSs, float* dSt, long ldd,
                        float loss_scale, float loss_beta, float* loss_out, float* gb, float gb_beta, float* workspace,
                        size_t workspace_bytes, void* stream) {
    if (!Ss || !St || !la_s || !la_t || !loss_out || !dSs || !dSt || !gb || rows <= 0 || K <= 0 || ld < K || ldd < K)
        return GX_ERR_BAD_ARG;
    if (!bigk_ce_ok(K) || !bigk_aligned(Ss, ld) || !bigk_aligned(St, ld) || !bigk_aligned(dSs, ldd) || !bigk_aligned(dSt, ldd))
        return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_swapped_ce_fused_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    long nb = (rows + CE_R - 1) / CE_R;
    const int nbk = (int)(nb < 2040 ? nb : 2040);
    const size_t lds = ((size_t)2 * K + 2 * CE_R * 16) * sizeof(float);
    float* part_s = workspace + 2048;
    float* part_t = part_s + (size_t)2048 * K;
    GX_BIGK_DISPATCH(bigk_nj(K), hipLaunchKernelGGL((ce_fused_bigk_kernel<NJ>), dim3(nbk), dim3(256), lds, (hipStream_t)stream,
                                                    Ss, St, rows, K, ld, 1.f / eps, 1.f / temperature, la_s, la_t, grad_scale,
                                                    dSs, dSt, ldd, workspace, part_s, part_t));
    GX_CHECK_LAUNCH();
    launch_partial_reduce(workspace, nbk, 1, loss_scale, loss_beta, loss_out, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    launch_partial_reduce(part_s, nbk, K, 1.f, gb_beta, gb, (hipStream_t)stream, part_s + (size_t)2040 * K);
    GX_CHECK_LAUNCH();
    launch_partial_reduce(part_t, nbk, K, 1.f, 1.f, gb, (hipStream_t)stream, part_t + (size_t)2040 * K);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// The swapped CE of every patch of a step with the d(scores) of a pixel added up over the patches that drew it
// (ce_pixels_bigk_kernel): S_all [P][2 views][rows][ld] (SinkhornBatch order), inv [P][npix] int32 = the pixel's row in patch
// p's matrices or -1, la_all [(p * 2 + view) * ngroups + group][K] with group = pixel / (npix / ngroups); dSs / dSt
// [npix][ldd] are WRITTEN (not accumulated): sum over patches of what gx_swapped_ce_fused writes per row, columns K .. kpad - 1
// zero.  loss_out / gb as there.  K as for gx_swapped_ce_fused; kpad in [K, ldd], a multiple of 4.
static int ce_pixels_impl(const float* S_all, const float* St_tab, int P, long rows, int K, long ld, const int* inv, long npix,
                          int ngroups, float eps, float temperature, const float* la_all, float grad_scale, float* dSs,
                          float* dSt, long ldd, int kpad, float loss_scale, float loss_beta, float* loss_out, float* gb,
                          float gb_beta, float* workspace, size_t workspace_bytes, void* stream, const int* live = nullptr,
                          const int* row_group = nullptr);
int gx_swapped_ce_pixels(const float* S_all, int P, long rows, int K, long ld, const int* inv, long npix, int ngroups,
                         float eps, float temperature, const float* la_all, float grad_scale, float* dSs, float* dSt, long ldd,
                         int kpad, float loss_scale, float loss_beta, float* loss_out, float* gb, float gb_beta,
                         float* workspace, size_t workspace_bytes, void* stream) {
    return ce_pixels_impl(S_all, nullptr, P, rows, K, ld, inv, npix, ngroups, eps, temperature, la_all, grad_scale, dSs, dSt, ldd,
                          kpad, loss_scale, loss_beta, loss_out, gb, gb_beta, workspace, workspace_bytes, stream);
}
// ... with the scores as per-pixel tables as well (Ss_tab / St_tab [npix][ld], one per view): inv only says which patches
// drew a pixel (>= 0), la_all as above.  live (device scalar, may be NULL): the tables hold the drawn pixels in their first
// *live rows; the d(scores) rows past them are left untouched.  row_group (int32 [npix], may be NULL; ngroups > 1): the Sinkhorn
// problem (image) of every table row, where the rows of an image are no longer npix / ngroups apart
int gx_swapped_ce_pixels_tab(const float* Ss_tab, const float* St_tab, int P, int K, long ld, const int* inv, long npix,
                             int ngroups, float eps, float temperature, const float* la_all, float grad_scale, float* dSs,
                             float* dSt, long ldd, int kpad, float loss_scale, float loss_beta, float* loss_out, float* gb,
                             float gb_beta, const int* live, const int* row_group, float* workspace, size_t workspace_bytes,
                             void* stream) {
    if (!St_tab || !bigk_aligned(St_tab, ld)) return GX_ERR_BAD_ARG;
    return ce_pixels_impl(Ss_tab, St_tab, P, 1, K, ld, inv, npix, ngroups, eps, temperature, la_all, grad_scale, dSs, dSt, ldd,
                          kpad, loss_scale, loss_beta, loss_out, gb, gb_beta, workspace, workspace_bytes, stream, live, row_group);
}
static int ce_pixels_impl(const float* S_all, const float* St_tab, int P, long rows, int K, long ld, const int* inv, long npix,
                          int ngroups, float eps, float temperature, const float* la_all, float grad_scale, float* dSs,
                          float* dSt, long ldd, int kpad, float loss_scale, float loss_beta, float* loss_out, float* gb,
                          float gb_beta, float* workspace, size_t workspace_bytes, void* stream, const int* live,
                          const int* row_group) {
    if (!S_all || !inv || !la_all || !loss_out || !dSs || !dSt || !gb || P <= 0 || P > CE_PMAX || rows <= 0 || K <= 0 || ld < K || npix <= 0 ||
        ngroups <= 0 || (!row_group && npix % ngroups) || kpad < K || kpad > ldd || (kpad & 3))
        return GX_ERR_BAD_ARG;
    const int cebt = bigk_nj(K) > CE_PIX_NJ2 ? CE_PIX_BT_WIDE : 256;              // ce_pixels_bigk_kernel's block
    if (!bigk_ok(K) || !bigk_aligned(S_all, ld) || !bigk_aligned(dSs, ldd) || !bigk_aligned(dSt, ldd) ||
        (((uintptr_t)la_all) & 15) || 4 * cebt * ((bigk_nj(K) * 64 + 4 * cebt - 1) / (4 * cebt)) < kpad)
        return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_swapped_ce_fused_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    const int nbk = (int)(npix < 2040 ? npix : 2040);
    float* part_s = workspace + 2048;
    float* part_t = part_s + (size_t)2048 * K;
    GX_BIGK_DISPATCH_PASS(bigk_nj(K), hipLaunchKernelGGL((ce_pixels_bigk_kernel<NJ>), dim3(nbk), dim3(cebt), (size_t)2 * K * sizeof(float), (hipStream_t)stream,
                                                    S_all, P, rows, K, ld, inv, npix, ngroups, npix / ngroups, la_all, 1.f / eps,
                                                    1.f / temperature, grad_scale, dSs, dSt, ldd, kpad, workspace, part_s,
                                                    part_t, St_tab, live, row_group));
    GX_CHECK_LAUNCH();
    launch_partial_reduce(workspace, nbk, 1, loss_scale, loss_beta, loss_out, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    launch_partial_reduce(part_s, nbk, K, 1.f, gb_beta, gb, (hipStream_t)stream, part_s + (size_t)2040 * K);
    GX_CHECK_LAUNCH();
    launch_partial_reduce(part_t, nbk, K, 1.f, 1.f, gb, (hipStream_t)stream, part_t + (size_t)2040 * K);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// Every (patch, row group) pair of a step in one launch (K <= 64, contiguous rows: ld == K).  Scores S_all
// [npatch][2 views][rows_per_patch][K] with row groups of `rows` rows (ngroups per patch: 1 for the global
// Sinkhorn scope, B for the per-image one); la_all [(patch*2 + view)*ngroups + group][K] (SinkhornBatch order);
// dS_all like S_all.  loss_out[0] = loss_scale * sum + loss_beta * loss_out[0].
// workspace >= npatch * ngroups * 512 floats.
int gx_swapped_ce_batch(const float* S_all, int npatch, int ngroups, long rows_per_patch, long rows, int K, float eps,
                        float temperature, const float* la_all, float grad_scale, float* dS_all, float loss_scale,
                        float loss_beta, float* loss_out, float* workspace, size_t workspace_bytes, void* stream) {
    if (!S_all || !la_all || !loss_out || npatch <= 0 || ngroups <= 0 || rows <= 0 || K <= 0 || K > 64 ||
        rows * ngroups != rows_per_patch || (long)npatch * ngroups > 65535)
        return GX_ERR_BAD_ARG;
    const int team = pick_team(K);
    const int nb = sk_blocks(rows, team);
    const long npair = (long)npatch * ngroups;
    if (!workspace || workspace_bytes < (size_t)npair * nb * sizeof(float)) return GX_ERR_WORKSPACE;
    const long vs = rows_per_patch * K;                       // view stride
    CePairs pr{ngroups, 2 * vs, rows * K, (long)2 * ngroups * K, (long)K};
    const bool rowlane = K <= 16 && (K & 3) == 0 && (rows_per_patch * K) % 4 == 0 &&
                         ((((uintptr_t)S_all | (uintptr_t)dS_all) & 15) == 0);
#define GX_CE_ROWLANE(KQ_)                                                                                                     \
    hipLaunchKernelGGL(swapped_ce_rowlane_kernel<KQ_>, dim3(nb, (unsigned)npair), dim3(256), 0, (hipStream_t)stream, S_all,     \
                       S_all + vs, rows, (long)K, 1.f / eps, 1.f / temperature, la_all, la_all + (long)ngroups * K, grad_scale, \
                       dS_all, dS_all ? dS_all + vs : nullptr, (long)K, workspace, pr)
    if (rowlane && K == 4) GX_CE_ROWLANE(1);
    else if (rowlane && K == 8) GX_CE_ROWLANE(2);
    else if (rowlane && K == 12) GX_CE_ROWLANE(3);
    else if (rowlane && K == 16) GX_CE_ROWLANE(4);
    else
        hipLaunchKernelGGL(swapped_ce_kernel, dim3(nb, (unsigned)npair), dim3(256), 0, (hipStream_t)stream, S_all, S_all + vs,
                           rows, K, (long)K, team, 1.f / eps, 1.f / temperature, la_all, la_all + (long)ngroups * K, grad_scale,
                           dS_all, dS_all ? dS_all + vs : nullptr, (long)K, workspace, pr);
#undef GX_CE_ROWLANE
    GX_CHECK_LAUNCH();
    launch_partial_reduce(workspace, (int)(npair * nb), 1, loss_scale, loss_beta, loss_out, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// ---- optimizer ---------------------------------------------------------------------------------
// norms2_out[0] = sum x^2 (deterministic two-stage)
size_t gx_sumsq_workspace_bytes(void) { return (size_t)GX_RED_BLOCKS * sizeof(float); }
int gx_sumsq(const float* x, long n, float* out, float* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || n <= 0) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_sumsq_workspace_bytes()) return GX_ERR_WORKSPACE;
    long want = (n + 255) / 256;
    const int nb = (int)(want < GX_RED_BLOCKS ? want : GX_RED_BLOCKS);
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, workspace);
    GX_CHECK_LAUNCH();
    launch_partial_reduce(workspace, nb, 1, 1.f, 0.f, out, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_larc_sgd(float* p, const float* g, float* buf, long n, const float* norms2, float lr, float momentum,
                float trust, int first_step, int use_larc, void* stream) {
    if (!p || !g || !buf || n <= 0 || (use_larc && !norms2)) return GX_ERR_BAD_ARG;
    long want = (n + 255) / 256;
    const int nb = (int)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(larc_sgd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, norms2, lr,
                       momentum, trust, 1e-8f, first_step, use_larc);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// ---- 'image' marginals: hist[bins] = normalised torch.histc(x, bins)+1e-9 with h[0]=h[1] ----
// Three stages so that the sharded form can all-reduce between them (SURVEY par. 8e): data range
// (MIN/MAX), raw bin counts (SUM), then the marginal.  scratch: 2 uints (ordered encodings).
int gx_hist_range(const float* x, long n, unsigned* scratch, float* lohi, void* stream) {
    if (!x || !scratch || !lohi || n <= 0) return GX_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0xFF, 4, s) != hipSuccess) return GX_ERR_LAUNCH;        // min <- +max
    if (hipMemsetAsync(scratch + 1, 0, 4, s) != hipSuccess) return GX_ERR_LAUNCH;       // max <- lowest
    long want = (n + 255) / 256;
    const int nb = (int)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, s, x, n, scratch);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(sk_ord_decode_kernel, dim3(1), dim3(64), 0, s, scratch, 2, lohi);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// hist[bins] (+)= counts of x in `bins` equal bins over [lohi[0], lohi[1]] (torch.histc binning)
int gx_hist_counts(const float* x, long n, const float* lohi, int bins, float* hist, unsigned* scratch, void* stream) {
    if (!x || !lohi || !hist || !scratch || n <= 0 || bins <= 0) return GX_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(sk_ord_encode_kernel, dim3(1), dim3(64), 0, s, lohi, 2, scratch);
    GX_CHECK_LAUNCH();
    if (hipMemsetAsync(hist, 0, (size_t)bins * 4, s) != hipSuccess) return GX_ERR_LAUNCH;
    long want = (n + 255) / 256;
    const int nb = (int)(want < 1024 ? want : 1024);
    launch_histc(x, n, scratch, bins, hist, nb, s);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// h += 1e-9 ; h[0] = h[1] ; h /= sum(h)
int gx_hist_finish(float* hist, int bins, void* stream) {
    if (!hist || bins <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(hist_marginal_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, hist, bins);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_hist_marginal(const float* x, long n, int bins, float* hist, unsigned* scratch, int scratch_words, void* stream) {
    if (!x || !hist || !scratch || n <= 0 || bins <= 0) return GX_ERR_BAD_ARG;
    if (scratch_words < 2 || (bins >= 16384 && scratch_words < 2 + HIST_PARTS)) return GX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0xFF, 4, s) != hipSuccess) return GX_ERR_LAUNCH;        // min <- +max
    if (hipMemsetAsync(scratch + 1, 0, 4, s) != hipSuccess) return GX_ERR_LAUNCH;       // max <- lowest
    if (hipMemsetAsync(hist, 0, (size_t)bins * 4, s) != hipSuccess) return GX_ERR_LAUNCH;
    long want = (n + 255) / 256;
    const int nb = (int)(want < 1024 ? want : 1024);
    hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, s, x, n, scratch);
    GX_CHECK_LAUNCH();
    launch_histc(x, n, scratch, bins, hist, nb, s);
    GX_CHECK_LAUNCH();
    if (bins >= 16384) {            // many bins: two launches over all CUs (scratch words 2 .. 257 hold the block partials)
        float* part = reinterpret_cast<float*>(scratch + 2);
        hipLaunchKernelGGL(hist_sum_parts_kernel, dim3(HIST_PARTS), dim3(256), 0, s, hist, bins, part);
        GX_CHECK_LAUNCH();
        hipLaunchKernelGGL(hist_scale_kernel, dim3(HIST_PARTS), dim3(256), 0, s, hist, bins, part);
    } else {
        hipLaunchKernelGGL(hist_marginal_kernel, dim3(1), dim3(1024), 0, s, hist, bins);
    }
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"

// ============================================================================================
// Pyramid projection (factorised S1+S2+S3): the projection commutes with the
// nearest-neighbour upsampling, so each tap is projected ONCE at its native resolution,
//   P_j = sum_{taps l at resolution r_j} tap_l Wp[:, cols_l]^T        (fp32-MFMA GEMMs)
//   Z_j = P_j + up2(Z_{j-1})                                          (gx_upsample2_add)
// and the projected descriptor of a sampled pixel is ONE row of the full-resolution table Z_J.
// Backward is the adjoint: scatter-add d z rows into G_J, G_{j-1} = sumpool2(G_j), then
// dWp[:, cols_l] += G_j^T tap_l.
namespace gx {

// fine[b][y][x][:] += coarse[b][y/2][x/2][:]     fine is (2rc x 2rc), float4 over D
__global__ void upsample2_add_kernel(const float* __restrict__ coarse, float* __restrict__ fine, int B, int rc,
                                     int D) {
    const int d4n = D >> 2, rf = 2 * rc;
    const long total = (long)B * rf * rf * d4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % d4n) * 4;
        const long pix = i / d4n;
        const int x = (int)(pix % rf), y = (int)((pix / rf) % rf), b = (int)(pix / ((long)rf * rf));
        const f32x4 c = *reinterpret_cast<const f32x4*>(coarse + (((long)b * rc + (y >> 1)) * rc + (x >> 1)) * D + d);
        f32x4* f = reinterpret_cast<f32x4*>(fine + pix * D + d);
        *f = *f + c;
    }
}

// coarse[b][y][x][:] = sum of the 2x2 block of fine (fixed order)
__global__ void sumpool2_kernel(const float* __restrict__ fine, float* __restrict__ coarse, int B, int rc, int D) {
    const int d4n = D >> 2, rf = 2 * rc;
    const long total = (long)B * rc * rc * d4n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int d = (int)(i % d4n) * 4;
        const long pix = i / d4n;
        const int x = (int)(pix % rc), y = (int)((pix / rc) % rc), b = (int)(pix / ((long)rc * rc));
        const float* f = fine + (((long)b * rf + 2 * y) * rf + 2 * x) * D + d;
        const f32x4 s = (*reinterpret_cast<const f32x4*>(f) + *reinterpret_cast<const f32x4*>(f + D)) +
                        (*reinterpret_cast<const f32x4*>(f + (long)rf * D) +
                         *reinterpret_cast<const f32x4*>(f + (long)rf * D + D));
        *reinterpret_cast<f32x4*>(coarse + pix * D + d) = s;
    }
}

// zhat[row] = normalize(Z[b][src(pick[row])]) (zero row where the rotation zero-filled), rnorm[row].
// One wave per row.  NV > 0: D = 256 NV and 16-byte aligned rows — the row is read ONCE, 16 bytes per lane and load (NV = 0: any
// D, dword loads, the row read twice: 2.6 TB/s of rows at D = 512 against 4+ for the vector form)
template <int NV>
__global__ void gather_rows_normalize_kernel(const float* __restrict__ Z, int HW, int D, const int* __restrict__ picks,
                                             const int* __restrict__ maps, int rows_per_img, long rows,
                                             float* __restrict__ zhat, float* __restrict__ rnorm) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int b = (int)(row / rows_per_img);
    int pix = picks[row];
    if (maps) pix = maps[(long)b * HW + pix];
    float* out = zhat + row * D;
    if (pix < 0) {
        for (int c = lane; c < D; c += 64) out[c] = 0.f;
        if (lane == 0) rnorm[row] = 1e12f;
        return;
    }
    const float* src = Z + ((long)b * HW + pix) * D;
    if constexpr (NV > 0) {
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            v[i] = *reinterpret_cast<const f32x4*>(src + 4 * (lane + 64 * i));
            s = fmaf(v[i][0], v[i][0], fmaf(v[i][1], v[i][1], fmaf(v[i][2], v[i][2], fmaf(v[i][3], v[i][3], s))));
        }
        s = gx_wave_sum(s);
        const float rn = 1.f / fmaxf(sqrtf(s), 1e-12f);
#pragma unroll
        for (int i = 0; i < NV; ++i) *reinterpret_cast<f32x4*>(out + 4 * (lane + 64 * i)) = v[i] * rn;
        if (lane == 0) rnorm[row] = rn;
    } else {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) { const float v = src[c]; s = fmaf(v, v, s); }
        s = gx_wave_sum(s);
        const float rn = 1.f / fmaxf(sqrtf(s), 1e-12f);
        for (int c = lane; c < D; c += 64) out[c] = src[c] * rn;
        if (lane == 0) rnorm[row] = rn;
    }
}

// G[b][src(pick[row])][:] += rnorm * (dzh - zh <zh, dzh>)   (normalise-backward fused with the
// adjoint of the row gather; float atomics — two output pixels of a rotated view can sample the same source pixel —, 256
// contiguous bytes per wave-instruction.  The 16-bytes-per-lane form of the gather above is NOT for this kernel: an atomic
// instruction whose lanes are 16 bytes apart touches eight times the cache lines — 3.3 -> 11.2 ms at the K = 5000 bench shape)
__global__ void normalize_bwd_scatter_kernel(const float* __restrict__ dzh, const float* __restrict__ zh,
                                             const float* __restrict__ rnorm, int HW, int D,
                                             const int* __restrict__ picks, const int* __restrict__ maps,
                                             int rows_per_img, long rows, float* __restrict__ G, unsigned* __restrict__ amax) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int b = (int)(row / rows_per_img);
    int pix = picks[row];
    if (maps) pix = maps[(long)b * HW + pix];
    if (pix < 0) return;                       // zero-filled pixel: X row is 0, no gradient reaches Wp
    const float* gp = dzh + row * D;
    const float* zp = zh + row * D;
    float dot = 0.f;
    for (int c = lane; c < D; c += 64) dot = fmaf(gp[c], zp[c], dot);
    dot = gx_wave_sum(dot);
    const float rn = rnorm[row];
    const bool clamped = rn >= 1e12f;
    float* dst = G + ((long)b * HW + pix) * D;
    float m = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float v = clamped ? gp[c] * rn : rn * (gp[c] - zp[c] * dot);
        atomicAdd(dst + c, v);
        m = fmaxf(m, fabsf(v));
    }
    if (amax) {     // the largest single contribution (float bits): times the rows that can meet in one pixel, a bound of |G|
        m = gx_wave_max(m);
        if (lane == 0 && __float_as_uint(m) > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, __float_as_uint(m));
    }
}

}  // namespace gx

extern "C" {

int gx_upsample2_add(const float* coarse, float* fine, int B, int rc, int D, void* stream) {
    if (!coarse || !fine || B <= 0 || rc <= 0 || D <= 0 || (D & 3)) return GX_ERR_BAD_ARG;
    const long total = (long)B * 4 * rc * rc * (D >> 2);
    long want = (total + 255) / 256;
    hipLaunchKernelGGL(gx::upsample2_add_kernel, dim3((int)(want < 16384 ? want : 16384)), dim3(256), 0,
                       (hipStream_t)stream, coarse, fine, B, rc, D);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_sumpool2(const float* fine, float* coarse, int B, int rc, int D, void* stream) {
    if (!coarse || !fine || B <= 0 || rc <= 0 || D <= 0 || (D & 3)) return GX_ERR_BAD_ARG;
    const long total = (long)B * rc * rc * (D >> 2);
    long want = (total + 255) / 256;
    hipLaunchKernelGGL(gx::sumpool2_kernel, dim3((int)(want < 16384 ? want : 16384)), dim3(256), 0,
                       (hipStream_t)stream, fine, coarse, B, rc, D);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_gather_rows_normalize(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps,
                             float* zhat, float* rnorm, void* stream) {
    if (!Z || !picks || !zhat || !rnorm || B <= 0 || HW <= 0 || D <= 0 || rows_per_img <= 0) return GX_ERR_BAD_ARG;
    const long rows = (long)B * rows_per_img;
    const bool vec = (D % 256) == 0 && D <= 1024 && ((((uintptr_t)Z | (uintptr_t)zhat) & 15) == 0);
#define GX_GRN(NV_) hipLaunchKernelGGL(gx::gather_rows_normalize_kernel<NV_>, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, Z, \
                                       HW, D, picks, maps, rows_per_img, rows, zhat, rnorm)
    switch (vec ? D / 256 : 0) {
        case 1: GX_GRN(1); break;
        case 2: GX_GRN(2); break;
        case 3: GX_GRN(3); break;
        case 4: GX_GRN(4); break;
        default: GX_GRN(0); break;
    }
#undef GX_GRN
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_normalize_bwd_scatter(const float* dzhat, const float* zhat, const float* rnorm, int B, int HW, int D,
                             const int* picks, int rows_per_img, const int* maps, float* G, float* amax, void* stream) {
    if (!dzhat || !zhat || !rnorm || !picks || !G || B <= 0 || HW <= 0 || D <= 0 || rows_per_img <= 0)
        return GX_ERR_BAD_ARG;
    const long rows = (long)B * rows_per_img;
    hipLaunchKernelGGL(gx::normalize_bwd_scatter_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       dzhat, zhat, rnorm, HW, D, picks, maps, rows_per_img, rows, G, reinterpret_cast<unsigned*>(amax));
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"

// ============================================================================================
// Single-rank Sinkhorn for few prototypes (K <= 64): one launch per iteration.  The la update
// that follows pass i-1 (fixed-order reduction of its per-block column sums) is folded into the
// prologue of pass i, so an iteration is ONE kernel instead of {pass, reduce, update}; the column
// maxima use a team-per-row kernel.  (The sharded form keeps the 3-kernel sequence because the
// column sums must be all-reduced between passes.)
namespace gx {

// blockIdx.y = problem: the (view, patch) score matrices of one step are independent Sinkhorn
// problems of one shape, so they run as ONE launch per iteration (S, r, c, cmax, la and the
// partial sums are strided by problem).
__global__ void sk_colmax_small_kernel(const float* __restrict__ S, long sprob, long rows, int K, long ld, int team,
                                       unsigned* __restrict__ cmax) {
    __shared__ unsigned smax[64];
    S += (long)blockIdx.y * sprob;
    cmax += (long)blockIdx.y * K;
    if (threadIdx.x < 64) smax[threadIdx.x] = 0u;
    __syncthreads();
    const int tpb = blockDim.x / team, ti = threadIdx.x / team, tl = threadIdx.x % team;
    float v = -INFINITY;
    if (tl < K)
        for (long row = (long)blockIdx.x * tpb + ti; row < rows; row += (long)gridDim.x * tpb)
            v = fmaxf(v, S[row * ld + tl]);
    if (tl < K) atomicMax(&smax[tl], f2ord(v));
    __syncthreads();
    if (threadIdx.x < K) atomicMax(&cmax[threadIdx.x], smax[threadIdx.x]);
}

// la_cur[k] = base[k] + log r_k - log(sum_b upart_prev[b][k]);  base = first_update ? -cmax/eps : la_prev.
// Result left in LDS las[0..K) for every thread (and written to la_out by the caller if wanted).
__device__ __forceinline__ void sk_fold_update(const float* __restrict__ upart_prev, int nblk_prev, int K, int kx,
                                               const float* __restrict__ la_prev, const unsigned* __restrict__ cmax,
                                               bool first_update, float inv_eps, const float* __restrict__ r,
                                               float* las, float* red) {
    const int tx = threadIdx.x % kx, ty = threadIdx.x / kx, ny = blockDim.x / kx;
    float s = 0.f;
    if (tx < K) for (int b = ty; b < nblk_prev; b += ny) s += upart_prev[(long)b * K + tx];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = ny >> 1; h > 0; h >>= 1) {
        if (ty < h) red[threadIdx.x] += red[threadIdx.x + h * kx];
        __syncthreads();
    }
    if (ty == 0 && tx < K) {
        const float base = first_update ? -ord2f(cmax[tx]) * inv_eps : la_prev[tx];
        las[tx] = base + logf(r ? r[tx] : 1.f / (float)K) - logf(red[tx]);
    }
    __syncthreads();
}

// it == 0: u_k = sum_n exp((S_nk - cmax_k)/eps).  it >= 1: fold the update of pass it-1, then sweep.
// Per-problem strides: S sprob, r rstride, c cstride, cmax/la K, partial sums gridDim.x * K.
__global__ void sk_pass_small_kernel(const float* __restrict__ S, long sprob, long rows, int K, long ld, int team,
                                     int kx, float inv_eps, int it, const unsigned* __restrict__ cmax,
                                     const float* __restrict__ r, long rstride, const float* __restrict__ c,
                                     long cstride, float c_uniform, const float* __restrict__ upart_prev,
                                     int nblk_prev, const float* __restrict__ la_prev, float* __restrict__ la_cur,
                                     float* __restrict__ upart) {
    __shared__ float las[64], uacc[64], red[256];
    const long y = blockIdx.y;
    S += y * sprob;
    cmax += y * K;
    if (r) r += y * rstride;
    if (c) c += y * cstride;
    upart_prev += y * nblk_prev * K;                  // nblk_prev = 1: already reduced (and all-reduced) sums
    upart += y * gridDim.x * K;
    la_prev += y * K;
    la_cur += y * K;
    if (threadIdx.x < 64) uacc[threadIdx.x] = 0.f;
    if (it == 0) {
        if (threadIdx.x < K) las[threadIdx.x] = -ord2f(cmax[threadIdx.x]) * inv_eps;
        __syncthreads();
    } else {
        sk_fold_update(upart_prev, nblk_prev, K, kx, la_prev, cmax, it == 1, inv_eps, r, las, red);
    }
    if (blockIdx.x == 0 && threadIdx.x < K) la_cur[threadIdx.x] = las[threadIdx.x];
    const int tpb = blockDim.x / team, ti = threadIdx.x / team, tl = threadIdx.x % team;
    const long nrow_iters = (rows + (long)gridDim.x * tpb - 1) / ((long)gridDim.x * tpb);
    float ul = 0.f;                                           // this lane's column (k = tl) partial: K <= team
    for (long itr = 0; itr < nrow_iters; ++itr) {
        const long row = (itr * gridDim.x + blockIdx.x) * tpb + ti;
        const bool live = row < rows && tl < K;
        const float w = live ? fmaf(S[row * ld + tl], inv_eps, las[tl]) : -INFINITY;
        float lb = 0.f;
        if (it > 0) {
            const float mx = gx_team_max(w, team);
            const float sum = gx_team_sum(live ? expf(w - mx) : 0.f, team);
            lb = logf(c ? c[row < rows ? row : 0] : c_uniform) - mx - logf(sum);
        }
        if (live) ul += expf(w + lb);
    }
    for (int o = team; o < 64; o <<= 1) ul += __shfl_xor(ul, o, 64);
    // the block's column sums in a FIXED order (one slot per wave, added by thread k): LDS float atomics made the step differ
    // from run to run in the last bits
    __syncthreads();                                          // (red: sk_fold_update is done with it)
    if ((threadIdx.x & 63) < team && tl < K) red[(threadIdx.x >> 6) * 64 + tl] = ul;
    __syncthreads();
    if (threadIdx.x < K) {
        float t = red[threadIdx.x];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t += red[w * 64 + threadIdx.x];
        upart[(long)blockIdx.x * K + threadIdx.x] = t;
    }
}

// The same pass with a ROW PER LANE for K = 4 KQ <= 16 (16-B aligned rows): a lane reads its row as KQ float4, so the
// row maximum, the sum of exponentials and the K column contributions are plain register arithmetic — no team shuffles
// (the team form above issues ~50 VALU instructions per score at K = 8, this one ~13), `exp` is bk_exp.  Column sums
// leave through K per-lane accumulators, a wave reduction and LDS atomics once per block.
template <int KQ>
__global__ __launch_bounds__(256) void sk_pass_small_rowlane_kernel(
    const float* __restrict__ S, long sprob, long rows, long ld, int kx, float inv_eps, int it, const unsigned* __restrict__ cmax,
    const float* __restrict__ r, long rstride, const float* __restrict__ c, long cstride, float c_uniform,
    const float* __restrict__ upart_prev, int nblk_prev, const float* __restrict__ la_prev, float* __restrict__ la_cur,
    float* __restrict__ upart) {
    constexpr int K = 4 * KQ;
    __shared__ float las[64], uacc[64], red[256];
    const long y = blockIdx.y;
    S += y * sprob;
    cmax += y * K;
    if (r) r += y * rstride;
    if (c) c += y * cstride;
    upart_prev += y * nblk_prev * K;
    upart += y * gridDim.x * K;
    la_prev += y * K;
    la_cur += y * K;
    if (threadIdx.x < 64) uacc[threadIdx.x] = 0.f;
    if (it == 0) {
        if (threadIdx.x < K) las[threadIdx.x] = -ord2f(cmax[threadIdx.x]) * inv_eps;
        __syncthreads();
    } else {
        sk_fold_update(upart_prev, nblk_prev, K, kx, la_prev, cmax, it == 1, inv_eps, r, las, red);
    }
    if (blockIdx.x == 0 && threadIdx.x < K) la_cur[threadIdx.x] = las[threadIdx.x];
    float lv[K], ul[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { lv[k] = las[k]; ul[k] = 0.f; }
    for (long row = (long)blockIdx.x * 256 + threadIdx.x; row < rows; row += (long)gridDim.x * 256) {
        float w[K];
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(S + row * ld + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) w[4 * q + e] = fmaf(v[e], inv_eps, lv[4 * q + e]);
        }
        if (it > 0) {
            float mx = w[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, w[k]);
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { w[k] = bk_exp(w[k] - mx); sum += w[k]; }
            const float f = (c ? c[row] : c_uniform) / sum;           // exp(lb + mx)
#pragma unroll
            for (int k = 0; k < K; ++k) ul[k] = fmaf(w[k], f, ul[k]);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) ul[k] += bk_exp(w[k]);
        }
    }
    __syncthreads();                                          // (red: sk_fold_update is done with it)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float t = gx_wave_sum(ul[k]);
        if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 64 + k] = t;      // one slot per wave: the block sum in a fixed order
    }
    __syncthreads();
    if (threadIdx.x < K) upart[(long)blockIdx.x * K + threadIdx.x] = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
}

// grid = nprob blocks
__global__ void sk_finish_small_kernel(const float* __restrict__ upart_prev, int nblk_prev, int K, int kx,
                                       const float* __restrict__ la_prev, const unsigned* __restrict__ cmax,
                                       int first_update, float inv_eps, const float* __restrict__ r, long rstride,
                                       float* __restrict__ la_out) {
    __shared__ float las[64], red[256];
    const long y = blockIdx.x;
    sk_fold_update(upart_prev + y * nblk_prev * K, nblk_prev, K, kx, la_prev + y * K, cmax + y * K, first_update != 0,
                   inv_eps, r ? r + y * rstride : nullptr, las, red);
    if (threadIdx.x < K) la_out[y * K + threadIdx.x] = las[threadIdx.x];
}

}  // namespace gx

extern "C" {
// Single-rank Sinkhorn for K <= 64 over `nprob` independent, equally shaped problems (problem i:
// scores S + i*sprob, prototype marginal r + i*rstride or uniform, sample marginal c + i*cstride or
// uniform 1/rows; result la + i*K).  niters + 2 launches for the whole batch; same result per
// problem as gx_sinkhorn.  Workspace: nprob * gx_sinkhorn_workspace_bytes(K).
int gx_sinkhorn_small_batch(const float* S, int nprob, long sprob, long rows, int K, long ld, float eps, int niters,
                            const float* r, long rstride, const float* c, long cstride, float* la,
                            float* workspace, size_t workspace_bytes, void* stream) {
    if (!S || !la || nprob <= 0 || nprob > 65535 || rows <= 0 || K <= 0 || K > 64 || ld < K || eps <= 0.f ||
        niters < 1 || (nprob > 1 && sprob < 0))
        return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < (size_t)nprob * gx_sinkhorn_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int team = gx::pick_team(K);
    const long tpb = 256 / team;
    long nbl = (rows + tpb - 1) / tpb;
    long cap = 4096 / nprob;                                    // ~2 waves of blocks over the whole batch
    cap = cap < 64 ? 64 : (cap > GX_SK_SMALL_BLOCKS ? GX_SK_SMALL_BLOCKS : cap);
    const int nb = (int)(nbl < cap ? nbl : cap);
    int kx = 1;
    while (kx < 64 && kx < K) kx <<= 1;
    const size_t PK = (size_t)nprob * K;
    unsigned* cmax = (unsigned*)workspace;                      // [nprob][K]
    float* labuf = workspace + PK;                              // [2][nprob][K]
    float* part = workspace + 3 * PK;                           // [2][nprob][nb][K]
    if (hipMemsetAsync(cmax, 0, PK * 4, st) != hipSuccess) return GX_ERR_LAUNCH;
    hipLaunchKernelGGL(gx::sk_colmax_small_kernel, dim3(nb, nprob), dim3(256), 0, st, S, sprob, rows, K, ld, team, cmax);
    GX_CHECK_LAUNCH();
    const float c_uniform = 1.f / (float)rows;
    // a row per lane for K in {4, 8, 12, 16} with 16-B aligned rows (GX_SK_ROWLANE=0: the team form, for A/B runs)
    const bool rowlane = K <= 16 && (K & 3) == 0 && (ld & 3) == 0 && (sprob & 3) == 0 && (((uintptr_t)S) & 15) == 0;
    for (int it = 0; it < niters; ++it) {
        float* up_prev = part + (size_t)((it + 1) & 1) * nb * PK;
        float* up_cur = part + (size_t)(it & 1) * nb * PK;
        float* la_p = labuf + (size_t)((it + 1) & 1) * PK;
        float* la_c = labuf + (size_t)(it & 1) * PK;
#define GX_SK_ROWLANE(KQ_)                                                                                                    \
    hipLaunchKernelGGL(gx::sk_pass_small_rowlane_kernel<KQ_>, dim3(nb, nprob), dim3(256), 0, st, S, sprob, rows, ld, kx,      \
                       1.f / eps, it, cmax, r, rstride, c, cstride, c_uniform, up_prev, nb, la_p, la_c, up_cur)
        if (rowlane && K == 4) GX_SK_ROWLANE(1);
        else if (rowlane && K == 8) GX_SK_ROWLANE(2);
        else if (rowlane && K == 12) GX_SK_ROWLANE(3);
        else if (rowlane && K == 16) GX_SK_ROWLANE(4);
        else
            hipLaunchKernelGGL(gx::sk_pass_small_kernel, dim3(nb, nprob), dim3(256), 0, st, S, sprob, rows, K, ld, team, kx,
                               1.f / eps, it, cmax, r, rstride, c, cstride, c_uniform, up_prev, nb, la_p, la_c, up_cur);
#undef GX_SK_ROWLANE
        GX_CHECK_LAUNCH();
    }
    const int last = niters - 1;
    hipLaunchKernelGGL(gx::sk_finish_small_kernel, dim3(nprob), dim3(256), 0, st, part + (size_t)(last & 1) * nb * PK,
                       nb, K, kx, labuf + (size_t)(last & 1) * PK, cmax, niters == 1 ? 1 : 0, 1.f / eps, r, rstride, la);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// One problem (see gx_sinkhorn_small_batch).  Workspace: gx_sinkhorn_workspace_bytes(K).
int gx_sinkhorn_small(const float* S, long rows, int K, long ld, float eps, int niters, const float* r,
                      const float* c, float* la, float* workspace, size_t workspace_bytes, void* stream) {
    return gx_sinkhorn_small_batch(S, 1, 0, rows, K, ld, eps, niters, r, 0, c, 0, la, workspace, workspace_bytes, stream);
}

// ---- the same batch, staged for row-sharded scores (one rank holds `rows` of each problem): the
// host all-reduces the nprob*K column maxima once (MAX) and the nprob*K column sums after every
// pass (SUM); 2 launches + 1 collective per iteration for the WHOLE batch.
//   colmax : local column maxima as floats -> cmax_out [nprob*K]
//   pass   : it == 0 takes the all-reduced maxima cmax_in; it >= 1 the all-reduced sums u_prev of
//            pass it-1 (folds the la update), sweeps, and leaves this rank's sums in u_out [nprob*K];
//            c_uniform = 1 / (rows over all ranks)
//   finish : last la update from the all-reduced sums of the last pass -> la [nprob*K]
static int sk_batch_geometry(int nprob, long rows, int K, int* team, int* nb, int* kx) {
    *team = gx::pick_team(K);
    const long tpb = 256 / *team;
    long nbl = (rows + tpb - 1) / tpb;
    long cap = 4096 / nprob;
    cap = cap < 64 ? 64 : (cap > GX_SK_SMALL_BLOCKS ? GX_SK_SMALL_BLOCKS : cap);
    *nb = (int)(nbl < cap ? nbl : cap);
    *kx = 1;
    while (*kx < 64 && *kx < K) *kx <<= 1;
    return 0;
}
int gx_sinkhorn_small_batch_colmax(const float* S, int nprob, long sprob, long rows, int K, long ld, float* workspace,
                                   size_t workspace_bytes, float* cmax_out, void* stream) {
    if (!S || !cmax_out || nprob <= 0 || nprob > 65535 || rows <= 0 || K <= 0 || K > 64 || ld < K) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < (size_t)nprob * gx_sinkhorn_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int team, nb, kx;
    sk_batch_geometry(nprob, rows, K, &team, &nb, &kx);
    const size_t PK = (size_t)nprob * K;
    if (hipMemsetAsync(workspace, 0, PK * 4, st) != hipSuccess) return GX_ERR_LAUNCH;
    hipLaunchKernelGGL(gx::sk_colmax_small_kernel, dim3(nb, nprob), dim3(256), 0, st, S, sprob, rows, K, ld, team,
                       (unsigned*)workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(gx::sk_ord_decode_kernel, dim3(gx_cdiv((long)PK, 256)), dim3(256), 0, st,
                       (const unsigned*)workspace, (int)PK, cmax_out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_sinkhorn_small_batch_pass(const float* S, int nprob, long sprob, long rows, int K, long ld, float eps, int it,
                                 const float* r, long rstride, const float* c, long cstride, float c_uniform,
                                 const float* cmax_in, const float* u_prev, float* workspace, size_t workspace_bytes,
                                 float* u_out, void* stream) {
    if (!S || !u_out || nprob <= 0 || nprob > 65535 || rows <= 0 || K <= 0 || K > 64 || ld < K || eps <= 0.f || it < 0 ||
        (it == 0 && !cmax_in) || (it > 0 && !u_prev))
        return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < (size_t)nprob * gx_sinkhorn_workspace_bytes(K)) return GX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int team, nb, kx;
    sk_batch_geometry(nprob, rows, K, &team, &nb, &kx);
    const size_t PK = (size_t)nprob * K;
    unsigned* cmax = (unsigned*)workspace;
    float* labuf = workspace + PK;
    float* part = workspace + 3 * PK;
    if (it == 0) {
        hipLaunchKernelGGL(gx::sk_ord_encode_kernel, dim3(gx_cdiv((long)PK, 256)), dim3(256), 0, st, cmax_in, (int)PK, cmax);
        GX_CHECK_LAUNCH();
    }
    {   // the sharded form of the pass takes the same row-per-lane kernel (gx_sinkhorn_small_batch)
        const bool rowlane = K <= 16 && (K & 3) == 0 && (ld & 3) == 0 && (sprob & 3) == 0 && (((uintptr_t)S) & 15) == 0;
        const float* up = u_prev ? u_prev : part;
        float* la_p = labuf + (size_t)((it + 1) & 1) * PK;
        float* la_c = labuf + (size_t)(it & 1) * PK;
#define GX_SK_ROWLANE(KQ_)                                                                                                    \
    hipLaunchKernelGGL(gx::sk_pass_small_rowlane_kernel<KQ_>, dim3(nb, nprob), dim3(256), 0, st, S, sprob, rows, ld, kx,      \
                       1.f / eps, it, cmax, r, rstride, c, cstride, c_uniform, up, 1, la_p, la_c, part)
        if (rowlane && K == 4) GX_SK_ROWLANE(1);
        else if (rowlane && K == 8) GX_SK_ROWLANE(2);
        else if (rowlane && K == 12) GX_SK_ROWLANE(3);
        else if (rowlane && K == 16) GX_SK_ROWLANE(4);
        else
            hipLaunchKernelGGL(gx::sk_pass_small_kernel, dim3(nb, nprob), dim3(256), 0, st, S, sprob, rows, K, ld, team, kx,
                               1.f / eps, it, cmax, r, rstride, c, cstride, c_uniform, up, 1, la_p, la_c, part);
#undef GX_SK_ROWLANE
    }
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(gx::partial_reduce_kernel, dim3((K + kx - 1) / kx, nprob), dim3(256), 0, st, part, nb, K, 1.f, 0.f,
                       u_out, kx);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_sinkhorn_small_batch_finish(int nprob, int K, float eps, int niters, const float* r, long rstride,
                                   const float* u_last, const float* workspace, float* la, void* stream) {
    if (!u_last || !workspace || !la || nprob <= 0 || K <= 0 || K > 64 || niters < 1 || eps <= 0.f) return GX_ERR_BAD_ARG;
    int kx = 1;
    while (kx < 64 && kx < K) kx <<= 1;
    const size_t PK = (size_t)nprob * K;
    const int last = niters - 1;
    hipLaunchKernelGGL(gx::sk_finish_small_kernel, dim3(nprob), dim3(256), 0, (hipStream_t)stream, u_last, 1, K, kx,
                       workspace + PK + (size_t)(last & 1) * PK, (const unsigned*)workspace, niters == 1 ? 1 : 0, 1.f / eps,
                       r, rstride, la);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
}  // extern "C"

// ============================================================================================
// Small-prototype fast path (K <= 16, e.g. BASELINE configs with 8 / 16 prototypes): with so few
// prototypes the prototype layer is HBM-bound row work, not a GEMM.  Two fused kernels replace
// {row gather, normalise, scores GEMM} and {dz GEMM, dWc GEMM, db column sums, normalise-backward,
// scatter}: each reads its rows exactly once; prototype rows live in LDS; row sums use wavefront
// shuffles (a halving butterfly reduces the K partial dot products with K-1 + log2(64/K) shuffles).
namespace gx {

constexpr int PS_MAXV = 4;     // float4 per lane: D <= 1024

// Reduce v[0..KP) across the 64 lanes; afterwards every lane holds the total of ONE index,
// k = ps_lane_k<KP>(lane).
template <int KP>
__device__ __forceinline__ float ps_reduce(float (&v)[KP], int lane) {
    int o = 32;
#pragma unroll
    for (int half = KP / 2; half >= 1; half >>= 1) {
        const bool hi = (lane & o) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            // read both elements as opaque scalars first: otherwise the compiler turns `hi ? v[a] : v[b]`
            // into v[hi ? a : b], a run-time index into the register array, and lowers it to a 16-step
            // compare/select chain per access (480 v_cndmask per reduction instead of 30)
            float lo_v = v[i], hi_v = v[half + i];
            asm volatile("" : "+v"(lo_v), "+v"(hi_v));
            const float keep = hi ? hi_v : lo_v;
            const float send = hi ? lo_v : hi_v;
            v[i] = keep + __shfl_xor(send, o, 64);
        }
        o >>= 1;
    }
    float r = v[0];
    for (; o >= 1; o >>= 1) r += __shfl_xor(r, o, 64);
    return r;
}
template <int KP>
__device__ __forceinline__ int ps_lane_k(int lane) {
    int k = 0, o = 32;
#pragma unroll
    for (int half = KP / 2; half >= 1; half >>= 1) { if (lane & o) k += half; o >>= 1; }
    return k;
}
template <int KP>
__device__ __forceinline__ bool ps_lane_owner(int lane) { return (lane & ((64 / KP) - 1)) == 0; }

// zhat[row] = normalize(Z[b][src(pick)]) ; rnorm ; scores[row][k] = <zhat, Wc[k]> + bias[k]
template <int KP>
__global__ __launch_bounds__(256) void proto_fwd_small_kernel(const float* __restrict__ Z, int HW, int D,
                                                              const int* __restrict__ picks,
                                                              const int* __restrict__ maps, int rows_per_img,
                                                              long rows, const float* __restrict__ Wc,
                                                              const float* __restrict__ bias, int K,
                                                              float* __restrict__ zhat, float* __restrict__ rnorm,
                                                              float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];          // [KP][D] (rows >= K are zero)
    for (int i = threadIdx.x; i < KP * D; i += blockDim.x) wsm[i] = (i / D) < K ? Wc[i] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, nv = D >> 8;                      // float4 per lane (D % 256 == 0)
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    const int myk = ps_lane_k<KP>(lane);
    const bool owner = ps_lane_owner<KP>(lane) && myk < K;
    const float myb = (owner && bias) ? bias[myk] : 0.f;
    for (long row = wave0; row < rows; row += nwave) {
        const int b = (int)(row / rows_per_img);
        int pix = picks[row];
        if (maps) pix = maps[(long)b * HW + pix];
        f32x4 z[PS_MAXV];
        float ss = 0.f;
        if (pix >= 0) {
            const float* src = Z + ((long)b * HW + pix) * D;
#pragma unroll
            for (int i = 0; i < PS_MAXV; ++i)
                if (i < nv) {
                    z[i] = *reinterpret_cast<const f32x4*>(src + (i * 64 + lane) * 4);
                    ss += z[i][0] * z[i][0] + z[i][1] * z[i][1] + z[i][2] * z[i][2] + z[i][3] * z[i][3];
                }
        } else {
#pragma unroll
            for (int i = 0; i < PS_MAXV; ++i) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        ss = gx_wave_sum(ss);
        const float rn = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        float p[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) p[k] = 0.f;
#pragma unroll
        for (int i = 0; i < PS_MAXV; ++i)
            if (i < nv) {
                z[i] *= rn;
                *reinterpret_cast<f32x4*>(zhat + row * D + (i * 64 + lane) * 4) = z[i];
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wsm + k * D + (i * 64 + lane) * 4);
                    p[k] += z[i][0] * w[0] + z[i][1] * w[1] + z[i][2] * w[2] + z[i][3] * w[3];
                }
            }
        const float sc = ps_reduce<KP>(p, lane);
        if (owner) scores[row * K + myk] = sc + myb;
        if (lane == 0) rnorm[row] = rn;
    }
}

// Per row: dzh = dS[row] Wc ; dz = normalize-backward(dzh) scattered (atomics) into G ;
// block partials of gWc[k][:] += dS[row][k] zhat[row] and gb[k] += dS[row][k].
template <int KP>
__global__ __launch_bounds__(256) void proto_bwd_small_kernel(const float* __restrict__ dS, const float* __restrict__ zhat,
                                                              const float* __restrict__ rnorm, int HW, int D,
                                                              const int* __restrict__ picks,
                                                              const int* __restrict__ maps, int rows_per_img,
                                                              long rows, const float* __restrict__ Wc, int K,
                                                              float* __restrict__ G, float* __restrict__ part_w,
                                                              float* __restrict__ part_b) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];          // [KP][D] prototypes, then reused
    for (int i = threadIdx.x; i < KP * D; i += blockDim.x) wsm[i] = (i / D) < K ? Wc[i] : 0.f;
    __syncthreads();
    // lane l owns columns c = j*64 + l (j < D/64 <= 8): every wave-instruction of the atomic
    // scatter then covers 256 CONTIGUOUS bytes (the full-rate shape of global float atomics)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nj = D >> 6;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + wid;
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    float gw[KP][8];                                                      // D <= 512 on this path
    float gbk[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        gbk[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) gw[k][j] = 0.f;
    }
    for (long row = wave0; row < rows; row += nwave) {
        const int b = (int)(row / rows_per_img);
        int pix = picks[row];
        if (maps) pix = maps[(long)b * HW + pix];
        float ds[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) ds[k] = k < K ? dS[row * K + k] : 0.f;
        const float rn = rnorm[row];
        float z[8], dzh[8];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < nj) {
                const int c = j * 64 + lane;
                z[j] = zhat[row * D + c];
                float a = 0.f;
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    a = fmaf(wsm[k * D + c], ds[k], a);
                    gw[k][j] = fmaf(z[j], ds[k], gw[k][j]);
                }
                dzh[j] = a;
                dot = fmaf(a, z[j], dot);
            }
#pragma unroll
        for (int k = 0; k < KP; ++k) gbk[k] += ds[k];
        dot = gx_wave_sum(dot);
        if (pix >= 0) {                                                   // zero-filled pixel: no gradient to Wp
            const bool clamped = rn >= 1e12f;
            float* dst = G + ((long)b * HW + pix) * D;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nj) atomicAdd(dst + j * 64 + lane, clamped ? dzh[j] * rn : rn * (dzh[j] - z[j] * dot));
        }
    }
    // block partial: sum the 4 waves through LDS (fixed order), one [K][D] slab + K bias sums per block
    __syncthreads();                                                       // prototypes no longer needed
    float* red = wsm;
#pragma unroll
    for (int k = 0; k < KP; ++k) {          // fully unrolled: gw[k] must stay a register (no dynamic index)
        if (k < K) {                        // K is block-uniform, so the barriers below are uniform too
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nj) red[wid * D + j * 64 + lane] = gw[k][j];
            __syncthreads();
            for (int c = threadIdx.x; c < D; c += blockDim.x)
                part_w[((long)blockIdx.x * K + k) * D + c] = (red[c] + red[D + c]) + (red[2 * D + c] + red[3 * D + c]);
            __syncthreads();
        }
    }
    // bias partial: every lane of a wave holds the same gbk
    if (lane == 0)
        for (int k = 0; k < K; ++k) red[wid * KP + k] = gbk[k];
    __syncthreads();
    if (threadIdx.x < K)
        part_b[(long)blockIdx.x * K + threadIdx.x] = (red[threadIdx.x] + red[KP + threadIdx.x]) +
                                                     (red[2 * KP + threadIdx.x] + red[3 * KP + threadIdx.x]);
}

}  // namespace gx

extern "C" {

#define GX_PS_BLOCKS 1024
// workspace for gx_proto_bwd_small: block partials of gWc and gb
size_t gx_proto_small_workspace_bytes(int K, int D) { return (size_t)GX_PS_BLOCKS * ((size_t)K * D + K) * sizeof(float); }

// Fused forward for K <= 16 prototypes: zhat (rows, D), rnorm (rows), scores (rows, K).
// Needs D % 256 == 0, D <= 1024.
int gx_proto_fwd_small(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps,
                       const float* Wc, const float* bias, int K, float* zhat, float* rnorm, float* scores,
                       void* stream) {
    if (!Z || !picks || !Wc || !zhat || !rnorm || !scores || B <= 0 || HW <= 0 || rows_per_img <= 0) return GX_ERR_BAD_ARG;
    if (K < 1 || K > 16 || (D & 255) || D > 256 * gx::PS_MAXV) return GX_ERR_BAD_ARG;
    const long rows = (long)B * rows_per_img;
    long nb = (rows + 3) / 4;
    if (nb > 2048) nb = 2048;
    const int kp = K <= 8 ? 8 : 16;
    const size_t lds = (size_t)kp * D * sizeof(float);
    if (kp == 8)
        hipLaunchKernelGGL((gx::proto_fwd_small_kernel<8>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, Z, HW, D,
                           picks, maps, rows_per_img, rows, Wc, bias, K, zhat, rnorm, scores);
    else
        hipLaunchKernelGGL((gx::proto_fwd_small_kernel<16>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, Z, HW,
                           D, picks, maps, rows_per_img, rows, Wc, bias, K, zhat, rnorm, scores);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// Fused backward for K <= 16: G += scatter(normalize-bwd(dS Wc)); gWc = beta*gWc + dS^T zhat;
// gb = beta*gb + colsum(dS).  Needs D % 256 == 0, D <= 512.
int gx_proto_bwd_small(const float* dS, const float* zhat, const float* rnorm, int B, int HW, int D, const int* picks,
                       int rows_per_img, const int* maps, const float* Wc, int K, float* G, float* gWc, float* gb,
                       float beta, float* workspace, size_t workspace_bytes, void* stream) {
    if (!dS || !zhat || !rnorm || !picks || !Wc || !G || !gWc || !gb || B <= 0 || HW <= 0 || rows_per_img <= 0)
        return GX_ERR_BAD_ARG;
    if (K < 1 || K > 16 || (D & 255) || D > 512) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_proto_small_workspace_bytes(K, D)) return GX_ERR_WORKSPACE;
    const long rows = (long)B * rows_per_img;
    long nb = (rows + 3) / 4;
    if (nb > GX_PS_BLOCKS) nb = GX_PS_BLOCKS;
    float* part_w = workspace;
    float* part_b = workspace + (size_t)GX_PS_BLOCKS * K * D;
    const int kp = K <= 8 ? 8 : 16;
    size_t lds = (size_t)kp * D * sizeof(float);
    if (lds < (size_t)4 * D * sizeof(float)) lds = (size_t)4 * D * sizeof(float);
    if (kp == 8)
        hipLaunchKernelGGL((gx::proto_bwd_small_kernel<8>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, dS, zhat,
                           rnorm, HW, D, picks, maps, rows_per_img, rows, Wc, K, G, part_w, part_b);
    else
        hipLaunchKernelGGL((gx::proto_bwd_small_kernel<16>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, dS,
                           zhat, rnorm, HW, D, picks, maps, rows_per_img, rows, Wc, K, G, part_w, part_b);
    GX_CHECK_LAUNCH();
    gx::launch_partial_reduce(part_w, (int)nb, K * D, 1.f, beta, gWc, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    gx::launch_partial_reduce(part_b, (int)nb, K, 1.f, beta, gb, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"

// ============================================================================================
// Low-rank prototype backward (K <= 15).  With few prototypes the gradient that reaches a
// projected pixel is confined to span{Wc_1..Wc_K, Z[q]}:
//     dz(row) = rn * (dS Wc - zhat <zhat, dS Wc>) = sum_k (rn dS_k) Wc_k - (rn^2 dot) Z[q],
//     dot = <zhat, dS Wc> = sum_k dS_k (score_k - b_k)          (zhat = rn Z[q], q = src(pick))
// so a row contributes K+1 COEFFICIENTS to its pixel instead of a D-vector: the per-patch
// scatter shrinks from D*4 = 2 KB of float atomics per row to (K+1)*4 B, and zhat is never
// written.  Once per view (after all patches) one dense pass rebuilds
//     G[q] = sum_k C[q][k] Wc_k + C[q][K] Z[q]        and        gWc += C[:, :K]^T Z
// (gWc[k] = sum_rows dS_k zhat = sum_q C[q][k] Z[q] with the same coefficients).
namespace gx {

constexpr int LR_MAXK = 16;            // low-rank path: K <= 16 prototypes
// coefficient slots per pixel: K prototype slots + 1, padded to a float4 multiple (16 for K <= 15, 20 for K = 16)
__host__ __device__ constexpr int lr_stride(int K) { return (K + 1 + 3) & ~3; }

// scores[row][k] = rn * <Z[q], Wc_k> + b_k ; rnorm[row] = rn.  One wave per row.
// RAW: scores[row][k] = <Z[q], Wc_k>, rnorm[row] = |Z[q]|^2 (what gx_fineq_fwd needs of the level below the finest).
template <int KP, bool RAW = false>
__global__ __launch_bounds__(256) void proto_fwd_lowrank_kernel(const float* __restrict__ Z, int HW, int D,
                                                                const int* __restrict__ picks,
                                                                const int* __restrict__ maps, int rows_per_img,
                                                                long rows, const float* __restrict__ Wc,
                                                                const float* __restrict__ bias, int K,
                                                                float* __restrict__ rnorm, float* __restrict__ scores,
                                                                long ld_rn = 1, long ld_sc = 0) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];          // [KP][D] (rows >= K are zero)
    for (int i = threadIdx.x; i < KP * D; i += blockDim.x) wsm[i] = (i / D) < K ? Wc[i] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, nv = D >> 8;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    const int myk = ps_lane_k<KP>(lane);
    const bool owner = ps_lane_owner<KP>(lane) && myk < K;
    const float myb = (owner && bias) ? bias[myk] : 0.f;
    if (ld_sc == 0) ld_sc = K;
    for (long row = wave0; row < rows; row += nwave) {
        const int b = (int)(row / rows_per_img);
        int pix = picks ? picks[row] : (int)(row - (long)b * rows_per_img);   // no picks: the whole table, in order
        if (maps) pix = maps[(long)b * HW + pix];
        float p[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) p[k] = 0.f;
        float ss = 0.f;
        if (pix >= 0) {
            const float* src = Z + ((long)b * HW + pix) * D;
#pragma unroll
            for (int i = 0; i < PS_MAXV; ++i)
                if (i < nv) {
                    const f32x4 z = *reinterpret_cast<const f32x4*>(src + (i * 64 + lane) * 4);
                    ss += z[0] * z[0] + z[1] * z[1] + z[2] * z[2] + z[3] * z[3];
#pragma unroll
                    for (int k = 0; k < KP; ++k) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(wsm + k * D + (i * 64 + lane) * 4);
                        p[k] += z[0] * w[0] + z[1] * w[1] + z[2] * w[2] + z[3] * w[3];
                    }
                }
        }
        ss = gx_wave_sum(ss);
        const float rn = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        const float sc = ps_reduce<KP>(p, lane);
        if (owner) scores[row * ld_sc + myk] = RAW ? sc : fmaf(sc, rn, myb);
        if (lane == 0) rnorm[row * ld_rn] = RAW ? ss : rn;
    }
}

// Sampled rows of the all-pixel score table: scores[row][:] = T[q][:], rnorm[row] = rnT[q] with
// q = map(pick(row)); a pixel rotated out of the frame (q < 0) has a zero descriptor: score = bias,
// rnorm = 1/1e-12 like F.normalize's eps clamp.  One thread per (row, k).
__global__ void gather_scores_kernel(const float* __restrict__ T, const float* __restrict__ rnT,
                                     const float* __restrict__ bias, int HW, int K, const int* __restrict__ picks,
                                     const int* __restrict__ maps, int rows_per_img, long rows,
                                     float* __restrict__ scores, float* __restrict__ rnorm) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * K) return;
    const long row = idx / K;
    const int k = (int)(idx - row * K);
    const int b = (int)(row / rows_per_img);
    int pix = picks[row];
    if (maps) pix = maps[(long)b * HW + pix];
    const long q = (long)b * HW + pix;
    scores[idx] = pix >= 0 ? T[q * K + k] : (bias ? bias[k] : 0.f);
    if (k == 0) rnorm[row] = pix >= 0 ? rnT[q] : 1e12f;
}

// One TEAM of 16 (K <= 15) or 32 lanes per row, lane tl owning coefficient tl: a row's K+1 float
// atomics leave as ONE wave instruction over contiguous addresses (a 64-B line per row) instead
// of K+1 instructions of 64 scattered lines each.  Also the per-block partial of gb.
__global__ void proto_bwd_lowrank_kernel(const float* __restrict__ dS, const float* __restrict__ scores,
                                         const float* __restrict__ rnorm, const float* __restrict__ bias, int HW, int K,
                                         const int* __restrict__ picks, const int* __restrict__ maps, int rows_per_img,
                                         long rows, float* __restrict__ C, float* __restrict__ part_b) {
    __shared__ float sb[LR_MAXK];
    const int ls = lr_stride(K);
    const int team = (K + 1 <= 16) ? 16 : 32;
    const int tpb = blockDim.x / team, ti = threadIdx.x / team, tl = threadIdx.x % team;
    if (threadIdx.x < LR_MAXK) sb[threadIdx.x] = 0.f;
    __syncthreads();
    const float myb = (bias && tl < K) ? bias[tl] : 0.f;
    float gbl = 0.f;
    const long niter = (rows + (long)gridDim.x * tpb - 1) / ((long)gridDim.x * tpb);
    for (long it = 0; it < niter; ++it) {
        const long row = (it * gridDim.x + blockIdx.x) * tpb + ti;
        const bool live = row < rows;
        float d = 0.f, sc = 0.f, rn = 0.f;
        long q = -1;
        if (live) {
            const int b = (int)(row / rows_per_img);
            int pix = picks[row];
            if (maps) pix = maps[(long)b * HW + pix];
            q = pix >= 0 ? (long)b * HW + pix : -1;
            rn = rnorm[row];
            if (tl < K) {
                d = dS[row * K + tl];
                sc = scores[row * K + tl] - myb;
            }
        }
        const float dot = gx_team_sum(d * sc, team);
        gbl += d;
        if (q >= 0 && tl <= K) atomicAdd(C + q * ls + tl, tl < K ? rn * d : -rn * rn * dot);
    }
    for (int o = team; o < 64; o <<= 1) gbl += __shfl_xor(gbl, o, 64);
    if ((threadIdx.x & 63) < team && tl < K) atomicAdd(&sb[tl], gbl);
    __syncthreads();
    if (threadIdx.x < K) part_b[(long)blockIdx.x * K + threadIdx.x] = sb[threadIdx.x];
}

// The whole step at once (low-rank path, both views in one 2B-image batch): row r enumerates
// (patch p, view v, image b, sample n); its pixel is map(picks[p][b][n]) of image v*B + b.
// shift > 0: the tables live `shift` pyramid levels below the picks' resolution (no taps above: every pixel of a
// 2^shift x 2^shift block has its ancestor's descriptor); wlog2 = log2 of the picks' image width.
__device__ __forceinline__ long lr_row_pixel(long r, int B, int N, int HW, const int* __restrict__ picks,
                                             const int* __restrict__ maps, int shift, int wlog2) {
    const int n = (int)(r % N);
    const long t = r / N;
    const int b = (int)(t % B), v = (int)((t / B) & 1);
    const long p = t / (2 * B);
    const int img = v * B + b;
    int pix = picks[(p * B + b) * N + n];
    if (maps) pix = maps[(long)img * HW + pix];
    if (pix < 0) return -1;
    if (shift > 0) {
        const int y = pix >> wlog2, x = pix & ((1 << wlog2) - 1);
        return (long)img * (HW >> (2 * shift)) + (((y >> shift) << (wlog2 - shift)) + (x >> shift));
    }
    return (long)img * HW + pix;
}
__global__ void gather_scores_batch_kernel(const float* __restrict__ T, const float* __restrict__ rnT,
                                           const float* __restrict__ bias, int B, int HW, int K, int N,
                                           const int* __restrict__ picks, const int* __restrict__ maps, long rows,
                                           float* __restrict__ scores, float* __restrict__ rnorm, int shift, int wlog2) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * K) return;
    const long r = idx / K;
    const int k = (int)(idx - r * K);
    const long q = lr_row_pixel(r, B, N, HW, picks, maps, shift, wlog2);
    scores[idx] = q >= 0 ? T[q * K + k] : (bias ? bias[k] : 0.f);
    if (k == 0) rnorm[r] = q >= 0 ? rnT[q] : 1e12f;
}
// DET (swav_args['deterministic']): the coefficients are accumulated as 64-bit FIXED-POINT integers (2^-44 units) — integer
// adds commute, so the sums no longer depend on the order the atomics land in — into Cfix, which lr_fix_to_float_kernel turns
// into the float grid the consumers read.  Resolution 5.7e-14 against coefficients of 1e-6 .. 1e-3 (a contribution below 1e-9
// loses relative precision, not the sum), range +-5e5.  The bias-gradient partial of a block is summed in a fixed order in
// both forms.
constexpr float LR_FIX_SCALE = 17592186044416.f;                 // 2^44
__global__ void lr_fix_to_float_kernel(const long long* __restrict__ Cfix, long n, float* __restrict__ C) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        C[i] = (float)((double)Cfix[i] * (1.0 / (double)LR_FIX_SCALE));
}
template <bool DET>
__global__ void proto_bwd_lowrank_batch_kernel(const float* __restrict__ dS, const float* __restrict__ scores,
                                               const float* __restrict__ rnorm, const float* __restrict__ bias, int B,
                                               int HW, int K, int N, const int* __restrict__ picks,
                                               const int* __restrict__ maps, long rows, float* __restrict__ C,
                                               long long* __restrict__ Cfix, float* __restrict__ part_b, int shift, int wlog2) {
    __shared__ float sb[4 * LR_MAXK];
    const int ls = lr_stride(K);
    const int team = (K + 1 <= 16) ? 16 : 32;
    const int tpb = blockDim.x / team, ti = threadIdx.x / team, tl = threadIdx.x % team;
    const float myb = (bias && tl < K) ? bias[tl] : 0.f;
    float gbl = 0.f;
    const long niter = (rows + (long)gridDim.x * tpb - 1) / ((long)gridDim.x * tpb);
    for (long it = 0; it < niter; ++it) {
        const long row = (it * gridDim.x + blockIdx.x) * tpb + ti;
        const bool live = row < rows;
        float d = 0.f, sc = 0.f, rn = 0.f;
        long q = -1;
        if (live) {
            q = lr_row_pixel(row, B, N, HW, picks, maps, shift, wlog2);
            rn = rnorm[row];
            if (tl < K) { d = dS[row * K + tl]; sc = scores[row * K + tl] - myb; }
        }
        const float dot = gx_team_sum(d * sc, team);
        gbl += d;
        if (q >= 0 && tl <= K) {
            const float v = tl < K ? rn * d : -rn * rn * dot;
            if (DET) atomicAdd(reinterpret_cast<unsigned long long*>(Cfix + q * ls + tl), (unsigned long long)__float2ll_rn(v * LR_FIX_SCALE));
            else atomicAdd(C + q * ls + tl, v);
        }
    }
    for (int o = team; o < 64; o <<= 1) gbl += __shfl_xor(gbl, o, 64);
    if ((threadIdx.x & 63) < team && tl < K) sb[(threadIdx.x >> 6) * LR_MAXK + tl] = gbl;      // one slot per wave
    __syncthreads();
    if (threadIdx.x < K) {
        float t = sb[threadIdx.x];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t += sb[w * LR_MAXK + threadIdx.x];
        part_b[(long)blockIdx.x * K + threadIdx.x] = t;
    }
}

// Dense finish: G[q] = sum_k C[q][k] Wc_k + C[q][K] Z[q]; block partials of gWc[k] += C[q][k] Z[q].
// blockIdx.y = 256-column slab of the D columns; one wave per group of 4 pixels (4 row slabs of 1 KB
// in flight per lane), lane l owns the float4 columns slab + 4 l.  POOL: the group is the 2x2 block
// under coarse pixel `g` and its sum is written to Gpool[g] — the first sum-pooling step of the
// pyramid backward, while the rows are in registers.
template <int KP, bool POOL>
__global__ __launch_bounds__(256) void proto_lowrank_finish_kernel(const float* __restrict__ C, const float* __restrict__ Z,
                                                                   long npix, int rc, int D, const float* __restrict__ Wc,
                                                                   int K, float* __restrict__ G, float* __restrict__ Gpool,
                                                                   float* __restrict__ part_w, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];          // [KP][256] (this slab of Wc)
    const int cb = blockIdx.y * 256;
    for (int i = threadIdx.x; i < KP * 256; i += blockDim.x) wsm[i] = (i >> 8) < K ? Wc[(long)(i >> 8) * D + cb + (i & 255)] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, col = cb + lane * 4;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + wid;
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    const int ls = lr_stride(K), rf = 2 * rc;
    const long ngroups = POOL ? npix / 4 : (npix + 3) / 4;
    f32x4 gw[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) gw[k] = zero4();
    for (long g = wave0; g < ngroups; g += nwave) {
        long pq[4];
        if (POOL) {
            const long per = (long)rc * rc;
            const long b = g / per, rem = g - b * per;
            const long cy = rem / rc, cx = rem - cy * rc;
            pq[0] = (b * rf + 2 * cy) * rf + 2 * cx;
            pq[1] = pq[0] + 1; pq[2] = pq[0] + rf; pq[3] = pq[2] + 1;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) pq[j] = g * 4 + j;
        }
        float cv[4];
        f32x4 z[4], g0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = pq[j] < npix;
            cv[j] = (ok && lane < ls) ? C[pq[j] * ls + lane] : 0.f;
            z[j] = ok ? *reinterpret_cast<const f32x4*>(Z + pq[j] * D + col) : zero4();
            g0[j] = (ok && accumulate) ? *reinterpret_cast<const f32x4*>(G + pq[j] * D + col) : zero4();
        }
        // coefficients as wave-uniform scalars (lane k of cv[j] holds C[q_j][k]); a never-sampled pixel has
        // all-zero coefficients and simply yields a zero row.  A prototype float4 is read from LDS once
        // for the four rows.
        f32x4 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = g0[j] + __shfl(cv[j], K, 64) * z[j];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wsm + k * 256 + lane * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cv[j]), k));
                a[j] += c * w;
                gw[k] += c * z[j];
            }
        }
        f32x4 pool = zero4();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (pq[j] < npix) *reinterpret_cast<f32x4*>(G + pq[j] * D + col) = a[j];
            pool += a[j];
        }
        if (POOL) *reinterpret_cast<f32x4*>(Gpool + g * D + col) = pool;
    }
    __syncthreads();
    float* red = wsm;                                  // [4 waves][256]  (KP >= 4)
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (k < K) {
            *reinterpret_cast<f32x4*>(red + wid * 256 + lane * 4) = gw[k];
            __syncthreads();
            for (int c2 = threadIdx.x; c2 < 256; c2 += blockDim.x)
                part_w[((long)blockIdx.x * K + k) * D + cb + c2] =
                    (red[c2] + red[256 + c2]) + (red[512 + c2] + red[768 + c2]);
            __syncthreads();
        }
    }
}

// Finest pyramid level of the low-rank path: fine += upsample2(coarse) AND the score / rnorm tables
// of the finished rows in the same pass.  One wave per coarse pixel = 2x2 block of fine pixels:
// one coarse row + four fine rows in flight per lane.  ADD = false: the table is already finished (the
// upsample-add rode on the projection GEMM's epilogue, gx_gemm_f32_upadd): tables only, nothing stored back.
// RAW (with ADD = false): sc_table[q * ld_sc + k] = <fine[q], Wc_k>, rn_table[q * ld_rn] = |fine[q]|^2 — no normalisation,
// no bias: the shares of the level below the finest that gx_fineq_fwd takes (gx_proto_raw_tables).
template <int KP, bool ADD, bool RAW = false>
__global__ __launch_bounds__(256) void upsample2_add_scores_kernel(const float* __restrict__ coarse, float* __restrict__ fine,
                                                                   long ncoarse, int rc, int D,
                                                                   const float* __restrict__ Wc,
                                                                   const float* __restrict__ bias, int K,
                                                                   float* __restrict__ rn_table,
                                                                   float* __restrict__ sc_table, long ld_rn = 1,
                                                                   long ld_sc = 0) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];          // [KP][D] (rows >= K are zero)
    for (int i = threadIdx.x; i < KP * D; i += blockDim.x) wsm[i] = (i / D) < K ? Wc[i] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, nv = D >> 8, rf = 2 * rc;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    const int myk = ps_lane_k<KP>(lane);
    const bool owner = ps_lane_owner<KP>(lane) && myk < K;
    const float myb = (owner && bias) ? bias[myk] : 0.f;
    const long per = (long)rc * rc;
    if (ld_sc == 0) ld_sc = K;
    for (long g = wave0; g < ncoarse; g += nwave) {
        const long b = g / per, rem = g - b * per;
        const long cy = rem / rc, cx = rem - cy * rc;
        long pq[4];
        pq[0] = (b * rf + 2 * cy) * rf + 2 * cx;
        pq[1] = pq[0] + 1; pq[2] = pq[0] + rf; pq[3] = pq[2] + 1;
        f32x4 c[2], z[4][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            c[i] = (ADD && i < nv) ? *reinterpret_cast<const f32x4*>(coarse + g * D + (i * 64 + lane) * 4) : zero4();
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                z[j][i] = i < nv ? *reinterpret_cast<const f32x4*>(fine + pq[j] * D + (i * 64 + lane) * 4) : zero4();
        // finished rows (stored), their squared norms, and the K dot products of all four pixels:
        // a prototype float4 is read from LDS once and used for the four rows
        float p[4][KP], ss[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ss[j] = 0.f;
#pragma unroll
            for (int k = 0; k < KP; ++k) p[j][k] = 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (i < nv) {                       // (no `break`: the loop must unroll, or z[.][i] is indexed at run time)
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = ADD ? z[j][i] + c[i] : z[j][i];
                    if (ADD) *reinterpret_cast<f32x4*>(fine + pq[j] * D + (i * 64 + lane) * 4) = v[j];
                    ss[j] += v[j][0] * v[j][0] + v[j][1] * v[j][1] + v[j][2] * v[j][2] + v[j][3] * v[j][3];
                }
#pragma unroll
                for (int k = 0; k < KP; ++k) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wsm + k * D + (i * 64 + lane) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        p[j][k] += v[j][0] * w[0] + v[j][1] * w[1] + v[j][2] * w[2] + v[j][3] * w[3];
                }
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float sj = gx_wave_sum(ss[j]);
            const float rn = 1.f / fmaxf(sqrtf(sj), 1e-12f);
            float pj[KP];                                 // a 1-D copy keeps the accumulators in registers
#pragma unroll
            for (int k = 0; k < KP; ++k) pj[k] = p[j][k];
            const float sc = ps_reduce<KP>(pj, lane);
            if (owner) sc_table[pq[j] * ld_sc + myk] = RAW ? sc : fmaf(sc, rn, myb);
            if (lane == 0) rn_table[pq[j] * ld_rn] = RAW ? sj : rn;
        }
    }
}

}  // namespace gx

extern "C" {

int gx_lowrank_stride(int K) { return gx::lr_stride(K); }

// scores (rows, K) and rnorm (rows) straight from the projected-pixel table (zhat is not written).
int gx_proto_fwd_lowrank(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps,
                         const float* Wc, const float* bias, int K, float* rnorm, float* scores, void* stream) {
    if (!Z || !Wc || !rnorm || !scores || B <= 0 || HW <= 0 || rows_per_img <= 0) return GX_ERR_BAD_ARG;
    if (!picks && (rows_per_img != HW || maps)) return GX_ERR_BAD_ARG;      // no picks: every pixel, in order
    if (K < 1 || K > gx::LR_MAXK || (D & 255) || D > 256 * gx::PS_MAXV) return GX_ERR_BAD_ARG;
    const long rows = (long)B * rows_per_img;
    long nb = (rows + 3) / 4;
    if (nb > 2048) nb = 2048;
    const int kp = K <= 8 ? 8 : 16;
    const size_t lds = (size_t)kp * D * sizeof(float);
    if (kp == 8)
        hipLaunchKernelGGL((gx::proto_fwd_lowrank_kernel<8>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, Z, HW,
                           D, picks, maps, rows_per_img, rows, Wc, bias, K, rnorm, scores);
    else
        hipLaunchKernelGGL((gx::proto_fwd_lowrank_kernel<16>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, Z, HW,
                           D, picks, maps, rows_per_img, rows, Wc, bias, K, rnorm, scores);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// dots [rows][K] = Z Wc^T and sumsq [rows] = |Z[row]|^2 of a finished table Z [rows][D] — no normalisation, no bias:
// the inputs gx_fineq_fwd takes from the level below the finest.
int gx_proto_raw_tables(const float* Z, long rows, int D, const float* Wc, int K, float* sumsq, long ld_sumsq, float* dots,
                        long ld_dots, void* stream) {
    if (!Z || !Wc || !sumsq || !dots || rows <= 0 || rows > 0x7FFFFFFFL || ld_sumsq < 1 || ld_dots < K) return GX_ERR_BAD_ARG;
    if (K < 1 || K > gx::LR_MAXK || (D & 255) || D > 256 * gx::PS_MAXV) return GX_ERR_BAD_ARG;
    long nb = (rows + 3) / 4;
    if (nb > 2048) nb = 2048;
    const int kp = K <= 8 ? 8 : 16;
    const size_t lds = (size_t)kp * D * sizeof(float);
    if (D <= 512 && rows % 4 == 0) {
        // four rows in flight per wave (upsample2_add_scores_kernel's row loop: 4.8 TB/s against 2.3 with one): rows are
        // independent here, so ANY grouping by four will do — the kernel's 2 x 2 blocks of a (1 x 2) x (rows / 2) "image"
        const long ncoarse = rows / 4;
        long nbk = (ncoarse + 3) / 4;
#ifndef RAWTAB_BLOCKS
#define RAWTAB_BLOCKS 4096     // lab: tools/lab/rawtab_lab.sh
#endif
        if (nbk > RAWTAB_BLOCKS) nbk = RAWTAB_BLOCKS;
        // B = 1 image of 2 rc x 2 rc pixels needs a square; use the degenerate geometry rc = rows / 4, rf = 2 rc with
        // per = rc * rc > ncoarse: every group g < ncoarse then has b = 0, cy = 0, cx = g: pixels 2 g, 2 g + 1, rf + 2 g, rf + 2 g + 1
        // -> rows {2g, 2g+1} of the first half and {2g, 2g+1} + rows/2 of the second: all rows exactly once
        const int rcq = (int)ncoarse;
        if (kp == 8)
            hipLaunchKernelGGL((gx::upsample2_add_scores_kernel<8, false, true>), dim3((int)nbk), dim3(256), lds, (hipStream_t)stream,
                               nullptr, const_cast<float*>(Z), ncoarse, rcq, D, Wc, nullptr, K, sumsq, dots, ld_sumsq, ld_dots);
        else
            hipLaunchKernelGGL((gx::upsample2_add_scores_kernel<16, false, true>), dim3((int)nbk), dim3(256), lds, (hipStream_t)stream,
                               nullptr, const_cast<float*>(Z), ncoarse, rcq, D, Wc, nullptr, K, sumsq, dots, ld_sumsq, ld_dots);
        GX_CHECK_LAUNCH();
        return GX_OK;
    }
    if (kp == 8)
        hipLaunchKernelGGL((gx::proto_fwd_lowrank_kernel<8, true>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, Z,
                           (int)rows, D, nullptr, nullptr, (int)rows, rows, Wc, nullptr, K, sumsq, dots, ld_sumsq, ld_dots);
    else
        hipLaunchKernelGGL((gx::proto_fwd_lowrank_kernel<16, true>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, Z,
                           (int)rows, D, nullptr, nullptr, (int)rows, rows, Wc, nullptr, K, sumsq, dots, ld_sumsq, ld_dots);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// Rows of the all-pixel score table (gx_proto_fwd_lowrank with picks == NULL) for one patch.
int gx_gather_scores(const float* table, const float* rn_table, const float* bias, int B, int HW, int K,
                     const int* picks, int rows_per_img, const int* maps, float* scores, float* rnorm, void* stream) {
    if (!table || !rn_table || !picks || !scores || !rnorm || B <= 0 || HW <= 0 || K <= 0 || rows_per_img <= 0)
        return GX_ERR_BAD_ARG;
    const long rows = (long)B * rows_per_img;
    hipLaunchKernelGGL(gx::gather_scores_kernel, dim3((unsigned)gx_cdiv(rows * K, 256)), dim3(256), 0, (hipStream_t)stream,
                       table, rn_table, bias, HW, K, picks, maps, rows_per_img, rows, scores, rnorm);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// All patches and both views of a step in one launch: tables over the 2B-image batch (view v = images
// [v*B, (v+1)*B)), picks [npatch][B][N], maps [2B][HW] or NULL; rows are ordered (patch, view, image, sample),
// i.e. scores is [npatch][2][B*N][K].
static int lr_wlog2(int HW, int shift) {          // log2 of the image width for a table `shift` levels below the picks; -1: bad
    if (shift < 0) return -1;
    if (shift == 0) return 0;
    int w = 0;
    while ((1L << (2 * w)) < HW) ++w;
    return ((1L << (2 * w)) == HW && shift <= w) ? w : -1;
}
int gx_gather_scores_batch(const float* table, const float* rn_table, const float* bias, int B, int HW, int K,
                           const int* picks, int npatch, int N, const int* maps, int shift, float* scores, float* rnorm,
                           void* stream) {
    if (!table || !rn_table || !picks || !scores || !rnorm || B <= 0 || HW <= 0 || K <= 0 || npatch <= 0 || N <= 0)
        return GX_ERR_BAD_ARG;
    const int wlog2 = lr_wlog2(HW, shift);
    if (wlog2 < 0) return GX_ERR_BAD_ARG;
    const long rows = (long)npatch * 2 * B * N;
    hipLaunchKernelGGL(gx::gather_scores_batch_kernel, dim3((unsigned)gx_cdiv(rows * K, 256)), dim3(256), 0,
                       (hipStream_t)stream, table, rn_table, bias, B, HW, K, N, picks, maps, rows, scores, rnorm, shift, wlog2);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// the matching coefficient scatter: C [2B*HW][gx_lowrank_stride(K)] += ..., gb = beta*gb + colsum(dS).
// workspace >= 1024*K floats.
int gx_proto_bwd_lowrank_batch(const float* dS, const float* scores, const float* rnorm, const float* bias, int B, int HW,
                               int K, const int* picks, int npatch, int N, const int* maps, int shift, float* C, long long* Cfix,
                               float* gb, float beta, float* workspace, size_t workspace_bytes, void* stream) {
    if (!dS || !scores || !rnorm || !picks || !C || !gb || B <= 0 || HW <= 0 || npatch <= 0 || N <= 0 || K < 1 ||
        K > gx::LR_MAXK)
        return GX_ERR_BAD_ARG;
    const int wlog2 = lr_wlog2(HW, shift);
    if (wlog2 < 0) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < (size_t)1024 * K * sizeof(float)) return GX_ERR_WORKSPACE;
    const long rows = (long)npatch * 2 * B * N;
    long nb = (rows + 15) / 16;
    if (nb > 1024) nb = 1024;
    if (Cfix) {     // deterministic form: fixed-point accumulation, then the float grid from ALL that has been accumulated so far
        hipLaunchKernelGGL(gx::proto_bwd_lowrank_batch_kernel<true>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, dS, scores,
                           rnorm, bias, B, HW, K, N, picks, maps, rows, C, Cfix, workspace, shift, wlog2);
        GX_CHECK_LAUNCH();
        const long n = (long)2 * B * (HW >> (2 * shift)) * gx::lr_stride(K);
        hipLaunchKernelGGL(gx::lr_fix_to_float_kernel, dim3(gx_cdiv(n, 1024) < 4096 ? gx_cdiv(n, 1024) : 4096), dim3(256), 0,
                           (hipStream_t)stream, Cfix, n, C);
    } else {
        hipLaunchKernelGGL(gx::proto_bwd_lowrank_batch_kernel<false>, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, dS, scores,
                           rnorm, bias, B, HW, K, N, picks, maps, rows, C, Cfix, workspace, shift, wlog2);
    }
    GX_CHECK_LAUNCH();
    gx::launch_partial_reduce(workspace, (int)nb, K, 1.f, beta, gb, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// C [B*HW][gx_lowrank_stride(K)] += per-row coefficients (float atomics); gb = beta*gb + colsum(dS).
// workspace >= 1024*K floats.
int gx_proto_bwd_lowrank(const float* dS, const float* scores, const float* rnorm, const float* bias, int B, int HW,
                         int K, const int* picks, int rows_per_img, const int* maps, float* C, float* gb, float beta,
                         float* workspace, size_t workspace_bytes, void* stream) {
    if (!dS || !scores || !rnorm || !picks || !C || !gb || B <= 0 || HW <= 0 || rows_per_img <= 0 || K < 1 ||
        K > gx::LR_MAXK)
        return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < (size_t)1024 * K * sizeof(float)) return GX_ERR_WORKSPACE;
    const long rows = (long)B * rows_per_img;
    long nb = (rows + 15) / 16;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(gx::proto_bwd_lowrank_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, dS, scores, rnorm,
                       bias, HW, K, picks, maps, rows_per_img, rows, C, workspace);
    GX_CHECK_LAUNCH();
    gx::launch_partial_reduce(workspace, (int)nb, K, 1.f, beta, gb, (hipStream_t)stream);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// G (npix, D) = C[:, :K] Wc + C[:, K] .* Z ; gWc = beta*gWc + C[:, :K]^T Z.   D in {256, 512}.
// Gpool != NULL: the pixels are B images of (2rc x 2rc) and Gpool (B*rc*rc, D) receives the 2x2
// sum-pooled G (what gx_sumpool2 would compute from G).  rc is ignored when Gpool == NULL.
// workspace >= gx_proto_small_workspace_bytes(K, D).
int gx_proto_lowrank_finish(const float* C, const float* Z, long npix, int rc, int D, const float* Wc, int K, float* G,
                            int accumulate, float* Gpool, float* gWc, float beta, float* workspace, size_t workspace_bytes,
                            void* stream) {
    if (!C || !Z || !Wc || !G || !gWc || npix <= 0 || K < 1 || K > gx::LR_MAXK || (D & 255) || D > 512)
        return GX_ERR_BAD_ARG;
    if (Gpool && (rc <= 0 || npix % (4L * rc * rc) != 0)) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_proto_small_workspace_bytes(K, D)) return GX_ERR_WORKSPACE;
    long nb = (npix + 15) / 16;
    if (nb > GX_PS_BLOCKS) nb = GX_PS_BLOCKS;
    const int kp = K <= 8 ? 8 : 16;
    const size_t lds = (size_t)kp * 256 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define GX_FINISH(KP, POOL)                                                                                         \
    hipLaunchKernelGGL((gx::proto_lowrank_finish_kernel<KP, POOL>), dim3((int)nb, D / 256), dim3(256), lds, st, C, Z, \
                       npix, rc, D, Wc, K, G, Gpool, workspace, accumulate)
    if (kp == 8) { if (Gpool) GX_FINISH(8, true); else GX_FINISH(8, false); }
    else         { if (Gpool) GX_FINISH(16, true); else GX_FINISH(16, false); }
#undef GX_FINISH
    GX_CHECK_LAUNCH();
    gx::launch_partial_reduce(workspace, (int)nb, K * D, 1.f, beta, gWc, st);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// fine (B, 2rc, 2rc, D) += upsample2(coarse (B, rc, rc, D)); rn_table (B*4rc^2), sc_table (B*4rc^2, K)
// = what gx_proto_fwd_lowrank(picks = NULL) gives on the finished table.  D in {256, 512}, K <= 16.
// coarse == NULL: fine is already finished (gx_gemm_f32_upadd); only the tables are written.
int gx_upsample2_add_scores(const float* coarse, float* fine, int B, int rc, int D, const float* Wc, const float* bias,
                            int K, float* rn_table, float* sc_table, void* stream) {
    if (!fine || !Wc || !rn_table || !sc_table || B <= 0 || rc <= 0) return GX_ERR_BAD_ARG;
    if (K < 1 || K > gx::LR_MAXK || (D & 255) || D > 512) return GX_ERR_BAD_ARG;
    const long ncoarse = (long)B * rc * rc;
    long nb = (ncoarse + 3) / 4;
    if (nb > 4096) nb = 4096;
    const int kp = K <= 8 ? 8 : 16;
    const size_t lds = (size_t)kp * D * sizeof(float);
#define GX_LAUNCH_UAS(KP_, ADD_)                                                                                   \
    hipLaunchKernelGGL((gx::upsample2_add_scores_kernel<KP_, ADD_>), dim3((int)nb), dim3(256), lds, (hipStream_t)stream, \
                       coarse, fine, ncoarse, rc, D, Wc, bias, K, rn_table, sc_table)
    if (kp == 8) { if (coarse) GX_LAUNCH_UAS(8, true); else GX_LAUNCH_UAS(8, false); }
    else { if (coarse) GX_LAUNCH_UAS(16, true); else GX_LAUNCH_UAS(16, false); }
#undef GX_LAUNCH_UAS
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"
