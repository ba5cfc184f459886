// One-shot fine-tune head: the reference's OneShotSegmentor
// (hfc_with_swav/swav_clustering.py:697-758) trained by OneShotPipeline.run_trainer
// (src/one_shot_pipeline.py:533-581: Adam + CrossEntropyLoss on one labelled image) and scored by
// run_tests (:595-886: argmax + lib/metrics/segmentation.py IoUs).
//
// Layout: every activation lives ZERO-PADDED and flat, [B][H+2P][W+2P][C] with P = the largest
// dilation of the head, so a (dilated) 3x3 tap is a constant row shift of the flat pixel index and
// each of conv forward / weight gradient / data gradient is nine plain GEMMs on row-offset views
// (gx_gemm_f32: NT, TN with split-K, NN).  The kernels here are the glue between those GEMMs:
// padding, bias + LeakyReLU + border mask (and its backward), softmax cross-entropy with its
// gradient written straight into the padded layout, Adam, argmax over the interior, and the
// confusion matrix the IoU metrics are read from.
#include "gx_common.h"

namespace gx {

__device__ __forceinline__ bool hd_interior(long r, int Hp, int Wp, int P) {
    const int x = (int)(r % Wp), y = (int)((r / Wp) % Hp);
    return x >= P && x < Wp - P && y >= P && y < Hp - P;
}

// out [B][H+2P][W+2P][C] = zero-padded in [B][H][W][C]
__global__ void hd_pad_kernel(const float* __restrict__ in, int B, int H, int W, int C, int P, float* __restrict__ out) {
    const int Hp = H + 2 * P, Wp = W + 2 * P;
    const long total = (long)B * Hp * Wp * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;
        const int x = (int)(r % Wp), y = (int)((r / Wp) % Hp), b = (int)(r / ((long)Wp * Hp));
        float v = 0.f;
        if (x >= P && x < Wp - P && y >= P && y < Hp - P) v = in[(((long)b * H + (y - P)) * W + (x - P)) * C + c];
        out[i] = v;
    }
}

// y[r][c] = interior(r) ? act(y[r][c]) : 0   over all R rows (bias was added by the first tap's GEMM)
__global__ void hd_act_mask_kernel(float* __restrict__ y, long R, int C, int Hp, int Wp, int P, int act) {
    const long total = R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        float v = 0.f;
        if (hd_interior(r, Hp, Wp, P)) { v = y[i]; if (act) v = v > 0.f ? v : 0.2f * v; }
        y[i] = v;
    }
}

// A 3x3 (dilated) conv with FEW output channels as ONE product and a tap sum: y9[r][t C + c] = sum_ci x[r][ci] W[t][c][ci] is a
// single GEMM over the un-shifted rows with the nine taps stacked along N (9 C <= 128 columns: one N tile instead of nine
// launches that each fill 12 of a tile's 128 columns), and   y[r][c] = interior(r) ? act(bias[c] + sum_t y9[r + shift_t][t C + c]) : 0.
// Taps are added in the order the nine accumulating GEMMs added them: (acc_0 + bias) + acc_1 + ... — same values.
__global__ void hd_tap_sum_kernel(const float* __restrict__ y9, long ld9, long R, int C, int Hp, int Wp, int P, int dil,
                                  const float* __restrict__ bias, int act, float* __restrict__ y) {
    const long total = R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i - r * C);
        float v = 0.f;
        if (hd_interior(r, Hp, Wp, P)) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const long sh = (long)((t / 3) - 1) * dil * Wp + ((t % 3) - 1) * dil;
                const float a = y9[(r + sh) * ld9 + t * C + c];
                v = t == 0 ? a + (bias ? bias[c] : 0.f) : v + a;
            }
            if (act) v = v > 0.f ? v : 0.2f * v;
        }
        y[i] = v;
    }
}

// dy[r][c] = interior(r) ? dy[r][c] * (act ? lrelu'(y[r][c]) : 1) : 0
__global__ void hd_act_mask_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, long R, int C, int Hp, int Wp,
                                       int P, int act) {
    const long total = R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        float v = 0.f;
        if (hd_interior(r, Hp, Wp, P)) { v = dy[i]; if (act && !(y[i] > 0.f)) v *= 0.2f; }
        dy[i] = v;
    }
}

// nn.CrossEntropyLoss (mean over the B*H*W pixels) of padded logits against labels [B][H][W], and
// d loss / d logits in the padded layout (zero on the border).  One thread per padded pixel.
__global__ void hd_softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int B, int H,
                                     int W, int C, int P, float inv_n, float* __restrict__ dlogits,
                                     float* __restrict__ loss_part) {
    __shared__ float red[16];
    const int Hp = H + 2 * P, Wp = W + 2 * P;
    const long R = (long)B * Hp * Wp;
    float lacc = 0.f;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (long)gridDim.x * blockDim.x) {
        const int x = (int)(r % Wp), y = (int)((r / Wp) % Hp), b = (int)(r / ((long)Wp * Hp));
        const bool in = x >= P && x < Wp - P && y >= P && y < Hp - P;
        const float* l = logits + r * C;
        float* d = dlogits ? dlogits + r * C : nullptr;
        if (!in) {
            if (d) for (int c = 0; c < C; ++c) d[c] = 0.f;
            continue;
        }
        const int t = (int)labels[((long)b * H + (y - P)) * W + (x - P)];
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, l[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(l[c] - mx);
        const float lse = mx + logf(se);
        const bool valid = t >= 0 && t < C;        // a label outside [0, C) contributes neither loss nor gradient
        if (valid) lacc += lse - l[t];
        if (d) for (int c = 0; c < C; ++c) d[c] = valid ? (expf(l[c] - lse) - (c == t ? 1.f : 0.f)) * inv_n : 0.f;
    }
    const float tot = gx_block_sum(lacc, red);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = tot * inv_n;
}

// out[0] = sum of n partials (one block, fixed order)
__global__ void hd_sum_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    const float tot = gx_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = tot;
}

// torch.optim.Adam step (eps 1e-8, no weight decay, no amsgrad)
__global__ void hd_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                               float* __restrict__ v, long n, float step_size, float beta1, float beta2, float inv_sqrt_bc2,
                               float eps) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    }
}

// pred [B][H][W] = argmax_c of the padded logits' interior (first maximum, like tensor.max(1)[1])
__global__ void hd_argmax_kernel(const float* __restrict__ logits, int B, int H, int W, int C, int P,
                                 long long* __restrict__ pred) {
    const int Hp = H + 2 * P, Wp = W + 2 * P;
    const long n = (long)B * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long)W * H));
        const float* l = logits + (((long)b * Hp + y + P) * Wp + x + P) * C;
        int best = 0; float bv = l[0];
        for (int c = 1; c < C; ++c) if (l[c] > bv) { bv = l[c]; best = c; }
        pred[i] = best;
    }
}

// conf[g][p] += 1 for every pixel with ground truth g and prediction p (both clamped to [0, C))
__global__ void hd_confusion_kernel(const long long* __restrict__ gt, const long long* __restrict__ pred, long n, int C,
                                    unsigned long long* __restrict__ conf) {
    extern __shared__ unsigned int sconf[];
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) sconf[i] = 0u;
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long long g = gt[i], p = pred[i];
        if (g >= 0 && g < C && p >= 0 && p < C) atomicAdd(&sconf[(int)g * C + (int)p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x)
        if (sconf[i]) atomicAdd(&conf[i], (unsigned long long)sconf[i]);
}

static inline int hd_blocks(long total) { long w = (total + 255) / 256; return (int)(w < 4096 ? (w < 1 ? 1 : w) : 4096); }

}  // namespace gx

using namespace gx;

extern "C" {

int gx_head_pad(const float* in, int B, int H, int W, int C, int P, float* out, void* stream) {
    if (!in || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || P < 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(hd_pad_kernel, dim3(hd_blocks((long)B * (H + 2 * P) * (W + 2 * P) * C)), dim3(256), 0,
                       (hipStream_t)stream, in, B, H, W, C, P, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_head_act_mask(float* y, int B, int H, int W, int C, int P, int act, void* stream) {
    if (!y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || P < 0) return GX_ERR_BAD_ARG;
    const long R = (long)B * (H + 2 * P) * (W + 2 * P);
    hipLaunchKernelGGL(hd_act_mask_kernel, dim3(hd_blocks(R * C)), dim3(256), 0, (hipStream_t)stream, y, R, C, H + 2 * P,
                       W + 2 * P, P, act);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_head_tap_sum(const float* y9, long ld9, int B, int H, int W, int C, int P, int dil, const float* bias, int act, float* y,
                    void* stream) {
    if (!y9 || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || P < 0 || dil < 1 || dil > P || ld9 < 9L * C) return GX_ERR_BAD_ARG;
    const long R = (long)B * (H + 2 * P) * (W + 2 * P);
    hipLaunchKernelGGL(hd_tap_sum_kernel, dim3(hd_blocks(R * C)), dim3(256), 0, (hipStream_t)stream, y9, ld9, R, C, H + 2 * P,
                       W + 2 * P, P, dil, bias, act, y);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_head_act_mask_bwd(float* dy, const float* y, int B, int H, int W, int C, int P, int act, void* stream) {
    if (!dy || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || P < 0) return GX_ERR_BAD_ARG;
    const long R = (long)B * (H + 2 * P) * (W + 2 * P);
    hipLaunchKernelGGL(hd_act_mask_bwd_kernel, dim3(hd_blocks(R * C)), dim3(256), 0, (hipStream_t)stream, dy, y, R, C,
                       H + 2 * P, W + 2 * P, P, act);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// loss_out[0] = mean CE; dlogits may be NULL (evaluation).  workspace >= 1024 floats.
int gx_head_softmax_ce(const float* logits, const long long* labels, int B, int H, int W, int C, int P, float* dlogits,
                       float* loss_out, float* workspace, size_t workspace_bytes, void* stream) {
    if (!logits || !labels || !loss_out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C > 4096 || P < 0) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < 1024 * sizeof(float)) return GX_ERR_WORKSPACE;
    const long R = (long)B * (H + 2 * P) * (W + 2 * P);
    long nb = (R + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(hd_softmax_ce_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, logits, labels, B, H, W, C, P,
                       1.f / (float)((long)B * H * W), dlogits, workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(hd_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, (int)nb, loss_out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_head_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                 int step, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return GX_ERR_BAD_ARG;
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(hd_adam_kernel, dim3(hd_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)(lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_head_argmax(const float* logits, int B, int H, int W, int C, int P, long long* pred, void* stream) {
    if (!logits || !pred || B <= 0 || H <= 0 || W <= 0 || C <= 0 || P < 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(hd_argmax_kernel, dim3(hd_blocks((long)B * H * W)), dim3(256), 0, (hipStream_t)stream, logits, B, H,
                       W, C, P, pred);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// conf [C][C] (uint64, zeroed here): conf[g][p] = #pixels with ground truth g predicted p
int gx_confusion(const long long* gt, const long long* pred, long n, int C, unsigned long long* conf, void* stream) {
    if (!gt || !pred || !conf || n <= 0 || C <= 0 || C > 64) return GX_ERR_BAD_ARG;
    if (hipMemsetAsync(conf, 0, (size_t)C * C * 8, (hipStream_t)stream) != hipSuccess) return GX_ERR_LAUNCH;
    hipLaunchKernelGGL(hd_confusion_kernel, dim3(hd_blocks(n) < 256 ? hd_blocks(n) : 256), dim3(256), (size_t)C * C * 4,
                       (hipStream_t)stream, gt, pred, n, C, conf);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"
