// The non-linear projection heads the plugin surface names beside 'linear' (hfc_with_swav/swav_clustering.py:250-269):
//   '1-layer'  Linear(hlen, D, bias=False) -> LeakyReLU(0.01)
//   '2-layer'  Linear(hlen, D, bias=False) -> BatchNorm1d(D) -> LeakyReLU(0.01) -> Linear(D, D, bias=False)
//              -> BatchNorm1d(D) -> Tanh          (BatchNorm always in train mode: the reference never calls .eval())
// No shipped config selects them, so they are row passes beside the GEMMs of the generic path and not fused into
// them: the first Linear stays the pyramid (or the panel GEMM), the gathered rows go through the kernels below, the
// second Linear is a plain (rows, D) x (D, D) GEMM.  Everything here is HBM-bound: one read and one write of a
// (rows, D) fp32 panel per pass (rows = images x patch_size; 655 MB at 16 x 20 000 x 512).
#include "gx_common.h"

namespace gx {

// out[row] = Z[b][src(pick[row])] (zero row where the rotation zero-filled).  One wave per row.
__global__ void gather_rows_kernel(const float* __restrict__ Z, int HW, int D, const int* __restrict__ picks,
                                   const int* __restrict__ maps, int rows_per_img, long rows, float* __restrict__ out) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int b = (int)(row / rows_per_img);
    int pix = picks[row];
    if (maps) pix = maps[(long)b * HW + pix];
    float* dst = out + row * D;
    if (pix < 0) {
        for (int c = lane; c < D; c += 64) dst[c] = 0.f;
        return;
    }
    const float* src = Z + ((long)b * HW + pix) * D;
    for (int c = lane; c < D; c += 64) dst[c] = src[c];
}

// G[b][src(pick[row])][:] += g[row][:]   (adjoint of the gather; float atomics, 256 contiguous bytes per instruction)
__global__ void scatter_rows_add_kernel(const float* __restrict__ g, int HW, int D, const int* __restrict__ picks,
                                        const int* __restrict__ maps, int rows_per_img, long rows, float* __restrict__ G) {
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int b = (int)(row / rows_per_img);
    int pix = picks[row];
    if (maps) pix = maps[(long)b * HW + pix];
    if (pix < 0) return;                       // zero-filled pixel: the descriptor is 0, nothing reaches Wp through it
    const float* src = g + row * D;
    float* dst = G + ((long)b * HW + pix) * D;
    for (int c = lane; c < D; c += 64) atomicAdd(dst + c, src[c]);
}

// y = x > 0 ? x : slope x, in place; and its backward from the OUTPUT (slope > 0 keeps the sign): dy *= y > 0 ? 1 : slope
__global__ void leaky_relu_kernel(float* __restrict__ x, long total, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const float v = x[i];
        x[i] = v > 0.f ? v : slope * v;
    }
}
__global__ void leaky_relu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, long total, float slope) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        if (!(y[i] > 0.f)) dy[i] *= slope;
}

// part[blk][c] = sum over this block's rows of (x - center[c]) or its square: fixed order, no atomics
__global__ void col_moment_kernel(const float* __restrict__ x, long rows, int D, const float* __restrict__ center,
                                  int square, float* __restrict__ part) {
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float m = center ? center[c] : 0.f;
        float s = 0.f;
        for (long r = r0; r < r1; ++r) {
            const float v = x[r * D + c] - m;
            s += square ? v * v : v;
        }
        part[(long)blockIdx.x * D + c] = s;
    }
}
__global__ void col_parts_finish_kernel(const float* __restrict__ part, int nblk, int D, float scale, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(long)b * D + c];
    out[c] = s * scale;
}

// pre = gamma xhat + beta, xhat = (x - mean) rsqrt(var + eps);  act 0: LeakyReLU(slope), 1: tanh
__device__ __forceinline__ float bn_act(float pre, int act, float slope) {
    return act == 1 ? tanhf(pre) : (pre > 0.f ? pre : slope * pre);
}
// d act / d pre from pre (tanh recomputed: no fourth panel kept for it)
__device__ __forceinline__ float bn_dact(float pre, int act, float slope) {
    if (act == 1) { const float t = tanhf(pre); return 1.f - t * t; }
    return pre > 0.f ? 1.f : slope;
}
__global__ void bn_act_fwd_kernel(const float* __restrict__ x, long total, int D, const float* __restrict__ mean,
                                  const float* __restrict__ var, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float eps, int act, float slope, float* __restrict__ a,
                                  float* __restrict__ xhat) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const float xh = (x[i] - mean[c]) * rsqrtf(var[c] + eps);
        if (xhat) xhat[i] = xh;
        a[i] = bn_act(fmaf(gamma[c], xh, beta[c]), act, slope);
    }
}
// per-block column partials of dpre and dpre * xhat (dpre = da * act'(pre)):  part[0][blk][c], part[1][blk][c]
__global__ void bn_act_bwd_sums_kernel(const float* __restrict__ da, const float* __restrict__ xhat, long rows, int D,
                                       const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                       float slope, float* __restrict__ part) {
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float g = gamma[c], bt = beta[c];
        float sb = 0.f, sg = 0.f;
        for (long r = r0; r < r1; ++r) {
            const float xh = xhat[r * D + c];
            const float dpre = da[r * D + c] * bn_dact(fmaf(g, xh, bt), act, slope);
            sb += dpre;
            sg = fmaf(dpre, xh, sg);
        }
        part[(long)blockIdx.x * D + c] = sb;
        part[((long)gridDim.x + blockIdx.x) * D + c] = sg;
    }
}
// dx = gamma rsqrt(var + eps) / n * (n dpre - sum dpre - xhat sum(dpre xhat)), n = rows of the whole batch
__global__ void bn_act_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ xhat, long total, int D,
                                        const float* __restrict__ var, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float eps, int act, float slope,
                                        const float* __restrict__ dgamma, const float* __restrict__ dbeta, float n,
                                        float* __restrict__ dx) {
    const float inv_n = 1.f / n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const float g = gamma[c], xh = xhat[i];
        const float dpre = da[i] * bn_dact(fmaf(g, xh, beta[c]), act, slope);
        dx[i] = rsqrtf(var[c] + eps) * inv_n * g * (n * dpre - dbeta[c] - xh * dgamma[c]);
    }
}

}  // namespace gx

using namespace gx;

#define GX_PH_BLOCKS 1024
static inline int ph_grid(long total) {
    const long want = (total + 255) / 256;
    return (int)(want < 16384 ? want : 16384);
}

extern "C" {

int gx_gather_rows(const float* Z, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps, float* out,
                   void* stream) {
    if (!Z || !picks || !out || B <= 0 || HW <= 0 || D <= 0 || rows_per_img <= 0) return GX_ERR_BAD_ARG;
    const long rows = (long)B * rows_per_img;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, Z, HW, D, picks, maps,
                       rows_per_img, rows, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_scatter_rows_add(const float* g, int B, int HW, int D, const int* picks, int rows_per_img, const int* maps, float* G,
                        void* stream) {
    if (!g || !picks || !G || B <= 0 || HW <= 0 || D <= 0 || rows_per_img <= 0) return GX_ERR_BAD_ARG;
    const long rows = (long)B * rows_per_img;
    hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, g, HW, D, picks,
                       maps, rows_per_img, rows, G);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_leaky_relu(float* x, long total, float slope, void* stream) {
    if (!x || total <= 0 || !(slope > 0.f)) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(leaky_relu_kernel, dim3(ph_grid(total)), dim3(256), 0, (hipStream_t)stream, x, total, slope);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_leaky_relu_bwd(float* dy, const float* y, long total, float slope, void* stream) {
    if (!dy || !y || total <= 0 || !(slope > 0.f)) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(leaky_relu_bwd_kernel, dim3(ph_grid(total)), dim3(256), 0, (hipStream_t)stream, dy, y, total, slope);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

size_t gx_col_moment_workspace_bytes(int D) { return (size_t)2 * GX_PH_BLOCKS * D * sizeof(float); }
// out[c] = scale * sum_rows (x[r][c] - center[c])^(1 + square)   (center NULL = 0).  BatchNorm1d's batch statistics are
// two calls: the mean (scale 1/n), then the biased variance about it (square, scale 1/n); across ranks each is
// all-reduced (SUM) with n the global row count before the next one runs.
int gx_col_moment(const float* x, long rows, int D, const float* center, int square, float scale, float* out,
                  float* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !out || rows <= 0 || D <= 0) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_col_moment_workspace_bytes(D)) return GX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)(rows < GX_PH_BLOCKS ? rows : GX_PH_BLOCKS);
    hipLaunchKernelGGL(col_moment_kernel, dim3(nb), dim3(256), 0, s, x, rows, D, center, square, workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_parts_finish_kernel, dim3(gx_cdiv(D, 256)), dim3(256), 0, s, workspace, nb, D, scale, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// a = act(gamma xhat + beta); xhat (may be NULL) kept for the backward.  act 0: LeakyReLU(slope), 1: tanh
int gx_bn_act_fwd(const float* x, long rows, int D, const float* mean, const float* var, const float* gamma,
                  const float* beta, float eps, int act, float slope, float* a, float* xhat, void* stream) {
    if (!x || !mean || !var || !gamma || !beta || !a || rows <= 0 || D <= 0 || act < 0 || act > 1) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ph_grid(rows * D)), dim3(256), 0, (hipStream_t)stream, x, rows * D, D, mean,
                       var, gamma, beta, eps, act, slope, a, xhat);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// dgamma[c] = sum dpre xhat, dbeta[c] = sum dpre over this rank's rows (all-reduce them before the apply pass)
int gx_bn_act_bwd_sums(const float* da, const float* xhat, long rows, int D, const float* gamma, const float* beta, int act,
                       float slope, float* dgamma, float* dbeta, float* workspace, size_t workspace_bytes, void* stream) {
    if (!da || !xhat || !gamma || !beta || !dgamma || !dbeta || rows <= 0 || D <= 0 || act < 0 || act > 1)
        return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_col_moment_workspace_bytes(D)) return GX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)(rows < GX_PH_BLOCKS ? rows : GX_PH_BLOCKS);
    hipLaunchKernelGGL(bn_act_bwd_sums_kernel, dim3(nb), dim3(256), 0, s, da, xhat, rows, D, gamma, beta, act, slope,
                       workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_parts_finish_kernel, dim3(gx_cdiv(D, 256)), dim3(256), 0, s, workspace, nb, D, 1.f, dbeta);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_parts_finish_kernel, dim3(gx_cdiv(D, 256)), dim3(256), 0, s, workspace + (size_t)nb * D, nb, D,
                       1.f, dgamma);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// dx from the (global) sums; n_total = rows of the whole batch (all ranks)
int gx_bn_act_bwd_apply(const float* da, const float* xhat, long rows, int D, const float* var, const float* gamma,
                        const float* beta, float eps, int act, float slope, const float* dgamma, const float* dbeta,
                        float n_total, float* dx, void* stream) {
    if (!da || !xhat || !var || !gamma || !beta || !dgamma || !dbeta || !dx || rows <= 0 || D <= 0 || act < 0 || act > 1 ||
        !(n_total >= 1.f))
        return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ph_grid(rows * D)), dim3(256), 0, (hipStream_t)stream, da, xhat,
                       rows * D, D, var, gamma, beta, eps, act, slope, dgamma, dbeta, n_total, dx);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"
