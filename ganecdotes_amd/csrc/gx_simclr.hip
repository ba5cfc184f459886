// SimCLR baseline head (BASELINE.json config 5; baseline/hfc_with_simclr/simclr_clustering.py of
// the reference): BatchNorm1d(train) + LeakyReLU forward/backward over a small sample batch, and
// the reference's inline NT-Xent-style loss with its gradient.  The heavy work of that path (view
// creation, descriptor gather, the two Linear layers) reuses the generator / gather / fp32-MFMA
// GEMM kernels; everything here is tiny and latency-bound by construction (2*batch = 40 samples).
#include "gx_common.h"

namespace gx {

// column statistics over rows: mean[c], and sum_r (x - mean)^2 (two passes -> no cancellation)
__global__ void col_sum_kernel(const float* __restrict__ x, long rows, int D, const float* __restrict__ center,
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
__global__ void col_finish_kernel(const float* __restrict__ part, int nblk, int D, float scale, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[(long)b * D + c];
    out[c] = s * scale;
}

// a = lrelu(gamma * xhat + beta, slope), xhat = (x - mean) * rsqrt(var + eps)
__global__ void bn_lrelu_apply_kernel(const float* __restrict__ x, long total, int D, const float* __restrict__ mean,
                                      const float* __restrict__ var, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float eps, float slope, float* __restrict__ a,
                                      float* __restrict__ xhat) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % D);
        const float xh = (x[i] - mean[c]) * rsqrtf(var[c] + eps);
        const float pre = fmaf(gamma[c], xh, beta[c]);
        if (xhat) xhat[i] = xh;
        a[i] = pre > 0.f ? pre : slope * pre;
    }
}

// backward of the above over n rows (n small): one thread per column
__global__ void bn_lrelu_bwd_kernel(const float* __restrict__ da, const float* __restrict__ xhat, int n, int D,
                                    const float* __restrict__ var, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, float eps, float slope, float* __restrict__ dx,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= D) return;
    const float g = gamma[c], bt = beta[c];
    float sg = 0.f, sb = 0.f;
    for (int r = 0; r < n; ++r) {
        const float xh = xhat[(long)r * D + c];
        const float dpre = da[(long)r * D + c] * (fmaf(g, xh, bt) > 0.f ? 1.f : slope);
        sg = fmaf(dpre, xh, sg);
        sb += dpre;
    }
    dgamma[c] = sg; dbeta[c] = sb;
    const float istd = rsqrtf(var[c] + eps), inv_n = 1.f / (float)n;
    for (int r = 0; r < n; ++r) {
        const float xh = xhat[(long)r * D + c];
        const float dpre = da[(long)r * D + c] * (fmaf(g, xh, bt) > 0.f ? 1.f : slope);
        // dxhat = dpre * g ; dx = istd/n * (n dxhat - sum dxhat - xhat sum(dxhat xhat))
        dx[(long)r * D + c] = istd * inv_n * g * ((float)n * dpre - sb - xh * sg);
    }
}

// The reference's inline loss (simclr_clustering.py:224-264) and d loss / d out.
// out (n, D) = projection of the n = 2*batch interleaved samples; R[i][s] = out[s][i], i < n, are
// the first n FEATURE rows (the reference indexes `scores[i]` of the (D, n) transposed scores).
//   c_ij = <R_i,R_j> / max(|R_i||R_j|, 1e-8); L_ij = -c_ij/T + log sum_{m != i} exp(c_im/T)
//   loss = (1/n) sum_k [ L[2k-1][2k] + L[2k][2k-1] ]   (index -1 wraps to n-1)
// One workgroup; n <= 64.
constexpr int NTX_MAX = 64;
__global__ __launch_bounds__(256) void ntxent_kernel(const float* __restrict__ out, int n, int D, float inv_T,
                                                     float* __restrict__ loss, float* __restrict__ dout) {
    __shared__ float R[NTX_MAX][NTX_MAX + 1], C[NTX_MAX][NTX_MAX + 1], Gm[NTX_MAX][NTX_MAX + 1];
    __shared__ float u[NTX_MAX], den[NTX_MAX], lsum[NTX_MAX];
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += blockDim.x) { const int f = i / n, s = i - f * n; R[f][s] = out[(long)s * D + f]; }
    __syncthreads();
    if (tid < n) { float a = 0.f; for (int s = 0; s < n; ++s) a = fmaf(R[tid][s], R[tid][s], a); u[tid] = sqrtf(a); }
    __syncthreads();
    for (int i = tid; i < n * n; i += blockDim.x) {
        const int a = i / n, b = i - a * n;
        float d = 0.f;
        for (int s = 0; s < n; ++s) d = fmaf(R[a][s], R[b][s], d);
        C[a][b] = d / fmaxf(u[a] * u[b], 1e-8f);
    }
    __syncthreads();
    if (tid < n) {
        float s = 0.f;
        for (int m = 0; m < n; ++m) if (m != tid) s += expf(C[tid][m] * inv_T);
        den[tid] = s;
        // the one pair whose first index is tid: partner = tid+1 (tid odd, wrapping) or tid-1 (tid even, wrapping)
        const int j = (tid & 1) ? (tid + 1) % n : (tid + n - 1) % n;
        lsum[tid] = -C[tid][j] * inv_T + logf(s);
    }
    __syncthreads();
    if (tid == 0) { float t = 0.f; for (int i = 0; i < n; ++i) t += lsum[i]; loss[0] = t / (float)n; }
    if (!dout) return;
    // g_ab = d loss / d c_ab = (1/(nT)) [ -P_ab + (a != b) exp(c_ab/T) / den_a ]
    for (int i = tid; i < n * n; i += blockDim.x) {
        const int a = i / n, b = i - a * n;
        const int j = (a & 1) ? (a + 1) % n : (a + n - 1) % n;
        float g = (a != b) ? expf(C[a][b] * inv_T) / den[a] : 0.f;
        if (b == j) g -= 1.f;
        Gm[a][b] = g * inv_T / (float)n;
    }
    __syncthreads();
    // dR_a = sum_b (g_ab + g_ba) (R_b / (u_a u_b) - c_ab R_a / u_a^2)      (no-clamp branch of the cosine)
    for (int i = tid; i < n * n; i += blockDim.x) {
        const int a = i / n, s = i - a * n;
        float acc = 0.f;
        for (int b = 0; b < n; ++b) {
            const float w = Gm[a][b] + Gm[b][a];
            acc += w * (R[b][s] / (u[a] * u[b]) - C[a][b] * R[a][s] / (u[a] * u[a]));
        }
        dout[(long)s * D + a] = acc;
    }
    for (long i = tid; i < (long)n * (D - n); i += blockDim.x) {          // features >= n do not reach the loss
        const int s = (int)(i / (D - n)), f = n + (int)(i - (long)s * (D - n));
        dout[(long)s * D + f] = 0.f;
    }
}

// z[row][:] /= max(nrm[row], 1e-12)     (F.normalize of the descriptor, folded after the projection)
__global__ void row_div_kernel(float* __restrict__ z, long rows, int D, const float* __restrict__ nrm) {
    const long total = rows * D;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
        z[i] /= fmaxf(nrm[i / D], 1e-12f);
}

}  // namespace gx

using namespace gx;

extern "C" {

#define GX_CS_BLOCKS 256
size_t gx_col_stats_workspace_bytes(int D) { return (size_t)GX_CS_BLOCKS * D * sizeof(float); }
// mean[c] and biased variance var[c] over the rows of x (rows, D); unbiased = var * rows/(rows-1).
int gx_col_stats(const float* x, long rows, int D, float* mean, float* var, float* workspace, size_t workspace_bytes,
                 void* stream) {
    if (!x || !mean || !var || rows <= 0 || D <= 0) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_col_stats_workspace_bytes(D)) return GX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)(rows < GX_CS_BLOCKS ? rows : GX_CS_BLOCKS);
    hipLaunchKernelGGL(col_sum_kernel, dim3(nb), dim3(256), 0, s, x, rows, D, (const float*)nullptr, 0, workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_finish_kernel, dim3(gx_cdiv(D, 256)), dim3(256), 0, s, workspace, nb, D, 1.f / (float)rows, mean);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_sum_kernel, dim3(nb), dim3(256), 0, s, x, rows, D, (const float*)mean, 1, workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(col_finish_kernel, dim3(gx_cdiv(D, 256)), dim3(256), 0, s, workspace, nb, D, 1.f / (float)rows, var);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_bn_lrelu_fwd(const float* x, long rows, int D, const float* mean, const float* var, const float* gamma,
                    const float* beta, float eps, float slope, float* a, float* xhat, void* stream) {
    if (!x || !mean || !var || !gamma || !beta || !a || rows <= 0 || D <= 0) return GX_ERR_BAD_ARG;
    long want = (rows * D + 255) / 256;
    hipLaunchKernelGGL(bn_lrelu_apply_kernel, dim3((int)(want < 4096 ? want : 4096)), dim3(256), 0, (hipStream_t)stream,
                       x, rows * D, D, mean, var, gamma, beta, eps, slope, a, xhat);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_bn_lrelu_bwd(const float* da, const float* xhat, int n, int D, const float* var, const float* gamma,
                    const float* beta, float eps, float slope, float* dx, float* dgamma, float* dbeta, void* stream) {
    if (!da || !xhat || !var || !gamma || !beta || !dx || !dgamma || !dbeta || n <= 0 || D <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(bn_lrelu_bwd_kernel, dim3(gx_cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, da, xhat, n, D, var,
                       gamma, beta, eps, slope, dx, dgamma, dbeta);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
// loss[0] and dout (n, D) (NULL = forward only); n = 2*batch <= 64 <= D.
int gx_simclr_ntxent(const float* out, int n, int D, float temperature, float* loss, float* dout, void* stream) {
    if (!out || !loss || n < 2 || (n & 1) || n > NTX_MAX || D < n || temperature <= 0.f) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(ntxent_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, n, D, 1.f / temperature, loss, dout);
    GX_CHECK_LAUNCH();
    return GX_OK;
}
int gx_row_div(float* z, long rows, int D, const float* nrm, void* stream) {
    if (!z || !nrm || rows <= 0 || D <= 0) return GX_ERR_BAD_ARG;
    long want = (rows * D + 255) / 256;
    hipLaunchKernelGGL(row_div_kernel, dim3((int)(want < 8192 ? want : 8192)), dim3(256), 0, (hipStream_t)stream, z, rows,
                       D, nrm);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"
