// hfc_kmeans baseline on gfx950 (SURVEY.md par. 8f rank 4): the per-layer k-means of
// baseline/hfc_kmeans/hfc_kmeans_clustering.py:127-209 (FlatKMeansHFC -> sklearn.cluster.KMeans:
// centring, greedy k-means++, Lloyd iterations with the |c|^2 - 2 x.c E step, label maps).
// The two contractions of an iteration (scores X.C^T and the per-cluster sums onehot^T.X) run on the
// gx_gemm_f32 engine; everything here is the HBM-bound glue around them: one pass over the panel or over
// the (rows x K) score matrix per kernel, fixed-order reductions (no float atomics) so a fit is reproducible.
#include "gx_common.h"

namespace gx {

// ---------------------------------------------------------------------------------------------
// panel: X[n] = [tap1[n] | tap2[n]] - mean, xsq[n] = |X[n]|^2.  One wave per row, 4 rows per block.
__global__ __launch_bounds__(256) void km_center_panel_kernel(const float* __restrict__ t1, int C1,
                                                              const float* __restrict__ t2, int C2, long rows,
                                                              const float* __restrict__ mean, float* __restrict__ X,
                                                              float* __restrict__ xsq) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C = C1 + C2;
    float acc = 0.f;
    float* xr = X + row * C;
    const float* a = t1 + row * C1;
    for (int c = lane; c < C1; c += 64) {
        const float v = a[c] - (mean ? mean[c] : 0.f);
        xr[c] = v;
        acc += v * v;
    }
    if (t2) {
        const float* b = t2 + row * C2;
        for (int c = lane; c < C2; c += 64) {
            const float v = b[c] - (mean ? mean[C1 + c] : 0.f);
            xr[C1 + c] = v;
            acc += v * v;
        }
    }
    acc = gx_wave_sum(acc);
    if (lane == 0 && xsq) xsq[row] = acc;
}

// ---------------------------------------------------------------------------------------------
// E step on the score matrix S = X.C^T: label = argmin_k (csq[k] - 2 S[n][k]), first minimum wins
// (_update_chunk_dense of sklearn's _k_means_lloyd.pyx).  16 lanes per row, 16 rows per block and pass.
// hist (LDS) collects the block's label histogram; one integer atomic per (block, cluster).
constexpr int KM_HIST = 2048;
__global__ __launch_bounds__(256) void km_assign_kernel(const float* __restrict__ S, long rows, int K, long ld,
                                                        const float* __restrict__ csq,
                                                        const int* __restrict__ prev, int* __restrict__ labels,
                                                        float* __restrict__ onehot, long ldo,
                                                        int* __restrict__ counts, int* __restrict__ nchanged) {
    __shared__ int hist[KM_HIST];
    __shared__ int changed;
    const bool use_lds = counts && K <= KM_HIST;
    if (use_lds)
        for (int k = threadIdx.x; k < K; k += 256) hist[k] = 0;
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    const int tl = threadIdx.x & 15;
    for (long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4); row < rows; row += (long)gridDim.x * 16) {
        const float* s = S + row * ld;
        float bv = __builtin_inff();
        int bi = 0x7fffffff;
        for (int k = tl; k < K; k += 16) {
            const float v = csq[k] - 2.f * s[k];
            if (v < bv) { bv = v; bi = k; }
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            const float ov = __shfl_xor(bv, m, 64);
            const int oi = __shfl_xor(bi, m, 64);
            if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (bi == 0x7fffffff) bi = 0;                 // a row of NaNs: sklearn's loop keeps label 0
        if (onehot)
            for (int k = tl; k < K; k += 16) onehot[row * ldo + k] = (k == bi) ? 1.f : 0.f;
        if (tl == 0) {
            labels[row] = bi;
            if (counts) {
                if (use_lds) atomicAdd(&hist[bi], 1);
                else atomicAdd(&counts[bi], 1);
            }
            if (nchanged && (!prev || prev[row] != bi)) atomicAdd(&changed, 1);
        }
    }
    __syncthreads();
    if (use_lds)
        for (int k = threadIdx.x; k < K; k += 256)
            if (hist[k]) atomicAdd(&counts[k], hist[k]);
    if (threadIdx.x == 0 && nchanged && changed) atomicAdd(nchanged, changed);
}

// ---------------------------------------------------------------------------------------------
// M step sums (the `centers_new[label] += X[i]` loop of _update_chunk_dense): every block owns a chunk of
// rows and a slab of W = blockDim.x columns, keeps K x W accumulators in LDS (thread t only ever touches
// column t, so no barrier and no atomics are needed; a row's label is wave-uniform) and writes its partial
// sums; km_reduce_sums_kernel adds the chunks in order.  X is read once.
__global__ void km_accumulate_kernel(const float* __restrict__ X, long rows, int C, long ldx,
                                     const int* __restrict__ labels, int K, int rows_per_block,
                                     float* __restrict__ part) {
    extern __shared__ float acc[];                       // [K][W]
    const int W = blockDim.x, t = threadIdx.x;
    const int col = blockIdx.y * W + t;
    for (int k = 0; k < K; ++k) acc[k * W + t] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const bool live = col < C;
    const float* x = X + (live ? col : 0);
    long r = r0;
    constexpr int U = 16;                                // rows in flight per thread: 16 x 4 B x 256 threads = 16 KB per block
    for (; r + U <= r1; r += U) {
        int l[U];
        float v[U];
#pragma unroll
        for (int i = 0; i < U; ++i) { l[i] = labels[r + i]; v[i] = x[(r + i) * ldx]; }
#pragma unroll
        for (int i = 0; i < U; ++i) acc[l[i] * W + t] += v[i];
    }
    for (; r < r1; ++r) acc[labels[r] * W + t] += x[r * ldx];
    if (live)
        for (int k = 0; k < K; ++k) part[((long)blockIdx.x * K + k) * C + col] = acc[k * W + t];
}
__global__ __launch_bounds__(64) void km_reduce_sums_kernel(const float* __restrict__ part, int nchunk, long kc,
                                                            float* __restrict__ sums) {
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= kc) return;
    float a = 0.f;
    int c = 0;
    for (; c + 8 <= nchunk; c += 8) {                    // eight loads in flight, summed in chunk order
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = part[(long)(c + j) * kc + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; c < nchunk; ++c) a += part[(long)c * kc + i];
    sums[i] = a;
}

// ---------------------------------------------------------------------------------------------
// M step tail (_average_centers + _center_shift): new centre k = sums[k] * (1 / count[k]);
// shift2[k] = |new - old|^2, csq[k] = |new|^2.  One block per centre.  A cluster with count 0 (only left when
// relocation was pointless: fewer distinct samples than clusters) follows sklearn's in-place, index-ordered loop:
// it copies the row of the biggest cluster am = argmax(counts) as it is at that moment — the raw sum when
// am > k, the mean when am < k.
__global__ __launch_bounds__(256) void km_update_kernel(const float* __restrict__ sums, const int* __restrict__ counts,
                                                        int K, int C, const float* __restrict__ centers_old,
                                                        float* __restrict__ centers, float* __restrict__ csq,
                                                        float* __restrict__ shift2) {
    __shared__ float red[16];
    const int k = blockIdx.x;
    int src = k;
    float alpha = 1.0f;
    const int cnt = counts[k];
    if (cnt > 0) alpha = 1.0f / (float)cnt;
    else {
        int am = 0;
        for (int j = 1; j < K; ++j) if (counts[j] > counts[am]) am = j;
        src = am;
        alpha = am < k ? 1.0f / (float)counts[am] : 1.0f;
    }
    float sh = 0.f, sq = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float old = centers_old[(long)k * C + c];
        const float nw = sums[(long)src * C + c] * alpha;
        centers[(long)k * C + c] = nw;
        sh += (nw - old) * (nw - old);
        sq += nw * nw;
    }
    sh = gx_block_sum(sh, red);
    sq = gx_block_sum(sq, red);
    if (threadIdx.x == 0) {
        shift2[k] = sh;
        csq[k] = sq;
    }
}

// ---------------------------------------------------------------------------------------------
// dist[n] = |X[n] - centers[labels[n]]|^2 by direct differences (_inertia_dense; also the distances of
// _relocate_empty_clusters_dense).  One wave per row.
__global__ __launch_bounds__(256) void km_row_dist_kernel(const float* __restrict__ X, long rows, int C, long ldx,
                                                          const float* __restrict__ centers,
                                                          const int* __restrict__ labels, float* __restrict__ dist) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = X + row * ldx;
    const float* c = centers + (long)labels[row] * C;
    float acc = 0.f;
    for (int i = lane; i < C; i += 64) {
        const float d = x[i] - c[i];
        acc += d * d;
    }
    acc = gx_wave_sum(acc);
    if (lane == 0) dist[row] = acc;
}

// sum of n floats in double, one block, fixed order
__global__ __launch_bounds__(1024) void km_sum_f64_kernel(const float* __restrict__ v, long n, double* __restrict__ out) {
    __shared__ double part[1024];
    const long per = (n + 1023) / 1024;
    const long lo = (long)threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    double a = 0.0;
    for (long i = lo; i < hi; ++i) a += (double)v[i];
    part[threadIdx.x] = a;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

// ---------------------------------------------------------------------------------------------
// greedy k-means++ (sklearn _kmeans_plusplus).  G = X.X[cand]^T (rows x T); squared distance of sample n to
// candidate t:  D = max(xsq[n] + xsq[cand[t]] - 2 G[n][t], 0).
__device__ __forceinline__ float km_pp_dist(const float* G, long ld, const float* xsq, const int* cand, long n, int t) {
    return fmaxf(xsq[n] + xsq[cand[t]] - 2.f * G[n * ld + t], 0.f);
}
constexpr int KM_PP_BLOCKS = 128;
// part[t][block] = sum over the block's rows of min(closest[n], D[t][n])  (closest == null: D itself)
__global__ __launch_bounds__(256) void km_pp_pots_kernel(const float* __restrict__ G, long rows, int T, long ld,
                                                         const float* __restrict__ xsq, const int* __restrict__ cand,
                                                         const float* __restrict__ closest, double* __restrict__ part) {
    __shared__ double red[256];
    const int t = blockIdx.y;
    const long per = (rows + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    double a = 0.0;
    for (long n = lo + threadIdx.x; n < hi; n += 256) {
        float d = km_pp_dist(G, ld, xsq, cand, n, t);
        if (closest) d = fminf(d, closest[n]);
        a += (double)d;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(long)t * gridDim.x + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(64) void km_pp_pots_finish_kernel(const double* __restrict__ part, int nblk,
                                                               double* __restrict__ pots) {
    const int t = blockIdx.x, lane = threadIdx.x;
    double a = 0.0;
    for (int b = lane; b < nblk; b += 64) a += part[(long)t * nblk + b];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) pots[t] = a;
}
__global__ __launch_bounds__(256) void km_pp_commit_kernel(const float* __restrict__ G, long rows, long ld,
                                                           const float* __restrict__ xsq, const int* __restrict__ cand,
                                                           int best, int first, float* __restrict__ closest) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= rows) return;
    const float d = km_pp_dist(G, ld, xsq, cand, n, best);
    closest[n] = first ? d : fminf(closest[n], d);
}
// cand[t] = min(searchsorted(cumsum_f64(closest), target[t], side='left'), rows - 1).  The rows are cut into
// 1024 contiguous chunks: km_pp_chunk_sums_kernel sums them (a wave per chunk, coalesced), then one block scans
// the chunk sums in LDS and the thread owning the chunk a target falls into walks it.
constexpr int KM_PP_MAXT = 16;
struct KmTargets { double v[KM_PP_MAXT]; };
__global__ __launch_bounds__(256) void km_pp_chunk_sums_kernel(const float* __restrict__ closest, long rows,
                                                               double* __restrict__ csum) {
    const long per = (rows + 1023) / 1024;
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * 4 + (threadIdx.x >> 6);
    const long c0 = (long)chunk * per;
    const long c1 = c0 + per < rows ? c0 + per : rows;
    double a = 0.0;
    for (long i = c0 + lane; i < c1; i += 64) a += (double)closest[i];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) a += __shfl_xor(a, m, 64);
    if (lane == 0) csum[chunk] = a;
}
__global__ __launch_bounds__(1024) void km_pp_sample_kernel(const float* __restrict__ closest, long rows,
                                                            const double* __restrict__ csum, KmTargets tg, int T,
                                                            int* __restrict__ cand) {
    __shared__ double incl[1024];
    const long per = (rows + 1023) / 1024;
    const long lo = (long)threadIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    incl[threadIdx.x] = csum[threadIdx.x];
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // inclusive scan of the 1024 chunk sums
        const double add = (int)threadIdx.x >= off ? incl[threadIdx.x - off] : 0.0;
        __syncthreads();
        incl[threadIdx.x] += add;
        __syncthreads();
    }
    const double before = threadIdx.x ? incl[threadIdx.x - 1] : 0.0;
    const double after = incl[threadIdx.x];
    for (int t = 0; t < T; ++t) {
        const double target = tg.v[t];
        if (threadIdx.x == 1023 && target > after) cand[t] = (int)(rows - 1);      // np.clip
        const bool mine = lo < hi && target <= after && (threadIdx.x == 0 || target > before);
        if (!mine) continue;
        double run = before;
        long i = lo;
        for (; i < hi; ++i) {
            run += (double)closest[i];
            if (run >= target) break;
        }
        cand[t] = (int)(i < hi ? i : hi - 1);
    }
}

// ---------------------------------------------------------------------------------------------
// label maps of FlatKMeansHFC.predict (:187-208) for all layers at once, NHWC:
//   out[b][y][x][off_l + k] = (labels_l[b][sy][sx] == k) ? hi : lo,  sy = floor(y * r_l / out) (torch 'nearest')
// with (hi, lo) = (1, -1): the `* 2 - 1` of HFCPreprocessor.predict_hfc_vectors (segmentor.py:230).
constexpr int KM_MAXL = 16;
struct KmMaps {
    const int* labels[KM_MAXL];
    int res[KM_MAXL], K[KM_MAXL], off[KM_MAXL];
    int nl;
};
__global__ __launch_bounds__(256) void km_label_maps_kernel(KmMaps m, int B, int out_size, int ctot, float hi, float lo,
                                                            float* __restrict__ out) {
    const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pix >= (long)B * out_size * out_size) return;
    const int x = (int)(pix % out_size), y = (int)((pix / out_size) % out_size), b = (int)(pix / ((long)out_size * out_size));
    float* o = out + pix * ctot;
    for (int l = 0; l < m.nl; ++l) {
        const int r = m.res[l];
        const float scale = (float)r / (float)out_size;
        int sy = (int)floorf((float)y * scale), sx = (int)floorf((float)x * scale);
        sy = sy < r - 1 ? sy : r - 1;
        sx = sx < r - 1 ? sx : r - 1;
        const int lab = m.labels[l][((long)b * r + sy) * r + sx];
        for (int k = lane; k < m.K[l]; k += 64) o[m.off[l] + k] = (k == lab) ? hi : lo;
    }
}

}  // namespace gx

extern "C" {

int gx_kmeans_center_panel(const float* tap1, int C1, const float* tap2, int C2, long rows, const float* mean, float* X,
                           float* xsq, void* stream) {
    if (!tap1 || C1 <= 0 || C2 < 0 || (C2 > 0 && !tap2) || rows <= 0 || !X) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(gx::km_center_panel_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, tap1, C1,
                       C2 > 0 ? tap2 : (const float*)nullptr, C2, rows, mean, X, xsq);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_kmeans_assign(const float* S, long rows, int K, long ld, const float* csq, const int* labels_prev, int* labels,
                     float* onehot, long ld_onehot, int* counts, int* nchanged, void* stream) {
    if (!S || !csq || !labels || rows <= 0 || K <= 0 || ld < K || (onehot && ld_onehot < K)) return GX_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (counts && hipMemsetAsync(counts, 0, sizeof(int) * K, s) != hipSuccess) return GX_ERR_LAUNCH;
    if (nchanged && hipMemsetAsync(nchanged, 0, sizeof(int), s) != hipSuccess) return GX_ERR_LAUNCH;
    // at most 512 blocks: the label histogram costs one global atomic per (block, cluster)
    const int nblk = (int)(rows < 512L * 16 ? gx_cdiv(rows, 16) : 512);
    hipLaunchKernelGGL(gx::km_assign_kernel, dim3(nblk), dim3(256), 0, s, S, rows, K, ld, csq, labels_prev,
                       labels, onehot, ld_onehot, counts, nchanged);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

// slab width (threads per block) for K clusters so that K x W floats fit 64 KB of LDS; 0: K too large
static int km_acc_width(int K) { return K <= 64 ? 256 : K <= 128 ? 128 : K <= 256 ? 64 : 0; }
static void km_acc_geometry(long rows, int K, int C, int* W, int* nslab, int* rpb, int* nchunk) {
    *W = km_acc_width(K);
    *nslab = *W ? gx_cdiv(C, *W) : 0;
    int want = *nslab ? 512 / *nslab : 1;                 // about two blocks per CU in total
    if (want < 1) want = 1;
    long per = (rows + want - 1) / want;
    if (per < 64) per = 64;
    per = (per + 15) & ~15L;
    *rpb = (int)per;
    *nchunk = gx_cdiv(rows, per);
}
int gx_kmeans_accumulate_supported(int K) { return km_acc_width(K) != 0; }
size_t gx_kmeans_accumulate_workspace_bytes(long rows, int K, int C) {
    int W, nslab, rpb, nchunk;
    km_acc_geometry(rows, K, C, &W, &nslab, &rpb, &nchunk);
    return W ? (size_t)nchunk * K * C * sizeof(float) : 0;
}
int gx_kmeans_accumulate(const float* X, long rows, int C, long ldx, const int* labels, int K, float* sums,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!X || !labels || !sums || rows <= 0 || C <= 0 || K <= 0 || ldx < C) return GX_ERR_BAD_ARG;
    int W, nslab, rpb, nchunk;
    km_acc_geometry(rows, K, C, &W, &nslab, &rpb, &nchunk);
    if (!W) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_kmeans_accumulate_workspace_bytes(rows, K, C)) return GX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gx::km_accumulate_kernel, dim3(nchunk, nslab), dim3(W), (size_t)K * W * sizeof(float), s, X, rows, C,
                       ldx, labels, K, rpb, (float*)workspace);
    GX_CHECK_LAUNCH();
    const long kc = (long)K * C;
    hipLaunchKernelGGL(gx::km_reduce_sums_kernel, dim3(gx_cdiv(kc, 64)), dim3(64), 0, s, (const float*)workspace, nchunk, kc,
                       sums);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_kmeans_update(const float* sums, const int* counts, int K, int C, const float* centers_old, float* centers_new,
                     float* csq, float* shift2, void* stream) {
    if (!sums || !counts || !centers_old || !centers_new || !csq || !shift2 || K <= 0 || C <= 0) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(gx::km_update_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, sums, counts, K, C, centers_old,
                       centers_new, csq, shift2);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_kmeans_row_dist(const float* X, long rows, int C, long ldx, const float* centers, int K, const int* labels,
                       float* dist, double* total, void* stream) {
    if (!X || !centers || !labels || !dist || rows <= 0 || C <= 0 || K <= 0 || ldx < C) return GX_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gx::km_row_dist_kernel, dim3(gx_cdiv(rows, 4)), dim3(256), 0, s, X, rows, C, ldx, centers, labels, dist);
    GX_CHECK_LAUNCH();
    if (total) {
        hipLaunchKernelGGL(gx::km_sum_f64_kernel, dim3(1), dim3(1024), 0, s, (const float*)dist, rows, total);
        GX_CHECK_LAUNCH();
    }
    return GX_OK;
}

size_t gx_kmeans_pp_workspace_bytes(int T) {
    const size_t n = (size_t)T * gx::KM_PP_BLOCKS;          // pots partials; sample: 1024 chunk sums
    return (n > 1024 ? n : 1024) * sizeof(double);
}

int gx_kmeans_pp_pots(const float* G, long rows, int T, long ld, const float* xsq, const int* cand, const float* closest,
                      double* pots, void* workspace, size_t workspace_bytes, void* stream) {
    if (!G || !xsq || !cand || !pots || rows <= 0 || T <= 0 || T > gx::KM_PP_MAXT || ld < T) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < gx_kmeans_pp_workspace_bytes(T)) return GX_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (int)(rows < (long)gx::KM_PP_BLOCKS * 256 ? gx_cdiv(rows, 256) : gx::KM_PP_BLOCKS);
    hipLaunchKernelGGL(gx::km_pp_pots_kernel, dim3(nblk, T), dim3(256), 0, s, G, rows, T, ld, xsq, cand, closest,
                       (double*)workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(gx::km_pp_pots_finish_kernel, dim3(T), dim3(64), 0, s, (const double*)workspace, nblk, pots);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_kmeans_pp_commit(const float* G, long rows, int T, long ld, const float* xsq, const int* cand, int best, int first,
                        float* closest, void* stream) {
    if (!G || !xsq || !cand || !closest || rows <= 0 || T <= 0 || best < 0 || best >= T || ld < T) return GX_ERR_BAD_ARG;
    hipLaunchKernelGGL(gx::km_pp_commit_kernel, dim3(gx_cdiv(rows, 256)), dim3(256), 0, (hipStream_t)stream, G, rows, ld, xsq,
                       cand, best, first, closest);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_kmeans_pp_sample(const float* closest, long rows, const double* targets /*host, T values*/, int T, int* cand,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!closest || !targets || !cand || rows <= 0 || T <= 0 || T > gx::KM_PP_MAXT) return GX_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < 1024 * sizeof(double)) return GX_ERR_WORKSPACE;
    gx::KmTargets tg;
    for (int t = 0; t < gx::KM_PP_MAXT; ++t) tg.v[t] = t < T ? targets[t] : 0.0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gx::km_pp_chunk_sums_kernel, dim3(256), dim3(256), 0, s, closest, rows, (double*)workspace);
    GX_CHECK_LAUNCH();
    hipLaunchKernelGGL(gx::km_pp_sample_kernel, dim3(1), dim3(1024), 0, s, closest, rows, (const double*)workspace, tg, T, cand);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

int gx_kmeans_label_maps(int nlayers, const int* const* labels /*host array of device ptrs*/, const int* res /*host*/,
                         const int* K /*host*/, int B, int out_size, float hi, float lo, float* out, void* stream) {
    if (nlayers <= 0 || nlayers > gx::KM_MAXL || !labels || !res || !K || B <= 0 || out_size <= 0 || !out)
        return GX_ERR_BAD_ARG;
    gx::KmMaps m;
    int off = 0;
    for (int l = 0; l < nlayers; ++l) {
        if (!labels[l] || res[l] <= 0 || K[l] <= 0) return GX_ERR_BAD_ARG;
        m.labels[l] = labels[l]; m.res[l] = res[l]; m.K[l] = K[l]; m.off[l] = off;
        off += K[l];
    }
    m.nl = nlayers;
    const long pix = (long)B * out_size * out_size;
    hipLaunchKernelGGL(gx::km_label_maps_kernel, dim3(gx_cdiv(pix, 4)), dim3(256), 0, (hipStream_t)stream, m, B, out_size,
                       off, hi, lo, out);
    GX_CHECK_LAUNCH();
    return GX_OK;
}

}  // extern "C"
