// Row kernels for MANY prototypes (64 < K <= 6144, e.g. the reference's K = 4000 / 5000): one
// wavefront owns a score row, lane l holds columns l + 64 j in registers (NJ = ceil(K/64) rounded
// up), so a Sinkhorn sweep reads the row ONCE from HBM, evaluates ONE exp per score, reduces
// with wavefront shuffles and accumulates the column sums in per-lane registers (no LDS atomics
// in the row loop).  The generic team kernels of gx_swav.hip re-read each row three times and
// pay an LDS atomic per score; at K = 5000, 320 000 rows they ran at 0.4 TB/s.
#pragma once
#include "gx_common.h"

namespace gx {

__device__ __forceinline__ unsigned bk_f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bk_ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

enum { BK_COLMAX = 0, BK_COLSUM = 1, BK_FIRST = 2, BK_SWEEP = 3 };

// MODE BK_COLMAX : cmax[k] = max_n S_nk                     (ordered-uint atomicMax into `cmax_out`)
//      BK_COLSUM : part[b][k] = sum_{rows of block} S_nk
//      BK_FIRST  : part[b][k] = sum exp((S_nk - cmax_k)/eps)
//      BK_SWEEP  : lb_n = log c_n - logsumexp_k(S_nk/eps + la_k); part[b][k] = sum exp(S_nk/eps + la_k + lb_n)
template <int NJ, int MODE>
__global__ __launch_bounds__(256) void bigk_pass_kernel(const float* __restrict__ S, long rows, int K, long ld,
                                                        float inv_eps, const unsigned* __restrict__ cmax,
                                                        const float* __restrict__ la, const float* __restrict__ c,
                                                        float c_uniform, float* __restrict__ lb_out,
                                                        float* __restrict__ part, unsigned* __restrict__ cmax_out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];           // vec[K] | acc[K]
    float* vec = sm;
    float* acc = sm + K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        if (MODE == BK_FIRST) vec[k] = -bk_ord2f(cmax[k]) * inv_eps;
        if (MODE == BK_SWEEP) vec[k] = la[k];
        if (MODE == BK_COLMAX) reinterpret_cast<unsigned*>(acc)[k] = 0u; else acc[k] = 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    float u[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) u[j] = (MODE == BK_COLMAX) ? -INFINITY : 0.f;
    for (long row = wave0; row < rows; row += nwave) {
        const float* p = S + row * ld;
        if (MODE == BK_COLMAX || MODE == BK_COLSUM) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int k = lane + 64 * j;
                if (k < K) u[j] = (MODE == BK_COLMAX) ? fmaxf(u[j], p[k]) : u[j] + p[k];
            }
        } else if (MODE == BK_FIRST) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int k = lane + 64 * j;
                if (k < K) u[j] += expf(fmaf(p[k], inv_eps, vec[k]));
            }
        } else {
            float w[NJ];
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int k = lane + 64 * j;
                w[j] = k < K ? fmaf(p[k], inv_eps, vec[k]) : -INFINITY;
                mx = fmaxf(mx, w[j]);
            }
            mx = gx_wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; ++j) { w[j] = expf(w[j] - mx); sum += w[j]; }      // exp(-inf) = 0 past K
            sum = gx_wave_sum(sum);
            const float cn = c ? c[row] : c_uniform;
            const float f = cn / sum;                       // exp(lb + mx)
#pragma unroll
            for (int j = 0; j < NJ; ++j) u[j] = fmaf(w[j], f, u[j]);
            if (lb_out && lane == 0) lb_out[row] = logf(cn) - mx - logf(sum);
        }
    }
    // combine the block's waves in LDS, then one partial row (or one global atomicMax) per block
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = lane + 64 * j;
        if (k < K) {
            if (MODE == BK_COLMAX) atomicMax(reinterpret_cast<unsigned*>(acc) + k, bk_f2ord(u[j]));
            else atomicAdd(acc + k, u[j]);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        if (MODE == BK_COLMAX) atomicMax(cmax_out + k, reinterpret_cast<unsigned*>(acc)[k]);
        else part[(long)blockIdx.x * K + k] = acc[k];
    }
}

// One direction of the swapped CE for one row: codes q = softmax_k(Sq/eps + laq) (kept in
// registers), prediction p = softmax_k(Sp/T):  returns sum_k q_k log p_k  and writes
// dSp = f (q - p sum_k q_k).
template <int NJ>
__device__ __forceinline__ float ce_direction(const float* __restrict__ pq, const float* __restrict__ pp, int K,
                                              const float* __restrict__ laq, float inv_eps, float inv_T, float f,
                                              float* __restrict__ dP, int lane) {
    float e[NJ];
    float wm = -INFINITY, m = -INFINITY;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = lane + 64 * j;
        e[j] = k < K ? fmaf(pq[k], inv_eps, laq[k]) : -INFINITY;
        wm = fmaxf(wm, e[j]);
        if (k < K) m = fmaxf(m, pp[k]);
    }
    wm = gx_wave_max(wm); m = gx_wave_max(m);
    float v = 0.f, z = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = lane + 64 * j;
        e[j] = expf(e[j] - wm);                            // exp(-inf) = 0 past K
        v += e[j];
        if (k < K) z += expf((pp[k] - m) * inv_T);
    }
    v = gx_wave_sum(v); z = gx_wave_sum(z);
    const float iv = 1.f / v, iz = 1.f / z, lz = logf(z);
    float qsum = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { e[j] *= iv; qsum += e[j]; }
    qsum = gx_wave_sum(qsum);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = lane + 64 * j;
        if (k < K) {
            const float x = (pp[k] - m) * inv_T;
            l = fmaf(e[j], x - lz, l);
            if (dP) dP[k] = f * (e[j] - expf(x) * iz * qsum);
        }
    }
    return l;
}

// Swapped-prediction CE forward + d(scores) for many prototypes (same maths as swapped_ce_kernel);
// the two directions run one after the other so only one row of codes lives in registers.
template <int NJ>
__global__ __launch_bounds__(256) void swapped_ce_bigk_kernel(const float* __restrict__ Ss, const float* __restrict__ St,
                                                              long rows, int K, long ld, float inv_eps, float inv_T,
                                                              const float* __restrict__ la_s,
                                                              const float* __restrict__ la_t, float gscale,
                                                              float* __restrict__ dSs, float* __restrict__ dSt,
                                                              long ldd, float* __restrict__ loss_part) {
    extern __shared__ __attribute__((aligned(16))) float sm[];           // las[K] | lat[K]
    float* las = sm;
    float* lat = sm + K;
    for (int k = threadIdx.x; k < K; k += blockDim.x) { las[k] = la_s[k]; lat[k] = la_t[k]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long wave0 = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nwave = (long)gridDim.x * (blockDim.x >> 6);
    const float f = -0.5f * gscale * inv_T;
    float lacc = 0.f;
    for (long row = wave0; row < rows; row += nwave) {
        const float* ps = Ss + row * ld;
        const float* pt = St + row * ld;
        lacc += ce_direction<NJ>(ps, pt, K, las, inv_eps, inv_T, f, dSt ? dSt + row * ldd : nullptr, lane);
        lacc += ce_direction<NJ>(pt, ps, K, lat, inv_eps, inv_T, f, dSs ? dSs + row * ldd : nullptr, lane);
    }
    __shared__ float red[16];
    const float tot = gx_block_sum(lacc, red);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = -0.5f * tot;
}

static inline int bigk_nj(int K) {          // registers per lane for a row of K scores; 0 = not supported
    const int need = (K + 63) / 64;
    const int opts[] = {16, 32, 48, 64, 80, 96};
    for (int o : opts)
        if (need <= o) return o;
    return 0;
}
static inline bool bigk_ok(int K) { return K > 64 && bigk_nj(K) != 0; }

#define GX_BIGK_DISPATCH(NJV, CALL)            \
    switch (NJV) {                             \
        case 16: { constexpr int NJ = 16; CALL; } break; \
        case 32: { constexpr int NJ = 32; CALL; } break; \
        case 48: { constexpr int NJ = 48; CALL; } break; \
        case 64: { constexpr int NJ = 64; CALL; } break; \
        case 80: { constexpr int NJ = 80; CALL; } break; \
        default: { constexpr int NJ = 96; CALL; } break; \
    }

static inline int bigk_blocks(long rows, int cap) {
    long nb = (rows + 3) / 4;
    return (int)(nb < cap ? nb : cap);
}

}  // namespace gx
