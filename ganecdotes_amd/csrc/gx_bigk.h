// Row kernels for MANY prototypes (64 < K <= 6144, e.g. the reference's K = 4000 / 5000): one
// wavefront owns a score row, lane l holds columns l + 64 j in registers (NJ = ceil(K/64) rounded
// up), so a Sinkhorn sweep reads the row ONCE from HBM, evaluates ONE exp per score, reduces
// with wavefront shuffles and accumulates the column sums in per-lane registers (no LDS atomics
// in the row loop).  The generic team kernels of gx_swav.hip re-read each row three times and
// pay an LDS atomic per score; at K = 5000, 320 000 rows they ran at 0.4 TB/s.
#pragma once
#include "gx_common.h"

namespace gx {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bk_f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bk_ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

enum { BK_COLMAX = 0, BK_COLSUM = 1, BK_FIRST = 2, BK_SWEEP = 3 };

// A score row through a buffer descriptor: columns past K read as 0 with no branch (hipcc turns
// `k < K ? p[k] : x` into one branch + s_waitcnt vmcnt(0) PER LOAD, i.e. 80 serial round trips per
// row).  `row` is wave-uniform by construction; readfirstlane makes that provable so the
// descriptor stays in SGPRs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bk_row_rsrc(const float* base, long row, long ld, int K) {
    const long off = row * ld;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(off & 0xffffffffL));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long)off >> 32));
    const float* p = base + (long)(((unsigned long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, K * 4, 0x00020000);
}
__device__ __forceinline__ f32x4 bk_load4(__amdgpu_buffer_rsrc_t r, int first_col) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, first_col * 4, 0, 0));
}
__device__ __forceinline__ void bk_store4(__amdgpu_buffer_rsrc_t r, int first_col, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, first_col * 4, 0, 0);
}

// Lane l owns columns 4*(l + 64*c) + q, q < 4, c < NJ/4 (one float4 per chunk: a wave-instruction
// covers 1 KiB of the row).  Needs K % 4 == 0, ld % 4 == 0, 16-B aligned base.
#define BK_COL(lane, c) (4 * ((lane) + 64 * (c)))

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
    constexpr int NC = NJ / 4;
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
    f32x4 u[NC];
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) {
        const float init = (MODE == BK_COLMAX) ? -INFINITY : 0.f;
        u[cidx] = f32x4{init, init, init, init};
    }
    for (long row = wave0; row < rows; row += nwave) {
        const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(S, row, ld, K);
        f32x4 w[NC];
#pragma unroll
        for (int cidx = 0; cidx < NC; ++cidx) w[cidx] = bk_load4(rs, BK_COL(lane, cidx));
        if (MODE == BK_COLMAX) {
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx)
                if (BK_COL(lane, cidx) < K)
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[cidx][q] = fmaxf(u[cidx][q], w[cidx][q]);
        } else if (MODE == BK_COLSUM) {
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) u[cidx] += w[cidx];                 // columns past K read 0
        } else if (MODE == BK_FIRST) {
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const int k = BK_COL(lane, cidx);
                if (k < K) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(vec + k);
#pragma unroll
                    for (int q = 0; q < 4; ++q) u[cidx][q] += expf(fmaf(w[cidx][q], inv_eps, v[q]));
                }
            }
        } else {
            float mx = -INFINITY;
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const int k = BK_COL(lane, cidx);
                const bool ok = k < K;
                const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(vec + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    w[cidx][q] = ok ? fmaf(w[cidx][q], inv_eps, v[q]) : -INFINITY;
                    mx = fmaxf(mx, w[cidx][q]);
                }
            }
            mx = gx_wave_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx)
#pragma unroll
                for (int q = 0; q < 4; ++q) { w[cidx][q] = expf(w[cidx][q] - mx); sum += w[cidx][q]; }   // exp(-inf) = 0
            sum = gx_wave_sum(sum);
            const float cn = c ? c[row] : c_uniform;
            const float f = cn / sum;                       // exp(lb + mx)
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) u[cidx] += w[cidx] * f;
            if (lb_out && lane == 0) lb_out[row] = logf(cn) - mx - logf(sum);
        }
    }
    // combine the block's waves in LDS, then one partial row (or one global atomicMax) per block
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) {
        const int k = BK_COL(lane, cidx);
        if (k < K)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == BK_COLMAX) atomicMax(reinterpret_cast<unsigned*>(acc) + k + q, bk_f2ord(u[cidx][q]));
                else atomicAdd(acc + k + q, u[cidx][q]);
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
__device__ __forceinline__ float ce_direction(__amdgpu_buffer_rsrc_t rq, __amdgpu_buffer_rsrc_t rp, int K,
                                              const float* __restrict__ laq, float inv_eps, float inv_T, float f,
                                              bool write, __amdgpu_buffer_rsrc_t rd, int lane) {
    constexpr int NC = NJ / 4;
    f32x4 e[NC];
    float wm = -INFINITY, m = -INFINITY;
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) {
        const int k = BK_COL(lane, cidx);
        const bool ok = k < K;
        const f32x4 sq = bk_load4(rq, k), sp = bk_load4(rp, k);
        const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(laq + k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e[cidx][q] = ok ? fmaf(sq[q], inv_eps, v[q]) : -INFINITY;
            wm = fmaxf(wm, e[cidx][q]);
            m = fmaxf(m, ok ? sp[q] : -INFINITY);
        }
    }
    wm = gx_wave_max(wm); m = gx_wave_max(m);
    float v = 0.f, z = 0.f;
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) {
        const int k = BK_COL(lane, cidx);
        const f32x4 sp = bk_load4(rp, k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e[cidx][q] = expf(e[cidx][q] - wm);            // exp(-inf) = 0 past K
            v += e[cidx][q];
            z += (k < K) ? expf((sp[q] - m) * inv_T) : 0.f;
        }
    }
    v = gx_wave_sum(v); z = gx_wave_sum(z);
    const float iv = 1.f / v, iz = 1.f / z, lz = logf(z);
    float qsum = 0.f;
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx)
#pragma unroll
        for (int q = 0; q < 4; ++q) { e[cidx][q] *= iv; qsum += e[cidx][q]; }
    qsum = gx_wave_sum(qsum);
    float l = 0.f;
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) {
        const int k = BK_COL(lane, cidx);
        const f32x4 sp = bk_load4(rp, k);
        f32x4 d;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x = (sp[q] - m) * inv_T;
            l += (k < K) ? e[cidx][q] * (x - lz) : 0.f;
            d[q] = f * (e[cidx][q] - expf(x) * iz * qsum);
        }
        if (write) bk_store4(rd, k, d);                    // columns past K are dropped by the range check
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
    const bool write = dSs != nullptr;
    float lacc = 0.f;
    for (long row = wave0; row < rows; row += nwave) {
        const __amdgpu_buffer_rsrc_t rs = bk_row_rsrc(Ss, row, ld, K), rt = bk_row_rsrc(St, row, ld, K);
        const __amdgpu_buffer_rsrc_t ds = bk_row_rsrc(write ? dSs : Ss, row, write ? ldd : ld, K);
        const __amdgpu_buffer_rsrc_t dt = bk_row_rsrc(write ? dSt : St, row, write ? ldd : ld, K);
        lacc += ce_direction<NJ>(rs, rt, K, las, inv_eps, inv_T, f, write, dt, lane);
        lacc += ce_direction<NJ>(rt, rs, K, lat, inv_eps, inv_T, f, write, ds, lane);
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
static inline bool bigk_ok(int K) { return K > 64 && (K % 4) == 0 && bigk_nj(K) != 0; }
static inline bool bigk_aligned(const void* p, long ld) { return (((uintptr_t)p) & 15) == 0 && (ld % 4) == 0; }

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
