// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of ganecdotes_amd.
// Wavefront = 64 lanes everywhere; no CUDA/warp-32 compatibility paths.
#pragma once
// Lab switches (tools/lab/*.sh) — compile-time ablations, some of which compute WRONG values on purpose — exist only in
// lab builds: tools/lab/labbuild.sh passes -DGX_LAB and links a SEPARATE library (libganecdotes_lab.so, loaded only when
// GX_LAB_LIB names it).  The production build (build.py) never defines GX_LAB, so a stray -D from GX_EXTRA_FLAGS stops here.
#if !defined(GX_LAB) && (defined(CE_PIX_ABL) || defined(SKT_ABL) || defined(SKT_FORCE_WIDE) || defined(XP_ABL_A2) || \
                         defined(XP_ABL_B2) || defined(XW_ABL_DIRECT) || defined(FIN_ABL) || defined(FQ_ABL) || \
                         defined(BK_LOAD_AUX) || defined(BK_STORE_AUX) || defined(BK_PASS_MINB) || defined(CE_PIX_NJ2) || \
                         defined(BLUR_MINB) || defined(FQ_LATE) || defined(BD_AHEAD) || defined(BD_ROWS) || defined(FQ_PRIO) || \
                         defined(XH_STAGES) || defined(RAWTAB_BLOCKS) || defined(XW_TALL_NS) || defined(WS_DEPTH) || defined(XP_ABL_A3))
#error "lab ablation macro set without -DGX_LAB: ablated kernels must not reach libganecdotes_hip.so (use tools/lab/labbuild.sh)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
// the public C ABI: every translation unit sees the prototypes, so a definition that drifts from
// include/ganecdotes_hip.h (argument type or order) is a compile error, not silent garbage through ctypes
#include "ganecdotes_hip.h"

// GX_OK / GX_ERR_BAD_ARG (null pointer, non-positive dimension, unsupported mode) / GX_ERR_ALIGN /
// GX_ERR_WORKSPACE / GX_ERR_LAUNCH (hipGetLastError() after launch) come from ganecdotes_hip.h

#define GX_CHECK_LAUNCH()                                  \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return GX_ERR_LAUNCH; \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
namespace gx {
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
}  // namespace gx
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int gx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

namespace gx {
// out[k] = alpha * sum_b part[b][k] + beta * out[k] over nblk partial rows, in a fixed order (gx_swav.hip).  `spare`
// (optional): >= 8 * K floats of scratch; with it many rows of many columns are reduced in two stages that fill the chip.
void launch_partial_reduce(const float* part, int nblk, int K, float alpha, float beta, float* out, hipStream_t stream,
                           float* spare = nullptr);
// the same for nprob problems at once: part [nprob][nblk][K] -> out [nprob][K]
void launch_partial_reduce_batch(const float* part, int nblk, int K, int nprob, float* out, hipStream_t stream);
// (max, sum) pairs of an online log-sum-exp merged exactly per problem: cmp / usp [nprob][per][K] -> cmax / u [nprob][K] (gx_swav.hip)
void launch_first_from_parts(const float* cmp, const float* usp, int per, int nprob, int K, float inv_eps, float* cmax, float* u,
                             hipStream_t stream);
}  // namespace gx

// Cross-lane exchange inside a row of 16 lanes by DPP (one VALU instruction, no LDS round trip, unlike
// ds_bpermute behind __shfl_xor).  For a commutative reduction any pairing that joins two already
// reduced, disjoint groups will do: xor 1 and xor 2 are quad permutes, the 8-lane step pairs lane i
// with 7 - i (row_half_mirror) and the 16-lane step lane i with 15 - i (row_mirror).
template <int CTRL>
__device__ __forceinline__ float gx_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int GX_DPP_XOR1 = 0xB1, GX_DPP_XOR2 = 0x4E, GX_DPP_HALF_MIRROR = 0x141, GX_DPP_MIRROR = 0x140;

__device__ __forceinline__ float gx_wave_sum(float v) {
    v += gx_dpp<GX_DPP_XOR1>(v);
    v += gx_dpp<GX_DPP_XOR2>(v);
    v += gx_dpp<GX_DPP_HALF_MIRROR>(v);
    v += gx_dpp<GX_DPP_MIRROR>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float gx_wave_max(float v) {
    v = fmaxf(v, gx_dpp<GX_DPP_XOR1>(v));
    v = fmaxf(v, gx_dpp<GX_DPP_XOR2>(v));
    v = fmaxf(v, gx_dpp<GX_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, gx_dpp<GX_DPP_MIRROR>(v));
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
// x(lane) + x(lane ^ 32) and x(lane) + x(lane ^ 16) in every lane by v_permlane32_swap / v_permlane16_swap (gfx950: swap the
// odd half-waves / 16-lane rows of the first operand with the even ones of the second, in place): two VALU instructions
// instead of a ds_bpermute through the LDS crossbar.  All 64 lanes must be active.  Inline asm: hipcc 7.2 returns the FIRST
// register for both elements of __builtin_amdgcn_permlane32_swap's result.
__device__ __forceinline__ float gx_xor32_sum(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return a + b;
}
__device__ __forceinline__ float gx_xor16_sum(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return a + b;
}
__device__ __forceinline__ float gx_xor32_max(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float gx_xor16_max(float x) {
    float a = x, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    return fmaxf(a, b);
}
// wave-wide sum / max with no LDS crossbar step (four DPP steps inside the rows of 16 lanes, then the two permlane swaps);
// all 64 lanes active
__device__ __forceinline__ float gx_wave_sum_valu(float v) {
    v += gx_dpp<GX_DPP_XOR1>(v);
    v += gx_dpp<GX_DPP_XOR2>(v);
    v += gx_dpp<GX_DPP_HALF_MIRROR>(v);
    v += gx_dpp<GX_DPP_MIRROR>(v);
    return gx_xor32_sum(gx_xor16_sum(v));
}
__device__ __forceinline__ float gx_wave_max_valu(float v) {
    v = fmaxf(v, gx_dpp<GX_DPP_XOR1>(v));
    v = fmaxf(v, gx_dpp<GX_DPP_XOR2>(v));
    v = fmaxf(v, gx_dpp<GX_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, gx_dpp<GX_DPP_MIRROR>(v));
    return gx_xor32_max(gx_xor16_max(v));
}
// reductions over a power-of-two team of consecutive lanes (team <= 64); every lane gets the result
__device__ __forceinline__ float gx_team_sum(float v, int team) {
    if (team >= 2) v += gx_dpp<GX_DPP_XOR1>(v);
    if (team >= 4) v += gx_dpp<GX_DPP_XOR2>(v);
    if (team >= 8) v += gx_dpp<GX_DPP_HALF_MIRROR>(v);
    if (team >= 16) v += gx_dpp<GX_DPP_MIRROR>(v);
    if (team >= 32) v += __shfl_xor(v, 16, 64);
    if (team >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float gx_team_max(float v, int team) {
    if (team >= 2) v = fmaxf(v, gx_dpp<GX_DPP_XOR1>(v));
    if (team >= 4) v = fmaxf(v, gx_dpp<GX_DPP_XOR2>(v));
    if (team >= 8) v = fmaxf(v, gx_dpp<GX_DPP_HALF_MIRROR>(v));
    if (team >= 16) v = fmaxf(v, gx_dpp<GX_DPP_MIRROR>(v));
    if (team >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (team >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS.
// Every thread gets the total.  Fixed summation order -> bitwise reproducible.
__device__ __forceinline__ float gx_block_sum(float v, float* red) {
    v = gx_wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float gx_block_max(float v, float* red) {
    v = gx_wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}
