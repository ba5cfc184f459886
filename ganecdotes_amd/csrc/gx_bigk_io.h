// Row access of the many-prototype row kernels (gx_bigk.h, gx_sktab.hip): a score row through a buffer descriptor,
// 16-byte loads / stores with the hardware range check as the column bound.
#pragma once
#include "gx_common.h"

namespace gx {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

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
#ifndef BK_LOAD_AUX
#define BK_LOAD_AUX 0      // lab: cache-policy bits of the row loads / stores (2 = nt), tools/lab/bigk_aux.sh
#endif
#ifndef BK_STORE_AUX
#define BK_STORE_AUX 0
#endif
__device__ __forceinline__ f32x4 bk_load4(__amdgpu_buffer_rsrc_t r, int first_col) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, first_col * 4, 0, BK_LOAD_AUX));
}
__device__ __forceinline__ void bk_store4(__amdgpu_buffer_rsrc_t r, int first_col, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, first_col * 4, 0, BK_STORE_AUX);
}

}  // namespace gx
