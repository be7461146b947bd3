// Split-K finished inside the GEMM kernel: the workgroup that arrives LAST at a tile's counter sums the partial tiles of all
// splits (in split order: the same additions as splitk_reduce_kernel, so the same bits) and applies the epilogue; the
// others store their partial and leave.  Nobody waits for anybody, so there is nothing to time out.
//
// Hand-off between workgroups that may sit on different XCDs (cdna_hip_programming.md, Guideline 16, first valid form):
// partial tiles are stored write-through (`sc1` buffer stores), every storing thread drains (`s_waitcnt vmcnt(0)`), a
// workgroup barrier, ONE lane's agent-scope atomic add; the last arriver reads every partial with `sc1` loads (they do not
// look in this CU's L1).  The counters live in the caller's split-K scratch: zeroed once per step by the entry point
// (SplitkScope::enable_fused), put back to zero by each last arriver, stream-ordered like the scratch itself.
#pragma once
#include <hip/hip_runtime.h>

namespace vita {

typedef float skf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int skf_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t skf_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ skf_f32x4 skf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(skf_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16));
}
__device__ __forceinline__ void skf_store16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const skf_f32x4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(skf_u32x4, v), r, (int)byte_off, 0, 16);
}

// Call with every thread of the workgroup, after its partial stores.  True in the workgroup that completed the tile.
__device__ __forceinline__ bool skf_arrive_last(int* counter, int splits, int* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == splits - 1 ? 1 : 0;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        *lds_flag = last;
    }
    __syncthreads();
    return *lds_flag != 0;
}

// sum_s partial[s] of one float4, s = 0 .. S-1 in index order; loads requested 8 at a time (one round trip per 8 splits)
__device__ __forceinline__ skf_f32x4 skf_sum(__amdgpu_buffer_rsrc_t r, unsigned byte_off, unsigned split_stride_bytes, int S) {
    skf_f32x4 a = skf_load16(r, byte_off);
    for (int s0 = 1; s0 < S; s0 += 8) {
        skf_f32x4 b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < S) b[u] = skf_load16(r, byte_off + (unsigned)(s0 + u) * split_stride_bytes);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < S) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] = a[k] + b[u][k];
            }
    }
    return a;
}

}  // namespace vita
