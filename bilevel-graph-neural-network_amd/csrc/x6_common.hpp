// Shared pieces of the bf16x6 kernels (gin_x6.hip, linear_x6.hip): operand split, packed-weight
// layout, LDS-only barrier.
#pragma once
#include "common.hpp"

namespace blgnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for
// every global load / store in flight -- exactly the prefetches and write-backs these kernels keep
// running across phase boundaries.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Split3 { __bf16 hi, mid, lo; };
__device__ __forceinline__ Split3 split3(float x) {
    Split3 s;
    s.hi = (__bf16)x;
    const float r1 = x - (float)s.hi;
    s.mid = (__bf16)r1;
    s.lo = (__bf16)(r1 - (float)s.mid);
    return s;
}

// Packed weight layout: W[K,N] fp32 -> [3 pieces][Kp/8][N][8] bf16, Kp = K rounded up to 32 (zero
// filled): the 8 consecutive k of one column are one 16-byte word = one MFMA fragment load.
__global__ void k_pack_x6(const float *__restrict__ w, int64_t ldw, int K, int N, int Kp, __bf16 *__restrict__ out);

}  // namespace blgnn
