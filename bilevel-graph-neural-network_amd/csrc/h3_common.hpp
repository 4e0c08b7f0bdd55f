// Shared pieces of the f16x3 kernels (gin_h3.hip, linear_h3.hip): two-piece fp16 operand split,
// exact power-of-two range normalisation, packed-weight layout.
//
// x = h1 + h2 with h1 = fp16(x), h2 = fp16(x - h1): 11 + 11 significand bits, representation error
// 2^-23 |x|.  A product a*w is a1*w1 + a1*w2 + a2*w1 (three fp16 MFMAs, fp32 accumulation); the
// dropped a2*w2 is at most 2^-22 |a||w| (|h2| <= 2^-11 |x| for each operand).
#pragma once
#include "x6_common.hpp"

namespace blgnn {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));

constexpr int HHDR = 8;          // packed image header, in fp16 elements (16 bytes): float inv_scale

struct Split2 { _Float16 h1, h2; };
__device__ __forceinline__ Split2 split2(float x) {
    Split2 s;
    s.h1 = (_Float16)x;
    s.h2 = (_Float16)(x - (float)s.h1);
    return s;
}

// power-of-two scale that moves a magnitude into [2^14, 2^15); returns (scale, 1 / scale)
__device__ __forceinline__ void pow2_scale(float amax, float &scale, float &inv) {
    unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;          // biased exponent of the maximum
    e = e < 32u ? 32u : (e > 222u ? 222u : e);                   // zeros / denormals / overflow guard
    scale = __uint_as_float((268u - e) << 23);                    // 2^(14 - (e - 127))
    inv = __uint_as_float((e - 14u) << 23);                       // 2^((e - 127) - 14)
}

}  // namespace blgnn
