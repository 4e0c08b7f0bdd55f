// "sp" (split-plane) activation format: what the f16x3 GEMM of linear_sp.hip reads and what its producers write.
//
// An fp32 activation matrix X[M, K] (K % 256 == 0) is kept as
//   xs : fp16 [M][K / 32][64]   per row and 32-column chunk: 32 high pieces h1, then 32 low pieces h2 (128 B =
//                               one cache line per (row, K step of the GEMM));
//   xe : uint8 [M][K / 256]     the power-of-two scale of every (row, 256-column block), as the biased fp32
//                               exponent field e of its inverse: x[m, k] = (h1 + h2) * 2^(e - 127),
//                               h1 = fp16(x * s), h2 = fp16(x * s - h1), s = 2^(127 - e) chosen so that the
//                               block's largest magnitude lies in [2^14, 2^15); one byte per block, so the GEMM
//                               fetches the (up to eight) block scales of a row with ONE 8-byte load
// -- the same 4 bytes per element as fp32, the same 22 significand bits as the in-kernel split of linear_h3.hip
// (h3_common.hpp), but formed ONCE by the kernel that produces X (its epilogue sees whole 256-column blocks of
// a row) instead of by every column block of every GEMM that reads X.  Block scales instead of one scale per
// row: no producer needs to know the other producers' columns (the GMN f_node input [x | S | u] has three), no
// pass over finished rows, and a block 2^13 below its row's maximum keeps its low-order bits.
#pragma once
#include "h3_common.hpp"

namespace blgnn {

constexpr int SP_CHUNK = 32;              // columns per chunk
constexpr int SP_BLOCK = 256;             // columns per scale block

// exponent byte <-> inverse scale (a power of two: its fp32 bit pattern is the exponent field alone)
__device__ __forceinline__ uint8_t sp_exp_of(float inv) { return (uint8_t)(__float_as_uint(inv) >> 23); }
__device__ __forceinline__ float sp_inv_of(uint8_t e) { return __uint_as_float((unsigned)e << 23); }

// lane-group helper for kernels whose lanes own 4 consecutive columns (float4) of a row: writes the two 8-byte
// piece groups of those columns.  `col` = column inside the row (multiple of 4), v already scaled.
__device__ __forceinline__ void sp_store4(_Float16 *__restrict__ row, int col, const f32x4v v) {
    const f16x4 h1 = __builtin_convertvector(v, f16x4);
    const f16x4 h2 = __builtin_convertvector(v - __builtin_convertvector(h1, f32x4v), f16x4);
    _Float16 *dst = row + (col >> 5) * 64 + (col & 31);
    *reinterpret_cast<f16x4 *>(dst) = h1;
    *reinterpret_cast<f16x4 *>(dst + 32) = h2;
}

// The same for a wavefront whose lanes l and l ^ 1 own ADJACENT column groups (col and col + 4, col % 8 == 0 on the even
// lane), called by all lanes together: the even lane stores both lanes' high pieces, the odd lane both low pieces --
// one 16-byte store per lane instead of two 8-byte ones (8-byte stores run at 0.5-0.7x the 16-byte rate).
__device__ __forceinline__ void sp_store4_pair(_Float16 *__restrict__ row, int col, const f32x4v v, int lane) {
    const f16x4 h1 = __builtin_convertvector(v, f16x4);
    const f16x4 h2 = __builtin_convertvector(v - __builtin_convertvector(h1, f32x4v), f16x4);
    const uint2 u1 = __builtin_bit_cast(uint2, h1), u2 = __builtin_bit_cast(uint2, h2);
    const bool odd = lane & 1;
    const uint2 keep = odd ? u2 : u1, send = odd ? u1 : u2;
    uint2 recv;
    recv.x = (unsigned)__shfl_xor((int)send.x, 1, 64);
    recv.y = (unsigned)__shfl_xor((int)send.y, 1, 64);
    const uint4 out = odd ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y);
    const int ce = col & ~4;                                   // the even lane's column: a multiple of 8 (16 bytes of fp16)
    *reinterpret_cast<uint4 *>(row + (ce >> 5) * 64 + (ce & 31) + (odd ? 32 : 0)) = out;
}

}  // namespace blgnn
