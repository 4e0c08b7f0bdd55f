// Shared pieces of the dense-transform kernels (linear.hip: exact-f32 MFMA, linear_x6.hip: bf16x6).
#pragma once
#include "common.hpp"

namespace blgnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KCH = 32;      // accumulation chain length (two-level accumulation)

struct LinearArgs {
    const float *x; int64_t ldx;
    const float *w; int64_t ldw;
    float *y; int64_t ldy;
    int32_t M, N, K;
    const float *rowvec, *rowvec_w;
    Epilogue epi;
    int vec_a, vec_b;      // float4 loads allowed for X / W
    float *amax_out;       // optional [M]: max |y| per row (store_out), zero-initialised by the caller
};

template <int ACT>
__device__ __forceinline__ float act_c(float v, float slope) {
    if (ACT == BLGNN_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == BLGNN_ACT_PRELU) return v > 0.f ? v : slope * v;
    if (ACT == BLGNN_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    if (ACT == BLGNN_ACT_TANH) return tanhf(v);
    return v;
}

// One wave's TM x TN accumulator tiles -> y, 32 rows at a time through its private LDS tile.
// ``rs`` (LDS, outside the staging tiles; indexed by row - row0) and ``cs`` undo the power-of-two
// operand scaling of the f16x3 kernels: acc * rs[row] * cs before the bias.  ``A.amax_out`` (optional,
// zero-initialised by the caller) receives max |y| of every stored row (atomic max over the column
// tiles): the row magnitudes a following f16x3 GEMM normalises by.
// Everything a lane needs per row (inverse scale, rank-1 factor) is fetched once per 32-row slab, and a
// slab that lies inside the matrix is stored without per-element guards: hipcc otherwise wraps every
// access in an exec-mask branch and waits for each rowvec load on its own (a 14 k-cycle epilogue).
template <int ACT, int TM, int TN, int WN, bool SYNC = true>
__device__ __forceinline__ void store_out(const LinearArgs &A, const f32x16 (&acc)[TM][TN], float *smem,
                                          int row0, int col0, int wid, int lane,
                                          const float *rs = nullptr, float cs = 1.f) {
    constexpr int LDO = WN + 4;
    constexpr int Q = WN / 4;                            // float4 per staged row
    float *stage = smem + wid * 32 * LDO;
    const int fi = lane & 31, fh = lane >> 5;
    const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);
    float bias[TN], slope[TN], scale[TN], shift[TN], rw[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = col0 + 32 * j + fi;
        const bool in = n < A.N;
        bias[j] = (in && A.epi.bias) ? A.epi.bias[n] : 0.f;
        slope[j] = (in && ACT == BLGNN_ACT_PRELU && A.epi.act_param) ? A.epi.act_param[n] : 0.f;
        scale[j] = (in && A.epi.bn_scale) ? A.epi.bn_scale[n] : 1.f;
        shift[j] = (in && A.epi.bn_scale && A.epi.bn_shift) ? A.epi.bn_shift[n] : 0.f;
        rw[j] = (in && A.rowvec_w) ? A.rowvec_w[n] : 0.f;
    }
    if (SYNC) __syncthreads();                           // every wave is done reading As / Bs
    const bool cols_in = col0 + WN <= A.N;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int slab0 = row0 + 32 * i;
        if (slab0 >= A.M) break;                         // wave-uniform
        const bool interior = cols_in && slab0 + 32 <= A.M && vec_out;
        float rsc[16], rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;              // row inside the 32-row slab
            rsc[r] = rs ? rs[32 * i + lr] * cs : 1.f;
            rv[r] = 0.f;
        }
        if (A.rowvec) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = slab0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                rv[r] = A.rowvec[m < A.M ? m : A.M - 1];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = acc[i][j][r];
                if (rs) v *= rsc[r];
                v = fmaf(rv[r], rw[j], v + bias[j]);
                v = act_c<ACT>(v, slope[j]);
                if (A.epi.bn_scale) v = fmaf(v, scale[j], shift[j]);
                stage[lr * LDO + 32 * j + fi] = v;
            }
        // wave-private tile: LDS operations of one wave complete in order, no barrier needed
        if (interior) {
#pragma unroll
            for (int t = lane; t < 32 * Q; t += WAVE) {
                const int lr = t / Q, c4 = t % Q;
                *reinterpret_cast<float4 *>(A.y + (int64_t)(slab0 + lr) * A.ldy + col0 + 4 * c4) =
                    *reinterpret_cast<const float4 *>(stage + lr * LDO + 4 * c4);
            }
        } else {
            for (int t = lane; t < 32 * Q; t += WAVE) {
                const int lr = t / Q, c4 = t % Q;
                const int m = slab0 + lr, n = col0 + 4 * c4;
                if (m >= A.M || n >= A.N) continue;
                const float4 v = *reinterpret_cast<const float4 *>(stage + lr * LDO + 4 * c4);
                float *dst = A.y + (int64_t)m * A.ldy + n;
                if (vec_out && n + 3 < A.N) *reinterpret_cast<float4 *>(dst) = v;
                else {
                    dst[0] = v.x;
                    if (n + 1 < A.N) dst[1] = v.y;
                    if (n + 2 < A.N) dst[2] = v.z;
                    if (n + 3 < A.N) dst[3] = v.w;
                }
            }
        }
        if (A.amax_out) {                                // one 32-lane atomic per slab: lane = row
            const int m = slab0 + fi;
            const int ncol = A.N - col0 < WN ? A.N - col0 : WN;
            float mx = 0.f;
            if (fh == 0) {
                for (int c = 0; c < ncol; ++c) mx = fmaxf(mx, fabsf(stage[fi * LDO + c]));
                if (m < A.M) atomicMax(reinterpret_cast<unsigned *>(A.amax_out) + m, __float_as_uint(mx));
            }
        }
    }
}

}  // namespace blgnn
