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
// operand scaling of the f16x3 kernels: acc * rs[row] * cs before the bias.
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
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;          // row inside the 32-row slab
                const int m = row0 + 32 * i + lr;
                float v = acc[i][j][r];
                if (rs) v = v * rs[32 * i + lr] * cs;
                v += bias[j];
                if (A.rowvec) v = fmaf((m < A.M) ? A.rowvec[m] : 0.f, rw[j], v);
                v = act_c<ACT>(v, slope[j]);
                if (A.epi.bn_scale) v = fmaf(v, scale[j], shift[j]);
                stage[lr * LDO + 32 * j + fi] = v;
            }
        // wave-private tile: LDS operations of one wave complete in order, no barrier needed
#pragma unroll
        for (int t = lane; t < 32 * Q; t += WAVE) {
            const int lr = t / Q, c4 = t % Q;
            const int m = row0 + 32 * i + lr, n = col0 + 4 * c4;
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
}

}  // namespace blgnn
