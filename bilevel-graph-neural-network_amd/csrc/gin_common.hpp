// Shared device helpers of the fused GIN kernels (gin_fused.hip: one layer per launch,
// gin_stack.hip: the whole lower stack per launch).
#pragma once
#include "common.hpp"

namespace blgnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;      // K chunk (two-level accumulation, see linear.hip)
constexpr int GCOLS = 1024; // column indices of a tile kept in LDS (64 rows x mean degree 2.1 = 136)

template <int ACT>
__device__ __forceinline__ float act_t(float v, float slope) {
    if (ACT == BLGNN_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == BLGNN_ACT_PRELU) return v > 0.f ? v : slope * v;
    if (ACT == BLGNN_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    if (ACT == BLGNN_ACT_TANH) return tanhf(v);
    return v;
}

// B fragments of one 32-deep K chunk: lane (fi, fh) holds B[kc + 2s + fh][n_base + 32 nt + fi].
// Rows past Kdim are clamped to the last row: the matching A columns are zero, so any finite
// value will do, and the loads stay unconditional (no exec-mask branches in the hot loop).
template <int N, int NTW>
__device__ __forceinline__ void load_b_chunk(float (&dst)[KC / 2][NTW], const float *__restrict__ B,
                                             int Kdim, int kc, int n_base, int fi, int fh) {
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) {
        int k = kc + 2 * s + fh;
        k = k < Kdim ? k : Kdim - 1;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) dst[s][nt] = B[(int64_t)k * N + n_base + nt * 32 + fi];
    }
}

// acc += A[GBM x Kpad](LDS, row-major, stride lda) @ B[Kdim x N](global).  ``bcur`` arrives
// holding chunk 0 (the caller issued its loads early, under other work).
template <int N, int MT, int NTW>
__device__ __forceinline__ void gemm_lds_global(f32x16 (&acc)[MT][NTW], const float *As, int lda,
                                                const float *__restrict__ B, int Kdim, int Kpad,
                                                int m_base, int n_base, int fi, int fh,
                                                float (&bcur)[KC / 2][NTW]) {
    float bnxt[KC / 2][NTW];
    const float *arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = As + (m_base + mt * 32 + fi) * lda + fh;
    for (int kc = 0; kc < Kpad; kc += KC) {
        const bool more = kc + KC < Kpad;
        if (more) load_b_chunk<N, NTW>(bnxt, B, Kdim, kc + KC, n_base, fi, fh);
        f32x16 part[MT][NTW];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) part[mt][nt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KC / 2; ++s) {
            float a[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[mt] = arow[mt][kc + 2 * s];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
                    part[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bcur[s][nt], part[mt][nt], 0, 0, 0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] += part[mt][nt];
        if (more) {
#pragma unroll
            for (int s = 0; s < KC / 2; ++s)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bcur[s][nt] = bnxt[s][nt];
        }
    }
}

// dst[m][c] = affine(act(acc + bias[c])) for the wave's MT x NTW tiles (lane = column, regs = rows)
template <int ACT, int MT, int NTW>
__device__ __forceinline__ void tile_to_lds(const f32x16 (&acc)[MT][NTW], float *dst, int ld,
                                            const float *bias, const float *slope_v, const float *scale_v,
                                            const float *shift_v, int m_base, int n_base, int fi, int fh) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = n_base + nt * 32 + fi;
        const float b = bias ? bias[c] : 0.f;
        const float sl = (ACT == BLGNN_ACT_PRELU && slope_v) ? slope_v[c] : 0.f;
        const float sc = scale_v ? scale_v[c] : 1.f;
        const float sh = (scale_v && shift_v) ? shift_v[c] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = act_t<ACT>(acc[mt][nt][r] + b, sl);
                if (scale_v) v = fmaf(v, sc, sh);
                dst[m * ld + c] = v;
            }
    }
}

}  // namespace blgnn
