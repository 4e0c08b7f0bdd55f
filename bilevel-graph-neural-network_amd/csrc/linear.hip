// Dense feature transform on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   Y = epi( X[M,K] @ W[K,N] + bias + rowvec (x) rowvec_w )      epi = act, then affine
//
// Exact-f32 MFMA (k-ordered fma chain, one rounding per product): results match a
// plain fp32 reference to summation-order noise, which the 1e-5 parity bar needs.
//
// Structure: 256 threads = 4 wavefronts; block tile BM x BN, K step 32, operands
// staged through LDS in k-major images (As[k][m], Bs[k][n]) so that every
// ds_read_b32 of a 32x32x2 fragment is bank-conflict free (lane i -> bank i).  The
// next K tile is fetched global -> registers while the current one is multiplied.
// Rows of X are the long dimension (atoms / drugs); W is small and L2 resident.
#include "linear_common.hpp"

namespace blgnn {

template <int BM, int BN, int WM, int WN, int BK>
__global__ void __launch_bounds__(256, 2) k_linear(const LinearArgs A) {
    constexpr int LDA = BM + 1;            // As[k][m]: +1 dword -> conflict-free transposed stores
    constexpr int LDB = BN + 4;            // Bs[k][n]: keeps 16 B alignment for b128 stores
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int A_KQ = BK / 4;           // float4 per X row per K tile
    constexpr int A_ROWS = 256 / A_KQ;     // X rows covered per pass
    constexpr int A_F4 = BM / A_ROWS;      // float4 per thread per K tile (X)
    constexpr int B_F4 = BN * BK / 1024;   // float4 per thread per K tile (W)
    constexpr int B_ROWQ = BN / 4;         // float4 per Bs row
    constexpr int B_KSTEP = 256 / B_ROWQ;
    static_assert((BM / WM) * (BN / WN) == 4, "4 wavefronts per workgroup");

    // operand images during the K loop; per-wave 32-row output staging tiles afterwards
    constexpr int LDO = WN + 4;
    constexpr int OPND = BK * LDA + 3 + BK * LDB;     // As, pad to 16 B, Bs
    constexpr int STAGE = 4 * 32 * LDO;
    __shared__ __attribute__((aligned(16))) float smem[OPND > STAGE ? OPND : STAGE];
    float *As = smem;
    float *Bs = smem + ((BK * LDA + 3) & ~3);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WAVES_N) * WM, wn0 = (wid % WAVES_N) * WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_F4], rb[B_F4];
    const int a_kq = tid % A_KQ, a_row = tid / A_KQ;     // X: A_KQ threads x 16 B per row
    const int b_nq = tid % B_ROWQ, b_k = tid / B_ROWQ;   // W: B_ROWQ threads per k row

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int m = m0 + a_row + A_ROWS * i, k = k0 + 4 * a_kq;
            float4 v = f4_zero();
            if (m < A.M) {
                const float *p = A.x + (int64_t)m * A.ldx + k;
                if (A.vec_a) { if (k < A.K) v = *reinterpret_cast<const float4 *>(p); }
                else {
                    if (k + 0 < A.K) v.x = p[0];
                    if (k + 1 < A.K) v.y = p[1];
                    if (k + 2 < A.K) v.z = p[2];
                    if (k + 3 < A.K) v.w = p[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int k = k0 + b_k + B_KSTEP * i, n = n0 + 4 * b_nq;
            float4 v = f4_zero();
            if (k < A.K) {
                const float *p = A.w + (int64_t)k * A.ldw + n;
                if (A.vec_b) { if (n < A.N) v = *reinterpret_cast<const float4 *>(p); }
                else {
                    if (n + 0 < A.N) v.x = p[0];
                    if (n + 1 < A.N) v.y = p[1];
                    if (n + 2 < A.N) v.z = p[2];
                    if (n + 3 < A.N) v.w = p[3];
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int m = a_row + A_ROWS * i, k = 4 * a_kq;
            As[(k + 0) * LDA + m] = ra[i].x;
            As[(k + 1) * LDA + m] = ra[i].y;
            As[(k + 2) * LDA + m] = ra[i].z;
            As[(k + 3) * LDA + m] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int k = b_k + B_KSTEP * i;
            *reinterpret_cast<float4 *>(&Bs[k * LDB + 4 * b_nq]) = rb[i];
        }
    };

    const int KT = (A.K + BK - 1) / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    const int fi = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_tile((kt + 1) * BK);
        // two-level accumulation: a fresh 32-deep fma chain per K tile, added to the running
        // sum afterwards.  Keeps the chain length at 32 for any K (the error of one long
        // k-ordered chain grows ~4x faster than a blocked CPU sum at K >= 128).
#pragma unroll
        for (int kk = 0; kk < BK; kk += KCH) {
            f32x16 part[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) part[i][j][r] = 0.f;
#pragma unroll
            for (int ks = kk; ks < kk + KCH; ks += 2) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = As[(ks + fh) * LDA + wm0 + 32 * i + fi];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = Bs[(ks + fh) * LDB + wn0 + 32 * j + fi];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        part[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], part[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] += part[i][j];
        }
        __syncthreads();
        if (kt + 1 < KT) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue.  C/D map of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // The activation switch is hoisted out of the element loop (one code path per activation),
    // and the tile goes through a wave-private LDS staging area so that global stores are whole
    // row segments at 16 B per lane (64 scalar stores per lane were store-issue bound).
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    store_out<BLGNN_ACT_RELU, TM, TN, WN>(A, acc, smem, m0 + wm0, n0 + wn0, wid, lane); break;
        case BLGNN_ACT_PRELU:   store_out<BLGNN_ACT_PRELU, TM, TN, WN>(A, acc, smem, m0 + wm0, n0 + wn0, wid, lane); break;
        case BLGNN_ACT_SIGMOID: store_out<BLGNN_ACT_SIGMOID, TM, TN, WN>(A, acc, smem, m0 + wm0, n0 + wn0, wid, lane); break;
        case BLGNN_ACT_TANH:    store_out<BLGNN_ACT_TANH, TM, TN, WN>(A, acc, smem, m0 + wm0, n0 + wn0, wid, lane); break;
        default:                store_out<BLGNN_ACT_IDENTITY, TM, TN, WN>(A, acc, smem, m0 + wm0, n0 + wn0, wid, lane); break;
    }
}

// Split-K variant for a few thousand rows (the interaction level): a 64x64 output tile is
// computed by 16 wavefronts = 4 K-groups x (2x2) tile waves.  K-group g multiplies the K tiles
// g, g+4, ... from its own LDS operand images; the four partial tiles meet in LDS in a fixed
// order (deterministic) and K-group 0 runs the epilogue.  With one wave per 32x32 tile the run
// time of such a GEMM is the SERIAL MFMA chain of a wave (K/2 x 64 cycles: 8.5 us at K=640)
// on 16 % of the SIMDs; splitting K four ways uses four times the SIMDs for a quarter as long.
template <int KG>
__global__ void __launch_bounds__(256 * KG) k_linear_splitk(const LinearArgs A) {
    constexpr int BM = 64, BN = 64, BKT = 32;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    constexpr int OPND = (BKT * LDA + 3 + BKT * LDB + 3) & ~3;
    constexpr int LDO = 32 + 4;
    __shared__ __attribute__((aligned(16))) float smem[KG * OPND];
    const int tid = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int lane = tid & 63, wid = tid >> 6;
    float *As = smem + grp * OPND;
    float *Bs = As + ((BKT * LDA + 3) & ~3);
    const int wm0 = (wid >> 1) * 32, wn0 = (wid & 1) * 32;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int fi = lane & 31, fh = lane >> 5;
    const int a_kq = tid & 7, a_row = tid >> 3;
    const int b_nq = tid & 15, b_k = tid >> 4;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra[2], rb[2];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + a_row + 32 * i, k = k0 + 4 * a_kq;
            float4 v = f4_zero();
            if (m < A.M) {
                const float *p = A.x + (int64_t)m * A.ldx + k;
                if (A.vec_a) { if (k < A.K) v = *reinterpret_cast<const float4 *>(p); }
                else {
                    if (k + 0 < A.K) v.x = p[0];
                    if (k + 1 < A.K) v.y = p[1];
                    if (k + 2 < A.K) v.z = p[2];
                    if (k + 3 < A.K) v.w = p[3];
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + b_k + 16 * i, n = n0 + 4 * b_nq;
            float4 v = f4_zero();
            if (k < A.K) {
                const float *p = A.w + (int64_t)k * A.ldw + n;
                if (A.vec_b) { if (n < A.N) v = *reinterpret_cast<const float4 *>(p); }
                else {
                    if (n + 0 < A.N) v.x = p[0];
                    if (n + 1 < A.N) v.y = p[1];
                    if (n + 2 < A.N) v.z = p[2];
                    if (n + 3 < A.N) v.w = p[3];
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = a_row + 32 * i, k = 4 * a_kq;
            As[(k + 0) * LDA + m] = ra[i].x;
            As[(k + 1) * LDA + m] = ra[i].y;
            As[(k + 2) * LDA + m] = ra[i].z;
            As[(k + 3) * LDA + m] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<float4 *>(&Bs[(b_k + 16 * i) * LDB + 4 * b_nq]) = rb[i];
    };

    const int KT = (A.K + BKT - 1) / BKT;
    const int iters = (KT + KG - 1) / KG;               // same trip count for every K-group
    load_tile(grp * BKT);                               // tiles past K load zeros
    store_tile();
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (it + 1 < iters) load_tile(((it + 1) * KG + grp) * BKT);
        f32x16 part;
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < BKT; ks += 2) {
            const float a = As[(ks + fh) * LDA + wm0 + fi];
            const float b = Bs[(ks + fh) * LDB + wn0 + fi];
            part = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, part, 0, 0, 0);
        }
        acc += part;
        __syncthreads();
        if (it + 1 < iters) {
            store_tile();
            __syncthreads();
        }
    }
    // partial tiles of K-groups 1..KG-1 -> LDS (operand images are dead), summed by group 0
    float *red = smem;                                   // [KG-1][4 waves][16][64]
    if (grp > 0) {
        float *dst = red + ((grp - 1) * 4 + wid) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float *src = red + ((g - 1) * 4 + wid) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += src[r * 64 + lane];
        }
        // epilogue (same as store_out, one 32x32 tile per wave, staging above the reduce area)
        float *stage = smem + (KG - 1) * 4 * 1024 + wid * 32 * LDO;
        const int n = n0 + wn0 + fi;
        const bool in = n < A.N;
        const float bias = (in && A.epi.bias) ? A.epi.bias[n] : 0.f;
        const float slope = (in && A.epi.act == BLGNN_ACT_PRELU && A.epi.act_param) ? A.epi.act_param[n] : 0.f;
        const float scale = (in && A.epi.bn_scale) ? A.epi.bn_scale[n] : 1.f;
        const float shift = (in && A.epi.bn_scale && A.epi.bn_shift) ? A.epi.bn_shift[n] : 0.f;
        const float rw = (in && A.rowvec_w) ? A.rowvec_w[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
            const int m = m0 + wm0 + lr;
            float v = acc[r] + bias;
            if (A.rowvec) v = fmaf((m < A.M) ? A.rowvec[m] : 0.f, rw, v);
            v = apply_act(v, A.epi.act, slope);
            if (A.epi.bn_scale) v = fmaf(v, scale, shift);
            stage[lr * LDO + fi] = v;
        }
        const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);
#pragma unroll
        for (int t = lane; t < 32 * 8; t += WAVE) {
            const int lr = t >> 3, c4 = t & 7;
            const int m = m0 + wm0 + lr, nn = n0 + wn0 + 4 * c4;
            if (m >= A.M || nn >= A.N) continue;
            const float4 v = *reinterpret_cast<const float4 *>(stage + lr * LDO + 4 * c4);
            float *dst = A.y + (int64_t)m * A.ldy + nn;
            if (vec_out && nn + 3 < A.N) *reinterpret_cast<float4 *>(dst) = v;
            else {
                dst[0] = v.x;
                if (nn + 1 < A.N) dst[1] = v.y;
                if (nn + 2 < A.N) dst[2] = v.z;
                if (nn + 3 < A.N) dst[3] = v.w;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BK = 32>
static void launch_linear(const LinearArgs &A, hipStream_t s) {
    dim3 grid((A.M + BM - 1) / BM, (A.N + BN - 1) / BN);
    hipLaunchKernelGGL((k_linear<BM, BN, WM, WN, BK>), grid, dim3(256), 0, s, A);
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_linear(const float *x, int64_t ldx, const float *w, int64_t ldw, float *y,
                            int64_t ldy, int32_t M, int32_t N, int32_t K, const float *bias,
                            const float *rowvec, const float *rowvec_w, int32_t act,
                            const float *act_param, const float *bn_scale, const float *bn_shift,
                            void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "linear: negative size");
    if (M == 0 || N == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(y && (K == 0 || (x && w)), "linear: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldw >= N && ldy >= N, "linear: leading dimension too small");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "linear: bad activation id");
    BLGNN_CHECK_ARG((rowvec == nullptr) == (rowvec_w == nullptr), "linear: rowvec and rowvec_w go together");
    BLGNN_CHECK_ARG(x != y, "linear: x and y must not alias");
    LinearArgs A;
    A.x = x; A.ldx = ldx; A.w = w; A.ldw = ldw; A.y = y; A.ldy = ldy;
    A.M = M; A.N = N; A.K = K; A.rowvec = rowvec; A.rowvec_w = rowvec_w;
    A.epi = Epilogue{bias, act_param, bn_scale, bn_shift, act};
    A.vec_a = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
    A.vec_b = (N % 4 == 0) && (ldw % 4 == 0) && aligned16(w); A.amax_out = nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (N <= 32) {
        if (M >= 32768) launch_linear<256, 32, 64, 32>(A, s);
        else launch_linear<128, 32, 32, 32>(A, s);
    } else if (N <= 64) {
        if (M >= 16384) launch_linear<128, 64, 32, 64>(A, s);
        else launch_linear<64, 64, 32, 32>(A, s);
    } else {
        // 128x128 tiles only when they fill the chip several times over; a grid of 1-2
        // tiles per CU (36 k atoms x 128) runs in whole "rounds" and loses up to half of one,
        // so mid-size problems take 64-row tiles that are all resident together.
        const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
        if (tiles >= 1024) launch_linear<128, 128, 64, 64>(A, s);
        else if (M >= 4096) launch_linear<64, 128, 32, 64>(A, s);
        // a few thousand rows (the interaction level): a handful of workgroups whose run time is
        // one wave's serial MFMA chain -> split K over 4x the wavefronts (k_linear_splitk)
        else if (K >= 128) {
            dim3 grid((M + 63) / 64, (N + 63) / 64);
            hipLaunchKernelGGL((k_linear_splitk<4>), grid, dim3(1024), 0, s, A);
        } else launch_linear<64, 64, 32, 32>(A, s);
    }
    return launch_status("linear");
}
