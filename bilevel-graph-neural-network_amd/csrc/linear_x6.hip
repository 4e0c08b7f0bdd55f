// Dense feature transform on the bf16 matrix cores at fp32 accuracy ("bf16x6", see x6_common.hpp /
// gin_x6.hip for the scheme and its measured error).
//
//   Y = epi( X[M,K] @ W[K,N] + bias + rowvec (x) rowvec_w )      epi = act, then affine
//
// Same contract as blgnn_linear; W arrives pre-split and packed (blgnn_pack_weight_x6).  Per
// 16 k the six bf16 MFMAs of a 32x32 tile take 192 cycles against 512 for eight
// v_mfma_f32_32x32x2_f32, so the large GEMMs of the path (GMN f_message / f_node, GAT / GCN
// x @ W at d >= 256, LinkPred MLP on big batches) run ~2.5x closer to the HBM roofline.
//
// Structure: 256 threads = 2 x 2 wavefronts, block tile 128 x 128, wave tile 64 x 64, K step 16.
// X tile: global fp32 -> registers (requested two K steps ahead) -> split -> three bf16 images
// [piece][m][k] in LDS.  W tile: the packed pieces are copied as they are into [piece][n][k].
// Row stride 48 B: ds_read_b128 fragments (8 consecutive k of one row) are bank-conflict free.
#include "linear_common.hpp"
#include "x6_common.hpp"

namespace blgnn {

struct LinearX6Args {
    LinearArgs base;             // x, ldx, y, ldy, M, N, K, rowvec, rowvec_w, epi (w / ldw unused)
    const __bf16 *wp;            // packed pieces [3][Kp/8][N][8]
    int32_t Kp;                  // K rounded up to 32
};

constexpr int XBK = 16;                  // K step = one MFMA k
constexpr int XLD = (XBK + 8) * 2;       // LDS row stride in bytes (48): conflict-free ds_read_b128 fragments
constexpr int X6_BM = 128, X6_BN = 128;
constexpr int X6_PIECE = X6_BM * XLD;    // bytes of one piece image (BM == BN)
constexpr int X6_BUF = 6 * X6_PIECE;     // 3 X pieces + 3 W pieces
constexpr int X6_STAGE = 4 * 32 * (64 + 4) * 4;          // store_out's per-wave staging tiles
constexpr int X6_LDS = 2 * X6_BUF > X6_STAGE ? 2 * X6_BUF : X6_STAGE;

// Double-buffered operand images, ONE barrier per K step: while the MFMAs of step k run out of
// buffer k&1, the same wave splits / parks the registers holding step k+1 into the other buffer and
// requests step k+2 from global memory.
template <int ACT>
__global__ void __launch_bounds__(256, 2) k_linear_x6(const LinearX6Args P) {
    constexpr int BM = X6_BM, BN = X6_BN, WM = 64, WN = 64, TM = 2, TN = 2, PIECE = X6_PIECE;
    extern __shared__ float4 x6_smem[];
    char *smem_raw = reinterpret_cast<char *>(x6_smem);
    const LinearArgs &A = P.base;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;
    // column blocks of one row block are neighbours in launch order: its X tile is read from HBM once
    const int nb = (A.N + BN - 1) / BN;
    const int m0 = (blockIdx.x / nb) * BM, n0 = (blockIdx.x % nb) * BN;

    f32x16 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    // staging registers: X tile 128 x 16 fp32 = 2 float4 per thread; W pieces 3 x 2 x 128 words of 16 B = 3
    float4 ra[2];
    bf16x8 rb[3];
    const int a_kq = tid & 3, a_row = tid >> 2;          // 4 threads x 16 B per X row, 64 rows per pass
    const float4 *__restrict__ X4 = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;
    const int64_t wpiece = (int64_t)(P.Kp / 8) * A.N;    // 16-byte words per packed piece
    const bf16x8 *__restrict__ W8 = reinterpret_cast<const bf16x8 *>(P.wp);

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int m = m0 + a_row + 64 * p, k = k0 + 4 * a_kq;
            ra[p] = (m < A.M && k < A.K) ? X4[(int64_t)m * ldx4 + (k >> 2)] : f4_zero();
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {                     // piece i: (kb, n) = (tid / 128, tid % 128)
            const int kb = tid >> 7, n = n0 + (tid & 127);
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
            rb[i] = n < A.N ? W8[i * wpiece + (int64_t)(k0 / 8 + kb) * A.N + n] : z;
        }
    };
    auto store_tile = [&](char *buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const Split3 s0 = split3(ra[p].x), s1 = split3(ra[p].y), s2 = split3(ra[p].z), s3 = split3(ra[p].w);
            bf16x4 h = {s0.hi, s1.hi, s2.hi, s3.hi}, md = {s0.mid, s1.mid, s2.mid, s3.mid},
                   lo = {s0.lo, s1.lo, s2.lo, s3.lo};
            char *dst = buf + (a_row + 64 * p) * XLD + a_kq * 8;
            *reinterpret_cast<bf16x4 *>(dst) = h;
            *reinterpret_cast<bf16x4 *>(dst + PIECE) = md;
            *reinterpret_cast<bf16x4 *>(dst + 2 * PIECE) = lo;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            *reinterpret_cast<bf16x8 *>(buf + (3 + i) * PIECE + (tid & 127) * XLD + (tid >> 7) * 16) = rb[i];
    };

    const int KT = P.Kp / XBK;
    load_tile(0);
    store_tile(smem_raw);
    if (KT > 1) load_tile(XBK);
    lds_barrier();
    const int aoff = (wm0 + fi) * XLD + fh * 16;         // lane (row fi, k half fh)
    const int boff = 3 * PIECE + (wn0 + fi) * XLD + fh * 16;
    for (int kt = 0; kt < KT; ++kt) {
        const char *cur = smem_raw + (kt & 1) * X6_BUF;
        char *nxt = smem_raw + ((kt & 1) ^ 1) * X6_BUF;
        bf16x8 a[TM][3], b[TN][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i][q] = *reinterpret_cast<const bf16x8 *>(cur + aoff + q * PIECE + 32 * i * XLD);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j][q] = *reinterpret_cast<const bf16x8 *>(cur + boff + q * PIECE + 32 * j * XLD);
        }
        // term-major order: consecutive MFMAs write different accumulators, so none waits on the result
        // of the one just issued (the five cross terms of one tile form a dependent chain)
#define X6_TERM(ACC, QA, QB)                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)       \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][QA], b[j][QB], ACC[i][j], 0, 0, 0);
        X6_TERM(acc0, 0, 0)
        X6_TERM(acc1, 0, 1)
        if (kt + 1 < KT) store_tile(nxt);                // registers hold step kt+1
        if (kt + 2 < KT) load_tile((kt + 2) * XBK);
        X6_TERM(acc1, 1, 0)
        X6_TERM(acc1, 0, 2)
        X6_TERM(acc1, 2, 0)
        X6_TERM(acc1, 1, 1)
#undef X6_TERM
        lds_barrier();       // buffer kt&1 fully read (it is overwritten in step kt+1), buffer (kt+1)&1 complete
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc0[i][j] += acc1[i][j];
    store_out<ACT, TM, TN, WN>(A, acc0, reinterpret_cast<float *>(smem_raw), m0 + wm0, n0 + wn0, wid, lane);
}

template <int ACT>
static int launch_linear_x6(const LinearX6Args &P, hipStream_t s) {
    static bool declared = false;
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_x6<ACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "linear_x6: cannot raise the dynamic LDS limit to %d bytes", X6_LDS);
        declared = true;
    }
    const int64_t blocks = (int64_t)((P.base.M + X6_BM - 1) / X6_BM) * ((P.base.N + X6_BN - 1) / X6_BN);
    hipLaunchKernelGGL(k_linear_x6<ACT>, dim3((unsigned)blocks), dim3(256), X6_LDS, s, P);
    return launch_status("linear_x6");
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_linear_x6(const float *x, int64_t ldx, const void *w_packed, float *y, int64_t ldy,
                               int32_t M, int32_t N, int32_t K, const float *bias, const float *rowvec,
                               const float *rowvec_w, int32_t act, const float *act_param,
                               const float *bn_scale, const float *bn_shift, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "linear_x6: bad sizes");
    if (M == 0 || N == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && w_packed && y, "linear_x6: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldy >= N, "linear_x6: leading dimension too small");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "linear_x6: bad activation id");
    BLGNN_CHECK_ARG((rowvec == nullptr) == (rowvec_w == nullptr), "linear_x6: rowvec and rowvec_w go together");
    BLGNN_CHECK_ARG(x != y, "linear_x6: x and y must not alias");
    if (K % 4 != 0 || ldx % 4 != 0 || !aligned16(x) || !aligned16(w_packed))
        return fail(BLGNN_ERR_UNSUPPORTED, "linear_x6: needs K %% 4 == 0, ldx %% 4 == 0 and 16-byte aligned x / packed "
                    "weights (got K=%d ldx=%lld); use blgnn_linear", K, (long long)ldx);
    LinearX6Args P;
    LinearArgs &A = P.base;
    A.x = x; A.ldx = ldx; A.w = nullptr; A.ldw = 0; A.y = y; A.ldy = ldy;
    A.M = M; A.N = N; A.K = K; A.rowvec = rowvec; A.rowvec_w = rowvec_w;
    A.epi = Epilogue{bias, act_param, bn_scale, bn_shift, act};
    A.vec_a = 1; A.vec_b = 0; A.amax_out = nullptr;
    P.wp = reinterpret_cast<const __bf16 *>(w_packed);
    P.Kp = (K + 31) / 32 * 32;
    hipStream_t s = (hipStream_t)stream;
    switch (act) {
        case BLGNN_ACT_RELU:    return launch_linear_x6<BLGNN_ACT_RELU>(P, s);
        case BLGNN_ACT_PRELU:   return launch_linear_x6<BLGNN_ACT_PRELU>(P, s);
        case BLGNN_ACT_SIGMOID: return launch_linear_x6<BLGNN_ACT_SIGMOID>(P, s);
        case BLGNN_ACT_TANH:    return launch_linear_x6<BLGNN_ACT_TANH>(P, s);
        default:                return launch_linear_x6<BLGNN_ACT_IDENTITY>(P, s);
    }
}
