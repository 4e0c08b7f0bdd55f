// Dense feature transform on the fp16 matrix cores at fp32 accuracy ("f16x3", see h3_common.hpp /
// gin_h3.hip for the scheme): half the matrix-core work of linear_x6.hip for the same result.
//
//   Y = epi( X[M,K] @ W[K,N] + bias + rowvec (x) rowvec_w )      epi = act, then affine
//
// Same contract as blgnn_linear; W arrives range-normalised, split and packed (blgnn_pack_weight_h3).
// These are the GEMMs of the GMN propagator (model/layers_gmn.py:19-26,39-47: f_message, f_node) and
// every other nn.Linear / x @ W of the path once M is large.
//
// Range normalisation of X: a workgroup first sweeps its 128 rows once (the panel then sits in the
// XCD's L2 for the K loop and for the sibling column blocks) and scales every row by the power of two
// that moves its largest magnitude into [2^14, 2^15); the inverse goes onto the fp32 accumulators in
// the epilogue together with the weight matrix's.
//
// Structure: 256 threads = 2 x 2 wavefronts, block tile 128 x 128, wave tile 64 x 64, K step 32
// (two MFMA k steps per barrier).  X tile: global fp32 -> registers (requested two K steps ahead) ->
// scale -> split -> two fp16 images [piece][m][32 k] in LDS.  W tile: the packed pieces are copied as
// they are into [piece][n][32 k].  Rows are 64 B with the 16-byte chunk index XORed by (row >> 2) & 3:
// the 16 lanes a ds_read_b128 serves together then cover all 16 chunk slots of the 256-byte bank row.
// Workgroups that share an X row panel run on one XCD (their ids are remapped so that an XCD owns a
// contiguous range of tiles), so the panel leaves HBM once.
#include "linear_common.hpp"
#include "h3_common.hpp"

namespace blgnn {

struct LinearH3Args {
    LinearArgs base;             // x, ldx, y, ldy, M, N, K, rowvec, rowvec_w, epi (w / ldw unused)
    const _Float16 *wp;          // header (inverse scale) + pieces [2][Kp/8][N][8]
    int32_t Kp;                  // K rounded up to 32
};

constexpr int HBK = 32;                   // K step
constexpr int HROW = HBK * 2;             // LDS row: 32 fp16 = 64 B = four 16-byte chunks, XOR swizzled
constexpr int H3_BM = 128, H3_BN = 128;
constexpr int H3_PIECE = H3_BM * HROW;    // bytes of one piece image (BM == BN)
constexpr int H3_BUF = 4 * H3_PIECE;      // 2 X pieces + 2 W pieces
constexpr int H3_STAGE = 4 * 32 * (64 + 4) * 4;          // store_out's per-wave staging tiles
constexpr int H3_MAIN = 2 * H3_BUF > H3_STAGE ? 2 * H3_BUF : H3_STAGE;
constexpr int H3_LDS = H3_MAIN + H3_BM * 4;              // + inverse row scales

__device__ __forceinline__ int h3_off(int row, int chunk) { return row * HROW + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <int ACT>
__global__ void __launch_bounds__(256, 2) k_linear_h3(const LinearH3Args P) {
    constexpr int BM = H3_BM, BN = H3_BN, WM = 64, WN = 64, TM = 2, TN = 2, PIECE = H3_PIECE;
    extern __shared__ float4 h3_smem[];
    char *smem_raw = reinterpret_cast<char *>(h3_smem);
    float *rinv = reinterpret_cast<float *>(smem_raw + H3_MAIN);      // [BM] 1 / row scale
    const LinearArgs &A = P.base;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;
    // tile <- block id: blocks b, b + 8, ... share an XCD; give each XCD a contiguous run of tiles
    // (row-panel major), so the column blocks of one row panel read it from the same L2
    const int nb = (A.N + BN - 1) / BN;
    int tile;
    {
        const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = (tile / nb) * BM, n0 = (tile % nb) * BN;

    const int a_kq = tid & 7, a_row = tid >> 3;          // 8 threads x 16 B per X row of a K step, 32 rows per pass
    const float4 *__restrict__ X4 = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;
    const int64_t wpiece = (int64_t)(P.Kp / 8) * A.N;    // 16-byte words per packed piece
    const f16x8 *__restrict__ W8 = reinterpret_cast<const f16x8 *>(P.wp + HHDR);
    const float winv = *reinterpret_cast<const float *>(P.wp);

    // ---- row maxima of the X panel -> power-of-two scales ------------------------------------------------
    float rscale[4];
    {
        float am[4] = {0.f, 0.f, 0.f, 0.f};
        const int k4n = A.K >> 2;
#pragma unroll 4
        for (int c = a_kq; c < k4n; c += 8)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int m = m0 + a_row + 32 * p;
                if (m < A.M) {
                    const float4 v = X4[(int64_t)m * ldx4 + c];
                    am[p] = fmaxf(fmaxf(am[p], fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                }
            }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float a = am[p];
            a = fmaxf(a, __shfl_xor(a, 1, WAVE));
            a = fmaxf(a, __shfl_xor(a, 2, WAVE));
            a = fmaxf(a, __shfl_xor(a, 4, WAVE));
            float inv;
            pow2_scale(a, rscale[p], inv);
            if (a_kq == 0) rinv[a_row + 32 * p] = inv;
        }
    }

    f32x16 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }

    // staging registers: X tile 128 x 32 fp32 = 4 float4 per thread; W pieces 2 x 4 x 128 words of 16 B = 4
    float4 ra[4];
    f16x8 rb[4];
    const int w_kb = tid >> 7, w_n = tid & 127;           // word (piece q, k block kb + 2 h, column n)
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int m = m0 + a_row + 32 * p, k = k0 + 4 * a_kq;
            ra[p] = (m < A.M && k < A.K) ? X4[(int64_t)m * ldx4 + (k >> 2)] : f4_zero();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i >> 1, kb = w_kb + 2 * (i & 1), n = n0 + w_n;
            f16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
            rb[i] = n < A.N ? W8[q * wpiece + (int64_t)(k0 / 8 + kb) * A.N + n] : z;
        }
    };
    auto store_tile = [&](char *buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float s = rscale[p];
            const Split2 s0 = split2(ra[p].x * s), s1 = split2(ra[p].y * s), s2 = split2(ra[p].z * s),
                         s3 = split2(ra[p].w * s);
            f16x4 h1 = {s0.h1, s1.h1, s2.h1, s3.h1}, h2 = {s0.h2, s1.h2, s2.h2, s3.h2};
            char *dst = buf + h3_off(a_row + 32 * p, a_kq >> 1) + (a_kq & 1) * 8;
            *reinterpret_cast<f16x4 *>(dst) = h1;
            *reinterpret_cast<f16x4 *>(dst + PIECE) = h2;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<f16x8 *>(buf + (2 + (i >> 1)) * PIECE + h3_off(w_n, w_kb + 2 * (i & 1))) = rb[i];
    };

    const int KT = P.Kp / HBK;
    load_tile(0);
    store_tile(smem_raw);
    if (KT > 1) load_tile(HBK);
    lds_barrier();
    // lane (row fi, k half fh) of a 32-row fragment tile; 32-row offsets keep (row >> 2) & 3 == (fi >> 2) & 3
    const int sw = (fi >> 2) & 3;
    const int arow = (wm0 + fi) * HROW, brow = 2 * PIECE + (wn0 + fi) * HROW;
    for (int kt = 0; kt < KT; ++kt) {
        const char *cur = smem_raw + (kt & 1) * H3_BUF;
        char *nxt = smem_raw + ((kt & 1) ^ 1) * H3_BUF;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int coff = ((2 * s + fh) ^ sw) << 4;
            f16x8 a[TM][2], b[TN][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i][q] = *reinterpret_cast<const f16x8 *>(cur + arow + coff + q * PIECE + 32 * i * HROW);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j][q] = *reinterpret_cast<const f16x8 *>(cur + brow + coff + q * PIECE + 32 * j * HROW);
            }
            // term-major order: consecutive MFMAs write different accumulators
#define H3_TERM(ACC, QA, QB)                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)       \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][QA], b[j][QB], ACC[i][j], 0, 0, 0);
            H3_TERM(acc0, 0, 0)
            H3_TERM(acc1, 0, 1)
            if (s == 0) {
                if (kt + 1 < KT) store_tile(nxt);            // registers hold step kt+1
                if (kt + 2 < KT) load_tile((kt + 2) * HBK);
            }
            H3_TERM(acc1, 1, 0)
#undef H3_TERM
        }
        lds_barrier();       // buffer kt&1 fully read (it is overwritten in step kt+1), buffer (kt+1)&1 complete
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc0[i][j] += acc1[i][j];
    store_out<ACT, TM, TN, WN>(A, acc0, reinterpret_cast<float *>(smem_raw), m0 + wm0, n0 + wn0, wid, lane,
                               rinv + wm0, winv);
}

template <int ACT>
static int launch_linear_h3(const LinearH3Args &P, hipStream_t s) {
    static bool declared = false;
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_h3<ACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "linear_h3: cannot raise the dynamic LDS limit to %d bytes", H3_LDS);
        declared = true;
    }
    const int64_t blocks = (int64_t)((P.base.M + H3_BM - 1) / H3_BM) * ((P.base.N + H3_BN - 1) / H3_BN);
    hipLaunchKernelGGL(k_linear_h3<ACT>, dim3((unsigned)blocks), dim3(256), H3_LDS, s, P);
    return launch_status("linear_h3");
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_linear_h3(const float *x, int64_t ldx, const void *w_packed, float *y, int64_t ldy,
                               int32_t M, int32_t N, int32_t K, const float *bias, const float *rowvec,
                               const float *rowvec_w, int32_t act, const float *act_param,
                               const float *bn_scale, const float *bn_shift, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "linear_h3: bad sizes");
    if (M == 0 || N == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && w_packed && y, "linear_h3: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldy >= N, "linear_h3: leading dimension too small");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "linear_h3: bad activation id");
    BLGNN_CHECK_ARG((rowvec == nullptr) == (rowvec_w == nullptr), "linear_h3: rowvec and rowvec_w go together");
    BLGNN_CHECK_ARG(x != y, "linear_h3: x and y must not alias");
    if (K % 4 != 0 || ldx % 4 != 0 || !aligned16(x) || !aligned16(w_packed))
        return fail(BLGNN_ERR_UNSUPPORTED, "linear_h3: needs K %% 4 == 0, ldx %% 4 == 0 and 16-byte aligned x / packed "
                    "weights (got K=%d ldx=%lld); use blgnn_linear", K, (long long)ldx);
    LinearH3Args P;
    LinearArgs &A = P.base;
    A.x = x; A.ldx = ldx; A.w = nullptr; A.ldw = 0; A.y = y; A.ldy = ldy;
    A.M = M; A.N = N; A.K = K; A.rowvec = rowvec; A.rowvec_w = rowvec_w;
    A.epi = Epilogue{bias, act_param, bn_scale, bn_shift, act};
    A.vec_a = 1; A.vec_b = 0;
    P.wp = reinterpret_cast<const _Float16 *>(w_packed);
    P.Kp = (K + 31) / 32 * 32;
    hipStream_t s = (hipStream_t)stream;
    switch (act) {
        case BLGNN_ACT_RELU:    return launch_linear_h3<BLGNN_ACT_RELU>(P, s);
        case BLGNN_ACT_PRELU:   return launch_linear_h3<BLGNN_ACT_PRELU>(P, s);
        case BLGNN_ACT_SIGMOID: return launch_linear_h3<BLGNN_ACT_SIGMOID>(P, s);
        case BLGNN_ACT_TANH:    return launch_linear_h3<BLGNN_ACT_TANH>(P, s);
        default:                return launch_linear_h3<BLGNN_ACT_IDENTITY>(P, s);
    }
}
