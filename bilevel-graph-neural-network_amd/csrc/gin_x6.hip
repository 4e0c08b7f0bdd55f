// Fused GIN layer with the feature transform on the bf16 matrix cores at fp32 accuracy
// ("bf16x6": every fp32 operand is split into three bf16 pieces x = hi + mid + lo, 24 mantissa
// bits in total, and a product a*b is the six bf16 MFMAs hi*hi, hi*mid, mid*hi, hi*lo, lo*hi,
// mid*mid accumulated in fp32; the dropped terms are below 2^-24 |a||b|).  Measured error vs
// fp64 is at or below that of a sequential fp32 fma chain (6e-7 vs 1.2e-6 at K=128), so the
// 1e-5 parity bar holds, while v_mfma_f32_32x32x16_bf16 moves 16 k per 32 cycles against 2 k per
// 64 cycles for v_mfma_f32_32x32x2_f32: 6 bf16 MFMAs = 192 cycles per 16 k instead of 512.
// bf16 keeps the fp32 exponent range, so there is no overflow / underflow hazard (an fp16
// three-piece split would be twice as fast again but is only safe for |x| < 65504).
//
// Structure = gin_fused.hip with 32-row tiles: stage tile -> LDS gather -> [split] -> GEMM ->
// act -> [split] -> GEMM -> act + eval-BN -> row stores.  Operand pieces live in LDS as
// [row][k] bf16 with a 16-byte row pad (conflict-free ds_read_b128 fragments); weights arrive
// pre-split and packed [piece][K/8][N][8] (blgnn_pack_weight_x6) so a B fragment is one 16-byte
// load per lane.
#include "gin_common.hpp"
#include <stdlib.h>

namespace blgnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifdef X6_STAMP
__device__ unsigned long long g_x6_stamps[16];
#define XSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_x6_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XSTAMP(i)
#endif

constexpr int XCOLS = 256;       // column indices of a tile kept in LDS (32 rows x mean degree 2.1 = 68)

struct Split3 { __bf16 hi, mid, lo; };
__device__ __forceinline__ Split3 split3(float x) {
    Split3 s;
    s.hi = (__bf16)x;
    const float r1 = x - (float)s.hi;
    s.mid = (__bf16)r1;
    s.lo = (__bf16)(r1 - (float)s.mid);
    return s;
}

// ---- weight packing: W[K,N] fp32 -> [3][Kp/8][N][8] bf16 -----------------------------------------
__global__ void k_pack_x6(const float *__restrict__ w, int64_t ldw, int K, int N, int Kp,
                          __bf16 *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // (kb, n)
    const int kb = idx / N, n = idx - kb * N;
    if (kb >= Kp / 8) return;
    bf16x8 p[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kb * 8 + j;
        const Split3 s = split3(k < K ? w[(int64_t)k * ldw + n] : 0.f);
        p[0][j] = s.hi; p[1][j] = s.mid; p[2][j] = s.lo;
    }
    const size_t piece = (size_t)(Kp / 8) * N;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        *reinterpret_cast<bf16x8 *>(out + ((size_t)q * piece + (size_t)kb * N + n) * 8) = p[q];
}

struct GinX6Args {
    const int32_t *rowptr, *col;
    const float *x; int64_t ldx;
    int32_t n, K, Kp;                        // Kp = K padded to 32
    const __bf16 *w1p, *w2p;                 // packed pieces
    const float *b1, *b2;
    float self_coef;
    Epilogue epi;
    float *y; int64_t ldy;
};

// B fragments of one 32-deep chunk (two 16-k MFMA steps): [step][piece][nt]
template <int N, int NTW>
__device__ __forceinline__ void load_b_x6(bf16x8 (&dst)[2][3][NTW], const __bf16 *__restrict__ Wp, int Kp,
                                          int kc, int n_base, int fi, int fh) {
    const size_t piece = (size_t)(Kp / 8) * N;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int kb = kc / 8 + 2 * s + fh;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                dst[s][q][nt] = *reinterpret_cast<const bf16x8 *>(
                    Wp + ((size_t)q * piece + (size_t)kb * N + n_base + nt * 32 + fi) * 8);
    }
}

// acc0 += A_hi B_hi ; acc1 += the five cross terms.  A pieces: LDS [piece][XBM][sa bytes], row-major bf16
template <int N, int NTW, int MT, int XBM>
__device__ __forceinline__ void gemm_x6(f32x16 (&acc0)[MT][NTW], f32x16 (&acc1)[MT][NTW], const char *Ap, int sa,
                                        const __bf16 *__restrict__ Wp, int Kp, int n_base, int fi, int fh,
                                        bf16x8 (&bcur)[2][3][NTW]) {
    bf16x8 bnxt[2][3][NTW];
    const int piece_bytes = XBM * sa;
    const char *arow = Ap + fi * sa + fh * 16;                   // lane (row fi, k half fh)
    for (int kc = 0; kc < Kp; kc += KC) {
        const bool more = kc + KC < Kp;
        if (more) load_b_x6<N, NTW>(bnxt, Wp, Kp, kc + KC, n_base, fi, fh);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                bf16x8 a[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    a[q] = *reinterpret_cast<const bf16x8 *>(arow + mt * 32 * sa + q * piece_bytes + (kc + 16 * s) * 2);
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    acc0[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bcur[s][0][nt], acc0[mt][nt], 0, 0, 0);
                    acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bcur[s][1][nt], acc1[mt][nt], 0, 0, 0);
                    acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bcur[s][0][nt], acc1[mt][nt], 0, 0, 0);
                    acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bcur[s][2][nt], acc1[mt][nt], 0, 0, 0);
                    acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bcur[s][0][nt], acc1[mt][nt], 0, 0, 0);
                    acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bcur[s][1][nt], acc1[mt][nt], 0, 0, 0);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int nt = 0; nt < NTW; ++nt) bcur[s][q][nt] = bnxt[s][q][nt];
        }
    }
}

template <int N, int LG, int ACT, int XBM>
__global__ void __launch_bounds__(256, XBM == 32 ? 2 : 1) k_gin_layer_x6(const GinX6Args A) {
    constexpr int NTW = N / 128;                    // 4 waves side by side over N
    constexpr int MT = XBM / 32;
    constexpr int RPW = WAVE / LG;
    constexpr int PASSES = XBM / (4 * RPW);
    constexpr int LDO = N + 4;
    constexpr int SH = N * 2 + 16;                  // hidden piece row stride (bytes)
    extern __shared__ float4 lds4[];
    char *lds = reinterpret_cast<char *>(lds4);
    const int sa = A.Kp * 2 + 16;                   // aggregated piece row stride (bytes)
    // region 0: aggregated pieces [3][XBM][sa], later the fp32 output staging tile [XBM][N+4]
    // region 1: staged fp32 input rows [XBM][K], later the hidden pieces [3][XBM][SH]
    int r0 = 3 * XBM * sa;
    if (XBM * LDO * 4 > r0) r0 = XBM * LDO * 4;
    int r1 = XBM * A.K * 4;
    if (3 * XBM * SH > r1) r1 = 3 * XBM * SH;
    char *Ap = lds;
    float *Ys = reinterpret_cast<float *>(lds);
    float *Xs = reinterpret_cast<float *>(lds + r0);
    char *Hp = lds + r0;
    int *rp = reinterpret_cast<int *>(lds + r0 + r1);       // [XBM+1]
    int *cs = rp + XBM + 4;                                 // [XCOLS]
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int sub = lane / LG, l = lane % LG;
    const int m0 = blockIdx.x * XBM;
    const int n_base = wid * (NTW * 32);
    const int k4n = A.K >> 2;
    const bool ok = l < k4n;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;

    XSTAMP(0);
    // ---- (0) requests: rowptr slice, first weight chunk, feature rows, column indices -------------
    int rpv = 0;
    if (threadIdx.x <= XBM) {
        const int r = m0 + threadIdx.x;
        rpv = A.rowptr[r < A.n ? r : A.n];
    }
    bf16x8 bfrag[2][3][NTW];
    load_b_x6<N, NTW>(bfrag, A.w1p, A.Kp, 0, n_base, fi, fh);
    float4 xin[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = m0 + (p * 4 + wid) * RPW + sub;
        xin[p] = (ok && row < A.n) ? X[(int64_t)row * ldx4 + l] : f4_zero();
    }
    if (threadIdx.x <= XBM) rp[threadIdx.x] = rpv;
    __syncthreads();
    XSTAMP(1);
    const int e_lo = rp[0];
    const int ncol_all = rp[XBM] - e_lo;
    const bool cols_in_lds = ncol_all <= XCOLS;
    {
        int cv[XCOLS / 256];
#pragma unroll
        for (int i = 0; i < XCOLS / 256; ++i) {
            const int idx = threadIdx.x + 256 * i;
            cv[i] = (cols_in_lds && idx < ncol_all) ? A.col[e_lo + idx] : 0;
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * 4 + wid) * RPW + sub;
            if (ok) *reinterpret_cast<float4 *>(Xs + m * A.K + 4 * l) = xin[p];
        }
#pragma unroll
        for (int i = 0; i < XCOLS / 256; ++i) {
            const int idx = threadIdx.x + 256 * i;
            if (cols_in_lds && idx < ncol_all) cs[idx] = cv[i];
        }
    }
    // zero the K padding columns of the three aggregated pieces
    for (int idx = threadIdx.x; idx < 3 * XBM * (A.Kp - A.K); idx += 256) {
        const int per = XBM * (A.Kp - A.K);
        const int q = idx / per, rem = idx - q * per;
        const int m = rem / (A.Kp - A.K), k = A.K + rem % (A.Kp - A.K);
        *reinterpret_cast<__bf16 *>(Ap + q * XBM * sa + m * sa + k * 2) = (__bf16)0.f;
    }
    __syncthreads();
    XSTAMP(2);

    // ---- (1) gather from LDS, split, park the three pieces --------------------------------------------
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * 4 + wid) * RPW + sub;
        const int row = m0 + m;
        const int e0 = rp[m], cnt = rp[m + 1] - e0;
        int j[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int ci = e0 + q - e_lo;
            ci = (q < cnt && ci < XCOLS) ? ci : 0;
            const int c = cs[ci];
            j[q] = (q < cnt && cols_in_lds) ? c : row;
        }
        float4 accv = xin[p];
        accv.x *= A.self_coef; accv.y *= A.self_coef; accv.z *= A.self_coef; accv.w *= A.self_coef;
        bool far[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jl = j[q] - m0;
            const bool in_tile = jl >= 0 && jl < XBM;
            far[q] = !in_tile;
            const int src = (in_tile && j[q] != row) ? jl : m;
            const float4 u = *reinterpret_cast<const float4 *>(Xs + src * A.K + (ok ? 4 * l : 0));
            f4_fma(accv, (in_tile && j[q] != row) ? 1.f : 0.f, u);
        }
        if (ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (far[q]) f4_add(accv, X[(int64_t)j[q] * ldx4 + l]);
            const int first = cols_in_lds ? 4 : 0;
            for (int e = e0 + first; e < e0 + cnt; ++e) {
                const int jj = A.col[e];
                if (jj != row) f4_add(accv, X[(int64_t)jj * ldx4 + l]);
            }
            const Split3 s0 = split3(accv.x), s1 = split3(accv.y), s2 = split3(accv.z), s3 = split3(accv.w);
            bf16x4 h = {s0.hi, s1.hi, s2.hi, s3.hi}, md = {s0.mid, s1.mid, s2.mid, s3.mid},
                   lo = {s0.lo, s1.lo, s2.lo, s3.lo};
            char *dst = Ap + m * sa + 8 * l;
            *reinterpret_cast<bf16x4 *>(dst) = h;
            *reinterpret_cast<bf16x4 *>(dst + XBM * sa) = md;
            *reinterpret_cast<bf16x4 *>(dst + 2 * XBM * sa) = lo;
        }
    }
    __syncthreads();
    XSTAMP(3);

    // ---- (2) hidden = act(agg @ W1 + b1), split, park ---------------------------------------------------
    f32x16 acc0[MT][NTW], acc1[MT][NTW];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[mt][nt][r] = 0.f; acc1[mt][nt][r] = 0.f; }
    };
    zero();
    gemm_x6<N, NTW, MT, XBM>(acc0, acc1, Ap, sa, A.w1p, A.Kp, n_base, fi, fh, bfrag);
    XSTAMP(4);
    load_b_x6<N, NTW>(bfrag, A.w2p, N, 0, n_base, fi, fh);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = n_base + nt * 32 + fi;
        const float b = A.b1 ? A.b1[c] : 0.f;
        const float sl = (ACT == BLGNN_ACT_PRELU && A.epi.act_param) ? A.epi.act_param[c] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            const Split3 s = split3(act_t<ACT>(acc0[mt][nt][r] + acc1[mt][nt][r] + b, sl));
            char *dst = Hp + m * SH + c * 2;
            *reinterpret_cast<__bf16 *>(dst) = s.hi;
            *reinterpret_cast<__bf16 *>(dst + XBM * SH) = s.mid;
            *reinterpret_cast<__bf16 *>(dst + 2 * XBM * SH) = s.lo;
        }
    }
    __syncthreads();
    XSTAMP(5);

    // ---- (3) y = affine(act(hidden @ W2 + b2)) -> LDS staging -> row stores -------------------------------
    zero();
    gemm_x6<N, NTW, MT, XBM>(acc0, acc1, Hp, SH, A.w2p, N, n_base, fi, fh, bfrag);
    XSTAMP(6);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = n_base + nt * 32 + fi;
        const float b = A.b2 ? A.b2[c] : 0.f;
        const float sl = (ACT == BLGNN_ACT_PRELU && A.epi.act_param) ? A.epi.act_param[c] : 0.f;
        const float sc = A.epi.bn_scale ? A.epi.bn_scale[c] : 1.f;
        const float sh = (A.epi.bn_scale && A.epi.bn_shift) ? A.epi.bn_shift[c] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            float v = act_t<ACT>(acc0[mt][nt][r] + acc1[mt][nt][r] + b, sl);
            if (A.epi.bn_scale) v = fmaf(v, sc, sh);
            Ys[m * LDO + c] = v;
        }
    }
    __syncthreads();
    XSTAMP(7);
    constexpr int N4 = N / 4;
    const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);
    for (int idx = threadIdx.x; idx < XBM * N4; idx += 256) {
        const int m = idx / N4, c4 = idx % N4;
        const int row = m0 + m;
        if (row >= A.n) continue;
        const float4 v = *reinterpret_cast<const float4 *>(Ys + m * LDO + 4 * c4);
        float *dst = A.y + (int64_t)row * A.ldy + 4 * c4;
        if (vec_out) *reinterpret_cast<float4 *>(dst) = v;
        else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
    }
    XSTAMP(8);
}

static size_t x6_lds_bytes(int K, int Kp, int N, int XBM) {
    size_t r0 = (size_t)3 * XBM * (Kp * 2 + 16);
    if ((size_t)XBM * (N + 4) * 4 > r0) r0 = (size_t)XBM * (N + 4) * 4;
    size_t r1 = (size_t)XBM * K * 4;
    if ((size_t)3 * XBM * (N * 2 + 16) > r1) r1 = (size_t)3 * XBM * (N * 2 + 16);
    return r0 + r1 + sizeof(int) * (XBM + 4 + XCOLS);
}

template <int N, int LG, int ACT, int XBM>
static int launch_x6(const GinX6Args &A, hipStream_t s) {
    static size_t declared = 0;
    const size_t bytes = x6_lds_bytes(A.K, A.Kp, N, XBM);
    if (bytes > declared && bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gin_layer_x6<N, LG, ACT, XBM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gin_layer_x6: cannot raise the dynamic LDS limit to %zu bytes", bytes);
        declared = bytes;
    }
    const int blocks = (A.n + XBM - 1) / XBM;
    hipLaunchKernelGGL((k_gin_layer_x6<N, LG, ACT, XBM>), dim3(blocks), dim3(256), bytes, s, A);
    return launch_status("gin_layer_x6");
}

template <int N, int LG, int XBM>
static int x6_pick_act(const GinX6Args &A, hipStream_t s) {
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    return launch_x6<N, LG, BLGNN_ACT_RELU, XBM>(A, s);
        case BLGNN_ACT_PRELU:   return launch_x6<N, LG, BLGNN_ACT_PRELU, XBM>(A, s);
        case BLGNN_ACT_SIGMOID: return launch_x6<N, LG, BLGNN_ACT_SIGMOID, XBM>(A, s);
        case BLGNN_ACT_TANH:    return launch_x6<N, LG, BLGNN_ACT_TANH, XBM>(A, s);
        default:                return launch_x6<N, LG, BLGNN_ACT_IDENTITY, XBM>(A, s);
    }
}

template <int N, int XBM>
static int x6_pick_lg(const GinX6Args &A, hipStream_t s) {
    const int k4 = A.K / 4;
    if (k4 <= 16) return x6_pick_act<N, 16, XBM>(A, s);
    if (k4 <= 32) return x6_pick_act<N, 32, XBM>(A, s);
    return x6_pick_act<N, 64, XBM>(A, s);
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int64_t blgnn_packed_weight_x6_elems(int32_t K, int32_t N) {
    const int64_t Kp = ((int64_t)K + 31) / 32 * 32;
    return 3 * Kp * N;                                    // bf16 elements
}

extern "C" int blgnn_pack_weight_x6(const float *w, int64_t ldw, int32_t K, int32_t N, void *out,
                                    void *stream) {
    BLGNN_CHECK_ARG(w && out && K > 0 && N > 0 && ldw >= N, "pack_weight_x6: bad args");
    const int Kp = (K + 31) / 32 * 32;
    const int64_t items = (int64_t)(Kp / 8) * N;
    hipLaunchKernelGGL(k_pack_x6, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw,
                       K, N, Kp, reinterpret_cast<__bf16 *>(out));
    return launch_status("pack_weight_x6");
}

extern "C" int blgnn_gin_layer_x6(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x,
                                  int64_t ldx, int32_t n, int32_t K, const void *w1_packed, const float *b1,
                                  const void *w2_packed, const float *b2, int32_t N, float eps, int32_t act,
                                  const float *act_param, const float *bn_scale, const float *bn_shift,
                                  float *y, int64_t ldy, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && K > 0 && N > 0 && nnz >= 0, "gin_layer_x6: bad sizes");
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(rowptr && (col || nnz == 0) && x && w1_packed && w2_packed && y, "gin_layer_x6: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldy >= N, "gin_layer_x6: leading dimension too small");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "gin_layer_x6: bad activation id");
    BLGNN_CHECK_ARG(x != y, "gin_layer_x6: x and y must not alias");
    if (!(N == 128 || N == 256) || K % 4 != 0 || K > 256 || ldx % 4 != 0 || !aligned16(x) ||
        !aligned16(w1_packed) || !aligned16(w2_packed))
        return fail(BLGNN_ERR_UNSUPPORTED, "gin_layer_x6: needs N in {128,256}, K %% 4 == 0, K <= 256 and 16-byte "
                    "aligned x / packed weights (got N=%d K=%d); use blgnn_gin_layer", N, K);
    GinX6Args A;
    A.rowptr = rowptr; A.col = col; A.x = x; A.ldx = ldx; A.n = n; A.K = K; A.Kp = (K + 31) / 32 * 32;
    A.w1p = reinterpret_cast<const __bf16 *>(w1_packed); A.w2p = reinterpret_cast<const __bf16 *>(w2_packed);
    A.b1 = b1; A.b2 = b2; A.self_coef = 1.f + eps;
    A.epi = Epilogue{b2, act_param, bn_scale, bn_shift, act};
    A.y = y; A.ldy = ldy;
    hipStream_t s = (hipStream_t)stream;
    // 32-row tiles: 64-row tiles halve the weight-fragment traffic per row but need 104 KiB of
    // LDS (one workgroup per CU) and measured 1.4x slower
    if (N == 256) return x6_pick_lg<256, 32>(A, s);
    return x6_pick_lg<128, 32>(A, s);
}
