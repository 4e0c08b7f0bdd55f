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
#include "x6_common.hpp"
#include <stdlib.h>

namespace blgnn {

#ifdef X6_STAMP
__device__ unsigned long long g_x6_stamps[16];
#define XSTAMP(i) do { if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_x6_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XSTAMP(i)
#endif

constexpr int XCOLS = 256;       // column indices of a tile kept in LDS (32 rows x mean degree 2.1 = 68)

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

// Weight fragments of one 32-deep chunk (two 16-k MFMA steps): [step][piece][nt].  Lane (fi, fh)
// holds the 8 consecutive k of output column n_base + 32 nt + fi: the MFMA *A* operand (rows = output
// columns), so the accumulators come out transposed -- see gemm_x6.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t wrsrc_t;

// Buffer descriptor over a packed weight matrix (built from kernel arguments only: wave-uniform).
__device__ __forceinline__ wrsrc_t weight_rsrc(const __bf16 *Wp, int Kp, int N) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(Wp), 0, 3 * Kp * N * 2, 0x00020000);
}

// Per-lane byte offset of a fragment load inside a [K/8][N][8] piece: lane (fi, fh) -> (kb = fh, n = n_base + fi).
template <int N>
__device__ __forceinline__ unsigned weight_lane_off(int n_base, int fi, int fh) {
    return ((unsigned)fh * N + (unsigned)(n_base + fi)) * 16u;
}

template <int N, int NTW>
__device__ __forceinline__ void load_b_x6(bf16x8 (&dst)[2][3][NTW], wrsrc_t W, int Kp, int kc, unsigned lane_off) {
    // buffer loads: address = descriptor base + one per-lane VGPR offset + a wave-uniform SGPR offset
    // (piece, chunk, step), so the 6 NTW loads of a chunk share a single address register -- with flat
    // 64-bit addresses hipcc hoists one address pair per load out of the tile loop and spills them
    const unsigned piece = (unsigned)Kp * N * 2u;                 // bytes
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const unsigned soff = (unsigned)q * piece + (unsigned)(kc / 8 + 2 * s) * (N * 16u) + nt * 32u * 16u;
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(W, lane_off, soff, 0);
                dst[s][q][nt] = __builtin_bit_cast(bf16x8, v);
            }
}

// The six products of one 32-deep chunk.  The GEMM runs transposed, D^T = W^T . act^T: the weight
// fragment is the MFMA A operand and the activation fragment (LDS [piece][XBM][sa bytes], row-major
// bf16) the B operand, so a lane ends up with FOUR CONSECUTIVE output columns of ONE tile row
// (n = n_base + 32 nt + 8 (r >> 2) + 4 fh + (r & 3), m = 32 mt + fi): the epilogues split / pack
// four values at a time and write 8- and 16-byte LDS words instead of 2- and 4-byte ones.
// acc0 += hi.hi ; acc1 += the five cross terms.
template <int NTW, int MT, int XBM>
__device__ __forceinline__ void mfma_chunk_x6(f32x16 (&acc0)[MT][NTW], f32x16 (&acc1)[MT][NTW], const char *arow,
                                              int sa, int kc, const bf16x8 (&w)[2][3][NTW]) {
    const int piece_bytes = XBM * sa;
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
                acc0[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][0][nt], a[0], acc0[mt][nt], 0, 0, 0);
                acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][1][nt], a[0], acc1[mt][nt], 0, 0, 0);
                acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][0][nt], a[1], acc1[mt][nt], 0, 0, 0);
                acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][2][nt], a[0], acc1[mt][nt], 0, 0, 0);
                acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][0][nt], a[2], acc1[mt][nt], 0, 0, 0);
                acc1[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s][1][nt], a[1], acc1[mt][nt], 0, 0, 0);
            }
        }
    }
}

// Streaming variant (wide layers): weight chunks double-buffered in registers, one chunk ahead.
// Weight chunks stream through a ring of RING register buffers, RING-1 chunks ahead of the MFMAs
// (an L2 round trip is ~1.5 k cycles under load, one chunk of MFMAs ~0.4 k).  On entry the caller has
// requested chunks 0 .. RING-2 into w[0] .. w[RING-2] (x6_prefetch).
template <int N, int NTW, int RING>
__device__ __forceinline__ void x6_prefetch(bf16x8 (&w)[RING][2][3][NTW], wrsrc_t Wp, int Kp, unsigned lane_off) {
#pragma unroll
    for (int u = 0; u < RING - 1; ++u)
        if (u * KC < Kp) load_b_x6<N, NTW>(w[u], Wp, Kp, u * KC, lane_off);
}

template <int N, int NTW, int MT, int XBM, int RING>
__device__ __forceinline__ void gemm_x6(f32x16 (&acc0)[MT][NTW], f32x16 (&acc1)[MT][NTW], const char *Ap, int sa,
                                        wrsrc_t Wp, int Kp, unsigned lane_off, int fi, int fh,
                                        bf16x8 (&w)[RING][2][3][NTW]) {
    const char *arow = Ap + fi * sa + fh * 16;                   // lane (row fi, k half fh)
    for (int kc = 0; kc < Kp; kc += RING * KC) {
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            const int k = kc + u * KC;
            if (k < Kp) {
                if (k + (RING - 1) * KC < Kp)
                    load_b_x6<N, NTW>(w[(u + RING - 1) % RING], Wp, Kp, k + (RING - 1) * KC, lane_off);
                mfma_chunk_x6<NTW, MT, XBM>(acc0, acc1, arow, sa, k, w[u]);
            }
        }
    }
}

template <int N, int LG, int ACT, int XBM>
__global__ void __launch_bounds__(256, (XBM == 32 && N == 128) ? 3 : 1) k_gin_layer_x6(const GinX6Args A) {
    constexpr int NTW = N / 128;                    // 4 waves side by side over N
    constexpr int MT = XBM / 32;
    constexpr int RPW = WAVE / LG;
    constexpr int PASSES = XBM / (4 * RPW);
    constexpr int LDO = N + 4;
    constexpr int SH = N * 2 + 16;                  // hidden piece row stride (bytes)
    constexpr int NCV = XCOLS / 256;
    constexpr int N4 = N / 4;
    constexpr int RING = NTW == 1 ? 3 : 2;          // weight chunk buffers (24 NTW VGPRs each)
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
    const int n_base = wid * (NTW * 32);
    const wrsrc_t W1 = weight_rsrc(A.w1p, A.Kp, N), W2 = weight_rsrc(A.w2p, N, N);
    const unsigned lane_off = weight_lane_off<N>(n_base, fi, fh);
    const int k4n = A.K >> 2;
    const bool ok = l < k4n;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;
    const int m0 = blockIdx.x * XBM;
    const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);

    XSTAMP(0);
    // ---- (0) requests: rowptr slice, feature rows, first weight chunk, edge range -> column indices ---------
    int rpv = 0;
    if (threadIdx.x <= XBM) {
        const int r = m0 + threadIdx.x;
        rpv = A.rowptr[r < A.n ? r : A.n];
    }
    float4 xin[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = m0 + (p * 4 + wid) * RPW + sub;
        xin[p] = (ok && row < A.n) ? X[(int64_t)row * ldx4 + l] : f4_zero();
    }
    bf16x8 wfrag[RING][2][3][NTW];
    x6_prefetch<N, NTW, RING>(wfrag, W1, A.Kp, lane_off);
    const int e_lo = A.rowptr[m0];                                   // uniform addresses: scalar loads
    const int ncol_all = A.rowptr[m0 + XBM < A.n ? m0 + XBM : A.n] - e_lo;
    const bool cols_in_lds = ncol_all <= XCOLS;
    int cv[NCV];
#pragma unroll
    for (int i = 0; i < NCV; ++i) {
        const int idx = threadIdx.x + 256 * i;
        cv[i] = (cols_in_lds && idx < ncol_all) ? A.col[e_lo + idx] : 0;
    }
    // park the tile's inputs in LDS
    if (threadIdx.x <= XBM) rp[threadIdx.x] = rpv;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * 4 + wid) * RPW + sub;
        if (ok) *reinterpret_cast<float4 *>(Xs + m * A.K + 4 * l) = xin[p];
    }
#pragma unroll
    for (int i = 0; i < NCV; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (cols_in_lds && idx < ncol_all) cs[idx] = cv[i];
    }
    // zero the K padding columns of the three aggregated pieces
    for (int idx = threadIdx.x; idx < 3 * XBM * (A.Kp - A.K); idx += 256) {
        const int per = XBM * (A.Kp - A.K);
        const int q = idx / per, rem = idx - q * per;
        const int m = rem / (A.Kp - A.K), k = A.K + rem % (A.Kp - A.K);
        *reinterpret_cast<__bf16 *>(Ap + q * XBM * sa + m * sa + k * 2) = (__bf16)0.f;
    }
    lds_barrier();
    XSTAMP(1); XSTAMP(2);

    // ---- (1) gather from LDS, split, park the three pieces --------------------------------------------------
    // Neighbours outside the tile (molecules straddle tile borders) come from global memory: the loads of
    // two passes are requested together, after the LDS part, so the tile pays one L2 round trip per two
    // passes instead of one per pass (four at once would need 64 more VGPRs: spills at 3 workgroups/CU).
    constexpr int FB = PASSES < 2 ? PASSES : 2;
#pragma unroll
    for (int pg = 0; pg < PASSES; pg += FB) {
        float4 accv[FB];
        int far[FB][4];
#pragma unroll
        for (int pp = 0; pp < FB; ++pp) {
            const int m = ((pg + pp) * 4 + wid) * RPW + sub;
            const int row = m0 + m;
            const int e0 = rp[m], cnt = rp[m + 1] - e0;
            const int lo4 = ok ? 4 * l : 0;
            accv[pp] = *reinterpret_cast<const float4 *>(Xs + m * A.K + lo4);
            accv[pp].x *= A.self_coef; accv[pp].y *= A.self_coef;
            accv[pp].z *= A.self_coef; accv[pp].w *= A.self_coef;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int ci = e0 + q - e_lo;
                ci = (q < cnt && ci < XCOLS) ? ci : 0;
                const int c = cs[ci];
                const int j = (q < cnt && cols_in_lds) ? c : row;
                const int jl = j - m0;
                const bool in_tile = jl >= 0 && jl < XBM;
                far[pp][q] = (in_tile || !ok) ? -1 : j;
                const bool use = in_tile && j != row;
                const float4 u = *reinterpret_cast<const float4 *>(Xs + (use ? jl : m) * A.K + lo4);
                f4_fma(accv[pp], use ? 1.f : 0.f, u);
            }
        }
        float4 fu[FB][4];
#pragma unroll
        for (int pp = 0; pp < FB; ++pp)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fu[pp][q] = f4_zero();
                if (far[pp][q] >= 0) fu[pp][q] = X[(int64_t)far[pp][q] * ldx4 + l];
            }
#pragma unroll
        for (int pp = 0; pp < FB; ++pp) {
            const int m = ((pg + pp) * 4 + wid) * RPW + sub;
            const int row = m0 + m;
#pragma unroll
            for (int q = 0; q < 4; ++q) f4_add(accv[pp], fu[pp][q]);
            if (ok) {
                const int e0 = rp[m], cnt = rp[m + 1] - e0;
                const int first = cols_in_lds ? 4 : 0;
                for (int e = e0 + first; e < e0 + cnt; ++e) {
                    const int jj = A.col[e];
                    if (jj != row) f4_add(accv[pp], X[(int64_t)jj * ldx4 + l]);
                }
                const Split3 s0 = split3(accv[pp].x), s1 = split3(accv[pp].y), s2 = split3(accv[pp].z),
                             s3 = split3(accv[pp].w);
                bf16x4 h = {s0.hi, s1.hi, s2.hi, s3.hi}, md = {s0.mid, s1.mid, s2.mid, s3.mid},
                       lo = {s0.lo, s1.lo, s2.lo, s3.lo};
                char *dst = Ap + m * sa + 8 * l;
                *reinterpret_cast<bf16x4 *>(dst) = h;
                *reinterpret_cast<bf16x4 *>(dst + XBM * sa) = md;
                *reinterpret_cast<bf16x4 *>(dst + 2 * XBM * sa) = lo;
            }
        }
    }
    lds_barrier();
    XSTAMP(3);

    // ---- (2) hidden = act(agg @ W1 + b1), split, park -----------------------------------------------------
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
    // epilogue parameters are requested BEFORE the GEMM they follow, so their L2 round trip runs under
    // the MFMAs (a lane owns the four consecutive columns c .. c+3 of each of its 4 NTW groups)
    float4 pb[NTW][4], psl[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n_base + nt * 32 + 8 * g + 4 * fh;
            pb[nt][g] = A.b1 ? *reinterpret_cast<const float4 *>(A.b1 + c) : f4_zero();
            psl[nt][g] = (ACT == BLGNN_ACT_PRELU && A.epi.act_param)
                             ? *reinterpret_cast<const float4 *>(A.epi.act_param + c) : f4_zero();
        }
    gemm_x6<N, NTW, MT, XBM, RING>(acc0, acc1, Ap, sa, W1, A.Kp, lane_off, fi, fh, wfrag);
    x6_prefetch<N, NTW, RING>(wfrag, W2, N, lane_off);
    XSTAMP(4);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n_base + nt * 32 + 8 * g + 4 * fh;             // four consecutive output columns
            const float4 b = pb[nt][g], sl = psl[nt][g];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 32 + fi;
                const Split3 s0 = split3(act_t<ACT>(acc0[mt][nt][4 * g + 0] + acc1[mt][nt][4 * g + 0] + b.x, sl.x));
                const Split3 s1 = split3(act_t<ACT>(acc0[mt][nt][4 * g + 1] + acc1[mt][nt][4 * g + 1] + b.y, sl.y));
                const Split3 s2 = split3(act_t<ACT>(acc0[mt][nt][4 * g + 2] + acc1[mt][nt][4 * g + 2] + b.z, sl.z));
                const Split3 s3 = split3(act_t<ACT>(acc0[mt][nt][4 * g + 3] + acc1[mt][nt][4 * g + 3] + b.w, sl.w));
                bf16x4 h = {s0.hi, s1.hi, s2.hi, s3.hi}, md = {s0.mid, s1.mid, s2.mid, s3.mid},
                       lo = {s0.lo, s1.lo, s2.lo, s3.lo};
                char *dst = Hp + m * SH + c * 2;
                *reinterpret_cast<bf16x4 *>(dst) = h;
                *reinterpret_cast<bf16x4 *>(dst + XBM * SH) = md;
                *reinterpret_cast<bf16x4 *>(dst + 2 * XBM * SH) = lo;
            }
        }
    lds_barrier();
    XSTAMP(5);

    // ---- (3) y = affine(act(hidden @ W2 + b2)) -> LDS staging -> row stores ---------------------------------
    zero();
    float4 psc[NTW][4], psh[NTW][4];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n_base + nt * 32 + 8 * g + 4 * fh;
            pb[nt][g] = A.b2 ? *reinterpret_cast<const float4 *>(A.b2 + c) : f4_zero();
            psc[nt][g] = float4{1.f, 1.f, 1.f, 1.f};
            psh[nt][g] = f4_zero();
            if (A.epi.bn_scale) psc[nt][g] = *reinterpret_cast<const float4 *>(A.epi.bn_scale + c);
            if (A.epi.bn_scale && A.epi.bn_shift) psh[nt][g] = *reinterpret_cast<const float4 *>(A.epi.bn_shift + c);
        }
    gemm_x6<N, NTW, MT, XBM, RING>(acc0, acc1, Hp, SH, W2, N, lane_off, fi, fh, wfrag);
    XSTAMP(6);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n_base + nt * 32 + 8 * g + 4 * fh;
            const float4 b = pb[nt][g], sl = psl[nt][g], sc = psc[nt][g], sh = psh[nt][g];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 32 + fi;
                float4 v;
                v.x = act_t<ACT>(acc0[mt][nt][4 * g + 0] + acc1[mt][nt][4 * g + 0] + b.x, sl.x);
                v.y = act_t<ACT>(acc0[mt][nt][4 * g + 1] + acc1[mt][nt][4 * g + 1] + b.y, sl.y);
                v.z = act_t<ACT>(acc0[mt][nt][4 * g + 2] + acc1[mt][nt][4 * g + 2] + b.z, sl.z);
                v.w = act_t<ACT>(acc0[mt][nt][4 * g + 3] + acc1[mt][nt][4 * g + 3] + b.w, sl.w);
                if (A.epi.bn_scale) {
                    v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                    v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                }
                *reinterpret_cast<float4 *>(Ys + m * LDO + c) = v;
            }
        }
    lds_barrier();
    XSTAMP(7);
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
