// One GIN layer in one launch (reference: model/layers.py:26-31,51-57):
//
//   y = affine( act( act( agg @ W1 + b1 ) @ W2 + b2 ) ),   agg_i = (1+eps) x_i + sum_{j in N(i), j != i} x_j
//
// A workgroup owns 64 consecutive node rows.
//  (0) The tile's feature rows, its rowptr slice and its CSR column indices are staged into
//      LDS with batches of independent, fully coalesced loads (16 B/lane) -- the only HBM
//      reads of the layer.  The first weight chunk is requested before anything else.
//  (1) CSR edge-gather FROM LDS: the merged molecular graph is block diagonal, so ~95 % of
//      a row's neighbours are rows of the same tile; the few that belong to a molecule
//      straddling the tile edge are read from L2.  Lane groups reduce their rows in CSR
//      order (deterministic); the aggregated tile lands in LDS, never in HBM.
//  (2) agg @ W1 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32): A fragments from the LDS
//      tile (row-major, odd stride -> every ds_read_b32 conflict free), B fragments straight
//      from the L2-resident weight matrix, register double-buffered one 32-deep K chunk
//      ahead.  K is padded to a multiple of 32 with zero columns so the MFMA loop is
//      branch free.
//  (3) hidden activation -> LDS, (4) second GEMM, (5) act + eval-BatchNorm, transposed
//      through LDS so that the layer's only HBM write is whole rows, 16 B per lane.
// HBM traffic = read x once + write y once (the algorithmic bytes of the SpMM alone); the
// two GEMMs ride on top until the fp32 MFMA roof.
#include "gin_common.hpp"
#include <stdlib.h>

namespace blgnn {

struct GinArgs {
    const int32_t *rowptr, *col;
    const float *x; int64_t ldx;
    int32_t n, K;
    const float *w1, *b1, *w2, *b2;
    float self_coef;
    Epilogue epi;           // bias = b2
    float *y; int64_t ldy;
};

#ifdef GIN_STAMP
__device__ unsigned long long g_gin_stamps[16];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_gin_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i)
#endif

static int g_gin_small_rows = 1 << 18;   // below this many rows use 32-row tiles
template <int N, int WCOLS, int LG, int ACT, int GBM>
__global__ void __launch_bounds__(256, (GBM == 32 && N <= 128) ? 4 : 2) k_gin_layer(const GinArgs A) {
    constexpr int WROWS = 4 / WCOLS;
    constexpr int MT = GBM / 32 / WROWS;
    constexpr int NTW = N / 32 / WCOLS;
    constexpr int RPW = WAVE / LG;                 // rows per wave pass (gather / staging)
    constexpr int PASSES = GBM / (4 * RPW);
    constexpr int LDH = N + 1;                     // hidden tile stride (odd)
    constexpr int LDO = N + 4;                     // output staging stride (16 B aligned rows)
    extern __shared__ float lds[];
    const int Kpad = (A.K + KC - 1) / KC * KC;
    const int lda = Kpad + 1;
    // region 0: aggregated tile [GBM][Kpad+1], later the output staging tile [GBM][N+4]
    // region 1: staged input rows [GBM][K], later the hidden tile [GBM][N+1]
    const int r0 = GBM * (lda > LDO ? lda : LDO);
    const int r1 = GBM * (A.K > LDH ? A.K : LDH);
    float *As = lds;
    float *Ys = lds;
    float *Hs = lds + r0;
    float *Xs = Hs;
    int *rp = reinterpret_cast<int *>(Hs + r1);    // [GBM+1] rowptr slice
    int *cs = rp + GBM + 4;                        // [GCOLS] column indices of the tile
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int sub = lane / LG, l = lane % LG;
    const int m0 = blockIdx.x * GBM;
    const int m_base = (wid / WCOLS) * (MT * 32), n_base = (wid % WCOLS) * (NTW * 32);
    const int k4n = A.K >> 2;
    const bool ok = l < k4n;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;

    STAMP(0);
#ifdef GIN_STAGGER
    // co-resident workgroups otherwise run in lockstep (both in their MFMA phase, then both in
    // their memory phase); offset the second dispatch round by about half a tile
    if (((blockIdx.x >> 3) >> 5) & 1) { for (int i = 0; i < GIN_STAGGER; ++i) __builtin_amdgcn_s_sleep(127); }
#endif
    // ---- (0) requests, oldest first: rowptr slice, weights chunk 0, feature rows -------------
    int rpv = 0;
    if (threadIdx.x <= GBM) {
        const int r = m0 + threadIdx.x;
        rpv = A.rowptr[r < A.n ? r : A.n];
    }
    float bfrag[KC / 2][NTW];
    load_b_chunk<N, NTW>(bfrag, A.w1, A.K, 0, n_base, fi, fh);
    float4 xin[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = m0 + (p * 4 + wid) * RPW + sub;
        xin[p] = (ok && row < A.n) ? X[(int64_t)row * ldx4 + l] : f4_zero();
    }
    if (threadIdx.x <= GBM) rp[threadIdx.x] = rpv;
    __syncthreads();
    const int e_lo = rp[0];
    int ncol = rp[GBM] - e_lo;
    ncol = ncol < GCOLS ? ncol : GCOLS;
    int cv[GCOLS / 256];
#pragma unroll
    for (int i = 0; i < GCOLS / 256; ++i) {
        const int idx = threadIdx.x + 256 * i;
        cv[i] = idx < ncol ? A.col[e_lo + idx] : 0;
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * 4 + wid) * RPW + sub;
        if (ok) *reinterpret_cast<float4 *>(Xs + m * A.K + 4 * l) = xin[p];
    }
#pragma unroll
    for (int i = 0; i < GCOLS / 256; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < ncol) cs[idx] = cv[i];
    }
    // zero the K padding columns of the aggregated tile
    for (int idx = threadIdx.x; idx < GBM * (Kpad - A.K); idx += 256) {
        const int m = idx / (Kpad - A.K), k = A.K + idx % (Kpad - A.K);
        As[m * lda + k] = 0.f;
    }
    __syncthreads();
    STAMP(1);

    // ---- (1) CSR edge-gather from LDS -> aggregated tile ------------------------------------
    // Fast path, branch free: the first four neighbours of every row are fetched with
    // unconditional LDS reads (a neighbour outside the tile, a dropped self loop or a missing
    // edge reads the row itself with weight 0), so several rows' reads are in flight at once.
    // Neighbours outside the tile (molecules straddling a tile edge) and edges beyond the
    // fourth are rare and take the exec-masked slow path below.
    const bool cols_in_lds = (rp[GBM] - e_lo) <= GCOLS;
#pragma unroll                                     // full: xin[p] must stay in registers (no scratch)
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * 4 + wid) * RPW + sub;
        const int row = m0 + m;
        const int e0 = rp[m], cnt = rp[m + 1] - e0;
        int j[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int ci = e0 + q - e_lo;
            ci = (q < cnt && ci < GCOLS) ? ci : 0;
            const int c = cs[ci];
            j[q] = (q < cnt && cols_in_lds) ? c : row;
        }
        float4 accv = xin[p];
        accv.x *= A.self_coef; accv.y *= A.self_coef; accv.z *= A.self_coef; accv.w *= A.self_coef;
        float4 u[4];
        bool far[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jl = j[q] - m0;
            const bool in_tile = jl >= 0 && jl < GBM;
            far[q] = !in_tile;                                   // j == row is in the tile: never far
            const int src = (in_tile && j[q] != row) ? jl : m;
            u[q] = *reinterpret_cast<const float4 *>(Xs + src * A.K + (ok ? 4 * l : 0));
            const float w = (in_tile && j[q] != row) ? 1.f : 0.f;
            f4_fma(accv, w, u[q]);
        }
        if (ok) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (far[q]) f4_add(accv, X[(int64_t)j[q] * ldx4 + l]);
            const int first = cols_in_lds ? 4 : 0;               // generic tail (and oversized tiles)
            for (int e = e0 + first; e < e0 + cnt; ++e) {
                const int jj = A.col[e];
                if (jj != row) f4_add(accv, X[(int64_t)jj * ldx4 + l]);
            }
            float *dst = As + m * lda + 4 * l;
            dst[0] = accv.x; dst[1] = accv.y; dst[2] = accv.z; dst[3] = accv.w;
        }
    }
    __syncthreads();
    STAMP(2);

    f32x16 acc[MT][NTW];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    };

    // ---- (2) hidden = act(agg @ W1 + b1) -> LDS ---------------------------------------------
    zero();
    gemm_lds_global<N, MT, NTW>(acc, As, lda, A.w1, A.K, Kpad, m_base, n_base, fi, fh, bfrag);
    STAMP(3);
    load_b_chunk<N, NTW>(bfrag, A.w2, N, 0, n_base, fi, fh);      // lands under the epilogue
    tile_to_lds<ACT, MT, NTW>(acc, Hs, LDH, A.b1, A.epi.act_param, nullptr, nullptr, m_base, n_base, fi, fh);
    __syncthreads();
    STAMP(4);

    // ---- (3) y = affine(act(hidden @ W2 + b2)) ------------------------------------------------
    zero();
    gemm_lds_global<N, MT, NTW>(acc, Hs, LDH, A.w2, N, N, m_base, n_base, fi, fh, bfrag);
    STAMP(5);
    tile_to_lds<ACT, MT, NTW>(acc, Ys, LDO, A.b2, A.epi.act_param, A.epi.bn_scale, A.epi.bn_shift, m_base,
                              n_base, fi, fh);
    __syncthreads();
    // whole rows, 16 B per lane: a wavefront writes 1 KiB of consecutive output per instruction
    constexpr int N4 = N / 4;
    const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);
    for (int idx = threadIdx.x; idx < GBM * N4; idx += 256) {
        const int m = idx / N4, c4 = idx % N4;
        const int row = m0 + m;
        if (row >= A.n) continue;
        const float4 v = *reinterpret_cast<const float4 *>(Ys + m * LDO + 4 * c4);
        float *dst = A.y + (int64_t)row * A.ldy + 4 * c4;
        if (vec_out) *reinterpret_cast<float4 *>(dst) = v;
        else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
    }
    STAMP(6);
}

static size_t gin_lds_bytes(int K, int N, int GBM) {
    const int Kpad = (K + KC - 1) / KC * KC;
    const size_t r0 = (size_t)GBM * ((Kpad + 1) > (N + 4) ? (Kpad + 1) : (N + 4));
    const size_t r1 = (size_t)GBM * (K > N + 1 ? K : N + 1);
    return sizeof(float) * (r0 + r1 + GBM + 4 + GCOLS);
}

template <int N, int WCOLS, int LG, int ACT, int GBM>
static int launch_gin(const GinArgs &A, hipStream_t s) {
    // opt in to more than 64 KiB of dynamic LDS, but only to what this shape needs: the limit
    // that is declared, not the amount that is launched, is what the dispatcher budgets per
    // workgroup (declaring 160 KiB pinned the kernel to ONE workgroup per CU).
    static size_t declared = 0;                     // per instantiation; monotone, benign race
    const size_t bytes = gin_lds_bytes(A.K, N, GBM);
    if (bytes > declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gin_layer<N, WCOLS, LG, ACT, GBM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gin_layer: cannot raise the dynamic LDS limit to %zu bytes", bytes);
        declared = bytes;
    }
    const int blocks = (A.n + GBM - 1) / GBM;
    hipLaunchKernelGGL((k_gin_layer<N, WCOLS, LG, ACT, GBM>), dim3(blocks), dim3(256), bytes, s, A);
    return launch_status("gin_layer");
}

template <int N, int WCOLS, int LG, int GBM>
static int pick_act(const GinArgs &A, hipStream_t s) {
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    return launch_gin<N, WCOLS, LG, BLGNN_ACT_RELU, GBM>(A, s);
        case BLGNN_ACT_PRELU:   return launch_gin<N, WCOLS, LG, BLGNN_ACT_PRELU, GBM>(A, s);
        case BLGNN_ACT_SIGMOID: return launch_gin<N, WCOLS, LG, BLGNN_ACT_SIGMOID, GBM>(A, s);
        case BLGNN_ACT_TANH:    return launch_gin<N, WCOLS, LG, BLGNN_ACT_TANH, GBM>(A, s);
        default:                return launch_gin<N, WCOLS, LG, BLGNN_ACT_IDENTITY, GBM>(A, s);
    }
}

template <int N, int WCOLS, int GBM>
static int pick_lg(const GinArgs &A, hipStream_t s) {
    const int k4 = A.K / 4;
    if (k4 <= 16) return pick_act<N, WCOLS, 16, GBM>(A, s);
    if (k4 <= 32) return pick_act<N, WCOLS, 32, GBM>(A, s);
    return pick_act<N, WCOLS, 64, GBM>(A, s);
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_gin_layer(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x,
                               int64_t ldx, int32_t n, int32_t K, const float *w1, const float *b1,
                               const float *w2, const float *b2, int32_t N, float eps, int32_t act,
                               const float *act_param, const float *bn_scale, const float *bn_shift,
                               float *y, int64_t ldy, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && K > 0 && N > 0 && nnz >= 0, "gin_layer: bad sizes");
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(rowptr && (col || nnz == 0) && x && w1 && w2 && y, "gin_layer: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldy >= N, "gin_layer: leading dimension too small");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "gin_layer: bad activation id");
    BLGNN_CHECK_ARG(x != y, "gin_layer: x and y must not alias");
    if (!(N == 64 || N == 128 || N == 256) || K % 4 != 0 || K > 256 || ldx % 4 != 0 || !aligned16(x))
        return fail(BLGNN_ERR_UNSUPPORTED,
                    "gin_layer: fused kernel needs N in {64,128,256}, K %% 4 == 0, K <= 256, float4-aligned x "
                    "(got N=%d K=%d); use spmm_sum + linear", N, K);
    GinArgs A;
    A.rowptr = rowptr; A.col = col; A.x = x; A.ldx = ldx; A.n = n; A.K = K;
    A.w1 = w1; A.b1 = b1; A.w2 = w2; A.b2 = b2; A.self_coef = 1.f + eps;
    A.epi = Epilogue{b2, act_param, bn_scale, bn_shift, act};
    A.y = y; A.ldy = ldy;
    hipStream_t s = (hipStream_t)stream;
    // 32-row tiles (4 waves side by side over N) halve the LDS footprint -> twice the resident
    // workgroups per CU, and halve the work quantum -> a shorter tail when the grid is only a
    // few tiles per CU; 64-row tiles reuse each weight fragment twice and win on big inputs.
    const char *force = getenv("BLGNN_GIN_ROWS");
    const bool small = force ? (atoi(force) == 32) : (n < g_gin_small_rows);
    if (N == 256) return (small || !force) ? pick_lg<256, 4, 32>(A, s) : pick_lg<256, 4, 64>(A, s);
    if (N == 128) return small ? pick_lg<128, 4, 32>(A, s) : pick_lg<128, 4, 64>(A, s);
    return pick_lg<64, 2, 64>(A, s);
}
