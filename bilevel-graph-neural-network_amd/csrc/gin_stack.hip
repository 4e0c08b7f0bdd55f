// The whole lower level in ONE launch: L stacked GIN layers + the (multi-scale) node -> graph
// readout (reference: model/layers.py:26-31,51-57 x L, model/layers_aggregation.py:27-42).
//
// The merged molecular graph is block diagonal, so a tile made of WHOLE molecules has no
// neighbour outside itself at any depth: a workgroup stages the tile's input features once
// and then runs layer after layer out of LDS --
//     gather (LDS) -> GEMM (fp32 MFMA) -> act -> GEMM -> act + eval-BN -> per-molecule mean
// -- writing only the pooled rows [G, L*d] (and optionally the last activations) to HBM.
// Per-layer launches, the L-1 intermediate activation round trips, L-1 staging phases and
// L-1 grid tails disappear.  Two LDS regions ping-pong: the aggregated tile / layer output
// live in one, the layer input / hidden activation in the other, and they swap roles every
// layer (72 KB per workgroup at d=128: two workgroups per CU).
// Tiles are built on the host from the molecule sizes (greedy, consecutive molecules, <= 64
// rows); molecules larger than a tile go through the per-layer kernels instead.
#include "gin_common.hpp"

namespace blgnn {

constexpr int STACK_MAX_LAYERS = 8;
constexpr int SBM = 64;                           // rows per tile

struct StackLayer {
    const float *w1, *b1, *w2, *b2, *slope, *scale, *shift;
    int32_t act;
    float self_coef;
    int32_t pool_col;                             // column offset of this layer in `pooled`, -1 = not pooled
};

struct StackArgs {
    const int32_t *rowptr, *col;
    const float *x0; int64_t ldx0; int32_t K0;
    int32_t n_tiles;
    const int32_t *tile_row;                      // [T+1] first row of each tile
    const int32_t *tile_mol;                      // [T+1] first molecule of each tile
    const int32_t *seg;                           // [G+1] molecule row offsets
    int32_t L;
    StackLayer layer[STACK_MAX_LAYERS];
    float *pooled; int64_t ldp;
    const int32_t *row_map;                       // nullable: molecule -> pooled row
    int32_t mean;                                 // 1 = scatter_mean, 0 = scatter_add
    float *y_last; int64_t ldy;                   // nullable
};

// agg[m] = self_coef * X[m] + sum_{j in row m, j != m} X[j]   -- everything inside the tile
template <int LG>
__device__ __forceinline__ void gather_in_tile(const float *Xs, int ldxs, float *As, int lda, int K,
                                               const int *rp, const int *cs, const int32_t *colg, int e_lo,
                                               bool cols_in_lds, int m0, int rows, float self_coef,
                                               int wid, int lane) {
    constexpr int RPW = WAVE / LG;
    constexpr int PASSES = SBM / (4 * RPW);
    const int sub = lane / LG, l = lane % LG;
    const bool ok = 4 * l < K;
    const int c0 = ok ? 4 * l : 0;
#pragma unroll 2                                   // two rows' LDS reads in flight; more only costs registers
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * 4 + wid) * RPW + sub;
        const bool live = m < rows;
        const int row = m0 + m;
        const int e0 = live ? rp[m] : 0, cnt = live ? rp[m + 1] - e0 : 0;
        int j[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int ci = e0 + q - e_lo;
            ci = (q < cnt && ci < GCOLS) ? ci : 0;
            const int c = cs[ci];
            j[q] = (q < cnt && cols_in_lds) ? c : row;
        }
        float4 acc = *reinterpret_cast<const float4 *>(Xs + (live ? m : 0) * ldxs + c0);
        acc.x *= self_coef; acc.y *= self_coef; acc.z *= self_coef; acc.w *= self_coef;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool use = j[q] != row;
            const float4 u = *reinterpret_cast<const float4 *>(Xs + (use ? j[q] - m0 : (live ? m : 0)) * ldxs + c0);
            f4_fma(acc, use ? 1.f : 0.f, u);
        }
        if (live) {
            const int first = cols_in_lds ? 4 : 0;
            for (int e = e0 + first; e < e0 + cnt; ++e) {
                const int jj = (cols_in_lds && e - e_lo < GCOLS) ? cs[e - e_lo] : colg[e];
                if (jj != row) f4_add(acc, *reinterpret_cast<const float4 *>(Xs + (jj - m0) * ldxs + c0));
            }
        } else {
            acc = f4_zero();
        }
        if (ok) {
            float *dst = As + m * lda + 4 * l;
            dst[0] = acc.x; dst[1] = acc.y; dst[2] = acc.z; dst[3] = acc.w;
        }
    }
}

// dst[m][c] = affine(leaky(acc + bias[c])): relu / identity / prelu are all
// v > 0 ? v : slope * v (slope 0, 1, per channel), so the stack needs ONE epilogue code path
// (sigmoid / tanh layers are routed to the per-layer kernels by the host).
template <int MT, int NTW>
__device__ __forceinline__ void tile_out(const f32x16 (&acc)[MT][NTW], float *dst, int ld, const float *bias,
                                         const float *slope_v, float slope_s, const float *scale_v,
                                         const float *shift_v, int m_base, int n_base, int fi, int fh) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int c = n_base + nt * 32 + fi;
        const float b = bias ? bias[c] : 0.f;
        const float sl = slope_v ? slope_v[c] : slope_s;
        const float sc = scale_v ? scale_v[c] : 1.f;
        const float sh = (scale_v && shift_v) ? shift_v[c] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_base + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float v = acc[mt][nt][r] + b;
                v = v > 0.f ? v : sl * v;
                dst[m * ld + c] = fmaf(v, sc, sh);
            }
    }
}

template <int N, int LG0>
__global__ void __launch_bounds__(256, 2) k_gin_stack(const StackArgs A) {
    constexpr int WCOLS = N >= 128 ? 4 : 2, WROWS = 4 / WCOLS;
    constexpr int MT = SBM / 32 / WROWS, NTW = N / 32 / WCOLS;
    constexpr int LGN = N / 4 >= WAVE ? WAVE : N / 4;           // lanes per row for layers >= 1
    constexpr int LDH = N + 1, LDX = N + 4;                     // hidden (odd) / layer in-out (16 B rows)
    extern __shared__ float lds[];
    const int K0pad = (A.K0 + KC - 1) / KC * KC;
    const int ldx0 = ((A.K0 + 3) & ~3) + 4;
    int reg = SBM * (N + 4);                                    // floats per ping-pong region
    if (SBM * (K0pad + 1) > reg) reg = SBM * (K0pad + 1);
    if (SBM * ldx0 > reg) reg = SBM * ldx0;
    int *rp = reinterpret_cast<int *>(lds + 2 * reg);           // [SBM+1]
    int *cs = rp + SBM + 4;                                     // [GCOLS]
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int m_base = (wid / WCOLS) * (MT * 32), n_base = (wid % WCOLS) * (NTW * 32);

    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        const int m0 = A.tile_row[tile], rows = A.tile_row[tile + 1] - m0;
        const int g0 = A.tile_mol[tile], g1 = A.tile_mol[tile + 1];
        __syncthreads();                                        // previous tile fully consumed
        // ---- stage: rowptr slice, layer-0 weights chunk, input rows, column indices ------------
        if (threadIdx.x <= SBM) {
            const int r = threadIdx.x < rows ? m0 + threadIdx.x : m0 + rows;
            rp[threadIdx.x] = A.rowptr[r];
        }
        float bfrag[KC / 2][NTW];
        load_b_chunk<N, NTW>(bfrag, A.layer[0].w1, A.K0, 0, n_base, fi, fh);
        {
            const int k4 = A.K0 >> 2;
            float *Xs = lds + reg;
            for (int idx = threadIdx.x; idx < rows * k4; idx += 256) {
                const int r = idx / k4, c4 = idx - r * k4;
                const float4 v = *reinterpret_cast<const float4 *>(A.x0 + (int64_t)(m0 + r) * A.ldx0 + 4 * c4);
                *reinterpret_cast<float4 *>(Xs + r * ldx0 + 4 * c4) = v;
            }
        }
        __syncthreads();
        const int e_lo = rp[0];
        const int ncol_all = rp[rows] - e_lo;
        const bool cols_in_lds = ncol_all <= GCOLS;
        if (cols_in_lds)
            for (int i = threadIdx.x; i < ncol_all; i += 256) cs[i] = A.col[e_lo + i];
        __syncthreads();

        for (int l = 0; l < A.L; ++l) {
            const StackLayer &P = A.layer[l];
            // layer input lives in R[(l+1)&1]; aggregated tile and layer output go to R[l&1]
            float *Ra = lds + ((l & 1) ? reg : 0), *Rb = lds + ((l & 1) ? 0 : reg);
            const float *Xs = Rb;
            float *As = Ra, *Hs = Rb, *Ys = Ra;
            const float slope_s = P.act == BLGNN_ACT_RELU ? 0.f : 1.f;
            const float *slope_v = P.act == BLGNN_ACT_PRELU ? P.slope : nullptr;
            const int K = l == 0 ? A.K0 : N;
            const int Kpad = l == 0 ? K0pad : N;
            const int lda = Kpad + 1, ldxs = l == 0 ? ldx0 : LDX;
            // ---- (1) gather inside the tile ---------------------------------------------------
            if (l == 0) {
                gather_in_tile<LG0>(Xs, ldxs, As, lda, K, rp, cs, A.col, e_lo, cols_in_lds, m0, rows,
                                    P.self_coef, wid, lane);
                for (int idx = threadIdx.x; idx < SBM * (Kpad - K); idx += 256) {   // zero K padding
                    const int m = idx / (Kpad - K), k = K + idx % (Kpad - K);
                    As[m * lda + k] = 0.f;
                }
            } else {
                gather_in_tile<LGN>(Xs, ldxs, As, lda, K, rp, cs, A.col, e_lo, cols_in_lds, m0, rows,
                                    P.self_coef, wid, lane);
            }
            __syncthreads();
            // ---- (2) hidden = act(agg @ W1 + b1) ------------------------------------------------
            f32x16 acc[MT][NTW];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            gemm_lds_global<N, MT, NTW>(acc, As, lda, P.w1, K, Kpad, m_base, n_base, fi, fh, bfrag);
            load_b_chunk<N, NTW>(bfrag, P.w2, N, 0, n_base, fi, fh);
            tile_out<MT, NTW>(acc, Hs, LDH, P.b1, slope_v, slope_s, nullptr, nullptr, m_base, n_base, fi, fh);
            __syncthreads();
            // ---- (3) out = affine(act(hidden @ W2 + b2)) -> the next layer's input ------------------
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
            gemm_lds_global<N, MT, NTW>(acc, Hs, LDH, P.w2, N, N, m_base, n_base, fi, fh, bfrag);
            if (l + 1 < A.L) load_b_chunk<N, NTW>(bfrag, A.layer[l + 1].w1, N, 0, n_base, fi, fh);
            tile_out<MT, NTW>(acc, Ys, LDX, P.b2, slope_v, slope_s, P.scale, P.shift, m_base, n_base, fi, fh);
            __syncthreads();
            // ---- (4) per-molecule readout of this layer (multi-scale column slice) -----------------
            if (P.pool_col >= 0) {
                constexpr int MPAR = 256 / N;                    // molecules reduced side by side
                const int c = threadIdx.x % N, ms = threadIdx.x / N;
                for (int g = g0 + ms; g < g1; g += MPAR) {
                    const int r0 = A.seg[g] - m0, r1 = A.seg[g + 1] - m0;
                    float s = 0.f;
                    for (int r = r0; r < r1; ++r) s += Ys[r * LDX + c];
                    if (A.mean) s = s / (float)((r1 - r0) < 1 ? 1 : (r1 - r0));
                    const int orow = A.row_map ? A.row_map[g] : g;
                    A.pooled[(int64_t)orow * A.ldp + P.pool_col + c] = s;
                }
            }
            if (l + 1 == A.L && A.y_last) {
                constexpr int N4 = N / 4;
                for (int idx = threadIdx.x; idx < rows * N4; idx += 256) {
                    const int m = idx / N4, c4 = idx % N4;
                    *reinterpret_cast<float4 *>(A.y_last + (int64_t)(m0 + m) * A.ldy + 4 * c4) =
                        *reinterpret_cast<const float4 *>(Ys + m * LDX + 4 * c4);
                }
            }
            // the next gather reads Ys (already behind a barrier) and writes the other region,
            // whose last readers (GEMM2 on Hs) are behind the same barrier
        }
    }
}

static size_t stack_lds_bytes(int K0, int N) {
    const int K0pad = (K0 + KC - 1) / KC * KC;
    const int ldx0 = ((K0 + 3) & ~3) + 4;
    size_t reg = (size_t)SBM * (N + 4);
    if ((size_t)SBM * (K0pad + 1) > reg) reg = (size_t)SBM * (K0pad + 1);
    if ((size_t)SBM * ldx0 > reg) reg = (size_t)SBM * ldx0;
    return sizeof(float) * (2 * reg + SBM + 4 + GCOLS);
}

template <int N, int LG0>
static int launch_stack(const StackArgs &A, hipStream_t s) {
    static size_t declared = 0;
    const size_t bytes = stack_lds_bytes(A.K0, N);
    if (bytes > declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gin_stack<N, LG0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gin_stack: cannot raise the dynamic LDS limit to %zu bytes", bytes);
        declared = bytes;
    }
    const int blocks = A.n_tiles < 256 * 4 ? A.n_tiles : 256 * 4;
    hipLaunchKernelGGL((k_gin_stack<N, LG0>), dim3(blocks), dim3(256), bytes, s, A);
    return launch_status("gin_stack");
}

template <int N>
static int pick_lg0(const StackArgs &A, hipStream_t s) {
    const int k4 = (A.K0 + 3) / 4;
    if (k4 <= 16) return launch_stack<N, 16>(A, s);
    if (k4 <= 32) return launch_stack<N, 32>(A, s);
    return launch_stack<N, 64>(A, s);
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_gin_stack(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x0,
                               int64_t ldx0, int32_t K0, int32_t N, int32_t num_tiles,
                               const int32_t *tile_row, const int32_t *tile_mol, const int32_t *seg,
                               int32_t num_layers, const float *const *w1_host, const float *const *b1_host,
                               const float *const *w2_host, const float *const *b2_host,
                               const float *const *slope_host, const float *const *scale_host,
                               const float *const *shift_host, const int32_t *act_host,
                               const float *eps_host, const int32_t *pool_col_host, int32_t reduce_kind,
                               float *pooled, int64_t ldp, const int32_t *row_map, float *y_last,
                               int64_t ldy, void *stream) {
    BLGNN_CHECK_ARG(num_tiles >= 0 && K0 > 0 && N > 0 && nnz >= 0, "gin_stack: bad sizes");
    if (num_tiles == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(num_layers >= 1 && num_layers <= STACK_MAX_LAYERS, "gin_stack: 1..8 layers");
    BLGNN_CHECK_ARG(rowptr && (col || nnz == 0) && x0 && tile_row && tile_mol && seg, "gin_stack: null pointer");
    BLGNN_CHECK_ARG(w1_host && b1_host && w2_host && b2_host && slope_host && scale_host && shift_host &&
                    act_host && eps_host && pool_col_host, "gin_stack: null parameter table");
    BLGNN_CHECK_ARG(reduce_kind == BLGNN_REDUCE_SUM || reduce_kind == BLGNN_REDUCE_MEAN, "gin_stack: bad reduce kind");
    if (!(N == 64 || N == 128 || N == 256) || K0 % 4 != 0 || K0 > 256 || ldx0 % 4 != 0 || !aligned16(x0) ||
        (y_last && (ldy % 4 != 0 || !aligned16(y_last))))
        return fail(BLGNN_ERR_UNSUPPORTED, "gin_stack: needs N in {64,128,256}, K0 %% 4 == 0, K0 <= 256 and "
                    "float4-aligned x0 / y_last (got N=%d K0=%d)", N, K0);
    StackArgs A = {};
    A.rowptr = rowptr; A.col = col; A.x0 = x0; A.ldx0 = ldx0; A.K0 = K0; A.n_tiles = num_tiles;
    A.tile_row = tile_row; A.tile_mol = tile_mol; A.seg = seg; A.L = num_layers;
    bool any_pool = false;
    for (int l = 0; l < num_layers; ++l) {
        BLGNN_CHECK_ARG(w1_host[l] && w2_host[l], "gin_stack: null weight in layer %d", l);
        BLGNN_CHECK_ARG(act_host[l] >= 0 && act_host[l] <= BLGNN_ACT_TANH, "gin_stack: bad activation id");
        if (act_host[l] == BLGNN_ACT_SIGMOID || act_host[l] == BLGNN_ACT_TANH)
            return fail(BLGNN_ERR_UNSUPPORTED, "gin_stack: sigmoid / tanh layers take the per-layer kernels");
        BLGNN_CHECK_ARG(act_host[l] != BLGNN_ACT_PRELU || slope_host[l], "gin_stack: PReLU layer without slopes");
        StackLayer &P = A.layer[l];
        P.w1 = w1_host[l]; P.b1 = b1_host[l]; P.w2 = w2_host[l]; P.b2 = b2_host[l];
        P.slope = slope_host[l]; P.scale = scale_host[l]; P.shift = shift_host[l];
        P.act = act_host[l]; P.self_coef = 1.f + eps_host[l]; P.pool_col = pool_col_host[l];
        any_pool |= P.pool_col >= 0;
    }
    BLGNN_CHECK_ARG(!any_pool || pooled, "gin_stack: pooled output required");
    BLGNN_CHECK_ARG(any_pool || y_last, "gin_stack: nothing to write (no pooled layer, no y_last)");
    A.pooled = pooled; A.ldp = ldp; A.row_map = row_map; A.mean = reduce_kind == BLGNN_REDUCE_MEAN;
    A.y_last = y_last; A.ldy = ldy;
    hipStream_t s = (hipStream_t)stream;
    if (N == 256) return pick_lg0<256>(A, s);
    if (N == 128) return pick_lg0<128>(A, s);
    return pick_lg0<64>(A, s);
}
