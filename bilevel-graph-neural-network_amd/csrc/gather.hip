// CSR edge-gather -> wavefront segment-reduce kernels (HBM / L2 bound).
//
// One "row worker" = LG adjacent lanes of a 64-lane wavefront owns one output row
// (LG = 16/32/64 chosen so that one float4 per lane covers the row when d <= 256;
// wider rows give each lane CH float4 chunks).  A worker walks its CSR row four
// edges at a time: the four neighbour-row loads are issued back to back (each
// wave-instruction is a fully coalesced 16 B/lane row read), then reduced in
// registers in CSR order, so the result is deterministic.  Neighbour rows of the
// block-diagonal molecular graph lie within a few KB of the row itself and are
// served by L1/L2; HBM sees every x row once and every y row once.
//
// Edge weight "modes" share the skeleton:
//   SUM : w_e (or 1), self term self_coef        (GINConv / scatter_add)
//   GCN : dis_i w_e dis_j, self dis_i^2           (GCNConv)
//   GAT : exp(lrelu(sd_i + ss_j) - mx_g)/(den_g+1e-16), appended self loop (GATConv)
//   RELU: acc += relu(p_i + q_j)                  (restructured GMN message)
#include "common.hpp"
#include "sp_common.hpp"

namespace blgnn {

enum { MODE_SUM = 0, MODE_GCN = 1, MODE_GAT = 2, MODE_RELU = 3 };

struct GatherArgs {
    const int32_t *rowptr;
    const int32_t *col;
    const float *x;      // gathered matrix (h for GCN/GAT, q for RELU)
    int64_t ldx;
    float *y;
    int64_t ldy;
    int32_t n, d;
    // SUM / GCN
    const float *w;      // per-edge weight, nullable
    float self_coef;
    int32_t skip_self;
    const float *dis;    // GCN
    // GAT
    const float *s_dst, *s_src, *mx, *den;
    const float4 *packed;  // optional [n]: (s_src, mx, den, -) of every node in one 16-byte word (SRC grouping: one gather per edge instead of three)
    int32_t group;
    float slope;
    // RELU
    const float *p;
    int64_t ldp;
    float *amax;         // optional [n]: atomic max of the result row's magnitudes into it (k_gather only)
    _Float16 *ys;        // optional sp-format output (sp_common.hpp; k_gather with one wavefront per row, d % 256 == 0);
    int64_t ldys;        //   y itself may then be null
    uint8_t *ye;         //   block scale exponents (one byte per 256 columns)
    int64_t ldye;
    // epilogue
    Epilogue epi;
    int32_t accumulate;
    // optional row subset (k_gather_wide / k_gather_wave only): compute rows row_list[0 .. n_list) instead of 0 .. n
    const int64_t *row_list;
    int32_t n_list;
    // degree window (k_gather_wave / k_gather_wide as a pair: every row is computed by exactly one of them): rows with fewer
    // than deg_min or more than deg_max edges are left alone
    int32_t deg_min = 0, deg_max = 0x7fffffff;
    int32_t hub_blocks = 0;            // k_gather_wave: the first hub_blocks workgroups compute the rows above deg_max
};

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

template <int MODE>
struct EdgeCtx {
    float a, b, c;   // per-row cached scalars
    __device__ __forceinline__ void init(const GatherArgs &A, int row) {
        if (MODE == MODE_GCN) a = A.dis[row];
        if (MODE == MODE_GAT) {
            a = A.s_dst[row];
            if (A.group == BLGNN_GAT_GROUP_DST) { b = A.mx[row]; c = 1.f / (A.den[row] + 1e-16f); }
        }
    }
    // weight of the explicit self term
    __device__ __forceinline__ float self_w(const GatherArgs &A, int row) const {
        if (MODE == MODE_SUM) return A.self_coef;
        if (MODE == MODE_GCN) return a * a;
        if (MODE == MODE_GAT) {
            if (A.packed) {
                const float4 t = A.packed[row];
                return expf(lrelu(a + t.x, A.slope) - t.y) / (t.z + 1e-16f);
            }
            float e = lrelu(a + A.s_src[row], A.slope);
            return expf(e - A.mx[row]) / (A.den[row] + 1e-16f);
        }
        return 0.f;
    }
    // weight of edge e = (j -> row); returns 0 for dropped self loops
    __device__ __forceinline__ float edge_w(const GatherArgs &A, int row, int j, int e) const {
        if (MODE == MODE_SUM) {
            if (A.skip_self && j == row) return 0.f;
            return A.w ? A.w[e] : 1.f;
        }
        if (MODE == MODE_GCN) {
            if (j == row) return 0.f;
            return a * (A.w ? A.w[e] : 1.f) * A.dis[j];
        }
        if (MODE == MODE_GAT) {
            if (j == row) return 0.f;
            if (A.packed && A.group != BLGNN_GAT_GROUP_DST) {
                const float4 t = A.packed[j];                    // same arithmetic, one gather
                return expf(lrelu(a + t.x, A.slope) - t.y) / (t.z + 1e-16f);
            }
            float ev = lrelu(a + A.s_src[j], A.slope);
            if (A.group == BLGNN_GAT_GROUP_DST) return expf(ev - b) * c;
            return expf(ev - A.mx[j]) / (A.den[j] + 1e-16f);
        }
        return 1.f;
    }
};

template <int MODE, int LG, int CH>
__global__ void __launch_bounds__(256) k_gather(const GatherArgs A) {
    constexpr int RPW = WAVE / LG;                       // rows per wave
    const int lane = threadIdx.x & (WAVE - 1);
    const int sub = lane / LG, l = lane % LG;
    const int d4 = A.d >> 2;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int wave_total = (gridDim.x * blockDim.x) / WAVE;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2, ldy4 = A.ldy >> 2, ldp4 = A.ldp >> 2;

    for (int row = wave_global * RPW + sub; row < A.n; row += wave_total * RPW) {
        EdgeCtx<MODE> ctx;
        ctx.init(A, row);
        float4 acc[CH], prow[CH];
        bool ok[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            ok[k] = (l + k * LG) < d4;
            acc[k] = f4_zero();
            prow[k] = f4_zero();
        }
        if (MODE == MODE_RELU) {
            const float4 *P = reinterpret_cast<const float4 *>(A.p) + (int64_t)row * ldp4;
#pragma unroll
            for (int k = 0; k < CH; ++k) if (ok[k]) prow[k] = P[l + k * LG];
        } else {
            const float sw = ctx.self_w(A, row);
            if (sw != 0.f) {
                const float4 *S = X + (int64_t)row * ldx4;
#pragma unroll
                for (int k = 0; k < CH; ++k) if (ok[k]) { float4 v = S[l + k * LG]; f4_fma(acc[k], sw, v); }
            }
        }
        const int e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
        for (int e = e0; e < e1; e += 4) {
            int j[4];
            float w[4];
            float4 v[4][CH];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = (e + u) < e1;
                j[u] = in ? A.col[e + u] : row;
                w[u] = in ? ctx.edge_w(A, row, j[u], e + u) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (w[u] != 0.f) {
                    const float4 *R = X + (int64_t)j[u] * ldx4;
#pragma unroll
                    for (int k = 0; k < CH; ++k) if (ok[k]) v[u][k] = R[l + k * LG];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (w[u] != 0.f) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (!ok[k]) continue;
                        if (MODE == MODE_RELU) {
                            acc[k].x += fmaxf(prow[k].x + v[u][k].x, 0.f);
                            acc[k].y += fmaxf(prow[k].y + v[u][k].y, 0.f);
                            acc[k].z += fmaxf(prow[k].z + v[u][k].z, 0.f);
                            acc[k].w += fmaxf(prow[k].w + v[u][k].w, 0.f);
                        } else if (MODE == MODE_SUM) {
                            if (A.w) f4_fma(acc[k], w[u], v[u][k]); else f4_add(acc[k], v[u][k]);
                        } else {
                            f4_fma(acc[k], w[u], v[u][k]);
                        }
                    }
                }
            }
        }
        if ((MODE == MODE_RELU || MODE == MODE_SUM) && LG == 64 && A.ys) {
            // sp output: chunk k of the row is exactly its k-th 256-column scale block, held by the 64 lanes of
            // this wavefront -- block maximum (sums of relus are non-negative; |.| for the plain sum), power-of-two
            // scale, two fp16 pieces
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (!ok[k]) continue;                                  // wave-uniform (d4 is a multiple of 64)
                float mx = fmaxf(fmaxf(fabsf(acc[k].x), fabsf(acc[k].y)), fmaxf(fabsf(acc[k].z), fabsf(acc[k].w)));
                mx = lanes_max<64>(mx);
                float sc, inv;
                pow2_scale(mx, sc, inv);
                if (l == 0) A.ye[(int64_t)row * A.ldye + k] = sp_exp_of(inv);
                const f32x4v sv = {acc[k].x * sc, acc[k].y * sc, acc[k].z * sc, acc[k].w * sc};
                sp_store4_pair(A.ys + (int64_t)row * A.ldys + k * (2 * SP_BLOCK), 4 * l, sv, l);
            }
            if (!A.y) continue;
        }
        float4 *Y = reinterpret_cast<float4 *>(A.y) + (int64_t)row * ldy4;
        if (MODE == MODE_RELU && A.amax) {
            // sums of relus are non-negative: the row maximum over the LG lanes of the row, one atomic per row
            float mx = 0.f;
#pragma unroll
            for (int k = 0; k < CH; ++k)
                if (ok[k]) mx = fmaxf(mx, fmaxf(fmaxf(acc[k].x, acc[k].y), fmaxf(acc[k].z, acc[k].w)));
            mx = lanes_max<LG>(mx);
            if (l == 0) atomicMax(reinterpret_cast<unsigned *>(A.amax) + row, __float_as_uint(mx));
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (!ok[k]) continue;
            const int c4 = l + k * LG, c = c4 * 4;
            float4 r = acc[k];
            if (MODE == MODE_GCN || MODE == MODE_GAT) {
                float4 o = A.accumulate ? Y[c4] : f4_zero();
                const Epilogue &E = A.epi;
                float t[4] = {r.x, r.y, r.z, r.w};
                float oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float val = t[q] + (E.bias ? E.bias[c + q] : 0.f) + oo[q];
                    float sl = (E.act == BLGNN_ACT_PRELU && E.act_param) ? E.act_param[c + q] : 0.f;
                    val = apply_act(val, E.act, sl);
                    if (E.bn_scale) val = fmaf(val, E.bn_scale[c + q], E.bn_shift ? E.bn_shift[c + q] : 0.f);
                    t[q] = val;
                }
                r = make_float4(t[0], t[1], t[2], t[3]);
            }
            Y[c4] = r;
        }
    }
}

// Low-degree fast path (molecular graphs, mean degree ~2): a lane group owns FOUR consecutive
// rows per pass.  The dependent chain rowptr -> col -> rows costs three memory latencies; one
// row per pass exposes all three per row (measured 3.4 TB/s at d=256, 20 waves/CU), four rows
// per pass share them: 5 rowptr entries in one load, the rows' column indices (contiguous in
// CSR) in one coalesced load, then 4 self rows + the first four neighbours of each row issued
// back to back before the first add.  Edges beyond the fourth take a short tail loop.
template <int MODE, int LG>
__global__ void __launch_bounds__(256) k_gather4(const GatherArgs A) {
    constexpr int RPW = WAVE / LG;                       // lane groups per wave
    constexpr int R = 4;                                 // rows per lane group per pass
    const int lane = threadIdx.x & (WAVE - 1);
    const int sub = lane / LG, l = lane % LG;
    const int d4 = A.d >> 2;
    const bool ok = l < d4;
    // XCD-aware work map: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share
    // an XCD, each with its own L2).  A molecule's rows are neighbours of each other, so every
    // XCD gets ONE contiguous eighth of the row range -- otherwise each row is fetched from
    // HBM/Infinity Cache by every XCD that owns one of its neighbours.
    const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    const int ngroups = (A.n + R - 1) / R;
    const int chunk = (ngroups + 7) >> 3;
    const int g_end = (xcd + 1) * chunk < ngroups ? (xcd + 1) * chunk : ngroups;
    const int gpb = (blockDim.x / WAVE) * RPW;           // lane groups per workgroup
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2, ldy4 = A.ldy >> 2, ldp4 = A.ldp >> 2;
    const int lbase = sub * LG;                          // first lane of this group in the wave

    for (int g = xcd * chunk + bix * gpb + (threadIdx.x / WAVE) * RPW + sub; g < g_end; g += bpx * gpb) {
        const int r0 = g * R;
        // rowptr[r0 .. r0+4] in one load, spread by shuffles
        const int rpi = r0 + l;
        const int rpv = (l <= R) ? A.rowptr[rpi <= A.n ? rpi : A.n] : 0;
        int rp[R + 1];
#pragma unroll
        for (int q = 0; q <= R; ++q) rp[q] = __shfl(rpv, lbase + q, WAVE);
        // the rows' column indices are contiguous: one coalesced load covers up to LG of them
        const int ebase = rp[0];
        const int ecnt = rp[R] - ebase;
        const int cv = (l < ecnt) ? A.col[ebase + l] : 0;
        float4 acc[R], prow[R];
        int j[R][4];
        float w[R][4];
        EdgeCtx<MODE> ctx[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int row = r0 + q;
            const bool live = row < A.n;
            if (live) ctx[q].init(A, row);
            acc[q] = f4_zero();
            prow[q] = f4_zero();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = rp[q] + u;
                const int rel = e - ebase;
                const bool in = live && e < rp[q + 1];
                int jj = __shfl(cv, lbase + (rel < LG ? rel : 0), WAVE);
                if (in && rel >= LG) jj = A.col[e];
                j[q][u] = in ? jj : row;
                w[q][u] = in ? ctx[q].edge_w(A, row, j[q][u], e) : 0.f;
            }
        }
        float4 v[R][4], self[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int row = r0 + q;
            const bool live = row < A.n && ok;
            if (MODE == MODE_RELU) {
                if (live) prow[q] = reinterpret_cast<const float4 *>(A.p)[(int64_t)row * ldp4 + l];
            } else {
                self[q] = live ? X[(int64_t)row * ldx4 + l] : f4_zero();
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (w[q][u] != 0.f && ok) v[q][u] = X[(int64_t)j[q][u] * ldx4 + l];
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int row = r0 + q;
            if (row >= A.n || !ok) continue;
            if (MODE != MODE_RELU) {
                const float sw = ctx[q].self_w(A, row);
                f4_fma(acc[q], sw, self[q]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (w[q][u] == 0.f) continue;
                if (MODE == MODE_RELU) {
                    acc[q].x += fmaxf(prow[q].x + v[q][u].x, 0.f);
                    acc[q].y += fmaxf(prow[q].y + v[q][u].y, 0.f);
                    acc[q].z += fmaxf(prow[q].z + v[q][u].z, 0.f);
                    acc[q].w += fmaxf(prow[q].w + v[q][u].w, 0.f);
                } else if (MODE == MODE_SUM && !A.w) {
                    f4_add(acc[q], v[q][u]);
                } else {
                    f4_fma(acc[q], w[q][u], v[q][u]);
                }
            }
            for (int e = rp[q] + 4; e < rp[q + 1]; ++e) {                // rare: degree > 4
                const int jj = A.col[e];
                const float ww = ctx[q].edge_w(A, row, jj, e);
                if (ww == 0.f) continue;
                const float4 t = X[(int64_t)jj * ldx4 + l];
                if (MODE == MODE_RELU) {
                    acc[q].x += fmaxf(prow[q].x + t.x, 0.f); acc[q].y += fmaxf(prow[q].y + t.y, 0.f);
                    acc[q].z += fmaxf(prow[q].z + t.z, 0.f); acc[q].w += fmaxf(prow[q].w + t.w, 0.f);
                } else if (MODE == MODE_SUM && !A.w) f4_add(acc[q], t);
                else f4_fma(acc[q], ww, t);
            }
            float4 r = acc[q];
            float4 *Y = reinterpret_cast<float4 *>(A.y) + (int64_t)row * ldy4;
            if (MODE == MODE_GCN || MODE == MODE_GAT) {
                const int c = 4 * l;
                const float4 o = A.accumulate ? Y[l] : f4_zero();
                const Epilogue &E = A.epi;
                float t[4] = {r.x, r.y, r.z, r.w};
                const float oo[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    float val = t[z] + (E.bias ? E.bias[c + z] : 0.f) + oo[z];
                    float sl = (E.act == BLGNN_ACT_PRELU && E.act_param) ? E.act_param[c + z] : 0.f;
                    val = apply_act(val, E.act, sl);
                    if (E.bn_scale) val = fmaf(val, E.bn_scale[c + z], E.bn_shift ? E.bn_shift[c + z] : 0.f);
                    t[z] = val;
                }
                r = make_float4(t[0], t[1], t[2], t[3]);
            }
            Y[l] = r;
        }
    }
}

template <int MODE, int LG>
static void launch_rows4(const GatherArgs &A, hipStream_t s) {
    constexpr int RPW = WAVE / LG;
    const int64_t groups = ((int64_t)A.n + 3) / 4;
    int64_t blocks = (groups + 4 * RPW - 1) / (4 * RPW);
    if (blocks > 256 * 16) blocks = 256 * 16;
    blocks = (blocks + 7) / 8 * 8;                       // whole multiples of the 8 XCDs
    hipLaunchKernelGGL((k_gather4<MODE, LG>), dim3((unsigned)blocks), dim3(256), 0, s, A);
}

// LDS-staged tile variant of the plain aggregation (GINConv's sum, scatter_add over gathered
// rows) for block-diagonal molecular graphs.  A workgroup owns TR consecutive rows: their
// features, rowptr slice and column indices (+ edge weights) are staged into LDS by batches
// of independent coalesced loads, and neighbours are then read from LDS -- ~95 % of a row's
// neighbours are rows of the same tile; the rest (molecules straddling a tile edge) come from
// L2.  The register-gather kernels above re-read every neighbour row through L2 (1.1 GB of
// L2 hits on top of 1.0 GB of compulsory HBM traffic at d=256: 3.5 TB/s); staging leaves only
// the compulsory traffic on the memory system.
constexpr int TILE_COLS = 512;

template <int LG, int TR>
__global__ void __launch_bounds__(256) k_spmm_tile(const GatherArgs A) {
    constexpr int RPW = WAVE / LG;
    constexpr int PASSES = TR / (4 * RPW);
    extern __shared__ float4 tile4[];
    float *Xs = reinterpret_cast<float *>(tile4);            // [TR][d]
    int *rp = reinterpret_cast<int *>(Xs + TR * A.d);        // [TR+1]
    int *cs = rp + TR + 4;                                   // [TILE_COLS]
    float *ws = reinterpret_cast<float *>(cs + TILE_COLS);   // [TILE_COLS] (weighted only)
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int sub = lane / LG, l = lane % LG;
    const int d4 = A.d >> 2;
    const bool ok = l < d4;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2, ldy4 = A.ldy >> 2;
    const int ntiles = (A.n + TR - 1) / TR;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * TR;
        int rpv = 0;
        if (threadIdx.x <= TR) {
            const int r = m0 + threadIdx.x;
            rpv = A.rowptr[r < A.n ? r : A.n];
        }
        float4 xin[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = m0 + (p * 4 + wid) * RPW + sub;
            xin[p] = (ok && row < A.n) ? X[(int64_t)row * ldx4 + l] : f4_zero();
        }
        if (threadIdx.x <= TR) rp[threadIdx.x] = rpv;
        __syncthreads();
        const int e_lo = rp[0];
        int ncol = rp[TR] - e_lo;
        const bool cols_in_lds = ncol <= TILE_COLS;
        ncol = cols_in_lds ? ncol : 0;
        int cv[TILE_COLS / 256];
        float wv[TILE_COLS / 256];
#pragma unroll
        for (int i = 0; i < TILE_COLS / 256; ++i) {
            const int idx = threadIdx.x + 256 * i;
            cv[i] = idx < ncol ? A.col[e_lo + idx] : 0;
            wv[i] = (A.w && idx < ncol) ? A.w[e_lo + idx] : 1.f;
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * 4 + wid) * RPW + sub;
            if (ok) *reinterpret_cast<float4 *>(Xs + m * A.d + 4 * l) = xin[p];
        }
#pragma unroll
        for (int i = 0; i < TILE_COLS / 256; ++i) {
            const int idx = threadIdx.x + 256 * i;
            if (idx < ncol) { cs[idx] = cv[i]; ws[idx] = wv[i]; }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * 4 + wid) * RPW + sub;
            const int row = m0 + m;
            const int e0 = rp[m], cnt = rp[m + 1] - e0;
            int j[4];
            float w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int ci = e0 + q - e_lo;
                ci = (q < cnt && cols_in_lds) ? ci : 0;
                const int c = cs[ci];
                const float wq = ws[ci];
                const bool use = q < cnt && cols_in_lds && !(A.skip_self && c == row);
                j[q] = use ? c : row;
                w[q] = use ? wq : 0.f;
            }
            float4 accv = xin[p];
            accv.x *= A.self_coef; accv.y *= A.self_coef; accv.z *= A.self_coef; accv.w *= A.self_coef;
            bool far[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jl = j[q] - m0;
                const bool in_tile = jl >= 0 && jl < TR;
                far[q] = !in_tile;                               // only real neighbours can be far
                const float4 u = *reinterpret_cast<const float4 *>(Xs + (in_tile ? jl : m) * A.d + (ok ? 4 * l : 0));
                f4_fma(accv, in_tile ? w[q] : 0.f, u);
            }
            if (ok && row < A.n) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (far[q]) f4_fma(accv, w[q], X[(int64_t)j[q] * ldx4 + l]);
                const int first = cols_in_lds ? 4 : 0;           // generic tail / oversized tiles
                for (int e = e0 + first; e < e0 + cnt; ++e) {
                    const int jj = A.col[e];
                    if (A.skip_self && jj == row) continue;
                    f4_fma(accv, A.w ? A.w[e] : 1.f, X[(int64_t)jj * ldx4 + l]);
                }
                reinterpret_cast<float4 *>(A.y)[(int64_t)row * ldy4 + l] = accv;
            }
        }
        __syncthreads();                                         // tile consumed before restaging
    }
}

template <int LG>
static void launch_spmm_tile(const GatherArgs &A, hipStream_t s) {
    constexpr int TR = 32;
    const int ntiles = (A.n + TR - 1) / TR;
    const int blocks = ntiles < 256 * 8 ? ntiles : 256 * 8;
    const size_t bytes = sizeof(float) * ((size_t)TR * A.d + TR + 4 + 2 * TILE_COLS);
    hipLaunchKernelGGL((k_spmm_tile<LG, TR>), dim3(blocks), dim3(256), bytes, s, A);
}

// "Wide row" variant for high-degree graphs (the interaction level: mean degree 20-60,
// maximum in the hundreds).  One 256-thread workgroup per output row.  Per pass of 256 edges:
//  (a) every thread takes ONE edge: reads its column index and computes its weight (for GAT the
//      s_src / max / denominator gathers and the exp run 256-wide instead of once per lane
//      group), parks (j, w) in LDS;
//  (b) the 4 waves x S = 64/LG lane groups are 4*S edge slots striding over the parked edges,
//      eight neighbour rows in flight per slot (coalesced 16 B/lane row reads from L2).
// Partial rows meet by wave shuffles, then in LDS ([4][d]) in a fixed order (deterministic).
// A kernel that gives each row to ONE lane group runs as long as its heaviest row: 241 us vs
// ~20 us for the upper GAT of the DrugBank-shape graph.
constexpr int WIDE_EDGES = 256;

// one output row by the whole 256-thread workgroup (the body of k_gather_wide; also the hub rows of k_gather_wave)
template <int MODE, int LG, int CH>
__device__ __forceinline__ void wide_row(const GatherArgs &A, int row, float4 *part4) {
    constexpr int S = WAVE / LG;                 // edge slots per wave
    constexpr int NSLOT = 4 * S;
    constexpr int U = CH <= 1 ? 8 : (CH <= 2 ? 4 : 2);   // neighbour rows in flight per slot
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int sub = lane / LG, l = lane % LG;
    const int slot = wid * S + sub;
    const int d4 = A.d >> 2;
    int *js = reinterpret_cast<int *>(part4 + 4 * d4);
    float *ws = reinterpret_cast<float *>(js + WIDE_EDGES);
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2, ldy4 = A.ldy >> 2, ldp4 = A.ldp >> 2;
    EdgeCtx<MODE> ctx;
    ctx.init(A, row);
    float4 acc[CH], prow[CH];
    bool ok[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        ok[k] = (l + k * LG) < d4;
        acc[k] = f4_zero();
        prow[k] = f4_zero();
    }
    if (MODE == MODE_RELU) {
        const float4 *P = reinterpret_cast<const float4 *>(A.p) + (int64_t)row * ldp4;
#pragma unroll
        for (int k = 0; k < CH; ++k) if (ok[k]) prow[k] = P[l + k * LG];
    } else if (slot == 0) {
        const float sw = ctx.self_w(A, row);
        if (sw != 0.f) {
            const float4 *Sx = X + (int64_t)row * ldx4;
#pragma unroll
            for (int k = 0; k < CH; ++k) if (ok[k]) { float4 v = Sx[l + k * LG]; f4_fma(acc[k], sw, v); }
        }
    }
    const int e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
    for (int base = e0; base < e1; base += WIDE_EDGES) {
        {   // (a) one edge per thread
            const int ee = base + threadIdx.x;
            int j = row;
            float w = 0.f;
            if (ee < e1) { j = A.col[ee]; w = ctx.edge_w(A, row, j, ee); }
            js[threadIdx.x] = j;
            ws[threadIdx.x] = w;
        }
        __syncthreads();
        const int cnt = (e1 - base) < WIDE_EDGES ? (e1 - base) : WIDE_EDGES;
        for (int t = slot; t < cnt; t += U * NSLOT) {
            int j[U];
            float w[U];
            float4 v[U][CH];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int tt = t + u * NSLOT;
                j[u] = tt < cnt ? js[tt] : row;
                w[u] = tt < cnt ? ws[tt] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (w[u] != 0.f) {
                    const float4 *R = X + (int64_t)j[u] * ldx4;
#pragma unroll
                    for (int k = 0; k < CH; ++k) if (ok[k]) v[u][k] = R[l + k * LG];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (w[u] != 0.f) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        if (!ok[k]) continue;
                        if (MODE == MODE_RELU) {
                            acc[k].x += fmaxf(prow[k].x + v[u][k].x, 0.f);
                            acc[k].y += fmaxf(prow[k].y + v[u][k].y, 0.f);
                            acc[k].z += fmaxf(prow[k].z + v[u][k].z, 0.f);
                            acc[k].w += fmaxf(prow[k].w + v[u][k].w, 0.f);
                        } else if (MODE == MODE_SUM && !A.w) {
                            f4_add(acc[k], v[u][k]);
                        } else {
                            f4_fma(acc[k], w[u], v[u][k]);
                        }
                    }
                }
            }
        }
        __syncthreads();                     // parked edges consumed before the next pass
    }
    // slots of one wave -> lane group 0
#pragma unroll
    for (int k = 0; k < CH; ++k) {
#pragma unroll
        for (int off = LG; off < WAVE; off <<= 1) {
            acc[k].x += __shfl_xor(acc[k].x, off, WAVE);
            acc[k].y += __shfl_xor(acc[k].y, off, WAVE);
            acc[k].z += __shfl_xor(acc[k].z, off, WAVE);
            acc[k].w += __shfl_xor(acc[k].w, off, WAVE);
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < CH; ++k) if (ok[k]) part4[wid * d4 + l + k * LG] = acc[k];
    }
    __syncthreads();
    float4 *Y = reinterpret_cast<float4 *>(A.y) + (int64_t)row * ldy4;
    for (int c4 = threadIdx.x; c4 < d4; c4 += blockDim.x) {
        float4 a = part4[c4], b = part4[d4 + c4], c = part4[2 * d4 + c4], dd = part4[3 * d4 + c4];
        float t[4] = {(a.x + b.x) + (c.x + dd.x), (a.y + b.y) + (c.y + dd.y),
                      (a.z + b.z) + (c.z + dd.z), (a.w + b.w) + (c.w + dd.w)};
        if (MODE == MODE_GCN || MODE == MODE_GAT) {
            const float4 o = A.accumulate ? Y[c4] : f4_zero();
            const float oo[4] = {o.x, o.y, o.z, o.w};
            const Epilogue &E = A.epi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * c4 + q;
                float val = t[q] + (E.bias ? E.bias[c] : 0.f) + oo[q];
                float sl = (E.act == BLGNN_ACT_PRELU && E.act_param) ? E.act_param[c] : 0.f;
                val = apply_act(val, E.act, sl);
                if (E.bn_scale) val = fmaf(val, E.bn_scale[c], E.bn_shift ? E.bn_shift[c] : 0.f);
                t[q] = val;
            }
        }
        Y[c4] = make_float4(t[0], t[1], t[2], t[3]);
    }
    __syncthreads();
}

template <int MODE, int LG, int CH>
__global__ void __launch_bounds__(256) k_gather_wide(const GatherArgs A) {
    extern __shared__ float4 part4[];            // [4][d4] float4, then the parked edges
    const int nrows = A.row_list ? A.n_list : A.n;
    for (int it = blockIdx.x; it < nrows; it += gridDim.x) {
        const int row = A.row_list ? (int)A.row_list[it] : it;         // (a row listed twice is computed twice: same values;
        if (row < 0) continue;                                         //  a negative entry is skipped: a de-duplicated list's padding)
        const int deg = A.rowptr[row + 1] - A.rowptr[row];            // (uniform: scalar loads)
        if (deg < A.deg_min || deg > A.deg_max) continue;
        wide_row<MODE, LG, CH>(A, row, part4);
    }
}

// "One wavefront per row" variant for LARGE high-degree graphs (configs[3]: 100 k interaction nodes, mean degree 21).  The
// workgroup-per-row kernel above spends a row's time in five dependent round trips (rowptr -> col -> scores -> rows ->
// LDS combine) and three barriers with 256 threads for ~21 edges: 500 us for 2.25 GB gathered out of a 102 MB
// Infinity-Cache-resident table = 4.3 TB/s, against 7.4-8.6 TB/s for such gathers (MI355X_MICROARCH.md, "Indexed rows").
// Here a row belongs to ONE wavefront -- no LDS, no barrier, 32 rows of the graph in flight per CU: its lanes compute the
// (source, weight) of up to 64 edges at once, then the S = 64 / LG lane groups walk those edges, U neighbour rows in
// flight per group (row-contiguous 16 B-per-lane loads; with LG = 64 the source index is a scalar and so is the row's
// base address).  Rows above ``deg_max`` edges (hubs: hundreds of sequential round trips for one wavefront; configs[3] has
// 107 of them, up to 4 611 edges) are computed by the first ``hub_blocks`` workgroups of the SAME launch with all 256
// threads per row (as their own launch behind this one they took 180 us after its 358).  The sum of a row is formed in a
// fixed order (edge order within a lane group, groups combined as a xor butterfly): deterministic.
constexpr int WAVE_ROW_HUB_DEGREE = 384;
constexpr int HUB_QUEUE = 1024;                  // rows a hub workgroup scans per round (its LDS queue holds them all)
constexpr int HUB_BLOCKS = 128;

template <int MODE, int LG, int CH>
__global__ void __launch_bounds__(256) k_gather_wave(const GatherArgs A) {
    constexpr int S = WAVE / LG;                 // lane groups = edge slots of the wavefront
    constexpr int U = CH <= 1 ? 12 : (CH <= 2 ? 4 : 2);   // neighbour rows in flight per slot (8: 381 us, 12: 357, 16: 376 at configs[3])
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int sub = lane / LG, l = lane % LG;
    const int d4 = A.d >> 2;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2, ldy4 = A.ldy >> 2;
    const int nrows = A.row_list ? A.n_list : A.n;
    if ((int)blockIdx.x < A.hub_blocks) {
        // ---- hub role: the first hub_blocks workgroups (dispatched first, so the longest rows start first) each scan a
        // slice of the rows for those above deg_max edges and compute them with all 256 threads (wide_row), beside the
        // other workgroups' one-wavefront rows instead of in a launch behind them
        extern __shared__ float4 part4[];                        // wide_row's [4][d4] + parked edges, then the hub queue
        int *queue = reinterpret_cast<int *>(part4 + 4 * d4) + 2 * WIDE_EDGES;      // [0] = count, [1 ..] = rows
        const int per = (nrows + A.hub_blocks - 1) / A.hub_blocks;
        const int lo = blockIdx.x * per, hi = lo + per < nrows ? lo + per : nrows;
        for (int base = lo; base < hi; base += HUB_QUEUE) {
            if (threadIdx.x == 0) queue[0] = 0;
            __syncthreads();
            for (int it = base + threadIdx.x; it < base + HUB_QUEUE && it < hi; it += blockDim.x) {
                const int r = A.row_list ? (int)A.row_list[it] : it;
                if (r >= 0 && A.rowptr[r + 1] - A.rowptr[r] > A.deg_max) queue[1 + atomicAdd(&queue[0], 1)] = r;
            }
            __syncthreads();
            const int found = queue[0];
            for (int q = 0; q < found; ++q) wide_row<MODE, LG, CH>(A, queue[1 + q], part4);     // (ends in a barrier)
            __syncthreads();
        }
        return;
    }
    const int waves = (gridDim.x - A.hub_blocks) * 4;
    bool ok[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) ok[k] = (l + k * LG) < d4;
    for (int it = (blockIdx.x - A.hub_blocks) * 4 + wid; it < nrows; it += waves) {
        const int row = __builtin_amdgcn_readfirstlane(A.row_list ? (int)A.row_list[it] : it);
        if (row < 0) continue;                                  // (padding of a de-duplicated row list)
        const int e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
        if (e1 - e0 > A.deg_max) continue;                      // a hub: the hub role's
        EdgeCtx<MODE> ctx;
        ctx.init(A, row);
        float4 acc[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[k] = f4_zero();
        if (sub == 0) {
            const float sw = ctx.self_w(A, row);
            if (sw != 0.f) {
                const float4 *Sx = X + (int64_t)row * ldx4;
#pragma unroll
                for (int k = 0; k < CH; ++k) if (ok[k]) { const float4 v = Sx[l + k * LG]; f4_fma(acc[k], sw, v); }
            }
        }
        for (int base = e0; base < e1; base += WAVE) {
            // (a) one edge per lane: source and weight (for GAT the score / max / denominator gathers and the exp)
            const int ee = base + lane;
            int j = row;
            float w = 0.f;
            if (ee < e1) { j = A.col[ee]; w = ctx.edge_w(A, row, j, ee); }
            const int cnt = (e1 - base) < WAVE ? (e1 - base) : WAVE;
            // (b) the lane groups walk the edges, U rows in flight each
            for (int t0 = 0; t0 < cnt; t0 += U * S) {
                int jj[U];
                float ww[U];
                float4 v[U][CH];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    // past the last edge: the last edge again with weight 0 -- the loads below stay UNCONDITIONAL (a load
                    // behind a branch makes hipcc wait for every request in flight before the next one: one row in flight
                    // instead of U), and the repeated row is an L1 hit
                    if (S == 1) {
                        const int tt = t0 + u;                                     // uniform: v_readlane, scalar row base
                        const int tc = tt < cnt ? tt : cnt - 1;
                        jj[u] = __builtin_amdgcn_readlane(j, tc);
                        ww[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w), tc));
                        if (tt >= cnt) ww[u] = 0.f;
                    } else {
                        const int tt = t0 + u * S + sub;
                        const int tc = tt < cnt ? tt : cnt - 1;
                        jj[u] = __shfl(j, tc, WAVE);
                        ww[u] = __shfl(w, tc, WAVE);
                        if (tt >= cnt) ww[u] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float4 *R = X + (int64_t)jj[u] * ldx4;
#pragma unroll
                    for (int k = 0; k < CH; ++k) v[u][k] = R[ok[k] ? l + k * LG : 0];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        // (a weight of exactly 0 -- a dropped self loop, a slot past the end -- adds nothing, whatever the row holds)
                        float4 t = acc[k];
                        if (MODE == MODE_SUM && !A.w) f4_add(t, v[u][k]);
                        else f4_fma(t, ww[u], v[u][k]);
                        const bool use = ww[u] != 0.f;
                        acc[k].x = use ? t.x : acc[k].x; acc[k].y = use ? t.y : acc[k].y;
                        acc[k].z = use ? t.z : acc[k].z; acc[k].w = use ? t.w : acc[k].w;
                    }
                }
            }
        }
        // lane groups -> every lane (xor butterfly: fixed order)
#pragma unroll
        for (int k = 0; k < CH; ++k) {
#pragma unroll
            for (int off = LG; off < WAVE; off <<= 1) {
                acc[k].x += __shfl_xor(acc[k].x, off, WAVE);
                acc[k].y += __shfl_xor(acc[k].y, off, WAVE);
                acc[k].z += __shfl_xor(acc[k].z, off, WAVE);
                acc[k].w += __shfl_xor(acc[k].w, off, WAVE);
            }
        }
        if (sub == 0) {
            float4 *Y = reinterpret_cast<float4 *>(A.y) + (int64_t)row * ldy4;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                if (!ok[k]) continue;
                const int c4 = l + k * LG;
                float t[4] = {acc[k].x, acc[k].y, acc[k].z, acc[k].w};
                if (MODE == MODE_GCN || MODE == MODE_GAT) {
                    const float4 o = A.accumulate ? Y[c4] : f4_zero();
                    const float oo[4] = {o.x, o.y, o.z, o.w};
                    const Epilogue &E = A.epi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = 4 * c4 + q;
                        float val = t[q] + (E.bias ? E.bias[c] : 0.f) + oo[q];
                        const float sl = (E.act == BLGNN_ACT_PRELU && E.act_param) ? E.act_param[c] : 0.f;
                        val = apply_act(val, E.act, sl);
                        if (E.bn_scale) val = fmaf(val, E.bn_scale[c], E.bn_shift ? E.bn_shift[c] : 0.f);
                        t[q] = val;
                    }
                }
                Y[c4] = make_float4(t[0], t[1], t[2], t[3]);
            }
        }
    }
}

// scalar fallback for rows whose width / leading dimension / base pointer is not
// float4-compatible: one wave per row, channel-outer loop.  Correctness path.
template <int MODE>
__global__ void __launch_bounds__(256) k_gather_scalar(const GatherArgs A) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int wave_total = (gridDim.x * blockDim.x) / WAVE;
    for (int row = wave_global; row < A.n; row += wave_total) {
        EdgeCtx<MODE> ctx;
        ctx.init(A, row);
        const int e0 = A.rowptr[row], e1 = A.rowptr[row + 1];
        for (int c = lane; c < A.d; c += WAVE) {
            float acc = 0.f, pr = 0.f;
            if (MODE == MODE_RELU) pr = A.p[(int64_t)row * A.ldp + c];
            else {
                float sw = ctx.self_w(A, row);
                if (sw != 0.f) acc = sw * A.x[(int64_t)row * A.ldx + c];
            }
            for (int e = e0; e < e1; ++e) {
                int j = A.col[e];
                float w = ctx.edge_w(A, row, j, e);
                if (w == 0.f) continue;
                float v = A.x[(int64_t)j * A.ldx + c];
                if (MODE == MODE_RELU) acc += fmaxf(pr + v, 0.f);
                else if (MODE == MODE_SUM && !A.w) acc += v;
                else acc = fmaf(w, v, acc);
            }
            float *yp = A.y + (int64_t)row * A.ldy + c;
            if (MODE == MODE_GCN || MODE == MODE_GAT) {
                float val = acc + (A.epi.bias ? A.epi.bias[c] : 0.f) + (A.accumulate ? *yp : 0.f);
                float sl = (A.epi.act == BLGNN_ACT_PRELU && A.epi.act_param) ? A.epi.act_param[c] : 0.f;
                val = apply_act(val, A.epi.act, sl);
                if (A.epi.bn_scale) val = fmaf(val, A.epi.bn_scale[c], A.epi.bn_shift ? A.epi.bn_shift[c] : 0.f);
                acc = val;
            }
            *yp = acc;
        }
    }
}

template <int MODE, int LG, int CH>
static void launch_vec(const GatherArgs &A, hipStream_t s) {
    constexpr int RPW = WAVE / LG;
    const int64_t waves = ((int64_t)A.n + RPW - 1) / RPW;
    static const int cap_env = [] { const char *e = getenv("BLGNN_GATHER_VEC_BLOCKS"); return e ? atoi(e) : 0; }();    // (A/B switch)
    const int64_t cap = cap_env > 0 ? cap_env : 256 * 16;
    int64_t blocks = (waves + 3) / 4;
    if (blocks > cap) blocks = cap;             // grid-stride beyond 16 blocks per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_gather<MODE, LG, CH>), dim3((unsigned)blocks), dim3(256), 0, s, A);
}

template <int MODE, int LG, int CH>
static void launch_wide(const GatherArgs &A, hipStream_t s) {
    const int rows = A.row_list ? A.n_list : A.n;
    const int blocks = rows < 256 * 32 ? rows : 256 * 32;
    hipLaunchKernelGGL((k_gather_wide<MODE, LG, CH>), dim3(blocks), dim3(256),
                       (size_t)A.d * 16 + WIDE_EDGES * 8, s, A);
}

// one launch: one wavefront per row up to WAVE_ROW_HUB_DEGREE edges, the rows above by the hub workgroups
template <int MODE, int LG, int CH>
static void launch_wave(GatherArgs A, hipStream_t s) {
    const int rows = A.row_list ? A.n_list : A.n;
    static const int hub_env = [] { const char *e = getenv("BLGNN_GATHER_HUB_DEGREE"); return e ? atoi(e) : 0; }();     // (A/B switch)
    static const int hubb_env = [] { const char *e = getenv("BLGNN_GATHER_HUB_BLOCKS"); return e ? atoi(e) : 0; }();    // (A/B switch)
    const int hub_blocks = hubb_env > 0 ? hubb_env : HUB_BLOCKS;
    A.deg_min = 0; A.deg_max = hub_env > 0 ? hub_env : WAVE_ROW_HUB_DEGREE; A.hub_blocks = hub_blocks;
    static const int cap_env = [] { const char *e = getenv("BLGNN_GATHER_WAVE_BLOCKS"); return e ? atoi(e) : 0; }();   // (A/B switch)
    const int cap = cap_env > 0 ? cap_env : 256 * 32;          // (configs[3]: 1 024 blocks 354 us, 2 048 344, 4 096 338, 8 192 331, 32 768 335)
    const int blocks = hub_blocks + ((rows + 3) / 4 < cap ? (rows + 3) / 4 : cap);
    hipLaunchKernelGGL((k_gather_wave<MODE, LG, CH>), dim3(blocks), dim3(256),
                       (size_t)A.d * 16 + WIDE_EDGES * 8 + (HUB_QUEUE + 1) * sizeof(int), s, A);
}

// rows this heavy on average go one-workgroup-per-row (see k_gather_wide) ...
constexpr int64_t WIDE_MEAN_DEGREE = 12;
// ... and, on graphs with at least this many rows to compute, one-wavefront-per-row with the hubs behind it (k_gather_wave);
// below it a launch has fewer rows than the chip has wavefront slots and the workgroup-per-row kernel's 256 threads per
// row are the parallelism there is (DrugBank interaction level: 1 326 rows)
// (a row LIST of a large graph -- a rank's own target rows, the rows a batch scores -- takes the wavefront kernel from 8 192 rows:
// 12.5 k listed rows of configs[3]: 122 us workgroup-per-row; BLGNN_GATHER_WAVE_MIN_ROWS overrides both for A/B runs)
constexpr int WAVE_ROW_MIN_ROWS = 16384;
constexpr int WAVE_LIST_MIN_ROWS = 8192;

// row magnitudes of a finished result merged into amax by atomic maximum (fallback of the variants without an amax epilogue)
__global__ void __launch_bounds__(256) k_row_absmax_merge(const float *__restrict__ y, int64_t ldy, int n, int d,
                                                          float *__restrict__ amax) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int row = blockIdx.x * 4 + threadIdx.x / WAVE;
    if (row >= n) return;
    float m = 0.f;
    for (int c = lane; c < d; c += WAVE) m = fmaxf(m, fabsf(y[(int64_t)row * ldy + c]));
    m = group_max(m, WAVE);
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(amax) + row, __float_as_uint(m));
}

template <int MODE>
static int dispatch(const GatherArgs &A, int64_t nnz, bool extra_aligned, hipStream_t s, const char *what,
                    bool *amax_native = nullptr) {
    if (amax_native) *amax_native = false;
    if (A.n == 0 || A.d == 0 || (A.row_list && A.n_list == 0)) return BLGNN_OK;
    const bool vec = (A.d % 4 == 0) && (A.ldx % 4 == 0) && (A.ldy % 4 == 0) &&
                     aligned16(A.x) && aligned16(A.y) && extra_aligned;
    const int d4 = A.d / 4;
    static const bool wave_rows = [] { const char *e = getenv("BLGNN_GATHER_WAVE"); return !(e && e[0] == '0'); }();   // (A/B switch)
    static const int min_env = [] { const char *e = getenv("BLGNN_GATHER_WAVE_MIN_ROWS"); return e ? atoi(e) : 0; }();
    const int wave_min = min_env > 0 ? min_env : (A.row_list && A.n >= WAVE_ROW_MIN_ROWS ? WAVE_LIST_MIN_ROWS : WAVE_ROW_MIN_ROWS);
    if (wave_rows && MODE != MODE_RELU && vec && d4 <= 128 && nnz >= WIDE_MEAN_DEGREE * (int64_t)A.n &&
        (A.row_list ? A.n_list : A.n) >= wave_min) {
        if (d4 <= 16) launch_wave<MODE, 16, 1>(A, s);
        else if (d4 <= 32) launch_wave<MODE, 32, 1>(A, s);
        else if (d4 <= 64) launch_wave<MODE, 64, 1>(A, s);
        else launch_wave<MODE, 64, 2>(A, s);
    } else if (vec && d4 <= 512 && nnz >= WIDE_MEAN_DEGREE * (int64_t)A.n) {
        if (d4 <= 16) launch_wide<MODE, 16, 1>(A, s);
        else if (d4 <= 32) launch_wide<MODE, 32, 1>(A, s);
        else if (d4 <= 64) launch_wide<MODE, 64, 1>(A, s);
        else if (d4 <= 128) launch_wide<MODE, 64, 2>(A, s);
        else if (d4 <= 192) launch_wide<MODE, 64, 3>(A, s);
        else if (d4 <= 256) launch_wide<MODE, 64, 4>(A, s);
        else if (d4 <= 320) launch_wide<MODE, 64, 5>(A, s);
        else launch_wide<MODE, 64, 8>(A, s);
    } else if (MODE == MODE_SUM && vec && d4 <= 64 && nnz <= 4 * (int64_t)A.n) {   // molecular graphs
        // (from here on a row_list is ignored: these kernels compute every row, a superset)
        if (d4 <= 16) launch_spmm_tile<16>(A, s);
        else if (d4 <= 32) launch_spmm_tile<32>(A, s);
        else launch_spmm_tile<64>(A, s);
    } else if (vec && d4 <= 64 && nnz <= 4 * (int64_t)A.n) {
        if (d4 <= 16) launch_rows4<MODE, 16>(A, s);
        else if (d4 <= 32) launch_rows4<MODE, 32>(A, s);
        else launch_rows4<MODE, 64>(A, s);
    } else if (vec && d4 <= 512) {
        if (amax_native) *amax_native = true;
        if (d4 <= 16) launch_vec<MODE, 16, 1>(A, s);
        else if (d4 <= 32) launch_vec<MODE, 32, 1>(A, s);
        else if (d4 <= 64) launch_vec<MODE, 64, 1>(A, s);
        else if (d4 <= 128) launch_vec<MODE, 64, 2>(A, s);
        else if (d4 <= 192) launch_vec<MODE, 64, 3>(A, s);
        else if (d4 <= 256) launch_vec<MODE, 64, 4>(A, s);
        else if (d4 <= 320) launch_vec<MODE, 64, 5>(A, s);
        else launch_vec<MODE, 64, 8>(A, s);
    } else {
        int64_t blocks = ((int64_t)A.n + 3) / 4;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL((k_gather_scalar<MODE>), dim3((unsigned)blocks), dim3(256), 0, s, A);
    }
    return launch_status(what);
}

// ---- GCN degree ---------------------------------------------------------------
__global__ void k_gcn_dis(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                          const float *__restrict__ w, int32_t n, float *__restrict__ dis) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float deg = 1.f;                                   // appended self loop, weight 1
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e)
        if (col[e] != i) deg += w ? w[e] : 1.f;
    float r = 1.f / sqrtf(deg);
    dis[i] = isinf(r) ? 0.f : r;
}

__global__ void k_row_degree(const int32_t *__restrict__ rowptr, int32_t n, float *__restrict__ deg) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) deg[i] = (float)(rowptr[i + 1] - rowptr[i]);
}

// ---- GAT scores: 16 lanes per row ---------------------------------------------
__global__ void __launch_bounds__(256)
k_gat_scores(const float *__restrict__ h, int64_t ldh, const float *__restrict__ att, int32_t n,
             int32_t d, float *__restrict__ s_dst, float *__restrict__ s_src) {
    constexpr int LG = 16;
    const int lane = threadIdx.x & (WAVE - 1), l = lane % LG;
    const int row0 = (blockIdx.x * blockDim.x + threadIdx.x) / LG;
    const int step = gridDim.x * blockDim.x / LG;
    for (int row = row0; row < n; row += step) {
        const float *hr = h + (int64_t)row * ldh;
        float a = 0.f, b = 0.f;
        for (int c = l; c < d; c += LG) {
            float v = hr[c];
            a = fmaf(v, att[c], a);
            b = fmaf(v, att[d + c], b);
        }
        a = group_sum(a, LG);
        b = group_sum(b, LG);
        if (l == 0) { s_dst[row] = a; s_src[row] = b; }
    }
}

// ---- GAT softmax statistics: 16 lanes per group node ----------------------------
template <int LG>
__global__ void __launch_bounds__(256)
k_gat_stats(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
            const float *__restrict__ s_dst, const float *__restrict__ s_src, int32_t n,
            int32_t group, float slope, float *__restrict__ mx, float *__restrict__ den, float4 *__restrict__ packed) {
    const int l = threadIdx.x % LG;
    const int g0 = (blockIdx.x * blockDim.x + threadIdx.x) / LG;
    const int step = gridDim.x * blockDim.x / LG;
    for (int g = g0; g < n; g += step) {
        const int e0 = rowptr[g], e1 = rowptr[g + 1];
        const float sd = s_dst[g], ss = s_src[g];
        const float self_a = lrelu(sd + ss, slope);
        float m = self_a;
        for (int e = e0 + l; e < e1; e += LG) {
            int k = col[e];
            if (k == g) continue;
            float a = group == BLGNN_GAT_GROUP_DST ? lrelu(sd + s_src[k], slope)
                                                   : lrelu(s_dst[k] + ss, slope);
            m = fmaxf(m, a);
        }
        m = group_max(m, LG);
        float sum = (l == 0) ? expf(self_a - m) : 0.f;
        for (int e = e0 + l; e < e1; e += LG) {
            int k = col[e];
            if (k == g) continue;
            float a = group == BLGNN_GAT_GROUP_DST ? lrelu(sd + s_src[k], slope)
                                                   : lrelu(s_dst[k] + ss, slope);
            sum += expf(a - m);
        }
        sum = group_sum(sum, LG);
        if (l == 0) {
            mx[g] = m; den[g] = sum;
            if (packed) packed[g] = make_float4(ss, m, sum, 0.f);
        }
    }
}

}  // namespace blgnn

using namespace blgnn;

static int check_csr(const int32_t *rowptr, const int32_t *col, int32_t n, int64_t nnz, const char *what) {
    if (n < 0 || nnz < 0) return fail(BLGNN_ERR_ARG, "%s: negative size", what);
    if (n > 0 && (!rowptr || (nnz > 0 && !col))) return fail(BLGNN_ERR_ARG, "%s: null CSR", what);
    return BLGNN_OK;
}

static int grid1d(int64_t items, int per_block) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int blgnn_spmm_sum(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *w,
                              const float *x, int64_t ldx, float *y, int64_t ldy, int32_t n,
                              int32_t d, float self_coef, int32_t skip_self, void *stream) {
    if (int rc = check_csr(rowptr, col, n, nnz, "spmm_sum")) return rc;
    BLGNN_CHECK_ARG(d >= 0 && (n == 0 || d == 0 || (x && y)), "spmm_sum: null x/y");
    BLGNN_CHECK_ARG(ldx >= d && ldy >= d, "spmm_sum: leading dimension < d");
    BLGNN_CHECK_ARG(x != y, "spmm_sum: x and y must not alias");
    GatherArgs A = {};
    A.rowptr = rowptr; A.col = col; A.x = x; A.ldx = ldx; A.y = y; A.ldy = ldy; A.n = n; A.d = d;
    A.w = w; A.self_coef = self_coef; A.skip_self = skip_self;
    return dispatch<MODE_SUM>(A, nnz, true, (hipStream_t)stream, "spmm_sum");
}

extern "C" int blgnn_spmm_sum_sp(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x, int64_t ldx,
                                 int32_t n, int32_t d, float self_coef, int32_t skip_self, void *y_sp, int64_t ld_ysp,
                                 uint8_t *y_spe, int64_t ld_yspe, void *stream) {
    if (int rc = check_csr(rowptr, col, n, nnz, "spmm_sum_sp")) return rc;
    BLGNN_CHECK_ARG(d >= 0 && (n == 0 || d == 0 || (x && y_sp && y_spe)), "spmm_sum_sp: null pointer");
    BLGNN_CHECK_ARG(ldx >= d && ld_ysp >= 2 * (int64_t)d && ld_ysp % 8 == 0 && aligned16(y_sp) && ld_yspe >= d / SP_BLOCK,
                    "spmm_sum_sp: bad leading dimensions / sp output planes");
    if (d % SP_BLOCK != 0 || d > 2048 || ldx % 4 != 0 || !aligned16(x))
        return fail(BLGNN_ERR_UNSUPPORTED, "spmm_sum_sp: needs d %% 256 == 0, d <= 2048 and 16-byte aligned rows (d=%d)", d);
    if (n == 0 || d == 0) return BLGNN_OK;
    GatherArgs A = {};
    A.rowptr = rowptr; A.col = col; A.x = x; A.ldx = ldx; A.y = nullptr; A.ldy = d; A.n = n; A.d = d;
    A.self_coef = self_coef; A.skip_self = skip_self;
    A.ys = reinterpret_cast<_Float16 *>(y_sp); A.ldys = ld_ysp; A.ye = y_spe; A.ldye = ld_yspe;
    hipStream_t s = (hipStream_t)stream;
    // the sp epilogue lives in the one-wavefront-per-row kernel (a chunk of 64 lanes x 4 columns = one scale block)
    const int d4 = d / 4;
    if (d4 <= 64) launch_vec<MODE_SUM, 64, 1>(A, s);
    else if (d4 <= 128) launch_vec<MODE_SUM, 64, 2>(A, s);
    else if (d4 <= 192) launch_vec<MODE_SUM, 64, 3>(A, s);
    else if (d4 <= 256) launch_vec<MODE_SUM, 64, 4>(A, s);
    else if (d4 <= 320) launch_vec<MODE_SUM, 64, 5>(A, s);
    else launch_vec<MODE_SUM, 64, 8>(A, s);
    return launch_status("spmm_sum_sp");
}

extern "C" int blgnn_gcn_deg_inv_sqrt(const int32_t *rowptr, const int32_t *col, const float *w,
                                      int32_t n, float *dis, void *stream) {
    if (int rc = check_csr(rowptr, col, n, 0, "gcn_deg_inv_sqrt")) return rc;
    BLGNN_CHECK_ARG(n == 0 || dis, "gcn_deg_inv_sqrt: null dis");
    if (n == 0) return BLGNN_OK;
    hipLaunchKernelGGL(k_gcn_dis, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       rowptr, col, w, n, dis);
    return launch_status("gcn_deg_inv_sqrt");
}

extern "C" int blgnn_row_degree(const int32_t *rowptr, int32_t n, float *deg, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && (n == 0 || (rowptr && deg)), "row_degree: bad args");
    if (n == 0) return BLGNN_OK;
    hipLaunchKernelGGL(k_row_degree, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       rowptr, n, deg);
    return launch_status("row_degree");
}

extern "C" int blgnn_gcn_aggregate(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *w,
                                   const float *dis, const float *h, int64_t ldh, float *y,
                                   int64_t ldy, int32_t n, int32_t d, const float *bias,
                                   int32_t act, const float *act_param, const float *bn_scale,
                                   const float *bn_shift, int32_t accumulate, void *stream) {
    if (int rc = check_csr(rowptr, col, n, nnz, "gcn_aggregate")) return rc;
    BLGNN_CHECK_ARG(d >= 0 && (n == 0 || d == 0 || (h && y && dis)), "gcn_aggregate: null pointer");
    BLGNN_CHECK_ARG(ldh >= d && ldy >= d, "gcn_aggregate: leading dimension < d");
    BLGNN_CHECK_ARG(h != y, "gcn_aggregate: h and y must not alias");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "gcn_aggregate: bad activation id");
    GatherArgs A = {};
    A.rowptr = rowptr; A.col = col; A.x = h; A.ldx = ldh; A.y = y; A.ldy = ldy; A.n = n; A.d = d;
    A.w = w; A.dis = dis;
    A.epi = Epilogue{bias, act_param, bn_scale, bn_shift, act};
    A.accumulate = accumulate;
    return dispatch<MODE_GCN>(A, nnz, true, (hipStream_t)stream, "gcn_aggregate");
}

extern "C" int blgnn_gat_scores(const float *h, int64_t ldh, const float *att, int32_t n,
                                int32_t d, float *s_dst, float *s_src, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && d > 0 && ldh >= d, "gat_scores: bad sizes");
    BLGNN_CHECK_ARG(n == 0 || (h && att && s_dst && s_src), "gat_scores: null pointer");
    if (n == 0) return BLGNN_OK;
    hipLaunchKernelGGL(k_gat_scores, dim3(grid1d(n, 16)), dim3(256), 0, (hipStream_t)stream, h, ldh,
                       att, n, d, s_dst, s_src);
    return launch_status("gat_scores");
}

extern "C" int blgnn_gat_softmax_stats(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                                       const float *s_dst, const float *s_src, int32_t n,
                                       int32_t group, float slope, float *mx, float *den, float *stats_packed,
                                       void *stream) {
    if (int rc = check_csr(rowptr, col, n, nnz, "gat_softmax_stats")) return rc;
    BLGNN_CHECK_ARG(group == BLGNN_GAT_GROUP_SRC || group == BLGNN_GAT_GROUP_DST,
                    "gat_softmax_stats: bad group");
    BLGNN_CHECK_ARG(n == 0 || (s_dst && s_src && mx && den), "gat_softmax_stats: null pointer");
    BLGNN_CHECK_ARG(aligned16(stats_packed), "gat_softmax_stats: stats_packed must be 16-byte aligned");
    if (n == 0) return BLGNN_OK;
    float4 *packed = reinterpret_cast<float4 *>(stats_packed);
    if (nnz >= WIDE_MEAN_DEGREE * (int64_t)n)      // high-degree graph: a whole wavefront per node
        hipLaunchKernelGGL(k_gat_stats<64>, dim3(grid1d(n, 4)), dim3(256), 0, (hipStream_t)stream, rowptr,
                           col, s_dst, s_src, n, group, slope, mx, den, packed);
    else
        hipLaunchKernelGGL(k_gat_stats<16>, dim3(grid1d(n, 16)), dim3(256), 0, (hipStream_t)stream, rowptr,
                           col, s_dst, s_src, n, group, slope, mx, den, packed);
    return launch_status("gat_softmax_stats");
}

extern "C" int blgnn_gat_aggregate(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *h,
                                   int64_t ldh, const float *s_dst, const float *s_src,
                                   const float *mx, const float *den, int32_t group, float slope,
                                   float *y, int64_t ldy, int32_t n, int32_t d, const float *bias,
                                   int32_t act, const float *act_param, const float *bn_scale,
                                   const float *bn_shift, int32_t accumulate, const int64_t *rows, int32_t num_rows,
                                   const float *stats_packed, void *stream) {
    if (int rc = check_csr(rowptr, col, n, nnz, "gat_aggregate")) return rc;
    BLGNN_CHECK_ARG(aligned16(stats_packed), "gat_aggregate: stats_packed must be 16-byte aligned");
    BLGNN_CHECK_ARG(!rows || num_rows >= 0, "gat_aggregate: negative row count");
    BLGNN_CHECK_ARG(group == BLGNN_GAT_GROUP_SRC || group == BLGNN_GAT_GROUP_DST,
                    "gat_aggregate: bad group");
    BLGNN_CHECK_ARG(d >= 0 && (n == 0 || d == 0 || (h && y && s_dst && s_src && mx && den)),
                    "gat_aggregate: null pointer");
    BLGNN_CHECK_ARG(ldh >= d && ldy >= d, "gat_aggregate: leading dimension < d");
    BLGNN_CHECK_ARG(h != y, "gat_aggregate: h and y must not alias");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "gat_aggregate: bad activation id");
    GatherArgs A = {};
    A.rowptr = rowptr; A.col = col; A.x = h; A.ldx = ldh; A.y = y; A.ldy = ldy; A.n = n; A.d = d;
    A.s_dst = s_dst; A.s_src = s_src; A.mx = mx; A.den = den; A.group = group; A.slope = slope;
    A.epi = Epilogue{bias, act_param, bn_scale, bn_shift, act};
    A.accumulate = accumulate;
    A.row_list = rows; A.n_list = rows ? num_rows : 0;
    A.packed = reinterpret_cast<const float4 *>(stats_packed);
    return dispatch<MODE_GAT>(A, nnz, true, (hipStream_t)stream, "gat_aggregate");
}

extern "C" int blgnn_edge_relu_sum(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *p,
                                   int64_t ldp, const float *q, int64_t ldq, float *y,
                                   int64_t ldy, int32_t n, int32_t d, float *row_amax, void *y_sp, int64_t ld_ysp,
                                   uint8_t *y_spe, int64_t ld_yspe, void *stream) {
    if (int rc = check_csr(rowptr, col, n, nnz, "edge_relu_sum")) return rc;
    BLGNN_CHECK_ARG(d >= 0 && (n == 0 || d == 0 || (p && q && (y || y_sp))), "edge_relu_sum: null pointer");
    BLGNN_CHECK_ARG(ldp >= d && ldq >= d && (!y || ldy >= d), "edge_relu_sum: leading dimension < d");
    BLGNN_CHECK_ARG(y != p && y != q, "edge_relu_sum: output must not alias inputs");
    GatherArgs A = {};
    A.rowptr = rowptr; A.col = col; A.x = q; A.ldx = ldq; A.y = y; A.ldy = y ? ldy : d; A.n = n; A.d = d;
    A.p = p; A.ldp = ldp;
    A.amax = row_amax;
    const bool pa = (ldp % 4 == 0) && aligned16(p);
    if (y_sp) {
        // the sp epilogue lives in the one-wavefront-per-row kernel (a chunk of 64 lanes x 4 columns = one scale block)
        BLGNN_CHECK_ARG(y_spe && ld_ysp >= 2 * (int64_t)d && ld_ysp % 8 == 0 && aligned16(y_sp) && ld_yspe >= d / SP_BLOCK,
                        "edge_relu_sum: bad sp output planes");
        if (d % SP_BLOCK != 0 || d > 2048 || !pa || ldq % 4 != 0 || !aligned16(q) || (y && (ldy % 4 != 0 || !aligned16(y))))
            return fail(BLGNN_ERR_UNSUPPORTED, "edge_relu_sum: sp output needs d %% 256 == 0, d <= 2048 and 16-byte aligned rows (d=%d)", d);
        if (n == 0 || d == 0) return BLGNN_OK;
        A.ys = reinterpret_cast<_Float16 *>(y_sp); A.ldys = ld_ysp; A.ye = y_spe; A.ldye = ld_yspe;
        hipStream_t s = (hipStream_t)stream;
        const int d4 = d / 4;
        if (d4 <= 64) launch_vec<MODE_RELU, 64, 1>(A, s);
        else if (d4 <= 128) launch_vec<MODE_RELU, 64, 2>(A, s);
        else if (d4 <= 192) launch_vec<MODE_RELU, 64, 3>(A, s);
        else if (d4 <= 256) launch_vec<MODE_RELU, 64, 4>(A, s);
        else if (d4 <= 320) launch_vec<MODE_RELU, 64, 5>(A, s);
        else launch_vec<MODE_RELU, 64, 8>(A, s);
        return launch_status("edge_relu_sum");
    }
    bool native = false;
    const int rc = dispatch<MODE_RELU>(A, nnz, pa, (hipStream_t)stream, "edge_relu_sum", &native);
    if (rc != BLGNN_OK || !row_amax || native || n == 0 || d == 0) return rc;
    hipLaunchKernelGGL(k_row_absmax_merge, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, ldy, n, d, row_amax);
    return launch_status("edge_relu_sum");
}
