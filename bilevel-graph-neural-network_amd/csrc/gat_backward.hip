// Backward of the graph half of GATConv (heads = 1): model/layers.py:33,54 under src/train.py:177-182.
//
// Forward (gather.hip):  a_ij = leaky_relu(s_dst[i] + s_src[j]),  alpha_ij = exp(a_ij - mx[g]) / (den[g] + 1e-16) with
// g = j (SRC grouping) or i (DST),  z_i = sum_{j in N(i) u {i}} alpha_ij h_j  (existing self loops dropped, one appended).
// Given dz:
//   d alpha_ij = dz_i . h_j                                  T_g = sum_{e in g} alpha_e d alpha_e
//   d a_ij     = alpha_ij (d alpha_ij - T_g)                 d pre_ij = d a_ij * (pre_ij > 0 ? 1 : slope)
//   ds_dst[i] += d pre_ij,  ds_src[j] += d pre_ij
//   dh_j       = sum_i alpha_ij dz_i + ds_dst[j] att[0:d] + ds_src[j] att[d:2d]
// Two launches, one workgroup per row, no d-wide atomics:
//   k_gat_bwd_scores    GROUP-major: all edges of a softmax group are the CSR row of its node (the graphs of this path are
//                       symmetric: model/layers_util.py:133), so T_g and the group side's ds are local sums; the other
//                       side's ds receives one scalar float atomic per edge;
//   k_gat_bwd_features  SOURCE-major: dh_j gathers alpha_ij dz_i over the row of j.
// The dot products are recomputed in the second sweep over a row rather than parked (rows of the interaction graph have
// tens to hundreds of edges; h and dz of a whole interaction graph sit in L2).
#include "common.hpp"

namespace blgnn {

struct GatBwdArgs {
    const int32_t *rowptr, *col;
    const float *h; int64_t ldh;
    const float *dz; int64_t lddz;
    const float *s_dst, *s_src, *mx, *den;
    const float *att;
    float *ds_dst, *ds_src;
    float *dh; int64_t lddh;
    int32_t n, d, group_src;
    float slope;
};

__device__ __forceinline__ float gat_alpha(const GatBwdArgs &A, int i, int j, int g, float &lrelu_grad) {
    const float pre = A.s_dst[i] + A.s_src[j];
    lrelu_grad = pre > 0.f ? 1.f : A.slope;
    const float a = pre > 0.f ? pre : A.slope * pre;
    return expf(a - A.mx[g]) / (A.den[g] + 1e-16f);
}

// dz_i . h_j over d features by the 16 lanes of an edge group (l = lane in the group); all 16 get the result
__device__ __forceinline__ float gat_dot16(const GatBwdArgs &A, int i, int j, int l, bool vec) {
    const float *a = A.dz + (int64_t)i * A.lddz, *b = A.h + (int64_t)j * A.ldh;
    float s = 0.f;
    if (vec) {
        for (int c = 4 * l; c < A.d; c += 64) {
            const float4 u = *reinterpret_cast<const float4 *>(a + c), v = *reinterpret_cast<const float4 *>(b + c);
            s = fmaf(u.x, v.x, s); s = fmaf(u.y, v.y, s); s = fmaf(u.z, v.z, s); s = fmaf(u.w, v.w, s);
        }
    } else {
        for (int c = l; c < A.d; c += 16) s = fmaf(a[c], b[c], s);
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) s += __shfl_xor(s, off, WAVE);
    return s;
}

constexpr int GB_PARK = 2048;           // edges of a group whose (alpha, d alpha, slope factor) stay in LDS between the sweeps

// One WORKGROUP per softmax group (= node r, its CSR row + the appended self loop), sixteen edges in flight (16 lanes per
// edge).  (One wavefront per group walked its edges one after the other, a 128-wide dot product and four dependent loads
// each: the hub rows of the DrugBank interaction graph, 500+ edges, made the launch 970 us.)  Sums over the group are
// formed in a fixed order; the other side's ds receives one scalar float atomic per edge as before.
__global__ void __launch_bounds__(256) k_gat_bwd_scores(const GatBwdArgs A) {
    __shared__ float s_al[GB_PARK], s_da[GB_PARK], s_lg[GB_PARK];
    __shared__ float red[16];
    const int r = blockIdx.x;                                                   // the group's node
    const int tid = threadIdx.x, sub = tid >> 4, l = tid & 15;
    const int e0 = A.rowptr[r], e1 = A.rowptr[r + 1];
    const int ne = e1 - e0 + 1;
    const bool park = ne <= GB_PARK;
    const bool vec = (A.d % 4 == 0) && (A.ldh % 4 == 0) && (A.lddz % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(A.h) | reinterpret_cast<uintptr_t>(A.dz)) & 15u) == 0;
    // edge list of the group: CSR partners k != r, then the appended self loop (k = r)
    auto edge = [&](int e, int &i, int &j) -> bool {
        const int k = e < e1 ? A.col[e] : r;
        if (e < e1 && k == r) return false;                                     // existing self loop: dropped
        i = A.group_src ? k : r;                                                // target
        j = A.group_src ? r : k;                                                // source
        return true;
    };
    float tp = 0.f;                                                             // this edge group's part of T (lane 0 of it)
    for (int e = e0 + sub; e <= e1; e += 16) {
        int i = 0, j = 0;
        float al = 0.f, da = 0.f, lg = 0.f;
        if (edge(e, i, j)) {
            al = gat_alpha(A, i, j, r, lg);
            da = gat_dot16(A, i, j, l, vec);
        }
        if (park && l == 0) { s_al[e - e0] = al; s_da[e - e0] = da; s_lg[e - e0] = lg; }
        tp = fmaf(al, da, tp);
    }
    if (l == 0) red[sub] = tp;
    __syncthreads();
    float T = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) T += red[q];                                   // fixed order
    __syncthreads();
    // second sweep: d pre per edge; one thread per parked edge, else the edge groups recompute
    float local = 0.f;
    if (park) {
        for (int t = tid; t < ne; t += 256) {
            const float lg = s_lg[t];
            if (lg == 0.f) continue;                                            // dropped self loop
            const int e = e0 + t;
            const int k = e < e1 ? A.col[e] : r;
            const float dpre = s_al[t] * (s_da[t] - T) * lg;
            local += dpre;
            unsafeAtomicAdd(A.group_src ? A.ds_dst + k : A.ds_src + k, dpre);   // the other side of the edge
        }
    } else {
        for (int e = e0 + sub; e <= e1; e += 16) {
            int i = 0, j = 0;
            if (!edge(e, i, j)) continue;
            float lg;
            const float al = gat_alpha(A, i, j, r, lg);
            const float dpre = al * (gat_dot16(A, i, j, l, vec) - T) * lg;
            if (l == 0) {
                local += dpre;
                unsafeAtomicAdd(A.group_src ? A.ds_dst + i : A.ds_src + j, dpre);
            }
        }
    }
    // the group's own side: sum of d pre over its edges, fixed order (lanes, then wavefronts)
    local = group_sum(local, WAVE);
    if ((tid & 63) == 0) red[tid >> 6] = local;
    __syncthreads();
    if (tid == 0) (A.group_src ? A.ds_src : A.ds_dst)[r] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One workgroup per source node j: its four wavefronts take every fourth edge of the row (two edges in flight each),
// lanes across features; the four partial rows are added in a fixed order.
__global__ void __launch_bounds__(256) k_gat_bwd_features(const GatBwdArgs A) {
    __shared__ float part[4][4 * WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    const int j = blockIdx.x;                                                   // source node
    const int e0 = A.rowptr[j], e1 = A.rowptr[j + 1];
    const float sd = A.ds_dst[j], ss = A.ds_src[j];
    auto alpha_of = [&](int e, int &i) -> float {
        i = e < e1 ? A.col[e] : j;                                              // target of the edge j -> i
        if (e > e1 || (e < e1 && i == j)) { i = j; return 0.f; }                // past the row / dropped self loop
        float lg;
        return gat_alpha(A, i, j, A.group_src ? j : i, lg);
    };
    for (int c0 = 0; c0 < A.d; c0 += 4 * WAVE) {                                // 4 features per lane and sweep
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = e0 + wid; e <= e1; e += 8) {
            int i0, i1;
            const float a0 = alpha_of(e, i0), a1 = alpha_of(e + 4, i1);
            const float *g0 = A.dz + (int64_t)i0 * A.lddz, *g1 = A.dz + (int64_t)i1 * A.lddz;
            float v0[4], v1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + lane + q * WAVE;
                v0[q] = c < A.d ? g0[c] : 0.f;
                v1[q] = c < A.d ? g1[c] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc[q] = fmaf(a0, v0[q], acc[q]); acc[q] = fmaf(a1, v1[q], acc[q]); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) part[wid][lane + q * WAVE] = acc[q];
        __syncthreads();
        for (int t = threadIdx.x; t < 4 * WAVE; t += 256) {
            const int c = c0 + t;
            if (c < A.d)
                A.dh[(int64_t)j * A.lddh + c] = ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t])) +
                                                sd * A.att[c] + ss * A.att[A.d + c];
        }
        __syncthreads();
    }
}

}  // namespace blgnn

using namespace blgnn;

static int gat_bwd_check(const int32_t *rowptr, const int32_t *col, int64_t nnz, int32_t n, int32_t d, int32_t group,
                         const char *what) {
    BLGNN_CHECK_ARG(n >= 0 && d > 0 && nnz >= 0, "%s: bad sizes", what);
    BLGNN_CHECK_ARG(group == BLGNN_GAT_GROUP_SRC || group == BLGNN_GAT_GROUP_DST, "%s: bad softmax group", what);
    BLGNN_CHECK_ARG(n == 0 || (rowptr && (col || nnz == 0)), "%s: null graph", what);
    return BLGNN_OK;
}

extern "C" int blgnn_gat_backward_scores(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *h,
                                         int64_t ldh, const float *dz, int64_t lddz, const float *s_dst,
                                         const float *s_src, const float *mx, const float *den, int32_t n, int32_t d,
                                         int32_t group, float slope, float *ds_dst, float *ds_src, void *stream) {
    const int rc = gat_bwd_check(rowptr, col, nnz, n, d, group, "gat_backward_scores");
    if (rc != BLGNN_OK) return rc;
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(h && dz && s_dst && s_src && mx && den && ds_dst && ds_src, "gat_backward_scores: null pointer");
    BLGNN_CHECK_ARG(ldh >= d && lddz >= d, "gat_backward_scores: leading dimension < d");
    GatBwdArgs A{};
    A.rowptr = rowptr; A.col = col; A.h = h; A.ldh = ldh; A.dz = dz; A.lddz = lddz;
    A.s_dst = s_dst; A.s_src = s_src; A.mx = mx; A.den = den; A.ds_dst = ds_dst; A.ds_src = ds_src;
    A.n = n; A.d = d; A.group_src = group == BLGNN_GAT_GROUP_SRC; A.slope = slope;
    hipLaunchKernelGGL(k_gat_bwd_scores, dim3(n), dim3(256), 0, (hipStream_t)stream, A);
    return launch_status("gat_backward_scores");
}

extern "C" int blgnn_gat_backward_features(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *dz,
                                           int64_t lddz, const float *s_dst, const float *s_src, const float *mx,
                                           const float *den, const float *ds_dst, const float *ds_src,
                                           const float *att, int32_t n, int32_t d, int32_t group, float slope,
                                           float *dh, int64_t lddh, void *stream) {
    const int rc = gat_bwd_check(rowptr, col, nnz, n, d, group, "gat_backward_features");
    if (rc != BLGNN_OK) return rc;
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(dz && s_dst && s_src && mx && den && ds_dst && ds_src && att && dh, "gat_backward_features: null pointer");
    BLGNN_CHECK_ARG(lddz >= d && lddh >= d, "gat_backward_features: leading dimension < d");
    GatBwdArgs A{};
    A.rowptr = rowptr; A.col = col; A.dz = dz; A.lddz = lddz; A.s_dst = s_dst; A.s_src = s_src; A.mx = mx; A.den = den;
    A.ds_dst = const_cast<float *>(ds_dst); A.ds_src = const_cast<float *>(ds_src); A.att = att;
    A.dh = dh; A.lddh = lddh; A.n = n; A.d = d; A.group_src = group == BLGNN_GAT_GROUP_SRC; A.slope = slope;
    hipLaunchKernelGGL(k_gat_bwd_features, dim3(n), dim3(256), 0, (hipStream_t)stream, A);
    return launch_status("gat_backward_features");
}
