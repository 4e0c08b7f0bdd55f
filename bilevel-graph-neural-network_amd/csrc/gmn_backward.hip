// Backward of the two GMN-specific operators (model/layers_gmn.py:34-84 under src/train.py:177-182).
//
// (1) edge_relu_sum:  S_i = sum_{j in N(i)} relu(P_i + Q_j)   (self loops kept)
//       dP_i = sum_{j in N(i)} [P_i + Q_j > 0] dS_i        dQ_i = sum_{j in N(i)} [P_j + Q_i > 0] dS_j
//     (the second sum uses the symmetry of the molecular graphs: the row of i lists every j whose row lists i).
//     One pass per row, lane-over-feature float4s, no atomics.
//
// (2) cross-graph attention of a pair (X1 [n1, D], X2 [n2, D]):
//       N = rows / max(|row|, eps),  C = N1 N2^T,  E = exp(C),  A = diag(1/rowsum E) E,  B = diag(1/colsum E) E^T,
//       U1 = X1 - A X2,  U2 = X2 - B X1.
//     Given dU1, dU2:
//       dA_ij = -dU1_i . X2_j                                dB_ji = -dU2_j . X1_i
//       dC_ij = A_ij (dA_ij - sum_j' A_ij' dA_ij') + B_ji (dB_ji - sum_i' B_ji' dB_ji')
//       dX1_i = dU1_i - sum_j B_ji dU2_j + Jn_i( sum_j dC_ij N2_j ),   Jn_i(v) = (v - N1_i (N1_i . v)) / |X1_i|
//       dX2_j = dU2_j - sum_i A_ij dU1_i + Jn_j( sum_i dC_ij N1_i ).
//     One workgroup per pair, any size: the n1 x n2 matrices (E, dA, dB, dC) live in a caller-provided scratch area
//     (cell offsets per pair), per-row scalars in a second one; every phase is a strided loop of wavefronts over cells
//     or rows with D-wide dot products / accumulations.  This is the training path: clarity over speed (the forward's
//     matrix-core kernels are where the time goes; a pair costs ~5 n1 n2 D flop here).
#include <cstdlib>
#include "common.hpp"

namespace blgnn {

// ---- (1) -----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_edge_relu_sum_bwd(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                           const float *__restrict__ p, int64_t ldp,
                                                           const float *__restrict__ q, int64_t ldq,
                                                           const float *__restrict__ ds, int64_t ldds, int n, int d,
                                                           float *__restrict__ dp, int64_t lddp, float *__restrict__ dq,
                                                           int64_t lddq) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int i = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (i >= n) return;
    const int e0 = rowptr[i], e1 = rowptr[i + 1];
    for (int c = lane; c < d; c += WAVE) {
        const float pi = p[(int64_t)i * ldp + c], qi = q[(int64_t)i * ldq + c], gi = ds[(int64_t)i * ldds + c];
        float ap = 0.f, aq = 0.f;
        for (int e = e0; e < e1; ++e) {
            const int j = col[e];
            if (pi + q[(int64_t)j * ldq + c] > 0.f) ap += gi;
            if (p[(int64_t)j * ldp + c] + qi > 0.f) aq += ds[(int64_t)j * ldds + c];
        }
        dp[(int64_t)i * lddp + c] = ap;
        dq[(int64_t)i * lddq + c] = aq;
    }
}

// ---- (2) -----------------------------------------------------------------------------------------------------------------
struct MatchBwdArgs {
    const int32_t *seg;                 // 2 P + 1 graph offsets
    const int64_t *cell_off;            // P + 1: first cell of every pair's n1 x n2 matrices
    const float *x; int64_t ldx;
    const float *du; int64_t lddu;
    float *dx; int64_t lddx;
    float *cells;                       // [4][total_cells]: E, dA, dB, dC
    float *rows;                        // [3][total_rows]: 1 / |x|, 1 / (row | column sum of E), softmax-backward row term
    int64_t total_cells, total_rows;
    int32_t num_pairs, d;
};

__device__ __forceinline__ float dot_rows(const float *a, const float *b, int d, int lane) {
    float s = 0.f;
    for (int c = lane; c < d; c += WAVE) s = fmaf(a[c], b[c], s);
    return group_sum(s, WAVE);
}

// a pair with every operand row read from global memory (L2): any graph size, any d
__device__ __forceinline__ void match_bwd_global(const MatchBwdArgs &A, const int pair) {
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    const int s0 = A.seg[2 * pair], s1 = A.seg[2 * pair + 1], s2 = A.seg[2 * pair + 2];
    const int n1 = s1 - s0, n2 = s2 - s1, d = A.d;
    const int64_t c0 = A.cell_off[pair];
    float *E = A.cells + c0, *dA = E + A.total_cells, *dB = dA + A.total_cells, *dC = dB + A.total_cells;
    float *inv = A.rows, *isum = inv + A.total_rows, *rterm = isum + A.total_rows;     // indexed by global row
    auto X = [&](int r) { return A.x + (int64_t)r * A.ldx; };
    auto DU = [&](int r) { return A.du + (int64_t)r * A.lddu; };
    // phase 0: 1 / max(|x_r|, eps) for every row of the pair
    for (int r = s0 + wid; r < s2; r += nw) {
        const float nn = dot_rows(X(r), X(r), d, lane);
        if (lane == 0) inv[r] = 1.f / fmaxf(sqrtf(nn), 1e-12f);
    }
    __threadfence(); __syncthreads();
    // phase 1: E, dA, dB per cell
    for (int cell = wid; cell < n1 * n2; cell += nw) {
        const int i = cell / n2, j = cell - i * n2;
        const float cs = dot_rows(X(s0 + i), X(s1 + j), d, lane) * inv[s0 + i] * inv[s1 + j];
        const float a = -dot_rows(DU(s0 + i), X(s1 + j), d, lane);
        const float b = -dot_rows(DU(s1 + j), X(s0 + i), d, lane);
        if (lane == 0) { E[cell] = expf(cs); dA[cell] = a; dB[cell] = b; }
    }
    __threadfence(); __syncthreads();
    // phase 2: per row of graph 1: 1 / rowsum, sum_j A_ij dA_ij; per row of graph 2: 1 / colsum, sum_i B_ji dB_ji
    for (int t = threadIdx.x; t < n1 + n2; t += blockDim.x) {
        float sum = 0.f, acc = 0.f;
        if (t < n1) {
            for (int j = 0; j < n2; ++j) { const float e = E[t * n2 + j]; sum += e; acc = fmaf(e, dA[t * n2 + j], acc); }
        } else {
            const int j = t - n1;
            for (int i = 0; i < n1; ++i) { const float e = E[i * n2 + j]; sum += e; acc = fmaf(e, dB[i * n2 + j], acc); }
        }
        const float is = sum > 0.f ? 1.f / sum : 0.f;
        isum[s0 + t] = is;
        rterm[s0 + t] = acc * is;
    }
    __threadfence(); __syncthreads();
    // phase 3: dC
    for (int cell = threadIdx.x; cell < n1 * n2; cell += blockDim.x) {
        const int i = cell / n2, j = cell - i * n2;
        const float e = E[cell];
        dC[cell] = e * isum[s0 + i] * (dA[cell] - rterm[s0 + i]) + e * isum[s1 + j] * (dB[cell] - rterm[s1 + j]);
    }
    __threadfence(); __syncthreads();
    // phase 4: one wavefront per row: dX = dU - (partner-weighted dU) + normalisation-backward(sum dC N_partner)
    for (int t = wid; t < n1 + n2; t += nw) {
        const bool first = t < n1;
        const int r = s0 + t;                                  // this row
        const int pb = first ? s1 : s0, np = first ? n2 : n1;  // partner graph
        const int li = first ? t : t - n1;
        const float ir = inv[r];
        // v = sum_k dC N_partner_k ;  w = sum_k (B or A weight) dU_partner_k
        float dotnv = 0.f;
        for (int cbase = 0; cbase < d; cbase += WAVE) {
            const int c = cbase + lane;
            float v = 0.f;
            if (c < d)
                for (int k = 0; k < np; ++k) {
                    const int cell = first ? li * n2 + k : k * n2 + li;
                    v = fmaf(dC[cell] * inv[pb + k], X(pb + k)[c], v);
                }
            dotnv += (c < d) ? v * X(r)[c] * ir : 0.f;
        }
        dotnv = group_sum(dotnv, WAVE);                        // N_r . v
        for (int c = lane; c < d; c += WAVE) {
            float v = 0.f, w = 0.f;
            for (int k = 0; k < np; ++k) {
                const int cell = first ? li * n2 + k : k * n2 + li;
                v = fmaf(dC[cell] * inv[pb + k], X(pb + k)[c], v);
                // row of graph 1 receives -sum_j B_ji dU2_j (B_ji = E_ij / colsum_j); row of graph 2: -sum_i A_ij dU1_i
                w = fmaf(E[cell] * isum[pb + k], DU(pb + k)[c], w);
            }
            const float nr = X(r)[c] * ir;
            A.dx[(int64_t)r * A.lddx + c] = DU(r)[c] - w + (v - nr * dotnv) * ir;
        }
    }
}


__global__ void __launch_bounds__(256) k_gmn_match_bwd(const MatchBwdArgs A) {
    if ((int)blockIdx.x < A.num_pairs) match_bwd_global(A, blockIdx.x);
}

// The same backward with ONE GRAPH OF THE PAIR STAGED IN LDS at a time (x and dU rows, stride d + 4 floats: a lane per
// row reads 16 bytes conflict-free), 512 threads per pair.  The first version above gives every n1 x n2 cell three
// D-wide dot products of a whole wavefront each (loads -> six shuffle steps, one after the other) and every row two
// passes over its partner's rows out of L2: 5.6 ms per pair and workgroup at the configs[2] shape (8 192 pairs of ~29 +
// 29 atoms, d = 256: 22.6 ms per call, 27 % of the GMN training step).  Here
//   phase 1: a wavefront takes rows i of graph 1, its LANES the rows j of (staged) graph 2: the three dot products of
//            cell (i, j) run down the lane with x1_i / dU1_i as wave-uniform operands -- no cross-lane reduction at all;
//   phase 4: a wavefront takes a row, its lanes 4 columns each: v and w accumulate over the staged partner rows with
//            wave-uniform coefficients, v stays in registers for the normalisation term (one pass, one reduction);
//            graph 1's rows first (partner = graph 2, staged), then graph 1 is staged and graph 2's rows follow.
// A graph with more rows than the budget (64 at d = 256) is staged in blocks.
template <int CH>
__global__ void __launch_bounds__(512) k_gmn_match_bwd_lds(const MatchBwdArgs A, const int lds_rows) {
    extern __shared__ float4 mb_smem4[];
    const int pair = blockIdx.x;
    const int s0 = A.seg[2 * pair], s1 = A.seg[2 * pair + 1], s2 = A.seg[2 * pair + 2];
    const int n1 = s1 - s0, n2 = s2 - s1, d = A.d, d4 = d >> 2;
    const int lane = threadIdx.x & (WAVE - 1), wid = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE), nw = blockDim.x / WAVE;
    const int ldl = d + 4;                                                           // floats per staged row
    float *xs = reinterpret_cast<float *>(mb_smem4), *gs = xs + (int64_t)lds_rows * ldl;
    float *wx = gs + (int64_t)lds_rows * ldl + (int64_t)wid * 2 * ldl, *wg = wx + ldl;    // a private row pair per wavefront
    const int64_t c0 = A.cell_off[pair];
    float *E = A.cells + c0, *dA = E + A.total_cells, *dB = dA + A.total_cells, *dC = dB + A.total_cells;
    float *inv = A.rows, *isum = inv + A.total_rows, *rterm = isum + A.total_rows;
    auto X4 = [&](int r) { return reinterpret_cast<const float4 *>(A.x + (int64_t)r * A.ldx); };
    auto G4 = [&](int r) { return reinterpret_cast<const float4 *>(A.du + (int64_t)r * A.lddu); };
    auto stage = [&](int g0, int ng) {                    // rows g0 .. g0 + ng of x and dU -> LDS (all threads; barriers by the caller)
        for (int idx = threadIdx.x; idx < ng * d4; idx += blockDim.x) {
            const int r = idx / d4, c4 = idx - r * d4;
            *reinterpret_cast<float4 *>(xs + r * ldl + 4 * c4) = X4(g0 + r)[c4];
            *reinterpret_cast<float4 *>(gs + r * ldl + 4 * c4) = G4(g0 + r)[c4];
        }
    };
    // phase 0: 1 / max(|x_r|, eps) for every row of the pair
    for (int r = s0 + wid; r < s2; r += nw) {
        float nn = 0.f;
        for (int c4 = lane; c4 < d4; c4 += WAVE) {
            const float4 v = X4(r)[c4];
            nn = fmaf(v.x, v.x, nn); nn = fmaf(v.y, v.y, nn); nn = fmaf(v.z, v.z, nn); nn = fmaf(v.w, v.w, nn);
        }
        nn = group_sum(nn, WAVE);
        if (lane == 0) inv[r] = 1.f / fmaxf(sqrtf(nn), 1e-12f);
    }
    __threadfence(); __syncthreads();
    // phase 1: E, dA, dB: graph 2 staged in blocks of lds_rows rows; wavefront = row i of graph 1, lane = row j of the block.
    // The wavefront's own row of x and dU goes through its private LDS line first: every lane needs all of it, and a
    // wave-uniform global load per column group inside the loop is a dependent round trip each
    for (int jb = 0; jb < n2; jb += lds_rows) {
        const int nb = min(lds_rows, n2 - jb);
        if (jb > 0) __syncthreads();
        stage(s1 + jb, nb);
        __syncthreads();
        for (int i = wid; i < n1; i += nw) {
            for (int c4 = lane; c4 < d4; c4 += WAVE) {
                *reinterpret_cast<float4 *>(wx + 4 * c4) = X4(s0 + i)[c4];
                *reinterpret_cast<float4 *>(wg + 4 * c4) = G4(s0 + i)[c4];
            }
            const float ii = inv[s0 + i];
            for (int j0 = 0; j0 < nb; j0 += WAVE) {
                const int jl = j0 + lane;
                const bool live = jl < nb;
                const float *x2 = xs + (live ? jl : 0) * ldl, *g2 = gs + (live ? jl : 0) * ldl;
                float cs = 0.f, a = 0.f, b = 0.f;
#pragma unroll 4
                for (int c4 = 0; c4 < d4; ++c4) {
                    const float4 p = *reinterpret_cast<const float4 *>(wx + 4 * c4), q = *reinterpret_cast<const float4 *>(wg + 4 * c4);
                    const float4 u = *reinterpret_cast<const float4 *>(x2 + 4 * c4), w = *reinterpret_cast<const float4 *>(g2 + 4 * c4);
                    cs = fmaf(p.x, u.x, cs); cs = fmaf(p.y, u.y, cs); cs = fmaf(p.z, u.z, cs); cs = fmaf(p.w, u.w, cs);
                    a = fmaf(q.x, u.x, a); a = fmaf(q.y, u.y, a); a = fmaf(q.z, u.z, a); a = fmaf(q.w, u.w, a);
                    b = fmaf(w.x, p.x, b); b = fmaf(w.y, p.y, b); b = fmaf(w.z, p.z, b); b = fmaf(w.w, p.w, b);
                }
                if (live) {
                    const int j = jb + jl, cell = i * n2 + j;
                    E[cell] = expf(cs * ii * inv[s1 + j]); dA[cell] = -a; dB[cell] = -b;
                }
            }
        }
    }
    __threadfence(); __syncthreads();
    // phase 2: per row of graph 1: 1 / rowsum, sum_j A_ij dA_ij; per row of graph 2: 1 / colsum, sum_i B_ji dB_ji
    for (int t = threadIdx.x; t < n1 + n2; t += blockDim.x) {
        float sum = 0.f, acc = 0.f;
        if (t < n1) {
            for (int j = 0; j < n2; ++j) { const float e = E[t * n2 + j]; sum += e; acc = fmaf(e, dA[t * n2 + j], acc); }
        } else {
            const int j = t - n1;
            for (int i = 0; i < n1; ++i) { const float e = E[i * n2 + j]; sum += e; acc = fmaf(e, dB[i * n2 + j], acc); }
        }
        const float is = sum > 0.f ? 1.f / sum : 0.f;
        isum[s0 + t] = is;
        rterm[s0 + t] = acc * is;
    }
    __threadfence(); __syncthreads();
    // phase 3: dC
    for (int cell = threadIdx.x; cell < n1 * n2; cell += blockDim.x) {
        const int i = cell / n2, j = cell - i * n2;
        const float e = E[cell];
        dC[cell] = e * isum[s0 + i] * (dA[cell] - rterm[s0 + i]) + e * isum[s1 + j] * (dB[cell] - rterm[s1 + j]);
    }
    __threadfence(); __syncthreads();
    // phase 4: dX = dU - (partner-weighted dU) + normalisation-backward(sum dC N_partner): a wavefront per own row, its
    // lanes 4 columns each, the partner graph staged (in blocks when it has more than lds_rows rows: then once per group
    // of own rows); the partner rows' coefficients: lane k fetches its own pair once per block, the loop broadcasts them
#pragma unroll 1
    for (int side = 0; side < 2; ++side) {
        const bool first = side == 0;
        const int own0 = first ? s0 : s1, nown = first ? n1 : n2, pb = first ? s1 : s0, np = first ? n2 : n1;
        const bool single = np <= lds_rows;
        if (single && !(first && n2 <= lds_rows)) {       // (graph 2 is still staged after a single-block phase 1)
            __syncthreads();
            stage(pb, np);
            __syncthreads();
        }
        for (int li0 = 0; li0 < nown; li0 += nw) {
            const int li = li0 + wid;
            const bool active = li < nown;
            const int r = own0 + (active ? li : 0);
            float4 v[CH], w[CH];
#pragma unroll
            for (int h = 0; h < CH; ++h) { v[h] = f4_zero(); w[h] = f4_zero(); }
            for (int kb = 0; kb < np; kb += lds_rows) {
                const int nk = min(lds_rows, np - kb);
                if (!single) {
                    __syncthreads();
                    stage(pb + kb, nk);
                    __syncthreads();
                }
                if (active)
                    for (int k0 = 0; k0 < nk; k0 += WAVE) {
                        float cvk = 0.f, cwk = 0.f;
                        const int kk = kb + k0 + lane;
                        if (k0 + lane < nk) {
                            const int cell = first ? li * n2 + kk : kk * n2 + li;
                            cvk = dC[cell] * inv[pb + kk];
                            cwk = E[cell] * isum[pb + kk];
                        }
                        const int kn = min(WAVE, nk - k0);
                        for (int k = 0; k < kn; ++k) {
                            const float cv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cvk), k));
                            const float cw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cwk), k));
#pragma unroll
                            for (int h = 0; h < CH; ++h) {
                                const int c4 = lane + WAVE * h;
                                if (c4 < d4) {
                                    f4_fma(v[h], cv, *reinterpret_cast<const float4 *>(xs + (k0 + k) * ldl + 4 * c4));
                                    f4_fma(w[h], cw, *reinterpret_cast<const float4 *>(gs + (k0 + k) * ldl + 4 * c4));
                                }
                            }
                        }
                    }
            }
            if (active) {
                const float ir = inv[r];
                float4 xr[CH];
                float dotnv = 0.f;
#pragma unroll
                for (int h = 0; h < CH; ++h) {
                    const int c4 = lane + WAVE * h;
                    xr[h] = c4 < d4 ? X4(r)[c4] : f4_zero();
                    dotnv = fmaf(v[h].x, xr[h].x, dotnv); dotnv = fmaf(v[h].y, xr[h].y, dotnv);
                    dotnv = fmaf(v[h].z, xr[h].z, dotnv); dotnv = fmaf(v[h].w, xr[h].w, dotnv);
                }
                dotnv = group_sum(dotnv, WAVE) * ir;                                  // N_r . v
#pragma unroll
                for (int h = 0; h < CH; ++h) {
                    const int c4 = lane + WAVE * h;
                    if (c4 < d4) {
                        const float4 g = G4(r)[c4];
                        float4 o;
                        o.x = g.x - w[h].x + (v[h].x - xr[h].x * ir * dotnv) * ir;
                        o.y = g.y - w[h].y + (v[h].y - xr[h].y * ir * dotnv) * ir;
                        o.z = g.z - w[h].z + (v[h].z - xr[h].z * ir * dotnv) * ir;
                        o.w = g.w - w[h].w + (v[h].w - xr[h].w * ir * dotnv) * ir;
                        *reinterpret_cast<float4 *>(A.dx + (int64_t)r * A.lddx + 4 * c4) = o;
                    }
                }
            }
        }
    }
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_edge_relu_sum_backward(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *p,
                                            int64_t ldp, const float *q, int64_t ldq, const float *ds, int64_t ldds,
                                            int32_t n, int32_t d, float *dp, int64_t lddp, float *dq, int64_t lddq,
                                            void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && d >= 0 && nnz >= 0, "edge_relu_sum_backward: negative size");
    if (n == 0 || d == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(rowptr && (col || nnz == 0) && p && q && ds && dp && dq, "edge_relu_sum_backward: null pointer");
    BLGNN_CHECK_ARG(ldp >= d && ldq >= d && ldds >= d && lddp >= d && lddq >= d, "edge_relu_sum_backward: leading dimension < d");
    hipLaunchKernelGGL(k_edge_relu_sum_bwd, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, rowptr, col, p, ldp, q, ldq,
                       ds, ldds, n, d, dp, lddp, dq, lddq);
    return launch_status("edge_relu_sum_backward");
}

extern "C" int blgnn_gmn_match_backward(const int32_t *seg, int32_t num_pairs, const float *x, int64_t ldx,
                                        const float *du, int64_t lddu, const int64_t *cell_off, int64_t total_cells,
                                        float *cell_scratch, float *row_scratch, float *dx, int64_t lddx, int32_t d,
                                        int64_t total_rows, void *stream) {
    BLGNN_CHECK_ARG(num_pairs >= 0 && d >= 0 && total_cells >= 0 && total_rows >= 0, "gmn_match_backward: negative size");
    if (num_pairs == 0 || d == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(seg && x && du && cell_off && cell_scratch && row_scratch && dx, "gmn_match_backward: null pointer");
    BLGNN_CHECK_ARG(ldx >= d && lddu >= d && lddx >= d, "gmn_match_backward: leading dimension < d");
    BLGNN_CHECK_ARG(dx != x && dx != du, "gmn_match_backward: dx must not alias the inputs");
    MatchBwdArgs A{};
    A.seg = seg; A.cell_off = cell_off; A.x = x; A.ldx = ldx; A.du = du; A.lddu = lddu; A.dx = dx; A.lddx = lddx;
    A.cells = cell_scratch; A.rows = row_scratch; A.total_cells = total_cells; A.total_rows = total_rows;
    A.num_pairs = num_pairs; A.d = d;
    // LDS-staged kernel: 16-byte aligned rows of d <= 1024 columns; one graph's x and dU rows must fit 150 KB
    static const bool lds_ok = [] { const char *e = getenv("BLGNN_MATCH_BWD_LDS"); return !(e && e[0] == '0'); }();
    const int lds_rows = d > 0 ? (150 * 1024 - 16 * (d + 4) * 4) / (2 * (d + 4) * 4) : 0;   // + a row pair per wavefront
    if (lds_ok && d % 4 == 0 && d <= 1024 && ldx % 4 == 0 && lddu % 4 == 0 && lddx % 4 == 0 && aligned16(x) && aligned16(du) &&
        aligned16(dx) && lds_rows >= 16) {
        const int rows = lds_rows > 64 ? 64 : lds_rows;
        const int bytes = (2 * rows + 16) * (d + 4) * 4;
        const int ch = (d / 4 + WAVE - 1) / WAVE;
#define BLGNN_MB_LAUNCH(CH_)                                                                                          \
    do {                                                                                                              \
        static int declared = 0;                                                                                      \
        if (declared < bytes) {                                                                                       \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gmn_match_bwd_lds<CH_>),                          \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)                 \
                return fail(BLGNN_ERR_LAUNCH, "gmn_match_backward: cannot raise the dynamic LDS limit");              \
            declared = bytes;                                                                                         \
        }                                                                                                             \
        hipLaunchKernelGGL(k_gmn_match_bwd_lds<CH_>, dim3(num_pairs), dim3(512), bytes, (hipStream_t)stream, A, rows); \
    } while (0)
        if (ch <= 1) BLGNN_MB_LAUNCH(1);
        else if (ch == 2) BLGNN_MB_LAUNCH(2);
        else BLGNN_MB_LAUNCH(4);
#undef BLGNN_MB_LAUNCH
        return launch_status("gmn_match_backward");
    }
    hipLaunchKernelGGL(k_gmn_match_bwd, dim3(num_pairs), dim3(256), 0, (hipStream_t)stream, A);
    return launch_status("gmn_match_backward");
}
