// Backward-pass and train-mode kernels (what src/train.py:177-182 --
// loss.backward(); optimizer.step() -- needs from this path for the lower-level GIN model).
//
// The big products of the backward pass reuse the forward kernels: dX = dY @ W is blgnn_linear on
// nn.Linear's own [out, in] weight, dW = dY^T @ X is blgnn_linear on a transposed copy of dY (split-K
// over the rows, two-level accumulation), the GIN aggregation's gradient is blgnn_spmm_sum on the same
// CSR (the molecular graphs are symmetric).  What is new here are the reductions and the pointwise
// pieces around them:
//   * column reductions over all rows (bias gradients, train-mode BatchNorm statistics and gradients):
//     per 512-row chunk partial sums + a fixed-order final pass -- deterministic, no atomics;
//   * BatchNorm1d train forward / backward (model/layers.py:39-40,56-57), activation backward;
//   * tiled transpose;
//   * segment-mean / segment-sum backward (the readout's gradient: model/layers_aggregation.py:16-19);
//   * L2-normalise backward and the pair gather's scatter-add (model/layers_link_pred.py:44,56-61).
// All are HBM-bound: rows are read with 16-byte accesses where alignment allows.
#include "common.hpp"

namespace blgnn {

constexpr int CR_ROWS = 512;            // most rows per partial-reduction chunk
// rows per chunk: enough chunks to fill the chip on the matrices of this path (37 k x 128 -> 290 chunks of 128 rows;
// with 512 rows the 146 workgroups each ran 128 dependent strided loads: 29 us for 19 MB)
static inline int cr_rows(int M) {
    int r = ((M + 511) / 512 + 31) / 32 * 32;
    return r < 64 ? 64 : (r > CR_ROWS ? CR_ROWS : r);
}
enum { COL_SUM = 0, COL_DOT = 1, COL_SQDEV = 2, COL_DOT_NORM = 3, COL_DOT_NEG = 4 };

// partial[chunk][c] = sum over the chunk's rows of  a  |  a * b  |  (a - mean[c])^2  |  a * (b - mean[c]) * rstd[c]  |
// a * min(b, 0)  (the slope gradient of a per-channel PReLU: a = dy, b = the pre-activation)
template <int MODE>
__global__ void __launch_bounds__(256) k_col_partial(const float *__restrict__ a, int64_t lda,
                                                     const float *__restrict__ b, int64_t ldb,
                                                     const float *__restrict__ mean,
                                                     const float *__restrict__ rstd, int M, int N, int rows,
                                                     float *__restrict__ partial) {
    // 64 columns x 4 row lanes per workgroup pass; grid = (column strips of 64, row chunks of ``rows``)
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;                                  // 0..3
    const int r0 = blockIdx.y * rows;
    const int r1 = min(r0 + rows, M);
    float s = 0.f;
    if (c < N) {
        const float mu = (MODE == COL_SQDEV || MODE == COL_DOT_NORM) ? mean[c] : 0.f;
        const float rs = MODE == COL_DOT_NORM ? rstd[c] : 1.f;
        for (int r = r0 + rl; r < r1; r += 4) {
            const float v = a[(int64_t)r * lda + c];
            if (MODE == COL_SUM) s += v;
            else if (MODE == COL_DOT) s = fmaf(v, b[(int64_t)r * ldb + c], s);
            else if (MODE == COL_DOT_NORM) s = fmaf(v, (b[(int64_t)r * ldb + c] - mu) * rs, s);
            else if (MODE == COL_DOT_NEG) s = fmaf(v, fminf(b[(int64_t)r * ldb + c], 0.f), s);
            else { const float dlt = v - mu; s = fmaf(dlt, dlt, s); }
        }
    }
    __shared__ float red[4][64];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N)
        partial[(int64_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) +
                                               (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[c] = scale * sum over the chunks' partials, in a FIXED order: four row lanes take chunks k = lane, lane + 4, ...
// (four loads in flight each), their sums are combined as (s0 + s1) + (s2 + s3).  64 columns per workgroup.
__global__ void __launch_bounds__(256) k_col_final(const float *__restrict__ partial, int chunks, int N, float scale,
                                                   float *__restrict__ out) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N) {
        int k = rl;
        for (; k + 12 < chunks; k += 16) {
            const float v0 = partial[(int64_t)k * N + c], v1 = partial[(int64_t)(k + 4) * N + c];
            const float v2 = partial[(int64_t)(k + 8) * N + c], v3 = partial[(int64_t)(k + 12) * N + c];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < chunks; k += 4) s += partial[(int64_t)k * N + c];
    }
    __shared__ float red[4][64];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < N)
        out[c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x])) * scale;
}

// the [N]-vector step between the two BatchNorm passes, one launch instead of eight pointwise ones:
// rstd = 1 / sqrt(var + eps); running_mean <- (1 - m) running_mean + m mean; running_var <- (1 - m) running_var + m var c
// (c = M / (M - 1): torch's unbiased running variance; torch.nn.BatchNorm1d, train mode)
__global__ void k_bn_finalize(const float *__restrict__ mean, const float *__restrict__ var, int N, float eps, float mom,
                              float unbias, float *__restrict__ rstd, float *__restrict__ run_mean,
                              float *__restrict__ run_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const float v = var[c];
    rstd[c] = 1.f / sqrtf(v + eps);
    if (run_mean) run_mean[c] = fmaf(mom, mean[c], (1.f - mom) * run_mean[c]);
    if (run_var) run_var[c] = fmaf(mom, v * unbias, (1.f - mom) * run_var[c]);
}

// y = (x - mean) * rstd * gamma + beta
__global__ void k_bn_apply(const float *__restrict__ x, int64_t ldx, const float *__restrict__ mean,
                           const float *__restrict__ rstd, const float *__restrict__ gamma,
                           const float *__restrict__ beta, int M, int N, int act, float *__restrict__ y, int64_t ldy) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int r = (int)(idx / N), c = (int)(idx % N);
    y[(int64_t)r * ldy + c] = apply_act(fmaf((x[(int64_t)r * ldx + c] - mean[c]) * rstd[c], gamma ? gamma[c] : 1.f,
                                             beta ? beta[c] : 0.f), act, 0.f);
}

// dx = gamma * rstd * (dy - dbeta / M - xhat * dgamma / M),  xhat = (x - mean) * rstd
__global__ void k_bn_backward(const float *__restrict__ dy, int64_t lddy, const float *__restrict__ x, int64_t ldx,
                              const float *__restrict__ mean, const float *__restrict__ rstd,
                              const float *__restrict__ gamma, const float *__restrict__ dgamma,
                              const float *__restrict__ dbeta, int M, int N, float inv_m, float *__restrict__ dx,
                              int64_t lddx) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int r = (int)(idx / N), c = (int)(idx % N);
    const float xh = (x[(int64_t)r * ldx + c] - mean[c]) * rstd[c];
    dx[(int64_t)r * lddx + c] = (gamma ? gamma[c] : 1.f) * rstd[c] *
                                (dy[(int64_t)r * lddy + c] - dbeta[c] * inv_m - xh * dgamma[c] * inv_m);
}

// dx = dy * act'(.) from the activation's OUTPUT y: relu y > 0, sigmoid y (1 - y), tanh 1 - y^2, identity 1.
// PReLU (v > 0 ? 1 : slope[c]) tests the sign of what the caller passes: the PRE-activation for trained slopes (any sign
// of the slope; blgnn/autograd.py keeps it), the output only where slope > 0 is known.  The slope gradient is
// blgnn_col_reduce's a * min(b, 0) mode.
__global__ void k_act_backward(const float *__restrict__ dy, int64_t lddy, const float *__restrict__ y, int64_t ldy,
                               int act, const float *__restrict__ slope, int M, int N, float *__restrict__ dx,
                               int64_t lddx) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int r = (int)(idx / N), c = (int)(idx % N);
    const float g = dy[(int64_t)r * lddy + c], v = y[(int64_t)r * ldy + c];
    float d = 1.f;
    if (act == BLGNN_ACT_RELU) d = v > 0.f ? 1.f : 0.f;
    else if (act == BLGNN_ACT_PRELU) d = v > 0.f ? 1.f : (slope ? slope[c] : 0.f);
    else if (act == BLGNN_ACT_SIGMOID) d = v * (1.f - v);
    else if (act == BLGNN_ACT_TANH) d = 1.f - v * v;
    dx[(int64_t)r * lddx + c] = g * d;
}

// y = act(x) with a per-channel PReLU slope: the activation as its own step where the backward needs the pre-activation
__global__ void k_act_forward(const float *__restrict__ x, int64_t ldx, int act, const float *__restrict__ slope, int M,
                              int N, float *__restrict__ y, int64_t ldy) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int r = (int)(idx / N), c = (int)(idx % N);
    y[(int64_t)r * ldy + c] = apply_act(x[(int64_t)r * ldx + c], act, slope ? slope[c] : 0.f);
}

// y[N, M] = x[M, N]^T through a padded 32 x 32 LDS tile
__global__ void __launch_bounds__(256) k_transpose(const float *__restrict__ x, int64_t ldx, int M, int N,
                                                   float *__restrict__ y, int64_t ldy) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = m0 + ty + 8 * k, n = n0 + tx;
        tile[ty + 8 * k][tx] = (m < M && n < N) ? x[(int64_t)m * ldx + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = n0 + ty + 8 * k, m = m0 + tx;
        if (n < N && m < M) y[(int64_t)n * ldy + m] = tile[tx][ty + 8 * k];
    }
}

// readout backward: dx[i, :] (+)= dp[row_of(g(i)), :] * (mean ? 1 / max(count_g, 1) : 1)
__global__ void k_segment_broadcast(const int32_t *__restrict__ seg, int G, const float *__restrict__ dp, int64_t lddp,
                                    const int32_t *__restrict__ row_map, int d, int mean, float *__restrict__ dx,
                                    int64_t lddx, int accumulate) {
    const int g = blockIdx.x;
    if (g >= G) return;
    const int r0 = seg[g], r1 = seg[g + 1];
    const float w = mean ? 1.f / (float)max(r1 - r0, 1) : 1.f;
    const float *src = dp + (int64_t)(row_map ? row_map[g] : g) * lddp;
    for (int idx = threadIdx.x; idx < (r1 - r0) * d; idx += blockDim.x) {
        const int r = r0 + idx / d, c = idx % d;
        float *dst = dx + (int64_t)r * lddx + c;
        const float v = src[c] * w;
        *dst = accumulate ? *dst + v : v;
    }
}

// zn = z / max(|z|, eps) row-wise  ->  dz = (dzn - zn (zn . dzn)) / max(|z|, eps)   (rows with |z| < eps: dzn / eps)
__global__ void __launch_bounds__(256) k_l2_normalize_backward(const float *__restrict__ z, int64_t ldz,
                                                               const float *__restrict__ dzn, int64_t lddzn, int M, int d,
                                                               float eps, float *__restrict__ dz, int64_t lddz) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *zr = z + (int64_t)row * ldz, *gr = dzn + (int64_t)row * lddzn;
    float nn = 0.f, dot = 0.f;
    for (int c = lane; c < d; c += 64) { nn = fmaf(zr[c], zr[c], nn); dot = fmaf(zr[c], gr[c], dot); }
    nn = group_sum(nn, WAVE); dot = group_sum(dot, WAVE);
    const float norm = sqrtf(nn);
    float *out = dz + (int64_t)row * lddz;
    if (norm < eps) {
        for (int c = lane; c < d; c += 64) out[c] = gr[c] / eps;
    } else {
        const float inv = 1.f / norm, k = dot * inv * inv * inv;     // zn (zn . g) / |z| = z (z . g) / |z|^3
        for (int c = lane; c < d; c += 64) out[c] = fmaf(-k, zr[c], gr[c] * inv);
    }
}

// dz[ids[b], :] += dcat[b, col0 : col0 + d]  (several pairs may name the same graph: float atomics at the memory side)
__global__ void k_pair_scatter_add(const float *__restrict__ dcat, int64_t ldc, int col0, const int64_t *__restrict__ ids,
                                   int B, int d, float *__restrict__ dz, int64_t lddz) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * d) return;
    const int b = (int)(idx / d), c = (int)(idx % d);
    unsafeAtomicAdd(dz + ids[b] * lddz + c, dcat[(int64_t)b * ldc + col0 + c]);
}

// dW[o, i] = sum_m g[m, o] x[m, i]  (the weight gradient of y = x W^T, nn.Linear layout [out, in]; also x^T dh for GATConv's
// [in, out] weight with the operands swapped).  The reduction runs over ALL rows (tens of thousands of atoms) while the
// output is tiny, so the rows are cut into chunks of GW_ROWS: one workgroup per (chunk, 128 x 128 output block) forms its
// partial product with exact-f32 MFMA straight from the row-major operands (lane fi reads g[row, 32 to + fi] as the A
// fragment and x[row, 32 ti + fi] as the B fragment: both coalesced, no transpose, no LDS), partials go to scratch and
// k_col_final adds them in a fixed order (deterministic).  The forward GEMM kernels are shaped for the opposite case.
constexpr int GW_ROWS = 1024;           // most rows per chunk
// rows per chunk: ONE workgroup per CU, all of them equally long (37 k x 128 x 128: 256 chunks of 144 rows.  37 chunks of
// 1 024 rows ran 64 dependent operand round trips each: 306 us per launch; 290 chunks of 128 rows: 46 us, because the 34
// workgroups beyond the 256th made a second round as long as the first)
static inline int gw_rows(int M, int O, int I) {
    const int tiles = ((O + 127) / 128) * ((I + 127) / 128);
    const int want = tiles >= 256 ? 1 : 256 / tiles;                   // chunks wanted
    int r = (M + want - 1) / want;
    return r < 64 ? 64 : (r > GW_ROWS ? GW_ROWS : r);
}
typedef float f32x16g __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_grad_weight(const float *__restrict__ g, int64_t ldg, const float *__restrict__ x,
                                                     int64_t ldx, int M, int O, int I, int rows,
                                                     float *__restrict__ partial) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int bi_n = (I + 127) / 128;
    const int o0 = (blockIdx.y / bi_n) * 128 + (wid >> 1) * 64, i0 = (blockIdx.y % bi_n) * 128 + (wid & 1) * 64;
    const int r0 = blockIdx.x * rows, r1 = min(r0 + rows, M);
    f32x16g acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // columns outside the matrices are clamped (their products land in rows / columns that are never written)
    const int oc[2] = {min(o0 + fi, O - 1), min(o0 + 32 + fi, O - 1)};
    const int ic[2] = {min(i0 + fi, I - 1), min(i0 + 32 + fi, I - 1)};
    if (o0 < O && i0 < I) {
        // eight K steps (16 rows) of operands are requested before the first MFMA of the group: the loop is a chain of
        // memory round trips otherwise (243 us per launch on 37 k rows; the products themselves need ~30 us; sixteen steps in
        // flight measured the same: what the loop pays for is its 4-byte operand loads, see k_grad_weight_lds)
        constexpr int GU = 8;                               // MFMA K steps (2 rows each) whose operands are in flight together
        for (int k = r0; k < r1; k += 2 * GU) {
            float a0[GU], a1[GU], b0[GU], b1[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int row = min(k + 2 * u + fh, r1 - 1);
                const float live = (k + 2 * u + fh) < r1 ? 1.f : 0.f;
                const float *gr = g + (int64_t)row * ldg, *xr = x + (int64_t)row * ldx;
                a0[u] = gr[oc[0]] * live; a1[u] = gr[oc[1]] * live; b0[u] = xr[ic[0]]; b1[u] = xr[ic[1]];
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc[1][1], 0, 0, 0);
            }
        }
    }
    float *dst = partial + (int64_t)blockIdx.x * O * I;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * fh, i = i0 + 32 * b + fi;
                if (o < O && i < I) dst[(int64_t)o * I + i] = acc[a][b][r];
            }
}

// The same product with the operands staged through LDS: a 32-row block of g and of x (the workgroup's 128 columns of
// each) arrives as coalesced 16-byte loads, the next block sits in registers while the matrix cores work on the current
// one, and the MFMA fragments are 4-byte LDS reads.  (k_grad_weight fetches every fragment element with its own 4-byte
// global load, twice over -- two of the four wavefronts share each operand half: 42-46 us per launch on 37 k x 128 x 128
// where the products need ~10.)  Needs 16-byte aligned rows and O, I multiples of 4; summation order per output element is
// that of k_grad_weight (rows in order inside a chunk, chunks added by k_col_final).
__global__ void __launch_bounds__(256) k_grad_weight_lds(const float *__restrict__ g, int64_t ldg, const float *__restrict__ x,
                                                         int64_t ldx, int M, int O, int I, int rows,
                                                         float *__restrict__ partial) {
    __shared__ float sg[32][128], sx[32][128];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int bi_n = (I + 127) / 128;
    const int ob = (blockIdx.y / bi_n) * 128, ib = (blockIdx.y % bi_n) * 128;
    const int ow = (wid >> 1) * 64, iw = (wid & 1) * 64;              // this wavefront's 64 x 64 corner of the 128 x 128 block
    const int r0 = blockIdx.x * rows, r1 = min(r0 + rows, M);
    f32x16g acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // staging plan: thread t moves float4 number t and t + 256, ... of a 32 x 128 block: row = q / 32, columns 4 (q % 32)
    float4 pg[4], px[4];
    auto fetch = [&](int k) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = tid + 256 * p, row = k + (q >> 5), c = (q & 31) * 4;
            const bool live = row < r1;
            pg[p] = (live && ob + c < O) ? *reinterpret_cast<const float4 *>(g + (int64_t)row * ldg + ob + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            px[p] = (live && ib + c < I) ? *reinterpret_cast<const float4 *>(x + (int64_t)row * ldx + ib + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(r0);
    for (int k = r0; k < r1; k += 32) {
        __syncthreads();                                    // the previous block's fragment reads are done
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int q = tid + 256 * p;
            *reinterpret_cast<float4 *>(&sg[q >> 5][(q & 31) * 4]) = pg[p];
            *reinterpret_cast<float4 *>(&sx[q >> 5][(q & 31) * 4]) = px[p];
        }
        __syncthreads();
        if (k + 32 < r1) fetch(k + 32);                     // in flight behind the 64 MFMAs below
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float a0 = sg[2 * u + fh][ow + fi], a1 = sg[2 * u + fh][ow + 32 + fi];
            const float b0 = sx[2 * u + fh][iw + fi], b1 = sx[2 * u + fh][iw + 32 + fi];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    float *dst = partial + (int64_t)blockIdx.x * O * I;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ob + ow + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * fh, i = ib + iw + 32 * b + fi;
                if (o < O && i < I) dst[(int64_t)o * I + i] = acc[a][b][r];
            }
}

static unsigned grid_for(int64_t items, int threads) { return (unsigned)((items + threads - 1) / threads); }

}  // namespace blgnn

using namespace blgnn;

extern "C" int64_t blgnn_col_reduce_scratch_elems(int32_t M, int32_t N) {
    const int rows = cr_rows(M);
    return (int64_t)((M + rows - 1) / rows) * N;
}

extern "C" int blgnn_col_reduce(const float *a, int64_t lda, const float *b, int64_t ldb, const float *mean,
                                const float *rstd, int32_t M, int32_t N, int32_t mode, float scale, float *scratch,
                                float *out, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N > 0 && mode >= COL_SUM && mode <= COL_DOT_NEG, "col_reduce: bad sizes / mode");
    BLGNN_CHECK_ARG(a && out && scratch && lda >= N, "col_reduce: null pointer / leading dimension");
    BLGNN_CHECK_ARG((mode != COL_DOT && mode != COL_DOT_NORM && mode != COL_DOT_NEG) || (b && ldb >= N), "col_reduce: dot modes need b");
    BLGNN_CHECK_ARG((mode != COL_SQDEV && mode != COL_DOT_NORM) || mean, "col_reduce: this mode needs the column means");
    BLGNN_CHECK_ARG(mode != COL_DOT_NORM || rstd, "col_reduce: dot_norm mode needs rstd");
    hipStream_t s = (hipStream_t)stream;
    const int rows = cr_rows(M);
    const int chunks = (M + rows - 1) / rows;
    if (chunks > 0) {
        dim3 grid((N + 63) / 64, chunks);
        if (mode == COL_SUM) hipLaunchKernelGGL(k_col_partial<COL_SUM>, grid, dim3(256), 0, s, a, lda, b, ldb, mean, rstd, M, N, rows, scratch);
        else if (mode == COL_DOT) hipLaunchKernelGGL(k_col_partial<COL_DOT>, grid, dim3(256), 0, s, a, lda, b, ldb, mean, rstd, M, N, rows, scratch);
        else if (mode == COL_SQDEV) hipLaunchKernelGGL(k_col_partial<COL_SQDEV>, grid, dim3(256), 0, s, a, lda, b, ldb, mean, rstd, M, N, rows, scratch);
        else if (mode == COL_DOT_NEG) hipLaunchKernelGGL(k_col_partial<COL_DOT_NEG>, grid, dim3(256), 0, s, a, lda, b, ldb, mean, rstd, M, N, rows, scratch);
        else hipLaunchKernelGGL(k_col_partial<COL_DOT_NORM>, grid, dim3(256), 0, s, a, lda, b, ldb, mean, rstd, M, N, rows, scratch);
    }
    hipLaunchKernelGGL(k_col_final, dim3((N + 63) / 64), dim3(256), 0, s, scratch, chunks, N, scale, out);
    return launch_status("col_reduce");
}

extern "C" int64_t blgnn_grad_weight_scratch_elems(int32_t M, int32_t O, int32_t I) {
    const int rows = gw_rows(M, O, I);
    return (int64_t)((M + rows - 1) / rows) * O * I;
}

extern "C" int blgnn_grad_weight(const float *g, int64_t ldg, const float *x, int64_t ldx, int32_t M, int32_t O, int32_t I,
                                 float *scratch, float *dw, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && O > 0 && I > 0 && ldg >= O && ldx >= I, "grad_weight: bad sizes");
    BLGNN_CHECK_ARG((M == 0 || (g && x)) && scratch && dw, "grad_weight: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int rows = gw_rows(M, O, I);
    const int chunks = (M + rows - 1) / rows;
    const bool vec = O % 4 == 0 && I % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 && aligned16(g) && aligned16(x);
    if (chunks > 0 && vec)
        hipLaunchKernelGGL(k_grad_weight_lds, dim3(chunks, ((O + 127) / 128) * ((I + 127) / 128)), dim3(256), 0, s, g, ldg, x,
                           ldx, M, O, I, rows, scratch);
    else if (chunks > 0)
        hipLaunchKernelGGL(k_grad_weight, dim3(chunks, ((O + 127) / 128) * ((I + 127) / 128)), dim3(256), 0, s, g, ldg, x, ldx,
                           M, O, I, rows, scratch);
    hipLaunchKernelGGL(k_col_final, dim3((O * I + 63) / 64), dim3(256), 0, s, scratch, chunks, O * I, 1.f, dw);
    return launch_status("grad_weight");
}

extern "C" int blgnn_bn_finalize(const float *mean, const float *var, int32_t N, float eps, float momentum, float unbias,
                                 float *rstd, float *running_mean, float *running_var, void *stream) {
    BLGNN_CHECK_ARG(N >= 0, "bn_finalize: negative size");
    if (N == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(var && rstd && ((!running_mean && !running_var) || mean), "bn_finalize: null pointer");
    hipLaunchKernelGGL(k_bn_finalize, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var, N, eps, momentum,
                       unbias, rstd, running_mean, running_var);
    return launch_status("bn_finalize");
}

extern "C" int blgnn_bn_apply(const float *x, int64_t ldx, const float *mean, const float *rstd, const float *gamma,
                              const float *beta, int32_t M, int32_t N, int32_t act, float *y, int64_t ldy,
                              void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N > 0 && ldx >= N && ldy >= N, "bn_apply: bad sizes");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH && act != BLGNN_ACT_PRELU, "bn_apply: bad activation id");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && mean && rstd && y, "bn_apply: null pointer");
    hipLaunchKernelGGL(k_bn_apply, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, mean, rstd,
                       gamma, beta, M, N, act, y, ldy);
    return launch_status("bn_apply");
}

extern "C" int blgnn_bn_backward(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *mean,
                                 const float *rstd, const float *gamma, const float *dgamma, const float *dbeta,
                                 int32_t M, int32_t N, int64_t count, float *dx, int64_t lddx, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N > 0 && lddy >= N && ldx >= N && lddx >= N && count >= M, "bn_backward: bad sizes");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(dy && x && mean && rstd && dgamma && dbeta && dx, "bn_backward: null pointer");
    hipLaunchKernelGGL(k_bn_backward, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, x, ldx,
                       mean, rstd, gamma, dgamma, dbeta, M, N, 1.f / (float)count, dx, lddx);
    return launch_status("bn_backward");
}

extern "C" int blgnn_act_backward(const float *dy, int64_t lddy, const float *y, int64_t ldy, int32_t act,
                                  const float *slope, int32_t M, int32_t N, float *dx, int64_t lddx, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N > 0 && lddy >= N && ldy >= N && lddx >= N, "act_backward: bad sizes");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "act_backward: bad activation id");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(dy && y && dx, "act_backward: null pointer");
    hipLaunchKernelGGL(k_act_backward, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, (hipStream_t)stream, dy, lddy, y, ldy,
                       act, slope, M, N, dx, lddx);
    return launch_status("act_backward");
}

extern "C" int blgnn_act_forward(const float *x, int64_t ldx, int32_t act, const float *slope, int32_t M, int32_t N,
                                 float *y, int64_t ldy, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N > 0 && ldx >= N && ldy >= N, "act_forward: bad sizes");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "act_forward: bad activation id");
    BLGNN_CHECK_ARG(act != BLGNN_ACT_PRELU || slope, "act_forward: PReLU needs its slopes");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && y, "act_forward: null pointer");
    hipLaunchKernelGGL(k_act_forward, dim3(grid_for((int64_t)M * N, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, act,
                       slope, M, N, y, ldy);
    return launch_status("act_forward");
}

extern "C" int blgnn_transpose(const float *x, int64_t ldx, int32_t M, int32_t N, float *y, int64_t ldy, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N >= 0 && ldx >= N && ldy >= M, "transpose: bad sizes");
    if (M == 0 || N == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && y && x != y, "transpose: null / aliased pointer");
    hipLaunchKernelGGL(k_transpose, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, ldx, M, N, y, ldy);
    return launch_status("transpose");
}

extern "C" int blgnn_segment_broadcast(const int32_t *seg, int32_t G, const float *dp, int64_t lddp,
                                       const int32_t *row_map, int32_t d, int32_t kind, float *dx, int64_t lddx,
                                       int32_t accumulate, void *stream) {
    BLGNN_CHECK_ARG(G >= 0 && d > 0 && lddp >= d && lddx >= d, "segment_broadcast: bad sizes");
    BLGNN_CHECK_ARG(kind == BLGNN_REDUCE_SUM || kind == BLGNN_REDUCE_MEAN, "segment_broadcast: bad reduce kind");
    if (G == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(seg && dp && dx, "segment_broadcast: null pointer");
    hipLaunchKernelGGL(k_segment_broadcast, dim3(G), dim3(256), 0, (hipStream_t)stream, seg, G, dp, lddp, row_map, d,
                       kind == BLGNN_REDUCE_MEAN, dx, lddx, accumulate);
    return launch_status("segment_broadcast");
}

extern "C" int blgnn_l2_normalize_backward(const float *z, int64_t ldz, const float *dzn, int64_t lddzn, int32_t M,
                                           int32_t d, float eps, float *dz, int64_t lddz, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && d > 0 && ldz >= d && lddzn >= d && lddz >= d && eps > 0.f, "l2_normalize_backward: bad sizes");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(z && dzn && dz, "l2_normalize_backward: null pointer");
    hipLaunchKernelGGL(k_l2_normalize_backward, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, ldz, dzn, lddzn, M, d,
                       eps, dz, lddz);
    return launch_status("l2_normalize_backward");
}

extern "C" int blgnn_pair_scatter_add(const float *dcat, int64_t ldc, int32_t col0, const int64_t *ids, int32_t B,
                                      int32_t d, float *dz, int64_t lddz, void *stream) {
    BLGNN_CHECK_ARG(B >= 0 && d > 0 && col0 >= 0 && ldc >= col0 + d && lddz >= d, "pair_scatter_add: bad sizes");
    if (B == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(dcat && ids && dz, "pair_scatter_add: null pointer");
    hipLaunchKernelGGL(k_pair_scatter_add, dim3(grid_for((int64_t)B * d, 256)), dim3(256), 0, (hipStream_t)stream, dcat, ldc,
                       col0, ids, B, d, dz, lddz);
    return launch_status("pair_scatter_add");
}
