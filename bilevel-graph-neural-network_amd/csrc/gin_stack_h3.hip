// The whole lower level in ONE launch with the dense transforms on the fp16 matrix cores at fp32
// accuracy (f16x3, h3_common.hpp): L stacked GIN layers + the (multi-scale) node -> graph readout
// (reference: model/layers.py:26-31,51-57 x L, model/layers_aggregation.py:27-42).
//
// The merged molecular graph is block diagonal, so a tile made of WHOLE molecules (<= 64 rows) has no
// neighbour outside itself at any depth.  A workgroup stages the tile's input features once and then
// runs layer after layer out of LDS:
//     CSR gather (LDS) -> row-normalise + split -> GEMM (3 fp16 MFMAs per product) -> act -> row maxima ->
//     split -> GEMM -> act + eval-BN -> fp32 tile in LDS -> per-molecule mean -> next layer
// Only the pooled rows [G, L*d] (and optionally the last activations) go to HBM: the L-1 intermediate
// activation round trips, L-1 launches / grid tails, the rowptr -> col -> far-row round trips of the
// per-layer kernel (blgnn_gin_layer_h3: a 32-row tile cut through molecules) and the separate readout
// launch disappear.  All four weight chunks of a GEMM are requested before the phase that precedes it
// (5-deep fragment ring), so the matrix cores do not wait for L2.
// Two LDS regions ping-pong (fp32 layer input / hidden pieces <-> aggregate pieces / fp32 layer
// output): 73 KB per workgroup at d=128, two workgroups per CU.
// Tiles are built on the host from the molecule sizes (greedy, consecutive molecules); molecules larger
// than a tile go through the per-layer kernels.
#include "gin_common.hpp"
#include "h3_common.hpp"

namespace blgnn {

constexpr int S3_MAX_LAYERS = 8;
constexpr int S3BM = 64;                          // rows per tile
constexpr int S3N = 128;                          // layer width
constexpr int S3RING = 5;                         // weight chunk buffers: a whole K <= 128 GEMM in flight
constexpr int S3COLS = 512;                       // column indices of a tile kept in LDS

struct Stack3Layer {
    const _Float16 *w1p, *w2p;                    // packed f16x3 images (blgnn_pack_weight_h3)
    const float *b1, *b2, *slope, *scale, *shift;
    int32_t act;
    float self_coef;
    int32_t pool_col;                             // column offset of this layer in `pooled`, -1 = not pooled
};

struct Stack3Args {
    const int32_t *rowptr, *col;
    const float *x0; int64_t ldx0; int32_t K0;
    int32_t n_tiles;
    const int32_t *tile_row;                      // [T+1] first row of each tile
    const int32_t *tile_mol;                      // [T+1] first molecule of each tile
    const int32_t *seg;                           // [G+1] molecule row offsets
    int32_t L;
    Stack3Layer layer[S3_MAX_LAYERS];
    float *pooled; int64_t ldp;
    const int32_t *row_map;                       // nullable: molecule -> pooled row
    int32_t mean;                                 // 1 = scatter_mean, 0 = scatter_add
    float *y_last; int64_t ldy;                   // nullable
};

#ifdef S3_STAMP
__device__ unsigned long long g_s3_stamps[96];
#define S3_ST() do { if (stamp_on && sti < 96) g_s3_stamps[sti] = __builtin_amdgcn_s_memtime(); ++sti; } while (0)
#else
#define S3_ST() do { } while (0)
#endif

typedef __amdgpu_buffer_rsrc_t s3rsrc_t;
__device__ __forceinline__ s3rsrc_t s3_rsrc(const _Float16 *Wp, int Kp) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(Wp), 0, HHDR * 2 + 2 * Kp * S3N * 2, 0x00020000);
}

// weight fragments of one 32-deep chunk, [step][piece]: lane (fi, fh) holds the 8 consecutive k of output
// column n_base + fi (the MFMA A operand of the transposed GEMM)
__device__ __forceinline__ void s3_load_chunk(f16x8 (&dst)[2][2], s3rsrc_t W, int Kp, int kc, unsigned lane_off) {
    const unsigned piece = (unsigned)Kp * S3N * 2u;               // bytes
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned soff = HHDR * 2u + (unsigned)q * piece + (unsigned)(kc / 8 + 2 * s) * (S3N * 16u);
            dst[s][q] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(W, lane_off, soff, 0));
        }
}

__device__ __forceinline__ void s3_prefetch(f16x8 (&w)[S3RING][2][2], s3rsrc_t Wp, int Kp, unsigned lane_off) {
#pragma unroll
    for (int u = 0; u < S3RING - 1; ++u)
        if (u * KC < Kp) s3_load_chunk(w[u], Wp, Kp, u * KC, lane_off);
}

// acc0 += w1.a1 ; acc1 += w1.a2 + w2.a1 for both 32-row halves of the tile; the lane ends up with
// n = n_base + 8 (r >> 2) + 4 fh + (r & 3), m = 32 mt + fi.  Kp <= 128: every chunk is already in the ring.
__device__ __forceinline__ void s3_gemm(f32x16 (&acc0)[2], f32x16 (&acc1)[2], const char *Ap, int sa, int Kp,
                                        int fi, int fh, const f16x8 (&w)[S3RING][2][2]) {
    const int piece_bytes = S3BM * sa;
    const char *arow = Ap + fi * sa + fh * 16;                   // lane (row fi, k half fh)
#pragma unroll
    for (int u = 0; u < S3RING - 1; ++u) {
        const int k = u * KC;
        if (k < Kp) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const f16x8 a1 = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + (k + 16 * s) * 2);
                    const f16x8 a2 = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + piece_bytes + (k + 16 * s) * 2);
                    acc0[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][s][0], a1, acc0[mt], 0, 0, 0);
                    acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][s][0], a2, acc1[mt], 0, 0, 0);
                    acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][s][1], a1, acc1[mt], 0, 0, 0);
                }
        }
    }
}

// agg[m] = self_coef * X[m] + sum_{j in row m, j != m} X[j] (everything inside the tile) -> row maximum ->
// power-of-two scale -> two fp16 pieces [2][S3BM][sa]; rows past the tile's end become zeros.
// LG lanes per row, NQ float4 per lane (columns 4 l + 64 q): 64 / LG rows of a wave in flight per pass.
// The row's first four neighbours come from the per-tile table ``nb`` (one byte each, local row or 0xFF),
// built once per tile: the CSR structure is the same for every layer, so the rowptr -> col -> row chain of
// dependent LDS reads is paid once, not per layer.  ``deg`` = CSR entries of the row (self loops included).
template <int LG, int NQ>
__device__ __forceinline__ void s3_gather_split(const float *Xs, int ldxs, char *Ap, int sa, int K, int Kp,
                                                const unsigned *nb, const int *deg, const int *rp, const int *cs,
                                                const int32_t *colg, int e_lo, bool cols_in_lds, int m0, int rows,
                                                float self_coef, float *ainv, int wid, int lane) {
    constexpr int RPW = WAVE / LG;
    constexpr int PASSES = S3BM / (4 * RPW);
    const int sub = lane / LG, l = lane % LG;
#pragma unroll 2
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * 4 + wid) * RPW + sub;
        const bool live = m < rows;
        const int row = m0 + m;
        const unsigned nbp = live ? nb[m] : 0xffffffffu;
        const int cnt = live ? deg[m] : 0;
        float4 acc[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c0 = 4 * l + 64 * q;
            acc[q] = *reinterpret_cast<const float4 *>(Xs + (live ? m : 0) * ldxs + (c0 < K ? c0 : 0));
            acc[q].x *= self_coef; acc[q].y *= self_coef; acc[q].z *= self_coef; acc[q].w *= self_coef;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned jl = (nbp >> (8 * t)) & 0xffu;
            const bool use = jl != 0xffu;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int c0 = 4 * l + 64 * q;
                const float4 u = *reinterpret_cast<const float4 *>(Xs + (use ? (int)jl : (live ? m : 0)) * ldxs + (c0 < K ? c0 : 0));
                f4_fma(acc[q], use ? 1.f : 0.f, u);
            }
        }
        if (cnt > 4) {                                           // rare: degree > 4, walk the rest of the CSR row
            const int e0 = rp[m];
            for (int e = e0 + 4; e < e0 + cnt; ++e) {
                const int jj = (cols_in_lds && e - e_lo < S3COLS) ? cs[e - e_lo] : colg[e];
                if (jj != row) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c0 = 4 * l + 64 * q;
                        f4_add(acc[q], *reinterpret_cast<const float4 *>(Xs + (jj - m0) * ldxs + (c0 < K ? c0 : 0)));
                    }
                }
            }
        }
        float amax = 0.f;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (!live || 4 * l + 64 * q >= K) acc[q] = f4_zero();
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc[q].x), fabsf(acc[q].y)), fmaxf(fabsf(acc[q].z), fabsf(acc[q].w))));
        }
#pragma unroll
        for (int off = 1; off < LG; off <<= 1) amax = fmaxf(amax, __shfl_xor(amax, off, WAVE));
        float scale, inv;
        pow2_scale(amax, scale, inv);
        if (l == 0) ainv[m] = inv;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c0 = 4 * l + 64 * q;
            if (c0 < Kp) {                                       // K padding columns of the pieces are zeroed
                const Split2 s0 = split2(acc[q].x * scale), s1 = split2(acc[q].y * scale), s2 = split2(acc[q].z * scale),
                             s3 = split2(acc[q].w * scale);
                f16x4 h1 = {s0.h1, s1.h1, s2.h1, s3.h1}, h2 = {s0.h2, s1.h2, s2.h2, s3.h2};
                char *dst = Ap + m * sa + 2 * c0;
                *reinterpret_cast<f16x4 *>(dst) = h1;
                *reinterpret_cast<f16x4 *>(dst + S3BM * sa) = h2;
            }
        }
    }
}

template <int LG0>
__global__ void __launch_bounds__(256, 2) k_gin_stack_h3(const Stack3Args A) {
    constexpr int N = S3N, LDO = N + 4, N4 = N / 4;
    constexpr int SH = N * 2 + 16;                              // piece row stride (bytes) for K = N
    constexpr int REG = 2 * S3BM * SH;                          // one ping-pong region: 34 816 B >= 64 x 132 x 4
    extern __shared__ float4 lds4[];
    char *lds = reinterpret_cast<char *>(lds4);
    int *rp = reinterpret_cast<int *>(lds + 2 * REG);          // [S3BM + 4]
    int *cs = rp + S3BM + 4;                                    // [S3COLS]
    float *ainv = reinterpret_cast<float *>(cs + S3COLS);      // [S3BM] 1 / row scale of the aggregate
    unsigned *hmax = reinterpret_cast<unsigned *>(ainv + S3BM);    // [S3BM] row maxima of the hidden tile (float bits)
    unsigned *nb = hmax + S3BM;                                 // [S3BM] first four neighbours of every row, one byte each
    int *deg = reinterpret_cast<int *>(nb + S3BM);              // [S3BM] CSR entries per row
    float *part = reinterpret_cast<float *>(deg + S3BM);        // [8][N] readout partial sums
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int n_base = wid * 32;
    const unsigned lane_off = ((unsigned)fh * N + (unsigned)(n_base + fi)) * 16u;
    const int K0p = (A.K0 + KC - 1) / KC * KC;
    const int sa0 = K0p * 2 + 16;

#ifdef S3_STAMP
    const bool stamp_on = blockIdx.x == 37 && threadIdx.x == 0;
    int sti = 0;
#endif
    for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
        S3_ST();
        const int m0 = A.tile_row[tile], rows = A.tile_row[tile + 1] - m0;
        const int g0 = A.tile_mol[tile], g1 = A.tile_mol[tile + 1];
        __syncthreads();                                        // previous tile fully consumed
        // ---- stage: rowptr slice, layer-0 weight chunks, input rows, column indices --------------------
        if (threadIdx.x <= S3BM) rp[threadIdx.x] = A.rowptr[m0 + (threadIdx.x < rows ? threadIdx.x : rows)];
        f16x8 wf[S3RING][2][2];
        s3_prefetch(wf, s3_rsrc(A.layer[0].w1p, K0p), K0p, lane_off);
        {
            const int k4 = A.K0 >> 2;
            float *Xs = reinterpret_cast<float *>(lds + REG);
            for (int idx = threadIdx.x; idx < rows * k4; idx += 256) {
                const int r = idx / k4, c4 = idx - r * k4;
                *reinterpret_cast<float4 *>(Xs + r * A.K0 + 4 * c4) =
                    *reinterpret_cast<const float4 *>(A.x0 + (int64_t)(m0 + r) * A.ldx0 + 4 * c4);
            }
        }
        __syncthreads();
        const int e_lo = rp[0];
        const int ncol_all = rp[rows] - e_lo;
        const bool cols_in_lds = ncol_all <= S3COLS;
        if (cols_in_lds)
            for (int i = threadIdx.x; i < ncol_all; i += 256) cs[i] = A.col[e_lo + i];
        if (threadIdx.x < S3BM) hmax[threadIdx.x] = 0u;
        __syncthreads();
        if (threadIdx.x < S3BM) {                               // neighbour table (layer invariant)
            const int m = threadIdx.x;
            unsigned pk = 0xffffffffu;
            int cnt = 0;
            if (m < rows) {
                const int e0 = rp[m];
                cnt = rp[m + 1] - e0;
                pk = 0u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int c = m0 + m;
                    if (q < cnt) c = (cols_in_lds && e0 + q - e_lo < S3COLS) ? cs[e0 + q - e_lo] : A.col[e0 + q];
                    const unsigned jl = c != m0 + m ? (unsigned)(c - m0) : 0xffu;
                    pk |= jl << (8 * q);
                }
            }
            nb[m] = pk;
            deg[m] = cnt;
        }
        __syncthreads();

        for (int l = 0; l < A.L; ++l) {
            S3_ST();
            const Stack3Layer &P = A.layer[l];
            // layer input (fp32) lives in R[(l+1)&1]; aggregate pieces and the layer output go to R[l&1]
            char *Ra = lds + ((l & 1) ? REG : 0), *Rb = lds + ((l & 1) ? 0 : REG);
            const float *Xs = reinterpret_cast<const float *>(Rb);
            char *Ap = Ra, *Hp = Rb;
            float *Ys = reinterpret_cast<float *>(Ra);
            const int K = l == 0 ? A.K0 : N, Kp = l == 0 ? K0p : N;
            const int sa = l == 0 ? sa0 : SH, ldxs = l == 0 ? A.K0 : LDO;
            const float slope_s = P.act == BLGNN_ACT_RELU ? 0.f : 1.f;       // relu / identity / prelu: one leaky form
            const bool prelu = P.act == BLGNN_ACT_PRELU && P.slope;
            const float w1inv = *reinterpret_cast<const float *>(P.w1p), w2inv = *reinterpret_cast<const float *>(P.w2p);
            // ---- (1) gather inside the tile, scale, split --------------------------------------------------
            if (l == 0)
                s3_gather_split<16, LG0 / 16>(Xs, ldxs, Ap, sa, K, Kp, nb, deg, rp, cs, A.col, e_lo, cols_in_lds, m0, rows,
                                              P.self_coef, ainv, wid, lane);
            else
                s3_gather_split<16, 2>(Xs, ldxs, Ap, sa, K, Kp, nb, deg, rp, cs, A.col, e_lo, cols_in_lds, m0, rows,
                                       P.self_coef, ainv, wid, lane);
            S3_ST();
            lds_barrier();
            S3_ST();
            // ---- (2) hidden = act(agg @ W1 + b1): row maxima -> scale -> split -> park ---------------------
            f32x16 acc0[2], acc1[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[mt][r] = 0.f; acc1[mt][r] = 0.f; }
            float4 pb[4], psl[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n_base + 8 * g + 4 * fh;             // four consecutive output columns
                pb[g] = P.b1 ? *reinterpret_cast<const float4 *>(P.b1 + c) : f4_zero();
                psl[g] = prelu ? *reinterpret_cast<const float4 *>(P.slope + c)
                               : float4{slope_s, slope_s, slope_s, slope_s};
            }
            s3_gemm(acc0, acc1, Ap, sa, Kp, fi, fh, wf);
            S3_ST();
            s3_prefetch(wf, s3_rsrc(P.w2p, N), N, lane_off);
#define S3_LEAKY(v, s) ((v) > 0.f ? (v) : (s) * (v))
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float rs = ainv[mt * 32 + fi] * w1inv;
                float hm = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v;
                    v.x = fmaf(acc0[mt][4 * g + 0] + acc1[mt][4 * g + 0], rs, pb[g].x); v.x = S3_LEAKY(v.x, psl[g].x);
                    v.y = fmaf(acc0[mt][4 * g + 1] + acc1[mt][4 * g + 1], rs, pb[g].y); v.y = S3_LEAKY(v.y, psl[g].y);
                    v.z = fmaf(acc0[mt][4 * g + 2] + acc1[mt][4 * g + 2], rs, pb[g].z); v.z = S3_LEAKY(v.z, psl[g].z);
                    v.w = fmaf(acc0[mt][4 * g + 3] + acc1[mt][4 * g + 3], rs, pb[g].w); v.w = S3_LEAKY(v.w, psl[g].w);
                    acc0[mt][4 * g + 0] = v.x; acc0[mt][4 * g + 1] = v.y; acc0[mt][4 * g + 2] = v.z; acc0[mt][4 * g + 3] = v.w;
                    hm = fmaxf(hm, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
                hm = fmaxf(hm, __shfl_xor(hm, 32, WAVE));           // the row's other 16 columns of this wave
                if (fh == 0) atomicMax(&hmax[mt * 32 + fi], __float_as_uint(hm));   // across the four column waves
            }
            S3_ST();
            lds_barrier();                                         // maxima complete; every wave is done reading Ap / Xs
            S3_ST();
            float hinv[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                float scale;
                pow2_scale(__uint_as_float(hmax[mt * 32 + fi]), scale, hinv[mt]);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = n_base + 8 * g + 4 * fh;
                    const Split2 s0 = split2(acc0[mt][4 * g + 0] * scale), s1 = split2(acc0[mt][4 * g + 1] * scale),
                                 s2 = split2(acc0[mt][4 * g + 2] * scale), s3 = split2(acc0[mt][4 * g + 3] * scale);
                    f16x4 h1 = {s0.h1, s1.h1, s2.h1, s3.h1}, h2 = {s0.h2, s1.h2, s2.h2, s3.h2};
                    char *dst = Hp + (mt * 32 + fi) * SH + c * 2;
                    *reinterpret_cast<f16x4 *>(dst) = h1;
                    *reinterpret_cast<f16x4 *>(dst + S3BM * SH) = h2;
                }
            }
            S3_ST();
            lds_barrier();
            S3_ST();
            if (threadIdx.x < S3BM) hmax[threadIdx.x] = 0u;        // for the next layer (read again two barriers on)
            // ---- (3) y = affine(act(hidden @ W2 + b2)) -> fp32 tile = the next layer's input ------------------
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[mt][r] = 0.f; acc1[mt][r] = 0.f; }
            float4 psc[4], psh[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n_base + 8 * g + 4 * fh;
                pb[g] = P.b2 ? *reinterpret_cast<const float4 *>(P.b2 + c) : f4_zero();
                psc[g] = float4{1.f, 1.f, 1.f, 1.f};
                psh[g] = f4_zero();
                if (P.scale) psc[g] = *reinterpret_cast<const float4 *>(P.scale + c);
                if (P.scale && P.shift) psh[g] = *reinterpret_cast<const float4 *>(P.shift + c);
            }
            s3_gemm(acc0, acc1, Hp, SH, N, fi, fh, wf);
            S3_ST();
            if (l + 1 < A.L) s3_prefetch(wf, s3_rsrc(A.layer[l + 1].w1p, N), N, lane_off);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float rs = hinv[mt] * w2inv;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = n_base + 8 * g + 4 * fh;
                    float4 v;
                    v.x = fmaf(acc0[mt][4 * g + 0] + acc1[mt][4 * g + 0], rs, pb[g].x); v.x = S3_LEAKY(v.x, psl[g].x);
                    v.y = fmaf(acc0[mt][4 * g + 1] + acc1[mt][4 * g + 1], rs, pb[g].y); v.y = S3_LEAKY(v.y, psl[g].y);
                    v.z = fmaf(acc0[mt][4 * g + 2] + acc1[mt][4 * g + 2], rs, pb[g].z); v.z = S3_LEAKY(v.z, psl[g].z);
                    v.w = fmaf(acc0[mt][4 * g + 3] + acc1[mt][4 * g + 3], rs, pb[g].w); v.w = S3_LEAKY(v.w, psl[g].w);
                    v.x = fmaf(v.x, psc[g].x, psh[g].x); v.y = fmaf(v.y, psc[g].y, psh[g].y);
                    v.z = fmaf(v.z, psc[g].z, psh[g].z); v.w = fmaf(v.w, psc[g].w, psh[g].w);
                    *reinterpret_cast<float4 *>(Ys + (mt * 32 + fi) * LDO + c) = v;
                }
            }
#undef S3_LEAKY
            S3_ST();
            lds_barrier();
            S3_ST();
            // ---- (4) per-molecule readout of this layer (multi-scale column slice) ------------------------------
            if (P.pool_col >= 0) {
                // eight row groups side by side (thread = column quad x row group), independent float4 reads,
                // then 128 threads add the eight partial sums of a molecule in a fixed order
                const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
                for (int g = g0; g < g1; ++g) {
                    const int r0 = A.seg[g] - m0, r1 = A.seg[g + 1] - m0;
                    float4 sacc = f4_zero();
                    for (int r = r0 + rg; r < r1; r += 8) f4_add(sacc, *reinterpret_cast<const float4 *>(Ys + r * LDO + 4 * c4));
                    *reinterpret_cast<float4 *>(part + rg * N + 4 * c4) = sacc;
                    lds_barrier();
                    if (threadIdx.x < N) {
                        const int c = threadIdx.x;
                        float v = ((part[c] + part[N + c]) + (part[2 * N + c] + part[3 * N + c])) +
                                  ((part[4 * N + c] + part[5 * N + c]) + (part[6 * N + c] + part[7 * N + c]));
                        if (A.mean) v = v / (float)((r1 - r0) < 1 ? 1 : (r1 - r0));
                        const int orow = A.row_map ? A.row_map[g] : g;
                        A.pooled[(int64_t)orow * A.ldp + P.pool_col + c] = v;
                    }
                    lds_barrier();
                }
            }
            if (l + 1 == A.L && A.y_last) {
                for (int idx = threadIdx.x; idx < rows * N4; idx += 256) {
                    const int m = idx / N4, c4 = idx % N4;
                    *reinterpret_cast<float4 *>(A.y_last + (int64_t)(m0 + m) * A.ldy + 4 * c4) =
                        *reinterpret_cast<const float4 *>(Ys + m * LDO + 4 * c4);
                }
            }
            S3_ST();
            // the next gather reads Ys (behind the barrier above) and writes pieces into the other region, whose
            // last readers (GEMM2 on Hp) are behind the same barrier
        }
    }
}

static size_t stack3_lds_bytes() {
    return (size_t)2 * (2 * S3BM * (S3N * 2 + 16)) + sizeof(int) * (S3BM + 4 + S3COLS) + sizeof(float) * 4 * S3BM +
           sizeof(float) * 8 * S3N;
}

template <int LG0>
static int launch_stack3(const Stack3Args &A, hipStream_t s) {
    static bool declared = false;
    const size_t bytes = stack3_lds_bytes();
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gin_stack_h3<LG0>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gin_stack_h3: cannot raise the dynamic LDS limit to %zu bytes", bytes);
        declared = true;
    }
    const int blocks = A.n_tiles < 256 * 2 ? A.n_tiles : 256 * 2;
    hipLaunchKernelGGL((k_gin_stack_h3<LG0>), dim3(blocks), dim3(256), bytes, s, A);
    return launch_status("gin_stack_h3");
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_gin_stack_h3(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x0,
                                  int64_t ldx0, int32_t K0, int32_t N, int32_t num_tiles,
                                  const int32_t *tile_row, const int32_t *tile_mol, const int32_t *seg,
                                  int32_t num_layers, const void *const *w1_packed_host, const float *const *b1_host,
                                  const void *const *w2_packed_host, const float *const *b2_host,
                                  const float *const *slope_host, const float *const *scale_host,
                                  const float *const *shift_host, const int32_t *act_host,
                                  const float *eps_host, const int32_t *pool_col_host, int32_t reduce_kind,
                                  float *pooled, int64_t ldp, const int32_t *row_map, float *y_last,
                                  int64_t ldy, void *stream) {
    BLGNN_CHECK_ARG(num_tiles >= 0 && K0 > 0 && N > 0 && nnz >= 0, "gin_stack_h3: bad sizes");
    if (num_tiles == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(num_layers >= 1 && num_layers <= S3_MAX_LAYERS, "gin_stack_h3: 1..8 layers");
    BLGNN_CHECK_ARG(rowptr && (col || nnz == 0) && x0 && tile_row && tile_mol && seg, "gin_stack_h3: null pointer");
    BLGNN_CHECK_ARG(w1_packed_host && b1_host && w2_packed_host && b2_host && slope_host && scale_host && shift_host &&
                    act_host && eps_host && pool_col_host, "gin_stack_h3: null parameter table");
    BLGNN_CHECK_ARG(reduce_kind == BLGNN_REDUCE_SUM || reduce_kind == BLGNN_REDUCE_MEAN, "gin_stack_h3: bad reduce kind");
    if (N != S3N || K0 % 4 != 0 || K0 > 128 || ldx0 % 4 != 0 || !aligned16(x0) ||
        (y_last && (ldy % 4 != 0 || !aligned16(y_last))))
        return fail(BLGNN_ERR_UNSUPPORTED, "gin_stack_h3: needs N == 128, K0 %% 4 == 0, K0 <= 128 and float4-aligned "
                    "x0 / y_last (got N=%d K0=%d); use blgnn_gin_stack", N, K0);
    Stack3Args A = {};
    A.rowptr = rowptr; A.col = col; A.x0 = x0; A.ldx0 = ldx0; A.K0 = K0; A.n_tiles = num_tiles;
    A.tile_row = tile_row; A.tile_mol = tile_mol; A.seg = seg; A.L = num_layers;
    bool any_pool = false;
    for (int l = 0; l < num_layers; ++l) {
        BLGNN_CHECK_ARG(w1_packed_host[l] && w2_packed_host[l], "gin_stack_h3: null weight in layer %d", l);
        BLGNN_CHECK_ARG(aligned16(w1_packed_host[l]) && aligned16(w2_packed_host[l]), "gin_stack_h3: packed weights must be 16-byte aligned");
        BLGNN_CHECK_ARG(act_host[l] >= 0 && act_host[l] <= BLGNN_ACT_TANH, "gin_stack_h3: bad activation id");
        if (act_host[l] == BLGNN_ACT_SIGMOID || act_host[l] == BLGNN_ACT_TANH)
            return fail(BLGNN_ERR_UNSUPPORTED, "gin_stack_h3: sigmoid / tanh layers take the per-layer kernels");
        BLGNN_CHECK_ARG(act_host[l] != BLGNN_ACT_PRELU || slope_host[l], "gin_stack_h3: PReLU layer without slopes");
        Stack3Layer &P = A.layer[l];
        P.w1p = reinterpret_cast<const _Float16 *>(w1_packed_host[l]);
        P.w2p = reinterpret_cast<const _Float16 *>(w2_packed_host[l]);
        P.b1 = b1_host[l]; P.b2 = b2_host[l];
        P.slope = slope_host[l]; P.scale = scale_host[l]; P.shift = shift_host[l];
        P.act = act_host[l]; P.self_coef = 1.f + eps_host[l]; P.pool_col = pool_col_host[l];
        any_pool |= P.pool_col >= 0;
    }
    BLGNN_CHECK_ARG(!any_pool || pooled, "gin_stack_h3: pooled output required");
    BLGNN_CHECK_ARG(any_pool || y_last, "gin_stack_h3: nothing to write (no pooled layer, no y_last)");
    A.pooled = pooled; A.ldp = ldp; A.row_map = row_map; A.mean = reduce_kind == BLGNN_REDUCE_MEAN;
    A.y_last = y_last; A.ldy = ldy;
    hipStream_t s = (hipStream_t)stream;
    const int k4 = (K0 + 3) / 4;
    if (k4 <= 16) return launch_stack3<16>(A, s);
    return launch_stack3<32>(A, s);
}
