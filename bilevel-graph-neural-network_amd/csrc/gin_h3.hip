// Fused GIN layer with the feature transform on the fp16 matrix cores at fp32 accuracy ("f16x3").
//
// Every fp32 operand is split into TWO fp16 pieces x = h1 + h2 (11 + 11 significand bits; the
// representation error is 2^-23 |x|, fp32's own half-ulp) and a product a*w is the three fp16 MFMAs
// a1*w1 + a1*w2 + a2*w1, accumulated in fp32; the dropped a2*w2 is at most 2^-22 |a||w|.  Against the
// three-piece bf16 scheme (gin_x6.hip) this is half the matrix-core work and two thirds of the
// operand bytes in LDS and of the weight-fragment traffic (the 32-row kernels are bound by that
// traffic: 196 KB of bf16 fragments per tile through a 64 B/clk L1 fill path), and the smaller
// images (37 KB of LDS, 128 VGPRs) let four workgroups share a CU instead of three.
//
// fp16 has a 5-bit exponent, so operands are range-normalised by exact powers of two:
//   * every activation row is scaled so that its largest magnitude lies in [2^14, 2^15) (row maximum
//     by cross-lane reduction; the inverse is applied to the fp32 accumulators in the epilogue);
//   * every weight matrix is scaled the same way as a whole when it is packed (blgnn_pack_weight_h3
//     stores the inverse scale in the packed image's header).
// Elements more than 2^13 below their row's (matrix's) maximum lose low-order bits of h2 -- an error
// below 2^-24 of that maximum per element, i.e. of the order of fp32 rounding of the dot product.
//
// Structure = gin_x6.hip: stage tile -> LDS gather -> split -> GEMM -> act -> split -> GEMM ->
// act + eval-BN -> row stores; GEMMs transposed (weight fragment = MFMA A operand) so a lane owns
// four consecutive output columns of one row.
#include "gin_common.hpp"
#include "h3_common.hpp"
#include "sp_common.hpp"

namespace blgnn {

#ifdef GIN_STAMP
__device__ unsigned long long g_gin_stamps[16];    // diagnostics: 100 MHz ticks at the phase boundaries of one workgroup
#ifndef GIN_STAMP_N
#define GIN_STAMP_N 256
#endif
__device__ unsigned long long g_gin_tl[8192][12];  // the same for every workgroup of the N = 256 kernel; [11] = HW_ID | XCC_ID << 32
#define GIN_TICK(i) do { if (N == GIN_STAMP_N && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
        if (blockIdx.x == gridDim.x / 2 + 5) g_gin_stamps[i] = t_; \
        if (N == 256 && blockIdx.x < 8192) { g_gin_tl[blockIdx.x][i] = t_; \
            if ((i) == 0) g_gin_tl[blockIdx.x][11] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); } } } while (0)
#else
#define GIN_TICK(i) do { } while (0)
#endif
constexpr int HCOLS = 512;       // column indices of a tile kept in LDS (64 rows x mean degree 2.1 = 136)

// ---- weight packing: W[K,N] fp32 -> header + [2][Kp/8][N][8] fp16, scaled by a power of two -------------
__global__ void k_absmax(const float *__restrict__ w, int64_t ldw, int K, int N, unsigned *out) {
    float m = 0.f;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)K * N;
         idx += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[(idx / N) * ldw + idx % N]));
    m = group_max(m, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order as uints
}

__global__ void k_pack_h3(const float *__restrict__ w, int64_t ldw, int K, int N, int Kp,
                          const unsigned *__restrict__ amax_bits, _Float16 *__restrict__ out) {
    float scale, inv;
    pow2_scale(__uint_as_float(*amax_bits), scale, inv);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // (kb, n)
    if (idx == 0) *reinterpret_cast<float *>(out) = inv;
    const int kb = idx / N, n = idx - kb * N;
    if (kb >= Kp / 8) return;
    f16x8 p[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kb * 8 + j;
        const Split2 s = split2(k < K ? w[(int64_t)k * ldw + n] * scale : 0.f);
        p[0][j] = s.h1; p[1][j] = s.h2;
    }
    const size_t piece = (size_t)(Kp / 8) * N;
#pragma unroll
    for (int q = 0; q < 2; ++q)
        *reinterpret_cast<f16x8 *>(out + HHDR + ((size_t)q * piece + (size_t)kb * N + n) * 8) = p[q];
}

struct GinH3Args {
    const int32_t *rowptr, *col;
    const float *x; int64_t ldx;
    int32_t n, K, Kp;                        // Kp = K padded to 32
    const _Float16 *w1p, *w2p;               // packed images (header + pieces)
    const float *b1, *b2;
    float self_coef;
    Epilogue epi;
    float *y; int64_t ldy;
    // categorical input (first lower layer): 8 bytes per atom = one code per attribute (<= 8), expanded to the
    // one-hot row of utils/data/representation_node_feat.py:86-97 on the fly; x / ldx unused then
    const uint2 *codes;
    uint8_t code_off[8];                     // first column of every attribute's one-hot block
    int32_t n_attr;
    // fused node -> graph readout (model/layers_aggregation.py:16-19,32-41): pool[row_graph[i], :] += y[i, :] * w
    const int32_t *row_graph, *seg;          // graph of every row (sorted) / graph row ranges [G + 1]
    float *pool; int64_t ldp;                // zero-initialised by the caller; nullptr = no readout
    int32_t pool_mean;                       // w = 1 / max(rows of the graph, 1), else 1
    float *y_amax;                           // optional [n]: max |y[i, :]| per row (row scale of a following f16x3 GEMM)
    _Float16 *ysp; int64_t ldysp;            // optional (N == 256): y also in the sp format (sp_common.hpp) ...
    uint8_t *yspe; int64_t ldyspe;           //   ... with its block scale exponent per row, for blgnn_linear_sp
};

// columns 4 l .. 4 l + 3 of the one-hot row of an atom
__device__ __forceinline__ float4 onehot4(const uint2 c, const uint8_t (&off)[8], int n_attr, int l) {
    float4 v = f4_zero();
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        if (a < n_attr) {
            const unsigned code = ((a < 4 ? c.x : c.y) >> (8 * (a & 3))) & 0xffu;
            const int rel = (int)off[a] + (int)code - 4 * l;
            v.x = rel == 0 ? 1.f : v.x; v.y = rel == 1 ? 1.f : v.y;
            v.z = rel == 2 ? 1.f : v.z; v.w = rel == 3 ? 1.f : v.w;
        }
    }
    return v;
}

typedef __amdgpu_buffer_rsrc_t hrsrc_t;
__device__ __forceinline__ hrsrc_t h3_rsrc(const _Float16 *Wp, int Kp, int N) {   // N = packed row width
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(Wp), 0, HHDR * 2 + 2 * Kp * N * 2, 0x00020000);
}

// Weight fragments of one 32-deep chunk: [step][piece].  Lane (fi, fh) holds the 8 consecutive k of
// output column n_base + fi (the MFMA A operand).  Buffer addressing: one per-lane VGPR offset +
// a wave-uniform SGPR offset per load.
template <int N>
__device__ __forceinline__ void load_b_h3(f16x8 (&dst)[2][2], hrsrc_t W, int Kp, int kc, unsigned lane_off) {
    const unsigned piece = (unsigned)Kp * N * 2u;                 // bytes
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned soff = HHDR * 2u + (unsigned)q * piece + (unsigned)(kc / 8 + 2 * s) * (N * 16u);
            const u32x4h v = __builtin_amdgcn_raw_buffer_load_b128(W, lane_off, soff, 0);
            dst[s][q] = __builtin_bit_cast(f16x8, v);
        }
}

// weight chunk buffers (16 VGPRs each): HR - 1 chunks are requested ahead of the MFMAs.  N = 128 (128 VGPRs, four
// workgroups per CU): 3.  N = 256 (one 512-thread workgroup per CU): GIN_HR256 -- depths 3 / 4 / 5 / 6 (209 .. 255 VGPRs,
// no spills) measured 878 / 894 / 866-885 / 892 us on 499 k rows: the weight stream is not what the layer waits for.
// Where a 64-row tile's 26-27 us go at K = N = 256 (GIN_STAMP build, csrc/diag/gin_stamp.hip; one workgroup per CU, so the
// phases of a tile are strictly serial): rows + CSR slice staged 4.7-5.0, gather + row scale + split 7.2 (NOT the
// out-of-tile neighbours: with 64-atom molecules, none of which straddles a tile, the phase still takes 7.2 -- it is
// eight passes per wavefront of LDS reads, a 64-lane row maximum and the two-piece split, ~80 vector instructions per
// row at two wavefronts per SIMD), GEMM1 3.7-4.0 (matrix-core bound: 3.4), act + row maxima + barrier 2.2, split of the
// hidden tile 0.9-1.0, GEMM2 4.9-5.0, staging + row stores 2.3-2.7.
#ifndef GIN_HR256
#define GIN_HR256 3
#endif
template <int N> struct HRing { static constexpr int V = N == 256 ? GIN_HR256 : 3; };

template <int N>
__device__ __forceinline__ void h3_prefetch(f16x8 (&w)[HRing<N>::V][2][2], hrsrc_t Wp, int Kp, unsigned lane_off) {
    constexpr int HRING = HRing<N>::V;
#pragma unroll
    for (int u = 0; u < HRING - 1; ++u)
        if (u * KC < Kp) load_b_h3<N>(w[u], Wp, Kp, u * KC, lane_off);
}

// acc0 += w1.a1 ; acc1 += w1.a2 + w2.a1, for both 32-row halves of the tile (transposed GEMM: the
// lane ends up with n = n_base + 8 (r >> 2) + 4 fh + (r & 3), m = 32 mt + fi)
template <int MT, int N>
__device__ __forceinline__ void gemm_h3(f32x16 (&acc0)[MT], f32x16 (&acc1)[MT], const char *Ap, int sa, hrsrc_t Wp,
                                        int Kp, unsigned lane_off, int fi, int fh, f16x8 (&w)[HRing<N>::V][2][2]) {
    constexpr int HRING = HRing<N>::V;
    const int piece_bytes = MT * 32 * sa;
    const char *arow = Ap + fi * sa + fh * 16;                   // lane (row fi, k half fh)
    for (int kc = 0; kc < Kp; kc += HRING * KC) {
#pragma unroll
        for (int u = 0; u < HRING; ++u) {
            const int k = kc + u * KC;
            if (k < Kp) {
                if (k + (HRING - 1) * KC < Kp) load_b_h3<N>(w[(u + HRING - 1) % HRING], Wp, Kp, k + (HRING - 1) * KC, lane_off);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f16x8 a1 = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + (k + 16 * s) * 2);
                        const f16x8 a2 = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + piece_bytes + (k + 16 * s) * 2);
                        acc0[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][s][0], a1, acc0[mt], 0, 0, 0);
                        acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][s][0], a2, acc1[mt], 0, 0, 0);
                        acc1[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][s][1], a1, acc1[mt], 0, 0, 0);
                    }
            }
        }
    }
}

// Two shapes are launched.  N = 128: XBM = 32 rows per 256-thread workgroup, four workgroups per CU.
// N = 256 (K <= 256): XBM = 64 rows per 512-thread workgroup (NW = 8 waves, one 32-column group each), one
// workgroup per CU: every weight fragment serves two row tiles, i.e. 8.2 KB of weight traffic per row
// against 24.6 KB for the bf16x6 kernel's 32-row tiles -- at d = 256 that traffic (L2 -> CU operand
// delivery, ~18 B/clk/CU), not the matrix cores, sets the time of these layers.
template <int LG, int ACT, int XBM, bool CODES = false, int N = 128, bool POOL = false>
__global__ void __launch_bounds__(N * 2, (XBM == 64 || N == 256) ? 2 : 4) k_gin_layer_h3(const GinH3Args A) {
    constexpr int MT = XBM / 32;
    constexpr int NW = N / 32, NTH = 64 * NW;               // waves / threads per workgroup
    constexpr int RPW = WAVE / LG;
    constexpr int PASSES = XBM / (NW * RPW);
    constexpr int LDO = N + 4;
    constexpr int SH = N * 2 + 16;                  // hidden piece row stride (bytes)
    constexpr int NCV = (HCOLS + NTH - 1) / NTH;
    constexpr int N4 = N / 4;
    extern __shared__ float4 lds4[];
    char *lds = reinterpret_cast<char *>(lds4);
    GIN_TICK(0);
    const int sa = A.Kp * 2 + 16;                   // aggregated piece row stride (bytes)
    // region 0: aggregated pieces [2][XBM][sa], later the fp32 output staging tile [XBM][N+4]
    // region 1: staged fp32 input rows [XBM][K], later the hidden pieces [2][XBM][SH]
    int r0 = 2 * XBM * sa;
    if (XBM * LDO * 4 > r0) r0 = XBM * LDO * 4;
    int r1 = XBM * A.K * 4;
    if (2 * XBM * SH > r1) r1 = 2 * XBM * SH;
    char *Ap = lds;
    float *Ys = reinterpret_cast<float *>(lds);
    float *Xs = reinterpret_cast<float *>(lds + r0);
    char *Hp = lds + r0;
    int *rp = reinterpret_cast<int *>(lds + r0 + r1);       // [XBM + 4]
    int *cs = rp + XBM + 4;                                 // [HCOLS]
    float *ainv = reinterpret_cast<float *>(cs + HCOLS);    // [XBM] 1 / row scale of the aggregate
    unsigned *hmax = reinterpret_cast<unsigned *>(ainv + XBM);   // [XBM] row maxima of the hidden tile (float bits)
    int *rgraph = reinterpret_cast<int *>(hmax + XBM);           // [XBM] graph of every tile row (fused readout)
    float *rweight = reinterpret_cast<float *>(rgraph + XBM);    // [XBM] readout weight of every tile row
    int *nb = reinterpret_cast<int *>(rweight + XBM);            // [XBM][4] first four neighbours of every tile row, resolved
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int sub = lane / LG, l = lane % LG;
    const int n_base = wid * 32;
    const hrsrc_t W1 = h3_rsrc(A.w1p, A.Kp, N), W2 = h3_rsrc(A.w2p, N, N);
    const unsigned lane_off = ((unsigned)fh * N + (unsigned)(n_base + fi)) * 16u;
    const int k4n = A.K >> 2;
    const bool ok = l < k4n;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;
    const int m0 = blockIdx.x * XBM;
    const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);
    auto load_x4 = [&](int row) -> float4 {                  // columns 4 l .. 4 l + 3 of feature row `row`
        if (CODES) return onehot4(A.codes[row], A.code_off, A.n_attr, l);
        return X[(int64_t)row * ldx4 + l];
    };

    // ---- (0) requests: rowptr slice, feature rows, weight chunks, edge range -> column indices ---------------
    int rpv = 0;
    if (threadIdx.x <= XBM) {
        const int r = m0 + threadIdx.x;
        rpv = A.rowptr[r < A.n ? r : A.n];
    }
    float4 xin[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int row = m0 + (p * NW + wid) * RPW + sub;
        xin[p] = (ok && row < A.n) ? load_x4(row) : f4_zero();
    }
    f16x8 wfrag[HRing<N>::V][2][2];
    h3_prefetch<N>(wfrag, W1, A.Kp, lane_off);
    const float w1inv = *reinterpret_cast<const float *>(A.w1p), w2inv = *reinterpret_cast<const float *>(A.w2p);
    const int e_lo = A.rowptr[m0];                                   // uniform addresses: scalar loads
    const int ncol_all = A.rowptr[m0 + XBM < A.n ? m0 + XBM : A.n] - e_lo;
    const bool cols_in_lds = ncol_all <= HCOLS;
    int cv[NCV];
#pragma unroll
    for (int i = 0; i < NCV; ++i) {
        const int idx = threadIdx.x + NTH * i;
        cv[i] = (cols_in_lds && idx < ncol_all) ? A.col[e_lo + idx] : 0;
    }
    if (threadIdx.x <= XBM) rp[threadIdx.x] = rpv;
    if (threadIdx.x < XBM) hmax[threadIdx.x] = 0u;
    if (POOL && threadIdx.x >= NTH - XBM) {          // the last XBM threads: their other start-up loads are fewest
        const int m = threadIdx.x - (NTH - XBM);
        const int r = m0 + m;
        const int g = A.row_graph[r < A.n ? r : A.n - 1];
        rgraph[m] = g;
        rweight[m] = A.pool_mean ? 1.f / (float)max(A.seg[g + 1] - A.seg[g], 1) : 1.f;
    }
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * NW + wid) * RPW + sub;
        if (ok) *reinterpret_cast<float4 *>(Xs + m * A.K + 4 * l) = xin[p];
    }
#pragma unroll
    for (int i = 0; i < NCV; ++i) {
        const int idx = threadIdx.x + NTH * i;
        if (cols_in_lds && idx < ncol_all) cs[idx] = cv[i];
    }
    // zero the K padding columns of the two aggregated pieces
    for (int idx = threadIdx.x; idx < 2 * XBM * (A.Kp - A.K); idx += NTH) {
        const int per = XBM * (A.Kp - A.K);
        const int q = idx / per, rem = idx - q * per;
        const int m = rem / (A.Kp - A.K), k = A.K + rem % (A.Kp - A.K);
        *reinterpret_cast<_Float16 *>(Ap + q * XBM * sa + m * sa + k * 2) = (_Float16)0.f;
    }
    lds_barrier();
    GIN_TICK(1);

    // neighbour table: slot q of tile row m resolved ONCE by one thread instead of by each of the row's 16-64 lanes in
    // the gather (the gather phase is vector-instruction bound: ~230 instructions per row pass, a quarter of them this
    // index arithmetic).  >= 0: tile-local row to add; -1: nothing (unused slot / self loop); <= -2: out-of-tile row -2 - v
    if (threadIdx.x < XBM * 4) {
        const int m = threadIdx.x >> 2, q = threadIdx.x & 3;
        const int row = m0 + m;
        const int e0 = rp[m], cnt = rp[m + 1] - e0;
        const int ci = e0 + q - e_lo;
        int v = -1;
        if (q < cnt && cols_in_lds && ci < HCOLS) {
            const int j = cs[ci], jl = j - m0;
            v = (jl >= 0 && jl < XBM) ? (j != row ? jl : -1) : -2 - j;
        }
        nb[threadIdx.x] = v;
    }
    lds_barrier();

    // ---- (1) gather from LDS (+ out-of-tile rows from L2, FB passes per round trip), scale, split, park --------
    constexpr int FBMAX = XBM == 64 ? 4 : 2;       // 16 VGPRs per pass in flight
    constexpr int FB = PASSES < FBMAX ? PASSES : FBMAX;
#pragma unroll
    for (int pg = 0; pg < PASSES; pg += FB) {
        float4 accv[FB];
        int far[FB][4];
#pragma unroll
        for (int pp = 0; pp < FB; ++pp) {
            const int m = ((pg + pp) * NW + wid) * RPW + sub;
            const int row = m0 + m;
            const int e0 = rp[m], cnt = rp[m + 1] - e0;
            const int lo4 = ok ? 4 * l : 0;
            accv[pp] = *reinterpret_cast<const float4 *>(Xs + m * A.K + lo4);
            accv[pp].x *= A.self_coef; accv[pp].y *= A.self_coef;
            accv[pp].z *= A.self_coef; accv[pp].w *= A.self_coef;
            const int4 nv = *reinterpret_cast<const int4 *>(nb + 4 * m);
            const int nq[4] = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool use = nq[q] >= 0;
                far[pp][q] = (nq[q] <= -2 && ok) ? -2 - nq[q] : -1;
                const float4 u = *reinterpret_cast<const float4 *>(Xs + (use ? nq[q] : m) * A.K + lo4);
                f4_fma(accv[pp], use ? 1.f : 0.f, u);
            }
        }
        // out-of-tile neighbours (the molecule that straddles the tile edge): most wavefronts have none in a group of
        // passes -- a wave-uniform test keeps their 16 zero-filled rows and 16 row additions out of the instruction stream
        bool has_far = false;
#pragma unroll
        for (int pp = 0; pp < FB; ++pp)
#pragma unroll
            for (int q = 0; q < 4; ++q) has_far |= far[pp][q] >= 0;
        if (__builtin_amdgcn_ballot_w64(has_far) != 0) {
            float4 fu[FB][4];
#pragma unroll
            for (int pp = 0; pp < FB; ++pp)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fu[pp][q] = f4_zero();
                    if (far[pp][q] >= 0) fu[pp][q] = load_x4(far[pp][q]);
                }
#pragma unroll
            for (int pp = 0; pp < FB; ++pp)
#pragma unroll
                for (int q = 0; q < 4; ++q) f4_add(accv[pp], fu[pp][q]);
        }
#pragma unroll
        for (int pp = 0; pp < FB; ++pp) {
            const int m = ((pg + pp) * NW + wid) * RPW + sub;
            const int row = m0 + m;
            if (ok) {
                const int e0 = rp[m], cnt = rp[m + 1] - e0;
                const int first = cols_in_lds ? 4 : 0;
                for (int e = e0 + first; e < e0 + cnt; ++e) {
                    const int jj = A.col[e];
                    if (jj != row) f4_add(accv[pp], load_x4(jj));
                }
            }
            // row maximum over the LG lanes of the row -> power-of-two scale
            float amax = ok ? fmaxf(fmaxf(fabsf(accv[pp].x), fabsf(accv[pp].y)), fmaxf(fabsf(accv[pp].z), fabsf(accv[pp].w))) : 0.f;
            amax = lanes_max<LG>(amax);
            float scale, inv;
            pow2_scale(amax, scale, inv);
            if (l == 0) ainv[m] = inv;
            if (ok) {
                const Split2 s0 = split2(accv[pp].x * scale), s1 = split2(accv[pp].y * scale),
                             s2 = split2(accv[pp].z * scale), s3 = split2(accv[pp].w * scale);
                f16x4 h1 = {s0.h1, s1.h1, s2.h1, s3.h1}, h2 = {s0.h2, s1.h2, s2.h2, s3.h2};
                char *dst = Ap + m * sa + 8 * l;
                *reinterpret_cast<f16x4 *>(dst) = h1;
                *reinterpret_cast<f16x4 *>(dst + XBM * sa) = h2;
            }
        }
    }
    lds_barrier();
    GIN_TICK(2);

    // ---- (2) hidden = act(agg @ W1 + b1): row maxima -> scale -> split -> park -----------------------------------
    f32x16 acc0[MT], acc1[MT];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[mt][r] = 0.f; acc1[mt][r] = 0.f; }
    };
    zero();
    float4 pb[4], psl[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = n_base + 8 * g + 4 * fh;             // four consecutive output columns
        pb[g] = A.b1 ? *reinterpret_cast<const float4 *>(A.b1 + c) : f4_zero();
        psl[g] = (ACT == BLGNN_ACT_PRELU && A.epi.act_param) ? *reinterpret_cast<const float4 *>(A.epi.act_param + c)
                                                            : f4_zero();
    }
    gemm_h3<MT, N>(acc0, acc1, Ap, sa, W1, A.Kp, lane_off, fi, fh, wfrag);
    GIN_TICK(9);
    h3_prefetch<N>(wfrag, W2, N, lane_off);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float rs = ainv[mt * 32 + fi] * w1inv;
        float hm = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v;
            v.x = act_t<ACT>(fmaf(acc0[mt][4 * g + 0] + acc1[mt][4 * g + 0], rs, pb[g].x), psl[g].x);
            v.y = act_t<ACT>(fmaf(acc0[mt][4 * g + 1] + acc1[mt][4 * g + 1], rs, pb[g].y), psl[g].y);
            v.z = act_t<ACT>(fmaf(acc0[mt][4 * g + 2] + acc1[mt][4 * g + 2], rs, pb[g].z), psl[g].z);
            v.w = act_t<ACT>(fmaf(acc0[mt][4 * g + 3] + acc1[mt][4 * g + 3], rs, pb[g].w), psl[g].w);
            acc0[mt][4 * g + 0] = v.x; acc0[mt][4 * g + 1] = v.y; acc0[mt][4 * g + 2] = v.z; acc0[mt][4 * g + 3] = v.w;
            hm = fmaxf(hm, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
        hm = fmaxf(hm, __shfl_xor(hm, 32, WAVE));           // the row's other 16 columns of this wave
        if (fh == 0) atomicMax(&hmax[mt * 32 + fi], __float_as_uint(hm));   // across the four column waves
    }
    lds_barrier();
    GIN_TICK(3);
    float hinv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
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
            *reinterpret_cast<f16x4 *>(dst + XBM * SH) = h2;
        }
    }
    lds_barrier();
    GIN_TICK(4);

    // ---- (3) y = affine(act(hidden @ W2 + b2)) -> LDS staging -> row stores ---------------------------------
    zero();
    float4 psc[4], psh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int c = n_base + 8 * g + 4 * fh;
        pb[g] = A.b2 ? *reinterpret_cast<const float4 *>(A.b2 + c) : f4_zero();
        psc[g] = float4{1.f, 1.f, 1.f, 1.f};
        psh[g] = f4_zero();
        if (A.epi.bn_scale) psc[g] = *reinterpret_cast<const float4 *>(A.epi.bn_scale + c);
        if (A.epi.bn_scale && A.epi.bn_shift) psh[g] = *reinterpret_cast<const float4 *>(A.epi.bn_shift + c);
    }
    gemm_h3<MT, N>(acc0, acc1, Hp, SH, W2, N, lane_off, fi, fh, wfrag);
    GIN_TICK(10);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float rs = hinv[mt] * w2inv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n_base + 8 * g + 4 * fh;
            float4 v;
            v.x = act_t<ACT>(fmaf(acc0[mt][4 * g + 0] + acc1[mt][4 * g + 0], rs, pb[g].x), psl[g].x);
            v.y = act_t<ACT>(fmaf(acc0[mt][4 * g + 1] + acc1[mt][4 * g + 1], rs, pb[g].y), psl[g].y);
            v.z = act_t<ACT>(fmaf(acc0[mt][4 * g + 2] + acc1[mt][4 * g + 2], rs, pb[g].z), psl[g].z);
            v.w = act_t<ACT>(fmaf(acc0[mt][4 * g + 3] + acc1[mt][4 * g + 3], rs, pb[g].w), psl[g].w);
            if (A.epi.bn_scale) {
                v.x = fmaf(v.x, psc[g].x, psh[g].x); v.y = fmaf(v.y, psc[g].y, psh[g].y);
                v.z = fmaf(v.z, psc[g].z, psh[g].z); v.w = fmaf(v.w, psc[g].w, psh[g].w);
            }
            *reinterpret_cast<float4 *>(Ys + (mt * 32 + fi) * LDO + c) = v;
        }
    }
    lds_barrier();
    GIN_TICK(5);
    for (int idx = threadIdx.x; idx < XBM * N4; idx += NTH) {
        const int m = idx / N4, c4 = idx % N4;
        const int row = m0 + m;
        const bool live = row < A.n;
        const float4 v = *reinterpret_cast<const float4 *>(Ys + m * LDO + 4 * c4);
        if (live) {
            float *dst = A.y + (int64_t)row * A.ldy + 4 * c4;
            if (vec_out) *reinterpret_cast<float4 *>(dst) = v;
            else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
        }
        if (A.y_amax || (N == 256 && A.ysp)) {
            // the N4 lanes that store one row sit in one wavefront (N4 = 32 or 64): their maximum is the row's
            float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            mx = lanes_max<N4>(mx);
            if (live && c4 == 0 && A.y_amax) A.y_amax[row] = mx;
            if (N == 256 && A.ysp && live) {
                // the row is one 256-column scale block: the GMN propagator's GEMMs read it from here
                float sc, inv;
                pow2_scale(mx, sc, inv);
                if (c4 == 0) A.yspe[(int64_t)row * A.ldyspe] = sp_exp_of(inv);
                const f32x4v sv = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
                sp_store4_pair(A.ysp + (int64_t)row * A.ldysp, 4 * c4, sv, threadIdx.x);   // (row: uniform per wavefront)
            }
        }
    }
    GIN_TICK(8);
    // ---- (4) fused readout: per-graph column sums of the staged tile, one atomic per (graph, column) and tile.
    // Rows are sorted by graph, so a tile touches few graphs (DrugBank: 27 atoms per molecule, ~2 per tile) and a
    // graph spans few tiles; with <= 2 contributions per graph the sum does not depend on their order.
    if (POOL) {
        // every thread: one column, RPT consecutive rows (all waves take part).  The staged values are requested at
        // once; graph boundaries come from one ballot, so the serial part is RPT adds and a few scalar branches.
        constexpr int RPT = XBM * N / NTH;                                   // 16 (N = 128) / 32 (N = 256)
        const int c = threadIdx.x % N;
        const int mb = __builtin_amdgcn_readfirstlane((threadIdx.x / N) * RPT);      // wave-uniform
        const int rows = min(A.n - m0 - mb, RPT);
        if (rows > 0) {
            float v[RPT];
#pragma unroll
            for (int m = 0; m < RPT; ++m) v[m] = Ys[(mb + m) * LDO + c];
            // bit m: row m starts another graph than row m - 1
            const int lr = lane & (RPT - 1);
            const int gl = rgraph[mb + lr], gp = rgraph[mb + (lr > 0 ? lr - 1 : 0)];
            const unsigned starts = (unsigned)__ballot(lr > 0 && lr < rows && gl != gp) & (RPT == 32 ? 0xffffffffu : ((1u << (RPT & 31)) - 1u));
            auto flush = [&](int first_row, float sum) {
                const int g = __builtin_amdgcn_readfirstlane(rgraph[mb + first_row]);
                const float w = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rweight[mb + first_row])));
                unsafeAtomicAdd(A.pool + (int64_t)g * A.ldp + c, sum * w);
            };
            int first = 0;
            float sum = 0.f;
#pragma unroll
            for (int m = 0; m < RPT; ++m) {
                if (m > 0 && ((starts >> m) & 1u)) { flush(first, sum); first = m; sum = 0.f; }
                if (m < rows) sum += v[m];
            }
            flush(first, sum);
        }
    }
}

static size_t h3_lds_bytes(int K, int Kp, int XBM, int N = 128) {
    size_t r0 = (size_t)2 * XBM * (Kp * 2 + 16);
    if ((size_t)XBM * (N + 4) * 4 > r0) r0 = (size_t)XBM * (N + 4) * 4;
    size_t r1 = (size_t)XBM * K * 4;
    if ((size_t)2 * XBM * (N * 2 + 16) > r1) r1 = (size_t)2 * XBM * (N * 2 + 16);
    return r0 + r1 + sizeof(int) * (XBM + 4 + HCOLS) + sizeof(float) * 2 * XBM + (sizeof(int) + sizeof(float)) * XBM +
           sizeof(int) * 4 * XBM;
}

template <int LG, int ACT, int XBM, bool CODES, int N, bool POOL>
static int launch_h3_inst(const GinH3Args &A, hipStream_t s) {
    static size_t declared = 0;
    const size_t bytes = h3_lds_bytes(A.K, A.Kp, XBM, N);
    if (bytes > declared && bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gin_layer_h3<LG, ACT, XBM, CODES, N, POOL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gin_layer_h3: cannot raise the dynamic LDS limit to %zu bytes", bytes);
        declared = bytes;
    }
    const int blocks = (A.n + XBM - 1) / XBM;
    hipLaunchKernelGGL((k_gin_layer_h3<LG, ACT, XBM, CODES, N, POOL>), dim3(blocks), dim3(N * 2), bytes, s, A);
    return launch_status("gin_layer_h3");
}

// the fused-readout epilogue is its own instantiation: the plain layer keeps its register allocation
template <int LG, int ACT, int XBM, bool CODES = false, int N = 128>
static int launch_h3_rows(const GinH3Args &A, hipStream_t s) {
    return A.pool ? launch_h3_inst<LG, ACT, XBM, CODES, N, true>(A, s) : launch_h3_inst<LG, ACT, XBM, CODES, N, false>(A, s);
}

template <int LG, int ACT>
static int launch_h3(const GinH3Args &A, hipStream_t s) {
    // 32-row workgroups, four per CU.  (64-row workgroups at two per CU -- every weight fragment
    // serving two row tiles -- measured the same on 471 k rows, 264 vs 259 us, and worse on graphs
    // that do not fill two rounds of them: 31.5 vs 24.3 us on 36 k rows.)
    return launch_h3_rows<LG, ACT, 32>(A, s);
}

// categorical input: 16 lanes x 4 columns cover the <= 64 one-hot columns of the first layer
template <int ACT>
static int launch_h3_codes(const GinH3Args &A, hipStream_t s) { return launch_h3_rows<16, ACT, 32, true>(A, s); }

static int h3_codes_pick_act(const GinH3Args &A, hipStream_t s) {
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    return launch_h3_codes<BLGNN_ACT_RELU>(A, s);
        case BLGNN_ACT_PRELU:   return launch_h3_codes<BLGNN_ACT_PRELU>(A, s);
        case BLGNN_ACT_SIGMOID: return launch_h3_codes<BLGNN_ACT_SIGMOID>(A, s);
        case BLGNN_ACT_TANH:    return launch_h3_codes<BLGNN_ACT_TANH>(A, s);
        default:                return launch_h3_codes<BLGNN_ACT_IDENTITY>(A, s);
    }
}

#ifdef GIN_OLD256       // diagnostics only (csrc/diag/gin_stamp.hip): round 3's one-workgroup-per-CU kernel
// N = 256: 64-row tiles, 512 threads
template <int LG>
static int h3w_pick_act(const GinH3Args &A, hipStream_t s) {
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    return launch_h3_rows<LG, BLGNN_ACT_RELU, 64, false, 256>(A, s);
        case BLGNN_ACT_PRELU:   return launch_h3_rows<LG, BLGNN_ACT_PRELU, 64, false, 256>(A, s);
        case BLGNN_ACT_SIGMOID: return launch_h3_rows<LG, BLGNN_ACT_SIGMOID, 64, false, 256>(A, s);
        case BLGNN_ACT_TANH:    return launch_h3_rows<LG, BLGNN_ACT_TANH, 64, false, 256>(A, s);
        default:                return launch_h3_rows<LG, BLGNN_ACT_IDENTITY, 64, false, 256>(A, s);
    }
}
#endif

template <int LG>
static int h3_pick_act(const GinH3Args &A, hipStream_t s) {
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    return launch_h3<LG, BLGNN_ACT_RELU>(A, s);
        case BLGNN_ACT_PRELU:   return launch_h3<LG, BLGNN_ACT_PRELU>(A, s);
        case BLGNN_ACT_SIGMOID: return launch_h3<LG, BLGNN_ACT_SIGMOID>(A, s);
        case BLGNN_ACT_TANH:    return launch_h3<LG, BLGNN_ACT_TANH>(A, s);
        default:                return launch_h3<LG, BLGNN_ACT_IDENTITY>(A, s);
    }
}

}  // namespace blgnn

#include "gin_h3q.hpp"       // N = 256: 64-row tiles of 256-thread workgroups, two per CU


using namespace blgnn;

extern "C" int64_t blgnn_packed_weight_h3_elems(int32_t K, int32_t N) {
    const int64_t Kp = ((int64_t)K + 31) / 32 * 32;
    return HHDR + 2 * Kp * N;                             // fp16 elements
}

extern "C" int blgnn_pack_weight_h3(const float *w, int64_t ldw, int32_t K, int32_t N, void *out, void *stream) {
    BLGNN_CHECK_ARG(w && out && K > 0 && N > 0 && ldw >= N, "pack_weight_h3: bad args");
    BLGNN_CHECK_ARG(aligned16(out), "pack_weight_h3: out must be 16-byte aligned");
    const int Kp = (K + 31) / 32 * 32;
    hipStream_t s = (hipStream_t)stream;
    // header word 1 is scratch for the absolute maximum (float bits), word 0 receives 1 / scale
    unsigned *amax = reinterpret_cast<unsigned *>(out) + 1;
    if (hipMemsetAsync(out, 0, HHDR * 2, s) != hipSuccess) return fail(BLGNN_ERR_LAUNCH, "pack_weight_h3: memset failed");
    const int64_t total = (int64_t)K * N;
    hipLaunchKernelGGL(k_absmax, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, s,
                       w, ldw, K, N, amax);
    const int64_t items = (int64_t)(Kp / 8) * N;
    hipLaunchKernelGGL(k_pack_h3, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, w, ldw, K, N, Kp, amax,
                       reinterpret_cast<_Float16 *>(out));
    return launch_status("pack_weight_h3");
}

extern "C" int blgnn_gin_layer_h3(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x,
                                  int64_t ldx, int32_t n, int32_t K, const void *w1_packed, const float *b1,
                                  const void *w2_packed, const float *b2, int32_t N, float eps, int32_t act,
                                  const float *act_param, const float *bn_scale, const float *bn_shift,
                                  float *y, int64_t ldy, const int32_t *row_graph, const int32_t *seg,
                                  float *pooled, int64_t ldp, int32_t pool_kind, float *y_row_amax, void *y_sp,
                                  int64_t ld_ysp, uint8_t *y_spe, int64_t ld_yspe, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && K > 0 && N > 0 && nnz >= 0, "gin_layer_h3: bad sizes");
    BLGNN_CHECK_ARG(!y_sp || (N == SP_BLOCK && y_spe && ld_ysp >= 2 * SP_BLOCK && ld_ysp % 8 == 0 && aligned16(y_sp) && ld_yspe >= 1),
                    "gin_layer_h3: sp output needs N == 256 and 16-byte aligned planes");
    BLGNN_CHECK_ARG(!pooled || (row_graph && seg && ldp >= N && (pool_kind == BLGNN_REDUCE_SUM || pool_kind == BLGNN_REDUCE_MEAN)),
                    "gin_layer_h3: fused readout needs row_graph, seg, ldp >= N and a reduce kind");
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(rowptr && (col || nnz == 0) && x && w1_packed && w2_packed && y, "gin_layer_h3: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldy >= N, "gin_layer_h3: leading dimension too small");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "gin_layer_h3: bad activation id");
    BLGNN_CHECK_ARG(x != y, "gin_layer_h3: x and y must not alias");
    if (!(N == 128 || N == 256) || K % 4 != 0 || K > N || ldx % 4 != 0 || !aligned16(x) || !aligned16(w1_packed) ||
        !aligned16(w2_packed))
        return fail(BLGNN_ERR_UNSUPPORTED, "gin_layer_h3: needs N in {128, 256}, K %% 4 == 0, K <= N and 16-byte aligned x / "
                    "packed weights (got N=%d K=%d); use blgnn_gin_layer_x6", N, K);
    GinH3Args A;
    A.rowptr = rowptr; A.col = col; A.x = x; A.ldx = ldx; A.n = n; A.K = K; A.Kp = (K + 31) / 32 * 32;
    A.w1p = reinterpret_cast<const _Float16 *>(w1_packed); A.w2p = reinterpret_cast<const _Float16 *>(w2_packed);
    A.b1 = b1; A.b2 = b2; A.self_coef = 1.f + eps;
    A.epi = Epilogue{b2, act_param, bn_scale, bn_shift, act};
    A.y = y; A.ldy = ldy;
    A.codes = nullptr; A.n_attr = 0;
    A.row_graph = row_graph; A.seg = seg; A.pool = pooled; A.ldp = ldp; A.pool_mean = pool_kind == BLGNN_REDUCE_MEAN;
    A.y_amax = y_row_amax;
    A.ysp = reinterpret_cast<_Float16 *>(y_sp); A.ldysp = ld_ysp; A.yspe = y_spe; A.ldyspe = ld_yspe;
    hipStream_t s = (hipStream_t)stream;
    const int k4 = K / 4;
    if (N == 256) {
#ifdef GIN_OLD256
        if (getenv("GIN_OLD256")) {
            if (k4 <= 16) return h3w_pick_act<16>(A, s);
            if (k4 <= 32) return h3w_pick_act<32>(A, s);
            return h3w_pick_act<64>(A, s);
        }
#endif
        return h3q_dispatch(A, s);
    }
    if (k4 <= 16) return h3_pick_act<16>(A, s);
    return h3_pick_act<32>(A, s);
}

extern "C" int blgnn_gin_layer_h3_codes(const int32_t *rowptr, const int32_t *col, int64_t nnz, const void *codes,
                                        const int32_t *code_offsets_host, int32_t num_attr, int32_t n,
                                        const void *w1_packed, const float *b1, const void *w2_packed,
                                        const float *b2, int32_t N, float eps, int32_t act, const float *act_param,
                                        const float *bn_scale, const float *bn_shift, float *y, int64_t ldy,
                                        const int32_t *row_graph, const int32_t *seg, float *pooled, int64_t ldp,
                                        int32_t pool_kind, float *y_row_amax, void *stream) {
    BLGNN_CHECK_ARG(!pooled || (row_graph && seg && ldp >= N && (pool_kind == BLGNN_REDUCE_SUM || pool_kind == BLGNN_REDUCE_MEAN)),
                    "gin_layer_h3_codes: fused readout needs row_graph, seg, ldp >= N and a reduce kind");
    BLGNN_CHECK_ARG(n >= 0 && N > 0 && nnz >= 0 && num_attr >= 1 && num_attr <= 8 && code_offsets_host,
                    "gin_layer_h3_codes: bad sizes");
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(rowptr && (col || nnz == 0) && codes && w1_packed && w2_packed && y, "gin_layer_h3_codes: null pointer");
    const int K = code_offsets_host[num_attr];
    BLGNN_CHECK_ARG(ldy >= N && act >= 0 && act <= BLGNN_ACT_TANH, "gin_layer_h3_codes: bad ldy / activation id");
    if (N != 128 || K <= 0 || K > 64 || K % 4 != 0 || (reinterpret_cast<uintptr_t>(codes) & 7u) || !aligned16(w1_packed) ||
        !aligned16(w2_packed))
        return fail(BLGNN_ERR_UNSUPPORTED, "gin_layer_h3_codes: needs N == 128, one-hot width %% 4 == 0 and <= 64, 8-byte "
                    "aligned codes (got N=%d width=%d)", N, K);
    GinH3Args A;
    A.rowptr = rowptr; A.col = col; A.x = nullptr; A.ldx = 0; A.n = n; A.K = K; A.Kp = (K + 31) / 32 * 32;
    A.w1p = reinterpret_cast<const _Float16 *>(w1_packed); A.w2p = reinterpret_cast<const _Float16 *>(w2_packed);
    A.b1 = b1; A.b2 = b2; A.self_coef = 1.f + eps;
    A.epi = Epilogue{b2, act_param, bn_scale, bn_shift, act};
    A.y = y; A.ldy = ldy;
    A.codes = reinterpret_cast<const uint2 *>(codes); A.n_attr = num_attr;
    A.row_graph = row_graph; A.seg = seg; A.pool = pooled; A.ldp = ldp; A.pool_mean = pool_kind == BLGNN_REDUCE_MEAN;
    A.y_amax = y_row_amax;
    A.ysp = nullptr; A.ldysp = 0; A.yspe = nullptr; A.ldyspe = 0;
    for (int a = 0; a < 8; ++a) {
        const int o = a < num_attr ? code_offsets_host[a] : 0;
        BLGNN_CHECK_ARG(o >= 0 && o < 256, "gin_layer_h3_codes: attribute offset out of range");
        A.code_off[a] = (uint8_t)o;
    }
    return h3_codes_pick_act(A, (hipStream_t)stream);
}
