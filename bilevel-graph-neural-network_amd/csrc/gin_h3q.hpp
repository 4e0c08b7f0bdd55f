// The fused f16x3 GIN layer at N = 256 (model/layers.py:26-31,51-57 at the d = 256 configs), TWO workgroups per CU.
//
// Round 3's N = 256 kernel (k_gin_layer_h3<.., 64, .., 256>, removed) ran ONE 512-thread workgroup per CU: 135 KB of LDS
// (staged fp32 rows and the aggregated fp16 pieces side by side, then hidden pieces and the fp32 output tile) and 203
// VGPRs.  A 64-row tile's phases -- rows + CSR slice staged (memory latency), gather + split (vector instructions),
// GEMM1 (matrix cores + 256 KB of weight fragments out of L2), activation + row maxima, split, GEMM2, row stores -- ran
// strictly one after the other, 26 us per tile, 0.15 of the HBM floor / 0.19 of the MFMA floor of the launch.
//
// Here a tile belongs to a 256-thread workgroup (one wavefront per SIMD) that needs 77 KB of LDS, so two tiles share a
// CU and the hardware interleaves one tile's memory / vector phases with the other's matrix phases:
//   * ONE LDS region is reused by every stage of the tile -- fp32 input rows -> aggregated pieces -> hidden pieces ->
//     fp32 output tile -- with the gathered rows held in registers across the barrier that retires the input rows;
//   * a wavefront owns 64 output columns x all 64 rows (2 x 2 MFMA blocks, two accumulators per block as before:
//     h1.h1 apart from the two cross terms), its weight fragments come straight from L2 into a ring of 16-deep K steps;
//     the weight traffic per row is the old kernel's (every fragment serves both 32-row halves);
//   * per-channel parameters (b1, b2, eval-BN scale / shift, PReLU slopes) are staged once into LDS: the epilogues
//     fetch them as broadcast reads instead of holding 96 registers across the GEMMs;
//   * the gather handles 8 columns per lane at K > 128 (two rows per wavefront pass), halving its index arithmetic.
// Results are bit for bit those of the one-workgroup kernel (same products, same accumulation order).
#pragma once

namespace blgnn {

constexpr int Q_N = 256, Q_BM = 64, Q_NW = 4, Q_NTH = 256;
constexpr int Q_SH = Q_N * 2 + 16;                      // piece row stride at K = 256 (bytes)
constexpr int Q_LDO = Q_N + 4;                          // fp32 output tile row stride (floats)
constexpr int Q_REGION = 2 * Q_BM * Q_SH;               // 67 584 B >= 64 x 260 x 4 (output tile), 64 x 256 x 4 (input rows)
constexpr int Q_RING = 4;                               // weight ring: 16-deep K steps, Q_RING - 1 requested ahead
constexpr int Q_PRM = 5;                                // b1, b2, bn scale, bn shift, PReLU slope

static size_t h3q_lds_bytes() {
    return (size_t)Q_REGION + sizeof(int) * (Q_BM + 4 + HCOLS) + sizeof(float) * 4 * Q_BM + sizeof(int) * 4 * Q_BM +
           sizeof(float) * Q_PRM * Q_N;
}

// weight fragments of one 16-deep K step: [column block][piece]; lane (fi, fh) holds the 8 consecutive k
// (k = 16 step + 8 fh ..) of output column n_base + 32 nb + fi
__device__ __forceinline__ void q_load_w(f16x8 (&dst)[2][2], hrsrc_t W, int Kp, int step, unsigned lane_off) {
    const unsigned piece = (unsigned)Kp * Q_N * 2u;                 // bytes
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned soff = HHDR * 2u + (unsigned)q * piece + (unsigned)(2 * step) * (Q_N * 16u);
            const u32x4h v = __builtin_amdgcn_raw_buffer_load_b128(W, lane_off + (unsigned)nb * 512u, soff, 0);
            dst[nb][q] = __builtin_bit_cast(f16x8, v);
        }
}

__device__ __forceinline__ void q_prefetch(f16x8 (&w)[Q_RING][2][2], hrsrc_t W, int Kp, unsigned lane_off) {
#pragma unroll
    for (int u = 0; u < Q_RING - 1; ++u)
        if (u * 16 < Kp) q_load_w(w[u], W, Kp, u, lane_off);
}

// acc0 += w1.a1 ; acc1 += w1.a2 + w2.a1 over the 64 rows x 64 columns of the wavefront (transposed GEMM: the lane ends up
// with n = n_base + 32 nb + 8 (r >> 2) + 4 fh + (r & 3), m = 32 mt + fi)
__device__ __forceinline__ void q_gemm(f32x16 (&acc0)[2][2], f32x16 (&acc1)[2][2], const char *Ap, int sa, hrsrc_t W,
                                       int Kp, unsigned lane_off, int fi, int fh, f16x8 (&w)[Q_RING][2][2]) {
    const int piece_bytes = Q_BM * sa;
    const char *arow = Ap + fi * sa + fh * 16;                   // lane (row fi, k half fh)
    const int steps = Kp / 16;
    for (int s0 = 0; s0 < steps; s0 += Q_RING) {
#pragma unroll
        for (int u = 0; u < Q_RING; ++u) {
            const int s = s0 + u;
            if (s < steps) {
                if (s + Q_RING - 1 < steps) q_load_w(w[(u + Q_RING - 1) % Q_RING], W, Kp, s + Q_RING - 1, lane_off);
                f16x8 a1[2], a2[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    a1[mt] = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + s * 32);
                    a2[mt] = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + piece_bytes + s * 32);
                }
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc0[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][nb][0], a1[mt], acc0[mt][nb], 0, 0, 0);
                        acc1[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][nb][0], a2[mt], acc1[mt][nb], 0, 0, 0);
                        acc1[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[u][nb][1], a1[mt], acc1[mt][nb], 0, 0, 0);
                    }
            }
        }
    }
}

// maximum over the LG lanes of a row; every lane gets it.  LG = 32: the two half-wave results travel through scalars
// (v_readlane) instead of the LDS crossbar (ds_bpermute)
template <int LG>
__device__ __forceinline__ float q_row_max(float v, int lane) {
    if (LG == 32) {
        v = lanes_max<16>(v);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
        return lane < 32 ? fmaxf(r0, r1) : fmaxf(r2, r3);
    }
    return lanes_max<LG>(v);
}

// LG lanes per row in the gather, V float4 per lane (columns 4 (l + LG v) ..): K <= 4 LG V
template <int LG, int V, int ACT, bool POOL>
__global__ void __launch_bounds__(Q_NTH, 2) k_gin_layer_h3q(const GinH3Args A) {
    constexpr int N = Q_N, XBM = Q_BM, NW = Q_NW, NTH = Q_NTH, LDO = Q_LDO, SH = Q_SH;
    constexpr int RPW = WAVE / LG;                     // rows per wavefront pass
    constexpr int PASSES = XBM / (NW * RPW);
    constexpr int NCV = (HCOLS + NTH - 1) / NTH;
    extern __shared__ float4 lds4[];
    char *lds = reinterpret_cast<char *>(lds4);
    GIN_TICK(0);
    const int sa = A.Kp * 2 + 16;                   // aggregated piece row stride (bytes)
    // the one region: fp32 input rows [XBM][K] -> aggregated pieces [2][XBM][sa] -> hidden pieces [2][XBM][SH] -> fp32
    // output tile [XBM][LDO]
    float *Xs = reinterpret_cast<float *>(lds);
    char *Ap = lds, *Hp = lds;
    float *Ys = reinterpret_cast<float *>(lds);
    int *rp = reinterpret_cast<int *>(lds + Q_REGION);      // [XBM + 4]
    int *cs = rp + XBM + 4;                                 // [HCOLS]
    float *ainv = reinterpret_cast<float *>(cs + HCOLS);    // [XBM] 1 / row scale of the aggregate
    unsigned *hmax = reinterpret_cast<unsigned *>(ainv + XBM);   // [XBM] row maxima of the hidden tile (float bits)
    int *rgraph = reinterpret_cast<int *>(hmax + XBM);           // [XBM] graph of every tile row (fused readout)
    float *rweight = reinterpret_cast<float *>(rgraph + XBM);    // [XBM] readout weight of every tile row
    int *nb = reinterpret_cast<int *>(rweight + XBM);            // [XBM][4] first four neighbours of every tile row, resolved
    float *prm = reinterpret_cast<float *>(nb + 4 * XBM);        // [Q_PRM][N] per-channel parameters
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int sub = lane / LG, l = lane % LG;
    const int n_base = wid * 64;
    const hrsrc_t W1 = h3_rsrc(A.w1p, A.Kp, N), W2 = h3_rsrc(A.w2p, N, N);
    const unsigned lane_off = ((unsigned)fh * N + (unsigned)(n_base + fi)) * 16u;
    const int k4n = A.K >> 2, k4p = A.Kp >> 2;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;
    const int m0 = blockIdx.x * XBM;
    const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);
    bool ok[V];
    int lo4[V];                                              // first column of the lane's v-th float4 (0 where it has none)
#pragma unroll
    for (int v = 0; v < V; ++v) { ok[v] = l + LG * v < k4n; lo4[v] = ok[v] ? 4 * (l + LG * v) : 0; }

    // ---- (0) requests: rowptr slice, feature rows, edge range -> column indices, per-channel parameters ------------
    int rpv = 0;
    if (threadIdx.x <= XBM) {
        const int r = m0 + threadIdx.x;
        rpv = A.rowptr[r < A.n ? r : A.n];
    }
    {
        float4 xin[PASSES][V];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int row = m0 + (p * NW + wid) * RPW + sub;
#pragma unroll
            for (int v = 0; v < V; ++v) xin[p][v] = (ok[v] && row < A.n) ? X[(int64_t)row * ldx4 + (lo4[v] >> 2)] : f4_zero();
        }
        const int e_lo0 = A.rowptr[m0];                                  // uniform addresses: scalar loads
        const int ncol0 = A.rowptr[m0 + XBM < A.n ? m0 + XBM : A.n] - e_lo0;
        int cv[NCV];
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            const int idx = threadIdx.x + NTH * i;
            cv[i] = (ncol0 <= HCOLS && idx < ncol0) ? A.col[e_lo0 + idx] : 0;
        }
        float pv[Q_PRM];
        {
            const int c = threadIdx.x;                                   // NTH == N: one channel per thread
            pv[0] = A.b1 ? A.b1[c] : 0.f;
            pv[1] = A.b2 ? A.b2[c] : 0.f;
            pv[2] = A.epi.bn_scale ? A.epi.bn_scale[c] : 1.f;
            pv[3] = (A.epi.bn_scale && A.epi.bn_shift) ? A.epi.bn_shift[c] : 0.f;
            pv[4] = (ACT == BLGNN_ACT_PRELU && A.epi.act_param) ? A.epi.act_param[c] : 0.f;
        }
        if (threadIdx.x <= XBM) rp[threadIdx.x] = rpv;
        if (threadIdx.x < XBM) hmax[threadIdx.x] = 0u;
        if (POOL && threadIdx.x >= NTH - XBM) {          // the last XBM threads
            const int m = threadIdx.x - (NTH - XBM);
            const int r = m0 + m;
            const int g = A.row_graph[r < A.n ? r : A.n - 1];
            rgraph[m] = g;
            rweight[m] = A.pool_mean ? 1.f / (float)max(A.seg[g + 1] - A.seg[g], 1) : 1.f;
        }
#pragma unroll
        for (int i = 0; i < Q_PRM; ++i) prm[i * N + threadIdx.x] = pv[i];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * NW + wid) * RPW + sub;
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (ok[v]) *reinterpret_cast<float4 *>(Xs + m * A.K + lo4[v]) = xin[p][v];
        }
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            const int idx = threadIdx.x + NTH * i;
            if (ncol0 <= HCOLS && idx < ncol0) cs[idx] = cv[i];
        }
    }
    const int e_lo = A.rowptr[m0];
    const int ncol_all = A.rowptr[m0 + XBM < A.n ? m0 + XBM : A.n] - e_lo;
    const bool cols_in_lds = ncol_all <= HCOLS;
    lds_barrier();
    GIN_TICK(1);

    // neighbour table: slot q of tile row m resolved ONCE.  >= 0: tile-local row to add; -1: nothing (unused slot /
    // self loop); <= -2: out-of-tile row -2 - v
    {
        const int m = threadIdx.x >> 2, q = threadIdx.x & 3;      // NTH == 4 XBM
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

    // ---- (1) gather from LDS (+ out-of-tile rows from L2) into registers ---------------------------------------------
    float4 accv[PASSES][V];
    constexpr int FB = PASSES < 2 ? PASSES : 2;        // passes per out-of-tile round trip
#pragma unroll
    for (int pg = 0; pg < PASSES; pg += FB) {
        int far[FB][4];
#pragma unroll
        for (int pp = 0; pp < FB; ++pp) {
            const int p = pg + pp;
            const int m = (p * NW + wid) * RPW + sub;
            const int4 nv = *reinterpret_cast<const int4 *>(nb + 4 * m);
            const int nq[4] = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
            for (int v = 0; v < V; ++v) {
                accv[p][v] = *reinterpret_cast<const float4 *>(Xs + m * A.K + lo4[v]);
                accv[p][v].x *= A.self_coef; accv[p][v].y *= A.self_coef;
                accv[p][v].z *= A.self_coef; accv[p][v].w *= A.self_coef;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool use = nq[q] >= 0;
                far[pp][q] = nq[q] <= -2 ? -2 - nq[q] : -1;
                const float *src = Xs + (use ? nq[q] : m) * A.K;
                const float wq = use ? 1.f : 0.f;
#pragma unroll
                for (int v = 0; v < V; ++v) f4_fma(accv[p][v], wq, *reinterpret_cast<const float4 *>(src + lo4[v]));
            }
        }
        // out-of-tile neighbours (the molecule that straddles the tile edge): most wavefronts have none in a group of
        // passes -- a wave-uniform test keeps their loads and additions out of the instruction stream
        bool has_far = false;
#pragma unroll
        for (int pp = 0; pp < FB; ++pp)
#pragma unroll
            for (int q = 0; q < 4; ++q) has_far |= far[pp][q] >= 0;
        if (__builtin_amdgcn_ballot_w64(has_far) != 0) {
            float4 fu[FB][4][V];
#pragma unroll
            for (int pp = 0; pp < FB; ++pp)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        fu[pp][q][v] = f4_zero();
                        if (far[pp][q] >= 0 && ok[v]) fu[pp][q][v] = X[(int64_t)far[pp][q] * ldx4 + (lo4[v] >> 2)];
                    }
#pragma unroll
            for (int pp = 0; pp < FB; ++pp)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v) f4_add(accv[pg + pp][v], fu[pp][q][v]);
        }
        // rows of degree > 4 (and every neighbour of a tile whose column slice does not fit LDS): from global memory
#pragma unroll
        for (int pp = 0; pp < FB; ++pp) {
            const int p = pg + pp;
            const int m = (p * NW + wid) * RPW + sub;
            const int row = m0 + m;
            const int e0 = rp[m], cnt = rp[m + 1] - e0;
            const int first = cols_in_lds ? 4 : 0;
            for (int e = e0 + first; e < e0 + cnt; ++e) {
                const int jj = A.col[e];
                if (jj != row) {
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (ok[v]) f4_add(accv[p][v], X[(int64_t)jj * ldx4 + (lo4[v] >> 2)]);
                }
            }
        }
    }
    f16x8 wfrag[Q_RING][2][2];
    q_prefetch(wfrag, W1, A.Kp, lane_off);
    lds_barrier();                                  // every wavefront is done with the fp32 input rows
    GIN_TICK(2);

    // ---- (1b) row maximum -> power-of-two scale -> two fp16 pieces, parked over the input rows ------------------------
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * NW + wid) * RPW + sub;
        float amax = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (!ok[v]) accv[p][v] = f4_zero();
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(accv[p][v].x), fabsf(accv[p][v].y)), fmaxf(fabsf(accv[p][v].z), fabsf(accv[p][v].w))));
        }
        amax = q_row_max<LG>(amax, lane);
        float scale, inv;
        pow2_scale(amax, scale, inv);
        if (l == 0) ainv[m] = inv;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (l + LG * v < k4p) {                         // the K padding columns receive zeros
                const f32x4v sv = {accv[p][v].x * scale, accv[p][v].y * scale, accv[p][v].z * scale, accv[p][v].w * scale};
                const f16x4 h1 = __builtin_convertvector(sv, f16x4);
                const f16x4 h2 = __builtin_convertvector(sv - __builtin_convertvector(h1, f32x4v), f16x4);
                char *dst = Ap + m * sa + 8 * (l + LG * v);
                *reinterpret_cast<f16x4 *>(dst) = h1;
                *reinterpret_cast<f16x4 *>(dst + XBM * sa) = h2;
            }
        }
    }
    lds_barrier();
    GIN_TICK(3);

    // ---- (2) hidden = act(agg @ W1 + b1): row maxima -> scale -> split -> parked over the aggregated pieces ----------
    const float w1inv = *reinterpret_cast<const float *>(A.w1p), w2inv = *reinterpret_cast<const float *>(A.w2p);
    f32x16 acc0[2][2], acc1[2][2];
    auto zero = [&]() {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[mt][nbk][r] = 0.f; acc1[mt][nbk][r] = 0.f; }
    };
    zero();
    q_gemm(acc0, acc1, Ap, sa, W1, A.Kp, lane_off, fi, fh, wfrag);
    GIN_TICK(4);
    q_prefetch(wfrag, W2, N, lane_off);
    {
        float hm[2] = {0.f, 0.f};
#pragma unroll
        for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n_base + 32 * nbk + 8 * g + 4 * fh;             // four consecutive output columns
                const float4 pb = *reinterpret_cast<const float4 *>(prm + c);
                const float4 psl = ACT == BLGNN_ACT_PRELU ? *reinterpret_cast<const float4 *>(prm + 4 * N + c) : f4_zero();
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const float rs = ainv[mt * 32 + fi] * w1inv;
                    float4 v;
                    v.x = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 0] + acc1[mt][nbk][4 * g + 0], rs, pb.x), psl.x);
                    v.y = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 1] + acc1[mt][nbk][4 * g + 1], rs, pb.y), psl.y);
                    v.z = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 2] + acc1[mt][nbk][4 * g + 2], rs, pb.z), psl.z);
                    v.w = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 3] + acc1[mt][nbk][4 * g + 3], rs, pb.w), psl.w);
                    acc0[mt][nbk][4 * g + 0] = v.x; acc0[mt][nbk][4 * g + 1] = v.y;
                    acc0[mt][nbk][4 * g + 2] = v.z; acc0[mt][nbk][4 * g + 3] = v.w;
                    hm[mt] = fmaxf(hm[mt], fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                }
            }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) atomicMax(&hmax[mt * 32 + fi], __float_as_uint(hm[mt]));   // both column halves, four waves
    }
    lds_barrier();                                  // row maxima complete; every wavefront is done with the aggregated pieces
    GIN_TICK(5);
    float hinv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float scale;
        pow2_scale(__uint_as_float(hmax[mt * 32 + fi]), scale, hinv[mt]);
#pragma unroll
        for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n_base + 32 * nbk + 8 * g + 4 * fh;
                const f32x4v sv = {acc0[mt][nbk][4 * g + 0] * scale, acc0[mt][nbk][4 * g + 1] * scale,
                                   acc0[mt][nbk][4 * g + 2] * scale, acc0[mt][nbk][4 * g + 3] * scale};
                const f16x4 h1 = __builtin_convertvector(sv, f16x4);
                const f16x4 h2 = __builtin_convertvector(sv - __builtin_convertvector(h1, f32x4v), f16x4);
                char *dst = Hp + (mt * 32 + fi) * SH + c * 2;
                *reinterpret_cast<f16x4 *>(dst) = h1;
                *reinterpret_cast<f16x4 *>(dst + XBM * SH) = h2;
            }
    }
    lds_barrier();
    GIN_TICK(6);

    // ---- (3) y = affine(act(hidden @ W2 + b2)) -> fp32 tile over the hidden pieces -> row stores ---------------------
    zero();
    q_gemm(acc0, acc1, Hp, SH, W2, N, lane_off, fi, fh, wfrag);
    GIN_TICK(7);
    lds_barrier();                                  // every wavefront is done with the hidden pieces
#pragma unroll
    for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n_base + 32 * nbk + 8 * g + 4 * fh;
            const float4 pb = *reinterpret_cast<const float4 *>(prm + N + c);
            const float4 psc = *reinterpret_cast<const float4 *>(prm + 2 * N + c);
            const float4 psh = *reinterpret_cast<const float4 *>(prm + 3 * N + c);
            const float4 psl = ACT == BLGNN_ACT_PRELU ? *reinterpret_cast<const float4 *>(prm + 4 * N + c) : f4_zero();
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float rs = hinv[mt] * w2inv;
                float4 v;
                v.x = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 0] + acc1[mt][nbk][4 * g + 0], rs, pb.x), psl.x);
                v.y = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 1] + acc1[mt][nbk][4 * g + 1], rs, pb.y), psl.y);
                v.z = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 2] + acc1[mt][nbk][4 * g + 2], rs, pb.z), psl.z);
                v.w = act_t<ACT>(fmaf(acc0[mt][nbk][4 * g + 3] + acc1[mt][nbk][4 * g + 3], rs, pb.w), psl.w);
                if (A.epi.bn_scale) {
                    v.x = fmaf(v.x, psc.x, psh.x); v.y = fmaf(v.y, psc.y, psh.y);
                    v.z = fmaf(v.z, psc.z, psh.z); v.w = fmaf(v.w, psc.w, psh.w);
                }
                *reinterpret_cast<float4 *>(Ys + (mt * 32 + fi) * LDO + c) = v;
            }
        }
    lds_barrier();
    GIN_TICK(8);
    // a wavefront stores whole rows: 64 lanes x 16 B = the row's 1 KB
#pragma unroll 4
    for (int m = wid; m < XBM; m += NW) {
        const int row = m0 + m;
        const bool live = row < A.n;                 // wave-uniform
        const float4 v = *reinterpret_cast<const float4 *>(Ys + m * LDO + 4 * lane);
        if (live) {
            float *dst = A.y + (int64_t)row * A.ldy + 4 * lane;
            if (vec_out) *reinterpret_cast<float4 *>(dst) = v;
            else { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
        }
        if (A.y_amax || A.ysp) {
            float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            mx = lanes_max<64>(mx);
            if (live && lane == 0 && A.y_amax) A.y_amax[row] = mx;
            if (A.ysp && live) {
                // the row is one 256-column scale block: the GMN propagator's GEMMs read it from here
                float sc, inv;
                pow2_scale(mx, sc, inv);
                if (lane == 0) A.yspe[(int64_t)row * A.ldyspe] = sp_exp_of(inv);
                const f32x4v sv = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
                sp_store4_pair(A.ysp + (int64_t)row * A.ldysp, 4 * lane, sv, lane);
            }
        }
    }
    GIN_TICK(9);
    // ---- (4) fused readout: per-graph column sums of the staged tile, one atomic per (graph, column) and tile.
    // Rows are sorted by graph; a thread owns one column and walks the tile's rows in chunks of 16 staged values.
    if (POOL) {
        const int c = threadIdx.x;                                            // NTH == N
        const int rows = min(A.n - m0, XBM);
        const int gl = rgraph[lane], gp = rgraph[lane > 0 ? lane - 1 : 0];
        // bit m: row m starts another graph than row m - 1
        const unsigned long long starts = __ballot(lane > 0 && lane < rows && gl != gp);
        auto flush = [&](int first_row, float sum) {
            const int g = __builtin_amdgcn_readfirstlane(rgraph[first_row]);
            const float w = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rweight[first_row])));
            unsafeAtomicAdd(A.pool + (int64_t)g * A.ldp + c, sum * w);
        };
        int first = 0;
        float sum = 0.f;
        for (int mb = 0; mb < XBM && mb < rows; mb += 16) {
            float v[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = Ys[(mb + m) * LDO + c];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int mm = mb + m;
                if (mm > 0 && ((starts >> mm) & 1ull)) { flush(first, sum); first = mm; sum = 0.f; }
                if (mm < rows) sum += v[m];
            }
        }
        flush(first, sum);
    }
}

template <int LG, int V, int ACT, bool POOL>
static int launch_h3q_inst(const GinH3Args &A, hipStream_t s) {
    static bool declared = false;
    const size_t bytes = h3q_lds_bytes();
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gin_layer_h3q<LG, V, ACT, POOL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gin_layer_h3: cannot raise the dynamic LDS limit to %zu bytes", bytes);
        declared = true;
    }
    const int blocks = (A.n + Q_BM - 1) / Q_BM;
    hipLaunchKernelGGL((k_gin_layer_h3q<LG, V, ACT, POOL>), dim3(blocks), dim3(Q_NTH), bytes, s, A);
    return launch_status("gin_layer_h3");
}

template <int LG, int V, int ACT>
static int launch_h3q(const GinH3Args &A, hipStream_t s) {
    return A.pool ? launch_h3q_inst<LG, V, ACT, true>(A, s) : launch_h3q_inst<LG, V, ACT, false>(A, s);
}

template <int LG, int V>
static int h3q_pick_act(const GinH3Args &A, hipStream_t s) {
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    return launch_h3q<LG, V, BLGNN_ACT_RELU>(A, s);
        case BLGNN_ACT_PRELU:   return launch_h3q<LG, V, BLGNN_ACT_PRELU>(A, s);
        case BLGNN_ACT_SIGMOID: return launch_h3q<LG, V, BLGNN_ACT_SIGMOID>(A, s);
        case BLGNN_ACT_TANH:    return launch_h3q<LG, V, BLGNN_ACT_TANH>(A, s);
        default:                return launch_h3q<LG, V, BLGNN_ACT_IDENTITY>(A, s);
    }
}

}  // namespace blgnn
