// The fused f16x3 GIN layer at N = 256 (model/layers.py:26-31,51-57 at the d = 256 configs), TWO workgroups per CU.
//
// Round 3's N = 256 kernel (k_gin_layer_h3<.., 64, .., 256>; kept behind GIN_OLD256 for csrc/diag/gin_stamp.hip only) ran
// ONE 512-thread workgroup per CU: 135 KB of LDS (staged fp32 rows and the aggregated fp16 pieces side by side, then
// hidden pieces and the fp32 output tile) and 203 VGPRs.  A 64-row tile's phases -- rows + CSR slice staged (memory
// latency), gather + split (vector instructions), GEMM1 (matrix cores + 256 KB of weight fragments out of L2),
// activation + row maxima, split, GEMM2, row stores -- ran strictly one after the other, 26 us per tile, 0.15 of the HBM
// floor / 0.19 of the MFMA floor of the launch.
//
// Here a tile belongs to a 256-thread workgroup (one wavefront per SIMD) that needs 78 KB of LDS, so two tiles share a
// CU and the hardware interleaves one tile's memory / vector phases with the other's matrix phases (per-workgroup
// timelines of a launch, csrc/diag/gin_stamp.hip: the two tiles of a CU drift out of phase by themselves; a forced
// half-tile stagger of the odd wave slot and s_setprio inside the GEMMs measured no difference and were removed):
//   * ONE LDS region is reused by every stage of the tile -- fp32 input rows -> aggregated pieces -> hidden pieces ->
//     fp32 output tile -- with the gathered rows held in registers across the barrier that retires the input rows;
//   * a wavefront owns 64 output columns x all 64 rows (2 x 2 MFMA blocks, two accumulators per block as before:
//     h1.h1 apart from the two cross terms); the weight traffic per row is the old kernel's (every fragment serves both
//     32-row halves).  Its weight fragments come straight from L2 as ONE stream over both GEMMs (QStream below): three
//     16-deep K steps are always in flight, no request sits behind a branch, and the order request -> A reads -> MFMAs
//     of a step is pinned;
//   * per-channel parameters (b1, b2, eval-BN scale / shift, PReLU slopes) are staged once into LDS: the epilogues
//     fetch them as broadcast reads instead of holding 96 registers across the GEMMs;
//   * the gather adds FOUR neighbour rows to every row unconditionally: the neighbour table holds LDS byte offsets and
//     the slots a row does not use point at a row of zeros behind the tile (x + 0 is exact), so a slot costs one address
//     add, the reads and the adds -- no compare / select per slot; out-of-tile neighbours (the molecules cut by the
//     tile's edges) are fetched from L2 up front for the first and last pass (where the edges are) and added at the end;
//   * rows come in and go out through tile-local buffer resources (rows past n read as zeros / are dropped by the range
//     check; no 64-bit address arithmetic, no row predicates).
// Results are bit for bit those of the one-workgroup kernel (same products, same accumulation order).
#pragma once

namespace blgnn {

constexpr int Q_N = 256, Q_BM = 64, Q_NW = 4, Q_NTH = 256;
constexpr int Q_SH = Q_N * 2 + 16;                      // piece row stride at K = 256 (bytes)
constexpr int Q_LDO = Q_N + 4;                          // fp32 output tile row stride (floats)
constexpr int Q_REGION = 2 * Q_BM * Q_SH;               // 67 584 B >= 64 x 260 x 4 (output tile), 65 x 256 x 4 (input rows + zero row)
constexpr int Q_RING = 4;                               // weight ring: 16-deep K steps, Q_RING - 1 requested ahead
constexpr int Q_PRM = 5;                                // b1, b2, bn scale, bn shift, PReLU slope

static size_t h3q_lds_bytes() {
    // region + rp[68] + cs[HCOLS] + ainv / hmax / rgraph / rweight / nfar [64] + nb, nbo [64][4] + prm
    return (size_t)Q_REGION + sizeof(int) * (Q_BM + 4 + HCOLS) + sizeof(float) * 5 * Q_BM + sizeof(int) * 8 * Q_BM +
           sizeof(float) * Q_PRM * Q_N;
}

// The order of a K step -- requests for step s + 3, A reads for step s + 1, MFMAs of step s -- is pinned: left to itself
// the scheduler sinks the sixteen requests of four unrolled steps to the end of the loop body (shorter live ranges at 256
// VGPRs), i.e. to just in front of their first use.
#ifndef Q_NO_PIN
#define Q_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define Q_PIN() do { } while (0)
#endif

// weight fragments of one 16-deep K step: [column block][piece]; lane (fi, fh) holds the 8 consecutive k
// (k = 16 step + 8 fh ..) of output column n_base + 32 nb + fi.  `piece` = bytes of one piece of the packed image.
__device__ __forceinline__ void q_load_w(f16x8 (&dst)[2][2], hrsrc_t W, unsigned piece, int step, unsigned lane_off) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned soff = HHDR * 2u + (unsigned)q * piece + (unsigned)(2 * step) * (Q_N * 16u);
            const u32x4h v = __builtin_amdgcn_raw_buffer_load_b128(W, lane_off + (unsigned)nb * 512u, soff, 0);
            dst[nb][q] = __builtin_bit_cast(f16x8, v);
        }
}

// The two GEMMs of a tile read ONE weight stream: the steps1 16-deep K steps of W1 (steps1 % 4 == 0: K padded to 64 with
// zero activation columns; the fragments fetched for the padding are whatever finite fp16 the packed image holds there,
// or the zeros a buffer load returns past its end), then the 16 steps of W2.  Q_RING - 1 steps are always in flight --
// the last steps of GEMM1 request the first of GEMM2, across the epilogue between them -- and NO request sits behind a
// branch: hipcc's wait-count pass gives up on loads issued in conditional blocks and waits for all but the last few
// requests in front of every K step (vmcnt(3) with sixteen in flight: the first version of this kernel ran 1 000 cycles
// per step for 384 cycles of MFMA).
struct QStream {
    hrsrc_t W1, W2;
    unsigned piece1, piece2;      // bytes per piece
    int steps1;
};

__device__ __forceinline__ void q_request(f16x8 (&dst)[2][2], const QStream &S, int g, unsigned lane_off) {
    const bool first = g < S.steps1;                               // wave-uniform: scalar selects
    q_load_w(dst, first ? S.W1 : S.W2, first ? S.piece1 : S.piece2, first ? g : g - S.steps1, lane_off);
}

// A fragments of K step s: [row half][piece]
__device__ __forceinline__ void q_read_a(f16x8 (&a)[2][2], const char *arow, int sa, int piece_bytes, int s) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        a[mt][0] = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + s * 32);
        a[mt][1] = *reinterpret_cast<const f16x8 *>(arow + mt * 32 * sa + piece_bytes + s * 32);
    }
}

// acc0 += w1.a1 ; acc1 += w1.a2 + w2.a1 over the 64 rows x 64 columns of the wavefront (transposed GEMM: the lane ends up
// with n = n_base + 32 nb + 8 (r >> 2) + 4 fh + (r & 3), m = 32 mt + fi).  FIRST: the accumulators start from the
// MFMA's zero operand instead of 128 v_mov
template <bool FIRST>
__device__ __forceinline__ void q_mfma(f32x16 (&acc0)[2][2], f32x16 (&acc1)[2][2], const f16x8 (&w)[2][2], const f16x8 (&a)[2][2]) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#ifdef Q_KO_MFMA        // (diagnostic knock-outs, csrc/diag/gin_stamp.hip: never defined in the library build)
            asm volatile("" :: "v"(w[nb][0]), "v"(w[nb][1]), "v"(a[mt][0]), "v"(a[mt][1]));
            if (FIRST) { acc0[mt][nb] = z; acc1[mt][nb] = z; }
#else
            acc0[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nb][0], a[mt][0], FIRST ? z : acc0[mt][nb], 0, 0, 0);
            acc1[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nb][0], a[mt][1], FIRST ? z : acc1[mt][nb], 0, 0, 0);
            acc1[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[nb][1], a[mt][0], acc1[mt][nb], 0, 0, 0);
#endif
        }
}

// One GEMM of the stream, fully unrolled: STEPS 16-deep K steps starting at stream position g0 (a multiple of 4, so ring
// slot = step & 3); its first three steps are in flight on entry, and it requests up to three steps past its own end
// when the stream goes on (CONT).  The A fragments of step s + 1 are read (into the other register set) before the MFMAs
// of step s.
template <int STEPS, bool CONT>
__device__ __forceinline__ void q_gemm(f32x16 (&acc0)[2][2], f32x16 (&acc1)[2][2], const char *Ap, int sa, const QStream &S,
                                       int g0, unsigned lane_off, int fi, int fh, f16x8 (&w)[Q_RING][2][2]) {
    static_assert(Q_RING == 4 && STEPS % 4 == 0, "slot arithmetic below");
    const int piece_bytes = Q_BM * sa;
    const char *arow = Ap + fi * sa + fh * 16;                   // lane (row fi, k half fh)
    f16x8 a[2][2][2];
    q_read_a(a[0], arow, sa, piece_bytes, 0);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
#ifndef Q_KO_W
        if (CONT || s + 3 < STEPS) q_request(w[(s + 3) & 3], S, g0 + s + 3, lane_off);
#endif
        if (s + 1 < STEPS) q_read_a(a[(s + 1) & 1], arow, sa, piece_bytes, s + 1);
        Q_PIN();
        if (s == 0) q_mfma<true>(acc0, acc1, w[s & 3], a[s & 1]);
        else q_mfma<false>(acc0, acc1, w[s & 3], a[s & 1]);
        Q_PIN();
    }
}

// the same with a run-time number of steps (a multiple of 4): K paddings no instantiation is built for
__device__ __forceinline__ void q_gemm_loop(f32x16 (&acc0)[2][2], f32x16 (&acc1)[2][2], const char *Ap, int sa, const QStream &S,
                                            unsigned lane_off, int fi, int fh, f16x8 (&w)[Q_RING][2][2]) {
    const int piece_bytes = Q_BM * sa;
    const char *arow = Ap + fi * sa + fh * 16;
    f16x8 a[2][2][2];
    q_read_a(a[0], arow, sa, piece_bytes, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[mt][nb][r] = 0.f; acc1[mt][nb][r] = 0.f; }
    for (int s0 = 0; s0 < S.steps1; s0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u;
#ifndef Q_KO_W
            q_request(w[(u + 3) & 3], S, s + 3, lane_off);
#endif
            q_read_a(a[(u + 1) & 1], arow, sa, piece_bytes, s + 1);     // (one step past the end: bytes of the tile, unused)
            Q_PIN();
            q_mfma<false>(acc0, acc1, w[u], a[u & 1]);
            Q_PIN();
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

__device__ __forceinline__ float4 q_bits4(const u32x4h v) { return __builtin_bit_cast(float4, v); }
__device__ __forceinline__ void q_keep(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// LG lanes per row in the gather, V float4 per lane (columns 4 (l + LG v) ..): K <= 4 LG V.  KQ: K padded to 64 when an
// instantiation is built for it (GEMM1 fully unrolled), 0 = run-time.  FULL: K == 4 LG V (no lane without columns).
template <int LG, int V, int ACT, bool POOL, int KQ, bool FULL>
__global__ void __launch_bounds__(Q_NTH, 2) k_gin_layer_h3q(const GinH3Args A) {
    constexpr int N = Q_N, XBM = Q_BM, NW = Q_NW, NTH = Q_NTH, LDO = Q_LDO, SH = Q_SH;
    constexpr int RPW = WAVE / LG;                     // rows per wavefront pass
    constexpr int PASSES = XBM / (NW * RPW);
    constexpr int NCV = (HCOLS + NTH - 1) / NTH;
    static_assert(!FULL || KQ == 4 * LG * V, "FULL means every lane owns columns of the padded K");
    extern __shared__ float4 lds4[];
    char *lds = reinterpret_cast<char *>(lds4);
    GIN_TICK(0);
    const int Kq = KQ ? KQ : ((A.Kp + 63) & ~63);   // K padded to whole groups of four 16-deep steps (zero columns)
    const int XS = FULL ? KQ : A.K;                 // fp32 input row stride (floats)
    const int sa = Kq * 2 + 16;                     // aggregated piece row stride (bytes)
    // the one region: fp32 input rows [XBM + 1][XS] (row XBM: zeros) -> aggregated pieces [2][XBM][sa] -> hidden pieces
    // [2][XBM][SH] -> fp32 output tile [XBM][LDO]
    float *Xs = reinterpret_cast<float *>(lds);
    char *Ap = lds, *Hp = lds;
    float *Ys = reinterpret_cast<float *>(lds);
    int *rp = reinterpret_cast<int *>(lds + Q_REGION);      // [XBM + 4]
    int *cs = rp + XBM + 4;                                 // [HCOLS]
    float *ainv = reinterpret_cast<float *>(cs + HCOLS);    // [XBM] 1 / row scale of the aggregate
    unsigned *hmax = reinterpret_cast<unsigned *>(ainv + XBM);   // [XBM] row maxima of the hidden tile (float bits)
    int *rgraph = reinterpret_cast<int *>(hmax + XBM);           // [XBM] graph of every tile row (fused readout)
    float *rweight = reinterpret_cast<float *>(rgraph + XBM);    // [XBM] readout weight of every tile row
    int *nfar = reinterpret_cast<int *>(rweight + XBM);          // [XBM] out-of-tile neighbours among a row's first four
    int *nb = nfar + XBM;                                        // [XBM][4] first four neighbours of every tile row, resolved
    int *nbo = nb + 4 * XBM;                                     // [XBM][4] the same as byte offsets of their staged rows
    float *prm = reinterpret_cast<float *>(nbo + 4 * XBM);       // [Q_PRM][N] per-channel parameters
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int sub = lane / LG, l = lane % LG;
    const int n_base = wid * 64;
    const hrsrc_t W1 = h3_rsrc(A.w1p, A.Kp, N), W2 = h3_rsrc(A.w2p, N, N);
    const unsigned lane_off = ((unsigned)fh * N + (unsigned)(n_base + fi)) * 16u;
    const int k4n = A.K >> 2, k4p = Kq >> 2;
    const float4 *__restrict__ X = reinterpret_cast<const float4 *>(A.x);
    const int64_t ldx4 = A.ldx >> 2;
    const int m0 = blockIdx.x * XBM;
    const int rows_here = min(A.n - m0, XBM);
    const bool vec_out = (A.ldy % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.y) & 15u) == 0);
    // tile-local buffer resources: offsets are 32-bit whatever the size of x / y, rows past n fail the range check
    const hrsrc_t XR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.x + (int64_t)m0 * A.ldx), 0,
                                                         (int)(((int64_t)(rows_here - 1) * A.ldx + A.K) * 4), 0x00020000);
    const hrsrc_t YR = __builtin_amdgcn_make_buffer_rsrc(A.y + (int64_t)m0 * A.ldy, 0,
                                                         (int)(((int64_t)(rows_here - 1) * A.ldy + N) * 4), 0x00020000);
    bool ok[V];
    int lb[V];                                               // byte offset of the lane's v-th float4 inside a row
#pragma unroll
    for (int v = 0; v < V; ++v) { ok[v] = FULL || l + LG * v < k4n; lb[v] = 16 * (l + LG * v); }

    // ---- (0) requests: rowptr slice, edge range -> column indices, feature rows, per-channel parameters ------------
    int rpv = 0;
    if (threadIdx.x <= XBM) {
        const int r = m0 + threadIdx.x;
        rpv = A.rowptr[r < A.n ? r : A.n];
    }
    const int e_lo = A.rowptr[m0];                                   // uniform addresses: scalar loads
    const int ncol_all = A.rowptr[m0 + XBM < A.n ? m0 + XBM : A.n] - e_lo;
    const bool cols_in_lds = ncol_all <= HCOLS;
    {
        int cv[NCV];
#pragma unroll
        for (int i = 0; i < NCV; ++i) {
            const int idx = threadIdx.x + NTH * i;
            cv[i] = (cols_in_lds && idx < ncol_all) ? A.col[e_lo + idx] : 0;
        }
        u32x4h xin[PASSES][V];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * NW + wid) * RPW + sub;
#pragma unroll
            for (int v = 0; v < V; ++v)     // (a lane without columns asks past the end of the resource: zeros)
                xin[p][v] = __builtin_amdgcn_raw_buffer_load_b128(XR, ok[v] ? (unsigned)(sub * A.ldx * 4 + lb[v]) : 0x40000000u,
                                                                  (unsigned)((m - sub) * A.ldx * 4), 0);
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
        for (int i = 0; i < NCV; ++i) {
            const int idx = threadIdx.x + NTH * i;
            if (cols_in_lds && idx < ncol_all) cs[idx] = cv[i];
        }
#pragma unroll
        for (int i = 0; i < Q_PRM; ++i) prm[i * N + threadIdx.x] = pv[i];
        // the row of zeros behind the tile: what an unused neighbour slot adds
        if (threadIdx.x < LG * V) *reinterpret_cast<float4 *>(Xs + XBM * XS + 4 * threadIdx.x) = f4_zero();
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * NW + wid) * RPW + sub;
#pragma unroll
            for (int v = 0; v < V; ++v)
                if (ok[v]) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(Xs) + m * XS * 4 + lb[v]) = q_bits4(xin[p][v]);
        }
    }
    lds_barrier();
    GIN_TICK(1);

    // neighbour table: slot q of tile row m resolved ONCE.  nb >= 0: tile-local row to add; -1: nothing (unused slot /
    // self loop); <= -2: out-of-tile row -2 - v.  nbo: byte offset of the row to add (the zero row for the last two kinds).
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
        nbo[threadIdx.x] = (v >= 0 ? v : XBM) * XS * 4;
        // the quad of a row counts its out-of-tile slots
        const unsigned long long fb = __ballot(v <= -2);
        if (q == 0) nfar[m] = __popcll((fb >> (lane & ~3)) & 15ull);
    }
    lds_barrier();

    // ---- (1) gather from LDS (+ out-of-tile rows from L2) into registers ---------------------------------------------
    float4 accv[PASSES][V];
    {
#pragma clang fp contract(off)      // (1 + eps) x + n0 stays a product and a sum, as in every other GIN kernel of the library
        // out-of-tile neighbours of the first and of the last pass (rows 0 .. 7 and 56 .. 63: where the tile's edges cut
        // molecules): requested now, added behind the LDS gather
        constexpr int NE = PASSES > 1 ? 2 : 1;
        float4 fu[NE][4][V];
        bool efar[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int p = e ? PASSES - 1 : 0;
            const int m = (p * NW + wid) * RPW + sub;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int v = 0; v < V; ++v) fu[e][q][v] = f4_zero();
            efar[e] = __builtin_amdgcn_ballot_w64(nfar[m] > 0) != 0;            // wave-uniform
            if (efar[e]) {
                const int4 nv = *reinterpret_cast<const int4 *>(nb + 4 * m);
                const int nq[4] = {nv.x, nv.y, nv.z, nv.w};
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (nq[q] <= -2 && ok[v]) fu[e][q][v] = X[(int64_t)(-2 - nq[q]) * ldx4 + (lb[v] >> 4)];
            }
        }
        const char *xb[V];
#pragma unroll
        for (int v = 0; v < V; ++v) xb[v] = reinterpret_cast<const char *>(Xs) + (ok[v] ? lb[v] : 0);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * NW + wid) * RPW + sub;
            const int4 o = *reinterpret_cast<const int4 *>(nbo + 4 * m);
            const int oq[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int v = 0; v < V; ++v) {
                accv[p][v] = *reinterpret_cast<const float4 *>(xb[v] + m * XS * 4);
                accv[p][v].x *= A.self_coef; accv[p][v].y *= A.self_coef;
                accv[p][v].z *= A.self_coef; accv[p][v].w *= A.self_coef;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int v = 0; v < V; ++v) f4_add(accv[p][v], *reinterpret_cast<const float4 *>(xb[v] + oq[q]));
            // (two passes = twenty 16-byte reads in flight per lane.  Unbounded, the reads of all passes are hoisted and
            //  their adds sunk into the per-pass code below, with every row read spilled in between: the empty asm makes
            //  the sums values that exist HERE)
            if (p & 1) {
#pragma unroll
                for (int pp = p - 1; pp <= p; ++pp)
#pragma unroll
                    for (int v = 0; v < V; ++v) q_keep(accv[pp][v]);
                Q_PIN();
            }
        }
#pragma unroll
        for (int e = 0; e < NE; ++e)
            if (efar[e]) {
                const int p = e ? PASSES - 1 : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v) f4_add(accv[p][v], fu[e][q][v]);
            }
        // out-of-tile neighbours of the passes in between (molecules larger than the tile), rows of degree > 4 and every
        // neighbour of a tile whose column slice does not fit LDS: from global memory, one pass at a time
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            const int m = (p * NW + wid) * RPW + sub;
            const int row = m0 + m;
            if (p > 0 && p < PASSES - 1 && __builtin_amdgcn_ballot_w64(nfar[m] > 0) != 0) {
                const int4 nv = *reinterpret_cast<const int4 *>(nb + 4 * m);
                const int nq[4] = {nv.x, nv.y, nv.z, nv.w};
                float4 fm[4][V];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        fm[q][v] = f4_zero();
                        if (nq[q] <= -2 && ok[v]) fm[q][v] = X[(int64_t)(-2 - nq[q]) * ldx4 + (lb[v] >> 4)];
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int v = 0; v < V; ++v) f4_add(accv[p][v], fm[q][v]);
            }
            const int e0 = rp[m], cnt = rp[m + 1] - e0;
            const int first = cols_in_lds ? 4 : 0;
            for (int e = e0 + first; e < e0 + cnt; ++e) {
                const int jj = A.col[e];
                if (jj != row) {
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        if (ok[v]) f4_add(accv[p][v], X[(int64_t)jj * ldx4 + (lb[v] >> 4)]);
                }
            }
        }
    }
    const QStream S{W1, W2, (unsigned)A.Kp * N * 2u, (unsigned)N * N * 2u, Kq / 16};
    f16x8 wfrag[Q_RING][2][2];
#pragma unroll
    for (int u = 0; u < Q_RING - 1; ++u) q_request(wfrag[u], S, u, lane_off);
    lds_barrier();                                  // every wavefront is done with the fp32 input rows
    GIN_TICK(2);

    // ---- (1b) row maximum -> power-of-two scale -> two fp16 pieces, parked over the input rows ------------------------
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int m = (p * NW + wid) * RPW + sub;
        float amax = 0.f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (!FULL && !ok[v]) accv[p][v] = f4_zero();
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(accv[p][v].x), fabsf(accv[p][v].y)), fmaxf(fabsf(accv[p][v].z), fabsf(accv[p][v].w))));
        }
        amax = q_row_max<LG>(amax, lane);
        float scale, inv;
        pow2_scale(amax, scale, inv);
        if (l == 0) ainv[m] = inv;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (FULL || l + LG * v < k4p) {                 // the K padding columns receive zeros
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
    if (KQ) q_gemm<(KQ ? KQ : 64) / 16, true>(acc0, acc1, Ap, sa, S, 0, lane_off, fi, fh, wfrag);
    else q_gemm_loop(acc0, acc1, Ap, sa, S, lane_off, fi, fh, wfrag);
    GIN_TICK(4);
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
    q_gemm<N / 16, false>(acc0, acc1, Hp, SH, S, S.steps1, lane_off, fi, fh, wfrag);
    GIN_TICK(7);
    lds_barrier();                                  // every wavefront is done with the hidden pieces
#pragma unroll
    for (int nbk = 0; nbk < 2; ++nbk)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = n_base + 32 * nbk + 8 * g + 4 * fh;
            const float4 pb = *reinterpret_cast<const float4 *>(prm + N + c);
            const float4 psc = *reinterpret_cast<const float4 *>(prm + 2 * N + c);      // (1, 0) without an affine: exact
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
                v.x = fmaf(v.x, psc.x, psh.x); v.y = fmaf(v.y, psc.y, psh.y);
                v.z = fmaf(v.z, psc.z, psh.z); v.w = fmaf(v.w, psc.w, psh.w);
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
        if (vec_out) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, v), YR, (unsigned)(16 * lane), (unsigned)(m * A.ldy * 4), 0);
        } else if (live) {
            float *dst = A.y + (int64_t)row * A.ldy + 4 * lane;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
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
        const int rows = rows_here;
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
        for (int mb = 0; mb < rows; mb += 16) {
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

template <int LG, int V, int ACT, bool POOL, int KQ, bool FULL>
static int launch_h3q_inst(const GinH3Args &A, hipStream_t s) {
    static bool declared = false;
    const size_t bytes = h3q_lds_bytes();
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gin_layer_h3q<LG, V, ACT, POOL, KQ, FULL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gin_layer_h3: cannot raise the dynamic LDS limit to %zu bytes", bytes);
        declared = true;
    }
    const int blocks = (A.n + Q_BM - 1) / Q_BM;
    hipLaunchKernelGGL((k_gin_layer_h3q<LG, V, ACT, POOL, KQ, FULL>), dim3(blocks), dim3(Q_NTH), bytes, s, A);
    return launch_status("gin_layer_h3");
}

template <int LG, int V, int ACT, int KQ, bool FULL>
static int launch_h3q(const GinH3Args &A, hipStream_t s) {
    return A.pool ? launch_h3q_inst<LG, V, ACT, true, KQ, FULL>(A, s) : launch_h3q_inst<LG, V, ACT, false, KQ, FULL>(A, s);
}

template <int LG, int V, int KQ, bool FULL>
static int h3q_pick_act(const GinH3Args &A, hipStream_t s) {
    switch (A.epi.act) {
        case BLGNN_ACT_RELU:    return launch_h3q<LG, V, BLGNN_ACT_RELU, KQ, FULL>(A, s);
        case BLGNN_ACT_PRELU:   return launch_h3q<LG, V, BLGNN_ACT_PRELU, KQ, FULL>(A, s);
        case BLGNN_ACT_SIGMOID: return launch_h3q<LG, V, BLGNN_ACT_SIGMOID, KQ, FULL>(A, s);
        case BLGNN_ACT_TANH:    return launch_h3q<LG, V, BLGNN_ACT_TANH, KQ, FULL>(A, s);
        default:                return launch_h3q<LG, V, BLGNN_ACT_IDENTITY, KQ, FULL>(A, s);
    }
}

// K <= 64: 16 lanes per row; K <= 128: 32 lanes; K <= 256: 32 lanes x 8 columns.  The widths the models use (64, 128, 256
// and the input widths below 64) get a fully unrolled GEMM1; K = 4 LG V exactly additionally drops the column predicates.
static int h3q_dispatch(const GinH3Args &A, hipStream_t s) {
    const int Kq = (A.Kp + 63) & ~63;
    if (A.K == 256) return h3q_pick_act<32, 2, 256, true>(A, s);
    if (A.K == 128) return h3q_pick_act<32, 1, 128, true>(A, s);
    if (A.K == 64) return h3q_pick_act<16, 1, 64, true>(A, s);
    if (Kq == 64) return h3q_pick_act<16, 1, 64, false>(A, s);
    if (Kq == 128) return h3q_pick_act<32, 1, 0, false>(A, s);
    return h3q_pick_act<32, 2, 0, false>(A, s);
}

}  // namespace blgnn
