// Cross-graph attention (model/layers_gmn.py:55-84) on the sp planes of x, result straight into the sp format:
// ONE WAVEFRONT PER PAIR, both directions from the same staged rows, f16x3 on v_mfma_f32_32x32x16_f16.
//
// Why (configs[2], 8 192 pairs, d = 256): the fp32 attention kernels (gmn_match.hip, gmn_match_wave.hpp) plus the
// streaming split of their result into the sp format that f_node's GEMM reads cost 2.7 ms of the 15.1 ms step -- the
// step without them: 12.3 ms.  They read x twice as fp32 through 32-column register-staged chunks with exact-f32
// MFMAs (64 cycles per 32 x 32 x 2), one chunk of prefetch, and write u as fp32 for a second kernel to split.  Here
//   * the rows arrive as the fp16 piece planes the producing GIN layer / GEMM already wrote (same bytes as fp32), by
//     LDS-DMA: per 32-column chunk (one 128-byte line per row: 32 high pieces, 32 low pieces) 8 rows per 1 KB
//     request, next chunk in flight while this one is consumed, no staging registers;
//   * cosines need no scale (a row's power-of-two scale cancels): S = G1 G2^T and S^T = G2 G1^T from the SAME operand
//     fragments with the roles swapped (A and B fragments of the 32 x 32 x 16 MFMA have the same lane layout), three
//     fp16 products per fp32 product; squared norms from the fragments with v_dot2_f32_f16;
//   * E = exp(cos) stays in the accumulator registers and becomes the A operand of the second product as it stands
//     (own node on the lane, partner nodes in the registers: U_own = X_own - diag(1 / sum) E^T X_partner sums over the
//     register index; the k order inside a step is the accumulator's row order, and the B operand -- partner rows, k =
//     node -- is fetched in that same order with ds_read_b64_tr_b16 from the row-major chunk image);
//   * the partner rows' scales ride on E (E' = E 2^(e_p - e_max + 10)); the result row's scale is a bound, not a
//     maximum: |u| <= |x_own| + max |x_partner| < 2^15 2^(max(e_own, e_max) + 1 - 127), so no pass over finished rows;
//   * a 32 x 32 result tile goes through 4.6 KB of LDS to meet its own values row-wise and leaves as two fp16 piece
//     groups per lane, 16-byte stores (sp_common.hpp's pair exchange).
// No workgroup barrier anywhere: a wavefront's LDS operations execute in order.
//
// Pairs with a graph above 64 nodes keep the fp32 kernels; blgnn_split_rows_sp_pairs converts only their rows.
#include "common.hpp"
#include "sp_common.hpp"

namespace blgnn {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef _Float16 f16x2s __attribute__((ext_vector_type(2)));
typedef short s16x4s __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) s16x4s msp_lds_s16x4;
typedef __attribute__((address_space(3))) void msp_lds_void;

struct MatchSpArgs {
    const int32_t *seg;
    const int32_t *list;
    int32_t n_list;
    const _Float16 *xs;
    int64_t ldxs;
    const uint8_t *xe;
    int64_t ldxe;
    const int32_t *row_map;          // null: plane row = merged atom; else plane row = row_map[merged atom]
    int64_t x_rows;                  // rows of the x planes
    _Float16 *us;
    int64_t ldus;
    uint8_t *ue;
    int64_t ldue;
};

#ifdef MSP_STAMP
__device__ unsigned long long g_msp_stamps[16];          // diag build (csrc/diag/msp_stamp.hip): one wavefront's phase stamps
#define MSP_T(k) do { if (stamp_on && lane == 0) g_msp_stamps[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MSP_T(k) do { } while (0)
#endif

template <int T> struct MatchSp {
    static constexpr int R = 32 * T;                       // rows per graph (padded)
    static constexpr int GRAPH = R * 128;                  // one graph's chunk image: R rows x (64 B high | 64 B low)
    static constexpr int STAGE = 2 * GRAPH;
    static constexpr int RING = 2 * STAGE;
    static constexpr int ZLD = 36;                         // floats per staged result row (16-byte aligned rows)
    static constexpr int ZST = 32 * ZLD * 4;
    static constexpr int ARR = R * 4;                      // one per-row float array of one graph
    static constexpr int BYTES = RING + ZST + 8 * ARR;     // 39 424 (T = 2) / 22 016 (T = 1): multiples of 128
    static constexpr int WAVES = T == 1 ? 2 : 4;           // per workgroup: 44 KB (three per CU) / 158 KB (one per CU)
};

// power of two from a biased exponent field; fields below 1 flush to zero
__device__ __forceinline__ float msp_pow2(int field) { return field >= 1 ? __uint_as_float((unsigned)field << 23) : 0.f; }

__device__ __forceinline__ float msp_sq(const f16x8 a, const f16x8 b) {       // sum (a + b)^2 over the 8 columns
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f16x2s a2 = {a[2 * k], a[2 * k + 1]}, b2 = {b[2 * k], b[2 * k + 1]};
        s = __builtin_amdgcn_fdot2(a2, a2, s, false);
        s = __builtin_amdgcn_fdot2(a2, b2 + b2, s, false);
        s = __builtin_amdgcn_fdot2(b2, b2, s, false);
    }
    return s;
}

// chunk C_ of both graphs into the ring stage at ST_: piece i of graph g only when the graph has rows there
// (wave-uniform).  The request offset goes through a named scalar: with an element of an array whose size depends on
// the template parameter as the builtin's argument, hipcc's host pass drops the kernel's stub without a diagnostic.
#define MSP_DMA(ST_, C_)                                                                                               \
    do {                                                                                                              \
        char *st_ = (ST_);                                                                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_)                                                         \
                if (i_ < Pn[g_]) {                                                                                    \
                    const int vo_ = voff[g_][i_];                                                                     \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(RX, (msp_lds_void *)(st_ + g_ * GRAPH + i_ * 1024), 16, vo_, \
                                                             (C_) * 128, 0, 0);                                       \
                }                                                                                                     \
    } while (0)

template <int T, bool MAP>
__global__ void __launch_bounds__(MatchSp<T>::WAVES * 64, 1) k_gmn_match_sp(const MatchSpArgs A) {
    typedef MatchSp<T> W;
    constexpr int R = W::R, GRAPH = W::GRAPH, STAGE = W::STAGE, ZLD = W::ZLD, NP = R / 8, NCH = 8;
    extern __shared__ float4 msp_smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * W::WAVES + wid;
    if (item >= A.n_list) return;                                              // (wave-uniform; no barrier in this kernel)
    const int pair = A.list[item];
    const int s0 = A.seg[2 * pair], s1 = A.seg[2 * pair + 1], s2 = A.seg[2 * pair + 2];
    const int n[2] = {s1 - s0, s2 - s1};
    if (n[0] > R || n[1] > R || (n[0] | n[1]) <= 0) return;                    // another size class / empty pair
    const int sg[2] = {s0, s1};
    int Tn[2], Pn[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) { Tn[g] = (n[g] + 31) >> 5; Pn[g] = (n[g] + 7) >> 3; }

#ifdef MSP_STAMP
    const bool stamp_on = item == A.n_list / 2 + 1;
#endif
    MSP_T(0);
    char *base = reinterpret_cast<char *>(msp_smem) + wid * W::BYTES;
    float *zst = reinterpret_cast<float *>(base + W::RING);
    float *arr = zst + 32 * ZLD;                                               // [8][R]: invn, pf, cA, cB x 2 graphs
    auto invn = [&](int g) { return arr + (0 + g) * R; };
    auto pfac = [&](int g) { return arr + (2 + g) * R; };
    auto cfa = [&](int g) { return arr + (4 + g) * R; };
    auto cfb = [&](int g) { return arr + (6 + g) * R; };

    // the ring once to zero: rows no request of this pair writes are then finite (they meet E = 0 / masked lanes only)
    {
        const float4 z = f4_zero();
#pragma unroll
        for (int k = 0; k < W::RING / 1024; ++k) *reinterpret_cast<float4 *>(base + k * 1024 + lane * 16) = z;
    }

    const int r = lane & 31, h = lane >> 5;
    const int64_t rowb = A.ldxs * 2;                                           // bytes per plane row
    // ---- request plan: piece i of graph g = rows 8 i + (lane >> 3) of the chunk image, LDS slot lane & 7 <- global
    // 16-byte piece slot ^ ((row >> 1) & 7) (linear_sp.hip's swizzle: fragment reads of 16 rows and the transposed
    // reads of 4 rows x 64 B then touch every bank once)
    const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(A.xs) + (MAP ? (int64_t)0 : (int64_t)s0 * rowb)), 0,
        (int)(unsigned)(MAP ? A.x_rows * rowb : (int64_t)(n[0] + n[1]) * rowb), 0x00020000);
    int voff[2][NP];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = 8 * i + (lane >> 3);
            const int rc = row < n[g] ? row : (n[g] > 0 ? n[g] - 1 : 0);
            int src = sg[g] - s0 + rc;
            if (MAP) src = (i < Pn[g]) ? A.row_map[sg[g] + rc] : 0;
            voff[g][i] = (int)((unsigned)src * (unsigned)rowb + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4));
        }
    // the rows' exponent bytes: lane r of either half holds row 32 t + r
    int eb[2][T];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int row = 32 * t + r;
            const bool ok = row < n[g];
            const int64_t src = MAP ? (ok ? A.row_map[sg[g] + row] : 0) : (int64_t)(sg[g] + (ok ? row : 0));
            const int e = A.xe[src * A.ldxe];
            eb[g][t] = ok ? e : 0;
        }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");               // zero fill done before any request lands
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(voff[g][i]));
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < T; ++t) asm volatile("" : "+v"(eb[g][t]));

    // ================================ pass 1: S, S^T and the squared norms ================================
    // C0[t2][t1]: rows = nodes of graph 2, columns (lanes) = nodes of graph 1;  C1[t1][t2]: the transpose
    f32x16s C0[T][T], C1[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) { C0[i][j][k] = 0.f; C1[i][j][k] = 0.f; }
    float nq[2][T];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < T; ++t) nq[g][t] = 0.f;
    const int fsw = (r >> 1) & 7;                                              // (32 t + r) >> 1 & 7: the same for every tile
    MSP_T(1);
    MSP_DMA(base, 0);
    for (int c = 0; c < NCH; ++c) {
        if (c == 3) MSP_T(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c == 3) MSP_T(8);
        if (c + 1 < NCH) MSP_DMA(base + ((c + 1) & 1) * STAGE, c + 1);
        const char *st = base + (c & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f16x8 a[2][T][2];
            const int slot = (((2 * s + h) ^ fsw) << 4);
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (t < Tn[g]) {
                        const int off = g * GRAPH + (32 * t + r) * 128 + slot;
                        a[g][t][0] = *reinterpret_cast<const f16x8 *>(st + off);
                        a[g][t][1] = *reinterpret_cast<const f16x8 *>(st + (off ^ 64));
                        nq[g][t] += msp_sq(a[g][t][0], a[g][t][1]);
                    }
#pragma unroll
            for (int t2 = 0; t2 < T; ++t2)
#pragma unroll
                for (int t1 = 0; t1 < T; ++t1)
                    if (t2 < Tn[1] && t1 < Tn[0]) {
                        C0[t2][t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][t2][0], a[0][t1][0], C0[t2][t1], 0, 0, 0);
                        C0[t2][t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][t2][0], a[0][t1][1], C0[t2][t1], 0, 0, 0);
                        C0[t2][t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][t2][1], a[0][t1][0], C0[t2][t1], 0, 0, 0);
                        C1[t1][t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][t1][0], a[1][t2][0], C1[t1][t2], 0, 0, 0);
                        C1[t1][t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][t1][0], a[1][t2][1], C1[t1][t2], 0, 0, 0);
                        C1[t1][t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][t1][1], a[1][t2][0], C1[t1][t2], 0, 0, 0);
                    }
        }
    }
    MSP_T(2);
    // pass 2's first chunk on its way while the softmax runs (both stages are free: every fragment read has returned)
    MSP_DMA(base, 0);

    // ================================ softmax ================================
    float invl[2][T];
    int eref[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int em = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float nrm = nq[g][t] + __shfl_xor(nq[g][t], 32, WAVE);
            invl[g][t] = 1.f / fmaxf(sqrtf(nrm), 1e-12f);
            invn(g)[32 * t + r] = invl[g][t];
            em = eb[g][t] > em ? eb[g][t] : em;
        }
        eref[g] = __builtin_amdgcn_readfirstlane(group_max(em, WAVE));
    }
    int eout[2][T];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int row = 32 * t + r;
            pfac(g)[row] = row < n[g] ? msp_pow2(127 + 10 + eb[g][t] - eref[g]) : 0.f;
            const int eo = (eb[g][t] > eref[1 - g] ? eb[g][t] : eref[1 - g]) + 1;
            eout[g][t] = eo > 254 ? 254 : eo;
            if (row < n[g] && h == 0) A.ue[(int64_t)(sg[g] + row) * A.ldue] = (uint8_t)eout[g][t];
        }
    asm volatile("" ::: "memory");
    // E fragments: Ef[d][to][tp][s][piece], own graph d on the lanes
    f16x8 Ef[2][T][T][2][2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int gp = 1 - d;
#pragma unroll
        for (int to = 0; to < T; ++to) {
            float sum = 0.f;
#pragma unroll
            for (int tp = 0; tp < T; ++tp) {
                if (to < Tn[d] && tp < Tn[gp]) {
                    const f32x16s Cc = d == 0 ? C0[tp][to] : C1[tp][to];
                    float ev[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int p0 = 32 * tp + 8 * q + 4 * h;
                        const float4 ip = *reinterpret_cast<const float4 *>(invn(gp) + p0);
                        const float4 pf = *reinterpret_cast<const float4 *>(pfac(gp) + p0);
                        const float ipv[4] = {ip.x, ip.y, ip.z, ip.w}, pfv[4] = {pf.x, pf.y, pf.z, pf.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float e = (p0 + j < n[gp]) ? expf(Cc[4 * q + j] * invl[d][to] * ipv[j]) : 0.f;
                            sum += e;
                            ev[4 * q + j] = e * pfv[j];
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const _Float16 e1 = (_Float16)ev[8 * s + j];
                            Ef[d][to][tp][s][0][j] = e1;
                            Ef[d][to][tp][s][1][j] = (_Float16)(ev[8 * s + j] - (float)e1);
                        }
                }
            }
            if (to < Tn[d]) {
                sum += __shfl_xor(sum, 32, WAVE);
                const int row = 32 * to + r;
                cfa(d)[row] = msp_pow2(127 + eb[d][to] - eout[d][to]);
                cfb(d)[row] = sum > 0.f ? -msp_pow2(127 + eref[gp] - 10 - eout[d][to]) / sum : 0.f;
            }
        }
    }
    asm volatile("" ::: "memory");

    // ================================ pass 2: U = X_own - diag(1 / sum) E^T X_partner ================================
    // transposed-read plan (ds_read_b64_tr_b16): lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of
    // the group's 16 columns; lanes 16 G .. 16 G + 15 of a half take columns 16 G ..; the half h takes partner rows
    // 16 s + 4 h .. + 3 and 16 s + 8 + 4 h .. + 3 (the accumulator's row order)
    MSP_T(3);
    const int tq = (lane & 15) >> 2, tpp = lane & 3, tG = (lane >> 4) & 1;
    const int tsw = (2 * h + (tq >> 1)) & 7;                                   // (row >> 1) & 7 of the first block's row
    const int trow0 = (4 * h + tq) * 128 + (((2 * tG + (tpp >> 1)) ^ tsw) << 4) + 8 * (tpp & 1);
    const int trow1 = (4 * h + tq + 8) * 128 + (((2 * tG + (tpp >> 1)) ^ tsw ^ 4) << 4) + 8 * (tpp & 1);
    const int cg = lane & 7, rw = lane >> 3;                                   // row-wise stage: 4 columns cg, row 8 w + rw
    for (int c = 0; c < NCH; ++c) {
        if (c == 3) MSP_T(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c == 3) MSP_T(6);
        if (c + 1 < NCH) MSP_DMA(base + ((c + 1) & 1) * STAGE, c + 1);
        const char *st = base + (c & 1) * STAGE;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int gp = 1 - d;
            f16x8 Bf[T][2][2];
#pragma unroll
            for (int tp = 0; tp < T; ++tp)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (tp < Tn[gp]) {
                        const int o0 = gp * GRAPH + (32 * tp + 16 * s) * 128;
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc) {
                            const s16x4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (msp_lds_s16x4 *)(st + ((o0 + trow0) ^ (64 * pc))));
                            const s16x4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (msp_lds_s16x4 *)(st + ((o0 + trow1) ^ (64 * pc))));
                            const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
                            Bf[tp][s][pc] = f16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                        }
                    }
#pragma unroll
            for (int to = 0; to < T; ++to) {
                if (to >= Tn[d]) continue;
                f32x16s Z;
#pragma unroll
                for (int k = 0; k < 16; ++k) Z[k] = 0.f;
#pragma unroll
                for (int tp = 0; tp < T; ++tp)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        if (tp < Tn[gp]) {
                            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ef[d][to][tp][s][0], Bf[tp][s][0], Z, 0, 0, 0);
                            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ef[d][to][tp][s][0], Bf[tp][s][1], Z, 0, 0, 0);
                            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ef[d][to][tp][s][1], Bf[tp][s][0], Z, 0, 0, 0);
                        }
                // result tile -> LDS: row (own node) = (k & 3) + 8 (k >> 2) + 4 h, column (feature) = r
#pragma unroll
                for (int k = 0; k < 16; ++k) zst[((k & 3) + 8 * (k >> 2) + 4 * h) * ZLD + r] = Z[k];
                asm volatile("" ::: "memory");
                // row-wise: u = x_own - z / sum in the result row's unit, two fp16 pieces, 16-byte stores
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int lrow = 8 * w + rw, orow = 32 * to + lrow;
                    const float4 z4 = *reinterpret_cast<const float4 *>(zst + lrow * ZLD + 4 * cg);
                    const int xo = d * GRAPH + orow * 128 + (((cg >> 1) ^ ((orow >> 1) & 7)) << 4) + 8 * (cg & 1);
                    const f16x4 x1 = *reinterpret_cast<const f16x4 *>(st + xo);
                    const f16x4 x2 = *reinterpret_cast<const f16x4 *>(st + (xo ^ 64));
                    const float ca = cfa(d)[orow], cb = cfb(d)[orow];
                    f32x4v v;
                    v[0] = fmaf(z4.x, cb, ((float)x1[0] + (float)x2[0]) * ca);
                    v[1] = fmaf(z4.y, cb, ((float)x1[1] + (float)x2[1]) * ca);
                    v[2] = fmaf(z4.z, cb, ((float)x1[2] + (float)x2[2]) * ca);
                    v[3] = fmaf(z4.w, cb, ((float)x1[3] + (float)x2[3]) * ca);
                    const f16x4 h1 = __builtin_convertvector(v, f16x4);
                    const f16x4 h2 = __builtin_convertvector(v - __builtin_convertvector(h1, f32x4v), f16x4);
                    const uint2 u1 = __builtin_bit_cast(uint2, h1), u2 = __builtin_bit_cast(uint2, h2);
                    const bool odd = lane & 1;
                    const uint2 keep = odd ? u2 : u1, send = odd ? u1 : u2;
                    uint2 recv;
                    recv.x = (unsigned)__shfl_xor((int)send.x, 1, WAVE);
                    recv.y = (unsigned)__shfl_xor((int)send.y, 1, WAVE);
                    const uint4 out = odd ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y);
                    if (orow < n[d]) {
                        _Float16 *dst = A.us + (int64_t)(sg[d] + orow) * A.ldus + c * 64 + 4 * (cg & ~1) + (odd ? 32 : 0);
                        *reinterpret_cast<uint4 *>(dst) = out;
                    }
                }
                asm volatile("" ::: "memory");
            }
        }
    }
    MSP_T(4);
}

// fp32 rows -> sp format for the rows of the listed pairs only (the pairs the fp32 attention kernels took)
__global__ void __launch_bounds__(256) k_split_sp_pairs(const float *__restrict__ x, int64_t ldx, const int32_t *__restrict__ seg,
                                                        const int32_t *__restrict__ list, int n_list, int KB,
                                                        _Float16 *__restrict__ xs, int64_t ldxs, uint8_t *__restrict__ xe,
                                                        int64_t ldxe) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pair = list[blockIdx.x];
    const int m0 = seg[2 * pair], m1 = seg[2 * pair + 2];
    for (int64_t unit = (int64_t)m0 * KB + wid; unit < (int64_t)m1 * KB; unit += 4) {
        const int m = (int)(unit / KB), kb = (int)(unit % KB);
        const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)m * ldx + kb * SP_BLOCK + lane * 4);
        float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
        mx = group_max(mx, WAVE);
        float s, inv;
        pow2_scale(mx, s, inv);
        if (lane == 0) xe[(int64_t)m * ldxe + kb] = sp_exp_of(inv);
        const f32x4v sv = {v.x * s, v.y * s, v.z * s, v.w * s};
        sp_store4_pair(xs + (int64_t)m * ldxs + kb * (2 * SP_BLOCK), lane * 4, sv, lane);
    }
}

template <int T, bool MAP>
static int launch_match_sp(const MatchSpArgs &A, hipStream_t s) {
    typedef MatchSp<T> W;
    static bool declared = false;
    const int bytes = W::WAVES * W::BYTES;
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gmn_match_sp<T, MAP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gmn_match_sp: cannot raise the dynamic LDS limit");
        declared = true;
    }
    hipLaunchKernelGGL((k_gmn_match_sp<T, MAP>), dim3((A.n_list + W::WAVES - 1) / W::WAVES), dim3(W::WAVES * 64), bytes, s, A);
    return launch_status("gmn_match_sp");
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_gmn_match_sp(const int32_t *seg, int32_t num_pairs, const void *xs, int64_t ldxs, const uint8_t *xe,
                                  int64_t ldxe, const int32_t *row_map, int64_t x_rows, void *us, int64_t ldus, uint8_t *ue,
                                  int64_t ldue, int32_t d, const int32_t *pair_order, const int32_t *class_counts_host,
                                  void *stream) {
    BLGNN_CHECK_ARG(num_pairs >= 0, "gmn_match_sp: negative size");
    if (num_pairs == 0) return BLGNN_OK;
    if (d != SP_BLOCK) return fail(BLGNN_ERR_UNSUPPORTED, "gmn_match_sp: d = %d (one 256-column scale block per row); use blgnn_gmn_match", d);
    BLGNN_CHECK_ARG(seg && xs && xe && us && ue && pair_order && class_counts_host, "gmn_match_sp: null pointer");
    BLGNN_CHECK_ARG(xs != us, "gmn_match_sp: x and u planes must not alias");
    BLGNN_CHECK_ARG(ldxs >= 2 * d && ldxs % 8 == 0 && aligned16(xs) && ldus >= 2 * d && ldus % 8 == 0 && aligned16(us) &&
                    ldxe >= 1 && ldue >= 1, "gmn_match_sp: bad planes (16-byte aligned rows of >= 2 d fp16)");
    BLGNN_CHECK_ARG(class_counts_host[0] >= 0 && class_counts_host[1] >= 0 &&
                    (int64_t)class_counts_host[0] + class_counts_host[1] <= num_pairs, "gmn_match_sp: bad class counts");
    BLGNN_CHECK_ARG(x_rows > 0, "gmn_match_sp: x_rows");
    if (row_map && x_rows * ldxs * 2 >= ((int64_t)1 << 32))
        return fail(BLGNN_ERR_UNSUPPORTED, "gmn_match_sp: mapped x planes must stay below 4 GiB (32-bit buffer offsets)");
    if (ldxs * 2 * 128 >= ((int64_t)1 << 31)) return fail(BLGNN_ERR_UNSUPPORTED, "gmn_match_sp: plane rows too long");
    MatchSpArgs A;
    A.seg = seg; A.xs = reinterpret_cast<const _Float16 *>(xs); A.ldxs = ldxs; A.xe = xe; A.ldxe = ldxe;
    A.row_map = row_map; A.x_rows = x_rows;
    A.us = reinterpret_cast<_Float16 *>(us); A.ldus = ldus; A.ue = ue; A.ldue = ldue;
    hipStream_t s = (hipStream_t)stream;
    int rc = BLGNN_OK;
    if (class_counts_host[0] > 0) {
        A.list = pair_order; A.n_list = class_counts_host[0];
        rc = row_map ? launch_match_sp<1, true>(A, s) : launch_match_sp<1, false>(A, s);
        if (rc != BLGNN_OK) return rc;
    }
    if (class_counts_host[1] > 0) {
        A.list = pair_order + class_counts_host[0]; A.n_list = class_counts_host[1];
        rc = row_map ? launch_match_sp<2, true>(A, s) : launch_match_sp<2, false>(A, s);
    }
    return rc;
}

extern "C" int blgnn_split_rows_sp_pairs(const float *x, int64_t ldx, const int32_t *seg, const int32_t *pair_list,
                                         int32_t n_list, int32_t K, void *xs, int64_t ldxs, uint8_t *xe, int64_t ldxe,
                                         void *stream) {
    BLGNN_CHECK_ARG(n_list >= 0 && K > 0 && K % SP_BLOCK == 0, "split_rows_sp_pairs: K must be a positive multiple of 256");
    if (n_list == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && seg && pair_list && xs && xe && ldx >= K && ldxs >= 2 * (int64_t)K && ldxe >= K / SP_BLOCK,
                    "split_rows_sp_pairs: bad args");
    BLGNN_CHECK_ARG(ldx % 4 == 0 && aligned16(x) && ldxs % 8 == 0 && aligned16(xs), "split_rows_sp_pairs: 16-byte alignment");
    hipLaunchKernelGGL(k_split_sp_pairs, dim3(n_list), dim3(256), 0, (hipStream_t)stream, x, ldx, seg, pair_list, n_list,
                       K / SP_BLOCK, reinterpret_cast<_Float16 *>(xs), ldxs, xe, ldxe);
    return launch_status("split_rows_sp_pairs");
}
