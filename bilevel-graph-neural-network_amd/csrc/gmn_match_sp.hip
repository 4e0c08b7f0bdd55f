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
typedef unsigned uint2v __attribute__((ext_vector_type(2)));

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

template <int R_> struct MatchSp {
    static constexpr int R = R_;                           // rows per graph in the chunk image (32, 48 or 64)
    static constexpr int GRAPH = R * 128;                  // one graph's chunk image: R rows x (64 B high | 64 B low)
    static constexpr int STAGE = 2 * GRAPH;
    static constexpr int NST = 3;                          // ring stages: two chunks in flight beside the one being read
    static constexpr int RING = NST * STAGE;
    static constexpr int BYTES = RING + 4 * 64 * 4;        // + inverse norms and scale factors of both graphs' rows:
    //   R = 64: 50 176 B (three wavefronts per CU), 48: 37 888 B (four: every SIMD has one), 32: 25 600 B (six)
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

// what lane l ^ 32 holds (v_permlane32_swap: no LDS round trip)
__device__ __forceinline__ float msp_other_half(float v, int h) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);       // r[0] = low half's v, r[1] = high half's v
    return __uint_as_float(h ? r[0] : r[1]);
}

// Chunk C_ of both graphs into the ring stage at ST_: ALWAYS 2 * NP requests -- a lane whose row the graph does not
// have carries an offset beyond the resource's num_records: the range check answers with zeros and fetches nothing, so
// the image's unused rows are zero, the number of requests per chunk is a constant and s_waitcnt vmcnt can count
// chunks.  (The request offset goes through a named scalar: with an element of an array whose size depends on a
// template parameter as the builtin's argument, hipcc's host pass drops the kernel's stub without a diagnostic.)
#define MSP_DMA(ST_, C_)                                                                                               \
    do {                                                                                                              \
        char *st_ = (ST_);                                                                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < NP; ++i_) {                                                       \
                const int vo_ = voff[g_][i_];                                                                         \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(RX, (msp_lds_void *)(st_ + g_ * GRAPH + i_ * 1024), 16, vo_,    \
                                                         (C_) * 128, 0, 0);                                           \
            }                                                                                                         \
    } while (0)
#define MSP_WAIT_2() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NP) : "memory")
#define MSP_WAIT_1() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory")

// One pair with T1 / T2 tiles of 32 nodes in its two graphs (compile-time: the whole body is straight-line code, so
// the scheduler can put one wavefront's vector work under its own MFMAs -- there is one wavefront per SIMD).
template <int T1, int T2, int R, bool MAP>
__device__ __forceinline__ void msp_pair(const MatchSpArgs &A, char *base, const int lane, const int s0, const int s1,
                                         const int s2, const bool stamp_on) {
    typedef MatchSp<R> W;
    constexpr int GRAPH = W::GRAPH, STAGE = W::STAGE, NST = W::NST, NP = R / 8, NCH = 8, TM = T1 > T2 ? T1 : T2;
    constexpr int TT[2] = {T1, T2};
    // k steps (16 rows) of tile t that lie inside an R-row image: R = 48 has rows 32 .. 47 of its second tile only
#define MSP_KS(t) ((R - 32 * (t)) >= 32 ? 2 : 1)
    const int n[2] = {s1 - s0, s2 - s1};
    const int sg[2] = {s0, s1};
    float *arr = reinterpret_cast<float *>(base + W::RING);                    // [4][64]: inverse norms, scale factors x 2 graphs
    auto invn = [&](int g) { return arr + (0 + g) * 64; };
    auto pfac = [&](int g) { return arr + (2 + g) * 64; };

    const int r = lane & 31, h = lane >> 5;
    const int64_t rowb = A.ldxs * 2;                                           // bytes per plane row
    // ---- request plan: piece i of graph g = rows 8 i + (lane >> 3) of the chunk image, LDS slot lane & 7 <- global
    // 16-byte piece slot ^ ((row >> 1) & 7) (linear_sp.hip's swizzle: fragment reads of 16 rows and the transposed
    // reads of 4 rows x 64 B then touch every bank once)
    const int64_t span = MAP ? A.x_rows * rowb : (int64_t)(n[0] + n[1]) * rowb;
    const unsigned records = span > 0xfffff000ll ? 0xfffff000u : (unsigned)span;
    const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(A.xs) + (MAP ? (int64_t)0 : (int64_t)s0 * rowb)), 0, (int)records,
        0x00020000);
    int voff[2][NP];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = 8 * i + (lane >> 3);
            const bool ok = row < n[g];
            int src = sg[g] - s0 + row;
            if (MAP) src = ok ? A.row_map[sg[g] + row] : 0;
            voff[g][i] = ok ? (int)((unsigned)src * (unsigned)rowb + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4))
                            : (int)0xfffff000u;                                // beyond num_records (+ the chunk offset: no wrap): zeros
        }
    // the rows' exponent bytes: lane r of either half holds row 32 t + r
    int eb[2][TM];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < TT[g]; ++t) {
            const int row = 32 * t + r;
            const bool ok = row < n[g];
            const int64_t src = MAP ? (ok ? A.row_map[sg[g] + row] : 0) : (int64_t)(sg[g] + (ok ? row : 0));
            const int e = A.xe[src * A.ldxe];
            eb[g][t] = ok ? e : 0;
        }
    // result rows by buffer stores: a row the graph does not have lies beyond its resource's num_records (no predicate)
    __amdgpu_buffer_rsrc_t RU[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
        RU[g] = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(A.us + (int64_t)sg[g] * A.ldus), 0,
                                                  (int)(n[g] > 0 ? ((int64_t)(n[g] - 1) * A.ldus + 512) * 2 : 0), 0x00020000);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(voff[g][i]));
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < TT[g]; ++t) asm volatile("" : "+v"(eb[g][t]));

    // the 16 chunk loads of the kernel (pass 1: 0 .. 7, pass 2: 8 .. 15) form one sequence through the ring: load j
    // goes to stage j % 3 and is issued two steps ahead; requests return in order, every load is 2 NP requests, so
    // "at most 2 * 2 NP requests outstanding" means load j has landed (the result stores of pass 2 only add to the count)

    // ================================ pass 1: S, S^T and the squared norms ================================
    // C0[t2][t1]: rows = nodes of graph 2, columns (lanes) = nodes of graph 1;  C1[t1][t2]: the transpose
    f32x16s C0[T2][T1], C1[T1][T2];
#pragma unroll
    for (int i = 0; i < T2; ++i)
#pragma unroll
        for (int j = 0; j < T1; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) { C0[i][j][k] = 0.f; C1[j][i][k] = 0.f; }
    float nq[2][TM];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < TM; ++t) nq[g][t] = 0.f;
    const int fsw = (r >> 1) & 7;                                              // (32 t + r) >> 1 & 7: the same for every tile
    MSP_DMA(base, 0);
    MSP_DMA(base + STAGE, 1);
    MSP_T(1);
    int stg = 0;                                                               // stage of the chunk being read
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        {   // load c + 2 (pass 2's first two chunks behind pass 1's last two) into the stage read at step c - 1
            const int nxt = stg == 0 ? 2 : stg - 1;
            MSP_DMA(base + nxt * STAGE, (c + 2) & 7);
        }
        if (c == 3) MSP_T(7);
        MSP_WAIT_2();
        if (c == 3) MSP_T(8);
        const char *st = base + stg * STAGE;
        stg = stg == NST - 1 ? 0 : stg + 1;
        f16x8 a[2][2][TM][2];                                                  // [k step][graph][tile][piece]
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < TT[g]; ++t) {
                    // (R = 48: lanes 16 .. 31 of the second tile would address rows 48 .. 63; they re-read row 47 -- same
                    // (row >> 1) & 7 as row 63, so the slot holds -- and their scores are masked as rows the graph lacks)
                    const int row = (32 * t + r) < R ? (32 * t + r) : (R - 1);
                    const int off = g * GRAPH + row * 128 + ((((2 * s + h) ^ ((row >> 1) & 7))) << 4);
                    a[s][g][t][0] = *reinterpret_cast<const f16x8 *>(st + off);
                    a[s][g][t][1] = *reinterpret_cast<const f16x8 *>(st + (off ^ 64));
                }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t2 = 0; t2 < T2; ++t2)
#pragma unroll
                for (int t1 = 0; t1 < T1; ++t1) {
                    C0[t2][t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1][t2][0], a[s][0][t1][0], C0[t2][t1], 0, 0, 0);
                    C1[t1][t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0][t1][0], a[s][1][t2][0], C1[t1][t2], 0, 0, 0);
                    C0[t2][t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1][t2][0], a[s][0][t1][1], C0[t2][t1], 0, 0, 0);
                    C1[t1][t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0][t1][0], a[s][1][t2][1], C1[t1][t2], 0, 0, 0);
                    C0[t2][t1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][1][t2][1], a[s][0][t1][0], C0[t2][t1], 0, 0, 0);
                    C1[t1][t2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s][0][t1][1], a[s][1][t2][0], C1[t1][t2], 0, 0, 0);
                }
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int t = 0; t < TT[g]; ++t) nq[g][t] += msp_sq(a[s][g][t][0], a[s][g][t][1]);
            // the norms' vector work under the matrix cores (one wavefront per SIMD: nobody else fills the gaps)
#pragma unroll
            for (int k = 0; k < 6 * T1 * T2; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (16 * (T1 + T2) + 6 * T1 * T2 - 1) / (6 * T1 * T2), 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    MSP_T(2);

    // ================================ softmax ================================
    float invl[2][TM];
    int eref[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        int em = 0;
#pragma unroll
        for (int t = 0; t < TT[g]; ++t) {
            const float nrm = nq[g][t] + msp_other_half(nq[g][t], h);
            invl[g][t] = 1.f / fmaxf(sqrtf(nrm), 1e-12f);
            invn(g)[32 * t + r] = invl[g][t];
            em = eb[g][t] > em ? eb[g][t] : em;
        }
        eref[g] = __builtin_amdgcn_readfirstlane(group_max(em, WAVE));
    }
    int eout[2][TM];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int t = 0; t < TT[g]; ++t) {
            const int row = 32 * t + r;
            pfac(g)[row] = row < n[g] ? msp_pow2(127 + 10 + eb[g][t] - eref[g]) : 0.f;
            const int eo = (eb[g][t] > eref[1 - g] ? eb[g][t] : eref[1 - g]) + 1;
            eout[g][t] = eo > 254 ? 254 : eo;
            if (row < n[g] && h == 0) A.ue[(int64_t)(sg[g] + row) * A.ldue] = (uint8_t)eout[g][t];
        }
    asm volatile("" ::: "memory");
    // E fragments Ef[d][to][tp][s][piece] (own graph d on the lanes) carry everything that multiplies the partner rows:
    // exp(cos) x the partner row's scale 2^(e_p - e_max) x the own row's -2^(e_max - e_out) / sum, times 2^10 (a softmax
    // weight is ~1 / n: without the factor its low piece would be an fp16 subnormal and the pair would hold 18 bits, not 22);
    // Df[d][to][s]: the own row's 2^(e_own - e_out + 10) on the diagonal of a 32 x 32 block (one fp16 piece: a power of
    // two), which adds the own values in the same product.  The product leaves the matrix cores as 2^10 x the result in
    // the result row's unit.
    f16x8 Ef[2][TM][TM][2][2], Df[2][TM][2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int gp = 1 - d;
#pragma unroll
        for (int to = 0; to < TT[d]; ++to) {
            float sum = 0.f;
            const float io = invl[d][to] * 1.44269504088896340736f;          // exp(x) = 2^(x log2 e)
#pragma unroll
            for (int tp = 0; tp < TT[gp]; ++tp) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int p0 = 32 * tp + 8 * q + 4 * h;
                    const float4 ip = *reinterpret_cast<const float4 *>(invn(gp) + p0);
                    const float4 pf = *reinterpret_cast<const float4 *>(pfac(gp) + p0);
                    const float ipv[4] = {ip.x, ip.y, ip.z, ip.w}, pfv[4] = {pf.x, pf.y, pf.z, pf.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float cv = d == 0 ? C0[tp][to][4 * q + j] : C1[tp][to][4 * q + j];
                        const float e = (p0 + j < n[gp]) ? __builtin_amdgcn_exp2f(cv * io * ipv[j]) : 0.f;
                        sum += e;
                        if (d == 0) C0[tp][to][4 * q + j] = e * pfv[j]; else C1[tp][to][4 * q + j] = e * pfv[j];
                    }
                }
            }
            sum += msp_other_half(sum, h);
            const float a_ = msp_pow2(127 + eb[d][to] - eout[d][to]);
            const float b_ = sum > 0.f ? -msp_pow2(127 + eref[gp] - eout[d][to]) / sum : 0.f;   // (x 2^10 with the factor on the partner rows)
#pragma unroll
            for (int tp = 0; tp < TT[gp]; ++tp) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float ev = (d == 0 ? C0[tp][to][8 * s + j] : C1[tp][to][8 * s + j]) * b_;
                        // ONE rounded product for both pieces: left to itself hipcc forms the high piece from the fp32
                        // product and the low piece against fp16(fused product) -- at a tie the two differ by one fp16 step
                        asm volatile("" : "+v"(ev));
                        const _Float16 e1 = (_Float16)ev;
                        Ef[d][to][tp][s][0][j] = e1;
                        Ef[d][to][tp][s][1][j] = (_Float16)(ev - (float)e1);
                    }
                // the pieces are formed HERE: without the pin hipcc keeps the fp32 values and converts them again in
                // every chunk of pass 2 (~400 instructions per chunk)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) asm volatile("" : "+v"(Ef[d][to][tp][s][pc]));
            }
            const _Float16 ah = (_Float16)(a_ * 1024.f);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    Df[d][to][s][j] = (16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) == r ? ah : (_Float16)0.f;
                asm volatile("" : "+v"(Df[d][to][s]));
            }
        }
    }
    asm volatile("" ::: "memory");

    // ================================ pass 2: U^T = X_own^T D - X_partner^T E ================================
    // A operands = the rows of both graphs transposed (rows of the product = the chunk's 32 features, k = node), fetched
    // with ds_read_b64_tr_b16 in the accumulator's row order: lane 4 q + p of a 16-lane group addresses row q, columns
    // 4 p .. 4 p + 3 of the group's 16 columns; lanes 16 G .. 16 G + 15 of a half take columns 16 G ..; the half h takes
    // rows 16 s + 4 h .. + 3 and 16 s + 8 + 4 h .. + 3.  B operands = E and D as they stand.  Result tile: the own node on
    // the lane, features (k & 3) + 8 (k >> 2) + 4 h in the registers, already in the result row's unit: each group of
    // four registers is one 8-byte piece of the row's two planes.
    const int tq = (lane & 15) >> 2, tpp = lane & 3, tG = (lane >> 4) & 1;
    const int tsw = (2 * h + (tq >> 1)) & 7;                                   // (row >> 1) & 7 of the first block's row
    const int trow0 = (4 * h + tq) * 128 + (((2 * tG + (tpp >> 1)) ^ tsw) << 4) + 8 * (tpp & 1);
    const int trow1 = (4 * h + tq + 8) * 128 + (((2 * tG + (tpp >> 1)) ^ tsw ^ 4) << 4) + 8 * (tpp & 1);
    const int uoff = r * (int)(A.ldus * 2) + 8 * h;                            // this lane's row inside a 32-row tile, its 8 bytes
    MSP_T(3);
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        if (c + 2 < NCH) {
            const int nxt = stg == 0 ? 2 : stg - 1;
            MSP_DMA(base + nxt * STAGE, c + 2);
        }
        if (c == 3) MSP_T(5);
        if (c + 2 < NCH) MSP_WAIT_2();
        else if (c + 1 < NCH) MSP_WAIT_1();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (c == 3) MSP_T(6);
        const char *st = base + stg * STAGE;
        stg = stg == NST - 1 ? 0 : stg + 1;
        f16x8 Xt[2][TM][2][2];                                                 // [graph][tile][k step][piece]
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < TT[g]; ++t)
#pragma unroll
                for (int s = 0; s < MSP_KS(t); ++s) {
                    const int o0 = g * GRAPH + (32 * t + 16 * s) * 128;
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        const s16x4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (msp_lds_s16x4 *)(st + ((o0 + trow0) ^ (64 * pc))));
                        const s16x4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (msp_lds_s16x4 *)(st + ((o0 + trow1) ^ (64 * pc))));
                        const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
                        Xt[g][t][s][pc] = f16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
                    }
                }
        // the tiles (d, to) of this chunk as a pipeline: the product of tile i + 1 is issued in front of the piece split
        // and the stores of tile i, and the scheduler is told to alternate them
        f32x16s Z[2];
        auto product = [&](int i) {
            const int d = i < T1 ? 0 : 1, to = i < T1 ? i : i - T1, gp = 1 - d;
            f32x16s z;
#pragma unroll
            for (int k = 0; k < 16; ++k) z[k] = 0.f;
#pragma unroll
            for (int s = 0; s < MSP_KS(to); ++s) {
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Xt[d][to][s][0], Df[d][to][s], z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Xt[d][to][s][1], Df[d][to][s], z, 0, 0, 0);
            }
#pragma unroll
            for (int tp = 0; tp < TT[gp]; ++tp)
#pragma unroll
                for (int s = 0; s < MSP_KS(tp); ++s) {
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Xt[gp][tp][s][0], Ef[d][to][tp][s][0], z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Xt[gp][tp][s][1], Ef[d][to][tp][s][0], z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_f16(Xt[gp][tp][s][0], Ef[d][to][tp][s][1], z, 0, 0, 0);
                }
            return z;
        };
        Z[0] = product(0);
#pragma unroll
        for (int i = 0; i < T1 + T2; ++i) {
            const int d = i < T1 ? 0 : 1, to = i < T1 ? i : i - T1;
            if (i + 1 < T1 + T2) Z[(i + 1) & 1] = product(i + 1);
            const f32x16s z = Z[i & 1];
            const int ub = uoff + to * 32 * (int)(A.ldus * 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4v v, rs;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = z[4 * q + j] * 0.0009765625f;                // 2^-10
                const f16x4 h1 = __builtin_convertvector(v, f16x4);
#pragma unroll
                for (int j = 0; j < 4; ++j) rs[j] = fmaf((float)h1[j], -1.f, v[j]);
                const f16x4 h2 = __builtin_convertvector(rs, f16x4);
                // features 8 q + 4 h .. + 3 of chunk c: 8 bytes of the high plane, 8 of the low plane 64 bytes on
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v, h1), RU[d], ub + 16 * q, c * 128, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v, h2), RU[d], ub + 16 * q + 64, c * 128, 0);
            }
            if (i + 1 < T1 + T2) {
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                }
            }
        }
    }
    MSP_T(4);
}

template <int R, bool MAP>
__global__ void __launch_bounds__(64, 1) k_gmn_match_sp(const MatchSpArgs A) {
    extern __shared__ float4 msp_smem[];
    const int lane = threadIdx.x;
    const int item = blockIdx.x;
    const int pair = A.list[item];
    const int s0 = A.seg[2 * pair], s1 = A.seg[2 * pair + 1], s2 = A.seg[2 * pair + 2];
    const int n0 = s1 - s0, n1 = s2 - s1;
    if (n0 > R || n1 > R || (n0 | n1) <= 0) return;                            // another size class / empty pair
#ifdef MSP_STAMP
    const bool stamp_on = item == (int)gridDim.x / 2 + 1;
#else
    const bool stamp_on = false;
#endif
    MSP_T(0);
    char *base = reinterpret_cast<char *>(msp_smem);
    // (an empty graph runs as one tile of rows that are all beyond the resource: zeros, masked in the softmax)
    if (R <= 32) msp_pair<1, 1, R, MAP>(A, base, lane, s0, s1, s2, stamp_on);
    else if (n0 > 32 && n1 > 32) msp_pair<2, 2, R, MAP>(A, base, lane, s0, s1, s2, stamp_on);
    else if (n0 > 32) msp_pair<2, 1, R, MAP>(A, base, lane, s0, s1, s2, stamp_on);
    else if (n1 > 32) msp_pair<1, 2, R, MAP>(A, base, lane, s0, s1, s2, stamp_on);
    else msp_pair<1, 1, R, MAP>(A, base, lane, s0, s1, s2, stamp_on);
}

// fp32 rows -> sp format for the rows of the listed pairs only (the pairs the fp32 attention kernels took)
__global__ void __launch_bounds__(256) k_split_sp_pairs(const float *__restrict__ x, int64_t ldx, const int32_t *__restrict__ seg,
                                                        const int32_t *__restrict__ list, int n_list, int KB,
                                                        _Float16 *__restrict__ xs, int64_t ldxs, uint8_t *__restrict__ xe,
                                                        int64_t ldxe) {
    // grid (pair, part): SPLIT_PARTS workgroups share a pair's (row, block) units -- a pair of this class has 70-300 rows, and
    // one workgroup walking them four at a time is a chain of ~50 dependent round trips (113 us per launch for 415 pairs)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int pair = list[blockIdx.x];
    const int m0 = seg[2 * pair], m1 = seg[2 * pair + 2];
    for (int64_t unit = (int64_t)m0 * KB + blockIdx.y * 4 + wid; unit < (int64_t)m1 * KB; unit += 4 * gridDim.y) {
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

template <int R, bool MAP>
static int launch_match_sp(const MatchSpArgs &A, hipStream_t s) {
    typedef MatchSp<R> W;
    static bool declared = false;
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gmn_match_sp<R, MAP>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, W::BYTES) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "gmn_match_sp: cannot raise the dynamic LDS limit");
        declared = true;
    }
    hipLaunchKernelGGL((k_gmn_match_sp<R, MAP>), dim3(A.n_list), dim3(64), W::BYTES, s, A);
    return launch_status("gmn_match_sp");
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_gmn_match_sp(const int32_t *seg, int32_t num_pairs, const void *xs, int64_t ldxs, const uint8_t *xe,
                                  int64_t ldxe, const int32_t *row_map, int64_t x_rows, void *us, int64_t ldus, uint8_t *ue,
                                  int64_t ldue, int32_t d, const int32_t *pair_order, const int32_t *class_counts_host,
                                  int32_t n_le48, void *stream) {
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
    BLGNN_CHECK_ARG(n_le48 >= 0 && n_le48 <= class_counts_host[1], "gmn_match_sp: n_le48 exceeds the <= 64-node class");
    if (row_map && x_rows * ldxs * 2 > 0xfffff000ll)
        return fail(BLGNN_ERR_UNSUPPORTED, "gmn_match_sp: mapped x planes must stay below 4 GiB (32-bit buffer offsets)");
    if (ldxs * 2 * 128 >= ((int64_t)1 << 31) || ldus * 2 * 128 >= ((int64_t)1 << 31))
        return fail(BLGNN_ERR_UNSUPPORTED, "gmn_match_sp: plane rows too long (32-bit offsets inside a pair)");
    MatchSpArgs A;
    A.seg = seg; A.xs = reinterpret_cast<const _Float16 *>(xs); A.ldxs = ldxs; A.xe = xe; A.ldxe = ldxe;
    A.row_map = row_map; A.x_rows = x_rows;
    A.us = reinterpret_cast<_Float16 *>(us); A.ldus = ldus; A.ue = ue; A.ldue = ldue;
    hipStream_t s = (hipStream_t)stream;
    int rc = BLGNN_OK;
    if (class_counts_host[0] > 0) {
        A.list = pair_order; A.n_list = class_counts_host[0];
        rc = row_map ? launch_match_sp<32, true>(A, s) : launch_match_sp<32, false>(A, s);
        if (rc != BLGNN_OK) return rc;
    }
    // the <= 64-node class: its first n_le48 listed pairs (both graphs <= 48 nodes) with 48-row chunk images -- 37 KB of
    // LDS, four wavefronts per CU instead of three --, the rest with 64-row images
    if (n_le48 > 0) {
        A.list = pair_order + class_counts_host[0]; A.n_list = n_le48;
        rc = row_map ? launch_match_sp<48, true>(A, s) : launch_match_sp<48, false>(A, s);
        if (rc != BLGNN_OK) return rc;
    }
    if (class_counts_host[1] > n_le48) {
        A.list = pair_order + class_counts_host[0] + n_le48; A.n_list = class_counts_host[1] - n_le48;
        rc = row_map ? launch_match_sp<64, true>(A, s) : launch_match_sp<64, false>(A, s);
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
    hipLaunchKernelGGL(k_split_sp_pairs, dim3(n_list, 8), dim3(256), 0, (hipStream_t)stream, x, ldx, seg, pair_list, n_list,
                       K / SP_BLOCK, reinterpret_cast<_Float16 *>(xs), ldxs, xe, ldxe);
    return launch_status("split_rows_sp_pairs");
}
