// f16x3 GEMM on PRE-SPLIT operands ("sp" format, sp_common.hpp): the GMN propagator's GEMMs over all atoms
// (model/layers_gmn.py:19-26,39-47: f_message's first Linear, f_node's two Linears).
//
//   Y = epi( X[M,K] @ W[K,N] + bias + rowvec (x) rowvec_w )      epi = act, then affine
//
// with X given as fp16 piece planes + per-(row, 256-column block) power-of-two scales and Y written as fp32
// rows, as the same sp format (for the next GEMM), or both.
//
// Why a second f16x3 GEMM next to linear_h3.hip (measured there, 462 k x 1024 x 1024): that kernel converts the
// fp32 X tile to two fp16 images INSIDE the GEMM -- four of its eight waves do nothing else, every column block
// of a row panel repeats the conversion, and its knock-outs show the operand streams (not the matrix cores)
// setting the time.  Here the producer of X has already split it, so
//   * both operand tiles of a K step arrive by LDS-DMA (buffer_load ... lds, 16 bytes per lane: no registers, no
//     conversion, no ds_write); the swizzle that makes the fragment reads conflict free is applied by choosing
//     which 16-byte chunk of the 128-byte row segment each lane fetches (SQ_LDS_BANK_CONFLICT = 0);
//   * all eight waves issue MFMAs (two per SIMD, 64 x 128 outputs each), the tile is 256 x 256 at K step 32:
//     64 KB of operands per 12.6 MFLOP-x3 instead of 48 KB per 6.3;
//   * the row-magnitude passes (blgnn_row_absmax, atomic row maxima in epilogues) disappear: scales travel with
//     the data.
// Block scales inside one accumulator: the accumulators hold sum_k x~ w~ in units of the CURRENT block's scale;
// at a block boundary (every 8 K steps) they are multiplied by the exact power of two u_next / u_cur.  u follows
// the blocks' scales but never exceeds 2^40 x the smallest scale seen (a block 2^40 below the row's largest
// contributes < 2^-40 of the dot product whatever is done with it), so nothing overflows on zero blocks or on
// rows spanning the whole fp32 range.
// Operand roles: W fragments are the MFMA A operand and X fragments the B operand, so a lane owns ROWS (one per
// 16-row block: scale bookkeeping = one register pair per row) and four consecutive columns per accumulator
// (epilogue parameters and stores are float4 wide).
//
// What bounds it (in-kernel stamps + PMC, K = N = 1024, 478 k rows): power.  The main loop
// runs 3 500-3 800 cycles per K step against 3 072 of MFMA issue, but at an in-kernel clock of 1.5-1.8 GHz (1.9-2.1
// with the operand requests knocked out, 2.4 with the MFMAs knocked out).  Measured on the way here (us per launch,
// K = N = 1024 / K = 256, N = 1024 / K = 1024, N = 256; linear_h3.hip: 3 590 / 1 335 / 1 200):
//   v_mfma_f32_32x32x16_f16, float scales                     2 735 / 1 030 /   905   (1.5 GHz)
//   v_mfma_f32_16x16x32_f16 (same cycles per flop, holds a higher clock)
//                                                             2 630 / 1 017 /   880   (1.8 GHz)
//   + block scales as exponent bytes (one 8-byte load per row) 2 591 / 1 002 /   860
// Tried and dropped: a 5-slot ring at K step 16 (four steps in flight instead of one: 2 906 -- the operand streams
// are throughput-, not latency-bound: 35 GB/s per CU whatever is in flight); column blocks in passes so that an
// XCD's L2 holds 1 or 2 of the 4 weight slices (no change); non-temporal X loads (no change); padded row strides
// (no change); staggered first-round starts to spread the epilogues' store bursts (2 790); static priority for
// waves 4-7 (no change); the next step's LDS-DMA issues spread between the MFMA groups (4 900 cycles per step: the
// compiler orders LDS-DMA against ds_read and waits for every fragment read first); a persistent form (one
// workgroup per CU walking its tiles, next tile's first stage and row parameters requested before the epilogue:
// 2 746 / 1 046 -- static tile assignment loses more than the hidden round trips gain); at K = 256 (a tile = 2.9 us of
// prologue + 18.5 us of main loop + 7.5 us of epilogue): fp32 rows stored straight from the accumulators, no LDS
// staging (the stamped tile's epilogue 7.8 -> 6.0 us, the launch 1 016 vs 1 022 us: no change) and non-temporal row
// stores (no change).
#include <cstdlib>
#include "linear_common.hpp"
#include "sp_common.hpp"

namespace blgnn {

struct LinearSpArgs {
    const _Float16 *xs; int64_t ldxs;      // sp planes of X: fp16 elements per row (2 K for a dense matrix)
    const uint8_t *xe; int64_t ldxe;       // block scale exponents [M][K / 256] (8-byte aligned rows: ldxe % 8 == 0)
    const _Float16 *wp;                    // blgnn_pack_weight_sp image
    float *y; int64_t ldy;                 // fp32 output (nullable)
    _Float16 *ys; int64_t ldys;            // sp output (nullable) ...
    uint8_t *ye; int64_t ldye;             // ... and its block scale exponents [M][N / 256]
    int32_t M, N, K;
    const float *rowvec, *rowvec_w;
    Epilogue epi;
    const float *addend; int64_t ldadd;            // optional: + addend[addend_rows ? addend_rows[m] : m, n] in front of the activation
    const int32_t *addend_rows; int32_t n_addend;  //   (n_addend = rows of ``addend``)
    // optional fused node -> graph readout of the RESULT (model/layers_aggregation.py:16-19,32-41):
    // pool[row_graph[m], n] += y[m, n] * (pool_mean ? 1 / rows of that graph : 1); rows sorted by graph
    const int32_t *row_graph, *seg; float *pool; int64_t ldp; int32_t pool_mean;
};

constexpr int SP_BM = 256, SP_BN = 256, SP_BK = 32;
constexpr int SP_ROWB = 128;                       // LDS row: [h1 x 32 | h2 x 32] fp16 = eight 16-byte chunks
constexpr int SP_XST = SP_BM * SP_ROWB;            // 32 KB
constexpr int SP_WST = SP_BN * SP_ROWB;            // 32 KB
constexpr int SP_STAGE = SP_XST + SP_WST;          // 64 KB
constexpr int SP_OSTRIDE = 528;                    // epilogue staging row: 512 B + 16
constexpr int SP_OWAVE = 32 * SP_OSTRIDE;          // one wave's staging tile (32 rows x 128 columns)
constexpr int SP_OSTG = 8 * SP_OWAVE;              // 132 KB >= two operand stages (128 KB); behind it:
constexpr int SP_NPAR = 5;                         // bias, rowvec_w, bn scale, bn shift, PReLU slope
constexpr int SP_LDS = SP_OSTG + SP_NPAR * SP_BN * 4 + SP_BM * 4;
static_assert(SP_OSTG >= 2 * SP_STAGE, "staging area must cover the operand stages");

#ifdef SP_STAMP
__device__ unsigned long long g_sp_stamps[8];      // diagnostics: shader cycles / 100 MHz ticks around the main loop
#endif
typedef __attribute__((address_space(3))) void sp_lds_void;
typedef __attribute__((address_space(1))) void sp_glb_void;

// ---- weight packing: W[K,N] fp32 -> header + [N / 256][K / 32] tile images of 32 KB --------------------------
// tile image = the LDS image of (column block, K step): [column 256][slot 8][8 fp16], slot = chunk ^ ((column >> 1) & 7),
// chunks 0-3 = high pieces of k = 0..31, chunks 4-7 = low pieces; scaled by the power of two that moves the
// largest |w| into [2^14, 2^15) (inverse in the header); zero filled past K.
__global__ void k_pack_sp(const float *__restrict__ w, int64_t ldw, int K, int N, int KS,
                          const unsigned *__restrict__ amax_bits, _Float16 *__restrict__ out) {
    float scale, inv;
    pow2_scale(__uint_as_float(*amax_bits), scale, inv);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one (column, 8-k group): high AND low word
    if (idx == 0) *reinterpret_cast<float *>(out) = inv;
    if (idx >= (int64_t)(N / SP_BN) * KS * SP_BN * 4) return;
    const int kc = (int)(idx & 3), r = (int)((idx >> 2) & (SP_BN - 1));
    const int64_t tile = idx >> 10;
    const int ks = (int)(tile % KS), nb = (int)(tile / KS);
    const int n = nb * SP_BN + r, sw = (r >> 1) & 7;
    f16x8 p[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ks * SP_BK + kc * 8 + j;
        const Split2 sp = split2(k < K ? w[(int64_t)k * ldw + n] * scale : 0.f);
        p[0][j] = sp.h1; p[1][j] = sp.h2;
    }
    _Float16 *base = out + HHDR + tile * (SP_WST / 2) + r * 64;
    *reinterpret_cast<f16x8 *>(base + ((kc ^ sw) << 3)) = p[0];
    *reinterpret_cast<f16x8 *>(base + (((kc + 4) ^ sw) << 3)) = p[1];
}

__global__ void k_absmax_sp(const float *__restrict__ w, int64_t ldw, int K, int N, unsigned *out) {
    float m = 0.f;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)K * N;
         idx += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[(idx / N) * ldw + idx % N]));
    m = group_max(m, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(out, __float_as_uint(m));
}

// ---- fp32 rows -> sp format (for producers that do not write it themselves): one wave per (row, block) ---------
__global__ void __launch_bounds__(256) k_split_sp(const float *__restrict__ x, int64_t ldx, int M, int KB,
                                                  _Float16 *__restrict__ xs, int64_t ldxs,
                                                  uint8_t *__restrict__ xe, int64_t ldxe) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= (int64_t)M * KB) return;
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

// ---- sp format -> fp32 rows (tests, and consumers outside the GEMM) ------------------------------------------
__global__ void __launch_bounds__(256) k_join_sp(const _Float16 *__restrict__ xs, int64_t ldxs,
                                                 const uint8_t *__restrict__ xe, int64_t ldxe, int M, int KB,
                                                 float *__restrict__ x, int64_t ldx) {
    const int lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= (int64_t)M * KB) return;
    const int m = (int)(unit / KB), kb = (int)(unit % KB);
    const float inv = sp_inv_of(xe[(int64_t)m * ldxe + kb]);
    const int col = lane * 4;
    const _Float16 *src = xs + (int64_t)m * ldxs + kb * (2 * SP_BLOCK) + (col >> 5) * 64 + (col & 31);
    const f16x4 h1 = *reinterpret_cast<const f16x4 *>(src), h2 = *reinterpret_cast<const f16x4 *>(src + 32);
    float4 o;
    o.x = ((float)h1[0] + (float)h2[0]) * inv; o.y = ((float)h1[1] + (float)h2[1]) * inv;
    o.z = ((float)h1[2] + (float)h2[2]) * inv; o.w = ((float)h1[3] + (float)h2[3]) * inv;
    *reinterpret_cast<float4 *>(x + (int64_t)m * ldx + kb * SP_BLOCK + col) = o;
}

// ---- the GEMM --------------------------------------------------------------------------------------------------
// The same GEMM on v_mfma_f32_16x16x32_f16: one MFMA per K step of 32 and 16 x 16 block (96 per wave and step, 16 cycles
// each: the cycles of the 32 x 32 x 16 form).  Why: the loop is power-limited (in-kernel clock 1.5 GHz with both operand
// streams running, 1.9 GHz without them, 2.4 GHz without the MFMAs), and the 16 x 16 shape holds a higher clock at equal
// cycles per flop.  A lane owns FOUR rows (one per 16-row block) and, per accumulator, four consecutive columns.
// (Tried and dropped: a persistent form -- one workgroup per CU walking its tiles, the next tile's first operand stage
// and row parameters requested before the current tile's epilogue, 16-row staging passes out of the stage-1 region --
// to hide the ~14 us of a tile's 90 that are neither main loop nor epilogue: 2 746 vs 2 591 us at K = N = 1024, 1 046 vs
// 1 002 at K = 256: static tile assignment loses more than the hidden round trips gain.)
template <bool GEN, bool ADD = false, bool POOL = false>
__global__ void __launch_bounds__(512, 2) k_linear_sp(const LinearSpArgs P) {
    extern __shared__ float4 sp_smem[];
    char *smem = reinterpret_cast<char *>(sp_smem);
    float *cpar = reinterpret_cast<float *>(smem + SP_OSTG);                  // [SP_NPAR][256]   (epilogue only)
    unsigned *rowmax = reinterpret_cast<unsigned *>(cpar + SP_NPAR * SP_BN);   // [256] bits of max |y| per row

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform by construction: keep it scalar
#ifdef SP_STAMP
    if (blockIdx.x == gridDim.x / 2 + 3 && tid == 0) { g_sp_stamps[6] = __builtin_amdgcn_s_memtime(); g_sp_stamps[7] = __builtin_amdgcn_s_memrealtime(); }
#endif
    const int nb = P.N / SP_BN;
    int tile;
    {   // blocks b, b + 8, ... share an XCD: give each XCD a contiguous run of tiles (row-panel major), so the
        // column blocks of one row panel read it from the same L2
        const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = (tile / nb) * SP_BM, cb = tile % nb, n0 = cb * SP_BN;
    const int KS = P.K / SP_BK;

    // ---- LDS-DMA plan: per K step this wave brings 32 X rows and 32 W columns (4 + 4 pieces of 1 KB) ----------
    // piece i of X: rows 32 wid + 8 i + (lane >> 3), LDS slot lane & 7 <- global chunk slot ^ ((row >> 1) & 7).
    // Buffer addressing: wave-uniform parts (tile base, row group, K step) ride in the resource / the scalar
    // offset, the lane's part is one 32-bit register for even and one for odd pieces; rows past M lie beyond
    // num_records and arrive as zeros.
    const int64_t rowb = P.ldxs * 2;
    const int rows_in = min(SP_BM, P.M - m0);
    const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(P.xs) + (int64_t)m0 * rowb), 0,
        (int)((int64_t)(rows_in - 1) * rowb + (int64_t)P.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t RW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char *>(reinterpret_cast<const char *>(P.wp + HHDR) + (size_t)cb * KS * SP_WST), 0,
        KS * SP_WST, 0x00020000);
    const int xv0 = (lane >> 3) * (int)rowb + (((lane & 7) ^ (lane >> 4)) << 4);
    const int xv1 = (lane >> 3) * (int)rowb + (((lane & 7) ^ (lane >> 4) ^ 4) << 4);
    const int wv = lane * 16;
    const int xs0 = 32 * wid * (int)rowb;
    auto dma_piece = [&](int ks, int p) {                   // piece p of K step ks: 0..3 X rows, 4..7 W columns
        char *st = smem + (ks & 1) * SP_STAGE;
        if (p < 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RX, (sp_lds_void *)(st + (32 * wid + 8 * p) * SP_ROWB), 16,
                                                     (p & 1) ? xv1 : xv0, xs0 + 8 * p * (int)rowb + ks * SP_ROWB, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(RW, (sp_lds_void *)(st + SP_XST + wid * 4096 + (p - 4) * 1024), 16, wv,
                                                     ks * SP_WST + wid * 4096 + (p - 4) * 1024, 0, 0);
        }
    };
    auto dma = [&](int ks) {
#pragma unroll
        for (int p = 0; p < 8; ++p) dma_piece(ks, p);
    };

    // ---- this wave's 64 x 128 outputs: row blocks im (4 x 16 rows), column blocks jn (8 x 16 columns) -----------
    const int r16 = lane & 15, kg = lane >> 4, wm = wid >> 1, wn = wid & 1;
    const int sw = (r16 >> 1) & 7;                               // 16-row offsets keep (row >> 1) & 7
    const int xrow = (wm * 64 + r16) * SP_ROWB, wrow = SP_XST + (wn * 128 + r16) * SP_ROWB;
    const int c1 = (kg ^ sw) << 4, c2 = ((4 + kg) ^ sw) << 4;   // this lane's 8 k of the high / low piece
    f32x4v acc[4][8];
#pragma unroll
    for (int im = 0; im < 4; ++im)
#pragma unroll
        for (int jn = 0; jn < 8; ++jn) acc[im][jn] = f32x4v{0.f, 0.f, 0.f, 0.f};

    dma(0);                                                 // K step 0 requested before anything else

    // epilogue parameters of this thread's column: fetched now (behind the ring prologue's requests), parked in
    // registers (the ring owns all of LDS), written to LDS after the last K step
    float par[SP_NPAR] = {0.f, 0.f, 1.f, 0.f, 0.f};
    if (tid < SP_BN) {
        const int n = n0 + tid;
        if (P.epi.bias) par[0] = P.epi.bias[n];
        if (P.rowvec_w) par[1] = P.rowvec_w[n];
        if (P.epi.bn_scale) par[2] = P.epi.bn_scale[n];
        if (P.epi.bn_scale && P.epi.bn_shift) par[3] = P.epi.bn_shift[n];
        if (P.epi.act == BLGNN_ACT_PRELU && P.epi.act_param) par[4] = P.epi.act_param[n];
    }

    // scale bookkeeping per owned row: the blocks' inverse scales are powers of two; their biased exponent fields
    // (8 bits each, K <= 2048: at most 8 blocks) are packed into one 64-bit word per row, so the main loop issues
    // no vector-memory instruction but the ring's -- a load the loop waited on would drain the ring
    const int KB = P.K / SP_BLOCK;
    unsigned long long epack[4];
    int eiu[4], eimax[4];
    float rv[4];
    unsigned aoff[4] = {0u, 0u, 0u, 0u};                    // ADD: byte offset of this lane's four columns in its rows' addend rows
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        const int mr = m0 + wm * 64 + 16 * im + r16;
        const int m = min(mr, P.M - 1);
        // the row's block exponents: one 8-byte load (bytes past KB belong to other blocks and are never consulted)
        const unsigned long long pk = *reinterpret_cast<const unsigned long long *>(P.xe + (int64_t)m * P.ldxe);
        epack[im] = pk;
        eiu[im] = eimax[im] = (int)(pk & 0xffu);
        rv[im] = P.rowvec ? P.rowvec[m] : 0.f;
        if (ADD) aoff[im] = (unsigned)(P.addend_rows ? P.addend_rows[m] : m) * (unsigned)(P.ldadd * 4) + (unsigned)kg * 16u;
    }
    int gl = -1;                                            // POOL: graph of row m0 + 64 wm + lane (-1 past M)
    if (POOL) {
        const int m = m0 + wm * 64 + lane;
        gl = m < P.M ? P.row_graph[m] : -1;
    }
    float winv = *reinterpret_cast<const float *>(P.wp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // K step 0 has landed (and the plain loads above)
    // the compiler does not see that wait: pin every loaded value here, or it places its own full wait at the
    // first use -- inside the loop (draining the ring at every block boundary) or in front of the epilogue
#pragma unroll
    for (int im = 0; im < 4; ++im) {
        unsigned lo = (unsigned)epack[im], hi = (unsigned)(epack[im] >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi), "+v"(rv[im]));
        if (ADD) asm volatile("" : "+v"(aoff[im]));
        epack[im] = ((unsigned long long)hi << 32) | lo;
    }
#pragma unroll
    for (int j = 0; j < SP_NPAR; ++j) asm volatile("" : "+v"(par[j]));
    asm volatile("" : "+v"(winv));
    if (POOL) asm volatile("" : "+v"(gl));
    lds_barrier();

#ifdef SP_STAMP
    const bool stamp_on = blockIdx.x == gridDim.x / 2 + 3 && tid == 0;
    if (stamp_on) { g_sp_stamps[0] = __builtin_amdgcn_s_memtime(); g_sp_stamps[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    for (int ks = 0; ks < KS; ++ks) {
        const char *cur = smem + (ks & 1) * SP_STAGE;
#ifndef SP_KNOCK_DMA
        if (ks + 1 < KS) dma(ks + 1);                  // the other stage: last read during step ks - 1
#endif
        if ((ks & 7) == 0 && ks > 0) {
            // block boundary: re-express the accumulators in the next block's unit
            const int kb = ks >> 3;
#pragma unroll
            for (int im = 0; im < 4; ++im) {
                const int ei = (int)((epack[im] >> (8 * kb)) & 0xffu);
                eimax[im] = max(eimax[im], ei);
                const int enew = max(ei, eimax[im] - 40);
                const int d = max(eiu[im] - enew, -126);
                eiu[im] = enew;
                const float ratio = __uint_as_float((unsigned)(127 + d) << 23);
#pragma unroll
                for (int jn = 0; jn < 8; ++jn) acc[im][jn] *= ratio;
            }
        }
        f16x8 a[8][2], b[4][2];
#pragma unroll
        for (int im = 0; im < 4; ++im) {
            b[im][0] = *reinterpret_cast<const f16x8 *>(cur + xrow + c1 + 16 * im * SP_ROWB);
            b[im][1] = *reinterpret_cast<const f16x8 *>(cur + xrow + c2 + 16 * im * SP_ROWB);
        }
#pragma unroll
        for (int jn = 0; jn < 8; ++jn) {
            a[jn][0] = *reinterpret_cast<const f16x8 *>(cur + wrow + c1 + 16 * jn * SP_ROWB);
            a[jn][1] = *reinterpret_cast<const f16x8 *>(cur + wrow + c2 + 16 * jn * SP_ROWB);
        }
        // column-block major, the small terms first: a W fragment pair serves four row blocks and retires.
        // (Tried: all 24 fragment reads first and the next step's eight LDS-DMA issues spread between the first MFMA
        // groups instead of at the head of the step -- 4 900 instead of 3 830 cycles per step: the compiler orders
        // LDS-DMA against ds_read and waits for every read in front of the first request.)
#ifndef SP_KNOCK_MFMA
#pragma unroll
        for (int jn = 0; jn < 8; ++jn) {
#pragma unroll
            for (int im = 0; im < 4; ++im)
                acc[im][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jn][1], b[im][0], acc[im][jn], 0, 0, 0);
#pragma unroll
            for (int im = 0; im < 4; ++im)
                acc[im][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jn][0], b[im][1], acc[im][jn], 0, 0, 0);
#pragma unroll
            for (int im = 0; im < 4; ++im)
                acc[im][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[jn][0], b[im][0], acc[im][jn], 0, 0, 0);
        }
#else
        acc[0][0][0] += (float)a[0][0][0] + (float)b[0][0][0];
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of step ks + 1 have landed
        lds_barrier();                                      // stage ks & 1 fully read, stage (ks + 1) & 1 complete
    }

#ifdef SP_STAMP
    if (stamp_on) { g_sp_stamps[2] = __builtin_amdgcn_s_memtime(); g_sp_stamps[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
    // park the epilogue parameters behind the staging tiles
    if (tid < SP_BN) {
#pragma unroll
        for (int j = 0; j < SP_NPAR; ++j) cpar[j * SP_BN + tid] = par[j];
        rowmax[tid] = 0u;
    }
    lds_barrier();

    // ================================================ epilogue ================================================
    // lane (r16, kg) of block (im, jn) holds row 16 im + r16, columns 16 jn + 4 kg + (0..3)
    const int act = P.epi.act;
    const bool relu = act == BLGNN_ACT_RELU;                   // GEN = false: relu or identity
    float rsc[4];
#pragma unroll
    for (int im = 0; im < 4; ++im) rsc[im] = __uint_as_float((unsigned)eiu[im] << 23) * winv;
    // straight-line: the staged parameters hold neutral values where the caller passed none (scale 1, shift 0,
    // rank-1 factor 0); row blocks innermost, so a column group's parameters are read once
    if constexpr (ADD) {
        // + addend rows (gathered by addend_rows) in front of the activation: this lane's 4 x 8 groups of four
        // columns, fetched in four groups of two column blocks, two groups in flight (the operand fragments'
        // registers are free now); a 16-lane quarter of a wavefront reads 64 contiguous bytes of 16 rows per request
        const __amdgpu_buffer_rsrc_t RA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(P.addend), 0, (int)((int64_t)(P.n_addend - 1) * P.ldadd * 4 + (int64_t)P.N * 4), 0x00020000);
        f32x4v ad[2][2][4];
        auto fetch = [&](int g, f32x4v (&dst)[2][4]) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int im = 0; im < 4; ++im)
                    dst[j][im] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(
                        RA, aoff[im], (n0 + wn * 128 + 16 * (2 * g + j)) * 4, 0));
        };
        fetch(0, ad[0]);
        fetch(1, ad[1]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int jn = 2 * g + j;
                const int nl = wn * 128 + 16 * jn + 4 * kg;
                const float4 b4 = *reinterpret_cast<const float4 *>(cpar + nl);
                const float4 w4 = *reinterpret_cast<const float4 *>(cpar + SP_BN + nl);
                const float4 s4 = *reinterpret_cast<const float4 *>(cpar + 2 * SP_BN + nl);
                const float4 t4 = *reinterpret_cast<const float4 *>(cpar + 3 * SP_BN + nl);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
                const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, sh[4] = {t4.x, t4.y, t4.z, t4.w};
                float sl[4] = {0.f, 0.f, 0.f, 0.f};
                if (GEN) {
                    const float4 p4 = *reinterpret_cast<const float4 *>(cpar + 4 * SP_BN + nl);
                    sl[0] = p4.x; sl[1] = p4.y; sl[2] = p4.z; sl[3] = p4.w;
                }
#pragma unroll
                for (int im = 0; im < 4; ++im)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = fmaf(acc[im][jn][q], rsc[im], ad[g & 1][j][im][q]);
                        v = fmaf(rv[im], ww[q], v + bb[q]);
                        if (GEN) v = apply_act(v, act, sl[q]);
                        else v = (relu && !(v > 0.f)) ? 0.f : v;
                        acc[im][jn][q] = fmaf(v, sc[q], sh[q]);
                    }
            }
            if (g + 2 < 4) fetch(g + 2, ad[g & 1]);
        }
    } else
#pragma unroll
    for (int jn = 0; jn < 8; ++jn) {
        const int nl = wn * 128 + 16 * jn + 4 * kg;
        const float4 b4 = *reinterpret_cast<const float4 *>(cpar + nl);
        const float4 w4 = *reinterpret_cast<const float4 *>(cpar + SP_BN + nl);
        const float4 s4 = *reinterpret_cast<const float4 *>(cpar + 2 * SP_BN + nl);
        const float4 t4 = *reinterpret_cast<const float4 *>(cpar + 3 * SP_BN + nl);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, ww[4] = {w4.x, w4.y, w4.z, w4.w};
        const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, sh[4] = {t4.x, t4.y, t4.z, t4.w};
        float sl[4] = {0.f, 0.f, 0.f, 0.f};
        if (GEN) {
            const float4 p4 = *reinterpret_cast<const float4 *>(cpar + 4 * SP_BN + nl);
            sl[0] = p4.x; sl[1] = p4.y; sl[2] = p4.z; sl[3] = p4.w;
        }
#pragma unroll
        for (int im = 0; im < 4; ++im)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = acc[im][jn][q] * rsc[im];
                v = fmaf(rv[im], ww[q], v + bb[q]);
                if (GEN) v = apply_act(v, act, sl[q]);
                else v = (relu && !(v > 0.f)) ? 0.f : v;
                acc[im][jn][q] = fmaf(v, sc[q], sh[q]);
            }
    }
    // every wave passed the barrier that ended the last K step: the operand stages are free.  Wave-private
    // staging tile of 32 rows x 128 columns (LDS operations of one wave complete in order: no barrier between its
    // writes and reads): row blocks 2 ib and 2 ib + 1 per pass.
    char *stg = smem + wid * SP_OWAVE;
    if (P.y || POOL) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int jn = 0; jn < 8; ++jn) {
                    const f32x4v v = acc[2 * ib + h][jn];
                    *reinterpret_cast<float4 *>(stg + (16 * h + r16) * SP_OSTRIDE + (16 * jn + 4 * kg) * 4) =
                        make_float4(v[0], v[1], v[2], v[3]);
                }
            const int mb = m0 + wm * 64 + 32 * ib;
            if (P.y) {
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    const int t = lane + 64 * it, row = t >> 5, c4 = t & 31;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + row * SP_OSTRIDE + c4 * 16);
                    if (mb + row < P.M)
                        *reinterpret_cast<float4 *>(P.y + (int64_t)(mb + row) * P.ldy + n0 + wn * 128 + c4 * 4) = v;
                }
            }
            if constexpr (POOL) {
                // readout of the staged 32 rows x 128 columns: a lane owns columns lane and lane + 64 and walks the rows
                // (sorted by graph: a tile touches few graphs); one atomic per (graph, column) and pass.  Graph ids are
                // wave-uniform (v_readlane), so the flush is a scalar branch.
                float s0 = 0.f, s1 = 0.f;
                int gcur = __builtin_amdgcn_readlane(gl, 32 * ib);
                auto flush = [&](int g, float a, float b) {
                    if (g < 0) return;
                    const float w = P.pool_mean ? 1.f / (float)max(P.seg[g + 1] - P.seg[g], 1) : 1.f;
                    float *dst = P.pool + (int64_t)g * P.ldp + n0 + wn * 128 + lane;
                    unsafeAtomicAdd(dst, a * w);
                    unsafeAtomicAdd(dst + 64, b * w);
                };
#pragma unroll
                for (int r0 = 0; r0 < 32; r0 += 8) {
                    // eight rows' values per batch of LDS reads (all 64 at once would not fit next to the accumulators)
                    float v0[8], v1[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        v0[q] = *reinterpret_cast<const float *>(stg + (r0 + q) * SP_OSTRIDE + lane * 4);
                        v1[q] = *reinterpret_cast<const float *>(stg + (r0 + q) * SP_OSTRIDE + (lane + 64) * 4);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int g = __builtin_amdgcn_readlane(gl, 32 * ib + r0 + q);
                        if (g != gcur) { flush(gcur, s0, s1); s0 = s1 = 0.f; gcur = g; }
                        s0 += v0[q]; s1 += v1[q];
                    }
                    asm volatile("" ::: "memory");
                }
                flush(gcur, s0, s1);
            }
        }
    }
    if (!POOL && P.ys) {                                     // (the readout variant writes fp32 rows or nothing: its
        //                                                        accumulators retire pass by pass, the flush branches are cheap)
        // block scale = the row's largest magnitude over the tile's 256 columns (this wave + its column neighbour)
#pragma unroll
        for (int im = 0; im < 4; ++im) {
            float mx = 0.f;
#pragma unroll
            for (int jn = 0; jn < 8; ++jn)
#pragma unroll
                for (int q = 0; q < 4; ++q) mx = fmaxf(mx, fabsf(acc[im][jn][q]));
            mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
            mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
            if (kg == 0) atomicMax(rowmax + wm * 64 + 16 * im + r16, __float_as_uint(mx));
        }
        lds_barrier();
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int im = 2 * ib + h;
                float s, inv;
                pow2_scale(__uint_as_float(rowmax[wm * 64 + 16 * im + r16]), s, inv);
                const int m = m0 + wm * 64 + 16 * im + r16;
                if (wn == 0 && kg == 0 && m < P.M) P.ye[(int64_t)m * P.ldye + (n0 >> 8)] = sp_exp_of(inv);
#pragma unroll
                for (int jn = 0; jn < 8; ++jn) {
                    const f32x4v v = acc[im][jn] * s;
                    const f16x4 h1 = __builtin_convertvector(v, f16x4);
                    const f16x4 h2 = __builtin_convertvector(v - __builtin_convertvector(h1, f32x4v), f16x4);
                    // columns 16 jn + 4 kg .. + 3 of the wave's 128: chunk jn >> 1, position (jn & 1) * 16 + 4 kg
                    char *dst = stg + (16 * h + r16) * SP_OSTRIDE + (jn >> 1) * 128 + ((jn & 1) * 16 + 4 * kg) * 2;
                    *reinterpret_cast<f16x4 *>(dst) = h1;
                    *reinterpret_cast<f16x4 *>(dst + 64) = h2;
                }
            }
            const int mb = m0 + wm * 64 + 32 * ib;
            char *yrow = reinterpret_cast<char *>(P.ys) + ((n0 + wn * 128) >> 5) * 128;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int t = lane + 64 * it, row = t >> 5, pc = t & 31;
                const float4 v = *reinterpret_cast<const float4 *>(stg + row * SP_OSTRIDE + pc * 16);
                if (mb + row < P.M)
                    *reinterpret_cast<float4 *>(yrow + (int64_t)(mb + row) * P.ldys * 2 + pc * 16) = v;
            }
        }
    }
#ifdef SP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (stamp_on) { g_sp_stamps[4] = __builtin_amdgcn_s_memtime(); g_sp_stamps[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

template <bool GEN, bool ADD = false, bool POOL = false>
static int launch_linear_sp_t(const LinearSpArgs &P, hipStream_t s) {
    static bool declared = false;
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_sp<GEN, ADD, POOL>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "linear_sp: cannot raise the dynamic LDS limit to %d bytes", SP_LDS);
        declared = true;
    }
    const int64_t blocks = (int64_t)((P.M + SP_BM - 1) / SP_BM) * (P.N / SP_BN);
    hipLaunchKernelGGL((k_linear_sp<GEN, ADD, POOL>), dim3((unsigned)blocks), dim3(512), SP_LDS, s, P);
    return launch_status("linear_sp");
}

static int launch_linear_sp(const LinearSpArgs &P, hipStream_t s) {
    if (P.pool) return launch_linear_sp_t<false, false, true>(P, s);        // (relu / identity, no addend: checked by the caller)
    if (P.addend)
        return P.epi.act <= BLGNN_ACT_RELU ? launch_linear_sp_t<false, true>(P, s) : launch_linear_sp_t<true, true>(P, s);
    return P.epi.act <= BLGNN_ACT_RELU ? launch_linear_sp_t<false>(P, s) : launch_linear_sp_t<true>(P, s);
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int64_t blgnn_packed_weight_sp_elems(int32_t K, int32_t N) {
    if (K <= 0 || N <= 0 || N % SP_BN != 0) return -1;
    return HHDR + ((int64_t)N / SP_BN) * (((int64_t)K + SP_BK - 1) / SP_BK) * (SP_WST / 2);   // fp16 elements
}

extern "C" int blgnn_pack_weight_sp(const float *w, int64_t ldw, int32_t K, int32_t N, void *out, void *stream) {
    BLGNN_CHECK_ARG(w && out && K > 0 && N > 0 && ldw >= N, "pack_weight_sp: bad args");
    BLGNN_CHECK_ARG(N % SP_BN == 0, "pack_weight_sp: N must be a multiple of 256");
    BLGNN_CHECK_ARG(aligned16(out), "pack_weight_sp: out must be 16-byte aligned");
    const int KS = (K + SP_BK - 1) / SP_BK;
    hipStream_t s = (hipStream_t)stream;
    unsigned *amax = reinterpret_cast<unsigned *>(out) + 1;       // header word 1: scratch for the maximum's bits
    if (hipMemsetAsync(out, 0, HHDR * 2, s) != hipSuccess) return fail(BLGNN_ERR_LAUNCH, "pack_weight_sp: memset failed");
    const int64_t total = (int64_t)K * N;
    hipLaunchKernelGGL(k_absmax_sp, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, s,
                       w, ldw, K, N, amax);
    const int64_t words = (int64_t)(N / SP_BN) * KS * SP_BN * 4;
    hipLaunchKernelGGL(k_pack_sp, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, w, ldw, K, N, KS, amax,
                       reinterpret_cast<_Float16 *>(out));
    return launch_status("pack_weight_sp");
}

extern "C" int blgnn_split_rows_sp(const float *x, int64_t ldx, int32_t M, int32_t K, void *xs, int64_t ldxs,
                                   uint8_t *xe, int64_t ldxe, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && K > 0 && K % SP_BLOCK == 0, "split_rows_sp: K must be a positive multiple of 256");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && xs && xe && ldx >= K && ldxs >= 2 * (int64_t)K && ldxe >= K / SP_BLOCK, "split_rows_sp: bad args");
    BLGNN_CHECK_ARG(ldx % 4 == 0 && aligned16(x) && ldxs % 8 == 0 && aligned16(xs), "split_rows_sp: 16-byte alignment");
    const int KB = K / SP_BLOCK;
    const int64_t units = (int64_t)M * KB;
    hipLaunchKernelGGL(k_split_sp, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, M, KB,
                       reinterpret_cast<_Float16 *>(xs), ldxs, xe, ldxe);
    return launch_status("split_rows_sp");
}

extern "C" int blgnn_join_rows_sp(const void *xs, int64_t ldxs, const uint8_t *xe, int64_t ldxe, int32_t M, int32_t K,
                                  float *x, int64_t ldx, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && K > 0 && K % SP_BLOCK == 0, "join_rows_sp: K must be a positive multiple of 256");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && xs && xe && ldx >= K && ldxs >= 2 * (int64_t)K && ldxe >= K / SP_BLOCK, "join_rows_sp: bad args");
    BLGNN_CHECK_ARG(ldx % 4 == 0 && aligned16(x) && ldxs % 8 == 0 && aligned16(xs), "join_rows_sp: 16-byte alignment");
    const int KB = K / SP_BLOCK;
    const int64_t units = (int64_t)M * KB;
    hipLaunchKernelGGL(k_join_sp, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16 *>(xs), ldxs, xe, ldxe, M, KB, x, ldx);
    return launch_status("join_rows_sp");
}

extern "C" int blgnn_linear_sp(const void *xs, int64_t ldxs, const uint8_t *xe, int64_t ldxe, const void *w_packed,
                               float *y, int64_t ldy, void *ys, int64_t ldys, uint8_t *ye, int64_t ldye,
                               int32_t M, int32_t N, int32_t K, const float *bias, const float *rowvec,
                               const float *rowvec_w, int32_t act, const float *act_param,
                               const float *bn_scale, const float *bn_shift, const float *addend, int64_t ld_addend,
                               const int32_t *addend_rows, int32_t num_addend_rows, const int32_t *row_graph,
                               const int32_t *seg, float *pooled, int64_t ldp, int32_t pool_kind, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N > 0 && K > 0, "linear_sp: bad sizes");
    if (M == 0) return BLGNN_OK;
    if (N % SP_BN != 0 || K % SP_BLOCK != 0 || K > 8 * SP_BLOCK)
        return fail(BLGNN_ERR_UNSUPPORTED, "linear_sp: needs N %% 256 == 0, K %% 256 == 0 and K <= 2048 (got N=%d K=%d); use "
                    "blgnn_linear_h3 / blgnn_linear", N, K);
    BLGNN_CHECK_ARG(!addend || (ld_addend >= N && ld_addend % 4 == 0 && aligned16(addend) && num_addend_rows > 0 &&
                                (addend_rows || num_addend_rows >= M)),
                    "linear_sp: the addend needs 16-byte aligned rows of >= N columns (and M rows without addend_rows)");
    BLGNN_CHECK_ARG(!addend || (int64_t)num_addend_rows * ld_addend * 4 < ((int64_t)1 << 31),
                    "linear_sp: the addend must be smaller than 2 GiB (32-bit buffer offsets)");
    BLGNN_CHECK_ARG(addend || !addend_rows, "linear_sp: addend_rows without addend");
    BLGNN_CHECK_ARG(xs && xe && w_packed && (y || ys || pooled), "linear_sp: null pointer");
    BLGNN_CHECK_ARG(!pooled || (row_graph && seg && ldp >= N && (pool_kind == BLGNN_REDUCE_SUM || pool_kind == BLGNN_REDUCE_MEAN)),
                    "linear_sp: fused readout needs row_graph, seg, ldp >= N and a reduce kind");
    if (pooled && (addend || ys || act > BLGNN_ACT_RELU))
        return fail(BLGNN_ERR_UNSUPPORTED, "linear_sp: the fused readout goes with relu / identity, fp32 rows or no rows at "
                    "all, and no addend");
    BLGNN_CHECK_ARG(ldxs >= 2 * (int64_t)K && ldxs % 8 == 0 && aligned16(xs) && ldxe >= K / SP_BLOCK,
                    "linear_sp: bad X planes (ldxs >= 2K, 16-byte aligned rows)");
    BLGNN_CHECK_ARG(ldxe % 8 == 0 && (reinterpret_cast<uintptr_t>(xe) & 7u) == 0,
                    "linear_sp: the X scale rows must be 8-byte aligned (ldxe %% 8 == 0: the kernel fetches a row's block "
                    "exponents with one 8-byte load)");
    BLGNN_CHECK_ARG(aligned16(w_packed), "linear_sp: packed weights must be 16-byte aligned");
    BLGNN_CHECK_ARG(!y || (ldy >= N && ldy % 4 == 0 && aligned16(y)), "linear_sp: y must have 16-byte aligned rows");
    BLGNN_CHECK_ARG(!ys || (ye && ldys >= 2 * (int64_t)N && ldys % 8 == 0 && aligned16(ys) && ldye >= N / SP_BLOCK),
                    "linear_sp: bad Y planes");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "linear_sp: bad activation id");
    BLGNN_CHECK_ARG((rowvec == nullptr) == (rowvec_w == nullptr), "linear_sp: rowvec and rowvec_w go together");
    BLGNN_CHECK_ARG(xs != ys, "linear_sp: x and y planes must not alias");
    LinearSpArgs P;
    P.xs = reinterpret_cast<const _Float16 *>(xs); P.ldxs = ldxs; P.xe = xe; P.ldxe = ldxe;
    P.wp = reinterpret_cast<const _Float16 *>(w_packed);
    P.y = y; P.ldy = ldy; P.ys = reinterpret_cast<_Float16 *>(ys); P.ldys = ldys; P.ye = ye; P.ldye = ldye;
    P.M = M; P.N = N; P.K = K; P.rowvec = rowvec; P.rowvec_w = rowvec_w;
    P.epi = Epilogue{bias, act_param, bn_scale, bn_shift, act};
    P.addend = addend; P.ldadd = ld_addend; P.addend_rows = addend_rows; P.n_addend = num_addend_rows;
    P.row_graph = row_graph; P.seg = seg; P.pool = pooled; P.ldp = ldp; P.pool_mean = pool_kind == BLGNN_REDUCE_MEAN;
    return launch_linear_sp(P, (hipStream_t)stream);
}
