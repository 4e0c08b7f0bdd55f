// f16x3 GEMM, wide variant: 128 x 256 output tile, K step 32, THREE LDS stages (included by linear_h3.hip).
//
// Why (knock-outs of the 128 x 128 kernel on 462 k x 1024 x 1024, 3.73 ms): with the feeders idle 2.84 ms, with the
// feeders AND the weight DMA off 2.14 ms (the matrix cores + barriers + epilogue alone), without the MFMAs still
// 2.96 ms.  Neither LDS fragment reads (no change without them) nor the matrix cores bound it: the two operand
// streams do, each ~16-19 B/clk/CU -- requests in flight over L2 latency (Little's law: 64 KB of X and 32 KB of W in
// flight per CU against ~2 us).  This variant needs fewer bytes per flop and keeps more of them in flight:
//   * one X tile (128 rows x 32 k) now serves 256 output columns: half the X bytes, half the feeder's loads,
//     conversions and ds_writes per flop;
//   * the W tile image of a K step (256 columns x 32 k x 2 pieces = 32 KB) is requested TWO steps ahead (three
//     48 KB stages), the X registers four steps ahead;
//   * one accumulator per 32 x 32 block (the second set does not fit next to 2 x 4 blocks per wave): the high x low
//     terms share it with the high x high term -- 7.8e-7 instead of 4.6e-7 of the largest entry, still far inside
//     the 1e-5 bar.
// Measured (462 k rows): K = N = 1024 3.47 ms (was 3.70-3.78), K = 256 / N = 1024 1.29 (1.46), K = 1024 / N = 256 1.16
// (1.21).  Letting the feeders carry the second W piece through registers (the LDS-DMA path then moves half) changed
// nothing: 3.53 ms.
// A persistent variant (one workgroup per CU walking tiles, stages rotating so that the feeders park step 0 of the next
// tile and its first W image flies while the MFMA waves run the epilogue) measured 1.26 vs 1.33 ms at K = 256 but 3.76 vs
// 3.63 ms at K = 1024 and 1.23 vs 1.19 at K = 1024 / N = 256 (static tile assignment, store drain at the tile start): dropped.
// LDS rows are 64 B (32 fp16); the 16-byte chunk index is XORed with (row >> 2) & 3, so the 16 lanes a ds_read_b128
// serves together (rows r .. r + 15, one chunk index) cover all 16 chunk slots of four consecutive rows' 256 bytes.
#pragma once

namespace blgnn {

constexpr int WBK = 32;                       // K step
constexpr int WROW = WBK * 2;                 // LDS row: 32 fp16 = 64 B = four 16-byte chunks
constexpr int W_BM = 128, W_BN = 256;
constexpr int W_XPIECE = W_BM * WROW;         // 8 KB
constexpr int W_WPIECE = W_BN * WROW;         // 16 KB
constexpr int W_STAGE = 2 * W_XPIECE + 2 * W_WPIECE;     // 48 KB
constexpr int W_NSTAGE = 3;
constexpr int W_MAIN = W_NSTAGE * W_STAGE;    // 144 KB (the epilogue's staging tiles, 68 KB, reuse it)
constexpr int W_LDS = W_MAIN + W_BM * 4;      // + inverse row scales
constexpr int W_WTILE = 2 * W_WPIECE;         // bytes of one packed tile image

__device__ __forceinline__ int h3w_off(int row, int chunk) { return row * WROW + ((chunk ^ ((row >> 2) & 3)) << 4); }

// packed image: header + [N / 256][K / 32] tile images, each [piece 2][column 256][slot 4][8 fp16],
// slot = chunk ^ ((column >> 2) & 3); scaled / zero filled like the 128-column layout
__global__ void k_pack_h3w(const float *__restrict__ w, int64_t ldw, int K, int N, int KS,
                           const unsigned *__restrict__ amax_bits, _Float16 *__restrict__ out) {
    float scale, inv;
    pow2_scale(__uint_as_float(*amax_bits), scale, inv);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte word of piece 0 AND piece 1
    if (idx == 0) *reinterpret_cast<float *>(out) = inv;
    const int NB = (N + W_BN - 1) / W_BN;
    if (idx >= (int64_t)NB * KS * W_BN * 4) return;
    const int cp = (int)(idx & 3), r = (int)((idx >> 2) & (W_BN - 1));
    const int64_t tile = idx >> 10;
    const int ks = (int)(tile % KS), nb = (int)(tile / KS);
    const int c = cp ^ ((r >> 2) & 3), n = nb * W_BN + r;
    f16x8 p[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ks * WBK + c * 8 + j;
        const Split2 sp = split2((k < K && n < N) ? w[(int64_t)k * ldw + n] * scale : 0.f);
        p[0][j] = sp.h1; p[1][j] = sp.h2;
    }
    _Float16 *base = out + HHDR + tile * (W_WTILE / 2);
#pragma unroll
    for (int q = 0; q < 2; ++q)
        *reinterpret_cast<f16x8 *>(base + ((q * W_BN + r) * 4 + cp) * 8) = p[q];
}

template <int ACT>
__global__ void __launch_bounds__(512, 2) k_linear_h3w(const LinearH3Args P) {
    constexpr int BM = W_BM, BN = W_BN, WM = 64, WN = 128, TM = 2, TN = 4;
    extern __shared__ float4 h3_smem[];
    char *smem_raw = reinterpret_cast<char *>(h3_smem);
    float *rinv = reinterpret_cast<float *>(smem_raw + W_MAIN);       // [BM] 1 / row scale
    const LinearArgs &A = P.base;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nb = (A.N + BN - 1) / BN;
    int tile;
    {
        const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = (tile / nb) * BM, n0 = (tile % nb) * BN;
    const int KT = P.KS;
    auto stage = [&](int kt) { return smem_raw + (kt % W_NSTAGE) * W_STAGE; };

    if (wid >= 4) {
        // =================================== feeder waves ===================================
        const int ptid = tid - 256;
        const int a_kq = ptid & 7, a_row = ptid >> 3;         // 8 threads x 16 B per X row of a K step, 32 rows per pass
        const float4 *__restrict__ X4 = reinterpret_cast<const float4 *>(A.x);
        const int64_t ldx4 = A.ldx >> 2;
        float rscale[4];
        // unconditional clamped loads (see linear_h3.hip: a predicated load becomes a branch + vmcnt(0))
        const int64_t xrow[4] = {
            (int64_t)min(m0 + a_row, A.M - 1) * ldx4,       (int64_t)min(m0 + a_row + 32, A.M - 1) * ldx4,
            (int64_t)min(m0 + a_row + 64, A.M - 1) * ldx4,  (int64_t)min(m0 + a_row + 96, A.M - 1) * ldx4};
        const int k4max = (A.K >> 2) - 1;
        const int last = KT - 1;
        auto load_tile = [&](float4 (&ra)[4], int kt) {
            const int k4 = min(min(kt, last) * (WBK / 4) + a_kq, k4max);
#pragma unroll
            for (int p = 0; p < 4; ++p) ra[p] = X4[xrow[p] + k4];
        };
        auto store_tile = [&](const float4 (&ra)[4], char *buf) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float s = rscale[p];
                const f32x4v v = {ra[p].x * s, ra[p].y * s, ra[p].z * s, ra[p].w * s};
                const f16x4 h1 = __builtin_convertvector(v, f16x4);
                const f16x4 h2 = __builtin_convertvector(v - __builtin_convertvector(h1, f32x4v), f16x4);
                char *dst = buf + h3w_off(a_row + 32 * p, a_kq >> 1) + (a_kq & 1) * 8;
                *reinterpret_cast<f16x4 *>(dst) = h1;
                *reinterpret_cast<f16x4 *>(dst + W_XPIECE) = h2;
            }
        };
        // register ring of four K steps: slot s holds step kt with kt % 4 == s
        float4 xa0[4], xa1[4], xa2[4], xa3[4];
        load_tile(xa0, 0); load_tile(xa1, 1); load_tile(xa2, 2); load_tile(xa3, 3);
        {
            float am[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int m = m0 + a_row + 32 * p;
                am[p] = P.row_amax[m < A.M ? m : A.M - 1];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float inv;
                pow2_scale(m0 + a_row + 32 * p < A.M ? am[p] : 0.f, rscale[p], inv);
                if (a_kq == 0) rinv[a_row + 32 * p] = inv;
            }
        }
        store_tile(xa0, stage(0));                            // step 0 -> stage 0
        load_tile(xa0, 4);
        lds_barrier();
        // during step kt the MFMA waves read stage kt % 3; this half writes step kt + 1 (ring slot (kt + 1) % 4) into
        // stage (kt + 1) % 3 -- last read during step kt - 2 -- and re-requests that slot for step kt + 5
#define H3W_FEED(KT_, RING)                                      \
    if ((KT_) < KT) {                                             \
        if ((KT_) + 1 < KT) {                                     \
            store_tile(RING, stage((KT_) + 1));                   \
            load_tile(RING, (KT_) + 5);                           \
        }                                                         \
        lds_barrier();                                            \
    }
        for (int kt = 0; kt < KT; kt += 4) {
            H3W_FEED(kt, xa1)
            H3W_FEED(kt + 1, xa2)
            H3W_FEED(kt + 2, xa3)
            H3W_FEED(kt + 3, xa0)
        }
#undef H3W_FEED
        lds_barrier();                                        // pairs with the barrier in front of the epilogue
        return;
    }

    // =================================== MFMA waves ===================================
    const int fi = lane & 31, fh = lane >> 5;
    const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;
    const float winv = *reinterpret_cast<const float *>(P.wp);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int sw = (fi >> 2) & 3;                              // 32-row offsets keep (row >> 2) & 3
    const int arow = (wm0 + fi) * WROW, brow = 2 * W_XPIECE + (wn0 + fi) * WROW;
    const char *wtiles = reinterpret_cast<const char *>(P.wp + HHDR) + (size_t)(tile % nb) * KT * W_WTILE;
    auto dma_w = [&](int kt) {                                 // this wave's quarter (8 x 1 KB) of step kt's image
        const char *g = wtiles + (size_t)kt * W_WTILE + wid * 8192 + lane * 16;
        char *l = stage(kt) + 2 * W_XPIECE + wid * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((h3_glb_void *)(g + i * 1024), (h3_lds_void *)(l + i * 1024), 16, 0, 0);
    };
    dma_w(0);
    if (KT > 1) {
        dma_w(1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // step 0 has landed, step 1 may still be in flight
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    lds_barrier();                                            // stage 0 is complete
    for (int kt = 0; kt < KT; ++kt) {
        const char *cur = stage(kt);
        if (kt + 2 < KT) dma_w(kt + 2);                        // stage (kt + 2) % 3: last read during step kt - 1
        f16x8 fa[2][TM][2], fb[2][TN][2];
        auto read_frags = [&](int s, f16x8 (&a)[TM][2], f16x8 (&b)[TN][2]) {
            const int coff = ((2 * s + fh) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                a[i][0] = *reinterpret_cast<const f16x8 *>(cur + arow + coff + 32 * i * WROW);
                a[i][1] = *reinterpret_cast<const f16x8 *>(cur + arow + coff + W_XPIECE + 32 * i * WROW);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                b[j][0] = *reinterpret_cast<const f16x8 *>(cur + brow + coff + 32 * j * WROW);
                b[j][1] = *reinterpret_cast<const f16x8 *>(cur + brow + coff + W_WPIECE + 32 * j * WROW);
            }
        };
        read_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s + 1 < 2) read_frags(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
            f16x8 (&a)[TM][2] = fa[s & 1];
            f16x8 (&b)[TN][2] = fb[s & 1];
            // term-major order: eight MFMAs between two writes of the same accumulator
#define H3W_TERM(QA, QB)                                                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][QA], b[j][QB], acc[i][j], 0, 0, 0);
            H3W_TERM(1, 0)                                     // the small terms first, the large one last
            H3W_TERM(0, 1)
            H3W_TERM(0, 0)
#undef H3W_TERM
        }
        // this wave's pieces of step kt + 1 have landed (those of step kt + 2, just requested, may be in flight)
        if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();       // stage kt % 3 fully read, stage (kt + 1) % 3 complete
    }
    lds_barrier();                                            // every wave is done with the operand stages
    store_out<ACT, TM, TN, WN, false>(A, acc, reinterpret_cast<float *>(smem_raw), m0 + wm0, n0 + wn0, wid, lane,
                                      rinv + wm0, winv);
}

}  // namespace blgnn
