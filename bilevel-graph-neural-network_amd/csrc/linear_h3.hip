// Dense feature transform on the fp16 matrix cores at fp32 accuracy ("f16x3", see h3_common.hpp /
// gin_h3.hip for the scheme): half the matrix-core work of linear_x6.hip for the same result.
//
//   Y = epi( X[M,K] @ W[K,N] + bias + rowvec (x) rowvec_w )      epi = act, then affine
//
// Same contract as blgnn_linear; W arrives range-normalised, split and packed (blgnn_pack_weight_h3g).
// These are the GEMMs of the GMN propagator (model/layers_gmn.py:19-26,39-47: f_message, f_node) and
// every other nn.Linear / x @ W of the path once M is large.
//
// Range normalisation of X: every row is scaled by the power of two that moves its largest magnitude
// (row_amax[m], from blgnn_row_absmax or from the kernel that produced X) into [2^14, 2^15); the
// inverse goes onto the fp32 accumulators in the epilogue together with the weight matrix's.
//
// Structure: one 512-thread workgroup per CU, block tile 128 x 128, K step 64, ROLE SPLIT:
//   waves 0-3 (one per SIMD) only read operand fragments from LDS and issue MFMAs (2 x 2 tiles of
//             32 x 32 each, 48 MFMAs per K step) and own the epilogue;
//             They also fetch the W tile image of the next step by LDS-DMA (32 linear 1 KB pieces).
//   waves 4-7 (the second wave of each SIMD) own the X operand and all vector work: X tile fp32 ->
//             registers (requested three K steps ahead) -> scale -> split -> two fp16 images in LDS.
// One barrier per K step, two LDS buffers of 64 KB.  A measured knock-out of the single-role version
// (four waves doing everything, two workgroups per CU) showed its phases adding up -- row sweep 2.3 ms,
// LDS / conversion 2.0 ms, MFMA 1.2 ms, loads 0.8 ms on 462 k x 1024 x 1024 -- instead of overlapping:
// with one wave per SIMD and workgroup, nothing runs under a wave's MFMAs but its own next instruction.
// LDS rows are 128 B (64 fp16) with the 16-byte chunk index XORed by (row >> 1) & 7: the 16 lanes a
// ds_read_b128 serves together then cover all 16 chunk slots of the 256-byte bank row.
// Workgroups that share an X row panel run on one XCD (ids remapped so an XCD owns a contiguous run
// of tiles).
// Tried and dropped: a 256 x 128 tile at K step 32 with ONE accumulator per tile (the second accumulator set
// does not fit): 25 % fewer operand bytes per MFMA, but the feeder's 32 KB of X per step stays the same and
// becomes the bound -- 3.71 ms vs 3.70 ms on 462 k x 1024 x 1024 (1.30 vs 1.44 ms at K = 256), with 1.7x the
// rounding error (7.8e-7 vs 4.6e-7 of the largest entry).
#include <cstdlib>
#include "linear_common.hpp"
#include "h3_common.hpp"

namespace blgnn {

struct LinearH3Args {
    LinearArgs base;             // x, ldx, y, ldy, M, N, K, rowvec, rowvec_w, epi (w / ldw unused)
    const _Float16 *wp;          // blgnn_pack_weight_h3g image: header (inverse scale) + 32 KB tile images
    const float *row_amax;       // [M] largest magnitude of every X row
    int32_t KS;                  // K steps of 64
};

constexpr int HBK = 64;                   // K step
constexpr int HROW = HBK * 2;             // LDS row: 64 fp16 = 128 B = eight 16-byte chunks, XOR swizzled
constexpr int H3_BM = 128, H3_BN = 128;
constexpr int H3_PIECE = H3_BM * HROW;    // bytes of one piece image (BM == BN): 16 KB
constexpr int H3_BUF = 4 * H3_PIECE;      // 2 X pieces + 2 W pieces: 64 KB
constexpr int H3_MAIN = 2 * H3_BUF;       // (the epilogue's staging tiles, 34 KB, reuse it)
constexpr int H3_LDS = H3_MAIN + H3_BM * 4;              // + inverse row scales

#ifdef H3_STAMP
__device__ unsigned long long g_h3_stamps[2][160];
#define H3_ST(role, idx) do { if (stamp_on && lane == 0 && sti < 160) g_h3_stamps[role][sti] = __builtin_amdgcn_s_memtime(); ++sti; } while (0)
#else
#define H3_ST(role, idx) do { } while (0)
#endif

__device__ __forceinline__ int h3_off(int row, int chunk) { return row * HROW + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- weight packing for this kernel: W[K,N] fp32 -> header + [N / 128][K / 64] tile images of 32 KB ----------
// A tile image is byte for byte what the kernel wants in LDS for one (column block, K step):
// [piece 2][column 128][chunk' 8][8 fp16] with chunk' = chunk ^ ((column >> 1) & 7) -- so the MFMA waves
// fetch it with 32 linear LDS-DMA instructions of 1 KB (global_load_lds_dwordx4: no registers, no ds_write)
// and read fragments from it without bank conflicts.  Scaled by the power of two that moves the largest
// |w| into [2^14, 2^15) (inverse in the header); zero filled past K and N.
typedef __attribute__((address_space(3))) void h3_lds_void;
typedef __attribute__((address_space(1))) void h3_glb_void;
constexpr int H3_WTILE = 2 * 128 * 128;          // bytes of one tile image

__global__ void k_pack_h3g(const float *__restrict__ w, int64_t ldw, int K, int N, int KS,
                           const unsigned *__restrict__ amax_bits, _Float16 *__restrict__ out) {
    float scale, inv;
    pow2_scale(__uint_as_float(*amax_bits), scale, inv);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte word of piece 0 AND piece 1
    if (idx == 0) *reinterpret_cast<float *>(out) = inv;
    const int NB = (N + 127) / 128;
    if (idx >= (int64_t)NB * KS * 128 * 8) return;
    const int cp = (int)(idx & 7), r = (int)((idx >> 3) & 127);
    const int64_t tile = idx >> 10;
    const int ks = (int)(tile % KS), nb = (int)(tile / KS);
    const int c = cp ^ ((r >> 1) & 7), n = nb * 128 + r;
    f16x8 p[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = ks * 64 + c * 8 + j;
        const Split2 sp = split2((k < K && n < N) ? w[(int64_t)k * ldw + n] * scale : 0.f);
        p[0][j] = sp.h1; p[1][j] = sp.h2;
    }
    _Float16 *base = out + HHDR + tile * (H3_WTILE / 2);
#pragma unroll
    for (int q = 0; q < 2; ++q)
        *reinterpret_cast<f16x8 *>(base + ((q * 128 + r) * 8 + cp) * 8) = p[q];
}

__global__ void k_absmax_g(const float *__restrict__ w, int64_t ldw, int K, int N, unsigned *out) {
    float m = 0.f;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)K * N;
         idx += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[(idx / N) * ldw + idx % N]));
    m = group_max(m, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order as uints
}

// ---- row maxima: out[m] = max_k |x[m, k]|, one wave per row -------------------------------------------------
__global__ void __launch_bounds__(256) k_row_absmax(const float *__restrict__ x, int64_t ldx, int M, int K,
                                                    float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *p = x + (int64_t)row * ldx;
    float m = 0.f;
    if ((ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {
        const float4 *p4 = reinterpret_cast<const float4 *>(p);
        const int k4 = K >> 2;
        for (int c = lane; c < k4; c += 64) {
            const float4 v = p4[c];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        for (int k = (k4 << 2) + lane; k < K; k += 64) m = fmaxf(m, fabsf(p[k]));
    } else {
        for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(p[k]));
    }
    m = group_max(m, WAVE);
    if (lane == 0) out[row] = m;
}

template <int ACT>
__global__ void __launch_bounds__(512, 2) k_linear_h3(const LinearH3Args P) {
    constexpr int BM = H3_BM, BN = H3_BN, WM = 64, WN = 64, TM = 2, TN = 2, PIECE = H3_PIECE;
    extern __shared__ float4 h3_smem[];
    char *smem_raw = reinterpret_cast<char *>(h3_smem);
    float *rinv = reinterpret_cast<float *>(smem_raw + H3_MAIN);      // [BM] 1 / row scale
    const LinearArgs &A = P.base;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // tile <- block id: blocks b, b + 8, ... share an XCD; give each XCD a contiguous run of tiles
    // (row-panel major), so the column blocks of one row panel read it from the same L2
    const int nb = (A.N + BN - 1) / BN;
    int tile;
    {
        const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int m0 = (tile / nb) * BM, n0 = (tile % nb) * BN;
    const int KT = P.KS;
#ifdef H3_STAMP
    const bool stamp_on = blockIdx.x == gridDim.x / 2 + 3 && (wid == 0 || wid == 4);
    int sti = 0;
#endif
    H3_ST(wid >> 2, sti);

    if (wid >= 4) {
        // =================================== feeder waves ===================================
        const int ptid = tid - 256;
        const int a_kq = ptid & 15, a_row = ptid >> 4;        // 16 threads x 16 B per X row of a K step, 16 rows per pass
        const float4 *__restrict__ X4 = reinterpret_cast<const float4 *>(A.x);
        const int64_t ldx4 = A.ldx >> 2;
        float rscale[8];
        // Loads are unconditional (row and k clamped into the matrix): a predicated load becomes a branch, and
        // behind branches hipcc waits vmcnt(0) -- for the requests of BOTH register stages -- instead of a
        // counted wait, which exposes a full memory round trip per K step.  Clamped elements are harmless:
        // rows >= M are never stored, and the k >= K columns meet zero weights (the packed image is zero
        // filled past K).
        const int64_t xrow[8] = {
            (int64_t)min(m0 + a_row, A.M - 1) * ldx4,       (int64_t)min(m0 + a_row + 16, A.M - 1) * ldx4,
            (int64_t)min(m0 + a_row + 32, A.M - 1) * ldx4,  (int64_t)min(m0 + a_row + 48, A.M - 1) * ldx4,
            (int64_t)min(m0 + a_row + 64, A.M - 1) * ldx4,  (int64_t)min(m0 + a_row + 80, A.M - 1) * ldx4,
            (int64_t)min(m0 + a_row + 96, A.M - 1) * ldx4,  (int64_t)min(m0 + a_row + 112, A.M - 1) * ldx4};
        const int k4max = (A.K >> 2) - 1;
        auto load_tile = [&](float4 (&ra)[8], int kt) {
            const int k4 = min(kt * (HBK / 4) + a_kq, k4max);
#pragma unroll
            for (int p = 0; p < 8; ++p) ra[p] = X4[xrow[p] + k4];
        };
        auto store_tile = [&](const float4 (&ra)[8], char *buf) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const float s = rscale[p];
                const f32x4v v = {ra[p].x * s, ra[p].y * s, ra[p].z * s, ra[p].w * s};
                const f16x4 h1 = __builtin_convertvector(v, f16x4);
                const f16x4 h2 = __builtin_convertvector(v - __builtin_convertvector(h1, f32x4v), f16x4);
                char *dst = buf + h3_off(a_row + 16 * p, a_kq >> 1) + (a_kq & 1) * 8;
                *reinterpret_cast<f16x4 *>(dst) = h1;
                *reinterpret_cast<f16x4 *>(dst + PIECE) = h2;
            }
        };
        // register ring of two K steps; loads past the end re-read the last step (never stored)
        float4 xa0[8], xa1[8];
        const int last = KT - 1;
        load_tile(xa0, 0);                                    // the long-latency requests first
        load_tile(xa1, 1 < last ? 1 : last);
        {
            float am[8];
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int m = m0 + a_row + 16 * p;
                am[p] = P.row_amax[m < A.M ? m : A.M - 1];
            }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float inv;
                pow2_scale(m0 + a_row + 16 * p < A.M ? am[p] : 0.f, rscale[p], inv);
                if (a_kq == 0) rinv[a_row + 16 * p] = inv;
            }
        }
        H3_ST(1, sti);
        store_tile(xa0, smem_raw);                            // step 0 -> buffer 0
        H3_ST(1, sti);
        load_tile(xa0, 2 < last ? 2 : last);
        lds_barrier();
        H3_ST(1, sti);
        // step kt: the MFMA waves read buffer kt & 1; this half writes step kt + 1 into the other buffer
        // (registers (kt + 1) & 1) and re-requests those registers for step kt + 3
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt + 1 < KT) {
                store_tile(xa1, smem_raw + H3_BUF);
                H3_ST(1, sti);
                load_tile(xa1, kt + 3 < last ? kt + 3 : last);
                H3_ST(1, sti);
            }
            lds_barrier();
            H3_ST(1, sti);
            if (kt + 1 < KT) {
                if (kt + 2 < KT) {
                    store_tile(xa0, smem_raw);
                    H3_ST(1, sti);
                    load_tile(xa0, kt + 4 < last ? kt + 4 : last);
                    H3_ST(1, sti);
                }
                lds_barrier();
                H3_ST(1, sti);
            }
        }
        lds_barrier();                                        // pairs with the barrier in front of the epilogue
        return;
    }

    // =================================== MFMA waves ===================================
    const int fi = lane & 31, fh = lane >> 5;
    const int wm0 = (wid >> 1) * WM, wn0 = (wid & 1) * WN;
    const float winv = *reinterpret_cast<const float *>(P.wp);
    f32x16 acc0[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[i][j][r] = 0.f; acc1[i][j][r] = 0.f; }
    // lane (row fi, k half fh) of a 32-row fragment tile; 32-row offsets keep (row >> 1) & 7 == (fi >> 1) & 7
    const int sw = (fi >> 1) & 7;
    const int arow = (wm0 + fi) * HROW, brow = 2 * PIECE + (wn0 + fi) * HROW;
    // The W tile image of a K step (32 KB, contiguous in the packed weights) comes in by LDS-DMA: these waves
    // issue its 32 x 1 KB pieces at the head of the step before the one that reads it -- no registers, no
    // ds_write, no vector work -- and wait for their own pieces in front of the barrier that publishes the
    // buffer.  (They issue no other vector-memory instruction, so the wait is exactly theirs; copying the
    // tile through registers in these waves cost 1.5 k of a 3.6 k-cycle step.)
    const char *wtiles = reinterpret_cast<const char *>(P.wp + HHDR) + (size_t)(tile % nb) * KT * H3_WTILE;
    auto dma_w = [&](int kt, char *buf) {
        const char *g = wtiles + (size_t)kt * H3_WTILE + wid * 8192 + lane * 16;
        char *l = buf + 2 * PIECE + wid * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_global_load_lds((h3_glb_void *)(g + i * 1024), (h3_lds_void *)(l + i * 1024), 16, 0, 0);
    };
    dma_w(0, smem_raw);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                            // buffer 0 is complete
    H3_ST(0, sti);
    for (int kt = 0; kt < KT; ++kt) {
        const char *cur = smem_raw + (kt & 1) * H3_BUF;
        if (kt + 1 < KT) dma_w(kt + 1, smem_raw + ((kt & 1) ^ 1) * H3_BUF);   // free since the barrier that ended step kt - 1
        // fragments of sub-step s + 1 are requested before the MFMAs of sub-step s are issued (two register sets)
        f16x8 fa[2][TM][2], fb[2][TN][2];
        auto read_frags = [&](int s, f16x8 (&a)[TM][2], f16x8 (&b)[TN][2]) {
            const int coff = ((2 * s + fh) ^ sw) << 4;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i][q] = *reinterpret_cast<const f16x8 *>(cur + arow + coff + q * PIECE + 32 * i * HROW);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j][q] = *reinterpret_cast<const f16x8 *>(cur + brow + coff + q * PIECE + 32 * j * HROW);
            }
        };
        read_frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) read_frags(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
            f16x8 (&a)[TM][2] = fa[s & 1];
            f16x8 (&b)[TN][2] = fb[s & 1];
            // term-major order: consecutive MFMAs write different accumulators
#define H3_TERM(ACC, QA, QB)                                                                          \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)       \
        ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][QA], b[j][QB], ACC[i][j], 0, 0, 0);
            H3_TERM(acc0, 0, 0)
            H3_TERM(acc1, 0, 1)
            H3_TERM(acc1, 1, 0)
#undef H3_TERM
        }
        H3_ST(0, sti);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next W tile have landed
        lds_barrier();       // buffer kt & 1 fully read (rewritten during step kt + 1), buffer (kt + 1) & 1 complete
        H3_ST(0, sti);
    }
    lds_barrier();                                            // every wave is done with the operand buffers

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc0[i][j] += acc1[i][j];
    store_out<ACT, TM, TN, WN, false>(A, acc0, reinterpret_cast<float *>(smem_raw), m0 + wm0, n0 + wn0, wid, lane,
                                      rinv + wm0, winv);
    H3_ST(0, sti);
}

template <int ACT>
static int launch_linear_h3(const LinearH3Args &P, hipStream_t s) {
    static bool declared = false;
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_h3<ACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "linear_h3: cannot raise the dynamic LDS limit to %d bytes", H3_LDS);
        declared = true;
    }
    const int64_t blocks = (int64_t)((P.base.M + H3_BM - 1) / H3_BM) * ((P.base.N + H3_BN - 1) / H3_BN);
    hipLaunchKernelGGL(k_linear_h3<ACT>, dim3((unsigned)blocks), dim3(512), H3_LDS, s, P);
    return launch_status("linear_h3");
}

}  // namespace blgnn

#include "linear_h3w.hpp"

namespace blgnn {

template <int ACT>
static int launch_linear_h3w(const LinearH3Args &P, hipStream_t s) {
    static bool declared = false;
    if (!declared) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_h3w<ACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS) != hipSuccess)
            return fail(BLGNN_ERR_LAUNCH, "linear_h3: cannot raise the dynamic LDS limit to %d bytes", W_LDS);
        declared = true;
    }
    const int64_t blocks = (int64_t)((P.base.M + W_BM - 1) / W_BM) * ((P.base.N + W_BN - 1) / W_BN);
    hipLaunchKernelGGL(k_linear_h3w<ACT>, dim3((unsigned)blocks), dim3(512), W_LDS, s, P);
    return launch_status("linear_h3");
}

// Which tile shape a weight of N columns is packed for -- and therefore which kernel multiplies by it.  A pure
// function of N (and of BLGNN_H3_WIDE=0, which pins the 128 x 128 kernel for A/B measurements), so the packer and
// the product agree without a tag in the image.
static bool h3_wide(int N) {
    static const bool allowed = [] { const char *e = getenv("BLGNN_H3_WIDE"); return !(e && e[0] == '0'); }();
    return allowed && N % W_BN == 0;
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int64_t blgnn_packed_weight_h3g_elems(int32_t K, int32_t N) {
    if (h3_wide(N)) return HHDR + ((int64_t)N / W_BN) * (((int64_t)K + WBK - 1) / WBK) * (W_WTILE / 2);
    const int64_t KS = ((int64_t)K + HBK - 1) / HBK, NB = ((int64_t)N + 127) / 128;
    return HHDR + NB * KS * (H3_WTILE / 2);               // fp16 elements
}

extern "C" int blgnn_pack_weight_h3g(const float *w, int64_t ldw, int32_t K, int32_t N, void *out, void *stream) {
    BLGNN_CHECK_ARG(w && out && K > 0 && N > 0 && ldw >= N, "pack_weight_h3g: bad args");
    BLGNN_CHECK_ARG(aligned16(out), "pack_weight_h3g: out must be 16-byte aligned");
    const int KS = (K + HBK - 1) / HBK, NB = (N + 127) / 128;
    hipStream_t s = (hipStream_t)stream;
    // header word 1 is scratch for the absolute maximum (float bits), word 0 receives 1 / scale
    unsigned *amax = reinterpret_cast<unsigned *>(out) + 1;
    if (hipMemsetAsync(out, 0, HHDR * 2, s) != hipSuccess) return fail(BLGNN_ERR_LAUNCH, "pack_weight_h3g: memset failed");
    const int64_t total = (int64_t)K * N;
    hipLaunchKernelGGL(k_absmax_g, dim3((unsigned)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024)), dim3(256), 0, s,
                       w, ldw, K, N, amax);
    if (h3_wide(N)) {
        const int KSW = (K + WBK - 1) / WBK;
        const int64_t wwords = (int64_t)(N / W_BN) * KSW * W_BN * 4;
        hipLaunchKernelGGL(k_pack_h3w, dim3((unsigned)((wwords + 255) / 256)), dim3(256), 0, s, w, ldw, K, N, KSW, amax,
                           reinterpret_cast<_Float16 *>(out));
        return launch_status("pack_weight_h3g");
    }
    const int64_t words = (int64_t)NB * KS * 128 * 8;
    hipLaunchKernelGGL(k_pack_h3g, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, w, ldw, K, N, KS, amax,
                       reinterpret_cast<_Float16 *>(out));
    return launch_status("pack_weight_h3g");
}

extern "C" int blgnn_row_absmax(const float *x, int64_t ldx, int32_t M, int32_t K, float *out, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && K > 0, "row_absmax: bad sizes");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && out && ldx >= K, "row_absmax: bad args");
    hipLaunchKernelGGL(k_row_absmax, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, M, K, out);
    return launch_status("row_absmax");
}

extern "C" int blgnn_linear_h3(const float *x, int64_t ldx, const float *row_amax, const void *w_packed, float *y,
                               int64_t ldy, float *y_row_amax, int32_t M, int32_t N, int32_t K, const float *bias, const float *rowvec,
                               const float *rowvec_w, int32_t act, const float *act_param,
                               const float *bn_scale, const float *bn_shift, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N >= 0 && K > 0, "linear_h3: bad sizes");
    if (M == 0 || N == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && row_amax && w_packed && y, "linear_h3: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldy >= N, "linear_h3: leading dimension too small");
    BLGNN_CHECK_ARG(act >= 0 && act <= BLGNN_ACT_TANH, "linear_h3: bad activation id");
    BLGNN_CHECK_ARG((rowvec == nullptr) == (rowvec_w == nullptr), "linear_h3: rowvec and rowvec_w go together");
    BLGNN_CHECK_ARG(x != y, "linear_h3: x and y must not alias");
    if (K % 4 != 0 || ldx % 4 != 0 || !aligned16(x) || !aligned16(w_packed))
        return fail(BLGNN_ERR_UNSUPPORTED, "linear_h3: needs K %% 4 == 0, ldx %% 4 == 0 and 16-byte aligned x / packed "
                    "weights (got K=%d ldx=%lld); use blgnn_linear", K, (long long)ldx);
    LinearH3Args P;
    LinearArgs &A = P.base;
    A.x = x; A.ldx = ldx; A.w = nullptr; A.ldw = 0; A.y = y; A.ldy = ldy;
    A.M = M; A.N = N; A.K = K; A.rowvec = rowvec; A.rowvec_w = rowvec_w;
    A.epi = Epilogue{bias, act_param, bn_scale, bn_shift, act};
    A.vec_a = 1; A.vec_b = 0; A.amax_out = y_row_amax;
    P.wp = reinterpret_cast<const _Float16 *>(w_packed);
    P.row_amax = row_amax;
    hipStream_t s = (hipStream_t)stream;
    if (h3_wide(N)) {
        P.KS = (K + WBK - 1) / WBK;
        switch (act) {
            case BLGNN_ACT_RELU:    return launch_linear_h3w<BLGNN_ACT_RELU>(P, s);
            case BLGNN_ACT_PRELU:   return launch_linear_h3w<BLGNN_ACT_PRELU>(P, s);
            case BLGNN_ACT_SIGMOID: return launch_linear_h3w<BLGNN_ACT_SIGMOID>(P, s);
            case BLGNN_ACT_TANH:    return launch_linear_h3w<BLGNN_ACT_TANH>(P, s);
            default:                return launch_linear_h3w<BLGNN_ACT_IDENTITY>(P, s);
        }
    }
    P.KS = (K + HBK - 1) / HBK;
    switch (act) {
        case BLGNN_ACT_RELU:    return launch_linear_h3<BLGNN_ACT_RELU>(P, s);
        case BLGNN_ACT_PRELU:   return launch_linear_h3<BLGNN_ACT_PRELU>(P, s);
        case BLGNN_ACT_SIGMOID: return launch_linear_h3<BLGNN_ACT_SIGMOID>(P, s);
        case BLGNN_ACT_TANH:    return launch_linear_h3<BLGNN_ACT_TANH>(P, s);
        default:                return launch_linear_h3<BLGNN_ACT_IDENTITY>(P, s);
    }
}
