// Cross-graph attention (model/layers_gmn.py:55-84) for pairs of small graphs, ONE WAVEFRONT PER PAIR
// (included by gmn_match.hip).
//
// Why: k_gmn_match_mfma gives a pair a 256-thread workgroup with 66 KB of LDS -- two pairs in flight per CU, ~20
// workgroup barriers per pair, ~37 us per pair = ~18 us of CU time for ~3 us of matrix-core work -- and while it runs,
// its workgroups keep the 141 KB workgroups of the propagator's GEMM off every CU (the step's kernel trace: the message
// GEMM next to it takes 1.85 ms instead of 1.0).  Here a pair belongs to one wavefront: no workgroup barrier at all
// (LDS operations of a wavefront complete in order), 13-35 KB of LDS per pair, so 4-12 pairs are in flight per CU and the
// matrix pipe of a SIMD always has another wave's MFMAs to run while one wave loads or stores.
//
//   T = 1: both graphs <= 32 nodes (one 32 x 32 tile of S),  T = 2: both <= 64 nodes (up to 2 x 2 tiles).
//   pass 1   S = G1 G2^T over the features, chunk by chunk (both graphs' chunk staged in LDS), squared row norms on the way
//   softmax  E = exp(cos) (cosines lie in [-1, 1]: no running maximum) -> LDS; 1 / row sums, 1 / column sums
//   pass 2   U1 = G1 - diag(1 / rowsum) E G2,  U2 = G2 - diag(1 / colsum) E^T G1   per 32-column chunk, both graphs' chunk
//            staged again: the partner's rows are the MFMA B operand, E (or E^T) rows the A operand, the own values LDS reads
// Exact-f32 MFMA (32x32x2) throughout, summation over k in index order: the same arithmetic as k_gmn_match_mfma.
//
// Measured (8 192 DrugCombo-shape pairs, d = 256, pairs handed over as per-class lists): 155 us (3 569 pairs <= 32 nodes) +
// 440 us (4 208 pairs <= 64) against 926 us for k_gmn_match_mfma on the same pairs.  What bounds it now is memory latency
// per chunk, not the matrix cores: in-kernel stamps of one pair (25 x 16 nodes) show 8 k cycles per 32-column chunk in
// pass 1 and 12 k in pass 2 for 1-2 k cycles of MFMA -- one chunk of prefetch (8 KB per wave, ~100 KB per CU in flight)
// against ~3 us of loaded latency; the kernel moves 520 MB in 155 us (floor: 70 us).  Inside the propagator step the
// gain is small (19.7 vs 19.85 ms: the attention shares the chip with the message GEMM either way).
#pragma once

namespace blgnn {

// LDS traffic between the lanes of ONE wavefront: the hardware executes a wavefront's LDS instructions in program order,
// so a read issued after a write sees it; the compiler only has to keep that order (it sees no dependence between one
// lane's store and another lane's load)
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }

template <int T> struct MatchWave {
    static constexpr int R = 32 * T;                       // rows per graph (padded)
    static constexpr int C1 = 32;                          // chunk width: both graphs' 2 R rows x 32 columns staged
    static constexpr int LD1 = C1 + 1;
    static constexpr int ELD = R + 1;
    static constexpr int BUF = 2 * R * LD1;
    static constexpr int FLOATS = BUF + R * ELD + 4 * R;   // chunk buffer, E, inv norms [2 R], 1 / row sums, 1 / col sums
    static constexpr int WAVES = T == 1 ? 4 : 2;           // per workgroup: 53 KB (three per CU) / 69 KB (two per CU)
};

template <int T>
__global__ void __launch_bounds__(MatchWave<T>::WAVES * 64)
k_gmn_match_wave(const int32_t *__restrict__ seg, int32_t num_pairs, const float *__restrict__ x, int64_t ldx,
                 float *__restrict__ u, int64_t ldu, int32_t d, const int32_t *__restrict__ list, int32_t n_list) {
    typedef MatchWave<T> W;
    constexpr int R = W::R, C1 = W::C1, LD1 = W::LD1, ELD = W::ELD;
    extern __shared__ float4 smem4[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *buf = reinterpret_cast<float *>(smem4) + wid * W::FLOATS;
    float *E = buf + W::BUF;
    float *inv = E + R * ELD;                              // [2 R]
    float *ril = inv + 2 * R, *cil = ril + R;
    const int fi = lane & 31, fh = lane >> 5;
    const int nch1 = d / C1;
    // staging: 8 lanes x 16 B per row, 8 rows per sweep, NSW sweeps cover the 2 R rows
    constexpr int LPR = C1 / 4, RPS = 64 / LPR, NSW = 2 * R / RPS;
    const int l_row = lane / LPR, l_c4 = lane % LPR;
    const unsigned uvoff = (unsigned)(4 * fh * ldu + fi) * 4u;

    const int wave_global = blockIdx.x * W::WAVES + wid, wave_total = gridDim.x * W::WAVES;
    // ``list`` (optional): the pairs of this size class, so that every wavefront has work (with a class of ~half the
    // pairs, scanning left 3 of 12 wave slots per CU busy); without it every wavefront inspects its share of all pairs
    const int n_items = list ? n_list : num_pairs;
    for (int item = wave_global; item < n_items; item += wave_total) {
        const int pair = list ? list[item] : item;
        const int s0 = seg[2 * pair], s1 = seg[2 * pair + 1], s2 = seg[2 * pair + 2];
        const int n1 = s1 - s0, n2 = s2 - s1;
        const int big = n1 > n2 ? n1 : n2;
        if (big > R || (T == 2 && big <= 32) || big == 0) continue;          // another size class (wave-uniform)
        const int TM = (n1 + 31) >> 5, TN = (n2 + 31) >> 5;
        const float *x0 = x + (int64_t)s0 * ldx;
        // per-pair buffer descriptors (offsets inside a pair stay far below 2^32 whatever the size of x / u)
        const __amdgpu_buffer_rsrc_t RU = __builtin_amdgcn_make_buffer_rsrc(
            u + (int64_t)s0 * ldu, 0, (int)(((int64_t)(n1 + n2 - 1) * ldu + d) * 4), 0x00020000);

        // ================================ pass 1: S and the squared norms ================================
        f32x16m acc[T][T];
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float nsq[NSW];
#pragma unroll
        for (int p = 0; p < NSW; ++p) nsq[p] = 0.f;
        // row of sweep p: local r = p * RPS + l_row; r < R -> graph 1 row r, else graph 2 row r - R
        const float4 *src1[NSW];
        bool live1[NSW];
#pragma unroll
        for (int p = 0; p < NSW; ++p) {
            const int r = p * RPS + l_row;
            const int g = r < R ? r : n1 + (r - R);                          // row inside the pair
            live1[p] = r < R ? r < n1 : (r - R) < n2;
            const int gc = live1[p] ? g : 0;                                 // clamped: the load is unconditional
            src1[p] = reinterpret_cast<const float4 *>(x0 + (int64_t)gc * ldx) + l_c4;
        }
        float4 pre[NSW];
#pragma unroll
        for (int p = 0; p < NSW; ++p) pre[p] = src1[p][0];
        for (int c = 0; c < nch1; ++c) {
#pragma unroll
            for (int p = 0; p < NSW; ++p) {
                const float4 v = live1[p] ? pre[p] : f4_zero();
                float *dst = buf + (p * RPS + l_row) * LD1 + 4 * l_c4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                nsq[p] = fmaf(v.x, v.x, nsq[p]); nsq[p] = fmaf(v.y, v.y, nsq[p]);
                nsq[p] = fmaf(v.z, v.z, nsq[p]); nsq[p] = fmaf(v.w, v.w, nsq[p]);
            }
            if (c + 1 < nch1) {
#pragma unroll
                for (int p = 0; p < NSW; ++p) pre[p] = src1[p][(c + 1) * LPR];
            }
            wave_lds_order();
#pragma unroll
            for (int i = 0; i < T; ++i) {
                if (i >= TM) continue;
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    if (j >= TN) continue;
                    const float *ar = buf + (32 * i + fi) * LD1 + fh, *br = buf + (R + 32 * j + fi) * LD1 + fh;
#pragma unroll
                    for (int k = 0; k < C1; k += 2)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[k], br[k], acc[i][j], 0, 0, 0);
                }
            }
            wave_lds_order();
        }
        // 1 / max(|row|, eps): the LPR lanes of a row hold its partial sums
#pragma unroll
        for (int p = 0; p < NSW; ++p) {
            float v = nsq[p];
#pragma unroll
            for (int off = 1; off < LPR; off <<= 1) v += __shfl_xor(v, off, WAVE);
            if (l_c4 == 0) inv[p * RPS + l_row] = 1.f / fmaxf(sqrtf(v), 1e-12f);
        }
        wave_lds_order();
        // ================================ softmax numerators and sums ================================
#pragma unroll
        for (int i = 0; i < T; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const int col = 32 * j + fi;
                const float ic = inv[R + col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const bool in = i < TM && j < TN && row < n1 && col < n2;
                    E[row * ELD + col] = in ? expf(acc[i][j][r] * inv[row] * ic) : 0.f;
                }
            }
        wave_lds_order();
        // (entries outside n1 x n2 are zero: the sums run over whole 32-blocks, eight independent reads in flight;
        //  the order of the additions is fixed)
        for (int t = lane; t < 2 * R; t += WAVE) {
            const bool rows = t < R;
            const int j = t - R;
            const float *p = rows ? E + t * ELD : E + j;
            const int step = rows ? 1 : ELD, cnt = 32 * (rows ? TN : TM);
            float l = 0.f;
            for (int k0 = 0; k0 < cnt; k0 += 8) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = p[(k0 + q) * step];
#pragma unroll
                for (int q = 0; q < 8; ++q) l += v[q];
            }
            if (rows) ril[t] = (t < n1 && n2 > 0) ? 1.f / l : 0.f;
            else cil[j] = (j < n2 && n1 > 0) ? 1.f / l : 0.f;
        }
        wave_lds_order();

        // ================================ pass 2: U1 and U2, chunk by chunk ================================
        // both graphs' 32-column chunk is staged (as in pass 1): graph 2's rows are the B operand of U1 = G1 - A G2 and
        // the own values of U2, graph 1's rows the B operand of U2 = G2 - A' G1 and the own values of U1 -- one more
        // pass over x instead of two, no separate loads of the own values, 32 MFMAs per tile pair and chunk to hide the
        // next chunk's loads behind
#pragma unroll
        for (int p = 0; p < NSW; ++p) pre[p] = src1[p][0];
        for (int c = 0; c < nch1; ++c) {
#pragma unroll
            for (int p = 0; p < NSW; ++p) {
                const float4 v = live1[p] ? pre[p] : f4_zero();
                float *dst = buf + (p * RPS + l_row) * LD1 + 4 * l_c4;
                dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
            }
            if (c + 1 < nch1) {
#pragma unroll
                for (int p = 0; p < NSW; ++p) pre[p] = src1[p][(c + 1) * LPR];
            }
            wave_lds_order();
#pragma unroll 1
            for (int job = 0; job < TM + TN; ++job) {
                // job < TM: row tile `job` of graph 1 (E row-major, partner = graph 2); else row tile of graph 2 (E transposed)
                const bool second = job >= TM;
                const int rt = second ? job - TM : job;
                const int n_own = second ? n2 : n1, own0 = second ? n1 : 0;
                const int T_par = second ? TM : TN;
                const float *isum = second ? cil : ril;
                const float *ownb = buf + ((second ? R : 0) + 32 * rt + 4 * fh) * LD1 + fi;      // + rc * LD1
                const float *er = second ? E + fh * ELD + 32 * rt + fi : E + (32 * rt + fi) * ELD + fh;
                const float *br = buf + ((second ? 0 : R) + fh) * LD1 + fi;
                const int estep = second ? ELD : 1;
                f32x16m o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.f;
                for (int k0 = 0; k0 < 32 * T_par; k0 += 32) {           // sixteen MFMAs per block: operand reads batched
                    float a[16], b[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) { a[q] = er[(k0 + 2 * q) * estep]; b[q] = br[(k0 + 2 * q) * LD1]; }
#pragma unroll
                    for (int q = 0; q < 16; ++q) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], o, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rc = 32 * rt + (r & 3) + 8 * (r >> 2);
                    const unsigned soff = (unsigned)((own0 + rc) * ldu + 32 * c) * 4u;
                    const float val = ownb[((r & 3) + 8 * (r >> 2)) * LD1] - o[r] * isum[rc + 4 * fh];
                    if (rc + 4 * fh < n_own)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, val), RU, uvoff, soff, 0);
                }
            }
            wave_lds_order();
        }
    }
}

}  // namespace blgnn
