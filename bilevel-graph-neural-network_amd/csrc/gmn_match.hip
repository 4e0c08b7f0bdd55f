// GMN cross-graph attention (model/layers_gmn.py:55-84 of the reference).
//
// One 256-thread workgroup per graph pair.  Both molecules are staged once into LDS
// (16 B/lane coalesced reads, row stride d+4 floats so that a ds_read_b128 of 64
// different rows is bank-conflict free); the reference's per-pair python loop,
// its two normalised copies, the n1 x n2 x D broadcast temporary and the second
// (transposed) similarity matrix all disappear.  Each wavefront owns query rows
// and runs a flash-style online softmax over the partner graph in chunks of 64
// rows: lane j scores partner row j (query row read through the scalar cache),
// then all lanes accumulate  sum_j p_j * B[j][:]  with lane-over-feature float4s.
// Pairs too large for the LDS budget take the same code path reading global memory.
#include <cstdlib>
#include "common.hpp"

namespace blgnn {

template <int CH>
__global__ void __launch_bounds__(512)
k_gmn_match(const int32_t *__restrict__ seg, int32_t num_pairs, const float *__restrict__ x,
            int64_t ldx, float *__restrict__ u, int64_t ldu, int32_t d, int32_t lds_floats, int32_t skip_small,
            int32_t parts, float *__restrict__ amax, const int32_t *__restrict__ list, int32_t n_list) {
    extern __shared__ float4 smem4[];
    float *smem = reinterpret_cast<float *>(smem4);
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int d4 = d >> 2;
    const int lds_ld = d + 4;

    // `parts` workgroups share one pair (interleaved query rows): when the matrix-core kernel has taken
    // the small pairs, the few big ones left would otherwise each serialise on a single workgroup
    auto process = [&](int pair, int part) {
        const int s0 = seg[2 * pair], s1 = seg[2 * pair + 1], s2 = seg[2 * pair + 2];
        const int n1 = s1 - s0, n2 = s2 - s1, nt = n1 + n2;
        const bool staged = (int64_t)nt * lds_ld <= (int64_t)lds_floats;
        __syncthreads();                                   // previous pair done with LDS
        if (staged) {
            for (int idx = threadIdx.x; idx < nt * d4; idx += blockDim.x) {
                const int r = idx / d4, c4 = idx - r * d4;
                const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t)(s0 + r) * ldx + 4 * c4);
                *reinterpret_cast<float4 *>(smem + (int64_t)r * lds_ld + 4 * c4) = v;
            }
        }
        __syncthreads();

        const int nwaves = (int)(blockDim.x / WAVE);
        for (int t = wid + nwaves * part; t < nt; t += nwaves * parts) {
            const bool first = t < n1;
            const int arow_g = s0 + t;                     // global row of the query node
            const int b0 = first ? n1 : 0;                 // partner rows (local index)
            const int nb = first ? n2 : n1;
            const float *arow = x + (int64_t)arow_g * ldx; // wave-uniform -> scalar loads
            const float *Bbase = staged ? smem + (int64_t)b0 * lds_ld : x + (int64_t)(s0 + b0) * ldx;
            const int64_t bld = staged ? lds_ld : ldx;

            // |a|^2 (every lane computes the same value)
            float na = 0.f;
            for (int k4 = 0; k4 < d4; ++k4) {
                const float4 a = *reinterpret_cast<const float4 *>(arow + 4 * k4);
                na = fmaf(a.x, a.x, na); na = fmaf(a.y, a.y, na);
                na = fmaf(a.z, a.z, na); na = fmaf(a.w, a.w, na);
            }
            const float inva = 1.f / fmaxf(sqrtf(na), 1e-12f);

            float m_run = -INFINITY, l_run = 0.f;
            float4 acc[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) acc[k] = f4_zero();

            // small partner graphs leave most of the 64 lanes idle with one lane per partner row;
            // P lanes share a row (each takes 1/P of the feature range) when nb <= 64/P
            const int P = nb <= 16 ? 4 : (nb <= 32 ? 2 : 1);
            const int rows_per_chunk = WAVE / P;
            for (int j0 = 0; j0 < nb; j0 += rows_per_chunk) {
                const int jloc = lane % rows_per_chunk, part = lane / rows_per_chunk;
                const int j = j0 + jloc;
                const bool live = j < nb;
                float s = -INFINITY;
                {
                    float dot = 0.f, nbq = 0.f;
                    if (live) {
                        const float *brow = Bbase + (int64_t)j * bld;
                        const int k_lo = (part * d4) / P, k_hi = ((part + 1) * d4) / P;
                        for (int k4 = k_lo; k4 < k_hi; ++k4) {
                            const float4 b = *reinterpret_cast<const float4 *>(brow + 4 * k4);
                            const float4 a = *reinterpret_cast<const float4 *>(arow + 4 * k4);
                            dot = fmaf(a.x, b.x, dot); dot = fmaf(a.y, b.y, dot);
                            dot = fmaf(a.z, b.z, dot); dot = fmaf(a.w, b.w, dot);
                            nbq = fmaf(b.x, b.x, nbq); nbq = fmaf(b.y, b.y, nbq);
                            nbq = fmaf(b.z, b.z, nbq); nbq = fmaf(b.w, b.w, nbq);
                        }
                    }
                    for (int off = rows_per_chunk; off < WAVE; off <<= 1) {      // combine the P parts
                        dot += __shfl_xor(dot, off, WAVE);
                        nbq += __shfl_xor(nbq, off, WAVE);
                    }
                    if (live) s = dot * inva * (1.f / fmaxf(sqrtf(nbq), 1e-12f));
                }
                const float m_new = fmaxf(m_run, group_max(s, WAVE));
                const float resc = expf(m_run - m_new);          // 0 on the first chunk
                const float p = live ? expf(s - m_new) : 0.f;
                l_run = l_run * resc + group_sum(part == 0 ? p : 0.f, WAVE);
                m_run = m_new;
#pragma unroll
                for (int k = 0; k < CH; ++k) { acc[k].x *= resc; acc[k].y *= resc; acc[k].z *= resc; acc[k].w *= resc; }
                const int cnt = (nb - j0) < rows_per_chunk ? (nb - j0) : rows_per_chunk;
                for (int jj = 0; jj < cnt; ++jj) {
                    const float pj = __shfl(p, jj, WAVE);
                    const float *brow = Bbase + (int64_t)(j0 + jj) * bld;
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int c4 = lane + k * WAVE;
                        if (c4 < d4) {
                            const float4 b = *reinterpret_cast<const float4 *>(brow + 4 * c4);
                            f4_fma(acc[k], pj, b);
                        }
                    }
                }
            }
            const float invl = nb > 0 ? 1.f / l_run : 0.f;
            float *urow = u + (int64_t)arow_g * ldu;
            float rmx = 0.f;
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int c4 = lane + k * WAVE;
                if (c4 < d4) {
                    const float4 a = *reinterpret_cast<const float4 *>(arow + 4 * c4);
                    float4 r;
                    r.x = a.x - acc[k].x * invl; r.y = a.y - acc[k].y * invl;
                    r.z = a.z - acc[k].z * invl; r.w = a.w - acc[k].w * invl;
                    *reinterpret_cast<float4 *>(urow + 4 * c4) = r;
                    rmx = fmaxf(rmx, fmaxf(fmaxf(fabsf(r.x), fabsf(r.y)), fmaxf(fabsf(r.z), fabsf(r.w))));
                }
            }
            if (amax) {
                rmx = group_max(rmx, WAVE);
                if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(amax) + arow_g, __float_as_uint(rmx));
            }
        }
    };
    if (list) {                                            // the class's pair list: `parts` workgroups per listed pair
        for (int item = blockIdx.x; item < n_list * parts; item += gridDim.x) process(list[item / parts], item % parts);
    } else if (skip_small < 0) {
        for (int item = blockIdx.x; item < num_pairs * parts; item += gridDim.x) process(item / parts, item % parts);
    } else {
        // the matrix-core kernels own every pair up to skip_small nodes per graph: 64 pairs are inspected with one
        // vector load of their segment offsets and only the (rare) larger ones are processed
        // (`parts` consecutive workgroups inspect the same 64 pairs and share each large one's query rows)
        const int group = blockIdx.x / parts, part = blockIdx.x - group * parts, groups = gridDim.x / parts;
        for (int base = group * WAVE; base < num_pairs; base += groups * WAVE) {
            const int pr = base + lane;
            bool need = false;
            if (pr < num_pairs) {
                const int a = seg[2 * pr + 1] - seg[2 * pr], b = seg[2 * pr + 2] - seg[2 * pr + 1];
                need = a > skip_small || b > skip_small;
            }
            unsigned long long mask = __ballot(need);
            while (mask) {
                const int q = __builtin_ctzll(mask);
                mask &= mask - 1;
                process(base + q, part);
            }
        }
    }
}


// ---- matrix-core kernels ---------------------------------------------------------------------------------------------------
// The work of a pair is three small GEMMs -- S = G1 G2^T over the D features, U1 = G1 - A G2, U2 = G2 - A' G1 -- so it
// runs on v_mfma_f32_32x32x2_f32 (exact f32; 6 n1 n2 D flop is nothing for the matrix cores, the scalar kernel above
// spends 25x the roofline time on it).  Cosines lie in [-1, 1], so the softmaxes need no running maximum:
//     E_ij = exp(cos_ij)   (0 outside n1 x n2),   A = diag(1 / rowsum E) E,   A' = diag(1 / colsum E) E^T
// E is formed ONCE in LDS; pass 2 feeds it to the matrix cores as it is (row-major for U1, read transposed for U2) and the
// 1 / sum factors scale the 16 results of a lane at store time -- no exp, no attention matrix, no VALU work inside the
// MFMA loops.  Features stream in 64-column chunks, two chunks in flight per thread (the chunk loads are what a pair waits
// for: ~20 barriers per pair against ~1 us of matrix-core time), pass 2's first chunks requested before the softmax phase.
//   k_gmn_match_mfma      both graphs <= 64 nodes (95 % of DrugBank / DrugCombo pairs): 256 threads, two workgroups per CU;
//                         pass 1 splits a chunk's k range over the four waves (partials summed in a fixed order)
//   k_gmn_match_mfma_big  a graph of 65 .. 128 nodes (5 % of the pairs, 9 % of the atoms -- the scalar kernel spent more time
//                         on them than on the other 95 %): 512 threads, one workgroup per CU, E (128 x 128) resident;
//                         pass 1: the <= 16 tiles of S, two per wave; pass 2: one U1 and one U2 tile per wave and chunk
typedef float f32x16m __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));
constexpr int MM = 64;              // max nodes per graph (small kernel)
constexpr int MCH = 64;             // feature chunk
constexpr int MLD = 65;             // LDS row stride (floats) of a chunk / of the small kernel's S
constexpr int MREG = MM * MLD;      // floats per LDS region (small kernel)
constexpr int BM = 128;             // max nodes per graph (big kernel)
constexpr int BLD = 129;            // LDS row stride of the big kernel's E
constexpr int BIG_PPW = 8;          // pairs inspected per workgroup of the big kernel (most are not big: the
                                    // hardware's workgroup dispatch balances the few that are)

// chunk c of both graphs of a pair -> 8 float4 per thread (rows lrow + RS p of graph 1 for p < 4, of graph 2 for p >= 4)
template <int RS>
__device__ __forceinline__ void match_fetch(float4 (&pre)[8], __amdgpu_buffer_rsrc_t RX, unsigned xvoff, int lrow, int n1, int n2,
                                            int64_t ldx, int c) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int r = lrow + RS * (p & 3);
        const bool first = p < 4;
        const bool live = first ? r < n1 : r < n2;
        const unsigned soff = (unsigned)(((first ? 0 : n1) + RS * (p & 3)) * ldx + c * MCH) * 4u;
        const u32x4m v = __builtin_amdgcn_raw_buffer_load_b128(RX, xvoff, soff, 0);
        const float4 f = __builtin_bit_cast(float4, v);
        pre[p] = live ? f : f4_zero();
    }
}

template <int RS>
__device__ __forceinline__ void match_park(const float4 (&pre)[8], float *G1c, float *G2c, int lrow, int lc4) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        float *dst = (p < 4 ? G1c : G2c) + (lrow + RS * (p & 3)) * MLD + 4 * lc4;
        dst[0] = pre[p].x; dst[1] = pre[p].y; dst[2] = pre[p].z; dst[3] = pre[p].w;
    }
}

__device__ __forceinline__ void match_norms(float (&nsq)[8], const float4 (&pre)[8]) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        nsq[p] = fmaf(pre[p].x, pre[p].x, nsq[p]); nsq[p] = fmaf(pre[p].y, pre[p].y, nsq[p]);
        nsq[p] = fmaf(pre[p].z, pre[p].z, nsq[p]); nsq[p] = fmaf(pre[p].w, pre[p].w, nsq[p]);
    }
}

// U tile of pass 2: o = E-operand x partner chunk, then own chunk - o * (1 / sum) -> global.
//   TRANS = false: rows of graph 1 (E row-major),  TRANS = true: rows of graph 2 (E read transposed)
template <bool TRANS, int ELD>
__device__ __forceinline__ void match_u_tile(const float *E, const float *isum, const float *own, const float *partner, int rt, int ct,
                                             int kmax, int nrows, int row0, int64_t ldu, int c, __amdgpu_buffer_rsrc_t RU,
                                             unsigned uvoff, int fi, int fh, unsigned *umax) {
    const float *er = TRANS ? E + fh * ELD + 32 * rt + fi : E + (32 * rt + fi) * ELD + fh;
    const float *br = partner + fh * MLD + 32 * ct + fi;
    f32x16m o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    for (int k = 0; k < kmax; k += 2)       // (requesting eight operand pairs ahead by hand measured slower: 314 vs 276 us)
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(TRANS ? er[k * ELD] : er[k], br[k * MLD], o, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rc = 32 * rt + (r & 3) + 8 * (r >> 2);                  // + 4 fh = row inside the graph
        const unsigned soff = (unsigned)((row0 + rc) * ldu + c * MCH + 32 * ct) * 4u;
        const float val = own[(rc + 4 * fh) * MLD + 32 * ct + fi] - o[r] * isum[rc + 4 * fh];
        if (rc + 4 * fh < nrows) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, val), RU, uvoff, soff, 0);
        if (umax) {
            // row magnitude: the 32 lanes of a half-wave hold 32 columns of row rc + 4 fh; one LDS atomic per row and tile
            float mx = fabsf(val);
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off, WAVE));
            if (fi == 0 && rc + 4 * fh < nrows) atomicMax(umax + row0 + rc + 4 * fh, __float_as_uint(mx));
        }
    }
}

__global__ void __launch_bounds__(256, 2)
k_gmn_match_mfma(const int32_t *__restrict__ seg, int32_t num_pairs, const float *__restrict__ x, int64_t ldx,
                 float *__restrict__ u, int64_t ldu, int32_t d, float *__restrict__ amax) {
    extern __shared__ float4 smem4[];
    float *R0 = reinterpret_cast<float *>(smem4);       // pass 1: G1 chunk | partial 2
    float *R1 = R0 + MREG;                              // pass 1: G2 chunk | partial 3 ; pass 2: G2 chunk
    float *R2 = R1 + MREG;                              // partial 0 -> E
    float *R3 = R2 + MREG;                              // partial 1 ; pass 2: G1 chunk
    float *inv = R3 + MREG;                             // [2 MM] 1 / max(|row|, eps)
    float *ril = inv + 2 * MM, *cil = ril + MM;         // 1 / row sums, 1 / column sums of E
    unsigned *umx = reinterpret_cast<unsigned *>(cil + MM);   // [2 MM] row magnitudes of the pair's result (float bits)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int lrow = tid >> 4, lc4 = tid & 15;          // chunk loads: 16 threads x 16 B per row, 16 rows per pass
    const int nch = d / MCH;
    // Buffer addressing: one per-thread VGPR offset + a wave-uniform SGPR offset per access, so the chunk loads and the
    // 32 result stores of a thread cost no address registers.  The descriptors are built PER PAIR (base = the pair's
    // first row): offsets stay far below 2^32 bytes whatever the size of x / u (d = 512 at 657 k atoms is 5.4 GB), and
    // rows past the pair read as zero.
    const unsigned xvoff = (unsigned)(lrow * ldx + 4 * lc4) * 4u;
    const unsigned uvoff = (unsigned)(4 * fh * ldu + fi) * 4u;

    for (int pair = blockIdx.x; pair < num_pairs; pair += gridDim.x) {
        const int s0 = seg[2 * pair], s1 = seg[2 * pair + 1], s2 = seg[2 * pair + 2];
        const int n1 = s1 - s0, n2 = s2 - s1;
        if (n1 > MM || n2 > MM || (n1 == 0 && n2 == 0)) continue;          // bigger pairs: k_gmn_match_mfma_big
        const int TM = (n1 + 31) >> 5, TN = (n2 + 31) >> 5;
        const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (int64_t)s0 * ldx), 0, (int)(((int64_t)(n1 + n2 - 1) * ldx + d) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t RU = __builtin_amdgcn_make_buffer_rsrc(
            u + (int64_t)s0 * ldu, 0, (int)(((int64_t)(n1 + n2 - 1) * ldu + d) * 4), 0x00020000);
        float4 preA[8];

        // ---- pass 1: raw similarities + squared norms -----------------------------------------------------
        f32x16m acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float nsq[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) nsq[p] = 0.f;
        auto s_chunk = [&]() {
            const int kb = wid * (MCH / 4);                    // this wave's 16 k of the chunk
            if (TM == 1 && TN == 1) {                          // wave-uniform: one 32 x 32 tile
#pragma unroll
                for (int ks = 0; ks < MCH / 4; ks += 2)
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(R0[fi * MLD + kb + ks + fh],
                                                                     R1[fi * MLD + kb + ks + fh], acc[0][0], 0, 0, 0);
            } else {                                           // all four tiles (padding rows are zero)
#pragma unroll
                for (int ks = 0; ks < MCH / 4; ks += 2) {
                    float a[2], b[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i] = R0[(32 * i + fi) * MLD + kb + ks + fh];
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[j] = R1[(32 * j + fi) * MLD + kb + ks + fh];
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
        };
        match_fetch<16>(preA, RX, xvoff, lrow, n1, n2, ldx, 0);
        for (int c = 0; c < nch; ++c) {
            __syncthreads();                                   // previous chunk / previous pair done with R0, R1
            match_park<16>(preA, R0, R1, lrow, lc4);
            match_norms(nsq, preA);
            if (c + 1 < nch) match_fetch<16>(preA, RX, xvoff, lrow, n1, n2, ldx, c + 1);
            __syncthreads();
            s_chunk();
        }
        match_fetch<16>(preA, RX, xvoff, lrow, n1, n2, ldx, 0);      // pass 2's first chunk: in flight during the softmax phase
        __syncthreads();                                       // all waves done reading the last chunk
        // partial S of wave w -> region (w == 0 ? R2 : w == 1 ? R3 : w == 2 ? R0 : R1); norms -> inv
        {
            float *dstp = wid == 0 ? R2 : (wid == 1 ? R3 : (wid == 2 ? R0 : R1));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        dstp[row * MLD + 32 * j + fi] = acc[i][j][r];
                    }
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                float v = nsq[p];
                v += __shfl_xor(v, 1, WAVE); v += __shfl_xor(v, 2, WAVE);
                v += __shfl_xor(v, 4, WAVE); v += __shfl_xor(v, 8, WAVE);
                if (lc4 == 0) inv[(p < 4 ? 0 : MM) + lrow + 16 * (p & 3)] = 1.f / fmaxf(sqrtf(v), 1e-12f);
            }
        }
        __syncthreads();
        for (int idx = tid; idx < MM * MM; idx += 256) {       // E = exp(cosine), fixed summation order of the partials
            const int i = idx >> 6, j = idx & 63;
            const int o = i * MLD + j;
            const float cs = (((R2[o] + R3[o]) + R0[o]) + R1[o]) * inv[i] * inv[MM + j];
            R2[o] = (i < n1 && j < n2) ? expf(cs) : 0.f;
        }
        __syncthreads();
        if (tid < MM) {                                        // 1 / row sums
            float l = 0.f;
            for (int j = 0; j < n2; ++j) l += R2[tid * MLD + j];
            ril[tid] = (tid < n1 && n2 > 0) ? 1.f / l : 0.f;
        } else if (tid < 2 * MM) {                             // 1 / column sums
            const int j = tid - MM;
            float l = 0.f;
            for (int i = 0; i < n1; ++i) l += R2[i * MLD + j];
            cil[j] = (j < n2 && n1 > 0) ? 1.f / l : 0.f;
        } else if (tid < 4 * MM) {
            umx[tid - 2 * MM] = 0u;
        }
        // ---- pass 2: U1 = G1 - A G2 (waves 0, 1), U2 = G2 - A' G1 (waves 2, 3), 32 columns per wave ---------------
        const bool second = wid >= 2;
        const int ct = wid & 1;
        auto u_chunk = [&](int c) {
            if (!second) {
#pragma unroll 1
                for (int i = 0; i < 2; ++i)
                    if (i < TM) match_u_tile<false, MLD>(R2, ril, R3, R1, i, ct, 32 * TN, n1, 0, ldu, c, RU, uvoff, fi, fh, amax ? umx : nullptr);
            } else {
#pragma unroll 1
                for (int i = 0; i < 2; ++i)
                    if (i < TN) match_u_tile<true, MLD>(R2, cil, R1, R3, i, ct, 32 * TM, n2, n1, ldu, c, RU, uvoff, fi, fh, amax ? umx : nullptr);
            }
        };
        for (int c = 0; c < nch; ++c) {
            __syncthreads();                                   // sums written / previous chunk consumed
            match_park<16>(preA, R3, R1, lrow, lc4);           // G1 chunk -> R3, G2 chunk -> R1
            if (c + 1 < nch) match_fetch<16>(preA, RX, xvoff, lrow, n1, n2, ldx, c + 1);
            __syncthreads();
            u_chunk(c);
        }
        if (amax) {                                            // the pair's row magnitudes -> global (atomic maximum)
            __syncthreads();
            if (tid < n1 + n2) atomicMax(reinterpret_cast<unsigned *>(amax) + s0 + tid, umx[tid]);
        }
    }
}

__global__ void __launch_bounds__(512)
k_gmn_match_mfma_big(const int32_t *__restrict__ seg, int32_t num_pairs, const float *__restrict__ x, int64_t ldx,
                     float *__restrict__ u, int64_t ldu, int32_t d, float *__restrict__ amax,
                     const int32_t *__restrict__ list, int32_t n_list) {
    extern __shared__ float4 smem4[];
    float *E = reinterpret_cast<float *>(smem4);        // [BM][BLD]
    float *G1c = E + BM * BLD;                          // [BM][MLD]
    float *G2c = G1c + BM * MLD;                        // [BM][MLD]
    float *inv = G2c + BM * MLD;                        // [2 BM] 1 / max(|row|, eps)
    float *ril = inv + 2 * BM, *cil = ril + BM;         // 1 / row sums, 1 / column sums of E
    unsigned *umx = reinterpret_cast<unsigned *>(cil + BM);   // [2 BM] row magnitudes of the pair's result (float bits)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fi = lane & 31, fh = lane >> 5;
    const int lrow = tid >> 4, lc4 = tid & 15;          // chunk loads: 16 threads x 16 B per row, 32 rows per pass
    const int nch = d / MCH;
    const unsigned xvoff = (unsigned)(lrow * ldx + 4 * lc4) * 4u;
    const unsigned uvoff = (unsigned)(4 * fh * ldu + fi) * 4u;

    // which of this workgroup's BIG_PPW pairs are big: one vector load of their segment offsets -- or, with the class's
    // pair list (one workgroup per listed pair), just that pair
    unsigned todo;
    if (list) todo = blockIdx.x < (unsigned)n_list ? 1u : 0u;
    else {
        const int q = lane < BIG_PPW ? lane : 0;
        const int pr = blockIdx.x + q * gridDim.x;
        bool big = false;
        if (lane < BIG_PPW && pr < num_pairs) {
            const int a = seg[2 * pr + 1] - seg[2 * pr], b = seg[2 * pr + 2] - seg[2 * pr + 1];
            big = (a > MM || b > MM) && a <= BM && b <= BM;
        }
        todo = (unsigned)__ballot(big);
    }
    while (todo) {
        const int q = __builtin_ctz(todo);
        todo &= todo - 1;
        const int pair = list ? list[blockIdx.x] : blockIdx.x + q * gridDim.x;
        const int s0 = seg[2 * pair], s1 = seg[2 * pair + 1], s2 = seg[2 * pair + 2];
        const int n1 = s1 - s0, n2 = s2 - s1;
        const int TM = (n1 + 31) >> 5, TN = (n2 + 31) >> 5;
        const __amdgpu_buffer_rsrc_t RX = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x + (int64_t)s0 * ldx), 0, (int)(((int64_t)(n1 + n2 - 1) * ldx + d) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t RU = __builtin_amdgcn_make_buffer_rsrc(
            u + (int64_t)s0 * ldu, 0, (int)(((int64_t)(n1 + n2 - 1) * ldu + d) * 4), 0x00020000);
        float4 preA[8], preB[8];

        // ---- pass 1: raw similarities (tiles wid and wid + 8 of the 4 x 4 grid) + squared norms ---------------------
        f32x16m acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float nsq[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) nsq[p] = 0.f;
        auto s_chunk = [&]() {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tile = wid + 8 * t, ti = tile >> 2, tj = tile & 3;
                if (ti >= TM || tj >= TN) continue;            // wave-uniform
                const float *ar = G1c + (32 * ti + fi) * MLD + fh, *br = G2c + (32 * tj + fi) * MLD + fh;
#pragma unroll 8
                for (int k = 0; k < MCH; k += 2)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[k], br[k], acc[t], 0, 0, 0);
            }
        };
        match_fetch<32>(preA, RX, xvoff, lrow, n1, n2, ldx, 0);
        if (nch > 1) match_fetch<32>(preB, RX, xvoff, lrow, n1, n2, ldx, 1);
        for (int c = 0; c < nch; c += 2) {
            __syncthreads();                                   // previous chunk / previous pair done with the chunks
            match_park<32>(preA, G1c, G2c, lrow, lc4);
            match_norms(nsq, preA);
            if (c + 2 < nch) match_fetch<32>(preA, RX, xvoff, lrow, n1, n2, ldx, c + 2);
            __syncthreads();
            s_chunk();
            if (c + 1 < nch) {
                __syncthreads();
                match_park<32>(preB, G1c, G2c, lrow, lc4);
                match_norms(nsq, preB);
                if (c + 3 < nch) match_fetch<32>(preB, RX, xvoff, lrow, n1, n2, ldx, c + 3);
                __syncthreads();
                s_chunk();
            }
        }
        match_fetch<32>(preA, RX, xvoff, lrow, n1, n2, ldx, 0);      // pass 2's first two chunks
        if (nch > 1) match_fetch<32>(preB, RX, xvoff, lrow, n1, n2, ldx, 1);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float v = nsq[p];
            v += __shfl_xor(v, 1, WAVE); v += __shfl_xor(v, 2, WAVE);
            v += __shfl_xor(v, 4, WAVE); v += __shfl_xor(v, 8, WAVE);
            if (lc4 == 0) inv[(p < 4 ? 0 : BM) + lrow + 32 * (p & 3)] = 1.f / fmaxf(sqrtf(v), 1e-12f);
        }
        __syncthreads();
        // E = exp(cosine) inside n1 x n2, 0 outside
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int tile = wid + 8 * t, ti = tile >> 2, tj = tile & 3;
            const int col = 32 * tj + fi;
            const float ic = inv[BM + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const bool in = ti < TM && tj < TN && row < n1 && col < n2;
                E[row * BLD + col] = in ? expf(acc[t][r] * inv[row] * ic) : 0.f;
            }
        }
        __syncthreads();
        if (tid < BM) {                                        // 1 / row sums
            float l = 0.f;
            for (int j = 0; j < n2; ++j) l += E[tid * BLD + j];
            ril[tid] = (tid < n1 && n2 > 0) ? 1.f / l : 0.f;
        } else if (tid < 2 * BM) {                             // 1 / column sums
            const int j = tid - BM;
            float l = 0.f;
            for (int i = 0; i < n1; ++i) l += E[i * BLD + j];
            cil[j] = (j < n2 && n1 > 0) ? 1.f / l : 0.f;
        } else {
            umx[tid - 2 * BM] = 0u;                            // threads 256 .. 511 = the 2 BM entries
        }
        // ---- pass 2: per chunk one U1 tile (rows 32 (wid >> 1), columns 32 (wid & 1)) and one U2 tile per wave ----
        const int rt = wid >> 1, ct = wid & 1;
        auto u_chunk = [&](int c) {
            if (rt < TM) match_u_tile<false, BLD>(E, ril, G1c, G2c, rt, ct, 32 * TN, n1, 0, ldu, c, RU, uvoff, fi, fh, amax ? umx : nullptr);
            if (rt < TN) match_u_tile<true, BLD>(E, cil, G2c, G1c, rt, ct, 32 * TM, n2, n1, ldu, c, RU, uvoff, fi, fh, amax ? umx : nullptr);
        };
        for (int c = 0; c < nch; c += 2) {
            __syncthreads();                                   // sums written / previous chunk consumed
            match_park<32>(preA, G1c, G2c, lrow, lc4);
            if (c + 2 < nch) match_fetch<32>(preA, RX, xvoff, lrow, n1, n2, ldx, c + 2);
            __syncthreads();
            u_chunk(c);
            if (c + 1 < nch) {
                __syncthreads();
                match_park<32>(preB, G1c, G2c, lrow, lc4);
                if (c + 3 < nch) match_fetch<32>(preB, RX, xvoff, lrow, n1, n2, ldx, c + 3);
                __syncthreads();
                u_chunk(c + 1);
            }
        }
        __syncthreads();                                       // E / chunks free for the next pair
        if (amax && tid < n1 + n2) atomicMax(reinterpret_cast<unsigned *>(amax) + s0 + tid, umx[tid]);
    }
}

}  // namespace blgnn

#include "gmn_match_wave.hpp"

using namespace blgnn;

extern "C" int blgnn_gmn_match(const int32_t *seg, int32_t num_pairs, const float *x, int64_t ldx,
                               float *u, int64_t ldu, int32_t d, float *row_amax, int32_t max_nodes, int32_t classes,
                               const int32_t *pair_order, const int32_t *class_counts_host, void *stream) {
    BLGNN_CHECK_ARG(num_pairs >= 0 && d >= 0, "gmn_match: negative size");
    if (num_pairs == 0 || d == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(seg && x && u, "gmn_match: null pointer");
    BLGNN_CHECK_ARG(ldx >= d && ldu >= d, "gmn_match: leading dimension < d");
    BLGNN_CHECK_ARG(x != u, "gmn_match: x and u must not alias");
    BLGNN_CHECK_ARG((pair_order == nullptr) == (class_counts_host == nullptr), "gmn_match: pair_order and class_counts_host go together");
    if (class_counts_host) {
        int64_t tot = 0;
        for (int k = 0; k < 4; ++k) { BLGNN_CHECK_ARG(class_counts_host[k] >= 0, "gmn_match: negative class count"); tot += class_counts_host[k]; }
        BLGNN_CHECK_ARG(tot <= num_pairs, "gmn_match: class counts exceed the number of pairs");
        if (max_nodes <= 0) max_nodes = class_counts_host[3] ? 1 << 30 : (class_counts_host[2] ? 128 : (class_counts_host[1] ? 64 : 32));
    }
    if (d % 4 != 0 || ldx % 4 != 0 || ldu % 4 != 0 || !aligned16(x) || !aligned16(u))
        return fail(BLGNN_ERR_UNSUPPORTED,
                    "gmn_match: needs d, ldx, ldu multiples of 4 and 16-byte aligned pointers");
    if (d > 2048) return fail(BLGNN_ERR_UNSUPPORTED, "gmn_match: d=%d > 2048", d);
    // LDS budget: room for a 128-row pair (the p95 of DrugBank/DrugCombo pairs is ~112 rows;
    // a pair that does not fit reads its partner rows from L2 with one row per lane, which is
    // an order of magnitude slower), between 64 KiB (two 256-thread workgroups per CU) and
    // 150 KiB (one 512-thread workgroup per CU).
    hipStream_t s = (hipStream_t)stream;
    // size classes: bit 0 = both graphs <= 64 nodes, bit 1 = a graph of 65 .. 128, bit 2 = larger (and every pair when the
    // width rules the matrix-core kernels out); 0 = all.  The classes are independent: a caller may issue them on
    // different streams (the three kernels each leave most of the chip idle).
    const bool do_small = classes == 0 || (classes & 1), do_big = classes == 0 || (classes & 2),
               do_rest = classes == 0 || (classes & 4);
    int skip_small = -1;
    if (d % MCH == 0) {                                    // matrix-core kernel for pairs up to 64 x 64 nodes
        static bool declared = false;
        const int mbytes = (4 * MREG + 6 * MM) * (int)sizeof(float);
        if (!declared) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gmn_match_mfma),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, mbytes) != hipSuccess)
                return fail(BLGNN_ERR_LAUNCH, "gmn_match: cannot raise the dynamic LDS limit");
            declared = true;
        }
        const int mblocks = num_pairs < 256 * 8 ? num_pairs : 256 * 8;
        int rc = BLGNN_OK;
        // pairs up to 64 x 64: one wavefront per pair (gmn_match_wave.hpp; two launches, <= 32 and <= 64 nodes) unless
        // the caller wants the row magnitudes merged (only the workgroup-per-pair kernel does that) or BLGNN_MATCH_WAVE=0
        static const bool wave_ok = [] { const char *e = getenv("BLGNN_MATCH_WAVE"); return !(e && e[0] == '0'); }();
        if (do_small && wave_ok && !row_amax && d % 32 == 0) {
            // pair_order / class_counts_host (optional): the pair indices grouped by size class (<= 32, <= 64, <= 128,
            // larger) -- one wavefront per listed pair, dealt to the CUs as workgroups finish
            const int32_t *l1 = pair_order, *l2 = pair_order ? pair_order + class_counts_host[0] : nullptr;
            const int c1 = pair_order ? class_counts_host[0] : num_pairs, c2 = pair_order ? class_counts_host[1] : num_pairs;
            if (c1 > 0) {
                const int b1 = (c1 + MatchWave<1>::WAVES - 1) / MatchWave<1>::WAVES;
                hipLaunchKernelGGL(k_gmn_match_wave<1>, dim3(pair_order ? b1 : (b1 < 256 * 6 ? b1 : 256 * 6)),
                                   dim3(MatchWave<1>::WAVES * 64), MatchWave<1>::WAVES * MatchWave<1>::FLOATS * sizeof(float), s,
                                   seg, num_pairs, x, ldx, u, ldu, d, l1, c1);
                rc = launch_status("gmn_match");
                if (rc != BLGNN_OK) return rc;
            }
            if (c2 > 0 && (max_nodes <= 0 || max_nodes > 32)) {
                const int b2 = (c2 + MatchWave<2>::WAVES - 1) / MatchWave<2>::WAVES;
                hipLaunchKernelGGL(k_gmn_match_wave<2>, dim3(pair_order ? b2 : (b2 < 256 * 6 ? b2 : 256 * 6)),
                                   dim3(MatchWave<2>::WAVES * 64), MatchWave<2>::WAVES * MatchWave<2>::FLOATS * sizeof(float), s,
                                   seg, num_pairs, x, ldx, u, ldu, d, l2, c2);
                rc = launch_status("gmn_match");
                if (rc != BLGNN_OK) return rc;
            }
        } else if (do_small) {
            hipLaunchKernelGGL(k_gmn_match_mfma, dim3(mblocks), dim3(256), mbytes, s, seg, num_pairs, x, ldx, u, ldu, d, row_amax);
            rc = launch_status("gmn_match");
            if (rc != BLGNN_OK) return rc;
        }
        // pairs with a graph of 65 .. 128 nodes: one 512-thread workgroup each, S resident in LDS
        static bool declared_big = false;
        const int bbytes = (BM * BLD + 2 * BM * MLD + 6 * BM) * (int)sizeof(float);
        if (!declared_big) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gmn_match_mfma_big),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, bbytes) != hipSuccess)
                return fail(BLGNN_ERR_LAUNCH, "gmn_match: cannot raise the dynamic LDS limit");
            declared_big = true;
        }
        // (max_nodes > 0: the caller knows the largest graph of the batch -- the size classes nobody is in are not
        // launched: their workgroups would each inspect their pairs' sizes and exit, 0.4 ms per call for the
        // 2 048 x 133 KB workgroups of the scalar kernel at 8 192 pairs)
        if (do_big && (max_nodes <= 0 || max_nodes > MM)) {
            const int32_t *l3 = pair_order ? pair_order + class_counts_host[0] + class_counts_host[1] : nullptr;
            const int c3 = pair_order ? class_counts_host[2] : 0;
            if (!pair_order || c3 > 0)
                hipLaunchKernelGGL(k_gmn_match_mfma_big, dim3(pair_order ? c3 : (num_pairs + BIG_PPW - 1) / BIG_PPW), dim3(512), bbytes,
                                   s, seg, num_pairs, x, ldx, u, ldu, d, row_amax, l3, c3);
            rc = launch_status("gmn_match");
            if (rc != BLGNN_OK) return rc;
        }
        skip_small = BM;
        if (max_nodes > 0 && max_nodes <= BM) return BLGNN_OK;
    }
    if (!do_rest || (pair_order && skip_small >= 0 && class_counts_host[3] == 0)) return BLGNN_OK;
    size_t want = (size_t)128 * (d + 4) * sizeof(float);
    if (want < 64 * 1024) want = 64 * 1024;
    if (want > 150 * 1024) want = 150 * 1024;
    const int lds_bytes = (int)want;
    const int lds_floats = lds_bytes / 4;
    const int threads = lds_bytes > 80 * 1024 ? 512 : 256;
    const int parts = skip_small >= 0 ? 32 : 1;   // the rare > 128-node pairs: 32 workgroups share the query rows of each
    const int32_t *l4 = (pair_order && skip_small >= 0) ? pair_order + class_counts_host[0] + class_counts_host[1] + class_counts_host[2] : nullptr;
    const int c4 = l4 ? class_counts_host[3] : 0;
    const int64_t items = l4 ? (int64_t)c4 * parts
                             : (skip_small >= 0 ? (int64_t)(num_pairs + WAVE - 1) / WAVE * parts : (int64_t)num_pairs);
    const int blocks = (int)(items < 256 * 8 ? items : 256 * 8);
    const int ch = (d / 4 + WAVE - 1) / WAVE;
#define BLGNN_GMN_LAUNCH(CH_)                                                                       \
    do {                                                                                            \
        static int declared = 0;                                                                    \
        if (lds_bytes > declared) {                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gmn_match<CH_>),              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) \
                return fail(BLGNN_ERR_LAUNCH, "gmn_match: cannot raise the dynamic LDS limit");     \
            declared = lds_bytes;                                                                   \
        }                                                                                           \
        hipLaunchKernelGGL((k_gmn_match<CH_>), dim3(blocks), dim3(threads), lds_bytes, s, seg,      \
                           num_pairs, x, ldx, u, ldu, d, lds_floats, skip_small, parts, row_amax, l4, c4); \
    } while (0)
    if (ch <= 1) BLGNN_GMN_LAUNCH(1);
    else if (ch == 2) BLGNN_GMN_LAUNCH(2);
    else if (ch <= 4) BLGNN_GMN_LAUNCH(4);
    else BLGNN_GMN_LAUNCH(8);
#undef BLGNN_GMN_LAUNCH
    return launch_status("gmn_match");
}
