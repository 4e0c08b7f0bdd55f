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
#include "common.hpp"

namespace blgnn {

template <int CH>
__global__ void __launch_bounds__(512)
k_gmn_match(const int32_t *__restrict__ seg, int32_t num_pairs, const float *__restrict__ x,
            int64_t ldx, float *__restrict__ u, int64_t ldu, int32_t d, int32_t lds_floats) {
    extern __shared__ float4 smem4[];
    float *smem = reinterpret_cast<float *>(smem4);
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const int d4 = d >> 2;
    const int lds_ld = d + 4;

    for (int pair = blockIdx.x; pair < num_pairs; pair += gridDim.x) {
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

        for (int t = wid; t < nt; t += (int)(blockDim.x / WAVE)) {
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
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const int c4 = lane + k * WAVE;
                if (c4 < d4) {
                    const float4 a = *reinterpret_cast<const float4 *>(arow + 4 * c4);
                    float4 r;
                    r.x = a.x - acc[k].x * invl; r.y = a.y - acc[k].y * invl;
                    r.z = a.z - acc[k].z * invl; r.w = a.w - acc[k].w * invl;
                    *reinterpret_cast<float4 *>(urow + 4 * c4) = r;
                }
            }
        }
    }
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_gmn_match(const int32_t *seg, int32_t num_pairs, const float *x, int64_t ldx,
                               float *u, int64_t ldu, int32_t d, void *stream) {
    BLGNN_CHECK_ARG(num_pairs >= 0 && d >= 0, "gmn_match: negative size");
    if (num_pairs == 0 || d == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(seg && x && u, "gmn_match: null pointer");
    BLGNN_CHECK_ARG(ldx >= d && ldu >= d, "gmn_match: leading dimension < d");
    BLGNN_CHECK_ARG(x != u, "gmn_match: x and u must not alias");
    if (d % 4 != 0 || ldx % 4 != 0 || ldu % 4 != 0 || !aligned16(x) || !aligned16(u))
        return fail(BLGNN_ERR_UNSUPPORTED,
                    "gmn_match: needs d, ldx, ldu multiples of 4 and 16-byte aligned pointers");
    if (d > 2048) return fail(BLGNN_ERR_UNSUPPORTED, "gmn_match: d=%d > 2048", d);
    // LDS budget: room for a 128-row pair (the p95 of DrugBank/DrugCombo pairs is ~112 rows;
    // a pair that does not fit reads its partner rows from L2 with one row per lane, which is
    // an order of magnitude slower), between 64 KiB (two 256-thread workgroups per CU) and
    // 150 KiB (one 512-thread workgroup per CU).
    size_t want = (size_t)128 * (d + 4) * sizeof(float);
    if (want < 64 * 1024) want = 64 * 1024;
    if (want > 150 * 1024) want = 150 * 1024;
    const int lds_bytes = (int)want;
    const int lds_floats = lds_bytes / 4;
    const int threads = lds_bytes > 80 * 1024 ? 512 : 256;
    const int blocks = num_pairs < 256 * 8 ? num_pairs : 256 * 8;
    hipStream_t s = (hipStream_t)stream;
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
                           num_pairs, x, ldx, u, ldu, d, lds_floats);                              \
    } while (0)
    if (ch <= 1) BLGNN_GMN_LAUNCH(1);
    else if (ch == 2) BLGNN_GMN_LAUNCH(2);
    else if (ch <= 4) BLGNN_GMN_LAUNCH(4);
    else BLGNN_GMN_LAUNCH(8);
#undef BLGNN_GMN_LAUNCH
    return launch_status("gmn_match");
}
