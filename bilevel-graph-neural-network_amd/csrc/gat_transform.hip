// GATConv's dense front end in one launch (PyG 1.1.2 GATConv.forward as called from
// model/layers.py:33,54 and model/layers_meta.py:70,78):
//     h = x @ W                      [M, N]
//     s_dst[i] = h_i . att[0:N]      s_src[i] = h_i . att[N:2N]
// for the interaction level, where M is a few thousand rows and N <= 128: a launch-latency problem,
// not a throughput one (x @ W is 43 MFLOP for DrugBank).  Same split-K idea as k_linear_splitk --
// 16 wavefronts = 4 K-groups x 4 tile waves, partial tiles meet in LDS in a fixed order -- but the
// workgroup tile is 32 rows x ALL N columns, so the rows of h are complete inside the workgroup
// and the two attention dot products come out of the epilogue instead of a second launch.
#include "linear_common.hpp"

namespace blgnn {

struct GatTransformArgs {
    const float *x; int64_t ldx;
    const float *w; int64_t ldw;
    const float *att;                 // [2 N]
    float *h; int64_t ldh;
    float *s_dst, *s_src;             // [M]
    int32_t M, N, K;
    int vec_a, vec_b;
};

template <int KG>
__global__ void __launch_bounds__(256 * KG) k_gat_transform(const GatTransformArgs A) {
    constexpr int BM = 32, BN = 128, BKT = 32;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    constexpr int OPND = (BKT * LDA + 3 + BKT * LDB + 3) & ~3;
    constexpr int LDO = BN + 4;
    __shared__ __attribute__((aligned(16))) float smem[KG * OPND];
    const int tid = threadIdx.x & 255, grp = threadIdx.x >> 8;
    const int lane = tid & 63, wid = tid >> 6;
    float *As = smem + grp * OPND;
    float *Bs = As + ((BKT * LDA + 3) & ~3);
    const int wn0 = wid * 32;
    const int m0 = blockIdx.x * BM;
    const int fi = lane & 31, fh = lane >> 5;
    const int a_kq = tid & 7, a_row = tid >> 3;          // X: 8 float4 per row, 32 rows
    const int b_nq = tid & 31, b_k = tid >> 5;           // W: 32 float4 per k row, 8 k rows per pass

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra, rb[4];
    auto load_tile = [&](int k0) {
        {
            const int m = m0 + a_row, k = k0 + 4 * a_kq;
            float4 v = f4_zero();
            if (m < A.M) {
                const float *p = A.x + (int64_t)m * A.ldx + k;
                if (A.vec_a) { if (k < A.K) v = *reinterpret_cast<const float4 *>(p); }
                else {
                    if (k + 0 < A.K) v.x = p[0];
                    if (k + 1 < A.K) v.y = p[1];
                    if (k + 2 < A.K) v.z = p[2];
                    if (k + 3 < A.K) v.w = p[3];
                }
            }
            ra = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + b_k + 8 * i, n = 4 * b_nq;
            float4 v = f4_zero();
            if (k < A.K) {
                const float *p = A.w + (int64_t)k * A.ldw + n;
                if (A.vec_b) { if (n < A.N) v = *reinterpret_cast<const float4 *>(p); }
                else {
                    if (n + 0 < A.N) v.x = p[0];
                    if (n + 1 < A.N) v.y = p[1];
                    if (n + 2 < A.N) v.z = p[2];
                    if (n + 3 < A.N) v.w = p[3];
                }
            }
            rb[i] = v;
        }
    };
    auto store_tile = [&]() {
        const int k = 4 * a_kq;
        As[(k + 0) * LDA + a_row] = ra.x;
        As[(k + 1) * LDA + a_row] = ra.y;
        As[(k + 2) * LDA + a_row] = ra.z;
        As[(k + 3) * LDA + a_row] = ra.w;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *reinterpret_cast<float4 *>(&Bs[(b_k + 8 * i) * LDB + 4 * b_nq]) = rb[i];
    };

    const int KT = (A.K + BKT - 1) / BKT;
    const int iters = (KT + KG - 1) / KG;               // same trip count for every K-group
    load_tile(grp * BKT);                               // tiles past K load zeros
    store_tile();
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (it + 1 < iters) load_tile(((it + 1) * KG + grp) * BKT);
        f32x16 part;                                     // two-level accumulation, see linear.hip
#pragma unroll
        for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < BKT; ks += 2) {
            const float a = As[(ks + fh) * LDA + fi];
            const float b = Bs[(ks + fh) * LDB + wn0 + fi];
            part = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, part, 0, 0, 0);
        }
        acc += part;
        __syncthreads();
        if (it + 1 < iters) {
            store_tile();
            __syncthreads();
        }
    }
    // partial tiles of K-groups 1..KG-1 -> LDS (operand images are dead), summed by group 0
    float *red = smem;                                   // [KG-1][4 waves][16][64]
    if (grp > 0) {
        float *dst = red + ((grp - 1) * 4 + wid) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64 + lane] = acc[r];
    }
    __syncthreads();
    float *stage = smem + (KG - 1) * 4 * 1024;           // [32][LDO]: the workgroup's complete rows of h
    if (grp == 0) {
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            const float *src = red + ((g - 1) * 4 + wid) * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += src[r * 64 + lane];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = (r & 3) + 8 * (r >> 2) + 4 * fh;
            stage[lr * LDO + wn0 + fi] = acc[r];
        }
    }
    __syncthreads();
    if (grp == 0) {
        // rows of h -> global (16 B per lane), 32 rows x N/4 float4
        const bool vec_out = (A.ldh % 4 == 0) && ((reinterpret_cast<uintptr_t>(A.h) & 15u) == 0);
        const int q = (A.N + 3) / 4;
        for (int t = tid; t < 32 * q; t += 256) {
            const int lr = t / q, c4 = t - lr * q;
            const int m = m0 + lr, nn = 4 * c4;
            if (m >= A.M) continue;
            const float4 v = *reinterpret_cast<const float4 *>(stage + lr * LDO + nn);
            float *dst = A.h + (int64_t)m * A.ldh + nn;
            if (vec_out && nn + 3 < A.N) *reinterpret_cast<float4 *>(dst) = v;
            else {
                dst[0] = v.x;
                if (nn + 1 < A.N) dst[1] = v.y;
                if (nn + 2 < A.N) dst[2] = v.z;
                if (nn + 3 < A.N) dst[3] = v.w;
            }
        }
    } else if (grp == 1) {
        // attention halves: 8 lanes per (row, half), fixed reduction tree
        const int pair = tid >> 3, l8 = tid & 7;         // 32 rows x 2 halves = 64 pairs... 256 / 8 = 32 pairs per pass
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int pr = pair + 32 * pass;             // 0..63
            const int lr = pr & 31, half = pr >> 5;
            const float *at = A.att + half * A.N;
            float s = 0.f;
            for (int c = l8; c < A.N; c += 8) s = fmaf(stage[lr * LDO + c], at[c], s);
            s += __shfl_xor(s, 1, WAVE); s += __shfl_xor(s, 2, WAVE); s += __shfl_xor(s, 4, WAVE);
            const int m = m0 + lr;
            if (l8 == 0 && m < A.M) (half == 0 ? A.s_dst : A.s_src)[m] = s;
        }
    }
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_gat_transform(const float *x, int64_t ldx, const float *w, int64_t ldw, const float *att,
                                   int32_t M, int32_t N, int32_t K, float *h, int64_t ldh, float *s_dst,
                                   float *s_src, void *stream) {
    BLGNN_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gat_transform: bad sizes");
    if (M == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && w && att && h && s_dst && s_src, "gat_transform: null pointer");
    BLGNN_CHECK_ARG(ldx >= K && ldw >= N && ldh >= N, "gat_transform: leading dimension too small");
    BLGNN_CHECK_ARG(x != h, "gat_transform: x and h must not alias");
    if (N > 128)
        return fail(BLGNN_ERR_UNSUPPORTED, "gat_transform: N=%d > 128 (use blgnn_linear + blgnn_gat_scores)", N);
    GatTransformArgs A;
    A.x = x; A.ldx = ldx; A.w = w; A.ldw = ldw; A.att = att; A.h = h; A.ldh = ldh;
    A.s_dst = s_dst; A.s_src = s_src; A.M = M; A.N = N; A.K = K;
    A.vec_a = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(x);
    A.vec_b = (N % 4 == 0) && (ldw % 4 == 0) && aligned16(w);
    hipLaunchKernelGGL((k_gat_transform<4>), dim3((M + 31) / 32), dim3(1024), 0, (hipStream_t)stream, A);
    return launch_status("gat_transform");
}
