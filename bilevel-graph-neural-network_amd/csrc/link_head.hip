// Fused pair scorer (reference: model/layers_link_pred.py:43-67 with type='mlp_concat'):
//   out[b] = final_act( MLP( [ z[ids1[b]] / |z[ids1[b]]| , z[ids2[b]] / |z[ids2[b]]| ] ) )
// One wavefront per pair.  The two embedding rows are gathered with coalesced loads, L2
// normalised with a wave reduction and parked in the wave's LDS slice; every MLP layer is then
// lane-per-output: lane o walks the (L2-resident, k-major) weight column o while the input
// activations are LDS broadcasts.  The MLP of this head is tiny (2D -> 2D/8 -> 2D/64 -> 1 or 3):
// as separate GEMM launches it is three grids of mostly idle 32x32 MFMA tiles plus a gather
// kernel plus two index copies; fused it is one short launch.
#include "common.hpp"

namespace blgnn {

constexpr int HEAD_MAX_LAYERS = 4;
constexpr int HEAD_MAX_WIDTH = 2048;          // floats per activation buffer per wave

struct HeadArgs {
    const float *z; int64_t ldz;
    const int64_t *ids1, *ids2;
    int32_t B, d;
    float eps;
    int32_t num_layers;
    const float *w[HEAD_MAX_LAYERS];
    const float *b[HEAD_MAX_LAYERS];
    int32_t dims[HEAD_MAX_LAYERS + 1];
    int32_t hidden_act, final_act;
    float *out; int64_t ldo;
    int32_t width;                            // LDS floats per buffer
    int32_t w_in_lds;                         // weights fit the LDS budget
};

__device__ __forceinline__ bool aligned16_dev(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
__device__ __forceinline__ int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

__global__ void __launch_bounds__(256) k_link_head(const HeadArgs A) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    float *buf0 = lds + (size_t)wid * 2 * A.width;
    float *buf1 = buf0 + A.width;
    // all layer weights staged once per workgroup (k-major, as stored): 33 KB for 256->32->4->1.
    // Read from L2 instead, every k-step of every pair is a dependent ~600-cycle round trip.
    float *wl = lds + 4 * 2 * (size_t)A.width;
    const float *wptr[HEAD_MAX_LAYERS];
    {
        int off = 0;
        for (int l = 0; l < A.num_layers; ++l) {
            const int cnt = A.dims[l] * A.dims[l + 1];
            if (A.w_in_lds) {
                const bool v4 = (cnt % 4 == 0) && aligned16_dev(A.w[l]);
                if (v4) {                                        // independent 16 B loads, all in flight
                    const float4 *src = reinterpret_cast<const float4 *>(A.w[l]);
                    float4 *dst = reinterpret_cast<float4 *>(wl + off);
#pragma unroll 8
                    for (int i = threadIdx.x; i < cnt / 4; i += 256) dst[i] = src[i];
                } else {
                    for (int i = threadIdx.x; i < cnt; i += blockDim.x) wl[off + i] = A.w[l][i];
                }
                wptr[l] = wl + off;
            } else {
                wptr[l] = A.w[l];
            }
            off += (cnt + 3) & ~3;
        }
    }
    __syncthreads();
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int wave_total = gridDim.x * blockDim.x / WAVE;
    for (int pair = wave_global; pair < A.B; pair += wave_total) {
        // gather + normalise (model/layers_link_pred.py:44,56-61); both rows requested together
        {
            const float *z1 = A.z + A.ids1[pair] * A.ldz, *z2 = A.z + A.ids2[pair] * A.ldz;
            float s1 = 0.f, s2 = 0.f;
            for (int c = lane; c < A.d; c += WAVE) {
                const float a = z1[c], b = z2[c];
                s1 = fmaf(a, a, s1);
                s2 = fmaf(b, b, s2);
            }
            s1 = group_sum(s1, WAVE);
            s2 = group_sum(s2, WAVE);
            const float n1 = fmaxf(sqrtf(s1), A.eps), n2 = fmaxf(sqrtf(s2), A.eps);
            for (int c = lane; c < A.d; c += WAVE) {
                buf0[c] = z1[c] / n1;
                buf0[A.d + c] = z2[c] / n2;
            }
        }
        float *in = buf0, *outb = buf1;
        for (int l = 0; l < A.num_layers; ++l) {
            const int K = A.dims[l], N = A.dims[l + 1];
            const bool last = l == A.num_layers - 1;
            const int act = last ? A.final_act : A.hidden_act;
            const float *W = wptr[l];
            // lanes = (output o, K-part): narrow layers split K over the idle lanes
            const int Np = N >= WAVE ? WAVE : pow2ceil(N);
            const int P = WAVE / Np;
            const int oo = lane % Np, part = lane / Np;
            for (int o0 = 0; o0 < N; o0 += Np) {
                const int o = o0 + oo;
                const bool live = o < N;
                float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
                int k = part;
                for (; k + 3 * P < K; k += 4 * P) {                 // 8 independent LDS reads in flight
                    const float x0 = in[k], x1 = in[k + P], x2 = in[k + 2 * P], x3 = in[k + 3 * P];
                    const float w0 = live ? W[(int64_t)k * N + o] : 0.f;
                    const float w1 = live ? W[(int64_t)(k + P) * N + o] : 0.f;
                    const float w2 = live ? W[(int64_t)(k + 2 * P) * N + o] : 0.f;
                    const float w3 = live ? W[(int64_t)(k + 3 * P) * N + o] : 0.f;
                    acc0 = fmaf(x0, w0, acc0);
                    acc1 = fmaf(x1, w1, acc1);
                    acc2 = fmaf(x2, w2, acc2);
                    acc3 = fmaf(x3, w3, acc3);
                }
                for (; k < K; k += P) acc0 = fmaf(in[k], live ? W[(int64_t)k * N + o] : 0.f, acc0);
                acc0 += acc2;
                acc1 += acc3;
                float acc = acc0 + acc1;
                for (int off = Np; off < WAVE; off <<= 1) acc += __shfl_xor(acc, off, WAVE);
                if (live && part == 0) {
                    const float v = apply_act(acc + (A.b[l] ? A.b[l][o] : 0.f), act, 0.f);
                    if (last) A.out[(int64_t)pair * A.ldo + o] = v;
                    else outb[o] = v;
                }
            }
            float *t = in; in = outb; outb = t;
        }
    }
}


// ---- matrix-core variant: 32 pairs per workgroup -----------------------------------------------------------
// For embedding widths up to 256 (first layer 2d <= 512 wide, <= 128 outputs) the head is re-blocked
// around the first layer, which holds 98 % of its flops: a workgroup gathers and normalises the rows of 32
// pairs into LDS (all 64 row loads of a wave in flight together instead of one dependent chain per
// pair), runs  [32 x 2d] @ W1  on v_mfma_f32_32x32x2_f32 with K split over its four waves and the
// weights read straight from L2 (each element once per 32 pairs, not once per 4), and finishes the
// narrow tail layers with one thread per (pair, output).
typedef float f32x16h __attribute__((ext_vector_type(16)));
constexpr int HP = 32;                      // pairs per workgroup
constexpr int HK_MAX = 512;                 // first-layer input width (2 d)
constexpr int HN_MAX = 128;                 // first-layer output width

// One wave's share of the first layer: acc[t] += X[:, k] (x) W1[k, 32 t .. 32 t + 31] over kq k's.
// The weight operands come straight from L2 (a ~1.5 k-cycle round trip): 16 k-steps are unrolled so
// their loads are all requested before the first MFMA of the batch needs one.
template <int NT>
__device__ __forceinline__ void first_layer_tiles(const float *xa, const float *wb, int kq, int N1, int fi, int fh,
                                                  float *dst, int ldh) {
    f32x16h acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bool in[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) in[t] = 32 * t + fi < N1;
    for (int k0 = 0; k0 < kq; k0 += 32) {
        float b[16][NT];
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                b[u][t] = (in[t] && k0 + 2 * u < kq) ? wb[(int64_t)(k0 + 2 * u) * N1 + 32 * t] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float a = k0 + 2 * u < kq ? xa[k0 + 2 * u] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[u][t], acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            dst[row * ldh + 32 * t + fi] = acc[t][r];
        }
}

__device__ __forceinline__ const float *wfrag_base(const HeadArgs &A, int wid, int fh, int fi) {
    return A.w[0] + (int64_t)(wid * (A.dims[0] >> 2) + fh) * A.dims[1] + fi;
}

__global__ void __launch_bounds__(256) k_link_head_mfma(const HeadArgs A) {
    extern __shared__ float lds[];
    const int K0 = A.dims[0], N1 = A.dims[1];
    const int ldx = K0 + 1;                              // odd stride: conflict-free A-operand reads
    float *X = lds;                                      // [HP][K0 + 1] normalised, concatenated rows
    const int ldh = A.width + 1;                         // A.width = widest layer output, rounded up to 32
    float *P = X + HP * ldx;                             // [4][HP][ldh] partial first-layer tiles
    float *H0 = P + 4 * HP * ldh;                        // [HP][ldh] activations (ping)
    float *H1 = H0 + HP * ldh;                           // [HP][ldh] activations (pong)
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const int fi = lane & 31, fh = lane >> 5;
    const int p0 = blockIdx.x * HP;
    const int d = A.d;

    // Requests that depend on nothing go first, so their L2 round trips run under the gather's:
    // this wave's first-layer weight operands (common shape: <= 32 outputs, <= 64 k per wave) and
    // the tail layers' weights (a few hundred floats) into LDS.
    const int kq = K0 >> 2;
    const bool early = N1 <= 32 && kq <= 64;
    const float *wb = wfrag_base(A, wid, fh, fi);
    float bpre[32];
    if (early) {
#pragma unroll
        for (int u = 0; u < 32; ++u) bpre[u] = (fi < N1 && 2 * u < kq) ? wb[(int64_t)(2 * u) * N1] : 0.f;
    }
    float *TW = H1 + HP * ldh;                           // tail-layer weights, layer after layer (A.w_in_lds)
    if (A.w_in_lds) {
        int off = 0;
        for (int l = 1; l < A.num_layers; ++l) {
            const int cnt = A.dims[l] * A.dims[l + 1];
            for (int i = tid; i < cnt; i += 256) TW[off + i] = A.w[l][i];
            off += cnt;
        }
    }

    // ---- gather + normalise: wave w owns pairs 8 w .. 8 w + 7 ------------------------------------------------
    {
        constexpr int CH = 4;                            // d <= 256: up to 4 elements per lane per row
        float v[8][2][CH];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pair = p0 + 8 * wid + i;
            const bool live = pair < A.B;
            const float *z1 = A.z + (live ? A.ids1[pair] : 0) * A.ldz;
            const float *z2 = A.z + (live ? A.ids2[pair] : 0) * A.ldz;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int col = lane + WAVE * c;
                v[i][0][c] = (live && col < d) ? z1[col] : 0.f;
                v[i][1][c] = (live && col < d) ? z2[col] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) { s1 = fmaf(v[i][0][c], v[i][0][c], s1); s2 = fmaf(v[i][1][c], v[i][1][c], s2); }
            s1 = group_sum(s1, WAVE);
            s2 = group_sum(s2, WAVE);
            const float n1 = fmaxf(sqrtf(s1), A.eps), n2 = fmaxf(sqrtf(s2), A.eps);
            float *row = X + (8 * wid + i) * ldx;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int col = lane + WAVE * c;
                if (col < d) { row[col] = v[i][0][c] / n1; row[d + col] = v[i][1][c] / n2; }
            }
        }
    }
    __syncthreads();

    // ---- first layer on the matrix cores: wave w multiplies k in [w K0/4, (w+1) K0/4) ---------------------------
    {
        const int NT = (N1 + 31) >> 5;
        float *dst = P + wid * HP * ldh;
        const float *xa = X + fi * ldx + wid * kq + fh;
        if (early) {
            f32x16h acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const float a = 2 * u < kq ? xa[2 * u] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bpre[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * fh) * ldh + fi] = acc[r];
        } else if (NT == 1) first_layer_tiles<1>(xa, wb, kq, N1, fi, fh, dst, ldh);
        else if (NT == 2) first_layer_tiles<2>(xa, wb, kq, N1, fi, fh, dst, ldh);
        else first_layer_tiles<4>(xa, wb, kq, N1, fi, fh, dst, ldh);
    }
    __syncthreads();
    const bool only = A.num_layers == 1;
    for (int idx = tid; idx < HP * N1; idx += 256) {     // partial sums (fixed order) + bias + activation
        const int pr = idx / N1, o = idx - pr * N1;
        const int off = pr * ldh + o;
        float s = ((P[off] + P[HP * ldh + off]) + P[2 * HP * ldh + off]) + P[3 * HP * ldh + off];
        s = apply_act(s + (A.b[0] ? A.b[0][o] : 0.f), only ? A.final_act : A.hidden_act, 0.f);
        if (only) { if (p0 + pr < A.B) A.out[(int64_t)(p0 + pr) * A.ldo + o] = s; }
        else H0[off] = s;
    }
    // ---- tail layers: one thread per (pair, output) -----------------------------------------------------------
    float *in = H0, *outb = H1;
    int tw_off = 0;
    for (int l = 1; l < A.num_layers; ++l) {
        __syncthreads();
        const int K = A.dims[l], N = A.dims[l + 1];
        const bool last = l == A.num_layers - 1;
        const float *W = A.w_in_lds ? TW + tw_off : A.w[l];
        tw_off += K * N;
        for (int idx = tid; idx < HP * N; idx += 256) {
            const int pr = idx / N, o = idx - pr * N;
            float s0 = 0.f, s1 = 0.f;
            int k = 0;
            for (; k + 1 < K; k += 2) {
                s0 = fmaf(in[pr * ldh + k], W[(int64_t)k * N + o], s0);
                s1 = fmaf(in[pr * ldh + k + 1], W[(int64_t)(k + 1) * N + o], s1);
            }
            if (k < K) s0 = fmaf(in[pr * ldh + k], W[(int64_t)k * N + o], s0);
            const float vv = apply_act(s0 + s1 + (A.b[l] ? A.b[l][o] : 0.f), last ? A.final_act : A.hidden_act, 0.f);
            if (last) { if (p0 + pr < A.B) A.out[(int64_t)(p0 + pr) * A.ldo + o] = vv; }
            else outb[pr * ldh + o] = vv;
        }
        float *t = in; in = outb; outb = t;
    }
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_link_pred_head(const float *z, int64_t ldz, const int64_t *ids1, const int64_t *ids2,
                                    int32_t B, int32_t d, float eps, int32_t num_layers,
                                    const float *const *w_host, const float *const *b_host,
                                    const int32_t *dims_host, int32_t hidden_act, int32_t final_act,
                                    float *out, int64_t ldo, void *stream) {
    BLGNN_CHECK_ARG(B >= 0 && d > 0, "link_pred_head: bad sizes");
    if (B == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(z && ids1 && ids2 && out && w_host && b_host && dims_host, "link_pred_head: null pointer");
    BLGNN_CHECK_ARG(num_layers >= 1 && num_layers <= HEAD_MAX_LAYERS, "link_pred_head: 1..4 layers");
    BLGNN_CHECK_ARG(dims_host[0] == 2 * d, "link_pred_head: dims[0] must be 2*d");
    BLGNN_CHECK_ARG(ldz >= d && ldo >= dims_host[num_layers], "link_pred_head: leading dimension too small");
    BLGNN_CHECK_ARG(hidden_act >= 0 && hidden_act <= BLGNN_ACT_TANH && hidden_act != BLGNN_ACT_PRELU &&
                    final_act >= 0 && final_act <= BLGNN_ACT_TANH && final_act != BLGNN_ACT_PRELU,
                    "link_pred_head: bad activation id");
    HeadArgs A = {};
    int width = 0;
    for (int l = 0; l <= num_layers; ++l) {
        BLGNN_CHECK_ARG(dims_host[l] > 0, "link_pred_head: bad layer width");
        A.dims[l] = dims_host[l];
        if (l < num_layers) {
            BLGNN_CHECK_ARG(w_host[l], "link_pred_head: null weight");
            A.w[l] = w_host[l]; A.b[l] = b_host[l];
            width = dims_host[l] > width ? dims_host[l] : width;
        }
    }
    width = (width + 3) & ~3;
    if (width > HEAD_MAX_WIDTH)
        return fail(BLGNN_ERR_UNSUPPORTED, "link_pred_head: layer width %d > %d; use pair_gather_normalize + linear",
                    width, HEAD_MAX_WIDTH);
    A.z = z; A.ldz = ldz; A.ids1 = ids1; A.ids2 = ids2; A.B = B; A.d = d; A.eps = eps;
    A.num_layers = num_layers; A.hidden_act = hidden_act; A.final_act = final_act;
    A.out = out; A.ldo = ldo; A.width = width;
    // matrix-core variant: first layer <= 512 -> <= 128, every later layer <= 128 wide
    bool mfma_ok = d <= 256 && dims_host[0] % 8 == 0 && dims_host[0] <= HK_MAX && dims_host[1] <= HN_MAX;
    for (int l = 2; l <= num_layers; ++l) mfma_ok = mfma_ok && dims_host[l] <= HN_MAX;
    if (mfma_ok) {
        static size_t declared = 0;
        int wmax = 0;
        for (int l = 1; l <= num_layers; ++l) wmax = dims_host[l] > wmax ? dims_host[l] : wmax;
        A.width = (wmax + 31) & ~31;
        size_t tail = 0;
        for (int l = 1; l < num_layers; ++l) tail += (size_t)dims_host[l] * dims_host[l + 1];
        A.w_in_lds = tail <= 4096;                       // tail-layer weights staged in LDS (<= 16 KB)
        const size_t mbytes = sizeof(float) * ((size_t)HP * (dims_host[0] + 1) + 6 * (size_t)HP * (A.width + 1) +
                                               (A.w_in_lds ? tail : 0));
        if (mbytes > declared && mbytes > 64 * 1024) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_link_head_mfma),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)mbytes) != hipSuccess)
                return fail(BLGNN_ERR_LAUNCH, "link_pred_head: cannot raise the dynamic LDS limit to %zu bytes", mbytes);
            declared = mbytes;
        }
        hipLaunchKernelGGL(k_link_head_mfma, dim3((unsigned)((B + HP - 1) / HP)), dim3(256), mbytes,
                           (hipStream_t)stream, A);
        return launch_status("link_pred_head");
    }
    size_t wfloats = 0;
    for (int l = 0; l < num_layers; ++l) wfloats += ((size_t)dims_host[l] * dims_host[l + 1] + 3) & ~(size_t)3;
    const size_t act_bytes = sizeof(float) * 4 * 2 * (size_t)width;
    A.w_in_lds = (act_bytes + wfloats * sizeof(float)) <= 60 * 1024;
    const size_t bytes = act_bytes + (A.w_in_lds ? wfloats * sizeof(float) : 0);
    // one pair per wavefront whenever the batch allows it: a pair is a chain of dependent
    // latencies (ids -> rows -> three layers), so pairs must run side by side, not in sequence
    int64_t blocks = ((int64_t)B + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_link_head, dim3((unsigned)blocks), dim3(256), bytes, (hipStream_t)stream, A);
    return launch_status("link_pred_head");
}
