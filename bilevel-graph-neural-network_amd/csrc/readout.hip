// Node -> graph readout (segment sum / mean over the sorted batch vector), the
// LinkPred front end (row gather + L2 normalise + concat) and row utilities.
// All HBM-bound streaming kernels: 16 B per lane, rows walked four at a time.
#include "common.hpp"

namespace blgnn {

// One 256-thread workgroup per graph.  The four waves take rows r = wave, wave+4, ...
// of the segment; each lane owns float4 columns c4 = lane, lane+64, ...  Partial
// sums meet in LDS ([4][d] floats) and wave 0 writes the reduced row.
// Rows are added in a fixed order -> bitwise reproducible.
constexpr int SEG_MAX_INPUTS = 8;
struct SegInputs {                                     // blockIdx.y selects the input
    const float *x[SEG_MAX_INPUTS];
    int64_t ldx[SEG_MAX_INPUTS];
    int32_t d[SEG_MAX_INPUTS], col0[SEG_MAX_INPUTS], vec[SEG_MAX_INPUTS];
};

__global__ void __launch_bounds__(256)
k_segment_reduce(const int32_t *__restrict__ seg, int32_t G, const SegInputs in, int32_t kind,
                 float *__restrict__ out_base, int64_t ldo, const int32_t *__restrict__ row_map) {
    extern __shared__ float part[];                    // [4][d]
    const float *__restrict__ x = in.x[blockIdx.y];
    const int64_t ldx = in.ldx[blockIdx.y];
    const int d = in.d[blockIdx.y], vec = in.vec[blockIdx.y];
    float *__restrict__ out = out_base + in.col0[blockIdx.y];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    for (int g = blockIdx.x; g < G; g += gridDim.x) {
        const int r0 = seg[g], r1 = seg[g + 1];
        if (vec) {
            const int d4 = d >> 2;
            const int64_t ldx4 = ldx >> 2;
            const float4 *X = reinterpret_cast<const float4 *>(x);
            float4 *P = reinterpret_cast<float4 *>(part) + wid * d4;
            for (int c4 = lane; c4 < d4; c4 += WAVE) {
                float4 a = f4_zero();
                int r = r0 + wid;
                for (; r + 12 < r1; r += 16) {
                    float4 v0 = X[(int64_t)r * ldx4 + c4];
                    float4 v1 = X[(int64_t)(r + 4) * ldx4 + c4];
                    float4 v2 = X[(int64_t)(r + 8) * ldx4 + c4];
                    float4 v3 = X[(int64_t)(r + 12) * ldx4 + c4];
                    f4_add(a, v0); f4_add(a, v1); f4_add(a, v2); f4_add(a, v3);
                }
                for (; r < r1; r += 4) f4_add(a, X[(int64_t)r * ldx4 + c4]);
                P[c4] = a;
            }
        } else {
            for (int c = lane; c < d; c += WAVE) {
                float a = 0.f;
                for (int r = r0 + wid; r < r1; r += 4) a += x[(int64_t)r * ldx + c];
                part[wid * d + c] = a;
            }
        }
        __syncthreads();
        const int orow = row_map ? row_map[g] : g;
        int cnt = r1 - r0;
        const float cntf = (float)(cnt < 1 ? 1 : cnt);   // scatter_mean clamps the count to >= 1
        for (int c = threadIdx.x; c < d; c += blockDim.x) {
            float s = (part[c] + part[d + c]) + (part[2 * d + c] + part[3 * d + c]);
            out[(int64_t)orow * ldo + c] = (kind == BLGNN_REDUCE_MEAN) ? s / cntf : s;
        }
        __syncthreads();
    }
}

// out[b] = [normalize(z[ids1[b]]), normalize(z[ids2[b]])]; one wave per (b, side)
__global__ void __launch_bounds__(256)
k_pair_gather_normalize(const float *__restrict__ z, int64_t ldz, const int64_t *__restrict__ ids1,
                        const int64_t *__restrict__ ids2, int32_t B, int32_t d, float eps,
                        float *__restrict__ out, int64_t ldo) {
    const int lane = threadIdx.x & (WAVE - 1);
    int item = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int step = gridDim.x * blockDim.x / WAVE;
    for (; item < 2 * B; item += step) {
        const int b = item >> 1, side = item & 1;
        const int64_t row = side ? ids2[b] : ids1[b];
        const float *zr = z + row * ldz;
        float ss = 0.f;
        for (int c = lane; c < d; c += WAVE) { float v = zr[c]; ss = fmaf(v, v, ss); }
        ss = group_sum(ss, WAVE);
        const float nrm = fmaxf(sqrtf(ss), eps);
        float *o = out + (int64_t)b * ldo + (int64_t)side * d;
        for (int c = lane; c < d; c += WAVE) o[c] = zr[c] / nrm;
    }
}

__global__ void __launch_bounds__(256)
k_l2_normalize_rows(const float *__restrict__ x, int64_t ldx, float *__restrict__ y, int64_t ldy,
                    int32_t n, int32_t d, float eps) {
    const int lane = threadIdx.x & (WAVE - 1);
    int row = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int step = gridDim.x * blockDim.x / WAVE;
    for (; row < n; row += step) {
        const float *xr = x + (int64_t)row * ldx;
        float ss = 0.f;
        for (int c = lane; c < d; c += WAVE) { float v = xr[c]; ss = fmaf(v, v, ss); }
        ss = group_sum(ss, WAVE);
        const float nrm = fmaxf(sqrtf(ss), eps);
        float *yr = y + (int64_t)row * ldy;
        for (int c = lane; c < d; c += WAVE) yr[c] = xr[c] / nrm;
    }
}

__global__ void __launch_bounds__(256)
k_scatter_rows(const float *__restrict__ src, int64_t lds, const int32_t *__restrict__ row_map,
               float *__restrict__ dst, int64_t ldd, int32_t G, int32_t d) {
    const int lane = threadIdx.x & (WAVE - 1);
    int g = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int step = gridDim.x * blockDim.x / WAVE;
    for (; g < G; g += step) {
        const float *s = src + (int64_t)g * lds;
        float *t = dst + (int64_t)row_map[g] * ldd;
        for (int c = lane; c < d; c += WAVE) t[c] = s[c];
    }
}

// dst row i = src row row_map[i], rows as opaque bytes in units of U (four dwords: 16-byte aligned rows; else bytes).  One
// thread per unit, consecutive threads on consecutive units of a row: reads of a source row repeat (every molecule of a
// pair batch occurs ~5 times) and hit in L2, the writes stream.
typedef unsigned int expand_u4 __attribute__((ext_vector_type(4)));

template <typename U>
__global__ void __launch_bounds__(256)
k_expand_rows(const char *__restrict__ src, int64_t lds, const int32_t *__restrict__ row_map, char *__restrict__ dst,
              int64_t ldd, int64_t rows, int32_t upr) {
    const int64_t total = rows * upr;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / upr;
        const int c = (int)(idx - r * upr);
        const U v = *reinterpret_cast<const U *>(src + (int64_t)row_map[r] * lds + (int64_t)c * sizeof(U));
        __builtin_nontemporal_store(v, reinterpret_cast<U *>(dst + r * ldd + (int64_t)c * sizeof(U)));
    }
}

}  // namespace blgnn

using namespace blgnn;

static int wave_grid(int64_t items) {
    int64_t b = (items + 3) / 4;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

static int segment_reduce_launch(const int32_t *seg, int32_t G, int32_t num, const float *const *xs,
                                 const int64_t *ldxs, const int32_t *ds, int32_t kind, float *out,
                                 int64_t ldo, const int32_t *row_map, hipStream_t stream, const char *what) {
    BLGNN_CHECK_ARG(G >= 0 && num >= 1 && num <= SEG_MAX_INPUTS, "%s: bad G / number of inputs", what);
    BLGNN_CHECK_ARG(kind == BLGNN_REDUCE_SUM || kind == BLGNN_REDUCE_MEAN, "%s: bad kind", what);
    if (G == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(seg && out && xs && ldxs && ds, "%s: null pointer", what);
    SegInputs in = {};
    int dmax = 0, col = 0;
    for (int i = 0; i < num; ++i) {
        BLGNN_CHECK_ARG(ds[i] >= 0 && (ds[i] == 0 || xs[i]) && ldxs[i] >= ds[i], "%s: bad input %d", what, i);
        in.x[i] = xs[i]; in.ldx[i] = ldxs[i]; in.d[i] = ds[i]; in.col0[i] = col;
        in.vec[i] = (ds[i] % 4 == 0) && (ldxs[i] % 4 == 0) && aligned16(xs[i]);
        col += ds[i];
        dmax = ds[i] > dmax ? ds[i] : dmax;
    }
    BLGNN_CHECK_ARG(ldo >= col, "%s: output leading dimension < total width", what);
    if (dmax == 0) return BLGNN_OK;
    if ((size_t)dmax * 16 > 64 * 1024)
        return fail(BLGNN_ERR_UNSUPPORTED, "%s: d=%d needs more than 64 KiB of LDS", what, dmax);
    const int blocks = G < 256 * 16 ? G : 256 * 16;
    hipLaunchKernelGGL(k_segment_reduce, dim3(blocks, num), dim3(256), (size_t)dmax * 16, stream, seg, G,
                       in, kind, out, ldo, row_map);
    return launch_status(what);
}

extern "C" int blgnn_segment_reduce(const int32_t *seg, int32_t G, const float *x, int64_t ldx,
                                    int32_t d, int32_t kind, float *out, int64_t ldo,
                                    const int32_t *row_map, void *stream) {
    return segment_reduce_launch(seg, G, 1, &x, &ldx, &d, kind, out, ldo, row_map, (hipStream_t)stream,
                                 "segment_reduce");
}

extern "C" int blgnn_segment_reduce_multi(const int32_t *seg, int32_t G, int32_t num_inputs,
                                          const float *const *xs_host, const int64_t *ldxs_host,
                                          const int32_t *ds_host, int32_t kind, float *out, int64_t ldo,
                                          const int32_t *row_map, void *stream) {
    return segment_reduce_launch(seg, G, num_inputs, xs_host, ldxs_host, ds_host, kind, out, ldo, row_map,
                                 (hipStream_t)stream, "segment_reduce_multi");
}

extern "C" int blgnn_pair_gather_normalize(const float *z, int64_t ldz, const int64_t *ids1,
                                           const int64_t *ids2, int32_t B, int32_t d, float eps,
                                           float *out, int64_t ldo, void *stream) {
    BLGNN_CHECK_ARG(B >= 0 && d >= 0, "pair_gather_normalize: negative size");
    if (B == 0 || d == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(z && ids1 && ids2 && out, "pair_gather_normalize: null pointer");
    BLGNN_CHECK_ARG(ldz >= d && ldo >= 2 * (int64_t)d, "pair_gather_normalize: bad leading dimension");
    hipLaunchKernelGGL(k_pair_gather_normalize, dim3(wave_grid(2 * (int64_t)B)), dim3(256), 0,
                       (hipStream_t)stream, z, ldz, ids1, ids2, B, d, eps, out, ldo);
    return launch_status("pair_gather_normalize");
}

extern "C" int blgnn_l2_normalize_rows(const float *x, int64_t ldx, float *y, int64_t ldy,
                                       int32_t n, int32_t d, float eps, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && d >= 0, "l2_normalize_rows: negative size");
    if (n == 0 || d == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(x && y && ldx >= d && ldy >= d, "l2_normalize_rows: bad args");
    hipLaunchKernelGGL(k_l2_normalize_rows, dim3(wave_grid(n)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, y, ldy, n, d, eps);
    return launch_status("l2_normalize_rows");
}

extern "C" int blgnn_scatter_rows(const float *src, int64_t lds, const int32_t *row_map, float *dst,
                                  int64_t ldd, int32_t G, int32_t d, void *stream) {
    BLGNN_CHECK_ARG(G >= 0 && d >= 0, "scatter_rows: negative size");
    if (G == 0 || d == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(src && dst && row_map && lds >= d && ldd >= d, "scatter_rows: bad args");
    hipLaunchKernelGGL(k_scatter_rows, dim3(wave_grid(G)), dim3(256), 0, (hipStream_t)stream, src, lds,
                       row_map, dst, ldd, G, d);
    return launch_status("scatter_rows");
}

extern "C" int blgnn_expand_rows(const void *src, int64_t ld_src, const int32_t *row_map, int64_t num_rows, void *dst,
                                 int64_t ld_dst, int64_t row_bytes, void *stream) {
    BLGNN_CHECK_ARG(num_rows >= 0 && row_bytes >= 0, "expand_rows: negative size");
    if (num_rows == 0 || row_bytes == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(src && dst && row_map && ld_src >= row_bytes && ld_dst >= row_bytes, "expand_rows: bad args");
    BLGNN_CHECK_ARG(row_bytes <= (1 << 24), "expand_rows: rows of at most 16 MiB");
    const bool wide = row_bytes % 16 == 0 && ld_src % 16 == 0 && ld_dst % 16 == 0 && aligned16(src) && aligned16(dst);
    const int upr = (int)(wide ? row_bytes / 16 : row_bytes);
    int64_t blocks = (num_rows * upr + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (wide)
        hipLaunchKernelGGL(k_expand_rows<expand_u4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const char *>(src), ld_src, row_map, reinterpret_cast<char *>(dst), ld_dst,
                           num_rows, upr);
    else
        hipLaunchKernelGGL(k_expand_rows<unsigned char>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const char *>(src), ld_src, row_map, reinterpret_cast<char *>(dst), ld_dst,
                           num_rows, upr);
    return launch_status("expand_rows");
}
