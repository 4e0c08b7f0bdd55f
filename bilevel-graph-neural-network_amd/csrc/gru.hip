// GRU cell gates for GMNPropagator(f_node='GRU') (model/layers_gmn.py:27-30,48-50; torch.nn.GRUCell):
//   gi = in W_ih^T + b_ih,  gh = h W_hh^T + b_hh   (two blgnn_linear calls, [n, 3D] each, gate order r | z | n)
//   r = sigmoid(gi_r + gh_r),  z = sigmoid(gi_z + gh_z),  c = tanh(gi_n + r * gh_n),  h' = (1 - z) * c + z * h
// One pass over the two gate tables and the state: HBM-bound, 28 D bytes per node.
#include "common.hpp"

namespace blgnn {

__global__ void __launch_bounds__(256) k_gru_gates(const float *__restrict__ gi, int64_t ldgi, const float *__restrict__ gh,
                                                   int64_t ldgh, const float *__restrict__ h, int64_t ldh, int n, int D,
                                                   float *__restrict__ out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * D) return;
    const int i = (int)(idx / D), c = (int)(idx % D);
    const float *a = gi + (int64_t)i * ldgi, *b = gh + (int64_t)i * ldgh;
    const float r = 1.f / (1.f + expf(-(a[c] + b[c])));
    const float z = 1.f / (1.f + expf(-(a[D + c] + b[D + c])));
    const float cand = tanhf(a[2 * D + c] + r * b[2 * D + c]);
    out[(int64_t)i * ldo + c] = (1.f - z) * cand + z * h[(int64_t)i * ldh + c];
}

// Backward of the gate arithmetic (training with f_node='GRU'): the gates are recomputed from the two tables, nothing but
// the tables and the state is kept from the forward.  With r, z, c as above and g = d loss / d h':
//   dc = g (1 - z),  dz = g (h - c),  dh = g z
//   dn = dc (1 - c^2)            -> dgi_n = dn, dgh_n = dn r, dr = dn gh_n
//   dgi_r = dgh_r = dr r (1 - r),  dgi_z = dgh_z = dz z (1 - z)
__global__ void __launch_bounds__(256) k_gru_gates_backward(const float *__restrict__ gi, int64_t ldgi, const float *__restrict__ gh,
                                                            int64_t ldgh, const float *__restrict__ h, int64_t ldh,
                                                            const float *__restrict__ g, int64_t ldg, int n, int D,
                                                            float *__restrict__ dgi, int64_t lddgi, float *__restrict__ dgh,
                                                            int64_t lddgh, float *__restrict__ dh, int64_t lddh) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * D) return;
    const int i = (int)(idx / D), c = (int)(idx % D);
    const float *a = gi + (int64_t)i * ldgi, *b = gh + (int64_t)i * ldgh;
    const float r = 1.f / (1.f + expf(-(a[c] + b[c])));
    const float z = 1.f / (1.f + expf(-(a[D + c] + b[D + c])));
    const float ghn = b[2 * D + c];
    const float cand = tanhf(a[2 * D + c] + r * ghn);
    const float hv = h[(int64_t)i * ldh + c], gv = g[(int64_t)i * ldg + c];
    const float dn = gv * (1.f - z) * (1.f - cand * cand);
    const float dr = dn * ghn * r * (1.f - r);
    const float dz = gv * (hv - cand) * z * (1.f - z);
    float *oa = dgi + (int64_t)i * lddgi, *ob = dgh + (int64_t)i * lddgh;
    oa[c] = dr; ob[c] = dr;
    oa[D + c] = dz; ob[D + c] = dz;
    oa[2 * D + c] = dn; ob[2 * D + c] = dn * r;
    dh[(int64_t)i * lddh + c] = gv * z;
}

// out = a * b elementwise (the gate of GMNAggregatorPairs: model/layers_aggregation.py:88-93)
__global__ void __launch_bounds__(256) k_hadamard(const float *__restrict__ a, int64_t lda, const float *__restrict__ b, int64_t ldb,
                                                  int n, int d, float *__restrict__ out, int64_t ldo) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * d) return;
    const int i = (int)(idx / d), c = (int)(idx % d);
    out[(int64_t)i * ldo + c] = a[(int64_t)i * lda + c] * b[(int64_t)i * ldb + c];
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_hadamard(const float *a, int64_t lda, const float *b, int64_t ldb, int32_t n, int32_t d, float *out,
                              int64_t ldo, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && d > 0 && lda >= d && ldb >= d && ldo >= d, "hadamard: bad sizes");
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(a && b && out, "hadamard: null pointer");
    hipLaunchKernelGGL(k_hadamard, dim3((unsigned)(((int64_t)n * d + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                       ldb, n, d, out, ldo);
    return launch_status("hadamard");
}

extern "C" int blgnn_gru_gates(const float *gi, int64_t ldgi, const float *gh, int64_t ldgh, const float *h, int64_t ldh,
                               int32_t n, int32_t D, float *out, int64_t ldo, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && D > 0 && ldgi >= 3 * D && ldgh >= 3 * D && ldh >= D && ldo >= D, "gru_gates: bad sizes");
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(gi && gh && h && out, "gru_gates: null pointer");
    hipLaunchKernelGGL(k_gru_gates, dim3((unsigned)(((int64_t)n * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gi, ldgi,
                       gh, ldgh, h, ldh, n, D, out, ldo);
    return launch_status("gru_gates");
}

extern "C" int blgnn_gru_gates_backward(const float *gi, int64_t ldgi, const float *gh, int64_t ldgh, const float *h, int64_t ldh,
                                        const float *g, int64_t ldg, int32_t n, int32_t D, float *dgi, int64_t lddgi,
                                        float *dgh, int64_t lddgh, float *dh, int64_t lddh, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && D > 0 && ldgi >= 3 * D && ldgh >= 3 * D && ldh >= D && ldg >= D && lddgi >= 3 * D &&
                    lddgh >= 3 * D && lddh >= D, "gru_gates_backward: bad sizes");
    if (n == 0) return BLGNN_OK;
    BLGNN_CHECK_ARG(gi && gh && h && g && dgi && dgh && dh, "gru_gates_backward: null pointer");
    hipLaunchKernelGGL(k_gru_gates_backward, dim3((unsigned)(((int64_t)n * D + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gi, ldgi, gh, ldgh, h, ldh, g, ldg, n, D, dgi, lddgi, dgh, lddgh, dh, lddh);
    return launch_status("gru_gates_backward");
}
