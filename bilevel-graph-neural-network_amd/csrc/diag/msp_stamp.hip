// diagnostics: launch time and one wavefront's phase stamps of the sp attention kernel (csrc/gmn_match_sp.hip) on
// synthetic pairs: msp_stamp [n1] [n2] [pairs]
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#define MSP_STAMP
#include "../gmn_match_sp.hip"
#include "../runtime.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
    const int n1 = argc > 1 ? atoi(argv[1]) : 45, n2 = argc > 2 ? atoi(argv[2]) : 45, P = argc > 3 ? atoi(argv[3]) : 4208;
    const int d = 256;
    std::vector<int> seg(2 * P + 1), order(P);
    for (int k = 0; k < P; ++k) { seg[2 * k + 1] = seg[2 * k] + n1; seg[2 * k + 2] = seg[2 * k + 1] + n2; order[k] = k; }
    const int64_t n = seg[2 * P];
    const int big = n1 > n2 ? n1 : n2;
    int counts[4] = {big <= 32 ? P : 0, big <= 32 ? 0 : P, 0, 0};
    int *dseg, *dord; _Float16 *xs, *us; uint8_t *xe, *ue;
    const int64_t ld = 8 * d;                                       // planes of an [n, 4 d] sp matrix, as in the propagator
    CK(hipMalloc(&dseg, seg.size() * 4)); CK(hipMalloc(&dord, P * 4));
    CK(hipMemcpy(dseg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dord, order.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&xs, n * ld * 2)); CK(hipMalloc(&us, n * ld * 2)); CK(hipMalloc(&xe, n * 8)); CK(hipMalloc(&ue, n * 8));
    std::vector<_Float16> h((size_t)1 << 22);
    for (auto &v : h) v = (_Float16)((rand() % 2001 - 1000) * 16.f);
    for (int64_t o = 0; o < n * ld; o += (int64_t)h.size())
        CK(hipMemcpy(xs + o, h.data(), std::min<int64_t>(h.size(), n * ld - o) * 2, hipMemcpyHostToDevice));
    CK(hipMemset(xe, 120, n * 8));
    auto run = [&]() { return blgnn_gmn_match_sp(dseg, P, xs, ld, xe, 8, nullptr, n, us + 6 * d, ld, ue, 8, d, dord, counts, (big > 32 && big <= 48 && !getenv("MSP_R64")) ? P : 0, nullptr); };
    for (int i = 0; i < 3; ++i) if (run()) { printf("rc %s\n", blgnn_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0); for (int i = 0; i < reps; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n * 1024 * 3;
    printf("gmn_match_sp %d pairs of %d x %d nodes, d = 256: %.1f us per launch (%.2f TB/s of x read twice + u written)\n", P, n1, n2,
           ms * 1000.0 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
    unsigned long long st[16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(blgnn::g_msp_stamps), sizeof(st));
    auto us_ = [&](int a, int b) { return (double)(st[b] - st[a]) / 100.0; };
    printf("  prologue (zero fill, offsets, exponents) %.2f us\n  pass 1 (8 chunks)  %.2f us   [chunk 3: wait %.2f]\n  softmax %.2f us\n"
           "  pass 2 (8 chunks) %.2f us   [chunk 3: wait %.2f]\n  total %.2f us\n",
           us_(0, 1), us_(1, 2), us_(7, 8), us_(2, 3), us_(3, 4), us_(5, 6), us_(0, 4));
    return 0;
}
