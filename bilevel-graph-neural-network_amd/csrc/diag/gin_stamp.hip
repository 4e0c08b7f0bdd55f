// diagnostics: phase stamps of the fused GIN layer at d = 256 (one workgroup's timeline)
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#define GIN_STAMP
#define GIN_OLD256          // also build round 3's one-workgroup-per-CU kernel (selected by the environment variable GIN_OLD256)
#include "../gin_h3.hip"
#include "../runtime.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
int main(int argc, char **argv) {
    const int per = argc > 1 ? atoi(argv[1]) : 40;              // atoms per molecule (64: no molecule straddles a 64-row tile)
    const int N = argc > 2 ? atoi(argv[2]) : 256, K = N;        // build with -DGIN_STAMP_N=128 for the d = 128 kernel
    const int n = (argc > 3 ? atoi(argv[3]) : 498729) / per * per;
    std::vector<int> rp(n + 1), col;
    for (int i = 0; i < n; ++i) {                        // chains of 40 atoms + a few ring closures: ~2.1 edges per atom
        rp[i] = (int)col.size();
        const int b = i / per * per, l = i % per;
        if (l > 0) col.push_back(i - 1);
        if (l < per - 1) col.push_back(i + 1);
        if (l % 7 == 0 && l + 5 < per) col.push_back(i + 5);
        if (l % 7 == 5 && l - 5 >= 0 && (l - 5) % 7 == 0) col.push_back(i - 5);
        (void)b;
    }
    rp[n] = (int)col.size();
    int *drp, *dcol; float *x, *y, *w, *b; void *w1p, *w2p;
    CK(hipMalloc(&drp, (n + 1) * 4)); CK(hipMalloc(&dcol, col.size() * 4));
    CK(hipMemcpy(drp, rp.data(), (n + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&x, (size_t)n * K * 4)); CK(hipMalloc(&y, (size_t)n * N * 4)); CK(hipMalloc(&w, (size_t)K * N * 4)); CK(hipMalloc(&b, N * 4));
    std::vector<float> h((size_t)1 << 22);
    for (auto &v : h) v = (rand() % 2001 - 1000) / 1000.f;
    for (size_t o = 0; o < (size_t)n * K; o += h.size()) CK(hipMemcpy(x + o, h.data(), std::min(h.size(), (size_t)n * K - o) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice)); CK(hipMemset(b, 0, N * 4));
    CK(hipMalloc(&w1p, blgnn_packed_weight_h3_elems(K, N) * 2)); CK(hipMalloc(&w2p, blgnn_packed_weight_h3_elems(N, N) * 2));
    blgnn_pack_weight_h3(w, N, K, N, w1p, nullptr); blgnn_pack_weight_h3(w, N, N, N, w2p, nullptr);
    auto run = [&]() { return blgnn_gin_layer_h3(drp, dcol, (int64_t)col.size(), x, K, n, K, w1p, b, w2p, b, N, 0.1f, 1, nullptr, nullptr, nullptr, y, N,
                                                nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, 0, nullptr); };
    std::vector<float> y_new((size_t)n * N), y_old((size_t)n * N);
    for (int variant = 0; variant < (N == 256 ? 2 : 1); ++variant) {
        if (variant) setenv("GIN_OLD256", "1", 1); else unsetenv("GIN_OLD256");
        CK(hipMemset(y, 0, (size_t)n * N * 4));
        for (int i = 0; i < 3; ++i) if (run()) { printf("rc %s\n", blgnn_last_error()); return 1; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy((variant ? y_old : y_new).data(), y, (size_t)n * N * 4, hipMemcpyDeviceToHost));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); for (int i = 0; i < 20; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("gin d=%d %d rows (%s): %.1f us per launch\n", N, n, variant ? "one 512-thread workgroup per CU, round 3" : (N == 256 ? "two 256-thread workgroups per CU" : "d = 128 kernel"), ms * 50.0);
        unsigned long long st[16];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(blgnn::g_gin_stamps), sizeof(st));
        if (variant || N != 256) {
            const char *names[] = {"entry", "staged (loads landed)", "gathered+split", "hmax atomics (GEMM1 + act done)", "hidden split parked", "GEMM2 + staged", "", "", "stores issued", "GEMM1 done", "GEMM2 done"};
            const int order[] = {0, 1, 2, 9, 3, 4, 10, 5, 8};
            for (int k = 1; k < 9; ++k)
                printf("  %-34s +%.2f us (at %.2f)\n", names[order[k]], (double)(st[order[k]] - st[order[k - 1]]) / 100.0, (double)(st[order[k]] - st[0]) / 100.0);
        } else {
            const char *names[] = {"entry", "staged (loads landed)", "gathered into registers", "pieces parked", "GEMM1 done", "act + row maxima", "hidden pieces parked", "GEMM2 done", "output tile staged", "stores issued"};
            for (int k = 1; k < 10; ++k)
                printf("  %-34s +%.2f us (at %.2f)\n", names[k], (double)(st[k] - st[k - 1]) / 100.0, (double)(st[k] - st[0]) / 100.0);
        }
    }
    if (N == 256) {
        size_t bad = 0; double worst = 0;
        for (size_t i = 0; i < y_new.size(); ++i) if (y_new[i] != y_old[i]) { ++bad; worst = std::max(worst, (double)fabsf(y_new[i] - y_old[i])); }
        printf("new vs round-3 kernel: %zu of %zu values differ (max |d| %.3g)\n", bad, y_new.size(), worst);
    }
    return 0;
}
