// diagnostics: phase stamps of the fused GIN layer at d = 256 (one workgroup's timeline)
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <map>
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
        const int reps = getenv("GIN_REPS") ? atoi(getenv("GIN_REPS")) : 20;      // (sustained load: GIN_REPS=3000 ~ 2 s)
        hipEventRecord(e0); for (int i = 0; i < reps; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("gin d=%d %d rows (%s): %.1f us per launch over %d launches\n", N, n, variant ? "one 512-thread workgroup per CU, round 3" : (N == 256 ? "two 256-thread workgroups per CU" : "d = 128 kernel"), ms * 1000.0 / reps, reps);
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
    if (N == 256 && getenv("GIN_TL")) {
        // per-CU timelines of the last launch of the new kernel?  No: the table holds the LAST launch, which was the round-3
        // kernel's unless GIN_TL_NEW re-runs the new one here
        unsetenv("GIN_OLD256");
        run(); CK(hipDeviceSynchronize());
        static unsigned long long tl[8192][12];
        CK(hipMemcpyFromSymbol(tl, HIP_SYMBOL(blgnn::g_gin_tl), sizeof(tl)));
        const int nb = std::min(8192, (n + 63) / 64);
        // key = (xcc, se, cu); HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
        std::map<unsigned, std::vector<int>> by_cu;
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; ++b) { t0 = std::min(t0, tl[b][0]); const unsigned hw = (unsigned)tl[b][11], xcc = (unsigned)(tl[b][11] >> 32) & 15u;
            by_cu[(xcc << 16) | (hw & 0xff00u)].push_back(b); }
        printf("%zu distinct (xcc, se, sh, cu) keys for %d workgroups\n", by_cu.size(), nb);
        // GEMM intervals: [3,4] and [6,7]; fraction of a workgroup's GEMM time during which another workgroup of its CU is in a GEMM too
        double gemm_total = 0, gemm_shared = 0, span = 0; int shown = 0;
        for (auto &kv : by_cu) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end(), [&](int a, int b) { return tl[a][0] < tl[b][0]; });
            for (size_t i = 0; i < v.size(); ++i) for (int g = 0; g < 2; ++g) {
                const unsigned long long a0 = tl[v[i]][3 + 3 * g], a1 = tl[v[i]][4 + 3 * g];
                gemm_total += (double)(a1 - a0);
                for (size_t j = 0; j < v.size(); ++j) if (j != i) for (int h = 0; h < 2; ++h) {
                    const unsigned long long b0 = tl[v[j]][3 + 3 * h], b1 = tl[v[j]][4 + 3 * h];
                    const unsigned long long lo = std::max(a0, b0), hi = std::min(a1, b1);
                    if (hi > lo) gemm_shared += (double)(hi - lo);
                }
            }
            if (shown < 2) {
                ++shown;
                printf("CU key %06x: %zu workgroups; first ten: block(slot) start..end [gemm1][gemm2] in us\n", kv.first, v.size());
                for (size_t i = 0; i < std::min<size_t>(10, v.size()); ++i) {
                    const int b = v[i];
                    printf("   %5d (w%llu simd%llu) %7.2f..%7.2f  [%7.2f..%7.2f] [%7.2f..%7.2f]\n", b, tl[b][11] & 15, (tl[b][11] >> 4) & 3, (tl[b][0] - t0) / 100.0, (tl[b][9] - t0) / 100.0,
                           (tl[b][3] - t0) / 100.0, (tl[b][4] - t0) / 100.0, (tl[b][6] - t0) / 100.0, (tl[b][7] - t0) / 100.0);
                }
            }
            span = std::max(span, (double)(tl[v.back()][9] - t0));
        }
        printf("GEMM time per workgroup-tile %.2f us; %.0f %% of it beside another GEMM of the same CU; last store at %.1f us\n",
               gemm_total / nb / 100.0, 100.0 * gemm_shared / gemm_total, span / 100.0);
    }
    if (N == 256) {
        size_t bad = 0; double worst = 0;
        for (size_t i = 0; i < y_new.size(); ++i) if (y_new[i] != y_old[i]) { ++bad; worst = std::max(worst, (double)fabsf(y_new[i] - y_old[i])); }
        printf("new vs round-3 kernel: %zu of %zu values differ (max |d| %.3g)\n", bad, y_new.size(), worst);
    }
    return 0;
}
