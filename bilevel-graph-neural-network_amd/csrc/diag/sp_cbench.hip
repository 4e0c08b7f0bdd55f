// standalone check + micro-benchmark for the pre-split f16x3 GEMM (diagnostics only)
#include <hip/hip_runtime.h>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include "../linear_sp.hip"
#include "../runtime.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static double check(int M, int K, int N, int act, bool with_rv) {
    std::vector<float> hx((size_t)M * K), hw((size_t)K * N), hb(N), hrv(M), hrw(N), hsc(N), hsh(N);
    srand(7);
    for (int m = 0; m < M; ++m) {
        float rs = powf(10.f, (rand() % 9) - 4);                      // rows 1e-4 .. 1e4
        for (int k = 0; k < K; ++k) {
            float bs = ((k / 256) % 2 == 1 && m % 3 == 0) ? 1e-3f : 1.f;   // blocks three decades apart in some rows
            if (m % 7 == 0 && k / 256 == 1) bs = 0.f;                   // an all-zero block
            hx[(size_t)m * K + k] = rs * bs * ((rand() % 2001 - 1000) / 1000.f);
        }
    }
    for (auto &v : hw) v = (rand() % 2001 - 1000) / 3000.f;
    for (int n = 0; n < N; ++n) { hb[n] = (rand() % 201 - 100) / 100.f; hrw[n] = (rand() % 201 - 100) / 100.f; hsc[n] = 0.5f + (rand() % 100) / 100.f; hsh[n] = (rand() % 201 - 100) / 200.f; }
    for (int m = 0; m < M; ++m) hrv[m] = (float)(rand() % 5);
    float *x, *w, *b, *rv, *rw, *sc, *sh, *y, *yj; uint8_t *xe, *ye; void *xs, *ys, *wp;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&rv, M * 4));
    CK(hipMalloc(&rw, N * 4)); CK(hipMalloc(&sc, N * 4)); CK(hipMalloc(&sh, N * 4));
    CK(hipMalloc(&y, (size_t)M * N * 4)); CK(hipMalloc(&yj, (size_t)M * N * 4));
    CK(hipMalloc(&xs, (size_t)M * K * 4)); CK(hipMalloc(&ys, (size_t)M * N * 4));
    CK(hipMalloc(&xe, (size_t)M * 8)); CK(hipMalloc(&ye, (size_t)M * 8));
    CK(hipMalloc(&wp, blgnn_packed_weight_sp_elems(K, N) * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(rv, hrv.data(), M * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rw, hrw.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sc, hsc.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(sh, hsh.data(), N * 4, hipMemcpyHostToDevice));
    int rc = blgnn_pack_weight_sp(w, N, K, N, wp, nullptr);
    rc |= blgnn_split_rows_sp(x, K, M, K, xs, 2 * K, xe, 8, nullptr);
    rc |= blgnn_linear_sp(xs, 2 * K, xe, 8, wp, y, N, ys, 2 * N, ye, 8, M, N, K, b, with_rv ? rv : nullptr,
                          with_rv ? rw : nullptr, act, nullptr, sc, sh, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr);
    rc |= blgnn_join_rows_sp(ys, 2 * N, ye, 8, M, N, yj, N, nullptr);
    if (rc) { printf("rc=%d %s\n", rc, blgnn_last_error()); exit(1); }
    CK(hipDeviceSynchronize());
    std::vector<float> hy((size_t)M * N), hyj((size_t)M * N), hxj((size_t)M * K);
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hyj.data(), yj, hy.size() * 4, hipMemcpyDeviceToHost));
    // split -> join round trip of X itself
    rc = blgnn_join_rows_sp(xs, 2 * K, xe, 8, M, K, x, K, nullptr);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hxj.data(), x, hxj.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_j = 0, worst_x = 0;
    for (int m = 0; m < M; ++m) {
        double rowmax = 0; std::vector<double> ref(N);
        for (int n = 0; n < N; ++n) {
            double a = 0;
            for (int k = 0; k < K; ++k) a += (double)hx[(size_t)m * K + k] * hw[(size_t)k * N + n];
            a += hb[n] + (with_rv ? (double)hrv[m] * hrw[n] : 0.0);
            if (act == 1) a = a > 0 ? a : 0;
            a = a * hsc[n] + hsh[n];
            ref[n] = a; rowmax = std::max(rowmax, fabs(a));
        }
        rowmax = std::max(rowmax, 1.0);
        for (int n = 0; n < N; ++n) {
            worst = std::max(worst, fabs(hy[(size_t)m * N + n] - ref[n]) / rowmax);
            worst_j = std::max(worst_j, fabs(hyj[(size_t)m * N + n] - ref[n]) / rowmax);
        }
        double xm = 1e-30;
        for (int k = 0; k < K; ++k) xm = std::max(xm, (double)fabs(hx[(size_t)m * K + k]));
        for (int k = 0; k < K; ++k) worst_x = std::max(worst_x, fabs((double)hxj[(size_t)m * K + k] - hx[(size_t)m * K + k]) / xm);
    }
    printf("check M=%d K=%d N=%d act=%d rv=%d: y err %.3e, sp-y err %.3e, x round trip %.3e (row-relative)\n", M, K, N, act, (int)with_rv, worst, worst_j, worst_x);
    hipFree(x); hipFree(w); hipFree(b); hipFree(rv); hipFree(rw); hipFree(sc); hipFree(sh); hipFree(y); hipFree(yj); hipFree(xs); hipFree(ys); hipFree(xe); hipFree(ye); hipFree(wp);
    return std::max(worst, worst_j);
}

static void bench(int M, int K, int N, int mode) {
    const int pad = getenv("SP_PAD") ? atoi(getenv("SP_PAD")) : 0;
    const int64_t LX = 2 * K + pad, LY = 2 * N + pad;
    float *x, *y = nullptr, *w; uint8_t *xe, *ye; void *xs, *ys = nullptr, *wp;
    CK(hipMalloc(&x, (size_t)M * K * 4)); CK(hipMalloc(&w, (size_t)K * N * 4));
    CK(hipMalloc(&xs, (size_t)M * LX * 2)); CK(hipMalloc(&xe, (size_t)M * 8)); CK(hipMalloc(&ye, (size_t)M * 8));
    if (mode & 1) CK(hipMalloc(&y, (size_t)M * N * 4));
    if (mode & 2) CK(hipMalloc(&ys, (size_t)M * LY * 2));
    std::vector<float> h((size_t)4 << 20);
    for (auto &v : h) v = (rand() % 2001 - 1000) / 1000.f;
    for (size_t o = 0; o < (size_t)M * K; o += h.size()) CK(hipMemcpy(x + o, h.data(), std::min(h.size(), (size_t)M * K - o) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&wp, blgnn_packed_weight_sp_elems(K, N) * 2));
    blgnn_pack_weight_sp(w, N, K, N, wp, nullptr);
    blgnn_split_rows_sp(x, K, M, K, xs, LX, xe, 8, nullptr);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() { return blgnn_linear_sp(xs, LX, xe, 8, wp, y, N, ys, LY, ye, 8, M, N, K, nullptr, nullptr, nullptr, 1, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, nullptr); };
    for (int it = 0; it < 3; ++it) run();
    CK(hipDeviceSynchronize());
    const int R = 10;
    hipEventRecord(e0);
    for (int it = 0; it < R; ++it) { int rc = run(); if (rc) { printf("rc=%d %s\n", rc, blgnn_last_error()); exit(1); } }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double us = ms * 1e3 / R, tf = 6.0 * M * (double)K * N / (us * 1e-6) / 1e12;
    printf("bench M=%d K=%d N=%d out=%s: %.1f us/launch, %.0f TF executed (%.3f of 2.5 PF)\n", M, K, N,
           mode == 1 ? "f32" : mode == 2 ? "sp" : "f32+sp", us, tf, tf / 2500.0);
#ifdef SP_STAMP
    {
        unsigned long long st[8];
        hipMemcpyFromSymbol(st, HIP_SYMBOL(blgnn::g_sp_stamps), sizeof(st));
        const double cyc = (double)(st[2] - st[0]), ticks = (double)(st[3] - st[1]);
        printf("   main loop: %.0f cycles = %.0f per K step of 32, %.2f us, in-kernel clock %.2f GHz; epilogue %.0f cycles, %.2f us\n", cyc, cyc / (K / 32),
               ticks / 100.0, cyc / ticks * 0.1, (double)(st[4] - st[2]), (double)(st[5] - st[3]) / 100.0);
        printf("   prologue (entry -> first K step): %.0f cycles, %.2f us; whole tile %.2f us\n", (double)(st[0] - st[6]), (double)(st[1] - st[7]) / 100.0, (double)(st[5] - st[7]) / 100.0);
    }
#endif
    // split pass alone
    hipEventRecord(e0);
    for (int it = 0; it < R; ++it) blgnn_split_rows_sp(x, K, M, K, xs, LX, xe, 8, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("   split_rows_sp of X: %.1f us\n", ms * 1e3 / R);
    hipFree(x); hipFree(w); hipFree(xs); hipFree(xe); hipFree(ye); if (y) hipFree(y); if (ys) hipFree(ys); hipFree(wp);
}

int main(int argc, char **argv) {
    if (argc > 5 && atoi(argv[1]) == 2) { bench(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5])); return 0; }
    double e = 0;
    e = std::max(e, check(300, 256, 256, 0, false));
    e = std::max(e, check(1000, 512, 512, 1, true));
    e = std::max(e, check(515, 1024, 256, 1, false));
    printf("worst %.3e %s\n", e, e < 1e-5 ? "OK" : "FAIL");
    if (argc > 1 && atoi(argv[1]) == 0) return 0;
    const int M = argc > 2 ? atoi(argv[2]) : 478540;
    bench(M, 256, 1024, 1);
    bench(M, 1024, 1024, 2);
    bench(M, 1024, 256, 1);
    bench(M, 1024, 256, 3);
    bench(M, 1024, 1024, 1);
    return 0;
}
