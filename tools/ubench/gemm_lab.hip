// Row-panel GEMM lab: standalone harness around anyedit_amd/csrc/gemm_rowpanel.hip (no Python / torch: a gpurun visit costs seconds).
// Checks sampled rows against an fp64 CPU reference (LayerNorm prologue, bias, residual, GEGLU), then times the level-1 UNet shapes.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I anyedit_amd/csrc -o tools/ubench/build/gemm_lab tools/ubench/gemm_lab.hip
#include "../../anyedit_amd/csrc/gemm_rowpanel.hip"
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include <cstring>
#include <stdarg.h>

void ae_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ae_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return AE_ERR_LAUNCH; } return AE_OK; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

static void run(int M, int N, int K, int epi, bool ln, bool res, int iters) {
    std::mt19937 rng(7 + M + N);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<uint16_t> A((size_t)M * K), W((size_t)N * K), R;
    for (auto& x : A) x = f2bf(nd(rng) * 1.3f + 0.4f);
    for (auto& x : W) x = f2bf(nd(rng) * 0.06f);
    std::vector<float> bias(N), g(K), b(K);
    for (auto& x : bias) x = nd(rng) * 0.1f;
    for (auto& x : g) x = 1.0f + 0.1f * nd(rng);
    for (auto& x : b) x = 0.1f * nd(rng);
    const int NO = epi == RP_EPI_GEGLU ? N / 2 : N;
    if (res) { R.resize((size_t)M * NO); for (auto& x : R) x = f2bf(nd(rng)); }
    uint16_t *dA, *dW, *dC, *dR = nullptr; float *dbias, *dg, *db;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dC, (size_t)M * NO * 2));
    CK(hipMalloc(&dbias, N * 4)); CK(hipMalloc(&dg, K * 4)); CK(hipMalloc(&db, K * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), K * 4, hipMemcpyHostToDevice));
    if (res) { CK(hipMalloc(&dR, R.size() * 2)); CK(hipMemcpy(dR, R.data(), R.size() * 2, hipMemcpyHostToDevice)); }
    CK(hipMemset(dC, 0xff, (size_t)M * NO * 2));
    auto launch = [&]() { return ae_ln_gemm_bf16(dA, K, dW, K, dC, NO, M, N, K, dbias, dR, NO, ln ? dg : nullptr, ln ? db : nullptr, 1e-5f, epi, 0); };
    if (launch() != AE_OK) { printf("launch failed\n"); exit(1); }
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> C((size_t)M * NO);
    CK(hipMemcpy(C.data(), dC, C.size() * 2, hipMemcpyDeviceToHost));
    // reference on sampled rows (first / last block edges + random)
    std::vector<int> rows;
    for (int r = 0; r < std::min(M, 200); ++r) rows.push_back(r);
    for (int r = std::max(0, M - 200); r < M; ++r) rows.push_back(r);
    for (int i = 0; i < 64; ++i) rows.push_back((int)(rng() % M));
    double num = 0, den = 0, maxabs = 0;
    std::vector<double> a(K), acc(N);
    for (int r : rows) {
        for (int k = 0; k < K; ++k) a[k] = bf2f(A[(size_t)r * K + k]);
        if (ln) {
            double mu = 0, v = 0;
            for (int k = 0; k < K; ++k) mu += a[k];
            mu /= K;
            for (int k = 0; k < K; ++k) v += (a[k] - mu) * (a[k] - mu);
            const double rs = 1.0 / std::sqrt(v / K + 1e-5);
            for (int k = 0; k < K; ++k) a[k] = bf2f(f2bf((float)((a[k] - mu) * rs * g[k] + b[k])));  // the kernel feeds bf16 operands to the MFMA
        }
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += a[k] * bf2f(W[(size_t)n * K + k]);
            acc[n] = s + bias[n];
        }
        for (int j = 0; j < NO; ++j) {
            double ref;
            if (epi == RP_EPI_GEGLU) {
                const int c = j / 16, i = j % 16;
                const double av = acc[c * 32 + i], gv = acc[c * 32 + 16 + i];
                ref = av * 0.5 * gv * (1.0 + std::erf(gv / std::sqrt(2.0)));
            } else {
                ref = acc[j] + (res ? bf2f(R[(size_t)r * NO + j]) : 0.0);
            }
            const double got = bf2f(C[(size_t)r * NO + j]);
            num += (got - ref) * (got - ref); den += ref * ref; maxabs = std::max(maxabs, std::fabs(got - ref));
        }
    }
    const double rel = std::sqrt(num / den);
    printf("check M=%d N=%d K=%d epi=%d ln=%d res=%d : rel-L2 %.3e max-abs %.3e %s\n", M, N, K, epi, ln, res, rel, maxabs, (rel < 4e-3 && rel == rel) ? "OK" : "FAIL");
    if (iters > 0) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(e1);
        CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 2.0 * M * N * K, by = 2.0 * ((double)M * K + (double)N * K + (double)M * NO * (res ? 2 : 1));
        unsigned long long z[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rp_dbg), z, sizeof(z)));
        launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_rp_dbg), sizeof(z)));
        const int nch = N / 64;
        printf("      block 0 wave 0 cycles: prologue %llu | per chunk pair: dma-wait %llu barrier %llu mfma %llu epilogue %llu\n", z[0], z[1] / nch, z[2] / nch, z[3] / nch, z[4] / nch);
        printf("time  M=%d N=%d K=%d epi=%d ln=%d res=%d : %7.1f us  %7.1f TFLOP/s  %7.1f GB/s algorithmic\n", M, N, K, epi, ln, res, us, fl / us / 1e6, by / us / 1e3);
    }
    hipFree(dA); hipFree(dW); hipFree(dC); hipFree(dbias); hipFree(dg); hipFree(db); if (dR) hipFree(dR);
}

int main() {
    run(200, 64, 320, RP_EPI_NONE, false, false, 0);
    run(500, 128, 320, RP_EPI_NONE, true, true, 0);
    run(777, 128, 320, RP_EPI_GEGLU, true, false, 0);
    run(49152, 320, 320, RP_EPI_NONE, false, true, 20);
    run(49152, 320, 320, RP_EPI_NONE, true, false, 20);
    run(49152, 960, 320, RP_EPI_NONE, true, false, 20);
    run(49152, 2560, 320, RP_EPI_GEGLU, true, false, 20);
    run(49152, 2560, 320, RP_EPI_GEGLU, false, false, 20);
    return 0;
}
