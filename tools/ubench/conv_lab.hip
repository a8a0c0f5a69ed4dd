// Conv lab: standalone harness around anyedit_amd/csrc/gemm_conv.hip built with -DAE_GEMM_LAB (no Python / torch).  Times the UNet's 3x3
// conv shapes at batch 12 and prints where one SIMD's two waves of a mid-grid block spend their cycles per K tile:
// DMA issue / LDS reads + MFMAs / barrier + DMA drain, plus prologue and epilogue.  Parity is covered by tests/test_hip_ops.py.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAE_GEMM_LAB -I anyedit_amd/csrc -o tools/ubench/build/conv_lab tools/ubench/conv_lab.hip
//   without -DAE_GEMM_LAB: plain timing of the shipped kernels; with -DAE_GEMM_LAB_NOEPI: the same kernels without their epilogue (ablation)
#include "../../anyedit_amd/csrc/gemm_conv.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <stdarg.h>

int ae_rowpanel_fold_covers(int, int, int, int, int) { return 0; }
int ae_rowpanel_fold_launch(const void*, long, const void*, long, void*, long, int, int, int, const float*, const void*, long, int, float*, const float*, int, const float*, float, void*) { return AE_ERR_UNSUPPORTED; }
void ae_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ae_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return AE_ERR_LAUNCH; } return AE_OK; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

static void run(int B, int H, int Cin, int Cout, const char* tag) {
    std::mt19937 rng(3 + H + Cin);
    std::normal_distribution<float> nd(0.f, 1.f);
    const size_t nx = (size_t)B * H * H * Cin, nw = (size_t)Cout * 9 * Cin, ny = (size_t)B * H * H * Cout;
    std::vector<uint16_t> hx(nx), hw(nw);
    for (auto& v : hx) v = f2bf(nd(rng));
    for (auto& v : hw) v = f2bf(nd(rng) * 0.02f);
    uint16_t *dx, *dw, *dy; float *dbias, *ws = nullptr;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy, ny * 2)); CK(hipMalloc(&dbias, Cout * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbias, 0, Cout * 4));
    const long wsf = ae_conv3x3_workspace_floats(B, H, H, Cin, Cout, 1, 0);
    if (wsf) CK(hipMalloc(&ws, wsf * 4));
    auto launch = [&]() { return ae_conv3x3_bf16(dx, dw, dbias, nullptr, 0, nullptr, dy, B, H, H, Cin, Cout, 1, 0, 0, ws, nullptr); };
    for (int i = 0; i < 3; ++i) if (launch() != AE_OK) { printf("launch failed\n"); exit(1); }
    CK(hipDeviceSynchronize());
    unsigned long long zero[16] = {0}, dbg[16] = {0};
#ifdef AE_GEMM_LAB
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg), zero, sizeof(zero)));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#ifdef AE_GEMM_LAB
    CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(g_gemm_dbg), sizeof(dbg)));
#endif
    const double us = 1e3 * ms / iters, fl = 2.0 * B * H * H * (double)Cout * 9 * Cin;
    printf("%-28s %8.1f us %7.1f TFLOP/s  split-K workspace %ld floats\n", tag, us, fl / us / 1e6, wsf);
    for (int w = 0; w < 2 && dbg[5]; ++w) {   // buckets only in the -DAE_GEMM_LAB build
        const unsigned long long* d = dbg + 8 * w;
        const double kt = d[5] ? (double)d[5] : 1.0;
        printf("    wave %d: per K tile: DMA issue %6.0f  LDS+MFMA %6.0f  barrier+drain %6.0f cycles   (K tiles/launch %.0f)\n             per launch: prologue %.0f, epilogue staging %.0f + output %.0f + tail %.0f cycles\n",
               4 * w, d[1] / kt, d[2] / kt, d[3] / kt, kt / iters, (double)d[0] / iters, (double)d[6] / iters, (double)d[7] / iters, (double)d[4] / iters);
    }
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dbias); if (ws) hipFree(ws);
}

int main() {
    run(12, 64, 320, 320, "L1 320->320 @64 (192x320)");
    run(12, 64, 960, 320, "L1 960->320 @64 (192x320)");
    run(12, 32, 640, 640, "L2 640->640 @32 (128x128)");
    run(12, 32, 1920, 640, "L2 1920->640 @32 (128x128)");
    run(12, 16, 1280, 1280, "L3 1280->1280 @16 (split)");
    return 0;
}
