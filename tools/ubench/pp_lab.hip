// Ping-pong lab (round 4): standalone harness around anyedit_amd/csrc/gemm_conv.hip (no Python / torch: a gpurun visit costs seconds).
// Times the 192x320-tile launches of the UNet at batch 12 under AE_GEMM_PP (read from the environment by the launcher) and, in
// -DAE_GEMM_LAB builds, prints where one SIMD's two waves (wave 0 = group 0, wave 4 = group 1) of a mid-grid block spend their cycles per K tile.
// Variants are compile-time macros of gemm_conv.hip: -DAE_PP_LAB=1|2|3 (ablations: results are wrong by construction), -DAE_PP_DMA_FIRST=1,
// -DAE_PP_PRIO=0, ...   Parity is covered by tests/ (the product build).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize [-DAE_GEMM_LAB] [-D...] -I anyedit_amd/csrc -o tools/ubench/build/pp_lab tools/ubench/pp_lab.hip
#include "../../anyedit_amd/csrc/gemm_conv.hip"
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <stdarg.h>

// (the row-panel kernel lives in another translation unit of the library: this lab links gemm_conv.hip alone, its K = 320 fold shapes are not exercised here)
int ae_rowpanel_fold_covers(int, int, int, int, int) { return 0; }
int ae_rowpanel_fold_launch(const void*, long, const void*, long, void*, long, int, int, int, const float*, const void*, long, int, float*, const float*, int, const float*, float, void*) { return AE_ERR_UNSUPPORTED; }
void ae_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int ae_check_launch(const char* what) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return AE_ERR_LAUNCH; } return AE_OK; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

static void fill(std::vector<uint16_t>& v, float scale, unsigned seed) {
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    // 64 Ki distinct values repeated: the host fill of 100 M normals would cost more than the GPU visit
    std::vector<uint16_t> pool(65536 + 17);
    for (auto& x : pool) x = f2bf(nd(rng) * scale);
    for (size_t i = 0; i < v.size(); ++i) v[i] = pool[(i * 2654435761u >> 7) % pool.size()];
}

template <typename F>
static void timed(const char* tag, double flops, F&& launch) {
    for (int i = 0; i < 3; ++i) if (launch() != AE_OK) { printf("%s: launch failed\n", tag); return; }
    CK(hipDeviceSynchronize());
    unsigned long long zero[32] = {0}, dbg[32] = {0};
#ifdef AE_GEMM_LAB
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_dbg), zero, sizeof(zero)));
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = getenv("AE_LAB_ITERS") ? atoi(getenv("AE_LAB_ITERS")) : 20;   // AE_LAB_ITERS=20000: a multi-second run of one shape for tools/throttle_probe.py
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
#ifdef AE_GEMM_LAB
    CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(g_gemm_dbg), sizeof(dbg)));
#endif
    const double us = 1e3 * ms / iters;
    printf("%-34s %8.1f us %7.1f TFLOP/s\n", tag, us, flops / us / 1e6);
    for (int w = 0; w < 2 && dbg[5]; ++w) {
        const unsigned long long* d = dbg + 16 * w;
        const double kt = (double)d[5];
        printf("    wave %d per K tile: L0 %5.0f bar %5.0f | M0 %5.0f bar %5.0f | L1 %5.0f vmcnt %5.0f bar %5.0f | M1 %5.0f bar %5.0f  = %6.0f cycles  (K tiles %.0f; prologue %.0f, epilogue %.0f cycles per launch)\n",
               4 * w, d[1] / kt, d[2] / kt, d[3] / kt, d[8] / kt, d[9] / kt, d[10] / kt, d[11] / kt, d[12] / kt, d[13] / kt,
               (d[1] + d[2] + d[3] + d[8] + d[9] + d[10] + d[11] + d[12] + d[13]) / kt, kt / iters, (double)d[0] / iters, (double)(d[6] + d[7] + d[4]) / iters);
    }
#ifdef AE_GEMM_TRACE
    {   // timeline of the traced block (last launch): per wave and K tile, cycles relative to the first stamp
        unsigned long long tr[8 * TR_T * 16];
        CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_pp_trace), sizeof(tr)));
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 8; ++w) if (tr[(w * TR_T) * 16 + 0] && tr[(w * TR_T) * 16 + 0] < t0) t0 = tr[(w * TR_T) * 16 + 0];
        if (t0 != ~0ull) {
            printf("    trace (cycles since the first stamp): L0start | L0done bar | M0done bar | L1reads L1vmcnt bar | M1done bar\n");
            const int ids[10] = {0, 1, 2, 3, 8, 9, 10, 11, 12, 13};
            for (int k = 0; k < TR_T; ++k)
                for (int w = 0; w < 8; ++w) {
                    printf("    kt %2d wave %d:", TR_K0 + k, w);
                    for (int q = 0; q < 10; ++q) { if (q == 1 || q == 3 || q == 5 || q == 8) printf(" |"); printf(" %6lld", (long long)(tr[(w * TR_T + k) * 16 + ids[q]] - t0)); }
                    printf("\n");
                }
        }
        unsigned long long z[8 * TR_T * 16] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), z, sizeof(z)));
    }
#endif
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

static void conv(int B, int H, int Cin, int Cout, int korder, const char* tag) {
    const size_t nx = (size_t)B * H * H * Cin, nw = (size_t)Cout * 9 * Cin, ny = (size_t)B * H * H * Cout;
    std::vector<uint16_t> hx(nx), hw(nw);
    fill(hx, 1.0f, 3 + H + Cin); fill(hw, 0.02f, 5 + Cin);
    uint16_t *dx, *dw, *dy; float *dbias, *ws = nullptr;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy, ny * 2)); CK(hipMalloc(&dbias, Cout * 4));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbias, 0, Cout * 4));
    const long wsf = ae_conv3x3_workspace_floats(B, H, H, Cin, Cout, 1, 0);
    if (wsf) CK(hipMalloc(&ws, wsf * 4));
    timed(tag, 2.0 * B * H * H * (double)Cout * 9 * Cin,
          [&]() { return ae_conv3x3_bf16(dx, dw, dbias, nullptr, 0, nullptr, dy, B, H, H, Cin, Cout, 1, 0, 0, ws, nullptr, korder, nullptr); });
    hipFree(dx); hipFree(dw); hipFree(dy); hipFree(dbias); if (ws) hipFree(ws);
}

static void dense(int M, int N, int K, int epi, const char* tag) {
    std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
    fill(ha, 1.0f, 7 + K); fill(hw, 0.03f, 11 + N);
    const int NO = epi == EPI_GEGLU ? N / 2 : N;
    uint16_t *da, *dw, *dc; float* dbias;
    CK(hipMalloc(&da, ha.size() * 2)); CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dc, (size_t)M * NO * 2)); CK(hipMalloc(&dbias, N * 4));
    CK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dbias, 0, N * 4));
    timed(tag, 2.0 * M * (double)N * K,
          [&]() { return ae_gemm_bf16(da, K, nullptr, 0, 0, dw, K, dc, NO, M, N, K, dbias, nullptr, 0, nullptr, 0, 0, epi, 0, nullptr, nullptr); });
    hipFree(da); hipFree(dw); hipFree(dc); hipFree(dbias);
}

int main(int argc, char** argv) {
    const char* pp = getenv("AE_GEMM_PP");
    printf("# AE_GEMM_PP=%s  AE_PP_LAB=%d AE_PP_PRIO=%d AE_PP_DMA_FIRST=%d\n", pp ? pp : "(default)", AE_PP_LAB, AE_PP_PRIO, AE_PP_DMA_FIRST);
    const bool all = argc > 1 && argv[1][0] == 'x';
    if (argc > 1 && argv[1][0] == 'c') {   // one case only (PMC passes)
        conv(12, 64, 960, 320, 1, "conv L1 960->320 @64 kmajor");
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'p') {   // one short-K dense launch only (round 5: is this family power-limited? tools/throttle_probe.py beside AE_LAB_ITERS=200000)
        dense(12288, 640, 640, EPI_NONE, "gemm proj L2 12288x640x640");
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'g') {   // one GEGLU launch only
        dense(12288, 5120, 640, EPI_GEGLU, "gemm ff1 L2 geglu 12288x5120x640");
        return 0;
    }
    conv(12, 64, 320, 320, 1, "conv L1 320->320 @64 kmajor");
    conv(12, 64, 960, 320, 1, "conv L1 960->320 @64 kmajor");
    conv(12, 16, 1280, 1280, 0, "conv L3 1280->1280 @16 splitK");
    dense(49152, 320, 1280, EPI_NONE, "gemm ff2 L1 49152x320x1280");
    if (argc > 1 && argv[1][0] == 'm') {   // the 32x32-level shapes (128x128 tile today)
        conv(12, 32, 640, 640, 0, "conv L2 640->640 @32");
        conv(12, 32, 1920, 640, 0, "conv L2 1920->640 @32");
        dense(12288, 640, 2560, EPI_NONE, "gemm ff2 L2 12288x640x2560");
        dense(12288, 1920, 640, EPI_NONE, "gemm qkv L2 12288x1920x640");
        dense(12288, 640, 640, EPI_NONE, "gemm proj L2 12288x640x640");
        return 0;
    }
    if (all) {
        conv(12, 16, 2560, 1280, 0, "conv L3 2560->1280 @16 splitK");
        dense(12288, 5120, 640, EPI_GEGLU, "gemm ff1 L2 geglu 12288x5120x640");
        dense(3072, 10240, 1280, EPI_GEGLU, "gemm ff1 L3 geglu 3072x10240x1280");
    }
    return 0;
}
