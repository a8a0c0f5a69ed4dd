// Sustained-clock calibration: a pure MFMA loop on every SIMD of the chip for several seconds, so that tools/throttle_probe.py can
// read which limiter the firmware reports and at which clock the matrix pipes settle.  Operands: random bf16 (default) or zeros ("z") — the guide's
// DVFS note says the same binary ran +19 % on zero-filled inputs, i.e. the sustained clock depends on the data toggling, not on the instruction stream.
//   mfma_burn [seconds=5] [kind: 0 = 16x16x32 bf16, 1 = 32x32x16 bf16] [z]
// Prints per launch batch: wall time, MFMA rate, shader cycles / wall (effective clock from s_memtime against the 100 MHz wall clock).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/build/mfma_burn tools/ubench/mfma_burn.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <int KIND>
__global__ __launch_bounds__(256) void burn(const unsigned* seed, float* out, unsigned long long* clk, int iters) {
    // 8 independent accumulators per wave, operands re-derived from a per-lane seed (random data: every multiplier input toggles)
    bf16x8 a[4], b[4];
    for (int j = 0; j < 4; ++j) {
        union { unsigned u[4]; bf16x8 v; } ua, ub;
        for (int i = 0; i < 4; ++i) {
            const unsigned s0 = seed[(threadIdx.x * 8 + j * 2 + 0) * 4 + i], s1 = seed[(threadIdx.x * 8 + j * 2 + 1) * 4 + i];
            ua.u[i] = s0; ub.u[i] = s1;
        }
        a[j] = ua.v; b[j] = ub.v;
    }
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float s = 0.f;
    if (KIND == 0) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 16; ++m) acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m % 4], b[(m / 4) % 4], acc[m % 8], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m % 4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m % 4], b[(m / 2) % 4], acc[m % 4], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 7) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    const bool zeros = argc > 3 && argv[3][0] == 'z';
    unsigned* seed; float* out; unsigned long long* clk;
    hipMalloc(&seed, 256 * 8 * 4 * 4); hipMalloc(&out, 1 << 22); hipMalloc(&clk, 16);
    unsigned h[256 * 8 * 4];
    unsigned x = 12345u;
    for (auto& v : h) {
        // two bf16 values in [0.5, 2) with random mantissas and signs: finite products, accumulators stay finite over any run length? no — they grow;
        // magnitudes around 2^-8 keep 10^9 accumulations below fp32 overflow
        x = x * 1664525u + 1013904223u; const unsigned lo = 0x3B80u | ((x >> 9) & 0x7Fu) | ((x >> 3) & 0x8000u);
        x = x * 1664525u + 1013904223u; const unsigned hi = 0x3B80u | ((x >> 9) & 0x7Fu) | ((x >> 3) & 0x8000u);
        v = zeros ? 0u : (lo | (hi << 16));
    }
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 200000;   // 16 (or 8) MFMAs per iteration: ~25 ms per launch
    const double flop_per_launch = 1024.0 /* blocks */ * 4 /* waves */ * iters * (kind == 0 ? 16 * 16.0 * 16 * 32 * 2 : 8 * 32.0 * 32 * 16 * 2);
    printf("# mfma_burn kind=%s data=%s seconds=%.1f\n", kind == 0 ? "16x16x32" : "32x32x16", zeros ? "zeros" : "random", seconds);
    auto t_start = std::chrono::steady_clock::now();
    int batch = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < seconds) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        const int nl = 8;
        for (int i = 0; i < nl; ++i) {
            if (kind == 0) hipLaunchKernelGGL((burn<0>), dim3(1024), dim3(256), 0, 0, seed, out, clk, iters);
            else hipLaunchKernelGGL((burn<1>), dim3(1024), dim3(256), 0, 0, seed, out, clk, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
        printf("batch %2d  t=%5.2f s  %7.1f ms  %7.1f TFLOP/s  block clock %.3f GHz (shader cycles / 100 MHz wall ticks)\n", batch++,
               std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(), ms, nl * flop_per_launch / (ms * 1e-3) / 1e12,
               hc[1] ? (double)hc[0] / (double)hc[1] * 0.1 : 0.0);
        fflush(stdout);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    return 0;
}
