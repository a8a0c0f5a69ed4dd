// MFMA issue rates on gfx950, one wave per SIMD (saturates the pipe): bf16 16x16x32, fp8 16x16x32, f8f6f4 16x16x128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(8 * sizeof(int)))) int i32x8;

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    long la = 0x3838383838383838L + threadIdx.x, lb = 0x3030303030303030L;
    i32x8 wa, wb;
    for (int i = 0; i < 8; ++i) { wa[i] = 0x38383838 + threadIdx.x; wb[i] = 0x30303030; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (KIND == 0) acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 8], 0, 0, 0);
            else if (KIND == 1) acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(la, lb, acc[m % 8], 0, 0, 0);
            else acc[m % 8] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, wb, acc[m % 8], 0, 0, 0, 0, 0, 0);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char* name, double flops_per_mfma) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double cyc_per = (double)h / iters / 16.0;
    printf("%-28s %6.1f cycles per MFMA  -> %7.1f flop/cycle/SIMD\n", name, cyc_per, flops_per_mfma / cyc_per);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("16x16x32 bf16", 16.0 * 16 * 32 * 2);
    run<1>("16x16x32 fp8", 16.0 * 16 * 32 * 2);
    run<2>("16x16x128 f8f6f4 (fp8)", 16.0 * 16 * 128 * 2);
    return 0;
}
