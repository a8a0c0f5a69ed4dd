// Issue-model microbenchmark for gfx950: do MFMA and VALU instructions from the SAME wave / from DIFFERENT waves of one SIMD
// overlap?  Each variant runs ITER iterations of {NM mfma, NV valu (fma or exp)} on W waves per SIMD; prints cycles/iteration.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

template <int NM, int NV, int KIND>  // KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pk_bf16 (via cast), 3: v_max3
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 8], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float& x = v[j % 16];
            if (KIND == 0) x = __builtin_fmaf(x, 1.0001f, 0.5f);
            else if (KIND == 1) x = __builtin_amdgcn_exp2f(x) ;
            else if (KIND == 3) x = __builtin_fmaxf(__builtin_fmaxf(x, v[(j + 1) % 16]), v[(j + 2) % 16]);
        }
    }
    __syncthreads();  // the block's LAST wave ends the interval (the oldest wave always wins issue arbitration)
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// interleaved: after every MFMA, NV / NM VALU instructions (what a software-pipelined loop looks like)
template <int NM, int NV, int KIND>
__global__ __launch_bounds__(1024) void ki(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc[m % 8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 8], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / NM; ++j) {
                float& x = v[(m * (NV / NM) + j) % 16];
                if (KIND == 0) x = __builtin_fmaf(x, 1.0001f, 0.5f);
                else if (KIND == 1) x = __builtin_amdgcn_exp2f(x);
                else if (KIND == 2) x = (j == 0) ? __builtin_amdgcn_exp2f(x) : __builtin_fmaf(x, 1.0001f, 0.5f);
                else if (KIND == 3) x = __builtin_fmaxf(x, v[(m + j + 5) % 16]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NM, int NV, int KIND>
void runi(const char* name, int waves_per_simd) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 2000;
    dim3 grid(256), block(256 * waves_per_simd);
    hipLaunchKernelGGL((ki<NM, NV, KIND>), grid, block, 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((ki<NM, NV, KIND>), grid, block, 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD=%d  %8.1f cycles/iter  (NM=%d NV=%d)\n", name, waves_per_simd, (double)h / iters, NM, NV);
    hipFree(out); hipFree(cyc);
}

template <int NM, int NV, int KIND>
void run(const char* name, int waves_per_simd) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 2000;
    // 256 threads = 4 waves = 1 wave per SIMD; launch `waves_per_simd` blocks per CU on all 256 CUs
    dim3 grid(256), block(256 * waves_per_simd);  // ONE block per CU, waves_per_simd waves on every SIMD
    hipLaunchKernelGGL((k<NM, NV, KIND>), grid, block, 0, 0, out, cyc, iters);
    hipLaunchKernelGGL((k<NM, NV, KIND>), grid, block, 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s waves/SIMD=%d  %8.1f cycles/iter  (NM=%d NV=%d)\n", name, waves_per_simd, (double)h / iters, NM, NV);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<16, 0, 0>("mfma only x16", w);
        runi<16, 16, 0>("mfma + 1 fma", w);
        runi<16, 32, 0>("mfma + 2 fma", w);
        runi<16, 48, 0>("mfma + 3 fma", w);
        runi<16, 64, 0>("mfma + 4 fma", w);
        runi<16, 96, 0>("mfma + 6 fma", w);
        runi<16, 128, 0>("mfma + 8 fma", w);
        runi<16, 16, 1>("mfma + 1 exp", w);
        runi<16, 32, 1>("mfma + 2 exp", w);
        runi<16, 32, 2>("mfma + 1 exp + 1 fma", w);
        runi<16, 48, 2>("mfma + 1 exp + 2 fma", w);
        runi<16, 64, 2>("mfma + 1 exp + 3 fma", w);
        runi<16, 32, 3>("mfma + 2 max", w);
        runi<16, 64, 3>("mfma + 4 max", w);
    }
    return 0;
}
