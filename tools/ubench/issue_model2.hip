// Issue-model microbenchmark #2 for gfx950 (MI355X): do VALU / transcendental instructions overlap with MFMAs on ONE SIMD?
//
// Three experiments, every instruction pinned by `asm volatile` (the compiler may not reorder, merge or drop them):
//   A. same wave:    ITER x { NM x [ 1 MFMA, K fillers ] } with 1 or 2 waves per SIMD -> cycles per MFMA as K grows.  If the pipes
//                    overlap, cycles/MFMA stays at the bare MFMA interval until the fillers' own issue time exceeds it (max, not sum).
//   B. role split:   512-thread blocks = two waves per SIMD; waves 0-3 issue ONLY MFMAs, waves 4-7 ONLY fillers.  Each half is
//                    timed alone (partner exits at once) and together.  Separate pipes -> together ~ max(alone_mfma, alone_valu).
//   C. mixed-shape accumulate chain: acc = mfma_16x16x16(a16, b16, 0); acc = mfma_16x16x32(a32, b32, acc) back to back on one
//                    accumulator vs the same products through separate accumulators; counts mismatching lanes (round-1 claim).
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_model2 issue_model2.hip ; run: ./issue_model2 > profiles/rNN_issue_model2.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// SHAPE 0: v_mfma_f32_16x16x32_bf16 (4 accumulator regs), 1: v_mfma_f32_32x32x16_bf16 (16 accumulator regs)
// KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_cvt_pk_bf16_f32, 3: v_max3_f32, 4: mix per 4 fillers = {exp, exp, max3, cvt_pk}
template <int KIND>
__device__ __forceinline__ void filler(float (&x)[16], int j, float c1, float c2) {
    float& r = x[j & 15];
    const int kind = KIND == 4 ? ((j & 3) < 2 ? 1 : ((j & 3) == 2 ? 3 : 2)) : KIND;
    if (kind == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
    else if (kind == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
    else if (kind == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(c1));
    else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
}

template <int SHAPE>
struct Acc;
template <>
struct Acc<0> {
    f32x4 a[4];
    __device__ __forceinline__ void init() { for (int i = 0; i < 4; ++i) a[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ void mfma(int m, bf16x8 A, bf16x8 B) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(a[m & 3]) : "v"(A), "v"(B));
    }
    __device__ __forceinline__ float sum() {
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += a[i][0] + a[i][1] + a[i][2] + a[i][3];
        return s;
    }
};
template <>
struct Acc<1> {
    f32x16 a[2];
    __device__ __forceinline__ void init() { for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) a[i][j] = 0.f; }
    __device__ __forceinline__ void mfma(int m, bf16x8 A, bf16x8 B) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(a[m & 1]) : "v"(A), "v"(B));
    }
    __device__ __forceinline__ float sum() {
        float s = 0.f;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += a[i][j];
        return s;
    }
};

constexpr int NM = 8;  // MFMAs per loop iteration

// ---- A: fillers in the MFMA gaps of the SAME wave
template <int SHAPE, int KIND, int K>
__global__ __launch_bounds__(512) void same_wave(float* out, unsigned long long* cyc, int iters) {
    Acc<SHAPE> acc;
    acc.init();
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)((threadIdx.x & 7) * 0.01f + i * 0.001f); B[i] = (__bf16)(i * 0.01f); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i * 1e-2f;
    const float c1 = 0.999f, c2 = 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            acc.mfma(m, A, B);
#pragma unroll
            for (int j = 0; j < K; ++j) filler<KIND>(x, m * K + j, c1, c2);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = acc.sum();
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// ---- B: role split, waves 0-3 MFMA only, waves 4-7 fillers only (KV fillers per MFMA-equivalent slot)
template <int SHAPE, int KIND, int KV>
__global__ __launch_bounds__(512) void role_split(float* out, unsigned long long* cyc, int iters, int mode) {
    Acc<SHAPE> acc;
    acc.init();
    bf16x8 A, B;
    for (int i = 0; i < 8; ++i) { A[i] = (__bf16)((threadIdx.x & 7) * 0.01f + i * 0.001f); B[i] = (__bf16)(i * 0.01f); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-3f + i * 1e-2f;
    const float c1 = 0.999f, c2 = 1e-3f;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < NM; ++m) acc.mfma(m, A, B);
            }
        }
    } else {
        if (mode & 2) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int j = 0; j < NM * KV; ++j) filler<KIND>(x, j, c1, c2);
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = acc.sum();
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// ---- C: mixed-shape accumulate chain
__global__ void mixed_chain(float* out_chain, float* out_sep, int pad) {
    const int lane = threadIdx.x & 63;
    s16x4 a16, b16;
    bf16x8 a32, b32;
    for (int i = 0; i < 4; ++i) {
        union { __bf16 b; short s; } u, v;
        u.b = (__bf16)(0.25f + 0.125f * ((lane + i) & 7));
        v.b = (__bf16)(0.5f - 0.0625f * ((lane * 3 + i) & 7));
        a16[i] = u.s; b16[i] = v.s;
    }
    for (int i = 0; i < 8; ++i) { a32[i] = (__bf16)(0.125f * ((lane + 2 * i) & 15) - 0.5f); b32[i] = (__bf16)(0.0625f * ((lane * 5 + i) & 15)); }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    // chained on ONE accumulator (compiler inserts whatever hazard padding it believes is needed)
    f32x4 c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, z, 0, 0, 0);
    if (pad) asm volatile("s_nop 15" : "+v"(c));
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, c, 0, 0, 0);
    // reverse order chain
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, z, 0, 0, 0);
    if (pad) asm volatile("s_nop 15" : "+v"(d));
    d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, d, 0, 0, 0);
    // separate accumulators, VALU sum
    f32x4 e = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, z, 0, 0, 0);
    f32x4 f = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, z, 0, 0, 0);
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(e), "+v"(f));
    for (int r = 0; r < 4; ++r) {
        out_chain[(threadIdx.x * 4 + r) * 2 + 0] = c[r];
        out_chain[(threadIdx.x * 4 + r) * 2 + 1] = d[r];
        out_sep[threadIdx.x * 4 + r] = e[r] + f[r];
    }
}

static double ghz = 0.0;

template <int SHAPE, int KIND, int K>
void run_same(const char* kind, int waves_per_simd) {
    float* out; unsigned long long* cyc;
    const int nthreads = 256 * waves_per_simd, nb = 256, iters = 4000;
    CHECK(hipMalloc(&out, (size_t)nb * nthreads * 4)); CHECK(hipMalloc(&cyc, (size_t)nb * 8 * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((same_wave<SHAPE, KIND, K>), dim3(nb), dim3(nthreads), 0, 0, out, cyc, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL((same_wave<SHAPE, KIND, K>), dim3(nb), dim3(nthreads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nb * nthreads / 64);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
    // per-SIMD cycles per MFMA: each SIMD hosts waves_per_simd waves that each issued iters*NM MFMAs
    const double per_mfma_wave = avg / ((double)iters * NM);
    const double per_mfma_simd = per_mfma_wave / waves_per_simd;
    const double wall_cyc = ms * 1e-3 * ghz * 1e9 / ((double)iters * NM * waves_per_simd);
    printf("A shape=%s waves/SIMD=%d filler=%-6s K=%2d : %7.2f cyc/MFMA per wave, %7.2f cyc/MFMA per SIMD (s_memtime) | wall %.2f cyc/MFMA/SIMD @%.2f GHz\n",
           SHAPE ? "32x32x16" : "16x16x32", waves_per_simd, kind, K, per_mfma_wave, per_mfma_simd, wall_cyc, ghz);
    hipFree(out); hipFree(cyc);
}

template <int SHAPE, int KIND, int KV>
void run_role(const char* kind) {
    float* out; unsigned long long* cyc;
    const int nb = 256, iters = 4000;
    CHECK(hipMalloc(&out, (size_t)nb * 512 * 4)); CHECK(hipMalloc(&cyc, (size_t)nb * 8 * 8));
    double res[4][2] = {};
    for (int mode = 1; mode <= 3; ++mode) {
        hipLaunchKernelGGL((role_split<SHAPE, KIND, KV>), dim3(nb), dim3(512), 0, 0, out, cyc, 200, mode);
        hipLaunchKernelGGL((role_split<SHAPE, KIND, KV>), dim3(nb), dim3(512), 0, 0, out, cyc, iters, mode);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(nb * 8);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        double a = 0, b = 0;
        for (int i = 0; i < nb; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)h[i * 8 + w];
        res[mode][0] = a / (nb * 4) / ((double)iters * NM);
        res[mode][1] = b / (nb * 4) / ((double)iters * NM);
    }
    printf("B shape=%s filler=%-6s %2d fillers per MFMA slot: MFMA half alone %6.2f cyc/MFMA | VALU half alone %6.2f cyc/slot | together: MFMA half %6.2f, VALU half %6.2f  (sum %.2f, max %.2f)\n",
           SHAPE ? "32x32x16" : "16x16x32", kind, KV, res[1][0], res[2][1], res[3][0], res[3][1], res[1][0] + res[2][1], std::max(res[1][0], res[2][1]));
    hipFree(out); hipFree(cyc);
}

__global__ void clock_probe(unsigned long long* o) {
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < 2000000; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (threadIdx.x == 0) { o[0] = c1 - c0; o[1] = w1 - w0; }
    if (x == 12345.f) o[2] = 1;
}

int main() {
    unsigned long long* d; CHECK(hipMalloc(&d, 64));
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, d);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[2]; CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    ghz = (double)h[0] / ((double)h[1] / 100e6) / 1e9;  // s_memrealtime runs at 100 MHz
    printf("s_memtime ticks per second (idle chip, one wave): %.3f GHz\n", ghz);

    printf("\n== A. fillers in the MFMA gaps of the same wave (NM=%d MFMAs per iteration, independent accumulators) ==\n", NM);
#define ROW(S, KIND, NAME) \
    run_same<S, KIND, 0>(NAME, 1); run_same<S, KIND, 1>(NAME, 1); run_same<S, KIND, 2>(NAME, 1); run_same<S, KIND, 3>(NAME, 1); \
    run_same<S, KIND, 4>(NAME, 1); run_same<S, KIND, 6>(NAME, 1); run_same<S, KIND, 8>(NAME, 1); run_same<S, KIND, 12>(NAME, 1); \
    run_same<S, KIND, 0>(NAME, 2); run_same<S, KIND, 1>(NAME, 2); run_same<S, KIND, 2>(NAME, 2); run_same<S, KIND, 3>(NAME, 2); \
    run_same<S, KIND, 4>(NAME, 2); run_same<S, KIND, 6>(NAME, 2); run_same<S, KIND, 8>(NAME, 2); run_same<S, KIND, 12>(NAME, 2);
    ROW(0, 0, "fma") ROW(0, 1, "exp") ROW(0, 4, "mix")
    ROW(1, 0, "fma") ROW(1, 1, "exp") ROW(1, 4, "mix")
    run_same<0, 2, 4>("cvt_pk", 1); run_same<0, 3, 4>("max3", 1); run_same<0, 2, 4>("cvt_pk", 2); run_same<0, 3, 4>("max3", 2);

    printf("\n== B. role split: waves 0-3 MFMA only, waves 4-7 fillers only (two waves per SIMD) ==\n");
    run_role<0, 0, 2>("fma"); run_role<0, 0, 4>("fma"); run_role<0, 0, 8>("fma");
    run_role<0, 1, 1>("exp"); run_role<0, 1, 2>("exp"); run_role<0, 1, 4>("exp");
    run_role<0, 4, 4>("mix"); run_role<0, 4, 8>("mix");
    run_role<1, 0, 4>("fma"); run_role<1, 0, 8>("fma"); run_role<1, 0, 16>("fma");
    run_role<1, 1, 2>("exp"); run_role<1, 1, 4>("exp"); run_role<1, 1, 8>("exp");
    run_role<1, 4, 8>("mix"); run_role<1, 4, 16>("mix");

    printf("\n== C. 16x16x16 + 16x16x32 MFMA on one accumulator ==\n");
    float *oc, *os; CHECK(hipMalloc(&oc, 64 * 4 * 2 * 4)); CHECK(hipMalloc(&os, 64 * 4 * 4));
    for (int pad = 0; pad < 2; ++pad) {
        int bad_c = 0, bad_d = 0;
        for (int rep = 0; rep < 200; ++rep) {
            hipLaunchKernelGGL(mixed_chain, dim3(1), dim3(64), 0, 0, oc, os, pad);
            CHECK(hipDeviceSynchronize());
            float hc[512], hs[256];
            CHECK(hipMemcpy(hc, oc, sizeof(hc), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hs, os, sizeof(hs), hipMemcpyDeviceToHost));
            for (int i = 0; i < 256; ++i) { bad_c += hc[2 * i] != hs[i]; bad_d += hc[2 * i + 1] != hs[i]; }
        }
        printf("C pad=%d: chain(16 then 32) mismatches %d / 51200, chain(32 then 16) mismatches %d / 51200 (vs separate accumulators + VALU add)\n", pad, bad_c, bad_d);
    }
    return 0;
}
