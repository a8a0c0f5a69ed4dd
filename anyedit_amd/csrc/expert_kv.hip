// AnySD per-expert adapter K/V projection for the training step, grouped over the samples of a batch (gfx950).
//
// the reference trains `adapter_modules` (train.py:25-28, 410-424, 536-541) inside AnySD.model.MoE, an
// un-pinned submodule, so the operator follows OUR documented spec: every sample b is routed to one expert e_b and
//     kv_ip[b] = ip_rows[b] @ bf16(W[e_b])^T        ip_rows[b]: T image-prompt tokens x Dc,   W: [E, N = 2*inner, Dc] fp32 master.
// Each sample contributes only T (= 4) rows, so these are matrix-VECTOR shaped: HBM-bound on the expert weights (N*Dc*4 bytes per
// sample), no MFMA.  Round 1 ran them as one 64x64-tile GEMM per sample plus per-expert fp32->bf16 conversions, transposes and
// zero-fills: ~40 launches per adapter layer of 5-18 us each.  Here a layer is four launches that read the fp32 masters directly
// (rounded to bf16 in registers, so the forward equals the inference path's packed bf16 weights):
//   forward   Y[b,t,n]   = sum_k X[b,t,k] * W[e_b][n][k]             one wave per 4 output columns, lanes split k (coalesced rows)
//   dgrad     dX[b,t,k]  = sum_n dY[b,t,n] * W[e_b][n][k]            thread per 4 k, N cut into 64-row slices; fp32 slice sums, added in
//                                                                     slice order by a second small launch (bit-reproducible)
//   wgrad     dW[e][n][k] = sum_{b: e_b = e} sum_t dY[b,t,n] X[b,t,k] fp32, every expert written (zeros where nothing was routed)
#include "common.hpp"

namespace {

constexpr int EK_TMAX = 8;    // tokens per sample supported
constexpr int EK_COLS = 4;    // forward: output columns per wave
constexpr int EK_NB = 8;      // wgrad: output rows (n) per block
constexpr int EK_MAXROWS = 256;  // wgrad: B*T rows staged in LDS

__device__ __forceinline__ f32x4 round_bf16x4(f32x4 w) {
    const uint32_t a = pack_bf16x2(w[0], w[1]), b = pack_bf16x2(w[2], w[3]);
    return (f32x4){bf16lo(a), bf16hi(a), bf16lo(b), bf16hi(b)};
}
__device__ __forceinline__ f32x4 load_bf16x4(const bf16_t* p) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(p);
    return (f32x4){bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y)};
}
__device__ __forceinline__ int clamp_expert(int e, int E) { return e < 0 ? 0 : (e >= E ? E - 1 : e); }

// grid (N / 16, B), 256 threads: wave w owns columns 16*blockIdx.x + 4*w .. +3 of sample blockIdx.y
__global__ __launch_bounds__(256) void expert_kv_fwd_kernel(const bf16_t* X, const float* W, const int* experts, bf16_t* Y, int T, int N, int Dc,
                                                            int E) {
    extern __shared__ __attribute__((aligned(16))) float sx[];  // [T][Dc] fp32 copy of the sample's rows
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* xb = X + (long)b * T * Dc;
    for (int i = tid * 4; i < T * Dc; i += 256 * 4) *reinterpret_cast<f32x4*>(sx + i) = load_bf16x4(xb + i);
    __syncthreads();
    const int n0 = blockIdx.x * (4 * EK_COLS) + wave * EK_COLS;
    const float* wb = W + ((long)clamp_expert(experts[b], E) * N + n0) * Dc;
    float acc[EK_COLS][EK_TMAX];
#pragma unroll
    for (int c = 0; c < EK_COLS; ++c)
#pragma unroll
        for (int t = 0; t < EK_TMAX; ++t) acc[c][t] = 0.f;
    for (int k = lane * 4; k < Dc; k += 256) {
        f32x4 w[EK_COLS];
#pragma unroll
        for (int c = 0; c < EK_COLS; ++c) w[c] = round_bf16x4(*reinterpret_cast<const f32x4*>(wb + (long)c * Dc + k));
#pragma unroll
        for (int t = 0; t < EK_TMAX; ++t) {
            if (t >= T) break;
            const f32x4 x = *reinterpret_cast<const f32x4*>(sx + t * Dc + k);
#pragma unroll
            for (int c = 0; c < EK_COLS; ++c) acc[c][t] += x[0] * w[c][0] + x[1] * w[c][1] + x[2] * w[c][2] + x[3] * w[c][3];
        }
    }
#pragma unroll
    for (int t = 0; t < EK_TMAX; ++t) {
        if (t >= T) break;
        float v[EK_COLS];
#pragma unroll
        for (int c = 0; c < EK_COLS; ++c) v[c] = wave_reduce_sum(acc[c][t]);
        if (lane == 0)
            *reinterpret_cast<u32x2*>(Y + ((long)b * T + t) * N + n0) = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}

// grid (S, B), G * Dc / 4 threads (G row groups x one thread per 4 k): block (s, b) owns a 64-row slice of n, row group g the rows
// 16g .. 16g+15 of it; the groups are added through LDS in group order and the slice sum goes to `partial`.
constexpr int EK_SLICE = 64;
__global__ __launch_bounds__(1024) void expert_kv_dgrad_kernel(const bf16_t* dY, const float* W, const int* experts, float* partial, int T, int N,
                                                               int Dc, int E, int G) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [T][EK_SLICE] dy | [G-1][T][Dc] group sums
    float* sdy = smem;
    float* sred = smem + T * EK_SLICE;
    const int s = blockIdx.x, b = blockIdx.y, B = gridDim.y, tid = threadIdx.x;
    const int KQ = Dc / 4, kq = tid % KQ, g = tid / KQ, k = kq * 4;
    const int n_begin = s * EK_SLICE, n_cnt = min(EK_SLICE, N - n_begin);
    for (int i = tid; i < T * EK_SLICE; i += blockDim.x) {
        const int t = i / EK_SLICE, j = i - t * EK_SLICE;
        sdy[i] = j < n_cnt ? bf16_to_f32(dY[((long)b * T + t) * N + n_begin + j]) : 0.f;
    }
    __syncthreads();
    const int per = EK_SLICE / G;
    const int j0 = g * per, j1 = min(j0 + per, n_cnt);
    const float* wb = W + ((long)clamp_expert(experts[b], E) * N + n_begin) * Dc + k;
    f32x4 acc[EK_TMAX];
#pragma unroll
    for (int t = 0; t < EK_TMAX; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int j = j0; j < j1; ++j) {
        const f32x4 w = round_bf16x4(*reinterpret_cast<const f32x4*>(wb + (long)j * Dc));
#pragma unroll
        for (int t = 0; t < EK_TMAX; ++t) {
            if (t >= T) break;
            const float d = sdy[t * EK_SLICE + j];
            acc[t][0] += d * w[0]; acc[t][1] += d * w[1]; acc[t][2] += d * w[2]; acc[t][3] += d * w[3];
        }
    }
    if (g > 0) {
#pragma unroll
        for (int t = 0; t < EK_TMAX; ++t) {
            if (t >= T) break;
            *reinterpret_cast<f32x4*>(sred + ((long)(g - 1) * T + t) * Dc + k) = acc[t];
        }
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int t = 0; t < EK_TMAX; ++t) {
            if (t >= T) break;
            for (int g2 = 1; g2 < G; ++g2) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(sred + ((long)(g2 - 1) * T + t) * Dc + k);
                acc[t][0] += q[0]; acc[t][1] += q[1]; acc[t][2] += q[2]; acc[t][3] += q[3];
            }
            *reinterpret_cast<f32x4*>(partial + (((long)s * B + b) * T + t) * Dc + k) = acc[t];
        }
    }
}

// grid (T, B), Dc / 4 threads: adds the S slice sums of one token row in slice order and rounds to bf16.  (Not a
// last-block-reduces tail inside the slice kernel: the device-scope release that needs is an L2 write-back per block on this
// multi-XCD part — measured on GroupNorm, norm.hip AE_GN_TAIL: ~50 us per launch.)
__global__ void expert_kv_dgrad_reduce_kernel(const float* partial, bf16_t* dX, int S, int T, int Dc) {
    const int t = blockIdx.x, b = blockIdx.y, B = gridDim.y, k = threadIdx.x * 4;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int s2 = 0; s2 < S; ++s2) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(partial + (((long)s2 * B + b) * T + t) * Dc + k);
        v[0] += q[0]; v[1] += q[1]; v[2] += q[2]; v[3] += q[3];
    }
    *reinterpret_cast<u32x2*>(dX + ((long)b * T + t) * Dc + k) = (u32x2){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

// grid (N / EK_NB, E), Dc / 4 threads
__global__ __launch_bounds__(1024) void expert_kv_wgrad_kernel(const bf16_t* dY, const bf16_t* X, const int* experts, float* dW, int B, int T, int N,
                                                               int Dc, int E) {
    __shared__ float sdy[EK_MAXROWS * EK_NB];
    __shared__ int s_sel[EK_MAXROWS];  // 1 where the row's sample is routed to this block's expert
    const int e = blockIdx.y, n0 = blockIdx.x * EK_NB, tid = threadIdx.x, k = tid * 4, R = B * T;
    for (int i = tid; i < R; i += blockDim.x) s_sel[i] = clamp_expert(experts[i / T], E) == e;
    for (int i = tid; i < R * EK_NB; i += blockDim.x) {
        const int r = i / EK_NB, j = i - r * EK_NB;
        sdy[i] = bf16_to_f32(dY[(long)r * N + n0 + j]);
    }
    __syncthreads();
    f32x4 acc[EK_NB];
#pragma unroll
    for (int j = 0; j < EK_NB; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
        if (!s_sel[r]) continue;  // block-uniform
        const f32x4 x = load_bf16x4(X + (long)r * Dc + k);
#pragma unroll
        for (int j = 0; j < EK_NB; ++j) {
            const float d = sdy[r * EK_NB + j];
            acc[j][0] += d * x[0]; acc[j][1] += d * x[1]; acc[j][2] += d * x[2]; acc[j][3] += d * x[3];
        }
    }
#pragma unroll
    for (int j = 0; j < EK_NB; ++j) *reinterpret_cast<f32x4*>(dW + ((long)e * N + n0 + j) * Dc + k) = acc[j];
}

bool shapes_ok(int B, int T, int N, int Dc, int E) {
    return B > 0 && T > 0 && T <= EK_TMAX && N > 0 && N % 16 == 0 && Dc > 0 && Dc % 4 == 0 && Dc / 4 <= 1024 && E > 0;
}

}  // namespace

extern "C" int ae_expert_kv_fwd(const void* x, const float* W, const int* experts, void* y, int B, int T, int N, int Dc, int E, void* stream) {
    AE_REQUIRE(x && W && experts && y, "ae_expert_kv_fwd: null pointer");
    AE_REQUIRE(shapes_ok(B, T, N, Dc, E), "ae_expert_kv_fwd: unsupported shape B=%d T=%d (<= %d) N=%d (%%16) Dc=%d (%%4, <= 4096) E=%d", B, T, EK_TMAX, N,
               Dc, E);
    AE_REQUIRE((long)T * Dc * 4 <= 64 * 1024, "ae_expert_kv_fwd: T*Dc=%ld floats exceed the LDS stage", (long)T * Dc);
    hipLaunchKernelGGL(expert_kv_fwd_kernel, dim3(N / 16, B), dim3(256), (size_t)T * Dc * 4, (hipStream_t)stream, (const bf16_t*)x, W, experts,
                       (bf16_t*)y, T, N, Dc, E);
    return ae_check_launch("ae_expert_kv_fwd");
}

/* number of N slices the dgrad uses for this shape (the partial buffer holds slices*B*T*Dc floats) */
extern "C" int ae_expert_kv_dgrad_slices(int N) { return (N + EK_SLICE - 1) / EK_SLICE; }

extern "C" int ae_expert_kv_dgrad(const void* dy, const float* W, const int* experts, void* dx, int B, int T, int N, int Dc, int E, float* partial,
                                  void* stream) {
    AE_REQUIRE(dy && W && experts && dx && partial, "ae_expert_kv_dgrad: null pointer");
    AE_REQUIRE(shapes_ok(B, T, N, Dc, E), "ae_expert_kv_dgrad: unsupported shape B=%d T=%d N=%d Dc=%d E=%d", B, T, N, Dc, E);
    const int S = ae_expert_kv_dgrad_slices(N);
    int G = 4;  // row groups per block: as many as fit 1024 threads and 64 KiB of LDS
    while (G > 1 && (G * (Dc / 4) > 1024 || ((size_t)T * EK_SLICE + (size_t)(G - 1) * T * Dc) * 4 > 64 * 1024)) G >>= 1;
    const size_t lds = ((size_t)T * EK_SLICE + (size_t)(G - 1) * T * Dc) * 4;
    hipLaunchKernelGGL(expert_kv_dgrad_kernel, dim3(S, B), dim3(G * (Dc / 4)), lds, (hipStream_t)stream, (const bf16_t*)dy, W, experts, partial, T, N,
                       Dc, E, G);
    const int rc = ae_check_launch("ae_expert_kv_dgrad(slices)");
    if (rc) return rc;
    hipLaunchKernelGGL(expert_kv_dgrad_reduce_kernel, dim3(T, B), dim3(Dc / 4), 0, (hipStream_t)stream, partial, (bf16_t*)dx, S, T, Dc);
    return ae_check_launch("ae_expert_kv_dgrad(reduce)");
}

extern "C" int ae_expert_kv_wgrad(const void* dy, const void* x, const int* experts, float* dW, int B, int T, int N, int Dc, int E, void* stream) {
    AE_REQUIRE(dy && x && experts && dW, "ae_expert_kv_wgrad: null pointer");
    AE_REQUIRE(shapes_ok(B, T, N, Dc, E) && N % EK_NB == 0, "ae_expert_kv_wgrad: unsupported shape B=%d T=%d N=%d Dc=%d E=%d", B, T, N, Dc, E);
    AE_REQUIRE(B * T <= EK_MAXROWS, "ae_expert_kv_wgrad: B*T=%d rows exceed %d", B * T, EK_MAXROWS);
    hipLaunchKernelGGL(expert_kv_wgrad_kernel, dim3(N / EK_NB, E), dim3(Dc / 4), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, experts, dW,
                       B, T, N, Dc, E);
    return ae_check_launch("ae_expert_kv_wgrad");
}
