// Small backward / optimiser kernels of the training step (train.py:625-710) for gfx950.
//
// The frozen UNet is differentiated w.r.t. its ACTIVATIONS only (the trainables are the AnySD adapter projections, the image
// projection and the task embeddings, train.py:483-485), so every layer's backward is a data-gradient:
//   * Linear / 1x1 / 3x3 conv  -> the forward GEMM kernels with transposed / rotated packed weights (ops.py, no new kernel);
//   * GroupNorm / LayerNorm     -> norm.hip;  attention -> attention_bwd.hip;
//   * everything elementwise    -> here: gradient accumulation, GEGLU (un-fused in training so a|g is kept), the adjoint of the
//     nearest-x2 upsample (2x2 sum pool), column sums (bias gradients of the small trainable Linear), eps-MSE gradient, AdamW.
// All are HBM-bound streaming kernels: 16-byte accesses, no reshaping into GEMMs.
#include "common.hpp"

namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void add_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, long n8) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n8; i += (long)gridDim.x * NT) {
        const u32x4 va = reinterpret_cast<const u32x4*>(a)[i], vb = reinterpret_cast<const u32x4*>(b)[i];
        const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16lo(wa[e]) + bf16lo(wb[e]), bf16hi(wa[e]) + bf16hi(wb[e]));
        reinterpret_cast<u32x4*>(y)[i] = (u32x4){o[0], o[1], o[2], o[3]};
    }
}

// y = a + alpha * b (ControlNet residuals: skip + scale * control, cldm.py:336-338, 40-41)
__global__ __launch_bounds__(NT) void axpy_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, long n8, float alpha) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n8; i += (long)gridDim.x * NT) {
        const u32x4 va = reinterpret_cast<const u32x4*>(a)[i], vb = reinterpret_cast<const u32x4*>(b)[i];
        const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16lo(wa[e]) + alpha * bf16lo(wb[e]), bf16hi(wa[e]) + alpha * bf16hi(wb[e]));
        reinterpret_cast<u32x4*>(y)[i] = (u32x4){o[0], o[1], o[2], o[3]};
    }
}

// GEGLU (attention.py:49-57): h = [a | g] (chunk(2, dim=-1)), y = a * gelu(g), exact-erf GELU.
__device__ __forceinline__ float gelu_grad_f(float g) {  // d/dg [g * Phi(g)] = Phi(g) + g * phi(g)
    const float cdf = 0.5f * (1.0f + erf_as_f(g * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * g * g);
    return cdf + g * pdf;
}

__global__ __launch_bounds__(NT) void geglu_fwd_kernel(const bf16_t* h, bf16_t* y, long M, int F) {
    const int f8 = F / 8;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * f8; i += (long)gridDim.x * NT) {
        const long m = i / f8;
        const int c = (int)(i - m * f8) * 8;
        const u32x4 va = *reinterpret_cast<const u32x4*>(h + m * 2 * F + c), vg = *reinterpret_cast<const u32x4*>(h + m * 2 * F + F + c);
        const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wg[4] = {vg.x, vg.y, vg.z, vg.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16lo(wa[e]) * gelu_erf_f(bf16lo(wg[e])), bf16hi(wa[e]) * gelu_erf_f(bf16hi(wg[e])));
        *reinterpret_cast<u32x4*>(y + m * F + c) = (u32x4){o[0], o[1], o[2], o[3]};
    }
}

__global__ __launch_bounds__(NT) void geglu_bwd_kernel(const bf16_t* h, const bf16_t* dy, bf16_t* dh, long M, int F) {
    const int f8 = F / 8;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < M * f8; i += (long)gridDim.x * NT) {
        const long m = i / f8;
        const int c = (int)(i - m * f8) * 8;
        const u32x4 va = *reinterpret_cast<const u32x4*>(h + m * 2 * F + c), vg = *reinterpret_cast<const u32x4*>(h + m * 2 * F + F + c);
        const u32x4 vd = *reinterpret_cast<const u32x4*>(dy + m * F + c);
        const uint32_t wa[4] = {va.x, va.y, va.z, va.w}, wg[4] = {vg.x, vg.y, vg.z, vg.w}, wd[4] = {vd.x, vd.y, vd.z, vd.w};
        uint32_t oa[4], og[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = bf16lo(wa[e]), a1 = bf16hi(wa[e]), g0 = bf16lo(wg[e]), g1 = bf16hi(wg[e]);
            const float d0 = bf16lo(wd[e]), d1 = bf16hi(wd[e]);
            oa[e] = pack_bf16x2(d0 * gelu_erf_f(g0), d1 * gelu_erf_f(g1));
            og[e] = pack_bf16x2(d0 * a0 * gelu_grad_f(g0), d1 * a1 * gelu_grad_f(g1));
        }
        *reinterpret_cast<u32x4*>(dh + m * 2 * F + c) = (u32x4){oa[0], oa[1], oa[2], oa[3]};
        *reinterpret_cast<u32x4*>(dh + m * 2 * F + F + c) = (u32x4){og[0], og[1], og[2], og[3]};
    }
}

// adjoint of nearest-x2 upsampling (openaimodel.py:108-118): y[b, i, j, :] = sum of the 2x2 block of x[b, 2i.., 2j.., :]
__global__ __launch_bounds__(NT) void sumpool2x2_kernel(const bf16_t* x, bf16_t* y, int B, int H, int W, int C) {
    const int c8 = C / 8;
    const long total = (long)B * H * W * c8;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int cc = (int)(i % c8);
        const long pix = i / c8;
        const int j = (int)(pix % W), ii = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const long src = (((long)b * 2 * H + 2 * ii + dy) * 2 * W + 2 * j + dx) * C + cc * 8;
                const u32x4 v = *reinterpret_cast<const u32x4*>(x + src);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[2 * e] += bf16lo(w[e]); acc[2 * e + 1] += bf16hi(w[e]); }
            }
        *reinterpret_cast<u32x4*>(y + pix * C + cc * 8) =
            (u32x4){pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7])};
    }
}

// out[n] = sum_m x[m][n] (fixed order over rows): bias gradient of a small trainable Linear.  One thread per column.
__global__ __launch_bounds__(NT) void colsum_kernel(const bf16_t* x, float* out, int M, int N, long ld) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += bf16_to_f32(x[(long)m * ld + n]);
    out[n] = s;
}

// d/d(pred) of mean((pred - target)^2) (train.py:696): 2 (pred - target) / n, times an optional loss scale
__global__ __launch_bounds__(NT) void mse_grad_kernel(const float* pred, const float* target, float* out, long n, float coef) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) out[i] = coef * (pred[i] - target[i]);
}

// AdamW (torch.optim.AdamW semantics, train.py:536-541): decoupled weight decay, bias-corrected moments, fp32 state.
__global__ __launch_bounds__(NT) void adamw_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float gscale) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float gi = g[i] * gscale;
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}

// dst[row = code[b]] += src[b] for all b with that code, in batch order (deterministic scatter-add of task-embedding gradients)
__global__ __launch_bounds__(NT) void scatter_rows_kernel(const float* src, const int* code, float* dst, int B, int D) {
    const int row = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += NT) {
        float s = 0.f;
        bool any = false;
        for (int b = 0; b < B; ++b)
            if (code[b] == row) { s += src[(long)b * D + d]; any = true; }
        if (any) dst[(long)row * D + d] += s;
    }
}

// out[r] = sum_j x[r][j], fp32, fixed order (gate gradient = sum over heads and rows of delta)
__global__ __launch_bounds__(NT) void rowsum_kernel(const float* x, float* out, long n) {
    __shared__ float red[NT];
    const float* src = x + (long)blockIdx.x * n;
    float s = 0.f;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        // 16-byte loads, four independent chains per thread (a row of 32 768 values was 128 dependent 4-byte loads per thread: 14 us per launch)
        const f32x4* v = reinterpret_cast<const f32x4*>(src);
        const long nv = n >> 2;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        long i = threadIdx.x;
        for (; i + 3 * NT < nv; i += 4 * NT) {
            const f32x4 v0 = v[i], v1 = v[i + NT], v2 = v[i + 2 * NT], v3 = v[i + 3 * NT];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; i < nv; i += NT) a0 += v[i];
        const f32x4 t = (a0 + a1) + (a2 + a3);
        s = (t[0] + t[1]) + (t[2] + t[3]);
    } else {
        for (long i = threadIdx.x; i < n; i += NT) s += src[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

int nblocks(long work) {
    long nb = (work + NT - 1) / NT;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    return (int)nb;
}

bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int ae_add_bf16(const void* a, const void* b, void* y, long n, void* stream) {
    AE_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "ae_add_bf16: n=%ld must be a positive multiple of 8", n);
    AE_REQUIRE(al16(a) && al16(b) && al16(y), "ae_add_bf16: 16-byte alignment");
    hipLaunchKernelGGL(add_kernel, dim3(nblocks(n / 8)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n / 8);
    return ae_check_launch("ae_add_bf16");
}

extern "C" int ae_axpy_bf16(const void* a, const void* b, float alpha, void* y, long n, void* stream) {
    AE_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "ae_axpy_bf16: n=%ld must be a positive multiple of 8", n);
    AE_REQUIRE(al16(a) && al16(b) && al16(y), "ae_axpy_bf16: 16-byte alignment");
    hipLaunchKernelGGL(axpy_kernel, dim3(nblocks(n / 8)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y,
                       n / 8, alpha);
    return ae_check_launch("ae_axpy_bf16");
}

extern "C" int ae_geglu_fwd_bf16(const void* h, void* y, long M, int F, void* stream) {
    AE_REQUIRE(h && y && M > 0 && F > 0 && F % 8 == 0, "ae_geglu_fwd_bf16: F=%d must be a positive multiple of 8", F);
    AE_REQUIRE(al16(h) && al16(y), "ae_geglu_fwd_bf16: 16-byte alignment");
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(nblocks(M * (F / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)h, (bf16_t*)y, M, F);
    return ae_check_launch("ae_geglu_fwd_bf16");
}

extern "C" int ae_geglu_bwd_bf16(const void* h, const void* dy, void* dh, long M, int F, void* stream) {
    AE_REQUIRE(h && dy && dh && M > 0 && F > 0 && F % 8 == 0, "ae_geglu_bwd_bf16: F=%d must be a positive multiple of 8", F);
    AE_REQUIRE(al16(h) && al16(dy) && al16(dh), "ae_geglu_bwd_bf16: 16-byte alignment");
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(nblocks(M * (F / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)h, (const bf16_t*)dy,
                       (bf16_t*)dh, M, F);
    return ae_check_launch("ae_geglu_bwd_bf16");
}

extern "C" int ae_sumpool2x2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    AE_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "ae_sumpool2x2_bf16: bad shape (C=%d must be a multiple of 8)", C);
    AE_REQUIRE(al16(x) && al16(y), "ae_sumpool2x2_bf16: 16-byte alignment");
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3(nblocks((long)B * H * W * (C / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, B, H, W, C);
    return ae_check_launch("ae_sumpool2x2_bf16");
}

extern "C" int ae_colsum_bf16_f32(const void* x, float* out, int M, int N, long ld, void* stream) {
    AE_REQUIRE(x && out && M > 0 && N > 0 && ld >= N, "ae_colsum_bf16_f32: bad shape M=%d N=%d", M, N);
    hipLaunchKernelGGL(colsum_kernel, dim3((N + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, out, M, N, ld);
    return ae_check_launch("ae_colsum_bf16_f32");
}

extern "C" int ae_mse_grad_f32(const float* pred, const float* target, float* out, long n, float loss_scale, void* stream) {
    AE_REQUIRE(pred && target && out && n > 0, "ae_mse_grad_f32: null pointer / empty");
    hipLaunchKernelGGL(mse_grad_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, pred, target, out, n, 2.0f * loss_scale / (float)n);
    return ae_check_launch("ae_mse_grad_f32");
}

extern "C" int ae_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
    AE_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "ae_adamw_f32: bad arguments (step counts from 1)");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(nblocks(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
    return ae_check_launch("ae_adamw_f32");
}

extern "C" int ae_scatter_add_rows_f32(const float* src, const int* code, float* dst, int B, int D, int n_rows, void* stream) {
    AE_REQUIRE(src && code && dst && B > 0 && D > 0 && n_rows > 0, "ae_scatter_add_rows_f32: bad arguments");
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n_rows), dim3(NT), 0, (hipStream_t)stream, src, code, dst, B, D);
    return ae_check_launch("ae_scatter_add_rows_f32");
}

extern "C" int ae_rowsum_f32(const float* x, float* out, int rows, long n, void* stream) {
    AE_REQUIRE(x && out && rows > 0 && n > 0, "ae_rowsum_f32: bad arguments");
    hipLaunchKernelGGL(rowsum_kernel, dim3(rows), dim3(NT), 0, (hipStream_t)stream, x, out, n);
    return ae_check_launch("ae_rowsum_f32");
}
