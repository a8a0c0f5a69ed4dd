// Multi-scale deformable attention forward for gfx950 — SURVEY.md §8(f) N2: the one native operator of the reference
// (GroundingDINO's `_C.ms_deform_attn_forward`, csrc/MsDeformAttn/ms_deform_attn.h:22-41; CPU build = AT_ERROR).
//
//   out[b, q, h, :] = sum_{l, p} attn_weight[b, q, h, l, p] * bilinear(value_l[b, :, h, :], sampling_loc[b, q, h, l, p])
//
// value [bs, sum_l H_l W_l, heads, d] fp32, sampling_loc in [0, 1] as (x, y), pixel coordinates x*W - 0.5, y*H - 0.5 (grid_sample
// align_corners=False), samples outside the level read zeros (ms_deform_attn.py:93-133 is the readable restatement).
// A gather-bound kernel: no MFMA.  One thread owns 4 consecutive channels of one (batch, query, head): its four bilinear corners are
// 16-byte loads, the d/4 threads of a head read one full 4*d-byte line per corner, and the location / weight loads are wave-level
// broadcasts.  Output stores are 16 bytes, contiguous over (head, channel).
#include "common.hpp"

namespace {

struct MsdaArgs {
    const float* value; const long* shapes; const long* level_start; const float* loc; const float* weight; float* out;
    int bs, S, heads, d, Q, L, P;
};

__global__ __launch_bounds__(256) void msda_fwd_kernel(const MsdaArgs p) {
    const int d4 = p.d / 4;
    const long total = (long)p.bs * p.Q * p.heads * d4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % d4);
        long r = idx / d4;
        const int h = (int)(r % p.heads);
        r /= p.heads;                      // r = b * Q + q
        const int b = (int)(r / p.Q);
        const long samp = (r * p.heads + h) * p.L * p.P;          // first (level, point) of this (b, q, h)
        const long row_stride = (long)p.heads * p.d;              // floats between consecutive value pixels
        const float* vb = p.value + (long)b * p.S * row_stride + (long)h * p.d + c4 * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < p.L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const float* vl = vb + p.level_start[l] * row_stride;
            for (int pt = 0; pt < p.P; ++pt) {
                const long s = samp + (long)l * p.P + pt;
                const float x = p.loc[2 * s] * (float)W - 0.5f, y = p.loc[2 * s + 1] * (float)H - 0.5f;
                const float wgt = p.weight[s];
                if (!(y > -1.f && x > -1.f && y < (float)H && x < (float)W)) continue;   // whole footprint outside: contributes 0
                const int y0 = (int)floorf(y), x0 = (int)floorf(x);
                const float ly = y - (float)y0, lx = x - (float)x0;
                const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (y0 >= 0 && x0 >= 0) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(vl + ((long)y0 * W + x0) * row_stride);
                    v[0] += w00 * t[0]; v[1] += w00 * t[1]; v[2] += w00 * t[2]; v[3] += w00 * t[3];
                }
                if (y0 >= 0 && x0 + 1 <= W - 1) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(vl + ((long)y0 * W + x0 + 1) * row_stride);
                    v[0] += w01 * t[0]; v[1] += w01 * t[1]; v[2] += w01 * t[2]; v[3] += w01 * t[3];
                }
                if (y0 + 1 <= H - 1 && x0 >= 0) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(vl + ((long)(y0 + 1) * W + x0) * row_stride);
                    v[0] += w10 * t[0]; v[1] += w10 * t[1]; v[2] += w10 * t[2]; v[3] += w10 * t[3];
                }
                if (y0 + 1 <= H - 1 && x0 + 1 <= W - 1) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(vl + ((long)(y0 + 1) * W + x0 + 1) * row_stride);
                    v[0] += w11 * t[0]; v[1] += w11 * t[1]; v[2] += w11 * t[2]; v[3] += w11 * t[3];
                }
                acc[0] += wgt * v[0]; acc[1] += wgt * v[1]; acc[2] += wgt * v[2]; acc[3] += wgt * v[3];
            }
        }
        *reinterpret_cast<f32x4*>(p.out + (r * p.heads + h) * p.d + c4 * 4) = acc;
    }
}

// Backward of the sampling core — `_C.ms_deform_attn_backward` (csrc/vision.cpp:57, called from MultiScaleDeformableAttnFunction.backward,
// ms_deform_attn.py:68-90).  Same thread mapping as the forward (4 channels of one (batch, query, head) per thread), so the re-gather of the
// four corners is the forward's access pattern; with g = grad_out[b, q, h, :]:
//   grad_value[corner] += attn_weight * w_corner * g                       (16-byte-wide rows of float atomics: many queries hit one pixel)
//   grad_attn_weight    = sum_c g[c] * bilinear[c]
//   grad_loc.x          = attn_weight * W * sum_c g[c] * ((1-ly) (t01 - t00) + ly (t11 - t10))[c]      (t = 0 outside the level)
//   grad_loc.y          = attn_weight * H * sum_c g[c] * ((1-lx) (t10 - t00) + lx (t11 - t01))[c]
// The channel sums run over the d/4 threads of a head: a xor-shuffle tree when d/4 is a power of two (the head's threads are consecutive
// lanes and take the same branches), float atomics into zeroed outputs otherwise.
struct MsdaBwdArgs {
    const float* value; const long* shapes; const long* level_start; const float* loc; const float* weight; const float* grad_out;
    float* grad_value; float* grad_loc; float* grad_weight;
    int bs, S, heads, d, Q, L, P, shuffle;
};

__device__ __forceinline__ float dot4(const f32x4 a, const f32x4 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }

__device__ __forceinline__ void atomic_add4(float* dst, const f32x4 g, float s) {
    unsafeAtomicAdd(dst + 0, s * g[0]);
    unsafeAtomicAdd(dst + 1, s * g[1]);
    unsafeAtomicAdd(dst + 2, s * g[2]);
    unsafeAtomicAdd(dst + 3, s * g[3]);
}

__global__ __launch_bounds__(256) void msda_bwd_kernel(const MsdaBwdArgs p) {
    const int d4 = p.d / 4;
    const long total = (long)p.bs * p.Q * p.heads * d4;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c4 = (int)(idx % d4);
        long r = idx / d4;
        const int h = (int)(r % p.heads);
        r /= p.heads;                      // r = b * Q + q
        const int b = (int)(r / p.Q);
        const long samp = (r * p.heads + h) * p.L * p.P;
        const long row_stride = (long)p.heads * p.d;
        const long voff = (long)b * p.S * row_stride + (long)h * p.d + c4 * 4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.grad_out + (r * p.heads + h) * p.d + c4 * 4);
        for (int l = 0; l < p.L; ++l) {
            const int H = (int)p.shapes[2 * l], W = (int)p.shapes[2 * l + 1];
            const long loff = voff + p.level_start[l] * row_stride;
            for (int pt = 0; pt < p.P; ++pt) {
                const long s = samp + (long)l * p.P + pt;
                const float x = p.loc[2 * s] * (float)W - 0.5f, y = p.loc[2 * s + 1] * (float)H - 0.5f;
                const float wgt = p.weight[s];
                float gw = 0.f, gx = 0.f, gy = 0.f;
                if (y > -1.f && x > -1.f && y < (float)H && x < (float)W) {
                    const int y0 = (int)floorf(y), x0 = (int)floorf(x);
                    const float ly = y - (float)y0, lx = x - (float)x0;
                    const bool top = y0 >= 0, bot = y0 + 1 <= H - 1, lef = x0 >= 0, rig = x0 + 1 <= W - 1;
                    const long o00 = loff + ((long)y0 * W + x0) * row_stride, o01 = o00 + row_stride, o10 = o00 + (long)W * row_stride,
                               o11 = o10 + row_stride;
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 t00 = top && lef ? *reinterpret_cast<const f32x4*>(p.value + o00) : z;
                    const f32x4 t01 = top && rig ? *reinterpret_cast<const f32x4*>(p.value + o01) : z;
                    const f32x4 t10 = bot && lef ? *reinterpret_cast<const f32x4*>(p.value + o10) : z;
                    const f32x4 t11 = bot && rig ? *reinterpret_cast<const f32x4*>(p.value + o11) : z;
                    const float d00 = dot4(g, t00), d01 = dot4(g, t01), d10 = dot4(g, t10), d11 = dot4(g, t11);
                    gw = (1.f - ly) * ((1.f - lx) * d00 + lx * d01) + ly * ((1.f - lx) * d10 + lx * d11);
                    gx = wgt * (float)W * ((1.f - ly) * (d01 - d00) + ly * (d11 - d10));
                    gy = wgt * (float)H * ((1.f - lx) * (d10 - d00) + lx * (d11 - d01));
                    if (top && lef) atomic_add4(p.grad_value + o00, g, wgt * (1.f - ly) * (1.f - lx));
                    if (top && rig) atomic_add4(p.grad_value + o01, g, wgt * (1.f - ly) * lx);
                    if (bot && lef) atomic_add4(p.grad_value + o10, g, wgt * ly * (1.f - lx));
                    if (bot && rig) atomic_add4(p.grad_value + o11, g, wgt * ly * lx);
                }
                if (p.shuffle) {
                    for (int m = d4 >> 1; m > 0; m >>= 1) {
                        gw += __shfl_xor(gw, m);
                        gx += __shfl_xor(gx, m);
                        gy += __shfl_xor(gy, m);
                    }
                    if (c4 == 0) {
                        p.grad_weight[s] = gw;
                        p.grad_loc[2 * s] = gx;
                        p.grad_loc[2 * s + 1] = gy;
                    }
                } else {
                    unsafeAtomicAdd(p.grad_weight + s, gw);
                    unsafeAtomicAdd(p.grad_loc + 2 * s, gx);
                    unsafeAtomicAdd(p.grad_loc + 2 * s + 1, gy);
                }
            }
        }
    }
}

// ---- fp32 Linear for the four small projections of MultiScaleDeformableAttention (ms_deform_attn.py:281-288, 330-352: value_proj,
// sampling_offsets, attention_weights, output_proj).  GroundingDINO runs in fp32 and its sampling offsets feed bilinear gathers, so
// these stay EXACT fp32: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate = an fmaf chain; 1/16 of the bf16 MFMA rate, far above what
// the 13 k-token x 256-channel problems need).  C[M,N] = A[M,K] W[N,K]^T + bias.  One wave owns a 32 x 32 output tile; the K order
// inside a 16-wide step is permuted identically on both operands (lane group g carries k = 4 g .. 4 g + 3 of the step as ONE
// 16-byte load and feeds component s to MFMA s), so no LDS staging is needed.  D[i][j]: lane holds i = 4 g + r, j = l15.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__global__ __launch_bounds__(256) void linear_f32_kernel(const float* A, long lda, const float* W, long ldw, const float* bias, float* C,
                                                         long ldc, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int ntn = (N + 63) / 64;
    const int m0 = (blockIdx.x / ntn) * 64 + (wave >> 1) * 32, n0 = (blockIdx.x % ntn) * 64 + (wave & 1) * 32;
    const float* a0 = A + (long)min(m0 + l15, M - 1) * lda + 4 * g;        // clamped rows / columns: their products are never stored
    const float* a1 = A + (long)min(m0 + 16 + l15, M - 1) * lda + 4 * g;
    const float* w0 = W + (long)min(n0 + l15, N - 1) * ldw + 4 * g;
    const float* w1 = W + (long)min(n0 + 16 + l15, N - 1) * ldw + 4 * g;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        const f32x4_t av[2] = {*reinterpret_cast<const f32x4_t*>(a0 + k0), *reinterpret_cast<const f32x4_t*>(a1 + k0)};
        const f32x4_t wv[2] = {*reinterpret_cast<const f32x4_t*>(w0 + k0), *reinterpret_cast<const f32x4_t*>(w1 + k0)};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][s4], wv[j][s4], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 16 * j + l15;
            if (n >= N) continue;
            const float bz = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * i + 4 * g + r;
                if (m < M) C[(long)m * ldc + n] = acc[i][j][r] + bz;
            }
        }
}

}  // namespace

// nn.Linear in exact fp32 (A [M,K], W [N,K], bias [N] or NULL -> C [M,N]); K % 16 == 0, rows of A and W 16-byte aligned.
extern "C" int ae_linear_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, int M, int N, int K,
                             void* stream) {
    AE_REQUIRE(A && W && C, "ae_linear_f32: null pointer");
    AE_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, "ae_linear_f32: M=%d N=%d must be positive, K=%d a positive multiple of 16", M, N, K);
    AE_REQUIRE(lda % 4 == 0 && ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0,
               "ae_linear_f32: rows of A and W must be 16-byte aligned");
    const long blocks = (long)((M + 63) / 64) * ((N + 63) / 64);
    AE_REQUIRE(blocks < (1L << 31), "ae_linear_f32: grid too large");
    hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, A, lda, W, ldw, bias, C, ldc, M, N, K);
    return ae_check_launch("ae_linear_f32");
}

// Drop-in for `_C.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)`
// (csrc/vision.cpp:53-56): same tensors as raw device pointers (spatial_shapes / level_start_index are int64 as the reference
// passes them); im2col_step only chunked the CUDA launch over the batch and has no counterpart here.  out: [bs, Q, heads*d] fp32.
extern "C" int ae_ms_deform_attn_fwd_f32(const float* value, const long* spatial_shapes, const long* level_start_index,
                                         const float* sampling_loc, const float* attn_weight, float* out, int bs, int S, int heads, int d,
                                         int Q, int L, int P, void* stream) {
    AE_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && out, "ae_ms_deform_attn_fwd_f32: null pointer");
    AE_REQUIRE(bs > 0 && S > 0 && heads > 0 && d > 0 && Q > 0 && L > 0 && P > 0, "ae_ms_deform_attn_fwd_f32: bad sizes");
    AE_REQUIRE(d % 4 == 0, "ae_ms_deform_attn_fwd_f32: channels per head d=%d must be a multiple of 4", d);
    AE_REQUIRE((reinterpret_cast<uintptr_t>(value) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
               "ae_ms_deform_attn_fwd_f32: value / out must be 16-byte aligned");
    MsdaArgs a{value, spatial_shapes, level_start_index, sampling_loc, attn_weight, out, bs, S, heads, d, Q, L, P};
    const long total = (long)bs * Q * heads * (d / 4);
    long nb = (total + 255) / 256;
    if (nb > 65535 * 4) nb = 65535 * 4;
    hipLaunchKernelGGL(msda_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a);
    return ae_check_launch("ae_ms_deform_attn_fwd_f32");
}

// Drop-in for `_C.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step)`
// (csrc/vision.cpp:57; ms_deform_attn.py:68-90): grad_output [bs, Q, heads*d] -> grad_value [bs, S, heads, d], grad_sampling_loc
// [bs, Q, heads, L, P, 2], grad_attn_weight [bs, Q, heads, L, P], all fp32 and fully overwritten (zeroed on `stream` first where they
// are accumulated).  grad_value is a sum of float atomics: equal to the reference up to fp32 summation order.
extern "C" int ae_ms_deform_attn_bwd_f32(const float* value, const long* spatial_shapes, const long* level_start_index,
                                         const float* sampling_loc, const float* attn_weight, const float* grad_out, float* grad_value,
                                         float* grad_sampling_loc, float* grad_attn_weight, int bs, int S, int heads, int d, int Q, int L, int P,
                                         void* stream) {
    AE_REQUIRE(value && spatial_shapes && level_start_index && sampling_loc && attn_weight && grad_out && grad_value && grad_sampling_loc &&
                   grad_attn_weight, "ae_ms_deform_attn_bwd_f32: null pointer");
    AE_REQUIRE(bs > 0 && S > 0 && heads > 0 && d > 0 && Q > 0 && L > 0 && P > 0, "ae_ms_deform_attn_bwd_f32: bad sizes");
    AE_REQUIRE(d % 4 == 0, "ae_ms_deform_attn_bwd_f32: channels per head d=%d must be a multiple of 4", d);
    AE_REQUIRE((reinterpret_cast<uintptr_t>(value) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_out) & 15) == 0,
               "ae_ms_deform_attn_bwd_f32: value / grad_out must be 16-byte aligned");
    const int d4 = d / 4;
    const int shuffle = (d4 & (d4 - 1)) == 0 && d4 <= 64;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)bs * S * heads * d, st);
    if (e == hipSuccess && !shuffle) {
        const size_t ns = (size_t)bs * Q * heads * L * P;
        e = hipMemsetAsync(grad_attn_weight, 0, sizeof(float) * ns, st);
        if (e == hipSuccess) e = hipMemsetAsync(grad_sampling_loc, 0, sizeof(float) * 2 * ns, st);
    }
    AE_REQUIRE(e == hipSuccess, "ae_ms_deform_attn_bwd_f32: hipMemsetAsync failed: %s", hipGetErrorString(e));
    MsdaBwdArgs a{value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_out, grad_value, grad_sampling_loc,
                  grad_attn_weight, bs, S, heads, d, Q, L, P, shuffle};
    const long total = (long)bs * Q * heads * d4;
    long nb = (total + 255) / 256;
    if (nb > 65535 * 4) nb = 65535 * 4;
    hipLaunchKernelGGL(msda_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, st, a);
    return ae_check_launch("ae_ms_deform_attn_bwd_f32");
}
