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

}  // namespace

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
