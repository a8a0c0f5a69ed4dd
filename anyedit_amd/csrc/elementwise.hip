// Elementwise / data-movement kernels of the denoising loop for gfx950 — all HBM- or latency-bound.
//
// Replaces:
//   A7  timestep_embedding (ldm/modules/diffusionmodules/util.py:154-174, [cos, sin] order), layout changes at the
//       UNet boundary (NCHW fp32 <-> channels-last bf16), th.cat skip-concat (openaimodel.py:780);
//   A8  p_sample_ddim arithmetic (ldm/models/diffusion/ddim.py:211-212, 228-250) fused with the CFG combine
//       (2-branch ddim.py:211-212 or 3-branch InstructPix2Pix global_tool.py:172-177), q_sample (ddpm.py:356-359)
//       and the masked-latent blend (ddim.py:154-157 / global_tool.py:183-184);
//   A10 SAM window_partition / window_unpartition (image_encoder.py:243-289), decomposed rel-pos terms (:325-355),
//       PatchEmbed im2col (:364-395);
//   A11 eps-MSE reduction (train.py:696).
// The DDIM update is compiled with fp contraction OFF so that, given the same eps, it is bit-identical to the
// reference's unfused fp32 torch expression sequence.
#include "common.hpp"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float ld_any(const void* p, long i, int is_bf16) {
    return is_bf16 ? bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_any(void* p, long i, float v, int is_bf16) {
    if (is_bf16) reinterpret_cast<bf16_t*>(p)[i] = f32_to_bf16(v);
    else reinterpret_cast<float*>(p)[i] = v;
}

// out[b, y, x] = in[b, x, y] for x < X, zero for X <= x < Xpad.  32x32 LDS tile, coalesced both ways.
__global__ void transpose_kernel(const void* in, void* out, int X, int Y, int Xpad, int in_bf16, int out_bf16) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int x = x0 + j, y = y0 + tx;
        tile[j][tx] = (x < X && y < Y) ? ld_any(in, ((long)b * X + x) * Y + y, in_bf16) : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int y = y0 + j, x = x0 + tx;
        if (y < Y && x < Xpad) st_any(out, ((long)b * Y + y) * Xpad + x, tile[tx][j], out_bf16);
    }
}

__global__ void concat_kernel(const bf16_t* a, int Ca, const bf16_t* b, int Cb, bf16_t* y, long rows) {
    const int C = Ca + Cb, ncc = C / 8;
    const long total = rows * ncc;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / ncc;
        const int ch = (int)(i % ncc) * 8;
        const u32x4 v = ch < Ca ? *reinterpret_cast<const u32x4*>(a + r * Ca + ch) : *reinterpret_cast<const u32x4*>(b + r * Cb + (ch - Ca));
        *reinterpret_cast<u32x4*>(y + r * C + ch) = v;
    }
}

// im2col of a 3x3 / pad 1 / stride 1 convolution over an 8-channel channels-last map: row m of y holds the nine taps' 8 channels (k = 8 tap + c,
// tap = 3 ky + kx; 16 bytes each, zeros outside the image) and seven zero chunks up to 128 columns — the UNet's stem conv (8 -> 320, openaimodel.py:536-542)
// then runs as a dense K = 128 GEMM instead of an implicit GEMM whose nine K tiles are 7/8 zero padding (Cin = 8 padded to the 64-deep K tile).
__global__ void im2col3x3_c8_kernel(const bf16_t* x, bf16_t* y, int B, int H, int W) {
    const long total = (long)B * H * W * 16;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long m = i >> 4;
        const int t = (int)(i & 15);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (t < 9) {
            const int hw = H * W;
            const int b = (int)(m / hw), rem = (int)(m - (long)b * hw);
            const int oy = rem / W, ox = rem - oy * W;
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = *reinterpret_cast<const u32x4*>(x + (((long)b * H + iy) * W + ix) * 8);
        }
        *reinterpret_cast<u32x4*>(y + m * 128 + t * 8) = v;
    }
}

// adjoint of concat_kernel: the two channel slices of dy go to da / db in one pass (written, or added to what is there)
__global__ void split_kernel(const bf16_t* y, int Ca, int Cb, bf16_t* a, bf16_t* b, long rows, int acc_a, int acc_b) {
    const int C = Ca + Cb, ncc = C / 8;
    const long total = rows * ncc;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const long r = i / ncc;
        const int ch = (int)(i % ncc) * 8;
        const bool first = ch < Ca;
        bf16_t* dst = first ? a + r * Ca + ch : b + r * Cb + (ch - Ca);
        if ((first ? a : b) == nullptr) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(y + r * C + ch);
        if (first ? acc_a : acc_b) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(dst);
            v.x = pack_bf16x2(bf16lo(v.x) + bf16lo(o.x), bf16hi(v.x) + bf16hi(o.x));
            v.y = pack_bf16x2(bf16lo(v.y) + bf16lo(o.y), bf16hi(v.y) + bf16hi(o.y));
            v.z = pack_bf16x2(bf16lo(v.z) + bf16lo(o.z), bf16hi(v.z) + bf16hi(o.z));
            v.w = pack_bf16x2(bf16lo(v.w) + bf16lo(o.w), bf16hi(v.w) + bf16hi(o.w));
        }
        *reinterpret_cast<u32x4*>(dst) = v;
    }
}

__global__ void timestep_embedding_kernel(const long* t_i64, const float* t_f32, bf16_t* out_bf16, float* out_f32, int B,
                                          int dim, float max_period) {
    const int half = dim / 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= B * dim) return;
    const int b = i / dim, j = i % dim;
    const float t = t_i64 ? (float)t_i64[b] : t_f32[b];
    float v = 0.f;
    if (j < 2 * half) {
        const int k = j < half ? j : j - half;
        const float freq = expf(-logf(max_period) * (float)k / (float)half);
        const float arg = t * freq;
        v = j < half ? cosf(arg) : sinf(arg);
    }
    if (out_bf16) out_bf16[i] = f32_to_bf16(v);
    if (out_f32) out_f32[i] = v;
}

struct DdimArgs {
    const float* x; const float* eps; const float* noise; float* x_prev; float* pred_x0; float* e_out;
    long n;         // elements per branch = B*C*H*W
    int branches;   // 1, 2 (uncond, cond) or 3 (text, image, uncond)
    float s0, s1;   // 2-branch: s0 = scale ; 3-branch: s0 = s_txt, s1 = s_img
    float sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma_t, temperature;
};

#pragma clang fp contract(off)
__global__ void ddim_step_kernel(const DdimArgs p) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < p.n; i += (long)gridDim.x * NT) {
        float e;
        if (p.branches == 1) {
            e = p.eps[i];
        } else if (p.branches == 2) {  // ddim.py:211-212: e_uncond + s * (e_cond - e_uncond); batch order [uncond, cond]
            const float eu = p.eps[i], ec = p.eps[p.n + i];
            e = eu + p.s0 * (ec - eu);
        } else {  // global_tool.py:172-177; batch order [text, image, uncond]
            const float et = p.eps[i], ei = p.eps[p.n + i], eu = p.eps[2 * p.n + i];
            e = eu + p.s0 * (et - ei) + p.s1 * (ei - eu);
        }
        const float x = p.x[i];
        const float px0 = (x - p.sqrt_one_minus_at * e) / p.sqrt_at;          // ddim.py:235
        const float dir = p.dir_coef * e;                                     // ddim.py:246
        const float nz = p.sigma_t * (p.noise ? p.noise[i] : 0.f) * p.temperature;  // ddim.py:247
        p.x_prev[i] = p.sqrt_a_prev * px0 + dir + nz;                         // ddim.py:250
        if (p.pred_x0) p.pred_x0[i] = px0;
        if (p.e_out) p.e_out[i] = e;
    }
}

// DDIM inversion step (DDIMSampler.encode, ddim.py:253-298): x_next = cx * x + ce * e with e the CFG combination
// e_uncond + s (e_cond - e_uncond) (ddim.py:277-280; batch order [uncond, cond]); two products and one sum, un-fused, like the
// reference's xt_weighted + weighted_noise_pred.  cx, ce are computed on the host in float64 exactly as the reference does.
__global__ void ddim_encode_step_kernel(const float* x, const float* eps, float* x_next, long n, int branches, float scale, float cx,
                                        float ce) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        float e = eps[i];
        if (branches == 2) {
            const float eu = eps[i], ec = eps[n + i];
            e = eu + scale * (ec - eu);
        }
        const float xw = cx * x[i];
        const float ew = ce * e;
        x_next[i] = xw + ew;
    }
}

// PLMS pseudo linear multistep combination of the current and up to three previous eps predictions (plms.py:226-240), in the
// reference's left-to-right fp32 expression order (products rounded, then summed, then one division).
//   order 0: (e + o1) / 2  [pseudo improved Euler, o1 = eps at the next timestep]      order 1: (3 e - o1) / 2
//   order 2: (23 e - 16 o1 + 5 o2) / 12                                                  order 3: (55 e - 59 o1 + 37 o2 - 9 o3) / 24
__global__ void plms_combine_kernel(const float* e, const float* o1, const float* o2, const float* o3, float* out, long n, int order) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        float r;
        if (order == 0) {
            r = (e[i] + o1[i]) / 2.0f;
        } else if (order == 1) {
            const float a = 3.0f * e[i];
            r = (a - o1[i]) / 2.0f;
        } else if (order == 2) {
            const float a = 23.0f * e[i], b = 16.0f * o1[i], c = 5.0f * o2[i];
            r = ((a - b) + c) / 12.0f;
        } else {
            const float a = 55.0f * e[i], b = 59.0f * o1[i], c = 37.0f * o2[i], d = 9.0f * o3[i];
            r = (((a - b) + c) - d) / 24.0f;
        }
        out[i] = r;
    }
}

// out = (sa*x0 + s1*noise) * mask + (1 - mask) * img   (ddim.py:154-157 with q_sample ddpm.py:356-359);
// mask is [B,1,H,W] broadcast over C.  order!=0 -> IP2P order: img*mask + q*(1-mask) (global_tool.py:183-184)
__global__ void mask_blend_kernel(const float* img, const float* x0, const float* noise, const float* mask, float* out,
                                  long n, int C, int HW, float sa, float s1, int ip2p_order) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const long b = i / ((long)C * HW);
        const int hw = (int)(i % HW);
        const float m = mask[b * HW + hw];
        const float q = sa * x0[i] + s1 * noise[i];
        out[i] = ip2p_order ? (img[i] * m + q * (1.f - m)) : (q * m + (1.f - m) * img[i]);
    }
}

__global__ void q_sample_kernel(const float* x0, const float* noise, const float* sa, const float* s1, float* out, long per_sample,
                                long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const long b = i / per_sample;
        out[i] = sa[b] * x0[i] + s1[b] * noise[i];
    }
}
// DPM-Solver / DPM-Solver++ multistep step (dpm_solver.py:246-316, 352-365, 469-513, 723-777), one launch per network evaluation:
//   e   = guided noise of the network output(s) (batch order [uncond, cond]; model_type 'v': alpha_s * out + sigma_s * x),
//   m   = predict_x0 ? (x - sigma_s e) / alpha_s : e                              (the "model value" kept in the history),
//   x'  = a x - b m - c D,  D = inv_r0 (m - m_prev)  (first order: m_prev == nullptr, D term dropped).
// a, b, c, inv_r0 are the per-step scalars of the reference's update formulas, computed on the host in fp32 exactly as it does.
// Un-fused arithmetic (contraction off) so that identical network outputs give the reference's fp32 results.
struct DpmArgs {
    const float* x; const float* out; const float* m_prev; float* m_cur; float* x_next;
    long n; int branches; int v_param; int predict_x0; int update;
    float scale, sigma_s, alpha_s, a, b, c, inv_r0;
};

__global__ void dpm_multistep_kernel(const DpmArgs p) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < p.n; i += (long)gridDim.x * NT) {
        const float x = p.x[i];
        float e = p.out[i];
        if (p.v_param) e = p.alpha_s * e + p.sigma_s * x;
        if (p.branches == 2) {
            float ec = p.out[p.n + i];
            if (p.v_param) ec = p.alpha_s * ec + p.sigma_s * x;
            e = e + p.scale * (ec - e);
        }
        const float m = p.predict_x0 ? (x - p.sigma_s * e) / p.alpha_s : e;
        if (p.m_cur) p.m_cur[i] = m;
        if (p.update) {
            float xn = p.a * x - p.b * m;
            if (p.m_prev) {
                const float D = p.inv_r0 * (m - p.m_prev[i]);
                xn = xn - p.c * D;
            }
            p.x_next[i] = xn;
        }
    }
}

#pragma clang fp contract(fast)

__global__ void add_bcast_kernel(const bf16_t* x, const bf16_t* p, bf16_t* y, long n, long period) {
    for (long i = ((long)blockIdx.x * NT + threadIdx.x) * 8; i < n; i += (long)gridDim.x * NT * 8) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(x + i);
        const u32x4 b = *reinterpret_cast<const u32x4*>(p + (i % period));
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(bf16lo(aw[e]) + bf16lo(bw[e]), bf16hi(aw[e]) + bf16hi(bw[e]));
        *reinterpret_cast<u32x4*>(y + i) = (u32x4){o[0], o[1], o[2], o[3]};
    }
}

__global__ void silu_kernel(const void* x, int in_bf16, bf16_t* y, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) y[i] = f32_to_bf16(silu_f(ld_any(x, i, in_bf16)));
}

// Resampling WITHOUT a convolution on channels-last rows (openaimodel.py:108-118 with use_conv=False; :154-155 avg_pool_nd — what ResBlock(up= / down=) puts
// between its GroupNorm + SiLU and its first conv, :215-221, 254-260).  mode 0: nearest x2, out [B, 2H, 2W, C]; mode 1: 2x2 mean, out [B, H/2, W/2, C]
// (floor, as AvgPool2d(2, 2) drops an odd last row / column), summed in fp32 in (ky, kx) order and rounded once.  One thread per 8 output channels.
__global__ void resample2x_rows_kernel(const bf16_t* x, bf16_t* y, int B, int H, int W, int C, int mode) {
    const int ncc = C / 8, Ho = mode ? H / 2 : 2 * H, Wo = mode ? W / 2 : 2 * W;
    const long n = (long)B * Ho * Wo * ncc;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int cc = (int)(i % ncc);
        long r = i / ncc;
        const int xo = (int)(r % Wo);
        r /= Wo;
        const int yo = (int)(r % Ho), b = (int)(r / Ho);
        u32x4 o;
        if (mode == 0) {
            o = *reinterpret_cast<const u32x4*>(x + (((long)b * H + (yo >> 1)) * W + (xo >> 1)) * C + cc * 8);
        } else {
            const bf16_t* s = x + (((long)b * H + 2 * yo) * W + 2 * xo) * C + cc * 8;
            const u32x4 p00 = *reinterpret_cast<const u32x4*>(s), p01 = *reinterpret_cast<const u32x4*>(s + C);
            const u32x4 p10 = *reinterpret_cast<const u32x4*>(s + (long)W * C), p11 = *reinterpret_cast<const u32x4*>(s + (long)W * C + C);
            const uint32_t a[4] = {p00.x, p00.y, p00.z, p00.w}, bb[4] = {p01.x, p01.y, p01.z, p01.w}, c[4] = {p10.x, p10.y, p10.z, p10.w}, d[4] = {p11.x, p11.y, p11.z, p11.w};
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = pack_bf16x2((((bf16lo(a[e]) + bf16lo(bb[e])) + bf16lo(c[e])) + bf16lo(d[e])) * 0.25f, (((bf16hi(a[e]) + bf16hi(bb[e])) + bf16hi(c[e])) + bf16hi(d[e])) * 0.25f);
            o = (u32x4){w[0], w[1], w[2], w[3]};
        }
        *reinterpret_cast<u32x4*>(y + i * 8) = o;
    }
}

// ResBlock(use_scale_shift_norm=True), openaimodel.py:264-268: h = out_norm(h) * (1 + scale) + shift, then SiLU (out_rest[0]); scale | shift are the two
// halves of the block's emb_layers output, fp32 [B, 2C] with row stride ld (a column slice of the batched time-embedding projection).  x: the GroupNorm's
// output rows [B * HW, C] bf16.
__global__ void scale_shift_rows_kernel(const bf16_t* x, const float* emb, long ld, bf16_t* y, long HW, int C, long n, int silu) {
    const int ncc = C / 8;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const int cc = (int)(i % ncc);
        const long row = i / ncc;
        const float* e = emb + (row / HW) * ld + cc * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
        const uint32_t vw[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lo = bf16lo(vw[k]) * (1.0f + e[2 * k]) + e[C + 2 * k], hi = bf16hi(vw[k]) * (1.0f + e[2 * k + 1]) + e[C + 2 * k + 1];
            if (silu) { lo = silu_f(lo); hi = silu_f(hi); }
            o[k] = pack_bf16x2(lo, hi);
        }
        *reinterpret_cast<u32x4*>(y + i * 8) = (u32x4){o[0], o[1], o[2], o[3]};
    }
}

// x [B,H,W,C] -> windows [B*nH*nW, ws, ws, C], zero padded bottom/right (image_encoder.py:243-264); reverse drops padding.
__global__ void window_kernel(const bf16_t* src, bf16_t* dst, int B, int H, int W, int C, int ws, int nH, int nW, int reverse) {
    const int ncc = C / 8;
    const long total = (long)B * nH * nW * ws * ws * ncc;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int cc = (int)(i % ncc);
        long r = i / ncc;
        const int wx = (int)(r % ws); r /= ws;
        const int wy = (int)(r % ws); r /= ws;
        const int jw = (int)(r % nW); r /= nW;
        const int jh = (int)(r % nH); r /= nH;
        const int b = (int)r;
        const int y = jh * ws + wy, x = jw * ws + wx;
        const bool in = y < H && x < W;
        const long img_off = (((long)b * H + y) * W + x) * C + cc * 8;
        const long win_off = (i / ncc) * C + cc * 8;
        if (!reverse) {
            *reinterpret_cast<u32x4*>(dst + win_off) = in ? *reinterpret_cast<const u32x4*>(src + img_off) : zero4;
        } else if (in) {
            *reinterpret_cast<u32x4*>(dst + img_off) = *reinterpret_cast<const u32x4*>(src + win_off);
        }
    }
}

// rel_h[bh, y*W+x, kh] = sum_c q[b, y*W+x, h, c] * Rh[y, kh, c] ; rel_w[bh, y*W+x, kw] = sum_c q[...] * Rw[x, kw, c]
// (image_encoder.py:349-355).  q addressed by (batch, head, row) strides; Rh/Rw fp32 [qH, kH, D] / [qW, kW, D].
__global__ void relpos_kernel(const bf16_t* q, long q_sb, long q_sh, long q_sn, const float* Rh, const float* Rw, float* rel_h,
                              float* rel_w, int B, int Hh, int qH, int qW, int kH, int kW, int D) {
    extern __shared__ float qs[];  // [D]
    const int bh = blockIdx.y, n = blockIdx.x;
    const int b = bh / Hh, h = bh % Hh;
    const int y = n / qW, x = n % qW;
    const bf16_t* qr = q + (long)b * q_sb + (long)h * q_sh + (long)n * q_sn;
    for (int c = threadIdx.x; c < D; c += blockDim.x) qs[c] = bf16_to_f32(qr[c]);
    __syncthreads();
    const long N = (long)qH * qW;
    for (int k = threadIdx.x; k < kH + kW; k += blockDim.x) {
        const float* R = k < kH ? Rh + ((long)y * kH + k) * D : Rw + ((long)x * kW + (k - kH)) * D;
        float acc = 0.f;
        for (int c = 0; c < D; ++c) acc += qs[c] * R[c];
        if (k < kH) rel_h[((long)bh * N + n) * kH + k] = acc;
        else rel_w[((long)bh * N + n) * kW + (k - kH)] = acc;
    }
}

// Same contraction, one block per (batch*head, query row y) [PART 0: rel_h] or (batch*head, query column x) [PART 1: rel_w]: the
// qW (qH) queries of that line share one [kH, D] (resp. [kW, D]) slice of the table, so both operands are staged in LDS once by
// coalesced loads and every thread produces several outputs (the per-query kernel above launched 65k 128-thread blocks with
// strided table reads: 112 us per SAM block vs ~15 us of traffic).  Limits: line length, kH/kW <= 64, D % 8 == 0, D <= 160.
template <int PART>
__global__ __launch_bounds__(256) void relpos_line_kernel(const bf16_t* q, long q_sb, long q_sh, long q_sn, const float* Rh, const float* Rw,
                                                          float* rel_h, float* rel_w, int Hh, int qH, int qW, int kH, int kW, int D) {
    extern __shared__ float smem_rp[];
    const int LD = D + 4;
    const int nline = PART == 0 ? qH : qW;          // lines per (b, h)
    const int Lq = PART == 0 ? qW : qH;             // queries on this line
    const int K = PART == 0 ? kH : kW;
    float* sQ = smem_rp;                            // [Lq][LD]
    float* sR = smem_rp + 64 * LD;                  // [K][LD]
    const int bh = blockIdx.x / nline, line = blockIdx.x % nline;
    const int b = bh / Hh, h = bh % Hh;
    const bf16_t* qb = q + (long)b * q_sb + (long)h * q_sh;
    const float* R = PART == 0 ? Rh + (long)line * kH * D : Rw + (long)line * kW * D;
    const int d8 = D / 8;
    for (int i = threadIdx.x; i < Lq * d8; i += 256) {
        const int qi = i / d8, c = (i - qi * d8) * 8;
        const int n = PART == 0 ? line * qW + qi : qi * qW + line;
        const u32x4 v = *reinterpret_cast<const u32x4*>(qb + (long)n * q_sn + c);
        float* d = sQ + qi * LD + c;
        d[0] = bf16lo(v.x); d[1] = bf16hi(v.x); d[2] = bf16lo(v.y); d[3] = bf16hi(v.y);
        d[4] = bf16lo(v.z); d[5] = bf16hi(v.z); d[6] = bf16lo(v.w); d[7] = bf16hi(v.w);
    }
    for (int i = threadIdx.x; i < K * D / 4; i += 256) {
        const int k = (i * 4) / D, c = i * 4 - k * D;
        *reinterpret_cast<f32x4*>(sR + k * LD + c) = *reinterpret_cast<const f32x4*>(R + (long)i * 4);
    }
    __syncthreads();
    const long N = (long)qH * qW;
    float* out = PART == 0 ? rel_h : rel_w;
    for (int o = threadIdx.x; o < Lq * K; o += 256) {
        const int qi = o / K, k = o - qi * K;
        const float* a = sQ + qi * LD;
        const float* r = sR + k * LD;
        float acc = 0.f;
        for (int c = 0; c < D; c += 4) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + c), rv = *reinterpret_cast<const f32x4*>(r + c);
            acc += av[0] * rv[0];
            acc += av[1] * rv[1];
            acc += av[2] * rv[2];
            acc += av[3] * rv[3];
        }
        const int n = PART == 0 ? line * qW + qi : qi * qW + line;
        out[((long)bh * N + n) * K + k] = acc;
    }
}

// Round 6: the same contraction on the matrix pipe in fp32 (v_mfma_f32_16x16x4_f32: fp32 products, fp32 accumulation — the terms keep their
// fp32 precision).  Per query line (PART 0: image row y, PART 1: column x) the terms are a small GEMM: the line's (batch, head, query)
// triples [BH * Lq, D] times the line's table slice [K, D]^T.  A wave takes one line, up to 16 table rows (the A operand: 16 rows x 4
// channels per step) and 64 triples (four B-operand tiles of 16), D / 4 steps per tile.  The contraction order is free, so step s of lane
// group g covers channel 16 (s / 4) + 4 g + s % 4: a lane's A values are 16-byte loads of the table and its B values 8-byte loads of the bf16
// query row.  Triples are ordered head-fastest, so with the heads of a token contiguous (a fused qkv row) a tile's loads fall on whole lines.
// The line kernel above staged both operands in LDS and spent two 16-byte LDS reads per four multiply-adds on 196 of 256 threads: 24 us per
// launch, two launches per SAM block = 1.55 ms of an 11.5 ms ViT-H encoder pass (profiles/r06_v20_sam_kernel_stats.csv) for 176 M
// multiply-adds and 21 MB of traffic; a form with the table through the scalar unit measured 21 us (70 dependent scalar-load round trips per
// wave: profiles/r06_v21_sam_relpos.txt).
struct RelposArgs {
    long q_sb, q_sh, q_sn;
    int BH, Hh, qH, qW, kH, kW;
    int items[2];   // wave items of PART 0 / PART 1
};

template <int D16>
__global__ __launch_bounds__(256) void relpos_mfma_kernel(const bf16_t* __restrict__ q, const float* __restrict__ Rh, const float* __restrict__ Rw,
                                                          float* __restrict__ rel_h, float* __restrict__ rel_w, const RelposArgs p) {
    constexpr int D = 16 * D16;
    const int lane = threadIdx.x & 63, l15 = lane & 15, g = lane >> 4;
    long item = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int part = 0;
    if (item >= p.items[0]) { item -= p.items[0]; part = 1; }
    if (item >= p.items[part]) return;
    const int Lq = part == 0 ? p.qW : p.qH, K = part == 0 ? p.kH : p.kW;
    const int nks = (K + 15) / 16;
    const long ntrip = (long)p.BH * Lq;                 // triples on a line
    const int ngrp = (int)((ntrip + 63) / 64);
    // item -> (line, k chunk, group of 64 triples): consecutive waves share the line and the k chunk = the same table rows
    int r = (int)item;
    const int grp = r % ngrp; r /= ngrp;
    const int ks = r % nks; r /= nks;
    const int line = r;
    const int k0 = ks * 16;
    // A operand: table row k0 + l15, channels 16 u + 4 g .. + 3 (rows past K read row K - 1 and are never stored)
    const int krow = k0 + l15 < K ? k0 + l15 : K - 1;
    const float* Rrow = (part == 0 ? Rh : Rw) + ((long)line * K + krow) * D + 4 * g;
    f32x4 av[D16];
#pragma unroll
    for (int u = 0; u < D16; ++u) av[u] = *reinterpret_cast<const f32x4*>(Rrow + 16 * u);
    float* outp = part == 0 ? rel_h : rel_w;
    const long N = (long)p.qH * p.qW;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        long trip = (long)grp * 64 + t * 16 + l15;      // head fastest, then the query on the line, then the batch
        const bool live = trip < ntrip;
        if (!live) trip = ntrip - 1;
        const int h = (int)(trip % p.Hh);
        const long rest = trip / p.Hh;
        const int qi = (int)(rest % Lq), b = (int)(rest / Lq);
        const long n = part == 0 ? (long)line * p.qW + qi : (long)qi * p.qW + line;
        const bf16_t* qr = q + (long)b * p.q_sb + (long)h * p.q_sh + n * p.q_sn + 4 * g;
        uint2 bv[D16];
#pragma unroll
        for (int u = 0; u < D16; ++u) bv[u] = *reinterpret_cast<const uint2*>(qr + 16 * u);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < D16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][0], bf16lo(bv[u].x), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][1], bf16hi(bv[u].x), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][2], bf16lo(bv[u].y), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][3], bf16hi(bv[u].y), acc, 0, 0, 0);
        }
        // lane (l15, g) holds the triple's terms k0 + 4 g .. + 3
        if (live) {
            float* o = outp + (((long)b * p.Hh + h) * N + n) * K;
            const int kk = k0 + 4 * g;
            if ((K & 3) == 0 && kk + 3 < K) *reinterpret_cast<f32x4*>(o + kk) = acc;
            else if ((K & 1) == 0) {
                if (kk + 1 < K) *reinterpret_cast<float2*>(o + kk) = make_float2(acc[0], acc[1]);
                if (kk + 3 < K) *reinterpret_cast<float2*>(o + kk + 2) = make_float2(acc[2], acc[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (kk + e < K) o[kk + e] = acc[e];
            }
        }
    }
}

// Row softmax for the first-stage AttnBlock (diffusionmodules/model.py:188-192: single head, c = 512, N = h*w = 4096 tokens): the
// logits are materialised by the GEMM kernel in fp32 (N x N per image: 67 MB at 512 px — outside the denoising loop, once per
// image) and this kernel writes softmax(scale * S) as the bf16 operand of the P V GEMM.  One block per row.
__global__ __launch_bounds__(NT) void softmax_rows_kernel(const float* S, long lds, bf16_t* P, long ldp, int cols, float scale) {
    __shared__ float red[NT / 64];
    const float* src = S + (long)blockIdx.x * lds;
    bf16_t* dst = P + (long)blockIdx.x * ldp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < cols; c += NT) mx = fmaxf(mx, src[c]);
    mx = wave_reduce_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    const float c2 = scale * 1.4426950408889634f;  // scale > 0: max of the scaled row = scale * max
    float sum = 0.f;
    for (int c = threadIdx.x; c < cols; c += NT) sum += __builtin_amdgcn_exp2f((src[c] - mx) * c2);
    sum = wave_reduce_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int c = threadIdx.x; c < cols; c += NT) dst[c] = f32_to_bf16(__builtin_amdgcn_exp2f((src[c] - mx) * c2) * inv);
}

// DiagonalGaussianDistribution (distributions.py:25-37): moments [B, 2C, HW] -> mean, logvar clamped to [-30, 20], std = exp(logvar/2),
// z = mean + std * noise (noise = NULL -> z = mean, the .mode()).
__global__ __launch_bounds__(NT) void gaussian_kernel(const float* moments, const float* noise, float* z, float* mean_o, float* logvar_o,
                                                      float* std_o, long per_half, long n) {
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const long b = i / per_half, r = i - b * per_half;
        const float mean = moments[b * 2 * per_half + r];
        const float logvar = fminf(fmaxf(moments[b * 2 * per_half + per_half + r], -30.0f), 20.0f);
        const float sd = expf(0.5f * logvar);
        z[i] = noise ? mean + sd * noise[i] : mean;
        if (mean_o) mean_o[i] = mean;
        if (logvar_o) logvar_o[i] = logvar;
        if (std_o) std_o[i] = sd;
    }
}

// PatchEmbed im2col: x [B,Cin,H,W] fp32 -> patches [B*(H/P)*(W/P), Cin*P*P] bf16, K order (c, ky, kx) = conv weight order
__global__ void patchify_kernel(const float* x, bf16_t* y, int B, int Cin, int H, int W, int P) {
    const int gh = H / P, gw = W / P, K = Cin * P * P;
    const long total = (long)B * gh * gw * K;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int k = (int)(i % K);
        long r = i / K;
        const int px = (int)(r % gw); r /= gw;
        const int py = (int)(r % gh); r /= gh;
        const int b = (int)r;
        const int c = k / (P * P), kk = k % (P * P), ky = kk / P, kx = kk % P;
        y[i] = f32_to_bf16(x[(((long)b * Cin + c) * H + py * P + ky) * W + px * P + kx]);
    }
}

// sum((a-b)^2) -> out[0] via one atomicAdd per block (caller zeroes out first; scaled by inv_n)
__global__ void mse_kernel(const float* a, const float* b, float* out, long n, float inv_n) {
    __shared__ float red[4];
    float s = 0.f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) {
        const float d = a[i] - b[i];
        s += d * d;
    }
    s = wave_reduce_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}

inline unsigned grid_for(long n, int per_thread = 1) {
    long b = (n + (long)NT * per_thread - 1) / ((long)NT * per_thread);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

// DPM-Solver general updates (dpm_solver.py:469-722 singlestep first / second / third, :780-826 multistep third): every one of them is
// x_t = c0 x + c1 m_a + c2 m_b + c3 m_c with host-side fp32 scalars — one launch instead of the reference's 6-15 elementwise torch ops.
__global__ __launch_bounds__(256) void lincomb4_kernel(float* __restrict__ out, const float* x0, float c0, const float* x1, float c1,
                                                        const float* x2, float c2, const float* x3, float c3, long n) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        f32x4 a = *reinterpret_cast<const f32x4*>(x0 + i);
        f32x4 r = {a[0] * c0, a[1] * c0, a[2] * c0, a[3] * c0};
        if (x1) { const f32x4 b = *reinterpret_cast<const f32x4*>(x1 + i); for (int e = 0; e < 4; ++e) r[e] = fmaf(b[e], c1, r[e]); }
        if (x2) { const f32x4 b = *reinterpret_cast<const f32x4*>(x2 + i); for (int e = 0; e < 4; ++e) r[e] = fmaf(b[e], c2, r[e]); }
        if (x3) { const f32x4 b = *reinterpret_cast<const f32x4*>(x3 + i); for (int e = 0; e < 4; ++e) r[e] = fmaf(b[e], c3, r[e]); }
        *reinterpret_cast<f32x4*>(out + i) = r;
    } else {
        for (long j = i; j < n; ++j) {
            float r = x0[j] * c0;
            if (x1) r = fmaf(x1[j], c1, r);
            if (x2) r = fmaf(x2[j], c2, r);
            if (x3) r = fmaf(x3[j], c3, r);
            out[j] = r;
        }
    }
}

// Error estimate of the adaptive DPM-Solver (dpm_solver.py:925-927): per sample sqrt(mean(((x_higher - x_lower) / delta)^2)) with
// delta = max(atol, rtol * max(|x_lower|, |x_prev|)).  One block per sample, fixed-order reduction (deterministic).
__global__ __launch_bounds__(1024) void dpm_adaptive_err_kernel(const float* __restrict__ xl, const float* __restrict__ xh,
                                                                const float* __restrict__ xp, float atol, float rtol, long n, float* out) {
    __shared__ float red[16];
    const long base = (long)blockIdx.x * n;
    float acc = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float l = xl[base + i];
        const float delta = fmaxf(atol, rtol * fmaxf(fabsf(l), fabsf(xp[base + i])));
        const float v = (xh[base + i] - l) / delta;
        acc = fmaf(v, v, acc);
    }
    acc = wave_reduce_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[blockIdx.x] = sqrtf(t / (float)n);
    }
}

extern "C" int ae_transpose_last2(const void* in, void* out, int B, int X, int Y, int Xpad, int in_bf16, int out_bf16, void* stream) {
    AE_REQUIRE(in && out && B > 0 && X > 0 && Y > 0 && Xpad >= X, "ae_transpose_last2: bad arguments");
    AE_REQUIRE(B <= 65535, "ae_transpose_last2: batch %d too large", B);
    dim3 grid((Xpad + 31) / 32, (Y + 31) / 32, B);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out, X, Y, Xpad, in_bf16, out_bf16);
    return ae_check_launch("ae_transpose_last2");
}

extern "C" int ae_concat_channels_bf16(const void* a, int Ca, const void* b, int Cb, void* y, long rows, void* stream) {
    AE_REQUIRE(a && b && y && rows > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0, "ae_concat_channels_bf16: bad arguments");
    hipLaunchKernelGGL(concat_kernel, dim3(grid_for(rows * ((Ca + Cb) / 8))), dim3(NT), 0, (hipStream_t)stream,
                       (const bf16_t*)a, Ca, (const bf16_t*)b, Cb, (bf16_t*)y, rows);
    return ae_check_launch("ae_concat_channels_bf16");
}

extern "C" int ae_im2col3x3_c8_bf16(const void* x, void* y, int B, int H, int W, void* stream) {
    AE_REQUIRE(x && y && B > 0 && H > 0 && W > 0, "ae_im2col3x3_c8_bf16: bad arguments");
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "ae_im2col3x3_c8_bf16: 16-byte alignment");
    hipLaunchKernelGGL(im2col3x3_c8_kernel, dim3(grid_for((long)B * H * W * 16)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W);
    return ae_check_launch("ae_im2col3x3_c8_bf16");
}

extern "C" int ae_split_channels_bf16(const void* y, int Ca, int Cb, void* a, void* b, long rows, int accumulate_a, int accumulate_b, void* stream) {
    AE_REQUIRE(y && (a || b) && rows > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0, "ae_split_channels_bf16: bad arguments");
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(rows * ((Ca + Cb) / 8))), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)y, Ca, Cb, (bf16_t*)a,
                       (bf16_t*)b, rows, accumulate_a, accumulate_b);
    return ae_check_launch("ae_split_channels_bf16");
}

extern "C" int ae_timestep_embedding(const long* t_i64, const float* t_f32, void* out_bf16, float* out_f32, int B, int dim,
                                     float max_period, void* stream) {
    AE_REQUIRE((t_i64 != nullptr) != (t_f32 != nullptr), "ae_timestep_embedding: pass exactly one of t_i64 / t_f32");
    AE_REQUIRE((out_bf16 || out_f32) && B > 0 && dim > 1, "ae_timestep_embedding: bad arguments");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((B * dim + NT - 1) / NT), dim3(NT), 0, (hipStream_t)stream, t_i64, t_f32,
                       (bf16_t*)out_bf16, out_f32, B, dim, max_period);
    return ae_check_launch("ae_timestep_embedding");
}

extern "C" int ae_ddim_step_f32(const float* x, const float* eps, const float* noise, float* x_prev, float* pred_x0, float* e_out,
                                long n, int branches, float s0, float s1, float sqrt_one_minus_at, float sqrt_at,
                                float sqrt_a_prev, float dir_coef, float sigma_t, float temperature, void* stream) {
    AE_REQUIRE(x && eps && x_prev && n > 0, "ae_ddim_step_f32: null pointer / bad n");
    AE_REQUIRE(branches >= 1 && branches <= 3, "ae_ddim_step_f32: branches must be 1, 2 or 3 (got %d)", branches);
    DdimArgs a{x, eps, noise, x_prev, pred_x0, e_out, n, branches, s0, s1, sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma_t, temperature};
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, a);
    return ae_check_launch("ae_ddim_step_f32");
}

extern "C" int ae_ddim_encode_step_f32(const float* x, const float* eps, float* x_next, long n, int branches, float scale, float cx,
                                       float ce, void* stream) {
    AE_REQUIRE(x && eps && x_next && n > 0, "ae_ddim_encode_step_f32: null pointer / bad n");
    AE_REQUIRE(branches == 1 || branches == 2, "ae_ddim_encode_step_f32: branches must be 1 or 2 (got %d)", branches);
    hipLaunchKernelGGL(ddim_encode_step_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, x, eps, x_next, n, branches, scale, cx, ce);
    return ae_check_launch("ae_ddim_encode_step_f32");
}

extern "C" int ae_plms_combine_f32(const float* e_t, const float* old1, const float* old2, const float* old3, float* out, long n, int order,
                                   void* stream) {
    AE_REQUIRE(e_t && out && n > 0 && order >= 0 && order <= 3, "ae_plms_combine_f32: bad arguments (order must be 0..3)");
    AE_REQUIRE(old1 && (order < 2 || old2) && (order < 3 || old3), "ae_plms_combine_f32: order %d needs that many previous predictions", order);
    hipLaunchKernelGGL(plms_combine_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, e_t, old1, old2, old3, out, n, order);
    return ae_check_launch("ae_plms_combine_f32");
}

extern "C" int ae_mask_blend_f32(const float* img, const float* x0, const float* noise, const float* mask, float* out, int B,
                                 int C, int HW, float sqrt_ac, float sqrt_one_minus_ac, int ip2p_order, void* stream) {
    AE_REQUIRE(img && x0 && noise && mask && out && B > 0 && C > 0 && HW > 0, "ae_mask_blend_f32: bad arguments");
    const long n = (long)B * C * HW;
    hipLaunchKernelGGL(mask_blend_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, img, x0, noise, mask, out, n, C, HW,
                       sqrt_ac, sqrt_one_minus_ac, ip2p_order);
    return ae_check_launch("ae_mask_blend_f32");
}

extern "C" int ae_q_sample_f32(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_one_minus_ac, float* out,
                               int B, long per_sample, void* stream) {
    AE_REQUIRE(x0 && noise && sqrt_ac && sqrt_one_minus_ac && out && B > 0 && per_sample > 0, "ae_q_sample_f32: bad arguments");
    const long n = (long)B * per_sample;
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, x0, noise, sqrt_ac, sqrt_one_minus_ac, out,
                       per_sample, n);
    return ae_check_launch("ae_q_sample_f32");
}

extern "C" int ae_silu_to_bf16(const void* x, int in_bf16, void* y, long n, void* stream) {
    AE_REQUIRE(x && y && n > 0, "ae_silu_to_bf16: bad arguments");
    hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, x, in_bf16, (bf16_t*)y, n);
    return ae_check_launch("ae_silu_to_bf16");
}

extern "C" int ae_add_bcast_bf16(const void* x, const void* p, void* y, long n, long period, void* stream) {
    AE_REQUIRE(x && p && y && n > 0 && period > 0 && n % 8 == 0 && period % 8 == 0 && n % period == 0, "ae_add_bcast_bf16: bad arguments");
    hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for(n, 8)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)p,
                       (bf16_t*)y, n, period);
    return ae_check_launch("ae_add_bcast_bf16");
}

extern "C" int ae_resample2x_rows_bf16(const void* x, void* y, int B, int H, int W, int C, int mode, void* stream) {
    AE_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (mode == 0 || mode == 1), "ae_resample2x_rows_bf16: bad arguments (C %% 8 == 0, mode 0 = nearest x2, 1 = 2x2 mean)");
    AE_REQUIRE(mode == 0 || (H >= 2 && W >= 2), "ae_resample2x_rows_bf16: a 2x2 mean needs H, W >= 2");
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "ae_resample2x_rows_bf16: 16-byte alignment");
    const long n = (long)B * (mode ? H / 2 : 2 * H) * (mode ? W / 2 : 2 * W) * (C / 8);
    hipLaunchKernelGGL(resample2x_rows_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)y, B, H, W, C, mode);
    return ae_check_launch("ae_resample2x_rows_bf16");
}

extern "C" int ae_scale_shift_rows_bf16(const void* x, const float* emb, long ld_emb, void* y, int B, long HW, int C, int silu, void* stream) {
    AE_REQUIRE(x && emb && y && B > 0 && HW > 0 && C > 0 && C % 8 == 0 && ld_emb >= 2L * C, "ae_scale_shift_rows_bf16: bad arguments (C %% 8 == 0, emb rows hold scale | shift = 2 C floats)");
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)emb & 3) == 0, "ae_scale_shift_rows_bf16: alignment");
    const long n = (long)B * HW * (C / 8);
    hipLaunchKernelGGL(scale_shift_rows_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, emb, ld_emb, (bf16_t*)y, HW, C, n, silu);
    return ae_check_launch("ae_scale_shift_rows_bf16");
}

extern "C" int ae_window_partition_bf16(const void* x, void* windows, int B, int H, int W, int C, int ws, int reverse, void* stream) {
    AE_REQUIRE(x && windows && B > 0 && H > 0 && W > 0 && ws > 0 && C % 8 == 0, "ae_window_partition_bf16: bad arguments");
    const int nH = (H + ws - 1) / ws, nW = (W + ws - 1) / ws;
    const long total = (long)B * nH * nW * ws * ws * (C / 8);
    // forward: x=image -> windows ; reverse: x=windows -> image (argument order stays (image, windows))
    if (!reverse)
        hipLaunchKernelGGL(window_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)x, (bf16_t*)windows, B, H, W, C, ws, nH, nW, 0);
    else
        hipLaunchKernelGGL(window_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, (const bf16_t*)windows, (bf16_t*)x, B, H, W, C, ws, nH, nW, 1);
    return ae_check_launch("ae_window_partition_bf16");
}

extern "C" int ae_sam_relpos_terms(const void* q, long q_sb, long q_sh, long q_sn, const float* Rh, const float* Rw, float* rel_h,
                                   float* rel_w, int B, int heads, int qH, int qW, int kH, int kW, int D, void* stream) {
    AE_REQUIRE(q && Rh && Rw && rel_h && rel_w && B > 0 && heads > 0 && D > 0, "ae_sam_relpos_terms: bad arguments");
    AE_REQUIRE((long)B * heads <= 65535, "ae_sam_relpos_terms: B*heads too large for grid.y");
    // round 6: both terms in ONE launch on the fp32 matrix pipe (relpos_mfma_kernel).  AE_RELPOS_MFMA=0: the two line launches (A/B)
    static const int mf = getenv("AE_RELPOS_MFMA") ? atoi(getenv("AE_RELPOS_MFMA")) : 1;
    if (mf && D % 16 == 0 && D <= 96 && (q_sn % 4) == 0 && (q_sb % 4) == 0 && (q_sh % 4) == 0 && (reinterpret_cast<uintptr_t>(q) & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(Rh) & 15) == 0 && (reinterpret_cast<uintptr_t>(Rw) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(rel_h) & 15) == 0 && (reinterpret_cast<uintptr_t>(rel_w) & 15) == 0) {
        RelposArgs p{};
        p.q_sb = q_sb; p.q_sh = q_sh; p.q_sn = q_sn;
        p.BH = B * heads; p.Hh = heads; p.qH = qH; p.qW = qW; p.kH = kH; p.kW = kW;
        const long i0 = (long)qH * ((kH + 15) / 16) * (((long)p.BH * qW + 63) / 64), i1 = (long)qW * ((kW + 15) / 16) * (((long)p.BH * qH + 63) / 64);
        AE_REQUIRE(i0 + i1 < (1L << 31), "ae_sam_relpos_terms: too many wave items");
        p.items[0] = (int)i0; p.items[1] = (int)i1;
        const unsigned blocks = (unsigned)((i0 + i1 + 3) / 4);
        hipStream_t s = (hipStream_t)stream;
#define AE_RPM(n) hipLaunchKernelGGL(relpos_mfma_kernel<n>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)q, Rh, Rw, rel_h, rel_w, p)
        switch (D / 16) { case 1: AE_RPM(1); break; case 2: AE_RPM(2); break; case 3: AE_RPM(3); break; case 4: AE_RPM(4); break;
                          case 5: AE_RPM(5); break; default: AE_RPM(6); break; }
#undef AE_RPM
        return ae_check_launch("ae_sam_relpos_terms(mfma)");
    }
    if (qH <= 64 && qW <= 64 && kH <= 64 && kW <= 64 && D % 8 == 0 && D <= 160 && (q_sn % 8) == 0 && (q_sb % 8) == 0 && (q_sh % 8) == 0 &&
        (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (reinterpret_cast<uintptr_t>(Rh) & 15) == 0 && (reinterpret_cast<uintptr_t>(Rw) & 15) == 0) {
        const size_t lds = (size_t)2 * 64 * (D + 4) * sizeof(float);
        static bool attr_done = false;
        if (lds > 64 * 1024 && !attr_done) {
            hipFuncSetAttribute(reinterpret_cast<const void*>(&relpos_line_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute(reinterpret_cast<const void*>(&relpos_line_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_done = true;
        }
        hipLaunchKernelGGL(relpos_line_kernel<0>, dim3(B * heads * qH), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q, q_sb, q_sh, q_sn,
                           Rh, Rw, rel_h, rel_w, heads, qH, qW, kH, kW, D);
        int rc = ae_check_launch("ae_sam_relpos_terms(rel_h)");
        if (rc) return rc;
        hipLaunchKernelGGL(relpos_line_kernel<1>, dim3(B * heads * qW), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q, q_sb, q_sh, q_sn,
                           Rh, Rw, rel_h, rel_w, heads, qH, qW, kH, kW, D);
        return ae_check_launch("ae_sam_relpos_terms(rel_w)");
    }
    dim3 grid(qH * qW, B * heads);
    hipLaunchKernelGGL(relpos_kernel, grid, dim3(128), D * sizeof(float), (hipStream_t)stream, (const bf16_t*)q, q_sb, q_sh, q_sn, Rh, Rw,
                       rel_h, rel_w, B, heads, qH, qW, kH, kW, D);
    return ae_check_launch("ae_sam_relpos_terms");
}

extern "C" int ae_softmax_rows_f32_bf16(const float* S, long lds, void* P, long ldp, int rows, int cols, float scale, void* stream) {
    AE_REQUIRE(S && P && rows > 0 && cols > 0 && lds >= cols && ldp >= cols, "ae_softmax_rows_f32_bf16: bad arguments");
    AE_REQUIRE(scale > 0.f, "ae_softmax_rows_f32_bf16: scale must be positive");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(NT), 0, (hipStream_t)stream, S, lds, (bf16_t*)P, ldp, cols, scale);
    return ae_check_launch("ae_softmax_rows_f32_bf16");
}

extern "C" int ae_gaussian_moments_f32(const float* moments, const float* noise, float* z, float* mean, float* logvar, float* std_out,
                                       int B, long per_half, void* stream) {
    AE_REQUIRE(moments && z && B > 0 && per_half > 0, "ae_gaussian_moments_f32: bad arguments");
    const long n = (long)B * per_half;
    hipLaunchKernelGGL(gaussian_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, moments, noise, z, mean, logvar, std_out,
                       per_half, n);
    return ae_check_launch("ae_gaussian_moments_f32");
}

extern "C" int ae_dpm_multistep_f32(const float* x, const float* model_out, const float* m_prev, float* m_cur, float* x_next, long n,
                                    int branches, float scale, int v_param, int predict_x0, float sigma_s, float alpha_s, int update,
                                    float a, float b, float c, float inv_r0, void* stream) {
    AE_REQUIRE(x && model_out && n > 0, "ae_dpm_multistep_f32: null pointer / bad n");
    AE_REQUIRE(branches == 1 || branches == 2, "ae_dpm_multistep_f32: branches must be 1 or 2 (got %d)", branches);
    AE_REQUIRE(m_cur || update, "ae_dpm_multistep_f32: nothing to write");
    AE_REQUIRE(!update || x_next, "ae_dpm_multistep_f32: update needs x_next");
    DpmArgs p{x, model_out, m_prev, m_cur, x_next, n, branches, v_param, predict_x0, update, scale, sigma_s, alpha_s, a, b, c, inv_r0};
    hipLaunchKernelGGL(dpm_multistep_kernel, dim3(grid_for(n)), dim3(NT), 0, (hipStream_t)stream, p);
    return ae_check_launch("ae_dpm_multistep_f32");
}

extern "C" int ae_patchify_f32_bf16(const float* x, void* y, int B, int Cin, int H, int W, int P, void* stream) {
    AE_REQUIRE(x && y && B > 0 && Cin > 0 && P > 0 && H % P == 0 && W % P == 0, "ae_patchify_f32_bf16: bad arguments");
    const long total = (long)B * Cin * H * W;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, x, (bf16_t*)y, B, Cin, H, W, P);
    return ae_check_launch("ae_patchify_f32_bf16");
}

extern "C" int ae_lincomb4_f32(float* out, const float* x0, float c0, const float* x1, float c1, const float* x2, float c2,
                               const float* x3, float c3, long n, void* stream) {
    AE_REQUIRE(out && x0 && n > 0, "ae_lincomb4_f32: null pointer / empty tensor");
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    AE_REQUIRE(al16(out) && al16(x0) && (!x1 || al16(x1)) && (!x2 || al16(x2)) && (!x3 || al16(x3)), "ae_lincomb4_f32: pointers must be 16-byte aligned");
    const long blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(lincomb4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, x0, c0, x1, c1, x2, c2, x3, c3, n);
    return ae_check_launch("ae_lincomb4_f32");
}

extern "C" int ae_dpm_adaptive_err_f32(const float* x_lower, const float* x_higher, const float* x_prev, float atol, float rtol, int B,
                                       long n_per_sample, float* out, void* stream) {
    AE_REQUIRE(x_lower && x_higher && x_prev && out && B > 0 && n_per_sample > 0, "ae_dpm_adaptive_err_f32: null pointer / empty tensor");
    AE_REQUIRE(atol > 0.f && rtol >= 0.f, "ae_dpm_adaptive_err_f32: atol must be positive, rtol non-negative");
    hipLaunchKernelGGL(dpm_adaptive_err_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream, x_lower, x_higher, x_prev, atol, rtol,
                       n_per_sample, out);
    return ae_check_launch("ae_dpm_adaptive_err_f32");
}

extern "C" int ae_mse_f32(const float* a, const float* b, float* out, long n, void* stream) {
    AE_REQUIRE(a && b && out && n > 0, "ae_mse_f32: bad arguments");
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) { ae_set_error("ae_mse_f32: memset failed: %s", hipGetErrorString(e)); return AE_ERR_LAUNCH; }
    hipLaunchKernelGGL(mse_kernel, dim3(grid_for(n) > 1024 ? 1024 : grid_for(n)), dim3(NT), 0, (hipStream_t)stream, a, b, out, n, 1.0f / (float)n);
    return ae_check_launch("ae_mse_f32");
}
