// SAM prompt-encoder / mask-decoder kernels for gfx950 — SURVEY.md §8(f) N3: what SamPredictor.predict_torch runs after the image
// encoder (segment_anything/predictor.py:168-245).  The matmul-shaped work (projections, MLPs, the two transposed convolutions as
// GEMMs over un-shuffled pixels) goes through ae_gemm_bf16 / ae_attn_fwd_bf16; this file holds the HBM-bound remainder, fused so that
// no permuted or padded intermediate is ever written:
//   * ae_sam_pe_encode_f32        random-Fourier positional encoding + label-selected learned offsets (prompt_encoder.py:73-102,183-214)
//   * ae_sam_mask_downscale_bf16  conv k2s2 -> LN2d -> GELU -> conv k2s2 -> LN2d -> GELU of a mask prompt in one pass (:46-54)
//   * ae_sam_mask_product_f32     hypernetwork-weights x upscaled-embedding product reading the un-shuffled ConvTranspose output
//                                 (mask_decoder.py:134-149) — the pixel shuffle of both transposed convolutions is folded into the read
//   * ae_sam_postprocess_masks    Sam.postprocess_masks (sam.py:133-162): bilinear -> crop -> bilinear composed per output pixel,
//                                 optional fused threshold, no 1024x1024 intermediate
//   * ae_sam_preprocess_f32       Sam.preprocess (sam.py:164-174): normalise + zero-pad to the square input
#include "common.hpp"

namespace {

constexpr int NT = 256;

inline unsigned blocks_for(long n) {
    long b = (n + NT - 1) / NT;
    if (b > 65535L * 16) b = 65535L * 16;
    return (unsigned)(b < 1 ? 1 : b);
}

// out[n, f] = sin(2 pi ((2x-1) g0[f] + (2y-1) g1[f])), out[n, F+f] = cos(..), then the learned offset of the point's label.
// labels == nullptr: pure encoding.  label -1: encoding zeroed, table row 0 (not_a_point); label l >= 0: + table row 1 + l.
__global__ void pe_encode_kernel(const float* coords, const int* labels, const float* gauss, const float* table, float* out, int N,
                                 int F, float offset, float inv_w, float inv_h) {
    const long total = (long)N * F;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int f = (int)(i % F);
        const long n = i / F;
        const float x = (coords[2 * n] + offset) * inv_w, y = (coords[2 * n + 1] + offset) * inv_h;
        const float ph = ((2.f * x - 1.f) * gauss[f] + (2.f * y - 1.f) * gauss[F + f]) * 6.283185307179586f;
        float s = sinf(ph), c = cosf(ph);
        if (labels) {
            const int l = labels[n];
            const float* row = table + (long)(l < 0 ? 0 : 1 + l) * 2 * F;
            if (l < 0) s = c = 0.f;
            s += row[f];
            c += row[F + f];
        }
        out[n * 2 * F + f] = s;
        out[n * 2 * F + F + f] = c;
    }
}

struct MaskDownArgs {
    const float* m;      // [B, 1, 4h, 4w]
    const float* w1;     // [4, 1, 2, 2]
    const float* b1; const float* g1; const float* e1;   // conv bias, LN weight, LN bias  [4]
    const float* w2;     // [16, 4, 2, 2]
    const float* b2; const float* g2; const float* e2;   // [16]
    bf16_t* out;         // rows [B*h*w, 16]
    int B, h, w;
    float eps;
};

// One thread per output pixel of the second convolution: its 4x4 input patch is four 16-byte loads; both LayerNorm2d's are over 4 / 16
// channels held in registers.  Weights are wave-uniform scalar loads.
__global__ __launch_bounds__(NT) void mask_downscale_kernel(const MaskDownArgs p) {
    const long total = (long)p.B * p.h * p.w;
    const long idx = (long)blockIdx.x * NT + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % p.w);
    const int y = (int)((idx / p.w) % p.h);
    const long b = idx / ((long)p.w * p.h);
    const int W4 = 4 * p.w;
    const float* src = p.m + (b * 4 * p.h + 4 * y) * W4 + 4 * x;
    float in[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (long)r * W4);
        in[r][0] = v[0]; in[r][1] = v[1]; in[r][2] = v[2]; in[r][3] = v[3];
    }
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = p.b2[o];
#pragma unroll
    for (int sy = 0; sy < 2; ++sy)
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
            float t[4], mu = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float v = p.b1[c];
#pragma unroll
                for (int ky = 0; ky < 2; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 2; ++kx) v += p.w1[c * 4 + ky * 2 + kx] * in[2 * sy + ky][2 * sx + kx];
                t[c] = v;
                mu += v;
            }
            mu *= 0.25f;
            float var = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) var += (t[c] - mu) * (t[c] - mu);
            const float rstd = 1.0f / sqrtf(var * 0.25f + p.eps);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a = gelu_erf_f((t[c] - mu) * rstd * p.g1[c] + p.e1[c]);
#pragma unroll
                for (int o = 0; o < 16; ++o) acc[o] += p.w2[(o * 4 + c) * 4 + sy * 2 + sx] * a;
            }
        }
    float mu = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) mu += acc[o];
    mu *= (1.0f / 16.f);
    float var = 0.f;
#pragma unroll
    for (int o = 0; o < 16; ++o) var += (acc[o] - mu) * (acc[o] - mu);
    const float rstd = 1.0f / sqrtf(var * (1.0f / 16.f) + p.eps);
    uint32_t o32[8];
#pragma unroll
    for (int o = 0; o < 16; o += 2) {
        const float a = gelu_erf_f((acc[o] - mu) * rstd * p.g2[o] + p.e2[o]);
        const float c = gelu_erf_f((acc[o + 1] - mu) * rstd * p.g2[o + 1] + p.e2[o + 1]);
        o32[o / 2] = pack_bf16x2(a, c);
    }
    u32x4* dst = reinterpret_cast<u32x4*>(p.out + idx * 16);
    dst[0] = (u32x4){o32[0], o32[1], o32[2], o32[3]};
    dst[1] = (u32x4){o32[4], o32[5], o32[6], o32[7]};
}

// masks[b, m, Y, X] = sum_c hyper[b, m, c] * up[b, pix(Y, X), c].  `up` is the second transposed convolution's GEMM output left in its
// natural order [b, y, x, dy1, dx1, dy2, dx2, c] (Y = 4y + 2 dy1 + dy2, X likewise): each thread reads the C contiguous channels of its
// output pixel (16-byte loads) and writes one fp32 per mask, coalesced over X.
template <int C>
__global__ __launch_bounds__(NT) void mask_product_kernel(const bf16_t* up, const float* hyper, float* out, int h, int w, int M) {
    __shared__ float sh[8 * C];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < M * C; i += NT) sh[i] = hyper[(long)b * M * C + i];
    __syncthreads();
    const int H4 = 4 * h, W4 = 4 * w;
    const long npix = (long)H4 * W4;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < npix; i += (long)gridDim.x * NT) {
        const int X = (int)(i % W4), Y = (int)(i / W4);
        const long pix = ((((long)(Y >> 2) * w + (X >> 2)) * 2 + ((Y >> 1) & 1)) * 2 + ((X >> 1) & 1)) * 4 + (Y & 1) * 2 + (X & 1);
        const bf16_t* src = up + ((long)b * npix + pix) * C;
        float v[C];
#pragma unroll
        for (int c8 = 0; c8 < C / 8; ++c8) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(src + c8 * 8);
            const uint32_t wv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[c8 * 8 + 2 * e] = bf16lo(wv[e]);
                v[c8 * 8 + 2 * e + 1] = bf16hi(wv[e]);
            }
        }
        for (int m = 0; m < M; ++m) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) s += sh[m * C + c] * v[c];
            out[((long)b * M + m) * npix + i] = s;
        }
    }
}

// F.interpolate(mode="bilinear", align_corners=False) source index: scale * (dst + 0.5) - 0.5 clamped at 0, upper neighbour clamped.
__device__ __forceinline__ void bilin_src(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

struct PostArgs {
    const float* low;   // [N, Hl, Wl]
    float* out_f32;     // [N, oh, ow] or null
    uint8_t* out_u8;    // [N, oh, ow] or null: out > threshold
    int N, Hl, Wl, S, ih, iw, oh, ow;
    float threshold;
    int merge;          // out_u8 [oh, ow] = OR over the N masks (tools/tool.py:239-241 mask_mode 'merge')
};

__device__ __forceinline__ float stage1_sample(const float* img, int Y, int X, const PostArgs& p, float sh, float sw) {
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_src(Y, sh, p.Hl, y0, y1, ly);
    bilin_src(X, sw, p.Wl, x0, x1, lx);
    const float a = img[y0 * p.Wl + x0], b = img[y0 * p.Wl + x1], c = img[y1 * p.Wl + x0], d = img[y1 * p.Wl + x1];
    return (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
}

__device__ __forceinline__ float postprocess_pixel(const float* img, int y, int x, const PostArgs& p, float sh1, float sw1, float sh2,
                                                   float sw2) {
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_src(y, sh2, p.ih, y0, y1, ly);
    bilin_src(x, sw2, p.iw, x0, x1, lx);
    const float a = stage1_sample(img, y0, x0, p, sh1, sw1), b = stage1_sample(img, y0, x1, p, sh1, sw1);
    const float c = stage1_sample(img, y1, x0, p, sh1, sw1), d = stage1_sample(img, y1, x1, p, sh1, sw1);
    return (1.f - ly) * ((1.f - lx) * a + lx * b) + ly * ((1.f - lx) * c + lx * d);
}

__global__ __launch_bounds__(NT) void postprocess_kernel(const PostArgs p) {
    const long per = (long)p.oh * p.ow, total = p.merge ? per : per * p.N;
    const float sh1 = (float)p.Hl / (float)p.S, sw1 = (float)p.Wl / (float)p.S;
    const float sh2 = (float)p.ih / (float)p.oh, sw2 = (float)p.iw / (float)p.ow;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int x = (int)(i % p.ow), y = (int)((i / p.ow) % p.oh);
        if (p.merge) {
            int any = 0;
            for (int n = 0; n < p.N; ++n) any |= postprocess_pixel(p.low + (long)n * p.Hl * p.Wl, y, x, p, sh1, sw1, sh2, sw2) > p.threshold;
            p.out_u8[i] = (uint8_t)any;
            continue;
        }
        const float v = postprocess_pixel(p.low + (i / per) * p.Hl * p.Wl, y, x, p, sh1, sw1, sh2, sw2);
        if (p.out_f32) p.out_f32[i] = v;
        if (p.out_u8) p.out_u8[i] = v > p.threshold ? 1 : 0;
    }
}

// Greedy non-maximum suppression over boxes already sorted by descending score (torchvision.ops.nms semantics, used by
// tools/tool.py:224): box i survives unless an earlier survivor overlaps it with IoU > thr.  One block; the boxes live in LDS and the
// N sequential rounds each clear the later boxes in parallel (detector outputs are tens of boxes, at most 900 queries).
__global__ __launch_bounds__(NT) void nms_kernel(const float* boxes, uint8_t* keep, int N, float thr) {
    extern __shared__ float sb[];          // [N][4] then N removed flags (as floats' bytes)
    uint8_t* removed = reinterpret_cast<uint8_t*>(sb + 4 * (long)N);
    for (int i = threadIdx.x; i < 4 * N; i += NT) sb[i] = boxes[i];
    for (int i = threadIdx.x; i < N; i += NT) removed[i] = 0;
    __syncthreads();
    for (int i = 0; i < N; ++i) {
        if (!removed[i]) {                  // uniform across the block (read after the barrier below)
            const float x1 = sb[4 * i], y1 = sb[4 * i + 1], x2 = sb[4 * i + 2], y2 = sb[4 * i + 3];
            const float ai = (x2 - x1) * (y2 - y1);
            for (int j = i + 1 + threadIdx.x; j < N; j += NT) {
                const float u1 = sb[4 * j], v1 = sb[4 * j + 1], u2 = sb[4 * j + 2], v2 = sb[4 * j + 3];
                const float iw = fmaxf(fminf(x2, u2) - fmaxf(x1, u1), 0.f), ih = fmaxf(fminf(y2, v2) - fmaxf(y1, v1), 0.f);
                const float inter = iw * ih, aj = (u2 - u1) * (v2 - v1);
                if (inter / (ai + aj - inter) > thr) removed[j] = 1;
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < N; i += NT) keep[i] = removed[i] ? 0 : 1;
}

template <typename TIn>
__global__ void preprocess_kernel(const TIn* x, float* y, int B, int C, int h, int w, int S, const float* mean, const float* stdv) {
    const long total = (long)B * C * S * S;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
        const int X = (int)(i % S), Y = (int)((i / S) % S);
        const long bc = i / ((long)S * S);
        const int c = (int)(bc % C);
        float v = 0.f;
        if (Y < h && X < w) v = ((float)x[(bc * h + Y) * w + X] - mean[c]) / stdv[c];
        y[i] = v;
    }
}

}  // namespace

extern "C" int ae_sam_pe_encode_f32(const float* coords, const int* labels, const float* gauss, const float* table, float* out, int N,
                                    int F, float offset, float inv_w, float inv_h, void* stream) {
    AE_REQUIRE(coords && gauss && out && N > 0 && F > 0, "ae_sam_pe_encode_f32: bad arguments");
    AE_REQUIRE(!labels || table, "ae_sam_pe_encode_f32: labels need the embedding table");
    hipLaunchKernelGGL(pe_encode_kernel, dim3(blocks_for((long)N * F)), dim3(NT), 0, (hipStream_t)stream, coords, labels, gauss, table,
                       out, N, F, offset, inv_w, inv_h);
    return ae_check_launch("ae_sam_pe_encode_f32");
}

extern "C" int ae_sam_mask_downscale_bf16(const float* masks, const float* w1, const float* b1, const float* g1, const float* e1,
                                          const float* w2, const float* b2, const float* g2, const float* e2, void* out, int B, int h,
                                          int w, float eps, void* stream) {
    AE_REQUIRE(masks && w1 && b1 && g1 && e1 && w2 && b2 && g2 && e2 && out, "ae_sam_mask_downscale_bf16: null pointer");
    AE_REQUIRE(B > 0 && h > 0 && w > 0, "ae_sam_mask_downscale_bf16: bad shape");
    AE_REQUIRE(((uintptr_t)masks & 15) == 0 && ((uintptr_t)out & 15) == 0, "ae_sam_mask_downscale_bf16: 16-byte alignment");
    MaskDownArgs p{masks, w1, b1, g1, e1, w2, b2, g2, e2, (bf16_t*)out, B, h, w, eps};
    const long total = (long)B * h * w;
    hipLaunchKernelGGL(mask_downscale_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, p);
    return ae_check_launch("ae_sam_mask_downscale_bf16");
}

extern "C" int ae_sam_mask_product_f32(const void* up, const float* hyper, float* out, int B, int h, int w, int M, int C, void* stream) {
    AE_REQUIRE(up && hyper && out && B > 0 && h > 0 && w > 0, "ae_sam_mask_product_f32: bad arguments");
    AE_REQUIRE(M >= 1 && M <= 8, "ae_sam_mask_product_f32: %d mask tokens (supported: 1..8)", M);
    AE_REQUIRE(B <= 65535, "ae_sam_mask_product_f32: batch %d too large", B);
    AE_REQUIRE(((uintptr_t)up & 15) == 0, "ae_sam_mask_product_f32: 16-byte alignment");
    const long npix = 16L * h * w;
    long gx = (npix + NT - 1) / NT;
    if (gx > 4096) gx = 4096;
    dim3 grid((unsigned)gx, B);
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 8: hipLaunchKernelGGL(mask_product_kernel<8>, grid, dim3(NT), 0, s, (const bf16_t*)up, hyper, out, h, w, M); break;
        case 16: hipLaunchKernelGGL(mask_product_kernel<16>, grid, dim3(NT), 0, s, (const bf16_t*)up, hyper, out, h, w, M); break;
        case 32: hipLaunchKernelGGL(mask_product_kernel<32>, grid, dim3(NT), 0, s, (const bf16_t*)up, hyper, out, h, w, M); break;
        case 64: hipLaunchKernelGGL(mask_product_kernel<64>, grid, dim3(NT), 0, s, (const bf16_t*)up, hyper, out, h, w, M); break;
        default:
            ae_set_error("ae_sam_mask_product_f32: unsupported channel count %d (supported: 8, 16, 32, 64)", C);
            return AE_ERR_UNSUPPORTED;
    }
    return ae_check_launch("ae_sam_mask_product_f32");
}

extern "C" int ae_nms_sorted_f32(const float* boxes, void* keep, int N, float iou_threshold, void* stream) {
    AE_REQUIRE(boxes && keep && N > 0, "ae_nms_sorted_f32: bad arguments");
    AE_REQUIRE(N <= 8192, "ae_nms_sorted_f32: %d boxes (supported: <= 8192)", N);
    const size_t lds = (size_t)N * 16 + (size_t)((N + 3) / 4 * 4);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { ae_set_error("ae_nms_sorted_f32: hipFuncSetAttribute(%zu) failed: %s", lds, hipGetErrorString(e)); return AE_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(NT), lds, (hipStream_t)stream, boxes, (uint8_t*)keep, N, iou_threshold);
    return ae_check_launch("ae_nms_sorted_f32");
}

extern "C" int ae_sam_postprocess_masks(const float* low, float* out_f32, void* out_u8, int N, int Hl, int Wl, int S, int ih, int iw,
                                        int oh, int ow, float threshold, int merge, void* stream) {
    AE_REQUIRE(low && (out_f32 || out_u8), "ae_sam_postprocess_masks: null pointer");
    AE_REQUIRE(N > 0 && Hl > 0 && Wl > 0 && S > 0 && oh > 0 && ow > 0, "ae_sam_postprocess_masks: bad shape");
    AE_REQUIRE(ih > 0 && iw > 0 && ih <= S && iw <= S, "ae_sam_postprocess_masks: input_size (%d, %d) must fit the %d-pixel square", ih, iw, S);
    AE_REQUIRE(!merge || (out_u8 && !out_f32), "ae_sam_postprocess_masks: merge writes the thresholded union only");
    PostArgs p{low, out_f32, (uint8_t*)out_u8, N, Hl, Wl, S, ih, iw, oh, ow, threshold, merge ? 1 : 0};
    hipLaunchKernelGGL(postprocess_kernel, dim3(blocks_for((long)(merge ? 1 : N) * oh * ow)), dim3(NT), 0, (hipStream_t)stream, p);
    return ae_check_launch("ae_sam_postprocess_masks");
}

extern "C" int ae_sam_preprocess_f32(const void* x, int x_is_u8, float* y, int B, int C, int h, int w, int S, const float* mean,
                                     const float* stdv, void* stream) {
    AE_REQUIRE(x && y && mean && stdv && B > 0 && C > 0, "ae_sam_preprocess_f32: bad arguments");
    AE_REQUIRE(h > 0 && w > 0 && h <= S && w <= S, "ae_sam_preprocess_f32: image (%d, %d) must fit the %d-pixel square", h, w, S);
    const unsigned g = blocks_for((long)B * C * S * S);
    if (x_is_u8) hipLaunchKernelGGL(preprocess_kernel<uint8_t>, dim3(g), dim3(NT), 0, (hipStream_t)stream, (const uint8_t*)x, y, B, C, h, w, S, mean, stdv);
    else hipLaunchKernelGGL(preprocess_kernel<float>, dim3(g), dim3(NT), 0, (hipStream_t)stream, (const float*)x, y, B, C, h, w, S, mean, stdv);
    return ae_check_launch("ae_sam_preprocess_f32");
}
