// anyedit_amd — shared device/host helpers for the gfx950 (CDNA4, wave64) kernels.
// No CUDA compatibility layer, no dual paths: this code targets MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits (storage type at the C ABI)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA 16x16x32 A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;       // MFMA 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(2))) float f32x2;       // packed fp32 pair (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;    // 16-byte global/LDS transaction
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define AE_OK 0
#define AE_ERR_ARG -1
#define AE_ERR_LAUNCH -2
#define AE_ERR_UNSUPPORTED -3

// error plumbing (c_api.hip)
void ae_set_error(const char* fmt, ...);
int ae_check_launch(const char* what);
// per-channel (sum, sum of squares) of a bf16 [M, N] tensor over each 32-row slab -> out [ceil(M/32)][N][2] fp32 (gemm_conv.hip): the
// statistics a producing kernel hands to the GroupNorm that consumes its output; N % 8 == 0, rows 16-byte aligned
int ae_launch_colstats(const uint16_t* x, long ld, int M, int N, float* out, hipStream_t stream);

// LayerNorm fold on the row-panel kernel (gemm_rowpanel.hip; K = 320 shapes of ae_gemm_ln_bf16 / ae_gemm_ln_plan, forwarded from gemm_conv.hip)
int ae_rowpanel_fold_covers(int M, int N, int K, int epilogue, int mode);
int ae_rowpanel_fold_launch(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias, const void* residual,
                            long ldr, int epilogue, float* rowstats_out, const float* ln_stats, int ln_parts, const float* ln_colsum, float ln_eps, void* stream);

#define AE_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            ae_set_error(__VA_ARGS__);        \
            return AE_ERR_ARG;                \
        }                                     \
    } while (0)

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// round-to-nearest-even in hardware: both forms compile to ONE v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    union { bf16x2_t b; uint32_t u; } c;
    c.b = (bf16x2_t){(__bf16)lo, (__bf16)hi};
    return c.u;
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ bf16x8_t as_bf16x8(u32x4 v) {
    union { u32x4 u; bf16x8_t b; } c;
    c.u = v;
    return c.b;
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;  // MFMA 16x16x16 bf16 A/B operand (2 VGPRs)
__device__ __forceinline__ s16x4_t as_s16x4(u32x2 v) {
    union { u32x2 u; s16x4_t s; } c;
    c.u = v;
    return c.s;
}

// Epilogue activations are VALU work that ADDS to the MFMA time (shared vector pipe), so they are written for instruction
// count: hardware rcp / exp2 (1 ulp, separate transcendental unit) instead of IEEE division and libm erff.
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// exact-erf GELU (F.gelu default, attention.py:56): erf by Abramowitz-Stegun 7.1.26, |abs error| <= 1.5e-7 — two orders below
// the bf16 rounding of the result.  ~13 VALU + 2 transcendental ops instead of ~35 for libm erff.
__device__ __forceinline__ float erf_as_f(float z) {
    const float az = __builtin_fabsf(z);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, az, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * az * az);
    const float r = __builtin_fmaf(-poly, e, 1.0f);
    return __builtin_copysignf(r, z);
}
// GELU for the fused epilogues, written for instruction count (round 4: the epilogue is 25-40 % of the short-K GEGLU / GELU launches, and two thirds of
// it is this arithmetic — profiles/r04_v31_epilogue_share.txt).  With w(x) = erf(|x| / sqrt 2) (the same 7.1.26 series, the 1 / sqrt 2 folded into its
// two constants):   gelu(x) = 0.5 x (1 + sign(x) w) = 0.5 (x + |x| w)   — no copysign, no 1 + erf, |x| is a free source modifier: 13 VALU + rcp + exp2
// instead of 16 + 2.  GEGLU's product a gelu(g) takes the 0.5 into a's bias add: (0.5 a) (g + |g| w) with 0.5 a = fma(acc, 0.5, 0.5 bias).
#ifndef AE_GELU_POLY
#define AE_GELU_POLY 0   // round 5, measured and NOT taken: 1 = w(x) by a polynomial alone (no v_rcp_f32 / v_exp_f32), 1.7e-5 absolute instead of 1.5e-7: UNet step 12.676 -> 12.636 ms in three
                         // alternating pairs (profiles/r05_v14_gelu_poly_ab.txt) — 0.3 % is not worth two orders of erf accuracy.  The form stays for A/B builds.
#endif
// Round 5: the K = 320 GEGLU launches are bound by this arithmetic, not by the matrix pipe (row-panel kernel: the chunk epilogue of a wave is ~2 250 cycles against
// 960 of MFMAs, profiles/r02_gemm_rowpanel_lab.txt), and a transcendental issues at ~1.8 x a plain VALU operation.  w(x) = erf(|x| / sqrt 2) for |x| <= 4.25 as
// |x| P(x^2), P of degree 8 (near-minimax fit, Horner in t = 2 x^2 / 4.25^2 - 1 so that fp32 evaluation is well conditioned), |x| clamped to 4.25 beyond
// (erf(4.25 / sqrt 2) = 1 - 2.1e-5): max |w - erf| = 1.7e-5 over the whole line INCLUDING fp32 evaluation error (tools: the fit and the check are in DESIGN.md §7.00),
// i.e. |delta gelu(x)| <= 0.5 |x| 1.7e-5 — a hundredth of the bf16 rounding of the stored result for x > 0; w stays below 1 (no sign flip of the tail).
// 12 VALU, no transcendental, against 13 + 2.
__device__ __forceinline__ float gelu_w_poly_f(float x) {
    const float ax = __builtin_fminf(__builtin_fabsf(x), 4.25f);
    const float t = __builtin_fmaf(ax * ax, 0.11072664707899094f, -1.0f);
    float p = __builtin_fmaf(0.004925727378576994f, t, -0.012811211869120598f);
    p = __builtin_fmaf(p, t, 0.017165469005703926f);
    p = __builtin_fmaf(p, t, -0.02821926586329937f);
    p = __builtin_fmaf(p, t, 0.05118509382009506f);
    p = __builtin_fmaf(p, t, -0.07870230078697205f);
    p = __builtin_fmaf(p, t, 0.11140184104442596f);
    p = __builtin_fmaf(p, t, -0.1615230143070221f);
    p = __builtin_fmaf(p, t, 0.33187076449394226f);
    return ax * p;
}
__device__ __forceinline__ float gelu_w_f(float x) {
    if (AE_GELU_POLY) return gelu_w_poly_f(x);
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    poly *= t;
    const float e = __builtin_amdgcn_exp2f((-0.5f * 1.4426950408889634f) * ax * ax);   // exp(-x^2 / 2)
    return __builtin_fmaf(-poly, e, 1.0f);
}
#ifndef AE_GELU_OLD
#define AE_GELU_OLD 0   // 1: the round-1 form 0.5 x (1 + erf(x / sqrt 2)) through erf_as_f (A/B builds)
#endif
__device__ __forceinline__ float gelu_erf_f(float x) {
    if (AE_GELU_OLD) return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f));
    return 0.5f * __builtin_fmaf(__builtin_fabsf(x), gelu_w_f(x), x);
}
// a_half * 2 gelu(g) = a gelu(g) for a_half = 0.5 a
__device__ __forceinline__ float geglu_half_f(float a_half, float g) {
    if (AE_GELU_OLD) return (a_half + a_half) * gelu_erf_f(g);
    return a_half * __builtin_fmaf(__builtin_fabsf(g), gelu_w_f(g), g);
}

// XCD-aware, bijective block remap: hardware places block b on XCD b % 8.  Give every XCD a contiguous
// run of logical tile ids so neighbouring tiles (which share operand panels) hit the same private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int NX = 8;
    int xcd = bid % NX, j = bid / NX;
    int q = nblocks / NX, r = nblocks % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + j;
}

// LDS-DMA, one wave: 64 x 16 B from global (buffer descriptor + per-lane byte offset + wave-uniform byte offset) straight to LDS
// at lds_byte + lane * 16.  Issued through inline asm ON PURPOSE: hipcc treats the builtin form as an LDS store that may alias
// every later LDS read and drains it (s_waitcnt vmcnt(0)) in front of the next ds_read, which turns a prefetch ring into a
// synchronous copy.  The asm form is invisible to that bookkeeping: the kernel places its own counted s_waitcnt vmcnt(N).
// N must count only DMA pieces issued AFTER the one waited for: other VMEM traffic of the wave (stores, loads) can only make
// the wait stricter, never let it pass early.
__device__ __forceinline__ void ae_dma16(__amdgpu_buffer_rsrc_t rs, int lds_byte, int voffset, int soffset = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voffset), "s"(rs), "s"(lds_byte), "s"(soffset)
                 : "memory", "m0");
#endif
}

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
