// fp8 (OCP e4m3) attention forward for gfx950 (MI355X): out = softmax(q k^T * scale + bias) v with Q, K, V and P in e4m3, fp32 logits /
// softmax statistics / accumulation — BASELINE.json configs[4] ("fp8 MFMA attention") for the SAM ViT image encoder.
//
// Replaces: segment_anything/modeling/image_encoder.py:224-240 Attention.forward with the decomposed
// relative-position bias of :325-361 (rel_h[q, key / kW] + rel_w[q, key % kW], from the UNSCALED q — ae_sam_relpos_terms).
//
// Only the block-scaled K = 64 / 128 MFMAs (v_mfma_scale_f32_32x32x64_f8f6f4) run faster than bf16 on gfx950 (tools/ubench/
// mfma_rates.hip: the unscaled fp8 16x16x32 issues at the bf16 rate), so the kernel is built on them:
//   1. ae_attn_fp8_prepare: per (batch, head) amax of q / k / v (one pass), then one pass that writes
//        Q8  [BH][Nq][128]  = q * (scale log2e s_k) / 2^e_q     (head_dim padded to 128 with zeros; 2^e_q rides in the MFMA's
//                                                                 E8M0 scale operand, so the logits come out in log2 units)
//        K8  [BH][Nk'][128] = k / s_k                             (s_k = amax_k / 448)
//        Vt8 [BH][96][Nk']  = (v / 2^e_v)^T, keys permuted inside each 64-key tile to the MFMA's contraction order, row
//                             `head_dim` = 1.0 for real keys (the PV MFMA then also yields the softmax denominator);
//   2. ae_attn_fp8_core: flash-style loop, structure of attention_fast.hip (a lane owns one query column of S^T, K / V^T tiles
//      by LDS-DMA with a source-side XOR swizzle, lazy softmax offset through the MFMA C operand — here the C operand also
//      carries the rel-pos bias: C = rel_w * log2e + rel_h * log2e - m~, one v_add per logit), P = exp2(S') converted with
//      v_cvt_pk_fp8_f32.  The offset keeps max P at 2^7: e4m3 then resolves 16 binades below the row maximum.
#include "common.hpp"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr float F8_LOG2E = 1.4426950408889634f;
constexpr float F8_MAX = 448.0f;
constexpr int F8_DP = 128;       // padded head_dim of Q8 / K8 rows (bytes)
constexpr int F8_DV = 96;        // rows of Vt8 (head_dim 80 + denominator row + padding)
constexpr float F8_PTOP = 7.0f;  // softmax offset target: max S' after a rebase (P' <= 2^7)
constexpr float F8_THR = 8.0f;   // rebase when some S' exceeds this (P' <= 2^8 < 448)

struct Fp8Args {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int B, H, Nq, Nk, D, NkP;  // NkP = Nk rounded up to 64
    long q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, o_sb, o_sh, o_sn;
    float scale;
    const float* rel_h; const float* rel_w; int kH, kW;
    float* amax;         // [BH][4]: amax q, k, v
    unsigned char* q8; unsigned char* k8; unsigned char* vt8;
};

// ---------------------------------------------------------------------------------------------- pass 1: amax per (batch, head)
__global__ __launch_bounds__(256) void fp8_amax_kernel(const Fp8Args p) {
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int which = blockIdx.z;  // 0 q, 1 k, 2 v
    const bf16_t* base = which == 0 ? p.q + (long)b * p.q_sb + (long)h * p.q_sh : which == 1 ? p.k + (long)b * p.k_sb + (long)h * p.k_sh
                                                                                              : p.v + (long)b * p.v_sb + (long)h * p.v_sh;
    const long sn = which == 0 ? p.q_sn : which == 1 ? p.k_sn : p.v_sn;
    const int N = which == 0 ? p.Nq : p.Nk;
    const int CH = p.D / 8;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)N * CH; i += (long)gridDim.x * 256) {
        const int n = (int)(i / CH), c = (int)(i - (long)n * CH);
        const u32x4 t = *reinterpret_cast<const u32x4*>(base + (long)n * sn + c * 8);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(bf16lo(t.x)), fabsf(bf16hi(t.x))), fmaxf(fabsf(bf16lo(t.y)), fabsf(bf16hi(t.y)))));
        m = fmaxf(m, fmaxf(fmaxf(fabsf(bf16lo(t.z)), fabsf(bf16hi(t.z))), fmaxf(fabsf(bf16lo(t.w)), fabsf(bf16hi(t.w)))));
    }
    m = wave_reduce_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(p.amax + bh * 4 + which), __float_as_uint(m));  // m >= 0: bit order = value order
}

__device__ __forceinline__ float pow2_ceil(float x) {  // smallest power of two >= x (x > 0)
    const unsigned u = __float_as_uint(x);
    return __uint_as_float(((u & 0x007FFFFFu) ? (u + 0x00800000u) : u) & 0x7F800000u);
}
struct Fp8Scales { float qmul, kmul, vmul; int e_q; float v_pow2; };
__device__ __forceinline__ Fp8Scales fp8_scales(const float* amax, float scale) {
    const float aq = fmaxf(amax[0], 1e-20f), ak = fmaxf(amax[1], 1e-20f), av = fmaxf(amax[2], 1e-20f);
    Fp8Scales s;
    const float sk = ak / F8_MAX;                       // K8 = k / sk
    const float qprime = aq * scale * F8_LOG2E * sk;    // amax of q * (scale log2e sk)
    const float sig = pow2_ceil(qprime / F8_MAX);       // Q8 = q' / 2^e_q
    s.kmul = 1.0f / sk;
    s.qmul = scale * F8_LOG2E * sk / sig;
    s.e_q = (int)((__float_as_uint(sig) >> 23) & 0xFF) - 127;
    s.v_pow2 = pow2_ceil(av / F8_MAX);
    s.vmul = 1.0f / s.v_pow2;
    return s;
}

// position of key `k` (0..63 inside its tile) in the contraction order of the PV MFMA: the S^T accumulator leaves lane half hi with
// keys 32 b2 + (r & 3) + 8 (r >> 2) + 4 hi as byte 16 b2 + r of its 32-byte operand
__device__ __forceinline__ int f8_kpos(int k) {
    const int b2 = k >> 5, kk = k & 31;
    const int hi = (kk >> 2) & 1, r = (kk & 3) + 4 * (kk >> 3);
    return 32 * hi + 16 * b2 + r;
}

// ---------------------------------------------------------------------------------------------- pass 2: quantise + lay out
// grid (tiles of 64 rows, BH, 2): z = 0 writes Q8 rows, z = 1 writes K8 rows and the Vt8 tile
__global__ __launch_bounds__(256) void fp8_quant_kernel(const Fp8Args p) {
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int n0 = blockIdx.x * 64, tid = threadIdx.x;
    const Fp8Scales sc = fp8_scales(p.amax + bh * 4, p.scale);
    const int CH = p.D / 8;  // 16-byte bf16 chunks per row
    if (blockIdx.z == 0) {
        if (n0 >= p.Nq) return;
        const bf16_t* qb = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
        unsigned char* dst = p.q8 + ((long)bh * p.Nq + n0) * F8_DP;
        for (int i = tid; i < 64 * (F8_DP / 8); i += 256) {  // 8 output bytes per item
            const int r = i / (F8_DP / 8), c = i % (F8_DP / 8);
            if (n0 + r >= p.Nq) continue;
            u32x2 w = {0u, 0u};
            if (c < CH) {
                const u32x4 t = *reinterpret_cast<const u32x4*>(qb + (long)(n0 + r) * p.q_sn + c * 8);
                int lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.x) * sc.qmul, bf16hi(t.x) * sc.qmul, 0, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.y) * sc.qmul, bf16hi(t.y) * sc.qmul, lo, true);
                int hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.z) * sc.qmul, bf16hi(t.z) * sc.qmul, 0, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.w) * sc.qmul, bf16hi(t.w) * sc.qmul, hi, true);
                w = (u32x2){(unsigned)lo, (unsigned)hi};
            }
            *reinterpret_cast<u32x2*>(dst + r * F8_DP + c * 8) = w;
        }
        return;
    }
    if (n0 >= p.NkP) return;
    const bf16_t* kb = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
    const bf16_t* vb = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
    unsigned char* dk = p.k8 + ((long)bh * p.NkP + n0) * F8_DP;
    for (int i = tid; i < 64 * (F8_DP / 8); i += 256) {
        const int r = i / (F8_DP / 8), c = i % (F8_DP / 8);
        u32x2 w = {0u, 0u};
        if (c < CH && n0 + r < p.Nk) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(kb + (long)(n0 + r) * p.k_sn + c * 8);
            int lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.x) * sc.kmul, bf16hi(t.x) * sc.kmul, 0, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.y) * sc.kmul, bf16hi(t.y) * sc.kmul, lo, true);
            int hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.z) * sc.kmul, bf16hi(t.z) * sc.kmul, 0, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf16lo(t.w) * sc.kmul, bf16hi(t.w) * sc.kmul, hi, true);
            w = (u32x2){(unsigned)lo, (unsigned)hi};
        }
        *reinterpret_cast<u32x2*>(dk + r * F8_DP + c * 8) = w;
    }
    // V^T tile through LDS: sv[d][pos(key)] fp8
    __shared__ unsigned char sv[F8_DV * 64];
    for (int i = tid; i < F8_DV * 64 / 4; i += 256) reinterpret_cast<unsigned*>(sv)[i] = 0u;
    __syncthreads();
    for (int i = tid; i < 64 * CH; i += 256) {
        const int r = i / CH, c = i % CH;
        if (n0 + r >= p.Nk) continue;
        const u32x4 t = *reinterpret_cast<const u32x4*>(vb + (long)(n0 + r) * p.v_sn + c * 8);
        const float f[8] = {bf16lo(t.x), bf16hi(t.x), bf16lo(t.y), bf16hi(t.y), bf16lo(t.z), bf16hi(t.z), bf16lo(t.w), bf16hi(t.w)};
        const int pos = f8_kpos(r);
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[(c * 8 + e) * 64 + pos] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(f[e] * sc.vmul, 0.f, 0, false) & 0xFF);
    }
    if (tid < 64 && n0 + tid < p.Nk) sv[p.D * 64 + f8_kpos(tid)] = 0x38;  // 1.0: denominator row
    __syncthreads();
    unsigned char* dv = p.vt8 + (long)bh * F8_DV * p.NkP + n0;
    for (int i = tid; i < F8_DV * 4; i += 256) {  // 16 bytes per item
        const int d = i >> 2, c = i & 3;
        *reinterpret_cast<u32x4*>(dv + (long)d * p.NkP + c * 16) = *reinterpret_cast<const u32x4*>(sv + d * 64 + c * 16);
    }
}

// ---------------------------------------------------------------------------------------------- pass 3: attention
__device__ __forceinline__ i32x8 cat16(u32x4 a, u32x4 b) {
    i32x8 r;
    r[0] = (int)a.x; r[1] = (int)a.y; r[2] = (int)a.z; r[3] = (int)a.w; r[4] = (int)b.x; r[5] = (int)b.y; r[6] = (int)b.z; r[7] = (int)b.w;
    return r;
}
__device__ __forceinline__ void f8_wait_and_barrier() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// BIAS: 0 none; 2 decomposed rel-pos bias with one key-grid ROW per 64-key tile (kW == 64: SAM's global attention on 64 x 64 tokens)
template <int BIAS>
__global__ __launch_bounds__(256, 2) void fp8_attn_kernel(const Fp8Args p) {
    constexpr int KTB = 64 * F8_DP;   // K tile bytes (8 KiB)
    constexpr int VTB = F8_DV * 64;   // V^T tile bytes (6 KiB)
    constexpr int BUFB = KTB + VTB;
    __shared__ __attribute__((aligned(16))) char smem[2 * BUFB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int nqb = (p.Nq + 127) / 128;
    const int vb = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const int bh = vb / nqb, qb = vb - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * 128 + wave * 32;
    const Fp8Scales sc = fp8_scales(p.amax + bh * 4, p.scale);
    const int scale_q = 127 + sc.e_q;  // E8M0

    // ---- Q8 operand (B of S^T = K Q^T): lane (q = l31, hi) holds Q8[q][64 s + 32 hi .. +32]
    i32x8 qf[2];
    {
        const int qrow = min(q0 + l31, p.Nq - 1);
        const unsigned char* qp = p.q8 + ((long)bh * p.Nq + qrow) * F8_DP + 32 * hi;
#pragma unroll
        for (int s = 0; s < 2; ++s) qf[s] = cat16(*reinterpret_cast<const u32x4*>(qp + 64 * s), *reinterpret_cast<const u32x4*>(qp + 64 * s + 16));
    }
    // ---- rel-pos bias (log2 units): this lane's 32 key columns of a tile; rel_h is one value per tile
    f32x16 wb[2];
    if (BIAS == 2) {
        const float* row = p.rel_w + ((long)bh * p.Nq + min(q0 + l31, p.Nq - 1)) * p.kW;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(row + 32 * b2 + 8 * r4 + 4 * hi);
                wb[b2][4 * r4] = t[0] * F8_LOG2E; wb[b2][4 * r4 + 1] = t[1] * F8_LOG2E; wb[b2][4 * r4 + 2] = t[2] * F8_LOG2E; wb[b2][4 * r4 + 3] = t[3] * F8_LOG2E;
            }
    }
    const float* rh_row = BIAS == 2 ? p.rel_h + ((long)bh * p.Nq + min(q0 + l31, p.Nq - 1)) * p.kH : nullptr;

    // ---- LDS-DMA plan: K tile = 8 pieces, V^T tile = 6 pieces of 1 KiB; piece j is issued by wave j % 4.  Both images are swizzled on
    // the SOURCE side (16-byte piece position ^ row bits) so that the 32-byte operand reads below are bank-conflict free.
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(p.k8 + (long)bh * p.NkP * F8_DP, 0, p.NkP * F8_DP, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(p.vt8 + (long)bh * F8_DV * p.NkP, 0, F8_DV * p.NkP, 0x00020000);
    int koff[2], voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int o = (wave + 4 * i) * 1024 + lane * 16;  // K image: [64 keys][8 pieces]
        const int key = o >> 7, pp = (o >> 4) & 7;
        koff[i] = key * F8_DP + ((pp ^ ((key >> 1) & 7)) << 4);
        const int o2 = (wave + 4 * i) * 1024 + lane * 16;  // V^T image: [96 d][4 pieces]
        const int d = o2 >> 6, p4 = (o2 >> 4) & 3;
        voff[i] = d * p.NkP + ((p4 ^ ((d >> 2) & 3)) << 4);
    }
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    auto issue = [&](int t) {
        const int boff = lds0 + (t & 1) * BUFB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave + 4 * i;
            ae_dma16(rsK, boff + j * 1024, koff[i], t * 64 * F8_DP);
            if (j < 6) ae_dma16(rsV, boff + KTB + j * 1024, voff[i], t * 64);
        }
    };
    // operand read addresses: K (A of S^T): lane (key = l31 + 32 b2, hi): pieces 4 s + 2 hi, + 1 of its row;  V^T (A of O^T): lane
    // (d = l31 + 32 db, hi): pieces 2 hi, + 1
    int kad[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 2; ++e) kad[s][e] = l31 * F8_DP + (((4 * s + 2 * hi + e) ^ ((l31 >> 1) & 7)) << 4);
    int vad[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) vad[e] = KTB + l31 * 64 + (((2 * hi + e) ^ ((l31 >> 2) & 3)) << 4);

    f32x16 o[3];
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float mt = 0.f;

    const int ntiles = p.NkP / 64;
    issue(0);
    for (int t = 0; t < ntiles; ++t) {
        f8_wait_and_barrier();
        if (t + 1 < ntiles) issue(t + 1);
        const char* buf = smem + (t & 1) * BUFB;
        const bool tail = (t + 1) * 64 > p.Nk;
        float hb = -mt;
        if (BIAS == 2) hb += rh_row[t] * F8_LOG2E;
        // ---- S'^T = K8 Q8^T 2^e_q + bias - m~ : lane holds S'[key = 32 b2 + (r&3) + 8 (r>>2) + 4 hi][q = l31]
        f32x16 s[2];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[b2][r] = BIAS == 2 ? wb[b2][r] + hb : hb;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* kp = buf + b2 * 32 * F8_DP;
                const i32x8 kf = cat16(*reinterpret_cast<const u32x4*>(kp + kad[ks][0]), *reinterpret_cast<const u32x4*>(kp + kad[ks][1]));
                s[b2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[ks], s[b2], 0, 0, 0, 127, 0, scale_q);
            }
        }
        if (tail) {
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (t * 64 + 32 * b2 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Nk) s[b2][r] = -1.0e30f;
        }
        float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
        if (__builtin_expect(t == 0 || __any(mx > F8_THR), 0)) {
            // rebase the offset (rare): rows whose tile maximum is above F8_PTOP move m~ so that it lands there (integer steps)
            const float m2 = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float d = (t == 0 || m2 > F8_PTOP) ? __builtin_ceilf(m2 - F8_PTOP) : 0.f;
            d = fmaxf(d, -1.0e4f);
            mt += d;
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[b2][r] -= d;
#pragma unroll
            for (int db = 0; db < 3; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }
        // ---- P' = exp2(S') in e4m3: byte 16 b2 + r of the lane's 32-byte B operand
        i32x8 pf;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_exp2f(s[b2][4 * r4]), __builtin_amdgcn_exp2f(s[b2][4 * r4 + 1]), 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_exp2f(s[b2][4 * r4 + 2]), __builtin_amdgcn_exp2f(s[b2][4 * r4 + 3]), w, true);
                pf[4 * b2 + r4] = w;
            }
        // ---- O^T += Vt8 P'^T : lane holds O^T[d = 32 db + (r&3) + 8 (r>>2) + 4 hi][q = l31]
#pragma unroll
        for (int db = 0; db < 3; ++db) {
            const char* vp = buf + db * 32 * 64;
            const i32x8 vf = cat16(*reinterpret_cast<const u32x4*>(vp + vad[0]), *reinterpret_cast<const u32x4*>(vp + vad[1]));
            o[db] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o[db], 0, 0, 0, 127, 0, 127);
        }
    }

    // ---- normalise (row D of O^T is sum_k P'), undo the V scale, store 4 consecutive d per lane
    const int LDB = p.D / 32, LREG = 4 * ((p.D % 32) / 8);
    float lraw = 0.f;
#pragma unroll
    for (int db = 0; db < 3; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (db == LDB && r == LREG) lraw = o[db][r];
    const float lsum = __shfl(lraw, l31, 64);
    const float inv = sc.v_pow2 / lsum;
    const int qrow = q0 + l31;
    if (qrow < p.Nq) {
        bf16_t* op = p.o + (long)b * p.o_sb + (long)h * p.o_sh + (long)qrow * p.o_sn;
#pragma unroll
        for (int db = 0; db < 3; ++db)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int d = 32 * db + 8 * r4 + 4 * hi;
                if (d < p.D)
                    *reinterpret_cast<u32x2*>(op + d) = (u32x2){pack_bf16x2(o[db][4 * r4] * inv, o[db][4 * r4 + 1] * inv), pack_bf16x2(o[db][4 * r4 + 2] * inv, o[db][4 * r4 + 3] * inv)};
            }
    }
}

}  // namespace

// bytes of device workspace ae_attn_fwd_fp8 needs for a problem (0 if unsupported)
extern "C" long ae_attn_fp8_workspace_bytes(int B, int H, int Nq, int Nk, int D) {
    if (D % 8 != 0 || D > 88 || B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return 0;
    const long BH = (long)B * H, NkP = (Nk + 63) / 64 * 64;
    return 256 + BH * 16 + BH * Nq * F8_DP + BH * NkP * F8_DP + BH * F8_DV * NkP + 256;
}

extern "C" int ae_attn_fwd_fp8(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D, long q_sb, long q_sh,
                               long q_sn, long k_sb, long k_sh, long k_sn, long v_sb, long v_sh, long v_sn, long o_sb, long o_sh, long o_sn,
                               float scale, const float* rel_h, const float* rel_w, int kH, int kW, void* workspace, long workspace_bytes,
                               void* stream) {
    AE_REQUIRE(q && k && v && out && workspace, "ae_attn_fwd_fp8: null pointer");
    const long need = ae_attn_fp8_workspace_bytes(B, H, Nq, Nk, D);
    AE_REQUIRE(need > 0, "ae_attn_fwd_fp8: unsupported sizes B=%d H=%d Nq=%d Nk=%d D=%d (head_dim % 8 == 0, <= 88)", B, H, Nq, Nk, D);
    AE_REQUIRE(workspace_bytes >= need, "ae_attn_fwd_fp8: workspace too small (%ld < %ld bytes)", workspace_bytes, need);
    AE_REQUIRE((q_sb | q_sh | q_sn | k_sb | k_sh | k_sn | v_sb | v_sh | v_sn) % 8 == 0 && (o_sb | o_sh | o_sn) % 4 == 0, "ae_attn_fwd_fp8: strides must keep rows 16-byte aligned");
    AE_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)workspace & 255) == 0, "ae_attn_fwd_fp8: pointer alignment");
    AE_REQUIRE((rel_h == nullptr) == (rel_w == nullptr), "ae_attn_fwd_fp8: rel_h and rel_w go together");
    if (rel_h) AE_REQUIRE(kW == 64 && (long)kH * kW == Nk, "ae_attn_fwd_fp8: the rel-pos bias path covers key grids with kW == 64 (SAM global attention); got kH=%d kW=%d Nk=%d", kH, kW, Nk);
    Fp8Args a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)out;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.NkP = (Nk + 63) / 64 * 64;
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_sn = q_sn; a.k_sb = k_sb; a.k_sh = k_sh; a.k_sn = k_sn; a.v_sb = v_sb; a.v_sh = v_sh; a.v_sn = v_sn;
    a.o_sb = o_sb; a.o_sh = o_sh; a.o_sn = o_sn; a.scale = scale; a.rel_h = rel_h; a.rel_w = rel_w; a.kH = kH; a.kW = kW;
    const long BH = (long)B * H;
    AE_REQUIRE(BH * a.NkP * F8_DP < (1L << 40) && (long)a.NkP * F8_DP < (1L << 31) && (long)F8_DV * a.NkP < (1L << 31), "ae_attn_fwd_fp8: problem too large");
    char* ws = (char*)workspace;
    a.amax = (float*)ws;
    long off = (BH * 16 + 255) / 256 * 256;
    a.q8 = (unsigned char*)(ws + off); off += BH * Nq * F8_DP;
    a.k8 = (unsigned char*)(ws + off); off += BH * a.NkP * F8_DP;
    a.vt8 = (unsigned char*)(ws + off);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(a.amax, 0, BH * 16, s) != hipSuccess) { ae_set_error("ae_attn_fwd_fp8: memset failed"); return AE_ERR_LAUNCH; }
    const int nmax = Nq > Nk ? Nq : Nk;
    const int gx = (int)(((long)nmax * (D / 8) + 255) / 256) < 64 ? (int)(((long)nmax * (D / 8) + 255) / 256) : 64;
    hipLaunchKernelGGL(fp8_amax_kernel, dim3(gx, (unsigned)BH, 3), dim3(256), 0, s, a);
    const int tiles = ((Nq > a.NkP ? Nq : a.NkP) + 63) / 64;
    hipLaunchKernelGGL(fp8_quant_kernel, dim3(tiles, (unsigned)BH, 2), dim3(256), 0, s, a);
    const unsigned blocks = (unsigned)(((Nq + 127) / 128) * BH);
    if (rel_h) hipLaunchKernelGGL((fp8_attn_kernel<2>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((fp8_attn_kernel<0>), dim3(blocks), dim3(256), 0, s, a);
    return ae_check_launch("ae_attn_fwd_fp8");
}
