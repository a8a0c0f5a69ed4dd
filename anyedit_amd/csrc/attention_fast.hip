// Long-sequence attention forward for gfx950 (MI355X): out = softmax(q k^T * scale) v, no bias / mask / second segment.
//
// Replaces: ldm/modules/attention.py:163-194 CrossAttention.forward (self-attention of the 64x64 / 32x32
// UNet levels: N = 4096 / 1024 tokens, head_dim 40 / 80) and the xformers call at attention.py:222-233.
//
// Why a second kernel: the general kernel (attention.hip) spends 6.7 VALU instructions per MFMA at head_dim 40 and is bound by
// VALU issue, not by the matrix pipe (profiles/r01_pmc_attn_d40_qf4_a.txt; tools/ubench/issue_model2.hip shows what a SIMD can
// overlap).  This one is built around the instruction budget instead:
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16: head_dim 40 is 3 K-steps (48) instead of a K=32 + a K=16 MFMA of the same
//     16 cycles each; a lane owns ONE query column (q = lane & 31) and 16 of the block's 32 keys;
//   * Q is pre-multiplied by scale*log2(e) once, and the running offset m~ of the online softmax enters through the MFMA's C
//     operand (a 16-register splat of -m~ that is rewritten only when m~ moves), so the MFMA result IS the exp2 argument:
//     per logit the VALU work is one v_exp_f32 and half a v_cvt_pk_bf16_f32 — no scale fma, no subtract;
//   * m~ is lazy: it moves only when some logit exceeds it by more than RESCALE_THR (P <= 2^THR is harmless for bf16 P and fp32
//     O); the test is a v_max3 tree + one wave-wide compare, the rebase itself a rarely taken wave-uniform branch.  Softmax is
//     invariant to the offset as long as P and the denominator use the same one — they do, the denominator is accumulated by the
//     PV MFMA itself from a row of ones (below);
//   * P goes from the 32x32 accumulator layout to the B operand of v_mfma_f32_16x16x32_bf16 (O^T = V^T P^T, head_dim padded to 48
//     instead of 64) with four v_permlane16_swap per 32x32 block, no LDS round trip;
//   * K and V tiles (64 keys) are copied by LDS-DMA (buffer_load_dwordx4 ... lds) as compact row-major images — no staging
//     registers, no ds_write, no transposing VALU; V^T operands come from ds_read_b64_tr_b16.  The lanes that would read the
//     padding column head_dim .. head_dim+3 of V read a constant {1,0,0,0} instead: O^T row `head_dim` is sum_k P = the softmax
//     denominator.  K's padding columns need no care: Q is zero there.
#include "attention.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4v;

constexpr int FKT = 64;  // keys per LDS tile
constexpr float FLOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 8.0f;  // log2 units
constexpr float MASKED = -1.0e30f;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* lds_dst, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, 0, 0, 0);
#endif
}

__device__ __forceinline__ s16x4v lds_tr16(const char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)p);
#else
    return s16x4v{};
#endif
}

__device__ __forceinline__ bf16x8_t cat_tr(s16x4v lo, s16x4v hi) {
    union { struct { s16x4v a, b; } s; bf16x8_t v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

template <int D, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_fast_kernel(const AttnArgs p) {
    static_assert(D % 8 == 0 && D <= 96, "head_dim: multiple of 8, <= 96");
    constexpr int KS = (D + 15) / 16;   // K=16 steps of S^T = K Q^T
    constexpr int NDF = D / 16 + 1;     // 16-row fragments of O^T; row D is the softmax denominator
    constexpr int ROWB = 2 * D;         // bytes per K / V row in LDS
    constexpr int CH = D / 8;           // 16-byte chunks per row = 1-KiB DMA pieces per 64-row tile
    constexpr int TILEB = FKT * ROWB;
    constexpr int BUFB = 2 * TILEB;     // K tile then V tile
    constexpr int CONST_OFF = 2 * BUFB; // {1,0,0,0} bf16, then zeros
    constexpr int LDSB = CONST_OFF + 64;
    constexpr int NPIECE = 2 * CH;
    constexpr int MAXP = (NPIECE + 3) / 4;
    constexpr int LDF = D / 16, LROW = D % 16;  // O^T fragment / row that carries the denominator

    __shared__ __attribute__((aligned(16))) char smem[LDSB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, g = lane >> 4;
    constexpr int QB = 128;
    const int nqb = (p.Nq + QB - 1) / QB;
    const int vb = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const int bh = vb / nqb, qb = vb - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * QB + wave * 32;

    if (tid < 16) reinterpret_cast<uint32_t*>(smem + CONST_OFF)[tid] = tid == 0 ? 0x00003F80u : 0u;

    // ---- Q^T operand of the 32x32x16 MFMA: lane (q = l31, hi) holds c * Q[q][16 ks + 8 hi .. +8], zero beyond head_dim
    const bf16_t* qp = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
    const float c = p.scale * FLOG2E;
    bf16x8_t qf[KS];
    {
        const int qrow = min(q0 + l31, p.Nq - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 16 * ks + 8 * hi;
            u32x4 t = *reinterpret_cast<const u32x4*>(qp + (long)qrow * p.q_sn + (d0 < D ? d0 : 0));
            if (d0 >= D) t = (u32x4){0u, 0u, 0u, 0u};
            u32x4 w;
            w.x = pack_bf16x2(bf16lo(t.x) * c, bf16hi(t.x) * c);
            w.y = pack_bf16x2(bf16lo(t.y) * c, bf16hi(t.y) * c);
            w.z = pack_bf16x2(bf16lo(t.z) * c, bf16hi(t.z) * c);
            w.w = pack_bf16x2(bf16lo(t.w) * c, bf16hi(t.w) * c);
            qf[ks] = as_bf16x8(w);
        }
    }

    // ---- LDS-DMA plan: piece j of a tile (K pieces 0..CH-1, V pieces CH..2CH-1) is issued by wave j % 4
    const bf16_t* kp = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
    const bf16_t* vp = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kp), 0, (int)(((long)(p.Nk - 1) * p.k_sn + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vp), 0, (int)(((long)(p.Nk - 1) * p.v_sn + D) * 2), 0x00020000);
    const int ksn2 = (int)p.k_sn * 2, vsn2 = (int)p.v_sn * 2;
    int voff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int j = wave + 4 * i;
        const bool isK = j < CH;
        const int cidx = (isK ? j : j - CH) * 64 + lane;
        const int row = cidx / CH, cc = cidx - row * CH;
        voff[i] = row * (isK ? ksn2 : vsn2) + cc * 16;
    }
    auto issue = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + 4 * i;
            if (j < NPIECE) {
                if (j < CH) dma16(rsK, smem + buf * BUFB + j * 1024, voff[i] + t * FKT * ksn2);
                else dma16(rsV, smem + buf * BUFB + TILEB + (j - CH) * 1024, voff[i] + t * FKT * vsn2);
            }
        }
    };

    // ---- per-lane LDS read addresses
    const int kaddr = l31 * ROWB + hi * 16;                                   // + ks * 32 + block / buffer offsets (immediates)
    const int vkey = 16 * (g & 1) + 4 * (g >> 1) + (l15 >> 2);                // key row this lane addresses in a tr read (+8 for the second)
    const int vaddr = vkey * ROWB + (l15 & 3) * 8;                            // + df * 32 + block / buffer offsets
    const bool ones_lane = (16 * LDF + 4 * (l15 & 3)) == D;                   // supplies columns D..D+3 of the last fragment

    f32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    float mt = 0.f;  // m~ (log2 units): S' = c q.k - m~
    f32x4 o[2][NDF];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int df = 0; df < NDF; ++df) o[a][df] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntiles = (p.Nk + FKT - 1) / FKT;

    // one 32-key block of the tile in buffer BUF
    auto block = [&](auto buf_tag, auto blk_tag, int k0, bool first, bool tail) {
        constexpr int BUF = decltype(buf_tag)::value, B2 = decltype(blk_tag)::value;
        constexpr int KOFF = BUF * BUFB + B2 * 32 * ROWB;
        constexpr int VOFF = BUF * BUFB + TILEB + B2 * 32 * ROWB;
        // ---- S'^T = K (cQ)^T - m~ : lane holds S'[key = (r&3) + 8 (r>>2) + 4 hi][q = l31]
        f32x16 s;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8_t kf = as_bf16x8(*reinterpret_cast<const u32x4*>(smem + kaddr + KOFF + ks * 32));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], ks == 0 ? cinit : s, 0, 0, 0);
        }
        if (tail) {  // keys past Nk (zero rows from the bounds-checked DMA) must not count
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + B2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Nk) s[r] = MASKED;
        }
        float mx = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[r]), s[r + 1]);
        mx = fmaxf(mx, s[15]);
        if (__builtin_expect(first || __any(mx > RESCALE_THR), 0)) {
            // rebase m~ (rare): rows whose block maximum is above the offset move it up to that maximum (integer steps: the
            // factor 2^-d is exact); the first block sets it whatever its sign.  O (with its denominator row) follows.
            const float m2 = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float d = (first || m2 > 0.f) ? __builtin_ceilf(m2) : 0.f;
            d = fmaxf(d, -1.0e4f);  // a fully masked first block leaves the offset finite
            mt += d;
#pragma unroll
            for (int r = 0; r < 16; ++r) { cinit[r] = -mt; s[r] -= d; }
            const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float aq = __shfl(alpha, a * 16 + l15, 64);
#pragma unroll
                for (int df = 0; df < NDF; ++df) { o[a][df][0] *= aq; o[a][df][1] *= aq; o[a][df][2] *= aq; o[a][df][3] *= aq; }
            }
        }
        // ---- P = exp2(S'), bf16, and into the B-operand layout of the 16x16x32 MFMA (both 16-query fragments)
        uint32_t pk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pk[j] = pack_bf16x2(__builtin_amdgcn_exp2f(s[2 * j]), __builtin_amdgcn_exp2f(s[2 * j + 1]));
        u32x4 w0, w1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x2 sw = __builtin_amdgcn_permlane16_swap(pk[j], pk[j + 4], false, false);
            w0[j] = sw[0];
            w1[j] = sw[1];
        }
        const bf16x8_t pb0 = as_bf16x8(w0), pb1 = as_bf16x8(w1);
        // ---- O^T += V^T P^T : lane holds O^T[d = 16 df + 4 g + r][q = 16 a + l15]
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
            int va = vaddr + VOFF + df * 32, vb2 = va + 8 * ROWB;
            if (df == LDF) {
                va = ones_lane ? CONST_OFF : va;
                vb2 = ones_lane ? CONST_OFF : vb2;
            }
            const bf16x8_t vf = cat_tr(lds_tr16(smem + va), lds_tr16(smem + vb2));
            o[0][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb0, o[0][df], 0, 0, 0);
            o[1][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb1, o[1][df], 0, 0, 0);
        }
    };

    auto tile = [&](auto buf_tag, int t) {
        constexpr int BUF = decltype(buf_tag)::value;
        __syncthreads();  // (drains this wave's DMA first) tile t has landed; every wave is done with tile t - 1
        if (t + 1 < ntiles) issue(t + 1, BUF ^ 1);
        const bool tail = (t + 1) * FKT > p.Nk;
        block(buf_tag, std::integral_constant<int, 0>{}, t * FKT, t == 0, tail);
        block(buf_tag, std::integral_constant<int, 1>{}, t * FKT, false, tail);
    };

    issue(0, 0);
    for (int t = 0; t < ntiles; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    }

    // ---- normalise and store: 4 consecutive d per lane -> 8-byte stores
    bf16_t* op = p.o + (long)b * p.o_sb + (long)h * p.o_sh;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float lsum = __shfl(o[a][LDF][LROW % 4], l15 + 16 * (LROW / 4), 64);
        const float inv = (p.out_scale ? p.out_scale[b] : 1.0f) / lsum;
        const int qrow = q0 + a * 16 + l15;
        if (qrow >= p.Nq) continue;
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
            const int d = df * 16 + g * 4;
            if (d < D) {
                float r0 = o[a][df][0] * inv, r1 = o[a][df][1] * inv, r2 = o[a][df][2] * inv, r3 = o[a][df][3] * inv;
                u32x2* dst = reinterpret_cast<u32x2*>(op + (long)qrow * p.o_sn + d);
                if (p.accum) {
                    const u32x2 prev = *dst;
                    r0 += bf16lo(prev.x); r1 += bf16hi(prev.x); r2 += bf16lo(prev.y); r3 += bf16hi(prev.y);
                }
                *dst = (u32x2){pack_bf16x2(r0, r1), pack_bf16x2(r2, r3)};
            }
        }
    }
}

template <int D>
int launch_fast(const AttnArgs& a, hipStream_t stream) {
    static const int occ = getenv("AE_ATTN_FAST_OCC") ? atoi(getenv("AE_ATTN_FAST_OCC")) : 3;
    const long blocks = (long)((a.Nq + 127) / 128) * a.B * a.H;
    dim3 grid((unsigned)blocks), block(256);
    if (occ == 2) hipLaunchKernelGGL((attn_fast_kernel<D, 2>), grid, block, 0, stream, a);
    else if (occ == 4) hipLaunchKernelGGL((attn_fast_kernel<D, 4>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_fast_kernel<D, 3>), grid, block, 0, stream, a);
    return ae_check_launch("ae_attn_fwd_bf16(fast)");
}

}  // namespace

int ae_attn_fast_launch(const AttnArgs& a, int D, hipStream_t stream) {
    if (a.rel_h || a.key_mask || a.k2 || a.lse || a.lse2) return AE_ERR_UNSUPPORTED;
    // 32-bit byte offsets inside one (batch, head) image of K / V
    if (((long)a.Nk * a.k_sn + D) * 2 >= (1L << 31) || ((long)a.Nk * a.v_sn + D) * 2 >= (1L << 31)) return AE_ERR_UNSUPPORTED;
    switch (D) {
        case 40: return launch_fast<40>(a, stream);
        case 80: return launch_fast<80>(a, stream);
        default: return AE_ERR_UNSUPPORTED;
    }
}
