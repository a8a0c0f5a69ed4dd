// Fused cross-attention half of a BasicTransformerBlock at the 64x64 UNet level (C = 320, 8 heads of 40) for gfx950 / MI355X:
//
//   Y[M,320] = ( softmax(q K^T) V  +  gate_b softmax(q K_ip^T) V_ip ) Wo^T + bo + X,     q = LayerNorm(X) Wq^T,   per head, scale 40^-1/2
//
// Replaces: `x = self.attn2(self.norm2(x), context=context) + x` of BasicTransformerBlock._forward (ldm/modules/attention.py:273), i.e.
// CrossAttention.forward (:163-194: to_q, the softmax(QK^T)V core over the 77 (+1 task) text keys, to_out) behind norm2, with AnySD's decoupled expert segment
// (shape template other_modules/ip_adapter/attention_processor.py:141-173) — until round 6 three launches (LayerNorm-fold q projection, short-K/V
// attention, to_out + residual + row statistics: 82 us at UNet batch 12, 220 MB moved for 63 MB of real traffic).  Here q, the logits, the probabilities and
// the attention output never leave the registers of the wave that owns their rows:
//   * one block = 128 rows of ONE sample (4096 % 128 == 0: a block never straddles two samples' keys), four waves (one per SIMD), 32 rows each: the normalised X
//     fragments (80 VGPRs) and the fp32 output accumulators [32 x 320] (160 registers, accumulator file) stay for the whole launch;
//   * per head: Q (q_h^T = Wq_h x^T, 60 MFMAs) -> the result registers, rounded and scaled, ARE the B operand of the logit product S^T = K_h q_h^T (the k order of an
//     MFMA contraction is free: K's LDS image is packed in the order the Q results come in) -> softmax of a row's 78 (+4) logits over its four lanes -> the
//     probabilities ARE the B operand of O^T = V_h^T P^T (V^T's image in the logits' order; its padding row of ones yields the softmax denominator) -> the
//     normalised, gate-combined output of two heads IS the B operand of three K steps of the output projection (Wo's image in that order);
//   * everything streamed — Wq_h images, the sample's K_h / V_h^T images (step-invariant: packed once per edit, ops.pack_xattn_kv), Wo images — goes through
//     a four-slot LDS ring by LDS-DMA with counted waits, one barrier per item; every image is stored ready-made in global memory (swizzles and paddings
//     included: linear copies), row strides chosen so that every ds_read_b128 lane group covers sixteen different bank slots;
//   * the MFMA stream is placed by hand (csrc/handplaced.hpp), like csrc/ff_fused.hip.
#include "common.hpp"
#include "handplaced.hpp"
#include <stdlib.h>

namespace {

struct XAArgs {
    const bf16_t* X; bf16_t* Y;
    const bf16_t* Wq;    // [8 heads][48 rows][320] bf16 as LDS images (XOR-swizzled rows of 640 B; rows 40 .. 47 zero)
    const bf16_t* KV;    // [B][8][XA_KVB bytes]: K image [96 keys][160 B] then V^T image [48 d][288 B] (+ pad)
    const bf16_t* Wo;    // [4 head pairs][2 halves][160 rows][224 B]
    const float* bo; const float* ln_g; const float* ln_b; const float* gate;   // gate: [B] or NULL
    float ln_eps, qscale;   // qscale = head_dim^-1/2 * log2(e)
    int M, rows_per_sample, Nk, T;
    long ldx, ldy;
};

constexpr int XA_K = 320, XA_KS = 10, XA_ROWB = 640, XA_MF = 2, XA_BM = 128, XA_NW = 4, XA_H = 8, XA_NCF = 20;
constexpr int XA_NSLOT = 4, XA_SLOTB = 36 * 1024, XA_PPW = 9;          // item: up to 36 pieces of 1 KiB, nine per wave
constexpr int XA_WQB = 48 * XA_ROWB;                                   // 30 KiB: Wq_h image
constexpr int XA_KSTR = 160, XA_KIMGB = 96 * XA_KSTR;                  // K image: 96 key rows (80 text, 16 expert) x (2 K steps x 64 B + 32 B pad)
constexpr int XA_VSTR = 288, XA_VIMGB = 14 * 1024;                     // V^T image: 48 d rows x (4 K steps x 64 B + 32 B pad) = 13 824 B, padded to whole pieces
constexpr int XA_KVB = XA_KIMGB + XA_VIMGB;                            // 29 pieces
constexpr int XA_OSTR = 224, XA_WOHB = 160 * XA_OSTR;                  // Wo half image: 160 rows x (3 K steps x 64 B + 32 B pad) = 35 pieces
#ifndef XA_LAB
#define XA_LAB 0   // lab builds only (timing ablations, wrong results): 1 no DMA in the loop, 2 no softmax arithmetic, 4 no q / o packing arithmetic, 8 no barrier / DMA wait
#endif
constexpr int XA_OOB = 0x40000000;
constexpr int XA_RING = 8, XA_LEAD = 5;


// maximum of a and of b over the four lanes l15 + 16 g of a row, in every one of them: v_permlane16_swap / v_permlane32_swap (gfx950) exchange the odd rows of 16 lanes
// of the first operand with the even rows of the second / the upper half of the first with the lower half of the second — two VALU operations instead of two
// ds_bpermute round trips per reduction (eight dependent LDS round trips per head in the first form: profiles/r06_xattn_fused_notes.txt).  (hipcc pads nothing inside asm:
// the s_nop covers the VALU write -> lane-swap read wait states.)
__device__ __forceinline__ void xa_rowmax4(float& a, float& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    {   // lanes 16 .. 31 <-> 0 .. 15 and 48 .. 63 <-> 32 .. 47: pair (g, g ^ 1)
        float a2 = a, b2 = b;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(a2));
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(b), "+v"(b2));
        a = fmaxf(a, a2); b = fmaxf(b, b2);
    }
    {   // lanes 32 .. 63 <-> 0 .. 31: pair (g, g ^ 2)
        float a2 = a, b2 = b;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(a2));
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(b), "+v"(b2));
        a = fmaxf(a, a2); b = fmaxf(b, b2);
    }
#endif
}

__global__ __launch_bounds__(64 * XA_NW, 1) void xattn_fused_kernel(const XAArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[XA_NSLOT * XA_SLOTB + 3 * XA_K * 4];
    float* const sbo = reinterpret_cast<float*>(smem + XA_NSLOT * XA_SLOTB);
    float* const sln = sbo + XA_K;   // gamma | beta

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * XA_BM + wave * (XA_MF * 16);
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int bs = (blockIdx.x * XA_BM) / p.rows_per_sample;     // the block's sample

    const __amdgpu_buffer_rsrc_t rsWq = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wq), 0, XA_H * XA_WQB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsKV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.KV) + (long)bs * XA_H * (XA_KVB / 2), 0, XA_H * XA_KVB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsWo = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.Wo), 0, XA_H * XA_WOHB, 0x00020000);
    // Item j of the stream (24 per block; pair pp = j / 6, kind r = j % 6): 0 Wq(2 pp), 1 KV(2 pp), 2 Wq(2 pp + 1), 3 KV(2 pp + 1), 4 Wo(pp, half 0), 5 Wo(pp, half 1).  All
    // images are linear copies; a wave's pieces are q = wave + 4 jj, jj < 9; pieces past the image's end (and items past the last) read zeros (out-of-range lane
    // offset).  The kind is a compile-time argument: no branch inside the hand-placed streams.
    auto issue_piece = [&](auto kind_c, int j, int jj) {
        constexpr int R = decltype(kind_c)::value;
        const int slot = lds0 + (j & (XA_NSLOT - 1)) * XA_SLOTB, q = wave + XA_NW * jj;
        const bool live = j < 24;
        const int ppc = live ? j / 6 : 0;
        if constexpr (R == 0 || R == 2)
            ae_dma16(rsWq, slot + q * 1024, lane * 16 + ((live && q < XA_WQB / 1024) ? 0 : XA_OOB), (2 * ppc + (R >> 1)) * XA_WQB + q * 1024);
        else if constexpr (R == 1 || R == 3)
            ae_dma16(rsKV, slot + q * 1024, lane * 16 + ((live && q < XA_KVB / 1024) ? 0 : XA_OOB), (2 * ppc + (R >> 1)) * XA_KVB + q * 1024);
        else
            ae_dma16(rsWo, slot + q * 1024, lane * 16 + ((live && q < XA_WOHB / 1024) ? 0 : XA_OOB), (2 * ppc + (R - 4)) * XA_WOHB + q * 1024);
    };

    // ---- X panel: 128 rows = four 32-row chunk images (XOR-swizzled 640-byte rows, as gemm_rowpanel.hip) at the start of the ring; every wave picks its fragments
    {
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, (int)((long)p.M * p.ldx * 2), 0x00020000);
        const int soff = blockIdx.x * XA_BM * (int)p.ldx * 2;
#pragma unroll
        for (int j = 0; j < 20; ++j) {
            const int q = wave + XA_NW * j;
            const int o = (q % 20) * 1024 + lane * 16;
            const int i = o / XA_ROWB, pp = (o - i * XA_ROWB) >> 4;
            ae_dma16(rsX, lds0 + q * 1024, ((q / 20) * 32 + i) * (int)p.ldx * 2 + ((pp & ~7) | ((pp ^ (i >> 1)) & 7)) * 16, soff);
        }
    }
    for (int i = tid; i < XA_K; i += 64 * XA_NW) { sbo[i] = p.bo ? p.bo[i] : 0.f; sln[i] = p.ln_g[i]; sln[XA_K + i] = p.ln_b[i]; }
    const float gate = p.gate ? p.gate[bs] : 0.f;
    hp_wait_dma<0>();
    __syncthreads();
    u32x4 af[XA_MF][XA_KS];
#pragma unroll
    for (int f = 0; f < XA_MF; ++f) {
        const int rl = wave * (XA_MF * 16) + 16 * f + l15;
        const int i = rl & 31;
        const char* base = smem + (rl >> 5) * (32 * XA_ROWB) + i * XA_ROWB;
#pragma unroll
        for (int ks = 0; ks < XA_KS; ++ks) {
            const int c16 = 4 * ks + g;
            af[f][ks] = *reinterpret_cast<const u32x4*>(base + (((c16 & ~7) | ((c16 ^ (i >> 1)) & 7)) << 4));
        }
    }
    __syncthreads();  // the ring now belongs to the streamed images
#pragma unroll
    for (int jj = 0; jj < XA_PPW; ++jj) issue_piece(std::integral_constant<int, 0>{}, 0, jj);
#pragma unroll
    for (int jj = 0; jj < XA_PPW; ++jj) issue_piece(std::integral_constant<int, 1>{}, 1, jj);
#pragma unroll
    for (int jj = 0; jj < XA_PPW; ++jj) issue_piece(std::integral_constant<int, 2>{}, 2, jj);

    // LayerNorm on the registers (as csrc/ff_fused.hip: every wave owns its rows alone), under the first items' DMA
    {
        float mean[XA_MF], rstd[XA_MF];
#pragma unroll
        for (int f = 0; f < XA_MF; ++f) {
            float sm = 0.f;
#pragma unroll
            for (int ks = 0; ks < XA_KS; ++ks) {
                const u32x4 t = af[f][ks];
                sm += (bf16lo(t.x) + bf16hi(t.x)) + (bf16lo(t.y) + bf16hi(t.y)) + (bf16lo(t.z) + bf16hi(t.z)) + (bf16lo(t.w) + bf16hi(t.w));
            }
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            const float mu = sm * (1.0f / XA_K);
#pragma unroll
            for (int ks = 0; ks < XA_KS; ++ks) asm volatile("" : "+v"(af[f][ks]));
            float v = 0.f;
#pragma unroll
            for (int ks = 0; ks < XA_KS; ++ks) {
                const u32x4 t = af[f][ks];
                const float d0 = bf16lo(t.x) - mu, d1 = bf16hi(t.x) - mu, d2 = bf16lo(t.y) - mu, d3 = bf16hi(t.y) - mu;
                const float d4 = bf16lo(t.z) - mu, d5 = bf16hi(t.z) - mu, d6 = bf16lo(t.w) - mu, d7 = bf16hi(t.w) - mu;
                v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
            }
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            mean[f] = mu;
            rstd[f] = __builtin_amdgcn_rsqf(v * (1.0f / XA_K) + p.ln_eps);
#pragma unroll
            for (int ks = 0; ks < XA_KS; ++ks) asm volatile("" : "+v"(af[f][ks]));
        }
#pragma unroll
        for (int ks = 0; ks < XA_KS; ++ks) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sln + 32 * ks + 8 * g), g1 = *reinterpret_cast<const f32x4*>(sln + 32 * ks + 8 * g + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sln + XA_K + 32 * ks + 8 * g), b1 = *reinterpret_cast<const f32x4*>(sln + XA_K + 32 * ks + 8 * g + 4);
#pragma unroll
            for (int f = 0; f < XA_MF; ++f) {
                const u32x4 t = af[f][ks];
                const float mu = mean[f], rs = rstd[f];
                u32x4 w;
                w.x = pack_bf16x2((bf16lo(t.x) - mu) * rs * g0[0] + b0[0], (bf16hi(t.x) - mu) * rs * g0[1] + b0[1]);
                w.y = pack_bf16x2((bf16lo(t.y) - mu) * rs * g0[2] + b0[2], (bf16hi(t.y) - mu) * rs * g0[3] + b0[3]);
                w.z = pack_bf16x2((bf16lo(t.z) - mu) * rs * g1[0] + b1[0], (bf16hi(t.z) - mu) * rs * g1[1] + b1[1]);
                w.w = pack_bf16x2((bf16lo(t.w) - mu) * rs * g1[2] + b1[2], (bf16hi(t.w) - mu) * rs * g1[3] + b1[3]);
                af[f][ks] = w;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // per-lane LDS offsets inside a slot.  Wq image (A operand: lane (i = l15 (+16 nf), g) holds image row i, k = 32 ks + 8 g .. + 8; the swizzle depends on ks & 1);
    // K image row = key, V^T image row = d slot, Wo image row = output column (row-panel order): 64 bytes per K step, lane group g reads bytes 16 g .. + 16.
    int wqoff[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) wqoff[e] = l15 * XA_ROWB + ((((4 * e + g) ^ (l15 >> 1)) & 7) << 4);
    const int koff = l15 * XA_KSTR + g * 16;
    const int voff = XA_KIMGB + l15 * XA_VSTR + g * 16;
    const int ooff = l15 * XA_OSTR + g * 16;
    // logit masks (C operands of the first K step): key 16 kf + 4 g + r of the text segment beyond Nk, expert key 4 g + r beyond T
    f32x4 mask4, mask5;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mask4[r] = (64 + 4 * g + r < p.Nk) ? 0.f : -1.0e30f;
        mask5[r] = (4 * g + r < p.T) ? 0.f : -1.0e30f;
    }

    f32x4 acc_out[XA_MF][XA_NCF];
#pragma unroll
    for (int f = 0; f < XA_MF; ++f)
#pragma unroll
        for (int cf = 0; cf < XA_NCF; ++cf) {
            acc_out[f][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc_out[f][cf]));
        }
    f32x4 accq[XA_MF][3];        // q_h^T pre-activations: d slots 16 nf + 4 g + r of row l15
    u32x4 qf[XA_MF][2];          // ... rounded, scaled: B operand of the logit product (K steps of 32 d slots)
    f32x4 S[XA_MF][6];           // logits (log2 domain): key 16 kf + 4 g + r (kf < 5 text, kf = 5 expert)
    u32x4 Pf[XA_MF][4];          // probabilities: B operand of the PV product (K steps 0 .. 2 text, 3 expert)
    f32x4 acco[XA_MF][3], acco2[XA_MF][3];
    u32x4 of[XA_MF][3];          // the head pair's attention output: B operand of the output projection (three K steps of 32 slots)
    // W fragment ring: a slot is only TWO MFMAs (32 rows per wave), so fragments are read XA_LEAD = 5 slots (160 matrix cycles) ahead — at three the reads of every slot
    // were waited for (profiles/r06_xattn_fused_notes.txt); the last two slots of every phase sit in the last two entries (see run_slots)
    u32x4 wfr[XA_RING];
    wfr[XA_RING - 2] = wfr[XA_RING - 1] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int f = 0; f < XA_MF; ++f) {
        qf[f][1] = (u32x4){0u, 0u, 0u, 0u};
        Pf[f][2] = Pf[f][3] = (u32x4){0u, 0u, 0u, 0u};
    }

    // item j: its pieces (and every other wave's) have landed; every wave is done with item j - 1, whose slot takes item j + 3 (pieces issued between the MFMAs)
    auto open_item = [&]() {
        if (!(XA_LAB & 8)) {
            hp_wait_dma<2 * XA_PPW>();
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // A phase runs NS fragment slots of XA_MF MFMAs each; slot e lives in ring entry (e + RO) % 6 with RO chosen so that the LAST two slots of every phase are
    // entries 4 and 5: they stay MFMA sources until the second MFMA of the next phase (csrc/ff_fused.hip).  `rd(e)` reads slot e's fragment, `mm(e, f)` issues its MFMA.
    auto run_slots = [&](auto ns_c, auto&& piece, auto&& rd, auto&& mm, auto&& valu) {
        constexpr int NS = decltype(ns_c)::value, NP = XA_MF * NS, RO = (XA_RING - NS % XA_RING) % XA_RING, DSTEP = NP / 9 > 0 ? NP / 9 : 1;
        hp_static_for<0, XA_LEAD>([&](auto ec) { constexpr int e = decltype(ec)::value; wfr[(RO + e) % XA_RING] = rd(e); });
        hp_static_for<0, NP>([&](auto pc) {
            constexpr int P = decltype(pc)::value, e = P / XA_MF, f = P % XA_MF;
            if constexpr (f == 0 && e + XA_LEAD < NS) wfr[(e + XA_LEAD + RO) % XA_RING] = rd(e + XA_LEAD);
            mm(std::integral_constant<int, e>{}, std::integral_constant<int, f>{}, wfr[(e + RO) % XA_RING]);
            if constexpr (f == XA_MF - 1 && e >= 1) hp_keep(wfr[(e - 1 + RO) % XA_RING]);     // the slot before: two MFMAs back now
            if constexpr (P == 1) { hp_keep(wfr[XA_RING - 2]); hp_keep(wfr[XA_RING - 1]); }  // the last two slots of the phase before
            if constexpr (!(XA_LAB & 1) && P % DSTEP == DSTEP / 2 && P / DSTEP < XA_PPW) piece(P / DSTEP);   // the refill item's nine pieces go out between the MFMAs
            valu(std::integral_constant<int, P>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto no_valu = [](auto) {};
    auto no_piece = [](int) {};
    auto slot_ptr = [&](int j) { return smem + (j & (XA_NSLOT - 1)) * XA_SLOTB; };

    // ---- Q phase: accq = Wq_h x^T (30 slots: ks major, nf minor)
    auto phase_q = [&](int j, auto rk_c, auto&& valu) {   // rk_c: kind of the item (j + 3) this phase refills
        const char* const sl = slot_ptr(j);
        run_slots(std::integral_constant<int, 30>{}, [&](int jj) { issue_piece(rk_c, j + 3, jj); },
                  [&](int e) { const int ks = e / 3, nf = e % 3; return *reinterpret_cast<const u32x4*>(sl + wqoff[ks & 1] + (ks >> 1) * 128 + nf * 16 * XA_ROWB); },
                  [&](auto ec, auto fc, const u32x4& w) {
                      constexpr int e = decltype(ec)::value, f = decltype(fc)::value, ks = e / 3, nf = e % 3;
                      if constexpr (ks == 0) hp_mfma_v0(accq[f][nf], w, af[f][ks]);
                      else hp_mfma_v(accq[f][nf], w, af[f][ks]);
                  },
                  valu);
    };
    // q -> B operand: rounded to bf16 (what the separate projection stores), scaled by 40^-1/2 log2 e, rounded again (what the attention kernels multiply)
    auto pack_q = [&]() {
        // asm MFMA results -> VALU; the W fragments of the phase's last slots stay sources until a VALU instruction has READ the last result (in-order matrix pipe)
        asm volatile("s_nop 11" : "+v"(accq[0][0]), "+v"(accq[0][1]), "+v"(accq[0][2]), "+v"(accq[1][0]), "+v"(accq[1][1]), "+v"(accq[1][2]) : "v"(wfr[XA_RING - 2]), "v"(wfr[XA_RING - 1]));
        {
            float pr = accq[XA_MF - 1][2][3];
            asm volatile("v_mov_b32 %0, %0" : "+v"(pr) : "v"(wfr[XA_RING - 2]), "v"(wfr[XA_RING - 1]));
            accq[XA_MF - 1][2][3] = pr;
        }
        const float c = p.qscale;
#pragma unroll
        for (int f = 0; f < XA_MF; ++f) {
            uint32_t w[6];
#pragma unroll
            for (int nf = 0; nf < 3; ++nf)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint32_t q2 = pack_bf16x2(accq[f][nf][2 * hh], accq[f][nf][2 * hh + 1]);
                    w[2 * nf + hh] = pack_bf16x2(bf16lo(q2) * c, bf16hi(q2) * c);
                }
            qf[f][0] = (u32x4){w[0], w[1], w[2], w[3]};
            qf[f][1].x = w[4]; qf[f][1].y = w[5];
        }
        asm volatile("s_nop 1" ::"v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[1][0]), "v"(qf[1][1]));   // VALU-written operands -> asm MFMA
        __builtin_amdgcn_sched_barrier(0);
    };
    // ---- A phase: logits, softmax, PV for one head over the KV item
    auto phase_a = [&](int j, auto rk_c) {
        const char* const sl = slot_ptr(j);
        pack_q();
        // S^T = K q^T: 12 slots (key fragment kf = e / 2, K step t = e % 2)
        run_slots(std::integral_constant<int, 12>{}, [&](int jj) { issue_piece(rk_c, j + 3, jj); },
                  [&](int e) { return *reinterpret_cast<const u32x4*>(sl + koff + (e / 2) * 16 * XA_KSTR + (e % 2) * 64); },
                  [&](auto ec, auto fc, const u32x4& w) {
                      constexpr int e = decltype(ec)::value, f = decltype(fc)::value, kf = e / 2, t = e % 2;
                      if constexpr (t == 1) hp_mfma_v(S[f][kf], w, qf[f][1]);
                      else if constexpr (kf < 4) hp_mfma_v0(S[f][kf], w, qf[f][0]);
                      else if constexpr (kf == 4) hp_mfma_vc(S[f][kf], w, qf[f][0], mask4);
                      else hp_mfma_vc(S[f][kf], w, qf[f][0], mask5);
                  },
                  no_valu);
        asm volatile("s_nop 11" : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[0][2]), "+v"(S[0][3]), "+v"(S[0][4]), "+v"(S[0][5]),
                     "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[1][2]), "+v"(S[1][3]), "+v"(S[1][4]), "+v"(S[1][5]) : "v"(wfr[XA_RING - 2]), "v"(wfr[XA_RING - 1]));
        {   // the K fragments stay sources until a VALU instruction has read the last logit result
            float pr = S[XA_MF - 1][5][3];
            asm volatile("v_mov_b32 %0, %0" : "+v"(pr) : "v"(wfr[XA_RING - 2]), "v"(wfr[XA_RING - 1]), "v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[1][0]), "v"(qf[1][1]));   // (and the q operands)
            S[XA_MF - 1][5][3] = pr;
        }
        // softmax of a row: its text logits sit in the row's four lanes (g) x five fragments x four registers; the expert logits in fragment 5
#pragma unroll
        for (int f = 0; f < XA_MF; ++f) {
            if (XA_LAB & 2) {
                Pf[f][0] = (u32x4){__float_as_uint(S[f][0][0]), __float_as_uint(S[f][1][0]), __float_as_uint(S[f][2][0]), __float_as_uint(S[f][3][0])};
                Pf[f][1] = (u32x4){__float_as_uint(S[f][4][0]), __float_as_uint(S[f][5][0]), __float_as_uint(S[f][0][1]), __float_as_uint(S[f][1][1])};
                continue;
            }
            float m1 = fmaxf(fmaxf(S[f][0][0], S[f][0][1]), S[f][0][2]);
            m1 = fmaxf(m1, S[f][0][3]);
#pragma unroll
            for (int kf = 1; kf < 5; ++kf) m1 = fmaxf(fmaxf(fmaxf(m1, S[f][kf][0]), fmaxf(S[f][kf][1], S[f][kf][2])), S[f][kf][3]);
            float m2 = fmaxf(fmaxf(S[f][5][0], S[f][5][1]), fmaxf(S[f][5][2], S[f][5][3]));
            xa_rowmax4(m1, m2);   // over the row's four lanes (l15 + 16 g): VALU lane swaps, no LDS round trip
            uint32_t w[12];
#pragma unroll
            for (int kf = 0; kf < 6; ++kf) {
                const float mm = kf < 5 ? m1 : m2;
                w[2 * kf] = pack_bf16x2(__builtin_amdgcn_exp2f(S[f][kf][0] - mm), __builtin_amdgcn_exp2f(S[f][kf][1] - mm));
                w[2 * kf + 1] = pack_bf16x2(__builtin_amdgcn_exp2f(S[f][kf][2] - mm), __builtin_amdgcn_exp2f(S[f][kf][3] - mm));
            }
            Pf[f][0] = (u32x4){w[0], w[1], w[2], w[3]};
            Pf[f][1] = (u32x4){w[4], w[5], w[6], w[7]};
            Pf[f][2].x = w[8]; Pf[f][2].y = w[9];
            Pf[f][3].x = w[10]; Pf[f][3].y = w[11];
        }
        asm volatile("s_nop 1" ::"v"(Pf[0][0]), "v"(Pf[0][1]), "v"(Pf[0][2]), "v"(Pf[0][3]), "v"(Pf[1][0]), "v"(Pf[1][1]), "v"(Pf[1][2]), "v"(Pf[1][3]));   // VALU-written operands -> asm MFMA
        __builtin_amdgcn_sched_barrier(0);
        // O^T = V^T P^T: 12 slots (e < 9: d fragment df = e / 3, text K step t = e % 3; e >= 9: df = e - 9, the expert K step)
        run_slots(std::integral_constant<int, 12>{}, no_piece /* the refill's pieces went out under the logits */,
                  [&](int e) { const int df = e < 9 ? e / 3 : e - 9, t = e < 9 ? e % 3 : 3; return *reinterpret_cast<const u32x4*>(sl + voff + df * 16 * XA_VSTR + t * 64); },
                  [&](auto ec, auto fc, const u32x4& w) {
                      constexpr int e = decltype(ec)::value, f = decltype(fc)::value;
                      if constexpr (e >= 9) hp_mfma_v0(acco2[f][e - 9], w, Pf[f][3]);
                      else if constexpr (e % 3 == 0) hp_mfma_v0(acco[f][e / 3], w, Pf[f][0]);
                      else hp_mfma_v(acco[f][e / 3], w, Pf[f][e % 3]);
                  },
                  no_valu);
    };
    // attention output of head `hh` of the pair -> its three 16-slot fragments of the projection operand: o = O / l + gate O2 / l2 (l, l2: V^T's row of ones, d slot 40)
    auto pack_o = [&](auto hh_c) {
        constexpr int HH = decltype(hh_c)::value;
        asm volatile("s_nop 11" : "+v"(acco[0][0]), "+v"(acco[0][1]), "+v"(acco[0][2]), "+v"(acco[1][0]), "+v"(acco[1][1]), "+v"(acco[1][2]),
                     "+v"(acco2[0][0]), "+v"(acco2[0][1]), "+v"(acco2[0][2]), "+v"(acco2[1][0]), "+v"(acco2[1][1]), "+v"(acco2[1][2]) : "v"(wfr[XA_RING - 2]), "v"(wfr[XA_RING - 1]));
        {
            float pr = acco2[XA_MF - 1][2][3];
            asm volatile("v_mov_b32 %0, %0" : "+v"(pr) : "v"(wfr[XA_RING - 2]), "v"(wfr[XA_RING - 1]), "v"(Pf[0][2]), "v"(Pf[0][3]), "v"(Pf[1][2]), "v"(Pf[1][3]));   // (and the probabilities)
            acco2[XA_MF - 1][2][3] = pr;
        }
#pragma unroll
        for (int f = 0; f < XA_MF; ++f) {
            const float l1 = __shfl(acco[f][2][0], 32 + l15, 64), l2 = __shfl(acco2[f][2][0], 32 + l15, 64);
            const float i1 = __builtin_amdgcn_rcpf(l1), i2 = l2 > 0.f ? gate * __builtin_amdgcn_rcpf(l2) : 0.f;
            uint32_t w[6];
#pragma unroll
            for (int df = 0; df < 3; ++df)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    w[2 * df + h2] = pack_bf16x2(fmaf(acco2[f][df][2 * h2], i2, acco[f][df][2 * h2] * i1), fmaf(acco2[f][df][2 * h2 + 1], i2, acco[f][df][2 * h2 + 1] * i1));
            // fragment q6 = 3 HH + df of the pair -> K step q6 / 2, half q6 % 2
            if constexpr (HH == 0) { of[f][0] = (u32x4){w[0], w[1], w[2], w[3]}; of[f][1].x = w[4]; of[f][1].y = w[5]; }
            else { of[f][1].z = w[0]; of[f][1].w = w[1]; of[f][2] = (u32x4){w[2], w[3], w[4], w[5]}; }
        }
    };
    // ---- O phase: output accumulators, column fragments 10 half .. + 9, += o (three K steps) x the Wo half image
    auto phase_o = [&](int j, auto half_c, auto rk_c) {
        constexpr int HALF = decltype(half_c)::value;
        const char* const sl = slot_ptr(j);
        run_slots(std::integral_constant<int, 30>{}, [&](int jj) { issue_piece(rk_c, j + 3, jj); },
                  [&](int e) { return *reinterpret_cast<const u32x4*>(sl + ooff + (e / 3) * 16 * XA_OSTR + (e % 3) * 64); },
                  [&](auto ec, auto fc, const u32x4& w) {
                      constexpr int e = decltype(ec)::value, f = decltype(fc)::value;
                      hp_mfma_a(acc_out[f][10 * HALF + e / 3], w, of[f][e % 3]);
                  },
                  no_valu);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;

    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    using C4 = std::integral_constant<int, 4>;
    using C5 = std::integral_constant<int, 5>;
    for (int pp = 0; pp < XA_H / 2; ++pp) {
        const int j = 6 * pp;   // phase of item j refills item j + 3: kinds 3, 4, 5, 0, 1, 2
        open_item(); phase_q(j, C3{}, no_valu);
        open_item(); phase_a(j + 1, C4{});
        open_item(); phase_q(j + 2, C5{}, no_valu);
        pack_o(C0{});                            // (the first head's output: its registers are not touched by the second head's projection)
        open_item(); phase_a(j + 3, C0{});
        pack_o(C1{});
        asm volatile("s_nop 1" ::"v"(of[0][0]), "v"(of[0][1]), "v"(of[0][2]), "v"(of[1][0]), "v"(of[1][1]), "v"(of[1][2]));   // VALU-written operands -> asm MFMA
        __builtin_amdgcn_sched_barrier(0);
        open_item(); phase_o(j + 4, C0{}, C1{});
        open_item(); phase_o(j + 5, C1{}, C2{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::"v"(wfr[0]), "v"(wfr[1]), "v"(wfr[2]), "v"(wfr[3]), "v"(wfr[4]), "v"(wfr[5]), "v"(wfr[6]), "v"(wfr[7]) : "memory");
#pragma unroll
    for (int f = 0; f < XA_MF; ++f)
#pragma unroll
        for (int cf = 0; cf < XA_NCF; ++cf) asm volatile("" : "+a"(acc_out[f][cf]));
    {
        float pr = acc_out[XA_MF - 1][XA_NCF - 1][3];
        asm volatile("" : "+v"(pr) : "v"(wfr[0]), "v"(wfr[1]), "v"(wfr[2]), "v"(wfr[3]), "v"(wfr[4]), "v"(wfr[5]), "v"(wfr[6]), "v"(wfr[7]), "v"(of[0][2]), "v"(of[1][2]));
        acc_out[XA_MF - 1][XA_NCF - 1][3] = pr;
    }

    // ---- epilogue: + bo + residual (the block's input rows, UN-normalised), 16-byte stores (a lane's fragments 2 j / 2 j + 1 are eight consecutive columns)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int f = 0; f < XA_MF; ++f) {
        const int row = m0 + 16 * f + l15;
        bf16_t* const yrow = p.Y + (long)row * p.ldy + 8 * g;
        const bf16_t* const rrow = p.X + (long)row * p.ldx + 8 * g;
#pragma unroll
        for (int jj = 0; jj < XA_NCF / 2; ++jj) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbo + 32 * jj + 8 * g), b1 = *reinterpret_cast<const f32x4*>(sbo + 32 * jj + 8 * g + 4);
            const u32x4 r = *reinterpret_cast<const u32x4*>(rrow + 32 * jj);
            const float o0 = acc_out[f][2 * jj][0] + b0[0] + bf16lo(r.x), o1 = acc_out[f][2 * jj][1] + b0[1] + bf16hi(r.x);
            const float o2 = acc_out[f][2 * jj][2] + b0[2] + bf16lo(r.y), o3 = acc_out[f][2 * jj][3] + b0[3] + bf16hi(r.y);
            const float o4 = acc_out[f][2 * jj + 1][0] + b1[0] + bf16lo(r.z), o5 = acc_out[f][2 * jj + 1][1] + b1[1] + bf16hi(r.z);
            const float o6 = acc_out[f][2 * jj + 1][2] + b1[2] + bf16lo(r.w), o7 = acc_out[f][2 * jj + 1][3] + b1[3] + bf16hi(r.w);
            *reinterpret_cast<u32x4*>(yrow + 32 * jj) = (u32x4){pack_bf16x2(o0, o1), pack_bf16x2(o2, o3), pack_bf16x2(o4, o5), pack_bf16x2(o6, o7)};
        }
    }
}

}  // namespace

// 1 when the fused cross-attention kernel is switched on (AE_XATTN_FUSED=1) and covers the shape.  OFF by default: parity-green, but measured SLOWER than the three launches
// it replaces (74.9 vs 67.1 us isolated and hot, +0.03 .. +0.07 ms per UNet evaluation in three alternating pairs: profiles/r06_xattn_fused_notes.txt) — 128-row blocks are
// one and a half rounds on 256 CUs, and with one block of one wave per SIMD on a CU nothing overlaps a block's 80 KB prologue read / LayerNorm and its epilogue's residual
// read + store (18 of the 54 us that remain with DMA, softmax and barriers ablated away).
// ae_xattn_fused_covers: the shapes the kernel can run (what ae_xattn_fused_bf16 itself requires).  ae_xattn_fused_supported: the PLAN question the module asks — switched on
// AND covered AND at least one block per CU (AE_ROWPANEL_ANY_M lifts the last for small test shapes).
extern "C" int ae_xattn_fused_covers(int M, int C, int heads, int head_dim, int rows_per_sample, int Nk, int T) {
    return (C == XA_K && heads == XA_H && head_dim == 40 && M % XA_BM == 0 && M >= XA_BM && rows_per_sample % XA_BM == 0 && rows_per_sample > 0 && M % rows_per_sample == 0 &&
            Nk > 64 && Nk <= 80 && T >= 0 && T <= 16) ? 1 : 0;
}
extern "C" int ae_xattn_fused_supported(int M, int C, int heads, int head_dim, int rows_per_sample, int Nk, int T) {
    static const int on = getenv("AE_XATTN_FUSED") ? atoi(getenv("AE_XATTN_FUSED")) : 0;
    static const int any_m = getenv("AE_ROWPANEL_ANY_M") ? atoi(getenv("AE_ROWPANEL_ANY_M")) : 0;
    if (!on) return 0;
    if (!any_m && M / XA_BM < 256) return 0;   // 128-row blocks: below one block per CU the three-launch path fills the chip better
    return ae_xattn_fused_covers(M, C, heads, head_dim, rows_per_sample, Nk, T);
}
extern "C" long ae_xattn_fused_kv_bytes(void) { return XA_KVB; }

extern "C" int ae_xattn_fused_bf16(const void* X, long ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* Wq_img, const void* KV_img,
                                   const float* gate, const void* Wo_img, const float* bo, void* Y, long ldy, int M, int rows_per_sample, int Nk, int T,
                                   float scale, void* stream) {
    AE_REQUIRE(X && ln_gamma && ln_beta && Wq_img && KV_img && Wo_img && Y, "ae_xattn_fused_bf16: null pointer");
    AE_REQUIRE(ae_xattn_fused_covers(M, XA_K, XA_H, 40, rows_per_sample, Nk, T), "ae_xattn_fused_bf16: unsupported shape M=%d rows/sample=%d Nk=%d T=%d (C = 320, 8 heads of 40, "
               "M and rows per sample multiples of 128, 64 < Nk <= 80, T <= 16)", M, rows_per_sample, Nk, T);
    AE_REQUIRE(ln_eps >= 0.f && scale > 0.f, "ae_xattn_fused_bf16: eps / scale");
    AE_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)Wq_img & 15) == 0 && ((uintptr_t)KV_img & 15) == 0 &&
               ((uintptr_t)Wo_img & 15) == 0, "ae_xattn_fused_bf16: 16-byte alignment");
    AE_REQUIRE(Y != X, "ae_xattn_fused_bf16: in-place operation is not supported");
    AE_REQUIRE(((long)M + XA_BM) * (ldx > ldy ? ldx : ldy) * 2 < (1L << 30), "ae_xattn_fused_bf16: tensors must stay below 1 GiB (32-bit offsets, out-of-range marker)");
    XAArgs a{};
    a.X = (const bf16_t*)X; a.Y = (bf16_t*)Y; a.Wq = (const bf16_t*)Wq_img; a.KV = (const bf16_t*)KV_img; a.Wo = (const bf16_t*)Wo_img;
    a.bo = bo; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.gate = gate; a.ln_eps = ln_eps; a.qscale = scale * 1.4426950408889634f;
    a.M = M; a.rows_per_sample = rows_per_sample; a.Nk = Nk; a.T = T; a.ldx = ldx; a.ldy = ldy;
    hipLaunchKernelGGL(xattn_fused_kernel, dim3((unsigned)(M / XA_BM)), dim3(64 * XA_NW), 0, (hipStream_t)stream, a);
    return ae_check_launch("ae_xattn_fused_bf16");
}
