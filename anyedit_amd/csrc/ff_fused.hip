// Fused feed-forward of a BasicTransformerBlock at the 64x64 UNet level (C = 320) for gfx950 / MI355X:
//
//   Y[M,320] = ( a .* gelu(g) ) W2^T + b2 + R,     [a | g] = LayerNorm(X) W1^T + b1
//
// Replaces: FeedForward / GEGLU of ldm/modules/attention.py:49-76 behind norm3 of BasicTransformerBlock._forward
// (attention.py:271-275, `x = self.ff(self.norm3(x)) + x`) — until round 5 two launches (row-panel GEGLU projection, 192x320 dense ff2) that wrote
// and re-read the [M, 1280] gated hidden activation (126 MB at UNet batch 12).  Here that activation never leaves the registers:
//   * one block per CU, FOUR waves (one per SIMD, 512 registers each); a wave owns 48 rows of the panel for the whole launch: the MFMA operand
//     fragments of their normalised X (120 VGPRs) AND their fp32 output accumulators [48 x 320] (240 registers: the accumulator half of the file);
//   * the hidden dimension is walked in 16-unit chunks: P1 = the chunk's [a | g] pre-activations (60 MFMAs, K = 320) -> gate in registers -> the
//     gated values of two chunks ARE the B operand of P2 (the k order of an MFMA contraction is free: W2's LDS image is permuted on the host to the
//     order the P1 result registers come in: ops.pack_ff2_fused) -> P2 = 60 MFMAs into the output accumulators.  No LDS round trip, no exchange
//     between waves;
//   * one wave per SIMD means nothing hides a VALU chain or an LDS read but the wave's own MFMAs, so the loop is a software pipeline PLACED BY HAND:
//     a phase = 90 MFMAs (60 of P1(c + 1), 30 of P2(step - 1)) with the 24 slices of gate(c) between them (2.1 VALU per MFMA: what one wave hides
//     behind a 16x16x32, DESIGN.md §7b issue model), W fragments read three slots ahead.  Every MFMA is an asm statement (program order kept; P1
//     results pinned to VGPRs, output accumulators to AGPRs), every slot ends in a scheduling barrier.  hipcc's own order of the same code —
//     builtins, with or without sched_group_barrier pipelines — clusters 60 MFMAs and then 250 VALU and moves ~80 values per step between the two
//     register files (profiles/r06_ff_fused_notes.txt);
//   * weights stream through a four-slot LDS ring of 32 KiB phase items (W1 chunk image 20 KiB, as gemm_rowpanel.hip; W2 half-chunk image 10 KiB)
//     by LDS-DMA with counted waits: two phases of lead, ONE barrier per phase (two per 32 hidden units).
#include "common.hpp"
#include "handplaced.hpp"
#include <stdlib.h>

namespace {

struct FFArgs {
    const bf16_t* X; const bf16_t* W1; const bf16_t* W2; bf16_t* Y; const bf16_t* res;
    const float* b1; const float* ln_g; const float* ln_b; const float* b2;
    float ln_eps;
    int M, H;
    long ldx, ldw1, ldy, ldr;
    // PROJ (SpatialTransformer.proj_out behind the block, attention.py:337-340): Y = (feed-forward result, bf16) W3^T + b3 + res3
    const bf16_t* W3; const float* b3; const bf16_t* res3;
    long ldw3, ldr3;
};

constexpr int FF_KS = 10, FF_K = 320, FF_ROWB = 640, FF_CHUNKB = 32 * FF_ROWB, FF_MF = 3, FF_BM = 192, FF_NW = 4, FF_NCF = 20;
constexpr int FF_NSLOT = 4, FF_SLOTB = 32768, FF_PPW = 8;   // phase item: 32 pieces of 1 KiB (20 W1, 10 W2 half, 2 unused), eight per wave
constexpr int FF_W2HALFB = 160 * 64;
constexpr int FF_MAXH2 = 2560;
#ifndef FF_LAB
#define FF_LAB 0      // lab builds only (timing ablations, wrong results): 1 no DMA in the loop, 2 no gate arithmetic, 4 no W fragment reads, 8 no barrier / DMA wait
#endif
constexpr int FF_OOB = 0x40000000;                            // a per-lane offset beyond every descriptor: the piece reads zeros

// The gate of one row fragment (four values of a lane: a_half * (g + |g| w(g)), the arithmetic of geglu_half_f / gelu_w_f in common.hpp — bit-identical gated
// values to the row-panel GEGLU epilogue) as 62 single VALU operations; operation u of a stage works on chain r = u & 3, so dependent operations sit four apart.
// Gate operations [ff_op_lo(P), ff_op_lo(P + 1)) of a phase (186 = 3 fragments x 62) go behind its MFMA P (90).
constexpr int ff_op_lo(int P) { return (P * 186 + 89) / 90; }
template <int U>
__device__ __forceinline__ void ff_gate_op(const f32x4& va, const f32x4& vg, const f32x4& ba, const f32x4& bg, float (&xa)[4], float (&xg)[4], float (&t4)[4],
                                           float (&pl)[4], float (&e4)[4], u32x4& h, int hi) {
    constexpr int r = U & 3;
    if constexpr (U < 4) xa[r] = fmaf(va[r], 0.5f, ba[r]);                                        // a_half = 0.5 acc + 0.5 bias (ba holds the halved bias)
    else if constexpr (U < 8) xg[r] = vg[r] + bg[r];
    else if constexpr (U < 12) t4[r] = __builtin_fmaf(0.3275911f * 0.70710678118654752440f, __builtin_fabsf(xg[r]), 1.0f);
    else if constexpr (U < 16) t4[r] = __builtin_amdgcn_rcpf(t4[r]);
    else if constexpr (U < 20) pl[r] = __builtin_fmaf(1.061405429f, t4[r], -1.453152027f);
    else if constexpr (U < 24) pl[r] = __builtin_fmaf(pl[r], t4[r], 1.421413741f);
    else if constexpr (U < 28) pl[r] = __builtin_fmaf(pl[r], t4[r], -0.284496736f);
    else if constexpr (U < 32) pl[r] = __builtin_fmaf(pl[r], t4[r], 0.254829592f);
    else if constexpr (U < 36) pl[r] *= t4[r];
    else if constexpr (U < 40) e4[r] = (-0.5f * 1.4426950408889634f) * __builtin_fabsf(xg[r]);
    else if constexpr (U < 44) e4[r] = e4[r] * __builtin_fabsf(xg[r]);
    else if constexpr (U < 48) e4[r] = __builtin_amdgcn_exp2f(e4[r]);
    else if constexpr (U < 52) e4[r] = __builtin_fmaf(-pl[r], e4[r], 1.0f);
    else if constexpr (U < 56) e4[r] = __builtin_fmaf(__builtin_fabsf(xg[r]), e4[r], xg[r]);
    else if constexpr (U < 60) xa[r] = xa[r] * e4[r];
    else if constexpr (U == 60) { if (hi) h.z = pack_bf16x2(xa[0], xa[1]); else h.x = pack_bf16x2(xa[0], xa[1]); }
    else { if (hi) h.w = pack_bf16x2(xa[2], xa[3]); else h.y = pack_bf16x2(xa[2], xa[3]); }
}

template <bool PROJ>
__global__ __launch_bounds__(64 * FF_NW, 1) void ff_fused_kernel(const FFArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[FF_NSLOT * FF_SLOTB + FF_MAXH2 * 4 + 4 * FF_K * 4];
    float* const sc1 = reinterpret_cast<float*>(smem + FF_NSLOT * FF_SLOTB);   // b1 in the packed column order, 'a' columns halved
    float* const sb2 = sc1 + FF_MAXH2;
    float* const sln = sb2 + FF_K;                                             // gamma | beta
    float* const sb3 = sln + 2 * FF_K;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * FF_BM + wave * (FF_MF * 16);
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    const int H2 = 2 * p.H, nsteps = p.H / 32, nchunks = p.H / 16;

    const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W1), 0, (int)((long)H2 * p.ldw1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W2), 0, nsteps * FF_CHUNKB, 0x00020000);
    // W1 chunk image: [32 rows][640 B], the 16-byte pieces of a row XOR-swizzled inside groups of eight (gemm_rowpanel.hip); a wave's pieces q = wave + 4 j, j < 5
    int dma1[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int o = (wave + FF_NW * j) * 1024 + lane * 16;
        const int i = o / FF_ROWB, pp = (o - i * FF_ROWB) >> 4;
        dma1[j] = i * (int)p.ldw1 * 2 + ((pp & ~7) | ((pp ^ (i >> 1)) & 7)) * 16;
    }
    // W2 half image: 10 pieces, stored ready-made (linear copy): a wave's pieces wave + 4 j', j' < 3 (waves 2 and 3 have no third one: it reads zeros into the pad)
    const int dma2 = lane * 16;
    const int dma2_last = lane * 16 + (wave >= 2 ? FF_OOB : 0);
    const int w1_chunk_stride = 32 * (int)p.ldw1 * 2;
    // phase item j: W1 chunk j and half (j + 1) & 1 of W2 chunk ((j + 1) >> 1) - 2; what does not exist (before the first / past the last) reads zeros.
    // A wave's eight pieces of an item: five of the W1 image, three of the W2 half image.
    struct Item { int slot, soff1, ob1, soff2, ob2; };
    auto item_of = [&](int j) {
        Item it;
        it.slot = lds0 + (j & (FF_NSLOT - 1)) * FF_SLOTB;
        it.ob1 = j < nchunks ? 0 : FF_OOB;
        it.soff1 = (j < nchunks ? j : 0) * w1_chunk_stride;
        const int s2 = ((j + 1) >> 1) - 2;
        const bool ok2 = s2 >= 0 && s2 < nsteps;
        it.ob2 = ok2 ? 0 : FF_OOB;
        it.soff2 = (ok2 ? s2 : 0) * FF_CHUNKB + ((j + 1) & 1) * FF_W2HALFB;
        return it;
    };
    auto issue_piece = [&](const Item& it, int jj) {
        if (jj < 5) ae_dma16(rsW1, it.slot + (wave + FF_NW * jj) * 1024, dma1[jj] + it.ob1, it.soff1);
        else if (jj < 7) ae_dma16(rsW2, it.slot + FF_CHUNKB + (wave + FF_NW * (jj - 5)) * 1024, dma2 + it.ob2, it.soff2 + (wave + FF_NW * (jj - 5)) * 1024);
        else ae_dma16(rsW2, it.slot + FF_CHUNKB + (wave + FF_NW * 2) * 1024, dma2_last + it.ob2, it.soff2 + (wave + FF_NW * 2) * 1024);
    };
    auto issue_item = [&](int j) {
        const Item it = item_of(j);
#pragma unroll
        for (int jj = 0; jj < FF_PPW; ++jj) issue_piece(it, jj);
    };

    // ---- X panel: 192 rows = six 32-row chunk images at the start of the ring (as gemm_rowpanel.hip), then every wave picks its fragments
    {
        const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.X), 0, (int)((long)p.M * p.ldx * 2), 0x00020000);
        const int soff = blockIdx.x * FF_BM * (int)p.ldx * 2;
#pragma unroll
        for (int j = 0; j < 30; ++j) {
            const int q = wave + FF_NW * j;
            const int o = (q % 20) * 1024 + lane * 16;
            const int i = o / FF_ROWB, pp = (o - i * FF_ROWB) >> 4;
            ae_dma16(rsX, lds0 + q * 1024, ((q / 20) * 32 + i) * (int)p.ldx * 2 + ((pp & ~7) | ((pp ^ (i >> 1)) & 7)) * 16, soff);
        }
    }
    for (int i = tid; i < H2; i += 64 * FF_NW) sc1[i] = p.b1[i] * ((i & 16) == 0 ? 0.5f : 1.0f);   // the bias of an 'a' column is kept halved (geglu_half_f)
    for (int i = tid; i < FF_K; i += 64 * FF_NW) { sb2[i] = p.b2 ? p.b2[i] : 0.f; sln[i] = p.ln_g[i]; sln[FF_K + i] = p.ln_b[i]; sb3[i] = (PROJ && p.b3) ? p.b3[i] : 0.f; }
    hp_wait_dma<0>();
    __syncthreads();
    u32x4 af[FF_MF][FF_KS];
#pragma unroll
    for (int f = 0; f < FF_MF; ++f) {
        const int rl = wave * (FF_MF * 16) + 16 * f + l15;
        const int i = rl & 31;
        const char* base = smem + (rl >> 5) * FF_CHUNKB + i * FF_ROWB;
#pragma unroll
        for (int ks = 0; ks < FF_KS; ++ks) {
            const int c16 = 4 * ks + g;
            af[f][ks] = *reinterpret_cast<const u32x4*>(base + (((c16 & ~7) | ((c16 ^ (i >> 1)) & 7)) << 4));
        }
    }
    __syncthreads();  // the ring now belongs to the weights
    issue_item(0);
    issue_item(1);
    issue_item(2);

    // LayerNorm on the registers (fp32 statistics, biased variance, eps inside the root: F.layer_norm) — the prologue form of gemm_rowpanel.hip.  Here every
    // wave owns its rows alone (nothing is normalised twice), and ~2 300 VALU instructions once per launch are 2 % of the loop, while the fold form would
    // put four more FMAs and 18 more live registers into every gated value of a kernel that lives at the edge of the register file.  Runs under the first
    // items' DMA.
    {
        float mean[FF_MF], rstd[FF_MF];
#pragma unroll
        for (int f = 0; f < FF_MF; ++f) {
            float sm = 0.f;
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) {
                const u32x4 t = af[f][ks];
                sm += (bf16lo(t.x) + bf16hi(t.x)) + (bf16lo(t.y) + bf16hi(t.y)) + (bf16lo(t.z) + bf16hi(t.z)) + (bf16lo(t.w) + bf16hi(t.w));
            }
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            const float mu = sm * (1.0f / FF_K);
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) asm volatile("" : "+v"(af[f][ks]));   // re-unpack instead of keeping 80 floats alive (see gemm_rowpanel.hip)
            float v = 0.f;
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) {
                const u32x4 t = af[f][ks];
                const float d0 = bf16lo(t.x) - mu, d1 = bf16hi(t.x) - mu, d2 = bf16lo(t.y) - mu, d3 = bf16hi(t.y) - mu;
                const float d4 = bf16lo(t.z) - mu, d5 = bf16hi(t.z) - mu, d6 = bf16lo(t.w) - mu, d7 = bf16hi(t.w) - mu;
                v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
            }
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            mean[f] = mu;
            rstd[f] = __builtin_amdgcn_rsqf(v * (1.0f / FF_K) + p.ln_eps);
#pragma unroll
            for (int ks = 0; ks < FF_KS; ++ks) asm volatile("" : "+v"(af[f][ks]));
        }
#pragma unroll
        for (int ks = 0; ks < FF_KS; ++ks) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sln + 32 * ks + 8 * g), g1 = *reinterpret_cast<const f32x4*>(sln + 32 * ks + 8 * g + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sln + FF_K + 32 * ks + 8 * g), b1 = *reinterpret_cast<const f32x4*>(sln + FF_K + 32 * ks + 8 * g + 4);
#pragma unroll
            for (int f = 0; f < FF_MF; ++f) {
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

    // W1 fragment addresses inside a slot (A operand: lane (i = l15 (+16), g) holds image row i, k = 32 ks + 8 g .. +8): the swizzle depends on ks & 1 only
    int w1off[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int c16 = 4 * e + g;
        w1off[e] = l15 * FF_ROWB + (((c16 ^ (l15 >> 1)) & 7) << 4);
    }
    // W2 fragment address: image row 16 cf' + l15 (64 bytes per row), logical 16-byte piece g at position (g + 2 (row >> 2)) & 3: a ds_read_b128 lane group
    // ({0-3, 12-15, 20-27}, ...) then covers sixteen different 16-byte bank slots
    const int w2off = FF_CHUNKB + l15 * 64 + (((g + 2 * (l15 >> 2)) & 3) << 4);
    // a ds_read's immediate offset is 16 bits and the ring is 128 KiB: slots 2 and 3 are addressed from a second set of base registers (made opaque,
    // else hipcc folds them back into base + constant and spends a v_add_u32 per read: 77 per trip in the first listing)
    int w1offH[2] = {w1off[0] + 2 * FF_SLOTB, w1off[1] + 2 * FF_SLOTB}, w2offH = w2off + 2 * FF_SLOTB;
    asm volatile("" : "+v"(w1offH[0]), "+v"(w1offH[1]), "+v"(w2offH));

    f32x4 acc2[FF_MF][FF_NCF];
#pragma unroll
    for (int f = 0; f < FF_MF; ++f)
#pragma unroll
        for (int cf = 0; cf < FF_NCF; ++cf) {
            acc2[f][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc2[f][cf]));
        }
    f32x4 acc1[2][FF_MF][2];
    u32x4 hf[2][FF_MF];
    // W fragment ring (slot k of a phase -> wfr[k % 6], read three slots ahead; 30 slots per phase, so the indices repeat phase after phase).  It lives
    // ACROSS the phases: the fragments of a phase's last two slots (wfr[4], wfr[5]) are MFMA sources until two further MFMAs have been issued, i.e. until
    // the second MFMA of the NEXT phase, whose own reads reach those entries only at its slots 1 and 2 — tests/test_isa_static.py found hipcc parking a
    // constant in such a register two instructions behind its MFMA when the ring was local to a phase.
    u32x4 wfr[6];
    wfr[4] = wfr[5] = (u32x4){0u, 0u, 0u, 0u};

    // ---- One phase over the item at `slot` (30 fragment slots of three MFMAs: P1, P1, P2, ...):
    //   P1: acc1[WB] = [a | g] pre-activations of the item's W1 chunk (20 slots);
    //   P2: output accumulators, column fragments 10 HALF .. + 9, += gated hidden of the step before (hf[1 - HB]) x the item's W2 half image (10 slots);
    //   gate: the chunk before (acc1[1 - WB], bias at sc1 + gate_n0) in 24 slices between the MFMAs -> half HALF of hf[HB].
    auto phase = [&](auto wb_c, auto hb_c, auto half_c, auto p1_c, auto gate_c, auto p2_c, auto slot_c, int gate_n0, int refill) {
        constexpr int WB = decltype(wb_c)::value, HB = decltype(hb_c)::value, HALF = decltype(half_c)::value, SLOT = decltype(slot_c)::value;
        // the slot of the item before (every wave has left it: this phase's barrier) takes item `refill`: its eight pieces go out BETWEEN the MFMAs (one per
        // ten: a piece costs 60-190 issue cycles, and eight of them in front of the phase were a quarter of it: profiles/r06_ff_fused_notes.txt)
        const Item rit = item_of(refill);
        const char* const slot = smem + (SLOT & 1) * FF_SLOTB;
        const int w1o0 = SLOT >= 2 ? w1offH[0] : w1off[0], w1o1 = SLOT >= 2 ? w1offH[1] : w1off[1], w2o = SLOT >= 2 ? w2offH : w2off;
        constexpr bool DP1 = decltype(p1_c)::value, DOG = decltype(gate_c)::value, DP2 = decltype(p2_c)::value;
        f32x4 ba, bg;
        if (DOG) {
            ba = *reinterpret_cast<const f32x4*>(sc1 + gate_n0 + 4 * g);
            bg = *reinterpret_cast<const f32x4*>(sc1 + gate_n0 + 16 + 4 * g);
        }
        // Fragment slots of the phase, in issue order (three MFMAs each): a full phase runs the 30 slots P1, P1, P2, ...; the prologue (P1 alone) and the tail
        // (P2 alone) run the 20 / 10 slots of their kind back to back, so that "two further MFMAs" means the same everywhere.  Slot e is fragment k(e) of
        // the item: k % 3 == 2 -> P2 column fragment k / 3 of the half; else P1 fragment i = 2 (k / 3) + k % 3 (ks = i >> 1, nf = i & 1).  Ring entry of slot
        // e: (e + 30 - NS) % 6 — the last two slots of EVERY phase sit in wfr[4], wfr[5].
        constexpr int NS = (DP1 ? 20 : 0) + (DP2 ? 10 : 0), NP = 3 * NS, ROFF = 30 - NS;
        constexpr int DSTEP = NP / 9, DOFF = DSTEP > 6 ? 6 : DSTEP / 2;           // refill piece i behind MFMA DSTEP i + DOFF (full phase: 10 i + 6)
        auto kslot = [](int e) { return (DP1 && DP2) ? e : (DP1 ? (e / 2) * 3 + e % 2 : 3 * e + 2); };
        auto wread = [&](int e) {
            const int k = kslot(e);
            if (k % 3 == 2) return *reinterpret_cast<const u32x4*>(slot + w2o + (k / 3) * 1024);
            const int i = 2 * (k / 3) + k % 3;
            return *reinterpret_cast<const u32x4*>(slot + (((i >> 1) & 1) ? w1o1 : w1o0) + (i >> 2) * 128 + (i & 1) * 16 * FF_ROWB);
        };
        wfr[ROFF % 6] = wread(0); wfr[(ROFF + 1) % 6] = wread(1); wfr[(ROFF + 2) % 6] = wread(2);
        float xa[4], xg[4], t4[4], pl[4], e4[4];
        hp_static_for<0, NP>([&](auto pc) {
            constexpr int P = decltype(pc)::value, e = P / 3, f = P % 3, k = (DP1 && DP2) ? e : (DP1 ? (e / 2) * 3 + e % 2 : 3 * e + 2);
            if constexpr (!(FF_LAB & 4) && f == 0 && e + 3 < NS) wfr[(e + 3 + ROFF) % 6] = wread(e + 3);
            if constexpr (k % 3 == 2) {
                hp_mfma_a(acc2[f][10 * HALF + k / 3], wfr[(e + ROFF) % 6], hf[1 - HB][f]);
            } else {
                constexpr int i = 2 * (k / 3) + k % 3, ks = i >> 1, nf = i & 1;
                if constexpr (ks == 0) hp_mfma_v0(acc1[WB][f][nf], wfr[(e + ROFF) % 6], af[f][ks]);
                else hp_mfma_v(acc1[WB][f][nf], wfr[(e + ROFF) % 6], af[f][ks]);
            }
            if constexpr (f == 1 && e >= 1) hp_keep(wfr[(e - 1 + ROFF) % 6]);   // the slot before: its last MFMA is two MFMAs back now
            if constexpr (P == 1) {   // sources of the last MFMAs of the phase before: its last two W fragments, and the gated values its P2 read (the buffer this step's gate refills)
                hp_keep(wfr[4]); hp_keep(wfr[5]);
                hp_keep(hf[HB][0]); hp_keep(hf[HB][1]); hp_keep(hf[HB][2]);
            }
            if constexpr (!(FF_LAB & 1) && P % DSTEP == DOFF && P / DSTEP < FF_PPW) issue_piece(rit, P / DSTEP);
            // the gate arithmetic as single VALU operations, two (sometimes three) behind every MFMA: 3 fragments x 62 operations over the 90 MFMAs.  (In
            // slices of eight behind every fourth MFMA — the first form — the wave hid two of each eight: +30 us per launch, profiles/r06_ff_fused_notes.txt.)
            if constexpr (DOG) {
                static_assert(!DOG || NP == 90, "the gate rides in full phases");
                hp_static_for<ff_op_lo(P), ff_op_lo(P + 1)>([&](auto tc) {
                    constexpr int t = decltype(tc)::value, gf = t / 62, u = t % 62;
                    ff_gate_op<u>(acc1[1 - WB][gf][0], acc1[1 - WB][gf][1], ba, bg, xa, xg, t4, pl, e4, hf[HB][gf], HALF);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // phase item j: its pieces (and every other wave's) have landed; every wave is done with item j - 1, whose slot takes item j + 3
    auto open_item = [&]() {
        if (!(FF_LAB & 8)) {
            hp_wait_dma<2 * FF_PPW>();
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;

    using GT = std::integral_constant<bool, !(FF_LAB & 2)>;
    open_item();                                                                           // P1(0) alone
    phase(C0{}, C0{}, C0{}, T{}, F{}, F{}, C0{}, 0, 3);
    // two steps per trip: items 2 s + 1 .. 2 s + 4 sit in slots 1, 2, 3, 0 (compile-time LDS offsets).  Step s: phase A = P1(2 s + 1) | gate(2 s) |
    // P2(s - 1) half 0, phase B = P1(2 s + 2) | gate(2 s + 1) | P2(s - 1) half 1; the gated values of step s go to hf[s & 1].
    for (int s = 0; s < nsteps; s += 2) {
        const int j = 2 * s + 1;
        open_item();
        phase(C1{}, C0{}, C0{}, T{}, GT{}, T{}, C1{}, (2 * s) * 32, j + 3);
        open_item();
        phase(C0{}, C0{}, C1{}, T{}, GT{}, T{}, C2{}, (2 * s + 1) * 32, j + 4);
        open_item();
        phase(C1{}, C1{}, C0{}, T{}, GT{}, T{}, C3{}, (2 * s + 2) * 32, j + 5);
        open_item();
        phase(C0{}, C1{}, C1{}, T{}, GT{}, T{}, C0{}, (2 * s + 3) * 32, j + 6);
    }
    // tail: P2 of the last step (nsteps even: its gated values are in hf[1], its items in slots 1 and 2)
    open_item();
    phase(C1{}, C0{}, C0{}, F{}, F{}, T{}, C1{}, 0, 2 * nsteps + 4);
    open_item();
    phase(C0{}, C0{}, C1{}, F{}, F{}, T{}, C2{}, 0, 2 * nsteps + 5);
    asm volatile("s_nop 15\n\ts_nop 15" ::"v"(wfr[0]), "v"(wfr[1]), "v"(wfr[2]), "v"(wfr[3]), "v"(wfr[4]), "v"(wfr[5]) : "memory");   // the last MFMAs' results are read below (asm MFMAs: hipcc pads nothing); their sources live until here
#pragma unroll
    for (int f = 0; f < FF_MF; ++f)
#pragma unroll
        for (int cf = 0; cf < FF_NCF; ++cf) asm volatile("" : "+a"(acc2[f][cf]));
    {   // the W fragments stay sources until an instruction has READ the result of the last MFMA issued (in-order matrix pipe: everything before it has finished)
        float pr = acc2[FF_MF - 1][FF_NCF - 1][3];
        asm volatile("" : "+v"(pr) : "v"(wfr[0]), "v"(wfr[1]), "v"(wfr[2]), "v"(wfr[3]), "v"(wfr[4]), "v"(wfr[5]));
        acc2[FF_MF - 1][FF_NCF - 1][3] = pr;
    }

    // ---- epilogue of the feed-forward: + b2 + residual.  A lane's fragments 2 j / 2 j + 1 are eight consecutive columns 32 j + 8 g .. + 7 (the W2 image's row
    // order): a 16-byte store — or, PROJ, exactly the B operand fragment of k step j of the next product: the rounded result replaces the X fragments in `af`.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    [[maybe_unused]] int dma3[5], dmaR[3];
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsW3 = rsW1, rsR3 = rsW1;
    if constexpr (PROJ) {
        // Phase 3 streams ten items [W3 chunk image (32 output columns, 20 KiB, the row-panel kernel's image and row order) | the 192 x 32 piece of the residual
        // (12 KiB: rows of 64 bytes, the 16-byte part of lane group g at position (g + 2 (row >> 2)) & 3)] through the same four slots.  The ring must be quiet
        // first: the tail phases' refills (items past the end: zeros) were drained by the wait above; this barrier says every wave has left the last slot.
        __builtin_amdgcn_s_barrier();
        rsW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W3), 0, (int)((long)FF_K * p.ldw3 * 2), 0x00020000);
        rsR3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.res3), 0, (int)((long)p.M * p.ldr3 * 2), 0x00020000);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int o = (wave + FF_NW * j) * 1024 + lane * 16;
            const int i = o / FF_ROWB, pp = (o - i * FF_ROWB) >> 4;
            const int wrow = 8 * ((i & 15) >> 2) + 4 * (i >> 4) + (i & 3);   // image row i holds W3 row n0 + wrow: a lane's two fragments are eight consecutive output columns
            dma3[j] = wrow * (int)p.ldw3 * 2 + ((pp & ~7) | ((pp ^ (i >> 1)) & 7)) * 16;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int row = 16 * (wave + FF_NW * j) + (lane >> 2), part = ((lane & 3) - 2 * (row >> 2)) & 3;
            dmaR[j] = row * (int)p.ldr3 * 2 + part * 16 + (blockIdx.x * FF_BM + row < p.M ? 0 : FF_OOB);
        }
    }
    auto issue_item3 = [&](int c) {   // item c of phase 3 into slot c & 3; c >= 10 reads zeros
        if constexpr (PROJ) {
            const int slot = lds0 + (c & (FF_NSLOT - 1)) * FF_SLOTB, ob = c < FF_K / 32 ? 0 : FF_OOB, cc = c < FF_K / 32 ? c : 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) ae_dma16(rsW3, slot + (wave + FF_NW * j) * 1024, dma3[j] + ob, cc * 32 * (int)p.ldw3 * 2);
#pragma unroll
            for (int j = 0; j < 3; ++j) ae_dma16(rsR3, slot + FF_CHUNKB + (wave + FF_NW * j) * 1024, dmaR[j] + ob, blockIdx.x * FF_BM * (int)p.ldr3 * 2 + cc * 64);
        }
    };
    if constexpr (PROJ) { issue_item3(0); issue_item3(1); issue_item3(2); }
#pragma unroll
    for (int f = 0; f < FF_MF; ++f) {
        const int row = m0 + 16 * f + l15;
        const bool rok = row < p.M;
        bf16_t* const yrow = p.Y + (long)min(row, p.M - 1) * p.ldy + 8 * g;
        const bf16_t* const rrow = p.res ? p.res + (long)min(row, p.M - 1) * p.ldr + 8 * g : nullptr;
#pragma unroll
        for (int j = 0; j < FF_NCF / 2; ++j) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb2 + 32 * j + 8 * g), b1 = *reinterpret_cast<const f32x4*>(sb2 + 32 * j + 8 * g + 4);
            float o0 = acc2[f][2 * j][0] + b0[0], o1 = acc2[f][2 * j][1] + b0[1], o2 = acc2[f][2 * j][2] + b0[2], o3 = acc2[f][2 * j][3] + b0[3];
            float o4 = acc2[f][2 * j + 1][0] + b1[0], o5 = acc2[f][2 * j + 1][1] + b1[1], o6 = acc2[f][2 * j + 1][2] + b1[2], o7 = acc2[f][2 * j + 1][3] + b1[3];
            if (rrow) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(rrow + 32 * j);
                o0 += bf16lo(r.x); o1 += bf16hi(r.x); o2 += bf16lo(r.y); o3 += bf16hi(r.y);
                o4 += bf16lo(r.z); o5 += bf16hi(r.z); o6 += bf16lo(r.w); o7 += bf16hi(r.w);
            }
            const u32x4 pk = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3), pack_bf16x2(o4, o5), pack_bf16x2(o6, o7)};
            if constexpr (PROJ) af[f][j] = pk;
            else if (rok) *reinterpret_cast<u32x4*>(yrow + 32 * j) = pk;
        }
    }
    if constexpr (PROJ) {
        // ---- phase 3: Y[rows, 32 c .. + 32) = h2 W3[chunk c]^T + b3 + res3, ten chunks of 60 MFMAs (the row-panel GEMM's loop with one wave per row group:
        // 5 % of the launch; the chunk epilogue is left to hipcc behind the chunk's MFMAs)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (hipcc has waited for its residual loads already, and with them for the first three items)
#pragma unroll 1
        for (int c = 0; c < FF_K / 32; ++c) {
            hp_wait_dma<2 * FF_PPW>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_item3(c + 3);
            __builtin_amdgcn_sched_barrier(0);
            const char* const slot = smem + (c & (FF_NSLOT - 1)) * FF_SLOTB;
            f32x4 acc3[FF_MF][2];
            auto wread = [&](int i) { return *reinterpret_cast<const u32x4*>(slot + w1off[(i >> 1) & 1] + (i >> 2) * 128 + (i & 1) * 16 * FF_ROWB); };
            wfr[0] = wread(0); wfr[1] = wread(1); wfr[2] = wread(2);
            hp_static_for<0, 60>([&](auto pc) {
                constexpr int P = decltype(pc)::value, i = P / 3, f = P % 3, ks = i >> 1, nf = i & 1;
                if constexpr (f == 0 && i + 3 < 20) wfr[(i + 3) % 6] = wread(i + 3);
                if constexpr (ks == 0) hp_mfma_v0(acc3[f][nf], wfr[i % 6], af[f][ks]);
                else hp_mfma_v(acc3[f][nf], wfr[i % 6], af[f][ks]);
                if constexpr (f == 1 && i >= 1) hp_keep(wfr[(i - 1) % 6]);
                __builtin_amdgcn_sched_barrier(0);
            });
            // asm MFMA results -> VALU; the last two slots' fragments (18, 19 -> wfr[0], wfr[1]) stay sources until the nops are through (the epilogue below
            // depends on this statement's outputs, the next chunk's reads sit behind a scheduling barrier)
            asm volatile("s_nop 15" : "+v"(acc3[0][0]), "+v"(acc3[0][1]), "+v"(acc3[1][0]), "+v"(acc3[1][1]), "+v"(acc3[2][0]), "+v"(acc3[2][1]) : "v"(wfr[0]), "v"(wfr[1]));
            {   // ... and until a VALU instruction has READ the result of the chunk's last MFMA (in-order matrix pipe: everything before it has finished too)
                float pr = acc3[FF_MF - 1][1][3];
                asm volatile("v_mov_b32 %0, %0" : "+v"(pr) : "v"(wfr[0]), "v"(wfr[1]));
                acc3[FF_MF - 1][1][3] = pr;
            }
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sb3 + 32 * c + 8 * g), b1 = *reinterpret_cast<const f32x4*>(sb3 + 32 * c + 8 * g + 4);
#pragma unroll
            for (int f = 0; f < FF_MF; ++f) {
                const int rl = wave * (FF_MF * 16) + 16 * f + l15, row = m0 + 16 * f + l15;
                float o0 = acc3[f][0][0] + b0[0], o1 = acc3[f][0][1] + b0[1], o2 = acc3[f][0][2] + b0[2], o3 = acc3[f][0][3] + b0[3];
                float o4 = acc3[f][1][0] + b1[0], o5 = acc3[f][1][1] + b1[1], o6 = acc3[f][1][2] + b1[2], o7 = acc3[f][1][3] + b1[3];
                if (p.res3) {
                    const u32x4 r = *reinterpret_cast<const u32x4*>(slot + FF_CHUNKB + rl * 64 + (((g + 2 * (rl >> 2)) & 3) << 4));
                    o0 += bf16lo(r.x); o1 += bf16hi(r.x); o2 += bf16lo(r.y); o3 += bf16hi(r.y);
                    o4 += bf16lo(r.z); o5 += bf16hi(r.z); o6 += bf16lo(r.w); o7 += bf16hi(r.w);
                }
                if (row < p.M)
                    *reinterpret_cast<u32x4*>(p.Y + (long)row * p.ldy + 32 * c + 8 * g) = (u32x4){pack_bf16x2(o0, o1), pack_bf16x2(o2, o3), pack_bf16x2(o4, o5), pack_bf16x2(o6, o7)};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

}  // namespace

// 1 when the fused feed-forward kernel covers the shape (the caller otherwise runs the GEGLU projection and ff2 as two GEMMs).  AE_FF_FUSED=0 turns it off (A/B).
extern "C" int ae_ff_fused_supported(int M, int C, int H) {
    static const int on = getenv("AE_FF_FUSED") ? atoi(getenv("AE_FF_FUSED")) : 1;
    static const int any_m = getenv("AE_ROWPANEL_ANY_M") ? atoi(getenv("AE_ROWPANEL_ANY_M")) : 0;
    if (!on) return 0;
    if (!any_m && (M + FF_BM - 1) / FF_BM < 192) return 0;   // one 192-row block per CU (as the row-panel kernel): below 3/4 of the chip the tiled kernels fill it better
    return (C == FF_K && H % 64 == 0 && 2 * H <= FF_MAXH2 && M >= FF_BM) ? 1 : 0;
}

extern "C" int ae_ff_fused_bf16(const void* X, long ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* W1, long ldw1, const float* b1,
                                const void* W2img, const float* b2, const void* residual, long ldr, const void* W3, long ldw3, const float* b3,
                                const void* residual3, long ldr3, float* colstats, void* Y, long ldy, int M, int C, int H, void* stream) {
    AE_REQUIRE(X && W1 && b1 && ln_gamma && ln_beta && W2img && Y, "ae_ff_fused_bf16: null pointer");
    AE_REQUIRE(ae_ff_fused_supported(M, C, H), "ae_ff_fused_bf16: unsupported shape M=%d C=%d H=%d (C must be 320, H %% 64 == 0, H <= 1280)", M, C, H);
    AE_REQUIRE(ln_eps >= 0.f, "ae_ff_fused_bf16: eps");
    AE_REQUIRE(ldx % 8 == 0 && ldw1 % 8 == 0 && ldy % 8 == 0 && (!residual || ldr % 8 == 0), "ae_ff_fused_bf16: row strides must keep 16-byte alignment");
    AE_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)W1 & 15) == 0 && ((uintptr_t)W2img & 15) == 0 && ((uintptr_t)Y & 15) == 0 && ((uintptr_t)residual & 15) == 0,
               "ae_ff_fused_bf16: pointer alignment");
    AE_REQUIRE(W3 || (!b3 && !residual3 && !colstats), "ae_ff_fused_bf16: b3 / residual3 / colstats go with the output projection W3");
    AE_REQUIRE(!W3 || (ldw3 % 8 == 0 && ((uintptr_t)W3 & 15) == 0 && (!residual3 || (ldr3 % 8 == 0 && ((uintptr_t)residual3 & 15) == 0))), "ae_ff_fused_bf16: W3 / residual3 alignment");
    AE_REQUIRE(Y != X && Y != residual3, "ae_ff_fused_bf16: in-place operation (Y aliasing X or residual3) is not supported");
    long ldmax = ldx > ldy ? ldx : ldy;
    if (residual && ldr > ldmax) ldmax = ldr;
    if (residual3 && ldr3 > ldmax) ldmax = ldr3;
    AE_REQUIRE((long)2 * H * ldw1 * 2 < (1L << 30) && ((long)M + FF_BM) * ldmax * 2 < (1L << 30) && (long)C * ldw3 * 2 < (1L << 30),
               "ae_ff_fused_bf16: tensors must stay below 1 GiB (32-bit offsets, out-of-range marker)");
    FFArgs a{};
    a.X = (const bf16_t*)X; a.W1 = (const bf16_t*)W1; a.W2 = (const bf16_t*)W2img; a.Y = (bf16_t*)Y; a.res = (const bf16_t*)residual;
    a.b1 = b1; a.ln_g = ln_gamma; a.ln_b = ln_beta; a.b2 = b2; a.ln_eps = ln_eps; a.M = M; a.H = H;
    a.ldx = ldx; a.ldw1 = ldw1; a.ldy = ldy; a.ldr = ldr;
    a.W3 = (const bf16_t*)W3; a.b3 = b3; a.res3 = (const bf16_t*)residual3; a.ldw3 = ldw3; a.ldr3 = ldr3;
    const dim3 grid((unsigned)((M + FF_BM - 1) / FF_BM)), block(64 * FF_NW);
    if (W3) hipLaunchKernelGGL(ff_fused_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ff_fused_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    const int rc = ae_check_launch("ae_ff_fused_bf16");
    if (rc || !colstats) return rc;
    // per-channel statistics of the output for the GroupNorm that consumes it: a lane stays on ONE row across all columns here, so the column sums come from the
    // stand-alone pass over the (L2-resident) output, as behind ae_ln_gemm_bf16
    AE_REQUIRE(((uintptr_t)colstats & 15) == 0, "ae_ff_fused_bf16: colstats alignment");
    return ae_launch_colstats((const bf16_t*)Y, ldy, M, C, colstats, (hipStream_t)stream);
}
