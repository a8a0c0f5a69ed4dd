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
//   * P stays where the QK^T MFMA left it: the 32x32 accumulator layout of S^T IS the B-operand layout of the same v_mfma_f32_32x32x16_bf16
//     shape, so O^T = V^T P^T needs no cross-lane move (the key order inside a K step is permuted consistently on the V side); O^T is D / 32 + 1
//     blocks of 32 rows — head_dim 40 runs 64 rows, i.e. 7 issued MFMAs per 32x32 logit block where 5 would be the un-padded work (1.40x).
//     (Round 2 measured the alternative, 16x16x32 PV over 48 rows behind four v_permlane16_swap per block: 413 vs 406 us, not kept.);
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
#ifndef AE_ATTN_V_DEFAULT
#define AE_ATTN_V_DEFAULT 7   // default variant of the plain long-sequence kernel (see launch_fast): decided by measurement (round 5: + flag 4, the software-pipelined kernel)
#endif
#ifndef AE_ATTN_FENCE_ALL
#define AE_ATTN_FENCE_ALL 1   // the MFMA-source fences (P registers, `hold`, `srcring`) in every instantiation, not only the two-query-group one
#endif
constexpr float FLOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 8.0f;  // log2 units
constexpr float MASKED = -1.0e30f;

// LDS-DMA, one wave: 64 x 16 B from global (descriptor + per-lane byte offset) straight to LDS at lds_byte + lane * 16.
// Issued through inline asm ON PURPOSE: hipcc treats the builtin form as an LDS store that may alias every later LDS read and
// drains it (s_waitcnt vmcnt(0)) in front of the first ds_read of the SAME tile loop iteration — i.e. it waits for the prefetch of
// tile t + 1 before computing tile t.  The asm form is invisible to that bookkeeping; dma_wait_all() below is the only wait.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, int lds_byte, int voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voffset), "s"(rs), "s"(lds_byte) : "memory", "m0");
#endif
}
// every DMA piece this wave issued has landed, and all of the wave's LDS accesses are done; then the block-wide barrier
__device__ __forceinline__ void dma_wait_all_and_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
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

#ifdef AE_ATTN_LAB
__device__ unsigned long long g_attn_dbg[4];  // lab only: sum of per-block shader cycles, 100 MHz ticks, blocks
#endif

// QG: 32-query column groups per wave (1 or 2).  QG = 2: a wave owns 64 queries; every K / V fragment it reads from LDS feeds two
// MFMAs (half the LDS reads, DMA bytes and barriers per FLOP), and the two groups' softmax chains are independent work the
// scheduler can lay beside each other's MFMAs.
// VSPLIT: the V tile is kept in LDS as one [64 keys][64 B] image per 32-wide d-block plus a narrow image for the remainder
// (head_dim 40: [64][64 B] + [64][16 B]) instead of rows of 2 D bytes: the four key rows a ds_read_b64_tr_b16 lane group touches are
// then 256 contiguous bytes — every bank once (rows of 80 / 160 bytes put rows 0 and 3 on the same banks: 22 % of the LDS
// instruction cycles were bank conflicts, profiles/r02_pmc_attn_fast_d40_b.txt).  The re-arrangement costs nothing: it is the
// per-lane SOURCE offset of the LDS-DMA pieces.
// SKV ("short K/V", cross-attention: 77 text tokens + task token, optionally the expert tokens as second segment): every key of the
// launch fits three tile buffers (two text tiles + one second-segment tile).  They are copied ONCE per block by one LDS-DMA burst
// behind one barrier, and the block then walks p.ng groups of 128 queries over them — no per-tile DMA round trip, no per-tile
// barrier, a quarter of the blocks (the 64x64 level: 3072 blocks in four rounds -> 768 blocks in one).
// NWV / SKT (round 4): waves per block and resident tile buffers of the short-K/V form.  SAM's 14 x 14 windows (196 queries = keys, head dim 80, the
// small-grid rel-pos bias) run it with NWV = 7 (224 query slots: one block per (window, head)) and SKT = 4: the window's 196 keys land by ONE LDS-DMA
// burst behind one barrier, where the tiled form pays a DMA round trip and a barrier per 64-key tile and loads the keys once per 128-query block.
template <int D, int OCC, bool SEG2 = false, int ABL = 0, int BIAS = 0, int QG = 1, bool VSPLIT = false, bool SKV = false, int NWV = 4, int SKT = 3>
__global__ __launch_bounds__(64 * NWV, OCC) void attn_fast_kernel(const AttnArgs p) {
    static_assert(!SKV || (QG == 1 && BIAS != 2 && !VSPLIT), "short-K/V variant: one query group per wave, no row-per-tile bias, row-major V image");
    static_assert(NWV == 4 || (SKV && !SEG2), "other wave counts exist for the short-K/V form only");
    static_assert(!SEG2 || SKT >= 3, "the second segment's tile sits in buffer 2");
    static_assert(D % 8 == 0 && (D <= 96 || D == 160), "head_dim: multiple of 8, <= 96, or 160");
    static_assert(QG == 1 || QG == 2, "one or two 32-query groups per wave");
    static_assert(BIAS == 0 || QG == 1, "the rel-pos variants keep one query group per wave");
    constexpr int KS = (D + 15) / 16;        // K=16 steps of S^T = K Q^T
    constexpr bool QSLOT = (D % 16) != 0;    // contraction slot `D` is free: it carries the softmax offset (K side reads 1.0)
    constexpr int NDB = D / 32 + 1;          // 32-row blocks of O^T; row D is the softmax denominator
    constexpr int LDB = D / 32, LREG = 4 * ((D % 32) / 8);  // block / accumulator register (lanes hi = 0) of that row
    constexpr int ROWB = 2 * D;              // bytes per K / V row in LDS
    constexpr int CH = D / 8;                // 16-byte chunks per row = 1-KiB DMA pieces per 64-row tile
    constexpr int TILEB = FKT * ROWB;
    constexpr int BUFB = 2 * TILEB;          // K tile then V tile
    constexpr int NTB = SKV ? SKT : 2;       // tile buffers (SKV: text tiles 0 / 1 + the second segment's tile; SAM windows: four tiles of keys)
    constexpr int ONES_OFF = NTB * BUFB;     // "ones tile": 64 rows of ROWB bytes, each starting with bf16 {1,0,0,0,0,0,0,0}
    constexpr int NFULL = D / 32, REM = D % 32;          // VSPLIT: full 32-wide d-blocks, remainder columns
    constexpr int RS = REM * 2 > 16 ? REM * 2 : 16;      // VSPLIT: row bytes of the remainder image (and of its ones tile)
    constexpr int RC = REM / 8;                          // 16-byte chunks per row of the remainder image
    constexpr int VONES_OFF = ONES_OFF + TILEB + 64;     // VSPLIT: ones tile with the remainder image's row stride
    constexpr int LDSB = VSPLIT ? VONES_OFF + 64 * RS : ONES_OFF + TILEB + 64;
    constexpr int NPIECE = 2 * CH;
    constexpr int MAXP = (NPIECE + NWV - 1) / NWV;

    __shared__ __attribute__((aligned(16))) char smem[LDSB];

#ifdef AE_ATTN_LAB
    const unsigned long long dbg_c0 = __builtin_readcyclecounter(), dbg_w0 = wall_clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, g = lane >> 4;
    constexpr int QB = 32 * NWV * QG;
    const int nqb = SKV ? (p.Nq + QB * p.ng - 1) / (QB * p.ng) : (p.Nq + QB - 1) / QB;
    const int vb = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const int bh = vb / nqb, qb = vb - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    int q0 = (SKV ? qb * QB * p.ng : qb * QB) + wave * 32 * QG;   // group gq of the wave holds queries q0 + 32 gq .. + 31

    if (tid < FKT) *reinterpret_cast<u32x4*>(smem + ONES_OFF + tid * ROWB) = (u32x4){0x00003F80u, 0u, 0u, 0u};
    if (VSPLIT && tid < FKT) *reinterpret_cast<u32x4*>(smem + VONES_OFF + tid * RS) = (u32x4){0x00003F80u, 0u, 0u, 0u};

    // ---- Q^T operand of the 32x32x16 MFMA: lane (q = l31, hi) holds c * Q[q][16 ks + 8 hi .. +8], zero beyond head_dim
    const bf16_t* qp = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
    const float c = p.scale * FLOG2E;
    u32x4 qf[QG][KS];
    auto load_q = [&]() {
#pragma unroll
    for (int gq = 0; gq < QG; ++gq) {
        const int qrow = min(q0 + 32 * gq + l31, p.Nq - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 16 * ks + 8 * hi;
            u32x4 t = *reinterpret_cast<const u32x4*>(qp + (long)qrow * p.q_sn + (d0 < D ? d0 : 0));
            if (d0 >= D) t = (u32x4){0u, 0u, 0u, 0u};
            qf[gq][ks].x = pack_bf16x2(bf16lo(t.x) * c, bf16hi(t.x) * c);
            qf[gq][ks].y = pack_bf16x2(bf16lo(t.y) * c, bf16hi(t.y) * c);
            qf[gq][ks].z = pack_bf16x2(bf16lo(t.z) * c, bf16hi(t.z) * c);
            qf[gq][ks].w = pack_bf16x2(bf16lo(t.w) * c, bf16hi(t.w) * c);
        }
    }
    };
    load_q();

    // ---- per-lane LDS read addresses (tile-buffer offset added per tile; key-block / K-step / d-block offsets are immediates)
    const int kaddr = l31 * ROWB + hi * 16;
    // V^T operand by ds_read_b64_tr_b16: 16-lane group g reads a [4 keys][16 d] block, lane i of the group addresses row i >> 2,
    // columns 4 (i & 3) .. +4, and receives column i.  Groups 0/1 -> d 0-15 / 16-31 of the lanes' hi = 0 keys, groups 2/3 -> hi = 1.
    const int vrow = 4 * hi + (l15 >> 2);
    const int vaddr = VSPLIT ? TILEB + vrow * 64 + (16 * (g & 1) + 4 * (l15 & 3)) * 2 : TILEB + vrow * ROWB + (16 * (g & 1) + 4 * (l15 & 3)) * 2;
    const int vcol = 32 * LDB + 16 * (g & 1) + 4 * (l15 & 3);  // first of this lane's 4 columns in the last d-block
    const bool ones_lane = vcol == D;  // supplies columns D..D+3: reads {1,0,0,0} instead
    const bool zero_lane = vcol > D;   // padding columns: read zeros (idle multipliers) instead of the neighbouring rows

    f32x16 cinit[QG];  // !QSLOT: the offset enters through the C operand
#pragma unroll
    for (int gq = 0; gq < QG; ++gq)
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[gq][r] = 0.f;
    // Decomposed rel-pos bias (image_encoder.py:325-361), entering S' through the MFMA's C operand:
    // BIAS 2: the key grid has one ROW per 64-key tile (kW == 64: SAM's global attention): rel_w[q, kw] * log2e of this lane's
    //         2 x 16 key columns stays in registers, rel_h[q, kh] is one value per tile — one v_add per logit, no index arithmetic;
    // BIAS 1: small key grids (kH, kW <= 16: SAM's 14 x 14 windows): the block's rows of both tables sit in LDS (times log2e),
    //         every logit looks up rel_h[q, key / kW] + rel_w[q, key % kW].
    constexpr bool BIAS2 = BIAS == 2, BIAS1 = BIAS == 1, BIAS3 = BIAS == 3;
    // BIAS 3 (round 4, SAM windows): the small-grid bias as TWO MORE K STEPS of the S^T MFMA chain.  rel_h[q, kh] + rel_w[q, kw] is an inner
    //         product of the query's 32-vector (rel_h[q, 0..15], rel_w[q, 0..15]) with a one-hot vector of the key (1 at kh and at 16 + kw): the key
    //         side is a table in LDS (64 bytes per key, the same for every window and head), the query side two operand registers per wave.
    //         BIAS 1 spends ~10 VALU / LDS operations per logit on the (kh, kw) index arithmetic and the two look-ups — the windowed launches were
    //         bound by that, not by their DMA round trips (resident keys alone: 48 vs 49 us, profiles/r04_v19_sam_win.txt).  The bias rides in
    //         bf16 here (rel-pos values are O(1): 2^-9 relative, the rounding the logits' own q and k operands carry).
    static_assert(!BIAS3 || (SKV && NTB * FKT * 64 <= 65536), "bias-as-K-steps exists in the resident-keys form");
    __shared__ __attribute__((aligned(16))) char skhot[BIAS3 ? NTB * FKT * 64 : 16];
    u32x4 qrel[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    if (BIAS3) {
        // key table: row `key` = bf16 one-hot of kh in columns 0..15 and of kw in columns 16..31 (zero rows past Nk)
        for (int i = tid; i < NTB * FKT * 4; i += 64 * NWV) {
            const int key = i >> 2, c4 = i & 3;                   // 16-byte chunk c4 of the row: columns 8 c4 .. 8 c4 + 7
            const int kh = key / p.kW, kw = key - kh * p.kW;
            const int hot = key < p.Nk ? (c4 < 2 ? kh : 16 + kw) - 8 * c4 : -1;   // position of the 1 inside this chunk, if any
            u32x4 v = {0u, 0u, 0u, 0u};
            if (hot >= 0 && hot < 8) {
                const uint32_t one = (hot & 1) ? 0x3F800000u : 0x00003F80u;
                if ((hot >> 1) == 0) v.x = one; else if ((hot >> 1) == 1) v.y = one; else if ((hot >> 1) == 2) v.z = one; else v.w = one;
            }
            *reinterpret_cast<u32x4*>(skhot + key * 64 + c4 * 16) = v;
        }
        // query side: lane (q = l31, hi) holds rel_h[q, 8 hi .. 8 hi + 7] (step 0) and rel_w[q, 8 hi .. + 7] (step 1), times log2(e)
        const long qc = (long)bh * p.Nq + min(q0 + l31, p.Nq - 1);
        float rh[8], rw[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            rh[e] = (8 * hi + e < p.kH) ? p.rel_h[qc * p.kH + 8 * hi + e] * FLOG2E : 0.f;
            rw[e] = (8 * hi + e < p.kW) ? p.rel_w[qc * p.kW + 8 * hi + e] * FLOG2E : 0.f;
        }
        qrel[0] = (u32x4){pack_bf16x2(rh[0], rh[1]), pack_bf16x2(rh[2], rh[3]), pack_bf16x2(rh[4], rh[5]), pack_bf16x2(rh[6], rh[7])};
        qrel[1] = (u32x4){pack_bf16x2(rw[0], rw[1]), pack_bf16x2(rw[2], rw[3]), pack_bf16x2(rw[4], rw[5]), pack_bf16x2(rw[6], rw[7])};
    }
    static_assert(BIAS == 0 || (!QSLOT && !SEG2), "the rel-pos variants are built for head dims that are multiples of 16, one segment");
    __shared__ float sbias[BIAS1 ? 32 * NWV * 33 : 1];  // row stride 33: lanes (= queries) reading one column hit 32 different banks
    const float inv_kw = BIAS1 ? 1.0f / (float)p.kW : 0.f;
    if (BIAS1) {
        for (int i = tid; i < 32 * NWV * 32; i += 64 * NWV) {
            const int ql = i >> 5, j = i & 31;
            const long qc = (long)bh * p.Nq + min(qb * QB + ql, p.Nq - 1);
            float v = 0.f;
            if (j < 16) { if (j < p.kH) v = p.rel_h[qc * p.kH + j]; }
            else if (j - 16 < p.kW) v = p.rel_w[qc * p.kW + j - 16];
            sbias[ql * 33 + j] = v * FLOG2E;
        }
        // visible after the first tile's barrier
    }
    f32x16 wb[BIAS2 ? 2 : 1];
    const float* rh_row = nullptr;
    float rh_tile = 0.f;
    if (BIAS2) {
        const long qc = (long)bh * p.Nq + min(q0 + l31, p.Nq - 1);
        const float* row = p.rel_w + qc * p.kW;
        rh_row = p.rel_h + qc * p.kH;
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(row + 32 * b2 + 8 * r4 + 4 * hi);
                wb[BIAS2 ? b2 : 0][4 * r4] = t[0] * FLOG2E; wb[BIAS2 ? b2 : 0][4 * r4 + 1] = t[1] * FLOG2E;
                wb[BIAS2 ? b2 : 0][4 * r4 + 2] = t[2] * FLOG2E; wb[BIAS2 ? b2 : 0][4 * r4 + 3] = t[3] * FLOG2E;
            }
    }
    float mt[QG];  // m~ (log2 units, always bf16-representable): S' = c q.k - m~
    f32x16 o[QG][NDB];
#pragma unroll
    for (int gq = 0; gq < QG; ++gq) {
        mt[gq] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[gq][db][r] = 0.f;
    }

    int seg_nk = p.Nk;
    int kcur = 0, klast = 0, vcur = 0, vlast = 0, khot_row = 0;

    u32x4 hold[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};  // QG = 2: ALL P registers of the previous block's last K-step (see the end of block())
    // The same exposure exists for SrcA (round 4, found by reading the listing: tools/isa_audit.py::mfma_source_overwrites — hipcc reused a V
    // fragment's registers for the other query group's v_exp_f32 results ONE instruction after the MFMA that reads them, and reloads a K fragment
    // by ds_read two instructions after its last MFMA).  The matrix pipe is in order and takes one MFMA at a time, so a fragment is safe once two
    // further MFMAs have been issued: the last two fragments stay live values in this ring until then.
    u32x4 srcring[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    // Round 5: with ONE query group the plain use above is too weak — an input-only asm is not ordered against the MFMAs, and the
    // listings of the head-dim 80 / 160 and SAM-window instantiations showed v_exp_f32 / address VALU writes into a K or V fragment one to four
    // instructions behind the MFMA that reads it (tools/isa_audit.py).  There the retiring statement also takes `acc`, the RESULT of the MFMA that read
    // `frag`, as a read-write operand: it can only sit behind that MFMA's issue, and the fragment of two MFMAs ago stays a live value until then.
    // The two-query-group instantiation (clean listing, tuned schedule) keeps the plain form: its device code is unchanged.
    auto retire_src = [&](bf16x8_t frag, f32x16& acc) __attribute__((always_inline)) {
        union { bf16x8_t b; u32x4 u; } cv;
        cv.b = frag;
        if constexpr (QG == 1 && D <= 96) asm volatile("" : "+v"(acc) : "v"(srcring[0]));
        else if constexpr (QG == 1 && SEG2 && SKV) asm volatile("" : "+a"(acc) : "v"(srcring[0]));   // head dim 160: hipcc keeps S and O in the accumulator file, an "a" tie costs no copies
        else asm volatile("" ::"v"(srcring[0]));
        srcring[0] = srcring[1];
        srcring[1] = cv.u;
    };
    // one 32-key block of the current tile
    auto block = [&](auto blk_tag, int k0, bool first, bool tail) {
        constexpr int B2 = decltype(blk_tag)::value;
        constexpr int BO = B2 * 32 * ROWB;
        // ---- S'^T = K (cQ)^T - m~ : lane holds S'[key = (r&3) + 8 (r>>2) + 4 hi][q = l31] of every query group
        if (ABL == 10 || ABL == 12) __builtin_amdgcn_s_setprio(1);
        if (ABL == 11) __builtin_amdgcn_s_setprio(0);
        f32x16 s[QG];
        if (ABL == 4) {
#pragma unroll
            for (int gq = 0; gq < QG; ++gq)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[gq][r] = -(float)(r + l31) - mt[gq];
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int ka = (QSLOT && ks == KS - 1) ? klast + BO : kcur + BO + ks * 32;
                const bf16x8_t kf = as_bf16x8(*reinterpret_cast<const u32x4*>(smem + ka));   // one K fragment feeds every query group
#pragma unroll
                for (int gq = 0; gq < QG; ++gq) {
                    if (ks == 0 && QSLOT) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        s[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, as_bf16x8(qf[gq][ks]), z, 0, 0, 0);
                    } else if (BIAS2 && ks == 0) {
                        f32x16 c0;
                        const float hb = rh_tile - mt[gq];
#pragma unroll
                        for (int r = 0; r < 16; ++r) c0[r] = wb[BIAS2 ? B2 : 0][r] + hb;
                        s[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, as_bf16x8(qf[gq][ks]), c0, 0, 0, 0);
                    } else if (BIAS1 && ks == 0) {
                        f32x16 c0;
                        const float* row = sbias + (wave * 32 + l31) * 33;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            // key -> (row, column) of the key grid without an integer division: (key + 0.5) / kW is never closer than
                            // 0.5 / kW to an integer, far above the fp32 error for a grid of at most 16 x 16
                            const int key = min(k0 + B2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, seg_nk - 1);
                            const int kh = (int)(((float)key + 0.5f) * inv_kw);
                            c0[r] = row[kh] + row[16 + key - kh * p.kW] - mt[gq];
                        }
                        s[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, as_bf16x8(qf[gq][ks]), c0, 0, 0, 0);
                    } else {
                        s[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, as_bf16x8(qf[gq][ks]), ks == 0 ? cinit[gq] : s[gq], 0, 0, 0);
                    }
                }
                retire_src(kf, s[QG - 1]);
            }
            if (BIAS3) {
                // (tile buffer index = kcur's buffer: the key row inside the resident image is buffer * 64 + B2 * 32 + l31)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    const bf16x8_t hf = as_bf16x8(*reinterpret_cast<const u32x4*>(skhot + (khot_row + B2 * 32) * 64 + hi * 16 + kb * 32));
#pragma unroll
                    for (int gq = 0; gq < QG; ++gq) s[gq] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hf, as_bf16x8(qrel[kb]), s[gq], 0, 0, 0);
                    retire_src(hf, s[QG - 1]);
                }
            }
            // the P registers of the previous block's last K-step (`hold`) are free from here on: six more MFMAs are in the pipe behind
            // the one that read them
            if constexpr (QG == 1 && D <= 96) { if (AE_ATTN_FENCE_ALL) asm volatile("" : "+v"(s[0]) : "v"(hold[0]), "v"(hold[1])); }   // behind the last logit MFMA's issue (see retire_src)
            else asm volatile("" ::"v"(hold[0]), "v"(hold[1]));
        }
        if (ABL == 10 || ABL == 12) __builtin_amdgcn_s_setprio(0);
        if (ABL == 11) __builtin_amdgcn_s_setprio(1);
        uint32_t pk[QG][8];
        float mx[QG];
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) {
            if (tail) {  // keys past Nk (zero rows from the bounds-checked DMA) must not count
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + B2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= seg_nk) s[gq][r] = MASKED;
            }
            mx[gq] = fmaxf(fmaxf(s[gq][0], s[gq][1]), s[gq][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx[gq] = fmaxf(fmaxf(mx[gq], s[gq][r]), s[gq][r + 1]);
            mx[gq] = fmaxf(mx[gq], s[gq][15]);
            if (ABL == 2) mx[gq] = s[gq][0];
        }
        // ONE rarely taken wave-uniform branch for all query groups: the common path stays a single straight-line block in which the
        // groups' exp2 / convert work and the other group's MFMAs are independent instructions
        if (__builtin_expect(first || __any((QG == 2 ? fmaxf(mx[0], mx[QG - 1]) : mx[0]) > RESCALE_THR), 0)) {
#pragma unroll
            for (int gq = 0; gq < QG; ++gq) {
                // rebase m~ (rare): rows whose block maximum is above the offset move it up to that maximum, rounded up to the next
                // bf16-representable value (the offset must survive the trip through the Q operand; 2^-(step) is exact either way);
                // the first block sets it whatever its sign.  O (with its denominator row) follows.  A group that did not trigger
                // keeps its offset unless its own maximum is positive (then it moves by an exact integer step too: harmless).
                const float m2 = fmaxf(mx[gq], __shfl_xor(mx[gq], 32, 64));
                float tgt = (first || m2 > 0.f) ? mt[gq] + __builtin_ceilf(m2) : mt[gq];
                tgt = fmaxf(tgt, -1.0e4f);
                uint32_t tb = __float_as_uint(tgt);
                tb = (tgt > 0.f) ? ((tb + 0xFFFFu) & 0xFFFF0000u) : (tb & 0xFFFF0000u);  // towards +inf
                const float mnew = __uint_as_float(tb);
                const float d = mnew - mt[gq];
                mt[gq] = mnew;
                if (QSLOT) {
                    if (hi) qf[gq][KS - 1].x = (qf[gq][KS - 1].x & 0xFFFF0000u) | ((tb >> 16) ^ 0x8000u);  // slot D holds -m~
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) cinit[gq][r] = -mt[gq];
                }
                const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                for (int r = 0; r < 16; ++r) s[gq][r] -= d;
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[gq][db][r] *= alpha;
            }
        }
        // ---- P = exp2(S'), bf16: pk[4 kk .. 4 kk + 3] is the B operand of K-step kk (keys 16 kk + 4 hi + {0..3, 8..11})
#pragma unroll
        for (int gq = 0; gq < QG; ++gq)
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[gq][j] = (ABL == 1) ? pack_bf16x2(s[gq][2 * j], s[gq][2 * j + 1]) : pack_bf16x2(__builtin_amdgcn_exp2f(s[gq][2 * j]), __builtin_amdgcn_exp2f(s[gq][2 * j + 1]));
        // ---- O^T += V^T P^T : lane holds O^T[d = 32 db + (r&3) + 8 (r>>2) + 4 hi][q = l31]; one V fragment feeds every query group
        if (ABL == 10) __builtin_amdgcn_s_setprio(1);
        if (ABL == 11) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                int va, vstep;
                if (VSPLIT) {
                    va = (db == LDB) ? vlast + B2 * 32 * RS + kk * 16 * RS : vcur + db * 4096 + B2 * 32 * 64 + kk * 16 * 64;
                    vstep = (db == LDB) ? 8 * RS : 8 * 64;
                } else {
                    va = (db == LDB ? vlast : vcur + db * 64) + BO + kk * 16 * ROWB;
                    vstep = 8 * ROWB;
                }
                if (ABL == 3) {  // lab: no V reads, no PV MFMA (P kept alive)
#pragma unroll
                    for (int gq = 0; gq < QG; ++gq) asm volatile("" ::"v"(pk[gq][4 * kk]), "v"(pk[gq][4 * kk + 3]));
                    continue;
                }
                bf16x8_t vf;
                if (ABL == 5) vf = as_bf16x8(qf[0][0]);  // lab: no V reads
                else vf = cat_tr(lds_tr16(smem + va), lds_tr16(smem + va + vstep));
#pragma unroll
                for (int gq = 0; gq < QG; ++gq) {
                    const bf16x8_t pb = as_bf16x8((u32x4){pk[gq][4 * kk], pk[gq][4 * kk + 1], pk[gq][4 * kk + 2], pk[gq][4 * kk + 3]});
                    o[gq][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb, o[gq][db], 0, 0, 0);
                }
                retire_src(vf, o[QG - 1][db]);
            }
        }
        if (AE_ATTN_FENCE_ALL || QG > 1) {
            // gfx950 / hipcc hazard (found with tools/diag_attn.py, profiles/r03_attn_qg2_hazard.txt): with two query groups the
            // scheduler lays group B's exp2 / convert work between group A's PV MFMAs and REUSES A's P registers for it — a VALU
            // write to an MFMA's SrcB registers one instruction after that v_mfma_f32_32x32x16_bf16 was issued.  The MFMA has not
            // finished reading the operand by then: lanes 16-31 / 48-63 (query columns 16-31) pick up the new value, run-to-run
            // different.  hipcc inserts no wait states for this write-after-read.  So every P register of the block stays a live
            // value until the block's last MFMA has been issued (these empty asm statements are uses), and the last K-step's until
            // the next block's QK^T MFMAs are out (`hold`).
            if constexpr (QG == 1 && D <= 96) {
                if (ABL != 3) asm volatile("" : "+v"(o[0][NDB - 1]) : "v"(pk[0][0]), "v"(pk[0][1]), "v"(pk[0][2]), "v"(pk[0][3]), "v"(pk[0][4]), "v"(pk[0][5]), "v"(pk[0][6]), "v"(pk[0][7]));   // behind the block's last MFMA's issue
            } else {
#pragma unroll
            for (int gq = 0; gq < QG; ++gq)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(pk[gq][j]));
            }
            hold[0] = (u32x4){pk[0][4], pk[0][5], pk[0][6], pk[0][7]};
            hold[1] = (u32x4){pk[QG - 1][4], pk[QG - 1][5], pk[QG - 1][6], pk[QG - 1][7]};
        }
    };

    f32x16 o_first[SEG2 ? QG : 1][SEG2 ? NDB : 1];  // SEG2: normalised result of the first segment while the second runs
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);  // LDS byte address of the tile buffers
    constexpr int NSEG = SEG2 ? 2 : 1;

    // one 64-key tile sitting in tile buffer `buf`: LDS read addresses, then its one or two 32-key blocks
    auto tile = [&](int t, int buf) {
        const int boff = buf * BUFB;
        khot_row = buf * FKT + l31;
        kcur = kaddr + boff;
        klast = (QSLOT && hi) ? ONES_OFF + l31 * ROWB : kcur + (KS - 1) * 32;
        vcur = vaddr + boff;
        if (VSPLIT)
            vlast = ones_lane ? VONES_OFF + vrow * RS : (zero_lane ? VONES_OFF + vrow * RS + 8 : TILEB + boff + NFULL * 4096 + vrow * RS + (16 * (g & 1) + 4 * (l15 & 3)) * 2);
        else
            vlast = ones_lane ? ONES_OFF + vrow * ROWB : (zero_lane ? ONES_OFF + vrow * ROWB + 8 : vcur + LDB * 64);
        const bool tail = (t + 1) * FKT > seg_nk;
        if (BIAS2) rh_tile = rh_row[t] * FLOG2E;
        block(std::integral_constant<int, 0>{}, t * FKT, t == 0, tail);
        if (t * FKT + 32 < seg_nk) block(std::integral_constant<int, 1>{}, t * FKT, false, tail);
    };
    // SEG2, end of the first segment: park its normalised output, restart the online softmax
    auto park = [&]() {
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) {
            const float l0 = __shfl(o[gq][LDB][LREG], l31, 64);
            const float inv0 = 1.0f / l0;
            const int qr = q0 + 32 * gq + l31;
            // log2-domain log-sum-exp of the first segment (kept for ae_attn_bwd_bf16): offset + log2(denominator)
            if (p.lse && hi == 0 && qr < p.Nq) p.lse[((long)b * p.H + h) * p.Nq + qr] = mt[gq] + __builtin_amdgcn_logf(l0);
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o_first[SEG2 ? gq : 0][SEG2 ? db : 0][r] = o[gq][db][r] * inv0;
                    o[gq][db][r] = 0.f;
                }
            mt[gq] = 0.f;
            if (QSLOT) {
                if (hi) qf[gq][KS - 1].x &= 0xFFFF0000u;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) cinit[gq][r] = 0.f;
            }
        }
    };
    // normalise and store: 4 consecutive d per lane -> 8-byte stores
    bf16_t* op = p.o + (long)b * p.o_sb + (long)h * p.o_sh;
    auto store_o = [&]() {
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) {
            const float lsum = __shfl(o[gq][LDB][LREG], l31, 64);
            const float inv = (SEG2 ? p.scale2[b] : (p.out_scale ? p.out_scale[b] : 1.0f)) / lsum;
            const int qrow = q0 + 32 * gq + l31;
            {
                float* lse_out = SEG2 ? p.lse2 : p.lse;  // v_log_f32 = log2
                if (lse_out && hi == 0 && qrow < p.Nq) lse_out[((long)b * p.H + h) * p.Nq + qrow] = mt[gq] + __builtin_amdgcn_logf(lsum);
            }
            if (qrow < p.Nq) {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int d = 32 * db + 8 * r4 + 4 * hi;
                        if (d < D) {
                            float r0 = o[gq][db][4 * r4] * inv, r1 = o[gq][db][4 * r4 + 1] * inv, r2 = o[gq][db][4 * r4 + 2] * inv, r3 = o[gq][db][4 * r4 + 3] * inv;
                            if (SEG2) {
                                r0 += o_first[SEG2 ? gq : 0][SEG2 ? db : 0][4 * r4]; r1 += o_first[SEG2 ? gq : 0][SEG2 ? db : 0][4 * r4 + 1];
                                r2 += o_first[SEG2 ? gq : 0][SEG2 ? db : 0][4 * r4 + 2]; r3 += o_first[SEG2 ? gq : 0][SEG2 ? db : 0][4 * r4 + 3];
                            }
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
    };

    for (int seg = 0; seg < NSEG; ++seg) {
        // ---- LDS-DMA plan: piece j of a tile (K pieces 0..CH-1, V pieces CH..2CH-1) is issued by wave j % NWV
        const bf16_t* kp = seg == 0 ? p.k + (long)b * p.k_sb + (long)h * p.k_sh : p.k2 + (long)b * p.k2_sb + (long)h * p.k2_sh;
        const bf16_t* vp = seg == 0 ? p.v + (long)b * p.v_sb + (long)h * p.v_sh : p.v2 + (long)b * p.v2_sb + (long)h * p.v2_sh;
        seg_nk = seg == 0 ? p.Nk : p.Nk2;
        const int ksn2 = (int)(seg == 0 ? p.k_sn : p.k2_sn) * 2, vsn2 = (int)(seg == 0 ? p.v_sn : p.v2_sn) * 2;
        // bounds-checked: rows >= Nk of a tile read as zeros
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kp), 0, (seg_nk - 1) * ksn2 + D * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vp), 0, (seg_nk - 1) * vsn2 + D * 2, 0x00020000);
        int voff[MAXP];
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + NWV * i;
            const bool isK = j < CH;
            const int cidx = (isK ? j : j - CH) * 64 + lane;
            int row = cidx / CH, cc = cidx - row * CH;
            if (VSPLIT && !isK) {  // chunk cidx of the V image: d-block images [64][4 chunks] first, then the remainder image [64][RC]
                if (cidx < NFULL * 256) { row = (cidx & 255) >> 2; cc = (cidx >> 8) * 4 + (cidx & 3); }
                else { const int c2 = cidx - NFULL * 256; row = c2 / (RC > 0 ? RC : 1); cc = NFULL * 4 + (c2 - row * (RC > 0 ? RC : 1)); }
            }
            voff[i] = row * (isK ? ksn2 : vsn2) + cc * 16;
        }
        auto issue = [&](int t, int buf) {
            const int boff = lds0 + buf * BUFB;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int j = wave + NWV * i;
                if (j < NPIECE) {
                    if (j < CH) dma16(rsK, boff + j * 1024, voff[i] + t * FKT * ksn2);
                    else dma16(rsV, boff + TILEB + (j - CH) * 1024, voff[i] + t * FKT * vsn2);
                }
            }
        };
        const int ntiles = (seg_nk + FKT - 1) / FKT;
        if (SKV) {  // every tile of every segment goes out in one burst (text tiles -> buffers 0 / 1, second segment -> buffer 2)
#pragma unroll 1
            for (int t = 0; t < ntiles; ++t) issue(t, seg == 0 ? t : 2);
            continue;
        }
        if (SEG2 && seg == 1) dma_wait_all_and_barrier();  // every wave is done with the first segment's last tile
        issue(0, 0);
        for (int t = 0; t < ntiles; ++t) {
            dma_wait_all_and_barrier();  // tile t has landed (every wave's pieces); every wave is done with tile t - 1
            if (t + 1 < ntiles) issue(t + 1, (t + 1) & 1);
            tile(t, t & 1);
        }
        // the last block's P registers stay live past the loop: park() / store_o() start with VALU work, and the last PV MFMAs (issued a few
        // instructions ago) may still be reading their SrcB operands (see the end of block(); tests/test_isa_static.py checks the listing for it)
        if (AE_ATTN_FENCE_ALL || QG > 1) asm volatile("" ::"v"(hold[0]), "v"(hold[1]));
        asm volatile("" ::"v"(srcring[0]), "v"(srcring[1]));
        if (SEG2 && seg == 0) park();
    }
    if (SKV) {
        dma_wait_all_and_barrier();  // every key of the launch is in LDS; nothing below writes LDS or waits for a load again
        const int nt0 = (p.Nk + FKT - 1) / FKT;
        u32x4 qraw[KS];  // the NEXT group's Q rows, in flight while this group is computed
        for (int gi = 0; gi < p.ng; ++gi) {
            if (gi) {  // next 128-query group of the block: new Q operand, fresh softmax state
                q0 += 32 * NWV;
                if (q0 - wave * 32 >= p.Nq) break;  // block-uniform: the group lies past the last query
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 t = (16 * ks + 8 * hi < D) ? qraw[ks] : (u32x4){0u, 0u, 0u, 0u};
                    qf[0][ks].x = pack_bf16x2(bf16lo(t.x) * c, bf16hi(t.x) * c);
                    qf[0][ks].y = pack_bf16x2(bf16lo(t.y) * c, bf16hi(t.y) * c);
                    qf[0][ks].z = pack_bf16x2(bf16lo(t.z) * c, bf16hi(t.z) * c);
                    qf[0][ks].w = pack_bf16x2(bf16lo(t.w) * c, bf16hi(t.w) * c);
                }
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[0][db][r] = 0.f;
                mt[0] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) cinit[0][r] = 0.f;
            }
            if (gi + 1 < p.ng) {  // (rows past Nq are clamped; their group is never computed)
                const int qrow = min(q0 + 32 * NWV + l31, p.Nq - 1);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int d0 = 16 * ks + 8 * hi;
                    qraw[ks] = *reinterpret_cast<const u32x4*>(qp + (long)qrow * p.q_sn + (d0 < D ? d0 : 0));
                }
            }
            seg_nk = p.Nk;
            for (int t = 0; t < nt0; ++t) tile(t, t);
            // the same MFMA-source fences as behind the tiled loop below: park() / store_o() open with VALU work while the last PV MFMAs may
            // still be reading their P / V fragment registers (tests/test_isa_static.py checks every instantiation's listing)
            if (AE_ATTN_FENCE_ALL || QG > 1) asm volatile("" ::"v"(hold[0]), "v"(hold[1]));
            asm volatile("" ::"v"(srcring[0]), "v"(srcring[1]));
            if (SEG2) {
                park();
                seg_nk = p.Nk2;
                tile(0, 2);
                if (AE_ATTN_FENCE_ALL || QG > 1) asm volatile("" ::"v"(hold[0]), "v"(hold[1]));
                asm volatile("" ::"v"(srcring[0]), "v"(srcring[1]));
            }
            store_o();
        }
        return;
    }

#ifdef AE_ATTN_LAB
    if (tid == 0 && p.kW == 777) {
        atomicAdd(&g_attn_dbg[0], __builtin_readcyclecounter() - dbg_c0);
        atomicAdd(&g_attn_dbg[1], wall_clock64() - dbg_w0);
        atomicAdd(&g_attn_dbg[2], 1ull);
    }
#endif
    if (AE_ATTN_FENCE_ALL || QG > 1) asm volatile("" ::"v"(hold[0]), "v"(hold[1]));
    asm volatile("" ::"v"(srcring[0]), "v"(srcring[1]));
    store_o();
}


// ---------------------------------------------------------------------------------------------------------------------------------------------------
// Round 5: the SOFTWARE-PIPELINED form of the long-sequence kernel (head dim 40, two query groups per wave, VSPLIT images).
//
// What round 4's counters said about attn_fast_kernel<40, ..., QG = 2> (profiles/r04_final3_pmc_attn.txt): the matrix pipe is busy 57 % of the
// SIMD cycles, the VALU port ~60 %, and the sum of the two per 32-key block (7 x 32 MFMA cycles + ~186 VALU cycles per query group) IS the
// measured block time — inside one wave the logit MFMAs, the maximum, exp2 / convert and the PV MFMAs of a block form one dependent chain, and
// three such chains per SIMD interleave only by accident.  The structure below makes the overlap a property of the instruction stream of ONE
// wave: while block i's probabilities are exponentiated (VALU), block i + 1's logits are multiplied (MFMA); while block i's PV products run
// (MFMA), block i + 1's maximum is taken (VALU):
//
//     step(i):   S'(i+1) = K(i+1) Q^T          6 MFMAs   ||   P(i) = exp2(S'(i)), first halves             16 v_exp + 8 v_cvt_pk
//                O += V(i) P(i), K-step 0      4 MFMAs   ||   P(i), second halves                            16 v_exp + 8 v_cvt_pk
//                O += V(i) P(i), K-step 1      4 MFMAs   ||   maximum of S'(i+1), the rebase decision        16 v_max3 + compare
//
// Every number the wave computes is the one attn_fast_kernel computes, in the same order per accumulator (the rebase of block i + 1 still sees an O
// that holds the PV products up to block i, S'(i+1) is still corrected by the same exact step, the offset still rides in Q's free slot):
// the outputs are BIT-IDENTICAL to the QG = 2 kernel's (tests/test_hip_ops.py::test_attention_pipelined_equals_two_group_kernel).
// Cost: two logit blocks live at once (64 registers instead of 32) -> 2 waves per SIMD instead of 3; three tile buffers instead of two, so that
// the logits of the NEXT tile's first block can be multiplied while the current tile's second block is still being reduced (one barrier per
// 64-key tile as before: tile t + 2 is issued into the buffer of tile t - 1 behind the barrier that proves every wave has left tile t - 1).
// Envelope: head dim 40, Nk a multiple of 64 and >= 128, no bias / mask / second segment (everything else: attn_fast_kernel).
// KT: keys per LDS tile (64: rounds 5's form; 128, round 6: half the block barriers — DESIGN 7.00c priced the two barriers' worth of waiting per 64-key tile at
// 16 % of the wave cycles — for 74 KiB instead of 31 KiB of LDS per block, still two blocks per CU).  Same MFMAs on the same operands in the same order: bit-identical.
// PV16 (round 6): the PV products as v_mfma_f32_16x16x32_bf16 on 48 output rows (d = 40 real, row 40 the denominator, 7 rows of zeros) instead of
// 32x32x16 on 64 rows: 12 MFMAs of 16 cycles per step and query-group pair instead of 8 of 32.  The probabilities leave the logit MFMAs as (query = lane & 31,
// keys by lane >> 5); the B operand of the 16-wide shape wants (query = lane & 15, keys by lane >> 4): one v_permlane16_swap per register between the two
// 16-key halves does exactly that exchange (lane bit 4 <-> key half), and the key order inside the contraction is free as long as the V^T fragments follow it.
template <int D, int KT, bool PV16 = false>
__global__ __launch_bounds__(256, 2) void attn_pipe_kernel(const AttnArgs p) {
    static_assert(KT == 64 || KT == 128, "keys per tile");
    constexpr int NBLK = KT / 32, VDB = KT * 64, KP = KT * (D / 8) / 64;   // 32-key blocks per tile; bytes of a V image per 32-wide d-block; 1-KiB pieces of a K (or V) tile
    static_assert(D == 40, "pipelined kernel: head dim 40 (K steps 3 with a free slot, two 32-row output blocks)");
    constexpr int QG = 2;
    constexpr int KS = (D + 15) / 16;
    constexpr int NDB = D / 32 + 1, LDB = D / 32, LREG = 4 * ((D % 32) / 8);
    constexpr int ROWB = 2 * D, CH = D / 8, TILEB = KT * ROWB, BUFB = 2 * TILEB, NTB = 3;
    constexpr int ONES_OFF = NTB * BUFB;
    constexpr int NFULL = D / 32, REM = D % 32, RS = REM * 2 > 16 ? REM * 2 : 16, RC = REM / 8;
    constexpr int VONES_OFF = ONES_OFF + TILEB + 64;
    constexpr int LDSB = VONES_OFF + KT * RS;
    constexpr int NPIECE = 2 * KP, NWV = 4, MAXP = (NPIECE + NWV - 1) / NWV;
    __shared__ __attribute__((aligned(16))) char smem[LDSB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5, l15 = lane & 15, g = lane >> 4;
    constexpr int QB = 32 * NWV * QG;
    const int nqb = (p.Nq + QB - 1) / QB;
    const int vb = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const int bh = vb / nqb, qb = vb - bh * nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q0 = qb * QB + wave * 32 * QG;

    if (tid < KT) *reinterpret_cast<u32x4*>(smem + ONES_OFF + tid * ROWB) = (u32x4){0x00003F80u, 0u, 0u, 0u};
    if (tid < KT) *reinterpret_cast<u32x4*>(smem + VONES_OFF + tid * RS) = (u32x4){0x00003F80u, 0u, 0u, 0u};

    // Q^T operand: lane (q = l31, hi) holds c Q[q][16 ks + 8 hi .. + 8], zero beyond head_dim (slot D carries -m~ later)
    const bf16_t* qp = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
    const float c = p.scale * FLOG2E;
    u32x4 qf[QG][KS];
#pragma unroll
    for (int gq = 0; gq < QG; ++gq) {
        const int qrow = min(q0 + 32 * gq + l31, p.Nq - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = 16 * ks + 8 * hi;
            u32x4 t = *reinterpret_cast<const u32x4*>(qp + (long)qrow * p.q_sn + (d0 < D ? d0 : 0));
            if (d0 >= D) t = (u32x4){0u, 0u, 0u, 0u};
            qf[gq][ks].x = pack_bf16x2(bf16lo(t.x) * c, bf16hi(t.x) * c);
            qf[gq][ks].y = pack_bf16x2(bf16lo(t.y) * c, bf16hi(t.y) * c);
            qf[gq][ks].z = pack_bf16x2(bf16lo(t.z) * c, bf16hi(t.z) * c);
            qf[gq][ks].w = pack_bf16x2(bf16lo(t.w) * c, bf16hi(t.w) * c);
        }
    }

    // per-lane LDS addresses relative to a tile buffer (see attn_fast_kernel: same images, same fragment maps)
    const int kaddr = l31 * ROWB + hi * 16;
    const int klast0 = hi ? ONES_OFF + l31 * ROWB : kaddr + (KS - 1) * 32;   // K step 2: the lanes of the padding columns read the ones tile (column D = 1.0)
    const int kl_mask = hi ? 0 : -1;                                          // ... which does not move with the tile buffer
    const int vrow = 4 * hi + (l15 >> 2);
    const int vaddr = TILEB + vrow * 64 + (16 * (g & 1) + 4 * (l15 & 3)) * 2;
    const int vcol = 32 * LDB + 16 * (g & 1) + 4 * (l15 & 3);
    const bool ones_lane = vcol == D, zero_lane = vcol > D;
    const int vlast0 = ones_lane ? VONES_OFF + vrow * RS : (zero_lane ? VONES_OFF + vrow * RS + 8 : TILEB + NFULL * VDB + vrow * RS + (16 * (g & 1) + 4 * (l15 & 3)) * 2);
    const int vl_mask = (ones_lane || zero_lane) ? 0 : -1;
    // PV16: 16-lane group g reads the [4 keys][16 d] blocks of its key group — keys 16 (g & 1) + 4 (g >> 1) + {0..3} and + 8 (the order the swapped probabilities have)
    const int vrow16 = 16 * (g & 1) + 4 * (g >> 1) + (l15 >> 2);
    const int vaddr16 = TILEB + vrow16 * 64 + 8 * (l15 & 3);
    const bool ones16 = (l15 & 3) == 2, zero16 = (l15 & 3) == 3;   // third 16-row block = d 32..47: columns 40..43 read {1,0,0,0}, 44..47 zeros
    const int vlast16 = ones16 ? VONES_OFF + vrow16 * RS : (zero16 ? VONES_OFF + vrow16 * RS + 8 : TILEB + NFULL * VDB + vrow16 * RS + 8 * (l15 & 3));
    const int vl16_mask = (ones16 || zero16) ? 0 : -1;
    static_assert(!PV16 || (D == 40 && NFULL == 1 && RS == 16), "PV16 is written out for head dim 40");

    float mt[QG];
    f32x16 o[QG][NDB];
    f32x4 o16[QG][2][3];   // PV16: [query group][query half][16-row block of O^T]; lane (l15, g) holds rows 4 g .. 4 g + 3 of query 16 half + l15
#pragma unroll
    for (int gq = 0; gq < QG; ++gq) {
        mt[gq] = 0.f;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[gq][db][r] = 0.f;
#pragma unroll
        for (int qh = 0; qh < 2; ++qh)
#pragma unroll
            for (int db3 = 0; db3 < 3; ++db3) o16[gq][qh][db3] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // ---- LDS-DMA plan (as attn_fast_kernel, VSPLIT): piece j of a tile (K pieces 0..CH-1, V pieces CH..2CH-1) is issued by wave j % 4
    const bf16_t* kp = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
    const bf16_t* vp = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
    const int ksn2 = (int)p.k_sn * 2, vsn2 = (int)p.v_sn * 2;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(kp), 0, (p.Nk - 1) * ksn2 + D * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(vp), 0, (p.Nk - 1) * vsn2 + D * 2, 0x00020000);
    int voff[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int j = wave + NWV * i;
        const bool isK = j < KP;
        const int cidx = (isK ? j : j - KP) * 64 + lane;
        int row = cidx / CH, cc = cidx - row * CH;
        if (!isK) {
            if (cidx < NFULL * KT * 4) { row = (cidx & (KT * 4 - 1)) >> 2; cc = (cidx / (KT * 4)) * 4 + (cidx & 3); }
            else { const int c2 = cidx - NFULL * KT * 4; row = c2 / (RC > 0 ? RC : 1); cc = NFULL * 4 + (c2 - row * (RC > 0 ? RC : 1)); }
        }
        voff[i] = row * (isK ? ksn2 : vsn2) + cc * 16;
    }
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
    auto issue = [&](int t, int boff) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int j = wave + NWV * i;
            if (j < NPIECE) {
                if (j < KP) dma16(rsK, lds0 + boff + j * 1024, voff[i] + t * KT * ksn2);
                else dma16(rsV, lds0 + boff + TILEB + (j - KP) * 1024, voff[i] + t * KT * vsn2);
            }
        }
    };

    // ---- the pieces of a step ------------------------------------------------------------------------------------------------------------------------
    // Source lifetimes.  gfx950 does not interlock a VALU write to a register an issued MFMA still reads as SrcA / SrcB (attn_fast_kernel, round 3), and
    // in a stream this dense hipcc reuses a fragment's registers for exp2 results right behind the MFMA (186 such writes in the first listing of
    // this kernel, tools/isa_audit.py).  Here it is excluded BY CONSTRUCTION: the 14 MFMAs of a step are chained into one program order by empty asm
    // statements that take the result of MFMA k as an input and the accumulator of MFMA k + 1 as a read-write operand (no instruction, no wait: the
    // statement can only sit between the two issues), and each statement also names, as inputs, the fragments whose last reader was issued two
    // MFMAs earlier — they stay live values exactly until the in-order matrix pipe is done with them.  The sources of a step's last two MFMAs are
    // carried into the next step (`hv`, `hp`) and retired behind its first two.  tests/test_isa_static.py reads the listing.
    // (device pass only: on the host pass the x86 meaning of the "v" constraint rejects 512-bit operands, and clang then drops the kernel's host stub
    // without a diagnostic — the library linked and failed to load with an undefined __device_stub__ symbol)
#if defined(__HIP_DEVICE_COMPILE__)
#define AE_TIE0(nxt, prev) asm volatile("" : "+v"(nxt) : "v"(prev))
#define AE_TIE1(nxt, prev, a) asm volatile("" : "+v"(nxt) : "v"(prev), "v"(a))
#define AE_TIE2(nxt, prev, a, b) asm volatile("" : "+v"(nxt) : "v"(prev), "v"(a), "v"(b))
#else
#define AE_TIE0(nxt, prev)
#define AE_TIE1(nxt, prev, a)
#define AE_TIE2(nxt, prev, a, b)
#endif
    u32x4 hv = {0u, 0u, 0u, 0u}, hp[QG] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};   // V fragment / P registers of the previous step's last two MFMAs
    auto rdK = [&](int byte) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(smem + byte); };
    // block maximum of both groups and the (rare) rebase: m~ moves up by an exact bf16-representable step, S' and O follow (see attn_fast_kernel).
    // `k1`, `k2`: the K fragments of the last logit MFMAs — live until the maximum has READ the logits (an instruction reading an MFMA's result
    // is issued only when that MFMA, and with the in-order pipe every earlier one, has finished).
    auto decide = [&](f32x16 (&sn)[QG], bool first, u32x4 k1, u32x4 k2) __attribute__((always_inline)) {
        float mx[QG];
#pragma unroll
        for (int gq = 0; gq < QG; ++gq) {
            mx[gq] = fmaxf(fmaxf(sn[gq][0], sn[gq][1]), sn[gq][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx[gq] = fmaxf(fmaxf(mx[gq], sn[gq][r]), sn[gq][r + 1]);
            mx[gq] = fmaxf(mx[gq], sn[gq][15]);
        }
        asm volatile("" : "+v"(mx[0]), "+v"(mx[1]) : "v"(k1), "v"(k2));
        if (__builtin_expect(first || __any(fmaxf(mx[0], mx[QG - 1]) > RESCALE_THR), 0)) {
#pragma unroll
            for (int gq = 0; gq < QG; ++gq) {
                const float m2 = fmaxf(mx[gq], __shfl_xor(mx[gq], 32, 64));
                float tgt = (first || m2 > 0.f) ? mt[gq] + __builtin_ceilf(m2) : mt[gq];
                tgt = fmaxf(tgt, -1.0e4f);
                uint32_t tb = __float_as_uint(tgt);
                tb = (tgt > 0.f) ? ((tb + 0xFFFFu) & 0xFFFF0000u) : (tb & 0xFFFF0000u);
                const float mnew = __uint_as_float(tb);
                const float d = mnew - mt[gq];
                mt[gq] = mnew;
                if (hi) qf[gq][KS - 1].x = (qf[gq][KS - 1].x & 0xFFFF0000u) | ((tb >> 16) ^ 0x8000u);
                const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
                for (int r = 0; r < 16; ++r) sn[gq][r] -= d;
                if constexpr (PV16) {
#pragma unroll
                    for (int qh = 0; qh < 2; ++qh) {
                        const float aq = __shfl(alpha, 16 * qh + l15, 64);   // the factor of THIS lane's query in the 16-wide layout
#pragma unroll
                        for (int db3 = 0; db3 < 3; ++db3) o16[gq][qh][db3] *= aq;
                    }
                } else {
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[gq][db][r] *= alpha;
                }
            }
        }
    };
    // One pipeline step.  The block (tile buffer `boff`, half B2) whose final logits are `sc` is exponentiated and multiplied into O (CUR) while the
    // logits of the block (buffer `nboff`, half NB2) are produced into `sn`, reduced and (rarely) rebased (NEXT).  The prologue runs NEXT alone.
    auto step = [&](auto blk_tag, auto next_tag, auto has_cur_tag, auto has_next_tag, int boff, int nboff, f32x16 (&sc)[QG], f32x16 (&sn)[QG], bool first) __attribute__((always_inline)) {
        constexpr int B2 = decltype(blk_tag)::value, NB2 = decltype(next_tag)::value;
        constexpr bool CUR = decltype(has_cur_tag)::value, NEXT = decltype(has_next_tag)::value;
        static_assert(NDB == 2 && KS == 3, "the MFMA order below is written out for head dim 40");
        u32x4 kf[KS];
        if constexpr (NEXT) {
            constexpr int BO = NB2 * 32 * ROWB;
            const int kc = kaddr + nboff, kl = klast0 + (nboff & kl_mask);
            kf[0] = rdK(kc + BO); kf[1] = rdK(kc + BO + 32); kf[2] = rdK(kl + BO);
            // MFMA 1, 2: K step 0 of both groups (C = 0); ordered behind the previous step's last MFMA through the Q operand
            // (PV16: the carried sources retire one MFMA later than in the 32-row form — a tie sits anywhere between the value it takes and the MFMA it feeds, so
            //  "behind MFMA 1" needs MFMA 1's RESULT as the ordering input: the source of the previous step's second-to-last MFMA behind 1, those of its last behind 2)
            if constexpr (PV16) AE_TIE0(qf[0][0], o16[QG - 1][1][2]);
            else AE_TIE1(qf[0][0], o[QG - 1][NDB - 1], hp[0]);
            {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf[0]), as_bf16x8(qf[0][0]), z, 0, 0, 0);
                if constexpr (PV16) AE_TIE1(qf[1][0], sn[0], hp[0]);
                else AE_TIE2(qf[1][0], sn[0], hv, hp[1]);
                sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf[0]), as_bf16x8(qf[1][0]), z, 0, 0, 0);
            }
            if constexpr (PV16) AE_TIE2(sn[0], sn[1], hv, hp[1]);
            else AE_TIE0(sn[0], sn[1]);
            sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf[1]), as_bf16x8(qf[0][1]), sn[0], 0, 0, 0);     // 3
            AE_TIE0(sn[1], sn[0]);
            sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf[1]), as_bf16x8(qf[1][1]), sn[1], 0, 0, 0);     // 4
            AE_TIE1(sn[0], sn[1], kf[0]);
            sn[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf[2]), as_bf16x8(qf[0][2]), sn[0], 0, 0, 0);     // 5
            AE_TIE0(sn[1], sn[0]);
            sn[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf[2]), as_bf16x8(qf[1][2]), sn[1], 0, 0, 0);     // 6
        }
        if constexpr (CUR && PV16) {
            const int vc = vaddr16 + boff, vl = vlast16 + (boff & vl16_mask);
            u32x4 vf3[3];
#pragma unroll
            for (int db3 = 0; db3 < 3; ++db3) {
                const int va = (db3 == 2) ? vl + B2 * 32 * RS : vc + db3 * 32 + B2 * 32 * 64;
                const int vstep = (db3 == 2) ? 8 * RS : 8 * 64;
                union { bf16x8_t b; u32x4 u; } cv;
                cv.b = cat_tr(lds_tr16(smem + va), lds_tr16(smem + va + vstep));
                vf3[db3] = cv.u;
            }
            u32x4 pk[QG][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int gq = 0; gq < QG; ++gq) {
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(__builtin_amdgcn_exp2f(sc[gq][8 * kk + 2 * j]), __builtin_amdgcn_exp2f(sc[gq][8 * kk + 2 * j + 1]));
                    pk[gq][kk] = (u32x4){w[0], w[1], w[2], w[3]};
                }
            // (query = lane & 31, 16-key half kk) -> (query = lane & 15 of half qh, key group): rows 1 / 3 of the kk = 0 register <-> rows 0 / 2 of the kk = 1 register
#pragma unroll
            for (int gq = 0; gq < QG; ++gq) {
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(pk[gq][0].x), "+v"(pk[gq][1].x));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(pk[gq][0].y), "+v"(pk[gq][1].y));
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(pk[gq][0].z), "+v"(pk[gq][1].z));
                asm volatile("v_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(pk[gq][0].w), "+v"(pk[gq][1].w));
            }
#define AE_PV(gq_, qh_, db_) o16[gq_][qh_][db_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(vf3[db_]), as_bf16x8(pk[gq_][qh_]), o16[gq_][qh_][db_], 0, 0, 0)
            if constexpr (NEXT) AE_TIE1(o16[0][0][0], sn[1], kf[1]);               // behind MFMA 6: K step 1's fragment (read by 3, 4) is free
            else AE_TIE0(o16[0][0][0], o16[QG - 1][1][2]);
            AE_PV(0, 0, 0);                                                        // 7
            if constexpr (NEXT) AE_TIE0(o16[0][1][0], o16[0][0][0]);
            else AE_TIE1(o16[0][1][0], o16[0][0][0], hp[0]);
            AE_PV(0, 1, 0);                                                        // 8
            // behind MFMA 8: the last K step's fragments of BOTH operands (read by 5, 6) are free — Q's matter in the last step that computes logits, where Q dies
            if constexpr (NEXT) asm volatile("" : "+v"(o16[1][0][0]) : "v"(o16[0][1][0]), "v"(kf[2]), "v"(qf[0][KS - 1]), "v"(qf[1][KS - 1]));
            else AE_TIE2(o16[1][0][0], o16[0][1][0], hv, hp[1]);
            AE_PV(1, 0, 0);                                                        // 9
            AE_TIE0(o16[1][1][0], o16[1][0][0]);
            AE_PV(1, 1, 0);                                                        // 10
            AE_TIE0(o16[0][0][1], o16[1][1][0]);
            AE_PV(0, 0, 1);                                                        // 11
            AE_TIE0(o16[0][1][1], o16[0][0][1]);
            AE_PV(0, 1, 1);                                                        // 12
            AE_TIE1(o16[1][0][1], o16[0][1][1], vf3[0]);                           // block 0's V fragment (last read by 10) is free behind 12
            AE_PV(1, 0, 1);                                                        // 13
            AE_TIE0(o16[1][1][1], o16[1][0][1]);
            AE_PV(1, 1, 1);                                                        // 14
            AE_TIE0(o16[0][0][2], o16[1][1][1]);
            AE_PV(0, 0, 2);                                                        // 15
            AE_TIE0(o16[0][1][2], o16[0][0][2]);
            AE_PV(0, 1, 2);                                                        // 16
            AE_TIE1(o16[1][0][2], o16[0][1][2], vf3[1]);                           // block 1's V fragment (last read by 14) is free behind 16
            AE_PV(1, 0, 2);                                                        // 17
            AE_TIE0(o16[1][1][2], o16[1][0][2]);
            AE_PV(1, 1, 2);                                                        // 18
#undef AE_PV
            asm volatile("" : "+v"(o16[1][1][2]) : "v"(pk[0][0]), "v"(pk[0][1]));  // behind MFMA 18: group 0's probabilities (last read by 15, 16) are free
            hv = vf3[2]; hp[0] = pk[1][0]; hp[1] = pk[1][1];                       // sources of 17, 18: retired behind the next step's first two MFMAs
        } else if constexpr (CUR) {
            const int vc = vaddr + boff, vl = vlast0 + (boff & vl_mask);
            u32x4 vf[2][NDB];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    const int va = (db == LDB) ? vl + B2 * 32 * RS + kk * 16 * RS : vc + db * VDB + B2 * 32 * 64 + kk * 16 * 64;
                    const int vstep = (db == LDB) ? 8 * RS : 8 * 64;
                    union { bf16x8_t b; u32x4 u; } cv;
                    cv.b = cat_tr(lds_tr16(smem + va), lds_tr16(smem + va + vstep));
                    vf[kk][db] = cv.u;
                }
            u32x4 pk[QG][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int gq = 0; gq < QG; ++gq) {
                    uint32_t w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) w[j] = pack_bf16x2(__builtin_amdgcn_exp2f(sc[gq][8 * kk + 2 * j]), __builtin_amdgcn_exp2f(sc[gq][8 * kk + 2 * j + 1]));
                    pk[gq][kk] = (u32x4){w[0], w[1], w[2], w[3]};
                }
            if constexpr (NEXT) AE_TIE1(o[0][0], sn[1], kf[1]);                    // behind MFMA 6: K step 1's fragment (read by 3, 4) is free
            else AE_TIE0(o[0][0], o[QG - 1][NDB - 1]);                             // (no logit MFMAs in this step: 7 follows the previous step's 14 ...
            o[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[0][0]), as_bf16x8(pk[0][0]), o[0][0], 0, 0, 0);   // 7
            if constexpr (NEXT) AE_TIE0(o[1][0], o[0][0]);
            else AE_TIE1(o[1][0], o[0][0], hp[0]);                                 //  ... and the carried sources retire behind 7 and 8 instead of 1 and 2)
            o[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[0][0]), as_bf16x8(pk[1][0]), o[1][0], 0, 0, 0);   // 8
            if constexpr (NEXT) AE_TIE1(o[0][1], o[1][0], kf[2]);
            else AE_TIE2(o[0][1], o[1][0], hv, hp[1]);
            o[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[0][1]), as_bf16x8(pk[0][0]), o[0][1], 0, 0, 0);   // 9
            AE_TIE0(o[1][1], o[0][1]);
            o[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[0][1]), as_bf16x8(pk[1][0]), o[1][1], 0, 0, 0);   // 10
            AE_TIE1(o[0][0], o[1][1], vf[0][0]);
            o[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[1][0]), as_bf16x8(pk[0][1]), o[0][0], 0, 0, 0);   // 11
            AE_TIE1(o[1][0], o[0][0], pk[0][0]);
            o[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[1][0]), as_bf16x8(pk[1][1]), o[1][0], 0, 0, 0);   // 12
            AE_TIE2(o[0][1], o[1][0], vf[0][1], pk[1][0]);
            o[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[1][1]), as_bf16x8(pk[0][1]), o[0][1], 0, 0, 0);   // 13
            AE_TIE0(o[1][1], o[0][1]);
            o[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf[1][1]), as_bf16x8(pk[1][1]), o[1][1], 0, 0, 0);   // 14
            asm volatile("" : "+v"(o[1][1]) : "v"(vf[1][0]));                      // behind MFMA 14: the fragment read by 11, 12 is free
            hv = vf[1][1]; hp[0] = pk[0][1]; hp[1] = pk[1][1];                     // sources of 13, 14: retired behind the next step's first two MFMAs
        }
        if constexpr (NEXT) decide(sn, first, CUR ? kf[2] : kf[1], kf[2]);
    };

    const int ntiles = p.Nk / KT;   // launcher: Nk % KT == 0, ntiles >= 2
    f32x16 sA[QG], sB[QG];
    issue(0, 0);
    issue(1, BUFB);
    dma_wait_all_and_barrier();      // tiles 0 and 1 (and the ones tiles) are in LDS
    if (ntiles > 2) issue(2, 2 * BUFB);
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    using YES = std::true_type;
    using NO = std::false_type;
    step(T0{}, T0{}, NO{}, YES{}, 0, 0, sB, sA, true);   // prologue: logits of block (0, 0), the first offset
    int boff = 0, nboff = BUFB, iboff = 0;   // buffers of tile t, tile t + 1, and (after the rotation below) the one tile t + 2 lands in
    for (int t = 0; t + 1 < ntiles; ++t) {
        // blocks (t, 0) .. (t, NBLK - 2); next = the following block of the same tile
        step(T0{}, T1{}, YES{}, YES{}, boff, boff, sA, sB, false);
        if constexpr (NBLK == 4) {
            step(T1{}, T2{}, YES{}, YES{}, boff, boff, sB, sA, false);
            step(T2{}, T3{}, YES{}, YES{}, boff, boff, sA, sB, false);
        }
        // block (t, NBLK - 1); next = (t + 1, 0): tile t + 1 must be visible to every wave.  Its pieces were issued a whole tile ago; the barrier also proves
        // that every wave has left tile t - 1 (it has finished step (t, 0), which follows its last read of tile t - 1): tile t + 2 goes there.
        if (t > 0) {
            dma_wait_all_and_barrier();
            if (t + 2 < ntiles) issue(t + 2, iboff);
        }
        step(std::integral_constant<int, NBLK - 1>{}, T0{}, YES{}, YES{}, boff, nboff, sB, sA, false);
        iboff = boff; boff = nboff; nboff = (nboff == 2 * BUFB) ? 0 : nboff + BUFB;
    }
    // last tile: its last block has no successor
    step(T0{}, T1{}, YES{}, YES{}, boff, boff, sA, sB, false);
    if constexpr (NBLK == 4) {
        step(T1{}, T2{}, YES{}, YES{}, boff, boff, sB, sA, false);
        step(T2{}, T3{}, YES{}, YES{}, boff, boff, sA, sB, false);
    }
    step(std::integral_constant<int, NBLK - 1>{}, T0{}, YES{}, NO{}, boff, boff, sB, sA, false);
#undef AE_TIE0
#undef AE_TIE1
#undef AE_TIE2

    // ---- normalise and store (as attn_fast_kernel::store_o without a second segment)
    bf16_t* op = p.o + (long)b * p.o_sb + (long)h * p.o_sh;
    // (the Q operand is dead behind the last logit MFMA: keep it a live value until here — asm volatile statements keep their order, and the last
    // step's ties sit eight MFMAs behind that MFMA)
    // (PV16 ties the last K step's Q fragments behind PV MFMA 8 of every step instead: a use this far away made hipcc keep a COPY of them live and free the
    //  registers the MFMA reads — the listing of the first form had the block maximum written into one 15 instructions behind logit MFMA 6)
    if constexpr (!PV16) asm volatile("" ::"v"(qf[0][0]), "v"(qf[0][1]), "v"(qf[0][2]), "v"(qf[1][0]), "v"(qf[1][1]), "v"(qf[1][2]));
    if constexpr (PV16) {
        // denominators: row 40 of O^T = third block, local row 8 = register 0 of the lanes g == 2; query 16 qh + l15 sits in column l15
        float ls[QG][2];
#pragma unroll
        for (int gq = QG - 1; gq >= 0; --gq)
#pragma unroll
            for (int qh = 1; qh >= 0; --qh) ls[gq][qh] = __shfl(o16[gq][qh][2][0], 32 + l15, 64);   // first: the result of the LAST MFMA issued ...
        asm volatile("" : "+v"(ls[QG - 1][1]) : "v"(hv), "v"(hp[0]), "v"(hp[1]));                 // ... whose sources stay live until that read is out
#pragma unroll
        for (int gq = 0; gq < QG; ++gq)
#pragma unroll
            for (int qh = 0; qh < 2; ++qh) {
                const float lsum = ls[gq][qh];
                const float inv = (p.out_scale ? p.out_scale[b] : 1.0f) / lsum;
                const float mq = __shfl(mt[gq], 16 * qh + l15, 64);
                const int qrow = q0 + 32 * gq + 16 * qh + l15;
                if (p.lse && g == 0 && qrow < p.Nq) p.lse[((long)b * p.H + h) * p.Nq + qrow] = mq + __builtin_amdgcn_logf(lsum);
                if (qrow < p.Nq) {
#pragma unroll
                    for (int db3 = 0; db3 < 3; ++db3) {
                        const int d = 16 * db3 + 4 * g;
                        if (d < D) {
                            float r0 = o16[gq][qh][db3][0] * inv, r1 = o16[gq][qh][db3][1] * inv, r2 = o16[gq][qh][db3][2] * inv, r3 = o16[gq][qh][db3][3] * inv;
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
        return;
    }
    float lsum_g[QG];
#pragma unroll
    for (int gq = QG - 1; gq >= 0; --gq) lsum_g[gq] = __shfl(o[gq][LDB][LREG], l31, 64);   // group 1 first: it reads the result of the LAST MFMA issued ...
    asm volatile("" : "+v"(lsum_g[QG - 1]) : "v"(hv), "v"(hp[0]), "v"(hp[1]));              // ... whose sources (and the one before's) stay live until that read is out
#pragma unroll
    for (int gq = 0; gq < QG; ++gq) {
        const float lsum = lsum_g[gq];
        const float inv = (p.out_scale ? p.out_scale[b] : 1.0f) / lsum;
        const int qrow = q0 + 32 * gq + l31;
        if (p.lse && hi == 0 && qrow < p.Nq) p.lse[((long)b * p.H + h) * p.Nq + qrow] = mt[gq] + __builtin_amdgcn_logf(lsum);
        if (qrow < p.Nq) {
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int d = 32 * db + 8 * r4 + 4 * hi;
                    if (d < D) {
                        float r0 = o[gq][db][4 * r4] * inv, r1 = o[gq][db][4 * r4 + 1] * inv, r2 = o[gq][db][4 * r4 + 2] * inv, r3 = o[gq][db][4 * r4 + 3] * inv;
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
}

#ifndef AE_ATTN_PV16_DEFAULT
#define AE_ATTN_PV16_DEFAULT 0
#endif
template <int D>
int launch_fast(const AttnArgs& a, hipStream_t stream) {
    const long blocks = (long)((a.Nq + 127) / 128) * a.B * a.H;
    dim3 grid((unsigned)blocks), block(256);
#ifdef AE_ATTN_LAB
    static const int abl = getenv("AE_ATTN_ABL") ? atoi(getenv("AE_ATTN_ABL")) : 0;
#define AE_ABL(n) if (abl == n) { hipLaunchKernelGGL((attn_fast_kernel<D, 3, false, n>), grid, block, 0, stream, a); return ae_check_launch("abl"); }
    AE_ABL(1) AE_ABL(2) AE_ABL(3) AE_ABL(4) AE_ABL(5) AE_ABL(10) AE_ABL(11) AE_ABL(12)
#endif
    if constexpr (D % 16 == 0) {
        if (a.rel_h) {
            // SAM windows (kH, kW <= 16; 196 tokens): one block of seven waves per (window, head), every key resident after one LDS-DMA burst, the
            // rel-pos bias as two more K steps of the logit MFMA chain (BIAS 3).  ViT-H encoder, 28 windowed blocks: 49 -> 29 us per launch, encoder
            // 11.35 -> 10.80 ms (profiles/r04_v19_sam_win.txt; resident keys with the look-up bias of BIAS 1 measured 48 us: the look-ups were the
            // bound).  Tuning knob AE_ATTN_WIN=0: the tiled look-up form.
            static const int win_env = getenv("AE_ATTN_WIN") ? atoi(getenv("AE_ATTN_WIN")) : 1;
            if constexpr (D <= 96) {
                if (win_env && a.kW != FKT && a.Nq <= 224 && a.Nk <= 4 * FKT) {
                    AttnArgs b = a;
                    b.ng = 1;
                    hipLaunchKernelGGL((attn_fast_kernel<D, 2, false, 0, 3, 1, false, true, 7, 4>), dim3((unsigned)((long)a.B * a.H)), dim3(448), 0, stream, b);
                    return ae_check_launch("ae_attn_fwd_bf16(fast, rel-pos, resident keys)");
                }
            }
            if (a.kW == FKT) hipLaunchKernelGGL((attn_fast_kernel<D, 2, false, 0, 2>), grid, block, 0, stream, a);
            else hipLaunchKernelGGL((attn_fast_kernel<D, 2, false, 0, 1>), grid, block, 0, stream, a);
            return ae_check_launch("ae_attn_fwd_bf16(fast, rel-pos)");
        }
    }
    // Short K/V (cross-attention): all keys resident, p.ng 128-query groups per block (see SKV at the kernel).  Tuning knobs:
    // AE_ATTN_SKV=0 off; AE_ATTN_SKV_NG=n forces the groups per block (default: the largest of 8/4/2/1 that still fills 3/4 of the block slots).
    static const int skv_env = getenv("AE_ATTN_SKV") ? atoi(getenv("AE_ATTN_SKV")) : 1;
    static const int skv_ng = getenv("AE_ATTN_SKV_NG") ? atoi(getenv("AE_ATTN_SKV_NG")) : 0;
    if (skv_env && !a.rel_h && a.Nk <= 2 * FKT && (!a.k2 || a.Nk2 <= FKT)) {
        AttnArgs b = a;
        b.ng = 1;
        constexpr int occ = D > 96 ? 1 : (D > 64 ? 2 : 3);
        for (int n = 8; n > 1; n >>= 1)   // at least 3/4 of the chip's 256 x occ block slots stay filled
            if ((long)((a.Nq + 128 * n - 1) / (128 * n)) * a.B * a.H >= 192 * occ) { b.ng = n; break; }
        if (skv_ng > 0) b.ng = skv_ng;
        dim3 gridk((unsigned)((long)((a.Nq + 128 * b.ng - 1) / (128 * b.ng)) * a.B * a.H));
        if (a.k2) hipLaunchKernelGGL((attn_fast_kernel<D, (D > 96 ? 1 : (D > 64 ? 2 : 3)), true, 0, 0, 1, false, true>), gridk, block, 0, stream, b);
        else hipLaunchKernelGGL((attn_fast_kernel<D, (D > 96 ? 1 : (D > 64 ? 2 : 3)), false, 0, 0, 1, false, true>), gridk, block, 0, stream, b);
        return ae_check_launch("ae_attn_fwd_bf16(fast, short K/V)");
    }
    if (a.k2) {
        hipLaunchKernelGGL((attn_fast_kernel<D, (D > 96 ? 1 : (D > 64 ? 2 : 3)), true>), grid, block, 0, stream, a);  // 168 VGPRs spill at head_dim 80
        return ae_check_launch("ae_attn_fwd_bf16(fast)");
    }
    // Variants of the plain long-sequence kernel (tuning knob AE_ATTN_V): 0 = one 32-query group per wave, V rows of 2 D bytes (round 2);
    // 1 = conflict-free V image (VSPLIT); 3 = VSPLIT + two query groups per wave (QG = 2: 64 queries per wave, 256 per block) where the
    // sequence is long enough to fill the chip with the larger blocks.  Measured (kbench, UNet batch 12, N = 4096, d = 40, one box):
    // 388.5 / 376.9 / 361.2 us for 0 / 1 / 3; in situ 13.78 -> 13.67 ms per UNet step (profiles/r03_v2_*).  QG = 2 on the row-major V
    // image (the former value 2) measured 367.7 us and is not kept: one variant fewer to validate.
    if constexpr (D > 96) {  // head_dim 160 (16x16 / 8x8 UNet levels): six 32-row output blocks per wave, two waves per SIMD
        hipLaunchKernelGGL((attn_fast_kernel<D, 2, false>), grid, block, 0, stream, a);
        return ae_check_launch("ae_attn_fwd_bf16(fast)");
    }
    static const int var_env = getenv("AE_ATTN_V") ? atoi(getenv("AE_ATTN_V")) : AE_ATTN_V_DEFAULT;
    // flag 4 (round 5): the software-pipelined kernel where the two-query-group kernel would run and the key count is whole tiles
    if constexpr (D == 40) {
        if ((var_env & 4) && (long)((a.Nq + 255) / 256) * a.B * a.H >= 512 && a.Nk % FKT == 0 && a.Nk >= 2 * FKT && !a.lse2) {
            dim3 grid2((unsigned)((long)((a.Nq + 255) / 256) * a.B * a.H));
            // AE_ATTN_KT128 (round 6): 128-key tiles where the key count allows (one barrier per 128 keys); 0 = the 64-key form everywhere (A/B)
            static const int kt128 = getenv("AE_ATTN_KT128") ? atoi(getenv("AE_ATTN_KT128")) : 1;
            // AE_ATTN_PV16 (round 6): the 48-row 16x16x32 PV products on the 128-key form (A/B; see the kernel)
            static const int pv16 = getenv("AE_ATTN_PV16") ? atoi(getenv("AE_ATTN_PV16")) : AE_ATTN_PV16_DEFAULT;
            if (pv16 && kt128 && a.Nk % 128 == 0 && a.Nk >= 256) hipLaunchKernelGGL((attn_pipe_kernel<D, 128, true>), grid2, block, 0, stream, a);
            else
            if (kt128 && a.Nk % 128 == 0 && a.Nk >= 256) hipLaunchKernelGGL((attn_pipe_kernel<D, 128>), grid2, block, 0, stream, a);
            else hipLaunchKernelGGL((attn_pipe_kernel<D, 64>), grid2, block, 0, stream, a);
            return ae_check_launch("ae_attn_fwd_bf16(fast, pipelined)");
        }
    }
    const bool qg2 = (var_env & 2) && D <= 48 && (long)((a.Nq + 255) / 256) * a.B * a.H >= 1024;   // head_dim 80 would spill with two groups
    if (qg2) {
        if constexpr (D <= 48) {
            dim3 grid2((unsigned)((long)((a.Nq + 255) / 256) * a.B * a.H));
            hipLaunchKernelGGL((attn_fast_kernel<D, 3, false, 0, 0, 2, true>), grid2, block, 0, stream, a);   // 3 waves per SIMD: 168 VGPRs
        }
    } else if (var_env & 1) hipLaunchKernelGGL((attn_fast_kernel<D, 3, false, 0, 0, 1, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_fast_kernel<D, 3, false>), grid, block, 0, stream, a);
    return ae_check_launch("ae_attn_fwd_bf16(fast)");
}

}  // namespace

int ae_attn_fast_launch(const AttnArgs& a, int D, hipStream_t stream) {
    if (a.key_mask) return AE_ERR_UNSUPPORTED;
    // rel-pos bias at head dims that are multiples of 16: one key-grid row per tile (kW == 64, SAM global attention) or a small grid
    // (kH, kW <= 16, SAM windows)
    if (a.rel_h && !(D % 16 == 0 && !a.k2 && !a.accum && !a.out_scale && ((a.kW == FKT && a.Nk % FKT == 0) || (a.kH <= 16 && a.kW <= 16)))) return AE_ERR_UNSUPPORTED;
    if (a.k2 && (a.accum || a.out_scale)) return AE_ERR_UNSUPPORTED;
    // 32-bit byte offsets inside one (batch, head) image of K / V
    if (((long)a.Nk * a.k_sn + D) * 2 >= (1L << 31) || ((long)a.Nk * a.v_sn + D) * 2 >= (1L << 31)) return AE_ERR_UNSUPPORTED;
    if (a.k2 && (((long)a.Nk2 * a.k2_sn + D) * 2 >= (1L << 31) || ((long)a.Nk2 * a.v2_sn + D) * 2 >= (1L << 31))) return AE_ERR_UNSUPPORTED;
    switch (D) {
        case 40: return launch_fast<40>(a, stream);
        case 80: return launch_fast<80>(a, stream);
        case 160: {  // the 16x16 / 8x8 UNet levels (tuning knob AE_ATTN_FAST160=0: general kernel, for A/B)
            static const int f160 = getenv("AE_ATTN_FAST160") ? atoi(getenv("AE_ATTN_FAST160")) : 1;
            return (a.rel_h || !f160) ? AE_ERR_UNSUPPORTED : launch_fast<160>(a, stream);
        }
        default: return AE_ERR_UNSUPPORTED;
    }
}
