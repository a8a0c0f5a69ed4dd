// bf16 MFMA GEMM and implicit-GEMM 3x3 convolution for gfx950 (MI355X).
//
//   C[M,N] = epi( A[M,K] * W[N,K]^T )            A, W bf16 K-contiguous; fp32 accumulate
//
// Replaces: nn.Linear / 1x1 nn.Conv2d created through ldm/modules/diffusionmodules/util.py:202-238
// (rows A1-A4), and the ResBlock / Downsample / Upsample 3x3 convolutions of openaimodel.py:108-118,157-159,254-274
// (row A5) on channels-last activations, where conv3x3 is the same GEMM with an on-the-fly im2col gather
// (M = B*Ho*Wo, K = 9*Cin ordered (ky,kx,cin)); stride-2 and nearest-x2 upsampling are folded into the gather.
//
// Structure (wave64; the instantiations and who launches which are listed at launch()):
//   * block tile BM x BN x 64 in LDS, filled by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write) with the 16-byte-chunk
//     XOR swizzle applied on the per-lane SOURCE side (conflict-free ds_read_b128 of MFMA fragments); a register-staged loader remains for
//     tiles that need per-chunk decisions (K tails, padded channels);
//   * main loops: 192x320 tile, 8 waves, one block per CU — the PING-PONG loop (two groups of four waves one barrier apart: LOAD interval ‖ MFMA
//     interval, hand-placed counted vmcnt, asm DMA; WA = 3) and its activation-SLAB form for stride-1 convs (WA = 4);  128x128 tile, 8 waves, two
//     blocks per CU — two-stage ring with the weights two K tiles ahead (WA = 1) or a three-stage ring where one block per CU runs anyway;
//     128x64 / 64x64 tiles for small grids; split-K (fp32 partials + fixed-order reduce) for M <= 3072 convs;
//   * v_mfma_f32_16x16x32_bf16 with the operands swapped (W fragment as the row operand) so each lane ends up with 4 consecutive output columns;
//   * fused epilogues staged through LDS into 16-byte row-contiguous stores: +bias, +per-(batch,channel) vector (time embedding), SiLU / exact-erf
//     GELU / GEGLU gate, +residual, bf16 or fp32 output, optional per-channel (CS) or per-row (XE 1) statistics of the stored values, optional
//     LayerNorm fold (XE 2);
//   * XCD-aware bijective tile remap so tiles sharing an operand panel run on the same XCD's L2.
#include "common.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef AE_CONV_SPEC
#define AE_CONV_SPEC 1   // round 4: on (prepared in round 3, measured at the start of round 4: 40 checksums identical, UNet step 13.47 -> 13.44 ms over three alternating runs, profiles/r04_v2_cspec_ab.txt)
#endif
#ifndef AE_GEMM_WA_DEFAULT
#define AE_GEMM_WA_DEFAULT 3
#endif
#ifndef AE_GEMM_PP_DEFAULT
#define AE_GEMM_PP_DEFAULT 63   // 15 + the slab form of the conv loop (16 un-split, 32 split-K): outputs bit-identical (56 checksums), every launch of the family 1-3 % faster
                                // un-graphed, UNet step -0.02 ms over three alternating A/B rounds on two boxes (profiles/r04_v24..v26_lnfold_slab_ab.txt)
#endif
#ifndef AE_XE_OCC
#define AE_XE_OCC 4         // waves per SIMD the 128x128 LayerNorm-fold instantiations must leave room for (two 8-wave blocks per CU); 0 = unconstrained (A/B builds)
#endif
#ifndef AE_PP_LAB
#define AE_PP_LAB 0         // lab builds only (tools/ubench/conv_lab.hip): 1 no DMA after the prologue, 2 DMA + barriers only (no LDS reads, no MFMAs), 3 MFMAs on stale registers (no LDS reads)
#endif
#ifndef AE_PP_PRIO
#define AE_PP_PRIO 1        // s_setprio 1 around the MFMA intervals of the ping-pong loop
#endif
#ifndef AE_PP_DMA_FIRST
#define AE_PP_DMA_FIRST 0   // ping-pong LOAD interval: DMA pieces in front of the fragment reads (1) or behind them (0)
#endif
constexpr int BK = 64;  // 64 bf16 = 128 B per tile row = 8 chunks of 16 B

enum { A_DENSE = 0, A_CONV3 = 1 };
enum { EPI_NONE = 0, EPI_GELU = 1, EPI_GEGLU = 2, EPI_SILU = 3, EPI_RELU = 4 };

struct GemmArgs {
    const bf16_t* A;
    const bf16_t* A2;  // dense only: columns [Ksplit, K) come from A2 (channel-concat without a copy)
    const bf16_t* W;
    void* C;
    const float* bias;
    const bf16_t* res;
    const float* addvec;
    int M, N, K, Ksplit;
    long lda, lda2, ldw, ldc, ldr, ldav;  // ldav: row stride of addvec (a column slice of a batched projection)
    int epi, out_f32, rows_per_batch;
    int H, Wd, Cin, CinPad, Ho, Wo, stride, ups;  // conv3x3: Cin = channels in memory, CinPad = per-tap K extent
    unsigned a_bytes, a2_bytes, w_bytes;  // buffer extents (bytes) for the bounds-checked fast loaders
    int splitk;      // > 1: K is cut into `splitk` ranges, each block writes raw fp32 partials (small-M, huge-K convs)
    float* partial;  // [splitk][M][N] fp32
    float* colstats; // optional [ceil(M/32)][N][2] fp32: per-channel (sum, sum of squares) of the bf16-rounded OUTPUT over each 32-row slab, written
                     // by the epilogue (CS instantiations) so that the GroupNorm that consumes this tensor needs no statistics pass
    int kmajor;      // conv: K ordered (64-channel chunk, tap, channel) instead of (tap, channel): the nine taps of a chunk in consecutive K
                     // tiles, so that a block's activation window (~40 KB) is re-read from L2, not from the MALL (weights packed to match)
    int m_fast;      // tile order inside an XCD's run: 1 = M tiles fastest (tiles sharing a WEIGHT panel are neighbours: small-M layers whose
                     // weights outweigh the activations), 0 = N tiles fastest (tiles sharing an ACTIVATION panel are neighbours)
    // LayerNorm folded into the GEMM that consumes it (round 4; attention.py:271-275's norm -> projection pairs of the 32x32 / 16x16 levels):
    //   LN(x) W^T + b = rstd_m (x W'^T - mu_m s) + c   with  W' = W diag(gamma),  s[n] = sum_k W'[n][k],  c = W beta + b,
    // so the consumer multiplies the UN-normalised rows and its epilogue applies two per-row scalars and two per-column vectors; the
    // row sums come from the epilogue of the GEMM that PRODUCED x (proj_in, attn1.to_out, attn2.to_out), as one (sum, sum of squares)
    // pair per row and 64-column slice (XE instantiations below) — the LayerNorm launch, its read of x and its write of LN(x) vanish.
    int xe;                  // 0 none, 1 this launch EMITS the row statistics of its bf16 output, 2 this launch CONSUMES them (LayerNorm fold)
    float* rowstats;         // xe 1: [M][N / 64][2] fp32 (sum, sum of squares) of the bf16-rounded output over each 64-column slice of a row
    const float* ln_stats;   // xe 2: [M][ln_parts][2] fp32, the row statistics of A as its producer emitted them
    const float* ln_colsum;  // xe 2: [N] fp32, s[n] = sum_k W'[n][k] over the bf16 values of W' (what the MFMA multiplies); `bias` holds c
    int ln_parts;
    float ln_eps;
    // Nearest-x2 upsample + 3x3 conv as FOUR 2x2 convs on the low-resolution input (round 5; openaimodel.py:108-118).  Output pixel (2y + py, 2x + px)
    // of conv3x3(upsample2(X)) reads X at rows {y - 1, y} (py = 0) or {y, y + 1} (py = 1) and likewise in x: per output parity (py, px) a 2 x 2 window
    // of the 3x3 frame, with the weights of the taps that fall on the same source pixel SUMMED at pack time (ops.pack_conv3x3_up2) — 4 / 9 of the
    // multiply-adds of the gather form, no upsampling arithmetic in the loader.  sub2 = 1: the grid holds four copies of the tile grid, copy `par`
    // (the split-K index of a launch that does not split) multiplies against weight set `par` ([N, 4 CinPad] each, K ordered (tap, channel) with tap =
    // 2 i + j over the window rows / columns) and scatters its rows to the parity's pixels of the [B, 2H, 2W] output.
    int sub2;
    // split-K launches only: 1 = stop at the fp32 partials (no splitk_reduce_kernel launch): the consumer folds the K ranges itself (ae_groupnorm_splitk_nhwc_bf16:
    // the GroupNorm of the 16x16 / 8x8 levels reads the partials instead of the reduced tensor — one launch and one round trip of the activation less)
    int defer_reduce;
};

// LDS-DMA: one wave moves 64 x 16 B from global straight into LDS at (wave-uniform dst) + lane*16.  The builtin exists only
// in the device pass of this translation unit (the host pass merely needs the kernel's launch stub).
// `soffset` is the wave-uniform part of the address (the K position): it rides in an SGPR, so the per-lane voffset is loop
// invariant and the K loop issues no address VALU at all (VALU and MFMA share the SIMD's vector pipe on gfx950: measured
// 2.3 cycles per VALU instruction ADDED to the MFMA time, tools/ubench/issue_model.hip).  The bounds check covers voffset only,
// so an out-of-range voffset (halo / tail rows) still reads as zero whatever soffset is.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, bf16_t* lds_dst, int voffset, int soffset = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
#endif
}

#ifdef AE_GEMM_LAB
// lab only (tools/ubench/conv_lab.hip): shader-cycle buckets of waves 0 and 4 (one SIMD's two waves) of one block in the middle of the grid:
// [0] prologue, [1] DMA issue, [2] LDS reads + MFMAs, [3] barrier + DMA drain, [4] epilogue tail, [5] K tiles, [6] epilogue staging (bias /
// activation + fp32 tile into LDS), [7] epilogue output (barrier, LDS reads, residual, global stores).  The stamps serialise the scalar
// pipe: the split between [1] and [2] is pessimistic (un-instrumented, the DMA issue overlaps the first MFMAs); totals per phase are sound.
// ping-pong loop (WA 3): [1] L(t,0) reads + DMA issue + lgkmcnt(0), [2] barrier behind it, [3] M(t,0) MFMA issue, [8] barrier behind it, [9] L(t,1) reads + DMA issue +
// lgkmcnt(0), [10] its vmcnt wait, [11] barrier, [12] M(t,1), [13] barrier; wave 0 (group 0) in [0, 16), wave 4 (group 1) in [16, 32).
__device__ unsigned long long g_gemm_dbg[32];
#define GL_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (lab_me) g_gemm_dbg[lab_base + (i)] += t_ - tlast; tlast = t_; } while (0)
#elif defined(AE_GEMM_TRACE)
// trace build (tools/ubench/pp_lab.hip): every wave of one mid-grid block stamps the cycle counter at each point GL_T marks, for TR_T consecutive K tiles
// from TR_K0 on, into the 8 KiB of LDS behind the ping-pong ring (no VMEM store inside the loop: it would count in vmcnt); dumped to g_pp_trace at the end.
constexpr int TR_K0 = 18, TR_T = 6;
__device__ unsigned long long g_pp_trace[8 * TR_T * 16];
// (stamps are collected in scalar registers and written once per K tile: a store — or just the wait for s_memtime's result — at every stamp cost ~200 cycles each)
#define GL_T(i) do { __builtin_amdgcn_sched_barrier(0); tr_t[(i)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GL_T(i)
#endif

// smallest divisor of FM (16-row fragments per wave) whose share of the fp32 tile fits the main-loop LDS
constexpr int epilogue_passes(int FM, int bytes_per_frag_row, int lds_bytes) {
    for (int ps = 1; ps <= FM; ++ps)
        if (FM % ps == 0 && (FM / ps) * bytes_per_frag_row <= lds_bytes) return ps;
    return FM;
}

// WAVES_K = 2: two groups of WAVES_M x WAVES_N waves split every 64-deep K tile between them (32 each) and are summed through
// LDS once at the end.  Same thread count and staging as an 8-wave block, but each wave owns a 2x larger output tile, so the
// block issues a third fewer LDS fragment reads per MFMA (the 128x128 tile is LDS-bound: with 15/16 of its MFMAs removed it
// still takes 73 % of the time).
// CS = true: the epilogue also emits the per-channel statistics of its output (GemmArgs::colstats).  The wave tile is then staged in
// 32-row passes (one statistics slab per pass) and the output loop gives every lane a FIXED 8-column chunk, so a lane can carry the
// column sums of its rows in registers; the R = 64 / (WN / 8) row-lanes of a chunk are folded through the wave's own staging slice.
// (Round 3, measured and removed: a hybrid loader — one operand global -> VGPR -> ds_write_b128, the other by LDS-DMA — to test whether
// the K-step time of these kernels (~4600 cycles for the 64 KiB of a 192x320 step, ~2100 for the 32 KiB of a 128x128 step = bytes /
// ~14 B/clk/CU) is the LDS-DMA path's rate.  It is not that simple: 192x320 conv 87.8 -> 92.6 us (A through registers) / 96.2 (W),
// 128x128 conv 105.9 -> 115.9 / 118.1, dense 128x128 +3..+10 %, UNet step 14.44 -> 14.51-14.60 ms (profiles/r03_v3_hybrid_loader.txt).)
// WA (round 3): operand-ahead loop on the two-stage LDS-DMA pipeline — 1: three W stages (weights two K tiles ahead), 2: three A stages
// (activations two tiles ahead); counted vmcnt + raw s_barrier instead of the per-step drain, unrolled by six so that every stage index is a
// constant.  See the comment at the loop, DESIGN.md §7a and profiles/r03_v30_weights_ahead.txt.
// LAB (AE_GEMM_LAB builds only, tools/ubench): 1 = no DMA after the first tile, 2 = no LDS reads / MFMAs, 3 = MFMAs on stale registers (no LDS reads)
// XE (round 4): 1 = the epilogue also emits per-row (sum, sum of squares) of its bf16 output per 64-column slice (GemmArgs::rowstats), 2 = the
// epilogue applies the LayerNorm fold from such statistics of A (GemmArgs::ln_stats / ln_colsum).  XE = 0 instantiations are unchanged code.
// (second launch bound = waves per SIMD the register allocation must leave room for; 0 = no constraint.  The LayerNorm-fold instantiations of the 8-wave
// 128x128 tile carry 36 more live registers than their XE = 0 twins and sit AT the 128-register line that decides whether two blocks share a CU: an unrelated
// edit moved them from 126 to 129 registers and the qkv / q launches from 38.2 to 47.3 us (profiles/r04_final2 / r04_final3_kernel_stats.csv) — so the
// line is stated here, not left to the allocator's mood.)
template <int BM, int BN, int AMODE, int WAVES_M = 2, int WAVES_N = 2, bool GLDS = false, int WAVES_K = 1, int STAGES = 2, bool CS = false, int LAB = 0, int WA = 0, int XE = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WAVES_K, (XE == 2 && BM == 128 && BN == 128 && STAGES == 2) ? AE_XE_OCC : 0) void gemm_kernel(const GemmArgs p) {
    static_assert(XE == 0 || (AMODE == A_DENSE && WAVES_K == 1 && !CS && GLDS), "row statistics / LayerNorm fold: dense LDS-DMA instantiations, one K group, no column statistics");
    static_assert(XE != 1 || BN / WAVES_N == 64, "row statistics are emitted per 64-column slice = one wave tile's width");
    static_assert(STAGES == 2 || (GLDS && WAVES_K == 1), "the deep LDS ring exists only for the LDS-DMA loader");
    static_assert(!WA || (GLDS && STAGES == 2 && WAVES_K == 1), "operand-ahead is a variant of the two-stage LDS-DMA pipeline");
    static_assert(WA >= 0 && WA <= 4, "WA: 0 none, 1 weights two tiles ahead (three W stages), 2 activations two tiles ahead (three A stages), 3 ping-pong (two wave groups one barrier apart), 4 ping-pong over an activation slab per (channel chunk, ky)");
    static_assert(WA < 3 || 64 * WAVES_M * WAVES_N * WAVES_K == 512, "the ping-pong loops pair wave w with wave w + 4 (one SIMD's two waves)");
    static_assert(WA != 4 || (AMODE == A_CONV3 && GLDS && BM % 16 == 0), "the slab loop is a 3x3-convolution loop");
    static_assert(WAVES_K == 1 || WAVES_K == 2, "K groups: 1 or 2");
    constexpr int NT = 64 * WAVES_M * WAVES_N * WAVES_K;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave tile
    constexpr int FM = WM / 16, FN = WN / 16;            // 16x16 fragments per wave
    constexpr int A_CH = BM * 8 / NT, B_CH = BN * 8 / NT;
    static_assert((BM * 8) % NT == 0 && (BN * 8) % NT == 0, "tile / thread-count combination not supported");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];  // 2 * (BM + BN) * BK bf16 (dynamic: > 64 KiB for 128x160)
    bf16_t* const smem = reinterpret_cast<bf16_t*>(smem_raw);
    bf16_t* sA = smem;
    bf16_t* sB = smem + (WA >= 2 ? 3 : STAGES) * BM * BK;

    // The wave index as a SCALAR (v_readfirstlane once): tid >> 6 is wave-uniform but lives in a VGPR, and every LDS-DMA destination derived
    // from it then costs a v_readfirstlane_b32 per piece per K step in front of its s_mov m0 (48-66 per six steps of the operand-ahead loops;
    // VALU ops per six steps 86 -> 20 dense, 308 -> 260 conv on the 192x320 tile).  Outputs identical (40 checksums), UNet step
    // 13.642 / 13.641 -> 13.552 / 13.559 ms in alternating A/B runs (profiles/r03_v52_wave_sgpr.txt).  -DAE_WAVE_SGPR=0 restores the old form.
#ifndef AE_WAVE_SGPR
#define AE_WAVE_SGPR 1
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = AE_WAVE_SGPR ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
#ifdef AE_GEMM_TRACE
    const bool tr_blk = blockIdx.x == gridDim.x / 2 && WA == 3;
    unsigned long long* const tr_lds = reinterpret_cast<unsigned long long*>(smem_raw + 152 * 1024);
    int tr_kt = -1;
    unsigned long long tr_t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) tr_t[q] = 0;
#endif
#ifdef AE_GEMM_LAB
    unsigned long long tlast = __builtin_readcyclecounter();
    const bool lab_me = blockIdx.x == gridDim.x / 2 && lane == 0 && (wave == 0 || wave == 4);
    const int lab_base = wave == 0 ? 0 : 16;
#endif
    const int wk = wave / (WAVES_M * WAVES_N), wmn = wave % (WAVES_M * WAVES_N);  // K group, position inside the group
    const int wm = wmn / WAVES_N, wn = wmn % WAVES_N;
    const int l15 = lane & 15, lg = lane >> 4;
    const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
    const int ntiles = ntm * ntn;
    const int split = blockIdx.x / ntiles;       // split-K index; sub2: the output parity 2 py + px
    const int py = (AMODE == A_CONV3 && p.sub2) ? split >> 1 : 0, px = (AMODE == A_CONV3 && p.sub2) ? split & 1 : 0;
    const int tile = xcd_remap(blockIdx.x % ntiles, ntiles);
    const int m0 = (p.m_fast ? tile % ntm : tile / ntn) * BM, n0 = (p.m_fast ? tile / ntm : tile % ntn) * BN;

    // ---- per-thread staging descriptors (loads are branch-free: indices are clamped, never predicated) --------
    long a_base[A_CH];            // dense: element offset of the (clamped) row ; conv: pixel base of the batch
    int a_iy[A_CH], a_ix[A_CH];   // conv: top-left tap coordinate in the (virtual) input; out-of-range rows get a far-away
                                  // coordinate so every tap is "halo" (zero)
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int row = (tid + i * NT) >> 3;
        const int m = m0 + row;
        const int mc = min(m, p.M - 1);
        if (AMODE == A_DENSE) {
            a_base[i] = (long)mc;
            a_iy[i] = a_ix[i] = 0;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = mc / hw, rem = mc - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_base[i] = (long)b * p.H * p.Wd;
            a_iy[i] = (m < p.M) ? oy * p.stride - 1 : -100000;
            a_ix[i] = ox * p.stride - 1;
        }
    }
    long b_base[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int row = (tid + i * NT) >> 3;
        b_base[i] = (long)min(n0 + row, p.N - 1) * p.ldw;
    }

    // ---- fast loaders: raw buffer loads with 32-bit byte offsets.  An offset past the descriptor's extent returns 0 in
    // hardware, so tails and the conv halo cost no clamps / selects / 64-bit address math (the gather was VALU-bound:
    // ~10 VALU per MFMA, profiles/r01_pmc_conv128x64.txt).  Tiles that need per-chunk decisions fall back to the generic loader.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W) + ((AMODE == A_CONV3 && p.sub2) ? (long)split * p.N * p.ldw : 0L), 0, (int)p.w_bytes, 0x00020000);   // (sub2: w_bytes is ONE parity's weight set)
    constexpr int OOB = (int)0x80000000;  // any offset >= 2 GiB is out of range for every tensor on this path
    int fa_off[A_CH], fa2_off[A_CH], fb_off[B_CH];
    unsigned fa_mask[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int id = tid + i * NT, row = id >> 3;
        const int c = GLDS ? ((id & 7) ^ (row & 7)) : (id & 7);  // GLDS: lane's LDS slot is fixed (lane-linear), pick the chunk that lives there
        const int m = m0 + row;
        fa_mask[i] = 0u;
        if (AMODE == A_DENSE) {
            fa_off[i] = (m < p.M) ? (int)((long)m * p.lda * 2) + c * 16 : OOB;
            fa2_off[i] = (m < p.M) ? (int)((long)m * p.lda2 * 2) + c * 16 : OOB;
        } else {
            fa2_off[i] = OOB;
            const int hw = p.Ho * p.Wo;
            const int mc = min(m, p.M - 1);
            const int b = mc / hw, rem = mc - b * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int iy0 = oy * p.stride - 1, ix0 = ox * p.stride - 1;
            fa_off[i] = (int)(((long)(b * p.H + iy0) * p.Wd + ix0) * p.Cin + c * 8) * 2;
            const int Hv = p.ups ? 2 * p.H : p.H, Wv = p.ups ? 2 * p.Wd : p.Wd;  // virtual (upsampled) input extent
            if (m < p.M) {
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) {
                    const int iy = iy0 + t9 / 3, ix = ix0 + t9 % 3;
                    // ups == 2: zero-insert upsampling (adjoint of a stride-2 conv): only even virtual pixels carry data
                    if ((unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv && (p.ups != 2 || ((iy | ix) & 1) == 0)) fa_mask[i] |= 1u << t9;
                }
            }

        }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int id = tid + i * NT, row = id >> 3;
        const int c = GLDS ? ((id & 7) ^ (row & 7)) : (id & 7);
        fb_off[i] = (n0 + row < p.N) ? (int)((long)(n0 + row) * p.ldw * 2) + c * 16 : OOB;
    }
    const bool conv_fast = (AMODE == A_CONV3) && p.Cin == p.CinPad;

    u32x4 ra[A_CH], rb[B_CH];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int KT_all = (p.K + BK - 1) / BK;
    const int kt_per = (KT_all + p.splitk - 1) / p.splitk;
    const int kt_begin = (AMODE == A_CONV3 && p.sub2) ? 0 : split * kt_per;
    const int KT = max(min(KT_all - kt_begin, kt_per), 0);  // K tiles of this block
    const int klast = p.K - 8;
    // conv: tap / channel offset of the NEXT tile to load
    const int conv_per = AMODE == A_CONV3 ? p.CinPad / BK : 1;                                            // K tiles per filter tap (tap-major order)
    const unsigned conv_per_magic = (AE_CONV_SPEC && conv_per > 1) ? 0xFFFFFFFFu / (unsigned)conv_per + 1u : 0u;   // ceil(2^32 / per)
    const unsigned pp_magic = (WA == 3 && conv_per > 1) ? 0xFFFFFFFFu / (unsigned)conv_per + 1u : 0u;                // the same reciprocal for the ping-pong loop (exact for lin < 2^16, per <= 64)
    int ld_tap = (kt_begin * BK) / (AMODE == A_CONV3 ? p.CinPad : 1 << 30), ld_ci = (AMODE == A_CONV3) ? (kt_begin * BK) % p.CinPad : 0;
    if (AMODE == A_CONV3 && p.kmajor) { ld_tap = kt_begin % 9; ld_ci = (kt_begin / 9) * BK; }  // (chunk, tap) order: LDS-DMA loader only

    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        const int k0 = (kt_begin + kt) * BK;
        const bool full_k = k0 + BK <= p.K;
        if (AMODE == A_DENSE && full_k && (k0 + BK <= p.Ksplit || k0 >= p.Ksplit)) {
            if (k0 < p.Ksplit) {
#pragma unroll
                for (int i = 0; i < A_CH; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, fa_off[i] + k0 * 2, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < A_CH; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA2, fa2_off[i] + (k0 - p.Ksplit) * 2, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_CH; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, fb_off[i] + k0 * 2, 0, 0);
            return;
        }
        if (conv_fast) {
            const int ky = ld_tap / 3, kx = ld_tap - ky * 3;
            const int tap_off = ((ky * p.Wd + kx) * p.Cin + ld_ci) * 2;  // wave-uniform
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                int src = fa_off[i] + tap_off;
                if (p.ups) {  // nearest-x2 upsample folded into the gather: source pixel = virtual pixel >> 1 (3 of 64 convs per UNet call)
                    const int cc = GLDS ? (((tid + i * NT) & 7) ^ (((tid + i * NT) >> 3) & 7)) : ((tid + i * NT) & 7);
                    src = (int)((a_base[i] + (long)(max(a_iy[i] + ky, 0) >> 1) * p.Wd + (max(a_ix[i] + kx, 0) >> 1)) * p.Cin + cc * 8 + ld_ci) * 2;
                }
                const int off = ((fa_mask[i] >> ld_tap) & 1u) ? src : OOB;  // halo / tail rows -> hardware zero
                ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0);
            }
            ld_ci += BK;
            if (ld_ci >= p.CinPad) { ld_ci = 0; ++ld_tap; }
#pragma unroll
            for (int i = 0; i < B_CH; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsW, fb_off[i] + k0 * 2, 0, 0);
            return;
        }
        if (AMODE == A_DENSE) {
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int c = (tid + i * NT) & 7;
                const int k = min(k0 + c * 8, klast);  // K tail: valid (finite) data, multiplied by the zeroed W tail
                const bf16_t* src = (k >= p.Ksplit) ? p.A2 + a_base[i] * p.lda2 + (k - p.Ksplit) : p.A + a_base[i] * p.lda + k;
                ra[i] = *reinterpret_cast<const u32x4*>(src);
            }
        } else {
            const int ky = ld_tap / 3, kx = ld_tap - ky * 3;
            const int Hv = p.ups ? 2 * p.H : p.H, Wv = p.ups ? 2 * p.Wd : p.Wd;
#pragma unroll
            for (int i = 0; i < A_CH; ++i) {
                const int c = (tid + i * NT) & 7;
                const int iy = a_iy[i] + ky, ix = a_ix[i] + kx;
                const int ci = ld_ci + c * 8;
                const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv && ci < p.Cin && (p.ups != 2 || ((iy | ix) & 1) == 0);
                const int cy = min(max(iy, 0), Hv - 1), cx = min(max(ix, 0), Wv - 1);
                const int sy = p.ups ? (cy >> 1) : cy, sx = p.ups ? (cx >> 1) : cx;
                const u32x4 v = *reinterpret_cast<const u32x4*>(p.A + (a_base[i] + (long)sy * p.Wd + sx) * p.Cin + min(ci, p.Cin - 8));
                ra[i] = ok ? v : zero4;  // zero padding of the conv halo (select, not a branch)
            }
            ld_ci += BK;
            if (ld_ci >= p.CinPad) { ld_ci = 0; ++ld_tap; }
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int k = k0 + ((tid + i * NT) & 7) * 8;
            const u32x4 v = *reinterpret_cast<const u32x4*>(p.W + b_base[i] + min(k, klast));
            rb[i] = (k < p.K) ? v : zero4;
        }
    };

    auto store_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            const int id = tid + i * NT, row = id >> 3, c = id & 7;
            *reinterpret_cast<u32x4*>(sA + buf * BM * BK + row * BK + ((c ^ (row & 7)) << 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            const int id = tid + i * NT, row = id >> 3, c = id & 7;
            *reinterpret_cast<u32x4*>(sB + buf * BN * BK + row * BK + ((c ^ (row & 7)) << 3)) = rb[i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // XE 2: rstd and -rstd * mean of this lane's row in each 16-row fragment, from the producer's per-slice sums, and the block's s / c vectors.
    // Everything the fold needs is LOADED here, in front of the main loop, and consumed in `xe_mid()`, which every loop variant calls right after
    // its first DMA wait — a point where the wave has just waited for memory anyway — so neither the statistics nor s / c cost the block a
    // round trip of their own (one block per CU on the 192x320 tile: nothing else would hide it; measured +7 .. +9 us per GEGLU launch when
    // the epilogue started with these loads, profiles/r04_v25_lnfold_slab_ab.txt).
    //  * statistics: the four lane groups of a row (lg) take every fourth slice and meet through two cross-lane adds: (s0 + s1) + (s2 + s3) in
    //    every lane, a fixed order.  All loads of all fragment rows go out together (LNP = 5 per row: up to 20 slices, K <= 1280; slices past
    //    ln_parts re-read the last one and count as zero): a first form looped over the slices and paid one L2 round trip per iteration and row.
    //  * s / c: the 128x128 tiles keep this lane's 2 x FN x 4 values in registers through the loop; the 192x320 tile (254 registers) parks the
    //    block's 2 x 320 values in the 8 KiB of LDS the ping-pong ring leaves free (behind its three A and two W stages).
    float ln_r[XE == 2 ? FM : 1], ln_t[XE == 2 ? FM : 1];
    constexpr int LNP = 5;
    constexpr bool SC_LDS = XE == 2 && WA == 3;
    static_assert(!SC_LDS || (BN % 4 == 0 && (3 * BM + 2 * BN) * BK * 2 + 2 * BN * 4 <= 160 * 1024), "s / c of the block must fit behind the ping-pong ring");
    float* const sc_lds = reinterpret_cast<float*>(smem_raw + (3 * BM + 2 * BN) * BK * 2);   // SC_LDS: [2][BN]: s, then c
    f32x2 raw[XE == 2 ? FM : 1][LNP];
    f32x4 szp[(XE == 2 && !SC_LDS) ? FN : 1], bzp[(XE == 2 && !SC_LDS) ? FN : 1], scv[2];
    if constexpr (XE == 2) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = min(m0 + wm * WM + i * 16 + l15, p.M - 1);
            const f32x2* sp = reinterpret_cast<const f32x2*>(p.ln_stats) + (long)m * p.ln_parts;
#pragma unroll
            for (int u = 0; u < LNP; ++u) raw[i][u] = sp[min(lg + 4 * u, p.ln_parts - 1)];
        }
        if constexpr (SC_LDS) {
            const int n = min(n0 + 4 * min(tid, BN / 4 - 1), p.N - 4);
            scv[0] = *reinterpret_cast<const f32x4*>(p.ln_colsum + n);
            scv[1] = *reinterpret_cast<const f32x4*>(p.bias + n);
        } else {
#pragma unroll
            for (int j = 0; j < FN; ++j) {   // the epilogue's column of fragment j (GEGLU: 'a' rows, gate rows at +16)
                const int nc = p.epi == EPI_GEGLU ? min(n0 + wn * WN + (j & ~1) * 16 + lg * 4, p.N - 20) + (j & 1) * 16 : min(n0 + wn * WN + j * 16 + lg * 4, p.N - 4);
                szp[j] = *reinterpret_cast<const f32x4*>(p.ln_colsum + nc);
                bzp[j] = *reinterpret_cast<const f32x4*>(p.bias + nc);
            }
        }
    }
    auto xe_mid = [&]() __attribute__((always_inline)) {
        if constexpr (XE == 2) {
            const float inv_k = 1.0f / (float)p.K;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int u = 0; u < LNP; ++u) {
                    const bool in = lg + 4 * u < p.ln_parts;
                    s += in ? raw[i][u][0] : 0.f;
                    q += in ? raw[i][u][1] : 0.f;
                }
                s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
                s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                const float mu = s * inv_k;
                const float r = rsqrtf(fmaxf(q * inv_k - mu * mu, 0.f) + p.ln_eps);
                ln_r[i] = r;
                ln_t[i] = -r * mu;
            }
            if constexpr (SC_LDS) {   // read by every wave in the epilogue, hundreds of barriers from here
                if (tid < BN / 4) {
                    // GEGLU: the packed columns interleave 16 'a' rows with their 16 gate rows; c of an 'a' column is stored halved (geglu_half_f)
                    const float hc = (p.epi == EPI_GEGLU && ((4 * tid) & 16) == 0) ? 0.5f : 1.0f;
                    reinterpret_cast<f32x4*>(sc_lds)[tid] = scv[0];
                    reinterpret_cast<f32x4*>(sc_lds + BN)[tid] = (f32x4){scv[1][0] * hc, scv[1][1] * hc, scv[1][2] * hc, scv[1][3] * hc};
                }
            }
        }
    };

    auto compute_tile = [&](int cur, int curB = -1) __attribute__((always_inline)) {
        if (LAB == 2) return;
        const bf16_t* cA = sA + cur * BM * BK + (wm * WM) * BK;
        const bf16_t* cB = sB + (curB < 0 ? cur : curB) * BN * BK + (wn * WN) * BK;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (WAVES_K == 2 && kk != wk) continue;  // wave-uniform: this K half belongs to the other group
            bf16x8_t af[FM], bfr[FN];
            const int ch = kk * 4 + lg;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = i * 16 + l15;  // (wm*(BM/2)) is a multiple of 8 -> same swizzle phase
                if (LAB == 3) af[i] = as_bf16x8((u32x4){(uint32_t)(i + cur), 0x3f803f80u, (uint32_t)lane, 0u});
                else af[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(cA + row * BK + ((ch ^ (row & 7)) << 3)));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = j * 16 + l15;
                if (LAB == 3) bfr[j] = as_bf16x8((u32x4){(uint32_t)(j + kk), 0x3f803f80u, (uint32_t)lane, 0u});
                else bfr[j] = as_bf16x8(*reinterpret_cast<const u32x4*>(cB + row * BK + ((ch ^ (row & 7)) << 3)));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    // D[n][m]: lane holds m = l15, n = 4*lg + r  (operands swapped on purpose)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    // Main-loop variants of the two-stage LDS-DMA pipeline that were built, verified against the tests and measured in round 3
    // (profiles/r03_v11_midb_ilv_regstage.txt) and are NOT kept:
    //  - mid-step refill: wait(tile kt) + barrier, first K half, second half's LDS reads, barrier, second half's MFMAs, DMA of tile kt + 2
    //    into the stage just read (two tiles in flight on two stages, counted vmcnt).  Isolated: 128x128 conv -3 %, 192x320 conv +7..11 %,
    //    dense +-4 %; UNet step 13.86 -> 13.84 ms (noise).
    //  - interleaved issue: the A_CH + B_CH pieces of tile kt + 1 issued one after every third MFMA of the first K half instead of as a
    //    burst at the top of the step (sched_barrier fences or sched_group_barrier, last step peeled).  Isolated (operands L2-hot) the
    //    192x320 launches gain 2..8 % (ff2 M = 49152 N = 320 K = 1280: 52.5 -> 48.0 us); inside the UNet evaluation, where each layer's
    //    weights arrive cold, the later pieces no longer land before the step's barrier: 14.01 -> 14.13 ms, two runs each way.
    //  - register-staged loader on the 192x320 conv tile (buffer_load -> 32 staging VGPRs -> ds_write_b128, 246 VGPRs, no spills):
    //    259 -> 308 us on the K = 8640 conv, 86 -> 106 us on K = 2880.
    // Together with the ablations in profiles/r03_v7*.txt: the DMA burst costs the waves ~0.7 us of a 1.9 us step, none of the three
    // re-arrangements hides it, and the arrangement below (burst at the top, one barrier per step) is the best of the measured ones.
    // Software pipeline: global loads of tile t+1 are issued into registers before tile t is multiplied out of LDS
    // (issue-early / write-late), one barrier per K tile.  (A two-tile-deep register ring was measured slower: it pushes
    // the 128x128 variant to 256 VGPRs + spills.)
    if (!GLDS) {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < KT; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < KT) load_tile(kt + 1);
            compute_tile(cur);
            if (kt + 1 < KT) store_tile(cur ^ 1);
            __syncthreads();
        }
    } else {
        // LDS-DMA pipeline: `buffer_load_dwordx4 ... lds` moves each 1-KiB piece (8 tile rows x 128 B) straight into the next
        // LDS stage — no staging VGPRs, no ds_write_b128 pass (the slowest LDS instruction, ~79 B/clk/CU).  The LDS image is
        // lane-linear, so the XOR swizzle is applied to the per-lane SOURCE chunk instead (same involution as the ds_read
        // side).  hipcc drains the DMA (vmcnt(0)) in front of each __syncthreads().
        // conv: this tap's per-lane gather offset (OOB for halo pixels).  The offsets are VGPR operands of the DMA pieces, and hipcc waits
        // for the pieces that read a register before it rewrites it: WA 2 (A tiles two ahead) therefore rotates three register sets with the
        // A stages — the set rewritten in step kt was last read by pieces issued three steps earlier, long landed.
        int fa_cur[WA == 2 ? 3 : 1][A_CH];
        // (always_inline: out of line, the by-reference captures — ld_tap, ld_ci, fa_cur — live in scratch memory, and a scratch load inside the
        // loop is a VMEM operation hipcc waits for with vmcnt(0): it drained the whole DMA queue once per step in the unrolled operand-ahead loop)
        auto dma_a = [&](int kt, int buf) __attribute__((always_inline)) {
            const int fs = WA == 2 ? buf : 0;
            const int k0 = (kt_begin + kt) * BK;
            if (AMODE == A_DENSE) {
                const bool second = k0 >= p.Ksplit;
                if (AE_CONV_SPEC) {   // one wave-uniform branch per step instead of one per piece (hipcc keeps the test inside the unrolled piece loop)
                    if (!second) {
#pragma unroll
                        for (int i = 0; i < A_CH; ++i) lds_dma16(rsA, sA + buf * BM * BK + (wave + (NT / 64) * i) * 8 * BK, fa_off[i], k0 * 2);
                    } else {
#pragma unroll
                        for (int i = 0; i < A_CH; ++i) lds_dma16(rsA2, sA + buf * BM * BK + (wave + (NT / 64) * i) * 8 * BK, fa2_off[i], (k0 - p.Ksplit) * 2);
                    }
                } else {
#pragma unroll
                for (int i = 0; i < A_CH; ++i) {
                    bf16_t* dst = sA + buf * BM * BK + (wave + (NT / 64) * i) * 8 * BK;
                    if (!second) lds_dma16(rsA, dst, fa_off[i], k0 * 2);
                    else lds_dma16(rsA2, dst, fa2_off[i], (k0 - p.Ksplit) * 2);
                }
                }
            } else {
                // operand-ahead loops and deep rings: the (tap, channel) position is derived from kt — carried as loop state (ld_tap / ld_ci, captured by
                // reference) it ended up in scratch memory in the unrolled loop, and every scratch load is a VMEM operation hipcc waits for
                // with vmcnt(0): the whole DMA queue drained once per step
                // AE_CONV_SPEC (build variant, prepared for an A/B): the activations-ahead conv instantiation is only launched for chunk-major K and
                // no upsample, so both run-time flags fold away — no generic division for the tap-major position, no per-piece upsample branch
                constexpr bool KM1 = AE_CONV_SPEC && WA == 2;
                int cur_tap = ld_tap, cur_ci = ld_ci;
                constexpr bool STATELESS = true;   // (the two-stage loops too: their scratch load of the tap state sat in front of every step's DMA issue)
                if (STATELESS) {
                    const int lin = kt_begin + kt;
                    if (KM1 || p.kmajor) { cur_tap = lin % 9; cur_ci = (lin / 9) * BK; }
                    else if (AE_CONV_SPEC) {   // tap-major: lin / per by the reciprocal set up once in front of the loops (exact for lin < 2^16, per <= 64)
                        cur_tap = conv_per == 1 ? lin : (int)__umulhi((unsigned)lin, conv_per_magic);
                        cur_ci = (lin - cur_tap * conv_per) * BK;
                    } else { const int per = p.CinPad / BK; cur_tap = lin / per; cur_ci = (lin - cur_tap * per) * BK; }
                }
                // sub2: cur_tap counts the 2 x 2 window's taps; (ky, kx) is the tap's place in the 3 x 3 frame the masks and offsets are laid out for
                const int ky = (!KM1 && p.sub2) ? py + (cur_tap >> 1) : cur_tap / 3, kx = (!KM1 && p.sub2) ? px + (cur_tap & 1) : cur_tap - ky * 3;
                const int mbit = (!KM1 && p.sub2) ? ky * 3 + kx : cur_tap;
                if (cur_ci == 0 || kt == 0 || KM1 || p.kmajor || WA == 2) {  // a new tap: the per-lane part (halo mask, upsample source pixel) changes only here
#pragma unroll
                    for (int i = 0; i < A_CH; ++i) {
                        int src = fa_off[i] + ((ky * p.Wd + kx) * p.Cin) * 2;  // >= 0 for every in-image tap (voffset is bounds-checked unsigned)
                        if (!KM1 && p.ups) {  // nearest-x2 upsample folded into the gather: source pixel = virtual pixel >> 1 (3 of 64 convs per UNet call)
                            const int cc = ((tid + i * NT) & 7) ^ (((tid + i * NT) >> 3) & 7);
                            src = (int)((a_base[i] + (long)(max(a_iy[i] + ky, 0) >> 1) * p.Wd + (max(a_ix[i] + kx, 0) >> 1)) * p.Cin + cc * 8) * 2;
                        }
                        fa_cur[fs][i] = ((fa_mask[i] >> mbit) & 1u) ? src : OOB;
                    }
                }
                const int tap_off = cur_ci * 2;  // wave-uniform -> SGPR offset
                if (!(LAB == 4 && (kt % 9) >= 2)) {   // LAB 4: the A tile only on 2 of 9 steps (= the DMA volume of a slab loader, 42 of 216 KiB)
#pragma unroll
                for (int i = 0; i < A_CH; ++i) {
                    bf16_t* dst = sA + buf * BM * BK + (wave + (NT / 64) * i) * 8 * BK;
                    lds_dma16(rsA, dst, fa_cur[fs][i], tap_off);
                }
                }
                if (STATELESS) {
                } else if (p.kmajor) {
                    if (++ld_tap == 9) { ld_tap = 0; ld_ci += BK; }
                } else {
                    ld_ci += BK;
                    if (ld_ci >= p.CinPad) { ld_ci = 0; ++ld_tap; }
                }
            }
        };
        auto dma_w = [&](int kt, int buf) __attribute__((always_inline)) {
            const int k0 = (kt_begin + kt) * BK;
#pragma unroll
            for (int i = 0; i < B_CH; ++i) {
                bf16_t* dst = sB + buf * BN * BK + (wave + (NT / 64) * i) * 8 * BK;
                lds_dma16(rsW, dst, fb_off[i], k0 * 2);
            }
        };
        auto dma_tile = [&](int kt, int buf) __attribute__((always_inline)) { dma_a(kt, buf); dma_w(kt, buf); };
        if constexpr (WA == 4) {
            // ---- Ping-pong over an activation SLAB (round 4, second half).  The ping-pong loop below sits at the LDS-DMA path's own rate
            // (36.8 B/clk/CU: 64 KiB per K tile against 1920 MFMA cycles, profiles/r04_v4_pp_lab.txt), so this form moves fewer bytes: in the
            // chunk-major K order the three taps (ky, 0..2) of a 64-channel chunk read the SAME input pixels one position apart.  The tile's
            // BM output pixels are whole image rows (W | BM, checked by the launcher), so per (chunk, ky) the BM / W input rows y + ky - 1 are
            // staged ONCE, as a slab with one ZERO row in front of every image row and one behind the last:
            //     slab row  j (W + 1)          = zero                       (j = 0 .. BM / W)
            //     slab row  j (W + 1) + 1 + x  = input pixel (y_j + ky - 1, x), zero when that row is outside the image
            // and output pixel (row j, x) reads slab row j (W + 1) + x + kx for tap kx: x - 1 = -1 and x + 1 = W land on the zero rows, a row
            // outside the image is zero as a whole — the conv's zero padding costs nothing in the loop (a first form that shifted through a
            // contiguous pixel range and zeroed the wrapped rows in registers cost 48-72 VALU per K tile in the LOAD intervals).  Zero rows and
            // out-of-image rows are lanes of the LDS-DMA pieces whose offset is out of the descriptor's range: the hardware writes zeros.
            // 26 one-KiB pieces per slab (208 rows: W >= 16) instead of 3 x 24: DMA per chunk 3 x 32 + 9 x 40 = 456 KiB instead of 576 (waves without a
            // piece in the last round repeat their previous one so that every wave's count is the same), activation reads from L2 a third.
            // Split-K launches: a block's K range starts at a chunk boundary (kt_begin % 9 == 0, checked by the launcher).
            // Fragment reads: 16 | W, so a 16-row fragment lies in one image row and its slab rows are consecutive; the XOR swizzle follows
            // the slab row, hence one per-lane address per (fragment, kx, half) — 36 registers, no address arithmetic in the loop.
            // Schedule: as in the loop below, with the A side in slab units.  Tile t = (chunk c, tap): L(t,0) reads half 0 and issues W(t + 1);
            // L(t,1) reads half 1 and, when kx = 0, issues the NEXT slab (c, ky + 1) / (c + 1, 0) into the other of two slab stages — that
            // stage held slab a - 1, last read in tile t - 1 by group 1's L(t - 1, 1), whose lgkmcnt(0) precedes the barrier that group 0
            // passed before this interval.  Waits: after a slab issue vmcnt(S_CH) (W(t + 1) has landed, the slab flies on), otherwise
            // vmcnt(0): by the end of the kx = 1 tile every wave has seen its slab pieces land, two barriers before their first reader.
            // Unrolled by 18 = lcm(9 taps, 2 stages): tap, stage offsets and the issue decision are constants of each unrolled tile.
            static_assert(BK * 2 == 128, "slab rows are 128 bytes");
            constexpr int SLAB_P = (BM + BM / 16 + 1 + 7) / 8;          // 1-KiB pieces per slab stage (8 rows each): 26 for BM = 192, i.e. 208 rows >= BM + BM / W + 1 for W >= 16
            constexpr int S_CH = (SLAB_P + NT / 64 - 1) / (NT / 64);    // pieces per wave; waves without a last-round piece repeat their previous one
            constexpr int A_ST = SLAB_P * 1024, W_ST = BN * BK * 2;     // stage sizes in bytes
            const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem_raw);
            const int ldsW = lds0 + 2 * A_ST + wave * 1024;
            const int grp = wave >> 2;
            const int W1 = p.Wd + 1, nrow = BM / p.Wd;                  // slab row pitch, image rows per tile
            const int y0 = (m0 / p.Wd) % p.H;                           // image row of the tile's first output row (m0 is a multiple of W)
            int sl_off[S_CH], sl_dst[S_CH];
            unsigned sl_ok[S_CH];                                       // bit ky: the slab row is an input pixel inside the image for that ky
#pragma unroll
            for (int j = 0; j < S_CH; ++j) {
                int q = wave + (NT / 64) * j;
                if (q >= SLAB_P) q -= NT / 64;
                const int srow = 8 * q + (lane >> 3);                  // slab row of this lane; its LDS slot holds logical chunk (lane & 7) ^ (srow & 7)
                const int jr = srow / W1, pos = srow - jr * W1;
                const int m = m0 + jr * p.Wd + pos - 1;                 // output pixel above / below which the input pixel sits
                // the tile's rows may cross into the next sample: (y0 + jr) mod H is the image row whatever the sample
                int y = y0 + jr;
                y -= (y >= p.H) ? p.H : 0;
                unsigned ok = 0u;
                if (pos > 0 && jr < nrow && m < p.M) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
                        if ((unsigned)(y + ky - 1) < (unsigned)p.H) ok |= 1u << ky;
                }
                sl_ok[j] = ok;
                sl_off[j] = m * p.Cin * 2 + (((lane & 7) ^ (lane >> 3)) << 4);
                sl_dst[j] = lds0 + q * 1024;
            }
            const int ky_step = p.Wd * p.Cin * 2;
            const int cbase = kt_begin / 9;                             // first channel chunk of this block's K range
            auto pp_slab = [&](int c, int ky, int st) __attribute__((always_inline)) {
                int off[S_CH];
#pragma unroll
                for (int j = 0; j < S_CH; ++j) off[j] = ((sl_ok[j] >> ky) & 1u) ? sl_off[j] + (ky - 1) * ky_step : OOB;   // zero rows / rows outside the image -> hardware zero
#pragma unroll
                for (int j = 0; j < S_CH; ++j) asm volatile("" : "+v"(off[j]));   // all offsets first, then the pieces back to back
#pragma unroll
                for (int j = 0; j < S_CH; ++j) ae_dma16(rsA, sl_dst[j] + st * A_ST, off[j], (cbase + c) * (BK * 2));
            };
            auto pp_w = [&](int kt, int st) __attribute__((always_inline)) {
                const int k0 = (kt_begin + kt) * BK;
#pragma unroll
                for (int i = 0; i < B_CH; ++i) ae_dma16(rsW, ldsW + st * W_ST + i * (NT / 64) * 1024, fb_off[i], k0 * 2);
            };
            // LDS read offsets (bytes).  A fragment i of tap kx, half kk: slab row r0 + r0 / W + l15 + kx with r0 = wm WM + 16 i (the zero rows in
            // front of its image row shift it), 16-byte chunk (4 kk + lg) ^ (row & 7).  W fragments as in the loop below.
            int a_rd[FM][3][2];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r0 = wm * WM + i * 16;
                const int base = r0 + r0 / p.Wd + l15;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) a_rd[i][kx][kk] = (base + kx) * 128 + (((kk * 4 + lg) ^ ((base + kx) & 7)) << 4);
            }
            int w_rd[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w_rd[kk] = 2 * A_ST + (wn * WN + l15) * 128 + (((kk * 4 + lg) ^ (l15 & 7)) << 4);
            bf16x8_t af[FM], bfr[FN];
            auto pp_read = [&](int kk, int kx, int sa, int sw) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < FM; ++i) af[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(smem_raw + sa * A_ST + a_rd[i][kx][kk]));
#pragma unroll
                for (int j = 0; j < FN; ++j) bfr[j] = as_bf16x8(*reinterpret_cast<const u32x4*>(smem_raw + sw * W_ST + w_rd[kk] + j * 16 * 128));
            };
            auto pp_mfma = [&]() __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                if (AE_PP_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                if (AE_PP_PRIO) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            };
            if (KT > 0) { pp_w(0, 0); pp_slab(0, 0, 0); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (grp) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind
            __builtin_amdgcn_sched_barrier(0);
            for (int kt0 = 0; kt0 < KT; kt0 += 18) {
                const int c0 = kt0 / 9;              // even: the slab stage of tile kt0 + u is ((u / 9) + ky) & 1
#pragma unroll
                for (int u = 0; u < 18; ++u) {
                    const int kt = kt0 + u;
                    if (kt >= KT) break;
                    const int tap = u % 9, ky = tap / 3, kx = tap % 3;
                    const int sa = ((u / 9) + ky) & 1, sw = u & 1;
                    // L(t,0)
                    pp_read(0, kx, sa, sw);
                    if (kt + 1 < KT) pp_w(kt + 1, sw ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    pp_mfma();   // M(t,0)
                    // L(t,1)
                    pp_read(1, kx, sa, sw);
                    const bool slab_next = kx == 0 && kt + 3 < KT;
                    if (slab_next) pp_slab(ky == 2 ? c0 + u / 9 + 1 : c0 + u / 9, ky == 2 ? 0 : ky + 1, sa ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (slab_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S_CH) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    pp_mfma();   // M(t,1)
                }
            }
            if (!grp) __builtin_amdgcn_s_barrier();  // group 0 waits for group 1's last MFMA interval
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (WA == 3) {
            // ---- Ping-pong (round 4).  The eight waves form two groups — waves 0-3 and 4-7: wave w and w + 4 share a SIMD — that run the SAME
            // instruction stream one barrier apart.  A K tile is two phases (its two 32-deep halves), a phase is a LOAD interval (the half's
            // 11-13 fragment reads LDS -> registers, then this interval's LDS-DMA pieces, then the waits) and an MFMA interval (the half's
            // FM x FN MFMAs on those registers, nothing else), each ended by a raw s_barrier.  Because of the one-barrier offset every SIMD
            // always holds one wave in an MFMA interval (s_setprio 1) and one in a LOAD interval: the matrix pipe never waits for an LDS read, a
            // DMA issue stall or a vmcnt wait of its OWN wave — in the lock-step loops above those costs add to the MFMA time
            // (profiles/r03_v7_gemm_mainloop_ablations.txt: compute alone 158 us, DMA + barriers alone 196 us, together 256 us).
            // Epochs (e = barriers passed); group 0: L(t,0) = 4t, M(t,0) = 4t + 1, L(t,1) = 4t + 2, M(t,1) = 4t + 3; group 1 one later.
            //   LDS: three A stages + two W stages (152 KiB for 192x320).
            //   L(t,0) issues W(t + 1) into W stage (t + 1) & 1: that stage held W(t - 1), last read by group 1 in epoch 4t - 1 with its
            //          lgkmcnt(0) BEFORE the barrier that ends the epoch; the earliest issue is group 0's in epoch 4t.
            //   L(t,1) issues A(t + 2) into A stage (t + 2) % 3 (held A(t - 1), same argument), then waits vmcnt(A_CH): DMA pieces retire in
            //          issue order, so A(t + 1) and W(t + 1) have landed while A(t + 2) keeps flying; group 1's wait ends in epoch 4t + 3, the
            //          first reader of tile t + 1 is group 0's L(t + 1,0) in epoch 4t + 4, behind the barrier.
            // The DMA pieces go through inline asm (ae_dma16): hipcc does no bookkeeping for them, every wait below is placed by hand.
            // Unrolled by six (lcm of the ring lengths) so that every stage offset is an immediate.
            const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem_raw);
            constexpr int A_ST = BM * BK * 2, W_ST = BN * BK * 2;   // stage sizes in bytes
            const int ldsA = lds0 + wave * 1024, ldsW = lds0 + 3 * A_ST + wave * 1024;
            const int grp = wave >> 2;
            auto pp_w = [&](int kt, int st) __attribute__((always_inline)) {
                const int k0 = (kt_begin + kt) * BK;
#pragma unroll
                for (int i = 0; i < B_CH; ++i) ae_dma16(rsW, ldsW + st * W_ST + i * (NT / 64) * 1024, fb_off[i], k0 * 2);
            };
            // Activation pieces.  conv: the (tap, channel chunk) position of the NEXT tile to issue is carried as scalar state and advanced by a few
            // s_add / s_cselect per tile — the stateless form (lin % 9, lin / 9, tap / 3, two multiplies, a branch per flag) cost the LOAD interval
            // that issues them ~250 cycles more than the weight pieces' (barrier-arrival trace, profiles/r04_v7_pp_trace.txt; moving that arithmetic
            // one interval ahead only moved the cost: profiles/r04_v8_pp_prep.txt).  The three per-lane offsets are worked out into separate
            // registers first, then the pieces go out back to back (an offset register rewritten between two pieces waits for the first piece's issue).
            // No upsampling gather here: the launcher keeps those three convs on the round-3 loop.
            int cs_tap = 0, cs_kx = 0, cs_base = 0, cs_ci = 0;   // tap index 0..8, its kx, ((ky * W + kx) * Cin) * 2, channel offset (elements)
            const int cs_cin2 = p.Cin * 2, cs_row2 = (p.Wd - 3) * p.Cin * 2;
            // sub2 (tap-major only): the window's taps in the order (py, px), (py, px + 1), (py + 1, px), (py + 1, px + 1); cs_tap stays the 3 x 3-frame index
            auto cs_set = [&](int kt) __attribute__((always_inline)) {   // stateless (prologue only)
                const int lin = kt_begin + kt;
                if (p.kmajor) { cs_tap = lin % 9; cs_ci = (lin / 9) * BK; }
                else { cs_tap = conv_per == 1 ? lin : (int)__umulhi((unsigned)lin, pp_magic); cs_ci = (lin - cs_tap * conv_per) * BK; }
                int ky = cs_tap / 3;
                cs_kx = cs_tap - ky * 3;
                if (p.sub2) { ky = py + (cs_tap >> 1); cs_kx = px + (cs_tap & 1); cs_tap = ky * 3 + cs_kx; }
                cs_base = ((ky * p.Wd + cs_kx) * p.Cin) * 2;
            };
            auto cs_next = [&]() __attribute__((always_inline)) {
                bool tap_step = true;
                if (!p.kmajor) { cs_ci += BK; tap_step = cs_ci >= p.CinPad; cs_ci = tap_step ? 0 : cs_ci; }
                if (tap_step) {
                    if (p.sub2) {
                        if (cs_kx == px) { ++cs_tap; ++cs_kx; cs_base += cs_cin2; }
                        else { cs_tap += 2; cs_kx = px; cs_base += cs_row2 + 2 * cs_cin2; }   // one row down, one column back: (W - 1) pixels on
                    } else {
                    ++cs_tap; ++cs_kx; cs_base += cs_cin2;
                    if (cs_kx == 3) { cs_kx = 0; cs_base += cs_row2; }
                    if (cs_tap == 9) { cs_tap = 0; cs_base = 0; cs_ci += BK; }   // (chunk-major order only: in the tap-major order the loop ends with tap 8)
                    }
                }
            };
            auto pp_a = [&](int kt, int st) __attribute__((always_inline)) {
                if constexpr (AMODE == A_DENSE) {
                    const int k0 = (kt_begin + kt) * BK;
                    if (k0 < p.Ksplit) {
#pragma unroll
                        for (int i = 0; i < A_CH; ++i) ae_dma16(rsA, ldsA + st * A_ST + i * (NT / 64) * 1024, fa_off[i], k0 * 2);
                    } else {
#pragma unroll
                        for (int i = 0; i < A_CH; ++i) ae_dma16(rsA2, ldsA + st * A_ST + i * (NT / 64) * 1024, fa2_off[i], (k0 - p.Ksplit) * 2);
                    }
                } else {
                    int off[A_CH];
#pragma unroll
                    for (int i = 0; i < A_CH; ++i) off[i] = ((fa_mask[i] >> cs_tap) & 1u) ? fa_off[i] + cs_base : OOB;   // halo / tail rows -> hardware zero
#pragma unroll
                    for (int i = 0; i < A_CH; ++i) asm volatile("" : "+v"(off[i]));   // all three offsets first ...
#pragma unroll
                    for (int i = 0; i < A_CH; ++i) ae_dma16(rsA, ldsA + st * A_ST + i * (NT / 64) * 1024, off[i], cs_ci * 2);   // ... then the pieces
                    cs_next();
                }
            };
            bf16x8_t af[FM], bfr[FN];
            auto pp_read = [&](int kk, int sa, int sw) __attribute__((always_inline)) {
                const bf16_t* cA = sA + sa * BM * BK + (wm * WM) * BK;
                const bf16_t* cB = sB + sw * BN * BK + (wn * WN) * BK;
                const int ch = kk * 4 + lg;
#pragma unroll
                for (int i = 0; i < FM; ++i) {
                    const int row = i * 16 + l15;
                    if (AE_PP_LAB == 2 || AE_PP_LAB == 3) af[i] = as_bf16x8((u32x4){(uint32_t)(i + sa), 0x3f803f80u, (uint32_t)lane, 0u});
                    else af[i] = as_bf16x8(*reinterpret_cast<const u32x4*>(cA + row * BK + ((ch ^ (row & 7)) << 3)));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int row = j * 16 + l15;
                    if (AE_PP_LAB == 2 || AE_PP_LAB == 3) bfr[j] = as_bf16x8((u32x4){(uint32_t)(j + kk), 0x3f803f80u, (uint32_t)lane, 0u});
                    else bfr[j] = as_bf16x8(*reinterpret_cast<const u32x4*>(cB + row * BK + ((ch ^ (row & 7)) << 3)));
                }
            };
            auto pp_mfma = [&](int lab_m, int lab_b) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
                if (AE_PP_PRIO) __builtin_amdgcn_s_setprio(1);
                if (AE_PP_LAB != 2) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
                }
                if (AE_PP_PRIO) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                GL_T(lab_m);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                GL_T(lab_b);
            };
            if constexpr (AMODE == A_CONV3) cs_set(0);
            pp_w(0, 0); pp_a(0, 0);
            if (KT > 1) { pp_a(1, 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_CH) : "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            xe_mid();
            __builtin_amdgcn_s_barrier();
            if (grp) __builtin_amdgcn_s_barrier();   // group 1 runs one barrier behind
            __builtin_amdgcn_sched_barrier(0);
            GL_T(0);
            for (int kt0 = 0; kt0 < KT; kt0 += 6) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int kt = kt0 + u;
                    if (kt >= KT) break;
                    const int sa = u % 3, sw = u & 1;
#ifdef AE_GEMM_TRACE
                    tr_kt = kt;
                    GL_T(0);
#endif
                    // L(t,0)
                    if (AE_PP_DMA_FIRST && AE_PP_LAB != 1 && kt + 1 < KT) pp_w(kt + 1, sw ^ 1);
                    pp_read(0, sa, sw);
                    if (!AE_PP_DMA_FIRST && AE_PP_LAB != 1 && kt + 1 < KT) pp_w(kt + 1, sw ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    GL_T(1);
                    __builtin_amdgcn_s_barrier();
                    GL_T(2);
                    pp_mfma(3, 8);   // M(t,0)
                    // L(t,1)
                    if (AE_PP_DMA_FIRST && AE_PP_LAB != 1 && kt + 2 < KT) pp_a(kt + 2, (u + 2) % 3);
                    pp_read(1, sa, sw);
                    if (!AE_PP_DMA_FIRST && AE_PP_LAB != 1 && kt + 2 < KT) pp_a(kt + 2, (u + 2) % 3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    GL_T(9);
                    if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_CH) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    GL_T(10);
                    __builtin_amdgcn_s_barrier();
                    GL_T(11);
                    pp_mfma(12, 13);   // M(t,1)
#ifdef AE_GEMM_TRACE
                    if (tr_blk && kt >= TR_K0 && kt < TR_K0 + TR_T && lane == 0) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) tr_lds[(wave * TR_T + (kt - TR_K0)) * 16 + q] = tr_t[q];
                    }
#endif
#ifdef AE_GEMM_LAB
                    if (lab_me) g_gemm_dbg[lab_base + 5] += 1;
#endif
                }
            }
            if (!grp) __builtin_amdgcn_s_barrier();  // group 0 waits for group 1's last MFMA interval: every wave has passed the same number of barriers
            __builtin_amdgcn_sched_barrier(0);
#ifdef AE_GEMM_TRACE
            tr_kt = -1;
            if (tr_blk) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                for (int q = tid; q < 8 * TR_T * 16; q += NT) g_pp_trace[q] = tr_lds[q];
            }
#endif
        } else if constexpr (WA != 0) {
            // One operand two tiles ahead (three stages of it, two of the other), for the operand that arrives COLD inside a UNet evaluation:
            //   WA 1: the weights (32x32 / 8x8 levels: a layer's weights come from HBM, its activations from L2 / the Infinity Cache;
            //         tools/cold_weight_probe.py: cold weights cost those launches 5-11 %);
            //   WA 2: the activations (64x64 level on the 192x320 tile: 31-126 MB tensors that no cache holds; 3 x 24 + 2 x 40 = 152 KiB).
            // Per step the near operand's tile of step kt + 1 and then the far operand's tile of step kt + 2 are issued; DMA pieces retire in
            // issue order, so vmcnt(FAR pieces) at the end of the step proves everything of step kt + 1 has landed while the far tile of
            // step kt + 2 keeps flying.  The loop is unrolled by six (lcm of the ring lengths): every stage index is a constant (with a
            // run-time index hipcc cannot tell the stages apart and waits for the fresh pieces in front of the step's own LDS reads).
            constexpr int FAR = WA == 1 ? B_CH : A_CH;
            if (WA == 1) { dma_a(0, 0); dma_w(0, 0); if (KT > 1) dma_w(1, 1); }
            else { dma_w(0, 0); dma_a(0, 0); if (KT > 1) dma_a(1, 1); }
            if (KT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FAR) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            xe_mid();
            __builtin_amdgcn_s_barrier();
            for (int kt0 = 0; kt0 < KT; kt0 += 6) {
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int kt = kt0 + u;
                    if (kt >= KT) break;
                    const int near_cur = u & 1, far_cur = u % 3, far_next = (u + 2) % 3;   // far_next: the stage read in step kt - 1
                    if (WA == 1) {
                        if (kt + 1 < KT) dma_a(kt + 1, near_cur ^ 1);
                        if (kt + 2 < KT) dma_w(kt + 2, far_next);
                        compute_tile(near_cur, far_cur);
                    } else {
                        if (kt + 1 < KT) dma_w(kt + 1, near_cur ^ 1);
                        if (kt + 2 < KT) dma_a(kt + 2, far_next);
                        compute_tile(far_cur, near_cur);
                    }
                    if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FAR) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        } else if constexpr (STAGES == 2) {
            dma_tile(0, 0);
            __syncthreads();
            GL_T(0);
            for (int kt = 0; kt < KT; ++kt) {
                const int cur = kt & 1;
                if (LAB != 1 && kt + 1 < KT) dma_tile(kt + 1, cur ^ 1);  // stage cur^1 was last read before the previous barrier
                GL_T(1);
                compute_tile(cur);
                GL_T(2);
                __syncthreads();
                GL_T(3);
#ifdef AE_GEMM_LAB
                if (lab_me) g_gemm_dbg[lab_base + 5] += 1;
#endif
            }
        } else {
            // Deep ring: STAGES - 1 tiles in flight.  Each wave waits (counted vmcnt: only the OLDEST tile must have landed) for its own
            // DMA pieces, then a raw s_barrier makes every wave's pieces visible and proves all waves are done with the stage that is
            // refilled next.  __syncthreads() is not used: its fence would drain the whole DMA queue (vmcnt(0)).
            constexpr int PIECES = A_CH + B_CH;  // DMA instructions per tile per wave
            constexpr int AHEAD = STAGES - 1;
#pragma unroll
            for (int t = 0; t < AHEAD; ++t)
                if (t < KT) dma_tile(t, t);
            xe_mid();
            int cur = 0, nxt = AHEAD % STAGES;
            for (int kt = 0; kt < KT; ++kt) {
                const int younger = min(KT - 1 - kt, AHEAD - 1);  // tiles issued after tile kt and still allowed in flight
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (kt + AHEAD < KT) dma_tile(kt + AHEAD, nxt);  // refills the stage read in iteration kt - 1
                compute_tile(cur);
                cur = cur + 1 == STAGES ? 0 : cur + 1;
                nxt = nxt + 1 == STAGES ? 0 : nxt + 1;
            }
            __builtin_amdgcn_s_barrier();  // the epilogue reuses the ring as fp32 staging
        }
    }

    if (WAVES_K == 2) {
        // sum the two K groups: group 1 parks its accumulators in LDS (the ring is free: the loop ended on a barrier), group 0 adds
        f32x4* red = reinterpret_cast<f32x4*>(smem_raw) + (long)wmn * FM * FN * 64 + lane;
        static_assert(WAVES_K == 1 || WAVES_M * WAVES_N * FM * FN * 64 * 16 <= 2 * (BM + BN) * BK * 2, "K-group reduction must fit the ring");
        if (wk == 1) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) red[(i * FN + j) * 64] = acc[i][j];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const f32x4 o = red[(i * FN + j) * 64];
                    acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
                }
        }
        __syncthreads();
    }
    const bool writer = wk == 0;  // with K groups only group 0 holds the result; group 1 keeps the block's barriers company

    // ---- epilogue ---------------------------------------------------------------------------
#ifdef AE_GEMM_LAB_NOEPI
    if (p.M > 0) {  // lab ablation: what the kernel costs WITHOUT its epilogue (one store keeps the accumulators alive)
        if (acc[0][0][0] == 123.456f) reinterpret_cast<float*>(p.C)[0] = acc[FM - 1][FN - 1][3];
        return;
    }
#endif
    if (p.splitk > 1) {  // raw fp32 partials; bias / vector / residual are applied by splitk_reduce_kernel
        if (!writer) return;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = m0 + wm * WM + i * 16 + l15;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int n = n0 + wn * WN + j * 16 + lg * 4;
                if (n >= p.N) continue;
                *reinterpret_cast<f32x4*>(p.partial + ((long)split * p.M + m) * p.N + n) = acc[i][j];
            }
        }
        return;
    }

    // ---- coalesced epilogue: the MFMA layout gives a lane 4 columns of 16 different rows (8-byte stores scattered over 16
    // rows: store-issue bound on the short-K GEMMs).  Stage the wave's fp32 tile through its private LDS slice (the A/B
    // buffers are free now; 16-byte chunks XOR-swizzled by row) and write 16-byte row-contiguous pieces: 8 lanes = one
    // 128-byte line; the residual is read the same way.
    {
        constexpr int NCH = WN / 4;  // fp32 16-byte chunks per staged row
        // staging must fit in the main-loop LDS: split the wave tile's rows into passes if it does not (128x160 tile)
        static_assert(!CS || 2 * (64 / (WN / 8)) <= 32, "column statistics: the R row-lanes of a chunk fold through 2 R rows of the wave's own 32-row staging slice");
        static_assert(!CS || (WAVES_K == 1 && FM % 2 == 0 && WN % 16 == 0 && WAVES_M * WAVES_N * 32 * WN * 4 <= STAGES * (BM + BN) * BK * 2),
                      "column statistics need 32-row staging passes that fit the main-loop LDS");
        constexpr int PASSES = CS ? FM / 2 : epilogue_passes(FM, WAVES_M * WAVES_N * 16 * WN * 4, STAGES * (BM + BN) * BK * 2);
        constexpr int FMP = FM / PASSES, WMP = WM / PASSES;
        static_assert(FM % PASSES == 0, "epilogue passes must divide the fragment rows");
        const bool geglu = p.epi == EPI_GEGLU;
        const int n_out = geglu ? p.N / 2 : p.N;
        // Synchronisation inside the staged epilogue is WAVE-local: every wave stages and reads back only its own slice (`st` below), and a
        // wave's LDS operations retire in order, so a later ds_read of the wave sees its earlier ds_write whatever lane issued it — what is needed
        // is that hipcc keeps the order and, for tidiness, that the writes have left the queue.  The block-wide barriers that stood here (two per
        // pass: six in a GEGLU epilogue of the 192x320 tile) made all eight waves wait for the slowest at every phase change and kept the two waves
        // of a SIMD in the same phase — the VALU-heavy staging (bias, erf-GELU) of one could never run beside the LDS reads and global stores of
        // the other.  Every main loop ends on a block barrier behind its last LDS read, so the first staging write is safe.  -DAE_EPI_BLOCK_SYNC=1
        // restores the barriers (A/B builds).
#ifndef AE_EPI_BLOCK_SYNC
#define AE_EPI_BLOCK_SYNC 0
#endif
        auto epi_sync = [&]() __attribute__((always_inline)) {
            if (AE_EPI_BLOCK_SYNC) __syncthreads();
            else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        };
        // sub2: GEMM row m = low-resolution pixel (b, y, x) -> output pixel (b, 2 y + py, 2 x + px) of the [B, 2H, 2W] map (whole rows of N channels move)
        auto orow = [&](int m) __attribute__((always_inline)) -> long {
            if (!(AMODE == A_CONV3 && p.sub2)) return (long)m;
            const int hw = p.H * p.Wd;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / p.Wd, x = rem - y * p.Wd;
            return ((long)(b * 2 * p.H + 2 * y + py) * (2 * p.Wd) + 2 * x + px);
        };
        const bool staged = (n_out % 8 == 0) && (p.ldc % 8 == 0) && (!p.res || (p.ldr % 8 == 0 && (reinterpret_cast<uintptr_t>(p.res) & 15) == 0));
        if (staged) {
            float* st = reinterpret_cast<float*>(smem_raw) + wmn * (WMP * WN);
            // Bias (and, per row fragment, the time-embedding vector) of this lane's 4 columns per fragment: ONE 16-byte load each, issued
            // together ahead of the arithmetic.  Round 1 loaded every value as a scalar right before its use — hipcc waits (vmcnt(0)) after
            // each: 120 dependent L2 round trips per wave on the 192x320 tile, 39 k of the kernel's ~190 k cycles (tools/ubench/conv_lab.hip).
            const bool bias_v4 = (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
            const bool av_v4 = (reinterpret_cast<uintptr_t>(p.addvec) & 15) == 0 && (p.ldav & 3) == 0;
            int ncol[FN];
            f32x4 bz[FN];
            // XE 2: s of this lane's 4 columns per fragment (bz holds c; both 16-byte aligned: checked by the launcher), loaded in front of the main
            // loop.  The 192x320 tile has no registers for 2 x 40 of them beside its 120 accumulators (hipcc spilled 275): there the block's s and
            // c sit in LDS behind the ring and are read at their use.
            f32x4 sz[(XE == 2 && !SC_LDS) ? FN : 1];
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if (geglu) ncol[j] = min(n0 + wn * WN + (j & ~1) * 16 + lg * 4, p.N - 20) + (j & 1) * 16;   // 'a' rows, gate rows at +16
                else ncol[j] = min(n0 + wn * WN + j * 16 + lg * 4, p.N - 4);
                bz[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (XE == 2) {
                    if constexpr (!SC_LDS) { sz[j] = szp[j]; bz[j] = bzp[j]; }
                } else if (p.bias) {
                    if (bias_v4) bz[j] = *reinterpret_cast<const f32x4*>(p.bias + ncol[j]);
                    else bz[j] = (f32x4){p.bias[ncol[j]], p.bias[ncol[j] + 1], p.bias[ncol[j] + 2], p.bias[ncol[j] + 3]};
                }
                if (geglu && (j & 1) == 0 && !SC_LDS) {   // GEGLU: fragment j is an 'a' fragment — its bias (XE 2: c) enters as 0.5 bias, see geglu_half_f
                    bz[j][0] *= 0.5f; bz[j][1] *= 0.5f; bz[j][2] *= 0.5f; bz[j][3] *= 0.5f;
                }
            }
            // XE 2: (s, c) of fragment j's four columns (local column wn * WN + 16 j + 4 lg in either packing: (j & ~1) * 16 + (j & 1) * 16 = 16 j)
            auto ln_sc = [&](int j, f32x4& sj, f32x4& cj) __attribute__((always_inline)) {
                if constexpr (SC_LDS) {
                    const int col = wn * WN + j * 16 + lg * 4;
                    sj = *reinterpret_cast<const f32x4*>(sc_lds + col);
                    cj = *reinterpret_cast<const f32x4*>(sc_lds + BN + col);
                } else if constexpr (XE == 2) {
                    sj = sz[j]; cj = bz[j];
                }
            };
#pragma unroll
            for (int ps = 0; ps < PASSES; ++ps) {
                if (ps > 0) epi_sync();
                // The staging loop is instantiated once per epilogue kind: inside it nothing depends on a runtime flag.  (Round 1 tested p.epi and
                // the addvec pointer per VALUE: four scalar branches around every one of the 120 values a wave stages on the 192x320 tile.)
                auto stage = [&](auto EPI_TAG, auto AV_TAG) {
                    constexpr int epi = decltype(EPI_TAG)::value;
                    constexpr bool has_av = decltype(AV_TAG)::value;
#pragma unroll
                    for (int ii = 0; ii < FMP; ++ii) {
                        const int i = ps * FMP + ii;
                        const int rl = ii * 16 + l15;  // row inside this pass
                        f32x4 az[FN];  // the row's time-embedding values; summation order as before: (acc + bias) + vector
                        if constexpr (has_av) {
                            const int m = min(m0 + wm * WM + i * 16 + l15, p.M - 1);
                            const float* av = p.addvec + (long)(m / p.rows_per_batch) * p.ldav;
#pragma unroll
                            for (int j = 0; j < FN; ++j) {
                                if (av_v4) az[j] = *reinterpret_cast<const f32x4*>(av + ncol[j]);
                                else az[j] = (f32x4){av[ncol[j]], av[ncol[j] + 1], av[ncol[j] + 2], av[ncol[j] + 3]};
                            }
                        }
                        if constexpr (epi == EPI_GEGLU) {
                            if constexpr (FN % 2 == 0) {
#pragma unroll
                                for (int j = 0; j < FN; j += 2) {
                                    f32x4 o;
                                    f32x4 sa, ca, sg, cg;
                                    if constexpr (XE == 2) { ln_sc(j, sa, ca); ln_sc(j + 1, sg, cg); }
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        float ah, g;   // ah = 0.5 a: the halves of the 'a' bias / c values were taken where they were loaded (bz, the LDS copy of c)
                                        if constexpr (XE == 2) {   // LayerNorm fold: rstd (acc - mu s) + c
                                            ah = fmaf(acc[i][j][r], 0.5f * ln_r[i], fmaf(0.5f * ln_t[i], sa[r], ca[r]));
                                            g = fmaf(acc[i][j + 1][r], ln_r[i], fmaf(ln_t[i], sg[r], cg[r]));
                                        } else {
                                            ah = fmaf(acc[i][j][r], 0.5f, bz[j][r]);
                                            g = acc[i][j + 1][r] + bz[j + 1][r];
                                        }
                                        o[r] = geglu_half_f(ah, g);
                                    }
                                    const int ch = (j / 2) * 4 + lg;
                                    *reinterpret_cast<f32x4*>(st + rl * WN + (((ch + rl) % NCH) << 2)) = o;
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < FN; ++j) {
                                f32x4 o;
                                f32x4 sj, cj;
                                if constexpr (XE == 2) ln_sc(j, sj, cj);
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float v;
                                    if constexpr (XE == 2) v = fmaf(acc[i][j][r], ln_r[i], fmaf(ln_t[i], sj[r], cj[r]));   // LayerNorm fold: rstd (acc - mu s) + c
                                    else v = acc[i][j][r] + bz[j][r];
                                    if constexpr (has_av) v += az[j][r];
                                    if constexpr (epi == EPI_SILU) v = silu_f(v);
                                    else if constexpr (epi == EPI_GELU) v = gelu_erf_f(v);
                                    else if constexpr (epi == EPI_RELU) v = fmaxf(v, 0.f);
                                    o[r] = v;
                                }
                                const int ch = j * 4 + lg;
                                *reinterpret_cast<f32x4*>(st + rl * WN + (((ch + rl) % NCH) << 2)) = o;
                            }
                        }
                    }
                };
                if (writer) {
                    using T = std::true_type;
                    using F = std::false_type;
                    if (geglu) stage(std::integral_constant<int, EPI_GEGLU>{}, F{});
                    else if (p.epi == EPI_NONE && p.addvec) stage(std::integral_constant<int, EPI_NONE>{}, T{});
                    else if (p.epi == EPI_NONE) stage(std::integral_constant<int, EPI_NONE>{}, F{});
                    else if (p.epi == EPI_SILU && p.addvec) stage(std::integral_constant<int, EPI_SILU>{}, T{});
                    else if (p.epi == EPI_SILU) stage(std::integral_constant<int, EPI_SILU>{}, F{});
                    else if (p.epi == EPI_GELU && p.addvec) stage(std::integral_constant<int, EPI_GELU>{}, T{});
                    else if (p.epi == EPI_GELU) stage(std::integral_constant<int, EPI_GELU>{}, F{});
                    else if (p.addvec) stage(std::integral_constant<int, EPI_RELU>{}, T{});
                    else stage(std::integral_constant<int, EPI_RELU>{}, F{});
                }
                GL_T(6);
                epi_sync();
                const int ow = geglu ? WN / 2 : WN;      // output columns of this wave's tile
                const int och = ow / 8;                  // 8-column output chunks per row
                const int nbase = geglu ? (n0 + wn * WN) / 2 : n0 + wn * WN;
                if constexpr (CS) {
                    // ---- output + per-channel statistics of this 32-row slab (bf16 output, no GEGLU: checked by the launcher)
                    constexpr int OCH = WN / 8, R = 64 / OCH;       // lanes = R rows x OCH column chunks (R * OCH <= 64)
                    const int oc = lane % OCH, rr = lane / OCH;
                    const int n = nbase + oc * 8;
                    float cs[8], cq[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
#pragma unroll
                    for (int r0 = 0; r0 < 32; r0 += R) {
                        const int rl = r0 + rr;
                        const int m = m0 + wm * WM + ps * 32 + rl;
                        if (rr < R && rl < 32 && m < p.M && n < n_out) {
                            const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + rl * WN + (((2 * oc + rl) % NCH) << 2));
                            const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + rl * WN + (((2 * oc + 1 + rl) % NCH) << 2));
                            float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                            if (p.res) {
                                const u32x4 rr4 = *reinterpret_cast<const u32x4*>(p.res + (long)m * p.ldr + n);
                                o[0] += bf16lo(rr4.x); o[1] += bf16hi(rr4.x); o[2] += bf16lo(rr4.y); o[3] += bf16hi(rr4.y);
                                o[4] += bf16lo(rr4.z); o[5] += bf16hi(rr4.z); o[6] += bf16lo(rr4.w); o[7] += bf16hi(rr4.w);
                            }
                            const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
                            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + orow(m) * p.ldc + n) = pk;
                            // statistics of the STORED (bf16-rounded) values: what the consuming GroupNorm reads
                            const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = bf16lo(w4[e]), c = bf16hi(w4[e]);
                                cs[2 * e] += a; cq[2 * e] += a * a;
                                cs[2 * e + 1] += c; cq[2 * e + 1] += c * c;
                            }
                        }
                    }
                    // fold the R row-lanes of every column chunk through this wave's own staging slice: the wave's LDS operations retire
                    // in order, so the writes below cannot overtake the loop's reads and the reads after them see the writes
                    asm volatile("" ::: "memory");
                    if (rr < R) {
                        float* d0 = st + (rr * 2) * WN + oc * 8;
                        *reinterpret_cast<f32x4*>(d0) = (f32x4){cs[0], cs[1], cs[2], cs[3]};
                        *reinterpret_cast<f32x4*>(d0 + 4) = (f32x4){cs[4], cs[5], cs[6], cs[7]};
                        *reinterpret_cast<f32x4*>(d0 + WN) = (f32x4){cq[0], cq[1], cq[2], cq[3]};
                        *reinterpret_cast<f32x4*>(d0 + WN + 4) = (f32x4){cq[4], cq[5], cq[6], cq[7]};
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const int mslab = m0 + wm * WM + ps * 32;   // multiple of 32: BM, WM are
                    for (int c2 = lane; c2 < WN / 2; c2 += 64) {
                        const int nc = nbase + 2 * c2;
                        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const f32x2 a = *reinterpret_cast<const f32x2*>(st + (r * 2) * WN + 2 * c2);
                            const f32x2 c = *reinterpret_cast<const f32x2*>(st + (r * 2 + 1) * WN + 2 * c2);
                            s0 += a[0]; s1 += a[1]; q0 += c[0]; q1 += c[1];
                        }
                        // sub2: the statistics buffer is laid out for the OUTPUT map, sample-major: sample b owns 4 hw / 32 slabs, parity `split` of them the
                        // run [split hw / 32, (split + 1) hw / 32) — which 32 rows of a sample a slab sums is immaterial to the GroupNorm's per-sample fold
                        long slab = mslab >> 5;
                        if (AMODE == A_CONV3 && p.sub2) {
                            const int spp = (p.H * p.Wd) >> 5;
                            const int bs = (int)slab / spp;
                            slab = (long)bs * 4 * spp + (long)split * spp + ((int)slab - bs * spp);
                        }
                        if (mslab < p.M && nc < n_out)
                            *reinterpret_cast<f32x4*>(p.colstats + (slab * n_out + nc) * 2) = (f32x4){s0, q0, s1, q1};
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else if constexpr (XE == 1) {
                    // ---- bf16 output + the row statistics of this wave's 64 columns (checked by the launcher: bf16 output, no GEGLU, N % 64 == 0).
                    // Eight consecutive lanes hold one row's eight 8-column chunks; they meet through three cross-lane adds (every lane of the
                    // wave takes part: the validity test guards the memory operations only), lane 0 of the eight writes the pair.
                    static_assert(XE != 1 || (WMP * 8) % 64 == 0, "row statistics: whole 8-row groups per pass");
                    const int nslice = p.N >> 6;
#pragma unroll
                    for (int it0 = 0; it0 < WMP * 8; it0 += 64) {
                        const int it = it0 + lane;
                        const int rl = it >> 3, oc = it & 7;
                        const int m = m0 + wm * WM + ps * WMP + rl, n = nbase + oc * 8;
                        const bool ok = m < p.M && n < n_out;
                        float rs = 0.f, rq = 0.f;
                        if (ok) {
                            const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + rl * WN + (((2 * oc + rl) % NCH) << 2));
                            const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + rl * WN + (((2 * oc + 1 + rl) % NCH) << 2));
                            float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                            if (p.res) {
                                const u32x4 rr = *reinterpret_cast<const u32x4*>(p.res + (long)m * p.ldr + n);
                                o[0] += bf16lo(rr.x); o[1] += bf16hi(rr.x); o[2] += bf16lo(rr.y); o[3] += bf16hi(rr.y);
                                o[4] += bf16lo(rr.z); o[5] += bf16hi(rr.z); o[6] += bf16lo(rr.w); o[7] += bf16hi(rr.w);
                            }
                            const u32x4 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
                            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = pk;
                            // statistics of the STORED (bf16-rounded) values: what the consuming GEMM multiplies
                            const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = bf16lo(w4[e]), c = bf16hi(w4[e]);
                                rs += a + c;
                                rq += a * a + c * c;
                            }
                        }
                        rs += __shfl_xor(rs, 1, 64); rq += __shfl_xor(rq, 1, 64);
                        rs += __shfl_xor(rs, 2, 64); rq += __shfl_xor(rq, 2, 64);
                        rs += __shfl_xor(rs, 4, 64); rq += __shfl_xor(rq, 4, 64);
                        if (ok && oc == 0) *reinterpret_cast<f32x2*>(p.rowstats + ((long)m * nslice + (nbase >> 6)) * 2) = (f32x2){rs, rq};
                    }
                } else
                for (int it = lane; it < (writer ? WMP * och : 0); it += 64) {
                    const int rl = it / och, oc = it - rl * och;
                    const int m = m0 + wm * WM + ps * WMP + rl, n = nbase + oc * 8;
                    if (m >= p.M || n >= n_out) continue;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(st + rl * WN + (((2 * oc + rl) % NCH) << 2));
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>(st + rl * WN + (((2 * oc + 1 + rl) % NCH) << 2));
                    float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if (p.res) {
                        const u32x4 rr = *reinterpret_cast<const u32x4*>(p.res + (long)m * p.ldr + n);
                        o[0] += bf16lo(rr.x); o[1] += bf16hi(rr.x); o[2] += bf16lo(rr.y); o[3] += bf16hi(rr.y);
                        o[4] += bf16lo(rr.z); o[5] += bf16hi(rr.z); o[6] += bf16lo(rr.w); o[7] += bf16hi(rr.w);
                    }
                    if (p.out_f32) {
                        float* dst = reinterpret_cast<float*>(p.C) + orow(m) * p.ldc + n;
                        *reinterpret_cast<f32x4*>(dst) = (f32x4){o[0], o[1], o[2], o[3]};
                        *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){o[4], o[5], o[6], o[7]};
                    } else {
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(p.C) + orow(m) * p.ldc + n) =
                            (u32x4){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
                    }
                }
                GL_T(7);
            }
            GL_T(4);
            return;
        }
    }

    if (!writer) return;
    // ---- direct epilogue (fallback for N % 8 != 0 or unaligned rows, e.g. the 320 -> 4 output conv) ---------------------
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int m = m0 + wm * WM + i * 16 + l15;
        if (m >= p.M) continue;
        const float* av = p.addvec ? p.addvec + (long)(m / p.rows_per_batch) * p.ldav : nullptr;
        if (p.epi == EPI_GEGLU) {
#pragma unroll
            for (int j = 0; j + 1 < FN; j += 2) {
                const int na = n0 + wn * WN + j * 16 + lg * 4;  // packed 'a' rows; gate rows at +16
                if (na >= p.N) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = acc[i][j][r] + (p.bias ? p.bias[na + r] : 0.f);
                    const float g = acc[i][j + 1][r] + (p.bias ? p.bias[na + 16 + r] : 0.f);
                    o[r] = a * gelu_erf_f(g);
                }
                const int nc = (n0 + wn * WN + j * 16) / 2 + lg * 4;
                u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + nc) = pk;
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = n0 + wn * WN + j * 16 + lg * 4;
            if (n >= p.N) continue;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[n + r];
                if (av) v += av[n + r];
                if (p.epi == EPI_SILU) v = silu_f(v);
                else if (p.epi == EPI_GELU) v = gelu_erf_f(v);
                                else if (p.epi == EPI_RELU) v = fmaxf(v, 0.f);
                o[r] = v;
            }
            if (p.res) {
                const u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + (long)m * p.ldr + n);
                o[0] += bf16lo(rr.x); o[1] += bf16hi(rr.x); o[2] += bf16lo(rr.y); o[3] += bf16hi(rr.y);
            }
            if (p.out_f32) {
                *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = (f32x4){o[0], o[1], o[2], o[3]};
            } else {
                u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = pk;
            }
        }
    }
}

// out[m][n] = sum_s partial[s][m][n] + bias[n] + addvec[m / rows_per_batch][n] (+ residual), fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p) {
    const long total = (long)p.M * p.N / 4;
    const long MN = (long)p.M * p.N;
    // every operand of an output quad is fetched as ONE vector load and all of them are issued before the first is used (the scalar
    // per-value form made hipcc wait after each load: 15 dependent round trips per quad in a kernel that is nothing but latency)
    const bool bias_v4 = (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
    const bool av_v4 = (reinterpret_cast<uintptr_t>(p.addvec) & 15) == 0 && (p.ldav & 3) == 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long e = i * 4;
        const int m = (int)(e / p.N), n = (int)(e % p.N);
        f32x4 w[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) w[s2] = s2 < p.splitk ? *reinterpret_cast<const f32x4*>(p.partial + (long)s2 * MN + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 bz = {0.f, 0.f, 0.f, 0.f}, az = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bz = bias_v4 ? *reinterpret_cast<const f32x4*>(p.bias + n) : (f32x4){p.bias[n], p.bias[n + 1], p.bias[n + 2], p.bias[n + 3]};
        if (p.addvec) {
            const float* av = p.addvec + (long)(m / p.rows_per_batch) * p.ldav + n;
            az = av_v4 ? *reinterpret_cast<const f32x4*>(av) : (f32x4){av[0], av[1], av[2], av[3]};
        }
        u32x2 rr = {0u, 0u};
        if (p.res) rr = *reinterpret_cast<const u32x2*>(p.res + (long)m * p.ldr + n);
        f32x4 v = w[0];
#pragma unroll
        for (int s2 = 1; s2 < 8; ++s2) {  // fixed summation order; splits beyond splitk contribute exact zeros
            if (s2 < p.splitk) { v[0] += w[s2][0]; v[1] += w[s2][1]; v[2] += w[s2][2]; v[3] += w[s2][3]; }
        }
        float o[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (p.bias) o[r] += bz[r];
            if (p.addvec) o[r] += az[r];
        }
        if (p.res) { o[0] += bf16lo(rr.x); o[1] += bf16hi(rr.x); o[2] += bf16lo(rr.y); o[3] += bf16hi(rr.y); }
        if (p.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = (f32x4){o[0], o[1], o[2], o[3]};
        else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + n) = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    }
}

// Stand-alone producer of the same per-channel statistics ([ceil(M/32)][N][2]: sum, sum of squares over each 32-row slab) for outputs
// whose kernel has no CS epilogue (split-K reduce, row-panel GEMM, small tiles): one read of the bf16 tensor.  Block = 256 threads =
// 32 column chunks (8 channels each) x 8 row lanes, 4 rows per thread in flight; grid = (slabs, ceil(N / 256)).
__global__ __launch_bounds__(256) void colstats_kernel(const bf16_t* x, long ld, int M, int N, float* out) {
    __shared__ float red[8][2][256];
    const int cc = threadIdx.x & 31, rr = threadIdx.x >> 5;
    const int n = blockIdx.y * 256 + cc * 8;
    const int r0 = blockIdx.x * 32;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (n < N) {
        u32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const u32x4*>(x + (long)min(r0 + rr + 8 * j, M - 1) * ld + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float keep = (r0 + rr + 8 * j < M) ? 1.f : 0.f;
            const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bf16lo(w[e]) * keep, c = bf16hi(w[e]) * keep;
                s[2 * e] += a; q[2 * e] += a * a;
                s[2 * e + 1] += c; q[2 * e + 1] += c * c;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[rr][0][cc * 8 + e] = s[e]; red[rr][1][cc * 8 + e] = q[e]; }
    __syncthreads();
    const int col = threadIdx.x;  // one column per thread, fixed summation order over the 8 row lanes
    if (blockIdx.y * 256 + col < N) {
        float a = 0.f, c = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a += red[r][0][col]; c += red[r][1][col]; }
        *reinterpret_cast<f32x2*>(out + ((long)blockIdx.x * N + blockIdx.y * 256 + col) * 2) = (f32x2){a, c};
    }
}

// tile choice: the largest tile that still fills the 256 CUs and does not waste >10% of N
int pick_tile(int M, int N) {
    const int cand[3][2] = {{128, 128}, {128, 64}, {64, 64}};
    // 128x160 divides every channel count of this UNet (320, 640, 960, 1280, 1920, ...); prefer it where 128x128 would
    // pad N (320, 960): 2.5x the A-tile reuse of 128x64
    if (N % 160 == 0 && N % 128 != 0 && (long)((M + 127) / 128) * (N / 160) >= 256) return 3;
    for (int c = 0; c < 3; ++c) {
        const long tm = (M + cand[c][0] - 1) / cand[c][0], tn = (N + cand[c][1] - 1) / cand[c][1];
        const double waste = (double)(tn * cand[c][1]) / (double)N;
        if (tm * tn >= 256 && waste <= 1.10) return c;
    }
    return 2;
}

// (tile, split-K) plan.  Launches that cannot fill the chip with 128x128 output tiles alone (16x16 / 8x8 latent levels:
// M = 3072 / 768 with K = 11520..23040) keep the big tile and cut K instead of shrinking the tile: fp32 partials + a
// fixed-order reduce.  tile: 0 128x128, 1 128x64, 2 64x64, 3 128x160.
struct Plan { int tile, splitk; };
Plan make_plan(int M, int N, int K, bool can_split) {
    const int t = pick_tile(M, N);
    const int kt = (K + BK - 1) / BK;
    static const int force_s = getenv("AE_GEMM_SPLITK") ? atoi(getenv("AE_GEMM_SPLITK")) : 0;  // dev knob
    // 16x16-level convs (UNet batch 12: M = 3072 = 64 tiles of 192x320) cut K four ways under the 192x320 tile (120 FLOP per LDS-DMA byte,
    // against 65 for the 128x128 tile the other split plans use): kbench 1280->1280 104.6 -> 97.5 us, 2560->1280 186 -> 162 us (1.12 PFLOP/s
    // incl. the reduce); in situ 15.35 -> 15.10 ms per UNet step (two runs each way).  Extending the rule to the 32x32 level's 128-tile
    // grids (split 2) gains on the K = 17280 convs in isolation (272 -> 241 us) and is neutral-to-negative inside the UNet: knob value 1
    // (3: only for >= 180 K tiles).  0 = round-1 plans.  tile id 4.
    // Round 5, re-measured under the ping-pong / slab loop: value 3 (the 32x32-level convs with >= 180 K tiles — 1920 -> 640 and 1280 -> 640 at UNet batch 12 — cut two ways onto
    // 256 blocks of the 192x320 tile instead of 480 blocks of 128x128) 12.828 / 12.792 -> 12.756 / 12.751 ms per UNet step in two alternating rounds, value 1 (every 128-tile
    // grid) 12.766 / 12.757 (profiles/r05_v11_t320_splitk_ab.txt).  Default 3.
    static const int t320_split = getenv("AE_CONV_T320_SPLITK") ? atoi(getenv("AE_CONV_T320_SPLITK")) : 3;
    if (t320_split && can_split && N % 320 == 0 && M % 192 == 0 && force_s <= 0) {
        const long t192 = (long)(M / 192) * (N / 320);
        if (t192 >= 32 && t192 <= (t320_split == 2 ? 64 : (t320_split == 3 && kt < 180 ? 64 : 128))) {
            int s = (int)(256 / t192);
            if (s > 8) s = 8;
            while (s > 1 && kt / s < 16) --s;
            if (s >= 2) return {4, s};
        }
    }
    if (!can_split || kt < 32 || t == 0 || t == 3) return {t, 1};
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const double waste128 = (double)(((N + 127) / 128) * 128) / (double)N;
    // small conv grids (<= 128 output tiles: 8x8 level, stride-2 convs, training batches): split only until the grid reaches one block
    // per CU and run those blocks under the three-stage ring, instead of splitting to two blocks per CU.  In situ: M = 768 convs
    // 46 -> 43 us, inference step unchanged, training step -3 %.  (For 240-tile grids the unsplit ring LOSES: 127 -> 158 us.)
    static const int conv_deep = getenv("AE_CONV_DEEP") ? atoi(getenv("AE_CONV_DEEP")) : 0;  // round 3: off — with the weights-ahead 128x128 kernel two blocks per CU win again (13.60 -> 13.55 ms, two runs each way)
    if (waste128 <= 1.10 && t128 < 256) {
        int s = force_s > 0 ? force_s : (int)((480 + t128 - 1) / t128);
        if (conv_deep && force_s <= 0 && t128 <= 128) s = (int)(256 / t128);   // grid <= 256: every block alone on its CU, latency hidden by the ring
        static const int smax = getenv("AE_CONV_SPLIT_MAX") ? atoi(getenv("AE_CONV_SPLIT_MAX")) : 8;   // A/B knob (round 5: 16 for the M = 256 convs of a training batch)
        if (s > smax) s = smax;
        if (s > kt / 8) s = kt / 8;
        if (s >= 2) return {0, s};
    }
    return {t, 1};
}

// Launch one instantiation; kernels that need more than 64 KiB of dynamic LDS get the opt-in attribute once.
// Plan query (ae_gemm_ln_plan): launch() runs its whole selection with g_plan_query set and launches nothing; the sites that hold a row-statistics /
// LayerNorm-fold instantiation report through `xe_done` in launch().  One selection code path for the launch and for the question "would it be covered".
thread_local int g_plan_query = 0;

template <typename Kern>
int launch_kernel(Kern kern, unsigned grid, int threads, size_t lds, hipStream_t stream, const GemmArgs& a, const char* what) {
    if (g_plan_query) return 0;
    if (lds > 64 * 1024) {
        static const void* done[32];
        static int ndone = 0;
        const void* fn = reinterpret_cast<const void*>(kern);
        bool seen = false;
        for (int i = 0; i < ndone; ++i) seen |= done[i] == fn;
        if (!seen) {
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                ae_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", lds);
                return AE_ERR_LAUNCH;
            }
            if (ndone < 32) done[ndone++] = fn;
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, a);
    return ae_check_launch(what);
}

template <int AMODE>
int launch(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    // each XCD's L2 sees a contiguous run of tile ids: put the tiles that share the LARGER operand panel next to each other.
    // Weights dominate at the 16x16 / 8x8 latent levels (M = 3072 / 768 rows against N x 9 Cin weights); the gather's 9x
    // re-reads of the activation hit L2 either way.
    static const int mfast_env = getenv("AE_GEMM_MFAST") ? atoi(getenv("AE_GEMM_MFAST")) : -1;  // tuning knob
    // measured (kbench, MI355X): M = 768 convs +15..17 %, M = 3072 x N = 10240 GEGLU GEMM +7 %; M = 3072 convs -2..-6 %, M = 12288 -3..-12 %
    a.m_fast = mfast_env >= 0 ? mfast_env : (AMODE == A_CONV3 ? (9L * a.N >= 8L * a.M) : ((long)a.N >= 3L * a.M)) ? 1 : 0;
    const int cand[4][2] = {{128, 128}, {128, 64}, {64, 64}, {128, 160}};
    static const int t160 = getenv("AE_GEMM_T160") ? atoi(getenv("AE_GEMM_T160")) : 1;  // tuning knob: 128x160 tile
    int pick = a.splitk > 1 ? 0 : pick_tile(a.M, a.N);  // a split plan always uses the 128x128 tile (make_plan)
    static const int force_tile = getenv("AE_GEMM_TILE") ? atoi(getenv("AE_GEMM_TILE")) : -1;  // dev knob: 0 128x128, 1 128x64, 2 64x64
    if (force_tile >= 0 && force_tile <= 2 && a.splitk <= 1) pick = force_tile;
    // measured (profiles/r01_kbench_t160.txt): 128x160 wins on the N=320 convs (-8..-17 %), loses on dense (4 vs 8 waves)
    if (pick == 3 && (!t160 || AMODE == A_DENSE || a.epi == EPI_GEGLU || a.splitk > 1)) pick = 1;
    // 8-wave (4x2) blocks: twice the waves per CU at the same LDS footprint -> twice the latency tolerance of the
    // one-tile-ahead pipeline.  Measured on MI355X (profiles/r01_kbench_w8.txt): dense GEMMs -15..-25 % time, the 128x64 conv
    // unchanged (+-2 %), so that one keeps 4 waves (larger wave tile, fewer LDS reads per MFMA).  AE_GEMM_W8=0 forces 4 waves.
    static const int w8 = getenv("AE_GEMM_W8") ? atoi(getenv("AE_GEMM_W8")) : 2;
    // two K groups of 2x2 waves on the 128x128 tile: pays when the K loop is long enough to amortise the final LDS reduction
    // (kbench, operands L2-hot: convs with >= 90 K tiles +4..7 %, GEMMs with 10-40 K tiles -5..-25 %).  Inside the UNet evaluation,
    // where every layer's weights arrive cold, the same rule LOSES 2.5 % of the step (in-situ A/B, one box), so it stays off:
    // AE_GEMM_WK = 0 off (default), 1 rule, 2 always.
    static const int wk_knob = getenv("AE_GEMM_WK") ? atoi(getenv("AE_GEMM_WK")) : 0;
    const int kt_block = ((a.K + BK - 1) / BK + a.splitk - 1) / a.splitk;
    const bool wk_env = wk_knob == 2 || (wk_knob == 1 && AMODE == A_CONV3 && kt_block >= 60);
    const bool conv = AMODE == A_CONV3;
    const char* what = conv ? "ae_conv3x3_bf16" : "ae_gemm_bf16";
    // LDS-DMA loaders need whole-tile decisions: no K tail, no mixed-source tile, no upsample gather / padded channels
    static const int glds_env = getenv("AE_GEMM_GLDS") ? atoi(getenv("AE_GEMM_GLDS")) : 1;  // tuning knob (A/B on hardware)
    const bool glds = glds_env && (conv ? (a.Cin == a.CinPad) : (a.K % BK == 0 && (!a.A2 || a.Ksplit % BK == 0)));
    if (conv && a.kmajor && !glds) { ae_set_error("%s: the chunk-major K order exists only in the LDS-DMA loader (AE_GEMM_GLDS=0 is set)", what); return AE_ERR_UNSUPPORTED; }
    auto lds_of = [](int bm, int bn, int st) { return (size_t)st * (bm + bn) * BK * sizeof(bf16_t); };
    auto lds_wa = [](int bm, int bn) { return (size_t)(2 * bm + 3 * bn) * BK * sizeof(bf16_t); };  // weights-ahead: two A stages, three W stages
    // tuning knob (bit flags): weights two tiles ahead on the 8-wave 128x128 two-stage kernel — 1 convs, 2 dense (80 KiB per block: still two
    // blocks per CU).  Measured (profiles/r03_v30_weights_ahead.txt, final form): 32x32-level conv 640 -> 640 hot 101.9 -> 87.9 us, with the
    // weight rotated through a pool larger than the Infinity Cache 113.4 -> 89.6 us; 1920 -> 640: 327 -> 251 us; in situ 126 -> 96 us;
    // UNet step 14.70 -> 14.33 ms (two runs each way, one box).  Default on for both.
    static const int wa = getenv("AE_GEMM_WA") ? atoi(getenv("AE_GEMM_WA")) : AE_GEMM_WA_DEFAULT;
    // (Round 3's activations-ahead loop on the 192x320 tile — three A stages, the round-3 default for its dense / un-split conv launches: ff2 of the
    // 64x64 level 70.8 -> 63.0 us in situ — is superseded by the ping-pong loop below, which keeps its LDS layout (three A stages + two W stages =
    // 152 KiB); its instantiations and the AE_GEMM_AA knob are gone, the loop itself remains as the weights-ahead form of the 128x128 kernel.)
#ifdef AE_GEMM_TRACE
    auto lds_aa = [](int bm, int bn) { return (size_t)(3 * bm + 2 * bn) * BK * sizeof(bf16_t) + 8192; };   // + the trace area
#else
    auto lds_aa = [](int bm, int bn) { return (size_t)(3 * bm + 2 * bn) * BK * sizeof(bf16_t); };
#endif
    // tuning knob (bit flags): the ping-pong main loop (WA = 3, round 4) on the 192x320 tile — 1 un-split convs, 2 split-K convs, 4 dense non-GEGLU,
    // 8 GEGLU (4 x 2 waves).  Same LDS footprint as the activations-ahead loop (three A stages + two W stages).  16 / 32: the un-split / split-K
    // convs on the activation-slab form of the loop (WA = 4; two slab stages + two W stages = 132 KiB).
    static const int pp = getenv("AE_GEMM_PP") ? atoi(getenv("AE_GEMM_PP")) : AE_GEMM_PP_DEFAULT;
    int rc = 0;

    // 192x320 tile, one block per CU (128 KiB LDS), 8 waves with 96x80 (or 48x160 for the GEGLU column pairing) wave tiles:
    // 0.37-0.43 LDS fragment reads per MFMA instead of 0.75.  Used when the tile grid fills whole rounds of 256 CUs.
    static const int t320 = getenv("AE_GEMM_T320") ? atoi(getenv("AE_GEMM_T320")) : 11;  // tuning knob, bit flags: 1 convs, 2 GEGLU GEMMs with K >= 640, 4 GEGLU K = 320, 8 other dense K >= 640.  Isolated (kbench) the
    // dense flags gain 4..20 %; inside the UNet evaluation flag 4 is neutral-to-negative.  Flag 8 (round 3: default on) only ever fires where the
    // 192x320 grid fills whole rounds of CUs — at UNet batch 12 the 64x64-level ff2 (M = 49152, N = 320, K = 1280: 62.6 -> 49.2 us) and the
    // decoder's skip 1x1 convs (K = 960: 45.0 -> 36.4 us): 13.82 -> 13.72 ms per UNet step, two runs each way (profiles/r03_v10_ab.txt)
    // (the same tile for the K = 320 dense GEMMs of the 64x64 level was A/B-ed too: N = 320 44.5 vs 34 us, N = 960 78 vs 63 us —
    // five K iterations do not amortise the big tile's prologue / epilogue.)
    bool done = false;
    bool xe_done = false;   // a.xe: the selected instantiation carries the row-statistics / LayerNorm-fold epilogue
    // Column statistics for the consuming GroupNorm (GemmArgs::colstats): emitted by the epilogue where a CS instantiation exists (the
    // un-split 192x320 conv tile of the 64x64 level, the 8-wave 128x128 tile of the 32x32 level), by the stand-alone kernel otherwise.
    const bool cs_epi_ok = a.colstats && a.epi != EPI_GEGLU && !a.out_f32 && a.splitk <= 1 && a.N % 8 == 0 && a.ldc % 8 == 0 &&
                           (!a.res || (a.ldr % 8 == 0 && (reinterpret_cast<uintptr_t>(a.res) & 15) == 0));
    bool cs_done = false;
#ifdef AE_GEMM_ABLATE
    static const int lab_abl = getenv("AE_GEMM_ABL") ? atoi(getenv("AE_GEMM_ABL")) : 0;   // lab build only (tools/build_ablate.sh): main-loop ablations
#endif
    // (192x320 with FOUR waves of 96x160 — 0.27 LDS fragment reads per MFMA instead of 0.37, 240 accumulator registers per lane — was
    // instantiated and measured: hipcc places the accumulators in AGPRs and brackets the MFMAs with v_accvgpr moves, 166-206 TFLOP/s
    // against 911-1123 for the 8-wave form.  Bigger wave tiles need hand-scheduled AGPR code; not kept.)
    // the slab form of the ping-pong loop (WA = 4): whole image rows per 192-row tile, 16-row fragments inside one image row, at most 208 slab rows,
    // chunk-major K with every block's K range starting at a chunk boundary
    const bool slab_ok = conv && a.kmajor && a.stride == 1 && !a.ups && a.Cin == a.CinPad && a.H == a.Ho && a.Wd == a.Wo && a.Wd % 16 == 0 && 192 % a.Wd == 0 &&
                         192 / a.Wd <= a.H + 1 &&   // the slab's image row is y0 + jr with ONE wrap into the next sample (a short-wide map would need a true modulo)
                         (a.splitk <= 1 || kt_block % 9 == 0);
    const size_t lds_slab = (size_t)2 * ((192 + 192 / 16 + 1 + 7) / 8) * 1024 + (size_t)2 * 320 * BK * sizeof(bf16_t);
    if (conv && a.splitk > 1 && glds && make_plan(a.M, a.N, a.K, true).tile == 4) {  // split-K under the 192x320 tile (make_plan)
        const long t = (long)(a.M / 192) * (a.N / 320) * a.splitk;
        if ((pp & 32) && (pp & 2) && slab_ok) { if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 0, 4>, (unsigned)t, 512, lds_slab, stream, a, what); }
        else if ((pp & 2) && !a.ups) { if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 0, 3>, (unsigned)t, 512, lds_aa(192, 320), stream, a, what); }
        else rc = launch_kernel(gemm_kernel<192, 320, AMODE, 2, 4, true>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what);
        done = true;
    }
    if (!done && t320 && glds && a.splitk <= 1 && a.N % 320 == 0 && ((conv && (t320 & 1)) || (!conv && a.epi == EPI_GEGLU && a.K >= 640 && (t320 & 2)) || (!conv && a.epi == EPI_GEGLU && a.K < 640 && a.K >= 320 && (t320 & 4)) ||
                                                                     (!conv && a.epi != EPI_GEGLU && a.K >= 640 && (t320 & 8)))) {
        const long t = (long)((a.M + 191) / 192) * (a.N / 320);
        const double fill = (double)t / (double)(((t + 255) / 256) * 256) * ((double)a.M / (double)(((a.M + 191) / 192) * 192));
        // 192x320 waves 2x4, or 4x2 for GEGLU (pairs of 16-column fragments must sit in one wave).  A 96x320 variant for the
        // 32x32 level (M = 12288) measured 9-13 % slower than the 128x128 tile there and was dropped.
        // Round 5: a LayerNorm-fold launch whose 128x128 grid would leave a bad tail ALSO takes this tile at a three-quarter-full single round —
        // qkv of the 16x16 level, M = 3072 x N = 3840 x K = 1280: 720 tiles of 128x128 on 512 block slots = 1.41 rounds (42.7 us, the library 29.7:
        // profiles/r05_ev1_gemm_vs_library.txt) against 192 tiles of 192x320 in one round of the ping-pong loop.  It runs the 4x2-wave LayerNorm-fold
        // instantiation the GEGLU launches use (its epilogue kind is a run-time argument): no new device code.  AE_GEMM_T320_XE=0 turns the rule off (A/B).
        static const int t320_xe = getenv("AE_GEMM_T320_XE") ? atoi(getenv("AE_GEMM_T320_XE")) : 1;
        const long t128x = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
        const bool xe_tail = t320_xe && !conv && a.xe == 2 && a.epi != EPI_GEGLU && (pp & 8) && fill >= 0.74 && fill < 0.85 &&
                             (double)t128x / (double)(((t128x + 511) / 512) * 512) <= 0.72;
        if (fill >= 0.85 || xe_tail) {
            if (xe_tail) {
                if constexpr (AMODE == A_DENSE) { rc = launch_kernel(gemm_kernel<192, 320, A_DENSE, 4, 2, true, 1, 2, false, 0, 3, 2>, (unsigned)t, 512, lds_aa(192, 320) + 2 * 320 * sizeof(float), stream, a, what); xe_done = true; }
            } else
            if (a.epi == EPI_GEGLU && (pp & 8)) {
                if constexpr (AMODE == A_DENSE) {
                    if (a.xe == 2) { rc = launch_kernel(gemm_kernel<192, 320, A_DENSE, 4, 2, true, 1, 2, false, 0, 3, 2>, (unsigned)t, 512, lds_aa(192, 320) + 2 * 320 * sizeof(float), stream, a, what); xe_done = true; }
                    else rc = launch_kernel(gemm_kernel<192, 320, A_DENSE, 4, 2, true, 1, 2, false, 0, 3>, (unsigned)t, 512, lds_aa(192, 320), stream, a, what);
                }
            }
            else if (a.epi == EPI_GEGLU) rc = launch_kernel(gemm_kernel<192, 320, AMODE, 4, 2, true>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what);
            else if (conv && (pp & 1) && !a.ups) {   // (the nearest-x2 gather: its per-piece address arithmetic measured 283 vs 273 us under the first form of the ping-pong loop; a second attempt with
                                                     //  per-piece row / column offset tables — source pixel = virtual pixel >> 1 is not linear in the tap — measured 409 vs 330 us (hipcc kept the
                                                     //  tables in scratch memory, profiles/r04_v23_ups_pp.txt).  Two launches per evaluation; they stay on the round-3 loop.)
                if constexpr (AMODE == A_CONV3) {
                    // flag 16: the slab form of the loop (WA = 4) where its preconditions hold — chunk-major K, stride 1, K range starting at tap 0
                    // (whole image rows per tile, 16-row fragments inside one image row, at most 200 slab rows)
                    const bool slab = (pp & 16) && slab_ok;
                    if (slab && cs_epi_ok) { rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, true, 0, 4>, (unsigned)t, 512, lds_slab, stream, a, what); cs_done = true; }
                    else if (slab) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 0, 4>, (unsigned)t, 512, lds_slab, stream, a, what);
                    else if (cs_epi_ok) { rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, true, 0, 3>, (unsigned)t, 512, lds_aa(192, 320), stream, a, what); cs_done = true; }
                    else rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 0, 3>, (unsigned)t, 512, lds_aa(192, 320), stream, a, what);
                }
            } else if (!conv && (pp & 4)) {
                if constexpr (AMODE == A_DENSE) rc = launch_kernel(gemm_kernel<192, 320, A_DENSE, 2, 4, true, 1, 2, false, 0, 3>, (unsigned)t, 512, lds_aa(192, 320), stream, a, what);
            } else if (cs_epi_ok && AMODE == A_CONV3) {
                if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, true>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what);
                cs_done = true;
            }
#ifdef AE_GEMM_ABLATE
            else if (conv && lab_abl == 1) { if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 1>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what); }
            else if (conv && lab_abl == 2) { if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 2>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what); }
            else if (conv && lab_abl == 3) { if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 3>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what); }
            else if (conv && lab_abl == 4) { if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 4>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what); }
            else if (conv && lab_abl == 20) { if constexpr (AMODE == A_CONV3) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, false>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what); }  // register-staged loader
#endif
            else rc = launch_kernel(gemm_kernel<192, 320, AMODE, 2, 4, true>, (unsigned)t, 512, lds_of(192, 320, 2), stream, a, what);
            done = true;
        }
    }
    // (A 256x128 dense tile — one 8-wave block per CU, 64x64 wave tiles, 0.5 LDS fragment reads per MFMA — was built and A/B-ed in
    // situ: SAM ViT-H encoder 16.2 vs 14.25 ms, every M = 4096 / 4900 GEMM 15-25 % slower, UNet step +1 %.  Like the deeper LDS ring, it
    // trades the second resident block per CU for per-wave reuse, and the second block is worth more.  Not kept.)
    // Three-stage LDS ring (two tiles in flight, counted vmcnt + raw s_barrier) on the 128x128 tile, for dense GEMMs whose 128x128 grid
    // is 128..256 blocks: there is at most one block per CU anyway, so the 96 KiB ring costs no co-residency and the second tile in
    // flight hides what the missing second block would have hidden.  In-situ A/B (UNet batch 12, same box): M = 3072 x N = 1280 at
    // K = 5120 72.8 -> 59.5 us, K = 2560 40.7 -> 36.8, K = 1920 32.7 -> 29.0, K = 1280 25.9 -> 24.7 (K = 640: 20.7 -> 22.5, excluded);
    // step 17.31 -> 17.22 ms.  A fourth stage adds nothing.  The same ring on grids with MORE than one block per CU loses (it evicts the
    // second resident block: SAM M = 4096 x N = 1280 x K = 5120 with 320 blocks 90 -> 109 us), as does a 256x128 tile under it
    // (M = 4096 x N = 5120: 93 -> 112 us).  AE_GEMM_DEEP=0 turns it off.
    // (Round 4: the ping-pong loop on a 256x128 tile — 11.4 KiB of LDS-DMA per MFLOP against 15.6 for 128x128 — was instantiated for the 32x32-level
    // launches and measured in the lab: convs 96.3 / 299.5 us against 94.8 / 293.0 for the two-blocks-per-CU 128x128 kernel, ff2 40.3 vs 43.4, qkv 44.0
    // vs 41.2 (profiles/r04_v17_pp256x128_experiment.txt): its 16-MFMA intervals are too short for the two barriers each costs.  Not kept.)
    static const int deep_pref = getenv("AE_GEMM_DEEP") ? atoi(getenv("AE_GEMM_DEEP")) : 1;
    if (!done && deep_pref && !conv && glds && a.splitk <= 1 && a.epi != EPI_GEGLU && a.N % 128 == 0 && a.K >= 1280) {
        const long t128 = (long)((a.M + 127) / 128) * (a.N / 128);
        const long t192 = (long)((a.M + 191) / 192) * (a.N / 128);
        if (t128 >= 128 && t128 <= 256) {
            if (a.xe && AMODE == A_DENSE) {
                if constexpr (AMODE == A_DENSE) {
                    if (a.xe == 1) rc = launch_kernel(gemm_kernel<128, 128, A_DENSE, 4, 2, true, 1, 3, false, 0, 0, 1>, (unsigned)t128, 512, lds_of(128, 128, 3), stream, a, what);
                    else rc = launch_kernel(gemm_kernel<128, 128, A_DENSE, 4, 2, true, 1, 3, false, 0, 0, 2>, (unsigned)t128, 512, lds_of(128, 128, 3), stream, a, what);
                    xe_done = true;
                }
            } else
            rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 3>, (unsigned)t128, 512, lds_of(128, 128, 3), stream, a, what);
            done = true;
        } else if (t128 > 256 && t192 >= 128 && t192 <= 256 && a.K >= 2560) {
            // a taller tile that brings the grid down to one block per CU (SAM: M = 4096 x N = 1280 x K = 5120, 320 -> 220 blocks:
            // 89 -> 80 us; at K = 1280 it is neutral, and a 256x128 tile for M = 4900 loses: 32.5 -> 38 us)
            rc = launch_kernel(gemm_kernel<192, 128, AMODE, 4, 2, true, 1, 3>, (unsigned)t192, 512, lds_of(192, 128, 3), stream, a, what);
            done = true;
        }
    }
    if (!done) {
        const int BM = cand[pick][0], BN = cand[pick][1];
        const unsigned grid = (unsigned)((long)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) * a.splitk);
#define AE_LAUNCH(BM_, BN_, WM_, WN_, THREADS)                                                                                            \
    do {                                                                                                                                  \
        if (glds) rc = launch_kernel(gemm_kernel<BM_, BN_, AMODE, WM_, WN_, true>, grid, THREADS, lds_of(BM_, BN_, 2), stream, a, what);   \
        else rc = launch_kernel(gemm_kernel<BM_, BN_, AMODE, WM_, WN_, false>, grid, THREADS, lds_of(BM_, BN_, 2), stream, a, what);       \
    } while (0)
        // 64x64 grids up to three blocks per CU keep the ring too (its 48 KiB leave room for three resident blocks): training batch
        // M = 1024 x N = 1280 x K = 1280 (320 blocks) 31.7 -> ~20 us, training step 29.0 -> 28.0 ms.  AE_GEMM_DEEP64_MAX=256 restores round 1.
        static const int deep64_max = getenv("AE_GEMM_DEEP64_MAX") ? atoi(getenv("AE_GEMM_DEEP64_MAX")) : 768;
        static const int conv_deep_l = getenv("AE_CONV_DEEP") ? atoi(getenv("AE_CONV_DEEP")) : 0;
        // (a FOURTH stage — three tiles in flight, 128 KiB — measured the same on the 8x8-level convs: 41.9 vs 40.4 us; these launches are
        // not latency-bound by ring depth but by LDS bandwidth: 0.75 ds_read_b128 per MFMA x 8 cycles each against 16-cycle MFMAs)
        if (conv && conv_deep_l && pick == 0 && glds && grid <= 256)
            rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 3>, grid, 512, lds_of(128, 128, 3), stream, a, what);
        else if (pick == 3) AE_LAUNCH(128, 160, 2, 2, 256);
        else if (pick == 0 && w8 == 1) AE_LAUNCH(128, 128, 2, 4, 512);
        else if (pick == 0 && wk_env && glds) rc = launch_kernel(gemm_kernel<128, 128, AMODE, 2, 2, true, 2>, grid, 512, lds_of(128, 128, 2), stream, a, what);
        else if (pick == 0 && w8 == 2 && cs_epi_ok && glds) {
            if (wa & (conv ? 1 : 2)) rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 2, true, 0, 1>, grid, 512, lds_wa(128, 128), stream, a, what);
            else rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 2, true>, grid, 512, lds_of(128, 128, 2), stream, a, what);
            cs_done = true;
        } else if (pick == 0 && w8 == 2 && glds && (wa & (conv ? 1 : 2)) && kt_block >= 3) {
            if (a.xe && AMODE == A_DENSE) {
                if constexpr (AMODE == A_DENSE) {
                    if (a.xe == 1) rc = launch_kernel(gemm_kernel<128, 128, A_DENSE, 4, 2, true, 1, 2, false, 0, 1, 1>, grid, 512, lds_wa(128, 128), stream, a, what);
                    else rc = launch_kernel(gemm_kernel<128, 128, A_DENSE, 4, 2, true, 1, 2, false, 0, 1, 2>, grid, 512, lds_wa(128, 128), stream, a, what);
                    xe_done = true;
                }
            } else
            rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 2, false, 0, 1>, grid, 512, lds_wa(128, 128), stream, a, what);
        }
#ifdef AE_GEMM_ABLATE
        else if (pick == 0 && w8 == 2 && glds && lab_abl == 1) rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 2, false, 1>, grid, 512, lds_of(128, 128, 2), stream, a, what);
        else if (pick == 0 && w8 == 2 && glds && lab_abl == 2) rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 2, false, 2>, grid, 512, lds_of(128, 128, 2), stream, a, what);
        else if (pick == 0 && w8 == 2 && glds && lab_abl == 3) rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 2, false, 3>, grid, 512, lds_of(128, 128, 2), stream, a, what);
        else if (pick == 0 && w8 == 2 && glds && lab_abl == 4) rc = launch_kernel(gemm_kernel<128, 128, AMODE, 4, 2, true, 1, 2, false, 4>, grid, 512, lds_of(128, 128, 2), stream, a, what);
#endif
        else if (pick == 0 && w8 == 2) AE_LAUNCH(128, 128, 4, 2, 512);
        else if (pick == 0) AE_LAUNCH(128, 128, 2, 2, 256);
        else if (pick == 1 && w8 && !conv) AE_LAUNCH(128, 64, 4, 2, 512);
        else if (pick == 1) AE_LAUNCH(128, 64, 2, 2, 256);
        else if (deep_pref && glds && !conv && a.splitk <= 1 && grid <= (unsigned)deep64_max && a.K >= 1280)
            // the same ring for small 64x64 grids (8x8 level, training batches): M = 768 x N = 1280 at K = 1280 / 2560 / 5120:
            // 23.3 -> 16.4, 37.4 -> 23.2, 67.3 -> 38.7 us; lower K thresholds and a fourth stage measured the same
            rc = launch_kernel(gemm_kernel<64, 64, AMODE, 2, 2, true, 1, 3>, grid, 256, lds_of(64, 64, 3), stream, a, what);
        else AE_LAUNCH(64, 64, 2, 2, 256);
#undef AE_LAUNCH
    }
    if (rc) return rc;
    if (a.xe && !xe_done) {
        // (ops.py asks ae_gemm_ln_plan first, so this is a caller error; ae_gemm_ln_bf16 runs this selection as a query before it launches, so nothing was written)
        ae_set_error("%s: no %s instantiation for M=%d N=%d K=%d epilogue %d (ae_gemm_ln_plan says which shapes are covered)", what,
                     a.xe == 1 ? "row-statistics" : "LayerNorm-fold", a.M, a.N, a.K, a.epi);
        return AE_ERR_UNSUPPORTED;
    }
    if (g_plan_query) return 0;
    if (a.splitk > 1 && !a.defer_reduce) {
        long nb = ((long)a.M * a.N / 4 + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a);
        rc = ae_check_launch("ae_conv3x3_bf16(split-K reduce)");
        if (rc) return rc;
    }
    if (a.colstats && !cs_done) return ae_launch_colstats(reinterpret_cast<const bf16_t*>(a.C), a.ldc, a.M, a.epi == EPI_GEGLU ? a.N / 2 : a.N, a.colstats, stream);
    return rc;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

int ae_launch_colstats(const bf16_t* x, long ld, int M, int N, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(colstats_kernel, dim3((unsigned)((M + 31) / 32), (unsigned)((N + 255) / 256)), dim3(256), 0, stream, x, ld, M, N, out);
    return ae_check_launch("column statistics");
}

namespace {
// the LayerNorm-fold extras of one launch (all null / zero: a plain ae_gemm_bf16)
struct LnExtras {
    float* rowstats_out = nullptr;
    const float* ln_stats = nullptr;
    const float* ln_colsum = nullptr;
    int ln_parts = 0;
    float ln_eps = 0.f;
};
int gemm_entry(const void* A, long lda, const void* A2, long lda2, int Ksplit, const void* W, long ldw,
               void* C, long ldc, int M, int N, int K, const float* bias, const void* residual, long ldr,
               const float* addvec, long addvec_ld, int rows_per_batch, int epilogue, int out_f32, float* colstats, const LnExtras& x, void* stream);
}  // namespace

extern "C" int ae_gemm_bf16(const void* A, long lda, const void* A2, long lda2, int Ksplit, const void* W, long ldw,
                            void* C, long ldc, int M, int N, int K, const float* bias, const void* residual, long ldr,
                            const float* addvec, long addvec_ld, int rows_per_batch, int epilogue, int out_f32, float* colstats, void* stream) {
    return gemm_entry(A, lda, A2, lda2, Ksplit, W, ldw, C, ldc, M, N, K, bias, residual, ldr, addvec, addvec_ld, rows_per_batch, epilogue, out_f32, colstats, LnExtras{}, stream);
}

extern "C" int ae_gemm_ln_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                               const void* residual, long ldr, int epilogue, float* rowstats_out, const float* ln_stats, int ln_parts,
                               const float* ln_colsum, float ln_eps, void* stream) {
    AE_REQUIRE((rowstats_out != nullptr) != (ln_stats != nullptr), "ae_gemm_ln_bf16: exactly one of rowstats_out (emit) and ln_stats (consume) must be given");
    AE_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "ae_gemm_ln_bf16: null pointer or empty shape");
    {   // K = 320 (the 64x64 UNet level): the row-panel kernel's fold forms (round 5); AE_ERR_UNSUPPORTED = outside its envelope, the tiled plan decides
        const int rc = ae_rowpanel_fold_launch(A, lda, W, ldw, C, ldc, M, N, K, bias, residual, ldr, epilogue, rowstats_out, ln_stats, ln_parts, ln_colsum, ln_eps, stream);
        if (rc != AE_ERR_UNSUPPORTED) return rc;
    }
    LnExtras x;
    x.rowstats_out = rowstats_out; x.ln_stats = ln_stats; x.ln_colsum = ln_colsum; x.ln_parts = ln_parts; x.ln_eps = ln_eps;
    return gemm_entry(A, lda, nullptr, 0, 0, W, ldw, C, ldc, M, N, K, bias, residual, ldr, nullptr, 0, 0, epilogue, 0, nullptr, x, stream);
}

extern "C" int ae_gemm_ln_plan(int M, int N, int K, int epilogue, int mode) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 8 || (mode != 1 && mode != 2) || epilogue < EPI_NONE || epilogue > EPI_RELU) return 0;
    if (ae_rowpanel_fold_covers(M, N, K, epilogue, mode)) return 1;   // the row-panel kernel's fold forms (K = 320)
    if (N % 64 || (mode == 1 && (epilogue == EPI_GEGLU || N > 1280)) || (mode == 2 && (K % 64 || K > 1280))) return 0;   // the consumer sums up to 20 slices per row
    if ((long)M * K * 2 >= (1L << 31) || (long)N * K * 2 >= (1L << 31)) return 0;
    GemmArgs a{};
    // (addresses are never dereferenced under g_plan_query: launch_kernel returns before the launch)
    a.A = a.W = reinterpret_cast<const bf16_t*>(uintptr_t(256)); a.C = reinterpret_cast<void*>(uintptr_t(256));
    a.M = M; a.N = N; a.K = K; a.Ksplit = K;
    a.lda = K; a.ldw = K; a.ldc = epilogue == EPI_GEGLU ? N / 2 : N; a.ldav = N;
    a.epi = epilogue; a.rows_per_batch = 1; a.splitk = 1; a.xe = mode;
    a.a_bytes = (unsigned)((long)M * K * 2); a.w_bytes = (unsigned)((long)N * K * 2);
    g_plan_query = 1;
    const int rc = launch<A_DENSE>(a, nullptr);
    g_plan_query = 0;
    return rc == 0 ? 1 : 0;
}

namespace {
int gemm_entry(const void* A, long lda, const void* A2, long lda2, int Ksplit, const void* W, long ldw,
               void* C, long ldc, int M, int N, int K, const float* bias, const void* residual, long ldr,
               const float* addvec, long addvec_ld, int rows_per_batch, int epilogue, int out_f32, float* colstats, const LnExtras& x, void* stream) {
    AE_REQUIRE(A && W && C, "ae_gemm_bf16: null pointer");
    if (colstats) AE_REQUIRE(!out_f32 && N % 8 == 0 && ldc % 8 == 0 && (reinterpret_cast<uintptr_t>(colstats) & 15) == 0, "ae_gemm_bf16: column statistics need a bf16 output with N %% 8 == 0 and 16-byte rows");
    AE_REQUIRE(M > 0 && N > 0 && K > 0, "ae_gemm_bf16: M,N,K must be positive (got %d,%d,%d)", M, N, K);
    AE_REQUIRE(K % 8 == 0, "ae_gemm_bf16: K=%d must be a multiple of 8", K);
    AE_REQUIRE(N % 4 == 0, "ae_gemm_bf16: N=%d must be a multiple of 4", N);
    AE_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && aligned16(A) && aligned16(W), "ae_gemm_bf16: A/W rows must be 16-byte aligned");
    AE_REQUIRE(ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0, "ae_gemm_bf16: C must be 16-byte aligned, ldc %% 4 == 0");
    AE_REQUIRE(epilogue >= EPI_NONE && epilogue <= EPI_RELU, "ae_gemm_bf16: bad epilogue %d", epilogue);
    if (A2) {
        AE_REQUIRE(Ksplit > 0 && Ksplit < K && Ksplit % 8 == 0 && lda2 % 8 == 0 && aligned16(A2),
                   "ae_gemm_bf16: bad two-source split (Ksplit=%d, K=%d)", Ksplit, K);
    }
    if (epilogue == EPI_GEGLU) {
        AE_REQUIRE(N % 32 == 0 && !residual && !addvec && !out_f32, "ae_gemm_bf16: GEGLU needs N %% 32 == 0, bf16 out, no residual");
    }
    if (residual) AE_REQUIRE(ldr % 4 == 0 && (reinterpret_cast<uintptr_t>(residual) & 7) == 0, "ae_gemm_bf16: residual alignment");
    if (addvec) AE_REQUIRE(rows_per_batch > 0, "ae_gemm_bf16: rows_per_batch must be > 0 with addvec");
    GemmArgs a{};
    a.A = (const bf16_t*)A; a.A2 = (const bf16_t*)A2; a.W = (const bf16_t*)W; a.C = C;
    a.bias = bias; a.res = (const bf16_t*)residual; a.addvec = addvec;
    a.M = M; a.N = N; a.K = K; a.Ksplit = A2 ? Ksplit : K;
    a.lda = lda; a.lda2 = lda2; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr; a.ldav = addvec_ld > 0 ? addvec_ld : N;
    a.epi = epilogue; a.out_f32 = out_f32; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : 1;
    a.splitk = 1; a.partial = nullptr; a.colstats = colstats;
    a.a_bytes = (unsigned)((((long)M - 1) * lda + (A2 ? Ksplit : K)) * 2);
    a.a2_bytes = A2 ? (unsigned)((((long)M - 1) * lda2 + (K - Ksplit)) * 2) : 0u;
    a.w_bytes = (unsigned)((((long)N - 1) * ldw + K) * 2);
    AE_REQUIRE(((long)M * lda * 2) < (1L << 31) && ((long)N * ldw * 2) < (1L << 31) && (!A2 || ((long)M * lda2 * 2) < (1L << 31)),
               "ae_gemm_bf16: operands must be smaller than 2 GiB");
    if (x.rowstats_out || x.ln_stats) {
        const int n_out = epilogue == EPI_GEGLU ? N / 2 : N;
        // the staged (row-contiguous) epilogue is the one that carries both forms
        AE_REQUIRE(N % 64 == 0 && n_out % 8 == 0 && ldc % 8 == 0 && (!residual || (ldr % 8 == 0 && aligned16(residual))),
                   "ae_gemm_ln_bf16: N %% 64 == 0 and 16-byte aligned output / residual rows");
        if (x.rowstats_out) {
            AE_REQUIRE(epilogue != EPI_GEGLU && (reinterpret_cast<uintptr_t>(x.rowstats_out) & 7) == 0, "ae_gemm_ln_bf16: row statistics: no GEGLU, 8-byte aligned buffer");
            a.xe = 1; a.rowstats = x.rowstats_out;
        } else {
            AE_REQUIRE(x.ln_colsum && bias && x.ln_parts > 0 && x.ln_parts <= 20 && x.ln_eps >= 0.f, "ae_gemm_ln_bf16: LayerNorm fold needs s (ln_colsum), c (bias), 1..20 statistics slices per row (K <= 1280)");
            AE_REQUIRE(x.ln_parts * 64 == K, "ae_gemm_ln_bf16: the statistics must cover the row: ln_parts * 64 = %d, K = %d (the kernel divides the summed slices by K)", x.ln_parts * 64, K);
            AE_REQUIRE(aligned16(x.ln_colsum) && aligned16(bias) && (reinterpret_cast<uintptr_t>(x.ln_stats) & 7) == 0, "ae_gemm_ln_bf16: s / c must be 16-byte aligned, the statistics 8-byte aligned");
            a.xe = 2; a.ln_stats = x.ln_stats; a.ln_colsum = x.ln_colsum; a.ln_parts = x.ln_parts; a.ln_eps = x.ln_eps;
        }
        // the selection runs once without launching: a shape whose plan has no such epilogue is refused before anything is written
        g_plan_query = 1;
        const int rc = launch<A_DENSE>(a, nullptr);
        g_plan_query = 0;
        if (rc) return rc;
    }
    return launch<A_DENSE>(a, (hipStream_t)stream);
}
}  // namespace

namespace {
// nearest-x2 upsample + 3x3 conv as four 2x2 convs (GemmArgs::sub2): the 192x320 ping-pong tile where the four parity copies of its tile grid fill whole
// rounds of the chip (UNet batch 12: the 32x32 -> 64x64 and 16x16 -> 32x32 up-convs), the 8-wave 128x128 weights-ahead tile otherwise.  Both are the
// instantiations the stride-1 convs of those levels already run (with / without the column statistics for the consuming GroupNorm).
int launch_sub2(GemmArgs a, hipStream_t stream) {
    const char* what = "ae_conv3x3_up2_bf16";
    a.m_fast = (9L * a.N >= 8L * a.M) ? 1 : 0;
    const bool cs = a.colstats != nullptr;
    int rc;
    const long t192 = (a.M % 192 == 0 && a.N % 320 == 0) ? (long)(a.M / 192) * (a.N / 320) * 4 : 0;
    if (t192 > 0 && (double)t192 / (double)(((t192 + 255) / 256) * 256) >= 0.85) {
        const size_t lds = (size_t)(3 * 192 + 2 * 320) * BK * sizeof(bf16_t);
        if (cs) rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, true, 0, 3>, (unsigned)t192, 512, lds, stream, a, what);
        else rc = launch_kernel(gemm_kernel<192, 320, A_CONV3, 2, 4, true, 1, 2, false, 0, 3>, (unsigned)t192, 512, lds, stream, a, what);
    } else {
        const unsigned grid = (unsigned)((long)((a.M + 127) / 128) * ((a.N + 127) / 128) * 4);
        const size_t lds = (size_t)(2 * 128 + 3 * 128) * BK * sizeof(bf16_t);
        if (cs) rc = launch_kernel(gemm_kernel<128, 128, A_CONV3, 4, 2, true, 1, 2, true, 0, 1>, grid, 512, lds, stream, a, what);
        else rc = launch_kernel(gemm_kernel<128, 128, A_CONV3, 4, 2, true, 1, 2, false, 0, 1>, grid, 512, lds, stream, a, what);
    }
    return rc;
}
}  // namespace

extern "C" int ae_conv3x3_up2_bf16(const void* x, const void* w4, const float* bias, void* y, int B, int H, int W, int Cin, int Cout, float* colstats, void* stream) {
    AE_REQUIRE(x && w4 && y, "ae_conv3x3_up2_bf16: null pointer");
    AE_REQUIRE(B > 0 && H > 0 && W > 0, "ae_conv3x3_up2_bf16: bad shape B=%d H=%d W=%d", B, H, W);
    AE_REQUIRE(Cin % 64 == 0 && Cout % 8 == 0, "ae_conv3x3_up2_bf16: Cin %% 64 == 0 (whole K tiles per tap) and Cout %% 8 == 0 (row-contiguous stores), got %d -> %d", Cin, Cout);
    AE_REQUIRE(aligned16(x) && aligned16(w4) && aligned16(y), "ae_conv3x3_up2_bf16: pointers must be 16-byte aligned");
    if (colstats) AE_REQUIRE((H * W) % 32 == 0 && (reinterpret_cast<uintptr_t>(colstats) & 15) == 0, "ae_conv3x3_up2_bf16: column statistics are kept per 32-row slab of a sample and parity: H * W = %d must be a multiple of 32", H * W);
    AE_REQUIRE((long)B * H * W * Cin * 2 < (1L << 31) && (long)Cout * 4 * Cin * 2 < (1L << 31) && (long)B * 4 * H * W < (1L << 31), "ae_conv3x3_up2_bf16: operands must be smaller than 2 GiB");
    GemmArgs a{};
    a.A = (const bf16_t*)x; a.A2 = nullptr; a.W = (const bf16_t*)w4; a.C = y;
    a.bias = bias; a.res = nullptr; a.addvec = nullptr;
    a.M = B * H * W; a.N = Cout; a.K = 4 * Cin; a.Ksplit = a.K;
    a.lda = 0; a.lda2 = 0; a.ldw = 4L * Cin; a.ldc = Cout; a.ldr = Cout; a.ldav = Cout;
    a.epi = EPI_NONE; a.out_f32 = 0; a.rows_per_batch = H * W;
    a.H = H; a.Wd = W; a.Cin = Cin; a.CinPad = Cin; a.Ho = H; a.Wo = W; a.stride = 1; a.ups = 0;
    a.a_bytes = (unsigned)((long)B * H * W * Cin * 2); a.a2_bytes = 0u; a.w_bytes = (unsigned)((long)Cout * 4 * Cin * 2);
    a.splitk = 1; a.partial = nullptr; a.colstats = colstats; a.kmajor = 0; a.sub2 = 1;
    return launch_sub2(a, (hipStream_t)stream);
}

extern "C" long ae_conv3x3_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride, int upsample2x) {
    const int Hv = upsample2x ? 2 * H : H, Wv = upsample2x ? 2 * W : W;
    const int Ho = (Hv + 2 - 3) / stride + 1, Wo = (Wv + 2 - 3) / stride + 1;
    const int CinPad = (Cin + BK - 1) / BK * BK;
    const long M = (long)B * Ho * Wo;
    const int s = make_plan((int)M, Cout, 9 * CinPad, true).splitk;
    return s > 1 ? (long)s * M * Cout : 0;
}

// The split-K plan of ae_conv3x3_bf16 (stride 1, no upsampling) stopped at its fp32 partials: workspace [splitk][B*H*W][Cout] holds the K ranges' raw products
// (no bias); *splitk_out = the number of ranges, 0 when the plan does not split this shape (nothing is launched then: call ae_conv3x3_bf16).
extern "C" int ae_conv3x3_partials_bf16(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, float* workspace, int k_order, int* splitk_out, void* stream) {
    AE_REQUIRE(x && w && workspace && splitk_out, "ae_conv3x3_partials_bf16: null pointer");
    AE_REQUIRE(k_order == 0 || k_order == 1, "ae_conv3x3_partials_bf16: k_order must be 0 (tap, channel) or 1 (64-channel chunk, tap, channel)");
    AE_REQUIRE(k_order == 0 || Cin % 64 == 0, "ae_conv3x3_partials_bf16: the chunk-major K order needs Cin %% 64 == 0 (Cin=%d)", Cin);
    AE_REQUIRE(B > 0 && H > 0 && W > 0 && Cin % 8 == 0 && Cout % 4 == 0, "ae_conv3x3_partials_bf16: bad shape B=%d H=%d W=%d Cin=%d Cout=%d", B, H, W, Cin, Cout);
    AE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(workspace), "ae_conv3x3_partials_bf16: pointers must be 16-byte aligned");
    GemmArgs a{};
    a.A = (const bf16_t*)x; a.W = (const bf16_t*)w; a.C = nullptr;
    const int CinPad = (Cin + BK - 1) / BK * BK;
    a.M = B * H * W; a.N = Cout; a.K = 9 * CinPad; a.Ksplit = a.K;
    a.ldw = 9L * CinPad; a.ldc = Cout; a.ldr = Cout; a.ldav = Cout;
    a.epi = EPI_NONE; a.rows_per_batch = H * W;
    a.H = H; a.Wd = W; a.Cin = Cin; a.CinPad = CinPad; a.Ho = H; a.Wo = W; a.stride = 1; a.ups = 0;
    a.a_bytes = (unsigned)((long)B * H * W * Cin * 2); a.w_bytes = (unsigned)((long)Cout * 9 * CinPad * 2);
    AE_REQUIRE((long)B * H * W * Cin * 2 < (1L << 31) && (long)Cout * 9 * CinPad * 2 < (1L << 31), "ae_conv3x3_partials_bf16: operands must be smaller than 2 GiB");
    a.splitk = make_plan(a.M, a.N, a.K, true).splitk;
    *splitk_out = a.splitk > 1 ? a.splitk : 0;
    if (a.splitk <= 1) return AE_OK;
    a.partial = workspace;
    a.kmajor = k_order;
    a.defer_reduce = 1;
    return launch<A_CONV3>(a, (hipStream_t)stream);
}

extern "C" int ae_conv3x3_bf16(const void* x, const void* w, const float* bias, const float* addvec, long addvec_ld,
                               const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int stride, int upsample2x,
                               int out_f32, float* workspace, float* colstats, int k_order, void* stream) {
    AE_REQUIRE(x && w && y, "ae_conv3x3_bf16: null pointer");
    AE_REQUIRE(k_order == 0 || k_order == 1, "ae_conv3x3_bf16: k_order must be 0 (tap, channel) or 1 (64-channel chunk, tap, channel)");
    AE_REQUIRE(k_order == 0 || (Cin % 64 == 0 && !upsample2x), "ae_conv3x3_bf16: the chunk-major K order needs Cin %% 64 == 0 and no upsampling (Cin=%d)", Cin);
    if (colstats) AE_REQUIRE(!out_f32 && Cout % 8 == 0 && (reinterpret_cast<uintptr_t>(colstats) & 15) == 0, "ae_conv3x3_bf16: column statistics need a bf16 output with Cout %% 8 == 0");
    AE_REQUIRE(B > 0 && H > 0 && W > 0, "ae_conv3x3_bf16: bad shape B=%d H=%d W=%d", B, H, W);
    AE_REQUIRE(Cin % 8 == 0, "ae_conv3x3_bf16: Cin=%d must be a multiple of 8", Cin);
    AE_REQUIRE(Cout % 4 == 0, "ae_conv3x3_bf16: Cout=%d must be a multiple of 4", Cout);
    AE_REQUIRE(stride == 1 || stride == 2, "ae_conv3x3_bf16: stride must be 1 or 2");
    AE_REQUIRE(upsample2x >= 0 && upsample2x <= 2, "ae_conv3x3_bf16: upsample2x must be 0, 1 (nearest) or 2 (zero-insert)");
    AE_REQUIRE(!(upsample2x && stride != 1), "ae_conv3x3_bf16: upsample2x requires stride 1");
    AE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), "ae_conv3x3_bf16: pointers must be 16-byte aligned");
    const int Hv = upsample2x ? 2 * H : H, Wv = upsample2x ? 2 * W : W;
    const int Ho = (Hv + 2 - 3) / stride + 1, Wo = (Wv + 2 - 3) / stride + 1;
    GemmArgs a{};
    a.A = (const bf16_t*)x; a.A2 = nullptr; a.W = (const bf16_t*)w; a.C = y;
    a.bias = bias; a.res = (const bf16_t*)residual; a.addvec = addvec;
    const int CinPad = (Cin + BK - 1) / BK * BK;  // weights are packed [Cout, 9*CinPad], zero padded per tap
    a.M = B * Ho * Wo; a.N = Cout; a.K = 9 * CinPad; a.Ksplit = a.K;
    a.lda = 0; a.lda2 = 0; a.ldw = 9L * CinPad; a.ldc = Cout; a.ldr = Cout; a.ldav = addvec_ld > 0 ? addvec_ld : Cout;
    a.epi = EPI_NONE; a.out_f32 = out_f32; a.rows_per_batch = Ho * Wo;
    a.H = H; a.Wd = W; a.Cin = Cin; a.CinPad = CinPad; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.ups = upsample2x;
    a.a_bytes = (unsigned)((long)B * H * W * Cin * 2); a.a2_bytes = 0u; a.w_bytes = (unsigned)((long)Cout * 9 * CinPad * 2);
    AE_REQUIRE((long)B * H * W * Cin * 2 < (1L << 31) && (long)Cout * 9 * CinPad * 2 < (1L << 31), "ae_conv3x3_bf16: operands must be smaller than 2 GiB");
    a.splitk = workspace ? make_plan(a.M, a.N, a.K, true).splitk : 1;  // without a workspace the kernel runs unsplit
    a.partial = workspace;
    a.colstats = colstats;
    a.kmajor = k_order;
    return launch<A_CONV3>(a, (hipStream_t)stream);
}
