/* libanyedit_hip.so — C ABI of the MI355X (gfx950) AnyEdit denoising hot path.
 *
 * The reference (DCDmllm/AnyEdit) has NO C ABI on this path: its boundaries are Python callables that it already
 * swaps at exactly these points (SURVEY.md §8b): the attention-class registry ldm/modules/attention.py:247-256,
 * the fused-op call xformers.ops.memory_efficient_attention at attention.py:222-233, the layer factories
 * conv_nd/linear/normalization at ldm/modules/diffusionmodules/util.py:202-238, and — the one real FFI in the tree —
 * pybind11 `_C.ms_deform_attn_forward` (GroundingDINO/.../csrc/vision.cpp:53-56).  Each entry point below names the
 * reference interface it replaces.  INTEGRATION.md shows the reference-side ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every function returns int: 0 = ok, <0 = error (AE_ERR_*); ae_last_error() returns the message (thread local);
 *   - never throws, never allocates caller-visible memory, never synchronises the device;
 *   - all data pointers are DEVICE pointers borrowed for the duration of the call; inputs are not mutated;
 *   - the last argument is the hipStream_t (as void*) the work is enqueued on (graph-capture safe);
 *   - "bf16" buffers are raw bfloat16 bits (uint16); activations are channels-last: [B, H*W, C] / [rows, C].
 */
#ifndef ANYEDIT_HIP_H
#define ANYEDIT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define AE_OK 0
#define AE_ERR_ARG (-1)
#define AE_ERR_LAUNCH (-2)
#define AE_ERR_UNSUPPORTED (-3)

/* epilogue selector of ae_gemm_bf16 */
#define AE_EPI_NONE 0  /* out = acc + bias + addvec (+ residual)                         */
#define AE_EPI_GELU 1  /* out = gelu_erf(acc + bias) (+ residual)   (SAM MLPBlock, common.py:13-27) */
#define AE_EPI_GEGLU 2 /* out[:, j] = a_j * gelu_erf(g_j), W rows interleaved 16 a / 16 g (attention.py:49-58) */
#define AE_EPI_SILU 3  /* out = silu(acc + bias)                    (time_embed, openaimodel.py:526-531) */
#define AE_EPI_RELU 4  /* out = max(acc + bias, 0) (+ residual)     (SAM decoder MLPs, mask_decoder.py:154-176, transformer.py:122) */

int ae_version(void);
const char* ae_last_error(void);
int ae_device_arch(char* buf, int n);                          /* e.g. "gfx950:sramecc+:xnack-" */
int ae_device_info(int* cus, long* hbm_bytes, int* clock_khz);

/* nn.Linear / 1x1 nn.Conv2d on channels-last rows (util.py:202-238 factories; attention.py:154-161 to_q/to_k/to_v/to_out,
 * :49-76 FeedForward/GEGLU, :296-318 proj_in/proj_out; openaimodel.py:233-240 skip 1x1, :526-531 time_embed, :214-219 emb_layers).
 *   C[M,N] = epi(A[M,K] @ W[N,K]^T); A2 != NULL: columns [Ksplit,K) of A come from A2 (skip-concat, openaimodel.py:780).
 *   K % 8 == 0, N % 4 == 0, rows 16-byte aligned.  addvec: fp32 [M/rows_per_batch, N] with row stride addvec_ld (0 = N).
 *   out_f32: C is fp32.
 *   colstats (optional, fp32 [ceil(M/32)][N][2]): per-channel (sum, sum of squares) of the stored bf16 output over each 32-row
 *   slab, produced by the GEMM's own epilogue where the tile plan allows (one extra pass over the L2-resident output otherwise):
 *   hand it to ae_groupnorm_nhwc_bf16 and the GroupNorm that consumes C (util.py:217-219) runs without its statistics pass.  */
int ae_gemm_bf16(const void* A, long lda, const void* A2, long lda2, int Ksplit, const void* W, long ldw, void* C, long ldc,
                 int M, int N, int K, const float* bias, const void* residual, long ldr, const float* addvec, long addvec_ld,
                 int rows_per_batch, int epilogue, int out_f32, float* colstats, void* stream);

/* Row-panel GEMM for the short-K (K = 320) Linear layers of the 64x64 UNet level, with the LayerNorm of BasicTransformerBlock
 * (attention.py:263-265, 271-275: norm1 -> to_q|k|v, norm2 -> to_q, norm3 -> GEGLU projection) optionally fused in front:
 *   C[M,N] = epi( LN(A)[M,K] @ W[N,K]^T + bias (+ residual) ),  ln_gamma == NULL: no LayerNorm.  epilogue: AE_EPI_NONE | AE_EPI_GEGLU.
 * A wave keeps 48 rows of A in registers for the whole launch (normalised there), W streams through an LDS ring by LDS-DMA.
 * ae_ln_gemm_supported() tells whether the kernel covers a shape (K == 320, N % 64 == 0, N <= 2560, M >= 192 * 192 rows: one 192-row block per CU); callers fall back to
 * ae_layernorm_bf16 + ae_gemm_bf16 otherwise.  A, W rows and C, residual rows 16-byte aligned.                                   */
int ae_ln_gemm_supported(int M, int N, int K, int epilogue);
int ae_ln_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                    const void* residual, long ldr, const float* ln_gamma, const float* ln_beta, float ln_eps, int epilogue,
                    float* colstats /* as for ae_gemm_bf16; NULL = none */, void* stream);

/* LayerNorm folded into the projection that consumes it (BasicTransformerBlock, attention.py:263-265, 271-275: norm1 -> to_q|k|v,
 * norm2 -> to_q, norm3 -> GEGLU projection) — at every channel width since round 5: 640 / 1280 on the tiled kernel's fold epilogues, 320 (the
 * shapes ae_ln_gemm_supported() covers) on the row-panel kernel's own fold forms, which replace its LayerNorm prologue (AE_RP_FOLD=0: off):
 *   LN(x) W^T + b = rstd_m (x W'^T - mu_m s) + c,   W' = W diag(gamma),  s[n] = sum_k W'[n][k] (over the bf16 values of W'),  c = W beta + b.
 * The GEMM that PRODUCES x (proj_in :296-318, attn1 / attn2 to_out :159-161) emits, next to its bf16 output, one (sum, sum of squares)
 * pair per row and 64-column slice (rowstats_out: fp32 [M][N / 64][2], statistics of the stored bf16 values); the GEMM that consumes
 * LN(x) multiplies the UN-normalised rows by W' and applies the two per-row scalars and the two per-column vectors in its epilogue
 * (ln_stats = the producer's rowstats_out with ln_parts = its N / 64 slices per row; ln_colsum = s; bias = c; the normalised width is K).
 * No LayerNorm launch, no read of x and write of LN(x) for it.  Exactly one of rowstats_out / ln_stats is non-NULL.  epilogue: AE_EPI_NONE
 * (+ residual) when emitting; AE_EPI_NONE or AE_EPI_GEGLU when consuming.  bf16 output, N % 64 == 0, 16-byte aligned rows.
 * ae_gemm_ln_plan(M, N, K, epilogue, mode) tells whether the kernel ae_gemm_ln_bf16 would run for this shape carries the epilogue
 * (mode 1 emit, 2 consume): 1 yes, 0 no — callers keep ae_layernorm_bf16 + ae_gemm_bf16 otherwise.                                    */
int ae_gemm_ln_plan(int M, int N, int K, int epilogue, int mode);
int ae_gemm_ln_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                    const void* residual, long ldr, int epilogue, float* rowstats_out, const float* ln_stats, int ln_parts,
                    const float* ln_colsum, float ln_eps, void* stream);

/* Fused feed-forward of a BasicTransformerBlock at the 64x64 UNet level (round 6; ldm/modules/attention.py:49-76 FeedForward / GEGLU behind norm3 of
 * BasicTransformerBlock._forward :271-275, `x = self.ff(self.norm3(x)) + x`, and — optionally — SpatialTransformer.proj_out + its residual behind the
 * block, :337-340) — ONE launch for LayerNorm -> GEGLU projection -> exact-erf gate -> ff2 (+ bias, + residual) [-> proj_out (+ bias, + residual3)];
 * the [M, H] gated hidden activation stays in registers (it was 126 MB written and re-read at UNet batch 12), and with W3 so does the block's output:
 *   F[M,C] = ( a .* gelu(g) ) W2^T + b2 (+ residual),   [a | g] = LayerNorm(X; gamma, beta, eps) W1^T + b1,   C = 320;
 *   Y = F (W3 == NULL)   or   Y[M,C] = bf16(F) W3^T + b3 (+ residual3).
 *   X, Y, residual, residual3 bf16 rows (16-byte aligned, strides % 8 == 0; Y aliases neither X nor residual3); W1 bf16 [2H, C] and b1 fp32 [2H] in the
 *   AE_EPI_GEGLU row order (16 'a' rows then their 16 gate rows); W2img bf16 [H / 32][C][32]: ff2's weight as the kernel's LDS images (row order, k
 *   permutation and 16-byte piece rotation as the header of csrc/ff_fused.hip states them; ops.pack_ff2_fused builds it once per weight version); b2 fp32
 *   [C] or NULL; W3 bf16 [C, C] row-major (ldw3) or NULL, b3 fp32 [C] or NULL; colstats: as for ae_gemm_bf16 (statistics of Y for the GroupNorm that
 *   consumes it; with W3 only), NULL = none.
 * ae_ff_fused_supported(M, C, H): 1 where the kernel covers the shape (C == 320, H % 64 == 0, H <= 1280, M >= 192 * 192 rows: one 192-row block per
 * CU), 0 otherwise — callers then run ae_gemm_ln_bf16 / ae_ln_gemm_bf16 (GEGLU) + ae_gemm_bf16 (+ ae_ln_gemm_bf16 for proj_out).  AE_FF_FUSED=0
 * answers 0 everywhere (A/B).                                                                                                                     */
int ae_ff_fused_supported(int M, int C, int H);
int ae_ff_fused_bf16(const void* X, long ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* W1, long ldw1, const float* b1,
                     const void* W2img, const float* b2, const void* residual, long ldr, const void* W3, long ldw3, const float* b3,
                     const void* residual3, long ldr3, float* colstats, void* Y, long ldy, int M, int C, int H, void* stream);

/* Fused cross-attention half of a BasicTransformerBlock at the 64x64 UNet level (round 6; ldm/modules/attention.py:273 `x = self.attn2(self.norm2(x), context=context) + x`:
 * norm2, CrossAttention.forward :163-194 (to_q, softmax(QK^T)V over the text keys, to_out) and the residual; with AnySD's decoupled expert segment, DESIGN.md §6) — ONE launch
 * instead of three (LayerNorm-fold q projection, short-K/V attention, to_out); q, logits, probabilities and the attention output stay in registers:
 *   Y[M,320] = ( softmax(q K^T) V + gate_b softmax(q K_ip^T) V_ip ) Wo^T + bo + X,    q = LayerNorm(X; gamma, beta, eps) Wq^T,   8 heads of 40, logits scaled by `scale`.
 *   X, Y bf16 rows (16-byte aligned, strides % 8 == 0, Y != X); rows_per_sample = tokens per sample (a multiple of 128; M a multiple of it);
 *   Wq_img bf16 [8][48][320], Wo_img bf16 [4][320][112], KV_img bf16 [B][8][ae_xattn_fused_kv_bytes() / 2]: the weights and the sample's K | V (Nk text keys, 64 < Nk <= 80,
 *   T <= 16 expert keys) as the kernel's LDS images (layouts: header of csrc/xattn_fused.hip; ops.pack_xattn_wq / _wo / _kv build them — the K | V images once per edit: they are
 *   step-invariant); gate fp32 [B] or NULL (no expert segment: T = 0); bo fp32 [320] or NULL.
 * ae_xattn_fused_supported(M, C, heads, head_dim, rows_per_sample, Nk, T): 1 where the kernel covers the shape (C == 320, 8 heads of 40, M >= 256 * 128 rows: one
 * 128-row block per CU and round) AND AE_XATTN_FUSED=1 is set, 0 otherwise — callers then run ae_gemm_ln_bf16 + ae_attn_fwd_bf16 + ae_gemm_ln_bf16.  Opt-in: results
 * match the three launches (tests/test_hip_ops.py), the launch is slower than they are at UNet batch 12 (profiles/r06_xattn_fused_notes.txt).                          */
int ae_xattn_fused_supported(int M, int C, int heads, int head_dim, int rows_per_sample, int Nk, int T);
/* shape envelope alone (what ae_xattn_fused_bf16 requires); ae_xattn_fused_supported adds the AE_XATTN_FUSED switch and the one-block-per-CU plan rule */
int ae_xattn_fused_covers(int M, int C, int heads, int head_dim, int rows_per_sample, int Nk, int T);
long ae_xattn_fused_kv_bytes(void);
int ae_xattn_fused_bf16(const void* X, long ldx, const float* ln_gamma, const float* ln_beta, float ln_eps, const void* Wq_img, const void* KV_img, const float* gate,
                        const void* Wo_img, const float* bo, void* Y, long ldy, int M, int rows_per_sample, int Nk, int T, float scale, void* stream);

/* 3x3 convolution, padding 1, as implicit GEMM (ResBlock in/out convs openaimodel.py:200-231, stem :536-542, head :726-730,
 * Downsample stride 2 :157-159, Upsample nearest-x2 + conv :108-118 via upsample2x=1).
 *   x [B,H,W,Cin] bf16 channels-last (Cin % 8 == 0), w [Cout, 9*CinPad] bf16 packed (ky,kx,cin) with CinPad = Cin
 *   rounded up to 64 (zero filled), y [B,Ho,Wo,Cout];
 *   addvec fp32 [B,Cout] (time-embedding add, openaimodel.py:262-272), residual bf16 [B,Ho,Wo,Cout] (skip, :274).
 *   workspace: NULL or ae_conv3x3_workspace_floats(...) fp32 elements (0 = not needed): enables split-K for the small-M,
 *   huge-K layers (8x8 / 16x16 latents) that cannot fill 256 CUs with output tiles alone.
 *   k_order: 0 = w packed (ky,kx,cin) as above; 1 = w packed (cin / 64, ky, kx, cin % 64) — the nine taps of a 64-channel chunk
 *   in consecutive K tiles, so a block re-reads its activation window from L2 (Cin % 64 == 0, no upsampling); same result.   */
long ae_conv3x3_workspace_floats(int B, int H, int W, int Cin, int Cout, int stride, int upsample2x);
/* The split-K plan of ae_conv3x3_bf16 (stride 1, no upsampling) stopped at its fp32 partials: workspace (ae_conv3x3_workspace_floats elements) receives the raw
 * products of the K ranges, [splitk][B*H*W][Cout]; *splitk_out = their number, or 0 when the plan does not cut K for this shape (nothing is launched: call
 * ae_conv3x3_bf16).  The bias, the time-embedding vector and the rounding are the consumer's: ae_groupnorm_splitk_nhwc_bf16 (openaimodel.py:262-272).            */
int ae_conv3x3_partials_bf16(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, float* workspace, int k_order, int* splitk_out, void* stream);
int ae_conv3x3_bf16(const void* x, const void* w, const float* bias, const float* addvec, long addvec_ld, const void* residual,
                    void* y, int B, int H, int W, int Cin, int Cout, int stride, int upsample2x, int out_f32, float* workspace,
                    float* colstats /* as for ae_gemm_bf16 (M = B*Ho*Wo, N = Cout); NULL = none */, int k_order, void* stream);

/* Upsample (openaimodel.py:108-118: F.interpolate(scale_factor=2, mode="nearest") followed by the 3x3 conv) as FOUR 2x2 convolutions on the
 * low-resolution input: output pixel (2y+py, 2x+px) reads input rows {y-1, y} (py = 0) or {y, y+1} (py = 1), columns likewise, and the 3x3
 * taps that land on the same input pixel are summed at pack time — 4/9 of the multiply-adds of ae_conv3x3_bf16(..., upsample2x = 1), the same
 * function up to the rounding of the summed bf16 weights.
 *   x [B,H,W,Cin] bf16 channels-last (Cin % 64 == 0); y [B,2H,2W,Cout] bf16 (Cout % 8 == 0); bias fp32 [Cout] or NULL;
 *   w4 [4][Cout][4*Cin] bf16: set p = 2*py + px, K ordered (tap t = 2*i + j, cin) with
 *       w4[p][co][t*Cin + ci] = sum_{ky in S(py,i)} sum_{kx in S(px,j)} w[co][ci][ky][kx],   S(0,0)={0} S(0,1)={1,2} S(1,0)={0,1} S(1,1)={2}
 *   (summed in fp32 from the fp32 master weights, then rounded to bf16 once);
 *   colstats: as for ae_conv3x3_bf16, for the OUTPUT map ([B*4*H*W/32][Cout][2]; needs H*W % 32 == 0), or NULL.                       */
int ae_conv3x3_up2_bf16(const void* x, const void* w4, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                        float* colstats, void* stream);

/* GroupNorm32 (+SiLU) (util.py:217-219 eps 1e-5; attention.py:88-89 eps 1e-6); input may be the channel-concat [x | x2].
 * workspace: fp32, ae_groupnorm_workspace_floats(B,HW,C,groups) elements.  act: 0 none, 1 SiLU.
 * counters: optional int32[B], ZERO on entry and zero again on exit, not shared with a launch running concurrently on another
 * stream: the last partial-sum block of each sample then runs the statistics fold itself (two launches instead of three, same
 * fixed summation order, bit-identical result).  NULL keeps the stand-alone finalize launch.  The tail is OFF unless AE_GN_TAIL=1:
 * measured 3 ms per UNet step slower on MI355X (every block's device-scope release is an L2 write-back).
 * stat_out: optional fp32 [B][groups][2] (mean, rstd) kept for ae_groupnorm_bwd_nhwc_bf16.                                      */
int ae_groupnorm_rows_per_chunk(int HW, int C);
long ae_groupnorm_workspace_floats(int B, int HW, int C, int groups);
/* GroupNorm(+SiLU) of x = bf16(sum_s partial[s] + bias + addvec[b]) where x is never written: ResBlock's `h = in_conv(h) + emb_out; h = out_norm(h); h = SiLU(h)`
 * (openaimodel.py:262-272) at the 16x16 / 8x8 levels, whose convs cut K.  partial: ae_conv3x3_partials_bf16's [splitk][B*HW][C] fp32; bias [C], addvec [B, >= C]
 * (row stride addvec_ld) fp32 or NULL.  Same arithmetic as ae_conv3x3_bf16's reduce launch followed by ae_groupnorm_nhwc_bf16 (one-launch slab form), bit for bit;
 * one launch and one round trip of the activation less.  ae_groupnorm_splitk_supported: maps up to 256 positions, 2..8 K ranges.                                */
int ae_groupnorm_splitk_supported(int B, int HW, int C, int groups, int splitk);
int ae_groupnorm_splitk_nhwc_bf16(const float* partial, int splitk, const float* bias, const float* addvec, long addvec_ld, const float* gamma, const float* beta,
                                  void* y, int B, int HW, int C, int groups, float eps, int act, void* stream);
 * colstats / colstats2: optional per-channel slab statistics of x / x2 as written by the kernels that PRODUCED them (the `colstats`
 * output of ae_gemm_bf16 / ae_ln_gemm_bf16 / ae_conv3x3_bf16: [B*HW/32][C1][2] and [B*HW/32][C-C1][2]; HW % 32 == 0).  With them the
 * statistics pass over the activation is skipped: one block per (sample, group) folds the slab sums, then the apply launch runs.    */
int ae_groupnorm_nhwc_bf16(const void* x, const void* x2, int C1, const float* gamma, const float* beta, void* y, int B, int HW,
                           int C, int groups, float eps, int act, float* workspace, int* counters, float* stat_out,
                           const float* colstats, const float* colstats2, void* stream);

/* nn.LayerNorm over the last dim (attention.py:263-265 eps 1e-5; SAM image_encoder.py:166-182 / common.py:30-43 eps 1e-6). */
int ae_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* y, int M, int C, float eps, void* stream);

/* Fused attention forward = CrossAttention.forward core (attention.py:171-193) / xformers.ops.memory_efficient_attention
 * (attention.py:233) / SAM Attention.forward core (image_encoder.py:231-238).  q/k/v/out addressed by element strides
 * (batch, head, row); head_dim D in {8,16,32,40,48,64,80,96,128,160}.  rel_h/rel_w: optional fp32 [B*H,Nq,kH] / [B*H,Nq,kW]
 * decomposed relative-position bias (image_encoder.py:325-361), Nk == kH*kW.  key_mask: optional uint8 [B,Nk], 0 = masked
 * (attention.py:183-187).  out_scale: optional fp32 [B]; accumulate != 0: out += out_scale[b] * result (decoupled adapter
 * attention, ip_adapter/attention_processor.py:141-173 — the shape template of AnySD's expert K/V, SURVEY.md A9).
 * k2/v2 (optional, head_dim <= 96 or 160): a second key/value segment with its own softmax fused into the same launch:
 * out = Attn(q,k,v) + scale2[b] * Attn(q,k2,v2).                                                                           */
int ae_attn_fwd_bf16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                     long q_sb, long q_sh, long q_sn, long k_sb, long k_sh, long k_sn, long v_sb, long v_sh, long v_sn,
                     long o_sb, long o_sh, long o_sn, float scale, const float* rel_h, const float* rel_w, int kH, int kW,
                     const unsigned char* key_mask, const float* out_scale, int accumulate, const void* k2, const void* v2,
                     int Nk2, long k2_sb, long k2_sh, long k2_sn, long v2_sb, long v2_sh, long v2_sn, const float* scale2,
                     float* lse, float* lse2, void* stream);
/* fp8 (OCP e4m3) attention forward (BASELINE.json configs[4], SAM image-encoder attention image_encoder.py:224-240, rel-pos bias
 * :325-361): Q, K, V and the probabilities are e4m3 operands of v_mfma_scale_f32_32x32x64_f8f6f4, logits / softmax / accumulation
 * fp32.  Scales: per (batch, head) amax of q, k, v measured on the fly (k: float scale folded into q; q and v: powers of two that
 * ride in the MFMA's E8M0 scale operand / the final normalisation).  head_dim % 8 == 0, <= 88.  rel_h / rel_w (optional): the
 * decomposed bias for key grids with kW == 64 (global attention on 64 x 64 tokens).  `workspace`: device scratch of at least
 * ae_attn_fp8_workspace_bytes(...) bytes, 256-byte aligned (quantised operands live there for the duration of the call).         */
long ae_attn_fp8_workspace_bytes(int B, int H, int Nq, int Nk, int D);
int ae_attn_fwd_fp8(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D, long q_sb, long q_sh,
                    long q_sn, long k_sb, long k_sh, long k_sn, long v_sb, long v_sh, long v_sn, long o_sb, long o_sh, long o_sn,
                    float scale, const float* rel_h, const float* rel_w, int kH, int kW, void* workspace, long workspace_bytes,
                    void* stream);
/* lse / lse2 (optional, fp32 [B,H,Nq]): log2-domain log-sum-exp of the first / second segment's softmax, kept for
 * ae_attn_bwd_bf16.
 *
 * ---- training step (SURVEY.md row A11: train.py:625-710; the in-tree restatement is LatentDiffusion.p_losses,
 * ddpm.py:889-932).  The UNet is frozen: layers are differentiated w.r.t. activations only; Linear / conv data-gradients reuse
 * ae_gemm_bf16 / ae_conv3x3_bf16 with transposed / rotated packed weights (upsample2x = 2: zero-insert gather = adjoint of the
 * stride-2 Downsample conv, openaimodel.py:157-159).
 *
 * Attention backward for one key/value segment (autograd of attention.py:171-193): delta [B,H,Nq] fp32 is an OUTPUT
 * (rowsum(P o dP) with the un-scaled dout; its sum over heads and rows is d/d out_scale[b]).  dk, dv may both be NULL.
 * out (optional, layout of dout): the forward output of THIS segment alone (no second segment, no out_scale folded in): delta is
 * then rowsum(dout o out), computed up front, and the dQ pass keeps one accumulator set instead of two.
 * workspace (optional): NULL or ae_attn_bwd_workspace_floats(...) fp32 elements, 16-byte aligned (0 = not needed): lets the dK / dV
 * pass of a FEW-key segment (cross-attention: 78 text or 16 adapter tokens against 4096 queries is only B H blocks) cut its query
 * tiles across blocks — fp32 partials, summed in split order by a second launch (deterministic).  Without it every shape runs
 * the one-block-per-128-keys pass.                                                                                             */
long ae_attn_bwd_workspace_floats(int B, int H, int Nq, int Nk, int D);
int ae_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dout, const void* out, const float* lse, float* delta, void* dq,
                     void* dk, void* dv, int B, int H, int Nq, int Nk, int D, long q_sb, long q_sh, long q_sn, long k_sb,
                     long k_sh, long k_sn, long v_sb, long v_sh, long v_sn, long o_sb, long o_sh, long o_sn, long dq_sb,
                     long dq_sh, long dq_sn, long dk_sb, long dk_sh, long dk_sn, long dv_sb, long dv_sh, long dv_sn,
                     float scale, const float* out_scale, int accumulate_dq, float* workspace, void* stream);
/* GroupNorm(+SiLU) backward w.r.t. the input(s) (autograd of util.py:217-219 + nn.SiLU); dx2 receives channels [C1, C).
 * counters: as for ae_groupnorm_nhwc_bf16.  stat_in: optional (mean, rstd) saved by the forward launch — the statistics pass over
 * x is skipped (three launches instead of five) and the gradient uses exactly the statistics the forward normalised with.      */
long ae_groupnorm_bwd_workspace_floats(int B, int HW, int C, int groups);
int ae_groupnorm_bwd_nhwc_bf16(const void* x, const void* x2, int C1, const float* gamma, const float* beta, const void* dy,
                               void* dx, void* dx2, int B, int HW, int C, int groups, float eps, int act, float* workspace,
                               int* counters, const float* stat_in, int accumulate, void* stream);
/* accumulate (GroupNorm: bit 0 -> dx, bit 1 -> dx2; LayerNorm: 0 / 1): the gradient is ADDED to what the output tensor already
 * holds (fp32 add, one rounding) — a tensor with several consumers collects its gradient without a separate ae_add_bf16 launch.
 * LayerNorm backward w.r.t. the input; row_stat (optional fp32 [M,2]) receives (mean, rstd) for the parameter gradients.    */
int ae_layernorm_bwd_bf16(const void* x, const float* gamma, const void* dy, void* dx, float* row_stat, int M, int C, float eps,
                          int accumulate, void* stream);
int ae_layernorm_param_grad_f32(const void* x, const void* dy, const float* row_stat, float* dgamma, float* dbeta, int M, int C,
                                void* stream);
/* y = a + b (gradient accumulation where a layer's input fans out).                                                          */
int ae_add_bf16(const void* a, const void* b, void* y, long n, void* stream);
/* y = a + alpha * b, bf16 (ControlNet residual injection: hs.pop() + scale * control.pop(), cldm.py:40-41, 336-338).          */
int ae_axpy_bf16(const void* a, const void* b, float alpha, void* y, long n, void* stream);
/* GEGLU un-fused for training (attention.py:49-57): h = [a | g] [M, 2F] -> y = a * gelu(g); backward -> dh.                   */
int ae_geglu_fwd_bf16(const void* h, void* y, long M, int F, void* stream);
int ae_geglu_bwd_bf16(const void* h, const void* dy, void* dh, long M, int F, void* stream);
/* adjoint of the nearest-x2 upsample (openaimodel.py:108-118): x [B,2H,2W,C] -> y [B,H,W,C] = 2x2 block sums.                  */
int ae_sumpool2x2_bf16(const void* x, void* y, int B, int H, int W, int C, void* stream);
/* out[n] = sum_m x[m, n] (bias gradient of the small trainable Linear layers).                                               */
int ae_colsum_bf16_f32(const void* x, float* out, int M, int N, long ld, void* stream);
/* d/dpred mean((pred - target)^2) * loss_scale (train.py:696).                                                               */
int ae_mse_grad_f32(const float* pred, const float* target, float* out, long n, float loss_scale, void* stream);
/* torch.optim.AdamW step on fp32 state (train.py:536-541); step counts from 1; grad is multiplied by grad_scale first.       */
int ae_adamw_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                 float eps, float weight_decay, int step, float grad_scale, void* stream);
/* out[r] = sum_j x[r, j] (fp32, fixed order): gate gradient = sum over heads and rows of the adapter segment's delta.     */
int ae_rowsum_f32(const float* x, float* out, int rows, long n, void* stream);
/* dst[code[b]] += src[b] in batch order (task-embedding gradient rows).                                                      */
int ae_scatter_add_rows_f32(const float* src, const int* code, float* dst, int B, int D, int n_rows, void* stream);
/* Backward of the task-router gate value w.r.t. the task embedding (ae_task_gate).                                           */
int ae_task_gate_bwd(const float* probs, const int* top1, const float* dgate, const float* Wg, int B, int Dt, int E, float* dte,
                     void* stream);
/* Gradient of the same gate value w.r.t. the router itself: dWg[e,:] = sum_b dlogit[b,e] task_embs[edit_code[b]],
 * dbg[e] = sum_b dlogit[b,e] (dbg may be NULL); samples summed in index order.  The router is a trainable of the adapter
 * group in our AnySD spec (DESIGN.md §6; train.py:483-485 optimises the adapter modules it lives in).                          */
int ae_task_gate_wgrad(const float* probs, const int* top1, const float* dgate, const float* task_emb, const long* edit_code, int B,
                       int n_tasks, int Dt, int E, float* dWg, float* dbg, void* stream);

/* AnySD per-expert adapter K/V projection of the training step, grouped over the samples of a batch (our spec, DESIGN.md §6; the
 * reference trains `adapter_modules` at train.py:410-424, 536-541).  x: [B*T, Dc] bf16 image-prompt rows (T <= 8 per sample),
 * W: [E, N, Dc] fp32 masters (rounded to bf16 in registers), experts: [B] int32 routed expert per sample.
 *   fwd    y[b*T+t, n]  = sum_k x[b*T+t, k] * W[experts[b], n, k]                              (bf16 out)
 *   dgrad  dx[b*T+t, k] = sum_n dy[b*T+t, n] * W[experts[b], n, k]                             (bf16 out, fixed summation order;
 *          partial: ae_expert_kv_dgrad_slices(N)*B*T*Dc floats of scratch)
 *   wgrad  dW[e, n, k]  = sum_{b: experts[b] == e} sum_t dy[b*T+t, n] * x[b*T+t, k]            (fp32, every expert written)      */
int ae_expert_kv_fwd(const void* x, const float* W, const int* experts, void* y, int B, int T, int N, int Dc, int E, void* stream);
int ae_expert_kv_dgrad_slices(int N);
int ae_expert_kv_dgrad(const void* dy, const float* W, const int* experts, void* dx, int B, int T, int N, int Dc, int E, float* partial,
                       void* stream);
int ae_expert_kv_wgrad(const void* dy, const void* x, const int* experts, float* dW, int B, int T, int N, int Dc, int E, void* stream);

/* out[b,y,x] = in[b,x,y], inner dim zero-padded to Xpad: NCHW <-> channels-last at the UNet boundary
 * ('b c h w -> b (h w) c', attention.py:329,337).                                                                           */
int ae_transpose_last2(const void* in, void* out, int B, int X, int Y, int Xpad, int in_bf16, int out_bf16, void* stream);
/* th.cat([h, hs.pop()], dim=1) (openaimodel.py:780) on channels-last rows.                                                   */
int ae_concat_channels_bf16(const void* a, int Ca, const void* b, int Cb, void* y, long rows, void* stream);
/* im2col of the UNet's stem conv (3x3, pad 1, stride 1 over the 8-channel input, openaimodel.py:536-542): x [B,H,W,8] bf16 -> y [B*H*W, 128] bf16,
 * column 8 tap + c (tap = 3 ky + kx), zeros outside the image and in columns 72..127; the conv is then ae_gemm_bf16 with the weight packed
 * [Cout, 128] in the same column order — K = 128 instead of nine K tiles that are 7/8 zero padding.                                            */
int ae_im2col3x3_c8_bf16(const void* x, void* y, int B, int H, int W, void* stream);
/* Adjoint of the channel concat: a (+)= y[:, :Ca], b (+)= y[:, Ca:] in one pass; a or b may be NULL (that half is skipped).       */
int ae_split_channels_bf16(const void* y, int Ca, int Cb, void* a, void* b, long rows, int accumulate_a, int accumulate_b, void* stream);
/* timestep_embedding (util.py:154-174): [cos | sin], t given as int64 or fp32.                                               */
int ae_timestep_embedding(const long* t_i64, const float* t_f32, void* out_bf16, float* out_f32, int B, int dim, float max_period,
                          void* stream);
/* CFG combine + DDIM update, fp32, unfused arithmetic (ddim.py:211-212, 228-250; 3-branch global_tool.py:172-177).
 * eps holds `branches` stacked copies of n elements: 2 -> [uncond, cond]; 3 -> [text, image, uncond].                        */
int ae_ddim_step_f32(const float* x, const float* eps, const float* noise, float* x_prev, float* pred_x0, float* e_out, long n,
                     int branches, float s0, float s1, float sqrt_one_minus_at, float sqrt_at, float sqrt_a_prev, float dir_coef,
                     float sigma_t, float temperature, void* stream);
/* PLMS multistep combination of eps predictions (PLMSSampler.p_sample_plms, plms.py:226-240); order 0 = (e + old1) / 2.           */
int ae_plms_combine_f32(const float* e_t, const float* old1, const float* old2, const float* old3, float* out, long n, int order,
                        void* stream);
/* DDIM inversion update (DDIMSampler.encode, ddim.py:253-298): x_next = cx*x + ce*e, e = CFG combination of `branches`
 * stacked predictions ([uncond, cond]); cx, ce from the host in float64 as the reference computes them.                       */
int ae_ddim_encode_step_f32(const float* x, const float* eps, float* x_next, long n, int branches, float scale, float cx, float ce,
                            void* stream);
/* q_sample (ddpm.py:356-359) fused with the mask blend (ddim.py:154-157; ip2p_order=1: global_tool.py:183-184).               */
int ae_mask_blend_f32(const float* img, const float* x0, const float* noise, const float* mask, float* out, int B, int C, int HW,
                      float sqrt_ac, float sqrt_one_minus_ac, int ip2p_order, void* stream);
int ae_q_sample_f32(const float* x0, const float* noise, const float* sqrt_ac, const float* sqrt_one_minus_ac, float* out, int B,
                    long per_sample, void* stream);
/* y = bf16(silu(x)): the nn.SiLU in front of ResBlock.emb_layers (openaimodel.py:212-219).                                    */
int ae_silu_to_bf16(const void* x, int in_bf16, void* y, long n, void* stream);
/* y = x + p broadcast with period (pos_embed add, image_encoder.py:108-109).                                                 */
int ae_add_bcast_bf16(const void* x, const void* p, void* y, long n, long period, void* stream);
/* Resampling without a convolution on channels-last rows: mode 0 = nearest x2 (Upsample(use_conv=False), openaimodel.py:108-118), mode 1 = 2x2 mean
 * (Downsample(use_conv=False) = avg_pool_nd, openaimodel.py:154-155; floor on odd sizes) — the h_upd / x_upd of ResBlock(up= / down=), :215-221, 254-260.
 * x [B, H, W, C] bf16 -> y [B, 2H, 2W, C] / [B, H/2, W/2, C]; C % 8 == 0.                                                                              */
int ae_resample2x_rows_bf16(const void* x, void* y, int B, int H, int W, int C, int mode, void* stream);
/* ResBlock(use_scale_shift_norm=True), openaimodel.py:264-268: y = act(x * (1 + scale[b]) + shift[b]) on the GroupNorm's output rows x [B * HW, C] bf16;
 * emb fp32 [B, >= 2C] with row stride ld_emb holds scale | shift (th.chunk(emb_out, 2, dim=1)); silu != 0 applies out_rest's SiLU.                     */
int ae_scale_shift_rows_bf16(const void* x, const float* emb, long ld_emb, void* y, int B, long HW, int C, int silu, void* stream);
/* window_partition / window_unpartition (image_encoder.py:243-289); arguments are always (image, windows).                    */
int ae_window_partition_bf16(const void* image, void* windows, int B, int H, int W, int C, int ws, int reverse, void* stream);
/* LayerNorm with the window partition of SAM's windowed blocks folded into its row addressing (image_encoder.py:166-182, 243-289).
 * mode 1: y[window rows] = window_partition(LayerNorm(x[image rows])), padding rows zero (norm1 + partition);
 * mode 2: xsum[image rows] = bf16(x[window rows] + shortcut[image rows]), y = LayerNorm(xsum) (un-partition + shortcut + norm2).
 * Same arithmetic as ae_layernorm_bf16 / ae_window_partition_bf16 / ae_add_bcast_bf16 in sequence (xsum bit for bit).             */
int ae_layernorm_window_supported(int C);
int ae_layernorm_window_bf16(const void* x, const void* shortcut, const float* gamma, const float* beta, void* y, void* xsum,
                             int B, int H, int W, int C, int ws, int mode, float eps, void* stream);
/* rel_h / rel_w einsums of add_decomposed_rel_pos (image_encoder.py:349-355); Rh [qH,kH,D], Rw [qW,kW,D] fp32.                */
int ae_sam_relpos_terms(const void* q, long q_sb, long q_sh, long q_sn, const float* Rh, const float* Rw, float* rel_h,
                        float* rel_w, int B, int heads, int qH, int qW, int kH, int kW, int D, void* stream);
/* PatchEmbed im2col (image_encoder.py:364-395): [B,Cin,H,W] fp32 -> [B*(H/P)*(W/P), Cin*P*P] bf16.                            */
int ae_patchify_f32_bf16(const float* x, void* y, int B, int Cin, int H, int W, int P, void* stream);
/* mean((a-b)^2) -> out[0] (train.py:696, ddpm.py:367-380).                                                                    */
int ae_mse_f32(const float* a, const float* b, float* out, long n, void* stream);
/* AnySD task router (OUR spec, reference source absent — SURVEY.md §8a row A9): logits = task_emb[edit_code] @ Wg^T + bg,
 * softmax over E experts, top-1 index + probability per sample.                                                              */
int ae_task_gate(const float* task_emb, const long* edit_code, const float* Wg, const float* bg, int B, int n_tasks, int Dt,
                 int E, float* probs, int* top1, float* top1_prob, void* stream);

/* ---- first-stage autoencoder (SURVEY.md §8f N1; the step on either side of the denoising loop) — reuses ae_conv3x3_bf16,
 * ae_gemm_bf16, ae_groupnorm_nhwc_bf16; two small kernels of its own:
 * P = softmax(scale * S) row-wise, fp32 logits -> bf16 (AttnBlock, diffusionmodules/model.py:188-192).                        */
int ae_softmax_rows_f32_bf16(const float* S, long lds, void* P, long ldp, int rows, int cols, float scale, void* stream);
/* DiagonalGaussianDistribution (distributions.py:25-37): moments [B, 2*per_half] -> z = mean + std * noise (noise NULL: mode),
 * optional mean / clamped logvar / std outputs.                                                                              */
int ae_gaussian_moments_f32(const float* moments, const float* noise, float* z, float* mean, float* logvar, float* std_out, int B,
                            long per_half, void* stream);

/* ---- GroundingDINO multi-scale deformable attention forward (SURVEY.md §8f N2): replaces the reference's only native op,
 * `_C.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)`
 * (csrc/vision.cpp:53-56, csrc/MsDeformAttn/ms_deform_attn.h:22-41, ms_deform_im2col_cuda.cuh:237-299).
 * value [bs, S, heads, d] fp32 (S = sum_l H_l*W_l), spatial_shapes [L,2] / level_start_index [L] int64, sampling_loc
 * [bs, Q, heads, L, P, 2] in [0,1] (x, y), attn_weight [bs, Q, heads, L, P]; out [bs, Q, heads*d] fp32.                      */
int ae_ms_deform_attn_fwd_f32(const float* value, const long* spatial_shapes, const long* level_start_index, const float* sampling_loc,
                              const float* attn_weight, float* out, int bs, int S, int heads, int d, int Q, int L, int P, void* stream);
/* Backward of the same operator: `_C.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
 * grad_output, im2col_step)` (csrc/vision.cpp:57, ms_deform_attn.h:43-64; called from MultiScaleDeformableAttnFunction.backward,
 * ms_deform_attn.py:68-90).  grad_out [bs, Q, heads*d] -> grad_value [bs, S, heads, d], grad_sampling_loc [bs, Q, heads, L, P, 2],
 * grad_attn_weight [bs, Q, heads, L, P]; all three fully overwritten.  grad_value accumulates with float atomics.              */
int ae_ms_deform_attn_bwd_f32(const float* value, const long* spatial_shapes, const long* level_start_index, const float* sampling_loc,
                              const float* attn_weight, const float* grad_out, float* grad_value, float* grad_sampling_loc,
                              float* grad_attn_weight, int bs, int S, int heads, int d, int Q, int L, int P, void* stream);
/* nn.Linear in EXACT fp32 (v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate) for the four projections of GroundingDINO's
 * MultiScaleDeformableAttention (ms_deform_attn.py:281-288, 330-352: value_proj, sampling_offsets, attention_weights, output_proj),
 * which run in fp32 in the reference and feed bilinear sampling locations.  C[M,N] = A[M,K] W[N,K]^T + bias; K % 16 == 0.       */
int ae_linear_f32(const float* A, long lda, const float* W, long ldw, const float* bias, float* C, long ldc, int M, int N, int K,
                  void* stream);

/* DPM-Solver / DPM-Solver++ multistep step (ldm/models/diffusion/dpm_solver/dpm_solver.py:246-316 guidance, :352-365 data prediction,
 * :469-513 first-order and :723-777 second-order multistep updates): m = predict_x0 ? (x - sigma_s e)/alpha_s : e with e the guided
 * noise of model_out ([uncond, cond] when branches == 2; v_param: alpha_s*out + sigma_s*x); update != 0: x_next = a x - b m
 * - c inv_r0 (m - m_prev) (m_prev NULL: first order).  a, b, c, inv_r0: the reference's per-step scalars, computed by the caller. */
int ae_dpm_multistep_f32(const float* x, const float* model_out, const float* m_prev, float* m_cur, float* x_next, long n, int branches,
                         float scale, int v_param, int predict_x0, float sigma_s, float alpha_s, int update, float a, float b, float c,
                         float inv_r0, void* stream);

/* The DPM-Solver updates outside the multistep-2 fast path (dpm_solver.py:469-722 singlestep first / second / third update, :780-826
 * multistep third update): each is out = c0 x0 + c1 x1 + c2 x2 + c3 x3 with host-side fp32 scalars (x1..x3 may be NULL).              */
int ae_lincomb4_f32(float* out, const float* x0, float c0, const float* x1, float c1, const float* x2, float c2, const float* x3, float c3,
                    long n, void* stream);
/* Error estimate of the adaptive step-size solver (dpm_solver.py:925-927): out[b] = sqrt(mean(((x_higher - x_lower) / delta)^2)) over the
 * n_per_sample elements of sample b, delta = max(atol, rtol * max(|x_lower|, |x_prev|)).                                            */
int ae_dpm_adaptive_err_f32(const float* x_lower, const float* x_higher, const float* x_prev, float atol, float rtol, int B,
                            long n_per_sample, float* out, void* stream);

/* ---- SAM prompt encoder / mask decoder (SURVEY.md §8f N3): the non-GEMM kernels behind SamPredictor.predict_torch
 * (segment_anything/predictor.py:168-245).
 * ae_layernorm_act_bf16: LayerNorm over a narrow last dim (C <= 512) with optional fused GELU (act 1) — the LayerNorm2d + GELU of
 *   MaskDecoder.output_upscaling (mask_decoder.py:53-60) on channels-last rows.
 * ae_sam_pe_encode_f32: PositionEmbeddingRandom (prompt_encoder.py:174-214) of N pixel coordinates (x, y): out [N, 2F] =
 *   cat(sin, cos)(2 pi ((2c-1) @ gauss[2,F])), c = (coords + offset) * (inv_w, inv_h).  labels (int32, optional): -1 -> encoding
 *   zeroed + table row 0 (not_a_point_embed); l >= 0 -> + table row 1+l (point_embeddings[l]) — PromptEncoder._embed_points /
 *   _embed_boxes (:73-102).  table [5, 2F].
 * ae_sam_mask_downscale_bf16: PromptEncoder.mask_downscaling[0:6] (:46-53) for mask_in_chans = 16: masks [B,1,4h,4w] fp32 ->
 *   channels-last rows [B*h*w, 16] bf16 (the closing 1x1 convolution is an ae_gemm_bf16).
 * ae_sam_mask_product_f32: masks = hyper_in @ upscaled_embedding (mask_decoder.py:141-149).  up: bf16 [B, h, w, 2,2, 2,2, C] = the
 *   two transposed convolutions' GEMM outputs left un-shuffled; hyper [B, M, C] fp32; out [B, M, 4h, 4w] fp32.
 * ae_sam_postprocess_masks: Sam.postprocess_masks (sam.py:133-162) fused: bilinear (Hl,Wl)->(S,S), crop [:ih,:iw], bilinear ->
 *   (oh,ow); writes fp32 logits and/or the (logit > threshold) uint8 mask.  merge != 0: out_u8 [oh,ow] = OR over the N masks
 *   (maskgeneration's mask_mode 'merge', tools/tool.py:239-241).
 * ae_nms_sorted_f32: greedy IoU suppression == torchvision.ops.nms (tools/tool.py:224) over XYXY boxes already sorted by
 *   descending score; keep[i] (uint8) = 1 for survivors.
 * ae_sam_preprocess_f32: Sam.preprocess (sam.py:164-174): (x - mean[c]) / std[c], zero-padded to [B, C, S, S]; x uint8 or fp32. */
int ae_layernorm_act_bf16(const void* x, const float* gamma, const float* beta, void* y, long M, int C, float eps, int act, void* stream);
int ae_sam_pe_encode_f32(const float* coords, const int* labels, const float* gauss, const float* table, float* out, int N, int F,
                         float offset, float inv_w, float inv_h, void* stream);
int ae_sam_mask_downscale_bf16(const float* masks, const float* w1, const float* b1, const float* g1, const float* e1, const float* w2,
                               const float* b2, const float* g2, const float* e2, void* out, int B, int h, int w, float eps, void* stream);
int ae_sam_mask_product_f32(const void* up, const float* hyper, float* out, int B, int h, int w, int M, int C, void* stream);
int ae_sam_postprocess_masks(const float* low, float* out_f32, void* out_u8, int N, int Hl, int Wl, int S, int ih, int iw, int oh, int ow,
                             float threshold, int merge, void* stream);
int ae_nms_sorted_f32(const float* boxes, void* keep, int N, float iou_threshold, void* stream);
int ae_sam_preprocess_f32(const void* x, int x_is_u8, float* y, int B, int C, int h, int w, int S, const float* mean, const float* stdv,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ANYEDIT_HIP_H */
