// Attention backward for gfx950 (training step, SURVEY.md §8a row A11: train.py:694-703 back-propagates the eps-MSE through the
// frozen UNet's self- / cross- / adapter attention, attention.py:163-194, to reach the trainable adapter projections).
//
//   S = Q K^T,  P = softmax(scale S),  O = P V      (forward, attention.hip; it also stores L2[q] = log2 sum_k 2^(c2 S[q,k]))
//   dV = P^T dO      dP = dO V^T      dS = P o (dP - delta) * scale,  delta[q] = sum_k P[q,k] dP[q,k]      dQ = dS K      dK = dS^T Q
//
// Two deterministic passes over one kernel template (no float atomics):
//   MODE_DQ : a block owns 128 query rows (operands Q, dO live in registers), streams 64-key tiles of K / V through LDS and
//             accumulates  T1 = sum_k (P o dP) K,  T2 = sum_k P K,  delta = rowsum(P o dP);  dQ = g scale (T1 - delta T2).
//             delta is written out for the second pass (and for the gate gradient of the adapter segment).
//   MODE_DKV: a block owns 128 key rows (K, V in registers), streams 64-query tiles of Q / dO (+ L2, delta) and accumulates
//             dV = g sum_q P^T dO,  dK = g scale sum_q dS^T Q.
// `g` = optional per-batch output scale (the adapter gate: out = Attn(q,K,V) + g_b Attn(q,K_ip,V_ip)); delta is kept UN-scaled.
// Same MFMA conventions as the forward: the streamed side is the row (A) operand read from LDS, the fixed side the column (B)
// operand held in registers, so every lane owns one fixed-side row; the second product takes its A operand — the transposed
// streamed tile, contraction slots in accumulator order — by ds_read_b64_tr_b16 from the same row-major image and its B operand
// straight from the accumulator registers.  The streamed tiles are double-buffered (next tile's global loads in registers under the
// MFMAs, one barrier per tile); with the segment's own forward output at hand, delta = rowsum(dO o O) comes from a small pre-pass
// and MODE_DQ keeps one accumulator set (PRE).
#include "common.hpp"
#include <stdlib.h>

namespace {

constexpr int ST = 64;  // streamed rows per tile
// Head dim padded to whole K = 32 steps (zero columns: a 16x16x16 step costs the matrix pipe the same 16 cycles as a 16x16x32 one, and the 8-byte tail reads
// could not share a conflict-free row stride with the 16-byte ones), and the LDS row stride of the streamed images: ds_read_b128 is served in the lane groups
// {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS) — lane (l15, lg) reads 16 bytes at row l15, column block lg, so a group holds rows {0-3, 12-15} at
// block 0 and rows {4-11} at block 1: conflict-free iff (stride / 16 B) = 2 (mod 4), i.e. stride = 32 k + 16 elements; the same strides keep the transposing
// ds_read_b64_tr_b16 reads of the second product conflict-free.  (Round 5: the former stride DQK + 8 made 7 of 8 lanes of every group collide pairwise:
// SQ_LDS_BANK_CONFLICT = 22 % of the LDS cycles of the dK / dV pass.)
constexpr int bwd_dqk(int D) { return (D + 31) / 32 * 32; }
constexpr int bwd_row(int D) { return (bwd_dqk(D) + 15) / 32 * 32 + 16; }
constexpr float LOG2E = 1.4426950408889634f;
enum { MODE_DQ = 0, MODE_DKV = 1 };

struct BwdArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dO;
    const float* lse;        // [B, H, Nq] log2-domain log-sum-exp written by the forward
    float* delta;            // [B, H, Nq] rowsum(P o dP): written by MODE_DQ, read by MODE_DKV
    const float* out_scale;  // optional [B]
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int B, H, Nq, Nk;
    long q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, o_sb, o_sh, o_sn;
    long dq_sb, dq_sh, dq_sn, dk_sb, dk_sh, dk_sn, dv_sb, dv_sh, dv_sn;
    float scale;
    int accum_dq;  // dq += (second segment sharing the same queries)
    // MODE_DKV with few key rows (cross-attention: 78 text / 16 adapter tokens against 4096 queries is B H blocks of 64 query tiles each): the query
    // tiles are cut `nsplit` ways, every block writes its fp32 partial (dK, dV) to `part` [nsplit][B H][Nk][2][DVP] and attn_bwd_reduce_kernel sums
    // them in split order (fixed order: deterministic) and rounds once.  nsplit <= 1: the block owns every query tile and writes bf16 itself.
    float* part;
    int nsplit;
    int abl;   // lab build only (-DAE_BWD_LAB, env AE_BWD_ABL): bit 0 skip the second product, 1 skip the exp2 / P block, 2 skip the first product, 3 no barrier / restage, 4 no global loads
};

typedef __attribute__((ext_vector_type(4))) short s16x4v;
// ds_read_b64_tr_b16: inside a 16-lane group lane i addresses row (i >> 2), columns 4 (i & 3) .. +3 of a [4 rows][16 columns] block of
// 16-bit elements and receives COLUMN i (its 4 rows, ascending) — a transposing fragment read straight from a row-major image.
__device__ __forceinline__ s16x4v lds_tr16(const bf16_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)p);
#else
    return s16x4v{};
#endif
}
// gfx950 has no interlock for a VALU write into the SrcA / SrcB registers of an MFMA that is issued but still queued for the matrix pipe
// (profiles/r03_attn_qg2_hazard.txt; tools/isa_audit.py).  The pipe is in order: an instruction that READS the result of the phase's LAST MFMA is issued only
// once every MFMA of the phase has finished — placed at a phase boundary (with a scheduling barrier behind it) it keeps the VALU block that follows, and its
// re-use of the operand registers, behind the whole MFMA phase.
__device__ __forceinline__ void mfma_fence_begin() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ void mfma_fence_read(const f32x4& r) {   // one of the results of the phase's last MFMA group (hipcc orders the MFMAs of a group freely: read them all)
#if defined(__HIP_DEVICE_COMPILE__)
    float t;
    const float l0 = r[0];
    asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(l0));
#endif
}
__device__ __forceinline__ void mfma_fence_end() { __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ float lab_f(bf16x8_t v) { union { bf16x8_t b; float f[4]; } u; u.b = v; return u.f[0]; }   // lab ablations: keeps an operand alive
__device__ __forceinline__ bf16x8_t cat_tr(s16x4v lo, s16x4v hi) {
    union { struct { s16x4v a, b; } s; bf16x8_t v; } u;
    u.s.a = lo; u.s.b = hi;
    return u.v;
}

// delta[b,h,q] = sum_d dO[b,q,h,d] O[b,q,h,d] (= rowsum(P o dP) of a single-segment attention whose output is O): one thread per
// (b, q, h), adjacent threads read adjacent head slices of a token row.
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* dO, const bf16_t* O, float* delta, int B, int H, int Nq, long o_sb, long o_sh,
                                                         long o_sn) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * Nq * H) return;
    const int h = (int)(i % H);
    const long bq = i / H;
    const int q = (int)(bq % Nq), b = (int)(bq / Nq);
    const long off = (long)b * o_sb + (long)q * o_sn + (long)h * o_sh;
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < D / 8; ++c) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(dO + off + c * 8), o = *reinterpret_cast<const u32x4*>(O + off + c * 8);
        acc += bf16lo(a.x) * bf16lo(o.x) + bf16hi(a.x) * bf16hi(o.x) + bf16lo(a.y) * bf16lo(o.y) + bf16hi(a.y) * bf16hi(o.y) +
               bf16lo(a.z) * bf16lo(o.z) + bf16hi(a.z) * bf16hi(o.z) + bf16lo(a.w) * bf16lo(o.w) + bf16hi(a.w) * bf16hi(o.w);
    }
    delta[((long)b * H + h) * Nq + q] = acc;
}

// PRE (MODE_DQ only): delta is already in p.delta (attn_delta_kernel), so dS = P o (dP - delta) is formed directly and the pass keeps
// ONE accumulator set (dQ = g scale sum_k dS K) instead of two (T1, T2) — a quarter of the pass's MFMAs and 40 registers less.
template <int D, int QF, int MODE, bool PRE = false>
__global__ __launch_bounds__(256, (D <= 48 ? 2 : 1)) void attn_bwd_kernel(const BwdArgs p) {
    static_assert(!PRE || MODE == MODE_DQ, "PRE is a variant of the dQ pass");
    constexpr int NW = 4, NT = 256;
    static_assert(D % 8 == 0, "head_dim % 8");
    constexpr int DQK = bwd_dqk(D);
    constexpr int NC = DQK / 32;
    constexpr int DV = (D + 15) / 16 * 16;
    constexpr int NDF = DV / 16;
    constexpr int DCH = D / 8;
    constexpr int ROW = bwd_row(D);   // row-major images (streamed rows x head dim): stride 32 k + 16 elements (see bwd_row)
    constexpr int FB = NW * 16 * QF;  // fixed-side rows per block

    // Two stages of the streamed tile, row-major only: X (K in DQ mode, Q in DKV mode) and Y (V / dO), 64 rows x (DQK + 8) each.  The
    // second product's transposed A operand (Z^T[d][streamed slot]) is read from the SAME images with ds_read_b64_tr_b16 — round 1
    // kept separate transposed images, written with 4-byte LDS stores after a register permute.
    constexpr int IMG = ST * ROW;                  // elements per image
    constexpr int STAGE = 2 * IMG;                 // X then Y
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16_t* const sImg = reinterpret_cast<bf16_t*>(smem_raw);
    float* const sStat = reinterpret_cast<float*>(sImg + 2 * STAGE);  // DKV: [stage][L2[64], delta[64]] of the query tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
#ifdef AE_BWD_LAB
    const int abl = p.abl;
#else
    constexpr int abl = 0;
#endif
    const int nfixed = MODE == MODE_DQ ? p.Nq : p.Nk, nstream = MODE == MODE_DQ ? p.Nk : p.Nq;
    const int nfb = (nfixed + FB - 1) / FB;
    const int nsp = MODE == MODE_DKV ? max(p.nsplit, 1) : 1;
    const int vbs = xcd_remap(blockIdx.x, nfb * p.B * p.H * nsp);
    const int sp = vbs % nsp, vb = vbs / nsp;
    const int bh = vb / nfb, fb = vb % nfb;
    const int b = bh / p.H, h = bh % p.H;
    const int f0 = fb * FB + wave * 16 * QF;

    const bf16_t* qp = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
    const bf16_t* kp = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
    const bf16_t* vp = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
    const bf16_t* dop = p.dO + (long)b * p.o_sb + (long)h * p.o_sh;
    // fixed side (registers): X_f pairs with the streamed X_s in S = X_s X_f^T, Y_f with Y_s in dP = Y_s Y_f^T
    const bf16_t* xf_p = MODE == MODE_DQ ? qp : kp;   const long xf_sn = MODE == MODE_DQ ? p.q_sn : p.k_sn;
    const bf16_t* yf_p = MODE == MODE_DQ ? dop : vp;  const long yf_sn = MODE == MODE_DQ ? p.o_sn : p.v_sn;
    const bf16_t* xs_p = MODE == MODE_DQ ? kp : qp;   const long xs_sn = MODE == MODE_DQ ? p.k_sn : p.q_sn;
    const bf16_t* ys_p = MODE == MODE_DQ ? vp : dop;  const long ys_sn = MODE == MODE_DQ ? p.v_sn : p.o_sn;

    // zero the pad columns of the four row-major images once (the staging below only writes columns < D)
    static_assert(DQK >= DV, "the transposing reads of the second product cover DV columns of a DQK-column image");
    if (DQK > D) {
        for (int i = tid; i < 4 * ST * (DQK - D); i += NT) {
            const int img = i / (ST * (DQK - D)), r = i % (ST * (DQK - D));
            sImg[img * IMG + (r / (DQK - D)) * ROW + D + r % (DQK - D)] = 0;
        }
    }

    bf16x8_t xf[QF][NC], yf[QF][NC];
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        const int row = min(f0 + a * 16 + l15, nfixed - 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int d = c * 32 + lg * 8;   // head-dim columns d .. d + 7 of this lane's K slots; past D: zero (the padded K steps)
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            xf[a][c] = as_bf16x8(d < D ? *reinterpret_cast<const u32x4*>(xf_p + (long)row * xf_sn + d) : z4);
            yf[a][c] = as_bf16x8(d < D ? *reinterpret_cast<const u32x4*>(yf_p + (long)row * yf_sn + d) : z4);
        }
    }

    const float c2 = p.scale * LOG2E;
    const float g = p.out_scale ? p.out_scale[b] : 1.0f;
    float l2f[QF], dlt[QF];  // MODE_DQ: per-lane L2 of its query row, running delta
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        dlt[a] = PRE ? p.delta[((long)b * p.H + h) * p.Nq + min(f0 + a * 16 + l15, p.Nq - 1)] : 0.f;
        l2f[a] = MODE == MODE_DQ ? p.lse[((long)b * p.H + h) * p.Nq + min(f0 + a * 16 + l15, p.Nq - 1)] : 0.f;
    }
    f32x4 acc0[QF][NDF], acc1[QF][NDF];  // DQ: T1, T2 ; DKV: dK^T, dV^T   (lane: [d = 16 df + 4 lg + r][fixed row l15])
#pragma unroll
    for (int a = 0; a < QF; ++a)
#pragma unroll
        for (int df = 0; df < NDF; ++df) acc0[a][df] = acc1[a][df] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntiles = (nstream + ST - 1) / ST;
    const int tpb = (ntiles + nsp - 1) / nsp;                       // streamed tiles of this block: [tb, te)
    const int tb = sp * tpb, te = min(ntiles, tb + tpb);
    // Software pipeline: the global loads of tile t + 1 are issued into registers before tile t is multiplied out of LDS and written
    // to the other stage after it — one barrier per tile, load latency under the MFMAs.
    constexpr int ITEMS = (ST / 2) * DCH;              // (pair of streamed rows) x (16-byte chunk)
    constexpr int NIT = (ITEMS + NT - 1) / NT;
    u32x4 px[NIT][4];                                  // x(row 2pr), x(row 2pr+1), y(row 2pr), y(row 2pr+1)
    float pst[2] = {0.f, 0.f};
    // per-thread source addresses of tile 0 (row pair 2 pr, 16-byte chunk c), advanced by ST rows per tile: a full tile (every tile but a ragged last one) is
    // 4 NIT loads at pointer + constant, no per-tile multiplies and no per-row predicates (round 5: ~40 of the loop's ~200 VALU instructions and nine
    // exec-mask branches per tile were address arithmetic and bounds checks of this staging)
    const bf16_t* bx[NIT]; const bf16_t* by[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(tid + it * NT, ITEMS - 1);
        const int pr = i / DCH, c = i - pr * DCH;
        bx[it] = xs_p + (long)(2 * pr) * xs_sn + c * 8;
        by[it] = ys_p + (long)(2 * pr) * ys_sn + c * 8;
    }
    const float* const lse_bh = p.lse + ((long)b * p.H + h) * p.Nq;
    const float* const dlt_bh = p.delta + ((long)b * p.H + h) * p.Nq;
    auto load_tile = [&](int t) {
        const int t0 = t * ST;
        const long ox = (long)t0 * xs_sn, oy = (long)t0 * ys_sn;
        if (t0 + ST <= nstream) {   // uniform: whole tile in range
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                px[it][0] = *reinterpret_cast<const u32x4*>(bx[it] + ox);
                px[it][1] = *reinterpret_cast<const u32x4*>(bx[it] + ox + xs_sn);
                px[it][2] = *reinterpret_cast<const u32x4*>(by[it] + oy);
                px[it][3] = *reinterpret_cast<const u32x4*>(by[it] + oy + ys_sn);
            }
            if (MODE == MODE_DKV && tid < ST) {
                pst[0] = lse_bh[t0 + tid];
                pst[1] = dlt_bh[t0 + tid];
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NT;
            const int pr = min(i, ITEMS - 1) / DCH;
            const int r0 = t0 + 2 * pr, r1 = r0 + 1;
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            px[it][0] = r0 < nstream ? *reinterpret_cast<const u32x4*>(bx[it] + ox) : z4;
            px[it][1] = r1 < nstream ? *reinterpret_cast<const u32x4*>(bx[it] + ox + xs_sn) : z4;
            px[it][2] = r0 < nstream ? *reinterpret_cast<const u32x4*>(by[it] + oy) : z4;
            px[it][3] = r1 < nstream ? *reinterpret_cast<const u32x4*>(by[it] + oy + ys_sn) : z4;
        }
        if (MODE == MODE_DKV && tid < ST) {
            const int qrow = t0 + tid;
            pst[0] = qrow < p.Nq ? lse_bh[qrow] : 1.0e30f;  // padding queries: P = 2^(s - 1e30) = 0
            pst[1] = qrow < p.Nq ? dlt_bh[qrow] : 0.f;
        }
    };
    auto store_tile = [&](int stage) {
        bf16_t* const dX = sImg + stage * STAGE;
        bf16_t* const dY = dX + IMG;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NT;
            if (i >= ITEMS) break;
            const int pr = i / DCH, c = i - pr * DCH;
            *reinterpret_cast<u32x4*>(dX + (2 * pr) * ROW + c * 8) = px[it][0];
            *reinterpret_cast<u32x4*>(dX + (2 * pr + 1) * ROW + c * 8) = px[it][1];
            *reinterpret_cast<u32x4*>(dY + (2 * pr) * ROW + c * 8) = px[it][2];
            *reinterpret_cast<u32x4*>(dY + (2 * pr + 1) * ROW + c * 8) = px[it][3];
        }
        if (MODE == MODE_DKV && tid < ST) {
            sStat[stage * 2 * ST + tid] = pst[0];
            sStat[stage * 2 * ST + ST + tid] = pst[1];
        }
    };
    load_tile(tb);
    __syncthreads();  // pad columns zeroed
    store_tile(0);
    __syncthreads();
    for (int t = tb; t < te; ++t) {
        const int t0 = t * ST;
        const int cur = (t - tb) & 1;
        const bf16_t* const sX = sImg + cur * STAGE;
        const bf16_t* const sY = sX + IMG;
        const float* const sSt = sStat + cur * 2 * ST;
        if (t + 1 < te && !(abl & 16)) load_tile(t + 1);

        // ---- S_T = X_s X_f^T and dP_T = Y_s Y_f^T : lane holds [streamed row 16 f + 4 lg + r][fixed row l15] -----------------
        // Every LDS operand of this iteration is read ONE STEP AHEAD of the MFMAs / VALU block that consumes it (fragment f + 1 while fragment f is multiplied,
        // the statistics of fragment f + 1 under the exp2 block of fragment f, the transposed operands of product (df, j) + 1 under the MFMAs of (df, j)), with a
        // scheduling barrier after each issue so that hipcc keeps the distance: round 5's PMC + listing showed 28 s_waitcnt per tile, most of them directly
        // between a ds_read and its first use — with two waves per SIMD the pass was LDS-LATENCY-bound (halving the VALU work or removing the bank
        // conflicts did not move it; matrix pipe 37 % busy).
        f32x4 s[QF][4], dp[QF][4];
        bf16x8_t xk[2][NC], yk[2][NC];
        auto ld1 = [&](int f, int buf) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                xk[buf][c] = as_bf16x8(*reinterpret_cast<const u32x4*>(sX + (f * 16 + l15) * ROW + c * 32 + lg * 8));
                yk[buf][c] = as_bf16x8(*reinterpret_cast<const u32x4*>(sY + (f * 16 + l15) * ROW + c * 32 + lg * 8));
            }
        };
        f32x4 lrow[2], drow[2];
        auto ldst = [&](int f, int buf) {
            if (MODE == MODE_DKV) {
                lrow[buf] = *reinterpret_cast<const f32x4*>(sSt + f * 16 + lg * 4);
                drow[buf] = *reinterpret_cast<const f32x4*>(sSt + ST + f * 16 + lg * 4);
            }
        };
        ld1(0, 0);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f + 1 < 4) ld1(f + 1, (f + 1) & 1);
            else ldst(0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < QF; ++a) s[a][f] = dp[a][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (!(abl & 4))
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int a = 0; a < QF; ++a) {
                    s[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xk[f & 1][c], xf[a][c], s[a][f], 0, 0, 0);
                    dp[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yk[f & 1][c], yf[a][c], dp[a][f], 0, 0, 0);
                }
            }
        }

        mfma_fence_begin();   // results of the last MFMA group issued above
#pragma unroll
        for (int a = 0; a < QF; ++a) { mfma_fence_read(s[a][3]); mfma_fence_read(dp[a][3]); }
        mfma_fence_end();

        // ---- P and the second-product operands --------------------------------------------------------------------------------
        bf16x8_t rb0[QF][2], rb1[QF][2];  // DQ: (P o dP, P) ; DKV: (dS, P)
        const int trow = l15 >> 2, tcol = 4 * (l15 & 3);
        bf16x8_t z0[2], z1[2];
        // contraction slot (group lg, e) of half j = streamed row 16 (2j + e / 4) + 4 lg + e % 4 (the accumulator order of rb): two
        // transposing reads of 4 consecutive rows each deliver this lane's 8 values of head-dim column 16 df + l15
        auto ldz = [&](int idx, int buf) {
            const int df = idx >> 1, j = idx & 1;
            const int a0 = (32 * j + 4 * lg + trow) * ROW + df * 16 + tcol;
            z0[buf] = cat_tr(lds_tr16(sX + a0), lds_tr16(sX + a0 + 16 * ROW));
            if (MODE == MODE_DKV) z1[buf] = cat_tr(lds_tr16(sY + a0), lds_tr16(sY + a0 + 16 * ROW));   // DQ: both products use K^T ; DKV: dV uses dO^T
        };
        float r0v[QF][2][4], r1v[QF][2][4];   // the two fragments of the current half j
        const f32x2 c2v = {c2, c2};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            if (f + 1 < 4) ldst(f + 1, (f + 1) & 1);
            else ldz(0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (abl & 2) {
#pragma unroll
                for (int a = 0; a < QF; ++a) { rb0[a][f >> 1] = as_bf16x8((u32x4){__float_as_uint(s[a][f][0]), 0u, 0u, 0u}); rb1[a][f >> 1] = as_bf16x8((u32x4){__float_as_uint(dp[a][f][0]), 0u, 0u, 0u}); }
                continue;
            }
#pragma unroll
            for (int a = 0; a < QF; ++a) {
                // two elements per instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): the logit rebase, dP - delta and the product are packed fp32 math,
                // only the exp2 is per element
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x2 sv = {s[a][f][2 * hh], s[a][f][2 * hh + 1]}, dpv = {dp[a][f][2 * hh], dp[a][f][2 * hh + 1]};
                    const f32x2 lv = MODE == MODE_DQ ? (f32x2){l2f[a], l2f[a]} : (f32x2){lrow[f & 1][2 * hh], lrow[f & 1][2 * hh + 1]};
                    const f32x2 tv = __builtin_elementwise_fma(sv, c2v, -lv);
                    f32x2 pv = {__builtin_amdgcn_exp2f(tv.x), __builtin_amdgcn_exp2f(tv.y)};
                    f32x2 r0, r1;
                    if (MODE == MODE_DQ) {
                        const int k0 = t0 + f * 16 + lg * 4 + 2 * hh;
                        if (k0 >= p.Nk) pv.x = 0.f;          // padding keys
                        if (k0 + 1 >= p.Nk) pv.y = 0.f;
                        if (PRE) {
                            r0 = pv * (dpv - (f32x2){dlt[a], dlt[a]});
                            r1 = (f32x2){0.f, 0.f};
                        } else {
                            r0 = pv * dpv;
                            dlt[a] += r0.x;
                            dlt[a] += r0.y;
                            r1 = pv;
                        }
                    } else {
                        r0 = pv * (dpv - (f32x2){drow[f & 1][2 * hh], drow[f & 1][2 * hh + 1]});
                        r1 = pv;
                    }
                    r0v[a][f & 1][2 * hh] = r0.x; r0v[a][f & 1][2 * hh + 1] = r0.y;
                    r1v[a][f & 1][2 * hh] = r1.x; r1v[a][f & 1][2 * hh + 1] = r1.y;
                }
                if (f & 1) {
                    const int j = f >> 1;
                    u32x4 w0, w1;
                    w0.x = pack_bf16x2(r0v[a][0][0], r0v[a][0][1]); w0.y = pack_bf16x2(r0v[a][0][2], r0v[a][0][3]);
                    w0.z = pack_bf16x2(r0v[a][1][0], r0v[a][1][1]); w0.w = pack_bf16x2(r0v[a][1][2], r0v[a][1][3]);
                    rb0[a][j] = as_bf16x8(w0);
                    if (!PRE) {
                        w1.x = pack_bf16x2(r1v[a][0][0], r1v[a][0][1]); w1.y = pack_bf16x2(r1v[a][0][2], r1v[a][0][3]);
                        w1.z = pack_bf16x2(r1v[a][1][0], r1v[a][1][1]); w1.w = pack_bf16x2(r1v[a][1][2], r1v[a][1][3]);
                        rb1[a][j] = as_bf16x8(w1);
                    }
                }
            }
        }

        // ---- acc^T[d][fixed] += Z^T[d][streamed] R[streamed][fixed] -------------------------------------------------------------
#pragma unroll
        for (int idx = 0; idx < 2 * NDF; ++idx) {
            if (idx + 1 < 2 * NDF) ldz(idx + 1, (idx + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const int df = idx >> 1, j = idx & 1;
            if (abl & 1) {
#pragma unroll
                for (int a = 0; a < QF; ++a) { acc0[a][df][0] += lab_f(z0[idx & 1]) + lab_f(rb0[a][j]); if (!PRE) acc1[a][df][0] += lab_f(rb1[a][j]); }
                continue;
            }
#pragma unroll
            for (int a = 0; a < QF; ++a) {
                acc0[a][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(z0[idx & 1], rb0[a][j], acc0[a][df], 0, 0, 0);
                if (!PRE) acc1[a][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(MODE == MODE_DKV ? z1[idx & 1] : z0[idx & 1], rb1[a][j], acc1[a][df], 0, 0, 0);
            }
        }
        mfma_fence_begin();   // the last MFMA group issued above: the staging / next tile's address math may re-use rb / z registers
#pragma unroll
        for (int a = 0; a < QF; ++a) { mfma_fence_read(acc0[a][NDF - 1]); if (!PRE) mfma_fence_read(acc1[a][NDF - 1]); }
        mfma_fence_end();
        if (abl & 8) continue;
        if (t + 1 < te) store_tile(cur ^ 1);  // the other stage was last read before the previous barrier
        __syncthreads();
    }

    // ---- epilogue ---------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        const int row = f0 + a * 16 + l15;
        if (MODE == MODE_DQ) {
            float dsum = 0.f;
            if (!PRE) {
                dsum = dlt[a];
                dsum += __shfl_xor(dsum, 16, 64);
                dsum += __shfl_xor(dsum, 32, 64);
                if (row < p.Nq && lg == 0) p.delta[((long)b * p.H + h) * p.Nq + row] = dsum;
            }
            if (row >= p.Nq) continue;
            bf16_t* dst = p.dq + (long)b * p.dq_sb + (long)h * p.dq_sh + (long)row * p.dq_sn;
            const float gs = g * p.scale;
#pragma unroll
            for (int df = 0; df < NDF; ++df) {
                const int d = df * 16 + lg * 4;
                if (d >= D) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = PRE ? gs * acc0[a][df][r] : gs * (acc0[a][df][r] - dsum * acc1[a][df][r]);
                if (p.accum_dq) {
                    const u32x2 old = *reinterpret_cast<const u32x2*>(dst + d);
                    o[0] += bf16lo(old.x); o[1] += bf16hi(old.x); o[2] += bf16lo(old.y); o[3] += bf16hi(old.y);
                }
                *reinterpret_cast<u32x2*>(dst + d) = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        } else {
            if (row >= p.Nk) continue;
            const float gs = g * p.scale;
            if (nsp > 1) {  // fp32 partial of this block's query range
                float* const pr = p.part + (((long)sp * p.B * p.H + bh) * p.Nk + row) * (2 * NDF * 16);
#pragma unroll
                for (int df = 0; df < NDF; ++df) {
                    const int d = df * 16 + lg * 4;
                    *reinterpret_cast<f32x4*>(pr + d) = (f32x4){gs * acc0[a][df][0], gs * acc0[a][df][1], gs * acc0[a][df][2], gs * acc0[a][df][3]};
                    *reinterpret_cast<f32x4*>(pr + NDF * 16 + d) = (f32x4){g * acc1[a][df][0], g * acc1[a][df][1], g * acc1[a][df][2], g * acc1[a][df][3]};
                }
                continue;
            }
            bf16_t* dkd = p.dk + (long)b * p.dk_sb + (long)h * p.dk_sh + (long)row * p.dk_sn;
            bf16_t* dvd = p.dv + (long)b * p.dv_sb + (long)h * p.dv_sh + (long)row * p.dv_sn;
#pragma unroll
            for (int df = 0; df < NDF; ++df) {
                const int d = df * 16 + lg * 4;
                if (d >= D) continue;
                *reinterpret_cast<u32x2*>(dkd + d) = (u32x2){pack_bf16x2(gs * acc0[a][df][0], gs * acc0[a][df][1]), pack_bf16x2(gs * acc0[a][df][2], gs * acc0[a][df][3])};
                *reinterpret_cast<u32x2*>(dvd + d) = (u32x2){pack_bf16x2(g * acc1[a][df][0], g * acc1[a][df][1]), pack_bf16x2(g * acc1[a][df][2], g * acc1[a][df][3])};
            }
        }
    }
}

// Sum of the query-range partials of a split dK / dV pass, in split order; one thread per (b h, key row, dK | dV, 4 head-dim columns).
template <int D>
__global__ __launch_bounds__(256) void attn_bwd_reduce_kernel(const BwdArgs p) {
    constexpr int DVP = (D + 15) / 16 * 16;
    const long n = (long)p.B * p.H * p.Nk * 2 * (DVP / 4);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int d = (int)(i % (DVP / 4)) * 4;
    const int which = (int)((i / (DVP / 4)) % 2);
    const long br = i / (2 * (DVP / 4));               // (b h) * Nk + row
    const int row = (int)(br % p.Nk);
    const int bh = (int)(br / p.Nk);
    if (d >= D) return;
    const long stride = (long)p.B * p.H * p.Nk * 2 * DVP;
    const float* src = p.part + br * (2 * DVP) + which * DVP + d;
    f32x4 s = *reinterpret_cast<const f32x4*>(src);
    for (int k = 1; k < p.nsplit; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + k * stride);
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int b = bh / p.H, h = bh % p.H;
    bf16_t* dst = which == 0 ? p.dk + (long)b * p.dk_sb + (long)h * p.dk_sh + (long)row * p.dk_sn : p.dv + (long)b * p.dv_sb + (long)h * p.dv_sh + (long)row * p.dv_sn;
    *reinterpret_cast<u32x2*>(dst + d) = (u32x2){pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3])};
}

// Query-range split of the dK / dV pass (see BwdArgs::part).  Only grids that leave most of the chip idle and stream at least 8 query tiles are cut: the
// UNet's cross-attention layers at training batch 4 (B H = 32 blocks of 64 / 16 tiles: 102 / 41 us per pass before).  AE_ATTN_BWD_SPLIT=0 turns it off.
int bwd_nsplit(int B, int H, int Nq, int Nk, int QF) {
    static const int on = getenv("AE_ATTN_BWD_SPLIT") ? atoi(getenv("AE_ATTN_BWD_SPLIT")) : 1;
    const int FB = 4 * 16 * QF;
    const long blocks0 = (long)((Nk + FB - 1) / FB) * B * H;
    const int ntiles = (Nq + ST - 1) / ST;
    if (!on || blocks0 >= 128 || ntiles < 8) return 1;
    long s = (512 + blocks0 - 1) / blocks0;
    if (s > ntiles / 2) s = ntiles / 2;
    if (s > 32) s = 32;
    return (int)s;
}

template <int D, int QF, int MODE, bool PRE = false>
int launch_bwd(const BwdArgs& a, hipStream_t stream) {
    constexpr int DV = (D + 15) / 16 * 16;
    constexpr size_t lds = (size_t)(4 * ST * bwd_row(D)) * sizeof(bf16_t) + 4 * ST * sizeof(float);  // two stages of (X, Y) + (L2, delta)
    constexpr int FB = 4 * 16 * QF;
    const int nfixed = MODE == MODE_DQ ? a.Nq : a.Nk;
    const long blocks = (long)((nfixed + FB - 1) / FB) * a.B * a.H * (MODE == MODE_DKV && a.nsplit > 1 ? a.nsplit : 1);
    if (lds > 64 * 1024) {
        static bool done = false;
        if (!done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<D, QF, MODE, PRE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                ae_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu) failed", lds);
                return AE_ERR_LAUNCH;
            }
            done = true;
        }
    }
    hipLaunchKernelGGL((attn_bwd_kernel<D, QF, MODE, PRE>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    int rc = ae_check_launch(MODE == MODE_DQ ? "ae_attn_bwd_bf16(dQ)" : "ae_attn_bwd_bf16(dK,dV)");
    if (rc || MODE != MODE_DKV || a.nsplit <= 1) return rc;
    const long n = (long)a.B * a.H * a.Nk * 2 * (DV / 4);
    hipLaunchKernelGGL((attn_bwd_reduce_kernel<D>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a);
    return ae_check_launch("ae_attn_bwd_bf16(dK,dV reduce)");
}

template <int D, int QF>
int launch_both(const BwdArgs& a_in, const bf16_t* out, float* workspace, hipStream_t stream) {
    BwdArgs a = a_in;
    a.nsplit = (workspace && a.dk) ? bwd_nsplit(a.B, a.H, a.Nq, a.Nk, QF) : 1;
    a.part = a.nsplit > 1 ? workspace : nullptr;
    int rc;
    if (out) {  // single-segment attention whose own output is known: delta = rowsum(dO o O) up front, one accumulator set in the dQ pass
        const long n = (long)a.B * a.Nq * a.H;
        hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a.dO, out, a.delta, a.B, a.H, a.Nq, a.o_sb, a.o_sh,
                           a.o_sn);
        rc = ae_check_launch("ae_attn_bwd_bf16(delta)");
        if (rc) return rc;
        rc = launch_bwd<D, QF, MODE_DQ, true>(a, stream);
    } else {
        // head dims 64 / 80 with TWO accumulator sets (no pre-pass delta): at two query fragments the pass needs 328 registers and hipcc shuffles accumulators
        // through AGPRs between the first product's MFMAs — v_accvgpr_read into the operand registers of an MFMA issued one instruction earlier, the write the
        // hardware does not interlock (tools/isa_audit.py flagged 28 such writes; the pre-pass form and the dK / dV pass are clean).  One fragment: 200 registers.
        rc = launch_bwd<D, (D >= 64 && D <= 80 ? 1 : QF), MODE_DQ>(a, stream);
    }
    if (rc) return rc;
    if (!a.dk) return 0;  // caller only needs dQ (frozen key/value side)
    return launch_bwd<D, QF, MODE_DKV>(a, stream);
}

}  // namespace

// fp32 elements of the optional workspace of ae_attn_bwd_bf16 (0: this shape's dK / dV pass is not split).
extern "C" long ae_attn_bwd_workspace_floats(int B, int H, int Nq, int Nk, int D) {
    if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || D <= 0) return 0;
    const int QF = D >= 96 ? 1 : 2;
    const int ns = bwd_nsplit(B, H, Nq, Nk, QF);
    return ns > 1 ? (long)ns * B * H * Nk * 2 * ((D + 15) / 16 * 16) : 0;
}

// Gradients of ae_attn_fwd_bf16 for one key/value segment.  `lse` is the forward's log2-domain log-sum-exp for THIS segment;
// `delta` ([B,H,Nq] fp32) is an output (rowsum(P o dP) with the UN-scaled dO: summed over heads and rows it is the gradient of
// the segment's out_scale).  dk / dv may both be NULL when only dQ is needed.  dq is overwritten, or accumulated if accumulate_dq.
extern "C" int ae_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dout, const void* out, const float* lse, float* delta,
                                void* dq, void* dk, void* dv, int B, int H, int Nq, int Nk, int D,
                                long q_sb, long q_sh, long q_sn, long k_sb, long k_sh, long k_sn, long v_sb, long v_sh, long v_sn,
                                long o_sb, long o_sh, long o_sn, long dq_sb, long dq_sh, long dq_sn, long dk_sb, long dk_sh, long dk_sn,
                                long dv_sb, long dv_sh, long dv_sn, float scale, const float* out_scale, int accumulate_dq,
                                float* workspace, void* stream) {
    AE_REQUIRE(q && k && v && dout && lse && delta && dq, "ae_attn_bwd_bf16: null pointer");
    AE_REQUIRE(((uintptr_t)workspace & 15) == 0, "ae_attn_bwd_bf16: workspace must be 16-byte aligned");
    AE_REQUIRE((dk == nullptr) == (dv == nullptr), "ae_attn_bwd_bf16: dk and dv go together");
    AE_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "ae_attn_bwd_bf16: bad sizes B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    AE_REQUIRE((q_sb | q_sh | q_sn | k_sb | k_sh | k_sn | v_sb | v_sh | v_sn | o_sb | o_sh | o_sn) % 8 == 0,
               "ae_attn_bwd_bf16: input strides must keep rows 16-byte aligned");
    AE_REQUIRE((dq_sb | dq_sh | dq_sn | dk_sb | dk_sh | dk_sn | dv_sb | dv_sh | dv_sn) % 4 == 0, "ae_attn_bwd_bf16: gradient strides must keep rows 8-byte aligned");
    AE_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)dout & 15) == 0 &&
                   ((uintptr_t)dq & 7) == 0 && ((uintptr_t)dk & 7) == 0 && ((uintptr_t)dv & 7) == 0,
               "ae_attn_bwd_bf16: pointer alignment");
    BwdArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.dO = (const bf16_t*)dout;
    a.lse = lse; a.delta = delta; a.out_scale = out_scale;
    a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk;
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_sn = q_sn; a.k_sb = k_sb; a.k_sh = k_sh; a.k_sn = k_sn;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_sn = v_sn; a.o_sb = o_sb; a.o_sh = o_sh; a.o_sn = o_sn;
    a.dq_sb = dq_sb; a.dq_sh = dq_sh; a.dq_sn = dq_sn; a.dk_sb = dk_sb; a.dk_sh = dk_sh; a.dk_sn = dk_sn;
    a.dv_sb = dv_sb; a.dv_sh = dv_sh; a.dv_sn = dv_sn;
    a.scale = scale; a.accum_dq = accumulate_dq;
#ifdef AE_BWD_LAB
    a.abl = getenv("AE_BWD_ABL") ? atoi(getenv("AE_BWD_ABL")) : 0;
#endif
    hipStream_t s = (hipStream_t)stream;
    // `out` (optional): THIS segment's own forward output, same layout as dout.  Only usable when nothing else was folded into it.
    AE_REQUIRE(!out || (((uintptr_t)out & 15) == 0), "ae_attn_bwd_bf16: out must be 16-byte aligned");
    const bf16_t* o_own = (out && !out_scale && !accumulate_dq) ? (const bf16_t*)out : nullptr;
    switch (D) {
        case 8: return launch_both<8, 2>(a, o_own, workspace, s);
        case 16: return launch_both<16, 2>(a, o_own, workspace, s);
        case 32: return launch_both<32, 2>(a, o_own, workspace, s);
        case 40: return launch_both<40, 2>(a, o_own, workspace, s);   // (64 fixed rows per block, three blocks per CU: 646 vs 552 us for the two passes at N = 4096 — operand reuse beats occupancy here)
        case 48: return launch_both<48, 2>(a, o_own, workspace, s);
        case 64: return launch_both<64, 2>(a, o_own, workspace, s);
        case 80: return launch_both<80, 2>(a, o_own, workspace, s);
        case 96: return launch_both<96, 1>(a, o_own, workspace, s);
        case 128: return launch_both<128, 1>(a, o_own, workspace, s);
        case 160: return launch_both<160, 1>(a, o_own, workspace, s);
        default:
            ae_set_error("ae_attn_bwd_bf16: unsupported head_dim %d (supported: 8,16,32,40,48,64,80,96,128,160)", D);
            return AE_ERR_UNSUPPORTED;
    }
}
