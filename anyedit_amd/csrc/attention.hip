// Fused multi-head attention forward for gfx950 (MI355X): out = softmax(q k^T * scale + bias) v
//
// Replaces:
//   A1/A2  ldm/modules/attention.py:163-194  CrossAttention.forward (self + cross; fp32 logits, optional bool mask)
//          and the xformers call it swaps in at attention.py:222-233;
//   A10    segment_anything/.../image_encoder.py:224-240 Attention.forward with the decomposed relative-position
//          bias of :325-361 added as rel_h[q, key/kW] + rel_w[q, key%kW] (computed from the UNSCALED q, G13);
//   A9     the decoupled expert attention of our AnySD spec: out += gate_b * Attn(q, K_ip, V_ip) (accumulate mode).
//
// Design (wave64, flash-style online softmax, never materialises the [N,N] logits):
//   * 256 threads = 4 waves; each wave owns 16*QF query rows and keeps Q in registers as MFMA operands;
//   * K/V are streamed in 64-key tiles through LDS (double-buffered when it fits: one barrier per tile), global
//     loads issued one tile ahead into registers, branch-free (row indices clamped instead of predicated);
//   * logits are computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_16x16x32_bf16, so a lane holds 16 logits of
//     ONE query row per tile: row max / row sum need 2 cross-lane steps, and the exponentiated P registers are
//     already laid out as the B operand of the O^T = V^T P^T MFMA (contraction slots permuted consistently on
//     both operands -> no cross-lane traffic for P);
//   * V is transposed while it is written to LDS (key index permuted so each lane's 8 contraction slots are one
//     16-byte ds_read_b128);
//   * head_dim is padded inside the kernel (40 -> 64 for QK^T, 48 for PV); q/k/v/out are addressed through
//     (batch, head, row) strides, so the 'b n (h d) -> (b h) n d' rearranges of the reference never happen;
//   * softmax in fp32, exp2 domain; bf16 P (v_cvt_pk_bf16_f32); fp32 accumulation of O.
#include "attention.hpp"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KT = 64;  // keys per tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;  // finite "masked" logit: behaves like masked_fill(-finfo.max) (attention.py:186-187)

__device__ __forceinline__ int vt_pos(int key) {  // key = 16 f + 4 g + r  ->  16 g + 4 f + r
    return ((key >> 2) & 3) * 16 + (key >> 4) * 4 + (key & 3);
}

// NW waves per block, QF 16-row query fragments per wave.  (NW=4,QF=2) and (NW=8,QF=1) cover the same 128 query rows per
// block with the same LDS; the latter halves the per-wave register state (4 instead of 2 waves per SIMD -> the exp-bound
// softmax of one wave overlaps the MFMAs of three others) at the price of twice the K/V fragment reads per FLOP.
// BIAS: 0 none; 1 decomposed rel-pos bias, any key grid; 2 the same when one key tile is exactly one key ROW (kW == 64, SAM's
// global attention on the 64x64 token grid): rel_w[q, kw] of the lane's 16 key slots lives in registers for the whole kernel
// and rel_h[q, kh] is one value per tile — no per-element index arithmetic or gathers.
// OCC4: cap the kernel at 128 VGPRs (4 waves per SIMD = two 8-wave blocks per CU) — the rel-pos variants otherwise sit at 145-161
// VGPRs, i.e. ONE 8-wave block per CU; the cap costs 40-160 bytes of scratch per lane.
template <int D, int QF, int NW, bool DBUF, int BIAS, bool HAS_MASK, bool SEG2 = false, bool OCC4 = false>
__global__ __launch_bounds__(64 * NW, (D <= 96 ? ((NW == 8 && (BIAS == 0 || OCC4)) ? 4 : 2) : 1)) void attn_kernel(const AttnArgs p) {
    constexpr bool HAS_BIAS = BIAS != 0;
    constexpr int NT = 64 * NW;
    constexpr int NC = D / 32;                 // full K=32 MFMAs per (key frag, q frag)
    constexpr bool TAIL16 = (D % 32) != 0;     // head-dim remainder (8 or 16) goes through ONE K=16 MFMA instead of padding to 32
    static_assert(D % 32 == 0 || D % 32 == 8 || D % 32 == 16, "head_dim % 32 must be 0, 8 or 16");
    constexpr int DQK = NC * 32 + (TAIL16 ? 16 : 0);
    constexpr int DV = (D + 15) / 16 * 16;
    constexpr bool ONES = DV > D;  // a free padding row of V^T holds 1.0: the PV MFMA then also produces sum_k P[q][k] (the softmax
                                   // denominator, rescaled together with O) and the 32 VALU adds per tile disappear
    constexpr int NDF = DV / 16;   // 16-row fragments of O^T
    constexpr int DCH = D / 8;     // 16-byte chunks per K/V row
    constexpr int KROW = DQK + 8;  // LDS row strides (elements), +16 B pad
    constexpr int VROW = KT + 8;
    constexpr int KCH = (KT * DCH + NT - 1) / NT;
    constexpr int VCH = ((KT / 2) * DCH + NT - 1) / NT;
    constexpr int NBUF = DBUF ? 2 : 1;
    constexpr int KSZ = KT * KROW, VSZ = DV * VROW;
    static_assert(D % 8 == 0, "head_dim must be a multiple of 8");

    __shared__ __attribute__((aligned(16))) bf16_t smem[NBUF * (KSZ + VSZ)];
    bf16_t* const sK = smem;
    bf16_t* const sVt = smem + NBUF * KSZ;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int QB = NW * 16 * QF;  // query rows per block
    const int nqb = (p.Nq + QB - 1) / QB;
    const int vb = xcd_remap(blockIdx.x, nqb * p.B * p.H);
    const int bh = vb / nqb, qb = vb % nqb;
    const int b = bh / p.H, h = bh % p.H;
    const int q0 = qb * QB + wave * 16 * QF;

    const bf16_t* qp = p.q + (long)b * p.q_sb + (long)h * p.q_sh;
    const bf16_t* kp = p.k + (long)b * p.k_sb + (long)h * p.k_sh;
    const bf16_t* vp = p.v + (long)b * p.v_sb + (long)h * p.v_sh;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // zero the pad columns of K (d in [D, DQK)) and pad rows of V^T (d in [D, DV)) once, in every buffer
    if (DQK > D) {
        for (int i = tid; i < NBUF * KT * (DQK - D); i += NT) sK[(i / (DQK - D)) * KROW + D + i % (DQK - D)] = 0;
    }
    if (DV > D) {
        for (int i = tid; i < NBUF * (DV - D) * VROW; i += NT) {
            const int bufi = i / ((DV - D) * VROW), r = i % ((DV - D) * VROW);
            sVt[bufi * VSZ + D * VROW + r] = (ONES && r < VROW) ? (bf16_t)0x3F80 : (bf16_t)0;  // row D = ones
        }
    }

    // ---- Q fragments (B operand of S^T = K Q^T): lane (q = l15, g) holds Q[q][32c + 8g .. +8]; rows clamped, pad zeroed
    bf16x8_t qf[QF][NC > 0 ? NC : 1];
    s16x4_t qt[QF];  // K=16 tail operand: lane (q = l15, g) holds Q[q][32 NC + 4g .. +4]
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        const int qrow = min(q0 + a * 16 + l15, p.Nq - 1);
#pragma unroll
        for (int c = 0; c < NC; ++c)
            qf[a][c] = as_bf16x8(*reinterpret_cast<const u32x4*>(qp + (long)qrow * p.q_sn + c * 32 + lg * 8));
        if (TAIL16) {
            const int d = NC * 32 + lg * 4;
            const u32x2 t = *reinterpret_cast<const u32x2*>(qp + (long)qrow * p.q_sn + (d < D ? d : 0));
            qt[a] = as_s16x4(d < D ? t : (u32x2){0u, 0u});
        }
    }

    // BIAS 1: this lane's rows of the two bias tables (32-bit indexing inside a row)
    const float* rh_row[BIAS == 1 ? QF : 1];
    const float* rw_row[BIAS == 1 ? QF : 1];
    const float inv_kw = BIAS == 1 ? 1.0f / (float)p.kW : 0.f;
    // small key grids (SAM's 14x14 windows): the block's bias rows [query][kH + kW] are copied into LDS once — the per-key
    // lookups are index-dependent, and as global loads each one exposed a full memory round trip (3x the kernel time)
    constexpr int BIAS_LDS_W = 32;                 // kH, kW <= 32 each
    constexpr int BQ = NW * 16 * QF;               // queries per block
    __shared__ float sBiasH[BIAS == 1 ? BQ * BIAS_LDS_W : 1];
    __shared__ float sBiasW[BIAS == 1 ? BQ * BIAS_LDS_W : 1];
    const bool bias_in_lds = BIAS == 1 && p.kH <= BIAS_LDS_W && p.kW <= BIAS_LDS_W;
    if (BIAS == 1) {
        if (bias_in_lds) {
            // the block's rows of each table are one contiguous range: straight coalesced copies, all loads in flight at once
            const int qfirst = qb * BQ;
            const int nq = min(BQ, p.Nq - qfirst);
            const float* gh = p.rel_h + ((long)bh * p.Nq + qfirst) * p.kH;
            const float* gw = p.rel_w + ((long)bh * p.Nq + qfirst) * p.kW;
            constexpr int IT = (BQ * BIAS_LDS_W + NT - 1) / NT;
            float th[IT], tw[IT];
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int i = tid + j * NT;
                th[j] = i < nq * p.kH ? gh[i] : 0.f;
                tw[j] = i < nq * p.kW ? gw[i] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int i = tid + j * NT;
                if (i < BQ * p.kH) sBiasH[i] = th[j];
                if (i < BQ * p.kW) sBiasW[i] = tw[j];
            }
            __syncthreads();
        }
#pragma unroll
        for (int a = 0; a < QF; ++a) {
            const long qc = (long)bh * p.Nq + min(q0 + a * 16 + l15, p.Nq - 1);
            rh_row[BIAS == 1 ? a : 0] = p.rel_h + qc * p.kH;
            rw_row[BIAS == 1 ? a : 0] = p.rel_w + qc * p.kW;
        }
    }
    f32x4 rw[BIAS == 2 ? QF : 1][4];  // BIAS 2: rel_w[q][16 f + 4 lg + r] * log2(e) of this lane's key slots
    if (BIAS == 2) {
#pragma unroll
        for (int a = 0; a < QF; ++a) {
            const int qc = min(q0 + a * 16 + l15, p.Nq - 1);
            const float* row = p.rel_w + ((long)bh * p.Nq + qc) * p.kW;
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(row + f * 16 + lg * 4);
                rw[BIAS == 2 ? a : 0][f] = (f32x4){t[0] * LOG2E, t[1] * LOG2E, t[2] * LOG2E, t[3] * LOG2E};
            }
        }
    }
    f32x4 o[QF][NDF];
    float m_run[QF], l_run[QF];
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        m_run[a] = NEG_BIG;
        l_run[a] = 0.f;
#pragma unroll
        for (int df = 0; df < NDF; ++df) o[a][df] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // softmax denominator of query fragment a, replicated over the 4 lane groups of its row
    auto row_sum = [&](int a) -> float {
        if (ONES) {  // O^T[d = D][q] lives in lane group (D % 16) / 4, register D % 4 of fragment D / 16
            const float v = o[a][D / 16][D % 4];
            return __shfl(v, l15 + 16 * ((D % 16) / 4), 64);
        }
        float l = l_run[a];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        return l;
    };

    f32x4 o_first[SEG2 ? QF : 1][SEG2 ? NDF : 1];  // normalised result of segment 0 while segment 1 runs
    constexpr int nseg = SEG2 ? 2 : 1;
    for (int seg_i = 0; seg_i < nseg; ++seg_i) {
        const bf16_t* seg_kp = seg_i == 0 ? kp : p.k2 + (long)b * p.k2_sb + (long)h * p.k2_sh;
        const bf16_t* seg_vp = seg_i == 0 ? vp : p.v2 + (long)b * p.v2_sb + (long)h * p.v2_sh;
        const int seg_Nk = seg_i == 0 ? p.Nk : p.Nk2;
        const long seg_ksn = seg_i == 0 ? p.k_sn : p.k2_sn, seg_vsn = seg_i == 0 ? p.v_sn : p.v2_sn;
        // ---- staging: per-thread (row, chunk) assignments are loop invariant
        int k_key[KCH], k_c[KCH], v_pr[VCH], v_c[VCH];
    #pragma unroll
        for (int i = 0; i < KCH; ++i) {
            const int id = min(tid + i * NT, KT * DCH - 1);
            k_key[i] = id / DCH;
            k_c[i] = id - k_key[i] * DCH;
        }
    #pragma unroll
        for (int i = 0; i < VCH; ++i) {
            const int id = min(tid + i * NT, (KT / 2) * DCH - 1);
            v_pr[i] = id / DCH;
            v_c[i] = id - v_pr[i] * DCH;
        }
        u32x4 rk[KCH], rv[VCH][2];
        const int last_key = seg_Nk - 1;
        // Bounds-checked buffer loads, 32-bit byte offsets: one add per load in the loop; keys >= Nk fall past the descriptor's
        // extent and read as 0 in hardware (their P is forced to 0 anyway).  Extent = this (batch, head)'s rows [0, Nk).
        const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(seg_kp), 0, (int)(((long)last_key * seg_ksn + D) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(seg_vp), 0, (int)(((long)last_key * seg_vsn + D) * 2), 0x00020000);
        int k_off[KCH], v_off[VCH];
    #pragma unroll
        for (int i = 0; i < KCH; ++i) k_off[i] = (int)((long)k_key[i] * seg_ksn + k_c[i] * 8) * 2;
    #pragma unroll
        for (int i = 0; i < VCH; ++i) v_off[i] = (int)((long)(2 * v_pr[i]) * seg_vsn + v_c[i] * 8) * 2;
        const int k_tile_bytes = (int)(KT * seg_ksn * 2), v_tile_bytes = (int)(KT * seg_vsn * 2), v_row_bytes = (int)(seg_vsn * 2);
        auto load_kv = [&](int k0) {
            const int tk = (k0 / KT) * k_tile_bytes, tv = (k0 / KT) * v_tile_bytes;  // wave-uniform
    #pragma unroll
            for (int i = 0; i < KCH; ++i) rk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, k_off[i] + tk, 0, 0);
    #pragma unroll
            for (int i = 0; i < VCH; ++i) {
                rv[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsV, v_off[i] + tv, 0, 0);
                rv[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsV, v_off[i] + tv + v_row_bytes, 0, 0);
            }
        };
        auto store_kv = [&](int buf) {
            bf16_t* dK = sK + buf * KSZ;
            bf16_t* dV = sVt + buf * VSZ;
    #pragma unroll
            for (int i = 0; i < KCH; ++i) {
                if (KT * DCH % NT == 0 || tid + i * NT < KT * DCH)
                    *reinterpret_cast<u32x4*>(dK + k_key[i] * KROW + k_c[i] * 8) = rk[i];
            }
    #pragma unroll
            for (int i = 0; i < VCH; ++i) {
                if ((KT / 2) * DCH % NT == 0 || tid + i * NT < (KT / 2) * DCH) {
                    const int pos = vt_pos(2 * v_pr[i]);  // even; key 2pr+1 lands at pos+1
                    const uint32_t a0[4] = {rv[i][0].x, rv[i][0].y, rv[i][0].z, rv[i][0].w};
                    const uint32_t a1[4] = {rv[i][1].x, rv[i][1].y, rv[i][1].z, rv[i][1].w};
    #pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // d = c*8 + 2e (low halves) and c*8 + 2e + 1 (high halves)
                        const uint32_t lo = __builtin_amdgcn_perm(a1[e], a0[e], 0x05040100u);  // {a0.lo16, a1.lo16}: one v_perm_b32
                        const uint32_t hi = __builtin_amdgcn_perm(a1[e], a0[e], 0x07060302u);  // {a0.hi16, a1.hi16}
                        *reinterpret_cast<uint32_t*>(dV + (v_c[i] * 8 + 2 * e) * VROW + pos) = lo;
                        *reinterpret_cast<uint32_t*>(dV + (v_c[i] * 8 + 2 * e + 1) * VROW + pos) = hi;
                    }
                }
            }
        };

        const float c2 = p.scale * LOG2E;
        const int ntiles = (seg_Nk + KT - 1) / KT;
        load_kv(0);
        __syncthreads();  // pad zero-fill ordered before the first tile write
        store_kv(0);
        __syncthreads();

        // One K/V tile: S^T = K Q^T, online softmax, O^T += V^T P^T.  TAIL = the (only) tile that contains padding keys.
        auto tile_body = [&](int k0, int cur, auto tail_tag) {
            constexpr bool TAIL = decltype(tail_tag)::value;
            const bf16_t* cK = sK + cur * KSZ;
            const bf16_t* cV = sVt + cur * VSZ;
            float rh[BIAS == 2 ? QF : 1];  // BIAS 2: rel_h[q][key row of this tile], issued ahead of the QK^T MFMAs
            if (BIAS == 2) {
#pragma unroll
                for (int a = 0; a < QF; ++a)
                    rh[BIAS == 2 ? a : 0] = p.rel_h[((long)bh * p.Nq + min(q0 + a * 16 + l15, p.Nq - 1)) * p.kH + k0 / KT] * LOG2E;
            }

            // ---- S^T = K Q^T : lane holds S^T[key = 16f + 4g + r][q = l15] --------------------------
            f32x4 s[QF][4];
            // K=16 remainder (d in [32 NC, 32 NC + 16); K pad columns / Q pad lanes are zero) first, for ALL key fragments, then
            // the K=32 chain.  A 16x16x16 and a 16x16x32 MFMA issued close together on one accumulator lose updates on gfx950 in
            // either order (measured; the compiler's dependency spacing is not enough), so the two shapes are kept 4 QF - 1
            // MFMAs apart and the scheduler may not mix the groups.  tests/test_hip_ops.py covers every head dim with a tail.
    #pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (TAIL16) {
                    const s16x4_t kt = as_s16x4(*reinterpret_cast<const u32x2*>(cK + (f * 16 + l15) * KROW + NC * 32 + lg * 4));
    #pragma unroll
                    for (int a = 0; a < QF; ++a)
                        s[a][f] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kt, qt[a], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                } else {
    #pragma unroll
                    for (int a = 0; a < QF; ++a) s[a][f] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            if (TAIL16) __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
            for (int f = 0; f < 4; ++f) {
    #pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const bf16x8_t kf = as_bf16x8(*reinterpret_cast<const u32x4*>(cK + (f * 16 + l15) * KROW + c * 32 + lg * 8));
    #pragma unroll
                    for (int a = 0; a < QF; ++a) s[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[a][c], s[a][f], 0, 0, 0);
                }
            }

            // ---- online softmax (fp32, exp2 domain).  Fast path (no bias / mask / padding): logits stay raw, the scale is
            // folded into one fma per element, and O / l are rescaled only when some row's running max actually moves (exact:
            // otherwise the factor is 1) — a wave-uniform branch that is almost never taken after the first tiles.
            bf16x8_t pb[QF][2];
    #pragma unroll
            for (int a = 0; a < QF; ++a) {
                constexpr bool PLAIN = !HAS_BIAS && !HAS_MASK && !TAIL;
                if (!PLAIN) {
                    float bval[BIAS == 1 ? 4 : 1][4];  // BIAS 1: all 16 lookups are issued before any is consumed
                    if (BIAS == 1) {
    #pragma unroll
                        for (int f = 0; f < 4; ++f)
    #pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                // key -> (row, column) of the key grid without an integer division: (key + 0.5) / kW is never
                                // closer than 0.5 / kW to an integer, far above the fp32 error for any grid that fits a kernel
                                const int key = min(k0 + f * 16 + lg * 4 + r, last_key);
                                const int khh = (int)(((float)key + 0.5f) * inv_kw);
                                const int kww = key - khh * p.kW;
                                if (bias_in_lds) {
                                    const int ql = wave * 16 * QF + a * 16 + l15;
                                    bval[BIAS == 1 ? f : 0][r] = sBiasH[ql * p.kH + khh] + sBiasW[ql * p.kW + kww];
                                } else {
                                    bval[BIAS == 1 ? f : 0][r] = rh_row[BIAS == 1 ? a : 0][khh] + rw_row[BIAS == 1 ? a : 0][kww];
                                }
                            }
                    }
    #pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int kb = k0 + f * 16 + lg * 4;
    #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = s[a][f][r] * c2;
                            if (BIAS == 2) {
                                v = fmaf(s[a][f][r], c2, rw[BIAS == 2 ? a : 0][f][r] + rh[BIAS == 2 ? a : 0]);
                            } else if (HAS_BIAS) {
                                v = fmaf(bval[BIAS == 1 ? f : 0][r], LOG2E, v);
                            }
                            if (HAS_MASK) {
                                if (p.key_mask[(long)b * seg_Nk + min(kb + r, last_key)] == 0) v = NEG_BIG;
                            }
                            if (TAIL) {
                                if (kb + r >= seg_Nk) v = NEG_BIG;
                            }
                            s[a][f][r] = v;
                        }
                    }
                }
                float mx = NEG_BIG;
    #pragma unroll
                for (int f = 0; f < 4; ++f) mx = fmaxf(mx, fmaxf(fmaxf(s[a][f][0], s[a][f][1]), fmaxf(s[a][f][2], s[a][f][3])));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                if (PLAIN) mx *= c2;  // scale > 0
                if (__any(mx > m_run[a])) {
                    const float m_new = fmaxf(m_run[a], mx);
                    const float alpha = __builtin_amdgcn_exp2f(m_run[a] - m_new);
                    m_run[a] = m_new;
                    l_run[a] *= alpha;
    #pragma unroll
                    for (int df = 0; df < NDF; ++df) {
                        o[a][df][0] *= alpha; o[a][df][1] *= alpha; o[a][df][2] *= alpha; o[a][df][3] *= alpha;
                    }
                }
                const float neg_m = -m_run[a];
                // (v_pk_fma_f32 on pairs of exponent arguments was measured: 16.87 vs 16.82 ms per UNet step, same box — the packed
                // form is not faster than two v_fma_f32 here; scalar form kept)
    #pragma unroll
                for (int f = 0; f < 4; ++f)
    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e = PLAIN ? __builtin_amdgcn_exp2f(fmaf(s[a][f][r], c2, neg_m)) : __builtin_amdgcn_exp2f(s[a][f][r] + neg_m);
                        if (TAIL) {  // padding keys never contribute (also when a whole row is masked -> uniform over REAL keys)
                            if (k0 + f * 16 + lg * 4 + r >= seg_Nk) e = 0.f;
                        }
                        s[a][f][r] = e;
                    }
                if (!ONES) {
                float rs = 0.f;
    #pragma unroll
                for (int f = 0; f < 4; ++f) rs += (s[a][f][0] + s[a][f][1]) + (s[a][f][2] + s[a][f][3]);
                l_run[a] += rs;
                }
    #pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 w;
                    w.x = pack_bf16x2(s[a][2 * j][0], s[a][2 * j][1]);
                    w.y = pack_bf16x2(s[a][2 * j][2], s[a][2 * j][3]);
                    w.z = pack_bf16x2(s[a][2 * j + 1][0], s[a][2 * j + 1][1]);
                    w.w = pack_bf16x2(s[a][2 * j + 1][2], s[a][2 * j + 1][3]);
                    pb[a][j] = as_bf16x8(w);
                }
            }

            // ---- O^T += V^T P^T : lane holds O^T[d = 16 df + 4g + r][q = l15] -------------------------
    #pragma unroll
            for (int df = 0; df < NDF; ++df) {
    #pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8_t vf = as_bf16x8(*reinterpret_cast<const u32x4*>(cV + (df * 16 + l15) * VROW + lg * 16 + j * 8));
    #pragma unroll
                    for (int a = 0; a < QF; ++a) o[a][df] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[a][j], o[a][df], 0, 0, 0);
                }
            }
        };

        for (int t = 0; t < ntiles; ++t) {
            const int k0 = t * KT;
            const int cur = DBUF ? (t & 1) : 0;
            if (t + 1 < ntiles) load_kv(k0 + KT);
            if (k0 + KT > seg_Nk) tile_body(k0, cur, std::true_type{});
            else tile_body(k0, cur, std::false_type{});
            if (DBUF) {
                if (t + 1 < ntiles) store_kv(cur ^ 1);
                __syncthreads();
            } else {
                __syncthreads();  // every wave is done reading this tile
                if (t + 1 < ntiles) {
                    store_kv(0);
                    __syncthreads();
                }
            }
        }

        if (SEG2 && seg_i == 0) {  // park segment 0's normalised output, restart the online softmax
#pragma unroll
            for (int a = 0; a < QF; ++a) {
                const float rsum0 = row_sum(a);
                if (p.lse && lg == 0 && q0 + a * 16 + l15 < p.Nq) p.lse[((long)b * p.H + h) * p.Nq + q0 + a * 16 + l15] = m_run[a] + __builtin_amdgcn_logf(rsum0);
                const float inv = 1.0f / rsum0;
#pragma unroll
                for (int df = 0; df < NDF; ++df) {
                    o_first[SEG2 ? a : 0][SEG2 ? df : 0] = (f32x4){o[a][df][0] * inv, o[a][df][1] * inv, o[a][df][2] * inv, o[a][df][3] * inv};
                    o[a][df] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                m_run[a] = NEG_BIG;
                l_run[a] = 0.f;
            }
        }
    }

    // ---- normalise and store: 4 consecutive d per lane -> 8-byte stores ------------------------------
    bf16_t* op = p.o + (long)b * p.o_sb + (long)h * p.o_sh;
#pragma unroll
    for (int a = 0; a < QF; ++a) {
        const float rsum = row_sum(a);
        const float inv = (SEG2 ? p.scale2[b] : (p.out_scale ? p.out_scale[b] : 1.0f)) / rsum;
        const int qrow = q0 + a * 16 + l15;
        if (qrow >= p.Nq) continue;
        float* lse_out = SEG2 ? p.lse2 : p.lse;
        if (lse_out && lg == 0) lse_out[((long)b * p.H + h) * p.Nq + qrow] = m_run[a] + __builtin_amdgcn_logf(rsum);  // v_log_f32 = log2
#pragma unroll
        for (int df = 0; df < NDF; ++df) {
            const int d = df * 16 + lg * 4;
            if (d < D) {
                float r0 = o[a][df][0] * inv, r1 = o[a][df][1] * inv, r2 = o[a][df][2] * inv, r3 = o[a][df][3] * inv;
                if (SEG2) {
                    const f32x4 f0 = o_first[SEG2 ? a : 0][SEG2 ? df : 0];
                    r0 += f0[0]; r1 += f0[1]; r2 += f0[2]; r3 += f0[3];
                }
                u32x2* dst = reinterpret_cast<u32x2*>(op + (long)qrow * p.o_sn + d);
                if (p.accum) {  // decoupled adapter attention: out += gate * Attn(q, K_ip, V_ip)
                    const u32x2 prev = *dst;
                    r0 += bf16lo(prev.x); r1 += bf16hi(prev.x); r2 += bf16lo(prev.y); r3 += bf16hi(prev.y);
                }
                *dst = (u32x2){pack_bf16x2(r0, r1), pack_bf16x2(r2, r3)};
            }
        }
    }
}

// tuning knob: which rel-pos (SAM) variants run under the 128-VGPR cap: 0 none, 1 the windowed (table-lookup) variant, 4 both.
// In-situ A/B on the ViT-H encoder (one box, 2 runs each): windowed 14x14 attention 61.0 -> 49.3 us per block (two resident
// blocks hide its short dependent load phases), global 64x64 attention 287 -> 414 us (the spills land in its long key loop): default 1.
inline int sam_occ() {
    static const int v = getenv("AE_ATTN_SAM_OCC") ? atoi(getenv("AE_ATTN_SAM_OCC")) : 1;
    return v;
}

template <int D, int QF, bool DBUF, int NW = 4>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
    constexpr int QB = NW * 16 * QF;
    constexpr int NT = 64 * NW;
    const long blocks = (long)((a.Nq + QB - 1) / QB) * a.B * a.H;
    dim3 grid((unsigned)blocks), block(NT);
    if (a.rel_h && a.key_mask) { ae_set_error("ae_attn_fwd_bf16: rel-pos bias together with key_mask is not supported"); return AE_ERR_UNSUPPORTED; }
    if (a.k2) {
        if constexpr (D <= 96 || (D == 160 && QF == 1)) hipLaunchKernelGGL((attn_kernel<D, QF, NW, DBUF, 0, false, true>), grid, block, 0, stream, a);
    } else if (a.rel_h && NW == 8 && a.kW == KT && a.Nk % KT == 0 && sam_occ() == 4) {
        hipLaunchKernelGGL((attn_kernel<D, QF, NW, DBUF, 2, false, false, NW == 8>), grid, block, 0, stream, a);
    } else if (a.rel_h && NW == 8 && !(a.kW == KT && a.Nk % KT == 0) && sam_occ() >= 1) {
        hipLaunchKernelGGL((attn_kernel<D, QF, NW, DBUF, 1, false, false, NW == 8>), grid, block, 0, stream, a);
    } else if (a.rel_h && a.kW == KT && a.Nk % KT == 0) hipLaunchKernelGGL((attn_kernel<D, QF, NW, DBUF, 2, false>), grid, block, 0, stream, a);
    else if (a.rel_h) hipLaunchKernelGGL((attn_kernel<D, QF, NW, DBUF, 1, false>), grid, block, 0, stream, a);
    else if (a.key_mask) hipLaunchKernelGGL((attn_kernel<D, QF, NW, DBUF, 0, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((attn_kernel<D, QF, NW, DBUF, 0, false>), grid, block, 0, stream, a);
    return ae_check_launch("ae_attn_fwd_bf16");
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

}  // namespace

extern "C" int ae_attn_fwd_bf16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk, int D,
                                long q_sb, long q_sh, long q_sn, long k_sb, long k_sh, long k_sn,
                                long v_sb, long v_sh, long v_sn, long o_sb, long o_sh, long o_sn, float scale,
                                const float* rel_h, const float* rel_w, int kH, int kW, const unsigned char* key_mask,
                                const float* out_scale, int accumulate, const void* k2, const void* v2, int Nk2, long k2_sb,
                                long k2_sh, long k2_sn, long v2_sb, long v2_sh, long v2_sn, const float* scale2, float* lse,
                                float* lse2, void* stream) {
    AE_REQUIRE(q && k && v && out, "ae_attn_fwd_bf16: null pointer");
    AE_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "ae_attn_fwd_bf16: bad sizes B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    AE_REQUIRE((q_sb | q_sh | q_sn | k_sb | k_sh | k_sn | v_sb | v_sh | v_sn) % 8 == 0 && (o_sb | o_sh | o_sn) % 4 == 0,
               "ae_attn_fwd_bf16: strides must keep rows 16-byte aligned");
    AE_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)out & 7) == 0,
               "ae_attn_fwd_bf16: pointers must be 16-byte aligned");
    AE_REQUIRE((rel_h == nullptr) == (rel_w == nullptr), "ae_attn_fwd_bf16: rel_h and rel_w go together");
    if (rel_h) AE_REQUIRE(kH > 0 && kW > 0 && (long)kH * kW == Nk, "ae_attn_fwd_bf16: kH*kW must equal Nk");
    AttnArgs a{};
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)out;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk;
    a.q_sb = q_sb; a.q_sh = q_sh; a.q_sn = q_sn; a.k_sb = k_sb; a.k_sh = k_sh; a.k_sn = k_sn;
    a.v_sb = v_sb; a.v_sh = v_sh; a.v_sn = v_sn; a.o_sb = o_sb; a.o_sh = o_sh; a.o_sn = o_sn;
    a.scale = scale; a.rel_h = rel_h; a.rel_w = rel_w; a.kH = kH; a.kW = kW; a.key_mask = key_mask;
    a.out_scale = out_scale; a.accum = accumulate;
    if (k2) {
        AE_REQUIRE(v2 && scale2 && Nk2 > 0 && (D <= 96 || D == 160), "ae_attn_fwd_bf16: second segment needs v2, scale2, Nk2 > 0 and head_dim <= 96 or 160");
        AE_REQUIRE(!rel_h && !key_mask && !accumulate && !out_scale, "ae_attn_fwd_bf16: second segment excludes bias / mask / accumulate / out_scale");
        AE_REQUIRE((k2_sb | k2_sh | k2_sn | v2_sb | v2_sh | v2_sn) % 8 == 0 && ((uintptr_t)k2 & 15) == 0 && ((uintptr_t)v2 & 15) == 0,
                   "ae_attn_fwd_bf16: second segment alignment");
    }
    a.k2 = (const bf16_t*)k2; a.v2 = (const bf16_t*)v2; a.Nk2 = Nk2; a.k2_sb = k2_sb; a.k2_sh = k2_sh; a.k2_sn = k2_sn;
    a.v2_sb = v2_sb; a.v2_sh = v2_sh; a.v2_sn = v2_sn; a.scale2 = scale2;
    a.lse = lse; a.lse2 = lse2;
    if (lse2) AE_REQUIRE(k2 != nullptr, "ae_attn_fwd_bf16: lse2 needs a second segment");
    hipStream_t s = (hipStream_t)stream;
    // long-sequence / plain-softmax shapes (UNet self- and cross-attention at head_dim 40 / 80) go to the 32x32x16 kernel of
    // attention_fast.hip; everything it does not cover (bias, masks, log-sum-exp outputs, other head dims) stays here.
    static const int use_fast = env_int("AE_ATTN_FAST", 1);  // tuning knob: 0 = general kernel only (A/B)
    if (use_fast) {
        const int rc = ae_attn_fast_launch(a, D, s);
        if (rc != AE_ERR_UNSUPPORTED) return rc;
    }
    static const int qf40 = env_int("AE_ATTN_QF40", 4);  // tuning knob (A/B on hardware): query fragments per wave for D=40
    static const int w8 = env_int("AE_ATTN_W8", 0);      // tuning knob: 8 waves x 1 query fragment instead of 4 x 2
    // tuning knob: short K/V (cross-attention to 77 text + adapter tokens): 1 query fragment per wave -> half the registers, twice the
    // resident blocks for a kernel that is a chain of dependent load phases rather than MFMA work.  In-situ A/B (2 runs each): UNet
    // step 17.51 ms off / 17.57 ms on — neutral, stays off.
    static const int shortk = env_int("AE_ATTN_SHORTK_QF1", 0);
    const bool short_kv = shortk && Nk < 256 && (!k2 || Nk2 < 256) && !rel_h;
    switch (D) {
        case 8: return launch_attn<8, 2, true>(a, s);
        case 16: return launch_attn<16, 2, true>(a, s);
        case 32: return launch_attn<32, 2, true>(a, s);
        case 40:
            if (w8) return launch_attn<40, 1, true, 8>(a, s);
            if (short_kv) return launch_attn<40, 1, true>(a, s);
            // (2 fragments per wave under a 168-VGPR cap — 153 VGPRs, no spills, three resident blocks per CU — was A/B-ed against the
            // 4-fragment variant: 18.11 vs 18.04 ms per step; the K/V reuse of 4 fragments is worth more than the third block)
            // 64 queries per wave halve the K/V staging and K-fragment reads per query (514 vs 547 us at N = 4096); only for long
            // self-attention: the two-segment variant would spill at 4 fragments, and short K/V has nothing to amortise
            return (qf40 == 4 && Nq > 1024 && Nk >= 1024 && !k2) ? launch_attn<40, 4, true>(a, s) : launch_attn<40, 2, true>(a, s);
        case 48: return launch_attn<48, 2, true>(a, s);
        case 64: return launch_attn<64, 2, true>(a, s);
        // SAM (rel-pos bias): 8 waves x 1 query fragment keeps the bias registers + softmax state under 256 VGPRs without spills
        case 80:
            if (short_kv) return launch_attn<80, 1, true>(a, s);
            return (w8 || rel_h) ? launch_attn<80, 1, true, 8>(a, s) : launch_attn<80, 2, true>(a, s);
        case 96: return launch_attn<96, 2, true>(a, s);
        case 128: return launch_attn<128, 2, false>(a, s);
        // two segments at d = 160 (the 16x16-level cross-attention + expert tokens): one query fragment per wave keeps both output
        // accumulators in registers (256 + 170 of the wave's 512, no spills; two fragments spill 78)
        case 160: return (short_kv || k2) ? launch_attn<160, 1, false>(a, s) : launch_attn<160, 2, false>(a, s);
        default:
            ae_set_error("ae_attn_fwd_bf16: unsupported head_dim %d (supported: 8,16,32,40,48,64,80,96,128,160)", D);
            return AE_ERR_UNSUPPORTED;
    }
}
