// Row-panel GEMM for the short-K Linear layers of the 64x64 UNet level (gfx950 / MI355X):
//
//   C[M,N] = epi( LN?(A)[M,K] * W[N,K]^T + bias (+ residual) ),   K = 320 (the level-1 channel count), bf16 in / out, fp32 accumulate
//
// Replaces: the K = 320 nn.Linear / 1x1 conv layers of BasicTransformerBlock and SpatialTransformer
// (ldm/modules/attention.py:154-161 to_q / to_k / to_v / to_out, :49-76 GEGLU projection, :296-318 proj_in / proj_out) and, fused in
// front of them, the LayerNorm of attention.py:263-265, 271-275 (norm1 -> to_qkv, norm2 -> to_q, norm3 -> GEGLU projection).
//
// Why not the tiled kernel (gemm_conv.hip): with K = 320 a 128 x 128 tile has only five 64-deep K steps, each a dependent
// "DMA -> barrier -> MFMA" round trip — the launch runs at the memory LATENCY, 0.11-0.21 of the MFMA rate and 0.25 of its HBM
// roofline, and every activation row is re-read once per N tile.  Here the roles are turned around:
//   * a wave owns 48 rows of A for the whole launch: their MFMA operand fragments (3 x 10 x 4 = 120 VGPRs) are loaded ONCE,
//     straight from global memory; with LN the row statistics and the normalisation happen on those registers (a row's 320
//     values sit in 4 lanes: two cross-lane adds per statistic) — the LayerNorm launch and its write + read of the activation
//     disappear;
//   * W streams through a six-slot LDS ring in 32-row chunks (20 KiB each) by LDS-DMA with counted waits: five chunks are always
//     in flight, nobody ever waits for a load that was just issued, and one barrier per chunk is the only synchronisation;
//   * one block per CU (4 waves, one per SIMD, 192 rows): M = 49152 = 256 x 192 is exactly one block per CU;
//   * v_mfma_f32_16x16x32_bf16 with the operands swapped (W fragment = row operand): a W fragment read from LDS feeds three MFMAs,
//     and a lane ends up with 4 consecutive output columns of one row -> 8-byte stores, GEGLU's a / gate pair in the same lane.
#include "common.hpp"
#include <stdlib.h>

namespace {

enum { RP_EPI_NONE = 0, RP_EPI_GEGLU = 2 };

struct RowPanelArgs {
    const bf16_t* A; const bf16_t* W; bf16_t* C;
    const float* bias; const bf16_t* res;
    const float* ln_g; const float* ln_b; float ln_eps;
    int M, N;
    long lda, ldw, ldc, ldr;
    // LayerNorm FOLD (round 5; the form the tiled kernel has had since round 4, gemm_conv.hip `XE`):  LN(x) W^T + b = rstd (x W'^T - mu s) + c.
    float* rowstats;          // RS instantiation: [M][N / 64][2] fp32 (sum, sum of squares) of the STORED bf16 output per row and 64-column slice
    const float* ln_stats;    // LNM 2: [M][ln_parts][2], the row statistics of A as its producer emitted them
    const float* ln_colsum;   // LNM 2: [N] s[n] = sum_k W'[n][k] over the bf16 values of W' (packed row order); `bias` holds c
    int ln_parts;
};

constexpr int RP_MF = 3;     // 16-row fragments per wave
constexpr int RP_RG = 4;     // row groups (48 rows each) per block: one per SIMD
constexpr int RP_NW = 8;     // waves per block: row group rg = wave & 3 is shared by TWO waves (half = wave >> 2) that take the
                             // even / odd W chunks — while one half is in its MFMA phase the SIMD's other wave runs its epilogue,
                             // stores and DMA issue (one wave per SIMD measured 5000 cycles per chunk for 960 cycles of MFMA)
constexpr int RP_BN = 32;    // W rows (output columns) per chunk
constexpr int RP_PAIRS = 3;  // LDS ring: pairs of chunks
constexpr int RP_BM = RP_RG * RP_MF * 16;

template <int N>
__device__ __forceinline__ void rp_wait_dma() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Inside the chunk loop NO load may be visible to hipcc: its s_waitcnt bookkeeping knows nothing of the asm DMA pieces, so the
// wait it places for its own newest load is vmcnt(0) — which drains the whole ring every chunk (seen in the .s, 2.4 us per chunk
// instead of 0.7).  Bias therefore lives in LDS (copied once), and the residual rows are loaded by these asm loads with a wait of
// our own.
constexpr int RP_MAXN = 2560;  // bias floats kept in LDS
// The residual rows are ordinary loads issued AFTER the chunk's DMA refill: hipcc's own vmcnt(0) in front of their first use (the
// epilogue) then also covers the refill, which by that time has had the whole MFMA phase to land (no time difference measured).
// Two other forms were tried and are wrong: asm loads with a counted vmcnt(pieces issued later) — LDS-DMA pieces and ordinary
// loads do NOT retire in issue order relative to each other on gfx950, the wait returned before the residual had landed; and asm
// loads with an asm vmcnt(0) — under register pressure hipcc copies the destination registers before the wait.

#ifdef AE_RP_LAB
__device__ unsigned long long g_rp_dbg[8];  // lab only: cycle buckets of block 0 / wave 0
#define RP_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && tid == 0) g_rp_dbg[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define RP_T(i)
#endif

// Column order inside a chunk.  The MFMA leaves lane group g with the chunk's W rows 4 g .. 4 g + 3 (fragment 0) and
// 16 + 4 g .. 16 + 4 g + 3 (fragment 1).  EPI_NONE: image row i of the chunk therefore holds W row n0 + 8 (i' >> 2) + 4 (i >> 4) +
// (i & 3) (i' = i & 15): a lane's two fragments are then EIGHT consecutive output columns 8 g .. 8 g + 7 -> one 16-byte store per
// row instead of two 8-byte ones.  GEGLU keeps the packed order (a rows then gate rows, ops.pack_geglu): a and gate share a lane.
template <int EPI>
__device__ __forceinline__ int rp_wrow(int i) {
    return EPI == RP_EPI_GEGLU ? i : 8 * ((i & 15) >> 2) + 4 * (i >> 4) + (i & 3);
}

// LNM: 0 plain, 1 LayerNorm on the A registers in the prologue (rounds 2-4), 2 LayerNorm FOLDED into the epilogue from the producer's row statistics
// (round 5: the prologue form costs every wave ~2300 VALU instructions — statistics in two passes and the re-packing of its 240 operand values — and
// the two waves of a row group both do it for the same 48 rows: ~8 us of a 32-54 us launch; the fold multiplies the RAW rows and spends two FMAs per
// output instead).  RS: the epilogue also emits the row statistics of its output for the NEXT LayerNorm (EPI_NONE only).
template <int KS, int EPI, int LNM, bool RS = false>
__global__ __launch_bounds__(64 * RP_NW, 2) void gemm_rowpanel_kernel(const RowPanelArgs p) {
    constexpr bool LN = LNM == 1, FOLD = LNM == 2;
    static_assert(!RS || (EPI == RP_EPI_NONE && LNM == 0), "row statistics: plain GEMM (+bias, +residual) only");
    constexpr int K = 32 * KS, ROWB = 2 * K, CHUNKB = RP_BN * ROWB, PIECES = CHUNKB / 1024;
    constexpr int PPW = 2 * PIECES / RP_NW;        // DMA pieces per wave per chunk PAIR
    constexpr int APW = 6 * PIECES / RP_NW;        // ... for the A panel (six chunks)
    static_assert((2 * PIECES) % RP_NW == 0 && (6 * PIECES) % RP_NW == 0 && KS % 2 == 0, "pieces must split evenly over the waves; K % 64 == 0");
    static_assert(RP_BM == 6 * RP_BN && RP_PAIRS == 3, "the A panel is staged through the six chunk slots of the ring");
    __shared__ __attribute__((aligned(16))) char smem[2 * RP_PAIRS * CHUNKB + RP_MAXN * 4 * (FOLD ? 2 : 1)];
    float* const sbias = reinterpret_cast<float*>(smem + 2 * RP_PAIRS * CHUNKB);
    float* const ssum = sbias + (FOLD ? RP_MAXN : 0);                     // FOLD: s[n] behind c[n]
    __shared__ __attribute__((aligned(16))) float sln[LN ? 2 * K : 4];  // gamma | beta
    // RS: (sum, sum of squares) of a wave's 32-column chunk per row, parked for the pair's other half: [pair parity][half][row group][fragment][row]
    __shared__ __attribute__((aligned(8))) f32x2 sred[RS ? 2 : 1][2][RP_RG][RP_MF][16];

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef AE_RP_LAB
    unsigned long long tlast = __builtin_readcyclecounter();
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 3, half = wave >> 2;
    const int l15 = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * RP_BM + rg * (RP_MF * 16);
    const int lds0 = (int)(uintptr_t)((__attribute__((address_space(3))) char*)smem);

    // ---- W ring: a chunk is W rows [32 c, 32 c + 32) (in rp_wrow order) as an image [32][K] with the 16-byte pieces of a row
    // XOR-swizzled inside groups of eight (piece position p = c16 ^ ((i >> 1) & 7)): a ds_read_b128 lane group then touches 16
    // different 4-bank slots.  Piece q of a chunk PAIR (q = wave + 8 j) belongs to chunk q / PIECES.
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), 0, (int)((long)p.N * p.ldw * 2), 0x00020000);
    int dma_off[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int q = wave + RP_NW * j;
        const int o = (q % PIECES) * 1024 + lane * 16;  // byte position inside the chunk image
        const int i = o / ROWB, pp = (o - i * ROWB) >> 4;
        dma_off[j] = ((q / PIECES) * RP_BN + rp_wrow<EPI>(i)) * (int)p.ldw * 2 + ((pp & ~7) | ((pp ^ (i >> 1)) & 7)) * 16;
    }
    const int npairs = p.N / (2 * RP_BN);
    const int pair_stride = 2 * RP_BN * (int)p.ldw * 2;
    auto issue = [&](int pi) {
        const int slot = lds0 + (pi % RP_PAIRS) * 2 * CHUNKB;
        const int soff = pi * pair_stride;
#pragma unroll
        for (int j = 0; j < PPW; ++j) ae_dma16(rsW, slot + (wave + RP_NW * j) * 1024, dma_off[j], soff);
    };

    // ---- A panel: the block's 192 rows are six 32-row chunks = the six chunk slots of the ring.  They are copied by the same
    // LDS-DMA pieces (whole 1-KiB runs of consecutive rows: fully coalesced; rows >= M read as zeros through the descriptor's bounds
    // check) and every wave then picks its MFMA operand fragments out of LDS (B operand: lane (m = l15, g) holds
    // A[row][32 ks + 8 g .. +8]).
    {
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, (int)((long)p.M * p.lda * 2), 0x00020000);
        const int soff = blockIdx.x * RP_BM * (int)p.lda * 2;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int q = wave + RP_NW * j;  // piece of the whole panel image
            const int o = (q % PIECES) * 1024 + lane * 16;
            const int i = o / ROWB, pp = (o - i * ROWB) >> 4;
            ae_dma16(rsA, lds0 + q * 1024, ((q / PIECES) * RP_BN + i) * (int)p.lda * 2 + ((pp & ~7) | ((pp ^ (i >> 1)) & 7)) * 16, soff);
        }
    }
    // (GEGLU: the packed columns interleave 16 'a' rows with their 16 gate rows; the bias of an 'a' column is kept halved: geglu_half_f)
    for (int i = tid; i < p.N; i += 64 * RP_NW) sbias[i] = p.bias ? p.bias[i] * ((EPI == RP_EPI_GEGLU && (i & 16) == 0) ? 0.5f : 1.0f) : 0.f;
    if (FOLD) {
        for (int i = tid; i < p.N; i += 64 * RP_NW) ssum[i] = p.ln_colsum[i];
    }
    if (LN) {
        for (int i = tid; i < K; i += 64 * RP_NW) { sln[i] = p.ln_g[i]; sln[K + i] = p.ln_b[i]; }
    }
    rp_wait_dma<0>();
    __syncthreads();
    u32x4 af[RP_MF][KS];
#pragma unroll
    for (int f = 0; f < RP_MF; ++f) {
        const int rl = rg * (RP_MF * 16) + 16 * f + l15;  // row inside the panel
        const int i = rl & 31;
        const char* base = smem + (rl >> 5) * CHUNKB + i * ROWB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c16 = 4 * ks + g;
            af[f][ks] = *reinterpret_cast<const u32x4*>(base + (((c16 & ~7) | ((c16 ^ (i >> 1)) & 7)) << 4));
        }
    }
    __syncthreads();  // every wave holds its fragments: the ring now belongs to W
    for (int pi = 0; pi < RP_PAIRS - 1 && pi < npairs; ++pi) issue(pi);

    if (LN) {  // LayerNorm on the registers (fp32 statistics, biased variance, eps inside the root: F.layer_norm)
        float mean[RP_MF], rstd[RP_MF];
#pragma unroll
        for (int f = 0; f < RP_MF; ++f) {
            float s = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 t = af[f][ks];
                s += (bf16lo(t.x) + bf16hi(t.x)) + (bf16lo(t.y) + bf16hi(t.y)) + (bf16lo(t.z) + bf16hi(t.z)) + (bf16lo(t.w) + bf16hi(t.w));
            }
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            const float mu = s * (1.0f / K);
            // hipcc otherwise keeps the 80 unpacked floats of pass 1 alive for pass 2 and for the normalisation (common
            // subexpressions) — 240 temporaries per wave, 300 spilled registers.  Re-unpacking costs one VALU per value.
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[f][ks]));
            float v = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 t = af[f][ks];
                const float d0 = bf16lo(t.x) - mu, d1 = bf16hi(t.x) - mu, d2 = bf16lo(t.y) - mu, d3 = bf16hi(t.y) - mu;
                const float d4 = bf16lo(t.z) - mu, d5 = bf16hi(t.z) - mu, d6 = bf16lo(t.w) - mu, d7 = bf16hi(t.w) - mu;
                v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) + (d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7);
            }
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            mean[f] = mu;
            rstd[f] = __builtin_amdgcn_rsqf(v * (1.0f / K) + p.ln_eps);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(af[f][ks]));
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(sln + 32 * ks + 8 * g), g1 = *reinterpret_cast<const f32x4*>(sln + 32 * ks + 8 * g + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sln + K + 32 * ks + 8 * g), b1 = *reinterpret_cast<const f32x4*>(sln + K + 32 * ks + 8 * g + 4);
#pragma unroll
            for (int f = 0; f < RP_MF; ++f) {
                const u32x4 t = af[f][ks];
                const float mu = mean[f], rs = rstd[f];
                u32x4 w;
                w.x = pack_bf16x2((bf16lo(t.x) - mu) * rs * g0[0] + b0[0], (bf16hi(t.x) - mu) * rs * g0[1] + b0[1]);
                w.y = pack_bf16x2((bf16lo(t.y) - mu) * rs * g0[2] + b0[2], (bf16hi(t.y) - mu) * rs * g0[3] + b0[3]);
                w.z = pack_bf16x2((bf16lo(t.z) - mu) * rs * g1[0] + b1[0], (bf16hi(t.z) - mu) * rs * g1[1] + b1[1]);
                w.w = pack_bf16x2((bf16lo(t.w) - mu) * rs * g1[2] + b1[2], (bf16hi(t.w) - mu) * rs * g1[3] + b1[3]);
                af[f][ks] = w;
            }
            __builtin_amdgcn_sched_barrier(0);  // one K step's gamma / beta at a time (hoisting all ten spilled 160 registers)
        }
    }

    // FOLD: rstd and -rstd * mean of this lane's row in each fragment, from the producer's per-slice sums (lane group g takes slices g, g + 4, ...;
    // two cross-lane adds: (s0 + s1) + (s2 + s3) in every lane, a fixed order).  Ordinary loads, issued and consumed in front of the chunk loop.
    float ln_r[FOLD ? RP_MF : 1], ln_t[FOLD ? RP_MF : 1];
    if (FOLD) {
#pragma unroll
        for (int f = 0; f < RP_MF; ++f) {
            const int row = min(m0 + 16 * f + l15, p.M - 1);
            const f32x2* sp = reinterpret_cast<const f32x2*>(p.ln_stats) + (long)row * p.ln_parts;
            float sm = 0.f, sq = 0.f;
            for (int u = g; u < p.ln_parts; u += 4) { const f32x2 v = sp[u]; sm += v[0]; sq += v[1]; }
            sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);
            sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
            const float mu = sm * (1.0f / K);
            const float r = __builtin_amdgcn_rsqf(fmaxf(sq * (1.0f / K) - mu * mu, 0.f) + p.ln_eps);
            ln_r[f] = r;
            ln_t[f] = -r * mu;
        }
    }

    // W fragment (A operand: lane (i = l15 (+16), g) holds image row i, k = 32 ks + 8 g .. +8) addresses inside a chunk slot
    int woff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int c16 = 4 * ks + g;
        woff[ks] = l15 * ROWB + (((c16 & ~7) | ((c16 ^ (l15 >> 1)) & 7)) << 4);
    }
    // per-lane element offsets of this lane's rows in C / residual (its columns of a chunk start at 8 g, or 4 g for GEGLU)
    constexpr int CPL = EPI == RP_EPI_GEGLU ? 4 : 8;
    int coff[RP_MF], roff[RP_MF];
    bool rok[RP_MF];
#pragma unroll
    for (int f = 0; f < RP_MF; ++f) {
        const int row = m0 + 16 * f + l15;
        rok[f] = row < p.M;
        coff[f] = min(row, p.M - 1) * (int)p.ldc + CPL * g;
        roff[f] = min(row, p.M - 1) * (int)p.ldr + CPL * g;
    }

    RP_T(0);  // prologue
    for (int pi = 0; pi < npairs; ++pi) {
        // pair pi has landed when at most the pieces of the pairs issued after it are outstanding (DMA pieces retire in order
        // among themselves; the wave's stores in between can only make the wait stricter)
        if (pi + 1 < npairs) rp_wait_dma<PPW>();
        else rp_wait_dma<0>();
        RP_T(1);  // DMA wait
        __builtin_amdgcn_s_barrier();  // every wave's pieces of pair pi are in LDS; every wave is done reading pair pi - 1
        asm volatile("" ::: "memory");
        RP_T(2);  // barrier
        const int c = 2 * pi + half;
        const int n0 = c * RP_BN;
        if (pi + RP_PAIRS - 1 < npairs) issue(pi + RP_PAIRS - 1);  // refills the slot of pair pi - 1
        if (RS && pi > 0 && half == 0 && g == 0) {   // pair pi - 1 = output columns [64 (pi - 1), 64 pi): both halves parked their 32-column sums before this pair's barrier
#pragma unroll
            for (int f = 0; f < RP_MF; ++f) {
                const f32x2 a0 = sred[(pi - 1) & 1][0][rg][f][l15], a1 = sred[(pi - 1) & 1][1][rg][f][l15];
                if (rok[f]) *reinterpret_cast<f32x2*>(p.rowstats + ((long)(m0 + 16 * f + l15) * (p.N >> 6) + (pi - 1)) * 2) = (f32x2){a0[0] + a1[0], a0[1] + a1[1]};
            }
        }
        u32x4 rres[RP_MF];
        if (EPI == RP_EPI_NONE && p.res) {  // residual rows of this chunk, in flight under the MFMAs
#pragma unroll
            for (int f = 0; f < RP_MF; ++f) rres[f] = *reinterpret_cast<const u32x4*>((p.res + n0) + roff[f]);
            asm volatile("" ::: "memory");  // the loads are issued here, not sunk to their use
        }

        const char* slot = smem + ((pi % RP_PAIRS) * 2 + half) * CHUNKB;
        f32x4 acc[RP_MF][2];
#pragma unroll
        for (int f = 0; f < RP_MF; ++f) acc[f][0] = acc[f][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int nf = 0; nf < 2; ++nf) {
                const bf16x8_t wf = as_bf16x8(*reinterpret_cast<const u32x4*>(slot + woff[ks] + nf * 16 * ROWB));
#pragma unroll
                for (int f = 0; f < RP_MF; ++f) acc[f][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, as_bf16x8(af[f][ks]), acc[f][nf], 0, 0, 0);
            }
        }
#ifdef AE_RP_LAB
        asm volatile("s_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[0][1]), "+v"(acc[1][1]), "+v"(acc[2][1]));
#endif
        RP_T(3);  // issue + MFMA

        // ---- epilogue of the chunk: lane holds D[i = 16 nf + 4 g + r][m = l15]
        if (EPI == RP_EPI_GEGLU) {  // W rows interleaved 16 a / 16 gate (ops.pack_geglu): a in nf = 0, its gate in nf = 1
            const f32x4 ba = *reinterpret_cast<const f32x4*>(sbias + n0 + 4 * g);
            const f32x4 bg = *reinterpret_cast<const f32x4*>(sbias + n0 + 16 + 4 * g);
            bf16_t* const cb = p.C + (n0 >> 1);
#pragma unroll
            for (int f = 0; f < RP_MF; ++f) {
                if (rok[f]) {
                    float o[4];
                    if (FOLD) {   // a half = 0.5 (rstd (acc - mu s) + c), gate = rstd (acc - mu s) + c; ba holds 0.5 c, the s values are un-halved
                        const f32x4 sa = *reinterpret_cast<const f32x4*>(ssum + n0 + 4 * g), sg = *reinterpret_cast<const f32x4*>(ssum + n0 + 16 + 4 * g);
                        const float rh = 0.5f * ln_r[f], th = 0.5f * ln_t[f];
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            o[r4] = geglu_half_f(fmaf(acc[f][0][r4], rh, fmaf(th, sa[r4], ba[r4])), fmaf(acc[f][1][r4], ln_r[f], fmaf(ln_t[f], sg[r4], bg[r4])));
                    } else {
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) o[r4] = geglu_half_f(fmaf(acc[f][0][r4], 0.5f, ba[r4]), acc[f][1][r4] + bg[r4]);   // ba holds 0.5 bias
                    }
                    *reinterpret_cast<u32x2*>(cb + coff[f]) = (u32x2){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                }
            }
        } else {  // columns n0 + 8 g .. + 7: fragment 0 holds the first four, fragment 1 the next four (rp_wrow)
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sbias + n0 + 8 * g), b1 = *reinterpret_cast<const f32x4*>(sbias + n0 + 8 * g + 4);
            bf16_t* const cb = p.C + n0;
#pragma unroll
            for (int f = 0; f < RP_MF; ++f) {
                float o0, o1, o2, o3, o4, o5, o6, o7;
                if (FOLD) {   // rstd (acc - mu s) + c
                    const f32x4 s0 = *reinterpret_cast<const f32x4*>(ssum + n0 + 8 * g), s1 = *reinterpret_cast<const f32x4*>(ssum + n0 + 8 * g + 4);
                    const float r = ln_r[f], t = ln_t[f];
                    o0 = fmaf(acc[f][0][0], r, fmaf(t, s0[0], b0[0])); o1 = fmaf(acc[f][0][1], r, fmaf(t, s0[1], b0[1]));
                    o2 = fmaf(acc[f][0][2], r, fmaf(t, s0[2], b0[2])); o3 = fmaf(acc[f][0][3], r, fmaf(t, s0[3], b0[3]));
                    o4 = fmaf(acc[f][1][0], r, fmaf(t, s1[0], b1[0])); o5 = fmaf(acc[f][1][1], r, fmaf(t, s1[1], b1[1]));
                    o6 = fmaf(acc[f][1][2], r, fmaf(t, s1[2], b1[2])); o7 = fmaf(acc[f][1][3], r, fmaf(t, s1[3], b1[3]));
                } else {
                    o0 = acc[f][0][0] + b0[0]; o1 = acc[f][0][1] + b0[1]; o2 = acc[f][0][2] + b0[2]; o3 = acc[f][0][3] + b0[3];
                    o4 = acc[f][1][0] + b1[0]; o5 = acc[f][1][1] + b1[1]; o6 = acc[f][1][2] + b1[2]; o7 = acc[f][1][3] + b1[3];
                }
                if (p.res) {
                    const u32x4 r = rres[f];
                    o0 += bf16lo(r.x); o1 += bf16hi(r.x); o2 += bf16lo(r.y); o3 += bf16hi(r.y);
                    o4 += bf16lo(r.z); o5 += bf16hi(r.z); o6 += bf16lo(r.w); o7 += bf16hi(r.w);
                }
                const u32x4 pk = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3), pack_bf16x2(o4, o5), pack_bf16x2(o6, o7)};
                if (rok[f]) *reinterpret_cast<u32x4*>(cb + coff[f]) = pk;
                if (RS) {   // statistics of the STORED values: this lane's 8 columns, then the row's four lane groups (every lane of the wave takes part)
                    const uint32_t w4[4] = {pk.x, pk.y, pk.z, pk.w};
                    float ps = 0.f, pq = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = bf16lo(w4[e]), c2 = bf16hi(w4[e]);
                        ps += a + c2;
                        pq += a * a + c2 * c2;
                    }
                    ps += __shfl_xor(ps, 16, 64); pq += __shfl_xor(pq, 16, 64);
                    ps += __shfl_xor(ps, 32, 64); pq += __shfl_xor(pq, 32, 64);
                    if (g == 0) sred[pi & 1][half][rg][f][l15] = (f32x2){ps, pq};
                }
            }
        }
        RP_T(4);  // epilogue
    }
    if (RS) {   // the last pair's statistics: every wave has parked its sums once the block passes this barrier
        __syncthreads();
        if (half == 0 && g == 0) {
#pragma unroll
            for (int f = 0; f < RP_MF; ++f) {
                const f32x2 a0 = sred[(npairs - 1) & 1][0][rg][f][l15], a1 = sred[(npairs - 1) & 1][1][rg][f][l15];
                if (rok[f]) *reinterpret_cast<f32x2*>(p.rowstats + ((long)(m0 + 16 * f + l15) * (p.N >> 6) + (npairs - 1)) * 2) = (f32x2){a0[0] + a1[0], a0[1] + a1[1]};
            }
        }
    }
}

template <int KS>
int launch_rowpanel(const RowPanelArgs& a, int epi, hipStream_t stream) {
    const unsigned grid = (unsigned)((a.M + RP_BM - 1) / RP_BM);
    const bool ln = a.ln_g != nullptr, fold = a.ln_stats != nullptr;
    if (epi == RP_EPI_GEGLU) {
        if (fold) hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, RP_EPI_GEGLU, 2>), dim3(grid), dim3(64 * RP_NW), 0, stream, a);
        else if (ln) hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, RP_EPI_GEGLU, 1>), dim3(grid), dim3(64 * RP_NW), 0, stream, a);
        else hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, RP_EPI_GEGLU, 0>), dim3(grid), dim3(64 * RP_NW), 0, stream, a);
    } else {
        if (fold) hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, RP_EPI_NONE, 2>), dim3(grid), dim3(64 * RP_NW), 0, stream, a);
        else if (a.rowstats) hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, RP_EPI_NONE, 0, true>), dim3(grid), dim3(64 * RP_NW), 0, stream, a);
        else if (ln) hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, RP_EPI_NONE, 1>), dim3(grid), dim3(64 * RP_NW), 0, stream, a);
        else hipLaunchKernelGGL((gemm_rowpanel_kernel<KS, RP_EPI_NONE, 0>), dim3(grid), dim3(64 * RP_NW), 0, stream, a);
    }
    return ae_check_launch("ae_ln_gemm_bf16");
}

}  // namespace

// 1 when the row-panel kernel covers the shape (the caller otherwise uses ae_layernorm_bf16 + ae_gemm_bf16)
extern "C" int ae_ln_gemm_supported(int M, int N, int K, int epilogue) {
    // one 192-row block per CU: below ~3/4 of the 256 CUs the tiled kernel (more, smaller blocks) fills the chip better.  The test hook
    // AE_ROWPANEL_ANY_M lifts the limit so that small shapes exercise the kernel.
    static const int any_m = getenv("AE_ROWPANEL_ANY_M") ? atoi(getenv("AE_ROWPANEL_ANY_M")) : 0;
    if (!any_m && (M + RP_BM - 1) / RP_BM < 192) return 0;
    return (K == 320 && N % (2 * RP_BN) == 0 && N <= RP_MAXN && M >= RP_BM && (epilogue == RP_EPI_NONE || epilogue == RP_EPI_GEGLU)) ? 1 : 0;
}

// LayerNorm fold on the row-panel kernel (round 5): the K = 320 shapes of ae_gemm_ln_bf16 (gemm_conv.hip forwards them here).  Returns AE_ERR_UNSUPPORTED
// when the shape / alignment is outside this kernel's envelope (the caller then takes the tiled kernel's plan).  AE_RP_FOLD=0 turns it off (A/B).
int ae_rowpanel_fold_covers(int M, int N, int K, int epilogue, int mode) {
    static const int on = getenv("AE_RP_FOLD") ? atoi(getenv("AE_RP_FOLD")) : 1;
    if (!on || !ae_ln_gemm_supported(M, N, K, epilogue)) return 0;
    // the 32-bit-offset envelope of ae_rowpanel_fold_launch, for contiguous rows of max(K, stored N) elements: a plan that says yes must not be refused at
    // launch (UNet batches >= ~200 samples at 64x64 then got a hard error instead of the tiled plan)
    const long ldmax = K > N ? K : N;
    if (((long)M + RP_BM) * ldmax * 2 >= (1L << 31) || (long)N * K * 2 >= (1L << 31)) return 0;
    return mode == 2 || (mode == 1 && epilogue == RP_EPI_NONE);
}
int ae_rowpanel_fold_launch(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias, const void* residual,
                            long ldr, int epilogue, float* rowstats_out, const float* ln_stats, int ln_parts, const float* ln_colsum, float ln_eps, void* stream) {
    if (!ae_rowpanel_fold_covers(M, N, K, epilogue, rowstats_out ? 1 : 2)) return AE_ERR_UNSUPPORTED;
    if (lda % 8 || ldw % 8 || ldc % 8 || (residual && ldr % 8) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 15) || ((uintptr_t)residual & 15)) return AE_ERR_UNSUPPORTED;
    const long ldmax = lda > ldc ? (lda > ldr ? lda : ldr) : (ldc > ldr ? ldc : ldr);
    if ((long)N * ldw * 2 >= (1L << 31) || ((long)M + RP_BM) * ldmax * 2 >= (1L << 31)) return AE_ERR_UNSUPPORTED;
    if (ln_stats) {
        AE_REQUIRE(ln_colsum && bias && ln_parts > 0 && ln_parts * 64 == K && ln_eps >= 0.f, "ae_gemm_ln_bf16 (row panel): LayerNorm fold needs s, c and K / 64 = %d statistics slices per row (got %d)", K / 64, ln_parts);
        AE_REQUIRE(!(epilogue == RP_EPI_GEGLU && residual), "ae_gemm_ln_bf16 (row panel): GEGLU has no residual");
    } else {
        AE_REQUIRE(epilogue == RP_EPI_NONE && ((uintptr_t)rowstats_out & 7) == 0, "ae_gemm_ln_bf16 (row panel): row statistics go with a plain GEMM (+bias, +residual)");
    }
    RowPanelArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias; a.res = (const bf16_t*)residual;
    a.ln_g = nullptr; a.ln_b = nullptr; a.ln_eps = ln_eps; a.M = M; a.N = N; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    a.rowstats = rowstats_out; a.ln_stats = ln_stats; a.ln_colsum = ln_colsum; a.ln_parts = ln_parts;
    return launch_rowpanel<10>(a, epilogue, (hipStream_t)stream);
}

extern "C" int ae_ln_gemm_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int N, int K, const float* bias,
                               const void* residual, long ldr, const float* ln_gamma, const float* ln_beta, float ln_eps, int epilogue,
                               float* colstats, void* stream) {
    AE_REQUIRE(A && W && C, "ae_ln_gemm_bf16: null pointer");
    AE_REQUIRE(ae_ln_gemm_supported(M, N, K, epilogue), "ae_ln_gemm_bf16: unsupported shape M=%d N=%d K=%d epilogue=%d (K must be 320, N %% 64 == 0)", M, N, K, epilogue);
    AE_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr), "ae_ln_gemm_bf16: gamma and beta go together");
    AE_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0), "ae_ln_gemm_bf16: row strides must keep 16-byte alignment");
    AE_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)residual & 15) == 0, "ae_ln_gemm_bf16: pointer alignment");
    AE_REQUIRE((long)N * ldw * 2 < (1L << 31), "ae_ln_gemm_bf16: W too large for 32-bit offsets");
    {   // A, C and the residual are addressed with 32-bit byte offsets too (descriptor extent, per-block base, row * ld): refuse instead of wrapping
        const long ldmax = lda > ldc ? (lda > ldr ? lda : ldr) : (ldc > ldr ? ldc : ldr);
        AE_REQUIRE(((long)M + RP_BM) * ldmax * 2 < (1L << 31), "ae_ln_gemm_bf16: M * row stride reaches 2 GiB (32-bit offsets): use ae_gemm_bf16 for this shape");
    }
    AE_REQUIRE(!(epilogue == RP_EPI_GEGLU && residual), "ae_ln_gemm_bf16: GEGLU has no residual");
    RowPanelArgs a{};
    a.A = (const bf16_t*)A; a.W = (const bf16_t*)W; a.C = (bf16_t*)C; a.bias = bias; a.res = (const bf16_t*)residual;
    a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_eps = ln_eps; a.M = M; a.N = N; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.ldr = ldr;
    const int rc = launch_rowpanel<10>(a, epilogue, (hipStream_t)stream);
    if (rc || !colstats) return rc;
    // per-channel statistics of the output for the GroupNorm that consumes it (proj_out of the 64x64 level): this kernel's epilogue keeps
    // a lane on ONE row across all columns, so the column sums come from the stand-alone pass over the (L2-resident) output
    AE_REQUIRE(((uintptr_t)colstats & 15) == 0, "ae_ln_gemm_bf16: colstats alignment");
    return ae_launch_colstats((const bf16_t*)C, ldc, M, epilogue == RP_EPI_GEGLU ? N / 2 : N, colstats, (hipStream_t)stream);
}
