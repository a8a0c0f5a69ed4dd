// GroupNorm(+SiLU) and LayerNorm on channels-last bf16 activations for gfx950 — HBM-bandwidth kernels.
//
// Replaces: GroupNorm32.forward (ldm/modules/diffusionmodules/util.py:217-219, eps 1e-5),
// Normalize (ldm/modules/attention.py:88-89, eps 1e-6) with the nn.SiLU that follows them in ResBlock / UNet.out
// (openaimodel.py:200-204, 224-231, 726-730); nn.LayerNorm in BasicTransformerBlock (attention.py:263-265) and in the
// SAM Block (image_encoder.py:166-182, eps 1e-6).  Statistics are fp32 exactly as the reference forces them.
//
// GroupNorm is three launches: (1) per-(batch, row-chunk) partial sums for all groups, deterministic (no float atomics
// to global memory); (2) finalize: one block per batch element folds the partials into per-channel scale/shift;
// (3) apply: pure streaming, every thread keeps its 8 scales/shifts in registers and all of its rows in flight
// (16-byte loads/stores), SiLU fused.  The input may be the channel-concat
// of two tensors (decoder skip connections, openaimodel.py:780) — read in place, never materialised.
#include "common.hpp"
#include <stdlib.h>

#ifndef GN_CS_U
#define GN_CS_U 4   // independent loads per thread of gn_finalize_cs_kernel.  Round 6: 8 (one round trip for the 1 280 / 1 920 slab sums of a group) measured
                    // 12.479 / 12.458 / 12.444 vs 12.457 / 12.443 / 12.436 ms per UNet step for 4 in three alternating pairs — nothing; 4 stays (-DGN_CS_U=8 builds it)
#endif

namespace {

constexpr int GN_MAXR = 8;  // rows per thread kept in flight

struct GNArgs {
    const bf16_t* x; const bf16_t* x2;  // x: channels [0, C1), x2: channels [C1, C)
    int C1;
    const float* gamma; const float* beta;
    bf16_t* y;
    float* part;   // [B][nchunk][groups][2]  partial (sum, sumsq)
    float* coef;   // [B][2][C]               per-channel scale / shift
    int B, HW, C, groups, rows_per_chunk, nchunk, act;
    float eps;
    float* stat;   // optional [B][groups][2] (mean, rstd): written by finalize for the backward pass
    // backward (ae_groupnorm_bwd_nhwc_bf16)
    const bf16_t* dy; bf16_t* dx; bf16_t* dx2;
    float* part2;  // [B][nchunk][groups][2]  partial (sum dz*gamma, sum dz*(z - beta))
    float* coef2;  // [B][2][C]               per-channel kA, kB of dx = dz*scale + x*kA + kB
    int* counters; // optional [B], zero on entry and on exit: the LAST partial-sum block of a sample runs the finalize fold itself
    const float* cs1; const float* cs2;  // optional per-channel (sum, sumsq) over 32-row slabs of x / x2, written by their PRODUCERS
                                         // ([B*HW/32][C1][2], [B*HW/32][C-C1][2]): replaces the statistics pass over the activation
    int accum;     // backward: bit 0 dx += (the tensor already holds a gradient from another consumer of x), bit 1 dx2 +=
    // ae_groupnorm_splitk_nhwc_bf16: x is not in memory yet — it is the fold of a split-K conv's fp32 partials [sk_n][B*HW][C] (+ bias + the per-sample vector)
    const float* sk_partial; const float* sk_bias; const float* sk_addvec;
    int sk_n; long sk_ldav, sk_MN;
};

// Fold of the per-chunk partials of sample b by the calling block (any block size): 16 slices x 64 group lanes, slice i takes chunks
// i, i+16, ..; the 16 slice sums are then added in slice order.  The stand-alone finalize kernels (1024 threads: one slot each) and
// the last-block tails of the partial-sum kernels run exactly this code, so both routes give bit-identical statistics.
__device__ __forceinline__ void gn_fold(const float* part, int b, int nchunk, int groups, float (*red)[16][64]) {
    for (int slot = threadIdx.x; slot < 1024; slot += blockDim.x) {
        const int g = slot & 63, sl = slot >> 6;
        float a = 0.f, c = 0.f;
        if (g < groups) {
            for (int i = sl; i < nchunk; i += 16) {
                const float* src = part + (((long)b * nchunk + i) * groups + g) * 2;
                a += src[0];
                c += src[1];
            }
        }
        red[0][sl][g] = a;
        red[1][sl][g] = c;
    }
    __syncthreads();
}

// per-channel scale = gamma * rstd, shift = beta - mean * scale of sample b (and (mean, rstd) for the backward pass)
__device__ __forceinline__ void gn_finalize_body(const GNArgs& p, int b, float (*red)[16][64], float* mean, float* rstd) {
    gn_fold(p.part, b, p.nchunk, p.groups, red);
    if (threadIdx.x < p.groups) {
        const int g = threadIdx.x;
        float sa = 0.f, sc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { sa += red[0][j][g]; sc += red[1][j][g]; }
        const float n = (float)(p.C / p.groups) * (float)p.HW;
        const float mu = sa / n;
        const float var = fmaxf(sc / n - mu * mu, 0.f);
        mean[g] = mu;
        rstd[g] = rsqrtf(var + p.eps);
        if (p.stat) {
            p.stat[((long)b * p.groups + g) * 2 + 0] = mu;
            p.stat[((long)b * p.groups + g) * 2 + 1] = rstd[g];
        }
    }
    __syncthreads();
    const int cpg = p.C / p.groups;
    for (int ch = threadIdx.x; ch < p.C; ch += blockDim.x) {
        const int gi = ch / cpg;
        const float sc = p.gamma[ch] * rstd[gi];
        p.coef[((long)b * 2 + 0) * p.C + ch] = sc;
        p.coef[((long)b * 2 + 1) * p.C + ch] = p.beta[ch] - mean[gi] * sc;
    }
}

// true in every thread of the block that finished LAST among the gridDim.x blocks of sample b (device-scope release / acquire around
// the ticket, so that block sees every other block's partial sums)
__device__ __forceinline__ bool gn_last_block(int* counters, int b, int* s_flag) {
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) *s_flag = atomicAdd(counters + b, 1) == (int)gridDim.x - 1;
    __syncthreads();
    const bool last = *s_flag != 0;
    if (last) __threadfence();
    return last;
}

__device__ __forceinline__ u32x4 gn_load(const GNArgs& p, long row, int cc) {
    const int ch = cc * 8;
    if (ch < p.C1) return *reinterpret_cast<const u32x4*>(p.x + row * p.C1 + ch);
    return *reinterpret_cast<const u32x4*>(p.x2 + row * (p.C - p.C1) + (ch - p.C1));
}

// The 8-channel piece (row, cc) of x = bf16(sum_s partial[s] + bias + addvec[b]): splitk_reduce_kernel's arithmetic statement for statement (K ranges added in
// order, then the bias, then the vector, ONE rounding), so a GroupNorm fed this way returns what it returns behind the reduce launch, bit for bit.
__device__ __forceinline__ u32x4 gn_load_splitk(const GNArgs& p, long row, int cc, int b) {
    const long e = row * p.C + cc * 8;
    f32x4 bz0 = {0.f, 0.f, 0.f, 0.f}, bz1 = bz0, az0 = bz0, az1 = bz0;
    if (p.sk_bias) { bz0 = *reinterpret_cast<const f32x4*>(p.sk_bias + cc * 8); bz1 = *reinterpret_cast<const f32x4*>(p.sk_bias + cc * 8 + 4); }
    if (p.sk_addvec) {
        const float* av = p.sk_addvec + (long)b * p.sk_ldav + cc * 8;
        az0 = *reinterpret_cast<const f32x4*>(av); az1 = *reinterpret_cast<const f32x4*>(av + 4);
    }
    // K ranges in two batches of four (eight 16-byte loads per piece and batch); the order of the adds is the reduce kernel's
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
#pragma unroll
    for (int h4 = 0; h4 < 8; h4 += 4) {
        if (h4 < p.sk_n) {   // block-uniform
            f32x4 w0[4], w1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool on = h4 + j < p.sk_n;
                w0[j] = on ? *reinterpret_cast<const f32x4*>(p.sk_partial + (long)(h4 + j) * p.sk_MN + e) : (f32x4){0.f, 0.f, 0.f, 0.f};
                w1[j] = on ? *reinterpret_cast<const f32x4*>(p.sk_partial + (long)(h4 + j) * p.sk_MN + e + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (h4 + j == 0) { v0 = w0[0]; v1 = w1[0]; }
                else if (h4 + j < p.sk_n) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v0[r] += w0[j][r]; v1[r] += w1[j][r]; }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (p.sk_bias) { v0[r] += bz0[r]; v1[r] += bz1[r]; }
        if (p.sk_addvec) { v0[r] += az0[r]; v1[r] += az1[r]; }
    }
    return (u32x4){pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
}

// (1) partial sums.  blockDim.x = ncc * rpp (ncc = C/8 column chunks, rpp rows in flight); grid = (nchunk, B).
// Each thread owns ONE 8-channel column chunk for the whole block: no index arithmetic in the row loop.
__global__ void gn_stats_kernel(const GNArgs p) {
    extern __shared__ float lds[];  // [rpp][2][C]: one slot per (row slice, channel) -> fixed-order, deterministic fold
    const int ncc = p.C / 8;
    const int cc = threadIdx.x % ncc, rr = threadIdx.x / ncc, rpp = blockDim.x / ncc;
    const int b = blockIdx.y, chunk = blockIdx.x;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    const int r0 = chunk * p.rows_per_chunk;
    const int r1 = min(r0 + p.rows_per_chunk, p.HW);
    if (rr < rpp) {
        u32x4 v[GN_MAXR];  // all of this thread's rows in flight at once (rows_per_chunk <= GN_MAXR * rpp)
#pragma unroll
        for (int j = 0; j < GN_MAXR; ++j) v[j] = gn_load(p, (long)b * p.HW + min(r0 + rr + j * rpp, r1 - 1), cc);
#pragma unroll
        for (int j = 0; j < GN_MAXR; ++j) {
            const float keep = (r0 + rr + j * rpp < r1) ? 1.f : 0.f;
            const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bf16lo(w[e]) * keep, c = bf16hi(w[e]) * keep;
                s[2 * e] += a; ss[2 * e] += a * a;
                s[2 * e + 1] += c; ss[2 * e + 1] += c * c;
            }
        }
        float* d0 = lds + (long)rr * 2 * p.C + cc * 8;
        *reinterpret_cast<f32x4*>(d0) = (f32x4){s[0], s[1], s[2], s[3]};
        *reinterpret_cast<f32x4*>(d0 + 4) = (f32x4){s[4], s[5], s[6], s[7]};
        *reinterpret_cast<f32x4*>(d0 + p.C) = (f32x4){ss[0], ss[1], ss[2], ss[3]};
        *reinterpret_cast<f32x4*>(d0 + p.C + 4) = (f32x4){ss[4], ss[5], ss[6], ss[7]};
    }
    __syncthreads();
    const int cpg = p.C / p.groups;
    if (threadIdx.x < p.groups) {
        float a = 0.f, c = 0.f;
        for (int j = 0; j < rpp; ++j)
            for (int i = 0; i < cpg; ++i) {
                a += lds[(long)j * 2 * p.C + threadIdx.x * cpg + i];
                c += lds[(long)j * 2 * p.C + p.C + threadIdx.x * cpg + i];
            }
        float* dst = p.part + (((long)b * p.nchunk + chunk) * p.groups + threadIdx.x) * 2;
        dst[0] = a;
        dst[1] = c;
    }
    if (p.counters) {  // block-uniform: the finalize launch is folded into the last block of the sample
        __shared__ float red[2][16][64];
        __shared__ float mean[64], rstd[64];
        __shared__ int s_flag;
        if (gn_last_block(p.counters, b, &s_flag)) {
            gn_finalize_body(p, b, red, mean, rstd);
            if (threadIdx.x == 0) p.counters[b] = 0;
        }
    }
}

// (2) finalize: one 1024-thread block per batch element folds the partials (fixed order -> deterministic) into per-channel
// scale = gamma * rstd, shift = beta - mean * scale.  16 slices x 64 group lanes so the partial loads run in parallel.
__global__ __launch_bounds__(1024) void gn_finalize_kernel(const GNArgs p) {
    __shared__ float red[2][16][64];
    __shared__ float mean[64], rstd[64];
    gn_finalize_body(p, blockIdx.x, red, mean, rstd);
}

// (2') finalize from the producers' per-channel slab statistics (ae_gemm_bf16 / ae_conv3x3_bf16 / ae_ln_gemm_bf16 `colstats`): one
// block per (group, sample) sums the group's channels over the sample's HW / 32 slabs in a fixed order (thread-strided partials,
// shuffle tree, four waves in wave order -> deterministic) and writes the per-channel scale / shift of that group.  The activation
// itself is not read: the statistics pass of GroupNorm (one full read of the tensor, 31 launches per UNet evaluation) is gone.
__global__ __launch_bounds__(256) void gn_finalize_cs_kernel(const GNArgs p) {
    __shared__ float red[2][4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = p.C / p.groups, nslab = p.HW / 32, C2 = p.C - p.C1;
    float a = 0.f, c = 0.f;
    // GN_CS_U independent loads per thread in flight (the launch is nothing but their latency: 6.2 -> ~3.5 us per GroupNorm with four); the slab index comes from a
    // float reciprocal —
    // (idx + 0.5) / cpg is never closer than 0.5 / cpg to an integer, exact for idx < 2^20.  The order of the thread's additions is fixed (u ascending).
    const int total = nslab * cpg;
    const float inv_cpg = 1.0f / (float)cpg;
    for (int base = threadIdx.x; base < total; base += 256 * GN_CS_U) {
        f32x2 v[GN_CS_U];
#pragma unroll
        for (int u = 0; u < GN_CS_U; ++u) {
            const int idx = base + 256 * u;
            const int i = (int)(((float)idx + 0.5f) * inv_cpg), ch = g * cpg + (idx - i * cpg);
            const long slab = (long)b * nslab + i;
            v[u] = (f32x2){0.f, 0.f};
            if (idx < total)
                v[u] = ch < p.C1 ? *reinterpret_cast<const f32x2*>(p.cs1 + (slab * p.C1 + ch) * 2)
                                 : *reinterpret_cast<const f32x2*>(p.cs2 + (slab * C2 + (ch - p.C1)) * 2);
        }
#pragma unroll
        for (int u = 0; u < GN_CS_U; ++u) { a += v[u][0]; c += v[u][1]; }
    }
    a = wave_reduce_sum(a);
    c = wave_reduce_sum(c);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = c; }
    __syncthreads();
    const float sa = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
    const float sq = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
    const float n = (float)cpg * (float)p.HW;
    const float mu = sa / n;
    const float var = fmaxf(sq / n - mu * mu, 0.f);
    const float rs = rsqrtf(var + p.eps);
    if (threadIdx.x == 0 && p.stat) {
        p.stat[((long)b * p.groups + g) * 2 + 0] = mu;
        p.stat[((long)b * p.groups + g) * 2 + 1] = rs;
    }
    for (int j = threadIdx.x; j < cpg; j += 256) {
        const int ch = g * cpg + j;
        const float sc = p.gamma[ch] * rs;
        p.coef[((long)b * 2 + 0) * p.C + ch] = sc;
        p.coef[((long)b * 2 + 1) * p.C + ch] = p.beta[ch] - mu * sc;
    }
}

// (3) apply: pure streaming.  Same thread layout as (1): a thread keeps its 8 scales + 8 shifts in registers.
// FUSED: the finalize fold (2) runs inside every apply block while that block's row loads are in flight — same slices, same order,
// same arithmetic as gn_finalize_kernel, so the results are bit-identical; it re-reads nchunk*groups*2 floats of partials per block
// (L2-resident, ~16 KiB at 64x64) and saves one launch per GroupNorm (61 per UNet evaluation).  Measured slower in situ (see the
// AE_GN_FUSE knob at the launch site): kept as the non-default variant.
template <bool FUSED>
__global__ void gn_apply_kernel(const GNArgs p) {
    __shared__ float red[FUSED ? 2 : 1][FUSED ? 16 : 1][FUSED ? 64 : 1];
    __shared__ float mean[FUSED ? 64 : 1], rstd[FUSED ? 64 : 1];
    const int ncc = p.C / 8;
    const int cc = threadIdx.x % ncc, rr = threadIdx.x / ncc, rpp = blockDim.x / ncc;
    const bool live = rr < rpp;
    if (!FUSED && !live) return;
    const int b = blockIdx.y;
    const int r0 = blockIdx.x * p.rows_per_chunk;
    const int r1 = min(r0 + p.rows_per_chunk, p.HW);
    u32x4 vv[GN_MAXR];
    if (live) {
#pragma unroll
        for (int j = 0; j < GN_MAXR; ++j) vv[j] = gn_load(p, (long)b * p.HW + min(r0 + rr + j * rpp, r1 - 1), cc);
    }
    float sc[8], sh[8];
    if (FUSED) {
        for (int slot = threadIdx.x; slot < 1024; slot += blockDim.x) {
            const int g = slot & 63, part = slot >> 6;
            float a = 0.f, c = 0.f;
            if (g < p.groups) {
                for (int i = part; i < p.nchunk; i += 16) {
                    const float* src = p.part + (((long)b * p.nchunk + i) * p.groups + g) * 2;
                    a += src[0];
                    c += src[1];
                }
            }
            red[0][part][g] = a;
            red[1][part][g] = c;
        }
        __syncthreads();
        if (threadIdx.x < p.groups) {
            const int g = threadIdx.x;
            float sa = 0.f, sq = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) { sa += red[0][j][g]; sq += red[1][j][g]; }
            const float n = (float)(p.C / p.groups) * (float)p.HW;
            const float mu = sa / n;
            const float var = fmaxf(sq / n - mu * mu, 0.f);
            mean[g] = mu;
            rstd[g] = rsqrtf(var + p.eps);
        }
        __syncthreads();
        if (!live) return;
        const int cpg = p.C / p.groups;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = cc * 8 + e, gi = ch / cpg;
            sc[e] = p.gamma[ch] * rstd[gi];
            sh[e] = p.beta[ch] - mean[gi] * sc[e];
        }
    } else {
        const float* cs = p.coef + ((long)b * 2) * p.C + cc * 8;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(cs), sc1 = *reinterpret_cast<const f32x4*>(cs + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(cs + p.C), sh1 = *reinterpret_cast<const f32x4*>(cs + p.C + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[e] = sc0[e]; sc[4 + e] = sc1[e]; sh[e] = sh0[e]; sh[4 + e] = sh1[e]; }
    }
#pragma unroll
    for (int j = 0; j < GN_MAXR; ++j) {
        const int r = r0 + rr + j * rpp;
        if (r >= r1) break;
        const long row = (long)b * p.HW + r;
        const u32x4 v = vv[j];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = bf16lo(w[e]) * sc[2 * e] + sh[2 * e];
            float c = bf16hi(w[e]) * sc[2 * e + 1] + sh[2 * e + 1];
            if (p.act == 1) { a = silu_f(a); c = silu_f(c); }
            o[e] = pack_bf16x2(a, c);
        }
        *reinterpret_cast<u32x4*>(p.y + row * p.C + cc * 8) = (u32x4){o[0], o[1], o[2], o[3]};
    }
}

// ---- Single-launch GroupNorm for the small feature maps (16x16 and below): one block owns the whole slab of GP groups of one
// sample — HW rows x (GP * C/groups) channels, at most 16 16-byte pieces per thread — and keeps it in REGISTERS: one read of the
// activation, exact two-pass statistics (mean, then centred sum of squares), normalise + SiLU, one write.  Replaces the
// stats / finalize / apply launches (3 passes over the data, 15-20 us for 2-15 MB) where the tensor is too small to fill the
// chip per launch anyway.  GP = 1, 2 or 4 groups per block makes the slab's row segment a multiple of 16 bytes.
// SK (round 6): the slab's pieces come from a split-K conv's partials (gn_load_splitk) instead of the reduced tensor — the reduce launch and one round trip of the
// activation vanish; everything after the load is the same code on the same bf16 values.
template <int MAXCH, int GP, int SK = 0>
__global__ __launch_bounds__(SK == 1 ? 256 : 1024) void gn_slab_kernel(const GNArgs p, int ncc, int cpg) {
    __shared__ float red[16][GP];
    __shared__ float bc[2][GP];
    const int tid = threadIdx.x, T = blockDim.x, nw = T >> 6;
    const int b = blockIdx.y, ch0 = blockIdx.x * GP * cpg;  // first channel of the pack (multiple of 8)
    const int total = p.HW * ncc;
    u32x4 v[MAXCH];
    int grp[MAXCH];  // packed 3 bits per element: group (0..GP-1) of each of the 8 channels of the piece
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int i = tid + k * T;
        const int ii = min(i, total - 1);
        const int row = ii / ncc, cc = ii - row * ncc;
        // SK blocks are 256 threads (the same thread <-> piece map and block-sum order as the plain launch of these shapes: the statistics come out bit-identical),
        // so a lane may hold several pieces' partial loads in flight
        v[k] = SK ? gn_load_splitk(p, (long)b * p.HW + row, ch0 / 8 + cc, b) : gn_load(p, (long)b * p.HW + row, ch0 / 8 + cc);
        if (SK == 2) __builtin_amdgcn_sched_barrier(0);   // lab form (AE_GN_SPLITK_T=1024): 1024-thread blocks, one piece's partial loads in flight at a time (128 registers per lane)
        int gbits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e;
            const int gi = GP == 1 ? 0 : ((c >= cpg) + (GP > 2 ? (c >= 2 * cpg) + (c >= 3 * cpg) : 0));
            gbits |= gi << (3 * e);
        }
        grp[k] = i < total ? gbits : -1;
    }
    auto block_sum = [&](float (&acc)[GP], int slot) {  // -> bc[slot][g] = sum over the block, fixed order (deterministic)
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) acc[gi] = wave_reduce_sum(acc[gi]);
        if ((tid & 63) == 0) {
#pragma unroll
            for (int gi = 0; gi < GP; ++gi) red[tid >> 6][gi] = acc[gi];
        }
        __syncthreads();
        if (tid < GP) {
            float t = 0.f;
            for (int w = 0; w < nw; ++w) t += red[w][tid];
            bc[slot][tid] = t;
        }
        __syncthreads();
    };
    const float inv_n = 1.0f / ((float)cpg * (float)p.HW);
    float acc[GP];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) acc[gi] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        if (grp[k] < 0) continue;
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? bf16hi(w[e >> 1]) : bf16lo(w[e >> 1]);
            const int gi = (grp[k] >> (3 * e)) & 7;
#pragma unroll
            for (int q = 0; q < GP; ++q) acc[q] += (gi == q) ? x : 0.f;
        }
    }
    block_sum(acc, 0);
    float mu[GP];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) { mu[gi] = bc[0][gi] * inv_n; acc[gi] = 0.f; }
    // hipcc otherwise keeps the 8 unpacked floats (and, GP > 1, the 8 decoded group indices) of every piece alive from one pass to the
    // next (common subexpressions): 8-16 x MAXCH registers beside the packed slab, spilled under the 128-register cap of a 1024-thread
    // block.  Re-unpacking is one VALU per value.
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) asm volatile("" : "+v"(v[k]), "+v"(grp[k]));
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        if (grp[k] < 0) continue;
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? bf16hi(w[e >> 1]) : bf16lo(w[e >> 1]);
            const int gi = (grp[k] >> (3 * e)) & 7;
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                const float d = x - mu[q];
                acc[q] += (gi == q) ? d * d : 0.f;
            }
        }
    }
    block_sum(acc, 1);
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) asm volatile("" : "+v"(v[k]), "+v"(grp[k]));
    float rs[GP];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) rs[gi] = rsqrtf(bc[1][gi] * inv_n + p.eps);
    if (p.stat && tid == 0) {  // (mean, rstd) kept for the backward pass
#pragma unroll
        for (int gi = 0; gi < GP; ++gi) {
            p.stat[((long)b * p.groups + blockIdx.x * GP + gi) * 2 + 0] = mu[gi];
            p.stat[((long)b * p.groups + blockIdx.x * GP + gi) * 2 + 1] = rs[gi];
        }
    }
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        if (grp[k] < 0) continue;
        const int i = tid + k * T;
        const int row = i / ncc, cc = i - row * ncc;
        const int ch = ch0 + cc * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ch), g1 = *reinterpret_cast<const f32x4*>(p.gamma + ch + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + ch), b1 = *reinterpret_cast<const f32x4*>(p.beta + ch + 4);
        const float gm[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bt[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? bf16hi(w[e >> 1]) : bf16lo(w[e >> 1]);
            const int gi = (grp[k] >> (3 * e)) & 7;
            float m = mu[0], r = rs[0];
#pragma unroll
            for (int q = 1; q < GP; ++q) { m = (gi == q) ? mu[q] : m; r = (gi == q) ? rs[q] : r; }
            const float sc = gm[e] * r;
            float y = (x - m) * sc + bt[e];
            if (p.act == 1) y = silu_f(y);
            o[e] = y;
        }
        *reinterpret_cast<u32x4*>(p.y + ((long)b * p.HW + row) * p.C + ch) =
            (u32x4){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        __builtin_amdgcn_sched_barrier(0);  // one piece's gamma / beta (16 registers) at a time: hoisting all MAXCH pieces' loads spilled
    }
}

template <int GP>
bool launch_gn_slab(const GNArgs& p, int cpg, hipStream_t s) {
    const int ncc = GP * cpg / 8;
    const long total = (long)p.HW * ncc;
    int T = 256;
    while (T < 1024 && total > (long)T * 8) T *= 2;   // aim at <= 8 pieces per thread, at most 16
    const long per = (total + T - 1) / T;
    if (per > 16) return false;
    dim3 grid(p.groups / GP, p.B);
    if (per <= 2) hipLaunchKernelGGL((gn_slab_kernel<2, GP>), grid, dim3(T), 0, s, p, ncc, cpg);
    else if (per <= 4) hipLaunchKernelGGL((gn_slab_kernel<4, GP>), grid, dim3(T), 0, s, p, ncc, cpg);
    else if (per <= 8) hipLaunchKernelGGL((gn_slab_kernel<8, GP>), grid, dim3(T), 0, s, p, ncc, cpg);
    else hipLaunchKernelGGL((gn_slab_kernel<16, GP>), grid, dim3(T), 0, s, p, ncc, cpg);
    return true;
}

template <int GP>
bool launch_gn_slab_splitk(const GNArgs& p, int cpg, hipStream_t s) {
    const int ncc = GP * cpg / 8;
    const long total = (long)p.HW * ncc;
    static const int t_env = getenv("AE_GN_SPLITK_T") ? atoi(getenv("AE_GN_SPLITK_T")) : 256;   // lab knob: 1024 = wide blocks (statistics summed in another order: not bit-identical to the plain path)
    if (t_env == 1024) {
        int T2 = 256;
        while (T2 < 1024 && total > (long)T2 * 2) T2 *= 2;
        if ((total + T2 - 1) / T2 > 2) return false;
        hipLaunchKernelGGL((gn_slab_kernel<2, GP, 2>), dim3(p.groups / GP, p.B), dim3(T2), 0, s, p, ncc, cpg);
        return true;
    }
    const int T = 256;                       // launch_gn_slab's choice for total <= 2048 pieces: same thread <-> piece map, same block-sum order
    const long per = (total + T - 1) / T;
    if (per > 8) return false;
    dim3 grid(p.groups / GP, p.B);
    if (per <= 2) hipLaunchKernelGGL((gn_slab_kernel<2, GP, 1>), grid, dim3(T), 0, s, p, ncc, cpg);
    else if (per <= 4) hipLaunchKernelGGL((gn_slab_kernel<4, GP, 1>), grid, dim3(T), 0, s, p, ncc, cpg);
    else hipLaunchKernelGGL((gn_slab_kernel<8, GP, 1>), grid, dim3(T), 0, s, p, ncc, cpg);
    return true;
}

// ---- GroupNorm(+SiLU) backward (training step, SURVEY.md row A11: the frozen UNet is differentiated w.r.t. its activations).
//   z = x*scale + shift, y = act(z);  dz = dy * act'(z);  per (batch, group): m1 = mean(dz*gamma), m2 = mean(dz*gamma*xhat)
//   dx = rstd * (dz*gamma - m1 - xhat*m2) = dz*scale + x*kA + kB,  kA = -rstd^2 m2,  kB = mu rstd^2 m2 - rstd m1
// and dz*gamma*xhat = dz*(z - beta): no division by gamma anywhere.  Same launch geometry and fixed-order folds as the forward.
__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == 0) return 1.0f;
    const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
    return sg * (1.0f + z * (1.0f - sg));
}

__device__ __forceinline__ void gnb_finalize_body(const GNArgs& p, int b, float (*red)[16][64], float* kA, float* kB) {
    gn_fold(p.part2, b, p.nchunk, p.groups, red);
    if (threadIdx.x < p.groups) {
        const int g = threadIdx.x;
        float sa = 0.f, sc = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { sa += red[0][j][g]; sc += red[1][j][g]; }
        const float n = (float)(p.C / p.groups) * (float)p.HW;
        const float m1 = sa / n, m2 = sc / n;
        const float mu = p.stat[((long)b * p.groups + g) * 2 + 0], rs = p.stat[((long)b * p.groups + g) * 2 + 1];
        kA[g] = -rs * rs * m2;
        kB[g] = mu * rs * rs * m2 - rs * m1;
    }
    __syncthreads();
    const int cpg = p.C / p.groups;
    for (int ch = threadIdx.x; ch < p.C; ch += blockDim.x) {
        p.coef2[((long)b * 2 + 0) * p.C + ch] = kA[ch / cpg];
        p.coef2[((long)b * 2 + 1) * p.C + ch] = kB[ch / cpg];
    }
}

// forward coefficients of 8 channels: from the coef table written by the finalize fold, or (backward on statistics saved by the
// forward launch: coef == nullptr) from (mean, rstd) of the channel's group — the same two operations the finalize performs
__device__ __forceinline__ void gn_coef8(const GNArgs& p, int b, int cc, float (&sc)[8], float (&sh)[8]) {
    if (p.coef) {
        const float* cs = p.coef + ((long)b * 2) * p.C + cc * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = cs[e]; sh[e] = cs[p.C + e]; }
        return;
    }
    const int cpg = p.C / p.groups;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = cc * 8 + e, g = ch / cpg;
        const float mu = p.stat[((long)b * p.groups + g) * 2 + 0], rs = p.stat[((long)b * p.groups + g) * 2 + 1];
        sc[e] = p.gamma[ch] * rs;
        sh[e] = p.beta[ch] - mu * sc[e];
    }
}

__global__ void gnb_partial_kernel(const GNArgs p) {
    extern __shared__ float lds[];  // [rpp][2][C]
    const int ncc = p.C / 8;
    const int cc = threadIdx.x % ncc, rr = threadIdx.x / ncc, rpp = blockDim.x / ncc;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int r0 = chunk * p.rows_per_chunk;
    const int r1 = min(r0 + p.rows_per_chunk, p.HW);
    if (rr < rpp) {
        float t1[8], t2[8], sc[8], sh[8], gm[8], bt[8];
        gn_coef8(p, b, cc, sc, sh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            t1[e] = t2[e] = 0.f;
            gm[e] = p.gamma[cc * 8 + e]; bt[e] = p.beta[cc * 8 + e];
        }
        for (int r = r0 + rr; r < r1; r += rpp) {
            const long row = (long)b * p.HW + r;
            const u32x4 xv = gn_load(p, row, cc);
            const u32x4 dv = *reinterpret_cast<const u32x4*>(p.dy + row * p.C + cc * 8);
            const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z0 = bf16lo(xw[e]) * sc[2 * e] + sh[2 * e], z1 = bf16hi(xw[e]) * sc[2 * e + 1] + sh[2 * e + 1];
                const float d0 = bf16lo(dw[e]) * act_grad(z0, p.act), d1 = bf16hi(dw[e]) * act_grad(z1, p.act);
                t1[2 * e] += d0 * gm[2 * e]; t2[2 * e] += d0 * (z0 - bt[2 * e]);
                t1[2 * e + 1] += d1 * gm[2 * e + 1]; t2[2 * e + 1] += d1 * (z1 - bt[2 * e + 1]);
            }
        }
        float* d0 = lds + (long)rr * 2 * p.C + cc * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { d0[e] = t1[e]; d0[p.C + e] = t2[e]; }
    }
    __syncthreads();
    const int cpg = p.C / p.groups;
    if (threadIdx.x < p.groups) {
        float a = 0.f, c = 0.f;
        for (int j = 0; j < rpp; ++j)
            for (int i = 0; i < cpg; ++i) {
                a += lds[(long)j * 2 * p.C + threadIdx.x * cpg + i];
                c += lds[(long)j * 2 * p.C + p.C + threadIdx.x * cpg + i];
            }
        float* dst = p.part2 + (((long)b * p.nchunk + chunk) * p.groups + threadIdx.x) * 2;
        dst[0] = a;
        dst[1] = c;
    }
    if (p.counters) {
        __shared__ float red[2][16][64];
        __shared__ float kA[64], kB[64];
        __shared__ int s_flag;
        if (gn_last_block(p.counters, b, &s_flag)) {
            gnb_finalize_body(p, b, red, kA, kB);
            if (threadIdx.x == 0) p.counters[b] = 0;
        }
    }
}

__global__ __launch_bounds__(1024) void gnb_finalize_kernel(const GNArgs p) {
    __shared__ float red[2][16][64];
    __shared__ float kA[64], kB[64];
    gnb_finalize_body(p, blockIdx.x, red, kA, kB);
}

__global__ void gnb_apply_kernel(const GNArgs p) {
    const int ncc = p.C / 8;
    const int cc = threadIdx.x % ncc, rr = threadIdx.x / ncc, rpp = blockDim.x / ncc;
    if (rr >= rpp) return;
    const int b = blockIdx.y;
    float sc[8], sh[8], ka[8], kb[8];
    gn_coef8(p, b, cc, sc, sh);
    const float* c2 = p.coef2 + ((long)b * 2) * p.C + cc * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ka[e] = c2[e]; kb[e] = c2[p.C + e]; }
    const int r0 = blockIdx.x * p.rows_per_chunk;
    const int r1 = min(r0 + p.rows_per_chunk, p.HW);
    const int ch = cc * 8;
    for (int r = r0 + rr; r < r1; r += rpp) {
        const long row = (long)b * p.HW + r;
        const u32x4 xv = gn_load(p, row, cc);
        const u32x4 dv = *reinterpret_cast<const u32x4*>(p.dy + row * p.C + ch);
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, dw[4] = {dv.x, dv.y, dv.z, dv.w};
        uint32_t o[4];
        bf16_t* const dst = ch < p.C1 ? p.dx + row * p.C1 + ch : p.dx2 + row * (p.C - p.C1) + (ch - p.C1);
        u32x4 old = {0u, 0u, 0u, 0u};
        if (p.accum & (ch < p.C1 ? 1 : 2)) old = *reinterpret_cast<const u32x4*>(dst);   // fp32 add of the gradient already there, one rounding
        const uint32_t ow[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = bf16lo(xw[e]), x1 = bf16hi(xw[e]);
            const float z0 = x0 * sc[2 * e] + sh[2 * e], z1 = x1 * sc[2 * e + 1] + sh[2 * e + 1];
            const float d0 = bf16lo(dw[e]) * act_grad(z0, p.act), d1 = bf16hi(dw[e]) * act_grad(z1, p.act);
            o[e] = pack_bf16x2(d0 * sc[2 * e] + x0 * ka[2 * e] + kb[2 * e] + bf16lo(ow[e]), d1 * sc[2 * e + 1] + x1 * ka[2 * e + 1] + kb[2 * e + 1] + bf16hi(ow[e]));
        }
        *reinterpret_cast<u32x4*>(dst) = (u32x4){o[0], o[1], o[2], o[3]};
    }
}

// ---- Single-launch GroupNorm backward for the small feature maps (round 5; the backward twin of gn_slab_kernel): one block owns the slab of GP groups of
// one sample — x AND dy pieces in registers (2 x MAXCH 16-byte pieces per thread) — with (mean, rstd) from the forward launch (stat_in):
//   z = xhat gamma + beta, dz = dy act'(z);  per group  m1 = mean(dz gamma), m2 = mean(dz gamma xhat);  dx = rstd (dz gamma - m1 - xhat m2)
// one read of x and dy, one block-wide fixed-order sum, one write: replaces the partial / finalize / apply launches (9 + 5 + 7 us on 0.3-2.6 MB at a
// training batch: 23 of a step's 58 GroupNorm backward calls).  Same group-pack geometry as the forward kernel.
template <int MAXCH, int GP>
__global__ __launch_bounds__(1024) void gnb_slab_kernel(const GNArgs p, int ncc, int cpg) {
    __shared__ float red[16][2 * GP];
    __shared__ float bc[2 * GP];
    const int tid = threadIdx.x, T = blockDim.x, nw = T >> 6;
    const int b = blockIdx.y, ch0 = blockIdx.x * GP * cpg;  // first channel of the pack (multiple of 8)
    const int total = p.HW * ncc;
    u32x4 v[MAXCH], dv[MAXCH];
    int grp[MAXCH];  // 3 bits per element: group (0..GP-1) of each of the 8 channels of the piece; -1: no piece
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        const int i = tid + k * T;
        const int ii = min(i, total - 1);
        const int row = ii / ncc, cc = ii - row * ncc;
        v[k] = gn_load(p, (long)b * p.HW + row, ch0 / 8 + cc);
        dv[k] = *reinterpret_cast<const u32x4*>(p.dy + ((long)b * p.HW + row) * p.C + ch0 + cc * 8);
        int gbits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e;
            const int gi = GP == 1 ? 0 : ((c >= cpg) + (GP > 2 ? (c >= 2 * cpg) + (c >= 3 * cpg) : 0));
            gbits |= gi << (3 * e);
        }
        grp[k] = i < total ? gbits : -1;
    }
    float mu[GP], rs[GP];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) {
        mu[gi] = p.stat[((long)b * p.groups + blockIdx.x * GP + gi) * 2 + 0];
        rs[gi] = p.stat[((long)b * p.groups + blockIdx.x * GP + gi) * 2 + 1];
    }
    // one element: xhat, dz gamma (dz = dy act'(xhat gamma + beta))
    auto elem = [&](float x, float d, float g, float bt, float m, float r, float& xh, float& dg) {
        xh = (x - m) * r;
        const float z = xh * g + bt;
        dg = d * act_grad(z, p.act) * g;
    };
    float acc[2 * GP];
#pragma unroll
    for (int q = 0; q < 2 * GP; ++q) acc[q] = 0.f;
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        if (grp[k] < 0) continue;
        const int i = tid + k * T;
        const int cc = i % ncc;
        const int ch = ch0 + cc * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ch), g1 = *reinterpret_cast<const f32x4*>(p.gamma + ch + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + ch), b1 = *reinterpret_cast<const f32x4*>(p.beta + ch + 4);
        const float gm[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bt[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, dw[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? bf16hi(w[e >> 1]) : bf16lo(w[e >> 1]);
            const float d = (e & 1) ? bf16hi(dw[e >> 1]) : bf16lo(dw[e >> 1]);
            const int gi = (grp[k] >> (3 * e)) & 7;
            float m = mu[0], r = rs[0];
#pragma unroll
            for (int q = 1; q < GP; ++q) { m = (gi == q) ? mu[q] : m; r = (gi == q) ? rs[q] : r; }
            float xh, dg;
            elem(x, d, gm[e], bt[e], m, r, xh, dg);
#pragma unroll
            for (int q = 0; q < GP; ++q) {
                acc[2 * q] += (gi == q) ? dg : 0.f;
                acc[2 * q + 1] += (gi == q) ? dg * xh : 0.f;
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // one piece's gamma / beta at a time (see gn_slab_kernel)
    }
    // block-wide sums, fixed order (deterministic)
#pragma unroll
    for (int q = 0; q < 2 * GP; ++q) acc[q] = wave_reduce_sum(acc[q]);
    if ((tid & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 2 * GP; ++q) red[tid >> 6][q] = acc[q];
    }
    __syncthreads();
    if (tid < 2 * GP) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += red[w][tid];
        bc[tid] = t;
    }
    __syncthreads();
    const float inv_n = 1.0f / ((float)cpg * (float)p.HW);
    float m1[GP], m2[GP];
#pragma unroll
    for (int gi = 0; gi < GP; ++gi) { m1[gi] = bc[2 * gi] * inv_n; m2[gi] = bc[2 * gi + 1] * inv_n; }
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) asm volatile("" : "+v"(v[k]), "+v"(dv[k]), "+v"(grp[k]));   // re-unpack instead of keeping 16 floats per piece alive (gn_slab_kernel)
#pragma unroll
    for (int k = 0; k < MAXCH; ++k) {
        if (grp[k] < 0) continue;
        const int i = tid + k * T;
        const int row = i / ncc, cc = i - row * ncc;
        const int ch = ch0 + cc * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ch), g1 = *reinterpret_cast<const f32x4*>(p.gamma + ch + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + ch), b1 = *reinterpret_cast<const f32x4*>(p.beta + ch + 4);
        const float gm[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bt[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w}, dw[4] = {dv[k].x, dv[k].y, dv[k].z, dv[k].w};
        const long grow = (long)b * p.HW + row;
        bf16_t* const dst = ch < p.C1 ? p.dx + grow * p.C1 + ch : p.dx2 + grow * (p.C - p.C1) + (ch - p.C1);
        u32x4 old = {0u, 0u, 0u, 0u};
        if (p.accum & (ch < p.C1 ? 1 : 2)) old = *reinterpret_cast<const u32x4*>(dst);
        const uint32_t ow[4] = {old.x, old.y, old.z, old.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (e & 1) ? bf16hi(w[e >> 1]) : bf16lo(w[e >> 1]);
            const float d = (e & 1) ? bf16hi(dw[e >> 1]) : bf16lo(dw[e >> 1]);
            const int gi = (grp[k] >> (3 * e)) & 7;
            float m = mu[0], r = rs[0], a1 = m1[0], a2 = m2[0];
#pragma unroll
            for (int q = 1; q < GP; ++q) { m = (gi == q) ? mu[q] : m; r = (gi == q) ? rs[q] : r; a1 = (gi == q) ? m1[q] : a1; a2 = (gi == q) ? m2[q] : a2; }
            float xh, dg;
            elem(x, d, gm[e], bt[e], m, r, xh, dg);
            o[e] = r * (dg - a1 - xh * a2) + ((e & 1) ? bf16hi(ow[e >> 1]) : bf16lo(ow[e >> 1]));
        }
        *reinterpret_cast<u32x4*>(dst) = (u32x4){pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int GP>
bool launch_gnb_slab(const GNArgs& p, int cpg, hipStream_t s) {
    const int ncc = GP * cpg / 8;
    const long total = (long)p.HW * ncc;
    int T = 256;
    while (T < 1024 && total > (long)T * 4) T *= 2;   // at most 4 (x, dy) piece pairs per thread (8 pairs spill under the 128-register cap of a 1024-thread block)
    const long per = (total + T - 1) / T;
    if (per > 4) return false;
    dim3 grid(p.groups / GP, p.B);
    if (per <= 2) hipLaunchKernelGGL((gnb_slab_kernel<2, GP>), grid, dim3(T), 0, s, p, ncc, cpg);
    else hipLaunchKernelGGL((gnb_slab_kernel<4, GP>), grid, dim3(T), 0, s, p, ncc, cpg);
    return true;
}

// LayerNorm backward w.r.t. the input: t = dy*gamma, dx = rstd * (t - mean(t) - xhat * mean(t*xhat)); one wave per row.
// Optionally writes (mean, rstd) per row for the parameter-gradient kernel.
template <int MAXCH>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* x, const float* gamma, const bf16_t* dy, bf16_t* dx,
                                                            float* row_stat, int M, int C, float eps, int accum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int ncc = C / 8;
    float xv[MAXCH][8], tv[MAXCH][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + i * 64;
        if (cc < ncc) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(x + row * C + cc * 8);
            const u32x4 d = *reinterpret_cast<const u32x4*>(dy + row * C + cc * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xv[i][2 * e] = bf16lo(w[e]); xv[i][2 * e + 1] = bf16hi(w[e]);
                tv[i][2 * e] = bf16lo(dw[e]) * gamma[cc * 8 + 2 * e]; tv[i][2 * e + 1] = bf16hi(dw[e]) * gamma[cc * 8 + 2 * e + 1];
                s += xv[i][2 * e] + xv[i][2 * e + 1];
            }
        }
    }
    const float mu = wave_reduce_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
        if (lane + i * 64 < ncc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float a = xv[i][e] - mu; q += a * a; }
        }
    const float rstd = rsqrtf(wave_reduce_sum(q) / (float)C + eps);
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
        if (lane + i * 64 < ncc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { a1 += tv[i][e]; a2 += tv[i][e] * (xv[i][e] - mu) * rstd; }
        }
    const float m1 = wave_reduce_sum(a1) / (float)C, m2 = wave_reduce_sum(a2) / (float)C;
    if (row_stat && lane == 0) { row_stat[row * 2] = mu; row_stat[row * 2 + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + i * 64;
        if (cc < ncc) {
            uint32_t o[4];
            u32x4 old = {0u, 0u, 0u, 0u};
            if (accum) old = *reinterpret_cast<const u32x4*>(dx + row * C + cc * 8);   // dx += : fp32 add of the gradient already there, one rounding
            const uint32_t ow[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h0 = (xv[i][2 * e] - mu) * rstd, h1 = (xv[i][2 * e + 1] - mu) * rstd;
                o[e] = pack_bf16x2(rstd * (tv[i][2 * e] - m1 - h0 * m2) + bf16lo(ow[e]), rstd * (tv[i][2 * e + 1] - m1 - h1 * m2) + bf16hi(ow[e]));
            }
            *reinterpret_cast<u32x4*>(dx + row * C + cc * 8) = (u32x4){o[0], o[1], o[2], o[3]};
        }
    }
}

// dgamma[c] = sum_m dy*xhat, dbeta[c] = sum_m dy (fixed order over rows: deterministic); one thread per channel, small M only
// (the image-projection LayerNorm of the AnySD adapter: 16 rows).
__global__ void layernorm_param_grad_kernel(const bf16_t* x, const bf16_t* dy, const float* row_stat, float* dgamma, float* dbeta,
                                            int M, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float g = 0.f, bsum = 0.f;
    for (int m = 0; m < M; ++m) {
        const float d = bf16_to_f32(dy[(long)m * C + c]);
        g += d * (bf16_to_f32(x[(long)m * C + c]) - row_stat[m * 2]) * row_stat[m * 2 + 1];
        bsum += d;
    }
    dgamma[c] = g;
    dbeta[c] = bsum;
}

// LayerNorm over the last dim: one wave per row, 16-byte loads, two-pass (mean, then centred variance) in registers.
template <int MAXCH>  // max 16-byte chunks per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y,
                                                        int M, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)blockIdx.x * 4 + wave;
    if (row >= M) return;
    const int ncc = C / 8;
    u32x4 v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + i * 64;
        if (cc < ncc) {
            v[i] = *reinterpret_cast<const u32x4*>(x + row * C + cc * 8);
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bf16lo(w[e]) + bf16hi(w[e]);
        }
    }
    const float mu = wave_reduce_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + i * 64;
        if (cc < ncc) {
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bf16lo(w[e]) - mu, c = bf16hi(w[e]) - mu;
                q += a * a + c * c;
            }
        }
    }
    const float rstd = rsqrtf(wave_reduce_sum(q) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + i * 64;
        if (cc < ncc) {
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            uint32_t o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c0 = cc * 8 + 2 * e;
                const float a = (bf16lo(w[e]) - mu) * rstd * gamma[c0] + beta[c0];
                const float c = (bf16hi(w[e]) - mu) * rstd * gamma[c0 + 1] + beta[c0 + 1];
                o[e] = pack_bf16x2(a, c);
            }
            *reinterpret_cast<u32x4*>(y + row * C + cc * 8) = (u32x4){o[0], o[1], o[2], o[3]};
        }
    }
}

// LayerNorm with L lanes per row and NCH 16-byte chunks per lane (C = 8 L NCH: 320 / 640 / 1280 / 2560 channels as 8 / 16 / 32 / 64
// lanes x 5 chunks): 64 / L rows per wave, every lane busy, NCH loads in flight per lane, gamma / beta as 16-byte loads at their use.
// The one-row-per-wave kernel above leaves 48 of 64 lanes idle on its second chunk at C = 640 and keeps one row per wave in flight:
// 12.8 us for the 31 MB of a [12288, 640] LayerNorm (2.4 TB/s, profiles/r03_v1_kernels_by_shape.json).
template <int L, int NCH>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, int M, int C,
                                                             float eps) {
    const int lane = threadIdx.x & 63, sub = lane % L;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / L) + lane / L;
    const bool live = row < M;
    const long rowc = live ? row : (long)M - 1;
    u32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = *reinterpret_cast<const u32x4*>(x + rowc * C + (sub + i * L) * 8);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) s += bf16lo(w[e]) + bf16hi(w[e]);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf16lo(w[e]) - mu, c = bf16hi(w[e]) - mu;
            q += a * a + c * c;
        }
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = (sub + i * L) * 8;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + c0), g1 = *reinterpret_cast<const f32x4*>(gamma + c0 + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + c0), b1 = *reinterpret_cast<const f32x4*>(beta + c0 + 4);
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        u32x4 o;
        o.x = pack_bf16x2((bf16lo(w[0]) - mu) * rstd * g0[0] + b0[0], (bf16hi(w[0]) - mu) * rstd * g0[1] + b0[1]);
        o.y = pack_bf16x2((bf16lo(w[1]) - mu) * rstd * g0[2] + b0[2], (bf16hi(w[1]) - mu) * rstd * g0[3] + b0[3]);
        o.z = pack_bf16x2((bf16lo(w[2]) - mu) * rstd * g1[0] + b1[0], (bf16hi(w[2]) - mu) * rstd * g1[1] + b1[1]);
        o.w = pack_bf16x2((bf16lo(w[3]) - mu) * rstd * g1[2] + b1[2], (bf16hi(w[3]) - mu) * rstd * g1[3] + b1[3]);
        *reinterpret_cast<u32x4*>(y + row * C + c0) = o;
    }
}

// SAM's windowed blocks (image_encoder.py:166-182, 243-289): the LayerNorm of layernorm_rows_kernel with the window partition / un-partition
// folded into its row addressing — the same operations in the same order per row as the launches they replace (the residual sum bit for bit; the
// normalised values up to hipcc's per-kernel choice of fused / unfused multiply-adds: a few elements per million, one bf16 step).
//   MODE 1 (norm1 + window_partition): one OUTPUT row (window order, [B*nH*nW*ws*ws, C]) per row slot; rows of the bottom / right padding
//          are written as zeros (the padding is applied AFTER the norm: :169-173), the others are LayerNorm(x[image row]).
//   MODE 2 (window_unpartition + shortcut + norm2): one IMAGE row per row slot; v = bf16(windows[window row] + shortcut[row]) is written to
//          `xsum` (the block's residual stream, :175-181) and LayerNorm(v) to y.
struct LnWinArgs {
    const bf16_t* x;         // MODE 1: image rows; MODE 2: window rows (attention output)
    const bf16_t* shortcut;  // MODE 2: image rows
    const float* gamma;
    const float* beta;
    bf16_t* y;               // MODE 1: window rows; MODE 2: image rows
    bf16_t* xsum;            // MODE 2
    int B, H, W, C, ws, nH, nW;
    long rows;               // row slots: MODE 1 B*nH*nW*ws*ws, MODE 2 B*H*W
    float eps;
};

template <int L, int NCH, int MODE>
__global__ __launch_bounds__(256) void layernorm_window_kernel(LnWinArgs p) {
    const int lane = threadIdx.x & 63, sub = lane % L;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / L) + lane / L;
    const bool live = row < p.rows;
    const long rowc = live ? row : p.rows - 1;
    const int C = p.C, ws = p.ws;
    long img, win;
    bool pad = false;
    if (MODE == 1) {
        long r = rowc;
        const int wx = (int)(r % ws); r /= ws;
        const int wy = (int)(r % ws); r /= ws;
        const int jw = (int)(r % p.nW); r /= p.nW;
        const int jh = (int)(r % p.nH); r /= p.nH;
        const int yy = jh * ws + wy, xx = jw * ws + wx;
        pad = yy >= p.H || xx >= p.W;
        img = pad ? 0 : (r * p.H + yy) * p.W + xx;
        win = rowc;
    } else {
        long r = rowc;
        const int xx = (int)(r % p.W); r /= p.W;
        const int yy = (int)(r % p.H); r /= p.H;
        win = (((r * p.nH + yy / ws) * p.nW + xx / ws) * ws + yy % ws) * ws + xx % ws;
        img = rowc;
    }
    u32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = *reinterpret_cast<const u32x4*>(p.x + (MODE == 1 ? img : win) * C + (sub + i * L) * 8);
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const u32x4 b = *reinterpret_cast<const u32x4*>(p.shortcut + img * C + (sub + i * L) * 8);
            // the argument order of ops.add_bcast(windows, shortcut): bf16(windows + shortcut)
            v[i].x = pack_bf16x2(bf16lo(v[i].x) + bf16lo(b.x), bf16hi(v[i].x) + bf16hi(b.x));
            v[i].y = pack_bf16x2(bf16lo(v[i].y) + bf16lo(b.y), bf16hi(v[i].y) + bf16hi(b.y));
            v[i].z = pack_bf16x2(bf16lo(v[i].z) + bf16lo(b.z), bf16hi(v[i].z) + bf16hi(b.z));
            v[i].w = pack_bf16x2(bf16lo(v[i].w) + bf16lo(b.w), bf16hi(v[i].w) + bf16hi(b.w));
            if (live) *reinterpret_cast<u32x4*>(p.xsum + img * C + (sub + i * L) * 8) = v[i];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) s += bf16lo(w[e]) + bf16hi(w[e]);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mu = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf16lo(w[e]) - mu, c = bf16hi(w[e]) - mu;
            q += a * a + c * c;
        }
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)C + p.eps);
    if (!live) return;
    bf16_t* yrow = p.y + (MODE == 1 ? win : img) * C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c0 = (sub + i * L) * 8;
        u32x4 o = {0u, 0u, 0u, 0u};
        if (!pad) {
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + c0), g1 = *reinterpret_cast<const f32x4*>(p.gamma + c0 + 4);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + c0), b1 = *reinterpret_cast<const f32x4*>(p.beta + c0 + 4);
            const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            o.x = pack_bf16x2((bf16lo(w[0]) - mu) * rstd * g0[0] + b0[0], (bf16hi(w[0]) - mu) * rstd * g0[1] + b0[1]);
            o.y = pack_bf16x2((bf16lo(w[1]) - mu) * rstd * g0[2] + b0[2], (bf16hi(w[1]) - mu) * rstd * g0[3] + b0[3]);
            o.z = pack_bf16x2((bf16lo(w[2]) - mu) * rstd * g1[0] + b1[0], (bf16hi(w[2]) - mu) * rstd * g1[1] + b1[1]);
            o.w = pack_bf16x2((bf16lo(w[3]) - mu) * rstd * g1[2] + b1[2], (bf16hi(w[3]) - mu) * rstd * g1[3] + b1[3]);
        }
        *reinterpret_cast<u32x4*>(yrow + c0) = o;
    }
}

// LayerNorm over a narrow last dim (C <= 512) with an optional fused GELU: L = pow2 lanes per row, 64 / L rows per wave, so the
// LayerNorm2d + GELU pairs of the SAM mask decoder (mask_decoder.py:53-60; C = 64 over 16384*B pixels) keep every lane busy.
template <int L, int ACT>
__global__ __launch_bounds__(256) void layernorm_narrow_kernel(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y,
                                                               long M, int C, float eps) {
    const int lane = threadIdx.x & 63, sub = lane % L;
    const long row = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / L) + lane / L;
    const bool live = row < M && sub < C / 8;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    if (live) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + row * C + sub * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[2 * e] = bf16lo(w[e]); f[2 * e + 1] = bf16hi(w[e]); }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += f[e];
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s / (float)C;
    float q = 0.f;
    if (live) {
#pragma unroll
        for (int e = 0; e < 8; ++e) q += (f[e] - mu) * (f[e] - mu);
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (live) {
        uint32_t o32[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c0 = sub * 8 + 2 * e;
            float a = (f[2 * e] - mu) * rstd * gamma[c0] + beta[c0];
            float c = (f[2 * e + 1] - mu) * rstd * gamma[c0 + 1] + beta[c0 + 1];
            if (ACT == 1) { a = gelu_erf_f(a); c = gelu_erf_f(c); }
            o32[e] = pack_bf16x2(a, c);
        }
        *reinterpret_cast<u32x4*>(y + row * C + sub * 8) = (u32x4){o32[0], o32[1], o32[2], o32[3]};
    }
}

template <int ACT>
static void launch_layernorm_narrow(const bf16_t* x, const float* gamma, const float* beta, bf16_t* y, long M, int C, float eps,
                                    hipStream_t s) {
    const int ncc = C / 8;
    int L = 1;
    while (L < ncc) L *= 2;
    const long rows_per_block = 4L * (64 / L);
    dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block)), block(256);
    switch (L) {
        case 1: hipLaunchKernelGGL((layernorm_narrow_kernel<1, ACT>), grid, block, 0, s, x, gamma, beta, y, M, C, eps); break;
        case 2: hipLaunchKernelGGL((layernorm_narrow_kernel<2, ACT>), grid, block, 0, s, x, gamma, beta, y, M, C, eps); break;
        case 4: hipLaunchKernelGGL((layernorm_narrow_kernel<4, ACT>), grid, block, 0, s, x, gamma, beta, y, M, C, eps); break;
        case 8: hipLaunchKernelGGL((layernorm_narrow_kernel<8, ACT>), grid, block, 0, s, x, gamma, beta, y, M, C, eps); break;
        case 16: hipLaunchKernelGGL((layernorm_narrow_kernel<16, ACT>), grid, block, 0, s, x, gamma, beta, y, M, C, eps); break;
        case 32: hipLaunchKernelGGL((layernorm_narrow_kernel<32, ACT>), grid, block, 0, s, x, gamma, beta, y, M, C, eps); break;
        default: hipLaunchKernelGGL((layernorm_narrow_kernel<64, ACT>), grid, block, 0, s, x, gamma, beta, y, M, C, eps); break;
    }
}

}  // namespace

extern "C" int ae_groupnorm_rows_per_chunk(int HW, int C) {
    // every thread keeps all its rows in flight (GN_MAXR 16-byte loads); ~6 blocks per CU at the 64x64 levels
    const int ncc = C / 8;
    int rpp = 256 / ncc;
    if (rpp < 1) rpp = 1;
    int r = GN_MAXR * rpp;
    if (r > 32) r = 32;
    if (r > HW) r = HW;
    return r;
}

extern "C" long ae_groupnorm_workspace_floats(int B, int HW, int C, int groups) {
    const int rpc = ae_groupnorm_rows_per_chunk(HW, C);
    const int nchunk = (HW + rpc - 1) / rpc;
    return (((long)B * nchunk * groups * 2 + 3) / 4) * 4 + (long)B * 2 * C;  // partials + per-channel scale/shift
}

extern "C" int ae_groupnorm_nhwc_bf16(const void* x, const void* x2, int C1, const float* gamma, const float* beta, void* y,
                                      int B, int HW, int C, int groups, float eps, int act, float* workspace, int* counters, float* stat_out,
                                      const float* colstats, const float* colstats2, void* stream) {
    AE_REQUIRE(x && gamma && beta && y && workspace, "ae_groupnorm_nhwc_bf16: null pointer");
    AE_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "ae_groupnorm_nhwc_bf16: bad shape C=%d groups=%d", C, groups);
    AE_REQUIRE(C % 8 == 0 && C <= 8192, "ae_groupnorm_nhwc_bf16: C=%d must be a multiple of 8 and <= 8192", C);
    AE_REQUIRE(groups <= 64, "ae_groupnorm_nhwc_bf16: groups=%d > 64 unsupported", groups);
    AE_REQUIRE(act == 0 || act == 1, "ae_groupnorm_nhwc_bf16: act must be 0 (none) or 1 (SiLU)");
    AE_REQUIRE(B <= 65535, "ae_groupnorm_nhwc_bf16: batch too large");
    if (x2) AE_REQUIRE(C1 > 0 && C1 < C && C1 % 8 == 0, "ae_groupnorm_nhwc_bf16: bad concat split C1=%d C=%d", C1, C);
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)x2 & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
               "ae_groupnorm_nhwc_bf16: 16-byte alignment");
    GNArgs p{};
    p.x = (const bf16_t*)x; p.x2 = (const bf16_t*)x2; p.C1 = x2 ? C1 : C;
    p.gamma = gamma; p.beta = beta; p.y = (bf16_t*)y;
    p.B = B; p.HW = HW; p.C = C; p.groups = groups; p.act = act; p.eps = eps;
    p.rows_per_chunk = ae_groupnorm_rows_per_chunk(HW, C);
    p.nchunk = (HW + p.rows_per_chunk - 1) / p.rows_per_chunk;
    p.part = workspace;
    p.coef = workspace + (((long)B * p.nchunk * groups * 2 + 3) / 4) * 4;
    p.stat = stat_out;
    // tuning knob: 1 lets the last partial-sum block of a sample run the finalize fold (one launch fewer).  Measured in situ (UNet batch 12,
    // two runs each way on one box): 18.6 ms per UNet step with the tail against 15.4 ms without — the device-scope release every block
    // needs before it takes its ticket is an L2 write-back on this multi-XCD part (buffer_wbl2), ~50 us per GroupNorm, far more than the
    // ~5 us launch it removes.  Training step 30.5 vs 27.4 ms.  Default off; the counters argument stays in the ABI for the A/B.
    static const int tail = getenv("AE_GN_TAIL") ? atoi(getenv("AE_GN_TAIL")) : 0;
    p.counters = tail ? counters : nullptr;
    const int ncc = C / 8;
    int rpp = 256 / ncc;
    if (rpp < 1) rpp = 1;
    int threads = ncc * rpp;
    if (threads < 64) threads = 64;
    dim3 grid(p.nchunk, B);
    hipStream_t s = (hipStream_t)stream;
    if (colstats) {  // the producers of x (and x2) already delivered the per-channel slab statistics: finalize from them, then apply
        AE_REQUIRE(HW % 32 == 0, "ae_groupnorm_nhwc_bf16: producer statistics are kept per 32-row slab: HW=%d must be a multiple of 32", HW);
        AE_REQUIRE((x2 == nullptr) == (colstats2 == nullptr), "ae_groupnorm_nhwc_bf16: statistics are needed for both sources of a concat input");
        AE_REQUIRE(((uintptr_t)colstats & 7) == 0 && ((uintptr_t)colstats2 & 7) == 0, "ae_groupnorm_nhwc_bf16: colstats alignment");
        // gn_finalize_cs_kernel finds (slab, channel) from a flat index by a float reciprocal: exact below 2^20 terms per group
        AE_REQUIRE((long)(HW / 32) * (C / groups) < (1L << 20), "ae_groupnorm_nhwc_bf16: %ld slab sums per group exceed the 2^20 the statistics fold indexes exactly", (long)(HW / 32) * (C / groups));
        p.cs1 = colstats; p.cs2 = colstats2;
        // lab knob (timing only, WRONG results): AE_GN_LAB_SKIP_FINALIZE=1 leaves this launch out — the upper bound of what ANY scheme that removes the finalize
        // launch could return (round 6: measured before building one)
        static const int lab_skip = getenv("AE_GN_LAB_SKIP_FINALIZE") ? atoi(getenv("AE_GN_LAB_SKIP_FINALIZE")) : 0;
        if (!lab_skip)
        hipLaunchKernelGGL(gn_finalize_cs_kernel, dim3(groups, B), dim3(256), 0, s, p);
        int rc0 = ae_check_launch("ae_groupnorm_nhwc_bf16(finalize from producer statistics)");
        if (rc0) return rc0;
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(threads), 0, s, p);
        return ae_check_launch("ae_groupnorm_nhwc_bf16(apply)");
    }
    // small feature maps: the whole (sample, group pack) slab fits the registers of one block -> one launch, one read, one write
    static const int slab = getenv("AE_GN_SLAB") ? atoi(getenv("AE_GN_SLAB")) : 1;  // tuning knob: 0 = three-launch path everywhere (A/B)
    if (slab && HW <= 256) {  // 32x32 maps measured slower this way (few, large slabs: 22-94 us against 20-35)
        const int cpg = C / groups;
        const int gp = (cpg % 8 == 0) ? 1 : ((2 * cpg) % 8 == 0 ? 2 : ((4 * cpg) % 8 == 0 ? 4 : 0));
        // a pack must not straddle the two sources of a deferred concat piecewise: pieces are 8 channels and C1 % 8 == 0, so any pack works
        bool done = false;
        if (gp == 1 && groups % 1 == 0) done = launch_gn_slab<1>(p, cpg, s);
        else if (gp == 2 && groups % 2 == 0) done = launch_gn_slab<2>(p, cpg, s);
        else if (gp == 4 && groups % 4 == 0) done = launch_gn_slab<4>(p, cpg, s);
        if (done) return ae_check_launch("ae_groupnorm_nhwc_bf16(slab)");
    }
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(threads), (size_t)(threads / ncc) * 2 * C * sizeof(float), s, p);
    int rc = ae_check_launch("ae_groupnorm_nhwc_bf16(stats)");
    if (rc) return rc;
    // tuning knob: 1 folds the finalize into every apply block.  In-situ A/B (one box, 2 runs each, UNet batch 12): 17.77 ms per UNet
    // step fused vs 17.60 ms with the separate 1-block-per-sample finalize launch — inside the captured graph a tiny dependent launch
    // costs less than the per-block re-fold it replaces.  Default off.
    static const int fuse_env = getenv("AE_GN_FUSE") ? atoi(getenv("AE_GN_FUSE")) : 0;
    const int fuse = fuse_env && !stat_out;  // the fused apply does not write the statistics
    if (p.counters) {  // the last stats block of every sample has already written the per-channel coefficients
        hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(threads), 0, s, p);
        return ae_check_launch("ae_groupnorm_nhwc_bf16(apply)");
    }
    if (fuse) {
        hipLaunchKernelGGL(gn_apply_kernel<true>, grid, dim3(threads), 0, s, p);
        return ae_check_launch("ae_groupnorm_nhwc_bf16(finalize+apply)");
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(1024), 0, s, p);
    rc = ae_check_launch("ae_groupnorm_nhwc_bf16(finalize)");
    if (rc) return rc;
    hipLaunchKernelGGL(gn_apply_kernel<false>, grid, dim3(threads), 0, s, p);
    return ae_check_launch("ae_groupnorm_nhwc_bf16(apply)");
}

// 1 when ae_groupnorm_splitk_nhwc_bf16 covers the shape (the one-launch slab form on 256-thread blocks, at most eight 16-byte pieces per thread: maps up to 16 x 16 at 40 channels per group)
extern "C" int ae_groupnorm_splitk_supported(int B, int HW, int C, int groups, int splitk) {
    if (B <= 0 || B > 65535 || HW <= 0 || HW > 256 || C <= 0 || groups <= 0 || groups > 64 || C % groups || C % 8 || splitk < 2 || splitk > 8) return 0;
    const int cpg = C / groups;
    const int gp = (cpg % 8 == 0) ? 1 : ((2 * cpg) % 8 == 0 ? 2 : ((4 * cpg) % 8 == 0 ? 4 : 0));
    if (gp == 0 || groups % gp) return 0;
    const long total = (long)HW * (gp * cpg / 8);
    return total <= 2048 ? 1 : 0;
}

// GroupNorm(+SiLU) of x = sum_s partial[s] + bias + addvec[b] where x has not been written: `partial` = the fp32 K ranges of ae_conv3x3_partials_bf16
// ([splitk][B*HW][C]), bias [C] / addvec [B, >= C] (row stride addvec_ld) fp32 or NULL.  Bit-identical to ae_conv3x3_bf16 (its reduce launch) followed by
// ae_groupnorm_nhwc_bf16 on the small-map slab path.  openaimodel.py:262-272: h = in_conv(...) + emb_out; h = out_layers[0:2](h).
extern "C" int ae_groupnorm_splitk_nhwc_bf16(const float* partial, int splitk, const float* bias, const float* addvec, long addvec_ld, const float* gamma, const float* beta,
                                             void* y, int B, int HW, int C, int groups, float eps, int act, void* stream) {
    AE_REQUIRE(partial && gamma && beta && y, "ae_groupnorm_splitk_nhwc_bf16: null pointer");
    AE_REQUIRE(act == 0 || act == 1, "ae_groupnorm_splitk_nhwc_bf16: act must be 0 (none) or 1 (SiLU)");
    AE_REQUIRE(ae_groupnorm_splitk_supported(B, HW, C, groups, splitk), "ae_groupnorm_splitk_nhwc_bf16: unsupported shape B=%d HW=%d C=%d groups=%d splitk=%d (maps up to 256 positions, "
               "2..8 K ranges, at most 2048 16-byte pieces per group pack)", B, HW, C, groups, splitk);
    AE_REQUIRE(((uintptr_t)partial & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)bias & 15) == 0 && ((uintptr_t)addvec & 15) == 0 && ((uintptr_t)gamma & 15) == 0 &&
               ((uintptr_t)beta & 15) == 0 && (!addvec || (addvec_ld >= C && addvec_ld % 4 == 0)), "ae_groupnorm_splitk_nhwc_bf16: 16-byte alignment (addvec rows too)");
    GNArgs p{};
    p.C1 = C; p.gamma = gamma; p.beta = beta; p.y = (bf16_t*)y;
    p.B = B; p.HW = HW; p.C = C; p.groups = groups; p.act = act; p.eps = eps;
    p.sk_partial = partial; p.sk_n = splitk; p.sk_bias = bias; p.sk_addvec = addvec; p.sk_ldav = addvec_ld; p.sk_MN = (long)B * HW * C;
    const int cpg = C / groups;
    const int gp = (cpg % 8 == 0) ? 1 : ((2 * cpg) % 8 == 0 ? 2 : 4);
    hipStream_t s = (hipStream_t)stream;
    bool done = false;
    if (gp == 1) done = launch_gn_slab_splitk<1>(p, cpg, s);
    else if (gp == 2) done = launch_gn_slab_splitk<2>(p, cpg, s);
    else done = launch_gn_slab_splitk<4>(p, cpg, s);
    if (!done) { ae_set_error("ae_groupnorm_splitk_nhwc_bf16: the slab does not fit eight pieces per thread of a 256-thread block"); return AE_ERR_UNSUPPORTED; }
    return ae_check_launch("ae_groupnorm_splitk_nhwc_bf16");
}

extern "C" int ae_layernorm_bf16(const void* x, const float* gamma, const float* beta, void* y, int M, int C, float eps,
                                 void* stream) {
    AE_REQUIRE(x && gamma && beta && y, "ae_layernorm_bf16: null pointer");
    AE_REQUIRE(M > 0 && C > 0 && C % 8 == 0, "ae_layernorm_bf16: C=%d must be a positive multiple of 8", C);
    AE_REQUIRE(C <= 4096, "ae_layernorm_bf16: C=%d > 4096 unsupported", C);
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "ae_layernorm_bf16: 16-byte alignment");
    dim3 grid((M + 3) / 4), block(256);
    const int ncc = C / 8;
    hipStream_t s = (hipStream_t)stream;
    // C = 40 L channels (the UNet's 320 / 640 / 1280, SAM's 1280): several rows per wave, all lanes busy.  AE_LN_ROWS=0: round-2 kernel (A/B)
    static const int rows_kernel = getenv("AE_LN_ROWS") ? atoi(getenv("AE_LN_ROWS")) : 1;
    const bool al16 = (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
    if (rows_kernel && al16 && ncc % 5 == 0 && (ncc == 40 || ncc == 80 || ncc == 160 || ncc == 320)) {
        const int L = ncc / 5, rpb = 4 * (64 / L);
        dim3 g2((unsigned)((M + rpb - 1) / rpb));
        if (L == 8) hipLaunchKernelGGL((layernorm_rows_kernel<8, 5>), g2, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
        else if (L == 16) hipLaunchKernelGGL((layernorm_rows_kernel<16, 5>), g2, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
        else if (L == 32) hipLaunchKernelGGL((layernorm_rows_kernel<32, 5>), g2, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
        else hipLaunchKernelGGL((layernorm_rows_kernel<64, 5>), g2, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
        return ae_check_launch("ae_layernorm_bf16");
    }
    if (ncc <= 64) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
    else if (ncc <= 128) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
    else if (ncc <= 192) hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
    else hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, (const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps);
    return ae_check_launch("ae_layernorm_bf16");
}

// LayerNorm with SAM's window partition folded in (layernorm_window_kernel).  mode 1: y[windows] = partition(LayerNorm(x[image])), padding
// rows zero; mode 2: xsum[image] = bf16(x[windows] + shortcut[image]), y[image] = LayerNorm(xsum).  C = 40 L channels, L in {8, 16, 32, 64}.
extern "C" int ae_layernorm_window_supported(int C) {
    const int ncc = C / 8;
    return C % 8 == 0 && (ncc == 40 || ncc == 80 || ncc == 160 || ncc == 320);
}

extern "C" int ae_layernorm_window_bf16(const void* x, const void* shortcut, const float* gamma, const float* beta, void* y, void* xsum,
                                        int B, int H, int W, int C, int ws, int mode, float eps, void* stream) {
    AE_REQUIRE(x && gamma && beta && y && (mode == 1 || (mode == 2 && shortcut && xsum)), "ae_layernorm_window_bf16: null pointer / mode %d", mode);
    AE_REQUIRE(B > 0 && H > 0 && W > 0 && ws > 0 && ae_layernorm_window_supported(C), "ae_layernorm_window_bf16: unsupported shape (C=%d)", C);
    AE_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)shortcut | (uintptr_t)xsum | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0,
               "ae_layernorm_window_bf16: 16-byte alignment");
    LnWinArgs p{};
    p.x = (const bf16_t*)x; p.shortcut = (const bf16_t*)shortcut; p.gamma = gamma; p.beta = beta; p.y = (bf16_t*)y; p.xsum = (bf16_t*)xsum;
    p.B = B; p.H = H; p.W = W; p.C = C; p.ws = ws; p.nH = (H + ws - 1) / ws; p.nW = (W + ws - 1) / ws; p.eps = eps;
    p.rows = mode == 1 ? (long)B * p.nH * p.nW * ws * ws : (long)B * H * W;
    const int L = C / 40, rpb = 4 * (64 / L);
    dim3 grid((unsigned)((p.rows + rpb - 1) / rpb)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define AE_LNW(LL) \
    if (mode == 1) hipLaunchKernelGGL((layernorm_window_kernel<LL, 5, 1>), grid, block, 0, s, p); \
    else hipLaunchKernelGGL((layernorm_window_kernel<LL, 5, 2>), grid, block, 0, s, p)
    if (L == 8) { AE_LNW(8); } else if (L == 16) { AE_LNW(16); } else if (L == 32) { AE_LNW(32); } else { AE_LNW(64); }
#undef AE_LNW
    return ae_check_launch("ae_layernorm_window_bf16");
}

extern "C" long ae_groupnorm_bwd_workspace_floats(int B, int HW, int C, int groups) {
    const int rpc = ae_groupnorm_rows_per_chunk(HW, C);
    const int nchunk = (HW + rpc - 1) / rpc;
    const long partials = (((long)B * nchunk * groups * 2 + 3) / 4) * 4;
    return 2 * partials + 2 * (long)B * 2 * C + (((long)B * groups * 2 + 3) / 4) * 4;
}

extern "C" int ae_groupnorm_bwd_nhwc_bf16(const void* x, const void* x2, int C1, const float* gamma, const float* beta, const void* dy,
                                          void* dx, void* dx2, int B, int HW, int C, int groups, float eps, int act,
                                          float* workspace, int* counters, const float* stat_in, int accumulate, void* stream) {
    AE_REQUIRE(x && gamma && beta && dy && dx && workspace, "ae_groupnorm_bwd_nhwc_bf16: null pointer");
    AE_REQUIRE(accumulate >= 0 && accumulate <= 3 && (x2 || !(accumulate & 2)), "ae_groupnorm_bwd_nhwc_bf16: accumulate %d (bit 0: dx +=, bit 1: dx2 +=)", accumulate);
    AE_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0, "ae_groupnorm_bwd_nhwc_bf16: bad shape C=%d groups=%d", C, groups);
    AE_REQUIRE(C % 8 == 0 && C <= 8192 && groups <= 64 && B <= 65535, "ae_groupnorm_bwd_nhwc_bf16: unsupported size");
    AE_REQUIRE(act == 0 || act == 1, "ae_groupnorm_bwd_nhwc_bf16: act must be 0 (none) or 1 (SiLU)");
    if (x2) AE_REQUIRE(dx2 && C1 > 0 && C1 < C && C1 % 8 == 0, "ae_groupnorm_bwd_nhwc_bf16: bad concat split C1=%d C=%d", C1, C);
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)x2 & 15) == 0 &&
                   ((uintptr_t)dx2 & 15) == 0 && ((uintptr_t)workspace & 15) == 0,
               "ae_groupnorm_bwd_nhwc_bf16: 16-byte alignment");
    GNArgs p{};
    p.x = (const bf16_t*)x; p.x2 = (const bf16_t*)x2; p.C1 = x2 ? C1 : C;
    p.gamma = gamma; p.beta = beta; p.y = nullptr;
    p.dy = (const bf16_t*)dy; p.dx = (bf16_t*)dx; p.dx2 = (bf16_t*)dx2; p.accum = accumulate;
    p.B = B; p.HW = HW; p.C = C; p.groups = groups; p.act = act; p.eps = eps;
    p.rows_per_chunk = ae_groupnorm_rows_per_chunk(HW, C);
    p.nchunk = (HW + p.rows_per_chunk - 1) / p.rows_per_chunk;
    const long partials = (((long)B * p.nchunk * groups * 2 + 3) / 4) * 4;
    p.part = workspace;
    p.part2 = workspace + partials;
    p.coef = workspace + 2 * partials;
    p.coef2 = p.coef + (long)B * 2 * C;
    p.stat = p.coef2 + (long)B * 2 * C;
    static const int tail = getenv("AE_GN_TAIL") ? atoi(getenv("AE_GN_TAIL")) : 0;  // see ae_groupnorm_nhwc_bf16: default off
    p.counters = tail ? counters : nullptr;
    const int ncc = C / 8;
    int rpp = 256 / ncc;
    if (rpp < 1) rpp = 1;
    int threads = ncc * rpp;
    if (threads < 64) threads = 64;
    dim3 grid(p.nchunk, B);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(threads / ncc) * 2 * C * sizeof(float);
    int rc = 0;
    if (stat_in) {  // (mean, rstd) saved by the forward launch: no second statistics pass over x, coefficients rebuilt per thread
        p.stat = const_cast<float*>(stat_in);
        p.coef = nullptr;
    } else {
        hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(threads), lds, s, p);
        rc = ae_check_launch("ae_groupnorm_bwd_nhwc_bf16(stats)");
        if (rc) return rc;
        if (!p.counters) {
            hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(1024), 0, s, p);
            rc = ae_check_launch("ae_groupnorm_bwd_nhwc_bf16(finalize)");
            if (rc) return rc;
        }
    }
    // small feature maps with the forward's statistics at hand: one launch (gnb_slab_kernel).  AE_GN_BWD_SLAB=0: three launches everywhere (A/B).
    static const int bslab = getenv("AE_GN_BWD_SLAB") ? atoi(getenv("AE_GN_BWD_SLAB")) : 1;
    if (bslab && stat_in && HW <= 256 && !p.counters) {
        const int cpg = C / groups;
        const int gp = (cpg % 8 == 0) ? 1 : ((2 * cpg) % 8 == 0 ? 2 : ((4 * cpg) % 8 == 0 ? 4 : 0));
        bool done = false;
        if (gp == 1) done = launch_gnb_slab<1>(p, cpg, s);
        else if (gp == 2 && groups % 2 == 0) done = launch_gnb_slab<2>(p, cpg, s);
        else if (gp == 4 && groups % 4 == 0) done = launch_gnb_slab<4>(p, cpg, s);
        if (done) return ae_check_launch("ae_groupnorm_bwd_nhwc_bf16(slab)");
    }
    hipLaunchKernelGGL(gnb_partial_kernel, grid, dim3(threads), lds, s, p);
    rc = ae_check_launch("ae_groupnorm_bwd_nhwc_bf16(partial)");
    if (rc) return rc;
    if (!p.counters) {
        hipLaunchKernelGGL(gnb_finalize_kernel, dim3(B), dim3(1024), 0, s, p);
        rc = ae_check_launch("ae_groupnorm_bwd_nhwc_bf16(finalize2)");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(gnb_apply_kernel, grid, dim3(threads), 0, s, p);
    return ae_check_launch("ae_groupnorm_bwd_nhwc_bf16(apply)");
}

extern "C" int ae_layernorm_act_bf16(const void* x, const float* gamma, const float* beta, void* y, long M, int C, float eps, int act,
                                     void* stream) {
    AE_REQUIRE(x && gamma && beta && y, "ae_layernorm_act_bf16: null pointer");
    AE_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 512, "ae_layernorm_act_bf16: C=%d must be a multiple of 8 and <= 512", C);
    AE_REQUIRE(act == 0 || act == 1, "ae_layernorm_act_bf16: act %d (0 = none, 1 = GELU)", act);
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "ae_layernorm_act_bf16: 16-byte alignment");
    if (act) launch_layernorm_narrow<1>((const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps, (hipStream_t)stream);
    else launch_layernorm_narrow<0>((const bf16_t*)x, gamma, beta, (bf16_t*)y, M, C, eps, (hipStream_t)stream);
    return ae_check_launch("ae_layernorm_act_bf16");
}

extern "C" int ae_layernorm_bwd_bf16(const void* x, const float* gamma, const void* dy, void* dx, float* row_stat, int M, int C,
                                     float eps, int accumulate, void* stream) {
    AE_REQUIRE(x && gamma && dy && dx, "ae_layernorm_bwd_bf16: null pointer");
    AE_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "ae_layernorm_bwd_bf16: C=%d must be a multiple of 8 and <= 2048", C);
    AE_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0, "ae_layernorm_bwd_bf16: 16-byte alignment");
    dim3 grid((M + 3) / 4), block(256);
    const int ncc = C / 8;
    hipStream_t s = (hipStream_t)stream;
    if (ncc <= 64) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, block, 0, s, (const bf16_t*)x, gamma, (const bf16_t*)dy, (bf16_t*)dx, row_stat, M, C, eps, accumulate);
    else if (ncc <= 128) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, block, 0, s, (const bf16_t*)x, gamma, (const bf16_t*)dy, (bf16_t*)dx, row_stat, M, C, eps, accumulate);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, block, 0, s, (const bf16_t*)x, gamma, (const bf16_t*)dy, (bf16_t*)dx, row_stat, M, C, eps, accumulate);
    return ae_check_launch("ae_layernorm_bwd_bf16");
}

extern "C" int ae_layernorm_param_grad_f32(const void* x, const void* dy, const float* row_stat, float* dgamma, float* dbeta, int M,
                                           int C, void* stream) {
    AE_REQUIRE(x && dy && row_stat && dgamma && dbeta, "ae_layernorm_param_grad_f32: null pointer");
    AE_REQUIRE(M > 0 && M <= 4096 && C > 0, "ae_layernorm_param_grad_f32: M=%d must be in [1, 4096] (small-batch parameter gradients only)", M);
    hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (const bf16_t*)dy, row_stat, dgamma, dbeta, M, C);
    return ae_check_launch("ae_layernorm_param_grad_f32");
}
