// Argument block shared by the attention forward kernels (attention.hip: general kernel with bias / mask / second segment /
// log-sum-exp outputs; attention_fast.hip: the long-sequence self-attention kernel) and their dispatcher.
#pragma once
#include "common.hpp"

struct AttnArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int B, H, Nq, Nk;
    long q_sb, q_sh, q_sn, k_sb, k_sh, k_sn, v_sb, v_sh, v_sn, o_sb, o_sh, o_sn;
    float scale;
    const float* rel_h; const float* rel_w; int kH, kW;  // optional decomposed bias, fp32 [B*H, Nq, kH|kW]
    const uint8_t* key_mask;                               // optional [B, Nk], 0 = masked
    const float* out_scale;                                // optional [B]: out = (accum ? out : 0) + out_scale[b] * result
    int accum;
    // optional second key/value segment with its OWN softmax (decoupled adapter attention fused into the same launch):
    // out = Attn(q,K,V) + scale2[b] * Attn(q,K2,V2)   — Q is read once, O is written once
    const bf16_t* k2; const bf16_t* v2; int Nk2;
    long k2_sb, k2_sh, k2_sn, v2_sb, v2_sh, v2_sn;
    const float* scale2;
    // optional [B, H, Nq] fp32 outputs for the backward pass: log2-domain log-sum-exp of each segment's softmax
    float* lse; float* lse2;
    int ng = 1;  // attention_fast.hip, short-K/V variant: 128-query groups per block (filled in by its launcher)
};

// attention_fast.hip: returns AE_OK when it launched, AE_ERR_UNSUPPORTED when the shape / options are outside its envelope
// (the caller then uses the general kernel), another error code on a failed launch.
int ae_attn_fast_launch(const AttnArgs& a, int D, hipStream_t stream);
