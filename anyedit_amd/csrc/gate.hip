// AnySD task router gate for gfx950 — latency-bound, one wave per sample.
//
// the reference's AnySD package (AnySD.model.MoE, train.py:25-28, 420-424, 694-695) is an empty,
// un-pinned submodule, so this kernel implements OUR documented spec (DESIGN.md "AnySD task router"):
//   logits[b, e] = <task_emb[edit_code[b]], Wg[e]> + bg[e];  probs = softmax_e(logits);
//   top1[b] = argmax_e probs (lowest index on ties);  top1_prob[b] = probs[b, top1[b]].
#include "common.hpp"

namespace {
__global__ __launch_bounds__(64) void task_gate_kernel(const float* task_emb, const long* edit_code, const float* Wg, const float* bg,
                                                       int n_tasks, int Dt, int E, float* probs, int* top1, float* top1_prob) {
    const int b = blockIdx.x, e = threadIdx.x;
    long code = edit_code[b];
    if (code < 0) code = 0;
    if (code >= n_tasks) code = n_tasks - 1;
    const float* te = task_emb + code * Dt;
    float logit = -INFINITY;
    if (e < E) {
        float acc = 0.f;
        for (int i = 0; i < Dt; ++i) acc += te[i] * Wg[(long)e * Dt + i];
        logit = acc + (bg ? bg[e] : 0.f);
    }
    const float mx = wave_reduce_max(logit);
    const float ex = e < E ? expf(logit - mx) : 0.f;
    const float sum = wave_reduce_sum(ex);
    const float pr = ex / sum;
    if (e < E && probs) probs[(long)b * E + e] = pr;
    // argmax with lowest-index tie-break
    float best = pr;
    int idx = e < E ? e : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if (e == 0) {
        if (top1) top1[b] = idx;
        if (top1_prob) top1_prob[b] = best;
    }
}
}  // namespace

extern "C" int ae_task_gate(const float* task_emb, const long* edit_code, const float* Wg, const float* bg, int B, int n_tasks, int Dt,
                            int E, float* probs, int* top1, float* top1_prob, void* stream) {
    AE_REQUIRE(task_emb && edit_code && Wg, "ae_task_gate: null pointer");
    AE_REQUIRE(B > 0 && n_tasks > 0 && Dt > 0 && E > 0 && E <= 64, "ae_task_gate: bad sizes (E must be <= 64, got %d)", E);
    hipLaunchKernelGGL(task_gate_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, task_emb, edit_code, Wg, bg, n_tasks, Dt, E, probs,
                       top1, top1_prob);
    return ae_check_launch("ae_task_gate");
}
