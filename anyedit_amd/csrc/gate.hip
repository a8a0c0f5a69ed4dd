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
// Backward of g_b = softmax(Wg te_b + bg)[top1_b] w.r.t. the task embedding row (argmax is a constant of the step):
//   dlogit[e] = dg_b * g_b * ((e == top1_b) - probs[b, e]);   dte_b = Wg^T dlogit;
//   router parameters (optional outputs): dWg[e, :] = sum_b dlogit[b, e] te_b,  dbg[e] = sum_b dlogit[b, e]  (fixed summation order)
__global__ __launch_bounds__(256) void task_gate_bwd_kernel(const float* probs, const int* top1, const float* dgate, const float* Wg,
                                                            int Dt, int E, float* dte) {
    const int b = blockIdx.x;
    const int t1 = top1[b];
    const float gb = probs[(long)b * E + t1], dg = dgate[b];
    for (int i = threadIdx.x; i < Dt; i += 256) {
        float acc = 0.f;
        for (int e = 0; e < E; ++e) acc += dg * gb * ((e == t1 ? 1.0f : 0.0f) - probs[(long)b * E + e]) * Wg[(long)e * Dt + i];
        dte[(long)b * Dt + i] = acc;
    }
}
// one block per expert row of Wg; samples are summed in index order (deterministic)
__global__ __launch_bounds__(256) void task_gate_wgrad_kernel(const float* probs, const int* top1, const float* dgate, const float* task_emb,
                                                              const long* edit_code, int B, int n_tasks, int Dt, int E, float* dWg, float* dbg) {
    const int e = blockIdx.x;
    for (int i = threadIdx.x; i < Dt; i += 256) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const int t1 = top1[b];
            const float dl = dgate[b] * probs[(long)b * E + t1] * ((e == t1 ? 1.0f : 0.0f) - probs[(long)b * E + e]);
            long code = edit_code[b];
            code = code < 0 ? 0 : (code >= n_tasks ? n_tasks - 1 : code);
            acc += dl * task_emb[code * Dt + i];
        }
        dWg[(long)e * Dt + i] = acc;
    }
    if (threadIdx.x == 0 && dbg) {
        float acc = 0.f;
        for (int b = 0; b < B; ++b) {
            const int t1 = top1[b];
            acc += dgate[b] * probs[(long)b * E + t1] * ((e == t1 ? 1.0f : 0.0f) - probs[(long)b * E + e]);
        }
        dbg[e] = acc;
    }
}
}  // namespace

extern "C" int ae_task_gate_wgrad(const float* probs, const int* top1, const float* dgate, const float* task_emb, const long* edit_code,
                                  int B, int n_tasks, int Dt, int E, float* dWg, float* dbg, void* stream) {
    AE_REQUIRE(probs && top1 && dgate && task_emb && edit_code && dWg, "ae_task_gate_wgrad: null pointer");
    AE_REQUIRE(B > 0 && n_tasks > 0 && Dt > 0 && E > 0 && E <= 64, "ae_task_gate_wgrad: bad sizes");
    hipLaunchKernelGGL(task_gate_wgrad_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, probs, top1, dgate, task_emb, edit_code, B, n_tasks,
                       Dt, E, dWg, dbg);
    return ae_check_launch("ae_task_gate_wgrad");
}

extern "C" int ae_task_gate_bwd(const float* probs, const int* top1, const float* dgate, const float* Wg, int B, int Dt, int E,
                                float* dte, void* stream) {
    AE_REQUIRE(probs && top1 && dgate && Wg && dte, "ae_task_gate_bwd: null pointer");
    AE_REQUIRE(B > 0 && Dt > 0 && E > 0 && E <= 64, "ae_task_gate_bwd: bad sizes");
    hipLaunchKernelGGL(task_gate_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, probs, top1, dgate, Wg, Dt, E, dte);
    return ae_check_launch("ae_task_gate_bwd");
}

extern "C" int ae_task_gate(const float* task_emb, const long* edit_code, const float* Wg, const float* bg, int B, int n_tasks, int Dt,
                            int E, float* probs, int* top1, float* top1_prob, void* stream) {
    AE_REQUIRE(task_emb && edit_code && Wg, "ae_task_gate: null pointer");
    AE_REQUIRE(B > 0 && n_tasks > 0 && Dt > 0 && E > 0 && E <= 64, "ae_task_gate: bad sizes (E must be <= 64, got %d)", E);
    hipLaunchKernelGGL(task_gate_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, task_emb, edit_code, Wg, bg, n_tasks, Dt, E, probs,
                       top1, top1_prob);
    return ae_check_launch("ae_task_gate");
}
