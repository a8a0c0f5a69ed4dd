"""Fused cross-attention half (ae_xattn_fused_bf16) against the three launches it replaces, at the UNet's 64x64 level (B x 4096 rows, C = 320, 78 + 4 keys).

    python tools/xattn_fused_lab.py [--batch 12] [--iters 200]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402

BF, DEV = torch.bfloat16, "cuda"


def timed(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    B, N, C, H, D, Nk, T = a.batch, 4096, 320, 8, 40, 78, 4
    M = B * N
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, C, generator=g) * 1.3).to(DEV, BF)
    wq, wo = torch.randn(C, C, generator=g) / C ** 0.5, torch.randn(C, C, generator=g) / C ** 0.5
    bo = (0.1 * torch.randn(C, generator=g)).to(DEV)
    gamma, beta = (1.0 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.2 * torch.randn(C, generator=g)).to(DEV)
    kv, kv_ip = torch.randn(B * Nk, 2 * C, generator=g).to(DEV, BF), torch.randn(B * T, 2 * C, generator=g).to(DEV, BF)
    gate = torch.rand(B, generator=g).to(DEV)
    scale = D ** -0.5
    wq_img, wo_img = ops.pack_xattn_wq(wq.to(DEV)), ops.pack_xattn_wo(wo.to(DEV))
    kv_img = ops.pack_xattn_kv(kv, kv_ip, B, Nk, T)
    wqd, wod = wq.to(DEV, BF), wo.to(DEV, BF)
    wq_f, s_f, c_f = ops.pack_ln_fold(wq.to(DEV), None, gamma, beta)
    st = ops.rowstats_buffer(M, C, DEV)
    xs = x.float().reshape(M, C // 64, 64)
    st[..., 0], st[..., 1] = xs.sum(-1), (xs * xs).sum(-1)
    y, y3 = torch.empty(M, C, dtype=BF, device=DEV), torch.empty(M, C, dtype=BF, device=DEV)
    qb, ob = torch.empty(M, C, dtype=BF, device=DEV), torch.empty(B, N, C, dtype=BF, device=DEV)
    st3 = ops.rowstats_buffer(M, C, DEV)
    qs, ks, ksi = (N * C, D, C), (Nk * 2 * C, D, 2 * C), (T * 2 * C, D, 2 * C)

    def fused():
        ops.xattn_fused(x, gamma, beta, 1e-5, wq_img, kv_img, gate, wo_img, bo, N, Nk, T, scale, out=y)

    def three():
        ops.gemm_ln(x, st, wq_f, s_f, c_f, 1e-5, out=qb)
        ops.attention(qb, kv, kv[:, C:], B, H, N, Nk, D, scale, qs, ks, ks, seg2=(kv_ip, kv_ip[:, C:], T, ksi, ksi, gate), out=ob)
        ops.gemm(ob.reshape(M, C), wod, bias=bo, residual=x, out=y3, rowstats=st3)

    res = {"M": M, "Nk": Nk, "T": T, "iters": a.iters}
    if not ops.xattn_fused_ok(M, C, H, D, N, Nk, T):
        print(json.dumps({"error": "fused kernel does not cover this shape", **res}))
        return
    fused(); three(); torch.cuda.synchronize()
    res["rel_l2_fused_vs_three_launches"] = float((y.float() - y3.float()).norm() / y3.float().norm())
    first = y.clone()
    fused(); torch.cuda.synchronize()
    res["bit_equal_repeat"] = bool(torch.equal(y, first))
    flop = 2.0 * M * C * (2 * C + 2 * (Nk + T))
    t = {"fused": [], "three": []}
    for _ in range(a.rounds):
        for name, fn in (("fused", fused), ("three", three)):
            timed(fn, 20)
            t[name].append(timed(fn, a.iters))
    for name, v in t.items():
        res[name + "_us"] = [round(u, 2) for u in v]
        res[name + "_tflops"] = round(flop / (min(v) * 1e-6) / 1e12, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
