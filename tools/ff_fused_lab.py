"""Fused feed-forward (ae_ff_fused_bf16) against the two launches it replaces, at the UNet's 64x64 level (M = B * 4096 rows, C = 320, H = 1280).

    python tools/ff_fused_lab.py [--batch 12] [--iters 200]

Prints per-call time (HIP events on the launch stream, alternating blocks of launches so both see the same clock state), TFLOP/s over the
3 * 2 * M * C * H algorithmic FLOP, the parity of both against the fp64 module arithmetic, and the bit-equality of repeated launches.
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

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
    M, C, H = a.batch * 4096, 320, 1280
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, C, generator=g) * 1.3).to(BF)
    w1, b1 = torch.randn(2 * H, C, generator=g) / C ** 0.5, 0.1 * torch.randn(2 * H, generator=g)
    w2, b2 = torch.randn(C, H, generator=g) / H ** 0.5, 0.1 * torch.randn(C, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    xd = x.to(DEV)
    w1p, b1p = ops.pack_geglu(w1.to(DEV), b1.to(DEV))
    w2img, w2d = ops.pack_ff2_fused(w2.to(DEV)), w2.to(DEV, BF)
    gd, bd, b2d = gamma.to(DEV), beta.to(DEV), b2.to(DEV)
    y = torch.empty(M, C, dtype=BF, device=DEV)
    hbuf = torch.empty(M, H, dtype=BF, device=DEV)
    y2 = torch.empty(M, C, dtype=BF, device=DEV)
    # the round-5 path with the LayerNorm fold needs the producer's statistics; its prologue-LayerNorm twin is what ln_gemm launches
    wq, s, c = ops.pack_ln_fold(w1.to(DEV), b1.to(DEV), gd, bd, geglu=True)
    st = ops.rowstats_buffer(M, C, DEV)
    xs = xd.float().reshape(M, C // 64, 64)
    st[..., 0], st[..., 1] = xs.sum(-1), (xs * xs).sum(-1)

    def fused():
        ops.ff_fused(xd, gd, bd, 1e-5, w1p, b1p, w2img, b2d, residual=xd, out=y)

    def two_fold():
        ops.gemm_ln(xd, st, wq, s, c, 1e-5, epilogue=ops.EPI_GEGLU, out=hbuf)
        ops.gemm(hbuf, w2d, bias=b2d, residual=xd, out=y2)

    def two_prologue():
        ops.ln_gemm(xd, gd, bd, 1e-5, w1p, b1p, epilogue=ops.EPI_GEGLU, out=hbuf)
        ops.gemm(hbuf, w2d, bias=b2d, residual=xd, out=y2)

    w3 = (torch.randn(C, C, generator=g) / C ** 0.5).to(DEV, BF)
    b3 = (0.1 * torch.randn(C, generator=g)).to(DEV)
    r3 = torch.randn(M, C, generator=g).to(DEV, BF)
    y3, y4 = torch.empty(M, C, dtype=BF, device=DEV), torch.empty(M, C, dtype=BF, device=DEV)

    def fused_tail():
        ops.ff_fused(xd, gd, bd, 1e-5, w1p, b1p, w2img, b2d, residual=xd, out=y3, w3=w3, b3=b3, residual3=r3)

    def three_fold():
        two_fold()
        ops.gemm(y2, w3, bias=b3, residual=r3, out=y4)

    res = {"M": M, "C": C, "H": H, "iters": a.iters}
    if not ops.ff_fused_ok(M, C, H):
        print(json.dumps({"error": "fused kernel does not cover this shape", **res}))
        return
    fused(); two_fold(); torch.cuda.synchronize()
    # parity on a sample of rows (fp64 module arithmetic on the bf16 inputs)
    idx = torch.randperm(M, generator=g)[:2048]
    xs_ = x[idx].double()
    z = F.layer_norm(xs_, (C,), gamma.double(), beta.double(), 1e-5) @ w1.to(BF).double().t() + b1.double()
    aa, gg = z.chunk(2, dim=-1)
    ref = ((aa * F.gelu(gg)) @ w2.to(BF).double().t() + b2.double() + xs_).float()
    rl2 = lambda t: float((t - ref).norm() / ref.norm())  # noqa: E731
    res["rel_l2_fused"] = rl2(y[idx.to(DEV)].float().cpu())
    res["rel_l2_two_launch_fold"] = rl2(y2[idx.to(DEV)].float().cpu())
    y_first = y.clone()
    fused(); torch.cuda.synchronize()
    res["bit_equal_repeat"] = bool(torch.equal(y, y_first))
    flop = 3 * 2.0 * M * C * H
    fused_tail(); three_fold(); torch.cuda.synchronize()
    res["rel_l2_tail_vs_three_launches"] = float((y3.float() - y4.float()).norm() / y4.float().norm())
    t = {"fused": [], "two_fold": [], "two_prologue": [], "fused_tail": [], "three_fold": []}
    for _ in range(a.rounds):
        for name, fn in (("fused", fused), ("two_fold", two_fold), ("two_prologue", two_prologue), ("fused_tail", fused_tail), ("three_fold", three_fold)):
            timed(fn, 20)
            t[name].append(timed(fn, a.iters))
    for name, v in t.items():
        res[name + "_us"] = [round(u, 2) for u in v]
        res[name + "_tflops"] = round((flop + (2.0 * M * C * C if name in ("fused_tail", "three_fold") else 0.0)) / (min(v) * 1e-6) / 1e12, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
