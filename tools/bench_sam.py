#!/usr/bin/env python3
"""SAM ViT-H image-encoder latency on MI355X (row A10; BASELINE.md §2: 5.96 TFLOP per 1024^2 image).
    python tools/bench_sam.py [--batch 1] [--iters 5]
Random-init weights of the build_sam_vit_h geometry (no checkpoints exist on the box), synthetic normalised image."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd import ops  # noqa: E402
from anyedit_amd.segment_anything.modeling.image_encoder import build_sam_vit_h_encoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=5)
    # full-size parity against the oracle lives in the GPU suite: tests/test_hip_bench_shapes.py::test_sam_vit_h_full_encoder_vs_oracle
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    with torch.device(dev):
        enc = build_sam_vit_h_encoder()
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "rel_pos" in n or "pos_embed" in n:
                p.normal_(0, 0.02)
    enc.eval().requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    x = ((torch.rand(a.batch, 3, 1024, 1024, generator=g) * 255 - 120.0) / 58.0).to(dev)
    with torch.no_grad():
        y = enc(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            y = enc(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        with ops.OpProfiler() as prof:
            enc(x)
        enc.use_hip_graph = True
        yg = enc(x)
        torch.cuda.synchronize()
        assert torch.equal(yg, y), "graph replay must reproduce the eager launches bit for bit"
        tg = []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            yg = enc(x)
            torch.cuda.synchronize()
            tg.append(time.perf_counter() - t0)
        tg.sort()
        enc.use_hip_graph = False
        summ = prof.summary()
        by_shape = prof.summary(by_shape=True)
    ts.sort()
    med = ts[len(ts) // 2]
    assert y.shape == (a.batch, 256, 64, 64) and torch.isfinite(y).all()
    out = {"what": "SAM ViT-H image encoder, 1024x1024", "batch": a.batch, "latency_ms_p50": 1e3 * med,
           "latency_ms_p50_hip_graph": 1e3 * tg[len(tg) // 2], "images_per_s": a.batch / tg[len(tg) // 2],
           "tflops": 5.96 * a.batch / tg[len(tg) // 2], "tflops_eager": 5.96 * a.batch / med,
           "kernels": {k: {"calls": v["calls"], "ms": v["ms"], "tflops": v["tflops"], "gbps": v["gbps"]}
                       for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])},
           "by_shape": {k: {"calls": v["calls"], "avg_us": v["avg_us"], "tflops": v["tflops"]}
                        for k, v in sorted(by_shape.items(), key=lambda kv: -kv[1]["ms"])[:12]}}
    # ---- configs[4]: the same encoder with e4m3 attention operands in the global-attention blocks (ae_attn_fwd_fp8)
    from anyedit_amd.segment_anything.modeling.image_encoder import Attention
    with torch.no_grad():
        for m in enc.modules():
            if isinstance(m, Attention):
                m.attn_fp8 = True
        y8 = enc(x)
        torch.cuda.synchronize()
        t8 = []
        for _ in range(a.iters):
            t0 = time.perf_counter()
            y8 = enc(x)
            torch.cuda.synchronize()
            t8.append(time.perf_counter() - t0)
        t8.sort()
        with ops.OpProfiler() as prof8:
            enc(x)
        for m in enc.modules():
            if isinstance(m, Attention):
                m.attn_fp8 = False
    d8 = float((y8.float() - y.float()).norm() / y.float().norm())
    out["fp8_attention"] = {"what": "global-attention blocks (4 of 32) with e4m3 q/k/v/p (ae_attn_fwd_fp8); windowed blocks stay bf16",
                            "latency_ms_p50_eager": 1e3 * t8[len(t8) // 2], "encoder_output_rel_l2_vs_bf16_attention": d8,
                            "attention_by_shape": {k: {"calls": v["calls"], "avg_us": v["avg_us"], "tflops": v["tflops"]}
                                                   for k, v in prof8.summary(by_shape=True).items() if "attn" in k}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
