#!/usr/bin/env python3
"""First-stage (kl-f8 AutoencoderKL) decode / encode latency on MI355X — SURVEY.md §8(f) N1: the step on either side of the
denoising loop (2.48 TFLOP decode + 1.08 TFLOP encode per 512x512 image).  Random-init weights of the SD kl-f8 geometry.
    python tools/bench_vae.py [--batch 4] [--iters 5]        (parity: tests/test_hip_bench_shapes.py::test_vae_kl_f8_full_size_vs_oracle)
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from anyedit_amd.ldm.models.autoencoder import AutoencoderKL  # noqa: E402

KL_F8 = dict(double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4], num_res_blocks=2,
             attn_resolutions=[], dropout=0.0)


def med(fn, iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    with torch.device(dev):
        vae = AutoencoderKL(ddconfig=KL_F8, embed_dim=4)
    vae.eval().requires_grad_(False)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(a.batch, 4, 64, 64, generator=g).to(dev)
    x = (torch.rand(a.batch, 3, 512, 512, generator=g) * 2 - 1).to(dev)
    t_dec, y = med(lambda: vae.decode(z), a.iters)
    t_enc, post = med(lambda: vae.encode(x).mean, a.iters)
    assert y.shape == (a.batch, 3, 512, 512) and torch.isfinite(y).all() and post.shape == (a.batch, 4, 64, 64)
    out = {"what": "kl-f8 AutoencoderKL, 512x512", "batch": a.batch, "decode_ms": 1e3 * t_dec, "encode_ms": 1e3 * t_enc,
           "decode_tflops": 2.48 * a.batch / t_dec, "encode_tflops": 1.08 * a.batch / t_enc,
           "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
