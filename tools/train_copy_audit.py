#!/usr/bin/env python3
"""Where do the torch-native copy / fill kernels of a training step come from?  Runs tools/bench_train.py's step with torch.profiler (CUDA activity,
Python stacks) and prints, per (kernel family, innermost anyedit_amd source line), the launches and device time of ONE steady-state step.

    python tools/train_copy_audit.py [--batch 4]
"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    from anyedit_amd.anysd.train import AnySDTrainer
    unet, moe, sched = bench.build_model(device)
    for p in list(moe.image_proj_model.parameters()) + list(moe.adapter_modules) + [moe.task_embs]:
        p.requires_grad_(True)
    B = a.batch
    g = torch.Generator(device="cpu").manual_seed(4)
    lat = torch.randn(B, 4, 64, 64, generator=g).to(device)
    img = (torch.randn(B, 4, 64, 64, generator=g) * 0.18215).to(device)
    ehs = torch.randn(B, 77, 768, generator=g).to(device)
    null = torch.randn(1, 77, 768, generator=g).to(device)
    ref = torch.randn(B, 257, 1280, generator=g).to(device)
    code = (torch.arange(B) % 3).to(device)
    tr = AnySDTrainer(moe, sched.sqrt_alphas_cumprod, sched.sqrt_one_minus_alphas_cumprod, lr=1e-5)
    noise, t, u = torch.randn(B, 4, 64, 64, generator=g).to(device), torch.randint(0, 1000, (B,), generator=g).to(device), torch.rand(B, generator=g).to(device)

    def step():
        return tr.train_step(lat, img, ehs, ref, code, noise, t, null_ehs=null.expand(B, -1, -1), dropout_u=u, dropout_p=0.05)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        # CPU-side aten ops that launched device kernels: keep those whose kernels are torch-native (our kernels launch through ctypes, no aten op)
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.kernels:
            continue
        dev_us = sum(k.duration for k in ev.kernels)
        where = "?"
        for fr in ev.stack:
            if "anyedit_amd" in fr or "/tools/" in fr or "bench" in fr:
                where = fr.strip().replace(ROOT + "/", "")
                break
        agg[(ev.name, where)][0] += len(ev.kernels)
        agg[(ev.name, where)][1] += dev_us
    tot = sum(v[1] for v in agg.values())
    print(f"torch-native device time in one step: {tot / 1e3:.3f} ms, {sum(v[0] for v in agg.values())} launches")
    for (name, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{us:9.1f} us {n:4d}x  {name:28s} {where}")


if __name__ == "__main__":
    main()
