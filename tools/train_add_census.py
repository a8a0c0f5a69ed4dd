#!/usr/bin/env python3
"""Which backward closures of one training step add into a gradient that already exists (one ae_add_bf16 launch each)?"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    from anyedit_amd.anysd.train import AnySDTrainer
    from anyedit_amd import autodiff
    unet, moe, sched = bench.build_model(dev)
    for p in list(moe.image_proj_model.parameters()) + list(moe.adapter_modules) + [moe.task_embs]:
        p.requires_grad_(True)
    B = 4
    g = torch.Generator(device="cpu").manual_seed(4)
    lat = torch.randn(B, 4, 64, 64, generator=g).to(dev)
    img = (torch.randn(B, 4, 64, 64, generator=g) * 0.18215).to(dev)
    ehs = torch.randn(B, 77, 768, generator=g).to(dev)
    null = torch.randn(1, 77, 768, generator=g).to(dev)
    ref = torch.randn(B, 257, 1280, generator=g).to(dev)
    code = (torch.arange(B) % 3).to(dev)
    tr = AnySDTrainer(moe, sched.sqrt_alphas_cumprod, sched.sqrt_one_minus_alphas_cumprod, lr=1e-5)
    noise, t, u = torch.randn(B, 4, 64, 64, generator=g).to(dev), torch.randint(0, 1000, (B,), generator=g).to(dev), torch.rand(B, generator=g).to(dev)
    step = lambda: tr.train_step(lat, img, ehs, ref, code, noise, t, null_ehs=null.expand(B, -1, -1), dropout_u=u, dropout_p=0.05)  # noqa: E731
    step()
    torch.cuda.synchronize()
    first, second = collections.Counter(), collections.Counter()
    orig = autodiff.Tape.accumulate

    def acc(self, t, g):
        b = autodiff._base(t)
        fr = traceback.extract_stack()[:-1]
        site = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-2:][::-1])
        (second if id(b) in self.grads else first)[(site, tuple(b.shape))] += 1
        return orig(self, t, g)

    autodiff.Tape.accumulate = acc
    step()
    torch.cuda.synchronize()
    print("accumulate() calls that ADD to an existing gradient (one add launch each):", sum(second.values()), "; first contributions:", sum(first.values()))
    by_site = collections.Counter()
    for (site, shape), n in second.items():
        by_site[site] += n
    for site, n in by_site.most_common():
        print(f"  {n:4d}  {site}")
    print("by site and shape:")
    for (site, shape), n in sorted(second.items(), key=lambda kv: -kv[1])[:40]:
        print(f"  {n:4d}  {str(shape):22s} {site}")


if __name__ == "__main__":
    main()
