#!/usr/bin/env python3
"""Which Python call sites of one training step end in a torch copy / elementwise kernel (the step's non-HIP-library launches)?
Runs one warm step under torch.profiler with stacks and prints, per aten op that launches a kernel, the innermost repo frames."""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    from anyedit_amd.anysd.train import AnySDTrainer
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

    def step():
        return tr.train_step(lat, img, ehs, ref, code, noise, t, null_ehs=null.expand(B, -1, -1), dropout_u=u, dropout_p=0.05)

    step(); step()
    torch.cuda.synchronize()
    # count aten-level calls by python call site: wrap the torch functions that can launch a copy
    sites = collections.Counter()
    import traceback

    def site():
        fr = [f for f in traceback.extract_stack()[:-2] if "/anyedit_amd/" in f.filename or "/tools/" in f.filename]
        return " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])

    from torch.utils._python_dispatch import TorchDispatchMode

    class M(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func)
            big = 0
            for a in list(args) + list((kwargs or {}).values()):
                if isinstance(a, torch.Tensor):
                    big = max(big, a.numel() * a.element_size())
            if not any(s in name for s in ("view", "reshape", "as_strided", "detach", "alias", "expand", "slice", "select", "unsqueeze", "squeeze", "permute", "transpose", "t.default", "_unsafe_view", "empty", "is_", "sym_", "stride", "size")):
                sites[(name, site(), big >> 10)] += 1
            return func(*args, **(kwargs or {}))

    with M():
        step()
    torch.cuda.synchronize()
    tot = collections.Counter()
    for (name, s, kb), n in sites.items():
        tot[name] += n
    print("aten ops that run in one training step (dispatch level), by op:")
    for name, n in tot.most_common(40):
        print(f"  {n:5d}  {name}")
    print("by call site (op, site, largest operand KiB):")
    for (name, s, kb), n in sorted(sites.items(), key=lambda kv: -kv[1] * max(kv[0][2], 1))[:70]:
        print(f"  {n:4d} x {kb:8d} KiB  {name:40s} {s}")


if __name__ == "__main__":
    main()
