#!/usr/bin/env python3
"""Training-step timing for BASELINE.json configs[3] (SURVEY.md §8d cfg 4): AnySD adapters on the frozen SD-1.5 UNet, 4 editing
pairs per GPU at 64x64 latents (north star: 512 px), bf16 activations, fp32 AdamW state.  One step = q_sample + forward on the
tape + backward through the frozen UNet + (multi-GPU: one gradient exchange) + AdamW.  Prints one JSON line on rank 0.

    python tools/bench_train.py [--steps 5] [--warmup 1] [--batch 4]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_train.py
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (model / synthetic-input builders shared with the inference bench)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--checkpoint", action="store_true", help="use_checkpoint=True: recompute every ResBlock / BasicTransformerBlock in the backward pass")
    ap.add_argument("--per-step", action="store_true", help="synchronise after every step and report each step's time")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    from anyedit_amd.anysd.train import AnySDTrainer
    unet, moe, sched = bench.build_model(device)
    for p in list(moe.image_proj_model.parameters()) + list(moe.adapter_modules) + [moe.task_embs]:
        p.requires_grad_(True)
    if args.checkpoint:
        for m in moe.modules():
            if hasattr(m, "use_checkpoint"):
                m.use_checkpoint = True
            if m.__class__.__name__ == "BasicTransformerBlock":
                m.checkpoint = True
    B = args.batch
    g = torch.Generator(device="cpu").manual_seed(4 + rank)
    lat = torch.randn(B, 4, 64, 64, generator=g).to(device)
    img = (torch.randn(B, 4, 64, 64, generator=g) * 0.18215).to(device)
    ehs = torch.randn(B, 77, 768, generator=g).to(device)
    null = torch.randn(1, 77, 768, generator=g).to(device)
    ref = torch.randn(B, 257, 1280, generator=g).to(device)
    code = (torch.arange(B) % 3).to(device)
    tr = AnySDTrainer(moe, sched.sqrt_alphas_cumprod, sched.sqrt_one_minus_alphas_cumprod, lr=1e-5)

    def draw(i):
        gi = torch.Generator(device="cpu").manual_seed(100 * rank + i)
        return (torch.randn(B, 4, 64, 64, generator=gi).to(device), torch.randint(0, 1000, (B,), generator=gi).to(device),
                torch.rand(B, generator=gi).to(device))

    # per-step noise / timestep / dropout draws are made up front: the timed region holds no host->device copies
    draws = [draw(i) for i in range(args.warmup + args.steps + 1)]
    torch.cuda.synchronize()

    def step(i):
        noise, t, u = draws[i]
        return tr.train_step(lat, img, ehs, ref, code, noise, t, null_ehs=null.expand(B, -1, -1), dropout_u=u, dropout_p=0.05)

    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    per_step = []
    for i in range(args.steps):
        ts = time.perf_counter()
        loss = step(args.warmup + i)
        if args.per_step:
            th = time.perf_counter()
            torch.cuda.synchronize()
            per_step.append((round(1e3 * (th - ts), 2), round(1e3 * (time.perf_counter() - ts), 2)))   # (host enqueue time, step time)
            if os.environ.get("AE_TRAIN_DIAG"):
                import gc
                ms = torch.cuda.memory_stats()
                print("diag", i, per_step[-1][1], "reserved", ms["reserved_bytes.all.current"] >> 20, "segs", ms["segment.all.allocated"], "freed", ms["segment.all.freed"],
                      "retries", ms["num_alloc_retries"], "gc", [g["collections"] for g in gc.get_stats()], file=sys.stderr)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    assert torch.isfinite(loss).all()
    by_shape = None
    if rank == 0 and os.environ.get("AE_TRAIN_PROFILE"):
        from anyedit_amd import ops
        with ops.OpProfiler() as prof:
            step(args.warmup + args.steps)
        by_shape = {k: {"calls": v["calls"], "avg_us": round(v["avg_us"], 1), "ms": round(v["ms"], 3), "tflops": round(v["tflops"], 1)}
                    for k, v in sorted(prof.summary(by_shape=True).items(), key=lambda kv: -kv[1]["ms"])[:40]}
        fam = {k: {"calls": v["calls"], "ms": round(v["ms"], 3)} for k, v in sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"])}
        by_shape = {"families": fam, "shapes": by_shape}
    if rank == 0:
        fwd_tflop = B * bench.GFLOP_PER_UNET_SAMPLE / 1e3
        print(json.dumps({"metric": "AnySD training step (adapters on frozen SD-1.5 UNet, 64x64 latents)", "ms_per_step": 1e3 * dt,
                          "pairs_per_sec": world * B / dt, "n_gpus": world, "batch_per_gpu": B, "dtype": "bf16 activations, fp32 state",
                          "forward_tflop": fwd_tflop, "approx_tflops": 3.0 * fwd_tflop / dt,
                          "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30, "loss": float(loss),
                          "exchange_bytes_per_step": tr.exchange.bytes_per_step if tr.exchange else 0, "activation_checkpointing": bool(args.checkpoint),
                          "per_step_ms": per_step or None,
                          "profile": by_shape}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
