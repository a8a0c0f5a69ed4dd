#!/usr/bin/env python3
"""bench.py — AnyEdit diffusion-denoising hot path on MI355X.

Metric (BASELINE.json): edited-images/sec @512x512, 50 DDIM steps.  Workload at N GPUs = BASELINE.json configs[1] per GPU:
AnySD (SD-1.5 UNet, in_channels=8, + task embedding / routed expert adapters), bf16, batch = 4 images per GPU,
3 CFG branches (text+image, image, uncond) -> UNet batch 12, 50 DDIM steps, synthetic latents / instruction embeddings
(SURVEY.md §8d cfg 2), random-init weights of that architecture with the zero-init layers re-initialised (G1).
A "step" = one full 50-DDIM-step edit of the per-GPU batch; 1 edited image = 150 UNet-sample evaluations = 120.5 TFLOP.
VAE / CLIP encoders are outside the timed loop and outside this path's scope.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the launch stream for the dominant kernel;
`cpu_baseline` times the oracle (oracle/, our fp32 CPU restatement — test infrastructure, here only as the measured baseline)
on the host cores for one UNet evaluation (bounded sample) and scales it to the metric's unit.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SD15 = dict(image_size=64, in_channels=8, model_channels=320, out_channels=4, num_res_blocks=2,
            attention_resolutions=[4, 2, 1], channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
            transformer_depth=1, context_dim=768, legacy=False)
GFLOP_PER_UNET_SAMPLE = 803.4  # BASELINE.md §2 (64x64 latent)
# multiply-adds the four-2x2-conv form of the three Upsample convs does not execute (5/9 of 2 * HW_out * Cout * 9 * Cin; 1280 -> 1280 @16^2, 1280 -> 1280 @32^2,
# 640 -> 640 @64^2 per sample), in GFLOP, by latent side; 0 when AE_UP2_SUBPIXEL=0 restores the gather form
_up2 = lambda side: (5.0 / 9.0) * 2 * 9 * ((side // 4) ** 2 * 1280 * 1280 + (side // 2) ** 2 * 1280 * 1280 + side ** 2 * 640 * 640) / 1e9  # noqa: E731
UP2_SKIPPED_GFLOP = {64: _up2(64) if os.environ.get("AE_UP2_SUBPIXEL", "1") != "0" else 0.0, 96: _up2(96) if os.environ.get("AE_UP2_SUBPIXEL", "1") != "0" else 0.0}
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0
# What the matrix pipes SUSTAIN on this part under its 1400 W package-power limit with random bf16 operands, measured by a pure-MFMA loop on every SIMD
# while the firmware reported the PPT limiter active in every sample (tools/ubench/mfma_burn.hip + tools/throttle_probe.py, profiles/r05_throttle_mfma*.json):
# v_mfma_f32_16x16x32_bf16 (the GEMM / conv kernels' shape) 1645 TFLOP/s at ~2.05 GHz, v_mfma_f32_32x32x16_bf16 (attention) 1873 TFLOP/s at ~1.85 GHz; with all-zero
# operands the same loop draws 840 W and holds 2.39 GHz.  Quoted NEXT TO the datasheet fraction, never instead of it.
SUSTAINED_BF16_TFLOPS = {"16x16x32": 1645.0, "32x32x16": 1873.0}


def build_model(device, seed=0):
    from anyedit_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel, ResBlock
    from anyedit_amd.ldm.modules.attention import SpatialTransformer
    from anyedit_amd.ldm.models.diffusion.ddpm import DDPM
    from anyedit_amd.anysd.model import MoE
    torch.manual_seed(seed)
    with torch.device(device):
        unet = UNetModel(**SD15)
    g = torch.Generator(device=device).manual_seed(seed + 1)
    with torch.no_grad():  # G1: zero-init layers -> N(0, 0.02^2), else the UNet outputs exactly 0
        for m in unet.modules():
            tgt = []
            if isinstance(m, ResBlock):
                tgt.append(m.out_layers[-1])
            if isinstance(m, SpatialTransformer):
                tgt.append(m.proj_out)
            if isinstance(m, UNetModel):
                tgt.append(m.out[-1])
            for t in tgt:
                for p in t.parameters():
                    p.copy_(torch.randn(p.shape, generator=g, device=device) * 0.02)
    unet.eval().requires_grad_(False)
    with torch.device(device):
        moe = MoE(unet, expert_num=11)
    moe.eval().requires_grad_(False)
    sched = DDPM(unet, timesteps=1000, linear_start=0.00085, linear_end=0.0120).to(device)
    return unet, moe, sched


def synthetic_inputs(B, device, rank=0, latent=64):
    g = lambda s: torch.Generator(device="cpu").manual_seed(1000 * rank + s)
    x_T = torch.randn(B, 4, latent, latent, generator=g(1)).to(device)
    img_lat = (torch.randn(B, 4, latent, latent, generator=g(2)) * 0.18215).to(device)
    ehs = torch.randn(B, 77, 768, generator=g(3)).to(device)
    null = torch.randn(1, 77, 768, generator=g(6)).to(device)
    ref = torch.randn(B, 257, 1280, generator=g(7)).to(device)
    code = (torch.arange(B) % 3).to(device)  # add / remove / replace mix (configs[2])
    return x_T, img_lat, ehs, null, ref, code


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median_time(fn, reps=3, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        s = time.time()
        fn()
        ts.append(time.time() - s)
    ts.sort()
    return ts[len(ts) // 2]


def cpu_e2e_config1(sd):
    """config 1 end to end on the host cores — ONE [1,4,64,64] latent, 20 DDIM steps, 8-channel UNet input,
    cfg 7.5 / 1.5, eta 0, with k = 3 branches (IP2P / AnySD) and k = 2 (ldm-DDIM style): the oracle loop, fp32."""
    from oracle import ddim_ref as D, ldm_ref as L, schedule_ref as S
    g = torch.Generator().manual_seed(3)
    x_T, img = torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g) * 0.18215
    ctx, null = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
    buffers = S.register_schedule("linear", 1000, 0.00085, 0.0120)
    unet_fn = lambda x, t, c: L.unet_forward(sd, SD15, x, t, c)
    out = {}
    with torch.no_grad():
        t0 = time.time()
        D.ip2p_edit_loop(unet_fn, buffers, 20, x_T, img, ctx, null, 7.5, 1.5)
        out["k3_seconds"] = time.time() - t0
        t0 = time.time()
        apply = lambda x, t, c: L.unet_forward(sd, SD15, torch.cat([x, img.expand(x.shape[0], -1, -1, -1)], 1), t, c)
        D.ddim_sample(apply, buffers, 20, (1, 4, 64, 64), ctx, eta=0.0, x_T=x_T, scale=7.5, uc=null)
        out["k2_seconds"] = time.time() - t0
    out["k3_images_per_s"], out["k2_images_per_s"] = 1.0 / out["k3_seconds"], 1.0 / out["k2_seconds"]
    return out


def cpu_op_timings(sd):
    """BASELINE.md §3 per-op legs on the host cores (oracle, fp32, median of 3 after one warm-up), seconds."""
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(4)
    emb = torch.randn(1, 1280, generator=g)
    x64, x16 = torch.randn(1, 320, 64, 64, generator=g), torch.randn(1, 1280, 16, 16, generator=g)
    tok = torch.randn(1, 4096, 320, generator=g)
    ctx = torch.randn(1, 77, 768, generator=g)
    p1 = "input_blocks.1.1.transformer_blocks.0."
    res = {}
    with torch.no_grad():
        res["self_attention_L1 [1,4096,320] 8 heads d=40"] = _median_time(lambda: L.cross_attention(sd, p1 + "attn1.", tok, None, heads=8))
        res["cross_attention_L1 Nk=77"] = _median_time(lambda: L.cross_attention(sd, p1 + "attn2.", tok, ctx, heads=8))
        res["resblock 320@64x64"] = _median_time(lambda: L.resblock(sd, "input_blocks.1.0.", x64, emb))
        res["resblock 1280@16x16"] = _median_time(lambda: L.resblock(sd, "input_blocks.8.0.", x16, emb))
        res["groupnorm32 [1,320,64,64]"] = _median_time(lambda: L.group_norm32(x64, sd["input_blocks.1.0.in_layers.0.weight"], sd["input_blocks.1.0.in_layers.0.bias"]), reps=9)
        qg, qw = torch.randn(16, 4096, 80, generator=g), torch.randn(400, 196, 80, generator=g)
        res["sam_global_attention [16,4096,80]"] = _median_time(lambda: L.sdpa_core(qg, qg, qg, 80 ** -0.5))
        res["sam_windowed_attention [400,196,80]"] = _median_time(lambda: L.sdpa_core(qw, qw, qw, 80 ** -0.5))
    return res


def cpu_baseline_and_parity(unet, device, max_seconds=40.0, e2e=False, per_op=False):
    """Oracle UNet evaluation (B=1, fp32, all host cores) on the SAME weights: CPU time + full-size parity of the HIP path."""
    from oracle import ldm_ref as L
    torch.set_num_threads(min(os.cpu_count(), 32))  # 256 threads on this shape is 30x slower than 8 (measured)
    t0 = time.time()
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 8, 64, 64, generator=g)
    t = torch.tensor([501], dtype=torch.long)
    ctx = torch.randn(1, 77, 768, generator=g)
    times = []
    with torch.no_grad():
        while True:
            s = time.time()
            ref = L.unet_forward(sd, SD15, x, t, ctx)
            times.append(time.time() - s)
            if len(times) >= 3 or time.time() - t0 > max_seconds:
                break
        got = unet(x.to(device), t.to(device), context=ctx.to(device)).float().cpu()
    times.sort()
    t_unet = times[len(times) // 2]
    err = float((got - ref).norm() / ref.norm())
    mse = float(((got - ref) ** 2).mean())
    peak = float(ref.max() - ref.min())
    import math
    psnr = 10 * math.log10(peak * peak / mse) if mse > 0 else float("inf")
    cpu = {"value": 1.0 / (150.0 * t_unet), "unit": "edited-images/sec", "cores": min(os.cpu_count(), 32), "kind": "port",
           "sample": f"oracle UNet forward B=1 [1,8,64,64], median of {len(times)} = {t_unet:.3f} s, scaled x150 evaluations/image",
           "unet_forward_s": t_unet, "cpu_model": _cpu_model(), "os_cpu_count": os.cpu_count(), "threads": torch.get_num_threads()}
    if e2e:    # opt-in (--cpu-e2e): ~3-5 minutes of host time
        cpu["config1_end_to_end_20_steps"] = cpu_e2e_config1(sd)
    if per_op:  # opt-in (--cpu-ops)
        cpu["per_op_seconds"] = cpu_op_timings(sd)
    parity = {"unet_full_size_rel_l2_vs_oracle": err, "psnr_db": psnr}
    return cpu, parity


def traffic_for(kernel):
    """(HBM-side bytes per launch of `kernel`, source file): FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes over this
    same workload (tools/traffic.sh -> profiles/r*_traffic.json).  PMC passes cannot run inside the timed process, so the figure
    comes from the newest committed collection (its "commit" field says which code it was taken on); (None, None) when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None, None
    doc = json.load(open(files[-1]))
    k = doc.get("kernels", {}).get(kernel)
    src = os.path.basename(files[-1]) + (f" @ {doc['commit']}" if doc.get("commit") else "")
    if not k or k.get("fetch_bytes") is None or k.get("write_bytes") is None:
        return None, src
    return k["fetch_bytes"] + k["write_bytes"], src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=4, help="images per GPU per step")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--latent", type=int, default=64, help="latent side (64 = the 512x512 metric; 96 = 768x768, configs[4], informational)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-e2e", action="store_true", help="BASELINE.md §3: also time config 1 (one latent, 20 DDIM steps, k = 3 and k = 2) end to end on the host cores (minutes)")
    ap.add_argument("--cpu-ops", action="store_true", help="BASELINE.md §3: also time the per-op legs on the host cores")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torchrun with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # under torch.distributed.run the process group is set up even for ONE rank, so that a 1-GPU box exercises the same init / barrier /
    # max-over-ranks / teardown calls the N-GPU job makes (python bench.py without the launcher stays collective-free)
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI

    from anyedit_amd import ops
    from anyedit_amd.anysd.pipeline import EditPipeline
    unet, moe, sched = build_model(device)
    B = args.images
    x_T, img_lat, ehs, null, ref, code = synthetic_inputs(B, device, rank, args.latent)
    pipe = EditPipeline(moe, sched, use_graph=not args.no_graph)

    def one_step():
        # inference shards by image: ranks never exchange data inside the loop
        return pipe.edit(x_T, img_lat, ehs, null, ref, code, steps=args.ddim_steps, s_txt=7.5, s_img=1.5, eta=0.0)

    def barrier():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_step()
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if dist_on:
        # every rank's own wall time (the scaling run's first question is "which rank was slow"), then the MAX over ranks as the job's time
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allt = torch.zeros(world, device=device, dtype=torch.float64)
        torch.distributed.all_gather_into_tensor(allt, tt)
        per_rank = [float(v) for v in allt.tolist()]
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all(), "non-finite latents"

    result = None
    if rank == 0:
        n_unet_steps = len(pipe.sampler.ddim_timesteps)
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        result = {
            "metric": "edited-images/sec @512x512, 50 DDIM steps" if args.latent == 64 else f"edited-images/sec @{8 * args.latent}x{8 * args.latent}, 50 DDIM steps (informational)",
            "value": value, "unit": "edited-images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[1]: AnySD (SD-1.5 UNet in=8 + task embedding/expert adapters) 512x512 bf16, " if args.latent == 64 else
                                    f"the denoising stage of configs[4] (local-edit path) at {8 * args.latent}x{8 * args.latent}: bf16 MFMA attention — the fp8 "
                                    "attention kernel (ae_attn_fwd_fp8) exists for the SAM global blocks, is slower than bf16 and is NOT used; ") +
                                   f"{n_unet_steps} DDIM steps, batch={B}/GPU, 3-branch CFG (UNet batch {3 * B})",
                       "images_per_gpu": B, "ddim_steps": n_unet_steps, "cfg_branches": 3, "hip_graph": not args.no_graph,
                       "parallelism": f"dp{world} (image-sharded, no data-path collective)"},
            "per_rank_ms_per_step": {"min": 1e3 * min(per_rank) / args.steps, "max": 1e3 * max(per_rank) / args.steps,
                                     "all": [round(1e3 * v / args.steps, 3) for v in per_rank]},
            "unet_step_ms": ms_per_step / n_unet_steps,
            # MODEL FLOPs of the reference graph (803.4 GFLOP per sample and evaluation) per second — the Upsample convs execute 4/9 of their
            # share since round 5 (four 2x2 convs), so the EXECUTED rate is lower: both are reported
            "unet_tflops": 3 * B * (GFLOP_PER_UNET_SAMPLE if args.latent == 64 else 2148.3 if args.latent == 96 else float("nan")) * n_unet_steps / (ms_per_step * 1e-3) / 1e3,
            "unet_tflops_kind": "model FLOPs of the reference graph",
            "unet_tflops_executed": 3 * B * ((GFLOP_PER_UNET_SAMPLE - UP2_SKIPPED_GFLOP[64]) if args.latent == 64 else (2148.3 - UP2_SKIPPED_GFLOP[96]) if args.latent == 96 else float("nan"))
            * n_unet_steps / (ms_per_step * 1e-3) / 1e3,
            "unet_step_ms_p50_note": "graph replay with the time-embedding chain hoisted out of the step since round 5 (4 launches fewer than rounds <= 4)",
        }
        # BASELINE.json's second metric: UNet-step ms p50 — per-replay HIP-event timing of the captured UNet evaluation
        if pipe._graph is not None:
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(41)]
            for e0, e1 in evs:
                e0.record()
                pipe._graph.replay()
                e1.record()
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
            result["unet_step_ms_p50"] = ts[len(ts) // 2]
            result["unet_step_ms_p90"] = ts[int(len(ts) * 0.9)]
        if not args.no_roofline:
            # eager (un-graphed) UNet evaluation with a HIP-event pair around every kernel launch on the launch stream
            pipe.prepare(img_lat, ehs, null, ref, code)
            pipe._x_in[:, :4].view(3, B, 4, args.latent, args.latent).copy_(x_T.unsqueeze(0))
            pipe.set_step(501)   # (also refreshes the hoisted time-embedding rows the evaluation reads: ADVICE r5)
            for _ in range(2):
                pipe._denoise_static()
            torch.cuda.synchronize()
            with ops.OpProfiler() as prof:
                for _ in range(3):
                    pipe._denoise_static()
            summ = prof.summary()
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "kernels_by_shape.json"), "w") as f:  # per-shape table for the tuning loop
                json.dump({k: {"calls": v["calls"] // 3, "launches": v["launches"] // 3, "avg_us": v["avg_us"], "tflops": v["tflops"], "gbps": v["gbps"]}
                           for k, v in sorted(prof.summary(by_shape=True).items(), key=lambda kv: -kv[1]["ms"])}, f, indent=1)
            # dominant KERNEL = the instantiation with the largest share of the step, as rocprofv3 --stats names it: the split-K and
            # single-pass launches of the 128x128 implicit-GEMM conv are one kernel symbol (gemm_kernel<128,128,conv,...>), so their
            # two profiler labels are folded before the maximum is taken (per-label rows stay in "kernels")
            merged = {}
            for k, v in summ.items():
                key = k.replace(",splitK", "")
                m = merged.setdefault(key, {"ms": 0.0, "flops": 0.0, "bytes": 0.0, "calls": 0, "labels": []})
                m["ms"] += v["ms"]; m["flops"] += v["flops"]; m["bytes"] += v.get("bytes", 0.0); m["calls"] += v["calls"]
                m["labels"].append((k, v["calls"]))
            name, m = max(merged.items(), key=lambda kv: kv[1]["ms"])
            a = {"ms": m["ms"], "flops": m["flops"], "calls": m["calls"], "avg_us": 1e3 * m["ms"] / m["calls"],
                 "tflops": m["flops"] / (m["ms"] * 1e-3) / 1e12 if m["ms"] > 0 else 0.0,
                 "gbps": m["bytes"] / (m["ms"] * 1e-3) / 1e9 if m["ms"] > 0 else 0.0}
            traffic, traffic_src = traffic_for(name)   # tools/traffic.sh keys its rows by kernel symbol, i.e. by the folded name
            mfma = a["flops"] > 0
            result["roofline"] = {
                "kernel": name, "bound": "mfma" if mfma else "hbm",
                "achieved": a["tflops"] if mfma else a["gbps"], "peak": PEAK_BF16_TFLOPS if mfma else PEAK_HBM_GBPS,
                "unit": "TFLOP/s" if mfma else "GB/s",
                "frac": (a["tflops"] / PEAK_BF16_TFLOPS) if mfma else (a["gbps"] / PEAK_HBM_GBPS),
                "frac_of_sustained_mfma_rate": (a["tflops"] / SUSTAINED_BF16_TFLOPS["32x32x16" if "attn" in name else "16x16x32"]) if mfma else None,
                "sustained_mfma_rate_tflops": (SUSTAINED_BF16_TFLOPS["32x32x16" if "attn" in name else "16x16x32"]) if mfma else None,
                "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes": m["bytes"] / m["calls"] if m["calls"] else None,
                "avg_launch_us": a["avg_us"], "launches_per_unet_step": a["calls"] // 3,
                "share_of_unet_step": a["ms"] / sum(v["ms"] for v in summ.values()),
            }
            # calls = module-level calls (a GroupNorm call is 1-3 launches, a split-K conv adds its reduce): launches_per_step counts kernels
            result["kernels"] = {k: {"calls_per_step": v["calls"] // 3, "launches_per_step": v["launches"] // 3, "ms_per_step": v["ms"] / 3, "avg_us": v["avg_us"],
                                     "tflops": v["tflops"], "gbps": v["gbps"]} for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        if world == 1 and not args.no_cpu_baseline:
            cpu, parity = cpu_baseline_and_parity(unet, device, e2e=args.cpu_e2e, per_op=args.cpu_ops)
            result["cpu_baseline"] = cpu
            result["parity"] = parity
        print(json.dumps(result), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
