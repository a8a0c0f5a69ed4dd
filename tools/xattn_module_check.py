"""The opt-in fused cross-attention half (AE_XATTN_FUSED=1) inside the product: a 4-step, 3-branch-CFG edit of bench.py's model at 64x64 (UNet batch 3, K/V images from
prepare_conditioning) with the switch on, against the same edit on the default plan (three launches per cross-attention).  The module reads the switch once
per process, so this script runs itself as children and compares what they wrote.

What "the same" means for two bf16 plans of a guided sampler: the guidance (7.5 / 1.5) amplifies every rounding difference between the branches and the steps compound it
(x1.6 per guided step), so two CORRECT plans differ by a few 1e-2 after four steps.  The yardstick is therefore measured in the same run: a third child runs
another plan of proven operator-level parity — the feed-forward half as two launches instead of the fused one (AE_FF_FUSED=0: five launches per evaluation differ, as
with the cross-attention switch) — and the fused cross-attention edit must sit within 3x of that plan's distance from the default (a second
child on the default plan must reproduce the first bit for bit, so the distances are the plans' and not the run's).

    python tools/xattn_module_check.py
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path):
    import bench
    from anyedit_amd import ops
    from anyedit_amd.anysd.pipeline import EditPipeline
    dev = torch.device("cuda", 0)
    unet, moe, sched = bench.build_model(dev)
    x_T, img_lat, ehs, null, ref, code = bench.synthetic_inputs(1, dev, 0, 64)
    pipe = EditPipeline(moe, sched, use_graph=False)   # eager: the profiler counts the launches
    with ops.OpProfiler() as prof:
        out = pipe.edit(x_T, img_lat, ehs, null, ref, code, steps=4, s_txt=7.5, s_img=1.5, eta=0.0)
    torch.cuda.synchronize()
    fused = sum(v["calls"] for k, v in prof.summary().items() if "xattn_fused" in k)
    assert torch.isfinite(out).all()
    torch.save(out.float().cpu(), path)
    print(f"AE_XATTN_FUSED={os.environ.get('AE_XATTN_FUSED', '0')}: fused cross-attention launches in the profiled edit: {fused}", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    outs = {}
    for tag, extra in (("base", {}), ("base2", {}), ("xattn", {"AE_XATTN_FUSED": "1"}), ("ffsplit", {"AE_FF_FUSED": "0"})):
        path = f"/tmp/xattn_module_check_{tag}.pt"
        env = dict(os.environ, AE_XATTN_FUSED="0", AE_ROWPANEL_ANY_M="1")   # ANY_M: UNet batch 3 is 96 blocks of 128 rows, below the one-block-per-CU plan rule
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(f"[{tag}] " + r.stdout.strip()[-300:].replace("\n", " | "))
        if r.returncode:
            print(f"child {tag} failed rc={r.returncode}")
            return 1
        outs[tag] = torch.load(path)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    e_x, e_f = rel(outs["xattn"], outs["base"]), rel(outs["ffsplit"], outs["base"])
    same = torch.equal(outs["base"], outs["base2"])   # two processes on the default plan: bit-equal, so the distances below are the plans', not the run's
    ok = same and e_x <= 3.0 * e_f and e_x > 0.0
    print(f"default plan, two processes: {'bit-equal' if same else 'DIFFERENT'}")
    print(f"4-step edit vs the default plan: fused cross-attention rel-L2 {e_x:.3e}; control (feed-forward as two launches) {e_f:.3e}; ratio {e_x / max(e_f, 1e-30):.2f} "
          f"{'OK' if ok else 'FAIL'}")
    return 0 if ok else 2


if __name__ == "__main__":
    sys.exit(main())
