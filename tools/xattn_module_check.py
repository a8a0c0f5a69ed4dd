"""The opt-in fused cross-attention half (AE_XATTN_FUSED=1) inside the product: the 3-step, 3-branch-CFG edit of bench.py's model at 64x64 (UNet batch 3, K/V images from
prepare_conditioning) with the switch on, against the same edit on the default plan (three launches per cross-attention).  The module reads the switch once
per process, so this script runs itself as two children and compares what they wrote.

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
        out = pipe.edit(x_T, img_lat, ehs, null, ref, code, steps=3, s_txt=7.5, s_img=1.5, eta=0.0)
    torch.cuda.synchronize()
    fused = sum(v["calls"] for k, v in prof.summary().items() if "xattn_fused" in k)
    assert torch.isfinite(out).all()
    torch.save(out.float().cpu(), path)
    print(f"AE_XATTN_FUSED={os.environ.get('AE_XATTN_FUSED', '0')}: fused cross-attention launches in the profiled edit: {fused}", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    outs = {}
    for sw in ("0", "1"):
        path = f"/tmp/xattn_module_check_{sw}.pt"
        env = dict(os.environ, AE_XATTN_FUSED=sw, AE_ROWPANEL_ANY_M="1")   # ANY_M: UNet batch 3 is 96 blocks of 128 rows, below the one-block-per-CU plan rule
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(r.stdout.strip()[-400:])
        if r.returncode:
            print(f"child AE_XATTN_FUSED={sw} failed rc={r.returncode}")
            return 1
        outs[sw] = torch.load(path)
    e = float((outs["1"] - outs["0"]).norm() / outs["0"].norm())
    ok = e < 5e-3
    print(f"3-step edit, fused cross-attention vs default plan: rel-L2 {e:.3e} {'OK' if ok else 'FAIL'}")
    return 0 if ok else 2


if __name__ == "__main__":
    sys.exit(main())
