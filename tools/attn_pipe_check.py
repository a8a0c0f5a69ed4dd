"""Software-pipelined self-attention kernel (round 5, AE_ATTN_V flag 4) against the two-query-group kernel it replaces (AE_ATTN_V = 3).

The two kernels issue the same MFMAs on the same operands in the same per-accumulator order and take the same rebase decisions, so their outputs
must be BIT-IDENTICAL.  The launcher reads AE_ATTN_V once per process: this script runs itself as two children (one per variant) that write
their outputs to a file, then compares.  Also: against an fp32 torch statement (sampled heads), run-to-run equality, and the kernel times.

    python tools/attn_pipe_check.py            # parent
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases(dev):
    from anyedit_amd import ops
    BF = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(5)
    out = {}

    def rnd(*s):
        return torch.randn(*s, generator=g, device=dev).to(BF)

    # 1: the UNet shape, contiguous [BH, N, D], with late spikes (the lazy offset moves in some query groups only)
    BH, N, D = 96, 4096, 40
    q, k, v = rnd(BH, N, D), rnd(BH, N, D), rnd(BH, N, D)
    k[:, 1000] = q[:, 7] * 5.0
    k[:, 3000] = q[:, 600] * 8.0
    out["bhnd_spikes"] = ops.attention_bhnd(q, k, v)
    ref_err = 0.0
    for h in (0, 1, 47, 95):
        s = (q[h].float() @ k[h].float().T) * D ** -0.5
        ref = torch.softmax(s, -1) @ v[h].float()
        ref_err = max(ref_err, float((out["bhnd_spikes"][h].float() - ref).norm() / ref.norm()))
    same = all(torch.equal(ops.attention_bhnd(q, k, v), out["bhnd_spikes"]) for _ in range(20))
    # 2: ragged query count (rows past Nq clamped, never stored)
    out["ragged_nq"] = ops.attention_bhnd(q[:, :N - 40].contiguous(), k, v)
    # 3: the fused-qkv strided layout CrossAttention.rows uses (B = 12, h = 8, C = 320)
    B, h, C = 12, 8, 320
    qkv = rnd(B * N, 3 * C)
    st = (N * 3 * C, D, 3 * C)
    out["fused_qkv"] = ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B, h, N, N, D, D ** -0.5, st, st, st)
    # 4: short key counts: two tiles (the minimum), three tiles (every buffer once), and a large logit scale (frequent rebases)
    for nk in (128, 192, 256, 384, 1024):   # (256 / 384: two / three tiles of the 128-key form, round 6)
        kk, vv = rnd(BH, nk, D), rnd(BH, nk, D)
        out[f"nk{nk}"] = ops.attention_bhnd(q, kk, vv)
        out[f"nk{nk}_hot"] = ops.attention_bhnd(q, kk, vv, scale=3.0)
    # 5: log-sum-exp output, per-batch output scale, accumulation into an existing output (training / adapter options)
    lse = torch.zeros(B, h, N, dtype=torch.float32, device=dev)
    osc = torch.linspace(0.5, 1.5, B, device=dev)
    acc = rnd(B, N, h * D)
    o5 = acc.clone()
    ops.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], B, h, N, N, D, D ** -0.5, st, st, st, out=o5, out_scale=osc, accumulate=True, lse=lse)
    out["opts_out"], out["opts_lse"] = o5, lse
    # timing: the UNet shape, fresh operands
    q, k, v = rnd(BH, N, D), rnd(BH, N, D), rnd(BH, N, D)
    for _ in range(3):
        ops.attention_bhnd(q, k, v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.attention_bhnd(q, k, v)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    return out, ref_err, same, us


def child(path):
    out, ref_err, same, us = cases("cuda")
    torch.save({k: v.cpu() for k, v in out.items()}, path)
    print(f"AE_ATTN_V={os.environ.get('AE_ATTN_V')} AE_ATTN_KT128={os.environ.get('AE_ATTN_KT128', 'default')} AE_ATTN_PV16={os.environ.get('AE_ATTN_PV16', '0')}: rel-L2 vs fp32 torch {ref_err:.3e} {'OK' if ref_err < 6e-3 else 'FAIL'}; 20 launches bit-identical: {same}; "
          f"{us:.1f} us = {4.0 * 96 * 4096 * 4096 * 40 / us / 1e6:.1f} TFLOP/s (algorithmic, d = 40)", flush=True)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    variants = sys.argv[1:] or ["3", "7k0", "7"]
    paths = {}
    for v in variants:
        paths[v] = f"/tmp/attn_pipe_check_{v}.pt"
        # "7" = AE_ATTN_V 7; "7k0" = the same with AE_ATTN_KT128=0 (the 64-key tiles of round 5)
        # "7p" = AE_ATTN_V 7 with AE_ATTN_PV16=1 (round 6: the 48-row 16x16x32 PV products — another summation order, compared by tolerance, not bit for bit)
        core = v.replace("p", "")
        env = dict(os.environ, AE_ATTN_V=core.split("k")[0], AE_ATTN_PV16="1" if "p" in v else "0")
        if "k" in core:
            env["AE_ATTN_KT128"] = core.split("k")[1]
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", paths[v]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        print(r.stdout.strip()[-600:])
        if r.returncode:
            print(f"variant {v}: child failed rc={r.returncode}")
            return 1
    a = torch.load(paths[variants[0]])
    ok = True
    for v in variants[1:]:
        b = torch.load(paths[v])
        if "p" in v or "p" in variants[0]:
            worst = 0.0
            for k in a:
                d = a[k].float() - b[k].float()
                e = float(d.norm() / a[k].float().norm())
                worst = max(worst, e)
                assert torch.isfinite(b[k].float()).all(), k
            close = worst < 2.5e-3   # two bf16 roundings of the same fp32 sums taken in another order
            print(f"variants {variants[0]} / {v}: worst rel-L2 over {len(a)} outputs {worst:.3e} {'OK' if close else 'FAIL'} (another PV summation order: tolerance, not identity)")
            ok &= close
            continue
        for k in a:
            eq = torch.equal(a[k], b[k])
            if not eq:
                d = (a[k].float() - b[k].float())
                print(f"  {k}: variants {variants[0]} / {v} DIFFER: max-abs {float(d.abs().max()):.3e}, {int((d != 0).sum())} of {d.numel()} values, rel-L2 {float(d.norm() / a[k].float().norm()):.3e}")
            ok &= eq
        print(f"variants {variants[0]} / {v}: {'bit-identical on all ' + str(len(a)) + ' outputs' if ok else 'NOT identical'}")
    return 0 if ok else 2


if __name__ == "__main__":
    sys.exit(main())
