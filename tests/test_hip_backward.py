"""GPU parity tests of the training-step kernels: every backward operator, called through the C ABI, against
torch.autograd of the oracle's forward restatement (CPU fp32) on the same seeded, bf16-rounded inputs.

Tolerances: gradients are stored in bf16 (one rounding, 2^-9 relative) and accumulate in fp32: relative L2 <= 6e-3 per operator
(1e-2 for attention, whose recomputed probabilities see bf16 P and bf16 dS), max-abs <= 3e-2 * max|ref|.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda"


def q(t):
    return t.to(BF).float()


def check_close(got, ref, rl2=6e-3, mabs=3e-2, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    e = rel_l2(got, ref)
    m = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-12)
    assert e <= rl2 and m <= mabs, f"{what}: rel_l2={e:.3e} (<= {rl2}), max_abs/max={m:.3e} (<= {mabs})"


@pytest.fixture(scope="module")
def ops():
    from anyedit_amd import ops as o
    return o


# ------------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("B,HW,C,C1,silu,eps", [(2, 64, 64, None, True, 1e-5), (3, 256, 320, None, False, 1e-6), (2, 1024, 320, None, True, 1e-5),
                                                (2, 64, 960, 640, True, 1e-5), (1, 4096, 320, None, True, 1e-5), (2, 16, 1280, None, True, 1e-5)])
def test_groupnorm_backward(ops, B, HW, C, C1, silu, eps):
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(B + HW + C)
    x = q(torch.randn(B, HW, C, generator=g) * 2 + 0.5)
    w, b = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    dy = q(torch.randn(B, HW, C, generator=g))
    xr = x.clone().requires_grad_(True)
    y = L.group_norm32(xr.permute(0, 2, 1).reshape(B, C, HW, 1), w, b, eps).reshape(B, C, HW).permute(0, 2, 1)
    if silu:
        y = F.silu(y)
    y.backward(dy)
    rows = x.reshape(B * HW, C).to(DEV, BF)
    if C1 is None:
        dx, _ = ops.groupnorm_bwd(rows, w.to(DEV), b.to(DEV), dy.reshape(B * HW, C).to(DEV, BF), B, HW, eps, silu=silu)
        check_close(dx.reshape(B, HW, C), xr.grad, what="groupnorm dx")
    else:
        x1, x2 = rows[:, :C1].contiguous(), rows[:, C1:].contiguous()
        dx1, dx2 = ops.groupnorm_bwd(x1, w.to(DEV), b.to(DEV), dy.reshape(B * HW, C).to(DEV, BF), B, HW, eps, silu=silu, x2=x2)
        check_close(torch.cat([dx1, dx2], 1).reshape(B, HW, C), xr.grad, what="groupnorm (concat) dx")
    # the training tape keeps the forward's (mean, rstd) and hands them to the backward: no second statistics pass over x
    x1, x2 = (rows, None) if C1 is None else (rows[:, :C1].contiguous(), rows[:, C1:].contiguous())
    stat = torch.empty(B, 32, 2, dtype=torch.float32, device=DEV)
    yf = ops.groupnorm(x1, w.to(DEV), b.to(DEV), B, HW, eps, silu=silu, x2=x2, stat_out=stat)
    check_close(yf.reshape(B, HW, C), y.detach(), what="groupnorm forward (with saved statistics)")
    xg = x.reshape(B, HW, 32, C // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)
    assert rel_l2(stat[..., 0].cpu(), xg.mean(-1)) <= 1e-4 and rel_l2(stat[..., 1].cpu(), (xg.var(-1, unbiased=False) + eps).rsqrt()) <= 1e-4
    dxs, dxs2 = ops.groupnorm_bwd(x1, w.to(DEV), b.to(DEV), dy.reshape(B * HW, C).to(DEV, BF), B, HW, eps, silu=silu, x2=x2, stat=stat)
    got = dxs if C1 is None else torch.cat([dxs, dxs2], 1)
    check_close(got.reshape(B, HW, C), xr.grad, what="groupnorm dx from saved statistics")


@pytest.mark.parametrize("M,C", [(16, 768), (4096, 320), (777, 1280), (10, 64)])
def test_layernorm_backward(ops, M, C):
    g = torch.Generator().manual_seed(M + C)
    x = q(torch.randn(M, C, generator=g) * 3 + 1)
    w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy = q(torch.randn(M, C, generator=g))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), wr, br, 1e-5).backward(dy)
    dx, dg, db = ops.layernorm_bwd(x.to(DEV, BF), w.to(DEV), dy.to(DEV, BF), 1e-5, want_param_grads=True)
    check_close(dx, xr.grad, what="layernorm dx")
    check_close(dg, wr.grad, rl2=2e-3, mabs=1e-2, what="layernorm dgamma")
    check_close(db, br.grad, rl2=2e-3, mabs=1e-2, what="layernorm dbeta")


@pytest.mark.parametrize("B,HW,C,C1,silu", [(2, 256, 320, None, True), (3, 100, 640, 320, True), (1, 64, 1280, 640, False), (4, 4096, 320, None, True)])
def test_groupnorm_backward_adds_into_an_existing_gradient(ops, B, HW, C, C1, silu):
    """accumulate flags of ae_groupnorm_bwd_nhwc_bf16: dx (+)= / dx2 (+)= in the apply kernel against the separate kernel + ae_add_bf16 route the tape
    used before (fp32 add of an existing bf16 gradient, one rounding: at least as close to the fp32 sum as the two-rounding route)."""
    g = torch.Generator().manual_seed(B + HW + C)
    rows = q(torch.randn(B * HW, C, generator=g) * 2 + 0.5).to(DEV, BF)
    w, b = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.2 * torch.randn(C, generator=g)).to(DEV)
    dy = q(torch.randn(B * HW, C, generator=g)).to(DEV, BF)
    old = q(torch.randn(B * HW, C, generator=g)).to(DEV, BF)
    x1, x2 = (rows, None) if C1 is None else (rows[:, :C1].contiguous(), rows[:, C1:].contiguous())
    o1, o2 = (old, None) if C1 is None else (old[:, :C1].contiguous(), old[:, C1:].contiguous())
    dx, dx2 = ops.groupnorm_bwd(x1, w, b, dy, B, HW, 1e-5, silu=silu, x2=x2)
    for mask in ((1, 2, 3) if C1 is not None else (1,)):
        t1 = o1.clone() if mask & 1 else None
        t2 = o2.clone() if (mask & 2) else None
        r1, r2 = ops.groupnorm_bwd(x1, w, b, dy, B, HW, 1e-5, silu=silu, x2=x2, dx_into=t1, dx2_into=t2)
        for name, got, into, plain, prev in (("dx", r1, t1, dx, o1), ("dx2", r2, t2, dx2, o2)):
            if plain is None:
                continue
            if into is None:
                assert torch.equal(got, plain), f"{name}: un-accumulated half changed (mask {mask})"
            else:
                assert got.data_ptr() == into.data_ptr()
                ref = plain.float() + prev.float()   # plain is bf16-rounded: the fused result is within one more rounding of this
                check_close(got, ref, rl2=4e-3, mabs=8e-3, what=f"groupnorm {name} += (mask {mask})")


@pytest.mark.parametrize("B,HW,C,C1,silu", [(4, 256, 1280, None, True), (4, 256, 2560, 1280, True), (4, 256, 1920, 1280, True), (4, 64, 2560, 1280, True), (4, 256, 640, None, True),
                                            (4, 64, 1280, None, False), (2, 200, 320, None, True), (1, 9, 960, 320, True)])
def test_groupnorm_backward_single_launch_on_small_maps(ops, B, HW, C, C1, silu):
    """gnb_slab_kernel (HW <= 256 with the forward's (mean, rstd)): one launch instead of partial / finalize / apply.  Against fp32 autograd of the oracle
    expression, against the three-launch path (no saved statistics), and with both accumulate flags; the training step's 16x16 / 8x8 shapes at batch 4."""
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(B + HW + C)
    x = q(torch.randn(B, HW, C, generator=g) * 2 + 0.5)
    w, b = 1 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    dy = q(torch.randn(B, HW, C, generator=g))
    xr = x.clone().requires_grad_(True)
    y = L.group_norm32(xr.permute(0, 2, 1).reshape(B, C, HW, 1), w, b, 1e-5).reshape(B, C, HW).permute(0, 2, 1)
    (F.silu(y) if silu else y).backward(dy)
    rows, dyd = x.reshape(B * HW, C).to(DEV, BF), dy.reshape(B * HW, C).to(DEV, BF)
    x1, x2 = (rows, None) if C1 is None else (rows[:, :C1].contiguous(), rows[:, C1:].contiguous())
    wd, bd = w.to(DEV), b.to(DEV)
    stat = torch.empty(B, 32, 2, dtype=torch.float32, device=DEV)
    ops.groupnorm(x1, wd, bd, B, HW, 1e-5, silu=silu, x2=x2, stat_out=stat)
    d3, d3b = ops.groupnorm_bwd(x1, wd, bd, dyd, B, HW, 1e-5, silu=silu, x2=x2)                 # three launches (own statistics pass)
    d1, d1b = ops.groupnorm_bwd(x1, wd, bd, dyd, B, HW, 1e-5, silu=silu, x2=x2, stat=stat)      # one launch
    cat = (lambda a, c: a if c is None else torch.cat([a, c], 1))
    check_close(cat(d1, d1b).reshape(B, HW, C), xr.grad, what="groupnorm dx (single launch)")
    check_close(cat(d1, d1b), cat(d3, d3b).float().cpu(), rl2=3e-3, mabs=1.5e-2, what="single launch vs three launches")
    again = ops.groupnorm_bwd(x1, wd, bd, dyd, B, HW, 1e-5, silu=silu, x2=x2, stat=stat)
    assert torch.equal(again[0], d1) and (d1b is None or torch.equal(again[1], d1b))
    old = q(torch.randn(B * HW, C, generator=g)).to(DEV, BF)
    o1, o2 = (old, None) if C1 is None else (old[:, :C1].contiguous(), old[:, C1:].contiguous())
    t1, t2 = o1.clone(), (o2.clone() if o2 is not None else None)
    r1, r2 = ops.groupnorm_bwd(x1, wd, bd, dyd, B, HW, 1e-5, silu=silu, x2=x2, stat=stat, dx_into=t1, dx2_into=t2)
    check_close(cat(r1, r2), cat(d1, d1b).float().cpu() + old.float().cpu(), rl2=4e-3, mabs=8e-3, what="single launch, dx += / dx2 +=")


@pytest.mark.parametrize("M,C", [(300, 320), (130, 640), (77, 1280), (16384, 320)])
def test_layernorm_backward_adds_into_an_existing_gradient(ops, M, C):
    g = torch.Generator().manual_seed(M + C)
    x = q(torch.randn(M, C, generator=g) * 3 + 1).to(DEV, BF)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    dy = q(torch.randn(M, C, generator=g)).to(DEV, BF)
    old = q(torch.randn(M, C, generator=g)).to(DEV, BF)
    dx, _, _ = ops.layernorm_bwd(x, w, dy, 1e-5)
    into = old.clone()
    got, _, _ = ops.layernorm_bwd(x, w, dy, 1e-5, dx_into=into)
    assert got.data_ptr() == into.data_ptr()
    check_close(got, dx.float() + old.float(), rl2=4e-3, mabs=8e-3, what="layernorm dx +=")


@pytest.mark.parametrize("M,N,K", [(312, 768, 640), (312, 768, 1280), (4096, 640, 640), (16384, 320, 320), (1024, 1280, 1280), (1024, 1280, 5120), (16384, 320, 1280),
                                   (4096, 640, 2560), (256, 1280, 1280), (49152, 320, 320), (12288, 640, 640), (3072, 1280, 5120)])
def test_gemm_output_may_alias_its_residual(ops, M, N, K):
    """The tape adds a data gradient to the one its input already has through the GEMM's residual, in place (out is residual): every tile plan, the split-K
    reduce and the row-panel kernel read a residual element in the thread that writes the same output element, so the result is bit-identical to the out-of-place call."""
    g = torch.Generator().manual_seed(M + N + K)
    a = q(torch.randn(M, K, generator=g)).to(DEV, BF)
    w = q(torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, BF)
    r = q(torch.randn(M, N, generator=g)).to(DEV, BF)
    ref = ops.gemm(a, w, residual=r)
    buf = r.clone()
    out = ops.gemm(a, w, residual=buf, out=buf)
    assert out.data_ptr() == buf.data_ptr() and torch.equal(out, ref)


def test_geglu_forward_backward(ops):
    g = torch.Generator().manual_seed(5)
    M, Fd = 300, 1280
    h = q(torch.randn(M, 2 * Fd, generator=g) * 1.5)
    dy = q(torch.randn(M, Fd, generator=g))
    hr = h.clone().requires_grad_(True)
    a, gate = hr.chunk(2, dim=-1)
    y = a * F.gelu(gate)
    y.backward(dy)
    check_close(ops.geglu(h.to(DEV, BF)), y, what="geglu fwd")
    check_close(ops.geglu_bwd(h.to(DEV, BF), dy.to(DEV, BF)), hr.grad, what="geglu bwd")


# ------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("BH,Nq,Nk,D", [(2, 64, 64, 40), (2, 200, 77, 80), (2, 128, 4, 40), (1, 144, 272, 160), (2, 130, 200, 16),
                                        (1, 1024, 1024, 40), (2, 256, 256, 80), (2, 70, 129, 64), (1, 33, 65, 48),
                                        # few keys against many queries (cross-attention): the dK / dV pass cuts its query tiles across blocks
                                        (8, 1024, 78, 40), (4, 600, 16, 80), (3, 520, 130, 160)])
def test_attention_backward(ops, BH, Nq, Nk, D):
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(BH + Nq + Nk + D)
    qq, kk, vv = (q(torch.randn(BH, n, D, generator=g)) for n in (Nq, Nk, Nk))
    do = q(torch.randn(BH, Nq, D, generator=g))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qq, kk, vv))
    L.sdpa_core(qr, kr, vr, D ** -0.5).backward(do)
    qd, kd, vd, dod = (t.to(DEV, BF).contiguous() for t in (qq, kk, vv, do))
    lse = torch.empty(BH, 1, Nq, dtype=torch.float32, device=DEV)
    sq, sk = (Nq * D, 0, D), (Nk * D, 0, D)
    out = ops.attention(qd, kd, vd, BH, 1, Nq, Nk, D, D ** -0.5, sq, sk, sk, lse=lse)
    # the stored log-sum-exp (log2 domain) against the oracle's logits
    sim = torch.einsum("bid,bjd->bij", qq, kk) * D ** -0.5
    check_close(lse.reshape(BH, Nq), torch.logsumexp(sim, -1) * 1.4426950408889634, rl2=2e-3, mabs=5e-3, what="lse")
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    delta = ops.attention_bwd(qd, kd, vd, dod, lse, BH, 1, Nq, Nk, D, D ** -0.5, sq, sk, sk, dq, dk, dv, sq, sk, sk)
    ref_o = L.sdpa_core(qq, kk, vv, D ** -0.5)
    check_close(delta.reshape(BH, Nq), (do * ref_o).sum(-1), rl2=2e-2, mabs=5e-2, what="delta")
    check_close(dv, vr.grad, rl2=1e-2, what=f"dV {BH}x{Nq}x{Nk}x{D}")
    check_close(dk, kr.grad, rl2=1.5e-2, mabs=5e-2, what=f"dK {BH}x{Nq}x{Nk}x{D}")
    check_close(dq, qr.grad, rl2=1.5e-2, mabs=5e-2, what=f"dQ {BH}x{Nq}x{Nk}x{D}")
    # with the segment's own forward output at hand, delta = rowsum(dO o O) is taken up front and the dQ pass keeps one accumulator set
    dq2, dk2, dv2 = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    delta2 = ops.attention_bwd(qd, kd, vd, dod, lse, BH, 1, Nq, Nk, D, D ** -0.5, sq, sk, sk, dq2, dk2, dv2, sq, sk, sk, out=out)
    check_close(delta2.reshape(BH, Nq), (do * ref_o).sum(-1), rl2=2e-2, mabs=5e-2, what="delta (from the output)")
    check_close(dv2, vr.grad, rl2=1e-2, what=f"dV' {BH}x{Nq}x{Nk}x{D}")
    check_close(dk2, kr.grad, rl2=1.5e-2, mabs=5e-2, what=f"dK' {BH}x{Nq}x{Nk}x{D}")
    check_close(dq2, qr.grad, rl2=1.5e-2, mabs=5e-2, what=f"dQ' {BH}x{Nq}x{Nk}x{D}")


@pytest.mark.parametrize("BH,Nq,Nk,D,nsplit", [(32, 4096, 78, 40, 16), (32, 1024, 78, 80, 8), (32, 4096, 16, 40, 16), (3, 520, 130, 160, 4), (32, 256, 78, 160, 0),
                                               (128, 4096, 78, 40, 0)])
def test_attention_backward_query_split_of_the_key_pass(ops, BH, Nq, Nk, D, nsplit):
    """ae_attn_bwd_bf16 with the workspace (query tiles of the dK / dV pass cut `nsplit` ways, fp32 partials, fixed-order reduce) against the same call
    without it (one block per 128 keys streams every query tile): equal to the order of an fp32 sum; run-to-run bit-identical.  The training step's
    cross-attention shapes at batch 4 (B H = 32), a shape with an EMPTY last split (9 tiles cut 4 ways), and shapes the rule leaves alone."""
    from anyedit_amd._lib import lib
    assert lib.ae_attn_bwd_workspace_floats(BH, 1, Nq, Nk, D) == nsplit * BH * Nk * 2 * ((D + 15) // 16 * 16)
    g = torch.Generator().manual_seed(Nq + Nk + D)
    qd, kd, vd = (q(torch.randn(BH, n, D, generator=g)).to(DEV, BF) for n in (Nq, Nk, Nk))
    dod = q(torch.randn(BH, Nq, D, generator=g)).to(DEV, BF)
    lse = torch.empty(BH, 1, Nq, dtype=torch.float32, device=DEV)
    sq, sk = (Nq * D, 0, D), (Nk * D, 0, D)
    ops.attention(qd, kd, vd, BH, 1, Nq, Nk, D, D ** -0.5, sq, sk, sk, lse=lse)
    res = []
    for split in (False, True, True):
        dq, dk, dv = torch.empty_like(qd), torch.full_like(kd, float("nan")), torch.full_like(vd, float("nan"))
        ops.attention_bwd(qd, kd, vd, dod, lse, BH, 1, Nq, Nk, D, D ** -0.5, sq, sk, sk, dq, dk, dv, sq, sk, sk, split_dkv=split)
        res.append((dq, dk, dv))
    assert all(torch.equal(a, b) for a, b in zip(res[1], res[2])), "split pass is not run-to-run identical"
    assert torch.equal(res[0][0], res[1][0])   # dQ does not depend on the split
    for name, a, b in (("dK", res[0][1], res[1][1]), ("dV", res[0][2], res[1][2])):
        assert torch.isfinite(b.float()).all()
        if nsplit == 0:
            assert torch.equal(a, b)
        else:
            check_close(b, a.float().cpu(), rl2=2e-3, mabs=8e-3, what=f"{name} split {nsplit} vs one block, {BH}x{Nq}x{Nk}x{D}")


@pytest.mark.parametrize("BH,N,Nk,D,gated", [(32, 4096, 4096, 40, False), (32, 1024, 1024, 80, False), (32, 1024, 78, 80, True), (32, 256, 256, 160, False), (8, 4096, 16, 40, True)])
def test_attention_backward_run_to_run_determinism(ops, BH, N, Nk, D, gated):
    """30 launches of the backward passes on identical inputs, bit for bit (dQ, dK, dV, delta): the training step's shapes, with and without the per-batch
    gate (the two-accumulator dQ pass).  Guards the MFMA-operand hazard of DESIGN 7.00h / tests/test_isa_static.py on the hardware, not only in the listing."""
    g = torch.Generator().manual_seed(N + Nk + D)
    qd, kd, vd = (q(torch.randn(BH, n, D, generator=g)).to(DEV, BF) for n in (N, Nk, Nk))
    dod = q(torch.randn(BH, N, D, generator=g)).to(DEV, BF)
    lse = torch.empty(BH, 1, N, dtype=torch.float32, device=DEV)
    sq, sk = (N * D, 0, D), (Nk * D, 0, D)
    ops.attention(qd, kd, vd, BH, 1, N, Nk, D, D ** -0.5, sq, sk, sk, lse=lse)
    gate = torch.linspace(0.5, 1.5, BH, device=DEV) if gated else None
    first = None
    for i in range(30):
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        delta = ops.attention_bwd(qd, kd, vd, dod, lse, BH, 1, N, Nk, D, D ** -0.5, sq, sk, sk, dq, dk, dv, sq, sk, sk, out_scale=gate)
        cur = (dq, dk, dv, delta)
        if first is None:
            first = cur
            assert all(torch.isfinite(t.float()).all() for t in cur)
        else:
            for name, a, b in zip(("dQ", "dK", "dV", "delta"), first, cur):
                assert torch.equal(a, b), f"launch {i}: {name} differs from launch 0 in {int((a != b).sum())} elements"


def test_fuzz_attention_and_norm_backward_shapes(ops):
    """Seeded random ragged shapes through the backward kernels (attention dQ/dK/dV, GroupNorm(+SiLU), LayerNorm) against autograd
    of the fp32 oracle expressions."""
    import numpy as np
    from oracle import ldm_ref as L
    rng = np.random.default_rng(31)
    g = torch.Generator().manual_seed(31)
    for i in range(12):
        D = [16, 40, 48, 64, 80, 160][i % 6]
        BH, Nq, Nk = int(rng.integers(1, 4)), int(rng.integers(1, 260)), int(rng.integers(1, 260))
        qq, kk, vv = (q(torch.randn(BH, n, D, generator=g)) for n in (Nq, Nk, Nk))
        do = q(torch.randn(BH, Nq, D, generator=g))
        qr, kr, vr = (t.clone().requires_grad_(True) for t in (qq, kk, vv))
        L.sdpa_core(qr, kr, vr, D ** -0.5).backward(do)
        qd, kd, vd, dod = (t.to(DEV, BF).contiguous() for t in (qq, kk, vv, do))
        lse = torch.empty(BH, 1, Nq, dtype=torch.float32, device=DEV)
        sq, sk = (Nq * D, 0, D), (Nk * D, 0, D)
        ops.attention(qd, kd, vd, BH, 1, Nq, Nk, D, D ** -0.5, sq, sk, sk, lse=lse)
        dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
        ops.attention_bwd(qd, kd, vd, dod, lse, BH, 1, Nq, Nk, D, D ** -0.5, sq, sk, sk, dq, dk, dv, sq, sk, sk)
        tag = f"fuzz attn bwd {BH}x{Nq}x{Nk}x{D}"
        check_close(dv, vr.grad, rl2=1.2e-2, mabs=5e-2, what="dV " + tag)
        check_close(dk, kr.grad, rl2=2e-2, mabs=6e-2, what="dK " + tag)
        check_close(dq, qr.grad, rl2=2e-2, mabs=6e-2, what="dQ " + tag)
    for i in range(8):
        B, HW, C = int(rng.integers(1, 4)), int(rng.integers(2, 300)), 32 * int(rng.integers(1, 12))
        silu = i % 2 == 0
        x = q(torch.randn(B, HW, C, generator=g) * 1.3 + 0.2)
        gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
        dy = q(torch.randn(B, HW, C, generator=g))
        xr = x.clone().requires_grad_(True)
        y = F.group_norm(xr.permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
        (F.silu(y) if silu else y).backward(dy)
        dx = ops.groupnorm_bwd(x.reshape(B * HW, C).to(DEV, BF), gamma.to(DEV), beta.to(DEV), dy.reshape(B * HW, C).to(DEV, BF), B, HW, 1e-5, silu=silu)
        dx = dx[0] if isinstance(dx, tuple) else dx
        check_close(dx.reshape(B, HW, C), xr.grad, rl2=1.5e-2, mabs=6e-2, what=f"fuzz groupnorm bwd B={B} HW={HW} C={C} silu={silu}")
    for i in range(8):
        M, C = int(rng.integers(1, 600)), 8 * int(rng.integers(1, 160))
        x = q(torch.randn(M, C, generator=g) * 1.5 + 0.3)
        gamma = torch.randn(C, generator=g)
        dy = q(torch.randn(M, C, generator=g))
        xr = x.clone().requires_grad_(True)
        F.layer_norm(xr, (C,), gamma, torch.zeros(C), 1e-5).backward(dy)
        dx = ops.layernorm_bwd(x.to(DEV, BF), gamma.to(DEV), dy.to(DEV, BF), 1e-5)
        dx = dx[0] if isinstance(dx, tuple) else dx
        check_close(dx, xr.grad, rl2=1.5e-2, mabs=6e-2, what=f"fuzz layernorm bwd {M}x{C}")


# ------------------------------------------------------------------------------------------------- tape: GEMM / conv adjoints
def test_tape_gemm_and_conv_adjoints(ops):
    """Data-gradients through ops.gemm (two-source K split, residual) and ops.conv3x3 (stride 1, nearest-x2 upsample, stride 2),
    produced by the tape with the forward kernels on transposed / rotated weights."""
    from anyedit_amd.autodiff import Tape
    g = torch.Generator().manual_seed(3)
    B, H, W, C1, C2, Co = 2, 16, 16, 64, 128, 64
    x1, x2 = q(torch.randn(B * H * W, C1, generator=g)), q(torch.randn(B * H * W, C2, generator=g))
    wl = q(torch.randn(Co, C1 + C2, generator=g) / (C1 + C2) ** 0.5)
    wc = q(torch.randn(Co, Co, 3, 3, generator=g) / (9 * Co) ** 0.5)
    wu = q(torch.randn(128, Co, 3, 3, generator=g) / (9 * Co) ** 0.5)
    wd = q(torch.randn(64, 128, 3, 3, generator=g) / (9 * 128) ** 0.5)
    dy = q(torch.randn(B * H * W, 64, generator=g))

    # oracle: NCHW torch graph
    x1r, x2r = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    hlin = torch.cat([x1r, x2r], 1) @ wl.t()                                         # 1x1 over the channel concat
    hn = hlin.reshape(B, H, W, Co).permute(0, 3, 1, 2)
    h1 = F.conv2d(hn, wc, padding=1) + hn                                           # stride-1 conv + residual
    h2 = F.conv2d(F.interpolate(h1, scale_factor=2, mode="nearest"), wu, padding=1)  # Upsample
    h3 = F.conv2d(h2, wd, stride=2, padding=1)                                       # Downsample
    out_ref = h3.permute(0, 2, 3, 1).reshape(B * H * W, 64)
    out_ref.backward(dy)

    tape = Tape()
    x1d, x2d = x1.to(DEV, BF), x2.to(DEV, BF)
    tape.require(x1d)
    tape.require(x2d)
    with tape.recording():
        hl = ops.gemm(x1d, wl.to(DEV, BF), a2=x2d)
        c1, _, _ = ops.conv3x3(hl, ops.pack_conv3x3(wc.to(DEV)), None, B, H, W, residual=hl)
        c2, H2, W2 = ops.conv3x3(c1, ops.pack_conv3x3(wu.to(DEV)), None, B, H, W, upsample2x=True)
        c3, H3, W3 = ops.conv3x3(c2, ops.pack_conv3x3(wd.to(DEV)), None, B, H2, W2, stride=2)
    assert (H3, W3) == (H, W)
    check_close(c3, out_ref, rl2=8e-3, what="tape forward")
    tape.accumulate(c3, dy.to(DEV, BF))
    tape.backward()
    check_close(tape.grad(x1d), x1r.grad, rl2=1.2e-2, what="d x1 through conv chain")
    check_close(tape.grad(x2d), x2r.grad, rl2=1.2e-2, what="d x2 through conv chain")


def test_tape_trainable_linear_and_layernorm(ops):
    """Parameter gradients of a small trainable Linear (+bias) followed by LayerNorm: the image-projection head of the adapter."""
    from anyedit_amd.autodiff import Tape
    g = torch.Generator().manual_seed(4)
    M, K, N = 4, 1280, 4 * 768
    x = q(torch.randn(M, K, generator=g))
    w, b = q(torch.randn(N, K, generator=g) / K ** 0.5), torch.randn(N, generator=g) * 0.1
    gam, bet = 1 + 0.1 * torch.randn(768, generator=g), 0.1 * torch.randn(768, generator=g)
    dy = q(torch.randn(M * 4, 768, generator=g))
    wr, br, gr, ber = (t.clone().requires_grad_(True) for t in (w, b, gam, bet))
    y = F.layer_norm((x @ wr.t() + br).reshape(M * 4, 768), (768,), gr, ber, 1e-5)
    y.backward(dy)
    tape = Tape()
    wd, bd, gd, bed = w.to(DEV, BF), b.to(DEV), gam.to(DEV), bet.to(DEV)
    for t, n in ((wd, "w"), (bd, "b"), (gd, "gamma"), (bed, "beta")):
        tape.mark_trainable(t, n)
    with tape.recording():
        h = ops.gemm(x.to(DEV, BF), wd, bd)
        yy = ops.layernorm(h.reshape(M * 4, 768), gd, bed, 1e-5)
    check_close(yy, y, what="forward")
    tape.accumulate(yy, dy.to(DEV, BF))
    pg = tape.backward()
    check_close(pg["w"], wr.grad, rl2=1e-2, what="dW")
    check_close(pg["b"], br.grad, rl2=1e-2, what="db")
    check_close(pg["gamma"], gr.grad, rl2=1e-2, what="dgamma")
    check_close(pg["beta"], ber.grad, rl2=1e-2, what="dbeta")


def test_tape_never_writes_into_a_borrowed_gradient(ops):
    """the tape stores a FIRST gradient by reference (a residual's gradient is its consumer's dy) and backward kernels may add to an existing gradient
    in place (`Tape.into`).  In-place writes are allowed only into buffers the tape allocated itself: here x fans out into a LayerNorm branch and a residual add,
    the residual's gradient is the caller's `dy` tensor — which must come back bit-for-bit unchanged — and dx must still be the sum of both contributions."""
    from anyedit_amd.autodiff import Tape
    g = torch.Generator().manual_seed(12)
    M, C = 256, 320
    x = q(torch.randn(M, C, generator=g))
    w = q(torch.randn(C, C, generator=g) / C ** 0.5)
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy = q(torch.randn(M, C, generator=g))
    xr = x.clone().requires_grad_(True)
    (F.layer_norm(xr, (C,), gam, bet, 1e-5) @ w.t() + xr).backward(dy)
    tape = Tape()
    xd = x.to(DEV, BF)
    tape.require(xd)
    with tape.recording():
        h = ops.layernorm(xd, gam.to(DEV), bet.to(DEV), 1e-5)
        y = ops.gemm(h, w.to(DEV, BF), residual=xd)
    dyd = dy.to(DEV, BF)
    keep = dyd.clone()
    tape.accumulate(y, dyd)
    tape.backward()
    assert torch.equal(dyd, keep), "the backward pass wrote into the caller's gradient tensor"
    dx = tape.grad(xd)
    check_close(dx, xr.grad, rl2=1e-2, what="dx through LayerNorm + residual")
    assert dx.data_ptr() != dyd.data_ptr() and dx.data_ptr() in tape.owned


# ------------------------------------------------------------------------------------------------- small kernels
def test_sumpool_add_mse_grad_rowsum(ops):
    g = torch.Generator().manual_seed(6)
    x = q(torch.randn(2, 8, 8, 64, generator=g))
    ref = x.reshape(2, 4, 2, 4, 2, 64).sum((2, 4))
    check_close(ops.sumpool2x2(x.reshape(-1, 64).to(DEV, BF), 2, 4, 4).reshape(2, 4, 4, 64), ref, what="sumpool2x2")
    a, b = q(torch.randn(100, 64, generator=g)), q(torch.randn(100, 64, generator=g))
    assert torch.equal(ops.add(a.to(DEV, BF), b.to(DEV, BF)).float().cpu(), (a + b).to(BF).float())
    p, t = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    pr = p.clone().requires_grad_(True)
    F.mse_loss(pr, t).backward()
    check_close(ops.mse_grad(p.to(DEV), t.to(DEV)), pr.grad, rl2=1e-6, mabs=1e-6, what="mse grad")
    r = torch.randn(3, 5000, generator=g)
    check_close(ops.rowsum_f32(r.to(DEV)), r.sum(1), rl2=1e-5, mabs=1e-5, what="rowsum")
    for n in (32768, 4100, 1027, 3):   # the gate-gradient rows of the 64x64 level (16-byte-load path, four chains + a tail), a ragged vector count, scalar paths
        r = torch.randn(4, n, generator=g)
        check_close(ops.rowsum_f32(r.to(DEV)), r.double().sum(1).float(), rl2=1e-5, mabs=1e-5, what=f"rowsum n={n}")


def test_adamw_matches_torch(ops):
    g = torch.Generator().manual_seed(7)
    p0 = torch.randn(1000, generator=g)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    pd, m, v = p0.to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        pt.grad = gr.clone()
        opt.step()
        ops.adamw_step(pd, gr.to(DEV), m, v, step, 1e-3)
    check_close(pd, pt.detach(), rl2=1e-6, mabs=1e-5, what="adamw")


@pytest.mark.parametrize("fused,D", [(True, 40), (False, 160), (True, 160), (False, 40), (True, 80)])
def test_tape_adapter_attention_two_segments(ops, fused, D):
    """out = Attn(q, K, V) + g_b * Attn(q, K_ip, V_ip): the fused two-segment launch (head_dim <= 96, and 160 with one query
    fragment per wave) and the un-fused accumulate path, gradients w.r.t. q, the text K|V, the expert K|V and the gate against autograd."""
    from anyedit_amd.autodiff import Tape
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(9 + D)
    B, H, N, Nk, T = 2, 2, 96, 78, 4
    inner = H * D
    qq = q(torch.randn(B * N, inner, generator=g))
    kv = q(torch.randn(B * Nk, 2 * inner, generator=g))
    kvip = q(torch.randn(B * T, 2 * inner, generator=g))
    gate = torch.tensor([0.7, 0.3])
    dy = q(torch.randn(B, N, inner, generator=g))

    def heads(t, n):  # [B*n, inner] -> [B*H, n, D]
        return t.reshape(B, n, H, D).permute(0, 2, 1, 3).reshape(B * H, n, D)

    qr, kvr, kvipr, gr = (t.clone().requires_grad_(True) for t in (qq, kv, kvip, gate))
    a1 = L.sdpa_core(heads(qr, N), heads(kvr[:, :inner], Nk), heads(kvr[:, inner:], Nk), D ** -0.5)
    a2 = L.sdpa_core(heads(qr, N), heads(kvipr[:, :inner], T), heads(kvipr[:, inner:], T), D ** -0.5)
    out_ref = (a1 + gr.repeat_interleave(H).view(B * H, 1, 1) * a2).reshape(B, H, N, D).permute(0, 2, 1, 3).reshape(B, N, inner)
    out_ref.backward(dy)

    tape = Tape()
    qd, kvd, kvipd, gd = qq.to(DEV, BF), kv.to(DEV, BF), kvip.to(DEV, BF), gate.to(DEV)
    for t in (qd, kvd, kvipd, gd):
        tape.require(t)
    qs, ks, ks2 = (N * inner, D, inner), (Nk * 2 * inner, D, 2 * inner), (T * 2 * inner, D, 2 * inner)
    with tape.recording():
        if fused:
            o = ops.attention(qd, kvd, kvd[:, inner:], B, H, N, Nk, D, D ** -0.5, qs, ks, ks,
                              seg2=(kvipd, kvipd[:, inner:], T, ks2, ks2, gd))
        else:
            o = ops.attention(qd, kvd, kvd[:, inner:], B, H, N, Nk, D, D ** -0.5, qs, ks, ks)
            ops.attention(qd, kvipd, kvipd[:, inner:], B, H, N, T, D, D ** -0.5, qs, ks2, ks2, out=o, out_scale=gd, accumulate=True)
    check_close(o, out_ref, what="two-segment forward")
    tape.accumulate(o, dy.to(DEV, BF))
    tape.backward()
    check_close(tape.grad(qd), qr.grad, rl2=1.5e-2, mabs=5e-2, what="dq")
    check_close(tape.grad(kvd), kvr.grad, rl2=1.5e-2, mabs=5e-2, what="d kv (text)")
    check_close(tape.grad(kvipd), kvipr.grad, rl2=1.5e-2, mabs=5e-2, what="d kv (expert)")
    check_close(tape.grads[id(gd)], gr.grad, rl2=3e-2, mabs=5e-2, what="d gate")


@pytest.mark.parametrize("B,T,N,Dc,E", [(4, 4, 2560, 768, 11), (3, 4, 640, 768, 5), (16, 4, 1280, 768, 11), (2, 3, 64, 256, 2)])
def test_expert_kv_grouped_forward_dgrad_wgrad(ops, B, T, N, Dc, E):
    """Grouped per-expert adapter K/V projection (csrc/expert_kv.hip; AnySD adapters, train.py:410-424 / DESIGN.md §6) against
    per-sample fp32 matmuls on the bf16-rounded expert weights; the data gradient is bit-reproducible run to run."""
    g = torch.Generator().manual_seed(B * 1000 + N)
    x = torch.randn(B * T, Dc, generator=g).to(DEV).to(torch.bfloat16)
    W = (torch.randn(E, N, Dc, generator=g) * 0.05).to(DEV)
    dy = torch.randn(B * T, N, generator=g).to(DEV).to(torch.bfloat16)
    experts = torch.randint(0, E, (B,), generator=g).to(DEV).to(torch.int32)
    Wb = W.to(torch.bfloat16).float()
    xs, dys = x.float().view(B, T, Dc), dy.float().view(B, T, N)
    y_ref = torch.stack([xs[b] @ Wb[experts[b]].t() for b in range(B)]).view(B * T, N)
    dx_ref = torch.stack([dys[b] @ Wb[experts[b]] for b in range(B)]).view(B * T, Dc)
    dW_ref = torch.zeros(E, N, Dc, device=DEV)
    for b in range(B):
        dW_ref[experts[b]] += dys[b].t() @ xs[b]
    y = ops.expert_kv(x, W, experts, T)
    dx = ops.expert_kv_dgrad(dy, W, experts, T)
    dx2 = ops.expert_kv_dgrad(dy, W, experts, T)
    dW = ops.expert_kv_wgrad(dy, x, experts, T, E)
    assert rel_l2(y.float(), y_ref) <= 4e-3 and rel_l2(dx.float(), dx_ref) <= 4e-3      # bf16 output rounding
    assert torch.equal(dx, dx2)
    assert rel_l2(dW, dW_ref) <= 1e-5
    absent = [e for e in range(E) if e not in experts.tolist()]
    assert all(float(dW[e].abs().max()) == 0.0 for e in absent)
