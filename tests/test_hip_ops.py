"""GPU parity tests (run with -m gpu on an MI355X): every HIP operator, called through the C ABI, against the
oracle (CPU fp32 restatement pinned to reference goldens) on the same seeded inputs.

Tolerances (stated per the north-star "within a stated fp tolerance"):
  * integer / index / layout work: bit-exact;
  * fp32 elementwise DDIM arithmetic: bit-exact vs the unfused fp32 expression sequence;
  * bf16-in / fp32-accumulate kernels, inputs pre-rounded to bf16 on both sides: relative L2 error <= 4e-3 per operator
    (bf16 output rounding alone is ~1.1e-3) and max-abs error <= 2e-2 * max|ref|.
"""
import os

os.environ.setdefault("AE_ROWPANEL_ANY_M", "1")  # read once by the library: lets the small row-panel GEMM cases reach the kernel

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, sub_sd, T, rel_l2

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DEV = "cuda"


def q(t):
    """round to bf16 and back (what the HIP path sees)"""
    return t.to(BF).float()


def check_close(got, ref, rl2=4e-3, mabs=2e-2, what=""):
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
    assert "gfx950" in __import__("anyedit_amd._lib", fromlist=["x"]).device_arch()
    return o


# ------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (128, 128, 128), (200, 320, 320), (12, 1280, 320), (924, 640, 768),
                                    (4096, 320, 1280), (333, 36, 72), (1, 4, 8), (4096, 2560, 320)])
def test_gemm_plain_bias_residual(ops, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a, w = q(torch.randn(M, K, generator=g)), q(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    res = q(torch.randn(M, N, generator=g))
    ref = a @ w.t() + bias + res
    out = ops.gemm(a.to(DEV, BF), w.to(DEV, BF), bias=bias.to(DEV), residual=res.to(DEV, BF))
    check_close(out, ref, what=f"gemm {M}x{N}x{K}")
    out32 = ops.gemm(a.to(DEV, BF), w.to(DEV, BF), out_f32=True)
    check_close(out32, a @ w.t(), rl2=5e-4, mabs=2e-3, what="gemm fp32 out")


def test_gemm_transpose_detection(ops):
    """A = I with an ASYMMETRIC W: catches a swapped output fragment mapping."""
    K = N = 64
    a = torch.eye(64)
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 17) - 8.0
    out = ops.gemm(a.to(DEV, BF), w.to(DEV, BF), out_f32=True).cpu()
    assert torch.equal(out, w.t().contiguous())


def test_gemm_epilogues_and_two_source(ops):
    g = torch.Generator().manual_seed(5)
    M, K, N = 300, 192, 256
    a, w = q(torch.randn(M, K, generator=g)), q(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g)
    ad, wd = a.to(DEV, BF), w.to(DEV, BF)
    check_close(ops.gemm(ad, wd, bias=bias.to(DEV), epilogue=ops.EPI_SILU), F.silu(a @ w.t() + bias), what="silu")
    check_close(ops.gemm(ad, wd, bias=bias.to(DEV), epilogue=ops.EPI_GELU), F.gelu(a @ w.t() + bias), what="gelu")
    # GEGLU: reference chunk semantics (attention.py:49-58) through the interleaved packing
    wp, bp = ops.pack_geglu(w.to(DEV), bias.to(DEV))
    h = a @ w.t() + bias
    xx, gate = h.chunk(2, dim=-1)
    check_close(ops.gemm(ad, wp, bias=bp, epilogue=ops.EPI_GEGLU), xx * F.gelu(gate), what="geglu")
    # two-source A (deferred channel concat), split not a multiple of 64
    a1, a2 = a[:, :72].contiguous(), a[:, 72:].contiguous()
    check_close(ops.gemm(a1.to(DEV, BF), wd, a2=a2.to(DEV, BF)), a @ w.t(), what="two-source")
    # per-batch add vector
    addv = torch.randn(3, N, generator=g)
    ref = a @ w.t() + addv.repeat_interleave(100, dim=0)
    check_close(ops.gemm(ad, wd, addvec=addv.to(DEV), rows_per_batch=100), ref, what="addvec")
    # strided A view (columns of a wider buffer)
    wide = q(torch.randn(M, 3 * K, generator=g)).to(DEV, BF)
    check_close(ops.gemm(wide[:, K:2 * K], wd), wide[:, K:2 * K].float().cpu() @ w.t(), what="strided A")


@pytest.mark.parametrize("M,N,epi,ln,res", [(192, 64, "none", False, False), (500, 320, "none", True, True), (777, 960, "none", True, False),
                                             (1000, 256, "geglu", True, False), (4096, 2560, "geglu", False, False), (391, 320, "none", False, True), (49152, 320, "none", True, True)])
def test_rowpanel_ln_gemm(ops, M, N, epi, ln, res, monkeypatch):
    """Row-panel K = 320 kernel (ae_ln_gemm_bf16): LayerNorm prologue, bias, residual, GEGLU, ragged last block, against fp32 torch
    on the same bf16 inputs (attention.py:263-275 norm -> projection pairs).  The normalised rows are rounded to bf16 before the
    MFMA, as the unfused path stores them."""
    K = 320
    g = torch.Generator().manual_seed(M + N)
    a = q(torch.randn(M, K, generator=g) * 1.3 + 0.4)
    w = q(torch.randn(N, K, generator=g) / K ** 0.5)
    bias = torch.randn(N, generator=g) * 0.1
    gamma, beta = 1.0 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    r = q(torch.randn(M, N, generator=g)) if res else None
    x = q(F.layer_norm(a, (K,), gamma, beta, 1e-5)) if ln else a
    z = x @ w.t() + bias
    if epi == "geglu":
        xx, gate = z.chunk(2, dim=-1)
        ref = xx * F.gelu(gate)
        wd, bd = ops.pack_geglu(w.to(DEV), bias.to(DEV))
        e = ops.EPI_GEGLU
    else:
        ref = z + (r if res else 0.0)
        wd, bd, e = w.to(DEV, BF), bias.to(DEV), ops.EPI_NONE
    if M < 192 * 192:  # production launches want one block per CU; the small cases reach the kernel through its test hook
        assert ops.lib.ae_ln_gemm_supported(M, N, K, e) == 0 or os.environ.get("AE_ROWPANEL_ANY_M") == "1"
    assert os.environ.get("AE_ROWPANEL_ANY_M") == "1" and ops.lib.ae_ln_gemm_supported(M, N, K, e) == 1
    rd = None if r is None else r.to(DEV, BF)
    if ln:
        out = ops.ln_gemm(a.to(DEV, BF), gamma.to(DEV), beta.to(DEV), 1e-5, wd, bd, residual=rd, epilogue=e)
    else:
        out = ops.gemm(a.to(DEV, BF), wd, bias=bd, residual=rd, epilogue=e)  # routed to the row-panel kernel
    check_close(out, ref, what=f"rowpanel {M}x{N} epi={epi} ln={ln} res={res}")
    # the fallback (tiled GEMM + LayerNorm kernel) agrees with it
    import anyedit_amd.ops as O
    O._ROWPANEL = False
    try:
        out2 = ops.ln_gemm(a.to(DEV, BF), gamma.to(DEV), beta.to(DEV), 1e-5, wd, bd, residual=rd, epilogue=e) if ln else \
            ops.gemm(a.to(DEV, BF), wd, bias=bd, residual=rd, epilogue=e)
    finally:
        O._ROWPANEL = True
    check_close(out, out2.float().cpu(), rl2=6e-3, mabs=6e-2, what="rowpanel vs tiled")


@pytest.mark.parametrize("M,C,N,epi", [(12288, 640, 1920, "none"), (12288, 640, 640, "none"), (12288, 640, 5120, "geglu"), (3072, 1280, 3840, "none"),
                                       (3072, 1280, 1280, "none"), (3072, 1280, 10240, "geglu"), (12200, 640, 1920, "none"), (8192, 640, 5120, "geglu"),
                                       # round 5: the 64x64 level (K = 320) — the row-panel kernel's fold forms (producer: statistics from its chunk epilogue,
                                       # the two halves of a row group combined through LDS; consumers: qkv, q, GEGLU without the LayerNorm prologue)
                                       (49152, 320, 960, "none"), (49152, 320, 320, "none"), (49152, 320, 2560, "geglu"), (49000, 320, 960, "none")])
def test_layernorm_folded_into_gemm(ops, M, C, N, epi):
    """attention.py:263-275 norm -> projection at the widths the row-panel kernel does not cover (C = 640 / 1280), with the LayerNorm folded
    into the consuming GEMM (ae_gemm_ln_bf16): LN(x) W^T + b = rstd (x W'^T - mu s) + c.
      1. the GEMM that produces x emits the row statistics beside an output that is BIT-IDENTICAL to the plain launch's;
      2. the statistics are the per-row, per-64-column (sum, sum of squares) of the stored bf16 x (fp32, 1e-5 of sum |x| / sum x^2);
      3. the folded GEMM against the fp32 LayerNorm + Linear of the same bf16 x and the fp32 master weights: within the operator tolerance
         of this file, and no worse than the unfused HIP path (LayerNorm kernel, bf16 LN(x), plain GEMM) by more than 25 % + 5e-4 —
         rows carry a mean of up to three standard deviations, the case the subtraction mu * s has to survive."""
    E = ops.EPI_GEGLU if epi == "geglu" else ops.EPI_NONE
    if not (ops.ln_fold_plan(M, C, C, ops.EPI_NONE, 1) and ops.ln_fold_plan(M, N, C, E, 2)):
        pytest.skip("the tile plan of this shape carries no fold epilogue (AE_LN_FOLD / tile knobs)")
    g = torch.Generator().manual_seed(M + C + N)
    a = q(torch.randn(M, C, generator=g))
    wp = q(torch.randn(C, C, generator=g) / C ** 0.5)
    bp = 0.1 * torch.randn(C, generator=g)
    r = q(torch.randn(M, C, generator=g) * 1.3 + torch.randn(M, 1, generator=g) * 3.0)
    ad, wpd, bpd, rd = a.to(DEV, BF), wp.to(DEV, BF), bp.to(DEV), r.to(DEV, BF)
    st = ops.rowstats_buffer(M, C, DEV)
    st.fill_(float("nan"))
    x = ops.gemm(ad, wpd, bpd, residual=rd, rowstats=st)
    x_plain = ops.gemm(ad, wpd, bpd, residual=rd)
    assert torch.equal(x, x_plain), "the statistics epilogue changed the GEMM's output"
    check_close(x, a @ wp.t() + bp + r, what="producer")
    xf = x.float().cpu()
    xs = xf.reshape(M, C // 64, 64)
    got = st.cpu()
    assert torch.isfinite(got).all()
    assert float((got[..., 0] - xs.sum(-1)).abs().max()) <= 1e-5 * float(xs.abs().sum(-1).max())
    assert float((got[..., 1] - (xs * xs).sum(-1)).abs().max()) <= 1e-5 * float((xs * xs).sum(-1).max())
    # consumer
    w = torch.randn(N, C, generator=g) / C ** 0.5
    b = 0.1 * torch.randn(N, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    z = (F.layer_norm(xf.double(), (C,), gamma.double(), beta.double(), 1e-5) @ w.double().t() + b.double()).float()
    if epi == "geglu":
        xx, gate = z.chunk(2, dim=-1)
        ref = xx * F.gelu(gate)
        wd, bd = ops.pack_geglu(w.to(DEV), b.to(DEV))
    else:
        ref = z
        wd, bd = w.to(DEV, BF), b.to(DEV)
    wq, s, c = ops.pack_ln_fold(w.to(DEV), b.to(DEV), gamma.to(DEV), beta.to(DEV), geglu=epi == "geglu")
    assert wq.dtype == BF and wq.shape == (N, C) and s.shape == (N,) and c.shape == (N,)
    y = ops.gemm_ln(x, st, wq, s, c, 1e-5, epilogue=E)
    y_unfused = ops.ln_gemm(x, gamma.to(DEV), beta.to(DEV), 1e-5, wd, bd, epilogue=E)
    e_fold, e_unf = rel_l2(y.float().cpu(), ref), rel_l2(y_unfused.float().cpu(), ref)
    print(f"LN fold M={M} C={C} N={N} {epi}: rel-L2 folded {e_fold:.3e}, unfused {e_unf:.3e}")
    check_close(y, ref, rl2=5e-3, mabs=3e-2, what=f"folded LayerNorm {M}x{C}->{N} {epi}")
    assert e_fold <= 1.25 * e_unf + 5e-4, (e_fold, e_unf)
    # a launch-side refusal, not a wrong answer, for a shape whose plan has no such epilogue
    with pytest.raises(RuntimeError):
        ops._gemm_ln(x[:64], wq, c, None, E, None, None, st[:64].contiguous(), s, 1e-5)


def test_cross_attention_half_fused_inside_the_product():
    """The opt-in plan (AE_XATTN_FUSED=1) end to end: a 4-step, 3-branch-CFG edit of the bench model at 64x64 with the fused cross-attention launches (K/V images from
    prepare_conditioning) against the same edit on the default plan, with the distance of a second proven plan (feed-forward as two launches) measured in the same run as the yardstick;
    the switches are read once per process, so tools/xattn_module_check.py runs one child per setting.  The rest of the suite runs the product default (the switch off)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "xattn_module_check.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    print(r.stdout[-1500:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "[base] " in r.stdout and "AE_XATTN_FUSED=0: fused cross-attention launches in the profiled edit: 0" in r.stdout
    assert "AE_XATTN_FUSED=1: fused cross-attention launches in the profiled edit: 0" not in r.stdout and "AE_XATTN_FUSED=1" in r.stdout and " OK" in r.stdout


@pytest.mark.parametrize("B,HW_side,Cin,Cout", [(12, 16, 1280, 1280), (12, 8, 1280, 1280), (12, 16, 2560, 1280), (12, 8, 2560, 1280), (12, 16, 640, 1280), (4, 16, 1280, 1280)])
def test_groupnorm_folds_the_splitk_partials_bit_identical(ops, B, HW_side, Cin, Cout):
    """openaimodel.py:262-272 at the 16x16 / 8x8 levels (round 6): conv1's split-K plan stopped at its fp32 partials (ae_conv3x3_partials_bf16) and the GroupNorm + SiLU
    folding the K ranges, the bias and the time-embedding vector itself (ae_groupnorm_splitk_nhwc_bf16) against conv3x3 (with its reduce launch) followed by groupnorm:
    the same adds in the same order and one rounding -> bit-identical, with and without bias / vector, strided vector rows; refusals outside the envelope."""
    H = W = HW_side
    ops._GN_SPLITK, knob = True, ops._GN_SPLITK          # opt-in in the product (measured slower in the graph: ops.py); the operators are tested regardless
    try:
        _splitk_fold_case(ops, B, H, W, Cin, Cout)
    finally:
        ops._GN_SPLITK = knob


def _splitk_fold_case(ops, B, H, W, Cin, Cout):
    if not ops.conv3x3_gn_splitk_ok(B, H, W, Cin, Cout):
        pytest.skip(f"the plan does not cut K for B={B} {H}x{W} {Cin}->{Cout}")
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn(B * H * W, Cin, generator=g).bfloat16().to(DEV)
    w = ops.pack_conv3x3((torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(DEV))
    bias = torch.randn(Cout, generator=g).to(DEV)
    emb_all = torch.randn(B, 3 * Cout, generator=g).to(DEV)
    addvec = emb_all[:, Cout:2 * Cout]                               # strided rows, as the batched time-embedding projection hands them over
    gamma, beta = (1.0 + 0.1 * torch.randn(Cout, generator=g)).to(DEV), (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    for bz, av in ((bias, addvec), (None, addvec), (bias, None), (None, None)):
        for silu in (True, False):
            h, _, _ = ops.conv3x3(x, w, bz, B, H, W, addvec=av)
            ref = ops.groupnorm(h, gamma, beta, B, H * W, 1e-5, silu=silu)
            part, sk = ops.conv3x3_partials(x, w, B, H, W)
            assert part.shape == (sk, B * H * W, Cout) and 2 <= sk <= 8
            got = ops.groupnorm_splitk(part, bz, av, gamma, beta, B, H * W, 1e-5, silu=silu)
            assert torch.equal(got, ref), (bz is not None, av is not None, silu, float((got.float() - ref.float()).abs().max()))
    assert torch.equal(ops.groupnorm_splitk(part, None, None, gamma, beta, B, H * W, 1e-5), got), "run-to-run bit-equal"
    with pytest.raises(ValueError):
        ops.groupnorm_splitk(part[:, :, :Cout - 8], None, None, gamma, beta, B, H * W, 1e-5)          # not contiguous
    with pytest.raises(ValueError):
        ops.groupnorm_splitk(part, None, emb_all, gamma, beta, B, H * W, 1e-5)                          # vector of the wrong width
    assert ops.lib.ae_groupnorm_splitk_supported(B, 1024, Cout, 32, sk) == 0 and ops.lib.ae_groupnorm_splitk_supported(B, H * W, Cout, 32, 1) == 0
    assert not ops.conv3x3_gn_splitk_ok(12, 64, 64, 320, 320) and not ops.conv3x3_gn_splitk_ok(1, 8, 8, 64, 64)   # maps beyond 256 positions; a K loop too short to cut
    with pytest.raises(ValueError):
        ops.conv3x3_partials(x[:64, :64].contiguous(), ops.pack_conv3x3(torch.zeros(64, 64, 3, 3, device=DEV)), 1, 8, 8)   # nine K tiles: the plan does not cut K


@pytest.mark.parametrize("M,H,res", [(192, 1280, True), (500, 1280, True), (777, 64, False), (1000, 256, True), (49152, 1280, True), (49000, 1280, True)])
def test_feed_forward_fused_one_launch(ops, M, H, res):
    """attention.py:49-76 behind norm3 (:271-275) as ONE launch (ae_ff_fused_bf16, round 6): LayerNorm -> GEGLU projection -> exact-erf gate -> ff2
    + bias + residual with the gated hidden activation kept in registers.
      1. against the fp64 module arithmetic on the same bf16 inputs and fp32 master weights (operator tolerance of this file);
      2. no worse than the two-launch HIP path it replaces (row-panel LayerNorm + GEGLU projection, then ff2) by more than 25 % + 5e-4;
      3. run-to-run bit-equal (fixed summation order); ragged last block; H down to two 32-unit steps."""
    C = 320
    assert ops.lib.ae_ff_fused_supported(M, C, H) == 1, "AE_ROWPANEL_ANY_M lets the small cases reach the kernel"
    g = torch.Generator().manual_seed(M + H)
    x = q(torch.randn(M, C, generator=g) * 1.3 + torch.randn(M, 1, generator=g) * 2.0)
    w1 = torch.randn(2 * H, C, generator=g) / C ** 0.5
    b1 = 0.1 * torch.randn(2 * H, generator=g)
    w2 = torch.randn(C, H, generator=g) / H ** 0.5
    b2 = 0.1 * torch.randn(C, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    z = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5) @ q(w1).double().t() + b1.double()
    a, gate = z.chunk(2, dim=-1)
    ref = ((a * F.gelu(gate)) @ q(w2).double().t() + b2.double() + (x.double() if res else 0.0)).float()
    xd = x.to(DEV, BF)
    w1p, b1p = ops.pack_geglu(w1.to(DEV), b1.to(DEV))
    w2img = ops.pack_ff2_fused(w2.to(DEV))
    assert w2img.shape == (H // 32, C, 32) and w2img.dtype == BF
    y = ops.ff_fused(xd, gamma.to(DEV), beta.to(DEV), 1e-5, w1p, b1p, w2img, b2.to(DEV), residual=xd if res else None)
    h = ops.ln_gemm(xd, gamma.to(DEV), beta.to(DEV), 1e-5, w1p, b1p, epilogue=ops.EPI_GEGLU)
    y2 = ops.gemm(h, w2.to(DEV, BF), bias=b2.to(DEV), residual=xd if res else None)
    e1, e2 = rel_l2(y.float().cpu(), ref), rel_l2(y2.float().cpu(), ref)
    print(f"fused feed-forward M={M} H={H}: rel-L2 fused {e1:.3e}, two launches {e2:.3e}")
    check_close(y, ref, rl2=5e-3, mabs=3e-2, what=f"fused feed-forward {M}x{H}")
    assert e1 <= 1.25 * e2 + 5e-4, (e1, e2)
    y3 = ops.ff_fused(xd, gamma.to(DEV), beta.to(DEV), 1e-5, w1p, b1p, w2img, b2.to(DEV), residual=xd if res else None)
    assert torch.equal(y, y3), "the fused feed-forward is not run-to-run bit-equal"
    # with the projection that follows the block in the same launch (SpatialTransformer.proj_out + its residual, attention.py:337-340): the bf16-rounded
    # feed-forward result times W3^T — the same arithmetic as the separate launch on the stored result, so the two agree to the order of an fp32 sum
    w3 = q(torch.randn(C, C, generator=g) / C ** 0.5)
    b3 = 0.1 * torch.randn(C, generator=g)
    r3 = q(torch.randn(M, C, generator=g))
    cs = ops.colstats_buffer(M, C, DEV)
    cs.fill_(float("nan"))
    z3 = ops.ff_fused(xd, gamma.to(DEV), beta.to(DEV), 1e-5, w1p, b1p, w2img, b2.to(DEV), residual=xd if res else None, w3=w3.to(DEV, BF), b3=b3.to(DEV),
                      residual3=r3.to(DEV, BF), colstats=cs)
    z3_two = ops.gemm(y, w3.to(DEV, BF), bias=b3.to(DEV), residual=r3.to(DEV, BF))
    ref3 = y.float().cpu().double() @ w3.double().t() + b3.double() + r3.double()
    check_close(z3, ref3.float(), what=f"fused feed-forward + proj_out {M}x{H}")
    e3 = rel_l2(z3.float().cpu(), z3_two.float().cpu())
    assert e3 <= 2.5e-3, f"fused proj_out tail vs its own launch: {e3:.3e}"
    zs = z3.float().cpu()
    pad = (-M) % 32
    zp = torch.cat([zs, torch.zeros(pad, C)]).reshape(-1, 32, C)
    got = cs.cpu()
    assert torch.isfinite(got).all()
    assert float((got[..., 0] - zp.sum(1)).abs().max()) <= 1e-4 * float(zp.abs().sum(1).max())
    assert float((got[..., 1] - (zp * zp).sum(1)).abs().max()) <= 1e-4 * float((zp * zp).sum(1).max())
    z4 = ops.ff_fused(xd, gamma.to(DEV), beta.to(DEV), 1e-5, w1p, b1p, w2img, b2.to(DEV), residual=xd if res else None, w3=w3.to(DEV, BF), b3=b3.to(DEV),
                      residual3=r3.to(DEV, BF))
    assert torch.equal(z3, z4), "the fused feed-forward + proj_out is not run-to-run bit-equal"


@pytest.mark.parametrize("B,N,Nk,T", [(1, 128, 78, 4), (2, 256, 77, 0), (3, 384, 80, 16), (12, 4096, 78, 4)])
def test_cross_attention_half_fused_one_launch(ops, B, N, Nk, T):
    """attention.py:273 `x = attn2(norm2(x), context) + x` as ONE launch (ae_xattn_fused_bf16, round 6): LayerNorm -> to_q -> softmax(QK^T)V over the text keys (+ the gated expert
    segment of DESIGN.md §6) -> to_out + bias + residual, with q / logits / probabilities / attention output kept in registers.
      1. against the fp64 module arithmetic on the same bf16 inputs (operator tolerance);
      2. no worse than the three-launch HIP path it replaces (LayerNorm + q projection, short-K/V attention with the second segment, to_out + residual) by more than 25 % + 5e-4;
      3. run-to-run bit-equal; key counts 77 / 78 / 80 (padding masks), no expert segment, a full 16-key expert segment; samples never mix (per-sample K | V and gate)."""
    C, H, D = 320, 8, 40
    M = B * N
    # the kernel is opt-in in the MODULE (AE_XATTN_FUSED, default off: measured slower); this operator test calls it directly, the rest of the suite runs the product default
    assert ops.lib.ae_xattn_fused_covers(M, C, H, D, N, Nk, T) == 1 and ops.lib.ae_xattn_fused_supported(M, C, H, D, N, Nk, T) == int(os.environ.get("AE_XATTN_FUSED", "0") != "0")
    g = torch.Generator().manual_seed(B * 1000 + Nk + T)
    x = q(torch.randn(M, C, generator=g) * 1.3 + torch.randn(M, 1, generator=g) * 2.0)
    wq = torch.randn(C, C, generator=g) / C ** 0.5
    wo = torch.randn(C, C, generator=g) / C ** 0.5
    bo = 0.1 * torch.randn(C, generator=g)
    gamma, beta = 1.0 + 0.2 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    kv = q(torch.randn(B * Nk, 2 * C, generator=g))
    kv_ip = q(torch.randn(B * T, 2 * C, generator=g)) if T else None
    gate = torch.rand(B, generator=g) if T else None
    scale = D ** -0.5
    # fp64 reference
    xn = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5)
    qq = (xn @ q(wq).double().t()).reshape(B, N, H, D).permute(0, 2, 1, 3)
    kk = kv.double().reshape(B, Nk, 2, H, D)
    att = torch.softmax(qq @ kk[:, :, 0].permute(0, 2, 3, 1) * scale, -1) @ kk[:, :, 1].permute(0, 2, 1, 3)
    if T:
        ki = kv_ip.double().reshape(B, T, 2, H, D)
        att = att + gate.double()[:, None, None, None] * (torch.softmax(qq @ ki[:, :, 0].permute(0, 2, 3, 1) * scale, -1) @ ki[:, :, 1].permute(0, 2, 1, 3))
    ref = (att.permute(0, 2, 1, 3).reshape(M, C) @ q(wo).double().t() + bo.double() + x.double()).float()
    xd, gd, bd = x.to(DEV, BF), gamma.to(DEV), beta.to(DEV)
    wq_img, wo_img = ops.pack_xattn_wq(wq.to(DEV)), ops.pack_xattn_wo(wo.to(DEV))
    kv_img = ops.pack_xattn_kv(kv.to(DEV, BF), None if kv_ip is None else kv_ip.to(DEV, BF), B, Nk, T)
    assert kv_img.shape == (B, 8, ops.lib.ae_xattn_fused_kv_bytes() // 2)
    gated = None if gate is None else gate.to(DEV)
    y = ops.xattn_fused(xd, gd, bd, 1e-5, wq_img, kv_img, gated, wo_img, bo.to(DEV), N, Nk, T, scale)
    # the three launches it replaces
    qd = ops.ln_gemm(xd, gd, bd, 1e-5, wq.to(DEV, BF))
    kvd = kv.to(DEV, BF)
    qs, ks = (N * C, D, C), (Nk * 2 * C, D, 2 * C)
    if T:
        kid = kv_ip.to(DEV, BF)
        ksi = (T * 2 * C, D, 2 * C)
        o = ops.attention(qd, kvd, kvd[:, C:], B, H, N, Nk, D, scale, qs, ks, ks, seg2=(kid, kid[:, C:], T, ksi, ksi, gated))
    else:
        o = ops.attention(qd, kvd, kvd[:, C:], B, H, N, Nk, D, scale, qs, ks, ks)
    y3 = ops.gemm(o.reshape(M, C), wo.to(DEV, BF), bias=bo.to(DEV), residual=xd)
    e1, e3 = rel_l2(y.float().cpu(), ref), rel_l2(y3.float().cpu(), ref)
    print(f"fused cross-attention half B={B} N={N} Nk={Nk}+{T}: rel-L2 fused {e1:.3e}, three launches {e3:.3e}")
    check_close(y, ref, rl2=5e-3, mabs=3e-2, what=f"fused cross-attention half B={B} N={N} Nk={Nk}+{T}")
    assert e1 <= 1.25 * e3 + 5e-4, (e1, e3)
    y2 = ops.xattn_fused(xd, gd, bd, 1e-5, wq_img, kv_img, gated, wo_img, bo.to(DEV), N, Nk, T, scale)
    assert torch.equal(y, y2), "the fused cross-attention half is not run-to-run bit-equal"


@pytest.mark.parametrize("B,H,W,Cout", [(2, 16, 16, 64), (1, 5, 7, 320), (12, 64, 64, 320)])
def test_stem_conv_as_im2col_gemm(ops, B, H, W, Cout, monkeypatch):
    """The UNet's 8-channel stem conv (openaimodel.py:536-542) as `im2col3x3_c8` + a K = 128 dense GEMM: the im2col rows bit-exact against the
    definition of a zero-padded 3x3 gather, the conv against fp32 F.conv2d on the same bf16 operands and against the implicit-GEMM kernel."""
    from anyedit_amd.ldm.modules.diffusionmodules.util import Conv2d
    import anyedit_amd.ops as O
    g = torch.Generator().manual_seed(B + H + Cout)
    x = q(torch.randn(B, 8, H, W, generator=g))
    rows = ops.nchw_to_rows(x.to(DEV))
    col = ops.im2col3x3_c8(rows, B, H, W).cpu()
    xp = F.pad(x, (1, 1, 1, 1))
    ref = torch.zeros(B, H, W, 128)
    for tap in range(9):
        ref[..., 8 * tap:8 * tap + 8] = xp[:, :, tap // 3:tap // 3 + H, tap % 3:tap % 3 + W].permute(0, 2, 3, 1)
    assert torch.equal(col.float().reshape(B, H, W, 128), ref), "im2col rows differ from the zero-padded gather"
    conv = Conv2d(8, Cout, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(q(torch.randn(Cout, 8, 3, 3, generator=g) / 72 ** 0.5))
        conv.bias.copy_(torch.randn(Cout, generator=g) * 0.1)
    want = F.conv2d(x, conv.weight.detach(), conv.bias.detach(), padding=1)
    conv = conv.to(DEV)
    assert O.stem_im2col_ok(8, 1, False)
    y, _, _ = conv.rows(rows, B, H, W)
    check_close(ops.rows_to_nchw(y, B, H, W), want, what=f"stem conv via im2col [{B},8,{H},{W}] -> {Cout}")
    monkeypatch.setattr(O, "_STEM_IM2COL", False)
    y0, _, _ = conv.rows(rows, B, H, W)
    check_close(y, y0.float().cpu(), rl2=3e-3, mabs=2e-2, what="im2col path vs implicit GEMM")


def test_gemm_linearity_full_size(ops):
    """Size-independent property at a BASELINE-size shape: f(a1 + a2) == f(a1) + f(a2) up to rounding."""
    g = torch.Generator(device=DEV).manual_seed(1)
    M, K, N = 12 * 4096, 320, 320
    a1 = torch.randn(M, K, generator=g, device=DEV).to(BF)
    a2 = torch.randn(M, K, generator=g, device=DEV).to(BF)
    w = (torch.randn(N, K, generator=g, device=DEV) / K ** 0.5).to(BF)
    s = (a1.float() + a2.float()).to(BF)
    y = ops.gemm(s, w, out_f32=True)
    y12 = ops.gemm(a1, w, out_f32=True) + ops.gemm(a2, w, out_f32=True)
    assert rel_l2(y.cpu(), y12.cpu()) < 6e-3  # bf16 rounding of the summed input only
    ref = s.float() @ w.float().t()
    assert rel_l2(y.cpu(), ref.cpu()) < 5e-4


# ------------------------------------------------------------------------------------------------- seeded shape fuzz
def test_fuzz_gemm_conv_attention_shapes(ops):
    """Seeded random ragged shapes through the three MFMA kernels against fp32 torch on the same bf16-rounded inputs: every tile /
    split-K / tail / mask path gets exercised by sizes nobody hand-picked (M, N, K not multiples of the tiles; 1-row problems;
    keys shorter than one tile; fully masked-out tails)."""
    from oracle import ldm_ref as L
    rng = np.random.default_rng(2024)
    g = torch.Generator().manual_seed(2024)
    for i in range(24):                                                                 # GEMM
        M, N, K = int(rng.integers(1, 700)), 4 * int(rng.integers(1, 170)), 8 * int(rng.integers(1, 110))
        a, w = q(torch.randn(M, K, generator=g)), q(torch.randn(N, K, generator=g) / K ** 0.5)
        bias = torch.randn(N, generator=g) if i % 2 else None
        res = q(torch.randn(M, N, generator=g)) if i % 3 == 0 else None
        epi = [ops.EPI_NONE, ops.EPI_GELU, ops.EPI_SILU, ops.EPI_RELU][i % 4]
        z = a @ w.t() + (bias if bias is not None else 0.0)
        z = {ops.EPI_NONE: z, ops.EPI_GELU: F.gelu(z), ops.EPI_SILU: F.silu(z), ops.EPI_RELU: F.relu(z)}[epi]
        ref = z + (res if res is not None else 0.0)
        out = ops.gemm(a.to(DEV, BF), w.to(DEV, BF), bias=None if bias is None else bias.to(DEV),
                       residual=None if res is None else res.to(DEV, BF), epilogue=epi)
        check_close(out, ref, rl2=5e-3, mabs=3e-2, what=f"fuzz gemm {M}x{N}x{K} epi={epi}")
    for i in range(14):                                                                 # conv3x3
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 21)), int(rng.integers(1, 21))
        Cin, Cout = 8 * int(rng.integers(1, 41)), 8 * int(rng.integers(1, 41))
        stride = 2 if (i % 4 == 1 and H > 1 and W > 1) else 1
        ups = i % 4 == 2
        x = q(torch.randn(B, Cin, H, W, generator=g))
        w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
        bias = torch.randn(Cout, generator=g)
        xi = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
        ref = F.conv2d(xi, w, bias, stride=stride, padding=1)
        y, Ho, Wo = ops.conv3x3(ops.nchw_to_rows(x.to(DEV)), ops.pack_conv3x3(w.to(DEV)), bias.to(DEV), B, H, W, stride=stride, upsample2x=ups)
        assert (Ho, Wo) == tuple(ref.shape[2:])
        check_close(ops.rows_to_nchw(y, B, Ho, Wo), ref, rl2=5e-3, mabs=3e-2, what=f"fuzz conv B={B} {H}x{W} {Cin}->{Cout} s{stride} ups={ups}")
    for i in range(20):                                                                 # attention (+ key mask)
        D = [8, 16, 32, 40, 48, 64, 80, 96, 128, 160][i % 10]
        BH, Nq, Nk = int(rng.integers(1, 5)), int(rng.integers(1, 300)), int(rng.integers(1, 300))
        qq, kk, vv = (q(torch.randn(BH, n, D, generator=g)) for n in (Nq, Nk, Nk))
        mask = None
        if i % 3 == 0:
            mask = torch.rand(BH, Nk, generator=g) > 0.4
            mask[:, int(rng.integers(0, Nk))] = True                                    # at least one visible key per row
        logits = (qq @ kk.transpose(1, 2)) * D ** -0.5
        if mask is not None:
            logits = logits.masked_fill(~mask[:, None, :], -torch.finfo(torch.float32).max)
        ref = torch.softmax(logits, -1) @ vv
        out = ops.attention_bhnd(qq.to(DEV, BF), kk.to(DEV, BF), vv.to(DEV, BF),
                                 key_mask=None if mask is None else mask.to(DEV).to(torch.uint8).contiguous())
        check_close(out, ref, rl2=7e-3, mabs=3e-2, what=f"fuzz attention BH={BH} {Nq}x{Nk} d={D} mask={mask is not None}")


def test_fuzz_norm_shapes(ops):
    """Seeded random GroupNorm(+SiLU, +two-source concat) / LayerNorm / narrow LayerNorm+GELU shapes against fp32 torch."""
    rng = np.random.default_rng(77)
    g = torch.Generator().manual_seed(77)
    for i in range(16):
        B, HW, C = int(rng.integers(1, 5)), int(rng.integers(1, 400)), 32 * int(rng.integers(1, 21))
        x = q(torch.randn(B, HW, C, generator=g) * float(rng.uniform(0.2, 3.0)) + float(rng.uniform(-1, 1)))
        gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
        eps = 1e-5 if i % 2 else 1e-6
        silu = i % 3 != 0
        ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, eps).permute(0, 2, 1)
        ref = F.silu(ref) if silu else ref
        rows = x.reshape(B * HW, C).to(DEV, BF)
        C1 = 8 * int(rng.integers(1, C // 8)) if (i % 4 == 1 and C > 8) else None
        if C1 is None:
            out = ops.groupnorm(rows, gamma.to(DEV), beta.to(DEV), B, HW, eps, silu=silu)
        else:
            out = ops.groupnorm(rows[:, :C1].contiguous(), gamma.to(DEV), beta.to(DEV), B, HW, eps, silu=silu, x2=rows[:, C1:].contiguous())
        check_close(out.reshape(B, HW, C), ref, rl2=6e-3, mabs=3e-2, what=f"fuzz groupnorm B={B} HW={HW} C={C} C1={C1} silu={silu}")
    for i in range(12):
        M, C = int(rng.integers(1, 900)), 8 * int(rng.integers(1, 200))
        x = q(torch.randn(M, C, generator=g) * 1.5 + 0.3)
        gamma, beta = torch.randn(C, generator=g), torch.randn(C, generator=g)
        check_close(ops.layernorm(x.to(DEV, BF), gamma.to(DEV), beta.to(DEV), 1e-5), F.layer_norm(x, (C,), gamma, beta, 1e-5),
                    rl2=5e-3, mabs=3e-2, what=f"fuzz layernorm {M}x{C}")
        if C <= 512:
            check_close(ops.layernorm_act(x.to(DEV, BF), gamma.to(DEV), beta.to(DEV), eps=1e-6, gelu=True),
                        F.gelu(F.layer_norm(x, (C,), gamma, beta, 1e-6)), rl2=5e-3, mabs=3e-2, what=f"fuzz layernorm_act {M}x{C}")


# ------------------------------------------------------------------------------------------------- conv3x3
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,ups", [(2, 8, 8, 64, 64, 1, False), (1, 16, 16, 8, 32, 1, False),
                                                       (2, 8, 8, 96, 64, 2, False), (1, 7, 9, 64, 64, 2, False),
                                                       (2, 8, 8, 64, 64, 1, True), (1, 32, 32, 320, 4, 1, False),
                                                       (1, 64, 64, 320, 320, 1, False), (3, 5, 6, 128, 72, 1, False),
                                                       # UNet-batch-12 16x16 level: M = 3072, the split-K plan under the 192x320 tile
                                                       (12, 16, 16, 256, 1280, 1, False), (12, 16, 16, 320, 640, 1, False)])
def test_conv3x3(ops, B, H, W, Cin, Cout, stride, ups):
    g = torch.Generator().manual_seed(B + H * 3 + Cin + Cout + stride)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    bias = torch.randn(Cout, generator=g)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    ref = F.conv2d(xi, w, bias, stride=stride, padding=1)
    rows = ops.nchw_to_rows(x.to(DEV))
    y, Ho, Wo = ops.conv3x3(rows, ops.pack_conv3x3(w.to(DEV)), bias.to(DEV), B, H, W, stride=stride, upsample2x=ups)
    assert (Ho, Wo) == tuple(ref.shape[2:])
    check_close(ops.rows_to_nchw(y, B, Ho, Wo), ref, what="conv3x3")


@pytest.mark.parametrize("B,H,W,Cin,Cout,stats", [(12, 32, 32, 640, 640, True),      # UNet 32x32 -> 64x64: 192x320 ping-pong tile, 4 x 128 tiles, statistics from the epilogue
                                                  (12, 32, 32, 640, 640, False),
                                                  (12, 16, 16, 1280, 1280, True),    # 16x16 -> 32x32: 4 x 64 tiles = one round of the chip
                                                  (12, 8, 8, 1280, 1280, False),     # 8x8 -> 16x16: 128x128 weights-ahead tile (240 blocks)
                                                  (3, 16, 16, 64, 320, True),        # small grid on the 128x128 tile, one K tile per tap
                                                  (2, 6, 10, 128, 72, False),        # ragged: M = 120 (tile tail rows), N = 72 (column tail), W not a power of two
                                                  (1, 8, 8, 64, 64, True)])
def test_conv3x3_up2_as_four_2x2_convs(ops, B, H, W, Cin, Cout, stats):
    """Round 5 — Upsample (openaimodel.py:108-118) as four 2x2 convs on the low-resolution map (`ae_conv3x3_up2_bf16`): against torch's
    interpolate + conv2d on the bf16-rounded input and ORIGINAL weights (the summed weights are rounded once more: the same 4e-3 operator bound must
    hold), against the gather form of `ae_conv3x3_bf16(upsample2x = 1)` it replaces, and — where asked — the producer statistics of the stored map."""
    g = torch.Generator().manual_seed(B + H * 3 + Cin + Cout + 500)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    bias = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), q(w), bias, padding=1)
    rows = ops.nchw_to_rows(x.to(DEV))
    w4 = ops.pack_conv3x3_up2(w.to(DEV))
    cs = None
    if stats:
        cs = ops.colstats_buffer(B * 4 * H * W, Cout, DEV)
        cs.fill_(float("nan"))
    y, Ho, Wo = ops.conv3x3_up2(rows, w4, bias.to(DEV), B, H, W, colstats=cs)
    assert (Ho, Wo) == (2 * H, 2 * W) and tuple(y.shape) == (B * 4 * H * W, Cout)
    check_close(ops.rows_to_nchw(y, B, Ho, Wo), ref, what="conv3x3_up2")
    y_old, _, _ = ops.conv3x3(rows, ops.pack_conv3x3(w.to(DEV)), bias.to(DEV), B, H, W, upsample2x=True)
    assert rel_l2(y.float().cpu(), y_old.float().cpu()) < 4e-3
    assert torch.equal(ops.conv3x3_up2(rows, w4, bias.to(DEV), B, H, W)[0], y), "asking for statistics must not change the output; launches must repeat bit for bit"
    if stats:
        # the GroupNorm that consumes them folds every slab of a SAMPLE: per-sample sums are what must match (which 32 rows a slab holds is the producer's choice)
        got = cs.cpu().double()
        assert torch.isfinite(got).all()
        yf = y.float().cpu().double().reshape(B, 4 * H * W, Cout)
        ref_s = torch.stack([yf.sum(1), (yf * yf).sum(1)], -1)
        got_s = got.reshape(B, 4 * H * W // 32, Cout, 2).sum(1)
        assert float((got_s - ref_s).abs().max()) <= 2e-4 * float(ref_s.abs().max())
        gamma, beta = torch.randn(Cout, generator=g).to(DEV), torch.randn(Cout, generator=g).to(DEV)
        if Cout % 32 == 0 and 4 * H * W > 256:
            n_own = ops.groupnorm(y, gamma, beta, B, 4 * H * W, 1e-5, silu=True)
            n_cs = ops.groupnorm(y, gamma, beta, B, 4 * H * W, 1e-5, silu=True, colstats=cs)
            assert rel_l2(n_cs.float().cpu(), n_own.float().cpu()) < 2e-3


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride", [(2, 8, 8, 64, 64, 1), (1, 7, 9, 64, 64, 2), (3, 5, 6, 128, 72, 1), (1, 64, 64, 320, 320, 1),
                                                   (12, 16, 16, 256, 1280, 1), (12, 16, 16, 320, 640, 1), (12, 8, 8, 1280, 1280, 1)])
def test_conv3x3_chunk_major_k_order(ops, B, H, W, Cin, Cout, stride):
    """K ordered (64-channel chunk, tap, channel) instead of (tap, channel) — `k_order=1` with the matching weight pack — on every tile
    plan (64x64, 128x128, the three-stage ring with split-K, the split 192x320 tile): the same convolution, another accumulation order."""
    g = torch.Generator().manual_seed(B + H * 3 + Cin + Cout + stride + 100)
    x = q(torch.randn(B, Cin, H, W, generator=g))
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5))
    bias, emb = torch.randn(Cout, generator=g), torch.randn(B, Cout, generator=g)
    ref = F.conv2d(x, w, bias, stride=stride, padding=1) + emb[:, :, None, None]
    rows = ops.nchw_to_rows(x.to(DEV))
    y1, Ho, Wo = ops.conv3x3(rows, ops.pack_conv3x3(w.to(DEV), k_order=1), bias.to(DEV), B, H, W, addvec=emb.to(DEV), stride=stride, out_f32=True, k_order=1)
    y0, _, _ = ops.conv3x3(rows, ops.pack_conv3x3(w.to(DEV)), bias.to(DEV), B, H, W, addvec=emb.to(DEV), stride=stride, out_f32=True)
    check_close(ops.rows_to_nchw(y1, B, Ho, Wo), ref, rl2=1e-3, what="conv3x3 chunk-major K order")
    assert rel_l2(y1.cpu(), y0.cpu()) <= 2e-6, "the two K orders differ only by the order of the fp32 accumulation"
    with pytest.raises(RuntimeError, match="Cin %% 64|Cin % 64"):
        ops.conv3x3(ops.nchw_to_rows(x[:, :8].contiguous().to(DEV)), ops.pack_conv3x3(w[:, :8].contiguous().to(DEV)), None, B, H, W, stride=stride, k_order=1)


def test_conv3x3_fused_epilogue(ops):
    """+bias +time-embedding vector (openaimodel.py:262-272) +skip residual (:274), fp32 output variant."""
    g = torch.Generator().manual_seed(3)
    B, H, W, C = 2, 8, 8, 64
    x = q(torch.randn(B, C, H, W, generator=g))
    w = q(torch.randn(C, C, 3, 3, generator=g) / 24)
    bias, emb = torch.randn(C, generator=g), torch.randn(B, C, generator=g)
    res = q(torch.randn(B, C, H, W, generator=g))
    ref = F.conv2d(x, w, bias, padding=1) + emb[:, :, None, None] + res
    y, _, _ = ops.conv3x3(ops.nchw_to_rows(x.to(DEV)), ops.pack_conv3x3(w.to(DEV)), bias.to(DEV), B, H, W,
                          addvec=emb.to(DEV), residual=ops.nchw_to_rows(res.to(DEV)), out_f32=True)
    check_close(ops.rows_to_nchw(y, B, H, W), ref, rl2=1e-3, what="conv fused epilogue")


# ------------------------------------------------------------------------------------------------- norms
def test_groupnorm_against_golden(ops):
    g = load_golden("norms")
    x = q(T(g["x"]))
    from oracle import ldm_ref as L
    for eps, wk, bk, silu in ((1e-5, "gn_w", "gn_b", False), (1e-5, "gn_w", "gn_b", True), (1e-6, "gn6_w", "gn6_b", False)):
        ref = F.group_norm(x, 32, T(g[wk]), T(g[bk]), eps)
        ref = L.silu(ref) if silu else ref
        y = ops.groupnorm(ops.nchw_to_rows(x.to(DEV)), T(g[wk]).to(DEV), T(g[bk]).to(DEV), 2, 64, eps, silu=silu)
        check_close(ops.rows_to_nchw(y, 2, 8, 8), ref, what=f"groupnorm eps={eps} silu={silu}")


@pytest.mark.parametrize("B,HW,C,C1", [(2, 4096, 320, None), (1, 1024, 960, 640), (3, 64, 2560, 1280), (1, 100, 64, 32)])
def test_groupnorm_shapes_and_concat(ops, B, HW, C, C1):
    g = torch.Generator().manual_seed(C + HW)
    x = q(torch.randn(B, HW, C, generator=g) * 2 + 0.5)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    from oracle import ldm_ref as L
    ref = L.silu(L.group_norm_nhwc_manual(x, gamma, beta, 1e-5))
    xd = x.reshape(B * HW, C).to(DEV, BF)
    if C1 is None:
        y = ops.groupnorm(xd, gamma.to(DEV), beta.to(DEV), B, HW, 1e-5, silu=True)
    else:
        y = ops.groupnorm(xd[:, :C1].contiguous(), gamma.to(DEV), beta.to(DEV), B, HW, 1e-5, silu=True, x2=xd[:, C1:].contiguous())
    check_close(y.reshape(B, HW, C), ref, what="groupnorm nhwc")


def _slab_stats(y):
    """fp32 reference of the producer statistics: per-channel (sum, sum of squares) of a bf16 [M, N] tensor over 32-row slabs."""
    M, N = y.shape
    yf = y.float().cpu().double()
    pad = (-M) % 32
    if pad:
        yf = torch.cat([yf, torch.zeros(pad, N, dtype=torch.double)])
    yf = yf.reshape(-1, 32, N)
    return torch.stack([yf.sum(1), (yf * yf).sum(1)], -1)


@pytest.mark.parametrize("kind,shape", [("conv", (12, 64, 64, 320, 320)),      # un-split 192x320 tile: statistics from the epilogue
                                        ("conv", (12, 32, 32, 640, 640)),      # 8-wave 128x128 tile (32x32 level)
                                        ("conv", (2, 32, 32, 64, 96)),         # small grid: stand-alone statistics pass
                                        ("conv_s2", (12, 64, 64, 320, 320)),   # split-K plan (down-sampling conv): stand-alone pass
                                        ("gemm", (12288, 640, 640)),           # proj_out of the 32x32 level: 128x128 dense tile
                                        ("gemm", (49152, 320, 320)),           # proj_out of the 64x64 level: row-panel kernel + pass
                                        ("gemm", (1000, 72, 64))])             # ragged: M % 32 != 0, N % 16 != 0
def test_producer_colstats(ops, kind, shape):
    """The per-channel slab statistics a producer hands to the consuming GroupNorm (`colstats=`) equal the statistics of the bf16
    tensor it stored, whichever route produced them (CS epilogue or the stand-alone pass); the output itself is unchanged by asking."""
    g = torch.Generator(device=DEV).manual_seed(sum(shape))
    if kind.startswith("conv"):
        B, H, W, Cin, Cout = shape
        stride = 2 if kind == "conv_s2" else 1
        x = torch.randn(B * H * W, Cin, generator=g, device=DEV).to(BF)
        w = ops.pack_conv3x3(torch.randn(Cout, Cin, 3, 3, generator=g, device=DEV) / (3 * Cin ** 0.5))
        bias = torch.randn(Cout, generator=g, device=DEV)
        Ho = (H - 1) // stride + 1
        res = torch.randn(B * Ho * Ho, Cout, generator=g, device=DEV).to(BF)
        emb = torch.randn(B, Cout, generator=g, device=DEV)
        y0, _, _ = ops.conv3x3(x, w, bias, B, H, W, addvec=emb, residual=res, stride=stride)
        cs = ops.colstats_buffer(B * Ho * Ho, Cout, DEV)
        cs.fill_(float("nan"))
        y1, _, _ = ops.conv3x3(x, w, bias, B, H, W, addvec=emb, residual=res, stride=stride, colstats=cs)
    else:
        M, N, K = shape
        a = torch.randn(M, K, generator=g, device=DEV).to(BF)
        w = (torch.randn(N, K, generator=g, device=DEV) / K ** 0.5).to(BF)
        bias = torch.randn(N, generator=g, device=DEV)
        res = torch.randn(M, N, generator=g, device=DEV).to(BF)
        y0 = ops.gemm(a, w, bias, residual=res)
        cs = ops.colstats_buffer(M, N, DEV)
        cs.fill_(float("nan"))
        y1 = ops.gemm(a, w, bias, residual=res, colstats=cs)
    assert torch.equal(y0, y1), "asking for statistics must not change the output"
    ref = _slab_stats(y1)
    got = cs.cpu().double()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), f"{kind} {shape}: {(got - ref).abs().max()} vs {ref.abs().max()}"


@pytest.mark.parametrize("B,HW,C,C1", [(12, 4096, 320, None), (3, 1024, 960, 640), (2, 1024, 1920, 1280), (2, 4096, 640, 320)])
def test_groupnorm_from_producer_statistics(ops, B, HW, C, C1):
    """GroupNorm(+SiLU) fed with the producers' slab statistics (one or two sources; a group may straddle the concat boundary:
    960 = 640 + 320 channels, 30 per group) against the fp32 reference and against the kernel's own statistics pass."""
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(C + HW)
    x = q(torch.randn(B, HW, C, generator=g) * 2 + 0.5)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = L.silu(L.group_norm_nhwc_manual(x, gamma, beta, 1e-5))
    xd = x.reshape(B * HW, C).to(DEV, BF)
    parts = [xd] if C1 is None else [xd[:, :C1].contiguous(), xd[:, C1:].contiguous()]
    stats = []
    for t in parts:
        cs = ops.colstats_buffer(t.shape[0], t.shape[1], DEV)
        # the stand-alone producer (what ae_gemm_bf16 / ae_conv3x3_bf16 run when their tile plan has no statistics epilogue)
        cs.copy_(_slab_stats(t).float().to(DEV))
        stats.append(cs)
    kw = dict(colstats=stats[0]) if C1 is None else dict(x2=parts[1], colstats=stats[0], colstats2=stats[1])
    y = ops.groupnorm(parts[0], gamma.to(DEV), beta.to(DEV), B, HW, 1e-5, silu=True, **kw)
    y_own = ops.groupnorm(parts[0], gamma.to(DEV), beta.to(DEV), B, HW, 1e-5, silu=True, **({} if C1 is None else dict(x2=parts[1])))
    check_close(y.reshape(B, HW, C), ref, what="groupnorm from producer statistics")
    assert rel_l2(y.float().cpu(), y_own.float().cpu()) < 2e-3      # a few bf16 outputs flip on the last statistics bit
    stat = torch.empty(B, 32, 2, dtype=torch.float32, device=DEV)
    ops.groupnorm(parts[0], gamma.to(DEV), beta.to(DEV), B, HW, 1e-5, silu=True, stat_out=stat, **kw)
    xg = x.reshape(B, HW, 32, C // 32)
    mean = xg.mean(dim=(1, 3))
    assert float((stat[:, :, 0].cpu() - mean).abs().max()) < 1e-4
    # a concat with statistics for only ONE source falls back to the kernel's own pass (same result as no statistics at all)
    if C1 is not None:
        y_half = ops.groupnorm(parts[0], gamma.to(DEV), beta.to(DEV), B, HW, 1e-5, silu=True, x2=parts[1], colstats=stats[0])
        assert torch.equal(y_half, y_own)


@pytest.mark.parametrize("M,C,eps", [(10, 64, 1e-5), (4096, 320, 1e-5), (777, 1280, 1e-6), (5, 2048, 1e-5)])
def test_layernorm(ops, M, C, eps):
    g = torch.Generator().manual_seed(M + C)
    x = q(torch.randn(M, C, generator=g) * 3 + 1)
    w, b = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    check_close(ops.layernorm(x.to(DEV, BF), w.to(DEV), b.to(DEV), eps), F.layer_norm(x, (C,), w, b, eps), what="layernorm")


# ------------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("BH,Nq,Nk,D", [(4, 64, 64, 40), (2, 256, 77, 80), (2, 196, 196, 80), (2, 144, 144, 160),
                                        (3, 100, 33, 8), (2, 130, 200, 16), (1, 4096, 4096, 40), (2, 1024, 1024, 80),
                                        (1, 37, 1, 64), (2, 64, 129, 32)])
def test_attention_core(ops, BH, Nq, Nk, D):
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(BH + Nq + Nk + D)
    qq, kk, vv = (q(torch.randn(BH, n, D, generator=g)) for n in (Nq, Nk, Nk))
    ref = L.sdpa_core(qq, kk, vv, D ** -0.5)
    out = ops.attention_bhnd(qq.to(DEV, BF), kk.to(DEV, BF), vv.to(DEV, BF))
    check_close(out, ref, rl2=6e-3, what=f"attention {BH}x{Nq}x{Nk}x{D}")


@pytest.mark.parametrize("BH,Nq,Nk,bias", [(4, 256, 256, False), (2, 200, 190, False), (2, 384, 4096, True)])
def test_attention_fp8(ops, BH, Nq, Nk, bias):
    """fp8 (e4m3) attention (ae_attn_fwd_fp8, BASELINE.json configs[4]) against fp32 softmax attention on the same bf16 inputs, with
    and without the SAM rel-pos bias (image_encoder.py:224-240, 325-361).  Tolerance = what e4m3 operands cost: a 3-bit mantissa on
    q, k, v and on the probabilities gives rel-L2 ~5e-2 at unit-variance logits (measured 5.2e-2 / 5.3e-2 / 5.3e-2); the bf16
    kernel's bound on the same inputs is 6e-3."""
    from oracle import ldm_ref as L
    D = 80
    g = torch.Generator().manual_seed(BH * 5 + Nq)
    qq, kk, vv = (q(torch.randn(BH, n, D, generator=g)) for n in (Nq, Nk, Nk))
    b = None
    rel_h = rel_w = None
    kH, kW = 0, 0
    if bias:
        kW, kH = 64, Nk // 64
        rel_h, rel_w = torch.randn(BH, Nq, kH, generator=g), torch.randn(BH, Nq, kW, generator=g)
        b = (rel_h[:, :, :, None] + rel_w[:, :, None, :]).reshape(BH, Nq, Nk)
    ref = L.sdpa_core(qq, kk, vv, D ** -0.5, bias=b)
    out = ops.attention_fp8(qq.to(DEV, BF), kk.to(DEV, BF), vv.to(DEV, BF), BH, 1, Nq, Nk, D, D ** -0.5, (Nq * D, 0, D), (Nk * D, 0, D), (Nk * D, 0, D),
                            rel_h=None if rel_h is None else rel_h.to(DEV), rel_w=None if rel_w is None else rel_w.to(DEV), kH=kH, kW=kW)
    e = rel_l2(out.reshape(BH, Nq, D).float().cpu(), ref)
    assert np.isfinite(e) and e <= 8e-2, f"fp8 attention rel-L2 {e:.3e}"
    # and it is a genuinely different operator from the bf16 one (guards against a silent bf16 fallback)
    out16 = ops.attention(qq.to(DEV, BF), kk.to(DEV, BF), vv.to(DEV, BF), BH, 1, Nq, Nk, D, D ** -0.5, (Nq * D, 0, D), (Nk * D, 0, D), (Nk * D, 0, D),
                          rel_h=None if rel_h is None else rel_h.to(DEV), rel_w=None if rel_w is None else rel_w.to(DEV), kH=kH, kW=kW)
    assert rel_l2(out16.reshape(BH, Nq, D).float().cpu(), ref) < 0.5 * e


def test_attention_rescale_branch_forced(ops):
    """A spiked key late in the sequence forces the online-softmax running max to jump (rescale of O and l)."""
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(11)
    BH, N, D = 2, 512, 40
    qq, kk, vv = (q(torch.randn(BH, N, D, generator=g)) for _ in range(3))
    kk[:, 300] = qq[:, 5] * 6.0   # row 5's logit at key 300 dwarfs everything seen in tiles 0..3
    kk[:, 450] = qq[:, 70] * 9.0
    ref = L.sdpa_core(qq, kk, vv, D ** -0.5)
    out = ops.attention_bhnd(qq.to(DEV, BF), kk.to(DEV, BF), vv.to(DEV, BF))
    check_close(out, ref, rl2=6e-3, what="attention with forced rescale")


def test_attention_rows_sum_property_full_size(ops):
    """V = const columns -> output equals that constant for every row (softmax rows sum to 1), at N = 4096."""
    g = torch.Generator(device=DEV).manual_seed(2)
    BH, N, D = 8, 4096, 40
    qq = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
    kk = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
    vv = torch.arange(D, device=DEV, dtype=torch.float32).reshape(1, 1, D).expand(BH, N, D).to(BF).contiguous()
    out = ops.attention_bhnd(qq, kk, vv).float()
    assert float((out - vv.float()).abs().max()) <= 0.26  # bf16 rounding of P (<=2^-9 rel) on values up to 39


def test_attention_key_permutation_invariance_full_size(ops):
    """Size-independent property at the BASELINE batch (4 images x 3 CFG branches x 8 heads, N = 4096, d = 40): attention does not
    depend on the ORDER of the keys — permuting K and V rows together only changes which 64-key tile each key lands in."""
    g = torch.Generator(device=DEV).manual_seed(21)
    BH, N, D = 96, 4096, 40
    qq = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
    kk = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
    vv = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
    perm = torch.randperm(N, generator=g, device=DEV)
    a = ops.attention_bhnd(qq, kk, vv).float()
    b = ops.attention_bhnd(qq, kk[:, perm].contiguous(), vv[:, perm].contiguous()).float()
    assert rel_l2(b.cpu(), a.cpu()) < 4e-3          # bf16 rounding of P and the tile-wise summation order
    shift = ops.attention_bhnd(qq, kk, (vv.float() + 3.0).to(BF)).float()      # rows of P sum to 1: V + c -> out + c
    assert float((shift - a - 3.0).abs().max()) < 6e-2


def test_attention_run_to_run_determinism_full_size(ops):
    """Bit-identical results over repeated launches at the BASELINE shape.  Guards a hazard met while building the two-query-group
    kernel (attention_fast.hip: a VALU write to the SrcB registers of a just-issued v_mfma_f32_32x32x16_bf16 changed query columns
    16-31 of the second group in ~1 of 6 launches, profiles/r03_attn_qg2_hazard.txt) — a parity test that compares one launch with
    the oracle passes most of the time with such a bug."""
    g = torch.Generator(device=DEV).manual_seed(23)
    BH, N, D = 96, 4096, 40
    qq = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
    kk = torch.randn(BH, N, D, generator=g, device=DEV).to(BF)
    vv = (torch.randn(BH, N, D, generator=g, device=DEV) + 3.0).to(BF)
    first = ops.attention_bhnd(qq, kk, vv)
    for i in range(40):
        again = ops.attention_bhnd(qq, kk, vv)
        assert torch.equal(again, first), f"launch {i + 1} differs from launch 0 in {int((again != first).sum())} elements"
    for (BHs, Ns, Ds) in ((96, 1024, 80), (96, 256, 160)):
        q2, k2, v2 = (torch.randn(BHs, Ns, Ds, generator=g, device=DEV).to(BF) for _ in range(3))
        f2 = ops.attention_bhnd(q2, k2, v2)
        for _ in range(10):
            assert torch.equal(ops.attention_bhnd(q2, k2, v2), f2)


@pytest.mark.parametrize("B,H,Nq,Nk,T,D", [(12, 8, 4096, 78, 4, 40), (16, 8, 1100, 78, 4, 40), (12, 8, 1024, 78, 0, 80), (64, 8, 300, 128, 64, 80)])
def test_attention_short_kv_query_groups_vs_oracle(ops, B, H, Nq, Nk, T, D):
    """Cross-attention shapes (text tokens + task token, optionally the expert tokens as a second softmax segment) on the short-K/V
    variant of attention_fast.hip: all keys resident in LDS, several 128-query groups per block once the grid is large enough
    (>= 512 blocks: 4 groups at the UNet's 64x64 level, 2 at Nq = 1100 with a ragged last block) — a size no small test reaches.
    Sampled heads against the fp32 oracle: out = Attn(q, K, V) + g_b * Attn(q, K_ip, V_ip)."""
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(B * 7 + Nq + D)
    inner = H * D
    qq = q(torch.randn(B * Nq, inner, generator=g))
    kv = q(torch.randn(B * Nk, 2 * inner, generator=g))
    kvip = q(torch.randn(B * max(T, 1), 2 * inner, generator=g))
    gate = torch.rand(B, generator=g) + 0.25
    qs, ks, ks2 = (Nq * inner, D, inner), (Nk * 2 * inner, D, 2 * inner), (T * 2 * inner, D, 2 * inner)
    qd, kvd, kvipd = qq.to(DEV, BF), kv.to(DEV, BF), kvip.to(DEV, BF)
    seg2 = (kvipd, kvipd[:, inner:], T, ks2, ks2, gate.to(DEV)) if T else None
    out = ops.attention(qd, kvd, kvd[:, inner:], B, H, Nq, Nk, D, D ** -0.5, qs, ks, ks, seg2=seg2).float().cpu().view(B, Nq, H, D)
    for (b, h) in ((0, 0), (B - 1, H - 1), (B // 2, 3), (1, H - 2)):
        qh = qq.view(B, Nq, H, D)[b, :, h][None]
        kh, vh = kv.view(B, Nk, 2, H, D)[b, :, 0, h][None], kv.view(B, Nk, 2, H, D)[b, :, 1, h][None]
        ref = L.sdpa_core(qh, kh, vh, D ** -0.5)
        if T:
            k2, v2 = kvip.view(B, T, 2, H, D)[b, :, 0, h][None], kvip.view(B, T, 2, H, D)[b, :, 1, h][None]
            ref = ref + gate[b] * L.sdpa_core(qh, k2, v2, D ** -0.5)
        check_close(out[b, :, h][None], ref, rl2=6e-3, what=f"short-K/V attention b={b} h={h}")


def test_attention_two_query_groups_vs_oracle(ops):
    """The 64-queries-per-wave variant of the long-sequence kernel (attention_fast.hip, QG = 2) only runs on grids of >= 1024 blocks
    of 256 queries — no small test reaches it.  B*H = 64 heads of N = 4096, d = 40 (the UNet's 64x64-level shape at a smaller batch):
    six of the heads, and the last query block, against the fp32 oracle."""
    from oracle import ldm_ref as L
    g = torch.Generator().manual_seed(640)
    BH, N, D = 64, 4096, 40
    qq, kk, vv = (torch.randn(BH, N, D, generator=g).to(BF) for _ in range(3))
    kk[:, 100] *= 6.0          # one key far above the rest: the lazy offset moves in SOME query groups only
    out = ops.attention_bhnd(qq.to(DEV), kk.to(DEV), vv.to(DEV)).float().cpu()
    for h in (0, 1, 17, 31, 32, 63):
        ref = L.sdpa_core(qq[h:h + 1].float(), kk[h:h + 1].float(), vv[h:h + 1].float(), D ** -0.5)
        check_close(out[h:h + 1], ref, rl2=6e-3, mabs=3e-2, what=f"attention QG=2 head {h}")
    # ragged query count: the second query group of the last wave is partly / wholly out of range
    Nq = 4096 - 40
    out2 = ops.attention_bhnd(qq[:, :Nq].contiguous().to(DEV), kk.to(DEV), vv.to(DEV)).float().cpu()
    assert torch.equal(out2[:, :3840], out[:, :3840]), "rows must not depend on how many other queries the launch holds"
    # (the last block's waves hold clamped duplicate rows instead of rows 4056..4095: they may move the shared lazy offset at other tiles)
    check_close(out2[:, 3840:], out[:, 3840:Nq], rl2=4e-3, mabs=2e-2, what="attention QG=2 ragged tail block")


def test_attention_pipelined_equals_two_group_kernel():
    """Round 5: the software-pipelined long-sequence kernel (attn_pipe_kernel, AE_ATTN_V flag 4: the default) issues the same MFMAs on the same
    operands in the same per-accumulator order as the two-query-group kernel (AE_ATTN_V = 3) and takes the same rebase decisions, so the two must
    agree BIT FOR BIT: the UNet shape with late logit spikes, a ragged query count, the fused-qkv strided layout, two / three / sixteen key tiles at
    the usual and at a hot logit scale (frequent rebases), and the log-sum-exp / output-scale / accumulate options.  The launcher reads its variant
    knob once per process, so tools/attn_pipe_check.py runs one child per variant and compares what they wrote; it also checks both against an fp32
    torch statement and for run-to-run equality."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "attn_pipe_check.py"), "3", "7k0", "7", "7p"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-3000:]
    # four variants: the two-group kernel, the pipelined kernel on 64-key tiles (round 5) and on 128-key tiles (round 6, the default), and — opt-in, AE_ATTN_PV16=1,
    # measured slower in the graph — the 128-key form with the PV products on 48 rows of v_mfma_f32_16x16x32_bf16: another summation order inside PV,
    # so it is held to 2.5e-3 of the two-group kernel's outputs (3.9e-5 at the time of writing) instead of to identity
    assert r.stdout.count("bit-identical on all") == 2 and "FAIL" not in r.stdout and "bit-identical: False" not in r.stdout
    assert r.stdout.count("tolerance, not identity") == 1


def test_conv3x3_linearity_and_groupnorm_scale_invariance_full_size(ops):
    """BASELINE-size layer ([12, 320, 64, 64] -> 320): conv(x1 + x2) = conv(x1) + conv(x2) - bias; GroupNorm(c x) = GroupNorm(x)."""
    g = torch.Generator(device=DEV).manual_seed(22)
    B, H, W, C = 12, 64, 64, 320
    x1 = torch.randn(B * H * W, C, generator=g, device=DEV).to(BF)
    x2 = torch.randn(B * H * W, C, generator=g, device=DEV).to(BF)
    w = ops.pack_conv3x3((torch.randn(C, C, 3, 3, generator=g, device=DEV) / (3 * C ** 0.5)))
    bias = torch.randn(C, generator=g, device=DEV)
    xs = (x1.float() + x2.float()).to(BF)
    ys, _, _ = ops.conv3x3(xs, w, bias, B, H, W, out_f32=True)
    y1, _, _ = ops.conv3x3(x1, w, bias, B, H, W, out_f32=True)
    y2, _, _ = ops.conv3x3(x2, w, bias, B, H, W, out_f32=True)
    assert rel_l2(ys.cpu(), (y1 + y2 - bias[None, :]).cpu()) < 6e-3    # bf16 rounding of the summed input only
    gamma, beta = torch.randn(C, generator=g, device=DEV), torch.randn(C, generator=g, device=DEV)
    n1 = ops.groupnorm(x1, gamma, beta, B, H * W, 1e-5, silu=True).float()
    n4 = ops.groupnorm((x1.float() * 4.0).to(BF), gamma, beta, B, H * W, 1e-5, silu=True).float()   # x4 is exact in bf16
    assert rel_l2(n4.cpu(), n1.cpu()) < 1e-3          # eps against var 1 vs 16, and a few bf16 output roundings that flip


def test_ddim_loop_telescopes_at_full_size(ops):
    """50 DDIM steps over the BASELINE latent batch with eps == 0: x_prev = sqrt(a_prev / a_t) x every step, so the loop must end at
    x_T * prod_i sqrt(a_prev_i / a_t_i) — a closed form that checks the schedule tables, the index bookkeeping and the fused step."""
    from anyedit_amd.ldm.models.diffusion.ddim import DDIMSampler
    from oracle import schedule_ref as S

    class ZeroEps:
        parameterization = "eps"

        def __init__(self):
            self.num_timesteps = 1000
            for k, v in S.register_schedule("linear", 1000, 0.00085, 0.0120).items():
                if isinstance(v, torch.Tensor):
                    setattr(self, k, v.to(DEV))
            self.device = torch.device(DEV)

        def apply_model(self, x, t, c):
            self.seen.append(int(t[0]))
            return torch.zeros_like(x)

    m = ZeroEps()
    m.seen = []
    sampler = DDIMSampler(m)
    x_T = torch.randn(12, 4, 64, 64, generator=torch.Generator().manual_seed(23)).to(DEV)
    out, _ = sampler.sample(50, 12, (4, 64, 64), torch.zeros(12, 1, device=DEV), eta=0.0, x_T=x_T, verbose=False)
    ts = S.make_ddim_timesteps("uniform", 50, 1000)
    assert m.seen == [int(v) for v in ts[::-1]]                                     # 981, 961, ..., 1 (bit-exact bookkeeping)
    a = np.asarray(sampler.ddim_alphas, dtype=np.float64)
    ap = np.asarray(sampler.ddim_alphas_prev, dtype=np.float64)
    factor = float(np.prod(np.sqrt(ap / a)))
    assert rel_l2(out.cpu(), (x_T * factor).cpu()) < 2e-6


def test_attention_golden_modules(ops):
    """CrossAttention module (fused qkv, strided heads, to_out) against the reference-derived goldens."""
    from anyedit_amd.ldm.modules.attention import CrossAttention
    g = load_golden("attention")
    for name in ("self_n64_d40", "cross_n256_d80_k77", "self_n196_d80", "self_n144_d160", "masked"):
        B, N, qd, h, dh, cd, Nk = (int(v) for v in g[f"{name}.cfg"])
        m = CrossAttention(qd, context_dim=cd or None, heads=h, dim_head=dh)
        m.load_state_dict(sub_sd(g, f"{name}.w."))
        m = m.to(DEV)
        ctx = T(g[f"{name}.ctx"]).to(DEV) if f"{name}.ctx" in g else None
        mask = T(g[f"{name}.mask"]).to(DEV) if f"{name}.mask" in g else None
        y = m(T(g[f"{name}.x"]).to(DEV), context=ctx, mask=mask)
        check_close(y, T(g[f"{name}.y"]), rl2=1.5e-2, mabs=4e-2, what=name)  # fp32 golden vs bf16 weights+activations


# ------------------------------------------------------------------------------------------------- layout / elementwise
def test_layout_round_trips_bit_exact(ops):
    g = torch.Generator().manual_seed(4)
    x = q(torch.randn(3, 20, 7, 9, generator=g))
    rows = ops.nchw_to_rows(x.to(DEV), 24)
    assert rows.shape == (3 * 63, 24) and float(rows[:, 20:].abs().sum()) == 0
    assert torch.equal(rows[:, :20].float().cpu().reshape(3, 7, 9, 20), x.permute(0, 2, 3, 1))
    back = ops.rows_to_nchw(rows[:, :20].contiguous(), 3, 7, 9)
    assert torch.equal(back.cpu(), x)
    a, b = rows[:, :8].contiguous(), rows[:, 8:24].contiguous()
    assert torch.equal(ops.concat_channels(a, b), rows)


def test_timestep_embedding(ops):
    from oracle.schedule_ref import timestep_embedding
    t = torch.tensor([1, 981, 500, 0, 21], dtype=torch.long)
    ref = timestep_embedding(t, 320)
    got = ops.timestep_embedding(t.to(DEV), 320, out_f32=True).cpu()
    assert float((got - ref).abs().max()) < 2e-4  # fp32 sin/cos of arguments up to ~1e3
    assert torch.equal(ops.timestep_embedding(t.to(DEV), 320).float().cpu(), got.to(BF).float())


def test_ddim_step_bit_exact_vs_unfused_fp32(ops):
    """The fused kernel reproduces the reference's fp32 expression sequence bit for bit (ddim.py:211-250)."""
    g = torch.Generator().manual_seed(6)
    B = 3
    x = torch.randn(B, 4, 16, 16, generator=g)
    noise = torch.randn(B, 4, 16, 16, generator=g)
    a_t, a_prev, sigma = np.float32(0.1163), np.float32(0.1581), np.float32(0.07)
    s1m = np.sqrt(np.float32(1) - a_t, dtype=np.float32)
    full = lambda v: torch.full((B, 1, 1, 1), float(v))
    for branches in (1, 2, 3):
        eps = torch.randn(branches * B, 4, 16, 16, generator=g)
        if branches == 1:
            e = eps
        elif branches == 2:
            eu, ec = eps.chunk(2)
            e = eu + 7.5 * (ec - eu)
        else:
            et, ei, eu = eps.chunk(3)
            e = eu + 7.5 * (et - ei) + 1.5 * (ei - eu)
        sq_at = full(np.sqrt(a_t, dtype=np.float32))
        sq_ap = full(np.sqrt(a_prev, dtype=np.float32))
        dirc = full(np.sqrt(np.float32(1) - a_prev - sigma * sigma, dtype=np.float32))
        pred_x0 = (x - full(s1m) * e) / sq_at
        x_prev = sq_ap * pred_x0 + dirc * e + full(sigma) * noise * 1.0
        coeffs = (float(s1m), float(sq_at[0]), float(sq_ap[0]), float(dirc[0]), float(sigma))
        xp, p0 = ops.ddim_step(x.to(DEV), eps.to(DEV), coeffs, branches, s0=7.5, s1=1.5, noise=noise.to(DEV))
        assert torch.equal(p0.cpu(), pred_x0), f"pred_x0 not bit-exact (branches={branches})"
        assert torch.equal(xp.cpu(), x_prev), f"x_prev not bit-exact (branches={branches})"


def test_q_sample_and_mask_blend_bit_exact(ops):
    g = torch.Generator().manual_seed(8)
    x0, noise, img = (torch.randn(2, 4, 8, 8, generator=g) for _ in range(3))
    mask = (torch.rand(2, 1, 8, 8, generator=g) > 0.5).float()
    sa, s1 = np.float32(0.73), np.float32(0.68)
    qs = float(sa) * x0 + float(s1) * noise
    ref = qs * mask + (1. - mask) * img
    got = ops.mask_blend(img.to(DEV), x0.to(DEV), noise.to(DEV), mask.to(DEV), float(sa), float(s1))
    assert torch.equal(got.cpu(), ref)
    ref2 = img * mask + qs * (1. - mask)
    assert torch.equal(ops.mask_blend(img.to(DEV), x0.to(DEV), noise.to(DEV), mask.to(DEV), float(sa), float(s1), ip2p_order=True).cpu(), ref2)
    # masks the reference's broadcasting accepts as well: per-channel [B, C, H, W] (latent masks) and a shared [1, 1, H, W] plane
    mc = (torch.rand(2, 4, 8, 8, generator=g) > 0.5).float()
    assert torch.equal(ops.mask_blend(img.to(DEV), x0.to(DEV), noise.to(DEV), mc.to(DEV), float(sa), float(s1)).cpu(), qs * mc + (1. - mc) * img)
    m1 = mask[:1]
    assert torch.equal(ops.mask_blend(img.to(DEV), x0.to(DEV), noise.to(DEV), m1.to(DEV), float(sa), float(s1)).cpu(), qs * m1 + (1. - m1) * img)
    sav, s1v = torch.tensor([0.73, 0.2]), torch.tensor([0.68, 0.97])
    ref3 = sav[:, None, None, None] * x0 + s1v[:, None, None, None] * noise
    assert torch.equal(ops.q_sample(x0.to(DEV), noise.to(DEV), sav.to(DEV), s1v.to(DEV)).cpu(), ref3)


def test_mse_and_task_gate(ops):
    g = torch.Generator().manual_seed(9)
    a, b = torch.randn(4, 4, 32, 32, generator=g), torch.randn(4, 4, 32, 32, generator=g)
    assert abs(float(ops.mse(a.to(DEV), b.to(DEV))) - float(F.mse_loss(a, b))) < 1e-5
    te, Wg, bg = torch.randn(6, 48, generator=g), torch.randn(11, 48, generator=g), torch.randn(11, generator=g)
    code = torch.tensor([0, 5, 3, 3, 1])
    probs, top1, top1p = ops.task_gate(te.to(DEV), code.to(DEV), Wg.to(DEV), bg.to(DEV))
    ref = torch.softmax(te[code] @ Wg.t() + bg, dim=-1)
    assert torch.allclose(probs.cpu(), ref, atol=1e-5)
    assert torch.equal(top1.cpu().long(), ref.argmax(-1)) and torch.allclose(top1p.cpu(), ref.max(-1).values, atol=1e-5)


# ------------------------------------------------------------------------------------------------- GroundingDINO MSDeformAttn (N2)
def test_ms_deform_attn_forward(ops):
    """ae_ms_deform_attn_fwd_f32 (the reference's `_C.ms_deform_attn_forward`) against the reference's own PyTorch statement
    (golden) and, on a larger seeded case with out-of-range samples, against the oracle.  fp32 gathers: tolerance 1e-5."""
    from oracle import msda_ref as MS
    g = load_golden("msda")
    for tag in ("a", "b"):
        out = ops.ms_deform_attn(T(g[f"{tag}.value"]).to(DEV), T(g[f"{tag}.shapes"]).to(DEV), T(g[f"{tag}.start"]).to(DEV),
                                 T(g[f"{tag}.loc"]).to(DEV), T(g[f"{tag}.w"]).to(DEV), 64)
        check_close(out, T(g[f"{tag}.out"]), rl2=1e-5, mabs=1e-5, what=f"ms_deform_attn golden {tag}")
    gen = torch.Generator().manual_seed(5)
    shapes = torch.tensor([(100, 134), (50, 67), (25, 34), (13, 17)], dtype=torch.long)      # an 800x1066 image's 4 feature levels
    S = int(shapes.prod(1).sum())
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    value = torch.randn(2, S, 8, 32, generator=gen)
    loc = torch.rand(2, 900, 8, 4, 4, 2, generator=gen) * 1.2 - 0.1
    w = torch.softmax(torch.randn(2, 900, 8, 16, generator=gen), -1).view(2, 900, 8, 4, 4)
    ref = MS.ms_deform_attn(value, shapes, start, loc, w)
    out = ops.ms_deform_attn(value.to(DEV), shapes.to(DEV), start.to(DEV), loc.to(DEV), w.to(DEV))
    check_close(out, ref, rl2=1e-5, mabs=1e-5, what="ms_deform_attn 900 queries x 4 levels")


def test_ms_deform_attn_backward(ops):
    """ae_ms_deform_attn_bwd_f32 (the reference's `_C.ms_deform_attn_backward`) against autograd through the reference's PyTorch statement
    (golden; case c has d/4 = 3 = the float-atomics reduction instead of the shuffle tree) and, at an 800x1066 image's four levels with 900
    queries, against the oracle's explicit float64 restatement.  grad_value is a sum of float atomics: fp32 round-off, any order."""
    from oracle import msda_ref as MS
    from anyedit_amd.groundingdino.ms_deform_attn import MultiScaleDeformableAttnFunction, multi_scale_deformable_attn
    g = load_golden("msda_bwd")
    for tag in ("a", "b", "c"):
        a = [T(g[f"{tag}.{k}"]).to(DEV) for k in ("value", "shapes", "start", "loc", "w", "go")]
        gv, gl, gw = ops.ms_deform_attn_bwd(*a, 64)
        check_close(gv, T(g[f"{tag}.gv"]), rl2=1e-5, mabs=2e-5, what=f"msda bwd golden {tag}: grad_value")
        check_close(gl, T(g[f"{tag}.gl"]), rl2=2e-5, mabs=2e-4, what=f"msda bwd golden {tag}: grad_sampling_loc")
        check_close(gw, T(g[f"{tag}.gw"]), rl2=1e-5, mabs=2e-5, what=f"msda bwd golden {tag}: grad_attn_weight")
    # the autograd Function of ms_deform_attn.py:42-90: forward + backward through torch's engine
    a = [T(g[f"a.{k}"]).to(DEV) for k in ("value", "shapes", "start", "loc", "w", "go")]
    v, loc, w = a[0].clone().requires_grad_(True), a[3].clone().requires_grad_(True), a[4].clone().requires_grad_(True)
    out = MultiScaleDeformableAttnFunction.apply(v, a[1], a[2], loc, w, 64)
    check_close(out.detach(), T(g["a.out"]), rl2=1e-5, mabs=1e-5, what="msda Function forward")
    out.backward(a[5])
    check_close(v.grad, T(g["a.gv"]), rl2=1e-5, mabs=2e-5, what="msda Function: value.grad")
    check_close(loc.grad, T(g["a.gl"]), rl2=2e-5, mabs=2e-4, what="msda Function: sampling_locations.grad")
    check_close(w.grad, T(g["a.gw"]), rl2=1e-5, mabs=2e-5, what="msda Function: attention_weights.grad")
    v2 = a[0].clone().requires_grad_(True)                                   # the functional entry routes through it when a gradient is wanted
    multi_scale_deformable_attn(v2, a[1], a[2], a[3], a[4]).backward(a[5])
    check_close(v2.grad, T(g["a.gv"]), rl2=1e-5, mabs=2e-5, what="msda functional entry: value.grad")
    gen = torch.Generator().manual_seed(6)
    shapes = torch.tensor([(100, 134), (50, 67), (25, 34), (13, 17)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    value = torch.randn(2, S, 8, 32, generator=gen)
    loc = torch.rand(2, 900, 8, 4, 4, 2, generator=gen) * 1.2 - 0.1
    w = torch.softmax(torch.randn(2, 900, 8, 16, generator=gen), -1).view(2, 900, 8, 4, 4)
    go = torch.randn(2, 900, 256, generator=gen)
    rv, rl, rw = MS.ms_deform_attn_backward(value, shapes, start, loc, w, go)
    gv, gl, gw = ops.ms_deform_attn_bwd(value.to(DEV), shapes.to(DEV), start.to(DEV), loc.to(DEV), w.to(DEV), go.to(DEV))
    check_close(gv, rv, rl2=1e-5, mabs=5e-5, what="msda bwd 900 queries: grad_value")
    # the location gradient is DISCONTINUOUS where a sample crosses a pixel centre (the bilinear cell changes), and whether x*W - 0.5 lands
    # on one side or the other of an integer within float rounding differs between the float64 oracle, the fp32 kernel (one fma) and the
    # reference's grid_sample arithmetic: samples within 1e-4 pixel of a cell boundary (about 4e-4 of them) are left out of THIS comparison
    # only — grad_value and grad_attn_weight are continuous there and are compared everywhere
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).double().view(1, 1, 1, 4, 1, 2)
    px = loc.double() * wh - 0.5
    keep = ~((px - px.round()).abs() < 1e-4).any(-1, keepdim=True)
    assert float((~keep).float().mean()) < 2e-3
    check_close(gl.cpu() * keep, rl * keep, rl2=2e-5, mabs=2e-4, what="msda bwd 900 queries: grad_sampling_loc")
    check_close(gw, rw, rl2=1e-5, mabs=5e-5, what="msda bwd 900 queries: grad_attn_weight")


@pytest.mark.parametrize("M,N,K", [(13294, 256, 256), (900, 128, 256), (100, 48, 64), (1, 4, 16), (65, 96, 64)])
def test_linear_f32_exact(ops, M, N, K):
    """ae_linear_f32 (f32-input MFMA, the MSDeformAttn projections) against fp64: fp32 round-off only — these layers feed sampling
    COORDINATES, so they must not go through the bf16 GEMM."""
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    ref = (x.double() @ w.double().t() + b.double()).float()
    out = ops.linear_f32(x.to(DEV), w.to(DEV), b.to(DEV))
    check_close(out, ref, rl2=2e-6, mabs=1e-5, what=f"linear_f32 {M}x{N}x{K}")
    out3 = ops.linear_f32(x.reshape(1, M, K).to(DEV), w.to(DEV))          # leading dims, no bias
    check_close(out3.reshape(M, N), (x.double() @ w.double().t()).float(), rl2=2e-6, mabs=1e-5, what="linear_f32 no bias")


def test_ms_deform_attn_module_backward_vs_oracle_autograd():
    """gradients must flow THROUGH the module — its four projections run `ae_linear_f32` behind an autograd Function (dX, dW, db on
    the same kernel) and the sampling core behind MultiScaleDeformableAttnFunction.  Every parameter's gradient and the gradients of query and
    value against autograd through a CPU statement of the same module (F.linear + the oracle's differentiable sampling core, same weights)."""
    from oracle import msda_ref as MS
    from anyedit_amd.groundingdino.ms_deform_attn import MultiScaleDeformableAttention
    gen = torch.Generator().manual_seed(9)
    E, H, L, P, bs, nq = 64, 4, 3, 4, 2, 50
    shapes = torch.tensor([(12, 16), (6, 8), (3, 4)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    start = torch.cat([shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]])
    m = MultiScaleDeformableAttention(embed_dim=E, num_heads=H, num_levels=L, num_points=P, batch_first=True)
    with torch.no_grad():   # the reference's initial state has zero offset / weight matrices: nothing would flow through them
        for prm in m.parameters():
            prm.copy_(torch.randn(prm.shape, generator=gen) * (0.05 if prm.dim() == 2 else 0.3))
    query, value = torch.randn(bs, nq, E, generator=gen), torch.randn(bs, S, E, generator=gen)
    ref2 = torch.rand(bs, nq, L, 2, generator=gen) * 0.8 + 0.1
    go = torch.randn(bs, nq, E, generator=gen)
    # CPU statement (autograd)
    pc = {k: v.detach().clone().requires_grad_(True) for k, v in m.named_parameters()}
    qc, vc = query.clone().requires_grad_(True), value.clone().requires_grad_(True)
    v = F.linear(vc, pc["value_proj.weight"], pc["value_proj.bias"]).view(bs, S, H, E // H)
    off = F.linear(qc, pc["sampling_offsets.weight"], pc["sampling_offsets.bias"]).view(bs, nq, H, L, P, 2)
    w = F.linear(qc, pc["attention_weights.weight"], pc["attention_weights.bias"]).view(bs, nq, H, L * P).softmax(-1).view(bs, nq, H, L, P)
    wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).float()
    loc = ref2[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    outc = F.linear(MS.ms_deform_attn(v, shapes, start, loc, w), pc["output_proj.weight"], pc["output_proj.bias"])
    outc.backward(go)
    # HIP module
    m = m.to(DEV)
    qd, vd = query.to(DEV).requires_grad_(True), value.to(DEV).requires_grad_(True)
    out = m(qd, value=vd, reference_points=ref2.to(DEV), spatial_shapes=shapes.to(DEV), level_start_index=start.to(DEV))
    assert out.grad_fn is not None, "the module output must carry a grad_fn when its inputs / parameters require grad"
    check_close(out.detach(), outc.detach(), rl2=2e-5, mabs=5e-5, what="MSDeformAttn module forward (random weights)")
    out.backward(go.to(DEV))
    check_close(qd.grad, qc.grad, rl2=2e-4, mabs=2e-4, what="MSDeformAttn module: query.grad")
    check_close(vd.grad, vc.grad, rl2=2e-4, mabs=2e-4, what="MSDeformAttn module: value.grad")
    for k, prm in m.named_parameters():
        assert prm.grad is not None, k
        check_close(prm.grad, pc[k].grad, rl2=2e-4, mabs=2e-4, what=f"MSDeformAttn module: {k}.grad")


def test_ms_deform_attn_module_golden():
    """MultiScaleDeformableAttention mirror (state-dict compatible) vs the reference module's CPU output: 2-d and 4-d reference
    points, query_pos, key_padding_mask."""
    from anyedit_amd.groundingdino.ms_deform_attn import MultiScaleDeformableAttention
    g = load_golden("msda")
    m = MultiScaleDeformableAttention(embed_dim=64, num_heads=4, num_levels=3, num_points=4, batch_first=True)
    m.load_state_dict(sub_sd(g, "mod.w."))
    m = m.to(DEV).eval()
    d = lambda k: T(g["mod." + k]).to(DEV)
    with torch.no_grad():
        out2 = m(d("query"), value=d("value"), query_pos=d("qpos"), key_padding_mask=d("mask"), reference_points=d("ref2"),
                 spatial_shapes=d("shapes"), level_start_index=d("start"))
        out4 = m(d("query"), value=d("value"), reference_points=d("ref4"), spatial_shapes=d("shapes"), level_start_index=d("start"))
    check_close(out2, T(g["mod.out2"]), rl2=2e-5, mabs=2e-5, what="MSDeformAttn module (2-d reference points)")
    check_close(out4, T(g["mod.out4"]), rl2=2e-5, mabs=2e-5, what="MSDeformAttn module (reference boxes)")
