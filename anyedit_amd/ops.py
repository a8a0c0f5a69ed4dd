"""Thin Python operators over the C ABI (include/anyedit_hip.h).  Tensor plumbing only: shape/dtype/contiguity
checks, output allocation, stream lookup.  All arithmetic happens in the HIP kernels; there is no fallback path.

Activations are channels-last bf16: [rows, C] with rows = B*H*W.
"""
import torch

import ctypes
import os
import threading

from ._lib import lib, check as _check

BF16 = torch.bfloat16
EPI_NONE, EPI_GELU, EPI_GEGLU, EPI_SILU, EPI_RELU = 0, 1, 2, 3, 4


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


_live = threading.local()


def _tmp(t):
    """Keeps a converted temporary (`x.float().contiguous()` ...) alive until the launch it feeds has been enqueued.

    Without this the temporary is freed as soon as `_p` has taken its address, and the NEXT temporary of the same argument
    list gets the same block back from the caching allocator: two arguments of one kernel would alias.  `check`
    releases the references once the launch call has returned — the allocator's stream ordering covers the rest."""
    if t is not None:
        lst = getattr(_live, "t", None)
        if lst is None:
            lst = _live.t = []
        lst.append(t)
    return t


def check(rc, what=""):
    lst = getattr(_live, "t", None)
    if lst:
        lst.clear()
    _check(rc, what)


def weights_token(*tensors):
    """Identity + in-place version of every tensor a packed-weight cache was built from: (storage address, autograd version counter).
    `load_state_dict` (on the module or on ANY parent), torch optimizer steps and `.to(device)` all change it, so a cache keyed on the
    token can never serve stale packed weights (only the outermost load_state_dict used to invalidate).  Kernels that
    update parameters through raw pointers do NOT move the counter by themselves: `AnySDTrainer.optimizer_step` bumps it after
    `adamw_step` (torch.autograd.graph.increment_version), and any other raw-pointer writer must do the same."""
    return tuple((t.data_ptr(), _version_of(t)) for t in tensors if t is not None)


def _version_of(t):
    try:
        return t._version
    except RuntimeError:      # inference tensors carry no version counter: identity only
        return -1


def cache_stale(mod, attr, *tensors):
    """True when `mod.<attr>` is missing or was built from other values of `tensors`; records the new token (the caller rebuilds now)."""
    tok = weights_token(*tensors)
    if getattr(mod, attr, None) is None or mod.__dict__.get(attr + "_tok") != tok:
        mod.__dict__[attr + "_tok"] = tok
        return True
    return False


def _chk(t, dtype, name, dims=None):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (anyedit_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if dims is not None and t.dim() != dims:
        raise ValueError(f"{name}: expected {dims} dims, got shape {tuple(t.shape)}")


# --------------------------------------------------------------------------- weight packing (host side, once)
def pack_linear(w):
    """nn.Linear / 1x1 conv weight -> bf16 [N, K] contiguous."""
    return w.detach().reshape(w.shape[0], -1).to(BF16).contiguous()


def pack_conv3x3(w, cin_pad=None, k_order=0):
    """[Cout, Cin, 3, 3] -> bf16 [Cout, 9*Cin_pad], K ordered (ky, kx, cin) to match the implicit-GEMM gather; k_order = 1: K ordered
    (cin // 64, ky, kx, cin % 64) — the chunk-major order of `conv3x3(..., k_order=1)` (same values, another column order)."""
    cout, cin = w.shape[0], w.shape[1]
    cin_pad = cin_pad or ((cin + 63) // 64 * 64)
    wp = torch.zeros(cout, 3, 3, cin_pad, dtype=BF16, device=w.device)
    wp[..., :cin] = w.detach().permute(0, 2, 3, 1).to(BF16)
    if k_order == 1:
        wp = wp.reshape(cout, 9, cin_pad // 64, 64).permute(0, 2, 1, 3)
    return wp.reshape(cout, 9 * cin_pad).contiguous()


def pack_conv3x3_taps8(w):
    """[Cout, 8, 3, 3] -> bf16 [Cout, 128], column 8 * (3 ky + kx) + cin, zeros in columns 72..127: the stem conv as a dense GEMM over
    `im2col3x3_c8` rows (same products as the implicit GEMM, a ninth of its K extent)."""
    cout, cin = w.shape[0], w.shape[1]
    assert cin == 8 and tuple(w.shape[2:]) == (3, 3)
    wp = torch.zeros(cout, 128, dtype=BF16, device=w.device)
    wp[:, :72] = w.detach().permute(0, 2, 3, 1).reshape(cout, 72).to(BF16)
    return wp.contiguous()


def pack_conv3x3_up2(w):
    """[Cout, Cin, 3, 3] (fp32 master) -> bf16 [4, Cout, 4*Cin]: the weights of `conv3x3_up2` — nearest-x2 upsampling followed by the 3x3 conv equals, per
    output parity (py, px), a 2x2 conv on the low-resolution input whose taps are SUMS of the 3x3 taps that land on the same input pixel
    (row pairs: py = 0 -> {w[0]}, {w[1] + w[2]}; py = 1 -> {w[0] + w[1]}, {w[2]}; columns likewise).  Summed in fp32, rounded to bf16 once."""
    cout, cin = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3) and cin % 64 == 0
    return up2_weight_sums(w).reshape(4, cout, 4 * cin).to(BF16).contiguous()


def up2_weight_sums(w):
    """fp32 [4, Cout, 4, Cin]: set 2 py + px, tap 2 i + j of `pack_conv3x3_up2` before the rounding (tests/test_host_logic.py restates the identity on CPU)."""
    cout, cin = w.shape[0], w.shape[1]
    w32 = w.detach().float()
    sets = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}
    out = torch.empty(4, cout, 4, cin, dtype=torch.float32, device=w.device)
    for py in range(2):
        for px in range(2):
            for i in range(2):
                for j in range(2):
                    acc = torch.zeros(cout, cin, dtype=torch.float32, device=w.device)
                    for ky in sets[(py, i)]:
                        for kx in sets[(px, j)]:
                            acc = acc + w32[:, :, ky, kx]
                    out[2 * py + px, :, 2 * i + j] = acc
    return out


_UP2_SUBPIXEL = os.environ.get("AE_UP2_SUBPIXEL", "1") != "0"  # tuning knob (A/B): 0 = the up-convs through the upsampling gather of ae_conv3x3_bf16 (9 taps per output pixel)


def up2_subpixel_ok(Cin, Cout, H, W, want_stats):
    """True where `Upsample`'s nearest-x2 + conv runs as four 2x2 convs on the low-resolution map (not under the training tape: its conv node and
    adjoint keep the gather form)."""
    return _UP2_SUBPIXEL and Cin % 64 == 0 and Cout % 8 == 0 and (not want_stats or (H * W) % 32 == 0) and not (_TAPE is not None and _TAPE.active)


def conv3x3_up2(x, w4, bias, B, H, W, out=None, colstats=None):
    """conv3x3(nearest_upsample_x2(x)) on channels-last rows: x [B*H*W, Cin] bf16, w4 = `pack_conv3x3_up2(weight)` -> ([B*2H*2W, Cout], 2H, 2W).
    colstats: optional `colstats_buffer(B*4*H*W, Cout)` for the consuming GroupNorm."""
    _chk(x, BF16, "conv3x3_up2.x", 2)
    _chk(w4, BF16, "conv3x3_up2.w4", 3)
    Cin, Cout = x.shape[1], w4.shape[1]
    if x.shape[0] != B * H * W or not x.is_contiguous():
        raise ValueError(f"conv3x3_up2: x must be contiguous [B*H*W, Cin] = [{B * H * W}, Cin], got {tuple(x.shape)}")
    if tuple(w4.shape) != (4, Cout, 4 * Cin) or not w4.is_contiguous():
        raise ValueError(f"conv3x3_up2: packed weight must be a contiguous [4, Cout, {4 * Cin}] tensor, got {tuple(w4.shape)}")
    if out is None:
        out = torch.empty(B * 4 * H * W, Cout, dtype=BF16, device=x.device)
    if colstats is not None:
        _chk(colstats, torch.float32, "conv3x3_up2.colstats", 3)
        if tuple(colstats.shape) != (B * 4 * H * W // 32, Cout, 2) or not colstats.is_contiguous() or (H * W) % 32:
            raise ValueError(f"conv3x3_up2: colstats must be a contiguous [{B * 4 * H * W // 32}, {Cout}, 2] fp32 buffer (H * W % 32 == 0)")
    check(lib.ae_conv3x3_up2_bf16(_p(x), _p(w4), _p(bias), _p(out), B, H, W, Cin, Cout, _p(colstats), _s()), "ae_conv3x3_up2_bf16")
    return out, 2 * H, 2 * W


_STEM_IM2COL = os.environ.get("AE_STEM_IM2COL", "1") != "0"  # tuning knob (A/B): 0 = the stem conv through the implicit-GEMM kernel (Cin = 8 padded to 64 per tap)


def stem_im2col_ok(Cin, stride, upsample2x):
    """True where a 3x3 conv runs as im2col + dense GEMM: the 8-channel stem (not under the training tape, whose conv node keeps the implicit-GEMM form)."""
    return _STEM_IM2COL and Cin == 8 and stride == 1 and not upsample2x and not (_TAPE is not None and _TAPE.active)


def im2col3x3_c8(x, B, H, W):
    """x: [B*H*W, 8] bf16 channels-last -> [B*H*W, 128] bf16 (`pack_conv3x3_taps8` column order)."""
    _chk(x, BF16, "im2col3x3_c8.x", 2)
    if tuple(x.shape) != (B * H * W, 8) or not x.is_contiguous():
        raise ValueError(f"im2col3x3_c8: x must be contiguous [{B * H * W}, 8], got {tuple(x.shape)}")
    y = torch.empty(B * H * W, 128, dtype=BF16, device=x.device)
    check(lib.ae_im2col3x3_c8_bf16(_p(x), _p(y), B, H, W, _s()), "ae_im2col3x3_c8_bf16")
    return y


_CONV_KMAJOR = int(os.environ.get("AE_CONV_KMAJOR", "2"))  # tuning knob: 0 off, 1 un-split 192x320 plan, 2 + its split-K form, 3 every eligible conv, 4 un-split 192x320 and 128x128 plans


def conv_k_order(M, Cin, Cout, stride=1, upsample2x=False):
    """Which packed K order `conv3x3` should be given for this launch (the planner's mirror): the chunk-major order where the 192x320 tile
    runs (64x64 UNet level at batch >= 12: a block's nine tap windows are then re-read from L2 instead of the MALL), the tap-major order
    elsewhere (measured slower there) and always under the tape (the backward passes use rotated tap-major packs)."""
    if _CONV_KMAJOR == 0 or upsample2x or Cin % 64 != 0 or (_TAPE is not None and _TAPE.active) or os.environ.get("AE_GEMM_GLDS", "1") == "0":
        return 0
    if _CONV_KMAJOR == 3:
        return 1
    label = _tile_label(M, Cout, True, 9 * Cin, False, True)
    if _CONV_KMAJOR == 4:   # (A/B only) the un-split plans of the 64x64 and 32x32 levels
        return 1 if label in ("192x320", "128x128") else 0
    return 1 if (label == "192x320" or (_CONV_KMAJOR >= 2 and label == "192x320,splitK")) else 0


def pack_geglu(w, b):
    """GEGLU proj [2*inner, dim]: interleave 16 'a' rows with the matching 16 'gate' rows (AE_EPI_GEGLU layout)."""
    inner = w.shape[0] // 2
    assert inner % 16 == 0
    a, g = w[:inner], w[inner:]
    wp = torch.stack([a.reshape(inner // 16, 16, -1), g.reshape(inner // 16, 16, -1)], dim=1).reshape(2 * inner, -1)
    bp = torch.stack([b[:inner].reshape(-1, 16), b[inner:].reshape(-1, 16)], dim=1).reshape(-1)
    return wp.detach().to(BF16).contiguous(), bp.detach().float().contiguous()


def pack_ff2_fused(w2):
    """ff2 weight [C = 320, H] -> the LDS images of `ae_ff_fused_bf16` (csrc/ff_fused.hip), bf16 [H / 32, C, 32], once per weight version.
    Image row i of a 32-hidden-unit chunk holds output column 32 (i >> 5) + 8 ((i & 15) >> 2) + 4 ((i & 31) >> 4) + (i & 3) (a lane's two result
    fragments are eight consecutive columns: 16-byte stores).  Its 64 bytes are four 16-byte pieces; logical piece g, element e, is hidden unit
    32 s + 16 (e >> 2) + 4 g + (e & 3) — the order the gated values of chunks 2 s, 2 s + 1 sit in the P1 result registers of lane group g —
    stored at position (g + 2 ((i & 15) >> 2)) & 3 (conflict-free ds_read_b128 lane groups)."""
    C, H = w2.shape
    assert C == 320 and H % 32 == 0
    dev = w2.device
    i = torch.arange(C, device=dev)
    i5 = i & 31
    col = 32 * (i >> 5) + 8 * ((i5 & 15) >> 2) + 4 * (i5 >> 4) + (i5 & 3)
    pos = torch.arange(4, device=dev)
    e = torch.arange(8, device=dev)
    gp = (pos[None, :] - 2 * ((i[:, None] & 15) >> 2)) & 3                                         # [C, 4]: the logical piece stored at (row, position)
    u = 16 * (e >> 2)[None, None, :] + 4 * gp[:, :, None] + (e & 3)[None, None, :]                  # [C, 4, 8] hidden unit inside the chunk
    wf = w2.detach().float()[col]                                                                  # rows in image order
    img = wf.reshape(C, H // 32, 32).permute(1, 0, 2)                                              # [S, C, 32]
    idx = u.reshape(1, C, 32).expand(H // 32, C, 32)
    return torch.gather(img, 2, idx).to(BF16).contiguous()


def ff_fused_ok(M, C, H):
    """True where `ff_fused` covers the feed-forward shape (never while the training tape records: its backward needs the pre-activations)."""
    return not (_TAPE is not None and _TAPE.active) and bool(lib.ae_ff_fused_supported(M, C, H))


def ff_fused(x, gamma, beta, eps, w1, b1, w2img, b2, residual=None, out=None, w3=None, b3=None, residual3=None, colstats=None):
    """out = FF(LayerNorm(x)) (+ residual) in ONE launch (attention.py:49-76 behind norm3, :271-275): x [M, 320] bf16 rows, (w1, b1) = `pack_geglu`,
    w2img = `pack_ff2_fused`.  With w3 (bf16 [320, 320], `pack_linear`): out = bf16(that) @ w3^T + b3 (+ residual3) — SpatialTransformer.proj_out and its
    residual (attention.py:337-340) in the same launch; colstats: `colstats_buffer(M, 320)` for the GroupNorm that consumes out.  Callers ask `ff_fused_ok` first."""
    _chk(x, BF16, "ff_fused.x", 2)
    _chk(w1, BF16, "ff_fused.w1", 2)
    _chk(w2img, BF16, "ff_fused.w2img", 3)
    _chk(gamma, torch.float32, "ff_fused.gamma", 1)
    _chk(beta, torch.float32, "ff_fused.beta", 1)
    _chk(b1, torch.float32, "ff_fused.b1", 1)
    M, C = x.shape
    H = w1.shape[0] // 2
    if w1.shape[1] != C or tuple(w2img.shape) != (H // 32, C, 32) or not w2img.is_contiguous() or x.stride(1) != 1 or w1.stride(1) != 1:
        raise ValueError(f"ff_fused: x {tuple(x.shape)}, w1 {tuple(w1.shape)}, w2img {tuple(w2img.shape)} do not fit together")
    if b1.numel() != 2 * H or gamma.numel() != C or beta.numel() != C:
        raise ValueError("ff_fused: b1 / gamma / beta sizes")
    if b2 is not None:
        _chk(b2, torch.float32, "ff_fused.b2", 1)
    if residual is not None:
        _chk(residual, BF16, "ff_fused.residual", 2)
    if w3 is not None:
        _chk(w3, BF16, "ff_fused.w3", 2)
        if tuple(w3.shape) != (C, C) or w3.stride(1) != 1:
            raise ValueError("ff_fused: w3 must be [C, C] with unit inner stride")
        if b3 is not None:
            _chk(b3, torch.float32, "ff_fused.b3", 1)
        if residual3 is not None:
            _chk(residual3, BF16, "ff_fused.residual3", 2)
        if colstats is not None:
            _chk(colstats, torch.float32, "ff_fused.colstats", 3)
            if tuple(colstats.shape) != ((M + 31) // 32, C, 2) or not colstats.is_contiguous():
                raise ValueError(f"ff_fused: colstats must be a contiguous [{(M + 31) // 32}, {C}, 2] fp32 buffer")
    elif b3 is not None or residual3 is not None or colstats is not None:
        raise ValueError("ff_fused: b3 / residual3 / colstats go with w3")
    if out is None:
        out = torch.empty(M, C, dtype=BF16, device=x.device)
    _chk(out, BF16, "ff_fused.out", 2)
    check(lib.ae_ff_fused_bf16(_p(x), x.stride(0), _p(gamma), _p(beta), float(eps), _p(w1), w1.stride(0), _p(b1), _p(w2img), _p(b2), _p(residual),
                               residual.stride(0) if residual is not None else 0, _p(w3), w3.stride(0) if w3 is not None else 0, _p(b3), _p(residual3),
                               residual3.stride(0) if residual3 is not None else 0, _p(colstats), _p(out), out.stride(0), M, C, H, _s()), "ae_ff_fused_bf16")
    return out


# --------------------------------------------------------------------------- fused cross-attention half (csrc/xattn_fused.hip)
def pack_xattn_wq(wq):
    """to_q weight [320, 320] (8 heads of 40) -> the Wq_h LDS images of `ae_xattn_fused_bf16`: bf16 [8, 48, 320], rows 40 .. 47 of a head zero, the 16-byte pieces of a
    row XOR-swizzled inside groups of eight (piece c of image row i at position (c & ~7) | ((c ^ (i >> 1)) & 7): the row-panel kernel's image).  Once per weight version."""
    assert tuple(wq.shape) == (320, 320)
    dev = wq.device
    img = torch.zeros(8, 48, 40, 8, dtype=torch.float32, device=dev)                       # [head][row][piece][8 elements]
    img[:, :40] = wq.detach().float().reshape(8, 40, 40, 8)
    i = torch.arange(48, device=dev)[:, None]
    c = torch.arange(40, device=dev)[None, :]
    pos = (c & ~7) | ((c ^ (i >> 1)) & 7)                                                   # [48, 40]: where piece c of row i is stored
    out = torch.zeros_like(img)
    out.scatter_(2, pos[None, :, :, None].expand(8, 48, 40, 8), img)
    return out.reshape(8, 48, 320).to(BF16).contiguous()


def pack_xattn_wo(wo):
    """to_out weight [320, 320 = 8 heads x 40] -> bf16 [4 head pairs, 320, 112]: image row i holds output column 32 (i >> 5) + 8 ((i & 15) >> 2) + 4 ((i & 31) >> 4) + (i & 3);
    its bytes are three K steps of 64 B (+ 32 B pad, row stride 224 B: conflict-free fragment reads); K step t, lane group g, element e is d slot 16 (q % 3) + 4 g + (e & 3) of head
    2 pair + (q >= 3), q = 2 t + (e >> 2) — the order the attention output of two heads sits in the result registers — and zero for slots 40 .. 47."""
    assert tuple(wo.shape) == (320, 320)
    dev = wo.device
    i = torch.arange(320, device=dev)
    i5 = i & 31
    col = 32 * (i >> 5) + 8 * ((i5 & 15) >> 2) + 4 * (i5 >> 4) + (i5 & 3)
    s = torch.arange(96, device=dev)
    t, g, e = s >> 5, (s >> 3) & 3, s & 7
    q6 = 2 * t + (e >> 2)
    dslot = 16 * (q6 % 3) + 4 * g + (e & 3)
    hloc = (q6 >= 3).long()
    wf = torch.cat([wo.detach().float()[col].reshape(320, 8, 40), torch.zeros(320, 8, 8, device=dev)], dim=2)   # [row, head, 48 slots]
    out = torch.zeros(4, 320, 112, dtype=torch.float32, device=dev)
    for pr in range(4):
        out[pr, :, :96] = wf[:, 2 * pr + hloc, dslot]
    return out.to(BF16).contiguous()


def pack_xattn_kv(kv, kv_ip, B, Nk, T):
    """K | V of a cross-attention layer's context [B * Nk, 640] (and of its T-token expert segment [B * T, 640] or None) -> the per-(sample, head) LDS images of
    `ae_xattn_fused_bf16`, bf16 [B, 8, 14848]: K image [96 key rows (80 text, 16 expert)][80] (64 d slots in the order the q results sit in registers + 16 pad: 160-byte rows)
    then V^T image [48 d rows][144] (4 K steps of 32 key slots in the logits' order + 16 pad; row 40 = ones: the softmax denominator) + 256 pad.  Step-invariant: once per edit."""
    assert 64 < Nk <= 80 and 0 <= T <= 16
    dev = kv.device
    kvb = kv.reshape(B, Nk, 640).float()
    Kt = torch.zeros(B, 96, 320, device=dev)
    Vt = torch.zeros(B, 96, 320, device=dev)
    Kt[:, :Nk], Vt[:, :Nk] = kvb[:, :, :320], kvb[:, :, 320:]
    if kv_ip is not None and T > 0:
        ipb = kv_ip.reshape(B, T, 640).float()
        Kt[:, 80:80 + T], Vt[:, 80:80 + T] = ipb[:, :, :320], ipb[:, :, 320:]
    s = torch.arange(64, device=dev)
    t, g, e = s >> 5, (s >> 3) & 3, s & 7
    dslot = 16 * (2 * t + (e >> 2)) + 4 * g + (e & 3)                                       # 0 .. 63 (>= 40: zero)
    Kh = torch.cat([Kt.reshape(B, 96, 8, 40).permute(0, 2, 1, 3), torch.zeros(B, 8, 96, 24, device=dev)], dim=3)
    Kimg = torch.zeros(B, 8, 96, 80, device=dev)
    Kimg[..., :64] = Kh[..., dslot]
    s = torch.arange(128, device=dev)
    t, g, e = s >> 5, (s >> 3) & 3, s & 7
    j = e >> 2
    key = torch.where(t < 3, 16 * (2 * t + j) + 4 * g + (e & 3), 80 + 4 * g + (e & 3))      # text K steps: key index; expert K step: rows 80 ..
    ok = torch.where(t < 3, key < 80, j == 0)
    Vh = Vt.reshape(B, 96, 8, 40).permute(0, 2, 3, 1)                                       # [B, 8, 40, 96]
    Vimg = torch.zeros(B, 8, 48, 144, device=dev)
    Vimg[:, :, :40, :128] = Vh[..., key.clamp(max=95)] * ok.float()
    Vimg[:, :, 40, :128] = 1.0
    out = torch.zeros(B, 8, 14848, device=dev)
    out[..., :7680] = Kimg.reshape(B, 8, 7680)
    out[..., 7680:7680 + 6912] = Vimg.reshape(B, 8, 6912)
    return out.to(BF16).contiguous()


def xattn_fused_ok(M, C, heads, head_dim, rows_per_sample, Nk, T):
    """True where `xattn_fused` covers the cross-attention half of a transformer block (never while the training tape records)."""
    return not (_TAPE is not None and _TAPE.active) and bool(lib.ae_xattn_fused_supported(M, C, heads, head_dim, rows_per_sample, Nk, T))


def xattn_fused(x, gamma, beta, eps, wq_img, kv_img, gate, wo_img, bo, rows_per_sample, Nk, T, scale, out=None):
    """out = to_out(Attn(to_q(LayerNorm(x)), K, V) + gate_b Attn(., K_ip, V_ip)) + x in ONE launch (attention.py:273 `x = attn2(norm2(x), context) + x`): x [M, 320] bf16
    rows, (wq_img, wo_img, kv_img) = `pack_xattn_wq / _wo / _kv`, gate fp32 [B] or None.  Callers ask `xattn_fused_ok` first."""
    _chk(x, BF16, "xattn_fused.x", 2)
    _chk(wq_img, BF16, "xattn_fused.wq_img", 3)
    _chk(wo_img, BF16, "xattn_fused.wo_img", 3)
    _chk(kv_img, BF16, "xattn_fused.kv_img", 3)
    _chk(gamma, torch.float32, "xattn_fused.gamma", 1)
    _chk(beta, torch.float32, "xattn_fused.beta", 1)
    M, C = x.shape
    B = M // rows_per_sample
    if tuple(wq_img.shape) != (8, 48, 320) or tuple(wo_img.shape) != (4, 320, 112) or tuple(kv_img.shape) != (B, 8, 14848) or not (wq_img.is_contiguous() and wo_img.is_contiguous()
                                                                                                                                    and kv_img.is_contiguous()) or x.stride(1) != 1:
        raise ValueError(f"xattn_fused: x {tuple(x.shape)}, images {tuple(wq_img.shape)} / {tuple(wo_img.shape)} / {tuple(kv_img.shape)} do not fit together")
    if gate is not None:
        _chk(gate, torch.float32, "xattn_fused.gate", 1)
        if gate.numel() != B:
            raise ValueError("xattn_fused: one gate per sample")
    if bo is not None:
        _chk(bo, torch.float32, "xattn_fused.bo", 1)
    if out is None:
        out = torch.empty(M, C, dtype=BF16, device=x.device)
    _chk(out, BF16, "xattn_fused.out", 2)
    check(lib.ae_xattn_fused_bf16(_p(x), x.stride(0), _p(gamma), _p(beta), float(eps), _p(wq_img), _p(kv_img), _p(gate), _p(wo_img), _p(bo), _p(out), out.stride(0), M,
                                  rows_per_sample, Nk, T, float(scale), _s()), "ae_xattn_fused_bf16")
    return out


# --------------------------------------------------------------------------- GEMM / conv
def colstats_buffer(M, N, device):
    """fp32 [ceil(M/32), N, 2]: per-channel (sum, sum of squares) over each 32-row slab of a bf16 [M, N] activation — filled by the kernel
    that PRODUCES the activation (`colstats=` of gemm / ln_gemm / conv3x3) and consumed by `groupnorm(..., colstats=...)`, which then
    skips its own statistics pass (the GroupNorm inputs of the 64x64 / 32x32 UNet levels: SURVEY.md §7 hard part (iii))."""
    return torch.empty((M + 31) // 32, N, 2, dtype=torch.float32, device=device)


_GN_COLSTATS = os.environ.get("AE_GN_COLSTATS", "1") != "0"  # tuning knob: 0 = GroupNorm computes its own statistics everywhere (A/B)


def want_colstats(HW):
    """True where a GroupNorm over HW positions per sample takes its statistics from the producer: maps above the single-launch slab
    kernel's range (HW > 256) whose samples are whole 32-row slabs; never while the training tape records (its GroupNorm keeps
    (mean, rstd) for the backward pass through the plain path)."""
    return _GN_COLSTATS and HW > 256 and HW % 32 == 0 and not (_TAPE is not None and _TAPE.active)


def gemm(a, w, bias=None, residual=None, addvec=None, rows_per_batch=0, epilogue=EPI_NONE, out_f32=False, a2=None, out=None, colstats=None,
         rowstats=None):
    """out[M,N] = epi(cat([a, a2], 1) @ w^T).  a: [M,K1] bf16 (row stride free), w: [N,K] bf16.
    colstats: optional `colstats_buffer(M, N)` that receives the per-channel slab statistics of the output.
    rowstats: optional `rowstats_buffer(M, N)` that receives the per-row statistics of the output for the LayerNorm folded into the
    GEMM that consumes it (`gemm_ln`); only where `ln_fold_plan(M, N, K, EPI_NONE, 1)` says the tile plan carries that epilogue."""
    if rowstats is not None:
        if _TAPE is not None and _TAPE.active:
            raise ValueError("gemm: row statistics are not recorded on the training tape (ln_fold_plan is False while it records)")
        if addvec is not None or a2 is not None or out_f32 or colstats is not None or epilogue != EPI_NONE:
            raise ValueError("gemm: rowstats goes with a plain bf16 GEMM (+bias, +residual) only")
        return _gemm_ln(a, w, bias, residual, EPI_NONE, out, rowstats, None, None, 0.0)
    if _TAPE is not None and _TAPE.active:
        return _TAPE.gemm(a, w, bias=bias, residual=residual, addvec=addvec, rows_per_batch=rows_per_batch, epilogue=epilogue, out_f32=out_f32, a2=a2, out=out)
    _chk(a, BF16, "gemm.a", 2)
    _chk(w, BF16, "gemm.w", 2)
    M, K1 = a.shape
    N, K = w.shape
    if a.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError("gemm: a and w must have unit inner stride")
    K2 = 0
    if a2 is not None:
        _chk(a2, BF16, "gemm.a2", 2)
        K2 = a2.shape[1]
        if a2.shape[0] != M or a2.stride(1) != 1:
            raise ValueError("gemm: a2 shape mismatch")
    if K1 + K2 != K:
        raise ValueError(f"gemm: K mismatch: a has {K1}+{K2} columns, w has {K}")
    n_out = N // 2 if epilogue == EPI_GEGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=torch.float32 if out_f32 else BF16, device=a.device)
    if bias is not None:
        _chk(bias, torch.float32, "gemm.bias", 1)
    if residual is not None:
        _chk(residual, BF16, "gemm.residual", 2)
    if addvec is not None:
        _chk(addvec, torch.float32, "gemm.addvec", 2)
    if colstats is not None:
        _chk(colstats, torch.float32, "gemm.colstats", 3)
        if tuple(colstats.shape) != ((M + 31) // 32, n_out, 2) or not colstats.is_contiguous():
            raise ValueError(f"gemm: colstats must be a contiguous [{(M + 31) // 32}, {n_out}, 2] fp32 buffer")
    if a2 is None and addvec is None and not out_f32 and _rowpanel_ok(a, w, out, residual, M, N, K, epilogue):
        return _ln_gemm_launch(a, w, bias, residual, None, None, 0.0, epilogue, out, M, N, K, colstats)
    check(lib.ae_gemm_bf16(_p(a), a.stride(0), _p(a2), a2.stride(0) if a2 is not None else 0, K1, _p(w), w.stride(0),
                           _p(out), out.stride(0), M, N, K, _p(bias), _p(residual),
                           residual.stride(0) if residual is not None else 0, _p(addvec),
                           addvec.stride(0) if addvec is not None else 0, rows_per_batch, epilogue,
                           1 if out_f32 else 0, _p(colstats), _s()), "ae_gemm_bf16")
    return out


# --------------------------------------------------------------------------- LayerNorm folded into the consuming GEMM
_LN_FOLD = os.environ.get("AE_LN_FOLD", "1") != "0"  # tuning knob (A/B): 0 = a LayerNorm launch in front of every K = 640 / 1280 projection
_LN_PLAN = {}


def ln_fold_plan(M, N, K, epilogue, mode):
    """True where the tile plan `gemm` takes for [M, K] x [N, K]^T carries the row-statistics epilogue (mode 1: this GEMM PRODUCES the
    rows a LayerNorm reads) or the LayerNorm-fold epilogue (mode 2: this GEMM CONSUMES the LayerNorm) — `ae_gemm_ln_plan`, cached per
    shape.  Never while the training tape records (its LayerNorm keeps what the backward pass needs)."""
    if not _LN_FOLD or (_TAPE is not None and _TAPE.active):
        return False
    key = (M, N, K, epilogue, mode)
    r = _LN_PLAN.get(key)
    if r is None:
        # shapes of the row-panel kernel (K = 320, the 64x64 level): round 5 gave that kernel the fold forms too (AE_RP_FOLD, read by the library: its
        # plan query says yes for them); with the row-panel kernel switched off (AE_GEMM_ROWPANEL=0, A/B) the tiled plan answers
        r = _LN_PLAN[key] = bool(lib.ae_gemm_ln_plan(M, N, K, epilogue, mode)) and \
            (_ROWPANEL or not lib.ae_ln_gemm_supported(M, N, K, epilogue) or os.environ.get("AE_RP_FOLD", "1") == "0")
    return r


def rowstats_buffer(M, N, device):
    """fp32 [M, N / 64, 2]: (sum, sum of squares) of every row of a bf16 [M, N] activation over each 64-column slice — filled by the GEMM
    that PRODUCES the activation (`gemm(..., rowstats=)`), consumed by `gemm_ln`."""
    return torch.empty(M, N // 64, 2, dtype=torch.float32, device=device)


def pack_ln_fold(w, bias, gamma, beta, geglu=False):
    """Folds LayerNorm's affine part into the projection behind it (attention.py:271-275): LN(x) W^T + b = rstd (x W'^T - mu s) + c.
    Returns (W' = W diag(gamma) as bf16 [N, K] (GEGLU: rows interleaved as `pack_geglu` does), s = row sums of the bf16 W' — exactly what
    the MFMA multiplies —, c = W beta + b), s and c fp32 [N] in the packed row order.  Once per weight version (host side)."""
    wf = w.detach().float().reshape(w.shape[0], -1).contiguous()
    g, be = gamma.detach().float(), beta.detach().float()
    c = linear_f32(be[None, :].contiguous(), wf, bias)[0]
    wp = wf * g[None, :]
    if geglu:
        wq, c = pack_geglu(wp, c)
    else:
        wq = wp.to(BF16).contiguous()
    s = linear_f32(torch.ones(1, wq.shape[1], dtype=torch.float32, device=wq.device), wq.float())[0]
    return wq, s.contiguous(), c.contiguous()


def _gemm_ln(a, w, bias, residual, epilogue, out, rowstats_out, ln_stats, ln_colsum, eps):
    _chk(a, BF16, "gemm_ln.a", 2)
    _chk(w, BF16, "gemm_ln.w", 2)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K or a.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError(f"gemm_ln: a is [{M}, {K}], w is {tuple(w.shape)}; both need unit inner stride")
    n_out = N // 2 if epilogue == EPI_GEGLU else N
    if out is None:
        out = torch.empty(M, n_out, dtype=BF16, device=a.device)
    _chk(out, BF16, "gemm_ln.out", 2)
    if bias is not None:
        _chk(bias, torch.float32, "gemm_ln.bias", 1)
    if residual is not None:
        _chk(residual, BF16, "gemm_ln.residual", 2)
    nparts = 0
    if rowstats_out is not None:
        if rowstats_out.dtype != torch.float32 or tuple(rowstats_out.shape) != (M, N // 64, 2) or not rowstats_out.is_contiguous():
            raise ValueError(f"gemm: rowstats must be a contiguous fp32 [{M}, {N // 64}, 2] buffer (rowstats_buffer)")
    else:
        if ln_stats.dtype != torch.float32 or ln_stats.dim() != 3 or ln_stats.shape[0] != M or ln_stats.shape[2] != 2 or not ln_stats.is_contiguous() \
                or ln_stats.shape[1] * 64 != K:
            raise ValueError(f"gemm_ln: statistics must be the producer's contiguous fp32 [{M}, {K // 64}, 2] buffer, got {tuple(ln_stats.shape)}")
        _chk(ln_colsum, torch.float32, "gemm_ln.s", 1)
        if bias is None or bias.numel() != N or ln_colsum.numel() != N:
            raise ValueError("gemm_ln: s and c must be fp32 [N]")
        nparts = ln_stats.shape[1]
    return _gemm_ln_launch(a, w, bias, residual, epilogue, out, M, N, K, rowstats_out, ln_stats, nparts, ln_colsum, float(eps))


def _gemm_ln_launch(a, w, bias, residual, epilogue, out, M, N, K, rowstats_out, ln_stats, nparts, ln_colsum, eps):
    check(lib.ae_gemm_ln_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K, _p(bias), _p(residual),
                              residual.stride(0) if residual is not None else 0, epilogue, _p(rowstats_out), _p(ln_stats), nparts,
                              _p(ln_colsum), eps, _s()), "ae_gemm_ln_bf16")
    return out


def gemm_ln(x, rowstats, wq, s, c, eps, residual=None, epilogue=EPI_NONE, out=None):
    """out = epi(LayerNorm(x) @ w^T + b (+ residual)) with the LayerNorm folded away: x are the UN-normalised rows, `rowstats` their
    statistics from the GEMM that produced them, (wq, s, c) = `pack_ln_fold(w, b, gamma, beta)`.  Callers ask `ln_fold_plan(M, N, K,
    epilogue, 2)` first and keep `ln_gemm` where it says no."""
    return _gemm_ln(x, wq, c, residual, epilogue, out, None, rowstats, s, eps)


_ROWPANEL = os.environ.get("AE_GEMM_ROWPANEL", "1") != "0"  # tuning knob: 0 = tiled kernel only (A/B)


def _rowpanel_ok(a, w, out, residual, M, N, K, epilogue):
    if not _ROWPANEL or not lib.ae_ln_gemm_supported(M, N, K, epilogue):
        return False
    ts = [a, w, out] + ([residual] if residual is not None else [])
    # the row-panel kernel addresses A, C and the residual with 32-bit byte offsets (buffer descriptors, per-block base offsets): shapes
    # whose rows x stride reach 2 GiB fall back to the tiled kernel instead of wrapping
    ld_max = max(t.stride(0) for t in ts if t is not w)
    return all(t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and t.stride(1) == 1 for t in ts) and N * w.stride(0) * 2 < 2 ** 31 \
        and (M + 192) * ld_max * 2 < 2 ** 31


def _ln_gemm_launch(a, w, bias, residual, gamma, beta, eps, epilogue, out, M, N, K, colstats=None):
    check(lib.ae_ln_gemm_bf16(_p(a), a.stride(0), _p(w), w.stride(0), _p(out), out.stride(0), M, N, K, _p(bias), _p(residual),
                              residual.stride(0) if residual is not None else 0, _p(gamma), _p(beta), float(eps), epilogue,
                              _p(colstats), _s()),
          "ae_ln_gemm_bf16")
    return out


def ln_gemm(x, gamma, beta, eps, w, bias=None, residual=None, epilogue=EPI_NONE, out=None, colstats=None):
    """out = epi(LayerNorm(x; gamma, beta, eps) @ w^T + bias (+ residual)) — attention.py:271-275's norm -> projection pairs in one
    launch (the normalised rows never go to HBM).  Falls back to layernorm + gemm where the fused kernel does not cover the shape
    and while the training tape records."""
    if not (_TAPE is not None and _TAPE.active):
        _chk(x, BF16, "ln_gemm.x", 2)
        _chk(w, BF16, "ln_gemm.w", 2)
        M, K = x.shape
        N = w.shape[0]
        if w.shape[1] == K and x.stride(1) == 1 and w.stride(1) == 1:
            n_out = N // 2 if epilogue == EPI_GEGLU else N
            o = out if out is not None else torch.empty(M, n_out, dtype=BF16, device=x.device)
            if _rowpanel_ok(x, w, o, residual, M, N, K, epilogue):
                _chk(gamma, torch.float32, "ln_gemm.gamma", 1)
                _chk(beta, torch.float32, "ln_gemm.beta", 1)
                if bias is not None:
                    _chk(bias, torch.float32, "ln_gemm.bias", 1)
                if residual is not None:
                    _chk(residual, BF16, "ln_gemm.residual", 2)
                return _ln_gemm_fused(x, w, bias, residual, gamma, beta, eps, epilogue, o, M, N, K, colstats)
    return gemm(layernorm(x, gamma, beta, eps), w, bias, residual=residual, epilogue=epilogue, out=out, colstats=colstats)


def conv3x3(x, w, bias, B, H, W, addvec=None, residual=None, stride=1, upsample2x=False, out_f32=False, out=None, colstats=None, k_order=0):
    """x: [B*H*W, Cin] bf16 channels-last, w: packed [Cout, 9*Cin] (`pack_conv3x3` with the same k_order).  Returns ([B*Ho*Wo, Cout], Ho, Wo).
    colstats: optional `colstats_buffer(B*Ho*Wo, Cout)` that receives the per-channel slab statistics of the output."""
    if _TAPE is not None and _TAPE.active:
        if k_order:
            raise ValueError("conv3x3: the tape records convolutions on tap-major packed weights (k_order=0)")
        return _TAPE.conv3x3(x, w, bias, B, H, W, addvec=addvec, residual=residual, stride=stride, upsample2x=upsample2x, out_f32=out_f32, out=out)
    _chk(x, BF16, "conv3x3.x", 2)
    _chk(w, BF16, "conv3x3.w", 2)
    Cin = x.shape[1]
    Cout = w.shape[0]
    if x.shape[0] != B * H * W or not x.is_contiguous():
        raise ValueError(f"conv3x3: x must be contiguous [B*H*W, Cin] = [{B * H * W}, Cin], got {tuple(x.shape)}")
    if w.shape[1] != 9 * ((Cin + 63) // 64 * 64):
        raise ValueError(f"conv3x3: packed weight has K={w.shape[1]}, expected {9 * ((Cin + 63) // 64 * 64)}")
    Hv, Wv = (2 * H, 2 * W) if upsample2x else (H, W)
    Ho, Wo = (Hv - 1) // stride + 1, (Wv - 1) // stride + 1
    if out is None:
        out = torch.empty(B * Ho * Wo, Cout, dtype=torch.float32 if out_f32 else BF16, device=x.device)
    if residual is not None:
        _chk(residual, BF16, "conv3x3.residual", 2)
        if residual.shape != (B * Ho * Wo, Cout) or not residual.is_contiguous():
            raise ValueError("conv3x3: residual shape mismatch")
    nws = lib.ae_conv3x3_workspace_floats(B, H, W, Cin, Cout, stride, int(upsample2x))
    ws = torch.empty(nws, dtype=torch.float32, device=x.device) if nws > 0 else None  # split-K partials (small-M layers)
    if colstats is not None:
        _chk(colstats, torch.float32, "conv3x3.colstats", 3)
        if tuple(colstats.shape) != ((B * Ho * Wo + 31) // 32, Cout, 2) or not colstats.is_contiguous():
            raise ValueError(f"conv3x3: colstats must be a contiguous [{(B * Ho * Wo + 31) // 32}, {Cout}, 2] fp32 buffer")
    check(lib.ae_conv3x3_bf16(_p(x), _p(w), _p(bias), _p(addvec), addvec.stride(0) if addvec is not None else 0, _p(residual),
                              _p(out), B, H, W, Cin, Cout, stride, int(upsample2x), 1 if out_f32 else 0, _p(ws), _p(colstats), int(k_order), _s()),
          "ae_conv3x3_bf16")
    return out, Ho, Wo


# Opt-in (AE_GN_SPLITK=1): ResBlock conv1 -> GroupNorm at the 16x16 / 8x8 levels through the split-K fold below.  Built, bit-identical, one launch fewer per ResBlock (12 per UNet
# evaluation) — and measured SLOWER in the graph: 12.345 -> 12.379 ms per evaluation (256-thread blocks) / 12.374 (1024-thread lab form), three alternating triples on one box
# (profiles/r06_v30_gn_splitk_ab.txt): the reduce launch streams the 31-63 MB of partials with the whole chip, the slab GroupNorm behind it reads 8 MB from L2; the folding
# GroupNorm reads the partials with 384 blocks in 160-byte row segments.  Default off.
_GN_SPLITK = os.environ.get("AE_GN_SPLITK", "0") == "1"


def conv3x3_gn_splitk_ok(B, H, W, Cin, Cout, groups=32):
    """True where `conv3x3_partials` + `groupnorm_splitk` replace conv3x3 (+ its split-K reduce launch) + groupnorm: a stride-1 conv whose plan cuts K (the 16x16 /
    8x8 UNet levels at batch 12) feeding a GroupNorm that runs the one-launch slab form (maps up to 256 positions).  Not under the training tape."""
    if not _GN_SPLITK or (_TAPE is not None and _TAPE.active) or H * W > 256 or Cin % 8 or Cout % 8:
        return False
    nws = lib.ae_conv3x3_workspace_floats(B, H, W, Cin, Cout, 1, 0)
    if nws <= 0:
        return False
    return bool(lib.ae_groupnorm_splitk_supported(B, H * W, Cout, groups, int(nws // (B * H * W * Cout))))


def conv3x3_partials(x, w, B, H, W, k_order=0):
    """The split-K plan of `conv3x3` (stride 1) stopped at its fp32 partials: returns (partials [splitk, B*H*W, Cout] fp32, splitk); the bias, the per-sample
    vector and the rounding belong to the consumer (`groupnorm_splitk`).  Raises where the plan does not split (ask `conv3x3_gn_splitk_ok` first)."""
    _no_tape("conv3x3_partials")
    _chk(x, BF16, "conv3x3_partials.x", 2)
    _chk(w, BF16, "conv3x3_partials.w", 2)
    Cin, Cout = x.shape[1], w.shape[0]
    if x.shape[0] != B * H * W or not x.is_contiguous() or w.shape[1] != 9 * ((Cin + 63) // 64 * 64):
        raise ValueError(f"conv3x3_partials: x [B*H*W, Cin] contiguous and w packed [Cout, 9*CinPad] expected, got {tuple(x.shape)} / {tuple(w.shape)}")
    nws = lib.ae_conv3x3_workspace_floats(B, H, W, Cin, Cout, 1, 0)
    if nws <= 0:
        raise ValueError(f"conv3x3_partials: the plan of [{B * H * W}, {Cin}] -> {Cout} does not cut K")
    ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    sk = ctypes.c_int(0)
    check(lib.ae_conv3x3_partials_bf16(_p(x), _p(w), B, H, W, Cin, Cout, _p(ws), int(k_order), ctypes.byref(sk), _s()), "ae_conv3x3_partials_bf16")
    if sk.value < 2 or sk.value * B * H * W * Cout != nws:
        raise RuntimeError(f"conv3x3_partials: plan mismatch (splitk {sk.value}, workspace {nws})")
    return ws.view(sk.value, B * H * W, Cout), sk.value


def groupnorm_splitk(partials, bias, addvec, gamma, beta, B, HW, eps, silu=False, groups=32, out=None):
    """GroupNorm(+SiLU) of x = bf16(sum_s partials[s] + bias + addvec[b]) without x ever being written (`ae_groupnorm_splitk_nhwc_bf16`): bit-identical to
    `conv3x3(..., bias, addvec)` followed by `groupnorm` on the small-map path.  partials: `conv3x3_partials`; bias fp32 [C] or None; addvec fp32 [B, C]
    (rows may be strided) or None."""
    _no_tape("groupnorm_splitk")
    sk, M, C = partials.shape
    if partials.dtype != torch.float32 or not partials.is_contiguous() or M != B * HW:
        raise ValueError(f"groupnorm_splitk: partials must be contiguous fp32 [splitk, B*HW, C], got {tuple(partials.shape)} {partials.dtype}")
    if addvec is not None and (addvec.dtype != torch.float32 or tuple(addvec.shape) != (B, C) or addvec.stride(1) != 1):
        raise ValueError(f"groupnorm_splitk: addvec must be fp32 [B, C] with unit column stride, got {tuple(addvec.shape)} {addvec.dtype}")
    if out is None:
        out = torch.empty(M, C, dtype=BF16, device=partials.device)
    check(lib.ae_groupnorm_splitk_nhwc_bf16(_p(partials), sk, _p(bias), _p(addvec), addvec.stride(0) if addvec is not None else 0, _p(gamma), _p(beta), _p(out),
                                            B, HW, C, groups, eps, 1 if silu else 0, _s()), "ae_groupnorm_splitk_nhwc_bf16")
    return out


# --------------------------------------------------------------------------- norms
_GN_COUNTERS = {}
_GN_LAB_SKIP = os.environ.get("AE_GN_LAB_SKIP_FINALIZE") == "1"   # see csrc/norm.hip: upper bound of removing the finalize launch
_GN_LAB_WS = {}


_GN_TAIL = os.environ.get("AE_GN_TAIL") == "1"   # the last-block finalize is a measured loss on MI355X: opt-in only


def _gn_counters(device, B):
    """int32 tickets for the last-block statistics fold (zero between launches), or None while that variant is off.  One buffer per
    (device, stream): launches on a stream are ordered, so they can share it."""
    if not _GN_TAIL:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    c = _GN_COUNTERS.get(key)
    if c is None or c.numel() < B:
        c = _GN_COUNTERS[key] = torch.zeros(max(B, 256), dtype=torch.int32, device=device)
    return c


def groupnorm(x, gamma, beta, B, HW, eps, silu=False, groups=32, x2=None, out=None, stat_out=None, colstats=None, colstats2=None):
    """GroupNorm(+SiLU) over channels-last [B*HW, C]; x2: optional second tensor concatenated along channels.
    stat_out: optional fp32 [B, groups, 2] that receives (mean, rstd) for `groupnorm_bwd`.
    colstats / colstats2: per-channel slab statistics of x / x2 from the kernels that produced them (`colstats_buffer`): the statistics
    pass over the activation is skipped.  Both or neither for a two-source input."""
    if _TAPE is not None and _TAPE.active:
        return _TAPE.groupnorm(x, gamma, beta, B, HW, eps, silu=silu, groups=groups, x2=x2, out=out)
    _chk(x, BF16, "groupnorm.x", 2)
    C1 = x.shape[1]
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    if not x.is_contiguous() or (x2 is not None and not x2.is_contiguous()):
        raise ValueError("groupnorm: inputs must be contiguous")
    if out is None:
        out = torch.empty(B * HW, C, dtype=BF16, device=x.device)
    if colstats is not None and (x2 is None) != (colstats2 is None):
        colstats = colstats2 = None   # statistics of only one of the two sources: fall back to the kernel's own pass
    if colstats is not None:
        for t, c, n in ((colstats, C1, "colstats"), (colstats2, C - C1, "colstats2")):
            if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (B * HW // 32, c, 2) or not t.is_contiguous() or HW % 32):
                raise ValueError(f"groupnorm: {n} must be a contiguous fp32 [{B * HW // 32}, {c}, 2] buffer (HW % 32 == 0), got {tuple(t.shape)}")
    nws = int(lib.ae_groupnorm_workspace_floats(B, HW, C, groups))
    if _GN_LAB_SKIP and colstats is not None:   # lab (timing only, wrong results): the finalize launch is left out, the apply reads an all-zero coefficient buffer
        ws = _GN_LAB_WS.get((nws, x.device))
        if ws is None:
            ws = _GN_LAB_WS[(nws, x.device)] = torch.zeros(nws, dtype=torch.float32, device=x.device)
            rpc = int(lib.ae_groupnorm_rows_per_chunk(HW, C))
            off = ((B * ((HW + rpc - 1) // rpc) * groups * 2 + 3) // 4) * 4
            ws[off:off + B * 2 * C].view(B, 2, C)[:, 0] = 1.0   # scale 1, shift 0: the data keeps flowing (all-zero activations would run the chip at zero-data clocks)
    else:
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
    check(lib.ae_groupnorm_nhwc_bf16(_p(x), _p(x2), C1, _p(gamma), _p(beta), _p(out), B, HW, C, groups, eps,
                                     1 if silu else 0, _p(ws), _p(_gn_counters(x.device, B)), _p(stat_out), _p(colstats), _p(colstats2), _s()),
          "ae_groupnorm_nhwc_bf16")
    return out


def layernorm(x, gamma, beta, eps=1e-5, out=None):
    if _TAPE is not None and _TAPE.active:
        return _TAPE.layernorm(x, gamma, beta, eps=eps, out=out)
    _chk(x, BF16, "layernorm.x", 2)
    if not x.is_contiguous():
        raise ValueError("layernorm: x must be contiguous")
    M, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.ae_layernorm_bf16(_p(x), _p(gamma), _p(beta), _p(out), M, C, eps, _s()), "ae_layernorm_bf16")
    return out


# --------------------------------------------------------------------------- attention
def attention(q, k, v, B, H, Nq, Nk, D, scale, q_strides, k_strides, v_strides, out=None, rel_h=None, rel_w=None,
              kH=0, kW=0, key_mask=None, out_scale=None, accumulate=False, seg2=None, lse=None, lse2=None):
    """q/k/v: bf16 tensors (any shape) addressed via (batch, head, row) element strides; out: [B, Nq, H*D] bf16.
    seg2 = (k2, v2, Nk2, k2_strides, v2_strides, scale2 [B] fp32): second key/value segment with its own softmax.
    lse / lse2: optional fp32 [B, H, Nq] outputs (log2-domain log-sum-exp per segment) kept for attention_bwd."""
    if _TAPE is not None and _TAPE.active:
        return _TAPE.attention(q, k, v, B, H, Nq, Nk, D, scale, q_strides, k_strides, v_strides, out=out, key_mask=key_mask,
                               out_scale=out_scale, accumulate=accumulate, seg2=seg2, rel_h=rel_h)
    if out is None:
        out = torch.empty(B, Nq, H * D, dtype=BF16, device=q.device)
    o_strides = (Nq * H * D, D, H * D)
    if seg2 is not None:
        k2, v2, Nk2, k2_strides, v2_strides, scale2 = seg2
        seg_args = (_p(k2), _p(v2), Nk2, *k2_strides, *v2_strides, _p(scale2))
    else:
        seg_args = (None, None, 0, 0, 0, 0, 0, 0, 0, None)
    check(lib.ae_attn_fwd_bf16(_p(q), _p(k), _p(v), _p(out), B, H, Nq, Nk, D, *q_strides, *k_strides, *v_strides, *o_strides,
                               scale, _p(rel_h), _p(rel_w), kH, kW, _p(key_mask), _p(out_scale), 1 if accumulate else 0,
                               *seg_args, _p(lse), _p(lse2), _s()), "ae_attn_fwd_bf16")
    return out


_FP8_WS = {}


def attention_fp8(q, k, v, B, H, Nq, Nk, D, scale, q_strides, k_strides, v_strides, out=None, rel_h=None, rel_w=None, kH=0, kW=0):
    """fp8 (e4m3) attention forward (ae_attn_fwd_fp8): same arguments as `attention` for the no-mask / single-segment case; the
    rel-pos bias path needs kW == 64 (SAM global attention).  Accuracy is that of e4m3 operands: rel-L2 ~5e-2 against the fp32
    reference at unit-variance logits (tests/test_hip_ops.py)."""
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _chk(t, BF16, "attention_fp8." + n)
    if out is None:
        out = torch.empty(B, Nq, H * D, dtype=BF16, device=q.device)
    need = lib.ae_attn_fp8_workspace_bytes(B, H, Nq, Nk, D)
    if need <= 0:
        raise ValueError(f"attention_fp8: unsupported sizes B={B} H={H} Nq={Nq} Nk={Nk} D={D}")
    key = (q.device, torch.cuda.current_stream(q.device).cuda_stream)
    ws = _FP8_WS.get(key)
    if ws is None or ws.numel() < need:  # grow-only scratch per (device, stream): the call owns it until the next call on that stream
        ws = _FP8_WS[key] = torch.empty(need, dtype=torch.uint8, device=q.device)
    if rel_h is not None:
        _chk(rel_h, torch.float32, "attention_fp8.rel_h")
        _chk(rel_w, torch.float32, "attention_fp8.rel_w")
    check(lib.ae_attn_fwd_fp8(_p(q), _p(k), _p(v), _p(out), B, H, Nq, Nk, D, *q_strides, *k_strides, *v_strides, Nq * H * D, D, H * D,
                              float(scale), _p(rel_h), _p(rel_w), kH, kW, _p(ws), ws.numel(), _s()), "ae_attn_fwd_fp8")
    return out


def attention_bwd(q, k, v, dout, lse, B, H, Nq, Nk, D, scale, q_strides, k_strides, v_strides, dq, dk, dv, dq_strides,
                  dk_strides, dv_strides, out_scale=None, accumulate_dq=False, out=None, split_dkv=True):
    """Gradients of one attention segment (ae_attn_bwd_bf16).  dq/dk/dv: bf16 tensors written through the given element strides
    (dk = dv = None when the key/value side needs no gradient).  out: the forward output [B, Nq, H*D] when it is this segment's alone
    (lets the kernel take delta = rowsum(dout o out) up front).  Returns delta [B, H, Nq] fp32."""
    delta = torch.empty(B, H, Nq, dtype=torch.float32, device=q.device)
    o_strides = (Nq * H * D, D, H * D)
    zero3 = (0, 0, 0)
    ws = None
    if dk is not None and split_dkv:   # few-key segments (cross-attention): the dK / dV pass cuts its query tiles across blocks through fp32 partials
        nws = lib.ae_attn_bwd_workspace_floats(B, H, Nq, Nk, D)
        if nws > 0:
            ws = torch.empty(nws, dtype=torch.float32, device=q.device)
    check(lib.ae_attn_bwd_bf16(_p(q), _p(k), _p(v), _p(dout), _p(out), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, H, Nq, Nk, D,
                               *q_strides, *k_strides, *v_strides, *o_strides, *dq_strides, *(dk_strides if dk is not None else zero3),
                               *(dv_strides if dv is not None else zero3), scale, _p(out_scale), 1 if accumulate_dq else 0, _p(ws), _s()),
          "ae_attn_bwd_bf16")
    return delta


def attention_bhnd(q, k, v, scale=None, key_mask=None):
    """xformers-style entry: q [BH, Nq, D], k/v [BH, Nk, D] contiguous bf16 -> [BH, Nq, D] (attention.py:222-233)."""
    _chk(q, BF16, "attention.q", 3)
    BH, Nq, D = q.shape
    Nk = k.shape[1]
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    out = torch.empty(BH, Nq, D, dtype=BF16, device=q.device)
    scale = scale if scale is not None else D ** -0.5
    check(lib.ae_attn_fwd_bf16(_p(q), _p(k), _p(v), _p(out), BH, 1, Nq, Nk, D, Nq * D, 0, D, Nk * D, 0, D, Nk * D, 0, D,
                               Nq * D, 0, D, scale, None, None, 0, 0, _p(key_mask), None, 0, None, None, 0, 0, 0, 0, 0, 0, 0, None, None, None, _s()),
          "ae_attn_fwd_bf16")
    return out


# --------------------------------------------------------------------------- layout / elementwise
def nchw_to_rows(x, c_pad=None):
    """[B,C,H,W] (fp32 or bf16) -> channels-last bf16 [B*H*W, Cpad] (zero padded channels)."""
    B, C, H, W = x.shape
    x = x.contiguous()
    c_pad = c_pad or C
    out = torch.empty(B * H * W, c_pad, dtype=BF16, device=x.device)
    check(lib.ae_transpose_last2(_p(x), _p(out), B, C, H * W, c_pad, 1 if x.dtype == BF16 else 0, 1, _s()), "ae_transpose_last2")
    return out


def rows_to_nchw(x, B, H, W, out_dtype=torch.float32):
    """channels-last [B*H*W, C] (bf16 or fp32) -> [B,C,H,W]."""
    if _TAPE is not None and _TAPE.active:
        return _TAPE.rows_to_nchw(x, B, H, W, out_dtype=out_dtype)
    C = x.shape[1]
    x = x.contiguous()
    out = torch.empty(B, C, H, W, dtype=out_dtype, device=x.device)
    check(lib.ae_transpose_last2(_p(x), _p(out), B, H * W, C, H * W, 1 if x.dtype == BF16 else 0,
                                 1 if out_dtype == BF16 else 0, _s()), "ae_transpose_last2")
    return out


def concat_channels(a, b):
    if _TAPE is not None and _TAPE.active:
        return _TAPE.concat_channels(a, b)
    out = torch.empty(a.shape[0], a.shape[1] + b.shape[1], dtype=BF16, device=a.device)
    check(lib.ae_concat_channels_bf16(_p(a), a.shape[1], _p(b), b.shape[1], _p(out), a.shape[0], _s()), "ae_concat_channels_bf16")
    return out


def split_channels(dy, ca, a=None, b=None, want_a=True, want_b=True):
    """Adjoint of `concat_channels`: (da, db) with da (+)= dy[:, :ca], db (+)= dy[:, ca:]; `a` / `b` are existing gradient
    buffers to add into (None: a fresh buffer is written).  One launch instead of two strided copies and two adds."""
    rows, C = dy.shape
    cb = C - ca
    da = a if a is not None else (torch.empty(rows, ca, dtype=BF16, device=dy.device) if want_a else None)
    db = b if b is not None else (torch.empty(rows, cb, dtype=BF16, device=dy.device) if want_b else None)
    check(lib.ae_split_channels_bf16(_p(dy), ca, cb, _p(da), _p(db), rows, int(a is not None), int(b is not None), _s()), "ae_split_channels_bf16")
    return da, db


def timestep_embedding(t, dim, max_period=10000.0, out_f32=False):
    B = t.shape[0]
    out = torch.empty(B, dim, dtype=torch.float32 if out_f32 else BF16, device=t.device)
    t_i64 = t if t.dtype == torch.int64 else None
    t_f32 = None if t_i64 is not None else t.float().contiguous()
    check(lib.ae_timestep_embedding(_p(t_i64), _p(t_f32), None if out_f32 else _p(out), _p(out) if out_f32 else None, B, dim,
                                    float(max_period), _s()), "ae_timestep_embedding")
    return out


def ddim_step(x, eps, coeffs, branches, s0=1.0, s1=0.0, noise=None, temperature=1.0, want_pred_x0=True, want_e=False):
    """coeffs = (sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef, sigma_t) as python floats holding fp32 values."""
    _chk(x, torch.float32, "ddim_step.x")
    _chk(eps, torch.float32, "ddim_step.eps")
    x = x.contiguous()
    eps = eps.contiguous()
    n = x.numel()
    if eps.numel() != branches * n:
        raise ValueError(f"ddim_step: eps has {eps.numel()} elements, expected {branches}*{n}")
    x_prev = torch.empty_like(x)
    pred_x0 = torch.empty_like(x) if want_pred_x0 else None
    s1m, sat, sap, dirc, sig = coeffs
    e_out = torch.empty_like(x) if want_e else None  # the guidance-combined eps (what PLMS keeps in its history)
    check(lib.ae_ddim_step_f32(_p(x), _p(eps), _p(noise), _p(x_prev), _p(pred_x0), _p(e_out), n, branches, s0, s1, s1m, sat, sap,
                               dirc, sig, temperature, _s()), "ae_ddim_step_f32")
    if want_e:
        return x_prev, pred_x0, e_out
    return x_prev, pred_x0


def ddim_encode_step(x, eps, cx, ce, branches=1, scale=1.0):
    """DDIM inversion update x_next = cx*x + ce*cfg(eps) (ddim.py:253-298); eps holds `branches` stacked predictions."""
    _chk(x, torch.float32, "ddim_encode_step.x")
    _chk(eps, torch.float32, "ddim_encode_step.eps")
    x, eps = x.contiguous(), eps.contiguous()
    out = torch.empty_like(x)
    check(lib.ae_ddim_encode_step_f32(_p(x), _p(eps), _p(out), x.numel(), branches, float(scale), float(cx), float(ce), _s()),
          "ae_ddim_encode_step_f32")
    return out


def dpm_multistep(x, model_out, branches, scale, sigma_s, alpha_s, predict_x0=True, v_param=False, m_prev=None, update=None,
                  want_m=True):
    """One DPM-Solver(++) multistep evaluation: returns (m, x_next).  update = (a, b, c, inv_r0) or None (history value only)."""
    _chk(x, torch.float32, "dpm_multistep.x")
    _chk(model_out, torch.float32, "dpm_multistep.model_out")
    n = x.numel()
    if model_out.numel() != branches * n or not x.is_contiguous() or not model_out.is_contiguous():
        raise ValueError("dpm_multistep: model_out must hold `branches` contiguous copies of x's shape")
    m = torch.empty_like(x) if want_m else None
    xn = torch.empty_like(x) if update is not None else None
    a, b, c, inv_r0 = update if update is not None else (0.0, 0.0, 0.0, 0.0)
    check(lib.ae_dpm_multistep_f32(_p(x), _p(model_out), _p(m_prev), _p(m), _p(xn), n, branches, float(scale), 1 if v_param else 0,
                                   1 if predict_x0 else 0, float(sigma_s), float(alpha_s), 1 if update is not None else 0, float(a),
                                   float(b), float(c), float(inv_r0), _s()), "ae_dpm_multistep_f32")
    return m, xn


def plms_combine(e_t, old_eps):
    """PLMS combination of e_t with the list of previous predictions (newest last, as plms.py keeps them); for the first step pass
    old_eps = [e_t_next] and order 0 via `plms_combine_first`."""
    order = min(len(old_eps), 3)
    e_t = e_t.contiguous()
    olds = [t.contiguous() for t in old_eps[::-1][:3]] + [None, None, None]
    out = torch.empty_like(e_t)
    check(lib.ae_plms_combine_f32(_p(e_t), _p(olds[0]), _p(olds[1]), _p(olds[2]), _p(out), e_t.numel(), order, _s()), "ae_plms_combine_f32")
    return out


def plms_combine_first(e_t, e_t_next):
    out = torch.empty_like(e_t)
    check(lib.ae_plms_combine_f32(_p(_tmp(e_t.contiguous())), _p(_tmp(e_t_next.contiguous())), None, None, _p(out), e_t.numel(), 0, _s()),
          "ae_plms_combine_f32")
    return out


def mask_blend_shape(mask, B, C, H, W):
    """How a mask broadcastable against [B, C, H, W] (the reference's `img_orig * mask + (1 - mask) * img`, ddim.py:154-157) is handed to
    the kernel, which reads one mask plane per sample: -> (mask expanded, samples, channels) with a per-channel mask ([B or 1, C, H, W])
    run as B*C single-channel samples."""
    while mask.dim() < 4:
        mask = mask.unsqueeze(0)
    if mask.shape[1] == 1:
        return mask.expand(B, 1, H, W), B, C
    if mask.shape[1] == C:
        return mask.expand(B, C, H, W).reshape(B * C, 1, H, W), B * C, 1
    raise ValueError(f"mask_blend: mask of shape {tuple(mask.shape)} does not broadcast against [{B}, {C}, {H}, {W}]")


def mask_blend(img, x0, noise, mask, sqrt_ac, sqrt_one_minus_ac, ip2p_order=False):
    B, C, H, W = img.shape
    out = torch.empty_like(img)
    m, Bk, Ck = mask_blend_shape(mask, B, C, H, W)
    check(lib.ae_mask_blend_f32(_p(_tmp(img.contiguous())), _p(_tmp(x0.contiguous())), _p(_tmp(noise.contiguous())), _p(_tmp(m.contiguous().float())),
                                _p(out), Bk, Ck, H * W, sqrt_ac, sqrt_one_minus_ac, 1 if ip2p_order else 0, _s()), "ae_mask_blend_f32")
    return out


def q_sample(x0, noise, sqrt_ac_t, sqrt_one_minus_ac_t):
    """per-sample fp32 coefficient vectors [B] (already gathered at t)."""
    out = torch.empty_like(x0)
    B = x0.shape[0]
    check(lib.ae_q_sample_f32(_p(_tmp(x0.contiguous())), _p(_tmp(noise.contiguous())), _p(_tmp(sqrt_ac_t.contiguous())),
                              _p(_tmp(sqrt_one_minus_ac_t.contiguous())), _p(out), B, x0.numel() // B, _s()), "ae_q_sample_f32")
    return out


def silu_to_bf16(x):
    """silu(x) -> bf16 (ResBlock.emb_layers[0], openaimodel.py:212-214)."""
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=BF16, device=x.device)
    check(lib.ae_silu_to_bf16(_p(x), 1 if x.dtype == BF16 else 0, _p(out), x.numel(), _s()), "ae_silu_to_bf16")
    return out


def add_bcast(x, p):
    out = torch.empty_like(x)
    check(lib.ae_add_bcast_bf16(_p(x), _p(p), _p(out), x.numel(), p.numel(), _s()), "ae_add_bcast_bf16")
    return out


def _no_tape(what):
    if _TAPE is not None and _TAPE.active:
        raise NotImplementedError(f"{what}: forward only (no backward closure on the tape; the AnyEdit training step never builds it)")


def resample2x_rows(x, B, H, W, down=False):
    """Channels-last rows [B*H*W, C] bf16 -> nearest x2 (Upsample(use_conv=False), openaimodel.py:108-118) or, down=True, the 2x2 mean of
    avg_pool_nd (Downsample(use_conv=False), :154-155).  Returns (rows, Ho, Wo)."""
    _no_tape("resample2x_rows")
    C = x.shape[1]
    if x.dtype != BF16 or not x.is_contiguous() or x.shape[0] != B * H * W or C % 8:
        raise ValueError(f"resample2x_rows: bf16 contiguous rows [B*H*W, C % 8 == 0] expected, got {tuple(x.shape)} {x.dtype}")
    Ho, Wo = (H // 2, W // 2) if down else (2 * H, 2 * W)
    out = torch.empty(B * Ho * Wo, C, dtype=BF16, device=x.device)
    check(lib.ae_resample2x_rows_bf16(_p(x), _p(out), B, H, W, C, 1 if down else 0, _s()), "ae_resample2x_rows_bf16")
    return out, Ho, Wo


def scale_shift_rows(x, emb, B, HW, silu=True):
    """ResBlock(use_scale_shift_norm=True), openaimodel.py:264-268: act(x * (1 + scale) + shift) over GroupNorm output rows x [B*HW, C] bf16; emb fp32
    [B, 2C] (rows may be strided: a column slice of the batched projection) = scale | shift."""
    _no_tape("scale_shift_rows")
    C = x.shape[1]
    if (x.dtype != BF16 or not x.is_contiguous() or x.shape[0] != B * HW or C % 8 or emb.dtype != torch.float32 or tuple(emb.shape) != (B, 2 * C)
            or emb.stride(1) != 1):
        raise ValueError(f"scale_shift_rows: x bf16 rows [B*HW, C % 8 == 0], emb fp32 [B, 2C] expected, got {tuple(x.shape)} {x.dtype} / {tuple(emb.shape)} {emb.dtype}")
    out = torch.empty_like(x)
    check(lib.ae_scale_shift_rows_bf16(_p(x), _p(emb), emb.stride(0), _p(out), B, HW, C, 1 if silu else 0, _s()), "ae_scale_shift_rows_bf16")
    return out


def window_partition(x, B, H, W, ws):
    C = x.shape[-1]
    nH, nW = (H + ws - 1) // ws, (W + ws - 1) // ws
    out = torch.empty(B * nH * nW * ws * ws, C, dtype=BF16, device=x.device)
    check(lib.ae_window_partition_bf16(_p(x), _p(out), B, H, W, C, ws, 0, _s()), "ae_window_partition_bf16")
    return out, (nH * ws, nW * ws)


def window_unpartition(windows, B, H, W, ws):
    C = windows.shape[-1]
    out = torch.empty(B * H * W, C, dtype=BF16, device=windows.device)
    check(lib.ae_window_partition_bf16(_p(out), _p(windows), B, H, W, C, ws, 1, _s()), "ae_window_partition_bf16")
    return out


def layernorm_window_ok(C):
    return bool(lib.ae_layernorm_window_supported(int(C)))


def layernorm_window_partition(x, gamma, beta, eps, B, H, W, ws):
    """window_partition(LayerNorm(x)) in one launch (SAM Block norm1 + partition, image_encoder.py:166-173): x [B*H*W, C] image rows ->
    ([B*nH*nW*ws*ws, C] window rows with zero padding rows, (Hp, Wp))."""
    _chk(x, BF16, "layernorm_window_partition.x", 2)
    _chk(gamma, torch.float32, "layernorm_window_partition.gamma", 1)
    _chk(beta, torch.float32, "layernorm_window_partition.beta", 1)
    C = x.shape[1]
    if not x.is_contiguous() or x.shape[0] != B * H * W or gamma.numel() != C or beta.numel() != C:
        raise ValueError("layernorm_window_partition: x must be contiguous [B*H*W, C] rows, gamma / beta [C]")
    nH, nW = (H + ws - 1) // ws, (W + ws - 1) // ws
    out = torch.empty(B * nH * nW * ws * ws, C, dtype=BF16, device=x.device)
    check(lib.ae_layernorm_window_bf16(_p(x), None, _p(gamma), _p(beta), _p(out), None, B, H, W, C, ws, 1, eps, _s()),
          "ae_layernorm_window_bf16")
    return out, (nH * ws, nW * ws)


def window_merge_layernorm(windows, shortcut, gamma, beta, eps, B, H, W, ws):
    """(x, LayerNorm(x)) with x = shortcut + window_unpartition(windows) in one launch (SAM Block :175-181: un-partition, residual add, norm2)."""
    _chk(windows, BF16, "window_merge_layernorm.windows", 2)
    _chk(shortcut, BF16, "window_merge_layernorm.shortcut", 2)
    _chk(gamma, torch.float32, "window_merge_layernorm.gamma", 1)
    _chk(beta, torch.float32, "window_merge_layernorm.beta", 1)
    C = shortcut.shape[1]
    nwin_rows = B * ((H + ws - 1) // ws) * ((W + ws - 1) // ws) * ws * ws
    if not (windows.is_contiguous() and shortcut.is_contiguous()) or shortcut.shape[0] != B * H * W or tuple(windows.shape) != (nwin_rows, C) \
            or gamma.numel() != C or beta.numel() != C:
        raise ValueError("window_merge_layernorm: windows must be contiguous [B*nH*nW*ws*ws, C], shortcut [B*H*W, C], gamma / beta [C]")
    xsum, y = torch.empty_like(shortcut), torch.empty_like(shortcut)
    check(lib.ae_layernorm_window_bf16(_p(windows), _p(shortcut), _p(gamma), _p(beta), _p(y), _p(xsum), B, H, W, C, ws, 2, eps, _s()),
          "ae_layernorm_window_bf16")
    return xsum, y


def sam_relpos_terms(q, q_strides, Rh, Rw, B, heads, qH, qW, D):
    kH, kW = Rh.shape[1], Rw.shape[1]
    rel_h = torch.empty(B * heads, qH * qW, kH, dtype=torch.float32, device=q.device)
    rel_w = torch.empty(B * heads, qH * qW, kW, dtype=torch.float32, device=q.device)
    check(lib.ae_sam_relpos_terms(_p(q), *q_strides, _p(Rh), _p(Rw), _p(rel_h), _p(rel_w), B, heads, qH, qW, kH, kW, D, _s()),
          "ae_sam_relpos_terms")
    return rel_h, rel_w


def softmax_rows(S, scale):
    """softmax(scale * S) over the last dim: fp32 [R, N] (row stride free) -> bf16 [R, N]."""
    _chk(S, torch.float32, "softmax_rows.S", 2)
    out = torch.empty(S.shape, dtype=BF16, device=S.device)
    check(lib.ae_softmax_rows_f32_bf16(_p(S), S.stride(0), _p(out), out.stride(0), S.shape[0], S.shape[1], float(scale), _s()),
          "ae_softmax_rows_f32_bf16")
    return out


def gaussian_moments(moments, noise=None, want_stats=False):
    """DiagonalGaussianDistribution on moments [B, 2C, H, W] fp32: returns z (and mean, logvar, std if want_stats)."""
    moments = moments.float().contiguous()
    B, C2 = moments.shape[0], moments.shape[1]
    shape = (B, C2 // 2, *moments.shape[2:])
    z = torch.empty(shape, dtype=torch.float32, device=moments.device)
    stats = [torch.empty_like(z) for _ in range(3)] if want_stats else [None, None, None]
    if noise is not None:
        noise = noise.float().contiguous()
    check(lib.ae_gaussian_moments_f32(_p(moments), _p(noise), _p(z), _p(stats[0]), _p(stats[1]), _p(stats[2]), B, z.numel() // B, _s()),
          "ae_gaussian_moments_f32")
    return (z, *stats) if want_stats else z


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step=None):
    """Drop-in for `_C.ms_deform_attn_forward` (GroundingDINO ms_deform_attn.py:42-60): value [bs, S, heads, d], spatial_shapes
    [L, 2] int64, level_start_index [L] int64, sampling_locations [bs, Q, heads, L, P, 2], attention_weights [bs, Q, heads, L, P]
    -> [bs, Q, heads*d] fp32.  `im2col_step` is accepted for signature compatibility (it only chunked the CUDA launch)."""
    _chk(value, torch.float32, "ms_deform_attn.value", 4)
    bs, S, heads, d = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    value = value.contiguous()
    loc = sampling_locations.float().contiguous()
    w = attention_weights.float().contiguous()
    shapes = spatial_shapes.to(torch.int64).contiguous()
    starts = level_start_index.to(torch.int64).contiguous()
    if int(shapes.prod(1).sum()) != S:
        raise ValueError("ms_deform_attn: spatial_shapes do not add up to value.shape[1]")
    out = torch.empty(bs, Q, heads * d, dtype=torch.float32, device=value.device)
    check(lib.ae_ms_deform_attn_fwd_f32(_p(value), _p(shapes), _p(starts), _p(loc), _p(w), _p(out), bs, S, heads, d, Q, L, P, _s()),
          "ae_ms_deform_attn_fwd_f32")
    return out


def ms_deform_attn_bwd(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, grad_output, im2col_step=None):
    """Drop-in for `_C.ms_deform_attn_backward` (GroundingDINO ms_deform_attn.py:68-90): returns (grad_value [bs, S, heads, d],
    grad_sampling_loc [bs, Q, heads, L, P, 2], grad_attn_weight [bs, Q, heads, L, P]), fp32."""
    _chk(value, torch.float32, "ms_deform_attn_bwd.value", 4)
    bs, S, heads, d = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    value = value.contiguous()
    loc = sampling_locations.float().contiguous()
    w = attention_weights.float().contiguous()
    go = grad_output.float().contiguous()
    if tuple(go.shape) != (bs, Q, heads * d):
        raise ValueError(f"ms_deform_attn_bwd: grad_output {tuple(go.shape)} != {(bs, Q, heads * d)}")
    shapes = spatial_shapes.to(torch.int64).contiguous()
    starts = level_start_index.to(torch.int64).contiguous()
    gv, gl, gw = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(w)
    check(lib.ae_ms_deform_attn_bwd_f32(_p(value), _p(shapes), _p(starts), _p(loc), _p(w), _p(go), _p(gv), _p(gl), _p(gw), bs, S, heads, d,
                                        Q, L, P, _s()), "ae_ms_deform_attn_bwd_f32")
    return gv, gl, gw


def linear_f32(x, w, bias=None):
    """nn.Linear in exact fp32 on the f32-input MFMA (ae_linear_f32): x [..., K] fp32, w [N, K] fp32, bias [N] -> [..., N] fp32.
    For layers the reference runs in fp32 and whose outputs are coordinates (GroundingDINO MSDeformAttn); K % 16 == 0."""
    _chk(x, torch.float32, "linear_f32.x")
    _chk(w, torch.float32, "linear_f32.w", 2)
    K = x.shape[-1]
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear_f32: x has {K} features, w has {w.shape[1]}")
    x2 = _tmp(x.reshape(-1, K).contiguous())
    w2 = _tmp(w.detach().contiguous())
    b2 = _tmp(bias.detach().float().contiguous()) if bias is not None else None
    M = x2.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    if M > 0:
        check(lib.ae_linear_f32(_p(x2), K, _p(w2), K, _p(b2), _p(out), N, M, N, K, _s()), "ae_linear_f32")
    return out.reshape(*x.shape[:-1], N)


def layernorm_act(x, gamma, beta, eps=1e-6, gelu=True, out=None):
    """LayerNorm over a narrow last dim (C <= 512) with fused GELU: the LayerNorm2d + GELU of the SAM mask decoder on rows."""
    _chk(x, BF16, "layernorm_act.x", 2)
    if not x.is_contiguous():
        raise ValueError("layernorm_act: x must be contiguous")
    M, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.ae_layernorm_act_bf16(_p(x), _p(gamma), _p(beta), _p(out), M, C, eps, 1 if gelu else 0, _s()), "ae_layernorm_act_bf16")
    return out


def sam_pe_encode(coords, gauss, image_size, labels=None, table=None, offset=0.5):
    """coords [N, 2] fp32 pixel (x, y); gauss [2, F] fp32; image_size (H, W).  -> [N, 2F] fp32 (prompt_encoder.py:73-102, 183-214)."""
    _chk(coords, torch.float32, "sam_pe_encode.coords", 2)
    _chk(gauss, torch.float32, "sam_pe_encode.gauss", 2)
    N, F = coords.shape[0], gauss.shape[1]
    if labels is not None:
        _chk(labels, torch.int32, "sam_pe_encode.labels", 1)
        _chk(table, torch.float32, "sam_pe_encode.table", 2)
        if table.shape != (5, 2 * F) or labels.shape[0] != N:
            raise ValueError("sam_pe_encode: table must be [5, 2F] and labels [N]")
    out = torch.empty(N, 2 * F, dtype=torch.float32, device=coords.device)
    if N == 0:
        return out
    check(lib.ae_sam_pe_encode_f32(_p(_tmp(coords.contiguous())), _p(labels), _p(_tmp(gauss.contiguous())), _p(table), _p(out), N, F, float(offset),
                                   1.0 / image_size[1], 1.0 / image_size[0], _s()), "ae_sam_pe_encode_f32")
    return out


def sam_mask_downscale(masks, w1, b1, g1, e1, w2, b2, g2, e2, eps=1e-6):
    """masks [B, 1, 4h, 4w] fp32 -> rows [B*h*w, 16] bf16 (PromptEncoder.mask_downscaling[0:6], mask_in_chans 16)."""
    _chk(masks, torch.float32, "sam_mask_downscale.masks", 4)
    B, one, H4, W4 = masks.shape
    if one != 1 or H4 % 4 or W4 % 4 or tuple(w1.shape) != (4, 1, 2, 2) or tuple(w2.shape) != (16, 4, 2, 2):
        raise ValueError("sam_mask_downscale: expects [B,1,4h,4w] masks and the mask_in_chans=16 weights")
    out = torch.empty(B * (H4 // 4) * (W4 // 4), 16, dtype=BF16, device=masks.device)
    check(lib.ae_sam_mask_downscale_bf16(_p(_tmp(masks.contiguous())), _p(w1), _p(b1), _p(g1), _p(e1), _p(w2), _p(b2), _p(g2), _p(e2), _p(out),
                                         B, H4 // 4, W4 // 4, eps, _s()), "ae_sam_mask_downscale_bf16")
    return out


def sam_mask_product(up, hyper, B, h, w):
    """up: bf16 rows [B*h*w*16, C] in un-shuffled (b, y, x, dy1, dx1, dy2, dx2) order; hyper [B, M, C] fp32 -> [B, M, 4h, 4w] fp32."""
    _chk(up, BF16, "sam_mask_product.up", 2)
    _chk(hyper, torch.float32, "sam_mask_product.hyper", 3)
    M, C = hyper.shape[1], hyper.shape[2]
    if up.shape != (B * h * w * 16, C) or not up.is_contiguous() or not hyper.is_contiguous() or hyper.shape[0] != B:
        raise ValueError("sam_mask_product: shape mismatch")
    out = torch.empty(B, M, 4 * h, 4 * w, dtype=torch.float32, device=up.device)
    check(lib.ae_sam_mask_product_f32(_p(up), _p(hyper), _p(out), B, h, w, M, C, _s()), "ae_sam_mask_product_f32")
    return out


def sam_postprocess_masks(low, img_size, input_size, original_size, threshold=None, want_logits=True, merge=False):
    """Sam.postprocess_masks fused (sam.py:133-162).  Returns (logits fp32 or None, bool mask or None).
    merge: the bool mask is the union over all B*M masks, shape [1, 1, oh, ow] (no logits)."""
    _chk(low, torch.float32, "sam_postprocess_masks.low", 4)
    B, M, Hl, Wl = low.shape
    oh, ow = int(original_size[0]), int(original_size[1])
    if merge and (threshold is None or want_logits):
        raise ValueError("sam_postprocess_masks: merge returns the thresholded union only (threshold required, want_logits=False)")
    logits = torch.empty(B, M, oh, ow, dtype=torch.float32, device=low.device) if want_logits else None
    mask = torch.empty((1, 1, oh, ow) if merge else (B, M, oh, ow), dtype=torch.uint8, device=low.device) if threshold is not None else None
    if B * M > 0:
        check(lib.ae_sam_postprocess_masks(_p(_tmp(low.contiguous())), _p(logits), _p(mask), B * M, Hl, Wl, int(img_size), int(input_size[0]),
                                           int(input_size[1]), oh, ow, float(threshold if threshold is not None else 0.0),
                                           1 if merge else 0, _s()), "ae_sam_postprocess_masks")
    elif mask is not None:
        mask.zero_()
    return logits, (mask.view(torch.bool) if mask is not None else None)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms: XYXY boxes [N, 4], scores [N] -> int64 indices of the kept boxes, by decreasing score."""
    if not boxes.is_cuda:
        raise ValueError("nms: expected GPU tensors (anyedit_amd has no CPU path)")
    if boxes.dim() != 2 or boxes.shape[1] != 4 or scores.shape != boxes.shape[:1]:
        raise ValueError(f"nms: boxes must be [N, 4] and scores [N], got {tuple(boxes.shape)} / {tuple(scores.shape)}")
    N = boxes.shape[0]
    if N == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    order = torch.sort(scores, descending=True, stable=True).indices
    keep = torch.empty(N, dtype=torch.uint8, device=boxes.device)
    check(lib.ae_nms_sorted_f32(_p(_tmp(boxes.float()[order].contiguous())), _p(keep), N, float(iou_threshold), _s()), "ae_nms_sorted_f32")
    return order[keep.bool()]


def sam_preprocess(x, img_size, mean, std):
    """Sam.preprocess (sam.py:164-174): x [B, C, h, w] uint8 or fp32 -> normalised, zero-padded [B, C, S, S] fp32."""
    if x.dtype not in (torch.uint8, torch.float32):
        x = x.float()
    if not x.is_cuda:
        raise ValueError("sam_preprocess: expected a GPU tensor (anyedit_amd has no CPU path)")
    B, C, h, w = x.shape
    out = torch.empty(B, C, img_size, img_size, dtype=torch.float32, device=x.device)
    check(lib.ae_sam_preprocess_f32(_p(_tmp(x.contiguous())), 1 if x.dtype == torch.uint8 else 0, _p(out), B, C, h, w, img_size, _p(mean),
                                    _p(std), _s()), "ae_sam_preprocess_f32")
    return out


def patchify(x, P):
    B, Cin, H, W = x.shape
    out = torch.empty(B * (H // P) * (W // P), Cin * P * P, dtype=BF16, device=x.device)
    check(lib.ae_patchify_f32_bf16(_p(_tmp(x.float().contiguous())), _p(out), B, Cin, H, W, P, _s()), "ae_patchify_f32_bf16")
    return out


def lincomb(terms, out=None):
    """sum_i c_i * x_i over up to four (fp32 tensor, scalar) terms of one shape (ae_lincomb4_f32): the general DPM-Solver updates."""
    terms = [(t, float(c)) for t, c in terms if t is not None]
    if not 1 <= len(terms) <= 4:
        raise ValueError("lincomb: one to four terms")
    x0 = terms[0][0]
    for t, _ in terms:
        _chk(t, torch.float32, "lincomb.term")
        if t.shape != x0.shape or not t.is_contiguous():
            raise ValueError("lincomb: terms must be contiguous tensors of one shape")
    if out is None:
        out = torch.empty_like(x0)
    ptr = [(_p(t), c) for t, c in terms] + [(None, 0.0)] * (4 - len(terms))
    check(lib.ae_lincomb4_f32(_p(out), ptr[0][0], ptr[0][1], ptr[1][0], ptr[1][1], ptr[2][0], ptr[2][1], ptr[3][0], ptr[3][1], x0.numel(), _s()),
          "ae_lincomb4_f32")
    return out


def dpm_adaptive_err(x_lower, x_higher, x_prev, atol, rtol):
    """Per-sample error norm of the adaptive DPM-Solver (ae_dpm_adaptive_err_f32) -> fp32 [B]."""
    for t in (x_lower, x_higher, x_prev):
        _chk(t, torch.float32, "dpm_adaptive_err")
        if t.shape != x_lower.shape or not t.is_contiguous():
            raise ValueError("dpm_adaptive_err: contiguous tensors of one shape")
    B = x_lower.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=x_lower.device)
    check(lib.ae_dpm_adaptive_err_f32(_p(x_lower), _p(x_higher), _p(x_prev), float(atol), float(rtol), B, x_lower.numel() // B, _p(out), _s()),
          "ae_dpm_adaptive_err_f32")
    return out


def mse(a, b):
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    check(lib.ae_mse_f32(_p(_tmp(a.float().contiguous())), _p(_tmp(b.float().contiguous())), _p(out), a.numel(), _s()), "ae_mse_f32")
    return out[0]


def task_gate(task_emb, edit_code, Wg, bg):
    B = edit_code.shape[0]
    n_tasks, Dt = task_emb.shape
    E = Wg.shape[0]
    probs = torch.empty(B, E, dtype=torch.float32, device=task_emb.device)
    top1 = torch.empty(B, dtype=torch.int32, device=task_emb.device)
    top1p = torch.empty(B, dtype=torch.float32, device=task_emb.device)
    check(lib.ae_task_gate(_p(_tmp(task_emb.float().contiguous())), _p(_tmp(edit_code.long().contiguous())), _p(_tmp(Wg.float().contiguous())),
                           _p(_tmp(bg.float().contiguous())) if bg is not None else None, B, n_tasks, Dt, E, _p(probs), _p(top1),
                           _p(top1p), _s()), "ae_task_gate")
    return probs, top1, top1p


# --------------------------------------------------------------------------- per-op profiler (bench.py roofline leg)
# --------------------------------------------------------------------------- training-step kernels (row A11)
_TAPE = None  # set by anyedit_amd.autodiff.Tape while a differentiated forward is being recorded


def add(a, b, out=None):
    """a + b, bf16, same shape (gradient accumulation)."""
    a, b = a.contiguous(), b.contiguous()
    if out is None:
        out = torch.empty_like(a)
    check(lib.ae_add_bf16(_p(a), _p(b), _p(out), a.numel(), _s()), "ae_add_bf16")
    return out


def axpy(a, b, alpha=1.0, out=None):
    """a + alpha * b (bf16, same shape)."""
    a, b = a.contiguous(), b.contiguous()
    if out is None:
        out = torch.empty_like(a)
    check(lib.ae_axpy_bf16(_p(a), _p(b), float(alpha), _p(out), a.numel(), _s()), "ae_axpy_bf16")
    return out


def geglu(h):
    """h = [a | g] [M, 2F] bf16 -> a * gelu(g) [M, F] (attention.py:49-57, un-fused so that h is kept for the backward)."""
    if _TAPE is not None and _TAPE.active:
        return _TAPE.geglu(h)
    M, F2 = h.shape
    out = torch.empty(M, F2 // 2, dtype=BF16, device=h.device)
    check(lib.ae_geglu_fwd_bf16(_p(_tmp(h.contiguous())), _p(out), M, F2 // 2, _s()), "ae_geglu_fwd_bf16")
    return out


def geglu_bwd(h, dy):
    dh = torch.empty_like(h)
    check(lib.ae_geglu_bwd_bf16(_p(_tmp(h.contiguous())), _p(_tmp(dy.contiguous())), _p(dh), h.shape[0], h.shape[1] // 2, _s()), "ae_geglu_bwd_bf16")
    return dh


def _grad_target(t, into):
    """`into`: a contiguous bf16 tensor of t's shape that already holds a gradient of t (the kernel then adds to it in place), or None."""
    if into is None:
        return torch.empty_like(t), 0
    if into.dtype != BF16 or into.shape != t.shape or not into.is_contiguous():
        raise ValueError("gradient accumulation target must be a contiguous bf16 tensor of the input's shape")
    return into, 1


def groupnorm_bwd(x, gamma, beta, dy, B, HW, eps, silu=False, groups=32, x2=None, stat=None, dx_into=None, dx2_into=None):
    """dx_into / dx2_into: tensors that already hold a gradient of x / x2 — the kernel adds to them in place (no separate add launch) and they are returned."""
    C1 = x.shape[1]
    C = C1 + (x2.shape[1] if x2 is not None else 0)
    dx, acc1 = _grad_target(x, dx_into)
    dx2, acc2 = _grad_target(x2, dx2_into) if x2 is not None else (None, 0)
    ws = torch.empty(lib.ae_groupnorm_bwd_workspace_floats(B, HW, C, groups), dtype=torch.float32, device=x.device)
    check(lib.ae_groupnorm_bwd_nhwc_bf16(_p(x), _p(x2), C1, _p(gamma), _p(beta), _p(_tmp(dy.contiguous())), _p(dx), _p(dx2), B, HW, C, groups,
                                         eps, 1 if silu else 0, _p(ws), _p(_gn_counters(x.device, B)), _p(stat), acc1 | (acc2 << 1), _s()), "ae_groupnorm_bwd_nhwc_bf16")
    return dx, dx2


def layernorm_bwd(x, gamma, dy, eps=1e-5, want_param_grads=False, dx_into=None):
    M, C = x.shape
    dx, acc = _grad_target(x, dx_into)
    stat = torch.empty(M, 2, dtype=torch.float32, device=x.device) if want_param_grads else None
    dy = dy.contiguous()
    check(lib.ae_layernorm_bwd_bf16(_p(x), _p(gamma), _p(dy), _p(dx), _p(stat), M, C, eps, acc, _s()), "ae_layernorm_bwd_bf16")
    if not want_param_grads:
        return dx, None, None
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    check(lib.ae_layernorm_param_grad_f32(_p(x), _p(dy), _p(stat), _p(dg), _p(db), M, C, _s()), "ae_layernorm_param_grad_f32")
    return dx, dg, db


def sumpool2x2(x, B, H, W):
    """x: [B*2H*2W, C] -> [B*H*W, C] (adjoint of the nearest-x2 upsample)."""
    out = torch.empty(B * H * W, x.shape[1], dtype=BF16, device=x.device)
    check(lib.ae_sumpool2x2_bf16(_p(_tmp(x.contiguous())), _p(out), B, H, W, x.shape[1], _s()), "ae_sumpool2x2_bf16")
    return out


def colsum(x):
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    check(lib.ae_colsum_bf16_f32(_p(x), _p(out), x.shape[0], x.shape[1], x.stride(0), _s()), "ae_colsum_bf16_f32")
    return out


def mse_grad(pred, target, loss_scale=1.0):
    pred, target = pred.contiguous(), target.contiguous()
    out = torch.empty_like(pred)
    check(lib.ae_mse_grad_f32(_p(pred), _p(target), _p(out), pred.numel(), float(loss_scale), _s()), "ae_mse_grad_f32")
    return out


def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
    """In-place torch.optim.AdamW update of one fp32 parameter tensor."""
    for t in (param, grad, exp_avg, exp_avg_sq):
        _chk(t, torch.float32, "adamw")
        if not t.is_contiguous():
            raise ValueError("adamw_step: tensors must be contiguous")
    check(lib.ae_adamw_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, betas[0], betas[1], eps, weight_decay,
                           int(step), float(grad_scale), _s()), "ae_adamw_f32")


def rowsum_f32(x):
    """x [R, n] fp32 -> [R] (fixed-order sums)."""
    x = x.contiguous()
    out = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(lib.ae_rowsum_f32(_p(x), _p(out), x.shape[0], x.shape[1], _s()), "ae_rowsum_f32")
    return out


def scatter_add_rows(src, code, dst):
    """dst[code[b]] += src[b] (fp32), deterministic."""
    code = code.to(torch.int32).contiguous()
    check(lib.ae_scatter_add_rows_f32(_p(_tmp(src.contiguous())), _p(code), _p(dst), src.shape[0], src.shape[1], dst.shape[0], _s()),
          "ae_scatter_add_rows_f32")
    return dst


def task_gate_bwd(probs, top1, dgate, Wg):
    B, E = probs.shape
    dte = torch.empty(B, Wg.shape[1], dtype=torch.float32, device=probs.device)
    check(lib.ae_task_gate_bwd(_p(_tmp(probs.contiguous())), _p(_tmp(top1.contiguous())), _p(_tmp(dgate.float().contiguous())), _p(_tmp(Wg.float().contiguous())),
                               B, Wg.shape[1], E, _p(dte), _s()), "ae_task_gate_bwd")
    return dte


def task_gate_wgrad(probs, top1, dgate, task_emb, edit_code):
    """(dWg [E, Dt], dbg [E]) of the routed gate value (router parameters of the AnySD spec)."""
    B, E = probs.shape
    n_tasks, Dt = task_emb.shape
    probs32, top1c, dg32 = probs.float().contiguous(), top1.contiguous(), dgate.float().contiguous()
    te32, code = task_emb.float().contiguous(), edit_code.long().contiguous()
    dW = torch.empty(E, Dt, dtype=torch.float32, device=probs.device)
    db = torch.empty(E, dtype=torch.float32, device=probs.device)
    check(lib.ae_task_gate_wgrad(_p(probs32), _p(top1c), _p(dg32), _p(te32), _p(code), B, n_tasks, Dt, E, _p(dW), _p(db), _s()),
          "ae_task_gate_wgrad")
    return dW, db


_EXPERT_KV_WS = {}


def _experts_i32(experts):
    return experts.to(torch.int32).contiguous()


def expert_kv(x, W, experts, tokens):
    """kv[b*T+t] = x[b*T+t] @ bf16(W[experts[b]])^T for every sample in ONE launch (training forward of the AnySD adapters).
    x [B*T, Dc] bf16, W [E, N, Dc] fp32 master, experts [B] int -> [B*T, N] bf16."""
    E, N, Dc = W.shape
    B = x.shape[0] // tokens
    assert x.dtype == BF16 and x.is_contiguous() and W.dtype == torch.float32 and W.is_contiguous() and x.shape == (B * tokens, Dc)
    ex = _tmp(_experts_i32(experts))
    y = torch.empty(B * tokens, N, dtype=BF16, device=x.device)
    check(lib.ae_expert_kv_fwd(_p(x), _p(W), _p(ex), _p(y), B, tokens, N, Dc, E, _s()), "ae_expert_kv_fwd")
    return y


def expert_kv_dgrad(dy, W, experts, tokens):
    """dx[b*T+t] = dy[b*T+t] @ bf16(W[experts[b]]) (bf16), fixed summation order."""
    E, N, Dc = W.shape
    B = dy.shape[0] // tokens
    assert dy.dtype == BF16 and dy.is_contiguous() and dy.shape == (B * tokens, N)
    S = lib.ae_expert_kv_dgrad_slices(N)
    key = (dy.device, torch.cuda.current_stream(dy.device).cuda_stream)
    need = S * B * tokens * Dc
    ws = _EXPERT_KV_WS.get(key)
    if ws is None or ws.numel() < need:   # grow-only scratch per (device, stream): launches on a stream are ordered
        ws = _EXPERT_KV_WS[key] = torch.empty(need, dtype=torch.float32, device=dy.device)
    ex = _tmp(_experts_i32(experts))
    dx = torch.empty(B * tokens, Dc, dtype=BF16, device=dy.device)
    check(lib.ae_expert_kv_dgrad(_p(dy), _p(W), _p(ex), _p(dx), B, tokens, N, Dc, E, _p(ws), _s()), "ae_expert_kv_dgrad")
    return dx


def expert_kv_wgrad(dy, x, experts, tokens, n_experts, out=None):
    """dW[e] = sum over the samples routed to e of dy[b]^T x[b] (fp32 [E, N, Dc]; zeros for experts nothing was routed to)."""
    B = dy.shape[0] // tokens
    N, Dc = dy.shape[1], x.shape[1]
    assert dy.dtype == BF16 and x.dtype == BF16 and dy.is_contiguous() and x.is_contiguous()
    if out is None:
        out = torch.empty(n_experts, N, Dc, dtype=torch.float32, device=dy.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == n_experts * N * Dc
    ex = _tmp(_experts_i32(experts))
    check(lib.ae_expert_kv_wgrad(_p(dy), _p(x), _p(ex), _p(out), B, tokens, N, Dc, n_experts, _s()), "ae_expert_kv_wgrad")
    return out


class OpProfiler:
    """Records (kernel label, algorithmic flops, algorithmic bytes, HIP-event duration) for every GEMM / conv / attention /
    norm launch issued through this module while active.  Events are recorded on torch's current stream, which is the
    stream the kernels are launched on."""

    def __init__(self):
        self.records = []

    def __enter__(self):
        global _PROF
        _PROF = self
        return self

    def __exit__(self, *a):
        global _PROF
        _PROF = None
        torch.cuda.synchronize()

    def summary(self, by_shape=False):
        agg = {}
        for label, flops, nbytes, e0, e1, nlaunch in self.records:
            if not by_shape:
                label = label.split("|")[0]
            ms = e0.elapsed_time(e1)
            a = agg.setdefault(label, {"calls": 0, "launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["calls"] += 1
            a["launches"] += nlaunch   # kernel launches behind the call (GroupNorm: 1-3, split-K conv: + the reduce; mirrors of the C launchers)
            a["ms"] += ms
            a["flops"] += flops
            a["bytes"] += nbytes
        for a in agg.values():
            a["avg_us"] = 1e3 * a["ms"] / a["calls"]
            a["tflops"] = a["flops"] / (a["ms"] * 1e-3) / 1e12 if a["ms"] > 0 else 0.0
            a["gbps"] = a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0.0
        return agg


_PROF = None


def _tile_label(M, N, conv=False, K=0, geglu=False, dma_ok=True, xe2=False):
    """Mirror of the tile choice in csrc/gemm_conv.hip::launch / make_plan (for labelling only)."""
    if conv and dma_ok and _conv_t320_split(M, N, K):
        return "192x320,splitK"
    if dma_ok and N % 320 == 0 and (conv or K >= 640):   # convs, GEGLU GEMMs with K >= 640 and (round 3) the other dense GEMMs with K >= 640
        tm = -(-M // 192)
        t = tm * (N // 320)
        fill = t / (-(-t // 256) * 256) * (M / (tm * 192))
        if fill >= 0.85 and not (conv and _conv_splitk(M, N, K) > 1):
            return "192x320"
        t128 = -(-M // 128) * -(-N // 128)
        if xe2 and not conv and not geglu and 0.74 <= fill < 0.85 and t128 / (-(-t128 // 512) * 512) <= 0.72 and os.environ.get("AE_GEMM_T320_XE", "1") != "0":
            return "192x320"   # round 5: a LayerNorm-fold launch whose 128x128 grid leaves a bad tail (qkv of the 16x16 level)
    if conv and _conv_splitk(M, N, K) > 1:  # small grids: one block per CU under the three-stage ring (a different kernel symbol)
        return "128x128,ring3,splitK" if dma_ok and _CONV_DEEP and -(-M // 128) * -(-N // 128) * _conv_splitk(M, N, K) <= 256 else "128x128,splitK"
    if conv and N % 160 == 0 and N % 128 != 0 and -(-M // 128) * (N // 160) >= 256:
        return "128x160"
    if not conv and not geglu and dma_ok and N % 128 == 0 and K >= 1280 and 128 <= -(-M // 128) * (N // 128) <= 256:
        return "128x128,ring3"  # three-stage ring, one block per CU
    if not conv and not geglu and dma_ok and N % 128 == 0 and K >= 2560 and -(-M // 128) * (N // 128) > 256 and 128 <= -(-M // 192) * (N // 128) <= 256:
        return "192x128,ring3"
    for bm, bn in ((128, 128), (128, 64), (64, 64)):
        tm, tn = -(-M // bm), -(-N // bn)
        if tm * tn >= 256 and tn * bn / N <= 1.10:
            if bm == 64 and not conv and dma_ok and K >= 1280 and tm * tn <= 768:
                return "64x64,ring3"
            return f"{bm}x{bn}"
    if not conv and dma_ok and K >= 1280 and -(-M // 64) * -(-N // 64) <= 768:
        return "64x64,ring3"
    return "64x64"


_CONV_DEEP = os.environ.get("AE_CONV_DEEP", "0") != "0"  # mirror of the library's knob (small conv grids under the three-stage ring; default off since round 3)


def _conv_t320_split(M, N, K):
    """split count of the 192x320 split-K plan (16x16-level convs), 0 when it does not apply (make_plan, tile id 4)."""
    kt = -(-K // 64)
    knob = int(os.environ.get("AE_CONV_T320_SPLITK", "3"))   # mirror of make_plan's knob: 2 = grids of 32..64 tiles, 3 (default, round 5) = + 65..128 tiles with >= 180 K tiles, 1 = every grid up to 128
    hi = 64 if (knob == 2 or (knob == 3 and kt < 180)) else 128
    if knob and N % 320 == 0 and M % 192 == 0 and 32 <= (M // 192) * (N // 320) <= hi:
        s_ = min(256 // ((M // 192) * (N // 320)), 8)
        while s_ > 1 and kt // s_ < 16:
            s_ -= 1
        return s_ if s_ >= 2 else 0
    return 0


def _conv_splitk(M, N, K):
    kt = -(-K // 64)
    if _conv_t320_split(M, N, K):
        return _conv_t320_split(M, N, K)
    t128 = -(-M // 128) * -(-N // 128)
    picked_big = any(-(-M // bm) * -(-N // bn) >= 256 and -(-N // bn) * bn / N <= 1.10 for bm, bn in ((128, 128),))
    is160 = N % 160 == 0 and N % 128 != 0 and -(-M // 128) * (N // 160) >= 256
    if kt < 32 or picked_big or is160:
        return 1
    if -(-N // 128) * 128 / N <= 1.10 and t128 < 256:
        s_ = min((256 // t128) if (t128 <= 128 and _CONV_DEEP) else -(-480 // t128), int(os.environ.get("AE_CONV_SPLIT_MAX", "8")), kt // 8)
        return s_ if s_ >= 2 else 1
    return 1


def _wrap_profiled(fn, label_fn):
    def wrapped(*args, **kwargs):
        if _PROF is None:
            return fn(*args, **kwargs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kwargs)
        e1.record()
        lab = label_fn(out, *args, **kwargs)
        if lab is not None:   # None: an inner wrapped launch already recorded this call
            _PROF.records.append((lab[0], lab[1], lab[2], e0, e1, lab[3] if len(lab) > 3 else 1))
        return out
    wrapped.__doc__ = fn.__doc__
    return wrapped


def _gemm_label(_r, a, w, bias=None, residual=None, addvec=None, rows_per_batch=0, epilogue=EPI_NONE, out_f32=False, a2=None, **_):
    if _.get("rowstats") is not None:
        return None   # recorded by _gemm_ln_launch as "...,rowstats"
    M, (N, K) = a.shape[0], w.shape
    nb = 2 * (M * K + N * K) + _r.numel() * _r.element_size() + (2 * M * N if residual is not None else 0)
    if a2 is None and addvec is None and not out_f32 and _rowpanel_ok(a, w, _r, residual, M, N, K, epilogue):
        return f"gemm_rowpanel_kernel<K=320{',geglu' if epilogue == EPI_GEGLU else ''}>|M={M} N={N}", 2.0 * M * N * K, nb
    return f"gemm_kernel<{_tile_label(M, N, False, K, epilogue == EPI_GEGLU, K % 64 == 0 and a2 is None)},dense>|M={M} N={N} K={K}", 2.0 * M * N * K, float(nb)


def _conv_label(_r, x, w, bias, B, H, W, addvec=None, residual=None, stride=1, upsample2x=False, out_f32=False, **_):
    y = _r[0]
    M, Cout, Cin = y.shape[0], w.shape[0], x.shape[1]
    nb = 2 * (x.numel() + 9 * Cin * Cout) + y.numel() * y.element_size() + (2 * y.numel() if residual is not None else 0)
    tl = _tile_label(M, Cout, True, 9 * (-(-Cin // 64) * 64), False, Cin % 64 == 0)
    return f"gemm_kernel<{tl},conv3x3>|M={M} Cin={Cin} Cout={Cout} s{stride}{'u' if upsample2x else ''}", 2.0 * M * Cout * 9 * Cin, float(nb), (2 if "splitK" in tl else 1)


def _attn_label(_r, q, k, v, B, H, Nq, Nk, D, *a, **kw):
    # mirror of ae_attn_fast_launch: attention_fast.hip takes head dims 40 / 80 / 160 without key mask (rel-pos bias at d = 80 only, in its
    # kW == 64 or small-window forms); with <= 128 keys (and <= 64 in a second segment) its short-K/V variant runs
    fast = D in (40, 80, 160) and kw.get("key_mask") is None and os.environ.get("AE_ATTN_FAST", "1") != "0" and \
        (D != 160 or os.environ.get("AE_ATTN_FAST160", "1") != "0") and \
        (kw.get("rel_h") is None or (D == 80 and ((kw.get("kW") == 64 and Nk % 64 == 0) or (kw.get("kH", 99) <= 16 and kw.get("kW", 99) <= 16))))
    seg2 = kw.get("seg2")
    skv = fast and kw.get("rel_h") is None and Nk <= 128 and (seg2 is None or seg2[2] <= 64) and os.environ.get("AE_ATTN_SKV", "1") != "0"
    name = ("attn_fast_kernel" if fast else "attn_kernel")
    return f"{name}<D={D}{',shortKV' if skv else ''}>|Nq={Nq} Nk={Nk}", 4.0 * B * H * Nq * Nk * D, 2.0 * B * H * D * (2 * Nq + 2 * Nk)


def _attn8_label(_r, q, k, v, B, H, Nq, Nk, D, *a, **kw):
    return f"fp8_attn_kernel<D={D}>+prepare|Nq={Nq} Nk={Nk}", 4.0 * B * H * Nq * Nk * D, 2.0 * B * H * D * (2 * Nq + 2 * Nk)


def _gn_label(_r, x, gamma, beta, B, HW, eps, silu=False, groups=32, x2=None, **_):
    # launches behind the call (mirror of ae_groupnorm_nhwc_bf16): producer statistics -> finalize_cs + apply; maps up to 16x16 -> the one-launch
    # slab kernel; otherwise stats + finalize + apply (stats + apply with the last-block fold, AE_GN_TAIL)
    nl = 2 if _.get("colstats") is not None and ((x2 is None) == (_.get("colstats2") is None)) else (1 if HW <= 256 and os.environ.get("AE_GN_SLAB", "1") != "0" else (2 if _GN_TAIL else 3))
    return f"groupnorm(stats+apply)|rows={_r.shape[0]} C={_r.shape[1]}", 0.0, 2.0 * _r.numel() * 2, nl  # 1 read + 1 write algorithmic


def _ln_label(_r, x, *a, **_):
    return f"layernorm_kernel|M={_r.shape[0]} C={_r.shape[1]}", 0.0, 2.0 * _r.numel() * 2


def _ln_gemm_label(_r, x, w, bias, residual, gamma, beta, eps, epilogue, o, M, N, K, colstats=None):
    nb = 2 * (M * K + N * K) + _r.numel() * 2 + (2 * M * N if residual is not None else 0)
    return f"gemm_rowpanel_kernel<K=320,LN{',geglu' if epilogue == EPI_GEGLU else ''}>|M={M} N={N}", 2.0 * M * N * K, nb


def _gemm_ln_label(_r, a, w, bias, residual, epilogue, out, M, N, K, rowstats_out, ln_stats, nparts, ln_colsum, eps):
    nb = 2 * (M * K + N * K) + _r.numel() * 2 + (2 * M * N if residual is not None else 0)
    tag = "rowstats" if rowstats_out is not None else "LNfold"
    if _ROWPANEL and os.environ.get("AE_RP_FOLD", "1") != "0" and lib.ae_ln_gemm_supported(M, N, K, epilogue) and (rowstats_out is None or epilogue == EPI_NONE):
        # round 5: the K = 320 shapes run the row-panel kernel's fold forms (ae_gemm_ln_bf16 forwards them)
        return f"gemm_rowpanel_kernel<K=320,{tag}{',geglu' if epilogue == EPI_GEGLU else ''}>|M={M} N={N}", 2.0 * M * N * K, nb
    return f"gemm_kernel<{_tile_label(M, N, False, K, epilogue == EPI_GEGLU, True, xe2=rowstats_out is None)},dense,{tag}>|M={M} N={N} K={K}", 2.0 * M * N * K, float(nb)


def _up2_label(_r, x, w4, bias, B, H, W, **_):
    y = _r[0]
    M, Cout, Cin = x.shape[0], w4.shape[1], x.shape[1]
    t192 = (M // 192) * (Cout // 320) * 4 if (M % 192 == 0 and Cout % 320 == 0) else 0
    tl = "192x320" if t192 and t192 / (-(-t192 // 256) * 256) >= 0.85 else "128x128"
    nb = 2 * (x.numel() + 16 * Cin * Cout) + 2 * y.numel()
    # FLOP: the 2x2 form's own count (4 taps per output pixel); the gather form it replaces spends 9/4 of it for the same function
    return f"gemm_kernel<{tl},conv3x3,up2x2>|M={y.shape[0]} Cin={Cin} Cout={Cout}", 2.0 * y.shape[0] * Cout * 4 * Cin, float(nb)


conv3x3_up2 = _wrap_profiled(conv3x3_up2, _up2_label)
xattn_fused = _wrap_profiled(xattn_fused, lambda _r, x, gamma, beta, eps, wq_img, kv_img, gate, wo_img, bo, rows_per_sample, Nk, T, scale, out=None: (
    f"xattn_fused_kernel<C=320>|M={x.shape[0]} Nk={Nk}+{T}", 2.0 * x.shape[0] * 320 * (2 * 320 + 2 * (Nk + T)), float(2 * (3 * x.numel() + wq_img.numel() + wo_img.numel() + kv_img.numel()))))
ff_fused = _wrap_profiled(ff_fused, lambda _r, x, gamma, beta, eps, w1, b1, w2img, b2, residual=None, out=None, w3=None, b3=None, residual3=None, colstats=None: (
    f"ff_fused_kernel<C=320{',proj_out' if w3 is not None else ''}>|M={x.shape[0]} H={w1.shape[0] // 2}",
    2.0 * x.shape[0] * x.shape[1] * (3 * (w1.shape[0] // 2) + (x.shape[1] if w3 is not None else 0)),
    float(2 * (x.numel() * (2 + (residual is not None and residual is not x) + (residual3 is not None)) + w1.numel() + w2img.numel() + (w3.numel() if w3 is not None else 0))),
    2 if colstats is not None else 1))
_ln_gemm_fused = _wrap_profiled(_ln_gemm_launch, _ln_gemm_label)
_gemm_ln_launch = _wrap_profiled(_gemm_ln_launch, _gemm_ln_label)
gemm = _wrap_profiled(gemm, _gemm_label)
attention_fp8 = _wrap_profiled(attention_fp8, _attn8_label)
conv3x3 = _wrap_profiled(conv3x3, _conv_label)
conv3x3_partials = _wrap_profiled(conv3x3_partials, lambda _r, x, w, B, H, W, k_order=0: (
    f"gemm_kernel<{_tile_label(x.shape[0], w.shape[0], True, 9 * (-(-x.shape[1] // 64) * 64), False, x.shape[1] % 64 == 0)},conv3x3>|M={x.shape[0]} Cin={x.shape[1]} Cout={w.shape[0]} s1 partials",
    2.0 * x.shape[0] * w.shape[0] * 9 * x.shape[1], float(2 * (x.numel() + 9 * x.shape[1] * w.shape[0]) + 2 * x.shape[0] * w.shape[0]), 1))
groupnorm_splitk = _wrap_profiled(groupnorm_splitk, lambda _r, partials, *a, **k: (
    f"groupnorm(splitK fold)|rows={_r.shape[0]} C={_r.shape[1]}", 0.0, 2.0 * _r.numel() * 2, 1))
attention = _wrap_profiled(attention, _attn_label)
groupnorm = _wrap_profiled(groupnorm, _gn_label)
layernorm = _wrap_profiled(layernorm, _ln_label)
expert_kv = _wrap_profiled(expert_kv, lambda _r, x, W, experts, tokens: (
    f"expert_kv_fwd_kernel|rows={x.shape[0]} N={W.shape[1]} Dc={W.shape[2]}", 2.0 * x.shape[0] * W.shape[1] * W.shape[2], 4.0 * (x.shape[0] // tokens) * W.shape[1] * W.shape[2]))
expert_kv_dgrad = _wrap_profiled(expert_kv_dgrad, lambda _r, dy, W, experts, tokens: (
    f"expert_kv_dgrad_kernel|rows={dy.shape[0]} N={W.shape[1]} Dc={W.shape[2]}", 2.0 * dy.shape[0] * W.shape[1] * W.shape[2], 4.0 * (dy.shape[0] // tokens) * W.shape[1] * W.shape[2]))
expert_kv_wgrad = _wrap_profiled(expert_kv_wgrad, lambda _r, dy, x, experts, tokens, n_experts, out=None: (
    f"expert_kv_wgrad_kernel|rows={dy.shape[0]} N={dy.shape[1]} Dc={x.shape[1]} E={n_experts}", 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1], 4.0 * n_experts * dy.shape[1] * x.shape[1]))
