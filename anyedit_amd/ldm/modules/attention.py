"""Mirror of ldm/modules/attention.py (reference file:line cited per class) on HIP kernels.

Public signatures, sub-module names and state-dict keys are the reference's, so a checkpoint / YAML written for
`ldm.modules.attention` loads unchanged.  Public `forward`s accept the reference layouts ([B,N,C] tokens or [B,C,H,W]
feature maps, any float dtype) and return the same; internally everything moves as channels-last bf16 rows through
`*_rows` methods, which is what UNetModel calls (no layout changes between layers).
"""
import os

import torch
import torch.nn as nn

from anyedit_amd import ops
from anyedit_amd.ldm.util import default, exists
from anyedit_amd.ldm.modules.diffusionmodules.util import Linear, Conv2d, LayerNorm, checkpoint  # noqa: F401

BF16 = torch.bfloat16
# tuning knob (the library reads the same variable): 1 = the cross-attention half of the 64x64-level blocks as ONE launch (ops.xattn_fused).  Off by default: parity-green
# but measured slower than the three launches it replaces (profiles/r06_xattn_fused_notes.txt)
_XATTN_FUSED = os.environ.get("AE_XATTN_FUSED", "0") != "0"
_FF_TAIL = os.environ.get("AE_FF_TAIL", "1") != "0"   # tuning knob (A/B): 0 = proj_out stays its own launch behind the fused feed-forward
_SEG2_160 = os.environ.get("AE_ATTN_SEG2_160", "1") != "0"  # tuning knob (A/B): 0 = the d = 160 expert segment as a second, accumulating launch


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class _GN6(nn.GroupNorm):
    """Normalize(): GroupNorm(32, C, eps=1e-6, affine) (attention.py:88-89)."""

    def _affine(self):
        if ops.cache_stale(self, "_pk", self.weight, self.bias):
            self._pk = (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous())
        return self._pk

    def repack(self):
        self._pk = None

    def rows(self, x, B, HW, colstats=None):
        g, b = self._affine()
        return ops.groupnorm(x, g, b, B, HW, self.eps, silu=False, groups=self.num_groups, colstats=colstats)

    def forward(self, x):
        B, C, H, W = x.shape
        return ops.rows_to_nchw(self.rows(ops.nchw_to_rows(x), B, H * W), B, H, W, out_dtype=x.dtype)


def Normalize(in_channels):
    return _GN6(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class GEGLU(nn.Module):
    """attention.py:49-58; the gate (exact-erf GELU) is fused into the projection GEMM's epilogue."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)
        self._pk = None

    def repack(self):
        self._pk = None
        self._pkln = None

    def _packed_ln(self, norm):
        """(W', s, c): the projection with `norm` folded in (ops.pack_ln_fold), rows interleaved like `_pk`."""
        if ops.cache_stale(self, "_pkln", self.proj.weight, self.proj.bias, norm.weight, norm.bias):
            self._pkln = ops.pack_ln_fold(self.proj.weight, self.proj.bias, norm.weight, norm.bias, geglu=True)
        return self._pkln

    def rows(self, x, norm=None, rowstats=None):
        """rowstats: statistics of x from the GEMM that produced it — `norm` is then folded into the projection (no LayerNorm launch)."""
        if ops._TAPE is not None and ops._TAPE.active:
            if norm is not None:
                x = norm.rows(x)
            # training step: the pre-activation [a | g] is kept for the backward, so the gate runs as its own kernel
            if ops.cache_stale(self, "_pk_plain", self.proj.weight, self.proj.bias):
                self._pk_plain = (ops.pack_linear(self.proj.weight), self.proj.bias.detach().float().contiguous())
            w, b = self._pk_plain
            return ops.geglu(ops.gemm(x, w, b))
        if ops.cache_stale(self, "_pk", self.proj.weight, self.proj.bias):
            self._pk = ops.pack_geglu(self.proj.weight, self.proj.bias)
        w, b = self._pk
        if norm is not None and rowstats is not None and ops.ln_fold_plan(x.shape[0], w.shape[0], x.shape[1], ops.EPI_GEGLU, 2):
            wq, s, c = self._packed_ln(norm)
            return ops.gemm_ln(x, rowstats, wq, s, c, norm.eps, epilogue=ops.EPI_GEGLU)
        if norm is not None:  # the block's norm3 fused in front of the projection (the caller passes the UN-normalised rows)
            g, be = norm._affine()
            return ops.ln_gemm(x, g, be, norm.eps, w, b, epilogue=ops.EPI_GEGLU)
        return ops.gemm(x, w, b, epilogue=ops.EPI_GEGLU)

    def forward(self, x):
        shp = x.shape
        return self.rows(x.reshape(-1, shp[-1]).to(BF16).contiguous()).reshape(*shp[:-1], -1).to(x.dtype)


class FeedForward(nn.Module):
    """attention.py:61-76."""

    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.):
        super().__init__()
        inner_dim = int(dim * mult)
        dim_out = default(dim_out, dim)
        self.glu = glu
        # attention.py:66-70: nn.Sequential(nn.Linear(dim, inner_dim), nn.GELU()) — the exact-erf GELU rides in the projection GEMM's epilogue (same state-dict keys:
        # net.0.0.weight / net.0.0.bias)
        project_in = GEGLU(dim, inner_dim) if glu else nn.Sequential(Linear(dim, inner_dim), nn.GELU())
        self.net = nn.Sequential(project_in, nn.Dropout(dropout), Linear(inner_dim, dim_out))

    def repack(self):
        self._pkf = None

    def fused_ok(self, M, C):
        """True where norm3 -> GEGLU -> ff2 (+ residual) runs as ONE launch (`ops.ff_fused`: the 64x64 UNet level at bench batch sizes)."""
        return self.glu and ops.ff_fused_ok(M, C, self.net[2].weight.shape[1])

    def rows(self, x, residual=None, norm=None, rowstats=None, tail=None):
        """tail: optional (w3 [C, C] bf16, b3, residual3, colstats) — the projection that follows the block (SpatialTransformer.proj_out, attention.py:337-340);
        only where `fused_ok` (the caller checks): it then rides in the fused launch and the result is proj_out's."""
        if norm is not None and self.fused_ok(x.shape[0], x.shape[1]):
            proj, ff2 = self.net[0].proj, self.net[2]
            if ops.cache_stale(self, "_pkf", proj.weight, proj.bias, ff2.weight, ff2.bias):
                w1, b1 = ops.pack_geglu(proj.weight, proj.bias)
                self._pkf = (w1, b1, ops.pack_ff2_fused(ff2.weight), None if ff2.bias is None else ff2.bias.detach().float().contiguous())
            w1, b1, w2img, b2 = self._pkf
            g, be = norm._affine()
            if tail is not None:
                w3, b3, res3, cs = tail
                return ops.ff_fused(x, g, be, norm.eps, w1, b1, w2img, b2, residual=residual, w3=w3, b3=b3, residual3=res3, colstats=cs)
            return ops.ff_fused(x, g, be, norm.eps, w1, b1, w2img, b2, residual=residual)
        assert tail is None, "FeedForward.rows: the projection tail goes with the fused launch only"
        if not self.glu:
            if norm is not None:
                x = norm.rows(x)
            return self.net[2].rows(self.net[0][0].rows(x, epilogue=ops.EPI_GELU), residual=residual)
        return self.net[2].rows(self.net[0].rows(x, norm=norm, rowstats=rowstats), residual=residual)

    def forward(self, x):
        shp = x.shape
        return self.rows(x.reshape(-1, shp[-1]).to(BF16).contiguous()).reshape(shp).to(x.dtype)


class CrossAttention(nn.Module):
    """attention.py:145-194.  q/k/v projections are fused into one (self) or two (cross) GEMMs, the softmax(QK^T)V core
    is the flash-style HIP kernel reading heads in place (no 'b n (h d) -> (b h) n d' copies), to_out fuses +bias."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.):
        super().__init__()
        inner_dim = dim_head * heads
        self.is_self = context_dim is None
        context_dim = default(context_dim, query_dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.to_q = Linear(query_dim, inner_dim, bias=False)
        self.to_k = Linear(context_dim, inner_dim, bias=False)
        self.to_v = Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self._pk = None

    def repack(self):
        self._pk = None
        self._pkx = None
        self._pkln_q = self._pkln_qkv = None

    def _packed_ln(self, wname, norm):
        """(W', s, c) of the query-side projection `wname` ('qkv' | 'q') with `norm` folded in (ops.pack_ln_fold)."""
        attr = "_pkln_" + wname
        mods = (self.to_q, self.to_k, self.to_v) if wname == "qkv" else (self.to_q,)
        if ops.cache_stale(self, attr, *(m.weight for m in mods), norm.weight, norm.bias):
            w = torch.cat([m.weight.detach().float() for m in mods], 0)
            setattr(self, attr, ops.pack_ln_fold(w, None, norm.weight, norm.bias))
        return getattr(self, attr)

    def _packed(self):
        if ops.cache_stale(self, "_pk", self.to_q.weight, self.to_k.weight, self.to_v.weight):
            wq, wk, wv = (ops.pack_linear(m.weight) for m in (self.to_q, self.to_k, self.to_v))
            pk = {"dev": self.to_q.weight.device, "q": wq, "kv": torch.cat([wk, wv], 0).contiguous()}
            if wq.shape[1] == wk.shape[1]:
                pk["qkv"] = torch.cat([wq, wk, wv], 0).contiguous()
            self._pk = pk
        return self._pk

    def _packed_x(self):
        """(Wq images, Wo images, to_out bias) of the fused cross-attention launch (`ops.xattn_fused`), once per weight version."""
        if ops.cache_stale(self, "_pkx", self.to_q.weight, self.to_out[0].weight, self.to_out[0].bias):
            bo = self.to_out[0].bias
            self._pkx = (ops.pack_xattn_wq(self.to_q.weight), ops.pack_xattn_wo(self.to_out[0].weight), None if bo is None else bo.detach().float().contiguous())
        return self._pkx

    def fused_kv_images(self, kv, adapter, B):
        """The K | V images of `ops.xattn_fused` for this layer's (step-invariant) context, or None where the fused launch does not apply to the layer at all
        (it needs C = 320 as 8 heads of 40, 65 .. 80 text keys, at most 16 expert keys).  Built once per edit (AnySD: MoE.prepare_conditioning)."""
        if not _XATTN_FUSED or self.is_self or self.heads != 8 or self.dim_head != 40 or self.to_q.weight.shape[1] != 320:
            return None
        Nk = kv.shape[0] // B
        T = 0 if adapter is None else adapter[0].shape[0] // B
        if not (64 < Nk <= 80 and T <= 16):
            return None
        return ops.pack_xattn_kv(kv, None if adapter is None else adapter[0], B, Nk, T)

    def project_kv(self, ctx_rows):
        """K|V of a context [B*Nk, Dc] -> [B*Nk, 2*inner] (step-invariant for text conditioning: cache it)."""
        return ops.gemm(ctx_rows, self._packed()["kv"])

    def rows(self, x, B, N, context_rows=None, Nk=None, kv=None, key_mask=None, residual=None, adapter=None, norm=None, rowstats=None,
             out_rowstats=None, kv_img=None):
        """x: [B*N, C] bf16 rows.  context_rows: [B*Nk, Dc] or None (self-attention).  Returns to_out(attn) (+residual).
        rowstats: row statistics of x from the GEMM that produced it (`ops.rowstats_buffer`): `norm` is then folded into the query-side
        projection; out_rowstats: buffer that receives the statistics of the result (to_out's epilogue) for the next block norm.
        adapter: optional (kv_ip [B*T, 2*inner] bf16, gate [B] fp32): decoupled expert attention added to the output,
        out = Attn(q,K,V) + gate_b * Attn(q,K_ip,V_ip)  (AnySD row A9, shape template ip_adapter/attention_processor.py:141-173)."""
        h, d = self.heads, self.dim_head
        inner = h * d
        if kv_img is not None and kv is not None and norm is not None and residual is x and key_mask is None and out_rowstats is None:
            # round 6: norm2 -> to_q -> softmax(QK^T)V (+ the gated expert segment) -> to_out + residual as ONE launch where the kernel covers the shape
            Nk_ = kv.shape[0] // B
            T_ = 0 if adapter is None else adapter[0].shape[0] // B
            if ops.xattn_fused_ok(x.shape[0], x.shape[1], h, d, N, Nk_, T_):
                wq_img, wo_img, bo = self._packed_x()
                g_, be_ = norm._affine()
                return ops.xattn_fused(x, g_, be_, norm.eps, wq_img, kv_img, None if adapter is None else adapter[1], wo_img, bo, N, Nk_, T_, self.scale)
        pk = self._packed()

        def proj(wname):  # norm: the block's LayerNorm fused in front of the query-side projection (x = UN-normalised rows)
            if norm is None:
                return ops.gemm(x, pk[wname])
            if rowstats is not None and ops.ln_fold_plan(x.shape[0], pk[wname].shape[0], x.shape[1], ops.EPI_NONE, 2):
                wq, s, c = self._packed_ln(wname, norm)
                return ops.gemm_ln(x, rowstats, wq, s, c, norm.eps)
            g, be = norm._affine()
            return ops.ln_gemm(x, g, be, norm.eps, pk[wname])

        if context_rows is None and kv is None:
            qkv = proj("qkv")  # [B*N, 3*inner]
            s = (N * 3 * inner, d, 3 * inner)
            o = ops.attention(qkv, qkv[:, inner:], qkv[:, 2 * inner:], B, h, N, N, d, self.scale, s, s, s, key_mask=key_mask)
        else:
            q = proj("q")
            if kv is None:
                kv = self.project_kv(context_rows)
            Nk = kv.shape[0] // B
            qs = (N * inner, d, inner)
            ks = (Nk * 2 * inner, d, 2 * inner)
            if adapter is not None and (d <= 96 or (d == 160 and _SEG2_160)) and key_mask is None:
                # both softmaxes in ONE launch: Q read once, O written once (the 4-token expert segment used to cost as much
                # as the 77-token text segment because it re-read Q and read-modify-wrote O)
                kv_ip, gate = adapter
                T_ip = kv_ip.shape[0] // B
                ks_ip = (T_ip * 2 * inner, d, 2 * inner)
                seg2 = (kv_ip, kv_ip[:, inner:], T_ip, ks_ip, ks_ip, gate)
                o = ops.attention(q, kv, kv[:, inner:], B, h, N, Nk, d, self.scale, qs, ks, ks, seg2=seg2)
            else:
                o = ops.attention(q, kv, kv[:, inner:], B, h, N, Nk, d, self.scale, qs, ks, ks, key_mask=key_mask)
                if adapter is not None:
                    kv_ip, gate = adapter
                    T_ip = kv_ip.shape[0] // B
                    ks_ip = (T_ip * 2 * inner, d, 2 * inner)
                    ops.attention(q, kv_ip, kv_ip[:, inner:], B, h, N, T_ip, d, self.scale, qs, ks_ip, ks_ip, out=o,
                                  out_scale=gate, accumulate=True)
        return self.to_out[0].rows(o.reshape(B * N, inner), residual=residual, rowstats=out_rowstats)

    def forward(self, x, context=None, mask=None):
        B, N, C = x.shape
        xr = x.reshape(B * N, C).to(BF16).contiguous()
        ctx = None
        if context is not None:
            ctx = context.reshape(-1, context.shape[-1]).to(BF16).contiguous()
        km = None
        if exists(mask):
            km = mask.reshape(B, -1).to(torch.uint8).contiguous()
        if ctx is None and not self.is_self:
            raise ValueError("CrossAttention built with context_dim needs a context of that width")
        if ctx is not None and self.is_self:
            # reference semantics: context_dim defaults to query_dim, any context of that width is accepted
            y = self.rows(xr, B, N, context_rows=ctx, key_mask=km)
        else:
            y = self.rows(xr, B, N, context_rows=ctx, key_mask=km)
        return y.reshape(B, N, -1).to(x.dtype)


class MemoryEfficientCrossAttention(CrossAttention):
    """attention.py:197-243 — same operator; the xformers call site (:233) is `ops.attention_bhnd`."""


class BasicTransformerBlock(nn.Module):
    """attention.py:246-275."""
    ATTENTION_MODES = {"softmax": CrossAttention, "softmax-xformers": MemoryEfficientCrossAttention,
                       "softmax-hip": CrossAttention}

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        attn_cls = self.ATTENTION_MODES["softmax-hip"]
        self.disable_self_attn = disable_self_attn
        self.attn1 = attn_cls(query_dim=dim, heads=n_heads, dim_head=d_head, dropout=dropout,
                              context_dim=context_dim if self.disable_self_attn else None)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = attn_cls(query_dim=dim, context_dim=context_dim, heads=n_heads, dim_head=d_head, dropout=dropout)
        self.norm1 = LayerNorm(dim)
        self.norm2 = LayerNorm(dim)
        self.norm3 = LayerNorm(dim)
        self.checkpoint = checkpoint

    def rows(self, x, B, N, context_rows=None, kv_cache=None, rowstats=None, tail=None):
        """x: [B*N, C] bf16.  kv_cache: optional dict id(attn)->projected K|V of the (step-invariant) context.
        rowstats: row statistics of x from the GEMM that produced it (see `wants_rowstats`).
        tail: the projection behind the block (see FeedForward.rows), only where `takes_tail(M, C)`."""
        tape = ops._TAPE
        if self.checkpoint and tape is not None and tape.active:  # attention.py:268: recompute this block in the backward pass
            return tape.checkpoint(lambda: self._rows(x, B, N, context_rows, kv_cache))
        return self._rows(x, B, N, context_rows, kv_cache, rowstats, tail)

    def takes_tail(self, M, C):
        """True where the block's feed-forward is the fused launch, which can carry the projection that follows the block."""
        return _FF_TAIL and self.ff.fused_ok(M, C)

    def _fold_plan(self, M, C):
        """Which of the three norms can be folded into its projection at M rows: (norm1, norm2, norm3).  A norm folds when the GEMM behind it
        carries the fold epilogue; norm2 / norm3 also need the to_out GEMM in front of them to emit the row statistics."""
        key = (M, C, ops._LN_FOLD, ops._TAPE is not None and ops._TAPE.active)
        if getattr(self, "_fold_key", None) != key:
            inner1 = self.attn1.heads * self.attn1.dim_head
            inner2 = self.attn2.heads * self.attn2.dim_head
            n1 = inner1 if (self.disable_self_attn and not self.attn1.is_self) else 3 * inner1
            f1 = ops.ln_fold_plan(M, n1, C, ops.EPI_NONE, 2)
            f2 = ops.ln_fold_plan(M, inner2, C, ops.EPI_NONE, 2) and ops.ln_fold_plan(M, C, inner1, ops.EPI_NONE, 1)
            # (where the feed-forward runs as one fused launch it normalises its rows itself: attn2's to_out need not emit statistics)
            f3 = self.ff.glu and not self.ff.fused_ok(M, C) and ops.ln_fold_plan(M, self.ff.net[0].proj.weight.shape[0], C, ops.EPI_GEGLU, 2) \
                and ops.ln_fold_plan(M, C, inner2, ops.EPI_NONE, 1)
            self._fold_key, self._fold = key, (f1, f2, f3)
        return self._fold

    def wants_rowstats(self, M, C):
        """True when norm1 would be folded into attn1's projection given the row statistics of the block's input."""
        return self._fold_plan(M, C)[0]

    def _rows(self, x, B, N, context_rows=None, kv_cache=None, rowstats=None, tail=None):
        c1 = context_rows if self.disable_self_attn else None
        M, C = x.shape
        f1, f2, f3 = self._fold_plan(M, C)
        kv2, adapter, kv_img = None, None, None
        if kv_cache is not None and context_rows is not None and not (ops._TAPE is not None and ops._TAPE.active):
            kv_img = kv_cache.get(("xattn_img", id(self.attn2)))     # the fused cross-attention launch's K | V images (MoE.prepare_conditioning, or built below)
        # where the cross-attention half runs as the fused launch (it normalises its rows itself) attn1's to_out need not emit row statistics
        use_x = kv_img is not None and not f3 and ops.xattn_fused_ok(M, C, self.attn2.heads, self.attn2.dim_head, N, 78, 0)
        st2 = ops.rowstats_buffer(M, C, x.device) if (f2 and not use_x) else None
        st3 = ops.rowstats_buffer(M, C, x.device) if f3 else None
        x = self.attn1.rows(x, B, N, context_rows=c1, residual=x, norm=self.norm1, rowstats=rowstats if f1 else None, out_rowstats=st2)
        if kv_cache is not None and context_rows is not None:
            key = id(self.attn2)
            kv2 = kv_cache.get(key)
            if kv2 is None:
                kv2 = self.attn2.project_kv(context_rows)
                if not (ops._TAPE is not None and ops._TAPE.active):
                    # Cached for the following DDIM steps.  NOT while a training tape records: this body may be a checkpoint segment
                    # that runs twice (throw-away forward, recompute in the backward), and a tensor created on the throw-away tape
                    # must not be what the recompute finds (its gradient would be dropped silently — ADVICE r2).
                    kv_cache[key] = kv2
            adapter = kv_cache.get(("adapter", key))  # installed by anysd.MoE.prepare_conditioning
            if callable(adapter):  # training: the expert K|V projection is recorded here, next to the attention that consumes it
                adapter = adapter()
            if kv_img is None and not f3 and not (ops._TAPE is not None and ops._TAPE.active):
                # plain (non-AnySD) callers: the images are built on the first evaluation with this cache — outside a graph capture, whose warm-up runs come
                # first — and used from the next one on (layers the fused launch does not apply to answer None at once and store nothing)
                img = self.attn2.fused_kv_images(kv2, adapter, B)
                if img is not None:
                    kv_cache[("xattn_img", key)] = img
        x = self.attn2.rows(x, B, N, context_rows=context_rows, kv=kv2, residual=x, adapter=adapter, norm=self.norm2, rowstats=st2, out_rowstats=st3,
                            kv_img=kv_img if use_x else None)
        x = self.ff.rows(x, residual=x, norm=self.norm3, rowstats=st3, tail=tail)
        return x

    def forward(self, x, context=None):
        B, N, C = x.shape
        ctx = None if context is None else context.reshape(-1, context.shape[-1]).to(BF16).contiguous()
        y = self.rows(x.reshape(B * N, C).to(BF16).contiguous(), B, N, context_rows=ctx)
        return y.reshape(B, N, C).to(x.dtype)

    _forward = forward


class SpatialTransformer(nn.Module):
    """attention.py:278-340.  On channels-last rows the two 'b c h w <-> b (h w) c' rearranges vanish and the 1x1 convs
    are plain GEMMs; proj_out fuses the residual add."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=True):
        super().__init__()
        if exists(context_dim) and not isinstance(context_dim, list):
            context_dim = [context_dim] * depth
        if context_dim is None:
            context_dim = [None] * depth
        self.in_channels = in_channels
        inner_dim = n_heads * d_head
        self.norm = Normalize(in_channels)
        if not use_linear:
            self.proj_in = Conv2d(in_channels, inner_dim, kernel_size=1, stride=1, padding=0)
        else:
            self.proj_in = Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner_dim, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                   disable_self_attn=disable_self_attn, checkpoint=use_checkpoint) for d in range(depth)])
        if not use_linear:
            self.proj_out = zero_module(Conv2d(inner_dim, in_channels, kernel_size=1, stride=1, padding=0))
        else:
            self.proj_out = zero_module(Linear(in_channels, inner_dim))
        self.use_linear = use_linear

    def rows(self, x, B, H, W, context_rows=None, kv_cache=None, colstats=None, out_colstats=None):
        """colstats: per-channel slab statistics of x from its producer (the norm then skips its statistics pass); out_colstats: buffer
        that receives the statistics of the result (proj_out's epilogue), for the GroupNorm of the block that follows."""
        if isinstance(context_rows, (list, tuple)):
            ctxs = list(context_rows)
        else:
            ctxs = [context_rows] * len(self.transformer_blocks)
        N = H * W
        h = self.norm.rows(x, B, N, colstats=colstats)
        pin = self.proj_in._packed()
        # the first block's norm1 rides in its qkv projection where proj_in can emit the row statistics it needs (attention.py:271)
        M, Ci = h.shape[0], pin["w"].shape[0]
        st = None
        if self.transformer_blocks[0].wants_rowstats(M, Ci) and ops.ln_fold_plan(M, Ci, h.shape[1], ops.EPI_NONE, 1):
            st = ops.rowstats_buffer(M, Ci, h.device)
        h = ops.gemm(h, pin["w"], pin["b"], rowstats=st)
        pout = self.proj_out._packed()
        last = len(self.transformer_blocks) - 1
        # round 6: where the last block's feed-forward runs as the fused launch, proj_out + the residual (attention.py:337-340) ride in it
        tail_ok = self.transformer_blocks[last].takes_tail(M, Ci) and pout["w"].shape == (Ci, Ci) and x.shape[1] == Ci
        for i, blk in enumerate(self.transformer_blocks):
            tail = (pout["w"], pout["b"], x, out_colstats) if (tail_ok and i == last) else None
            h = blk.rows(h, B, N, context_rows=ctxs[i], kv_cache=kv_cache, rowstats=st if i == 0 else None, tail=tail)
        if tail_ok:
            return h
        return ops.gemm(h, pout["w"], pout["b"], residual=x, colstats=out_colstats)

    def forward(self, x, context=None):
        B, C, H, W = x.shape
        if isinstance(context, list):
            ctx = [None if c is None else c.reshape(-1, c.shape[-1]).to(BF16).contiguous() for c in context]
        else:
            ctx = None if context is None else context.reshape(-1, context.shape[-1]).to(BF16).contiguous()
        y = self.rows(ops.nchw_to_rows(x), B, H, W, context_rows=ctx)
        return ops.rows_to_nchw(y, B, H, W, out_dtype=x.dtype)
