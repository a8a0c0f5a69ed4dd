"""Mirror of ldm/modules/diffusionmodules/openaimodel.py (UNetModel and its blocks) on HIP kernels.

Constructor arguments, sub-module names and state-dict keys follow the reference (openaimodel.py:412-730; key schema in
SURVEY.md Appendix A), including the parameter CREATION ORDER, so `torch.manual_seed(s); UNetModel(**cfg)` yields the
same initial weights as the reference constructor.  Forward data flow (openaimodel.py:754-786) runs on channels-last bf16
rows end to end; NCHW fp32 exists only at the model boundary.
"""
from abc import abstractmethod
import math

import torch
import torch.nn as nn

from anyedit_amd import ops
from anyedit_amd.ldm.util import exists
from anyedit_amd.ldm.modules.attention import SpatialTransformer
from anyedit_amd.ldm.modules.diffusionmodules.util import avg_pool_nd, conv_nd, linear, normalization, zero_module

BF16 = torch.bfloat16


class TimestepBlock(nn.Module):
    """openaimodel.py:60-69."""

    @abstractmethod
    def forward(self, x, emb):
        """Apply the module to `x` given `emb` timestep embeddings."""


class Feat:
    """Channels-last activation handle: rows [B*H*W, C] bf16 (+ optional second source = pending channel-concat).
    st / st2: per-channel slab statistics of t / t2 written by the kernels that produced them (`ops.colstats_buffer`), or None: the
    GroupNorm that consumes the tensor then needs no statistics pass (SURVEY.md §7 hard part (iii))."""
    __slots__ = ("t", "B", "H", "W", "t2", "st", "st2")

    def __init__(self, t, B, H, W, t2=None, st=None, st2=None):
        self.t, self.B, self.H, self.W, self.t2, self.st, self.st2 = t, B, H, W, t2, st, st2

    def materialize(self):
        if self.t2 is not None:
            self.t = ops.concat_channels(self.t, self.t2)
            self.t2 = None
            self.st = self.st2 = None
        return self.t


def _stats_for(B, H, W, C, device):
    """A statistics buffer for a [B*H*W, C] output whose consumer is a GroupNorm taking producer statistics, else None."""
    return ops.colstats_buffer(B * H * W, C, device) if ops.want_colstats(H * W) else None


class EmbPack:
    """SiLU(emb) plus the emb_layers projections of ALL ResBlocks computed by one batched GEMM (22 M=12 launches -> 1)."""
    __slots__ = ("silu", "all", "offsets")

    def __init__(self, silu, all_out=None, offsets=None):
        self.silu, self.all, self.offsets = silu, all_out, offsets or {}


class TimestepEmbedSequential(nn.Sequential, TimestepBlock):
    """openaimodel.py:73-87."""

    def rows(self, f, emb_silu, context_rows=None, kv_cache=None):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                f = layer.rows(f, emb_silu)
            elif isinstance(layer, SpatialTransformer):
                x_st = f.st if f.t2 is None else None
                x = f.materialize()
                st = _stats_for(f.B, f.H, f.W, x.shape[1], x.device)
                f = Feat(layer.rows(x, f.B, f.H, f.W, context_rows=context_rows, kv_cache=kv_cache, colstats=x_st, out_colstats=st),
                         f.B, f.H, f.W, st=st)
            elif isinstance(layer, AttentionBlock):
                x_st = f.st if f.t2 is None else None
                f = Feat(layer.rows(f.materialize(), f.B, f.H * f.W, colstats=x_st), f.B, f.H, f.W)
            else:
                f = layer.rows(f)
        return f

    def forward(self, x, emb, context=None):
        for layer in self:
            if isinstance(layer, TimestepBlock):
                x = layer(x, emb)
            elif isinstance(layer, SpatialTransformer):
                x = layer(x, context)
            else:
                x = layer(x)
        return x


class Upsample(nn.Module):
    """openaimodel.py:90-118: nearest x2 folded into the conv's gather (no 4x intermediate tensor); use_conv=False (conv_resample=False, and the
    h_upd / x_upd of ResBlock(up=True)) is the nearest x2 alone."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if dims != 2:
            raise NotImplementedError("Upsample: only dims=2 (the 1-D / 3-D forms of openaimodel.py:108-112 have no caller in AnyEdit)")
        if use_conv:
            self.conv = conv_nd(dims, self.channels, self.out_channels, 3, padding=padding)

    def rows(self, f):
        if not self.use_conv:
            y, Ho, Wo = ops.resample2x_rows(f.materialize(), f.B, f.H, f.W)
            return Feat(y, f.B, Ho, Wo)
        Ho, Wo = self.conv.out_hw(f.H, f.W, upsample2x=True)
        st = _stats_for(f.B, Ho, Wo, self.out_channels, f.t.device)
        y, Ho, Wo = self.conv.rows(f.materialize(), f.B, f.H, f.W, upsample2x=True, colstats=st)
        return Feat(y, f.B, Ho, Wo, st=st)

    def forward(self, x):
        assert x.shape[1] == self.channels
        B, C, H, W = x.shape
        f = self.rows(Feat(ops.nchw_to_rows(x), B, H, W))
        return ops.rows_to_nchw(f.t, B, f.H, f.W, out_dtype=x.dtype)


class Downsample(nn.Module):
    """openaimodel.py:131-159: conv3x3 stride 2 pad 1, or with use_conv=False the 2x2 mean of avg_pool_nd (:152-155)."""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.dims = dims
        if dims != 2:
            raise NotImplementedError("Downsample: only dims=2 (the 1-D / 3-D forms of openaimodel.py:146 have no caller in AnyEdit)")
        if use_conv:
            self.op = conv_nd(dims, self.channels, self.out_channels, 3, stride=2, padding=padding)
        else:
            assert self.channels == self.out_channels
            self.op = avg_pool_nd(dims, kernel_size=2, stride=2)

    def rows(self, f):
        if not self.use_conv:
            y, Ho, Wo = self.op.rows(f.materialize(), f.B, f.H, f.W)
            return Feat(y, f.B, Ho, Wo)
        Ho, Wo = self.op.out_hw(f.H, f.W)
        st = _stats_for(f.B, Ho, Wo, self.out_channels, f.t.device)
        y, Ho, Wo = self.op.rows(f.materialize(), f.B, f.H, f.W, colstats=st)
        return Feat(y, f.B, Ho, Wo, st=st)

    def forward(self, x):
        assert x.shape[1] == self.channels
        B, C, H, W = x.shape
        f = self.rows(Feat(ops.nchw_to_rows(x), B, H, W))
        return ops.rows_to_nchw(f.t, B, f.H, f.W, out_dtype=x.dtype)


class ResBlock(TimestepBlock):
    """openaimodel.py:162-274.

    HIP data flow: GN+SiLU kernel -> conv3x3 (+bias +time-embedding vector in the epilogue) -> GN+SiLU -> conv3x3
    (+bias +skip residual in the epilogue).  The skip is the input itself or a 1x1-conv GEMM; a pending channel-concat
    input (decoder) is consumed in place by the GN and the skip GEMM (two-source kernels), never materialised.
    up / down (resblock_updown, :215-221, 254-260): the nearest x2 of h rides in conv1's gather, the 2x2 mean and the resampling of x are
    `ops.resample2x_rows`; use_scale_shift_norm (:264-268): GroupNorm, then `ops.scale_shift_rows` applies (1 + scale), shift and the SiLU.
    Neither is on the SD-1.5 / AnySD configuration (forward only: no tape closures)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        self.channels = channels
        self.emb_channels = emb_channels
        self.dropout = dropout
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.use_checkpoint = use_checkpoint
        self.use_scale_shift_norm = use_scale_shift_norm
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), conv_nd(dims, channels, self.out_channels, 3, padding=1))
        self.updown = up or down
        self.up, self.down = bool(up), bool(down) and not up
        if up:
            self.h_upd, self.x_upd = Upsample(channels, False, dims), Upsample(channels, False, dims)
        elif down:
            self.h_upd, self.x_upd = Downsample(channels, False, dims), Downsample(channels, False, dims)
        else:
            self.h_upd = self.x_upd = nn.Identity()
        self.emb_layers = nn.Sequential(nn.SiLU(), linear(emb_channels, 2 * self.out_channels if use_scale_shift_norm else self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        zero_module(conv_nd(dims, self.out_channels, self.out_channels, 3, padding=1)))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        elif use_conv:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 3, padding=1)
        else:
            self.skip_connection = conv_nd(dims, channels, self.out_channels, 1)

    def rows(self, f, emb_silu):
        """f: Feat (possibly with a pending concat); emb_silu: bf16 [B, emb_channels] = SiLU(emb)."""
        tape = ops._TAPE
        if self.use_checkpoint and tape is not None and tape.active:  # openaimodel.py:250: recompute this block in the backward pass
            return Feat(tape.checkpoint(lambda: self._rows(f, emb_silu).t), f.B, f.H, f.W)
        return self._rows(f, emb_silu)

    def _rows(self, f, emb_silu):
        B, H, W = f.B, f.H, f.W
        if isinstance(emb_silu, EmbPack) and id(self) in emb_silu.offsets:
            off = emb_silu.offsets[id(self)]
            emb_out = emb_silu.all[:, off:off + self.emb_layers[1].out_features]  # column slice of the batched projection (fp32, strided rows)
        else:
            silu = emb_silu.silu if isinstance(emb_silu, EmbPack) else emb_silu
            emb_out = self.emb_layers[1].rows(silu, out_f32=True)  # [B, Cout] (scale-shift: [B, 2 Cout]) fp32
        dev = f.t.device
        ssn = self.use_scale_shift_norm
        h = self.in_layers[0].rows(f.t, B, H * W, silu=True, x2=f.t2, colstats=f.st, colstats2=f.st2)
        # This body may run twice as a checkpoint segment (throw-away forward + recompute): it must not change `f` — a concat
        # materialised into f.t on the throw-away tape would be unknown to the recompute's tape and its gradient dropped.
        whole = (lambda: f.t if f.t2 is None else ops.concat_channels(f.t, f.t2))
        Ho, Wo, xs = H, W, None
        if self.up:        # openaimodel.py:255-260: h = in_conv(h_upd(in_rest(x))), x = x_upd(x); the x2 of h is conv1's gather
            Ho, Wo = 2 * H, 2 * W
            xs, _, _ = ops.resample2x_rows(whole(), B, H, W)
        elif self.down:
            h, Ho, Wo = ops.resample2x_rows(h, B, H, W, down=True)
            xs, _, _ = ops.resample2x_rows(whole(), B, H, W, down=True)
        conv1, norm2 = self.in_layers[2], self.out_layers[0]
        if not self.up and not ssn and ops.conv3x3_gn_splitk_ok(B, Ho, Wo, conv1.in_channels, self.out_channels, norm2.num_groups):
            # 16x16 / 8x8 levels: conv1's plan cuts K, and `h + emb_out` (:272) exists only as the input of out_norm — the GroupNorm folds the K ranges (+ bias +
            # time-embedding vector) itself instead of reading what a reduce launch wrote (bit-identical; one launch and one round trip of h less)
            g2, b2 = norm2._affine()
            h = ops.groupnorm_splitk(conv1.rows_partials(h, B, Ho, Wo), conv1.packed_bias(), emb_out, g2, b2, B, Ho * Wo, norm2.eps, silu=True, groups=norm2.num_groups)
        else:
            st1 = _stats_for(B, Ho, Wo, self.out_channels, dev)   # conv1's epilogue delivers the statistics out_layers[0] needs
            if self.up:
                h, _, _ = conv1.rows(h, B, H, W, addvec=None if ssn else emb_out, upsample2x=True, colstats=st1)
            else:
                h, _, _ = conv1.rows(h, B, Ho, Wo, addvec=None if ssn else emb_out, colstats=st1)
            if ssn:        # :264-268: out_norm(h) * (1 + scale) + shift, then out_rest (SiLU, dropout, conv)
                h = ops.scale_shift_rows(norm2.rows(h, B, Ho * Wo, silu=False, colstats=st1), emb_out, B, Ho * Wo, silu=True)
            else:
                h = norm2.rows(h, B, Ho * Wo, silu=True, colstats=st1)
        if isinstance(self.skip_connection, nn.Identity):
            res = whole() if xs is None else xs
        elif self.skip_connection.kernel_size[0] == 1 and xs is None:
            res, _, _ = self.skip_connection.rows(f.t, B, H, W, a2=f.t2)
        else:
            res, _, _ = self.skip_connection.rows(whole() if xs is None else xs, B, Ho, Wo)
        st = _stats_for(B, Ho, Wo, self.out_channels, dev)      # ... and conv2's those of the block's output (next norm / decoder concat)
        y, _, _ = self.out_layers[3].rows(h, B, Ho, Wo, residual=res, colstats=st)
        return Feat(y, B, Ho, Wo, st=st)

    def forward(self, x, emb):
        B, C, H, W = x.shape
        f = self.rows(Feat(ops.nchw_to_rows(x), B, H, W), ops.silu_to_bf16(emb))
        return ops.rows_to_nchw(f.t, B, f.H, f.W, out_dtype=x.dtype)

    _forward = forward


class _QKVAttentionBase(nn.Module):
    """openaimodel.py:344-409: softmax(q k^T / sqrt(ch)) v per head over the packed [N, 3 H ch, T] projection (the reference scales q and k by ch^-1/4 each).
    `rows` takes the projection as channels-last rows [N*T, 3 H ch]: a head's q / k / v are column slices, handed to the attention kernel as strides."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads

    def _offsets(self, ch):
        raise NotImplementedError

    def rows(self, qkv, B, T):
        width = qkv.shape[1]
        assert width % (3 * self.n_heads) == 0
        ch = width // (3 * self.n_heads)
        (oq, ok, ov), hs = self._offsets(ch)
        st = (T * width, hs, width)                                  # (batch, head, row) element strides, the same for q, k and v
        return ops.attention(qkv[:, oq:], qkv[:, ok:], qkv[:, ov:], B, self.n_heads, T, T, ch, 1.0 / math.sqrt(ch), st, st, st).reshape(B * T, self.n_heads * ch)

    def forward(self, qkv):
        bs, width, length = qkv.shape
        y = self.rows(ops.nchw_to_rows(qkv.reshape(bs, width, length, 1)), bs, length)
        return ops.rows_to_nchw(y, bs, length, 1, out_dtype=qkv.dtype).reshape(bs, -1, length)


class QKVAttentionLegacy(_QKVAttentionBase):
    """openaimodel.py:344-373: heads split first — channels are (head, {q, k, v}, ch)."""

    def _offsets(self, ch):
        return (0, ch, 2 * ch), 3 * ch


class QKVAttention(_QKVAttentionBase):
    """openaimodel.py:376-409 (use_new_attention_order): q | k | v split first — channels are ({q, k, v}, head, ch)."""

    def _offsets(self, ch):
        return (0, self.n_heads * ch, 2 * self.n_heads * ch), ch


class AttentionBlock(nn.Module):
    """openaimodel.py:277-324: GroupNorm -> pointwise qkv -> per-head attention over all positions -> zero-init pointwise proj_out + residual: the UNet's
    attention layer when use_spatial_transformer=False (no AnyEdit configuration; forward only).  State-dict keys and shapes are the reference's
    (Conv1d weights [Cout, Cin, 1])."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        if num_head_channels == -1:
            self.num_heads = num_heads
        else:
            assert channels % num_head_channels == 0, f"q,k,v channels {channels} is not divisible by num_head_channels {num_head_channels}"
            self.num_heads = channels // num_head_channels
        self.use_checkpoint = use_checkpoint
        self.norm = normalization(channels)
        self.qkv = conv_nd(1, channels, channels * 3, 1)
        self.attention = QKVAttention(self.num_heads) if use_new_attention_order else QKVAttentionLegacy(self.num_heads)
        self.proj_out = zero_module(conv_nd(1, channels, channels, 1))

    def rows(self, x, B, T, colstats=None):
        h = self.norm.rows(x, B, T, silu=False, colstats=colstats)
        return self.proj_out.rows(self.attention.rows(self.qkv.rows(h), B, T), residual=x)

    def forward(self, x):
        b, c = x.shape[:2]
        T = int(x.numel() // (b * c))
        y = self.rows(ops.nchw_to_rows(x.reshape(b, c, T, 1)), b, T)
        return ops.rows_to_nchw(y, b, T, 1, out_dtype=x.dtype).reshape(x.shape)

    _forward = forward


class UNetModel(nn.Module):
    """openaimodel.py:412-786: use_spatial_transformer=True is the AnySD / SD-1.5 / AnyDoor configuration; False builds the guided-diffusion UNet with
    AttentionBlock layers (:277-324, forward only)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True, disable_self_attentions=None, num_attention_blocks=None,
                 disable_middle_self_attn=False, use_linear_in_transformer=False):
        super().__init__()
        if use_spatial_transformer:
            assert context_dim is not None, "use_spatial_transformer needs context_dim (openaimodel.py:474-475)"
        if context_dim is not None:
            assert use_spatial_transformer, "a cross-attention context_dim needs use_spatial_transformer (openaimodel.py:477-478)"
            if not isinstance(context_dim, int):
                context_dim = list(context_dim)
        if dims != 2:
            raise NotImplementedError("dims != 2: the 1-D / 3-D UNets of openaimodel.py have no caller in AnyEdit")
        if num_classes is not None and not isinstance(num_classes, int) and num_classes != "continuous":
            raise ValueError(f"num_classes must be None, an int or 'continuous' (openaimodel.py:533-540), got {num_classes!r}")
        if num_heads_upsample == -1:
            num_heads_upsample = num_heads
        if num_heads == -1:
            assert num_head_channels != -1, "Either num_heads or num_head_channels has to be set"
        if num_head_channels == -1:
            assert num_heads != -1, "Either num_heads or num_head_channels has to be set"
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        if isinstance(num_res_blocks, int):
            self.num_res_blocks = len(channel_mult) * [num_res_blocks]
        else:
            if len(num_res_blocks) != len(channel_mult):
                raise ValueError("provide num_res_blocks either as an int or as a per-level list")
            self.num_res_blocks = list(num_res_blocks)
        if disable_self_attentions is not None:
            assert len(disable_self_attentions) == len(channel_mult)
        if num_attention_blocks is not None:
            assert len(num_attention_blocks) == len(self.num_res_blocks)
        self.attention_resolutions = attention_resolutions
        self.dropout = dropout
        self.channel_mult = channel_mult
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads_upsample
        self.predict_codebook_ids = n_embed is not None       # openaimodel.py:524

        def heads_for(ch, nh):
            if num_head_channels == -1:
                return nh, ch // nh
            return ch // num_head_channels, num_head_channels

        def make_st(ch, nh, level, is_middle=False):
            if not use_spatial_transformer:
                # openaimodel.py:568-575, 583-588 (and :640-645, 680-699): dim_head follows num_head_channels; with neither it nor `legacy` set it is
                # ch // num_heads of the INPUT side's head count (the reference computes it from `num_heads` in the output blocks too)
                d_h = num_head_channels if (num_head_channels != -1 or legacy) else ch // num_heads
                return AttentionBlock(ch, use_checkpoint=use_checkpoint, num_heads=nh, num_head_channels=d_h, use_new_attention_order=use_new_attention_order)
            n_h, d_h = heads_for(ch, nh)
            if legacy:
                d_h = ch // n_h
            dsa = disable_middle_self_attn if is_middle else (disable_self_attentions[level] if exists(disable_self_attentions) else False)
            return SpatialTransformer(ch, n_h, d_h, depth=transformer_depth, context_dim=context_dim, disable_self_attn=dsa,
                                      use_linear=use_linear_in_transformer, use_checkpoint=use_checkpoint)

        time_embed_dim = model_channels * 4
        self.time_embed = nn.Sequential(linear(model_channels, time_embed_dim), nn.SiLU(), linear(time_embed_dim, time_embed_dim))
        if num_classes is not None:  # openaimodel.py:533-538 (created here: the reference's parameter order, seed-reproducible init)
            self.label_emb = nn.Embedding(num_classes, time_embed_dim) if isinstance(num_classes, int) else nn.Linear(1, time_embed_dim)
        self.input_blocks = nn.ModuleList([TimestepEmbedSequential(conv_nd(dims, in_channels, model_channels, 3, padding=1))])
        self._feature_size = model_channels
        input_block_chans = [model_channels]
        ch = model_channels
        ds = 1
        for level, mult in enumerate(channel_mult):
            for nr in range(self.num_res_blocks[level]):
                layers = [ResBlock(ch, time_embed_dim, dropout, out_channels=mult * model_channels, dims=dims,
                                   use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm)]
                ch = mult * model_channels
                if ds in attention_resolutions:
                    if not exists(num_attention_blocks) or nr < num_attention_blocks[level]:
                        layers.append(make_st(ch, num_heads, level))
                self.input_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch
                input_block_chans.append(ch)
            if level != len(channel_mult) - 1:
                out_ch = ch
                self.input_blocks.append(TimestepEmbedSequential(
                    ResBlock(ch, time_embed_dim, dropout, out_channels=out_ch, dims=dims, use_checkpoint=use_checkpoint,
                             use_scale_shift_norm=use_scale_shift_norm, down=True)
                    if resblock_updown else Downsample(ch, conv_resample, dims=dims, out_channels=out_ch)))
                ch = out_ch
                input_block_chans.append(ch)
                ds *= 2
                self._feature_size += ch
        self.middle_block = TimestepEmbedSequential(
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm),
            make_st(ch, num_heads, 0, is_middle=True),
            ResBlock(ch, time_embed_dim, dropout, dims=dims, use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm))
        self._feature_size += ch
        self.output_blocks = nn.ModuleList([])
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(self.num_res_blocks[level] + 1):
                ich = input_block_chans.pop()
                layers = [ResBlock(ch + ich, time_embed_dim, dropout, out_channels=model_channels * mult, dims=dims,
                                   use_checkpoint=use_checkpoint, use_scale_shift_norm=use_scale_shift_norm)]
                ch = model_channels * mult
                if ds in attention_resolutions:
                    if not exists(num_attention_blocks) or i < num_attention_blocks[level]:
                        layers.append(make_st(ch, num_heads_upsample, level))
                if level and i == self.num_res_blocks[level]:
                    out_ch = ch
                    layers.append(ResBlock(ch, time_embed_dim, dropout, out_channels=out_ch, dims=dims, use_checkpoint=use_checkpoint,
                                           use_scale_shift_norm=use_scale_shift_norm, up=True)
                                  if resblock_updown else Upsample(ch, conv_resample, dims=dims, out_channels=out_ch))
                    ds //= 2
                self.output_blocks.append(TimestepEmbedSequential(*layers))
                self._feature_size += ch
        self.out = nn.Sequential(normalization(ch), nn.SiLU(), zero_module(conv_nd(dims, model_channels, out_channels, 3, padding=1)))
        if self.predict_codebook_ids:   # openaimodel.py:731-736: GroupNorm + pointwise conv to n_embed logits per position (no SiLU; `out` stays in the state dict unused)
            self.id_predictor = nn.Sequential(normalization(ch), conv_nd(dims, model_channels, n_embed, 1))

    # ------------------------------------------------------------------------------------------------ forward
    def repack(self):
        """Drop cached packed weights (call after load_state_dict / optimizer steps on UNet parameters)."""
        self._emb_pk = None
        for m in self.modules():
            if m is not self and hasattr(m, "repack"):
                m.repack()

    def _emb_pack(self, emb_silu):
        """All ResBlock.emb_layers Linears as ONE GEMM: W = cat over blocks [sum(Cout), 4*mc]; each block reads its column slice."""
        blocks = self.__dict__.get("_emb_blocks")
        if blocks is None:
            blocks = self.__dict__["_emb_blocks"] = [m for m in self.modules() if isinstance(m, ResBlock)]
        pk = getattr(self, "_emb_pk", None)
        if ops.cache_stale(self, "_emb_pk", *[p for b in blocks for p in (b.emb_layers[1].weight, b.emb_layers[1].bias)]):
            w = torch.cat([ops.pack_linear(b.emb_layers[1].weight) for b in blocks], 0).contiguous()
            bias = torch.cat([b.emb_layers[1].bias.detach().float() for b in blocks], 0).contiguous()
            offsets, off = {}, 0
            for b in blocks:
                offsets[id(b)] = off
                off += b.emb_layers[1].out_features
            pk = (w, bias, offsets)
            self._emb_pk = pk
        w, bias, offsets = pk
        return EmbPack(emb_silu, ops.gemm(emb_silu, w, bias, out_f32=True), offsets)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.repack()
        return r

    def context_rows(self, context):
        """[B, L, Dc] (any float dtype) -> bf16 rows [B*L, Dc] (once per prompt; the context is step-invariant)."""
        if isinstance(context, (list, tuple)):
            return [self.context_rows(c) for c in context]
        return context.reshape(-1, context.shape[-1]).to(BF16).contiguous()

    def _label_rows(self, y, device):
        """label_emb(y) as fp32 rows [B, 4*mc]: a table lookup for integer classes (indexing), and for num_classes == "continuous" the
        Linear(1, 4*mc) of openaimodel.py:536-538 through the exact-fp32 MFMA GEMM with K zero-padded from 1 to its 16-element step."""
        if isinstance(self.num_classes, int):
            return self.label_emb.weight.detach().float()[y.reshape(-1).long().to(device)].contiguous()
        w, b = self.label_emb.weight.detach().float(), self.label_emb.bias.detach().float().contiguous()
        yp = torch.zeros(y.shape[0], 16, dtype=torch.float32, device=device)
        yp[:, 0] = y.reshape(-1).to(device=device, dtype=torch.float32)
        wp = torch.zeros(w.shape[0], 16, dtype=torch.float32, device=device)
        wp[:, 0] = w[:, 0]
        return ops.linear_f32(yp, wp, b)

    def time_embedding_rows(self, timesteps):
        """The whole time-embedding chain of openaimodel.py:762-763 + every ResBlock's emb_layers (:262-263) for a VECTOR of time steps, one row each:
        timestep_embedding -> Linear + SiLU -> Linear -> SiLU -> the batched emb_layers projection.  It depends on t alone (not on x, the context or the
        sample), so a sampler that knows its schedule computes it ONCE for all steps (EditPipeline.edit: one M = 50 pass instead of fifty M = 3B ones)
        and hands row i, broadcast over the batch, to `forward_rows(..., emb_pack=)`.  Returns an EmbPack whose `.all` is fp32 [len(timesteps), sum Cout]."""
        assert self.num_classes is None, "class-conditional models add label_emb(y) per sample: use forward_rows(y=...)"
        t_emb = ops.timestep_embedding(timesteps, self.model_channels)
        emb = self.time_embed[0].rows(t_emb, epilogue=ops.EPI_SILU)
        return self._emb_pack(self.time_embed[2].rows(emb, epilogue=ops.EPI_SILU))

    def forward_rows(self, x, timesteps, context_rows, kv_cache=None, y=None, emb_pack=None):
        """x: [B, Cin, H, W] fp32/bf16 NCHW; context_rows: bf16 [B*L, Dc]; y: class labels [B] of a class-conditional model
        (openaimodel.py:764-772).  emb_pack: optional EmbPack with `.all` = fp32 [B, sum Cout] from `time_embedding_rows` (then `timesteps` is not
        read).  Returns eps [B, Cout, H, W] fp32."""
        B, C, H, W = x.shape
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"
        if emb_pack is not None:
            assert y is None and emb_pack.all is not None and emb_pack.all.shape[0] == B and emb_pack.all.is_contiguous()
            emb_silu = emb_pack
        else:
            t_emb = ops.timestep_embedding(timesteps, self.model_channels)                 # bf16 [B, mc]
            emb = self.time_embed[0].rows(t_emb, epilogue=ops.EPI_SILU)                   # Linear + SiLU fused
            if y is None:
                emb_silu = self.time_embed[2].rows(emb, epilogue=ops.EPI_SILU)            # SiLU(emb): what every ResBlock consumes
            else:  # emb + label_emb(y) in fp32, then the SiLU every ResBlock starts with
                assert y.shape[0] == B
                e32 = self.time_embed[2].rows(emb, out_f32=True)
                lab = self._label_rows(y, e32.device)
                emb_silu = ops.silu_to_bf16(ops.lincomb([(e32, 1.0), (lab, 1.0)]))
            emb_silu = self._emb_pack(emb_silu)
        f = Feat(ops.nchw_to_rows(x, (C + 7) // 8 * 8), B, H, W)
        hs = []
        for module in self.input_blocks:
            if isinstance(module[0], TimestepBlock) or isinstance(module[0], (Downsample, Upsample)) or len(module) > 1:
                f = module.rows(f, emb_silu, context_rows, kv_cache)
            else:  # stem conv
                st = _stats_for(B, H, W, module[0].out_channels, f.t.device)
                stem, _, _ = module[0].rows(f.t, B, H, W, colstats=st)
                f = Feat(stem, B, H, W, st=st)
            hs.append(f)
        f = self.middle_block.rows(f, emb_silu, context_rows, kv_cache)
        for module in self.output_blocks:
            skip = hs.pop()
            st_a = f.st if f.t2 is None else None
            st_b = skip.st if skip.t2 is None else None
            f = Feat(f.materialize(), f.B, f.H, f.W, t2=skip.materialize(), st=st_a, st2=st_b)  # th.cat([h, hs.pop()], 1), deferred
            f = module.rows(f, emb_silu, context_rows, kv_cache)
        if self.predict_codebook_ids:   # openaimodel.py:783-784
            h = self.id_predictor[0].rows(f.materialize(), f.B, f.H * f.W, silu=False, colstats=f.st if f.t2 is None else None)
            logits, _, _ = self.id_predictor[1].rows(h, f.B, f.H, f.W, out_f32=True)
            return ops.rows_to_nchw(logits, f.B, f.H, f.W, out_dtype=torch.float32)
        h = self.out[0].rows(f.materialize(), f.B, f.H * f.W, silu=True, colstats=f.st if f.t2 is None else None)
        eps, _, _ = self.out[2].rows(h, f.B, f.H, f.W, out_f32=True)
        return ops.rows_to_nchw(eps, f.B, f.H, f.W, out_dtype=torch.float32)

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        """openaimodel.py:754-786."""
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"
        out = self.forward_rows(x, timesteps, self.context_rows(context) if context is not None else None, y=y)
        return out.to(x.dtype) if x.dtype != torch.float32 else out
