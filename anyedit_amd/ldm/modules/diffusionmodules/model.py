"""First-stage autoencoder blocks on the HIP path — mirror of ldm/modules/diffusionmodules/model.py:
`ResnetBlock` (:90-149), `AttnBlock` (:152-202), `Downsample` (:68-87), `Upsample` (:50-65), `Encoder` (:452-543), `Decoder`
(:546-653), same constructor arguments, same submodule names -> the reference's kl-f8 checkpoints load unchanged.

Everything runs on channels-last bf16 rows with the kernels of the denoising path: GroupNorm(32, eps 1e-6)+swish =
ae_groupnorm_nhwc_bf16 (act = SiLU), 3x3 convs = ae_conv3x3_bf16 (nearest-x2 upsample folded into the gather), 1x1 convs =
ae_gemm_bf16.  Two things are specific to this stage:
  * Downsample pads (right, bottom) only and convolves with stride 2 / no padding (:80-84).  On the flipped image that is the
    symmetric pad-1 stride-2 convolution with the flipped kernel, so it runs on the same kernel (the two flips are row moves).
  * AttnBlock is ONE head of width c = 512 over h*w = 4096 tokens.  It is off the hot loop (once per image, 34 GFLOP), so it is
    three GEMMs around a row-softmax kernel (fp32 logits N x N per image) rather than a head_dim-512 flash kernel; the value bias
    is added after P V (softmax rows sum to 1).
"""
import torch
import torch.nn as nn

from anyedit_amd import ops

BF16 = torch.bfloat16


def nonlinearity(x):
    """model.py:41-43 (swish); only used by the module-level forward()s on NCHW tensors."""
    return x * torch.sigmoid(x)


def Normalize(in_channels, num_groups=32):
    """model.py:46-47."""
    return torch.nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _pad8(n):
    return (n + 7) // 8 * 8


class _Rows:
    """Per-module cache of packed weights (bf16, channel counts padded to multiples of 8 so every row stays 16-byte aligned)."""

    def _cache(self):
        if ops.cache_stale(self, "_pk", *self.parameters()):   # off the denoising loop: a walk over the block's own parameters is fine
            self._pk = {"dev": next(self.parameters()).device}
        return self._pk

    def repack(self):
        self._pk = None


def _conv3x3_packed(cache, name, conv, flip=False):
    key = ("c3", name, flip)
    if key not in cache:
        w = conv.weight.detach().float()
        cout, cin = w.shape[0], w.shape[1]
        cop = _pad8(cout)
        wp = torch.zeros(cop, cin, 3, 3, device=w.device)
        wp[:cout] = w.flip(2, 3) if flip else w
        b = torch.zeros(cop, device=w.device)
        if conv.bias is not None:
            b[:cout] = conv.bias.detach().float()
        cache[key] = (ops.pack_conv3x3(wp), b.contiguous(), cout)
    return cache[key]


def _conv1x1_packed(cache, name, conv):
    key = ("c1", name)
    if key not in cache:
        w = conv.weight.detach().float().reshape(conv.weight.shape[0], -1)
        cout, cin = w.shape
        wp = torch.zeros(_pad8(cout), _pad8(cin), device=w.device)
        wp[:cout, :cin] = w
        b = torch.zeros(_pad8(cout), device=w.device)
        if conv.bias is not None:
            b[:cout] = conv.bias.detach().float()
        cache[key] = (wp.to(BF16).contiguous(), b.contiguous(), cout)
    return cache[key]


def _gn(norm, x, B, HW, swish):
    return ops.groupnorm(x, norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous(), B, HW, norm.eps,
                         silu=swish, groups=norm.num_groups)


def _to_rows(x):
    B, C, H, W = x.shape
    return ops.nchw_to_rows(x, _pad8(C)), B, H, W


class Upsample(nn.Module, _Rows):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def rows(self, x, B, H, W):
        if not self.with_conv:   # model.py:54-58: the nearest x2 alone
            return ops.resample2x_rows(x, B, H, W)
        w, b, _ = _conv3x3_packed(self._cache(), "conv", self.conv)
        return ops.conv3x3(x, w, b, B, H, W, upsample2x=True)  # nearest x2 folded into the gather (model.py:61-65)

    def forward(self, x):
        r, B, H, W = _to_rows(x)
        y, Ho, Wo = self.rows(r, B, H, W)
        if y.shape[1] != x.shape[1] and not self.with_conv:   # rows carry the 8-channel padding of _to_rows; a conv's output does not
            y = y[:, :x.shape[1]].contiguous()
        return ops.rows_to_nchw(y, B, Ho, Wo, out_dtype=x.dtype)


class Downsample(nn.Module, _Rows):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def rows(self, x, B, H, W):
        """pad (0,1,0,1) + stride-2 valid conv (model.py:80-84) == flip(conv_s2_pad1(flip(x), flip(w))) for even H, W; with_conv=False: avg_pool2d(2, 2) (:85-86)."""
        if not self.with_conv:
            return ops.resample2x_rows(x, B, H, W, down=True)
        if H % 2 or W % 2:
            raise NotImplementedError("Downsample: odd spatial sizes are not on the AnyEdit path")
        C = x.shape[1]
        w, b, _ = _conv3x3_packed(self._cache(), "conv", self.conv, flip=True)
        xf = x.reshape(B, H * W, C).flip(1).reshape(B * H * W, C).contiguous()      # row move: reverses y and x together
        y, Ho, Wo = ops.conv3x3(xf, w, b, B, H, W, stride=2)
        return y.reshape(B, Ho * Wo, -1).flip(1).reshape(B * Ho * Wo, -1).contiguous(), Ho, Wo

    def forward(self, x):
        r, B, H, W = _to_rows(x)
        y, Ho, Wo = self.rows(r, B, H, W)
        if y.shape[1] != x.shape[1] and not self.with_conv:   # rows carry the 8-channel padding of _to_rows; a conv's output does not
            y = y[:, :x.shape[1]].contiguous()
        return ops.rows_to_nchw(y, B, Ho, Wo, out_dtype=x.dtype)


class ResnetBlock(nn.Module, _Rows):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        if temb_channels > 0:
            raise NotImplementedError("ResnetBlock with a timestep embedding belongs to the DDPM `Model`, not to the autoencoder")
        self.norm1 = Normalize(in_channels)
        self.conv1 = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = Normalize(out_channels)
        self.dropout = torch.nn.Dropout(dropout)
        self.conv2 = torch.nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = torch.nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def rows(self, x, B, H, W):
        c = self._cache()
        w1, b1, _ = _conv3x3_packed(c, "conv1", self.conv1)
        w2, b2, _ = _conv3x3_packed(c, "conv2", self.conv2)
        h, _, _ = ops.conv3x3(_gn(self.norm1, x, B, H * W, True), w1, b1, B, H, W)
        h = _gn(self.norm2, h, B, H * W, True)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                ws, bs, _ = _conv3x3_packed(c, "conv_shortcut", self.conv_shortcut)
                x, _, _ = ops.conv3x3(x, ws, bs, B, H, W)
            else:
                ws, bs, _ = _conv1x1_packed(c, "nin_shortcut", self.nin_shortcut)
                x = ops.gemm(x, ws, bs)
        y, _, _ = ops.conv3x3(h, w2, b2, B, H, W, residual=x)   # x + h fused into the conv epilogue (model.py:149)
        return y

    def forward(self, x, temb=None):
        assert temb is None
        r, B, H, W = _to_rows(x)
        return ops.rows_to_nchw(self.rows(r, B, H, W), B, H, W, out_dtype=x.dtype)


class AttnBlock(nn.Module, _Rows):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def rows(self, x, B, H, W):
        c = self._cache()
        C, N = self.in_channels, H * W
        if "qk" not in c:
            wq, bq, _ = _conv1x1_packed(c, "q", self.q)
            wk, bk, _ = _conv1x1_packed(c, "k", self.k)
            c["qk"] = (torch.cat([wq, wk], 0).contiguous(), torch.cat([bq, bk], 0).contiguous())
            c["wv"] = ops.pack_linear(self.v.weight)                        # [C, C]: V^T = Wv h^T, bias added after P V
            c["bv"] = self.v.bias.detach().float().contiguous()
            c["po"] = _conv1x1_packed(c, "proj_out", self.proj_out)
        h = _gn(self.norm, x, B, N, False)
        qk = ops.gemm(h, *c["qk"])                                          # [B*N, 2C]
        out = torch.empty(B * N, C, dtype=BF16, device=x.device)
        for b in range(B):                                                  # one image at a time: the logits are N x N fp32
            hb = h[b * N:(b + 1) * N]
            q, k = qk[b * N:(b + 1) * N, :C], qk[b * N:(b + 1) * N, C:]
            S = ops.gemm(q, k, out_f32=True)                                # w_[i, j] = sum_c q[i, c] k[j, c]   (model.py:188)
            P = ops.softmax_rows(S, int(C) ** (-0.5))                       # (model.py:189-190)
            vt = ops.gemm(c["wv"], hb)                                      # [C, N] = Wv h^T  (v without its bias, transposed)
            ops.gemm(P, vt, c["bv"], out=out[b * N:(b + 1) * N])            # h_[i, c] = sum_j P[i, j] v[j, c] + b_v[c]
        wo, bo, _ = c["po"]
        return ops.gemm(out, wo, bo, residual=x)                            # x + proj_out(h_)  (model.py:200-202)

    def forward(self, x):
        r, B, H, W = _to_rows(x)
        return ops.rows_to_nchw(self.rows(r, B, H, W), B, H, W, out_dtype=x.dtype)


def make_attn(in_channels, attn_type="vanilla", attn_kwargs=None):
    """model.py:280-297: every attention flavour of the reference computes the same function; one implementation here."""
    if attn_type == "none":
        return nn.Identity(in_channels)
    if attn_type not in ("vanilla", "vanilla-xformers"):
        raise NotImplementedError(f"attn_type {attn_type} is not used by the AnyEdit first stage")
    return AttnBlock(in_channels)


class Encoder(nn.Module, _Rows):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, double_z=True, use_linear_attn=False, attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = torch.nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)
        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.in_ch_mult = in_ch_mult
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = torch.nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, kernel_size=3, stride=1, padding=1)

    def rows(self, x, B, H, W):
        c = self._cache()
        w, b, _ = _conv3x3_packed(c, "conv_in", self.conv_in)
        h, _, _ = ops.conv3x3(x, w, b, B, H, W)
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block].rows(h, B, H, W)
                if len(self.down[i_level].attn) > 0:
                    h = self.down[i_level].attn[i_block].rows(h, B, H, W)
            if i_level != self.num_resolutions - 1:
                h, H, W = self.down[i_level].downsample.rows(h, B, H, W)
        h = self.mid.block_1.rows(h, B, H, W)
        h = self.mid.attn_1.rows(h, B, H, W)
        h = self.mid.block_2.rows(h, B, H, W)
        h = _gn(self.norm_out, h, B, H * W, True)
        w, b, cout = _conv3x3_packed(c, "conv_out", self.conv_out)
        y, _, _ = ops.conv3x3(h, w, b, B, H, W, out_f32=True)
        return y, H, W, cout

    def forward(self, x):
        r, B, H, W = _to_rows(x)
        y, H, W, cout = self.rows(r, B, H, W)
        return ops.rows_to_nchw(y, B, H, W, out_dtype=torch.float32)[:, :cout].to(x.dtype if x.dtype != torch.uint8 else torch.float32)


class Decoder(nn.Module, _Rows):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False, attn_type="vanilla",
                 **ignorekwargs):
        super().__init__()
        self.ch, self.temb_ch = ch, 0
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.give_pre_end, self.tanh_out = give_pre_end, tanh_out
        if give_pre_end or tanh_out:
            raise NotImplementedError("give_pre_end / tanh_out are not used by the kl-f8 autoencoder")
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = torch.nn.Conv2d(z_channels, block_in, kernel_size=3, stride=1, padding=1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type=attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(self.num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(make_attn(block_in, attn_type=attn_type))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)  # prepend to get consistent order (model.py:610)
        self.norm_out = Normalize(block_in)
        self.conv_out = torch.nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

    def rows(self, z, B, H, W):
        c = self._cache()
        w, b, _ = _conv3x3_packed(c, "conv_in", self.conv_in)
        h, _, _ = ops.conv3x3(z, w, b, B, H, W)
        h = self.mid.block_1.rows(h, B, H, W)
        h = self.mid.attn_1.rows(h, B, H, W)
        h = self.mid.block_2.rows(h, B, H, W)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block].rows(h, B, H, W)
                if len(self.up[i_level].attn) > 0:
                    h = self.up[i_level].attn[i_block].rows(h, B, H, W)
            if i_level != 0:
                h, H, W = self.up[i_level].upsample.rows(h, B, H, W)
        h = _gn(self.norm_out, h, B, H * W, True)
        w, b, cout = _conv3x3_packed(c, "conv_out", self.conv_out)
        y, _, _ = ops.conv3x3(h, w, b, B, H, W, out_f32=True)
        return y, H, W, cout

    def forward(self, z):
        self.last_z_shape = z.shape
        r, B, H, W = _to_rows(z)
        y, H, W, cout = self.rows(r, B, H, W)
        return ops.rows_to_nchw(y, B, H, W, out_dtype=torch.float32)[:, :cout].to(z.dtype)
