"""Mirror of ldm/modules/diffusionmodules/util.py — schedules (host, numpy f64, bit-exact integer bookkeeping) and
the layer factories through which every UNet layer is created (util.py:202-238), here returning HIP-backed layers.
"""

import numpy as np
import torch
import torch.nn as nn

from anyedit_amd import ops


# ---------------------------------------------------------------------------------------------- schedules (host)
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """util.py:21-43 (float64 on the host, like the reference)."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "cosine":
        timesteps = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = timesteps / (1 + cosine_s) * np.pi / 2
        alphas = torch.cos(alphas).pow(2)
        alphas = alphas / alphas[0]
        betas = 1 - alphas[1:] / alphas[:-1]
        betas = torch.clamp(betas, min=0, max=0.999)
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """util.py:46-60 — integer bookkeeping, bit-exact (length != S when S does not divide T, as in the reference)."""
    if ddim_discr_method == "uniform":
        c = num_ddpm_timesteps // num_ddim_timesteps
        ddim_timesteps = np.asarray(list(range(0, num_ddpm_timesteps, c)))
    elif ddim_discr_method == "quad":
        ddim_timesteps = ((np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = ddim_timesteps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """util.py:63-74."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, this results in the following sigma_t schedule for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev


def extract_into_tensor(a, t, x_shape):
    """util.py:96-99."""
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


def checkpoint(func, inputs, params, flag):
    """util.py:102-118: gradient checkpointing is a training-memory device; the forward value is func(*inputs)."""
    return func(*inputs)


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """util.py:154-174 on the GPU ([cos, sin] order); returns fp32 [N, dim]."""
    if repeat_only:
        return timesteps[:, None].repeat(1, dim)
    return ops.timestep_embedding(timesteps, dim, max_period, out_f32=True)


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def noise_like(shape, device, repeat=False):
    """util.py:267-270."""
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


# ---------------------------------------------------------------------------------------------- HIP-backed layers
class _Packed:
    """Mixin: lazily packed bf16 weights for the HIP kernels; call .repack() after changing parameters."""

    def repack(self):
        self._pk = None

    def _packed(self):
        if ops.cache_stale(self, "_pk", self.weight, getattr(self, "bias", None)):
            pk = self._pack()
            pk["device"] = self.weight.device
            self._pk = pk
        return self._pk


class Linear(nn.Linear, _Packed):
    """nn.Linear whose forward is ae_gemm_bf16.  State-dict identical to nn.Linear."""

    def _pack(self):
        return {"w": ops.pack_linear(self.weight), "b": None if self.bias is None else self.bias.detach().float().contiguous()}

    def rows(self, x, residual=None, epilogue=ops.EPI_NONE, out_f32=False, a2=None, rowstats=None):
        """rowstats: optional `ops.rowstats_buffer` that receives the row statistics of the result (for a LayerNorm folded into the next GEMM)."""
        pk = self._packed()
        if rowstats is not None:
            # forwarded in full: ops.gemm refuses the combinations the row-statistics epilogue does not carry (it raised nothing here before, the
            # arguments were silently dropped — ADVICE r4)
            return ops.gemm(x, pk["w"], pk["b"], residual=residual, epilogue=epilogue, out_f32=out_f32, a2=a2, rowstats=rowstats)
        return ops.gemm(x, pk["w"], pk["b"], residual=residual, epilogue=epilogue, out_f32=out_f32, a2=a2)

    def forward(self, x):
        shp = x.shape
        y = self.rows(x.reshape(-1, shp[-1]).to(torch.bfloat16).contiguous())
        return y.reshape(*shp[:-1], -1).to(x.dtype)


class Conv2d(nn.Conv2d, _Packed):
    """nn.Conv2d (3x3 pad 1 stride 1/2, or 1x1) whose forward is the implicit-GEMM / GEMM HIP kernel."""

    def _pack(self):
        k = self.kernel_size[0]
        b = None if self.bias is None else self.bias.detach().float().contiguous()
        if k == 1:
            return {"w": ops.pack_linear(self.weight), "b": b}
        if k == 3 and self.padding[0] == 1 and self.stride[0] in (1, 2) and self.groups == 1 and self.dilation[0] == 1:
            return {"w": ops.pack_conv3x3(self.weight), "b": b}
        raise NotImplementedError(f"anyedit_amd Conv2d: unsupported configuration k={k} stride={self.stride} padding={self.padding}")

    def rows(self, x, B, H, W, addvec=None, residual=None, upsample2x=False, out_f32=False, a2=None, colstats=None):
        """x: channels-last rows [B*H*W, Cin] -> (rows [B*Ho*Wo, Cout], Ho, Wo).  colstats: `ops.colstats_buffer` for the output."""
        pk = self._packed()
        if self.kernel_size[0] == 1:
            return ops.gemm(x, pk["w"], pk["b"], residual=residual, out_f32=out_f32, a2=a2, colstats=colstats), H, W
        if addvec is None and residual is None and not out_f32 and a2 is None and ops.stem_im2col_ok(self.in_channels, self.stride[0], upsample2x):
            # the 8-channel stem: nine taps x 8 channels are 72 real K columns — as im2col rows + a K = 128 dense GEMM instead of nine 64-deep K tiles of padding
            if "w_taps" not in pk:
                pk["w_taps"] = ops.pack_conv3x3_taps8(self.weight)
            return ops.gemm(ops.im2col3x3_c8(x, B, H, W), pk["w_taps"], pk["b"], colstats=colstats), H, W
        if upsample2x and addvec is None and residual is None and not out_f32 and a2 is None and \
                ops.up2_subpixel_ok(self.in_channels, self.out_channels, H, W, colstats is not None) and self.stride[0] == 1:
            # Upsample (openaimodel.py:108-118): nearest x2 + conv3x3 = four 2x2 convs on the low-resolution map with summed taps — 4/9 of the multiply-adds
            if "w_up2" not in pk:
                pk["w_up2"] = ops.pack_conv3x3_up2(self.weight)
            return ops.conv3x3_up2(x, pk["w_up2"], pk["b"], B, H, W, colstats=colstats)
        Ho, Wo = self.out_hw(H, W, upsample2x)
        ko = ops.conv_k_order(B * Ho * Wo, self.in_channels, self.out_channels, self.stride[0], upsample2x)
        w = pk["w"]
        if ko:  # second pack of the same weights in the chunk-major K order, made on first use (the 64x64-level convs: 2-6 MB each)
            if "w_kmajor" not in pk:
                pk["w_kmajor"] = ops.pack_conv3x3(self.weight, k_order=1)
            w = pk["w_kmajor"]
        return ops.conv3x3(x, w, pk["b"], B, H, W, addvec=addvec, residual=residual, stride=self.stride[0],
                           upsample2x=upsample2x, out_f32=out_f32, colstats=colstats, k_order=ko)

    def rows_partials(self, x, B, H, W):
        """The stride-1 3x3 conv stopped at the fp32 partials of its split-K plan (`ops.conv3x3_partials`; ask `ops.conv3x3_gn_splitk_ok` first): the bias is NOT
        applied — it goes to `ops.groupnorm_splitk` with the partials (`self.packed_bias()`)."""
        pk = self._packed()
        ko = ops.conv_k_order(B * H * W, self.in_channels, self.out_channels, 1, False)
        w = pk["w"]
        if ko:
            if "w_kmajor" not in pk:
                pk["w_kmajor"] = ops.pack_conv3x3(self.weight, k_order=1)
            w = pk["w_kmajor"]
        return ops.conv3x3_partials(x, w, B, H, W, k_order=ko)[0]

    def packed_bias(self):
        return self._packed()["b"]

    def out_hw(self, H, W, upsample2x=False):
        """Output extent of `rows` for an H x W input."""
        if self.kernel_size[0] == 1:
            return H, W
        Hv, Wv = (2 * H, 2 * W) if upsample2x else (H, W)
        return (Hv - 1) // self.stride[0] + 1, (Wv - 1) // self.stride[0] + 1

    def forward(self, x):
        B, C, H, W = x.shape
        cpad = (C + 7) // 8 * 8
        if cpad != C and self.kernel_size[0] == 1:
            raise NotImplementedError("1x1 conv with channels not a multiple of 8")
        y, Ho, Wo = self.rows(ops.nchw_to_rows(x, cpad), B, H, W)
        return ops.rows_to_nchw(y, B, Ho, Wo, out_dtype=x.dtype)


class GroupNorm32(nn.GroupNorm):
    """util.py:217-219: statistics in fp32.  `rows` runs the fused GroupNorm(+SiLU) HIP kernel on channels-last rows."""

    def _affine(self):
        if ops.cache_stale(self, "_pk", self.weight, self.bias):
            self._pk = (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous())
        return self._pk

    def repack(self):
        self._pk = None

    def rows(self, x, B, HW, silu=False, x2=None, colstats=None, colstats2=None):
        g, b = self._affine()
        return ops.groupnorm(x, g, b, B, HW, self.eps, silu=silu, groups=self.num_groups, x2=x2, colstats=colstats, colstats2=colstats2)

    def forward(self, x):
        B, C, H, W = x.shape
        y = self.rows(ops.nchw_to_rows(x), B, H * W)
        return ops.rows_to_nchw(y, B, H, W, out_dtype=x.dtype)


class LayerNorm(nn.LayerNorm):
    def _affine(self):
        if ops.cache_stale(self, "_pk", self.weight, self.bias):
            self._pk = (self.weight.detach().float().contiguous(), self.bias.detach().float().contiguous())
        return self._pk

    def repack(self):
        self._pk = None

    def rows(self, x):
        g, b = self._affine()
        return ops.layernorm(x, g, b, self.eps)

    def forward(self, x):
        shp = x.shape
        return self.rows(x.reshape(-1, shp[-1]).to(torch.bfloat16).contiguous()).reshape(shp).to(x.dtype)


class Conv1d(nn.Conv1d, _Packed):
    """nn.Conv1d with kernel_size 1 (what AttentionBlock's qkv / proj_out are, openaimodel.py:304, 312) as the dense GEMM over rows [B*T, Cin]."""

    def _pack(self):
        if self.kernel_size[0] != 1 or self.stride[0] != 1 or self.padding[0] != 0 or self.groups != 1:
            raise NotImplementedError(f"anyedit_amd Conv1d: only the 1-wide pointwise form (k={self.kernel_size} stride={self.stride} padding={self.padding})")
        return {"w": ops.pack_linear(self.weight), "b": None if self.bias is None else self.bias.detach().float().contiguous()}

    def rows(self, x, residual=None):
        pk = self._packed()
        return ops.gemm(x, pk["w"], pk["b"], residual=residual)

    def forward(self, x):
        B, C, T = x.shape
        y = self.rows(ops.nchw_to_rows(x.reshape(B, C, T, 1)))
        return ops.rows_to_nchw(y, B, T, 1, out_dtype=x.dtype).reshape(B, -1, T)


def conv_nd(dims, *args, **kwargs):
    """util.py:222-232."""
    if dims == 1:
        return Conv1d(*args, **kwargs)
    if dims == 2:
        return Conv2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims} (the AnyEdit hot path is 2-D)")


def linear(*args, **kwargs):
    """util.py:235-238."""
    return Linear(*args, **kwargs)


def normalization(channels):
    """util.py:202-208."""
    return GroupNorm32(32, channels)


class AvgPool2d(nn.AvgPool2d):
    """nn.AvgPool2d(2, 2) (what Downsample(use_conv=False) holds, openaimodel.py:152-155; no parameters, no state-dict keys) on the HIP resampling kernel."""

    def __init__(self, kernel_size=2, stride=None, **kw):
        super().__init__(kernel_size, stride, **kw)
        ks = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        st = self.stride if isinstance(self.stride, int) else self.stride[0]
        if ks != 2 or st != 2 or kw:
            raise NotImplementedError("AvgPool2d: only the 2x2 / stride-2 mean of openaimodel.py:146-155")

    def rows(self, x, B, H, W):
        return ops.resample2x_rows(x, B, H, W, down=True)

    def forward(self, x):
        B, C, H, W = x.shape
        y, Ho, Wo = self.rows(ops.nchw_to_rows(x), B, H, W)
        return ops.rows_to_nchw(y, B, Ho, Wo, out_dtype=x.dtype)


def avg_pool_nd(dims, *args, **kwargs):
    """util.py:241-251."""
    if dims == 2:
        return AvgPool2d(*args, **kwargs)
    raise ValueError(f"unsupported dimensions: {dims} (the AnyEdit hot path is 2-D)")
