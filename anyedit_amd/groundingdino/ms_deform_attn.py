"""GroundingDINO `MultiScaleDeformableAttention` on MI355X — mirror of
GroundingDINO/groundingdino/models/GroundingDINO/ms_deform_attn.py:136-352.

Same constructor, parameter names (`sampling_offsets`, `attention_weights`, `value_proj`, `output_proj`) and forward contract, so
a GroundingDINO checkpoint's encoder / decoder layers load unchanged.  The sampling core — the reference's only native operator,
`_C.ms_deform_attn_forward` — is `ae_ms_deform_attn_fwd_f32`; the four small projections are exact-fp32 MFMA GEMMs (`ae_linear_f32`:
GroundingDINO runs in fp32 and the sampling offsets are coordinates, so they do not go through the bf16 GEMM).
"""
import math
import warnings
from typing import Optional

import torch
import torch.nn as nn
from torch.nn.init import xavier_uniform_

from anyedit_amd import ops


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MultiScaleDeformableAttnFunction(torch.autograd.Function):
    """ms_deform_attn.py:42-90: forward = `_C.ms_deform_attn_forward`, backward = `_C.ms_deform_attn_backward` (once-differentiable;
    gradients for value, sampling_locations and attention_weights, None for the shape tensors and im2col_step)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        out = ops.ms_deform_attn(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, w = ctx.saved_tensors
        gv, gl, gw = ops.ms_deform_attn_bwd(value, shapes, starts, loc, w, grad_output, ctx.im2col_step)
        return gv, None, None, gl.to(loc.dtype), gw.to(w.dtype), None


class _LinearF32(torch.autograd.Function):
    """nn.Linear on `ae_linear_f32` with its adjoints on the same kernel (the raw kernel call has no autograd node, so the module's
    projections dropped the gradients of query / src / weights silently and MultiScaleDeformableAttnFunction.backward was unreachable from the
    module).  dX = dY W, dW = dY^T X, db = sum dY; the row count is zero-padded to the kernel's K % 16 == 0 for the weight gradient."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return ops.linear_f32(x, w, b)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        K, N = w.shape[1], w.shape[0]
        gy2 = gy.reshape(-1, N).float().contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.linear_f32(gy2, w.detach().t().contiguous()).reshape(x.shape)            # [M, N] x [K, N]^T
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, K).float()
            M = x2.shape[0]
            Mp = (M + 15) // 16 * 16
            gyt = torch.zeros(N, Mp, dtype=torch.float32, device=gy2.device)
            gyt[:, :M] = gy2.t()
            xt = torch.zeros(K, Mp, dtype=torch.float32, device=gy2.device)
            xt[:, :M] = x2.t()
            gw = ops.linear_f32(gyt, xt)                                                        # [N, Mp] x [K, Mp]^T
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gy2.sum(0)
        return gx, gw, gb


def _linear_f32(m, t):
    t = t.float()
    if torch.is_grad_enabled() and (t.requires_grad or m.weight.requires_grad or (m.bias is not None and m.bias.requires_grad)):
        return _LinearF32.apply(t, m.weight, m.bias)
    return ops.linear_f32(t, m.weight, m.bias)   # exact fp32 (f32-input MFMA), no torch / library GEMM


def multi_scale_deformable_attn(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                                im2col_step=64):
    """Functional entry with the argument list of MultiScaleDeformableAttnFunction.forward (ms_deform_attn.py:42-60)."""
    if torch.is_grad_enabled() and (value.requires_grad or sampling_locations.requires_grad or attention_weights.requires_grad):
        return MultiScaleDeformableAttnFunction.apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                                      attention_weights, im2col_step)
    return ops.ms_deform_attn(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)


class MultiScaleDeformableAttention(nn.Module):
    def __init__(self, embed_dim: int = 256, num_heads: int = 8, num_levels: int = 4, num_points: int = 4, img2col_step: int = 64,
                 batch_first: bool = False):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("embed_dim must be divisible by num_heads, but got {} and {}".format(embed_dim, num_heads))
        head_dim = embed_dim // num_heads
        self.batch_first = batch_first
        if not _is_power_of_2(head_dim):
            warnings.warn("MSDeformAttn: a power-of-2 head dim is more efficient")
        if head_dim % 4:
            raise ValueError("head_dim must be a multiple of 4 for the 16-byte gathers of the HIP kernel")
        self.im2col_step = img2col_step
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.num_levels = num_levels
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dim, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dim, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self.init_weights()

    def _reset_parameters(self):
        return self.init_weights()

    def init_weights(self):
        """Initial state of ms_deform_attn.py:197-219: zero offset / weight matrices, and an offset bias that places point k of
        head h at distance k+1 along direction 2*pi*h/heads (normalised to the unit square), identical for every level."""
        H, L, P = self.num_heads, self.num_levels, self.num_points
        ang = torch.arange(H, dtype=torch.float32) * (2.0 * math.pi / H)
        dirs = torch.stack([ang.cos(), ang.sin()], -1)
        dirs = dirs / dirs.abs().max(-1, keepdim=True)[0]
        steps = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)
        bias = dirs.view(H, 1, 1, 2).repeat(1, L, P, 1) * steps
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias = nn.Parameter(bias.reshape(-1))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def _locations(self, reference_points, offsets, spatial_shapes):
        """Sampling locations in [0, 1]^2 (x, y): around reference points (offsets in pixels of each level, :315-320) or inside
        reference boxes (offsets in units of half a box per num_points, :321-327)."""
        ref = reference_points[:, :, None, :, None, :]
        if reference_points.shape[-1] == 2:
            wh = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            return ref + offsets / wh[None, None, None, :, None, :]
        if reference_points.shape[-1] == 4:
            return ref[..., :2] + offsets / self.num_points * ref[..., 2:] * 0.5
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(reference_points.shape[-1]))

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None, value: Optional[torch.Tensor] = None,
                query_pos: Optional[torch.Tensor] = None, key_padding_mask: Optional[torch.Tensor] = None,
                reference_points: Optional[torch.Tensor] = None, spatial_shapes: Optional[torch.Tensor] = None,
                level_start_index: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        """Same contract as ms_deform_attn.py:231-352: (n, bs, c) tensors unless batch_first; returns the projected output."""
        src = query if value is None else value
        q = query if query_pos is None else query + query_pos
        if not self.batch_first:
            q, src = q.permute(1, 0, 2), src.permute(1, 0, 2)
        bs, nq, _ = q.shape
        ns = src.shape[1]
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == ns
        lin = _linear_f32
        v = lin(self.value_proj, src)
        if key_padding_mask is not None:
            v = v.masked_fill(key_padding_mask[..., None], 0.0)
        H, L, P = self.num_heads, self.num_levels, self.num_points
        offsets = lin(self.sampling_offsets, q).view(bs, nq, H, L, P, 2)
        weights = lin(self.attention_weights, q).view(bs, nq, H, L * P).softmax(-1).view(bs, nq, H, L, P)
        loc = self._locations(reference_points, offsets, spatial_shapes)
        out = multi_scale_deformable_attn(v.view(bs, ns, H, -1).float(), spatial_shapes, level_start_index, loc.float(),
                                          weights.float(), self.im2col_step)
        out = lin(self.output_proj, out).to(query.dtype)
        return out if self.batch_first else out.permute(1, 0, 2)
