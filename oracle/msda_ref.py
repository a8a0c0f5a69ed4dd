"""Oracle (test infrastructure) for SURVEY.md §8(f) N2: multi-scale deformable attention forward.

Independent CPU restatement of what GroundingDINO's CUDA op computes (csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:237-299; the
reference's own readable statement is multi_scale_deformable_attn_pytorch, ms_deform_attn.py:93-133, which uses F.grid_sample):
explicit bilinear gathers with zero padding at pixel coordinates (x W - 0.5, y H - 0.5).  Pinned to tests/golden/msda.npz, which
tools/gen_golden.py produced by running the reference function.
"""
import torch


def ms_deform_attn(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    bs, S, heads, d = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    out = torch.zeros(bs, Q, heads, d, dtype=torch.float32)
    bi = torch.arange(bs).view(bs, 1, 1, 1).expand(bs, Q, heads, P)
    hi = torch.arange(heads).view(1, 1, heads, 1).expand(bs, Q, heads, P)
    for l in range(L):
        H, W = int(spatial_shapes[l, 0]), int(spatial_shapes[l, 1])
        v = value[:, int(level_start_index[l]):int(level_start_index[l]) + H * W].reshape(bs, H, W, heads, d)
        x = sampling_locations[:, :, :, l, :, 0] * W - 0.5
        y = sampling_locations[:, :, :, l, :, 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        acc = torch.zeros(bs, Q, heads, P, d)
        for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = (y0 + dy).long(), (x0 + dx).long()
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            g = v[bi, yy.clamp(0, H - 1), xx.clamp(0, W - 1), hi]          # [bs, Q, heads, P, d]
            acc = acc + (wgt * ok)[..., None] * g
        out = out + (attention_weights[:, :, :, l, :, None] * acc).sum(3)
    return out.reshape(bs, Q, heads * d)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, grad_output):
    """What `_C.ms_deform_attn_backward` returns (csrc/MsDeformAttn/ms_deform_im2col_cuda.cuh:88-160, ms_deform_attn_col2im_bilinear:
    value gradient scattered to the four corners, location gradient = level size x derivative of the bilinear weights, weight gradient =
    <grad_output, sampled value>), written out explicitly (no autograd).  Pinned to tests/golden/msda_bwd.npz = autograd through the
    reference's multi_scale_deformable_attn_pytorch.  Accumulates in float64 so that the scatter order cannot matter, returns fp32."""
    bs, S, heads, d = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    value = value.double()
    go = grad_output.double().reshape(bs, Q, heads, 1, d)                      # broadcast over the P samples of a level
    gv = torch.zeros(bs * S * heads, d, dtype=torch.float64)
    gl = torch.zeros(bs, Q, heads, L, P, 2, dtype=torch.float64)
    gw = torch.zeros(bs, Q, heads, L, P, dtype=torch.float64)
    bi = torch.arange(bs).view(bs, 1, 1, 1).expand(bs, Q, heads, P)
    hi = torch.arange(heads).view(1, 1, heads, 1).expand(bs, Q, heads, P)
    for l in range(L):
        H, W = int(spatial_shapes[l, 0]), int(spatial_shapes[l, 1])
        s0 = int(level_start_index[l])
        v = value[:, s0:s0 + H * W].reshape(bs, H, W, heads, d)
        aw = attention_weights[:, :, :, l, :].double()
        x = sampling_locations[:, :, :, l, :, 0].double() * W - 0.5
        y = sampling_locations[:, :, :, l, :, 1].double() * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        dots = {}
        for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = (y0 + dy).long(), (x0 + dx).long()
            ok = ((yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)).double()
            yc, xc = yy.clamp(0, H - 1), xx.clamp(0, W - 1)
            t = v[bi, yc, xc, hi] * ok[..., None]                               # [bs, Q, heads, P, d], zero outside the level
            dots[(dy, dx)] = (t * go).sum(-1)                                   # <grad_output, corner value>
            rows = ((bi * S + s0 + yc * W + xc) * heads + hi).reshape(-1)       # row of value[b, pixel, h, :]
            gv.index_add_(0, rows, ((aw * wgt * ok)[..., None] * go).reshape(-1, d))
        d00, d01, d10, d11 = dots[(0, 0)], dots[(0, 1)], dots[(1, 0)], dots[(1, 1)]
        gw[:, :, :, l, :] = (1 - ly) * ((1 - lx) * d00 + lx * d01) + ly * ((1 - lx) * d10 + lx * d11)
        gl[:, :, :, l, :, 0] = aw * W * ((1 - ly) * (d01 - d00) + ly * (d11 - d10))
        gl[:, :, :, l, :, 1] = aw * H * ((1 - lx) * (d10 - d00) + lx * (d11 - d01))
    return gv.reshape(bs, S, heads, d).float(), gl.float(), gw.float()
