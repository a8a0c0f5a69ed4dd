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
