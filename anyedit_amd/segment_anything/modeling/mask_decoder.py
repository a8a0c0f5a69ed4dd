"""Mirror of segment_anything/segment_anything/modeling/mask_decoder.py on HIP kernels — SURVEY.md §8(f) N3.

Class names, constructor arguments and state-dict keys follow the reference (MaskDecoder :16-151, MLP :154-176).  The two
ConvTranspose2d(k=2, s=2) layers are GEMMs whose 4 * Cout output columns are the (dy, dx, co) sub-pixels of each input pixel; their
outputs are left in that un-shuffled order (LayerNorm2d + GELU are per-pixel, so they run on a [pixels*4, C/4] view) and the pixel
shuffle of both layers is folded into the final `hyper_in @ upscaled_embedding` kernel's read (ae_sam_mask_product_f32).
"""
from typing import Tuple, Type

import torch
from torch import nn

from anyedit_amd import ops
from anyedit_amd.ldm.modules.diffusionmodules.util import Linear, _Packed
from .image_encoder import LayerNorm2d

BF16 = torch.bfloat16


class ConvTranspose2x2(nn.ConvTranspose2d, _Packed):
    """nn.ConvTranspose2d(kernel_size=2, stride=2) as a GEMM: rows [P, Cin] -> [P, 4*Cout], column = (dy*2 + dx)*Cout + co."""

    def _pack(self):
        if self.kernel_size != (2, 2) or self.stride != (2, 2) or self.padding != (0, 0) or self.groups != 1:
            raise NotImplementedError("ConvTranspose2x2: only kernel_size=2, stride=2 (mask_decoder.py:54-59)")
        w = self.weight.detach().permute(2, 3, 1, 0).reshape(4 * self.out_channels, self.in_channels)  # [Cin,Cout,2,2] -> [(dy,dx,co), ci]
        return {"w": w.to(BF16).contiguous(), "b": self.bias.detach().float().repeat(4).contiguous()}

    def rows(self, x, epilogue=ops.EPI_NONE):
        pk = self._packed()
        return ops.gemm(x, pk["w"], pk["b"], epilogue=epilogue)

    def forward(self, x):
        B, C, H, W = x.shape
        y = self.rows(ops.nchw_to_rows(x.float().contiguous()))                       # [B*H*W, (dy, dx, co)]
        y = y.reshape(B, H, W, 2, 2, self.out_channels).permute(0, 5, 1, 3, 2, 4)   # public-API path only: shuffle in torch
        return y.reshape(B, self.out_channels, 2 * H, 2 * W).to(x.dtype)


class MLP(nn.Module):
    """mask_decoder.py:154-176; ReLU fused into each hidden layer's GEMM epilogue."""

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int, sigmoid_output: bool = False) -> None:
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output

    def rows(self, x, out=None):
        """x: bf16 rows (row stride free) -> fp32 [M, output_dim] (written into `out` when given)."""
        for i, layer in enumerate(self.layers):
            pk = layer._packed()
            if i < self.num_layers - 1:
                x = ops.gemm(x, pk["w"], pk["b"], epilogue=ops.EPI_RELU)
            else:
                x = ops.gemm(x, pk["w"], pk["b"], out_f32=True, out=out)
        return torch.sigmoid(x) if self.sigmoid_output else x

    def forward(self, x):
        shp = x.shape
        return self.rows(x.reshape(-1, shp[-1]).to(BF16).contiguous()).reshape(*shp[:-1], -1).to(x.dtype)


class MaskDecoder(nn.Module):
    """mask_decoder.py:16-151."""

    def __init__(self, *, transformer_dim: int, transformer: nn.Module, num_multimask_outputs: int = 3,
                 activation: Type[nn.Module] = nn.GELU, iou_head_depth: int = 3, iou_head_hidden_dim: int = 256) -> None:
        super().__init__()
        if activation is not nn.GELU:
            raise NotImplementedError("MaskDecoder: only nn.GELU (the SAM configuration) is implemented")
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_multimask_outputs = num_multimask_outputs
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_multimask_outputs + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.output_upscaling = nn.Sequential(
            ConvTranspose2x2(transformer_dim, transformer_dim // 4, kernel_size=2, stride=2),
            LayerNorm2d(transformer_dim // 4),
            activation(),
            ConvTranspose2x2(transformer_dim // 4, transformer_dim // 8, kernel_size=2, stride=2),
            activation(),
        )
        self.output_hypernetworks_mlps = nn.ModuleList(
            [MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(self.num_mask_tokens)])
        self.iou_prediction_head = MLP(transformer_dim, iou_head_hidden_dim, self.num_mask_tokens, iou_head_depth)

    def forward(self, image_embeddings: torch.Tensor, image_pe: torch.Tensor, sparse_prompt_embeddings: torch.Tensor,
                dense_prompt_embeddings: torch.Tensor, multimask_output: bool) -> Tuple[torch.Tensor, torch.Tensor]:
        """:71-110."""
        masks, iou_pred = self.predict_masks(image_embeddings=image_embeddings, image_pe=image_pe,
                                             sparse_prompt_embeddings=sparse_prompt_embeddings,
                                             dense_prompt_embeddings=dense_prompt_embeddings)
        mask_slice = slice(1, None) if multimask_output else slice(0, 1)
        return masks[:, mask_slice, :, :], iou_pred[:, mask_slice]

    def predict_masks(self, image_embeddings: torch.Tensor, image_pe: torch.Tensor, sparse_prompt_embeddings: torch.Tensor,
                      dense_prompt_embeddings: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """:112-151 over NCHW / [B, N, C] tensors as in the reference; converts to rows and calls predict_masks_rows."""
        if image_embeddings.shape[0] != 1 or image_pe.shape[0] != 1:
            raise NotImplementedError("MaskDecoder: one image per call (as SamPredictor / Sam.forward use it, sam.py:110-125)")
        _, c, h, w = image_embeddings.shape
        img_rows = ops.nchw_to_rows(image_embeddings.float().contiguous())
        pe_rows = ops.nchw_to_rows(image_pe.float().contiguous()).reshape(-1)
        d = dense_prompt_embeddings
        if d.stride(0) == 0 and d.stride(2) == 0 and d.stride(3) == 0:     # PromptEncoder's no-mask broadcast view
            dense_vec, dense_rows = d[0, :, 0, 0].to(BF16).contiguous(), None
        else:
            dense_vec, dense_rows = None, ops.nchw_to_rows(d.float().contiguous())
        return self.predict_masks_rows(img_rows, pe_rows, sparse_prompt_embeddings, h, w, dense_vec=dense_vec, dense_rows=dense_rows)

    def predict_masks_rows(self, img_rows, pe_rows, sparse, h, w, dense_vec=None, dense_rows=None):
        """img_rows [h*w, C] bf16, pe_rows [h*w*C] bf16 (flat), sparse [B, N, C]; dense as a [C] vector (no mask prompt) or rows
        [B*h*w, C].  Returns (masks [B, num_mask_tokens, 4h, 4w] fp32, iou_pred [B, num_mask_tokens] fp32)."""
        C, M = self.transformer_dim, self.num_mask_tokens
        B, hw = sparse.shape[0], h * w
        out_tokens = torch.cat([self.iou_token.weight.detach(), self.mask_tokens.weight.detach()], dim=0)
        tokens = torch.cat([out_tokens.unsqueeze(0).expand(B, -1, -1).to(sparse.dtype), sparse], dim=1)
        T = tokens.shape[1]
        tokens = tokens.reshape(B * T, C).to(BF16).contiguous()
        if dense_rows is None:
            src = ops.add_bcast(img_rows, dense_vec).repeat(B, 1)        # identical for every prompt before the first block
        else:
            src = ops.add_bcast(dense_rows, img_rows.reshape(-1))

        hs, src = self.transformer.rows(src, pe_rows, tokens, B, T, hw)
        hs = hs.reshape(B, T, C)

        up = self.output_upscaling[0].rows(src)                                              # [B*hw, 4 * C/4]
        ln = self.output_upscaling[1]
        up = ops.layernorm_act(up.reshape(B * hw * 4, C // 4), ln.weight.detach().float().contiguous(),
                               ln.bias.detach().float().contiguous(), eps=ln.eps, gelu=True)
        up = self.output_upscaling[3].rows(up, epilogue=ops.EPI_GELU).reshape(B * hw * 16, C // 8)

        hyper = torch.empty(B, M, C // 8, dtype=torch.float32, device=src.device)
        for i in range(M):
            self.output_hypernetworks_mlps[i].rows(hs[:, 1 + i, :], out=hyper[:, i, :])
        masks = ops.sam_mask_product(up, hyper, B, h, w)
        iou_pred = self.iou_prediction_head.rows(hs[:, 0, :])
        return masks, iou_pred
