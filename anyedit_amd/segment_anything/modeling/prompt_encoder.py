"""Mirror of segment_anything/segment_anything/modeling/prompt_encoder.py on HIP kernels — SURVEY.md §8(f) N3.

Class names, constructor arguments and state-dict keys follow the reference (PromptEncoder :16-171, PositionEmbeddingRandom
:174-214).  Point and box prompts are one kernel launch each (random-Fourier encoding + the label-selected learned embedding,
ae_sam_pe_encode_f32); a mask prompt is one fused kernel for conv-LN-GELU-conv-LN-GELU plus a GEMM for the closing 1x1 convolution.
"""
from typing import Any, Optional, Tuple, Type

import torch
from torch import nn

from anyedit_amd import ops
from .image_encoder import LayerNorm2d


class PositionEmbeddingRandom(nn.Module):
    """prompt_encoder.py:174-214: positional encoding using random spatial frequencies."""

    def __init__(self, num_pos_feats: int = 64, scale: Optional[float] = None) -> None:
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer("positional_encoding_gaussian_matrix", scale * torch.randn((2, num_pos_feats)))
        self._grid = None

    def _gauss(self):
        return self.positional_encoding_gaussian_matrix.detach().float().contiguous()

    def encode(self, coords_xy: torch.Tensor, image_size: Tuple[int, int], labels=None, table=None, offset: float = 0.0):
        """coords_xy [N, 2] pixel coordinates -> [N, 2F] fp32; see ops.sam_pe_encode for labels / table."""
        return ops.sam_pe_encode(coords_xy.float().contiguous(), self._gauss(), image_size, labels=labels, table=table, offset=offset)

    def _pe_encoding(self, coords: torch.Tensor) -> torch.Tensor:
        """:183-190: coords normalised to [0, 1], shape d_1 x ... x d_n x 2."""
        return self.encode(coords.reshape(-1, 2), (1, 1)).reshape(*coords.shape[:-1], -1)

    def forward(self, size: Tuple[int, int]) -> torch.Tensor:
        """:192-204: encoding of the pixel centres of an h x w grid, C x H x W.  Constant per size: computed once."""
        h, w = size
        device: Any = self.positional_encoding_gaussian_matrix.device
        if self._grid is None or self._grid[0] != (h, w, device):
            ys, xs = torch.meshgrid(torch.arange(h, device=device, dtype=torch.float32),
                                    torch.arange(w, device=device, dtype=torch.float32), indexing="ij")
            pe = self.encode(torch.stack([xs, ys], dim=-1).reshape(-1, 2), (h, w), offset=0.5)   # rows [h*w, C]
            self._grid = ((h, w, device), pe.reshape(h, w, -1).permute(2, 0, 1).contiguous(), pe.to(torch.bfloat16).reshape(-1))
        return self._grid[1]

    def grid_rows(self, size: Tuple[int, int]) -> torch.Tensor:
        """The same encoding as flat channels-last bf16 rows [h*w*C] (what the decoder kernels consume)."""
        self.forward(size)
        return self._grid[2]

    def forward_with_coords(self, coords_input: torch.Tensor, image_size: Tuple[int, int]) -> torch.Tensor:
        """:206-214: un-normalised (x, y) pixel coordinates, B x N x 2 -> B x N x C."""
        return self.encode(coords_input.reshape(-1, 2), image_size).reshape(*coords_input.shape[:-1], -1)


class PromptEncoder(nn.Module):
    """prompt_encoder.py:16-171."""

    def __init__(self, embed_dim: int, image_embedding_size: Tuple[int, int], input_image_size: Tuple[int, int],
                 mask_in_chans: int, activation: Type[nn.Module] = nn.GELU) -> None:
        super().__init__()
        if activation is not nn.GELU:
            raise NotImplementedError("PromptEncoder: only nn.GELU (the SAM configuration) is implemented")
        self.embed_dim = embed_dim
        self.input_image_size = input_image_size
        self.image_embedding_size = image_embedding_size
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings: int = 4  # pos/neg point + 2 box corners
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, embed_dim) for _ in range(self.num_point_embeddings)])
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_input_size = (4 * image_embedding_size[0], 4 * image_embedding_size[1])
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_in_chans // 4, kernel_size=2, stride=2),
            LayerNorm2d(mask_in_chans // 4),
            activation(),
            nn.Conv2d(mask_in_chans // 4, mask_in_chans, kernel_size=2, stride=2),
            LayerNorm2d(mask_in_chans),
            activation(),
            nn.Conv2d(mask_in_chans, embed_dim, kernel_size=1),
        )
        self.no_mask_embed = nn.Embedding(1, embed_dim)
        self._pk = None

    def repack(self):
        self._pk = None

    def _packed(self):
        dev = self.no_mask_embed.weight.device
        if ops.cache_stale(self, "_pk", *self.parameters()):
            f = lambda t: t.detach().float().contiguous()
            md = self.mask_downscaling
            self._pk = {
                "device": dev,
                # row 0: not_a_point (label -1); rows 1..4: point_embeddings[0..3] (labels 0 / 1 = points, 2 / 3 = box corners)
                "table": f(torch.cat([self.not_a_point_embed.weight] + [e.weight for e in self.point_embeddings], dim=0)),
                "md": [f(md[0].weight), f(md[0].bias), f(md[1].weight), f(md[1].bias), f(md[3].weight), f(md[3].bias), f(md[4].weight),
                       f(md[4].bias)],
                "w6": ops.pack_linear(md[6].weight), "b6": f(md[6].bias),
            }
        return self._pk

    def get_dense_pe(self) -> torch.Tensor:
        """:62-71: 1 x embed_dim x h x w."""
        return self.pe_layer(self.image_embedding_size).unsqueeze(0)

    def _embed_points(self, points: torch.Tensor, labels: torch.Tensor, pad: bool) -> torch.Tensor:
        """:73-93 (the +0.5 pixel-centre shift is the kernel's offset)."""
        if pad:
            points = torch.cat([points, torch.full((points.shape[0], 1, 2), -0.5, device=points.device, dtype=points.dtype)], dim=1)
            labels = torch.cat([labels, -torch.ones((labels.shape[0], 1), device=labels.device, dtype=labels.dtype)], dim=1)
        lab = labels.reshape(-1).to(torch.int32)
        if bool(((lab < -1) | (lab > 1)).any()):
            raise ValueError("PromptEncoder: point labels must be -1 (padding), 0 (background) or 1 (foreground)")
        e = self.pe_layer.encode(points.reshape(-1, 2), self.input_image_size, labels=lab.contiguous(), table=self._packed()["table"],
                                 offset=0.5)
        return e.reshape(points.shape[0], points.shape[1], self.embed_dim)

    def _embed_boxes(self, boxes: torch.Tensor) -> torch.Tensor:
        """:95-102: the two corners, tagged with point_embeddings[2] / [3]."""
        coords = boxes.reshape(-1, 2)
        lab = torch.tensor([2, 3], device=boxes.device, dtype=torch.int32).repeat(coords.shape[0] // 2)
        e = self.pe_layer.encode(coords, self.input_image_size, labels=lab, table=self._packed()["table"], offset=0.5)
        return e.reshape(-1, 2, self.embed_dim)

    def _embed_masks_rows(self, masks: torch.Tensor) -> torch.Tensor:
        if self.mask_downscaling[3].out_channels != 16:
            raise NotImplementedError("PromptEncoder: mask prompts are implemented for mask_in_chans = 16 (build_sam.py:87)")
        pk = self._packed()
        w1, b1, g1, e1, w2, b2, g2, e2 = pk["md"]
        rows16 = ops.sam_mask_downscale(masks.float(), w1, b1, g1, e1, w2, b2, g2, e2, eps=self.mask_downscaling[1].eps)
        return ops.gemm(rows16, pk["w6"], pk["b6"])

    def _embed_masks(self, masks: torch.Tensor) -> torch.Tensor:
        """:104-107."""
        B, _, H4, W4 = masks.shape
        return ops.rows_to_nchw(self._embed_masks_rows(masks), B, H4 // 4, W4 // 4)

    def _get_batch_size(self, points, boxes, masks) -> int:
        if points is not None:
            return points[0].shape[0]
        elif boxes is not None:
            return boxes.shape[0]
        elif masks is not None:
            return masks.shape[0]
        return 1

    def _get_device(self) -> torch.device:
        return self.point_embeddings[0].weight.device

    def forward(self, points: Optional[Tuple[torch.Tensor, torch.Tensor]], boxes: Optional[torch.Tensor],
                masks: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """:128-171: (sparse B x N x embed_dim, dense B x embed_dim x h x w).  Without a mask prompt the dense embedding is the
        broadcast (stride-0) view of no_mask_embed, exactly as in the reference; MaskDecoder recognises that view."""
        bs = self._get_batch_size(points, boxes, masks)
        sparse_embeddings = torch.empty((bs, 0, self.embed_dim), device=self._get_device())
        if points is not None:
            coords, labels = points
            point_embeddings = self._embed_points(coords, labels, pad=(boxes is None))
            sparse_embeddings = torch.cat([sparse_embeddings, point_embeddings], dim=1)
        if boxes is not None:
            sparse_embeddings = torch.cat([sparse_embeddings, self._embed_boxes(boxes)], dim=1)
        if masks is not None:
            dense_embeddings = self._embed_masks(masks)
        else:
            dense_embeddings = self.no_mask_embed.weight.detach().reshape(1, -1, 1, 1).expand(
                bs, -1, self.image_embedding_size[0], self.image_embedding_size[1])
        return sparse_embeddings, dense_embeddings
