"""Mirror of segment_anything/segment_anything/modeling/sam.py — SURVEY.md §8(f) N3.

`Sam` (sam.py:18-174) ties the image encoder, prompt encoder and mask decoder together; preprocess / postprocess_masks are single
fused HIP kernels (normalise + pad; bilinear -> crop -> bilinear composed per output pixel, no padded-square intermediate).
"""
from typing import Any, Dict, List, Tuple

import torch
from torch import nn

from anyedit_amd import ops
from .image_encoder import ImageEncoderViT
from .mask_decoder import MaskDecoder
from .prompt_encoder import PromptEncoder


class Sam(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, image_encoder: ImageEncoderViT, prompt_encoder: PromptEncoder, mask_decoder: MaskDecoder,
                 pixel_mean: List[float] = [123.675, 116.28, 103.53], pixel_std: List[float] = [58.395, 57.12, 57.375]) -> None:
        super().__init__()
        self.image_encoder = image_encoder
        self.prompt_encoder = prompt_encoder
        self.mask_decoder = mask_decoder
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1), False)

    @property
    def device(self) -> Any:
        return self.pixel_mean.device

    @torch.no_grad()
    def forward(self, batched_input: List[Dict[str, Any]], multimask_output: bool) -> List[Dict[str, torch.Tensor]]:
        """sam.py:53-131: end-to-end masks for a list of images with their prompts."""
        input_images = torch.cat([self.preprocess(x["image"][None]) for x in batched_input], dim=0)
        image_embeddings = self.image_encoder(input_images)
        outputs = []
        for image_record, curr_embedding in zip(batched_input, image_embeddings):
            points = (image_record["point_coords"], image_record["point_labels"]) if "point_coords" in image_record else None
            sparse_embeddings, dense_embeddings = self.prompt_encoder(points=points, boxes=image_record.get("boxes", None),
                                                                      masks=image_record.get("mask_inputs", None))
            low_res_masks, iou_predictions = self.mask_decoder(image_embeddings=curr_embedding.unsqueeze(0),
                                                               image_pe=self.prompt_encoder.get_dense_pe(),
                                                               sparse_prompt_embeddings=sparse_embeddings,
                                                               dense_prompt_embeddings=dense_embeddings,
                                                               multimask_output=multimask_output)
            _, masks = self.postprocess_masks_fused(low_res_masks, image_record["image"].shape[-2:], image_record["original_size"],
                                                    threshold=self.mask_threshold, want_logits=False)
            outputs.append({"masks": masks, "iou_predictions": iou_predictions, "low_res_logits": low_res_masks})
        return outputs

    def postprocess_masks_fused(self, masks, input_size, original_size, threshold=None, want_logits=True):
        """(logits or None, logits > threshold or None) in one kernel launch."""
        return ops.sam_postprocess_masks(masks.float(), self.image_encoder.img_size, tuple(input_size), tuple(original_size),
                                         threshold=threshold, want_logits=want_logits)

    def postprocess_masks(self, masks: torch.Tensor, input_size: Tuple[int, ...], original_size: Tuple[int, ...]) -> torch.Tensor:
        """sam.py:133-162: remove padding and upscale masks to the original image size (B x C x H x W logits)."""
        return self.postprocess_masks_fused(masks, input_size, original_size)[0]

    def preprocess(self, x: torch.Tensor) -> torch.Tensor:
        """sam.py:164-174: normalise pixel values and pad to a square input.  x: [..., 3, h, w] uint8 or float."""
        lead = x.shape[:-3]
        y = ops.sam_preprocess(x.reshape(-1, *x.shape[-3:]), self.image_encoder.img_size, self.pixel_mean.reshape(-1).float().contiguous(),
                               self.pixel_std.reshape(-1).float().contiguous())
        return y.reshape(*lead, *y.shape[-3:])
