"""Mirror of segment_anything/segment_anything/predictor.py (SamPredictor :17-269) — SURVEY.md §8(f) N3.

Same methods, arguments, return values and error behaviour.  set_image / set_torch_image run the HIP image encoder once and keep the
embedding both as the reference's 1xCxHxW tensor (`features`, get_image_embedding) and as the channels-last bf16 rows the decoder
kernels consume; predict_torch is prompt kernels -> two-way transformer -> un-shuffled upscaling -> one fused
resize-crop-resize(-threshold) kernel.  AnyEdit's callers: tools/tool.py:182 (set_image) and :232-237 (predict_torch with boxes).
"""
from typing import Optional, Tuple

import numpy as np
import torch

from anyedit_amd import ops
from .modeling import Sam
from .utils.transforms import ResizeLongestSide


class SamPredictor:
    def __init__(self, sam_model: Sam) -> None:
        super().__init__()
        self.model = sam_model
        self.transform = ResizeLongestSide(sam_model.image_encoder.img_size)
        self.reset_image()

    def set_image(self, image: np.ndarray, image_format: str = "RGB") -> None:
        """:32-60: HWC uint8 image in [0, 255] -> embedding."""
        assert image_format in ["RGB", "BGR"], f"image_format must be in ['RGB', 'BGR'], is {image_format}."
        if image_format != self.model.image_format:
            image = image[..., ::-1]
        input_image = self.transform.apply_image(np.ascontiguousarray(image))
        input_image_torch = torch.as_tensor(input_image, device=self.device).permute(2, 0, 1).contiguous()[None, :, :, :]
        self.set_torch_image(input_image_torch, image.shape[:2])

    @torch.no_grad()
    def set_torch_image(self, transformed_image: torch.Tensor, original_image_size: Tuple[int, ...]) -> None:
        """:62-90: 1x3xHxW image already resized by ResizeLongestSide."""
        assert (
            len(transformed_image.shape) == 4
            and transformed_image.shape[1] == 3
            and max(*transformed_image.shape[2:]) == self.model.image_encoder.img_size
        ), f"set_torch_image input must be BCHW with long side {self.model.image_encoder.img_size}."
        self.reset_image()
        self.original_size = original_image_size
        self.input_size = tuple(transformed_image.shape[-2:])
        self.features = self.model.image_encoder(self.model.preprocess(transformed_image))
        self._feature_rows = ops.nchw_to_rows(self.features[:1].float().contiguous())
        self.is_image_set = True

    def predict(self, point_coords: Optional[np.ndarray] = None, point_labels: Optional[np.ndarray] = None,
                box: Optional[np.ndarray] = None, mask_input: Optional[np.ndarray] = None, multimask_output: bool = True,
                return_logits: bool = False) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """:92-166: numpy prompts in the ORIGINAL image frame -> (masks CxHxW, iou C, low-res logits Cx256x256)."""
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        coords_torch = labels_torch = box_torch = mask_input_torch = None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            pc = self.transform.apply_coords(point_coords, self.original_size)
            coords_torch = torch.as_tensor(pc, dtype=torch.float, device=self.device)[None, :, :]
            labels_torch = torch.as_tensor(point_labels, dtype=torch.int, device=self.device)[None, :]
        if box is not None:
            box_torch = torch.as_tensor(self.transform.apply_boxes(box, self.original_size), dtype=torch.float, device=self.device)[None, :]
        if mask_input is not None:
            mask_input_torch = torch.as_tensor(mask_input, dtype=torch.float, device=self.device)[None, :, :, :]
        masks, iou_predictions, low_res_masks = self.predict_torch(coords_torch, labels_torch, box_torch, mask_input_torch,
                                                                   multimask_output, return_logits=return_logits)
        return masks[0].detach().cpu().numpy(), iou_predictions[0].detach().cpu().numpy(), low_res_masks[0].detach().cpu().numpy()

    @torch.no_grad()
    def predict_torch(self, point_coords: Optional[torch.Tensor], point_labels: Optional[torch.Tensor],
                      boxes: Optional[torch.Tensor] = None, mask_input: Optional[torch.Tensor] = None, multimask_output: bool = True,
                      return_logits: bool = False, _merge: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """:168-245: batched torch prompts already in the resized frame -> (masks BxCxHxW bool or logits, iou BxC, low-res BxCx256x256)."""
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
        pe, md = self.model.prompt_encoder, self.model.mask_decoder
        points = (point_coords, point_labels) if point_coords is not None else None
        bs = pe._get_batch_size(points, boxes, mask_input)
        sparse = torch.empty((bs, 0, pe.embed_dim), device=self.device)
        if points is not None:
            sparse = torch.cat([sparse, pe._embed_points(point_coords, point_labels, pad=(boxes is None))], dim=1)
        if boxes is not None:
            sparse = torch.cat([sparse, pe._embed_boxes(boxes)], dim=1)
        h, w = pe.image_embedding_size
        if mask_input is not None:
            dense_vec, dense_rows = None, pe._embed_masks_rows(mask_input)
        else:
            dense_vec, dense_rows = pe.no_mask_embed.weight.detach().reshape(-1).to(torch.bfloat16).contiguous(), None
        masks_all, iou_all = md.predict_masks_rows(self._feature_rows, pe.pe_layer.grid_rows((h, w)), sparse, h, w,
                                                   dense_vec=dense_vec, dense_rows=dense_rows)
        sl = slice(1, None) if multimask_output else slice(0, 1)
        low_res_masks, iou_predictions = masks_all[:, sl, :, :], iou_all[:, sl]
        if _merge:
            return ops.sam_postprocess_masks(low_res_masks.contiguous(), self.model.image_encoder.img_size, self.input_size,
                                             self.original_size, threshold=self.model.mask_threshold, want_logits=False, merge=True)[1]
        logits, binary = self.model.postprocess_masks_fused(low_res_masks.contiguous(), self.input_size, self.original_size,
                                                            threshold=None if return_logits else self.model.mask_threshold,
                                                            want_logits=return_logits)
        return (logits if return_logits else binary), iou_predictions, low_res_masks

    def predict_torch_merged(self, boxes: torch.Tensor) -> torch.Tensor:
        """Union over all box prompts of the single-mask predictions, [1, 1, H, W] bool: what maskgeneration's mask_mode 'merge' does
        with `torch.sum(masks, dim=0) > 0` (tools/tool.py:239-241), with the union taken inside the post-processing kernel."""
        return self.predict_torch(None, None, boxes=boxes, multimask_output=False, _merge=True)

    def get_image_embedding(self) -> torch.Tensor:
        """:247-259."""
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) to generate an embedding.")
        assert self.features is not None, "Features must exist if an image has been set."
        return self.features

    @property
    def device(self) -> torch.device:
        return self.model.device

    def reset_image(self) -> None:
        """:265-272."""
        self.is_image_set = False
        self.features = None
        self._feature_rows = None
        self.orig_h = None
        self.orig_w = None
        self.input_h = None
        self.input_w = None
