"""Host-side prompt / image geometry of SamPredictor: mirror of segment_anything/segment_anything/utils/transforms.py
(ResizeLongestSide :16-102) — same names, arguments and arithmetic.

The uint8 image resize stays on the host as in the reference (torchvision's `resize(to_pil_image(x), size)` is PIL's bilinear
resize, called here directly); everything else is the (new / old) scale applied to x and y.
"""
from typing import Tuple

import numpy as np
import torch


class ResizeLongestSide:
    """Resizes images so the longest side equals `target_length`, and maps points / boxes into the resized frame."""

    def __init__(self, target_length: int) -> None:
        self.target_length = target_length

    @staticmethod
    def get_preprocess_shape(oldh: int, oldw: int, long_side_length: int) -> Tuple[int, int]:
        """transforms.py:93-102: (H, W) after the resize, rounded half up."""
        s = long_side_length * 1.0 / max(oldh, oldw)
        return int(oldh * s + 0.5), int(oldw * s + 0.5)

    def _xy_scale(self, original_size) -> Tuple[float, float]:
        nh, nw = self.get_preprocess_shape(original_size[0], original_size[1], self.target_length)
        return nw / original_size[1], nh / original_size[0]

    # ---- numpy (SamPredictor.set_image / predict)
    def apply_image(self, image: np.ndarray) -> np.ndarray:
        """:26-31: HxWxC uint8 -> resized uint8."""
        from PIL import Image
        nh, nw = self.get_preprocess_shape(image.shape[0], image.shape[1], self.target_length)
        return np.asarray(Image.fromarray(image).resize((nw, nh), Image.BILINEAR)).copy()

    def apply_coords(self, coords: np.ndarray, original_size: Tuple[int, ...]) -> np.ndarray:
        """:33-46: [..., 2] (x, y) in the original frame -> resized frame (float64)."""
        sx, sy = self._xy_scale(original_size)
        out = np.array(coords, dtype=float, copy=True)
        out[..., 0] *= sx
        out[..., 1] *= sy
        return out

    def apply_boxes(self, boxes: np.ndarray, original_size: Tuple[int, ...]) -> np.ndarray:
        """:48-54: Bx4 XYXY boxes."""
        return self.apply_coords(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)

    # ---- torch (SamPredictor.predict_torch callers, tools/tool.py:227)
    def apply_image_torch(self, image: torch.Tensor) -> torch.Tensor:
        """:56-67: BCHW float image, antialiased bilinear.  The target size is taken from the H, W axes (the reference reads axes 0, 1
        of the BCHW tensor, which only gives the intended size for HxWxC input); no AnyEdit caller uses this method."""
        size = self.get_preprocess_shape(image.shape[-2], image.shape[-1], self.target_length)
        return torch.nn.functional.interpolate(image, size, mode="bilinear", align_corners=False, antialias=True)

    def apply_coords_torch(self, coords: torch.Tensor, original_size: Tuple[int, ...]) -> torch.Tensor:
        """:69-82."""
        sx, sy = self._xy_scale(original_size)
        out = coords.detach().clone().to(torch.float)
        out[..., 0] = out[..., 0] * sx
        out[..., 1] = out[..., 1] * sy
        return out

    def apply_boxes_torch(self, boxes: torch.Tensor, original_size: Tuple[int, ...]) -> torch.Tensor:
        """:84-91."""
        return self.apply_coords_torch(boxes.reshape(-1, 2, 2), original_size).reshape(-1, 4)
