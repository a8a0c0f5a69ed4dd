"""Mask generation for the local-edit pipelines: mirror of `maskgeneration`
(AnyEdit_Collection/adaptive_editing_pipelines/tools/tool.py:166-269) from the detector's output onwards — SURVEY.md §8(f) N3.

The reference runs GroundingDINO (`get_grounding_output`, tool.py:117-146), converts its boxes, filters them by the phrase they were
grounded to, de-duplicates them with torchvision NMS, prompts SAM with the surviving boxes and merges the masks.  Here the detector is
the boundary: `det_model(image_pil, det_prompt, box_threshold, text_threshold)` must return what get_grounding_output returns — boxes
[n, 4] as normalised (cx, cy, w, h) on the CPU and phrases of the form "name(0.87)".  Everything after that runs on the HIP SamPredictor:
NMS is ae_nms_sorted_f32, the prompt -> mask path is predict_torch, and mask_mode 'merge' is folded into the post-processing kernel.
Same arguments, return tuples and early-outs as the reference.
"""
import numpy as np
import torch
from PIL import Image

from anyedit_amd import ops


def load_image_512(image_path):
    """tool.py:169-174: file path or PIL image -> RGB, 512x512, Lanczos."""
    img = Image.open(image_path) if isinstance(image_path, str) else image_path
    return img.convert("RGB").resize((512, 512), resample=Image.Resampling.LANCZOS)


def boxes_to_pixels_xyxy(boxes_filt, W, H):
    """tool.py:184-188: normalised (cx, cy, w, h) -> pixel (x0, y0, x1, y1)."""
    b = boxes_filt.clone().float() * torch.tensor([W, H, W, H], dtype=torch.float32)
    b[:, :2] -= b[:, 2:] / 2
    b[:, 2:] += b[:, :2]
    return b


def _name(phrase):
    return phrase.split('(')[0]


def _score(phrase):
    return float(phrase.split('(')[1].strip(')'))


def select_target_boxes(boxes, pred_phrases, target_object):
    """tool.py:191-222: boxes whose phrase names the target (exact match first, then word overlap); a list of targets is tried in order.
    Returns (boxes [k, 4], scores [k]) or None when nothing matches."""
    targets = [target_object] if isinstance(target_object, str) else list(target_object)
    picked = [(box, _score(ph)) for obj in targets for box, ph in zip(boxes, pred_phrases) if _name(ph) == obj]
    for obj in targets:
        if picked:
            break
        picked = [(box, _score(ph)) for box, ph in zip(boxes, pred_phrases)
                  if _name(ph) in obj.split(' ') or obj.split(' ')[-1] in _name(ph).split(' ')]
    if not picked:
        return None
    return torch.stack([b for b, _ in picked]), torch.tensor([s for _, s in picked])


def maskgeneration(det_model, sam_model, image_path, det_prompt, mask_mode='max', box_threshold=0.25, text_threshold=0.25,
                   target_object=None, device="cuda"):
    """tool.py:166-269.  Returns (mask_pil, image_pil, mask_bbox_pil, union_region); (None, image_pil, None, None) when no box or an
    empty mask; for mask_mode 'count' (all masks kept): (masks bool [n,1,H,W], image_pil, None, union_region)."""
    image_pil = load_image_512(image_path)
    boxes_filt, pred_phrases = det_model(image_pil, det_prompt, box_threshold, text_threshold)
    image = np.array(image_pil)
    sam_model.set_image(image)
    W, H = image_pil.size
    boxes_filt = boxes_to_pixels_xyxy(boxes_filt.cpu(), W, H)
    if target_object is not None:
        sel = select_target_boxes(boxes_filt, pred_phrases, target_object)
        if sel is None:
            return None, image_pil, None, None
        boxes_filt, boxes_score = sel
        keep = ops.nms(boxes_filt.to(device), boxes_score.to(device), 0.5).cpu()        # torchvision.ops.nms (tool.py:224)
        boxes_filt = boxes_filt[keep]
    if len(boxes_filt) == 0:
        return None, image_pil, None, None
    transformed_boxes = sam_model.transform.apply_boxes_torch(boxes_filt, image.shape[:2]).to(device)
    union_region = ((boxes_filt[0][2] - boxes_filt[0][0]).item() / 512) * ((boxes_filt[0][3] - boxes_filt[0][1]).item() / 512)

    if mask_mode == 'merge':                       # union of all instance masks, computed inside the post-processing kernel
        masks = sam_model.predict_torch_merged(boxes=transformed_boxes)
    else:
        masks, _, _ = sam_model.predict_torch(point_coords=None, point_labels=None, boxes=transformed_boxes, multimask_output=False)
        if mask_mode == 'count':
            return masks, Image.fromarray(image), None, union_region
    mask = masks[0][0].cpu().numpy()
    if not mask.any():
        return None, image_pil, None, None

    box_mask = np.zeros((H, W), dtype=np.uint8)
    for bbox in (boxes_filt if mask_mode == 'merge' else boxes_filt[:1]):
        x0, y0, x1, y1 = (int(v) for v in bbox.tolist())
        box_mask[y0:y1, x0:x1] = 255
    return Image.fromarray(mask), Image.fromarray(image), Image.fromarray(box_mask), union_region
