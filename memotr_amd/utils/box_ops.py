"""Box helpers (cxcywh / xyxy, IoU, GIoU); semantics of the reference's utils/box_ops.py.
``box_area`` is inlined: torchvision is not a dependency of this package."""
from __future__ import annotations

import torch


def box_area(boxes: torch.Tensor) -> torch.Tensor:
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_xyxy_to_cxcywh(boxes: torch.Tensor) -> torch.Tensor:
    x1, y1, x2, y2 = boxes.unbind(-1)
    return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), dim=-1)


_TO_XYXY = {}


def box_cxcywh_to_xyxy(boxes: torch.Tensor) -> torch.Tensor:
    """(cx - w/2, cy - h/2, cx + w/2, cy + h/2) as ONE product with a constant 4x4 matrix instead of unbind +
    4 x (mul, add/sub) + stack: 1 kernel instead of 9 forward (and 1 instead of ~14 backward), ~9 calls per frame
    in the launch-bound criterion / matcher.  Bit-identical to the elementwise form: every product is with 0, 1 or
    +-0.5 (exact) and each output has two non-zero terms, so there is a single rounding in either form."""
    reduced = boxes.dtype == torch.float32 and boxes.is_cuda and (
        torch.backends.cuda.matmul.allow_tf32 or torch.get_float32_matmul_precision() != "highest")
    if not boxes.is_floating_point() or boxes.dtype in (torch.float16, torch.bfloat16) or reduced:
        # (fp32 GEMMs in a reduced internal precision would round the 0.5 * w products: keep the elementwise form)
        cx, cy, w, h = boxes.unbind(-1)
        return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)
    key = (boxes.device, boxes.dtype)
    m = _TO_XYXY.get(key)
    if m is None:
        m = torch.tensor([[1, 0, 1, 0], [0, 1, 0, 1], [-0.5, 0, 0.5, 0], [0, -0.5, 0, 0.5]], dtype=boxes.dtype,
                         device=boxes.device)
        _TO_XYXY[key] = m
    with torch.autocast(device_type=boxes.device.type, enabled=False):      # never in reduced precision
        return boxes @ m


def box_cxcywh_to_xywh(boxes: torch.Tensor) -> torch.Tensor:
    cx, cy, w, h = boxes.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, w, h), dim=-1)


def box_iou_union(boxes1: torch.Tensor, boxes2: torch.Tensor):
    """Pairwise (N,M) IoU and union of xyxy boxes."""
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = box_area(boxes1)[:, None] + box_area(boxes2) - inter
    return inter / union, union


def generalized_box_iou(boxes1: torch.Tensor, boxes2: torch.Tensor) -> torch.Tensor:
    assert (boxes1[:, 2:] >= boxes1[:, :2]).all()
    assert (boxes2[:, 2:] >= boxes2[:, :2]).all()
    iou, union = box_iou_union(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull
