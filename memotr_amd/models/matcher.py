"""Hungarian matching of detect queries to ground-truth boxes (semantics of the reference's models/matcher.py).

Cost = cost_class * focal-style class cost + cost_bbox * L1(cxcywh) + cost_giou * (-GIoU), solved with
``scipy.optimize.linear_sum_assignment`` on the host.  Unlike the reference (one ``.cpu()`` per call, six calls
per frame: matcher.py:122), several cost matrices can be solved from ONE device->host copy
(``solve_many``) -- the per-frame synchronisations are what limits multi-GPU scaling.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn as nn
from scipy.optimize import linear_sum_assignment

from ..utils.box_ops import box_cxcywh_to_xyxy, generalized_box_iou


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class = cost_class
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou

    @torch.no_grad()
    def cost_matrix(self, pred_logits: torch.Tensor, pred_boxes: torch.Tensor, tgt_labels: torch.Tensor,
                    tgt_boxes: torch.Tensor, use_focal: bool = True) -> torch.Tensor:
        """(num_queries, num_targets) cost for one image."""
        if use_focal:
            prob = pred_logits.sigmoid()
            alpha, gamma = 0.25, 2.0
            neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
            pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
            cost_class = pos[:, tgt_labels] - neg[:, tgt_labels]
        else:
            cost_class = -pred_logits.softmax(-1)[:, tgt_labels]
        cost_bbox = torch.cdist(pred_boxes, tgt_boxes, p=1)
        cost_giou = -generalized_box_iou(box_cxcywh_to_xyxy(pred_boxes), box_cxcywh_to_xyxy(tgt_boxes))
        return self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou

    @torch.no_grad()
    def cost_matrix_stacked(self, pred_logits: torch.Tensor, pred_boxes: torch.Tensor, tgt_labels: torch.Tensor,
                            tgt_boxes: torch.Tensor) -> torch.Tensor:
        """Focal-style cost of several decoder layers at once: (n_layers, Q, K) / (n_layers, Q, 4) against (T,) /
        (T, 4) -> (n_layers, Q, T).  Elementwise the same arithmetic as ``cost_matrix`` per layer."""
        prob = pred_logits.sigmoid()
        alpha, gamma = 0.25, 2.0
        neg = (1 - alpha) * (prob ** gamma) * (-(1 - prob + 1e-8).log())
        pos = alpha * ((1 - prob) ** gamma) * (-(prob + 1e-8).log())
        cost_class = pos[..., tgt_labels] - neg[..., tgt_labels]
        cost_bbox = (pred_boxes[:, :, None, :] - tgt_boxes[None, None, :, :]).abs().sum(-1)
        a = box_cxcywh_to_xyxy(pred_boxes)[:, :, None, :]
        b = box_cxcywh_to_xyxy(tgt_boxes)[None, None, :, :]
        lt = torch.max(a[..., :2], b[..., :2])
        rb = torch.min(a[..., 2:], b[..., 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[..., 0] * wh[..., 1]
        union = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1]) + (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1]) - inter
        iou = inter / union
        wh_h = (torch.max(a[..., 2:], b[..., 2:]) - torch.min(a[..., :2], b[..., :2])).clamp(min=0)
        hull = wh_h[..., 0] * wh_h[..., 1]
        giou = iou - (hull - union) / hull
        return self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * (-giou)

    @staticmethod
    def solve(cost):
        """One (Q, T) assignment problem on the host (numpy array or CPU tensor)."""
        import numpy as np
        c = np.asarray(cost)
        if c.size == 0:
            return np.zeros((0,), dtype=np.int64), np.zeros((0,), dtype=np.int64)
        i, j = linear_sum_assignment(c)
        return i.astype(np.int64), j.astype(np.int64)

    @staticmethod
    def solve_many(costs: Sequence[torch.Tensor]) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """Solve every (Q_i, T_i) assignment problem with a single device->host transfer."""
        if not costs:
            return []
        flat = torch.cat([c.reshape(-1) for c in costs]).cpu()
        out, pos = [], 0
        for c in costs:
            n = c.numel()
            m = flat[pos:pos + n].view(c.shape)
            pos += n
            i, j = linear_sum_assignment(m) if n > 0 else ([], [])
            out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
        return out

    def forward(self, outputs, targets, use_focal=True):
        """Reference-compatible call: outputs {"pred_logits" (B,Q,K), "pred_boxes" (B,Q,4)}, targets = per-image
        objects with ``.labels``/``.boxes`` (or dicts).  Returns [(query_idx, target_idx)] per image."""
        get = (lambda t, k: t[k]) if isinstance(targets[0], dict) else getattr
        costs = [self.cost_matrix(outputs["pred_logits"][b], outputs["pred_boxes"][b], get(t, "labels"),
                                  get(t, "boxes"), use_focal) for b, t in enumerate(targets)]
        return self.solve_many(costs)


def build(config: dict) -> HungarianMatcher:
    return HungarianMatcher(cost_class=config["MATCH_COST_CLASS"], cost_bbox=config["MATCH_COST_BBOX"],
                            cost_giou=config["MATCH_COST_GIOU"])
