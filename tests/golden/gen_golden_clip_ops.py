"""Golden vectors for the callers' small-tensor chains (include/clip_ops_hip.h) AT THE MODEL'S SIZES, produced by
importing the REFERENCE (build container only):

    python tests/golden/gen_golden_clip_ops.py        ->  tests/golden/clipops_C1_k{1,8}.npz

The reference's train-step golden (M6) pins those kernels only at toy geometry (20 queries, 6 ground truths, one class);
these fixtures pin them where the benchmark runs them: 6 decoder layers x 300 detect queries (+ carried tracks) against
17 ground truths, 1 class (DanceTrack / MOT17) and 8 classes (BDD100K).  Every value below is an output of a reference
function called on seeded inputs:

  * matching cost + assignment : ``models/matcher.py`` HungarianMatcher.forward (:79-131); the cost matrix is what it hands
    to scipy (captured by wrapping the ``linear_sum_assignment`` name the module imported, :14)
  * focal loss                 : ``models/criterion.py`` sigmoid_focal_loss (:442-467), value per layer and d/d logits
  * L1 + GIoU box losses       : ``models/criterion.py`` ClipCriterion.get_loss_box (:417-440), sums and d/d boxes
  * IoU of tracks with their GT: ``utils/box_ops.py`` box_iou_union (:49-60) as used at ``models/criterion.py:354-367``
  * inverse_sigmoid, box refinement ``sigmoid(delta + inverse_sigmoid(ref))`` (``utils/utils.py:61-74``,
    ``models/deformable_decoder.py:139-149``) and ``pos_to_pos_embed`` (``models/utils.py:78-85``) on 320 rows, with the
    gradients autograd gives the reference's formulation.
Only data is written; no reference source is copied.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden_model import install_stubs  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def boxes_like(g, n, spread=0.25):
    """cxcywh boxes as a detector produces them: centres anywhere, sides 2-25 % of the frame."""
    c = torch.rand(n, 2, generator=g) * 0.9 + 0.05
    wh = torch.rand(n, 2, generator=g) * spread + 0.02
    return torch.cat((c, wh), 1)


def case(K, seed):
    import models.matcher as ref_matcher
    from models.criterion import ClipCriterion, sigmoid_focal_loss
    from models.utils import pos_to_pos_embed
    from structures.track_instances import TrackInstances
    from utils.box_ops import box_cxcywh_to_xyxy, box_iou_union
    from utils.utils import inverse_sigmoid
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(seed)
    n_layers, nd, n_tr, T = 6, 300, 11, 17
    nq = nd + n_tr
    logits = torch.randn(n_layers, nq, K, generator=g) * 2.0 - 2.0
    gt_boxes = boxes_like(g, T)
    gt_labels = torch.randint(0, K, (T,), generator=g)
    boxes = boxes_like(g, n_layers * nq).view(n_layers, nq, 4)
    # some predictions sit on a ground truth (what a trained model produces): ties in the sub-gradients are exercised
    # by the exact copies, near hits by the jittered ones
    for l in range(n_layers):
        pick = torch.randperm(nd, generator=g)[:T]
        boxes[l, pick] = gt_boxes + (0.01 * torch.randn(T, 4, generator=g) if l % 2 else 0.0)
    boxes = boxes.clamp(1e-3, 1 - 1e-3)
    out = {"logits": logits, "boxes": boxes, "gt_boxes": gt_boxes, "gt_labels": gt_labels}

    # ---- matching cost + assignment, per layer, detect queries only (reference models/criterion.py:150-165, 252-262)
    seen = []
    orig = ref_matcher.linear_sum_assignment

    def spy(c):
        seen.append(np.array(c, dtype=np.float32, copy=True))
        return orig(c)

    ref_matcher.linear_sum_assignment = spy
    try:
        m = ref_matcher.HungarianMatcher(cost_class=2, cost_bbox=5, cost_giou=2)
        idx = [m(outputs={"pred_logits": logits[l, :nd][None], "pred_boxes": boxes[l, :nd][None]},
                 targets=[{"labels": gt_labels, "boxes": gt_boxes}], use_focal=True)[0] for l in range(n_layers)]
    finally:
        ref_matcher.linear_sum_assignment = orig
    out["cost"] = np.stack(seen)                                           # (n_layers, nd, T)
    out["match_q"] = torch.stack([i for i, _ in idx])
    out["match_g"] = torch.stack([j for _, j in idx])

    # ---- focal loss per layer over all queries: matched detect queries carry their ground truth's label, the carried
    #      tracks the label of the ground truth they own (every third owns none), the rest background (K)
    owner = torch.randint(0, T, (n_tr,), generator=g)
    owner[::3] = -1
    labels = torch.full((n_layers, nq), K, dtype=torch.int64)
    for l in range(n_layers):
        labels[l, idx[l][0]] = gt_labels[idx[l][1]]
        if l >= 1:
            has = owner >= 0
            labels[l, nd:][has] = gt_labels[owner[has]]
    out["labels"] = labels
    out["track_owner"] = owner
    x = logits.clone().requires_grad_(True)
    per_layer = torch.stack([sigmoid_focal_loss(inputs=x[l], targets=F.one_hot(labels[l], K + 1)[:, :-1].to(x.dtype),
                                                alpha=0.25, gamma=2) for l in range(n_layers)])
    w = torch.linspace(0.5, 1.5, n_layers)
    (per_layer * w).sum().backward()
    out["focal_per_layer"] = per_layer.detach()
    out["focal_weights"] = w
    out["focal_grad_logits"] = x.grad

    # ---- box losses of the matched pairs, per layer (get_loss_box is a staticmethod: sums over one layer's pairs)
    bx = boxes.clone().requires_grad_(True)
    l1s, gis = [], []
    for l in range(n_layers):
        gt = TrackInstances()
        gt.boxes = gt_boxes
        l1, gi = ClipCriterion.get_loss_box(outputs={"pred_bboxes": [bx[l]]}, gt_trackinstances=[gt],
                                            idx_to_gts_idx=[(idx[l][0], idx[l][1])])
        l1s.append(l1)
        gis.append(gi)
    l1s, gis = torch.stack(l1s), torch.stack(gis)
    ((l1s * w).sum() + (gis * w.flip(0)).sum()).backward()
    out["box_l1_per_layer"], out["box_giou_per_layer"] = l1s.detach(), gis.detach()
    out["box_grad"] = bx.grad

    # ---- IoU of the carried tracks with the ground truth they own (criterion.py:354-367)
    tr_boxes = boxes[-1, nd:]
    has = owner >= 0
    out["track_iou"] = torch.diag(box_iou_union(box_cxcywh_to_xyxy(tr_boxes[has]),
                                                box_cxcywh_to_xyxy(gt_boxes[owner[has]]))[0])

    # ---- inverse_sigmoid / box refinement / sine embedding on a decoder-sized tensor, with gradients
    ref = torch.rand(320, 4, generator=g)
    ref[:4] = torch.tensor([[0.0, 1.0, 1e-6, 1 - 1e-6]]).t().expand(4, 4)      # the clamps' edges
    delta = torch.randn(320, 4, generator=g)
    r = ref.clone().requires_grad_(True)
    d = delta.clone().requires_grad_(True)
    inv = inverse_sigmoid(r)
    refined = (d + inv).sigmoid()
    cot = torch.randn(320, 4, generator=g)
    (refined * cot).sum().backward()
    out.update(ref=ref, delta=delta, inv_sigmoid=inv.detach(), refined=refined.detach(), refine_cot=cot,
               refine_grad_delta=d.grad, refine_grad_ref=r.grad)
    p = torch.rand(320, 4, generator=g).requires_grad_(True)
    emb = pos_to_pos_embed(p, 128)
    cot_e = torch.randn(320, 512, generator=g)
    (emb * cot_e).sum().backward()
    out.update(sine_pos=p.detach(), sine_embed=emb.detach(), sine_cot=cot_e, sine_grad=p.grad)
    np.savez_compressed(os.path.join(OUT, f"clipops_C1_k{K}.npz"),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print(f"clipops_C1_k{K}: cost {out['cost'].shape}, focal {per_layer.tolist()[:2]}, l1 {l1s.tolist()[:2]}")


def main():
    install_stubs()
    torch.set_grad_enabled(True)
    case(1, 101)
    case(8, 108)


if __name__ == "__main__":
    main()
