"""The callers' small-tensor chains against vectors the REFERENCE produced at the model's sizes
(tests/golden/clipops_C1_k{1,8}.npz, made by tests/golden/gen_golden_clip_ops.py: 6 decoder layers x 300 detect queries +
11 carried tracks against 17 ground truths; 1 class = DanceTrack / MOT17, 8 classes = BDD100K).

On the CPU the package's torch formulations are held to them (what the golden-model tests run through); on the GPU
(`-m gpu`) the HIP kernels of include/clip_ops_hip.h are -- so the kernels are pinned to the reference directly, not
only through this package's own statement of it (round-5 verdict, weak point 4)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]
CASES = [1, 8]


def load(K, device):
    if device == "cuda" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = np.load(os.path.join(HERE, "golden", f"clipops_C1_k{K}.npz"))
    return {k: torch.from_numpy(g[k]).to(device) for k in g.files}


def close(got, want, rtol, atol, what=""):
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.cpu().numpy(), rtol=rtol, atol=atol, err_msg=what)


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("K", CASES)
def test_matching_cost_and_assignment(K, device):
    """models/matcher.py:79-131 of the reference: the cost matrix it hands to scipy and the pairs scipy returns."""
    from memotr_amd.functions import clip_ops
    from memotr_amd.models.matcher import HungarianMatcher
    g = load(K, device)
    nd = g["cost"].shape[1]
    m = HungarianMatcher(cost_class=2, cost_bbox=5, cost_giou=2)
    lg, bx = g["logits"][:, :nd], g["boxes"][:, :nd]
    if device == "cuda":
        assert clip_ops.fused(lg, bx, g["gt_boxes"])
        cost = clip_ops.match_cost(lg, bx, g["gt_labels"], g["gt_boxes"], 2, 5, 2)
    else:
        cost = m.cost_matrix_stacked(lg, bx, g["gt_labels"], g["gt_boxes"])
    close(cost, g["cost"], 2e-5, 2e-5, "cost")
    for l in range(cost.shape[0]):                       # the assignment on the cost THIS path produced
        qi, gj = m.solve(cost[l].cpu().numpy())
        assert np.array_equal(np.asarray(qi), g["match_q"][l].cpu().numpy()), l
        assert np.array_equal(np.asarray(gj), g["match_g"][l].cpu().numpy()), l


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("K", CASES)
def test_focal_loss_per_layer_and_its_gradient(K, device):
    """models/criterion.py:442-467 (sigmoid_focal_loss), one call per decoder layer in the reference."""
    from memotr_amd.functions import clip_ops
    g = load(K, device)
    x = g["logits"].clone().requires_grad_(True)
    if device == "cuda":
        per_layer = clip_ops.focal_loss_per_layer(x, g["labels"])
    else:
        per_layer = clip_ops.focal_loss_per_layer_reference(x, g["labels"])
    close(per_layer, g["focal_per_layer"], 2e-5, 1e-4, "loss")
    (per_layer * g["focal_weights"]).sum().backward()
    close(x.grad, g["focal_grad_logits"], 1e-4, 1e-6, "d/d logits")


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("K", CASES)
def test_focal_labels_of_all_layers(K, device):
    """The class targets the reference assembles per layer (models/criterion.py:300-330): background, matched detect
    queries, and -- from the first merged layer on -- the carried tracks' ground truths."""
    from memotr_amd.functions import clip_ops
    g = load(K, device)
    n_layers, nq = g["labels"].shape
    n_tr = g["track_owner"].shape[0]
    nd = nq - n_tr
    lay = torch.arange(n_layers, device=device).repeat_interleave(g["match_q"].shape[1])
    q, t = g["match_q"].reshape(-1), g["match_g"].reshape(-1)
    late = torch.tensor([l >= 1 for l in range(n_layers)], device=device)
    fn = clip_ops.focal_labels if device == "cuda" else clip_ops.focal_labels_reference
    got = fn(lay, q, t, g["gt_labels"], g["track_owner"], late, nd, n_tr, K)
    assert torch.equal(got, g["labels"])


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("K", CASES)
def test_box_losses_of_the_matched_pairs_and_their_gradient(K, device):
    """models/criterion.py:417-440 (get_loss_box): L1 and 1 - GIoU summed over a layer's pairs; exact hits included (the
    sub-gradient rules of abs / max / min / clamp at ties)."""
    from memotr_amd.functions import clip_ops
    g = load(K, device)
    n_layers = g["boxes"].shape[0]
    n_pairs = g["match_q"].shape[1]
    bx = g["boxes"].clone()[:, None].contiguous().requires_grad_(True)            # (n_layers, B = 1, Nq, 4)
    lay = torch.arange(n_layers, device=device).repeat_interleave(n_pairs)
    q, t = g["match_q"].reshape(-1), g["match_g"].reshape(-1)
    fn = clip_ops.pair_box_loss if device == "cuda" else clip_ops.pair_box_loss_reference
    l1, gi = fn(bx, lay, q, 0, g["gt_boxes"], t)
    l1s, gis = l1.view(n_layers, n_pairs).sum(1), gi.view(n_layers, n_pairs).sum(1)
    close(l1s, g["box_l1_per_layer"], 2e-5, 1e-5, "l1")
    close(gis, g["box_giou_per_layer"], 2e-5, 1e-5, "giou")
    w = g["focal_weights"]
    ((l1s * w).sum() + (gis * w.flip(0)).sum()).backward()
    close(bx.grad[:, 0], g["box_grad"], 2e-4, 2e-5, "d/d boxes")


@pytest.mark.parametrize("device", DEVICES)
@pytest.mark.parametrize("K", CASES)
def test_iou_of_tracks_with_their_ground_truth(K, device):
    """utils/box_ops.py:49-60 as used at models/criterion.py:354-367."""
    from memotr_amd.functions import clip_ops
    g = load(K, device)
    n_tr = g["track_owner"].shape[0]
    has = g["track_owner"] >= 0
    boxes = g["boxes"][-1, -n_tr:][has]
    fn = clip_ops.pair_iou if device == "cuda" else clip_ops.pair_iou_reference
    close(fn(boxes, g["gt_boxes"], g["track_owner"][has]), g["track_iou"], 1e-5, 1e-6)


@pytest.mark.parametrize("device", DEVICES)
def test_inverse_sigmoid_box_refinement_and_sine_embedding(device):
    """utils/utils.py:61-74, models/deformable_decoder.py:139-149, models/utils.py:78-85 on 320 rows, with gradients."""
    from memotr_amd.models.utils import pos_to_pos_embed
    from memotr_amd.utils.utils import inverse_sigmoid, refine_boxes
    g = load(1, device)
    close(inverse_sigmoid(g["ref"]), g["inv_sigmoid"], 1e-5, 1e-5, "inverse_sigmoid")
    r = g["ref"].clone().requires_grad_(True)
    d = g["delta"].clone().requires_grad_(True)
    out = refine_boxes(d, r)
    close(out, g["refined"], 1e-5, 1e-6, "refined")
    (out * g["refine_cot"]).sum().backward()
    close(d.grad, g["refine_grad_delta"], 1e-4, 1e-6, "d/d delta")
    # (rows 0-3 sit on the clamps' edges: 0, 1, 1e-6, 1 - 1e-6 -- where the reference's gradient is 0 or huge)
    close(r.grad, g["refine_grad_ref"], 1e-4, 1e-5, "d/d ref")
    p = g["sine_pos"].clone().requires_grad_(True)
    e = pos_to_pos_embed(p, 128)
    close(e, g["sine_embed"], 1e-5, 2e-6, "sine embedding")
    (e * g["sine_cot"]).sum().backward()
    close(p.grad, g["sine_grad"], 2e-4, 2e-3, "d/d pos")
