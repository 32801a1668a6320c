"""CPU: container quirks callers can observe, and criterion edge cases the golden clip does not reach."""
import numpy as np
import pytest
import torch

from model_helpers import small_config


def test_track_instances_reference_quirks():
    """structures/track_instances.py:40-41,58-59,89: `to` / `__getitem__` / `cat` rebuild with default meta."""
    from memotr_amd.structures.track_instances import TrackInstances
    t = TrackInstances(frame_height=0.9, frame_width=0.8, hidden_dim=64, num_classes=8, use_dab=True)
    assert t.query_embed.shape == (0, 64) and t.ref_pts.shape == (0, 4) and t.logits.shape == (0, 8)
    assert TrackInstances(hidden_dim=64).query_embed.shape == (0, 128)          # non-DAB: 2C
    t.ref_pts, t.query_embed, t.ids = torch.rand(3, 4), torch.rand(3, 64), torch.tensor([5, 6, 7])
    t.boxes, t.labels, t.logits = torch.rand(3, 4), torch.zeros(3, dtype=torch.long), torch.rand(3, 8)
    assert len(t) == 3
    moved = t.to("cpu")
    assert moved.use_dab is True and moved.num_classes == 8 and moved is not t    # attrs are copied over after the rebuild
    one = t[1]
    assert len(one) == 1 and int(one.ids[0]) == 6 and one.scores.shape == (0,)  # empty fields stay empty
    with pytest.raises(IndexError):
        t[3]
    sel = t[torch.tensor([True, False, True])]
    assert sel.ids.tolist() == [5, 7]
    other = t[torch.tensor([0, 1])]
    # cat copies tensor fields only and takes the default meta (hidden_dim 256, num_classes 1, use_dab False);
    # fields that are empty on both sides keep their (0, default) shapes
    u = TrackInstances.cat_tracked_instances(t, other)
    assert len(u) == 5 and u.num_classes == 1 and u.hidden_dim == 256 and u.use_dab is False
    meta, tensors = TrackInstances.tracks_to_meta_tensors([t, other])
    back = TrackInstances.meta_tensors_to_tracks(meta, tensors)
    assert len(back) == 2 and torch.equal(back[0].ids, t.ids) and back[1].hidden_dim == 64
    init = TrackInstances.init_tracks({"imgs": [[torch.zeros(3, 40, 60)], [torch.zeros(3, 50, 30)]]}, hidden_dim=64,
                                      num_classes=1, use_dab=True)
    assert init[0].frame_height == pytest.approx(0.8) and init[1].frame_width == pytest.approx(0.5)


def test_cat_packed_equals_the_per_field_concatenation():
    """``TrackInstances.cat_packed``: same fields as ``cat_tracked_instances`` (values, shapes, gradients), the float
    fields as column views of one tensor; ``[index]`` keeps them packed with one gather; assigning a field ends the
    arrangement for that container; parts that cannot be packed take the per-field path."""
    from memotr_amd.structures.track_instances import TrackInstances

    def make(n, seed, C=16, K=1):
        g = torch.Generator().manual_seed(seed)
        t = TrackInstances(hidden_dim=C, num_classes=K, use_dab=True)
        t.ref_pts, t.boxes = torch.rand(n, 4, generator=g), torch.rand(n, 4, generator=g)
        t.query_embed = torch.randn(n, C, generator=g, requires_grad=True)
        t.output_embed = torch.randn(n, C, generator=g, requires_grad=True)
        t.last_output, t.long_memory = torch.randn(n, C, generator=g), torch.randn(n, C, generator=g)
        t.logits, t.iou = torch.randn(n, K, generator=g), torch.rand(n, generator=g)
        t.ids, t.matched_idx = torch.arange(n) + 10 * seed, torch.arange(n)
        return t

    a, b, c = make(3, 1), make(0, 2), make(5, 3)
    ref = TrackInstances.cat_tracked_instances(a, b, c)
    got = TrackInstances.cat_packed(a, b, c)
    assert got._packed_base() is not None and len(got) == len(ref) == 8
    for k in ("ref_pts", "boxes", "query_embed", "output_embed", "last_output", "long_memory", "logits", "iou", "ids",
              "matched_idx", "labels", "scores"):
        assert getattr(got, k).shape == getattr(ref, k).shape, k
        assert torch.equal(getattr(got, k), getattr(ref, k)), k
    idx = torch.tensor([7, 0, 4])
    sel_ref, sel = ref[idx], got[idx]
    assert sel._packed_base() is not None and sel._packed_base()[0].shape[0] == 3
    for k in ("query_embed", "output_embed", "iou", "logits", "ids"):
        assert torch.equal(getattr(sel, k), getattr(sel_ref, k)), k
    assert torch.equal(got[torch.tensor([True] * 3 + [False] * 5)].ids, a.ids)          # boolean masks too
    (sel.query_embed.sum() * 2 + sel.output_embed.square().sum()).backward()
    ga, gc = a.query_embed.grad.clone(), c.output_embed.grad.clone()
    a.query_embed.grad = c.output_embed.grad = a.output_embed.grad = c.query_embed.grad = None
    (sel_ref.query_embed.sum() * 2 + sel_ref.output_embed.square().sum()).backward()
    assert torch.equal(ga, a.query_embed.grad) and torch.allclose(gc, c.output_embed.grad)
    sel.ref_pts = torch.zeros(3, 4)                      # a field assigned: no longer one tensor
    assert sel._packed_base() is None and sel[torch.tensor([1])].ref_pts.shape == (1, 4)
    c.iou = torch.zeros(5, dtype=torch.float64)          # not packable -> the per-field concatenation
    assert TrackInstances.cat_packed(a, c)._packed_base() is None


def fake_outputs(n_det, n_tr, hidden, n_layers=2, K=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    nq = n_det + n_tr

    def layer():
        return {"pred_logits": torch.randn(1, nq, K, generator=g, requires_grad=True),
                "pred_bboxes": (torch.rand(1, nq, 4, generator=g) * 0.4 + 0.2).requires_grad_(True),
                "query_mask": torch.zeros(1, nq, dtype=torch.bool),
                "queries": torch.randn(1, nq, hidden, generator=g)}
    out = layer()
    out.update(last_ref_pts=torch.randn(1, nq, 4, generator=g), init_ref_pts=torch.randn(1, nq, 4, generator=g),
               det_query_embed=torch.randn(n_det, hidden, generator=g), outputs=torch.randn(1, nq, hidden, generator=g),
               aux_outputs=[layer() for _ in range(n_layers - 1)])
    return out


def build_criterion():
    from memotr_amd.models.criterion import build
    cfg = small_config()
    cfg.update(MATCH_COST_CLASS=2, MATCH_COST_BBOX=5, MATCH_COST_GIOU=2, LOSS_WEIGHT_FOCAL=2, LOSS_WEIGHT_L1=5,
               LOSS_WEIGHT_GIOU=2, AUX_LOSS_WEIGHT=[1.0], SAMPLE_LENGTHS=[2, 3])
    return build(cfg)


def clip(n_gt, ids=None):
    boxes = torch.rand(n_gt, 4) * 0.3 + 0.3
    info = {"ids": torch.arange(n_gt) if ids is None else torch.as_tensor(ids), "labels": torch.zeros(n_gt, dtype=torch.long),
            "boxes": boxes}
    return {"imgs": [[torch.zeros(3, 32, 32)]], "infos": [[info]]}


def test_criterion_frame_without_ground_truth():
    from memotr_amd.structures.track_instances import TrackInstances
    crit = build_criterion()
    crit.init_a_clip(clip(0), hidden_dim=64, num_classes=1, device=torch.device("cpu"))
    tracks = [TrackInstances(hidden_dim=64, num_classes=1, use_dab=True)]
    prev, new, unm = crit.process_single_frame(fake_outputs(20, 0, 64), tracks, 0)
    assert len(new[0]) == 0 and len(unm[0]) == 20 and len(prev[0]) == 0
    loss, _ = crit.get_mean_by_n_gts()
    assert float(loss["box_l1_loss"]) == 0.0 and float(loss["label_focal_loss"]) > 0.0   # background-only focal term
    crit.get_sum_loss_dict(loss).backward()


def test_criterion_tracks_own_all_ground_truths_and_lost_identity():
    """Every GT already owned by a track -> nothing to match; a track whose identity left the scene gets
    matched_idx -1, background label, no box loss, and keeps its previous IoU (criterion.py:166-194,337-349)."""
    from memotr_amd.structures.track_instances import TrackInstances
    torch.manual_seed(0)
    crit = build_criterion()
    crit.init_a_clip(clip(3, ids=[10, 11, 12]), hidden_dim=64, num_classes=1, device=torch.device("cpu"))
    tr = TrackInstances(hidden_dim=64, num_classes=1, use_dab=True)
    n = 4
    tr.ref_pts, tr.query_embed = torch.randn(n, 4), torch.randn(n, 64)
    tr.ids = torch.tensor([12, 99, 10, 11])                 # 99 is gone
    tr.boxes, tr.logits, tr.output_embed = torch.rand(n, 4), torch.randn(n, 1), torch.randn(n, 64)
    tr.iou = torch.full((n,), 0.25)
    tr.last_output, tr.long_memory = torch.randn(n, 64), torch.randn(n, 64)
    out = fake_outputs(20, n, 64, seed=1)
    prev, new, unm = crit.process_single_frame(out, [tr], 0)
    assert prev[0].matched_idx.tolist() == [2, -1, 0, 1]
    assert len(new[0]) == 0 and len(unm[0]) == 20
    assert float(prev[0].iou[1]) == pytest.approx(0.25) and not torch.equal(prev[0].iou, torch.full((n,), 0.25))
    assert torch.equal(prev[0].boxes, out["pred_bboxes"][0, 20:])
    loss, _ = crit.get_mean_by_n_gts()
    total = crit.get_sum_loss_dict(loss)
    total.backward()
    g = out["pred_bboxes"].grad[0]
    assert not g[:20].any() and not g[21].any() and g[20].any() and g[22].any()   # only owned tracks get box gradients
    # early aux layer (index 0 < MERGE_DET_TRACK_LAYER): detect queries are matched against ALL ground truths
    ga = out["aux_outputs"][0]["pred_bboxes"].grad[0]
    assert int((ga[:20].abs().sum(1) > 0).sum()) == 3 and not ga[20:].any()


def test_matcher_reference_call_signature():
    from memotr_amd.models.matcher import HungarianMatcher
    m = HungarianMatcher(2, 5, 2)
    torch.manual_seed(0)
    outputs = {"pred_logits": torch.randn(2, 6, 1), "pred_boxes": torch.rand(2, 6, 4) * 0.3 + 0.3}
    targets = [{"labels": torch.zeros(3, dtype=torch.long), "boxes": torch.rand(3, 4) * 0.3 + 0.3},
               {"labels": torch.zeros(0, dtype=torch.long), "boxes": torch.zeros(0, 4)}]
    res = m(outputs, targets)
    assert len(res) == 2 and res[0][0].shape == (3,) and res[1][0].shape == (0,)
    assert sorted(res[0][1].tolist()) == [0, 1, 2]
    # stacked cost == per-layer cost
    c1 = m.cost_matrix(outputs["pred_logits"][0], outputs["pred_boxes"][0], targets[0]["labels"], targets[0]["boxes"])
    cs = m.cost_matrix_stacked(outputs["pred_logits"][:1], outputs["pred_boxes"][:1], targets[0]["labels"],
                               targets[0]["boxes"])
    np.testing.assert_allclose(cs[0].numpy(), c1.numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(AssertionError):
        HungarianMatcher(0, 0, 0)


def _xyxy_elementwise(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_box_conversion_as_matrix_product_is_bit_identical(device):
    """utils/box_ops.py:21-25 of the reference is elementwise; the product with the constant 4x4 matrix must
    give the same bits (products with 0 / 1 / 0.5 are exact, one rounding per output) and the same gradient."""
    from memotr_amd.utils.box_ops import box_cxcywh_to_xyxy
    g = torch.Generator().manual_seed(5)
    for shape in ((7, 4), (6, 300, 4), (6, 1, 1, 4), (0, 4), (100000, 4)):
        x = (torch.randn(shape, generator=g) * 10.0).to(device).requires_grad_(True)
        got, want = box_cxcywh_to_xyxy(x), _xyxy_elementwise(x)
        assert torch.equal(got, want), shape
        if x.numel():
            seed = torch.randn(shape, generator=g).to(device)
            (ga,) = torch.autograd.grad(got, x, seed)
            (gb,) = torch.autograd.grad(want, x, seed)
            assert torch.equal(ga, gb), shape
    sig = torch.rand((50, 4), generator=g).to(device)                    # sigmoid-range boxes, as in the criterion
    assert torch.equal(box_cxcywh_to_xyxy(sig), _xyxy_elementwise(sig))


def test_ground_truth_ownership_takes_the_last_duplicate_id():
    """reference criterion.py:166-170 builds ``gt_ids_to_idx`` with a dict comprehension: when a frame repeats an
    id the LAST ground truth wins.  The tensor form must agree."""
    ids_tr = torch.tensor([5, 9, 7, 3])
    ids_gt = torch.tensor([7, 5, 7, 1, 5])
    want = []
    lut = {int(v): i for i, v in enumerate(ids_gt)}
    for v in ids_tr.tolist():
        want.append(lut.get(v, -1))
    eq = ids_tr[:, None] == ids_gt[None, :]
    got = (eq * torch.arange(1, len(ids_gt) + 1)).amax(1) - 1
    assert got.tolist() == want == [4, -1, 2, -1]
