"""CPU: host-side model path against golden vectors produced by the reference (tests/golden/gen_golden_model.py).

The HIP operator cannot run here, so these tests inject the oracle's torch statement of the operator into
``MSDeformAttn`` (tests may; the product never does) and pin everything around it: sine embeddings,
reference points, the module's projection / softmax / location arithmetic, encoder, decoder (DAB anchors,
box refinement, detect/track split, key-padding mask), heads, query updater, runtime tracker and the
two-frame MeMOTR loop.  Tolerance: 2e-5 abs on O(1) float32 data (GEMM summation-order noise only).
"""
import numpy as np
import pytest
import torch

from model_helpers import (TinyBackbone, assert_tracks_close, load_model_golden, patch_operator, small_config,
                           state_from, t, tracks_from)

TOL = dict(rtol=1e-4, atol=2e-5)


def test_small_functions_match_reference():
    from memotr_amd.models.deformable_encoder import DeformableEncoder
    from memotr_amd.models.deformable_transformer import DeformableTransformer
    from memotr_amd.models.position_embedding import build as build_pe
    from memotr_amd.models.utils import pos_to_pos_embed
    from memotr_amd.utils.nested_tensor import NestedTensor, tensor_list_to_nested_tensor
    from memotr_amd.utils.utils import inverse_sigmoid
    g = load_model_golden("M1_small_functions")
    # padding to /32 and mask (utils/nested_tensor.py:41-60)
    nt = tensor_list_to_nested_tensor([t(g["pad_tensors"])[0, :, :50, :70], t(g["pad_tensors"])[1, :, :64, :61]])
    assert np.array_equal(nt.tensors.numpy(), g["pad_tensors"]) and np.array_equal(nt.masks.numpy(), g["pad_masks"])
    pe = build_pe({"HIDDEN_DIM": 64})(NestedTensor(torch.zeros(2, 1, 8, 9), t(g["masks"])))
    np.testing.assert_allclose(pe.numpy(), g["pe"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pos_to_pos_embed(t(g["boxes"]), 32).numpy(), g["box_embed"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pos_to_pos_embed(t(g["boxes"])[:, :2], 16, temperature=20).numpy(),
                               g["box_embed_t20"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(inverse_sigmoid(t(g["inv_sig_in"])).numpy(), g["inv_sig_out"])
    masks = [torch.nn.functional.interpolate(t(g["pad_masks"])[None].float(), size=(int(h), int(w))).to(torch.bool)[0]
             for h, w in g["shapes"]]
    vr = torch.stack([DeformableTransformer.get_valid_ratio(m) for m in masks], 1)
    assert np.array_equal(vr.numpy(), g["valid_ratios"])
    for shapes in (t(g["shapes"]), [(int(h), int(w)) for h, w in g["shapes"]]):
        ref = DeformableEncoder.get_reference_points(shapes, vr, device="cpu")
        np.testing.assert_allclose(ref.numpy(), g["reference_points"], rtol=1e-6, atol=1e-7)


def test_msdeform_module_matches_reference(monkeypatch):
    from memotr_amd.modules import MSDeformAttn
    patch_operator(monkeypatch)
    g = load_model_golden("M2_msdeform_module")
    mod = MSDeformAttn(d_model=64, n_levels=3, n_heads=8, n_points=4)
    mod.load_state_dict(state_from(g))
    for tag in ("ref2", "ref4"):
        query = t(g[f"{tag}_query"]).requires_grad_(True)
        out = mod(query, t(g[f"{tag}_ref"]), t(g["src"]), t(g["shapes"]), t(g["level_start"]), t(g["mask"]))
        np.testing.assert_allclose(out.detach().numpy(), g[f"{tag}_out"], **TOL)
        (gq,) = torch.autograd.grad(out, query, t(g[f"{tag}_grad_out"]))
        np.testing.assert_allclose(gq.numpy(), g[f"{tag}_grad_query"], rtol=1e-4, atol=1e-4)
    with pytest.raises(ValueError):
        mod(query, torch.zeros(2, 11, 3, 3), t(g["src"]), t(g["shapes"]), t(g["level_start"]))


def test_module_state_dict_names_and_init():
    """Parameter names / shapes of SURVEY.md appendix C and the star-shaped offset bias (reference :72-86)."""
    from memotr_amd.modules import MSDeformAttn
    mod = MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    assert shapes == {
        "sampling_offsets.weight": (256, 256), "sampling_offsets.bias": (256,),
        "attention_weights.weight": (128, 256), "attention_weights.bias": (128,),
        "value_proj.weight": (256, 256), "value_proj.bias": (256,),
        "output_proj.weight": (256, 256), "output_proj.bias": (256,)}
    bias = mod.sampling_offsets.bias.view(8, 4, 4, 2)
    assert torch.allclose(bias[0, :, :, 0], torch.tensor([1.0, 2, 3, 4]).expand(4, 4))   # head 0 points along +x
    assert torch.allclose(bias[2, 0, :, 1], torch.tensor([1.0, 2, 3, 4]), atol=1e-6)      # head 2 along +y
    assert not mod.sampling_offsets.weight.any() and not mod.attention_weights.weight.any()


def build_transformer(g):
    from memotr_amd.models.deformable_transformer import build
    from memotr_amd.models.mlp import MLP
    from memotr_amd.models.utils import get_clones
    tr = build(small_config())
    bbox = get_clones(MLP(64, 64, 4, 3), 2)
    tr.set_refine_bbox_embed(bbox)
    tr.load_state_dict(state_from(g))          # includes the aliased decoder.bbox_embed.* entries
    for k, v in state_from(g, "b::").items():
        assert torch.equal(bbox.state_dict()[k], v)
    return tr


@pytest.mark.parametrize("use_checkpoint,level", [(False, 2), (True, 1), (True, 2)])
def test_transformer_matches_reference(monkeypatch, use_checkpoint, level):
    patch_operator(monkeypatch)
    g = load_model_golden("M3_transformer")
    tr = build_transformer(g)
    tr.use_checkpoint = use_checkpoint
    tr.checkpoint_level = level
    tr.encoder.use_checkpoint = use_checkpoint and level == 1
    tr.decoder.use_checkpoint = use_checkpoint
    srcs = [t(g[f"src{i}"]) for i in range(4)]
    masks = [t(g[f"mask{i}"]) for i in range(4)]
    poss = [t(g[f"pos{i}"]) for i in range(4)]
    query = t(g["query"]).requires_grad_(True)
    out, init_ref, inter_ref, inter_q = tr(srcs, masks, poss, query, t(g["ref"]), t(g["qmask"]))
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **TOL)
    np.testing.assert_allclose(init_ref.detach().numpy(), g["init_ref"], **TOL)
    np.testing.assert_allclose(inter_ref.detach().numpy(), g["inter_ref"], **TOL)
    np.testing.assert_allclose(inter_q.detach().numpy(), g["inter_q"], **TOL)
    loss = (out[-1] * torch.linspace(-1, 1, 64)).sum() + inter_ref[-1].sum()
    (gq,) = torch.autograd.grad(loss, query)
    np.testing.assert_allclose(gq.numpy(), g["grad_query"], rtol=1e-3, atol=2e-4)


def test_transformer_parameter_count_full_size():
    """17,297,920 parameters at the DanceTrack config."""
    from memotr_amd.models.deformable_transformer import build
    cfg = small_config()
    cfg.update(HIDDEN_DIM=256, FFN_DIM=2048, NUM_ENC_LAYERS=6, NUM_DEC_LAYERS=6, NUM_DET_QUERIES=300)
    tr = build(cfg)
    assert sum(p.numel() for p in tr.parameters()) == 17_297_920


def test_query_updater_matches_reference():
    from memotr_amd.models.query_updater import build
    g = load_model_golden("M4_query_updater")
    qu = build(small_config())
    qu.load_state_dict(state_from(g))
    assert sum(p.numel() for p in qu.parameters()) == 95_808
    qu.eval()
    with torch.no_grad():
        out = qu.update_tracks_embedding([tracks_from(g, "in_")])[0]
    assert_tracks_close(out, g, "out_")
    qu.train()
    res = qu([tracks_from(g, "prev_")], [tracks_from(g, "new_")], [tracks_from(g, "unm_")])[0]
    assert_tracks_close(res, g, "train_out_")


def test_query_updater_fake_track_when_nothing_survives():
    from memotr_amd.models.query_updater import build
    from memotr_amd.structures.track_instances import TrackInstances
    qu = build(small_config()).train()
    empty = lambda: TrackInstances(hidden_dim=64, num_classes=1, use_dab=True)   # noqa: E731
    out = qu([empty()], [empty()], [empty()])[0]
    assert len(out) == 1 and int(out.ids[0]) == -2
    loss = out.query_embed.sum()
    loss.backward()
    assert all(p.grad is not None for p in qu.parameters())      # every parameter reached (DDP contract)


def build_memotr(g):
    from memotr_amd.models.backbone import BackboneWithPE
    from memotr_amd.models.deformable_transformer import build as build_tr
    from memotr_amd.models.memotr import MeMOTR
    from memotr_amd.models.position_embedding import build as build_pe
    from memotr_amd.models.query_updater import build as build_qu
    cfg = small_config()
    model = MeMOTR(backbone=BackboneWithPE(TinyBackbone(), build_pe(cfg)), transformer=build_tr(cfg),
                   query_updater=build_qu(cfg), num_classes=1, n_det_queries=20, n_feature_levels=4, hidden_dim=64,
                   ffn_dim=128, dropout=0.0, use_dab=True)
    missing, unexpected = model.load_state_dict(state_from(g), strict=True), None
    return model.eval()


def test_memotr_two_frame_inference_matches_reference(monkeypatch):
    from memotr_amd.models.runtime_tracker import RuntimeTracker
    from memotr_amd.structures.track_instances import TrackInstances
    from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor
    patch_operator(monkeypatch)
    g = load_model_golden("M5_memotr_two_frames")
    model = build_memotr(g)
    # the aliased box heads appear under both names, as in the reference checkpoint layout
    sd = model.state_dict()
    assert "transformer.decoder.bbox_embed.0.layers.0.weight" in sd and "bbox_embed.0.layers.0.weight" in sd
    thresh = float(g["score_thresh"])
    tracker = RuntimeTracker(det_score_thresh=thresh, track_score_thresh=thresh, miss_tolerance=30, use_dab=True)
    tracks = [TrackInstances(hidden_dim=64, num_classes=1, use_dab=True)]
    with torch.no_grad():
        for i in range(2):
            res = model(frame=tensor_list_to_nested_tensor([t(g[f"frame{i}"])]), tracks=tracks)
            for k in ("pred_logits", "pred_bboxes", "last_ref_pts", "query_mask", "det_query_embed", "init_ref_pts",
                      "outputs"):
                want = g[f"f{i}_{k}"]
                assert res[k].shape == want.shape, k
                if want.dtype == bool:
                    assert np.array_equal(res[k].numpy(), want)
                else:
                    np.testing.assert_allclose(res[k].numpy(), want, rtol=1e-4, atol=5e-5, err_msg=f"f{i}_{k}")
            assert len(res["aux_outputs"]) == 1
            for k in ("pred_logits", "pred_bboxes", "queries"):
                np.testing.assert_allclose(res["aux_outputs"][0][k].numpy(), g[f"f{i}_aux0_{k}"], rtol=1e-4,
                                           atol=5e-5)
            prev, new = tracker.update(model_outputs=res, tracks=tracks)
            assert_tracks_close(prev[0], g, f"f{i}_prev_", atol=5e-5)
            assert_tracks_close(new[0], g, f"f{i}_new_", atol=5e-5)
            tracks = model.postprocess_single_frame(prev, new, None)
            assert_tracks_close(tracks[0], g, f"f{i}_next_", atol=1e-4)
    assert len(tracks[0]) == g["f1_next_ids"].shape[0] > 0


@pytest.mark.parametrize("chunks", ["0", "1", "all", "1,2", "lazy:2", "auto", "enc:1", "enc:1,2"])
def test_train_step_matches_reference(monkeypatch, chunks):
    """the body of train_engine.py:192-238 (criterion + matcher + query updater + backward)
    on the reference's seeded 3-frame clip: per-frame track sets, every loss term, and the gradient norm of
    every trainable parameter."""
    from memotr_amd.engine import clip_forward_backward
    from memotr_amd.models.criterion import build as build_criterion
    patch_operator(monkeypatch)
    g = load_model_golden("M6_train_step")
    model = build_memotr(g).train()
    cfg = small_config()
    cfg.update(MATCH_COST_CLASS=2, MATCH_COST_BBOX=5, MATCH_COST_GIOU=2, LOSS_WEIGHT_FOCAL=2, LOSS_WEIGHT_L1=5,
               LOSS_WEIGHT_GIOU=2, AUX_LOSS_WEIGHT=[1.0], SAMPLE_LENGTHS=[2, 3, 4, 5])
    criterion = build_criterion(cfg)
    T = 3
    batch = {"imgs": [[t(g[f"img{i}"]) for i in range(T)]],
             "infos": [[{"ids": t(g[f"gt{i}_ids"]), "labels": torch.zeros(6, dtype=torch.long),
                         "boxes": t(g[f"gt{i}_boxes"])} for i in range(T)]]}

    seen = {}
    orig = criterion.finish_tracks      # both loop orders hand a frame's tracks on here (finish_frame = finish_tracks + finish_losses)

    def spy(state):
        res = orig(state)
        # snapshot: the query updater later rewrites some of these objects in place
        seen[state["frame_idx"]] = [[tr[torch.ones(len(tr), dtype=torch.bool)] if len(tr) else tr for tr in group]
                                    for group in res]
        return res

    criterion.finish_tracks = spy
    # engine.clip_forward_backward: reference order / per-frame encode ahead / all frames in one batched encode / 1+2
    model.encode_chunks = chunks
    loss, loss_dict = clip_forward_backward(model, criterion, batch, torch.device("cpu"), use_dab=True)
    for i in range(T):
        prev, new, unm = seen[i]
        assert_tracks_close(prev[0], g, f"t{i}_prev_", atol=1e-4)
        assert_tracks_close(new[0], g, f"t{i}_new_", atol=1e-4)
        assert_tracks_close(unm[0], g, f"t{i}_unm_", atol=1e-4)
    for k, v in loss_dict.items():
        np.testing.assert_allclose(float(v), float(g[f"loss::{k}"]), rtol=2e-4, err_msg=k)
    np.testing.assert_allclose(float(loss), float(g["total_loss"]), rtol=2e-4)
    n = 0
    for name, p in model.named_parameters():
        if p.requires_grad:
            assert p.grad is not None, name
            np.testing.assert_allclose(float(p.grad.norm()), float(g[f"g::{name}"]), rtol=5e-3, atol=1e-5,
                                       err_msg=name)
            n += 1
    assert n == sum(1 for k in g if k.startswith("g::"))


def test_param_groups_follow_the_reference_keywords():
    from memotr_amd.engine import get_param_groups
    g = load_model_golden("M5_memotr_two_frames")
    model = build_memotr(g)
    cfg = dict(LR=2e-4, LR_BACKBONE=2e-5, LR_POINTS=1e-5)
    groups, names = get_param_groups(cfg, model)
    assert names == ["lr_backbone", "lr_points", "lr_query_updater", "lr"]
    assert [gr["lr"] for gr in groups] == [2e-5, 1e-5, 2e-4, 2e-4]
    n_all = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert sum(p.numel() for gr in groups for p in gr["params"]) == n_all
    point_names = [n for n, _ in model.named_parameters() if "sampling_offsets" in n]
    assert len(groups[1]["params"]) == len(point_names) > 0


def test_sequence_tracker_reproduces_the_submit_loop(monkeypatch):
    """submit_engine.py:58-120,171-184: per-frame filter (score > thresh, area > 100), xyxy pixel boxes and MOT lines,
    checked against the tracks the reference produced for the same two frames."""
    from memotr_amd.inference import SequenceTracker
    patch_operator(monkeypatch)
    g = load_model_golden("M5_memotr_two_frames")
    model = build_memotr(g)
    thresh = float(g["score_thresh"])
    st = SequenceTracker(model, dataset_name="DanceTrack", det_score_thresh=thresh, track_score_thresh=thresh,
                         result_score_thresh=thresh, miss_tolerance=30, use_dab=True)
    ori_h, ori_w = 240, 300
    for i in range(2):
        out = st.step(t(g[f"frame{i}"]), ori_h, ori_w)
        boxes, scores, ids = g[f"f{i}_next_boxes"], g[f"f{i}_next_scores"], g[f"f{i}_next_ids"]
        area = boxes[:, 2] * ori_w * boxes[:, 3] * ori_h
        keep = (scores.max(-1) > thresh) & (area > 100)
        assert np.array_equal(out.ids.numpy(), ids[keep])
        cx, cy, w, h = boxes[keep].T
        want = np.stack([(cx - 0.5 * w) * ori_w, (cy - 0.5 * h) * ori_h, (cx + 0.5 * w) * ori_w,
                         (cy + 0.5 * h) * ori_h], -1)
        np.testing.assert_allclose(out.boxes.numpy(), want, rtol=1e-4, atol=2e-2)
        lines = st.mot_lines(i, out)
        assert len(lines) == int(keep.sum()) and all(ln.endswith(",1,-1,-1,-1\n") for ln in lines)
        if lines:
            f, tid = lines[0].split(",")[:2]
            assert int(f) == i + 1 and int(tid) == int(ids[keep][0])
    js = st.bdd_frame_result(1, out, "a/b/video-0000001.jpg")
    assert js["frameIndex"] == 1 and len(js["labels"]) == len(out) and js["name"] == "video-0000001.jpg"


def test_activation_checkpointing_matches_plain_train_step(monkeypatch):
    """--use-checkpoint (CHECKPOINT_LEVEL 2: backbone + whole encoder + every decoder layer, all
    use_reentrant=False; reference memotr.py:102-103, deformable_transformer.py:223-226, deformable_decoder.py:104-118):
    same loss and gradients as the plain step (the reference run gives bit-identical values)."""
    from memotr_amd.engine import clip_forward_backward
    from memotr_amd.models.criterion import build as build_criterion
    patch_operator(monkeypatch)
    g = load_model_golden("M6_train_step")
    cfg = small_config()
    cfg.update(MATCH_COST_CLASS=2, MATCH_COST_BBOX=5, MATCH_COST_GIOU=2, LOSS_WEIGHT_FOCAL=2, LOSS_WEIGHT_L1=5,
               LOSS_WEIGHT_GIOU=2, AUX_LOSS_WEIGHT=[1.0], SAMPLE_LENGTHS=[2, 3, 4, 5])
    batch = {"imgs": [[t(g[f"img{i}"]) for i in range(3)]],
             "infos": [[{"ids": t(g[f"gt{i}_ids"]), "labels": torch.zeros(6, dtype=torch.long),
                         "boxes": t(g[f"gt{i}_boxes"])} for i in range(3)]]}
    results = []
    for level in (None, 1, 2, 3):
        model = build_memotr(g).train()
        if level is not None:
            model.use_checkpoint = True
            model.checkpoint_level = level
            tr = model.transformer
            tr.use_checkpoint, tr.checkpoint_level = True, level
            tr.encoder.use_checkpoint = level == 1
            tr.decoder.use_checkpoint = True
        loss, _ = clip_forward_backward(model, build_criterion(cfg), batch, torch.device("cpu"))
        gn = torch.sqrt(sum((p.grad ** 2).sum() for p in model.parameters() if p.grad is not None))
        results.append((float(loss.detach()), float(gn)))
    for loss, gn in results[1:]:
        assert loss == pytest.approx(results[0][0], rel=1e-6) and gn == pytest.approx(results[0][1], rel=1e-5)
    assert results[0][0] == pytest.approx(float(g["total_loss"]), rel=2e-4)


def test_bdd100k_eight_class_model_runs_two_frames(monkeypatch):
    """BDD100K config (8 classes): heads, criterion one-hot and track logits widen to K=8 (memotr.py:291-297)."""
    from memotr_amd.configs import bdd100k_config
    from memotr_amd.engine import clip_forward_backward, make_synthetic_clip
    from memotr_amd.models.backbone import BackboneWithPE
    from memotr_amd.models.criterion import build as build_criterion
    from memotr_amd.models.deformable_transformer import build as build_tr
    from memotr_amd.models.memotr import DATASET_NUM_CLASSES, MeMOTR
    from memotr_amd.models.position_embedding import build as build_pe
    from memotr_amd.models.query_updater import build as build_qu
    patch_operator(monkeypatch)
    cfg = bdd100k_config(HIDDEN_DIM=64, FFN_DIM=128, NUM_ENC_LAYERS=1, NUM_DEC_LAYERS=2, NUM_DET_QUERIES=20,
                         DEVICE="cpu", AUX_LOSS_WEIGHT=[1.0])
    K = DATASET_NUM_CLASSES[cfg["DATASET"]]
    assert K == 8 and cfg["MISS_TOLERANCE"] == 10 and cfg["SAMPLE_LENGTHS"] == [2, 3, 4]
    torch.manual_seed(0)
    model = MeMOTR(backbone=BackboneWithPE(TinyBackbone(), build_pe(cfg)), transformer=build_tr(cfg),
                   query_updater=build_qu(cfg), num_classes=K, n_det_queries=20, n_feature_levels=4, hidden_dim=64,
                   ffn_dim=128, dropout=0.0, use_dab=True).train()
    batch = make_synthetic_clip(clip_len=2, height=96, width=160, n_gts=5, seed=3, num_classes=K)
    loss, loss_dict = clip_forward_backward(model, build_criterion(cfg), batch, torch.device("cpu"))
    assert torch.isfinite(loss) and set(loss_dict) >= {"label_focal_loss", "aux_box_giou_loss"}
    assert model.class_embed[0].weight.shape == (8, 64)
    assert all(p.grad is not None for p in model.parameters() if p.requires_grad)


def _m7_module(g, tag):
    from memotr_amd.modules import MSDeformAttn
    d_model, heads, levels, points = (int(x) for x in g[f"{tag}_dims"])
    mod = MSDeformAttn(d_model=d_model, n_levels=levels, n_heads=heads, n_points=points)
    mod.load_state_dict(state_from(g, f"{tag}_w::"))
    return mod


@pytest.mark.parametrize("tag", ["a", "b"])
def test_msdeform_module_d32_matches_reference(monkeypatch, tag):
    """Fixture M7 (D = 32 per head, padding mask, both reference branches): output and the gradient of every
    differentiable input -- the CPU twin (oracle operator injected) of the fused-prologue GPU test."""
    patch_operator(monkeypatch)
    g = load_model_golden("M7_msdeform_module_d32")
    mod = _m7_module(g, tag)
    for rtag in ("ref2", "ref4"):
        k = f"{tag}_{rtag}_"
        query, src, ref = (t(g[k + n]).requires_grad_(True) for n in ("query", "src", "ref"))
        out = mod(query, ref, src, t(g[f"{tag}_shapes"]), t(g[f"{tag}_level_start"]), t(g[f"{tag}_mask"]))
        np.testing.assert_allclose(out.detach().numpy(), g[k + "out"], **TOL)
        gq, gs, gr = torch.autograd.grad(out, (query, src, ref), t(g[k + "grad_out"]))
        np.testing.assert_allclose(gq.numpy(), g[k + "grad_query"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(gs.numpy(), g[k + "grad_src"], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(gr.numpy(), g[k + "grad_ref"], rtol=1e-4, atol=2e-4)


def test_memotr_without_dab_matches_reference(monkeypatch):
    """Fixture M8 (USE_DAB False, configs/*_deformable_detr.yaml) with carried tracks: the level-0 head adds the full
    4-d inverse sigmoid of the initial reference (reference models/memotr.py:148-158), so the decoder's own
    refinement (2-d) must NOT be reused for the exposed boxes."""
    from memotr_amd.models.backbone import BackboneWithPE
    from memotr_amd.models.deformable_transformer import build as build_tr
    from memotr_amd.models.memotr import MeMOTR
    from memotr_amd.models.position_embedding import build as build_pe
    from memotr_amd.models.query_updater import build as build_qu
    from memotr_amd.structures.track_instances import TrackInstances
    from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor
    from model_helpers import TRACK_FIELDS
    patch_operator(monkeypatch)
    g = load_model_golden("M8_memotr_no_dab")
    cfg = small_config()
    cfg.update(USE_DAB=False)
    model = MeMOTR(backbone=BackboneWithPE(TinyBackbone(), build_pe(cfg)), transformer=build_tr(cfg),
                   query_updater=build_qu(cfg), num_classes=1, n_det_queries=20, n_feature_levels=4, hidden_dim=64,
                   ffn_dim=128, dropout=0.0, use_dab=False)
    model.load_state_dict(state_from(g), strict=True)
    model.train()
    tr = TrackInstances(hidden_dim=64, num_classes=1, use_dab=False)
    for k in TRACK_FIELDS:
        setattr(tr, k, t(g["in_" + k]))
    res = model(frame=tensor_list_to_nested_tensor([t(g["frame"])]), tracks=[tr])
    for k in ("pred_logits", "pred_bboxes", "last_ref_pts", "init_ref_pts", "outputs"):
        np.testing.assert_allclose(res[k].detach().numpy(), g[k], rtol=1e-4, atol=5e-5, err_msg=k)
    loss = res["pred_bboxes"].square().sum() + res["pred_logits"].sum()
    for j, aux in enumerate(res["aux_outputs"]):
        for k in ("pred_logits", "pred_bboxes"):
            np.testing.assert_allclose(aux[k].detach().numpy(), g[f"aux{j}_{k}"], rtol=1e-4, atol=5e-5,
                                       err_msg=f"aux{j}_{k}")
        loss = loss + (aux["pred_bboxes"] * torch.linspace(0.5, 2.0, 4)).square().sum()
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-4)
    checked = 0
    for name, p in model.named_parameters():
        if f"g::{name}" in g:
            want = float(g[f"g::{name}"])
            assert p.grad is not None, name
            np.testing.assert_allclose(float(p.grad.norm()), want, rtol=2e-3, atol=1e-5, err_msg=name)
            checked += 1
    assert checked > 50


def test_no_grad_frames_follow_the_reference_loop(monkeypatch):
    """NO_GRAD_FRAMES (train_engine.py:202-230): leading frames run without a graph and -- all but the last of
    them -- without augmentation; the loss of the remaining frames and its gradients still flow."""
    from memotr_amd.engine import clip_forward_backward, make_synthetic_clip
    from memotr_amd.models.criterion import build as build_criterion
    from model_helpers import build_small_memotr
    patch_operator(monkeypatch)
    cfg = small_config()
    cfg.update(MATCH_COST_CLASS=2, MATCH_COST_BBOX=5, MATCH_COST_GIOU=2, LOSS_WEIGHT_FOCAL=2, LOSS_WEIGHT_L1=5,
               LOSS_WEIGHT_GIOU=2, AUX_LOSS_WEIGHT=[1.0], SAMPLE_LENGTHS=[2, 3])
    torch.manual_seed(3)
    model = build_small_memotr().train()
    batch = make_synthetic_clip(clip_len=3, height=96, width=128, n_gts=4, seed=11)
    seen = []
    orig = model.postprocess_single_frame

    def spy(prev, new, unm, no_augment=False):
        seen.append((torch.is_grad_enabled(), no_augment))
        return orig(prev, new, unm, no_augment)

    monkeypatch.setattr(model, "postprocess_single_frame", spy)
    loss, _ = clip_forward_backward(model, build_criterion(cfg), batch, torch.device("cpu"), no_grad_frames=2)
    assert seen == [(False, True), (False, False)]       # frame 0: frozen + no augmentation; frame 1: frozen
    assert torch.isfinite(loss) and loss.requires_grad
    assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in model.parameters())


def test_active_track_rows_from_the_host_equal_the_boolean_mask(monkeypatch):
    """engine / criterion / query updater, round 6: the rows the updater keeps active come as an index the host computed
    from flags sent along with the matching costs (criterion.finish_tracks: keep_rows) -- the same rows, in the same
    order, as the reference's boolean mask ``(scores > update_thresh) | (ids >= 0)`` (models/query_updater.py:170-176),
    with detections above the threshold in the set (a low threshold, so that unclaimed detections DO stay active)."""
    from memotr_amd.engine import clip_forward_backward
    from memotr_amd.models.criterion import build as build_criterion
    from memotr_amd.models.query_updater import QueryUpdater
    patch_operator(monkeypatch)
    g = load_model_golden("M6_train_step")
    T = 3
    batch = {"imgs": [[t(g[f"img{i}"]) for i in range(T)]],
             "infos": [[{"ids": t(g[f"gt{i}_ids"]), "labels": torch.zeros(6, dtype=torch.long),
                         "boxes": t(g[f"gt{i}_boxes"])} for i in range(T)]]}
    results = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MEMOTR_KEEP_ROWS", mode)
        torch.manual_seed(0)
        model = build_memotr(g).train()
        model.query_updater.update_threshold = 0.50915      # some, not all, unclaimed detections pass (scores: 0.5085-0.5093)
        cfg = small_config()
        cfg.update(MATCH_COST_CLASS=2, MATCH_COST_BBOX=5, MATCH_COST_GIOU=2, LOSS_WEIGHT_FOCAL=2, LOSS_WEIGHT_L1=5,
                   LOSS_WEIGHT_GIOU=2, AUX_LOSS_WEIGHT=[1.0], SAMPLE_LENGTHS=[2, 3, 4, 5])
        criterion = build_criterion(cfg)
        used, kept = [], []
        orig = QueryUpdater.select_active_tracks

        def spy(self, prev, new, unm, no_augment=False):
            used.append("_keep_rows" in unm[0].__dict__)
            out = orig(self, prev, new, unm, no_augment=no_augment)
            kept.append((len(prev[0]), len(new[0]), len(unm[0]), out[0].ids.clone(), out[0].boxes.detach().clone()))
            return out

        monkeypatch.setattr(QueryUpdater, "select_active_tracks", spy)
        model.encode_chunks = "all"
        loss, _ = clip_forward_backward(model, criterion, batch, torch.device("cpu"), use_dab=True)
        monkeypatch.setattr(QueryUpdater, "select_active_tracks", orig)
        assert used == [mode == "1"] * (T - 1)
        results[mode] = (float(loss), kept, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    (l1, k1, g1), (l0, k0, g0) = results["1"], results["0"]
    assert l1 == l0
    assert any(len(ids) > n_prev + n_new for n_prev, n_new, _, ids, _ in k1)       # detections above the threshold stayed
    assert any(len(ids) < n_prev + n_new + n_unm for n_prev, n_new, n_unm, ids, _ in k1)     # ... and some rows went
    for a, b in zip(k1, k0):
        assert a[:3] == b[:3] and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert g1.keys() == g0.keys() and all(torch.equal(g1[n], g0[n]) for n in g1)
