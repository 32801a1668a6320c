"""Generate model-level golden vectors by importing the REFERENCE (build container only).

    python tests/golden/gen_golden_model.py

Pins every host-side piece of the per-frame path that is not the HIP kernel itself
(fixtures F3-F6): sine embeddings, reference-point arithmetic, the ``MSDeformAttn``
module (2-d and 4-d reference branches), the full ``DeformableTransformer``, ``QueryUpdater`` and a
two-frame ``MeMOTR`` + ``RuntimeTracker`` inference loop.  The reference runs on CPU with
  * ``MultiScaleDeformableAttention`` stubbed by the reference's own pure-PyTorch statement
    (``ms_deform_attn_core_pytorch``, forward; autograd through it, backward), and
  * ``torchvision`` stubbed (only ``box_area`` is ever called);
model-level cases use the reference's own config knobs at reduced size (HIDDEN_DIM=64, FFN_DIM=128,
2+2 layers, 20 detect queries) and a stand-in backbone body (the ResNet-50 convolutions are
torchvision's arithmetic -- unpinned).  Weights are stored inside each fixture.
Only data is written (tests/golden/model_*.npz); no reference source is copied.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- stubs
def install_stubs():
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "models", "ops"))
    msda = types.ModuleType("MultiScaleDeformableAttention")
    sys.modules["MultiScaleDeformableAttention"] = msda
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvmu = types.ModuleType("torchvision.models._utils")
    tvo = types.ModuleType("torchvision.ops")
    tvob = types.ModuleType("torchvision.ops.boxes")
    tvm.resnet50 = None
    tvm.ResNet50_Weights = None
    tvmu.IntermediateLayerGetter = None
    tvob.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    for name, mod in (("torchvision", tv), ("torchvision.models", tvm), ("torchvision.models._utils", tvmu),
                      ("torchvision.ops", tvo), ("torchvision.ops.boxes", tvob)):
        sys.modules[name] = mod
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch as core

    def fwd(value, shapes, lsi, loc, attn, step):
        return core(value, shapes, loc, attn)

    def bwd(value, shapes, lsi, loc, attn, grad_out, step):
        with torch.enable_grad():
            v, l, a = (t.detach().requires_grad_(True) for t in (value, loc, attn))
            out = core(v, shapes, l, a)
            return list(torch.autograd.grad(out, (v, l, a), grad_out))

    msda.ms_deform_attn_forward = fwd
    msda.ms_deform_attn_backward = bwd


def small_config():
    from utils.utils import yaml_to_dict
    cfg = yaml_to_dict(os.path.join(REF, "configs", "train_dancetrack.yaml"))
    cfg.update(HIDDEN_DIM=64, FFN_DIM=128, NUM_ENC_LAYERS=2, NUM_DEC_LAYERS=2, NUM_DET_QUERIES=20, DEVICE="cpu",
               VISUALIZE=False, USE_CHECKPOINT=False, AUX_LOSS_WEIGHT=[1.0])
    return cfg


def np_state(module, prefix="w::"):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def randomize(module, seed, scale=0.3):
    """Deterministic non-degenerate weights (the reference init zeroes the offset/attention projections)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            if "sampling_offsets.bias" in name or p.dim() == 0:
                continue
            p.copy_(torch.randn(p.shape, generator=g) * (scale / max(1.0, p.shape[-1] ** 0.5) if p.dim() > 1 else 0.1))
    return module


class TinyBody(nn.Module):
    """Stand-in for the ResNet-50 body: three maps at strides 8/16/32 (same module on both sides)."""

    def __init__(self):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 8, stride=8)
        self.c2 = nn.Conv2d(8, 12, 3, stride=2, padding=1)
        self.c3 = nn.Conv2d(12, 16, 3, stride=2, padding=1)

    def forward(self, x):
        a = torch.tanh(self.c1(x))
        b = torch.tanh(self.c2(a))
        c = torch.tanh(self.c3(b))
        return {"0": a, "1": b, "2": c}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            out.update(v)
        else:
            out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **out)
    print(name, len(out), "arrays")


# --------------------------------------------------------------------------- cases
def case_small_functions():
    from models.position_embedding import build as build_pe
    from models.utils import pos_to_pos_embed
    from models.deformable_encoder import DeformableEncoder
    from models.deformable_transformer import DeformableTransformer
    from utils.nested_tensor import NestedTensor, tensor_list_to_nested_tensor
    from utils.utils import inverse_sigmoid
    g = torch.Generator().manual_seed(1)
    nt = tensor_list_to_nested_tensor([torch.rand(3, 50, 70, generator=g), torch.rand(3, 64, 61, generator=g)])
    masks = F.interpolate(nt.masks[None].float(), size=(8, 9)).to(torch.bool)[0]
    pe = build_pe({"HIDDEN_DIM": 64})(NestedTensor(torch.zeros(2, 1, 8, 9), masks))
    boxes = torch.rand(5, 4, generator=g)
    shapes = torch.tensor([[8, 9], [4, 5], [2, 3]])
    vr = torch.stack([DeformableTransformer.get_valid_ratio(
        F.interpolate(nt.masks[None].float(), size=(int(h), int(w))).to(torch.bool)[0]) for h, w in shapes], 1)
    ref = DeformableEncoder.get_reference_points(shapes, vr, device="cpu")
    x = torch.tensor([-0.5, 0.0, 1e-7, 1e-5, 0.3, 0.5, 1 - 1e-6, 1.0, 1.5])
    save("M1_small_functions", pad_tensors=nt.tensors, pad_masks=nt.masks, masks=masks, pe=pe, boxes=boxes,
         box_embed=pos_to_pos_embed(boxes, 32), box_embed_t20=pos_to_pos_embed(boxes[:, :2], 16, temperature=20),
         shapes=shapes, valid_ratios=vr, reference_points=ref, inv_sig_in=x, inv_sig_out=inverse_sigmoid(x))


def case_msdeform_module():
    from models.ops.modules import MSDeformAttn
    g = torch.Generator().manual_seed(2)
    mod = randomize(MSDeformAttn(d_model=64, n_levels=3, n_heads=8, n_points=4), 20)
    shapes_l = [(6, 8), (3, 4), (2, 2)]
    shapes = torch.tensor(shapes_l)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    src = torch.randn(2, S, 64, generator=g)
    mask = torch.zeros(2, S, dtype=torch.bool)
    mask[1, 40:48] = True
    mask[1, 66:] = True
    arrs = dict(shapes=shapes, level_start=lsi, src=src, mask=mask)
    for tag, nref in (("ref2", 2), ("ref4", 4)):
        Lq = 11
        query = torch.randn(2, Lq, 64, generator=g).requires_grad_(True)
        ref = torch.rand(2, Lq, 3, nref, generator=g)
        if nref == 4:
            ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
        out = mod(query, ref, src, shapes, lsi, mask)
        grad_out = torch.randn(out.shape, generator=g)
        (gq,) = torch.autograd.grad(out, query, grad_out)
        arrs.update({f"{tag}_query": query, f"{tag}_ref": ref, f"{tag}_out": out, f"{tag}_grad_out": grad_out,
                     f"{tag}_grad_query": gq})
    save("M2_msdeform_module", weights=np_state(mod), **arrs)


def transformer_inputs(g, B, shapes_l, C, n_det, n_track, valid_hw):
    srcs, masks, poss = [], [], []
    for (h, w) in shapes_l:
        srcs.append(torch.randn(B, C, h, w, generator=g))
        poss.append(torch.randn(B, C, h, w, generator=g) * 0.5)
    for (h, w) in shapes_l:
        m = torch.zeros(B, h, w, dtype=torch.bool)
        for b in range(B):
            fh, fw = valid_hw[b]
            m[b, max(1, int(round(h * fh))):, :] = True
            m[b, :, max(1, int(round(w * fw))):] = True
        masks.append(m)
    Nq = n_det + n_track
    query = torch.randn(B, Nq, C, generator=g)
    ref = torch.randn(B, Nq, 4, generator=g)
    qmask = torch.zeros(B, Nq, dtype=torch.bool)
    qmask[1, n_det + 1:] = True          # clip 1 carries one track, clip 0 carries n_track
    return srcs, masks, poss, query, ref, qmask


def case_transformer():
    from models.deformable_transformer import build
    from models.mlp import MLP
    from models.utils import get_clones
    cfg = small_config()
    tr = randomize(build(cfg), 30)
    bbox = get_clones(randomize(MLP(64, 64, 4, 3), 31, scale=0.1), 2)
    tr.set_refine_bbox_embed(bbox)
    g = torch.Generator().manual_seed(3)
    shapes_l = [(6, 8), (3, 4), (2, 2), (1, 1)]
    srcs, masks, poss, query, ref, qmask = transformer_inputs(g, 2, shapes_l, 64, 20, 3, [(1.0, 1.0), (0.7, 0.8)])
    query.requires_grad_(True)
    out, init_ref, inter_ref, inter_q = tr(srcs, masks, poss, query, ref, qmask)
    loss = (out[-1] * torch.linspace(-1, 1, 64)).sum() + inter_ref[-1].sum()
    (gq,) = torch.autograd.grad(loss, query)
    arrs = {f"src{i}": s for i, s in enumerate(srcs)}
    arrs.update({f"mask{i}": m for i, m in enumerate(masks)})
    arrs.update({f"pos{i}": p for i, p in enumerate(poss)})
    save("M3_transformer", weights=np_state(tr), bbox=np_state(bbox, "b::"), query=query, ref=ref, qmask=qmask,
         out=out, init_ref=init_ref, inter_ref=inter_ref, inter_q=inter_q, grad_query=gq, **arrs)


def make_tracks(g, n, C, K, TrackInstances):
    t = TrackInstances(hidden_dim=C, num_classes=K, use_dab=True)
    t.ref_pts = torch.randn(n, 4, generator=g)
    t.query_embed = torch.randn(n, C, generator=g)
    t.ids = torch.arange(n) - 1                     # first id is -1
    t.boxes = torch.rand(n, 4, generator=g) * 0.5 + 0.2
    t.labels = torch.zeros(n, dtype=torch.long)
    t.logits = torch.randn(n, K, generator=g) * 2   # about half above UPDATE_THRESH
    t.matched_idx = torch.arange(n)
    t.output_embed = torch.randn(n, C, generator=g)
    t.disappear_time = torch.zeros(n, dtype=torch.long)
    t.scores = t.logits.sigmoid()
    t.area = torch.rand(n, generator=g)
    t.iou = torch.rand(n, generator=g)
    t.last_output = torch.randn(n, C, generator=g)
    t.long_memory = torch.randn(n, C, generator=g)
    t.last_appear_boxes = torch.rand(n, 4, generator=g)
    return t


TRACK_FIELDS = ("ref_pts", "query_embed", "ids", "boxes", "labels", "logits", "matched_idx", "output_embed",
                "disappear_time", "scores", "area", "iou", "last_output", "long_memory", "last_appear_boxes")


def track_arrays(prefix, t):
    return {f"{prefix}{k}": getattr(t, k).detach().clone() for k in TRACK_FIELDS}


def case_query_updater():
    from models.query_updater import build
    from structures.track_instances import TrackInstances
    cfg = small_config()
    qu = randomize(build(cfg), 40).eval()
    g = torch.Generator().manual_seed(4)
    t = make_tracks(g, 7, 64, 1, TrackInstances)
    arrs = track_arrays("in_", t)
    with torch.no_grad():
        out = qu.update_tracks_embedding([t])[0]
    arrs.update(track_arrays("out_", out))
    # training-mode selection (TP_DROP = FP_INSERT = 0 branch) + update
    qu.train()
    prev = make_tracks(g, 4, 64, 1, TrackInstances)
    new = make_tracks(g, 3, 64, 1, TrackInstances)
    unm = make_tracks(g, 5, 64, 1, TrackInstances)
    unm.ids[:] = -1
    arrs.update(track_arrays("prev_", prev))
    arrs.update(track_arrays("new_", new))
    arrs.update(track_arrays("unm_", unm))
    res = qu([prev], [new], [unm])[0]
    arrs.update(track_arrays("train_out_", res))
    save("M4_query_updater", weights=np_state(qu), **arrs)


def case_memotr_two_frames():
    import models.backbone as ref_backbone
    from models.memotr import MeMOTR
    from models.deformable_transformer import build as build_tr
    from models.position_embedding import build as build_pe
    from models.query_updater import build as build_qu
    from models.runtime_tracker import RuntimeTracker
    from structures.track_instances import TrackInstances
    from utils.nested_tensor import NestedTensor, tensor_list_to_nested_tensor
    cfg = small_config()

    class TinyBackbone(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = TinyBody()
            self.strides = [8, 16, 32]
            self.num_channels = [8, 12, 16]

        def forward(self, nt):
            res = {}
            for name, out in self.backbone(nt.tensors).items():
                m = F.interpolate(nt.masks[None].float(), mode="nearest", size=out.shape[-2:]).to(nt.masks.dtype)[0]
                res[name] = NestedTensor(out, m)
            return res

    torch.manual_seed(50)
    model = MeMOTR(backbone=ref_backbone.BackboneWithPE(TinyBackbone(), build_pe(cfg)), transformer=build_tr(cfg),
                   query_updater=build_qu(cfg), num_classes=1, n_det_queries=20, n_feature_levels=4, hidden_dim=64,
                   ffn_dim=128, dropout=0.0, aux_loss=True, with_box_refine=True, use_checkpoint=False,
                   checkpoint_level=2, use_dab=True, visualize=False)
    randomize(model, 51)
    with torch.no_grad():
        for ce in model.class_embed:
            ce.bias.zero_()
    model.eval()
    g = torch.Generator().manual_seed(5)
    frames = [torch.rand(3, 120, 150, generator=g) for _ in range(2)]
    tracks = [TrackInstances(hidden_dim=64, num_classes=1, use_dab=True)]
    with torch.no_grad():   # thresholds at the median first-frame score, so about half of the queries spawn tracks
        probe = model(frame=tensor_list_to_nested_tensor([frames[0]]), tracks=tracks)
    thresh = float(probe["pred_logits"].sigmoid().median()) + 1e-6
    tracker = RuntimeTracker(det_score_thresh=thresh, track_score_thresh=thresh, miss_tolerance=30, use_dab=True)
    arrs = {"frame0": frames[0], "frame1": frames[1], "score_thresh": torch.tensor(thresh, dtype=torch.float64)}
    with torch.no_grad():
        for i, fr in enumerate(frames):
            nt = tensor_list_to_nested_tensor([fr])
            res = model(frame=nt, tracks=tracks)
            for k in ("pred_logits", "pred_bboxes", "last_ref_pts", "query_mask", "det_query_embed", "init_ref_pts",
                      "outputs"):
                arrs[f"f{i}_{k}"] = res[k].clone()
            for j, aux in enumerate(res["aux_outputs"]):
                for k in ("pred_logits", "pred_bboxes", "queries"):
                    arrs[f"f{i}_aux{j}_{k}"] = aux[k].clone()
            prev, new = tracker.update(model_outputs=res, tracks=tracks)
            arrs.update(track_arrays(f"f{i}_prev_", prev[0]))
            arrs.update(track_arrays(f"f{i}_new_", new[0]))
            tracks = model.postprocess_single_frame(prev, new, None)
            arrs.update(track_arrays(f"f{i}_next_", tracks[0]))
    assert len(tracks[0]) > 0, "fixture would not exercise the track path"
    print("tracks after frame 2:", len(tracks[0]))
    save("M5_memotr_two_frames", weights=np_state(model), **arrs)


def case_train_step():
    """the body of train_engine.py:192-238 on a seeded 3-frame clip with 6 ground-truth tracks."""
    import models.backbone as ref_backbone
    from models.criterion import build as build_criterion
    from models.memotr import MeMOTR
    from models.deformable_transformer import build as build_tr
    from models.position_embedding import build as build_pe
    from models.query_updater import build as build_qu
    from structures.track_instances import TrackInstances
    from utils.nested_tensor import NestedTensor, tensor_list_to_nested_tensor
    cfg = small_config()

    class TinyBackbone(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = TinyBody()
            self.strides = [8, 16, 32]
            self.num_channels = [8, 12, 16]

        def forward(self, nt):
            res = {}
            for name, out in self.backbone(nt.tensors).items():
                m = F.interpolate(nt.masks[None].float(), mode="nearest", size=out.shape[-2:]).to(nt.masks.dtype)[0]
                res[name] = NestedTensor(out, m)
            return res

    torch.manual_seed(60)
    model = MeMOTR(backbone=ref_backbone.BackboneWithPE(TinyBackbone(), build_pe(cfg)), transformer=build_tr(cfg),
                   query_updater=build_qu(cfg), num_classes=1, n_det_queries=20, n_feature_levels=4, hidden_dim=64,
                   ffn_dim=128, dropout=0.0, aux_loss=True, with_box_refine=True, use_checkpoint=False,
                   checkpoint_level=2, use_dab=True, visualize=False)
    randomize(model, 61)
    with torch.no_grad():
        for ce in model.class_embed:
            ce.bias.zero_()
    model.train()
    criterion = build_criterion(cfg)
    g = torch.Generator().manual_seed(6)
    T, n_gt = 3, 6
    centre = torch.rand(n_gt, 2, generator=g) * 0.6 + 0.2
    size = torch.rand(n_gt, 2, generator=g) * 0.15 + 0.05
    imgs, infos = [], []
    for t_ in range(T):
        imgs.append(torch.rand(3, 224, 352, generator=g))
        centre = (centre + torch.randn(n_gt, 2, generator=g) * 0.01).clamp(0.1, 0.9)
        ids = torch.arange(n_gt) if t_ < 2 else torch.tensor([0, 1, 2, 3, 7, 8])   # two identities change in frame 3
        infos.append({"ids": ids, "labels": torch.zeros(n_gt, dtype=torch.long), "boxes": torch.cat((centre, size), -1)})
    batch = {"imgs": [imgs], "infos": [infos]}
    arrs = {}
    for t_ in range(T):
        arrs[f"img{t_}"] = imgs[t_]
        arrs[f"gt{t_}_ids"], arrs[f"gt{t_}_boxes"] = infos[t_]["ids"], infos[t_]["boxes"]
    tracks = TrackInstances.init_tracks(batch=batch, hidden_dim=64, num_classes=1, device="cpu", use_dab=True)
    criterion.init_a_clip(batch=batch, hidden_dim=64, num_classes=1, device=torch.device("cpu"))
    for t_ in range(T):
        frame = tensor_list_to_nested_tensor([imgs[t_]])
        res = model(frame=frame, tracks=tracks)
        prev, new, unm = criterion.process_single_frame(model_outputs=res, tracked_instances=tracks, frame_idx=t_)
        arrs.update(track_arrays(f"t{t_}_prev_", prev[0]))
        arrs.update(track_arrays(f"t{t_}_new_", new[0]))
        arrs.update(track_arrays(f"t{t_}_unm_", unm[0]))
        if t_ < T - 1:
            tracks = model.postprocess_single_frame(prev, new, unm)
            arrs.update(track_arrays(f"t{t_}_next_", tracks[0]))
    loss_dict, log = criterion.get_mean_by_n_gts()
    loss = criterion.get_sum_loss_dict(loss_dict=loss_dict)
    loss.backward()
    for k, v in loss_dict.items():
        arrs[f"loss::{k}"] = v.detach()
    arrs["total_loss"] = loss.detach()
    n_none = 0
    for name, p in model.named_parameters():
        if p.requires_grad:
            if p.grad is None:
                n_none += 1
            else:
                arrs[f"g::{name}"] = p.grad.norm().double()
    assert n_none == 0, "a trainable parameter got no gradient"
    print("train step: loss", float(loss), "n_tracks last frame", len(tracks[0]))
    save("M6_train_step", weights=np_state(model), **arrs)


def case_msdeform_module_d32():
    """Fixture M7: the ``MSDeformAttn`` module at the MeMOTR head width (D = 32 channels per head -- the geometry the
    specialised and the fused-prologue kernels take), both reference-point branches, with a padding mask, and the
    gradients of every differentiable input (query, input_flatten, reference_points)."""
    from models.ops.modules import MSDeformAttn
    arrs = {}
    for tag, (d_model, heads, levels, points, shapes_l, Lq, N) in {
            "a": (64, 2, 4, 4, [(12, 16), (6, 8), (3, 4), (2, 2)], 23, 2),     # M = 2, L*P = 16
            "b": (96, 3, 2, 3, [(9, 7), (4, 5)], 10, 1)}.items():             # M = 3, L*P = 6 (not a multiple of 8)
        g = torch.Generator().manual_seed(70 + ord(tag))
        mod = randomize(MSDeformAttn(d_model=d_model, n_levels=levels, n_heads=heads, n_points=points), 71 + ord(tag))
        with torch.no_grad():     # offsets of a few pixels around the star, non-uniform attention
            mod.sampling_offsets.weight.mul_(3.0)
            mod.attention_weights.weight.mul_(4.0)
        shapes = torch.tensor(shapes_l)
        lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
        S = int(shapes.prod(1).sum())
        mask = torch.zeros(N, S, dtype=torch.bool)
        mask[-1, 5:19] = True
        mask[-1, S - 7:] = True
        arrs.update({f"{tag}_shapes": shapes, f"{tag}_level_start": lsi, f"{tag}_mask": mask,
                     f"{tag}_dims": torch.tensor([d_model, heads, levels, points])})
        arrs.update(np_state(mod, f"{tag}_w::"))
        for rtag, nref in (("ref2", 2), ("ref4", 4)):
            src = torch.randn(N, S, d_model, generator=g).requires_grad_(True)
            query = torch.randn(N, Lq, d_model, generator=g).requires_grad_(True)
            ref = torch.rand(N, Lq, levels, nref, generator=g)
            if nref == 4:
                ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
            ref.requires_grad_(True)
            out = mod(query, ref, src, shapes, lsi, mask)
            grad_out = torch.randn(out.shape, generator=g)
            gq, gs, gr = torch.autograd.grad(out, (query, src, ref), grad_out)
            k = f"{tag}_{rtag}_"
            arrs.update({k + "query": query, k + "src": src, k + "ref": ref, k + "out": out, k + "grad_out": grad_out,
                         k + "grad_query": gq, k + "grad_src": gs, k + "grad_ref": gr})
    save("M7_msdeform_module_d32", **arrs)


def case_memotr_no_dab():
    """Fixture M8: one training-mode frame with carried tracks of the Deformable-DETR variant (USE_DAB False, configs/*_deformable_detr.yaml): 2-d
    decoder references, ``reference_points`` Linear, 2C query embeddings, 4-d inverse sigmoid at head level 0."""
    import models.backbone as ref_backbone
    from models.memotr import MeMOTR
    from models.deformable_transformer import build as build_tr
    from models.position_embedding import build as build_pe
    from models.query_updater import build as build_qu
    from structures.track_instances import TrackInstances
    from utils.nested_tensor import NestedTensor, tensor_list_to_nested_tensor
    cfg = small_config()
    cfg.update(USE_DAB=False)

    class TinyBackbone(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = TinyBody()
            self.strides = [8, 16, 32]
            self.num_channels = [8, 12, 16]

        def forward(self, nt):
            res = {}
            for name, out in self.backbone(nt.tensors).items():
                m = F.interpolate(nt.masks[None].float(), mode="nearest", size=out.shape[-2:]).to(nt.masks.dtype)[0]
                res[name] = NestedTensor(out, m)
            return res

    torch.manual_seed(80)
    model = MeMOTR(backbone=ref_backbone.BackboneWithPE(TinyBackbone(), build_pe(cfg)), transformer=build_tr(cfg),
                   query_updater=build_qu(cfg), num_classes=1, n_det_queries=20, n_feature_levels=4, hidden_dim=64,
                   ffn_dim=128, dropout=0.0, aux_loss=True, with_box_refine=True, use_checkpoint=False,
                   checkpoint_level=2, use_dab=False, visualize=False)
    randomize(model, 81)
    with torch.no_grad():
        for ce in model.class_embed:
            ce.bias.zero_()
    model.train()
    g = torch.Generator().manual_seed(8)
    frame = torch.rand(3, 120, 150, generator=g)
    # hand-made carried tracks (the reference's RuntimeTracker hard-codes a 256-wide slice for this variant, so the
    # tracks are built directly): 5 track queries with 2C embeddings and 4-d logit-space reference points
    tr = TrackInstances(hidden_dim=64, num_classes=1, use_dab=False)
    n = 5
    tr.ref_pts = torch.randn(n, 4, generator=g)
    tr.query_embed = torch.randn(n, 128, generator=g)
    tr.ids = torch.arange(n)
    tr.boxes = torch.rand(n, 4, generator=g)
    tr.labels = torch.zeros(n, dtype=torch.long)
    tr.logits = torch.randn(n, 1, generator=g)
    tr.matched_idx = torch.arange(n)
    tr.output_embed = torch.randn(n, 64, generator=g)
    tr.disappear_time = torch.zeros(n, dtype=torch.long)
    tr.scores = tr.logits.sigmoid()
    tr.area = torch.rand(n, generator=g)
    tr.iou = torch.rand(n, generator=g)
    tr.last_output = torch.randn(n, 64, generator=g)
    tr.long_memory = torch.randn(n, 64, generator=g)
    tr.last_appear_boxes = torch.rand(n, 4, generator=g)
    arrs = {"frame": frame}
    arrs.update(track_arrays("in_", tr))
    res = model(frame=tensor_list_to_nested_tensor([frame]), tracks=[tr])
    for k in ("pred_logits", "pred_bboxes", "last_ref_pts", "init_ref_pts", "outputs"):
        arrs[k] = res[k].detach().clone()
    loss = res["pred_bboxes"].square().sum() + res["pred_logits"].sum()
    for j, aux in enumerate(res["aux_outputs"]):
        for k in ("pred_logits", "pred_bboxes"):
            arrs[f"aux{j}_{k}"] = aux[k].detach().clone()
        loss = loss + (aux["pred_bboxes"] * torch.linspace(0.5, 2.0, 4)).square().sum()
    loss.backward()
    arrs["loss"] = loss.detach()
    for name, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            arrs[f"g::{name}"] = p.grad.norm().double()
    save("M8_memotr_no_dab", weights=np_state(model), **arrs)


def main():
    install_stubs()
    torch.set_num_threads(1)
    only = set(sys.argv[1:])          # e.g. `gen_golden_model.py M7 M8` regenerates just those fixtures
    if only:
        for tag, fn in (("M7", case_msdeform_module_d32), ("M8", case_memotr_no_dab)):
            if tag in only:
                fn()
        return
    case_small_functions()
    case_msdeform_module()
    case_transformer()
    case_query_updater()
    case_memotr_two_frames()
    case_train_step()
    case_msdeform_module_d32()
    case_memotr_no_dab()


if __name__ == "__main__":
    main()
