"""Shared helpers for the model-level CPU parity tests (test infrastructure)."""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import GOLDEN_DIR
from oracle import msda_oracle as oracle


def load_model_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, f"model_{name}.npz")) as z:
        return {k: z[k] for k in z.files}


def state_from(g, prefix="w::"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)}


def t(a):
    return torch.from_numpy(np.array(a, copy=True))


class OracleMSDeformAttnFunction:
    """CPU stand-in for the HIP operator in host-logic tests: the oracle's torch statement of the
    reference formula (differentiable).  The product never uses this -- tests inject it explicitly."""

    @staticmethod
    def apply(value, shapes, level_start, loc, attn, im2col_step):
        shapes_list = [(int(h), int(w)) for h, w in shapes.tolist()]
        return oracle.grid_sample_forward(value, shapes_list, loc, attn)


def patch_operator(monkeypatch):
    import memotr_amd.modules.ms_deform_attn as mod
    monkeypatch.setattr(mod, "MSDeformAttnFunction", OracleMSDeformAttnFunction)


def small_config():
    """The reference's train_dancetrack.yaml values that the model path reads, at the fixture's reduced size."""
    return dict(
        BACKBONE="resnet50", HIDDEN_DIM=64, FFN_DIM=128, NUM_FEATURE_LEVELS=4, NUM_HEADS=8, NUM_ENC_POINTS=4,
        NUM_DEC_POINTS=4, NUM_ENC_LAYERS=2, NUM_DEC_LAYERS=2, MERGE_DET_TRACK_LAYER=1, ACTIVATION="ReLU",
        RETURN_INTER_DEC=True, EXTRA_TRACK_ATTN=False, AUX_LOSS=True, USE_DAB=True, UPDATE_THRESH=0.5,
        LONG_MEMORY_LAMBDA=0.01, DROPOUT=0.0, NUM_DET_QUERIES=20, TP_DROP_RATE=0.0, FP_INSERT_RATE=0.0,
        USE_CHECKPOINT=False, CHECKPOINT_LEVEL=2, VISUALIZE=False, DATASET="DanceTrack", DEVICE="cpu",
        AVAILABLE_GPUS=None)


class TinyBody(nn.Module):
    """Same stand-in backbone body as tests/golden/gen_golden_model.py."""

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


class TinyBackbone(nn.Module):
    def __init__(self):
        super().__init__()
        from memotr_amd.utils.nested_tensor import NestedTensor
        self._nt = NestedTensor
        self.backbone = TinyBody()
        self.strides = [8, 16, 32]
        self.num_channels = [8, 12, 16]

    def forward(self, nt):
        res = {}
        for name, out in self.backbone(nt.tensors).items():
            m = F.interpolate(nt.masks[None].float(), mode="nearest", size=out.shape[-2:]).to(nt.masks.dtype)[0]
            res[name] = self._nt(out, m)
        return res


TRACK_FIELDS = ("ref_pts", "query_embed", "ids", "boxes", "labels", "logits", "matched_idx", "output_embed",
                "disappear_time", "scores", "area", "iou", "last_output", "long_memory", "last_appear_boxes")


def tracks_from(g, prefix, hidden_dim=64, num_classes=1):
    from memotr_amd.structures.track_instances import TrackInstances
    tr = TrackInstances(hidden_dim=hidden_dim, num_classes=num_classes, use_dab=True)
    for k in TRACK_FIELDS:
        setattr(tr, k, t(g[prefix + k]))
    return tr


def assert_tracks_close(tr, g, prefix, atol=2e-5, skip=()):
    for k in TRACK_FIELDS:
        if k in skip:
            continue
        got, want = getattr(tr, k).detach().numpy(), g[prefix + k]
        assert got.shape == want.shape, (k, got.shape, want.shape)
        if want.dtype.kind in "iub":
            assert np.array_equal(got, want), k
        else:
            np.testing.assert_allclose(got, want, rtol=1e-4, atol=atol, err_msg=k)


def build_small_memotr():
    """Random-init MeMOTR at the fixtures' reduced size (stand-in backbone body)."""
    from memotr_amd.models.backbone import BackboneWithPE
    from memotr_amd.models.deformable_transformer import build as build_tr
    from memotr_amd.models.memotr import MeMOTR
    from memotr_amd.models.position_embedding import build as build_pe
    from memotr_amd.models.query_updater import build as build_qu
    cfg = small_config()
    return MeMOTR(backbone=BackboneWithPE(TinyBackbone(), build_pe(cfg)), transformer=build_tr(cfg),
                  query_updater=build_qu(cfg), num_classes=1, n_det_queries=20, n_feature_levels=4, hidden_dim=64,
                  ffn_dim=128, dropout=0.0, use_dab=True)
