#!/usr/bin/env python
"""Wall time per phase of one clip train step (synchronising between phases; run on the GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import build_optimizer, clip_to_device, make_synthetic_clip, optimizer_step  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402
from memotr_amd.structures.track_instances import TrackInstances  # noqa: E402
from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor  # noqa: E402

cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
T = 5
batch = clip_to_device(make_synthetic_clip(T, 800, 1333, 10, seed=42), dev)
acc = {}


class Timer:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        torch.cuda.synchronize()
        self.t = time.perf_counter()

    def __exit__(self, *a):
        torch.cuda.synchronize()
        acc[self.name] = acc.get(self.name, 0.0) + (time.perf_counter() - self.t) * 1e3


def timed_forward(frame, tracks):
    m = model
    with Timer("fwd.backbone"):
        features, pos = m.backbone(frame)
    with Timer("fwd.proj+queries"):
        pos = list(pos)
        srcs, masks = [], []
        for lvl, feat in enumerate(features):
            src, mask = feat.decompose()
            srcs.append(m.feature_projs[lvl](src))
            masks.append(mask)
        src = m.feature_projs[3](features[-1].tensors)
        mask = torch.nn.functional.interpolate(frame.masks[None].float(), size=src.shape[-2:])[0].to(torch.bool)
        from memotr_amd.utils.nested_tensor import NestedTensor
        pos.append(m.backbone.position_embedding(NestedTensor(src, mask)))
        srcs.append(src); masks.append(mask)
        ref = m.get_reference_points(tracks); qe = m.get_query_embed(tracks); qm = m.get_query_mask(tracks)
    return srcs, masks, pos, qe, ref, qm


import contextlib
BF16 = os.environ.get("PHASE_BF16", "0") == "1"


def step(profile):
    with (torch.autocast("cuda", dtype=torch.bfloat16) if BF16 else contextlib.nullcontext()):
        _step(profile)


def _step(profile):
    tracks = TrackInstances.init_tracks(batch, hidden_dim=256, num_classes=1, device=dev, use_dab=True)
    criterion.init_a_clip(batch, 256, 1, dev)
    for t in range(T):
        frame = tensor_list_to_nested_tensor([batch["imgs"][0][t]]).to(dev)
        if profile:
            with Timer("fwd.model_total"):
                res = model(frame=frame, tracks=tracks)
        else:
            res = model(frame=frame, tracks=tracks)
        with Timer("criterion"):
            prev, new, unm = criterion.process_single_frame(res, tracks, t)
        if t < T - 1:
            with Timer("query_updater"):
                tracks = model.postprocess_single_frame(prev, new, unm)
    with Timer("loss_reduce"):
        loss_dict, _ = criterion.get_mean_by_n_gts()
        loss = criterion.get_sum_loss_dict(loss_dict)
    with Timer("backward"):
        loss.backward()
    with Timer("clip+optimizer"):
        optimizer_step(model, opt, 0.1)


for _ in range(3):
    step(False)
acc.clear()
N = 3
for _ in range(N):
    step(True)
tot = sum(acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:20s} {v/N:9.2f} ms/step  {100*v/tot:5.1f}%")
print(f"{'sum':20s} {tot/N:9.2f} ms/step")
# finer split of the forward (separate run; hooks synchronise)
acc.clear()
import types
names = {"backbone": model.backbone, "encoder": model.transformer.encoder, "decoder": model.transformer.decoder}
for name, mod in names.items():
    def pre(m, a, name=name):
        torch.cuda.synchronize(); m._t0 = time.perf_counter()
    def post(m, a, o, name=name):
        torch.cuda.synchronize(); acc["fwd." + name] = acc.get("fwd." + name, 0.0) + (time.perf_counter() - m._t0) * 1e3
    mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
for _ in range(N):
    step(True)
for k in ("fwd.backbone", "fwd.encoder", "fwd.decoder", "fwd.model_total"):
    print(f"{k:20s} {acc[k]/N:9.2f} ms/step")
