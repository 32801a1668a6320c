#!/usr/bin/env python
"""Eager vs hipGraph decoder loop of one frame (forward + backward), host and wall time (run on the GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.modules.linear import configure_blas  # noqa: E402
from memotr_amd.structures.track_instances import TrackInstances  # noqa: E402
from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor  # noqa: E402

configure_blas()
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
frame = tensor_list_to_nested_tensor([torch.randn(3, 800, 1333, device=dev)])
tracks = [TrackInstances(hidden_dim=256, num_classes=1, use_dab=True).to(dev)]


def run(slot):
    enc = model(frame=frame, stage="encode")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = model(tracks=tracks, encoded=dict(enc, frame_slot=slot) if slot is not None else enc)
    (res["pred_bboxes"].sum() + res["pred_logits"].sum()).backward(inputs=[enc["memory"]] + list(model.transformer.decoder.parameters()))
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return host * 1e3, (time.perf_counter() - t0) * 1e3


for name, slot in (("eager", None), ("graphed", 0)):
    for _ in range(3):
        run(slot)
    h, w = zip(*[run(slot) for _ in range(8)])
    print(f"{name:8s} decode + heads, forward + backward of one frame: host {sum(h)/8:.2f} ms, wall {sum(w)/8:.2f} ms "
          f"(captures: {model.transformer.decoder.graphs().captures})")
