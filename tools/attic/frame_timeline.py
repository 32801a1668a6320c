#!/usr/bin/env python
"""Host / GPU timeline of the frame loop of one clip (run on the GPU box).

Marks are taken at the same program points on the host (perf_counter: when the host got there) and on the GPU
(events: when the GPU got there).  host << gpu: the host runs ahead (GPU-bound); host == gpu: the GPU waits
for the host (launch-bound).  Usage: python tools/frame_timeline.py [0|1]   (1 = encoder of frame t+1 queued
before the host waits on frame t; 0 = the reference's order)
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip, optimizer_step  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402
from memotr_amd.structures.track_instances import TrackInstances  # noqa: E402
from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor  # noqa: E402

ahead = (sys.argv[1] if len(sys.argv) > 1 else "0") == "1"
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)
for _ in range(3):
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)
marks = []


def mark(name, gpu=True):
    ev = None
    if gpu:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
    marks.append((name, time.perf_counter(), ev))


def frame_at(i):
    return tensor_list_to_nested_tensor([clip[i] for clip in batch["imgs"]]).to(dev)


torch.cuda.synchronize()
tracks = TrackInstances.init_tracks(batch=batch, hidden_dim=model.hidden_dim, num_classes=model.num_classes, device=dev,
                                    use_dab=True)
criterion.init_a_clip(batch=batch, hidden_dim=model.hidden_dim, num_classes=model.num_classes, device=dev)
mark("start")
T = 5
enc = None
if ahead:
    enc = model(frame=frame_at(0), stage="encode")
    mark("f0 encode issued")
for t in range(T):
    if not ahead:
        enc = model(frame=frame_at(t), stage="encode")
        mark(f"f{t} encode issued")
    res = model(tracks=tracks, encoded=enc)
    mark(f"f{t} decode issued")
    pend = criterion.begin_frame(res, tracks, t)
    mark(f"f{t} cost matrices + copy issued")
    if ahead and t < T - 1:
        enc = model(frame=frame_at(t + 1), stage="encode")
        mark(f"f{t + 1} encode issued")
    if pend["ready"] is not None:
        pend["ready"].synchronize()
    mark(f"f{t} copy arrived", gpu=False)
    prev, new, unm = criterion.finish_frame(pend)
    mark(f"f{t} losses issued")
    if t < T - 1:
        tracks = model.postprocess_single_frame(prev, new, unm)
        mark(f"f{t} query updater issued")
loss_dict, _ = criterion.get_mean_by_n_gts()
loss = criterion.get_sum_loss_dict(loss_dict=loss_dict)
mark("loss")
loss.backward()
mark("backward issued")
optimizer_step(model, opt, 0.1)
mark("optimizer issued")
torch.cuda.synchronize()
t_end = time.perf_counter()
t0, e0 = marks[0][1], marks[0][2]
print(f"encode-ahead={ahead}  total {1e3 * (t_end - t0):.1f} ms")
print(f"{'mark':34s} {'host ms':>9s} {'gpu ms':>9s}")
for name, tc, ev in marks:
    g = f"{e0.elapsed_time(ev):9.1f}" if ev is not None else "        -"
    print(f"{name:34s} {1e3 * (tc - t0):9.1f} {g}")
