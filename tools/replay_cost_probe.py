#!/usr/bin/env python
"""Round 6: what does replaying a captured graph cost the HOST?  CUDAGraph.replay wrapped with a host clock and a pair of
events, over the clip forward of the train step (decoder graphs: ~185 kernel nodes forward; updater graphs: ~52)."""
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip, optimizer_step  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402

cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)
for _ in range(3):
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)
torch.cuda.synchronize()
log = []
orig = torch.cuda.CUDAGraph.replay


def timed(self):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    orig(self)
    t1 = time.perf_counter()
    e1.record()
    log.append((t0, t1, e0, e1))


torch.cuda.CUDAGraph.replay = timed
waits = []
orig_sync = torch.cuda.Event.synchronize


def timed_sync(self):
    t0 = time.perf_counter()
    orig_sync(self)
    waits.append((t0, time.perf_counter()))


torch.cuda.Event.synchronize = timed_sync
for _ in range(2):
    log.clear()
    waits.clear()
    t_start = time.perf_counter()
    loss, _ = clip_forward_backward(model, criterion, batch, dev, backward=False)
    t_fwd = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    optimizer_step(model, opt, 0.1)
torch.cuda.CUDAGraph.replay = orig
torch.cuda.Event.synchronize = orig_sync
for t0, t1 in waits:
    print(f"event wait: entered at {(t0 - t_start) * 1e3:7.2f} ms, waited {(t1 - t0) * 1e6:7.0f} us")
print(f"clip forward on the host: {(t_fwd - t_start) * 1e3:.2f} ms")
for i, (t0, t1, e0, e1) in enumerate(log):
    print(f"replay {i:2d}: host {(t1 - t0) * 1e6:7.0f} us   GPU (event to event) {e0.elapsed_time(e1) * 1e3:7.0f} us   issued at {(t0 - t_start) * 1e3:7.2f} ms")
