#!/usr/bin/env python
"""In-process kernel breakdown of the clip train step (torch.profiler / roctracer), run on the GPU box.
rocprofv3 changes which MIOpen library / solver database the process picks up, so convolution timings are
taken in-process instead."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import (build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip,  # noqa: E402
                               optimizer_step)
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402

clip_len = int(os.environ.get("MEMOTR_BENCH_CLIP_LEN", "5"))
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(clip_len, 800, 1333, 10, seed=42), dev)


def step():
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)


for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("step wall ms", (time.perf_counter() - t0) * 1e3)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
tot = sum(getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) for e in ka)
print(f"total device time {tot/1e3:.1f} ms")
for e in rows[:45]:
    dt = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    print(f"{dt/1e3:9.2f} ms {100*dt/tot:5.1f}%  n={e.count:5d}  {e.key[:110]}")
