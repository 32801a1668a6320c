#!/usr/bin/env python
"""Where the GPU idles inside one clip train step: the largest gaps between consecutive kernels (torch.profiler),
with the kernels on either side (run on the GPU box)."""
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
from torch.autograd import DeviceType  # noqa: E402

cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)


def step():
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = sorted((e for e in prof.events() if e.device_type == DeviceType.CUDA), key=lambda e: e.time_range.start)
t0 = ev[0].time_range.start
gaps = []
end = ev[0].time_range.end
for a, b in zip(ev, ev[1:]):
    end = max(end, a.time_range.end)
    g = b.time_range.start - end
    if g > 0:
        gaps.append((g, end - t0, a.name[:60], b.name[:60]))
tot = sum(g[0] for g in gaps)
print(f"span {(ev[-1].time_range.end - t0)/1e3:.1f} ms, idle {tot/1e3:.1f} ms in {len(gaps)} gaps; gaps > 100 us: "
      f"{sum(g[0] for g in gaps if g[0] > 100)/1e3:.1f} ms")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"{g[0]/1e3:7.2f} ms at t={g[1]/1e3:7.1f} ms  after {g[2]:60s} before {g[3]}")
