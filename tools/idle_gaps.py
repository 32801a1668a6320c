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


BF16 = "--bf16" in sys.argv          # the bf16 extension (autocast) instead of the fp32 step


def step():
    if BF16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            clip_forward_backward(model, criterion, batch, dev)
    else:
        clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack="--stack" in sys.argv) as prof:
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
        gaps.append((g, end - t0, a.name[:60], b.name[:60], end, b.time_range.start))
tot = sum(g[0] for g in gaps)
busy = sum(e.time_range.end - e.time_range.start for e in ev)
print(f"kernels {len(ev)}, busy {busy/1e3:.1f} ms")
# the span in 10 ms windows: busy fraction per window (where the host falls behind)
win = 10_000
nwin = int((ev[-1].time_range.end - t0) // win) + 1
occ = [0.0] * nwin
for e in ev:
    a, b = e.time_range.start - t0, e.time_range.end - t0
    w = int(a // win)
    while a < b and w < nwin:
        seg = min(b, (w + 1) * win) - a
        occ[w] += seg
        a += seg
        w += 1
print("busy % per 10 ms window:", " ".join(f"{100 * o / win:.0f}" for o in occ))
print(f"span {(ev[-1].time_range.end - t0)/1e3:.1f} ms, idle {tot/1e3:.1f} ms in {len(gaps)} gaps; gaps > 100 us: "
      f"{sum(g[0] for g in gaps if g[0] > 100)/1e3:.1f} ms")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"{g[0]/1e3:7.2f} ms at t={g[1]/1e3:7.1f} ms  after {g[2]:60s} before {g[3]}")

# what the host was doing during the largest gaps: the CPU operators open at the middle of each gap, outermost first
cpu = [e for e in prof.events() if e.device_type == DeviceType.CPU]
for g in sorted(gaps, reverse=True)[:4]:
    mid = (g[4] + g[5]) / 2
    open_ = sorted((e for e in cpu if e.time_range.start <= mid <= e.time_range.end),
                   key=lambda e: e.time_range.start)
    print(f"gap {g[0]/1e3:.2f} ms at t={g[1]/1e3:.1f}: " + " > ".join(f"{e.name[:48]}({(e.time_range.end - e.time_range.start)/1e3:.1f}ms)" for e in open_[-6:]))
    if open_:
        print("    shapes:", getattr(open_[-1], "input_shapes", None), [str(f) for f in (getattr(open_[-1], "stack", None) or [])[:12]])
