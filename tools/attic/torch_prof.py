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

from memotr_amd.train_bench import load_gemm_tuning  # noqa: E402
print("tuned GEMM entries:", load_gemm_tuning())
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
from collections import defaultdict
from torch.autograd import DeviceType

agg = defaultdict(lambda: [0.0, 0])
t_min, t_max = None, None
for e in prof.events():
    if e.device_type != DeviceType.CUDA:
        continue
    dur = e.time_range.end - e.time_range.start
    agg[e.name][0] += dur
    agg[e.name][1] += 1
    t_min = e.time_range.start if t_min is None else min(t_min, e.time_range.start)
    t_max = e.time_range.end if t_max is None else max(t_max, e.time_range.end)
tot = sum(v[0] for v in agg.values())
n_k = sum(v[1] for v in agg.values())
print(f"device kernels: {n_k} launches, busy {tot/1e3:.1f} ms, span {(t_max - t_min)/1e3:.1f} ms "
      f"(idle {(t_max - t_min - tot)/1e3:.1f} ms)")


def family(name):
    if name.startswith("Cijk_") or "gemm" in name.lower() and "conv" not in name.lower():
        return "gemm"
    if "conv" in name.lower() or "igemm" in name or "Sp3Asm" in name or "transpose" in name.lower():
        return "conv"
    if "msda_" in name:
        return "msda"
    if "Memcpy" in name or "copyBuffer" in name or "Memset" in name:
        return "copy"
    if "reduce_kernel" in name or "layer_norm" in name or "softmax" in name.lower():
        return "reduce/norm"
    if "multi_tensor" in name or "adam" in name.lower():
        return "optimizer"
    return "elementwise/other"


fam = defaultdict(lambda: [0.0, 0])
for k, (d, n) in agg.items():
    fam[family(k)][0] += d
    fam[family(k)][1] += n
for k, (d, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:20s} {d/1e3:8.2f} ms {100*d/tot:5.1f}%  n={n}")
print("top kernels:")
for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{d/1e3:9.2f} ms {100*d/tot:5.1f}%  n={n:5d}  avg {d/n:8.1f} us  {k[:120]}")
