#!/usr/bin/env python
"""Paired A/B timing of the clip train step inside ONE process (run on the GPU box).

Separate bench.py runs differ by several percent from box to box and with the thermal state of the GPU, which
hides small host-side changes.  This alternates two settings step by step and reports the median of each.
Usage: python tools/ab_step.py [steps_per_setting] [encode-chunk specs ...]   e.g.  8 0 1 5 1,4
A spec may carry environment switches read at run time: "auto@MEMOTR_ENCODE_STREAM=0", "2,3@A=1;B=2".
"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import (build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip,  # noqa: E402
                               optimizer_step)
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
specs = sys.argv[2:] or ["0", "1"]          # MEMOTR_ENCODE_CHUNKS-style strings, e.g. 0 1 5 1,4
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)


def step(spec):
    chunks, _, env = spec.partition("@")
    saved = {}
    for kv in filter(None, env.split(";")):
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        return _step(chunks)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _step(spec):
    model.encode_chunks = spec
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, _ = clip_forward_backward(model, criterion, batch, dev, backward=False)
    t1 = time.perf_counter()                 # host done issuing the forward
    loss.backward()
    t2 = time.perf_counter()                 # host done issuing the backward
    optimizer_step(model, opt, 0.1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3


for _ in range(2):
    for spec in specs:
        step(spec)
res = {spec: [] for spec in specs}
for i in range(n):
    for spec in (specs if i % 2 == 0 else specs[::-1]):
        res[spec].append(step(spec))
for spec in specs:
    wall, fwd, bwd = ([r[k] for r in res[spec]] for k in range(3))
    print(f"encode chunks {spec:32s}: step median {statistics.median(wall):7.1f} ms (min {min(wall):7.1f})   "
          f"host: forward {statistics.median(fwd):6.1f} ms, backward call {statistics.median(bwd):6.1f} ms   "
          f"mem {torch.cuda.max_memory_allocated() >> 20} MiB")
