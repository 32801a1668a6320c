#!/usr/bin/env python
"""Host-side (python) profile of the clip forward: where the launch-bound time goes.  Run on the GPU box."""
import cProfile
import os
import pstats
import sys

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
pr = cProfile.Profile()
n = 3
for _ in range(n):
    torch.cuda.synchronize()
    pr.enable()
    loss, _ = clip_forward_backward(model, criterion, batch, dev, backward=False)
    pr.disable()
    loss.backward()
    optimizer_step(model, opt, 0.1)
st = pstats.Stats(pr)
st.sort_stats("cumulative")
print(f"(totals over {n} clip forwards)")
st.print_stats(70)
st.sort_stats("tottime")
st.print_stats(35)
