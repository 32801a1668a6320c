#!/usr/bin/env python
"""Round 6 (late): the device-to-device MEMCPYs of one clip train step with the graphs off (inside a capture each becomes a
memcpy NODE: ~27 us of host time per replay against 2.8 us for a kernel node, tools/graph_launch_probe.py) -- by the
operator that issued them."""
import collections
import os
import sys

os.environ["MEMOTR_DECODER_GRAPHS"] = "0"
os.environ["MEMOTR_UPDATER_GRAPHS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
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
for _ in range(2):
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)
    torch.cuda.synchronize()
ev = prof.events()
tab = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    n = sum(1 for k in e.kernels if "Memcpy DtoD" in k.name)
    if n:
        tab[(e.name, str(e.input_shapes)[:90], "backward" if e.is_async or e.thread != ev[0].thread else "forward")] += n
print("device-to-device memcpys of one step:", sum(tab.values()))
for (name, shapes, where), n in sorted(tab.items(), key=lambda kv: -kv[1]):
    print(f"{n:5d}  {where:8s} {name}  {shapes}")
