#!/usr/bin/env python
"""Which reductions does a clip train step launch?  reduce_kernel time by operator and input shape (decoder graphs
off so that every launch is attributed); run on the GPU box."""
import os
import sys

os.environ.setdefault("MEMOTR_DECODER_GRAPHS", "0")
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd import engine  # noqa: E402
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402

cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = engine.build_optimizer(cfg, model)
batch = engine.clip_to_device(engine.make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)


def step():
    engine.clip_forward_backward(model, criterion, batch, dev)
    engine.optimizer_step(model, opt, 0.1)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.events():
    red = [k for k in e.kernels if "reduce_kernel" in k.name]
    if red:
        rows.append((e.name, str(e.input_shapes)[:90], len(red), sum(k.duration for k in red)))
agg = {}
for name, shp, n, t in rows:
    a = agg.setdefault((name, shp), [0, 0.0])
    a[0] += n
    a[1] += t
tot_n = sum(v[0] for v in agg.values())
tot_t = sum(v[1] for v in agg.values())
print(f"reduce_kernel launches {tot_n}, {tot_t / 1e3:.2f} ms")
for (name, shp), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t / 1e3:7.2f} ms n={n:4d} avg {t / n:7.1f} us  {name:34s} {shp}")
