#!/usr/bin/env python
"""Round 5: who launches the element-wise debris of the clip train step?  torch.profiler with the decoder graphs OFF (so
every operator is attributable), `add` / `fill` / `copy` / `mul` kernels grouped by the autograd node (or forward
module call) they run under and by input shape.

    python tools/add_census.py [--out gpurun_out/add_census.txt] [--ops add,fill,copy]
"""
import argparse
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MEMOTR_DECODER_GRAPHS", os.environ.get("CENSUS_GRAPHS", "0"))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import (build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip,  # noqa: E402
                               optimizer_step)
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/add_census.txt")
ap.add_argument("--ops", default="aten::add,aten::add_,aten::fill_,aten::zero_,aten::copy_,aten::mul,aten::mul_,aten::sum,"
                                 "aten::cat,aten::zeros,aten::zeros_like,aten::clone,aten::index_select,aten::index_add_")
args = ap.parse_args()
OPS = set(args.ops.split(","))
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)


def step():
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step()
    torch.cuda.synchronize()


def top_parent(e):
    """Nearest ancestor that names an autograd node, else the outermost aten op."""
    p, best = e.cpu_parent, None
    while p is not None:
        n = p.name
        if n.startswith("autograd::engine::evaluate_function") or "AccumulateGrad" in n or n.startswith("Optimizer"):
            return n.replace("autograd::engine::evaluate_function: ", "bwd ")
        best = n
        p = p.cpu_parent
    return "fwd " + (best or "(top)")


by_parent = collections.defaultdict(lambda: [0, 0.0])
by_shape = collections.defaultdict(lambda: [0, 0.0])
tot = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name not in OPS:
        continue
    ks = getattr(e, "kernels", None) or []
    if not ks:
        continue
    t = sum(k.duration for k in ks)
    par = top_parent(e)
    by_parent[(e.name, par)][0] += len(ks)
    by_parent[(e.name, par)][1] += t
    by_shape[(e.name, par, str(e.input_shapes)[:80])][0] += len(ks)
    by_shape[(e.name, par, str(e.input_shapes)[:80])][1] += t
    tot[e.name][0] += len(ks)
    tot[e.name][1] += t
lines = ["# tools/add_census.py: element-wise kernels of ONE clip train step (decoder graphs %s), by operator" %
         ("off" if os.environ["MEMOTR_DECODER_GRAPHS"] == "0" else "on")]
for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"{k:22s} {n:6d} kernels {t/1e3:8.2f} ms")
lines.append("")
lines.append("# by (operator, enclosing autograd node)")
for (op, par), (n, t) in sorted(by_parent.items(), key=lambda kv: -kv[1][0])[:70]:
    lines.append(f"{n:6d} {t/1e3:8.2f} ms  {op:16s} {par[:90]}")
lines.append("")
lines.append("# by (operator, node, input shapes)")
for (op, par, shp), (n, t) in sorted(by_shape.items(), key=lambda kv: -kv[1][0])[:90]:
    lines.append(f"{n:6d} {t/1e3:8.2f} ms  {op:14s} {par[:52]:52s} {shp}")
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
open(args.out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
