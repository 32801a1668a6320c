#!/usr/bin/env python
"""MFMA utilisation of the projection / FFN / attention GEMMs of the clip train step (run on the GPU box).

    python tools/gemm_util.py [--dtype f32|bf16] [--out profiles/...md] [--json gpurun_out/profiles/gemm_mfma.json]

torch.profiler with shapes: for every aten::mm / addmm / bmm call, FLOPs = 2*M*N*K (x batch) over the device time of
the kernels it launched, against the dense MFMA peak of MI355X for the GEMM's input type: 157.3 TFLOP/s fp32 (the
reference trains in strict fp32, main.py:96-97), 2500 TFLOP/s bf16 (the autocast extension of BASELINE config 5; a call
is counted as bf16 when its library kernel is a bf16 one -- `BBS` / `bf16` in the kernel name).
Writes a markdown table and merges the totals of this dtype into a small JSON that bench.py reports under `kernels`."""
import argparse
import json
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
from memotr_amd.modules.linear import configure_blas  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
ap.add_argument("--out", default=None)
ap.add_argument("--json", default="gpurun_out/profiles/gemm_mfma.json")
ap.add_argument("--tag", default="r05")
args = ap.parse_args()
PEAK = {"f32": 157.3, "bf16": 2500.0}
out_path = args.out or f"gpurun_out/profiles/{args.tag}_gemm_mfma_utilisation_{args.dtype}.md"
os.environ.setdefault("MEMOTR_DECODER_GRAPHS", "0")     # ops inside a replayed graph carry no shapes
os.environ.setdefault("MEMOTR_ENCODE_GRAPHS", "0")
configure_blas()
torch.backends.cuda.matmul.allow_tf32 = False
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)


def step():
    if args.dtype == "bf16":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            clip_forward_backward(model, criterion, batch, dev)
    else:
        clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()


def flops(key, shapes):
    try:
        if key == "aten::mm":
            (m, k), (_, n) = shapes[0], shapes[1]
            return 2.0 * m * n * k
        if key == "aten::addmm":
            (m, k), (_, n) = shapes[1], shapes[2]
            return 2.0 * m * n * k
        if key == "aten::bmm":
            (b, m, k), (_, _, n) = shapes[0], shapes[1]
            return 2.0 * b * m * n * k
    except Exception:  # noqa: BLE001
        return None
    return None


agg = {}
for e in prof.events():
    if e.name not in ("aten::mm", "aten::addmm", "aten::bmm"):
        continue
    ks = getattr(e, "kernels", None) or []
    t_us = sum(k.duration for k in ks)
    if t_us <= 0:
        continue
    f = flops(e.name, e.input_shapes)
    if not f:
        continue
    names = " ".join(k.name for k in ks)
    low = ("BBS" in names) or ("bf16" in names.lower()) or ("_BB_" in names) or ("BF16" in names)
    key = (e.name, str(e.input_shapes)[:70], "bf16" if low else "f32")
    a = agg.setdefault(key, [0.0, 0, f])
    a[0] += t_us
    a[1] += 1
rows = []
for (op, shp, dt), (t_us, n, f) in agg.items():
    t = t_us / n * 1e-6
    rows.append((t_us / 1e3, n, op, shp, f / 1e9, t * 1e6, f / t / 1e12, dt))
rows.sort(key=lambda r: -r[0])
tot = {}
for r in rows:
    d = tot.setdefault(r[7], [0.0, 0.0])
    d[0] += r[0]
    d[1] += r[4] * r[1]
tot_t = sum(v[0] for v in tot.values())
tot_f = sum(v[1] for v in tot.values())
# time-weighted: each GEMM priced against the peak of its own input type
ideal_ms = sum(v[1] / PEAK[k] for k, v in tot.items())       # GFLOP / (TFLOP/s) = ms
lines = [f"# GEMM MFMA utilisation in the DanceTrack clip train step ({args.dtype} step, 1 x MI355X, clip 5, 800x1333)", "",
         "torch.profiler (roctracer) with shapes, decoder / encode graphs off so that every call carries its shape; peaks: "
         "157.3 TFLOP/s fp32 MFMA, 2500 TFLOP/s dense bf16 MFMA (MI355X_MICROARCH.md).",
         f"All mm/addmm/bmm calls of one step: {tot_f/1e3:.2f} TFLOP in {tot_t:.1f} ms of kernel time = "
         f"**{tot_f/tot_t:.1f} TFLOP/s**; priced per input type: " +
         ", ".join(f"{k}: {v[1]/1e3:.2f} TFLOP in {v[0]:.1f} ms = {v[1]/v[0]:.1f} TFLOP/s = "
                   f"**{100*v[1]/v[0]/PEAK[k]:.0f} % of the {k} MFMA peak**" for k, v in sorted(tot.items())) +
         f"; time at peak {ideal_ms:.2f} ms -> **{100*ideal_ms/tot_t:.0f} % of MFMA peak, time-weighted**.", "",
         "| total ms | calls | op | input shapes | type | GFLOP/call | us/call | TFLOP/s | % of that type's MFMA peak |",
         "|---|---|---|---|---|---|---|---|---|"]
for r in rows[:40]:
    lines.append(f"| {r[0]:.2f} | {r[1]} | {r[2]} | `{r[3]}` | {r[7]} | {r[4]:.2f} | {r[5]:.1f} | {r[6]:.1f} | "
                 f"{100*r[6]/PEAK[r[7]]:.0f} % |")
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
with open(out_path, "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines[:30]))
summary = {}
if os.path.exists(args.json):
    try:
        summary = json.load(open(args.json))
    except Exception:  # noqa: BLE001
        summary = {}
summary[args.dtype] = {"gemm_ms_per_step": tot_t, "gemm_tflop_per_step": tot_f / 1e3,
                       "gemm_frac_of_mfma_peak": ideal_ms / tot_t,
                       "by_type": {k: {"ms": v[0], "tflop": v[1] / 1e3, "frac_of_peak": v[1] / v[0] / PEAK[k]}
                                   for k, v in tot.items()},
                       "note": "torch.profiler, graphs off, tools/gemm_util.py", "tag": args.tag}
os.makedirs(os.path.dirname(args.json) or ".", exist_ok=True)
json.dump(summary, open(args.json, "w"), indent=1)
