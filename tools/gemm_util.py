#!/usr/bin/env python
"""MFMA utilisation of the projection / FFN / attention GEMMs of the clip train step (run on the GPU box).

torch.profiler with shapes: for every aten::mm / addmm / bmm call shape, FLOPs = 2*M*N*K (x batch) over the device
time of the kernels it launched, against the fp32 MFMA peak of MI355X (157.3 TFLOP/s -- the reference trains in strict
fp32, main.py:96-97, so that is the roofline of these GEMMs; bf16 would be 2.5 PFLOP/s).
Writes a markdown table (default profiles/r02_gemm_mfma_utilisation.md)."""
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

PEAK_TF = 157.3
out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/profiles/r02_gemm_mfma_utilisation.md"
os.environ.setdefault("MEMOTR_DECODER_GRAPHS", "0")     # ops inside a replayed graph carry no shapes
configure_blas()
torch.backends.cuda.matmul.allow_tf32 = False
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


rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key not in ("aten::mm", "aten::addmm", "aten::bmm") or e.device_time_total <= 0:
        continue
    f = flops(e.key, e.input_shapes)
    if not f:
        continue
    t = e.device_time_total / e.count * 1e-6
    rows.append((e.device_time_total / 1e3, e.count, e.key, str(e.input_shapes)[:70], f / 1e9, t * 1e6, f / t / 1e12))
rows.sort(key=lambda r: -r[0])
tot_t = sum(r[0] for r in rows)
tot_f = sum(r[4] * r[1] for r in rows)
lines = ["# GEMM MFMA utilisation in the DanceTrack clip train step (fp32, 1 x MI355X)", "",
         f"torch.profiler (roctracer) with shapes; peak = {PEAK_TF} TFLOP/s (fp32 MFMA = fp32 vector peak on gfx950; no TF32).",
         f"All mm/addmm/bmm calls of one step: {tot_f/1e3:.2f} TFLOP in {tot_t:.1f} ms of kernel time = "
         f"**{tot_f/tot_t:.1f} TFLOP/s = {100*tot_f/tot_t/PEAK_TF:.0f} % of peak** overall.", "",
         "| total ms | calls | op | input shapes | GFLOP/call | us/call | TFLOP/s | % of fp32 MFMA peak |", "|---|---|---|---|---|---|---|---|"]
for r in rows[:40]:
    lines.append(f"| {r[0]:.2f} | {r[1]} | {r[2]} | `{r[3]}` | {r[4]:.2f} | {r[5]:.1f} | {r[6]:.1f} | {100*r[6]/PEAK_TF:.0f} % |")
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
with open(out_path, "w") as f:
    f.write("\n".join(lines) + "\n")
print("\n".join(lines[:30]))
