#!/usr/bin/env python
"""Feasibility + gain of capturing one decoder iteration (forward + backward) in hipGraphs (run on the GPU box)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.decoder_graphs import DecoderStep  # noqa: E402
from memotr_amd.modules.linear import configure_blas  # noqa: E402
from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor  # noqa: E402

configure_blas()
cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
frame = tensor_list_to_nested_tensor([torch.randn(3, 800, 1333, device=dev)])
with torch.no_grad():
    enc = model(frame=frame, stage="encode")
dec = model.transformer.decoder
Nq = 320
for lid in (1, 0):
    step = DecoderStep(dec, lid, enc["spatial_shapes"], enc["level_start_index"])
    g = torch.Generator(device=dev).manual_seed(lid)
    args = (torch.randn(1, Nq, 256, device=dev, generator=g).requires_grad_(True),
            torch.rand(1, Nq, 4, device=dev, generator=g).mul(0.6).add(0.2).requires_grad_(lid == 0),
            enc["memory"].detach().clone().requires_grad_(True),
            torch.cat([enc["valid_ratios"], enc["valid_ratios"]], -1)[:, None].contiguous(),
            torch.zeros(1, Nq, dtype=torch.bool, device=dev),
            enc["mask_flatten"].clone())
    args[4][:, 310:] = True

    def run(fn, a):
        out, ref = fn(*a)
        loss = (out * out).sum() * 1e-3 + ref.sum()
        grads = torch.autograd.grad(loss, [x for x in a if x.requires_grad] + list(step.parameters()), allow_unused=True)
        return out.detach().clone(), ref.detach().clone(), [None if x is None else x.detach().clone() for x in grads]

    def timeit(fn, a, n=20):
        for _ in range(3):
            run(fn, a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            run(fn, a)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return t_host / n * 1e3, (time.perf_counter() - t0) / n * 1e3

    o1, r1, g1 = run(step, args)
    he, we = timeit(step, args)
    sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
    t0 = time.perf_counter()
    graphed = torch.cuda.make_graphed_callables(DecoderStep(dec, lid, enc["spatial_shapes"], enc["level_start_index"]),
                                                sample, num_warmup_iters=2, allow_unused_input=True)
    torch.cuda.synchronize()
    print(f"layer {lid}: capture took {time.perf_counter() - t0:.2f} s")
    o2, r2, g2 = run(graphed, args)
    hg, wg = timeit(graphed, args)
    err_o = float((o1 - o2).abs().max()); err_r = float((r1 - r2).abs().max())
    err_g = max(float((a - b).abs().max() / (a.abs().max() + 1e-12)) for a, b in zip(g1, g2) if a is not None)
    print(f"layer {lid}: eager host {he:.2f} ms wall {we:.2f} ms | graphed host {hg:.2f} ms wall {wg:.2f} ms | "
          f"max err out {err_o:.2e} ref {err_r:.2e} grads(rel) {err_g:.2e}")
