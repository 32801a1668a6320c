#!/usr/bin/env python
"""Decoder-sized fp32 GEMMs (a few hundred rows): GPU time and host issue cost per call, hipBLASLt vs rocBLAS,
and alternative formulations of the weight gradient dW = dY^T X whose (256 x 256) output the hipBLASLt heuristic
maps to ONE 256x256 macro-tile (80 us for 40 MFLOP).  Run on the GPU box."""
import sys
import time

import torch
import torch.nn.functional as F


def bench(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, host


for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    print(f"== preferred_blas_library = {lib}")
    for rows in ([int(a) for a in sys.argv[1:]] or [300, 310, 347, 512]):
        for n_out, k_in in ((256, 256), (2048, 256), (256, 2048), (768, 256), (128, 256)):
            x = torch.randn(rows, k_in, device="cuda")
            w = torch.randn(n_out, k_in, device="cuda")
            b = torch.randn(n_out, device="cuda")
            dy = torch.randn(rows, n_out, device="cuda")
            ref = dy.t() @ x
            forms = {
                "fwd addmm": lambda: F.linear(x, w, b),
                "dgrad mm": lambda: dy @ w,
                "wgrad dy.t()@x": lambda: dy.t() @ x,
                "wgrad (x.t()@dy).t()": lambda: (x.t() @ dy).t(),
                "wgrad blocks8 bmm": lambda: torch.matmul(dy.t().reshape(8, n_out // 8, rows), x).reshape(n_out, k_in),
                "wgrad contiguous dyT": lambda: dy.t().contiguous() @ x,
            }
            line = f"rows {rows:4d} out {n_out:4d} in {k_in:4d}: "
            for name, fn in forms.items():
                if name.startswith("wgrad"):
                    err = float((fn() - ref).abs().max())
                    assert err < 1e-2, (name, err)
                g, h = bench(fn)
                line += f"{name} {g:6.1f}us (host {h:4.1f}) | "
            print(line, flush=True)
