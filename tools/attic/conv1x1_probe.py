#!/usr/bin/env python
"""Round 5: the backbone's 1 x 1 stride-1 convolutions (30 per clip forward, 13.6 ms per train step as MIOpen's
rocBLAS-GEMM path: `Cijk_Ailk_Bljk_SB_MT128x64x16`, profiles/r05a) against the same product written as a batched
matmul on the NCHW tensor (W (Co, Ci) @ x (N, Ci, H*W)) through torch's BLAS choice, forward and forward + backward, at
the shapes of a 5-frame 800 x 1333 clip.

    python tools/conv1x1_probe.py [--out gpurun_out/conv1x1_probe.txt]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.modules.linear import configure_blas  # noqa: E402

SHAPES = [  # (name, Ci, Co, H, W, count per forward, trains weights / needs input gradient)
    ("layer1 64->64", 64, 64, 200, 336, 1, False), ("layer1 256->64", 256, 64, 200, 336, 2, False),
    ("layer1 64->256", 64, 256, 200, 336, 4, False),
    ("layer2 256->128", 256, 128, 200, 336, 1, True), ("layer2 512->128", 512, 128, 100, 168, 3, True),
    ("layer2 128->512", 128, 512, 100, 168, 4, True),
    ("layer3 512->256", 512, 256, 100, 168, 1, True), ("layer3 1024->256", 1024, 256, 50, 84, 5, True),
    ("layer3 256->1024", 256, 1024, 50, 84, 6, True),
    ("layer4 1024->512", 1024, 512, 50, 84, 1, True), ("layer4 2048->512", 2048, 512, 25, 42, 2, True),
    ("layer4 512->2048", 512, 2048, 25, 42, 3, True),
]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/conv1x1_probe.txt")
    ap.add_argument("--batch", type=int, default=5)
    args = ap.parse_args()
    configure_blas()
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    lines = [f"# tools/conv1x1_probe.py, batch {args.batch}, fp32, us per call (forward | forward + backward); x count per clip forward",
             f"{'shape':22s} {'conv2d fwd':>11s} {'matmul fwd':>11s} {'conv2d f+b':>11s} {'matmul f+b':>11s} {'bmm-lt f+b':>11s}  per-step saving (ms)"]
    tot = [0.0, 0.0, 0.0, 0.0]
    for name, ci, co, h, w, cnt, train in SHAPES:
        x = torch.randn(args.batch, ci, h, w, device="cuda", requires_grad=train)
        wt = (torch.randn(co, ci, 1, 1, device="cuda") * 0.05).requires_grad_(train)
        g = torch.randn(args.batch, co, h, w, device="cuda")

        def conv_f():
            return F.conv2d(x, wt)

        def mm_f():
            return torch.matmul(wt.view(co, ci), x.view(args.batch, ci, h * w)).view(args.batch, co, h, w)

        def fb(f):
            def run():
                y = f()
                if train:
                    y.backward(g)
                    x.grad = None
                    wt.grad = None
            return run

        def mm_lt():
            prev = torch.backends.cuda.preferred_blas_library()
            torch.backends.cuda.preferred_blas_library("cublaslt")
            try:
                fb(mm_f)()
            finally:
                torch.backends.cuda.preferred_blas_library(prev)

        with torch.no_grad():
            err = float((conv_f() - mm_f()).abs().max())
            t_cf, t_mf = timed(conv_f), timed(mm_f)
        t_cb, t_mb, t_lt = timed(fb(conv_f)), timed(fb(mm_f)), timed(mm_lt)
        best = min(t_mb, t_lt)
        lines.append(f"{name:22s} {t_cf:11.1f} {t_mf:11.1f} {t_cb:11.1f} {t_mb:11.1f} {t_lt:11.1f}  x{cnt}: "
                     f"{(t_cb - best) * cnt / 1e3:6.2f}   (max |diff| {err:.1e})")
        tot[0] += t_cf * cnt
        tot[1] += t_mf * cnt
        tot[2] += t_cb * cnt
        tot[3] += best * cnt
        print(lines[-1], flush=True)
    lines.append(f"per clip: conv2d fwd {tot[0]/1e3:.2f} ms, matmul fwd {tot[1]/1e3:.2f} ms; conv2d fwd+bwd {tot[2]/1e3:.2f} ms, "
                 f"matmul fwd+bwd (best library) {tot[3]/1e3:.2f} ms")
    print(lines[-1])
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
