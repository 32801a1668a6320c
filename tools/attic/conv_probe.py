#!/usr/bin/env python
"""Time the ResNet-50 / feature-projection convolutions of one 800x1344 frame on MIOpen (fwd and fwd+bwd),
to find shapes that fall onto slow solvers.  Run on the GPU box."""
import sys
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = len(sys.argv) > 1 and sys.argv[1] == "find"
CL = len(sys.argv) > 2 and sys.argv[2] == "cl"

# (name, Cin, Cout, k, stride, pad, Hin, Win)
SHAPES = [
    ("conv1 7x7s2", 3, 64, 7, 2, 3, 800, 1344),
    ("l1 1x1 64->64", 64, 64, 1, 1, 0, 200, 336),
    ("l1 3x3 64->64", 64, 64, 3, 1, 1, 200, 336),
    ("l1 1x1 64->256", 64, 256, 1, 1, 0, 200, 336),
    ("l2.0 1x1 256->128", 256, 128, 1, 1, 0, 200, 336),
    ("l2.0 3x3s2 128", 128, 128, 3, 2, 1, 200, 336),
    ("l2.0 ds 1x1s2 256->512", 256, 512, 1, 2, 0, 200, 336),
    ("l2 3x3 128", 128, 128, 3, 1, 1, 100, 168),
    ("l2 1x1 128->512", 128, 512, 1, 1, 0, 100, 168),
    ("l3.0 3x3s2 256", 256, 256, 3, 2, 1, 100, 168),
    ("l3.0 ds 1x1s2 512->1024", 512, 1024, 1, 2, 0, 100, 168),
    ("l3 3x3 256", 256, 256, 3, 1, 1, 50, 84),
    ("l3 1x1 256->1024", 256, 1024, 1, 1, 0, 50, 84),
    ("l4.0 3x3s2 512", 512, 512, 3, 2, 1, 50, 84),
    ("l4.0 ds 1x1s2 1024->2048", 1024, 2048, 1, 2, 0, 50, 84),
    ("l4 3x3 512", 512, 512, 3, 1, 1, 25, 42),
    ("l4 1x1 512->2048", 512, 2048, 1, 1, 0, 25, 42),
    ("proj0 1x1 512->256", 512, 256, 1, 1, 0, 100, 168),
    ("proj2 1x1 2048->256", 2048, 256, 1, 1, 0, 25, 42),
    ("proj3 3x3s2 2048->256", 2048, 256, 3, 2, 1, 25, 42),
]


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, ci, co, k, s, p, h, w in SHAPES:
    x = torch.randn(1, ci, h, w, device="cuda", requires_grad=True)
    wt = torch.randn(co, ci, k, k, device="cuda", requires_grad=True)
    if CL:
        x = x.detach().to(memory_format=torch.channels_last).requires_grad_(True)
        wt = wt.detach().to(memory_format=torch.channels_last).requires_grad_(True)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    gflop = 2.0 * ho * wo * co * ci * k * k / 1e9

    def fwd():
        return F.conv2d(x, wt, None, s, p)

    def fwdbwd():
        y = F.conv2d(x, wt, None, s, p)
        y.backward(torch.ones_like(y))

    tf = timeit(fwd)
    tb = timeit(fwdbwd)
    print(f"{name:28s} {gflop:7.2f} GF  fwd {tf:8.3f} ms ({gflop/tf:7.1f} TF/s)  fwd+bwd {tb:8.3f} ms ({3*gflop/tb:7.1f} TF/s)",
          flush=True)
