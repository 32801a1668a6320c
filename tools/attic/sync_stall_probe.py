"""Does a loop of [a few ms of kernels, one blocking read] stall periodically?  Prints every iteration."""
import os
import sys
import time

import torch

dev = torch.device("cuda", 0)
a = torch.randn(2048, 2048, device=dev)
flag = torch.zeros(300, dtype=torch.bool, device=dev)
flag[::7] = True
big = torch.randn(64, 1024, 1024, device=dev)
for _ in range(5):
    (a @ a).sum().item()
mode = sys.argv[1] if len(sys.argv) > 1 else "matmul"
for n in (8, 40):
    ts = []
    for i in range(45):
        t = time.perf_counter()
        for _ in range(n):
            x = a @ a
        if mode == "alloc":          # a frame also allocates and frees: do the same
            tmp = [torch.empty(1 << 22, device=dev) for _ in range(8)]
            del tmp
        flag.nonzero()
        ts.append((time.perf_counter() - t) * 1e3)
    print(f"{mode} x{n}: " + " ".join(f"{x:.1f}" for x in ts))
