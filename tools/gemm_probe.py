#!/usr/bin/env python
"""fp32 GEMM shapes of the encoder (S = 22323 rows) under rocBLAS vs hipBLASLt (run on the GPU box)."""
import sys
import time

import torch
import torch.nn.functional as F

S = int(sys.argv[1]) if len(sys.argv) > 1 else 22323
SHAPES = [  # (name, M, K, N, kind)  kind: "fwd" y = x W^T + b ; "dgrad" dx = dy W ; "wgrad" dW = dy^T x
    ("ffn1 fwd", S, 256, 2048), ("ffn2 fwd", S, 2048, 256), ("proj256 fwd", S, 256, 256), ("proj384 fwd", S, 256, 384),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for lib in ("cublaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:  # noqa: BLE001
        print("cannot select", lib, e)
        continue
    print("== preferred_blas_library =", lib)
    for name, M, K, N in SHAPES:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda")
        b = torch.randn(N, device="cuda")
        dy = torch.randn(M, N, device="cuda")
        gf = 2.0 * M * K * N / 1e9
        t_f = timeit(lambda: F.linear(x, w, b))
        t_d = timeit(lambda: dy @ w)
        t_w = timeit(lambda: dy.t() @ x)
        print(f"{name:12s} {gf:6.1f} GF  fwd {t_f:8.1f} us ({gf/t_f*1e3:6.1f} TF/s)  dgrad {t_d:8.1f} us ({gf/t_d*1e3:6.1f})  "
              f"wgrad {t_w:8.1f} us ({gf/t_w*1e3:6.1f})", flush=True)

print("== wgrad via split-K (bmm over row chunks + sum) ==")
torch.backends.cuda.preferred_blas_library("cublaslt")
for name, M, K, N in SHAPES:
    x = torch.randn(M, K, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    ref = dy.t() @ x
    gf = 2.0 * M * K * N / 1e9
    for chunks in (3, 7, 21, 35, 63, 105):
        if M % chunks:
            continue
        r = M // chunks

        def f():
            return torch.bmm(dy.view(chunks, r, N).transpose(1, 2), x.view(chunks, r, K)).sum(0)

        err = float((f() - ref).abs().max() / ref.abs().max())
        t = timeit(f)
        print(f"{name:12s} chunks {chunks:3d}: {t:8.1f} us ({gf/t*1e3:6.1f} TF/s) rel err {err:.1e}", flush=True)
