#!/usr/bin/env python
"""Do a chain of small kernels on one HIP stream and large GEMMs on another overlap on this GPU, and do stream
priorities matter?  (Background for engine.encode_stream; run on the GPU box.)"""
import time

import torch

dev = torch.device("cuda", 0)
x = torch.randn(320, 256, device=dev)
a = torch.randn(44646, 256, device=dev)
w = torch.randn(256, 2048, device=dev)
print("priority range:", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")


def small_chain(n=1500):
    y = x
    for _ in range(n):
        y = y + 1.0
    return y


def big_chain(n=40):
    for _ in range(n):
        out = a @ w
    return out


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


g = torch.cuda.CUDAGraph()
s0 = torch.cuda.Stream()
with torch.cuda.stream(s0):
    small_chain(10)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s0):
        small_chain()
for _ in range(2):
    g.replay()
    big_chain()
print(f"small chain (graph replay, 1500 adds): {timed(g.replay):7.2f} ms")
print(f"big chain (40 GEMMs 44646x256x2048):    {timed(big_chain):7.2f} ms")


def both(side):
    def run():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            big_chain()
        g.replay()
        torch.cuda.current_stream().wait_stream(side)
    return run


def chain_under_load(side, main):
    """Duration of the small chain itself (events on its stream) while the GEMMs run on the other stream."""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        big_chain()
    with torch.cuda.stream(main):
        small_chain(3)                     # let the GEMMs get going first
        e0.record(main)
        g.replay()
        e1.record(main)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for prio_side, prio_main in ((0, 0), (-1, 0), (0, -1)):
    side, main = torch.cuda.Stream(priority=prio_side), torch.cuda.Stream(priority=prio_main)
    both(side)()
    with torch.cuda.stream(main):
        total = timed(both(side))
    print(f"GEMM stream priority {prio_side:2d}, chain stream priority {prio_main:2d}: total {total:7.2f} ms, "
          f"the chain itself {chain_under_load(side, main):7.2f} ms")

# eager small chain (host-issued) against side-stream GEMMs
side = torch.cuda.Stream()


def eager_both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        big_chain()
    small_chain()
    torch.cuda.current_stream().wait_stream(side)


eager_both()
print(f"eager small chain alone: {timed(small_chain):7.2f} ms; with side-stream GEMMs: {timed(eager_both):7.2f} ms")
