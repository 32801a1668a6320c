#!/usr/bin/env python
"""Round 6 (late): what does hipGraphLaunch cost the host per node, and what does a memcpy node add?  A chain of small
kernels, with and without device-to-device copies in between (`Tensor.copy_` of a contiguous tensor is captured as a
MEMCPY node), replayed 200 times; host time per replay from perf_counter, graph time from events.

    python tools/graph_launch_probe.py   (DEBUG_CLR_GRAPH_PACKET_CAPTURE as set by the caller; default 0 via memotr_amd)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import memotr_amd  # noqa: F401,E402  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 unless the caller chose)
import torch  # noqa: E402
from memotr_amd.models.decoder_graphs import graph_node_census  # noqa: E402


def build(n_kernels, n_copies, big_copy=False):
    x = torch.zeros(320, 256, device="cuda")
    y = [torch.zeros_like(x) for _ in range(max(n_copies, 1))]
    src = torch.zeros(22323, 256, device="cuda") if big_copy else x
    dst = torch.zeros_like(src)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        every = max(n_kernels // max(n_copies, 1), 1)
        c = 0
        for i in range(n_kernels):
            x.add_(1.0)
            if n_copies and i % every == every - 1 and c < n_copies:
                (dst if big_copy else y[c]).copy_(src if big_copy else x)
                c += 1
    return g, graph_node_census(g)


def timed(g, reps=200):
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = 0.0
    e0.record()
    for _ in range(reps):
        t0 = time.perf_counter()
        g.replay()
        host += time.perf_counter() - t0
    e1.record()
    torch.cuda.synchronize()
    return host / reps * 1e6, e0.elapsed_time(e1) / reps * 1e3


print(f"# DEBUG_CLR_GRAPH_PACKET_CAPTURE={os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')}; host us per replay | wall us per replay")
for nk, nc, big in ((50, 0, False), (180, 0, False), (270, 0, False), (500, 0, False), (270, 5, False), (270, 20, False),
                    (270, 5, True)):
    try:
        g, census = build(nk, nc, big)
        h, w = timed(g)
        print(f"{nk:4d} kernels + {nc:2d} copies{' (22 MB each)' if big else ''}: host {h:7.1f} us  wall {w:7.1f} us  "
              f"({h / (nk + nc):.2f} us host per node)  nodes {census}")
    except Exception as exc:  # noqa: BLE001
        print(f"{nk} + {nc}: failed: {exc}")
