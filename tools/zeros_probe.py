import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, "/root/repo")
import torch
from memotr_amd.models import decoder_graphs as dg
dev = torch.device("cuda")
for shape in [(1, 22323, 6, 8, 32), (1, 22323, 8, 32), (6,), (1, 1000, 6, 8, 32)]:
    census = []
    x = torch.ones(16, device=dev)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        z = torch.zeros(shape, device=dev)
    torch.cuda.current_stream().wait_stream(s)
    with dg._thread_local_capture(census):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = x + 1
            z = torch.zeros(shape, device=dev)
            z2 = torch.empty(shape, device=dev).fill_(0.0)
            w = z.sum() + y.sum()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print(shape, census, f"{(time.perf_counter() - t0) * 100:.3f} ms per replay")
