"""How long does the host wait in a blocking device->host read AFTER the GPU work it depends on has finished?
(round 3: an inference frame blocked ~30 ms in one `nonzero` behind 10 ms of kernels.)

    python tools/sync_latency_probe.py
    HSA_ENABLE_INTERRUPT=0 python tools/sync_latency_probe.py
"""
import os
import time

import torch

dev = torch.device("cuda", 0)
a = torch.randn(4096, 4096, device=dev)
flag = torch.zeros(300, dtype=torch.bool, device=dev)
flag[::7] = True
for _ in range(3):
    (a @ a).sum().item()


def gpu_ms(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); e.synchronize()
    return s.elapsed_time(e)


def work(n):
    def run():
        x = a
        for _ in range(n):
            x = a @ a
        return x
    return run


print("env:", {k: os.environ.get(k) for k in ("HSA_ENABLE_INTERRUPT", "ROC_ACTIVE_WAIT_TIMEOUT", "DEBUG_CLR_GRAPH_PACKET_CAPTURE")})
for n in (0, 4, 16, 64):
    g = gpu_ms(work(n)) if n else 0.0
    res = {}
    for name, wait in (("nonzero", lambda: flag.nonzero()), ("item", lambda: flag.sum().item()),
                       ("stream.synchronize", lambda: torch.cuda.current_stream().synchronize()),
                       ("pinned copy + event", None)):
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            time.sleep(0.02)                      # the GPU idles between frames of an online loop
            t0 = time.perf_counter()
            work(n)()
            if wait is not None:
                wait()
            else:
                h = torch.empty(300, dtype=torch.bool, pin_memory=True)
                h.copy_(flag, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(); ev.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        res[name] = sorted(ts)[2]
    print(f"{n:3d} matmuls = {g:6.2f} ms of kernels; host time until the read returns: " +
          ", ".join(f"{k} {v:6.2f} ms" for k, v in res.items()))
