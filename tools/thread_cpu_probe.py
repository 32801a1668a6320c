#!/usr/bin/env python
"""CPU time per THREAD of the process over a few clip train steps (round 6: `host_ms_per_step` reads ~230 ms for a
~128 ms step -- who burns it?).  Reads /proc/self/task/*/stat before and after.   python tools/thread_cpu_probe.py [steps]"""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("MEMOTR_REQUIRE_GRAPHS", "1")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip, optimizer_step  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402
from memotr_amd.utils.host import pin_near_gpu, respect_cpu_quota  # noqa: E402


def threads():
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            with open(f"/proc/self/task/{tid}/stat") as f:
                s = f.read()
            comm = s[s.index("(") + 1:s.rindex(")")]
            rest = s[s.rindex(")") + 2:].split()
            out[int(tid)] = (comm, (int(rest[11]) + int(rest[12])) / tck, int(rest[11]) / tck, int(rest[12]) / tck)
        except OSError:
            pass
    return out


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    respect_cpu_quota()
    print("pinned:", pin_near_gpu(0, 0))
    cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
    dev = torch.device("cuda", 0)
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    model = build_model(cfg).to(dev).train()
    criterion = build_criterion(cfg)
    opt = build_optimizer(cfg, model)
    batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)
    for _ in range(4):
        clip_forward_backward(model, criterion, batch, dev)
        optimizer_step(model, opt, 0.1)
    torch.cuda.synchronize()
    a, t0 = threads(), time.perf_counter()
    for _ in range(steps):
        clip_forward_backward(model, criterion, batch, dev)
        optimizer_step(model, opt, 0.1)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    b = threads()
    rows = []
    for tid, (comm, tot, u, s) in b.items():
        tot0, u0, s0 = (a[tid][1], a[tid][2], a[tid][3]) if tid in a else (0.0, 0.0, 0.0)
        rows.append(((tot - tot0) / steps * 1e3, (u - u0) / steps * 1e3, (s - s0) / steps * 1e3, comm, tid))
    rows.sort(reverse=True)
    print(f"wall {wall:.1f} ms per step; CPU ms per step by thread (user / system):")
    for tot, u, s, comm, tid in rows[:14]:
        print(f"  {tot:8.1f}  ({u:7.1f} / {s:7.1f})  {comm:20s} tid {tid}{'  <- main' if tid == os.getpid() else ''}")
    print(f"  sum {sum(r[0] for r in rows):.1f} ms over {len(rows)} threads")


if __name__ == "__main__":
    main()
