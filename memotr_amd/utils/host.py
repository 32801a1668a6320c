"""Host-side environment: the CPU quota of the container.

PyTorch sizes its intra-op thread pool from the machine's core count (128 on the 256-thread EPYC hosts of the MI355X
boxes) and its OpenMP workers spin after every parallel region.  Inside a container with a CFS quota (16 CPUs on those
boxes: /sys/fs/cgroup/cpu.max = "1600000 100000") that pool burns the whole quota in a fraction of each 100 ms period
and the kernel then FREEZES every thread of the container for the rest of it -- the host thread that should be issuing
the next frame included.  Measured on the online-tracking loop (tools/infer_stall_probe.py, round 4): 530 ms of CPU time
per 10 ms frame, one frame in three stalled for 75-85 ms with `nr_throttled` counting up in step, 30.7 frames/s;
with the pool capped at 4 threads 95.4 frames/s and no throttling.  (Round 3 chased this as a GPU-runtime stall.)
"""
from __future__ import annotations

import math
import os


def cpu_quota() -> float:
    """CPUs this process may use: the cgroup's CFS quota (v2 cpu.max, v1 cpu.cfs_quota_us) or, without one, the size
    of the scheduler affinity mask."""
    n = float(len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else float(os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, float(quota) / float(period))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = float(f.read())
            if quota > 0:
                n = min(n, quota / period)
        except (OSError, ValueError):
            pass
    return max(1.0, n)


def respect_cpu_quota(reserve: float = 0.5, processes: int = 1) -> int:
    """Cap torch's intra-op pool at `reserve` x this process's share of the quota (`processes` = how many processes
    share the container, e.g. one per GPU); the rest stays with the launching thread, autograd's thread and the HIP
    runtime's own.  Never raises the count.  Returns the thread count in force."""
    import torch
    if os.environ.get("MEMOTR_NO_QUOTA_CAP", "0") == "1":       # (A/B measurements only)
        return torch.get_num_threads()
    want = max(1, int(math.floor(cpu_quota() * reserve / max(1, processes))))
    if torch.get_num_threads() > want:
        torch.set_num_threads(want)
    return torch.get_num_threads()
