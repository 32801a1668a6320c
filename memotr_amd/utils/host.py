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


_AFFINITY_BEFORE_PIN = None      # what pin_near_gpu narrowed (unpin() restores it)


def unpin():
    """Give every thread of the process the affinity mask it had before ``pin_near_gpu`` (a CPU-side measurement -- the
    bench's host baseline -- wants all the cores the container may use)."""
    global _AFFINITY_BEFORE_PIN
    if _AFFINITY_BEFORE_PIN is None:
        return
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), _AFFINITY_BEFORE_PIN)
        except OSError:
            pass
    _AFFINITY_BEFORE_PIN = None


def _parse_cpulist(text: str):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_near_gpu(device_index: int = 0, local_rank: int = 0, n_cpus: int = 2):
    """Pin this process (its launch thread, autograd's thread and whatever it starts later) to ``n_cpus`` CPUs of the NUMA
    node the GPU hangs off, a different pair per local rank.

    Why (round 6, ``profiles/r06_host_cpu_probe.txt``, the clip train step on a 2-socket EPYC 9575F box, GPU on node 0):
    left to the scheduler the process floats over 256 hardware threads of two sockets -- 38.99 frames/s; ``taskset -c 0,1``
    39.78, one CPU 39.62, the SMT pair (0, 128) 39.72, four or eight CPUs 39.1-39.2, two CPUs of the OTHER socket 38.88, a
    pair split across the sockets 38.06.  The step issues ~9 k launches from two threads that hand work to the HIP
    runtime's own threads through shared memory: keeping them on one L3 next to the GPU's PCIe root is worth 2 %, and at
    eight ranks per node it is what keeps a rank's threads off the other ranks' cores.  ``MEMOTR_PIN_CPUS=0`` switches it
    off; an affinity mask somebody already narrowed (<= 8 CPUs) is left alone.  Returns the CPUs chosen, or None."""
    if os.environ.get("MEMOTR_PIN_CPUS", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        if len(allowed) <= 8:
            return None
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            local = [c for c in _parse_cpulist(f.read()) if c in set(allowed)]
        if len(local) < n_cpus:
            return None
        start = (local_rank * n_cpus) % (len(local) - n_cpus + 1)
        chosen = local[start:start + n_cpus]
        global _AFFINITY_BEFORE_PIN
        _AFFINITY_BEFORE_PIN = set(allowed)
        # every thread that exists already (the HIP runtime's, torch's pool: the device had to be initialised to be asked
        # where it sits) and, through inheritance, every later one
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), chosen)
            except OSError:
                pass
        return chosen
    except Exception:       # (no sysfs, an exotic topology, a sandbox that forbids it: speed only)
        return None
