#!/usr/bin/env python
"""Is the periodic 65-85 ms stall of the online-tracking loop (profiles/r03_infer_timeline.txt) the container's CPU
quota?  Prints the cgroup's CFS quota and, per frame, the wall time next to the growth of the cgroup's throttle
counters (cpu.stat: nr_throttled / throttled time) and of this process's context switches.

    python tools/infer_stall_probe.py [--threads N] [--frames 60]
"""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=0, help="torch.set_num_threads / OMP_NUM_THREADS (0: leave alone)")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--no-lookahead", action="store_true")
args = ap.parse_args()
if args.threads:
    os.environ["OMP_NUM_THREADS"] = str(args.threads)
    os.environ["MKL_NUM_THREADS"] = str(args.threads)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def read_first(paths):
    for p in paths:
        try:
            with open(p) as f:
                return p, f.read().strip()
        except OSError:
            continue
    return None, None


def throttle():
    _, txt = read_first(["/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat"])
    out = {}
    for ln in (txt or "").splitlines():
        k, _, v = ln.partition(" ")
        if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time", "usage_usec"):
            out[k] = int(v)
    return out


def ctx():
    v = n = 0
    with open("/proc/self/status") as f:
        for ln in f:
            if ln.startswith("voluntary_ctxt"):
                v = int(ln.split()[1])
            elif ln.startswith("nonvoluntary_ctxt"):
                n = int(ln.split()[1])
    return v, n


def main():
    if args.threads:
        torch.set_num_threads(args.threads)
    for p in (["/sys/fs/cgroup/cpu.max"], ["/sys/fs/cgroup/cpu/cpu.cfs_quota_us"], ["/sys/fs/cgroup/cpu/cpu.cfs_period_us"]):
        print(p[0], "=", read_first(p)[1])
    print("os.cpu_count", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)), "torch threads",
          torch.get_num_threads(), "interop", torch.get_num_interop_threads())
    from memotr_amd import configs as C
    from memotr_amd.inference import SequenceTracker
    from memotr_amd.models import build_model
    from memotr_amd.models.utils import logits_to_scores
    from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor
    dev = torch.device("cuda", 0)
    cfg = C.dancetrack_config()
    model = build_model(dict(cfg, DEVICE="cuda", AVAILABLE_GPUS="0")).to(dev).eval()
    tracker = SequenceTracker.from_config(model, cfg)
    tracker.result_score_thresh = 0.0
    g = torch.Generator().manual_seed(1)
    frames = [torch.randn(3, 800, 1333, generator=g).to(dev) for _ in range(4)]
    with torch.no_grad():
        res = model(frame=tensor_list_to_nested_tensor([frames[0]]).to(dev), tracks=tracker.tracks)
        best = logits_to_scores(res["pred_logits"])[0, :len(res["det_query_embed"])].max(-1).values
    tracker.tracker.det_score_thresh = float(best.topk(20).values[-1])
    tracker.tracker.track_score_thresh = 0.0
    tracker.step(frames[0], 800, 1333)
    tracker.tracker.det_score_thresh = 2.0
    for i in range(8):
        tracker.step(frames[i % 4], 800, 1333, next_image=None if args.no_lookahead else frames[(i + 1) % 4])
    torch.cuda.synchronize()
    rows = []
    t_all = time.perf_counter()
    for i in range(args.frames):
        th0, c0 = throttle(), ctx()
        t0 = time.perf_counter()
        tracker.step(frames[i % 4], 800, 1333, next_image=None if args.no_lookahead else frames[(i + 1) % 4])
        dt = (time.perf_counter() - t0) * 1e3
        th1, c1 = throttle(), ctx()
        rows.append((dt, {k: th1.get(k, 0) - th0.get(k, 0) for k in th1}, c1[0] - c0[0], c1[1] - c0[1]))
    total = time.perf_counter() - t_all
    print(f"{args.frames / total:.1f} frames/s back to back (threads={args.threads or 'default'})")
    for dt, th, v, n in rows:
        flag = "  <-- stall" if dt > 40 else ""
        print(f"  {dt:7.1f} ms  throttle {th}  ctxt vol {v} invol {n}{flag}")


if __name__ == "__main__":
    main()
