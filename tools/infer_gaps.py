#!/usr/bin/env python
"""GPU busy time, span and launch count of one online-tracking frame (bench.py --workload infer set-up; run on the GPU
box): how much of a frame is the host issuing kernels."""
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
from torch.autograd import DeviceType
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from memotr_amd import configs as C  # noqa: E402
from memotr_amd.inference import SequenceTracker  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.utils import logits_to_scores  # noqa: E402
from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor  # noqa: E402

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
    tracker.step(frames[i % 4], 800, 1333)
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(N):
        tracker.step(frames[i % 4], 800, 1333)
    torch.cuda.synchronize()
ev = sorted((e for e in prof.events() if e.device_type == DeviceType.CUDA), key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in ev)
span = ev[-1].time_range.end - ev[0].time_range.start
print(f"per frame: {len(ev) / N:.0f} device activities, busy {busy / N / 1e3:.2f} ms, span {span / N / 1e3:.2f} ms")
names = {}
for e in ev:
    names[e.name[:70]] = names.get(e.name[:70], 0.0) + (e.time_range.end - e.time_range.start)
for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v / N / 1e3:7.3f} ms  {k}")
ka = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)
print("top self-CPU operators per frame:")
for e in ka[:14]:
    print(f"  {e.self_cpu_time_total / N / 1e3:7.3f} ms n={e.count / N:6.1f}  {e.key[:80]}")

# host side: where a frame's wall time goes (cProfile over 10 frames, synchronising at the end only)
import cProfile
import pstats
import time
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    tracker.step(frames[i % 4], 800, 1333)
torch.cuda.synchronize()
pr.disable()
print(f"wall per frame with cProfile on: {(time.perf_counter() - t0) * 100:.2f} ms")
st = pstats.Stats(pr)
st.sort_stats("cumulative")
import io
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(45)
for line in buf.getvalue().splitlines():
    if "/" in line or "{" in line:
        print(line[:170])

# wall-clock segments of a frame (no profiler): model launch | tracker.update (blocks on the scores) | query updater
# launch | copy of the tracks to the host
seg = {}


def timed(name, fn):
    def wrapper(*a, **k):
        t = time.perf_counter()
        out = fn(*a, **k)
        seg[name] = seg.get(name, 0.0) + time.perf_counter() - t
        return out
    return wrapper


tracker.tracker.update = timed("tracker.update", tracker.tracker.update)
tracker.core.postprocess_single_frame = timed("postprocess (query updater)", tracker.core.postprocess_single_frame)
tracker._encoded = timed("encode (or take the queued one)", tracker._encoded)
tracker._prefetch = timed("queue next encode", tracker._prefetch)
for look in (False, True):
    for k in list(seg):
        seg[k] = 0.0
    for i in range(4):
        tracker.step(frames[i % 4], 800, 1333, next_image=frames[(i + 1) % 4] if look else None)
    torch.cuda.synchronize()
    for k in list(seg):
        seg[k] = 0.0
    t0 = time.perf_counter()
    n = 20
    for i in range(n):
        tracker.step(frames[i % 4], 800, 1333, next_image=frames[(i + 1) % 4] if look else None)
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n * 1e3
    print(f"lookahead={look}: {total:.2f} ms per frame; " + "; ".join(f"{k} {v / n * 1e3:.2f}" for k, v in seg.items()))

