"""One online-tracking frame on a time axis: host calls (hipGraphLaunch, blocking copies) against the first / last
kernel of each burst on the GPU (torch profiler; run on the GPU box)."""
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
look = "--lookahead" in sys.argv
for i in range(8):
    tracker.step(frames[i % 4], 800, 1333, next_image=frames[(i + 1) % 4] if look else None)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(3):
        tracker.step(frames[i % 4], 800, 1333, next_image=frames[(i + 1) % 4] if look else None)
    torch.cuda.synchronize()
evs = list(prof.events())
gpu = sorted((e for e in evs if e.device_type == DeviceType.CUDA), key=lambda e: e.time_range.start)
cpu = sorted((e for e in evs if e.device_type == DeviceType.CPU and
              e.name in ("hipGraphLaunch", "hipMemcpyWithStream", "hipEventSynchronize", "hipStreamSynchronize",
                         "hipDeviceSynchronize")), key=lambda e: e.time_range.start)
t0 = min(gpu[0].time_range.start, cpu[0].time_range.start)
# GPU bursts: runs of kernels separated by > 300 us of idle
bursts, cur = [], [gpu[0]]
for e in gpu[1:]:
    if e.time_range.start - max(x.time_range.end for x in cur[-8:]) > 300:
        bursts.append(cur)
        cur = []
    cur.append(e)
bursts.append(cur)
rows = [(e.time_range.start - t0, f"HOST {e.name} ({(e.time_range.end - e.time_range.start) / 1e3:.2f} ms)") for e in cpu]
rows += [(b[0].time_range.start - t0, f"GPU  burst of {len(b)} activities, {(max(x.time_range.end for x in b) - b[0].time_range.start) / 1e3:.2f} ms "
          f"(busy {sum(x.time_range.end - x.time_range.start for x in b) / 1e3:.2f}), first: {b[0].name[:50]}") for b in bursts]
for t, s in sorted(rows):
    print(f"{t / 1e3:8.2f} ms  {s}")

# the same loop without the profiler: wall time of consecutive frames, back to back and with idle time between them
import time
for look, pause in ((False, 0.0), (True, 0.0), (False, 0.03)):
    ts = []
    for i in range(30):
        t = time.perf_counter()
        tracker.step(frames[i % 4], 800, 1333, next_image=frames[(i + 1) % 4] if look else None)
        ts.append((time.perf_counter() - t) * 1e3)
        time.sleep(pause)
    print(f"no profiler, lookahead={look}, {pause * 1e3:.0f} ms pause between frames: " + " ".join(f"{x:.0f}" for x in ts)
          + f" | median {sorted(ts)[len(ts) // 2]:.1f} mean {sum(ts) / len(ts):.1f} ms")
