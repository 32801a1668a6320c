#!/usr/bin/env python
"""Kernel census of one clip train step: how many device kernels (and how much device time) each phase of the
forward and each kind of autograd node of the backward launches.  Decoder hipGraphs are switched off for the census
(a graph replay hides its kernels from the per-operator attribution); run on the GPU box."""
import collections
import os
import sys

os.environ.setdefault("MEMOTR_DECODER_GRAPHS", "0")
import torch  # noqa: E402
from torch.autograd import DeviceType  # noqa: E402
from torch.profiler import ProfilerActivity, profile, record_function  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd import engine  # noqa: E402
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402

cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = engine.build_optimizer(cfg, model)
batch = engine.clip_to_device(engine.make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)


def phase(obj, attr, label):
    fn = getattr(obj, attr)

    def wrapped(*a, **k):
        with record_function("P:" + label):
            return fn(*a, **k)

    setattr(obj, attr, wrapped)


phase(model.backbone, "forward", "backbone")
phase(model.transformer, "encode", "encoder")
phase(model.transformer, "decode", "decoder")
phase(model, "decode_frame", "decode_frame (queries+decoder+heads)")
phase(model, "encode_frame", "encode_frame (backbone+proj+encoder)")
phase(criterion, "begin_frame", "criterion.begin")
phase(criterion, "finish_frame", "criterion.finish")
phase(model, "postprocess_single_frame", "query_updater")
phase(criterion, "get_mean_by_n_gts", "criterion.mean")


def step():
    loss, _ = engine.clip_forward_backward(model, criterion, batch, dev, backward=False)
    with record_function("P:backward"):
        loss.backward()
    with record_function("P:optimizer_step"):
        engine.optimizer_step(model, opt, 0.1)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()

events = prof.events()


def subtree(e):
    n, t = len(e.kernels), sum(k.duration for k in e.kernels)
    for c in e.cpu_children:
        cn, ct = subtree(c)
        n += cn
        t += ct
    return n, t


total_n = sum(1 for e in events if e.device_type == DeviceType.CUDA and "Memcpy" not in e.name and "Memset" not in e.name)
total_t = sum(e.time_range.end - e.time_range.start for e in events if e.device_type == DeviceType.CUDA)
print(f"device activities: {total_n} kernels, {total_t / 1e3:.1f} ms busy (decoder graphs off)")
print("\nforward phases (inclusive of nested phases):")
agg = collections.OrderedDict()
for e in events:
    if e.device_type == DeviceType.CPU and e.name.startswith("P:"):
        n, t = subtree(e)
        a = agg.setdefault(e.name[2:], [0, 0.0, 0])
        a[0] += n
        a[1] += t
        a[2] += 1
for k, (n, t, c) in agg.items():
    print(f"  {k:46s} calls {c:3d}  kernels {n:6d}  device {t / 1e3:8.2f} ms")

print("\nbackward, by autograd node (top by kernel count):")
bw = collections.defaultdict(lambda: [0, 0.0, 0])
for e in events:
    if e.device_type == DeviceType.CPU and e.name.startswith("autograd::engine::evaluate_function: "):
        n, t = subtree(e)
        a = bw[e.name.split(": ", 1)[1]]
        a[0] += n
        a[1] += t
        a[2] += 1
for k, (n, t, c) in sorted(bw.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"  {k:46s} nodes {c:5d}  kernels {n:6d}  device {t / 1e3:8.2f} ms")
print(f"  total backward: {sum(v[0] for v in bw.values())} kernels, {sum(v[1] for v in bw.values()) / 1e3:.1f} ms")

print("\nforward, by operator inside the decoder phase (top by kernel count):")
ops = collections.defaultdict(lambda: [0, 0.0, 0])


def walk(e, inside):
    inside = inside or e.name == "P:decoder"
    if inside and e.kernels:
        a = ops[e.name]
        a[0] += len(e.kernels)
        a[1] += sum(k.duration for k in e.kernels)
        a[2] += 1
    for c in e.cpu_children:
        walk(c, inside)


for e in events:
    if e.device_type == DeviceType.CPU and e.cpu_parent is None:
        walk(e, False)
for k, (n, t, c) in sorted(ops.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"  {k:46s} calls {c:5d}  kernels {n:6d}  device {t / 1e3:8.2f} ms")
