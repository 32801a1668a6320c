#!/usr/bin/env python
"""Kernel census by module: for every nn.Module (exclusive of its child modules) and every criterion / engine phase,
the number of device kernels its forward launches and the number its backward nodes launch (linked through autograd
sequence numbers).  Decoder hipGraphs off (a replay hides its kernels from the attribution); run on the GPU box."""
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


def label_module(name, mod):
    stack = []

    def pre(_m, _a, _k=None):
        rf = record_function("M:" + name)
        rf.__enter__()
        stack.append(rf)

    def post(_m, _a, _o):
        stack.pop().__exit__(None, None, None)

    mod.register_forward_pre_hook(pre)
    mod.register_forward_hook(post, always_call=True)


for name, mod in model.named_modules():
    if name:
        label_module(f"{name} [{type(mod).__name__}]", mod)


def phase(obj, attr, label):
    fn = getattr(obj, attr)

    def wrapped(*a, **k):
        with record_function("M:" + label):
            return fn(*a, **k)

    setattr(obj, attr, wrapped)


phase(model, "decode_frame", "memotr.decode_frame (own ops)")
phase(model, "encode_frame", "memotr.encode_frame (own ops)")
phase(model.transformer, "encode", "transformer.encode (own ops)")
phase(model.transformer, "decode", "transformer.decode (own ops)")
phase(criterion, "begin_frame", "criterion.begin_frame")
phase(criterion, "finish_tracks", "criterion.finish_tracks")
phase(criterion, "finish_losses", "criterion.finish_losses")
phase(criterion, "get_mean_by_n_gts", "criterion.get_mean_by_n_gts")
phase(criterion, "get_sum_loss_dict", "criterion.get_sum_loss_dict")
phase(model, "postprocess_single_frame", "memotr.postprocess_single_frame (own ops)")


def step():
    engine.clip_forward_backward(model, criterion, batch, dev)
    with record_function("M:optimizer_step"):
        engine.optimizer_step(model, opt, 0.1)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
events = prof.events()


def label_of(e):
    p_ = e
    while p_ is not None:
        if p_.name.startswith("M:"):
            return p_.name[2:]
        p_ = p_.cpu_parent
    return "(outside)"


def in_backward(e):
    p_ = e
    while p_ is not None:
        if p_.name.startswith("autograd::engine"):
            return True
        p_ = p_.cpu_parent
    return False


def subtree(e):
    n, t = len(e.kernels), sum(k.duration for k in e.kernels)
    for c in e.cpu_children:
        cn, ct = subtree(c)
        n += cn
        t += ct
    return n, t


import re  # noqa: E402


def generic(label):      # layers.3.x -> layers.*.x : one row per kind of module
    return re.sub(r"\.\d+(?=[.\s\[])", ".*", label)


per = collections.defaultdict(lambda: [0, 0.0, 0, 0.0, collections.Counter()])
seq_label = {}
for e in events:
    if e.device_type != DeviceType.CPU or e.name.startswith("M:") or in_backward(e):
        continue
    lab = generic(label_of(e))
    if e.sequence_nr is not None and e.sequence_nr >= 0:
        seq_label.setdefault(e.sequence_nr, lab)
    if e.kernels:
        a = per[lab]
        a[0] += len(e.kernels)
        a[1] += sum(k.duration for k in e.kernels)
unlinked = collections.Counter()
for e in events:
    if e.device_type == DeviceType.CPU and e.name.startswith("autograd::engine::evaluate_function: "):
        n, t = subtree(e)
        if not n:
            continue
        node = e.name.split(": ", 1)[1]
        lab = seq_label.get(e.sequence_nr)
        if lab is None:
            unlinked[node] += n
            lab = "(backward node without a forward operator in the trace)"
        a = per[lab]
        a[2] += n
        a[3] += t
        a[4][node] += n
print(f"{'fwd k':>6s} {'bwd k':>6s} {'fwd ms':>7s} {'bwd ms':>7s}  module (exclusive)")
tot = [0, 0, 0.0, 0.0]
for lab, (fn, ft, bn, bt, nodes) in sorted(per.items(), key=lambda kv: -(kv[1][0] + kv[1][2])):
    top = ", ".join(f"{k.replace('Backward', 'B')}:{v}" for k, v in nodes.most_common(6))
    print(f"{fn:6d} {bn:6d} {ft / 1e3:7.2f} {bt / 1e3:7.2f}  {lab[:64]:64s} {top[:110]}")
    tot[0] += fn
    tot[1] += bn
    tot[2] += ft
    tot[3] += bt
print(f"total: fwd {tot[0]} kernels {tot[2] / 1e3:.1f} ms, bwd {tot[1]} kernels {tot[3] / 1e3:.1f} ms")
print("unlinked backward nodes:", dict(unlinked.most_common(12)))
