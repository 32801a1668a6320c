#!/usr/bin/env python
"""Round 6 (late): the autograd graph of one clip forward at full size (decoder / updater graphs on: a captured callable is
ONE node) -- node types by count.  What to look for: Slice / Select / Narrow backward nodes on tensors that carry a
gradient (each is a zero-fill of the WHOLE source + a copy, and every further use of the source an add), Index /
IndexSelect, Cat / Stack / Split."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip, optimizer_step  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import build as build_criterion  # noqa: E402

cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)
for _ in range(2):
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)
loss, _ = clip_forward_backward(model, criterion, batch, dev, backward=False)
seen, todo = set(), [loss.grad_fn]
cnt = collections.Counter()
shapes = collections.Counter()
while todo:
    fn = todo.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    name = type(fn).__name__
    cnt[name] += 1
    if any(k in name for k in ("Slice", "Select", "Narrow", "Index", "Unbind", "Split", "Cat", "Stack", "Roll", "Expand", "Repeat")):
        meta = []
        for attr in ("_saved_self_sym_sizes", "_saved_self_sizes", "_saved_dim", "_saved_start", "_saved_end", "_saved_index"):
            if hasattr(fn, attr):
                try:
                    meta.append(f"{attr[7:]}={tuple(getattr(fn, attr)) if hasattr(getattr(fn, attr), '__iter__') else getattr(fn, attr)}")
                except Exception:  # noqa: BLE001
                    pass
        shapes[(name, " ".join(meta)[:100])] += 1
    todo += [f for f, _ in fn.next_functions]
print(len(seen), "autograd nodes in the clip's graph")
for k, v in cnt.most_common(60):
    print(f"{v:5d} {k}")
print("\n# slicing / indexing nodes by source shape")
for (k, m), v in sorted(shapes.items(), key=lambda kv: -kv[1])[:70]:
    print(f"{v:5d} {k:28s} {m}")
loss.backward()
