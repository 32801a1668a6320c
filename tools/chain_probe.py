#!/usr/bin/env python
"""Round 6 (late): the host's side of the per-frame chain of the clip forward, function by function on one time axis --
what sits between the return of a frame's assignment wait and the launch of the next decoder graph (that stretch is
the critical path: the GPU has nothing queued during it; everything the host does while the decoder graph runs is free).

    python tools/chain_probe.py [> gpurun_out/chain_probe.txt]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.configs import dancetrack_config  # noqa: E402
from memotr_amd.engine import build_optimizer, clip_forward_backward, clip_to_device, make_synthetic_clip, optimizer_step  # noqa: E402
from memotr_amd.models import build_model  # noqa: E402
from memotr_amd.models.criterion import ClipCriterion, build as build_criterion  # noqa: E402
from memotr_amd.models.memotr import MeMOTR  # noqa: E402
from memotr_amd.models.query_updater import QueryUpdater  # noqa: E402

cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0")
dev = torch.device("cuda", 0)
model = build_model(cfg).train()
criterion = build_criterion(cfg)
opt = build_optimizer(cfg, model)
batch = clip_to_device(make_synthetic_clip(5, 800, 1333, 10, seed=42), dev)
for _ in range(3):
    clip_forward_backward(model, criterion, batch, dev)
    optimizer_step(model, opt, 0.1)
torch.cuda.synchronize()

log, depth = [], [0]


def wrap(owner, name, label=None):
    orig = getattr(owner, name)

    def timed(*a, **k):
        d = depth[0]
        depth[0] += 1
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            depth[0] -= 1
            log.append((t0, time.perf_counter(), d, label or name))
    setattr(owner, name, staticmethod(timed) if isinstance(owner.__dict__.get(name), staticmethod) else timed)
    return orig


wrap(ClipCriterion, "begin_frame")
wrap(ClipCriterion, "finish_tracks")
wrap(ClipCriterion, "finish_losses")
wrap(MeMOTR, "decode_frame")
wrap(MeMOTR, "encode_frame")
wrap(QueryUpdater, "select_active_tracks")
wrap(QueryUpdater, "update_tracks_embedding")
wrap(torch.cuda.CUDAGraph, "replay", "graph replay")
wrap(torch.cuda.Event, "synchronize", "event wait")
from memotr_amd.models import matcher as _m  # noqa: E402
wrap(_m.HungarianMatcher, "solve", "scipy")
if "--fine" in sys.argv:        # one level further down
    from memotr_amd.functions import clip_ops as _co  # noqa: E402
    from memotr_amd.models import criterion as _cr  # noqa: E402
    from memotr_amd.models.decoder_graphs import DecoderGraphs  # noqa: E402
    from memotr_amd.models.deformable_decoder import DeformableDecoder  # noqa: E402
    from memotr_amd.models.deformable_transformer import DeformableTransformer  # noqa: E402
    from memotr_amd.models.updater_graphs import UpdaterGraphs  # noqa: E402
    from memotr_amd.structures.track_instances import TrackInstances  # noqa: E402
    for name in ("get_reference_points", "get_query_embed", "get_query_mask", "_class_head_stack", "set_aux_loss"):
        wrap(MeMOTR, name)
    wrap(DeformableTransformer, "decode")
    wrap(DeformableDecoder, "_forward_graphed")
    wrap(DecoderGraphs, "run", "DecoderGraphs.run")
    wrap(DecoderGraphs, "_flat_parameters")
    wrap(UpdaterGraphs, "run", "UpdaterGraphs.run")
    wrap(ClipCriterion, "update_tracked_instances")
    wrap(_cr, "upload")
    for name in ("to", "__getitem__", "cat_packed"):
        wrap(TrackInstances, name, "TrackInstances." + name)
    for name in ("match_cost", "track_ownership", "pair_iou", "focal_loss_per_layer", "focal_labels", "pair_box_loss"):
        wrap(_co, name, "clip_ops." + name)

for _ in range(2):
    log.clear()
    t_start = time.perf_counter()
    loss, _ = clip_forward_backward(model, criterion, batch, dev, backward=False)
    t_fwd = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize()
    optimizer_step(model, opt, 0.1)
print(f"# clip forward on the host: {(t_fwd - t_start) * 1e3:.2f} ms; entries: start ms | duration us | function")
solve_t, solve_n, first = 0.0, 0, None
for t0, t1, d, name in sorted(log):
    if name == "scipy":                 # six per frame: one line
        solve_t += t1 - t0
        solve_n += 1
        first = t0 if first is None else first
        if solve_n == 6:
            print(f"{(first - t_start) * 1e3:8.2f} {solve_t * 1e6:8.0f}   {'  ' * d}scipy x 6")
            solve_t, solve_n, first = 0.0, 0, None
        continue
    print(f"{(t0 - t_start) * 1e3:8.2f} {(t1 - t0) * 1e6:8.0f}   {'  ' * d}{name}")
