"""The clip train step of the reference's train_engine.py:183-246, as a reusable function.

``clip_forward_backward`` is the body of ``train_one_epoch`` for one batch: empty tracks -> T sequential
``model(frame, tracks)`` calls with the criterion and the query updater between frames -> one backward.
``get_param_groups`` reproduces the four AdamW groups of train_engine.py:291-336.  Synthetic clips
(BASELINE.json: random frames, persistent ground-truth identities with a small per-frame drift) stand in
for the data pipeline, which is out of scope.
"""
from __future__ import annotations

import contextlib

from typing import Dict, List, Tuple

import os

import torch
import torch.nn as nn

from .models.utils import get_model
from .structures.track_instances import TrackInstances
from .utils.nested_tensor import tensor_list_to_nested_tensor


def get_param_groups(config: dict, model: nn.Module) -> Tuple[List[Dict], List[str]]:
    def has(name, keys):
        return any(k in name for k in keys)

    backbone, points, updater = ["backbone.backbone"], ["reference_points", "sampling_offsets"], ["query_updater"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [
        {"params": [p for n, p in named if has(n, backbone)], "lr": config["LR_BACKBONE"]},
        {"params": [p for n, p in named if has(n, points)], "lr": config["LR_POINTS"]},
        {"params": [p for n, p in named if has(n, updater)], "lr": config["LR"]},
        {"params": [p for n, p in named if not has(n, backbone) and not has(n, points) and not has(n, updater)],
         "lr": config["LR"]},
    ]
    return groups, ["lr_backbone", "lr_points", "lr_query_updater", "lr"]


def build_optimizer(config: dict, model: nn.Module) -> torch.optim.Optimizer:
    groups, _ = get_param_groups(config, model)
    kwargs = dict(lr=config["LR"], weight_decay=config["WEIGHT_DECAY"])
    on_gpu = any(p.is_cuda for g in groups for p in g["params"])
    if on_gpu:      # same update rule as the reference's AdamW, one multi-tensor kernel instead of a foreach chain
        try:
            return torch.optim.AdamW(params=groups, fused=True, **kwargs)
        except (RuntimeError, TypeError):
            pass
    return torch.optim.AdamW(params=groups, **kwargs)


def make_synthetic_clip(clip_len: int, height: int, width: int, n_gts: int, seed: int, batch_size: int = 1,
                        num_classes: int = 1) -> dict:
    """{"imgs": [B][T] (3,H,W) tensors, "infos": [B][T] {"ids","labels","boxes"}} -- the collate format of the
    reference (data/utils.py:7-12); boxes are normalised cxcywh, identities persist over the clip."""
    g = torch.Generator().manual_seed(seed)
    imgs, infos = [], []
    for _ in range(batch_size):
        centre = torch.rand(n_gts, 2, generator=g) * 0.6 + 0.2
        size = torch.rand(n_gts, 2, generator=g) * 0.15 + 0.05
        labels = torch.randint(0, num_classes, (n_gts,), generator=g)
        clip_imgs, clip_infos = [], []
        for _t in range(clip_len):
            clip_imgs.append(torch.randn(3, height, width, generator=g))
            centre = (centre + torch.randn(n_gts, 2, generator=g) * 0.01).clamp(0.1, 0.9)
            clip_infos.append({"ids": torch.arange(n_gts), "labels": labels.clone(),
                               "boxes": torch.cat((centre, size), -1)})
        imgs.append(clip_imgs)
        infos.append(clip_infos)
    return {"imgs": imgs, "infos": infos}


def clip_to_device(batch: dict, device) -> dict:
    """Pre-stage a clip in HBM (the bench measures with inputs resident)."""
    return {"imgs": [[f.to(device) for f in clip] for clip in batch["imgs"]],
            "infos": [[{k: v.to(device) for k, v in info.items()} for info in clip] for clip in batch["infos"]]}


def encode_chunks(core, clip_len: int):
    """How the frames of a clip are grouped for the backbone + encoder.  Returns (groups, lazy): ``groups`` is
    ``None`` (frame by frame in the reference's order) or a list of group sizes summing to clip_len; ``lazy``
    = each group is encoded right before its first frame is decoded (else: as early as possible).

    ``core.encode_chunks`` or the environment variable MEMOTR_ENCODE_CHUNKS override the default "all":
    "0" (off), "all", "auto" (two just-in-time groups, ~60 % of the clip first), "5", "1,4", "lazy:3,2", ... .
    Measured on MI355X (DanceTrack clip of 5, tools/ab_step.py, same process).  Round 1, host-bound step: reference
    order 235 ms, "all" 232, "lazy:3,2" 219 -- two groups kept GPU-bound encoder backward work queued while the host
    was busy with the launch-bound decoder backward of the earlier frames.  End of round 2, GPU-bound step (decoder
    graphs, fused small-tensor kernels): "auto" 157.6, "2,3" 155.3, "4,1" 152.7, "all" **147.3 ms** -- the largest
    kernels and the fewest launches win once the host is out of the way.  Gradient checkpointing (round 3) groups the
    same way: the checkpointed backbone / encoder segments then recompute per group instead of per frame
    (MEMOTR_CHECKPOINT_REFERENCE_ORDER=1 restores the reference's frame order)."""
    if getattr(core, "use_checkpoint", False) and os.environ.get("MEMOTR_CHECKPOINT_REFERENCE_ORDER", "0") == "1":
        return None, False
    spec = _chunk_spec(core)
    lazy = False
    if isinstance(spec, str):
        if spec.startswith("enc:"):                  # (see backbone_batched)
            spec = spec[4:]
        if spec.startswith("lazy:"):
            lazy, spec = True, spec[5:]
        if spec.strip() == "auto":                   # two groups, ~60 % of the clip first, each encoded just in time
            first = -(-3 * clip_len // 5)
            lazy, spec = True, [first, clip_len - first]
        else:
            spec = [clip_len] if spec.strip() == "all" else [int(x) for x in spec.split(",") if x.strip()]
    if isinstance(spec, int):
        spec = [spec]
    spec = [int(x) for x in spec if int(x) > 0]
    if not spec:
        return None, False
    out, left = [], clip_len
    for n in spec:
        if left <= 0:
            break
        out.append(min(n, left))
        left -= out[-1]
    while left > 0:                                  # the last group size repeats
        out.append(min(spec[-1], left))
        left -= out[-1]
    return out, lazy


def _chunk_spec(core):
    spec = getattr(core, "encode_chunks", None)
    return os.environ.get("MEMOTR_ENCODE_CHUNKS", DEFAULT_ENCODE_CHUNKS) if spec is None else spec


def backbone_batched(core) -> bool:
    """"enc:<groups>": the groups apply to the transformer encoder only -- backbone + feature projections of the whole clip
    run as ONE batch first (``model(stage="features")``; the convolutions lose the most at small batches), and each
    group's encoder is queued one group ahead of its first decode, so the GPU has a group's encoder to run while the
    host is in a frame's launch-bound chain (assignment, track bookkeeping, query updater, the next decoder's launch)."""
    spec = _chunk_spec(core)
    return isinstance(spec, str) and spec.startswith("enc:")


DEFAULT_ENCODE_CHUNKS = "all"
# Frame t's losses are issued under frame t + 1's decoder: the next frame needs a frame's TRACKS (criterion.finish_tracks),
# not its losses, and the host would otherwise sit in the wait for that decoder (0.45-0.5 ms per frame, tools/
# replay_cost_probe.py: clip forward on the host 46.1 -> 45.2 ms).  MEMOTR_DEFER_LOSSES=0: both halves back to back.
DEFER_LOSSES = os.environ.get("MEMOTR_DEFER_LOSSES", "1") != "0"

_ENCODE_STREAMS = {}


def encode_stream(device):
    """Opt-in (MEMOTR_ENCODE_STREAM=1): the HIP stream the later encode groups of a clip run on.

    Idea: the decoder / criterion / query-updater chain of a frame is a serial string of hundreds of small kernels
    that leave most of the 256 CUs idle, forward and backward; the backbone + encoder of the NEXT group of frames do
    not depend on it, so queued on a second stream their large GEMMs and convolutions could fill the idle CUs (and
    autograd replays every backward node on the stream of its forward, so the same holds for the backward).
    Measured on MI355X it does not pay: 192.4 vs 192.7 ms per step (tools/ab_step.py).  A chain of small kernels
    that takes 2.6 ms alone takes 16 ms next to 15 ms of GEMMs on another stream, at either stream priority
    (tools/stream_overlap_probe.py, profiles/r02_stream_overlap_probe.txt): the chain is slowed to the length of
    the large kernels it shares the CUs with, and the chain is the critical path.  Kept for clips whose groups are
    not chain-bound; results are identical either way (tests/test_model_gpu.py)."""
    dev = torch.device(device)
    if dev.type != "cuda" or os.environ.get("MEMOTR_ENCODE_STREAM", "0") != "1":
        return None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _ENCODE_STREAMS:
        _ENCODE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return _ENCODE_STREAMS[idx]


def clip_forward_backward(model: nn.Module, criterion, batch: dict, device, use_dab: bool = True,
                          accumulation_steps: int = 1, backward: bool = True, no_grad_frames: int = None):
    """One clip through model + criterion (+ backward).  Returns (loss tensor, loss_dict).
    ``no_grad_frames`` (config NO_GRAD_FRAMES, reference train_engine.py:202-230): the first that many frames run
    without a graph, and all but the last of them without query augmentation; they take the reference's frame
    order (no batched encode: nothing of theirs is differentiated)."""
    core = get_model(model)
    clip_len = len(batch["imgs"][0])
    n_clips = len(batch["imgs"])

    def frames(lo, hi):
        """Frames lo..hi-1 of every clip as ONE padded batch, frame-major: rows [k*B, (k+1)*B) are frame lo+k."""
        return tensor_list_to_nested_tensor([clip[t] for t in range(lo, hi) for clip in batch["imgs"]]).to(device)

    # The backbone + encoder do not depend on the tracks, so the frames of a clip need not take that half one at
    # a time (train_engine.py:196-221 does): ``chunks`` groups consecutive frames into one batched encode call.
    # Same operations per frame (FrozenBN / GroupNorm / LayerNorm are per-sample), 1/len(chunk) of the kernel
    # launches, larger GEMMs and convolutions; the decoder still runs frame by frame on the carried tracks.
    chunks, lazy = encode_chunks(core, clip_len)
    if no_grad_frames:
        chunks, lazy = None, False
    starts = [sum(chunks[:i]) for i in range(len(chunks))] if chunks is not None else []
    encoded = {}                                   # frame index -> encode result of that frame
    clip_key = object()                            # identifies this clip's autograd graph to per-clip caches

    split = chunks is not None and backbone_batched(core) and not getattr(core, "use_checkpoint", False)
    feats = None                                   # split: the clip's backbone features, encoders per group
    side = encode_stream(device) if chunks is not None and len(chunks) > 1 and not split else None
    if side is not None:
        lazy = False                               # later groups are queued one group ahead, on the side stream
    on_side = set()                                # frames whose encode result was produced on the side stream

    def encode_chunk(ci):
        nonlocal feats
        lo, n = starts[ci], chunks[ci]
        if side is not None and ci > 0:
            main = torch.cuda.current_stream()
            batch_frames = frames(lo, lo + n)      # assembled on the main stream, read on the side stream
            batch_frames.encode_slot = ci
            side.wait_stream(main)
            with torch.cuda.stream(side):
                enc = model(frame=batch_frames, stage="encode")
            for t in (batch_frames.tensors, batch_frames.masks):
                t.record_stream(side)
            for v in enc.values():                 # produced on the side stream, consumed by the decoder on main
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)
            on_side.update(range(lo, lo + n))
        elif split:
            if feats is None:
                feats = model(frame=frames(0, clip_len), stage="features")
            enc = model(frame=(feats, lo * n_clips, (lo + n) * n_clips), stage="encode_features")
            if lo + n >= clip_len:
                feats = None
        else:
            batch_frames = frames(lo, lo + n)
            batch_frames.encode_slot = ci           # which of the clip's encode calls this is (models/encode_graphs.py)
            enc = model(frame=batch_frames, stage="encode")
        if n == 1:
            encoded[lo] = dict(enc, frame_slot=lo, clip_key=clip_key)   # the frame's slot for the decoder's hipGraphs
            return
        per_frame = {k: (v.split(n_clips, dim=0) if k in ("memory", "valid_ratios", "mask_flatten") else None)
                     for k, v in enc.items()}
        for j in range(n):
            encoded[lo + j] = {k: (per_frame[k][j] if per_frame[k] is not None else v) for k, v in enc.items()}
            encoded[lo + j]["frame_slot"] = lo + j
            encoded[lo + j]["clip_key"] = clip_key

    # The query updater's selection of active tracks (training, no drop / insert augmentation) from flags that travel
    # to the host with the matching costs instead of a boolean mask on the device (models/criterion.py: keep_rows);
    # MEMOTR_KEEP_ROWS=0: the mask.
    upd = getattr(core, "query_updater", None)
    hint = (chunks is not None and upd is not None and core.training and getattr(upd, "tp_drop_ratio", 1.0) == 0.0
            and getattr(upd, "fp_insert_ratio", 1.0) == 0.0 and os.environ.get("MEMOTR_KEEP_ROWS", "1") != "0")
    criterion.keep_threshold = upd.update_threshold if hint else None
    if upd is not None:
        upd.__dict__["_keep_rows_ok"] = hint

    def set_up():
        tr = TrackInstances.init_tracks(batch=batch, hidden_dim=core.hidden_dim, num_classes=core.num_classes,
                                        device=device, use_dab=use_dab)
        criterion.init_a_clip(batch=batch, hidden_dim=core.hidden_dim, num_classes=core.num_classes, device=device)
        return tr

    # ground truth already on the device: the set-up is host-only work, so queue the first encode before it;
    # otherwise its (pageable) uploads would wait behind the encode -- do them first
    resident = all(v.device.type == torch.device(device).type for v in batch["infos"][0][0].values()
                   if torch.is_tensor(v))
    tracks = None if resident else set_up()
    if chunks is not None:
        encode_chunk(0)
    if tracks is None:
        tracks = set_up()
    owed = None                                     # criterion state of the frame whose losses are still to be issued
    for frame_idx in range(clip_len):
        if lazy and frame_idx in starts and frame_idx > 0:   # just in time: encoded right before its first decode
            encode_chunk(starts.index(frame_idx))
        if chunks is None:                          # the reference's order: everything of a frame, then the next
            frozen = bool(no_grad_frames) and frame_idx < no_grad_frames
            with (torch.no_grad() if frozen else contextlib.nullcontext()):
                res = model(frame=frames(frame_idx, frame_idx + 1), tracks=tracks)
                previous, new, unmatched = criterion.process_single_frame(model_outputs=res,
                                                                          tracked_instances=tracks,
                                                                          frame_idx=frame_idx)
                if frozen and frame_idx < clip_len - 1:
                    tracks = core.postprocess_single_frame(previous, new, unmatched,
                                                           no_augment=frame_idx < no_grad_frames - 1)
            if frozen:
                continue
        else:
            if frame_idx in on_side:               # first use of a side-stream result: order the streams
                torch.cuda.current_stream().wait_stream(side)
                on_side.difference_update(range(frame_idx, clip_len))
            res = model(tracks=tracks, encoded=encoded.pop(frame_idx))
            pending = criterion.begin_frame(model_outputs=res, tracked_instances=tracks, frame_idx=frame_idx)
            if not lazy and frame_idx in starts and starts.index(frame_idx) + 1 < len(chunks):
                encode_chunk(starts.index(frame_idx) + 1)     # queued before the host blocks on this frame's costs
            if DEFER_LOSSES:
                if owed is not None:
                    criterion.finish_losses(owed)         # the previous frame's, while the GPU runs this frame's decoder
                previous, new, unmatched = criterion.finish_tracks(pending)
                owed = pending
            else:
                previous, new, unmatched = criterion.finish_frame(pending)
        if frame_idx < clip_len - 1:
            # (batched-encode path: the frame's slot for the updater's hipGraphs, models/updater_graphs.py)
            tracks = core.postprocess_single_frame(previous, new, unmatched,
                                                   **({} if chunks is None else {"frame_slot": frame_idx,
                                                                                 "clip_key": clip_key}))
    if owed is not None:
        criterion.finish_losses(owed)
    # (no log values here: each is a device->host read, i.e. a stream synchronisation in front of the backward)
    loss_dict, _ = criterion.get_mean_by_n_gts(with_log=os.environ.get("MEMOTR_LOSS_LOG_SYNC", "0") == "1")
    loss = criterion.get_sum_loss_dict(loss_dict=loss_dict)
    if backward:
        # the backward pass never runs under autocast (the engine's threads inherit the caller's autocast state):
        # backward ops take the dtypes their forward ran in; with autocast left on, the float32 islands' backward GEMMs
        # -- (6, 256, 310) x (6, 310, 1) class heads, a (3, 2) x (2, 6) box product -- were cast to bf16 and cost
        # ~11 ms of HOST time each in hipBLASLt (10 per step: 110 of the bf16 step's 220 ms, tools/idle_gaps.py --bf16)
        with (torch.autocast(device_type=torch.device(device).type, enabled=False)
              if torch.is_autocast_enabled() else contextlib.nullcontext()):
            (loss / accumulation_steps).backward()
    return loss, loss_dict


def optimizer_step(model: nn.Module, optimizer: torch.optim.Optimizer, max_norm: float):
    """clip_grad_norm_ with the reference's hard-coded 0.1 whenever clipping is on (train_engine.py:241-246)."""
    if max_norm > 0:
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
    optimizer.step()
    optimizer.zero_grad()
