"""Clip-level training criterion (API and arithmetic of the reference's models/criterion.py:26-494).

Per frame: refresh the tracked instances from the model output, Hungarian-match the detect queries to the
ground truths that no track owns, build the new / unmatched TrackInstances for the query updater, and add
focal + L1 + GIoU losses for the last decoder layer and the auxiliary layers.  Same public methods as the
reference (``init_a_clip``, ``process_single_frame``, ``get_mean_by_n_gts``, ``get_sum_loss_dict``).

Host stalls removed relative to the reference: id <-> ground-truth bookkeeping is tensor arithmetic instead
of python dict loops over ``.item()`` (criterion.py:166-194); the six assignment problems of a frame (main +
aux layers) are solved from one device->host copy; scalar logs stay on the device until the clip ends; the two
normalisation all-reduces (criterion.py:122-124) travel as one message.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.distributed
import torch.nn.functional as F

from ..functions import clip_ops
from ..structures.track_instances import TrackInstances
from ..utils.box_ops import box_cxcywh_to_xyxy, box_iou_union, generalized_box_iou
from ..utils.utils import distributed_world_size, is_distributed
from .matcher import HungarianMatcher
from .matcher import build as build_matcher
from .utils import logits_to_scores

_LOSS_KEYS = ("box_l1_loss", "box_giou_loss", "label_focal_loss")



def batch_item(x: torch.Tensor, b: int) -> torch.Tensor:
    """x[b]; a view without a Select node when the batch is a single clip (its backward would be a zero-fill + a copy
    per use)."""
    return x.squeeze(0) if x.shape[0] == 1 else x[b]


def rows_of(x: torch.Tensor, b: int, idx: torch.Tensor) -> torch.Tensor:
    """x[b][idx] for a 1-d index: index_select, whose backward is one index_add (advanced indexing differentiates
    through a sort-based accumulate, ~7 kernels for a handful of rows)."""
    return batch_item(x, b).index_select(0, idx)


def upload(values, dtype, device):
    """Host list -> device tensor without stalling the host: a pageable-memory copy blocks until everything
    already queued on the stream has run, a pinned-memory one is queued like a kernel."""
    t = torch.as_tensor(values, dtype=dtype)
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


def _fp32_island(fn):
    """Losses and matching costs are float32 work on query-sized tensors: under autocast (the bf16 extension) they run
    with autocast off -- the model's decode half hands over float32 outputs -- i.e. what autocast's own promotion
    rules would compute for these loss formulas, minus the casts."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        if torch.is_autocast_enabled():
            with torch.autocast(device_type="cuda", enabled=False):
                return fn(self, *args, **kwargs)
        return fn(self, *args, **kwargs)
    return wrapped


_MAX_CLIP_FRAMES = 63      # frames per clip the distributed count all-reduce has room for (64 floats)


class ClipCriterion:
    def __init__(self, num_classes, matcher: HungarianMatcher, n_det_queries, aux_loss: bool, weight: dict,
                 max_frame_length: int, n_aux: int, merge_det_track_layer: int = 0, aux_weights: List = None,
                 hidden_dim: int = 256, use_dab: bool = True):
        self.device = None
        self.aux_loss = aux_loss
        self.weight = weight
        self.num_classes = num_classes
        self.matcher = matcher
        self.n_det_queries = n_det_queries
        # (checked HERE, where every rank fails alike: a rank that raised on its own long clip in front of the count
        #  all-reduce would leave the others waiting in it -- advisor, round 5)
        if max_frame_length > _MAX_CLIP_FRAMES:
            raise ValueError(f"max_frame_length {max_frame_length}: the distributed count all-reduce carries at most "
                             f"{_MAX_CLIP_FRAMES} frames per clip")
        self.max_frame_length = max_frame_length
        self.n_aux = n_aux
        self.use_dab = use_dab
        self.frame_weights = [1.0] * max_frame_length
        self.aux_weights = aux_weights
        self.hidden_dim = hidden_dim
        self.merge_det_track_layer = merge_det_track_layer
        self.gt_trackinstances_list = None      # [clip_len][B]
        self._acc: torch.Tensor = None          # (3 losses, main | aux) running sums of the clip
        self.log: Dict[str, torch.Tensor] = {}
        self.n_gts: List[int] = []

    def set_device(self, device: torch.device):
        self.device = device

    # ------------------------------------------------------------------ clip set-up / reduction
    def init_a_clip(self, batch: Dict, hidden_dim: int, num_classes: int, device: torch.device):
        self.device = device
        clip_len, bs = len(batch["imgs"][0]), len(batch["imgs"])
        self.gt_trackinstances_list = []
        for c in range(clip_len):
            gts = TrackInstances.init_tracks(batch, hidden_dim=hidden_dim, num_classes=num_classes, device=device)
            for b in range(bs):
                info = batch["infos"][b][c]
                gts[b].ids, gts[b].labels, gts[b].boxes = info["ids"], info["labels"], info["boxes"]
                gts[b] = gts[b].to(device)
            self.gt_trackinstances_list.append(gts)
        if clip_len > len(self.frame_weights):     # clips longer than SAMPLE_LENGTHS (synthetic stress runs)
            self.frame_weights = self.frame_weights + [1.0] * (clip_len - len(self.frame_weights))
        self.n_gts = []
        self.log = {}
        self._acc = torch.zeros((len(_LOSS_KEYS), 2 if self.aux_loss else 1), device=device)

    @property
    def loss(self) -> Dict[str, torch.Tensor]:
        """The clip's running loss sums by name (the reference's ``self.loss`` dict, criterion.py:86-102).  They are
        kept as ONE (3, 2) tensor -- a frame adds ``per-layer losses @ weights`` to it, one small product instead
        of six select / scale / add chains (and their zero-fill + copy backward nodes) per frame."""
        cols = self._acc.t().reshape(-1).unbind(0)            # main losses first, then the aux ones
        keys = _LOSS_KEYS + tuple("aux_" + k for k in _LOSS_KEYS) if self._acc.shape[1] == 2 else _LOSS_KEYS
        return dict(zip(keys, cols))

    def get_sum_loss_dict(self, loss_dict: dict):
        def w(name):
            for k in _LOSS_KEYS:
                if k in name:
                    return self.weight[k]
        return sum(w(k) * v for k, v in loss_dict.items())

    def get_mean_by_n_gts(self, with_log: bool = True) -> Tuple[Dict, Dict]:
        """Losses divided by the clip's (world-averaged) ground-truth count (reference models/criterion.py:196-216) and,
        with ``with_log``, the per-frame log values as python floats.  The counts are host numbers: a single process
        divides by them directly; under torch.distributed ONE all-reduce of 1 + T floats runs and the total stays on
        the device.  ``with_log=False`` (the training loop) therefore reads nothing back: every float() here is a
        stream synchronisation between the forward and the backward of the step."""
        host = [float(sum(self.n_gts))] + [float(n) for n in self.n_gts]
        per_frame = None
        if is_distributed():
            # a fixed-length vector: ranks whose clips differ in length (a sampler that mixes sample lengths) still issue
            # the SAME collective -- the reference's per-frame all-reduces (criterion.py:208-214) would deadlock there
            # (the constructor bounds max_frame_length; a longer clip is a data error on THIS rank only -- its extra frames
            #  stay out of the vector, so the collective still matches the other ranks', and the error is raised after it)
            too_long = len(host) > _MAX_CLIP_FRAMES + 1
            sent = host[:_MAX_CLIP_FRAMES + 1]
            padded = sent + [0.0] * (_MAX_CLIP_FRAMES + 1 - len(sent))
            # second half: 1 for every frame this rank's clip has -- a frame's mean count is over the ranks that HAVE it
            padded += [1.0] * len(sent) + [0.0] * (_MAX_CLIP_FRAMES + 1 - len(sent))
            counts = torch.as_tensor(padded, dtype=torch.float, device=self.device)
            torch.distributed.all_reduce(counts)
            if too_long:
                raise ValueError(f"clip of {len(host) - 1} frames: the count all-reduce carries at most {_MAX_CLIP_FRAMES}")
            half = _MAX_CLIP_FRAMES + 1
            total = torch.clamp(counts[0] / distributed_world_size(), min=1)
            if with_log:
                have = torch.clamp(counts[half + 1:half + len(host)], min=1)
                per_frame = torch.clamp(counts[1:len(host)] / have, min=1).tolist()
        else:
            total = max(host[0], 1.0)
            per_frame = [max(c, 1.0) for c in host[1:]]
        loss = {k: v / total for k, v in self.loss.items()}
        log = {}
        if with_log:
            for k, v in self.log.items():
                for i, n in enumerate(per_frame):
                    if f"frame{i}" in k:
                        log[k] = (float(v) / n, 1)
                        break
        return loss, log

    # ------------------------------------------------------------------ one frame
    def process_single_frame(self, model_outputs: dict, tracked_instances: List[TrackInstances], frame_idx: int):
        """Returns (tracked, new, unmatched) TrackInstances lists and accumulates this frame's losses.

        All decoder layers are handled together: one stacked cost tensor -> ONE device->host copy -> scipy on the
        host -> one small index upload -> stacked focal / L1 / GIoU losses.  Track bookkeeping uses slices and
        ``torch.where`` (boolean-mask indexing would synchronise once per field).

        ``begin_frame`` / ``finish_frame`` are the two halves either side of the device->host copy; a training
        loop may queue other GPU work (the next frame's backbone + encoder) between them.
        """
        return self.finish_frame(self.begin_frame(model_outputs, tracked_instances, frame_idx))

    @_fp32_island
    def begin_frame(self, model_outputs: dict, tracked_instances: List[TrackInstances], frame_idx: int) -> dict:
        """Device side of the matching: ownership of ground truths + stacked cost tensors, and the (asynchronous,
        pinned-memory) copy of both to the host.  Returns the state ``finish_frame`` consumes."""
        nd = self.n_det_queries
        gts = self.gt_trackinstances_list[frame_idx]
        B = len(tracked_instances)
        dev = self.device
        tracked_instances = self.update_tracked_instances(model_outputs, tracked_instances)
        layers = [model_outputs] + (list(model_outputs["aux_outputs"]) if self.aux_loss else [])
        n_layers = len(layers)
        early = [li > 0 and (li - 1) < self.merge_det_track_layer for li in range(n_layers)]
        if self.aux_loss and "pred_logits_all" in model_outputs:          # stacks in decoder order: main layer last
            logits_all = torch.roll(model_outputs["pred_logits_all"], 1, 0)
            boxes_all = torch.roll(model_outputs["pred_bboxes_all"], 1, 0)
        else:
            logits_all = torch.stack([o["pred_logits"] for o in layers])  # (n_layers, B, Nq, K)
            boxes_all = torch.stack([o["pred_bboxes"] for o in layers])   # (n_layers, B, Nq, 4)

        # ---- device side: ownership of ground truths + stacked cost tensors, then one transfer ----
        payload, n_gt_list, n_flag_list = [], [], []
        thr = getattr(self, "keep_threshold", None)
        for b in range(B):
            tr, gt = tracked_instances[b], gts[b]
            n_gt, n_tr = len(gt), len(tr)
            n_gt_list.append(n_gt)
            lg, bx = logits_all.detach()[:, b, :nd], boxes_all.detach()[:, b, :nd]
            # what the query updater's selection of active tracks will ask of the device -- score above its threshold?
            # identity still alive? -- answered here for every query / track / ground truth and sent along with the
            # costs: the host then knows which rows stay active and uploads their indices with the matching's, where
            # the boolean mask's nonzero() was a second stream synchronisation per frame (finish_tracks: keep_rows)
            n_fl = (nd + 2 * n_tr + n_gt) if thr is not None else 0
            n_flag_list.append(n_fl)
            flags = None
            if n_fl:
                score_ok = torch.max(logits_to_scores(batch_item(model_outputs["pred_logits"], b).detach()[:nd + n_tr]),
                                     dim=1).values > thr
                flags = (score_ok, tr.ids >= 0, gt.ids >= 0)
            kernels = (clip_ops.fused(lg, bx, gt.boxes) and n_gt > 0 and tr.ids.dtype == gt.ids.dtype == torch.int64
                       and os.environ.get("MEMOTR_FUSED_BOOKKEEPING", "1") != "0")
            if kernels:
                # ownership and the whole cost tensor written straight into the buffer that travels to the host:
                # two launches (include/clip_ops_hip.h: clipops_track_ownership_i64, clipops_match_cost_f32)
                buf = torch.empty((n_fl + n_gt + n_layers * nd * n_gt,), dtype=torch.float32, device=lg.device)
                if n_fl:
                    buf[:nd + n_tr].copy_(flags[0])
                    if n_tr:
                        buf[nd + n_tr:nd + 2 * n_tr].copy_(flags[1])
                    buf[nd + 2 * n_tr:n_fl].copy_(flags[2])
                tr.matched_idx, _ = clip_ops.track_ownership(tr.ids, gt.ids, free_out=buf[n_fl:n_fl + n_gt])
                clip_ops.match_cost(lg, bx, gt.labels, gt.boxes, self.matcher.cost_class, self.matcher.cost_bbox,
                                    self.matcher.cost_giou, out=buf[n_fl + n_gt:])
                payload.append(buf)
                continue
            if n_fl:
                payload += [f.to(torch.float32) for f in flags]
            if n_tr > 0 and n_gt > 0:
                # index of the LAST ground truth carrying the id (the reference's ``gt_ids_to_idx`` dict keeps the
                # last one when a frame repeats an id, criterion.py:166-170), -1 when the identity is gone
                tr.matched_idx, free = clip_ops.track_ownership_reference(tr.ids, gt.ids)
            else:
                tr.matched_idx = torch.full((n_tr,), -1, dtype=torch.long, device=dev)
                free = torch.ones((n_gt,), dtype=torch.float32, device=dev)
            if clip_ops.fused(lg, bx, gt.boxes):        # one kernel for the whole cost tensor
                cost = clip_ops.match_cost(lg, bx, gt.labels, gt.boxes, self.matcher.cost_class,
                                           self.matcher.cost_bbox, self.matcher.cost_giou)
            else:
                cost = self.matcher.cost_matrix_stacked(lg, bx, gt.labels, gt.boxes)      # (n_layers, nd, n_gt)
            payload += [free.to(cost.dtype).reshape(-1), cost.reshape(-1)]
        ready = keep = None
        if not payload:
            host = torch.zeros(0)
        else:
            flat = payload[0] if len(payload) == 1 else torch.cat(payload)
            if flat.is_cuda:
                # copy on a side stream: the event then depends on the work queued up to here only, and the host
                # wait in finish_frame is not held back by kernels the caller queues on the main stream meanwhile
                side = self._copy_stream(flat.device)
                computed = torch.cuda.Event()
                computed.record()
                side.wait_event(computed)
                host = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
                with torch.cuda.stream(side):
                    host.copy_(flat, non_blocking=True)
                    ready = torch.cuda.Event()
                    ready.record(side)
                keep = flat                             # stays referenced until the copy has completed
            else:
                host = flat
        return {"model_outputs": model_outputs, "tracked_instances": tracked_instances, "frame_idx": frame_idx,
                "layers": layers, "early": early, "logits_all": logits_all, "boxes_all": boxes_all,
                "n_gt_list": n_gt_list, "n_flag_list": n_flag_list, "host": host, "ready": ready, "keep": keep}

    def _constant(self, key, values, dtype, device):
        """Small per-configuration device constants (uploaded once, not once per frame)."""
        cache = self.__dict__.setdefault("_constants", {})
        key = (key, dtype, str(device))
        if key not in cache:
            cache[key] = torch.as_tensor(values, dtype=dtype, device=device)
        return cache[key]

    def _copy_stream(self, device):
        streams = self.__dict__.setdefault("_copy_streams", {})
        if device not in streams:
            streams[device] = torch.cuda.Stream(device=device)
        return streams[device]

    def finish_frame(self, state: dict):
        """Host side (assignment problems) and the device work that depends on it; see process_single_frame.
        = ``finish_tracks`` (what the next frame needs) + ``finish_losses`` (what only the backward needs).  The training
        loop issues the losses of frame t while the GPU runs the decoder of frame t + 1 (engine.DEFER_LOSSES): the wait
        for that decoder -- 0.45-0.5 ms per frame -- is then spent launching instead (clip forward on the host 46.1 ->
        45.2 ms, tools/replay_cost_probe.py; the whole-step A/B could not resolve it: 124.7 vs 125.1 ms)."""
        out = self.finish_tracks(state)
        self.finish_losses(state)
        return out

    @_fp32_island
    def finish_tracks(self, state: dict):
        """The assignment problems on the host, the index upload, and the track sets the next frame needs: returns
        (tracked, new, unmatched).  Leaves in ``state`` what ``finish_losses`` reads."""
        model_outputs, tracked_instances, frame_idx = state["model_outputs"], state["tracked_instances"], state["frame_idx"]
        early, logits_all, boxes_all, n_gt_list = state["early"], state["logits_all"], state["boxes_all"], state["n_gt_list"]
        nd, dev, B, n_layers = self.n_det_queries, self.device, len(tracked_instances), len(state["layers"])
        gts = self.gt_trackinstances_list[frame_idx]
        if state["ready"] is not None:
            state["ready"].synchronize()            # waits for the copy only, not for work queued after it
        host = state["host"]

        # ---- host side: the assignment problems ----
        pos = 0
        rows_layer, rows_q, rows_g = [[] for _ in range(B)], [[] for _ in range(B)], [[] for _ in range(B)]
        main_q, flags_h = [], []
        for b in range(B):
            n_gt = n_gt_list[b]
            n_fl = state["n_flag_list"][b]
            flags_h.append(host[pos:pos + n_fl].bool().numpy() if n_fl else None)
            pos += n_fl
            free_h = host[pos:pos + n_gt].bool().numpy()
            pos += n_gt
            cost_h = host[pos:pos + n_layers * nd * n_gt].view(n_layers, nd, n_gt).numpy()
            pos += n_layers * nd * n_gt
            free_idx = free_h.nonzero()[0]
            for li in range(n_layers):
                if early[li]:
                    qi, gj = self.matcher.solve(cost_h[li])
                else:
                    qi, tj = self.matcher.solve(cost_h[li][:, free_idx])
                    gj = free_idx[tj]
                rows_layer[b].append(np.full((len(qi),), li, dtype=np.int64))
                rows_q[b].append(qi)
                rows_g[b].append(gj)
                if li == 0:
                    main_q.append((qi, gj))

        # ---- back on the device: new tracks, unmatched detections, IoU of the carried tracks ----
        new_tracks, unmatched = [], []
        n_det_out = len(model_outputs["det_query_embed"])
        per_clip = []                                   # what finish_losses reads, per clip of the batch
        for b in range(B):
            tr, gt = tracked_instances[b], gts[b]
            n_tr = len(tr)
            n_main = len(main_q[b][0])
            mq, mg = (np.asarray(x, dtype=np.int64) for x in main_q[b])
            # detections nobody claimed (the host knows the matched detect queries of the last layer)
            unclaimed = np.ones((n_det_out,), dtype=bool)
            unclaimed[mq] = False
            free_np = np.flatnonzero(unclaimed)
            # rows of (carried tracks; new tracks; unclaimed detections) the query updater keeps active: the reference's
            # ``(scores > update_thresh) | (ids >= 0)`` (models/query_updater.py:170-176), from the flags of begin_frame
            keep_np = None
            if flags_h[b] is not None:
                fl = flags_h[b]
                sc, idf, gtf = fl[:nd + n_tr], fl[nd + n_tr:nd + 2 * n_tr], fl[nd + 2 * n_tr:]
                keep_np = np.concatenate((np.flatnonzero(sc[nd:nd + n_tr] | idf),
                                          n_tr + np.flatnonzero(sc[mq] | gtf[mg]),
                                          n_tr + n_main + np.flatnonzero(sc[free_np])))
            # ONE upload for everything the host found out
            cat = lambda rows: (np.concatenate([np.asarray(r, dtype=np.int64) for r in rows])          # noqa: E731
                                if rows else np.zeros((0,), dtype=np.int64))
            lay_np, q_np, g_np = cat(rows_layer[b]), cat(rows_q[b]), cat(rows_g[b])
            n_pairs = len(q_np)
            up = upload(np.concatenate((lay_np, q_np, g_np, free_np) + (() if keep_np is None else (keep_np,))),
                        torch.long, dev)
            lay_i, q_i, g_i = up[:n_pairs], up[n_pairs:2 * n_pairs], up[2 * n_pairs:3 * n_pairs]
            free_q = up[3 * n_pairs:3 * n_pairs + len(free_np)]
            keep_rows = up[3 * n_pairs + len(free_np):] if keep_np is not None else None
            q_idx, gt_idx = q_i[:n_main], g_i[:n_main]                              # layer 0 comes first

            nt = TrackInstances(frame_height=tr.frame_height, frame_width=tr.frame_width, hidden_dim=tr.hidden_dim,
                                num_classes=self.num_classes, device=dev)
            nt.ids = gt.ids[gt_idx]
            nt.matched_idx = gt_idx
            queries = rows_of(model_outputs["aux_outputs"][-1]["queries"], b, q_idx)
            nt.query_embed = queries if self.use_dab else torch.cat(
                (model_outputs["det_query_embed"][q_idx][:, :self.hidden_dim], queries), dim=-1)
            nt.ref_pts = rows_of(model_outputs["last_ref_pts"], b, q_idx)
            nt.output_embed = rows_of(model_outputs["outputs"], b, q_idx)
            nt.boxes = rows_of(model_outputs["pred_bboxes"], b, q_idx)
            nt.logits = rows_of(model_outputs["pred_logits"], b, q_idx)
            nt.iou = torch.zeros((n_main,), dtype=torch.float, device=dev)
            if not nt.on_device(dev):
                nt = nt.to(dev)
            per_clip.append({"idx": (lay_i, q_i, g_i), "n_tr": n_tr, "matched_idx": tr.matched_idx if n_tr > 0 else None})

            d = TrackInstances(hidden_dim=model_outputs["outputs"].shape[-1],
                               num_classes=model_outputs["pred_logits"].shape[-1], device=dev)
            d.ref_pts = rows_of(model_outputs["init_ref_pts"], b, free_q)
            d.output_embed = rows_of(model_outputs["outputs"], b, free_q)
            d.logits = rows_of(model_outputs["pred_logits"], b, free_q)
            d.boxes = rows_of(model_outputs["pred_bboxes"], b, free_q)
            queries = rows_of(model_outputs["aux_outputs"][-1]["queries"], b, free_q)
            d.query_embed = queries if self.use_dab else torch.cat(
                (model_outputs["det_query_embed"][free_q][:, :self.hidden_dim], queries), dim=-1)
            d.ids = -torch.ones((len(free_q),), dtype=torch.long, device=dev)
            d.matched_idx = -torch.ones((len(free_q),), dtype=torch.long, device=dev)
            d.iou = torch.zeros((len(free_q),), dtype=torch.float, device=dev)
            if keep_rows is not None:
                # (QueryUpdater.select_active_tracks: index instead of boolean mask -- valid for THIS threshold only)
                d._keep_rows = (keep_rows, float(self.keep_threshold))
            unmatched.append(d)

            # IoU of every track with the ground truth it owns (kept where it owns none)
            if not tr.on_device(dev):
                tr = tr.to(dev)
            tracked_instances[b] = tr
            if len(gt) > 0:
                iou_of = clip_ops.pair_iou if clip_ops.fused(logits_all, boxes_all, gt.boxes) else clip_ops.pair_iou_reference
                if n_main > 0:
                    nt.iou = iou_of(nt.boxes, gt.boxes, nt.matched_idx)
                if n_tr > 0:
                    has = tr.matched_idx >= 0
                    tr.iou = torch.where(has, iou_of(tr.boxes, gt.boxes, tr.matched_idx.clamp(min=0)), tr.iou)
            new_tracks.append(nt)
        self.n_gts.append(sum(n_gt_list))
        state["per_clip"] = per_clip
        return tracked_instances, new_tracks, unmatched

    @_fp32_island
    def finish_losses(self, state: dict):
        """Focal / L1 / GIoU losses of every decoder layer for the frame of ``state`` (after ``finish_tracks``), added
        to the clip's running sums.  Frames must come in order (the sums are accumulated in frame order)."""
        frame_idx, early, logits_all, boxes_all = state["frame_idx"], state["early"], state["logits_all"], state["boxes_all"]
        nd, dev, n_layers = self.n_det_queries, self.device, len(state["layers"])
        gts = self.gt_trackinstances_list[frame_idx]
        loss_label = torch.zeros((n_layers,), device=dev)
        loss_l1 = torch.zeros((n_layers,), device=dev)
        loss_giou = torch.zeros((n_layers,), device=dev)
        for b, clip in enumerate(state.pop("per_clip")):
            gt, n_tr, matched_idx = gts[b], clip["n_tr"], clip["matched_idx"]
            lay_i, q_i, g_i = clip["idx"]
            # classification targets of every layer: matched detect queries + (late layers) the carried tracks
            n_q = nd + n_tr                                                        # real (unpadded) queries of clip b
            late = self._constant(("late", tuple(early)), [not e for e in early], torch.bool, dev)
            has = matched_idx >= 0 if n_tr > 0 else None
            if (clip_ops.fused(logits_all, boxes_all, gt.boxes) and gt.labels.dtype == torch.int64
                    and os.environ.get("MEMOTR_FUSED_BOOKKEEPING", "1") != "0"):
                labels = clip_ops.focal_labels(lay_i, q_i, g_i, gt.labels, matched_idx, late, nd, n_tr, self.num_classes)
            else:
                labels = clip_ops.focal_labels_reference(lay_i, q_i, g_i, gt.labels, matched_idx, late, nd, n_tr,
                                                         self.num_classes)
            use_kernels = clip_ops.fused(logits_all, boxes_all, gt.boxes)
            if use_kernels:
                loss_label = loss_label + clip_ops.focal_loss_per_layer(logits_all[:, b, :n_q], labels)
            else:
                one_hot = F.one_hot(labels, self.num_classes + 1)[..., :-1].to(logits_all.dtype)
                loss_label = loss_label + sigmoid_focal_loss_per_layer(logits_all[:, b, :n_q], one_hot)

            # box losses: detect pairs of every layer + tracked pairs of the late layers
            pair_loss = clip_ops.pair_box_loss if use_kernels else clip_ops.pair_box_loss_reference
            if n_tr > 0 and len(gt) > 0:
                # every (layer, track) pair, weight 1 where the layer carries tracks and the track owns a ground truth
                lay_t = self._constant(("lay_t", n_layers, n_tr), [li for li in range(n_layers) for _ in range(n_tr)],
                                       torch.long, dev)
                q_t = self._constant(("q_t", n_layers, n_tr, nd), [nd + j for _ in range(n_layers) for j in range(n_tr)],
                                     torch.long, dev)
                w_pair = (has[None, :] & late[:, None]).to(boxes_all.dtype).reshape(-1)
                g_t = matched_idx.clamp(min=0).repeat(n_layers)
                l1_t, gi_t = pair_loss(boxes_all, lay_t, q_t, b, gt.boxes, g_t, w_pair)
                loss_l1 = loss_l1 + l1_t.view(n_layers, n_tr).sum(1)
                loss_giou = loss_giou + gi_t.view(n_layers, n_tr).sum(1)
            l1_pair, gi_pair = pair_loss(boxes_all, lay_i, q_i, b, gt.boxes, g_i)
            loss_l1 = loss_l1.index_add(0, lay_i, l1_pair)
            loss_giou = loss_giou.index_add(0, lay_i, gi_pair)

        fw = self.frame_weights[frame_idx]
        per_layer = torch.stack((loss_l1, loss_giou, loss_label))               # rows in the order of _LOSS_KEYS
        log = per_layer.detach()
        self.log[f"frame{frame_idx}_box_l1_loss"] = log[0, 0]
        self.log[f"frame{frame_idx}_box_giou_loss"] = log[1, 0]
        self.log[f"frame{frame_idx}_label_focal_loss"] = log[2, 0]
        # column 0: the last decoder layer (x frame weight); column 1: the auxiliary layers (x their weights)
        cols = [[fw] + [0.0] * (n_layers - 1)]
        if self._acc.shape[1] == 2:
            cols.append([0.0] + [w * fw for w in self.aux_weights[:n_layers - 1]])
        weights = self._constant(("loss_w", n_layers, fw, self._acc.shape[1]), cols, per_layer.dtype, dev)
        self._acc = self._acc + per_layer @ weights.t()

    def update_tracked_instances(self, model_outputs: dict, tracked_instances: List[TrackInstances]):
        """Refresh the carried tracks from their query slots.  Padded slots (other clips of the batch carrying
        more tracks) sit behind the real ones, so a slice equals the reference's ``[~query_mask]`` selection."""
        nd = self.n_det_queries
        for b, tr in enumerate(tracked_instances):
            n = len(tr)
            if n > 0:
                tr.boxes = batch_item(model_outputs["pred_bboxes"], b)[nd:nd + n]
                tr.logits = batch_item(model_outputs["pred_logits"], b)[nd:nd + n]
                tr.output_embed = batch_item(model_outputs["outputs"], b)[nd:nd + n]
                tr.matched_idx = torch.zeros((0,), dtype=tr.matched_idx.dtype)
                tr.labels = torch.zeros((0,), dtype=tr.matched_idx.dtype)
        return tracked_instances

    # ------------------------------------------------------------------ reference-shaped single-layer losses
    def get_loss_label(self, outputs, gt_trackinstances: List[TrackInstances], idx_to_gts_idx):
        logits, labels = [], []
        for b, gt in enumerate(gt_trackinstances):
            lg = outputs["pred_logits"][b][~outputs["query_mask"][b]]
            lab = torch.full(lg.shape[:1], self.num_classes, dtype=torch.int64, device=self.device)
            q, g = idx_to_gts_idx[b]
            ok = g >= 0
            lab[q[ok]] = gt.labels[g[ok]]
            logits.append(lg)
            labels.append(lab)
        logits, labels = torch.cat(logits), torch.cat(labels)
        one_hot = F.one_hot(labels, self.num_classes + 1)[:, :-1].to(logits.dtype)
        return sigmoid_focal_loss(inputs=logits, targets=one_hot, alpha=0.25, gamma=2)

    @staticmethod
    def get_loss_box(outputs, gt_trackinstances: List[TrackInstances], idx_to_gts_idx):
        pred, tgt = [], []
        for b, gt in enumerate(gt_trackinstances):
            q, g = idx_to_gts_idx[b]
            ok = g >= 0
            pred.append(outputs["pred_bboxes"][b][q[ok]])
            tgt.append(gt.boxes[g[ok]])
        pred, tgt = torch.cat(pred), torch.cat(tgt).to(pred[0].device)
        loss_l1 = F.l1_loss(pred, tgt, reduction="none").sum()
        loss_giou = (1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(pred), box_cxcywh_to_xyxy(tgt)))).sum()
        return loss_l1, loss_giou


def paired_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """IoU of xyxy boxes paired along the leading dims (the diagonal of box_iou_union, without the matrix)."""
    lt = torch.max(a[..., :2], b[..., :2])
    rb = torch.min(a[..., 2:], b[..., 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    return inter / (area_a + area_b - inter)


def paired_giou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Generalised IoU of paired xyxy boxes: elementwise the same arithmetic as diag(generalized_box_iou)."""
    lt = torch.max(a[..., :2], b[..., :2])
    rb = torch.min(a[..., 2:], b[..., 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    union = area_a + area_b - inter
    iou = inter / union
    lt_h = torch.min(a[..., :2], b[..., :2])
    rb_h = torch.max(a[..., 2:], b[..., 2:])
    wh_h = (rb_h - lt_h).clamp(min=0)
    hull = wh_h[..., 0] * wh_h[..., 1]
    return iou - (hull - union) / hull


def sigmoid_focal_loss_per_layer(inputs, targets, alpha: float = 0.25, gamma: float = 2):
    """Focal loss of stacked layers: (n_layers, Nq, K) -> (n_layers,) (mean over classes, sum over queries)."""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(2).sum(1)


def sigmoid_focal_loss(inputs, targets, alpha: float = 0.25, gamma: float = 2):
    """RetinaNet focal loss, mean over classes and sum over queries (criterion.py:438-463)."""
    prob = inputs.sigmoid()
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum()


def build(config: dict) -> ClipCriterion:
    from .memotr import DATASET_NUM_CLASSES
    return ClipCriterion(
        num_classes=DATASET_NUM_CLASSES[config["DATASET"]], matcher=build_matcher(config),
        n_det_queries=config["NUM_DET_QUERIES"], aux_loss=config["AUX_LOSS"],
        weight={"box_l1_loss": config["LOSS_WEIGHT_L1"], "box_giou_loss": config["LOSS_WEIGHT_GIOU"],
                "label_focal_loss": config["LOSS_WEIGHT_FOCAL"]},
        max_frame_length=max(config["SAMPLE_LENGTHS"]), n_aux=config["NUM_DEC_LAYERS"] - 1,
        merge_det_track_layer=config.get("MERGE_DET_TRACK_LAYER", 0), aux_weights=config["AUX_LOSS_WEIGHT"],
        hidden_dim=config["HIDDEN_DIM"], use_dab=config["USE_DAB"])
