"""Online track management for inference (contract of the reference's models/runtime_tracker.py:13-101).

Same thresholds, id assignment and TrackInstances fields as the reference; the per-track python loop
with ``.item()`` reads (runtime_tracker.py:43-54) is replaced by tensor ops on the device.
The optional motion post-process (``USE_MOTION``, off in every shipped config) is not implemented.
"""
from __future__ import annotations

from typing import List

import torch

from ..structures.track_instances import TrackInstances
from .utils import logits_to_scores


class RuntimeTracker:
    def __init__(self, det_score_thresh: float = 0.7, track_score_thresh: float = 0.6, miss_tolerance: int = 5,
                 use_motion: bool = False, motion_min_length: int = 3, motion_max_length: int = 5,
                 visualize: bool = False, use_dab: bool = True):
        if use_motion:
            raise NotImplementedError("USE_MOTION is not supported (unused by the shipped configs)")
        self.det_score_thresh = det_score_thresh
        self.track_score_thresh = track_score_thresh
        self.miss_tolerance = miss_tolerance
        self.max_obj_id = 0
        self.use_motion = False
        self.visualize = visualize
        self.use_dab = use_dab

    def update(self, model_outputs: dict, tracks: List[TrackInstances]):
        assert len(tracks) == 1
        t = tracks[0]
        scores_all = logits_to_scores(model_outputs["pred_logits"])
        model_outputs["scores"] = scores_all
        n_dets = len(model_outputs["det_query_embed"])

        # existing tracks: refresh from the track-query slots, age the ones that scored low
        t.boxes = model_outputs["pred_bboxes"][0][n_dets:]
        t.logits = model_outputs["pred_logits"][0][n_dets:]
        t.output_embed = model_outputs["outputs"][0][n_dets:]
        t.scores = logits_to_scores(t.logits)
        if len(t) > 0:
            own = t.scores.gather(1, t.labels[:, None]).squeeze(1)
            t.disappear_time = torch.where(own < self.track_score_thresh, t.disappear_time + 1,
                                           torch.zeros_like(t.disappear_time))
            t.ids = torch.where(t.disappear_time >= self.miss_tolerance, torch.full_like(t.ids, -1), t.ids)

        # newborn targets from the detect-query slots.  ONE nonzero (= one device synchronisation) for all the
        # fields: a boolean index per field costs a blocking device->host copy each, ~1 ms apiece on this runtime
        # (seven of them were most of an inference frame's host time, tools/infer_gaps.py)
        keep = torch.max(scores_all[0][:n_dets], dim=-1).values >= self.det_score_thresh
        idx = keep.nonzero().squeeze(1)
        pick = lambda x: x[0][:n_dets].index_select(0, idx)  # noqa: E731
        new = TrackInstances(hidden_dim=t.hidden_dim, num_classes=t.num_classes)
        new.logits = pick(model_outputs["pred_logits"])
        new.boxes = pick(model_outputs["pred_bboxes"])
        new.ref_pts = pick(model_outputs["last_ref_pts"])
        new.scores = pick(scores_all)
        new.output_embed = pick(model_outputs["outputs"])
        queries = pick(model_outputs["aux_outputs"][-1]["queries"])
        if self.use_dab:
            new.query_embed = queries
        else:
            new.query_embed = torch.cat((model_outputs["det_query_embed"].index_select(0, idx)[:, :256], queries), dim=-1)
        device = new.logits.device
        n_new = new.logits.shape[0]
        new.disappear_time = torch.zeros((n_new,), dtype=torch.long, device=device)
        new.labels = torch.max(new.scores, dim=-1).indices
        new.ids = torch.arange(self.max_obj_id, self.max_obj_id + n_new, dtype=torch.long, device=device)
        self.max_obj_id += n_new
        return tracks, [new.to(device)]
