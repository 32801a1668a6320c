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

from typing import Dict, List, Tuple

import torch
import torch.distributed
import torch.nn.functional as F

from ..structures.track_instances import TrackInstances
from ..utils.box_ops import box_cxcywh_to_xyxy, box_iou_union, generalized_box_iou
from ..utils.utils import distributed_world_size, is_distributed
from .matcher import HungarianMatcher
from .matcher import build as build_matcher

_LOSS_KEYS = ("box_l1_loss", "box_giou_loss", "label_focal_loss")


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
        self.max_frame_length = max_frame_length
        self.n_aux = n_aux
        self.use_dab = use_dab
        self.frame_weights = [1.0] * max_frame_length
        self.aux_weights = aux_weights
        self.hidden_dim = hidden_dim
        self.merge_det_track_layer = merge_det_track_layer
        self.gt_trackinstances_list = None      # [clip_len][B]
        self.loss: Dict[str, torch.Tensor] = {}
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
        self.n_gts = []
        self.log = {}
        keys = _LOSS_KEYS + tuple("aux_" + k for k in _LOSS_KEYS) if self.aux_loss else _LOSS_KEYS
        self.loss = {k: torch.zeros((), device=device) for k in keys}

    def get_sum_loss_dict(self, loss_dict: dict):
        def w(name):
            for k in _LOSS_KEYS:
                if k in name:
                    return self.weight[k]
        return sum(w(k) * v for k, v in loss_dict.items())

    def get_mean_by_n_gts(self) -> Tuple[Dict, Dict]:
        counts = torch.as_tensor([float(sum(self.n_gts))] + [float(n) for n in self.n_gts], dtype=torch.float,
                                 device=self.device)
        if is_distributed():
            torch.distributed.all_reduce(counts)
        counts = torch.clamp(counts / distributed_world_size(), min=1).tolist()
        total, per_frame = counts[0], counts[1:]
        loss = {k: v / total for k, v in self.loss.items()}
        log = {}
        for k, v in self.log.items():
            for i, n in enumerate(per_frame):
                if f"frame{i}" in k:
                    log[k] = (float(v) / n, 1)
                    break
        return loss, log

    # ------------------------------------------------------------------ one frame
    def process_single_frame(self, model_outputs: dict, tracked_instances: List[TrackInstances], frame_idx: int):
        nd = self.n_det_queries
        gts = self.gt_trackinstances_list[frame_idx]
        B = len(tracked_instances)
        tracked_instances = self.update_tracked_instances(model_outputs, tracked_instances)

        # which ground truth does every carried track own (-1: its identity left the scene / never had one)
        untracked, untracked_global = [], []
        for b in range(B):
            tr, gt = tracked_instances[b], gts[b]
            if len(tr) > 0 and len(gt) > 0:
                eq = tr.ids[:, None] == gt.ids[None, :]
                tr.matched_idx = torch.where(eq.any(1), eq.float().argmax(1), torch.full_like(tr.ids, -1))
            else:
                tr.matched_idx = torch.full((len(tr),), -1, dtype=torch.long, device=gt.ids.device)
            free = torch.ones((len(gt),), dtype=torch.bool, device=gt.ids.device)
            free[tr.matched_idx[tr.matched_idx >= 0]] = False
            untracked.append(gt[free] if len(gt) > 0 else gt)
            untracked_global.append(torch.nonzero(free).squeeze(1))

        # every assignment problem of this frame (last layer + aux layers) from one host transfer
        layers = [model_outputs] + (list(model_outputs["aux_outputs"]) if self.aux_loss else [])
        costs, plan = [], []
        for li, out in enumerate(layers):
            against_all = li > 0 and (li - 1) < self.merge_det_track_layer
            for b in range(B):
                tgt = gts[b] if against_all else untracked[b]
                costs.append(self.matcher.cost_matrix(out["pred_logits"][b, :nd].detach(),
                                                      out["pred_bboxes"][b, :nd].detach(), tgt.labels, tgt.boxes))
                plan.append((li, b, against_all))
        solved = self.matcher.solve_many(costs)
        dev = self.device
        match = {}
        for (li, b, against_all), (qi, tj) in zip(plan, solved):
            qi, tj = qi.to(dev), tj.to(dev)
            match[(li, b)] = (qi, tj if against_all else untracked_global[b][tj])

        # new tracks from the matched detect queries of the last layer
        new_tracks = []
        for b in range(B):
            q_idx, gt_idx = match[(0, b)]
            nt = TrackInstances(frame_height=tracked_instances[b].frame_height,
                                frame_width=tracked_instances[b].frame_width,
                                hidden_dim=tracked_instances[b].hidden_dim, num_classes=self.num_classes)
            nt.ids = gts[b].ids[gt_idx]
            nt.matched_idx = gt_idx
            queries = model_outputs["aux_outputs"][-1]["queries"][b][q_idx]
            nt.query_embed = queries if self.use_dab else torch.cat(
                (model_outputs["det_query_embed"][q_idx][:, :self.hidden_dim], queries), dim=-1)
            nt.ref_pts = model_outputs["last_ref_pts"][b][q_idx]
            nt.output_embed = model_outputs["outputs"][b][q_idx]
            nt.boxes = model_outputs["pred_bboxes"][b][q_idx]
            nt.logits = model_outputs["pred_logits"][b][q_idx]
            nt.iou = torch.zeros((len(gt_idx),), dtype=torch.float)
            new_tracks.append(nt.to(dev))

        tracked_pairs = [(torch.arange(nd, nd + len(tracked_instances[b]), device=dev),
                          tracked_instances[b].matched_idx.to(dev)) for b in range(B)]

        def pairs_for(li):
            res = []
            for b in range(B):
                q, g = match[(li, b)]
                if li > 0 and (li - 1) < self.merge_det_track_layer:
                    res.append((q, g))          # early layers: detect queries against all ground truths
                else:
                    res.append((torch.cat((q, tracked_pairs[b][0])), torch.cat((g, tracked_pairs[b][1]))))
            return res

        main_pairs = pairs_for(0)
        loss_label = self.get_loss_label(model_outputs, gts, main_pairs)
        loss_l1, loss_giou = self.get_loss_box(model_outputs, gts, main_pairs)
        fw = self.frame_weights[frame_idx]
        self.loss["box_l1_loss"] = self.loss["box_l1_loss"] + loss_l1 * fw
        self.loss["box_giou_loss"] = self.loss["box_giou_loss"] + loss_giou * fw
        self.loss["label_focal_loss"] = self.loss["label_focal_loss"] + loss_label * fw
        self.log[f"frame{frame_idx}_box_l1_loss"] = loss_l1.detach()
        self.log[f"frame{frame_idx}_box_giou_loss"] = loss_giou.detach()
        self.log[f"frame{frame_idx}_label_focal_loss"] = loss_label.detach()
        self.n_gts.append(sum(len(g) for g in gts))

        if self.aux_loss:
            for i, aux in enumerate(model_outputs["aux_outputs"]):
                pairs = pairs_for(i + 1)
                a_label = self.get_loss_label(aux, gts, pairs)
                a_l1, a_giou = self.get_loss_box(aux, gts, pairs)
                w = fw * self.aux_weights[i]
                self.loss["aux_box_l1_loss"] = self.loss["aux_box_l1_loss"] + a_l1 * w
                self.loss["aux_box_giou_loss"] = self.loss["aux_box_giou_loss"] + a_giou * w
                self.loss["aux_label_focal_loss"] = self.loss["aux_label_focal_loss"] + a_label * w

        # detections nobody claimed, handed to the query updater
        unmatched = []
        n_det_out = len(model_outputs["det_query_embed"])
        for b in range(B):
            taken = torch.zeros((n_det_out,), dtype=torch.bool, device=dev)
            q_all = main_pairs[b][0]
            taken[q_all[q_all < n_det_out]] = True
            idx = torch.nonzero(~taken).squeeze(1)
            d = TrackInstances(hidden_dim=model_outputs["outputs"].shape[-1],
                               num_classes=model_outputs["pred_logits"].shape[-1]).to(dev)
            d.ref_pts = model_outputs["init_ref_pts"][b][idx]
            d.output_embed = model_outputs["outputs"][b][idx]
            d.logits = model_outputs["pred_logits"][b][idx]
            d.boxes = model_outputs["pred_bboxes"][b][idx]
            queries = model_outputs["aux_outputs"][-1]["queries"][b][idx]
            d.query_embed = queries if self.use_dab else torch.cat(
                (model_outputs["det_query_embed"][idx][:, :self.hidden_dim], queries), dim=-1)
            d.ids = -torch.ones((len(idx),), dtype=torch.long, device=dev)
            d.matched_idx = -torch.ones((len(idx),), dtype=torch.long, device=dev)
            d.iou = torch.zeros((len(idx),), dtype=torch.float, device=dev)
            unmatched.append(d)

        for b in range(B):
            tracked_instances[b] = tracked_instances[b].to(dev)
            for tr in (new_tracks[b], tracked_instances[b]):
                has = tr.matched_idx >= 0
                if len(tr) > 0:
                    iou = box_iou_union(box_cxcywh_to_xyxy(tr.boxes[has]),
                                        box_cxcywh_to_xyxy(gts[b].boxes[tr.matched_idx[has]]))[0]
                    tr.iou[has] = torch.diag(iou)
        return tracked_instances, new_tracks, unmatched

    def update_tracked_instances(self, model_outputs: dict, tracked_instances: List[TrackInstances]):
        nd = self.n_det_queries
        for b, tr in enumerate(tracked_instances):
            if len(tr) > 0:
                keep = ~model_outputs["query_mask"][b][nd:]
                tr.boxes = model_outputs["pred_bboxes"][b][nd:][keep]
                tr.logits = model_outputs["pred_logits"][b][nd:][keep]
                tr.output_embed = model_outputs["outputs"][b][nd:][keep]
                tr.matched_idx = torch.zeros((0,), dtype=tr.matched_idx.dtype)
                tr.labels = torch.zeros((0,), dtype=tr.matched_idx.dtype)
        return tracked_instances

    # ------------------------------------------------------------------ losses
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
