"""MeMOTR per-frame model (API of the reference's models/memotr.py:28-321).

``forward(frame: NestedTensor, tracks: list[TrackInstances]) -> dict`` and
``postprocess_single_frame(previous, new, unmatched, no_augment=False)`` are what train_engine.py /
submit_engine.py call; parameter names are state-dict compatible with the reference.
Query / reference / mask assembly happens on the device the tracks already live on (the reference builds
them on the CPU and copies, models/memotr.py:222-278 -- a device->host->device round trip per frame).
"""
from __future__ import annotations

import os

import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from ..structures.track_instances import TrackInstances
from ..utils.nested_tensor import NestedTensor
from ..utils.utils import inverse_sigmoid
from .backbone import BackboneWithPE
from .backbone import build as build_backbone_with_pe
from .deformable_transformer import DeformableTransformer
from .deformable_transformer import build as build_deformable_transformer
from .mlp import MLP
from .query_updater import build as build_query_updater
from .utils import get_clones


class MeMOTR(nn.Module):
    def __init__(self, backbone: BackboneWithPE, transformer: DeformableTransformer, query_updater: nn.Module,
                 num_classes: int, n_det_queries: int, n_feature_levels: int, hidden_dim: int, ffn_dim: int,
                 dropout: float, aux_loss: bool = True, with_box_refine: bool = True, use_checkpoint: bool = False,
                 checkpoint_level: int = 2, use_dab: bool = False, visualize: bool = False):
        super().__init__()
        self.num_classes = num_classes
        self.n_det_queries = n_det_queries
        self.n_feature_levels = n_feature_levels
        self.hidden_dim = hidden_dim
        self.ffn_dim = ffn_dim
        self.dropout = dropout
        self.aux_loss = aux_loss
        self.with_box_refine = with_box_refine
        self.use_checkpoint = use_checkpoint
        self.checkpoint_level = checkpoint_level
        self.use_dab = use_dab
        self.visualize = visualize

        self.backbone = backbone
        self.transformer = transformer
        self.query_updater = query_updater
        self.class_embed = nn.Linear(hidden_dim, num_classes)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 4, 3)
        if use_dab:
            self.det_anchor = nn.Parameter(torch.randn(n_det_queries, 4))
            self.det_query_embed = nn.Parameter(torch.randn(n_det_queries, hidden_dim))
        else:
            self.det_query_embed = nn.Parameter(torch.randn(n_det_queries, hidden_dim * 2))
        assert n_feature_levels > 1
        channels = backbone.n_inter_channels()
        projs = [nn.Sequential(nn.Conv2d(c, hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim))
                 for c in channels]
        projs += [nn.Sequential(nn.Conv2d(channels[-1], hidden_dim, kernel_size=3, stride=2, padding=1),
                                nn.GroupNorm(32, hidden_dim))
                  for _ in range(n_feature_levels - backbone.n_inter_layers())]
        self.feature_projs = nn.ModuleList(projs)

        prior_prob = 0.01
        self.class_embed.bias.data = torch.ones(num_classes) * (-math.log((1 - prior_prob) / prior_prob))
        nn.init.constant_(self.bbox_embed.layers[-1].weight.data, 0)
        nn.init.constant_(self.bbox_embed.layers[-1].bias.data, 0)
        for proj in self.feature_projs:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        n_dec = transformer.get_n_dec_layers()
        if with_box_refine:
            self.class_embed = get_clones(self.class_embed, n_dec)
            self.bbox_embed = get_clones(self.bbox_embed, n_dec)
            nn.init.constant_(self.bbox_embed[0].layers[-1].bias.data[2:], -2.0)
            self.transformer.set_refine_bbox_embed(self.bbox_embed)   # aliased: both names in the state dict
        else:
            nn.init.constant_(self.bbox_embed.layers[-1].bias.data[2:], -2.0)
            self.class_embed = nn.ModuleList([self.class_embed for _ in range(n_dec)])
            self.bbox_embed = nn.ModuleList([self.bbox_embed for _ in range(n_dec)])

    # ------------------------------------------------------------------ forward
    def forward(self, frame: Optional[NestedTensor] = None, tracks: Optional[List[TrackInstances]] = None,
                encoded: Optional[dict] = None, stage: Optional[str] = None):
        """``model(frame, tracks)`` is the reference contract (models/memotr.py:94-160).  Two extra keywords split
        it so a training loop can queue the query-independent half of the NEXT frame on the GPU while the host
        still waits for / solves this frame's assignment: ``model(frame=f, stage="encode")`` returns the encoder
        result, ``model(tracks=t, encoded=enc)`` finishes the frame.  Both go through ``forward`` so a
        DistributedDataParallel wrapper sees every call."""
        if stage == "features":
            return self._encode_frame_eager(frame, features_only=True)
        if stage == "encode_features":                 # frame = (features, lo, hi)
            return self.encode_features(*frame)
        if encoded is None:
            encoded = self._encode_frame_eager(frame) if stage == "encode_eager" else self.encode_frame(frame)
        if stage in ("encode", "encode_eager"):
            return encoded
        return self.decode_frame(encoded, tracks)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_encode_graphs", None)          # captured hipGraphs are per-process objects
        state.pop("_infer_graphs", None)
        state.pop("_cls_stack", None)              # per-clip cache of the class heads' stacked parameters
        return state

    def infer_graphs(self):
        g = self.__dict__.get("_infer_graphs")
        if g is None:
            from .infer_graphs import InferGraphs
            g = self.__dict__["_infer_graphs"] = InferGraphs(self)
        return g

    def encode_graphs(self):
        g = self.__dict__.get("_encode_graphs")
        if g is None:
            from .encode_graphs import EncodeGraphs
            g = self.__dict__["_encode_graphs"] = EncodeGraphs(self)
        return g

    def encode_frame(self, frame: NestedTensor) -> dict:
        """Backbone -> feature projections -> encoder (independent of the track queries); replayed from a hipGraph
        pair where that pays (models/encode_graphs.py: the launch-bound bf16 step), else kernel by kernel."""
        if not torch.is_grad_enabled():            # inference: forward-only capture (models/infer_graphs.py)
            infer = self.infer_graphs()
            if infer.encode_usable(frame):
                enc = infer.run_encode(frame)
                if enc is not None:
                    return enc
            return self._encode_frame_eager(frame)
        graphs = self.encode_graphs()
        if graphs.usable(frame):
            enc = graphs.run(frame, getattr(frame, "encode_slot", 0))
            if enc is not None:
                return enc
        return self._encode_frame_eager(frame)

    def _encode_frame_eager(self, frame: NestedTensor, features_only: bool = False) -> dict:
        """Backbone -> feature projections -> encoder (independent of the track queries).  ``features_only``: stop in
        front of the encoder and return its inputs (``encode_features`` finishes any sub-batch of them)."""
        if self.use_checkpoint and self.checkpoint_level != 3:
            features, pos = checkpoint(self.backbone, frame, use_reentrant=False)
        else:
            features, pos = self.backbone(frame)
        pos = list(pos)
        srcs, masks = [], []
        for lvl, feat in enumerate(features):
            src, mask = feat.decompose()
            srcs.append(self.feature_projs[lvl](src))
            masks.append(mask)
        for lvl in range(len(srcs), self.n_feature_levels):   # extra levels: stride-2 conv on the raw last map
            src = self.feature_projs[lvl](features[-1].tensors if lvl == len(features) else srcs[-1])
            mask = F.interpolate(frame.masks[None].float(), size=src.shape[-2:])[0].to(torch.bool)
            level = NestedTensor(src, mask, getattr(frame, "sizes", None))
            pos.append(self.backbone.position_embedding(level).to(src.device))
            srcs.append(src)
            masks.append(mask)
        if features_only:
            return {"srcs": srcs, "masks": masks, "pos": pos, "sizes": getattr(frame, "sizes", None)}
        return self.transformer.encode(srcs=srcs, masks=masks, pos_embeds=pos, geometry=getattr(frame, "sizes", None))

    def encode_features(self, feats: dict, lo: int, hi: int) -> dict:
        """The encoder over images lo..hi-1 of a ``stage="features"`` result (backbone + projections of a whole clip
        in one batch -- the convolutions want the large batch -- the transformer encoder in smaller groups, so that a
        training loop has GPU work queued while the host runs a frame's launch-bound decoder chain; engine.py)."""
        sizes = feats["sizes"]
        geometry = None if sizes is None else (sizes[0],) + tuple(sizes[1 + lo:1 + hi])
        return self.transformer.encode(srcs=[s[lo:hi] for s in feats["srcs"]], masks=[m[lo:hi] for m in feats["masks"]],
                                       pos_embeds=[p[lo:hi] for p in feats["pos"]], geometry=geometry)

    def _class_head_stack(self, clip_key):
        """The class heads' weights (n, C, K) and biases (n, 1, K) as two stacks.  With a ``clip_key`` (the training
        loop's per-clip object) the stacks are made once per clip and shared by its frames: the frames' gradients then
        meet in one add per frame and the stack's backward (twelve slices handed to the parameters) runs once per
        clip, not once per frame."""
        cache = self.__dict__.get("_cls_stack")
        if clip_key is not None and cache is not None and cache[0] is clip_key and cache[3] == torch.is_grad_enabled():
            return cache[1], cache[2]
        w = torch.stack([m.weight for m in self.class_embed]).transpose(1, 2)
        b = torch.stack([m.bias for m in self.class_embed])[:, None, :]
        if clip_key is not None:
            self.__dict__["_cls_stack"] = (clip_key, w, b, torch.is_grad_enabled())
        return w, b

    def decode_frame(self, encoded: dict, tracks: List[TrackInstances]) -> dict:
        """Query assembly -> decoder -> heads over an ``encode_frame`` result.

        Under autocast (the bf16 extension) this half runs as a float32 island: its GEMMs have a few hundred rows --
        nothing for the matrix cores to win, while every cast is one more launch in a launch-bound chain -- and in
        float32 the decoder loop keeps its hipGraphs and hand-written kernels.  bf16 stays where the FLOPs are
        (backbone, encoder).  MEMOTR_AUTOCAST_DECODER=1 restores plain autocast semantics."""
        if torch.is_autocast_enabled() and os.environ.get("MEMOTR_AUTOCAST_DECODER", "0") != "1":
            enc32 = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in encoded.items()}
            with torch.autocast(device_type=encoded["memory"].device.type, enabled=False):
                return self.decode_frame(enc32, tracks)
        device = encoded["memory"].device
        reference_points = self.get_reference_points(tracks).to(device)     # (B, Nd+Nt, 4) logit space
        query_embed = self.get_query_embed(tracks).to(device)               # (B, Nd+Nt, C | 2C)
        query_mask = self.get_query_mask(tracks).to(device)                 # (B, Nd+Nt) bool

        outputs, init_reference, inter_references, inter_queries, refined = self.transformer.decode(
            encoded, query_embed=query_embed, ref_pts=reference_points, query_mask=query_mask, return_boxes=True)
        assert outputs.ndim == 4, \
            f"Deformable Transformer's outputs should have shape (n_dec_layers, B, Nd+Nq, C, but get n_dim={outputs.ndim}"
        classes, boxes = [], []
        # with box refinement the decoder already evaluated this head (same layer output, same reference, aliased
        # bbox_embed weights) to move its anchors: reuse that one evaluation instead of repeating it per layer
        # (DAB only: without DAB the decoder refines from the 2-d slice of the reference while this head adds the
        # full 4-d inverse sigmoid at level 0, reference models/memotr.py:148-158 -- not the same boxes for tracks)
        reuse = refined is not None and self.use_dab and self.transformer.decoder.bbox_embed is self.bbox_embed
        per_layer = outputs.unbind(0)          # one backward node for the six uses (a select each: zero-fill + copy)
        n_lvl, B_, Nq_, C_ = outputs.shape
        batched_heads = reuse and all(isinstance(m, nn.Linear) for m in self.class_embed)
        if batched_heads:      # the per-layer class heads as ONE batched product (6 x {GEMM, bias} forward, 18 kernels backward)
            w, b = self._class_head_stack(encoded.get("clip_key"))                          # (n, C, K), (n, 1, K)
            classes = torch.baddbmm(b, outputs.reshape(n_lvl, B_ * Nq_, C_), w).view(n_lvl, B_, Nq_, -1)
        for lvl in range(outputs.shape[0]):
            if not batched_heads:
                classes.append(self.class_embed[lvl](per_layer[lvl]))
            if reuse:
                continue
            reference = inverse_sigmoid(init_reference if lvl == 0 else inter_references[lvl - 1])
            box = self.bbox_embed[lvl](per_layer[lvl])
            if reference.shape[-1] == 4:
                box = box + reference
            else:
                assert reference.shape[-1] == 2, f"Reference should have only 2 coord, but get {reference.shape[-1]}."
                box = torch.cat((box[..., :2] + reference, box[..., 2:]), -1)
            boxes.append(box.sigmoid())
        if not batched_heads:
            classes = torch.stack(classes, dim=0)
        boxes = refined if reuse else torch.stack(boxes, dim=0)
        res = {
            "pred_logits": classes[-1],
            "pred_bboxes": boxes[-1],
            "last_ref_pts": inverse_sigmoid(inter_references[-2, :, :, :]),
            "query_mask": query_mask,
            "det_query_embed": query_embed[0][:self.n_det_queries],
            "init_ref_pts": inverse_sigmoid(init_reference),
        }
        if self.aux_loss:
            res["aux_outputs"] = self.set_aux_loss(classes, boxes, query_mask, inter_queries)
            # the same predictions as stacks over the decoder layers (last = the main output): the criterion reads
            # these instead of re-stacking the per-layer views above
            res["pred_logits_all"], res["pred_bboxes_all"] = classes, boxes
        res["outputs"] = per_layer[-1]
        return res

    @torch.jit.unused
    def set_aux_loss(self, output_classes, output_bboxes, query_mask, queries):
        return [{"pred_logits": a, "pred_bboxes": b, "query_mask": query_mask, "queries": c}
                for a, b, c in zip(output_classes[:-1], output_bboxes[:-1], queries[1:])]

    # ------------------------------------------------------------------ query assembly
    def get_det_reference_points(self) -> torch.Tensor:
        if self.use_dab:
            return self.det_anchor
        return self.transformer.reference_points(self.det_query_embed[:, :self.hidden_dim])

    def _pad_stack(self, parts: List[torch.Tensor], width: int) -> torch.Tensor:
        """(B, max_len, width) zero-padded stack of per-clip (n_i, width) tensors, on the model's device.
        Built from views / pads / one stack: writing the parts into a zero tensor slice by slice would put a
        CopySlices node (zero-fill + strided copies, forward and backward) on every carried track tensor."""
        device = self.det_query_embed.device
        max_len = max(p.shape[0] for p in parts)
        if max_len == 0:
            return torch.zeros((len(parts), 0, width), device=device)
        parts = [p.to(device) for p in parts]
        if len(parts) == 1:
            return parts[0][None]
        return torch.stack([p if p.shape[0] == max_len else F.pad(p, (0, 0, 0, max_len - p.shape[0])) for p in parts])

    def get_track_reference_points(self, tracks: List[TrackInstances]):
        return self._pad_stack([t.ref_pts for t in tracks], 4)

    def get_track_query_embed(self, tracks: List[TrackInstances]):
        return self._pad_stack([t.query_embed for t in tracks],
                               self.hidden_dim if self.use_dab else self.hidden_dim * 2)

    def get_reference_points(self, tracks: List[TrackInstances]):
        det = self.get_det_reference_points()
        det = det[None] if det.dim() == 2 and len(tracks) == 1 else det.repeat(len(tracks), 1, 1)
        if det.shape[-1] == 2:
            det = torch.cat((det, torch.zeros_like(det)), dim=-1)
        return torch.cat((det, self.get_track_reference_points(tracks).to(det.device)), dim=1)

    def get_query_embed(self, tracks: List[TrackInstances]):
        det = self.det_query_embed[None] if len(tracks) == 1 else self.det_query_embed.repeat(len(tracks), 1, 1)
        return torch.cat((det, self.get_track_query_embed(tracks).to(det.device)), dim=1)

    def get_query_mask(self, tracks: List[TrackInstances]):
        """True on padded track slots; a clip with no track at all keeps its padding unmasked (reference :271-274)."""
        lens = [len(t.query_embed) for t in tracks]
        max_len = max(lens)
        device = self.det_query_embed.device
        mask = torch.zeros((len(tracks), self.n_det_queries + max_len), dtype=torch.bool, device=device)
        for i, n in enumerate(lens):
            if n > 0:
                mask[i, self.n_det_queries + n:] = True
        # host-side knowledge the attention can use without reading the mask back: nothing is masked unless two
        # clips of the batch carry different (non-zero) numbers of tracks -- never at batch size 1
        mask._no_padding = not any(0 < n < max_len for n in lens)
        return mask

    def postprocess_single_frame(self, previous_tracks: List[TrackInstances], new_tracks: List[TrackInstances],
                                 unmatched_dets: Optional[List[TrackInstances]], no_augment: bool = False,
                                 frame_slot: int = None, clip_key=None):
        """Query updating between frames (a float32 island under autocast, like ``decode_frame``).  ``frame_slot`` /
        ``clip_key``: see ``QueryUpdater.forward`` (the training loop's hipGraph slot of this frame)."""
        if torch.is_autocast_enabled() and os.environ.get("MEMOTR_AUTOCAST_DECODER", "0") != "1":
            with torch.autocast(device_type="cuda", enabled=False):
                return self.query_updater(previous_tracks, new_tracks, unmatched_dets, no_augment,
                                          frame_slot=frame_slot, clip_key=clip_key)
        return self.query_updater(previous_tracks, new_tracks, unmatched_dets, no_augment, frame_slot=frame_slot,
                                  clip_key=clip_key)


DATASET_NUM_CLASSES = {"DanceTrack": 1, "SportsMOT": 1, "MOT17": 1, "MOT17_SPLIT": 1, "BDD100K": 8}


def build(config: dict) -> MeMOTR:
    assert config["DATASET"] in DATASET_NUM_CLASSES, f"Do not know the class num of {config['DATASET']} dataset."
    return MeMOTR(
        backbone=build_backbone_with_pe(config), transformer=build_deformable_transformer(config),
        query_updater=build_query_updater(config), num_classes=DATASET_NUM_CLASSES[config["DATASET"]],
        n_det_queries=config["NUM_DET_QUERIES"], n_feature_levels=config["NUM_FEATURE_LEVELS"],
        hidden_dim=config["HIDDEN_DIM"], ffn_dim=config["FFN_DIM"], dropout=config["DROPOUT"], aux_loss=True,
        with_box_refine=True, use_checkpoint=config["USE_CHECKPOINT"], checkpoint_level=config["CHECKPOINT_LEVEL"],
        use_dab=config["USE_DAB"], visualize=config["VISUALIZE"])
