"""Long-term-memory query updater (reference models/query_updater.py:18-271).

Between frames: pick the tracks that stay alive, then rewrite their query embeddings from the
short-term memory (last output), the long-term memory (EMA of outputs) and one memory-attention layer.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from ..structures.track_instances import TrackInstances
from ..utils.box_ops import box_cxcywh_to_xyxy, box_iou_union
from ..functions.clip_ops import add_layer_norm
from ..modules.attention import memory_attention
from ..utils.utils import inverse_sigmoid
from .ffn import FFN
from .mlp import MLP
from .utils import logits_to_scores, pos_to_pos_embed


PACKED_TRACKS = __import__("os").environ.get("MEMOTR_PACKED_TRACKS", "1") != "0"


class QueryUpdater(nn.Module):
    def __init__(self, hidden_dim: int, ffn_dim: int, tp_drop_ratio: float, fp_insert_ratio: float, dropout: float,
                 use_checkpoint: bool, use_dab: bool, update_threshold: float, long_memory_lambda: float,
                 visualize: bool = False):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.ffn_dim = ffn_dim
        self.tp_drop_ratio = tp_drop_ratio
        self.fp_insert_ratio = fp_insert_ratio
        self.dropout = dropout
        self.use_checkpoint = use_checkpoint
        self.use_dab = use_dab
        self.visualize = visualize
        self.update_threshold = update_threshold
        self.long_memory_lambda = long_memory_lambda

        self.confidence_weight_net = nn.Sequential(MLP(hidden_dim, hidden_dim, hidden_dim, 2), nn.Sigmoid())
        self.short_memory_fusion = MLP(2 * hidden_dim, 2 * hidden_dim, hidden_dim, 2)
        self.memory_attn = nn.MultiheadAttention(embed_dim=hidden_dim, num_heads=8, batch_first=True)
        self.memory_dropout = nn.Dropout(dropout)
        self.memory_norm = nn.LayerNorm(hidden_dim)
        self.memory_ffn = FFN(d_model=hidden_dim, d_ffn=ffn_dim, dropout=dropout)
        self.query_feat_dropout = nn.Dropout(dropout)
        self.query_feat_norm = nn.LayerNorm(hidden_dim)
        self.query_feat_ffn = FFN(d_model=hidden_dim, d_ffn=ffn_dim, dropout=dropout)
        self.query_pos_head = MLP(hidden_dim * 2, hidden_dim, hidden_dim, 2)
        if not self.use_dab:     # Deformable-DETR variant also refreshes the positional half
            self.linear_pos1 = nn.Linear(256, 256)
            self.linear_pos2 = nn.Linear(256, 256)
            self.norm_pos = nn.LayerNorm(256)
            self.activation = nn.ReLU(inplace=True)
        self.reset_parameters()

    def reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def forward(self, previous_tracks: List[TrackInstances], new_tracks: List[TrackInstances],
                unmatched_dets: Optional[List[TrackInstances]], no_augment: bool = False, frame_slot: int = None,
                clip_key=None):
        """``frame_slot`` (the frame's index inside its clip) and ``clip_key`` (any object, one per clip) come from the
        training loop: with them the embedding update replays from the hipGraph pair of that slot
        (models/updater_graphs.py); without them -- and whenever a capture is not possible -- it runs kernel by kernel."""
        tracks = self.select_active_tracks(previous_tracks, new_tracks, unmatched_dets, no_augment=no_augment)
        return self.update_tracks_embedding(tracks, frame_slot=frame_slot, clip_key=clip_key)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_updater_graphs", None)         # captured hipGraphs are per-process objects
        return state

    def graphs(self):
        g = self.__dict__.get("_updater_graphs")
        if g is None:
            from .updater_graphs import UpdaterGraphs
            g = self.__dict__["_updater_graphs"] = UpdaterGraphs(self)
        return g

    # ------------------------------------------------------------------ embedding update
    FIELDS = ("logits", "boxes", "ref_pts", "output_embed", "long_memory", "last_output", "query_embed")

    def update_fields(self, logits, boxes, ref_pts, out_embed, long_memory_in, last_output, query_embed, key_mask=None):
        """The update of reference models/query_updater.py:96-158 as a function of the track fields (n rows each):
        returns the new (ref_pts, long_memory, last_output, query_embed).  ``key_mask`` (1, n) bool marks rows that are
        padding (the captured form runs on a padded row count): their keys take no part in the memory attention, and
        nothing else mixes rows."""
        C = self.hidden_dim
        lam = self.long_memory_lambda
        scores = torch.max(logits_to_scores(logits), dim=1).values
        is_pos = scores > self.update_threshold
        pos_col = is_pos.reshape(-1, 1)
        # (masked assignments of the reference are written as selects: no boolean-index synchronisation)
        ref_pts = torch.where(pos_col, inverse_sigmoid(boxes.detach()), ref_pts)

        query_pos = self.query_pos_head(pos_to_pos_embed(ref_pts.sigmoid(), num_pos_feats=C // 2))
        long_memory = long_memory_in.detach()

        confidence = self.confidence_weight_net(out_embed)
        short_memory = self.short_memory_fusion(torch.cat((confidence * out_embed, last_output), dim=-1))

        q = (short_memory + query_pos)[None]
        k = (long_memory + query_pos)[None]
        attn = memory_attention(self.memory_attn, q, k, out_embed[None], key_padding_mask=key_mask)[0]
        tgt = self.memory_ffn(add_layer_norm(out_embed, self.memory_dropout(attn), self.memory_norm))
        query_feat = self.query_feat_ffn(add_layer_norm(long_memory, self.query_feat_dropout(tgt),
                                                        self.query_feat_norm))

        new_long = (1 - lam) * long_memory + lam * out_embed
        new_long_memory = long_memory_in * ~pos_col + new_long * pos_col
        new_last_output = last_output * ~pos_col + out_embed * pos_col

        if self.use_dab:
            new_query = torch.where(pos_col, query_feat, query_embed)
        else:
            refreshed = self.norm_pos(query_embed[:, :C]
                                      + self.linear_pos2(self.activation(self.linear_pos1(out_embed))))
            new_query = torch.cat((torch.where(pos_col, refreshed, query_embed[:, :C]),
                                   torch.where(pos_col, query_feat, query_embed[:, C:])), dim=-1)
        return ref_pts, new_long_memory, new_last_output, new_query

    def update_tracks_embedding(self, tracks: List[TrackInstances], frame_slot: int = None, clip_key=None):
        for b, t in enumerate(tracks):
            fields = tuple(getattr(t, f) for f in self.FIELDS)
            new = None
            if frame_slot is not None and self.graphs().usable(fields):
                packed = t._packed_base()           # the fields as adjacent columns of one tensor, in FIELDS' order?
                if packed is not None and tuple(packed[1][:len(self.FIELDS)]) != tuple(self.FIELDS):
                    packed = None
                new = self.graphs().run((frame_slot, b), fields, clip_key,
                                        packed=None if packed is None else packed[0])
            if new is None:
                new = self.update_fields(*fields)
            t.ref_pts, t.long_memory, t.last_output, t.query_embed = new
        return tracks

    # ------------------------------------------------------------------ track selection
    def _seed_memories(self, t: TrackInstances):
        t.last_output = t.output_embed
        t.long_memory = t.query_embed if self.use_dab else t.query_embed[:, self.hidden_dim:]

    def _fake_track(self, n_logits: int) -> TrackInstances:
        """One random track with id -2: keeps every parameter of this module in the autograd graph when a
        clip has no active track (DDP runs with find_unused_parameters=False)."""
        device = next(self.query_feat_ffn.parameters()).device
        C = self.hidden_dim
        f = TrackInstances(frame_height=1.0, frame_width=1.0, hidden_dim=C).to(device=device)
        rn = lambda *s: torch.randn(s, dtype=torch.float, device=device)  # noqa: E731
        f.query_embed = rn(1, C if self.use_dab else 2 * C)
        f.output_embed = rn(1, C)
        f.ref_pts = rn(1, 4)
        f.ids = torch.as_tensor([-2], dtype=torch.long, device=device)
        f.matched_idx = torch.as_tensor([-2], dtype=torch.long, device=device)
        f.boxes = rn(1, 4)
        f.logits = rn(1, n_logits)
        f.iou = torch.zeros((1,), dtype=torch.float, device=device)
        f.last_output = rn(1, C)
        f.long_memory = rn(1, C)
        return f

    def select_active_tracks(self, previous_tracks, new_tracks, unmatched_dets, no_augment: bool = False):
        cat = TrackInstances.cat_tracked_instances
        if not self.training:
            assert len(previous_tracks) == 1 and len(new_tracks) == 1     # eval runs one sequence at a time
            self._seed_memories(new_tracks[0])
            active = cat(previous_tracks[0], new_tracks[0])
            return [active[active.ids >= 0]]
        tracks = []
        for b in range(len(new_tracks)):
            self._seed_memories(new_tracks[b])
            self._seed_memories(unmatched_dets[b])
            if self.tp_drop_ratio == 0.0 and self.fp_insert_ratio == 0.0:
                # (float fields packed into one tensor: the selection below and the embedding update then move all of
                #  them with one gather / read them in place -- structures/track_instances.py: cat_packed)
                active = (TrackInstances.cat_packed if PACKED_TRACKS else cat)(previous_tracks[b], new_tracks[b],
                                                                               unmatched_dets[b])
                keep_rows, keep_thr = unmatched_dets[b].__dict__.pop("_keep_rows", None) or (None, None)
                if (keep_rows is not None and self.__dict__.get("_keep_rows_ok", False)
                        and keep_thr == float(self.update_threshold)):
                    # the criterion already knows which rows pass (models/criterion.py: finish_tracks, from flags that
                    # travelled to the host with the matching costs): an index, not a boolean mask -- no nonzero(), no
                    # second stream synchronisation per frame
                    active = active[keep_rows]
                else:
                    scores = torch.max(logits_to_scores(active.logits), dim=1).values
                    active = active[(scores > self.update_threshold) | (active.ids >= 0)]
                active.ids = torch.where(active.iou < 0.5, torch.full_like(active.ids, -1), active.ids)
            else:
                active = cat(previous_tracks[b], new_tracks[b])
                active = active[(active.iou > 0.5) & (active.ids >= 0)]
                if self.tp_drop_ratio > 0.0 and not no_augment and len(active) > 0:
                    active = active[torch.rand((len(active),)) > self.tp_drop_ratio]
                if self.fp_insert_ratio > 0.0 and not no_augment:
                    picked = active[torch.bernoulli(torch.ones((len(active),)) * self.fp_insert_ratio).bool()]
                    if len(unmatched_dets[b]) > 0 and len(picked) > 0:
                        if len(picked) >= len(unmatched_dets[b]):
                            fp = unmatched_dets[b]
                        else:       # the unmatched detection overlapping each picked track the most
                            iou, _ = box_iou_union(box_cxcywh_to_xyxy(unmatched_dets[b].boxes),
                                                   box_cxcywh_to_xyxy(picked.boxes))
                            fp = unmatched_dets[b][torch.unique(torch.max(iou, dim=0).indices)]
                        active = cat(active, fp)
            if len(active) == 0:
                active = self._fake_track(active.logits.shape[1])
            tracks.append(active)
        return tracks


def build(config: dict) -> QueryUpdater:
    return QueryUpdater(
        hidden_dim=config["HIDDEN_DIM"], ffn_dim=config["FFN_DIM"], dropout=config["DROPOUT"],
        tp_drop_ratio=config.get("TP_DROP_RATE", 0.0), fp_insert_ratio=config.get("FP_INSERT_RATE", 0.0),
        use_checkpoint=config["USE_CHECKPOINT"], use_dab=config["USE_DAB"],
        update_threshold=config["UPDATE_THRESH"], long_memory_lambda=config["LONG_MEMORY_LAMBDA"],
        visualize=config["VISUALIZE"])
