"""Per-clip track state handed between frames (contract of structures/track_instances.py:7-129).

A bag of per-track tensors.  The quirks of the reference container that callers can observe are kept:
``to`` / ``__getitem__`` / ``cat_tracked_instances`` rebuild the object with *default* meta
(``use_dab=False``; ``cat`` also resets ``hidden_dim``/``num_classes``) and then overwrite every
attribute, so only tensor fields are reliable after the first query update.
"""
from __future__ import annotations

from typing import List

import torch

_TENSOR_FIELDS = (
    "ref_pts", "query_embed", "ids", "boxes", "labels", "logits", "matched_idx", "output_embed",
    "disappear_time", "scores", "area", "iou", "last_output", "long_memory", "last_appear_boxes",
)


_EMPTY = {}


def _empty(shape, dtype=torch.float):
    """Shared zero-row CPU tensors for the default fields: the container is rebuilt dozens of times per frame and
    every field is overwritten right away; a tensor with no elements cannot be modified, so sharing is safe."""
    key = (shape, dtype)
    t = _EMPTY.get(key)
    if t is None:
        t = _EMPTY[key] = torch.zeros(shape, dtype=dtype)
    return t


class TrackInstances:
    def __init__(self, frame_height: float = 1.0, frame_width: float = 1.0, hidden_dim: int = 256,
                 num_classes: int = 1, use_dab: bool = False):
        self.use_dab = use_dab
        self.frame_height = frame_height
        self.frame_width = frame_width
        self.hidden_dim = hidden_dim
        self.num_classes = num_classes
        self.ref_pts = _empty((0, 4))
        self.query_embed = _empty((0, hidden_dim if use_dab else 2 * hidden_dim))
        self.ids = _empty((0,), torch.long)
        self.boxes = _empty((0, 4))
        self.labels = _empty((0,), torch.long)
        self.logits = _empty((0, num_classes))
        self.matched_idx = _empty((0,), torch.long)
        self.output_embed = _empty((0, hidden_dim))
        self.disappear_time = _empty((0,), torch.long)
        self.scores = _empty((0,))
        self.area = _empty((0,))
        self.iou = _empty((0,))
        self.last_output = _empty((0, hidden_dim))
        self.long_memory = _empty((0, hidden_dim))
        self.last_appear_boxes = _empty((0, 4))

    def _blank_like(self) -> "TrackInstances":
        return TrackInstances(frame_height=self.frame_height, frame_width=self.frame_width,
                              hidden_dim=self.hidden_dim, num_classes=self.num_classes)

    def to(self, device) -> "TrackInstances":
        res = self._blank_like()
        for k, v in vars(self).items():
            setattr(res, k, v.to(device) if hasattr(v, "to") else v)
        return res

    def __len__(self) -> int:
        assert self.ref_pts.shape[0] == self.query_embed.shape[0]
        return max(self.query_embed.shape[0], self.labels.shape[0])

    def __getitem__(self, item) -> "TrackInstances":
        if type(item) == int:
            n = len(self)
            if item >= n or item < -n:
                raise IndexError("TrackInstances index out of range!")
            item = slice(item, None, n)
        if torch.is_tensor(item) and item.dtype == torch.bool:
            # one nonzero (one device sync) for all fields instead of one per boolean-indexed field
            item = item.nonzero().squeeze(1)
        # a 1-d index tensor gathers through index_select: its backward is one index_add, where advanced indexing
        # differentiates through a sort-based accumulate (~7 kernels per field that carries a gradient)
        rows = torch.is_tensor(item) and item.dim() == 1 and item.dtype in (torch.int64, torch.int32)
        res = self._blank_like()
        for k, v in vars(self).items():
            if hasattr(v, "__getitem__") and v.shape[0] != 0:
                if rows and torch.is_tensor(v):
                    # index_select wants the index on the field's device (advanced indexing did not: the query
                    # updater's drop / insert masks are built on the CPU, reference models/query_updater.py:148-150)
                    idx = item if item.device == v.device else item.to(v.device)
                    setattr(res, k, v.index_select(0, idx))
                else:
                    setattr(res, k, v[item])
            else:
                setattr(res, k, v)
        return res

    @staticmethod
    def init_tracks(batch: dict, hidden_dim: int, num_classes: int, device="cpu", use_dab: bool = False):
        """One empty TrackInstances per clip of the batch (``batch["imgs"][b][t]`` is a (3,H,W) frame)."""
        first_frames = [clip[0] for clip in batch["imgs"]]
        h_max = max(f.shape[-2] for f in first_frames)
        w_max = max(f.shape[-1] for f in first_frames)
        return [
            TrackInstances(frame_height=float(f.shape[-2] / h_max), frame_width=float(f.shape[-1] / w_max),
                           hidden_dim=hidden_dim, num_classes=num_classes, use_dab=use_dab).to(device)
            for f in first_frames
        ]

    @staticmethod
    def cat_tracked_instances(tracked1: "TrackInstances", tracked2: "TrackInstances",
                              *more: "TrackInstances") -> "TrackInstances":
        """Concatenation of two (the reference's signature) or more instance lists, one ``cat`` per field."""
        res = TrackInstances(frame_height=tracked1.frame_height, frame_width=tracked1.frame_width)
        for k, v in vars(tracked1).items():
            if type(v) is torch.Tensor:
                setattr(res, k, torch.cat((v, getattr(tracked2, k)) + tuple(getattr(t, k) for t in more)))
        return res

    @staticmethod
    def tracks_to_meta_tensors(tracks: List["TrackInstances"]):
        keys = [k for k in vars(tracks[0]) if type(getattr(tracks[0], k)) is torch.Tensor]
        meta = {"frame_height": [], "frame_width": [], "hidden_dim": [], "num_classes": [], "keys": keys}
        tensors = []
        for t in tracks:
            for name in ("frame_height", "frame_width", "hidden_dim", "num_classes"):
                meta[name].append(getattr(t, name))
            tensors.extend(getattr(t, k) for k in keys)
        return meta, tensors

    @staticmethod
    def meta_tensors_to_tracks(meta: dict, tensors: List[torch.Tensor]):
        n_keys = len(meta["keys"])
        tracks = []
        for b in range(len(meta["frame_height"])):
            t = TrackInstances(frame_height=meta["frame_height"][b], frame_width=meta["frame_width"][b],
                               hidden_dim=meta["hidden_dim"][b], num_classes=meta["num_classes"][b])
            for i, k in enumerate(meta["keys"]):
                setattr(t, k, tensors[b * n_keys + i])
            tracks.append(t)
        return tracks
