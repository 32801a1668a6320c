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


def _empty(shape, dtype=torch.float, device=None):
    """Shared zero-row tensors for the default fields (CPU, or ``device``): the container is rebuilt dozens of times per
    frame and every field is overwritten right away; a tensor with no elements cannot be modified, so sharing is safe."""
    key = (shape, dtype) if device is None else (shape, dtype, str(device))
    t = _EMPTY.get(key)
    if t is None:
        t = _EMPTY[key] = torch.zeros(shape, dtype=dtype, device=device)
    return t


class TrackInstances:
    def __init__(self, frame_height: float = 1.0, frame_width: float = 1.0, hidden_dim: int = 256,
                 num_classes: int = 1, use_dab: bool = False, device=None):
        """``device``: where the (empty) default fields live -- ``TrackInstances(...).to(dev)`` without the fifteen
        per-field moves (the criterion builds two of these per frame behind the assignment wait)."""
        self.use_dab = use_dab
        self.frame_height = frame_height
        self.frame_width = frame_width
        self.hidden_dim = hidden_dim
        self.num_classes = num_classes
        d = device
        self.ref_pts = _empty((0, 4), device=d)
        self.query_embed = _empty((0, hidden_dim if use_dab else 2 * hidden_dim), device=d)
        self.ids = _empty((0,), torch.long, d)
        self.boxes = _empty((0, 4), device=d)
        self.labels = _empty((0,), torch.long, d)
        self.logits = _empty((0, num_classes), device=d)
        self.matched_idx = _empty((0,), torch.long, d)
        self.output_embed = _empty((0, hidden_dim), device=d)
        self.disappear_time = _empty((0,), torch.long, d)
        self.scores = _empty((0,), device=d)
        self.area = _empty((0,), device=d)
        self.iou = _empty((0,), device=d)
        self.last_output = _empty((0, hidden_dim), device=d)
        self.long_memory = _empty((0, hidden_dim), device=d)
        self.last_appear_boxes = _empty((0, 4), device=d)

    def on_device(self, device) -> bool:
        """Every tensor field already lives on ``device`` (then ``to(device)`` would only rebuild the container)."""
        device = torch.device(device)
        return all((v.device.type == device.type and (device.index is None or v.device.index == device.index))
                   for v in vars(self).values() if type(v) is torch.Tensor)

    def _blank_like(self) -> "TrackInstances":
        return TrackInstances(frame_height=self.frame_height, frame_width=self.frame_width,
                              hidden_dim=self.hidden_dim, num_classes=self.num_classes)

    def to(self, device) -> "TrackInstances":
        res = self._blank_like()
        for k, v in vars(self).items():
            setattr(res, k, v.to(device) if hasattr(v, "to") else v)     # (_packed: kept, checked by identity when used)
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
        packed = self._packed_base() if rows else None
        if packed is not None:
            # the float fields are column views of one tensor (cat_packed): ONE gather for all of them, and one
            # index_add instead of one per field on the way back
            base, names, widths = packed
            idx = item if item.device == base.device else item.to(base.device)
            res._set_packed(base.index_select(0, idx), names, widths)
        for k, v in vars(self).items():
            if k.startswith("_") or (packed is not None and k in packed[1]):
                continue
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

    # ------------------------------------------------------------------ float fields as columns of one tensor
    PACKED_FLOAT = ("logits", "boxes", "ref_pts", "output_embed", "long_memory", "last_output", "query_embed", "iou")

    def _set_packed(self, base: torch.Tensor, names, widths):
        views = base.split(list(widths), dim=1)
        for name, v in zip(names, views):
            setattr(self, name, v.squeeze(1) if name == "iou" else v)
        self._packed = (base, tuple(names), tuple(widths), tuple(getattr(self, n) for n in names))

    def _packed_base(self):
        """(base, names, widths) while every packed field still IS the view made by ``_set_packed`` (assigning a field
        replaces the attribute, which ends the arrangement for good), else None."""
        p = self.__dict__.get("_packed")
        if p is None:
            return None
        base, names, widths, views = p
        if all(getattr(self, n) is v for n, v in zip(names, views)):
            return base, names, widths
        del self.__dict__["_packed"]
        return None

    @staticmethod
    def cat_packed(*parts: "TrackInstances") -> "TrackInstances":
        """``cat_tracked_instances`` with the float fields of ``PACKED_FLOAT`` gathered into ONE (n, W) tensor whose
        column views become the fields: a following ``[index]`` then costs one gather (and one index_add backward)
        for all eight instead of one each, and the query updater's captured update reads the packed rows as they are
        (models/updater_graphs.py).  Same values as the per-field concatenation; falls back to it when the parts do
        not carry those fields as float32 rows on one device."""
        names = TrackInstances.PACKED_FLOAT
        live = [t for t in parts if len(t) > 0]

        def packable(t):
            n = len(t)
            dev = t.query_embed.device
            return all(torch.is_tensor(getattr(t, k)) and getattr(t, k).dtype == torch.float32
                       and getattr(t, k).shape[0] == n and getattr(t, k).device == dev
                       and getattr(t, k).dim() == (1 if k == "iou" else 2) for k in names)

        if not live or not all(packable(t) for t in live) or len({t.query_embed.device for t in live}) != 1:
            return TrackInstances.cat_tracked_instances(*parts) if len(parts) > 1 else parts[0]
        widths = [1 if k == "iou" else getattr(live[0], k).shape[1] for k in names]
        if any([1 if k == "iou" else getattr(t, k).shape[1] for k in names] != widths for t in live):
            return TrackInstances.cat_tracked_instances(*parts) if len(parts) > 1 else parts[0]
        blocks = [torch.cat([getattr(t, k)[:, None] if k == "iou" else getattr(t, k) for k in names], dim=1) for t in live]
        base = blocks[0] if len(blocks) == 1 else torch.cat(blocks, dim=0)
        res = TrackInstances(frame_height=parts[0].frame_height, frame_width=parts[0].frame_width)
        for k, v in vars(parts[0]).items():
            if type(v) is torch.Tensor and k not in names:
                vals = tuple(getattr(t, k) for t in parts)
                if all(x.shape[0] == 0 for x in vals) and all(x.device == v.device and x.shape == v.shape for x in vals):
                    setattr(res, k, v)              # nobody holds a row of it: the concatenation of empties is an empty
                else:
                    setattr(res, k, torch.cat(vals))
        res._set_packed(base, names, widths)
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
