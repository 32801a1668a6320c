"""hipGraph capture of the query updater's embedding update (forward AND backward), one graph pair per (frame slot,
clip of the batch, row bucket).

Why (round 6, tools/small_trace.py on the train step): between the decoder graphs of two frames the step is bound by
the host.  The update of reference models/query_updater.py:96-158 -- three MLPs, one memory-attention layer, two FFNs,
the selects -- is ~55 launches forward (16 us of host time each, from python) and ~130 backward (9 us each, from the
autograd engine, a third of them the fan-in adds of parameter gradients that meet across the frames of a clip), with the
GPU idle two thirds of that time.  Replayed from a graph the same kernels run back to back.

What is captured: ``QueryUpdater.update_fields`` on ONE packed tensor [logits | boxes | ref_pts | output_embed |
long_memory | last_output | query_embed] of the active tracks, rows padded with zeros to a multiple of ``BUCKET``
(padded rows are excluded from the memory attention as keys -- the only place rows meet -- and are sliced away after the
call, so they receive and produce zero gradient), returning the packed [ref_pts | long_memory | last_output |
query_embed].  Parameters enter as one flat tensor made once per clip, as in models/decoder_graphs.py: the frames'
parameter gradients meet in one add per frame and DistributedDataParallel's hooks fire once per parameter.
Anything that cannot be captured (CPU tensors, autocast, dropout in training mode, a capture error) runs eagerly.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..functions import clip_ops
from .decoder_graphs import checked_capture
from .graph_cache import GraphCache

BUCKET = 16
# (frame slots x clips of the batch x row buckets) kept alive; least recently used go first -- small graphs (a few MB each);
# sized so that the working set of a training run (5 slots x up to a dozen buckets of 16 rows) fits
MAX_GRAPHS = int(os.environ.get("MEMOTR_MAX_UPDATER_GRAPHS", "64"))


def enabled() -> bool:
    return os.environ.get("MEMOTR_UPDATER_GRAPHS", "1") != "0"


class UpdateStep(nn.Module):
    """``update_fields`` on packed tensors.  Holds the updater as its only child and is never attached to the model
    tree (every parameter appears once: see ``DecoderLoop``)."""

    def __init__(self, updater, widths):
        super().__init__()
        self.updater = updater
        self.widths = list(widths)

    def forward(self, packed, key_mask):
        fields = list(packed.split(self.widths, dim=1))
        fields[3] = fields[3].contiguous()       # output_embed feeds the row-linear / LayerNorm kernels (contiguous rows)
        return torch.cat(self.updater.update_fields(*fields, key_mask=key_mask), dim=1)


class UpdaterGraphs(GraphCache):
    """Cache of captured embedding updates, owned by a ``QueryUpdater``."""

    def __init__(self, updater):
        super().__init__("query updater", MAX_GRAPHS, grow_cap=4)
        self.updater = updater

    def usable(self, fields) -> bool:
        u = self.updater
        return (enabled() and not self.failed and torch.is_grad_enabled() and not torch.is_autocast_enabled()
                and all(f.is_cuda and f.dtype == torch.float32 and f.dim() == 2 for f in fields)
                and any(f.requires_grad for f in fields) and len(fields[0]) > 0
                and not (u.training and u.dropout > 0) and clip_ops.fused(fields[3]))

    def run(self, slot, fields, clip_key=None, packed=None):
        """The update of the track set ``fields`` (the tensors of ``QueryUpdater.FIELDS``) through the graph of ``slot``
        (captured on first use).  Returns the four new fields, or None if the capture failed.  ``packed``: a tensor whose
        leading columns ARE the fields, in order (``TrackInstances.cat_packed``): no concatenation then."""
        n = fields[0].shape[0]
        rows = (n + BUCKET - 1) // BUCKET * BUCKET
        widths = tuple(f.shape[1] for f in fields)
        key = (slot, rows, widths, clip_ops.config_key())
        entry = self.lookup(key, lambda: self._capture(rows, widths, fields[0].device))
        if entry is None:
            return None
        fn, params, mask_for = entry
        if packed is not None and packed.shape[0] == n and packed.shape[1] >= sum(widths):
            packed = packed[:, :sum(widths)]
        else:
            packed = torch.cat(fields, dim=1)                      # one launch; its backward hands out views
        if rows > n:
            packed = F.pad(packed, (0, 0, 0, rows - n))
        self.replays += 1
        out = fn(packed, mask_for(n), self._flat_parameters(params, clip_key))
        C = self.updater.hidden_dim
        return out[:n].split([4, C, C, out.shape[1] - 4 - 2 * C], dim=1)

    def _flat_parameters(self, params, clip_key):
        cache = self.__dict__.get("_flat")
        if (clip_key is not None and cache is not None and cache[0] is clip_key and len(cache[1]) == len(params)
                and all(a is b for a, b in zip(cache[1], params))):
            return cache[2]
        flat = torch.cat([p.reshape(-1) for p in params])
        self.__dict__["_flat"] = (clip_key, params, flat)
        return flat

    def _capture(self, rows, widths, device):
        step = UpdateStep(self.updater, widths)
        names, params = zip(*step.named_parameters())
        if len(names) != sum(1 for _ in step.named_parameters(remove_duplicate=False)):
            self.failed = True
            return None
        sizes = [p.numel() for p in params]
        views = [p.shape for p in params]

        def run(packed, key_mask, flat):
            pieces = flat.split(sizes)
            return torch.func.functional_call(step, {n: w.view(s) for n, w, s in zip(names, pieces, views)},
                                              (packed, key_mask))

        with torch.no_grad():
            flat = torch.cat([p.reshape(-1) for p in params])
        g = torch.Generator(device="cpu").manual_seed(0)
        sample = (torch.randn(rows, sum(widths), generator=g).to(device).requires_grad_(True),
                  torch.zeros((1, rows), dtype=torch.bool, device=device), flat.requires_grad_(True))
        try:
            fn = checked_capture(lambda: torch.cuda.make_graphed_callables(run, sample, num_warmup_iters=2,
                                                                           allow_unused_input=True))
        except Exception as exc:  # noqa: BLE001 -- capture is an optimisation; eager stays valid
            return self.capture_failed(exc)
        assert all(isinstance(p, nn.Parameter) for p in step.parameters()) and \
            [id(p) for p in step.parameters()] == [id(p) for p in params], "updater parameters were replaced"
        self.captures += 1
        masks = {}

        def mask_for(n):           # (1, rows) bool, True on the padded slots; one tensor per live count, made once
            m = masks.get(n)
            if m is None:
                m = masks[n] = (torch.arange(rows, device=device) >= n)[None]
            return m

        return fn, params, mask_for
