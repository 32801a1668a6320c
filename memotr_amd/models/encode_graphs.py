"""hipGraph capture of the query-independent half of a frame: backbone -> feature projections -> encoder, forward AND
backward, one graph pair per (call slot, batch shape, image geometry, precision).

Why (round 3): with the decoder loop in graphs the fp32 step is GPU-bound, but the bf16 extension (BASELINE config 5)
is not -- its kernels take ~100 ms per step while autocast's casts push the launch count to ~10 k and the host needs
~200 ms to issue them.  The encode half of a clip is ONE batched call on a fixed geometry (``engine.encode_chunks``),
~2-5 k launches forward + backward: captured, it costs the host two launches.

Same rules as the decoder capture (models/decoder_graphs.py): parameters travel as one flat tensor argument
(``torch.func.functional_call``; a captured backward must not see live ``nn.Parameter`` accumulators), captures run
in thread-local error mode, a failed capture falls back to eager unless MEMOTR_REQUIRE_GRAPHS=1, every new geometry
costs a capture so the cache stops capturing when geometries never recur.  Under autocast the capture runs with the
cast cache off (a cached cast made during the capture would be a dangling pointer on replay).

What is captured is ``MeMOTR.encode_frame`` itself; what depends on the masks alone (flattened masks, valid ratios,
pyramid tensors) comes back from the warm-up call as constants, the graph's only output is ``memory``.
"""
from __future__ import annotations

import os
import torch

from ..functions import clip_ops
from ..utils.nested_tensor import NestedTensor
from .decoder_graphs import checked_capture
from .graph_cache import GraphCache, selector_signature

MAX_GRAPHS = 6           # each holds the activations of a whole batched encode


def enabled() -> bool:
    """MEMOTR_ENCODE_GRAPHS: "1" on, "0" off, default "auto" = under autocast only (the fp32 step is GPU-bound: the
    capture would buy idle gaps at the price of a second copy of the encode activations)."""
    v = os.environ.get("MEMOTR_ENCODE_GRAPHS", "auto")
    if v == "auto":
        return torch.is_autocast_enabled()
    return v != "0"


class EncodeGraphs(GraphCache):
    """Cache of captured encode calls, owned by a ``MeMOTR``."""

    def __init__(self, core):
        super().__init__("encode", MAX_GRAPHS)
        self.core = core

    # ------------------------------------------------------------------ eligibility
    def usable(self, frame: NestedTensor) -> bool:
        c = self.core
        return (enabled() and os.environ.get("MEMOTR_DECODER_GRAPHS", "1") != "0" and not self.failed
                and frame is not None and frame.tensors.is_cuda and frame.masks is not None
                and getattr(frame, "sizes", None) is not None and torch.is_grad_enabled() and c.training
                and not c.use_checkpoint)

    def _names(self):
        """Parameters the encode half reads: backbone, projections, encoder, level embedding."""
        pre = ("backbone.", "feature_projs.", "transformer.encoder.", "transformer.level_embed")
        return [(n, p) for n, p in self.core.named_parameters() if n.startswith(pre)]

    # ------------------------------------------------------------------ one call
    def run(self, frame: NestedTensor, slot: int):
        """``core.encode_frame(frame)`` through the graph of call slot ``slot`` (the index of the encode call inside
        its clip: a graphed callable owns its activations until its backward has run); None -> caller runs eager."""
        amp = (torch.is_autocast_enabled(), str(torch.get_autocast_dtype("cuda")) if torch.is_autocast_enabled() else "")
        # (the folded batch-norm constants are baked in: an in-place write to a buffer -- a checkpoint load -- must
        # not replay the old ones)
        bufver = sum(b._version for b in self.core.backbone.buffers())
        # (a capture bakes the kernel choice of the encoder's self-attention calls in: the selector's signature moves
        # when the measured off-window share asks for another kernel -- replayed launches keep counting, msda_select.h)
        key = (slot, tuple(frame.tensors.shape), frame.sizes, amp, clip_ops.config_key(), bufver,
               selector_signature(self, self.core.transformer.encoder))
        entry = self.lookup(key, lambda: self._capture(frame, amp))
        if entry is None:
            return None
        fn, params, constants, state = entry[:4]
        if state["busy"]:
            # this slot's activations are still waiting for their backward (a second encode call with the same slot
            # inside one clip): a replay would overwrite them -- this call runs eagerly
            self.eager += 1
            return None
        self.replays += 1
        flat = torch.cat([p.reshape(-1) for p in params])
        memory = fn(frame.tensors, flat)
        state["busy"] = True

        def _released(grad, state=state):
            state["busy"] = False        # the slot's backward is being queued: stream order protects the replay
            return grad

        memory.register_hook(_released)
        return dict(constants, memory=memory)

    def _capture(self, frame: NestedTensor, amp):
        core = self.core
        named = self._names()
        names = [n for n, _ in named]
        params = tuple(p for _, p in named)
        sizes = [p.numel() for p in params]
        views = [p.shape for p in params]
        masks, geometry = frame.masks, frame.sizes
        amp_on, _ = amp
        amp_dtype = torch.get_autocast_dtype("cuda") if amp_on else None
        constants = {}

        def run(images, flat):
            pieces = flat.split(sizes)
            sub = {n: w.view(s) for n, w, s in zip(names, pieces, views)}
            nested = NestedTensor(images, masks, geometry)
            if amp_on:       # the ambient context does not reach a replay: the graph carries its own
                with torch.autocast("cuda", dtype=amp_dtype, cache_enabled=False):
                    enc = torch.func.functional_call(core, sub, (), {"frame": nested, "stage": "encode_eager"})
            else:
                with torch.autocast("cuda", enabled=False):
                    enc = torch.func.functional_call(core, sub, (), {"frame": nested, "stage": "encode_eager"})
            if not constants:
                constants.update({k: v for k, v in enc.items() if k != "memory"})
            return enc["memory"]

        with torch.no_grad():
            flat = torch.cat([p.reshape(-1) for p in params])
        sample = (frame.tensors.detach().clone(), flat.requires_grad_(True))
        try:
            # (make_graphed_callables refuses an ambient autocast with its cast cache on; `run` opens its own)
            with torch.autocast("cuda", enabled=False):
                fn = checked_capture(lambda: torch.cuda.make_graphed_callables(run, sample, num_warmup_iters=2,
                                                                               allow_unused_input=True))
        except Exception as exc:  # noqa: BLE001 -- capture is an optimisation; eager stays valid
            return self.capture_failed(exc)
        live = dict(core.named_parameters())
        assert all(live[n] is p for n, p in zip(names, params)), "encode parameters were replaced by the capture"
        # `constants` were taken from the FIRST (eager, warm-up) call: the geometry caches' own tensors -- ordinary
        # allocations that carry the host tag of the pyramid, not memory of the graph's pool
        self.captures += 1
        return fn, params, dict(constants), {"busy": False}, self._pins(run, frame)

    def _pins(self, run, frame):
        """Everything the captured kernels read through a baked pointer that is NOT in the graph's own pool: the
        capture's masks (`run` closes over them; the callable torch returns does not keep `run` alive, and the engine
        builds a new NestedTensor every step -- found the hard way: replay 1 read a recycled mask), and the tensors the
        geometry caches handed out during the capture (a cache may evict; the graph may not notice)."""
        core = self.core
        pins = [run, frame.masks]
        tr = core.transformer
        pins += [dict(tr.__dict__.get("_mask_derived", {})), dict(tr.__dict__.get("_pyramids", {}))]
        for m in core.modules():
            for attr in ("_cache", "_folded"):
                v = m.__dict__.get(attr)
                if v is not None:
                    pins.append(dict(v) if isinstance(v, dict) else v)
        return pins
