"""Forward-only hipGraphs for ``torch.no_grad()`` inference (the online tracking loop, reference
submit_engine.py:58-120): the encode half of a frame and the decoder loop, each replayed with one launch.

Why (round 3): one 800 x 1333 frame is 685 device activities, 11.2 ms of kernels in a 22-31 ms frame -- the host
issuing launches is two thirds of it (tools/infer_gaps.py).  Training captures forward AND backward through
``make_graphed_callables`` (models/decoder_graphs.py, models/encode_graphs.py); without autograd a plain
``torch.cuda.CUDAGraph`` over static input copies is enough, and the live parameters can be read in place (their
storage does not move between frames; a fingerprint of the data pointers re-captures when it does).

Rules kept from the training captures: thread-local capture mode, the memset-node check of this runtime
(``decoder_graphs.checked_capture``), every tensor the capture read through a raw pointer pinned by the entry, a cache
that stops capturing when keys never recur.  Outputs are static buffers of the graph: what outlives the frame (decoder
stacks: the tracks keep slices of them; ``memory`` unless the caller alternates slots itself) is cloned after the replay.
"""
from __future__ import annotations

import os

import torch

from ..functions import clip_ops
from ..utils.nested_tensor import NestedTensor
from .decoder_graphs import DecoderLoop, checked_capture
from .graph_cache import MISS_LIMIT, RETRY_AFTER, GraphCache, require_graphs, selector_signature  # noqa: F401 (re-exported for the tests)

MAX_GRAPHS = 8


def enabled() -> bool:
    return os.environ.get("MEMOTR_INFER_GRAPHS", "1") != "0" and os.environ.get("MEMOTR_DECODER_GRAPHS", "1") != "0"


class ForwardGraphs(GraphCache):
    """LRU cache of forward-only captures: ``run(key, make_fn, inputs)``."""

    def __init__(self, what: str):
        super().__init__(what + " inference", MAX_GRAPHS)
        # a capture stream of its own: the BLAS workspaces torch hands out are per (handle, stream), and a capture
        # bakes the pointer in -- two caches whose graphs may replay concurrently (the next frame's encode on a side
        # stream next to this frame's decoder loop) must not share one
        self._capture_stream = None

    def run(self, key, make_fn, inputs, pins=()):
        """Outputs of ``make_fn()(*inputs)`` through the graph stored under ``key`` (captured on first use; ``make_fn``
        builds the function to capture and is only called then).  None -> the caller runs eagerly."""
        entry = self.lookup(key, lambda: self._capture(make_fn(), inputs, pins))
        if entry is None:
            return None
        graph, static_in, static_out = entry[:3]
        for dst, src in zip(static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        graph.replay()
        self.replays += 1
        return static_out

    def _capture(self, fn, inputs, pins):
        static_in = tuple(t.detach().clone() for t in inputs)
        try:
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):                 # warm-up: library handles, find results, geometry caches
                    fn(*static_in)
            cur.wait_stream(side)

            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream()

            def make():
                g = torch.cuda.CUDAGraph()
                # (thread-local error mode: checked_capture patches the context class)
                with torch.cuda.graph(g, stream=self._capture_stream):
                    out = fn(*static_in)
                return g, out

            graph, static_out = checked_capture(make)
        except Exception as exc:  # noqa: BLE001 -- capture is an optimisation; eager stays valid
            return self.capture_failed(exc)
        self.captures += 1
        return graph, static_in, static_out, (fn, tuple(pins))


def _fingerprint(module) -> int:
    """Changes when the parameters / buffers of ``module`` move to other storage (``.to()``, ``.half()``, a replaced
    Parameter): the captures read them in place.  In-place updates (``load_state_dict``, an optimiser step) keep it
    -- the captured kernels then read the new values through the same pointers.  (Constants DERIVED from buffers, i.e.
    the folded batch-norm scale / shift of the backbone, are not covered by this: the encode key carries the buffer
    versions for them.)  The walk over the module tree costs ~3 ms for the full model, so it is redone every 256 calls;
    the first and the last parameter are looked at on every call."""
    cache = module.__dict__.setdefault("_infer_fingerprint", [0, None, None, None])
    first = next(module.parameters(), None)
    last = cache[3]() if cache[3] is not None else None
    quick = (None if first is None else (first.data_ptr(), first.dtype),
             None if last is None else (last.data_ptr(), last.dtype))
    cache[0] -= 1
    if cache[0] <= 0 or cache[2] != quick or (cache[3] is not None and last is None):
        import weakref
        h = 0
        tensors = list(module.parameters()) + list(module.buffers())
        for t in tensors:
            h = (h * 1000003 + t.data_ptr()) & 0xFFFFFFFFFFFF
        params = list(module.parameters())
        last = params[-1] if params else None
        cache[3] = weakref.ref(last) if last is not None else None
        quick = (quick[0], None if last is None else (last.data_ptr(), last.dtype))
        cache[0], cache[1], cache[2] = 256, h, quick
    return cache[1]


class InferGraphs:
    """Owned by a ``MeMOTR``: the encode half and the decoder loop of no-grad calls."""

    def __init__(self, core):
        self.core = core
        self.encode = ForwardGraphs("encode")
        self.decode = ForwardGraphs("decoder")

    # ------------------------------------------------------------------ encode half
    def encode_usable(self, frame: NestedTensor) -> bool:
        return (enabled() and not torch.is_grad_enabled() and not self.core.training and frame is not None
                and frame.tensors.is_cuda and frame.masks is not None and getattr(frame, "sizes", None) is not None
                and not torch.is_autocast_enabled() and frame.tensors.dtype == torch.float32 and not self.encode.failed)

    def run_encode(self, frame: NestedTensor):
        core = self.core
        masks, geometry = frame.masks, frame.sizes
        # `encode_slot`: a caller that queues the next frame's encode while this frame's `memory` is still being read
        # (inference.SequenceTracker) alternates between two captures, each with its own static output
        # (the folded batch-norm constants are baked in: an in-place write to a buffer -- a checkpoint load -- must not
        # replay the old ones; models/encode_graphs.py keys on the same)
        bufver = sum(b._version for b in core.backbone.buffers())
        # the selector's signature: a capture bakes the kernel choice of the encoder's self-attention calls in; replayed
        # launches keep counting the points that leave their windows, and when the share asks for another kernel the
        # signature moves and the graph captured (or to be captured) at the new levels takes over (msda_select.h)
        key = (getattr(frame, "encode_slot", 0), tuple(frame.tensors.shape), geometry, clip_ops.config_key(),
               _fingerprint(core), bufver, selector_signature(self, core.transformer.encoder))
        constants = {}

        def make_fn():
            def fn(images):
                enc = core._encode_frame_eager(NestedTensor(images, masks, geometry))
                if not constants:       # (first warm-up call: the geometry caches' own tensors, kept by the entry)
                    constants.update({k: v for k, v in enc.items() if k != "memory"})
                return enc["memory"]
            fn.constants = constants
            return fn

        entry_known = key in self.encode.slots
        memory = self.encode.run(key, make_fn, (frame.tensors,), pins=(masks,))
        if memory is None:
            return None
        if not entry_known:      # what depends on the masks alone travels with the entry
            tr = core.transformer
            extra = [dict(tr.__dict__.get("_mask_derived", {})), dict(tr.__dict__.get("_pyramids", {}))]
            for m in core.modules():
                for attr in ("_cache", "_folded"):
                    v = m.__dict__.get(attr)
                    if v is not None:
                        extra.append(dict(v) if isinstance(v, dict) else v)
            e = self.encode.slots[key]
            self.encode.slots[key] = e[:3] + (e[3] + (dict(constants), extra),)
        consts = self.encode.slots[key][3][2]
        # `memory` is the graph's static output: the next replay of this (slot, shape) overwrites it.  A caller that
        # alternates slots itself (inference.SequenceTracker sets `encode_static_ok`) reads it in place; everyone else
        # gets a copy (1.4 % of a frame's traffic), so two encode results can be held at once
        if not getattr(frame, "encode_static_ok", False):
            memory = memory.clone()
        return dict(consts, memory=memory)

    # ------------------------------------------------------------------ decoder loop
    def decode_usable(self, decoder, output, src) -> bool:
        return (enabled() and not torch.is_grad_enabled() and not decoder.training and output.is_cuda
                and decoder.use_dab and decoder.bbox_embed is not None and not torch.is_autocast_enabled()
                and output.dtype == torch.float32 and src.dtype == torch.float32 and not self.decode.failed
                and not any(getattr(layer, "extra_track_attn", False) for layer in decoder.layers))

    def run_decode(self, decoder, args, shapes, lsi):
        from .decoder_graphs import DecoderGraphs
        key = (tuple(a.shape for a in args), DecoderGraphs._geometry(shapes), clip_ops.config_key(),
               _fingerprint(decoder))

        def make_fn():
            loop = DecoderLoop(decoder, shapes, lsi).eval()
            return lambda *xs: loop(*xs)

        out = self.decode.run(key, make_fn, args, pins=(shapes, lsi))
        if out is None:
            return None
        return tuple(t.clone() for t in out)      # the tracks keep slices of the stacks beyond the next replay
