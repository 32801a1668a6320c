"""hipGraph capture of the decoder loop (forward AND backward), one graph pair per (frame slot, query bucket).

Why: the decoder / box-refinement chain of one frame is ~150 tiny kernels per layer forward and ~2x that backward,
each costing the host 13-27 us to issue; with the encoder batched per clip the train step is bound by that launch
rate.  A captured graph replays the whole per-layer chain with one launch.

What is captured: ``DeformableDecoder.forward``'s loop as a tensor-in / tensor-out module (``DecoderLoop``): per layer anchors -> sine embedding -> ``ref_point_head`` (x ``query_scale``) -> the decoder layer
(self-attention over the queries, MSDeformAttn cross-attention through the fused HIP kernels, FFN) -> box head ->
refined reference (detached for the next layer).  ``torch.cuda.make_graphed_callables`` records its forward and its
backward as two graphs.

Constraints handled here:
  * static shapes: the query count (300 detect + n track queries) is padded to a multiple of ``BUCKET`` with masked
    slots (padded keys are excluded from the self-attention softmax exactly; padded queries are sliced away, so
    their rows receive zero gradient);
  * a graphed callable owns its activations: it cannot run twice before its backward.  A clip holds T frames of
    activations at once, so every frame index gets its own capture ("slot");
  * the pyramid geometry is part of the key (multi-scale training re-captures per geometry, LRU-bounded);
  * parameters are shared by all slots and enter a graph as ONE flat tensor (made once per clip): the frames'
    parameter gradients meet in one add per frame and DistributedDataParallel's hooks fire once per parameter.
Anything that cannot be captured (CPU tensors, checkpointing, no grad mode mismatch, a capture error) takes the eager
path -- same arithmetic, kernel by kernel.
"""
from __future__ import annotations

import contextlib
import os
import torch
import torch.nn as nn

from ..functions import clip_ops
from ..utils.utils import inverse_sigmoid, refine_boxes
from .graph_cache import MISS_LIMIT, RETRY_AFTER, GraphCache, require_graphs  # noqa: F401 (re-exported)
from .utils import pos_to_pos_embed

BUCKET = 32
# (frame slots x geometries x query buckets) kept alive; least recently used go first.  A clip of five frames whose track
# counts wander over five buckets of 32 is already 25 keys -- one more than round 5's limit of 24, and an LRU cache one
# entry short of its working set misses on EVERY lookup (a capture costs two warm-up passes and the capture: ~0.1 s).
# A graph pair holds ~0.25 GB (its static copies of `src`, the flat parameters and the value bank): 48 of them is 12 GB
# of 288.
MAX_GRAPHS = int(os.environ.get("MEMOTR_MAX_DECODER_GRAPHS", "48"))
# (the thrash guard -- MISS_LIMIT / RETRY_AFTER -- and MEMOTR_REQUIRE_GRAPHS live in graph_cache.py)


class DecoderLoop(nn.Module):
    """All iterations of the decoder loop of one frame (what ``DeformableDecoder.forward`` does with DAB anchors and
    box refinement, reference models/deformable_decoder.py:70-140), tensors in -> four stacks out.  One capture per
    frame slot: `src` and the parameters enter the graph once per frame.

    Holds references to modules owned by the decoder and is never attached to the model tree.  Every module appears
    exactly ONCE in this tree: ``torch.func.functional_call`` does not restore parameters of a module that is
    reachable under two names (it leaves the substituted tensors behind -- observed with torch 2.10), so the heads
    shared by all layers (``ref_point_head``, ``query_scale``) live at this level, not inside per-layer children."""

    def __init__(self, decoder, spatial_shapes, level_start_index):
        super().__init__()
        self.layers = nn.ModuleList(decoder.layers)
        self.bbox_embed = nn.ModuleList(decoder.bbox_embed)
        self.ref_point_head = decoder.ref_point_head
        self.query_scale = decoder.query_scale
        self.nd = decoder.n_det_queries
        self.merge_from = decoder.merge_det_track_layer
        self.d_model = decoder.d_model
        # constants of the geometry: closed over, not graph inputs (the operator plans from their host tag)
        self._shapes = spatial_shapes
        self._lsi = level_start_index

    def forward(self, output, reference_points, src, ratios4, query_mask, src_padding_mask):
        outs, refs, layer_inputs, boxes = [], [], [], []
        from ..modules.ms_deform_attn import project_values
        # the six value projections of `src` as one GEMM, each layer reading its columns in place
        values = project_values([layer.cross_attn for layer in self.layers], src, src_padding_mask)
        for lid, layer in enumerate(self.layers):
            layer_inputs.append(output)
            ref_in = reference_points[:, :, None] * ratios4
            anchor = pos_to_pos_embed(ref_in[:, :, 0, :], num_pos_feats=self.d_model // 2)
            raw_pos = self.ref_point_head(anchor)
            query_pos = raw_pos if lid == 0 else self.query_scale(output) * raw_pos
            merge = lid >= self.merge_from
            output = layer(output, query_pos, ref_in, src, self._shapes, self._lsi, query_mask, src_padding_mask, merge,
                           None if values is None else values[lid])
            new_ref = refine_boxes(self.bbox_embed[lid](output), reference_points)
            boxes.append(new_ref)
            if merge:
                reference_points = new_ref.detach()
            else:   # track queries did not go through the layer: keep their anchors
                reference_points = torch.cat((new_ref[:, :self.nd].detach(), reference_points[:, self.nd:]), dim=1)
            outs.append(output)
            refs.append(reference_points)
        return torch.stack(outs), torch.stack(refs), torch.stack(layer_inputs), torch.stack(boxes)


@contextlib.contextmanager
def _thread_local_capture(census=None):
    """``make_graphed_callables`` captures in the "global" error mode: ANY thread that touches the runtime while a
    capture is open (the RCCL watchdog polling its events under DistributedDataParallel, a data-loader thread pinning
    memory) invalidates it.  The decoder capture only needs the capturing threads themselves to behave, so the graph
    context is switched to "thread_local" for its duration.  ``census`` (a list, default: the module's ``CENSUS``)
    receives {node type: count} of every graph captured inside."""
    CENSUS = census if census is not None else globals()["CENSUS"]
    # The patch below replaces process-global names: one capture at a time (re-entrant for the capturing thread), and
    # the originals are read under the lock so that a nested use restores what it found.
    with _PATCH_LOCK:
        orig = torch.cuda.graph
        orig_graph_cls = torch.cuda.CUDAGraph

        class _Graph(orig):
            def __init__(self, *args, **kwargs):
                kwargs.setdefault("capture_error_mode", "thread_local")
                super().__init__(*args, **kwargs)

            def __exit__(self, *exc):
                out = super().__exit__(*exc)
                if exc[0] is None and CENSUS is not None:
                    CENSUS.append(graph_node_census(self.cuda_graph))
                return out

        if CENSUS is not None:       # (the raw hipGraph_t only survives capture_end when asked for)
            class _KeepGraph(orig_graph_cls):      # a subclass: isinstance(x, torch.cuda.CUDAGraph) keeps working
                def __new__(cls, *a, **k):
                    return orig_graph_cls.__new__(cls, keep_graph=True)

                def __init__(self, *a, **k):       # (the binding constructs in __init__, from the CALL's arguments)
                    super().__init__(True)

            torch.cuda.CUDAGraph = _KeepGraph
        torch.cuda.graph = _Graph
        # No cyclic garbage collection while a capture is open: a collection that happens to run then may finalise a
        # hipGraph of an EARLIER capture (a refused one, an evicted cache entry still held by a reference cycle or a
        # traceback), and releasing its memory pool under an open capture aborts the process (seen in round 6: the
        # query updater's capture right after a refused decoder capture).  Garbage is collected before, outside.
        import gc
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            yield
        finally:
            if gc_was_on:
                gc.enable()
            torch.cuda.graph = orig
            torch.cuda.CUDAGraph = orig_graph_cls


_PATCH_LOCK = __import__("threading").RLock()


# MEMOTR_GRAPH_CENSUS=1: every capture appends {node type: count} of its hipGraph here (tools/graph_census.py, the GPU
# tests).  What it is for: on ROCm 7.2 a MEMSET node is not ordered behind the kernels before it when a graph is
# replayed with the runtime's AQL-packet capture on (the default) -- tools/graph_memset_probe.py shows it in ten
# lines, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 restores the order.  torch's multi-block reductions zero their semaphores
# with such a node (bias gradients of captured linears came back as garbage from the second replay on), so nothing
# inside the captured regions may reduce through them: the census is how the tests hold the graphs to ZERO memset
# nodes, whatever the runtime flag says.
CENSUS = [] if os.environ.get("MEMOTR_GRAPH_CENSUS", "0") == "1" else None
_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
               7: "event_record"}


_MEMSET_SAFE = None


def memset_nodes_replay_safe() -> bool:
    """Does THIS process's HIP runtime order a memset node behind the kernels before it when a graph is replayed?
    (tools/graph_memset_probe.py in miniature, run once: kernel dirties a buffer | memset | kernel reads it.)  False
    on ROCm 7.2 unless DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was in the environment when the runtime loaded."""
    global _MEMSET_SAFE
    if _MEMSET_SAFE is None:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        n = 1 << 16
        dev = torch.device("cuda", torch.cuda.current_device())
        buf, x, out = torch.zeros(n, device=dev), torch.ones(n, device=dev), torch.empty(n, device=dev)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            buf.add_(1.0)
            torch.add(buf, x, out=out)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            buf.add_(1.0)
            rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(n * 4),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            torch.add(buf, x, out=out)
            buf.add_(3.0)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        _MEMSET_SAFE = rc == 0 and bool((out == x).all())
    return _MEMSET_SAFE


def checked_capture(make):
    """Run ``make()`` (a ``make_graphed_callables`` call) in thread-local capture mode.  When this runtime does not
    order memset nodes on replay, the captured graphs are inspected and a graph that contains one is refused -- a
    library may zero a workspace that way (the bf16 encode backward at 800 x 1333 holds six such nodes), and replaying
    it would corrupt results silently."""
    census = [] if (CENSUS is None and not memset_nodes_replay_safe()) else None
    with _thread_local_capture(census):
        fn = make()
    bad = [c for c in (census or []) if c.get("memset", 0) or "error" in c]
    if bad:
        raise RuntimeError(f"captured graph contains memset nodes {bad} and this HIP runtime does not order them on "
                           "replay: set DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before torch is imported")
    return fn


def graph_node_census(cuda_graph) -> dict:
    """{node type: count} of a ``torch.cuda.CUDAGraph`` created with ``keep_graph=True``."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    raw = ctypes.c_void_p(cuda_graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
        return {"error": 1}
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    if hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
        return {"error": 1}
    out = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
        name = _NODE_TYPES.get(t.value, f"type{t.value}")
        out[name] = out.get(name, 0) + 1
    return out


def enabled() -> bool:
    return os.environ.get("MEMOTR_DECODER_GRAPHS", "1") != "0"


def paired_query_projections(root: nn.Module, named):
    """Order the parameters of ``root`` for the flat argument of a capture so that, for every deformable-attention module,
    ``sampling_offsets.weight`` is directly followed by ``attention_weights.weight`` and the two biases likewise: the
    stacked (offsets; logits) projection the module feeds its one query GEMM with is then a VIEW of the flat tensor, not a
    concatenation made by every replay (two kernels per layer and frame).
    Returns (names, parameters, groups): ``groups`` = [(member names, member shapes, stacked shape or None), ...] in flat
    order -- the flat tensor is split ONCE, by group; a pair is one piece of that split (``split_parameters``)."""
    from ..modules.ms_deform_attn import MSDeformAttn
    by_name = dict(named)
    pairs = []
    for mod_name, m in root.named_modules():
        if isinstance(m, MSDeformAttn) and os.environ.get("MEMOTR_QPROJ_VIEW", "1") != "0":
            pre = mod_name + "." if mod_name else ""
            keys = [pre + k for k in ("sampling_offsets.weight", "attention_weights.weight", "sampling_offsets.bias",
                                      "attention_weights.bias")]
            if all(k in by_name for k in keys) and by_name[keys[0]].dtype == by_name[keys[1]].dtype \
                    and by_name[keys[0]].shape[1:] == by_name[keys[1]].shape[1:]:
                pairs.append(keys)
    taken = {k for keys in pairs for k in keys}
    groups = [((n,), (tuple(p.shape),), None) for n, p in named if n not in taken]
    for w0, w1, b0, b1 in pairs:
        for a, b in ((w0, w1), (b0, b1)):
            pa, pb = by_name[a], by_name[b]
            groups.append(((a, b), (tuple(pa.shape), tuple(pb.shape)), (pa.shape[0] + pb.shape[0],) + tuple(pa.shape[1:])))
    names = tuple(n for g in groups for n in g[0])
    params = tuple(by_name[n] for n in names)
    return names, params, groups


def split_parameters(flat: torch.Tensor, groups) -> dict:
    """{name: view} of the flat parameter tensor for ``torch.func.functional_call``.  ONE split of ``flat`` (its backward:
    one concatenation); a pair's piece is the module's stacked weight (bias) as it lies, tagged on the
    ``sampling_offsets`` stand-ins for ``MSDeformAttn._fused_query_projection``.  The pair's members are views of the
    PIECE, and the fused module does not read them: nothing but the split stands between the flat tensor and the stack.
    (Round 6 first took the stack as ``flat.narrow(...)``: each narrow's backward is a zero-fill of the WHOLE flat tensor
    -- 46 MB -- a memcpy node and a full-size add: 12 memcpy nodes and 48 kernels per decoder backward graph, +3 ms per
    train step for the 0.06 ms the view saved in the forward; tools/qproj_ab.sh, profiles/r06_qproj_ab.txt.)"""
    def numel(shape):
        n = 1
        for d in shape:
            n *= d
        return n

    pieces = flat.split([sum(numel(sh) for sh in g[1]) for g in groups])
    sub, stacks = {}, {}
    for (names, shapes, stacked), piece in zip(groups, pieces):
        if stacked is None:
            sub[names[0]] = piece.view(shapes[0])
            continue
        off = 0
        for n, sh in zip(names, shapes):
            sub[n] = piece.narrow(0, off, numel(sh)).view(sh)
            off += numel(sh)
        stacks[names[0]] = piece.view(stacked)
    for first in [k for k in stacks if k.endswith("sampling_offsets.weight")]:
        sub[first]._msda_fused_qproj = (stacks[first], stacks[first[:-len("weight")] + "bias"])
    return sub


class DecoderGraphs(GraphCache):
    """Cache of captured decoder steps, owned by a ``DeformableDecoder``."""

    def __init__(self, decoder):
        super().__init__("decoder", MAX_GRAPHS, grow_cap=2)
        self.decoder = decoder

    def usable(self, output, src) -> bool:
        d = self.decoder
        # use_checkpoint: the graph pair replaces the per-layer checkpoints of the decoder -- its activations are
        # (300 + n) x 256 per layer, nothing next to the backbone / encoder segments that stay checkpointed
        # (MEMOTR_CHECKPOINT_DECODER=1 keeps the reference's per-layer recompute, eager).
        # extra_track_attn: a frame without tracks still carries one masked (padded) track slot in the graphed path,
        # and attention over keys that are ALL masked is 0 * inf -- its rows are sliced away, but the NaN reaches the
        # parameter gradients through the norms; those models keep the eager loop
        return (enabled() and not self.failed and output.is_cuda and d.use_dab and d.bbox_embed is not None
                and (not d.use_checkpoint or os.environ.get("MEMOTR_CHECKPOINT_DECODER", "0") != "1")
                and torch.is_grad_enabled() and src.requires_grad
                and not torch.is_autocast_enabled() and output.dtype == torch.float32
                and not any(getattr(layer, "extra_track_attn", False) for layer in d.layers))

    @staticmethod
    def bucket(n_queries: int, n_det: int) -> int:
        # the TOTAL is rounded up to a multiple of BUCKET (the fused attention kernels want aligned key counts) with
        # at least one (masked) track slot: a zero-sized track part would put empty copy nodes into the capture
        n = max(n_queries, n_det + 1)
        return (n + BUCKET - 1) // BUCKET * BUCKET

    def run(self, frame_slot: int, args, shapes, lsi, clip_key=None):
        """The decoder loop of frame ``frame_slot`` through its graph (captured on first use); None if capture failed."""
        key = (frame_slot, tuple(a.shape for a in args), tuple(bool(a.requires_grad) for a in args),
               self._geometry(shapes), clip_ops.config_key())          # (a capture bakes the kernel choice in)
        slot = self.lookup(key, lambda: self._capture(args, shapes, lsi))
        if slot is None:
            return None
        fn, params = slot[0], slot[1]
        self.replays += 1
        return fn(*args, self._flat_parameters(params, clip_key))

    @staticmethod
    def _geometry(shapes):
        """The pyramid as a hashable value ((H, W), ...): the identity of the tensor object is not a key -- an id can
        be recycled once the geometry cache evicts the tensor."""
        from ..MultiScaleDeformableAttention import host_shapes
        return tuple(map(tuple, host_shapes(shapes).tolist()))

    def _flat_parameters(self, params, clip_key):
        """All decoder parameters as ONE tensor, made once per clip and read by the graphs of all its frames.

        A graph returns the gradient of each tensor argument in its own buffer and autograd adds the frames up
        argument by argument: with ~170 parameter tensors as arguments that was 170 copy / add kernels per frame
        outside the graphs (850 per train step, ~5 ms).  With one flat argument the frames' gradients meet in four
        adds, the split back to the parameters happens once per clip (the cat's backward hands out views), and
        DistributedDataParallel's hooks still fire once per parameter."""
        cache = self.__dict__.get("_flat")
        if (clip_key is not None and cache is not None and cache[0] is clip_key and len(cache[1]) == len(params)
                and all(a is b for a, b in zip(cache[1], params))):
            return cache[2]
        flat = torch.cat([p.reshape(-1) for p in params])
        self.__dict__["_flat"] = (clip_key, params, flat)
        return flat

    def _capture(self, args, shapes, lsi):
        """Capture ``DecoderLoop`` as a function of (inputs..., flat parameters).

        The parameters travel as an ordinary tensor ARGUMENT (``torch.func.functional_call`` substitutes views of
        it): the captured backward then differentiates with respect to a fresh leaf tensor only.  Capturing with
        respect to the live ``nn.Parameter`` objects instead makes autograd reuse their gradient-accumulator nodes,
        which remember the stream they were created on -- any earlier use of a decoder parameter on the default
        stream (an eager step, a kept-alive graph) then drags the legacy stream into the capture and
        hipStreamEndCapture faults.  Inside the graph the flat tensor is split into the parameter shapes (views);
        the split's backward is one concatenation."""
        loop = DecoderLoop(self.decoder, shapes, lsi)
        named = list(loop.named_parameters())
        if len(named) != sum(1 for _ in loop.named_parameters(remove_duplicate=False)):
            self.failed = True      # a module shared between layers (no box refinement clones): see DecoderLoop
            return None
        names, params, groups = paired_query_projections(loop, named)
        n_user = len(args)

        def run(*flat_in):
            return torch.func.functional_call(loop, split_parameters(flat_in[n_user], groups), tuple(flat_in[:n_user]))

        with torch.no_grad():
            flat = torch.cat([p.reshape(-1) for p in params])
        sample = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args) + (flat.requires_grad_(True),)
        try:
            fn = checked_capture(lambda: torch.cuda.make_graphed_callables(run, sample, num_warmup_iters=2,
                                                                           allow_unused_input=True))
        except Exception as exc:  # noqa: BLE001 -- capture is an optimisation; eager stays valid
            return self.capture_failed(exc)
        # functional_call must have put every nn.Parameter back (see DecoderLoop)
        assert all(isinstance(p, nn.Parameter) for p in loop.parameters()) and \
            sorted(id(p) for p in loop.parameters()) == sorted(id(p) for p in params), "decoder parameters were replaced"
        self.captures += 1
        return fn, params
