"""CPU: the host-side logic around the hipGraph captures and the online-tracking loop (no device needed: captures are
stood in for, tensors are on the CPU)."""
import contextlib

import pytest
import torch

from memotr_amd.models import decoder_graphs as dg
from memotr_amd.models import infer_graphs as ig
from memotr_amd.models.runtime_tracker import RuntimeTracker
from memotr_amd.structures.track_instances import TrackInstances


# ----------------------------------------------------------------------------- capture guard
def test_checked_capture_refuses_memset_nodes_only_on_a_runtime_that_misorders_them(monkeypatch):
    seen = []

    @contextlib.contextmanager
    def fake_capture(census=None):
        seen.append(census)
        if census is not None:
            census.append({"kernel": 3, "memset": 1})
        yield

    monkeypatch.setattr(dg, "_thread_local_capture", fake_capture)
    monkeypatch.setattr(dg, "CENSUS", None)
    monkeypatch.setattr(dg, "memset_nodes_replay_safe", lambda: True)
    assert dg.checked_capture(lambda: "graph") == "graph" and seen[-1] is None      # safe runtime: no inspection
    monkeypatch.setattr(dg, "memset_nodes_replay_safe", lambda: False)
    with pytest.raises(RuntimeError, match="DEBUG_CLR_GRAPH_PACKET_CAPTURE"):
        dg.checked_capture(lambda: "graph")
    assert seen[-1] == [{"kernel": 3, "memset": 1}]


def test_encode_graph_switch_follows_the_environment_and_autocast(monkeypatch):
    from memotr_amd.models import encode_graphs as eg
    monkeypatch.setenv("MEMOTR_ENCODE_GRAPHS", "1")
    assert eg.enabled()
    monkeypatch.setenv("MEMOTR_ENCODE_GRAPHS", "0")
    assert not eg.enabled()
    monkeypatch.setenv("MEMOTR_ENCODE_GRAPHS", "auto")
    assert not eg.enabled()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        assert eg.enabled() == torch.is_autocast_enabled()


# ----------------------------------------------------------------------------- forward-only graph cache
class _FakeCache(ig.ForwardGraphs):
    def __init__(self):
        super().__init__("test")
        self.made = []

    def _capture(self, fn, inputs, pins):
        self.made.append(fn)
        self.captures += 1

        class _G:
            def replay(self_inner):
                pass
        return _G(), tuple(t.clone() for t in inputs), ("out", fn), (fn, tuple(pins))


def test_forward_graph_cache_reuses_evicts_and_stops_thrashing(monkeypatch):
    monkeypatch.delenv("MEMOTR_REQUIRE_GRAPHS", raising=False)
    cache = _FakeCache()
    x = (torch.zeros(2),)
    out = cache.run("a", lambda: "fa", x)
    assert out == ("out", "fa") and cache.captures == 1 and cache.replays == 1
    assert cache.run("a", lambda: pytest.fail("captured twice"), x) == ("out", "fa") and cache.replays == 2
    for i in range(ig.MAX_GRAPHS):                      # least recently used entries leave
        cache._misses = 0
        cache.run(("k", i), lambda: "f", x)
    assert "a" not in cache.slots and len(cache.slots) == ig.MAX_GRAPHS
    cache = _FakeCache()
    for i in range(ig.MISS_LIMIT + 4):                  # a new key every call: captures stop after MISS_LIMIT
        cache.run(("n", i), lambda: "f", x)
    assert cache.captures == ig.MISS_LIMIT and cache.eager == 4
    assert cache.run(("n", ig.MISS_LIMIT - 1), lambda: "f", x) is not None         # what exists still replays
    monkeypatch.setenv("MEMOTR_REQUIRE_GRAPHS", "1")    # the benchmark setting never gives up
    cache = _FakeCache()
    for i in range(ig.MISS_LIMIT + 4):
        cache.run(("n", i), lambda: "f", x)
    assert cache.captures == ig.MISS_LIMIT + 4 and cache.eager == 0


def test_forward_graph_cache_copies_inputs_into_the_static_buffers():
    cache = _FakeCache()
    a = torch.arange(4.0)
    cache.run("k", lambda: "f", (a,))
    b = torch.arange(4.0) + 10
    cache.run("k", lambda: "f", (b,))
    assert torch.equal(cache.slots["k"][1][0], b)


def test_parameter_fingerprint_follows_storage_not_values():
    m = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    f0 = ig._fingerprint(m)
    with torch.no_grad():
        m[0].weight.add_(1.0)                           # in-place update (optimiser step, load_state_dict): same storage
    assert ig._fingerprint(m) == f0
    m[0].weight = torch.nn.Parameter(torch.zeros(4, 4))     # the first parameter moved: seen at once
    assert ig._fingerprint(m) != f0
    f1 = ig._fingerprint(m)
    m.double()                                          # every tensor re-allocated
    assert ig._fingerprint(m) != f1


# ----------------------------------------------------------------------------- runtime tracker / result packing
def _outputs(n_det=6, n_track=3, K=2, C=8, seed=0):
    g = torch.Generator().manual_seed(seed)
    n = n_det + n_track
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    return {"pred_logits": r(1, n, K) * 3, "pred_bboxes": torch.rand(1, n, 4, generator=g), "outputs": r(1, n, C),
            "last_ref_pts": r(1, n, 4), "det_query_embed": r(n_det, C),
            "aux_outputs": [{"queries": r(1, n, C)}]}


def test_runtime_tracker_births_match_the_boolean_mask_formulation():
    """One nonzero + index_select per field (models/runtime_tracker.py) against the reference's per-field boolean
    indexing (runtime_tracker.py:60-75): same rows, same order, ids counted up from max_obj_id."""
    from memotr_amd.models.utils import logits_to_scores
    out = _outputs()
    n_det = 6
    tracks = TrackInstances(hidden_dim=8, num_classes=2, use_dab=True)
    tracks.ids = torch.tensor([3, 4, 5])
    tracks.labels = torch.tensor([0, 1, 0])
    tracks.disappear_time = torch.tensor([0, 2, 4])
    tracks.query_embed, tracks.ref_pts = torch.zeros(3, 8), torch.zeros(3, 4)
    rt = RuntimeTracker(det_score_thresh=0.5, track_score_thresh=0.5, miss_tolerance=5, use_dab=True)
    rt.max_obj_id = 6
    prev, new = rt.update(model_outputs=out, tracks=[tracks])
    scores = logits_to_scores(out["pred_logits"])[0]
    keep = scores[:n_det].max(-1).values >= 0.5
    assert 0 < int(keep.sum()) < n_det
    n = new[0]
    assert torch.equal(n.logits, out["pred_logits"][0][:n_det][keep])
    assert torch.equal(n.boxes, out["pred_bboxes"][0][:n_det][keep])
    assert torch.equal(n.output_embed, out["outputs"][0][:n_det][keep])
    assert torch.equal(n.query_embed, out["aux_outputs"][-1]["queries"][0][:n_det][keep])
    assert n.ids.tolist() == list(range(6, 6 + int(keep.sum()))) and rt.max_obj_id == 6 + int(keep.sum())
    assert torch.equal(n.labels, scores[:n_det][keep].max(-1).indices)
    own = scores[n_det:].gather(1, torch.tensor([[0], [1], [0]])).squeeze(1)
    want_dt = torch.where(own < 0.5, torch.tensor([1, 3, 5]), torch.zeros(3, dtype=torch.long))
    assert torch.equal(prev[0].disappear_time, want_dt)
    assert torch.equal(prev[0].ids, torch.where(want_dt >= 5, torch.tensor(-1), torch.tensor([3, 4, 5])))


def test_sequence_tracker_report_filters_and_converts_like_the_submit_loop():
    """submit_engine.py:95-112 on a CPU TrackInstances: score > threshold, area > threshold, cxcywh -> xyxy pixels."""
    from memotr_amd.inference import SequenceTracker
    st = SequenceTracker.__new__(SequenceTracker)
    st.result_score_thresh, st.area_thresh, st.use_dab = 0.5, 100, True
    t = TrackInstances(hidden_dim=8, num_classes=2, use_dab=True)
    t.boxes = torch.tensor([[0.5, 0.5, 0.2, 0.2], [0.3, 0.3, 0.01, 0.01], [0.7, 0.6, 0.4, 0.1], [0.2, 0.8, 0.3, 0.3]])
    t.scores = torch.tensor([[0.9, 0.1], [0.8, 0.2], [0.2, 0.3], [0.1, 0.7]])
    t.ids, t.labels = torch.tensor([7, 8, 9, 10]), torch.tensor([0, 0, 1, 1])
    t.query_embed, t.ref_pts = torch.zeros(4, 8), torch.zeros(4, 4)
    out = st._report(t, 200, 400)
    assert out.ids.tolist() == [7, 10] and out.labels.tolist() == [0, 1] and len(out) == 2     # tiny box and low score gone
    want = torch.tensor([[160.0, 80.0, 240.0, 120.0], [20.0, 130.0, 140.0, 190.0]])
    assert torch.allclose(out.boxes, want, atol=1e-3)
    assert torch.allclose(out.area, torch.tensor([0.2 * 400 * 0.2 * 200, 0.3 * 400 * 0.3 * 200]))
    empty = st._report(TrackInstances(hidden_dim=8, num_classes=2, use_dab=True), 200, 400)
    assert len(empty) == 0 and empty.boxes.shape == (0, 4)


def test_flat_parameters_split_once_with_the_stacked_query_projection_as_one_piece():
    """models/decoder_graphs.py: the decoder's parameters enter a capture as ONE flat tensor; ``split_parameters`` hands
    ``functional_call`` a view per parameter and, per deformable-attention module, the stacked (offsets; logits) weight
    and bias as they lie in the flat tensor.  Nothing but the one split may stand between the flat tensor and a stack: a
    ``narrow`` of the flat tensor would put a zero-fill of the whole tensor, a copy and a full-size add per use into the
    captured backward (+3 ms per train step when round 6 did that)."""
    import torch.nn as nn
    from memotr_amd.models.decoder_graphs import paired_query_projections, split_parameters
    from memotr_amd.modules.ms_deform_attn import MSDeformAttn

    class Root(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = MSDeformAttn(d_model=64, n_levels=2, n_heads=4, n_points=2)
            self.lin = nn.Linear(3, 5)
            self.b = MSDeformAttn(d_model=64, n_levels=2, n_heads=4, n_points=2)

    torch.manual_seed(0)
    root = Root()
    for p in root.parameters():
        nn.init.normal_(p)
    named = list(root.named_parameters())
    names, params, groups = paired_query_projections(root, named)
    assert sorted(names) == sorted(n for n, _ in named) and sum(len(g[0]) for g in groups) == len(named)
    flat = torch.cat([p.reshape(-1) for p in params]).detach().requires_grad_(True)
    sub = split_parameters(flat, groups)
    for n, p in named:
        assert torch.equal(sub[n], p), n
    for mod in ("a", "b"):
        m = getattr(root, mod)
        w, b = sub[f"{mod}.sampling_offsets.weight"]._msda_fused_qproj
        assert torch.equal(w, torch.cat((m.sampling_offsets.weight, m.attention_weights.weight), 0))
        assert torch.equal(b, torch.cat((m.sampling_offsets.bias, m.attention_weights.bias), 0))
    w, b = sub["a.sampling_offsets.weight"]._msda_fused_qproj
    loss = (w * 2).sum() + b.sum() + sub["lin.weight"].sum() * 3
    # the autograd graph from the stack to the flat tensor: views and the ONE split, no narrow / slice node
    seen, todo = set(), [loss.grad_fn]
    while todo:
        fn = todo.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        todo += [f for f, _ in fn.next_functions]
    kinds = {type(f).__name__ for f in seen}
    assert not any("Slice" in k or "Narrow" in k for k in kinds), kinds
    assert sum(1 for f in seen if "Split" in type(f).__name__) == 1, kinds
    loss.backward()
    assert float(flat.grad.sum()) == 2 * w.numel() + b.numel() + 3 * root.lin.weight.numel()


def test_graph_cache_grows_when_an_evicted_key_comes_back(monkeypatch):
    """models/graph_cache.py: a working set one key larger than the limit would miss on every lookup of a cyclic access
    pattern (five frame slots x the track-count buckets of a training run); the limit doubles, up to ``grow_cap`` x, the
    first time an evicted key is asked for again -- and not before."""
    from memotr_amd.models.graph_cache import GraphCache
    monkeypatch.setenv("MEMOTR_REQUIRE_GRAPHS", "1")
    made = []

    def cap(k):
        made.append(k)
        return ("graph", k)

    cache = GraphCache("test", 4, grow_cap=2)
    for rnd in range(3):
        for k in range(5):                               # a working set of five on a limit of four
            assert cache.lookup(k, lambda k=k: cap(k)) == ("graph", k)
    # round 0: five captures, key 0 evicted; round 1: key 0 comes back -> limit 8, captured once more; then all hits
    assert made == [0, 1, 2, 3, 4, 0] and cache.max_graphs == 8 and cache.regrown == 1 and len(cache.slots) == 5
    fixed = GraphCache("test", 4)                        # grow_cap 1: the old behaviour
    n = []
    for rnd in range(3):
        for k in range(5):
            fixed.lookup(k, lambda k=k: n.append(k) or ("graph", k))
    assert len(n) == 15 and fixed.max_graphs == 4
