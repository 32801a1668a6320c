"""CPU checks of the host-side plumbing added around the reference's frame loop: encode grouping, the lean
self-attention, the activation-before-reshape linear, BLAS selection knobs and the encode/decode split."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Core:
    pass


@pytest.mark.parametrize("spec,clip_len,want", [
    (None, 5, ([5], False)), ("auto", 5, ([3, 2], True)), ("auto", 4, ([3, 1], True)), ("auto", 2, ([2], True)),
    ("auto", 1, ([1], True)), ("all", 5, ([5], False)), ("0", 5, (None, False)), ("1", 3, ([1, 1, 1], False)),
    ("lazy:2", 5, ([2, 2, 1], True)), ("1,4", 5, ([1, 4], False)), ([1, 1, 3], 5, ([1, 1, 3], False)),
    (8, 5, ([5], False)), ("2,9", 5, ([2, 3], False)),
])
def test_encode_chunks_parsing(monkeypatch, spec, clip_len, want):
    from memotr_amd.engine import encode_chunks
    monkeypatch.delenv("MEMOTR_ENCODE_CHUNKS", raising=False)
    core = _Core()
    if spec is not None:
        core.encode_chunks = spec
    assert encode_chunks(core, clip_len) == want
    groups = want[0]
    if groups is not None:
        assert sum(groups) == clip_len and all(g > 0 for g in groups)


def test_encode_chunks_env_and_checkpoint(monkeypatch):
    from memotr_amd.engine import encode_chunks
    monkeypatch.setenv("MEMOTR_ENCODE_CHUNKS", "all")
    assert encode_chunks(_Core(), 4) == ([4], False)
    core = _Core()
    core.use_checkpoint = True          # activation checkpointing groups like the plain step (segments recompute per
    assert encode_chunks(core, 4) == ([4], False)                                   # group) unless told otherwise
    monkeypatch.setenv("MEMOTR_CHECKPOINT_REFERENCE_ORDER", "1")
    assert encode_chunks(core, 4) == (None, False)


@pytest.mark.parametrize("masked", [False, True])
def test_self_attention_equals_multihead_attention(masked):
    """models/deformable_decoder.py self_attn call of the reference: mha(q=k=tgt+pos, v=tgt, key_padding_mask)."""
    from memotr_amd.modules.attention import self_attention
    torch.manual_seed(0)
    mha = nn.MultiheadAttention(64, 8, dropout=0.0, batch_first=True).train()
    x = torch.randn(2, 37, 64, requires_grad=True)
    pos = torch.randn(2, 37, 64)
    mask = None
    if masked:
        mask = torch.zeros(2, 37, dtype=torch.bool)
        mask[1, 30:] = True
    got = self_attention(mha, x + pos, x, mask)
    want = mha(x + pos, x + pos, x, key_padding_mask=mask, need_weights=False)[0]
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    params = [x] + list(mha.parameters())
    g_got = torch.autograd.grad(got.square().sum(), params)
    g_want = torch.autograd.grad(want.square().sum(), params)
    for a, b in zip(g_got, g_want):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


def test_self_attention_falls_back_for_other_layouts():
    from memotr_amd.modules.attention import self_attention
    torch.manual_seed(1)
    mha = nn.MultiheadAttention(32, 4, batch_first=True, kdim=16, vdim=16)       # separate projection weights
    q = torch.randn(1, 5, 32)
    with pytest.raises(Exception):                   # the module itself rejects value of the wrong width
        self_attention(mha, q, q, None)


def test_long_linear_applies_activation_before_the_reshape():
    """An in-place ReLU on the reshaped view of the custom Function's output makes autograd insert CopySlices
    (whole-gradient copies); applied through ``activation=`` the graph is ReLU -> view."""
    from memotr_amd.modules.linear import long_linear
    torch.manual_seed(2)
    x = torch.randn(1, 64, 16, requires_grad=True)
    w = torch.randn(32, 16, requires_grad=True)
    b = torch.randn(32, requires_grad=True)
    h = long_linear(x, w, b, min_rows=8, activation=nn.ReLU(True))
    assert type(h.grad_fn).__name__ == "ViewBackward0"
    assert type(h.grad_fn.next_functions[0][0]).__name__ == "ReluBackward0"
    ref = torch.relu(F.linear(x, w, b))
    assert torch.equal(h, ref)
    g = torch.autograd.grad(h.sum(), (x, w, b))
    g_ref = torch.autograd.grad(ref.sum(), (x, w, b))
    for a, r in zip(g, g_ref):
        torch.testing.assert_close(a, r, rtol=1e-5, atol=1e-5)
    small = long_linear(x, w, b, activation=torch.relu)          # below min_rows: plain F.linear path
    assert torch.equal(small, ref)


def test_blas_selection_knobs(monkeypatch):
    from memotr_amd.modules import linear
    before = torch.backends.cuda.preferred_blas_library()
    try:
        monkeypatch.setenv("MEMOTR_BLAS", "keep")
        assert linear.configure_blas() == "keep"
        assert torch.backends.cuda.preferred_blas_library() == before
        monkeypatch.setenv("MEMOTR_BLAS", "rocblas")
        assert linear.configure_blas() == "rocblas"
        assert torch.backends.cuda.preferred_blas_library() == torch._C._BlasBackend.Cublas
        with linear.prefer_blas("cublaslt"):
            assert torch.backends.cuda.preferred_blas_library() == torch._C._BlasBackend.Cublaslt
        assert torch.backends.cuda.preferred_blas_library() == torch._C._BlasBackend.Cublas     # restored
    finally:
        torch.backends.cuda.preferred_blas_library(before)


def test_model_forward_equals_encode_then_decode(monkeypatch):
    """``model(frame, tracks)`` (the reference contract) == ``model(tracks=, encoded=model(frame=, stage="encode"))``,
    and a batched encode split per frame equals the per-frame encodes."""
    from model_helpers import build_small_memotr, patch_operator
    from memotr_amd.structures.track_instances import TrackInstances
    from memotr_amd.utils.nested_tensor import tensor_list_to_nested_tensor
    patch_operator(monkeypatch)
    torch.manual_seed(3)
    model = build_small_memotr().eval()
    imgs = [torch.randn(3, 96, 128), torch.randn(3, 96, 128)]
    tracks = [TrackInstances(hidden_dim=model.hidden_dim, num_classes=model.num_classes, use_dab=True)]
    with torch.no_grad():
        whole = model(frame=tensor_list_to_nested_tensor([imgs[0]]), tracks=tracks)
        enc = model(frame=tensor_list_to_nested_tensor([imgs[0]]), stage="encode")
        halves = model(tracks=tracks, encoded=enc)
        both = model(frame=tensor_list_to_nested_tensor(imgs), stage="encode")
        enc1 = model(frame=tensor_list_to_nested_tensor([imgs[1]]), stage="encode")
    for k in ("pred_logits", "pred_bboxes", "outputs", "last_ref_pts"):
        assert torch.equal(whole[k], halves[k]), k
    torch.testing.assert_close(both["memory"][0:1], enc["memory"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(both["memory"][1:2], enc1["memory"], rtol=1e-5, atol=1e-5)


def test_query_mask_tag_drops_an_all_false_mask():
    """get_query_mask tags the mask when the host knows nothing is padded; the attention then runs mask-free."""
    from model_helpers import build_small_memotr
    from memotr_amd.modules.attention import self_attention
    from memotr_amd.structures.track_instances import TrackInstances
    torch.manual_seed(4)
    model = build_small_memotr()

    def tracks_with(n):
        t = TrackInstances(hidden_dim=model.hidden_dim, num_classes=model.num_classes, use_dab=True)
        t.query_embed = torch.randn(n, model.hidden_dim)
        t.ref_pts = torch.randn(n, 4)
        return t

    assert model.get_query_mask([tracks_with(0)])._no_padding
    assert model.get_query_mask([tracks_with(3)])._no_padding
    assert model.get_query_mask([tracks_with(3), tracks_with(3)])._no_padding
    ragged = model.get_query_mask([tracks_with(3), tracks_with(1)])
    assert not ragged._no_padding and bool(ragged[1, -2:].all()) and not bool(ragged[0].any())
    assert model.get_query_mask([tracks_with(3), tracks_with(0)])._no_padding      # reference :271-274: stays unmasked

    mha = nn.MultiheadAttention(64, 8, batch_first=True)
    x = torch.randn(1, 9, 64)
    mask = torch.zeros(1, 9, dtype=torch.bool)
    mask._no_padding = True
    torch.testing.assert_close(self_attention(mha, x, x, mask), mha(x, x, x, need_weights=False)[0], rtol=1e-5,
                               atol=1e-6)


def test_decoder_loop_module_equals_the_eager_decoder_and_leaves_parameters_alone(monkeypatch):
    """models/decoder_graphs.DecoderLoop is what gets captured in hipGraphs on the GPU: on the CPU it must reproduce
    DeformableDecoder.forward exactly, give every parameter a gradient through torch.func.functional_call, and leave
    the model's nn.Parameter objects in place (functional_call does not restore modules reachable under two names)."""
    import torch
    from model_helpers import build_small_memotr, patch_operator
    from memotr_amd.models.decoder_graphs import DecoderGraphs, DecoderLoop
    patch_operator(monkeypatch)
    torch.manual_seed(0)
    model = build_small_memotr().train()
    dec = model.transformer.decoder
    before = {n: id(p) for n, p in model.named_parameters()}
    shapes, lsi, S, nq = torch.tensor([[6, 8], [3, 4], [2, 2], [1, 1]]), torch.tensor([0, 48, 60, 64]), 65, 32
    loop = DecoderLoop(dec, shapes, lsi)
    names, params = zip(*loop.named_parameters())
    assert len(names) == sum(1 for _ in loop.named_parameters(remove_duplicate=False))
    mask = torch.zeros(1, nq, dtype=torch.bool)
    mask[:, 26:] = True
    args = (torch.randn(1, nq, 64).requires_grad_(True), torch.rand(1, nq, 4).requires_grad_(True),
            torch.randn(1, S, 64).requires_grad_(True), torch.ones(1, 1, 4, 4), mask, torch.zeros(1, S, dtype=torch.bool))
    flat = tuple(p.detach().requires_grad_(True) for p in params)
    outs = torch.func.functional_call(loop, dict(zip(names, flat)), args)
    grads = torch.autograd.grad(sum(o.sum() for o in outs if o.requires_grad), flat, allow_unused=True)
    assert all(g is not None for g in grads)
    assert {n: id(p) for n, p in model.named_parameters()} == before
    eager = dec(args[0], args[1], args[2], shapes, lsi, torch.ones(1, 4, 2), None, mask, args[5])
    for a, b in zip(outs, eager):
        assert torch.equal(a, b)
    assert DecoderGraphs.bucket(300, 300) == 320 and DecoderGraphs.bucket(311, 300) == 320
    assert DecoderGraphs.bucket(321, 300) == 352 and DecoderGraphs.bucket(20, 20) == 32


def test_position_embedding_cache_by_image_geometry():
    """The sine embedding is a function of the padding mask; NestedTensors that carry their image sizes are served
    from a cache keyed by them, anything else is computed."""
    import math

    import torch
    from memotr_amd.models.position_embedding import PositionEmbeddingSine
    from memotr_amd.utils.nested_tensor import NestedTensor, tensor_list_to_nested_tensor
    pe = PositionEmbeddingSine(num_pos_feats=8, normalize=True, scale=2 * math.pi, temperature=20)
    a = tensor_list_to_nested_tensor([torch.zeros(3, 40, 50), torch.zeros(3, 33, 64)])
    assert a.sizes == ((64, 64), (40, 50), (33, 64))
    first = pe(a)
    again = pe(tensor_list_to_nested_tensor([torch.ones(3, 40, 50), torch.ones(3, 33, 64)]))
    assert again is first                                   # same geometry, other pixels: the cached tensor
    plain = pe(NestedTensor(a.tensors, a.masks))            # no sizes: computed
    assert plain is not first and torch.equal(plain, first)
    other = pe(tensor_list_to_nested_tensor([torch.zeros(3, 40, 50), torch.zeros(3, 40, 64)]))
    assert other is not first and not torch.equal(other[1], first[1])
    assert len(pe._cache) == 2
    import copy
    assert "_cache" not in copy.deepcopy(pe).__dict__


def test_model_with_runtime_caches_can_be_deep_copied_and_pickled():
    """Per-process / per-geometry caches (decoder graphs, pyramid tensors, mask-derived tensors, position embeddings)
    stay out of copies and pickles; parameters do not."""
    import copy
    import io

    import torch
    from memotr_amd.configs import dancetrack_config
    from memotr_amd.models.deformable_transformer import build as build_tr
    cfg = dancetrack_config(DEVICE="cpu", HIDDEN_DIM=64, FFN_DIM=128, NUM_ENC_LAYERS=1, NUM_DEC_LAYERS=2)
    tr = build_tr(cfg)
    tr.decoder.graphs()                                   # creates the (empty) graph cache object
    tr.__dict__["_mask_derived"] = {("k",): (torch.zeros(1),)}
    tr._pyramid_tensors([(4, 6), (2, 3)], torch.device("cpu"))
    twin = copy.deepcopy(tr)
    assert "_decoder_graphs" not in twin.decoder.__dict__ and "_mask_derived" not in twin.__dict__
    assert "_pyramids" not in twin.__dict__
    buf = io.BytesIO()
    torch.save(tr, buf)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    assert sorted(back.state_dict()) == sorted(tr.state_dict())
    assert "_decoder_graphs" in tr.decoder.__dict__        # the original keeps its caches


def test_packed_in_projection_node_matches_sliced_linears():
    """_PackedInProj (one autograd node over in_proj_weight / in_proj_bias) against the sliced F.linear formulation,
    values and all four gradients, in float64 on the CPU."""
    from memotr_amd.modules.attention import _PackedInProj
    g = torch.Generator().manual_seed(0)
    E = 16
    qk = torch.randn(2, 5, E, generator=g, dtype=torch.float64)
    v = torch.randn(2, 5, E, generator=g, dtype=torch.float64)
    w = torch.randn(3 * E, E, generator=g, dtype=torch.float64)
    b = torch.randn(3 * E, generator=g, dtype=torch.float64)
    up = torch.randn(2, 5, 3 * E, generator=g, dtype=torch.float64)
    res = {}
    for name in ("node", "sliced"):
        a, c, wi, bi = (t.clone().requires_grad_(True) for t in (qk, v, w, b))
        if name == "node":
            y_qk, y_v = _PackedInProj.apply(a, c, wi, bi)
        else:
            y_qk, y_v = F.linear(a, wi[:2 * E], bi[:2 * E]), F.linear(c, wi[2 * E:], bi[2 * E:])
        (torch.cat((y_qk, y_v), -1) * up).sum().backward()
        res[name] = (y_qk.detach(), y_v.detach(), a.grad, c.grad, wi.grad, bi.grad)
    for x, y in zip(res["node"], res["sliced"]):
        torch.testing.assert_close(x, y, rtol=1e-12, atol=1e-12)


def test_track_instances_row_gather_and_padded_stack():
    """A 1-d index tensor gathers through index_select (same rows as advanced indexing); the (B, max_len, width) stack
    of per-clip track tensors is zero-padded without in-place writes and keeps the gradient path."""
    from memotr_amd.structures.track_instances import TrackInstances
    t = TrackInstances(hidden_dim=8, num_classes=1, use_dab=True)
    n = 6
    t.ids = torch.arange(n)
    t.query_embed = torch.randn(n, 8, requires_grad=True)
    t.ref_pts = torch.randn(n, 4)
    t.boxes, t.logits = torch.rand(n, 4), torch.randn(n, 1)
    t.output_embed, t.last_output, t.long_memory = torch.randn(n, 8), torch.randn(n, 8), torch.randn(n, 8)
    t.matched_idx, t.labels, t.iou = torch.zeros(n, dtype=torch.long), torch.zeros(n, dtype=torch.long), torch.zeros(n)
    idx = torch.tensor([4, 0, 4])
    picked = t[idx]
    assert picked.ids.tolist() == [4, 0, 4] and torch.equal(picked.query_embed, t.query_embed[idx])
    picked.query_embed.sum().backward()
    assert t.query_embed.grad[4].tolist() == [2.0] * 8 and t.query_embed.grad[1].tolist() == [0.0] * 8
    kept = t[torch.tensor([True, False, True, False, False, True])]
    assert kept.ids.tolist() == [0, 2, 5]

    from memotr_amd.models.memotr import MeMOTR

    class _Stub:
        det_query_embed = torch.zeros(3, 8)
    parts = [torch.ones(2, 4, requires_grad=True), torch.full((3, 4), 2.0), torch.zeros(0, 4)]
    out = MeMOTR._pad_stack(_Stub(), parts, 4)
    assert out.shape == (3, 3, 4) and out[0, 2].abs().sum() == 0 and out[2].abs().sum() == 0
    assert torch.equal(out[1], parts[1]) and out.requires_grad
    single = MeMOTR._pad_stack(_Stub(), [parts[0]], 4)
    assert single.shape == (1, 2, 4) and single._base is parts[0]          # a view, no copy
    assert MeMOTR._pad_stack(_Stub(), [torch.zeros(0, 4)], 4).shape == (1, 0, 4)


def test_decoder_graphs_flat_parameters_are_shared_within_a_clip_only():
    from memotr_amd.models.decoder_graphs import DecoderGraphs
    g = DecoderGraphs(decoder=None)
    params = (nn.Parameter(torch.randn(3, 2)), nn.Parameter(torch.randn(5)))
    clip_a, clip_b = object(), object()
    f1 = g._flat_parameters(params, clip_a)
    assert f1.shape == (11,) and f1.requires_grad
    assert g._flat_parameters(tuple(params), clip_a) is f1               # same clip, same parameters: the same tensor
    assert g._flat_parameters(params, clip_b) is not f1                   # a new clip re-reads the parameters
    assert g._flat_parameters(params, None) is not g._flat_parameters(params, None)   # no key: never cached
    f1.sum().backward()
    assert torch.equal(params[0].grad, torch.ones(3, 2))
