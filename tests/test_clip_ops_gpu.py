"""GPU: the fused criterion kernels (include/clip_ops_hip.h) against the element-wise torch formulation of the same
reference formulas (models/matcher.py:83-121, models/criterion.py:417-467 of the reference), values and gradients."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(clip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def rand_boxes(*shape, gen):
    c = torch.rand(*shape, 2, generator=gen) * 0.6 + 0.2
    wh = torch.rand(*shape, 2, generator=gen) * 0.3 + 0.02
    return torch.cat((c, wh), -1)


@pytest.mark.parametrize("n_layers,B,Q,Nq,K,T", [(6, 1, 300, 310, 1, 10), (3, 2, 40, 57, 8, 7), (1, 1, 5, 5, 1, 1),
                                                 (6, 1, 300, 300, 1, 0)])
def test_match_cost_equals_the_stacked_torch_cost(n_layers, B, Q, Nq, K, T):
    from memotr_amd.functions import clip_ops
    from memotr_amd.models.matcher import HungarianMatcher
    g = torch.Generator().manual_seed(n_layers * 100 + T)
    logits = (torch.randn(n_layers, B, Nq, K, generator=g) * 3).cuda()
    boxes = rand_boxes(n_layers, B, Nq, gen=g).cuda()
    gt_labels = torch.randint(0, K, (T,), generator=g).cuda()
    gt_boxes = rand_boxes(T, gen=g).cuda()
    m = HungarianMatcher(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
    for b in range(B):
        lg, bx = logits[:, b, :Q], boxes[:, b, :Q]               # strided views, as the criterion passes them
        got = clip_ops.match_cost(lg, bx, gt_labels, gt_boxes, 2.0, 5.0, 2.0)
        want = m.cost_matrix_stacked(lg, bx, gt_labels, gt_boxes)
        assert got.shape == want.shape == (n_layers, Q, T)
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)


def test_match_cost_keeps_the_assignment_of_the_torch_cost():
    from scipy.optimize import linear_sum_assignment
    from memotr_amd.functions import clip_ops
    from memotr_amd.models.matcher import HungarianMatcher
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(6, 1, 300, 1, generator=g) * 2).cuda()
    boxes = rand_boxes(6, 1, 300, gen=g).cuda()
    gt_labels = torch.zeros(12, dtype=torch.long).cuda()
    gt_boxes = rand_boxes(12, gen=g).cuda()
    got = clip_ops.match_cost(logits[:, 0], boxes[:, 0], gt_labels, gt_boxes, 2.0, 5.0, 2.0).cpu().numpy()
    want = HungarianMatcher(2.0, 5.0, 2.0).cost_matrix_stacked(logits[:, 0], boxes[:, 0], gt_labels, gt_boxes).cpu().numpy()
    for l in range(6):
        assert [list(x) for x in linear_sum_assignment(got[l])] == [list(x) for x in linear_sum_assignment(want[l])]


@pytest.mark.parametrize("weighted,indexed", [(False, True), (True, True), (False, False)])
def test_pair_box_loss_values_and_gradients(weighted, indexed):
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(11)
    n_layers, B, Nq, n = 6, 2, 310, 64
    boxes = rand_boxes(n_layers, B, Nq, gen=g).cuda()
    flat = torch.randperm(n_layers * Nq, generator=g)[:n]        # distinct (layer, query) pairs
    lay, q = (flat // Nq).cuda(), (flat % Nq).cuda()
    tgt = rand_boxes(9 if indexed else n, gen=g).cuda()
    gidx = torch.randint(0, 9, (n,), generator=g).cuda() if indexed else None
    w = (torch.rand(n, generator=g) > 0.3).float().cuda() if weighted else None
    up = torch.randn(2, n, generator=g).cuda()
    res = {}
    for name, fn in (("kernel", clip_ops.pair_box_loss), ("torch", clip_ops.pair_box_loss_reference)):
        x = boxes.clone().requires_grad_(True)
        l1, gl = fn(x, lay, q, 1, tgt, gidx, w)
        (l1 * up[0] + gl * up[1]).sum().backward()
        res[name] = (l1.detach(), gl.detach(), x.grad)
    for a, b_, what in zip(res["kernel"], res["torch"], ("l1", "giou loss", "gradient")):
        torch.testing.assert_close(a, b_, rtol=2e-5, atol=2e-6, msg=lambda m, what=what: f"{what}: {m}")
    assert float(res["kernel"][2][:, 0].abs().max()) == 0.0      # batch element 0 was not addressed


def test_pair_box_loss_edge_cases():
    """Disjoint boxes (zero intersection: the clamp's sub-gradient), nested boxes and an exact match."""
    from memotr_amd.functions import clip_ops
    boxes = torch.tensor([[[[0.2, 0.2, 0.1, 0.1], [0.5, 0.5, 0.4, 0.4], [0.5, 0.5, 0.2, 0.2], [0.3, 0.6, 0.2, 0.1]]]]).cuda()
    tgt = torch.tensor([[0.7, 0.7, 0.1, 0.1], [0.5, 0.5, 0.1, 0.1], [0.5, 0.5, 0.2, 0.2], [0.3, 0.6, 0.25, 0.1]]).cuda()
    lay = torch.zeros(4, dtype=torch.long).cuda()
    q = torch.arange(4).cuda()
    res = {}
    for name, fn in (("kernel", clip_ops.pair_box_loss), ("torch", clip_ops.pair_box_loss_reference)):
        x = boxes.clone().requires_grad_(True)
        l1, gl = fn(x, lay, q, 0, tgt)
        (l1 + 2 * gl).sum().backward()
        res[name] = (l1.detach(), gl.detach(), x.grad)
    torch.testing.assert_close(res["kernel"][0], res["torch"][0], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(res["kernel"][1], res["torch"][1], rtol=1e-5, atol=1e-6)
    # the exact match (pair 2) sits on max/min ties, where torch halves the sub-gradient: rows 0, 1, 3 compared
    torch.testing.assert_close(res["kernel"][2][0, 0, [0, 1, 3]], res["torch"][2][0, 0, [0, 1, 3]], rtol=2e-5, atol=2e-6)
    assert torch.isfinite(res["kernel"][2]).all()


@pytest.mark.parametrize("n_layers,B,Nq,n_q,K", [(6, 1, 320, 310, 1), (6, 2, 320, 303, 8), (1, 1, 7, 7, 3)])
def test_focal_loss_per_layer_values_and_gradients(n_layers, B, Nq, n_q, K):
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(K * 10 + B)
    logits = (torch.randn(n_layers, B, Nq, K, generator=g) * 4).cuda()
    logits[0, 0, 0, 0] = 40.0            # saturated either way
    logits[-1, -1, 1, 0] = -40.0
    labels = torch.randint(0, K + 1, (n_layers, n_q), generator=g).cuda()
    up = torch.randn(n_layers, generator=g).cuda()
    res = {}
    for name, fn in (("kernel", clip_ops.focal_loss_per_layer), ("torch", clip_ops.focal_loss_per_layer_reference)):
        x = logits.clone().requires_grad_(True)
        loss = fn(x[:, B - 1, :n_q], labels)
        (loss * up).sum().backward()
        res[name] = (loss.detach(), x.grad)
    torch.testing.assert_close(res["kernel"][0], res["torch"][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(res["kernel"][1], res["torch"][1], rtol=1e-4, atol=1e-6)
    again = clip_ops.focal_loss_per_layer(logits[:, B - 1, :n_q], labels)
    assert torch.equal(again, res["kernel"][0])                  # fixed-order reduction: run-to-run identical


def test_criterion_with_and_without_the_kernels(monkeypatch):
    """One synthetic clip step of the small model: identical matching, losses within float rounding."""
    from memotr_amd.configs import dancetrack_config
    from memotr_amd.engine import clip_forward_backward, clip_to_device, make_synthetic_clip
    from memotr_amd.models import build_model
    from memotr_amd.models.criterion import build as build_criterion
    cfg = dancetrack_config(DEVICE="cuda", AVAILABLE_GPUS="0", NUM_ENC_LAYERS=1, NUM_DEC_LAYERS=3, AUX_LOSS_WEIGHT=[1.0, 1.0])
    torch.manual_seed(3)
    model = build_model(cfg).train()
    batch = clip_to_device(make_synthetic_clip(3, 128, 160, 5, seed=2), torch.device("cuda"))
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("MEMOTR_FUSED_CLIP_OPS", flag)
        monkeypatch.setenv("MEMOTR_DECODER_GRAPHS", "0")
        model.zero_grad()
        loss, loss_dict = clip_forward_backward(model, build_criterion(cfg), batch, torch.device("cuda"))
        out[flag] = (float(loss.detach()), {k: float(v.detach()) for k, v in loss_dict.items()},
                     {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert out["1"][0] == pytest.approx(out["0"][0], rel=1e-5)
    for k, v in out["0"][1].items():
        assert out["1"][1][k] == pytest.approx(v, rel=1e-4, abs=1e-6), k
    for n, gref in out["0"][2].items():
        err = float((out["1"][2][n] - gref).norm()) / (float(gref.norm()) + 1e-8)
        assert err < 2e-3, (n, err)


def _composite_sine_embed(pos, F_=128, temperature=10000, scale=6.283185307179586):
    from memotr_amd.models.utils import _sine_dims
    dim_i = _sine_dims(F_, temperature, pos.device)
    e = (pos * scale)[..., None] / dim_i
    e = torch.stack((e[..., 0::2].sin(), e[..., 1::2].cos()), dim=-1)
    return torch.flatten(e, start_dim=-3)


@pytest.mark.parametrize("shape", [(1, 320, 4), (7, 4), (2, 33, 2)])
def test_sine_embed_values_and_gradient(shape):
    from memotr_amd.models.utils import pos_to_pos_embed
    g = torch.Generator().manual_seed(len(shape))
    pos = torch.rand(*shape, generator=g).cuda()
    up = torch.randn(*shape[:-1], shape[-1] * 128, generator=g).cuda()
    res = {}
    for name, fn in (("kernel", lambda p: pos_to_pos_embed(p, num_pos_feats=128)), ("torch", _composite_sine_embed)):
        x = pos.clone().requires_grad_(True)
        y = fn(x)
        (y * up).sum().backward()
        res[name] = (y.detach(), x.grad)
    assert res["kernel"][0].shape == res["torch"][0].shape
    torch.testing.assert_close(res["kernel"][0], res["torch"][0], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(res["kernel"][1], res["torch"][1], rtol=1e-4, atol=1e-4)


def test_inverse_sigmoid_and_refine_boxes_values_masks_and_gradients():
    from memotr_amd.utils.utils import inverse_sigmoid, inverse_sigmoid_reference, refine_boxes
    g = torch.Generator().manual_seed(9)
    x = torch.cat((torch.rand(500, generator=g), torch.tensor([0.0, 1.0, 1e-5, 1 - 1e-5, 5e-6, 1 - 5e-6, -0.2, 1.3, 0.5])))
    x = x.cuda()
    up = torch.randn(x.shape, generator=g).cuda()
    res = {}
    for name, fn in (("kernel", inverse_sigmoid), ("torch", inverse_sigmoid_reference)):
        v = x.clone().requires_grad_(True)
        y = fn(v)
        (y * up).sum().backward()
        res[name] = (y.detach(), v.grad)
    torch.testing.assert_close(res["kernel"][0], res["torch"][0], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(res["kernel"][1], res["torch"][1], rtol=1e-5, atol=1e-6)
    assert torch.isnan(inverse_sigmoid(torch.tensor([float("nan")]).cuda())).all()

    delta = (torch.randn(2, 40, 4, generator=g) * 2).cuda()
    ref = torch.rand(2, 40, 4, generator=g).cuda()
    ref[0, 0] = torch.tensor([0.0, 1.0, 5e-6, 0.5])
    up = torch.randn(2, 40, 4, generator=g).cuda()
    res = {}
    for name in ("kernel", "torch"):
        d, r = delta.clone().requires_grad_(True), ref.clone().requires_grad_(True)
        y = refine_boxes(d, r) if name == "kernel" else (d + inverse_sigmoid_reference(r)).sigmoid()
        (y * up).sum().backward()
        res[name] = (y.detach(), d.grad, r.grad)
    for a, b_ in zip(res["kernel"], res["torch"]):
        torch.testing.assert_close(a, b_, rtol=1e-5, atol=1e-6)
    # the decoder's use: a detached reference gets no gradient buffer
    d = delta.clone().requires_grad_(True)
    refine_boxes(d, ref).sum().backward()
    assert d.grad is not None


@pytest.mark.parametrize("rows,cols", [(310, 256), (300, 2048), (1, 4), (0, 8), (2048, 33), (320, 512)])
def test_colsum_equals_torch_sum(rows, cols):
    from memotr_amd.functions import clip_ops
    x = torch.randn(rows, cols, generator=torch.Generator().manual_seed(rows + cols)).cuda()
    got = clip_ops.colsum(x)
    torch.testing.assert_close(got, x.double().sum(0).float(), rtol=1e-5, atol=1e-4)
    out = torch.empty(3 * cols, device="cuda")
    clip_ops.colsum(x, out=out[cols:2 * cols])
    assert torch.equal(out[cols:2 * cols], got)                  # fixed summation order
    for tall in (clip_ops.COLSUM_MAX_ROWS + 1, 22323, 66969):     # two passes: per-chunk partials, then the partials
        big = torch.randn(tall, 40, generator=torch.Generator().manual_seed(tall)).cuda()
        torch.testing.assert_close(clip_ops.colsum(big), big.double().sum(0).float(), rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(clip_ops.colsum(x.t().contiguous().t()), x.sum(0)) # not contiguous: torch's reduction


def test_row_linear_matches_f_linear():
    import torch.nn.functional as F
    from memotr_amd.modules.linear import row_linear
    g = torch.Generator().manual_seed(4)
    for shape in ((1, 310, 256), (300, 256)):
        x = torch.randn(*shape, generator=g).cuda()
        w, b = torch.randn(512, 256, generator=g).cuda(), torch.randn(512, generator=g).cuda()
        up = torch.randn(*shape[:-1], 512, generator=g).cuda()
        res = {}
        for relu in (False, True):
            for name, fn in (("row", lambda *a: row_linear(*a, relu=relu)),
                             ("torch", lambda *a: F.relu(F.linear(*a)) if relu else F.linear(*a))):
                xi, wi, bi = (t.clone().requires_grad_(True) for t in (x, w, b))
                y = fn(xi, wi, bi)
                (y * up).sum().backward()
                res[name] = (y.detach(), xi.grad, wi.grad, bi.grad)
            for a, b_ in zip(res["row"], res["torch"]):
                torch.testing.assert_close(a, b_, rtol=1e-5, atol=1e-4)
    with torch.no_grad():                                         # nothing to differentiate: plain F.linear
        assert row_linear(x, w, b).grad_fn is None


@pytest.mark.parametrize("B,L,H,masked", [(1, 320, 8, 12), (2, 307, 8, 0), (1, 512, 8, 100), (1, 33, 4, 3), (1, 1, 8, 0),
                                          (2, 64, 8, 63)])
def test_self_attention_kernels_values_and_gradients(B, L, H, masked):
    """The hand-written attention (packed projections in, concatenated heads out) against softmax(q k^T / sqrt d) v."""
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(B * 1000 + L)
    E = H * 32
    qk = (torch.randn(B, L, 2 * E, generator=g) * 1.5).cuda()
    v = torch.randn(B, L, E, generator=g).cuda()
    mask = None
    if masked:
        mask = torch.zeros(B, L, dtype=torch.bool)
        mask[:, L - masked:] = True                      # padded track slots sit at the end
        mask[-1, 0] = L > 1 and masked < L - 1           # and one in front
        mask = mask.cuda()
    up = torch.randn(B, L, E, generator=g).cuda()
    assert clip_ops.self_attention_supported(qk, H)
    res = {}
    for name, fn in (("kernel", clip_ops.self_attention), ("torch", clip_ops.self_attention_reference)):
        a, b_ = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
        out = fn(a, b_, mask, H)
        (out * up).sum().backward()
        res[name] = (out.detach(), a.grad, b_.grad)
    for x, y, what in zip(res["kernel"], res["torch"], ("out", "grad q|k", "grad v")):
        torch.testing.assert_close(x, y, rtol=2e-4, atol=2e-5, msg=lambda m, what=what: f"{what}: {m}")
    if mask is not None:                                 # masked keys receive no gradient
        gk = res["kernel"][1][..., E:]
        assert float(gk[mask].abs().max()) == 0.0 and float(res["kernel"][2][mask].abs().max()) == 0.0
    assert not clip_ops.self_attention_supported(torch.zeros(1, 600, 2 * E).cuda(), H)      # L > 512: the library path
    assert not clip_ops.self_attention_supported(torch.zeros(1, 64, 2 * 64).cuda(), 8)      # head_dim 8


def test_decoder_self_attention_module_path_matches_nn_multiheadattention():
    """modules.attention.self_attention (packed in-projection node + attention kernels + row linear) against the
    module the reference calls, outputs and parameter gradients."""
    import torch.nn as nn
    from memotr_amd.modules.attention import self_attention
    torch.manual_seed(0)
    mha = nn.MultiheadAttention(256, 8, dropout=0.0, batch_first=True).cuda()
    tgt, pos = torch.randn(1, 320, 256).cuda(), torch.randn(1, 320, 256).cuda()
    mask = torch.zeros(1, 320, dtype=torch.bool).cuda()
    mask[0, 310:] = True
    up = torch.randn(1, 320, 256).cuda()
    res = {}
    for name in ("ours", "torch"):
        mha.zero_grad()
        x = tgt.clone().requires_grad_(True)
        if name == "ours":
            out = self_attention(mha, x + pos, x, key_padding_mask=mask)
        else:
            out = mha(x + pos, x + pos, x, key_padding_mask=mask, need_weights=False)[0]
        (out * up)[:, :310].sum().backward()
        res[name] = [out.detach()[:, :310], x.grad] + [p.grad.clone() for p in mha.parameters()]
    for a, b_ in zip(res["ours"], res["torch"]):
        torch.testing.assert_close(a, b_, rtol=2e-4, atol=2e-4)


def test_memory_attention_matches_nn_multiheadattention():
    """The query updater's attention (three different inputs, a handful of tracks) through the kernels."""
    import torch.nn as nn
    from memotr_amd.modules.attention import memory_attention
    torch.manual_seed(1)
    mha = nn.MultiheadAttention(256, 8, batch_first=True).cuda()
    for n in (1, 13, 40):
        q, k, v = (torch.randn(1, n, 256).cuda() for _ in range(3))
        up = torch.randn(1, n, 256).cuda()
        res = {}
        for name in ("ours", "torch"):
            mha.zero_grad()
            qi, ki, vi = (t.clone().requires_grad_(True) for t in (q, k, v))
            out = memory_attention(mha, qi, ki, vi) if name == "ours" else mha(qi, ki, vi, need_weights=False)[0]
            (out * up).sum().backward()
            res[name] = [out.detach(), qi.grad, ki.grad, vi.grad] + [p.grad.clone() for p in mha.parameters()]
        for a, b_ in zip(res["ours"], res["torch"]):
            torch.testing.assert_close(a, b_, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("shape", [(1, 320, 256), (3, 22323, 256), (7, 256), (1, 4097, 256)])
def test_add_layer_norm_values_and_gradients(shape):
    import torch.nn as nn
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(shape[-2])
    norm = nn.LayerNorm(256).cuda()
    with torch.no_grad():
        norm.weight.copy_(torch.rand(256, generator=g) + 0.5)
        norm.bias.copy_(torch.randn(256, generator=g) * 0.2)
    x = (torch.randn(*shape, generator=g) * 2 + 0.3).cuda()
    r = torch.randn(*shape, generator=g).cuda()
    up = torch.randn(*shape, generator=g).cuda()
    assert clip_ops.add_layer_norm_supported(x, r, norm)
    res = {}
    for name in ("kernel", "torch"):
        norm.zero_grad()
        a, b_ = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
        y = clip_ops.add_layer_norm(a, b_, norm) if name == "kernel" else norm(a + b_)
        (y * up).sum().backward()
        res[name] = (y.detach(), a.grad.clone(), b_.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone())
    names = ("y", "grad x", "grad res", "grad gamma", "grad beta")
    rows = x.numel() // 256
    for a, b_, what in zip(res["kernel"], res["torch"], names):
        tol = dict(rtol=1e-4, atol=1e-4 * max(1.0, rows ** 0.5 / 16)) if "gamma" in what or "beta" in what else \
            dict(rtol=1e-5, atol=2e-5)
        torch.testing.assert_close(a, b_, msg=lambda m, what=what: f"{what}: {m}", **tol)
    assert not clip_ops.add_layer_norm_supported(x[..., :128], r[..., :128], nn.LayerNorm(128).cuda())


# ----------------------------------------------------------------------------- device-side assignment
def test_device_assignment_is_identical_to_scipy_on_1000_random_matrices():
    """clipops_assign_f32 (one wavefront per problem, csrc/assign_core.h) against scipy.optimize.linear_sum_assignment
    -- the reference matcher's host call (models/matcher.py:122-124) -- on the matrices of tests/test_assign_core.py
    (matcher-shaped, transposed, square, heavy ties, duplicated rows / columns, constants): identical pairs."""
    from scipy.optimize import linear_sum_assignment
    from test_assign_core import cases
    from memotr_amd.functions import clip_ops
    n_ties = 0
    pending = []
    for c in cases():
        t = torch.from_numpy(c).cuda()
        # the (rows, cols) problem and, from the same memory, its transpose as a strided view
        batch = t[None]
        pending.append((c, clip_ops.assign(batch), clip_ops.assign(batch.transpose(1, 2))))
        n_ties += len(np.unique(c)) < c.size
    torch.cuda.synchronize()
    for c, (r, col, st), (rt, colt, stt) in pending:
        want_r, want_c = linear_sum_assignment(c)
        assert int(st[0]) == min(c.shape) == int(stt[0])
        assert np.array_equal(r[0].cpu().numpy(), want_r) and np.array_equal(col[0].cpu().numpy(), want_c), c.shape
        want_rt, want_ct = linear_sum_assignment(c.T)
        assert np.array_equal(rt[0].cpu().numpy(), want_rt) and np.array_equal(colt[0].cpu().numpy(), want_ct), c.shape
    assert n_ties > 300


def test_device_assignment_batches_the_layers_of_a_frame_and_flags_infeasible_costs():
    from scipy.optimize import linear_sum_assignment
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(5)
    cost = torch.randn(6, 310, 17, generator=g).cuda()               # six decoder layers x (queries, ground truths)
    r, c, st = clip_ops.assign(cost)
    assert st.tolist() == [17] * 6
    for layer in range(6):
        wr, wc = linear_sum_assignment(cost[layer].cpu().numpy())
        assert np.array_equal(r[layer].cpu().numpy(), wr) and np.array_equal(c[layer].cpu().numpy(), wc)
    bad = torch.tensor([[[float("inf"), float("inf")], [1.0, 2.0]]]).cuda()
    assert clip_ops.assign(bad)[2].tolist() == [-1]
    big = torch.zeros(1, 4, 3000).cuda()
    with pytest.raises(RuntimeError, match="CLIPOPS_ASSIGN_MAX_DIM"):
        clip_ops.assign(big)
    # 13 n_r + 29 n_c bytes of LDS: (1700, 1800) needs 74 KB -- past the default 64 KB of a launch, the launcher opts in
    wide = torch.rand(1, 1700, 1800, generator=g).cuda()
    r, c, st = clip_ops.assign(wide)
    wr, wc = linear_sum_assignment(wide[0].cpu().numpy())
    assert st.tolist() == [1700]
    assert np.array_equal(r[0].cpu().numpy(), wr) and np.array_equal(c[0].cpu().numpy(), wc)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 20, 24), (3, 8, 25, 42), (1, 5, 7, 3)], ids=["vec", "hw1050", "tiny"])
@pytest.mark.parametrize("with_res", [False, True], ids=["plain", "residual"])
def test_shift_relu_is_the_three_pass_chain_in_one(dtype, shape, with_res):
    """relu(conv_out + shift[c] (+ identity)) -- the frozen-BN bias add, the residual add and the ReLU of a ResNet
    bottleneck (torchvision Bottleneck.forward behind models/backbone.py:70-76 of the reference) -- in one kernel, in
    place: bit-equal to the torch chain in fp32 and bf16 (same rounding points), NaN kept, gradients those of the chain."""
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(sum(shape) + with_res)
    x0 = torch.randn(*shape, generator=g).to(dtype).cuda()
    x0[0, 0, 0, 0] = float("nan")
    shift = torch.randn(shape[1], generator=g).cuda()
    res0 = torch.randn(*shape, generator=g).to(dtype).cuda() if with_res else None

    def chain(x, res):      # what torch runs: conv bias add (bias cast by autocast for bf16), `+= identity`, relu
        y = x + shift.to(dtype)[None, :, None, None]
        if res is not None:
            y = y + res
        return torch.relu(y)

    xa = x0.clone().requires_grad_(True)
    ra = res0.clone().requires_grad_(True) if with_res else None
    want = chain(xa, ra)
    xb = x0.clone().requires_grad_(True)
    rb = res0.clone().requires_grad_(True) if with_res else None
    got = clip_ops.shift_relu_(xb * 1, shift, rb)        # (`* 1`: a fresh non-leaf tensor, like a convolution's output)
    assert torch.equal(torch.isnan(got), torch.isnan(want)) and bool(torch.isnan(got[0, 0, 0, 0]))
    assert torch.equal(torch.nan_to_num(got), torch.nan_to_num(want))
    go = torch.randn(*shape, generator=g).to(dtype).cuda()
    want.backward(go)
    got.backward(go)
    assert torch.equal(torch.nan_to_num(xa.grad), torch.nan_to_num(xb.grad))
    if with_res:
        assert torch.equal(torch.nan_to_num(ra.grad), torch.nan_to_num(rb.grad))


@pytest.mark.parametrize("rows,in_f,out_f", [(320, 256, 256), (310, 512, 256), (320, 256, 4), (10, 256, 512), (33, 40, 70),
                                             (1, 256, 256), (1024, 256, 384), (320, 1024, 256)])
@pytest.mark.parametrize("relu", [False, True], ids=["linear", "relu"])
def test_linear_backward_in_one_launch(rows, in_f, out_f, relu):
    """grad_x = G' W, grad_w = G'^T x, grad_b = colsum G' (G' = grad_y masked by y > 0 behind a fused ReLU): the backward
    of torch.nn.functional.linear on the decoder's query-sized linears, against float64 products of the same fp32
    inputs -- fp32 MFMA is an fmaf chain, so the error is fp32 round-off of a K-term sum."""
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(rows * 7 + in_f + out_f + relu)
    x = torch.randn(rows, in_f, generator=g).cuda()
    w = (torch.randn(out_f, in_f, generator=g) / in_f ** 0.5).cuda()
    b = torch.randn(out_f, generator=g).cuda()
    gy = torch.randn(rows, out_f, generator=g).cuda()
    y = torch.relu(x @ w.t() + b) if relu else None
    assert clip_ops.linear_bwd_usable(gy, x, w)
    gx, gw, gb = clip_ops.linear_bwd(gy, y, x, w)
    gm = gy.double() * ((y > 0).double() if relu else 1.0)
    wx, ww, wb = gm @ w.double(), gm.t() @ x.double(), gm.sum(0)
    for got, want, k in ((gx, wx, out_f), (gw, ww, rows), (gb, wb, rows)):
        scale = float(want.abs().max()) + 1e-6
        assert float((got.double() - want).abs().max()) <= 4e-7 * k ** 0.5 * scale + 1e-6, (rows, in_f, out_f)
    # the parts a caller does not need are not computed
    only_w = clip_ops.linear_bwd(gy, y, x, w, need_x=False, need_w=True, need_b=False)
    assert only_w[0] is None and only_w[2] is None and torch.equal(only_w[1], gw)


def test_row_linear_backward_uses_the_fused_kernel_and_matches_autograd(monkeypatch):
    from memotr_amd.modules.linear import row_linear
    g = torch.Generator().manual_seed(12)
    x0 = torch.randn(2, 160, 256, generator=g).cuda()
    lin = torch.nn.Linear(256, 256).cuda()
    go = torch.randn(2, 160, 256, generator=g).cuda()
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MEMOTR_FUSED_LINEAR_BWD", flag)
        x = x0.clone().requires_grad_(True)
        lin.zero_grad()
        y = row_linear(x, lin.weight, lin.bias, relu=True)
        y.backward(go)
        outs.append((x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()))
    for a, b in zip(*outs):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("rows,in_f,out_f", [(320, 256, 256), (310, 512, 256), (320, 256, 4), (10, 1024, 512), (33, 40, 70)])
@pytest.mark.parametrize("relu", [False, True], ids=["linear", "relu"])
def test_linear_forward_tile_kernel(rows, in_f, out_f, relu, monkeypatch):
    from memotr_amd.functions import clip_ops
    monkeypatch.setattr(clip_ops, "LINEAR_FWD_MAX_IN", 4096)       # (the kernel itself, beyond the shapes the model gives it)
    g = torch.Generator().manual_seed(rows + in_f + out_f)
    x = torch.randn(rows, in_f, generator=g).cuda()
    w = (torch.randn(out_f, in_f, generator=g) / in_f ** 0.5).cuda()
    b = torch.randn(out_f, generator=g).cuda()
    assert clip_ops.linear_fwd_usable(x, w, b)
    got = clip_ops.linear_fwd(x, w, b, relu)
    want = x.double() @ w.double().t() + b.double()
    if relu:
        want = torch.relu(want)
    assert float((got.double() - want).abs().max()) <= 4e-7 * in_f ** 0.5 * (float(want.abs().max()) + 1.0)
    if relu:
        assert float(got.min()) >= 0.0


@pytest.mark.parametrize("dtype", [torch.float32], ids=["f32"])
@pytest.mark.parametrize("rows,cols", [(5000, 300), (2049, 32), (22323, 2048)])
def test_relu_mask_and_bias_gradient_partials_in_one_pass(dtype, rows, cols):
    """g * (y > 0) (bit-equal to torch's threshold_backward) and its column sums (fp32, fixed order) from one pass."""
    from memotr_amd.functions import clip_ops
    gen = torch.Generator().manual_seed(rows + cols)
    g = torch.randn(rows, cols, generator=gen).to(dtype).cuda()
    y = torch.relu(torch.randn(rows, cols, generator=gen)).to(dtype).cuda()
    assert clip_ops.relu_bwd_colsum_usable(g, y)
    g2, gb = clip_ops.relu_bwd_colsum(g, y)
    want = torch.ops.aten.threshold_backward(g, y, 0.0)
    assert g2.dtype == dtype and torch.equal(g2, want)
    ref = want.double().sum(0)
    assert gb.dtype == torch.float32
    assert float((gb.double() - ref).abs().max()) <= 1e-5 * rows ** 0.5 * float(want.float().abs().max())


# ----------------------------------------------------------------------------- per-frame bookkeeping (ABI 10)
@pytest.mark.parametrize("n_tr,n_gt", [(0, 5), (7, 5), (40, 13), (3, 300), (300, 3)])
def test_track_ownership_matches_the_torch_formulation(n_tr, n_gt):
    """clipops_track_ownership_i64 against the compare / amax / any chain it replaces (models/criterion.py), duplicated
    ground-truth ids included (the LAST one owns), writing ``free`` into the head of a larger buffer."""
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(n_tr * 1000 + n_gt)
    gt_ids = torch.randint(0, max(n_gt // 2, 2), (n_gt,), generator=g).cuda()            # duplicates on purpose
    tr_ids = torch.randint(-1, max(n_gt // 2, 2) + 3, (n_tr,), generator=g).cuda()
    buf = torch.full((n_gt + 9,), 7.0, device="cuda")
    matched, free = clip_ops.track_ownership(tr_ids, gt_ids, free_out=buf[:n_gt])
    if n_tr:
        want_m, want_f = clip_ops.track_ownership_reference(tr_ids, gt_ids)
        assert torch.equal(matched, want_m) and torch.equal(free, want_f)
    else:
        assert matched.numel() == 0 and bool((free == 1).all())
    assert bool((buf[n_gt:] == 7.0).all())


@pytest.mark.parametrize("n_tr", [0, 11])
def test_focal_labels_match_the_torch_formulation(n_tr):
    from memotr_amd.functions import clip_ops
    g = torch.Generator().manual_seed(n_tr)
    n_layers, nd, n_gt, K = 6, 300, 9, 8
    late = torch.tensor([False, True, True, False, True, True]).cuda()
    gt_labels = torch.randint(0, K, (n_gt,), generator=g).cuda()
    pairs = [(l, int(q), int(t)) for l in range(n_layers)
             for q, t in zip(torch.randperm(nd, generator=g)[:n_gt], torch.randperm(n_gt, generator=g))]
    lay, q, t = (torch.tensor(c).cuda() for c in zip(*pairs))
    matched = torch.randint(-1, n_gt, (n_tr,), generator=g).cuda() if n_tr else None
    got = clip_ops.focal_labels(lay, q, t, gt_labels, matched, late, nd, n_tr, K)
    want = clip_ops.focal_labels_reference(lay, q, t, gt_labels, matched, late, nd, n_tr, K)
    assert got.shape == (n_layers, nd + n_tr) and torch.equal(got, want)
    empty = torch.zeros((0,), dtype=torch.long, device="cuda")
    got0 = clip_ops.focal_labels(empty, empty, empty, gt_labels, matched, late, nd, n_tr, K)
    assert torch.equal(got0, clip_ops.focal_labels_reference(empty, empty, empty, gt_labels, matched, late, nd, n_tr, K))

