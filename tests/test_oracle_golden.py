"""CPU: the oracle restatements against the golden vectors produced by the reference itself.

Golden vectors: tests/golden/msda_*.npz (generator: tests/golden/gen_golden.py, which imports
models/ops/functions/ms_deform_attn_func.py:44-64 from the reference and differentiates it).
F1 is the reference's own test case (models/ops/test.py:21-36).
"""
import numpy as np
import pytest
import torch

from conftest import golden_cases, load_golden
from oracle import msda_oracle as oracle

CASES = golden_cases()


def tol(dtype):
    # float64: ~ulp level; float32: the two formulations round differently (grid_sample
    # un-normalises 2*loc-1, the kernel uses loc*size-0.5) -> 1e-5 abs on O(1) data.
    return dict(rtol=1e-9, atol=1e-12) if dtype == np.float64 else dict(rtol=2e-4, atol=2e-5)


def test_golden_set_is_complete():
    assert len(CASES) == 12
    assert "msda_F1_testpy_f64" in CASES and "msda_F1_testpy_f32" in CASES


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_forward_matches_reference(name):
    g = load_golden(name)
    out = oracle.forward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"])
    np.testing.assert_allclose(out, g["out"], **tol(g["value"].dtype))


@pytest.mark.parametrize("name", CASES)
def test_c_oracle_backward_matches_reference_autograd(name):
    g = load_golden(name)
    gv, gl, ga = oracle.backward(g["value"], g["shapes"], g["level_start"], g["loc"], g["attn"], g["grad_out"])
    t = tol(g["value"].dtype)
    np.testing.assert_allclose(gv, g["grad_value"], **t)
    # grad_loc is scaled by the level size (up to 14 here): loosen absolutely, not relatively
    t_loc = dict(rtol=t["rtol"], atol=t["atol"] * 20)
    np.testing.assert_allclose(gl, g["grad_loc"], **t_loc)
    np.testing.assert_allclose(ga, g["grad_attn"], **t_loc)


@pytest.mark.parametrize("name", CASES)
def test_grid_sample_restatement_matches_reference(name):
    """The torch-CPU port used as cpu_baseline is the reference's formula bit for bit."""
    g = load_golden(name)
    shapes = [tuple(int(x) for x in hw) for hw in g["shapes"]]
    out = oracle.grid_sample_forward(torch.from_numpy(g["value"]), shapes, torch.from_numpy(g["loc"]),
                                     torch.from_numpy(g["attn"]))
    np.testing.assert_array_equal(out.numpy(), g["out"])


def test_reference_test_py_tolerances_hold_for_oracle():
    """models/ops/test.py:40 (double, allclose defaults) and :56 (float, rtol 1e-2 / atol 1e-3)."""
    g64, g32 = load_golden("msda_F1_testpy_f64"), load_golden("msda_F1_testpy_f32")
    o64 = oracle.forward(g64["value"], g64["shapes"], g64["level_start"], g64["loc"], g64["attn"])
    o32 = oracle.forward(g32["value"], g32["shapes"], g32["level_start"], g32["loc"], g32["attn"])
    assert np.allclose(o64, g64["out"])
    assert np.allclose(o32, g32["out"], rtol=1e-2, atol=1e-3)


def test_indices_agree_between_precisions_away_from_pixel_edges():
    g = load_golden("msda_F2_memotr_heads_f64")
    h64, w64, g64 = oracle.indices(g["shapes"], g["loc"])
    h32, w32, g32 = oracle.indices(g["shapes"], g["loc"].astype(np.float32))
    # float32 rounding of loc may flip a floor only if loc*size-0.5 sits within 1e-4 of an integer
    L = g["shapes"].shape[0]
    size_h = g["shapes"][:, 0].reshape(1, 1, 1, L, 1)
    size_w = g["shapes"][:, 1].reshape(1, 1, 1, L, 1)
    fy = g["loc"][..., 1] * size_h - 0.5
    fx = g["loc"][..., 0] * size_w - 0.5
    safe = (np.abs(fy - np.round(fy)) > 1e-4) & (np.abs(fx - np.round(fx)) > 1e-4)
    assert safe.mean() > 0.99
    assert np.array_equal(h64[safe], h32[safe]) and np.array_equal(w64[safe], w32[safe])
    assert np.array_equal(g64[safe], g32[safe])


def test_gate_and_zero_padding_semantics():
    """Points outside (-1,H)x(-1,W) contribute nothing; border corners are zero padded (.cuh:288, :53-72)."""
    shapes = np.array([[2, 3]], dtype=np.int64)
    lsi = np.array([0], dtype=np.int64)
    value = np.ones((1, 6, 1, 1), dtype=np.float64)
    attn = np.ones((1, 4, 1, 1, 1), dtype=np.float64)
    loc = np.zeros((1, 4, 1, 1, 1, 2), dtype=np.float64)
    loc[0, 0, 0, 0, 0] = (0.5, 0.5)        # interior: full weight
    loc[0, 1, 0, 0, 0] = (0.0, 0.5)        # x = -0.5 px: half of the weight falls on padding
    loc[0, 2, 0, 0, 0] = (-0.2, 0.5)       # x = -1.1 px: gated out
    loc[0, 3, 0, 0, 0] = (1.0 + 1e-9, 0.5)  # x just right of W-0.5: gate w_im < W still true
    out = oracle.forward(value, shapes, lsi, loc, attn).reshape(-1)
    np.testing.assert_allclose(out, [1.0, 0.5, 0.0, 0.5], atol=1e-8)


def test_index_claim_is_scoped_to_the_uncontracted_reading_of_the_reference():
    """.cuh:285-286 is `loc * size - 0.5`; the oracle and the HIP kernels round the product first (two operations).
    Whether the reference binary fuses them into one multiply-add is UNDECIDED without nvcc (models/ops/setup.py:41-46
    sets no -fmad=false, but the literal 0.5 is a double: a float multiply feeding a double subtract, which a compiler
    may or may not narrow back to a float fma).  How far apart are the two readings at the BASELINE encoder shape?
      * sampling distributions with any noise in them (both bench distributions): zero of 2.86 M indices flip;
      * locations EXACTLY on pixel centres (the encoder's reference points with zero / whole-pixel offsets -- a set of
        measure zero, but the one a hand-made test would pick): ~0.6 % of the points floor to the neighbouring pixel,
        with the fractional weight at the other end of [0, 1), i.e. the same interpolated value to one ulp.
    The bit-exact index claim (include/msda_hip.h) is a claim against the uncontracted source."""
    import torch
    from memotr_amd.synth import encoder_reference_points, make_inputs, pyramid_shapes, star_offsets, valid_ratios
    from oracle import msda_oracle as oracle
    for dist in ("encoder_like", "uniform"):
        x = make_inputs(dist=dist)
        loc, sh = x["loc"].numpy(), x["shapes"].numpy()
        h, w, g = oracle.indices(sh, loc)
        hf, wf, gf = oracle.indices_fma(sh, loc)
        assert h.size == 22323 * 8 * 16
        assert int(((h != hf) | (w != wf)).sum()) == 0 and int((g != gf).sum()) == 0, dist
    shapes = pyramid_shapes(800, 1333)
    ref = encoder_reference_points(shapes, valid_ratios(800, 1333, shapes))
    wh = torch.tensor([[w_, h_] for h_, w_ in shapes], dtype=torch.float32)
    off = star_offsets(8, 4, 4)[None]
    loc = (ref[:, None, :, None, :] + off / wh[None, None, :, None, :])[None].contiguous().numpy().astype(np.float32)
    sh = np.asarray(shapes, dtype=np.int64)
    h, w, g = oracle.indices(sh, loc)
    hf, wf, gf = oracle.indices_fma(sh, loc)
    flips = int(((h != hf) | (w != wf)).sum())
    assert 0 < flips < 0.02 * h.size, flips          # measured 17,063 of 2,857,344 (0.6 %)
    assert int((np.abs(h - hf) > 1).sum()) == 0 and int((np.abs(w - wf) > 1).sum()) == 0      # always the neighbour
    assert int((g != gf).sum()) == 0                 # the (-1, size) gate is the same in both readings here
