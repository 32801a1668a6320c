"""GPU: kernel selection from the measured off-window share (memotr_amd/csrc/msda_select.h).

Results never depend on the level -- every level is checked against the C oracle -- and the level follows the data:
sampling points near their queries keep the windowed kernels, uniformly random locations (the reference's own test
distribution, models/ops/test.py:33) move the backward to the kernel without windows and the forward to the gather
kernel within a few calls."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_SITE = [7000]


@pytest.fixture()
def site(hip_lib):
    """A fresh (call site, geometry) record per test."""
    _SITE[0] += 1
    hip_lib.set_call_site(_SITE[0])
    yield _SITE[0]
    hip_lib.set_call_site(0)
    for k, v in (("fwd_variant", 0), ("bwd_variant", 0), ("sel_level", -1), ("auto_select", 1)):
        hip_lib.set_option(k, v)


def _inputs(dist, off_scale=1.0, seed=5):
    from memotr_amd.MultiScaleDeformableAttention import tag_host_shapes
    from memotr_amd.synth import make_inputs
    x = make_inputs(height=320, width=448, dist=dist, off_scale=off_scale, device="cuda", seed=seed)
    tag_host_shapes(x["shapes"], x["shapes_list"])
    return x


def _oracle(x):
    from oracle import msda_oracle as oracle
    c = {k: v.detach().cpu().numpy() for k, v in x.items() if isinstance(v, torch.Tensor)}
    out = oracle.forward(c["value"], c["shapes"], c["level_start"], c["loc"], c["attn"])
    return (out,) + tuple(oracle.backward(c["value"], c["shapes"], c["level_start"], c["loc"], c["attn"], c["grad_out"]))


def _check_bwd(got, want, what):
    np.testing.assert_allclose(got[0].cpu().numpy(), want[1], rtol=1e-4, atol=1e-4, err_msg=what + " grad_value")
    np.testing.assert_allclose(got[1].cpu().numpy(), want[2], rtol=1e-4, atol=2e-3, err_msg=what + " grad_loc")
    np.testing.assert_allclose(got[2].cpu().numpy(), want[3], rtol=1e-4, atol=2e-4, err_msg=what + " grad_attn")


def test_every_backward_level_matches_the_oracle(hip_lib, site):
    from memotr_amd import MultiScaleDeformableAttention as MSDA
    for dist, osc in (("encoder_like", 1.0), ("encoder_like", 4.0), ("uniform", 1.0)):
        x = _inputs(dist, osc)
        want = _oracle(x)
        args = (x["value"], x["shapes"], x["level_start"], x["loc"], x["attn"], x["grad_out"], 64)
        for level, kernel in ((0, "tile_bins"), (1, "tile_bins"), (2, "rows")):
            hip_lib.set_option("sel_level", level)
            for _ in range(3):          # (three launches: both parities of the statistics record are exercised)
                got = MSDA.ms_deform_attn_backward(*args)
            assert kernel in hip_lib.last_kernel(), (level, hip_lib.last_kernel())
            _check_bwd(got, want, f"{dist} x{osc} level {level}")


def test_backward_level_follows_the_data(hip_lib, site):
    from memotr_amd import MultiScaleDeformableAttention as MSDA
    near, far = _inputs("encoder_like"), _inputs("uniform")
    want_far = _oracle(far)

    def run(x, n):
        levels = []
        for _ in range(n):
            got = MSDA.ms_deform_attn_backward(x["value"], x["shapes"], x["level_start"], x["loc"], x["attn"],
                                               x["grad_out"], 64)
            torch.cuda.synchronize()
            levels.append(hip_lib.selector_last()[0])
        return got, levels

    _, levels = run(near, 6)
    assert levels == [0] * 6 and "tile_bins" in hip_lib.last_kernel(), (levels, hip_lib.last_kernel())
    share = hip_lib.selector_last()[1]
    assert 0.0 <= share < 0.005, share                  # measured: (almost) nothing leaves a 6-pixel margin
    got, levels = run(far, 12)                          # same geometry, same record: the locations changed
    assert levels[-1] == 2 and sorted(levels) == levels, levels          # 0 -> 1 -> 2, monotone
    assert "tile_bins" not in hip_lib.last_kernel(), hip_lib.last_kernel()
    assert hip_lib.selector_last()[1] > 0.1              # (the share outside the LARGE window, from the last probe)
    _check_bwd(got, want_far, "uniform, after the switch")
    hip_lib.set_option("auto_select", 0)                # switched off: level 0 whatever the data say
    run(far, 3)
    assert "tile_bins" in hip_lib.last_kernel()


def test_forward_level_follows_the_data(hip_lib, site):
    from memotr_amd import MultiScaleDeformableAttention as MSDA
    near, far = _inputs("encoder_like"), _inputs("uniform")
    for x, kernel in ((near, "win"), (far, "gather")):
        want = _oracle(x)[0]
        for _ in range(8):
            out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["level_start"], x["loc"], x["attn"], 64)
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=2e-5)
        assert kernel in hip_lib.last_kernel(), hip_lib.last_kernel()
