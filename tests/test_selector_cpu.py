"""CPU: the kernel-selection transition function (memotr_amd/csrc/msda_select.h) through the C ABI -- pure host logic,
no device call.  Levels, backward: 0 small windows, 1 large windows, 2 no windows; forward: 0 windows, 1 gather."""
import pytest


@pytest.fixture()
def nxt(hip_lib):
    saved = {k: hip_lib.get_option(k) for k in ("sel_up0", "sel_up1", "sel_down1", "sel_down2", "sel_fwd_up", "sel_fwd_down")}
    for k, v in (("sel_up0", 5), ("sel_up1", 100), ("sel_down1", 2), ("sel_down2", 60), ("sel_fwd_up", 50),
                 ("sel_fwd_down", 20)):
        hip_lib.set_option(k, v)
    yield lambda kind, level, off, inner=0: hip_lib.lib.msda_selector_next(kind, level, off, inner)
    for k, v in saved.items():
        hip_lib.set_option(k, v)


def test_backward_levels_follow_the_off_window_share_with_hysteresis(nxt):
    assert nxt(1, 0, 0) == 0 and nxt(1, 0, 5) == 0       # up to 0.5 % of the corners outside: small windows stay
    assert nxt(1, 0, 6) == 1                             # more: large windows
    assert nxt(1, 1, 50, 50) == 1                        # large windows hold 95 %: stay
    assert nxt(1, 1, 0, 3) == 1 and nxt(1, 1, 0, 1) == 0     # back only when the SMALL window would lose < 0.2 %
    assert nxt(1, 1, 101, 500) == 2                      # more than 10 % outside the large window: no windows
    assert nxt(1, 2, 80) == 2 and nxt(1, 2, 59) == 1     # a probe brings windows back below 6 %
    # no level is left and re-entered at the same share (the thresholds do not overlap)
    for off in range(0, 1001, 7):
        up = nxt(1, 0, off)
        if up == 1:
            assert nxt(1, 1, 0, off) == 1
        top = nxt(1, 1, off, 1000)
        if top == 2:
            assert nxt(1, 2, off) == 2


def test_forward_has_two_levels(nxt):
    assert nxt(0, 0, 50) == 0 and nxt(0, 0, 51) == 1
    assert nxt(0, 1, 20) == 1 and nxt(0, 1, 19) == 0


def test_selector_options_and_call_site_round_trip(hip_lib):
    hip_lib.set_option("sel_level", 2)
    assert hip_lib.get_option("sel_level") == 2
    hip_lib.set_option("sel_level", -1)                  # the one option that takes -1 (follow the data)
    assert hip_lib.get_option("sel_level") == -1
    with pytest.raises(ValueError):
        hip_lib.set_option("auto_select", -1)
    hip_lib.set_call_site(12345)
    hip_lib.set_call_site(0)
    level, off, inner = hip_lib.selector_last()
    assert level >= 0 and off <= 1.0 and inner <= 1.0
