"""CPU: the C-ABI shared library loads and exports exactly what include/msda_hip.h declares.

No compute is launched here (no GPU in the build container); argument validation is host-side and
must work without a device.
"""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "msda_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msda_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("msda_forward_f32", "msda_forward_f64", "msda_forward_bf16", "msda_backward_f32",
              "msda_backward_f64", "msda_backward_bf16", "msda_sample_indices_f32", "msda_abi_version",
              "msda_last_error", "msda_last_kernel", "msda_set_option", "msda_get_option"):
        assert s in syms


def test_library_exports_every_declared_symbol(hip_lib):
    raw = ctypes.CDLL(hip_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(raw, s), f"libmsda_hip.so does not export {s}"
    assert sorted(hip_lib.SYMBOLS) == declared_symbols()


def test_abi_version_and_options(hip_lib):
    assert hip_lib.lib.msda_abi_version() == hip_lib.ABI_VERSION
    assert hip_lib.get_option("fwd_variant") == 0
    hip_lib.set_option("fwd_variant", 1)
    assert hip_lib.get_option("fwd_variant") == 1
    hip_lib.set_option("fwd_variant", 0)
    with pytest.raises(ValueError):
        hip_lib.set_option("no_such_knob", 1)


def test_argument_errors_are_reported_without_a_device(hip_lib):
    lib = hip_lib.lib
    dummy = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    rc = lib.msda_forward_f32(None, dummy, dummy, dummy, dummy, 1, 4, 1, 32, 1, 1, 1, dummy, None, None)
    assert rc == -1 and "null" in hip_lib.last_error()
    rc = lib.msda_forward_f32(dummy, dummy, dummy, dummy, dummy, 1, 4, 1, 0, 1, 1, 1, dummy, None, None)
    assert rc == -1
    # > 2^31 elements: same envelope as the reference's int indexing (.cuh:255-270), but reported
    rc = lib.msda_forward_f32(dummy, dummy, dummy, dummy, dummy, 64, 1 << 20, 8, 32, 4, 1, 4, dummy, None, None)
    assert rc == -2 and "2^31" in hip_lib.last_error()
    rc = lib.msda_backward_f32(dummy, dummy, dummy, dummy, dummy, dummy, 1, 4, 1, 32, 1, 1, 1, None, dummy, dummy,
                               0, None, None)
    assert rc == -1


def test_cpu_tensors_are_rejected_like_the_reference():
    """ms_deform_attn.h:38 -- AT_ERROR("Not implemented on the CPU")."""
    import torch
    from memotr_amd import MultiScaleDeformableAttention as MSDA
    value = torch.zeros(1, 6, 1, 4)
    shapes = torch.tensor([[2, 3]])
    lsi = torch.tensor([0])
    loc = torch.zeros(1, 2, 1, 1, 1, 2)
    attn = torch.zeros(1, 2, 1, 1, 1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, attn, 64)
    with pytest.raises(RuntimeError, match="contiguous"):
        MSDA.ms_deform_attn_forward(torch.zeros(1, 1, 6, 4).transpose(2, 3), shapes, lsi, loc, attn, 64)


def test_dropin_module_name_resolves():
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, "memotr_amd", "dropin"))
    try:
        mod = importlib.import_module("MultiScaleDeformableAttention")
        assert callable(mod.ms_deform_attn_forward) and callable(mod.ms_deform_attn_backward)
    finally:
        sys.path.pop(0)
        sys.modules.pop("MultiScaleDeformableAttention", None)


def test_algorithmic_byte_counts_match_the_survey():
    """80,005,632 B (+96 B of int64 metadata) forward, 137,152,512 B backward at the encoder shape;
    decoder-shape calls cannot read more of `value` than the corners they touch."""
    from memotr_amd.synth import algorithmic_bytes, level_start_index, pyramid_shapes
    shapes = pyramid_shapes(800, 1333)
    assert shapes == [(100, 168), (50, 84), (25, 42), (13, 21)]
    assert level_start_index(shapes) == [0, 16800, 21000, 22050]
    S = sum(h * w for h, w in shapes)
    assert S == 22323
    assert algorithmic_bytes(1, S, S, 8, 32, 4, 4) == 80_005_632 + 96
    assert algorithmic_bytes(1, S, S, 8, 32, 4, 4, backward=True) == 137_152_512 + 96
    assert algorithmic_bytes(1, S, 300, 8, 32, 4, 4) == 20_428_800 + 96
    assert pyramid_shapes(720, 1280) == [(92, 160), (46, 80), (23, 40), (12, 20)]
