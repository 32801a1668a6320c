"""CPU: libclip_ops_hip.so loads and exports exactly what include/clip_ops_hip.h declares; argument validation is
host-side and works without a device; CPU tensors take the element-wise formulation."""
import ctypes
import os
import re

import torch

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "clip_ops_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clipops_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(clip_lib):
    raw = ctypes.CDLL(clip_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 7
    for s in syms:
        assert hasattr(raw, s), f"libclip_ops_hip.so does not export {s}"
    assert sorted(clip_lib.SYMBOLS) == syms
    assert clip_lib.lib.clipops_abi_version() == clip_lib.ABI_VERSION


def test_argument_errors_are_reported_without_a_device(clip_lib):
    lib = clip_lib.lib
    dummy = ctypes.c_void_p(16)
    assert lib.clipops_match_cost_f32(None, 4, 1, dummy, 4, 4, dummy, dummy, 1, 2, 1, 3, 1.0, 1.0, 1.0, dummy, None) != 0
    assert b"null" in lib.clipops_last_error()
    assert lib.clipops_focal_fwd_f32(dummy, 4, 1, dummy, 1, 2, 0, 0.25, 2.0, dummy, None) != 0
    assert lib.clipops_pair_box_loss_fwd_f32(dummy, dummy, dummy, 1, 0, dummy, None, None, -1, dummy, dummy, None) != 0
    # empty problems are fine and launch nothing
    assert lib.clipops_match_cost_f32(dummy, 4, 1, dummy, 4, 4, dummy, dummy, 6, 300, 1, 0, 1.0, 1.0, 1.0, dummy, None) == 0
    assert lib.clipops_pair_box_loss_fwd_f32(dummy, dummy, dummy, 1, 0, dummy, None, None, 0, dummy, dummy, None) == 0


def test_cpu_tensors_use_the_elementwise_formulation():
    from memotr_amd.functions import clip_ops
    assert not clip_ops.fused(torch.zeros(3))
    boxes = torch.rand(2, 1, 5, 4) * 0.4 + 0.3
    lay, q = torch.tensor([0, 1, 1]), torch.tensor([4, 0, 2])
    tgt = torch.rand(3, 4) * 0.4 + 0.3
    l1, gl = clip_ops.pair_box_loss_reference(boxes, lay, q, 0, tgt)
    assert l1.shape == gl.shape == (3,)
    torch.testing.assert_close(l1[0], (boxes[0, 0, 4] - tgt[0]).abs().sum())
    labels = torch.tensor([[0, 1, 1, 1, 1], [1, 1, 0, 1, 1]])
    loss = clip_ops.focal_loss_per_layer_reference(torch.randn(2, 5, 1), labels)
    assert loss.shape == (2,) and bool((loss > 0).all())
