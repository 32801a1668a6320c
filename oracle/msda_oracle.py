"""oracle/msda_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Python face of the CPU oracle for the multi-scale deformable attention path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; nothing under ``memotr_amd/`` does.

Two independent restatements live here:

* ``forward`` / ``backward`` / ``indices`` -- ctypes wrappers over
  ``oracle/msda_oracle.c`` (scalar loops following the reference kernels
  ``models/ops/src/cuda/ms_deform_im2col_cuda.cuh:33-159,237-403``); numpy in,
  numpy out, float32 or float64.
* ``grid_sample_forward`` -- the reference's *pure-PyTorch* statement of the same
  operator (``models/ops/functions/ms_deform_attn_func.py:44-64``: per level
  ``F.grid_sample(bilinear, zeros, align_corners=False)`` then a weighted sum),
  restated for torch CPU tensors.  This is the "reference CPU fallback" that
  ``BASELINE.json`` asks to be timed next to the GPU kernel (kind = "port").

Parity is pinned: both restatements are checked against ``tests/golden/*.npz``,
vectors produced by importing the reference itself (tests/golden/gen_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile oracle/msda_oracle.c with gcc (seconds). Returns the .so path."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "libmsda_oracle.so"])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, level_start, loc, attn):
    dt = value.dtype
    if dt not in (np.float32, np.float64):
        raise TypeError(f"oracle supports float32/float64, got {dt}")
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    level_start = np.ascontiguousarray(level_start, dtype=np.int64)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and attn.shape == (N, Lq, M, L, P)
    assert shapes.shape == (L, 2) and level_start.shape == (L,)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    suf = "f32" if dt == np.float32 else "f64"
    return value, shapes, level_start, loc, attn, (N, S, M, D, L, Lq, P), suf


def forward(value, shapes, level_start, loc, attn) -> np.ndarray:
    """out (N, Lq, M*D); semantics of ms_deform_attn_cuda_forward (.cu:20-80)."""
    value, shapes, level_start, loc, attn, dims, suf = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    out = np.empty((N, Lq, M * D), dtype=value.dtype)
    fn = getattr(_load(), f"msda_oracle_forward_{suf}")
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(attn),
       *[ctypes.c_int(x) for x in (N, S, M, D, L, Lq, P)], _ptr(out))
    return out


def backward(value, shapes, level_start, loc, attn, grad_out):
    """(grad_value, grad_loc, grad_attn); semantics of ms_deform_attn_cuda_backward (.cu:83-153)."""
    value, shapes, level_start, loc, attn, dims, suf = _prep(value, shapes, level_start, loc, attn)
    N, S, M, D, L, Lq, P = dims
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype).reshape(N, Lq, M * D)
    gv = np.zeros_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(_load(), f"msda_oracle_backward_{suf}")
    fn(_ptr(value), _ptr(shapes), _ptr(level_start), _ptr(loc), _ptr(attn), _ptr(grad_out),
       *[ctypes.c_int(x) for x in (N, S, M, D, L, Lq, P)], _ptr(gv), _ptr(gl), _ptr(ga))
    return gv, gl, ga


def indices(shapes, loc):
    """Integer side of the sampling arithmetic: (h_low, w_low, gate), each (N,Lq,M,L,P).

    h_low/w_low are floor(loc*size - 0.5) computed in loc's dtype with the
    reference's rounding points; gate is the (-1,H)x(-1,W) test of .cuh:288.
    """
    loc = np.ascontiguousarray(loc)
    suf = "f32" if loc.dtype == np.float32 else "f64"
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    N, Lq, M, L, P, _ = loc.shape
    h = np.empty((N, Lq, M, L, P), dtype=np.int32)
    w = np.empty_like(h)
    g = np.empty((N, Lq, M, L, P), dtype=np.uint8)
    fn = getattr(_load(), f"msda_oracle_indices_{suf}")
    fn(_ptr(shapes), _ptr(loc), *[ctypes.c_int(x) for x in (N, M, L, Lq, P)], _ptr(h), _ptr(w), _ptr(g))
    return h, w, g


def indices_fma(shapes, loc):
    """The same with `loc * size - 0.5` contracted into ONE fused multiply-add (nvcc's default for the reference build,
    models/ops/setup.py:41-46 passes no -fmad=false): float32 only.  For counting how many points the two readings of
    .cuh:285-286 put on different pixels."""
    loc = np.ascontiguousarray(loc, dtype=np.float32)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    N, Lq, M, L, P, _ = loc.shape
    h = np.empty((N, Lq, M, L, P), dtype=np.int32)
    w = np.empty_like(h)
    g = np.empty((N, Lq, M, L, P), dtype=np.uint8)
    _load().msda_oracle_indices_fma_f32(_ptr(shapes), _ptr(loc), *[ctypes.c_int(x) for x in (N, M, L, Lq, P)], _ptr(h),
                                        _ptr(w), _ptr(g))
    return h, w, g


def grid_sample_forward(value, shapes_hw, loc, attn):
    """torch-CPU restatement of the reference fallback (ms_deform_attn_func.py:44-64).

    ``shapes_hw`` is a python list of (H, W).  Differentiable through autograd,
    which is what the reference's own gradcheck compares the CUDA backward to.
    """
    import torch
    import torch.nn.functional as F

    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    per_level = value.split([h * w for h, w in shapes_hw], dim=1)
    grids = 2 * loc - 1
    sampled = []
    for lvl, (h, w) in enumerate(shapes_hw):
        v = per_level[lvl].flatten(2).transpose(1, 2).reshape(N * M, D, h, w)
        g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    a = attn.transpose(1, 2).reshape(N * M, 1, Lq, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(N, M * D, Lq)
    return out.transpose(1, 2).contiguous()
