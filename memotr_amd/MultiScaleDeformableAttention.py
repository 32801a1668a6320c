"""Drop-in for the reference's compiled extension module ``MultiScaleDeformableAttention``.

Same two entry points, argument order and error behaviour as the pybind module built from
``models/ops/src/vision.cpp:13-16`` (signatures ``models/ops/src/ms_deform_attn.h:20-61``,
host logic ``models/ops/src/cuda/ms_deform_attn_cuda.cu:20-153``), backed by the gfx950 HIP
kernels behind the C ABI of ``include/msda_hip.h``.

Differences, all deliberate:
  * launch failures raise ``RuntimeError`` (the reference only ``printf``s them,
    ``ms_deform_im2col_cuda.cuh:948-952``);
  * ``torch.bfloat16`` value/grad_output with fp32 locations/weights is accepted (extension);
  * ``im2col_step`` only participates in the divisibility check -- the kernels take the whole
    batch in one launch (the reference loops over ``batch / im2col_step`` chunks, same result).
"""
from __future__ import annotations

import torch

from . import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64", torch.bfloat16: "bf16"}


def _check_inputs(named):
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
    for name, t in named:
        if not t.is_cuda:
            # ms_deform_attn.h:38,60 -- the reference has no CPU implementation either
            raise RuntimeError("Not implemented on the CPU" if name == "value" else f"{name} must be a CUDA tensor")


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5:
        raise RuntimeError("expected value (N,S,M,D), sampling_loc (N,Lq,M,L,P,2), attn_weight (N,Lq,M,L,P)")
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if tuple(sampling_loc.shape) != (N, Lq, M, L, P, 2) or tuple(attn_weight.shape) != (N, Lq, M, L, P):
        raise RuntimeError("sampling_loc / attn_weight shapes do not match value / spatial_shapes")
    if tuple(spatial_shapes.shape) != (L, 2) or tuple(level_start_index.shape) != (L,):
        raise RuntimeError("spatial_shapes must be (L,2) and level_start_index (L,)")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 tensors")
    step = min(N, int(im2col_step))
    if N > 0 and (step <= 0 or N % step != 0):
        raise RuntimeError(f"batch({N}) must divide im2col_step({step})")
    return N, S, M, D, L, Lq, P


def _suffix(value, sampling_loc, attn_weight):
    suf = _SUFFIX.get(value.dtype)
    if suf is None:
        raise RuntimeError(f"ms_deform_attn: unsupported value dtype {value.dtype}")
    want = torch.float32 if suf == "bf16" else value.dtype
    if sampling_loc.dtype != want or attn_weight.dtype != want:
        raise RuntimeError(f"ms_deform_attn: sampling_loc/attn_weight must be {want} for value dtype {value.dtype}")
    return suf


def host_shapes(spatial_shapes):
    """HOST copy of ``spatial_shapes`` as a numpy int64 array kept on the tensor object itself.

    The level-aware kernels plan their launch from it.  Callers that know the pyramid as python ints
    attach it up front (``tag_host_shapes``) so no device->host read ever happens on the hot path;
    otherwise the first call with a given tensor *object* pays one synchronising ``.cpu()`` (the copy is
    stored as an attribute of that object together with its version counter, so it can never go stale
    or be confused with another tensor that reuses the same device address).
    """
    tagged = getattr(spatial_shapes, "_msda_host", None)
    if tagged is not None and tagged[1] == spatial_shapes._version:
        return tagged[0]
    import numpy as np
    arr = np.ascontiguousarray(spatial_shapes.detach().cpu().numpy(), dtype=np.int64)
    spatial_shapes._msda_host = (arr, spatial_shapes._version)
    return arr


def tag_host_shapes(spatial_shapes, shapes_list):
    """Attach the python-side pyramid [(H, W), ...] to a device ``spatial_shapes`` tensor."""
    import numpy as np
    arr = np.ascontiguousarray(np.asarray(shapes_list, dtype=np.int64).reshape(-1, 2))
    spatial_shapes._msda_host = (arr, spatial_shapes._version)
    return spatial_shapes


def _host_ptr(spatial_shapes, eligible: bool):
    """(keep-alive, pointer) of the host shapes -- only fetched for calls the level-aware kernels can take
    (fp32, D = 32, one query per pyramid pixel); every other call passes NULL and never synchronises."""
    if not eligible:
        return None, None
    arr = host_shapes(spatial_shapes)
    return arr, arr.ctypes.data


# ---- kernel selection (include/msda_hip.h, "kernel selection"): which module a call belongs to ----
import itertools
import threading

_SITE = threading.local()
_SITE_IDS = itertools.count(1)


def new_call_site() -> int:
    """A fresh tag for one attention module (the library keeps one off-window record per (tag, geometry))."""
    return next(_SITE_IDS)


def set_call_site(site: int) -> None:
    """Tag this thread's following operator calls; 0 = untagged."""
    _SITE.value = int(site)
    _lib.set_call_site(int(site))


def get_call_site() -> int:
    return getattr(_SITE, "value", 0)


# Kernel the last forward / backward call dispatched to, whatever thread made it (msda_last_kernel() is thread-local
# and autograd runs backward calls on its own thread: __graft_entry__.smoke() asserts the backward kernel through this)
LAST_KERNEL = {"forward": "", "backward": ""}


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _raise(rc: int, what: str):
    raise RuntimeError(f"{what} failed (code {rc}): {_lib.last_error()}")


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """-> output (N, Lq, M*D); cf. ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:20-80)."""
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)))
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    suf = _suffix(value, sampling_loc, attn_weight)
    with torch.cuda.device(value.device):
        output = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        if output.numel() == 0:  # a frame without queries: nothing to launch (empty tensors have no storage)
            return output
        keep, hptr = _host_ptr(spatial_shapes, suf in ("f32", "bf16") and D == 32 and Lq == S and L <= 4)
        rc = getattr(_lib.lib, f"msda_forward_{suf}")(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
            attn_weight.data_ptr(), N, S, M, D, L, Lq, P, output.data_ptr(), hptr, _stream(value.device))
        del keep
    if rc != 0:
        _raise(rc, "ms_deform_attn_forward")
    LAST_KERNEL["forward"] = _lib.last_kernel()
    return output


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight]; cf. ms_deform_attn_cuda_backward (.cu:83-153)."""
    _check_inputs((("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)))
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)
    suf = _suffix(value, sampling_loc, attn_weight)
    if grad_output.dtype != value.dtype or grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must match the forward output (N, Lq, M*D) and value's dtype")
    with torch.cuda.device(value.device):
        # bf16: accumulate grad_value in fp32, round once at the end
        acc_dtype = torch.float32 if suf == "bf16" else value.dtype
        grad_loc = torch.empty_like(sampling_loc)
        grad_attn = torch.empty_like(attn_weight)
        if grad_output.numel() == 0:
            return [torch.zeros(value.shape, dtype=value.dtype, device=value.device), grad_loc, grad_attn]
        # (zeroed by the library -- `zero_grad_value` -- in the same launch as whatever else its chosen kernels want cleared)
        grad_value = torch.empty(value.shape, dtype=acc_dtype, device=value.device)
        keep, hptr = _host_ptr(spatial_shapes, suf in ("f32", "bf16") and D == 32 and Lq == S and L <= 4)
        stream = _stream(value.device)
        # scratch for the sort + gather form of grad_value (sampling points far from their queries: no float atomics);
        # 0 for every other call
        ws_bytes = 0
        if suf != "f64":
            ws_bytes = int(_lib.lib.msda_backward_workspace_bytes(0, N, S, M, D, L, Lq, P, value.element_size(), stream))
        if ws_bytes:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device)
            rc = getattr(_lib.lib, f"msda_backward_ws_{suf}")(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                grad_loc.data_ptr(), grad_attn.data_ptr(), 1, hptr, ws.data_ptr(), ws_bytes, stream)
            del ws
        else:
            rc = getattr(_lib.lib, f"msda_backward_{suf}")(
                value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                attn_weight.data_ptr(), grad_output.data_ptr(), N, S, M, D, L, Lq, P, grad_value.data_ptr(),
                grad_loc.data_ptr(), grad_attn.data_ptr(), 1, hptr, stream)
        del keep
    if rc != 0:
        _raise(rc, "ms_deform_attn_backward")
    LAST_KERNEL["backward"] = _lib.last_kernel()
    if grad_value.dtype != value.dtype:
        grad_value = grad_value.to(value.dtype)
    return [grad_value, grad_loc, grad_attn]


# ----------------------------------------------------------------------------------------------------------
# Fused prologue (no reference counterpart at the extension level: it replaces the elementwise chain of the
# reference MODULE, models/ops/modules/ms_deform_attn.py:104-123).  C ABI: msda_fused_* in include/msda_hip.h.
# ----------------------------------------------------------------------------------------------------------
FUSED_MAX_POINTS = 64       # L * P limit of the fused kernels


def _fused_dims(value, spatial_shapes, level_start_index, proj, ref, pad_mask, n_heads, n_points):
    if value.dim() != 4 or proj.dim() != 3 or ref.dim() != 4:
        raise RuntimeError("expected value (N,S,M,D), proj (N,Lq,>=3*M*L*P), reference_points (N,Lq,L,2|4)")
    N, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Lq, P = proj.shape[1], int(n_points)
    if M != n_heads or proj.shape[0] != N or proj.shape[2] < 3 * M * L * P:
        raise RuntimeError("proj does not match value / n_heads / n_points")
    if tuple(ref.shape[:3]) != (N, Lq, L) or ref.shape[3] not in (2, 4):
        raise RuntimeError("reference_points must be (N, Lq, L, 2|4)")
    if proj.dtype != torch.float32 or ref.dtype != torch.float32:
        raise RuntimeError("proj / reference_points must be float32")
    if pad_mask is not None and (pad_mask.dtype != torch.bool or tuple(pad_mask.shape) != (N, S)):
        raise RuntimeError("padding mask must be a bool (N, S) tensor")
    if tuple(spatial_shapes.shape) != (L, 2) or tuple(level_start_index.shape) != (L,):
        raise RuntimeError("spatial_shapes must be (L,2) and level_start_index (L,)")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 tensors")
    if L * P > FUSED_MAX_POINTS:
        raise RuntimeError(f"fused prologue supports at most {FUSED_MAX_POINTS} sampling points per head")
    return N, S, M, D, L, Lq, P


def value_pixel_stride(value):
    """How a (N, S, M, D) ``value`` reaches the kernels: 0 -- contiguous; e > 0 -- in place with e elements between
    pixels (rows that are a slice of a wider projection: include/msda_hip.h, msda_next_value_pixel_stride); None -- neither,
    the caller makes a contiguous copy."""
    if value.is_contiguous():
        return 0
    N, S, M, D = value.shape
    st = value.stride()
    if (D == 32 and st[3] == 1 and st[2] == D and st[1] >= M * D and (N == 1 or st[0] == S * st[1])
            and (st[1] * value.element_size()) % 16 == 0 and value.data_ptr() % 16 == 0):
        return int(st[1])
    return None


def _fused_suffix(value):
    suf = {torch.float32: "f32", torch.bfloat16: "bf16"}.get(value.dtype)
    if suf is None:
        raise RuntimeError(f"fused ms_deform_attn: unsupported value dtype {value.dtype}")
    return suf


def fused_supported(value_dtype, head_dim: int, n_levels: int, n_points: int) -> bool:
    """Shapes / dtypes the fused entry points take (everything else keeps the reference-compatible operator)."""
    return value_dtype in (torch.float32, torch.bfloat16) and n_levels * n_points <= FUSED_MAX_POINTS


def ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, proj, reference_points, pad_mask,
                                 n_heads, n_points):
    """-> output (N, Lq, M*D) from the raw query projection [offsets | logits], the reference points and the
    padding mask of ``value`` (softmax, location arithmetic and mask fill in-kernel)."""
    vstride = value_pixel_stride(value) if value.dim() == 4 else 0
    if vstride is None:
        value, vstride = value.contiguous(), 0
    named = [("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
             ("proj", proj), ("reference_points", reference_points)]
    if not vstride:
        named.insert(0, ("value", value))
    elif not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    if pad_mask is not None:
        named.append(("padding_mask", pad_mask))
    _check_inputs(named)
    N, S, M, D, L, Lq, P = _fused_dims(value, spatial_shapes, level_start_index, proj, reference_points, pad_mask,
                                       n_heads, n_points)
    suf = _fused_suffix(value)
    with torch.cuda.device(value.device):
        output = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        if output.numel() == 0:
            return output
        keep, hptr = _host_ptr(spatial_shapes, D == 32 and Lq == S and L <= 4 and not vstride)
        if vstride:
            _lib.lib.msda_next_value_pixel_stride(vstride)
        rc = getattr(_lib.lib, f"msda_fused_forward_{suf}")(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), proj.data_ptr(),
            proj.shape[2], reference_points.data_ptr(), reference_points.shape[3],
            pad_mask.data_ptr() if pad_mask is not None else None, N, S, M, D, L, Lq, P, output.data_ptr(), hptr,
            _stream(value.device))
        del keep
    if rc != 0:
        _raise(rc, "ms_deform_attn_fused_forward")
    LAST_KERNEL["forward"] = _lib.last_kernel()
    return output


def ms_deform_attn_fused_backward(value, spatial_shapes, level_start_index, proj, reference_points, pad_mask,
                                  grad_output, n_heads, n_points, need_ref_grad=False, fwd_output=None,
                                  grad_value_out=None):
    """-> [grad_value, grad_proj, grad_reference_points | None].  ``fwd_output``: the tensor the fused forward of the same
    call returned, when the caller still holds it (an autograd function does): the encoder's backward is then one kernel
    (include/msda_hip.h, msda_fused_backward_out_*).  ``grad_value_out``: for a ``value`` that is a slice of a wider
    tensor (``value_pixel_stride`` > 0) the float32 slice of the wide gradient tensor that belongs to it -- same shape and
    strides, already zeroed by the caller; the gradient is accumulated there and that tensor is returned."""
    vstride = value_pixel_stride(value) if value.dim() == 4 else 0
    if vstride and (grad_value_out is None or grad_value_out.dtype != torch.float32 or not grad_value_out.is_cuda
                    or grad_value_out.shape != value.shape or grad_value_out.stride() != value.stride()):
        vstride = None
    if vstride is None or (not vstride and grad_value_out is not None):
        value, vstride, grad_value_out = value.contiguous(), 0, None
    named = [("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
             ("proj", proj), ("reference_points", reference_points), ("grad_output", grad_output)]
    if not vstride:
        named.insert(0, ("value", value))
    elif not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    if pad_mask is not None:
        named.append(("padding_mask", pad_mask))
    _check_inputs(named)
    N, S, M, D, L, Lq, P = _fused_dims(value, spatial_shapes, level_start_index, proj, reference_points, pad_mask,
                                       n_heads, n_points)
    suf = _fused_suffix(value)
    if grad_output.dtype != value.dtype or grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must match the forward output (N, Lq, M*D) and value's dtype")
    with torch.cuda.device(value.device):
        # (zeroed by the library, `zero_grad_value`, next to whatever else its chosen kernels want cleared)
        if vstride:
            grad_value = grad_value_out
        else:
            grad_value = (torch.empty if grad_output.numel() else torch.zeros)(value.shape, dtype=torch.float32, device=value.device)
        used = 3 * M * L * P
        grad_proj = (torch.empty_like(proj) if proj.shape[2] == used else torch.zeros_like(proj))
        ref_part = None
        if need_ref_grad:
            ref_part = torch.empty((N, Lq, M, L, reference_points.shape[3]), dtype=torch.float32, device=value.device)
        if grad_output.numel() == 0:
            gref = ref_part.sum(2) if ref_part is not None else None
            return [grad_value.to(value.dtype), grad_proj, gref]
        keep, hptr = _host_ptr(spatial_shapes, D == 32 and Lq == S and L <= 4 and not need_ref_grad and not vstride)
        # scratch: the three-kernel form of the region-tiled backward (the prologue materialised once per row) and, for
        # sampling points far from their queries, the records of the sort + gather form of grad_value
        stream = _stream(value.device)
        ws_bytes = int(_lib.lib.msda_backward_workspace_bytes(1, N, S, M, D, L, Lq, P, value.element_size(), stream)) if hptr else 0
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=value.device) if ws_bytes else None
        if fwd_output is not None and not (fwd_output.is_contiguous() and fwd_output.dtype == value.dtype
                                           and fwd_output.numel() == grad_output.numel()):
            fwd_output = None
        if vstride:
            _lib.lib.msda_next_value_pixel_stride(vstride)
        rc = getattr(_lib.lib, f"msda_fused_backward_out_{suf}")(
            value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), proj.data_ptr(),
            proj.shape[2], reference_points.data_ptr(), reference_points.shape[3],
            pad_mask.data_ptr() if pad_mask is not None else None, grad_output.data_ptr(),
            fwd_output.data_ptr() if fwd_output is not None else None, N, S, M, D, L, Lq, P,
            grad_value.data_ptr(), grad_proj.data_ptr(), ref_part.data_ptr() if ref_part is not None else None,
            0 if vstride else 1, hptr, ws.data_ptr() if ws is not None else None, ws_bytes, stream)
        del keep, ws
    if rc != 0:
        _raise(rc, "ms_deform_attn_fused_backward")
    LAST_KERNEL["backward"] = _lib.last_kernel()
    if grad_value.dtype != value.dtype and not vstride:       # (a strided call's gradient stays in the caller's fp32 bank)
        grad_value = grad_value.to(value.dtype)
    return [grad_value, grad_proj, ref_part.sum(2) if ref_part is not None else None]


def fused_points(spatial_shapes, proj, reference_points, n_heads, n_points):
    """Parity hook: (sampling locations (N,Lq,M,L,P,2), attention weights (N,Lq,M,L,P)) as the fused kernels
    compute them from the raw projection and the reference points."""
    if not (proj.is_cuda and proj.dtype == torch.float32 and proj.is_contiguous() and reference_points.is_contiguous()):
        raise RuntimeError("fused_points expects contiguous float32 CUDA tensors")
    N, Lq = proj.shape[0], proj.shape[1]
    L, M, P = spatial_shapes.shape[0], int(n_heads), int(n_points)
    with torch.cuda.device(proj.device):
        loc = torch.empty((N, Lq, M, L, P, 2), dtype=torch.float32, device=proj.device)
        attn = torch.empty((N, Lq, M, L, P), dtype=torch.float32, device=proj.device)
        rc = _lib.lib.msda_fused_points_f32(spatial_shapes.data_ptr(), proj.data_ptr(), proj.shape[2],
                                            reference_points.data_ptr(), reference_points.shape[3], N, M, L, Lq, P,
                                            loc.data_ptr(), attn.data_ptr(), _stream(proj.device))
    if rc != 0:
        _raise(rc, "msda_fused_points_f32")
    return loc, attn


def sample_indices(spatial_shapes, sampling_loc):
    """Parity hook: (h_low, w_low, gate) of every sampling point, as the kernels compute them."""
    if not (sampling_loc.is_cuda and sampling_loc.dtype == torch.float32 and sampling_loc.is_contiguous()):
        raise RuntimeError("sample_indices expects a contiguous float32 CUDA sampling_loc")
    N, Lq, M, L, P, _ = sampling_loc.shape
    with torch.cuda.device(sampling_loc.device):
        h = torch.empty((N, Lq, M, L, P), dtype=torch.int32, device=sampling_loc.device)
        w = torch.empty_like(h)
        g = torch.empty((N, Lq, M, L, P), dtype=torch.uint8, device=sampling_loc.device)
        rc = _lib.lib.msda_sample_indices_f32(spatial_shapes.data_ptr(), sampling_loc.data_ptr(), N, M, L, Lq, P,
                                              h.data_ptr(), w.data_ptr(), g.data_ptr(),
                                              _stream(sampling_loc.device))
    if rc != 0:
        _raise(rc, "msda_sample_indices_f32")
    return h, w, g
