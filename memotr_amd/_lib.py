"""ctypes binding of libmsda_hip.so (C ABI in include/msda_hip.h).

No CPU fallback: if the library is missing or does not load, importing this module
raises, and with it every operator of the package.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEMOTR_MSDA_LIB") or os.path.join(_HERE, "lib", "libmsda_hip.so")   # (override: A/B builds)

ABI_VERSION = 7

c_int = ctypes.c_int
c_void_p = ctypes.c_void_p

_FWD_ARGS = [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_void_p, c_void_p]
_BWD_ARGS = [c_void_p] * 6 + [c_int] * 7 + [c_void_p] * 3 + [c_int, c_void_p, c_void_p]
# fused prologue: value, shapes, lstart, proj, proj_stride, ref, ref_dim, pad_mask, [grad_out,] N..P, outputs...
_FUSED_FWD_ARGS = [c_void_p] * 4 + [c_int, c_void_p, c_int, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, c_void_p]
_FUSED_BWD_ARGS = ([c_void_p] * 4 + [c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p] * 3 +
                   [c_int, c_void_p, c_void_p])

_FUSED_BWD_WS_ARGS = _FUSED_BWD_ARGS[:-1] + [c_void_p, ctypes.c_size_t, c_void_p]
_BWD_WS_ARGS = _BWD_ARGS[:-1] + [c_void_p, ctypes.c_size_t, c_void_p]
# ... + fwd_out after grad_out
_FUSED_BWD_OUT_ARGS = _FUSED_BWD_WS_ARGS[:9] + [c_void_p] + _FUSED_BWD_WS_ARGS[9:]

SYMBOLS = {
    "msda_abi_version": ([], c_int),
    "msda_last_error": ([], ctypes.c_char_p),
    "msda_last_kernel": ([], ctypes.c_char_p),
    "msda_forward_f32": (_FWD_ARGS, c_int),
    "msda_forward_f64": (_FWD_ARGS, c_int),
    "msda_forward_bf16": (_FWD_ARGS, c_int),
    "msda_backward_f32": (_BWD_ARGS, c_int),
    "msda_backward_f64": (_BWD_ARGS, c_int),
    "msda_backward_bf16": (_BWD_ARGS, c_int),
    "msda_backward_ws_f32": (_BWD_WS_ARGS, c_int),
    "msda_backward_ws_bf16": (_BWD_WS_ARGS, c_int),
    "msda_backward_workspace_bytes": ([c_int] * 9 + [c_void_p], ctypes.c_size_t),
    "msda_fused_forward_f32": (_FUSED_FWD_ARGS, c_int),
    "msda_fused_forward_bf16": (_FUSED_FWD_ARGS, c_int),
    "msda_fused_backward_f32": (_FUSED_BWD_ARGS, c_int),
    "msda_fused_backward_bf16": (_FUSED_BWD_ARGS, c_int),
    "msda_fused_backward_ws_f32": (_FUSED_BWD_WS_ARGS, c_int),
    "msda_fused_backward_ws_bf16": (_FUSED_BWD_WS_ARGS, c_int),
    "msda_fused_backward_out_f32": (_FUSED_BWD_OUT_ARGS, c_int),
    "msda_fused_backward_out_bf16": (_FUSED_BWD_OUT_ARGS, c_int),
    "msda_fused_workspace_bytes": ([c_int] * 5, ctypes.c_size_t),
    "msda_sample_indices_f32": ([c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 4, c_int),
    "msda_fused_points_f32": ([c_void_p, c_void_p, c_int, c_void_p, c_int] + [c_int] * 5 + [c_void_p] * 3, c_int),
    "msda_set_call_site": ([ctypes.c_uint64], None),
    "msda_selector_last": ([ctypes.POINTER(c_int), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)], c_int),
    "msda_selector_next": ([c_int] * 4, c_int),
    "msda_selector_poll": ([ctypes.POINTER(ctypes.c_uint64)], c_int),
    "msda_selector_poll_sites": ([ctypes.POINTER(ctypes.c_uint64), c_int, c_int, ctypes.POINTER(ctypes.c_uint64)], c_int),
    "msda_next_value_pixel_stride": ([ctypes.c_long], c_int),
    "msda_selector_reset": ([], c_int),
    "msda_set_option": ([ctypes.c_char_p, c_int], c_int),
    "msda_get_option": ([ctypes.c_char_p, ctypes.POINTER(c_int)], c_int),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m memotr_amd.build` "
            "(hipcc --offload-arch=gfx950). memotr_amd has no CPU fallback.")
    # torch ships its own libamdhip64 (same SONAME); importing it first makes the HIP
    # library bind to the runtime torch's streams/allocations live in.
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.msda_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libmsda_hip.so ABI {got} != binding ABI {ABI_VERSION}; rebuild the library")
    return lib


lib = _load()


def last_error() -> str:
    return lib.msda_last_error().decode()


def last_kernel() -> str:
    return lib.msda_last_kernel().decode()


def set_option(key: str, value: int) -> None:
    if lib.msda_set_option(key.encode(), int(value)) != 0:
        raise ValueError(f"msda_set_option({key!r}, {value}) rejected: {last_error()}")


def get_option(key: str) -> int:
    out = c_int(0)
    if lib.msda_get_option(key.encode(), ctypes.byref(out)) != 0:
        raise ValueError(f"unknown option {key!r}")
    return out.value


def set_call_site(site: int) -> None:
    """Tag this thread's following operator calls (kernel selection keeps one record per call site)."""
    lib.msda_set_call_site(ctypes.c_uint64(site & 0xFFFFFFFFFFFFFFFF))


def selector_poll() -> int:
    """Read every selector record of the current device, move the levels, return the signature of the levels a call
    would run at now (0: every record at level 0).  For code that replays captured launches: key the graph on it."""
    sig = ctypes.c_uint64(0)
    lib.msda_selector_poll(ctypes.byref(sig))
    return int(sig.value)


def selector_poll_sites(sites, probe: bool = False) -> int:
    """`selector_poll` for the records of the call sites in `sites` only (a graph cache hashes the modules its graphs
    hold); `probe`: records at a level without windows announce one level down -- the caller counts its own polls."""
    sites = [int(s) & 0xFFFFFFFFFFFFFFFF for s in sites]
    arr = (ctypes.c_uint64 * max(len(sites), 1))(*sites)
    sig = ctypes.c_uint64(0)
    if lib.msda_selector_poll_sites(arr, len(sites), 1 if probe else 0, ctypes.byref(sig)) < 0:
        raise ValueError(f"msda_selector_poll_sites: {last_error()}")
    return int(sig.value)


def selector_reset() -> None:
    """Forget every selector record (waits for the device).  Captured graphs stay valid."""
    rc = lib.msda_selector_reset()
    if rc != 0:
        raise RuntimeError(f"msda_selector_reset: {last_error()}")


def selector_last():
    """(level, off-window share, share outside the next smaller window) of this thread's last selected call;
    shares < 0: nothing measured yet."""
    lv, fr, fi = c_int(0), ctypes.c_float(0.0), ctypes.c_float(0.0)
    lib.msda_selector_last(ctypes.byref(lv), ctypes.byref(fr), ctypes.byref(fi))
    return lv.value, fr.value, fi.value
