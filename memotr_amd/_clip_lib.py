"""ctypes binding of libclip_ops_hip.so (C ABI in include/clip_ops_hip.h).

Like the operator library there is no substitute: a CUDA tensor reaching one of these chains without the library
raises.  (CPU tensors -- the golden-vector tests -- take the element-wise torch formulation the kernels restate.)
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libclip_ops_hip.so")

ABI_VERSION = 10

c_int, c_long, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p

_PAIR = [c_void_p, c_void_p, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_int]
_FOCAL = [c_void_p, c_long, c_long, c_void_p, c_int, c_int, c_int, c_float, c_float]

SYMBOLS = {
    "clipops_abi_version": ([], c_int),
    "clipops_last_error": ([], ctypes.c_char_p),
    "clipops_match_cost_f32": ([c_void_p, c_long, c_long, c_void_p, c_long, c_long, c_void_p, c_void_p] + [c_int] * 4 +
                               [c_float] * 3 + [c_void_p, c_void_p], c_int),
    "clipops_pair_box_loss_fwd_f32": (_PAIR + [c_void_p, c_void_p, c_void_p], c_int),
    "clipops_pair_box_loss_bwd_f32": (_PAIR + [c_void_p, c_void_p, c_void_p, c_void_p], c_int),
    "clipops_pair_iou_f32": ([c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p], c_int),
    "clipops_track_ownership_i64": ([c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "clipops_focal_labels_i64": ([c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                  c_int, c_int, c_void_p, c_void_p], c_int),
    "clipops_focal_fwd_f32": (_FOCAL + [c_void_p, c_void_p], c_int),
    "clipops_focal_bwd_f32": (_FOCAL + [c_void_p, c_void_p, c_void_p], c_int),
    "clipops_colsum_f32": ([c_void_p, c_long, c_int, c_void_p, c_void_p], c_int),
    "clipops_colsum_partial_f32": ([c_void_p, c_long, c_int, c_int, c_void_p, c_void_p], c_int),
    "clipops_colsum_partial_bf16": ([c_void_p, c_long, c_int, c_int, c_void_p, c_void_p], c_int),
    "clipops_assign_f32": ([c_void_p, c_long, c_long, c_long, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                            c_void_p], c_int),
    "clipops_mha_fwd_f32": ([c_void_p] * 3 + [c_long] * 6 + [c_void_p] + [c_int] * 3 + [c_float, c_void_p, c_void_p,
                                                                                          c_void_p], c_int),
    "clipops_mha_bwd_f32": ([c_void_p] * 3 + [c_long] * 6 + [c_void_p] * 4 + [c_int] * 3 + [c_float] +
                            [c_void_p, c_long, c_long] * 3 + [c_void_p], c_int),
    "clipops_add_layer_norm_fwd_f32": ([c_void_p] * 4 + [c_long, c_float] + [c_void_p] * 4, c_int),
    "clipops_add_layer_norm_bwd_f32": ([c_void_p] * 4 + [c_long, c_int] + [c_void_p] * 3, c_int),
    "clipops_sine_embed_fwd_f32": ([c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_void_p, c_void_p], c_int),
    "clipops_sine_embed_bwd_f32": ([c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p],
                                   c_int),
    "clipops_inverse_sigmoid_fwd_f32": ([c_void_p, c_long, c_float, c_void_p, c_void_p], c_int),
    "clipops_inverse_sigmoid_bwd_f32": ([c_void_p, c_void_p, c_long, c_float, c_void_p, c_void_p], c_int),
    "clipops_refine_boxes_fwd_f32": ([c_void_p, c_void_p, c_long, c_float, c_void_p, c_void_p], c_int),
    "clipops_linear_bwd_f32": ([c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 4, c_int),
    "clipops_linear_fwd_f32": ([c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_void_p], c_int),
    "clipops_relu_bwd_colsum_partial_f32": ([c_void_p, c_void_p, c_long, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "clipops_shift_relu_f32": ([c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_void_p], c_int),
    "clipops_shift_relu_bf16": ([c_void_p, c_void_p, c_void_p, c_long, c_int, c_long, c_void_p], c_int),
    "clipops_refine_boxes_bwd_f32": ([c_void_p, c_void_p, c_void_p, c_long, c_float, c_void_p, c_void_p, c_void_p],
                                     c_int),
}


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m memotr_amd.build` "
                          "(hipcc --offload-arch=gfx950).")
    import torch  # noqa: F401  (binds the HIP runtime torch's streams live in; see _lib.py)

    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    got = lib.clipops_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"libclip_ops_hip.so ABI {got} != binding ABI {ABI_VERSION}; rebuild the library")
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {lib.clipops_last_error().decode()}")
