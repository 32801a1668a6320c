"""Build memotr_amd/lib/libmsda_hip.so (the operator) and libclip_ops_hip.so (fused small-tensor chains of the train
step) with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "msda_hip.hip")
HDR = os.path.join(os.path.dirname(_HERE), "include", "msda_hip.h")
# the operator's kernels, one header per family, all included by msda_hip.hip
KERNEL_HEADERS = tuple(os.path.join(_HERE, "csrc", n) for n in (
    "msda_common.h", "msda_select.h", "msda_generic.h", "msda_fwd_gather.h", "msda_fwd_win.h", "msda_tile.h",
    "msda_bwd_tile_lv.h", "msda_bwd_bins.h", "msda_bwd_rows.h", "msda_bwd_sorted.h", "msda_fused_side.h"))
LIB_DIR = os.path.join(_HERE, "lib")
LIB = os.path.join(LIB_DIR, "libmsda_hip.so")
CLIP_SRC = os.path.join(_HERE, "csrc", "clip_ops.hip")
CLIP_HDR = os.path.join(os.path.dirname(_HERE), "include", "clip_ops_hip.h")
CLIP_LIB = os.path.join(LIB_DIR, "libclip_ops_hip.so")
ASSIGN_CORE = os.path.join(_HERE, "csrc", "assign_core.h")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-munsafe-fp-atomics",      # float/double atomicAdd -> global_atomic_add_f32/_f64
]


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libmsda_hip.so)")


def source_hash() -> str:
    """sha256 (first 16 hex digits) of the kernel sources: stamps measurements (profiles/traffic.json) so that a
    number taken on other kernels is recognised as stale."""
    import hashlib
    h = hashlib.sha256()
    for p in (SRC,) + KERNEL_HEADERS:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _stale(lib: str, deps) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(p) > t for p in deps)


def needs_build() -> bool:
    return _stale(LIB, (SRC, HDR) + KERNEL_HEADERS)


def _compile(src: str, lib: str, verbose: bool) -> str:
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc_path(), *HIPCC_FLAGS, src, "-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


def build_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    return _compile(SRC, LIB, verbose)


def build_clip_lib(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale(CLIP_LIB, (CLIP_SRC, CLIP_HDR, ASSIGN_CORE)):
        return CLIP_LIB
    return _compile(CLIP_SRC, CLIP_LIB, verbose)


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_clip_lib(force=True, verbose=True))
