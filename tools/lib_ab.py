#!/usr/bin/env python
"""A/B of BUILDS of libmsda_hip.so inside one process on one box (box-to-box spread is 2-4 %, so decisions between
builds are taken here): every library given is loaded through ctypes directly (no ABI check -- older rounds' builds
qualify) and timed on the encoder call, forward fused / plain and (ABI >= 3) backward fused with workspace.

    python tools/lib_ab.py [--out gpurun_out/lib_ab.txt] [--rounds 3] name=path.so [name=path.so ...]
    (a name ending in "+opt=val,opt=val" applies msda_set_option pairs to that library first)
"""
import argparse
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from memotr_amd.synth import make_inputs, to_fused_inputs  # noqa: E402

c_int, c_void_p = ctypes.c_int, ctypes.c_void_p
FWD = [c_void_p] * 5 + [c_int] * 7 + [c_void_p, c_void_p, c_void_p]
FUSED_FWD = [c_void_p] * 4 + [c_int, c_void_p, c_int, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, c_void_p]
FUSED_BWD_WS = ([c_void_p] * 4 + [c_int, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p] * 3 +
                [c_int, c_void_p, c_void_p, ctypes.c_size_t, c_void_p])


def timed(fn, iters=200, min_warm_ms=40.0, batches=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < min_warm_ms:
        fn()
    per = iters // batches
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(batches)]
    for s, e in ev:
        s.record()
        for _ in range(per):
            fn()
        e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) / per for s, e in ev)
    return t[len(t) // 2] * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/lib_ab.txt")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--mask", action="store_true", help="pass an all-false padding mask (what the model does)")
    ap.add_argument("libs", nargs="+")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    x = make_inputs(device="cuda", batch=args.batch)
    f = to_fused_inputs(x)
    N, S, M, D = x["value"].shape
    Lq, L, P = x["loc"].shape[1], x["loc"].shape[3], x["loc"].shape[4]
    out = torch.empty(N, Lq, M * D, device="cuda")
    gv, gp = torch.empty_like(x["value"]), torch.empty_like(f["proj"])
    hs = x["shapes"].cpu().contiguous()
    stream = torch.cuda.current_stream().cuda_stream
    mask = torch.zeros(N, S, dtype=torch.bool, device="cuda") if args.mask else None
    mptr = mask.data_ptr() if mask is not None else None
    entries = []
    for spec in args.libs:
        name, path = spec.split("=", 1)
        opts = []
        if "+" in name:
            name, o = name.split("+", 1)
            opts = [kv.split(":") for kv in o.split(",")]
        lib = ctypes.CDLL(os.path.abspath(path))
        lib.msda_forward_f32.argtypes, lib.msda_fused_forward_f32.argtypes = FWD, FUSED_FWD
        lib.msda_last_kernel.restype = ctypes.c_char_p
        lib.msda_last_error.restype = ctypes.c_char_p
        lib.msda_set_option.argtypes = [ctypes.c_char_p, c_int]
        for k, v in opts:
            assert lib.msda_set_option(k.encode(), int(v)) == 0, (k, v)
        has_ws = hasattr(lib, "msda_fused_backward_ws_f32")
        ws = None
        if has_ws:
            lib.msda_fused_backward_ws_f32.argtypes = FUSED_BWD_WS
            lib.msda_fused_workspace_bytes.argtypes = [c_int] * 5
            lib.msda_fused_workspace_bytes.restype = ctypes.c_size_t
            ws = torch.empty((int(lib.msda_fused_workspace_bytes(N, Lq, M, L, P)),), dtype=torch.uint8, device="cuda")

        def fused(lib=lib):
            rc = lib.msda_fused_forward_f32(x["value"].data_ptr(), x["shapes"].data_ptr(), x["level_start"].data_ptr(),
                                            f["proj"].data_ptr(), f["proj"].shape[2], f["ref"].data_ptr(), 2, mptr,
                                            N, S, M, D, L, Lq, P, out.data_ptr(), hs.data_ptr(), stream)
            assert rc == 0, lib.msda_last_error()

        def plain(lib=lib):
            rc = lib.msda_forward_f32(x["value"].data_ptr(), x["shapes"].data_ptr(), x["level_start"].data_ptr(),
                                      x["loc"].data_ptr(), x["attn"].data_ptr(), N, S, M, D, L, Lq, P,
                                      out.data_ptr(), hs.data_ptr(), stream)
            assert rc == 0, lib.msda_last_error()

        def bwd(lib=lib, ws=ws):
            rc = lib.msda_fused_backward_ws_f32(x["value"].data_ptr(), x["shapes"].data_ptr(),
                                                x["level_start"].data_ptr(), f["proj"].data_ptr(), f["proj"].shape[2],
                                                f["ref"].data_ptr(), 2, mptr, x["grad_out"].data_ptr(), N, S, M, D, L,
                                                Lq, P, gv.data_ptr(), gp.data_ptr(), None, 1, hs.data_ptr(),
                                                ws.data_ptr(), ws.numel(), stream)
            assert rc == 0, lib.msda_last_error()
        entries.append((name + ("+" + ",".join(f"{k}:{v}" for k, v in opts) if opts else ""), lib, fused, plain,
                        bwd if has_ws else None))
    ref = None
    lines = []
    for name, lib, fused, plain, bwd in entries:          # same outputs first
        fused()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        lines.append(f"{name:40s} fused output vs first library: max |diff| {float((out - ref).abs().max()):.2e}"
                     f"  [{lib.msda_last_kernel().decode()}]")
    for r in range(args.rounds):
        for name, lib, fused, plain, bwd in entries:
            tf, tp = timed(fused), timed(plain)
            tb = timed(bwd, iters=50) if bwd is not None else float("nan")
            lines.append(f"round {r} {name:40s} fwd fused {tf:7.2f} us   plain {tp:7.2f} us   bwd fused {tb:8.2f} us")
    with open(args.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
