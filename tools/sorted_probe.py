#!/usr/bin/env python
"""Backward at the encoder shape under each kernel family, for sampling locations from near to uniform:
   python tools/sorted_probe.py [--fused] [--batch N]
variant 12 (counting sort inside windows, margins 6 / 9), rows (whole-row float atomics), 13 (global sort + gather)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def time_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--only", default="", help="e.g. uniform or encoder_like:4 -- that case only (profiling runs)")
    ap.add_argument("--variants", default="", help="e.g. 13 -- that family only")
    ap.add_argument("--opt", default="", help="msda_set_option pairs, e.g. bwd_sort_qc=32,bwd_sort_emult=4")
    args = ap.parse_args()
    from memotr_amd import MultiScaleDeformableAttention as MSDA
    from memotr_amd import _lib
    from memotr_amd.MultiScaleDeformableAttention import tag_host_shapes
    from memotr_amd.synth import make_inputs, to_fused_inputs
    for kv in filter(None, args.opt.split(",")):
        k, v = kv.split("=")
        _lib.set_option(k, int(v))
    cases = [("encoder_like", 1.0), ("encoder_like", 2.0), ("encoder_like", 3.0), ("encoder_like", 4.0),
             ("encoder_like", 6.0), ("encoder_like", 8.0), ("uniform", 1.0)]
    if args.only:
        d, _, o = args.only.partition(":")
        cases = [(d, float(o) if o else 1.0)]
    fams = ((12, 0), (12, 1), (0, 2), (13, -1))
    if args.variants:
        fams = tuple(f for f in fams if str(f[0]) in args.variants.split(","))
    print("dist x scale | bins m6 | bins m9 | rows | sorted (13)   [ms]  fused=%s batch=%d bf16=%s" % (args.fused, args.batch, args.bf16))
    for dist, osc in cases:
        x = make_inputs(dist=dist, off_scale=osc, device="cuda", seed=3, batch=args.batch)
        tag_host_shapes(x["shapes"], x["shapes_list"])
        dt = torch.bfloat16 if args.bf16 else torch.float32
        value, go = x["value"].to(dt), x["grad_out"].to(dt)
        if args.fused:
            f = to_fused_inputs(x)
            call = lambda: MSDA.ms_deform_attn_fused_backward(value, x["shapes"], x["level_start"], f["proj"], f["ref"],   # noqa: E731
                                                              None, go, 8, 4)
        else:
            call = lambda: MSDA.ms_deform_attn_backward(value, x["shapes"], x["level_start"], x["loc"], x["attn"], go, 64)   # noqa: E731
        row = []
        for variant, level in fams:
            _lib.set_option("bwd_variant", variant)
            _lib.set_option("sel_level", level)
            _lib.set_option("bwd_sorted", 0 if variant == 0 else 1)      # (level 2 without the sorted form: the rows kernel)
            ms = time_ms(call)
            row.append((ms, _lib.last_kernel() if not args.fused else MSDA.LAST_KERNEL["backward"]))
        _lib.set_option("bwd_variant", 0)
        _lib.set_option("sel_level", -1)
        _lib.set_option("bwd_sorted", 1)
        print("%-12s x%-3g | " % (dist, osc) + " | ".join("%.3f %s" % (ms, k.replace("msda_bwd_d32_", "")) for ms, k in row))


if __name__ == "__main__":
    main()
