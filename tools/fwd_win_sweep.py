#!/usr/bin/env python
"""Round 5: the windowed forward's launch options timed in ONE process on one box (region shape x workgroup size x
register budget x early loads x margins), N = 1 fused / plain and N = 5 fused, each checked against the generic kernel.

    python tools/fwd_win_sweep.py [--out gpurun_out/fwd_win_sweep.txt] [--lib path.so]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, MsdaCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

DEFAULTS = dict(fwd_variant=0, fwd_win_rlog=0, fwd_win_rlogx=0, fwd_win_block=0, fwd_win_l0=1,
                fwd_win_margins=0x3333, fwd_head_major=0, fwd_win_early=9, fwd_win_wps=0, fwd_win_place=0, fwd_win_grid=1, fwd_win_rsy=0, fwd_win_rsx=0,
                sel_level=-1, auto_select=1)

R3 = dict(fwd_variant=12, fwd_win_rlog=3)
CONFIGS = [
    ("win default (16x16, 512 thr, w4)", dict(fwd_variant=12)),
    ("win 16x16 512 thr measured placement", dict(fwd_variant=12, fwd_win_place=1)),
    ("win 16x16 512 thr e0", dict(fwd_variant=12, fwd_win_early=0)),
    ("win 16x16 512 thr e2", dict(fwd_variant=12, fwd_win_early=2)),
    ("gather<4> head-major", dict(fwd_variant=3, fwd_head_major=1)),
    ("gather<4>", dict(fwd_variant=3)),
    ("win 8x8 256 thr w3 e4 (r3-4 default)", dict(R3, fwd_win_block=256)),
    ("win 8x8 256 thr w4 e2", dict(R3, fwd_win_block=256, fwd_win_wps=4, fwd_win_early=2)),
    ("win 16x8 512 thr e0", dict(R3, fwd_win_rlogx=4, fwd_win_block=512, fwd_win_early=0)),
    ("win 16x8 512 thr e2", dict(R3, fwd_win_rlogx=4, fwd_win_block=512, fwd_win_early=2)),
    ("win 32x8 512 thr e2", dict(R3, fwd_win_rlogx=5, fwd_win_block=512, fwd_win_early=2)),
    ("win 32x16 512 thr e2", dict(fwd_variant=12, fwd_win_rlog=4, fwd_win_rlogx=5, fwd_win_block=512, fwd_win_early=2)),
    ("win 16x16 512 thr m2333", dict(fwd_variant=12, fwd_win_margins=0x2333)),
    ("win 16x16 512 thr m2222", dict(fwd_variant=12, fwd_win_margins=0x2222)),
    ("win 16x16 512 thr m4333", dict(fwd_variant=12, fwd_win_margins=0x4333)),
    ("win 16x16 256 thr", dict(fwd_variant=12, fwd_win_rlog=4, fwd_win_block=256)),
    # round 6: fewer levels in LDS (level 1 through the vector L1 next to level 0): half the fill, ~44 KB of windows
    ("r6 win l0=2 (levels 2-3 in LDS) e0", dict(fwd_variant=12, fwd_win_l0=2, fwd_win_early=0)),
    ("r6 win l0=2 e2", dict(fwd_variant=12, fwd_win_l0=2, fwd_win_early=2)),
    ("r6 win l0=2 16x8 e2", dict(R3, fwd_win_rlogx=4, fwd_win_block=512, fwd_win_l0=2, fwd_win_early=2)),
    ("r6 win l0=2 8x8 256 w3 e4", dict(R3, fwd_win_block=256, fwd_win_l0=2)),
    ("r6 win l0=3 e2", dict(fwd_variant=12, fwd_win_l0=3, fwd_win_early=2)),
    ("r6 win margins 3,3,2,2 (levels 2-3 smaller)", dict(fwd_variant=12, fwd_win_margins=0x2233)),
    # round 6, late: two wavefronts per SIMD at 256 registers, 12 (6) LDS points per wait ("fwd_win_wps" 2, 256 threads)
    ("r6w 16x16 256 thr w2 p12 e0", dict(fwd_variant=12, fwd_win_rlog=4, fwd_win_block=256, fwd_win_wps=2, fwd_win_early=0)),
    ("r6w 16x16 256 thr w2 p12 e4", dict(fwd_variant=12, fwd_win_rlog=4, fwd_win_block=256, fwd_win_wps=2, fwd_win_early=4)),
    ("r6w 16x8 256 thr w2 p12 e4", dict(R3, fwd_win_rlogx=4, fwd_win_block=256, fwd_win_wps=2, fwd_win_early=4)),
    ("r6w 16x8 256 thr w2 p12 e0", dict(R3, fwd_win_rlogx=4, fwd_win_block=256, fwd_win_wps=2, fwd_win_early=0)),
    ("r6w 8x8 256 thr w2 p12 e4", dict(R3, fwd_win_block=256, fwd_win_wps=2, fwd_win_early=4)),
    ("r6w 16x16 256 thr w4 (4 pts per wait)", dict(fwd_variant=12, fwd_win_rlog=4, fwd_win_block=256, fwd_win_wps=4, fwd_win_early=0)),
    # round 6, last: equal regions of any size ("grid"): one round of workgroups per image at N = 1
    ("r6g power-of-two regions only (grid=0)", dict(fwd_variant=12, fwd_win_grid=0)),
    ("r6g grid by estimate (default)", dict(fwd_variant=12)),
    ("r6g grid 12x24", dict(fwd_variant=12, fwd_win_rsy=12, fwd_win_rsx=24)),
    ("r6g grid 15x19", dict(fwd_variant=12, fwd_win_rsy=15, fwd_win_rsx=19)),
    ("r6g grid 13x21 m2333", dict(fwd_variant=12, fwd_win_rsy=13, fwd_win_rsx=21, fwd_win_margins=0x2333)),
    ("r6g grid 13x21 m2233", dict(fwd_variant=12, fwd_win_rsy=13, fwd_win_rsx=21, fwd_win_margins=0x2233)),
    ("r6g grid 17x17 m2333", dict(fwd_variant=12, fwd_win_rsy=17, fwd_win_rsx=17, fwd_win_margins=0x2333)),
    ("r6g grid 20x14", dict(fwd_variant=12, fwd_win_rsy=20, fwd_win_rsx=14)),
    ("r6g grid 10x28", dict(fwd_variant=12, fwd_win_rsy=10, fwd_win_rsx=28)),
    ("r6g grid 12x24 e2", dict(fwd_variant=12, fwd_win_rsy=12, fwd_win_rsx=24, fwd_win_early=2)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/fwd_win_sweep.txt")
    ap.add_argument("--only", default="", help="substring filter on the configuration names (the default is always kept)")
    args = ap.parse_args()
    if args.only:
        CONFIGS[:] = [c for i, c in enumerate(CONFIGS) if i == 0 or args.only in c[0]]
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    say(f"# tools/fwd_win_sweep.py on {torch.cuda.get_device_name(0)}; encoder shape, encoder-like locations; us per launch")
    say(f"{'config':42s} {'N=1 fused':>10s} {'N=1 plain':>10s} {'N=5 fused':>10s}   kernel / max err vs generic")
    calls = []
    for batch, kinds in ((1, ("fused", "plain")), (5, ("fused",))):
        x = make_inputs(device="cuda", batch=batch)
        for kind in kinds:
            c = FusedCall(x) if kind == "fused" else MsdaCall(x)
            for k, v in DEFAULTS.items():
                _lib.set_option(k, v)
            _lib.set_option("fwd_variant", 1)
            c.fwd()
            torch.cuda.synchronize()
            calls.append((c, c.out.clone()))
    _lib.set_option("sel_level", 0)       # (every configuration at the selector's level 0; the records keep counting)
    DEFAULTS["sel_level"] = 0
    for name, opts in CONFIGS:
        cells, errs, kern = [], [], ""
        for c, ref in calls:
            for k, v in DEFAULTS.items():
                _lib.set_option(k, v)
            for k, v in opts.items():
                _lib.set_option(k, v)
            c.out.zero_()
            try:
                c.fwd()
                torch.cuda.synchronize()
            except RuntimeError as exc:
                cells.append("   failed")
                errs.append(str(exc)[:40])
                continue
            for _ in range(4):      # (the window means of this call site settle within two launches)
                c.fwd()
            torch.cuda.synchronize()
            errs.append("%.1e" % float((c.out - ref).abs().max()))
            kern = _lib.last_kernel()
            cells.append("%10.1f" % (time_kernel(c.fwd, iters=100) * 1e3))
            share = _lib.selector_last()[1]
        say(f"{name:42s} {' '.join(cells)}   {kern}  err {' '.join(errs)}  off-window share {share:.4f}")
    DEFAULTS["sel_level"] = -1
    for k, v in DEFAULTS.items():
        _lib.set_option(k, v)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
