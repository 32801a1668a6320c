#!/usr/bin/env python
"""Round-4 kernel selection (memotr_amd/csrc/msda_select.h): what the windowed backward measures and what it costs,
against the magnitude of the sampling offsets; then the selector left to itself.

    python tools/selector_probe.py [--out gpurun_out/selector_probe.txt]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, MsdaCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/selector_probe.txt")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    log = open(args.out, "w")

    def say(*a):
        s = " ".join(str(t) for t in a)
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()

    site = 100
    for dist, osc in (("encoder_like", 1.0), ("encoder_like", 1.5), ("encoder_like", 2.0), ("encoder_like", 3.0),
                      ("encoder_like", 4.0), ("encoder_like", 6.0), ("encoder_like", 8.0), ("uniform", 1.0)):
        x = make_inputs(dist=dist, off_scale=osc, device="cuda")
        call, fcall = MsdaCall(x), FusedCall(x)
        line = f"{dist:12s} x{osc}:"
        for lvl in (0, 1):
            site += 1
            _lib.set_call_site(site)
            _lib.set_option("bwd_variant", 12)
            _lib.set_option("sel_level", lvl)
            for _ in range(3):
                call.bwd()
            torch.cuda.synchronize()
            call.bwd()
            _, f, fi = _lib.selector_last()
            torch.cuda.synchronize()
            ms = time_kernel(call.bwd, iters=20)
            line += f"  level {lvl}: {ms*1e3:7.1f} us  off {f*100:6.2f} % inner {fi*100:6.2f} %"
        _lib.set_option("sel_level", -1)
        _lib.set_option("bwd_variant", 1)
        ms = time_kernel(call.bwd, iters=10)
        line += f"  generic {ms*1e3:7.1f} us"
        # the selector on its own: 48 calls from a fresh record, then steady state
        site += 1
        _lib.set_call_site(site)
        _lib.set_option("bwd_variant", 0)
        trail = []
        for i in range(48):
            call.bwd()
            if i % 4 == 3:
                torch.cuda.synchronize()
            trail.append(_lib.selector_last()[0])
        ms = time_kernel(call.bwd, iters=64)
        msf = time_kernel(fcall.bwd, iters=64)
        line += f"  auto {ms*1e3:7.1f} us (fused {msf*1e3:7.1f})  levels {''.join(map(str, trail))} [{_lib.last_kernel()}]"
        # forward: windows (level 0) vs head-major gather (level 1), then the selector
        _lib.set_option("fwd_variant", 0)
        for lvl in (0, 1):
            _lib.set_option("sel_level", lvl)
            ms = time_kernel(fcall.fwd, iters=50)
            _, f, _ = _lib.selector_last()
            line += f"  | fwd level {lvl}: {ms*1e3:6.1f} us (off {f*100:5.1f} %)"
        _lib.set_option("sel_level", -1)
        site += 1
        _lib.set_call_site(site)
        for i in range(24):
            fcall.fwd()
            if i % 4 == 3:
                torch.cuda.synchronize()
        ms = time_kernel(fcall.fwd, iters=64)
        line += f"  auto {ms*1e3:6.1f} us [{_lib.last_kernel()}]"
        say(line)
    _lib.set_call_site(0)
    log.close()


if __name__ == "__main__":
    main()
