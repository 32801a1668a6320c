#!/usr/bin/env python
"""Where the counting-sort backward (msda_bwd_d32_bins, `bwd_variant` 12) spends its time: the plain encoder call with
parts of the kernel switched off (`msda_set_option("bwd_ablate", bits)`; results are wrong by construction).

    bits  1 no flush | 2 stop before the gather | 4 no value loads | 8 stop after phase 0 (inputs staged)
         16 stop after phase 1 (tickets, records) | 32 skip the row phase | 64 skip the sort (scan + entries)

    python tools/bins_ablate.py [--out gpurun_out/bins_ablate.txt]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MsdaCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

CASES = (("complete", 0), ("stop after phase 0", 8), ("stop after phase 1", 16),
         ("no row phase, no sort, no gather", 64 + 32 + 2), ("row phase only (no sort, no gather)", 64 + 2),
         ("row phase without value loads, no sort, no gather", 64 + 2 + 4), ("no gather", 2), ("no row phase", 32),
         ("no value loads", 4), ("no flush", 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/bins_ablate.txt")
    args = ap.parse_args()
    call = MsdaCall(make_inputs(device="cuda"))
    _lib.set_option("bwd_variant", 12)
    lines = ["# tools/bins_ablate.py: plain encoder call, us per launch"]
    for rnd in range(2):
        for name, bits in CASES:
            _lib.set_option("bwd_ablate", bits)
            us = time_kernel(call.bwd, iters=30) * 1e3
            lines.append(f"round {rnd} ablate {bits:3d} {name:52s} {us:8.1f}")
            print(lines[-1], flush=True)
    _lib.set_option("bwd_ablate", 0)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
