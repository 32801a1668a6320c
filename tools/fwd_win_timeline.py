#!/usr/bin/env python
"""Where a workgroup of the windowed forward (msda_fwd_d32_win) spends its life: s_memtime stamps per wavefront
(`fwd_win_trace_*` options; profiling build path only), averaged over the launch.

    python tools/fwd_win_timeline.py [plain] [--out gpurun_out/fwd_win_timeline.txt]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, MsdaCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

NAMES = ["tables+sync", "first loads + offsets", "sync", "placement (3 threads)", "sync", "fill issue"]


def main():
    plain = "plain" in sys.argv
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else "gpurun_out/fwd_win_timeline.txt"
    x = make_inputs(device="cuda")
    call = MsdaCall(x) if plain else FusedCall(x)
    _lib.set_option("fwd_variant", 12)
    _lib.set_option("sel_level", 0)
    for w in sys.argv[1:]:
        if "=" in w:
            k, v = w.split("=")
            _lib.set_option(k, int(v, 0))
    ms = time_kernel(call.fwd, iters=50)
    rl = _lib.get_option("fwd_win_rlog") or 4
    rlx = _lib.get_option("fwd_win_rlogx") or rl
    thr = _lib.get_option("fwd_win_block") or (512 if _lib.get_option("fwd_win_rlog") == 0 else 256)
    h0, w0 = x["shapes_list"][0]
    n_wg, nw = -(-h0 // (1 << rl)) * -(-w0 // (1 << rlx)) * 8, thr // 64
    for ab, what in ((1, "prologue only (tables, first loads, placement, fill)"), (2, "no gather (prologue + staging + records)")):
        _lib.set_option("fwd_win_ablate", ab)
        print(f"ablate {ab}: {time_kernel(call.fwd, iters=50)*1e3:.1f} us  -- {what}")
    _lib.set_option("fwd_win_ablate", 0)
    buf = torch.zeros(n_wg * nw * 32, dtype=torch.int64, device="cuda")
    p = buf.data_ptr()
    _lib.set_option("fwd_win_trace_lo", p & 0x7FFFFFFF)
    _lib.set_option("fwd_win_trace_hi", p >> 31)
    call.fwd()
    torch.cuda.synchronize()
    _lib.set_option("fwd_win_trace_lo", 0)
    _lib.set_option("fwd_win_trace_hi", 0)
    t = buf.cpu().numpy().reshape(n_wg, nw, 32).astype(np.float64)
    lines = [f"msda_fwd_d32_win {'plain' if plain else 'fused'}: {ms*1e3:.1f} us per launch untraced; kernel {_lib.last_kernel()}"]
    t0 = t[:, :, 0:1]
    span = (t.max(axis=2) - t[:, :, 0])
    lines.append(f"wavefront lifetime: mean {span.mean():.0f} cycles, min {span.min():.0f}, max {span.max():.0f} "
                 f"({n_wg} workgroups x {nw} wavefronts; s_memtime ticks)")
    launch = t[:, :, 0].min()
    lines.append(f"launch span (first stamp -> last stamp anywhere): {(t.max() - launch):.0f} ticks; "
                 f"workgroup start times: median {np.median(t[:, 0, 0] - launch):.0f}, 90 % {np.percentile(t[:, 0, 0] - launch, 90):.0f}")
    d = np.diff(t, axis=2)
    for k in range(len(NAMES)):
        lines.append(f"  {NAMES[k]:28s} {d[:, :, k].mean():8.0f}")
    for it in range(8):
        base = len(NAMES) + 3 * it
        valid = t[:, :, base + 3] > 0
        if not valid.any():
            break
        names = ("stage", "records visible / early loads" + (" / windows landed" if it == 0 else ""), "gather + store")
        for j in range(3):
            v = d[:, :, base + j][valid]
            lines.append(f"  step {it} {names[j]:44s} {v.mean():8.0f}   (p10 {np.percentile(v, 10):.0f}, p90 {np.percentile(v, 90):.0f})")
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    with open(out, "a") as f:
        f.write("\n".join(lines) + "\n\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
