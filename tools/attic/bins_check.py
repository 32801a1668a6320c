#!/usr/bin/env python
"""Round-4 backward: the counting-sort gather kernel (`bwd_variant` 12, msda_bwd_bins.h) against the C oracle and
against the round-2 tile_lv kernel (`bwd_variant` 10) -- values, then times, at the encoder shape.

    python tools/bins_check.py [--quick] [--out gpurun_out/bins_check.txt]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, MsdaCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402


def oracle_bwd(x):
    from oracle import msda_oracle as oracle
    c = {k: v.detach().cpu().numpy() for k, v in x.items() if isinstance(v, torch.Tensor)}
    return oracle.backward(c["value"], c["shapes"], c["level_start"], c["loc"], c["attn"], c["grad_out"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default="gpurun_out/bins_check.txt")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    log = open(args.out, "w")

    def say(*a):
        s = " ".join(str(t) for t in a)
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()

    def setopts(**kw):
        for k, v in (("bwd_variant", 0), ("bwd_tile_margin", 4), ("bwd_bins_margin", 4), ("bwd_ablate", 0),
                     ("bwd_split", 1), ("bwd_bins_strip", 4)):
            _lib.set_option(k, v)
        for k, v in kw.items():
            _lib.set_option(k, v)

    # ---- values ----
    cases = [("encoder_like", 1.0, dict(height=800, width=1333)), ("uniform", 1.0, dict(height=800, width=1333)),
             ("encoder_like", 3.0, dict(height=800, width=1333)),
             ("encoder_like", 1.0, dict(height=720, width=1280)), ("encoder_like", 1.0, dict(height=400, width=667, batch=2))]
    if args.quick:
        cases = cases[:2]
    for dist, osc, kw in cases:
        x = make_inputs(dist=dist, off_scale=osc, device="cuda", seed=11, **kw)
        call, fcall = MsdaCall(x), FusedCall(x)
        rgv, rgl, rga = oracle_bwd(x)
        setopts(bwd_variant=1)
        fcall.bwd(); torch.cuda.synchronize()
        fgv, fgp = fcall.gv.clone(), fcall.gp.clone()
        for var, mg in ((10, 4), (12, 3), (12, 4), (12, 8)):
            setopts(bwd_variant=var, bwd_bins_margin=mg)
            call.gv.zero_(); call.gl.zero_(); call.ga.zero_()
            call.bwd(); torch.cuda.synchronize()
            k = _lib.last_kernel()
            e = [float(np.abs(a.cpu().numpy() - b).max()) for a, b in ((call.gv, rgv), (call.gl, rgl), (call.ga, rga))]
            fcall.gv.zero_(); fcall.gp.zero_()
            fcall.bwd(); torch.cuda.synchronize()
            fk = _lib.last_kernel()
            fe = [float((fcall.gv - fgv).abs().max()), float((fcall.gp - fgp).abs().max())]
            say(f"values {dist:12s} x{osc} {kw} v{var} m{mg}: vs oracle gv {e[0]:.2e} gl {e[1]:.2e} ga {e[2]:.2e} [{k}]"
                f" | fused vs generic gv {fe[0]:.2e} gp {fe[1]:.2e} [{fk}]")
    # ---- times (encoder shape) ----
    for dist, osc in (("encoder_like", 1.0), ("encoder_like", 2.0), ("encoder_like", 4.0), ("uniform", 1.0)):
        x = make_inputs(dist=dist, off_scale=osc, device="cuda")
        call, fcall = MsdaCall(x), FusedCall(x)
        cfgs = [("v10 tile_lv m4", dict(bwd_variant=10))]
        for mg in (3, 4, 6, 8, 10):
            cfgs.append((f"v12 bins m{mg}", dict(bwd_variant=12, bwd_bins_margin=mg)))
        if dist == "encoder_like":
            for mg in (4, 8):
                cfgs.append((f"v12 bins m{mg} raster order", dict(bwd_variant=12, bwd_bins_margin=mg, bwd_bins_strip=1)))
                cfgs.append((f"v12 bins m{mg} strips of 2", dict(bwd_variant=12, bwd_bins_margin=mg, bwd_bins_strip=2)))
        if osc == 1.0 and dist == "encoder_like":
            for ab in (1, 3, 4, 7):
                cfgs.append((f"v12 bins m4 ablate={ab}", dict(bwd_variant=12, bwd_ablate=ab)))
        if dist == "uniform":
            cfgs.append(("v1 generic", dict(bwd_variant=1)))
        for name, opts in cfgs:
            setopts(**opts)
            ms = time_kernel(call.bwd, iters=20)
            k = _lib.last_kernel()
            msf = time_kernel(fcall.bwd, iters=20)
            say(f"time {dist:12s} x{osc} {name:26s} plain {ms*1e3:8.1f} us   fused {msf*1e3:8.1f} us   [{k}]")
    setopts()
    log.close()


if __name__ == "__main__":
    main()
