#!/usr/bin/env python
"""Kernel micro-benchmark sweep for the MSDeformAttn HIP kernels (run on the GPU box).

    python tools/kbench.py [--quick] [--out gpurun_out/kbench.json]

For the BASELINE shapes (encoder Lq=S=22323; decoder Lq=300/320/400) and both location
distributions it times every forward/backward variant x block size x grid multiplier with HIP
events on the launch stream, checks each specialised variant against the generic kernel, and
prints achieved GB/s on the algorithmic bytes of SURVEY.md 8(d).
"""
import argparse
import itertools
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import MsdaCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default="gpurun_out/kbench.json")
    ap.add_argument("--fwd-variants", default="1,2,3,4,5")
    ap.add_argument("--bwd-variants", default="1,2,3,90,5,6,7")
    ap.add_argument("--blocks", default="64,256")
    ap.add_argument("--margins", default="1,2,3,4,5")
    ap.add_argument("--grid-mults", default="8,16,32")
    args = ap.parse_args()
    fv = [int(x) for x in args.fwd_variants.split(",")]
    bv = [int(x) for x in args.bwd_variants.split(",")]
    blocks = [int(x) for x in args.blocks.split(",")]
    gms = [int(x) for x in args.grid_mults.split(",")]
    margins = [int(x) for x in args.margins.split(",")]
    if args.quick:
        blocks, gms, margins = [256], [16], [2, 3]
    rows = []
    shapes = [("enc", None), ("dec320", 320)] if args.quick else [("enc", None), ("dec300", 300), ("dec400", 400)]
    for dist in ("encoder_like", "uniform"):
        for sname, nq in shapes:
            x = make_inputs(dist=dist, n_queries=nq, device="cuda")
            call = MsdaCall(x)
            # reference results from the generic kernels
            _lib.set_option("fwd_variant", 1)
            _lib.set_option("bwd_variant", 1)
            call.fwd(); call.bwd(); torch.cuda.synchronize()
            ref = (call.out.clone(), call.gv.clone(), call.gl.clone(), call.ga.clone())
            for v in fv:
                combos = [(256, 8)] if v == 1 else itertools.product(blocks, gms)
                if v == 5:
                    if nq is not None:
                        continue
                    combos = [(256, mg) for mg in margins]
                for blk, gm in combos:
                    _lib.set_option("fwd_variant", v); _lib.set_option("fwd_block", blk)
                    _lib.set_option("fwd_tile_margin" if v == 5 else "fwd_grid_mult", gm)
                    call.out.zero_(); call.fwd(); torch.cuda.synchronize()
                    err = float((call.out - ref[0]).abs().max())
                    ms = time_kernel(call.fwd, iters=30 if nq is None else 100)
                    gbps = call.bytes() / (ms * 1e-3) / 1e9
                    rows.append(dict(op="fwd", dist=dist, shape=sname, variant=v, block=blk, grid_mult=gm, ms=ms,
                                     GBps=gbps, frac=gbps / 8000, err=err, kernel=_lib.last_kernel()))
                    print(f"fwd {dist:12s} {sname:6s} v{v} blk{blk:4d} gm{gm:2d}  {ms*1e3:9.1f} us  {gbps:8.1f} GB/s "
                          f"({gbps/80:5.1f}%)  err {err:.1e}  {_lib.last_kernel()}", flush=True)
            for v in bv:
                combos = [(256, 8)] if v == 1 else itertools.product(blocks, gms)
                if v in (5, 6, 7):
                    if nq is not None:
                        continue
                    combos = [(256, mg) for mg in margins]
                for blk, gm in combos:
                    _lib.set_option("bwd_variant", v); _lib.set_option("bwd_block", blk)
                    _lib.set_option("bwd_tile_margin" if v in (5, 6, 7) else "bwd_grid_mult", gm)
                    call.bwd(); torch.cuda.synchronize()
                    errs = [float((a - b).abs().max()) for a, b in zip((call.gv, call.gl, call.ga), ref[1:])]
                    ms = time_kernel(call.bwd, iters=10 if nq is None else 50)
                    gbps = call.bytes(True) / (ms * 1e-3) / 1e9
                    rows.append(dict(op="bwd", dist=dist, shape=sname, variant=v, block=blk, grid_mult=gm, ms=ms,
                                     GBps=gbps, frac=gbps / 8000, err=errs, kernel=_lib.last_kernel()))
                    print(f"bwd {dist:12s} {sname:6s} v{v} blk{blk:4d} gm{gm:2d}  {ms*1e3:9.1f} us  {gbps:8.1f} GB/s "
                          f"({gbps/80:5.1f}%)  err {errs[0]:.1e} {errs[1]:.1e} {errs[2]:.1e}  {_lib.last_kernel()}",
                          flush=True)
            _lib.set_option("fwd_variant", 0); _lib.set_option("bwd_variant", 0)
    # memset + copy baselines for context (same bytes as value)
    v = torch.empty(22323 * 8 * 32, device="cuda")
    w = torch.empty_like(v)
    ms_copy = time_kernel(lambda: w.copy_(v), iters=100)
    ms_zero = time_kernel(lambda: w.zero_(), iters=100)
    print(f"copy 22.9MB {ms_copy*1e3:.1f} us ({2*v.numel()*4/ms_copy/1e6:.0f} GB/s)  zero {ms_zero*1e3:.1f} us")
    rows.append(dict(op="copy", ms=ms_copy)); rows.append(dict(op="zero", ms=ms_zero))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
