#!/usr/bin/env python
"""Kernel micro-benchmark sweep for the MSDeformAttn HIP kernels (run on the GPU box).

    python tools/kbench.py [--quick] [--out gpurun_out/kbench.json]

For the BASELINE shapes (encoder Lq=S=22323; decoder Lq=320) and both location distributions it times the
forward / backward variants (plain operator and fused prologue) with HIP events on the launch stream, checks each
against the generic kernel, and prints achieved GB/s on the algorithmic bytes of SURVEY.md 8(d).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, MsdaCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402


def reset():
    for k, v in (("fwd_variant", 0), ("bwd_variant", 0), ("bwd_tile_margin", 4),
                 ("fwd_block", 256), ("fwd_grid_mult", 32), ("bwd_split", 1),
                 ("fwd_win_rlog", 0), ("fwd_win_rlogx", 0), ("fwd_win_block", 0), ("fwd_win_l0", 1), ("fwd_win_margins", 0x3333),
                 ("fwd_head_major", 0), ("fwd_win_early", 9), ("fwd_win_wps", 0),
                 ("fwd_win_ablate", 0), ("bwd_rows_block", 0)):
        _lib.set_option(k, v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default="gpurun_out/kbench.json")
    ap.add_argument("--dists", default="encoder_like,uniform")
    ap.add_argument("--fwd-only", action="store_true")
    args = ap.parse_args()
    rows = []

    def record(**kw):
        rows.append(kw)
        err = kw.get("err")
        errs = " ".join(f"{e:.1e}" for e in err) if isinstance(err, (list, tuple)) else f"{err:.1e}"
        print(f"{kw['op']:9s} {kw['dist']:12s} {kw['shape']:6s} {kw['cfg']:28s} {kw['ms']*1e3:9.1f} us "
              f"{kw['GBps']:8.1f} GB/s ({kw['GBps']/80:5.1f}%)  err {errs}  {kw['kernel']}", flush=True)

    for dist in args.dists.split(","):
        for sname, nq in (("enc", None), ("dec320", 320)):
            x = make_inputs(dist=dist, n_queries=nq, device="cuda")
            call, fcall = MsdaCall(x), FusedCall(x)
            reset()
            _lib.set_option("fwd_variant", 1)
            _lib.set_option("bwd_variant", 1)
            call.fwd(); call.bwd(); torch.cuda.synchronize()
            ref = (call.out.clone(), call.gv.clone(), call.gl.clone(), call.ga.clone())
            fcall.fwd(); fcall.bwd(); torch.cuda.synchronize()
            fref = (fcall.out.clone(), fcall.gv.clone(), fcall.gp.clone())
            enc = nq is None
            fwd_cfgs = [("v1 generic", dict(fwd_variant=1)), ("v3 gather<4>", dict(fwd_variant=3))]
            if enc:
                fwd_cfgs.append(("v3 gather<4> head-major", dict(fwd_variant=3, fwd_head_major=1)))
            if enc:
                # windowed forward (variant 12): region size x workgroup size x margins x windowed levels x fill
                win = [dict()]
                if not args.quick:
                    win += [dict(fwd_win_rlog=3, fwd_win_block=256), dict(fwd_win_rlog=3, fwd_win_block=512),
                            dict(fwd_win_rlog=3, fwd_win_rlogx=4, fwd_win_block=512), dict(fwd_win_early=0),
                            dict(fwd_win_margins=0x2333),
                            dict(fwd_win_margins=0x2222), dict(fwd_win_margins=0x2233), dict(fwd_win_margins=0x4444),
                            dict(fwd_win_margins=0x2330), dict(fwd_win_l0=0), dict(fwd_win_l0=0, fwd_win_margins=0x2222),
                            dict(fwd_win_l0=2), dict(fwd_win_l0=4)]
                for w in win:
                    tag = " ".join(f"{k[8:]}={v:x}" for k, v in w.items()) or "default"
                    fwd_cfgs.append((f"v12 win {tag}", dict(fwd_variant=12, **w)))
            for name, opts in fwd_cfgs:
                for c, r, tag in ((call, ref[0], "fwd"), (fcall, fref[0], "fwd_fused")):
                    reset()
                    for k, v in opts.items():
                        _lib.set_option(k, v)
                    c.out.zero_(); c.fwd(); torch.cuda.synchronize()
                    err = float((c.out - r).abs().max())
                    ms = time_kernel(c.fwd, iters=50 if enc else 100)
                    gbps = c.bytes() / (ms * 1e-3) / 1e9
                    record(op=tag, dist=dist, shape=sname, cfg=name, ms=ms, GBps=gbps, err=err,
                           kernel=_lib.last_kernel())
            bwd_cfgs = [("v1 generic", dict(bwd_variant=1))]
            if not enc:
                bwd_cfgs.append(("v0 rows (32 lanes/row)", dict(bwd_variant=0)))
                bwd_cfgs.append(("v0 rows block=128", dict(bwd_variant=0, bwd_rows_block=128)))
                bwd_cfgs.append(("v0 rows block=256", dict(bwd_variant=0, bwd_rows_block=256)))
            if enc:
                for v in (10,):
                    for mg in ((3, 4, 5) if not args.quick else (4,)):
                        bwd_cfgs.append((f"v{v} tile_lv m{mg}", dict(bwd_variant=v, bwd_tile_margin=mg)))
                        # fused call without the three-kernel split (the plain call ignores the knob)
                        bwd_cfgs.append((f"v{v} tile_lv m{mg} one-kernel", dict(bwd_variant=v, bwd_tile_margin=mg,
                                                                                 bwd_split=0)))
                bwd_cfgs.append(("v12 bins", dict(bwd_variant=12)))
            for name, opts in ([] if args.fwd_only else bwd_cfgs):
                for c, tag in ((call, "bwd"), (fcall, "bwd_fused")):
                    reset()
                    for k, v in opts.items():
                        _lib.set_option(k, v)
                    c.bwd(); torch.cuda.synchronize()
                    if tag == "bwd":
                        errs = [float((a - b).abs().max()) for a, b in zip((c.gv, c.gl, c.ga), ref[1:])]
                    else:
                        errs = [float((a - b).abs().max()) for a, b in zip((c.gv, c.gp), fref[1:])]
                    ms = time_kernel(c.bwd, iters=20 if enc else 50)
                    gbps = c.bytes(True) / (ms * 1e-3) / 1e9
                    record(op=tag, dist=dist, shape=sname, cfg=name, ms=ms, GBps=gbps, err=errs,
                           kernel=_lib.last_kernel())
            reset()
    # bf16 storage (BASELINE config 5 extension): value / out / grad_out bf16, everything else fp32
    for dist in args.dists.split(","):
        x = make_inputs(dist=dist, device="cuda")
        call = MsdaCall(x)
        vb, gob = x["value"].bfloat16().contiguous(), x["grad_out"].bfloat16().contiguous()
        outb = torch.empty(call.N, call.Lq, call.M * call.D, device="cuda", dtype=torch.bfloat16)
        st = lambda: torch.cuda.current_stream().cuda_stream

        def fwd16():
            rc = _lib.lib.msda_forward_bf16(vb.data_ptr(), x["shapes"].data_ptr(), x["level_start"].data_ptr(),
                                            x["loc"].data_ptr(), x["attn"].data_ptr(), call.N, call.S, call.M, call.D,
                                            call.L, call.Lq, call.P, outb.data_ptr(), call.hptr, st())
            assert rc == 0, _lib.last_error()

        def bwd16():
            rc = _lib.lib.msda_backward_bf16(vb.data_ptr(), x["shapes"].data_ptr(), x["level_start"].data_ptr(),
                                             x["loc"].data_ptr(), x["attn"].data_ptr(), gob.data_ptr(), call.N, call.S,
                                             call.M, call.D, call.L, call.Lq, call.P, call.gv.data_ptr(),
                                             call.gl.data_ptr(), call.ga.data_ptr(), 1, call.hptr, st())
            assert rc == 0, _lib.last_error()

        reset()
        call.fwd(); torch.cuda.synchronize()
        fwd16(); torch.cuda.synchronize()
        err = float((outb.float() - call.out).abs().max())
        ms = time_kernel(fwd16, iters=50)
        from memotr_amd.synth import algorithmic_bytes
        by = algorithmic_bytes(call.N, call.S, call.Lq, call.M, call.D, call.L, call.P, 2)
        record(op="fwd_bf16", dist=dist, shape="enc", cfg="default", ms=ms, GBps=by / (ms * 1e-3) / 1e9, err=err,
               kernel=_lib.last_kernel())
        ms = time_kernel(bwd16, iters=20)
        by = algorithmic_bytes(call.N, call.S, call.Lq, call.M, call.D, call.L, call.P, 2, True)
        record(op="bwd_bf16", dist=dist, shape="enc", cfg="default", ms=ms, GBps=by / (ms * 1e-3) / 1e9, err=0.0,
               kernel=_lib.last_kernel())
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
