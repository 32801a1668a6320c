"""Forward kernels against the spread of the sampling offsets: the windowed kernel (variant 12) keeps the level 1-3
windows of a region in LDS and pays a per-point fallback for samples outside them; the gather kernel does not care.

    python tools/fwd_offset_sweep.py      -> profiles/r03_fwd_offset_sweep.txt

Two sweeps over memotr_amd.synth.make_inputs("encoder_like"): ``jitter`` scales the N(0, 1)-pixel noise around the
initial bias star (1.0 = what bench.py times); ``off_scale`` multiplies the whole offset, so the longest star arm is
4 * off_scale pixels (off_px = 2, 4, 8, 16 <-> off_scale = 0.5, 1, 2, 4).  Per setting: forward time of the gather and
the windowed kernel (each output checked against the generic kernel), the share of the windowed levels' points that
left their LDS window, and the default backward's time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import FusedCall, MsdaCall, time_kernel  # noqa: E402
from kbench import reset  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

print(f"{'jitter':>6s} {'variant':28s} {'plain us':>9s} {'fused us':>9s}  max|err| vs generic (plain, fused)")


def left_window_fraction(c):
    """Share of the live points of the windowed levels (1-3) that left their LDS window and took the global path
    (the kernel's profiling mode fwd_win_ablate = 6 writes 2 / 1 / 0 per (row, point) into `out` instead of the
    result)."""
    reset()
    _lib.set_option("fwd_variant", 12)
    _lib.set_option("fwd_win_ablate", 6)
    c.out.zero_(); c.fwd(); torch.cuda.synchronize()
    flags = c.out.view(-1, 32)[:, :16]
    left, served = int((flags == 2).sum()), int((flags == 1).sum())
    _lib.set_option("fwd_win_ablate", 0)
    return left / max(left + served, 1)


SWEEP = [("jitter", j, dict(jitter=j)) for j in (0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 8.0)] + \
        [("off_px", 4 * s, dict(off_scale=s)) for s in (0.5, 1.0, 2.0, 4.0)]
for kind, jit, kw in SWEEP:
    print(f"-- {kind} = {jit}")
    x = make_inputs(dist="encoder_like", device="cuda", **kw)
    call, fcall = MsdaCall(x), FusedCall(x)
    reset()
    _lib.set_option("fwd_variant", 1)
    refs = []
    for c in (call, fcall):
        c.fwd(); torch.cuda.synchronize(); refs.append(c.out.clone())
    for name, opts in (("v3 gather<4>", dict(fwd_variant=3)), ("v12 windows (default)", dict(fwd_variant=12)),
                       ("v0 auto", dict(fwd_variant=0))):
        reset()
        for k, v in opts.items():
            _lib.set_option(k, v)
        t, e = [], []
        for c, r in zip((call, fcall), refs):
            c.out.zero_(); c.fwd(); torch.cuda.synchronize()
            e.append(float((c.out - r).abs().max()))
            t.append(time_kernel(c.fwd, iters=50) * 1e3)
        print(f"{jit:6.2f} {name:28s} {t[0]:9.1f} {t[1]:9.1f}  {e[0]:.1e} {e[1]:.1e}   {_lib.last_kernel()}", flush=True)
    frac = left_window_fraction(fcall)
    reset()
    tb = []
    for c in (call, fcall):
        c.bwd(); torch.cuda.synchronize()
        tb.append(time_kernel(c.bwd, iters=20) * 1e3)
    off = x["loc"].float()
    print(f"{jit:6.2f} {'windowed points off-window':28s} {100 * frac:8.2f}%   backward (default) plain {tb[0]:7.1f} us  "
          f"fused {tb[1]:7.1f} us   {_lib.last_kernel()}", flush=True)
reset()
