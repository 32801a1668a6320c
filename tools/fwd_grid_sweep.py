#!/usr/bin/env python
"""Round 6 (last): the windowed forward in grid mode over region sizes (rsy x rsx level-0 pixels), encoder shape, fused,
N = 1 (and N = 5 for the best few): us per launch, regions per image, against the power-of-two default."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, time_kernel  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

x1 = make_inputs(device="cuda", batch=1)
c1 = FusedCall(x1)
_lib.set_option("fwd_variant", 1)
c1.fwd()
ref = c1.out.clone()
_lib.set_option("fwd_variant", 12)
_lib.set_option("sel_level", 0)
H0, W0 = [int(v) for v in x1["shapes"][0].tolist()]
res = []
for grid, rsy, rsx in [(0, 0, 0), (1, 0, 0)] + [(1, a, b) for a in range(8, 34) for b in range(8, 33)]:
    nreg = -(-H0 // rsy) * -(-W0 // rsx) if rsy else 0
    if rsy and not (48 <= nreg <= 64):
        continue
    _lib.set_option("fwd_win_grid", grid)
    _lib.set_option("fwd_win_rsy", rsy)
    _lib.set_option("fwd_win_rsx", rsx)
    for _ in range(4):
        c1.fwd()
    torch.cuda.synchronize()
    err = float((c1.out - ref).abs().max())
    us = time_kernel(c1.fwd, iters=60) * 1e3
    res.append((us, grid, rsy, rsx, nreg, err))
    print(f"grid={grid} rs {rsy:2d} x {rsx:2d}  regions {nreg:3d}  {us:6.1f} us  err {err:.1e}  {_lib.last_kernel()}", flush=True)
res.sort()
print("# best:", res[:8])
for k in ("fwd_win_grid", "fwd_win_rsy", "fwd_win_rsx"):
    _lib.set_option(k, {"fwd_win_grid": 1}.get(k, 0))
_lib.set_option("sel_level", -1)
_lib.set_option("fwd_variant", 0)
