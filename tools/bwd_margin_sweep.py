"""Default backward (region-tiled, fixed-point LDS windows) against the window margin and the offset magnitude ->
profiles/r03_bwd_margin_sweep.txt (run on the GPU box)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench import FusedCall, MsdaCall, time_kernel
from kbench import reset
from memotr_amd import _lib
from memotr_amd.synth import make_inputs
for s in (1.0, 1.5, 2.0, 4.0):
    x = make_inputs(dist="encoder_like", device="cuda", off_scale=s)
    call, fcall = MsdaCall(x), FusedCall(x)
    for mg in (3, 4, 5, 6):
        reset(); _lib.set_option("bwd_tile_margin", mg)
        t = []
        for c in (call, fcall):
            try:
                c.bwd(); torch.cuda.synchronize(); t.append(time_kernel(c.bwd, iters=20) * 1e3)
            except Exception as e:
                t.append(float("nan"))
        print(f"off_px {4*s:5.1f} margin {mg}: plain {t[0]:7.1f} us fused {t[1]:7.1f} us  {_lib.last_kernel()}", flush=True)
reset()
