import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import MsdaCall, FusedCall, time_kernel
from memotr_amd import _lib
from memotr_amd.synth import make_inputs
x = make_inputs(device="cuda")
call, fcall = MsdaCall(x), FusedCall(x)
for rep in range(2):
    for auto in (0, 1):
        _lib.set_option("auto_select", auto)
        _lib.set_option("fwd_variant", 12)
        a = time_kernel(call.fwd, iters=100); b = time_kernel(fcall.fwd, iters=100)
        _lib.set_option("fwd_variant", 3)
        c = time_kernel(fcall.fwd, iters=100)
        _lib.set_option("bwd_variant", 12)
        d = time_kernel(call.bwd, iters=40)
        print(f"auto_select={auto}: win plain {a*1e3:.1f} fused {b*1e3:.1f} | gather fused {c*1e3:.1f} | bins bwd {d*1e3:.1f}", flush=True)
