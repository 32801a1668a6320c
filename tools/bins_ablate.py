import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import MsdaCall, time_kernel
from memotr_amd import _lib
from memotr_amd.synth import make_inputs
x = make_inputs(device="cuda")
call = MsdaCall(x)
_lib.set_option("bwd_variant", 12)
for name, ab in (("full",0),("stop after P0",8),("stop after P1",16),("no P2,P3,P4",64+32+2),("no P3,P4",32+2),("no P2,P4 (P3 only)",64+2),("no P4",2),("no P3",32),("no flush",1)):
    _lib.set_option("bwd_ablate", ab)
    ms = time_kernel(call.bwd, iters=20)
    print(f"ablate {ab:3d} {name:24s} {ms*1e3:8.1f} us", flush=True)
_lib.set_option("bwd_ablate", 0)
