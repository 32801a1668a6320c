"""Price the parts of the tiled backward (msda_set_option "bwd_ablate": 1 no flush, 2 no LDS scatter, 4 no value loads):
encoder shape, HIP-event timing.  Results with a non-zero mask are wrong by construction -- profiling only."""
import sys, torch
sys.path.insert(0, ".")
from bench import MsdaCall, FusedCall, time_kernel
from memotr_amd import _lib
from memotr_amd.synth import make_inputs
x = make_inputs(dist="encoder_like", device="cuda")
call = MsdaCall(x)
for pts in (8, 10, 11):
    _lib.set_option("bwd_variant", pts)
    for ab in (0, 1, 2, 3, 4, 5, 6, 7):
        _lib.set_option("bwd_ablate", ab)
        ms = time_kernel(call.bwd, iters=20)
        print(f"variant {pts} ablate {ab} (1 no flush, 2 no scatter, 4 no value loads): {ms*1e3:.1f} us  {_lib.last_kernel()}", flush=True)
_lib.set_option("bwd_ablate", 0)
# memset alone
gv = call.gv
print("memset", time_kernel(lambda: gv.zero_(), iters=50)*1e3, "us")
