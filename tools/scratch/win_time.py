"""time the windowed forward under option sets:  python tools/scratch/win_time.py "k=v,k=v" "k=v" ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import FusedCall, MsdaCall, time_kernel
from memotr_amd import _lib
from memotr_amd.synth import make_inputs
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from kbench import reset
jit = float(os.environ.get("JITTER", "1.0"))
x = make_inputs(dist="encoder_like", device="cuda", jitter=jit)
call, fcall = MsdaCall(x), FusedCall(x)
print("jitter", jit)
for spec in sys.argv[1:]:
    reset()
    _lib.set_option("fwd_win_early", 2)
    _lib.set_option("fwd_variant", 12)
    _lib.set_option("fwd_win_ablate", 0); _lib.set_option("fwd_win_wps", 0)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        _lib.set_option(k, int(v, 0))
    res = []
    for c in (call, fcall):
        c.fwd(); torch.cuda.synchronize()
        res.append((time_kernel(c.fwd, iters=50) * 1e3, _lib.last_kernel()))
    print(f"{spec:50s} plain {res[0][0]:7.1f} us  fused {res[1][0]:7.1f} us   {res[1][1]}", flush=True)
