#!/usr/bin/env python
"""Launch one MSDeformAttn kernel variant a few times (encoder shape) -- the target of tools/pmc_probe.sh."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import MsdaCall
from memotr_amd import _lib
from memotr_amd.synth import make_inputs

op, variant = sys.argv[1], int(sys.argv[2])
margin = int(sys.argv[3]) if len(sys.argv) > 3 else 2
call = MsdaCall(make_inputs(dist="encoder_like", device="cuda"))
_lib.set_option(f"{op}_variant", variant)
_lib.set_option(f"{op}_tile_margin", margin)
fn = call.fwd if op == "fwd" else call.bwd
for _ in range(6):
    fn()
torch.cuda.synchronize()
print(_lib.last_kernel())
