#!/usr/bin/env python
"""Launch one MSDeformAttn kernel variant a few times (encoder shape, fused-prologue entry points -- what the model
issues) -- the target of tools/pmc_probe.sh.   usage: pmc_probe.py fwd|bwd <variant> [margin]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

op, variant = sys.argv[1], int(sys.argv[2])
margin = int(sys.argv[3]) if len(sys.argv) > 3 else 3
call = FusedCall(make_inputs(dist="encoder_like", device="cuda"))
_lib.set_option(f"{op}_variant", variant)
_lib.set_option(f"{op}_tile_margin", margin)
fn = call.fwd if op == "fwd" else call.bwd
for _ in range(6):
    fn()
torch.cuda.synchronize()
print(_lib.last_kernel())
