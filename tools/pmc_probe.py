#!/usr/bin/env python
"""Launch one MSDeformAttn kernel configuration a few times (encoder shape; fused-prologue entry points -- what the
model issues -- unless `plain` is given) -- the target of tools/pmc_probe.sh.

    pmc_probe.py fwd|bwd [plain] [bf16] [uniform] key=value ...      e.g.  pmc_probe.py fwd fwd_variant=12 fwd_win_rlog=4
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import FusedCall, FusedCallBf16, MsdaCall  # noqa: E402
from memotr_amd import _lib  # noqa: E402
from memotr_amd.synth import make_inputs  # noqa: E402

op = sys.argv[1]
words = sys.argv[2:]
dist = "uniform" if "uniform" in words else "encoder_like"
x = make_inputs(dist=dist, device="cuda")
call = MsdaCall(x) if "plain" in words else (FusedCallBf16(x) if "bf16" in words else FusedCall(x))
for w in words:
    if "=" in w:
        k, v = w.split("=")
        _lib.set_option(k, int(v, 0))
if op == "bwd" and hasattr(call, "have_out"):
    call.fwd()        # (the backward of the model's call gets the forward's output)
fn = call.fwd if op == "fwd" else call.bwd
for _ in range(6):
    fn()
torch.cuda.synchronize()
print(_lib.last_kernel())
