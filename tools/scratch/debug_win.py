import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from memotr_amd import _lib
from memotr_amd import MultiScaleDeformableAttention as MSDA
from memotr_amd.MultiScaleDeformableAttention import tag_host_shapes
from memotr_amd.synth import make_inputs
torch.set_printoptions(precision=4, linewidth=200, sci_mode=False)
x = make_inputs(height=256, width=352, dist="encoder_like", device="cuda", seed=9)
tag_host_shapes(x["shapes"], x["shapes_list"])
N, S, M, D = x["value"].shape
def run(variant, **opts):
    _lib.set_option("fwd_variant", variant)
    for k, v in opts.items():
        _lib.set_option(k, v)
    out = MSDA.ms_deform_attn_forward(x["value"], x["shapes"], x["level_start"], x["loc"], x["attn"], 64)
    torch.cuda.synchronize()
    return out.view(N, -1, M, D)
for name, val in (("channel", torch.arange(D, device="cuda").float().view(1, 1, 1, D).expand(N, S, M, D).contiguous()),
                  ("pixel", torch.arange(S, device="cuda").float().view(1, S, 1, 1).expand(N, S, M, D).contiguous()),
                  ("randn", x["value"])):
    x["value"] = val
    ref = run(3)
    for l0 in (4, 1):
        got = run(12, fwd_win_l0=l0)
        print("=====", name, "l0", l0, _lib.last_kernel(), "max err", float((got - ref).abs().max()))
        for q in (0, 1, 500, 2000):
            for m in (0, 3):
                print(" q", q, "m", m, "ref", ref[0, q, m, :12].cpu().numpy().round(3))
                print(" q", q, "m", m, "got", got[0, q, m, :12].cpu().numpy().round(3))
