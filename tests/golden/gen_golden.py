"""Generate the MSDeformAttn operator golden vectors by importing the REFERENCE.

Run in the build container only (needs /root/reference; the GPU box has none):

    python tests/golden/gen_golden.py

The reference's own statement of the operator is
``models/ops/functions/ms_deform_attn_func.py:44-64``
(``ms_deform_attn_core_pytorch``); the backward expected values are
``torch.autograd`` through it, which is exactly the relation the reference's
``models/ops/test.py:63-78`` gradchecks its CUDA kernel against.  The compiled
extension ``MultiScaleDeformableAttention`` is stubbed with an empty module so the
file imports without CUDA.  Only data (inputs + expected outputs) is written to
``tests/golden/msda_*.npz``; no reference source is copied.

Cases
  F1  models/ops/test.py shapes (N=1,M=2,D=2,Lq=2,L=2,P=2, pyramid (6,4),(3,2),
      manual_seed(3), value=rand*0.01, loc=rand, attn=rand+1e-5 normalised) in
      float64 and float32 -- the reference test's own vectors.
  F2  MeMOTR head geometry (M=8,D=32,L=4,P=4), pyramid (9,14),(5,7),(3,4),(2,2),
      N=2, Lq=40, loc in [-0.1,1.1] so every border/gate branch is exercised.
  F3  odd channel counts D in {30, 71} (reference gradcheck sizes, test.py:85) on a
      small pyramid -- exercises the generic (non-D=32) kernels.
  F4  L=1,P=1 and a single-pixel level (degenerate shapes).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def import_reference_core():
    sys.modules.setdefault("MultiScaleDeformableAttention", types.ModuleType("MultiScaleDeformableAttention"))
    sys.path.insert(0, os.path.join(REF, "models", "ops"))
    from functions.ms_deform_attn_func import ms_deform_attn_core_pytorch  # noqa: E402
    return ms_deform_attn_core_pytorch


def make_case(core, seed, N, M, D, Lq, P, shapes, dtype, loc_lo=0.0, loc_hi=1.0, value_scale=0.01):
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.as_tensor(shapes, dtype=torch.long)
    L = len(shapes)
    S = int((shapes_t[:, 0] * shapes_t[:, 1]).sum())
    lsi = torch.cat((shapes_t.new_zeros((1,)), shapes_t.prod(1).cumsum(0)[:-1]))
    value = (torch.rand(N, S, M, D, generator=g) * value_scale).to(dtype)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g) * (loc_hi - loc_lo) + loc_lo).to(dtype)
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    grad_out = torch.randn(N, Lq, M * D, generator=g).to(dtype)
    value.requires_grad_(True)
    loc.requires_grad_(True)
    attn.requires_grad_(True)
    out = core(value, shapes_t, loc, attn)
    gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), grad_out)
    return dict(
        value=value.detach().numpy(), shapes=shapes_t.numpy(), level_start=lsi.numpy(),
        loc=loc.detach().numpy(), attn=attn.detach().numpy(), grad_out=grad_out.numpy(),
        out=out.detach().numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy(),
    )


def test_py_case(core, dtype):
    """The literal sequence of models/ops/test.py:21-36 (seed 3, three rand() draws)."""
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes_t = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    lsi = torch.cat((shapes_t.new_zeros((1,)), shapes_t.prod(1).cumsum(0)[:-1]))
    S = int(shapes_t.prod(1).sum())
    torch.manual_seed(3)
    value = (torch.rand(N, S, M, D) * 0.01).to(dtype)
    loc = torch.rand(N, Lq, M, L, P, 2).to(dtype)
    attn = torch.rand(N, Lq, M, L, P) + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dtype)
    grad_out = torch.randn(N, Lq, M * D).to(dtype)
    value.requires_grad_(True)
    loc.requires_grad_(True)
    attn.requires_grad_(True)
    out = core(value, shapes_t, loc, attn)
    gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), grad_out)
    return dict(
        value=value.detach().numpy(), shapes=shapes_t.numpy(), level_start=lsi.numpy(),
        loc=loc.detach().numpy(), attn=attn.detach().numpy(), grad_out=grad_out.numpy(),
        out=out.detach().numpy(), grad_value=gv.numpy(), grad_loc=gl.numpy(), grad_attn=ga.numpy(),
    )


def main():
    core = import_reference_core()
    cases = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        cases[f"F1_testpy_{name}"] = test_py_case(core, dt)
        cases[f"F2_memotr_heads_{name}"] = make_case(
            core, 11, N=2, M=8, D=32, Lq=40, P=4, shapes=[(9, 14), (5, 7), (3, 4), (2, 2)],
            dtype=dt, loc_lo=-0.1, loc_hi=1.1, value_scale=1.0)
        cases[f"F3_D30_{name}"] = make_case(
            core, 12, N=1, M=2, D=30, Lq=9, P=2, shapes=[(6, 4), (3, 2)], dtype=dt, loc_lo=-0.2, loc_hi=1.2)
        cases[f"F3_D71_{name}"] = make_case(
            core, 13, N=3, M=3, D=71, Lq=5, P=3, shapes=[(5, 7), (2, 2), (1, 3)], dtype=dt, loc_lo=-0.2, loc_hi=1.2)
        cases[f"F4_L1P1_{name}"] = make_case(
            core, 14, N=2, M=4, D=32, Lq=7, P=1, shapes=[(1, 1)], dtype=dt, loc_lo=-0.5, loc_hi=1.5,
            value_scale=1.0)
        cases[f"F4_D64_{name}"] = make_case(
            core, 15, N=1, M=4, D=64, Lq=33, P=4, shapes=[(9, 12), (5, 6)], dtype=dt, loc_lo=-0.1, loc_hi=1.1,
            value_scale=1.0)
    for cname, arrs in cases.items():
        np.savez_compressed(os.path.join(OUT, f"msda_{cname}.npz"), **arrs)
        print(cname, {k: v.shape for k, v in arrs.items() if k in ("value", "loc", "out")})


if __name__ == "__main__":
    main()
