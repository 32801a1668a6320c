"""Test infrastructure: the fused prologue of MSDeformAttn restated with torch CPU ops (reference module order,
models/ops/modules/ms_deform_attn.py:104-123) + the C oracle + autograd through the prologue.  Pinned against the
reference's own module goldens by tests/test_fused_reference_cpu.py."""
import numpy as np
import torch


def prologue_cpu(proj, ref, shapes, M, L, P):
    """loc (N,Lq,M,L,P,2), attn (N,Lq,M,L,P) with the module's operations (ms_deform_attn.py:109-122), float32."""
    N, Lq = proj.shape[:2]
    n_off = 2 * M * L * P
    off = proj[..., :n_off].reshape(N, Lq, M, L, P, 2)
    logits = proj[..., n_off:n_off + M * L * P].reshape(N, Lq, M, L * P)
    attn = torch.softmax(logits, -1).reshape(N, Lq, M, L, P)
    if ref.shape[-1] == 2:
        wh = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(proj.dtype)
        loc = ref[:, :, None, :, None, :] + off / wh[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2] + off / P * ref[:, :, None, :, None, 2:] * 0.5
    return loc, attn


def make_case(seed, N, M, D, L, P, shapes, Lq=None, ref_dim=2, with_mask=True, pyramid=False, off_px=3.0):
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.tensor(shapes, dtype=torch.int64)
    S = int(shapes_t.prod(1).sum())
    lsi = torch.cat((shapes_t.new_zeros(1), shapes_t.prod(1).cumsum(0)[:-1]))
    value = torch.randn(N, S, M, D, generator=g)
    if pyramid:           # one query per pixel, reference point = its own pixel centre on every level
        Lq = S
        refs = []
        for (h, w) in shapes:
            ys, xs = torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w, indexing="ij")
            refs.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], -1))
        ref = torch.cat(refs, 0)[None, :, None, :].expand(N, S, L, 2).contiguous()
        if ref_dim == 4:
            ref = torch.cat([ref, torch.rand(N, S, L, 2, generator=g) * 0.2 + 0.05], -1)
    else:
        ref = torch.rand(N, Lq, L, ref_dim, generator=g)
        if ref_dim == 4:
            ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
    width = 3 * M * L * P
    proj = torch.randn(N, Lq, width, generator=g)
    proj[..., :2 * M * L * P] *= off_px if ref_dim == 2 else 1.5
    mask = None
    if with_mask:
        mask = torch.zeros(N, S, dtype=torch.bool)
        mask[-1, S // 3: S // 3 + max(2, S // 9)] = True
        mask[-1, S - 5:] = True
    grad_out = torch.randn(N, Lq, M * D, generator=g)
    return dict(value=value, shapes=shapes_t, level_start=lsi, proj=proj.contiguous(), ref=ref.contiguous(),
                mask=mask, grad_out=grad_out, M=M, L=L, P=P, shapes_list=list(shapes))


def expected(c, need_ref=True):
    """Oracle forward/backward on the CPU-restated prologue + autograd through the prologue."""
    from oracle import msda_oracle as oracle
    M, L, P = c["M"], c["L"], c["P"]
    proj = c["proj"].detach().clone().requires_grad_(True)
    ref = c["ref"].detach().clone().requires_grad_(True)
    loc, attn = prologue_cpu(proj, ref, c["shapes"], M, L, P)
    value = c["value"].detach()
    if c["mask"] is not None:
        value = value.masked_fill(c["mask"][..., None, None], 0.0)
    args = (value.contiguous().numpy(), c["shapes"].numpy(), c["level_start"].numpy(), loc.detach().contiguous().numpy(),
            attn.detach().contiguous().numpy())
    out = oracle.forward(*args)
    gv, gl, ga = oracle.backward(*args, c["grad_out"].detach().contiguous().numpy())
    if c["mask"] is not None:
        gv[c["mask"].numpy()] = 0.0
    torch.autograd.backward([loc, attn], [torch.from_numpy(gl), torch.from_numpy(ga)])
    return dict(out=out, grad_value=gv, grad_proj=proj.grad.numpy(), grad_ref=ref.grad.numpy(),
                loc=loc.detach().numpy(), attn=attn.detach().numpy())


