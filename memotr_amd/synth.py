"""Synthetic inputs for the MSDeformAttn hot path (benchmarks, smoke, size-independent tests).

Shapes follow BASELINE.json / SURVEY.md section 8: an 800x1333 frame padded to a multiple of 32
(utils/nested_tensor.py:41-50 in the reference) gives the pyramid
[[100,168],[50,84],[25,42],[13,21]], S = 22323; M = 8 heads, D = 32, L = P = 4.

Two location distributions:
  * ``uniform``       loc ~ U[0,1)  -- the reference's own test distribution
                      (models/ops/test.py:33), worst-case locality.
  * ``encoder_like``  pixel-centre reference points of the encoder
                      (models/deformable_encoder.py:29-40) + the 8-direction star of
                      MSDeformAttn.reset_parameters (models/ops/modules/ms_deform_attn.py:72-86)
                      + N(0, jitter) pixels.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch


def pad32(x: int) -> int:
    return (x + 31) // 32 * 32


def pyramid_shapes(height: int, width: int, levels: int = 4) -> List[Tuple[int, int]]:
    """Feature-map sizes of ResNet strides 8/16/32 plus 3x3-stride-2 extra levels on the padded frame."""
    h, w = pad32(height), pad32(width)
    for _ in range(3):           # conv1 s2, maxpool s2, layer2 s2  (each ceil(x/2))
        h, w = (h + 1) // 2, (w + 1) // 2
    shapes = [(h, w)]
    for _ in range(levels - 1):  # layer3, layer4, extra conv(s)
        h, w = (h + 1) // 2, (w + 1) // 2
        shapes.append((h, w))
    return shapes


def level_start_index(shapes: Sequence[Tuple[int, int]]) -> List[int]:
    out, acc = [], 0
    for h, w in shapes:
        out.append(acc)
        acc += h * w
    return out


def valid_ratios(height: int, width: int, shapes: Sequence[Tuple[int, int]]) -> torch.Tensor:
    """(L, 2) = (ratio_w, ratio_h) of the un-padded area per level, nearest-neighbour mask down-sampling
    as models/backbone.py:96 + models/deformable_transformer.py:175-190."""
    hp, wp = pad32(height), pad32(width)
    mask = torch.ones(1, 1, hp, wp)
    mask[..., :height, :width] = 0
    out = []
    for h, w in shapes:
        m = torch.nn.functional.interpolate(mask, size=(h, w)).to(torch.bool)[0, 0]
        valid_h = int((~m[:, 0]).sum())
        valid_w = int((~m[0, :]).sum())
        out.append((valid_w / w, valid_h / h))
    return torch.tensor(out, dtype=torch.float32)


def encoder_reference_points(shapes: Sequence[Tuple[int, int]], vr: torch.Tensor) -> torch.Tensor:
    """(S, L, 2) reference points, formula of DeformableEncoder.get_reference_points (batch 1)."""
    refs = []
    for lvl, (h, w) in enumerate(shapes):
        ys = torch.linspace(0.5, h - 0.5, h)
        xs = torch.linspace(0.5, w - 0.5, w)
        ry, rx = torch.meshgrid(ys, xs, indexing="ij")
        ry = ry.reshape(-1) / (vr[lvl, 1] * h)
        rx = rx.reshape(-1) / (vr[lvl, 0] * w)
        refs.append(torch.stack((rx, ry), -1))
    ref = torch.cat(refs, 0)
    return ref[:, None, :] * vr[None, :, :]


def star_offsets(n_heads: int, n_levels: int, n_points: int) -> torch.Tensor:
    """(M, L, P, 2) pixel offsets of the reference's sampling_offsets bias initialisation."""
    thetas = torch.arange(n_heads, dtype=torch.float32) * (2.0 * math.pi / n_heads)
    grid = torch.stack([thetas.cos(), thetas.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(n_heads, 1, 1, 2).repeat(1, n_levels, n_points, 1)
    for i in range(n_points):
        grid[:, :, i, :] *= i + 1
    return grid


def make_inputs(height: int = 800, width: int = 1333, n_queries: int | None = None, batch: int = 1,
                n_heads: int = 8, head_dim: int = 32, n_levels: int = 4, n_points: int = 4,
                dist: str = "encoder_like", jitter: float = 1.0, seed: int = 3, dtype=torch.float32,
                device="cpu", value_dist: str = "normal", off_scale: float = 1.0):
    """Returns dict(value, shapes, level_start, loc, attn, grad_out, shapes_list).

    ``n_queries=None`` means one query per pyramid pixel (encoder self-attention, Lq = S);
    otherwise decoder-style queries with random reference boxes.  ``off_scale`` multiplies the whole sampling offset
    (star + noise) of the "encoder_like" pattern: 1.0 is the initialisation (star arms of 1..4 pixels), a trained
    model's offsets are larger (tools/fwd_offset_sweep.py).
    """
    g = torch.Generator().manual_seed(seed)
    shapes = pyramid_shapes(height, width, n_levels)
    S = sum(h * w for h, w in shapes)
    M, D, L, P = n_heads, head_dim, n_levels, n_points
    if value_dist == "normal":
        value = torch.randn(batch, S, M, D, generator=g)
    else:  # models/ops/test.py:31
        value = torch.rand(batch, S, M, D, generator=g) * 0.01
    Lq = S if n_queries is None else n_queries
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)  # (L,2) as (W,H)
    if dist == "uniform":
        loc = torch.rand(batch, Lq, M, L, P, 2, generator=g)
    elif dist == "encoder_like":
        vr = valid_ratios(height, width, shapes)
        if n_queries is None:
            ref = encoder_reference_points(shapes, vr)                       # (S,L,2)
        else:
            centre = torch.rand(Lq, 1, 2, generator=g) * 0.8 + 0.1
            ref = centre * vr[None]                                           # (Lq,L,2)
        off = star_offsets(M, L, P)[None] + jitter * torch.randn(Lq, M, L, P, 2, generator=g)
        off = off * off_scale
        loc = ref[:, None, :, None, :] + off / wh[None, None, :, None, :]
        loc = loc[None].expand(batch, -1, -1, -1, -1, -1).contiguous()
    else:
        raise ValueError(dist)
    attn = torch.softmax(torch.rand(batch, Lq, M, L * P, generator=g), -1).view(batch, Lq, M, L, P)
    grad_out = torch.randn(batch, Lq, M * D, generator=g)
    shapes_t = torch.tensor(shapes, dtype=torch.int64)
    lsi = torch.tensor(level_start_index(shapes), dtype=torch.int64)
    out = dict(value=value.to(dtype), loc=loc.float(), attn=attn.float(), grad_out=grad_out.to(dtype))
    if dtype == torch.float64:
        out["loc"], out["attn"] = loc.double(), attn.double()
    out = {k: v.contiguous().to(device) for k, v in out.items()}
    out["shapes"] = shapes_t.to(device)
    out["level_start"] = lsi.to(device)
    out["shapes_list"] = shapes
    return out


def to_fused_inputs(x: dict) -> dict:
    """The same sampling pattern as ``x`` (a ``make_inputs`` result) expressed as the fused entry points' inputs:
    reference points = mean location per (query, level), offsets = (loc - ref) * (W, H), logits = log(attn), packed
    as projection rows [offsets (M,L,P,2) | logits (M,L,P)].  Returns dict(proj, ref)."""
    loc, attn = x["loc"].float(), x["attn"].float()
    N, Lq, M, L, P, _ = loc.shape
    wh = torch.tensor([[w, h] for h, w in x["shapes_list"]], dtype=torch.float32, device=loc.device)
    ref = loc.mean(dim=(2, 4))                                                   # (N, Lq, L, 2)
    off = (loc - ref[:, :, None, :, None, :]) * wh[None, None, None, :, None, :]
    logits = attn.clamp_min(1e-30).log()
    proj = torch.cat((off.reshape(N, Lq, M * L * P * 2), logits.reshape(N, Lq, M * L * P)), -1).contiguous()
    return dict(proj=proj, ref=ref.contiguous())


def algorithmic_bytes(batch: int, S: int, Lq: int, M: int, D: int, L: int, P: int, value_size: int = 4,
                      backward: bool = False) -> int:
    """bytes one call must move.  The gather cannot read more of ``value``
    than it touches, so the value term is min(all of value, 4 corners of every point)."""
    value_el = min(batch * S * M * D, batch * Lq * M * L * P * 4 * D)
    loc_b = 4 * 2 * batch * Lq * M * L * P
    attn_b = 4 * batch * Lq * M * L * P
    out_b = value_size * batch * Lq * M * D
    fwd = value_size * value_el + loc_b + attn_b + out_b + 8 * 3 * L
    if not backward:
        return fwd
    # reads: grad_out + value + loc + attn; writes: grad_value (fp32 accum when bf16) + grad_loc + grad_attn
    gv = (4 if value_size == 2 else value_size) * batch * S * M * D
    return fwd + gv + loc_b + attn_b
