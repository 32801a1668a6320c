"""Self-attention over the decoder queries with the parameters of an ``nn.MultiheadAttention``.

The reference calls ``nn.MultiheadAttention(batch_first=True)`` with query = key = tgt + pos and value = tgt
(models/deformable_decoder.py, self_attn / track_attn).  ``F.multi_head_attention_forward`` spends ~200 us of host
time per call on argument checks and runs three separate input projections in that case; the decoder is bound by
the host's launch rate, so this does the same computation with the query and key projections as ONE GEMM (they
share their input), the value projection, fused scaled-dot-product attention and the output projection -- the
module keeps owning the parameters, so state-dict keys do not change.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class _PackedInProj(torch.autograd.Function):
    """(qk W_qk^T + b_qk, v W_v^T + b_v) from the packed ``in_proj_weight`` (3E, E) / ``in_proj_bias`` (3E).

    Written as one autograd node because slicing the packed parameters in the graph costs more in the backward than
    the projections themselves: every slice gets a zero-filled full-size gradient, a strided copy and an add into the
    parameter's buffer (~20 small kernels per call against the 6 below, x 6 layers x every frame of a clip)."""

    @staticmethod
    def forward(ctx, qk, v, w, b):
        E = w.shape[1]
        qk2, v2 = qk.reshape(-1, E), v.reshape(-1, E)
        qk_p = torch.addmm(b[:2 * E], qk2, w[:2 * E].t())
        v_p = torch.addmm(b[2 * E:], v2, w[2 * E:].t())
        ctx.save_for_backward(qk2, v2, w)
        ctx.shapes = (qk.shape, v.shape)
        return qk_p.view(*qk.shape[:-1], 2 * E), v_p.view(*v.shape[:-1], E)

    @staticmethod
    def backward(ctx, g_qk, g_v):
        qk2, v2, w = ctx.saved_tensors
        E = w.shape[1]
        g_qk2, g_v2 = g_qk.reshape(-1, 2 * E), g_v.reshape(-1, E)
        gw, gb = torch.empty_like(w), torch.empty((3 * E,), dtype=w.dtype, device=w.device)
        from ..functions import clip_ops
        if clip_ops.linear_bwd_usable(g_qk2, qk2, w[:2 * E]) and clip_ops.linear_bwd_usable(g_v2, v2, w[2 * E:]):
            # two launches (clipops_linear_bwd_f32) straight into the packed gradient's row slices instead of four
            # GEMMs and two column sums
            g_in_qk = clip_ops.linear_bwd(g_qk2, None, qk2, w[:2 * E], ctx.needs_input_grad[0], True, True,
                                          gw_out=gw[:2 * E], gb_out=gb[:2 * E])[0]
            g_in_v = clip_ops.linear_bwd(g_v2, None, v2, w[2 * E:], ctx.needs_input_grad[1], True, True,
                                         gw_out=gw[2 * E:], gb_out=gb[2 * E:])[0]
            return (None if g_in_qk is None else g_in_qk.view(ctx.shapes[0]),
                    None if g_in_v is None else g_in_v.view(ctx.shapes[1]), gw, gb)
        torch.mm(g_qk2.t(), qk2, out=gw[:2 * E])
        torch.mm(g_v2.t(), v2, out=gw[2 * E:])
        clip_ops.colsum(g_qk2.contiguous(), out=gb[:2 * E])          # (tiled kernel for query-sized inputs)
        clip_ops.colsum(g_v2.contiguous(), out=gb[2 * E:])
        g_in_qk = (g_qk2 @ w[:2 * E]).view(ctx.shapes[0]) if ctx.needs_input_grad[0] else None
        g_in_v = (g_v2 @ w[2 * E:]).view(ctx.shapes[1]) if ctx.needs_input_grad[1] else None
        return g_in_qk, g_in_v, gw, gb


class _PackedInProj3(torch.autograd.Function):
    """(q W_q^T + b_q, k W_k^T + b_k, v W_v^T + b_v) from the packed parameters for three different inputs (the query
    updater's memory attention).  One node for the same reason as ``_PackedInProj``: three row slices of the packed
    weight and bias in the graph cost a zero-fill, a strided copy and an add into the parameter's buffer EACH in the
    backward (18 small kernels per call); here the three launches write the row slices of one gradient buffer."""

    @staticmethod
    def forward(ctx, q, k, v, w, b):
        from ..functions import clip_ops
        E = w.shape[1]
        xs = tuple(t.reshape(-1, E) for t in (q, k, v))
        outs = []
        for i, x2 in enumerate(xs):
            wi, bi = w[i * E:(i + 1) * E], b[i * E:(i + 1) * E]
            y = clip_ops.linear_fwd(x2, wi, bi, False) if clip_ops.linear_fwd_usable(x2, wi, bi) else torch.addmm(bi, x2, wi.t())
            outs.append(y.view(*(q, k, v)[i].shape[:-1], E))
        ctx.save_for_backward(*xs, w)
        ctx.shapes = (q.shape, k.shape, v.shape)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        from ..functions import clip_ops
        *xs, w = ctx.saved_tensors
        E = w.shape[1]
        gw, gb = torch.empty_like(w), torch.empty((3 * E,), dtype=w.dtype, device=w.device)
        g_in = []
        for i, (g, x2) in enumerate(zip(grads, xs)):
            g2 = g.reshape(-1, E)
            wi, rows = w[i * E:(i + 1) * E], slice(i * E, (i + 1) * E)
            if clip_ops.linear_bwd_usable(g2, x2, wi):
                gx = clip_ops.linear_bwd(g2, None, x2, wi, ctx.needs_input_grad[i], True, True, gw_out=gw[rows],
                                         gb_out=gb[rows])[0]
            else:
                torch.mm(g2.t(), x2, out=gw[rows])
                clip_ops.colsum(g2.contiguous(), out=gb[rows])
                gx = g2 @ wi if ctx.needs_input_grad[i] else None
            g_in.append(None if gx is None else gx.view(ctx.shapes[i]))
        return (*g_in, gw, gb)


def self_attention(mha: nn.MultiheadAttention, qk: torch.Tensor, v: torch.Tensor,
                   key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``mha(qk, qk, v, key_padding_mask=..., need_weights=False)[0]`` for a batch-first module with packed
    input projections.  ``qk``/``v``: (B, L, E); ``key_padding_mask``: (B, L) bool, True = ignore that key.  A mask
    tagged ``_no_padding = True`` by its maker (``MeMOTR.get_query_mask`` knows the track counts on the host) is
    all-False and is dropped, which lets the fused attention run without a bias tensor."""
    if (not mha.batch_first or not mha._qkv_same_embed_dim or mha.in_proj_bias is None or mha.bias_k is not None
            or mha.add_zero_attn):
        return mha(qk, qk, v, key_padding_mask=key_padding_mask, need_weights=False)[0]
    E, H = mha.embed_dim, mha.num_heads
    d = E // H
    B, L, _ = qk.shape
    w, b = mha.in_proj_weight, mha.in_proj_bias
    if not torch.is_autocast_enabled() and qk.dtype == w.dtype and v.dtype == w.dtype:
        if torch.is_grad_enabled() and w.requires_grad:
            qk_p, v_p = _PackedInProj.apply(qk, v, w, b)
        else:       # inference (torch.no_grad): the same two GEMMs without the autograd node
            qk_p, v_p = F.linear(qk, w[:2 * E], b[:2 * E]), F.linear(v, w[2 * E:], b[2 * E:])
        from ..functions import clip_ops
        if clip_ops.self_attention_supported(qk_p, H) and not (mha.training and mha.dropout > 0):
            # hand-written kernels (head_dim 32, K / V of a head in LDS) on the packed projections, heads come out
            # concatenated: no unbind / transposes / copy, and no AOTriton kernels on the path -- training and
            # inference alike
            no_pad = key_padding_mask is None or getattr(key_padding_mask, "_no_padding", False)
            out = clip_ops.self_attention(qk_p, v_p, None if no_pad else key_padding_mask, H)
            from .linear import row_linear
            return row_linear(out, mha.out_proj.weight, mha.out_proj.bias)
        q, k = (t.transpose(1, 2) for t in qk_p.view(B, L, 2, H, d).unbind(2))     # (B, H, L, d)
        vh = v_p.view(B, L, H, d).transpose(1, 2)
    else:
        qk_p = F.linear(qk, w[:2 * E], b[:2 * E])
        v_p = F.linear(v, w[2 * E:], b[2 * E:])
        from ..functions import clip_ops
        if (torch.is_autocast_enabled() and clip_ops.self_attention_supported(qk_p, H, any_float=True)
                and not (mha.training and mha.dropout > 0)):
            # autocast (bf16 projections on MFMA): the attention itself as a float32 island in the hand-written
            # kernels -- query-sized tensors, the casts are noise, and no AOTriton kernel runs
            no_pad = key_padding_mask is None or getattr(key_padding_mask, "_no_padding", False)
            out = clip_ops.self_attention(qk_p.float(), v_p.float(), None if no_pad else key_padding_mask, H)
            from .linear import row_linear
            return row_linear(out, mha.out_proj.weight, mha.out_proj.bias)
        qk_p = qk_p.view(B, L, 2, H, d)
        q, k = qk_p[:, :, 0].transpose(1, 2), qk_p[:, :, 1].transpose(1, 2)        # (B, H, L, d)
        vh = v_p.view(B, L, H, d).transpose(1, 2)
    mask = None
    if key_padding_mask is not None and not getattr(key_padding_mask, "_no_padding", False):
        mask = ~key_padding_mask.view(B, 1, 1, L)                                   # True = take part
    out = F.scaled_dot_product_attention(q, k, vh, attn_mask=mask,
                                         dropout_p=mha.dropout if mha.training else 0.0)
    out = out.transpose(1, 2).reshape(B, L, E)
    from .linear import row_linear
    return row_linear(out, mha.out_proj.weight, mha.out_proj.bias)


def memory_attention(mha: nn.MultiheadAttention, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                     key_padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``mha(q, k, v, need_weights=False)[0]`` for batch-first (B, L, E) inputs of one length with three different
    sources (the query updater: short memory + pos, long memory + pos, output embedding; reference
    models/query_updater.py:123-125).  On CUDA fp32 with head_dim 32 the attention itself runs in the hand-written
    kernels; anything else goes through the module.  ``key_padding_mask`` (B, L) bool, True = ignore that key: the
    padded slots of the captured update (models/updater_graphs.py); the reference call has none."""
    from ..functions import clip_ops
    from .linear import row_linear
    E, H = mha.embed_dim, mha.num_heads
    island = torch.is_autocast_enabled()          # bf16 projections, float32 attention (as in self_attention)
    q32, k32 = (q.float(), k.float()) if island else (q, k)
    ok = (mha.batch_first and mha._qkv_same_embed_dim and mha.in_proj_bias is not None and mha.bias_k is None
          and not mha.add_zero_attn and not (mha.training and mha.dropout > 0)
          and q.dim() == 3 and q.shape == k.shape == v.shape and q.is_cuda and q32.dtype == torch.float32
          and clip_ops.attention_supported(q32, k32, H) and v.is_cuda)
    if not ok:
        return mha(q, k, v, key_padding_mask=key_padding_mask, need_weights=False)[0]
    w, b = mha.in_proj_weight, mha.in_proj_bias
    if (not island and torch.is_grad_enabled() and w.requires_grad and q.dtype == w.dtype == v.dtype == k.dtype
            and w.is_contiguous() and b.is_contiguous()):
        q_p, k_p, v_p = _PackedInProj3.apply(q, k, v, w, b)
    else:
        q_p = row_linear(q, w[:E], b[:E])
        k_p = row_linear(k, w[E:2 * E], b[E:2 * E])
        v_p = row_linear(v, w[2 * E:], b[2 * E:])
    if island:
        q_p, k_p, v_p = q_p.float(), k_p.float(), v_p.float()
    out = clip_ops.attention(q_p, k_p, v_p, H, key_padding_mask)
    return row_linear(out, mha.out_proj.weight, mha.out_proj.bias)
