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
    qk_p = F.linear(qk, w[:2 * E], b[:2 * E]).view(B, L, 2, H, d)
    q, k = qk_p[:, :, 0].transpose(1, 2), qk_p[:, :, 1].transpose(1, 2)            # (B, H, L, d)
    vh = F.linear(v, w[2 * E:], b[2 * E:]).view(B, L, H, d).transpose(1, 2)
    mask = None
    if key_padding_mask is not None and not getattr(key_padding_mask, "_no_padding", False):
        mask = ~key_padding_mask.view(B, 1, 1, L)                                   # True = take part
    out = F.scaled_dot_product_attention(q, k, vh, attn_mask=mask,
                                         dropout_p=mha.dropout if mha.training else 0.0)
    out = out.transpose(1, 2).reshape(B, L, E)
    return F.linear(out, mha.out_proj.weight, mha.out_proj.bias)
