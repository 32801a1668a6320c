"""Sine position embedding over the un-padded area (reference models/position_embedding.py:10-47)."""
from __future__ import annotations

import math

import torch
from torch import nn

from ..utils.nested_tensor import NestedTensor


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if normalize and scale is None:
            raise ValueError("Scale should be NOT NONE when normalize is True.")
        if scale is not None and not normalize:
            raise ValueError("Normalize should be True when scale is not None.")
        self.num_pos_feats = num_pos_feats
        self.temperature = temperature
        self.normalize = normalize
        self.scale = scale

    CACHE_ENTRIES = 16

    def forward(self, ntensor: NestedTensor) -> torch.Tensor:
        """The embedding depends on the padding mask only.  When the NestedTensor says which image sizes its mask
        was drawn from (``sizes``), the result is cached by them: a clip's frames share one size, so the ~30
        element-wise kernels per level run once per geometry instead of once per encode call."""
        tensors, masks = ntensor.decompose()
        assert masks is not None, "Masks in ntensor should be NOT NONE."
        key = None
        if getattr(ntensor, "sizes", None) is not None:
            key = (ntensor.sizes, tuple(masks.shape), str(masks.device))
            cache = self.__dict__.setdefault("_cache", {})
            hit = cache.get(key)
            if hit is not None:
                return hit
        pos = self._embed(tensors, masks)
        if key is not None:
            while len(cache) >= self.CACHE_ENTRIES:
                cache.pop(next(iter(cache)))
            cache[key] = pos
        return pos

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_cache", None)
        return state

    def _embed(self, tensors: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        valid = ~masks
        y = valid.cumsum(dim=1, dtype=torch.float32)
        x = valid.cumsum(dim=2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y = (y - 0.5) / (y[:, -1:, :] + eps) * self.scale
            x = (x - 0.5) / (x[:, :, -1:] + eps) * self.scale
        i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=tensors.device)
        dim_i = self.temperature ** (2 * torch.div(i, 2, rounding_mode="trunc") / self.num_pos_feats)
        px = x[:, :, :, None] / dim_i
        py = y[:, :, :, None] / dim_i
        px = torch.stack((px[..., 0::2].sin(), px[..., 1::2].cos()), dim=4).flatten(3)
        py = torch.stack((py[..., 0::2].sin(), py[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)      # (B, 2*num_pos_feats, H, W)


def build(config: dict) -> PositionEmbeddingSine:
    assert config["HIDDEN_DIM"] % 2 == 0, f"Hidden dim should be 2x, but get {config['HIDDEN_DIM']}."
    # the reference passes the float HIDDEN_DIM / 2 (position_embedding.py:46); arange(128.0) == arange(128)
    return PositionEmbeddingSine(num_pos_feats=config["HIDDEN_DIM"] / 2, normalize=True, scale=2 * math.pi,
                                 temperature=20)
