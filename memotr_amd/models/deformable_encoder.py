"""Deformable encoder: 6 x {MSDeformAttn self-attention over the pyramid, LN, FFN, LN}
(reference models/deformable_encoder.py:21-131)."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ..modules import MSDeformAttn
from ..functions.clip_ops import add_layer_norm
from ..modules.linear import long_linear
from .utils import get_activation_layer, get_clones


class DeformableEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, use_checkpoint: bool):
        super().__init__()
        self.layers = get_clones(module=encoder_layer, n=num_layers)
        self.num_layers = num_layers
        self.use_checkpoint = use_checkpoint

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres normalised by the valid extent of their own level, re-scaled to every level:
        (B, S, L, 2) in (x, y).  ``spatial_shapes`` may be a tensor or a python list of (H, W)."""
        shapes = spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes
        refs = []
        for lvl, (h, w) in enumerate(shapes):
            h, w = int(h), int(w)
            ys = torch.linspace(0.5, h - 0.5, h, dtype=torch.float32, device=device)
            xs = torch.linspace(0.5, w - 0.5, w, dtype=torch.float32, device=device)
            ry, rx = torch.meshgrid(ys, xs, indexing="ij")
            ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * h)
            rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * w)
            refs.append(torch.stack((rx, ry), -1))
        ref = torch.cat(refs, 1)
        return ref[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None,
                shapes_list=None, reference_points=None):
        if reference_points is None:        # (the caller may hold them for this pyramid / mask geometry)
            reference_points = self.get_reference_points(shapes_list if shapes_list is not None else spatial_shapes,
                                                         valid_ratios, device=src.device)
        output = src
        if self.use_checkpoint:
            # CHECKPOINT_LEVEL 1: recompute in groups of three layers (reference :46-57)
            def run_group(x, first):
                for i in range(first, min(first + 3, self.num_layers)):
                    x = self.layers[i](x, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
                return x
            for first in range(0, self.num_layers, 3):
                output = checkpoint(run_group, output, first, use_reentrant=False)
            return output
        for layer in self.layers:
            output = layer(output, pos, reference_points, spatial_shapes, level_start_index, padding_mask)
        return output


class DeformableEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="ReLU", n_levels=4, n_heads=8, n_points=4,
                 sigmoid_attn=False):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model=d_model, n_levels=n_levels, n_heads=n_heads, n_points=n_points,
                                      sigmoid_attn=sigmoid_attn)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = get_activation_layer(activation=activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_ffn(self, src):
        hidden = self.dropout2(long_linear(src, self.linear1.weight, self.linear1.bias, activation=self.activation))
        return add_layer_norm(src, self.dropout3(long_linear(hidden, self.linear2.weight, self.linear2.bias)), self.norm2)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        attn = self.self_attn(self.with_pos_embed(src, pos), reference_points, src, spatial_shapes,
                              level_start_index, padding_mask)
        src = add_layer_norm(src, self.dropout1(attn), self.norm1)
        return self.forward_ffn(src)
