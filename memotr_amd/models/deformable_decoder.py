"""Deformable decoder with DAB anchors, iterative box refinement and the detect/track split of the
first layers (reference models/deformable_decoder.py:22-319)."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from ..modules import MSDeformAttn
from ..modules.ms_deform_attn import project_values
from ..functions.clip_ops import add_layer_norm
from ..modules.attention import self_attention
from ..modules.linear import row_linear
from ..utils.utils import inverse_sigmoid, refine_boxes
from .decoder_graphs import DecoderGraphs
from .mlp import MLP
from .utils import get_activation_layer, get_clones, pos_to_pos_embed


class DeformableDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, return_intermediate=False, merge_det_track_layer: int = 0,
                 n_det_queries: int = 300, d_model: int = 256, use_checkpoint: bool = False, use_dab: bool = False,
                 visualize: bool = False):
        super().__init__()
        self.layers = get_clones(module=decoder_layer, n=num_layers)
        self.num_layers = num_layers
        self.return_intermediate = return_intermediate
        self.merge_det_track_layer = merge_det_track_layer
        self.n_det_queries = n_det_queries
        self.d_model = d_model
        self.bbox_embed = None      # set by MeMOTR (shared with its box heads)
        self.class_embed = None
        self.use_checkpoint = use_checkpoint
        self.use_dab = use_dab
        self.visualize = visualize
        if self.use_dab:
            self.query_scale = MLP(d_model, d_model, d_model, 2)
            self.ref_point_head = MLP(d_model * 2, d_model, d_model, 2)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_decoder_graphs", None)        # captured hipGraphs are per-process objects: never pickled / deep-copied
        state.pop("_infer_graphs", None)
        return state

    def graphs(self) -> DecoderGraphs:
        g = self.__dict__.get("_decoder_graphs")
        if g is None:
            g = self.__dict__["_decoder_graphs"] = DecoderGraphs(self)
        return g

    def infer_graphs(self):
        g = self.__dict__.get("_infer_graphs")
        if g is None:
            from .infer_graphs import InferGraphs
            g = self.__dict__["_infer_graphs"] = InferGraphs(None)
        return g

    def _forward_graphed(self, tgt, reference_points, src, spatial_shapes, level_start_index, valid_ratios, query_mask,
                         src_padding_mask, frame_slot, clip_key=None, infer=False):
        """The loop below with every iteration replayed from a hipGraph (models/decoder_graphs.py; ``infer``: the
        forward-only capture of models/infer_graphs.py).  Returns None when a capture fails; the caller then runs the
        eager loop."""
        graphs = self.graphs()
        nd = self.n_det_queries
        B, nq, C = tgt.shape
        nb = graphs.bucket(nq, nd)
        if nb > nq:      # static shapes: pad the track part with masked slots (excluded as keys, sliced away below)
            pad = nb - nq
            tgt = torch.cat((tgt, tgt.new_zeros(B, pad, C)), 1)
            reference_points = torch.cat((reference_points, reference_points.new_full((B, pad, 4), 0.5)), 1)
            query_mask = torch.cat((query_mask, query_mask.new_ones(B, pad)), 1)
        ratios4 = torch.cat([valid_ratios, valid_ratios], -1)[:, None].contiguous()
        query_mask = query_mask.contiguous()
        args = (tgt.contiguous(), reference_points.contiguous(), src, ratios4, query_mask, src_padding_mask)
        if infer:
            res = self.infer_graphs().run_decode(self, args, spatial_shapes, level_start_index)
        else:
            res = graphs.run(frame_slot, args, spatial_shapes, level_start_index, clip_key)
        if res is None:
            return None
        outs, refs, layer_inputs, boxes = res
        if nb > nq:
            outs, refs, layer_inputs, boxes = (x[:, :, :nq] for x in (outs, refs, layer_inputs, boxes))
        return outs, refs, layer_inputs, boxes

    def forward(self, tgt, reference_points, src, src_spatial_shapes, src_level_start_index, src_valid_ratios,
                query_pos, query_mask, src_padding_mask, frame_slot=None, clip_key=None):
        """tgt (B,Nq,C); reference_points (B,Nq,4) in [0,1]; src (B,S,C).
        Returns stacks over layers: outputs (n,B,Nq,C), refined references (n,B,Nq,4), layer inputs (n,B,Nq,C),
        refined boxes with their graph (n,B,Nq,4) or None without box refinement.
        ``frame_slot`` (the frame's index inside its clip, given by the training loop) selects the hipGraph slot the
        loop is replayed from; without it -- inference, the reference's frame order -- the loop runs eagerly.
        ``clip_key`` (any object, one per clip) lets the frames of a clip share the flat copy of the decoder
        parameters their graphs read."""
        if not self.return_intermediate:
            raise NotImplementedError("Not Support for no Inter Outputs.")
        if (frame_slot is not None and reference_points.shape[-1] == 4 and src_padding_mask is not None
                and self.graphs().usable(tgt, src)):
            res = self._forward_graphed(tgt, reference_points, src, src_spatial_shapes, src_level_start_index,
                                        src_valid_ratios, query_mask, src_padding_mask, frame_slot, clip_key)
            if res is not None:
                return res
        if (not torch.is_grad_enabled() and reference_points.shape[-1] == 4 and src_padding_mask is not None
                and self.infer_graphs().decode_usable(self, tgt, src)):
            res = self._forward_graphed(tgt, reference_points, src, src_spatial_shapes, src_level_start_index,
                                        src_valid_ratios, query_mask, src_padding_mask, None, infer=True)
            if res is not None:
                return res
        nd = self.n_det_queries
        output = tgt
        outs, refs, layer_inputs, boxes = [], [], [], []
        ref_backup = None
        ratios4 = torch.cat([src_valid_ratios, src_valid_ratios], -1)[:, None]      # same for every layer
        # the layers' value projections of `src` as one GEMM (modules/ms_deform_attn.py: project_values); None: each
        # layer projects for itself (CPU, autocast, other head sizes; checkpointed layers recompute on their own)
        values = None if self.use_checkpoint else project_values([layer.cross_attn for layer in self.layers], src,
                                                                 src_padding_mask)
        for lid, layer in enumerate(self.layers):
            if lid == 0 and not self.use_dab:        # Deformable-DETR variant: 2-d references
                ref_backup = reference_points.clone()
                reference_points = reference_points[:, :, :2]
            if reference_points.shape[-1] == 4:
                ref_in = reference_points[:, :, None] * ratios4
            else:
                assert reference_points.shape[-1] == 2
                ref_in = reference_points[:, :, None] * src_valid_ratios[:, None]
            if self.use_dab:
                anchor = pos_to_pos_embed(ref_in[:, :, 0, :], num_pos_feats=self.d_model // 2)
                raw_pos = self.ref_point_head(anchor)
                query_pos = raw_pos if lid == 0 else self.query_scale(output) * raw_pos
            layer_inputs.append(output)
            merge = lid >= self.merge_det_track_layer
            if self.use_checkpoint:
                output = checkpoint(layer, output, query_pos, ref_in, src, src_spatial_shapes, src_level_start_index,
                                    query_mask, src_padding_mask, merge, use_reentrant=False)
            else:
                output = layer(output, query_pos, ref_in, src, src_spatial_shapes, src_level_start_index, query_mask,
                               src_padding_mask, merge, None if values is None else values[lid])
            if self.bbox_embed is not None:
                delta = self.bbox_embed[lid](output)
                if reference_points.shape[-1] == 4:
                    new_ref = refine_boxes(delta, reference_points)
                else:
                    xy = delta[..., :2] + inverse_sigmoid(reference_points)
                    new_ref = torch.cat((xy, delta[..., 2:]), -1).sigmoid()
                boxes.append(new_ref)
                if not merge:   # track queries did not go through the layer: keep their anchors
                    keep = reference_points if self.use_dab else ref_backup
                    reference_points = torch.cat((new_ref[:, :nd].detach(), keep[:, nd:]), dim=1)
                else:
                    reference_points = new_ref.detach()
            outs.append(output)
            refs.append(reference_points)
        # ``boxes``: the refined boxes of every layer BEFORE the detach -- exactly what the box head computes from
        # the same layer output, reference and (aliased) bbox_embed weights (reference models/memotr.py:131-143
        # recomputes it: 3 GEMMs + inverse_sigmoid + sigmoid per layer, forward and backward).  The decoder's own
        # use of them is detached, so handing this one copy to the head leaves values and gradients unchanged.
        return (torch.stack(outs), torch.stack(refs), torch.stack(layer_inputs),
                torch.stack(boxes) if len(boxes) == len(outs) else None)


class DeformableDecoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="ReLU", n_levels=4, n_heads=8, n_points=4,
                 sigmoid_attn=False, extra_track_attn=False, n_det_queries=300, visualize: bool = False):
        super().__init__()
        self.visualize = visualize
        self.n_det_queries = n_det_queries
        self.n_heads = n_heads
        self.self_attn = nn.MultiheadAttention(embed_dim=d_model, num_heads=n_heads, dropout=dropout,
                                               batch_first=True)
        self.dropout2 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.cross_attn = MSDeformAttn(d_model=d_model, n_levels=n_levels, n_heads=n_heads, n_points=n_points,
                                       sigmoid_attn=sigmoid_attn, visualize=visualize)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = get_activation_layer(activation=activation)
        self.dropout3 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.extra_track_attn = extra_track_attn
        if extra_track_attn:
            self.track_attn = nn.MultiheadAttention(embed_dim=d_model, num_heads=n_heads, dropout=dropout,
                                                    batch_first=True)
            self.dropout5 = nn.Dropout(dropout)
            self.norm4 = nn.LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def forward_self_attn(self, tgt, query_pos, query_mask):
        qk = self.with_pos_embed(tgt, query_pos)
        attn = self_attention(self.self_attn, qk, tgt, key_padding_mask=query_mask)
        return add_layer_norm(tgt, self.dropout2(attn), self.norm2)

    def forward_track_attn(self, tgt, query_pos, query_mask):
        nd = self.n_det_queries
        if tgt.shape[1] <= nd:
            return tgt
        qk = self.with_pos_embed(tgt, query_pos)[:, nd:]
        track_mask = query_mask[:, nd:]
        track_mask._no_padding = getattr(query_mask, "_no_padding", False)
        attn = self_attention(self.track_attn, qk, tgt[:, nd:], key_padding_mask=track_mask)
        return torch.cat([tgt[:, :nd], self.norm4(tgt[:, nd:] + self.dropout5(attn))], dim=1)

    def forward_ffn(self, tgt):
        # ReLU in the GEMM epilogue, never in place: nn.Linear on a (B, L, E) input returns a VIEW of its product, and an
        # in-place op on a view makes autograd rebase the graph (CopySlices + AsStrided backward nodes, six extra kernels
        # per layer and frame)
        if isinstance(self.activation, nn.ReLU):
            hidden = row_linear(tgt, self.linear1.weight, self.linear1.bias, relu=True)
        else:
            hidden = self.activation(row_linear(tgt, self.linear1.weight, self.linear1.bias))
        hidden = self.dropout3(hidden)
        return add_layer_norm(tgt, self.dropout4(row_linear(hidden, self.linear2.weight, self.linear2.bias)), self.norm3)

    def forward(self, tgt, query_pos, reference_points, src, src_spatial_shapes, level_start_index, query_mask,
                src_padding_mask=None, merge_det_track=False, value=None):
        nd = self.n_det_queries
        track_tgt = None
        if not merge_det_track:   # early layers see the detect queries only
            # (ONE split, not two slices: a slice's backward is a zero-fill of the whole tensor plus a copy -- inside a
            #  capture a memcpy node, ~27 us of host time per replay, tools/graph_launch_probe.py -- and the two meet
            #  in an add; the split's backward is one concatenation)
            tgt, track_tgt = tgt.split((nd, tgt.shape[1] - nd), dim=1)
            query_pos = query_pos[:, :nd, :]
            reference_points, query_mask = reference_points[:, :nd], query_mask[:, :nd]
            query_mask._no_padding = True             # padding only ever sits behind the track queries
        if self.extra_track_attn:
            tgt = self.forward_track_attn(tgt, query_pos, query_mask)
        tgt = self.forward_self_attn(tgt, query_pos, query_mask)
        cross = self.cross_attn(self.with_pos_embed(tgt, query_pos), reference_points, src, src_spatial_shapes,
                                level_start_index, src_padding_mask, value)
        tgt = add_layer_norm(tgt, self.dropout1(cross), self.norm1)
        tgt = self.forward_ffn(tgt)
        if track_tgt is not None:
            tgt = torch.cat((tgt, track_tgt), dim=1)
        return tgt
