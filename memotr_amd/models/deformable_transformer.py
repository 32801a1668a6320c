"""Pyramid flattening + encoder + decoder (reference models/deformable_transformer.py:23-299; the
two-stage branch there is dead code -- it raises at :234 -- and is not reproduced)."""
from __future__ import annotations

import os
from typing import List

import torch
import torch.nn as nn
from torch.nn.init import constant_, normal_, xavier_uniform_
from torch.utils.checkpoint import checkpoint

from ..functions.clip_ops import add_row_bias

from .. import MultiScaleDeformableAttention as MSDA
from ..modules import MSDeformAttn
from ..modules.ms_deform_attn import tag_masked_rows
from .deformable_decoder import DeformableDecoder, DeformableDecoderLayer
from .deformable_encoder import DeformableEncoder, DeformableEncoderLayer


class DeformableTransformer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, n_feature_levels=4, n_heads=8, n_enc_points=4, n_dec_points=4,
                 n_enc_layers=6, n_dec_layers=6, merge_det_track_layer=0, dropout=0.1, activation="ReLU",
                 return_intermediate_dec=False, n_det_queries=300, extra_track_attn=False, two_stage=False,
                 two_stage_num_proposals=300, use_checkpoint: bool = False, checkpoint_level: int = 2,
                 use_dab: bool = False, visualize: bool = False):
        super().__init__()
        if two_stage:
            raise RuntimeError("Do not support two stage model for Deformable Transformer.")
        self.d_model = d_model
        self.n_heads = n_heads
        self.two_stage = False
        self.two_stage_num_proposals = two_stage_num_proposals
        self.use_checkpoint = use_checkpoint
        self.checkpoint_level = checkpoint_level
        self.use_dab = use_dab
        self.visualize = visualize

        encoder_layer = DeformableEncoderLayer(d_model=d_model, d_ffn=d_ffn, dropout=dropout, activation=activation,
                                               n_levels=n_feature_levels, n_heads=n_heads, n_points=n_enc_points,
                                               sigmoid_attn=False)
        decoder_layer = DeformableDecoderLayer(d_model=d_model, d_ffn=d_ffn, dropout=dropout, activation=activation,
                                               n_levels=n_feature_levels, n_heads=n_heads, n_points=n_dec_points,
                                               sigmoid_attn=False, extra_track_attn=extra_track_attn,
                                               n_det_queries=n_det_queries, visualize=visualize)
        self.encoder = DeformableEncoder(encoder_layer, n_enc_layers,
                                         use_checkpoint=(use_checkpoint and checkpoint_level == 1))
        self.decoder = DeformableDecoder(decoder_layer, n_dec_layers, return_intermediate=return_intermediate_dec,
                                         merge_det_track_layer=merge_det_track_layer, n_det_queries=n_det_queries,
                                         d_model=d_model, use_checkpoint=use_checkpoint, use_dab=use_dab,
                                         visualize=visualize)
        self.level_embed = nn.Parameter(torch.Tensor(n_feature_levels, d_model))
        if not use_dab:
            self.reference_points = nn.Linear(d_model, 2)
        self.reset_parameters()

    def reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m.reset_parameters()
        if not self.use_dab:
            xavier_uniform_(self.reference_points.weight.data, gain=1.0)
            constant_(self.reference_points.bias.data, 0.0)
        normal_(self.level_embed)

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_pyramids", "_mask_derived"):   # per-geometry device tensors, rebuilt on demand
            state.pop(k, None)
        return state

    @staticmethod
    def get_valid_ratio(mask):
        """(B, 2) = (valid_w / W, valid_h / H) from the first row / column of the padding mask."""
        _, H, W = mask.shape
        valid_h = torch.sum(~mask[:, :, 0], 1).float() / H
        valid_w = torch.sum(~mask[:, 0, :], 1).float() / W
        return torch.stack([valid_w, valid_h], -1)

    def _pyramid_tensors(self, shapes_list, device):
        """(L, 2) int64 ``spatial_shapes`` and (L,) ``level_start_index`` on the device, uploaded once per pyramid
        geometry (a pageable host->device copy per frame would stall the host behind everything queued)."""
        cache = self.__dict__.setdefault("_pyramids", {})
        key = (tuple(shapes_list), str(device))
        if key not in cache:
            if len(cache) >= 16:
                cache.clear()
            spatial_shapes = torch.as_tensor(shapes_list, dtype=torch.long, device=device)
            starts = [0]
            for h, w in shapes_list[:-1]:
                starts.append(starts[-1] + h * w)
            level_start_index = torch.as_tensor(starts, dtype=torch.long, device=device)
            # the pyramid is known as python ints: hand it to the operator so it never reads it back
            MSDA.tag_host_shapes(spatial_shapes, shapes_list)
            cache[key] = (spatial_shapes, level_start_index)
        return cache[key]

    def encode(self, srcs: List[torch.Tensor], masks: List[torch.Tensor], pos_embeds: List[torch.Tensor],
               geometry=None) -> dict:
        """Query-independent half: flatten the pyramid and run the encoder.  Returns what ``decode`` needs.
        ``geometry`` (optional, hashable: ``NestedTensor.sizes`` of the frames) says which image sizes the masks were
        drawn from; what depends on the masks alone -- their flattened copy, the valid ratios, the encoder's
        reference points, ~70 small kernels -- is then computed once per geometry."""
        shapes_list = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
        src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)                 # (B, S, C)
        lvl_pos_embed_flatten = torch.cat(
            [add_row_bias(p.flatten(2).transpose(1, 2), self.level_embed[lvl]) for lvl, p in enumerate(pos_embeds)],
            1)
        spatial_shapes, level_start_index = self._pyramid_tensors(shapes_list, src_flatten.device)
        key = None if geometry is None else (geometry, tuple(shapes_list), str(src_flatten.device))
        cache = self.__dict__.setdefault("_mask_derived", {})
        enc_ref = None
        if key is not None and key in cache:
            mask_flatten, valid_ratios, enc_ref = cache[key]
        else:
            mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)                            # (B, S)
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)               # (B, L, 2)
            if key is not None:
                with torch.no_grad():
                    enc_ref = self.encoder.get_reference_points(shapes_list, valid_ratios, src_flatten.device)
                if len(cache) >= 16:
                    cache.clear()
                if os.environ.get("MEMOTR_MASKED_ROWS", "1") != "0":
                    tag_masked_rows(mask_flatten)      # once per geometry: the modules zero these rows of `value`
                cache[key] = (mask_flatten, valid_ratios, enc_ref)

        if self.use_checkpoint and self.checkpoint_level in (2, 3):
            memory = checkpoint(self.encoder, src_flatten, spatial_shapes, level_start_index, valid_ratios,
                                lvl_pos_embed_flatten, mask_flatten, shapes_list, enc_ref, use_reentrant=False)
        else:
            memory = self.encoder(src=src_flatten, spatial_shapes=spatial_shapes,
                                  level_start_index=level_start_index, valid_ratios=valid_ratios,
                                  pos=lvl_pos_embed_flatten, padding_mask=mask_flatten, shapes_list=shapes_list,
                                  reference_points=enc_ref)
        return {"memory": memory, "spatial_shapes": spatial_shapes, "level_start_index": level_start_index,
                "valid_ratios": valid_ratios, "mask_flatten": mask_flatten}

    def decode(self, enc: dict, query_embed, ref_pts, query_mask, return_boxes: bool = False):
        """Query-dependent half: the decoder over an ``encode`` result."""
        assert query_embed is not None
        memory = enc["memory"]
        c = memory.shape[2]
        if self.use_dab:
            tgt, query_pos = query_embed, None
        else:
            query_pos, tgt = torch.split(query_embed, c, dim=2)
        assert ref_pts is not None, "ref_pts should not be None."
        init_reference_points = ref_pts.sigmoid()
        output, res_reference_points, inter_queries, boxes = self.decoder(
            tgt=tgt, reference_points=init_reference_points, src=memory, src_spatial_shapes=enc["spatial_shapes"],
            src_level_start_index=enc["level_start_index"], src_valid_ratios=enc["valid_ratios"],
            query_pos=query_pos, query_mask=query_mask, src_padding_mask=enc["mask_flatten"],
            frame_slot=enc.get("frame_slot"), clip_key=enc.get("clip_key"))
        if return_boxes:
            return output, init_reference_points, res_reference_points, inter_queries, boxes
        return output, init_reference_points, res_reference_points, inter_queries

    def forward(self, srcs: List[torch.Tensor], masks: List[torch.Tensor], pos_embeds: List[torch.Tensor],
                query_embed, ref_pts, query_mask):
        assert query_embed is not None
        return self.decode(self.encode(srcs, masks, pos_embeds), query_embed, ref_pts, query_mask)

    def get_d_model(self):
        return self.d_model

    def get_n_dec_layers(self):
        return self.decoder.num_layers

    def set_refine_bbox_embed(self, bbox_embed: nn.Module):
        self.decoder.bbox_embed = bbox_embed


def build(config: dict) -> DeformableTransformer:
    return DeformableTransformer(
        d_model=config["HIDDEN_DIM"], d_ffn=config["FFN_DIM"], n_feature_levels=config["NUM_FEATURE_LEVELS"],
        n_heads=config["NUM_HEADS"], n_enc_points=config["NUM_ENC_POINTS"], n_dec_points=config["NUM_DEC_POINTS"],
        n_enc_layers=config["NUM_ENC_LAYERS"], n_dec_layers=config["NUM_DEC_LAYERS"],
        merge_det_track_layer=config.get("MERGE_DET_TRACK_LAYER", 0), dropout=config["DROPOUT"],
        activation=config["ACTIVATION"], return_intermediate_dec=config["RETURN_INTER_DEC"],
        n_det_queries=config["NUM_DET_QUERIES"], extra_track_attn=config["EXTRA_TRACK_ATTN"], two_stage=False,
        use_checkpoint=config["USE_CHECKPOINT"], checkpoint_level=config["CHECKPOINT_LEVEL"],
        use_dab=config["USE_DAB"], visualize=config["VISUALIZE"])
