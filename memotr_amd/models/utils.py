"""Model-side helpers (semantics of the reference's models/utils.py)."""
from __future__ import annotations

import copy
import math

import torch
import torch.nn as nn

from ..utils.utils import is_distributed, is_main_process


def get_activation_layer(activation: str) -> nn.Module:
    if activation == "ReLU":
        return nn.ReLU(True)
    if activation == "GELU":
        return nn.GELU()
    raise ValueError(f"Do not support activation layer: {activation}")


def get_clones(module: nn.Module, n: int) -> nn.ModuleList:
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


def get_model(model):
    return model.module if is_distributed() and hasattr(model, "module") else model


def logits_to_scores(logits: torch.Tensor) -> torch.Tensor:
    return logits.sigmoid()


_DIM_CACHE = {}


def _sine_dims(num_pos_feats, temperature, device) -> torch.Tensor:
    key = (float(num_pos_feats), float(temperature), str(device))
    d = _DIM_CACHE.get(key)
    if d is None:
        i = torch.arange(num_pos_feats, dtype=torch.float32, device=device)
        d = temperature ** (2 * torch.div(i, 2, rounding_mode="trunc") / num_pos_feats)
        _DIM_CACHE[key] = d
    return d


def pos_to_pos_embed(pos: torch.Tensor, num_pos_feats: int = 64, temperature: int = 10000,
                     scale: float = 2 * math.pi) -> torch.Tensor:
    """Sine embedding of the last axis: (..., K) -> (..., K * num_pos_feats), sin/cos interleaved
    (models/utils.py:78-85; (n,4) boxes with 128 feats give (n,512))."""
    dim_i = _sine_dims(num_pos_feats, temperature, pos.device)
    from ..functions import clip_ops
    if clip_ops.fused(pos) and pos.dtype == torch.float32:
        return clip_ops.sine_embed(pos, dim_i, scale)      # one kernel instead of eight (and one backward)
    e = (pos * scale)[..., None] / dim_i
    e = torch.stack((e[..., 0::2].sin(), e[..., 1::2].cos()), dim=-1)
    return torch.flatten(e, start_dim=-3)


def save_checkpoint(model: nn.Module, path: str, states: dict = None, optimizer=None, scheduler=None):
    """Rank-0 writes {"model","optimizer","scheduler","states"} (models/utils.py:15-28)."""
    if not is_main_process():
        return
    torch.save({
        "model": get_model(model).state_dict(),
        "optimizer": None if optimizer is None else optimizer.state_dict(),
        "scheduler": None if scheduler is None else scheduler.state_dict(),
        "states": states,
    }, path)


def load_checkpoint(model: nn.Module, path: str, states: dict = None, optimizer=None, scheduler=None):
    """Model weights load on rank 0 only; DDP's construction-time broadcast syncs the rest (models/utils.py:31-45)."""
    state = torch.load(path, map_location="cpu")
    if is_main_process():
        model.load_state_dict(state["model"])
    if optimizer is not None:
        optimizer.load_state_dict(state["optimizer"])
    if scheduler is not None:
        scheduler.load_state_dict(state["scheduler"])
    if states is not None:
        states.update(state["states"])


def remap_pretrained_state_dict(pretrained: dict, model_state: dict) -> dict:
    """DAB-Deformable-DETR COCO checkpoint -> MeMOTR key names (models/utils.py:88-168).

    backbone.0.body.* -> backbone.backbone.backbone.*, input_proj.* -> feature_projs.*,
    tgt_embed/query_embed -> det_query_embed, refpoint_embed -> det_anchor, class_embed rows sliced to the
    person class(es); everything missing keeps the model's own initialisation.
    """
    out = dict(pretrained)
    for k in list(pretrained.keys()):
        v = pretrained[k]
        if k in model_state:
            if model_state[k].shape != v.shape and "class_embed" in k:
                n = model_state[k].shape[0]
                if n in (1, 2, 3):
                    out[k] = v[1:1 + n]
                elif n == 8:        # BDD100K: keep the fresh head
                    out[k] = model_state[k]
                else:
                    raise NotImplementedError(f"invalid shape: {model_state[k].shape}")
        elif "query_embed" in k or "tgt_embed" in k:
            tgt = model_state["det_query_embed"]
            out["det_query_embed"] = v.clone() if v.shape == tgt.shape else tgt
            del out[k]
        elif "refpoint_embed" in k:
            tgt = model_state["det_anchor"]
            out["det_anchor"] = v.clone() if v.shape == tgt.shape else tgt
            del out[k]
        elif "backbone" in k:
            out["backbone.backbone.backbone" + k[15:]] = v.clone()
            del out[k]
        elif "input_proj" in k:
            out["feature_projs" + k[10:]] = v.clone()
            del out[k]
    for k, v in model_state.items():
        out.setdefault(k, v)
    return out


def load_pretrained_model(model: nn.Module, pretrained_path: str, show_details: bool = False):
    if not is_main_process():
        return model
    ckpt = torch.load(pretrained_path, map_location="cpu")
    mapped = remap_pretrained_state_dict(ckpt["model"], model.state_dict())
    missing = model.load_state_dict(mapped, strict=False)
    if show_details:
        print(missing)
    return model
