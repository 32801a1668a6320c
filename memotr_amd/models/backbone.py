"""ResNet-50 backbone with frozen BatchNorm + sine position embedding (reference models/backbone.py)."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.nested_tensor import NestedTensor
from .position_embedding import build as build_position_embedding
from .resnet import ResNet50Body


class FrozenBatchNorm2d(nn.Module):
    """BatchNorm2d with fixed statistics and affine: y = x * scale + bias, scale = w * rsqrt(var + eps)
    (models/backbone.py:16-52; all four tensors are buffers)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.eps = eps

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        state_dict.pop(prefix + "num_batches_tracked", None)
        self._folded = None
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def _apply(self, fn, *args, **kwargs):
        self._folded = None                      # .to() / .cuda() replace the buffers
        return super()._apply(fn, *args, **kwargs)

    def train(self, mode: bool = True):
        self._folded = None                      # a mode switch is the cheap moment to forget `.data` writes too
        return super().train(mode)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_folded"] = None                  # derived tensors do not travel with pickles / deepcopies
        return state

    def scale_shift(self):
        """(scale, shift), cached: the four buffers are constants, so the five tiny kernels that derive them run
        once instead of once per convolution per frame; any in-place write to a buffer (checkpoint load, DDP
        buffer broadcast) bumps its version counter and refreshes the cache."""
        bufs = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((id(b), b._version) for b in bufs) + (torch.is_inference_mode_enabled(),)
        cached = getattr(self, "_folded", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                scale = self.weight * (self.running_var + self.eps).rsqrt()
                shift = self.bias - self.running_mean * scale
            cached = (key, scale, shift)
            self._folded = cached
        return cached[1], cached[2]

    def fold_into_conv(self, conv_weight: torch.Tensor):
        """(W * scale[:,None,None,None], shift): conv(x, W') + shift == norm(conv(x, W))."""
        scale, shift = self.scale_shift()
        return conv_weight * scale.view(-1, 1, 1, 1), shift

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


class Backbone(nn.Module):
    """Returns {'0','1','2'} -> NestedTensor for layer2/3/4 (strides 8/16/32, 512/1024/2048 channels);
    conv1/bn1/layer1 are frozen, layer2-4 train (models/backbone.py:72-74)."""

    def __init__(self, backbone_name: str, train_backbone: bool, return_interm_layers: bool):
        super().__init__()
        assert backbone_name == "resnet50", f"Backbone do not support '{backbone_name}'."
        if return_interm_layers:
            return_layers = {"layer2": "0", "layer3": "1", "layer4": "2"}
            self.strides = [8, 16, 32]
            self.num_channels = [512, 1024, 2048]
        else:
            return_layers = {"layer4": "0"}
            self.strides = [32]
            self.num_channels = [2048]
        # random init: pretrained ImageNet weights are loaded from a checkpoint, never downloaded
        self.backbone = ResNet50Body(norm_layer=FrozenBatchNorm2d, return_layers=return_layers)
        for name, p in self.backbone.named_parameters():
            if not train_backbone or not any(k in name for k in ("layer2", "layer3", "layer4")):
                p.requires_grad_(False)

    def forward(self, ntensor: NestedTensor) -> Dict[str, NestedTensor]:
        masks = ntensor.masks
        assert masks is not None, "Masks should be NOT NONE."
        res = {}
        for name, feat in self.backbone(ntensor.tensors).items():
            m = F.interpolate(masks[None].float(), mode="nearest", size=feat.shape[-2:]).to(masks.dtype)[0]
            res[name] = NestedTensor(feat, m, getattr(ntensor, "sizes", None))
        return res


class BackboneWithPE(nn.Module):
    def __init__(self, backbone: nn.Module, position_embedding: nn.Module):
        super().__init__()
        self.backbone = backbone
        self.position_embedding = position_embedding
        self.strides = backbone.strides
        self.num_channels = backbone.num_channels

    def forward(self, ntensor: NestedTensor):
        outputs = self.backbone(ntensor)
        features: List[NestedTensor] = [outputs[k] for k in sorted(outputs)]
        pos_embeds = [self.position_embedding(f) for f in features]
        return features, pos_embeds

    def n_inter_layers(self):
        return len(self.strides)

    def n_inter_channels(self):
        return self.num_channels


def build(config: dict) -> BackboneWithPE:
    return BackboneWithPE(backbone=Backbone(config["BACKBONE"], train_backbone=True, return_interm_layers=True),
                          position_embedding=build_position_embedding(config))
