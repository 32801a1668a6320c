"""ResNet-50 (v1.5: stride on the 3x3 conv) written from its definition.

The reference takes ``torchvision.models.resnet50`` (models/backbone.py:8,70); torchvision is not a
dependency here.  Module and parameter names match torchvision's so published checkpoints load
(``conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}``).
"""
from __future__ import annotations

from typing import Callable, Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int, downsample, norm_layer: Callable):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        extra = None
        if self.downsample is None:
            identity = x
        else:
            fold = getattr(self.downsample[1], "fold_into_conv", None)
            if fold is not None and x.is_cuda:     # the projection's own shift rides in the block's last pass
                ds = self.downsample[0]
                w, extra = fold(ds.weight)
                identity = F.conv2d(x, w, None, ds.stride, ds.padding, ds.dilation, ds.groups)
            else:
                identity = _conv_norm(self.downsample[0], self.downsample[1], x)
        out = _conv_norm_relu(self.conv1, self.bn1, x)
        out = _conv_norm_relu(self.conv2, self.bn2, out)
        return _conv_norm_relu(self.conv3, self.bn3, out, identity, extra)   # relu(bn3(conv3) + identity)


def _conv_norm(conv: nn.Conv2d, norm: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """conv followed by its norm; a frozen affine norm is folded into the conv (one pass over the
    activation instead of two -- the elementwise scale/shift is the memory-bound part)."""
    fold = getattr(norm, "fold_into_conv", None)
    if fold is not None:
        w, b = fold(conv.weight)
        return F.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups)
    return norm(conv(x))


def _conv_norm_relu(conv: nn.Conv2d, norm: nn.Module, x: torch.Tensor, res: torch.Tensor = None,
                    extra_shift: torch.Tensor = None) -> torch.Tensor:
    """relu(norm(conv(x)) (+ res)).  With a frozen norm folded into the convolution the shift, the residual and the
    ReLU are ONE pass over the convolution's output (functions/clip_ops.shift_relu_) instead of three -- these
    activations are the largest tensors of the step (344 MB for layer1's outputs of a 5-frame clip)."""
    fold = getattr(norm, "fold_into_conv", None)
    if fold is not None:
        w, b = fold(conv.weight)
        if extra_shift is not None:            # (`res` came without its own shift: Bottleneck.forward)
            b = b + extra_shift
        if x.is_cuda and not b.requires_grad:
            from ..functions.clip_ops import shift_relu_
            return shift_relu_(F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups), b, res)
        out = F.conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, conv.groups)
    else:
        out = norm(conv(x))
        if extra_shift is not None:
            out = out + extra_shift.view(1, -1, 1, 1)
    if res is not None:
        out = out + res
    return F.relu(out, inplace=True)


class ResNet50Body(nn.Module):
    """conv1 .. layer4; ``forward`` returns {'0': layer2, '1': layer3, '2': layer4} like the reference's
    IntermediateLayerGetter(return_layers={"layer2": "0", "layer3": "1", "layer4": "2"})."""

    def __init__(self, norm_layer: Callable, return_layers: Dict[str, str]):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 3, 1, norm_layer)
        self.layer2 = self._make_layer(128, 4, 2, norm_layer)
        self.layer3 = self._make_layer(256, 6, 2, norm_layer)
        self.layer4 = self._make_layer(512, 3, 2, norm_layer)
        self.return_layers = dict(return_layers)
        for m in self.modules():  # torchvision's default init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes: int, blocks: int, stride: int, norm_layer: Callable) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
                norm_layer(planes * Bottleneck.expansion))
        layers: List[nn.Module] = [Bottleneck(self.inplanes, planes, stride, downsample, norm_layer)]
        self.inplanes = planes * Bottleneck.expansion
        layers += [Bottleneck(self.inplanes, planes, 1, None, norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        x = _conv_norm_relu(self.conv1, self.bn1, x)
        x = self.maxpool(x)
        out = {}
        for name in ("layer1", "layer2", "layer3", "layer4"):
            x = getattr(self, name)(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out
