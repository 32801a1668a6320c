"""Padded image batch + padding mask (API of the reference's utils/nested_tensor.py:9-60)."""
from __future__ import annotations

from typing import List, Optional

import torch


class NestedTensor:
    """``tensors`` (B, C, H, W) and ``masks`` (B, H, W) bool, True on padding.

    ``sizes`` (optional, host metadata): ((H_pad, W_pad), (h_0, w_0), ..., (h_B-1, w_B-1)) -- the padded size and
    the image sizes the masks were drawn from.  Everything derived from the masks alone (their per-level
    down-samplings, the sine position embeddings) is a function of this tuple, so it can be cached by it without
    reading the masks back."""

    def __init__(self, tensors: torch.Tensor, masks: Optional[torch.Tensor], sizes=None):
        if masks is not None and tensors.shape[0] != masks.shape[0]:
            raise AssertionError(
                f"tensors have batch size {tensors.shape[0]} but get {masks.shape[0]} for mask.")
        self.tensors = tensors
        self.masks = masks
        self.sizes = sizes

    def to(self, device, non_blocking: bool = False) -> "NestedTensor":
        masks = None if self.masks is None else self.masks.to(device, non_blocking=non_blocking)
        return NestedTensor(self.tensors.to(device, non_blocking=non_blocking), masks, self.sizes)

    def decompose(self):
        return self.tensors, self.masks

    def __repr__(self):
        return str(self.tensors)


def tensor_list_to_nested_tensor(tensor_list: List[torch.Tensor], size_divisibility: int = 32) -> NestedTensor:
    """Zero-pad (C,H,W) images to the common max size rounded up to ``size_divisibility``."""
    first = tensor_list[0]
    assert first.dim() == 3, f"Tensor should have 3 dimensions, but get {first.dim()}"
    channels = first.shape[0]
    height = max(t.shape[1] for t in tensor_list)
    width = max(t.shape[2] for t in tensor_list)
    if size_divisibility > 0:
        d = size_divisibility
        height = -(-height // d) * d
        width = -(-width // d) * d
    batch = first.new_zeros((len(tensor_list), channels, height, width))
    masks = torch.ones((len(tensor_list), height, width), dtype=torch.bool, device=first.device)
    for i, img in enumerate(tensor_list):
        assert img.shape[0] == channels, "Tensor channel size should be equal."
        batch[i, :, : img.shape[1], : img.shape[2]].copy_(img)
        masks[i, : img.shape[1], : img.shape[2]] = False
    sizes = ((height, width),) + tuple((int(t.shape[1]), int(t.shape[2])) for t in tensor_list)
    return NestedTensor(batch, masks, sizes=sizes)
