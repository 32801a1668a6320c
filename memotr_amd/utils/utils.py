"""Small helpers of the reference's utils/utils.py that the per-frame path touches."""
from __future__ import annotations

import os
import random

import numpy as np
import torch
import torch.distributed as dist
import yaml


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def distributed_rank() -> int:
    return dist.get_rank() if is_distributed() else 0


def distributed_world_size() -> int:
    return dist.get_world_size() if is_distributed() else 1


def is_main_process() -> bool:
    return distributed_rank() == 0


def set_seed(seed: int) -> None:
    """seed + rank, as utils/utils.py:37-38."""
    seed = seed + distributed_rank()
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def yaml_to_dict(path: str) -> dict:
    with open(path) as f:
        return yaml.load(f.read(), yaml.FullLoader)


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """logit with both odds clamped at eps (utils/utils.py:61-74)."""
    from ..functions import clip_ops
    if clip_ops.fused(x) and x.dtype == torch.float32:
        return clip_ops.inverse_sigmoid(x, eps)            # one kernel (and one backward) instead of five (twelve)
    return inverse_sigmoid_reference(x, eps)


def inverse_sigmoid_reference(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    # clamp(clamp(x, 0, 1), min=eps) == clamp(x, eps, 1) and clamp(1 - clamp(x, 0, 1), min=eps) == clamp(1 - x, eps, 1)
    # value for value and gradient mask for gradient mask (eps > 0): one kernel fewer, forward and backward
    return torch.log(x.clamp(min=eps, max=1) / (1 - x).clamp(min=eps, max=1))


def refine_boxes(delta: torch.Tensor, reference: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """sigmoid(delta + inverse_sigmoid(reference)): the decoder's iterative box refinement for 4-d references
    (models/deformable_decoder.py:139-149 of the reference)."""
    from ..functions import clip_ops
    if delta.shape == reference.shape and clip_ops.fused(delta, reference) and delta.dtype == torch.float32:
        return clip_ops.refine_boxes(delta, reference, eps)
    return (delta + inverse_sigmoid_reference(reference, eps)).sigmoid()
