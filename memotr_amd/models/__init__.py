"""Model factory with the reference's entry point name (models/__init__.py:9-15): ``build_model(config)``
returns a MeMOTR on ``config["DEVICE"]`` (one GPU per process: the device index is the process rank)."""
import torch

from ..utils.utils import distributed_rank
from .memotr import MeMOTR
from .memotr import build as build_memotr


def _target_device(config: dict) -> torch.device:
    on_gpu = config["DEVICE"] == "cuda" and config["AVAILABLE_GPUS"] is not None
    return torch.device("cuda", distributed_rank()) if on_gpu else torch.device(config["DEVICE"])


def build_model(config: dict) -> MeMOTR:
    device = _target_device(config)
    if device.type == "cuda":
        from ..modules.linear import configure_blas
        configure_blas()          # rocBLAS for mm / bmm: see the measurements quoted there
    return build_memotr(config=config).to(device=device)
