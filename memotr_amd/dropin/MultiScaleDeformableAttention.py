"""Put this directory on PYTHONPATH and the reference's
``import MultiScaleDeformableAttention as MSDA`` (models/ops/functions/ms_deform_attn_func.py:21)
resolves to the gfx950 implementation.  See INTEGRATION.md."""
from memotr_amd.MultiScaleDeformableAttention import (  # noqa: F401
    ms_deform_attn_backward,
    ms_deform_attn_forward,
)
