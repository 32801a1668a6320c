"""memotr_amd -- MI355X-native (gfx950) hot path of MeMOTR.

Package layout mirrors the reference's interface for the path it replaces:

  memotr_amd.csrc/                      hand-written HIP kernels + the C ABI (include/msda_hip.h)
  memotr_amd.MultiScaleDeformableAttention   same two entry points as the reference's compiled
                                        extension (models/ops/src/vision.cpp:13-16)
  memotr_amd.functions / .modules       MSDeformAttnFunction / MSDeformAttn
                                        (models/ops/functions, models/ops/modules)
  memotr_amd.models / .structures / .utils   per-frame model path (models/memotr.py etc.)

The HIP library is mandatory: importing the operator without a built
``memotr_amd/lib/libmsda_hip.so`` raises (there is no CPU fallback in the product).
"""

__version__ = "0.1.0"
