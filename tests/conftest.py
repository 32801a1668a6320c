import glob
import os
import sys

# ROCm 7.2: with the runtime's AQL-packet capture on (the default) a MEMSET node of a replayed hipGraph is not ordered
# behind the kernels before it (tools/graph_memset_probe.py); torch's multi-block reductions, and whatever else a
# library zeroes that way, then read garbage from the second replay on.  Read when the HIP runtime loads.
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """``gpu``-marked tests need a device: skip them (instead of erroring) on a host without one."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a GPU (MI355X): run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden_cases(pattern="msda_*.npz"):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, pattern)))


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from memotr_amd.build import build_lib
    build_lib()
    from memotr_amd import _lib
    return _lib


@pytest.fixture(scope="session")
def clip_lib():
    """libclip_ops_hip.so (fused small-tensor chains of the train step); built on demand."""
    from memotr_amd.build import build_clip_lib
    build_clip_lib()
    from memotr_amd import _clip_lib
    return _clip_lib


def has_gpu():
    import torch
    return torch.cuda.is_available()


_TEST_SITE = [1 << 40]


@pytest.fixture(autouse=True)
def _fresh_call_site(request):
    """Kernel selection keeps one record per CALL SITE (memotr_amd/csrc/msda_select.h; round 5: no longer per geometry --
    the measured off-window share is a property of a module's learnt offsets).  Direct operator calls of a test are
    untagged, so without this every GPU test would inherit the level the previous test's data left behind: each GPU
    test gets a call site of its own (the model's modules tag their calls themselves)."""
    if "gpu" not in request.keywords or not has_gpu():
        yield
        return
    from memotr_amd import MultiScaleDeformableAttention as MSDA
    from memotr_amd import _lib
    # ... and starts without records: msda_selector_poll()'s signature covers every record of the device, those of
    # earlier tests' modules included (a record left at the top level announces a probe every 32nd poll, and a graph
    # cache of the next test captures once more than that test counts on)
    _lib.selector_reset()
    _TEST_SITE[0] += 1
    MSDA.set_call_site(_TEST_SITE[0])
    yield
    MSDA.set_call_site(0)
