import os
import sys

# (before anything initialises the HIP runtime — _have_gpu() below does: streams share this many hardware queues, nmpc_amd/csrc/capi.hip)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def _have_gpu() -> bool:
    try:
        import ctypes as C
        for p in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                hip = C.CDLL(p)
                break
            except OSError:
                continue
        else:
            return False
        n = C.c_int(0)
        return hip.hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


HAVE_GPU = _have_gpu()


def pytest_collection_modifyitems(config, items):
    if HAVE_GPU:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
