import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_HAS_GPU = None


def _has_gpu():
    """Asked of the PRODUCT library (ctypes + HIP, no torch needed): can a context be created on device 0?  A box whose
    torch is CPU-only or missing must not turn the parity suite into 150 silent skips."""
    global _HAS_GPU
    if _HAS_GPU is None:
        try:
            import ctypes as C
            try:
                # tests that hand torch tensors to the library need torch's own bundled HIP runtime to be the one in the
                # process: load it BEFORE libi2s_hip.so pulls in /opt/rocm's (the other order leaves torch without GPUs)
                import torch
                torch.cuda.is_available()
            except Exception:
                pass
            from img2sgf_amd import _lib
            lib = _lib.load()
            ctx = C.c_void_p()
            ok = lib.dll.i2s_create(C.byref(ctx), 0, 1, 64, 64) == 0
            if ok:
                lib.dll.i2s_destroy(ctx)
            _HAS_GPU = ok
        except Exception:
            _HAS_GPU = False
    return _HAS_GPU


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    if (config.getoption("markexpr") or "").strip() == "gpu" and not config.getoption("collectonly"):
        raise pytest.UsageError("-m gpu was requested but libi2s_hip.so cannot create a context on device 0 "
                                "(library missing, or no MI355X visible): nothing would run")
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
