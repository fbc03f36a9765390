import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """libepiekf.so, built in-tree if missing/stale (hipcc cross-compiles on CPU)."""
    from epidemicmodeling_amd import _build, _lib
    if _build.is_stale():
        _build.build_library()
    return _lib.lib()


@pytest.fixture(scope="session")
def gpu_device(hip_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (the HIP path has no CPU fallback)")
    return "cuda:0"
