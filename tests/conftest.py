import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """HipContext: counterpart of the reference's helpers.CudaContext (test/helpers.py:29-74)."""
    # torch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's): import it before libmifft.so is loaded so
    # that one HIP runtime serves both and the torch-interop test can run (bench.py does the same)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    from helpers import HipContext
    return HipContext()
