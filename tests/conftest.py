import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a visible MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def ctx():
    """One HIP context per test session (GPU tests only)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from bk_amd import hip
    c = hip.Context(0)
    yield c
    c.close()
