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


def probe(name, value, bound, tight=None, **kw):
    """Assert ``value <= bound`` and, on the GPU box (gpurun_out/ exists), log the measured value next to the bound in force and a
    candidate tighter bound to gpurun_out/tolerance_probe.jsonl -- the record the tolerances of the parity tests are set from
    (profiles/r5_tolerance_probe.jsonl: every bound carries its measured margin)."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "tolerance_probe.jsonl"), "a") as f:
            f.write(json.dumps(dict(name=name, value=float(value), bound=float(bound), tight=None if tight is None else float(tight),
                                    holds_tight=None if tight is None else bool(value <= tight), **kw)) + "\n")
    assert value <= bound, (name, value, bound, kw)
