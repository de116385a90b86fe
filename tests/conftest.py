import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def es_ctx():
    """One library context on cuda:0 for the whole GPU session (fails loudly if the HIP library is missing)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from eigensolver_amd import _lib
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()
