import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle as o
    return o.Oracle()


@pytest.fixture(scope="session")
def gpu_ctx():
    # PyTorch wheels bundle their own HIP/HSA runtime; when libptcore's (/opt/rocm) initialises the
    # device first, torch's copy later reports "No HIP GPUs".  Tests that hand torch device buffers
    # to the ABI therefore need torch initialised first (bench.py does the same by construction).
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    from pathtracer_amd import core
    ctx = core.Context(0)
    yield ctx
    ctx.close()
