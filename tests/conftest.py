import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))   # CPU references: the GPU box grants ~16 cores
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip():
    """The C-ABI library wrapper; GPU tests fail loudly if it is missing."""
    import torch
    from unet_bssfp_amd import _lib
    assert torch.cuda.is_available(), "gpu-marked test needs a GPU"
    return _lib.load()
