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
    from oracle import mfo
    mfo.lib()
    return mfo


@pytest.fixture(scope="session")
def hip():
    """The HIP extension + a CUDA(=HIP) torch device; GPU tests fail loudly if either is missing."""
    if os.environ.get("MF_EMU") == "1":
        # explicit opt-in: the product's kernels compiled for and executed on the CPU (tests/hipcpu) -- a logic check without a GPU
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipcpu"))
        import emu
        return emu.activate()
    import torch
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    from maskfusion_amd.lib import load
    return load()
