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


@pytest.fixture(scope="session", autouse=True)
def _literal_weight_rehearsal():
    """MF_LITERAL_WEIGHT=1 (with MF_EMU=1): rehearse the default flip of finding F5 -- the oracle in its literal fusion-weight mode and the
    CPU-executed test build compiled with literalFusionWeight on by default"""
    if os.environ.get("MF_LITERAL_WEIGHT") == "1":
        from oracle import mfo
        mfo.lib().mfo_set_weight_literal(1)
        # ... and every context the tests create through the Python mirror switches its literal mode on (this is what makes the rehearsal
        # work on the GPU as well, where the library's compiled-in default is not touched)
        from maskfusion_amd import api
        plain_init = api.MaskFusion.__init__

        def init_with_literal_weight(self, *a, **kw):
            plain_init(self, *a, **kw)
            self.setParam("literalFusionWeight", 1)
        api.MaskFusion.__init__ = init_with_literal_weight
    yield


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
