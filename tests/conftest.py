import os
import sys

import pytest

# the oracle's OpenMP regions are short per-kernel loops: beyond ~16 threads they get slower (bench.py's sweep on the 256-thread GPU box), and
# several xdist workers share the box (pytest.ini)
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _prebuild():
    """Every native artefact the tests load, built (when stale; a no-op otherwise) ONCE, serially, by the process that starts the xdist workers:
    the builders check time stamps and are not safe against each other."""
    import time
    t0 = time.time()
    steps = []

    def run(name, fn):
        t = time.time()
        try:
            fn()
        except BaseException as e:     # a builder that cannot run here (no hipcc, no /root/reference) fails the tests that need it, not the session
            steps.append(f"{name}: {type(e).__name__}")
            return
        if time.time() - t > 1.0:
            steps.append(f"{name}: {time.time() - t:.0f} s")

    from maskfusion_amd import build as product_build
    run("libmaskfusion_amd.so", product_build.build)
    from oracle import mfo
    run("libmf_oracle.so", mfo.build)
    for mod in ("build_ref", "build_glsl", "build_seg", "build_track", "build_weight", "build_io"):
        run("oracle/_ref " + mod, lambda mod=mod: __import__("oracle." + mod, fromlist=["build"]).build())
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import devmath
    run("libdevmath.so", devmath.build)
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipcpu"))
    import build as emu_build
    run("libmaskfusion_emu.so", emu_build.build)
    if steps or time.time() - t0 > 2.0:
        sys.stderr.write(f"[conftest] prebuild {time.time() - t0:.0f} s: " + "; ".join(steps) + "\n")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if not hasattr(config, "workerinput") and not config.getoption("collectonly", False) and os.environ.get("MF_NO_PREBUILD") != "1":
        _prebuild()


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


def pytest_sessionstart(session):
    """MF_TEST_PARAMS="key=value,key=value": implementation switches (mf_set_param) applied to every context the tests create -- tooling for
    bisecting a failure to a switch on the GPU box (e.g. MF_TEST_PARAMS=objectStream=0,fusedPreprocessLaunch=0); unset in every regular run."""
    spec = os.environ.get("MF_TEST_PARAMS", "")
    if not spec:
        return
    from maskfusion_amd import api
    pairs = [(kv.partition("=")[0], float(kv.partition("=")[2])) for kv in spec.split(",") if kv]
    orig = api.MaskFusion.__init__

    def patched(self, *a, **k):
        orig(self, *a, **k)
        for key, val in pairs:
            self.setParam(key, val)
    api.MaskFusion.__init__ = patched
