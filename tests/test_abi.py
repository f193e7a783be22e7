"""The C-ABI library loads and exports exactly the symbols include/maskfusion_amd.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "maskfusion_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_binding_table():
    from maskfusion_amd.lib import SYMBOLS
    assert header_symbols() == sorted(SYMBOLS)


def test_library_exports_every_symbol():
    from maskfusion_amd import build as b
    from maskfusion_amd.lib import LIB_PATH
    b.build()
    lib = ctypes.CDLL(LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_default_config_and_no_gpu_is_loud():
    import ctypes as C
    from maskfusion_amd.lib import load, Config
    L = load()
    cfg = Config()
    assert L.mf_default_config(C.byref(cfg), 640, 480, 528.0, 528.0, 320.0, 240.0) == 0
    assert (cfg.time_delta, cfg.conf_global, cfg.depth_cutoff, cfg.icp_weight) == (200, 4.0, 3.0, 10.0)
    assert cfg.num_gsurfels == 9437184 and cfg.num_osurfels == 1048576
    import torch
    if not torch.cuda.is_available():
        h = C.c_void_p()
        assert L.mf_create(C.byref(cfg), C.byref(h)) == -2  # MF_ENODEV: no CPU fallback
        assert not h.value


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "maskfusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "mf_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
