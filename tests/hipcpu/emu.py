"""Loads tests/_emu/libmaskfusion_emu.so (the product's kernels compiled for and executed on the CPU, tests/hipcpu/build.py) behind the
same ctypes table as the real library.  TEST TOOLING: `activate()` swaps it into maskfusion_amd.lib for the current PROCESS so that the
Python mirror (maskfusion_amd.api) drives it; only tests call this, explicitly.  "Device pointers" are host pointers here."""
from __future__ import annotations

import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build as _build  # noqa: E402

_emu = None


def load():
    """MF_EMU_COOP=1 selects the build that can run all workgroups of a launch at once (kernels named in HIPCPU_COOPERATIVE);
    MF_EMU_ASAN=1 the AddressSanitizer build (start Python under LD_PRELOAD=libasan.so, see tests/hipcpu/build.py)"""
    global _emu
    if _emu is None:
        from maskfusion_amd import lib as mflib
        L = C.CDLL(_build.build(coop=os.environ.get("MF_EMU_COOP") == "1", asan=os.environ.get("MF_EMU_ASAN") == "1",
                                 ubsan=os.environ.get("MF_EMU_UBSAN") == "1"))
        for name, (res, args) in mflib.SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _emu = L
    return _emu


def activate():
    """maskfusion_amd.lib.load() returns the emulated library from now on (this process only)"""
    from maskfusion_amd import lib as mflib
    mflib._lib = load()
    return mflib._lib
