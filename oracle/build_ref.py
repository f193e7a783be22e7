"""Builds oracle/_ref/libmf_ref.so: the reference's own CUDA translation units compiled for the CPU.

TEST INFRASTRUCTURE ONLY.  Sources are read where they lie under /root/reference (Core/Cuda/reduce.cu, cudafuncs.cu,
segmentation.cu, containers/device_memory.cpp) and are never written into the repository: the only edit -- the
`kernel<<<grid, block>>>(args)` launch syntax, which is not C++, becomes MFREF_LAUNCH(kernel, grid, block, args) -- is
applied in memory and the text is piped to g++ on stdin.  oracle/ref_shim/ supplies stand-ins for the CUDA toolkit
headers and a fiber-based grid/block/thread runtime (mfref_cuda.h, mfref_runtime.cpp) plus plain-pointer C entry points
around the reference's host wrappers (mfref_api.cpp).  Output: oracle/_ref/libmf_ref.so (+ object files), git-ignored.

The GPU box has no /root/reference: there the prebuilt .so (it travels with the snapshot) is used as is.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "ref_shim")
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libmf_ref.so")
REF_CUDA = os.environ.get("MF_REFERENCE_CUDA_DIR", "/root/reference/Core/Cuda")
REF_UNITS = ["reduce.cu", "cudafuncs.cu", "segmentation.cu", os.path.join("containers", "device_memory.cpp")]
SHIM_UNITS = ["mfref_runtime.cpp", "mfref_api.cpp"]
CXXFLAGS = ["-std=c++14", "-O2", "-ffp-contract=off", "-fPIC", "-w", "-I", SHIM, "-iquote", REF_CUDA, "-iquote",
            os.path.join(REF_CUDA, "containers"), "-I", REF_CUDA, "-include", os.path.join(SHIM, "mfref_cuda.h")]

_LAUNCH = re.compile(r"([A-Za-z_]\w*(?:<\s*\w+\s*>)?)\s*<<\s*<\s*([^,<>]+?)\s*,\s*([^<>]+?)\s*>>>\s*\(")


def rewrite_launches(text: str) -> str:
    """kernel<<<grid, block>>>(args...) -> MFREF_LAUNCH(kernel, grid, block, args...)"""
    return _LAUNCH.sub(lambda m: f"MFREF_LAUNCH({m.group(1)}, {m.group(2)}, {m.group(3)}, ", text)


def reference_available() -> bool:
    return all(os.path.exists(os.path.join(REF_CUDA, u)) for u in REF_UNITS)


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SHIM, f) for f in os.listdir(SHIM)] + [os.path.abspath(__file__)]
    deps += [os.path.join(REF_CUDA, u) for u in REF_UNITS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False) -> str | None:
    """Returns the library path, or None when neither the reference sources nor a prebuilt library are present."""
    if not reference_available():
        return LIB if os.path.exists(LIB) else None
    if not (force or _stale()):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for unit in REF_UNITS:
        with open(os.path.join(REF_CUDA, unit), "r", encoding="utf-8", errors="replace") as f:
            src = rewrite_launches(f.read())
        obj = os.path.join(OUT, os.path.basename(unit).rsplit(".", 1)[0] + ".o")
        subprocess.run(["g++", *CXXFLAGS, "-x", "c++", "-c", "-", "-o", obj], input=src.encode(), check=True)
        objs.append(obj)
    for unit in SHIM_UNITS:
        obj = os.path.join(OUT, unit.replace(".cpp", ".o"))
        subprocess.check_call(["g++", *CXXFLAGS, "-c", os.path.join(SHIM, unit), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["g++", "-shared", "-o", LIB, *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
