"""Builds oracle/_ref/libmf_seg.so: the host half of the reference's MfSegmentation::performSegmentation (label propagation,
Core/Segmentation/MfSegmentation.cpp:219-523) compiled from the reference's own text -- what pins the oracle's restatement of the CPU
stage of SURVEY.md row a20 (and through it the device label stage, which tests/test_gpu_labels.py holds to the oracle exactly).

TEST INFRASTRUCTURE ONLY.  Nothing is copied into the repository: the statement range is cut out of the reference file in memory, put
where oracle/cv_shim/mfseg_api.cpp says MFSEG_SLICE, preceded by oracle/cv_shim/mfcv.h (cv::Mat / Eigen::MatrixXi stand-ins; the three
OpenCV primitives underneath are the oracle's restatements) and the reference's Core/Utils/BoundingBox.h, and piped to g++ on stdin.
The slice is taken verbatim (its #ifdef'd debug blocks stay, inactive).  Output: oracle/_ref/libmf_seg.so.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "cv_shim")
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libmf_seg.so")
REF_CORE = os.environ.get("MF_REFERENCE_CORE_DIR", "/root/reference/Core")
SRC = os.path.join(REF_CORE, "Segmentation", "MfSegmentation.cpp")
BBOX = os.path.join(REF_CORE, "Utils", "BoundingBox.h")
START, END = "// Build use ignore map", "cudaDeviceSynchronize();"


def reference_available() -> bool:
    return os.path.exists(SRC) and os.path.exists(BBOX)


def slice_of_reference() -> str:
    lines = open(SRC, encoding="utf-8", errors="replace").read().split("\n")
    a = next(i for i, l in enumerate(lines) if START in l)
    b = next(i for i in range(a, len(lines)) if END in lines[i])
    return "\n".join(lines[a:b])


def translation_unit() -> str:
    api = open(os.path.join(SHIM, "mfseg_api.cpp")).read()
    assert api.count("\nMFSEG_SLICE\n") == 1
    bbox = open(BBOX, encoding="utf-8", errors="replace").read().replace("#pragma once", "")
    return '#include "mfcv.h"\n' + bbox + "\n" + api.replace("\nMFSEG_SLICE\n", "\n" + slice_of_reference() + "\n")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SHIM, f) for f in os.listdir(SHIM)] + [os.path.abspath(__file__), os.path.join(HERE, "mf_oracle.c"), SRC, BBOX]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False) -> str | None:
    """Returns the library path, or None when neither the reference nor a prebuilt library are present."""
    if not reference_available():
        return LIB if os.path.exists(LIB) else None
    if not (force or _stale()):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "mf_seg.o")
    subprocess.run(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fPIC", "-w", "-I", SHIM, "-x", "c++", "-c", "-", "-o", obj],
                   input=translation_unit().encode(), check=True)
    orc = os.path.join(OUT, "mf_oracle_for_seg.o")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-std=gnu11", "-w", "-c", os.path.join(HERE, "mf_oracle.c"), "-o", orc])
    subprocess.check_call(["g++", "-shared", "-fopenmp", "-o", LIB, obj, orc, "-lm"])
    return LIB


if __name__ == "__main__":
    if "--print" in sys.argv:
        print(translation_unit())
    else:
        print(build(force="--force" in sys.argv))
