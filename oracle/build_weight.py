"""Builds oracle/_ref/libmf_weight.so: the reference's Model::computeFusionWeight and Model::rodrigues2 (Core/Model/Model.cpp:449-464,
891-932) with Model::getLastTransform's expression (Core/Model/Model.h:239), compiled from the reference's own text -- what pins the oracle's
restatement of SURVEY.md row a15 (mfo_fusion_weight; the device's pose_derive is bit-identical to that, tests/test_devmath_host.py).

TEST INFRASTRUCTURE ONLY.  Nothing is copied into the repository: the two member-function definitions and the one expression are cut out of
the reference files in memory, put where oracle/weight_shim/mfweight_api.cpp says MFWEIGHT_SLICES / MFWEIGHT_LAST_TRANSFORM and piped to g++.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "weight_shim")
EIGEN = os.path.join(HERE, "eigen_shim")
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libmf_weight.so")
REF_CORE = os.environ.get("MF_REFERENCE_CORE_DIR", "/root/reference/Core")
SRC = os.path.join(REF_CORE, "Model", "Model.cpp")
HDR = os.path.join(REF_CORE, "Model", "Model.h")


def reference_available() -> bool:
    return os.path.exists(SRC) and os.path.exists(HDR)


def cut_definition(text: str, start: str) -> str:
    a = text.index(start)
    i = text.index("{", a)
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
        if depth == 0:
            return text[a:j]


def translation_unit() -> str:
    src = open(SRC, encoding="utf-8", errors="replace").read()
    hdr = open(HDR, encoding="utf-8", errors="replace").read()
    slices = cut_definition(src, "float Model::computeFusionWeight(") + "\n\n" + cut_definition(src, "Eigen::Vector3f Model::rodrigues2(")
    m = re.search(r"getLastTransform\(\)\s*const\s*\{\s*return\s+([^;]+);", hdr)
    assert m, "getLastTransform not found"
    api = open(os.path.join(SHIM, "mfweight_api.cpp")).read()
    assert api.count("\nMFWEIGHT_SLICES\n") == 1 and api.count("MFWEIGHT_LAST_TRANSFORM;") == 1
    return api.replace("MFWEIGHT_LAST_TRANSFORM;", m.group(1) + ";").replace("\nMFWEIGHT_SLICES\n", "\n" + slices + "\n")


def build(force: bool = False) -> str | None:
    if not reference_available():
        return LIB if os.path.exists(LIB) else None
    deps = [os.path.join(SHIM, "mfweight_api.cpp"), os.path.join(EIGEN, "Eigen", "Core"), os.path.join(EIGEN, "Eigen", "Geometry"), os.path.abspath(__file__), SRC, HDR]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    subprocess.run(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-I", EIGEN, "-x", "c++", "-", "-o", LIB],
                   input=translation_unit().encode(), check=True)
    return LIB


if __name__ == "__main__":
    if "--print" in sys.argv:
        print(translation_unit())
    else:
        print(build(force="--force" in sys.argv))
