"""Builds oracle/_ref/libmf_track.so: the reference's Gauss-Newton tracking loop, RGBDOdometry::getIncrementalTransformation
(Core/Utils/RGBDOdometry.cpp:227-497: SO(3) pre-alignment, the three-level ICP + photometric loop, the joint solve, computeUpdateSE3,
the 0.3 m rule), compiled from the reference's own text on top of the reference's own device functions (oracle/_ref objects of
build_ref.py) -- what pins the oracle's restatement of the host half of SURVEY.md rows a8-a12 (mfo_track_icp, mfo_track_rgbd,
mfo_so3_prealign).

TEST INFRASTRUCTURE ONLY.  Nothing is copied into the repository: the member-function definition and Core/Utils/OdometryProvider.h are
cut out of the reference files in memory, put where oracle/track_shim/mftrack_api.cpp says MFTRACK_SLICE / MFTRACK_ODOMETRY_PROVIDER
and piped to g++ on stdin, verbatim.  oracle/eigen_shim stands in for the fixed-size slice of Eigen the loop uses.
Output: oracle/_ref/libmf_track.so.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

try:
    from . import build_ref
except ImportError:                      # run as a script: python oracle/build_track.py
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import build_ref

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "track_shim")
EIGEN = os.path.join(HERE, "eigen_shim")
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libmf_track.so")
REF_CORE = os.environ.get("MF_REFERENCE_CORE_DIR", "/root/reference/Core")
SRC = os.path.join(REF_CORE, "Utils", "RGBDOdometry.cpp")
PROVIDER = os.path.join(REF_CORE, "Utils", "OdometryProvider.h")
START, END = "Eigen::Matrix4f RGBDOdometry::getIncrementalTransformation(", "Eigen::MatrixXd RGBDOdometry::getCovariance()"
REF_OBJS = ["reduce.o", "cudafuncs.o", "device_memory.o", "mfref_runtime.o"]


def reference_available() -> bool:
    return os.path.exists(SRC) and os.path.exists(PROVIDER) and build_ref.reference_available()


def slice_of_reference() -> str:
    lines = open(SRC, encoding="utf-8", errors="replace").read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(START))
    b = next(i for i in range(a, len(lines)) if lines[i].startswith(END))
    return "\n".join(lines[a:b])


def provider_of_reference() -> str:
    text = open(PROVIDER, encoding="utf-8", errors="replace").read()
    return re.sub(r"/\*.*?\*/", "", text, count=1, flags=re.S)          # licence header


def translation_unit() -> str:
    api = open(os.path.join(SHIM, "mftrack_api.cpp")).read()
    assert api.count("\nMFTRACK_SLICE\n") == 1 and api.count("\nMFTRACK_ODOMETRY_PROVIDER\n") == 1
    api = api.replace("\nMFTRACK_ODOMETRY_PROVIDER\n", "\n" + provider_of_reference() + "\n")
    return api.replace("\nMFTRACK_SLICE\n", "\n" + slice_of_reference() + "\n")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SHIM, f) for f in os.listdir(SHIM)] + [os.path.join(EIGEN, "Eigen", f) for f in ("Core", "Geometry")]
    deps += [os.path.abspath(__file__), SRC, PROVIDER] + [os.path.join(OUT, o) for o in REF_OBJS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False) -> str | None:
    """Returns the library path, or None when neither the reference nor a prebuilt library are present."""
    if not reference_available():
        return LIB if os.path.exists(LIB) else None
    build_ref.build()
    if not (force or _stale()):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, "mf_track.o")
    flags = ["-std=c++14", "-O2", "-ffp-contract=off", "-fPIC", "-w", "-I", EIGEN, "-I", build_ref.SHIM, "-iquote", build_ref.REF_CUDA,
             "-iquote", os.path.join(build_ref.REF_CUDA, "containers"), "-I", build_ref.REF_CUDA]
    subprocess.run(["g++", *flags, "-x", "c++", "-c", "-", "-o", obj], input=translation_unit().encode(), check=True)
    subprocess.check_call(["g++", "-shared", "-o", LIB, obj, *[os.path.join(OUT, o) for o in REF_OBJS]])
    return LIB


if __name__ == "__main__":
    if "--print" in sys.argv:
        print(translation_unit())
    else:
        print(build(force="--force" in sys.argv))
