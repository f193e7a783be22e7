"""Builds tests/_build/libdevmath.so: the product's device-side scalar math compiled for the host (tests/cpp/devmath_host.cpp), and binds
it.  TEST INFRASTRUCTURE: a second compilation of product source, so that tests without a GPU can hold the device's solve / exp /
quaternion / shader-math / window-walk code to the oracle, numpy and SciPy."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "maskfusion_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libdevmath.so")
SLICES = [("mf_odometry.hip", "ldlt6_solve"), ("mf_odometry.hip", "gn_update_from_x"), ("mf_odometry.hip", "gn_solve_update_serial"),
          ("mf_surfel.hip", "window_slots_literal")]

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lib = None


def cut_function(text: str, name: str) -> str:
    """the definition of the __device__ function `name` (signature to matching closing brace), `__device__ __forceinline__` included"""
    m = re.search(r"^(?:template\s*<[^>]*>\s*\n)?__device__[^\n;{]*\b" + re.escape(name) + r"\s*\(", text, re.M)
    assert m, name
    i = text.index("{", m.end())
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(text[j], 0)
        j += 1
        if depth == 0:
            return text[m.start():j]


def translation_unit() -> str:
    parts = []
    for fn, name in SLICES:
        parts.append(cut_function(open(os.path.join(CSRC, fn)).read(), name))
    api = open(os.path.join(HERE, "cpp", "devmath_host.cpp")).read()
    assert api.count("\nDEVMATH_SLICES\n") == 1
    return api.replace("\nDEVMATH_SLICES\n", "\n" + "\n\n".join(parts) + "\n")


def build() -> str:
    deps = [os.path.join(CSRC, f) for f in ("mf_device.h", "mf_internal.h", "mf_labels.h", "mf_odometry.hip", "mf_surfel.hip")]
    deps += [os.path.join(HERE, "cpp", "devmath_host.cpp"), os.path.abspath(__file__)]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"),
           "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-x", "c++", "-", "-o", LIB]
    subprocess.run(cmd, input=translation_unit().encode(), check=True)
    return LIB


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        for n in ("dm_shader_exp", "dm_shader_acos"):
            getattr(L, n).argtypes = [C.c_float]; getattr(L, n).restype = C.c_float
        L.dm_surfel_radius.argtypes = [C.c_float] * 6; L.dm_surfel_radius.restype = C.c_float
        L.dm_surfel_confidence.argtypes = [C.c_float] * 7; L.dm_surfel_confidence.restype = C.c_float
        L.dm_encode_color.argtypes = [C.c_float] * 3; L.dm_encode_color.restype = C.c_float
        L.dm_decode_color.argtypes = [C.c_float, f32p]
        L.dm_mask_id.argtypes = [C.c_int, C.c_int]; L.dm_mask_id.restype = C.c_int
        L.dm_m33_inverse.argtypes = [f32p, f32p]
        L.dm_rodrigues.argtypes = [f64p, f64p]
        L.dm_rodrigues2.argtypes = [f32p, f64p]
        L.dm_quat_from_rot.argtypes = [f32p, f32p]
        L.dm_window_slots_literal.argtypes = [C.c_float, C.c_int, i32p, i32p]
        L.dm_gn_solve_update.argtypes = [f64p, f64p, f32p, f32p, f64p, f64p, f32p, f32p, f32p, f32p, f32p]
        L.dm_pose_derive.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p, f32p, C.c_int]
        _lib = L
    return _lib
