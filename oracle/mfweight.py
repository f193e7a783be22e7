"""ctypes binding of oracle/_ref/libmf_weight.so: the reference's Model::computeFusionWeight / rodrigues2 compiled from its own text
(oracle/build_weight.py).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:
    from . import build_weight
except ImportError:
    import build_weight

_lib = None
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def available() -> bool:
    return build_weight.reference_available() or os.path.exists(build_weight.LIB)


def lib():
    global _lib
    if _lib is None:
        path = build_weight.build()
        if path is None:
            raise RuntimeError("oracle/_ref/libmf_weight.so is absent and /root/reference is not here to build it from")
        _lib = C.CDLL(path)
        _lib.mfweight_fusion_weight.argtypes = [f32p, f32p, C.c_float]
        _lib.mfweight_fusion_weight.restype = C.c_float
        _lib.mfweight_rodrigues2.argtypes = [f32p, f32p]
    return _lib


def fusion_weight(T, T_last, weight_multiplier=1.0) -> float:
    """T, T_last: 4x4 (row-major numpy); Eigen::Matrix4f storage is column-major"""
    a = np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(16))
    b = np.ascontiguousarray(np.asarray(T_last, np.float32).T.reshape(16))
    return float(lib().mfweight_fusion_weight(a, b, weight_multiplier))


def rodrigues2(R) -> np.ndarray:
    out = np.zeros(3, np.float32)
    lib().mfweight_rodrigues2(np.ascontiguousarray(np.asarray(R, np.float32).T.reshape(9)), out)
    return out
