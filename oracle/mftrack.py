"""ctypes binding of oracle/_ref/libmf_track.so: the reference's RGBDOdometry::getIncrementalTransformation compiled from the
reference's own text over the reference's own device functions (oracle/build_track.py).  TEST INFRASTRUCTURE ONLY; same call shape as
mfo_rgbd.track_rgbd / mfo.track_icp."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build_track

_lib = None
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def available() -> bool:
    return build_track.reference_available() or os.path.exists(build_track.LIB)


def lib():
    global _lib
    if _lib is None:
        path = build_track.build()
        if path is None:
            raise RuntimeError("oracle/_ref/libmf_track.so is absent and /root/reference is not here to build it from")
        _lib = C.CDLL(path)
        pp = C.POINTER(C.c_void_p)
        _lib.mftrack_run.argtypes = [pp] * 8 + [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_int] * 4 + [C.c_float] * 3 + \
            [f32p, f32p, f32p, f32p, f64p, f64p, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]
        _lib.mftrack_run.restype = C.c_int
    return _lib


def _pyr3(lst, dtype, keep):
    if lst is None:
        return None
    a = (C.c_void_p * 3)()
    for i, m in enumerate(lst):
        m = np.ascontiguousarray(m, dtype)
        keep.append(m)
        a[i] = m.ctypes.data
    keep.append(a)
    return C.cast(a, C.POINTER(C.c_void_p))


def track(curr_v, curr_n, prev_v, prev_n, W, H, fx, fy, cx, cy, R, t, *, last_depth=None, next_depth=None, last_image=None,
          next_image=None, last_next2=None, pyramid=True, fast_odom=False, so3=False, rgb_only=False, icp_weight=100.0,
          dist_thresh=0.10, angle_thresh=float(np.sin(np.deg2rad(20.0)))):
    """One call of the reference's getIncrementalTransformation.  Pyramids are 3-element lists (level 0..2), maps planar (3, h, w).
    Returns (R 3x3, t, inc 4x4, stats dict (the six last* members + launch counts), lastA 6x6, lastb 6)."""
    keep = []
    ln2 = None
    if last_next2 is not None:
        ln2 = np.ascontiguousarray(last_next2, np.uint8)
        keep.append(ln2)
    Rf = np.ascontiguousarray(R, np.float32).reshape(9).copy()
    tf = np.ascontiguousarray(t, np.float32).copy()
    inc = np.zeros(16, np.float32)
    st = np.zeros(6, np.float32)
    A = np.zeros(36, np.float64)
    b = np.zeros(6, np.float64)
    ticks = np.zeros(4, np.int32)
    lib().mftrack_run(_pyr3(curr_v, np.float32, keep), _pyr3(curr_n, np.float32, keep), _pyr3(prev_v, np.float32, keep),
                      _pyr3(prev_n, np.float32, keep), _pyr3(last_depth, np.float32, keep), _pyr3(next_depth, np.float32, keep),
                      _pyr3(last_image, np.uint8, keep), _pyr3(next_image, np.uint8, keep), ln2.ctypes.data if ln2 is not None else None,
                      W, H, fx, fy, cx, cy, int(pyramid), int(fast_odom), int(so3), int(rgb_only), icp_weight, dist_thresh, angle_thresh,
                      Rf, tf, inc, st, A, b, ticks)
    names = ["lastICPError", "lastICPCount", "lastRGBError", "lastRGBCount", "lastSO3Error", "lastSO3Count"]
    stats = dict(zip(names, map(float, st)))
    stats.update(zip(["so3Steps", "rgbResiduals", "icpSteps", "rgbSteps"], map(int, ticks)))       # launches of each device function
    return Rf.reshape(3, 3), tf, inc.reshape(4, 4).T.copy(), stats, A.reshape(6, 6), b
