"""ctypes binding of the photometric / SO(3) part of the oracle (oracle/mf_oracle.c).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import mfo

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")

DATATERM = np.dtype([("zx", np.int16), ("zy", np.int16), ("ox", np.int16), ("oy", np.int16), ("diff", np.float32),
                     ("valid", np.int32)])


class TrackStats(C.Structure):
    _fields_ = [("lastICPError", C.c_float), ("lastICPCount", C.c_float), ("lastRGBError", C.c_float),
                ("lastRGBCount", C.c_float), ("lastSO3Error", C.c_float), ("lastSO3Count", C.c_float),
                ("so3Iterations", C.c_int), ("iterationsRun", C.c_int), ("rejected", C.c_int)]


class RgbdInputs(C.Structure):
    _fields_ = [("lastDepth", C.c_void_p * 3), ("nextDepth", C.c_void_p * 3), ("lastImage", C.c_void_p * 3),
                ("nextImage", C.c_void_p * 3), ("lastNextImage2", C.c_void_p)]


_ready = False


def rlib():
    global _ready
    L = mfo.lib()
    if _ready:
        return L
    L.mfo_vertices_to_depth.argtypes = [f32p, f32p, C.c_int, C.c_float]
    L.mfo_image_to_intensity.argtypes = [u8p, C.c_int, u8p, C.c_int]
    L.mfo_derivative_images.argtypes = [u8p, i16p, i16p, C.c_int, C.c_int]
    L.mfo_project_to_cloud.argtypes = [f32p, f32p, C.c_int, C.c_int] + [C.c_float] * 4
    L.mfo_rgb_residual.argtypes = [C.c_float, i16p, i16p, f32p, f32p, u8p, u8p, C.c_void_p, C.c_float, f32p, f32p, C.c_int,
                                   C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.mfo_rgb_step.argtypes = [C.c_void_p, C.c_float, f32p, C.c_float, C.c_float, i16p, i16p, C.c_float, C.c_int, C.c_int,
                               f32p, f32p]
    L.mfo_so3_step.argtypes = [u8p, u8p, f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p, f32p]
    L.mfo_ldlt3f_solve.argtypes = [f32p, f32p, f32p]
    L.mfo_so3_prealign.argtypes = [u8p, u8p, C.c_int, C.c_int] + [C.c_float] * 4 + \
        [f64p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.mfo_track_rgbd.argtypes = [C.POINTER(C.c_void_p)] * 4 + [C.POINTER(RgbdInputs), C.c_int, C.c_int] + [C.c_float] * 4 + \
        [C.POINTER(mfo.TrackOpts), f32p, f32p, f32p, C.POINTER(TrackStats)]
    L.mfo_get_track_stats.argtypes = [C.c_void_p, C.POINTER(TrackStats)]
    _ready = True
    return L


def vertices_to_depth(v4, cutoff=6.0):
    H, W, _ = v4.shape
    d = np.empty((H, W), np.float32)
    rlib().mfo_vertices_to_depth(np.ascontiguousarray(v4, np.float32).reshape(-1), d, W * H, cutoff)
    return d


def image_to_intensity(img):
    H, W, ch = img.shape
    out = np.empty((H, W), np.uint8)
    rlib().mfo_image_to_intensity(np.ascontiguousarray(img, np.uint8).reshape(-1), ch, out, W * H)
    return out


def derivative_images(img):
    H, W = img.shape
    dx = np.empty((H, W), np.int16)
    dy = np.empty((H, W), np.int16)
    rlib().mfo_derivative_images(np.ascontiguousarray(img, np.uint8), dx, dy, W, H)
    return dx, dy


def project_to_cloud(depth, fx, fy, cx, cy):
    H, W = depth.shape
    c = np.empty((H, W, 3), np.float32)
    rlib().mfo_project_to_cloud(np.ascontiguousarray(depth, np.float32), c.reshape(-1), W, H, fx, fy, cx, cy)
    return c


def rgb_residual(min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, kt, krkinv, max_depth_delta=0.07):
    H, W = next_image.shape
    corres = np.zeros(W * H, DATATERM)
    sig, cnt = C.c_int32(0), C.c_int32(0)
    rlib().mfo_rgb_residual(min_scale, dIdx, dIdy, np.ascontiguousarray(last_depth, np.float32),
                            np.ascontiguousarray(next_depth, np.float32), last_image, next_image, corres.ctypes.data,
                            max_depth_delta, np.ascontiguousarray(kt, np.float32),
                            np.ascontiguousarray(krkinv, np.float32).reshape(9), W, H, C.byref(sig), C.byref(cnt))
    return corres, sig.value, cnt.value


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, W, H, sobel_scale=0.125):
    A = np.zeros(36, np.float32)
    b = np.zeros(6, np.float32)
    rlib().mfo_rgb_step(corres.ctypes.data, sigma, np.ascontiguousarray(cloud, np.float32).reshape(-1), fx, fy, dIdx, dIdy,
                        sobel_scale, W, H, A, b)
    return A.reshape(6, 6), b


def so3_step(last_image, next_image, image_basis, kinv, krlr):
    H, W = next_image.shape
    A = np.zeros(9, np.float32)
    b = np.zeros(3, np.float32)
    r = np.zeros(2, np.float32)
    f = lambda a: np.ascontiguousarray(a, np.float32).reshape(9)
    rlib().mfo_so3_step(last_image, next_image, f(image_basis), f(kinv), f(krlr), W, H, A, b, r)
    return A.reshape(3, 3), b, r


def so3_prealign(last2, next2, fx2, fy2, cx2, cy2):
    H, W = next2.shape
    R = np.zeros(9, np.float64)
    e, c, it = C.c_float(0), C.c_float(0), C.c_int(0)
    rlib().mfo_so3_prealign(last2, next2, W, H, fx2, fy2, cx2, cy2, R, C.byref(e), C.byref(c), C.byref(it))
    return R.reshape(3, 3), e.value, c.value, it.value


def track_stats(oracle_obj) -> TrackStats:
    st = TrackStats()
    rlib().mfo_get_track_stats(oracle_obj.h, C.byref(st))
    return st


def _pyr3(lst, dtype, keep):
    a = (C.c_void_p * 3)()
    for i, m in enumerate(lst):
        m = np.ascontiguousarray(m, dtype)
        keep.append(m)
        a[i] = m.ctypes.data
    return a


def track_rgbd(curr_v, curr_n, prev_v, prev_n, last_depth, next_depth, last_image, next_image, last_next2, W, H, fx, fy, cx, cy,
               R, t, opts):
    """mfo_track_rgbd.  Pyramids are 3-element lists (level 0..2).  Returns (R, t, inc 4x4, TrackStats)."""
    keep = []
    ptrs = lambda lst: C.cast(_pyr3(lst, np.float32, keep), C.POINTER(C.c_void_p))
    inp = RgbdInputs()
    inp.lastDepth = _pyr3(last_depth, np.float32, keep)
    inp.nextDepth = _pyr3(next_depth, np.float32, keep)
    inp.lastImage = _pyr3(last_image, np.uint8, keep)
    inp.nextImage = _pyr3(next_image, np.uint8, keep)
    ln2 = np.ascontiguousarray(last_next2, np.uint8)
    inp.lastNextImage2 = ln2.ctypes.data
    Rf = np.ascontiguousarray(R, np.float32).reshape(9).copy()
    tf = np.ascontiguousarray(t, np.float32).copy()
    inc = np.zeros(16, np.float32)
    st = TrackStats()
    rlib().mfo_track_rgbd(ptrs(curr_v), ptrs(curr_n), ptrs(prev_v), ptrs(prev_n), C.byref(inp), W, H, fx, fy, cx, cy,
                          C.byref(opts), Rf, tf, inc, C.byref(st))
    return Rf.reshape(3, 3), tf, mfo.from_pose16(inc), st


def u8_pyramid(img0):
    out = [np.ascontiguousarray(img0, np.uint8)]
    for _ in range(2):
        out.append(mfo.pyrdown_u8(out[-1]))
    return out


def f32_pyramid(d0):
    out = [np.ascontiguousarray(d0, np.float32)]
    for _ in range(2):
        out.append(mfo.pyrdown_f(out[-1]))
    return out
