"""ctypes binding of oracle/_ref/libmf_glsl.so: the reference's own GLSL shaders (Core/Shaders/*.vert / *.frag) compiled as C++ by
oracle/build_glsl.py and run by oracle/glsl_shim/mfgl_api.cpp.  TEST INFRASTRUCTURE ONLY: pins oracle/mf_oracle.c's restatement of the
OpenGL half of the hot path (tests/test_glsl_pin.py) and generates tests/golden/glsl_vectors.npz.

Array conventions follow oracle/mfo.py (surfels: (n, 12) float32; poses: 4x4 float arrays, handed over column-major)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build_glsl

_lib = None
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
TEXDIM = 1024     # side of the "update map" (upstream: TEXTURE_DIMENSION_MAX = 3072; any value >= sqrt(surfel count) behaves the same)


def available() -> bool:
    return build_glsl.reference_available() or os.path.exists(build_glsl.LIB)


def lib():
    global _lib
    if _lib is None:
        path = build_glsl.build()
        if path is None:
            raise RuntimeError("oracle/_ref/libmf_glsl.so is absent and /root/reference is not here to build it from")
        L = C.CDLL(path)
        i, f = C.c_int, C.c_float
        L.mfglsl_bilateral.argtypes = [f32p, f32p, i, i, f]
        L.mfglsl_init_surfels.argtypes = [u8p, f32p, f32p, i, i, f, f, f, f, i, f, f32p, i, C.POINTER(C.c_int)]
        L.mfglsl_init_surfels.restype = i
        L.mfglsl_predict_indices.argtypes = [f32p, f32p, i, i, f, i, i, i, f, f, f, f, i32p, f32p, f32p, f32p]
        L.mfglsl_fuse_data.argtypes = [f32p, u8p, f32p, f32p, u8p, i, i, f, f, i, i, f, f, f, f, i32p, f32p, f32p, f32p, i, u8p, i32p, f32p]
        L.mfglsl_fuse_update.argtypes = [f32p, f32p, i, i, i, u8p, i32p, f32p, i]
        L.mfglsl_clean.argtypes = [f32p, f32p, i, u8p, f32p, i, i, i, f, f, f, i, i, i, f, f, f, f, i32p, f32p, f32p, f32p, f32p, u8p, f32p, i, u8p]
        L.mfglsl_clean.restype = i
        L.mfglsl_combined_predict.argtypes = [f32p, f32p, i, f, f, i, i, i, i, i, f, f, f, f, u8p, f32p, f32p, u16p]
        L.mfglsl_global_projection.argtypes = [C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               i, i, i, i, f, i, i, f, f, f, f, u8p]
        L.mfglsl_fill_in.argtypes = [u8p, f32p, f32p, u8p, f32p, i, i, i, f, f, f, f, u8p, f32p, f32p]
        _lib = L
    return _lib


def _p16(pose):
    return np.ascontiguousarray(np.asarray(pose, np.float32).T.reshape(16))


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def bilateral(depth, maxD=20.0):
    H, W = depth.shape
    out = np.zeros((H, W), np.float32)
    lib().mfglsl_bilateral(_f(depth), out, W, H, maxD)
    return out


def init_surfels(rgb, depth, depthF, K, time, maxDepth, capacity=None):
    H, W = depth.shape
    cap = capacity or W * H
    out = np.zeros((cap, 12), np.float32)
    nf = C.c_int(0)
    n = lib().mfglsl_init_surfels(np.ascontiguousarray(rgb, np.uint8), _f(depth), _f(depthF), W, H, *K, time, maxDepth, out, cap, C.byref(nf))
    return out[:n], nf.value


def predict_indices(pose, surfels, time, maxDepth, timeDelta, W, H, K):
    P = W * H
    index = np.zeros(P, np.int32)
    vc, ct, nr = (np.zeros((P, 4), np.float32) for _ in range(3))
    s = _f(surfels).reshape(-1, 12)
    lib().mfglsl_predict_indices(_p16(pose), s, len(s), time, maxDepth, timeDelta, W, H, *K, index, vc, ct, nr)
    return index.reshape(H, W), vc.reshape(H, W, 4), ct.reshape(H, W, 4), nr.reshape(H, W, 4)


def fuse_data(pose, rgb, depth, depthF, mask, maskID, time, weighting, maxDepth, K, index, vc, ct, nr):
    """per pixel in the uv buffer's column-major order: op (0 / 1 merge / 2 new), best surfel index, emitted record"""
    H, W = depth.shape
    P = W * H
    op, best, rec = np.zeros(P, np.uint8), np.zeros(P, np.int32), np.zeros((P, 12), np.float32)
    lib().mfglsl_fuse_data(_p16(pose), np.ascontiguousarray(rgb, np.uint8), _f(depth), _f(depthF), np.ascontiguousarray(mask, np.uint8), maskID, time,
                           weighting, maxDepth, W, H, *K, np.ascontiguousarray(index, np.int32).reshape(-1), _f(vc).reshape(-1, 4),
                           _f(ct).reshape(-1, 4), _f(nr).reshape(-1, 4), TEXDIM, op, best, rec)
    return op, best, rec


def fuse_update(surfels, time, op, best, rec):
    s = _f(surfels).reshape(-1, 12)
    out = np.zeros_like(s)
    lib().mfglsl_fuse_update(s, out, len(s), time, TEXDIM, op, best, rec, len(op))
    return out


def clean(pose, surfels, op, rec, time, timeDelta, confThreshold, maxDepth, outlierCoeff, maskID, K, index, vc, ct, nr, depthF, mask):
    H, W = depthF.shape
    s = _f(surfels).reshape(-1, 12)
    cap = len(s) + int((op > 0).sum())
    out = np.zeros((max(cap, 1), 12), np.float32)
    keep = np.zeros(max(cap, 1), np.uint8)
    n = lib().mfglsl_clean(_p16(pose), s, len(s), op, rec, len(op), time, timeDelta, confThreshold, maxDepth, outlierCoeff, maskID, W, H, *K,
                           np.ascontiguousarray(index, np.int32).reshape(-1), _f(vc).reshape(-1, 4), _f(ct).reshape(-1, 4), _f(nr).reshape(-1, 4),
                           _f(depthF), np.ascontiguousarray(mask, np.uint8), out, cap, keep)
    return out[:n], keep[:cap]


def combined_predict(pose, surfels, maxDepth, confThreshold, time, maxTime, timeDelta, W, H, K):
    P = W * H
    img = np.zeros((P, 4), np.uint8)
    vc, nr = np.zeros((P, 4), np.float32), np.zeros((P, 4), np.float32)
    tm = np.zeros(P, np.uint16)
    s = _f(surfels).reshape(-1, 12)
    lib().mfglsl_combined_predict(_p16(pose), s, len(s), maxDepth, confThreshold, time, maxTime, timeDelta, W, H, *K, img, vc, nr, tm)
    return img.reshape(H, W, 4), vc.reshape(H, W, 4), nr.reshape(H, W, 4), tm.reshape(H, W)


def fill_in(predImage, predVertex, predNormal, rawRgb, rawDepth, passthrough, K):
    H, W = rawDepth.shape
    fi = np.zeros((H, W, 4), np.uint8)
    fv, fn = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    lib().mfglsl_fill_in(np.ascontiguousarray(predImage, np.uint8), _f(predVertex), _f(predNormal), np.ascontiguousarray(rawRgb, np.uint8), _f(rawDepth),
                         int(passthrough), W, H, *K, fi, fv, fn)
    return fi, fv, fn


def global_projection(models, time, timeDelta, depthCutoff, W, H, K):
    """models: list of (surfels (n,12), pose 4x4, id) in model-list order -> id image (H, W) uint8"""
    n = len(models)
    keep = []
    poses = (C.POINTER(C.c_float) * n)()
    bufs = (C.POINTER(C.c_float) * n)()
    counts, ids = (C.c_int * n)(), (C.c_int * n)()
    for k, (s, T, mid) in enumerate(models):
        s = _f(s).reshape(-1, 12)
        p = _p16(T)
        keep += [s, p]
        poses[k] = p.ctypes.data_as(C.POINTER(C.c_float))
        bufs[k] = s.ctypes.data_as(C.POINTER(C.c_float))
        counts[k], ids[k] = len(s), mid
    out = np.zeros(W * H, np.uint8)
    lib().mfglsl_global_projection(poses, bufs, counts, ids, n, time, time, timeDelta, depthCutoff, W, H, *K, out)
    return out.reshape(H, W)
