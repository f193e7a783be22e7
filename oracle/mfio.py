"""ctypes binding of oracle/_ref/libmf_io.so: the reference's own log-reader text (oracle/build_io.py).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build_io

_lib = None


def available() -> bool:
    return build_io.reference_available() or os.path.exists(build_io.LIB)


def lib():
    global _lib
    if _lib is None:
        path = build_io.build()
        if path is None:
            raise RuntimeError("oracle/_ref/libmf_io.so is absent and /root/reference is not there to build it")
        L = C.CDLL(path)
        L.mfio_klg_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]; L.mfio_klg_open.restype = C.c_void_p
        L.mfio_klg_close.argtypes = [C.c_void_p]
        L.mfio_klg_num_frames.argtypes = [C.c_void_p]; L.mfio_klg_num_frames.restype = C.c_int
        L.mfio_klg_has_more.argtypes = [C.c_void_p]; L.mfio_klg_has_more.restype = C.c_int
        L.mfio_klg_next.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]; L.mfio_klg_next.restype = C.c_int
        L.mfio_load_mask_ids.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.mfio_load_mask_ids.restype = C.c_int
        _lib = L
    return _lib


def read_klg(path: str, W: int, H: int, flip: bool = False):
    """every frame MainController::run would process (`if (hasMore()) getNext()`: the last frame is never delivered) ->
    (numFrames, [(timestamp, depth float32 HxW, rgb uint8 HxWx3)])"""
    L = lib()
    h = L.mfio_klg_open(path.encode(), W, H, int(flip))
    if not h:
        raise ValueError("the reference's KlgLogReader could not open " + path)
    try:
        n = L.mfio_klg_num_frames(h)
        out = []
        while L.mfio_klg_has_more(h):
            ts = C.c_int64(0)
            depth = np.zeros((H, W), np.float32)
            rgb = np.zeros((H, W, 3), np.uint8)
            if L.mfio_klg_next(h, C.byref(ts), depth.ctypes.data, rgb.ctypes.data) != 0:
                raise ValueError("the reference's KlgLogReader failed on a frame")
            out.append((ts.value, depth, rgb))
        return n, out
    finally:
        L.mfio_klg_close(h)


def load_mask_ids(path: str):
    """ImageLogReader::loadMaskIDs -> (class ids incl. the leading 0, rois as (x, y, width, height)); raises like the reference"""
    ids = np.zeros(256, np.int32)
    rois = np.zeros((256, 4), np.int32)
    n_ids, n_rois = C.c_int(0), C.c_int(0)
    rc = lib().mfio_load_mask_ids(path.encode(), ids.ctypes.data, 256, C.byref(n_ids), rois.ctypes.data, 256, C.byref(n_rois))
    if rc != 0:
        raise ValueError("the reference's loadMaskIDs threw")
    return ids[:n_ids.value].tolist(), [tuple(int(v) for v in r) for r in rois[:n_rois.value]]
