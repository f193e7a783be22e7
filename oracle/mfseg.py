"""ctypes binding of oracle/_ref/libmf_seg.so: the host half of the reference's MfSegmentation::performSegmentation compiled from the
reference's own text (oracle/build_seg.py).  TEST INFRASTRUCTURE ONLY; same call shape as mfo_mm.mf_segmentation_cpu."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build_seg

_lib = None


def available() -> bool:
    return build_seg.reference_available() or os.path.exists(build_seg.LIB)


def lib():
    global _lib
    if _lib is None:
        path = build_seg.build()
        if path is None:
            raise RuntimeError("oracle/_ref/libmf_seg.so is absent and /root/reference is not here to build it from")
        _lib = C.CDLL(path)
    return _lib


def mf_segmentation(W, H, binary, depth, mask, class_ids, projected_ids, model_ids, model_class_ids, next_id, allow_new, ignore_map, prm):
    """-> (full segmentation (H, W) uint8, hasNewLabel, newClassID); ignore_map is updated in place.  prm: mfo_mm seg params struct"""
    p = np.array([prm.threshold, prm.weightDistance, prm.weightConvexity, prm.morphEdgeIterations, prm.morphEdgeRadius, prm.morphMaskIterations,
                  prm.morphMaskRadius, prm.removeEdges, prm.minRelSizeNew, prm.maxRelSizeNew, prm.personClassID], np.float32)
    full = np.zeros((H, W), np.uint8)
    has_new, new_cls = C.c_int(0), C.c_int(-1)
    g = lambda a, t: np.ascontiguousarray(a, t)
    binary, depth, mask, proj = g(binary, np.uint8), g(depth, np.float32), g(mask, np.uint8), g(projected_ids, np.uint8)
    cid = g(class_ids if len(class_ids) else [0], np.int32)
    mids, mcls = g(model_ids, np.int32), g(model_class_ids, np.int32)
    lib().mfseg_labels(W, H, binary.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                       cid.ctypes.data_as(C.c_void_p), len(class_ids), proj.ctypes.data_as(C.c_void_p), mids.ctypes.data_as(C.c_void_p),
                       mcls.ctypes.data_as(C.c_void_p), len(model_ids), int(next_id), int(allow_new), p.ctypes.data_as(C.c_void_p),
                       ignore_map.ctypes.data_as(C.c_void_p), full.ctypes.data_as(C.c_void_p), C.byref(has_new), C.byref(new_cls))
    return full, bool(has_new.value), new_cls.value
