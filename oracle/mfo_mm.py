"""ctypes binding of the oracle's multi-model path (GlobalProjection, MfSegmentation, object-model life cycle).

TEST INFRASTRUCTURE ONLY (see oracle/mfo.py).
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from .mfo import lib, Cam, Config, f32p, u8p, i32p, pose16, from_pose16


class SegParams(C.Structure):
    _fields_ = [("threshold", C.c_float), ("weightDistance", C.c_float), ("weightConvexity", C.c_float),
                ("morphEdgeIterations", C.c_int), ("morphEdgeRadius", C.c_int), ("morphMaskIterations", C.c_int),
                ("morphMaskRadius", C.c_int), ("removeEdges", C.c_int), ("minRelSizeNew", C.c_float),
                ("maxRelSizeNew", C.c_float), ("personClassID", C.c_int)]


class MMConfig(C.Structure):
    _fields_ = [("base", Config), ("confObject", C.c_float), ("capacityObject", C.c_int), ("trackAllModels", C.c_int),
                ("modelSpawnOffset", C.c_int), ("maxModels", C.c_int), ("seg", SegParams)]


class ModelView(C.Structure):
    _fields_ = [("surfels", C.POINTER(C.c_float)), ("count", C.c_int), ("pose16", C.POINTER(C.c_float)), ("id", C.c_int)]


_ready = False


def mm_lib():
    global _ready
    L = lib()
    if _ready:
        return L
    L.mfo_geometric_edge_map.argtypes = [f32p, f32p, f32p, C.c_int, C.c_int, C.c_float, C.c_float]
    L.mfo_threshold_map.argtypes = [f32p, u8p, C.c_int, C.c_float]
    L.mfo_invert_map.argtypes = [u8p, u8p, C.c_int]
    L.mfo_morph_closing_u8.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.mfo_global_projection.argtypes = [C.POINTER(Cam), C.POINTER(ModelView), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_float, u8p]
    L.mfo_connected_components4.argtypes = [u8p, i32p, i32p, C.c_int, C.c_int, C.c_int]
    L.mfo_connected_components4.restype = C.c_int
    L.mfo_default_seg_params.argtypes = [C.POINTER(SegParams)]
    L.mfo_mf_segmentation_cpu.argtypes = [C.POINTER(SegParams), C.c_int, C.c_int, u8p, f32p, u8p, i32p, C.c_int, u8p,
                                          i32p, i32p, C.c_int, C.c_int, C.c_int, u8p, u8p, C.POINTER(C.c_int),
                                          C.POINTER(C.c_int)]
    L.mfo_mm_default_config.argtypes = [C.POINTER(MMConfig), C.c_int, C.c_int] + [C.c_float] * 4
    L.mfo_mm_create.argtypes = [C.POINTER(MMConfig)]
    L.mfo_mm_create.restype = C.c_void_p
    L.mfo_mm_destroy.argtypes = [C.c_void_p]
    L.mfo_mm_process_frame.argtypes = [C.c_void_p, u8p, f32p, C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    L.mfo_mm_override_filtered_depth.argtypes = [C.c_void_p, f32p]
    L.mfo_mm_set_frame_to_frame_rgb.argtypes = [C.c_void_p, C.c_int]
    L.mfo_mm_set_bbox_limit.argtypes = [C.c_void_p, C.c_int]
    L.mfo_mm_set_trackable_class_ids.argtypes = [C.c_void_p, i32p, C.c_int]
    L.mfo_mm_make_nonstatic.argtypes = [C.c_void_p, C.c_int]
    L.mfo_mm_make_static.argtypes = [C.c_void_p, C.c_int]
    L.mfo_mm_is_nonstatic.argtypes = [C.c_void_p, C.c_int]
    L.mfo_mm_is_nonstatic.restype = C.c_int
    L.mfo_mm_model_class.argtypes = [C.c_void_p, C.c_int]
    L.mfo_mm_model_class.restype = C.c_int
    L.mfo_mm_force_tracking.argtypes = [C.c_void_p, i32p, f32p, C.c_int]
    L.mfo_mm_model_tracked_pose.argtypes = [C.c_void_p, C.c_int, f32p, C.POINTER(C.c_int)]
    L.mfo_mm_dbg_map.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mfo_mm_dbg_map.restype = C.POINTER(C.c_float)
    L.mfo_mm_dbg_pred.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mfo_mm_dbg_pred.restype = C.POINTER(C.c_float)
    L.mfo_mm_model_track_log.argtypes = [C.c_void_p, C.c_int, f32p]
    L.mfo_mm_model_track_log.restype = C.c_int
    L.mfo_mm_model_tracked_pose_alt.argtypes = [C.c_void_p, C.c_int, f32p]
    L.mfo_mm_set_probe_poses.argtypes = [C.c_void_p, i32p, f32p, C.c_int]
    L.mfo_mm_model_probe_log.argtypes = [C.c_void_p, C.c_int, f32p]
    L.mfo_mm_model_probe_log.restype = C.c_int
    L.mfo_mm_upload_map.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int]
    L.mfo_mm_upload_map.restype = C.c_int
    L.mfo_mm_num_models.argtypes = [C.c_void_p]
    L.mfo_mm_num_models.restype = C.c_int
    for n in ("mfo_mm_model_id", "mfo_mm_model_count"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int]
        getattr(L, n).restype = C.c_int
    L.mfo_mm_model_pose.argtypes = [C.c_void_p, C.c_int, f32p]
    L.mfo_mm_model_surfels.argtypes = [C.c_void_p, C.c_int]
    L.mfo_mm_model_surfels.restype = C.POINTER(C.c_float)
    for n, t in (("mfo_mm_segmentation", C.c_uint8), ("mfo_mm_projected_ids", C.c_uint8), ("mfo_mm_edge_map", C.c_float)):
        getattr(L, n).argtypes = [C.c_void_p]
        getattr(L, n).restype = C.POINTER(t)
    _ready = True
    return L


def default_seg_params(**kw) -> SegParams:
    p = SegParams()
    mm_lib().mfo_default_seg_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def geometric_edge_map(vmap, nmap, wD, wC):
    _, H, W = vmap.shape
    out = np.zeros((H, W), np.float32)
    mm_lib().mfo_geometric_edge_map(np.ascontiguousarray(vmap, np.float32), np.ascontiguousarray(nmap, np.float32), out,
                                    W, H, wD, wC)
    return out


def edge_binary(edge, threshold, radius, iterations):
    """threshold -> closing -> invert (MfSegmentation.cpp:205-207); returns (binaryEdge, inverted)."""
    H, W = edge.shape
    L = mm_lib()
    b = np.zeros((H, W), np.uint8)
    buf = np.zeros((H, W), np.uint8)
    L.mfo_threshold_map(np.ascontiguousarray(edge, np.float32), b, W * H, threshold)
    L.mfo_morph_closing_u8(b, buf, W, H, radius, iterations)
    inv = np.zeros((H, W), np.uint8)
    L.mfo_invert_map(b, inv, W * H)
    return b, inv


def connected_components4(binary):
    H, W = binary.shape
    labels = np.zeros((H, W), np.int32)
    maxc = W * H // 2 + 2
    stats = np.zeros((maxc, 5), np.int32)
    n = mm_lib().mfo_connected_components4(np.ascontiguousarray(binary, np.uint8), labels, stats.reshape(-1), maxc, W, H)
    return n, labels, stats[:n]


def global_projection(camera: Cam, models, time, time_delta, depth_cutoff):
    """models: list of (surfels (n,12) float32, pose 4x4, id) in model-list order."""
    L = mm_lib()
    views = (ModelView * len(models))()
    keep = []
    for i, (s, T, mid) in enumerate(models):
        s = np.ascontiguousarray(s, np.float32)
        p = pose16(T)
        keep += [s, p]
        views[i].surfels = s.ctypes.data_as(C.POINTER(C.c_float))
        views[i].count = len(s)
        views[i].pose16 = p.ctypes.data_as(C.POINTER(C.c_float))
        views[i].id = mid
    ids = np.zeros((camera.H, camera.W), np.uint8)
    L.mfo_global_projection(C.byref(camera), views, len(models), time, time, time_delta, depth_cutoff, ids)
    return ids


def mf_segmentation_cpu(W, H, binary, depth, mask, class_ids, projected_ids, model_ids, model_class_ids, next_id,
                        allow_new, ignore_map=None, params=None):
    L = mm_lib()
    prm = params if params is not None else default_seg_params()
    if ignore_map is None:
        ignore_map = np.zeros((H, W), np.uint8)
    full = np.zeros((H, W), np.uint8)
    has_new, new_cls = C.c_int(0), C.c_int(-1)
    cid = np.ascontiguousarray(class_ids if len(class_ids) else [0], np.int32)
    L.mfo_mf_segmentation_cpu(C.byref(prm), W, H, np.ascontiguousarray(binary, np.uint8),
                              np.ascontiguousarray(depth, np.float32), np.ascontiguousarray(mask, np.uint8), cid,
                              len(class_ids), np.ascontiguousarray(projected_ids, np.uint8),
                              np.ascontiguousarray(model_ids, np.int32), np.ascontiguousarray(model_class_ids, np.int32),
                              len(model_ids), next_id, int(allow_new), ignore_map, full, C.byref(has_new), C.byref(new_cls))
    return full, bool(has_new.value), new_cls.value


class OracleMM:
    """Multi-model pipeline (MaskFusion::processFrame with enableMultipleModels)."""

    def __init__(self, W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, seg=None, **kw):
        L = mm_lib()
        self.cfg = MMConfig()
        L.mfo_mm_default_config(C.byref(self.cfg), W, H, fx, fy, cx, cy)
        for k, v in kw.items():
            if k in ("confObject", "capacityObject", "trackAllModels", "modelSpawnOffset", "maxModels"):
                setattr(self.cfg, k, v)
            elif hasattr(self.cfg.base, k):
                setattr(self.cfg.base, k, v)
            else:
                raise AttributeError(k)
        for k, v in (seg or {}).items():
            setattr(self.cfg.seg, k, v)
        self.h = L.mfo_mm_create(C.byref(self.cfg))
        self.W, self.H = W, H

    def close(self):
        if self.h:
            mm_lib().mfo_mm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, rgb, depth, mask=None, class_ids=(), weight_multiplier=1.0, depth_filtered=None):
        """depth_filtered: test isolation -- taken as the bilateral filter's output instead of running the filter (so that a pipeline-level
        comparison is not perturbed by the last bits of two exp() implementations; the filter is compared on its own)"""
        if depth_filtered is not None:
            self._dF = np.ascontiguousarray(depth_filtered, np.float32)
            mm_lib().mfo_mm_override_filtered_depth(self.h, self._dF)
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        cid = np.ascontiguousarray(class_ids, np.int32) if len(class_ids) else None
        self._keep = (m, cid)
        return mm_lib().mfo_mm_process_frame(self.h, np.ascontiguousarray(rgb, np.uint8),
                                             np.ascontiguousarray(depth, np.float32),
                                             m.ctypes.data if m is not None else None,
                                             cid.ctypes.data if cid is not None else None, len(class_ids),
                                             weight_multiplier)

    def set_trackable_class_ids(self, ids):
        a = np.ascontiguousarray(list(ids) if len(ids) else [0], np.int32)
        mm_lib().mfo_mm_set_trackable_class_ids(self.h, a, len(ids))

    def make_nonstatic(self, i):
        mm_lib().mfo_mm_make_nonstatic(self.h, i)

    def make_static(self, i):
        mm_lib().mfo_mm_make_static(self.h, i)

    def is_nonstatic(self, i):
        return bool(mm_lib().mfo_mm_is_nonstatic(self.h, i))

    def model_class(self, i):
        return mm_lib().mfo_mm_model_class(self.h, i)

    def force_tracking(self, ids, poses):
        """teacher forcing: the next frame continues with these poses (4x4 each, by model id) after its own tracking steps, and its
        drop decisions follow the list (see mf_oracle.c)"""
        a = np.ascontiguousarray(list(ids), np.int32)
        p = np.ascontiguousarray(np.stack([pose16(T) for T in poses]), np.float32) if len(ids) else np.zeros((1, 16), np.float32)
        mm_lib().mfo_mm_force_tracking(self.h, a if len(ids) else np.zeros(1, np.int32), p.reshape(-1), len(ids))

    def model_tracked_pose(self, i):
        """(pose this side's own tracking step returned in the last frame, whether the model was tracked at all)"""
        p = np.zeros(16, np.float32)
        t = C.c_int(0)
        mm_lib().mfo_mm_model_tracked_pose(self.h, i, p, C.byref(t))
        return from_pose16(p), bool(t.value)

    def dbg_map(self, what, level):
        """'vmap_g' / 'nmap_g' (model-side pyramid of the model tracked last) or 'vmap' / 'nmap' (current frame), planar (3, h, w)"""
        w = ("vmap_g", "nmap_g", "vmap", "nmap").index(what)
        h_, w_ = self.H >> level, self.W >> level
        return np.ctypeslib.as_array(mm_lib().mfo_mm_dbg_map(self.h, w, level), shape=(3, h_, w_)).copy()

    def dbg_pred(self, model, what):
        return np.ctypeslib.as_array(mm_lib().mfo_mm_dbg_pred(self.h, model, ("vertex", "normal").index(what)), shape=(self.H, self.W, 4)).copy()

    def model_track_log(self, i):
        """reduced geometric systems of model i's last tracking step: (iterations, 32) in the device log's layout"""
        out = np.zeros((20, 32), np.float32)
        n = mm_lib().mfo_mm_model_track_log(self.h, i, out.reshape(-1))
        return out[:n]

    def set_probe_poses(self, ids, poses):
        """per-iteration teacher forcing: poses[k] = (20, 12) float32, the (Rcurr row-major, tcurr) another implementation used in each
        Gauss-Newton iteration of model ids[k]; the next frame evaluates this side's systems there (model_probe_log)"""
        a = np.ascontiguousarray(list(ids) if len(ids) else [0], np.int32)
        p = np.ascontiguousarray(np.stack(poses) if len(ids) else np.zeros((1, 20, 12)), np.float32)
        mm_lib().mfo_mm_set_probe_poses(self.h, a, p.reshape(-1), len(ids))

    def model_probe_log(self, i):
        out = np.zeros((20, 32), np.float32)
        n = mm_lib().mfo_mm_model_probe_log(self.h, i, out.reshape(-1))
        return out[:n]

    def model_step_sensitivity(self, i):
        """max |entry| by which this side's own tracking step of the last (teacher-forced) frame moves when its start pose is shifted by
        one micrometre: the conditioning of that step, measured"""
        p = np.zeros(16, np.float32)
        mm_lib().mfo_mm_model_tracked_pose_alt(self.h, i, p)
        own, _ = self.model_tracked_pose(i)
        return float(np.abs(from_pose16(p) - own).max())

    def upload_map(self, i, surfels):
        """replace model i's surfel buffer with `surfels` ((n, 12) float32): the twin of Model.uploadMap on the product's side"""
        s = np.ascontiguousarray(surfels, np.float32).reshape(-1, 12)
        if mm_lib().mfo_mm_upload_map(self.h, i, s.reshape(-1), len(s)) != 0:
            raise ValueError("upload_map: no such model, or more records than its capacity")

    @property
    def n_models(self):
        return mm_lib().mfo_mm_num_models(self.h)

    def model_id(self, i):
        return mm_lib().mfo_mm_model_id(self.h, i)

    def model_count(self, i):
        return mm_lib().mfo_mm_model_count(self.h, i)

    def model_pose(self, i):
        p = np.zeros(16, np.float32)
        mm_lib().mfo_mm_model_pose(self.h, i, p)
        return from_pose16(p)

    def model_surfels(self, i):
        n = self.model_count(i)
        ptr = mm_lib().mfo_mm_model_surfels(self.h, i)
        return np.ctypeslib.as_array(ptr, shape=(n, 12)).copy() if n else np.zeros((0, 12), np.float32)

    def segmentation(self):
        return np.ctypeslib.as_array(mm_lib().mfo_mm_segmentation(self.h), shape=(self.H, self.W)).copy()

    def projected_ids(self):
        return np.ctypeslib.as_array(mm_lib().mfo_mm_projected_ids(self.h), shape=(self.H, self.W)).copy()

    def edge_map(self):
        return np.ctypeslib.as_array(mm_lib().mfo_mm_edge_map(self.h), shape=(self.H, self.W)).copy()
