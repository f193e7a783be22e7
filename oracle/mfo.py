"""ctypes binding of the CPU oracle (oracle/libmf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (maskfusion_amd/) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmf_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mf_oracle.c")
    hdr = os.path.join(_HERE, "mf_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmf_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_FAST_PATH = os.path.join(_HERE, "libmf_oracle_fast.so")


def use_fast_build() -> bool:
    """bench.py's cpu_baseline leg only: switch this process to a -O3 -march=native build of the same source (compiled HERE, for
    this machine's cores; the parity build stays -O2 -ffp-contract=off with no -march so that it is the same everywhere).
    Returns False -- and keeps the parity build -- if the compile fails.  Must be called before the library is first used."""
    global _LIB_PATH, _lib
    if _lib is not None:
        return _LIB_PATH == _FAST_PATH
    try:
        src = os.path.join(_HERE, "mf_oracle.c")
        tag = _FAST_PATH + ".host"
        host = open("/proc/cpuinfo").read().split("model name", 1)[-1].split("\n", 1)[0] if os.path.exists("/proc/cpuinfo") else "?"
        stale = (not os.path.exists(_FAST_PATH)) or os.path.getmtime(src) > os.path.getmtime(_FAST_PATH) or \
            (not os.path.exists(tag)) or open(tag).read() != host
        if stale:
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=gnu11", "-shared", "-o", _FAST_PATH, src, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            open(tag, "w").write(host)
        _LIB_PATH = _FAST_PATH
        return True
    except Exception:
        return False


f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class Cam(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float)]


class TrackOpts(C.Structure):
    _fields_ = [("pyramid", C.c_int), ("fastOdom", C.c_int), ("so3", C.c_int), ("rgbOnly", C.c_int),
                ("icpWeight", C.c_float), ("distThresh", C.c_float), ("angleThresh", C.c_float)]


class TrackLog(C.Structure):
    _fields_ = [("n_iters", C.c_int), ("A", C.c_float * 36 * 19), ("b", C.c_float * 6 * 19),
                ("residual", C.c_float * 2 * 19), ("x", C.c_double * 6 * 19)]


class Config(C.Structure):
    _fields_ = [("W", C.c_int), ("H", C.c_int), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("timeDelta", C.c_int), ("confGlobal", C.c_float), ("depthCutoff", C.c_float),
                ("icpWeight", C.c_float), ("maxDepthProcessed", C.c_float), ("outlierCoeff", C.c_float),
                ("fastOdom", C.c_int), ("pyramid", C.c_int), ("so3", C.c_int), ("capacity", C.c_int),
                ("rgbOnly", C.c_int)]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if _LIB_PATH != _FAST_PATH:
        build()
    L = C.CDLL(_LIB_PATH)
    L.mfo_bilateral.argtypes = [f32p, f32p, C.c_int, C.c_int]
    L.mfo_pyrdown_gauss_f.argtypes = [f32p, f32p, C.c_int, C.c_int]
    L.mfo_pyrdown_gauss_u8.argtypes = [u8p, u8p, C.c_int, C.c_int]
    L.mfo_create_vmap.argtypes = [f32p, f32p, C.c_int, C.c_int] + [C.c_float] * 5
    L.mfo_create_nmap.argtypes = [f32p, f32p, C.c_int, C.c_int]
    L.mfo_copy_maps.argtypes = [f32p, f32p, f32p, f32p, C.c_int, C.c_int]
    L.mfo_resize_map.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int]
    L.mfo_transform_maps.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int]
    L.mfo_icp_step.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p] + [C.c_float] * 4 + [f32p, f32p] + \
        [C.c_float] * 2 + [C.c_int] * 2 + [f32p, f32p, f32p]
    L.mfo_ldlt_solve.argtypes = [f64p, f64p, f64p, C.c_int]
    L.mfo_ldlt_solve.restype = C.c_int
    L.mfo_rodrigues.argtypes = [f64p, f64p]
    L.mfo_update_se3.argtypes = [f64p, f64p]
    L.mfo_track_icp.argtypes = [C.POINTER(C.c_void_p)] * 4 + [C.c_int] * 2 + [C.c_float] * 4 + \
        [C.POINTER(TrackOpts), f32p, f32p, f32p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(TrackLog)]
    L.mfo_encode_color.argtypes = [C.c_float] * 3
    L.mfo_encode_color.restype = C.c_float
    L.mfo_decode_color.argtypes = [C.c_float, f32p]
    L.mfo_get_radius.argtypes = [C.c_float] * 4
    L.mfo_get_radius.restype = C.c_float
    L.mfo_confidence.argtypes = [C.c_float] * 5
    L.mfo_confidence.restype = C.c_float
    L.mfo_init_surfels.argtypes = [C.POINTER(Cam), u8p, f32p, f32p, C.c_int, C.c_float, f32p, C.c_int]
    L.mfo_init_surfels.restype = C.c_int
    L.mfo_predict_indices.argtypes = [C.POINTER(Cam), f32p, f32p, C.c_int, C.c_int, C.c_float, C.c_int, i32p, f32p,
                                      f32p, f32p]
    L.mfo_fuse_data.argtypes = [C.POINTER(Cam), f32p, u8p, f32p, f32p, u8p, C.c_int, C.c_int, C.c_float, C.c_float,
                                i32p, f32p, f32p, u8p, i32p, f32p, C.POINTER(C.c_int)]
    L.mfo_fuse_update.argtypes = [f32p, f32p, C.c_int, C.c_int, u8p, i32p, f32p, C.c_int]
    L.mfo_clean.argtypes = [C.POINTER(Cam), f32p, f32p, C.c_int, u8p, f32p, C.c_int, C.c_int, C.c_int, C.c_float,
                            C.c_float, C.c_float, C.c_int, i32p, f32p, f32p, f32p, f32p, u8p, f32p, C.c_int]
    L.mfo_clean.restype = C.c_int
    L.mfo_combined_predict.argtypes = [C.POINTER(Cam), f32p, f32p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                       C.c_int, u8p, f32p, f32p, u16p]
    L.mfo_fill_in.argtypes = [C.POINTER(Cam), u8p, f32p, f32p, u8p, f32p, C.c_int, u8p, f32p, f32p]
    L.mfo_requires_fill_in.argtypes = [u8p, C.c_int, C.c_int, C.c_float]
    L.mfo_requires_fill_in.restype = C.c_int
    L.mfo_fusion_weight.argtypes = [f32p, f32p, C.c_float]
    L.mfo_fusion_weight.restype = C.c_float
    L.mfo_default_config.argtypes = [C.POINTER(Config), C.c_int, C.c_int] + [C.c_float] * 4
    L.mfo_create.argtypes = [C.POINTER(Config)]
    L.mfo_create.restype = C.c_void_p
    L.mfo_destroy.argtypes = [C.c_void_p]
    L.mfo_process_frame.argtypes = [C.c_void_p, u8p, f32p, C.c_float]
    L.mfo_process_frame.restype = C.c_int
    L.mfo_process_frame_ex.argtypes = [C.c_void_p, u8p, f32p, C.c_float, f32p, C.c_int]
    L.mfo_process_frame_ex.restype = C.c_int
    L.mfo_override_filtered_depth.argtypes = [C.c_void_p, f32p]
    L.mfo_last_track_ill.argtypes = []
    L.mfo_set_frame_to_frame_rgb.argtypes = [C.c_void_p, C.c_int]
    L.mfo_last_track_ill.restype = C.c_int
    L.mfo_get_pose.argtypes = [C.c_void_p, f32p]
    L.mfo_get_count.argtypes = [C.c_void_p]
    L.mfo_get_count.restype = C.c_int
    L.mfo_get_tick.argtypes = [C.c_void_p]
    L.mfo_get_tick.restype = C.c_int
    L.mfo_get_surfels.argtypes = [C.c_void_p]
    L.mfo_get_surfels.restype = C.POINTER(C.c_float)
    L.mfo_get_icp_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.mfo_get_timings.argtypes = [C.c_void_p, f64p]
    for name, rt in (("mfo_dbg_depthF", C.c_float), ("mfo_dbg_pred_vertex", C.c_float),
                     ("mfo_dbg_pred_normal", C.c_float), ("mfo_dbg_pred_image", C.c_uint8)):
        getattr(L, name).argtypes = [C.c_void_p]
        getattr(L, name).restype = C.POINTER(rt)
    L.mfo_dbg_last_fillin.argtypes = [C.c_void_p]
    L.mfo_dbg_last_fillin.restype = C.c_int
    _lib = L
    return L


# ------------------------------------------------------------------------------------------------
# numpy-level helpers
# ------------------------------------------------------------------------------------------------
def cam(W, H, fx, fy, cx, cy) -> Cam:
    return Cam(W, H, fx, fy, cx, cy)


def pose16(T: np.ndarray) -> np.ndarray:
    """4x4 -> 16 floats column-major."""
    return np.ascontiguousarray(np.asarray(T, np.float32).T.reshape(16))


def from_pose16(p: np.ndarray) -> np.ndarray:
    return np.asarray(p, np.float64).reshape(4, 4).T.copy()


def bilateral(depth):
    H, W = depth.shape
    out = np.empty_like(depth)
    lib().mfo_bilateral(np.ascontiguousarray(depth, np.float32), out, W, H)
    return out


def pyrdown_f(src):
    H, W = src.shape
    dst = np.empty((H // 2, W // 2), np.float32)
    lib().mfo_pyrdown_gauss_f(np.ascontiguousarray(src, np.float32), dst, W, H)
    return dst


def pyrdown_u8(src):
    H, W = src.shape
    dst = np.empty((H // 2, W // 2), np.uint8)
    lib().mfo_pyrdown_gauss_u8(np.ascontiguousarray(src, np.uint8), dst, W, H)
    return dst


def create_vmap(depth, fx, fy, cx, cy, cutoff):
    H, W = depth.shape
    v = np.empty((3, H, W), np.float32)
    lib().mfo_create_vmap(np.ascontiguousarray(depth, np.float32), v, W, H, fx, fy, cx, cy, cutoff)
    return v


def create_nmap(vmap):
    _, H, W = vmap.shape
    n = np.empty((3, H, W), np.float32)
    lib().mfo_create_nmap(np.ascontiguousarray(vmap, np.float32), n, W, H)
    return n


def copy_maps(v4, n4):
    H, W, _ = v4.shape
    v = np.empty((3, H, W), np.float32)
    n = np.empty((3, H, W), np.float32)
    lib().mfo_copy_maps(np.ascontiguousarray(v4, np.float32), np.ascontiguousarray(n4, np.float32), v, n, W, H)
    return v, n


def resize_map(m, normalize):
    _, H, W = m.shape
    out = np.empty((3, H // 2, W // 2), np.float32)
    lib().mfo_resize_map(np.ascontiguousarray(m, np.float32), out, W, H, int(normalize))
    return out


def transform_maps(v, n, R, t):
    _, H, W = v.shape
    vo, no = np.empty_like(v), np.empty_like(n)
    lib().mfo_transform_maps(np.ascontiguousarray(v), np.ascontiguousarray(n),
                             np.ascontiguousarray(R, np.float32).reshape(9), np.ascontiguousarray(t, np.float32),
                             vo, no, W, H)
    return vo, no


def icp_step(Rcurr, tcurr, vc, nc, Rprev_inv, tprev, fx, fy, cx, cy, vp, npv, dist=0.10,
             angle=float(np.sin(np.float32(20.0 * 3.14159254 / 180.0)))):
    _, H, W = vc.shape
    A = np.zeros(36, np.float32)
    b = np.zeros(6, np.float32)
    res = np.zeros(2, np.float32)
    f = lambda a: np.ascontiguousarray(a, np.float32).reshape(-1)
    lib().mfo_icp_step(f(Rcurr), f(tcurr), f(vc), f(nc), f(Rprev_inv), f(tprev), fx, fy, cx, cy, f(vp), f(npv),
                       dist, angle, W, H, A, b, res)
    return A.reshape(6, 6), b, res


def default_track_opts(**kw) -> TrackOpts:
    o = TrackOpts(1, 0, 0, 0, 100.0, 0.10, float(np.sin(np.float32(20.0 * 3.14159254 / 180.0))))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def track_icp(curr_v, curr_n, prev_v, prev_n, W, H, fx, fy, cx, cy, R, t, opts=None, want_log=False):
    """R: 3x3, t: 3 (pose at entry).  Returns (R, t, inc 4x4, err, count, log)."""
    opts = opts or default_track_opts()
    keep = []

    def arr(lst):
        a = (C.c_void_p * 3)()
        for i, m in enumerate(lst):
            m = np.ascontiguousarray(m, np.float32)
            keep.append(m)
            a[i] = m.ctypes.data
        return C.cast(a, C.POINTER(C.c_void_p))

    Rf = np.ascontiguousarray(R, np.float32).reshape(9).copy()
    tf = np.ascontiguousarray(t, np.float32).copy()
    inc = np.zeros(16, np.float32)
    err, cnt = C.c_float(0), C.c_float(0)
    log = TrackLog() if want_log else None
    lib().mfo_track_icp(arr(curr_v), arr(curr_n), arr(prev_v), arr(prev_n), W, H, fx, fy, cx, cy, C.byref(opts), Rf,
                        tf, inc, C.byref(err), C.byref(cnt), C.byref(log) if log is not None else None)
    return Rf.reshape(3, 3), tf, from_pose16(inc), err.value, cnt.value, log


class Oracle:
    """Single-model pipeline (MaskFusion::processFrame, -static)."""

    def __init__(self, W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, **kw):
        self.cfg = Config()
        lib().mfo_default_config(C.byref(self.cfg), W, H, fx, fy, cx, cy)
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(k)
            setattr(self.cfg, k, v)
        self.h = lib().mfo_create(C.byref(self.cfg))
        self.W, self.H = W, H

    def close(self):
        if self.h:
            lib().mfo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def process_frame(self, rgb, depth, weight_multiplier=1.0, in_pose=None, bootstrap=False, depth_filtered=None):
        if depth_filtered is not None:   # test isolation: this frame's bilateral output is given (see mf_oracle.h)
            self._dF = np.ascontiguousarray(depth_filtered, np.float32)
            lib().mfo_override_filtered_depth(self.h, self._dF)
        if in_pose is None:
            return lib().mfo_process_frame(self.h, np.ascontiguousarray(rgb, np.uint8),
                                           np.ascontiguousarray(depth, np.float32), weight_multiplier)
        p = pose16(in_pose)
        return lib().mfo_process_frame_ex(self.h, np.ascontiguousarray(rgb, np.uint8), np.ascontiguousarray(depth, np.float32),
                                          weight_multiplier, p, int(bootstrap))

    @property
    def pose(self):
        p = np.zeros(16, np.float32)
        lib().mfo_get_pose(self.h, p)
        return from_pose16(p)

    @property
    def count(self):
        return lib().mfo_get_count(self.h)

    @property
    def tick(self):
        return lib().mfo_get_tick(self.h)

    def surfels(self):
        n = self.count
        ptr = lib().mfo_get_surfels(self.h)
        return np.ctypeslib.as_array(ptr, shape=(n, 12)).copy() if n else np.zeros((0, 12), np.float32)

    def icp_stats(self):
        e, c = C.c_float(0), C.c_float(0)
        lib().mfo_get_icp_stats(self.h, C.byref(e), C.byref(c))
        return e.value, c.value

    def timings(self):
        t = np.zeros(8, np.float64)
        lib().mfo_get_timings(self.h, t)
        names = ["preprocess", "odomInit", "odom", "indexMap", "fuseData", "fuseUpdate", "clean", "predict"]
        return dict(zip(names, t.tolist()))

    def dbg(self, what):
        P = self.W * self.H
        L = lib()
        if what == "depthF":
            return np.ctypeslib.as_array(L.mfo_dbg_depthF(self.h), shape=(self.H, self.W)).copy()
        if what == "pred_vertex":
            return np.ctypeslib.as_array(L.mfo_dbg_pred_vertex(self.h), shape=(self.H, self.W, 4)).copy()
        if what == "pred_normal":
            return np.ctypeslib.as_array(L.mfo_dbg_pred_normal(self.h), shape=(self.H, self.W, 4)).copy()
        if what == "pred_image":
            return np.ctypeslib.as_array(L.mfo_dbg_pred_image(self.h), shape=(self.H, self.W, 4)).copy()
        if what == "fillin":
            return L.mfo_dbg_last_fillin(self.h)
        raise KeyError(what)


# ------------------------------------------------------------------------------------------------
# surfel passes, one call each (for the per-pass differential tests)
# ------------------------------------------------------------------------------------------------
def predict_indices(camera: Cam, T, surfels, count, time, max_depth, time_delta):
    """-> index (H,W) int32, vertConf, colorTime, normRad (H,W,4) float32"""
    H, W = camera.H, camera.W
    idx = np.zeros((H, W), np.int32)
    vc, ct, nr = (np.zeros((H, W, 4), np.float32) for _ in range(3))
    lib().mfo_predict_indices(C.byref(camera), pose16(T), np.ascontiguousarray(surfels, np.float32).reshape(-1), count, time,
                              max_depth, time_delta, idx, vc.reshape(-1), ct.reshape(-1), nr.reshape(-1))
    return idx, vc, ct, nr


def n_candidates(W, H, time):
    par = time & 1
    return ((W - par + 1) // 2) * ((H - par + 1) // 2)


def fuse_data(camera: Cam, T, rgb, depth_raw, depth_f, mask, mask_id, time, weighting, max_depth, idx, vc, nr):
    """-> cand_op (n,), cand_best (n,), cand_rec (n,12); candidates in column-major order of the quarter-rate pixels"""
    maxc = ((camera.W + 1) // 2) * ((camera.H + 1) // 2)
    op = np.zeros(maxc, np.uint8)
    best = np.zeros(maxc, np.int32)
    rec = np.zeros((maxc, 12), np.float32)
    n = C.c_int(0)
    lib().mfo_fuse_data(C.byref(camera), pose16(T), np.ascontiguousarray(rgb, np.uint8).reshape(-1), np.ascontiguousarray(depth_raw, np.float32).reshape(-1),
                        np.ascontiguousarray(depth_f, np.float32).reshape(-1), np.ascontiguousarray(mask, np.uint8).reshape(-1), mask_id, time,
                        weighting, max_depth, np.ascontiguousarray(idx, np.int32).reshape(-1), np.ascontiguousarray(vc, np.float32).reshape(-1),
                        np.ascontiguousarray(nr, np.float32).reshape(-1), op, best, rec.reshape(-1), C.byref(n))
    return op[:n.value], best[:n.value], rec[:n.value]


def fuse_update(surfels, count, time, op, best, rec):
    src = np.ascontiguousarray(surfels, np.float32).reshape(-1)
    dst = np.zeros_like(src)
    lib().mfo_fuse_update(src, dst, count, time, np.ascontiguousarray(op), np.ascontiguousarray(best), np.ascontiguousarray(rec, np.float32).reshape(-1),
                          len(op))
    return dst.reshape(-1, 12)


def clean(camera: Cam, T, surfels, count, op, rec, time, time_delta, conf_threshold, max_depth, outlier_coeff, mask_id, idx, vc, ct, nr,
          depth_f, mask, capacity):
    dst = np.zeros((capacity, 12), np.float32)
    g = lambda a, t: np.ascontiguousarray(a, t).reshape(-1)
    n = lib().mfo_clean(C.byref(camera), pose16(T), g(surfels, np.float32), count, np.ascontiguousarray(op), g(rec, np.float32), len(op), time,
                        time_delta, conf_threshold, max_depth, outlier_coeff, mask_id, g(idx, np.int32), g(vc, np.float32), g(ct, np.float32),
                        g(nr, np.float32), g(depth_f, np.float32), g(mask, np.uint8), dst.reshape(-1), capacity)
    return dst[:n].copy(), n


def combined_predict(camera: Cam, T, surfels, count, max_depth, conf_threshold, time, max_time, time_delta):
    """-> image (H,W,4) u8, vertexConf, normalRad (H,W,4) f32, time (H,W) u16"""
    H, W = camera.H, camera.W
    img = np.zeros((H, W, 4), np.uint8)
    v, n = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    tm = np.zeros((H, W), np.uint16)
    lib().mfo_combined_predict(C.byref(camera), pose16(T), np.ascontiguousarray(surfels, np.float32).reshape(-1), count, max_depth, conf_threshold,
                               time, max_time, time_delta, img.reshape(-1), v.reshape(-1), n.reshape(-1), tm.reshape(-1))
    return img, v, n, tm


def fill_in(camera: Cam, pred_image, pred_vertex, pred_normal, raw_rgb, raw_depth, passthrough=False):
    H, W = camera.H, camera.W
    fi = np.zeros((H, W, 4), np.uint8)
    fv, fn = np.zeros((H, W, 4), np.float32), np.zeros((H, W, 4), np.float32)
    g = lambda a, t: np.ascontiguousarray(a, t).reshape(-1)
    lib().mfo_fill_in(C.byref(camera), g(pred_image, np.uint8), g(pred_vertex, np.float32), g(pred_normal, np.float32), g(raw_rgb, np.uint8),
                      g(raw_depth, np.float32), int(passthrough), fi.reshape(-1), fv.reshape(-1), fn.reshape(-1))
    return fi, fv, fn


def fusion_weight(T, T_last, weight_multiplier=1.0) -> float:
    return float(lib().mfo_fusion_weight(pose16(T), pose16(T_last), weight_multiplier))
