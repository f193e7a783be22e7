"""ctypes binding of oracle/_ref/libmf_ref.so: the reference's own CUDA translation units (Core/Cuda/reduce.cu, cudafuncs.cu,
segmentation.cu) compiled for the CPU by oracle/build_ref.py.  TEST INFRASTRUCTURE ONLY: used to pin oracle/mf_oracle.c
(tests/test_ref_pin.py) and to generate tests/golden/ref_vectors.npz (tests/golden/make_ref_golden.py).

Helper signatures mirror oracle/mfo.py, mfo_rgbd.py and mfo_mm.py, so the same arrays go to both.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build_ref
from .mfo_rgbd import DATATERM

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")

# launch configuration of the reductions (the reference reads it from GPUConfig per device; any multiple of the warp
# size is legal, results differ only in float summation order)
THREADS, BLOCKS = 128, 6

_lib = None


def available() -> bool:
    return build_ref.reference_available() or os.path.exists(build_ref.LIB)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = build_ref.build()
    if path is None:
        raise RuntimeError("oracle/_ref/libmf_ref.so is absent and /root/reference is not here to build it from")
    L = C.CDLL(path)
    i, f = C.c_int, C.c_float
    L.mfref_pyrdown_gauss_f.argtypes = [f32p, f32p, i, i]
    L.mfref_pyrdown_gauss_u8.argtypes = [u8p, u8p, i, i]
    L.mfref_create_vmap.argtypes = [f32p, f32p, i, i, f, f, f, f, f]
    L.mfref_create_nmap.argtypes = [f32p, f32p, i, i]
    L.mfref_copy_maps.argtypes = [f32p, f32p, f32p, f32p, i, i]
    L.mfref_resize_map.argtypes = [f32p, f32p, i, i, i]
    L.mfref_transform_maps.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p, i, i]
    L.mfref_icp_step.argtypes = [f32p, f32p, f32p, f32p, f32p, f32p, f, f, f, f, f32p, f32p, f, f, i, i, f32p, f32p, f32p, i, i]
    L.mfref_vertices_to_depth.argtypes = [f32p, f32p, i, i, f]
    L.mfref_image_to_intensity.argtypes = [u8p, u8p, i, i]
    L.mfref_derivative_images.argtypes = [u8p, i16p, i16p, i, i]
    L.mfref_project_to_cloud.argtypes = [f32p, f32p, i, i, f, f, f, f]
    L.mfref_rgb_residual.argtypes = [f, i16p, i16p, f32p, f32p, u8p, u8p, C.c_void_p, f, f32p, f32p, i, i, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), i, i]
    L.mfref_rgb_step.argtypes = [C.c_void_p, f, f32p, f, f, i16p, i16p, f, i, i, f32p, f32p, i, i]
    L.mfref_so3_step.argtypes = [u8p, u8p, f32p, f32p, f32p, i, i, f32p, f32p, f32p, i, i]
    L.mfref_geometric_edge_map.argtypes = [f32p, f32p, f32p, i, i, f, f]
    L.mfref_threshold_map.argtypes = [f32p, u8p, i, i, f]
    L.mfref_invert_map.argtypes = [u8p, u8p, i, i]
    L.mfref_morph_closing_u8.argtypes = [u8p, i, i, i, i]
    _lib = L
    return L


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def pyrdown_f(src):
    H, W = src.shape
    dst = np.empty((H // 2, W // 2), np.float32)
    lib().mfref_pyrdown_gauss_f(_f(src), dst, W, H)
    return dst


def pyrdown_u8(src):
    H, W = src.shape
    dst = np.empty((H // 2, W // 2), np.uint8)
    lib().mfref_pyrdown_gauss_u8(np.ascontiguousarray(src, np.uint8), dst, W, H)
    return dst


def create_vmap(depth, fx, fy, cx, cy, cutoff):
    H, W = depth.shape
    v = np.empty((3, H, W), np.float32)
    lib().mfref_create_vmap(_f(depth), v, W, H, fx, fy, cx, cy, cutoff)
    return v


def create_nmap(vmap):
    _, H, W = vmap.shape
    n = np.empty((3, H, W), np.float32)
    lib().mfref_create_nmap(_f(vmap), n, W, H)
    return n


def copy_maps(v4, n4):
    H, W, _ = v4.shape
    v = np.empty((3, H, W), np.float32)
    n = np.empty((3, H, W), np.float32)
    lib().mfref_copy_maps(_f(v4), _f(n4), v, n, W, H)
    return v, n


def resize_map(m, normalize):
    _, H, W = m.shape
    out = np.empty((3, H // 2, W // 2), np.float32)
    lib().mfref_resize_map(_f(m), out, W, H, int(normalize))
    return out


def transform_maps(v, n, R, t):
    _, H, W = v.shape
    vo, no = np.empty((3, H, W), np.float32), np.empty((3, H, W), np.float32)
    lib().mfref_transform_maps(_f(v), _f(n), _f(R).reshape(9), _f(t), vo, no, W, H)
    return vo, no


def icp_step(Rcurr, tcurr, vc, nc, Rprev_inv, tprev, fx, fy, cx, cy, vp, npv, dist=0.10,
             angle=float(np.sin(np.float32(20.0 * 3.14159254 / 180.0))), threads=THREADS, blocks=BLOCKS):
    _, H, W = vc.shape
    A = np.zeros(36, np.float32)
    b = np.zeros(6, np.float32)
    res = np.zeros(2, np.float32)
    g = lambda a: _f(a).reshape(-1)
    lib().mfref_icp_step(g(Rcurr), g(tcurr), g(vc), g(nc), g(Rprev_inv), g(tprev), fx, fy, cx, cy, g(vp), g(npv), dist, angle, W, H,
                         A, b, res, threads, blocks)
    return A.reshape(6, 6), b, res


def vertices_to_depth(v4, cutoff=6.0):
    H, W, _ = v4.shape
    d = np.empty((H, W), np.float32)
    lib().mfref_vertices_to_depth(_f(v4).reshape(-1), d, W, H, cutoff)
    return d


def image_to_intensity(img4):
    H, W, ch = img4.shape
    assert ch == 4, "the reference reads uchar4 texels of a cudaArray"
    out = np.empty((H, W), np.uint8)
    lib().mfref_image_to_intensity(np.ascontiguousarray(img4, np.uint8).reshape(-1), out, W, H)
    return out


def derivative_images(img):
    H, W = img.shape
    dx = np.empty((H, W), np.int16)
    dy = np.empty((H, W), np.int16)
    lib().mfref_derivative_images(np.ascontiguousarray(img, np.uint8), dx, dy, W, H)
    return dx, dy


def project_to_cloud(depth, fx, fy, cx, cy):
    H, W = depth.shape
    c = np.empty((H, W, 3), np.float32)
    lib().mfref_project_to_cloud(_f(depth), c.reshape(-1), W, H, fx, fy, cx, cy)
    return c


def rgb_residual(min_scale, dIdx, dIdy, last_depth, next_depth, last_image, next_image, kt, krkinv, max_depth_delta=0.07,
                 threads=THREADS, blocks=BLOCKS):
    H, W = next_image.shape
    corres = np.zeros(W * H, DATATERM)
    sig, cnt = C.c_int32(0), C.c_int32(0)
    lib().mfref_rgb_residual(min_scale, dIdx, dIdy, _f(last_depth), _f(next_depth), last_image, next_image, corres.ctypes.data,
                             max_depth_delta, _f(kt), _f(krkinv).reshape(9), W, H, C.byref(sig), C.byref(cnt), threads, blocks)
    return corres, sig.value, cnt.value


def rgb_step(corres, sigma, cloud, fx, fy, dIdx, dIdy, W, H, sobel_scale=0.125, threads=THREADS, blocks=BLOCKS):
    A = np.zeros(36, np.float32)
    b = np.zeros(6, np.float32)
    lib().mfref_rgb_step(corres.ctypes.data, sigma, _f(cloud).reshape(-1), fx, fy, dIdx, dIdy, sobel_scale, W, H, A, b, threads, blocks)
    return A.reshape(6, 6), b


def so3_step(last_image, next_image, image_basis, kinv, krlr, threads=THREADS, blocks=BLOCKS):
    H, W = next_image.shape
    A = np.zeros(9, np.float32)
    b = np.zeros(3, np.float32)
    r = np.zeros(2, np.float32)
    g = lambda a: _f(a).reshape(9)
    lib().mfref_so3_step(last_image, next_image, g(image_basis), g(kinv), g(krlr), W, H, A, b, r, threads, blocks)
    return A.reshape(3, 3), b, r


def geometric_edge_map(vmap, nmap, wD, wC):
    _, H, W = vmap.shape
    out = np.empty((H, W), np.float32)
    lib().mfref_geometric_edge_map(_f(vmap), _f(nmap), out, W, H, wD, wC)
    return out


def threshold_map(edge, threshold):
    H, W = edge.shape
    out = np.empty((H, W), np.uint8)
    lib().mfref_threshold_map(_f(edge), out, W, H, threshold)
    return out


def invert_map(m):
    H, W = m.shape
    out = np.empty((H, W), np.uint8)
    lib().mfref_invert_map(np.ascontiguousarray(m, np.uint8), out, W, H)
    return out


def morph_closing_u8(m, radius, iterations):
    H, W = m.shape
    d = np.ascontiguousarray(m, np.uint8).copy()
    lib().mfref_morph_closing_u8(d, W, H, radius, iterations)
    return d
