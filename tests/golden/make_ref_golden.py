#!/usr/bin/env python3
"""Generates tests/golden/ref_vectors.npz: outputs of the REFERENCE's own code on small seeded inputs.

Unlike oracle_vectors.npz (frozen oracle outputs), every array "<case>/out*" in this file is produced by
oracle/_ref/libmf_ref.so -- Core/Cuda/reduce.cu, cudafuncs.cu and segmentation.cu of martinruenz/maskfusion compiled for
the CPU by oracle/build_ref.py (host stand-ins for the CUDA headers, fibers for the thread grid; see
oracle/ref_shim/mfref_cuda.h for what that does and does not pin).  It needs /root/reference, so it runs in the build
container only; the vectors are committed so that
  * tests/test_ref_pin.py checks oracle/mf_oracle.c against the reference on CPU, anywhere, and
  * tests/test_gpu_ref_golden.py checks the HIP kernels against the same reference outputs on the GPU box.
Each case stores its inputs ("<case>/in_*") next to its outputs: no test depends on regenerating an input bit-exactly.
Regenerate with:  python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from maskfusion_amd import synth  # noqa: E402
from oracle import mfo, mfref  # noqa: E402

W, H, F = 160, 120, 132.0
CX, CY = W / 2.0, H / 2.0


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


def build():
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=CX, cy=CY, noise=True, n_objects=2, seed=11)
    fr = [st.frame(k) for k in range(3)]
    out = {}

    def case(name, ins, outs):
        for k, v in ins.items():
            v = np.asarray(v)
            # an input that is byte-identical to an array already stored becomes an alias (a 0-d string naming it)
            for ok_, ov in out.items():
                if ov.dtype == v.dtype and ov.shape == v.shape and v.size > 64 and ov.tobytes() == v.tobytes():
                    v = np.array(ok_)
                    break
            out[f"{name}/in_{k}"] = v
        for k, v in outs.items():
            out[f"{name}/out_{k}"] = np.asarray(v)

    # inputs only: the sensor noise is smoothed as in the pipeline (the filter itself is GLSL upstream and not part of this pin)
    d0, d1 = mfo.bilateral(fr[0][1].astype(np.float32)), mfo.bilateral(fr[1][1].astype(np.float32))
    # ---- a3: depth pyramid, vertex / normal maps ----
    p1 = mfref.pyrdown_f(d0)
    p2 = mfref.pyrdown_f(p1)
    case("pyrdown_f", dict(src=d0), dict(l1=p1, l2=p2))
    v0 = mfref.create_vmap(d0, F, F, CX, CY, 3.0)
    n0 = mfref.create_nmap(v0)
    case("vmap_nmap", dict(depth=d0, K=np.array([F, F, CX, CY, 3.0], np.float32)), dict(vmap=v0, nmap=n0))
    v1 = mfref.create_vmap(d1, F, F, CX, CY, 3.0)
    n1 = mfref.create_nmap(v1)
    # ---- a4: model-side maps (copyMaps -> resize x2 -> transform) from a float4 prediction ----
    v4 = np.zeros((H, W, 4), np.float32)
    n4 = np.zeros((H, W, 4), np.float32)
    ok = ~np.isnan(v0[0]) & ~np.isnan(n0[0])
    for c in range(3):
        v4[..., c] = np.where(ok, v0[c], 0.0)
        n4[..., c] = np.where(ok, n0[c], 0.0)
    v4[..., 3] = np.where(ok, 5.0, 0.0)
    n4[..., 3] = np.where(ok, 0.01, 0.0)
    cv, cn = mfref.copy_maps(v4, n4)
    rv1, rn1 = mfref.resize_map(cv, False), mfref.resize_map(cn, True)
    rv2, rn2 = mfref.resize_map(rv1, False), mfref.resize_map(rn1, True)
    R = rot(0.01, -0.02, 0.015)
    t = np.array([0.02, -0.01, 0.03], np.float32)
    tv, tn = mfref.transform_maps(cv, cn, R, t)
    case("model_maps", dict(v4=v4, n4=n4, R=R, t=t), dict(copy_v=cv, copy_n=cn, res_v1=rv1, res_n1=rn1, res_v2=rv2, res_n2=rn2,
                                                           tr_v=tv, tr_n=tn))
    # ---- a7: icpStep, frame 1 against frame 0 (global frame = camera frame 0), a non-trivial pose guess ----
    Rc = rot(0.004, -0.003, 0.002)
    tc = np.array([0.003, -0.002, 0.004], np.float32)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    A, b, res = mfref.icp_step(Rc, tc, v1, n1, I3, z3, F, F, CX, CY, v0, n0)
    case("icp_step", dict(Rcurr=Rc, tcurr=tc, vc=v1, nc=n1, Rprev_inv=I3, tprev=z3, vp=v0, np=n0, K=np.array([F, F, CX, CY], np.float32)),
         dict(A=A, b=b, res=res))
    # same inputs, another launch shape: only the float summation order may differ
    A2, b2, res2 = mfref.icp_step(Rc, tc, v1, n1, I3, z3, F, F, CX, CY, v0, n0, threads=64, blocks=11)
    case("icp_step_alt_launch", {}, dict(A=A2, b=b2, res=res2))
    # ---- a5 / a12: intensity, u8 pyramid, derivative images, vertices -> depth, point cloud ----
    rgba0 = np.concatenate([fr[0][0], np.full((H, W, 1), 255, np.uint8)], axis=2)
    rgba1 = np.concatenate([fr[1][0], np.full((H, W, 1), 255, np.uint8)], axis=2)
    g0, g1 = mfref.image_to_intensity(rgba0), mfref.image_to_intensity(rgba1)
    case("intensity", dict(rgba=rgba1), dict(gray=g1))
    g1_1 = mfref.pyrdown_u8(g1)
    g1_2 = mfref.pyrdown_u8(g1_1)
    case("pyrdown_u8", dict(src=g1), dict(l1=g1_1, l2=g1_2))
    dx, dy = mfref.derivative_images(g1)
    case("derivative", dict(src=g1), dict(dx=dx, dy=dy))
    vd = mfref.vertices_to_depth(v4, 2.5)
    case("vertices_to_depth", dict(v4=v4, cutoff=np.float32(2.5)), dict(depth=vd))
    vd_full = mfref.vertices_to_depth(v4, 20.0)
    cloud = mfref.project_to_cloud(vd_full, F, F, CX, CY)
    case("project_cloud", dict(depth=vd_full, K=np.array([F, F, CX, CY], np.float32)), dict(cloud=cloud))
    # ---- a8 / a9: computeRgbResidual + rgbStep (model depth on both sides, Q1) ----
    T = np.linalg.inv(st.gt_pose(0)) @ st.gt_pose(1)
    K = np.array([[F, 0, CX], [0, F, CY], [0, 0, 1.0]])
    krk = (K @ T[:3, :3] @ np.linalg.inv(K)).astype(np.float32)
    kt = (K @ T[:3, 3]).astype(np.float32)
    for name, ms in (("rgb_residual", 64.0), ("rgb_residual_l0scale", 1600.0)):
        cor, sig, cnt = mfref.rgb_residual(ms, dx, dy, vd_full, vd_full, g0, g1, kt, krk)
        inval = cor["valid"] == 0     # the reference leaves the other fields of an invalid DataTerm uninitialised (reduce.cu:818-820)
        for fld in ("zx", "zy", "ox", "oy", "diff"):
            cor[fld][inval] = 0
        case(name, dict(minScale=np.float32(ms), dIdx=dx, dIdy=dy, lastDepth=vd_full, nextDepth=vd_full, lastImage=g0, nextImage=g1, kt=kt,
                        krkinv=krk),
             dict(corres=cor, sigma_count=np.array([sig, cnt], np.int64)))
        if name == "rgb_residual":
            cor64, cnt64 = cor, cnt
    Argb, brgb = mfref.rgb_step(cor64, float(cnt64), cloud, F, F, dx, dy, W, H)
    case("rgb_step", dict(corres=cor64, sigma=np.float32(cnt64), cloud=cloud, dIdx=dx, dIdy=dy, K=np.array([F, F], np.float32)),
         dict(A=Argb, b=brgb))
    # ---- a10: so3Step on the level-2 images, identity and a small rotation ----
    g0_2 = mfref.pyrdown_u8(mfref.pyrdown_u8(g0))
    K2 = np.array([[F / 4, 0, CX / 4], [0, F / 4, CY / 4], [0, 0, 1.0]])
    for name, Rr in (("so3_step_identity", np.eye(3)), ("so3_step_rotated", rot(0.01, -0.015, 0.005).astype(np.float64))):
        # RGBDOdometry.cpp:277-289: imageBasis = K R K^-1 (the homography), kinv = K^-1, krlr = K R, all cast to float
        homography = (K2 @ Rr @ np.linalg.inv(K2)).astype(np.float32)
        kinv = np.linalg.inv(K2).astype(np.float32)
        krlr = (K2 @ Rr).astype(np.float32)
        As, bs, rs = mfref.so3_step(g0_2, g1_2, homography, kinv, krlr)
        case(name, dict(last=g0_2, next=g1_2, imageBasis=homography, kinv=kinv, krlr=krlr), dict(A=As, b=bs, res=rs))
    # ---- a20 (GPU half): geometric edge map, threshold, closing, invert; both parameter sets ----
    for name, (wD, wC, th, rad, it) in (("edges_gui", (150.0, 2.8, 0.3, 1, 0)), ("edges_core", (1.0, 1.0, 0.1, 1, 3)),
                                        ("edges_r2", (150.0, 2.8, 0.3, 2, 2))):
        e = mfref.geometric_edge_map(v0, n0, wD, wC)
        tb = mfref.threshold_map(e, th)
        mb = mfref.morph_closing_u8(tb, rad, it)
        inv = mfref.invert_map(mb)
        case(name, dict(vmap=v0, nmap=n0, prm=np.array([wD, wC, th, rad, it], np.float32)), dict(edge=e, thresh=tb, morph=mb, inverted=inv))
    return out


if __name__ == "__main__":
    o = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.npz")
    np.savez_compressed(path, **o)
    print("written", path, os.path.getsize(path), "bytes,", len(o), "arrays")
