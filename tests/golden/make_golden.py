#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.npz.

The reference has no tests, fixtures or golden vectors for this path and cannot be built or run here, so these
vectors are NOT reference outputs: they are outputs of the oracle (oracle/mf_oracle.c) on small seeded inputs, frozen so that
  * an accidental change of the oracle's arithmetic is caught on CPU (tests/test_golden.py), and
  * the HIP kernels can be compared with fixed numbers on the GPU box (tests/test_gpu_golden.py).
Regenerate only together with a deliberate, documented change of the restatement:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from maskfusion_amd import synth  # noqa: E402
from oracle import mfo, mfo_mm, mfo_rgbd  # noqa: E402

W, H, F = 160, 120, 132.0


def inputs():
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, noise=True, n_objects=2, seed=7)
    return st, [st.frame(k) for k in range(6)]


def build():
    st, fr = inputs()
    out = {}
    depth = fr[0][1]
    dF = mfo.bilateral(depth)
    out["bilateral"] = dF
    d1 = mfo.pyrdown_f(dF)
    out["pyrdown_f"] = d1
    v0 = mfo.create_vmap(dF, F, F, W / 2.0, H / 2.0, 3.0)
    n0 = mfo.create_nmap(v0)
    out["vmap0"], out["nmap0"] = v0, n0
    # icpStep: frame 1 against frame 0's maps (global frame = camera frame 0), pose guess = identity
    dF1 = mfo.bilateral(fr[1][1])
    v1 = mfo.create_vmap(dF1, F, F, W / 2.0, H / 2.0, 3.0)
    n1 = mfo.create_nmap(v1)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    A, b, res = mfo.icp_step(I3, z3, v1, n1, I3, z3, F, F, W / 2.0, H / 2.0, v0, n0)
    out["icp_A"], out["icp_b"], out["icp_res"] = A, b, res
    # photometric pieces
    g0 = mfo_rgbd.image_to_intensity(fr[0][0]); g1 = mfo_rgbd.image_to_intensity(fr[1][0])
    out["gray1"] = g1
    out["gray1_l1"] = mfo.pyrdown_u8(g1)
    dx, dy = mfo_rgbd.derivative_images(g1)
    out["dIdx"], out["dIdy"] = dx, dy
    dd = depth.astype(np.float32).copy(); dd[(dd <= 0) | (dd > 6)] = np.nan
    T = np.linalg.inv(st.gt_pose(0)) @ st.gt_pose(1)
    K = np.array([[F, 0, W / 2.0], [0, F, H / 2.0], [0, 0, 1.0]])
    krk = (K @ T[:3, :3] @ np.linalg.inv(K)).astype(np.float32); kt = (K @ T[:3, 3]).astype(np.float32)
    cor, sig, cnt = mfo_rgbd.rgb_residual(64.0, dx, dy, dd, dd, g0, g1, kt, krk)
    out["rgb_krk"], out["rgb_kt"], out["rgb_depth"], out["gray0"] = krk, kt, dd, g0
    out["rgb_count_sigma"] = np.array([cnt, sig], np.int64)
    out["rgb_corr_valid"] = (cor["valid"] != 0).reshape(H, W)
    out["rgb_corr_u0"] = cor["zx"].reshape(H, W); out["rgb_corr_v0"] = cor["zy"].reshape(H, W); out["rgb_corr_diff"] = cor["diff"].reshape(H, W)
    # whole single-model pipeline, reference-default tracking
    o = mfo.Oracle(W, H, F, F, W / 2.0, H / 2.0, icpWeight=20.0, so3=1, capacity=W * H * 3)
    poses, counts = [], []
    for k in range(6):
        o.process_frame(fr[k][0], fr[k][1])
        poses.append(o.pose); counts.append(o.count)
    out["pipeline_poses"], out["pipeline_counts"] = np.array(poses), np.array(counts)
    o.close()
    # label stage
    edge = mfo_mm.geometric_edge_map(v0, n0, 150.0, 2.8)
    _, inv = mfo_mm.edge_binary(edge, 0.3, 1, 0)
    prm = mfo_mm.default_seg_params(morphMaskIterations=1, minRelSizeNew=0.002)
    ign = np.zeros((H, W), np.uint8)
    full, has_new, new_cls = mfo_mm.mf_segmentation_cpu(W, H, inv, depth, fr[0][2], [0, 41, 42], np.zeros((H, W), np.uint8), [0], [-1], 1,
                                                         True, ign, prm)
    out["seg_binary"], out["seg_full"], out["seg_new"] = inv, full, np.array([int(has_new), new_cls])
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz"), **build())
    print("written")
