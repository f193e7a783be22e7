#!/usr/bin/env python3
"""Generates tests/golden/glsl_vectors.npz: what the REFERENCE's own GLSL shaders compute on small seeded inputs.

Every array "out_*" is produced by oracle/_ref/libmf_glsl.so -- Core/Shaders/*.vert / *.frag of martinruenz/maskfusion compiled as
C++ by oracle/build_glsl.py (mechanical edits only, listed there) and run by oracle/glsl_shim/mfgl_api.cpp under the documented
OpenGL rules.  It needs /root/reference, so it runs in the build container only; the vectors are committed so that
tests/test_glsl_pin.py can hold oracle/mf_oracle.c to the reference anywhere.  Inputs are stored next to the outputs.
Two cases: 96x72 (general: uv * cols carries fp32 rounding, the window loops make 4 or 5 steps) and 64x32 (powers of two: both
effects vanish and every pass must agree bit for bit with no tolerance anywhere).
Regenerate with:  python tests/golden/make_glsl_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from maskfusion_amd import synth  # noqa: E402
from oracle import mfo, mfglsl  # noqa: E402


def case(W, H, F, n_warm, seed):
    K = (F, F, W / 2.0, H / 2.0)
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, noise=True, n_objects=1, seed=seed)
    fr = [st.frame(k) for k in range(n_warm + 1)]
    out = dict(W=W, H=H, K=np.array(K, np.float32))
    rgb0, d0, _ = fr[0]
    out["in_rgb0"], out["in_depth0"] = rgb0, d0
    out["out_bilateral0"] = mfglsl.bilateral(d0)
    s0, nf = mfglsl.init_surfels(rgb0, d0, out["out_bilateral0"], K, 1, 20.0)
    out["out_init"], out["out_init_filtered_count"] = s0, nf
    # a map a few frames old, built by the oracle with the poses given (the input of the per-pass comparisons)
    o = mfo.Oracle(W, H, *K, icpWeight=100.0, capacity=W * H * 4, so3=0, confGlobal=2.0)
    for k in range(n_warm):
        o.process_frame(fr[k][0], fr[k][1], in_pose=st.gt_pose(k).astype(np.float32) if k else None)
    surf, cnt, t = o.surfels().copy(), o.count, o.tick
    o.close()
    T = st.gt_pose(n_warm).astype(np.float32)
    rgb, depth, mask = fr[n_warm]
    mask = (mask > 0).astype(np.uint8)                     # object pixels carry id 1: they are not the background's to fuse
    dF = mfglsl.bilateral(depth)
    out.update(in_surfels=surf[:cnt], in_time=t, in_pose=T, in_rgb=rgb, in_depth=depth, in_mask=mask, out_bilateral=dF)
    idx = mfglsl.predict_indices(T, surf[:cnt], t, 20.0, 200, W, H, K)
    out.update(out_index=idx[0], out_index_vc=idx[1], out_index_ct=idx[2], out_index_nr=idx[3])
    op, best, rec = mfglsl.fuse_data(T, rgb, depth, dF, mask, 0, t, 0.8, 3.0, K, *idx)
    out.update(out_data_op=op, out_data_best=best, out_data_rec=rec)
    upd = mfglsl.fuse_update(surf[:cnt], t, op, best, rec)
    out["out_update"] = upd
    idx2 = mfglsl.predict_indices(T, upd, t, 20.0, 200, W, H, K)
    out.update(out_index2=idx2[0], out_index2_vc=idx2[1], out_index2_ct=idx2[2], out_index2_nr=idx2[3])
    cl, keep = mfglsl.clean(T, upd, op, rec, t, 200, 2.0, 20.0, 0.9, 0, K, *idx2, dF, mask)
    out.update(out_clean=cl, out_clean_keep=keep)
    img, vc, nr, tm = mfglsl.combined_predict(T, cl, 20.0, 2.0, t, t, 200, W, H, K)
    out.update(out_splat_image=img, out_splat_vc=vc, out_splat_nr=nr, out_splat_time=tm)
    fi, fv, fn = mfglsl.fill_in(img, vc, nr, rgb, dF, False, K)
    out.update(out_fill_image=fi, out_fill_vertex=fv, out_fill_normal=fn)
    # GlobalProjection of two models: the cleaned map (id 0; its confidences are raised so that part of it passes the fixed threshold
    # 12) and a copy of its near half pulled 5 cm towards the camera as "object" id 3 in front of it
    bgm = cl.copy()
    bgm[::2, 3] += 20.0
    near = bgm[bgm[:, 3] > 12.0][::3].copy()
    Tinv = np.linalg.inv(T.astype(np.float64))
    loc = near[:, :3] @ Tinv[:3, :3].T + Tinv[:3, 3]
    loc *= (1.0 - 0.05 / np.maximum(np.linalg.norm(loc, axis=1, keepdims=True), 1e-6))
    near[:, :3] = (loc @ T[:3, :3].astype(np.float64).T + T[:3, 3]).astype(np.float32)
    out.update(in_gp_bg=bgm, in_gp_obj=near)
    out["out_gp_ids"] = mfglsl.global_projection([(bgm, T, 0), (near, T, 3)], t, 200, 3.0, W, H, K)
    return out


def main():
    data = {}
    for name, args in (("general", (96, 72, 79.2, 6, 21)), ("pow2", (64, 32, 52.8, 5, 22))):
        for k, v in case(*args).items():
            data[f"{name}/{k}"] = np.asarray(v)
    path = os.path.join(ROOT, "tests", "golden", "glsl_vectors.npz")
    np.savez_compressed(path, **data)
    print(path, os.path.getsize(path), "bytes,", len(data), "arrays")


if __name__ == "__main__":
    main()
