"""-m gpu: HIP kernels and the pipeline against the frozen vectors of tests/golden/ (oracle outputs on small seeded inputs; see
tests/golden/make_golden.py for why they are not reference outputs).  Tolerances as in the per-kernel parity tests."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from gpu_util import dev, empty, host, nan_equal_close

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
W, H, F = make_golden.W, make_golden.H, make_golden.F


def test_kernels_against_golden(hip):
    import torch
    st, fr = make_golden.inputs()
    d = dev(fr[0][1]); out = empty((H, W))
    assert hip.mf_k_bilateral(d.data_ptr(), out.data_ptr(), W, H, None) == 0
    dF = host(out)
    assert nan_equal_close(dF, GOLD["bilateral"], 2e-5, 1e-6)[1] == 0
    dg = dev(GOLD["bilateral"]); o1 = empty((H // 2, W // 2))
    assert hip.mf_k_pyrdown_f(dg.data_ptr(), o1.data_ptr(), W, H, None) == 0
    assert nan_equal_close(host(o1), GOLD["pyrdown_f"], 2e-6, 1e-7)[1] == 0
    v, n = empty((3, H, W)), empty((3, H, W))
    assert hip.mf_k_vmap_nmap(dg.data_ptr(), v.data_ptr(), n.data_ptr(), W, H, F, F, W / 2.0, H / 2.0, 3.0, None) == 0
    assert nan_equal_close(host(v), GOLD["vmap0"], 2e-6, 1e-7)[1] == 0
    assert nan_equal_close(host(n), GOLD["nmap0"], 5e-5, 1e-5)[1] == 0
    # intensity / pyramid / derivative images: exact
    img = dev(fr[1][0]); g = empty((H, W), torch.uint8)
    assert hip.mf_k_intensity(img.data_ptr(), 3, g.data_ptr(), W * H, None) == 0
    assert np.array_equal(host(g), GOLD["gray1"])
    g1 = empty((H // 2, W // 2), torch.uint8)
    assert hip.mf_k_pyrdown_u8(g.data_ptr(), g1.data_ptr(), W, H, None) == 0
    assert np.array_equal(host(g1), GOLD["gray1_l1"])
    dx, dy = empty((H, W), torch.int16), empty((H, W), torch.int16)
    assert hip.mf_k_derivative_images(g.data_ptr(), dx.data_ptr(), dy.data_ptr(), W, H, None) == 0
    assert np.array_equal(host(dx), GOLD["dIdx"]) and np.array_equal(host(dy), GOLD["dIdy"])
    # photometric correspondences: exact
    dd, g0 = dev(GOLD["rgb_depth"]), dev(GOLD["gray0"])
    cor = empty((W * H * 8,), torch.uint8)
    sums = np.zeros(2, np.int32)
    kt, krk = np.ascontiguousarray(GOLD["rgb_kt"]), np.ascontiguousarray(GOLD["rgb_krk"])
    assert hip.mf_k_rgb_residual(64.0, dx.data_ptr(), dy.data_ptr(), dd.data_ptr(), dd.data_ptr(), g0.data_ptr(), g.data_ptr(), 0.07,
                                 kt.ctypes.data, krk.ctypes.data, W, H, cor.data_ptr(), sums.ctypes.data, None) == 0
    assert sums.tolist() == GOLD["rgb_count_sigma"].tolist()
    got = host(cor).view(np.dtype([("u0", np.int16), ("v0", np.int16), ("diff", np.float32)])).reshape(H, W)
    vmask = GOLD["rgb_corr_valid"]
    assert np.array_equal(got["u0"] >= 0, vmask)
    assert np.array_equal(got["u0"][vmask], GOLD["rgb_corr_u0"][vmask]) and np.array_equal(got["diff"][vmask], GOLD["rgb_corr_diff"][vmask])


def test_pipeline_and_labels_against_golden(hip):
    from maskfusion_amd import MaskFusion
    st, fr = make_golden.inputs()
    mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=20.0, so3=True, enableMultipleModels=False, numGSurfels=W * H * 3)
    for k in range(6):
        mf.processFrame(fr[k][0], fr[k][1])
        assert np.abs(mf.getCurrPose() - GOLD["pipeline_poses"][k]).max() < 1e-4, k
        assert abs(mf.getBackgroundModel().lastCount() - int(GOLD["pipeline_counts"][k])) <= max(20, 0.005 * GOLD["pipeline_counts"][k])
    mf.close()
    # label stage (device form), exact
    import test_gpu_labels as tgl
    import test_segmentation_host as tsh
    from oracle import mfo_mm
    old = (tsh.W, tsh.H)
    tsh.W, tsh.H = W, H
    try:
        prm = mfo_mm.default_seg_params(morphMaskIterations=1, minRelSizeNew=0.002)
        ign = np.zeros((H, W), np.uint8)
        full, has_new, new_cls = tgl._device_labels(hip, GOLD["seg_binary"], fr[0][1], fr[0][2], [0, 41, 42], np.zeros((H, W), np.uint8), [0],
                                                    [-1], 1, True, prm, ign)
        assert np.array_equal(full, GOLD["seg_full"]) and [int(has_new), new_cls] == GOLD["seg_new"].tolist()
    finally:
        tsh.W, tsh.H = old
