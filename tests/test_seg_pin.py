"""Pins the oracle's restatement of the CPU half of MfSegmentation (ignore map, connected components, the five
edge-growing sweeps, the votes, the 65 % / 60 % / 5 % rules, label closing, the new-model rule) to the REFERENCE: oracle/build_seg.py
compiles lines 219-523 of Core/Segmentation/MfSegmentation.cpp from the reference's own text (cv::Mat / Eigen stand-ins in
oracle/cv_shim/mfcv.h; the OpenCV primitives underneath are the oracle's restatements) and both must label every pixel identically, on
the cases of tests/test_segmentation_host.py and on a 14-frame sequence in which the segmentation feeds back into itself through the
persistent ignore map.  Needs /root/reference (build container); the device label stage is then held to the oracle exactly by
tests/test_gpu_labels.py."""
import numpy as np
import pytest

import test_segmentation_host as tsh
from oracle import mfseg

pytestmark = pytest.mark.skipif(not mfseg.available(), reason="/root/reference is not here and no prebuilt oracle/_ref/libmf_seg.so")

W, H = tsh.W, tsh.H


@pytest.mark.parametrize("case", [c for c in tsh.CASES if not c.get("out_of_range")])   # (upstream reads out of bounds on that input)
def test_label_propagation_is_the_reference_text(oracle, scene, case):
    from oracle import mfo_mm
    depth, mask, inv = scene
    prm = mfo_mm.default_seg_params(**case["seg"])
    proj = np.zeros((H, W), np.uint8)
    if case["proj"] is not None:
        proj[mask == case["proj"]] = case["models"][1]
    m_in = tsh.case_mask(case, mask)
    ign_o, ign_r = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    ref = mfseg.mf_segmentation(W, H, inv, depth, m_in, case["cls"], proj, case["models"], case["mcls"], case["next_id"], case["allow"], ign_r, prm)
    got = mfo_mm.mf_segmentation_cpu(W, H, inv, depth, m_in, case["cls"], proj, case["models"], case["mcls"], case["next_id"], case["allow"], ign_o, prm)
    assert got[1] == ref[1] and got[2] == ref[2]
    assert np.array_equal(got[0], ref[0]), f"{int((got[0] != ref[0]).sum())} pixels differ"
    assert np.array_equal(ign_o, ign_r)
    assert len(np.unique(ref[0])) >= (2 if case["cls"] else 1)


scene = tsh.scene


def test_sequence_with_persistent_ignore_map(oracle):
    """Several frames through both, each side carrying its own semanticIgnoreMap from frame to frame (a person mask in every other
    frame, none in between: the no-mask branch re-uses the map), models appearing as they are spawned."""
    from maskfusion_amd import synth
    from oracle import mfo_mm
    st = synth.Stream(W=W, H=H, fx=tsh.F, fy=tsh.F, cx=W / 2, cy=H / 2, n_objects=3, noise=True, seed=5)
    prm = mfo_mm.default_seg_params(minRelSizeNew=0.002)
    ign_o, ign_r = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    models, mcls, next_id = [0], [-1], 1
    spawned = 0
    for k in range(10):
        rgb, depth, mask = st.frame(k)
        dF = oracle.bilateral(depth)
        v = oracle.create_vmap(dF, tsh.F, tsh.F, W / 2, H / 2, 3.0)
        edge = mfo_mm.geometric_edge_map(v, oracle.create_nmap(v), 150.0, 2.8)
        _, inv = mfo_mm.edge_binary(edge, 0.3, 1, 0)
        cls = [0, 41, 255, 43] if k % 2 == 0 else []          # object 2 is a "person" when masks are present
        m_in = mask if cls else np.zeros_like(mask)
        proj = np.zeros((H, W), np.uint8)
        for mid in models[1:]:
            proj[mask == mid] = mid                            # spawned models project where their object is
        ref = mfseg.mf_segmentation(W, H, inv, depth, m_in, cls, proj, models, mcls, next_id, True, ign_r, prm)
        got = mfo_mm.mf_segmentation_cpu(W, H, inv, depth, m_in, cls, proj, models, mcls, next_id, True, ign_o, prm)
        assert got[1] == ref[1] and got[2] == ref[2], k
        assert np.array_equal(got[0], ref[0]), (k, int((got[0] != ref[0]).sum()))
        assert np.array_equal(ign_o, ign_r), k
        if ref[1]:
            models.append(next_id); mcls.append(ref[2]); next_id += 1; spawned += 1
    assert spawned >= 1 and ign_r.any()
