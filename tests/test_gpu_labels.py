"""-m gpu: the device label stage (mf_labels_gpu.hip) against the oracle's restatement of
MfSegmentation.cpp:220-522 on the same cases as the host form (tests/test_segmentation_host.py) -- exact -- and the whole
multi-model pipeline with the device stage against the pipeline with the host stage -- identical label images and models."""
import ctypes as C

import numpy as np
import pytest

import test_segmentation_host as tsh

pytestmark = pytest.mark.gpu


def _device_labels(L, binary, depth, mask, class_ids, proj, model_ids, model_cls, next_id, allow_new, prm, ignore):
    W, H = tsh.W, tsh.H
    p = np.array([prm.threshold, prm.weightDistance, prm.weightConvexity, prm.morphEdgeIterations, prm.morphEdgeRadius,
                  prm.morphMaskIterations, prm.morphMaskRadius, prm.removeEdges, prm.minRelSizeNew, prm.maxRelSizeNew,
                  prm.personClassID], np.float32)
    full = np.zeros((H, W), np.uint8)
    has_new, new_cls = C.c_int32(0), C.c_int32(-1)
    cid = np.ascontiguousarray(class_ids if len(class_ids) else [0], np.int32)
    mids, mcls = np.ascontiguousarray(model_ids, np.int32), np.ascontiguousarray(model_cls, np.int32)
    binary, depth = np.ascontiguousarray(binary, np.uint8), np.ascontiguousarray(depth, np.float32)
    mask, proj = np.ascontiguousarray(mask, np.uint8), np.ascontiguousarray(proj, np.uint8)
    rc = L.mf_k_segmentation_labels(W, H, binary.ctypes.data, depth.ctypes.data, mask.ctypes.data, cid.ctypes.data, len(class_ids),
                                    proj.ctypes.data, mids.ctypes.data, mcls.ctypes.data, len(model_ids), next_id, int(allow_new),
                                    p.ctypes.data, ignore.ctypes.data, full.ctypes.data, C.byref(has_new), C.byref(new_cls))
    assert rc == 0
    return full, bool(has_new.value), new_cls.value


@pytest.mark.parametrize("case", tsh.CASES)
def test_device_label_stage_matches_oracle(hip, oracle, case):
    from oracle import mfo_mm
    from maskfusion_amd import synth
    W, H, F = tsh.W, tsh.H, tsh.F
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2, cy=H / 2, n_objects=3, noise=True)
    rgb, depth, mask = st.frame(4)
    dF = oracle.bilateral(depth)
    v = oracle.create_vmap(dF, F, F, W / 2, H / 2, 3.0)
    edge = mfo_mm.geometric_edge_map(v, oracle.create_nmap(v), 150.0, 2.8)
    _, inv = mfo_mm.edge_binary(edge, 0.3, 1, 0)
    prm = mfo_mm.default_seg_params(**case["seg"])
    proj = np.zeros((H, W), np.uint8)
    if case["proj"] is not None:
        proj[mask == case["proj"]] = case["models"][1]
    m_in = tsh.case_mask(case, mask)
    ign_o, ign_d = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    ref = mfo_mm.mf_segmentation_cpu(W, H, inv, depth, m_in, case["cls"], proj, case["models"], case["mcls"], case["next_id"],
                                     case["allow"], ign_o, prm)
    got = _device_labels(hip, inv, depth, m_in, case["cls"], proj, case["models"], case["mcls"], case["next_id"], case["allow"], prm, ign_d)
    assert got[1] == ref[1] and got[2] == ref[2]
    assert np.array_equal(got[0], ref[0]), f"{int((got[0] != ref[0]).sum())} pixels differ"
    assert np.array_equal(ign_d, ign_o)


def test_component_numbering_stress(hip, oracle):
    """Random binary images (many small components, long snakes): label image via the full stage with no masks and one model
    that projects nowhere -> every pixel stays 0; the component count returned through the stage is checked indirectly by
    running a mask image whose ids equal the oracle's component labels (each large component must map to its own mask)."""
    from oracle import mfo_mm
    W, H = tsh.W, tsh.H
    rng = np.random.default_rng(5)
    binary = (rng.random((H, W)) < 0.62).astype(np.uint8) * 255
    binary[::7, :] = 255                      # long horizontal runs joined by
    binary[:, ::11] = 255                     # vertical ones: a few huge components plus debris
    binary[100:140, 60:200] = 0
    depth = np.full((H, W), 2.0, np.float32)
    n, labels, stats = mfo_mm.connected_components4(binary)
    big = [c for c in range(1, n) if stats[c][4] > 400][:200]
    mask = np.zeros((H, W), np.uint8)
    for k, c in enumerate(big):
        mask[labels == c] = k + 1
    cls = [0] + [40 + (k % 50) for k in range(len(big))]
    prm = mfo_mm.default_seg_params(morphMaskIterations=0, minRelSizeNew=0.9)
    ign_o, ign_d = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
    proj = np.zeros((H, W), np.uint8)
    ref = mfo_mm.mf_segmentation_cpu(W, H, binary, depth, mask, cls, proj, [0], [-1], 1, False, ign_o, prm)
    got = _device_labels(hip, binary, depth, mask, cls, proj, [0], [-1], 1, False, prm, ign_d)
    assert np.array_equal(got[0], ref[0]) and got[1] == ref[1]


def test_pipeline_device_labels_equal_host_labels(hip):
    from maskfusion_amd import MaskFusion, synth
    st = synth.Stream(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, n_objects=2, noise=True, object_motion=0.0)
    frames = [st.frame(k) for k in range(10)]
    runs = []
    for gpu in (1, 0):
        m = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, numGSurfels=1 << 20, numOSurfels=1 << 18,
                       enableMultipleModels=True, modelSpawnOffset=3, trackAllModels=False)
        for k, v in dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0,
                         mfMorphMaskIterations=1, newModelMinRelativeSize=0.004, gpuLabels=gpu).items():
            m.setParam(k, v)
        if gpu:
            m.preallocateModels(2)   # MaskFusion::preallocateModels: spawning from the pool must not change anything
        out = []
        for k in range(10):
            m.processFrame(frames[k][0], frames[k][1], mask=frames[k][2], classIDs=(0, 41, 42))
            out.append((m.downloadSegmentation().copy(), [(x.getID(), x.lastCount()) for x in m.getModels()], m.getCurrPose()))
        runs.append(out)
        m.close()
    assert max(len(o[1]) for o in runs[0]) >= 2
    for k, (a, b) in enumerate(zip(*runs)):
        assert a[1] == b[1], (k, a[1], b[1])
        assert np.array_equal(a[0], b[0]), (k, int((a[0] != b[0]).sum()))
        assert np.array_equal(a[2], b[2])


def test_device_label_stage_odd_width(hip, oracle):
    """Width 328: a wavefront of the run detection covers the end of one row and the start of the next."""
    from oracle import mfo_mm
    old = (tsh.W, tsh.H)
    tsh.W, tsh.H = 328, 248
    try:
        W, H = tsh.W, tsh.H
        rng = np.random.default_rng(11)
        binary = (rng.random((H, W)) < 0.7).astype(np.uint8) * 255
        binary[:, -3:] = 255; binary[:, :2] = 255          # runs that touch both image borders
        binary[60:90, 100:220] = 0
        depth = (2.0 + 0.001 * rng.random((H, W))).astype(np.float32)
        n, labels, stats = mfo_mm.connected_components4(binary)
        big = [c for c in range(1, n) if stats[c][4] > 300][:60]
        mask = np.zeros((H, W), np.uint8)
        for k, c in enumerate(big):
            mask[labels == c] = k + 1
        cls = [0] + [41 + (k % 7) for k in range(len(big))]
        proj = np.zeros((H, W), np.uint8); proj[100:200, 30:90] = 4
        prm = mfo_mm.default_seg_params(morphMaskIterations=1, minRelSizeNew=0.002)
        ign_o, ign_d = np.zeros((H, W), np.uint8), np.zeros((H, W), np.uint8)
        ref = mfo_mm.mf_segmentation_cpu(W, H, binary, depth, mask, cls, proj, [0, 4], [-1, 43], 5, True, ign_o, prm)
        got = _device_labels(hip, binary, depth, mask, cls, proj, [0, 4], [-1, 43], 5, True, prm, ign_d)
        assert got[1] == ref[1] and got[2] == ref[2]
        assert np.array_equal(got[0], ref[0]), f"{int((got[0] != ref[0]).sum())} pixels differ"
    finally:
        tsh.W, tsh.H = old
