"""Host half of MfSegmentation (label propagation): the product's C++ implementation behind mf_segmentation_labels against the
oracle's C restatement (MfSegmentation.cpp:220-522) -- integer / label work, so the comparison is exact.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from maskfusion_amd import synth

W, H = 320, 240
F = 264.0


def _product_labels(binary, depth, mask, class_ids, proj, model_ids, model_cls, next_id, allow_new, prm, ignore):
    from maskfusion_amd.lib import load
    L = load()
    p = np.array([prm.threshold, prm.weightDistance, prm.weightConvexity, prm.morphEdgeIterations, prm.morphEdgeRadius,
                  prm.morphMaskIterations, prm.morphMaskRadius, prm.removeEdges, prm.minRelSizeNew, prm.maxRelSizeNew,
                  prm.personClassID], np.float32)
    full = np.zeros((H, W), np.uint8)
    has_new, new_cls = C.c_int32(0), C.c_int32(-1)
    cid = np.ascontiguousarray(class_ids if len(class_ids) else [0], np.int32)
    mids = np.ascontiguousarray(model_ids, np.int32)
    mcls = np.ascontiguousarray(model_cls, np.int32)
    binary = np.ascontiguousarray(binary, np.uint8)
    depth = np.ascontiguousarray(depth, np.float32)
    mask = np.ascontiguousarray(mask, np.uint8)
    proj = np.ascontiguousarray(proj, np.uint8)
    rc = L.mf_segmentation_labels(W, H, binary.ctypes.data, depth.ctypes.data, mask.ctypes.data, cid.ctypes.data, len(class_ids),
                                  proj.ctypes.data, mids.ctypes.data, mcls.ctypes.data, len(model_ids), next_id, int(allow_new),
                                  p.ctypes.data, ignore.ctypes.data, full.ctypes.data, C.byref(has_new), C.byref(new_cls))
    assert rc == 0
    return full, bool(has_new.value), new_cls.value


@pytest.fixture(scope="module")
def scene(oracle):
    from oracle import mfo_mm
    st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2, cy=H / 2, n_objects=3, noise=True)
    rgb, depth, mask = st.frame(4)
    dF = oracle.bilateral(depth)
    v = oracle.create_vmap(dF, F, F, W / 2, H / 2, 3.0)
    n = oracle.create_nmap(v)
    edge = mfo_mm.geometric_edge_map(v, n, 150.0, 2.8)
    _, inv = mfo_mm.edge_binary(edge, 0.3, 1, 0)
    return depth, mask, inv


CASES = [
    # (class ids per mask id, model ids, model classes, projected-id painter, next id, allow new, seg overrides)
    dict(cls=[0, 41, 42, 43], models=[0], mcls=[-1], proj=None, next_id=1, allow=True, seg={}),
    dict(cls=[0, 41, 42, 43], models=[0], mcls=[-1], proj=None, next_id=1, allow=False, seg={}),
    dict(cls=[0, 41, 255, 43], models=[0], mcls=[-1], proj=None, next_id=1, allow=True, seg={}),               # person -> ignore
    dict(cls=[0, 41, 42, 43], models=[0, 1], mcls=[-1, 41], proj=1, next_id=2, allow=True, seg={}),             # mask 1 -> model 1
    dict(cls=[0, 41, 42, 43], models=[0, 3], mcls=[-1, 99], proj=2, next_id=4, allow=True, seg={}),             # class mismatch
    dict(cls=[], models=[0, 2], mcls=[-1, 42], proj=2, next_id=3, allow=True, seg={}),                           # no masks at all
    dict(cls=[0, 41, 42, 43], models=[0], mcls=[-1], proj=None, next_id=1, allow=True,
         seg=dict(morphMaskIterations=0, removeEdges=0, minRelSizeNew=0.001)),
    dict(cls=[0, 41, 42, 43], models=[0], mcls=[-1], proj=None, next_id=1, allow=True,
         seg=dict(morphMaskIterations=2, morphMaskRadius=2, minRelSizeNew=0.001)),
    # mask values without a class id (object 3 and a 255 "ignore" patch, as a precomputed Mask####.png may carry) count as "no mask":
    # upstream reads classIDs[mask] out of bounds there (MfSegmentation.cpp:226,311); round-1 code wrote past its vote table
    dict(cls=[0, 41, 42], models=[0], mcls=[-1], proj=None, next_id=1, allow=True, seg=dict(minRelSizeNew=0.001), out_of_range=True),
]


def case_mask(case, mask):
    if not len(case["cls"]):
        return np.zeros_like(mask)
    if case.get("out_of_range"):
        m = mask.copy()
        m[10:40, 10:80] = 255
        assert (m >= len(case["cls"])).sum() > 2000
        return m
    return mask


@pytest.mark.parametrize("case", CASES)
def test_label_propagation_matches_oracle(oracle, scene, case):
    from oracle import mfo_mm
    depth, mask, inv = scene
    prm = mfo_mm.default_seg_params(**case["seg"])
    proj = np.zeros((H, W), np.uint8)
    if case["proj"] is not None:
        obj = mask == case["proj"]
        proj[obj] = case["models"][1]          # the existing model projects where that object is
    m_in = case_mask(case, mask)
    ign_o = np.zeros((H, W), np.uint8)
    ign_p = np.zeros((H, W), np.uint8)
    ref = mfo_mm.mf_segmentation_cpu(W, H, inv, depth, m_in, case["cls"], proj, case["models"], case["mcls"], case["next_id"],
                                     case["allow"], ign_o, prm)
    got = _product_labels(inv, depth, m_in, case["cls"], proj, case["models"], case["mcls"], case["next_id"], case["allow"], prm, ign_p)
    assert got[1] == ref[1] and got[2] == ref[2]
    assert np.array_equal(got[0], ref[0]), f"{int((got[0] != ref[0]).sum())} pixels differ"
    assert np.array_equal(ign_p, ign_o)
    # sanity on the semantics (not only self-consistency)
    if case["cls"] and 255 in case["cls"]:
        person = case["cls"].index(255)
        assert (got[0][mask == person] == 255).all()
    if case["allow"] and case["cls"] and case["proj"] is None and ref[1]:
        assert (got[0] == case["next_id"]).sum() > 0
    if case.get("out_of_range"):   # the same labels as if those pixels had carried no mask at all
        clean = np.where(m_in >= len(case["cls"]), 0, m_in).astype(np.uint8)
        again = _product_labels(inv, depth, clean, case["cls"], proj, case["models"], case["mcls"], case["next_id"], case["allow"], prm,
                                np.zeros((H, W), np.uint8))
        assert np.array_equal(again[0], got[0])


def test_connected_components_numbering(oracle):
    """Raster-order numbering with 4-connectivity (cv::connectedComponentsWithStats(..., 4)) on a hand-made image."""
    from oracle import mfo_mm
    img = np.zeros((6, 8), np.uint8)
    img[0, 5:8] = 255          # comp 1 (first in raster order)
    img[1, 0:2] = 255          # comp 2
    img[2, 1] = 255            # joins comp 2 (4-connected through (1,1))
    img[3, 2] = 255            # diagonal to (2,1): NOT connected -> comp 3
    img[4, 5] = 255; img[5, 5] = 255; img[5, 4] = 255   # comp 4, U-shape resolved by union
    n, labels, stats = mfo_mm.connected_components4(img)
    assert n == 5
    assert labels[0, 6] == 1 and labels[1, 0] == 2 and labels[2, 1] == 2 and labels[3, 2] == 3 and labels[5, 4] == 4
    assert stats[2].tolist() == [0, 1, 2, 2, 3] and stats[4].tolist() == [4, 4, 2, 2, 3]
