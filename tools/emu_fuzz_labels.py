"""Differential fuzzing of the label stage (MfSegmentation::performSegmentation, CPU half: SURVEY.md row a20) without a GPU: random edge
images, depths, instance masks, projected ids, model lists and parameters through FOUR implementations that must agree to the pixel --
  device : mf_k_segmentation_labels, the product's GPU label stage (mf_labels_gpu.hip) EXECUTED ON THE CPU (tests/hipcpu),
  host   : mf_segmentation_labels, the product's host form (mf_labels.hip),
  oracle : mfo_mf_segmentation_cpu (oracle/mf_oracle.c),
  ref    : oracle/_ref/libmf_seg.so, lines 219-523 of the reference's MfSegmentation.cpp compiled from its own text.
Two consecutive calls per case share the persistent ignore map.  Development tooling.   python tools/emu_fuzz_labels.py [n] [seed0]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipcpu"))
import numpy as np  # noqa: E402
from scipy import ndimage  # noqa: E402

import emu  # noqa: E402

L = emu.activate()
from oracle import mfo_mm, mfseg  # noqa: E402


def product(fn, W, H, binary, depth, mask, class_ids, proj, model_ids, model_cls, next_id, allow_new, prm, ignore):
    p = np.array([prm.threshold, prm.weightDistance, prm.weightConvexity, prm.morphEdgeIterations, prm.morphEdgeRadius, prm.morphMaskIterations,
                  prm.morphMaskRadius, prm.removeEdges, prm.minRelSizeNew, prm.maxRelSizeNew, prm.personClassID], np.float32)
    full = np.zeros((H, W), np.uint8)
    has_new, new_cls = C.c_int32(0), C.c_int32(-1)
    cid = np.ascontiguousarray(class_ids if len(class_ids) else [0], np.int32)
    mids, mcls = np.ascontiguousarray(model_ids, np.int32), np.ascontiguousarray(model_cls, np.int32)
    rc = fn(W, H, binary.ctypes.data, depth.ctypes.data, mask.ctypes.data, cid.ctypes.data, len(class_ids), proj.ctypes.data, mids.ctypes.data,
            mcls.ctypes.data, len(model_ids), next_id, int(allow_new), p.ctypes.data, ignore.ctypes.data, full.ctypes.data, C.byref(has_new), C.byref(new_cls))
    assert rc == 0, rc
    return full, bool(has_new.value), new_cls.value


def blobs(rng, W, H, n, lo, hi):
    """n random rectangles / discs, value k+1 for blob k (later ones overwrite)"""
    img = np.zeros((H, W), np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(n):
        cx, cy = rng.integers(0, W), rng.integers(0, H)
        rx, ry = rng.integers(lo, hi), rng.integers(lo, hi)
        if rng.integers(0, 2):
            img[max(0, cy - ry):cy + ry, max(0, cx - rx):cx + rx] = k + 1
        else:
            img[((xx - cx) / max(rx, 1)) ** 2 + ((yy - cy) / max(ry, 1)) ** 2 < 1] = k + 1
    return img


def one(seed):
    rng = np.random.default_rng(seed)
    W = int(rng.choice([64, 72, 96, 104, 160, 200, 264, 320]))
    H = int(rng.choice([48, 56, 88, 120, 152, 240]))
    # edges: random lines + noise; binary = 255 where NOT an edge
    edge = rng.random((H, W)) < rng.choice([0.0, 0.02, 0.1])
    for _ in range(int(rng.integers(2, 12))):
        if rng.integers(0, 2):
            edge[rng.integers(0, H), :] = True
        else:
            edge[:, rng.integers(0, W)] = True
    if rng.integers(0, 3) == 0:
        edge = ndimage.binary_dilation(edge, iterations=1)
    binary = np.where(edge, 0, 255).astype(np.uint8)
    depth = (1.0 + 2.0 * ndimage.gaussian_filter(rng.random((H, W)), 6)).astype(np.float32)
    depth[rng.random((H, W)) < 0.05] = 0.0
    n_masks = int(rng.integers(0, 6))
    mask = blobs(rng, W, H, int(rng.integers(0, n_masks)), 4, max(6, W // 4)) if n_masks else np.zeros((H, W), np.uint8)
    # Mask values WITHOUT a class id (>= n_masks, e.g. a 255 "ignore" label): upstream indexes classIDs[mask] unchecked there
    # (MfSegmentation.cpp:223,437,454) -- an out-of-bounds read whose result is whatever the heap holds (AddressSanitizer on the compiled
    # reference text confirms it) -- so such cases are compared among the three implementations of this repository only, which treat the
    # value as "no mask" (mf_labels.h, mask_id)
    undefined_upstream = bool(n_masks) and rng.integers(0, 4) == 0
    if undefined_upstream:
        mask[rng.integers(0, H), :] = 255
        mask[:, rng.integers(0, W)] = n_masks
    class_ids = [0] + [int(rng.choice([41, 42, 56, 63, 255])) for _ in range(max(0, n_masks - 1))] if n_masks else []
    n_models = int(rng.integers(1, 5))
    model_ids = [0] + sorted(rng.choice(np.arange(1, 12), n_models - 1, replace=False).tolist())
    model_cls = [-1] + [int(rng.choice([41, 42, 56, 63])) for _ in range(n_models - 1)]
    pb = blobs(rng, W, H, n_models - 1, 5, max(8, W // 5))
    proj = np.zeros((H, W), np.uint8)
    for k in range(1, n_models):
        proj[pb == k] = model_ids[k]
    next_id = int(max(model_ids) + 1)
    prm = mfo_mm.default_seg_params(threshold=0.3, weightDistance=150.0, weightConvexity=2.8, morphEdgeIterations=int(rng.integers(0, 3)),
                                    morphEdgeRadius=int(rng.integers(1, 3)), morphMaskIterations=int(rng.integers(0, 3)), morphMaskRadius=int(rng.integers(1, 3)),
                                    removeEdges=int(rng.integers(0, 2)), minRelSizeNew=float(rng.choice([0.004, 0.02, 0.07])), maxRelSizeNew=0.4)
    desc = f"seed {seed}: {W}x{H}{' (values without a class id)' if undefined_upstream else ''} masks={n_masks} models={model_ids} morph={prm.morphEdgeIterations}/{prm.morphMaskIterations} removeEdges={prm.removeEdges}"
    ign = {k: np.zeros((H, W), np.uint8) for k in ("device", "host", "oracle", "ref")}
    bad = []
    for call in range(2):
        allow = bool(rng.integers(0, 2))
        args = (binary, depth, mask, class_ids, proj, model_ids, model_cls, next_id, allow)
        out = {"device": product(L.mf_k_segmentation_labels, W, H, *args, prm, ign["device"]),
               "host": product(L.mf_segmentation_labels, W, H, *args, prm, ign["host"]),
               "oracle": mfo_mm.mf_segmentation_cpu(W, H, *args, ign["oracle"], prm)}
        if undefined_upstream:
            out["ref"], ign["ref"] = out["oracle"], ign["oracle"]
        else:
            out["ref"] = mfseg.mf_segmentation(W, H, *args, ign["ref"], prm)
        for k in ("device", "host", "oracle"):
            d = int((out[k][0] != out["ref"][0]).sum())
            if d or out[k][1:] != out["ref"][1:] or not np.array_equal(ign[k], ign["ref"]):
                bad.append((call, k, d, out[k][1:], out["ref"][1:], int((ign[k] != ign["ref"]).sum())))
        # second call: the scene moves a little, the ignore map persists
        mask = np.roll(mask, 2, axis=1)
        proj = np.roll(proj, 1, axis=0)
    return desc, bad


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    t0, nbad = time.time(), 0
    for s in range(seed0, seed0 + n_cases):
        desc, bad = one(s)
        if bad:
            nbad += 1
            print("DISAGREE", desc, bad, flush=True)
        else:
            print("ok      ", desc, flush=True)
    print(f"{n_cases} cases, {nbad} with disagreements, {time.time() - t0:.0f} s")
