"""The compiled C++ facade (include/maskfusion/MaskFusion.h, driven by tests/cpp/facade_main.cpp like GUI/MainController.cpp drives
the reference) against the Python mirror of the same C ABI: same frames, same settings -> identical poses, counts, model ids."""
import os
import subprocess

import numpy as np
import pytest

from test_abi import build_facade_exe

pytestmark = pytest.mark.gpu


def _python_run(frames, st, multi):
    from maskfusion_amd import MaskFusion
    m = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, numGSurfels=1 << 20, numOSurfels=1 << 18,
                   enableMultipleModels=multi, modelSpawnOffset=3, trackAllModels=False)
    for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                 ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004)):
        m.setParam(k, v)
    m.preallocateModels(1)
    out = {}
    for k, (rgb, depth, mask) in enumerate(frames):
        if multi:
            m.processFrame(rgb, depth, mask=mask, classIDs=[0, 41, 42], timestamp=k)
        else:
            m.processFrame(rgb, depth, timestamp=k)
        for mod in m.getModels():
            out[(k, mod.getID())] = (mod.getPose(), mod.lastCount())
    m.close()
    return out


@pytest.mark.parametrize("multi", [False, True], ids=["static", "multi"])
def test_cpp_facade_matches_python_mirror(hip, tmp_path, multi):
    from maskfusion_amd import synth
    n = 8 if multi else 3
    st = synth.Stream(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, n_objects=2 if multi else 0, noise=True, object_motion=0.0)
    frames = [st.frame(k) for k in range(n)]
    ref = _python_run(frames, st, multi)
    blob = tmp_path / "frames.bin"
    with open(blob, "wb") as f:
        for rgb, depth, mask in frames:
            f.write(np.ascontiguousarray(rgb, np.uint8).tobytes())
            f.write(np.ascontiguousarray(depth, np.float32).tobytes())
            f.write(np.ascontiguousarray(mask if mask is not None else np.zeros((st.H, st.W)), np.uint8).tobytes())
    outdir = str(tmp_path) + os.sep
    exe = build_facade_exe(tmp_path, lib=hip._name if os.environ.get("MF_EMU") == "1" else None)
    r = subprocess.run([exe, "640", "480", "528", "528", "320", "240", str(n), str(blob), outdir, "1" if multi else "0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    got, counts, born = {}, {}, []
    for line in r.stdout.splitlines():
        w = line.split()
        if w[0] == "pose":
            got[(int(w[1]), int(w[2]))] = np.array(list(map(float, w[3:])), np.float64).reshape(4, 4).T
        elif w[0] == "count":
            counts[(int(w[1]), int(w[2]))] = int(w[3])
        elif w[0] == "new":
            born.append(int(w[1]))
    assert set(got) == set(ref), (sorted(got), sorted(ref))
    for key, (pose, cnt) in ref.items():
        assert np.array_equal(got[key].astype(np.float32), pose.astype(np.float32)), key      # same library, same inputs: bit-identical
        assert counts[key] == cnt, key
    ids = sorted({i for (_, i) in ref})
    assert born == [i for i in ids if i != 0]                # one new-model callback per spawned object, none for the background
    if multi:
        assert len(ids) >= 2, "the scenario must spawn an object model"
        segs = [f for f in os.listdir(outdir) if f.startswith("Segmentation") and f.endswith(".png")]
        assert len(segs) == n - 1                             # exportSegmentationResults: every frame but the init frame
        from PIL import Image
        img = np.asarray(Image.open(os.path.join(outdir, sorted(segs)[-1])))
        assert img.shape == (480, 640) and img.max() < 255 and set(np.unique(img)) <= set(ids)
    tail = r.stdout.splitlines()[-3:]
    assert tail[0].startswith("tick %d models" % (n + 1)) and tail[1].startswith("map ") and tail[2].startswith("manual %d" % (n + 2))
    assert os.path.exists(os.path.join(outdir, "poses-0.txt")) and os.path.exists(os.path.join(outdir, "cloud-0.ply"))
