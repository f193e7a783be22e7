"""Race check without a GPU: the same frames through the CPU-executed kernels (tests/hipcpu) under the forward and the REVERSED thread /
workgroup schedule (HIPCPU_SCHEDULE=reverse).  A kernel whose result depends on the order in which the threads of a workgroup or the
workgroups of a launch run -- a missing barrier, an unordered float atomic, a compaction that is not order-stable -- shows up as
different bits.  Development tooling.   python tools/emu_schedule_check.py   (MF_BIG_FORMS=1: with the full-map forms of the fuse / clean passes forced)"""
import os
import pickle
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, pickle, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests", "hipcpu"))
import numpy as np
import emu
emu.activate()
from maskfusion_amd import MaskFusion, synth
out = {}
# (1) single model, ICP + photometric + SO(3), 6 frames
W, H = 240, 160
f = 528.0 * W / 640.0
st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=20.0, so3=True, enableMultipleModels=False, numGSurfels=1 << 17)
if os.environ.get("MF_BIG_FORMS") == "1":   # the passes of full maps (run table + culling, in-place update, in-place clean) on these small ones
    mf.setParam("bigMapElements", 0); mf.setParam("inPlaceElements", 0)
for k in range(6):
    rgb, d, _ = st.frame(k)
    mf.processFrame(rgb, d, timestamp=k)
out["single"] = dict(pose=mf.getCurrPose(), cloud=mf.getBackgroundModel().downloadMap(), pred=mf.debugRead("pred_vertex"))
mf.close()
# (2) multi-model with moving, tracked objects (batched Gauss-Newton loop, label stage, spawn), 9 frames
st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=2, noise=True, object_motion=1.0)
mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 17, numOSurfels=1 << 15, enableMultipleModels=True,
                modelSpawnOffset=2, trackAllModels=True)
for k, v in dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=1, mfMorphMaskIterations=1, newModelMinRelativeSize=0.004).items():
    mf.setParam(k, v)
if os.environ.get("MF_BIG_FORMS") == "1":
    mf.setParam("bigMapElements", 0); mf.setParam("inPlaceElements", 0)
segs = []
for k in range(9):
    rgb, d, mask = st.frame(k)
    mf.processFrame(rgb, d, mask=mask, classIDs=[0, 41, 42], timestamp=k)
    segs.append(mf.downloadSegmentation())
ms = mf.getModels()
out["multi"] = dict(ids=[m.getID() for m in ms], poses=[m.getPose() for m in ms], clouds=[m.downloadMap() for m in ms], segs=segs)
mf.close()
pickle.dump(out, open(sys.argv[1], "wb"))
'''


def run(schedule, path):
    env = dict(os.environ)
    env.pop("HIPCPU_SCHEDULE", None)
    if schedule:
        env["HIPCPU_SCHEDULE"] = schedule
    subprocess.run([sys.executable, "-c", WORKER % dict(root=ROOT), path], check=True, env=env)
    return pickle.load(open(path, "rb"))


def same(a, b, path=""):
    import numpy as np
    if isinstance(a, dict):
        return all(same(a[k], b[k], path + "/" + str(k)) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y, f"{path}[{i}]") for i, (x, y) in enumerate(zip(a, b)))
    ok = np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)
    if not ok:
        print("DIFFERENT:", path)
    return ok


if __name__ == "__main__":
    with tempfile.TemporaryDirectory() as d:
        fwd = run(None, os.path.join(d, "fwd.pkl"))
        rev = run("reverse", os.path.join(d, "rev.pkl"))
    print("models:", fwd["multi"]["ids"], "surfels:", [len(c) for c in fwd["multi"]["clouds"]], "background:", len(fwd["single"]["cloud"]))
    print("bit-identical under both schedules" if same(fwd, rev) else "SCHEDULE-DEPENDENT RESULTS (see above)")
