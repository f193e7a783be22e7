"""Bit-level trace of the product on the scenario of tests/test_gpu_switches.py::test_static_switches_vs_oracle: per frame, hashes of the filtered
depth, the frame maps, the model-side maps, poses and surfel counts -- run in two trees, diff the outputs to find the first kernel whose bits moved.
    PYTHONPATH=<tree> python tools/state_dump.py > out.txt"""
import hashlib
import sys
import numpy as np
sys.path.insert(0, "tests")
from maskfusion_amd import MaskFusion, synth

SEG_D = dict(mfThreshold=0.1, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0, newModelMinRelativeSize=0.002)
try:
    import test_gpu_switches as T
    SEG_D = T.SEG_D
except Exception as e:  # noqa
    print("using fallback SEG_D", e)
W, H, f = 640, 480, 528.0
st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=1, noise=True, object_motion=0.0)
mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 20, numOSurfels=1 << 17, enableMultipleModels=True, modelSpawnOffset=2, trackAllModels=False)
for k, v in SEG_D.items():
    mf.setParam(k, v)
h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]
for k in range(8):
    rgb, d, m = st.frame(k)
    ms = mf.getModels()
    if k == 6:
        ms[1].makeNonStatic()
    mf.processFrame(rgb, d, mask=m, classIDs=[0, 41], timestamp=k)
    ms = mf.getModels()
    line = [f"frame {k}", "depthF", h(mf.debugRead("depthF"))]
    for t in ("vmap0", "nmap0", "vmap1", "nmap1", "vmap2", "nmap2"):
        line += [t, h(mf.debugRead(t))]
    for i, x in enumerate(ms):
        line += [f"m{i}", h(x.getPose()), str(x.lastCount())]
        for t in ("vmap_g0", "nmap_g0", "vmap_g2", "nmap_g2", "pred_vertex"):
            try:
                line += [t, h(mf.debugRead(t, model=i))]
            except Exception as e:
                line += [t, "n/a"]
    line += ["seg", h(mf.downloadSegmentation())]
    print(" ".join(line))
mf.close()
