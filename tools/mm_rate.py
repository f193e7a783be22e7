"""Multi-model frame rate (standing objects, host label stage included) -- where a multi-model frame spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from maskfusion_amd import MaskFusion, synth

SEG = dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0,
           newModelMinRelativeSize=0.004)
st = synth.Stream(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, n_objects=2, noise=True, object_motion=0.0)
frames = [st.frame(k) for k in range(24)]
m = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, numGSurfels=1 << 20, numOSurfels=1 << 18,
               enableMultipleModels=True, modelSpawnOffset=3, trackAllModels=False)
for k, v in SEG.items():
    m.setParam(k, v)
if len(sys.argv) > 1:   # A/B: python tools/mm_rate.py earlyBackgroundFusion=0
    for kv in sys.argv[1:]:
        key, val = kv.split("=")
        m.setParam(key, float(val))
cls = (0, 41, 42)
for k in range(12):
    m.processFrame(frames[k][0], frames[k][1], mask=frames[k][2], classIDs=cls)
m.enableTimings(True)
acc, n = {}, 0
t0 = time.perf_counter()
for rep in range(3):
    for k in list(range(12, 24)) + list(range(22, 11, -1)):
        m.processFrame(frames[k][0], frames[k][1], mask=frames[k][2], classIDs=cls)
        for kk, vv in m.timings().items():
            acc[kk] = acc.get(kk, 0.0) + vv
        n += 1
dt = time.perf_counter() - t0
print("models", [(x.getID(), x.lastCount()) for x in m.getModels()])
print(f"{n} frames, {1e3 * dt / n:.3f} ms/frame wall (host-pointer API incl. H2D), GPU stage ms:", {k: round(v / n, 3) for k, v in acc.items()})
