"""The six model-side maps and the pose of the background after three frames of the scenario of tools/state_dump.py, saved as an .npz -- run in two
trees (or with switches: `key=value,...` as the second argument), compare with tools/state_cmp.py.
    PYTHONPATH=<tree> python tools/state_dump2.py out.npz [fusedPreprocessLaunch=0,...]"""
import sys
import numpy as np
from maskfusion_amd import MaskFusion, synth
W, H, f = 640, 480, 528.0
st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=1, noise=True, object_motion=0.0)
mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 20, numOSurfels=1 << 17, enableMultipleModels=True, modelSpawnOffset=2, trackAllModels=False)
if len(sys.argv) > 2:
    for kv in sys.argv[2].split(","):
        mf.setParam(kv.split("=")[0], float(kv.split("=")[1]))
out = {}
for k in range(3):
    rgb, d, m = st.frame(k)
    mf.processFrame(rgb, d, mask=m, classIDs=[0, 41], timestamp=k)
for t in ("vmap_g0", "nmap_g0", "vmap_g1", "nmap_g1", "vmap_g2", "nmap_g2"):
    out[t] = mf.debugRead(t, model=0)
out["pose"] = mf.getCurrPose()
np.savez(sys.argv[1], **out)
mf.close()
