#!/usr/bin/env python3
"""Model list over time on the S2 stream (ids, surfel counts, ICP inliers): a scenario sanity check for bench.py --config 2s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from maskfusion_amd import MaskFusion, synth

n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 8
motion = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nfr = int(sys.argv[4]) if len(sys.argv) > 4 else 60
st = synth.Stream(noise=True, n_objects=n_obj, object_motion=motion)
mf = MaskFusion(640, 480, 528., 528., 320., 240., icpThresh=100.0, so3=False, enableMultipleModels=True, numGSurfels=1 << 21, numOSurfels=1 << 19,
                trackAllModels=True, modelSpawnOffset=2, initConfidenceGlobal=10.0, initConfidenceObject=0.01)
for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
             ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004), ("batchTracking", batch)):
    mf.setParam(k, v)
cls = [0] + [41 + i for i in range(n_obj)]
for k in range(nfr):
    rgb, depth, mask = st.frame(k)
    mf.processFrame(rgb, depth, mask=mask, classIDs=cls, timestamp=k)
    ms = mf.getModels()
    seg = mf.downloadSegmentation()
    d = np.linalg.norm(mf.getCurrPose()[:3, 3] - st.gt_pose(k)[:3, 3])
    print(k, "ids", [m.getID() for m in ms], "cls", [m.getClassID() for m in ms], "n", [m.lastCount() for m in ms], "inl", [int(m.getICPStats()[1]) for m in ms],
          "seg ids", sorted(set(np.unique(seg).tolist())), "drift %.4f" % d)
