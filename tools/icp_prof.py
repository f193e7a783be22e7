import sys, numpy as np
sys.path.insert(0,'.')
import torch
from maskfusion_amd import MaskFusion
import bench
st, frames = bench.gen_frames(bench.CONFIGS["1"], 12, 1)
mf = MaskFusion(640,480,528,528,320,240, icpThresh=100.0, so3=False, enableMultipleModels=False)
mf.setParam("icpProfile", 1)
acc=[]
for k in list(range(12))+list(range(10,0,-1)):
    mf.processFrame(frames[k][0], frames[k][1])
    if mf.getTick()>3:
        acc.append(mf.debugRead("icp_prof").astype(np.int64))
a=np.array(acc)  # frames x 19 x 8
d=np.diff(a,axis=2).mean(0)  # 19 x 7
print("phase deltas in shader-clock ticks (s_memtime); cols: issue-loads, reduce, solve, sync, pixels+gathers, wave-reduce, store")
np.set_printoptions(linewidth=200, suppress=True)
print(np.round(d).astype(int))
print("total ticks per launch", np.round((a[:,:,7]-a[:,:,0]).mean(0)).astype(int))
