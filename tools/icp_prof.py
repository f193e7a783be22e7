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
a=np.array(acc)  # frames x 19 x 16
main=a[:,:,:8]
d=np.diff(main,axis=2).mean(0)  # 19 x 7
print("phase deltas in shader-clock ticks (s_memtime, 100 MHz-domain ticks as reported); cols: issue-loads, reduce, solve, sync, pixels+gathers, block-reduce, store")
np.set_printoptions(linewidth=200, suppress=True)
print(np.round(d).astype(int))
print("total ticks per launch", np.round((main[:,:,7]-main[:,:,0]).mean(0)).astype(int))
# inside the solve (launches 1..18): stamp[2] = after reduce_partials; 8 after the system is in registers, 9 after LDL^T, 10 after exp + composition,
# 11 after the wavefront exchange and the state writes (round 3: three barrier-separated LDS steps, stamps 11 / 12 / 13); stamp[3] = back in the kernel
s=a[:,1:,:]
seq=np.stack([s[:,:,2], s[:,:,8], s[:,:,9], s[:,:,10], s[:,:,11], s[:,:,12], s[:,:,13], s[:,:,3]], axis=2)
print("inside the solve; cols: unpack, LDL^T, exp+compose, shuffle exchange + state writes, -, -, return + s_pose tail")
print(np.round(np.diff(seq,axis=2).mean(0)).astype(int))
