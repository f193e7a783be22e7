import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from maskfusion_amd import MaskFusion
st, frames = bench.gen_frames(8)
dev = torch.device('cuda', 0)
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
mf = MaskFusion(bench.W, bench.H, bench.FX, bench.FY, bench.CX, bench.CY, icpThresh=100.0, so3=False, device=0, enableMultipleModels=False, numGSurfels=9437184)
order = bench.pingpong(8, 2000)
for i in range(50): mf.processFrameDevice(d_rgb[order[i]].data_ptr(), d_depth[order[i]].data_ptr())
mf.sync()
for n in (20, 100, 400):
    t0 = time.perf_counter()
    for i in range(n): mf.processFrameDevice(d_rgb[order[50+i]].data_ptr(), d_depth[order[50+i]].data_ptr())
    t1 = time.perf_counter()
    mf.sync()
    t2 = time.perf_counter()
    print(n, "enqueue us/frame", 1e6*(t1-t0)/n, "total us/frame", 1e6*(t2-t0)/n)
# steady-state stage times of the LAST frame of a free-running burst (events recorded, no per-frame sync)
mf.enableTimings(True)
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(200): mf.processFrameDevice(d_rgb[order[600+i]].data_ptr(), d_depth[order[600+i]].data_ptr())
    mf.sync(); t2 = time.perf_counter()
    print("timed burst us/frame", 1e6*(t2-t0)/200, {k: round(v*1e3,1) for k, v in mf.timings().items()})
mf.enableTimings(False)
