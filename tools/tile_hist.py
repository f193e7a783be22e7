"""Distribution of stable surfels over the 16x16-pixel tiles of the splat prediction (load balance of k_splat_tile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from maskfusion_amd import MaskFusion
st, frames = bench.gen_frames(24)
mf = MaskFusion(bench.W, bench.H, bench.FX, bench.FY, bench.CX, bench.CY, icpThresh=100.0, so3=False, enableMultipleModels=False)
for k in bench.pingpong(24, 200):
    mf.processFrame(frames[k][0], frames[k][1])
m = mf.getBackgroundModel().downloadMap()
T = mf.getCurrPose()
Ti = np.linalg.inv(T)
stable = m[:, 3] >= mf.getBackgroundModel().getConfidenceThreshold()
p = (Ti[:3, :3] @ m[stable, :3].T).T + Ti[:3, 3]
ok = p[:, 2] > 0
u = bench.FX * p[ok, 0] / p[ok, 2] + bench.CX
v = bench.FY * p[ok, 1] / p[ok, 2] + bench.CY
ins = (u >= 0) & (u < bench.W) & (v >= 0) & (v < bench.H)
tx, ty = (u[ins] // 16).astype(int), (v[ins] // 16).astype(int)
h = np.bincount(ty * 40 + tx, minlength=1200)
print("surfels", len(m), "stable", int(stable.sum()), "in view", int(ins.sum()))
print("per tile: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %d" % (h.mean(), np.median(h), np.percentile(h, 90), np.percentile(h, 99), h.max()))
r = m[stable, 11][ok][ins]
z = p[ok, 2][ins]
print("sprite side px (2*sqrt2*r*f/z): mean %.1f p90 %.1f max %.1f" % tuple(np.percentile(2 * 1.414 * r * bench.FX / z, q) if q else (2 * 1.414 * r * bench.FX / z).mean() for q in (0, 90, 100)))
