"""How uneven are the tile lists of the tiled prediction?  Runs the configs[1] stream for N frames on the GPU, downloads the background map and
pose, bins the stable surfels' sprite boxes into 16x16 tiles on the host (centre +- radius: an approximation of splat.vert's box) and prints the
distribution of entries per tile -- the tile pass is one workgroup per tile, so its duration follows the LONGEST list, not the mean."""
import sys
import numpy as np
from maskfusion_amd import MaskFusion, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
W, H, F = 640, 480, 528.0
st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, n_objects=0, noise=True)
mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 22)
for k in range(N):
    rgb, depth, _ = st.frame(k)
    mf.processFrame(rgb, depth)
m = mf.getModels()[0].downloadMap()
T = mf.getCurrPose()
mf.close()
conf = m[:, 3]
stable = conf >= 10.0 if False else conf >= 4.0
p = m[stable, :3]
r = m[stable, 11]
Ti = np.linalg.inv(T)
h = p @ Ti[:3, :3].T + Ti[:3, 3]
ok = h[:, 2] > 0.05
h, r = h[ok], r[ok]
u = F * h[:, 0] / h[:, 2] + W / 2.0
v = F * h[:, 1] / h[:, 2] + H / 2.0
half = np.clip(F * r * 1.41421356 / h[:, 2], 0.5, 32.0)
x0 = np.clip(np.ceil(u - half - 0.5), 0, W - 1).astype(int); x1 = np.clip(np.ceil(u + half - 0.5) - 1, 0, W - 1).astype(int)
y0 = np.clip(np.ceil(v - half - 0.5), 0, H - 1).astype(int); y1 = np.clip(np.ceil(v + half - 0.5) - 1, 0, H - 1).astype(int)
vis = (u >= 0) & (u <= W) & (v >= 0) & (v <= H) & (x0 <= x1) & (y0 <= y1)
cnt = np.zeros((H // 16, W // 16), np.int64)
area = np.zeros_like(cnt)
for a0, a1, b0, b1 in zip(x0[vis] // 16, x1[vis] // 16, y0[vis] // 16, y1[vis] // 16):
    cnt[b0:b1 + 1, a0:a1 + 1] += 1
c = cnt.reshape(-1)
print(f"frames {N}  surfels {len(m)}  stable+visible {int(vis.sum())}  entries {int(c.sum())}  tiles {len(c)}")
print("entries per tile: mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %d  empty %d" % (c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max(), int((c == 0).sum())))
print("box side px: mean %.1f  p90 %.1f  max %.1f" % ((2 * half[vis]).mean(), np.percentile(2 * half[vis], 90), (2 * half[vis]).max()))
print("rows of tiles (sum of entries per tile row):", cnt.sum(1).tolist())
