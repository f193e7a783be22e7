"""Per-phase shader-clock stamps of the tiled prediction's tile pass (k_splat_tile, thread 0 of every tile workgroup; mf_set_param
"splatProfile"): where does one tile workgroup spend its time, and how does that follow the tile's list length?  Run on the GPU:
    PYTHONPATH=. python tools/splat_prof.py [frames]
phases: setup (rays + count + barrier) | list loop of wave 0 | barrier (waiting for the other waves) | outputs issued | stores drained"""
import sys
import numpy as np
from maskfusion_amd import MaskFusion, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H, F = 640, 480, 528.0
st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, n_objects=0, noise=True)
mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 22)
mf.setParam("splatProfile", 1)
acc = []
for k in range(N):
    rgb, depth, _ = st.frame(k)
    mf.processFrame(rgb, depth)
    if k >= N - 20:
        acc.append(mf.debugRead("splat_prof").astype(np.int64))
mf.close()
a = np.array(acc)                      # frames x tiles x 8
cnt = a[:, :, 6]
t = a[:, :, :6]
d = np.diff(t, axis=2)                 # frames x tiles x 5
start = t[:, :, 0] - t[:, :, 0].min(axis=1, keepdims=True)
end = t[:, :, 5] - t[:, :, 0].min(axis=1, keepdims=True)
np.set_printoptions(linewidth=200, suppress=True)
print("frames", N, " list length per tile: mean %.0f  max %d" % (cnt.mean(), cnt.max()))
names = ["setup", "list loop (wave 0)", "barrier wait", "outputs issued", "stores drained"]
for i, n in enumerate(names):
    print(f"  {n:20s} mean {d[:, :, i].mean():9.0f}  p90 {np.percentile(d[:, :, i], 90):9.0f}  max {d[:, :, i].max():9.0f}  ticks")
print("  workgroup lifetime   mean %9.0f  p90 %9.0f  max %9.0f" % ((t[:, :, 5] - t[:, :, 0]).mean(), np.percentile(t[:, :, 5] - t[:, :, 0], 90), (t[:, :, 5] - t[:, :, 0]).max()))
print("  workgroup START after the first one: mean %9.0f  p90 %9.0f  max %9.0f" % (start.mean(), np.percentile(start, 90), start.max()))
print("  launch span (last end - first start): mean %9.0f ticks" % end.max(axis=1).mean())
c = cnt.reshape(-1); l = d[:, :, 1].reshape(-1)
for lo, hi in ((0, 100), (100, 300), (300, 600), (600, 1000), (1000, 100000)):
    m = (c >= lo) & (c < hi)
    if m.any():
        print(f"  list {lo:5d}..{hi:6d}: tiles {int(m.sum()):6d}  loop ticks mean {l[m].mean():9.0f}  per entry {l[m].sum() / max(1, c[m].sum()):7.1f}")
