"""tooling: where does the one-launch clean pass (k_clean) spend its time on the 26.9 M-surfel map of configs[4]?  Needs the library built with
-DMF_CLEAN_PROF (tools/ab/libmaskfusion_amd_prof.so copied over maskfusion_amd/libmaskfusion_amd.so on the GPU box: tools/gpu_call_r05j.sh); every
chunk of the background's clean pass then leaves {start, ticket, sweep 1, look-back, sweep 2} in 100 MHz ticks.  Prints the phase statistics of the
last frame's launch and writes the raw table to gpurun_out/<tag>_clean_prof.npy."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from maskfusion_amd import MaskFusion, stress, synth
from maskfusion_amd import lib as mflib

tag = sys.argv[1] if len(sys.argv) > 1 else "prof"
W, H, F = stress.W, stress.H, stress.F
st = stress.stream(4)
mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=stress.NUM_GSURFELS, initConfidenceGlobal=10.0)
frames = [st.frame(k) for k in range(8)]
mf.processFrame(frames[0][0], frames[0][1], timestamp=0)
cap = stress.surfel_capacity(stress.NUM_GSURFELS)
cache = "/tmp/mf_room_single.npy"
if os.path.exists(cache):
    room = np.load(cache)
else:
    room = synth.dense_room_map(st.scene, int(1.005 * 0.8 * cap), last_time=1.0, furniture_above=0)
    np.save(cache, room)
mf.getBackgroundModel().uploadMap(room)
for k in range(1, 8):
    mf.processFrame(frames[k][0], frames[k][1], timestamp=k)
mf.sync()
n = mf.getBackgroundModel().lastCount()
chunks = (n + W * H // 4 + 2047) // 2048
L = C.CDLL(mflib.LIB_PATH)
buf = np.zeros((chunks, 8), np.uint32)
rc = L.mf_debug_clean_prof(C.c_void_p(buf.ctypes.data), C.c_int(chunks))
print("surfels", n, "chunks", chunks, "rc", rc)
os.makedirs("gpurun_out", exist_ok=True)
np.save(f"gpurun_out/{tag}_clean_prof.npy", buf)
buf = buf[buf[:, 1:5].sum(1) > 0]            # (the last chunk index may lie beyond the launch's chunks)
chunks = len(buf)
t0 = buf[:, 0].astype(np.int64)
t0 = (t0 - t0[0]) & 0xFFFFFFFF
t0 = np.where(t0 > (1 << 31), t0 - (1 << 32), t0)
t0 = t0 - t0.min()
ph = buf[:, 1:5].astype(np.float64) / 100.0          # us
names = ["ticket", "sweep1", "lookback", "sweep2"]
end = t0 / 100.0 + ph.sum(1)
print(f"span of the launch (first chunk start .. last chunk end): {end.max():.1f} us; workgroups that drew a chunk: {len(np.unique(buf[:, 5]))}")
tot = ph.sum(1)
print(f"per chunk: total mean {tot.mean():.1f} us  median {np.median(tot):.1f}  p90 {np.percentile(tot, 90):.1f}  max {tot.max():.1f}")
for i, nm in enumerate(names):
    v = ph[:, i]
    print(f"  {nm:9s} mean {v.mean():7.2f} us  median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  p99 {np.percentile(v, 99):7.2f}  max {v.max():7.2f}  share {v.sum() / tot.sum():.1%}")
slow = ph[:, 1] > 2.0 * np.median(ph[:, 1])
print(f"chunks with sweep 1 > 2 x median: {slow.mean():.1%}; their sweep 1 mean {ph[slow, 1].mean() if slow.any() else 0:.1f} us, the others' {ph[~slow, 1].mean():.1f} us")
for x in range(8):
    m = buf[:, 6] == x
    if m.any():
        print(f"  XCD {x}: {m.sum()} chunks, total mean {tot[m].mean():.1f} us")
# timeline: how many chunks are in flight over time, and the order of completion vs the order of the chunks
order = np.argsort(t0)
print("start time of chunk index quantiles (us):", [round(float(t0[int(q * (chunks - 1))]) / 100.0, 1) for q in (0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0)])
lag = (t0[1:] - t0[:-1]) / 100.0
print(f"start(i+1) - start(i): mean {lag.mean():.3f} us, p1 {np.percentile(lag, 1):.1f}, p99 {np.percentile(lag, 99):.1f}")
mf.close()
