"""Untraced A/B of the two boundaries of one context (configs[1], VGA, one model): device-resident frames (mf_process_frame_dev) against
host frames (mf_process_frame: staging copy + one packed upload on the input stream), alternated so that the map's growth cancels:
frames/s by wall clock, and the GPU's per-stage times of the last frame of a burst ("timings": HIP events on the library's stream,
eager launches).  usage: python tools/host_stage_times.py [bursts] [short]  -> one JSON object on stdout"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = bench.CONFIGS["1"]
st, frames = bench.gen_frames(cfg, 48)

import torch  # noqa: E402
from maskfusion_amd import MaskFusion  # noqa: E402

dev = torch.device("cuda", 0)
W, H, F = cfg["W"], cfg["H"], cfg["f"]
mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, device=0, enableMultipleModels=False, numGSurfels=cfg["surfels"])
d_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
d_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]
order = bench.pingpong(len(frames), 1 << 16)
pos = [0]


def burst(kind, n):
    for _ in range(n):
        k = order[pos[0]]; pos[0] += 1
        if kind == "device":
            mf.processFrameDevice(d_rgb[k].data_ptr(), d_depth[k].data_ptr())
        else:
            mf.processFrame(frames[k][0], frames[k][1])


def timed(kind, n=300):
    burst(kind, 12)
    mf.sync()
    t0 = time.perf_counter()
    burst(kind, n)
    mf.sync()
    return 1e6 * (time.perf_counter() - t0) / n


def stages(kind, n=120):
    mf.enableTimings(True)
    burst(kind, n)
    mf.sync()
    t = {k: round(1e3 * v, 1) for k, v in mf.timings().items() if v}
    mf.enableTimings(False)
    return t


burst("host", 400)        # the map near its natural size
mf.sync()
out = {"us_per_frame": [], "stage_us_last_frame_of_a_burst": []}
modes = [("device", {}), ("host", {}), ("host", {"hostWaitUpload": 0}), ("host", {"frameGraph": 1}), ("host", {"hostCopyHelper": 0}),
         ("host", {"hostWaitUpload": 0, "hostCopyHelper": 0}), ("host", {"hostLockstep": 0}), ("host", {"hostUploadOnMain": 1}), ("host", {"hostInputAsync": 0})]
if len(sys.argv) > 2 and sys.argv[2] == "short":
    modes = modes[:7]
DEFAULTS = {"hostUploadOnMain": 0, "frameGraph": 0, "hostInputAsync": 1, "hostUploadAfterTracking": 0, "hostLockstep": 1, "hostUploadKernel": 0, "hostWaitUpload": 1,
            "hostCopyHelper": 1}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for kind, params in modes:
        for k, v in params.items():
            mf.setParam(k, v)
        out["us_per_frame"].append({"boundary": kind, "params": params, "rep": rep, "us": round(timed(kind), 1)})
        if rep == 0 and not params.get("hostInputAsync", 1) == 0 and not (len(sys.argv) > 2 and sys.argv[2] == "short"):
            out["stage_us_last_frame_of_a_burst"].append({"boundary": kind, "params": params, "stages": stages(kind)})
        for k in params:
            mf.setParam(k, DEFAULTS[k])
out["surfels"] = sum(m.lastCount() for m in mf.getModels())
print(json.dumps(out))
