"""How much does the fp32 trip count of the association / clean window loops matter?  (CPU only, oracle only.)

data.vert:139-141 and copy_unstable.vert:85-86 walk their 4x4 half-pixel window with an fp32 induction variable; in exact
arithmetic that is 4 steps per axis, in fp32 it is 4 or 5 depending on the rounding of the centre coordinate.  Round 1 used the
exact-arithmetic reading on both sides; since round 2 the clean pass (where every tap counts) uses the literal one, oracle and device.  This tool fuses the same synthetic stream with the camera poses GIVEN (so only
the surfel life cycle differs) under both readings and reports what changes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskfusion_amd import synth  # noqa: E402
from oracle import mfo  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
W, H, F = 640, 480, 528.0
st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, noise=True)
frames = [st.frame(k) for k in range(N)]
L = mfo.lib()
out = {}
for literal in (0, 1):
    L.mfo_set_window_literal(literal)
    L.mfo_set_clean_literal(literal)
    o = mfo.Oracle(W, H, F, F, W / 2.0, H / 2.0, icpWeight=100.0, capacity=1 << 20, so3=0, confGlobal=2.0)
    counts = []
    for k, (rgb, depth, _) in enumerate(frames):
        o.process_frame(rgb, depth, in_pose=st.gt_pose(k).astype(np.float32) if k else None)
        counts.append(o.count)
    s = o.surfels()
    out[literal] = dict(counts=counts, stable=int((s[:, 3] > 2.0).sum()), conf=float(s[:, 3].mean()))
    o.close()
L.mfo_set_window_literal(0)
L.mfo_set_clean_literal(1)
a, b = out[0], out[1]
print("frames", N)
print("exact-arithmetic window (4x4 taps):     surfels", a["counts"][-1], "stable (conf > 2)", a["stable"], "mean confidence %.3f" % a["conf"])
print("literal fp32 window (4..5 taps / axis): surfels", b["counts"][-1], "stable (conf > 2)", b["stable"], "mean confidence %.3f" % b["conf"])
print("relative difference in surfel count per frame:", [round((y - x) / x, 4) for x, y in zip(a["counts"], b["counts"])][::max(1, N // 8)])
