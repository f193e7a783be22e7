"""Kernel LOGIC on every CPU run: a few frames through the product's own kernels executed on the CPU (tests/hipcpu: the .hip sources
compiled with g++ against a HIP stand-in, fibers for threads, 64-wide wavefronts) next to the oracle.  This is test tooling, not a
product path -- it runs in a subprocess so that nothing of it leaks into this process; the product library is the HIP build only, and
the parity claims rest on the -m gpu runs on MI355X (which MF_EMU=1 can rehearse here in full, see tests/gpu_util.py)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulated_kernels_track_and_fuse_like_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipcpu", "smoke.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for k, fr in enumerate(res["single"]):                       # geometric tracking: one launch per Gauss-Newton iteration
        assert fr["count"] == fr["ocount"], (k, fr)              # fused surfel count exact, every frame
        assert fr["pose_diff"] < 2e-5, (k, fr)
        assert k == 0 or fr["inliers"] > 10000
    for k, fr in enumerate(res["rgbd"]):                         # ICP + photometric term + SO(3) pre-alignment
        assert abs(fr["count"] - fr["ocount"]) <= max(5, fr["ocount"] // 500), (k, fr)
        assert fr["pose_diff"] < 1e-4, (k, fr)
        assert k == 0 or (fr["rgb_count"] > 0 and fr["so3_iterations"] >= 1)
