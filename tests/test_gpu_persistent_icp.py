"""EXPERIMENTAL, opt-in (MF_TEST_PERSISTENT=1): the geometric Gauss-Newton loop as ONE persistent launch with device-wide barriers
(mf_set_param("persistentIcp", 1); mf_odometry.hip, k_icp_persist) against the launch-per-iteration loop it would replace.  Not part of
the default -m gpu run: the kernel was written at the end of round 2 with no GPU time left to run it.  Run it first thing, under a
timeout:   MF_TEST_PERSISTENT=1 timeout 300 python -m pytest tests/test_gpu_persistent_icp.py -x -q -s
Both loops do the same per-pixel arithmetic, the same fixed-order fp64 reduction and the same solve; they group pixels into workgroups
differently (240 slices of every level vs the per-level grids), so the fp32 block sums differ in the last bits: poses must agree to
1e-5, inlier counts exactly.
Without a GPU the kernel's LOGIC (not its speed, not the hardware's memory model) can be exercised on the CPU-executed build with all
240 workgroups resident, one OS thread each:
    MF_EMU=1 MF_EMU_COOP=1 HIPCPU_COOPERATIVE=k_icp_persist MF_TEST_PERSISTENT=1 python -m pytest tests/test_gpu_persistent_icp.py -x -q -s"""
import os
import time

import numpy as np
import pytest

from gpu_util import EMU

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("MF_TEST_PERSISTENT") != "1", reason="opt-in: MF_TEST_PERSISTENT=1 (unvalidated experimental kernel)")]


def _run(persistent, frames, st, n):
    from maskfusion_amd import MaskFusion
    mf = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 20)
    mf.setParam("persistentIcp", 1.0 if persistent else 0.0)
    poses, stats = [], []
    for k in range(n):
        mf.processFrame(frames[k][0], frames[k][1], timestamp=k)
        poses.append(mf.getCurrPose().copy())
        stats.append(mf.trackStats(0))
    # timing: 100 more frames over the same data, inputs through the host-pointer API on both sides
    mf.enableTimings(True)
    t0 = time.perf_counter()
    acc = 0.0
    reps = 2 if EMU else 100
    for rep in range(reps):
        k = n - 1 - (rep % 2)
        mf.processFrame(frames[k][0], frames[k][1], timestamp=n + rep)
        acc += mf.timings().get("icpIterations", 0.0)
    dt = time.perf_counter() - t0
    mf.close()
    return np.array(poses), stats, acc / reps, dt / reps


@pytest.mark.parametrize("W,H", [(320, 240), (640, 480)])
def test_persistent_loop_equals_launch_per_iteration(hip, W, H):
    from maskfusion_amd import synth
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    n = 5 if EMU else 12
    frames = [st.frame(k) for k in range(n)]
    pa, sa, ms_a, wall_a = _run(False, frames, st, n)
    pb, sb, ms_b, wall_b = _run(True, frames, st, n)
    print(f"\n{W}x{H}: 19 iterations  launch-per-iteration {1e3 * ms_a:.1f} us  persistent {1e3 * ms_b:.1f} us   "
          f"(frame wall {1e3 * wall_a:.3f} vs {1e3 * wall_b:.3f} ms);  max pose diff {np.abs(pa - pb).max():.2e}")
    assert np.abs(pa - pb).max() < 1e-5
    for x, y in zip(sa[1:], sb[1:]):
        assert x["lastICPCount"] == y["lastICPCount"]
