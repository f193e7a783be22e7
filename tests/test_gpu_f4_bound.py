"""Finding F4 with a stated domain: the device solves the 6x6 Gauss-Newton system with an unpivoted LDL^T on fp64 sums, the
reference (RGBDOdometry.cpp:447-459) hands float-rounded sums to Eigen::LDLT (diagonal pivoting).  For a well-posed system the two are the
same step to rounding; for a rank-deficient one both divide rounding noise by rounding noise and may disagree arbitrarily.  The domain:

    every iteration of the step has >= 6 inliers and smallest pivot >= 1e-8 x largest diagonal entry (=> cond(A) <= 1e8)

Both sides COUNT the iterations outside it (mf_get_gn_condition / mfo_last_track_ill) instead of using them silently:
  * inside the domain (the noisy S1 stream): the counts are 0 on both sides on every frame and the poses agree per frame;
  * outside it (a frame that leaves a 6 x 6-pixel patch of depth): both sides flag the step.
The solver itself against the pivoted restatement, numpy and SciPy on systems up to cond 1e10: tests/test_gpu_kernels.py::test_gn_solve_update."""
import numpy as np
import pytest

from gpu_util import scene_frames

pytestmark = pytest.mark.gpu


def _pair(oracle, st):
    from maskfusion_amd import MaskFusion
    cap = 1 << 20
    o = oracle.Oracle(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpWeight=100.0, capacity=cap, so3=0)
    m = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, numGSurfels=cap, enableMultipleModels=False)
    return o, m


def test_inside_the_domain_both_sides_agree_and_flag_nothing(hip, oracle):
    st, frames = scene_frames(12, noise=True)
    o, m = _pair(oracle, st)
    for k, (rgb, depth, _) in enumerate(frames):
        o.process_frame(rgb, depth)
        m.processFrame(rgb, depth, timestamp=k)
        if k == 0:
            continue
        g_ill, o_ill = m.gnIllIterations(0), oracle.lib().mfo_last_track_ill()
        d = float(np.linalg.norm(m.getCurrPose()[:3, 3] - o.pose[:3, 3]))
        log = m.debugRead("icp_log")
        print(k, "ill iterations hip / oracle", g_ill, o_ill, "min inliers", int(log[:, 28].min()), "|dt| %.2e" % d)
        assert g_ill == 0 and o_ill == 0, k
        assert log[:, 28].min() >= 6
        assert d < 1e-4, k          # inside the domain the step is the reference's to rounding
    o.close(); m.close()


def test_outside_the_domain_both_sides_flag_the_step(hip, oracle):
    st, frames = scene_frames(4, noise=True)
    o, m = _pair(oracle, st)
    flagged = []
    for k, (rgb, depth, _) in enumerate(frames):
        if k == 2:
            # the sensor loses everything but a 6 x 6-pixel patch: no valid 4x4 cell at the coarse levels, 36 coplanar points at level 0
            d2 = np.zeros_like(depth)
            d2[240:246, 320:326] = depth[240:246, 320:326]
            depth = d2
        o.process_frame(rgb, depth)
        m.processFrame(rgb, depth, timestamp=k)
        if k == 0:
            continue
        g_ill, o_ill = m.gnIllIterations(0), oracle.lib().mfo_last_track_ill()
        log = m.debugRead("icp_log")
        print(k, "ill iterations hip / oracle", g_ill, o_ill, "inliers per iteration", log[:, 28].astype(int).tolist())
        flagged.append((k, g_ill, o_ill))
        if k == 1:
            assert g_ill == 0 and o_ill == 0
        if k == 2:
            assert g_ill > 0 and o_ill > 0, "a step on 36 coplanar points must be flagged on both sides"
            assert (g_ill == 19) == (o_ill == 19)     # either every iteration of the step, on both sides, or the same judgement of the finer ones
    o.close(); m.close()
    assert any(g for _, g, _ in flagged)
