"""-m gpu: how faithful is the CPU-executed test build (tests/hipcpu) to the hardware?  The same small stream through
maskfusion_amd/libmaskfusion_amd.so on the MI355X (this process) and through the same sources compiled with g++ and executed on the CPU
(a subprocess: tests/hipcpu/smoke.py; tests/_emu/libmaskfusion_emu.so travels to the box prebuilt and is rebuilt there only if stale).  Same kernels, same order; what differs is the
device's own arithmetic (v_exp / v_rcp seeds, contraction), so inlier counts and surfel counts must be equal and poses agree to 2e-5.
Written when the round's GPU minutes were spent: non-strict xfail until a hardware run has been seen."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200)]   # first hardware run: GPUTEST_r02 (XPASS); a plain test since round 3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_executed_kernels_agree_with_the_hardware(hip):
    from maskfusion_amd import MaskFusion, synth
    env = dict(os.environ)
    env.pop("MF_EMU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipcpu", "smoke.py"), "single"], capture_output=True, text=True, timeout=1100, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    emu = json.loads(r.stdout.strip().splitlines()[-1])["single"]
    W, H = 160, 120
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 17)
    for k, e in enumerate(emu):
        rgb, d, _ = st.frame(k)
        mf.processFrame(rgb, d, timestamp=k)
        assert mf.getBackgroundModel().lastCount() == e["count"], k
        assert mf.trackStats(0)["lastICPCount"] == e["inliers"], k
        assert np.abs(mf.getCurrPose().reshape(-1) - np.array(e["pose"])).max() < 2e-5, k
    mf.close()
