"""-m gpu tests of the remaining processFrame arguments and exports of the drop-in boundary:
inPose / bootstrap (Core/MaskFusion.cpp:243,280-283,413-415), the pose log and exportPoses (:580-596,:851-879),
savePly (:733-849)."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(n, W=320, H=240):
    from maskfusion_amd import synth
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=False)
    return st, [st.frame(k) for k in range(n)]


def _quat_xyzw(R):
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(R).as_quat()


def test_in_pose_and_bootstrap_match_oracle(hip, oracle):
    from maskfusion_amd import MaskFusion, synth
    st, fr = _frames(7)
    o = oracle.Oracle(st.W, st.H, st.fx, st.fy, st.cx, st.cy, capacity=1 << 19, icpWeight=100.0, so3=0)
    mf = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, enableMultipleModels=False,
                    numGSurfels=1 << 19)
    nudge = synth.make_pose(synth.rot_xyz(0.001, -0.002, 0.0005), [0.001, 0.0, -0.002])
    for k in range(7):
        kw = {}
        if k in (2, 3):
            kw = dict(in_pose=st.gt_pose(k).astype(np.float32))              # pose supplied, no tracking
        elif k == 5:
            kw = dict(in_pose=nudge.astype(np.float32), bootstrap=True)       # tracked pose * inPose
        o.process_frame(fr[k][0], fr[k][1], **kw)
        mf.processFrame(fr[k][0], fr[k][1], timestamp=1000 * k, inPose=kw.get("in_pose"), bootstrap=kw.get("bootstrap", False))
        po, ph = o.pose, mf.getCurrPose()
        assert np.abs(po - ph).max() < 1e-4, k
        if k in (2, 3):
            assert np.abs(ph - st.gt_pose(k)).max() < 1e-6
        assert abs(o.count - mf.getBackgroundModel().lastCount()) <= max(20, 0.005 * o.count), k
    with pytest.raises(Exception):
        mf.processFrame(fr[0][0], fr[0][1], bootstrap=True)   # bootstrap without inPose (assert at MaskFusion.cpp:281)
    o.close(); mf.close()


def test_pose_log_exports_and_ply(hip, oracle, tmp_path):
    from maskfusion_amd import MaskFusion
    st, fr = _frames(6)
    mf = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, enableMultipleModels=False,
                    numGSurfels=1 << 19)
    poses = []
    for k in range(6):
        mf.processFrame(fr[k][0], fr[k][1], timestamp=33333 * (k + 1))
        poses.append(mf.getCurrPose())
    ts, p = mf.getPoseLog(0)
    assert ts.tolist() == [33333 * (k + 1) for k in range(6)]
    for k in range(6):
        assert np.abs(p[k, :3] - poses[k][:3, 3]).max() < 1e-6
        q = _quat_xyzw(poses[k][:3, :3])
        assert min(np.abs(p[k, 3:] - q).max(), np.abs(p[k, 3:] + q).max()) < 1e-5
        assert p[k, 6] > 0   # Eigen's conversion: w >= 0 when the trace is positive
    d = str(tmp_path) + os.sep
    mf.exportPoses(d)
    rows = [l.split() for l in open(d + "poses-0.txt").read().strip().split("\n")]
    assert len(rows) == 6 and all(len(r) == 8 for r in rows)
    assert rows[2][0] == "%.6f" % (33333 * 3 * 1e-6)
    assert np.allclose(np.array(rows, np.float64)[:, 1:], p, atol=1e-6)
    # PLY: header + one 31-byte record per surfel above the confidence threshold, normals negated
    mf.savePly(d)
    raw = open(d + "cloud-0.ply", "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().split("\n")
    assert lines[0] == "ply" and lines[1] == "format binary_little_endian 1.0"
    n = int([l for l in lines if l.startswith("element vertex")][0].split()[2])
    assert [l for l in lines if l.startswith("property")] == [
        "property float x", "property float y", "property float z", "property uchar red", "property uchar green",
        "property uchar blue", "property float nx", "property float ny", "property float nz", "property float radius"]
    assert len(body) == n * 31
    m = mf.getBackgroundModel().downloadMap()
    thr = mf.getBackgroundModel().getConfidenceThreshold()
    keep = m[:, 3] > thr
    assert n == int(keep.sum())
    if n:
        rec = struct.unpack("<3f3B4f", body[:31])
        s0 = m[keep][0]
        assert np.allclose(rec[:3], s0[:3]) and np.allclose(rec[6:9], -s0[8:11]) and np.isclose(rec[9], s0[11])
        col = int(s0[4])
        assert rec[3:6] == ((col >> 16) & 255, (col >> 8) & 255, col & 255)
    mf.close()


def test_cli_runs_image_directory(hip, tmp_path, capsys):
    """The headless driver (reference flags) over an image directory in the reference's layout: tracks the synthetic camera,
    writes poses-0.txt and cloud-0.ply."""
    from maskfusion_amd import cli
    from maskfusion_amd.io import write_image_dir
    st, fr = _frames(8)
    seq = str(tmp_path / "seq") + os.sep
    write_image_dir(seq, [(f[0], f[1]) for f in fr], calibration=(st.fx, st.fy, st.cx, st.cy, st.W, st.H))
    out = str(tmp_path / "out") + os.sep
    assert cli.main(["-dir", seq, "-static", "-run", "-q", "-ep", "-em", "-exportdir", out, "-i", "100", "-nso", "-confG", "2"]) == 0
    rows = np.array([l.split() for l in open(out + "poses-0.txt").read().strip().split("\n")], np.float64)
    assert rows.shape == (8, 8)
    gt = st.gt_pose(7)
    assert np.linalg.norm(rows[-1, 1:4] - gt[:3, 3]) < 6e-3         # depth quantised to millimetres by the 16-bit PNGs (3.7 mm here)
    assert os.path.getsize(out + "cloud-0.ply") > 1000
    assert "processed 8 frames" in capsys.readouterr().out


def test_set_tick_skips_frame_numbers(hip):
    """MaskFusion::setTick: surfel time stamps continue from the new tick, the pipeline keeps tracking."""
    from maskfusion_amd import MaskFusion
    st, fr = _frames(5)
    mf = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 19)
    with pytest.raises(Exception):
        mf.setTick(50)                      # not before the map exists
    mf.processFrame(fr[0][0], fr[0][1])
    mf.processFrame(fr[1][0], fr[1][1])
    assert mf.getTick() == 3
    mf.setTick(50)
    mf.processFrame(fr[2][0], fr[2][1])
    assert mf.getTick() == 51
    m = mf.getBackgroundModel().downloadMap()
    assert m[:, 7].max() == 50.0 and m[:, 6].max() == 50.0    # lastTime / initTime of surfels touched / created at tick 50
    assert np.linalg.norm(mf.getCurrPose()[:3, 3] - st.gt_pose(2)[:3, 3]) < 5e-3
    mf.close()


@pytest.mark.parametrize("multi", [False, True], ids=["single-model", "multi-model"])
def test_asynchronous_host_frames_equal_the_blocking_form(hip, multi):
    """mf_process_frame since round 4: the caller's buffers are copied into a pinned double buffer, uploaded (one packed copy) on their own
    stream under the previous frame's kernels, and the call returns when the frame is enqueued.  Same frames through
    `hostInputAsync = 0` (rounds 1-3: upload on the main stream, one synchronisation per frame): poses, counts, label images and clouds
    bit-identical -- also when the caller overwrites its buffers right after the call returns (they must have been consumed by then)."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True, n_objects=2 if multi else 0, object_motion=0.0)

    def run(asynchronous, lockstep=True, wait_upload=True):
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=multi, numGSurfels=1 << 18, numOSurfels=1 << 16,
                        modelSpawnOffset=2, trackAllModels=False)
        mf.setParam("hostLockstep", 1 if lockstep else 0)                # the call waits for frame k-2 before it enqueues frame k's upload
        mf.setParam("hostWaitUpload", 1 if wait_upload else 0)           # the call waits for its own upload: no cross-queue wait for the frame
        if multi:
            for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                         ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004)):
                mf.setParam(k, v)
        mf.setParam("hostInputAsync", 1 if asynchronous else 0)
        rgb_buf, d_buf, m_buf = np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.float32), np.zeros((H, W), np.uint8)
        poses = []
        for k in range(10):
            rgb, d, m = st.frame(k)
            rgb_buf[...] = rgb; d_buf[...] = d; m_buf[...] = m
            if multi:
                mf.processFrame(rgb_buf, d_buf, mask=m_buf, classIDs=[0, 41, 42], timestamp=k)
            else:
                mf.processFrame(rgb_buf, d_buf, timestamp=k)
            rgb_buf[...] = 0; d_buf[...] = np.nan; m_buf[...] = 7       # the caller's buffers are the caller's again
            if k % 3 == 2:
                poses.append(mf.getCurrPose())                          # (a getter synchronises; the frames in between stay queued)
        ms = mf.getModels()
        out = dict(poses=poses, ids=[x.getID() for x in ms], counts=[x.lastCount() for x in ms], clouds=[x.downloadMap() for x in ms],
                   labels=mf.downloadSegmentation() if multi else None, final=[x.getPose() for x in ms])
        mf.close()
        return out

    b = run(False)
    for a in (run(True), run(True, lockstep=False), run(True, wait_upload=False)):
        assert a["ids"] == b["ids"] and a["counts"] == b["counts"]
        assert len(a["ids"]) == (3 if multi else 1)
        for x, y in zip(a["poses"] + a["final"], b["poses"] + b["final"]):
            assert np.array_equal(x, y)
        for x, y in zip(a["clouds"], b["clouds"]):
            assert np.array_equal(x, y, equal_nan=True)
        if multi:
            assert np.array_equal(a["labels"], b["labels"])
