"""Arguments, switches and configurations of MaskFusion::processFrame that earlier rounds only exercised on the CPU-executed kernels
(tests/hipcpu/smoke.py), now on the MI355X against the oracle ("Next round" item 1):

  * Model::makeNonStatic / makeStatic / updateStaticPose (Core/Model/Model.h:263-268, MaskFusion.cpp:263-276) and
    MaskFusion::setTrackableClassIds (MaskFusion.cpp:261,940) vs OracleMM;
  * processFrame's weightMultiplier (MaskFusion.h:69-70) at 0.3 and 3.0;
  * the iteration schedules of fastOdom / pyramid off (RGBDOdometry.cpp:272,327-329), with and without the photometric term;
  * mf_process_frame_dev + mf_set_mask_class_ids == the host-pointer run, bit for bit;
  * sensor garbage (NaN / +inf / negative depth, also inside an object's mask);
  * configs[0] stand-in: S1 rendered with the -tum3 intrinsics (GUI/MainController.cpp:122), written to a .klg, run through
    `python -m maskfusion_amd.cli -l ... -tum3 -static -ep`, exported poses-0.txt against the oracle's trajectory;
  * configs[2] stand-in: image directory + -maskdir with one moving object at the reference's GUI defaults (GUI/Tools/GUI.h:189,195:
    SO(3) on, icpWeight 20) through the CLI's settings, against OracleMM.
TUM data itself is not in the container: these are plumbing stand-ins on synthetic streams and say so.
"""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

SEG_O = dict(threshold=0.3, weightDistance=150.0, weightConvexity=2.8, morphEdgeIterations=0, morphMaskIterations=0, minRelSizeNew=0.004)
SEG_D = dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0,
             newModelMinRelativeSize=0.004)


def _dev(a):
    """device copy of a numpy array (host memory under MF_EMU=1) -> (keep-alive object, raw pointer)"""
    from gpu_util import dev
    t = dev(a)
    return t, t.data_ptr()


def _mm_pair(W, H, f, track_all, n_objects, motion, **kw):
    from maskfusion_amd import MaskFusion, synth
    from oracle import mfo_mm
    cap_g, cap_o = (1 << 20, 1 << 17) if W * H > 320 * 240 else (1 << 18, 1 << 16)
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=n_objects, noise=True, object_motion=motion)
    o = mfo_mm.OracleMM(W, H, f, f, W / 2.0, H / 2.0, icpWeight=100.0, so3=0, capacity=cap_g, capacityObject=cap_o, modelSpawnOffset=2,
                        trackAllModels=int(track_all), seg=SEG_O, **kw)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=cap_g, numOSurfels=cap_o, enableMultipleModels=True,
                    modelSpawnOffset=2, trackAllModels=track_all)
    for k, v in SEG_D.items():
        mf.setParam(k, v)
    return st, o, mf


def _compare_lists(mf, o, k, pose_tol=2e-4, count_rel=0.01, label_tol=2e-3):
    ms = mf.getModels()
    g_ids, o_ids = [x.getID() for x in ms], [o.model_id(i) for i in range(o.n_models)]
    assert g_ids == o_ids, (k, g_ids, o_ids)
    for i, x in enumerate(ms):
        assert np.abs(x.getPose() - o.model_pose(i)).max() < pose_tol, (k, i, np.abs(x.getPose() - o.model_pose(i)).max())
        a, b = x.lastCount(), o.model_count(i)
        assert abs(a - b) <= max(20, count_rel * b), (k, i, a, b)
        assert x.getClassID() == o.model_class(i), (k, i)
    assert float((mf.downloadSegmentation() != o.segmentation()).mean()) <= label_tol, k
    return ms


def _first_system(mf, o, i):
    """(inliers device, inliers oracle, relative A / b difference) of the first Gauss-Newton system of model i's last tracking step"""
    dl, ol = mf.debugRead("icp_log", model=i), o.model_track_log(i)
    scale = max(1e-30, float(np.abs(ol[0][:27]).max()))
    return int(dl[0][28]), int(ol[0][28]), float(np.abs(dl[0][:27] - ol[0][:27]).max() / scale)


def test_static_switches_vs_oracle(hip, oracle):
    """trackAllModels off (the GUI default, GUI.h:344): a spawned object is static and follows the camera -- pose = initialC2Winv * globalPose
    (Model.h:263).  makeNonStatic before frame 6 puts it under its own tracker (MaskFusion.cpp:263), makeStatic before frame 10 records
    pose * globalPose^-1 and lets it follow again.  Both sides get the same calls on the same frames; the oracle takes the product's filtered
    depth and, while the object is tracked, the product's poses (teacher forcing: a small box's ICP is ill-conditioned and chaotic in its last
    digits -- see test_s2_eight_objects_tracked_teacher_forced -- the comparison here is about the switch): what is gated on the tracked
    frames is that BOTH sides track the object, from the same state to the same first Gauss-Newton system (inlier count exact), and that the
    surfel counts stay exact."""
    W, H, f = 640, 480, 528.0
    st, o, mf = _mm_pair(W, H, f, False, 1, 0.0)
    cls = [0, 41]
    rel, tracked_frames, step_diffs = [], 0, []
    for k in range(14):
        rgb, d, m = st.frame(k)
        ms = mf.getModels()
        if k == 6:
            assert len(ms) == 2 and o.n_models == 2, "the scenario must have spawned its object by frame 6"
            assert not ms[1].isNonstatic() and not o.is_nonstatic(1)
            ms[1].makeNonStatic(); o.make_nonstatic(1)
            assert ms[1].isNonstatic() and o.is_nonstatic(1)
        if k == 10:
            assert len(ms) == 2 and ms[1].isNonstatic(), "the object must have survived its tracked stretch"
            ms[1].makeStatic(); o.make_static(1)
            assert not ms[1].isNonstatic() and not o.is_nonstatic(1)
        mf.processFrame(rgb, d, mask=m, classIDs=cls, timestamp=k)
        if 6 <= k < 10:
            gm = mf.getModels()
            o.force_tracking([x.getID() for x in gm], [x.getPose() for x in gm])
        o.process_frame(rgb, d, m, cls, depth_filtered=mf.debugRead("depthF"))
        ms = _compare_lists(mf, o, k)
        if len(ms) > 1:
            rel.append((k, ms[1].getPose() @ np.linalg.inv(ms[0].getPose())))
            if 6 <= k < 10:
                own, was_tracked = o.model_tracked_pose(1)
                assert was_tracked, k
                assert ms[1].lastCount() == o.model_count(1), k
                gi, oi, sysrel = _first_system(mf, o, 1)
                assert gi == oi and gi > 50 and sysrel < 2e-4, (k, gi, oi, sysrel)
                tracked_frames += 1
                step_diffs.append(float(np.abs(own - ms[1].getPose()).max()))
                assert np.abs(o.model_tracked_pose(0)[0] - ms[0].getPose()).max() < 1e-5, k     # the background's step, same state: float noise
            elif k >= 4:
                assert not o.model_tracked_pose(1)[1], k
    mf.close(); o.close()
    print("oracle's own tracking step vs the device's pose on the tracked frames:", step_diffs)
    assert tracked_frames == 4
    assert max(step_diffs) < 5e-2
    # static stretches: pose_obj * pose_bg^-1 is constant (what makeStatic recorded)
    before = [r for k, r in rel if k < 6]
    after = [r for k, r in rel if k >= 10]
    assert len(before) >= 3 and len(after) == 4
    for seq in (before, after):
        for r in seq[1:]:
            assert np.abs(r - seq[0]).max() < 5e-6
    assert np.abs(after[0] - before[0]).max() > 1e-5, "the tracked stretch must have moved the object relative to the camera frame it was spawned in"


def test_trackable_class_ids_vs_oracle(hip, oracle):
    """setTrackableClassIds({41}) with trackAllModels on: the object of class 41 is tracked, the one of class 42 follows the background
    (MaskFusion.cpp:261-275), on both sides.  (VGA: at 320x240 a tracked 0.3 m box is a few hundred pixels, falls to the 0.2 m jump rule at
    once and is re-spawned every other frame -- on both sides -- so that the second object never gets its turn.)"""
    W, H, f = 640, 480, 528.0
    st, o, mf = _mm_pair(W, H, f, True, 2, 0.0)
    mf.setTrackableClassIds([41]); o.set_trackable_class_ids([41])
    cls = [0, 41, 42]
    seen = {41: [], 42: []}
    for k in range(11):
        rgb, d, m = st.frame(k)
        mf.processFrame(rgb, d, mask=m, classIDs=cls, timestamp=k)
        gm = mf.getModels()
        o.force_tracking([x.getID() for x in gm], [x.getPose() for x in gm])
        o.process_frame(rgb, d, m, cls, depth_filtered=mf.debugRead("depthF"))
        _compare_lists(mf, o, k)
        for i in range(1, o.n_models):
            seen[o.model_class(i)].append(o.model_tracked_pose(i)[1])
    mf.close(); o.close()
    assert len(seen[41]) >= 4 and len(seen[42]) >= 4, "both objects must have been spawned"
    assert any(seen[41][1:]) and not any(seen[42]), seen


@pytest.mark.parametrize("wm", [0.3, 3.0])
def test_weight_multiplier(hip, oracle, wm):
    """processFrame's third argument (Model::fuse weighting = computeFusionWeight(weightMultiplier), Model.cpp:449-464): poses given,
    filtered depth shared -- every surfel in the same slot with the same confidence, colour and time stamps."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    o = oracle.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 18, icpWeight=100.0, so3=0, confGlobal=2.0)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 18, initConfidenceGlobal=2.0)
    conf = []
    for k in range(8):
        rgb, d, _ = st.frame(k)
        T = st.gt_pose(k).astype(np.float32)
        mf.processFrame(rgb, d, timestamp=k, weightMultiplier=wm, inPose=T if k else None)
        o.process_frame(rgb, d, weight_multiplier=wm, in_pose=T if k else None, depth_filtered=mf.debugRead("depthF"))
        g, oc = mf.getBackgroundModel().downloadMap(), o.surfels()
        assert len(g) == len(oc), (k, len(g), len(oc))
        assert np.abs(g[:, :4] - oc[:, :4]).max() < 1e-6 * max(1.0, float(np.abs(oc[:, :4]).max())), k
        assert np.array_equal(g[:, 4:8], oc[:, 4:8]), k
        assert np.abs(g[:, 8:12] - oc[:, 8:12]).max() < 2e-6, k
        conf.append(float(g[:, 3].mean()))
    mf.close(); o.close()
    print("mean confidence per frame at weightMultiplier", wm, conf)
    assert conf[-1] > conf[1]


@pytest.mark.parametrize("name,fast,pyr,icp,so3", [("fast", 1, 1, 100.0, 0), ("nopyramid", 0, 0, 100.0, 0), ("fast_nopyramid", 1, 0, 100.0, 0),
                                                   ("fast_rgbd_so3", 1, 1, 20.0, 1), ("nopyramid_rgbd", 0, 0, 20.0, 0)])
def test_iteration_schedules(hip, oracle, name, fast, pyr, icp, so3):
    """{fastOdom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0} (RGBDOdometry.cpp:327-329; -fo / pyramid off), with and without the photometric
    term: the device loop against the oracle's, six frames."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    o = oracle.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 18, icpWeight=icp, so3=so3, fastOdom=fast, pyramid=pyr)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=icp, so3=bool(so3), fastOdom=bool(fast), enableMultipleModels=False, numGSurfels=1 << 18)
    if not pyr:
        mf.setPyramid(0)
    worst = 0.0
    for k in range(6):
        rgb, d, _ = st.frame(k)
        mf.processFrame(rgb, d, timestamp=k)
        o.process_frame(rgb, d)
        dp = float(np.abs(mf.getCurrPose() - o.pose).max())
        worst = max(worst, dp)
        gc, oc = mf.getBackgroundModel().lastCount(), o.count
        assert dp < 1e-4, (name, k, dp)
        assert abs(gc - oc) <= max(8, oc // 100), (name, k, gc, oc)
        if icp < 100.0 and k > 0:
            s = mf.trackStats(0)
            assert s["lastRGBCount"] > 0, (name, k)
            if so3:
                assert s["so3Iterations"] >= 1, (name, k)
    mf.close(); o.close()
    print(name, "max pose difference", worst)


def test_device_resident_masks_equal_host_pointer_run(hip):
    """mf_process_frame_dev + mf_set_mask_class_ids (frames and masks already in HBM: what bench.py --config 2s times) against
    mf_process_frame with host pointers and FrameData::classIDs: the same multi-model run bit for bit (tracked objects, spawns, drops)."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0

    def make():
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 18, numOSurfels=1 << 16, enableMultipleModels=True,
                        modelSpawnOffset=2, trackAllModels=True)
        for k, v in SEG_D.items():
            mf.setParam(k, v)
        return mf

    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=2, noise=True, object_motion=1.0)
    a, b = make(), make()
    b.setMaskClassIDs([0, 41, 42])
    keep = []
    n_obj = 0
    for k in range(12):
        rgb, d, m = st.frame(k)
        a.processFrame(rgb, d, mask=m, classIDs=[0, 41, 42], timestamp=k)
        bufs = [_dev(np.ascontiguousarray(rgb)), _dev(np.ascontiguousarray(d, np.float32)), _dev(np.ascontiguousarray(m, np.uint8))]
        keep.append(bufs)
        b.processFrameDevice(bufs[0][1], bufs[1][1], bufs[2][1], timestamp=k)
        b.sync()
        ma, mb = a.getModels(), b.getModels()
        assert [x.getID() for x in ma] == [x.getID() for x in mb], k
        assert [x.getClassID() for x in ma] == [x.getClassID() for x in mb], k
        assert [x.lastCount() for x in ma] == [x.lastCount() for x in mb], k
        for x, y in zip(ma, mb):
            assert np.array_equal(x.getPose(), y.getPose()), k
        assert np.array_equal(a.downloadSegmentation(), b.downloadSegmentation()), k
        n_obj = max(n_obj, len(ma) - 1)
    for x, y in zip(a.getModels(), b.getModels()):
        assert np.array_equal(x.downloadMap(), y.downloadMap())
    a.close(); b.close()
    assert n_obj >= 2, "the scenario must spawn its objects"


def _garbage(d, m=None):
    d = d.copy()
    H, W = d.shape
    d[H // 12:H // 12 + 4, W // 8:W // 8 + 10] = np.nan
    d[H // 2:H // 2 + 3, (5 * W) // 8:(5 * W) // 8 + 4] = np.inf
    d[(3 * H) // 4:(3 * H) // 4 + 4, W // 4:W // 4 + 8] = -1.0
    if m is not None:
        ys, xs = np.where(m == 1)
        if len(ys) > 20:
            d[ys[:6], xs[:6]] = np.nan
    return d


def test_sensor_garbage_single_model(hip, oracle):
    """NaN, +inf and negative patches in the depth image: the filter includes every in-image tap like the shader does
    (depth_bilateral_metric.frag:30-76: a NaN tap poisons its 13x13 neighbourhood, which then fails the z > 0 tests downstream), so the
    frame loses those regions and nothing else, on both sides."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    o = oracle.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 18, icpWeight=100.0, so3=0)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 18)
    for k in range(6):
        rgb, d, _ = st.frame(k)
        d = _garbage(d)
        mf.processFrame(rgb, d, timestamp=k)
        o.process_frame(rgb, d)
        gF, oF = mf.debugRead("depthF"), o.dbg("depthF")
        assert np.array_equal(np.isnan(gF), np.isnan(oF)), k
        assert int(np.isnan(gF).sum()) > 100, k
        assert float(np.nanmax(np.abs(gF - oF))) < 2e-5, k
        pose = mf.getCurrPose()
        assert np.isfinite(pose).all(), k
        assert np.abs(pose - o.pose).max() < 1e-4, k
        gc, oc = mf.getBackgroundModel().lastCount(), o.count
        assert abs(gc - oc) <= max(8, oc // 100), (k, gc, oc)
    mf.close(); o.close()


def test_sensor_garbage_multi_model(hip, oracle):
    """the multi-model frame (global projection, label stage, spawn, per-model fusion) on the same kind of input, some of it inside an object's
    mask; the oracle takes the product's filtered depth: ids, surfel counts and label images identical."""
    W, H, f = 320, 240, 264.0
    st, o, mf = _mm_pair(W, H, f, False, 2, 0.0)
    cls = [0, 41, 42]
    for k in range(9):
        rgb, d, m = st.frame(k)
        d = _garbage(d, m)
        mf.processFrame(rgb, d, mask=m, classIDs=cls, timestamp=k)
        o.process_frame(rgb, d, m, cls, depth_filtered=mf.debugRead("depthF"))
        ms = mf.getModels()
        assert [x.getID() for x in ms] == [o.model_id(i) for i in range(o.n_models)], k
        assert [x.lastCount() for x in ms] == [o.model_count(i) for i in range(o.n_models)], k
        assert int((mf.downloadSegmentation() != o.segmentation()).sum()) == 0, k
        for i, x in enumerate(ms):
            assert np.abs(x.getPose() - o.model_pose(i)).max() < 2e-4, (k, i)
    n = len(mf.getModels())
    mf.close(); o.close()
    assert n == 3


def _read_pose_file(path):
    return np.array([l.split() for l in open(path).read().strip().split("\n")], np.float64)


def test_config0_standin_tum3_klg_through_cli(hip, oracle, tmp_path, capsys):
    """configs[0] stand-in ("TUM fr3/long_office .klg, -static"): the S1 room rendered with the -tum3 intrinsics (fx != fy, off-centre principal
    point: 535.4, 539.2, 320.1, 247.6 -- GUI/MainController.cpp:122), written to a .klg (16-bit millimetre depth, deflated; raw colour), run
    through the headless driver with the reference's flags.  The exported poses-0.txt must be the oracle's trajectory on the frames the .klg
    reader delivers, at the GUI's effective defaults (SO(3) on, icpWeight 20, depth cutoff 4, confidence 10, open loop)."""
    from maskfusion_amd import cli, synth
    from maskfusion_amd.io import KlgLogReader, write_klg
    n = int(os.environ.get("MF_STANDIN_FRAMES", "12"))
    st = synth.Stream(W=640, H=480, fx=535.4, fy=539.2, cx=320.1, cy=247.6, noise=True)
    frames = [st.frame(k) for k in range(n)]
    klg = str(tmp_path / "s1_tum3.klg")
    write_klg(klg, [(33333 * k, f[0], f[1]) for k, f in enumerate(frames)])
    out = str(tmp_path / "out") + os.sep
    assert cli.main(["-l", klg, "-tum3", "-static", "-run", "-q", "-ep", "-exportdir", out]) == 0
    rows = _read_pose_file(out + "poses-0.txt")
    s = cli.settings(cli.parse(["-l", klg, "-tum3", "-static"]))
    o = oracle.Oracle(s["W"], s["H"], s["fx"], s["fy"], s["cx"], s["cy"], capacity=1 << 21, icpWeight=s["icpWeight"], so3=int(s["so3"]),
                      confGlobal=s["confGlobal"], depthCutoff=s["depthCutoff"], outlierCoeff=s["outlierCoefficient"], timeDelta=s["timeDelta"],
                      fastOdom=int(s["fastOdom"]))
    ref = []
    for fr in KlgLogReader(klg, 640, 480):
        o.process_frame(fr.rgb, fr.depth)
        ref.append((fr.timestamp, o.pose.copy()))
    o.close()
    assert len(ref) == n - 1                       # upstream's hasMore() never delivers the last frame (KlgLogReader.cpp:118)
    assert rows.shape == (len(ref), 8)
    dt = np.array([np.linalg.norm(rows[k, 1:4] - ref[k][1][:3, 3]) for k in range(len(ref))])
    print("configs[0] stand-in: |t_cli - t_oracle| per frame (mm):", np.round(dt * 1e3, 4).tolist())
    assert np.abs(rows[:, 0] - np.array([r[0] for r in ref], np.float64) / 1e6).max() < 1e-6    # seconds, six decimals (MaskFusion.cpp:733-760)
    assert dt.max() < 1e-4
    # ... and the quaternion columns describe the same rotations (sign-free comparison through the rotation matrix)
    from scipy.spatial.transform import Rotation
    for k in range(len(ref)):
        Rm = Rotation.from_quat(rows[k, 4:8]).as_matrix()
        assert np.abs(Rm - ref[k][1][:3, :3]).max() < 2e-4, k
    gt = st.gt_pose(len(ref) - 1)
    assert np.linalg.norm(rows[-1, 1:4] - gt[:3, 3]) < 1.5e-2   # and it is the camera's motion (millimetre-quantised depth)
    assert f"processed {len(ref)} frames" in capsys.readouterr().out


def test_config2_standin_maskdir_one_moving_object_gui_defaults(hip, oracle, tmp_path):
    """configs[2] stand-in ("walking_xyz with precomputed masks, 1 background + 1 dynamic object"): an image directory in the reference's
    layout (Color####.png, Depth####.png, Mask####.png + Mask####.txt) with ONE moving, instance-masked box, read back through ImageLogReader and
    processed with the settings the CLI / GUI push (GUI/Tools/GUI.h:189,195: SO(3) on, icpWeight 20; depth cutoff 4, confidences 10 / 0.01,
    trackAllModels off, open loop) -- against OracleMM on the same decoded frames.  Only the spawn offset (22 upstream) is shortened so that the
    object exists within the run."""
    from maskfusion_amd import MaskFusion, cli, synth
    from maskfusion_amd.io import ImageLogReader, write_image_dir
    from oracle import mfo_mm
    n = int(os.environ.get("MF_STANDIN_FRAMES", "14"))
    st = synth.Stream(W=640, H=480, fx=535.4, fy=539.2, cx=320.1, cy=247.6, n_objects=1, noise=True, object_motion=1.0)
    frames = [st.frame(k) for k in range(n)]
    seq = str(tmp_path / "seq") + os.sep
    write_image_dir(seq, [(f[0], f[1]) for f in frames], masks=[f[2] for f in frames], class_ids=[[0, 41]] * n,
                    calibration=(st.fx, st.fy, st.cx, st.cy, st.W, st.H))
    flags = cli.parse(["-dir", seq, "-maskdir", seq, "-offset", "3"])
    s = cli.settings(flags)
    reader = cli.open_reader(flags, s)
    assert (s["fx"], s["fy"], s["cx"], s["cy"]) == (st.fx, st.fy, st.cx, st.cy)
    mf = MaskFusion(s["W"], s["H"], s["fx"], s["fy"], s["cx"], s["cy"], timeDelta=s["timeDelta"], initConfidenceGlobal=s["confGlobal"],
                    initConfidenceObject=s["confObject"], depthCut=s["depthCutoff"], icpThresh=s["icpWeight"], fastOdom=s["fastOdom"], so3=s["so3"],
                    enableMultipleModels=s["multi"], outlierCoefficient=s["outlierCoefficient"], modelSpawnOffset=s["modelSpawnOffset"],
                    trackAllModels=s["trackAllModels"], numGSurfels=1 << 21, numOSurfels=1 << 18)
    for k, v in s["mf"].items():
        mf.setParam(k, v)
    assert s["so3"] and s["icpWeight"] == 20.0 and s["multi"] and not s["trackAllModels"]
    seg = dict(threshold=s["mf"]["mfThreshold"], weightDistance=s["mf"]["mfWeightDistance"], weightConvexity=s["mf"]["mfWeightConvexity"],
               morphEdgeIterations=s["mf"]["mfMorphEdgeIterations"], morphEdgeRadius=s["mf"]["mfMorphEdgeRadius"],
               morphMaskIterations=s["mf"]["mfMorphMaskIterations"], morphMaskRadius=s["mf"]["mfMorphMaskRadius"],
               minRelSizeNew=s["mf"]["newModelMinRelativeSize"], maxRelSizeNew=s["mf"]["newModelMaxRelativeSize"])
    o = mfo_mm.OracleMM(s["W"], s["H"], s["fx"], s["fy"], s["cx"], s["cy"], icpWeight=s["icpWeight"], so3=int(s["so3"]), capacity=1 << 21,
                        capacityObject=1 << 18, modelSpawnOffset=s["modelSpawnOffset"], trackAllModels=0, seg=seg, confGlobal=s["confGlobal"],
                        confObject=s["confObject"], depthCutoff=s["depthCutoff"], outlierCoeff=s["outlierCoefficient"], timeDelta=s["timeDelta"])
    worst, k = 0.0, 0
    for k, fr in enumerate(reader):
        assert fr.mask is not None and list(fr.classIDs) == [0, 41]
        mf.processFrame(fr.rgb, fr.depth, mask=fr.mask, timestamp=int(fr.timestamp), classIDs=tuple(fr.classIDs))
        o.process_frame(fr.rgb, fr.depth, fr.mask, list(fr.classIDs), depth_filtered=mf.debugRead("depthF"))
        ms = _compare_lists(mf, o, k, pose_tol=2e-4, count_rel=0.01, label_tol=2e-3)
        worst = max(worst, max(float(np.abs(x.getPose() - o.model_pose(i)).max()) for i, x in enumerate(ms)))
        if k > 0:
            stt = mf.trackStats(0)
            assert stt["lastRGBCount"] > 0 and stt["so3Iterations"] >= 1, k
    n_models = len(mf.getModels())
    mf.close(); o.close()
    print("configs[2] stand-in:", k + 1, "frames, models", n_models, "max pose difference", worst)
    assert k + 1 == n and n_models == 2


def test_run_culling_changes_nothing(hip):
    """`cullRuns` (round 5): the projection passes (index map x 2, prediction, GlobalProjection) only visit the runs of the surfel buffer whose
    bounding box meets the viewing frustum; round 6: Model::clean in place only those in which one of its rules can apply (k_cull_clean).  On a
    pre-filled room map of which the camera sees a fraction, with culling on and off: poses, counts, every surfel and the prediction maps
    bit-identical on every frame -- and the visibility list and the clean list really are fractions of the table."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    frames = [st.frame(k) for k in range(8)]
    room = synth.dense_room_map(st.scene, 600_000, last_time=1.0)

    def run(cull):
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 20, initConfidenceGlobal=10.0)
        mf.setParam("cullRuns", 1 if cull else 0)
        mf.setParam("bigMapElements", 0)          # (by default only maps of >= 6 M surfels are kept as runs: in-place clean, culled passes)
        out = []
        for k, (rgb, d, _) in enumerate(frames):
            mf.processFrame(rgb, d, timestamp=k)
            if k == 0:
                mf.getBackgroundModel().uploadMap(room)
            bg = mf.getBackgroundModel()
            out.append(dict(pose=mf.getCurrPose(), count=bg.lastCount(), pv=bg.debugRead("pred_vertex"), pi=bg.debugRead("pred_image")))
        vis, runs, cleaned = mf.getParam("visibleRuns"), mf.getParam("backgroundRuns"), mf.getParam("cleanRuns")   # (before the download: it compacts the buffer)
        cloud = mf.getBackgroundModel().downloadMap()
        mf.close()
        return out, cloud, vis, runs, cleaned

    (a, ca, vis, runs, cleaned), (b, cb, _, _, _) = run(True), run(False)
    print("visible runs", vis, "of", runs, "-- runs the in-place clean visited:", cleaned)
    assert 0 < vis < 0.6 * runs and runs >= 600_000 / 512
    # Model::clean in place: the runs in view (no far limit) + those holding young unstable surfels (the visibility list is the prediction's, taken
    # after the frame's new surfels were appended: a few runs more)
    assert vis - 8 <= cleaned < 0.7 * runs
    for k, (x, y) in enumerate(zip(a, b)):
        assert x["count"] == y["count"] and np.array_equal(x["pose"], y["pose"]), k
        assert np.array_equal(x["pv"], y["pv"], equal_nan=True) and np.array_equal(x["pi"], y["pi"]), k
    assert np.array_equal(ca, cb, equal_nan=True)


@pytest.mark.parametrize("multi", [False, True], ids=["single-model", "multi-model"])
def test_clean_forms_agree(hip, multi):
    """A model's fuse / clean passes take one of three forms by its size (mf_frame.inl: enqueue_fuse_clean): below `inPlaceElements` update.vert
    as a copy with the second index scatter riding on it + Model::clean in two launches over a static partition; from there to
    `bigMapElements` update.vert in place + the two-launch clean; above, update.vert in place + clean IN PLACE on the buffer's runs (round 6:
    k_cull_clean + k_clean_runs, the frame's new surfels appended; the buffer becomes sparse and is read through its run table).  The same
    frames through all three: model list, counts, every surfel of every model in its slot, poses and label images bit-identical -- single
    model, and background + object models (whose passes are batched: one launch per pass for all of them) --, and through a run that
    changes form from frame to frame (a sparse buffer is compacted on its way back to the small forms)."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True, n_objects=3 if multi else 0, object_motion=0.0)
    frames = [st.frame(k) for k in range(12)]

    cycle = [(1 << 30, 1 << 30), (0, 0), (1 << 30, 0), (0, 0), (1 << 30, 1 << 30), (1 << 30, 0)]

    def run(big, in_place, changing=False):
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=multi, numGSurfels=1 << 19, numOSurfels=1 << 16,
                        modelSpawnOffset=2, trackAllModels=False, initConfidenceGlobal=10.0, initConfidenceObject=0.01)
        if multi:
            for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                         ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004)):
                mf.setParam(k, v)
        mf.setParam("bigMapElements", big)
        mf.setParam("inPlaceElements", in_place)
        poses = []
        for k, (rgb, d, m) in enumerate(frames):
            if changing:      # a map that crosses the thresholds from frame to frame: the live buffer alternates or not, the run table comes and goes
                mf.setParam("bigMapElements", cycle[k % len(cycle)][0])
                mf.setParam("inPlaceElements", cycle[k % len(cycle)][1])
            mf.processFrame(rgb, d, mask=m if multi else None, classIDs=[0, 41, 42, 43] if multi else (), timestamp=k)
            poses.append([x.getPose() for x in mf.getModels()])
        ms = mf.getModels()
        out = dict(poses=poses, ids=[x.getID() for x in ms], counts=[x.lastCount() for x in ms], clouds=[x.downloadMap() for x in ms],
                   labels=mf.downloadSegmentation() if multi else None)
        mf.close()
        return out

    a = run(0, 0)
    for b in (run(1 << 30, 1 << 30), run(1 << 30, 0), run(0, 0, changing=True)):
        assert a["ids"] == b["ids"] and a["counts"] == b["counts"], (a["ids"], b["ids"], a["counts"], b["counts"])
        if multi:
            assert len(a["ids"]) >= 3, a["ids"]     # at least two objects: their passes really were batched
            assert np.array_equal(a["labels"], b["labels"])
        for pa, pb in zip(a["poses"], b["poses"]):
            assert len(pa) == len(pb) and all(np.array_equal(x, y) for x, y in zip(pa, pb))
        for x, y in zip(a["clouds"], b["clouds"]):
            assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("multi", [False, True], ids=["single-model", "multi-model"])
def test_in_place_clean_compaction_paths(hip, multi):
    """Model::clean in place appends a frame's new surfels behind the buffer's last run and leaves holes inside the runs that lose surfels; the
    host compacts the buffer when its bounds on the used slots / table entries run out (mf_frame.inl: prepare_in_place), and a map within a
    frame's candidates of its capacity takes the two-launch form, whose ordered copy stops at the capacity like the reference's transform
    feedback.  A background map that grows INTO its capacity (320 x 240: 19 200 candidates per frame against 102 400 slots; 80 k surfels loaded
    behind the first frame) goes through all of that: in place until the slots behind the last run could run out, a compaction, in place again,
    then the near-full regime.  Against the small-map forms on the same frames, and with a compaction forced on every frame: counts and poses
    on every frame, every surfel of every model in its slot bit-identical."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True, n_objects=3 if multi else 0, object_motion=0.0)
    frames = [st.frame(k) for k in range(26)]
    room = synth.dense_room_map(st.scene, 80_000, last_time=1.0)

    def run(big, every=0):
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=multi, numGSurfels=1 << 17, numOSurfels=1 << 15,
                        modelSpawnOffset=2, trackAllModels=False, initConfidenceGlobal=10.0, initConfidenceObject=0.01)
        if multi:
            for k, v in (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
                         ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004)):
                mf.setParam(k, v)
        mf.setParam("bigMapElements", big)
        mf.setParam("inPlaceElements", big)
        mf.setParam("densifyEvery", every)
        per_frame = []
        for k, (rgb, d, m) in enumerate(frames):
            mf.processFrame(rgb, d, mask=m if multi else None, classIDs=[0, 41, 42, 43] if multi else (), timestamp=k)
            if k == 0:
                mf.getBackgroundModel().uploadMap(room)
            per_frame.append(([x.getID() for x in mf.getModels()], [x.lastCount() for x in mf.getModels()], [x.getPose() for x in mf.getModels()]))
        compactions = mf.getParam("densifyCount")
        clouds = [x.downloadMap() for x in mf.getModels()]
        mf.close()
        return per_frame, clouds, compactions

    (pa, ca, na), (pb, cb, nb), (pc, cc, nc) = run(0), run(1 << 30), run(0, every=1)
    print("compactions: in place", na, "-- small forms", nb, "-- in place, forced every frame", nc, "; final counts", pa[-1][1])
    assert nb == 0 and na >= 1 and nc > na
    assert pa[-1][1][0] + (W // 2) * (H // 2) > (64 * int(np.sqrt(float(1 << 17)) / 64)) ** 2       # the background ended within a frame's candidates of its capacity
    for other, clouds in ((pb, cb), (pc, cc)):
        for k, ((ia, na_, qa), (ib, nb_, qb)) in enumerate(zip(pa, other)):
            assert ia == ib and na_ == nb_, (k, ia, ib, na_, nb_)
            assert all(np.array_equal(x, y) for x, y in zip(qa, qb)), k
        assert len(ca) == len(clouds)
        for x, y in zip(ca, clouds):
            assert np.array_equal(x, y, equal_nan=True)


def test_in_place_clean_first_surfel_rule(hip):
    """The first surfel of a buffer is vertex 0, which the index map cannot tell from "no surfel" (index_map.frag writes the vertex id into a texture
    cleared to 0; data.vert / copy_unstable.vert test `> 0`): it occludes but is never merged into and never counted in Model::clean's window.
    In a sparse buffer that surfel is the first live slot (FrameDev::first), and it moves when the first run loses surfels or empties.  A map
    whose first 700 surfels (more than a run) are unstable, out of view, and die by the age rule in the first frame, followed by surfels in
    view: in place against the small-map forms, every frame's count and pose, every surfel in its slot."""
    from maskfusion_amd import MaskFusion, synth
    W, H, f = 320, 240, 264.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    frames = [st.frame(k) for k in range(6)]
    room = synth.dense_room_map(st.scene, 150_000, last_time=40.0)
    z = room[:, 2]
    u, v = f * room[:, 0] / np.maximum(z, 1e-3) + W / 2.0, f * room[:, 1] / np.maximum(z, 1e-3) + H / 2.0
    inview = (z > 0.3) & (u > 8) & (u < W - 8) & (v > 8) & (v < H - 8)
    out = room[~inview]
    room = np.concatenate([out[:700], room[inview], out[700:]])    # 700 surfels out of view (nothing merges into them), then the part in view
    room[:700, 3] = 1.0            # unstable (confidence threshold 10) ...
    room[:700, 7] = 10.0           # ... and last seen 31 frames before tick 41: dropped by the age rule (copy_unstable.vert:118-125)

    def run(big):
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 20, initConfidenceGlobal=10.0)
        mf.setParam("bigMapElements", big)
        mf.setParam("inPlaceElements", big)
        mf.processFrame(*frames[0][:2], timestamp=0)
        mf.getBackgroundModel().uploadMap(room)
        mf.setTick(41)
        out = []
        for k, (rgb, d, _) in enumerate(frames[1:]):
            mf.processFrame(rgb, d, timestamp=k + 1)
            out.append((mf.getBackgroundModel().lastCount(), mf.getCurrPose()))
        cloud = mf.getBackgroundModel().downloadMap()
        mf.close()
        return out, cloud

    (a, ca), (b, cb) = run(0), run(1 << 30)
    print("counts", [n for n, _ in a], "of", len(room), "uploaded")
    assert a[0][0] < len(room) - 300            # the unstable head is gone after the first frame (a few hundred new surfels came in)
    for k, ((na, pa), (nb, pb)) in enumerate(zip(a, b)):
        assert na == nb and np.array_equal(pa, pb), (k, na, nb)
    assert np.array_equal(ca, cb, equal_nan=True)


@pytest.mark.parametrize("size", [(640, 480), (200, 152)], ids=["vga", "200x152"])
def test_fused_rgb_pyramid_equals_the_single_kernels(hip, size):
    """`fusedRgbPyramid` (round 5): the frame's intensity pyramid and its derivative / gate images come out of ONE LDS-tiled launch instead of
    imageBGRToIntensity + 2 x pyrDownUcharGauss + computeDerivativeImages x 3 (cudafuncs.cu:534-588,602-639,658-718; the single kernels are pinned
    to the reference's vectors in test_gpu_ref_golden.py / test_gpu_rgbd.py).  Every image of every level identical on every frame -- a size
    with partial tiles included, colour images with zero patches (the zero-skipping rule, the gate's 4 x 4 window) --, and with them poses and map."""
    from maskfusion_amd import MaskFusion, synth
    W, H = size
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    frames = [st.frame(k) for k in range(6)]
    rgb2 = frames[-1][0].copy(); rgb2[10:30, 20:50] = 0; rgb2[H - 9:, :40] = 0; rgb2[:, W - 7:] = 0
    frames[-1] = (rgb2, frames[-1][1], frames[-1][2])
    names = [f"{p}{i}" for p in ("gray", "dIdx", "dIdy", "rgb_gate") for i in range(3)]

    def run(fused):
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=20.0, so3=True, enableMultipleModels=False, numGSurfels=1 << 19)
        mf.setParam("fusedRgbPyramid", fused)
        taps, poses = [], []
        for k, (rgb, d, _) in enumerate(frames):
            mf.processFrame(rgb, d, timestamp=k)
            taps.append({t: mf.debugRead(t).copy() for t in names})
            poses.append(mf.getCurrPose().copy())
        cloud = mf.getBackgroundModel().downloadMap().copy()
        mf.close()
        return taps, poses, cloud

    (ta, pa, ca), (tb, pb, cb) = run(0), run(1)
    for k in range(len(frames)):
        for t in names:
            assert np.array_equal(ta[k][t], tb[k][t]), (k, t, int((ta[k][t] != tb[k][t]).sum()))
        assert np.array_equal(pa[k], pb[k]), k
    assert np.array_equal(ca, cb, equal_nan=True)
    assert int(ta[-1]["rgb_gate0"].sum()) > 100 and int((ta[-1]["gray0"] == 0).sum()) > 500


@pytest.mark.parametrize("size", [(640, 480), (200, 152)], ids=["vga", "200x152"])
def test_fused_preprocess_launch_equals_the_two_kernels(hip, size):
    """`fusedPreprocessLaunch` (round 6): the frame's depth filter (depth_bilateral_metric.frag) and the model-side pyramid of the same frame's tracking
    step (initICPModel: copyMaps + 2 resizes + 3 transforms + fill-in, RGBDOdometry.cpp:153-185) run as the two halves of ONE launch instead of
    k_bilateral followed by k_model_pyramid -- the same bodies on the same data.  Filtered depth, all six model-side maps, poses and the map must be
    the same bits on every frame; a size with partial tiles included, and a stream whose holes switch the fill-in on."""
    from maskfusion_amd import MaskFusion, synth
    W, H = size
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    frames = [st.frame(k) for k in range(8)]
    names = ["depthF"] + [f"{p}{i}" for p in ("vmap_g", "nmap_g") for i in range(3)]

    def run(fused):
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 19)
        mf.setParam("fusedPreprocessLaunch", fused)
        taps, poses, fill = [], [], []
        for k, (rgb, d, _) in enumerate(frames):
            if k == 5:
                d = d.copy(); d[: H // 2] = 0.0          # half the image without depth: the next frame's prediction is sparse -> fill-in
            mf.processFrame(rgb, d, timestamp=k)
            taps.append({t: mf.debugRead(t).copy() for t in names})
            poses.append(mf.getCurrPose().copy())
        cloud = mf.getBackgroundModel().downloadMap().copy()
        mf.close()
        return taps, poses, cloud

    (ta, pa, ca), (tb, pb, cb) = run(0), run(1)
    for k in range(len(frames)):
        for t in names:
            assert np.array_equal(ta[k][t], tb[k][t], equal_nan=True), (k, t, int((ta[k][t] != tb[k][t]).sum()))
        assert np.array_equal(pa[k], pb[k]), k
    assert np.array_equal(ca, cb, equal_nan=True)
    assert np.isfinite(ta[-1]["vmap_g0"]).sum() > W * H        # the model-side maps hold a surface


@pytest.mark.parametrize("size", [(640, 480), (200, 152), (168, 136)], ids=["vga", "200x152", "168x136"])
def test_frame_pyramid_launch_equals_the_level_kernels(hip, size):
    """k_frame_pyramid (Model::generateCUDATextures in one LDS-tiled launch: pyrDownGaussF x 2 + createVMap / createNMap x 3, Model.cpp:350-389) against
    the level-by-level kernels (mf_k_pyrdown_f, mf_k_vmap_nmap -- each pinned to the oracle / the reference's vectors in test_gpu_kernels.py and
    test_gpu_ref_golden.py) on the context's own filtered depth: all six maps the same bits.  Round 6 gave the launch an unrolled branch-free
    interior path for the 5 x 5 window; the border quirk, holes (NaN-skipping: the filtered depth of a hole is 0, the vertex map's NaN
    never enters a pyramid level, so holes are cut into the level-0 depth as NaNs too) and sizes with partial tiles are what could tell them apart."""
    from gpu_util import dev, empty, host
    from maskfusion_amd import MaskFusion, synth
    W, H = size
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 19)
    for k in range(3):
        rgb, d, _ = st.frame(k)
        if k == 2:
            d = d.copy(); d[H // 3: H // 3 + 9, W // 4: W // 4 + 31] = np.nan; d[:3, :] = np.nan; d[:, W - 2:] = np.nan   # NaNs the filter passes on
        mf.processFrame(rgb, d, timestamp=k)
    depthF = mf.debugRead("depthF").reshape(H, W).copy()
    got = {f"{p}{i}": mf.debugRead(f"{p}{i}").copy() for p in ("vmap", "nmap") for i in range(3)}
    mf.close()
    assert np.isnan(depthF).any(), "the scenario must push NaNs through the pyramid"
    lvl = dev(depthF)
    for i in range(3):
        w, h = W >> i, H >> i
        v, n = empty((3, h, w)), empty((3, h, w))
        assert hip.mf_k_vmap_nmap(lvl.data_ptr(), v.data_ptr(), n.data_ptr(), w, h, f / (1 << i), f / (1 << i), (W / 2.0) / (1 << i), (H / 2.0) / (1 << i), 3.0, None) == 0
        assert np.array_equal(host(v).reshape(-1), got[f"vmap{i}"].reshape(-1), equal_nan=True), ("vmap", i)
        assert np.array_equal(host(n).reshape(-1), got[f"nmap{i}"].reshape(-1), equal_nan=True), ("nmap", i)
        if i < 2:
            nxt = empty((h >> 1, w >> 1))
            assert hip.mf_k_pyrdown_f(lvl.data_ptr(), nxt.data_ptr(), w, h, None) == 0
            host(nxt)   # (synchronises)
            lvl = nxt
