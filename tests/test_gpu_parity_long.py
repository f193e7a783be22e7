"""Round-3 parity gates on the workloads bench.py actually times ("Next round" item 1):

  * test_long_horizon_ate            600 frames of the noisy S1 stream (configs[1]'s own stream), HIP vs oracle on identical inputs:
                                     ATE RMSE < 1 mm (north star), per-frame deltas printed every 50 frames, first frame above 1 mm reported.
  * test_s2_eight_objects_*          the 8-object S2 scene (configs[3]) against OracleMM for 40 frames at modelSpawnOffset = 2: ids per
                                     frame, model count, label image, background pose -- standing objects (strict, every frame) and
                                     moving + tracked objects (what bench.py --config 2s / 3 run).
  * test_config4_dense_maps          configs[4] as specified: 1280x960, 32M / 4M surfel budgets pre-filled to >= 80 %, 4 objects, 5 frames
                                     against OracleMM on the same uploaded maps (counts exact, every surfel in its slot).

MF_PARITY_FRAMES=<n> shortens the 600-frame run (rehearsals against the CPU-executed kernels, MF_EMU=1)."""
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

SEG = dict(threshold=0.3, weightDistance=150.0, weightConvexity=2.8, morphEdgeIterations=0, morphMaskIterations=0, minRelSizeNew=0.004)


def _render_job(job):
    kw, k = job
    from maskfusion_amd import synth
    return synth.Stream(**kw).frame(k)


def render(kw, n):
    """n frames of synth.Stream(**kw), ray-cast on the host cores (spawned workers: this process may already hold a HIP context)."""
    from maskfusion_amd import synth
    workers = max(1, min(32, (os.cpu_count() or 1) - 1, n // 8))
    if workers <= 1:
        st = synth.Stream(**kw)
        return [st.frame(k) for k in range(n)]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(workers) as pool:
        return pool.map(_render_job, [(kw, k) for k in range(n)], chunksize=max(1, n // (4 * workers)))


def _last_step(log_row):
    """the Gauss-Newton step the LAST iteration of a frame applied, from the device's log of that iteration's reduced system (27 packed
    upper-triangle products of the 7-vector row in reduce.cu:378-411 order, then residual and inlier count)"""
    A, b = np.zeros((6, 6)), np.zeros(6)
    k = 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6:
                b[i] = log_row[k]
            else:
                A[i, j] = A[j, i] = log_row[k]
            k += 1
    x = np.linalg.solve(A, b)
    return float(np.linalg.norm(x[:3])), float(np.linalg.norm(x[3:]))


def test_long_horizon_ate(hip, oracle):
    """The bench's own workload (600 frames of S1, Kinect-like noise, one background model, geometric ICP) through both sides.

    What can be expected (seen first on the CPU-executed kernels, 60 frames): while the reference's fixed 4/5/10 iteration schedule
    CONVERGES (last step of a frame below ~1e-5 m / rad; inlier count stationary over its last iterations) the two trajectories agree to
    micrometres.  From the first frame on which it does NOT converge (frame 15 of this stream: the inlier count is still climbing by
    40-180 per iteration at the tenth level-0 iteration and the last step is ~1e-4 m) the truncated iteration is no longer a fixed
    point, the 1e-7 differences of two summation orders are amplified by what is left of the descent, and the trajectories separate by
    1-2 mm -- and come back together: both are tied to the same frames, each is ~13-20 mm from the ground truth.  So the gates are:
      * per frame 1e-4 m while every frame so far has converged;
      * north star: the ATE RMSE against the ground truth differs by < 1 mm between the two;
      * the ATE RMSE between the two trajectories is < 1 mm as well (first hardware run, 600 frames: 0.26 mm).
    The first unconverged frame and the largest separation are printed (DESIGN.md quotes them)."""
    from maskfusion_amd import MaskFusion, synth
    n = int(os.environ.get("MF_PARITY_FRAMES", "600"))
    kw = dict(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, noise=True)
    st = synth.Stream(**kw)
    frames = render(kw, n)
    cap = 1 << 21
    o = oracle.Oracle(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpWeight=100.0, capacity=cap, so3=0)
    m = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, numGSurfels=cap, enableMultipleModels=False)
    gp, op, gc, oc, last = [], [], [], [], [(0.0, 0.0)]
    for k, (rgb, depth, _) in enumerate(frames):
        o.process_frame(rgb, depth)
        m.processFrame(rgb, depth, timestamp=k)
        gp.append(m.getCurrPose()); op.append(o.pose)
        gc.append(m.getBackgroundModel().lastCount()); oc.append(o.count)
        if k > 0:
            last.append(_last_step(m.debugRead("icp_log")[18]))
            assert m.gnIllIterations(0) == 0, k     # every system of the run inside the solver's stated domain (finding F4)
    o.close(); m.close()
    gp, op, last = np.array(gp), np.array(op), np.array(last)
    gt = np.array([st.gt_pose(k) for k in range(n)])
    d = np.linalg.norm(gp[:, :3, 3] - op[:, :3, 3], axis=1)
    dR = np.abs(gp[:, :3, :3] - op[:, :3, :3]).max(axis=(1, 2))
    for k in range(0, n, 50):
        print(f"frame {k:4d}: |t_hip - t_oracle| {d[k] * 1e3:.4f} mm, max |dR| {dR[k]:.2e}, last step {last[k][0]:.1e} m {last[k][1]:.1e} rad, "
              f"surfels hip/oracle {gc[k]}/{oc[k]}")
    unconverged = (last[:, 0] > 2e-5) | (last[:, 1] > 2e-5)
    first_unc = int(np.argmax(unconverged)) if unconverged.any() else n
    above = np.nonzero(d > 1e-3)[0]
    ate = synth.ate_rmse(gp, op)
    ate_g, ate_o = synth.ate_rmse(gp, gt), synth.ate_rmse(op, gt)
    print(f"{n} frames: ATE RMSE hip vs oracle {ate * 1e3:.4f} mm (max per-frame {d.max() * 1e3:.4f} mm at frame {int(d.argmax())}); "
          f"vs GT: hip {ate_g * 1e3:.3f} mm, oracle {ate_o * 1e3:.3f} mm; first frame above 1 mm: {int(above[0]) if len(above) else None}; "
          f"first frame whose Gauss-Newton loop did not converge (last step > 2e-5): {first_unc}, {int(unconverged.sum())} of {n} frames did not; "
          f"max |t_hip - t_oracle| before it: {d[:first_unc].max() * 1e3:.4f} mm; final surfels hip/oracle {gc[-1]}/{oc[-1]}")
    assert first_unc >= min(n, 5), "the stream must start with converging frames"
    assert d[:first_unc].max() < 1e-4          # float noise while the iteration is a fixed point (MI355X: 0.0017 mm over the first 7 frames)
    assert abs(ate_g - ate_o) < 1e-3           # north star: ATE RMSE (vs the ground truth) delta < 1 mm (MI355X, 600 frames: 20.300 vs 20.291 mm)
    assert ate < 1e-3                          # ... and the two trajectories themselves within 1 mm RMSE (MI355X, 600 frames: 0.26 mm; largest
    #                                            single-frame separation 1.96 mm at frame 18, in the wake of the unconverged frames 15-17)
    rel = np.abs(np.array(gc, float) - np.array(oc, float)) / np.maximum(np.array(oc, float), 1.0)
    print("surfel count relative difference: max %.4f at frame %d" % (rel.max(), int(rel.argmax())))
    assert rel.max() < 1e-2


def test_long_horizon_ate_reference_defaults(hip, oracle):
    """The configuration MaskFusion ships with (GUI/Tools/GUI.h:189,195: icpWeight 20 -> photometric term on, two launches per Gauss-Newton
    iteration; SO(3) pre-alignment on) over 200 frames of the VGA S1 stream, HIP vs oracle on identical inputs, with the gates of
    test_long_horizon_ate: ATE RMSE between the two trajectories < 1 mm, ATE against the ground truth within 1 mm of each other, surfel counts
    within 1 %; per frame 1e-4 m over the first frames.  Rows a8-a10 of SURVEY.md 8a (computeRgbResidual, rgbStep, so3Step) at full size and
    over a horizon (their kernel-level tests are bit-exact on single calls, tests/test_gpu_ref_golden.py)."""
    from maskfusion_amd import MaskFusion, synth
    n = int(os.environ.get("MF_PARITY_RGBD_FRAMES", "200"))
    kw = dict(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, noise=True)
    st = synth.Stream(**kw)
    frames = render(kw, n)
    cap = 1 << 21
    o = oracle.Oracle(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpWeight=20.0, capacity=cap, so3=1)
    m = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=20.0, so3=True, numGSurfels=cap, enableMultipleModels=False)
    gp, op, gc, oc, so3_its, rgb_cnt = [], [], [], [], [], []
    for k, (rgb, depth, _) in enumerate(frames):
        o.process_frame(rgb, depth)
        m.processFrame(rgb, depth, timestamp=k)
        gp.append(m.getCurrPose()); op.append(o.pose)
        gc.append(m.getBackgroundModel().lastCount()); oc.append(o.count)
        if k > 0:
            s = m.trackStats(0)
            so3_its.append(s["so3Iterations"]); rgb_cnt.append(s["lastRGBCount"])
            assert s["rejected"] == 0, k
    o.close(); m.close()
    gp, op = np.array(gp), np.array(op)
    gt = np.array([st.gt_pose(k) for k in range(n)])
    d = np.linalg.norm(gp[:, :3, 3] - op[:, :3, 3], axis=1)
    for k in range(0, n, 25):
        print(f"frame {k:4d}: |t_hip - t_oracle| {d[k] * 1e3:.4f} mm, surfels hip/oracle {gc[k]}/{oc[k]}")
    ate = synth.ate_rmse(gp, op)
    ate_g, ate_o = synth.ate_rmse(gp, gt), synth.ate_rmse(op, gt)
    print(f"RGB-D + SO(3), {n} frames: ATE RMSE hip vs oracle {ate * 1e3:.4f} mm (max per-frame {d.max() * 1e3:.4f} mm at frame {int(d.argmax())}); "
          f"vs GT: hip {ate_g * 1e3:.3f} mm, oracle {ate_o * 1e3:.3f} mm; SO(3) iterations per frame {min(so3_its):.0f}..{max(so3_its):.0f}, photometric "
          f"correspondences {min(rgb_cnt):.0f}..{max(rgb_cnt):.0f}; final surfels hip/oracle {gc[-1]}/{oc[-1]}")
    assert min(so3_its) >= 1 and min(rgb_cnt) > 0        # the photometric term and the pre-alignment really ran on every frame
    assert d[:5].max() < 1e-4
    assert abs(ate_g - ate_o) < 1e-3
    assert ate < 1e-3
    rel = np.abs(np.array(gc, float) - np.array(oc, float)) / np.maximum(np.array(oc, float), 1.0)
    assert rel.max() < 1e-2


def _pair(oracle, kw, n_frames, track_all, cap_g=1 << 20, cap_o=1 << 18, spawn_offset=2, share_filter=True, edges=False):
    """share_filter: the oracle takes the product's filtered depth (the bilateral filter is compared on its own, test_gpu_kernels.py: a few
    ulp between v_exp_f32 and expf).  Without it the last bits of the filter move ~0.3-0.6 % of the label pixels of this 8-object scene from
    the very first segmentation on -- geometric-edge values within rounding of the 0.3 threshold, each flip re-cutting a thin component --
    which says nothing about the label stage; with it every later stage is compared on identical inputs."""
    n_frames = int(os.environ.get("MF_PARITY_MM_FRAMES", n_frames))
    from maskfusion_amd import MaskFusion, synth
    from oracle import mfo_mm
    st = synth.Stream(**kw)
    frames = render(kw, n_frames)
    cls = [0] + [41 + i for i in range(kw["n_objects"])]
    # confO = 0.01, confG = 10 (what bench.py --config 2s / 3 use)
    o = mfo_mm.OracleMM(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpWeight=100.0, so3=0, capacity=cap_g, capacityObject=cap_o,
                        modelSpawnOffset=spawn_offset, trackAllModels=int(track_all), seg=SEG, confGlobal=10.0, confObject=0.01)
    m = MaskFusion(st.W, st.H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, numGSurfels=cap_g, numOSurfels=cap_o,
                   enableMultipleModels=True, modelSpawnOffset=spawn_offset, trackAllModels=track_all, initConfidenceGlobal=10.0,
                   initConfidenceObject=0.01)
    for k, v in (("mfThreshold", SEG["threshold"]), ("mfWeightDistance", SEG["weightDistance"]), ("mfWeightConvexity", SEG["weightConvexity"]),
                 ("mfMorphEdgeIterations", 0), ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", SEG["minRelSizeNew"])):
        m.setParam(k, v)
    rec = []
    for k, (rgb, depth, mask) in enumerate(frames):
        m.processFrame(rgb, depth, mask=mask, classIDs=cls, timestamp=k)
        o.process_frame(rgb, depth, mask, cls, depth_filtered=m.debugRead("depthF") if share_filter else None)
        gm = m.getModels()
        extra = {}
        if edges and k > 0:   # the geometric edge map of both sides and where their binary edges differ (the label stage's input)
            ge, oe = m.debugRead("edge_map"), o.edge_map()
            flip = (ge > SEG["threshold"]) != (oe > SEG["threshold"])
            extra = dict(edge_flips=int(flip.sum()), flip_margin=float(np.abs(oe[flip] - SEG["threshold"]).max()) if flip.any() else 0.0,
                         edge_diff=float(np.nanmax(np.abs(np.nan_to_num(ge) - np.nan_to_num(oe)))))
        rec.append(dict(extra, o_ids=[o.model_id(i) for i in range(o.n_models)], g_ids=[x.getID() for x in gm],
                        o_cnt=[o.model_count(i) for i in range(o.n_models)], g_cnt=[x.lastCount() for x in gm],
                        seg_diff=float((o.segmentation() != m.downloadSegmentation()).mean()), bg_ill=m.gnIllIterations(0),
                        o_pose=[o.model_pose(i) for i in range(o.n_models)], g_pose=[x.getPose() for x in gm]))
    o.close(); m.close()
    return rec


def _report(rec):
    for k, r in enumerate(rec):
        dp = [float(np.abs(a - b).max()) for a, b in zip(r["o_pose"], r["g_pose"])]
        print(k, "ids oracle/hip", r["o_ids"], r["g_ids"], "label diff %.5f" % r["seg_diff"], "pose diff", [round(x, 6) for x in dp],
              "counts", r["o_cnt"], r["g_cnt"])


def test_s2_eight_objects_standing(hip, oracle):
    """8 instance-masked boxes standing still, object models follow the camera (the shipped default: static objects): the whole
    multi-model state machine at the scale of configs[3] -- compared strictly on every one of the 40 frames."""
    kw = dict(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, n_objects=8, noise=True, object_motion=0.0)
    rec = _pair(oracle, kw, 40, False)
    _report(rec)
    for k, r in enumerate(rec):
        assert r["o_ids"] == r["g_ids"], f"frame {k}"                                  # same models, same ids, same order
        assert r["bg_ill"] == 0, f"frame {k}"                                          # the background's systems stay inside the solver's domain (F4)
        assert r["seg_diff"] < 2e-3, f"frame {k}"
        for i in range(len(r["o_pose"])):
            assert np.abs(r["o_pose"][i] - r["g_pose"][i]).max() < 2e-4, (k, i)
        for a, b in zip(r["o_cnt"], r["g_cnt"]):
            assert abs(a - b) <= max(20, 0.01 * a), (k, a, b)
    assert len(rec[-1]["o_ids"]) >= 7, "the scenario must spawn the object models"


def test_s2_eight_objects_standing_own_filters(hip, oracle):
    """The same scene with each side running ITS OWN bilateral filter (`share_filter=False`): the two filters differ by a few ulp of exp
    (v_exp_f32 / exp2 against libm's expf, 2e-5 m at most, tests/test_gpu_kernels.py::test_bilateral).  Every other multi-model test isolates that
    by handing the oracle the product's filtered depth; this one runs the composition bilateral -> edges -> labels -> spawn on each side's own
    filter for 20 frames.  GATED: what the filters' last bits must not change -- the same models with the same ids in the same order on every
    frame, every pose within 2e-4.  REPORTED, not gated (round 4 gated label pixels at 3 % and surfel counts at 10 %, bounds that had followed
    two failed hardware runs; no bound can be derived: the geometric edge map, MfSegmentation.cpp:106-119, switches its concavity term on the
    SIGN of a dot product, so a last-bit difference moves an edge value by up to the whole term, not by a band around the threshold): the
    binary-edge flips and how far from the 0.3 threshold they occur, the label pixels and the surfels they move -- a flipped edge pixel re-cuts
    a thin component, which then changes owner as a whole."""
    kw = dict(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, n_objects=8, noise=True, object_motion=0.0)
    rec = _pair(oracle, kw, 20, False, share_filter=False, edges=True)
    _report(rec)
    worst_lab = worst_cnt = worst_margin = 0.0
    flips = 0
    for k, r in enumerate(rec):
        assert r["o_ids"] == r["g_ids"], f"frame {k}"
        for i in range(len(r["o_pose"])):
            assert np.abs(r["o_pose"][i] - r["g_pose"][i]).max() < 2e-4, (k, i)
        if "edge_flips" in r:
            flips += r["edge_flips"]; worst_margin = max(worst_margin, r["flip_margin"])
        worst_lab = max(worst_lab, r["seg_diff"])
        for a, b in zip(r["o_cnt"], r["g_cnt"]):
            worst_cnt = max(worst_cnt, abs(a - b) / max(a, 1))
    print("own filters, 20 frames (report): %d binary-edge flips in all, the farthest %.3f from the threshold; they moved at most %.4f of the label "
          "pixels and %.4f of a model's surfels" % (flips, worst_margin, worst_lab, worst_cnt))
    assert len(rec[-1]["o_ids"]) >= 6


def test_s2_eight_objects_tracked(hip, oracle):
    """The scene bench.py --config 2s / --config 3 time: 8 moving boxes, trackAllModels.

    Each object's ICP runs on 2-8 k surfels of two or three planar box faces: its 6x6 system is ill-conditioned from the spawn on, and the
    1e-7 difference between two summation orders grows to millimetres within two frames and to the 0.2 m jump rule's threshold within ten
    (first hardware run: object 1 differs by 2.8 mm at frame 4, 1.5 cm at frame 10, is dropped by one side at frame 16).  From then on the
    two model lists are different scenes.  What is comparable, and gated:
      * the background pose on EVERY frame (2e-4; it never notices the objects) and a unique, background-first id list;
      * model list (ids, order) and label image (5e-3) for as long as every object pose agrees within 1 cm -- at least the first 4 frames
        (how long that lasts is itself chaotic: 13 frames in round 3, 7 in round 4; test_s2_eight_objects_tracked_teacher_forced compares
        every pass of all 40 frames of this scene with the chaos taken out);
      * the same ORDER of magnitude of models at the end (objects lost and re-spawned on both sides: the "14 models" of round 2's
        --config 2s against the "8" of --config 3 were this, 4 000 against 120 frames of drops and re-spawns, not an implementation difference).
    The standing-object test above is the strict one."""
    kw = dict(W=640, H=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, n_objects=8, noise=True, object_motion=1.0)
    rec = _pair(oracle, kw, 40, True)
    _report(rec)
    same_ids = [r["o_ids"] == r["g_ids"] for r in rec]
    first_split = same_ids.index(False) if False in same_ids else None
    print("model lists identical on", sum(same_ids), "of", len(rec), "frames; first difference at frame", first_split,
          "; final model count oracle/hip", len(rec[-1]["o_ids"]), len(rec[-1]["g_ids"]))
    comparable = True
    n_comparable = 0
    for k, r in enumerate(rec):
        assert np.abs(r["o_pose"][0] - r["g_pose"][0]).max() < 2e-4, f"background pose, frame {k}"
        assert r["bg_ill"] == 0, f"frame {k}"
        assert r["g_ids"][0] == 0 and len(set(r["g_ids"])) == len(r["g_ids"])
        if comparable and r["o_ids"] == r["g_ids"]:
            dp = [float(np.abs(a - b).max()) for a, b in zip(r["o_pose"][1:], r["g_pose"][1:])]
            comparable = all(x < 1e-2 for x in dp)
        else:
            comparable = False
        if comparable:
            n_comparable += 1
            assert r["seg_diff"] < 5e-3, f"frame {k}"
    print("frames on which every object pose agreed within 1 cm (report):", n_comparable)
    # How long the two free-running sides stay comparable moves with the last bits of either (13 frames at the end of round 3, 7 after round 4's
    # changes to the Gauss-Newton prologue's summation order) and is REPORTED, not gated: the strict comparison of this scene -- every pass of
    # every frame, every one of the 19 Gauss-Newton iterations of every tracked model -- is the teacher-forced test below.  What is asserted
    # here does not depend on those bits: the background (above, every frame), and that the first frames -- before any object has been
    # tracked twice -- agree.
    assert n_comparable >= 2


def _unpack_system(row):
    """(A 6x6, b 6) of a reduced Gauss-Newton system in the packed order of reduce.cu:378-411 (A[i][i..5], b[i] per row i)"""
    A, b = np.zeros((6, 6)), np.zeros(6)
    k = 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6:
                b[i] = row[k]
            else:
                A[i, j] = A[j, i] = row[k]
            k += 1
    return A, b


def _gn_update(oracle, resultRt, A, b, Rprev, tprev):
    """one Gauss-Newton update as RGBDOdometry.cpp:447-474 performs it, with the oracle's own pieces (Eigen-style LDLT in double,
    OdometryProvider::computeUpdateSE3) on a system handed in: -> (resultRt', Rcurr', tcurr')"""
    L = oracle.lib()
    x = np.zeros(6)
    L.mfo_ldlt_solve(np.ascontiguousarray(A.reshape(-1)), np.ascontiguousarray(b), x, 6)
    rt = np.ascontiguousarray(resultRt.reshape(-1).copy())
    L.mfo_update_se3(rt, x)
    rt = rt.reshape(4, 4)
    trR, trt = rt[:3, :3].astype(np.float32), rt[:3, 3].astype(np.float32)      # Isometry3f transform
    iR = trR.T
    it3 = -(iR @ trt)
    return rt, (Rprev.astype(np.float32) @ iR), (Rprev.astype(np.float32) @ it3 + tprev.astype(np.float32))


def _log_system_diff(dev_row, orc_row):
    """(inliers device, inliers oracle, max |A, b difference| relative to the largest entry) of one logged Gauss-Newton system"""
    scale = max(1e-30, float(np.abs(orc_row[:27]).max()))
    return int(dev_row[28]), int(orc_row[28]), float(np.abs(dev_row[:27] - orc_row[:27]).max() / scale)


class _StepStats:
    """running figures of _check_tracking_step over a test"""
    def __init__(self):
        self.worst_it0 = self.worst_sys = self.worst_rt = self.worst_rc = 0.0
        self.n_systems = self.n_inlier_flips = 0


def _check_tracking_step(oracle, m, o, i, tag, trace, start_pose, pose, stats):
    """One tracked model's Gauss-Newton loop of the frame just processed, iteration by iteration on equal input (RGBDOdometry.cpp:339-474):
    the first logged system against the oracle's own first system (inliers exact, A / b 2e-4), the oracle's normal equations at the device's
    pose of each of the 19 iterations (OracleMM.set_probe_poses; inliers within two gate-boundary pixels, A / b 2e-4 of the largest entry),
    every one of the device's 19 updates against the oracle's solve + computeUpdateSE3 of the device's own fp64 system (resultRt 1e-9,
    Rcurr / tcurr 2e-6), the last one landing on the model's pose.  Returns the inlier count of the first system."""
    dl, ol = m.debugRead("icp_log", model=i), o.model_track_log(i)
    assert len(ol) == 19
    gi, oi, rel = _log_system_diff(dl[0], ol[0])
    assert gi == oi, (tag, i, "inliers of the first Gauss-Newton system", gi, oi)
    assert rel < 2e-4, (tag, i, rel)
    stats.worst_it0 = max(stats.worst_it0, rel)
    pl = o.model_probe_log(i)
    assert len(pl) == 19, (tag, i, len(pl))
    for it in range(19):
        gi_k, oi_k, rel_k = _log_system_diff(trace[it, :32].astype(np.float32), pl[it])
        # (the oracle evaluates on ITS model-side maps, which equal the device's to 2e-6 relative, not bit for bit -- tests/test_gpu_kernels.py
        # -- so a pixel within rounding of the 0.10 m / 20 degree gates may fall on the other side: at most two of them per system;
        # the first hardware run had ONE such pixel, 59 806 against 59 805 inliers, in ~5 000 systems)
        assert abs(gi_k - oi_k) <= 2, (tag, i, it, "inliers", gi_k, oi_k)
        stats.n_inlier_flips += int(gi_k != oi_k)
        assert rel_k < 2e-4, (tag, i, it, rel_k)
        stats.worst_sys = max(stats.worst_sys, rel_k)
        stats.n_systems += 1
    Rprev, tprev = start_pose[:3, :3], start_pose[:3, 3]            # the pose the step started from
    for it in range(19):
        A, b = _unpack_system(trace[it, :32])
        rt, Rc, tc = _gn_update(oracle, trace[it, 32:48].reshape(4, 4), A, b, Rprev, tprev)
        scale = max(1.0, float(np.abs(rt).max()))
        d_rt = float(np.abs(rt.reshape(-1) - trace[it + 1, 32:48]).max()) / scale
        d_rc = max(float(np.abs(Rc.reshape(-1) - trace[it + 1, 48:57]).max()), float(np.abs(tc - trace[it + 1, 57:60]).max()))
        assert d_rt < 1e-9 and d_rc < 2e-6, (tag, i, it, d_rt, d_rc)
        stats.worst_rt, stats.worst_rc = max(stats.worst_rt, d_rt), max(stats.worst_rc, d_rc)
    assert np.abs(trace[19, 48:57].reshape(3, 3) - pose[:3, :3]).max() < 1e-7 and np.abs(trace[19, 57:60] - pose[:3, 3]).max() < 1e-7, (tag, i)
    return gi


def test_s2_eight_objects_tracked_teacher_forced(hip, oracle):
    """The tracked 8-object scene with the chaos taken out: after every frame the oracle is handed the product's
    model list and poses (OracleMM.force_tracking: its own tracking steps still run from its own -- identical -- state and stay readable, its
    drop decisions follow the list) and the product's filtered depth.  Every pass of every frame is then compared on equal input for all
    40 frames, whatever the ill-conditioned object trackers do:
      * model list (ids, order, class) identical on every frame;
      * surfel count of every model EXACT and every surfel in the same slot (position / normal / radius 1e-5, confidence 1e-4 rel,
        colour and time stamps exact) -- checked on every frame for the objects, every fifth for the background;
      * label image within 1e-3 of the pixels (a handful of prediction pixels where two coincident surfels tie in depth decide differently);
      * the tracking STEP of every tracked model, iteration by iteration (round 5; RGBDOdometry.cpp:339-474).  The device traces every
        iteration of its Gauss-Newton loop (debug tap "gn_trace": the reduced system as summed in fp64 and the state the iteration used);
        the oracle evaluates ITS normal equations at the device's pose of each of the 19 iterations (OracleMM.set_probe_poses) on its own,
        maps (identical to 2e-6): inlier count within two pixels and A, b within 2e-4 of the largest entry for ALL 19 systems of every tracked model; and every
        one of the device's 19 updates equals the oracle's solve + computeUpdateSE3 of the device's own system (resultRt 1e-9, Rcurr / tcurr
        2e-6), the last one landing on the model's pose.  A 2-3-face box seen in 2-8 k pixels amplifies 1e-7 of summation noise to millimetres
        over 19 free-running iterations (printed: the oracle's own step and its sensitivity to a 1 um start shift), on the oracle's side
        exactly as on the device's -- compared iteration by iteration on equal input that chaos cannot compound, so the 5 cm sanity bound of
        round 4 is gone."""
    from maskfusion_amd import MaskFusion, synth
    from oracle import mfo_mm
    n_frames = int(os.environ.get("MF_PARITY_MM_FRAMES", 40))
    W, H = int(os.environ.get("MF_PARITY_MM_W", 640)), int(os.environ.get("MF_PARITY_MM_H", 480))
    f = 528.0 * W / 640.0
    kw = dict(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=8, noise=True, object_motion=1.0)
    st = synth.Stream(**kw)
    frames = render(kw, n_frames)
    cls = [0] + [41 + i for i in range(8)]
    cap_g, cap_o = 1 << 20, 1 << 18
    o = mfo_mm.OracleMM(W, H, f, f, W / 2.0, H / 2.0, icpWeight=100.0, so3=0, capacity=cap_g, capacityObject=cap_o, modelSpawnOffset=2, trackAllModels=1,
                        seg=SEG, confGlobal=10.0, confObject=0.01)
    m = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=cap_g, numOSurfels=cap_o, enableMultipleModels=True,
                   modelSpawnOffset=2, trackAllModels=True, initConfidenceGlobal=10.0, initConfidenceObject=0.01)
    for k, v in (("mfThreshold", SEG["threshold"]), ("mfWeightDistance", SEG["weightDistance"]), ("mfWeightConvexity", SEG["weightConvexity"]),
                 ("mfMorphEdgeIterations", 0), ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", SEG["minRelSizeNew"])):
        m.setParam(k, v)
    n_obj_steps, worst_obj, worst_lab, worst_cloud, dropped, max_models = 0, 0.0, 0.0, 0.0, 0, 0
    stats = _StepStats()
    prev_ids = [0]
    prev_pose = {}
    for k, (rgb, depth, mask) in enumerate(frames):
        m.processFrame(rgb, depth, mask=mask, classIDs=cls, timestamp=k)
        gm = m.getModels()
        ids = [x.getID() for x in gm]
        poses = [x.getPose() for x in gm]
        traces = [m.debugRead("gn_trace", model=i) for i in range(len(gm))]
        o.force_tracking(ids, poses)
        o.set_probe_poses(ids, [t[:, 48:60] for t in traces])        # the pose every iteration of every model ran at, on the device
        o.process_frame(rgb, depth, mask, cls, depth_filtered=m.debugRead("depthF"))
        o_ids = [o.model_id(i) for i in range(o.n_models)]
        assert ids == o_ids, (k, ids, o_ids)
        dropped += len([i for i in prev_ids if i not in ids])
        prev_ids = ids
        max_models = max(max_models, len(ids))
        assert m.gnIllIterations(0) == 0, k
        line = []
        for i, x in enumerate(gm):
            assert x.getClassID() == o.model_class(i), (k, i)
            gc, oc = x.lastCount(), o.model_count(i)
            assert gc == oc, (k, i, ids[i], gc, oc)
            if i > 0 or k < 10 or k % 5 == 0 or k == n_frames - 1:
                g, c = x.downloadMap(), o.model_surfels(i)
                if not np.array_equal(g[:, 4:8], c[:, 4:8]):
                    bad = np.nonzero((g[:, 4:8] != c[:, 4:8]).any(axis=1))[0]
                    print(f"frame {k}, model {i} (id {ids[i]}): {len(bad)} of {len(g)} surfels differ in colour / time stamps; first rows:")
                    for r in bad[:6]:
                        print("   slot", int(r), "device", g[r].tolist(), "\n             oracle", c[r].tolist())
                assert np.array_equal(g[:, 4:8], c[:, 4:8]), (k, i, "colour / time stamps")
                cols = [0, 1, 2, 8, 9, 10, 11]
                assert np.array_equal(np.isnan(g[:, cols]), np.isnan(c[:, cols])), (k, i, "NaN pattern")
                dpos = float(np.nan_to_num(np.abs(g[:, cols] - c[:, cols])).max()) if len(g) else 0.0
                dconf = float((np.abs(g[:, 3] - c[:, 3]) / np.maximum(1.0, np.abs(c[:, 3]))).max()) if len(g) else 0.0
                worst_cloud = max(worst_cloud, dpos)
                assert dpos < 1e-5 and dconf < 1e-4, (k, i, dpos, dconf)
            own, tracked = o.model_tracked_pose(i)
            if not tracked:
                continue          # spawned in this frame
            gi = _check_tracking_step(oracle, m, o, i, k, traces[i], prev_pose[ids[i]], poses[i], stats)
            dstep = float(np.abs(own - poses[i]).max())
            if i == 0:
                assert dstep < 1e-5, (k, dstep)
            else:
                n_obj_steps += 1
                worst_obj = max(worst_obj, dstep)
                line.append("%d: %.1e (probe %.1e, %d inl)" % (ids[i], dstep, o.model_step_sensitivity(i), gi))
        lab = float((o.segmentation() != m.downloadSegmentation()).mean())
        worst_lab = max(worst_lab, lab)
        assert lab < 1e-3, (k, lab)
        prev_pose = {i_: p_ for i_, p_ in zip(ids, poses)}
        print("frame %2d: %d models, label diff %.2e; free-running object steps |device - oracle's own| (probe = oracle's step under a 1 um start shift): %s"
              % (k, len(ids), lab, "; ".join(line)))
    o.close(); m.close()
    print("teacher-forced: %d frames, up to %d models, %d drops followed, %d object tracking steps compared; %d Gauss-Newton systems compared iteration by "
          "iteration (%d of them with an inlier count off by one or two): worst A / b difference %.2e, worst update: resultRt %.2e, Rcurr / tcurr %.2e; "
          "free-running object step (report) %.2e; label image %.2e, cloud %.2e"
          % (n_frames, max_models, dropped, n_obj_steps, stats.n_systems, stats.n_inlier_flips, stats.worst_sys, stats.worst_rt, stats.worst_rc, worst_obj, worst_lab,
             worst_cloud))
    assert max_models >= 8 and n_obj_steps >= 4 * (n_frames - 6)


def _config4_dense_maps(hip, oracle, tracked):
    from maskfusion_amd import stress
    from oracle import mfo_mm
    n_dense = int(os.environ.get("MF_PARITY_C4_FRAMES", 5))
    num_g, num_o = int(os.environ.get("MF_PARITY_C4_GSURFELS", stress.NUM_GSURFELS)), int(os.environ.get("MF_PARITY_C4_OSURFELS", stress.NUM_OSURFELS))
    scale = int(os.environ.get("MF_PARITY_C4_SCALE", 1))          # > 1: the scenario at 1 / scale of the resolution (tests/test_emu_dense_maps.py)
    n_obj = int(os.environ.get("MF_PARITY_C4_OBJECTS", 4))
    st = stress.stream(n_obj, scale=scale)
    kw = stress.stream_kwargs(n_obj, scale=scale)
    max_lead = int(os.environ.get("MF_PARITY_C4_LEADIN", 16))
    frames = render(kw, max_lead + n_dense)
    cls = [0] + [41 + i for i in range(n_obj)]
    W, H, f = st.W, st.H, st.fx
    o = mfo_mm.OracleMM(W, H, f, f, W / 2.0, H / 2.0, icpWeight=100.0, so3=0, capacity=stress.surfel_capacity(num_g), capacityObject=stress.surfel_capacity(num_o),
                        modelSpawnOffset=2, trackAllModels=0, seg=SEG, confGlobal=10.0, confObject=0.01)
    m = stress.make_context(0, num_g, num_o, n_objects=n_obj, scale=scale)
    if os.environ.get("MF_PARITY_C4_FORMS") == "big":              # small budgets, the passes of FULL maps (run table + culling, in-place update and clean)
        m.setParam("bigMapElements", 0)
        m.setParam("inPlaceElements", 0)

    def oracle_frame(k, rgb, depth, mask):
        gm = m.getModels()
        ids = [x.getID() for x in gm]
        o.force_tracking(ids, [x.getPose() for x in gm])
        if tracked:          # the pose every iteration of every model ran at, on the device (debug tap "gn_trace")
            o.set_probe_poses(ids, [m.debugRead("gn_trace", model=i)[:, 48:60] for i in range(len(gm))])
        o.process_frame(rgb, depth, mask, cls, depth_filtered=m.debugRead("depthF"))

    k0, loaded = stress.lead_in(m, st, frames, cls, n_objects=n_obj, on_frame=oracle_frame, on_upload=lambda i, s: o.upload_map(i, s), log=print, max_frames=max_lead)
    assert len(loaded) == n_obj + 1, loaded                           # the background and four object models, each on its own box
    assert loaded[0] >= 0.8 * stress.surfel_capacity(num_g) and min(loaded[i] for i in range(1, n_obj + 1)) >= 0.8 * stress.surfel_capacity(num_o)
    if tracked:              # S3 = S2's settings: every object model is tracked from here on (Model::makeNonStatic, Model.h:263-268)
        stress.track_objects(m, on_model=lambda i: o.make_nonstatic(i))
    worst_lab, n_obj_steps, worst_obj = 0.0, 0, 0.0
    stats = _StepStats()
    prev_pose = {x.getID(): x.getPose() for x in m.getModels()}
    for k in range(k0, k0 + n_dense):
        rgb, depth, mask = frames[k]
        m.processFrame(rgb, depth, mask=mask, classIDs=cls, timestamp=k)
        oracle_frame(k, rgb, depth, mask)
        gm = m.getModels()
        ids, o_ids = [x.getID() for x in gm], [o.model_id(i) for i in range(o.n_models)]
        assert ids == o_ids, (k, ids, o_ids)
        gc, oc = [x.lastCount() for x in gm], [o.model_count(i) for i in range(o.n_models)]
        lab = float((o.segmentation() != m.downloadSegmentation()).mean())
        worst_lab = max(worst_lab, lab)
        own, was_tracked = o.model_tracked_pose(0)
        dstep = float(np.abs(own - gm[0].getPose()).max())
        print(f"dense frame {k}: ids {ids}, surfels hip {gc} oracle {oc}, label diff {lab:.2e}, background step |device - oracle's own| {dstep:.1e}")
        assert gc == oc, (k, gc, oc)
        assert lab < 1e-3, (k, lab)
        assert was_tracked and dstep < 1e-5, (k, dstep)
        assert m.gnIllIterations(0) == 0, k
        if tracked:          # all 19 systems and all 19 updates of ALL tracked models (the background and every object) on equal input
            line = []
            for i, x in enumerate(gm):
                own_i, tr_i = o.model_tracked_pose(i)
                if not tr_i or ids[i] not in prev_pose:
                    continue      # spawned in this frame
                assert i == 0 or x.isNonstatic(), (k, i)
                pose = x.getPose()
                gi = _check_tracking_step(oracle, m, o, i, k, m.debugRead("gn_trace", model=i), prev_pose[ids[i]], pose, stats)
                if i > 0:
                    n_obj_steps += 1
                    d_own = float(np.abs(own_i - pose).max())
                    worst_obj = max(worst_obj, d_own)
                    line.append("%d: %.1e (%d inl, %d ill)" % (ids[i], d_own, gi, m.gnIllIterations(i)))
            print("   tracked objects, free-running step |device - oracle's own| (inliers of the first system, iterations outside the solver's domain): " + "; ".join(line))
        prev_pose = {x.getID(): x.getPose() for x in gm}
    assert gc[0] >= 0.8 * stress.surfel_capacity(num_g)               # the maps stayed dense through the clean passes
    for i, x in enumerate(m.getModels()):
        g, c = x.downloadMap(), o.model_surfels(i)
        assert g.shape == c.shape, (i, g.shape, c.shape)
        assert np.array_equal(g[:, 4:8], c[:, 4:8]), (i, "colour / time stamps")
        cols = [0, 1, 2, 8, 9, 10, 11]
        dpos = 0.0
        for q in range(0, len(g), 1 << 22):      # in pieces: 26 M x 12 floats per side are enough to hold
            a, b = g[q:q + (1 << 22)], c[q:q + (1 << 22)]
            assert np.array_equal(np.isnan(a[:, cols]), np.isnan(b[:, cols])), (i, "NaN pattern")
            dpos = max(dpos, float(np.nan_to_num(np.abs(a[:, cols] - b[:, cols])).max()))
            dconf = float((np.abs(a[:, 3] - b[:, 3]) / np.maximum(1.0, np.abs(b[:, 3]))).max())
            assert dconf < 1e-5, (i, dconf)
        print(f"model {i} (id {x.getID()}): {len(g)} surfels, every one in its slot; max |position / normal / radius difference| {dpos:.2e}")
        assert dpos < 1e-6, (i, dpos)
    n_models_end = len(m.getModels())
    o.close(); m.close()
    print(f"configs[4] dense{' (objects tracked)' if tracked else ''}: {n_dense} frames at {gc[0]} + {gc[1:]} surfels, worst label difference {worst_lab:.2e}")
    if tracked:
        print("   %d object tracking steps, %d Gauss-Newton systems compared iteration by iteration (%d with an inlier count off by one or two): worst A / b "
              "difference %.2e (first system %.2e), worst update: resultRt %.2e, Rcurr / tcurr %.2e; free-running object step (report) %.2e"
              % (n_obj_steps, stats.n_systems, stats.n_inlier_flips, stats.worst_sys, stats.worst_it0, stats.worst_rt, stats.worst_rc, worst_obj))
        # every object is tracked on every dense frame unless the 0.2 m rule (MaskFusion.cpp:268-272) dropped it -- on both sides, the list is compared above
        if scale == 1:       # (the quarter-resolution rehearsal on the CPU-executed kernels sees its ~1 k-pixel boxes dropped: both sides, same frames)
            assert n_models_end == n_obj + 1 and n_obj_steps == n_obj * n_dense, (n_models_end, n_obj_steps)


def test_config4_dense_maps(hip, oracle):
    """configs[4] as BASELINE.json / SURVEY.md 8d S3 define it: 1280x960, MASKFUSION_NUM_GSURFELS = 32M / NUM_OSURFELS = 4M (capacities 5760^2 /
    2048^2, Model.cpp:101-108), 4 object models, every map pre-filled to >= 80 % of its capacity (26.5 M background surfels, 3.4 M per
    object: maskfusion_amd/stress.py loads generated maps where a long orbit would have grown them), then 5 frames against OracleMM fed the
    same frames, the same uploaded maps, the product's filtered depth and -- teacher forcing, as in the tracked 8-object test -- the
    product's poses, so that every surfel pass of every frame runs on equal input at the budgets the reference is compiled with
    (Core/CMakeLists.txt:27-28).  Gated on every dense frame: model list, surfel count of every model EXACT, label image, the background's own
    tracking step; on the last frame every surfel of every model in its slot (position / normal / radius 1e-6, confidence 1e-5 rel,
    colour and time stamps exact).  Here the objects stand and follow the camera (trackAllModels off); the next test tracks them."""
    _config4_dense_maps(hip, oracle, tracked=False)


def test_config4_dense_maps_tracked(hip, oracle):
    """S3 as SURVEY.md 8d defines it -- "S2 with 4 objects" at 1280x960, and S2 tracks every model: after the lead-in every object model is made
    non-static (Model.h:263-268) on both sides, so that each dense frame runs FIVE Gauss-Newton loops (MaskFusion.cpp:263-276,
    RGBDOdometry.cpp:227-497: the background's and the four objects', each against a 3.4 M-surfel map's prediction at 1280x960) through the
    batched kernels.  Teacher-forced like test_s2_eight_objects_tracked_teacher_forced: on every dense frame all 19 systems of all 5 models
    (inliers within two gate-boundary pixels, A / b 2e-4) and all 19 updates (resultRt 1e-9) on equal input, beside test_config4_dense_maps'
    gates (model list, every count exact, label image; every surfel of every model in its slot on the last frame)."""
    _config4_dense_maps(hip, oracle, tracked=True)
