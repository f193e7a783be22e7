"""Run by tests/test_emu_smoke.py in a subprocess: a few frames of a small synthetic stream through the product's kernels EXECUTED ON
THE CPU (tests/hipcpu), next to the oracle.  Prints one JSON line.  TEST TOOLING."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402

import emu  # noqa: E402

emu.activate()
from maskfusion_amd import MaskFusion, synth  # noqa: E402
from oracle import mfo, mfo_mm  # noqa: E402


def single_model(n=4, W=160, H=120):
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    o = mfo.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 17, icpWeight=100.0, so3=0)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 17)
    out = []
    for k in range(n):
        rgb, d, _ = st.frame(k)
        mf.processFrame(rgb, d, timestamp=k)
        o.process_frame(rgb, d)
        out.append(dict(count=int(mf.getBackgroundModel().lastCount()), ocount=int(o.count), pose_diff=float(np.abs(mf.getCurrPose() - o.pose).max()),
                        inliers=float(mf.trackStats(0)["lastICPCount"]), pose=mf.getCurrPose().reshape(-1).tolist()))
    mf.close(); o.close()
    return out


def rgbd_so3(n=3, W=160, H=120):
    """the GUI configuration: photometric term at weight 20 + SO(3) pre-alignment (two launches per iteration, the RGB-D kernels)"""
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    o = mfo.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 17, icpWeight=20.0, so3=1)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=20.0, so3=True, enableMultipleModels=False, numGSurfels=1 << 17)
    out = []
    for k in range(n):
        rgb, d, _ = st.frame(k)
        mf.processFrame(rgb, d, timestamp=k)
        o.process_frame(rgb, d)
        s = mf.trackStats(0)
        out.append(dict(count=int(mf.getBackgroundModel().lastCount()), ocount=int(o.count), pose_diff=float(np.abs(mf.getCurrPose() - o.pose).max()),
                        rgb_count=float(s["lastRGBCount"]), so3_iterations=float(s["so3Iterations"])))
    mf.close(); o.close()
    return out


def bad_depth_pixels(n=4, W=160, H=120):
    """sensor garbage in the depth image -- NaN, +inf and negative patches: the filter includes every in-image tap like the shader does
    (depth_bilateral_metric.frag:30-76: a NaN tap poisons its 13x13 neighbourhood, which then fails the `z > 0` tests downstream), so the
    frame loses those regions and nothing else, on both sides"""
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    o = mfo.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 17, icpWeight=100.0, so3=0)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 17)
    out = []
    for k in range(n):
        rgb, d, _ = st.frame(k)
        d = d.copy()
        d[10:14, 20:30] = np.nan
        d[60:63, 100:104] = np.inf
        d[90:94, 40:48] = -1.0
        mf.processFrame(rgb, d, timestamp=k)
        o.process_frame(rgb, d)
        gF, oF = mf.debugRead("depthF"), o.dbg("depthF")
        pose = mf.getCurrPose()
        out.append(dict(count=int(mf.getBackgroundModel().lastCount()), ocount=int(o.count), pose_finite=bool(np.isfinite(pose).all()),
                        pose_diff=float(np.abs(pose - o.pose).max()), nan_pattern_equal=bool(np.array_equal(np.isnan(gF), np.isnan(oF))),
                        nan_pixels=int(np.isnan(gF).sum()),
                        filtered_diff=float(np.nanmax(np.abs(gF - oF)))))
    mf.close(); o.close()
    return out


def schedule_switches(n=5, W=160, H=120):
    """the iteration schedule {fastOdom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0} (RGBDOdometry.cpp:327-329; -fo / pyramid off) in its four
    combinations, the last one with the photometric term and SO(3): device loop against the oracle's"""
    f = 528.0 * W / 640.0
    out = {}
    for name, fast, pyr, icp, so3 in (("fast", 1, 1, 100.0, 0), ("nopyramid", 0, 0, 100.0, 0), ("fast_nopyramid", 1, 0, 100.0, 0), ("fast_rgbd_so3", 1, 1, 20.0, 1)):
        st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
        o = mfo.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 17, icpWeight=icp, so3=so3, fastOdom=fast, pyramid=pyr)
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=icp, so3=bool(so3), fastOdom=bool(fast), enableMultipleModels=False, numGSurfels=1 << 17)
        if not pyr:
            mf.setPyramid(0)
        rows = []
        for k in range(n):
            rgb, d, _ = st.frame(k)
            mf.processFrame(rgb, d, timestamp=k)
            o.process_frame(rgb, d)
            rows.append(dict(count=int(mf.getBackgroundModel().lastCount()), ocount=int(o.count), pose_diff=float(np.abs(mf.getCurrPose() - o.pose).max())))
        mf.close(); o.close()
        out[name] = rows
    return out


def multimodel_bad_depth(n=7, W=240, H=160):
    """the multi-model frame (global projection, label stage, spawn, per-model fusion) on the same kind of sensor garbage, some of it inside
    an object's mask; the oracle gets the device's filtered depth (the two filters differ by a few ulp of exp, which is the ONLY source of
    the 1 % count differences of freshly spawned objects): ids, surfel counts and label images must then be identical"""
    f = 198.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=2, noise=True, object_motion=0.0)
    seg_o = dict(threshold=0.3, weightDistance=150.0, weightConvexity=2.8, morphEdgeIterations=0, morphMaskIterations=0, minRelSizeNew=0.004)
    seg_d = dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0, newModelMinRelativeSize=0.004)
    o = mfo_mm.OracleMM(W, H, f, f, W / 2.0, H / 2.0, icpWeight=100.0, so3=0, capacity=1 << 17, capacityObject=1 << 15, modelSpawnOffset=2, trackAllModels=0, seg=seg_o)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 17, numOSurfels=1 << 15, enableMultipleModels=True,
                    modelSpawnOffset=2, trackAllModels=False)
    for k, v in seg_d.items():
        mf.setParam(k, v)
    out = []
    for k in range(n):
        rgb, d, m = st.frame(k)
        d = d.copy()
        d[10:14, 20:30] = np.nan
        d[100:103, 150:154] = np.inf
        d[120:124, 40:48] = -1.0
        ys, xs = np.where(m == 1)
        if len(ys) > 20:
            d[ys[:6], xs[:6]] = np.nan
        mf.processFrame(rgb, d, mask=m, classIDs=[0, 41, 42], timestamp=k)
        o.process_frame(rgb, d, m, [0, 41, 42], depth_filtered=mf.debugRead("depthF"))
        ms = mf.getModels()
        out.append(dict(ids=[x.getID() for x in ms], oids=[o.model_id(i) for i in range(o.n_models)], counts=[x.lastCount() for x in ms],
                        ocounts=[o.model_count(i) for i in range(o.n_models)], label_diff=int((mf.downloadSegmentation() != o.segmentation()).sum()),
                        pose_diff=max(float(np.abs(ms[i].getPose() - o.model_pose(i)).max()) for i in range(min(len(ms), o.n_models)))))
    mf.close(); o.close()
    return out


def weight_multiplier_cases(n=6, W=160, H=120):
    """processFrame's third argument (MaskFusion.cpp:200; Model::fuse weighting = computeFusionWeight(weightMultiplier), Model.cpp:449-464)
    at 0.3 and 3.0, poses given and filtered depth shared: every surfel in the same slot with the same confidence"""
    f = 528.0 * W / 640.0
    out = {}
    for wm in (0.3, 3.0):
        st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
        o = mfo.Oracle(W, H, f, f, W / 2.0, H / 2.0, capacity=1 << 17, icpWeight=100.0, so3=0, confGlobal=2.0)
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 17, initConfidenceGlobal=2.0)
        rows = []
        for k in range(n):
            rgb, d, _ = st.frame(k)
            T = st.gt_pose(k).astype(np.float32)
            mf.processFrame(rgb, d, timestamp=k, weightMultiplier=wm, inPose=T if k else None)
            o.process_frame(rgb, d, weight_multiplier=wm, in_pose=T if k else None, depth_filtered=mf.debugRead("depthF"))
            g, oc = mf.getBackgroundModel().downloadMap(), o.surfels()
            same = len(g) == len(oc)
            rows.append(dict(count=len(g), ocount=len(oc), max_diff=float(np.abs(g[:, :4] - oc[:, :4]).max()) if same else -1.0,
                             stamps_equal=bool(same and np.array_equal(g[:, 4:8], oc[:, 4:8])), conf_mean=float(g[:, 3].mean())))
        mf.close(); o.close()
        out[str(wm)] = rows
    return out


def device_resident_masks(n=8, W=240, H=160):
    """mf_process_frame_dev + mf_set_mask_class_ids (frames and masks already in device memory: what bench.py --config 2s times) against
    mf_process_frame with host pointers and FrameData::classIDs: the same multi-model run, bit for bit (tracked objects, spawns and drops)"""
    f = 198.0
    seg_d = dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0, newModelMinRelativeSize=0.004)

    def make():
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 17, numOSurfels=1 << 15, enableMultipleModels=True,
                        modelSpawnOffset=2, trackAllModels=True)
        for k, v in seg_d.items():
            mf.setParam(k, v)
        return mf

    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=2, noise=True, object_motion=1.0)
    a, b = make(), make()
    b.setMaskClassIDs([0, 41, 42])
    keep, out = [], []
    for k in range(n):
        rgb, d, m = st.frame(k)
        a.processFrame(rgb, d, mask=m, classIDs=[0, 41, 42], timestamp=k)
        bufs = (np.ascontiguousarray(rgb), np.ascontiguousarray(d, np.float32), np.ascontiguousarray(m, np.uint8))
        keep.append(bufs)                      # ("device" pointers are host pointers under the CPU-executed kernels)
        b.processFrameDevice(bufs[0].ctypes.data, bufs[1].ctypes.data, bufs[2].ctypes.data, timestamp=k)
        b.sync()
        ma, mb = a.getModels(), b.getModels()
        out.append(dict(ids=[x.getID() for x in ma], ids_dev=[x.getID() for x in mb], classes=[x.getClassID() for x in ma], classes_dev=[x.getClassID() for x in mb],
                        counts_equal=[x.lastCount() for x in ma] == [x.lastCount() for x in mb],
                        poses_equal=bool(len(ma) == len(mb) and all(np.array_equal(x.getPose(), y.getPose()) for x, y in zip(ma, mb))),
                        label_diff=int((a.downloadSegmentation() != b.downloadSegmentation()).sum())))
    a.close(); b.close()
    return out


def static_switches(n=9, W=240, H=160):
    """Model::makeNonStatic / makeStatic / isNonstatic / updateStaticPose (Core/Model/Model.h:263-268; MaskFusion.cpp:263-276) with
    trackAllModels off: a static object follows the background -- pose_obj * pose_bg^-1 stays what makeStatic recorded -- a non-static one is
    tracked (and may fall to the 0.2 m jump rule, which replaces it by a fresh, static model)"""
    f = 198.0
    seg_d = dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0, newModelMinRelativeSize=0.004)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 17, numOSurfels=1 << 15, enableMultipleModels=True,
                    modelSpawnOffset=2, trackAllModels=False)
    for k, v in seg_d.items():
        mf.setParam(k, v)
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=1, noise=True, object_motion=0.0)
    rel, out = [], dict(flag_after_nonstatic=None, flag_after_static=None, tracked_or_replaced=None)
    for k in range(n):
        rgb, d, m = st.frame(k)
        ms = mf.getModels()
        id_before = ms[1].getID() if len(ms) > 1 else None
        if k == 5 and len(ms) > 1:
            ms[1].makeNonStatic()
            out["flag_after_nonstatic"] = bool(ms[1].isNonstatic())
        if k == 6 and len(ms) > 1:
            ms[1].makeStatic()
            out["flag_after_static"] = bool(ms[1].isNonstatic())
        mf.processFrame(rgb, d, mask=m, classIDs=[0, 41], timestamp=k)
        ms = mf.getModels()
        if k == 5 and len(ms) > 1:
            out["tracked_or_replaced"] = bool(ms[1].getID() != id_before or ms[1].isNonstatic())
        if len(ms) > 1:
            rel.append((k, ms[1].getID(), (ms[1].getPose() @ np.linalg.inv(ms[0].getPose())).reshape(-1).tolist()))
    mf.close()
    out["relative"] = rel
    return out


def host_paths(n=8, W=160, H=120):
    """mf_process_frame's host-side variants (pinned double buffer with ONE packed upload, hostLockstep, hostWaitUpload) against the blocking
    form of rounds 1-3: every variant is the same kernels on the same inputs, so poses, counts and the cloud's bytes must be identical.
    (Streams and events are synchronous here: this checks the bookkeeping -- slots, views into the packed block --, the overlap itself is
    tests/test_gpu_api.py on the MI355X.)"""
    import hashlib
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    frames = [st.frame(k) for k in range(n)]
    variants = {"blocking": {"hostInputAsync": 0}, "default": {}, "three_ahead": {"hostLockstep": 0}, "stream_wait": {"hostWaitUpload": 0}}
    out = {}
    for name, params in variants.items():
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 17)
        for k, v in params.items():
            mf.setParam(k, v)
        rgb_buf, d_buf = np.zeros((H, W, 3), np.uint8), np.zeros((H, W), np.float32)
        poses = []
        for k, (rgb, d, _) in enumerate(frames):
            rgb_buf[...] = rgb; d_buf[...] = d
            mf.processFrame(rgb_buf, d_buf, timestamp=k)
            rgb_buf[...] = 0; d_buf[...] = np.nan                       # the caller's buffers are the caller's again
            if k % 3 == 2:
                poses.append(mf.getCurrPose().reshape(-1).tolist())
        poses.append(mf.getCurrPose().reshape(-1).tolist())
        cloud = np.ascontiguousarray(mf.getBackgroundModel().downloadMap())
        out[name] = dict(poses=poses, count=int(mf.getBackgroundModel().lastCount()), cloud_sha1=hashlib.sha1(cloud.tobytes()).hexdigest())
        mf.close()
    return out


def map_forms(n=8, W=160, H=120):
    """The three size-dependent forms of a model's fuse / clean passes (mf_frame.inl: enqueue_fuse_clean; by default chosen by the map's size,
    here forced): copy-update + two-launch clean, in-place update + two-launch clean, in-place update + in-place clean on the buffer's runs
    (run table, culled projection passes, the buffer sparse) -- and a run that changes form from frame to frame, as a map does that grows across
    a threshold (the live buffer alternates or not, the run table comes and goes, a sparse buffer is compacted).  Same frames: poses, counts and the cloud's bytes identical."""
    import hashlib
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
    frames = [st.frame(k) for k in range(n)]
    forms = {"copy_two_launch": (1 << 30, 1 << 30), "in_place_two_launch": (1 << 30, 0), "in_place_runs": (0, 0), "changing": None}
    cycle = [(1 << 30, 1 << 30), (0, 0), (1 << 30, 0), (0, 0), (1 << 30, 1 << 30), (1 << 30, 0)]   # a map that crosses the thresholds from frame to frame
    out = {}
    for name, form in forms.items():
        mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=False, numGSurfels=1 << 17)
        poses = []
        for k, (rgb, d, _) in enumerate(frames):
            big, in_place = form if form else cycle[k % len(cycle)]
            mf.setParam("bigMapElements", big)
            mf.setParam("inPlaceElements", in_place)
            mf.processFrame(rgb, d, timestamp=k)
            poses.append(mf.getCurrPose().reshape(-1).tolist())
        taps = dict(visible_runs=int(mf.getParam("visibleRuns")), runs=int(mf.getParam("backgroundRuns")), clean_runs=int(mf.getParam("cleanRuns")))   # (before the download: it compacts)
        cloud = np.ascontiguousarray(mf.getBackgroundModel().downloadMap())
        out[name] = dict(poses=poses, count=int(mf.getBackgroundModel().lastCount()), cloud_sha1=hashlib.sha1(cloud.tobytes()).hexdigest(), **taps)
        mf.close()
    return out


def rgb_pyramid(n=3):
    """The frame's intensity pyramid + derivative / gate images: the one-launch form (k_rgb_pyramid, LDS-tiled) against the four single kernels it
    replaces (mf_set_param "fusedRgbPyramid" 0), on sizes with partial tiles and colour images with zero patches (pyrDownUcharGauss skips zero
    texels, the gate wants a 4 x 4 window > 0): every image of every level identical, and with them pose and map."""
    out = {}
    for (W, H) in ((160, 120), (200, 152), (136, 104)):
        f = 528.0 * W / 640.0
        st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True)
        frames = [st.frame(k) for k in range(n)]
        rgb2 = frames[-1][0].copy(); rgb2[10:30, 20:50] = 0; rgb2[H - 9:, :40] = 0; rgb2[:, W - 7:] = 0
        frames[-1] = (rgb2, frames[-1][1], frames[-1][2])
        res = []
        for fused in (0, 1):
            mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=20.0, so3=True, enableMultipleModels=False, numGSurfels=1 << 17)
            mf.setParam("fusedRgbPyramid", fused)
            taps = []
            for k, (rgb, d, _) in enumerate(frames):
                mf.processFrame(rgb, d, timestamp=k)
                taps.append([mf.debugRead(f"{p}{i}").copy() for p in ("gray", "dIdx", "dIdy", "rgb_gate") for i in range(3)])
            res.append((taps, mf.getCurrPose().copy(), mf.getBackgroundModel().downloadMap().copy()))
            mf.close()
        (ta, pa, ca), (tb, pb, cb) = res
        out[f"{W}x{H}"] = dict(images_equal=all(np.array_equal(x, y) for fa, fb in zip(ta, tb) for x, y in zip(fa, fb)),
                               pose_equal=bool(np.array_equal(pa, pb)), cloud_equal=bool(np.array_equal(ca, cb, equal_nan=True)),
                               gate_pixels=int(ta[-1][9].sum()), zero_texels=int((ta[-1][0] == 0).sum()))
    return out


if __name__ == "__main__":
    scenarios = dict(single=single_model, rgbd=rgbd_so3, bad_depth=bad_depth_pixels, schedule=schedule_switches, mm_bad_depth=multimodel_bad_depth,
                     weight=weight_multiplier_cases, dev_masks=device_resident_masks, static=static_switches, host_paths=host_paths, map_forms=map_forms, rgb_pyramid=rgb_pyramid)
    wanted = sys.argv[1:] or list(scenarios)      # (tests/test_gpu_emu_agrees.py asks for "single" only; the CPU suite runs all of them)
    print(json.dumps({k: scenarios[k]() for k in wanted}))
