"""Host-side mirror of the reference's MaskFusion / Model interface for the hot path.

Names and argument meaning follow Core/MaskFusion.h:45-307 and Core/Model/Model.h:108-268 of
martinruenz/maskfusion, over the C ABI in include/maskfusion_amd.h.  numpy arrays stand in for cv::Mat /
Eigen::Matrix4f.  Errors raise MFError (the reference throws std::runtime_error / exits on CUDA errors).
"""
from __future__ import annotations

import ctypes as C
import numpy as np

from .lib import load, MFError, Config, ModelInfo, MF_N_TIMINGS, TIMING_LABELS, MF_N_PASSES, PASS_LABELS


class Model:
    """View of one surfel model (Core/Model/Model.h)."""

    def __init__(self, owner: "MaskFusion", index: int):
        self._o = owner
        self._i = index

    def info(self) -> ModelInfo:
        out = ModelInfo()
        self._o._chk(self._o._L.mf_model_info(self._o._h, self._i, C.byref(out)))
        return out

    def getID(self) -> int:
        return int(self.info().id)

    def getClassID(self) -> int:
        return int(self.info().class_id)

    def getConfidenceThreshold(self) -> float:
        return float(self.info().confidence_threshold)

    def getPose(self) -> np.ndarray:
        out = np.zeros(16, np.float32)
        self._o._chk(self._o._L.mf_get_pose(self._o._h, self._i, out.ctypes.data))
        return out.reshape(4, 4).T.astype(np.float64)

    def lastCount(self) -> int:
        n = C.c_uint32(0)
        self._o._chk(self._o._L.mf_get_surfel_count(self._o._h, self._i, C.byref(n)))
        return int(n.value)

    def downloadMap(self) -> np.ndarray:
        """Model::downloadMap -> (count, 12) float32: pos+conf | colour,unused,initTime,lastTime | normal+radius."""
        n = self.lastCount()
        out = np.zeros((max(n, 1), 12), np.float32)
        cnt = C.c_uint32(0)
        self._o._chk(self._o._L.mf_download_map(self._o._h, self._i, out.ctypes.data, n, C.byref(cnt)))
        return out[:n]

    def getICPStats(self):
        e, c = C.c_float(0), C.c_float(0)
        self._o._chk(self._o._L.mf_get_icp_stats(self._o._h, self._i, C.byref(e), C.byref(c)))
        return e.value, c.value

    # -- the per-model operations of Core/Model/Model.h:126-162,233-268, on the frame staged with MaskFusion.stageFrame ----
    def initialise(self):
        self._o._chk(self._o._L.mf_model_initialise(self._o._h, self._i))

    def overridePose(self, pose):
        p = np.ascontiguousarray(np.asarray(pose, np.float32).T.reshape(16))
        self._o._chk(self._o._L.mf_model_override_pose(self._o._h, self._i, p.ctypes.data))

    def computeFusionWeight(self, weightMultiplier: float = 1.0) -> float:
        out = C.c_float(0)
        self._o._chk(self._o._L.mf_model_fusion_weight(self._o._h, self._i, weightMultiplier, C.byref(out)))
        return out.value

    def performTracking(self, frameToFrameRGB=False, rgbOnly=False, icpWeight=None, pyramid=True, fastOdom=False, so3=True,
                        maxDepthProcessed=20.0, logTimestamp=0, tryFillIn=False):
        # icpWeight=None: the weight the context was created with (the images a photometric term needs only exist if the context has one:
        # mf_model_perform_tracking returns MF_ESTATE otherwise instead of tracking against images that were never computed)
        if icpWeight is None:
            icpWeight = self._o.getParam("icpWeight")
        self._o._chk(self._o._L.mf_model_perform_tracking(self._o._h, self._i, int(frameToFrameRGB), int(rgbOnly), icpWeight,
                                                          int(pyramid), int(fastOdom), int(so3), maxDepthProcessed, logTimestamp,
                                                          int(tryFillIn)))

    def predictIndices(self, time: int, maxDepth: float, timeDelta: int):
        self._o._chk(self._o._L.mf_model_predict_indices(self._o._h, self._i, time, maxDepth, timeDelta))

    def fuse(self, time: int, depthCutoff: float, weightMultiplier: float = 1.0):
        self._o._chk(self._o._L.mf_model_fuse(self._o._h, self._i, time, depthCutoff, weightMultiplier))

    def clean(self, time: int, timeDelta: int, depthCutoff: float):
        self._o._chk(self._o._L.mf_model_clean(self._o._h, self._i, time, timeDelta, depthCutoff))

    def combinedPredict(self, maxDepth: float, time: int, maxTime: int, timeDelta: int):
        self._o._chk(self._o._L.mf_model_combined_predict(self._o._h, self._i, maxDepth, time, maxTime, timeDelta))

    def uploadMap(self, surfels: np.ndarray):
        s = np.ascontiguousarray(surfels, np.float32).reshape(-1, 12)
        self._o._chk(self._o._L.mf_model_upload_map(self._o._h, self._i, s.ctypes.data, len(s)))

    def makeNonStatic(self):
        self._o._chk(self._o._L.mf_make_nonstatic(self._o._h, self._i))

    def makeStatic(self):
        self._o._chk(self._o._L.mf_make_static(self._o._h, self._i))

    def isNonstatic(self) -> bool:
        return not bool(self.info().is_static)

    def debugRead(self, what: str) -> np.ndarray:
        return self._o.debugRead(what, model=self._i)


class MaskFusion:
    """MaskFusion facade (constructor arguments: Core/MaskFusion.h:47-53; Resolution/Intrinsics are explicit)."""

    def __init__(self, width=640, height=480, fx=528.0, fy=528.0, cx=320.0, cy=240.0, *, timeDelta=200,
                 initConfidenceGlobal=4.0, initConfidenceObject=2.0, depthCut=3.0, icpThresh=10.0, fastOdom=False,
                 so3=True, device=0, numGSurfels=9437184, numOSurfels=1048576, enableMultipleModels=True,
                 outlierCoefficient=0.9, modelSpawnOffset=20, trackAllModels=True, rgbOnly=False):
        self._L = load()
        cfg = Config()
        self._L.mf_default_config(C.byref(cfg), width, height, fx, fy, cx, cy)
        cfg.device = device
        cfg.time_delta = timeDelta
        cfg.conf_global = initConfidenceGlobal
        cfg.conf_object = initConfidenceObject
        cfg.depth_cutoff = depthCut
        cfg.icp_weight = icpThresh
        cfg.fast_odom = int(fastOdom)
        cfg.so3 = int(so3)
        cfg.rgb_only = int(rgbOnly)
        cfg.num_gsurfels = numGSurfels
        cfg.num_osurfels = numOSurfels
        cfg.enable_multiple_models = int(enableMultipleModels)
        cfg.outlier_coefficient = outlierCoefficient
        cfg.model_spawn_offset = modelSpawnOffset
        cfg.track_all_models = int(trackAllModels)
        self.cfg = cfg
        self.width, self.height = width, height
        h = C.c_void_p()
        rc = self._L.mf_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            why = {-1: "invalid configuration: width and height must be positive multiples of 8 (three pyramid levels), surfel capacities > 0",
                   -2: "needs a gfx950 GPU; there is no CPU path", -4: "out of device memory"}.get(rc, "see the HIP runtime's message on stderr")
            raise MFError(f"mf_create failed with code {rc} ({why})")
        self._h = h

    # -- lifetime -----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.mf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            msg = self._L.mf_last_error(self._h)
            raise MFError(f"code {rc}: {msg.decode() if msg else ''}")

    # -- MaskFusion::processFrame ---------------------------------------------------------------
    def processFrame(self, rgb: np.ndarray, depth: np.ndarray, mask: np.ndarray | None = None, timestamp: int = 0,
                     inPose=None, weightMultiplier: float = 1.0, bootstrap: bool = False, classIDs=()) -> bool:
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        assert rgb.shape == (self.height, self.width, 3) and depth.shape == (self.height, self.width)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, np.uint8)
        pose = None
        if inPose is not None:
            pose = np.ascontiguousarray(np.asarray(inPose, np.float32).T.reshape(16))
        cid = np.ascontiguousarray(classIDs, np.int32) if len(classIDs) else None
        self._chk(self._L.mf_process_frame(self._h, rgb.ctypes.data, depth.ctypes.data,
                                           m.ctypes.data if m is not None else None,
                                           cid.ctypes.data if cid is not None else None, len(classIDs), timestamp,
                                           pose.ctypes.data if pose is not None else None, weightMultiplier,
                                           int(bootstrap)))
        return False  # the reference always returns false (MaskFusion.cpp:606)

    def processFrameDevice(self, d_rgb: int, d_depth: int, d_mask: int = 0, timestamp: int = 0,
                           weightMultiplier: float = 1.0):
        """Inputs already resident in HBM (raw device pointers, e.g. torch tensor.data_ptr()); asynchronous."""
        self._chk(self._L.mf_process_frame_dev(self._h, d_rgb, d_depth, d_mask or None, timestamp, weightMultiplier))

    def setMaskClassIDs(self, classIDs):
        """FrameData::classIDs for the frames handed to processFrameDevice (class of mask value v = classIDs[v])"""
        a = np.ascontiguousarray(list(classIDs), np.int32)
        self._chk(self._L.mf_set_mask_class_ids(self._h, a.ctypes.data if len(a) else None, len(a)))

    def modelStateDevice(self, model: int, d_out16: int):
        """Enqueue a copy of {R, t, ICP error, inliers, surfels, alive} of `model` into a 16-float device buffer."""
        self._chk(self._L.mf_model_state_dev(self._h, model, d_out16))

    def sync(self):
        self._chk(self._L.mf_sync(self._h))

    def stageFrame(self, rgb: np.ndarray, depth: np.ndarray, mask: np.ndarray | None = None):
        """upload + filterDepth + Model::generateCUDATextures + intensity pyramid: what processFrame does before it touches a model"""
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None else None
        self._chk(self._L.mf_stage_frame(self._h, rgb.ctypes.data, depth.ctypes.data, m.ctypes.data if m is not None else None))

    def stageFrameDevice(self, d_rgb: int, d_depth: int, d_mask: int = 0):
        """stageFrame for buffers already in device memory (pointers as integers), asynchronous: see mf_stage_frame_dev"""
        self._chk(self._L.mf_stage_frame_dev(self._h, d_rgb, d_depth, d_mask or None))

    def endFrame(self, timestamp: int = 0):
        self._chk(self._L.mf_end_frame(self._h, timestamp))

    # the per-model loops of processFrame over this context's list (mf_track_models / mf_fuse_models / mf_predict_models): one batched
    # Gauss-Newton loop, one launch per surfel pass for all object models.  firstModel = 1: models[0] is a background stand-in.
    def trackModels(self, firstModel: int = 0, trackAllModels: bool = True):
        """the tracking loop, Core/MaskFusion.cpp:247-276"""
        self._chk(self._L.mf_track_models(self._h, firstModel, int(bool(trackAllModels))))

    def fuseModels(self, firstModel: int = 0, weightMultiplier: float = 1.0, spawnedModel: int = -1):
        """:335-374 object parameters and the spawn-frame pass of models[spawnedModel], then the fusion loop :539-565"""
        self._chk(self._L.mf_fuse_models(self._h, firstModel, float(weightMultiplier), spawnedModel))

    def predictModels(self, firstModel: int = 0, timestamp: int = 0):
        """predict() :569 + tick++ / pose log / age++ (the end of a frame sequenced by the caller; replaces endFrame)"""
        self._chk(self._L.mf_predict_models(self._h, firstModel, timestamp))

    def modelIDs(self):
        """ids of the model list in list order (host state: does not wait for the device, unlike Model.getID)"""
        ids = (C.c_int32 * 256)()
        n = C.c_int32(0)
        self._chk(self._L.mf_get_model_ids(self._h, ids, 256, C.byref(n)))
        return [int(ids[i]) for i in range(n.value)]

    def modelsStateDevice(self, d_out16: int, capacity: int):
        """modelStateDevice for the whole list: 16 floats per model into one device buffer"""
        self._chk(self._L.mf_models_state_dev(self._h, d_out16, capacity))

    def setTrackableClassIds(self, ids):
        a = np.ascontiguousarray(list(ids), np.int32)
        self._chk(self._L.mf_set_trackable_class_ids(self._h, a.ctypes.data if len(a) else None, len(a)))

    def predict(self):
        self._chk(self._L.mf_predict(self._h))

    def preallocateModels(self, count: int):
        self._chk(self._L.mf_preallocate_models(self._h, count))

    # -- exports (Core/MaskFusion.h:282-284) --------------------------------------------------------
    def savePly(self, exportDir: str):
        self._chk(self._L.mf_save_ply(self._h, exportDir.encode()))

    def exportPoses(self, exportDir: str):
        self._chk(self._L.mf_export_poses(self._h, exportDir.encode()))

    def getPoseLog(self, model: int = 0):
        """(timestamps int64[n], poses float32[n, 7] = tx ty tz qx qy qz qw), Model::getPoseLog."""
        n = C.c_uint32(0)
        self._chk(self._L.mf_get_pose_log(self._h, model, None, None, 0, C.byref(n)))
        ts = np.zeros(n.value, np.int64)
        p = np.zeros((n.value, 7), np.float32)
        if n.value:
            self._chk(self._L.mf_get_pose_log(self._h, model, ts.ctypes.data, p.ctypes.data, n.value, C.byref(n)))
        return ts, p

    # -- getters ----------------------------------------------------------------------------------
    def getTick(self) -> int:
        t = C.c_int32(0)
        self._chk(self._L.mf_get_tick(self._h, C.byref(t)))
        return t.value

    def setTick(self, tick: int):
        self._chk(self._L.mf_set_tick(self._h, int(tick)))

    def getBackgroundModel(self) -> Model:
        return Model(self, 0)

    def getModels(self):
        n = C.c_int32(0)
        self._chk(self._L.mf_num_models(self._h, C.byref(n)))
        return [Model(self, i) for i in range(n.value)]

    def getCurrPose(self) -> np.ndarray:
        return self.getBackgroundModel().getPose()

    def downloadSegmentation(self) -> np.ndarray:
        """SegmentationResult::fullSegmentation of the last frame (model id per pixel, 255 = ignored)."""
        out = np.zeros((self.height, self.width), np.uint8)
        self._chk(self._L.mf_download_segmentation(self._h, out.ctypes.data))
        return out

    def exportSegmentation(self, path: str):
        """the exportSegmentation branch of processFrame (MaskFusion.cpp:299-303): label image, 255 zeroed, as an 8-bit PNG"""
        self._chk(self._L.mf_export_segmentation_png(self._h, path.encode()))

    def getLastFillIn(self) -> bool:
        u = C.c_int32(0)
        self._chk(self._L.mf_get_last_fillin(self._h, C.byref(u)))
        return bool(u.value)

    # -- setters (MaskFusion.h:132-182) --------------------------------------------------------------
    def setParam(self, key: str, value: float):
        self._chk(self._L.mf_set_param(self._h, key.encode(), float(value)))

    def getParam(self, key: str) -> float:
        v = C.c_double(0)
        self._chk(self._L.mf_get_param(self._h, key.encode(), C.byref(v)))
        return v.value

    def setDepthCutoff(self, v): self.setParam("depthCutoff", v)
    def setIcpWeight(self, v): self.setParam("icpWeight", v)
    def setConfidenceThreshold(self, v): self.setParam("confidenceThreshold", v)
    def setOutlierCoefficient(self, v): self.setParam("outlierCoefficient", v)
    def setFastOdom(self, v): self.setParam("fastOdom", int(v))
    def setSo3(self, v): self.setParam("so3", int(v))
    def setFrameToFrameRGB(self, v): self.setParam("frameToFrameRGB", int(v))   # Core/MaskFusion.cpp:910
    def setRgbOnly(self, v): self.setParam("rgbOnly", int(v))

    def trackStats(self, model: int = 0) -> dict:
        """Statistics of the last tracking step (RGBDOdometry.h:68-75)."""
        out = np.zeros(8, np.float32)
        self._chk(self._L.mf_get_track_stats(self._h, model, out.ctypes.data))
        keys = ("lastICPError", "lastICPCount", "lastRGBError", "lastRGBCount", "lastSO3Error", "lastSO3Count", "so3Iterations",
                "rejected")
        return dict(zip(keys, out.tolist()))
    def gnIllIterations(self, model: int = 0) -> int:
        """Gauss-Newton iterations of the last geometric tracking step whose system was outside the solver's stated domain (finding F4)"""
        n = C.c_int32(0)
        self._chk(self._L.mf_get_gn_condition(self._h, model, C.byref(n)))
        return n.value

    def setPyramid(self, v): self.setParam("pyramid", int(v))
    def setEnableMultipleModels(self, v): self.setParam("enableMultipleModels", int(v))

    def enableTimings(self, on=True):
        self.setParam("timings", 1 if on else 0)

    def timings(self) -> dict:
        t = np.zeros(MF_N_TIMINGS, np.float32)
        self._chk(self._L.mf_get_timings(self._h, t.ctypes.data))
        return dict(zip(TIMING_LABELS, t.tolist()))

    def passTimings(self) -> dict:
        """GPU milliseconds of the surfel passes of the last frame, pass by pass (setParam("passTimings", 1); labels: maskfusion_amd.h MF_PASS_*)"""
        t = np.zeros(MF_N_PASSES, np.float32)
        self._chk(self._L.mf_get_pass_timings(self._h, t.ctypes.data))
        return dict(zip(PASS_LABELS, t.tolist()))

    def stream(self) -> int:
        return int(self._L.mf_get_stream(self._h) or 0)

    def inputStream(self) -> int:
        """hipStream_t on which frames handed to processFrameDevice are first read (see include/maskfusion_amd.h)."""
        return int(self._L.mf_get_input_stream(self._h) or 0)

    # -- differential-test taps ----------------------------------------------------------------------
    def debugRead(self, what: str, model: int = 0, count: int | None = None) -> np.ndarray:
        """count: number of records for the variable-length taps (cand_op, cand_rec, clean_flags, clean_newconf)"""
        W, H = self.width, self.height
        shapes = {"depthF": ((H, W), np.float32), "pred_vertex": ((H, W, 4), np.float32),
                  "pred_normal": ((H, W, 4), np.float32), "pred_image": ((H, W, 4), np.uint8), "pred_time": ((H, W), np.uint16),
                  "index": ((H, W), np.int32), "index_vc": ((H, W, 4), np.float32), "index_nr": ((H, W, 4), np.float32),
                  "index_ct": ((H, W, 4), np.float32), "index_packed": ((W, H, 2, 4), np.float32), "icp_log": ((19, 32), np.float32), "gn_trace": ((20, 64), np.float64),
                  "icp_prof": ((19, 16), np.uint64), "splat_prof": ((((H + 15) // 16) * ((W + 15) // 16), 8), np.uint64),
                  "edge_map": ((H, W), np.float32),
                  "edge_binary": ((H, W), np.uint8), "projected_ids": ((H, W), np.uint8)}
        for i in range(3):      # the frame's intensity pyramid and its derivative / gate images (photometric term, SO(3))
            shapes.update({f"gray{i}": ((H >> i, W >> i), np.uint8), f"dIdx{i}": ((H >> i, W >> i), np.int16), f"dIdy{i}": ((H >> i, W >> i), np.int16),
                           f"rgb_gate{i}": ((H >> i, W >> i), np.uint8)})
        if count is not None:
            shapes.update({"cand_op": ((count,), np.uint8), "cand_rec": ((count, 12), np.float32),
                           "clean_flags": ((count,), np.uint8), "clean_newconf": ((count,), np.float32)})
        for pre in ("vmap_g", "nmap_g", "vmap", "nmap"):
            for i in range(3):
                shapes[f"{pre}{i}"] = ((3, H >> i, W >> i), np.float32)
        shape, dt = shapes[what]
        out = np.zeros(shape, dt)
        self._chk(self._L.mf_debug_read_model(self._h, model, what.encode(), out.ctypes.data, out.nbytes))
        return out
