"""A CPU stand-in for maskfusion_amd.api.MaskFusion with just the surface maskfusion_amd/sharded.py drives (the model-level calls of the
C ABI), so that the SPMD orchestration of a model-sharded scene -- broadcasts, the key all-reduce, the state gather, the control
record, who spawns / drops what -- runs under `gloo` on a machine without a GPU.  The "scene" is a toy with the same couplings:

  * a model projects where the previous label image carried its id (the background everywhere), objects nearer than the background;
    keys use the library's layout float_bits(z) << 32 | order << 8 | id (include/maskfusion_amd.h);
  * the label stage explains a mask region by the model that projects into most of it, otherwise (if allowed) reports a new label;
  * tracking moves a model by 1 mm * (id + 1) per frame; a model of class 99 "jumps" once it is three frames old (alive = 0,
    the 0.2 m rule of MaskFusion.cpp:268-272);
  * every model-level call is appended to `log`, which is what the tests compare between the SPMD and the in-process form.
"""
import ctypes as C

import numpy as np


def _arr(ptr, n, ctype, dtype):
    return np.frombuffer((ctype * n).from_address(int(ptr)), dtype=dtype)


class _Info:
    def __init__(self, m):
        self.id, self.class_id, self.is_static, self.surfels, self.age = m.id, m.cls, int(m.static), m.surfels, m.age


class _ModelState:
    def __init__(self, mid, cls):
        self.id, self.cls, self.static, self.alive, self.surfels, self.age = mid, cls, True, 1, 0, 0
        self.pose = np.eye(4, dtype=np.float32)


class _ModelView:
    def __init__(self, ctx, i):
        self.c, self.i = ctx, i

    def _m(self):
        return self.c.models[self.i]

    def info(self):
        return _Info(self._m())

    def getID(self):
        return self._m().id

    def initialise(self):
        self.c.log.append(("initialise", self._m().id))
        self._m().surfels = self.c.P

    def performTracking(self, *a):
        m = self._m()
        self.c.log.append(("track", m.id, bool(a[8]) if len(a) > 8 else False))
        m.pose[0, 3] += 0.001 * (m.id + 1)
        if m.cls == 99 and m.age >= 3:
            m.alive = 0

    def predictIndices(self, t, max_depth, time_delta):
        self.c.log.append(("predictIndices", self._m().id, t, float(max_depth), time_delta))

    def fuse(self, t, depth_cutoff, weight):
        m = self._m()
        m.surfels += int((self.c.labels == m.id).sum())
        self.c.log.append(("fuse", m.id, t, float(depth_cutoff), float(weight), m.surfels))

    def clean(self, t, time_delta, max_depth):
        self.c.log.append(("clean", self._m().id, t, time_delta, float(max_depth)))

    def combinedPredict(self, max_depth, t, max_time, time_delta):
        self.c.log.append(("combinedPredict", self._m().id, t, max_time))


class _Lib:
    """the mf_* entry points sharded.py calls through mf._L (same argument lists; pointers arrive as integers)"""

    def __init__(self, ctx):
        self.c = ctx

    def mf_model_override_pose(self, h, model, ptr):
        self.c.models[model].pose = _arr(ptr, 16, C.c_float, np.float32).reshape(4, 4).T.copy()
        return 0

    def mf_model_update_static_pose(self, h, i):
        self.c.log.append(("static_pose", self.c.models[i].id))
        self.c.models[i].pose = self.c.models[0].pose.copy()
        return 0

    def mf_export_projection_keys_dev(self, h, orders_ptr, n, keys_ptr):
        c = self.c
        orders = _arr(orders_ptr, n, C.c_int32, np.int32)
        assert n == len(c.models)
        keys = np.full(c.P, np.uint64(0xFFFFFFFFFFFFFFFF))
        for m, order in zip(c.models, orders):
            if order < 0:
                continue
            z = np.float32(3.0 if m.id == 0 else 1.0 + 0.01 * m.id)
            where = np.ones(c.P, bool) if m.id == 0 else (c.labels == m.id)
            key = (np.uint64(z.view(np.uint32)) << np.uint64(32)) | np.uint64((int(order) << 8) | m.id)
            keys[where] = np.minimum(keys[where], key)
        _arr(keys_ptr, c.P, C.c_uint64, np.uint64)[:] = keys
        c.log.append(("project", tuple(int(o) for o in orders)))
        return 0

    def mf_import_projection_keys_dev(self, h, keys_ptr):
        k = _arr(keys_ptr, self.c.P, C.c_uint64, np.uint64)
        self.c.proj = np.where(k == np.uint64(0xFFFFFFFFFFFFFFFF), 0, k & np.uint64(0xFF)).astype(np.uint8)
        return 0

    def mf_perform_segmentation(self, h, mask_ptr, cid_ptr, n_masks, ids_ptr, cls_ptr, n_models, next_id, allow_new, has_new, new_cls):
        c = self.c
        labels = c.proj.copy()
        hn, nc = 0, -1
        if n_masks > 0:
            mask = _arr(mask_ptr, c.P, C.c_uint8, np.uint8)
            cids = _arr(cid_ptr, n_masks, C.c_int32, np.int32)
            known = set(int(x) for x in _arr(ids_ptr, n_models, C.c_int32, np.int32))
            for v in range(1, n_masks):
                region = mask == v
                if not region.any():
                    continue
                votes = np.bincount(c.proj[region], minlength=256)
                votes[0] = 0
                best = int(votes.argmax())
                if votes[best] * 2 > region.sum() and best in known:
                    labels[region] = best
                elif allow_new and not hn:
                    hn, nc = 1, int(cids[v])
                    labels[region] = next_id
        c.labels = labels
        has_new._obj.value, new_cls._obj.value = hn, nc
        c.log.append(("segment", n_models, int(next_id), int(allow_new), hn, nc))
        return 0

    def mf_perform_segmentation_begin(self, h, mask_ptr, cid_ptr, n_masks, ids_ptr, cls_ptr, n_models, next_id, allow_new):
        hn, nc = C.c_int32(0), C.c_int32(-1)
        self.mf_perform_segmentation(h, mask_ptr, cid_ptr, n_masks, ids_ptr, cls_ptr, n_models, next_id, allow_new, C.byref(hn), C.byref(nc))
        self._pending = (hn.value, nc.value)
        return 0

    def mf_perform_segmentation_end(self, h, has_new, new_cls):
        has_new._obj.value, new_cls._obj.value = self._pending
        return 0

    def mf_fuse_background(self, h, weight):
        c = self.c
        v = _ModelView(c, 0)
        v.predictIndices(c.tick, c.max_depth_processed, c.time_delta)
        v.fuse(c.tick, c.depth_cutoff, weight)
        v.predictIndices(c.tick, c.max_depth_processed, c.time_delta)
        v.clean(c.tick, c.time_delta, c.max_depth_processed)
        c.bg_fused_tick = c.tick
        return 0

    def mf_export_segmentation_dev(self, h, ptr):
        _arr(ptr, self.c.P, C.c_uint8, np.uint8)[:] = self.c.labels
        return 0

    def mf_import_segmentation_dev(self, h, ptr):
        self.c.labels = _arr(ptr, self.c.P, C.c_uint8, np.uint8).copy()
        return 0

    def mf_drop_model(self, h, i):
        self.c.log.append(("drop", self.c.models[i].id))
        del self.c.models[i]
        return 0

    def mf_spawn_object_model(self, h, mid, cls):
        self.c.log.append(("spawn", mid, cls))
        m = _ModelState(mid, cls)
        m.pose = self.c.models[0].pose.copy()
        self.c.models.append(m)
        return 0

    def mf_update_object_params(self, h):
        return 0


class FakeMaskFusion:
    def __init__(self, W, H):
        self.width, self.height, self.P = W, H, W * H
        self.models = [_ModelState(0, -1)]
        self.labels = np.zeros(self.P, np.uint8)
        self.proj = np.zeros(self.P, np.uint8)
        self.log = []
        self.frames = 0
        self.tick = 1
        self.bg_fused_tick = 0
        self.max_depth_processed, self.depth_cutoff, self.time_delta = 20.0, 3.0, 200    # the context's configuration (sharded.default_cfg)
        self._L = _Lib(self)
        self._h = 0

    def _chk(self, rc):
        assert rc == 0

    def stageFrame(self, rgb, depth, mask=None):
        assert rgb.shape == (self.height, self.width, 3) and depth.shape == (self.height, self.width)
        self.log.append(("stage", int(rgb[0, 0, 0]), float(depth[0, 0])))
        self.frames += 1

    def stageFrameDevice(self, d_rgb, d_depth, d_mask=0):
        """"device" pointers are host pointers here (CPU tensors of the gloo tests)"""
        rgb = _arr(d_rgb, self.P * 3, C.c_uint8, np.uint8).reshape(self.height, self.width, 3)
        depth = _arr(d_depth, self.P, C.c_float, np.float32).reshape(self.height, self.width)
        self.stageFrame(rgb, depth)

    def getModels(self):
        return [_ModelView(self, i) for i in range(len(self.models))]

    def getBackgroundModel(self):
        return _ModelView(self, 0)

    def modelStateDevice(self, i, ptr):
        m = self.models[i]
        out = _arr(ptr, 16, C.c_float, np.float32)
        out[:9] = m.pose[:3, :3].reshape(9)
        out[9:12] = m.pose[:3, 3]
        out[12:] = (0.0, 100.0, m.surfels, m.alive)

    def getCurrPose(self):
        return self.models[0].pose.copy()

    def sync(self):
        pass

    def endFrame(self, timestamp=0):
        for m in self.models:
            m.age += 1
        self.tick += 1
        self.log.append(("end", int(timestamp), tuple(m.id for m in self.models)))

    # ---- the loop-level calls (mf_track_models / mf_fuse_models / mf_predict_models): the same per-model steps, logged one by one ----
    def modelIDs(self):
        return [m.id for m in self.models]

    def modelsStateDevice(self, ptr, capacity):
        assert capacity >= len(self.models)
        for i in range(len(self.models)):
            self.modelStateDevice(i, int(ptr) + 64 * i)

    def trackModels(self, firstModel=0, trackAllModels=True):
        for i in range(firstModel, len(self.models)):
            if i == 0 or (not self.models[i].static) or trackAllModels:
                _ModelView(self, i).performTracking(False, False, 100.0, True, False, False, self.max_depth_processed, 0, i == 0)
            else:
                self._L.mf_model_update_static_pose(self._h, i)

    def fuseModels(self, firstModel=0, weightMultiplier=1.0, spawnedModel=-1):
        t = self.tick
        if spawnedModel > 0:
            nm = _ModelView(self, spawnedModel)
            nm.predictIndices(t, self.max_depth_processed, self.time_delta)
            nm.fuse(t, self.max_depth_processed, 100.0)
            nm.clean(t, self.time_delta, self.max_depth_processed)
        for i in range(firstModel, len(self.models)):
            if i == 0 and self.bg_fused_tick == t:
                continue                      # mf_fuse_background has run for this frame
            m = _ModelView(self, i)
            m.predictIndices(t, self.max_depth_processed, self.time_delta)
            m.fuse(t, self.depth_cutoff, weightMultiplier)
            m.predictIndices(t, self.max_depth_processed, self.time_delta)
            m.clean(t, self.time_delta, self.max_depth_processed)

    def predictModels(self, firstModel=0, timestamp=0):
        t = self.tick
        for i in range(firstModel, len(self.models)):
            _ModelView(self, i).combinedPredict(self.max_depth_processed, t, t, self.time_delta)
        self.endFrame(timestamp)
