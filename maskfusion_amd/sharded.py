"""One multi-model scene sharded BY MODEL over several contexts (one per GPU): SURVEY.md 8e, BASELINE.json configs[3] / [4].

MaskFusion's per-model state (surfel map, tracker pyramids, pose) is independent (Core/Model/Model.h:271-323).  What
MaskFusion::processFrame couples between models crosses the contexts here:

  1 frame                    rank 0 -> all     ONE broadcast of rgb | depth packed in one buffer (7 P bytes); every rank filters / builds
                                               its own pyramids
  2 z-merged model-id image  all   -> all      all-reduce(MIN) of the uint64 projection keys (GlobalProjection.cpp:43-114)
  3 per-model state          all   -> rank 0   gather {pose, ICP error, inliers, surfels, alive} (the 0.2 m jump rule, logging)
  4 labels + pose + control  rank 0 -> all     ONE broadcast of the label image | the background's state record | {has_new, new id, new class,
                                               owner rank, global list} packed in one buffer (P + 64 + 288 bytes; MaskFusion.cpp:289-297)
  (static objects only: the background's NEW pose, 64 B, rank 0 -> all before the projection -- they follow it, Model.h:263)

FOUR collectives per frame with tracked objects (tests/test_sharded_gloo.py counts them); round 3 issued seven (rgb and depth, labels, pose
and control as separate broadcasts) and rank 0 drained its stream after the label stage to read the pose it was about to publish -- the
record now goes from the library's device state into the packed buffer on the stream (mf_model_state_dev).

Rank 0 owns the background model, the label stage and the model-id allocator; an object model lives on the rank chosen when it
is spawned (the rank with the fewest models; rank 0 only when it is alone).  The frame logic is written as three phases per rank
(`phase_track`, `phase_segment` on rank 0, `phase_fuse`) so that the same code runs
  * SPMD, one process per GPU, with torch.distributed collectives between the phases (`ShardedMaskFusion.process_frame`), and
  * in ONE process over several contexts on one GPU, the "collectives" being plain tensor ops (`LocalGroup`): the parity test
    compares that with the single-context multi-model run bit for bit.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from .api import MaskFusion
from . import dist as mfd

STATE_W = mfd.STATS_WIDTH        # R(9) t(3) icpError icpCount surfels alive
MAX_LOCAL = 32                   # object models per rank in the gathered state block
CTL_WORDS = 8 + 64               # control record: has_new, new_id, new_class, owner, 0, 0, 0, len(order), order[64]


def pose16_from_state(rec: np.ndarray) -> np.ndarray:
    """column-major 4x4 (what mf_model_override_pose takes) from a state record's R (row-major 9) and t (3)"""
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.asarray(rec[:9], np.float32).reshape(3, 3)
    T[:3, 3] = rec[9:12]
    return np.ascontiguousarray(T.T.reshape(16))


@dataclass
class GlobalModel:
    """one entry of MaskFusion::models as rank 0 sees it"""
    id: int
    class_id: int
    rank: int


@dataclass
class Control:
    """what rank 0 tells everybody after the label stage (16 int32 on the wire)"""
    has_new: int = 0
    new_id: int = 0
    new_class: int = -1
    owner: int = 0
    order: List[int] = field(default_factory=list)     # global list (ids in order) after drops, before the spawn


class Shard:
    """One rank: a context plus the bookkeeping of which global models it holds."""

    def __init__(self, rank: int, world: int, mf: MaskFusion, device: torch.device):
        self.rank, self.world, self.mf, self.device = rank, world, mf, device
        P = mf.width * mf.height
        self.keys = torch.empty(P, dtype=torch.int64, device=device)
        # what rank 0 publishes after the label stage, in ONE buffer: label image (P) | background state record (16 floats: R row-major,
        # t, ICP error, inliers, surfels, alive) | control record (CTL_WORDS int32).  labels / bg_state / ctl are views into it.
        self.post = torch.zeros(P + 64 + 4 * CTL_WORDS, dtype=torch.uint8, device=device)
        self.labels = self.post[:P]
        self.bg_state = self.post[P:P + 64].view(torch.float32)
        self.ctl = self.post[P + 64:].view(torch.int32)
        self.bg_pose = torch.zeros(16, dtype=torch.float32, device=device)   # static objects: the early pose broadcast
        self.state = torch.zeros((MAX_LOCAL + 1, STATE_W), dtype=torch.float32, device=device)
        # rank 0 only
        self.table: List[GlobalModel] = [GlobalModel(0, -1, 0)]
        self.next_id = 1
        self.spawn_offset = 0
        self.tick = 1

    # ---------------------------------------------------------------------------------------------
    def local_ids(self) -> List[int]:
        return self.mf.modelIDs()      # host state: no device wait (Model.getID() would drain the stream once per model)

    def owns_background(self) -> bool:
        return self.rank == 0

    # ---- phase 1: everything up to the global projection ---------------------------------------------
    def phase_track(self, rgb, depth, order_of_id: dict, cfg: dict, first: bool, bg_pose16: Optional[np.ndarray] = None):
        """stage the frame, track the local models, scatter them into this rank's key image.  rgb / depth: host arrays (uploaded,
        synchronous) or torch tensors on this shard's device (staged in place, asynchronous: mf_stage_frame_dev -- the caller orders their
        producer on the library's stream).  bg_pose16: the background's NEW pose, needed here only by ranks that hold static objects
        (they follow it before they are projected, MaskFusion.cpp:274,289)"""
        mf = self.mf
        if isinstance(rgb, torch.Tensor):
            mf.stageFrameDevice(rgb.data_ptr(), depth.data_ptr())
        else:
            mf.stageFrame(rgb, depth)
        if first:
            if self.owns_background():
                mf.getBackgroundModel().initialise()
            return
        if not self.owns_background() and bg_pose16 is not None:
            mf._chk(mf._L.mf_model_override_pose(mf._h, 0, np.ascontiguousarray(bg_pose16, np.float32).ctypes.data))
        # the tracking loop (MaskFusion.cpp:247-276) over the local list in ONE call: the Gauss-Newton loops of all tracked models run as
        # one batch, static objects follow the background (updateStaticPose, :274).  rgbOnly / icpWeight / fastOdom / so3 /
        # maxDepthProcessed are the context's configuration (cfg carries the same values for the calls that still take them).
        first_local = 0 if self.owns_background() else 1
        mf.trackModels(first_local, cfg["trackAllModels"])
        ids = self.local_ids()
        orders = np.array([order_of_id.get(mid, -1) if (i > 0 or self.owns_background()) else -1 for i, mid in enumerate(ids)], np.int32)
        mf._chk(mf._L.mf_export_projection_keys_dev(mf._h, orders.ctypes.data, len(orders), self.keys.data_ptr()))
        mf.modelsStateDevice(self.state.data_ptr(), MAX_LOCAL + 1)

    # ---- phase 2 (rank 0): label stage on the merged projection ---------------------------------------
    def phase_segment(self, mask: Optional[np.ndarray], class_ids: Sequence[int], merged_keys: torch.Tensor, alive_of_id: dict, cfg: dict,
                      weight_multiplier: float = 1.0) -> Control:
        mf = self.mf
        import ctypes as C
        # inactivateModel for objects the jump rule dropped on their ranks (MaskFusion.cpp:268-272)
        self.table = [g for g in self.table if g.id == 0 or alive_of_id.get(g.id, 1)]
        mf._chk(mf._L.mf_import_projection_keys_dev(mf._h, merged_keys.data_ptr()))
        if self.spawn_offset < cfg["modelSpawnOffset"]:
            self.spawn_offset += 1                          # :294
        ids = np.array([g.id for g in self.table], np.int32)
        cls = np.array([g.class_id for g in self.table], np.int32)
        has_new, new_cls = C.c_int32(0), C.c_int32(-1)
        m = np.ascontiguousarray(mask, np.uint8) if mask is not None and len(class_ids) else None
        cid = np.ascontiguousarray(class_ids, np.int32) if m is not None else None
        # label stage enqueued; the background's fusion (it reads the label image on the stream, and the background is never spawned or
        # dropped) keeps the GPU busy while the host waits for the new-model decision -- as mf_process_frame does
        mf._chk(mf._L.mf_perform_segmentation_begin(mf._h, m.ctypes.data if m is not None else None, cid.ctypes.data if cid is not None else None,
                                                    len(class_ids) if m is not None else 0, ids.ctypes.data, cls.ctypes.data, len(ids), self.next_id,
                                                    int(self.spawn_offset >= cfg["modelSpawnOffset"])))
        if not cfg["rgbOnly"]:
            mf._chk(mf._L.mf_fuse_background(mf._h, float(weight_multiplier)))
        mf._chk(mf._L.mf_perform_segmentation_end(mf._h, C.byref(has_new), C.byref(new_cls)))
        ctl = Control(order=[g.id for g in self.table])
        if has_new.value and len(self.table) < cfg["maxModels"]:
            ctl.has_new, ctl.new_id, ctl.new_class = 1, self.next_id, new_cls.value
            load = [sum(1 for g in self.table if g.rank == r) for r in range(self.world)]
            ctl.owner = 0 if self.world == 1 else 1 + int(np.argmin(load[1:]))
            self.table.append(GlobalModel(ctl.new_id, ctl.new_class, ctl.owner))
            used = {g.id for g in self.table}
            while True:                                     # getNextModelID (MaskFusion.cpp:715-731)
                self.next_id = (self.next_id + 1) & 255
                if self.next_id not in used:
                    break
            self.spawn_offset = 0
        mf._chk(mf._L.mf_export_segmentation_dev(mf._h, self.labels.data_ptr()))
        return ctl

    # ---- phase 3: labels + control in, fuse / clean / predict the local models -------------------------
    def phase_fuse(self, ctl: Control, bg_pose16: np.ndarray, cfg: dict, weight_multiplier: float, timestamp: int, first: bool):
        mf = self.mf
        t = self.tick
        if first:
            if self.owns_background():
                mf.getBackgroundModel().combinedPredict(cfg["maxDepthProcessed"], t, t, cfg["timeDelta"])
            mf.endFrame(timestamp)
            self.tick += 1
            return
        if not self.owns_background():
            mf._chk(mf._L.mf_import_segmentation_dev(mf._h, self.labels.data_ptr()))
            mf._chk(mf._L.mf_model_override_pose(mf._h, 0, np.ascontiguousarray(bg_pose16, np.float32).ctypes.data))
        # drops decided by the jump rule: the list rank 0 kept
        keep = set(ctl.order)
        ids = self.local_ids()
        for i in reversed(range(1, len(ids))):
            if ids[i] not in keep:
                mf._chk(mf._L.mf_drop_model(mf._h, i))
                del ids[i]
        spawned = -1
        if ctl.has_new and ctl.owner == self.rank:
            mf._chk(mf._L.mf_spawn_object_model(mf._h, ctl.new_id, ctl.new_class))
            spawned = len(ids)
        # :335-339 / :369-374 object parameters, :342-353 the spawn-frame pass of the new model (fuse with weight 100 under
        # initConfidenceObject, clean without a second index pass), then the fusion loop :539-565 and predict() :569 + the frame's tail --
        # the object models go through one launch per surfel pass
        first_local = 0 if self.owns_background() else 1
        mf.fuseModels(first_local, weight_multiplier, spawned)
        mf.predictModels(first_local, timestamp)
        self.tick += 1


def default_cfg(**kw) -> dict:
    cfg = dict(trackAllModels=True, rgbOnly=False, icpWeight=100.0, fastOdom=False, so3=False, maxDepthProcessed=20.0, depthCutoff=3.0,
               timeDelta=200, modelSpawnOffset=20, maxModels=32)
    cfg.update(kw)
    return cfg


def _alive_from_states(states: Sequence[torch.Tensor], ids_per_rank: Sequence[Sequence[int]]) -> dict:
    """{model id: alive} from the gathered per-rank state blocks"""
    out = {}
    for st, ids in zip(states, ids_per_rank):
        a = st.cpu().numpy()
        for i, mid in enumerate(ids):
            out[mid] = int(a[i, 15] != 0)
    return out


class LocalGroup:
    """All shards in one process (one GPU): the parity-test form.  Collectives are tensor ops on the same device."""

    def __init__(self, shards: Sequence[Shard], cfg: dict):
        self.shards, self.cfg = list(shards), cfg
        self.frame = 0

    def process_frame(self, rgb, depth, mask=None, class_ids=(), weight_multiplier=1.0, timestamp=0):
        first = self.frame == 0
        s0 = self.shards[0]
        order_of_id = {g.id: i for i, g in enumerate(s0.table)}
        s0.phase_track(rgb, depth, order_of_id, self.cfg, first)
        early_pose = None
        if not first and not self.cfg["trackAllModels"]:
            s0.mf.sync()
            early_pose = np.ascontiguousarray(s0.mf.getCurrPose().astype(np.float32).T.reshape(16))
        for s in self.shards[1:]:
            s.phase_track(rgb, depth, order_of_id, self.cfg, first, early_pose)
        ctl = Control(order=[g.id for g in s0.table])
        bg_pose = np.eye(4, dtype=np.float32).T.reshape(16)
        if not first:
            for s in self.shards:
                s.mf.sync()
            merged = mfd.keys_to_wire(self.shards[0].keys).clone()
            for s in self.shards[1:]:
                merged = torch.minimum(merged, mfd.keys_to_wire(s.keys))
            merged = _wire_to_keys(merged)
            _device_sync(self.shards[0].device)
            alive = _alive_from_states([s.state for s in self.shards], [s.local_ids() for s in self.shards])
            for s in self.shards[1:]:
                alive.pop(0, None)       # the background stand-ins of the object ranks say nothing about the background
            alive[0] = 1
            ctl = s0.phase_segment(mask, class_ids, merged, alive, self.cfg, weight_multiplier)
            s0.mf.sync()
            bg_pose = np.ascontiguousarray(s0.mf.getCurrPose().astype(np.float32).T.reshape(16))
            for s in self.shards[1:]:
                s.labels.copy_(s0.labels)
            _device_sync(s0.device)
        for s in self.shards:
            s.phase_fuse(ctl, bg_pose, self.cfg, weight_multiplier, timestamp, first)
        self.frame += 1
        return ctl


def _device_sync(device: torch.device):
    """the collectives above run on torch's stream, the library on its own: drain the device between them (no-op for the CPU tensors
    of the gloo tests, which drive this file with a stand-in context)"""
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _wire_to_keys(wire: torch.Tensor) -> torch.Tensor:
    """int64 wire values back to the library's uint64 keys (INT64_MAX -> all ones)"""
    return torch.where(wire == mfd.INT64_MAX, torch.full_like(wire, -1), wire)


class ShardedMaskFusion:
    """SPMD form: one process per GPU, torch.distributed (RCCL) between the phases.  Every rank calls process_frame with the same
    arguments; only rank 0's rgb / depth / mask are used (the others may pass None)."""

    def __init__(self, mf: MaskFusion, device: torch.device, cfg: dict):
        import torch.distributed as dist
        self.dist = dist
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.shard = Shard(self.rank, self.world, mf, device)
        self.cfg = cfg
        self.device = device
        H, W = mf.height, mf.width
        # a ring of 3 frame buffers: the library may still read frame k-1 (fill-in intensity at predict time) while frame k is published.
        # rgb (3 P bytes) and depth (4 P bytes) of a frame share ONE buffer, so that the frame is ONE broadcast (north star); 3 P is a multiple
        # of 4 (width and height are multiples of 8), the float view is aligned.
        P = H * W
        self.frames = [torch.empty(7 * P, dtype=torch.uint8, device=device) for _ in range(3)]
        self.rgbs = [f[:3 * P].view(H, W, 3) for f in self.frames]
        self.depths = [f[3 * P:].view(torch.float32).view(H, W) for f in self.frames]
        self.collectives = 0           # issued by this rank so far (the gloo test asserts four per frame)
        self.frame = 0
        self._order = [0]              # host copy of the global model list (ids in order) as of the end of the last frame
        # On a GPU every tensor op and collective of a frame is enqueued on the LIBRARY's stream (torch.cuda.ExternalStream): the frame
        # broadcast, the staging of the broadcast buffer (mf_stage_frame_dev: no host copy of the frame on any rank), tracking, the key
        # all-reduce, the label stage and fusion are ordered by that one stream; the host waits only where it has to read something
        # (the gathered alive flags on rank 0, the control record on every rank).
        self._ext = torch.cuda.ExternalStream(mf.stream(), device=device) if device.type == "cuda" else None

    def process_frame(self, rgb=None, depth=None, mask=None, class_ids=(), weight_multiplier=1.0, timestamp=0):
        if self._ext is None:
            return self._process_frame(rgb, depth, mask, class_ids, weight_multiplier, timestamp)
        with torch.cuda.stream(self._ext):
            return self._process_frame(rgb, depth, mask, class_ids, weight_multiplier, timestamp)

    def _process_frame(self, rgb, depth, mask, class_ids, weight_multiplier, timestamp):
        dist, s, first = self.dist, self.shard, self.frame == 0
        on_gpu = self._ext is not None
        d_rgb, d_depth = self.rgbs[self.frame % 3], self.depths[self.frame % 3]
        if self.rank == 0:
            d_rgb.copy_(torch.from_numpy(np.ascontiguousarray(rgb, np.uint8)))
            d_depth.copy_(torch.from_numpy(np.ascontiguousarray(depth, np.float32)))
        if self.world > 1:
            dist.broadcast(self.frames[self.frame % 3], 0)      # collective 1: rgb | depth
            self.collectives += 1
        # GPU: the broadcast buffers are staged in place; CPU tensors (gloo tests over a stand-in context): as host arrays
        rgb_h, depth_h = (d_rgb, d_depth) if on_gpu else (d_rgb.numpy(), d_depth.numpy())
        # the global list as of the end of the previous frame travelled in the control record: every rank kept its host copy (reading the
        # device record here would drain the stream before the frame has even started)
        order = list(self._order) if self.frame else [0]
        order_of_id = {mid: i for i, mid in enumerate(order)}
        if first or self.cfg["trackAllModels"] or self.world == 1:
            s.phase_track(rgb_h, depth_h, order_of_id, self.cfg, first)
        else:
            # static objects are projected with the background's NEW pose: rank 0 tracks first and publishes it (64 B)
            if self.rank == 0:
                s.phase_track(rgb_h, depth_h, order_of_id, self.cfg, first)
                s.mf.sync()
                s.bg_pose.copy_(torch.from_numpy(np.ascontiguousarray(s.mf.getCurrPose().astype(np.float32).T.reshape(16))))
            dist.broadcast(s.bg_pose, 0)
            self.collectives += 1
            if self.rank != 0:
                s.phase_track(rgb_h, depth_h, order_of_id, self.cfg, first, s.bg_pose.cpu().numpy())
        ctl = Control(order=order)
        bg_pose = np.eye(4, dtype=np.float32).T.reshape(16)
        if not first:
            if not on_gpu:
                s.mf.sync()
            wire = mfd.merge_projection_keys(mfd.keys_to_wire(s.keys).clone())      # collective 2
            self.collectives += 1 if self.world > 1 else 0
            ids_block = torch.full((MAX_LOCAL + 1,), -1, dtype=torch.float32, device=self.device)
            ids = s.local_ids()
            ids_block[:len(ids)] = torch.tensor(ids, dtype=torch.float32)
            send = torch.cat([s.state.reshape(-1), ids_block])
            if self.world > 1:
                recv = [torch.empty_like(send) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(send, recv, dst=0)                                       # collective 3
                self.collectives += 1
            else:
                recv = [send]
            if not on_gpu:
                _device_sync(self.device)
            if self.rank == 0:
                alive = {}
                for r, blk in enumerate(recv):
                    b = blk.cpu().numpy()
                    st = b[:(MAX_LOCAL + 1) * STATE_W].reshape(MAX_LOCAL + 1, STATE_W)
                    for i, mid in enumerate(b[(MAX_LOCAL + 1) * STATE_W:]):
                        if mid >= 0 and not (r > 0 and i == 0):
                            alive[int(mid)] = int(st[i, 15] != 0)
                alive[0] = 1
                keys = _wire_to_keys(wire)
                if not on_gpu:
                    _device_sync(self.device)
                ctl = s.phase_segment(mask, class_ids, keys, alive, self.cfg, weight_multiplier)
                # the background's state record goes from the library's device state into the packed buffer ON THE STREAM (no drained stream
                # on rank 0: round 3 synchronised here to read the pose it was about to publish); rank 0 itself never needs it on the host
                s.mf.modelStateDevice(0, s.bg_state.data_ptr())
                rec = [ctl.has_new, ctl.new_id, ctl.new_class, ctl.owner, 0, 0, 0, len(ctl.order)] + ctl.order
                rec_t = torch.zeros(CTL_WORDS, dtype=torch.int32)
                rec_t[:len(rec)] = torch.tensor(rec, dtype=torch.int32)
                s.ctl.copy_(rec_t, non_blocking=True)
            if self.world > 1:
                dist.broadcast(s.post, 0)                                            # collective 4: labels | background state | control
                self.collectives += 1
            if not on_gpu:
                _device_sync(self.device)
            if self.rank != 0:
                # (the one host wait the object ranks need: what to drop / spawn, and the background's pose; rank 0 wrote both itself)
                tail = s.post[s.post.numel() - 64 - 4 * CTL_WORDS:].cpu().numpy()
                c = tail[64:].view(np.int32)
                ctl = Control(int(c[0]), int(c[1]), int(c[2]), int(c[3]), [int(x) for x in c[8:8 + int(c[7])]])
                bg_pose = pose16_from_state(tail[:64].view(np.float32))
        s.phase_fuse(ctl, bg_pose, self.cfg, weight_multiplier, timestamp, first)
        # the list that the NEXT frame's projection orders by includes the new model
        self._order = list(ctl.order) + ([ctl.new_id] if (not first and ctl.has_new) else [])
        self.frame += 1
        return ctl
