"""Deterministic synthetic RGB-D(+instance mask) streams for tests and bench.py (S1/S2/S3).

The reference repository ships no datasets and this environment has no network, so every workload is ray-cast
here: an axis-aligned room (5 planes) furnished with boxes, optional rigid moving boxes (instance ids 1..N),
a smooth 6-DoF Lissajous camera (<= 2 cm / 1 deg per frame, identity at frame 0), textured with smooth
view-consistent procedural colour.  Depth is metric float32 along the camera z axis, 0 = invalid, like
`FrameData::depth` (reference Core/FrameData.h:25-48); rgb is HxWx3 uint8; mask is HxW uint8 (0 = background).

Pure numpy; no dependency on the HIP library or on oracle/.
"""
from __future__ import annotations

import dataclasses
import numpy as np


def rot_xyz(ax: float, ay: float, az: float) -> np.ndarray:
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return rz @ ry @ rx


def make_pose(R: np.ndarray, t) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def camera_pose(frame: int, speed: float = 1.0) -> np.ndarray:
    """Camera-to-world pose of frame `frame` (identity at frame 0)."""
    s = frame / 30.0 * speed
    t = np.array([0.15 * np.sin(0.8 * s), 0.08 * np.sin(1.1 * s), 0.10 * np.sin(0.5 * s)])
    R = rot_xyz(0.05 * np.sin(0.7 * s), 0.08 * np.sin(0.9 * s), 0.03 * np.sin(1.3 * s))
    return make_pose(R, t)


@dataclasses.dataclass
class Box:
    center: np.ndarray            # object-frame origin in world at frame 0
    half: np.ndarray              # half extents
    color: np.ndarray             # base colour (3,)
    instance: int = 0             # mask id (0 = static furniture)
    vel: np.ndarray | None = None  # per-frame translation amplitude (moving objects)
    freq: float = 0.0
    spin: float = 0.0

    def pose(self, frame: int) -> np.ndarray:
        if self.instance == 0 or self.vel is None:
            return make_pose(np.eye(3), self.center)
        s = frame / 30.0
        t = self.center + self.vel * np.sin(self.freq * s)
        return make_pose(rot_xyz(0.0, self.spin * np.sin(0.6 * self.freq * s), 0.0), t)


class Scene:
    """Room [-2.5,2.5] x [-1.5,1.2] x [-1.0, zback] seen from around the origin looking down +z (y down)."""

    def __init__(self, n_objects: int = 0, seed: int = 1234, zback: float = 2.8, object_motion: float = 1.0, ring_slots=()):
        rng = np.random.RandomState(seed)
        self.zback = zback
        self.planes = [  # (axis, value, base colour)
            (2, zback, np.array([150, 140, 120.0])),
            (0, -2.5, np.array([120, 150, 130.0])),
            (0, 2.5, np.array([130, 120, 150.0])),
            (1, 1.2, np.array([140, 130, 110.0])),
            (1, -1.5, np.array([160, 160, 165.0])),
        ]
        self.boxes: list[Box] = []
        # 6 static boxes standing on the floor / hanging on the back wall
        static = [
            ((-1.4, 0.8, 2.2), (0.35, 0.40, 0.30)), ((1.3, 0.7, 2.3), (0.30, 0.50, 0.25)),
            ((0.0, 0.9, 2.4), (0.45, 0.30, 0.20)), ((-0.6, 1.0, 1.6), (0.20, 0.20, 0.20)),
            ((0.8, 1.0, 1.5), (0.18, 0.20, 0.22)), ((0.2, -0.9, 2.6), (0.50, 0.25, 0.15)),
        ]
        for c, h in static:
            self.boxes.append(Box(np.array(c, float), np.array(h, float), rng.uniform(60, 220, 3)))
        for k in range(n_objects):
            # ring_slots: which position of the ring instance k + 1 stands on (default: its own).  A box level with the camera (slots 0 and
            # n / 2 of the ring) shows TWO faces: a tracker on its exact planes is rank-deficient by construction (maskfusion_amd/stress.py)
            slot = ring_slots[k] if k < len(ring_slots) else k
            ang = 2 * np.pi * slot / max(n_objects, 1)
            c = np.array([1.1 * np.cos(ang) * 0.9, 0.15 + 0.45 * np.sin(ang), 1.45 + 0.25 * np.cos(2 * ang)])
            h = rng.uniform(0.10, 0.18, 3)
            self.boxes.append(Box(c, h, rng.uniform(60, 240, 3), instance=k + 1,
                                  vel=rng.uniform(-0.12, 0.12, 3) * object_motion, freq=rng.uniform(0.6, 1.4),
                                  spin=rng.uniform(0.1, 0.4) * object_motion))

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _texture(p: np.ndarray, base: np.ndarray, phase: float) -> np.ndarray:
        x, y, z = p[..., 0], p[..., 1], p[..., 2]
        a = np.sin(7.0 * x + phase) * np.cos(5.0 * y - 0.7 * phase) + 0.6 * np.sin(9.0 * z + 2.0 * x + phase)
        b = np.cos(6.0 * x - 4.0 * y + 1.3 * phase) * np.sin(3.0 * z + phase)
        c = np.sin(11.0 * (x + y + z) + phase)
        col = np.stack([base[0] + 45 * a + 20 * c, base[1] + 45 * b - 15 * c, base[2] + 30 * (a - b)], -1)
        return np.clip(col, 1, 255)

    def render(self, T_wc: np.ndarray, frame: int, W: int, H: int, fx: float, fy: float, cx: float, cy: float,
               max_depth: float = 0.0):
        """Returns (rgb uint8 HxWx3, depth float32 HxW, mask uint8 HxW)."""
        u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        d_cam = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1)
        R, t = T_wc[:3, :3], T_wc[:3, 3]
        d = d_cam @ R.T
        o = t
        best = np.full((H, W), np.inf)
        rgb = np.zeros((H, W, 3))
        mask = np.zeros((H, W), np.uint8)
        with np.errstate(divide="ignore", invalid="ignore"):
            for k, (axis, val, base) in enumerate(self.planes):
                tt = (val - o[axis]) / d[..., axis]
                hit = (tt > 1e-6) & (tt < best)
                p = o + tt[..., None] * d
                best = np.where(hit, tt, best)
                rgb = np.where(hit[..., None], self._texture(p, base, 0.9 * k), rgb)
                mask = np.where(hit, 0, mask).astype(np.uint8)
            for k, bx in enumerate(self.boxes):
                T_wo = bx.pose(frame)
                Ro, to = T_wo[:3, :3], T_wo[:3, 3]
                oo = (o - to) @ Ro          # world -> object
                do = d @ Ro
                inv = 1.0 / do
                t1 = (-bx.half - oo) * inv
                t2 = (bx.half - oo) * inv
                tn = np.minimum(t1, t2).max(-1)
                tf = np.maximum(t1, t2).min(-1)
                hit = (tn <= tf) & (tn > 1e-6) & (tn < best)
                p_obj = oo + tn[..., None] * do   # texture in the object frame: moves with the object
                best = np.where(hit, tn, best)
                rgb = np.where(hit[..., None], self._texture(p_obj * 1.7, bx.color, 1.7 + 0.37 * k), rgb)
                mask = np.where(hit, bx.instance, mask).astype(np.uint8)
        depth = np.where(np.isfinite(best), best, 0.0)
        if max_depth > 0:
            depth = np.where(depth > max_depth, 0.0, depth)
        return rgb.astype(np.uint8), depth.astype(np.float32), mask


def add_sensor_noise(depth: np.ndarray, seed: int, hole_frac: float = 0.02, hole_radius: int = 3) -> np.ndarray:
    """Kinect-like axial noise sigma_z = 0.0012 + 0.0019 (z-0.4)^2 plus missing-depth holes.

    Holes cover ~`hole_frac` of the image as (2r+1)^2 blobs: real sensors lose depth in patches (edges, specular
    spots), not in independent pixels -- i.i.d. pixel drop-outs would wipe out the coarse pyramid levels, where one
    invalid pixel invalidates a whole 4x4 cell (resizeMapKernel, Core/Cuda/cudafuncs.cu:385-389).
    """
    rng = np.random.RandomState(seed)
    sigma = 0.0012 + 0.0019 * (depth - 0.4) ** 2
    out = depth + rng.standard_normal(depth.shape).astype(np.float32) * sigma
    holes = np.zeros(depth.shape, bool)
    side = 2 * hole_radius + 1
    n_seed = int(hole_frac * depth.size / (side * side))
    ys = rng.randint(0, depth.shape[0], n_seed)
    xs = rng.randint(0, depth.shape[1], n_seed)
    for y, x in zip(ys, xs):
        holes[max(0, y - hole_radius):y + hole_radius + 1, max(0, x - hole_radius):x + hole_radius + 1] = True
    out = np.where(holes | (depth <= 0), 0.0, out)
    return out.astype(np.float32)


@dataclasses.dataclass
class Stream:
    W: int = 640
    H: int = 480
    fx: float = 528.0
    fy: float = 528.0
    cx: float = 320.0
    cy: float = 240.0
    n_objects: int = 0
    seed: int = 1234
    noise: bool = False
    speed: float = 1.0
    max_depth: float = 0.0
    object_motion: float = 1.0   # 0 = the instance-masked boxes stand still
    masked_objects: int = -1     # >= 0: only instance ids 1..masked_objects carry a mask; the other object boxes are rendered as furniture (mask 0)
    ring_slots: tuple = ()       # Scene: the ring position of every instance (default: instance k + 1 on position k)

    def __post_init__(self):
        self.scene = Scene(self.n_objects, self.seed, object_motion=self.object_motion, ring_slots=tuple(self.ring_slots))

    def gt_pose(self, frame: int) -> np.ndarray:
        return camera_pose(frame, self.speed)

    def frame(self, k: int):
        rgb, depth, mask = self.scene.render(self.gt_pose(k), k, self.W, self.H, self.fx, self.fy, self.cx, self.cy,
                                             self.max_depth)
        if self.noise:
            depth = add_sensor_noise(depth, 99 + k)
        if self.masked_objects >= 0:
            mask = np.where(mask > self.masked_objects, 0, mask).astype(np.uint8)
        return rgb, depth, mask


def _rect_cells(la, lb, spacing):
    return max(1, int(round(la / spacing))), max(1, int(round(lb / spacing)))


def _rect_records(origin, ea, eb, la, lb, normal, spacing, base, phase, conf, init_time, last_time, radius, dst=None):
    """surfel records on the rectangle origin + a ea + b eb, a in [0, la], b in [0, lb], one per `spacing` x `spacing` cell (cell centres,
    a-major order -- spatially coherent like the creation order of a real map), all with the same normal / radius / confidence / stamps"""
    na, nb = _rect_cells(la, lb, spacing)
    av = (np.arange(na, dtype=np.float32) + 0.5) * np.float32(la / na)
    b = ((np.arange(nb, dtype=np.float32) + 0.5) * np.float32(lb / nb))[None, :]
    out = (np.empty((na * nb, 12), np.float32) if dst is None else dst).reshape(na, nb, 12)   # dst: a (na * nb, 12) slice of the caller's array
    rows = max(1, 16384 // nb)      # a block of records that stays in cache while its twelve interleaved columns are written
    for r0 in range(0, na, rows):
        o = out[r0:r0 + rows]
        a = av[r0:r0 + rows, None]
        for k in range(3):
            o[..., k] = np.float32(origin[k]) + a * np.float32(ea[k]) + b * np.float32(eb[k])
        x, y, z = o[..., 0], o[..., 1], o[..., 2]
        # smooth procedural colour (cheaper than Scene._texture: 26 M records are generated per call), packed as surfels carry it
        r = np.clip(base[0] + 45.0 * np.sin(7.0 * x + 5.0 * y + np.float32(phase)), 1, 255).astype(np.int32)
        g = np.clip(base[1] + 45.0 * np.sin(6.0 * z - 4.0 * y + np.float32(1.3 * phase)), 1, 255).astype(np.int32)
        bl = np.clip(base[2] + 30.0 * np.sin(9.0 * (x + z) + np.float32(phase)), 1, 255).astype(np.int32)
        o[..., 3] = conf
        o[..., 4] = ((r << 16) + (g << 8) + bl).astype(np.float32)
        o[..., 5] = 0.0
        o[..., 6] = init_time
        o[..., 7] = last_time
        o[..., 8], o[..., 9], o[..., 10] = normal
        o[..., 11] = radius
    return out.reshape(-1, 12)


def _box_faces(half):
    """the six faces of an axis-aligned box [-half, half]^3 as (origin, ea, eb, la, lb, normal); normals point INTO the box, the direction
    a depth camera's normals have (away from the viewer: geometry.glsl:28-40 on a fronto-parallel wall gives +z)"""
    faces = []
    for ax in range(3):
        a1, a2 = (ax + 1) % 3, (ax + 2) % 3
        for sgn in (-1.0, 1.0):
            o = -np.asarray(half, float)
            o[ax] = sgn * half[ax]
            ea, eb, n = np.zeros(3), np.zeros(3), np.zeros(3)
            ea[a1], eb[a2], n[ax] = 1.0, 1.0, -sgn
            faces.append((o, ea, eb, 2 * half[a1], 2 * half[a2], n))
    return faces


def dense_room_map(scene: "Scene", n_target: int, last_time: float, conf: float = 20.0, init_time: float = 1.0, zfront: float = -1.0,
                   furniture_above: int = 1 << 30):
    """A pre-filled BACKGROUND map ("pre-filled to >= 80 % capacity by a long orbit"): ~n_target surfels at uniform
    density on the room's five planes and its static boxes, world coordinates = the background model's frame (camera 0 = identity).
    Radius = sqrt(2) x spacing, what a surfel created at one surfel per pixel carries (surfels.glsl getRadius); every record is stable
    (confidence above the threshold) and active (last seen at `last_time`), with equal initTime so that Model::clean's duplicate rule
    (copy_unstable.vert:100-106 needs an OLDER neighbour) keeps the density.  Standing object boxes whose instance id is above
    `furniture_above` carry no mask (Stream.masked_objects) and belong to the room.  Returns (n, 12) float32 in Model::downloadMap's layout."""
    x0, x1, y0, y1, z0, z1 = -2.5, 2.5, -1.5, 1.2, zfront, scene.zback
    rects = [  # (origin, ea, eb, la, lb, normal pointing out of the room, base colour)
        ((x0, y0, z1), (1, 0, 0), (0, 1, 0), x1 - x0, y1 - y0, (0, 0, 1), scene.planes[0][2]),
        ((x0, y0, z0), (0, 0, 1), (0, 1, 0), z1 - z0, y1 - y0, (-1, 0, 0), scene.planes[1][2]),
        ((x1, y0, z0), (0, 0, 1), (0, 1, 0), z1 - z0, y1 - y0, (1, 0, 0), scene.planes[2][2]),
        ((x0, y1, z0), (1, 0, 0), (0, 0, 1), x1 - x0, z1 - z0, (0, 1, 0), scene.planes[3][2]),
        ((x0, y0, z0), (1, 0, 0), (0, 0, 1), x1 - x0, z1 - z0, (0, -1, 0), scene.planes[4][2]),
    ]
    for bx in scene.boxes:
        if bx.instance != 0 and bx.instance <= furniture_above:
            continue
        for (o, ea, eb, la, lb, n) in _box_faces(bx.half):
            rects.append((o + bx.center, ea, eb, la, lb, n, bx.color))
    area = sum(r[3] * r[4] for r in rects)
    spacing = float(np.sqrt(area / n_target))
    sizes = [int(np.prod(_rect_cells(r[3], r[4], spacing))) for r in rects]
    out = np.empty((sum(sizes), 12), np.float32)
    at = 0
    for k, ((o, ea, eb, la, lb, n, base), sz) in enumerate(zip(rects, sizes)):
        _rect_records(o, ea, eb, la, lb, n, spacing, base, 0.9 * k, conf, init_time, last_time, np.float32(1.41421356 * spacing), out[at:at + sz])
        at += sz
    return out


def dense_object_map(box: "Box", n_target: int, T_obj_from_world: np.ndarray, last_time: float, conf: float = 20.0, init_time: float = 1.0):
    """The same for one (standing) instance-masked box: ~n_target surfels on its six faces, transformed from world coordinates into the
    OBJECT model's frame with `T_obj_from_world` (= objectPose . backgroundPose^-1)."""
    faces = _box_faces(box.half)
    area = sum(f[3] * f[4] for f in faces)
    spacing = float(np.sqrt(area / n_target))
    T = np.asarray(T_obj_from_world, np.float64) @ make_pose(np.eye(3), box.center)
    R, t = T[:3, :3], T[:3, 3]
    parts = []
    for k, (o, ea, eb, la, lb, n) in enumerate(faces):
        parts.append(_rect_records(R @ o + t, R @ ea, R @ eb, la, lb, (R @ n).astype(np.float32), spacing, box.color, 1.7 + 0.37 * k, conf,
                                   init_time, last_time, np.float32(1.41421356 * spacing)))
    return np.concatenate(parts)


def ate_rmse(est: np.ndarray, gt: np.ndarray) -> float:
    """Translational ATE RMSE between two (N,4,4) trajectories that share frame 0 (no alignment needed)."""
    d = est[:, :3, 3] - gt[:, :3, 3]
    return float(np.sqrt((d * d).sum(-1).mean()))
