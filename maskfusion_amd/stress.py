"""configs[4] as BASELINE.json / SURVEY.md 8d S3 define it: 1280x960, MASKFUSION_NUM_GSURFELS = 32M / NUM_OSURFELS = 4M, 4 objects, the maps
pre-filled to >= 80 % of their capacity so that the surfel passes (Core/Model/Model.cpp:466-772) are bandwidth-relevant.

A long orbit would fill the maps; here they are generated on the scene's surfaces (synth.dense_room_map / dense_object_map) and loaded with
Model.uploadMap while the scene is driven through its lead-in frames:

  frame 0            the background map is initialised from the frame (Model::initialise);
  every spawn        the label stage spawns an object model for an instance-masked box (modelSpawnOffset = 2: at most one every other
                     frame); its ~10^4-surfel map is replaced AT ONCE by a dense one on the same box -- confident surfels, so that the
                     box is drawn by GlobalProjection (fixed confidence threshold 12, GlobalProjection.cpp:61) from the next frame on and
                     the following spawn goes to ANOTHER box (left alone, a fresh model's confidence needs ~12 frames to get there and
                     the same mask spawns again and again);
  n_objects models   the background's map is replaced by the dense room map; the scene is ready.

bench.py (--config 4, variants.config4_stress) and tests/test_gpu_parity_long.py::test_config4_dense_maps share this driver; the test hands in
callbacks that take the oracle through the same frames and uploads.  The objects stand still and follow the camera (trackAllModels off): every
frame is then comparable with the oracle, and none of the dense maps is lost to the 0.2 m jump rule of an ill-conditioned object tracker.
"""
from __future__ import annotations

import numpy as np

from . import synth

W, H, F = 1280, 960, 1056.0
NUM_GSURFELS, NUM_OSURFELS = 32 * 1024 * 1024, 4 * 1024 * 1024
SEG_PARAMS = (("mfThreshold", 0.3), ("mfWeightDistance", 150.0), ("mfWeightConvexity", 2.8), ("mfMorphEdgeIterations", 0),
              ("mfMorphMaskIterations", 0), ("newModelMinRelativeSize", 0.004))


def surfel_capacity(num: int) -> int:
    """Model::TEXTURE_DIMENSION^2 (Core/Model/Model.cpp:101-108): 64 * floor(sqrt(num) / 64), squared"""
    d = 64 * int(np.sqrt(float(num)) / 64)
    return d * d


SCENE_BOXES = 6       # object boxes in the scene: with synth.Scene's ring of six, four boxes are in view with THREE faces each
# The instance-masked boxes 1..4 stand on the ring positions from which the camera sees three of their faces (below and above its axis); the two
# positions level with the camera (two faces: a Gauss-Newton system on a dense map's exact planes is rank 5 there -- the edge direction slides)
# go to the unmasked boxes 5 and 6.  Round 6: S3 tracks its objects (track_objects), so their trackers have to be conditioned.
RING_SLOTS = (1, 2, 4, 5, 0, 3)


def stream_kwargs(n_objects: int = 4, noise: bool = True, scale: int = 1) -> dict:
    """scale > 1: the same scene at 1 / scale of the resolution (the CPU-executed rehearsal of the scenario, tests/test_emu_dense_maps.py)"""
    w, h, f = W // scale, H // scale, F / scale
    return dict(W=w, H=h, fx=f, fy=f, cx=w / 2.0, cy=h / 2.0, n_objects=SCENE_BOXES, noise=noise, object_motion=0.0, seed=1234, masked_objects=n_objects,
                ring_slots=RING_SLOTS)


def stream(n_objects: int = 4, noise: bool = True, scale: int = 1) -> synth.Stream:
    """S3: the room with six standing boxes of which the first `n_objects` are instance-masked (the others are furniture)"""
    return synth.Stream(**stream_kwargs(n_objects, noise, scale))


def make_context(device: int = 0, num_g: int = NUM_GSURFELS, num_o: int = NUM_OSURFELS, n_objects: int = 4, scale: int = 1):
    """the product's context for the scenario (SURVEY.md 8d S2 / S3 settings: confG = 10, confO = 0.01, the GUI's segmentation parameters)"""
    from . import MaskFusion
    w, h, f = W // scale, H // scale, F / scale
    mf = MaskFusion(w, h, f, f, w / 2.0, h / 2.0, icpThresh=100.0, so3=False, device=device, enableMultipleModels=True, numGSurfels=num_g,
                    numOSurfels=num_o, trackAllModels=False, modelSpawnOffset=2, initConfidenceGlobal=10.0, initConfidenceObject=0.01)
    for k, v in SEG_PARAMS:
        mf.setParam(k, v)
    mf.preallocateModels(n_objects)
    return mf


def box_of_model(mf, index: int, st: synth.Stream, taken=()):
    """which instance-masked box the object model at `index` was spawned on: the box nearest to the centroid of its surfels (object frame ->
    world: backgroundPose . objectPose^-1); returns (box, objectPose . backgroundPose^-1)"""
    models = mf.getModels()
    T = models[index].getPose() @ np.linalg.inv(models[0].getPose())
    s = models[index].downloadMap()
    cw = (np.linalg.inv(T) @ np.r_[s[:, :3].astype(np.float64).mean(0), 1.0])[:3]
    box = min((b for b in st.scene.boxes if 0 < b.instance <= st.masked_objects and b.instance not in taken), key=lambda b: float(np.linalg.norm(b.center - cw)))
    return box, T


def lead_in(mf, st: synth.Stream, frames, cls, n_objects: int = 4, fill: float = 0.8, max_frames: int = 24, on_frame=None, on_upload=None,
            log=None, room_map=None):
    """Drives `mf` through the lead-in (module docstring).  frames[k] = (rgb, depth, mask) or a callable k -> that.
    on_frame(k, rgb, depth, mask): called after the product has processed frame k (the test runs the oracle's frame there);
    on_upload(model_index, surfels): called with every map that is uploaded.  room_map: a background map generated ahead of time
    (synth.dense_room_map; its lastTime column is set here).  Returns (index of the next frame, {model index: surfel count})."""
    get = frames if callable(frames) else (lambda k: frames[k])
    cap_g = surfel_capacity(mf.cfg.num_gsurfels)
    cap_o = surfel_capacity(mf.cfg.num_osurfels)
    taken, loaded = [], {}
    k = 0
    while True:
        rgb, depth, mask = get(k)
        mf.processFrame(rgb, depth, mask=mask, classIDs=cls, timestamp=k)
        if on_frame:
            on_frame(k, rgb, depth, mask)
        tick = k + 1                                 # the tick frame k ran with: "last seen in this frame"
        n_models = len(mf.getModels())
        for i in range(1, n_models):
            if i in loaded:
                continue
            box, T = box_of_model(mf, i, st, taken)
            taken.append(box.instance)
            m = synth.dense_object_map(box, int(1.01 * fill * cap_o), T, last_time=float(tick))
            mf.getModels()[i].uploadMap(m)
            if on_upload:
                on_upload(i, m)
            loaded[i] = len(m)
            if log:
                log(f"frame {k}: object model {i} (id {mf.getModels()[i].getID()}) on box {box.instance}: {len(m)} surfels of {cap_o}")
        k += 1
        if len(loaded) >= n_objects or k >= max_frames:
            break
    if room_map is None:
        bg = synth.dense_room_map(st.scene, int(1.005 * fill * cap_g), last_time=float(k), furniture_above=st.masked_objects)
    else:
        bg = room_map
        bg[:, 7] = float(k)
    mf.getBackgroundModel().uploadMap(bg)
    if on_upload:
        on_upload(0, bg)
    loaded[0] = len(bg)
    if log:
        log(f"after frame {k - 1}: background map {len(bg)} surfels of {cap_g} ({len(bg) / cap_g:.1%}); {len(loaded) - 1} object models")
    return k, loaded


def track_objects(mf, on_model=None):
    """S3 = S2's settings at 1280x960, and S2 tracks every model: makes every object model of `mf` non-static
    (Model::makeNonStatic, Core/Model/Model.h:263-268 -- from the next frame on MaskFusion.cpp:263-276 tracks it instead of moving it with the
    camera).  The lead-in itself runs with standing objects: a fresh ~10^4-surfel model tracked before its dense map is loaded is at the mercy
    of the 0.2 m rule.  on_model(index): called for every model switched (the parity test switches the oracle's model there)."""
    for i, x in enumerate(mf.getModels()):
        if i == 0:
            continue
        x.makeNonStatic()
        if on_model:
            on_model(i)
