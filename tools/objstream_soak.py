"""Soak check of the two-stream multi-model frame ("objectStream", "batchSolveInPixelPass", "fusedPreprocessLaunch"): an 8-object scene with tracked,
spawned and dropped objects for N frames, once with every round-6 launch re-arrangement off and REPS times with all of them on; every frame's model
list, poses, surfel counts and label image must be the same bits (what a missing dependency between the streams would break, sooner or later).
    PYTHONPATH=. python tools/objstream_soak.py [frames] [reps]"""
import hashlib
import sys
import numpy as np
from maskfusion_amd import MaskFusion, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H, F = 640, 480, 528.0
st = synth.Stream(W=W, H=H, fx=F, fy=F, cx=W / 2.0, cy=H / 2.0, n_objects=8, noise=True, object_motion=1.0)
frames = [st.frame(k) for k in range(N)]
cls = [0] + [41 + i for i in range(8)]
h = lambda a: hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]


def run(**params):
    mf = MaskFusion(W, H, F, F, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, numGSurfels=1 << 21, numOSurfels=1 << 18, enableMultipleModels=True,
                    modelSpawnOffset=2, trackAllModels=True)
    for k, v in dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0,
                     newModelMinRelativeSize=0.004, **params).items():
        mf.setParam(k, v)
    rec, most = [], 0
    for k, (rgb, d, m) in enumerate(frames):
        mf.processFrame(rgb, d, mask=m, classIDs=cls, timestamp=k)
        ms = mf.getModels()
        most = max(most, len(ms))
        rec.append((tuple(x.getID() for x in ms), h(np.stack([x.getPose() for x in ms])), tuple(x.lastCount() for x in ms), h(mf.downloadSegmentation())))
    clouds = [h(x.downloadMap()) for x in mf.getModels()]
    mf.close()
    return rec, clouds, most


base, base_clouds, most = run(objectStream=0, batchSolveInPixelPass=0, fusedPreprocessLaunch=0)
print(f"{N} frames, up to {most} models, {len(set(r[0] for r in base))} different model lists along the way")
bad = 0
for r in range(REPS):
    rec, clouds, _ = run()
    first = next((k for k, (a, b) in enumerate(zip(base, rec)) if a != b), None)
    ok = first is None and clouds == base_clouds
    bad += 0 if ok else 1
    print(f"  run {r + 1} with the defaults: {'identical' if ok else 'DIFFERS from frame ' + str(first)}")
sys.exit(1 if bad else 0)
