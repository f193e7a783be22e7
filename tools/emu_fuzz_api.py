"""API-sequence fuzzing without a GPU: random sequences of calls on the C ABI (through maskfusion_amd.api) against a context of the product's
kernels EXECUTED ON THE CPU -- frames, stand-alone predictions, parameter and implementation switches flipped between frames, tick
jumps, model-level calls, preallocation, spawn / drop, static / non-static, exports, small download buffers, invalid indices.  Nothing is
compared: the point is that no sequence crashes, corrupts memory (run it on the AddressSanitizer build) or bricks the context -- after
every sequence a plain frame must still go through.  Development tooling.
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 MF_EMU_ASAN=1 python tools/emu_fuzz_api.py 30"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hipcpu"))
import numpy as np  # noqa: E402

import emu  # noqa: E402

emu.activate()
from maskfusion_amd import MaskFusion, synth  # noqa: E402
from maskfusion_amd.lib import MFError  # noqa: E402

SWITCHES = ["splatTiles", "globalTiles", "gpuLabels", "batchTracking", "earlyBackgroundFusion", "cleanLiteralWindow",
            "timings"]


def one(seed, tmp):
    rng = np.random.default_rng(seed)
    W, H = [(160, 120), (200, 152), (240, 160), (128, 96)][int(rng.integers(0, 4))]
    f = 528.0 * W / 640.0
    multi = bool(rng.integers(0, 2))
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, n_objects=int(rng.integers(1, 3)) if multi else 0, noise=True, seed=int(seed))
    cap = int(rng.choice([64 * 64, 128 * 128, 1 << 17]))           # 4096: the map overflows its buffer on the first frame (capacities are (64 k)^2, Model.cpp:101-105)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=float(rng.choice([100.0, 20.0])), so3=bool(rng.integers(0, 2)), enableMultipleModels=multi,
                    numGSurfels=cap, numOSurfels=int(rng.choice([64 * 64, 128 * 128])), modelSpawnOffset=int(rng.integers(1, 4)),
                    trackAllModels=bool(rng.integers(0, 2)))
    if multi:
        for k, v in dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0, newModelMinRelativeSize=0.004).items():
            mf.setParam(k, v)
    log = []
    k = 0
    expected_errors = 0
    for step in range(int(rng.integers(8, 20))):
        op = int(rng.integers(0, 16))
        log.append(op)
        try:
            if op <= 4:
                rgb, d, mask = st.frame(k % 16)
                k += 1
                if rng.integers(0, 8) == 0:
                    d = np.zeros_like(d)
                if multi:
                    mf.processFrame(rgb, d, mask=mask, classIDs=[0, 41, 42][:int(rng.integers(1, 4))], timestamp=k)
                elif rng.integers(0, 5) == 0 and k > 1:
                    mf.processFrame(rgb, d, timestamp=k, inPose=st.gt_pose(k % 16).astype(np.float32))
                else:
                    mf.processFrame(rgb, d, timestamp=k)
            elif op == 5:
                mf.predict()
            elif op == 6:
                if rng.integers(0, 3) == 0:      # the size thresholds of the fuse / clean forms (round 5): any combination, changed between frames
                    mf.setParam(str(rng.choice(["bigMapElements", "inPlaceElements"])), float(rng.choice([0, 1 << 30])))
                    mf.setParam("cullRuns", float(rng.integers(0, 2)))
                else:
                    mf.setParam(SWITCHES[int(rng.integers(0, len(SWITCHES)))], float(rng.integers(0, 2)))
            elif op == 7:
                mf.setParam(str(rng.choice(["depthCutoff", "confidenceThreshold", "outlierCoefficient", "icpWeight", "fastOdom", "pyramid", "so3"])),
                            float(rng.choice([0.0, 0.5, 1.0, 3.0, 10.0, 100.0])))
            elif op == 8:
                mf.preallocateModels(int(rng.integers(0, 3)))
            elif op == 9:
                ms = mf.getModels()
                m = ms[int(rng.integers(0, len(ms)))]
                m.info(); m.getPose(); m.lastCount(); m.downloadMap(); m.getICPStats()
                mf.getPoseLog(int(rng.integers(0, len(ms))))
            elif op == 10:
                mf.savePly(tmp + os.sep)
                mf.exportPoses(tmp + os.sep)
                if k > 0:
                    mf.exportSegmentation(os.path.join(tmp, "seg.png"))
            elif op == 11:
                ms = mf.getModels()
                i = int(rng.integers(0, len(ms) + 2))               # possibly out of range: must be an error, not a crash
                if i >= len(ms):
                    expected_errors += 1
                if rng.integers(0, 2):
                    mf._chk(mf._L.mf_make_nonstatic(mf._h, i))
                else:
                    mf._chk(mf._L.mf_make_static(mf._h, i))
            elif op == 12 and k > 0:
                rgb, d, mask = st.frame(k % 16)
                mf.stageFrame(rgb, d, mask if multi else None)
                ms = mf.getModels()
                t = mf.getTick()
                for m in ms:
                    m.performTracking(False, False, 100.0, True, bool(rng.integers(0, 2)), False, 20.0, k, False)
                    m.predictIndices(t, 20.0, 200)
                    m.fuse(t, 3.0, 1.0)
                    m.predictIndices(t, 20.0, 200)
                    m.clean(t, 200, 20.0)
                    m.combinedPredict(20.0, t, t, 200)
                mf.endFrame(k)
                k += 1
            elif op == 13:
                mf.setTick(int(rng.integers(1, 500)))
            elif op == 14 and multi:
                ms = mf.getModels()
                if len(ms) > 1 and rng.integers(0, 2):
                    mf._chk(mf._L.mf_drop_model(mf._h, int(rng.integers(1, len(ms)))))
                else:
                    mf._chk(mf._L.mf_spawn_object_model(mf._h, int(rng.integers(1, 40)), 41))
            elif op == 15:
                mf.setTrackableClassIds([41] if rng.integers(0, 2) else [])
                mf.downloadSegmentation()
                mf.timings()
        except MFError:
            pass                                                    # an error code is a fine answer to a nonsensical call; a crash is not
    # the context must still work
    rgb, d, mask = st.frame(k % 16)
    if multi:
        mf.processFrame(rgb, d, mask=mask, classIDs=[0, 41, 42], timestamp=k + 1)
    else:
        mf.processFrame(rgb, d, timestamp=k + 1)
    n = mf.getBackgroundModel().lastCount()
    pose = mf.getCurrPose()
    ok = np.isfinite(pose).all() and 0 <= n <= cap
    mf.close()
    return f"seed {seed}: {W}x{H} multi={multi} cap={cap} ops={log}", ok, n


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    t0, nbad = time.time(), 0
    with tempfile.TemporaryDirectory() as tmp:
        for s in range(seed0, seed0 + n_cases):
            desc, ok, n = one(s, tmp)
            nbad += not ok
            print("ok      " if ok else "BROKEN  ", desc, "surfels", n, flush=True)
    print(f"{n_cases} sequences, {nbad} left the context broken, {time.time() - t0:.0f} s")
