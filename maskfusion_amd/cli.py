"""Headless driver with the reference's command-line flags (GUI/MainController.cpp:99-344).

    python -m maskfusion_amd.cli -l log.klg -run -q -ep -em -exportdir out/
    python -m maskfusion_amd.cli -dir seq/ -maskdir seq/masks/ -tum3 -run -q -ep

Flags keep their upstream names and meaning; GUI-only flags (-sc, -ev, -el, -en, -es, -run, -q ...) are accepted and ignored
(this driver always runs to the end of the log and quits).  Parameter defaults are the ones the GUI pushes into the core
every frame (GUI/Tools/GUI.h:188-196,342-347,367-374; MainController.cpp:215-228,528-571): depth cutoff 4 m, ICP
weight 20, outlier coefficient 0.1, confidence 10 / 0.01, spawn offset 22, open loop, trackAllModels OFF, MfSegmentation
threshold 0.3 / weights 150, 2.8 / morphology 0x1, 0x2, new-model size 0.015 .. 0.4.  The frame queue (-frameQ, default 30) is
forced to 0 exactly as upstream does with precomputed masks (MaskFusion.cpp:37): masks here always come from -maskdir / the log.
Trackable classes come from config.toml ([MaskRCNN] class_names / trackable_classes, MainController.cpp:273-287) when it exists
in the working directory (upstream refuses to start without it; here: every class is trackable then).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

_VALUE_FLAGS = {"-l", "-dir", "-depthdir", "-maskdir", "-colorprefix", "-depthprefix", "-maskprefix", "-indexW", "-cal", "-basedir",
                "-exportdir", "-d", "-i", "-or", "-confG", "-confO", "-s", "-e", "-nm", "-offset", "-t", "-ie", "-cv", "-pt", "-ft",
                "-ic", "-a", "-frameQ", "-method", "-p", "-segMinNew", "-segMaxNew", "-thNew", "-gpu", "-name", "-k", "-crfRGB", "-crfDepth",
                "-crfPos", "-crfAppearance", "-crfSmooth"}
_BOOL_FLAGS = {"-static", "-run", "-q", "-ep", "-em", "-es", "-ev", "-el", "-en", "-fo", "-nso", "-f", "-tum3", "-v2", "-icl", "-rl",
               "-fs", "-r", "-ftf", "-sc", "-keep", "-o", "-v1", "-rgbonly"}


def parse(argv):
    """Parse::arg semantics (Core/Utils/Parse.cpp): a flag is present if it appears; value flags take the next token."""
    out, i = {}, 0
    while i < len(argv):
        a = argv[i]
        if a in _VALUE_FLAGS:
            if i + 1 >= len(argv):
                raise SystemExit(f"flag {a} needs a value")
            out[a] = argv[i + 1]
            i += 2
        elif a in _BOOL_FLAGS:
            out[a] = True
            i += 1
        else:
            raise SystemExit(f"unknown flag: {a}")
    return out


def settings(flags):
    """Resolution / intrinsics (MainController.cpp:117-128) and the core parameters the GUI sets per frame."""
    if "-v2" in flags:
        W, H, fx, fy, cx, cy = 512, 424, 528.0, 528.0, 256.0, 212.0
    elif "-tum3" in flags:
        W, H, fx, fy, cx, cy = 640, 480, 535.4, 539.2, 320.1, 247.6
    else:
        W, H, fx, fy, cx, cy = 640, 480, 528.0, 528.0, 320.0, 240.0
    cal = flags.get("-cal")
    if flags.get("-method", "") == "cofusion":
        raise SystemExit("-method cofusion: the Co-Fusion CRF segmentation is out of scope of this build (DESIGN.md section 1)")
    return dict(W=W, H=H, fx=fx, fy=fy, cx=cx, cy=cy, cal=cal,
                trackAllModels=False,                                   # GUI/Tools/GUI.h:344 "oi.Track all models" = false
                mf=dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphEdgeRadius=1,
                        mfMorphMaskIterations=0, mfMorphMaskRadius=2,  # GUI.h:367-374, pushed every frame (MainController.cpp:556-567)
                        newModelMinRelativeSize=float(flags.get("-segMinNew", 0.015)),      # GUI.h:345, MainController.cpp:293
                        newModelMaxRelativeSize=float(flags.get("-segMaxNew", 0.4))),       # GUI.h:346, MainController.cpp:294
                preallocate=int(flags.get("-a", 0)),                    # MainController.cpp:221,240,407
                frameQueueRequested=int(flags.get("-frameQ", 30)), frameQueue=0,   # MainController.cpp:223,241; MaskFusion.cpp:37
                exportSegmentation="-es" in flags,
                depthCutoff=float(flags.get("-d", 4.0)), icpWeight=float(flags.get("-i", 20.0)),
                outlierCoefficient=float(flags.get("-or", 0.1)), confGlobal=float(flags.get("-confG", 10.0)),
                confObject=float(flags.get("-confO", 0.01)), so3="-nso" not in flags, fastOdom="-fo" in flags,
                multi="-static" not in flags, modelSpawnOffset=int(float(flags.get("-offset", 22))),
                timeDelta=(2 ** 31 - 1) // 2,   # openLoop = true (MainController.cpp:246)
                start=int(flags.get("-s", 1)), end=int(flags.get("-e", 65535)), rgbOnly="-rgbonly" in flags,
                flipColors="-f" in flags, device=int(flags.get("-gpu", 0)),
                frameToFrameRGB="-ftf" in flags)                       # MainController.cpp:252 -> MaskFusion::frameToFrameRGB


def trackable_class_ids(path="config.toml"):
    """MainController.cpp:273-287: ids of [MaskRCNN].trackable_classes within class_names; None when there is no config.toml"""
    if not os.path.exists(path):
        return None
    try:
        import tomllib as toml_reader          # Python >= 3.11
    except ImportError:
        try:
            import tomli as toml_reader        # its backport
        except ImportError as e:
            raise RuntimeError(f"{path} exists (upstream reads its [MaskRCNN] table, MainController.cpp:273-287) but neither tomllib "
                               "(Python >= 3.11) nor tomli is installed; remove the file or install tomli") from e
    with open(path, "rb") as f:
        cfg = toml_reader.load(f)["MaskRCNN"]
    names = list(cfg["class_names"])
    return sorted({names.index(c) if c in names else len(names) for c in cfg["trackable_classes"]})


def open_reader(flags, st):
    from .io import ImageLogReader, KlgLogReader, load_calibration
    base = flags.get("-basedir", "")
    if st["cal"]:
        fx, fy, cx, cy, w, h = load_calibration(os.path.join(base, st["cal"]))
        st.update(fx=fx, fy=fy, cx=cx, cy=cy)
        if w:
            st.update(W=w, H=h)
    if "-l" in flags:
        return KlgLogReader(os.path.join(base, flags["-l"]), st["W"], st["H"], flipColors=st["flipColors"])
    if "-dir" in flags:
        d = os.path.join(base, flags["-dir"])
        r = ImageLogReader(d, flags.get("-depthdir", ""), flags.get("-maskdir", ""), int(flags.get("-indexW", 4)),
                           flags.get("-colorprefix", ""), flags.get("-depthprefix", ""), flags.get("-maskprefix", ""),
                           flipColors=st["flipColors"], maxMasks=int(flags.get("-nm", -1)))
        if r.calibrationFile and not st["cal"]:
            fx, fy, cx, cy, w, h = load_calibration(r.calibrationFile)
            st.update(fx=fx, fy=fy, cx=cx, cy=cy)
            if w:
                st.update(W=w, H=h)
        return r
    raise SystemExit("no input: use -l <file.klg> or -dir <directory>")


def main(argv=None):
    flags = parse(sys.argv[1:] if argv is None else argv)
    st = settings(flags)
    reader = open_reader(flags, st)
    from .api import MaskFusion
    mf = MaskFusion(st["W"], st["H"], st["fx"], st["fy"], st["cx"], st["cy"], timeDelta=st["timeDelta"],
                    initConfidenceGlobal=st["confGlobal"], initConfidenceObject=st["confObject"], depthCut=st["depthCutoff"],
                    icpThresh=st["icpWeight"], fastOdom=st["fastOdom"], so3=st["so3"], device=st["device"],
                    enableMultipleModels=st["multi"], outlierCoefficient=st["outlierCoefficient"],
                    modelSpawnOffset=st["modelSpawnOffset"], rgbOnly=st["rgbOnly"], trackAllModels=st["trackAllModels"])
    for key, value in st["mf"].items():
        mf.setParam(key, value)
    if st["frameToFrameRGB"]:
        mf.setFrameToFrameRGB(True)
    ids = trackable_class_ids()
    if ids is not None:
        mf.setTrackableClassIds(ids)
    if st["preallocate"]:
        mf.preallocateModels(st["preallocate"])
    export_dir = flags.get("-exportdir", "")
    if export_dir and not export_dir.endswith(os.sep):
        export_dir += os.sep
    if export_dir:
        os.makedirs(export_dir, exist_ok=True)
    n, t0 = 0, time.time()
    for frame in reader:
        if mf.getTick() >= st["end"]:
            break
        if n + 1 < st["start"]:   # -s: skip ahead like fastForward(start)
            n += 1
            continue
        tick = mf.getTick()
        mf.processFrame(frame.rgb, frame.depth, mask=frame.mask, timestamp=int(frame.timestamp), classIDs=tuple(frame.classIDs))
        if st["exportSegmentation"] and st["multi"] and tick > 1:   # MaskFusion.cpp:299-303
            mf.exportSegmentation(os.path.join(export_dir, f"Segmentation{tick}.png"))
        n += 1
    dt = time.time() - t0
    models = mf.getModels()
    print(f"processed {n} frames in {dt:.2f} s ({n / max(dt, 1e-9):.1f} fps incl. decoding); models: "
          + ", ".join(f"id {m.getID()}: {m.lastCount()} surfels" for m in models))
    if "-ep" in flags:
        mf.exportPoses(export_dir)
    if "-em" in flags:
        mf.savePly(export_dir)
    pose = mf.getCurrPose()
    print("final camera pose:\n" + np.array2string(pose, precision=5, suppress_small=True))
    mf.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
