"""-m gpu: mf_set_param("gnLoopGraph", 1) -- the 19 launches + finalize of the geometric Gauss-Newton loop captured once per frame parity
with hipStreamBeginCapture / EndCapture and replayed as ONE hipGraphLaunch per tracking step -- must not change a bit: same kernels, same
arguments, same order.  Single model, and object models tracked one after the other (each model owns its graphs)."""
import numpy as np
import pytest

# Written when the round's GPU minutes were spent: it has run against the CPU-executed kernels (MF_EMU=1, whose runtime records and replays
# captured launches) but not yet on hardware -- non-strict xfail until a hardware run has been seen, and a time limit so that a runtime
# that mishandles the capture cannot stall the suite.
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]   # first hardware run: GPUTEST_r02 (XPASS); a plain test since round 3


def _run(graph, multi):
    from maskfusion_amd import MaskFusion, synth
    W, H = 320, 240
    f = 528.0 * W / 640.0
    st = synth.Stream(W=W, H=H, fx=f, fy=f, cx=W / 2.0, cy=H / 2.0, noise=True, n_objects=2 if multi else 0, object_motion=1.0)
    mf = MaskFusion(W, H, f, f, W / 2.0, H / 2.0, icpThresh=100.0, so3=False, enableMultipleModels=multi, numGSurfels=1 << 18, numOSurfels=1 << 16,
                    modelSpawnOffset=2, trackAllModels=True)
    if multi:
        for k, v in dict(mfThreshold=0.3, mfWeightDistance=150.0, mfWeightConvexity=2.8, mfMorphEdgeIterations=0, mfMorphMaskIterations=0,
                         newModelMinRelativeSize=0.004, batchTracking=0).items():
            mf.setParam(k, v)
    mf.setParam("gnLoopGraph", 1 if graph else 0)
    out = []
    for k in range(9):
        rgb, d, mask = st.frame(k)
        if k == 5:
            mf.setFastOdom(True)           # another iteration schedule: the graphs are captured again
        if multi:
            mf.processFrame(rgb, d, mask=mask, classIDs=[0, 41, 42], timestamp=k)
        else:
            mf.processFrame(rgb, d, timestamp=k)
        ms = mf.getModels()
        out.append(dict(ids=[m.getID() for m in ms], poses=[m.getPose() for m in ms], counts=[m.lastCount() for m in ms],
                        stats=[mf.trackStats(i) for i in range(len(ms))]))
    clouds = [m.downloadMap() for m in mf.getModels()]
    on = mf.getParam("gnLoopGraph")
    mf.close()
    return out, clouds, on


@pytest.mark.parametrize("multi", [False, True], ids=["single-model", "tracked-objects"])
def test_graph_replay_equals_eager_launches(hip, multi):
    eager, ce, _ = _run(False, multi)
    graph, cg, still_on = _run(True, multi)
    assert still_on == 1.0, "the runtime refused the stream capture: the eager launches were used instead"
    for k, (a, b) in enumerate(zip(eager, graph)):
        assert a["ids"] == b["ids"] and a["counts"] == b["counts"], k
        for pa, pb in zip(a["poses"], b["poses"]):
            assert np.array_equal(pa, pb), k
        for sa, sb in zip(a["stats"], b["stats"]):
            assert sa == sb or all((sa[q] == sb[q]) or (sa[q] != sa[q] and sb[q] != sb[q]) for q in sa), k
    assert len(ce) == len(cg) and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ce, cg))
    if multi:
        assert max(len(r["ids"]) for r in eager) >= 2
