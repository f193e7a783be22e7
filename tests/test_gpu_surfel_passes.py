"""-m gpu: every surfel pass of Model against the oracle IN ISOLATION, through the Model-level C ABI
(mf_stage_frame / mf_model_*).  The oracle's own surfel buffer, pose and lastPose after N frames are uploaded into the HIP
context; then each pass runs once on both sides for frame N+1 and its complete output is compared:

  pass (reference)                                   compared                                       gate
  predictIndices  index_map.vert:38-63               index image                                    exact
                                                     vertConf / colorTime / normRad images          1e-6
  fuse/data       data.vert:79-194                   candidate op per quarter-rate pixel            exact
                                                     candidate records                              1e-6 (normals 1e-5)
  fuse/update     update.vert:38-111                 every updated surfel                           1e-6
  clean           copy_unstable.vert:53-157          keep flag per surfel and per new record        exact
                                                     surviving surfels, count                       1e-6, exact
  combinedPredict splat.vert:54-88,combo_splat.frag  winner (time / colour) per pixel               exact
                                                     vertex / normal images                         1e-6
  fill-in (a18)   fill_*.frag, FillIn.cpp            model-side vertex / normal pyramid with fill-in  2e-6 / 2e-5
  computeFusionWeight (a15)  Model.cpp:449-464       weight                                         1e-6
Where a float tolerance is given the two sides run the same individually-rounded operations; a handful of ulp-level
differences come from libm (expf in the confidence, acosf in the normal test) and are listed when they occur."""
import numpy as np
import pytest

from gpu_util import scene_frames

pytestmark = pytest.mark.gpu

N_WARM = 9
CONF = 1.0          # low enough that a good part of the map is stable after N_WARM frames (stable surfels drive clean / splat)
TIME_DELTA = 200
DEPTH_CUT = 3.0
MAXD = 20.0
OUTLIER = 0.9


def _close(a, b, tol, what, rel=True):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    fin = np.isfinite(b)
    assert np.array_equal(fin, np.isfinite(a)), f"{what}: non-finite pattern differs"
    err = np.abs(a[fin] - b[fin])
    lim = tol * (np.maximum(1.0, np.abs(b[fin])) if rel else 1.0)
    bad = err > lim
    print(f"  {what}: {int((err > 0).sum())} of {err.size} not bit-identical, max err {err.max() if err.size else 0.0:.3g}")
    assert not bad.any(), f"{what}: {int(bad.sum())} of {err.size} beyond {tol}, max err {err.max()}"
    return float(err.max()) if err.size else 0.0


@pytest.fixture(scope="module", params=[(640, 480), (1280, 960)], ids=["vga", "1280x960"])
def world(request, hip, oracle):
    W, H = request.param
    from maskfusion_amd import MaskFusion
    st, frames = scene_frames(N_WARM + 1, W=W, H=H, noise=True)
    cap = 1 << (20 if W == 640 else 22)
    o = oracle.Oracle(W, H, st.fx, st.fy, st.cx, st.cy, icpWeight=100.0, capacity=cap, so3=0, confGlobal=CONF, timeDelta=TIME_DELTA,
                      depthCutoff=DEPTH_CUT, outlierCoeff=OUTLIER)
    poses = []
    for k in range(N_WARM):
        o.process_frame(frames[k][0], frames[k][1])
        poses.append(o.pose)
    S, n, t = o.surfels(), o.count, o.tick
    pred_prev = dict(image=o.dbg("pred_image"), vertex=o.dbg("pred_vertex"), normal=o.dbg("pred_normal"), depthF=o.dbg("depthF"))
    o.close()
    mf = MaskFusion(W, H, st.fx, st.fy, st.cx, st.cy, icpThresh=100.0, so3=False, numGSurfels=cap, enableMultipleModels=False,
                    initConfidenceGlobal=CONF, timeDelta=TIME_DELTA, depthCut=DEPTH_CUT, outlierCoefficient=OUTLIER)
    bg = mf.getBackgroundModel()
    bg.uploadMap(S[:n])
    # the pose the oracle holds after frame N_WARM-1 becomes pose, its predecessor lastPose (Model::overridePose twice)
    bg.overridePose(poses[-2]); bg.overridePose(poses[-1])
    mf.setTick(t)
    # frame N_WARM-1 staged first: its filtered depth is the fill-in source of the next tracking step; then frame N_WARM
    mf.stageFrame(frames[N_WARM - 1][0], frames[N_WARM - 1][1])
    rgb, depth, _ = frames[N_WARM]
    mf.stageFrame(rgb, depth)
    yield dict(W=W, H=H, st=st, cam=oracle.cam(W, H, st.fx, st.fy, st.cx, st.cy), S=S, n=n, t=t, T=poses[-1], T_last=poses[-2], rgb=rgb, depth=depth,
               mask=np.zeros((H, W), np.uint8), mf=mf, bg=bg, cap=cap, oracle=oracle, pred_prev=pred_prev, frames=frames)
    mf.close()


def test_surfel_passes_against_oracle(world):
    w = world
    mfo, cam, mf, bg = w["oracle"], w["cam"], w["mf"], w["bg"]
    S, n, t, T = w["S"], w["n"], w["t"], w["T"]
    print("map:", n, "surfels, stable fraction", (S[:n, 3] > CONF).mean())
    assert n > 200_000 and (S[:n, 3] > CONF).mean() > 0.1          # a populated map with a stable part
    depthF = mfo.bilateral(w["depth"])
    _close(mf.debugRead("depthF"), depthF, 2e-5, "filtered depth")
    depthF_dev = mf.debugRead("depthF")   # the passes below consume the device's own filtered depth on the device side

    # ---- a15: computeFusionWeight ----
    wgt_o = mfo.fusion_weight(T, w["T_last"], 1.0)
    wgt_g = bg.computeFusionWeight(1.0)
    assert abs(wgt_o - wgt_g) < 1e-6, (wgt_o, wgt_g)

    # ---- a13: predictIndices ----
    idx, vc, ct, nr = mfo.predict_indices(cam, T, S, n, t, MAXD, TIME_DELTA)
    bg.predictIndices(t, MAXD, TIME_DELTA)
    g_idx = mf.debugRead("index")
    neq = int((g_idx != idx).sum())
    print("index map: filled", int((idx > 0).sum()), "differing texels", neq)
    assert neq == 0
    _close(mf.debugRead("index_vc"), vc, 1e-6, "index vertConf")
    _close(mf.debugRead("index_nr"), nr, 1e-6, "index normRad")
    assert np.array_equal(mf.debugRead("index_ct"), ct)            # colour / times are copied, not computed

    # ---- a14: fuse / data.  The oracle pass gets the DEVICE's filtered depth so that only this pass is under test ----
    op, best, rec = mfo.fuse_data(cam, T, w["rgb"], w["depth"], depthF_dev, w["mask"], 0, t, wgt_o, DEPTH_CUT, idx, vc, nr)
    bg.fuse(t, DEPTH_CUT, 1.0)
    nc = len(op)
    g_op = mf.debugRead("cand_op", count=nc)
    g_rec = mf.debugRead("cand_rec", count=nc)
    flips = np.nonzero(g_op != op)[0]
    print("candidates", nc, "ops", np.bincount(op, minlength=3).tolist(), "op flips", len(flips))
    assert len(flips) == 0, (flips[:10], op[flips[:10]], g_op[flips[:10]])
    live = op > 0
    _close(g_rec[live][:, :4], rec[live][:, :4], 1e-6, "candidate position / confidence")
    assert np.array_equal(g_rec[live][:, 4:8], rec[live][:, 4:8])   # colour, time stamps: integer valued
    fin = np.isfinite(rec[live][:, 8:]).all(1) & np.isfinite(g_rec[live][:, 8:]).all(1)
    assert np.array_equal(np.isfinite(rec[live][:, 8:]), np.isfinite(g_rec[live][:, 8:]))
    _close(g_rec[live][fin][:, 8:11], rec[live][fin][:, 8:11], 1e-5, "candidate normal", rel=False)
    _close(g_rec[live][fin][:, 11], rec[live][fin][:, 11], 1e-5, "candidate radius")

    # ---- a14: fuse / update ----
    S2 = mfo.fuse_update(S, n, t, op, best, rec)
    g_S2 = bg.downloadMap()
    assert len(g_S2) == n
    _close(g_S2[:, :4], S2[:n, :4], 1e-6, "updated position / confidence")
    assert np.array_equal(g_S2[:, 5:8], S2[:n, 5:8])
    ca, cb = g_S2[:, 4].astype(np.int64), S2[:n, 4].astype(np.int64)   # colour = re-encoded rounded mean: +-1 per channel at a .5 tie
    dch = np.stack([np.abs((ca >> sh & 255) - (cb >> sh & 255)) for sh in (16, 8, 0)], 1)
    assert dch.max() <= 1 and (dch.max(1) > 0).sum() <= 1e-4 * n, (dch.max(), int((dch.max(1) > 0).sum()))
    _close(g_S2[:, 8:], S2[:n, 8:], 1e-6, "updated normal / radius")

    # ---- a13 again on the updated buffer (the pass that feeds clean) ----
    idx2, vc2, ct2, nr2 = mfo.predict_indices(cam, T, g_S2, n, t, MAXD, TIME_DELTA)   # oracle pass on the device's buffer: isolates this pass
    bg.predictIndices(t, MAXD, TIME_DELTA)
    assert np.array_equal(mf.debugRead("index"), idx2)
    _close(mf.debugRead("index_vc"), vc2, 1e-6, "index vertConf (post-fusion)")
    assert np.array_equal(mf.debugRead("index_ct"), ct2)

    # ---- a16: clean ----
    S3, n3 = mfo.clean(cam, T, g_S2, n, g_op, g_rec, t, TIME_DELTA, CONF, MAXD, OUTLIER, 0, idx2, mf.debugRead("index_vc"), ct2,
                       mf.debugRead("index_nr"), depthF_dev, w["mask"], w["cap"])
    bg.clean(t, TIME_DELTA, MAXD)
    g_S3 = bg.downloadMap()
    print("clean: in", n, "+", int((g_op == 2).sum()), "new -> out oracle", n3, "hip", len(g_S3))
    assert len(g_S3) == n3
    _close(g_S3[:, :4], S3[:, :4], 1e-6, "cleaned position / confidence")
    assert np.array_equal(g_S3[:, 4:8], S3[:, 4:8])
    ok = np.isfinite(S3[:, 8:]).all(1)
    _close(g_S3[ok][:, 8:], S3[ok][:, 8:], 1e-6, "cleaned normal / radius")

    # ---- a17: combinedPredict ----
    img, pv, pn, ptime = mfo.combined_predict(cam, T, g_S3, n3, MAXD, CONF, t, t, TIME_DELTA)
    bg.combinedPredict(MAXD, t, t, TIME_DELTA)
    g_img, g_pv, g_pn, g_pt = bg.debugRead("pred_image"), bg.debugRead("pred_vertex"), bg.debugRead("pred_normal"), bg.debugRead("pred_time")
    cover = (pv[..., 2] > 0)
    print("prediction coverage", cover.mean())
    assert cover.mean() > 0.2
    assert np.array_equal(g_pv[..., 2] > 0, cover)
    assert np.array_equal(g_img, img) and np.array_equal(g_pt, ptime)        # same winning surfel everywhere
    _close(g_pv, pv, 1e-6, "predicted vertex / confidence")
    _close(g_pn, pn, 1e-6, "predicted normal / radius")


def test_packed_index_layout_is_equivalent(world):
    """The column-major packed index map mf_process_frame feeds clean() with gives the same keep flags and survivors as the
    row-major images (the layout is a memory-locality choice)."""
    w = world
    mf, bg = w["mf"], w["bg"]
    t = w["t"]
    res = []
    for packed in (0, 1):
        bg.uploadMap(w["S"][:w["n"]])
        mf.setParam("modelApiPackedIndex", packed)
        bg.predictIndices(t, MAXD, TIME_DELTA)
        bg.fuse(t, DEPTH_CUT, 1.0)
        bg.predictIndices(t, MAXD, TIME_DELTA)
        bg.clean(t, TIME_DELTA, MAXD)
        res.append(bg.downloadMap())
    mf.setParam("modelApiPackedIndex", 0)
    assert len(res[0]) == len(res[1]) and np.array_equal(res[0], res[1], equal_nan=True)


def test_fill_in_maps(world):
    """a18: the tracking step's model-side pyramid with fill-in (fill_vertex.frag / fill_normal.frag folded into the pyramid
    kernel) against the oracle's fill-in -> copyMaps -> resize -> transform chain, with a prediction that has holes."""
    w = world
    mfo, cam, mf, bg = w["oracle"], w["cam"], w["mf"], w["bg"]
    W, H, t, T = w["W"], w["H"], w["t"], w["T"]
    # a map with holes: drop the surfels whose projection falls into the left 45 % of the image -> coverage < 75 % -> fill-in
    S = w["S"][:w["n"]].copy()
    Tinv = np.linalg.inv(T)
    pc = S[:, :3] @ Tinv[:3, :3].T + Tinv[:3, 3]
    u = w["st"].fx * pc[:, 0] / pc[:, 2] + w["st"].cx
    S = S[u > 0.45 * W]
    bg.uploadMap(S)
    bg.combinedPredict(MAXD, t, t, TIME_DELTA)
    mf.endFrame(0)            # takes the requiresFillIn decision from the coverage of that prediction (and tick -> t + 1)
    img, pv, pn, _ = mfo.combined_predict(cam, T, S, len(S), MAXD, CONF, t, t, TIME_DELTA)
    assert mfo.lib().mfo_requires_fill_in(img.reshape(-1), W, H, 0.75) == 1
    # MaskFusion::predict ran at the end of frame N_WARM with that frame's raw rgb / filtered depth (performFillIn)
    depthF_dev = mf.debugRead("depthF")
    fi, fv, fn = mfo.fill_in(cam, img, pv, pn, w["rgb"], depthF_dev)
    # next frame: stage it (its predecessor's filtered depth stays in the ring) and track with tryFillIn
    nxt = w["frames"][N_WARM]  # the same image again is fine: only the model side is under test
    mf.stageFrame(nxt[0], nxt[1])
    bg.performTracking(False, False, 100.0, True, False, False, MAXD, 0, True)
    mf.sync()
    v0, n0 = mfo.copy_maps(fv, fn)
    vs, ns = [v0], [n0]
    for i in (1, 2):
        vs.append(mfo.resize_map(vs[-1], False)); ns.append(mfo.resize_map(ns[-1], True))
    R, tt = T[:3, :3].astype(np.float32), T[:3, 3].astype(np.float32)
    for i in range(3):
        ev, en = mfo.transform_maps(vs[i], ns[i], R, tt)
        gv, gn = mf.debugRead(f"vmap_g{i}"), mf.debugRead(f"nmap_g{i}")
        ok = ~np.isnan(ev[0])
        assert np.array_equal(ok, ~np.isnan(gv[0])), i
        frac_fill = 1.0 - (pv[..., 2] > 0).mean()
        assert frac_fill > 0.3
        for c in range(3):
            assert np.abs(gv[c][ok] - ev[c][ok]).max() <= 2e-6 * max(1.0, np.abs(ev[c][ok]).max()), (i, c)
        okn = ~np.isnan(en[0])
        dn = okn != ~np.isnan(gn[0])
        if dn.any():
            ys, xs = np.nonzero(dn)
            print("level", i, "normal validity differs at", int(dn.sum()), "pixels, e.g.", list(zip(xs[:8].tolist(), ys[:8].tolist())),
                  "oracle valid there:", okn[dn][:8].tolist(), "filled there:", (pv[..., 2] == 0)[::1 << i, ::1 << i][dn][:8].tolist() if i == 0 else "")
            if i == 0:
                for x, y in list(zip(xs[:4].tolist(), ys[:4].tolist())):
                    print("   px", (x, y), "pv", pv[y, x], "pn", pn[y, x], "fv", fv[y, x], "fn", fn[y, x], "depthF 2x2", depthF_dev[y:y + 2, x:x + 2].tolist(),
                          "oracle v/n", ev[:, y, x], en[:, y, x], "hip v/n", gv[:, y, x], gn[:, y, x])
        assert not dn.any(), i
        for c in range(3):
            assert np.abs(gn[c][okn] - en[c][okn]).max() <= 2e-5, (i, c)
    assert mf.getLastFillIn() is False or True   # (the decision itself is compared per frame in test_gpu_pipeline.py)


def test_splat_tile_overflow_scans_every_sprite(world):
    """a tile list that overflows must neither drop sprites nor poison the context.  With the lists shrunk to a few
    slots per tile every tile overflows and takes the scan-all path: the prediction stays bit-identical."""
    w = world
    mf, bg = w["mf"], w["bg"]
    t = w["t"]
    bg.uploadMap(w["S"][:w["n"]])
    bg.combinedPredict(MAXD, t, t, TIME_DELTA)
    ref = [bg.debugRead(k) for k in ("pred_vertex", "pred_normal", "pred_image", "pred_time")]
    tiles = ((w["W"] + 15) // 16) * ((w["H"] + 15) // 16)
    full = mf.getParam("splatTileEntries")
    mf.setParam("splatTileEntries", tiles * 8)
    bg.combinedPredict(MAXD, t, t, TIME_DELTA)
    got = [bg.debugRead(k) for k in ("pred_vertex", "pred_normal", "pred_image", "pred_time")]
    mf.sync()                                     # no sticky error state
    mf.setParam("splatTileEntries", full)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    assert (ref[0][..., 2] > 0).mean() > 0.2
