"""The oracle's restatement of the reference's tracking loop (mfo_track_icp, mfo_track_rgbd: SURVEY.md rows a8-a12, host half) held to
the reference's own RGBDOdometry::getIncrementalTransformation (Core/Utils/RGBDOdometry.cpp:227-497) compiled from its own text on top
of the reference's own device functions (oracle/build_track.py -> oracle/_ref/libmf_track.so).

Tolerances: the two sides sum the per-pixel rows in different orders (the reference's block / warp tree with the launch shapes of
GPUConfig.h:47-54; the oracle's fixed order) and oracle/eigen_shim's inverse / LDLT agree with Eigen to rounding only, so the normal
equations differ in the last float digits; the poses after 19 Gauss-Newton steps agree to <= 7e-8 in every ICP / joint / SO(3) case
measured (gate 1e-6), 6e-7 in the ill-conditioned rgbOnly branch (gate 5e-6).  Inlier COUNTS are integers and must be equal.
Not compared: lastICPError / lastICPCount when icp is false -- the reference reads its `float residual[2]` uninitialised there
(RGBDOdometry.cpp:421-436) -- and lastSO3Count without so3 (the reference leaves the constructor's width * height).
"""
import numpy as np
import pytest

from maskfusion_amd import synth
from oracle import mfo, mfo_rgbd, mftrack

pytestmark = pytest.mark.skipif(not mftrack.available(), reason="oracle/_ref/libmf_track.so absent and no /root/reference to build it")

SIN20 = float(np.sin(np.float32(20.0 * 3.14159254 / 180.0)))


def _maps(depth, f, cx, cy, cutoff=20.0):
    vs, ns, d = [], [], depth
    for lvl in range(3):
        s = 1 << lvl
        v = mfo.create_vmap(d, f / s, f / s, cx / s, cy / s, cutoff)
        vs.append(v)
        ns.append(mfo.create_nmap(v))
        d = mfo.pyrdown_f(d)
    return vs, ns


def _rgbd_pyramids(vmap0, rgb):
    """populateRGBDData (RGBDOdometry.cpp:187-204): depth from the vertex map (NaN beyond 6 m), gaussian pyramids"""
    v4 = np.concatenate([np.moveaxis(vmap0, 0, -1), np.ones(vmap0.shape[1:] + (1,), np.float32)], -1)
    return mfo_rgbd.f32_pyramid(mfo_rgbd.vertices_to_depth(v4)), mfo_rgbd.u8_pyramid(mfo_rgbd.image_to_intensity(rgb))


def _make_pair(w, h, f):
    """frame 0 = the model (global frame = camera 0), frame 1 = the current frame after a small camera motion"""
    cx, cy = w / 2.0, h / 2.0
    st = synth.Stream(W=w, H=h, fx=f, fy=f, cx=cx, cy=cy)
    T1 = synth.make_pose(synth.rot_xyz(0.006, -0.009, 0.004), [0.012, -0.007, 0.009])
    rgb0, d0, _ = st.scene.render(np.eye(4), 0, w, h, f, f, cx, cy)
    rgb1, d1, _ = st.scene.render(T1, 0, w, h, f, f, cx, cy)
    pv, pn = _maps(d0, f, cx, cy)
    cv, cn = _maps(d1, f, cx, cy)
    ld, li = _rgbd_pyramids(pv[0], rgb0)
    nd, ni = _rgbd_pyramids(cv[0], rgb1)
    return dict(w=w, h=h, f=f, cx=cx, cy=cy, T1=T1, pv=pv, pn=pn, cv=cv, cn=cn, ld=ld, li=li, nd=nd, ni=ni)


@pytest.fixture(scope="module", params=[(160, 120, 132.0), (320, 240, 264.0)], ids=["160x120", "320x240"])
def pair(request):
    return _make_pair(*request.param)


def _both(p, R0, t0, **o):
    """-> ((R, t, inc, stats) of the reference loop, the same of the oracle)"""
    opts = mfo.default_track_opts(pyramid=int(o.get("pyramid", True)), fastOdom=int(o.get("fast_odom", False)), so3=int(o.get("so3", False)),
                                  rgbOnly=int(o.get("rgb_only", False)), icpWeight=o.get("icp_weight", 100.0), distThresh=o.get("dist_thresh", 0.10),
                                  angleThresh=SIN20)
    Rr, tr, ir, sr, A, b = mftrack.track(p["cv"], p["cn"], p["pv"], p["pn"], p["w"], p["h"], p["f"], p["f"], p["cx"], p["cy"], R0, t0,
                                         last_depth=p["ld"], next_depth=p["nd"], last_image=p["li"], next_image=p["ni"],
                                         last_next2=p["li"][2], angle_thresh=SIN20, **o)
    Ro, to, io, so = mfo_rgbd.track_rgbd(p["cv"], p["cn"], p["pv"], p["pn"], p["ld"], p["nd"], p["li"], p["ni"], p["li"][2], p["w"], p["h"],
                                         p["f"], p["f"], p["cx"], p["cy"], R0, t0, opts)
    return (Rr, tr, ir, sr), (Ro, to, io, so)


def _agree(ref, orc, tol, stats=("ICP",)):
    (Rr, tr, ir, sr), (Ro, to, io, so) = ref, orc
    assert np.abs(Rr - Ro).max() < tol and np.abs(tr - to).max() < tol, (np.abs(Rr - Ro).max(), np.abs(tr - to).max())
    assert np.abs(ir - io).max() < tol
    for s in stats:
        assert sr[f"last{s}Count"] == getattr(so, f"last{s}Count"), s
        e_r, e_o = sr[f"last{s}Error"], getattr(so, f"last{s}Error")
        assert abs(e_r - e_o) <= 1e-4 * max(abs(e_r), 1e-12) + 1e-9, (s, e_r, e_o)
    # how often each exit let the loop go round: launches of so3Step / completed Gauss-Newton iterations (TICK counters of the harness)
    assert sr["so3Steps"] == so.so3Iterations
    assert max(sr["icpSteps"], sr["rgbSteps"]) == so.iterationsRun


def test_icp_branch(pair):
    """icpWeight >= 100: icp && !rgb (RGBDOdometry.cpp:237-238), 4 + 5 + 10 iterations"""
    ref, orc = _both(pair, np.eye(3), np.zeros(3), icp_weight=100.0)
    _agree(ref, orc, 1e-6)
    # ... and the ICP-only entry point of the oracle is the same loop
    Ri, ti, inci, err, cnt, _ = mfo.track_icp(pair["cv"], pair["cn"], pair["pv"], pair["pn"], pair["w"], pair["h"], pair["f"], pair["f"],
                                              pair["cx"], pair["cy"], np.eye(3), np.zeros(3))
    assert np.abs(Ri - ref[0]).max() < 1e-6 and np.abs(ti - ref[1]).max() < 1e-6 and cnt == ref[3]["lastICPCount"]
    # both recover the synthetic motion
    assert np.linalg.norm(ref[1] - pair["T1"][:3, 3]) < 3e-3


def test_icp_branch_from_a_nonidentity_pose(pair):
    """Rprev / tprev enter through Rprev_inv, device_tprev and currentT = T_prev * transform^-1 (RGBDOdometry.cpp:331-334, 479-486)"""
    T0 = synth.make_pose(synth.rot_xyz(0.3, -0.2, 0.1), [0.4, -0.1, 0.25])
    pv, pn = [], []
    for v, n in zip(pair["pv"], pair["pn"]):
        g, gn = mfo.transform_maps(v, n, T0[:3, :3].astype(np.float32), T0[:3, 3].astype(np.float32))
        pv.append(g)
        pn.append(gn)
    p = dict(pair, pv=pv, pn=pn)
    ref, orc = _both(p, T0[:3, :3], T0[:3, 3], icp_weight=100.0)
    _agree(ref, orc, 2e-6)
    assert np.linalg.norm(ref[1] - (T0 @ pair["T1"])[:3, 3]) < 3e-3


@pytest.mark.parametrize("fast_odom,pyramid", [(True, True), (False, False), (True, False)], ids=["fast", "nopyr", "fast-nopyr"])
def test_iteration_schedules(pair, fast_odom, pyramid):
    """iterations = {fastOdom ? 3 : 10, pyramid ? 5 : 0, pyramid ? 4 : 0} (RGBDOdometry.cpp:327-329)"""
    ref, orc = _both(pair, np.eye(3), np.zeros(3), icp_weight=100.0, fast_odom=fast_odom, pyramid=pyramid)
    _agree(ref, orc, 1e-6)


def test_joint_icp_rgb(pair):
    """icp && rgb: lastA = A_rgbd + w^2 A_icp, lastb = b_rgbd + w b_icp (RGBDOdometry.cpp:452-456), w = 10 (the core default)"""
    ref, orc = _both(pair, np.eye(3), np.zeros(3), icp_weight=10.0)
    _agree(ref, orc, 1e-6, stats=("ICP", "RGB"))


def test_joint_with_so3_prealignment(pair):
    """so3: ten-iteration SO(3) photometric loop at level 2 seeds resultRt (RGBDOdometry.cpp:253-324, 338-344)"""
    ref, orc = _both(pair, np.eye(3), np.zeros(3), icp_weight=10.0, so3=True)
    _agree(ref, orc, 1e-6, stats=("ICP", "RGB", "SO3"))


def test_joint_with_so3_prealignment_vga():
    """the GUI configuration (icpWeight 20, so3) at the frame size of BASELINE.json's configs"""
    p = _make_pair(640, 480, 528.0)
    ref, orc = _both(p, np.eye(3), np.zeros(3), icp_weight=20.0, so3=True)
    _agree(ref, orc, 1e-6, stats=("ICP", "RGB", "SO3"))


def test_rgb_only(pair):
    """rgbOnly: no ICP term, sigma = -1, early exit when the photometric error grows (RGBDOdometry.cpp:393-404)"""
    ref, orc = _both(pair, np.eye(3), np.zeros(3), rgb_only=True, icp_weight=10.0)
    _agree(ref, orc, 5e-6, stats=("RGB",))


def test_large_jump_is_rejected(pair):
    """rgb && |tcurr - tprev| > 0.3 m: the pose is left where it was and the increment is identity (RGBDOdometry.cpp:470-474).  The
    camera moved 0.35 m towards the scene and the association gate is opened to 1 m so that ICP does follow it; without the photometric
    term (icpWeight >= 100, rgb false) the rule does not apply and the motion is returned."""
    w, h, f, cx, cy = (pair[k] for k in ("w", "h", "f", "cx", "cy"))
    st = synth.Stream(W=w, H=h, fx=f, fy=f, cx=cx, cy=cy)
    T1 = synth.make_pose(np.eye(3), [0.0, 0.0, 0.35])
    rgb1, d1, _ = st.scene.render(T1, 0, w, h, f, f, cx, cy)
    cv, cn = _maps(d1, f, cx, cy)
    nd, ni = _rgbd_pyramids(cv[0], rgb1)
    p = dict(pair, cv=cv, cn=cn, nd=nd, ni=ni)
    ref, orc = _both(p, np.eye(3), np.zeros(3), icp_weight=10.0, dist_thresh=1.0)
    for R, t, inc, _ in (ref, orc):
        assert np.array_equal(R, np.eye(3, dtype=np.float32)) and np.array_equal(t, np.zeros(3, np.float32))
        assert np.array_equal(np.asarray(inc, np.float32), np.eye(4, dtype=np.float32))
    ref, orc = _both(p, np.eye(3), np.zeros(3), icp_weight=100.0, dist_thresh=1.0)
    _agree(ref, orc, 2e-6)
    assert abs(ref[1][2] - 0.35) < 5e-3
