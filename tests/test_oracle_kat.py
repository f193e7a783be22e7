"""Known-answer tests that pin the CPU oracle (oracle/mf_oracle.c).

The reference ships no tests / golden vectors and cannot be built here, so the oracle is pinned
by closed-form answers that follow from the reference's formulas (cited per test), not by reference outputs.
All tests run on CPU in seconds.
"""
import ctypes as C
import numpy as np
import pytest

from maskfusion_amd import synth

W, H = 160, 120
FX = FY = 132.0
CX, CY = 80.0, 60.0
SIN20 = float(np.sin(np.float32(20.0 * 3.14159254 / 180.0)))


# ------------------------------------------------------------------------------------------------ math
def test_rodrigues_matches_scipy(oracle):
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(0)
    for w in [np.zeros(3), [1e-9, 0, 0], [0.01, -0.02, 0.03], [1.0, 2.0, -0.5]] + list(rng.randn(20, 3)):
        w = np.asarray(w, np.float64)
        R = np.zeros(9)
        oracle.lib().mfo_rodrigues(np.ascontiguousarray(w), R)
        assert np.allclose(R.reshape(3, 3), Rotation.from_rotvec(w).as_matrix(), atol=1e-12)


def test_update_se3_left_multiplies(oracle):
    # OdometryProvider::computeUpdateSE3 (Core/Utils/OdometryProvider.h:69-90): resultRt <- [exp(w)|t] * resultRt
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(1)
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec([0.1, 0.2, -0.1]).as_matrix()
    T[:3, 3] = [0.3, -0.2, 0.1]
    x = rng.randn(6) * 0.05
    D = np.eye(4)
    D[:3, :3] = Rotation.from_rotvec(x[3:]).as_matrix()
    D[:3, 3] = x[:3]
    got = np.ascontiguousarray(T.reshape(16).copy())
    oracle.lib().mfo_update_se3(got, np.ascontiguousarray(x))
    assert np.allclose(got.reshape(4, 4), D @ T, atol=1e-13)


def test_ldlt_solve(oracle):
    rng = np.random.RandomState(2)
    for n in (3, 6):
        for _ in range(50):
            M = rng.randn(n, n + 3)
            A = M @ M.T
            b = rng.randn(n)
            x = np.zeros(n)
            assert oracle.lib().mfo_ldlt_solve(np.ascontiguousarray(A.reshape(-1)), b, x, n) == 0
            assert np.allclose(A @ x, b, atol=1e-9)
    # singular (all-zero) system -> zero update, like Eigen::LDLT::solve (no inf / NaN)
    x = np.ones(6)
    oracle.lib().mfo_ldlt_solve(np.zeros(36), np.zeros(6), x, 6)
    assert np.array_equal(x, np.zeros(6))


def test_colour_encoding_round_trip(oracle):
    # color_encoding.glsl:19-34
    L = oracle.lib()
    rgb = np.zeros(3, np.float32)
    rng = np.random.RandomState(3)
    cols = [(0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255), (1, 2, 3)] + [tuple(c) for c in rng.randint(0, 256, (500, 3))]
    for r, g, b in cols:
        enc = L.mfo_encode_color(r / 255.0, g / 255.0, b / 255.0)
        assert enc == float((int(r) << 16) + (int(g) << 8) + int(b))
        L.mfo_decode_color(enc, rgb)
        assert np.allclose(rgb * 255.0, [r, g, b], atol=1e-4)


def test_radius_and_confidence_spot_values(oracle):
    L = oracle.lib()
    # surfels.glsl:19-34: radius = min(2 r0, r0/|nz|), r0 = z*sqrt(2)/((fx+fy)/2)
    r0 = 2.0 * np.sqrt(2.0) / 528.0
    assert np.isclose(L.mfo_get_radius(2.0, 1.0, 528.0, 528.0), r0, rtol=1e-6)
    assert np.isclose(L.mfo_get_radius(2.0, 0.8, 528.0, 528.0), r0 / 0.8, rtol=1e-6)
    assert np.isclose(L.mfo_get_radius(2.0, 0.1, 528.0, 528.0), 2 * r0, rtol=1e-6)       # capped at 2x
    assert np.isclose(L.mfo_get_radius(2.0, -0.8, 500.0, 556.0), r0 / 0.8, rtol=1e-6)    # |nz|, mean focal
    # surfels.glsl:36-46: exp(-(d/400)^2/0.72) * w   (quirk Q6: 400 px hard-coded)
    assert np.isclose(L.mfo_confidence(320.0, 240.0, 1.0, 320.0, 240.0), 1.0)
    assert np.isclose(L.mfo_confidence(320.0 + 400.0, 240.0, 2.0, 320.0, 240.0), 2.0 * np.exp(-1 / 0.72), rtol=1e-6)


# ------------------------------------------------------------------------------------------------ images
def test_bilateral_properties(oracle):
    # depth_bilateral_metric.frag:30-76
    d = np.full((H, W), 1.7, np.float32)
    assert np.allclose(oracle.bilateral(d), 1.7, rtol=1e-6)          # constant image is a fixed point
    d2 = d.copy()
    d2[10:20, 10:20] = 0.02                                           # <= 0.03 is zeroed
    d2[40, 40] = 0.0
    out = oracle.bilateral(d2)
    assert (out[10:20, 10:20] == 0).all() and out[40, 40] == 0
    assert np.allclose(out[80:, 100:], 1.7, rtol=1e-6)
    # a 1 m step edge is preserved: weight exp(-555.556) == 0 in float
    d3 = d.copy()
    d3[:, W // 2:] = 2.7
    out = oracle.bilateral(d3)
    assert np.allclose(out[:, :W // 2], 1.7, rtol=1e-6) and np.allclose(out[:, W // 2:], 2.7, rtol=1e-6)


def _pyrdown_ref(src):
    """Independent numpy restatement of pyrDownKernelGaussF incl. the border quirk (cudafuncs.cu:345-359)."""
    k = np.array([1, 4, 6, 4, 1], np.float64)
    G = np.outer(k, k)
    sh, sw = src.shape
    dst = np.zeros((sh // 2, sw // 2))
    for y in range(sh // 2):
        for x in range(sw // 2):
            tx, ty = min(2 * x + 3, sw - 1), min(2 * y + 3, sh - 1)
            s = c = 0.0
            for cy in range(max(0, 2 * y - 2), ty):
                for cx in range(max(0, 2 * x - 2), tx):
                    v = src[cy, cx]
                    if not np.isnan(v):
                        w = G[ty - cy - 1, tx - cx - 1]
                        s += v * w
                        c += w
            dst[y, x] = s / c if c else np.nan
    return dst


def test_pyrdown_matches_independent_restatement(oracle):
    rng = np.random.RandomState(4)
    src = (1.0 + rng.rand(24, 32)).astype(np.float32)
    src[5:8, 7:12] = np.nan
    src[0:6, 0:6] = np.nan                                             # a fully-NaN footprint -> NaN out
    got = oracle.pyrdown_f(src)
    ref = _pyrdown_ref(src)
    assert (np.isnan(got) == np.isnan(ref)).all()
    ok = ~np.isnan(ref)
    assert np.allclose(got[ok], ref[ok], rtol=1e-6)
    const = np.full((24, 32), 2.5, np.float32)
    assert np.allclose(oracle.pyrdown_f(const), 2.5, rtol=1e-6)


def test_vmap_nmap_of_a_plane(oracle):
    # createVMap / createNMap (cudafuncs.cu:109-189): a fronto-parallel plane has normal (0,0,+1)
    d = np.full((H, W), 2.0, np.float32)
    d[30, 40] = 0.0
    d[31, 50] = 3.5            # beyond the cut-off
    v = oracle.create_vmap(d, FX, FY, CX, CY, 3.0)
    n = oracle.create_nmap(v)
    u, vv = np.meshgrid(np.arange(W), np.arange(H))
    ok = ~np.isnan(v[0])
    assert not ok[30, 40] and not ok[31, 50] and ok.sum() == W * H - 2
    assert np.allclose(v[0][ok], (2.0 * (u - CX) / FX)[ok], rtol=1e-6, atol=1e-7)
    assert np.allclose(v[1][ok], (2.0 * (vv - CY) / FY)[ok], rtol=1e-6, atol=1e-7)
    assert np.isnan(n[0][:, -1]).all() and np.isnan(n[0][-1, :]).all()     # last row / column
    assert np.isnan(n[0][30, 39]) and np.isnan(n[0][29, 40])                # neighbours of an invalid vertex
    okn = ~np.isnan(n[0])
    assert np.allclose(n[0][okn], 0, atol=1e-6) and np.allclose(n[1][okn], 0, atol=1e-6) and np.allclose(n[2][okn], 1, atol=1e-6)


# ------------------------------------------------------------------------------------------------ ICP
def _plane_maps(oracle, z):
    d = np.full((H, W), z, np.float32)
    v = oracle.create_vmap(d, FX, FY, CX, CY, 20.0)
    return v, oracle.create_nmap(v)


def test_icp_step_closed_form_on_a_plane(oracle):
    """Identical fronto-parallel planes, identity poses: every pixel with a valid normal is an inlier with r = 0 and
    J = [n, s x n] with n = (0,0,1), s = vertex (reduce.cu:355-415), so A has the closed form below and b = 0."""
    v, n = _plane_maps(oracle, 2.0)
    I3 = np.eye(3, dtype=np.float32)
    z3 = np.zeros(3, np.float32)
    A, b, res = oracle.icp_step(I3, z3, v, n, I3, z3, FX, FY, CX, CY, v, n, 0.10, SIN20)
    ok = ~np.isnan(n[0])
    x, y = v[0][ok].astype(np.float64), v[1][ok].astype(np.float64)
    N = ok.sum()
    J = np.stack([np.zeros(N), np.zeros(N), np.ones(N), y, -x, np.zeros(N)], 1)    # s x n = (y, -x, 0)
    A_ref = J.T @ J
    assert res[1] == N and res[0] == 0
    assert np.allclose(A, A_ref, rtol=1e-5, atol=1e-6 * np.abs(A_ref).max())
    assert np.allclose(b, 0, atol=1e-9)


def test_icp_step_plane_offset_gives_pure_z_gradient(oracle):
    """Current plane 1 cm in front of the model plane: r = n.(s - d) = -0.01 for every inlier, b = J^T r."""
    vc, nc = _plane_maps(oracle, 1.99)
    vp, npv = _plane_maps(oracle, 2.00)
    I3 = np.eye(3, dtype=np.float32)
    z3 = np.zeros(3, np.float32)
    A, b, res = oracle.icp_step(I3, z3, vc, nc, I3, z3, FX, FY, CX, CY, vp, npv, 0.10, SIN20)
    n_in = res[1]
    assert n_in > 0.95 * W * H
    assert np.isclose(res[0], n_in * 1e-4, rtol=1e-3)            # sum r^2
    assert np.isclose(b[2], -0.01 * n_in, rtol=1e-3)             # J^T r on the z row
    # gates: a 20 cm offset exceeds distThres = 0.10 -> no inliers (reduce.cu:352)
    vc2, nc2 = _plane_maps(oracle, 1.80)
    _, _, res2 = oracle.icp_step(I3, z3, vc2, nc2, I3, z3, FX, FY, CX, CY, vp, npv, 0.10, SIN20)
    assert res2[1] == 0


def _maps_from_depth(oracle, depth, cutoff):
    lv_v, lv_n = [], []
    d = depth
    for lvl in range(3):
        s = 1 << lvl
        v = oracle.create_vmap(d, FX / s, FY / s, CX / s, CY / s, cutoff)
        lv_v.append(v)
        lv_n.append(oracle.create_nmap(v))
        d = oracle.pyrdown_f(d)
    return lv_v, lv_n


def _maps_from_depth_k(oracle, depth, cutoff, fx, fy, cx, cy):
    lv_v, lv_n = [], []
    d = depth
    for lvl in range(3):
        s = 1 << lvl
        v = oracle.create_vmap(d, fx / s, fy / s, cx / s, cy / s, cutoff)
        lv_v.append(v)
        lv_n.append(oracle.create_nmap(v))
        d = oracle.pyrdown_f(d)
    return lv_v, lv_n


@pytest.mark.parametrize("w,h,f,tol", [(160, 120, 132.0, 3e-3), (640, 480, 528.0, 2e-4)])
def test_track_icp_recovers_known_motion(oracle, w, h, f, tol):
    """RGBDOdometry::getIncrementalTransformation, ICP branch (RGBDOdometry.cpp:227-497): frame 0 is the model (global
    frame = camera 0), frame 1 is the current frame; the recovered pose must be the synthetic camera motion.  Projective
    point-to-plane ICP with forward-difference normals is accurate to a fraction of a pixel footprint (~15 mm at
    160x120, ~4 mm at 640x480 for this scene): 0.04 mm at VGA, ~1.7 mm at 160x120."""
    cx, cy = w / 2.0, h / 2.0
    st = synth.Stream(W=w, H=h, fx=f, fy=f, cx=cx, cy=cy)
    T1 = synth.make_pose(synth.rot_xyz(0.006, -0.009, 0.004), [0.012, -0.007, 0.009])
    _, d0, _ = st.scene.render(np.eye(4), 0, w, h, f, f, cx, cy)
    _, d1, _ = st.scene.render(T1, 0, w, h, f, f, cx, cy)
    pv, pn = _maps_from_depth_k(oracle, d0, 20.0, f, f, cx, cy)
    cv, cn = _maps_from_depth_k(oracle, d1, 20.0, f, f, cx, cy)
    R, t, inc, err, cnt, log = oracle.track_icp(cv, cn, pv, pn, w, h, f, f, cx, cy, np.eye(3), np.zeros(3), want_log=True)
    assert log.n_iters == 19                                        # 4 + 5 + 10 (RGBDOdometry.cpp:327-329)
    assert np.linalg.norm(t - T1[:3, 3]) < tol
    assert np.abs(R - T1[:3, :3]).max() < tol
    assert cnt > 0.5 * w * h and err < 1e-4
    # the residual shrinks monotonically over the last level's iterations
    r = np.array([log.residual[i][0] for i in range(9, 19)])
    assert r[-1] <= r[0]
    # the returned increment is resultRt (camera motion increment): T_curr = T_prev * inc^-1 with T_prev = I
    assert np.allclose(np.linalg.inv(inc)[:3, 3], t, atol=1e-5)


# ------------------------------------------------------------------------------------------------ surfels
@pytest.fixture(scope="module")
def frame0():
    st = synth.Stream(W=W, H=H, fx=FX, fy=FY, cx=CX, cy=CY)
    rgb, depth, _ = st.frame(0)
    return st, rgb, depth


def test_init_surfels_column_major_and_values(oracle, frame0):
    st, rgb, depth = frame0
    depth = depth.copy()
    depth[20:25, 30:33] = 0
    dF = oracle.bilateral(depth)
    cam = oracle.cam(W, H, FX, FY, CX, CY)
    surf = np.zeros((W * H, 12), np.float32)
    import ctypes as C
    n = oracle.lib().mfo_init_surfels(C.byref(cam), rgb, depth, dF, 1, 20.0, surf.reshape(-1), W * H)
    assert n == (depth > 0).sum()
    # column-major (FeedbackBuffer.cpp:44-50): first surfel is pixel (0,0), the second (0,1)
    xs, ys = np.nonzero((depth > 0).T)
    for k in (0, 1, n // 2, n - 1):
        i, j = xs[k], ys[k]
        z = depth[j, i]
        assert np.allclose(surf[k, :3], [(i + 0.5 - CX) * z / FX, (j + 0.5 - CY) * z / FY, z], rtol=1e-5)
        assert surf[k, 4] == (int(rgb[j, i, 0]) << 16) + (int(rgb[j, i, 1]) << 8) + int(rgb[j, i, 2])
        assert surf[k, 6] == 1 and surf[k, 7] == 1
        r = np.hypot(i + 0.5 - CX, j + 0.5 - CY) / 400.0
        assert np.isclose(surf[k, 3], np.exp(-r * r / 0.72), rtol=1e-5)


def test_surfel_cycle_on_a_static_frame(oracle, frame0):
    """Feeding the same frame twice from the identity pose: almost every candidate merges (update), hardly any new
    surfel appears, confidences go up, and the splat prediction reproduces the input depth."""
    st, rgb, depth = frame0
    o = oracle.Oracle(W, H, FX, FY, CX, CY, icpWeight=100.0, capacity=W * H * 2, confGlobal=1.5, so3=0)
    o.process_frame(rgb, depth)
    n0 = o.count
    c0 = o.surfels()[:, 3].copy()
    o.process_frame(rgb, depth)
    s1 = o.surfels()
    assert np.abs(o.pose - np.eye(4)).max() < 1e-4                 # no motion
    assert abs(o.count - n0) < 0.03 * n0
    # quarter-rate merge (data.vert:117): about a quarter of the surfels gained confidence, none lost any
    gained = (s1[:n0, 3] > c0[:len(s1[:n0])] + 1e-6).mean()
    assert 0.15 < gained < 0.35
    assert (s1[:n0, 3] >= c0[:n0] - 1e-6).all()
    # a third pass makes the first surfels stable (conf > 1.5) and the prediction matches the input depth
    o.process_frame(rgb, depth)
    pv = o.dbg("pred_vertex")
    seen = pv[..., 2] > 0
    assert seen.mean() > 0.3
    # (3 cm discs at this resolution: flat discs on slanted walls and silhouette overhang bound the tail)
    err = np.abs(pv[..., 2][seen] - depth[seen])
    assert np.median(err) < 1e-3 and np.percentile(err, 90) < 0.02
    o.close()


def test_predict_indices_points_land_in_their_pixel(oracle, frame0):
    st, rgb, depth = frame0
    import ctypes as C
    cam = oracle.cam(W, H, FX, FY, CX, CY)
    dF = oracle.bilateral(depth)
    surf = np.zeros((W * H, 12), np.float32)
    n = oracle.lib().mfo_init_surfels(C.byref(cam), rgb, depth, dF, 1, 20.0, surf.reshape(-1), W * H)
    idx = np.zeros((H, W), np.int32)
    vc, ct, nr = (np.zeros((H, W, 4), np.float32) for _ in range(3))
    pose = oracle.pose16(np.eye(4))
    oracle.lib().mfo_predict_indices(C.byref(cam), pose, surf.reshape(-1), n, 2, 20.0, 200, idx, vc.reshape(-1),
                                     ct.reshape(-1), nr.reshape(-1))
    ys, xs = np.nonzero(idx)
    assert len(ys) > 0.9 * n
    s = surf[idx[ys, xs]]
    u = FX * s[:, 0] / s[:, 2] + CX
    v = FY * s[:, 1] / s[:, 2] + CY
    assert (np.floor(u) == xs).all() and (np.floor(v) == ys).all()
    assert np.allclose(vc[ys, xs, :3], s[:, :3], atol=1e-6)       # identity pose: camera frame == model frame


def test_fusion_weight(oracle):
    # Model::computeFusionWeight (Model.cpp:449-464): max(1 - min(max(|t|, |w|), 0.01)/0.01, 0.5) * multiplier
    I = oracle.pose16(np.eye(4))
    L = oracle.lib()
    assert np.isclose(L.mfo_fusion_weight(I, I, 1.0), 1.0)
    T = synth.make_pose(np.eye(3), [0.004, 0, 0])
    assert np.isclose(L.mfo_fusion_weight(oracle.pose16(T), I, 1.0), 0.6, atol=1e-4)
    T = synth.make_pose(synth.rot_xyz(0.002, 0, 0), [0, 0, 0])
    # the closed form holds for the ACCURATE log map; the default since round 3 is the reference's own arithmetic (finding F5: cos(theta) from
    # a float trace quantises theta in steps of ~4.9e-4 rad), which lands within one quantum (0.049 of the weight, x the multiplier) of it
    L.mfo_set_weight_literal(0)
    try:
        assert np.isclose(L.mfo_fusion_weight(oracle.pose16(T), I, 2.0), 1.6, atol=1e-3)
    finally:
        L.mfo_set_weight_literal(1)
    assert abs(L.mfo_fusion_weight(oracle.pose16(T), I, 2.0) - 1.6) <= 2.0 * 0.05
    T = synth.make_pose(np.eye(3), [0.5, 0, 0])
    assert np.isclose(L.mfo_fusion_weight(oracle.pose16(T), I, 1.0), 0.5)


def test_pipeline_tracks_ground_truth(oracle):
    """processFrame on the noise-free stream: the estimated trajectory follows the synthetic ground truth."""
    st = synth.Stream(W=W, H=H, fx=FX, fy=FY, cx=CX, cy=CY)
    o = oracle.Oracle(W, H, FX, FY, CX, CY, icpWeight=100.0, capacity=W * H * 3, so3=0)
    est, gt = [], []
    for k in range(12):
        rgb, depth, _ = st.frame(k)
        o.process_frame(rgb, depth)
        est.append(o.pose)
        gt.append(st.gt_pose(k))
    assert synth.ate_rmse(np.array(est), np.array(gt)) < 2e-3
    assert o.tick == 13 and o.count > W * H
    o.close()


def test_shader_exp_and_acos_are_accurate(oracle):
    """The shared deterministic exp / acos of the surfel shaders (surfels.glsl:44, data.vert:167) stay within 2 ulp of the
    correctly rounded functions over the ranges the shaders use."""
    L = oracle.lib()
    for f in (L.mfo_shader_exp, L.mfo_shader_acos):
        f.restype = C.c_float
        f.argtypes = [C.c_float]
    rs = np.random.RandomState(3)
    xs = -(rs.rand(20000).astype(np.float32) * 30)
    e = np.array([L.mfo_shader_exp(float(x)) for x in xs], np.float32)
    ref = np.exp(xs.astype(np.float64))
    assert (np.abs(e - ref) / np.spacing(ref.astype(np.float32))).max() < 2.0
    cs = rs.rand(20000).astype(np.float32) * 2 - 1
    a = np.array([L.mfo_shader_acos(float(c)) for c in cs], np.float32)
    ref = np.arccos(cs.astype(np.float64))
    assert (np.abs(a - ref) / np.spacing(ref.astype(np.float32))).max() < 2.0
    assert L.mfo_shader_exp(0.0) == 1.0 and L.mfo_shader_acos(1.0) == 0.0
    assert np.isnan(L.mfo_shader_acos(float("nan"))) and np.isnan(L.mfo_shader_acos(1.5))
