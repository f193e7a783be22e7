"""Kernel-level parity: each HIP entry point of include/maskfusion_amd.h against the CPU oracle (oracle/) on the same
seeded inputs.  Tolerances (fp32 path; the reference itself is nvcc fast-math, so bit equality is undefined):
  images / maps: 2e-6 relative + 1e-6 absolute;  normal equations: 1e-4 relative to the largest |A| entry."""
import numpy as np
import pytest

from gpu_util import dev, empty, host, nan_equal_close, scene_frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    st, fr = scene_frames(3, noise=True)
    return st, fr


def test_bilateral(hip, oracle, frames):
    st, fr = frames
    depth = fr[0][1].copy()
    depth[100:140, 200:260] = 0.0          # hole
    depth[300:304, :] = 0.02               # below the 0.03 gate
    ref = oracle.bilateral(depth)
    d = dev(depth)
    out = empty(depth.shape)
    assert hip.mf_k_bilateral(d.data_ptr(), out.data_ptr(), st.W, st.H, None) == 0
    got = host(out)
    err, bad = nan_equal_close(got, ref, 2e-5, 1e-6)   # __expf vs expf
    print("bilateral max abs err", err)
    assert bad == 0
    assert ((got == 0) == (ref == 0)).all()


def test_bilateral_constant_is_identity(hip):
    depth = np.full((480, 640), 1.5, np.float32)
    out = empty(depth.shape)
    d = dev(depth)   # keep every device tensor alive until the result is read back
    assert hip.mf_k_bilateral(d.data_ptr(), out.data_ptr(), 640, 480, None) == 0
    assert np.allclose(host(out), 1.5, rtol=1e-6)


def test_pyrdown(hip, oracle, frames):
    st, fr = frames
    src = oracle.bilateral(fr[0][1])
    src[50:60, 70:90] = np.nan
    for _ in range(2):
        ref = oracle.pyrdown_f(src)
        out = empty(ref.shape)
        d = dev(src)
        assert hip.mf_k_pyrdown_f(d.data_ptr(), out.data_ptr(), src.shape[1], src.shape[0], None) == 0
        err, bad = nan_equal_close(host(out), ref, 2e-6, 1e-7)
        assert bad == 0
        src = ref


def test_vmap_nmap(hip, oracle, frames):
    st, fr = frames
    depth = oracle.bilateral(fr[0][1])
    for lvl in range(3):
        W, H = st.W >> lvl, st.H >> lvl
        k = [st.fx / (1 << lvl), st.fy / (1 << lvl), st.cx / (1 << lvl), st.cy / (1 << lvl)]
        v_ref = oracle.create_vmap(depth, *k, 3.0)
        n_ref = oracle.create_nmap(v_ref)
        v, n = empty((3, H, W)), empty((3, H, W))
        d = dev(depth)
        assert hip.mf_k_vmap_nmap(d.data_ptr(), v.data_ptr(), n.data_ptr(), W, H, *k, 3.0, None) == 0
        ev, bv = nan_equal_close(host(v), v_ref, 2e-6, 1e-7)
        en, bn = nan_equal_close(host(n), n_ref, 5e-5, 1e-5)   # v_rsq_f32 vs 1/sqrtf on near-degenerate cross products
        print("level", lvl, "vmap err", ev, "nmap err", en)
        assert bv == 0 and bn == 0
        depth = oracle.pyrdown_f(depth)


def _model_maps(oracle, st, depth, T):
    """Model-side maps as the oracle builds them from a (fake) prediction = back-projection of `depth`."""
    H, W = depth.shape
    v = oracle.create_vmap(depth, st.fx, st.fy, st.cx, st.cy, 20.0)
    n = oracle.create_nmap(v)
    v4 = np.zeros((H, W, 4), np.float32)
    n4 = np.zeros((H, W, 4), np.float32)
    valid = ~np.isnan(v[0]) & ~np.isnan(n[0])
    for c in range(3):
        v4[..., c] = np.where(valid, v[c], 0)
        n4[..., c] = np.where(valid, n[c], 0)
    v4[..., 3] = 1
    n4[..., 3] = 0.01
    return v4, n4


def test_model_pyramid(hip, oracle, frames):
    from maskfusion_amd import synth
    st, fr = frames
    v4, n4 = _model_maps(oracle, st, fr[0][1], None)
    T = synth.make_pose(synth.rot_xyz(0.02, -0.03, 0.01), [0.05, -0.02, 0.03])
    R = T[:3, :3].astype(np.float32)
    t = T[:3, 3].astype(np.float32)
    # oracle: copyMaps -> resize x2 -> transform x3
    vs, ns = [None] * 3, [None] * 3
    vs[0], ns[0] = oracle.copy_maps(v4, n4)
    for i in (1, 2):
        vs[i] = oracle.resize_map(vs[i - 1], False)
        ns[i] = oracle.resize_map(ns[i - 1], True)
    ref = [oracle.transform_maps(vs[i], ns[i], R, t) for i in range(3)]
    tot = sum((st.W >> i) * (st.H >> i) * 3 for i in range(3))
    dv, dn = empty(tot), empty(tot)
    Rc = np.ascontiguousarray(R.reshape(9))
    d_v4, d_n4 = dev(v4), dev(n4)
    assert hip.mf_k_model_pyramid(d_v4.data_ptr(), d_n4.data_ptr(), Rc.ctypes.data, t.ctypes.data,
                                  dv.data_ptr(), dn.data_ptr(), st.W, st.H, None) == 0
    gv, gn = host(dv), host(dn)
    off = 0
    for i in range(3):
        sz = (st.W >> i) * (st.H >> i) * 3
        ev, bv = nan_equal_close(gv[off:off + sz].reshape(ref[i][0].shape), ref[i][0], 2e-6, 2e-6)
        en, bn = nan_equal_close(gn[off:off + sz].reshape(ref[i][1].shape), ref[i][1], 2e-5, 2e-6)
        print("level", i, "v err", ev, "n err", en)
        assert bv == 0 and bn == 0
        off += sz


def test_icp_step(hip, oracle, frames):
    from maskfusion_amd import synth
    st, fr = frames
    dF0 = oracle.bilateral(fr[0][1])
    dF1 = oracle.bilateral(fr[1][1])
    for lvl in range(3):
        W, H = st.W >> lvl, st.H >> lvl
        k = [st.fx / (1 << lvl), st.fy / (1 << lvl), st.cx / (1 << lvl), st.cy / (1 << lvl)]
        vc = oracle.create_vmap(dF1, *k, 3.0)
        nc = oracle.create_nmap(vc)
        vp = oracle.create_vmap(dF0, *k, 20.0)
        npv = oracle.create_nmap(vp)
        T = synth.make_pose(synth.rot_xyz(0.004, -0.003, 0.002), [0.003, -0.002, 0.001])
        Rcurr = T[:3, :3].astype(np.float32)
        tcurr = T[:3, 3].astype(np.float32)
        Rpi = np.eye(3, dtype=np.float32)
        tprev = np.zeros(3, np.float32)
        A, b, res = oracle.icp_step(Rcurr, tcurr, vc, nc, Rpi, tprev, *k, vp, npv)
        out = empty(32)
        args = [np.ascontiguousarray(x.reshape(-1)) for x in (Rcurr, tcurr, Rpi, tprev)]
        d_vc, d_nc, d_vp, d_np = dev(vc), dev(nc), dev(vp), dev(npv)
        rc = hip.mf_k_icp_step(args[0].ctypes.data, args[1].ctypes.data, d_vc.data_ptr(), d_nc.data_ptr(),
                               args[2].ctypes.data, args[3].ctypes.data, *k, d_vp.data_ptr(), d_np.data_ptr(),
                               0.10, float(np.sin(np.float32(20.0 * 3.14159254 / 180.0))), W, H, out.data_ptr(), None)
        assert rc == 0
        g = host(out)
        gA = np.zeros((6, 6))
        gb = np.zeros(6)
        s = 0
        for i in range(6):
            for j in range(i, 7):
                if j == 6:
                    gb[i] = g[s]
                else:
                    gA[i, j] = gA[j, i] = g[s]
                s += 1
        print("level", lvl, "inliers gpu/oracle", g[28], res[1], "res", g[27], res[0])
        assert g[28] == res[1], "inlier count must match exactly"
        scale = np.abs(A).max()
        assert np.abs(gA - A).max() <= 1e-4 * scale
        assert np.abs(gb - b).max() <= 1e-4 * max(np.abs(b).max(), 1e-3 * scale)
        # sum r^2: the per-pixel residuals are bit-identical to the oracle's (same operations, each rounded on its own); what is
        # left is fp32 block sums against the oracle's double accumulation
        assert abs(g[27] - res[0]) <= 2e-5 * max(res[0], 1e-9)
        dF0, dF1 = oracle.pyrdown_f(dF0), oracle.pyrdown_f(dF1)


def test_gn_solve_update(hip, oracle):
    """the device-side stand-ins for Eigen's LDLT (RGBDOdometry.cpp:447-459), OdometryProvider::rodrigues /
    computeUpdateSE3 (OdometryProvider.h:32-90) and the pose composition of RGBDOdometry.cpp:461-474, through mf_k_gn_solve:
    the production one-thread LDL^T and the wave-parallel Gauss-Jordan agree with each other, with the oracle's restatement and
    with numpy / SciPy, on real ICP systems (well and badly conditioned) and on rotations from 1e-9 to 0.4 rad."""
    import ctypes as C
    from scipy.spatial.transform import Rotation as Rot
    L = oracle.lib()
    rng = np.random.default_rng(5)

    def pack(A, b, res, inl):
        out = []
        for i in range(6):
            for j in range(i, 7):
                out.append(b[i] if j == 6 else A[i, j])
        return np.array(out + [res, inl], np.float64)

    cases = []
    st, fr = scene_frames(2, noise=True)
    dF0, dF1 = oracle.bilateral(fr[0][1]), oracle.bilateral(fr[1][1])
    K = (st.fx, st.fy, st.cx, st.cy)
    v0 = oracle.create_vmap(dF0, *K, 3.0); n0 = oracle.create_nmap(v0)
    v1 = oracle.create_vmap(dF1, *K, 3.0); n1 = oracle.create_nmap(v1)
    A, b, res = oracle.icp_step(np.eye(3, dtype=np.float32), np.zeros(3, np.float32), v1, n1, np.eye(3, dtype=np.float32),
                                np.zeros(3, np.float32), *K, v0, n0, 0.10, float(np.sin(np.deg2rad(20.0))))
    cases.append((np.asarray(A, np.float64).reshape(6, 6), np.asarray(b, np.float64), float(res[0]), float(res[1])))   # a real frame pair
    for scale in (1e-9, 1e-4, 1e-2, 0.4):                      # synthetic SPD systems whose solution has a rotation of `scale` rad
        M = rng.normal(size=(40, 6))
        A_ = M.T @ M * rng.uniform(1.0, 1e4)
        x_ = np.concatenate([rng.normal(size=3) * 0.01, scale * np.array([0.6, -0.64, 0.48])])
        cases.append((A_, A_ @ x_, 12.5, 1000.0))
    Mi = rng.normal(size=(6, 6))                               # condition number ~1e10 (a corridor-like degenerate scene)
    Ai = Mi @ np.diag([1e6, 1e5, 1e3, 10.0, 1e-2, 1e-4]) @ Mi.T
    cases.append((Ai, Ai @ np.array([0.01, -0.02, 0.005, 1e-3, 2e-3, -1e-3]), 3.0, 500.0))

    Rprev = Rot.from_rotvec([0.2, -0.1, 0.05]).as_matrix().astype(np.float32)
    tprev = np.array([0.3, -0.2, 1.1], np.float32)
    rt0 = np.eye(4)
    rt0[:3, :3] = Rot.from_rotvec([0.01, 0.02, -0.015]).as_matrix(); rt0[:3, 3] = [0.004, -0.002, 0.001]
    for ci, (A_, b_, res_, inl_) in enumerate(cases):
        sys29 = pack(A_, b_, res_, inl_)
        xs, xw, rt = np.zeros(6), np.zeros(6), np.zeros(16)
        Rc, tc, stats = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)
        rt_in = np.ascontiguousarray(rt0.reshape(16))
        rc = hip.mf_k_gn_solve(sys29.ctypes.data, rt_in.ctypes.data, Rprev.ctypes.data, tprev.ctypes.data, xs.ctypes.data, xw.ctypes.data,
                               rt.ctypes.data, Rc.ctypes.data, tc.ctypes.data, stats.ctypes.data, None)
        assert rc == 0
        # (1) the solve: serial LDL^T == wave Gauss-Jordan == oracle (pivoted LDLT restatement) == numpy, relative to |x|
        xo = np.zeros(6)
        assert L.mfo_ldlt_solve(np.ascontiguousarray(A_, np.float64), np.ascontiguousarray(b_, np.float64), xo, 6) == 0
        xn = np.linalg.solve(A_, b_)
        cond = np.linalg.cond(A_)
        tol = max(1e-12, 50 * cond * 2.2e-16) * np.abs(xn).max()
        # the wave solver rounds A, b to fp32 first (as the reference does on the way to the host): its tolerance carries that rounding
        tol_w = max(tol, 50 * cond * 6e-8 * np.abs(xn).max())
        assert np.abs(xs - xn).max() <= tol and np.abs(xs - xo).max() <= tol, (ci, xs, xo, xn)
        assert np.abs(xw - xn).max() <= tol_w, (ci, xw, xn)
        # (2) exp + composition: resultRt <- [exp(w) | t] * resultRt, against the oracle's computeUpdateSE3 and against SciPy
        rto = np.ascontiguousarray(rt0.reshape(16).copy())
        L.mfo_update_se3(rto, np.ascontiguousarray(xs, np.float64))
        T = np.eye(4); T[:3, :3] = Rot.from_rotvec(xs[3:]).as_matrix(); T[:3, 3] = xs[:3]
        want = T @ rt0
        assert np.abs(rt.reshape(4, 4) - want).max() < 1e-14 and np.abs(rt - rto).max() < 1e-14, ci
        # (3) currentT = [Rprev | tprev] * transform^-1 in float (RGBDOdometry.cpp:461-474)
        inc = want.astype(np.float32)
        iR = inc[:3, :3].T
        it = -(iR @ inc[:3, 3])
        assert np.abs(Rc.reshape(3, 3) - Rprev @ iR).max() < 2e-6 and np.abs(tc - (Rprev @ it + tprev)).max() < 2e-6, ci
        assert abs(stats[0] - np.sqrt(np.float32(res_)) / np.float32(inl_)) < 1e-9 and stats[1] == np.float32(inl_)
