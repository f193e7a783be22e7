"""The device's own scalar math -- maskfusion_amd/csrc/mf_device.h and the solve / update / window-walk functions of mf_odometry.hip and
mf_surfel.hip -- compiled for the host (tests/devmath.py) and held to the oracle, numpy and SciPy WITHOUT a GPU.  SURVEY.md rows a11
(LDL^T, rodrigues, computeUpdateSE3, pose composition), a15 (computeFusionWeight), a13-a16 shader helpers (exp, acos, radius,
confidence, colour code), a16 (the clean window walk), a22 (quaternion of the pose log).  The same source runs on the device; what the
GPU tests add on top is the device's own arithmetic (v_rcp_f64 seeds, contraction) -- tests/test_gpu_kernels.py::test_gn_solve_update."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import devmath  # noqa: E402
from oracle import mfo  # noqa: E402


@pytest.fixture(scope="module")
def dm():
    return devmath.lib()


@pytest.fixture(scope="module")
def orc():
    L = mfo.lib()
    for n in ("mfo_shader_exp", "mfo_shader_acos"):
        getattr(L, n).argtypes = [C.c_float]; getattr(L, n).restype = C.c_float
    L.mfo_get_radius.argtypes = [C.c_float] * 4; L.mfo_get_radius.restype = C.c_float
    L.mfo_confidence.argtypes = [C.c_float] * 5; L.mfo_confidence.restype = C.c_float
    return L


def _bits(x):
    return np.float32(x).view(np.uint32)


def test_shader_exp_acos_bit_identical_to_oracle(dm, orc):
    """exp() of surfels.glsl:44 and acos() of data.vert:167 are left to the vendor by GLSL; the device and the oracle evaluate ONE shared
    fp32 polynomial each -- the same bits, and within 2 ulp of libm"""
    xs = np.concatenate([np.linspace(-40.0, 3.0, 20001), -np.logspace(-9, 1.5, 2000), [0.0, -0.0, -87.0, -104.0]]).astype(np.float32)
    for x in xs:
        a, b = dm.dm_shader_exp(float(x)), orc.mfo_shader_exp(float(x))
        assert _bits(a) == _bits(b), x
    near = xs[(xs > -80) & (xs < 3)]
    got = np.array([dm.dm_shader_exp(float(x)) for x in near], np.float32)
    ref = np.exp(near.astype(np.float64))
    assert (np.abs(got - ref) <= 2.0 * np.spacing(ref.astype(np.float32))).all()
    cs = np.concatenate([np.linspace(-1.0, 1.0, 20001), 1.0 - np.logspace(-8, -1, 500), [1.0, -1.0, 1.0000001, -1.0000001]]).astype(np.float32)
    for c in cs:
        a, b = dm.dm_shader_acos(float(c)), orc.mfo_shader_acos(float(c))
        assert _bits(a) == _bits(b) or (np.isnan(a) and np.isnan(b)), c
    inside = cs[np.abs(cs) <= 1]
    got = np.array([dm.dm_shader_acos(float(c)) for c in inside], np.float64)
    assert np.abs(got - np.arccos(inside.astype(np.float64))).max() < 4e-7


def test_surfel_helpers_bit_identical_to_oracle(dm, orc):
    """getRadius / confidence (surfels.glsl:19-46), encodeColor / decodeColor (color_encoding.glsl:19-34)"""
    rng = np.random.default_rng(3)
    fx, fy, cx, cy = 528.0, 531.5, 320.0, 240.0
    for _ in range(4000):
        d, nz = float(rng.uniform(0.2, 6.0)), float(rng.uniform(-1.0, 1.0))
        assert _bits(dm.dm_surfel_radius(d, nz, fx, fy, cx, cy)) == _bits(orc.mfo_get_radius(d, nz, fx, fy))
        x, y, w = float(rng.uniform(0, 640)), float(rng.uniform(0, 480)), float(rng.uniform(0.5, 100.0))
        assert _bits(dm.dm_surfel_confidence(x, y, w, fx, fy, cx, cy)) == _bits(orc.mfo_confidence(x, y, w, cx, cy))
        r8 = rng.integers(0, 256, 3)
        r, g, b = (float(np.float32(v) / np.float32(255.0)) for v in r8)          # the shaders code colours in [0, 1]
        e = dm.dm_encode_color(r, g, b)
        assert _bits(e) == _bits(orc.mfo_encode_color(r, g, b))
        o1, o2 = np.zeros(3, np.float32), np.zeros(3, np.float32)
        dm.dm_decode_color(e, o1); orc.mfo_decode_color(e, o2)
        assert np.array_equal(o1, o2) and np.array_equal(np.rint(o1 * 255.0).astype(int), r8)


def _taps_shader_text(c, size):
    """copy_unstable.vert:85-86 along one axis, in float32: for (i = c - 2s; i < c + 2s; i += s), texel = floor after a 1/256 snap"""
    f = np.float32
    fs = f(size)
    step = f(f(f(1.0) / f(fs * f(1.0))) * f(0.5))
    half = f(f(f(1.0) * step) * f(2.0))
    end = f(c + half)
    i = f(c - half)
    out = []
    while i < end and len(out) < 8:
        t = int(np.floor(f(np.rint(f(f(i * fs) * f(256.0))) * f(1.0 / 256.0))))
        out.append(min(max(t, 0), size - 1))
        i = f(i + step)
    return out


@pytest.mark.parametrize("size", [640, 480, 1280, 160, 333])
def test_clean_window_walk_is_the_shader_loop(dm, size):
    """finding F1: the fp32 induction variable makes 4 or 5 taps; the device's three slots with multiplicities are exactly those taps"""
    rng = np.random.default_rng(size)
    xs = np.concatenate([rng.uniform(0.0, size, 3000), np.arange(0, size, 7) + 0.5, [0.01, size - 0.01, 0.5, 1.0, size - 1.0]]).astype(np.float32)
    five = 0
    for x in xs:
        c = np.float32(x / np.float32(size))
        u, m = np.zeros(3, np.int32), np.zeros(3, np.int32)
        dm.dm_window_slots_literal(float(c), size, u, m)
        taps = _taps_shader_text(c, size)
        got = [int(u[k]) for k in range(3) for _ in range(int(m[k]))]
        assert got == taps, (x, got, taps)
        assert len(taps) in (4, 5)
        five += len(taps) == 5
    assert 0 < five < len(xs)          # both trip counts occur


def test_rodrigues_exp_and_log(dm):
    """OdometryProvider::rodrigues (OdometryProvider.h:32-66) as the device evaluates it (even power series up to 0.5 rad, sin / cos
    beyond) against SciPy, 1e-12 .. 3 rad; Model::rodrigues2 (Model.cpp:891-932, the log map behind the fusion weight) back again"""
    rng = np.random.default_rng(1)
    for th in np.concatenate([np.logspace(-12, np.log10(3.0), 120), [0.0, 0.4999, 0.5, 0.5001, 0.7071]]):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        w = np.ascontiguousarray(ax * th)
        R = np.zeros(9)
        dm.dm_rodrigues(w, R)
        want = Rot.from_rotvec(w).as_matrix()
        assert np.abs(R.reshape(3, 3) - want).max() < 4e-16 + 1e-15 * th, th
        if 1e-4 < th < 3.0:
            r = np.zeros(3)
            dm.dm_rodrigues2(np.ascontiguousarray(want.reshape(9), np.float32), r)
            assert np.abs(r - w).max() < 2e-6 * max(1.0, 1.0 / th) , th


def test_quaternion_of_the_pose_log(dm):
    """Eigen::Quaternionf(R) (Quaternion.h, the branch on the trace) -> x y z w of poses-<id>.txt"""
    rng = np.random.default_rng(2)
    for k in range(300):
        rv = rng.normal(size=3); rv *= rng.uniform(0.0, np.pi) / np.linalg.norm(rv)
        if k < 20:
            rv = rv / np.linalg.norm(rv) * (np.pi - 1e-4 * k)          # trace near -1: the three non-w branches
        R = Rot.from_rotvec(rv).as_matrix().astype(np.float32)
        q = np.zeros(4, np.float32)
        dm.dm_quat_from_rot(np.ascontiguousarray(R.reshape(9)), q)
        want = Rot.from_matrix(R.astype(np.float64)).as_quat()
        assert min(np.abs(q - want).max(), np.abs(q + want).max()) < 3e-6, k


def _pack(A, b, res, inl):
    out = []
    for i in range(6):
        for j in range(i, 7):
            out.append(b[i] if j == 6 else A[i, j])
    return np.array(out + [res, inl], np.float64)


def test_gn_solve_and_update(dm):
    """the one-thread LDL^T + exp + composition of k_icp_iter / k_icp_finalize (stand-ins for Eigen's LDLT, RGBDOdometry.cpp:447-474, and
    OdometryProvider::computeUpdateSE3) against the oracle's pivoted LDLT restatement, numpy and SciPy: well and badly conditioned
    systems, rotations from 1e-9 to 0.4 rad.  Same cases and gates as the GPU test of the same code."""
    L = mfo.lib()
    rng = np.random.default_rng(5)
    cases = []
    for scale in (1e-9, 1e-4, 1e-2, 0.4):
        M = rng.normal(size=(40, 6))
        A_ = M.T @ M * rng.uniform(1.0, 1e4)
        x_ = np.concatenate([rng.normal(size=3) * 0.01, scale * np.array([0.6, -0.64, 0.48])])
        cases.append((A_, A_ @ x_, 12.5, 1000.0))
    Mi = rng.normal(size=(6, 6))
    Ai = Mi @ np.diag([1e6, 1e5, 1e3, 10.0, 1e-2, 1e-4]) @ Mi.T
    cases.append((Ai, Ai @ np.array([0.01, -0.02, 0.005, 1e-3, 2e-3, -1e-3]), 3.0, 500.0))
    Rprev = Rot.from_rotvec([0.2, -0.1, 0.05]).as_matrix().astype(np.float32)
    tprev = np.array([0.3, -0.2, 1.1], np.float32)
    rt0 = np.eye(4)
    rt0[:3, :3] = Rot.from_rotvec([0.01, 0.02, -0.015]).as_matrix(); rt0[:3, 3] = [0.004, -0.002, 0.001]
    for ci, (A_, b_, res_, inl_) in enumerate(cases):
        x, rt = np.zeros(6), np.zeros(16)
        Rc, tc, trR, trt, st = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(2, np.float32)
        dm.dm_gn_solve_update(_pack(A_, b_, res_, inl_), np.ascontiguousarray(rt0.reshape(16)), np.ascontiguousarray(Rprev.reshape(9)), tprev,
                              x, rt, Rc, tc, trR, trt, st)
        xo = np.zeros(6)
        assert L.mfo_ldlt_solve(np.ascontiguousarray(A_, np.float64), np.ascontiguousarray(b_, np.float64), xo, 6) == 0
        xn = np.linalg.solve(A_, b_)
        tol = max(1e-12, 50 * np.linalg.cond(A_) * 2.2e-16) * np.abs(xn).max()
        assert np.abs(x - xn).max() <= tol and np.abs(x - xo).max() <= tol, ci
        rto = np.ascontiguousarray(rt0.reshape(16).copy())
        L.mfo_update_se3(rto, np.ascontiguousarray(x, np.float64))
        T = np.eye(4); T[:3, :3] = Rot.from_rotvec(x[3:]).as_matrix(); T[:3, 3] = x[:3]
        want = T @ rt0
        assert np.abs(rt.reshape(4, 4) - want).max() < 1e-14 and np.abs(rt - rto).max() < 1e-14, ci
        inc = want.astype(np.float32)
        assert np.array_equal(trR.reshape(3, 3), inc[:3, :3]) and np.array_equal(trt, inc[:3, 3])
        iR = inc[:3, :3].T
        it = -(iR @ inc[:3, 3])
        assert np.abs(Rc.reshape(3, 3) - Rprev @ iR).max() < 2e-6 and np.abs(tc - (Rprev @ it + tprev)).max() < 2e-6, ci
        assert abs(st[0] - np.sqrt(np.float32(res_)) / np.float32(inl_)) < 1e-9 and st[1] == np.float32(inl_)
    # a vanishing pivot zeroes that component instead of dividing by it (Eigen::LDLT::solve on a singular system)
    A0 = np.diag([4.0, 2.0, 0.0, 1.0, 3.0, 5.0]); b0 = np.array([4.0, 2.0, 0.0, 1.0, 3.0, 5.0])
    x = np.zeros(6)
    dm.dm_gn_solve_update(_pack(A0, b0, 1.0, 10.0), np.ascontiguousarray(np.eye(4).reshape(16)), np.ascontiguousarray(np.eye(3, dtype=np.float32).reshape(9)),
                          np.zeros(3, np.float32), x, np.zeros(16), np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(9, np.float32),
                          np.zeros(3, np.float32), np.zeros(2, np.float32))
    assert np.array_equal(x, [1.0, 1.0, 0.0, 1.0, 1.0, 1.0])


def test_pose_inverse_and_fusion_weight(dm):
    """pose_derive: Model::pose^-1 (index map / fuse / clean uniforms) and Model::computeFusionWeight (Model.cpp:449-464) from pose and
    lastPose, against the oracle's restatement (JacobiSVD-free log map) and numpy"""
    rng = np.random.default_rng(7)
    for k in range(200):
        T0 = np.eye(4); T0[:3, :3] = Rot.from_rotvec(rng.normal(size=3) * 0.3).as_matrix(); T0[:3, 3] = rng.normal(size=3)
        ang = 10.0 ** rng.uniform(-5, -1.3)          # 1e-5 .. 0.05 rad: below and above the 0.01 saturation
        d = np.eye(4); d[:3, :3] = Rot.from_rotvec(rng.normal(size=3) / np.sqrt(3) * ang).as_matrix(); d[:3, 3] = rng.normal(size=3) * 10.0 ** rng.uniform(-5, -1.5)
        T1 = T0 @ d
        R1, t1 = np.ascontiguousarray(T1[:3, :3], np.float32), np.ascontiguousarray(T1[:3, 3], np.float32)
        R0, t0 = np.ascontiguousarray(T0[:3, :3], np.float32), np.ascontiguousarray(T0[:3, 3], np.float32)
        Ri, ti, w = np.zeros(9, np.float32), np.zeros(3, np.float32), np.zeros(1, np.float32)
        dm.dm_pose_derive(R1.reshape(9), t1, R0.reshape(9), t0, Ri, ti, w, 0)
        Tinv = np.linalg.inv(T1)
        assert np.abs(Ri.reshape(3, 3) - Tinv[:3, :3]).max() < 1e-6 and np.abs(ti - Tinv[:3, 3]).max() < 5e-6
        T1f = np.eye(4, dtype=np.float32); T1f[:3, :3] = R1; T1f[:3, 3] = t1
        T0f = np.eye(4, dtype=np.float32); T0f[:3, :3] = R0; T0f[:3, 3] = t0
        mfo.lib().mfo_set_weight_literal(0)                      # the accurate double log map (the switch's "off" position since round 3)
        try:
            wo = mfo.fusion_weight(T1f, T0f, 1.0)
        finally:
            mfo.lib().mfo_set_weight_literal(1)
        assert float(w[0]) == wo, (k, float(w[0]), wo)          # the same operations in the same order: identical
        assert 0.5 <= float(w[0]) <= 1.0
        # "literalFusionWeight" (finding F5: the reference's float trace quantises the rotation): the device's literal mode and the
        # oracle's literal mode are the same operations too
        wl = np.zeros(1, np.float32)
        dm.dm_pose_derive(R1.reshape(9), t1, R0.reshape(9), t0, Ri, ti, wl, 1)
        wol = mfo.fusion_weight(T1f, T0f, 1.0)                   # the oracle's default IS the literal mode
        assert float(wl[0]) == wol, (k, float(wl[0]), wol)


def test_mask_id_and_m33_inverse(dm):
    assert [dm.dm_mask_id(v, 3) for v in (0, 1, 2, 3, 255)] == [0, 1, 2, 0, 0] and dm.dm_mask_id(5, 0) == 0
    rng = np.random.default_rng(9)
    for _ in range(50):
        M = (Rot.from_rotvec(rng.normal(size=3)).as_matrix() * rng.uniform(0.5, 2.0)).astype(np.float32)
        inv = np.zeros(9, np.float32)
        dm.dm_m33_inverse(np.ascontiguousarray(M.reshape(9)), inv)
        assert np.abs(inv.reshape(3, 3) @ M - np.eye(3)).max() < 2e-6
