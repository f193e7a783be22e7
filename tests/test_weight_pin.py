"""Model::computeFusionWeight against the reference's own Model::computeFusionWeight / Model::rodrigues2 /
getLastTransform (Core/Model/Model.cpp:449-464, 891-932, Model.h:239) compiled from their text (oracle/build_weight.py; Eigen underneath
is the stand-in of oracle/eigen_shim).

Finding F5: the reference takes cos(theta) from the FLOAT trace of a float matrix, so theta = acos(c) is quantised in steps
of ~4.9e-4 rad near zero -- a rotation of 1e-4 rad per frame reads as none at all, one of 1e-3 rad as 8.5e-4 -- and which step a given
rotation lands on depends on the last bits of the SVD product U V^T.  The weight of a slowly rotating camera is therefore only defined to
one quantum (0.049 of the weight range) in the reference itself.  What can be pinned, and is:
  * everything but that quantisation -- the translation term, the saturation at 0.01, the 0.5 floor, the multiplier, the branches of the
    log map -- exactly;
  * the oracle's LITERAL mode (mfo_set_weight_literal(1): float matrix, float trace), which the device's "literalFusionWeight" switch
    reproduces bit for bit (tests/test_devmath_host.py), lands on the same quantum as the compiled reference text most of the time and never
    further than one and a half quanta away;
  * the accurate log map in double (rounds 1-2's default, now `mfo_set_weight_literal(0)` / `literalFusionWeight = 0`) is equally close in
    the worst case and systematically off below the first quantum.  Since round 3 the LITERAL mode is the default of oracle and device."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from oracle import mfo, mfweight

pytestmark = pytest.mark.skipif(not mfweight.available(), reason="oracle/_ref/libmf_weight.so absent and no /root/reference to build it")


def _pose(rv, t):
    T = np.eye(4)
    T[:3, :3] = Rot.from_rotvec(rv).as_matrix()
    T[:3, 3] = t
    return T.astype(np.float32)


def _sample(rng):
    T0 = _pose(rng.normal(size=3) * 0.4, rng.normal(size=3))
    ang = 10.0 ** rng.uniform(-6.5, -1.0)
    tr = 10.0 ** rng.uniform(-6.5, -1.0)
    rv, t = rng.normal(size=3) / np.sqrt(3) * ang, rng.normal(size=3) / np.sqrt(3) * tr
    d = _pose(rv, t)
    return (T0.astype(np.float64) @ d.astype(np.float64)).astype(np.float32), T0, float(np.linalg.norm(rv)), float(np.linalg.norm(t))


def test_everything_but_the_quantisation_is_exact():
    """translation-dominated motion (the rotation term does not win the max), saturation, floor, multiplier, standing still"""
    rng = np.random.default_rng(3)
    n = 0
    for k in range(3000):
        T1, T0, ang, tr = _sample(rng)
        if not (tr > ang + 1.5e-3 or ang > 0.012):                              # translation wins by more than three quanta, or the rotation saturates
            continue
        n += 1
        w = float(rng.choice([1.0, 0.5, 100.0]))
        ref, orc = mfweight.fusion_weight(T1, T0, w), mfo.fusion_weight(T1, T0, w)
        # 5e-5: float rounding of the relative translation (a general 4x4 inverse there, R^T here; poses ~1 m from the origin) on the 0.01 m scale
        assert abs(ref - orc) <= 5e-5 * w, (k, ang, tr, ref, orc)
        assert 0.5 * w - 1e-6 <= ref <= w + 1e-6
    assert n > 300
    I = np.eye(4, dtype=np.float32)
    assert mfweight.fusion_weight(I, I, 1.0) == 1.0 == mfo.fusion_weight(I, I, 1.0)


def test_slow_rotations_within_the_reference_own_quantum():
    quantum = float(np.sqrt(2 * 2.0 ** -23)) / 0.01                               # one step of acos(1 - k * 2^-24 * 2) on the 0.01 rad scale: 0.049
    rng = np.random.default_rng(5)
    L = mfo.lib()
    d_lit, d_acc = [], []
    for k in range(2000):
        T1, T0, ang, tr = _sample(rng)
        ref = mfweight.fusion_weight(T1, T0, 1.0)
        lit = mfo.fusion_weight(T1, T0, 1.0)                                       # literal mode: the default of oracle and device
        L.mfo_set_weight_literal(0)
        try:
            acc = mfo.fusion_weight(T1, T0, 1.0)
        finally:
            L.mfo_set_weight_literal(1)
        d_lit.append(abs(ref - lit)); d_acc.append(abs(ref - acc))
    d_lit, d_acc = np.array(d_lit), np.array(d_acc)
    print("literal mode: same quantum in %.0f %%, mean |dw| %.4f, max %.3f;  accurate mode: within 1e-3 in %.0f %%, mean %.4f, max %.3f"
          % (100 * (d_lit < 1e-3).mean(), d_lit.mean(), d_lit.max(), 100 * (d_acc < 1e-3).mean(), d_acc.mean(), d_acc.max()))
    assert d_lit.max() <= 1.5 * quantum and d_acc.max() <= 1.5 * quantum
    assert (d_lit < 1e-3).mean() > 0.85 and d_lit.mean() < d_acc.mean()


def test_log_map_branches():
    """rodrigues2: the regular branch, the s < 1e-5 && c > 0 branch (zero vector) and the s < 1e-5 && c <= 0 branch (rotation by pi with
    its sign logic) against SciPy"""
    rng = np.random.default_rng(4)
    for th in (1e-7, 1e-4, 1e-2, 0.5, 2.0, 3.0):
        ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
        r = mfweight.rodrigues2(Rot.from_rotvec(ax * th).as_matrix())
        if th < 1e-5:
            assert np.abs(r).max() < 2e-6          # float rounding of the matrix is all that is left; the branch returns zeros for exact input
        else:
            assert np.abs(r - ax * th).max() < 3e-6 * max(1.0, 1.0 / th), (th, r, ax * th)
    assert np.array_equal(mfweight.rodrigues2(np.eye(3)), np.zeros(3, np.float32))
    for ax in (np.array([1.0, 0, 0]), np.array([0, 1.0, 0]), np.array([0.6, 0.0, 0.8]), np.array([0.36, 0.48, 0.8])):
        r = mfweight.rodrigues2(Rot.from_rotvec(ax * np.pi).as_matrix())
        assert abs(np.linalg.norm(r) - np.pi) < 1e-5 and np.abs(np.abs(r / np.pi) - np.abs(ax)).max() < 2e-3, (ax, r)
