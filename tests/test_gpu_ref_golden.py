"""-m gpu: the HIP kernels against REFERENCE outputs (tests/golden/ref_vectors.npz = Core/Cuda/{reduce,cudafuncs,segmentation}.cu
of martinruenz/maskfusion compiled for the CPU, see tests/golden/make_ref_golden.py), through the C ABI.  No oracle in between.
Tolerances: integer / byte / index results bit-exact; float maps 2e-6 relative (normals 5e-5: v_rsq_f32 vs 1/sqrtf);
reduced normal equations 1e-4 of max|A| with the inlier count exact.  SURVEY.md rows a3-a5, a7-a10, a12, a20 (GPU half)."""
import os

import numpy as np
import pytest

from gpu_util import dev, empty, host, nan_equal_close

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_vectors.npz"))


def gi(case, k):
    a = G[f"{case}/in_{k}"]
    return G[str(a)] if a.dtype.kind == "U" else a


def go(case, k):
    return G[f"{case}/out_{k}"]


def _valid_close(got, ref, rtol, atol):
    """planar 3-channel maps: validity (x plane NaN) identical, channels close where valid (the reference leaves y/z of an
    invalid pixel as allocated, cudafuncs.cu:128-131)"""
    ok = ~np.isnan(ref[0])
    assert np.array_equal(ok, ~np.isnan(got[0]))
    for c in range(3):
        err = np.abs(got[c][ok].astype(np.float64) - ref[c][ok])
        assert (err <= atol + rtol * np.abs(ref[c][ok])).all(), (c, err.max())


def test_depth_pyramid_and_maps(hip):
    src = gi("pyrdown_f", "src")
    H, W = src.shape
    d = dev(src); o1 = empty((H // 2, W // 2)); o2 = empty((H // 4, W // 4))
    assert hip.mf_k_pyrdown_f(d.data_ptr(), o1.data_ptr(), W, H, None) == 0
    assert nan_equal_close(host(o1), go("pyrdown_f", "l1"), 2e-6, 1e-7)[1] == 0
    d1 = dev(go("pyrdown_f", "l1"))
    assert hip.mf_k_pyrdown_f(d1.data_ptr(), o2.data_ptr(), W // 2, H // 2, None) == 0
    assert nan_equal_close(host(o2), go("pyrdown_f", "l2"), 2e-6, 1e-7)[1] == 0
    fx, fy, cx, cy, cut = [float(x) for x in gi("vmap_nmap", "K")]
    dd = dev(gi("vmap_nmap", "depth")); v, n = empty((3, H, W)), empty((3, H, W))
    assert hip.mf_k_vmap_nmap(dd.data_ptr(), v.data_ptr(), n.data_ptr(), W, H, fx, fy, cx, cy, cut, None) == 0
    _valid_close(host(v), go("vmap_nmap", "vmap"), 2e-6, 1e-7)
    _valid_close(host(n), go("vmap_nmap", "nmap"), 5e-5, 1e-5)


def test_model_pyramid(hip):
    v4, n4 = gi("model_maps", "v4"), gi("model_maps", "n4")
    H, W = v4.shape[:2]
    R, t = np.ascontiguousarray(gi("model_maps", "R").reshape(9)), np.ascontiguousarray(gi("model_maps", "t"))
    tot = sum((W >> i) * (H >> i) * 3 for i in range(3))
    dv, dn = empty(tot), empty(tot)
    a, b = dev(v4), dev(n4)
    assert hip.mf_k_model_pyramid(a.data_ptr(), b.data_ptr(), R.ctypes.data, t.ctypes.data, dv.data_ptr(), dn.data_ptr(), W, H, None) == 0
    gv, gn = host(dv), host(dn)
    # level 0 of the fused kernel = tranformMaps(copyMaps(.)): compare with the reference's transform output
    sz = W * H * 3
    _valid_close(gv[:sz].reshape(3, H, W), go("model_maps", "tr_v"), 2e-6, 2e-6)
    _valid_close(gn[:sz].reshape(3, H, W), go("model_maps", "tr_n"), 2e-5, 2e-6)
    # levels 1, 2 = transform of the reference's resized maps: validity pattern of the resize + rigid transform in numpy float64
    off = sz
    Rm = gi("model_maps", "R").astype(np.float64); tv = gi("model_maps", "t").astype(np.float64)
    for i, (kv, kn) in enumerate((("res_v1", "res_n1"), ("res_v2", "res_n2")), start=1):
        w, h = W >> i, H >> i
        s = w * h * 3
        rv, rn = go("model_maps", kv).astype(np.float64), go("model_maps", kn).astype(np.float64)
        ev = np.einsum("ij,jhw->ihw", Rm, rv) + tv[:, None, None]
        en = np.einsum("ij,jhw->ihw", Rm, rn)
        _valid_close(gv[off:off + s].reshape(3, h, w), ev, 2e-6, 2e-6)
        _valid_close(gn[off:off + s].reshape(3, h, w), en, 2e-5, 2e-6)
        off += s


def test_icp_step(hip):
    c = "icp_step"
    fx, fy, cx, cy = [float(x) for x in gi(c, "K")]
    vc, nc, vp, npv = gi(c, "vc"), gi(c, "nc"), gi(c, "vp"), gi(c, "np")
    _, H, W = vc.shape
    args = [np.ascontiguousarray(gi(c, k).reshape(-1)) for k in ("Rcurr", "tcurr", "Rprev_inv", "tprev")]
    out = empty(32)
    d = [dev(x) for x in (vc, nc, vp, npv)]
    assert hip.mf_k_icp_step(args[0].ctypes.data, args[1].ctypes.data, d[0].data_ptr(), d[1].data_ptr(), args[2].ctypes.data,
                             args[3].ctypes.data, fx, fy, cx, cy, d[2].data_ptr(), d[3].data_ptr(), 0.10,
                             float(np.sin(np.float32(20.0 * 3.14159254 / 180.0))), W, H, out.data_ptr(), None) == 0
    g = host(out)
    gA, gb = np.zeros((6, 6)), np.zeros(6)
    s = 0
    for i in range(6):           # the reference's host unpack (reduce.cu:510-521)
        for j in range(i, 7):
            if j == 6:
                gb[i] = g[s]
            else:
                gA[i, j] = gA[j, i] = g[s]
            s += 1
    rA, rb, rres = go(c, "A"), go(c, "b"), go(c, "res")
    assert g[28] == rres[1]
    scale = np.abs(rA).max()
    assert np.abs(gA - rA).max() <= 1e-4 * scale and np.abs(gb - rb).max() <= 1e-4 * np.abs(rb).max()
    assert abs(g[27] - rres[0]) <= 1e-4 * rres[0]


def test_intensity_pyramid_derivatives(hip):
    import torch
    rgba = gi("intensity", "rgba")
    H, W = rgba.shape[:2]
    a = dev(rgba); g = empty((H, W), torch.uint8)
    assert hip.mf_k_intensity(a.data_ptr(), 4, g.data_ptr(), W * H, None) == 0
    assert np.array_equal(host(g), go("intensity", "gray"))
    a3 = dev(np.ascontiguousarray(rgba[..., :3]))
    assert hip.mf_k_intensity(a3.data_ptr(), 3, g.data_ptr(), W * H, None) == 0
    assert np.array_equal(host(g), go("intensity", "gray"))
    l1, l2 = empty((H // 2, W // 2), torch.uint8), empty((H // 4, W // 4), torch.uint8)
    assert hip.mf_k_pyrdown_u8(g.data_ptr(), l1.data_ptr(), W, H, None) == 0
    assert hip.mf_k_pyrdown_u8(l1.data_ptr(), l2.data_ptr(), W // 2, H // 2, None) == 0
    assert np.array_equal(host(l1), go("pyrdown_u8", "l1")) and np.array_equal(host(l2), go("pyrdown_u8", "l2"))
    dx, dy = empty((H, W), torch.int16), empty((H, W), torch.int16)
    assert hip.mf_k_derivative_images(g.data_ptr(), dx.data_ptr(), dy.data_ptr(), W, H, None) == 0
    assert np.array_equal(host(dx), go("derivative", "dx")) and np.array_equal(host(dy), go("derivative", "dy"))


@pytest.mark.parametrize("case", ["rgb_residual", "rgb_residual_l0scale"])
def test_rgb_residual(hip, case):
    import torch
    dxh, dyh = gi(case, "dIdx"), gi(case, "dIdy")
    H, W = dxh.shape
    t = [dev(gi(case, k)) for k in ("dIdx", "dIdy", "lastDepth", "nextDepth", "lastImage", "nextImage")]
    kt, krk = np.ascontiguousarray(gi(case, "kt")), np.ascontiguousarray(gi(case, "krkinv"))
    cor = empty((W * H * 8,), torch.uint8)
    sums = np.zeros(2, np.int32)
    assert hip.mf_k_rgb_residual(float(gi(case, "minScale")), t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(),
                                 t[4].data_ptr(), t[5].data_ptr(), 0.07, kt.ctypes.data, krk.ctypes.data, W, H, cor.data_ptr(),
                                 sums.ctypes.data, None) == 0
    rs, rc = [int(x) for x in go(case, "sigma_count")]
    assert sums.tolist() == [rc, rs]
    ref = go(case, "corres")
    got = host(cor).view(np.dtype([("u0", np.int16), ("v0", np.int16), ("diff", np.float32)]))
    v = ref["valid"] != 0
    assert np.array_equal(got["u0"] >= 0, v)
    assert np.array_equal(got["u0"][v], ref["zx"][v]) and np.array_equal(got["v0"][v], ref["zy"][v]) and np.array_equal(got["diff"][v], ref["diff"][v])


def test_rgb_step(hip):
    import torch
    c = "rgb_step"
    ref_c = gi(c, "corres")
    H, W = gi(c, "dIdx").shape
    packed = np.zeros(W * H, np.dtype([("u0", np.int16), ("v0", np.int16), ("diff", np.float32)]))
    v = ref_c["valid"] != 0
    packed["u0"] = np.where(v, ref_c["zx"], -1); packed["v0"] = np.where(v, ref_c["zy"], -1); packed["diff"] = np.where(v, ref_c["diff"], 0)
    # the device form evaluates projectToPointCloud on the fly from the depth the cloud was made of
    depth = np.ascontiguousarray(gi(c, "cloud")[..., 2])
    fx, fy = [float(x) for x in gi(c, "K")]
    K4 = gi("project_cloud", "K")
    d = [dev(packed.view(np.uint8)), dev(depth), dev(gi(c, "dIdx")), dev(gi(c, "dIdy"))]
    out = np.zeros(32, np.float64)
    assert hip.mf_k_rgb_step(d[0].data_ptr(), float(gi(c, "sigma")), d[1].data_ptr(), fx, fy, float(K4[2]), float(K4[3]), d[2].data_ptr(),
                             d[3].data_ptr(), 0.125, W, H, out.ctypes.data, None) == 0
    gA, gb = np.zeros((6, 6)), np.zeros(6)
    s = 0
    for i in range(6):
        for j in range(i, 7):
            if j == 6:
                gb[i] = out[s]
            else:
                gA[i, j] = gA[j, i] = out[s]
            s += 1
    rA, rb = go(c, "A"), go(c, "b")
    assert np.abs(gA - rA).max() <= 1e-4 * np.abs(rA).max() and np.abs(gb - rb).max() <= 1e-4 * np.abs(rb).max()


@pytest.mark.parametrize("case", ["edges_gui", "edges_core", "edges_r2"])
def test_geometric_edges(hip, case):
    import torch
    wD, wC, th, rad, it = [float(x) for x in gi(case, "prm")]
    v, n = gi(case, "vmap"), gi(case, "nmap")
    _, H, W = v.shape
    # invalid pixels: the reference has NaN in x and allocation garbage (here: zeros) in y/z; the device maps carry NaN in all three
    vv, nn = v.copy(), n.copy()
    vv[:, np.isnan(v[0])] = np.nan
    nn[:, np.isnan(n[0])] = np.nan
    dv, dn = dev(vv), dev(nn)
    e, b, tmp = empty((H, W)), empty((H, W), torch.uint8), empty((H, W), torch.uint8)
    assert hip.mf_k_geometric_edges(dv.data_ptr(), dn.data_ptr(), e.data_ptr(), b.data_ptr(), tmp.data_ptr(), W, H, wD, wC, th, int(rad),
                                    int(it), None) == 0
    ge, gb = host(e), host(b)
    ref_e = go(case, "edge")
    assert np.array_equal(np.isnan(ge), np.isnan(ref_e))
    ok = ~np.isnan(ref_e)
    assert np.abs(ge[ok] - ref_e[ok]).max() <= 1e-5 * max(1.0, np.abs(ref_e[ok]).max())
    # the binary image flips only where the float edge value sits within rounding of the threshold
    diff = gb != go(case, "inverted")
    near = np.zeros_like(diff)
    near[ok] = np.abs(ref_e[ok] - th) < 1e-5 * max(1.0, th)
    assert diff.sum() <= 9 * near.sum(), (int(diff.sum()), int(near.sum()))
