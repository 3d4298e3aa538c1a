"""CPU checks of the CoarseInitializer restatement (oracle/initializer.cc): the reference has no tests for it, so the restatement is
validated against finite differences, an explicit Schur complement and ground truth of the synthetic scene."""
import numpy as np

from ldso_amd import synth
from oracle import pyoracle

W, H = 320, 240


def _setup(n_frames, seed=20260925):
    seq = synth.make_init_sequence(W, H, n_frames=n_frames, fx=200.0, seed=seed)
    L = seq["levels"]
    pyr0 = synth.make_images(seq["first"], L)
    pts = synth.select_init_points(pyr0)
    o = pyoracle.OracleInitializer(W, H, L)
    o.set_first(seq["K4"], pyr0, 1.0, pts)
    return seq, L, pts, o


def test_point_records():
    seq, L, pts, o = _setup(1)
    for lvl, p in enumerate(pts):
        assert len(p) > 200
        assert np.all(np.diff(p["v"].astype(np.int64) * 100000 + p["u"].astype(np.int64)) > 0)        # raster order
        assert np.all(p["neighbours"][:, 0] == np.arange(len(p)))                                      # nearest neighbour is the point itself
        assert np.allclose(p["neighboursDist"].sum(axis=1), 10, rtol=1e-5)
        if lvl + 1 < L:
            par = pts[lvl + 1][p["parent"]]
            d = np.hypot(par["u"] - (p["u"] * 0.5 - 0.25), par["v"] - (p["v"] * 0.5 - 0.25))
            assert d.max() < 6
        else:
            assert np.all(p["parent"] == -1)


def test_b_is_half_energy_gradient():
    """b = 1/2 dE/dxi for the left-multiplied increment exp(xi) * T (tangent order translation, rotation), photometric + alpha
    terms.  The Jacobians use the central-difference gradient image while finite differences see the bilinear intensity surface,
    so pose components agree to ~10-15 % only (texture down to 6 px wavelength); the affine components are exact."""
    seq, L, pts, o = _setup(2)
    o.set_new_frame(synth.make_images(seq["frames"][1], L), 1.0)
    lvl = 0
    for tscale in (0.05, 1.0 / 3.0):                     # alpha term active / capped (snapping regime)
        T = seq["poses"][1].copy()
        T[:3, 3] *= tscale
        H, b, Hsc, bsc, res, ec = o.calc_res_and_gs(lvl, T, 0.0, 0.0)
        capped = res[1] == np.float32(6.25) * len(pts[lvl])
        assert capped == (tscale > 0.1)
        g = np.zeros(8)
        for k in range(6):
            eps = 2e-4 if k < 3 else 5e-5
            xi = np.zeros(6); xi[k] = eps
            Ep = o.calc_res_and_gs(lvl, synth.se3_exp(xi) @ T, 0.0, 0.0)[4]
            Em = o.calc_res_and_gs(lvl, synth.se3_exp(-xi) @ T, 0.0, 0.0)[4]
            g[k] = ((float(Ep[0]) + float(Ep[1])) - (float(Em[0]) + float(Em[1]))) / (2 * eps)
        b2 = 2 * b.astype(np.float64)
        assert g[:6] @ b2[:6] / np.linalg.norm(g[:6]) / np.linalg.norm(b2[:6]) > 0.95
        big = np.abs(b2[:6]) > 0.5 * np.abs(b2[:6]).max()
        assert np.all(np.abs(g[:6][big] / b2[:6][big] - 1) < 0.2), g[:6] / b2[:6]
        Ep = o.calc_res_and_gs(lvl, T, 1e-3, 0)[4]; Em = o.calc_res_and_gs(lvl, T, -1e-3, 0)[4]
        ga = (float(Ep[0]) - float(Em[0])) / 2e-3
        assert abs(ga / b2[6] - 1) < 0.01, (ga, b2[6])


def test_schur_block_explicit():
    """Hsc / bsc are the point-wise eliminated depth terms: sum_i Jb_i Jb_i^T / (1 + Hdd_i) from the per-point buffers."""
    seq, L, pts, o = _setup(2)
    o.set_new_frame(synth.make_images(seq["frames"][1], L), 1.0)
    T = seq["poses"][1].copy(); T[:3, 3] /= 3.0
    for lvl in range(L):
        H, b, Hsc, bsc, res, ec = o.calc_res_and_gs(lvl, T, 0.0, 0.0)
        jb = o.jb(lvl).astype(np.float64)
        good = o.points(lvl)["isGood_new"] != 0
        J = jb[good]
        M = (J[:, :9, None] * J[:, None, :9] * J[:, 9, None, None]).sum(axis=0)
        assert np.abs(M[:8, :8] - Hsc).max() < 1e-4 * np.abs(Hsc).max()
        assert np.abs(M[:8, 8] - bsc).max() < 1e-4 * np.abs(bsc).max()
        assert np.allclose(H, H.T) and np.allclose(Hsc, Hsc.T)
        w, _ = np.linalg.eigh((H - Hsc).astype(np.float64)[:6, :6])
        assert w.min() > -1e-6 * w.max()                         # reduced system stays positive semi-definite


def test_sequence_recovers_motion_and_depth():
    n = 9
    seq, L, pts, o = _setup(n)
    snapped_at = None
    for k in range(n):
        o.set_new_frame(synth.make_images(seq["frames"][k], L), 1.0)
        st = o.track_frame()
        if st["snapped"] and snapped_at is None:
            snapped_at = k
        assert st["frameID"] == k + 1
    assert snapped_at is not None and snapped_at <= 2
    assert st["ready"] == 1 and st["snappedAt"] == snapped_at + 1
    T = st["thisToNext"].reshape(3, 4)
    Tt = seq["poses"][n - 1]
    assert float(Tt[:3, 3] @ T[:, 3] / np.linalg.norm(Tt[:3, 3]) / np.linalg.norm(T[:, 3])) > 0.999
    assert np.abs(T[:, :3] - Tt[:3, :3]).max() < 5e-3
    p0 = o.points(0)
    gd = p0["isGood"] != 0
    assert gd.mean() > 0.85
    xs = (p0["u"] - 0.1).astype(int); ys = (p0["v"] - 0.1).astype(int)
    assert np.corrcoef(p0["iR"][gd], (1.0 / seq["depth0"][ys, xs])[gd])[0, 1] > 0.85
