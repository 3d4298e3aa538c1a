"""CPU tests: the oracle (C++ restatement of the reference) against the independent float64 NumPy spec,
finite differences and metamorphic properties.  The reference ships no tests or golden vectors
(SURVEY.md §4), so this is how the oracle is pinned ("parity unpinned" otherwise)."""
import copy

import numpy as np
import pytest

from conftest import rel, blockrel, get_window
from ldso_amd import synth
from oracle import pyoracle as po, spec_np as sp


def _lin(win):
    o = po.OracleWindow(win)
    o.collect_active()
    E = o.linearize_all(False)
    o.apply_res()
    return o, E


def test_struct_layouts_match_header():
    import re, os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "ldso_window.h")).read()
    assert "float color[LDSO_PATTERN_NUM]" in hdr
    assert synth.FRAME_DTYPE.itemsize == 736 and synth.POINT_DTYPE.itemsize == 96 and synth.RESIDUAL_DTYPE.itemsize == 32
    assert synth.RAWJAC_DTYPE.itemsize == 296 and synth.RES_OUT_DTYPE.itemsize == 56 and synth.POINT_OUT_DTYPE.itemsize == 76
    assert synth.SETTINGS_DTYPE.itemsize == 72


def test_make_images_matches_numpy(tiny):
    rng = np.random.default_rng(3)
    img = rng.uniform(1, 254, (tiny.h, tiny.w)).astype(np.float32)
    a = synth.make_images(img, 3)
    b = po.make_images(img, 3)
    for l in range(3):
        assert np.array_equal(a[l], b[l])


def test_pyr_levels_rule():
    assert synth.pyr_levels_used(640, 480) == 4          # 80x60 = 4800 <= 5000 stops (GlobalCalib.cc:24)
    assert synth.pyr_levels_used(1232, 368) == 5


def test_nullspaces_numpy_vs_oracle(tiny):
    T = sp.T44(tiny.frames[2]["worldToCam_evalPT"])
    p, s, a = po.nullspaces(T, 0.03, 1.0)
    P2, S2, A2 = synth.nullspaces_for_evalpt(T, 0.03, 1.0)
    assert rel(p, P2) < 1e-8 and rel(s, S2, 1e-6) < 1e-6 and rel(a, A2) < 1e-6


def test_linearize_vs_float64_spec(tiny):
    o, E = _lin(tiny)
    r = o.get_residuals()
    Jn = sp.linearize_np(tiny)
    act = r["is_active"].astype(bool)
    assert act.sum() > 0.9 * tiny.R
    for k, tol in [("Jpdxi", 1e-5), ("Jpdc", 1e-4), ("Jpdd", 1e-5), ("JIdx", 1e-4), ("JabF", 1e-5), ("resF", 5e-4)]:
        assert rel(r["J"][k][act], Jn[k][act]) < tol, k
    assert rel(r["out"]["state_NewEnergyWithOutlier"][act], Jn["energy"][act]) < 5e-4
    assert rel(r["out"]["centerProjectedTo"][act], Jn["center"][act]) < 1e-5


def test_adjoints_vs_spec(tiny):
    o, _ = _lin(tiny)
    ah, at, _ = o.get_adjoints()
    adH, adT = sp.adjoints_np(tiny.frames)
    F = tiny.F
    for h in range(F):
        for t in range(F):
            assert rel(ah[h + t * F], adH[h, t]) < 1e-6
            assert rel(at[h + t * F], adT[h, t]) < 1e-6


def test_geometric_jacobians_vs_finite_differences(tiny):
    """d(centre projection)/d(absolute frame state, calib, idepth) by central differences vs Jpdxi*Ad^T, Jpdc, Jpdd."""
    w2 = copy.deepcopy(tiny)
    w2.frames["state"] = w2.frames["state_zero"]       # FEJ point == current point
    o, _ = _lin(w2)
    J = o.get_residuals()["J"]
    adH, adT = sp.adjoints_np(w2.frames)
    evalPTs = [f["worldToCam_evalPT"] for f in w2.frames]
    st = [np.array(f["state"]) for f in w2.frames]
    cv = np.array(w2.calib["value"])
    worst = 0
    for ri in [0, 7, 31, 64, 100, 150, 191]:
        h, t, p = (int(w2.residuals[k][ri]) for k in ("host", "target", "point"))
        u, v, idp = (float(w2.points[k][p]) for k in ("u", "v", "idepth"))
        Jrel = np.zeros((2, 8))
        Jrel[:, :6] = J["Jpdxi"][ri]
        for fidx, JA in ((h, Jrel @ adH[h, t].T), (t, Jrel @ adT[h, t].T)):
            for c in range(6):
                sp_, sm = [s.copy() for s in st], [s.copy() for s in st]
                sp_[fidx][c] += 1e-6
                sm[fidx][c] -= 1e-6
                fd = (sp.center_projection(sp_, evalPTs, cv, u, v, idp, h, t) - sp.center_projection(sm, evalPTs, cv, u, v, idp, h, t)) / 2e-6
                worst = max(worst, np.abs(fd - JA[:, c]).max() / max(np.abs(JA[:, c]).max(), 1e-3))
        for c in range(4):
            cp, cm = cv.copy(), cv.copy()
            cp[c] += 1e-7
            cm[c] -= 1e-7
            fd = (sp.center_projection(st, evalPTs, cp, u, v, idp, h, t) - sp.center_projection(st, evalPTs, cm, u, v, idp, h, t)) / 2e-7
            worst = max(worst, np.abs(fd - J["Jpdc"][ri][:, c]).max() / max(np.abs(J["Jpdc"][ri][:, c]).max(), 1e-3))
        fd = (sp.center_projection(st, evalPTs, cv, u, v, idp + 1e-6, h, t) - sp.center_projection(st, evalPTs, cv, u, v, idp - 1e-6, h, t)) / 2e-6
        worst = max(worst, np.abs(fd - J["Jpdd"][ri]).max() / np.abs(J["Jpdd"][ri]).max())
    assert worst < 5e-4


def test_accumulate_stitch_schur_vs_explicit_normal_equations(tiny):
    o, _ = _lin(tiny)
    o.backup_state()
    o.solve_system(0)
    r = o.get_residuals()
    s = o.get_system()
    adH, adT = sp.adjoints_np(tiny.frames)
    ex = sp.explicit_system(tiny, r["J"], r["is_active"].astype(bool), adH, adT)
    assert rel(s["HA"], ex["Hcc"]) < 1e-6 and blockrel(s["HA"], ex["Hcc"]) < 1e-5
    assert rel(s["bA"], ex["bc"]) < 1e-6
    Hpp = np.maximum(ex["Hpp"], 1e-10)
    Hsc = (ex["Hcp"] / Hpp[None, :]) @ ex["Hcp"].T
    bsc = (ex["Hcp"] / Hpp[None, :]) @ ex["bp"]
    assert rel(s["Hsc"], Hsc) < 1e-6 and rel(s["bsc"], bsc) < 1e-6
    assert np.abs(s["HA"] - s["HA"].T).max() <= 1e-9 * np.abs(s["HA"]).max()
    # back-substitution
    pts, _ = o.get_points()
    step = -(ex["bp"] - ex["Hcp"].T @ s["x"]) / Hpp
    assert rel(pts["step"], step) < 1e-5
    # solve: backward error and agreement with numpy on the scaled system
    assert np.linalg.norm(s["HFinal"] @ s["x"] - s["bFinal"]) / np.linalg.norm(s["bFinal"]) < 1e-9
    S = 1 / np.sqrt(np.diag(s["HFinal"]) + 10)
    xs = S * np.linalg.solve(S[:, None] * s["HFinal"] * S[None, :], S * s["bFinal"])
    # the reduced system is ill-conditioned along the gauge directions (cond ~1e8+): x itself is only pinned
    # to ~1e-3 even in fp64, which is why parity on x is asserted through the backward error
    assert rel(s["x"], xs) < 5e-2


def test_priors_only_in_L(tiny):
    o, _ = _lin(tiny)
    o.backup_state()
    o.solve_system(0)
    s = o.get_system()
    d = np.diag(s["HL"])
    assert np.allclose(d[:4], 5e9) and np.allclose(d[4:7], 1e10) and np.allclose(d[7:10], 1e11) and np.allclose(d[10:12], 1e14)
    assert np.abs(s["HL"] - np.diag(d)).max() == 0


def test_mixed_window_mode1(tiny):
    """linearised residuals: H_L must equal the explicit normal equations of the stored J with
    resApprox = res_toZeroF + J*delta (AccumulatedTopHessian.cc:44-63)."""
    w2 = po.make_mixed_window(tiny)
    assert w2.residuals["is_linearized"].sum() > 0
    o, _ = _lin(w2)
    o.backup_state()
    o.solve_system(0)
    a, l, _ = o.counts()
    assert a > 0 and l > 0 and a + l <= w2.R
    s = o.get_system()
    assert np.abs(s["HL"] - np.diag(np.diag(s["HL"]))).max() > 0
    assert np.abs(s["HL"] - s["HL"].T).max() <= 1e-9 * np.abs(s["HL"]).max()


def test_point_permutation_invariance(small):
    """Reversing the point order within each host must not change the stitched system beyond fp32 noise."""
    o, _ = _lin(small)
    o.backup_state(); o.solve_system(0)
    s1 = o.get_system()
    w2 = copy.deepcopy(small)
    order = np.concatenate([np.nonzero(small.points["host"] == h)[0][::-1] for h in range(small.F)])
    inv = np.empty_like(order); inv[order] = np.arange(len(order))
    w2.points = small.points[order].copy()
    res = []
    rb = 0
    for i, pi in enumerate(order):
        sl = slice(small.points["res_begin"][pi], small.points["res_begin"][pi] + small.points["res_count"][pi])
        rr = small.residuals[sl].copy()
        rr["point"] = i
        w2.points["res_begin"][i] = rb
        rb += len(rr)
        res.append(rr)
    w2.residuals = np.concatenate(res)
    o2, _ = _lin(w2)
    o2.backup_state(); o2.solve_system(0)
    s2 = o2.get_system()
    for k in ("HA", "Hsc"):
        assert blockrel(s2[k], s1[k], 8) < 2e-5, k


def test_mt_vs_st_reproducibility_band(small):
    """6-worker IndexThreadReduce mode vs single thread: documents the reference's own fp32 reproducibility."""
    o, E1 = _lin(small)
    o.backup_state(); o.solve_system(0)
    om = po.OracleWindow(small, multithreading=True)
    om.collect_active(); E2 = om.linearize_all(False); om.apply_res(); om.backup_state(); om.solve_system(0)
    assert abs(E1 - E2) / E1 < 1e-9
    s1, s2 = o.get_system(), om.get_system()
    assert blockrel(s2["HA"], s1["HA"], 8) < 1e-5 and blockrel(s2["Hsc"], s1["Hsc"], 8) < 1e-4


def test_optimize_decreases_energy_and_converges(small):
    o = po.OracleWindow(small)
    o.set_force_all_iterations(True)
    rm = o.optimize(6)
    e = o.energy_log()
    assert len(e) == 8 and e[1] < 0.5 * e[0] and e[-1] <= e[1] and 0.5 < rm < 5


def test_optimize_recovers_perturbed_poses():
    """ground truth: with a large initial pose error the optimised rotations end closer to the truth."""
    win = get_window("small", pose_noise_t=1e-2, pose_noise_r=3e-3, state_noise=False)
    o = po.OracleWindow(win)
    o.set_force_all_iterations(True)
    o.optimize(8)
    fr = o.get_frames()
    err0 = err1 = 0
    for k in range(1, win.F):
        T0 = sp.T44(win.frames[k]["worldToCam_evalPT"])
        T1 = sp.T44(fr["pre_worldToCam"][k])
        Tt = win.truth["w2c"][k]
        err0 += np.linalg.norm(synth.so3_log(T0[:3, :3] @ Tt[:3, :3].T))
        err1 += np.linalg.norm(synth.so3_log(T1[:3, :3] @ Tt[:3, :3].T))
    assert err1 < 0.5 * err0


def test_marginalization_flow(small):
    """C5-style: flag oldest frame, flagPointsForRemoval, marginalizePointsF, marginalizeFrame, continue at F-1."""
    o = po.OracleWindow(small)
    o.set_force_all_iterations(True)
    o.optimize(3)
    o.flag_frame(0)
    o.flag_points_for_removal()
    o.drop_points()
    o.marginalize_points()
    HM, bM = o.get_prior()
    assert np.abs(HM).max() > 0 and np.abs(HM - HM.T).max() <= 1e-6 * np.abs(HM).max()
    o.marginalize_frame(0)
    assert o.num_frames() == small.F - 1
    HM2, bM2 = o.get_prior()
    assert HM2.shape[0] == 8 * (small.F - 1) + 4 and np.all(np.isfinite(HM2))
    ex = o.export_window()
    assert ex["F"] == small.F - 1 and len(ex["points"]) > 0 and (ex["residuals"]["target"] < small.F - 1).all()
    rm = o.optimize(3)
    assert np.isfinite(rm)


def test_ldlt_matches_numpy():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(20, 20))
    A = A @ A.T + np.eye(20)
    # exercised through the tracker's 8x8 path indirectly; here: the dense solve inside solve_system was checked
    # by backward error above. This test pins the stand-alone numpy relation used there.
    x = np.linalg.solve(A, np.ones(20))
    assert np.allclose(A @ x, 1)
