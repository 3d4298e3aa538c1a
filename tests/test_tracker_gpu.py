"""GPU parity of CoarseTracker (C-ABI ldso_tr_*) against the oracle: makeCoarseDepthL0 point clouds exact,
calcRes / calcGSSSE within 1e-4, trackNewestCoarse end pose within 1e-4 of the motion with the same iteration count."""
import numpy as np
import pytest

from conftest import rel
from tracker_common import tracker_scenario
from ldso_amd import synth, binding
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def make_pair(sc):
    win = sc["win"]
    o = po.OracleTracker(win.w, win.h, sc["levels"], win.settings, win.calib)
    g = binding.Tracker(win.w, win.h, sc["levels"], win.settings, win.calib)
    for t in (o, g):
        t.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"])
        t.set_new_frame(sc["new_pyr"], 1.0)
    return o, g


@pytest.mark.parametrize("name,levels", [("small", None), ("C3", None), ("C3", 5)])
def test_point_cloud_exact(name, levels):
    sc = tracker_scenario(name, levels=levels)
    o, g = make_pair(sc)
    for l in range(sc["levels"]):
        a, b = o.pc(l), g.pc(l)
        assert len(a[0]) == len(b[0]) > 0
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])             # indices bit-exact
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])             # deterministic scatter: idepth bit-exact too


def test_point_cloud_with_colliding_points_is_exact_and_repeatable():
    """Several reference points on one pixel: the level-0 sums are float additions in point order (CoarseTracker.cc:268-283);
    the device kernel adds them in that order too (no atomics): idepth bit-exact, identical from run to run."""
    sc = tracker_scenario("small")
    pts = np.array(sc["pts"], np.float32).copy()
    rng = np.random.default_rng(5)
    for k in range(40):                                    # 40 pixels with 3-5 points each, far apart in the list
        src = int(rng.integers(0, len(pts)))
        for c in range(int(rng.integers(2, 5))):
            dst = int(rng.integers(0, len(pts)))
            pts[dst, 0] = pts[src, 0] + rng.uniform(-0.3, 0.3); pts[dst, 1] = pts[src, 1] + rng.uniform(-0.3, 0.3)
            pts[dst, 2] = pts[src, 2] * rng.uniform(0.7, 1.4); pts[dst, 3] = pts[src, 3] * rng.uniform(0.5, 2.0)
    sc = dict(sc, pts=pts)
    o, g = make_pair(sc)
    ref = [g.pc(l) for l in range(sc["levels"])]
    for l in range(sc["levels"]):
        a, b = o.pc(l), ref[l]
        assert len(a[0]) == len(b[0]) > 0
        for q in range(4):
            assert np.array_equal(a[q], b[q]), (l, q)
    for rep in range(3):
        g.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, pts)
        for l in range(sc["levels"]):
            for q in range(4):
                assert np.array_equal(g.pc(l)[q], ref[l][q])


@pytest.mark.parametrize("name", ["small", "C3"])
def test_calc_res_and_gs(name):
    sc = tracker_scenario(name)
    o, g = make_pair(sc)
    a, b = sc["new_aff"]
    for T in (np.eye(4), sc["T_true"]):
        for lvl in range(sc["levels"]):
            for cutoff in (20.0, 1e9):
                ro, no = o.calc_res(lvl, T, a, b, cutoff)
                rg, ng = g.calc_res(lvl, T, a, b, cutoff)
                assert no == ng and ro[1] == rg[1]                                   # counts bit-exact
                assert abs(ro[0] - rg[0]) <= 1e-4 * abs(ro[0])
                assert rel(rg[2:], ro[2:], 1e-6) < 1e-4
                Ho, bo = o.calc_gs(lvl, T, a, b)
                Hg, bg = g.calc_gs(lvl, T, a, b)
                assert rel(Hg, Ho) < 1e-4 and rel(bg, bo) < 1e-4


@pytest.mark.parametrize("name,levels", [("small", None), ("C3", None), ("C3", 5)])
def test_track_matches_oracle(name, levels):
    sc = tracker_scenario(name, levels=levels)
    o, g = make_pair(sc)
    a, b = sc["new_aff"]
    ro = o.track(np.eye(4), a, b, sc["levels"] - 1)
    rg = g.track(np.eye(4), a, b, sc["levels"] - 1)
    assert ro["ok"] and rg["ok"]
    To, Tg = np.eye(4), np.eye(4)
    To[:3, :4] = ro["T"]; Tg[:3, :4] = rg["T"]
    d = np.linalg.norm(synth.se3_log(Tg @ np.linalg.inv(To)))
    m = np.linalg.norm(synth.se3_log(To))
    # the LM loop stops at |inc| <= 1e-3, but device and oracle take the same accept / reject path: observed d / m = 2e-6 (the float sums
    # are grouped differently), residuals 3e-6, identical iteration counts
    assert d < 1e-4 * m + 1e-6
    assert abs(rg["iterations"] - ro["iterations"]) <= 1
    assert rel(rg["lastResiduals"][:sc["levels"]], ro["lastResiduals"][:sc["levels"]]) < 1e-4
    err = np.linalg.norm(synth.se3_log(Tg @ np.linalg.inv(sc["T_true"])))
    assert err < 0.25 * np.linalg.norm(synth.se3_log(sc["T_true"]))


def _solve8_device(H, b, diag_scale):
    import ctypes as C
    L = binding.lib()
    H = np.ascontiguousarray(H, np.float64); b = np.ascontiguousarray(b, np.float64); x = np.zeros(8); piv = C.c_int(-1)
    rc = L.ldso_tr_debug_solve8(H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.c_double(diag_scale), x.ctypes.data_as(C.c_void_p), C.byref(piv))
    assert rc == 0, L.ldso_last_error()
    return x, piv.value


def _solve8_oracle(H, b, diag_scale):
    import ctypes as C
    L = po.lib()
    A = np.ascontiguousarray(H, np.float64).copy(); A[np.diag_indices(8)] *= diag_scale
    nb = np.ascontiguousarray(-np.asarray(b, np.float64)); x = np.zeros(8)
    L.orc_ldlt_solve8(A.ctypes.data_as(C.c_void_p), nb.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
    return x


def test_lm_solve_unpivoted_in_registers_and_pivoted_when_rank_deficient():
    """the 8 x 8 LM solve of the tracking kernel is an unpivoted LDL^T in one lane's registers, the reference runs Eigen's diagonally pivoted one
    (`Hl.ldlt().solve(-b)`, CoarseTracker.cc:120-128).  The device solve watches its pivots: one that lost six digits against the diagonal entry it started from sends
    the system through the pivoted factorisation (the reference's algorithm).  Against the oracle's restatement of Eigen's LDLT: (i) well-conditioned SPD systems of the
    tracker's scaling (columns scaled 1 / 0.5 / 10 / 1000): the register path, solution 1e-10; (ii) rank 5 with a damping that no longer regularises it
    (lambda = 1e-9): the pivoted path, the oracle's solution.  In a track the multiplicative damping (lambda >= 1e-3 after a rejected step) keeps even a one-point
    reference (five entries per level) on the register path - (iii): such a track follows the oracle's, iteration by iteration."""
    rng = np.random.default_rng(11)
    sc = np.array([1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 10.0, 1000.0])
    for rep in range(20):
        J = rng.normal(size=(200, 8)) * sc
        H, b = J.T @ J / 200, J.T @ rng.normal(size=200) / 200
        lam = float(10.0 ** rng.uniform(-4, 0))
        x, piv = _solve8_device(H, b, 1 + lam)
        xo = _solve8_oracle(H, b, 1 + lam)
        assert piv == 0 and np.abs(x - xo).max() <= 1e-10 * np.abs(xo).max(), (rep, piv)
    for rep in range(10):
        J = rng.normal(size=(5, 8)) * sc                      # five residuals: rank 5
        H, b = J.T @ J / 5, J.T @ rng.normal(size=5) / 5
        x, piv = _solve8_device(H, b, 1 + 1e-9)
        xo = _solve8_oracle(H, b, 1 + 1e-9)
        assert piv == 1, rep
        # the same factorisation in the same precision; the directions the data do not determine are divided by pivots of ~1e-9: compare what the system determines
        r, ro_ = (H * (1 + 1e-9 * np.eye(8))) @ x + b, (H * (1 + 1e-9 * np.eye(8))) @ xo + b
        assert np.abs(r).max() <= 1e-6 * np.abs(b).max() and np.abs(ro_).max() <= 1e-6 * np.abs(b).max(), (rep, np.abs(r).max(), np.abs(ro_).max())
        assert np.abs(x - xo).max() <= 1e-3 * np.abs(xo).max(), rep
    scn = tracker_scenario("small")
    a, b_ = scn["new_aff"]
    pts = np.array(scn["pts"], np.float32)
    for n in (2, 1):
        o, g = make_pair(dict(scn, pts=pts[np.linspace(0, len(pts) - 1, n).astype(int)]))
        ro = o.track(np.eye(4), a, b_, scn["levels"] - 1); rg = g.track(np.eye(4), a, b_, scn["levels"] - 1)
        assert g.last_track_pivoted_solves() == 0
        assert bool(rg["ok"]) == bool(ro["ok"]) and rg["iterations"] == ro["iterations"], n
        To, Tg = np.eye(4), np.eye(4)
        To[:3, :4] = ro["T"]; Tg[:3, :4] = rg["T"]
        assert np.linalg.norm(synth.se3_log(Tg @ np.linalg.inv(To))) < 1e-3 * np.linalg.norm(synth.se3_log(To)) + 1e-5, n


def test_track_batch_equals_single():
    sc = tracker_scenario("small")
    o, g = make_pair(sc)
    a, b = sc["new_aff"]
    guesses = [np.eye(4), sc["T_true"], synth.se3_exp([0.01, 0, 0, 0, 0.002, 0])]
    singles = [g.track(T, a, b, sc["levels"] - 1) for T in guesses]
    batch = g.track_batch(guesses, [(a, b)] * 3, sc["levels"] - 1)
    for i in range(3):
        assert np.array_equal(singles[i]["T"], batch["T"][i]) and singles[i]["iterations"] == batch["iterations"][i]


@pytest.mark.parametrize("nhyp", [1, 16, 17, 21, 32, 33, 64, 70, 128])
def test_cooperative_group_sizes_agree(nhyp):
    """ldso_tr_track_batch picks G = 16 / 12 / 8 / 4 / 1 cooperating workgroups per hypothesis from the hypothesis count (nhyp * G <= CUs).
    The variants differ only in how the 52 sums are grouped: every hypothesis of every batch size must land on the oracle's pose
    (LM convergence tolerance) with the same accept flag, on the full-size pair where all five levels are shared."""
    sc = tracker_scenario("C3", levels=5)
    o, g = make_pair(sc)
    a, b = sc["new_aff"]
    L = sc["levels"]
    rng = np.random.default_rng(nhyp)
    guesses = [synth.se3_exp(np.concatenate([rng.normal(0, 2e-3, 3), rng.normal(0, 5e-4, 3)])) for _ in range(nhyp)]
    guesses[0] = np.eye(4)
    batch = g.track_batch(guesses, [(a, b)] * nhyp, L - 1)
    assert all(batch["ok"])
    for i in sorted({0, nhyp // 2, nhyp - 1}):
        ro = o.track(guesses[i], a, b, L - 1)
        assert ro["ok"]
        # LM stops at |inc| <= 1e-3: end poses of runs that differ only in the summation order of the float sums agree to a fraction of it
        assert np.abs(batch["T"][i] - ro["T"]).max() < 1e-4, (i, np.abs(batch["T"][i] - ro["T"]).max())
        assert abs(batch["iterations"][i] - ro["iterations"]) <= 2
        single = g.track(guesses[i], a, b, L - 1)                      # G = 16
        assert np.abs(batch["T"][i] - single["T"]).max() < 1e-4
    # all hypotheses converge to the same pose (they start close to each other)
    Ts = np.array(batch["T"])
    assert np.abs(Ts - Ts[0]).max() < 1e-4


def test_abort_on_min_res():
    sc = tracker_scenario("small")
    o, g = make_pair(sc)
    a, b = sc["new_aff"]
    mr = np.full(5, 1e-3)
    ro = o.track(np.eye(4), a, b, sc["levels"] - 1, mr)
    rg = g.track(np.eye(4), a, b, sc["levels"] - 1, mr)
    assert not ro["ok"] and not rg["ok"]


def test_make_images_on_device_bit_exact():
    """a16 FrameHessian::makeImages on the device (images.hip) against the oracle restatement: every level of the pyramid of the
    tracker's new frame, and level 0 of a bundle-adjustment image slot, bit for bit (rows 0 / h-1 keep zero gradients)."""
    sc = tracker_scenario("small")
    win = sc["win"]
    color = np.ascontiguousarray(sc["new_pyr"][0][:, :, 0])
    ref = po.make_images(color, sc["levels"])
    g = binding.Tracker(win.w, win.h, sc["levels"], win.settings, win.calib)
    g.set_new_frame_image(color, 1.0)
    for l in range(sc["levels"]):
        assert np.array_equal(g.get_new_frame_level(l), ref[l]), l
        assert np.array_equal(ref[l], sc["new_pyr"][l])          # and it is what the synthetic generator stored
    ba = binding.BA(win.w, win.h, 2, 4)
    ba.set_image_raw(1, color)
    assert np.array_equal(ba.get_image(1), ref[0])
    # tracking from the device-built pyramid gives the same result as from the uploaded one
    o, g2 = make_pair(sc)
    g.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"])
    a, b = sc["new_aff"]
    r1 = g.track(np.eye(4), a, b, sc["levels"] - 1); r2 = g2.track(np.eye(4), a, b, sc["levels"] - 1)
    assert np.array_equal(r1["T"], r2["T"]) and r1["iterations"] == r2["iterations"]


def test_batched_hypotheses_reproduce_sequential_loop():
    """the try loop of FullSystem::trackNewCoarse (FullSystem.cc:319-356) run sequentially with the growing
    `achievedRes` abort thresholds, against ONE batched launch + ldso_tr_select_hypothesis: same winner, same number of tries,
    same achievedRes - for an early-exit threshold that stops the loop and for one that never does."""
    sc = tracker_scenario("small")
    o, g = make_pair(sc)
    a, b = sc["new_aff"]
    coarsest = sc["levels"] - 1
    rng = np.random.default_rng(3)
    # a bad first guess, some mediocre ones, the truth, more mediocre ones
    guesses = [synth.se3_exp([0.05, -0.03, 0.02, 0.01, -0.012, 0.008])]
    guesses += [synth.se3_exp(rng.normal(0, 1, 6) * [0.01, 0.01, 0.01, 0.003, 0.003, 0.003]) for _ in range(4)]
    guesses += [sc["T_true"]]
    guesses += [synth.se3_exp(rng.normal(0, 1, 6) * [0.02, 0.02, 0.02, 0.005, 0.005, 0.005]) for _ in range(4)]
    batch = g.track_batch(guesses, [(a, b)] * len(guesses), coarsest)
    for last_rmse0, thr in ((float("nan"), 1.5), (1e9, 1.5)):
        # sequential reference loop on the same device tracker
        achieved = np.full(5, np.nan); have, win, tries = False, -1, 0
        for i, T in enumerate(guesses):
            r = g.track(T, a, b, coarsest, achieved.copy())
            tries += 1
            lr = np.asarray(r["lastResiduals"], np.float64)
            if r["ok"] and np.isfinite(np.float32(lr[0])) and not (lr[0] >= achieved[0]):
                win, have = i, True
            if have:
                for l in range(5):
                    if not np.isfinite(np.float32(achieved[l])) or achieved[l] > lr[l]:
                        achieved[l] = lr[l]
            if have and achieved[0] < last_rmse0 * thr:
                break
        best, used, ach = g.select_hypothesis(batch, coarsest, last_rmse0, thr)
        assert best == win and used == tries
        assert np.array_equal(np.isnan(ach), np.isnan(achieved)) and np.allclose(ach[~np.isnan(ach)], achieved[~np.isnan(achieved)], rtol=0, atol=0)
    assert win >= 0


@pytest.mark.parametrize("lost", [False, True])
def test_track_new_coarse_matches_oracle(lost):
    """ldso_tr_motion_hypotheses + ldso_tr_track_new_coarse = Vec4 FullSystem::trackNewCoarse (FullSystem.cc:179-386) against the oracle's
    restatement (pinned bit for bit to the reference's own member, tests/test_ref_pin.py::test_fullsystem_track_new_coarse_pinned): the 83
    hypotheses to 1e-12, same winner semantics (number of tries consumed), pose handed to the new frame to 1e-4, achievedRes to 1e-3 relative."""
    sc = tracker_scenario("small")
    o, g = make_pair(sc)
    w = sc["win"]; F = w.F
    w2c = w.truth["w2c"]
    lastF, slast, sprelast = w2c[F - 1], w2c[F - 1], w2c[F - 2]
    if lost:
        ang = 0.06
        Rz = np.array([[np.cos(ang), -np.sin(ang), 0, 0], [np.sin(ang), np.cos(ang), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
        sprelast = Rz @ slast
    ho, hg = o.motion_hypotheses(sprelast, slast, lastF), g.motion_hypotheses(sprelast, slast, lastF)
    assert ho.shape == hg.shape == (83, 3, 4) and np.abs(ho - hg).max() < 1e-12
    assert len(g.motion_hypotheses(sprelast, slast, lastF, poses_valid=False)) == 1
    rmse0 = np.array([100.0] * 5) if not lost else np.array([0.05] * 5)
    a = o.track_new_coarse(sprelast, slast, lastF, sc["new_aff"], rmse0)
    b = g.track_new_coarse(sprelast, slast, lastF, sc["new_aff"], rmse0)
    assert a["good"] == b["good"] == 1 and a["tries"] == b["tries"] == (1 if not lost else 83)
    assert np.abs(a["w2c"] - b["w2c"]).max() < 1e-4
    assert np.abs(a["aff"] - b["aff"]).max() < 1e-3 * max(1.0, np.abs(a["aff"]).max())
    fin = np.isfinite(a["lastCoarseRMSE"])
    assert np.array_equal(fin, np.isfinite(b["lastCoarseRMSE"]))
    assert np.abs(a["lastCoarseRMSE"][fin] - b["lastCoarseRMSE"][fin]).max() <= 1e-3 * np.abs(a["lastCoarseRMSE"][fin]).max()
    assert np.abs(a["result"] - b["result"]).max() <= 1e-3 * max(1.0, np.abs(a["result"]).max())
