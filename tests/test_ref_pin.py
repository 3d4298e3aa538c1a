"""The oracle (oracle/*.cc, the CPU restatement every GPU parity test is checked against) pinned to REFERENCE-COMPILED code:
oracle/_ref/libldso_ref.so is built from the reference's own hot-path translation units, unmodified, where they lie under
/root/reference (oracle/Makefile `ref`, oracle/ref_driver.cc), against a header shim for the absent third-party headers
(oracle/ref_shim: what it replaces is stated in ref_shim/Eigen/Core).  On identical windows the two must agree BIT FOR BIT on
everything the reference's own source spells out: per-residual Jacobians / energies / states (Residuals.cc), the hand-written SSE
accumulators (MatrixAccumulators.h, AccumulatedTopHessian.cc, AccumulatedSCHessian.cc), the per-point Schur quantities, the stitched
system, the step and the back-substitution (EnergyFunctional.cc), the pair precalc (FrameFramePrecalc.cc), fixLinearizationF and the
marginalisation (EnergyFunctional.cc:72-222), makeImages (FrameHessian.cc:44-113).
Runs wherever the library exists or can be built (this container); skipped otherwise."""
import copy

import numpy as np
import pytest

from conftest import get_window
from ldso_amd import synth
from oracle import pyoracle as po, pyref as pr

pytestmark = pytest.mark.skipif(not pr.available(), reason="oracle/_ref/libldso_ref.so missing and /root/reference not present to build it")


def _same(a, b, what):
    assert np.array_equal(np.asarray(a), np.asarray(b)), f"{what}: max |diff| {np.abs(np.asarray(a, float) - np.asarray(b, float)).max()}"


def _close(a, b, tol, what):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert np.abs(a - b).max() <= tol * np.abs(b).max(), f"{what}: max |diff| {np.abs(a - b).max()} of {np.abs(b).max()}"


def _stage(win, iteration=0, x_tol=0.0):
    o, r = po.OracleWindow(win), pr.RefWindow(win)
    _same(o.get_precalc(), r.get_precalc(), "FrameFramePrecalc::Set")
    for a, b, n in zip(o.get_adjoints(), r.get_adjoints(), ("adHost", "adTarget", "adHTdeltaF")):
        _same(a, b, n)
    o.collect_active(); r.collect_active()
    Eo, Er = o.linearize_all(False), r.linearize_all()
    assert Eo == Er
    ro, rr = o.get_residuals(), r.get_residuals()
    for k in ("state_NewState", "state_NewEnergy", "state_NewEnergyWithOutlier"):
        _same(ro["out"][k], rr["out"][k], k)
    lin = win.residuals["is_linearized"].astype(bool)
    ok = (ro["out"]["state_NewState"] != 1) & ~lin
    _same(ro["out"]["centerProjectedTo"][ok], rr["out"]["centerProjectedTo"][ok], "centerProjectedTo")      # never written on an early OOB return
    for k in ro["J"].dtype.names:
        _same(ro["J"][k][ok], rr["J"][k][ok], "J." + k)
    o.apply_res(); r.apply_res()
    ro, rr = o.get_residuals(), r.get_residuals()
    for k in ("state_state", "is_active"):
        _same(ro[k], rr[k], k)
    _same(ro["out"]["JpJdF"], rr["out"]["JpJdF"], "JpJdF (takeData)")
    o.backup_state(); o.solve_system(iteration); r.solve_system(iteration)
    ao, ar = o.get_accumulators(), r.get_accumulators()
    for k in ao:
        _same(ao[k], ar[k], "accumulator " + k)
    pto, _ = o.get_points(); ptr, _ = r.get_points()
    for k in pto.dtype.names:
        if k != "step":
            _same(pto[k], ptr[k], "point." + k)
    so, sr = o.get_system(), r.get_system()
    for k in ("lastHS", "lastbS"):
        _same(so[k], sr[k], k)
    if x_tol == 0.0:
        _same(so["x"], sr["x"], "x")
        _same(pto["step"], ptr["step"], "point step")
    else:      # orthogonalize(): projector built from an SVD and three dense products whose association differs between the two sides
        assert np.abs(so["x"] - sr["x"]).max() <= x_tol * np.abs(so["x"]).max()
        assert np.abs(pto["step"] - ptr["step"]).max() <= 1e-6 * np.abs(pto["step"]).max()
    assert o.counts() == r.counts()
    fo, fr = o.get_frames(), r.get_frames()
    if x_tol == 0.0:
        _same(fo["step"], fr["step"], "frame step"); _same(fo["calib_step"], fr["calib_step"], "calib step")
    _same(fo["frames"]["prior"], fr["frames"]["prior"], "FrameHessian::getPrior")
    return o, r


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_stage_bit_exact(name):
    _stage(get_window(name))


def test_stage_bit_exact_orthogonalized_iteration(small):
    """iteration >= 2: x is projected off the gauge nullspaces (EnergyFunctional::orthogonalize, EF.cc:685-717)."""
    _stage(small, iteration=2, x_tol=1e-10)


def test_stage_bit_exact_with_prior_and_linearized_residuals(small):
    """H_M / b_M present and a third of the residuals linearised: addPoint<1> (res_toZeroF + J delta), H_L, b_M + H_M delta."""
    w = po.make_mixed_window(synth.add_synthetic_prior(copy.deepcopy(small)))
    assert 0 < w.residuals["is_linearized"].sum() < w.R
    o, r = _stage(w)
    assert r.counts()[1] > 0


def test_lm_energies_pinned(small):
    """calcMEnergyF / calcLEnergyF_MT (EnergyFunctional.cc:353-378, 627-682: hand-written SSE over the linearised residuals)."""
    w = po.make_mixed_window(synth.add_synthetic_prior(copy.deepcopy(small)))
    (mo, lo), (mr, lr) = po.OracleWindow(w).calc_lm_energies(), pr.RefWindow(w).calc_lm_energies()
    assert mo == mr and abs(lo - lr) <= 1e-9 * abs(lr) and lr > 0


def test_full_size_c3_bit_exact():
    """BASELINE configs[2] (7 KF x 2000 pt, 640x480) with the marginalisation prior: the bench window."""
    _stage(synth.add_synthetic_prior(copy.deepcopy(get_window("C3"))))


@pytest.mark.parametrize("F,P,iteration", [(9, 500, 0), (12, 600, 0), (12, 600, 2)])
def test_stage_bit_exact_two_slot_groups(F, P, iteration):
    """F > 8: two slot groups, 144-column Schur rows, the (8F+4)-wide system the 100 x 100 factorisation serves (C5's code path)
    - AccumulatedTopHessian.cc:193-255 / AccumulatedSCHessian.cc:53-119 stitched over nframes^2 > 64 pairs, EF.cc:240-351 at n = 76 / 100;
    with the marginalisation prior, at iteration 0 and at an orthogonalised iteration."""
    w = synth.add_synthetic_prior(synth.make_config("small", F=F, P=P))
    assert w.F == F
    _stage(w, iteration=iteration, x_tol=1e-10 if iteration >= 2 else 0.0)


def test_full_size_c4_bit_exact():
    """BASELINE configs[3]'s window (7 KF x 3000 pt, 1232 x 368, KITTI intrinsics) with the prior."""
    _stage(synth.add_synthetic_prior(copy.deepcopy(get_window("C4"))))


def test_marginalization_bit_exact(small):
    """flagPointsForRemoval's relinearise + fixLinearizationF (Residuals.cc:216-242), marginalizePointsF (EF.cc:165-222, addPoint<2>)
    and marginalizeFrame (EF.cc:72-151) after three GN iterations of the oracle; the reference graph is rebuilt from the oracle's
    post-optimize state, the host policy (which points go) is taken from the oracle."""
    win = synth.add_synthetic_prior(copy.deepcopy(small))
    o = po.OracleWindow(win); o.set_force_all_iterations(True)
    o.optimize(3)
    ex, fo = o.export_window(), o.get_frames()
    w2 = copy.deepcopy(win)
    w2.points, w2.residuals, w2.lin_J, w2.lin_res_toZeroF = ex["points"], ex["residuals"], ex["lin_J"], ex["lin_res_toZeroF"]
    w2.frames = fo["frames"]
    w2.calib = w2.calib.copy(); w2.calib["value"] = fo["calib_value"]
    assert np.array_equal(ex["orig_point"], np.arange(win.P))
    o2, r = o, pr.RefWindow(w2)                       # the oracle keeps going; the reference graph holds the same state
    o2.flag_frame(0); o2.flag_points_for_removal()
    _, status = o2.get_points()
    assert (status % 100 == 3).sum() > 10
    r.flag_points(status)
    ro, rr = o2.get_residuals(), r.get_residuals()
    ro = {k: (v[ex["orig_res"]] if v is not None else None) for k, v in ro.items()}      # oracle flat order -> exported order
    marg = np.isin(w2.residuals["point"], np.nonzero(status % 100 == 3)[0])
    act = marg & (rr["is_active"] != 0)
    assert act.sum() > 10
    _same(ro["res_toZeroF"][act], rr["res_toZeroF"][act], "fixLinearizationF res_toZeroF")
    _same(ro["is_active"][marg], rr["is_active"][marg], "is_active of re-linearised residuals")
    o2.drop_points(); r.drop_points()
    o2.marginalize_points(); r.marginalize_points()
    # fp64 from here on: M - Msc and HM += margWeightFac * H are evaluated through the shim's / the oracle's own matrix operators,
    # whose association of the scalar factor differs by one rounding
    for a, b, n in zip(o2.get_prior(), r.get_prior(), ("HM after marginalizePointsF", "bM after marginalizePointsF")):
        _close(a, b, 1e-15, n)
    o2.marginalize_frame(0); r.marginalize_frame(0)
    (HMo, bMo), (HMr, bMr) = o2.get_prior(), r.get_prior()
    assert HMo.shape == HMr.shape == (8 * (win.F - 1) + 4,) * 2
    _close(HMo, HMr, 1e-12, "HM after marginalizeFrame"); _close(bMo, bMr, 1e-12, "bM after marginalizeFrame")
    _close(o2.get_precalc(), r.get_precalc(), 1e-7, "precalc after marginalizeFrame")      # the reference graph got its poses through a [R|t] round trip


@pytest.mark.parametrize("w,h,levels", [(160, 128, 3), (640, 480, 4)])
def test_make_images_bit_exact(w, h, levels):
    rng = np.random.default_rng(3)
    color = (rng.random((h, w)) * 255).astype(np.float32)
    a, b = po.make_images(color, levels), pr.make_images(color, levels)
    for l in range(levels):
        # the border rows of the gradient channels are never written by makeImages (memset of 3*w*h BYTES, FrameHessian.cc:50): compare the interior
        _same(a[l][1:-1, :, :], b[l][1:-1, :, :], f"makeImages level {l}")
        _same(a[l][:, :, 0], b[l][:, :, 0], f"makeImages intensity level {l}")


@pytest.mark.parametrize("cfg,levels", [("small", None), ("C3", 5)])
def test_coarse_tracker_bit_exact(cfg, levels):
    """CoarseTracker.cc compiled unmodified: makeK, makeCoarseDepthL0 (point clouds of every level), calcRes (all warped buffers),
    calcGSSSE (hand-written SSE accumulation) and the whole trackNewestCoarse LM loop."""
    from tracker_common import tracker_scenario
    sc = tracker_scenario(cfg) if levels is None else tracker_scenario(cfg, levels=levels)
    w = sc["win"]
    o = po.OracleTracker(w.w, w.h, sc["levels"], w.settings, w.calib); r = pr.RefTracker(w.w, w.h, sc["levels"], w.settings, w.calib)
    for t in (o, r):
        t.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"]); t.set_new_frame(sc["new_pyr"], 1.0)
    for a, b, n in zip(o.K(), r.K(), ("fx", "fy", "cx", "cy")):
        _same(a, b, "makeK " + n)
    for l in range(sc["levels"]):
        for a, b, n in zip(o.pc(l), r.pc(l), ("pc_u", "pc_v", "pc_idepth", "pc_color")):
            _same(a, b, f"{n}[{l}]")
    a, b = sc["new_aff"]
    T0 = np.eye(4)
    for l in range(sc["levels"]):
        (rso, no), (rsr, nr) = o.calc_res(l, T0, a, b, 20.0), r.calc_res(l, T0, a, b, 20.0)
        assert no == nr
        _same(rso, rsr, f"calcRes[{l}]")
        wo, wr = o.warped(no), r.warped(nr)
        for k in wo:
            _same(wo[k], wr[k], f"buf_warped_{k}[{l}]")
        (Ho, bo), (Hr, br) = o.calc_gs(l, T0, a, b), r.calc_gs(l, T0, a, b)
        _same(Ho, Hr, f"calcGSSSE H[{l}]"); _same(bo, br, f"calcGSSSE b[{l}]")
    to, tr = o.track(T0, a, b, sc["levels"] - 1), r.track(T0, a, b, sc["levels"] - 1)
    assert to["ok"] == tr["ok"] and to["a"] == tr["a"] and to["b"] == tr["b"]
    _same(to["T"], tr["T"], "trackNewestCoarse pose"); _same(to["flow"], tr["flow"], "lastFlowIndicators")
    assert np.array_equal(to["lastResiduals"], tr["lastResiduals"], equal_nan=True)


def test_trace_on_bit_exact():
    """ImmaturePoint::traceOn (ImmaturePoint.cc:47-310): epipolar search, Gauss-Newton refinement and interval update of 300
    immature points against a new frame - the records after the call are byte-identical."""
    win = synth.make_config('small', extra_frames=1)
    pts, _ = synth.make_immature_points(win, 60)
    KRKi, Kt, aff = synth.trace_poses(win, win.F)
    a, b = pts.copy(), pts.copy()
    ca = po.trace_on(a, win.images[win.F][0], KRKi, Kt, aff)
    cb = pr.trace_on(b, win.images[win.F][0], KRKi, Kt, aff)
    assert np.array_equal(ca, cb) and ca[0] > 100
    assert a.tobytes() == b.tobytes()
    # second round on the updated intervals (the GN refinement and the skip / bad-condition exits are reached from here)
    ca = po.trace_on(a, win.images[win.F][0], KRKi, Kt, aff); cb = pr.trace_on(b, win.images[win.F][0], KRKi, Kt, aff)
    assert np.array_equal(ca, cb) and a.tobytes() == b.tobytes()


def test_coarse_initializer_tracks_the_reference():
    """CoarseInitializer::trackFrame (CoarseInitializer.cc compiled unmodified) over a 5-frame sequence from identical point records:
    identical control flow (snapping frame, frame counters, isGood sets), pose to 1e-6, inverse depths to 1e-4.  Not bit-exact: the
    restatement associates a few float expressions of calcResAndGS differently from the shim's operators (observed 4e-9 after the
    first frame, 4e-7 after the fifth)."""
    seq = synth.make_init_sequence(160, 120, n_frames=5, fx=100.0, seed=11, levels=3)
    L = seq['levels']
    pyr0 = synth.make_images(seq['first'], L)
    ipts = synth.select_init_points(pyr0)
    o = po.OracleInitializer(160, 120, L); o.set_first(seq['K4'], pyr0, 1.0, ipts)
    r = pr.RefInitializer(160, 120, L); r.set_first(seq['K4'], pyr0, 1.0, ipts)
    snapped_seen = False
    for k in range(5):
        pyr = synth.make_images(seq['frames'][k], L)
        o.set_new_frame(pyr, 1.0)
        so, sr = o.track_frame(), r.track_frame(pyr, 1.0)
        for f in ("snapped", "snappedAt", "frameID", "ready"):
            assert int(so[f]) == int(sr[f]), (k, f)
        assert np.abs(so["thisToNext"] - sr["thisToNext"]).max() < 1e-6 and abs(so["aff_a"] - sr["aff_a"]) < 1e-6 and abs(so["aff_b"] - sr["aff_b"]) < 1e-5
        snapped_seen |= bool(sr["snapped"])
        for l in range(L):
            a, b = o.points(l), r.points(l)
            assert np.array_equal(a["isGood"], b["isGood"]) and np.array_equal(a["isGood_new"], b["isGood_new"])
            _close(a["idepth"], b["idepth"], 1e-4, f"idepth[{l}] frame {k}"); _close(a["iR"], b["iR"], 1e-4, f"iR[{l}] frame {k}")
    assert snapped_seen


# ---- the FullSystem.cc slice (round 3): the reference's own members, FullSystem.cc compiled unmodified --------------------------------
def _compare_window_state(o, r, tol_state=1e-12, what=""):
    """Everything optimize() leaves behind that the adapter writes back / the next stage reads."""
    fo, fr = o.get_frames(), r.get_frames()
    _same(fo["frames"]["frameEnergyTH"], fr["frames"]["frameEnergyTH"], what + "frameEnergyTH (setNewFrameEnergyTH)")
    for k in ("state", "state_zero", "worldToCam_evalPT"):
        assert np.abs(fo["frames"][k] - fr["frames"][k]).max() <= tol_state, what + k      # x passes through orthogonalize(): 1e-17 observed
    assert np.abs(fo["step"] - fr["step"]).max() <= tol_state and np.abs(fo["calib_value"] - fr["calib_value"]).max() <= tol_state
    assert np.abs(fo["pre_worldToCam"] - fr["pre_worldToCam"]).max() <= tol_state
    (pto, so), (ptr, sr) = o.get_points(), r.get_points()
    _same(so, sr, what + "point status")
    for k in ("idepth", "maxRelBaseline", "numGoodResiduals"):
        _same(pto[k], ptr[k], what + "point." + k)
    _close(pto["HdiF"], ptr["HdiF"], 1e-6, what + "HdiF"); _close(pto["idepth_hessian"], ptr["idepth_hessian"], 1e-6, what + "idepth_hessian")
    ro, rr = o.get_residuals(), r.get_residuals()
    for k in ("state_state", "is_active", "alive", "is_linearized"):
        _same(ro[k], rr[k], what + k)
    live = rr["alive"] != 0
    _same(ro["out"]["JpJdF"][live], rr["out"]["JpJdF"][live], what + "JpJdF")
    _same(ro["out"]["state_NewEnergy"][live], rr["out"]["state_NewEnergy"][live], what + "state_NewEnergy")
    assert o.counts() == r.counts()
    return rr


@pytest.mark.parametrize("name,iters", [("tiny", 6), ("small", 6), ("C3", 4)])
def test_fullsystem_optimize_pinned(name, iters):
    """float FullSystem::optimize(int) (FullSystem.cc:725-864) itself - activeResiduals, linearizeAll(false) + setNewFrameEnergyTH,
    applyRes, backupState, solveSystem, doStepFromBackup incl. the `canbreak` rule, the final setEvalPT / linearizeAll(true) with residual
    removal - against the oracle's restatement: return value and every printed energy identical, flags / thresholds / inverse depths bit
    for bit, frame states to 1e-12 (x passes through orthogonalize(), see test_stage_bit_exact_orthogonalized_iteration)."""
    win = synth.add_synthetic_prior(copy.deepcopy(get_window(name))) if name == "C3" else get_window(name)
    o, r = po.OracleWindow(win), pr.RefWindow(win)
    r.fs_attach()
    rv_r, log_r = r.fs_optimize(iters)
    rv_o = o.optimize(iters); log_o = o.energy_log()
    assert rv_r == rv_o and np.isfinite(rv_r)
    # printOptRes prints the energy once before the loop and once per executed iteration (%f: six decimals); the oracle's log also holds the
    # final linearizeAll(true).  Equal length = the loop ended (canbreak && iteration >= setting_minOptIterations) at the same iteration.
    assert len(log_o) == len(log_r) + 1, (log_o, log_r)
    assert np.abs(log_o[:-1] - log_r).max() <= 1e-6 + 1e-12 * np.abs(log_r).max()
    assert not r.fs_is_lost()
    _compare_window_state(o, r)


def test_fullsystem_optimize_drops_residuals_like_the_reference(small):
    """linearizeAll(true) (FullSystem.cc:1472-1492) removes the residuals that ended OOB / OUTLIER and updates lastResiduals: a window whose
    newest frame is badly perturbed so that residuals really go (the other tests rarely drop any)."""
    win = copy.deepcopy(small)
    win.frames["state"][-1, :3] += 0.08            # a few pixels of parallax error for every residual into / out of the newest frame
    o, r = po.OracleWindow(win), pr.RefWindow(win)
    r.fs_attach()
    rv_r, log_r = r.fs_optimize(1)
    rv_o = o.optimize(1)
    assert rv_r == rv_o
    rr = _compare_window_state(o, r)
    assert 0 < (rr["alive"] == 0).sum() < win.R, "the scenario is meant to drop some residuals"


def test_fullsystem_stage_members_pinned(small):
    """The private members one by one, in the order optimize() calls them: linearizeAll(false) -> Vec3 + frameEnergyTH, applyRes_Reductor,
    backupState, solveSystem, doStepFromBackup -> canbreak, loadSateBackup, calcLEnergy / calcMEnergy."""
    win = po.make_mixed_window(synth.add_synthetic_prior(copy.deepcopy(small)))
    win.settings = win.settings.copy(); win.settings["forceAcceptStep"] = 0          # calcLEnergy / calcMEnergy return 0 when steps are forced
    o, r = po.OracleWindow(win), pr.RefWindow(win)
    r.fs_attach()
    o.collect_active(); r.fs_collect_active()
    Eo, Er = o.linearize_all(False), r.fs_linearize_all(False)
    assert Eo == Er[0] and Er[1] == 0 and Er[2] == 0
    fo, fr = o.get_frames(), r.get_frames()
    _same(fo["frames"]["frameEnergyTH"], fr["frames"]["frameEnergyTH"], "setNewFrameEnergyTH")
    assert fr["frames"]["frameEnergyTH"][-1] != win.frames["frameEnergyTH"][-1]
    (mo, lo), (lr, mr) = o.calc_lm_energies(), r.fs_calc_energies()
    assert mo == mr and abs(lo - lr) <= 1e-9 * abs(lr) and lr > 0 and mr > 0
    o.apply_res(); r.fs_apply_res()
    for it in range(3):
        o.backup_state(); r.fs_backup_state(it != 0)
        o.solve_system(it); r.fs_solve_system(it)
        so, sr = o.get_system(), r.get_system()
        _same(so["lastHS"], sr["lastHS"], "lastHS"); _same(so["lastbS"], sr["lastbS"], "lastbS")
        cb_o, cb_r = o.do_step(), r.fs_do_step(1.0)
        assert cb_o == cb_r
        fo, fr = o.get_frames(), r.get_frames()
        assert np.abs(fo["frames"]["state"] - fr["frames"]["state"]).max() <= 1e-12
        assert np.abs(fo["pre_worldToCam"] - fr["pre_worldToCam"]).max() <= 1e-12
        _close(o.get_precalc(), r.get_precalc(), 1e-6, "precalc after doStepFromBackup")
        (pto, _), (ptr, _) = o.get_points(), r.get_points()
        _close(pto["idepth"], ptr["idepth"], 1e-6, "idepth after doStepFromBackup")
        Eo, Er = o.linearize_all(False), r.fs_linearize_all(False)
        assert abs(Eo - Er[0]) <= 1e-6 * abs(Er[0])
        o.apply_res(); r.fs_apply_res()
    # loadSateBackup (FullSystem.cc:1662-1680): back to the state of the last backupState
    before = r.get_frames()["frames"]["state"].copy()
    r.fs_load_state_backup()
    after = r.get_frames()["frames"]["state"]
    assert np.abs(after - before).max() > 0


def test_fullsystem_canbreak_rule(small):
    """doStepFromBackup's convergence test (FullSystem.cc:1617-1622) on both sides of the threshold: a zero step can break, the first real
    step of a perturbed window cannot."""
    o, r = po.OracleWindow(small), pr.RefWindow(small)
    r.fs_attach()
    o.collect_active(); r.fs_collect_active()
    o.linearize_all(False); r.fs_linearize_all(False); o.apply_res(); r.fs_apply_res()
    o.backup_state(); r.fs_backup_state(False)
    assert o.do_step() and r.fs_do_step(1.0), "all steps are zero before the first solve: canbreak"
    o.solve_system(0); r.fs_solve_system(0)
    assert (not o.do_step()) and (not r.fs_do_step(1.0))


def test_fullsystem_marginalization_members_pinned(small):
    """FullSystem::flagPointsForRemoval (FullSystem.cc:1208-1270: OOB / inlier policy, re-linearise + fixLinearizationF) and
    FullSystem::marginalizeFrame (:602-640) - the members themselves - against the oracle's restatement of both."""
    win = synth.add_synthetic_prior(copy.deepcopy(small))
    o, r = po.OracleWindow(win), pr.RefWindow(win)
    r.fs_attach()
    o.optimize(3); r.fs_optimize(3)
    o.flag_frame(0); r.fs_flag_frame(0)
    o.flag_points_for_removal(); r.fs_flag_points_for_removal()
    (_, so), (_, sr) = o.get_points(), r.get_points()
    _same(so % 100, sr % 100, "point status after flagPointsForRemoval")
    assert (sr % 100 == 3).sum() > 10
    ro, rr = o.get_residuals(), r.get_residuals()
    marg = np.isin(win.residuals["point"], np.nonzero(sr % 100 == 3)[0]) & (rr["alive"] != 0)
    act = marg & (rr["is_active"] != 0)
    assert act.sum() > 10
    _close(ro["res_toZeroF"][act], rr["res_toZeroF"][act], 1e-5, "fixLinearizationF res_toZeroF")
    _same(ro["is_active"][marg], rr["is_active"][marg], "is_active of re-linearised residuals")
    o.drop_points(); r.drop_points()
    o.marginalize_points(); r.marginalize_points()
    for a, b, n in zip(o.get_prior(), r.get_prior(), ("HM after marginalizePointsF", "bM after marginalizePointsF")):
        _close(a, b, 1e-9, n)
    o.marginalize_frame(0); r.fs_marginalize_frame(0)
    (HMo, bMo), (HMr, bMr) = o.get_prior(), r.get_prior()
    assert HMo.shape == HMr.shape == (8 * (win.F - 1) + 4,) * 2
    _close(HMo, HMr, 1e-9, "HM after marginalizeFrame"); _close(bMo, bMr, 1e-9, "bM after marginalizeFrame")
    assert o.num_frames() == r.num_frames() == win.F - 1
    ro, rr = o.get_residuals(), r.get_residuals()
    _same(ro["alive"], rr["alive"], "residuals dropped with the frame")


def _traced_points(win, per_frame):
    """immature points of the window's key frames, traced against the two extra frames so that they carry depth intervals"""
    pts, true_id = synth.make_immature_points(win, per_frame)
    for fidx in (win.F, win.F + 1):
        KRKi, Kt, aff = synth.trace_poses(win, fidx)
        po.trace_on(pts, win.images[fidx][0], KRKi, Kt, aff)
    keep = np.isfinite(pts["idepth_max"]) & (pts["lastTraceStatus"] != 1)
    return pts[keep].copy(), true_id[keep]


def test_immature_points_of_the_harness_are_released_cleanly():
    """ref_fs_build_immature / ref_fs_free_immature (oracle/ref_driver.cc: the ImmaturePoint objects the adapter's activation test hands to GpuBackend::activatePoints) with points
    that were never activated - the case in which the harness once let Feature::ReleaseImmature (Feature.cc:26-31) destroy the feature it was running in.  Nothing to compare:
    the test is for scripts/asan_check.sh (AddressSanitizer build of this library); without it, it must simply not crash, many times over."""
    import ctypes as C
    win = synth.make_config("small", extra_frames=2)
    pts, _ = _traced_points(win, 40)
    r = pr.RefWindow(win); r.fs_attach()
    r.L.ref_fs_build_immature.restype = C.c_void_p
    rec = np.ascontiguousarray(pts)
    for _ in range(20):
        vec = C.c_void_p(r.L.ref_fs_build_immature(r.h, C.c_int(len(rec)), rec.ctypes.data_as(C.c_void_p), C.c_int(1), C.c_float(100.0), C.c_int(3)))
        assert vec.value
        r.L.ref_fs_free_immature(vec)


@pytest.mark.parametrize("name,per_frame", [("small", 120), ("C3", 60)])
def test_fullsystem_optimize_immature_point_pinned(name, per_frame):
    """shared_ptr<PointHessian> FullSystem::optimizeImmaturePoint (FullSystem.cc:892-1010, the member; ImmaturePoint::linearizeResidual from
    ImmaturePoint.cc compiled unmodified) against the oracle's restatement (oracle/trace.cc orc_activate_points) on the same records and
    the same pair transforms: verdict, per-target residual states and the activated inverse depth bit for bit; incl. rejected candidates."""
    win = synth.make_config(name, extra_frames=2)
    pts, _ = _traced_points(win, per_frame)
    pts = pts.copy()
    pts["idepth_min"][0] = np.nan                                   # non-finite start: return 0 (:924-926)
    pts["idepth_min"][1] = 50.0; pts["idepth_max"][1] = 60.0         # absurdly close: every residual OOB
    pts["u"][2] = 2.0; pts["v"][2] = 2.0                             # the pattern leaves the target images
    r = pr.RefWindow(win); r.fs_attach()
    pairs = r.get_pair_rt()
    K4 = np.asarray([np.float32(50.0 * v) for v in win.calib["value"]], np.float32)
    a = po.activate_points(pts, [win.images[f][0] for f in range(win.F)], K4, pairs, win.w, win.h)
    b = r.fs_activate_points(pts)
    _same(a["ok"], b["ok"], "verdict"); _same(a["res_state"], b["res_state"], "temporary residual states"); _same(a["numGoodRes"], b["numGoodRes"], "numGoodRes")
    ok = b["ok"] == 1
    assert 0.3 < ok.mean() < 1.0 and b["ok"][0] == 0
    _same(a["idepth"][ok].view(np.uint32), b["idepth"][ok].view(np.uint32), "activated inverse depth")


def _track_new_coarse_scenario(cfg="small", levels=None, lost=False):
    """A tracker with reference + new frame and the three poses trackNewCoarse builds its motion hypotheses from: the true poses of the two
    frames before the new one (constant-motion guess = a good start) or, lost = True, wrong ones so that several tries are consumed."""
    from tracker_common import tracker_scenario
    sc = tracker_scenario(cfg) if levels is None else tracker_scenario(cfg, levels=levels)
    w = sc["win"]; F = w.F
    w2c = w.truth["w2c"]
    lastF, slast, sprelast = w2c[F - 1], w2c[F - 1], w2c[F - 2]
    if lost:      # the two previous frames suggest a rotation that did not happen: the first tries start far off
        ang = 0.06
        Rz = np.array([[np.cos(ang), -np.sin(ang), 0, 0], [np.sin(ang), np.cos(ang), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
        sprelast = Rz @ slast
    return sc, w, sprelast, slast, lastF


@pytest.mark.parametrize("lost", [False, True])
def test_fullsystem_track_new_coarse_pinned(lost):
    """Vec4 FullSystem::trackNewCoarse (FullSystem.cc:179-386, the member): the 83 motion hypotheses (:189-309), the try loop with the
    growing achievedRes abort thresholds and the reTrackThreshold exit (:319-356), the pose / affine hand-over (:367-379) - against the
    oracle's restatement (tracker_capi.inc): bit for bit (both sides compute poses through oracle/lie.h)."""
    sc, w, sprelast, slast, lastF = _track_new_coarse_scenario(lost=lost)
    o = po.OracleTracker(w.w, w.h, sc["levels"], w.settings, w.calib); r = pr.RefTracker(w.w, w.h, sc["levels"], w.settings, w.calib)
    for t in (o, r):
        t.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"]); t.set_new_frame(sc["new_pyr"], 1.0)
    rmse0 = np.array([100.0] * 5) if not lost else np.array([0.05] * 5)           # an unreachable lastCoarseRMSE[0] keeps the loop going (:355)
    a = o.track_new_coarse(sprelast, slast, lastF, sc["new_aff"], rmse0)
    b = r.track_new_coarse(sprelast, slast, lastF, sc["new_aff"], rmse0)
    _same(a["result"], b["result"], "Vec4 result"); _same(a["w2c"], b["w2c"], "pose handed to the new frame"); _same(a["aff"], b["aff"], "aff_g2l")
    assert np.array_equal(a["lastCoarseRMSE"], b["lastCoarseRMSE"], equal_nan=True)
    assert a["tries"] == (1 if not lost else 83) and a["good"] == 1
    T_true = w.truth["w2c"][w.F]
    assert np.abs(b["w2c"] - T_true[:3]).max() < 5e-3                       # and it is the right pose
    assert len(o.motion_hypotheses(sprelast, slast, lastF)) == 83 and len(o.motion_hypotheses(sprelast, slast, lastF, poses_valid=False)) == 1


def nonfinite_windows(base):
    """The reference's failure semantics on four corrupted copies of a window: what it does is 'propagate non-finite numbers'."""
    def imgs(w):
        w = copy.deepcopy(w); w.images = [[l.copy() for l in im] for im in w.images]; return w
    a = imgs(base); a.images[2][0][60:120, 80:200, 0] = np.nan          # NaN irradiance: hitColor[0] not finite -> OOB (Residuals.cc:142-145)
    b = imgs(base); b.images[2][0][60:120, 80:200, 0] = np.inf          # Inf likewise
    c = imgs(base); c.images[2][0][60:120, 80:200, 1] = np.nan          # NaN gradient: finite residual, NaN weight -> NaN energy -> isLost (FullSystem.cc:853-857)
    d = copy.deepcopy(base); d.frames = d.frames.copy(); d.frames["state"][2, 0] = np.nan      # NaN pose: NaN precalc, every projection fails its bounds test
    return {"nan_intensity": a, "inf_intensity": b, "nan_gradient": c, "nan_state": d}


@pytest.mark.parametrize("case", ["nan_intensity", "inf_intensity", "nan_gradient", "nan_state"])
def test_nonfinite_inputs_pinned(small, case):
    """NaN / Inf pixels and a NaN frame state through the reference's own optimize() (Release semantics: asserts off) and the oracle's:
    same residual states / removal, same energies (NaN where the reference's are NaN), same isLost."""
    win = nonfinite_windows(small)[case]
    o, r = po.OracleWindow(win), pr.RefWindow(win)
    r.fs_attach()
    rv_r, log_r = r.fs_optimize(3)
    rv_o = o.optimize(3); log_o = o.energy_log()
    assert (rv_r == rv_o) or (np.isnan(rv_r) and np.isnan(rv_o))
    assert r.fs_is_lost() == o.is_lost() == (case == "nan_gradient")
    assert len(log_o) == len(log_r) + 1
    assert np.allclose(log_o[:-1], log_r, rtol=1e-12, atol=1e-6, equal_nan=True)
    ro, rr = o.get_residuals(), r.get_residuals()
    for k in ("state_state", "is_active", "alive"):
        _same(ro[k], rr[k], k)
    fo, fr = o.get_frames(), r.get_frames()
    assert np.array_equal(np.isnan(fo["frames"]["state"]), np.isnan(fr["frames"]["state"]))
    (pto, _), (ptr, _) = o.get_points(), r.get_points()
    assert np.array_equal(pto["idepth"], ptr["idepth"], equal_nan=True)
    if case in ("nan_intensity", "inf_intensity"):
        assert 0 < (rr["alive"] == 0).sum() < win.R and np.isfinite(rv_r) and not np.isnan(fr["frames"]["state"]).any()
    if case == "nan_state":
        assert (rr["alive"] == 0).all() and np.isinf(rv_r)          # everything ends OOB; sqrtf(E / (8 * 0)) - and the reference does NOT flag isLost
