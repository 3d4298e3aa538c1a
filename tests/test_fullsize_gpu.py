"""Full-size GPU parity of exactly what bench.py times (BASELINE configs C3 / C4 with the marginalisation prior, the fused fast
path `k_reduce_solve -> k_linearize`, 10 forced iterations) and the C5 stress flow (12 KF x 8000 pt: stage-wise pass, fast-path
iterations, marginalizePointsF + marginalizeFrame of the oldest frame, 10 more iterations at F = 11), all against the oracle.

Reference loop being reproduced: src/frontend/FullSystem.cc:777-831 (optimize), :1208-1270 (flagPointsForRemoval),
src/internal/OptimizationBackend/EnergyFunctional.cc:72-151 (marginalizeFrame), :165-222 (marginalizePointsF).
Tolerances (north_star): 1e-4 relative on energies and Hessian entries (entries relative to the 4x4 block maximum); states exact."""
import copy

import numpy as np
import pytest

from conftest import observe, rel, blockrel, get_window
from ldso_amd import synth, binding
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _one_stream():
    import torch
    ts = torch.cuda.Stream()
    torch.cuda.set_stream(ts)
    return torch, ts.cuda_stream


def _fast_system(torch, g, win):
    """HFinal (lower triangle) | bFinal | energy of the fast path at the handle's applied state: the accumulator that
    k_linearize initialised (H_M, priors) and k_reduce's fp64 atomics completed — the very buffer k_reduce_solve factorises."""
    n = 8 * win.F + 4
    buf = torch.zeros(g.gn_reduce_doubles(), dtype=torch.float64, device="cuda")
    g.gn_reduce_local(buf.data_ptr(), 1e-1)
    g.sync(); torch.cuda.synchronize()
    b = buf.cpu().numpy()
    return np.tril(b[:n * n].reshape(n, n)), b[n * n:n * n + n].copy(), float(b[n * n + n])


def _oracle_loop(o, first, its):
    """bench.py's step on the oracle: solveSystem + doStepFromBackup + linearizeAll(false) + applyRes, forced accept."""
    E = []
    for it in range(first, first + its):
        o.backup_state(); o.solve_system(it); o.do_step()
        E.append(o.linearize_all(False)); o.apply_res()
    return np.array(E)


def _transplant(win, g):
    """A window whose evaluation state is the GPU handle's applied state (frames, calibration, inverse depths, residual states)."""
    w2 = copy.deepcopy(win)
    fg = g.get_frames()
    w2.frames = fg["frames"].copy()
    w2.calib = w2.calib.copy(); w2.calib["value"] = fg["calib_value"]
    pt = g.get_points()
    w2.points = w2.points.copy()
    w2.points["idepth"] = pt["idepth"]; w2.points["idepth_zero"] = pt["idepth"]      # setIdepthZero on every step (FullSystem.cc:1598-1602)
    rg = g.get_residuals()
    w2.residuals = w2.residuals.copy()
    w2.residuals["state_state"] = rg["state_state"]; w2.residuals["is_active"] = rg["is_active"]
    w2.residuals["state_energy"] = rg["out"]["state_NewEnergy"]
    return w2


def gauge_basis(frames):
    """The 7 gauge directions (6 pose + scale) in the 8-per-frame state, as FullSystem::getNullspaces assembles them
    (FullSystem.cc:1711-1760: per-frame nullspaces_pose / nullspaces_scale with SCALE_XI_*_INVERSE)."""
    F = len(frames)
    N = np.zeros((8 * F, 7))
    for f in range(F):
        P = frames["nullspaces_pose"][f].reshape(6, 6)
        for i in range(6):
            N[8 * f:8 * f + 6, i] = P[:, i]
        N[8 * f:8 * f + 6, 6] = frames["nullspaces_scale"][f]
        N[8 * f:8 * f + 3, :] *= 2.0                                                     # SCALE_XI_TRANS_INVERSE
    return N


@pytest.mark.parametrize("name", ["C3", "C4"])
def test_timed_configuration_parity(name):
    """bench.py's timed workload: config + synthetic prior, enqueue_gn (fused k_reduce_solve -> k_linearize), 10 iterations."""
    torch, st = _one_stream()
    win = synth.add_synthetic_prior(copy.deepcopy(get_window(name)))
    n = 8 * win.F + 4
    its = 10
    o = po.OracleWindow(win)
    g = binding.BA.from_window(win, stream=st)
    o.collect_active(); g.collect_active()
    Eo0, Eg0 = o.linearize_all(False), g.linearize_all(False)
    o.apply_res(); g.apply_res()
    assert abs(Eo0 - Eg0) <= 1e-6 * Eo0
    ro, rg = o.get_residuals(False), g.get_residuals()
    assert np.array_equal(ro["state_state"], rg["state_state"]) and np.array_equal(ro["is_active"], rg["is_active"])     # iteration 0: exact
    assert rel(rg["out"]["state_NewEnergy"], ro["out"]["state_NewEnergy"]) < 1e-5                                        # per-residual energies

    # iteration 0: the system the fast path factorises
    Hg, bg, Eg = _fast_system(torch, g, win)
    o.backup_state(); o.solve_system(0)
    so = o.get_system()
    assert blockrel(Hg, np.tril(so["HFinal"]), 4) < TOL and rel(Hg, np.tril(so["HFinal"])) < TOL
    assert rel(bg, so["bFinal"]) < TOL
    assert abs(Eg - Eo0) <= 1e-6 * Eo0
    # the step of iteration 0 (step-wise entry point, same solve_core): x is gauge-limited entry by entry, but it solves the ORACLE's
    # system, achieves the oracle's model decrease and differs from the oracle's x only by 1e-5 in the energy norm
    g0 = binding.BA.from_window(win, stream=st)
    g0.collect_active(); g0.linearize_all(False); g0.apply_res(); g0.backup_state(); g0.solve_system(0)
    xg, xo, Ho, bo = g0.get_system()["x"], so["x"], so["HFinal"], so["bFinal"]
    assert np.linalg.norm(Ho @ xg - bo) / np.linalg.norm(bo) < 1e-6
    mo, mg = 2 * bo @ xo - xo @ Ho @ xo, 2 * bo @ xg - xg @ Ho @ xg
    assert abs(mo - mg) <= 1e-9 * abs(mo)
    assert np.sqrt(abs((xg - xo) @ Ho @ (xg - xo)) / abs(xo @ Ho @ xo)) < 1e-5

    # the timed call itself: 10 iterations in one enqueue (2 launches each, no host sync)
    g.enqueue_gn(0, its); g.sync()
    o.do_step(); Eo = [o.linearize_all(False)]; o.apply_res()
    Eo = np.concatenate([Eo, _oracle_loop(o, 1, its - 1)])
    H9, b9, E9 = _fast_system(torch, g, win)
    assert abs(E9 - Eo[-1]) <= TOL * Eo[-1], (E9, Eo)
    fo, fg = o.get_frames(), g.get_frames()
    # states: the reduced system is ill-conditioned along the scale gauge (cond 1e5 after scaling), so fp32-level differences of
    # H / b (1e-9 relative here) move x by 1e-4..1e-3 along the weakest eigenvector; the oracle's own FMA / 6-thread variants
    # differ from its portable single-thread build by 5e-4 (C3) .. 3e-3 (C4) after 10 iterations.  Off the gauge directions
    # (projector of EnergyFunctional::orthogonalize) the states agree to 2e-4.
    xs_o, xs_g = fo["frames"]["state"][:, :8].reshape(-1), fg["frames"]["state"][:, :8].reshape(-1)
    assert np.abs(xs_g - xs_o).max() < 5e-3 * np.abs(xs_o).max()
    Q, _ = np.linalg.qr(gauge_basis(fo["frames"]))
    d = xs_g - xs_o
    assert np.abs(d - Q @ (Q.T @ d)).max() < 2e-4 * np.abs(xs_o).max()
    assert rel(g.get_points()["idepth"], o.get_points()[0]["idepth"]) < 2e-4
    assert rel(fg["frames"]["frameEnergyTH"], fo["frames"]["frameEnergyTH"]) < 1e-3

    # per-iteration energies of the same fast path: optimize(force_all) logs one energy per iteration
    o2 = po.OracleWindow(win); o2.set_force_all_iterations(True)
    g2 = binding.BA.from_window(win, stream=st)
    rmo = o2.optimize(its); rmg, done = g2.optimize(its, force_all=True)
    eo, eg = o2.energy_log(), g2.get_energy_log()
    assert done == its and len(eo) == len(eg) == its + 2
    assert np.all(np.abs(eg - eo) <= TOL * np.abs(eo)), (eo, eg)
    assert abs(rmo - rmg) <= TOL * rmo
    assert np.all(np.abs(eo[1:its + 1] - Eo) <= 1e-9 * Eo)                               # the oracle loop above is optimize()'s loop

    # the system after the 10 iterations, at the GPU's own iterate (removes the gauge drift of the states from the comparison)
    w9 = _transplant(win, g)
    o9 = po.OracleWindow(w9); g9 = binding.BA.from_window(w9, stream=st)
    o9.collect_active(reset_oob=False)                                                   # continue: OOB residuals stay out, as inside the loop
    Eo9, Eg9 = o9.linearize_all(False), g9.linearize_all(False)
    o9.apply_res(); g9.apply_res()
    assert abs(Eo9 - Eg9) <= 1e-6 * Eo9 and abs(Eg9 - E9) <= 1e-6 * E9                   # the fast path's energy at that iterate, re-derived
    assert np.array_equal(o9.get_residuals(False)["state_state"], g9.get_residuals()["state_state"])
    H9g, b9g, _ = _fast_system(torch, g9, w9)
    o9.backup_state(); o9.solve_system(its)
    s9 = o9.get_system()
    assert blockrel(H9g, np.tril(s9["HFinal"]), 4) < TOL and rel(b9g, s9["bFinal"]) < TOL
    assert blockrel(H9, np.tril(s9["HFinal"]), 4) < TOL and rel(b9, s9["bFinal"]) < TOL  # ... and as left behind by the timed run itself


def test_c5_end_to_end():
    """BASELINE configs[4]: 12 KF x 8000 pt (R = 88000): stage-wise pass, 4 fast-path iterations, marginalise the oldest frame
    (points, then the frame), rebuild at F = 11 with the device's own prior, 10 more iterations."""
    torch, st = _one_stream()
    win = get_window("C5")
    assert win.F == 12 and win.P == 8000 and win.R == 88000
    o = po.OracleWindow(win); g = binding.BA.from_window(win, stream=st)
    o.collect_active(); g.collect_active()
    Eo, Eg = o.linearize_all(False), g.linearize_all(False)
    assert abs(Eo - Eg) <= 1e-6 * Eo
    assert np.array_equal(o.get_residuals(False)["out"]["state_NewState"], g.get_residuals()["out"]["state_NewState"])
    o.apply_res(); g.apply_res(); o.backup_state(); g.backup_state(); o.solve_system(0); g.solve_system(0)
    so, sg = o.get_system(), g.get_system()
    for k in ("HA", "Hsc", "HFinal"):
        assert blockrel(sg[k], so[k], 4) < TOL, k
    assert rel(sg["bFinal"], so["bFinal"]) < TOL
    assert np.linalg.norm(sg["HFinal"] @ sg["x"] - sg["bFinal"]) / np.linalg.norm(sg["bFinal"]) < 1e-8
    Hf, bf, _ = _fast_system(torch, g, win)                                              # fast-path accumulator of the same state
    assert blockrel(Hf, np.tril(so["HFinal"]), 4) < TOL and rel(bf, so["bFinal"]) < TOL

    # 4 fast-path iterations
    o = po.OracleWindow(win); o.set_force_all_iterations(True)
    g = binding.BA.from_window(win, stream=st)
    rmo = o.optimize(4); rmg, done = g.optimize(4, force_all=True)
    eo, eg = o.energy_log(), g.get_energy_log()
    assert done == 4 and np.all(np.abs(eg - eo) <= TOL * np.abs(eo)), (eo, eg)
    assert abs(rmo - rmg) <= TOL * rmo

    # marginalise the oldest frame; the device starts from the oracle's post-optimize state (identical applied state on both sides)
    ex, fo = o.export_window(), o.get_frames()
    assert ex["F"] == 12 and len(ex["points"]) == win.P
    w2 = copy.deepcopy(win)
    w2.points, w2.residuals, w2.lin_J, w2.lin_res_toZeroF = ex["points"], ex["residuals"], ex["lin_J"], ex["lin_res_toZeroF"]
    w2.frames = fo["frames"]
    w2.calib = w2.calib.copy(); w2.calib["value"] = fo["calib_value"]
    g = binding.BA.from_window(w2, stream=st)
    o.flag_frame(0); o.flag_points_for_removal()
    _, status = o.get_points()
    flags = (status == 3).astype(np.int32)
    assert 300 < flags.sum() < win.P
    o.drop_points(); o.marginalize_points()
    HMo, bMo = o.get_prior()
    HMg, bMg = g.marginalize_points(flags)
    assert blockrel(HMg, HMo, 4) < TOL and rel(bMg, bMo) < TOL
    o.marginalize_frame(0)
    HM2o, bM2o = o.get_prior()
    HM2g, bM2g = g.marginalize_frame(0)
    assert HM2g.shape == (92, 92)
    # marginalizeFrame is fp64 on both sides; what separates HM2g from HM2o is the fp32 difference of their INPUTS (HMg vs HMo, within TOL above) pushed through
    # the inverse of the frame's 8 x 8 block.  Measured apart (round 6; until round 5 this was one comparison at 5 x TOL):
    #  (i) the device on the ORACLE's prior is the oracle's result to fp64 rounding,
    g.set_prior(HMo, bMo)
    HM2i, bM2i = g.marginalize_frame(0)
    observe("c5_marginalize_frame_same_input_HM", blockrel(HM2i, HM2o, 4), 1e-9); observe("c5_marginalize_frame_same_input_bM", rel(bM2i, bM2o), 1e-9)
    #  (ii) the oracle's own arithmetic on the DEVICE's prior is the device's result (same bound),
    w2b = copy.deepcopy(w2); w2b.HM, w2b.bM = HMg, bMg
    ob = po.OracleWindow(w2b); ob.marginalize_frame(0)
    HM2b, bM2b = ob.get_prior(); ob.close()
    observe("c5_marginalize_frame_device_input_HM", blockrel(HM2g, HM2b, 4), 1e-9); observe("c5_marginalize_frame_device_input_bM", rel(bM2g, bM2b), 1e-9)
    #  (iii) and the chained difference is the oracle's own sensitivity to that input difference (blockrel(HM2b, HM2o)): no tolerance of its own
    sens_H, sens_b = blockrel(HM2b, HM2o, 4), rel(bM2b, bM2o)
    assert blockrel(HM2g, HM2o, 4) <= 1.01 * sens_H + 1e-8 and rel(bM2g, bM2o) <= 1.01 * sens_b + 1e-8, (blockrel(HM2g, HM2o, 4), sens_H, rel(bM2g, bM2o), sens_b)
    print("C5 marginalizeFrame: chained difference to the oracle", blockrel(HM2g, HM2o, 4), rel(bM2g, bM2o), "= the oracle's sensitivity to the prior's fp32 difference", sens_H, sens_b)

    # F = 11: window rebuilt without frame 0, carrying the DEVICE's prior; 10 more iterations on both sides
    ex = o.export_window(); fo = o.get_frames()
    assert ex["F"] == 11
    w3 = copy.deepcopy(win)
    w3.points, w3.residuals, w3.lin_J, w3.lin_res_toZeroF = ex["points"], ex["residuals"], ex["lin_J"], ex["lin_res_toZeroF"]
    w3.frames = fo["frames"]
    w3.calib = w3.calib.copy(); w3.calib["value"] = fo["calib_value"]
    w3.images = win.images[1:]
    w3.HM, w3.bM = HM2g, bM2g
    assert w3.F == 11 and win.P - flags.sum() - 200 < w3.P <= win.P - flags.sum()      # marginalised points gone, a few more dropped as OOB / outliers
    o3 = po.OracleWindow(w3); o3.set_force_all_iterations(True)
    g3 = binding.BA.from_window(w3, stream=st)
    rmo = o3.optimize(10); rmg, done = g3.optimize(10, force_all=True)
    eo, eg = o3.energy_log(), g3.get_energy_log()
    assert done == 10 and len(eo) == len(eg) == 12
    assert np.all(np.abs(eg - eo) <= TOL * np.abs(eo)), (eo, eg)
    assert abs(rmo - rmg) <= TOL * rmo
    ro, rg = o3.get_residuals(False), g3.get_residuals()
    # Residual states after 10 iterations: a state can only differ where the residual sits AT its outlier threshold - the energies of the two sides agree to
    # TOL, so a residual whose energy is further than 100 x TOL (relative) from max(frameEnergyTH of host, target) in the oracle's run is on the same side on
    # the device.  (Until round 5: "at most 0.2 % differ", a number taken from the product's own output.  Observed on MI355X: 0 differ, 3 of 63 250 are that
    # close; the oracle's -O3 and six-thread builds against its portable build: 0 differ.)
    flipped = np.nonzero(ro["state_state"] != rg["state_state"])[0]
    fr3 = o3.get_frames()["frames"]
    th = np.maximum(fr3["frameEnergyTH"][w3.residuals["host"]], fr3["frameEnergyTH"][w3.residuals["target"]])
    near = np.abs(ro["out"]["state_NewEnergyWithOutlier"] - th) <= 100 * TOL * th
    assert np.all(near[flipped]), ("residual states differ away from the outlier threshold", flipped[~near[flipped]][:10])
    print("C5 / F = 11 after 10 iterations: residual states differing", len(flipped), "of", w3.R, "| within 1e-2 of their threshold in the oracle:", int(near.sum()))
