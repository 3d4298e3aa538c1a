"""Golden fixtures produced by the REFERENCE ITSELF: tests/golden/ref_*.npz are outputs of oracle/_ref/libldso_ref.so - the
reference's hot-path translation units compiled unmodified from /root/reference (scripts/make_golden_ref.py, oracle/ref_driver.cc).
/root/reference does not travel to the GPU box; these vectors do.
CPU: the oracle reproduces them BIT FOR BIT (so every GPU-vs-oracle test is a GPU-vs-reference test on the quantities listed here).
GPU: the HIP path matches them within the north_star tolerances (1e-4 on energies / Hessian entries, exact states)."""
import copy
import os

import numpy as np
import pytest

from conftest import rel, blockrel, get_window
from ldso_amd import synth
from oracle import pyoracle as po

G = os.path.join(os.path.dirname(__file__), "golden")


def _win(name):
    w = get_window(name)
    return synth.add_synthetic_prior(copy.deepcopy(w)) if name == "small" else w


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_oracle_reproduces_reference_vectors_bit_for_bit(name):
    g = np.load(os.path.join(G, f"ref_ba_{name}.npz"))
    win = _win(name)
    o = po.OracleWindow(win)
    assert np.array_equal(o.get_precalc(), g["precalc"])
    ah, at, d = o.get_adjoints()
    assert np.array_equal(ah, g["adHost"]) and np.array_equal(at, g["adTarget"]) and np.array_equal(d, g["adHTdeltaF"])
    o.collect_active(); E0 = o.linearize_all(False); r0 = o.get_residuals()
    assert E0 == float(g["E0"])
    assert np.array_equal(r0["out"]["state_NewState"], g["newState"]) and np.array_equal(r0["out"]["state_NewEnergy"], g["newEnergy"])
    assert np.array_equal(r0["out"]["state_NewEnergyWithOutlier"], g["newEnergyWO"])
    ok = g["newState"] != 1
    for k in r0["J"].dtype.names:
        assert np.array_equal(r0["J"][k][ok], g["J_" + k][ok]), k
    o.apply_res(); r1 = o.get_residuals(False)
    assert np.array_equal(r1["out"]["JpJdF"], g["JpJdF"]) and np.array_equal(r1["is_active"], g["is_active"]) and np.array_equal(r1["state_state"], g["state_state"])
    o.backup_state(); o.solve_system(0)
    acc = o.get_accumulators()
    for k in ("topA", "accD", "accE", "accEB", "accHcc", "accbc"):
        assert np.array_equal(acc[k], g[k]), k
    pts, _ = o.get_points()
    for k in pts.dtype.names:
        assert np.array_equal(pts[k], g["pt_" + k]), k
    s = o.get_system()
    for k in ("lastHS", "lastbS", "x"):
        assert np.array_equal(s[k], g[k]), k
    fr = o.get_frames()
    assert np.array_equal(fr["step"], g["frame_step"]) and np.array_equal(fr["calib_step"], g["calib_step"]) and np.array_equal(fr["frames"]["prior"], g["prior"])
    assert tuple(g["counts"]) == o.counts()


def test_oracle_make_images_reproduces_reference_vectors():
    g = np.load(os.path.join(G, "ref_make_images.npz"))
    lv = po.make_images(g["color"], 3)
    for l in range(3):
        assert np.array_equal(lv[l][1:-1], g[f"l{l}"][1:-1]) and np.array_equal(lv[l][:, :, 0], g[f"l{l}"][:, :, 0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "small"])
def test_gpu_matches_reference_vectors(name):
    from ldso_amd import binding
    g = np.load(os.path.join(G, f"ref_ba_{name}.npz"))
    win = _win(name)
    b = binding.BA.from_window(win)
    assert rel(b.get_precalc(), g["precalc"]) < 1e-6
    b.collect_active(); b.set_debug_dump(True)
    E0 = b.linearize_all(False)
    assert abs(E0 - float(g["E0"])) <= 1e-6 * float(g["E0"])
    r0 = b.get_residuals()
    assert np.array_equal(r0["out"]["state_NewState"], g["newState"])                      # exact states
    assert rel(r0["out"]["state_NewEnergy"], g["newEnergy"]) < 1e-5
    J = b.get_jacobians(); ok = g["newState"] != 1
    for k in ("resF", "Jpdxi", "Jpdc", "Jpdd", "JIdx", "JabF", "JIdx2", "JabJIdx", "Jab2"):
        assert rel(J[k][ok], g["J_" + k][ok]) < 1e-5, k
    b.set_debug_dump(False)
    b.apply_res(); r1 = b.get_residuals()
    assert np.array_equal(r1["is_active"], g["is_active"]) and np.array_equal(r1["state_state"], g["state_state"])
    act = g["is_active"].astype(bool)
    assert rel(r1["out"]["JpJdF"][act], g["JpJdF"][act]) < 1e-5
    b.backup_state(); b.solve_system(0)
    s = b.get_system()
    HS = s["HFinal"]                                                                        # lastHS is HFinal before the lambda scaling (EF.cc:327-334)
    lam = 1e-5
    HSg = s["HA"] + s["HL"] - s["Hsc"] + win.HM
    assert blockrel(HSg, g["lastHS"], 4) < 1e-4 and rel(s["bFinal"], g["lastbS"]) < 1e-4
    pts = b.get_points()
    for k in ("HdiF", "bdSumF", "idepth_hessian", "Hdd_accAF", "bd_accAF", "Hcd_accAF"):
        assert rel(pts[k], g["pt_" + k]) < 1e-5, k
    xg, xo, Ho, bo = s["x"], g["x"], g["lastHS"], g["lastbS"]
    mo, mg = 2 * bo @ xo - xo @ Ho @ xo, 2 * bo @ xg - xg @ Ho @ xg
    assert abs(mo - mg) <= 1e-6 * abs(mo)                                                   # the step achieves the reference step's model decrease
    assert tuple(g["counts"][:2]) == b.get_counts()


@pytest.mark.gpu
def test_gpu_make_images_matches_reference_vectors():
    from ldso_amd import binding
    g = np.load(os.path.join(G, "ref_make_images.npz"))
    h, w = g["color"].shape
    b = binding.BA(w, h, 2, 8)
    b.set_image_raw(0, g["color"])
    out = b.get_image(0)
    assert np.array_equal(out[1:-1], g["l0"][1:-1]) and np.array_equal(out[:, :, 0], g["l0"][:, :, 0])


def _tracker(cls_module):
    from tracker_common import tracker_scenario
    sc = tracker_scenario("small"); w = sc["win"]
    return sc, w


def test_oracle_tracker_reproduces_reference_vectors():
    g = np.load(os.path.join(G, "ref_tracker_small.npz"))
    sc, w = _tracker(po)
    tr = po.OracleTracker(w.w, w.h, sc["levels"], w.settings, w.calib)
    tr.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"]); tr.set_new_frame(sc["new_pyr"], 1.0)
    a, b = sc["new_aff"]
    assert np.array_equal([len(tr.pc(l)[0]) for l in range(sc["levels"])], g["pc_n"]) and np.array_equal(np.stack(tr.pc(0)), g["pc0"])
    rs, n = tr.calc_res(1, np.eye(4), a, b, 20.0); H, bb = tr.calc_gs(1, np.eye(4), a, b)
    assert n == int(g["n"]) and np.array_equal(rs, g["rs"]) and np.array_equal(H, g["H"]) and np.array_equal(bb, g["b"])
    t = tr.track(np.eye(4), a, b, sc["levels"] - 1)
    assert np.array_equal(t["T"], g["T"]) and np.array_equal([t["a"], t["b"]], g["ab"]) and np.array_equal(t["flow"], g["flow"])
    assert np.array_equal(t["lastResiduals"], g["lastResiduals"], equal_nan=True)


def test_oracle_trace_on_reproduces_reference_vectors():
    g = np.load(os.path.join(G, "ref_trace_small.npz"))
    win = synth.make_config('small', extra_frames=1)
    pts, _ = synth.make_immature_points(win, 60)
    KRKi, Kt, aff = synth.trace_poses(win, win.F)
    counts = po.trace_on(pts, win.images[win.F][0], KRKi, Kt, aff)
    assert np.array_equal(counts, g["counts"]) and pts.tobytes() == g["records"].tobytes()


@pytest.mark.gpu
def test_gpu_tracker_matches_reference_vectors():
    from ldso_amd import binding
    g = np.load(os.path.join(G, "ref_tracker_small.npz"))
    sc, w = _tracker(po)
    tr = binding.Tracker(w.w, w.h, sc["levels"], w.settings, w.calib)
    tr.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"]); tr.set_new_frame(sc["new_pyr"], 1.0)
    a, b = sc["new_aff"]
    assert np.array_equal([len(tr.pc(l)[0]) for l in range(sc["levels"])], g["pc_n"])            # point clouds: index-exact
    u, v, d, c = tr.pc(0)
    assert np.array_equal(u, g["pc0"][0]) and np.array_equal(v, g["pc0"][1]) and rel(d, g["pc0"][2]) < 1e-6 and rel(c, g["pc0"][3]) < 1e-6
    rs, n = tr.calc_res(1, np.eye(4), a, b, 20.0); H, bb = tr.calc_gs(1, np.eye(4), a, b)
    assert n == int(g["n"]) and abs(rs[0] - g["rs"][0]) <= 1e-4 * g["rs"][0]
    assert blockrel(H, g["H"], 8) < 1e-4 and rel(bb, g["b"]) < 1e-4
    t = tr.track(np.eye(4), a, b, sc["levels"] - 1)
    assert np.abs(t["T"] - g["T"]).max() < 2e-2 and abs(t["lastResiduals"][0] - g["lastResiduals"][0]) < 1e-3 * g["lastResiduals"][0]


@pytest.mark.gpu
def test_gpu_trace_on_matches_reference_vectors():
    from ldso_amd import binding
    g = np.load(os.path.join(G, "ref_trace_small.npz"))
    win = synth.make_config('small', extra_frames=1)
    pts, _ = synth.make_immature_points(win, 60)
    KRKi, Kt, aff = synth.trace_poses(win, win.F)
    t = binding.Tracer(win.w, win.h, len(pts))
    t.set_frame(win.images[win.F][0]); t.set_points(pts)
    c = t.trace_on(KRKi, Kt, aff)
    assert np.array_equal(np.asarray(c)[:3], g["counts"][:3])
    assert t.get_points().tobytes() == g["records"].tobytes()                                     # byte-identical records
