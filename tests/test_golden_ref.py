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
