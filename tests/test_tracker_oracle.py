"""CPU tests of the CoarseTracker restatement (oracle): makeCoarseDepthL0 invariants, calcRes/calcGSSSE
consistency (b is the gradient of the energy), convergence of trackNewestCoarse to the true motion."""
import numpy as np

from conftest import rel
from tracker_common import tracker_scenario
from ldso_amd import synth
from oracle import pyoracle as po


def make_tracker(sc):
    win = sc["win"]
    tr = po.OracleTracker(win.w, win.h, sc["levels"], win.settings, win.calib)
    tr.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"])
    tr.set_new_frame(sc["new_pyr"], 1.0)
    return tr


def test_point_cloud_levels():
    sc = tracker_scenario("small")
    tr = make_tracker(sc)
    n = []
    for l in range(sc["levels"]):
        u, v, d, c = tr.pc(l)
        n.append(len(u))
        w, h = sc["win"].w >> l, sc["win"].h >> l
        assert (u >= 2).all() and (u < w - 2).all() and (v >= 2).all() and (v < h - 2).all() and (d > 0).all()
        # row-major compaction order
        key = v.astype(np.int64) * w + u.astype(np.int64)
        assert (np.diff(key) > 0).all()
    assert n[0] >= len(sc["pts"]) * 0.9 and n[0] <= 5 * len(sc["pts"])          # dilation adds up to 4 neighbours


def test_calc_gs_is_gradient_of_calc_res():
    sc = tracker_scenario("small")
    tr = make_tracker(sc)
    T = np.eye(4)
    a, b = sc["new_aff"]
    lvl = 1
    rs, n = tr.calc_res(lvl, T, a, b, 1e9)
    H, bb = tr.calc_gs(lvl, T, a, b)
    assert np.abs(H - H.T).max() <= 1e-9 * np.abs(H).max() and np.linalg.eigvalsh(H).min() > -1e-6 * np.abs(H).max()
    # finite differences of E/n w.r.t. a left-multiplied increment, in the solver's scaled coordinates
    sc8 = np.array([1, 1, 1, 0.5, 0.5, 0.5, 10, 1000.0])
    g = np.zeros(8)
    for i in range(8):
        e = np.zeros(8); e[i] = 1e-4
        def E(sign):
            inc = sign * e * sc8
            Tn = synth.se3_exp(inc[:6]) @ T
            r, nn = tr.calc_res(lvl, Tn, a + inc[6], b + inc[7], 1e9)
            return r[0] / nn
        g[i] = (E(+1) - E(-1)) / 2e-4
    # E/n ~ sum hw r^2 (2-hw)/n, d/dx = 2 * sum hw r J / n = 2 b (Huber), b uses the padded n
    scale = n / rs[1]
    assert rel(g[:6], 2 * bb[:6] * scale) < 5e-2


def test_track_converges_to_truth():
    sc = tracker_scenario("small")
    tr = make_tracker(sc)
    r = tr.track(np.eye(4), sc["new_aff"][0], sc["new_aff"][1], sc["levels"] - 1)
    assert r["ok"]
    T = np.eye(4); T[:3, :4] = r["T"]
    dT = T @ np.linalg.inv(sc["T_true"])
    err = np.linalg.norm(synth.se3_log(dT))
    err0 = np.linalg.norm(synth.se3_log(np.linalg.inv(sc["T_true"])))
    assert err < 0.25 * err0
    assert r["lastResiduals"][0] < 10
