"""GPU parity of point activation (ldso_ba_activate_points: FullSystem::optimizeImmaturePoint / ImmaturePoint::linearizeResidual)
against the oracle restatement on the same inputs: inverse depth, energy, Hdd, bd bit for bit, verdict and per-target residual
states exact; plus the purpose of the function - the activated inverse depths are close to the scene's."""
import numpy as np
import pytest

from ldso_amd import synth, binding
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _traced_points(win, per_frame):
    """immature points of the window's key frames, traced (oracle) against the two extra frames so that they carry intervals"""
    pts, true_id = synth.make_immature_points(win, per_frame)
    F = win.F
    for fidx in (F, F + 1):
        KRKi, Kt, aff = synth.trace_poses(win, fidx)
        po.trace_on(pts, win.images[fidx][0], KRKi, Kt, aff)
    keep = np.isfinite(pts["idepth_max"]) & (pts["lastTraceStatus"] != 1)
    return pts[keep].copy(), true_id[keep]


@pytest.mark.parametrize("name,per_frame", [("small", 120), ("C3", 200)])
def test_activation_matches_oracle(name, per_frame):
    win = synth.make_config(name, extra_frames=2)
    pts, true_id = _traced_points(win, per_frame)
    assert len(pts) > 50
    g = binding.BA.from_window(win)
    pairs = g.get_pair_rt()
    K4 = (np.float32(50.0) * win.calib["value"]).astype(np.float32) if False else np.asarray([np.float32(50.0 * v) for v in win.calib["value"]], np.float32)
    F = win.F
    ref = po.activate_points(pts, [win.images[f][0] for f in range(F)], K4, pairs, win.w, win.h)
    out = g.activate_points(pts)
    assert np.array_equal(out["ok"], ref["ok"]) and np.array_equal(out["res_state"], ref["res_state"])
    assert np.array_equal(out["numGoodRes"], ref["numGoodRes"]) and np.array_equal(out["iterations"], ref["iterations"])
    for k in ("idepth", "energy", "Hdd", "bd"):
        assert np.array_equal(out[k].view(np.uint32), ref[k].view(np.uint32)), k
    ok = out["ok"] == 1
    assert ok.mean() > 0.5
    relerr = np.abs(out["idepth"][ok] - true_id[ok]) / true_id[ok]
    assert np.median(relerr) < 0.05                      # activation lands on the scene's inverse depth


def test_activation_edge_cases():
    win = synth.make_config("small", extra_frames=2)
    pts, _ = _traced_points(win, 40)
    pts = pts[:24].copy()
    pts["idepth_min"][0] = np.nan                        # non-finite start: rejected
    pts["idepth_min"][1] = 50.0; pts["idepth_max"][1] = 60.0     # absurdly close: projections leave the images (OOB residuals)
    pts["u"][2] = 2.0; pts["v"][2] = 2.0                 # pattern leaves the image in the targets
    g = binding.BA.from_window(win)
    pairs = g.get_pair_rt()
    K4 = np.asarray([np.float32(50.0 * v) for v in win.calib["value"]], np.float32)
    ref = po.activate_points(pts, [win.images[f][0] for f in range(win.F)], K4, pairs, win.w, win.h)
    out = g.activate_points(pts)
    assert out.tobytes() == ref.tobytes() or (np.array_equal(out["ok"], ref["ok"]) and np.array_equal(out["res_state"], ref["res_state"]))
    assert out["ok"][0] == 0
    assert len(g.activate_points(pts[:0])) == 0
