import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from tracker_common import tracker_scenario
from ldso_amd import synth, binding
from oracle import pyoracle as po
for name, levels in (("small", None), ("C3", None), ("C3", 5)):
    sc = tracker_scenario(name, levels=levels); win = sc["win"]
    o = po.OracleTracker(win.w, win.h, sc["levels"], win.settings, win.calib); g = binding.Tracker(win.w, win.h, sc["levels"], win.settings, win.calib)
    for t in (o, g):
        t.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"]); t.set_new_frame(sc["new_pyr"], 1.0)
    a, b = sc["new_aff"]
    ro = o.track(np.eye(4), a, b, sc["levels"] - 1); rg = g.track(np.eye(4), a, b, sc["levels"] - 1)
    To, Tg = np.eye(4), np.eye(4); To[:3, :4] = ro["T"]; Tg[:3, :4] = rg["T"]
    d = np.linalg.norm(synth.se3_log(Tg @ np.linalg.inv(To))); m = np.linalg.norm(synth.se3_log(To))
    print(name, levels, "d", d, "m", m, "d/m", d / m, "its", ro["iterations"], rg["iterations"], "res rel", np.abs(np.array(rg["lastResiduals"][:sc["levels"]]) / np.array(ro["lastResiduals"][:sc["levels"]]) - 1).max())
