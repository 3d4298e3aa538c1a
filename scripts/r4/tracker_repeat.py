"""Is ldso_tr_track deterministic?  The 'five times brighter' scenario of tests/test_nonfinite_gpu.py (an LM run that never converges: every
accept / reject decision matters) repeated on one handle and on fresh handles; prints the distinct results."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
from tracker_common import tracker_scenario
from ldso_amd import binding
sc = tracker_scenario("small")
win = sc["win"]
pyr = [l.copy() * np.float32(5.0) for l in sc["new_pyr"]]
a, b = sc["new_aff"]


def make():
    g = binding.Tracker(win.w, win.h, sc["levels"], win.settings, win.calib)
    g.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"])
    g.set_new_frame(pyr, 1.0)
    return g


def key(r):
    return (round(float(r["a"]), 5), int(r["iterations"]), tuple(np.round(np.nan_to_num(r["lastResiduals"], nan=-1), 4)))


g = make()
same = {}
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    k = key(g.track(np.eye(4), a, b, sc["levels"] - 1)); same[k] = same.get(k, 0) + 1
print("one handle :", same)
fresh = {}
for i in range(15):
    h = make(); k = key(h.track(np.eye(4), a, b, sc["levels"] - 1)); fresh[k] = fresh.get(k, 0) + 1; h.close()
print("fresh      :", fresh)
