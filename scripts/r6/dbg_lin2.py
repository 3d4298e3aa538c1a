"""which residuals differ between the stage-wise device linearisation and the oracle (by point index inside its chunk)"""
import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from ldso_amd import synth, binding
from oracle import pyoracle as po
P = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W_ = int(sys.argv[2]) if len(sys.argv) > 2 else 256
win = synth.make_window(F=5, P=P, w=W_, h=192, fx=160.0, seed=3)
o = po.OracleWindow(win); g = binding.BA.from_window(win)
o.collect_active(); g.collect_active()
Eo, Eg = o.linearize_all(False), g.linearize_all(False)
print("energy", Eo, Eg)
ro, rg = o.get_residuals(False), g.get_residuals()
so, sg = ro["out"]["state_NewState"], rg["out"]["state_NewState"]
bad = np.nonzero(so != sg)[0]
print("residuals", len(so), "state differs", len(bad))
pt = win.residuals["point"] if "point" in win.residuals.dtype.names else None
print(win.residuals.dtype.names)
if pt is not None and len(bad):
    bp = np.unique(pt[bad]); print("points with differences", len(bp), "of", win.P, bp[:64])
    try:
        cuts = np.asarray(g.get_chunk_cuts()); print("chunk cuts", cuts[:20])
        ch = np.searchsorted(cuts, bp, side="right"); st = np.concatenate([[0], cuts])[ch]
        print("index inside the chunk of the differing points", np.unique(bp - st)[:64], "chunk sizes", np.unique(np.diff(np.concatenate([[0], cuts]))))
    except Exception as e: print("cuts:", e)
eo, eg = ro["out"]["state_NewEnergy"], rg["out"]["state_NewEnergy"]
d = np.abs(eo - eg) > 1e-3 * np.maximum(1, np.abs(eo))
print("energy differs", d.sum())

