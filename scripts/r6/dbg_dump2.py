import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ldso_amd import synth, binding
from oracle import pyoracle as po
win = synth.make_config("tiny")
o = po.OracleWindow(win); o.collect_active(); o.linearize_all(False); ro = o.get_residuals()
g = binding.BA.from_window(win); g.collect_active(); g.linearize_all(False); g.apply_res()
a = g.get_jacobians()["resF"]; b = ro["J"]["resF"]
print("recompute path rows written", int((np.abs(a).max(1) != 0).sum()), "of", len(a))
