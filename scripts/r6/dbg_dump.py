import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ldso_amd import synth, binding
from oracle import pyoracle as po
cfg = sys.argv[1] if len(sys.argv) > 1 else "tiny"
win = synth.make_config(cfg)
o = po.OracleWindow(win); o.collect_active(); o.linearize_all(False); ro = o.get_residuals()
def rep(tag, Jg):
    a, b = Jg["resF"], ro["J"]["resF"]
    nz = (np.abs(a).max(1) != 0)
    print(cfg, tag, "rows written", int(nz.sum()), "of", len(a), "rows equal", int((np.abs(a - b).max(1) <= 1e-5 * np.abs(b).max()).sum()), "first unwritten", np.nonzero(~nz)[0][:12], "points", win.residuals["point"][~nz][:12])
g = binding.BA.from_window(win); g.collect_active(); g.set_debug_dump(True); g.linearize_all(False)
rep("a: first linearisation with the dump", g.get_jacobians())
g = binding.BA.from_window(win); g.collect_active(); g.linearize_all(False); g.apply_res(); g.set_debug_dump(True); g.linearize_all(False)
rep("b: second linearisation with the dump", g.get_jacobians())
g = binding.BA.from_window(win); g.collect_active(); g.linearize_all(False); g.apply_res(); g.set_debug_dump(False)
rep("c: recompute path", g.get_jacobians())
