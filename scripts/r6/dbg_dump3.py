import sys
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from ldso_amd import synth, binding
win = synth.make_config("tiny")
g = binding.BA.from_window(win); g.collect_active(); g.set_debug_dump(True); g.linearize_all(False)
a = g.get_jacobians()["resF"]
print("debug-dump path rows written", int((np.abs(a).max(1) != 0).sum()), "of", len(a))
