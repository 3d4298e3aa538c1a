"""Run-to-run reproducibility of optimize() (fp64 atomics / LDS float atomics only reorder sums). Run on the GPU box."""
import sys
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding
from oracle import pyoracle as po
for name in ['small', 'mixed', 'C3']:
    win = po.make_mixed_window(synth.make_config('small')) if name == 'mixed' else synth.make_config(name)
    if name == 'C3': synth.add_synthetic_prior(win)
    res = []
    for rep in range(4):
        g = binding.BA.from_window(win)
        g.optimize(3, force_all=True)
        fr = g.get_frames()['frames']['state'].copy(); e = np.array(g.get_energy_log())
        res.append((fr, e)); g.close() if hasattr(g, 'close') else None
    d = max(np.abs(r[0] - res[0][0]).max() for r in res[1:]); de = max(np.abs(r[1] - res[0][1]).max() / np.abs(res[0][1]).max() for r in res[1:])
    print(name, 'max state diff between runs', d, 'max rel energy-log diff', de)
