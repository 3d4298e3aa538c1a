"""End-to-end wall time of ldso_ba_optimize (the FullSystem::optimize drop-in) on a resident window. Run on the GPU box."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
win = synth.add_synthetic_prior(synth.make_config(cfg))
for force in (True, False):
    ts, its = [], []
    for rep in range(12):
        g = binding.BA.from_window(win)
        g.sync()
        t0 = time.perf_counter()
        rm, n = g.optimize(6, force_all=force)
        ts.append(time.perf_counter() - t0); its.append(n)
        g.close()
    print(cfg, 'optimize(6, force_all=%s): median %.1f us, iterations %s, rmse %.4f' % (force, np.median(ts[2:]) * 1e6, sorted(set(its)), rm))
