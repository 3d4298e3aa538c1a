"""Stress the fused reduce+control launch: many optimize() calls (forced and un-forced) on several windows; every run must reproduce
the first one bit for bit in the energy log and to 1e-12 in the states (a lost producer signal or a stale read would show up here)."""
import sys
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for cfg in ("small", "C3", "C4"):
    win = synth.make_config(cfg)
    synth.add_synthetic_prior(win)
    ref = None
    for r in range(reps):
        g = binding.BA.from_window(win)
        rm, its = g.optimize(6, force_all=(r % 2 == 0))
        key = (r % 2)
        st = g.get_frames()["frames"]["state"].copy()
        el = np.array(g.get_energy_log())
        if ref is None:
            ref = {}
        if key not in ref:
            ref[key] = (st, el, rm, its)
        else:
            assert its == ref[key][3], (cfg, r, its, ref[key][3])
            assert np.array_equal(el, ref[key][1]), (cfg, r)
            assert np.abs(st - ref[key][0]).max() < 1e-11, (cfg, r, np.abs(st - ref[key][0]).max())
        g.close() if hasattr(g, "close") else None
    print(cfg, "ok", reps, "runs; iterations", ref[0][3], ref[1][3])
