import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, faulthandler
from tracker_common import tracker_scenario
from ldso_amd import binding
sc = tracker_scenario('small'); win = sc['win']
print('levels', sc['levels'], 'pts', sc['pts'].shape, flush=True)
g = binding.Tracker(win.w, win.h, sc['levels'], win.settings, win.calib)
print('created', flush=True)
g.set_new_frame(sc['new_pyr'], 1.0); print('new frame ok', flush=True)
g.set_ref(sc['ref_pyr'], sc['ref_aff'][0], sc['ref_aff'][1], 1.0, sc['pts']); print('ref ok', flush=True)
print([len(g.pc(l)[0]) for l in range(sc['levels'])])
