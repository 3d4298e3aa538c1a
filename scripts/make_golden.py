"""Generate tests/golden/*.npz from the oracle (the reference itself cannot be built here: Eigen/OpenCV/glog absent).
Run from the repo root: python scripts/make_golden.py"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ldso_amd import synth
from oracle import pyoracle as po
from tracker_common import tracker_scenario

os.makedirs('tests/golden', exist_ok=True)
win = synth.make_config('tiny')
o = po.OracleWindow(win)
o.collect_active(); E0 = o.linearize_all(False); r0 = o.get_residuals()
o.apply_res(); o.backup_state(); o.solve_system(0)
s = o.get_system(); pts, _ = o.get_points()
o2 = po.OracleWindow(win); o2.set_force_all_iterations(True); rm = o2.optimize(4)
np.savez_compressed('tests/golden/ba_tiny.npz', E0=E0, newState=r0['out']['state_NewState'], newEnergy=r0['out']['state_NewEnergy'],
                    J_resF=r0['J']['resF'], J_Jpdxi=r0['J']['Jpdxi'], J_JIdx=r0['J']['JIdx'], HA=s['HA'], bA=s['bA'], Hsc=s['Hsc'], bsc=s['bsc'],
                    HFinal=s['HFinal'], bFinal=s['bFinal'], x=s['x'], HdiF=pts['HdiF'], step=pts['step'], energy_log=o2.energy_log(), rmse=rm)
sc = tracker_scenario('small'); w = sc['win']
tr = po.OracleTracker(w.w, w.h, sc['levels'], w.settings, w.calib)
tr.set_ref(sc['ref_pyr'], sc['ref_aff'][0], sc['ref_aff'][1], 1.0, sc['pts']); tr.set_new_frame(sc['new_pyr'], 1.0)
a, b = sc['new_aff']
rs, n = tr.calc_res(1, np.eye(4), a, b, 20.0); H, bb = tr.calc_gs(1, np.eye(4), a, b)
r = tr.track(np.eye(4), a, b, sc['levels'] - 1)
np.savez_compressed('tests/golden/tracker_small.npz', pc_n=np.array([len(tr.pc(l)[0]) for l in range(sc['levels'])]), rs=rs, n=n, H=H, b=bb,
                    T=r['T'], ab=np.array([r['a'], r['b']]), lastResiduals=r['lastResiduals'], iterations=r['iterations'])
print('written', os.listdir('tests/golden'))
