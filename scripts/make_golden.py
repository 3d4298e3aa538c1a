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

# ---- immature-point tracing (ImmaturePoint::traceOn) ----
win = synth.make_config('small', extra_frames=1)
pts, _ = synth.make_immature_points(win, 60)
KRKi, Kt, aff = synth.trace_poses(win, win.F)
counts = po.trace_on(pts, win.images[win.F][0], KRKi, Kt, aff)
np.savez_compressed('tests/golden/trace_small.npz', records=np.frombuffer(pts.tobytes(), dtype=np.uint8), counts=counts)

# ---- monocular initialiser (CoarseInitializer) ----
seq = synth.make_init_sequence(160, 120, n_frames=5, fx=100.0, seed=11, levels=3)
L = seq['levels']
pyr0 = synth.make_images(seq['first'], L)
ipts = synth.select_init_points(pyr0)
oi = po.OracleInitializer(160, 120, L)
oi.set_first(seq['K4'], pyr0, 1.0, ipts)
states = []
for k in range(5):
    oi.set_new_frame(synth.make_images(seq['frames'][k], L), 1.0)
    st = oi.track_frame()
    states.append(np.concatenate([st['thisToNext'], [st['aff_a'], st['aff_b'], st['snapped'], st['snappedAt'], st['frameID'], st['ready'], st['evals']]]))
p0 = oi.points(0)
T = seq['poses'][4].copy(); T[:3, 3] /= 3.0
H, b, Hsc, bsc, res, ec = oi.calc_res_and_gs(1, T, 0.01, 0.5)
np.savez_compressed('tests/golden/init_small.npz', n_points=np.array([len(p) for p in ipts]), states=np.array(states), iR0=p0['iR'], good0=p0['isGood'],
                    H=H, b=b, Hsc=Hsc, bsc=bsc, res=res, ec=ec)
print('written', os.listdir('tests/golden'))
