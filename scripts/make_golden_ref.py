"""Generate tests/golden/ref_*.npz from the REFERENCE-COMPILED library oracle/_ref/libldso_ref.so (the reference's own hot-path
translation units, compiled unmodified from /root/reference; see oracle/ref_driver.cc).  These are outputs of the reference itself,
run in this container; /root/reference does not exist on the GPU box, so the vectors travel as fixtures.
Run from the repo root: python scripts/make_golden_ref.py"""
import copy
import os
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from ldso_amd import synth
from oracle import pyoracle as po, pyref as pr

os.makedirs('tests/golden', exist_ok=True)
for name, prior in (("tiny", False), ("small", True)):
    win = synth.make_config(name)
    if prior:
        synth.add_synthetic_prior(win)
    r = pr.RefWindow(win)
    pre = r.get_precalc(); adH, adT, dpF = r.get_adjoints()
    r.collect_active(); E0 = r.linearize_all(); r0 = r.get_residuals()
    r.apply_res(); r1 = r.get_residuals(False)
    r.solve_system(0)
    acc = r.get_accumulators(); pts, _ = r.get_points(); s = r.get_system(); fr = r.get_frames()
    J = r0["J"]
    np.savez_compressed(f'tests/golden/ref_ba_{name}.npz', E0=E0, precalc=pre, adHost=adH, adTarget=adT, adHTdeltaF=dpF,
                        newState=r0['out']['state_NewState'], newEnergy=r0['out']['state_NewEnergy'], newEnergyWO=r0['out']['state_NewEnergyWithOutlier'],
                        **{"J_" + k: J[k] for k in J.dtype.names}, JpJdF=r1['out']['JpJdF'], is_active=r1['is_active'], state_state=r1['state_state'],
                        topA=acc['topA'], accD=acc['accD'], accE=acc['accE'], accEB=acc['accEB'], accHcc=acc['accHcc'], accbc=acc['accbc'],
                        **{"pt_" + k: pts[k] for k in pts.dtype.names}, lastHS=s['lastHS'], lastbS=s['lastbS'], x=s['x'],
                        frame_step=fr['step'], calib_step=fr['calib_step'], prior=fr['frames']['prior'], counts=np.array(r.counts()))
    r.close()
# FrameHessian::makeImages
rng = np.random.default_rng(3)
color = (rng.random((128, 160)) * 255).astype(np.float32)      # 160x128: three pyramid levels by the rule of GlobalCalib.cc:24
lv = pr.make_images(color, 3)
np.savez_compressed('tests/golden/ref_make_images.npz', color=color, **{f"l{l}": lv[l] for l in range(3)})
# CoarseTracker (calcRes / calcGSSSE / trackNewestCoarse) and ImmaturePoint::traceOn
from tracker_common import tracker_scenario
sc = tracker_scenario('small'); w = sc['win']
tr = pr.RefTracker(w.w, w.h, sc['levels'], w.settings, w.calib)
tr.set_ref(sc['ref_pyr'], sc['ref_aff'][0], sc['ref_aff'][1], 1.0, sc['pts']); tr.set_new_frame(sc['new_pyr'], 1.0)
a, b = sc['new_aff']
rs, n = tr.calc_res(1, np.eye(4), a, b, 20.0); H, bb = tr.calc_gs(1, np.eye(4), a, b)
t = tr.track(np.eye(4), a, b, sc['levels'] - 1)
np.savez_compressed('tests/golden/ref_tracker_small.npz', pc_n=np.array([len(tr.pc(l)[0]) for l in range(sc['levels'])]), pc0=np.stack(tr.pc(0)),
                    rs=rs, n=n, H=H, b=bb, T=t['T'], ab=np.array([t['a'], t['b']]), lastResiduals=t['lastResiduals'], flow=t['flow'])
win = synth.make_config('small', extra_frames=1)
pts, _ = synth.make_immature_points(win, 60)
KRKi, Kt, aff = synth.trace_poses(win, win.F)
counts = pr.trace_on(pts, win.images[win.F][0], KRKi, Kt, aff)
np.savez_compressed('tests/golden/ref_trace_small.npz', records=np.frombuffer(pts.tobytes(), dtype=np.uint8), counts=counts)
print('written', sorted(f for f in os.listdir('tests/golden') if f.startswith('ref_')))
