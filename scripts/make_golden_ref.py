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
print('written', sorted(f for f in os.listdir('tests/golden') if f.startswith('ref_')))
