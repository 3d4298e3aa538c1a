"""Stage-by-stage parity report GPU (libldso_hip.so) vs oracle. Run on the GPU box: python scripts/gpu_report.py [config]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding
from oracle import pyoracle as po

def rel(a, b, floor=0.0):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    d = np.abs(a - b).max() if a.size else 0.0
    return d / max(np.abs(b).max() if b.size else 0.0, floor, 1e-300)

def blockrel(A, B, bs=4):
    n = A.shape[0]; worst = 0
    for i in range(0, n, bs):
        for j in range(0, n, bs):
            a = A[i:i+bs, j:j+bs]; b = B[i:i+bs, j:j+bs]
            m = np.abs(b).max()
            if m > 0: worst = max(worst, np.abs(a-b).max()/m)
    return worst

cfg = sys.argv[1] if len(sys.argv) > 1 else 'small'
win = synth.make_config(cfg)
print(cfg, 'F', win.F, 'P', win.P, 'R', win.R)
o = po.OracleWindow(win)
g = binding.BA.from_window(win)

pc_o = o.get_precalc(); pc_g = g.get_precalc()
print('precalc rel', rel(pc_g, pc_o), 'exact frac', (pc_g == pc_o).mean())

o.collect_active(); g.collect_active()
g.set_debug_dump(True)
Eo = o.linearize_all(False); Eg = g.linearize_all(False)
print('E0', Eo, Eg, 'rel', abs(Eo-Eg)/Eo)
ro = o.get_residuals(); rg = g.get_residuals()
print('NewState mismatch', (ro['out']['state_NewState'] != rg['out']['state_NewState']).sum(), 'of', win.R)
print('NewEnergy rel', rel(rg['out']['state_NewEnergy'], ro['out']['state_NewEnergy']), 'exact', (rg['out']['state_NewEnergy'] == ro['out']['state_NewEnergy']).mean())
print('NewEnergyWO exact', (rg['out']['state_NewEnergyWithOutlier'] == ro['out']['state_NewEnergyWithOutlier']).mean())
fo = o.get_frames(); fg = g.get_frames()
print('frameEnergyTH', fo['frames']['frameEnergyTH'], fg['frames']['frameEnergyTH'])
Jg = g.get_jacobians()
ok = ro['out']['state_NewState'] != 1
for k in ['resF','Jpdxi','Jpdc','Jpdd','JIdx','JabF','JIdx2','JabJIdx','Jab2']:
    print('  J', k, rel(Jg[k][ok], ro['J'][k][ok]), 'exact', (Jg[k][ok] == ro['J'][k][ok]).mean())
o.apply_res(); g.apply_res(); g.set_debug_dump(False)
ro = o.get_residuals(); rg = g.get_residuals()
print('is_active mismatch', (ro['is_active'] != rg['is_active']).sum(), 'JpJdF rel', rel(rg['out']['JpJdF'], ro['out']['JpJdF']), 'center rel', rel(rg['out']['centerProjectedTo'], ro['out']['centerProjectedTo']))
o.backup_state(); g.backup_state()
o.solve_system(0); g.solve_system(0)
so = o.get_system(); sg = g.get_system()
for k in ['HA','HL','Hsc','HFinal']:
    print(' ', k, 'rel(max)', rel(sg[k], so[k]), 'blockrel', blockrel(sg[k], so[k]), 'sym', np.abs(sg[k]-sg[k].T).max()/np.abs(sg[k]).max())
for k in ['bA','bL','bsc','bFinal','x']:
    print(' ', k, 'rel', rel(sg[k], so[k]))
print('  x backward err gpu', np.linalg.norm(sg['HFinal']@sg['x']-sg['bFinal'])/np.linalg.norm(sg['bFinal']))
pto, _ = o.get_points(); ptg = g.get_points()
for k in ['step','HdiF','bdSumF','idepth_hessian','Hdd_accAF','bd_accAF','Hcd_accAF']:
    print('  pt', k, rel(ptg[k], pto[k]))
print('counts', o.counts(), g.get_counts())
cbo = o.do_step(); cbg = g.do_step()
print('canbreak', cbo, cbg)
fo = o.get_frames(); fg = g.get_frames()
print('state rel', rel(fg['frames']['state'], fo['frames']['state']), 'step rel', rel(fg['step'], fo['step']), 'calib', rel(fg['calib_value'], fo['calib_value']))
pto, _ = o.get_points(); ptg = g.get_points()
print('idepth rel', rel(ptg['idepth'], pto['idepth']))
Eo = o.linearize_all(False); Eg = g.linearize_all(False)
print('E1', Eo, Eg, 'rel', abs(Eo-Eg)/Eo)

# full optimize
o2 = po.OracleWindow(win); o2.set_force_all_iterations(True)
g2 = binding.BA.from_window(win)
t=time.time(); rmo = o2.optimize(6); to=time.time()-t
t=time.time(); rmg, its = g2.optimize(6, force_all=True); tg=time.time()-t
print('optimize rmse', rmo, rmg, 'its', its, 'time cpu %.3f ms gpu %.3f ms' % (to*1e3, tg*1e3))
eo = o2.energy_log(); eg = g2.get_energy_log()
print('energy log cpu', eo); print('energy log gpu', eg)
print('energy log rel', rel(eg, eo))
fo = o2.get_frames(); fg = g2.get_frames()
print('final state rel', rel(fg['frames']['state'], fo['frames']['state']), 'evalPT rel', rel(fg['frames']['worldToCam_evalPT'], fo['frames']['worldToCam_evalPT']))
pto,_ = o2.get_points(); ptg = g2.get_points()
print('final idepth rel', rel(ptg['idepth'], pto['idepth']), 'numGood mismatch', (ptg['numGoodResiduals'] != pto['numGoodResiduals']).sum(), 'maxRelBS rel', rel(ptg['maxRelBaseline'], pto['maxRelBaseline']))
ro = o2.get_residuals(); rg = g2.get_residuals()
print('final state mismatch', (ro['state_state'] != rg['state_state']).sum(), 'active mismatch', (ro['is_active'] != rg['is_active']).sum(), 'to_remove', rg['to_remove'].sum(), 'oracle dropped', (ro['alive']==0).sum())
# timing of repeated optimize
g2.load_window(win)
t=time.time(); g2.optimize(10, force_all=True); print('gpu optimize(10) wall ms', (time.time()-t)*1e3)
g2.load_window(win); g2.collect_active(); g2.linearize_all(False); g2.apply_res()
g2.sync(); t=time.time(); g2.enqueue_gn(0, 100); g2.sync(); dt=time.time()-t
print('enqueue_gn 100 its: %.3f ms/iter' % (dt*1e3/100))
g2.profile(True); g2.enqueue_gn(0, 20); g2.sync()
for i,nm in enumerate(['linearize','reduce','solve','pointstep']):
    print('  kernel', nm, g2.kernel_time_ms(i))
# phase timing of the control kernel through the step-wise API
g3 = binding.BA.from_window(win); g3.collect_active(); g3.linearize_all(False); g3.apply_res(); g3.backup_state()
for nm, fn in [('post+thresh (linearize_all)', lambda: g3.linearize_all(False)), ('apply', lambda: g3.apply_res()), ('gather+solve (solve_system)', lambda: g3.solve_system(0)), ('step+precalc (do_step)', lambda: g3.do_step())]:
    g3.profile(True); fn(); t = [g3.kernel_time_ms(i) for i in range(4)]; g3.profile(False)
    print('  phase', nm, ' '.join('%s=%.1fus' % (k, v[0]*1e3) for k, v in zip(['lin','red','solve','pstep'], t) if v[1] > 0))
g3.solve_system(0); g3.get_energy_log(); print('stamps solve_system (cycles):', g3._dbg[:10])
g3.do_step(); g3.get_energy_log(); print('stamps do_step:', g3._dbg[:10])
g3.linearize_all(False); g3.get_energy_log(); print('stamps post:', g3._dbg[:10])
