import sys; sys.path.insert(0,'.')
import numpy as np
from ldso_amd import synth, binding
for cfg in ['tiny','small','C3']:
    w = synth.make_config(cfg)
    g = binding.BA.from_window(w); g.collect_active(); g.linearize_all(False); g.apply_res(); g.backup_state(); g.solve_system(0)
    s = g.get_system(); H=s['HFinal']; b=s['bFinal']; n=len(b); x=s['x']
    S = 1/np.sqrt(np.diag(H)+10); A = S[:,None]*H*S[None,:]; rhs = S*b
    L=np.eye(n); D=np.zeros(n); M=A.copy()
    for k in range(n):
        D[k]=M[k,k]; L[k+1:,k]=M[k+1:,k]/D[k]
        M[k+1:,k+1:] -= np.outer(L[k+1:,k], L[k+1:,k])*D[k]
    y=np.linalg.solve(L,rhs); xn=np.linalg.solve(L.T, y/D)*S
    def berr(x): return np.linalg.norm(H@x-b)/np.linalg.norm(b), np.linalg.norm(S*(H@x-b))/np.linalg.norm(S*b)
    print(cfg, 'gpu', berr(x), 'numpy-nopivot', berr(xn), 'x rel', np.abs(x-xn).max()/np.abs(xn).max(), 'asym', np.abs(H-H.T).max()/np.abs(H).max())
    d = np.abs(x-xn)/np.abs(xn).max(); print('   worst idx', np.argsort(d)[-5:], d[np.argsort(d)[-5:]])
