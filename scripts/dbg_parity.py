"""Debug: how do the GPU step x and the states after k fast-path iterations differ from the oracle's, and in which norm?"""
import sys, copy
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from ldso_amd import synth, binding
from oracle import pyoracle as po

def nullspace_basis(fr):
    F = len(fr); N = np.zeros((8 * F, 7))
    for f in range(F):
        P = fr["nullspaces_pose"][f].reshape(6, 6)
        for i in range(6):
            v = P[:, i].copy(); v[:3] *= 2.0
            N[8 * f:8 * f + 6, i] = v
        s = fr["nullspaces_scale"][f].copy(); s[:3] *= 2.0
        N[8 * f:8 * f + 6, 6] = s
    return N

for name in sys.argv[1:] or ["C3"]:
    win = synth.add_synthetic_prior(copy.deepcopy(synth.make_config(name)))
    o = po.OracleWindow(win); g = binding.BA.from_window(win)
    for w in (o, g):
        w.collect_active(); w.linearize_all(False); w.apply_res(); w.backup_state(); w.solve_system(0)
    so, sg = o.get_system(), g.get_system()
    H, b, xo, xg = so["HFinal"], so["bFinal"], so["x"], sg["x"]
    dx = xg - xo
    print(name, "x rel", np.abs(dx).max() / np.abs(xo).max(), "H blockrel-ish", np.abs(sg["HFinal"] - H).max() / np.abs(H).max(), "b rel", np.abs(sg["bFinal"] - b).max() / np.abs(b).max())
    print("  backward err of x_g in oracle system", np.linalg.norm(H @ xg - b) / np.linalg.norm(b), " x_o:", np.linalg.norm(H @ xo - b) / np.linalg.norm(b))
    print("  energy-norm rel", np.sqrt(abs(dx @ H @ dx) / abs(xo @ H @ xo)))
    mo, mg = 2 * b @ xo - xo @ H @ xo, 2 * b @ xg - xg @ H @ xg
    print("  model decrease rel diff", abs(mo - mg) / abs(mo))
    w_, V = np.linalg.eigh(H)
    c = V.T @ dx
    print("  eig range", w_[0], w_[-1], " dx along 5 weakest eigvecs", np.abs(c[:5]), "of |dx|", np.linalg.norm(dx))
    S = 1 / np.sqrt(np.diag(H) + 10); Hs = H * S[:, None] * S[None, :]
    print("  cond scaled", np.linalg.cond(Hs))
    N = nullspace_basis(o.get_frames()["frames"]); Q, _ = np.linalg.qr(N)
    for k in (1, 2, 3, 5, 10):
        o2 = po.OracleWindow(win); g2 = binding.BA.from_window(win)
        for w in (o2, g2):
            w.collect_active(); w.linearize_all(False); w.apply_res()
        for it in range(k):
            o2.backup_state(); o2.solve_system(it); o2.do_step(); Eo = o2.linearize_all(False); o2.apply_res()
        g2.enqueue_gn(0, k); g2.sync()
        fo, fg = o2.get_frames()["frames"]["state"][:, :8].reshape(-1), g2.get_frames()["frames"]["state"][:, :8].reshape(-1)
        d = fg - fo; dp = d - Q @ (Q.T @ d)
        ido, idg = o2.get_points()[0]["idepth"], g2.get_points()["idepth"]
        print("  k=%d state rel %.3e gauge-projected %.3e idepth rel %.3e" % (k, np.abs(d).max() / np.abs(fo).max(), np.abs(dp).max() / np.abs(fo).max(), np.abs(idg - ido).max() / np.abs(ido).max()))
