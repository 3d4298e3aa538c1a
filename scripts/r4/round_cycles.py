"""Where a round of the 4-column factorisation and the pair precalc of k_reduce_solve spend their time (stamps build: LDSO_HIP_LIB=libldso_hip_stamps.so)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import ctypes as C
from ldso_amd import synth, binding
win = synth.add_synthetic_prior(synth.make_config(sys.argv[1] if len(sys.argv) > 1 else 'C3'))
g = binding.BA.from_window(win)
g.collect_active(); g.linearize_all(False); g.apply_res()
g.enqueue_gn(0, 10); g.sync()
print("cycles per round (wave 0): panel read + 4x4 block + multipliers | - | stores | read-back + matrix cores + publish | barrier ;  us after the control workgroup "
      "started: loads done, factor done, back-substitution done | tail: poses done (wave 0), wave 0 at the barrier (pair records issued), wave 1 there, wave 3 there, barrier over")
for rep in range(12):
    g.enqueue_gn(2, 1); g.sync()
    buf = np.zeros(64)
    g.L.ldso_ba_get_energy_log(g.h, buf.ctypes.data_as(C.c_void_p), C.c_int(64))
    t0 = buf[39]
    rounds = (win.F * 8 + 4 + 3) // 4
    cyc = [int(buf[53 + u] / rounds) for u in range(5)]
    f = lambda i: round((buf[i] - t0) / 100, 2)
    print(cyc, sum(cyc), [f(41), f(42), f(43)], [f(59), f(62), f(58), f(63), round(buf[46] / 100, 2)])
