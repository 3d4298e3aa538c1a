"""Which reduce workgroup of k_reduce_solve signals last, and when (stamps build: LDSO_HIP_LIB=libldso_hip_stamps.so)."""
import sys
import numpy as np
sys.path.insert(0, '.')
import ctypes as C
from ldso_amd import synth, binding
win = synth.add_synthetic_prior(synth.make_config(sys.argv[1] if len(sys.argv) > 1 else 'C3'))
g = binding.BA.from_window(win)
g.collect_active(); g.linearize_all(False); g.apply_res()
g.enqueue_gn(0, 10); g.sync()
rows = []
for rep in range(24):
    g.enqueue_gn(2, 1); g.sync()
    buf = np.zeros(64)
    g.L.ldso_ba_get_energy_log(g.h, buf.ctypes.data_as(C.c_void_p), C.c_int(64))
    t0 = buf[39]
    rows.append((int(buf[30]), round((buf[29] - t0) / 100, 2), round((buf[31] - t0) / 100, 2), round((buf[48] - t0) / 100, 2), round((buf[41] - t0) / 100, 2), round((buf[42] - t0) / 100, 2)))
print("(last bid, signals at, had started at, wait over, loads done, factor done) us after the control workgroup started")
for r in rows: print(r)
