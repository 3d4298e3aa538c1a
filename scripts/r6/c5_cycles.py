"""Where a workgroup of the C5 linearisation (k_linearize_one<2>) spends its time: -DLDSO_STAMPS build, wave 0 of chunk 0."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import numpy as np
from ldso_amd import synth, binding
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
win = synth.make_config(cfg)
g = binding.BA.from_window(win); g.collect_active(); g.linearize_all(False); g.apply_res()
print(cfg, "chunk points / workgroups", g.get_chunk_points())
g.enqueue_gn(0, 4); g.sync()
for rep in range(3):
    g.enqueue_gn(4, 1); g.sync()
    buf = np.zeros(64)
    g.L.ldso_ba_get_energy_log(g.h, buf.ctypes.data_as(C.c_void_p), C.c_int(64))
    st = buf[9:16] / 100.0          # 10 ns ticks -> us
    c = buf[24:30]; n = max(c[5], 1)
    print("stamps us: staged %.2f | points done %.2f | reduction barrier %.2f" % (st[0], st[5], st[6]), "| points of wave 0: %d, cycles per point:" % n, [int(v / n) for v in c[1:5]], "sum", int(c[1:5].sum() / n))
