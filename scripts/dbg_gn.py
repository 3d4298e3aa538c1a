"""Device wall-clock stamps (10 ns ticks) of k_gn_solve on a window. Run on the GPU box."""
import sys
import numpy as np
sys.path.insert(0, '.')
from ldso_amd import synth, binding
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
win = synth.make_config(cfg)
g = binding.BA.from_window(win)
g.collect_active(); g.linearize_all(False); g.apply_res()
for rep in range(3):
    g.enqueue_gn(rep, 4); g.sync()
    g.get_energy_log()
    d = g._dbg
    buf = np.zeros(64); 
    import ctypes as C
    g.L.ldso_ba_get_energy_log(g.h, buf.ctypes.data_as(C.c_void_p), C.c_int(64))
    t0 = buf[39]
    print('block0 [us]: loads done', (buf[41] - t0) / 100, 'factor done', (buf[42] - t0) / 100, 'backsub done', (buf[43] - t0) / 100, 'core done', buf[44] / 100, 'canbreak', buf[45] / 100, 'precalc', buf[46] / 100)
    print('round k=8 cycles: wave0 p1', buf[27]-buf[26], 'p2', buf[28]-buf[27], '| wave1 wait', buf[30]-buf[29], 'p2', buf[31]-buf[30], '| wave0 start->wave1 p2 end', buf[31]-buf[26])
    print('k_reduce [us]: A loads', buf[20]/100, 'A end', buf[21]/100, '| B loads', buf[22]/100, 'B mfma', buf[23]/100, 'B end', buf[24]/100)
    print('block1 [us]: sums', buf[50] / 100, 'counts', buf[51] / 100, 'thresh', buf[52] / 100)
