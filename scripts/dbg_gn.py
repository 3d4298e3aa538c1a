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
    print('block0 [us]: reaches the wait', (buf[47] - t0) / 100, 'wait over', (buf[48] - t0) / 100, 'loads done', (buf[41] - t0) / 100, 'factor done', (buf[42] - t0) / 100, 'backsub done', (buf[43] - t0) / 100, 'core done', buf[44] / 100, 'canbreak', buf[45] / 100, 'precalc', buf[46] / 100)
    print('k_reduce [us]: A start', buf[22]/100, 'A partials in', buf[23]/100, 'A loads', buf[20]/100, 'A end', buf[21]/100, '| B loads', buf[22]/100, 'B mfma', buf[23]/100, 'B end', buf[24]/100)
    print('last reduce workgroup: bid', int(buf[30]), 'signals at', (buf[29] - t0) / 100, 'us after the control workgroup started; it had started at', (buf[31] - t0) / 100, 'us')
    print('k_reduce part B tile 0 [us]: staged', buf[25]/100, 'mfma done', buf[26]/100, 'end', buf[27]/100, '| extras end', buf[28]/100)
    print('k_linearize block0 wave0 [us]: staged', buf[9]/100, 'pt0 start', buf[10]/100, 'before taps', buf[11]/100, 'taps in', buf[12]/100, 'pt0 slots done', buf[13]/100, 'points done', buf[14]/100, 'end', buf[15]/100)
    print('block1 [us]: sums', buf[50] / 100, 'counts', buf[51] / 100, 'thresh', buf[52] / 100)
