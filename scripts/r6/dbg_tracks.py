import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
from oracle import pyref as pr
from tracker_common import tracker_scenario
import test_adapter_threads_gpu as T
sc = tracker_scenario("small")
keep = pr.RefWindow(sc["win"])
A = pr.GpuAdapter(max_frames=8, max_points=4000)
runs = []
for rep in range(3):
    rts = T._tracker_pair(sc)
    runs.append([T._track(A, rts[i % 2], sc) for i in range(6)])
    for t in rts: t.close()
for rep in (1, 2):
    for i in range(6):
        d = np.nan_to_num(np.abs(runs[rep][i] - runs[0][i]))
        print("rep", rep, "call", i, "max diff", d.max(), "at", int(d.argmax()), "values", runs[0][i][int(d.argmax())], runs[rep][i][int(d.argmax())])
print("within a run: call 0 vs 2 (same tracker):", np.nanmax(np.abs(runs[0][0] - runs[0][2])), "call 0 vs 1 (other tracker):", np.nanmax(np.abs(runs[0][0] - runs[0][1])))
print("layout: result4 [0:4], w2c [4:16], aff [16:18], rmse [18:23]")
