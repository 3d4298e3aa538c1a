"""What a maintainer sees: wall time of ldso::GpuBackend::optimize(6) against the reference's own FullSystem::optimize(6) on the SAME reference
object graph (C3: 7 key frames x 2000 points, the reference at -O2 with the header shim's Eigen: a floor for its speed), with the adapter's time split
into flatten + upload | device | fetch | write-back into the objects.  Prints one JSON line."""
import copy, json, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ldso_amd import synth
from oracle import pyref as pr

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
win = synth.add_synthetic_prior(synth.make_config(name))
out = {"workload": f"{name}: {win.F} KF x {win.P} pt, R = {win.R}; optimize(6) on the reference's object graph"}
A = pr.GpuAdapter(max_frames=win.F + 1, max_points=win.P + 16)
for wb in (False, True):
    A.set_write_back_jacobians(wb)
    ts, splits, its = [], [], 0
    for rep in range(6):
        r = pr.RefWindow(win)
        t0 = time.perf_counter(); rv, its, lost = A.optimize(r, 6); ts.append(time.perf_counter() - t0); splits.append(A.last_optimize_times().copy())
        r.close()
    sp = np.median(np.array(splits[1:]), axis=0)
    out["adapter_writeBackJacobians_" + ("on" if wb else "off")] = {"optimize_ms": round(float(np.median(ts[1:])) * 1e3, 3), "iterations_executed": its,
        "split_ms": {"flatten_upload": round(sp[0] * 1e3, 3), "device": round(sp[1] * 1e3, 3), "fetch": round(sp[2] * 1e3, 3), "write_back": round(sp[3] * 1e3, 3)}}
tr = []
for rep in range(3):
    r = pr.RefWindow(win); r.fs_attach()
    t0 = time.perf_counter(); r.fs_optimize(6); tr.append(time.perf_counter() - t0)
    r.close()
out["reference_FullSystem_optimize_ms"] = round(float(np.median(tr)) * 1e3, 3)
out["reference_build"] = "reference translation units, g++ -O2, Eigen = oracle/ref_shim (eager evaluation): a floor"
A.close()
print(json.dumps(out))
