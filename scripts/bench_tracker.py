"""CoarseTracker::trackNewestCoarse timing, BASELINE config C2 (640x480 pair, 5 pyramid levels): ldso_tr_track on the GPU
(images resident, includes the host round trip of one call) vs the oracle on one host core.  Run on the GPU box."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from tracker_common import tracker_scenario
from ldso_amd import synth, binding
from oracle import pyoracle as po

def main():
    sc = tracker_scenario("C3", levels=5)
    win = sc["win"]
    a, b = sc["new_aff"]
    g = binding.Tracker(win.w, win.h, sc["levels"], win.settings, win.calib)
    o = po.OracleTracker(win.w, win.h, sc["levels"], win.settings, win.calib, fast=True) if hasattr(po, 'build') else None
    for t in (g, o):
        t.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"])
        t.set_new_frame(sc["new_pyr"], 1.0)
    def timeit(t, n):
        r = None
        t0 = time.perf_counter()
        for _ in range(n): r = t.track(np.eye(4), a, b, sc["levels"] - 1)
        return (time.perf_counter() - t0) / n, r
    timeit(g, 3)
    tg, rg = timeit(g, 50)
    to, ro = timeit(o, 5)
    # batch of 20 motion hypotheses in one launch (FullSystem::trackNewCoarse tries up to 83)
    guesses = [synth.se3_exp([0.002 * i, 0, 0, 0, 0.0005 * i, 0]) for i in range(20)]
    g.track_batch(guesses, [(a, b)] * 20, sc["levels"] - 1)
    t0 = time.perf_counter()
    for _ in range(10): g.track_batch(guesses, [(a, b)] * 20, sc["levels"] - 1)
    tb = (time.perf_counter() - t0) / 10
    # set_ref (makeCoarseDepthL0) + set_new_frame (H2D of the pyramid) cost
    t0 = time.perf_counter()
    for _ in range(10): g.set_ref(sc["ref_pyr"], sc["ref_aff"][0], sc["ref_aff"][1], 1.0, sc["pts"])
    ts = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): g.set_new_frame(sc["new_pyr"], 1.0)
    tn = (time.perf_counter() - t0) / 10
    print(json.dumps({"workload": "C2: 640x480 pair, 5 levels, %d reference points" % len(sc["pts"]),
                      "gpu_track_ms": round(tg * 1e3, 4), "gpu_iterations": int(rg["iterations"]), "gpu_track_batch20_ms": round(tb * 1e3, 4),
                      "gpu_set_ref_ms": round(ts * 1e3, 4), "gpu_set_new_frame_ms": round(tn * 1e3, 4),
                      "cpu_oracle_track_ms": round(to * 1e3, 4), "cpu_iterations": int(ro["iterations"]), "cpu_cores": 1}))
main()
