"""A key-frame SEQUENCE through the drop-in (tests/test_adapter_sequence_gpu.py, scripts/time_adapter.py): the synthetic scene of a window plus K
further frames; every further frame becomes a key frame in FullSystem::makeKeyFrame's order (adapter/adapter_capi.cc: adp_make_keyframe,
FullSystem.cc:410-640) - trace the immature points into it, insert it, add the new residuals of the old points, activate, optimize, remove outliers,
flag / drop / marginalise points, marginalise the oldest frame once the window holds `max_frames`, hand new immature points to the new frame."""
import numpy as np

from ldso_amd import synth
from oracle import pyref as pr


def noisy_pose(T_w2c, k, t_sigma=2e-3, r_sigma=2e-4):
    """what the coarse tracker would hand over: the true pose with a small deterministic error"""
    rng = np.random.default_rng(1000 + k)
    xi = np.concatenate([rng.normal(0, t_sigma, 3), rng.normal(0, r_sigma, 3)])
    return synth.se3_exp(xi) @ T_w2c


def run_sequence(win, K, adapter=None, max_frames=6, per_frame=120, iterations=6, on_keyframe=None, multithreading=False):
    """-> (RefWindow, list of per-key-frame records).  win = synth.make_config(name, extra_frames=K).  multithreading: the reference's own IndexThreadReduce
    (6 workers, include/internal/IndexThreadReduce.h) under its members - chunks go to whichever worker asks first, so its float sums differ from run to run."""
    r = pr.RefWindow(win)
    r.fs_attach(multithreading)
    F0 = win.F
    imm, _ = synth.make_immature_points(win, per_frame, seed=7)
    r.fs_add_immature(imm)
    log = []
    next_id = int(win.frames["frameID"].max()) + 1
    for k in range(K):
        img = win.images[F0 + k][0]
        T = noisy_pose(win.truth["w2c"][F0 + k], k)
        fh = r.fs_new_frame(img, T, float(win.truth["aff_a"][F0 + k]), float(win.truth["aff_b"][F0 + k]), 1.0)
        r.fs_set_frame_id(fh, next_id + k)
        nF = r.num_frames()
        marg = 0 if nF + 1 > max_frames else -1                     # flagFramesForMarginalization stand-in: the oldest frame once the window is full
        rmse, st = pr.make_keyframe(r, adapter, fh, marg, next_id + k, iterations)
        if not st["lost"]:
            # makeNewTraces stand-in (pixel selection is upstream of the hot path): fresh immature points on the new key frame
            hostIdx = r.num_frames() - 1
            imm, _ = synth.make_immature_points(win, per_frame, seed=100 + k, frames=[F0 + k], hosts=[hostIdx])
            r.fs_add_immature(imm)
        rec = dict(k=k, rmse=rmse, **st, summary=pr.graph_summary(r))
        if on_keyframe is not None:
            on_keyframe(rec)
        log.append(rec)
        if st["lost"]:
            break
    return r, log


QUANTITIES = ("rmse", "pose", "aff", "HM", "bM", "idepth_med", "idepth_max", "counts", "residual_counts", "unmatched_points")


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def sequence_distance(log_a, log_b):
    """How far two runs of the same key-frame sequence are apart - the worst value over the key frames of: rmse (relative), camToWorld (of the scene scale),
    affine parameters (absolute), the prior H_M / b_M (relative to their largest entry), inverse depths of the points both graphs hold (matched by host key
    frame and pixel; median and maximum), object counts (candidates / activated / new residuals / points per frame / immature per frame), residuals per
    frame, points held by one graph only.  Also returns whether the two runs kept the same key frames (they must)."""
    worst = {q: 0.0 for q in QUANTITIES}
    same_frames = len(log_a) == len(log_b)
    for a, b in zip(log_a, log_b):
        sa, sb = a["summary"], b["summary"]
        if a["lost"] or b["lost"] or sa["F"] != sb["F"] or not np.array_equal(sa["ids"], sb["ids"]):
            same_frames = False
            break
        dc = max(abs(a["candidates"] - b["candidates"]), abs(a["new_residuals"] - b["new_residuals"]), abs(a["activated"] - b["activated"]), abs(a["points"] - b["points"]),
                 int(np.abs(sa["points"] - sb["points"]).max()), int(np.abs(sa["immature"] - sb["immature"]).max()))
        worst["counts"] = max(worst["counts"], dc)
        worst["residual_counts"] = max(worst["residual_counts"], int(np.abs(sa["residuals"] - sb["residuals"]).max()))
        worst["rmse"] = max(worst["rmse"], abs(a["rmse"] - b["rmse"]) / a["rmse"])
        scale = np.abs(sa["c2w"][:, :, 3]).max()
        worst["pose"] = max(worst["pose"], float(np.abs(sa["c2w"] - sb["c2w"]).max() / max(scale, 1.0)))
        worst["aff"] = max(worst["aff"], float(np.abs(sa["aff"] - sb["aff"]).max()))
        if np.abs(sa["HM"]).max() > 0:
            worst["HM"] = max(worst["HM"], _rel(sb["HM"], sa["HM"])); worst["bM"] = max(worst["bM"], _rel(sb["bM"], sa["bM"]))
        ka = {(int(h), float(u), float(v)): float(d) for h, (u, v), d in zip(sa["host"], sa["uv"], sa["idepth"])}
        kb = {(int(h), float(u), float(v)): float(d) for h, (u, v), d in zip(sb["host"], sb["uv"], sb["idepth"])}
        both = sorted(set(ka) & set(kb))
        worst["unmatched_points"] = max(worst["unmatched_points"], len(set(ka) ^ set(kb)))
        if len(both) <= 0.97 * max(len(ka), len(kb)):
            same_frames = False
            break
        e = np.array([abs(ka[q] - kb[q]) / max(abs(ka[q]), 1e-3) for q in both])
        worst["idepth_max"] = max(worst["idepth_max"], float(e.max())); worst["idepth_med"] = max(worst["idepth_med"], float(np.median(e)))
    return worst, same_frames


def fast_reference_sequence(cfg, K):
    """The reference leg on the reference's translation units AT THEIR OWN optimisation level (oracle: `make ref_fast`, adapter: `make fast`; -O3, x86-64-v3,
    contraction on - every float sum of the pipeline rounds differently from the -O2 / no-contraction pin build), in a process of its own.  None where the
    libraries are missing or the host lacks AVX2 / FMA."""
    import os, pickle, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = os.path.join(root, "oracle", "_ref", "fast", "libldso_ref.so")
    adp = os.path.join(root, "adapter", "_build_fast", "libldso_adapter_test.so")
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        flags = ""
    if not (os.path.exists(ref) and os.path.exists(adp)) or " avx2" not in flags or " fma" not in flags:
        return None
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "log.pkl")
        env = dict(os.environ, LDSO_REF_LIB=ref, LDSO_ADAPTER_LIB=adp)
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "ref_sequence_worker.py"), cfg, str(K), "0", out], env=env, capture_output=True, timeout=900)
        if r.returncode != 0 or not os.path.exists(out):
            return None
        with open(out, "rb") as f:
            return pickle.load(f)


_YARD = {}


def reference_yardstick(cfg, K, log_ref=None, mt_runs=2):
    """The reference against ITSELF on the same key-frame sequence: the single-threaded pin build (log_ref) against (i) `mt_runs` runs with the reference's own
    6-worker IndexThreadReduce (IndexThreadReduce.h:126-139 hands chunks to whichever worker asks first: the per-thread accumulators sum in another order every
    run) and (ii) the -O3 build of the same translation units.  -> (per quantity the LARGEST distance between two reference runs, the individual distances).
    A drop-in whose distance to the reference is a small multiple of this spread is as close to the reference as the reference is to itself."""
    key = (cfg, K, mt_runs)
    if key in _YARD:
        return _YARD[key]
    win = synth.make_config(cfg, extra_frames=K)
    if log_ref is None:
        r, log_ref = run_sequence(win, K); r.close()
    runs = {}
    for i in range(mt_runs):
        r, log = run_sequence(win, K, multithreading=True); r.close()
        runs["six_threads_run_%d" % i] = log
    fast = fast_reference_sequence(cfg, K)
    if fast is not None:
        runs["O3_build"] = fast
    per = {}
    for name, log in runs.items():
        d, same = sequence_distance(log_ref, log)
        assert same, ("two runs of the reference disagree on the key-frame set", name)
        per[name] = d
    yard = {q: max(d[q] for d in per.values()) for q in QUANTITIES}
    _YARD[key] = (yard, per)
    return yard, per
