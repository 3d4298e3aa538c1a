"""One rank of the multi-process tests of the one-shot peer-write all-reduce (tests/test_p2p_gpu.py):
    python p2p_worker.py <rank> <dir> [<ranks> = 2] [<config> = small]
All processes share GPU 0 (what ranks on different GPUs of an xGMI node do, minus the fabric); the receive windows cross the process boundary as
hipIpcMemHandle_t (files in <dir>).  Every rank owns the contiguous whole-point shard ldso_amd.dist.shard_range gives it, runs four forced
Gauss-Newton iterations through ldso_ba_enqueue_gn_p2p and leaves state<rank>.npy (frame states), idepth<rank>.npy (all points; its shard is
valid), shard<rank>.npy = [begin, end, rank-local energy of the final state]."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ldso_amd import synth, binding, dist as ldist


def wait_for(path, timeout=180.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise SystemExit(f"rank timed out waiting for {path}")
        time.sleep(0.01)


def main():
    rank, d = int(sys.argv[1]), sys.argv[2]
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    cfg = sys.argv[4] if len(sys.argv) > 4 else "small"
    win = synth.add_synthetic_prior(synth.make_config(cfg))
    pb, pe = ldist.shard_range(win.P, rank, N)
    g = binding.BA.from_window(win)
    g.set_shard(pb, pe)
    own, hnd = g.p2p_window_alloc(N, with_ipc_handle=True)
    with open(os.path.join(d, f"h{rank}.tmp"), "wb") as f:
        f.write(hnd)
    os.rename(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}.bin"))
    windows, opened = [], []
    for q in range(N):
        if q == rank:
            windows.append(own)
            continue
        wait_for(os.path.join(d, f"h{q}.bin"))
        w = g.p2p_window_open(open(os.path.join(d, f"h{q}.bin"), "rb").read())
        windows.append(w); opened.append(w)
    g.collect_active(); g.linearize_all(False); g.apply_res()
    open(os.path.join(d, f"ready{rank}"), "w").close()
    for q in range(N):
        wait_for(os.path.join(d, f"ready{q}"))
    g.enqueue_gn_p2p(rank, N, windows, 0, 2)
    g.enqueue_gn_p2p(rank, N, windows, 2, 2)          # a second call continues the exchange numbering
    g.sync(); g.p2p_check()
    np.save(os.path.join(d, f"state{rank}.npy"), g.get_frames()["frames"]["state"])
    np.save(os.path.join(d, f"idepth{rank}.npy"), g.get_points()["idepth"])
    # rank-local energy of the final state: the scalar block behind HFinal | bFinal of the rank's own reduce
    import torch
    n = 8 * win.F + 4
    buf = torch.zeros(g.gn_reduce_doubles(), dtype=torch.float64, device="cuda")
    g.gn_reduce_local(buf.data_ptr(), 1e-1); g.sync(); torch.cuda.synchronize()
    host = buf.cpu().numpy()
    np.save(os.path.join(d, f"shard{rank}.npy"), np.array([pb, pe, host[n * n + n]], np.float64))
    open(os.path.join(d, f"done{rank}"), "w").close()
    for q in range(N):
        wait_for(os.path.join(d, f"done{q}"))          # keep the windows mapped until every peer has finished with them
    for w in opened:
        g.p2p_window_close(w, opened_from_handle=True)


if __name__ == "__main__":
    main()
