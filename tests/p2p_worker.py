"""One rank of the two-process test of the one-shot peer-write all-reduce (tests/test_p2p_gpu.py): python p2p_worker.py <rank> <dir>.
The two processes share GPU 0; the receive windows cross the process boundary as hipIpcMemHandle_t (files in <dir>)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from ldso_amd import synth, binding


def wait_for(path, timeout=60.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise SystemExit(f"rank timed out waiting for {path}")
        time.sleep(0.01)


def main():
    rank, d = int(sys.argv[1]), sys.argv[2]
    win = synth.add_synthetic_prior(synth.make_config("small"))
    half = win.P // 2
    g = binding.BA.from_window(win)
    g.set_shard(*((0, half) if rank == 0 else (half, win.P)))
    own, hnd = g.p2p_window_alloc(2, with_ipc_handle=True)
    with open(os.path.join(d, f"h{rank}.tmp"), "wb") as f:
        f.write(hnd)
    os.rename(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}.bin"))
    wait_for(os.path.join(d, f"h{1 - rank}.bin"))
    peer = g.p2p_window_open(open(os.path.join(d, f"h{1 - rank}.bin"), "rb").read())
    windows = [own, peer] if rank == 0 else [peer, own]
    g.collect_active(); g.linearize_all(False); g.apply_res()
    open(os.path.join(d, f"ready{rank}"), "w").close()
    wait_for(os.path.join(d, f"ready{1 - rank}"))
    g.enqueue_gn_p2p(rank, 2, windows, 0, 4)
    g.sync(); g.p2p_check()
    np.save(os.path.join(d, f"state{rank}.npy"), g.get_frames()["frames"]["state"])
    np.save(os.path.join(d, f"idepth{rank}.npy"), g.get_points()["idepth"])
    open(os.path.join(d, f"done{rank}"), "w").close()
    wait_for(os.path.join(d, f"done{1 - rank}"))          # keep the window mapped until the peer has finished with it
    g.p2p_window_close(peer, opened_from_handle=True)


if __name__ == "__main__":
    main()
