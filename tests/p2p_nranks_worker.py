"""N ranks of the sharded Gauss-Newton iteration in ONE process on ONE GPU (tests/test_p2p_gpu.py): python p2p_nranks_worker.py <config> <ranks> <out.npz>
[--rccl-layout].  Every rank is its own handle on its own stream and owns a contiguous whole-point shard (ldso_amd.dist.shard_range: uneven
when P is not a multiple of the rank count); the exchange is the library's one-shot peer-write all-reduce (ldso_ba_enqueue_gn_p2p), or - with
--hand - the all-reduce formed by hand in rank order around ldso_ba_gn_reduce_local / ldso_ba_gn_solve_reduced (what bench.py does around
torch.distributed).  The parent sets GPU_MAX_HW_QUEUES >= ranks: a polling kernel must never sit in front of a peer's push in one hardware queue.
Writes every rank's frame states, the merged inverse depths, the total energy at the final state, and the same for the unsharded handle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from ldso_amd import synth, binding, dist as ldist

ITERS = 4


def total_energy(handles, n):
    e = 0.0
    for g in handles:
        b = torch.zeros(g.gn_reduce_doubles(), dtype=torch.float64, device="cuda")
        g.gn_reduce_local(b.data_ptr(), 1e-1); g.sync(); torch.cuda.synchronize()
        e += float(b[n * n + n].item())
    return e


def main():
    cfg, N, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    hand = "--hand" in sys.argv
    win = synth.add_synthetic_prior(synth.make_config(cfg))
    n = 8 * win.F + 4
    ref = binding.BA.from_window(win)
    ref.collect_active(); ref.linearize_all(False); ref.apply_res()
    ref.enqueue_gn(0, ITERS); ref.sync()
    ranks, shards = [], []
    ts = torch.cuda.Stream() if hand else None
    for r in range(N):
        a, b = ldist.shard_range(win.P, r, N)
        g = binding.BA.from_window(win, stream=ts.cuda_stream) if hand else binding.BA.from_window(win)
        g.set_shard(a, b); g.collect_active(); g.linearize_all(False); g.apply_res()
        ranks.append(g); shards.append((a, b))
    if hand:
        torch.cuda.set_stream(ts)
        bufs = [torch.zeros(g.gn_reduce_doubles(), dtype=torch.float64, device="cuda") for g in ranks]
        for it in range(ITERS):
            for g, b in zip(ranks, bufs):
                g.gn_reduce_local(b.data_ptr(), 1e-1)
            tot = bufs[0].clone()
            for b in bufs[1:]:
                tot += b
            for g, b in zip(ranks, bufs):
                b.copy_(tot); g.gn_solve_reduced(b.data_ptr(), it, 1e-1)
        torch.cuda.synchronize()
    else:
        windows = [g.p2p_window_alloc(N) for g in ranks]
        for it0 in (0, ITERS // 2):                          # two calls: the second continues the exchange numbering
            for r, g in enumerate(ranks):
                g.enqueue_gn_p2p(r, N, windows, it0, ITERS // 2)
        for g in ranks:
            g.sync(); g.p2p_check()
    states = np.stack([g.get_frames()["frames"]["state"] for g in ranks])
    idepth = ref.get_points()["idepth"].copy() * 0
    for g, (a, b) in zip(ranks, shards):
        idepth[a:b] = g.get_points()["idepth"][a:b]
    fr = ref.get_frames()["frames"]
    np.savez(out, states=states, idepth=idepth, energy=total_energy(ranks, n), ref_state=fr["state"], ref_idepth=ref.get_points()["idepth"],
             ref_energy=total_energy([ref], n), ns_pose=fr["nullspaces_pose"], ns_scale=fr["nullspaces_scale"], shards=np.array(shards))
    if not hand:
        for g, w in zip(ranks, windows):
            g.p2p_window_close(w)


if __name__ == "__main__":
    main()
