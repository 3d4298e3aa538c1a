"""world_size-2 gloo test of the multi-GPU plumbing (ldso_amd/dist.py): contiguous whole-point shards, the layout
of the all-reduce buffer, and that the SUM of the rank-local stitched systems equals the unsharded one.  The
rank-local sums are produced here by the oracle's explicit normal equations (no GPU in this container); on the
GPU the same invariance is asserted in test_ba_gpu.py::test_shard_and_sum_invariance."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ldso_amd import dist as ldist, synth


def test_shard_ranges_cover_points():
    for P in (1, 7, 64, 2000, 3001):
        for W in (1, 2, 3, 8):
            r = [ldist.shard_range(P, k, W) for k in range(W)]
            assert r[0][0] == 0 and r[-1][1] == P
            assert all(r[i][1] == r[i + 1][0] for i in range(W - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_reduce_layout():
    L = ldist.reduce_layout(7, 2000)
    assert L["n"] == 60 and L["size"] == 3 * (3600 + 60) + 8 + 2000
    buf = np.arange(L["size"], dtype=np.float64)
    u = ldist.unpack(buf, 7, 2000)
    assert u["HA"].shape == (60, 60) and u["bsc"].shape == (60,) and u["cand"].shape == (2000,)
    c = np.zeros(10); c[3] = 1 + 2.5; c[7] = 1 + 0.0
    assert np.array_equal(ldist.decode_candidates(c), np.array([2.5, 0.0], np.float32))


def _fast_worker(rank, world, port, q):
    """Fast layout (ldso_ba_gn_reduce_local): every rank contributes its share of the lower triangle of HFinal and of bFinal;
    the rank that owns point 0 also adds the prior / lambda terms (EnergyFunctional.cc:257-291)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po, spec_np as sp
    win = synth.add_synthetic_prior(synth.make_config("tiny"))
    o = po.OracleWindow(win)
    o.collect_active(); o.linearize_all(False); o.apply_res()
    r = o.get_residuals()
    adH, adT = sp.adjoints_np(win.frames)
    pb, pe = ldist.shard_range(win.P, rank, world)
    mine = (win.residuals["point"] >= pb) & (win.residuals["point"] < pe)
    ex = sp.explicit_system(win, r["J"], r["is_active"].astype(bool) & mine, adH, adT)
    Hpp = np.maximum(ex["Hpp"], 1e-10)
    own = np.zeros(win.P, bool); own[pb:pe] = True
    G = ex["Hcp"][:, own]
    Hsc = (G / Hpp[own][None, :]) @ G.T
    bsc = (G / Hpp[own][None, :]) @ ex["bp"][own]
    L = ldist.gn_reduce_layout(win.F, win.P)
    n = L["n"]
    lam = 1e-5                                   # FIX_LAMBDA
    l1, il = 1 + lam, float(np.float32(1.0) / np.float32(1 + lam))
    H = ex["Hcc"].copy()
    H[np.diag_indices(n)] *= l1
    H -= Hsc * il
    b = ex["bc"] - bsc
    if pb == 0:                                  # prior terms once
        fr = o.get_frames()
        prior = np.concatenate([np.full(4, float(win.settings["initialCalibHessian"])), fr["frames"]["prior"].ravel()])
        delta = np.concatenate([fr["calib_value"] - win.calib["value_zero"], (fr["frames"]["state"][:, :8] - fr["frames"]["state_zero"][:, :8]).ravel()])
        delta_prior = np.concatenate([fr["calib_value"] - win.calib["value_zero"], fr["frames"]["state"][:, :8].ravel()])
        Hp = win.HM + np.diag(prior)
        Hp[np.diag_indices(n)] *= l1
        H += Hp
        b += prior * delta_prior + win.bM + win.HM @ delta
    buf = np.zeros(L["size"])
    buf[L["HFinal_lower"][0]:L["HFinal_lower"][1]] = np.tril(H).ravel()
    buf[L["bFinal"][0]:L["bFinal"][1]] = b
    t = torch.from_numpy(buf)
    dist.all_reduce(t)
    if rank == 0:
        q.put(t.numpy().copy())
    dist.destroy_process_group()


def test_two_rank_fast_layout_sums_to_hfinal():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    summed = q.get(timeout=120)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    from oracle import pyoracle as po
    win = synth.add_synthetic_prior(synth.make_config("tiny"))
    o = po.OracleWindow(win)
    o.collect_active(); o.linearize_all(False); o.apply_res(); o.backup_state(); o.solve_system(0)
    sref = o.get_system()
    L = ldist.gn_reduce_layout(win.F, win.P)
    n = L["n"]
    Hl = summed[L["HFinal_lower"][0]:L["HFinal_lower"][1]].reshape(n, n)
    ref = np.tril(sref["HFinal"])
    assert np.abs(Hl - ref).max() <= 1e-6 * np.abs(ref).max()
    bf = summed[L["bFinal"][0]:L["bFinal"][1]]
    assert np.abs(bf - sref["bFinal"]).max() <= 1e-6 * np.abs(sref["bFinal"]).max()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po, spec_np as sp
    win = synth.make_config("tiny")
    o = po.OracleWindow(win)
    o.collect_active(); o.linearize_all(False); o.apply_res()
    r = o.get_residuals()
    adH, adT = sp.adjoints_np(win.frames)
    pb, pe = ldist.shard_range(win.P, rank, world)
    mine = (win.residuals["point"] >= pb) & (win.residuals["point"] < pe)
    ex = sp.explicit_system(win, r["J"], r["is_active"].astype(bool) & mine, adH, adT)
    Hpp = np.maximum(ex["Hpp"], 1e-10)
    own = np.zeros(win.P, bool); own[pb:pe] = True
    L = ldist.reduce_layout(win.F, win.P)
    n = L["n"]
    buf = np.zeros(L["size"])
    buf[L["HA"][0]:L["HA"][1]] = ex["Hcc"].ravel()
    buf[L["bA"][0]:L["bA"][1]] = ex["bc"]
    G = ex["Hcp"][:, own]
    buf[L["Hsc"][0]:L["Hsc"][1]] = ((G / Hpp[own][None, :]) @ G.T).ravel()
    buf[L["bsc"][0]:L["bsc"][1]] = (G / Hpp[own][None, :]) @ ex["bp"][own]
    e = r["out"]["state_NewEnergyWithOutlier"]
    tN = win.F - 1
    for i in np.nonzero(mine & (win.residuals["target"] == tN) & (e >= 0))[0]:
        buf[L["cand"][0] + win.residuals["point"][i]] = e[i] + 1.0
    t = torch.from_numpy(buf)
    dist.all_reduce(t)
    if rank == 0:
        q.put(t.numpy().copy())
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_unsharded():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    summed = q.get(timeout=120)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    from oracle import pyoracle as po
    win = synth.make_config("tiny")
    o = po.OracleWindow(win)
    o.collect_active(); o.linearize_all(False); o.apply_res(); o.backup_state(); o.solve_system(0)
    sref = o.get_system()
    u = ldist.unpack(summed, win.F, win.P)
    def r_(a, b): return np.abs(a - b).max() / np.abs(b).max()
    assert r_(u["HA"], sref["HA"]) < 1e-6 and r_(u["Hsc"], sref["Hsc"]) < 1e-6 and r_(u["bsc"], sref["bsc"]) < 1e-6
    e = o.get_residuals(False)["out"]["state_NewEnergyWithOutlier"]
    ref_c = np.sort(e[(win.residuals["target"] == win.F - 1) & (e >= 0)])
    assert np.array_equal(np.sort(ldist.decode_candidates(u["cand"])), ref_c.astype(np.float32))
