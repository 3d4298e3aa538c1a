"""Multi-GPU plumbing of the windowed BA: contiguous whole-point shards and the layout of the all-reduce buffer.

The window is replicated on every rank (images, frame states); rank r linearises / accumulates only the points of
its shard and contributes rank-local sums.  One all-reduce (sum, fp64) per Gauss-Newton iteration carries
[H_A | b_A | H_L | b_L | H_sc | b_sc | 8 scalars | P newest-frame energy candidates]; the solve is replicated.
"""
from __future__ import annotations

import numpy as np


def shard_range(P: int, rank: int, world: int):
    """Contiguous range of EnergyFunctional::allPoints owned by `rank` (whole points only)."""
    base, rem = divmod(P, world)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)


def reduce_layout(F: int, P: int):
    n = 8 * F + 4
    blk = n * n + n
    return dict(n=n, HA=(0, n * n), bA=(n * n, blk), HL=(blk, blk + n * n), bL=(blk + n * n, 2 * blk), Hsc=(2 * blk, 2 * blk + n * n),
                bsc=(2 * blk + n * n, 3 * blk), scalars=(3 * blk, 3 * blk + 8), cand=(3 * blk + 8, 3 * blk + 8 + P), size=3 * blk + 8 + P)


def gn_reduce_layout(F: int, P: int):
    """Fast path (ldso_ba_gn_reduce_local): the all-reduce buffer is the HFinal / bFinal accumulator itself."""
    n = 8 * F + 4
    blk = n * n + n
    return dict(n=n, HFinal_lower=(0, n * n), bFinal=(n * n, blk), scalars=(blk, blk + 8), cand=(blk + 8, blk + 8 + P), size=blk + 8 + P)


def unpack(buf: np.ndarray, F: int, P: int):
    L = reduce_layout(F, P)
    n = L["n"]
    out = {}
    for k in ("HA", "HL", "Hsc"):
        out[k] = buf[L[k][0]:L[k][1]].reshape(n, n)
    for k in ("bA", "bL", "bsc", "scalars", "cand"):
        out[k] = buf[L[k][0]:L[k][1]]
    return out


def decode_candidates(cand: np.ndarray):
    """value+1 encoding -> float32 energies of the residuals that target the newest frame."""
    c = cand[cand > 0] - 1.0
    return c.astype(np.float32)
