"""Host logic of the batched launch (no device): ldso_ba_balance_chunks (include/ldso_hip.h) - how ldso_ba_batch_create cuts the windows of a batch so that every workgroup
of the batched linearisation (one per CU) works through a run of consecutive chunks carrying the same load.  A chunk = one pass of the kernel's block code (operand staging,
pipeline fill, block reduction): it costs `chunk_cost` points on top of its own, and it never spans two segments (a segment = one window's points of one host frame)."""
import ctypes as C

import numpy as np
import pytest

from ldso_amd import binding

WAVES = 8          # wavefronts per workgroup of the linearisation kernel (LD_WAVES)


def balance(seg, n_wg, cost):
    L = binding.lib()
    L.ldso_ba_balance_chunks.restype = C.c_int
    seg = np.ascontiguousarray(seg, np.int32)
    cap = int(len(seg) + n_wg + 8)
    ends = np.zeros(cap, np.int32); wg = np.zeros(n_wg + 1, np.int32); budget = C.c_int64()
    n = L.ldso_ba_balance_chunks(C.c_int(len(seg)), seg.ctypes.data_as(C.c_void_p), C.c_int(n_wg), C.c_int(cost), ends.ctypes.data_as(C.c_void_p), C.c_int(cap),
                                 wg.ctypes.data_as(C.c_void_p), C.byref(budget))
    assert n > 0, binding.lib().ldso_last_error()
    return ends[:n].copy(), wg, int(budget.value)


def check(seg, n_wg, cost):
    ends, wg, budget = balance(seg, n_wg, cost)
    total = int(np.sum(seg))
    # every point in exactly one chunk, in order
    assert ends[-1] == total and np.all(np.diff(ends) > 0)
    # a chunk never spans two segments: every segment boundary is a chunk boundary
    bounds = np.cumsum(seg)
    assert np.all(np.isin(bounds, ends))
    # the workgroups partition the chunk list in order
    assert wg[0] == 0 and wg[-1] == len(ends) and np.all(np.diff(wg) >= 0)
    sizes = np.diff(np.concatenate([[0], ends]))
    loads = np.array([int(np.sum(sizes[wg[w]:wg[w + 1]])) + cost * int(wg[w + 1] - wg[w]) for w in range(n_wg)])
    # no workgroup above the budget by more than one round of its wavefronts (a chunk is cut in whole rounds; a segment's crumbs stay with its last chunk).
    # With more workgroups than work the budget falls below what the smallest possible chunk carries (its cost + a round): that chunk is the floor.
    assert loads.max() <= max(budget, cost) + 2 * WAVES, (loads.max(), budget)
    # the budget is tight: close to the ideal share of the work (the greedy cut wastes at most a chunk's cost + a round per workgroup)
    ideal = (total + cost * len(ends)) / n_wg
    assert budget <= ideal + cost + 2 * WAVES + 1, (budget, ideal)
    return ends, wg, loads, budget


def test_thirty_two_c3_windows_on_224_workgroups():
    """the bench's batch: 16 windows (one half-batch) x 7 host frames x 285-286 points on 224 workgroups"""
    seg = [286 if h < 5 else 285 for _ in range(16) for h in range(7)]
    ends, wg, loads, budget = check(seg, 224, 16)
    busy = loads[loads > 0]
    assert len(busy) >= 0.95 * 224, "almost every workgroup gets work"
    assert busy[:-1].min() >= 0.8 * budget, "and the same amount of it (the last one takes what is left)"
    sizes = np.diff(np.concatenate([[0], ends]))
    assert np.median(sizes) >= 8 * WAVES, "chunks stay long enough for the pipeline (>= 8 points per wavefront)"


@pytest.mark.parametrize("n_wg,cost", [(1, 16), (7, 0), (64, 16), (256, 16), (256, 40), (1000, 16)])
def test_invariants_over_ragged_segments(n_wg, cost):
    rng = np.random.default_rng(7 + n_wg + cost)
    seg = rng.integers(1, 700, size=83)
    check(seg, n_wg, cost)


def test_single_short_segment_and_more_workgroups_than_points():
    ends, wg, loads, budget = check([5], 256, 16)
    assert len(ends) == 1 and ends[0] == 5 and (loads > 0).sum() == 1
    ends, wg, loads, budget = check([3, 2, 9], 64, 0)
    assert ends[-1] == 14
