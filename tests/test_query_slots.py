"""Dynamic batching of the query-lane kernels, host side (BASELINE configs[4]).

dcp_plan_query_slots is the plan dcp_gpu_scan makes for a batch: 64-query groups (one wavefront's lanes, consecutive
in the length order) packed into the wavefront slots of 256-lane blocks so that a block's slots finish together.
Pure host code: its invariants are checked here on the CPU; that the kernels compute the same bits through it is
tests/test_gpu_parity.py (every query-lane parity test runs through a plan) and test_gpu_configs.py (C5).
"""
import ctypes as C

import numpy as np
import pytest


def plan(dcp, lens, slots=4):
    lens = np.sort(np.asarray(lens, np.uint32))
    nq = len(lens)
    ng = (nq + 63) // 64
    nblocks, rows, plane_rows = C.c_uint(0), C.c_ulonglong(0), C.c_uint(0)
    groups = np.zeros((ng, 4), np.uint32)
    slot_first = np.zeros(ng * slots + slots + 1, np.uint32)
    f = dcp.lib.dcp_plan_query_slots
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint),
                  C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
    rc = f(lens.ctypes.data, nq, slots, C.byref(nblocks), C.byref(rows), C.byref(plane_rows), groups.ctypes.data, ng,
           slot_first.ctypes.data, len(slot_first))
    assert rc == 0, rc
    nb = nblocks.value
    return dict(lens=lens, nb=nb, cost=rows.value, plane_rows=plane_rows.value, groups=groups,
                slot_first=slot_first[:nb * slots + 1], slots=slots)


def group_rows(lmax):
    return (int(lmax) + 11) & ~1  # dcp_qlane_group_rows: the longest member + 10, even


def check_invariants(p):
    lens, groups, sf, slots, nb = p["lens"], p["groups"], p["slot_first"], p["slots"], p["nb"]
    nq = len(lens)
    # every query is in exactly one group; a group = up to 64 consecutive queries of the length order, full groups
    # from the long end (a partial group holds the SHORTEST queries: idle lanes cost the fewest rows)
    rem = nq % 64
    bounds = [0] + ([rem] if rem else []) + list(range(rem + 64, nq + 1, 64)) if nq > rem else [0, rem]
    firsts = sorted(int(g[0]) for g in groups)
    assert firsts == bounds[:-1], (firsts, bounds)
    for first, n, rowbase, lmax in groups:
        want_n = rem if (rem and first == 0) else 64
        assert n == want_n and lmax == lens[first + n - 1] and rowbase % 2 == 0
    # slots: consecutive group ranges; inside a slot the regions follow each other without overlap
    assert sf[0] == 0 and sf[-1] == len(groups) and (np.diff(sf.astype(np.int64)) >= 0).all()
    loads = []
    for s in range(nb * slots):
        at = 0
        for g in groups[sf[s]:sf[s + 1]]:
            assert g[2] == at
            at += group_rows(g[3])
        loads.append(at)
    loads = np.array(loads).reshape(nb, slots)
    assert p["plane_rows"] == loads.max()
    assert p["cost"] == loads.max(axis=1).sum()  # a block sits through a tile until its longest slot is done
    return loads


def consecutive_cost(lens, slots=4):
    """Rounds 1-3: block b = groups 4b .. 4b+3 of the length order, one group per slot."""
    lens = np.sort(lens)
    g = [group_rows(lens[min(len(lens), i + 64) - 1]) for i in range(0, len(lens), 64)]
    rem = len(lens) % 64  # the plan's own groups (full ones from the long end): what "perfectly even slots" is measured on
    starts = ([0] if rem else []) + list(range(rem, len(lens), 64))
    own = [group_rows(lens[min(len(lens), (s + 64) if (s or not rem) else rem) - 1]) for s in starts]
    return sum(max(g[i:i + slots]) for i in range(0, len(g), slots)), sum(own)


def test_uniform_batch_is_laid_out_as_before(dcp):
    """C3's step: 1 000 queries of 1 000 nt = 16 groups -> 4 blocks of 4 slots, one group each."""
    p = plan(dcp, np.full(1000, 1000))
    loads = check_invariants(p)
    assert p["nb"] == 4 and (loads == group_rows(1000)).all() and p["cost"] == 4 * group_rows(1000)
    assert (np.diff(p["slot_first"].astype(int)) == 1).all()
    p = plan(dcp, np.full(10000, 1000))  # the headline's literal 10 000-query batch: 157 groups in 40 blocks
    loads = check_invariants(p)
    assert p["nb"] == 40 and p["cost"] == 40 * group_rows(1000)


def test_mixed_lengths_are_balanced_across_a_block_s_slots(dcp, bench_mod):
    """C5's shape at 1 000 queries per step (log-uniform 100 nt .. 10 kbp): four consecutive groups per block leave
    the slots idle 40 % of the time; packed, the blocks' slots carry the same rows to within a few per cent."""
    lens = np.array([len(q) for q in bench_mod.make_queries(0, 1000, 0)], np.uint32)
    p = plan(dcp, lens)
    loads = check_invariants(p)
    old_cost, total = consecutive_cost(lens)
    ideal = total / 4.0
    assert old_cost / ideal > 1.35           # rounds 1-3: a tile cost 1.41 x what evenly loaded slots would take
    assert p["cost"] / ideal < 1.05          # now within 5 % of perfectly even slots: the longest group alone is a slot
    assert p["cost"] == group_rows(lens.max())
    assert p["nb"] == 1 and loads.min() > 0.9 * loads.max()
    # the longest group alone bounds a slot from below
    assert p["plane_rows"] >= group_rows(lens.max())


@pytest.mark.parametrize("seed", range(6))
def test_never_worse_than_consecutive_blocks(dcp, seed):
    rng = np.random.default_rng(seed)
    kind = seed % 3
    nq = int(rng.integers(1, 5000))
    if kind == 0:
        lens = rng.integers(1, 3000, nq)
    elif kind == 1:
        lens = np.round(np.exp(rng.uniform(np.log(20), np.log(50000), nq))).astype(np.int64)
    else:
        lens = np.concatenate([np.full(nq, 300), [200000]])  # one very long query among short ones
    p = plan(dcp, lens)
    check_invariants(p)
    old_cost, total = consecutive_cost(np.asarray(lens))
    assert p["cost"] <= old_cost
    assert p["cost"] * 4 >= total  # cannot beat perfectly even slots


def test_one_slot_per_block_variant_and_small_batches(dcp):
    """The 64-lane variant (one slot per block; the library uses it for <= 64 queries = one group)."""
    p = plan(dcp, [5, 9, 200], slots=1)
    check_invariants(p)
    assert p["nb"] == 1 and len(p["groups"]) == 1 and tuple(p["groups"][0]) == (0, 3, 0, 200)
    p = plan(dcp, np.arange(1, 301), slots=1)  # five groups: short ones share a slot, no slot exceeds the longest group's rows by much
    loads = check_invariants(p)
    assert p["cost"] == sum(group_rows(x) for x in (44, 108, 172, 236, 300)) and loads.max() <= group_rows(300) + group_rows(44)
    p = plan(dcp, [7])
    loads = check_invariants(p)
    assert p["nb"] == 1 and sorted(loads.ravel()) == [0, 0, 0, group_rows(7)]


def test_bad_input(dcp):
    f = dcp.lib.dcp_plan_query_slots
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint),
                  C.c_void_p, C.c_uint, C.c_void_p, C.c_uint]
    lens = np.array([5, 3], np.uint32)  # not ascending
    nb = C.c_uint(0)
    assert f(lens.ctypes.data, 2, 4, C.byref(nb), None, None, None, 0, None, 0) == dcp.RC_EINVAL
    lens = np.array([3, 5], np.uint32)
    assert f(lens.ctypes.data, 2, 3, C.byref(nb), None, None, None, 0, None, 0) == dcp.RC_EINVAL  # 1 or 4 slots
    assert f(lens.ctypes.data, 0, 4, C.byref(nb), None, None, None, 0, None, 0) == dcp.RC_EINVAL
    g = np.zeros(4, np.uint32)
    assert f(lens.ctypes.data, 2, 4, C.byref(nb), None, None, g.ctypes.data, 0, None, 0) == dcp.RC_ENOMEM
    assert f(lens.ctypes.data, 2, 4, C.byref(nb), None, None, g.ctypes.data, 1, None, 0) == 0 and nb.value == 1
