"""Multi-GPU layer on CPU: world_size-2 gloo processes.

The scan itself needs a GPU (no CPU fallback), so here the N>1 logic is covered with what does
not: shard maps agree across ranks and tile the DB; the hit gather returns the union of every
rank's records, remapped to global profile indices and sorted, identically on all ranks."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_product

    dcp = load_product()
    from deciphon_old_amd import dist as ddist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(7)
        sizes = np.clip(np.round(np.exp(rng.normal(np.log(150), 0.6, 999))), 30, 2000).astype(np.uint32)
        b, e = ddist.shard_range(sizes, world, rank)
        bounds = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(bounds, torch.tensor([b, e]))
        bounds = [tuple(int(x) for x in t) for t in bounds]
        assert bounds[0][0] == 0 and bounds[-1][1] == len(sizes)
        assert all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
        loads = [int(sizes[x:y].sum()) for x, y in bounds]
        assert max(loads) - min(loads) <= 2 * int(sizes.max())

        # every rank fabricates the hits "its shard" produced (shard-local profile indices)
        def local_hits(r, n):
            g = np.random.default_rng(100 + r)
            h = np.zeros(n, dcp.HIT_DTYPE)
            h["seq_idx"] = g.integers(0, 50, n)
            h["profile_idx"] = g.permutation(bounds[r][1] - bounds[r][0])[:n]
            h["null_loglik"] = -g.random(n).astype(np.float32) * 1000
            h["alt_loglik"] = h["null_loglik"] + 20
            return h

        for counts in ([3, 5], [0, 4], [0, 0], [40, 2]):
            n = counts[rank]
            cap = 64
            words = torch.zeros((cap, 4), dtype=torch.int32)
            mine = local_hits(rank, n)
            if n:
                words[:n] = torch.from_numpy(mine.view(np.int32).reshape(n, 4))
            got = ddist.gather_hits(words, torch.tensor([n], dtype=torch.int32), b, slab=8)
            want = []
            for r in range(world):
                h = local_hits(r, counts[r]).copy()
                h["profile_idx"] += bounds[r][0]
                want.append(h)
            want = np.concatenate(want)
            want = want[np.lexsort((want["profile_idx"], want["seq_idx"]))]
            assert got.dtype == dcp.HIT_DTYPE and len(got) == sum(counts)
            assert np.array_equal(got, want)
            assert (got["profile_idx"] < len(sizes)).all()
        # the C bookkeeping on its own: counts -> displacements -> global indices -> (seq, profile) order
        a, c = local_hits(0, 6), local_hits(1, 0)
        merged = ddist.merge_hits([6, 0, 0], [bounds[0][0], 17, 99], np.concatenate([a, c]))
        exp = a.copy()
        exp["profile_idx"] += bounds[0][0]
        assert np.array_equal(merged, exp[np.lexsort((exp["profile_idx"], exp["seq_idx"]))])
        assert len(ddist.merge_hits([0, 0], [0, 5], np.zeros(0, dcp.HIT_DTYPE))) == 0
        # overflow is an error, not silent truncation
        try:
            ddist.gather_hits(torch.zeros((4, 4), dtype=torch.int32), torch.tensor([9], dtype=torch.int32), b)
            raise AssertionError("overflow not detected")
        except RuntimeError:
            pass
        # ... on EVERY rank, also when only one rank's buffer overflowed (the peers of an
        # overflowed rank must not return a silently truncated list)
        try:
            ddist.gather_hits(torch.zeros((4, 4), dtype=torch.int32),
                              torch.tensor([9 if rank == 1 else 2], dtype=torch.int32), b)
            raise AssertionError(f"rank {rank}: a peer's overflow went unnoticed")
        except RuntimeError:
            pass
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_gather_plan_bookkeeping(dcp):
    """dcp_dist_gather_plan: what every rank derives from the {held, offset, found} words of the meta
    all-gather -- counts, offsets, 64-bit displacements, and ONE overflow verdict for all ranks."""
    from deciphon_old_amd import dist as ddist

    counts, offs, displ, ovf, total = ddist.gather_plan([[3, 0, 3], [0, 70, 0], [5, 90, 5]])
    assert list(counts) == [3, 0, 5] and list(offs) == [0, 70, 90] and list(displ) == [0, 3, 3, 8]
    assert not ovf and total == 8
    # rank 1 found 9 records but its buffer holds 4: everybody learns it
    counts, _, displ, ovf, total = ddist.gather_plan([[2, 0, 2], [4, 50, 9]])
    assert ovf and list(counts) == [2, 4] and total == 6 and list(displ) == [0, 2, 6]
    # displacements are 64-bit: three ranks with 2^31 records each do not wrap, but the total does not fit the interface
    big = 1 << 31
    with pytest.raises(dcp.DcpError):
        ddist.gather_plan([[big, 0, big], [big, 1, big], [big, 2, big]])
    counts, _, displ, ovf, total = ddist.gather_plan([[big, 0, big], [big - 1, 1, big - 1]])
    assert total == 2 * big - 1 and int(displ[2]) == 2 * big - 1 and not ovf
    # a rank cannot hold more than it found
    with pytest.raises(dcp.DcpError):
        ddist.gather_plan([[5, 0, 4]])


def test_gather_plan_tells_a_failed_scan_from_no_hits(dcp):
    """a rank whose scan failed used to join the gather with {0, off, 0} -- "no hits" -- and every other
    rank, a receiving root included, returned DCP_OK with a list that silently lacked one shard.  It now marks its
    meta words (found = DCP_DIST_FOUND_FAILED, held = 0) and the plan every rank derives says any_failed."""
    from deciphon_old_amd import dist as ddist

    F = ddist.FOUND_FAILED
    assert F == 0xFFFFFFFF
    counts, offs, displ, ovf, total, failed = ddist.gather_plan([[3, 0, 3], [0, 70, 0], [5, 90, 5]], with_failed=True)
    assert not failed and not ovf and total == 8
    counts, offs, displ, ovf, total, failed = ddist.gather_plan([[3, 0, 3], [0, 70, F], [5, 90, 5]], with_failed=True)
    assert failed and not ovf           # a failure is not an overflow (held 0 < found)
    assert list(counts) == [3, 0, 5] and list(displ) == [0, 3, 3, 8] and total == 8  # the exchange still completes
    # failure and overflow on different ranks: both are seen
    *_, ovf, total, failed = ddist.gather_plan([[2, 0, 7], [0, 50, F]], with_failed=True)
    assert failed and ovf and total == 2
    # a failed rank holds nothing
    with pytest.raises(dcp.DcpError):
        ddist.gather_plan([[1, 0, F]], with_failed=True)
    # the torch.distributed transport's use of the plan raises on a failed rank too
    with pytest.raises(dcp.DcpError):
        ddist.gather_plan([[3, 0, 3], [0, 70, F]])


def test_id_file_must_be_fresh(dcp, tmp_path):
    """dcp_dist_init_from_file, rank > 0: a left-over file (old layout, another rank count, or older than
    the staleness bound) is refused -- the call times out instead of joining a communicator of its own
    (ADVICE r2).  Fails before anything touches a device, so it runs on CPU."""
    import ctypes as C
    import struct
    import time
    from deciphon_old_amd import dist as ddist

    lib = dcp.lib
    path = tmp_path / "id"
    path.write_bytes(bytes(128))  # round 2's layout: a bare 128-byte id
    t0 = time.time()
    assert not lib.dcp_dist_init_from_file(str(path).encode(), 1, 2, 0, 0.3)
    assert time.time() - t0 < 5
    path.write_bytes(struct.pack("<II", 0xDC9D1573, 2) + bytes(128))  # round 3's layout (no nonce field)
    assert not lib.dcp_dist_init_from_file(str(path).encode(), 1, 2, 0, 0.3)
    path.write_bytes(struct.pack("<IIQ", 0xDC9D1574, 4, 0) + bytes(128))  # a run with 4 ranks, we are one of 2
    assert not lib.dcp_dist_init_from_file(str(path).encode(), 1, 2, 0, 0.3)
    path.write_bytes(struct.pack("<IIQ", 0xDC9D1574, 2, 0) + bytes(128))  # right layout, written long ago
    old = time.time() - 3600
    os.utime(path, (old, old))
    assert not lib.dcp_dist_init_from_file(str(path).encode(), 1, 2, 0, 0.3)
    assert not lib.dcp_dist_init_from_file(None, 0, 1, 0, 0.1)
    assert not lib.dcp_dist_init_from_file(str(path).encode(), 2, 2, 0, 0.1)  # rank out of range
    # with a run nonce a YOUNG left-over file of a previous run (same path, same rank count) is not
    # taken either -- a peer that arrives before rank 0 has replaced it waits for this run's nonce
    ddist.bind(lib)
    path.write_bytes(struct.pack("<IIQ", 0xDC9D1574, 2, 1111) + bytes(128))  # previous run, written just now
    t0 = time.time()
    assert not lib.dcp_dist_init_from_file_run(str(path).encode(), 2222, 1, 2, 0, 0.3)
    assert time.time() - t0 < 5
    path.write_bytes(struct.pack("<IIQ", 0xDC9D1574, 2, 0) + bytes(128))     # a nonce-less run's file
    assert not lib.dcp_dist_init_from_file_run(str(path).encode(), 2222, 1, 2, 0, 0.3)
    # and a nonce-less peer does not take a nonce-carrying run's file
    path.write_bytes(struct.pack("<IIQ", 0xDC9D1574, 2, 1111) + bytes(128))
    assert not lib.dcp_dist_init_from_file(str(path).encode(), 1, 2, 0, 0.3)


def _gpu_worker(rank, world, port, tmpdir):
    """Two ranks share the one GPU of the test box: each holds its shard of the DB resident, scans
    ALL queries, and the hit records are all-gathered (gloo here, RCCL in bench.py)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_product

    dcp = load_product()
    from deciphon_old_amd import dist as ddist
    import test_gpu_parity as tp
    from oracle_py import Oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        sizes = [int(m) for m in rng.integers(20, 200, 24)]
        params = [tp.pfam_like_params(rng, M) for M in sizes]
        cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
        orc = Oracle(32)
        seqs = tp.rand_seqs(rng, 150, 30, 200)
        for q, p in ((3, 2), (77, 13), (149, 23), (20, 12)):
            seqs[q] = tp.planted_query(rng, orc.new(*params[p], 2, 0.01), sizes[p], flank=10)
        b, e = ddist.shard_range(sizes, world, rank)
        sc = dcp.Scanner(0)
        sc.upload_db([dcp.ProteinProfile.from_params(*params[p], cfg) for p in range(b, e)])
        sc.upload_seqs(seqs)
        sc.scan(True, False, 10.0)
        mine = sc.hits()
        nl, al = sc.scores()
        cap = 256
        words = torch.zeros((cap, 4), dtype=torch.int32)
        if len(mine):
            words[:len(mine)] = torch.from_numpy(mine.view(np.int32).reshape(len(mine), 4).copy())
        allh = ddist.gather_hits(words, torch.tensor([len(mine)], dtype=torch.int32), b)
        sc.close()
        got = {(int(h["seq_idx"]), int(h["profile_idx"])) for h in allh}
        assert {(3, 2), (77, 13), (149, 23), (20, 12)} <= got
        if rank == 0:  # the unsharded scan gives the same hit list, record for record
            full = dcp.Scanner(0)
            full.upload_db([dcp.ProteinProfile.from_params(*prm, cfg) for prm in params])
            full.upload_seqs(seqs)
            full.scan(True, False, 10.0)
            want = full.hits()
            fn, fa = full.scores()
            full.close()
            assert np.array_equal(allh, want)
            assert np.array_equal(fn[:, b:e], nl) and np.array_equal(fa[:, b:e], al)
        open(os.path.join(tmpdir, f"gpu_ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_c_rccl_gather_one_rank(dcp):
    """The C host's RCCL path on real hardware, as far as one GPU allows: librccl.so is loaded on first
    use, ncclCommInitRank(1 rank), the 2-word all-gather, the local-copy leg of the gather-v, the merge.
    (RCCL refuses two ranks on one device; the N>1 exchange itself runs only on the driver's 8-GPU node.)"""
    import ctypes as C
    from deciphon_old_amd import dist as ddist
    import test_gpu_parity as tp
    from oracle_py import Oracle

    rng = np.random.default_rng(11)
    sizes = [int(m) for m in rng.integers(20, 120, 12)]
    params = [tp.pfam_like_params(rng, M) for M in sizes]
    cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
    orc = Oracle(32)
    seqs = tp.rand_seqs(rng, 40, 30, 200)
    for q, p in ((3, 2), (17, 9)):
        seqs[q] = tp.planted_query(rng, orc.new(*params[p], 2, 0.01), sizes[p], flank=10)
    sc = dcp.Scanner(0)
    comm = ddist.CDist.create(ddist.CDist.unique_id(), 0, 1, 0)
    try:
        sc.upload_db([dcp.ProteinProfile.from_params(*prm, cfg) for prm in params])
        sc.upload_seqs(seqs)
        sc.scan(True, False, 10.0)
        want = sc.hits()
        assert len(want) >= 2
        import torch
        cap = 1024
        hits_dev = torch.zeros((cap, 4), dtype=torch.int32, device="cuda")
        count_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
        sc.set_hit_buffer(hits_dev.data_ptr(), cap, count_dev.data_ptr())
        sc.scan(True, False, 10.0, sync=False)
        for root in (-1, 0):
            got, total = comm.gather_hits(hits_dev.data_ptr(), count_dev.data_ptr(), cap, 0, sc.stream, root=root)
            assert total == len(want) and np.array_equal(got, want)
        # a shard that does not start at profile 0: indices come back global
        got, _ = comm.gather_hits(hits_dev.data_ptr(), count_dev.data_ptr(), cap, 1000, sc.stream)
        assert np.array_equal(got["profile_idx"], want["profile_idx"] + 1000)
        # the form hosts call: the gather completes the scan itself (dcp_gpu_sync) and takes the context's buffer
        sc.scan(True, False, 10.0, sync=False)
        got, total = comm.gather_scan_hits(sc, 0)
        assert total == len(want) and np.array_equal(got, want)
        # a buffer too small for the scan's hits: DCP_ENOMEM, not a short list taken for the whole
        sc.set_hit_buffer(hits_dev.data_ptr(), 1, count_dev.data_ptr())
        sc.scan(True, False, 10.0, sync=False)
        with pytest.raises(dcp.DcpError) as ei:
            comm.gather_scan_hits(sc, 0)
        assert ei.value.rc == dcp.RC_ENOMEM
        sc.set_hit_buffer(None, 0, None)
        # without a caller buffer the context's own one is gathered
        sc.scan(True, False, 10.0, sync=False)
        got, _ = comm.gather_scan_hits(sc, 0)
        assert np.array_equal(got, want)
        assert comm.comm_count == 1 and comm.last_gather_ms > 0.0  # what RCCL itself says; the launcher's diagnostics
        # a rank with no valid scan (here: a context that never scanned) still completes both exchanges -- marked
        # DCP_DIST_FOUND_FAILED, holding nothing -- and returns ITS error (what its peers return is
        # test_gather_plan_tells_a_failed_scan_from_no_hits)
        never = dcp.Scanner(0)
        try:
            with pytest.raises(dcp.DcpError) as ei:
                comm.gather_scan_hits(never, 0)
            assert ei.value.rc == dcp.RC_EINVAL and "no scan yet" in str(ei.value)
        finally:
            never.close()
        got, _ = comm.gather_scan_hits(sc, 0)  # the communicator is still usable
        assert np.array_equal(got, want)
    finally:
        comm.close()
        sc.close()


@pytest.mark.gpu
def test_gather_after_redo_list_overflow(dcp):
    """a query-lane scan whose redo lists overflowed loses pairs until dcp_gpu_sync repeats it
    with the row sweep.  dcp_dist_gather_scan_hits runs that completion before it reads the hit buffer,
    so the gathered list is the full one (context of the tests' -DDCP_TEST_HOOKS build: that is where the
    redo capacity can be shrunk to one pair per size class)."""
    from deciphon_old_amd import dist as ddist
    import test_gpu_parity as tp
    from oracle_py import Oracle

    rng = np.random.default_rng(33)
    M = 60
    prm = tp.pfam_like_params(rng, M)
    cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
    oprof = Oracle(32).new(*prm, dcp.ENTRY_DIST_OCCUPANCY, 0.01)
    two = lambda: np.concatenate([tp.planted_query(rng, oprof, M), tp.planted_query(rng, oprof, M)])
    seqs = [two(), two(), two()] + tp.rand_seqs(rng, 5, 100, 300)
    hooks = dcp.load_testhooks()
    sc = dcp.Scanner(0, lib=hooks)
    comm = ddist.CDist.create(ddist.CDist.unique_id(), 0, 1, 0, lib_=hooks)
    try:
        sc.upload_db([dcp.ProteinProfile.from_params(*prm, cfg)])
        sc.upload_seqs(seqs)
        sc.scan(True, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
        want = sc.hits()
        assert [(int(h["seq_idx"]), int(h["profile_idx"])) for h in want][:3] == [(0, 0), (1, 0), (2, 0)]
        sc.test_set_redo_cap(1)
        sc.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE, sync=False)  # 3 feedback pairs, room for 1
        got, total = comm.gather_scan_hits(sc, 0)
        assert total == len(want) and np.array_equal(got, want)
        sc.test_set_redo_cap(0)
    finally:
        comm.close()
        sc.close()


@pytest.mark.gpu
def test_sharded_scan_two_ranks_one_gpu(tmp_path):
    world = 2
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"gpu_ok{r}").exists() for r in range(world))
