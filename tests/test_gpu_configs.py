"""BASELINE.json's configs on the device, each checked against the CPU oracle.

configs[1] "C2"  1k-profile synthetic DB x 1k 300-nt queries            -> test_c2_*
configs[3] "C4"  the C3 DB sharded over 8 GPUs, 8 000-query steps        -> test_c4_* (one rank's share)
configs[4] "C5"  M 50-2000 x queries 100 nt-10 kbp, 8 GPUs               -> test_c5_* (one rank's share)
(configs[0] = the reference's golden test: tests/test_golden_fixtures.py, tests/test_c_host.py;
 configs[2] "C3" = test_gpu_parity.py::test_full_size_c3_step_both_kernels_agree.)

Every test runs ALL device kernels (row sweep, query lane + redo, two-stage query lane + redo) through the C-ABI, requires them to
agree bit for bit on every pair and on the hit list, and compares a sample of pairs -- always including
the extreme ones -- with the oracle's independent model build and Viterbi (<= 5e-5 relative, the
reference's float32 bar, test/hope_support.h:26).  Generators: bench.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REL = 5e-5


def scan_both(dcp, sc):
    out = {}
    for name, k in (("qlane", dcp.KERNEL_QLANE), ("qlane2", dcp.KERNEL_QLANE2), ("rowsweep", dcp.KERNEL_ROWSWEEP)):
        sc.scan(True, False, 10.0, kernel=k)
        n, a = sc.scores()
        out[name] = (n.view(np.uint32).copy(), a.view(np.uint32).copy(), sc.hits())
    for other in ("qlane2", "rowsweep"):
        assert np.array_equal(out["qlane"][0], out[other][0]), other
        assert np.array_equal(out["qlane"][1], out[other][1]), other
        assert np.array_equal(out["qlane"][2], out[other][2]), other
    nul, alt = out["qlane"][0].view(np.float32), out["qlane"][1].view(np.float32)
    assert np.isfinite(alt).all() and np.isfinite(nul).all()
    return nul, alt, out["qlane"][2]


def check_pairs_against_oracle(oracle32, sizes, first_profile, queries, nul, alt, pairs):
    worst = 0.0
    for q, p in pairs:
        gp = first_profile + p
        op = oracle32.sample(0xDEC1F0 + gp, int(sizes[gp]))
        seq = bytes(queries[q])
        assert op.setup(len(seq), True, False) == 0
        rc, on, oa = op.viterbi_fast(seq)
        assert rc == 0
        worst = max(worst, abs(nul[q, p] - on) / abs(on), abs(alt[q, p] - oa) / abs(oa))
    assert worst <= REL, worst
    return worst


def check_hits(hits, nul, alt, thr=10.0):
    """the hit list is exactly the LRT filter of scan_thread.c:121-123 over the dense scores"""
    lrt = np.float32(-2) * (nul - alt)
    want = np.argwhere(np.isfinite(lrt) & ~(lrt < np.float32(thr)))
    got = np.stack([hits["seq_idx"], hits["profile_idx"]], axis=1).astype(np.int64)
    assert np.array_equal(got, want)
    assert np.array_equal(hits["null_loglik"], nul[want[:, 0], want[:, 1]])
    assert np.array_equal(hits["alt_loglik"], alt[want[:, 0], want[:, 1]])


def test_c2_config_full(dcp, oracle32, bench_mod):
    """BASELINE.json configs[1] at full size: 1 000 profiles (M = 100 + 37p mod 201) x 1 000 queries of
    300 nt = 1e6 pairs, 6e10 cells."""
    from concurrent.futures import ThreadPoolExecutor

    sizes = bench_mod.core_sizes_for("c2", 1000)
    assert sizes.min() == 100 and sizes.max() == 300 and len(sizes) == 1000
    cfg = dcp.ProteinCfg(2, 0.01)
    with ThreadPoolExecutor(16) as ex:
        profiles = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg), range(1000)))
    queries = bench_mod.make_queries(0, 1000, 300)
    sc = dcp.Scanner(0)
    try:
        sc.upload_db(profiles)
        sc.upload_seqs_flat(queries.reshape(-1), (np.arange(1001, dtype=np.uint64) * 300).astype(np.uint32))
        nul, alt, hits = scan_both(dcp, sc)
        check_hits(hits, nul, alt)
        rng = np.random.default_rng(2)
        pairs = [(int(rng.integers(0, 1000)), int(rng.integers(0, 1000))) for _ in range(60)]
        pairs += [(0, int(np.argmax(sizes))), (999, int(np.argmin(sizes))), (999, 999)]
        pairs += [(int(h["seq_idx"]), int(h["profile_idx"])) for h in hits[:8]]  # and some that passed the filter
        check_pairs_against_oracle(oracle32, sizes, 0, queries, nul, alt, pairs)
        # the library's own kernel choice on this small DB scores the same
        sc.scan(True, False, 10.0)
        n0, a0 = sc.scores()
        assert np.array_equal(n0.view(np.uint32), nul.view(np.uint32)) and np.array_equal(a0.view(np.uint32), alt.view(np.uint32))
    finally:
        sc.close()


def test_c5_config_one_gpu_share(dcp, oracle32, bench_mod):
    """BASELINE.json configs[4]: the 20 000-profile draw (M log-uniform 50..2000, seed 50) sharded over 8
    GPUs by cells -- this is rank 3's shard -- x the first 1 000 queries of the 50 000 (100 nt..10 kbp
    log-uniform).  Oracle sample includes the longest query x the largest profile of the shard."""
    from concurrent.futures import ThreadPoolExecutor
    from deciphon_old_amd import dist as ddist

    sizes = bench_mod.core_sizes_for("c5", 20000)
    assert sizes.min() >= 50 and sizes.max() <= 2000
    b, e = ddist.shard_range(sizes, 8, 3)
    cfg = dcp.ProteinCfg(2, 0.01)
    with ThreadPoolExecutor(16) as ex:
        profiles = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg), range(b, e)))
    queries = bench_mod.make_queries(0, 1000, 0)
    lens = np.array([len(q) for q in queries])
    assert lens.min() < 120 and lens.max() > 9000
    sc = dcp.Scanner(0)
    try:
        sc.upload_db(profiles)
        del profiles
        sc.upload_seqs(queries)
        nul, alt, hits = scan_both(dcp, sc)
        check_hits(hits, nul, alt)
        shard = sizes[b:e]
        qmax, qmin, pmax, pmin = int(np.argmax(lens)), int(np.argmin(lens)), int(np.argmax(shard)), int(np.argmin(shard))
        rng = np.random.default_rng(5)
        pairs = [(qmax, pmax), (qmax, pmin), (qmin, pmax), (qmin, pmin)]
        pairs += [(int(rng.integers(0, 1000)), int(rng.integers(0, e - b))) for _ in range(16)]
        pairs += [(int(h["seq_idx"]), int(h["profile_idx"])) for h in hits[:6]]
        check_pairs_against_oracle(oracle32, sizes, b, queries, nul, alt, pairs)
    finally:
        sc.close()


def test_c4_config_one_gpu_share_vs_unsharded(dcp, oracle32, bench_mod, c3_profiles):
    """BASELINE.json configs[3]: the C3 DB sharded over 8 GPUs, weak scaling = 8 000 queries per step.
    One rank's share (rank 5's contiguous shard x all 8 000 queries of a step) must produce exactly the
    records the unsharded scan of the same queries produces for those profiles (what the hit gather then
    concatenates), and the shards must tile the DB."""
    from deciphon_old_amd import dist as ddist

    sizes, profiles = c3_profiles
    bounds = [ddist.shard_range(sizes, 8, r) for r in range(8)]
    assert bounds[0][0] == 0 and bounds[-1][1] == len(sizes)
    assert all(bounds[r][1] == bounds[r + 1][0] for r in range(7))
    cells = np.array([int(sizes[b:e].sum()) for b, e in bounds], np.float64)
    assert cells.max() / cells.mean() < 1.01  # balanced by sum of M, not by count
    b, e = bounds[5]
    queries = bench_mod.make_queries(0, 8000, 1000)
    off = (np.arange(8001, dtype=np.uint64) * 1000).astype(np.uint32)
    full = dcp.Scanner(0)
    part = dcp.Scanner(0)
    try:
        full.upload_db(profiles)
        full.upload_seqs_flat(queries.reshape(-1), off)
        full.scan(True, False, 10.0, keep_scores=False)
        hf = full.hits()
        part.upload_db(profiles[b:e])
        part.upload_seqs_flat(queries.reshape(-1), off)
        part.scan(True, False, 10.0, keep_scores=True)
        hp = part.hits()
        want = hf[(hf["profile_idx"] >= b) & (hf["profile_idx"] < e)].copy()
        want["profile_idx"] -= b
        assert len(hp) > 100
        assert np.array_equal(hp, want)
        nul, alt = part.scores()
        check_hits(hp, nul, alt)
        rng = np.random.default_rng(4)
        pairs = [(int(rng.integers(0, 8000)), int(rng.integers(0, e - b))) for _ in range(6)]
        pairs += [(int(h["seq_idx"]), int(h["profile_idx"])) for h in hp[:4]]
        check_pairs_against_oracle(oracle32, sizes, b, queries, nul, alt, pairs)
    finally:
        full.close()
        part.close()
