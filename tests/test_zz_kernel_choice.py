"""What `dcp_scan_params.kernel = 0` picks: TUNING assertions, kept apart from every parity test.

A refit of the library's cost model may move a switch point; it must not turn a bit-for-bit comparison red, and under
`pytest -x` it must not hide later files -- so these checks live in the file pytest collects last, and their points
sit far from the switches (on the 20 000-profile DB the row sweep hands over to the query-lane kernels at about
150 queries, the two-stage kernel takes over near 190: 8 and 1 000 queries are a factor of five or more away).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_choice_on_the_c3_database(dcp, c3_profiles, bench_mod):
    sizes, profiles = c3_profiles
    queries = bench_mod.make_queries(0, 1000, 1000)
    sc = dcp.Scanner(0)
    try:
        sc.upload_db(profiles)
        sc.upload_seqs_flat(queries.reshape(-1), (np.arange(1001, dtype=np.uint64) * 1000).astype(np.uint32))
        picked = {}
        for nq in (1, 8, 1000):
            sc.scan(True, False, 10.0, keep_scores=False, q_range=(0, nq))
            picked[nq] = (sc.last_scan_kernel, sc.launch_infos()[0]["W"])
        assert picked[1][0] == dcp.KERNEL_ROWSWEEP and picked[1][1] >= 1, picked
        assert picked[8][0] == dcp.KERNEL_ROWSWEEP and picked[8][1] >= 1, picked
        assert picked[1000] == (dcp.KERNEL_QLANE2, 0), picked
    finally:
        sc.close()


def test_choice_on_a_db_of_four_profiles(dcp):
    """A DB of a few profiles cannot fill the query-lane kernels' persistent grid: the row sweep at any batch size."""
    cfg = dcp.ProteinCfg(2, 0.01)
    profiles = [dcp.ProteinProfile.sample(900 + i, m, cfg) for i, m in enumerate((3, 70, 130, 300))]
    rng = np.random.default_rng(48)
    seqs = [rng.integers(0, 4, int(rng.integers(20, 120)), dtype=np.uint8) for _ in range(60)]
    sc = dcp.Scanner(0)
    try:
        sc.upload_db(profiles)
        sc.upload_seqs(seqs)
        sc.scan(True, False, 10.0)
        assert sc.last_scan_kernel == dcp.KERNEL_ROWSWEEP
        assert all(li["W"] >= 1 for li in sc.launch_infos())
    finally:
        sc.close()


def test_choice_for_one_very_long_sequence(dcp):
    """A 300 000-nt query among short ones: the row sweep (a query-lane block would keep one lane busy for minutes)."""
    cfg = dcp.ProteinCfg(2, 0.01)
    profiles = [dcp.ProteinProfile.sample(301, 40, cfg), dcp.ProteinProfile.sample(302, 120, cfg)]
    rng = np.random.default_rng(1 << 20)
    seqs = [rng.integers(0, 4, n, dtype=np.uint8) for n in (300000, 33, 2000)]
    sc = dcp.Scanner(0)
    try:
        sc.upload_db(profiles)
        sc.upload_seqs(seqs)
        sc.scan(True, False, 10.0, keep_scores=False)
        assert sc.last_scan_kernel == dcp.KERNEL_ROWSWEEP
    finally:
        sc.close()
