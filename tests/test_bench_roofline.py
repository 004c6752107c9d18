"""bench.py's JSON assembly on the CPU with a stubbed scanner.

The `roofline` object must describe the kernel of the TIMED leg.  BENCH_r03.json named
`viterbi_qlane_kernel` and replayed that kernel's HBM traffic (10.9 TB) although the timed steps ran
`viterbi_qlane2_kernel` (5.74 TB): the label was read from the context after the small-batch leg had run
row-sweep scans on it.  Here a later leg switches the stub's kernel and the label must not move.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("dcp_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


DCP = types.SimpleNamespace(KERNEL_AUTO=0, KERNEL_ROWSWEEP=1, KERNEL_QLANE=2, KERNEL_QLANE2=3)


class StubScanner:
    """The attributes bench.py reads of a Scanner.  Big batches 'run' the two-stage query-lane kernel plus two
    redo launches of the row sweep, batches of <= 64 queries the row sweep (what the cost model does)."""

    def __init__(self, sum_m, qlen):
        self.sum_m, self.qlen = sum_m, qlen
        self.nq = 0
        self.last_scan_kernel = 0
        self.last_scan_ms = 0.0
        self.last_scan_redo_pairs = 0
        self._infos = []
        self.scans = 0

    def upload_seqs_flat(self, cat, off):
        self.nq = len(off) - 1

    def scan(self, multi_hits=True, hmmer3_compat=False, lrt=10.0, keep_scores=False, sync=True, q_range=None,
             kernel=0):
        nq = (q_range[1] - q_range[0]) if q_range else self.nq
        cells = self.sum_m * nq * self.qlen
        self.scans += 1
        if nq > 64:
            self.last_scan_kernel = DCP.KERNEL_QLANE2
            self._infos = [dict(R=8, W=0, nprofiles=20000, ms=2400.0, cells=cells, algorithmic_bytes=20 * cells),
                           dict(R=3, W=1, nprofiles=9000, ms=30.0, cells=cells // 100, algorithmic_bytes=cells // 5),
                           dict(R=6, W=1, nprofiles=3000, ms=20.0, cells=cells // 100, algorithmic_bytes=cells // 5)]
            self.last_scan_redo_pairs = 2000
        else:
            self.last_scan_kernel = DCP.KERNEL_ROWSWEEP
            self._infos = [dict(R=3, W=1, nprofiles=9000, ms=6.0, cells=cells // 2, algorithmic_bytes=10 * cells),
                           dict(R=6, W=1, nprofiles=3000, ms=5.0, cells=cells // 2, algorithmic_bytes=10 * cells)]
            self.last_scan_redo_pairs = 0
        self.last_scan_ms = sum(li["ms"] for li in self._infos)

    def sync(self):
        pass

    def launch_infos(self):
        return list(self._infos)


def run_legs(bench, later_leg=True):
    sizes = bench.core_sizes_for("c3", 20000)
    qlen, qstep, steps, warmup = 1000, 1000, 3, 1
    sc = StubScanner(int(sizes.sum()), qlen)

    def step(i):
        sc.scan(True, False, 10.0, keep_scores=False, sync=False, q_range=(i * qstep, (i + 1) * qstep))
        sc.sync()

    timed = bench.timed_leg(sc, step, warmup, warmup + steps, lambda: None)
    if later_leg:  # the leg that moved the label in round 3
        src = np.zeros((64, qlen), np.uint8)
        small = bench.small_batches_leg(sc, DCP, src, qlen, float(sizes.sum()), lambda: None)
        assert set(small) == {"1", "8", "64", "what"}
        assert sc.last_scan_kernel == DCP.KERNEL_ROWSWEEP  # the context HAS moved on
    roof = bench.roofline_block(timed, DCP, "c3", steps, sizes, 0, len(sizes), qstep, 1, qlen)
    return timed, roof, sizes


def test_label_and_traffic_are_the_timed_legs():
    bench = load_bench()
    timed, roof, sizes = run_legs(bench)
    assert timed.kernels == {DCP.KERNEL_QLANE2: 3}
    assert roof["kernel"].startswith("viterbi_qlane2_kernel")
    assert roof["timed_step_kernels"] == {"3": 3}
    pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_hbm.json")))
    want = [v["hbm_bytes_per_launch"] for k, v in pmc["kernels"].items() if "viterbi_qlane2_kernel" in k]
    assert len(want) == 1 and roof["traffic"] == want[0]
    assert 5.0e12 < roof["traffic"] < 6.5e12  # the two-stage kernel's planes, not the single-stage kernel's 10.9 TB
    # achieved HBM fraction = counter bytes / this run's launch time / 8 TB/s, stated explicitly
    assert roof["hbm"]["achieved_frac"] == round(roof["traffic"] / 2.4 / 8e12, 4)
    assert roof["hbm"]["achieved_frac"] == roof["hbm"]["measured_frac"]
    # analytic plane bytes follow the two-stage layout (odd -> even boundaries only)
    ntiles = (sizes.astype(np.int64) + 7) // 8
    assert roof["hbm"]["analytic_scratch_plane_bytes_per_launch"] == 24 * int(((ntiles - 1) // 2).sum()) * 1024 * 1000
    # frac uses the dominant launch's own cells and HIP-event time
    cells = int(sizes.sum()) * 1000 * 1000
    assert roof["cells_per_launch"] == cells and roof["avg_launch_ms"] == 2400.0
    assert abs(roof["frac"] - cells * 28 / 2.4 / 78.6e12) < 1e-4
    assert roof["redo_pairs_per_step"] == 2000.0


def test_later_leg_does_not_change_the_block():
    bench = load_bench()
    _, with_leg, _ = run_legs(bench, later_leg=True)
    _, without, _ = run_legs(bench, later_leg=False)
    assert with_leg == without


def test_rowsweep_run_is_labelled_rowsweep():
    """--kernel rowsweep: the dominant launch is a size class of the row sweep, no replayed query-lane traffic."""
    bench = load_bench()
    sizes = bench.core_sizes_for("c3", 20000)
    sc = StubScanner(int(sizes.sum()), 1000)

    def step(i):
        sc.scan(q_range=(0, 64))

    timed = bench.timed_leg(sc, step, 0, 2, lambda: None)
    roof = bench.roofline_block(timed, DCP, "c3", 2, sizes, 0, len(sizes), 64, 1, 1000)
    assert roof["kernel"] == "viterbi_rowsweep_kernel<R=3,W=1>"
    assert roof["traffic"] is None and roof["hbm"]["achieved_frac"] is None
    assert roof["l2_gather"] is not None and roof["lds"] is None


def test_a_rank_of_eight_counts_all_the_step_s_queries():
    """Weak scaling: with N ranks the step has N x 1 000 queries and EVERY rank scans all of them against its shard.  The
    analytic byte figures of the block are per launch of that rank (until round 4 they divided the step's queries by N)."""
    bench = load_bench()
    sizes = bench.core_sizes_for("c3", 20000)
    world, qstep, qlen, steps = 8, 8000, 1000, 2
    b, e = 2500, 5000
    sc = StubScanner(int(sizes[b:e].sum()), qlen)

    def step(i):
        sc.scan(q_range=(i * qstep, (i + 1) * qstep))

    timed = bench.timed_leg(sc, step, 0, steps, lambda: None)
    roof = bench.roofline_block(timed, DCP, "c3", steps, sizes, b, e, qstep, world, qlen)
    sum_m = int(sizes[b:e].sum())
    assert roof["cells_per_launch"] == sum_m * qstep * qlen
    assert roof["hbm"]["compulsory_bytes_per_launch"] == 548 * sum_m * qstep + (qlen + 8) * (e - b) * qstep
    ntiles = (sizes[b:e].astype(np.int64) + 7) // 8
    assert roof["hbm"]["analytic_scratch_plane_bytes_per_launch"] == 24 * int(((ntiles - 1) // 2).sum()) * qstep * qlen
    assert roof["hbm"]["analytic_tile_image_bytes_per_launch"] == int(ntiles.sum()) * 8 * 1364 * 4 * ((qstep + 255) // 256)
    assert roof["traffic"] is None  # the committed counters are those of the one-GPU command
