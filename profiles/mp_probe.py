"""K profiles per wavefront (viterbi_mp_kernel) against the one-profile kernels for the two smallest size classes, on the
C3 DB with 1 kbp queries, through the tests' -DDCP_TEST_HOOKS build (bits 26..27 of the row-sweep variant word):
whole-scan ms of a forced row-sweep scan and the launch records of the R = 1 and R = 2 classes, per batch size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
dcp = bench.load_product()
sizes = bench.core_sizes_for("c3", 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
sc = dcp.Scanner(0, lib=dcp.load_testhooks())
sc.upload_db(profs)
del profs
q = bench.make_queries(0, 1000, 1000)
sc.upload_seqs_flat(q.reshape(-1), (np.arange(1001, dtype=np.uint64) * 1000).astype(np.uint32))
for nq in (1, 4, 8, 16, 32, 64, 128, 192, 256, 384, 512, 1000):
    row = []
    for name, mode in (("one-profile", 1), ("K-per-wavefront", 2), ("auto", 0)):
        sc.test_set_rowsweep_variant(-1, mode << 26)
        best = None
        for rep in range(2):
            t = time.perf_counter()
            sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0, nq), kernel=dcp.KERNEL_ROWSWEEP)
            dt = time.perf_counter() - t
            infos = {(li["R"], li["W"]): li["ms"] for li in sc.launch_infos()}
            cur = (sc.last_scan_ms, infos.get((1, 1)), infos.get((2, 1)))
            best = cur if best is None or cur[0] < best[0] else best
        row.append(f"{name}: scan {best[0]:8.2f} ms  R1W1 {best[1]:7.2f}  R2W1 {best[2]:7.2f}")
    print(f"nq={nq:4d}  " + "  |  ".join(row), flush=True)
sc.test_set_rowsweep_variant(-1, 0)
