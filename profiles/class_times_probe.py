"""Per size class: when its launch completed (ms since the scan's start) for small batches on the C3 / C5 DB.
    python3 profiles/class_times_probe.py c3|c5"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
dcp = bench.load_product()
WL = sys.argv[1] if len(sys.argv) > 1 else "c3"
sizes = bench.core_sizes_for(WL, 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
sc = dcp.Scanner(0)
sc.upload_db(profs)
del profs
q = bench.make_queries(0, 64, 1000)
sc.upload_seqs_flat(q.reshape(-1), (np.arange(65, dtype=np.uint64) * 1000).astype(np.uint32))
for nq in (1, 2, 8, 32):
    for rep in range(3):
        sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0, nq))
    print(f"nq={nq} scan {sc.last_scan_ms:.2f} ms:", "  ".join(f"R{li['R']}W{li['W']}[{li['nprofiles']}] {li['ms']:.2f}" for li in sc.launch_infos()), flush=True)
