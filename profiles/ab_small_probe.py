"""Alternate builds of libdcp_hip.so on small batches in ONE process (each build its own context on the same DB):
    python3 profiles/ab_small_probe.py c3|c5 deciphon-old_amd/libdcp_hip.A.so deciphon-old_amd/libdcp_hip.B.so [nq=1,2,4,...]
Automatic kernel choice, 1 .. 64 queries of 1 kbp against the 20 000 profiles; median of 5 alternating repetitions."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
dcp = bench.load_product()
WL = sys.argv[1]
libs = [a for a in sys.argv[2:] if not a.startswith("nq=")]
NQS = [int(x) for a in sys.argv[2:] if a.startswith("nq=") for x in a[3:].split(",")] or [1, 2, 4, 8, 16, 32, 64]
sizes = bench.core_sizes_for(WL, 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
NQMAX = max(NQS)
q = bench.make_queries(0, NQMAX, 1000)
off = (np.arange(NQMAX + 1, dtype=np.uint64) * 1000).astype(np.uint32)
scs = []
for path in libs:
    sc = dcp.Scanner(0, lib=dcp._load(os.path.abspath(path)))
    sc.upload_db(profs)
    sc.upload_seqs_flat(q.reshape(-1), off)
    scs.append(sc)
del profs
print(f"workload {WL}: sum M = {int(sizes.sum())}; ms per scan, median of 5 (alternating)", flush=True)
for nq in NQS:
    t = {p: [] for p in libs}
    for rep in range(6):
        for p, sc in zip(libs, scs):
            sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0, nq))
            if rep:
                t[p].append(sc.last_scan_ms)
    print(f"nq={nq:3d}  " + "  ".join(f"{os.path.basename(p)} {np.median(t[p]):8.2f}" for p in libs), flush=True)
