"""Where the automatic kernel choice should leave the row sweep: row sweep vs the query-lane kernels at 128..256
   queries on the C3 DB.  python3 profiles/switch_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
dcp = bench.load_product()
sizes = bench.core_sizes_for("c3", 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
sc = dcp.Scanner(0); sc.upload_db(profs); del profs
q = bench.make_queries(0, 256, 1000)
sc.upload_seqs_flat(q.reshape(-1), (np.arange(257, dtype=np.uint64) * 1000).astype(np.uint32))
for nq in (128, 144, 160, 176, 191, 192, 224, 256):
    row = []
    for name, k in (("rowsweep", dcp.KERNEL_ROWSWEEP), ("qlane", dcp.KERNEL_QLANE), ("qlane2", dcp.KERNEL_QLANE2), ("auto", dcp.KERNEL_AUTO)):
        for rep in range(2):
            sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0, nq), kernel=k)
        row.append(f"{name} {sc.last_scan_ms:7.1f}" + (f" (= kernel {sc.last_scan_kernel})" if name == "auto" else ""))
    print(f"nq={nq:4d}  " + "   ".join(row), flush=True)
