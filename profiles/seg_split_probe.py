"""Per-launch times of one row-sweep scan of the C3 step (1 000 queries): a segmented class shows as two launches
   (the sweep, then the exact kernel on its redo list).  python3 profiles/seg_split_probe.py"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, bench
dcp = bench.load_product()
sizes = bench.core_sizes_for("c3", 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
sc = dcp.Scanner(0); sc.upload_db(profs); del profs
q = bench.make_queries(0, 1000, 1000)
off = (np.arange(1001, dtype=np.uint64) * 1000).astype(np.uint32)
sc.upload_seqs_flat(q.reshape(-1), off)
for rep in range(2):
    sc.scan(True, False, 10.0, keep_scores=False, sync=True, kernel=dcp.KERNEL_ROWSWEEP)
print([(f"R{li['R']}W{li['W']}", round(li['ms'],1), li['cells']>0) for li in sc.launch_infos()])
