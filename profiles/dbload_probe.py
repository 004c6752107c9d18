import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from concurrent.futures import ThreadPoolExecutor
dcp = bench.load_product()
sizes = bench.core_sizes_for("c3", 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
t=time.perf_counter()
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
print("sample %.2f s" % (time.perf_counter()-t))
sc = dcp.Scanner(0)
t=time.perf_counter(); sc.upload_db(profs); print("upload_db (compact H2D + query-lane layout) %.3f s" % (time.perf_counter()-t))
q = bench.make_queries(0, 1000, 1000)
t=time.perf_counter(); sc.upload_seqs_flat(q.reshape(-1), (np.arange(1001, dtype=np.uint64) * 1000).astype(np.uint32)); print("upload_seqs %.3f s" % (time.perf_counter()-t))
for i in range(3):
    t=time.perf_counter(); sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0,4), kernel=dcp.KERNEL_ROWSWEEP); print("rowsweep scan of 4 queries #%d: %.3f s" % (i, time.perf_counter()-t))
for i in range(2):
    t=time.perf_counter(); sc.scan(True, False, 10.0, keep_scores=False, sync=True); print("auto scan of 1000 queries #%d: %.3f s" % (i, time.perf_counter()-t))
