"""How long does the device traceback of one C3 step's hits take (N1: hits -> alt paths -> product rows)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from concurrent.futures import ThreadPoolExecutor
dcp = bench.load_product()
sizes = bench.core_sizes_for("c3", 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
sc = dcp.Scanner(0)
sc.upload_db(profs)
q = bench.make_queries(0, 1000, 1000)
sc.upload_seqs_flat(q.reshape(-1), (np.arange(1001, dtype=np.uint64) * 1000).astype(np.uint32))
t = time.perf_counter(); sc.scan(True, False, 10.0, keep_scores=False, sync=True); t_scan = time.perf_counter() - t
hits = sc.hits()
print(f"scan {t_scan*1e3:.1f} ms, {len(hits)} hits", flush=True)
for rep in range(2):
    t = time.perf_counter()
    paths, alts = sc.trace_paths(hits, True, False)
    dt = time.perf_counter() - t
    nsteps = sum(len(p) for p in paths)
    print(f"trace_paths: {dt*1e3:.1f} ms for {len(hits)} hits ({dt/len(hits)*1e6:.0f} us/hit), {nsteps} steps; "
          f"alt bit-equal to the scan's: {bool(np.array_equal(np.asarray(alts, np.float32), hits['alt_loglik']))}", flush=True)
t = time.perf_counter()
rows = [profs[int(h['profile_idx'])].prod_row(q[int(h['seq_idx'])], p, 1, int(h['seq_idx']), float(h['alt_loglik']), float(h['null_loglik']))
        for h, p in zip(hits[:500], paths[:500])]
print(f"500 product rows formatted in {(time.perf_counter()-t)*1e3:.1f} ms (host)", flush=True)
