"""Small-batch scan latency on the C3 DB -- or, `python3 profiles/latency_probe.py c5`, on the C5 DB (20 000 profiles of
50 .. 2 000 nodes: 37 % of them in the multi-wavefront classes) -- with 1 kbp queries (the reference scans one sequence
at a time: scan.c:227-258)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
dcp = bench.load_product()
WL = sys.argv[1] if len(sys.argv) > 1 else "c3"
sizes = bench.core_sizes_for(WL, 20000)
print(f"workload {WL}: 20000 profiles, sum M = {int(sizes.sum())}", flush=True)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
sc = dcp.Scanner(0)
sc.upload_db(profs)
del profs
q = bench.make_queries(0, 1024, 1000)
off = (np.arange(1025, dtype=np.uint64) * 1000).astype(np.uint32)
sc.upload_seqs_flat(q.reshape(-1), off)
cellsM = int(sizes.sum())
for kname, kern in (("auto", dcp.KERNEL_AUTO), ("rowsweep", dcp.KERNEL_ROWSWEEP), ("qlane", dcp.KERNEL_QLANE),
                    ("qlane2", dcp.KERNEL_QLANE2)):
    for nq in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 256, 512, 1024):
        if kname == "rowsweep" and nq > 64 or kname.startswith("qlane") and nq < 16:
            continue
        if WL != "c3" and (nq > 256 or nq in (2, 4, 48, 96) or kname == "qlane"):
            continue
        for rep in range(2):
            t = time.perf_counter()
            sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0, nq), kernel=kern)
            dt = time.perf_counter() - t
        print(f"{kname:8s} nq={nq:4d} wall {dt*1e3:8.2f} ms  kernel {sc.last_scan_ms:8.2f} ms  launches {sc.last_scan_launches}  "
              f"{cellsM*nq*1000/dt/1e9:7.1f} Gcell/s", flush=True)
        if nq == 1 and kname == "auto":
            print("   per size class (ms since scan start):",
                  {f"R{li['R']}W{li['W']}": (round(li["ms"], 2), li["nprofiles"]) for li in sc.launch_infos()}, flush=True)

# one sequence at a time, as scan.c:227-258 drives thread_run: upload + scan + fetch hits per sequence
for i in range(6):
    t0 = time.perf_counter()
    sc.upload_seqs([q[i]])
    t1 = time.perf_counter()
    sc.scan(True, False, 10.0, keep_scores=False, sync=True)
    t2 = time.perf_counter()
    h = sc.hits()
    t3 = time.perf_counter()
    print(f"per-sequence loop {i}: upload {1e3*(t1-t0):6.2f} ms  scan {1e3*(t2-t1):6.2f} ms (kernel {sc.last_scan_ms:6.2f})  hits {1e3*(t3-t2):5.2f} ms  n={len(h)}", flush=True)
