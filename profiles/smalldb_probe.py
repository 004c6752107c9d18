"""Kernel choice for small DBs: query-lane vs row sweep when there are fewer (profile, query-block) tasks
than the persistent grid has blocks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
dcp = bench.load_product()
sizes = bench.core_sizes_for("c3", 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
q = bench.make_queries(0, 512, 1000)
off = (np.arange(513, dtype=np.uint64) * 1000).astype(np.uint32)
for nprof in (25, 50, 100, 200, 400, 800):
    profs = [dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg) for p in range(nprof)]
    sc = dcp.Scanner(0)
    sc.upload_db(profs)
    sc.upload_seqs_flat(q.reshape(-1), off)
    for nq in (64, 256, 512):
        res = {}
        for kname, kern in (("rowsweep", dcp.KERNEL_ROWSWEEP), ("qlane", dcp.KERNEL_QLANE), ("qlane2", dcp.KERNEL_QLANE2),
                            ("auto", dcp.KERNEL_AUTO)):
            for rep in range(2):
                sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0, nq), kernel=kern)
            res[kname] = sc.last_scan_ms
        chosen = {dcp.KERNEL_ROWSWEEP: "rowsweep", dcp.KERNEL_QLANE: "qlane", dcp.KERNEL_QLANE2: "qlane2"}[sc.last_scan_kernel]
        best = min(("rowsweep", "qlane", "qlane2"), key=lambda k: res[k])
        tasks = nprof * ((nq + 255) // 256)
        print(f"nprof={nprof:4d} nq={nq:4d} tasks={tasks:5d}  rowsweep {res['rowsweep']:8.2f} ms  qlane {res['qlane']:8.2f} ms  qlane2 {res['qlane2']:8.2f} ms  auto {res['auto']:8.2f} ms = {chosen}"
              + ("" if res[chosen] <= 1.05 * res[best] else f"   <-- {best} is faster"), flush=True)
    sc.close()
