"""Row-sweep latency points on the C3 DB (20 000 profiles): ms per scan for small batches, kernel = row sweep.
   python3 profiles/rs_probe.py [nq ...]      (default 1 2 4 8 16 32 64 128)
   RS_VARIANTS="20:4 84:8 ..." runs each forced (rows staged : wavefronts per block) variant through the
   test-hooks build (libdcp_hip_testhooks.so); without it the shipped library's own choice is measured."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
dcp = bench.load_product()
nqs = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128]
sizes = bench.core_sizes_for("c3", 20000)
cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
from concurrent.futures import ThreadPoolExecutor
with ThreadPoolExecutor(16) as ex:
    profs = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"), range(20000)))
variants = [tuple(int(x) for x in v.split(":")) for v in os.environ.get("RS_VARIANTS", "").split()]
sc = dcp.Scanner(0, lib=dcp.load_testhooks()) if variants else dcp.Scanner(0)
sc.upload_db(profs)
del profs
nmax = max(nqs)
q = bench.make_queries(0, nmax, 1000)
off = (np.arange(nmax + 1, dtype=np.uint64) * 1000).astype(np.uint32)
sc.upload_seqs_flat(q.reshape(-1), off)
cellsM = int(sizes.sum())
for var in variants or [None]:
    if var: sc.test_set_rowsweep_variant(var[0], var[1] | ((var[2] if len(var) > 2 else 0) << 8) | ((var[3] if len(var) > 3 else 0) << 16) | ((var[4] if len(var) > 4 else 0) << 20))  # stage : waves [: KiB of LDS padding [: two-row prefetch [: only class R]]]
    out = []
    for nq in nqs:
        for rep in range(3):
            sc.scan(True, False, 10.0, keep_scores=False, sync=True, q_range=(0, nq), kernel=dcp.KERNEL_ROWSWEEP)
        out.append(f"{nq}: {sc.last_scan_ms:.1f} ms ({cellsM*nq*1000/(sc.last_scan_ms*1e-3)/1e9:.0f})")
        if os.environ.get("RS_CLASSES") and nq * 20000 >= (1 << 21):  # smaller scans fork the class launches
            per = {}
            for li in sc.launch_infos():  # ms of each launch (a segmented class: its sweep + its redo launch)
                key = f"R{li['R']}W{li['W']}"
                per[key] = round(per.get(key, 0.0) + li["ms"], 1)
            out.append(str(per))
    print(f"rowsweep {var if var else 'auto'}  " + "   ".join(out), flush=True)
