#!/usr/bin/env python3
"""Headline benchmark: Gcell-updates/s of the profile-HMM scan (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both command shapes work for N > 1: under torch.distributed.run every process is one rank (RANK /
LOCAL_RANK / WORLD_SIZE from the environment); started plainly, `python bench.py --gpus N` starts its
own N rank processes (children created before anything in this process touches the GPU, rendezvous
on 127.0.0.1), passes rank 0's JSON line through and exits with the children's return code.

Workload (config.workload): BASELINE.json configs[2] "Pfam-A-like 20k protein profiles x
1 kbp queries": 20 000 sampled profiles (protein_profile_sample semantics, seeds 0xDEC1F0+p,
core sizes clip(round(exp(N(ln 150, 0.6^2))), 30, 2000), seed 20000, OCCUPANCY entry, eps 0.01),
uniform-ACGT 1 000-nt queries (seed 0x5E9+q), multi_hits, lrt threshold 10.  A step = one pass of
the scan path over one batch of 1 000*N distinct queries against the whole (sharded) DB, inputs
resident in HBM; 10 steps at N=1 are the full 10 000-query config.  With N ranks the DB is sharded
by cells and the hits are all-gathered over RCCL each step (weak scaling: per-GPU pairs fixed).
"""
import argparse
import importlib.util
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def load_product():
    name = "deciphon_old_amd"
    if name in sys.modules:
        return sys.modules[name]
    path = os.path.join(ROOT, "deciphon-old_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(name, path, submodule_search_locations=[os.path.dirname(path)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


WORKLOADS = {
    # name: (nprofiles, core-size rule, query length, queries per step per GPU, total queries)
    "c3": dict(nprof=20000, qlen=1000, qstep=1000, label="C3 Pfam-A-like 20k profiles x 1 kbp queries"),
    "c2": dict(nprof=1000, qlen=300, qstep=1000, label="C2 1k-profile synthetic DB x 300 bp queries"),
    # mixed-length stress (BASELINE configs[4]): M log-uniform 50..2000 (seed 50), L log-uniform 100..10000
    # 4 096 queries per step: the "dynamic batching" of the config is the length sort inside a scan, and 256-query
    # blocks of a 1 000-query batch each span a quarter of the length range (915 Gcell/s in round 2)
    "c5": dict(nprof=20000, qlen=0, qstep=4096, label="C5 mixed-length stress: 20k profiles (50-2000 states) x queries 100 bp-10 kbp"),
}


def core_sizes_for(workload, nprof):
    if workload == "c2":
        p = np.arange(nprof)
        return (100 + (p * 37) % 201).astype(np.uint32)
    if workload == "c5":
        rng = np.random.default_rng(50)
        return np.round(np.exp(rng.uniform(np.log(50.0), np.log(2000.0), nprof))).astype(np.uint32)
    rng = np.random.default_rng(20000)
    return np.clip(np.round(np.exp(rng.normal(np.log(150.0), 0.6, nprof))), 30, 2000).astype(np.uint32)


def make_queries(q_begin, q_end, qlen):
    """qlen > 0: fixed-length batch [n, qlen]; qlen == 0: list of log-uniform 100..10000-nt queries."""
    if qlen == 0:
        out = []
        for q in range(q_begin, q_end):
            rng = np.random.default_rng(0x5E9 + q)
            n = int(round(np.exp(rng.uniform(np.log(100.0), np.log(10000.0)))))
            out.append(rng.integers(0, 4, n, dtype=np.uint8))
        return out
    out = np.empty((q_end - q_begin, qlen), np.uint8)
    for i, q in enumerate(range(q_begin, q_end)):
        out[i] = np.random.default_rng(0x5E9 + q).integers(0, 4, qlen, dtype=np.uint8)
    return out


def plant_hits(dcp, queries, q_begin, sizes, cfg, every=100):
    """Planted-hit variant: every `every`-th query carries the most likely codon of
    each match state of one profile of the DB (flanked by its own random bases), i.e. a real hit.
    Any rank can rebuild the profile from its seed, so all ranks plant the same sequences."""
    planted = []
    nq, qlen = queries.shape
    for i in range(nq):
        q = q_begin + i
        if q % every != every // 2:
            continue
        p = (q * 2654435761) % len(sizes)
        prof = dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}")
        codon = prof.match_dist[:, 4:].reshape(-1, 5, 5, 5)[:, :4, :4, :4].reshape(-1, 64).argmax(axis=1)
        core = np.stack([(codon >> 4) & 3, (codon >> 2) & 3, codon & 3], axis=1).reshape(-1).astype(np.uint8)
        core = core[: max(3, (qlen - 60) // 3 * 3)]
        at = (qlen - len(core)) // 2
        queries[i, at:at + len(core)] = core
        planted.append((q, p))
    return planted


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_native_oracle():
    """The CPU-baseline legs time a build of oracle/oracle.c made ON THIS BOX with the flags
    BASELINE.md §3 states (-O3 -march=native -fopenmp, +LTO like the reference's Release IPO,
    CMakeLists.txt:111-115); the in-tree liboracle_f32.so is built without -march=native because
    it travels between machines.  Building the checker is not using it.  Returns (path, flags);
    falls back to the in-tree build (and says so) when gcc is missing."""
    import subprocess
    import tempfile
    flags = ["-O3", "-march=native", "-flto", "-fPIC", "-std=c11", "-fopenmp", "-ffp-contract=off", "-fno-fast-math"]
    src = os.path.join(ROOT, "oracle", "oracle.c")
    out_dir = tempfile.mkdtemp(prefix="dcp_oracle_native_")
    out = os.path.join(out_dir, "liboracle_f32.so")
    try:
        subprocess.check_call(["gcc"] + flags + ["-shared", "-o", out, src, "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out, "gcc " + " ".join(flags)
    except (OSError, subprocess.CalledProcessError):
        return None, "in-tree oracle/liboracle_f32.so (-O3, no -march=native: gcc unavailable on this box)"


def cpu_baseline(dcp, sizes, workload, qlen, budget_s=15.0, mode=0, lib=None):
    """The oracle (a from-scratch port of thread_run + imm_dp_viterbi; the reference itself cannot be
    built here) timed on this box's host cores on a bounded sample of the same workload.
    mode 0: reference-faithful -- per (sequence, profile) pair protein_profile_setup, generic graph
            Viterbi for null and alt, LRT; serial over sequences, OpenMP schedule(static,1) over
            count-balanced partitions (scan.c:227-258).
    mode 2: the optimised CPU variant -- DB resident (tables exported once per profile,
            outside the timed pair loop), end-indexed recursion of SURVEY Appendix B, null score once
            per (sequence, distinct null table), no allocation and no barrier inside the pair loop."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_py import Oracle  # checker / baseline only

    lib_path, flags = lib if lib else (None, "in-tree oracle/liboracle_f32.so")
    orc = Oracle(32, lib_path)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(cores, 64)  # NUM_THREADS, limits.h:8
    nprof = max(threads * 2, 16)
    stride = max(1, len(sizes) // nprof)
    pidx = list(range(0, len(sizes), stride))[:nprof]
    profs = [orc.sample(0xDEC1F0 + p, int(sizes[p])) for p in pidx]
    sumM = int(sum(int(sizes[p]) for p in pidx))

    def run(seqs):
        if mode == 0:
            t = time.perf_counter()
            orc.scan(profs, seqs, True, False, 10.0, threads, 0)
            return time.perf_counter() - t, 0.0
        _, _, _, tp, td = orc.scan_resident(profs, seqs, True, False, 10.0, threads, want_scores=False)
        return td, tp

    # calibrate on two queries per thread-round, then size the sample for ~budget_s of CPU work
    q = make_queries(0, 2, qlen)
    dt1 = max(run([bytes(q[0]), bytes(q[1])])[0] / 2, 1e-4)
    nq = int(max(2, min(4096, budget_s / dt1)))
    q = make_queries(0, nq, qlen)
    dt, prep = run([bytes(q[i]) for i in range(nq)])
    cells = sumM * nq * qlen
    algo = ("per pair: protein_profile_setup + generic graph Viterbi null+alt + LRT (thread_run restatement), "
            "serial over sequences, OpenMP schedule(static,1) over count-balanced partitions" if mode == 0 else
            "DB resident (tables exported once per profile: %.2f s, not timed), end-indexed recursion "
            "(SURVEY Appendix B), null once per (sequence, null table), partitions run all sequences "
            "without a per-sequence barrier" % prep)
    return {"value": round(cells / dt / 1e9, 4), "unit": "Gcell/s", "cores": threads, "kind": "port",
            "nproc": cores, "cpu_model": cpu_model_name(), "build": flags,
            "sample": f"{len(pidx)} profiles (every {stride}th of the DB, sum M={sumM}) x {nq} queries x {qlen} nt, "
                      f"{algo}, float32, {threads} threads, {dt:.1f} s"}


def parity_sample(dcp, sc, sizes, shard_begin, qlen, q_range, nsample=48):
    """Outside the timed region: re-scan a few queries of the benchmarked batch keeping dense scores and
    compare a random sample of (query, profile) pairs with the CPU oracle on the same synthetic inputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_py import Oracle  # checker only

    orc = Oracle(32)
    q0, q1 = q_range[0], min(q_range[1], q_range[0] + 8)
    batch = make_queries(q0, q1, qlen)  # the first queries of the first timed step, uploaded again
    sc.upload_seqs([np.ascontiguousarray(x) for x in batch])
    sc.scan(True, False, 10.0, keep_scores=True, sync=True)
    nl, al = sc.scores()
    rng = np.random.default_rng(12345)
    worst, n = 0.0, 0
    queries = {q: bytes(batch[q - q0]) for q in range(q0, q1)}
    for _ in range(nsample):
        q = int(rng.integers(q0, q1))
        p = int(rng.integers(0, sc.nprofiles))
        op = orc.sample(0xDEC1F0 + shard_begin + p, int(sizes[shard_begin + p]))
        op.setup(len(queries[q]), True, False)
        _, on, oa = op.viterbi_fast(queries[q])
        worst = max(worst, abs(nl[q - q0, p] - on) / abs(on), abs(al[q - q0, p] - oa) / abs(oa))
        n += 1
    return {"pairs": n, "max_rel_err_vs_oracle": float("%.3g" % worst), "tolerance": 5e-5, "ok": bool(worst <= 5e-5)}


def profiler_preload():
    """A profiler's preloaded library (rocprofv3 and friends) initialises the GPU in THIS process before main()
    runs; starting rank children from such a process is the exec-after-HIP-init hop this pool forbids.  Returns
    what gives the preload away, or None."""
    pre = os.environ.get("LD_PRELOAD", "")
    if any(k in pre for k in ("rocprof", "roctracer", "rocprofiler")):
        return "LD_PRELOAD=" + pre
    for k in ("ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"):
        if os.environ.get(k):
            return k + "=" + os.environ[k]
    return None


def self_launch(argv, ngpus, grace_s=20.0):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves.

    The parent imports neither torch nor the product library and makes no HIP call: the ranks are plain child
    processes -- never an exec of a process that initialised HIP -- with RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR=127.0.0.1 and a free MASTER_PORT in their environment, exactly what torch.distributed.run
    would give them.  (Under a profiler that is not true -- its preload touches the GPU in the parent -- so
    main() refuses to self-launch there: profile ONE rank directly, `rocprofv3 ... -- python3 bench.py --gpus 1`,
    or run the profiler under torch.distributed.run.)  Rank 0 inherits stdout (its one JSON line is the
    parent's output), the other ranks' stdout goes to stderr.  Returns the first non-zero child return code (a
    failed rank leaves the others blocked in a collective: they are given `grace_s`, then terminated by PID).
    If the parent itself is told to stop (SIGTERM from `timeout`, Ctrl-C) or dies on an exception, exactly
    the PIDs started here are terminated, waited for, then killed: no rank is left holding a GPU in a collective."""
    import signal
    import socket
    import subprocess

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []

    def reap(alive):
        for pr in alive:
            if pr.poll() is None:
                pr.terminate()  # exactly the PIDs started below
        for pr in alive:
            try:
                pr.wait(10)
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()

    class Stopped(Exception):
        pass

    def on_signal(signum, _frame):
        raise Stopped(signum)

    old = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    rc = 0
    try:
        for r in range(ngpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus), LOCAL_WORLD_SIZE=str(ngpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                       DCP_BENCH_SELF_LAUNCHED="1", DCP_BENCH_PARENT_PID=str(os.getpid()))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=None if r == 0 else sys.stderr))
        deadline = None
        alive = list(procs)
        while alive:
            for pr in list(alive):
                code = pr.poll()
                if code is None:
                    continue
                alive.remove(pr)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 128 - code
                    deadline = time.monotonic() + grace_s
            if deadline is not None and alive and time.monotonic() > deadline:
                break
            time.sleep(0.05)
    except Stopped as st:
        rc = 128 + int(st.args[0])
    finally:
        for sig in old:
            signal.signal(sig, signal.SIG_IGN)  # a second signal must not interrupt the clean-up
        reap([pr for pr in procs if pr.poll() is None])
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


def die_with_parent():
    """A self-launched rank asks the kernel for SIGTERM when its launcher dies (PR_SET_PDEATHSIG): a parent killed
    with SIGKILL cannot run its clean-up, and the ranks would stay blocked in a collective holding the GPUs."""
    ppid = os.environ.get("DCP_BENCH_PARENT_PID")
    if not ppid:
        return
    try:
        import ctypes
        import signal
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)  # PR_SET_PDEATHSIG = 1
        if os.getppid() != int(ppid):  # the launcher died before the prctl took effect
            sys.exit(143)
    except (OSError, AttributeError, ValueError):
        pass


def multi_gpu_block(dist, torch, device, rank, world, mine, transport, rccl_comm_count):
    """N > 1 diagnostics for the JSON line: every rank contributes
    [timed-leg seconds, cells scanned, shard begin, shard end, sum M of the shard, gather seconds, kernel ms,
     ranks its RCCL communicator reports]; rank 0 prints per-rank rows, min / max / imbalance and the transport.
    One all_gather of 8 doubles after the timed region."""
    v = torch.tensor([float(x) for x in mine], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(allv, v)
    rows = [t.cpu().tolist() for t in allv]
    if rank != 0:
        return None
    steps = max(1.0, rows[0][7])
    per_rank = [{"rank": r, "ms_per_step": round(x[0] / steps * 1e3, 3), "cells": int(x[1]),
                 "shard": [int(x[2]), int(x[3])], "shard_sum_m": int(x[4]),
                 "gather_ms_per_step": round(x[5] / steps * 1e3, 3),
                 "kernel_ms_per_step": round(x[6] / steps, 3),
                 "rccl_comm_count": int(rows[r][8]) if len(x) > 8 else None} for r, x in enumerate(rows)]
    ms = [p["ms_per_step"] for p in per_rank]
    sm = [p["shard_sum_m"] for p in per_rank]
    tiles = all(per_rank[r]["shard"][1] == per_rank[r + 1]["shard"][0] for r in range(world - 1)) and per_rank[0]["shard"][0] == 0
    return {"ranks": world, "transport": transport,
            "rccl_comm_count": rccl_comm_count,  # ncclCommCount of the C library's communicator on rank 0 (None: not in use)
            "shards_tile_the_db": bool(tiles),
            "sum_m_imbalance": round(max(sm) / (sum(sm) / world) - 1.0, 5) if sum(sm) else None,
            "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
            "time_imbalance": round(max(ms) / (sum(ms) / world) - 1.0, 5) if sum(ms) else None,
            "per_rank": per_rank}


def stub_rank(args, rank, world):
    """--stub-scan (tests/test_bench_launcher.py only): the N > 1 plumbing of this file on a box WITHOUT a
    GPU -- rank bootstrap on gloo, the shard map, the per-step hit gather through the same
    dcp_dist_merge_hits bookkeeping, max-over-ranks timing, one JSON line from rank 0, return codes --
    with the scan itself replaced by fabricated hit records.  It measures nothing and says so."""
    import torch
    import torch.distributed as dist

    load_product()
    from deciphon_old_amd import dist as ddist
    from deciphon_old_amd import HIT_DTYPE

    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = WORKLOADS[args.workload]
    nprof = args.nprof or wl["nprof"]
    qstep = (args.qstep or wl["qstep"]) * world
    sizes = core_sizes_for(args.workload, nprof)
    b, e = ddist.shard_range(sizes, world, rank)
    if args.stub_fail_rank == rank:
        sys.exit(7)  # rc propagation through the launcher
    piddir = os.environ.get("DCP_BENCH_STUB_PIDDIR")
    if piddir:  # the launcher clean-up test wants to know which PIDs to look for afterwards
        open(os.path.join(piddir, "rank%d.pid" % rank), "w").write(str(os.getpid()))
    cap = 64
    words = torch.zeros((cap, 4), dtype=torch.int32)
    count = torch.zeros(1, dtype=torch.int32)
    seen = 0

    def step(i):
        nonlocal seen
        n = (i + rank) % 5  # this rank's fabricated hits of step i: shard-local profile indices
        rec = np.zeros(n, HIT_DTYPE)
        rec["seq_idx"] = (np.arange(n) * 7 + i) % qstep
        rec["profile_idx"] = (np.arange(n) * 13 + rank) % max(1, e - b)
        rec["null_loglik"], rec["alt_loglik"] = -100.0, -90.0
        words.zero_()
        if n:
            words[:n] = torch.from_numpy(rec.view(np.int32).reshape(n, 4))
        count[0] = n
        if args.stub_sleep:
            time.sleep(args.stub_sleep)
        tg = time.perf_counter()
        h = ddist.gather_hits(words, count, b)
        gather_s[0] += time.perf_counter() - tg
        assert len(h) == sum((i + r) % 5 for r in range(world))
        assert (np.diff(h["seq_idx"].astype(np.int64)) >= 0).all() and (h["profile_idx"] < nprof).all()
        seen += len(h)

    gather_s = [0.0]
    for i in range(args.warmup):
        step(i)
    dist.barrier()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    mine_s = time.perf_counter() - t0
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    bounds = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(bounds, torch.tensor([b, e]))
    sum_m = int(sizes[b:e].sum())
    mg = multi_gpu_block(dist, torch, "cpu", rank, world,
                         [mine_s, float(sum_m) * qstep * 1000 * args.steps, b, e, sum_m, gather_s[0], 0.0, args.steps, -1],
                         "gloo + dcp_dist_merge_hits (stub)", None)
    if rank == 0:
        print(json.dumps({"metric": "STUB (launcher test: no scan ran, nothing was measured)", "value": 0.0,
                          "unit": "Gcell/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(float(tt.item()) / max(1, args.steps) * 1e3, 3),
                          "stub": True, "transport": "gloo + dcp_dist_merge_hits",
                          "shards": [[int(x) for x in t] for t in bounds], "hits_gathered": seen,
                          "multi_gpu": mg,
                          "self_launched": os.environ.get("DCP_BENCH_SELF_LAUNCHED") == "1"}))
    dist.barrier()
    dist.destroy_process_group()


class TimedLeg:
    """What the timed region measured, captured INSIDE it (the later legs scan on the same context)."""

    def __init__(self):
        self.per_class = {}      # (R, W) -> ms / algorithmic bytes / cells / launches, HIP events on the launch stream
        self.kernel_ms = 0.0
        self.redo_pairs = 0
        self.kernels = {}        # dcp_gpu_last_scan_kernel of every timed step -> steps
        self.elapsed = 0.0
        self.elapsed_rank = 0.0

    @property
    def kernel(self):
        """The kernel the timed steps ran with (the one most steps ran with, should they differ)."""
        return max(self.kernels, key=self.kernels.get) if self.kernels else None


def timed_leg(sc, step, first, last, fence):
    """EXACTLY last - first steps between two fences; launch records and the kernel id are read per step."""
    cap = TimedLeg()
    t0 = time.perf_counter()
    for i in range(first, last):
        step(i)
        for li in sc.launch_infos():  # HIP events on the launch stream, read after the sync
            k = (li["R"], li["W"])
            acc = cap.per_class.setdefault(k, dict(ms=0.0, bytes=0, cells=0, launches=0))
            acc["ms"] += li["ms"]
            acc["bytes"] += li["algorithmic_bytes"]
            acc["cells"] += li["cells"]
            acc["launches"] += 1
        cap.kernel_ms += sc.last_scan_ms
        cap.redo_pairs += sc.last_scan_redo_pairs
        kid = sc.last_scan_kernel
        cap.kernels[kid] = cap.kernels.get(kid, 0) + 1
    cap.elapsed_rank = time.perf_counter() - t0  # this rank's own steps, before it waits for the others
    fence()
    cap.elapsed = time.perf_counter() - t0
    return cap


def small_batches_leg(sc, dcp, src, qlen, sum_m, fetch_hits):
    """N = 1, after the timed regions: what the reference's own loop sees -- it scans ONE sequence at a time
    (src/server/scan.c:227-258) -- and batches of 8 and 64: upload + scan of the whole DB + hits on the host per
    batch, automatic kernel choice (the row sweep at these sizes), median of three."""
    small = {}
    for nq_small in (1, 8, 64):
        if nq_small > len(src):
            continue
        batch = src[:nq_small]
        ts = []
        for _rep in range(4):
            t0 = time.perf_counter()
            if qlen:
                sc.upload_seqs_flat(np.ascontiguousarray(batch).reshape(-1),
                                    (np.arange(nq_small + 1, dtype=np.uint64) * qlen).astype(np.uint32))
            else:
                sc.upload_seqs(batch)
            sc.scan(True, False, 10.0, keep_scores=False, sync=True, kernel=dcp.KERNEL_AUTO)
            fetch_hits()
            ts.append(time.perf_counter() - t0)
        t_med = sorted(ts[1:])[1]
        cells_small = sum_m * float(sum(len(x) for x in batch))
        small[str(nq_small)] = {"ms": round(t_med * 1e3, 2), "kernel_ms": round(sc.last_scan_ms, 2),
                                "seqs_per_sec": round(nq_small / t_med, 1),
                                "gcells_per_s": round(cells_small / t_med / 1e9, 1)}
    small["what"] = ("per batch: host sequences -> upload -> scan of the whole resident DB (automatic kernel choice: "
                     "the row sweep at these sizes) -> hit records on the host; median of 3 after one untimed pass")
    return small


def roofline_block(timed, dcp, workload, steps, sizes, b, e, qstep, world, qlen):
    """The `roofline` object of the JSON line, from the TIMED leg's capture only (never from the context's
    current state: tests/test_bench_roofline.py runs a kernel-switching later leg in between)."""
    per_class, kernel_ms, redo_pairs = timed.per_class, timed.kernel_ms, timed.redo_pairs
    dom_key = max(per_class, key=lambda k: per_class[k]["ms"])
    dom = per_class[dom_key]
    dom_ms = dom["ms"] / dom["launches"]
    dom_cells = dom["cells"] / dom["launches"]
    dom_algo_bytes = dom["bytes"] / dom["launches"]
    is_qlane = not dom_key[1]
    kname = (f"viterbi_rowsweep_kernel<R={dom_key[0]},W={dom_key[1]}>" if dom_key[1] else
             f"viterbi_qlane{'2' if timed.kernel == dcp.KERNEL_QLANE2 else ''}_kernel<KT={dom_key[0]}>")
    # What binds the dominant kernel: VALU ISSUE.  28 max/add lane-ops per cell
    # (11 for M_k, 10 for I_k, 3 for D_k, 4 for E/B) against the chip's f32 VALU rate:
    # 256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz = 78.6e12 lane-ops/s (= 157.3 TFLOPS / 2: max and
    # add, no FMA to fuse).  SURVEY 8d's HBM model (20 B/cell) is reported as `algorithmic_hbm`:
    # those bytes are LDS gathers in the query-lane kernel and L2 hits in the row sweep.
    VALU_PEAK = 78.6e12
    OPS_PER_CELL = 28
    lane_ops = dom_cells * OPS_PER_CELL / (dom_ms * 1e-3)
    # HBM bytes per launch of that kernel: PMC counters cannot be read inside this process, so the
    # figure is REPLAYED from the committed rocprofv3 --pmc passes of this same command (see
    # traffic_source); null when no matching profile is committed.
    traffic, traffic_source = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc_hbm.json")))
        if pmc.get("workload") == workload and pmc.get("queries_per_step") == qstep and world == 1:
            for pmc_name, pmc_entry in pmc["kernels"].items():
                if kname.split("<")[0] in pmc_name and "hbm_bytes_per_launch" in pmc_entry:
                    traffic = pmc_entry["hbm_bytes_per_launch"]
                    traffic_source = ("replayed (not measured in this run) from profiles/latest_pmc_hbm.json: "
                                      + pmc.get("note", ""))
    except (OSError, ValueError, KeyError):
        pass
    # What the VALU can do with THIS instruction mix: the row's arithmetic alone, measured in a registers-only
    # microbenchmark (profiles/microbench/row_valu.hip; replayed like the PMC traffic).  28 ops per cell at
    # the 2-cycle rate is not attainable for a max-plus recursion on gfx950: v_max_f32 / v_max3_f32 --
    # a third of the ops -- and adds with an SGPR operand issue at half that rate (profiles/r03/valu_issue.txt).
    valu_ceiling = None
    try:
        vc = json.load(open(os.path.join(ROOT, "profiles", "latest_valu_ceiling.json")))
        cyc = float(vc["cycles_per_wavefront_row"])
        at_spec = 256 * 4 * 64 * vc["nodes_per_lane"] / cyc * 2.4e9
        valu_ceiling = {"cycles_per_wavefront_row_arithmetic_only": cyc,
                        "gcells_per_s_at_2.4GHz": round(at_spec / 1e9, 1),
                        "frac_of_ceiling": (round(dom_cells / (dom_ms * 1e-3) / at_spec, 4) if is_qlane else None),
                        "source": "replayed from profiles/latest_valu_ceiling.json: " + vc["source"]}
    except (OSError, ValueError, KeyError):
        pass
    # analytic HBM traffic of the design, to hold against the measured figure:
    #   query lane: 12 B written + 12 B read per (row, lane, tile boundary) -- the Xm/Xd/Em planes --
    #               plus every tile image once per (profile, 256-query block)
    #   compulsory: SURVEY 8d note 2 -- 548*M + L + 8 bytes per pair (compact profile streamed once)
    ntiles = (sizes[b:e].astype(np.int64) + 7) // 8
    # (weak scaling: every rank scans ALL `qstep` queries of the step against its shard b..e)
    lanes = ((qstep + 63) // 64) * 64 if qlen else None
    two_stage = timed.kernel == dcp.KERNEL_QLANE2
    hbm_boundaries = ((ntiles - 1) // 2) if two_stage else (ntiles - 1)  # odd -> even only / every boundary
    scratch_bytes = (int(24 * int(hbm_boundaries.sum()) * lanes * qlen) if (is_qlane and qlen) else None)
    tile_bytes = (int(ntiles.sum()) * 8 * 1364 * 4 * ((qstep + 255) // 256) if is_qlane else None)
    npairs_launch = (e - b) * qstep
    compulsory = int(548 * int(sizes[b:e].sum()) * qstep + (qlen + 8) * npairs_launch) if qlen else None
    roof = {
        "bound": "valu-issue",
        "kernel": kname,
        # dcp_gpu_last_scan_kernel of every timed step (1 row sweep, 2 single-stage, 3 two-stage query lane)
        "timed_step_kernels": {str(k): v for k, v in sorted(timed.kernels.items())},
        "achieved": round(lane_ops / 1e12, 3), "peak": round(VALU_PEAK / 1e12, 1), "unit": "Tlane-op/s",
        "frac": round(lane_ops / VALU_PEAK, 4),
        "ops_per_cell": OPS_PER_CELL,
        "valu_only_ceiling": valu_ceiling,
        "peak_note": ("256 CUs x 4 SIMDs x 32 lanes/clk x 2.4 GHz (spec clock); measured clock under this load: two-stage "
                      "query-lane kernel 2.23 GHz, single-stage 1.81 GHz, row sweep 2.3-2.4 GHz (profiles/r02/*pmc*)"),
        "avg_launch_ms": round(dom_ms, 3),
        "cells_per_launch": int(dom_cells),
        "gcells_per_s": round(dom_cells / (dom_ms * 1e-3) / 1e9, 1),
        "traffic": traffic,
        "traffic_source": traffic_source,
        "hbm": {
            "peak_gbs": 8000.0,
            "measured_gbs": (round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic else None),
            "measured_frac": (round(traffic / (dom_ms * 1e-3) / 8e12, 4) if traffic else None),
            # north_star's "achieved HBM GB/s against the chip's peak", spelled out: counter bytes of THIS kernel
            # (replayed, see traffic_source) / this run's average launch time / 8 TB/s
            "achieved_gbs": (round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic else None),
            "achieved_frac": (round(traffic / (dom_ms * 1e-3) / 8e12, 4) if traffic else None),
            "analytic_scratch_plane_bytes_per_launch": scratch_bytes,
            "analytic_tile_image_bytes_per_launch": tile_bytes,
            "compulsory_bytes_per_launch": compulsory,
        },
        # the row sweep reads SURVEY 8d's 20 B/cell of match emissions on chip: since round 3 the rows of the
        # 1- and 2-base words (8 of the 20 B) from the block's LDS image, the other 12 B/cell from the XCD's L2
        # (the profile's table is L2-resident while its queries run): against the L2 peak
        # (MI355X_MICROARCH.md: 34.5 TB/s; its measured rate for gathering L2-resident rows is 16.8-18.8 TB/s)
        "l2_gather": (None if is_qlane else {
            "bytes_per_launch": int(dom_algo_bytes * 12 / 20),
            "bytes_per_cell": 12,
            "gbs": round(dom_algo_bytes * 0.6 / (dom_ms * 1e-3) / 1e9, 1),
            "peak_gbs": 34500.0,
            "frac": round(dom_algo_bytes * 0.6 / (dom_ms * 1e-3) / 34.5e12, 3),
            "lds_bytes_per_cell": 8,
        }),
        # SURVEY.md 8d's per-pair byte model 20*M*L + 32*(M+1) + L + 8: NOT a bound for this design
        "algorithmic_hbm": {
            "bytes_per_launch": int(dom_algo_bytes),
            "gbs": round(dom_algo_bytes / (dom_ms * 1e-3) / 1e9, 1),
            "ratio_to_hbm_peak": round(dom_algo_bytes / (dom_ms * 1e-3) / 8e12, 3),
            "note": ("served from LDS: the 20 B/cell of match emissions are ds_read_b128 gathers from the "
                     "LDS-resident tile image, never HBM reads -- a ratio above 1 is on-chip reuse, not a bound"
                     if is_qlane else
                     "served from L2/Infinity Cache: many queries re-read one profile's table rows"),
        },
        # where the query-lane kernel reads those bytes from: LDS, 256 B/clk/CU x 256 CUs x 2.4 GHz
        # = 157 TB/s conflict-free (random 16-byte gathers serialise ~2.1x)
        "lds": ({"bytes_per_cell": 20, "peak": 157286.0, "unit": "GB/s",
                 "frac": round(dom_cells / (dom_ms * 1e-3) * 20 / 157.286e12, 4)} if is_qlane else None),
        "kernel_ms_per_step": round(kernel_ms / steps, 3),
        "per_class_ms_per_step": {(f"R{k[0]}W{k[1]}" if k[1] else f"qlane_KT{k[0]}"): round(v["ms"] / steps, 3)
                                  for k, v in sorted(per_class.items())},
        "per_class_gcells_per_s": {(f"R{k[0]}W{k[1]}" if k[1] else f"qlane_KT{k[0]}"):
                                   round(v["cells"] / (v["ms"] * 1e-3) / 1e9, 1)
                                   for k, v in sorted(per_class.items()) if v["cells"]},
        # query-lane scans: pairs with multi-hit feedback are re-scored by the row-sweep
        # launches (R*W* above) right after the query-lane kernel
        "redo_pairs_per_step": round(redo_pairs / steps, 1),
    }
    return roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--nprof", type=int, default=0, help="override profile count (debug)")
    ap.add_argument("--qstep", type=int, default=0, help="override queries per step per GPU (debug)")
    ap.add_argument("--qlen", type=int, default=0, help="override query length (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", default="auto", choices=["auto", "rowsweep", "qlane", "qlane2"],
                    help="dcp_scan_params.kernel (auto = the library's cost model)")
    ap.add_argument("--hit-gather", default="c", choices=["c", "torch"],
                    help="N>1: hit gather through the C host's RCCL path (dcp_dist_*) or torch.distributed")
    ap.add_argument("--dense", type=int, default=0, metavar="F",
                    help="hit-dense stress: the DB is F distinct profiles replicated nprof/F times and EVERY query "
                         "carries one family's consensus, so ~1/F of all pairs are real hits (redo-list regime); "
                         "with --dense-random the same DB is scanned with random queries (the reference rate)")
    ap.add_argument("--dense-random", action="store_true")
    ap.add_argument("--planted", action="store_true",
                    help="planted-hit variant: 1 %% of the queries carry a real hit (exercises hits / gather)")
    ap.add_argument("--e2e-steps", type=int, default=-1,
                    help="steps of the end-to-end leg (each step uploads its own sequences, scans, fetches the hits "
                         "to the host); default min(steps, 5), 0 = skip")
    ap.add_argument("--as-rank", default="", metavar="R/N",
                    help="one-GPU rehearsal of an N-GPU run's rank R: this GPU holds shard R of N of the DB and scans the "
                         "N x larger query step an N-GPU run uses (value = this one rank's rate; --gpus 1 only)")
    ap.add_argument("--one-layout", action="store_true",
                    help="DCP_DB_ONE_LAYOUT: only the row-sweep tables resident, the query-lane kernels gather their tile "
                         "images from them (half the DB's footprint; not the default, not the headline line)")
    ap.add_argument("--stub-scan", action="store_true", help=argparse.SUPPRESS)  # launcher test on CPU (gloo)
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--stub-sleep", type=float, default=0.0, help=argparse.SUPPRESS)  # seconds per stub step
    ap.add_argument("--launch-grace", type=float, default=20.0, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process only starts and reaps the N ranks
        why = profiler_preload()
        if why:
            sys.stderr.write("bench.py: refusing to start %d rank processes from a process a profiler has preloaded "
                             "(%s): its library initialises the GPU here, and a child started from a GPU-initialised "
                             "process is forbidden on this pool.  Profile one rank directly (`--gpus 1`), or put the "
                             "profiler under torch.distributed.run.\n" % (args.gpus, why))
            sys.exit(2)
        sys.exit(self_launch(sys.argv[1:], args.gpus, args.launch_grace))
    die_with_parent()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    if args.stub_scan:
        return stub_rank(args, rank, world)

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("DCP_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    dcp = load_product()
    from deciphon_old_amd import dist as ddist

    wl = WORKLOADS[args.workload]
    nprof = args.nprof or wl["nprof"]
    qlen = args.qlen or wl["qlen"]
    shape_rank, shape_world = rank, world
    if args.as_rank:
        if world != 1:
            sys.exit("--as-rank is a one-GPU rehearsal: run it with --gpus 1")
        shape_rank, shape_world = (int(x) for x in args.as_rank.split("/"))
        if not 0 <= shape_rank < shape_world:
            sys.exit("--as-rank R/N needs 0 <= R < N")
    qstep = (args.qstep or wl["qstep"]) * shape_world
    sizes = core_sizes_for(args.workload, nprof)

    # ---- resident DB shard -------------------------------------------------------------
    t0 = time.perf_counter()
    b, e = ddist.shard_range(sizes, shape_world, shape_rank)
    cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
    nthreads = min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 8)
    if args.dense:
        fam = [dcp.ProteinProfile.sample(0xDEC1F0 + f, int(sizes[f]), cfg, f"FAM{f:02d}") for f in range(args.dense)]
        sizes = np.array([sizes[p % args.dense] for p in range(nprof)], np.uint32)
        b, e = ddist.shard_range(sizes, shape_world, shape_rank)
        profiles = [fam[p % args.dense] for p in range(b, e)]
    else:
        with ThreadPoolExecutor(max(1, nthreads // max(1, min(world, 8)))) as ex:
            profiles = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg, f"PF{p:05d}"),
                                   range(b, e)))
    t_build = time.perf_counter() - t0
    sc = dcp.Scanner(local_rank)
    t0 = time.perf_counter()
    sc.upload_db(profiles, one_layout=args.one_layout)
    db_tables = {"one_layout": bool(sc.one_layout), "table_bytes_after_upload": int(sc.table_bytes)}
    t_upload = time.perf_counter() - t0
    del profiles

    # ---- resident queries: every step scans its own distinct batch --------------------------------
    nsteps = args.steps + args.warmup
    queries = make_queries(0, nsteps * qstep, qlen)
    planted = plant_hits(dcp, queries, 0, sizes, cfg) if (args.planted and qlen) else []
    if args.dense and not args.dense_random and qlen:
        # every query carries the most likely codon of each match state of family q % F
        for f in range(args.dense):
            md = fam[f].match_dist
            codon = md[:, 4:].reshape(-1, 5, 5, 5)[:, :4, :4, :4].reshape(-1, 64).argmax(axis=1)
            core = np.stack([(codon >> 4) & 3, (codon >> 2) & 3, codon & 3], axis=1).reshape(-1).astype(np.uint8)
            core = core[: max(3, (qlen - 60) // 3 * 3)]
            at = (qlen - len(core)) // 2
            queries[f::args.dense, at:at + len(core)] = core
    if qlen:
        off = (np.arange(nsteps * qstep + 1, dtype=np.uint64) * qlen).astype(np.uint32)
        sc.upload_seqs_flat(queries.reshape(-1), off)
    else:
        sc.upload_seqs(queries)
    # the end-to-end leg re-uploads the batches of the first timed steps from these host copies
    n_e2e = (min(args.steps, 5) if args.e2e_steps < 0 else min(args.e2e_steps, args.steps)) if not args.dense else 0
    e2e_batches = [queries[i * qstep:(i + 1) * qstep] for i in range(args.warmup, args.warmup + n_e2e)]
    if qlen:
        e2e_batches = [np.ascontiguousarray(x) for x in e2e_batches]
    small_src = queries[:min(64, qstep)].copy() if qlen else list(queries[:min(64, qstep)])  # the small-batch leg's queries
    del queries

    cap = 1 << 16
    hit_words = torch.zeros((cap, 4), dtype=torch.int32, device="cuda")
    hit_count = torch.zeros(1, dtype=torch.int32, device="cuda")
    sc.set_hit_buffer(hit_words.data_ptr(), cap, hit_count.data_ptr())

    hits_seen = []
    # N > 1: the hit gather of the C host (dcp_dist_*: librccl loaded by the library, counts all-gather +
    # grouped send/recv), bootstrapped with torch.distributed (rank 0's ncclUniqueId is broadcast).  If the
    # C communicator cannot be created on every rank, all ranks fall back to the torch.distributed
    # transport (same C bookkeeping) and the JSON line says so.
    gather_kind = "torch.distributed all_gather + dcp_dist_merge_hits"
    cdist = None
    if (world > 1 or force_dist) and args.hit_gather == "c":
        idt = torch.zeros(ddist.ID_BYTES, dtype=torch.uint8, device="cuda")
        why = ""
        # every rank first proves it can load librccl and get an id of its own: a rank that could not would never
        # enter ncclCommInitRank, and the others would wait there for ever
        my_id = None
        try:
            my_id = ddist.CDist.unique_id()
        except Exception as ex:
            why = str(ex)
        can = torch.tensor([1 if my_id else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(can, op=dist.ReduceOp.MIN)
        if int(can.item()) == 1 and rank == 0:
            idt.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
        dist.broadcast(idt, 0)  # all zero when some rank cannot load librccl
        id_bytes = bytes(idt.cpu().numpy().tobytes())
        if any(id_bytes):
            try:
                cdist = ddist.CDist.create(id_bytes, rank, world, local_rank)
            except Exception as ex:
                why = str(ex)
        ok = torch.tensor([1 if cdist else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            gather_kind = "C host dcp_dist_gather_hits: librccl all-gather of counts + grouped ncclSend/ncclRecv"
        else:
            if cdist:
                cdist.close()
            cdist = None
            gather_kind += f" (C RCCL communicator unavailable{': ' + why if why else ''})"
    kernel_id = {"auto": dcp.KERNEL_AUTO, "rowsweep": dcp.KERNEL_ROWSWEEP, "qlane": dcp.KERNEL_QLANE,
                 "qlane2": dcp.KERNEL_QLANE2}[args.kernel]

    cdist_state = {"comm": cdist, "kind": gather_kind}

    gather_s = [0.0]

    def step(i):
        sc.scan(True, False, 10.0, keep_scores=False, sync=False, q_range=(i * qstep, (i + 1) * qstep),
                kernel=kernel_id)
        sc.sync()
        tg = time.perf_counter()
        if cdist_state["comm"]:
            h, total = cdist_state["comm"].gather_scan_hits(sc, b)
            if i < args.warmup:
                # untimed cross-check of the C gather against the torch.distributed transport, record for record;
                # every rank takes the same decision, and a mismatch demotes the run to the torch transport
                ref = ddist.gather_hits(hit_words, hit_count, b)
                same = torch.tensor([1 if (total == len(h) == len(ref) and np.array_equal(h, ref)) else 0],
                                    dtype=torch.int32, device="cuda")
                dist.all_reduce(same, op=dist.ReduceOp.MIN)
                if int(same.item()) != 1:
                    cdist_state["comm"].close()
                    cdist_state["comm"] = None
                    cdist_state["kind"] = "torch.distributed all_gather + dcp_dist_merge_hits (the C RCCL gather DISAGREED in warm-up)"
                    h = ref
        elif world > 1 or force_dist:
            h = ddist.gather_hits(hit_words, hit_count, b)
        else:
            h = None
        gather_s[0] += time.perf_counter() - tg
        if args.planted:  # a few hundred records: negligible beside the scan
            if h is None:
                rec = hit_words[:int(hit_count.item())].cpu().numpy()
                hits_seen.append({(int(r[0]), int(r[1]) + b) for r in rec})
            else:
                hits_seen.append({(int(r["seq_idx"]), int(r["profile_idx"])) for r in h})
        return h

    def fence():
        sc.sync()
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    fence()
    gather_s[0] = 0.0
    # everything the roofline block says about "the dominant kernel" is captured HERE, inside the timed leg: the
    # later legs (e2e, small batches, parity sample) scan on the same context and move sc.last_scan_kernel
    timed = timed_leg(sc, step, args.warmup, nsteps, fence)
    per_class, kernel_ms, redo_pairs, elapsed = timed.per_class, timed.kernel_ms, timed.redo_pairs, timed.elapsed

    cells_rank = float(sum(v["cells"] for v in per_class.values()))

    # ---- end-to-end leg (SURVEY 8d "kernel-only AND end-to-end"): nothing of a step is resident but the DB.
    # Each step hands its own host sequences to the boundary (dcp_gpu_seqs_upload: H2D, 2-bit packing, the
    # 13 length-dependent transitions per query), scans them against the whole shard, and brings the hit
    # list back to the host (N = 1: D2H of the records; N > 1: the RCCL gather, which ends on the host).
    e2e = None
    if n_e2e:
        def e2e_step(k):
            batch = e2e_batches[k]
            if qlen:
                sc.upload_seqs_flat(batch.reshape(-1), (np.arange(len(batch) + 1, dtype=np.uint64) * qlen).astype(np.uint32))
            else:
                sc.upload_seqs(batch)
            sc.scan(True, False, 10.0, keep_scores=False, sync=False, kernel=kernel_id)
            sc.sync()
            if cdist_state["comm"]:
                h, _ = cdist_state["comm"].gather_scan_hits(sc, b)
            elif world > 1 or force_dist:
                h = ddist.gather_hits(hit_words, hit_count, b)
            else:
                h = hit_words[:min(cap, int(hit_count.item()))].cpu().numpy()
            return len(h) if h is not None else 0

        e2e_step(0)  # untimed: the sequence buffers are re-sized to one step's batch
        fence()
        t0 = time.perf_counter()
        e2e_hits = sum(e2e_step(k) for k in range(n_e2e))
        fence()
        e2e_elapsed = time.perf_counter() - t0
        e2e_cells = cells_rank / args.steps * n_e2e
        if world > 1:
            tt = torch.tensor([e2e_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cc = torch.tensor([e2e_cells], dtype=torch.float64, device="cuda")
            dist.all_reduce(cc, op=dist.ReduceOp.SUM)
            e2e_elapsed, e2e_cells = float(tt.item()), float(cc.item())
        e2e = {"value": round(e2e_cells / e2e_elapsed / 1e9, 3), "unit": "Gcell/s",
               "ms_per_step": round(e2e_elapsed / n_e2e * 1e3, 3), "steps": n_e2e,
               "seqs_per_sec": round(n_e2e * qstep / e2e_elapsed, 2), "hits_fetched": int(e2e_hits),
               "what": "per step: host sequences -> dcp_gpu_seqs_upload (H2D + pack + per-query transitions) -> "
                       "scan of the whole resident DB shard -> hit records on the host"
                       + (" through the hit gather" if (world > 1 or force_dist) else " (D2H)")
                       + "; the batches are those of the first timed steps of the resident leg"}
    # ---- small batches (N = 1 only): what the reference's own loop sees -- it scans ONE sequence at a time
    # (src/server/scan.c:227-258) -- and batches of 8 and 64: upload + scan of the whole DB + hits on the host per
    # batch, automatic kernel choice (the row sweep at these sizes), median of three; outside every timed region
    small = None
    if world == 1 and not force_dist and not args.dense and not args.stub_scan and args.kernel == "auto":
        small = small_batches_leg(sc, dcp, small_src, qlen, float(sizes[b:e].sum()),
                                  lambda: hit_words[:min(cap, int(hit_count.item()))].cpu().numpy())
    multi_gpu = None
    if world > 1 or force_dist:
        # per-rank figures of the timed leg BEFORE the max-over-ranks reduction hides them: a slow rank, an uneven
        # shard or a slow gather must be readable from the one JSON line of a run nobody can repeat by hand
        comm_count = cdist_state["comm"].comm_count if cdist_state["comm"] else -1
        multi_gpu = multi_gpu_block(dist, torch, "cuda", rank, world,
                                    [timed.elapsed_rank, cells_rank, b, e, int(sizes[b:e].sum()), gather_s[0],
                                     kernel_ms, args.steps, comm_count],
                                    cdist_state["kind"], comm_count if cdist_state["comm"] else None)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        cc = torch.tensor([cells_rank], dtype=torch.float64, device="cuda")
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)
        elapsed, cells_total = float(tt.item()), float(cc.item())
    else:
        cells_total = cells_rank

    if rank == 0:
        roof = roofline_block(timed, dcp, args.workload, args.steps, sizes, b, e, qstep, world, qlen)
        out = {
            "metric": "Gcell-updates/sec",
            "value": round(cells_total / elapsed / 1e9, 3),
            "unit": "Gcell/s",
            "seqs_per_sec": round(args.steps * qstep / elapsed, 2),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{wl['label']}: {nprof} sampled profiles (sum M={int(sizes.sum())}, mean {sizes.mean():.0f}), "
                            f"{qstep} distinct {str(qlen) + '-nt' if qlen else '100-10000-nt'} queries per step, multi_hits, lrt>=10",
                "profiles_per_gpu": e - b, "queries_per_step": qstep, "query_len": qlen,
                "kernel": args.kernel, "one_table_layout": bool(args.one_layout),
                "dense_families": args.dense or None, "dense_queries": (None if not args.dense else
                                                                         "random" if args.dense_random else "consensus"),
                "parallelism": f"profile-shard x{world}" + (f", RCCL hit gather per step ({cdist_state['kind']})" if (world > 1 or force_dist) else ""),
                "rehearsal": (None if not args.as_rank else
                              f"rank {shape_rank} of {shape_world} on ONE GPU: profiles {b}..{e} (sum M {int(sizes[b:e].sum())}) x the "
                              f"{qstep}-query step of a {shape_world}-GPU run; value is this one rank's rate, not a {shape_world}-GPU figure"),
            },
            "roofline": roof,
            "multi_gpu": multi_gpu,
            "e2e": e2e,
            "small_batches": small,
            "setup_s": {"profile_build": round(t_build, 1), "db_upload_expand": round(t_upload, 1)},
            "db_tables": dict(db_tables, table_bytes_at_end=int(sc.table_bytes)),
            # every DCP_* variable in the environment (the library reads none of them; bench.py reads
            # DCP_BENCH_FORCE_DIST only)
            "env_dcp": {k: v for k, v in sorted(os.environ.items()) if k.startswith("DCP_")},
        }
        if world == 1 and not args.no_cpu_baseline and qlen:
            native = build_native_oracle()
            out["cpu_baseline"] = cpu_baseline(dcp, sizes, args.workload, qlen, lib=native)
            out["cpu_baseline_optimised"] = cpu_baseline(dcp, sizes, args.workload, qlen, budget_s=8.0, mode=2, lib=native)
            out["parity_check"] = parity_sample(dcp, sc, sizes, b, qlen, (args.warmup * qstep, (args.warmup + 1) * qstep))
        if args.planted:
            found = set().union(*hits_seen) if hits_seen else set()
            out["planted_hits"] = {"planted": len(planted), "found": sum((q, p) in found for q, p in planted),
                                   "hits_total": len(found)}
        print(json.dumps(out))
    sc.close()
    if cdist_state["comm"]:
        cdist_state["comm"].close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
