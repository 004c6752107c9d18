#!/usr/bin/env python3
"""Randomised differential run of every scan kernel on an MI355X (test infrastructure; uses the CPU oracle).

    python3 tests/tools/fuzz_gpu.py [--seconds 420] [--seed 1] [--oracle-cells 6e7]

The parity tests in tests/test_gpu_parity.py fix their seeds and shapes.  This tool draws them: each round builds a
small database whose core sizes cluster around the kernels' size-class, segment and tile boundaries, a batch of
queries of mixed lengths (1 nt .. a few kbp, now and then tens of kbp) with a few planted hits, a scoring mode
(multi-hit / uni-hit / HMMER3-compatible) and then scans the batch

  * with the automatic kernel choice, the row sweep, both query-lane kernels (shipped library), and
  * with a random forced row-sweep variant of the tests' -DDCP_TEST_HOOKS build: rows staged in LDS, wavefronts per
    block, two-row prefetch, segment-major sweep on/off, K profiles per wavefront on/off, a tiny column budget, a tiny
    redo-list capacity; in half of the rounds on a one-layout DB (DCP_DB_ONE_LAYOUT), where the query-lane kernels run too.

All scans must give the same bits and the same hit list; a sample of pairs bounded by --oracle-cells (the long pairs
are drawn with the same probability as the short ones) must equal the oracle's float32 recursion on the product's own
tables, bit for bit.  The round's hits (and a few other pairs) are traced back twice -- forward pass by the row-sweep
kernels, and by the trace kernel's own loop -- and must give the same steps and the scan's score.  The first difference prints the round's seed and shapes and exits 1; `--seed S --rounds 1`
replays a round.  One line per round; no file of the reference is read.
"""
import argparse
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import conftest  # noqa: E402  (tests/conftest.py: load_product, Oracle)
from oracle_py import ENTRY_DIST_OCCUPANCY, ENTRY_DIST_UNIFORM  # noqa: E402
from test_gpu_parity import delete_heavy_params, pfam_like_params, planted_query  # noqa: E402

EDGES = (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 191, 192, 193, 255, 256, 257,
         319, 320, 321, 383, 384, 385, 447, 448, 449, 511, 512, 513, 640, 767, 768, 769, 1023, 1024, 1025, 1279, 1280,
         1281, 1535, 1536, 1537, 2047, 2048, 2049, 2559, 2560, 2561, 3071, 3072, 3073, 3583, 3584, 3585, 4095, 4096)


def draw_sizes(rng, n, cap):
    out = []
    for _ in range(n):
        u = rng.random()
        if u < 0.45:
            m = int(rng.choice(EDGES))
        elif u < 0.8:
            m = int(rng.integers(1, 600))
        else:
            m = int(rng.integers(1, 4097))
        out.append(min(m, cap))
    return out


def draw_params(rng, M):
    u = rng.random()
    if u < 0.6:
        prm = pfam_like_params(rng, M)
    elif u < 0.85:
        prm = delete_heavy_params(rng, M)
    else:  # flagged: positive MD / DD, the delete states decide E(j)
        null, match, trans = pfam_like_params(rng, M)
        trans = trans.copy()
        if M > 1:
            trans[1:M, 2] = np.float32(rng.random() * 0.9)
            trans[1:M, 6] = np.float32(rng.random() * 0.6)
        prm = (null, match, trans)
    return prm


def draw_lengths(rng, nq, small_db):
    top = int(rng.choice([12, 40, 150, 400, 1500, 4000]))
    lens = rng.integers(1, top + 1, nq)
    if small_db and rng.random() < 0.25:  # a few very long queries among the short ones
        for i in rng.choice(nq, min(nq, int(rng.integers(1, 4))), replace=False):
            lens[i] = int(rng.integers(5000, 40000))
    return lens


def same_bits(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


def one_round(dcp, oracle32, sc, hk, seed, oracle_cells, pool, big_every=0):
    rng = np.random.default_rng(seed)
    nprof = int(rng.choice([1, 2, 3, 5, 8, 13, 24, 40]))
    big = rng.random() < 0.3
    sizes = draw_sizes(rng, nprof, 4096 if big else 700)
    nq = int(rng.choice([1, 2, 5, 9, 21, 63, 64, 65, 100, 129, 200, 257, 400]))
    if sum(sizes) * nq > 3_000_000:  # keep a round's device work and the sample's spread in hand
        nq = max(1, 3_000_000 // sum(sizes))
    if big_every and seed % big_every == 0:
        # a round of many queries against few profiles: several 256-query blocks, slots holding several 64-query groups
        # of very different lengths (the query-lane kernels' dynamic batching), redo lists that fill
        nprof = int(rng.choice([1, 2, 4, 7]))
        sizes = draw_sizes(rng, nprof, 700)
        nq = int(rng.choice([513, 777, 1024, 1500, 2311]))
    entry = int(rng.choice([ENTRY_DIST_UNIFORM, ENTRY_DIST_OCCUPANCY]))
    eps = float(rng.choice([0.01, 0.05, 0.1]))
    cfg = dcp.ProteinCfg(entry, eps)
    params = [draw_params(rng, M) for M in sizes]
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    lens = draw_lengths(rng, nq, sum(sizes) < 3000 and nq <= 400)
    seqs = [rng.integers(0, 4, int(L), dtype=np.uint8) for L in lens]
    nplant = int(rng.integers(0, 4))
    for _ in range(nplant):
        p = int(rng.integers(0, nprof))
        if sizes[p] > 400:
            continue
        op = oracle32.new(*params[p], entry, eps)
        q = int(rng.integers(0, nq))
        body = planted_query(rng, op, sizes[p], flank=int(rng.integers(0, 40)))
        seqs[q] = np.concatenate([body, body]) if rng.random() < 0.3 else body
    multi = bool(rng.random() < 0.7)
    h3 = bool(multi and rng.random() < 0.2)
    on_host = bool(rng.random() < 0.5)
    shape = f"seed {seed}: {nprof} profiles {sizes if nprof <= 13 else sizes[:13] + ['...']} x {nq} queries " \
            f"(1..{max(len(s) for s in seqs)} nt) multi={int(multi)} h3={int(h3)} host_tables={int(on_host)}"

    results = {}
    sc.upload_db(profiles, expand_on_host=on_host)
    sc.upload_seqs(seqs)
    for name, k in (("auto", dcp.KERNEL_AUTO), ("rowsweep", dcp.KERNEL_ROWSWEEP), ("qlane", dcp.KERNEL_QLANE),
                    ("qlane2", dcp.KERNEL_QLANE2)):
        sc.scan(multi, h3, 10.0, kernel=k)
        n, a = sc.scores()
        results[name] = (n.copy(), a.copy(), sc.hits().copy())
    # a forced variant of the tests' build
    stage = int(rng.choice([0, 20, 84]))
    waves = int(rng.choice([1, 2, 3, 4, 5, 8, 16]))
    pf2 = int(rng.integers(0, 2))
    seg = int(rng.integers(0, 3))
    mp = int(rng.integers(0, 3))
    colb = int(rng.choice([0, 0, 200 << 10, 1 << 20]))
    cap = int(rng.choice([0, 0, 1, 7]))
    one = bool(rng.random() < 0.5)  # DCP_DB_ONE_LAYOUT: the query-lane kernels gather their tile images from the row-sweep tables
    hk.upload_db(profiles, expand_on_host=on_host, one_layout=one)
    hk.upload_seqs(seqs)
    try:
        if one:
            for name, k in (("auto", dcp.KERNEL_AUTO), ("qlane", dcp.KERNEL_QLANE), ("qlane2", dcp.KERNEL_QLANE2)):
                hk.scan(multi, h3, 10.0, kernel=k)
                n, a = hk.scores()
                results[f"{name}[one layout]"] = (n.copy(), a.copy(), hk.hits().copy())
        hk.test_set_rowsweep_variant(stage, waves | (pf2 << 16) | (seg << 24) | (mp << 26))
        hk.test_set_seg_col_bytes(colb)
        hk.scan(multi, h3, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
        n, a = hk.scores()
        results[f"rowsweep[stage={stage} waves={waves} pf2={pf2} seg={seg} mp={mp} colb={colb}]"] = (n.copy(), a.copy(), hk.hits().copy())
        if cap:
            hk.test_set_redo_cap(cap)
            k = dcp.KERNEL_QLANE2 if rng.random() < 0.5 else dcp.KERNEL_QLANE
            hk.scan(multi, h3, 10.0, kernel=k)
            n, a = hk.scores()
            results[f"qlane{'2' if k == dcp.KERNEL_QLANE2 else ''}[redo cap {cap}, same variant]"] = (n.copy(), a.copy(), hk.hits().copy())
        # traceback of the hits and of a few other pairs: the row-sweep kernels' forward pass (the shipped one) against
        # the trace kernel's own loop, step for step, and the score the scan gave (a profile with gains on its delete
        # transitions has no bounded best path: both must then refuse alike)
        hk.test_set_rowsweep_variant(-1, 0)
        hk.test_set_seg_col_bytes(0)
        hk.test_set_redo_cap(0)
        hk.scan(multi, h3, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
        tn, ta = hk.scores()
        cand = [(int(h["seq_idx"]), int(h["profile_idx"])) for h in hk.hits()[:24]]
        fin = np.argwhere(np.isfinite(ta))
        for i in rng.permutation(len(fin))[:8]:
            cand.append((int(fin[i][0]), int(fin[i][1])))
        cand = [(q, p) for q, p in cand if sizes[p] * len(seqs[q]) <= 4_000_000]
        traced = 0
        if cand:
            th = np.array([(q, p, tn[q, p], ta[q, p]) for q, p in cand], dcp.HIT_DTYPE)
            out = {}
            for mode in (0, 1):
                hk.test_set_trace_mode(mode, int(rng.choice([0, 0, 8 << 20])))
                try:
                    out[mode] = hk.trace_paths(th, multi, h3)
                except dcp.DcpError as e:
                    out[mode] = str(e)
            hk.test_set_trace_mode(0, 0)
            if isinstance(out[0], str) or isinstance(out[1], str):
                if out[0] != out[1]:
                    print(f"TRACE: the two forward passes fail differently: {out[0]!r} vs {out[1]!r}\n  {shape}", flush=True)
                    return False, 0
            else:
                for (q, p), a0, a1, s0, s1 in zip(cand, out[0][0], out[1][0], out[0][1], out[1][1]):
                    if not np.array_equal(a0, a1) or not (same_bits(s0, ta[q, p]) and same_bits(s1, ta[q, p])) \
                            or int(a0["seqlen"].sum()) != len(seqs[q]):
                        print(f"TRACE MISMATCH (query {q} len {len(seqs[q])}, profile {p} M {sizes[p]}): {len(a0)} vs {len(a1)} steps, "
                              f"alt {s0!r} / {s1!r} / scan {ta[q, p]!r}\n  {shape}", flush=True)
                        return False, 0
                traced = len(cand)
    finally:
        hk.test_set_rowsweep_variant(-1, 0)
        hk.test_set_seg_col_bytes(0)
        hk.test_set_redo_cap(0)
        hk.test_set_trace_mode(0, 0)

    rn, ra, rh = results["rowsweep"]
    for name, (n, a, h) in results.items():
        if not (same_bits(n, rn) and same_bits(a, ra)):
            bad = np.argwhere((n.view(np.uint32) != rn.view(np.uint32)) | (a.view(np.uint32) != ra.view(np.uint32)))
            q, p = bad[0]
            print(f"MISMATCH {name} vs rowsweep: {len(bad)} pairs, first (query {q} len {len(seqs[q])}, profile {p} M {sizes[p]}): "
                  f"alt {a[q, p]!r} vs {ra[q, p]!r}, null {n[q, p]!r} vs {rn[q, p]!r}\n  {shape}", flush=True)
            return False, 0
        if not np.array_equal(np.sort(h, order=["seq_idx", "profile_idx"]), np.sort(rh, order=["seq_idx", "profile_idx"])):
            print(f"HIT LISTS DIFFER {name} vs rowsweep ({len(h)} vs {len(rh)})\n  {shape}", flush=True)
            return False, 0

    # the oracle on a sample of pairs
    pairs = [(q, p) for q in range(nq) for p in range(nprof)]
    order = rng.permutation(len(pairs))
    tables = {}
    chosen, cells = [], 0
    for i in order:
        q, p = pairs[i]
        c = sizes[p] * len(seqs[q])
        if chosen and cells + c > oracle_cells:
            continue
        chosen.append((q, p))
        cells += c
        if cells > oracle_cells:
            break
    for q, p in chosen:
        if p not in tables:
            em = sc.match_table(p)
            e32 = float(np.float32(eps))
            tables[p] = (profiles[p].trans8, em, dcp.frame_table_host(profiles[p].insert_dist, e32),
                         dcp.frame_table_host(profiles[p].null_dist, e32))
    xts = {len(s): dcp.xtrans(len(s), multi, h3) for s in seqs}

    def score(qp):
        q, p = qp
        t8, em, ei, en = tables[p]
        rc, nl, al = oracle32.dp_tables(t8, em, ei, en, xts[len(seqs[q])], bytes(seqs[q]))
        return q, p, rc, nl, al

    for q, p, rc, nl, al in pool.map(score, chosen):
        if rc != 0 or not (same_bits(nl, rn[q, p]) and same_bits(al, ra[q, p])):
            print(f"ORACLE MISMATCH (query {q} len {len(seqs[q])}, profile {p} M {sizes[p]}): rc {rc}, alt oracle {al!r} device "
                  f"{ra[q, p]!r}, null oracle {nl!r} device {rn[q, p]!r}\n  {shape}", flush=True)
            return False, 0
    # the hit list is the LRT filter over those scores
    lrt = np.float32(-2) * (rn - ra)
    want = {(int(q), int(p)) for q, p in zip(*np.nonzero(np.isfinite(lrt) & ~(lrt < np.float32(10.0))))}
    got = {(int(h["seq_idx"]), int(h["profile_idx"])) for h in rh}
    if got != want:
        print(f"HIT LIST != LRT FILTER ({len(got)} vs {len(want)})\n  {shape}", flush=True)
        return False, 0
    print(f"ok  {shape}; {len(results)} scans agree, {len(chosen)} pairs ({cells / 1e6:.1f} Mcell) == oracle, {len(rh)} hits, {traced} paths traced twice, "
          f"{list(results)[-1] if cap else list(results)[7 if one else 4]}{' one-layout' if one else ''}", flush=True)
    return True, len(chosen)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=420.0)
    ap.add_argument("--rounds", type=int, default=0, help="stop after this many rounds (0: by --seconds)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--oracle-cells", type=float, default=6e7)
    ap.add_argument("--big-every", type=int, default=7, help="every N-th seed draws 513 .. 2311 queries against 1 .. 7 profiles (0: never)")
    ap.add_argument("--threads", type=int, default=min(16, os.cpu_count() or 1))
    a = ap.parse_args()
    dcp = conftest.load_product()
    oracle32 = conftest.Oracle(32)
    sc = dcp.Scanner(0)
    hk = dcp.Scanner(0, lib=dcp.load_testhooks())
    t0 = time.time()
    rounds = pairs = 0
    ok = True
    with ThreadPoolExecutor(a.threads) as pool:
        while ok and (a.rounds == 0 or rounds < a.rounds) and (a.rounds > 0 or time.time() - t0 < a.seconds):
            ok, n = one_round(dcp, oracle32, sc, hk, a.seed + rounds, a.oracle_cells, pool, a.big_every)
            rounds += 1
            pairs += n
    sc.close()
    hk.close()
    print(f"{'PASS' if ok else 'FAIL'}: {rounds} rounds (seeds {a.seed} .. {a.seed + rounds - 1}), {pairs} pairs checked against the "
          f"oracle, {time.time() - t0:.0f} s", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
