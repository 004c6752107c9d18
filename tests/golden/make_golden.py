#!/usr/bin/env python3
"""Generates tests/golden/*.json -- small input/output vectors for the scan path.

The reference itself cannot be built or run here (its DP engine, EBI-Metagenomics/imm v2.0.3,
is a network fetch: /root/reference/CMakeLists.txt:16), so the vectors come from this repo's CPU
oracle (oracle/), which is pinned to the reference's own known-answer tests G1-G3
(test/protein_profile.c:41,65,157; see tests/test_oracle_goldens.py).  G1-G3 themselves are
copied below as data (inputs + expected outputs).

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_py import ENTRY_DIST_OCCUPANCY, ENTRY_DIST_UNIFORM, Oracle, encode  # noqa: E402


def main():
    ref = {
        "source": "/root/reference/test/protein_profile.c",
        "seq": "ATGAAACGCATTAGCACCACCATTACCACCAC",
        "seed": 1, "core_size": 2, "epsilon": 0.1, "multi_hits": True, "hmmer3_compat": False,
        "null_loglik": -48.9272687711, "null_nsteps": 11, "null_first": ["R", 3], "null_last": ["R", 2],
        "alt_loglik": {"UNIFORM": -55.59428153448, "OCCUPANCY": -54.35543421312},
        "alt_nsteps": 14, "alt_first": ["S", 0], "alt_last": ["T", 0],
        "codons": "ATG AAA CGC ATA GCA CCA CCT TAC CAC CAC".split(),
        "rel_tol": {"float32": 5e-05, "float64": 1e-09},
    }
    json.dump(ref, open(os.path.join(HERE, "reference_protein_profile.json"), "w"), indent=1)

    # oracle-generated vectors: sampled profiles (protein_profile_sample seeds) x seeded queries
    o32, o64 = Oracle(32), Oracle(64)
    cases = []
    rng = np.random.default_rng(2024)
    specs = [(1, 2, ENTRY_DIST_UNIFORM, 0.1), (2, 2, ENTRY_DIST_OCCUPANCY, 0.01), (3, 5, ENTRY_DIST_OCCUPANCY, 0.01),
             (4, 37, ENTRY_DIST_UNIFORM, 0.01), (5, 64, ENTRY_DIST_OCCUPANCY, 0.01), (6, 65, ENTRY_DIST_OCCUPANCY, 0.05),
             (7, 130, ENTRY_DIST_OCCUPANCY, 0.01), (8, 200, ENTRY_DIST_OCCUPANCY, 0.01), (9, 257, ENTRY_DIST_OCCUPANCY, 0.01),
             (10, 300, ENTRY_DIST_UNIFORM, 0.01), (11, 520, ENTRY_DIST_OCCUPANCY, 0.01)]
    seqs = ["ATGAAACGCATTAGCACCACCATTACCACCAC", "A", "AC", "ACGTT", "GATTACA"]
    seqs += ["".join("ACGT"[b] for b in rng.integers(0, 4, n)) for n in (11, 33, 64, 150, 301)]
    for multi, h3 in ((True, False), (False, False), (True, True)):
        for seed, M, entry, eps in specs:
            p32, p64 = o32.sample(seed, M, entry, eps), o64.sample(seed, M, entry, eps)
            rows = []
            for s in seqs:
                e = encode(s)
                p32.setup(len(e), multi, h3)
                p64.setup(len(e), multi, h3)
                n32, a32 = p32.viterbi(0, e, False)[1], p32.viterbi(1, e, False)[1]
                n64, a64 = p64.viterbi(0, e, False)[1], p64.viterbi(1, e, False)[1]
                f32 = p32.viterbi_fast(e)
                assert (np.float32(f32[1]), np.float32(f32[2])) == (np.float32(n32), np.float32(a32))
                rows.append({"null_f32": float(n32), "alt_f32": float(a32), "null_f64": n64, "alt_f64": a64})
            cases.append({"seed": seed, "core_size": M, "entry_dist": entry, "epsilon": eps,
                          "multi_hits": multi, "hmmer3_compat": h3, "scores": rows})
    json.dump({"generator": "tests/golden/make_golden.py (CPU oracle, pinned to G1-G3)",
               "seqs": seqs, "cases": cases}, open(os.path.join(HERE, "oracle_scores.json"), "w"))
    print("wrote", len(cases), "cases x", len(seqs), "sequences")


if __name__ == "__main__":
    main()
