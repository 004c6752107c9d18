"""Committed golden vectors (tests/golden/): the oracle still reproduces them (CPU), and the HIP
path reproduces them on the GPU."""
import json
import math
import os

import numpy as np
import pytest

from oracle_py import encode

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "oracle_scores.json")))
REF = json.load(open(os.path.join(HERE, "golden", "reference_protein_profile.json")))


def test_reference_fixture_is_the_reference_test_data():
    assert REF["null_loglik"] == -48.9272687711 and REF["alt_loglik"]["UNIFORM"] == -55.59428153448
    assert REF["alt_loglik"]["OCCUPANCY"] == -54.35543421312 and len(REF["codons"]) == 10


@pytest.mark.parametrize("ci", range(0, len(GOLD["cases"]), 3))
def test_oracle_reproduces_fixture(oracle32, oracle64, ci):
    c = GOLD["cases"][ci]
    for orc, tag, tol in ((oracle32, "f32", 0.0), (oracle64, "f64", 1e-12)):
        p = orc.sample(c["seed"], c["core_size"], c["entry_dist"], c["epsilon"])
        for s, row in zip(GOLD["seqs"], c["scores"]):
            e = encode(s)
            assert p.setup(len(e), c["multi_hits"], c["hmmer3_compat"]) == 0
            rc, nl, al = p.viterbi_fast(e)
            assert rc == 0
            assert math.isclose(nl, row["null_" + tag], rel_tol=tol, abs_tol=0)
            assert math.isclose(al, row["alt_" + tag], rel_tol=tol, abs_tol=0)
            # float32 and float64 chains agree to the reference's float32 bar
            assert math.isclose(row["alt_f32"], row["alt_f64"], rel_tol=5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(True, False), (False, False), (True, True)])
def test_gpu_reproduces_fixture(dcp, mode):
    cases = [c for c in GOLD["cases"] if (c["multi_hits"], c["hmmer3_compat"]) == mode]
    profiles = [dcp.ProteinProfile.sample(c["seed"], c["core_size"], dcp.ProteinCfg(c["entry_dist"], c["epsilon"]))
                for c in cases]
    sc = dcp.Scanner(0)
    sc.upload_db(profiles)
    sc.upload_seqs(GOLD["seqs"])
    sc.scan(mode[0], mode[1], 10.0)
    gn, ga = sc.scores()
    sc.close()
    for p, c in enumerate(cases):
        want_n = np.array([r["null_f64"] for r in c["scores"]])
        want_a = np.array([r["alt_f64"] for r in c["scores"]])
        np.testing.assert_allclose(gn[:, p], want_n, rtol=5e-5)
        np.testing.assert_allclose(ga[:, p], want_a, rtol=5e-5)
        # and far tighter against the float32 chain
        np.testing.assert_allclose(ga[:, p], [r["alt_f32"] for r in c["scores"]], rtol=2e-6)
