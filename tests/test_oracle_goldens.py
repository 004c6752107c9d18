"""Pins the CPU oracle to the reference's own known-answer tests.

Goldens G1-G3 of SURVEY.md §8(c) = /root/reference/test/protein_profile.c:
  :41,:133  null loglik -48.9272687711, 11 steps, step0 (R,3), step10 (R,2)
  :65       alt loglik (ENTRY_DIST_UNIFORM)   -55.59428153448, 14 steps (S,0)..(T,0)
  :157      alt loglik (ENTRY_DIST_OCCUPANCY) -54.35543421312
  :83-102   10 decoded codons
Tolerances are the reference harness's own (test/hope_support.h:26):
5e-5 relative for float32, 1e-9 relative for float64.
"""
import math

import pytest

from oracle_py import (ENTRY_DIST_OCCUPANCY, ENTRY_DIST_UNIFORM, R_STATE, S_STATE,
                       T_STATE, encode)

SEQ = "ATGAAACGCATTAGCACCACCATTACCACCAC"
CODONS = "ATG AAA CGC ATA GCA CCA CCT TAC CAC CAC".split()
NULL_LL = -48.9272687711
ALT_LL = {ENTRY_DIST_UNIFORM: -55.59428153448, ENTRY_DIST_OCCUPANCY: -54.35543421312}


def isclose(a, b, bits):
    return math.isclose(a, b, rel_tol=1e-9 if bits == 64 else 5e-5, abs_tol=0.0)


@pytest.mark.parametrize("bits", [32, 64])
@pytest.mark.parametrize("entry", [ENTRY_DIST_UNIFORM, ENTRY_DIST_OCCUPANCY])
def test_protein_profile_goldens(bits, entry, oracle32, oracle64):
    orc = oracle64 if bits == 64 else oracle32
    prof = orc.sample(1, 2, entry, 0.1)
    seq = encode(SEQ)
    assert prof.setup(0) == 3  # RC_EINVAL, test/protein_profile.c:31
    assert prof.setup(len(seq), True, False) == 0

    rc, ll, path = prof.viterbi(0, seq)
    assert rc == 0 and isclose(ll, NULL_LL, bits)
    assert len(path) == 11
    assert path[0] == (R_STATE, 3) and path[10] == (R_STATE, 2)
    assert orc.state_name(path[0][0]) == "R"

    rc, ll, path = prof.viterbi(1, seq)
    assert rc == 0 and isclose(ll, ALT_LL[entry], bits)
    assert len(path) == 14
    assert path[0] == (S_STATE, 0) and path[13] == (T_STATE, 0)
    assert orc.state_name(path[0][0]) == "S" and orc.state_name(path[13][0]) == "T"

    # protein_codec_next: skip mute steps, slice by seqlen, decode (protein_codec.c:6-24)
    pos, got = 0, []
    for sid, ln in path:
        if ln == 0:
            continue
        got.append(prof.decode(seq[pos:pos + ln], sid)[1])
        pos += ln
    assert pos == len(seq)
    assert got == CODONS


def test_f64_matches_goldens_to_1e11(oracle64):
    """Stronger than the reference's own bar: the f64 chain lands within 1e-11."""
    seq = encode(SEQ)
    p = oracle64.sample(1, 2, ENTRY_DIST_UNIFORM, 0.1)
    p.setup(len(seq))
    assert abs(p.viterbi(0, seq)[1] - NULL_LL) < 1e-10
    assert abs(p.viterbi(1, seq)[1] - ALT_LL[ENTRY_DIST_UNIFORM]) < 1e-10
    p = oracle64.sample(1, 2, ENTRY_DIST_OCCUPANCY, 0.1)
    p.setup(len(seq))
    assert abs(p.viterbi(1, seq)[1] - ALT_LL[ENTRY_DIST_OCCUPANCY]) < 1e-10
