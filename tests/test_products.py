"""hits -> alt paths -> product rows.

CPU part: codon decode, state names, genetic code and row formatting of the product against the
oracle (which reproduces the reference's golden path shape and ten decoded codons,
test/protein_profile.c:67-102).  GPU part: the device traceback returns exactly the steps of the
oracle's generic graph Viterbi (first maximum wins, candidates in the reference's wiring order)."""
import numpy as np
import pytest

from oracle_py import ENTRY_DIST_OCCUPANCY, ENTRY_DIST_UNIFORM, encode

SEQ = "ATGAAACGCATTAGCACCACCATTACCACCAC"
CODONS = "ATG AAA CGC ATA GCA CCA CCT TAC CAC CAC".split()


def test_state_names(dcp, oracle32):
    ids = [(3 << 14) | i for i in range(8)] + [1, 12, 4096, (1 << 14) | 3, (2 << 14) | 77, (2 << 14) | 4096]
    for sid in ids:
        assert dcp.state_name(sid) == oracle32.state_name(sid)
    assert dcp.state_name(12) == "M12" and dcp.state_name((1 << 14) | 3) == "I3"
    assert [dcp.state_name((3 << 14) | i) for i in range(8)] == list("RSNBEJCT")


def test_gc_decode(dcp):
    # NCBI table 1 spot checks (imm_gc_decode(1, codon), used at src/server/protein_match.c:47)
    for codon, aa in (("ATG", "M"), ("TGG", "W"), ("TAA", "*"), ("TAG", "*"), ("TGA", "*"), ("AAA", "K"),
                      ("GCT", "A"), ("CGC", "R"), ("ATA", "I"), ("CAC", "H"), ("TAC", "Y"), ("CCT", "P")):
        assert dcp.gc_decode(codon) == aa


@pytest.mark.parametrize("entry", [ENTRY_DIST_UNIFORM, ENTRY_DIST_OCCUPANCY])
def test_decode_matches_oracle(dcp, oracle32, entry):
    rng = np.random.default_rng(entry)
    prof = dcp.ProteinProfile.sample(3, 9, dcp.ProteinCfg(entry, 0.01))
    op = oracle32.sample(3, 9, entry, 0.01)
    states = [1, 5, 9, (1 << 14) | 2, (3 << 14) | 2, (3 << 14) | 5, (3 << 14) | 6, (3 << 14) | 0]
    n = ties = 0
    for sid in states:
        for ln in range(1, 6):
            for _ in range(12):
                frag = rng.integers(0, 4, ln, dtype=np.uint8)
                best, want = op.decode(bytes(frag), sid)
                got = prof.decode(frag, sid)
                if got != want:
                    # exact ties exist (insert states have a flat codon distribution): the product must
                    # still return a maximiser, up to float32 noise of the oracle's log-domain chain
                    assert abs(op.codon_lprob(bytes(frag), sid, got) - best) <= 2e-6 * abs(best)
                    ties += 1
                n += 1
    assert n == len(states) * 5 * 12 and ties <= n // 50
    for mute in ((3 << 14) | 1, (3 << 14) | 3, (3 << 14) | 4, (3 << 14) | 7, (2 << 14) | 3):
        with pytest.raises(dcp.DcpError):  # assert(!protein_state_is_mute(state_id)) protein_profile.c:310
            prof.decode("ACG", mute)


@pytest.mark.parametrize("entry", [ENTRY_DIST_UNIFORM, ENTRY_DIST_OCCUPANCY])
def test_product_row_of_the_reference_golden(dcp, oracle64, entry):
    """The oracle's path for the reference's test sequence (14 steps, the ten golden codons) written as a
    product row: `frag,state,codon,amino;...` (src/server/protein_match.c:5-19)."""
    op = oracle64.sample(1, 2, entry, 0.1)
    seq = encode(SEQ)
    op.setup(len(seq), True, False)
    _, alt, path = op.viterbi(1, seq)
    _, nul, _ = op.viterbi(0, seq)
    prof = dcp.ProteinProfile.sample(1, 2, dcp.ProteinCfg(entry, 0.1), accession="PF00001")
    steps = np.array([(s, l, 0) for s, l in path], dcp.STEP_DTYPE)
    row = prof.prod_row(SEQ, steps, scan_id=7, seq_id=42, alt_loglik=alt, null_loglik=nul, version="1.2.3")
    assert row.endswith("\n") and row.count("\n") == 1
    f = row[:-1].split("\t")
    assert len(f) == len(dcp.PROD_HEADER.strip().split("\t")) == 9
    assert f[0:4] == ["7", "42", "PF00001", "dna"] and f[6:8] == ["protein", "1.2.3"]
    assert f[4] == "%.17g" % alt and f[5] == "%.17g" % nul  # Fg "%.17g" prod.c:20
    matches = f[8].split(";")
    assert len(matches) == 14
    assert matches[0] == ",S,," and matches[-1] == ",T,,"
    emitting = [m.split(",") for m in matches if m.split(",")[0]]
    assert [m[2] for m in emitting] == CODONS
    assert "".join(m[0] for m in emitting) == SEQ
    assert [m[3] for m in emitting] == [dcp.gc_decode(c) for c in CODONS]
    assert dcp.PROD_HEADER == ("scan_id\tseq_id\tprofile_name\tabc_name\talt_loglik\tnull_loglik\t"
                               "profile_typeid\tversion\tmatch\n")


# ------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def oracle_path(op, seq, multi, h3):
    op.setup(len(seq), multi, h3)
    rc, ll, path = op.viterbi(1, bytes(seq))
    assert rc == 0
    return ll, path


@gpu
def test_trace_reference_golden(dcp):
    for entry, gold in ((ENTRY_DIST_UNIFORM, -55.59428153448), (ENTRY_DIST_OCCUPANCY, -54.35543421312)):
        prof = dcp.ProteinProfile.sample(1, 2, dcp.ProteinCfg(entry, 0.1))
        sc = dcp.Scanner(0)
        sc.upload_db([prof])
        sc.upload_seqs([SEQ])
        sc.scan(True, False, 10.0)
        nl, al = sc.scores()
        hit = np.array([(0, 0, nl[0, 0], al[0, 0])], dcp.HIT_DTYPE)
        paths, alt = sc.trace_paths(hit, True, False)
        npaths, nscore = sc.trace_paths(hit, True, False, null_model=True)
        sc.close()
        # null path: 11 steps, first (R,3), last (R,2)  test/protein_profile.c:43-54
        assert nscore[0] == nl[0, 0] and abs(nscore[0] - (-48.9272687711)) <= 5e-5 * 48.93
        assert [(int(s["state_id"]), int(s["seqlen"])) for s in (npaths[0][0], npaths[0][10])] == \
            [((3 << 14) | 0, 3), ((3 << 14) | 0, 2)] and len(npaths[0]) == 11
        assert alt[0] == al[0, 0] and abs(alt[0] - gold) <= 5e-5 * abs(gold)
        p = paths[0]
        assert len(p) == 14  # EQ(imm_path_nsteps(&prod.path), 14) test/protein_profile.c:67
        assert (p[0]["state_id"], p[0]["seqlen"]) == ((3 << 14) | 1, 0)  # S
        assert (p[13]["state_id"], p[13]["seqlen"]) == ((3 << 14) | 7, 0)  # T
        row = prof.prod_row(SEQ, p)
        got = [m.split(",")[2] for m in row[:-1].split("\t")[8].split(";") if m.split(",")[0]]
        assert got == CODONS  # test/protein_profile.c:83-102


@gpu
@pytest.mark.parametrize("multi,h3", [(True, False), (False, False), (True, True)])
def test_trace_matches_oracle_paths(dcp, oracle32, multi, h3):
    import test_gpu_parity as tp

    rng = np.random.default_rng(11)
    sizes = [2, 9, 40, 64, 65, 130, 200, 300, 520]
    params = [tp.pfam_like_params(rng, M) for M in sizes]
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    oprofs = [oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01) for prm in params]
    seqs = tp.rand_seqs(rng, 6, 1, 120)
    seqs += [tp.planted_query(rng, oprofs[2], sizes[2]), tp.planted_query(rng, oprofs[6], sizes[6]),
             np.concatenate([tp.planted_query(rng, oprofs[4], sizes[4]), tp.planted_query(rng, oprofs[4], sizes[4])])]
    sc = dcp.Scanner(0)
    sc.upload_db(profiles, expand_on_host=True)
    sc.upload_seqs(seqs)
    sc.scan(multi, h3, 10.0)
    nl, al = sc.scores()
    # trace every pair with a finite alt score, hit or not
    pairs = [(q, p) for q in range(len(seqs)) for p in range(len(profiles)) if np.isfinite(al[q, p])]
    hits = np.array([(q, p, nl[q, p], al[q, p]) for q, p in pairs], dcp.HIT_DTYPE)
    paths, alt = sc.trace_paths(hits, multi, h3)
    real_hits = sc.hits()
    sc.close()
    assert len(paths) == len(pairs) > 60
    exact = 0
    for (q, p), path, a in zip(pairs, paths, alt):
        assert a == al[q, p]  # the trace recomputes the scan's score bit for bit
        ll, want = oracle_path(oprofs[p], seqs[q], multi, h3)
        got = [(int(s["state_id"]), int(s["seqlen"])) for s in path]
        assert sum(l for _, l in got) == len(seqs[q])
        assert got[0] == ((3 << 14) | 1, 0) and got[-1] == ((3 << 14) | 7, 0)
        # the oracle scores the DEVICE's path in its own model: it must be a path of the graph and
        # optimal there too (the oracle's own optimum scores itself exactly)
        assert oprofs[p].path_score(1, bytes(seqs[q]), want) == ll
        mine = oprofs[p].path_score(1, bytes(seqs[q]), got)
        assert np.isfinite(mine) and abs(mine - ll) <= 2e-6 * abs(ll)
        assert abs(mine - a) <= 2e-6 * abs(a)
        # identical steps, except where two alignments tie to the last float32 bits (the oracle builds
        # its tables in float32 log domain, the device in float64 probability domain)
        exact += got == want
    assert exact >= len(pairs) - max(1, len(pairs) // 50)
    # product rows of the real hits: the fragments tile the query, match states carry codons
    hp, _ = sc2_rows(dcp, profiles, seqs, real_hits, multi, h3)
    assert hp == len(real_hits) >= 3


@gpu
@pytest.mark.parametrize("multi,h3", [(True, False), (False, False), (True, True)])
def test_trace_forward_by_the_row_sweep_equals_the_trace_kernels_own(dcp, oracle32, multi, h3):
    """dcp_gpu_trace_paths fills the hits' work areas with the row-sweep kernel of each profile's size class (round 4:
    viterbi_rowsweep_kernel<R, W, 0, false, TRACE>, the scan's own rows) and walks back through them; the trace kernel's
    own forward loop of rounds 1-3 is kept in the tests' build as a second implementation.  Both must give the same
    steps and the same score for every pair: every size class (one to eight nodes per lane, four to sixteen
    wavefronts per pair, the two classes whose profiles share table rows), a flagged (a few positive MD / DD) profile,
    planted one- and two-domain queries, queries of 1 .. 9 nt (shorter than the five-row look-back), and a work budget
    that cuts the hits into several rounds of launches; the order of the caller's hits is kept."""
    import test_gpu_parity as tp

    rng = np.random.default_rng(404 + int(multi) + 2 * int(h3))
    sizes = [1, 3, 40, 64, 65, 128, 129, 192, 250, 300, 380, 448, 512, 520, 768, 900, 1100, 1536, 2049, 3000, 4096]
    params = [tp.pfam_like_params(rng, M) for M in sizes]
    # a flagged profile (finite MD / DD > 0 on a few nodes: the delete states enter E(j); with gains on every node a
    # path collects score per delete and has no bounded optimum -- neither implementation traces that)
    null, match, trans = tp.pfam_like_params(rng, 90)
    trans = trans.copy()
    trans[30:34, 2] = np.float32(0.05)
    trans[31:34, 6] = np.float32(0.02)
    params.append((null, match, trans))
    sizes.append(90)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    oprofs = {p: oracle32.new(*params[p], ENTRY_DIST_OCCUPANCY, 0.01) for p in (2, 6, 9, 13)}
    seqs = [rng.integers(0, 4, L, dtype=np.uint8) for L in (1, 2, 3, 4, 5, 6, 9)] + tp.rand_seqs(rng, 5, 20, 400)
    seqs += [tp.planted_query(rng, oprofs[2], sizes[2], flank=5), tp.planted_query(rng, oprofs[6], sizes[6], flank=40),
             tp.planted_query(rng, oprofs[13], sizes[13], flank=12),
             np.concatenate([tp.planted_query(rng, oprofs[9], sizes[9]), tp.planted_query(rng, oprofs[9], sizes[9])])]
    sc = dcp.Scanner(0, lib=dcp.load_testhooks())
    try:
        sc.upload_db(profiles, expand_on_host=True)
        sc.upload_seqs(seqs)
        sc.scan(multi, h3, 10.0)
        nl, al = sc.scores()
        pairs = [(q, p) for q in range(len(seqs)) for p in range(len(profiles)) if np.isfinite(al[q, p])]
        order = rng.permutation(len(pairs))  # not in class order, not in sequence order
        pairs = [pairs[i] for i in order]
        hits = np.array([(q, p, nl[q, p], al[q, p]) for q, p in pairs], dcp.HIT_DTYPE)
        assert len(hits) > 300
        sc.test_set_trace_mode(0, 0)
        new_paths, new_alt = sc.trace_paths(hits, multi, h3)
        sc.test_set_trace_mode(0, 40 << 20)  # 160 MB of work area per round: the 4 096-node pairs take one round each
        cut_paths, cut_alt = sc.trace_paths(hits, multi, h3)
        sc.test_set_trace_mode(1, 0)
        old_paths, old_alt = sc.trace_paths(hits, multi, h3)
    finally:
        sc.test_set_trace_mode(0, 0)
        sc.close()
    for (q, p), a, b, c_, x, y, z in zip(pairs, new_paths, cut_paths, old_paths, new_alt, cut_alt, old_alt):
        assert x == al[q, p] and y == al[q, p] and z == al[q, p], (q, p)
        assert np.array_equal(a, c_) and np.array_equal(b, c_), (q, p, sizes[p], len(seqs[q]))
        assert int(a["seqlen"].sum()) == len(seqs[q])
    # and against the oracle's own traceback for the planted pairs
    for q, p in ((12, 2), (13, 6), (14, 13), (15, 9)):
        i = pairs.index((q, p))
        ll, want = oracle_path(oprofs[p], seqs[q], multi, h3)
        got = [(int(s_["state_id"]), int(s_["seqlen"])) for s_ in new_paths[i]]
        mine = oprofs[p].path_score(1, bytes(seqs[q]), got)
        assert np.isfinite(mine) and abs(mine - ll) <= 2e-6 * abs(ll)


def sc2_rows(dcp, profiles, seqs, hits, multi, h3):
    sc = dcp.Scanner(0)
    sc.upload_db(profiles, expand_on_host=True)
    sc.upload_seqs(seqs)
    paths, _ = sc.trace_paths(hits, multi, h3)
    sc.close()
    n = 0
    for h, path in zip(hits, paths):
        q, p = int(h["seq_idx"]), int(h["profile_idx"])
        row = profiles[p].prod_row(seqs[q], path, scan_id=1, seq_id=q, alt_loglik=h["alt_loglik"],
                                   null_loglik=h["null_loglik"])
        f = row[:-1].split("\t")
        ms = [m.split(",") for m in f[8].split(";")]
        assert all(len(m) == 4 for m in ms)
        assert "".join(m[0] for m in ms) == "".join("ACGT"[b] for b in seqs[q])
        for frag, state, codon, amino in ms:
            if frag:
                assert len(codon) == 3 and amino == dcp.gc_decode(codon)
            else:
                assert codon == "" and amino == "" and state[0] in "SBETD"
        assert sum(1 for m in ms if m[1].startswith("M")) >= 10
        n += 1
    return n, None
