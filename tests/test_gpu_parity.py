"""Parity of the HIP scan path against the CPU oracle (runs on a real MI355X).

Two levels:
  * DP level -- the oracle's float32 recursion is fed the PRODUCT's own tables
    (transitions, emission tables, special transitions) and must match the
    kernel BIT FOR BIT: every candidate is (pred + trans) + emis in IEEE float32
    and only max combines them.
  * end to end -- oracle builds everything itself (imm-style float32 log-domain
    model build): scores agree within 5e-5 relative, the reference's own float32
    bar (test/hope_support.h:26); tighter in practice.
"""
import numpy as np
import pytest

from oracle_py import ENTRY_DIST_OCCUPANCY, ENTRY_DIST_UNIFORM, encode

pytestmark = pytest.mark.gpu

REL = 5e-5


@pytest.fixture(scope="module")
def scanner(dcp):
    s = dcp.Scanner(0)
    yield s
    s.close()


@pytest.fixture(scope="module")
def hooks_scanner(dcp):
    """A context of the tests' own -DDCP_TEST_HOOKS build of the library (libdcp_hip_testhooks.so): the
    shipped library does not export dcp_gpu_test_set_redo_cap."""
    s = dcp.Scanner(0, lib=dcp.load_testhooks())
    yield s
    s.close()


@pytest.fixture(params=["rowsweep", "qlane", "qlane2"])
def kern(request, dcp):
    """Every parity test runs on every kernel: the row sweep (one wavefront group per pair), the
    query-lane throughput kernel and its two-stage variant (both forced even for tiny batches)."""
    return {"rowsweep": dcp.KERNEL_ROWSWEEP, "qlane": dcp.KERNEL_QLANE, "qlane2": dcp.KERNEL_QLANE2}[request.param]


def rand_seqs(rng, n, lo, hi):
    return [rng.integers(0, 4, int(rng.integers(lo, hi + 1)), dtype=np.uint8) for _ in range(n)]


def oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, multi, h3, expand_on_host):
    """null/alt [nseq, nprof] from orc_dp_tables fed the tables the device holds."""
    nl = np.zeros((len(seqs), len(profiles)), np.float32)
    al = np.zeros_like(nl)
    for p, prof in enumerate(profiles):
        em = scanner.match_table(p)  # what the kernel reads
        eps = prof_eps[id(prof)]
        ei = dcp.frame_table_host(prof.insert_dist, eps)
        en = dcp.frame_table_host(prof.null_dist, eps)
        if expand_on_host:  # device copy must be the host table exactly
            for k in (0, prof.core_size - 1):
                assert np.array_equal(em[:, k], dcp.frame_table_host(prof.match_dist[k], eps))
        for q, s in enumerate(seqs):
            xt = dcp.xtrans(len(s), multi, h3)
            rc, a, b = oracle32.dp_tables(prof.trans8, em, ei, en, xt, bytes(s))
            assert rc == 0
            nl[q, p], al[q, p] = a, b
    return nl, al


prof_eps = {}


def make_profiles(dcp, specs):
    out = []
    for seed, M, entry, eps in specs:
        p = dcp.ProteinProfile.sample(seed, M, dcp.ProteinCfg(entry, eps))
        prof_eps[id(p)] = float(np.float32(eps))
        out.append(p)
    return out


def same_bits(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("multi,h3", [(True, False), (False, False), (True, True)])
def test_dp_bit_exact_small(dcp, oracle32, scanner, multi, h3, kern):
    rng = np.random.default_rng(1)
    specs = [(1, 2, ENTRY_DIST_UNIFORM, 0.1), (2, 2, ENTRY_DIST_OCCUPANCY, 0.01),
             (3, 7, ENTRY_DIST_OCCUPANCY, 0.01), (4, 63, ENTRY_DIST_OCCUPANCY, 0.01),
             (5, 64, ENTRY_DIST_UNIFORM, 0.01), (6, 65, ENTRY_DIST_OCCUPANCY, 0.01),
             (7, 128, ENTRY_DIST_OCCUPANCY, 0.01), (8, 129, ENTRY_DIST_OCCUPANCY, 0.05),
             (9, 192, ENTRY_DIST_OCCUPANCY, 0.01), (10, 200, ENTRY_DIST_OCCUPANCY, 0.01),
             (11, 256, ENTRY_DIST_OCCUPANCY, 0.01)]
    profiles = make_profiles(dcp, specs)
    seqs = [np.array(list(encode("ATGAAACGCATTAGCACCACCATTACCACCAC")), np.uint8)]
    seqs += [rng.integers(0, 4, L, dtype=np.uint8) for L in (1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 14, 15, 16, 17, 33, 64, 100, 301)]
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    scanner.scan(multi, h3, 10.0, kernel=kern)
    gn, ga = scanner.scores()
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, multi, h3, True)
    assert same_bits(gn, on)
    assert same_bits(ga, oa)
    assert np.isfinite(ga).all() and np.isfinite(gn).all()


def test_device_expansion_matches_host(dcp, scanner):
    profiles = make_profiles(dcp, [(21, 5, ENTRY_DIST_OCCUPANCY, 0.01), (22, 70, ENTRY_DIST_UNIFORM, 0.1),
                                   (23, 256, ENTRY_DIST_OCCUPANCY, 0.01)])
    scanner.upload_db(profiles, expand_on_host=False)
    for p, prof in enumerate(profiles):
        em = scanner.match_table(p)
        eps = prof_eps[id(prof)]
        host = np.stack([dcp.frame_table_host(prof.match_dist[k], eps) for k in range(prof.core_size)], 1)
        assert np.isfinite(em).all()
        np.testing.assert_allclose(em, host, rtol=3e-7, atol=1e-6)


def test_end_to_end_vs_oracle(dcp, oracle32, scanner, kern):
    """Device-expanded tables, oracle's independent float32 chain, incl. generic graph Viterbi."""
    rng = np.random.default_rng(2)
    specs = [(31, 2, ENTRY_DIST_UNIFORM, 0.1), (32, 17, ENTRY_DIST_OCCUPANCY, 0.01),
             (33, 100, ENTRY_DIST_OCCUPANCY, 0.01), (34, 180, ENTRY_DIST_OCCUPANCY, 0.01),
             (35, 256, ENTRY_DIST_UNIFORM, 0.01)]
    profiles = make_profiles(dcp, specs)
    seqs = rand_seqs(rng, 12, 20, 400)
    scanner.upload_db(profiles)
    scanner.upload_seqs(seqs)
    scanner.scan(True, False, 10.0, kernel=kern)
    gn, ga = scanner.scores()
    oprofs = [oracle32.sample(s, M, e, eps) for s, M, e, eps in specs]
    hits, on, oa = oracle32.scan(oprofs, [bytes(s) for s in seqs], True, False, 10.0, nthreads=4, mode=1)
    np.testing.assert_allclose(gn, on, rtol=REL, atol=0)
    np.testing.assert_allclose(ga, oa, rtol=REL, atol=0)
    # generic imm-style graph Viterbi on a subset
    for p in (0, 2):
        for q in (0, 5):
            oprofs[p].setup(len(seqs[q]), True, False)
            assert abs(oprofs[p].viterbi(1, bytes(seqs[q]), want_path=False)[1] - ga[q, p]) <= REL * abs(ga[q, p])
            assert abs(oprofs[p].viterbi(0, bytes(seqs[q]), want_path=False)[1] - gn[q, p]) <= REL * abs(gn[q, p])


def test_reference_goldens_through_the_gpu(dcp, scanner, kern):
    """test/protein_profile.c:41,65,157 on the HIP path, at the reference's float32 tolerance."""
    seq = "ATGAAACGCATTAGCACCACCATTACCACCAC"
    for entry, gold in ((ENTRY_DIST_UNIFORM, -55.59428153448), (ENTRY_DIST_OCCUPANCY, -54.35543421312)):
        prof = dcp.ProteinProfile.sample(1, 2, dcp.ProteinCfg(entry, 0.1))
        for host in (True, False):
            scanner.upload_db([prof], expand_on_host=host)
            scanner.upload_seqs([seq])
            scanner.scan(True, False, 10.0, kernel=kern)
            nl, al = scanner.scores()
            assert abs(nl[0, 0] - (-48.9272687711)) <= REL * 48.93
            assert abs(al[0, 0] - gold) <= REL * abs(gold)
            assert len(scanner.hits()) == 0  # lrt < 10


def test_rejects_bad_sequences(dcp, scanner):
    prof = dcp.ProteinProfile.sample(1, 2)
    scanner.upload_db([prof])
    with pytest.raises(dcp.DcpError) as e:  # protein_profile_setup(L=0) -> RC_EINVAL
        scanner.upload_seqs([np.zeros(0, np.uint8)])
    assert e.value.rc == dcp.RC_EINVAL
    with pytest.raises(dcp.DcpError) as e:  # symbol outside the alphabet
        scanner.upload_seqs(["ACGTNACGT"])
    assert e.value.rc == dcp.RC_EINVAL
    with pytest.raises(dcp.DcpError):
        scanner.upload_seqs([])


def pfam_like_params(rng, M):
    """Peaked match distributions and Pfam-like transitions (MM ~ 0.95), so that a query
    emitted from the consensus is a real hit."""
    def norm(x):
        return x - np.logaddexp.reduce(x, axis=-1, keepdims=True)

    null = norm(np.log(rng.random(20) + 0.5)).astype(np.float32)
    match = np.log(rng.random((M, 20)) * 0.02 + 1e-3)
    match[np.arange(M), rng.integers(0, 20, M)] = np.log(0.8)
    match = norm(match).astype(np.float32)
    trans = np.tile(np.log(np.array([0.95, 0.025, 0.025, 0.6, 0.4, 0.6, 0.4])), (M + 1, 1))
    trans[0, 6] = -np.inf
    trans[M, 2] = trans[M, 6] = -np.inf
    trans[:, 0:3] = norm(trans[:, 0:3])
    trans[:, 3:5] = norm(trans[:, 3:5])
    with np.errstate(invalid="ignore"):
        dm = norm(trans[:, 5:7])
    trans[:, 5:7] = np.where(np.isnan(dm), trans[:, 5:7], dm)
    return null, match, trans.astype(np.float32)


def planted_query(rng, oprof, prof_len, flank=30):
    """A query carrying the most likely codon of each match state: a real hit."""
    body = []
    for k in range(prof_len):
        best, arg = -np.inf, None
        for c in range(64):
            frag = bytes([(c >> 4) & 3, (c >> 2) & 3, c & 3])
            lp, _ = oprof.decode(frag, (0 << 14) | (k + 1))
            if lp > best:
                best, arg = lp, frag
        body.append(arg)
    core = np.frombuffer(b"".join(body), np.uint8)
    return np.concatenate([rng.integers(0, 4, flank, dtype=np.uint8), core,
                           rng.integers(0, 4, flank, dtype=np.uint8)])


def test_hits_and_lrt_filter(dcp, oracle32, scanner, kern):
    """Planted hits pass the LRT filter exactly where the oracle says (scan_thread.c:121-123)."""
    rng = np.random.default_rng(3)
    sizes = [40, 53, 66, 79, 92, 105]
    params = [pfam_like_params(rng, M) for M in sizes]
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    for p in profiles:
        prof_eps[id(p)] = cfg.epsilon
    oprofs = [oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01) for prm in params]
    seqs = rand_seqs(rng, 10, 100, 300)
    seqs[2] = planted_query(rng, oprofs[1], sizes[1])
    seqs[7] = planted_query(rng, oprofs[4], sizes[4])
    # multi-hit: two copies of the same domain in one query
    seqs[9] = np.concatenate([planted_query(rng, oprofs[3], sizes[3]), planted_query(rng, oprofs[3], sizes[3])])
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    for multi in (True, False):
        scanner.scan(multi, False, 10.0, kernel=kern)
        gn, ga = scanner.scores()
        # the query-lane kernel hands pairs with E -> B / J -> B feedback (the two-domain query
        # at least) to the row sweep; uni-hit scans and row-sweep scans have no redo pairs
        redo = scanner.last_scan_redo_pairs
        assert (redo >= 1) if (multi and kern in (dcp.KERNEL_QLANE, dcp.KERNEL_QLANE2)) else (redo == 0)
        assert redo < len(seqs) * len(profiles)
        on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, multi, False, True)
        assert same_bits(gn, on) and same_bits(ga, oa)
        lrt = np.float32(-2) * (on - oa)
        want = sorted((q, p) for q in range(len(seqs)) for p in range(len(profiles))
                      if np.isfinite(lrt[q, p]) and lrt[q, p] >= 10.0)
        hits = scanner.hits()
        got = [(int(h["seq_idx"]), int(h["profile_idx"])) for h in hits]
        assert got == want
        assert (2, 1) in got and (7, 4) in got and (9, 3) in got
        for h in hits:
            assert h["null_loglik"] == on[h["seq_idx"], h["profile_idx"]]
            assert h["alt_loglik"] == oa[h["seq_idx"], h["profile_idx"]]
        # end to end against the oracle's own model build as well
        _, en, ea = oracle32.scan(oprofs, [bytes(s) for s in seqs], multi, False, 10.0, 4, 1)
        np.testing.assert_allclose(ga, ea, rtol=REL)
        np.testing.assert_allclose(gn, en, rtol=REL)
    # the two-domain query scores higher with multi-hit than without
    scanner.scan(True, False, 10.0, kernel=kern)
    a_multi = scanner.scores()[1][9, 3]
    scanner.scan(False, False, 10.0, kernel=kern)
    a_uni = scanner.scores()[1][9, 3]
    assert a_multi > a_uni


def test_explicit_special_transitions(dcp, oracle32, scanner, kern):
    """imm_dp_viterbi scores with the special transitions the profile holds NOW, whatever they are
    (dcp_gpu_seqs_set_xtrans): the LOG1 = 0 defaults of a profile that never saw
    protein_profile_setup (protein_model.c:322-340; test/protein_db.c:73 runs exactly that), a setup
    of another length, uni-hit values although the scan flag says multi-hit, arbitrary values.
    Bit-exact against the oracle's recursion fed the same 13 numbers."""
    rng = np.random.default_rng(91)
    profiles = make_profiles(dcp, [(301, 2, ENTRY_DIST_OCCUPANCY, 0.01), (302, 77, ENTRY_DIST_OCCUPANCY, 0.01),
                                   (303, 300, ENTRY_DIST_UNIFORM, 0.05)])
    seqs = rand_seqs(rng, 7, 20, 260)
    xt = np.zeros((len(seqs), 13), np.float32)            # row 0: never set up
    xt[1] = dcp.xtrans(5000, True, False)                  # stale: some other length
    xt[2] = dcp.xtrans(len(seqs[2]), False, False)         # uni-hit numbers
    xt[3] = dcp.xtrans(len(seqs[3]), True, True)           # hmmer3_compat
    xt[4] = -rng.random(13).astype(np.float32) * 3         # arbitrary finite
    xt[5] = dcp.xtrans(len(seqs[5]), True, False)
    xt[5, 9] = 0.0                                         # E -> B for free: every pair re-enters the core
    xt[6] = dcp.xtrans(len(seqs[6]), True, False)
    xt[6, 4] = -np.inf                                     # N -> B closed: only S -> B at row 0 enters
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    scanner.set_xtrans(xt)
    for multi_flag in (True, False):  # the flags are ignored once explicit transitions are in force
        scanner.scan(multi_flag, False, 10.0, kernel=kern)
        gn, ga = scanner.scores()
        for p, prof in enumerate(profiles):
            em = scanner.match_table(p)
            eps = prof_eps[id(prof)]
            ei = dcp.frame_table_host(prof.insert_dist, eps)
            en = dcp.frame_table_host(prof.null_dist, eps)
            for q, sq in enumerate(seqs):
                rc, a, b = oracle32.dp_tables(prof.trans8, em, ei, en, xt[q], bytes(sq))
                assert rc == 0
                assert same_bits(gn[q, p], a) and same_bits(ga[q, p], b), (q, p, gn[q, p], a, ga[q, p], b)
    # a new upload returns to length-derived transitions
    scanner.upload_seqs(seqs)
    scanner.scan(True, False, 10.0, kernel=kern)
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, True)
    gn, ga = scanner.scores()
    assert same_bits(gn, on) and same_bits(ga, oa)
    # NaN is refused, a wrong count too
    bad = xt.copy()
    bad[0, 0] = np.nan
    with pytest.raises(dcp.DcpError):
        scanner.set_xtrans(bad)
    with pytest.raises(dcp.DcpError):
        scanner.set_xtrans(xt[:3])


def test_redo_list_overflow_falls_back_to_row_sweep(dcp, oracle32, hooks_scanner):
    """More feedback pairs than a redo list holds: the scan is repeated by the row-sweep kernel
    and stays bit-exact (the test-only setter shrinks the lists to one pair per size class)."""
    scanner = hooks_scanner
    rng = np.random.default_rng(33)
    M = 60
    prm = pfam_like_params(rng, M)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg)]
    prof_eps[id(profiles[0])] = cfg.epsilon
    oprof = oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01)
    two = lambda: np.concatenate([planted_query(rng, oprof, M), planted_query(rng, oprof, M)])
    seqs = [two(), two(), two()] + rand_seqs(rng, 5, 100, 300)
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, True)
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE)
    assert scanner.last_scan_redo_pairs >= 3
    gn, ga = scanner.scores()
    assert same_bits(gn, on) and same_bits(ga, oa)
    scanner.test_set_redo_cap(1)
    try:
        scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE)
        gn, ga = scanner.scores()
        assert same_bits(gn, on) and same_bits(ga, oa)
        got = sorted((int(h["seq_idx"]), int(h["profile_idx"])) for h in scanner.hits())
        assert got[:3] == [(0, 0), (1, 0), (2, 0)]
    finally:
        scanner.test_set_redo_cap(0)


@pytest.mark.parametrize("M", [300, 400, 450, 600, 1300, 2600])
def test_redo_pairs_of_multi_wavefront_classes(dcp, oracle32, scanner, M):
    """A real hit against a big profile leaves the query-lane kernel through the redo list of a many-nodes-
    per-lane (R = 5, 7, 8) or multi-wavefront (W = 4, 8, 16) size class: the row sweep's pair mode must
    score it, and every other pair of the scan, bit-exactly."""
    rng = np.random.default_rng(M + 1)
    prm = pfam_like_params(rng, M)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    small = pfam_like_params(rng, 37)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg), dcp.ProteinProfile.from_params(*small, cfg)]
    for p in profiles:
        prof_eps[id(p)] = cfg.epsilon
    oprof = oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01)
    seqs = rand_seqs(rng, 6, 50, 400)
    seqs[1] = planted_query(rng, oprof, M, flank=12)
    seqs[4] = planted_query(rng, oracle32.new(*small, ENTRY_DIST_OCCUPANCY, 0.01), 37, flank=40)
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE)
    assert scanner.last_scan_redo_pairs >= 2
    assert scanner.last_scan_launches >= 3  # query lane + one redo launch per populated size class
    gn, ga = scanner.scores()
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, True)
    assert same_bits(gn, on) and same_bits(ga, oa)
    got = {(int(h["seq_idx"]), int(h["profile_idx"])) for h in scanner.hits()}
    assert {(1, 0), (4, 1)} <= got


def test_forced_kernels_give_the_automatic_choice_s_bits(dcp, scanner):
    """Whatever kernel = 0 picks (a cost model: tests/test_zz_kernel_choice.py), forcing each kernel gives the same
    bits; launch infos tell a query-lane scan (one W = 0 launch carrying all cells + redo launches carrying none)
    from a row-sweep scan."""
    rng = np.random.default_rng(48)
    profiles = make_profiles(dcp, [(900 + i, int(m), ENTRY_DIST_OCCUPANCY, 0.01) for i, m in enumerate((3, 70, 130, 300))])
    seqs = rand_seqs(rng, 60, 20, 120)
    scanner.upload_db(profiles)
    scanner.upload_seqs(seqs)
    with pytest.raises(dcp.DcpError):
        dcp.Scanner(0).last_scan_redo_pairs  # no scan yet
    scanner.scan(True, False, 10.0)
    n_auto, a_auto = scanner.scores()
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
    assert all(li["W"] >= 1 for li in scanner.launch_infos())       # row-sweep launches only
    assert scanner.last_scan_redo_pairs == 0 and scanner.last_scan_kernel == dcp.KERNEL_ROWSWEEP
    n_rs, a_rs = scanner.scores()
    assert same_bits(n_auto, n_rs) and same_bits(a_auto, a_rs)
    for k in (dcp.KERNEL_QLANE, dcp.KERNEL_QLANE2):
        scanner.scan(True, False, 10.0, kernel=k)
        infos = scanner.launch_infos()
        assert infos[0]["W"] == 0 and infos[0]["cells"] == sum(p.core_size for p in profiles) * sum(len(s) for s in seqs)
        assert all(li["cells"] == 0 for li in infos[1:])                 # redo launches carry no cells of their own
        n_ql, a_ql = scanner.scores()
        assert same_bits(n_auto, n_ql) and same_bits(a_auto, a_ql)


def test_new_batch_of_equal_count_is_not_scanned_with_the_old_one_s_layout(dcp, scanner):
    """The query-lane kernel keeps a per-range length order and transposed word planes; a new upload with
    the same number of sequences (other lengths, other bases) must rebuild them -- regression test for a
    stale cache that read the previous batch's planes."""
    rng = np.random.default_rng(606)
    profiles = make_profiles(dcp, [(700 + i, int(m), ENTRY_DIST_OCCUPANCY, 0.01) for i, m in enumerate((9, 40, 77))])
    scanner.upload_db(profiles)
    a = rand_seqs(rng, 12, 20, 60)
    b = rand_seqs(rng, 12, 200, 900)  # longer: the old planes would be too small
    scanner.upload_seqs(a)
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE)
    for batch in (b, a, b):
        scanner.upload_seqs(batch)
        scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE)
        nq, aq = scanner.scores()
        scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
        nr, ar = scanner.scores()
        assert same_bits(nq, nr) and same_bits(aq, ar)
    with pytest.raises(dcp.DcpError):
        scanner.upload_seqs([np.array([0, 1, 7, 2], np.uint8)] * 12)  # a failed upload keeps nothing half-built
    scanner.upload_seqs(a)
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE)
    n2, a2 = scanner.scores()
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
    n3, a3 = scanner.scores()
    assert same_bits(n2, n3) and same_bits(a2, a3)


def test_smallest_profiles(dcp, oracle32, scanner, kern):
    """core_size 1, 2 and 3 (protein_model_setup accepts 1..4096, protein_model.c:157-160): no I state at
    all for one node, D_1 without an incoming edge; queries of 1 to 20 nt."""
    rng = np.random.default_rng(123)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = []
    for M in (1, 2, 3, 1):
        p = dcp.ProteinProfile.from_params(*pfam_like_params(rng, M), cfg)
        prof_eps[id(p)] = cfg.epsilon
        profiles.append(p)
    seqs = [rng.integers(0, 4, L, dtype=np.uint8) for L in (1, 2, 3, 4, 5, 6, 7, 11, 20)]
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    for multi in (True, False):
        scanner.scan(multi, False, 10.0, kernel=kern)
        gn, ga = scanner.scores()
        on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, multi, False, True)
        assert same_bits(gn, on) and same_bits(ga, oa)


def test_random_api_sequences(dcp, scanner):
    """State-machine stress: random interleavings of DB uploads, sequence uploads (sometimes invalid),
    ranged / full scans, flag changes and kernel choices.  After every scan the other kernel must reproduce
    the scores bit for bit on the same range, and the hit lists must match."""
    import os
    rng = np.random.default_rng(int(os.environ.get("DCP_STRESS_SEED", "20261004")))
    nsteps = int(os.environ.get("DCP_STRESS_STEPS", "40"))  # longer offline runs: DCP_STRESS_STEPS=400
    cfg_specs = lambda n: [(int(rng.integers(1, 1 << 20)), int(rng.integers(2, 400)), ENTRY_DIST_OCCUPANCY, 0.01) for _ in range(n)]
    profiles = make_profiles(dcp, cfg_specs(5))
    scanner.upload_db(profiles)
    seqs = rand_seqs(rng, 7, 5, 300)
    scanner.upload_seqs(seqs)
    nscans = 0
    for step in range(nsteps):
        op = rng.integers(0, 10)
        if op == 0:
            profiles = make_profiles(dcp, cfg_specs(int(rng.integers(1, 9))))
            scanner.upload_db(profiles)
        elif op in (1, 2):
            n = len(seqs) if op == 2 else int(rng.integers(1, 70))   # op 2: same count, new content
            seqs = rand_seqs(rng, n, 1, int(rng.choice([40, 300, 1500])))
            scanner.upload_seqs(seqs)
        elif op == 3:
            with pytest.raises(dcp.DcpError):
                scanner.upload_seqs([np.array([0, 9], np.uint8)])
        else:
            n = len(seqs)
            lo = int(rng.integers(0, n))
            hi = int(rng.integers(lo + 1, n + 1))
            q_range = None if rng.random() < 0.4 else (lo, hi)
            multi, h3 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            first = [dcp.KERNEL_QLANE, dcp.KERNEL_ROWSWEEP, dcp.KERNEL_AUTO, dcp.KERNEL_QLANE2][int(rng.integers(0, 4))]
            other = dcp.KERNEL_ROWSWEEP if first in (dcp.KERNEL_QLANE, dcp.KERNEL_QLANE2) else \
                [dcp.KERNEL_QLANE, dcp.KERNEL_QLANE2][int(rng.integers(0, 2))]
            sl = slice(0, n) if q_range is None else slice(lo, hi)
            res = []
            for k in (first, other):
                scanner.scan(multi, h3, 10.0, kernel=k, q_range=q_range)
                nn, aa = scanner.scores()
                res.append((nn[sl].copy(), aa[sl].copy(), scanner.hits()))
            assert same_bits(res[0][0], res[1][0]) and same_bits(res[0][1], res[1][1]), (step, q_range, multi, h3)
            assert np.array_equal(res[0][2], res[1][2])
            nscans += 1
    assert nscans >= nsteps // 3


def test_scan_is_idempotent_and_order_free(dcp, scanner, kern):
    """Size-independent properties: same scores on re-scan, under profile permutation and
    when the batch is split (pairs are independent)."""
    rng = np.random.default_rng(4)
    specs = [(51 + i, int(m), ENTRY_DIST_OCCUPANCY, 0.01) for i, m in enumerate(rng.integers(2, 257, 24))]
    profiles = make_profiles(dcp, specs)
    seqs = rand_seqs(rng, 40, 30, 250)
    scanner.upload_db(profiles)
    scanner.upload_seqs(seqs)
    scanner.scan(kernel=kern)
    n1, a1 = scanner.scores()
    scanner.scan(kernel=kern)
    n2, a2 = scanner.scores()
    assert same_bits(n1, n2) and same_bits(a1, a2)
    perm = rng.permutation(len(profiles))
    scanner.upload_db([profiles[i] for i in perm])
    scanner.scan(kernel=kern)
    n3, a3 = scanner.scores()
    assert same_bits(n3, n1[:, perm]) and same_bits(a3, a1[:, perm])
    scanner.upload_seqs(seqs[10:25])
    scanner.scan(kernel=kern)
    n4, a4 = scanner.scores()
    assert same_bits(n4, n1[10:25][:, perm]) and same_bits(a4, a1[10:25][:, perm])


def delete_heavy_params(rng, M):
    """Transitions that favour long delete runs (MD, DD large): the in-row delete chain then
    carries across many lanes and wavefronts, the worst case for the fixed-point iteration."""
    null, match, trans = pfam_like_params(rng, M)
    t = np.tile(np.log(np.array([0.3, 0.05, 0.65, 0.5, 0.5, 0.08, 0.92])), (M + 1, 1))
    t[0, 6] = -np.inf
    t[M, 2] = t[M, 6] = -np.inf
    return null, match, t.astype(np.float32)


@pytest.mark.parametrize("sizes", [(257, 300, 384), (320, 321, 448, 449), (385, 512, 513), (700, 768, 1024),
                                   (1025, 1536, 2048), (2049, 3072, 4096)])
def test_dp_bit_exact_large_profiles(dcp, oracle32, scanner, sizes, kern):
    """core_size > 256: the single-wavefront classes with 5..8 nodes per lane (up to 512, both sides of every
    class boundary) and the multi-wavefront ones, incl. the maximum core size 4096
    (PROTEIN_MODEL_CORE_SIZE_MAX, limits.h:11)."""
    rng = np.random.default_rng(sum(sizes))
    profiles = make_profiles(dcp, [(100 + i, M, ENTRY_DIST_OCCUPANCY, 0.01) for i, M in enumerate(sizes)])
    seqs = [rng.integers(0, 4, L, dtype=np.uint8) for L in (1, 4, 5, 6, 23, 60)]
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    scanner.scan(True, False, 10.0, kernel=kern)
    gn, ga = scanner.scores()
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, True)
    assert same_bits(gn, on)
    assert same_bits(ga, oa)


@pytest.mark.parametrize("M", [40, 200, 256, 300, 420, 500, 1000, 2500])
def test_dp_bit_exact_delete_heavy(dcp, oracle32, scanner, M, kern):
    rng = np.random.default_rng(M)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    prm = delete_heavy_params(rng, M)
    prof = dcp.ProteinProfile.from_params(*prm, cfg)
    prof_eps[id(prof)] = cfg.epsilon
    oprof = oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01)
    # a query made of the first and last few consensus codons: the best path must delete
    # most of the profile in one run
    head = planted_query(rng, oprof, 6, flank=0)
    tail_codons = []
    for k in range(M - 6, M):
        best, arg = -np.inf, None
        for c in range(64):
            frag = bytes([(c >> 4) & 3, (c >> 2) & 3, c & 3])
            lp, _ = oprof.decode(frag, k + 1)
            if lp > best:
                best, arg = lp, frag
        tail_codons.append(arg)
    jump = np.concatenate([head, np.frombuffer(b"".join(tail_codons), np.uint8)])
    seqs = [jump, rng.integers(0, 4, 50, dtype=np.uint8), rng.integers(0, 4, 7, dtype=np.uint8)]
    scanner.upload_db([prof], expand_on_host=True)
    scanner.upload_seqs(seqs)
    for multi in (True, False):
        scanner.scan(multi, False, 10.0, kernel=kern)
        gn, ga = scanner.scores()
        on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, [prof], seqs, multi, False, True)
        assert same_bits(gn, on)
        assert same_bits(ga, oa)
    if M == 40:
        # uni-hit: the generic graph Viterbi really crosses the profile in one delete run
        oprof.setup(len(jump), False, False)
        _, _, path = oprof.viterbi(1, bytes(jump))
        ndel = sum(1 for sid, _ in path if (sid >> 14) == 2)
        assert ndel >= 20


@pytest.mark.parametrize("multi", [True, False])
def test_positive_delete_transitions_keep_d_in_e(dcp, oracle32, scanner, kern, multi):
    """The query-lane kernels take E(j) as the maximum over the MATCH states, which is exact when MD, DD <= 0
    (log-probabilities).  A profile whose transitions are not -- a positive MD or DD lets a delete state
    beat every match state of its row -- is flagged at upload and scored by the row sweep through the redo
    lists, uni-hit scans included.  Mixed with ordinary profiles, all kernels stay bit-exact."""
    rng = np.random.default_rng(77 + int(multi))
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    params = []
    for i, M in enumerate((9, 40, 70, 130, 300)):
        null, match, trans = pfam_like_params(rng, M)
        if i % 2 == 0:  # not a probability model: the delete path GAINS score
            trans = trans.copy()
            trans[1:M, 2] = np.float32(0.7)   # MD
            trans[1:M, 6] = np.float32(0.4)   # DD
        params.append((null, match, trans))
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    for pr in profiles:
        prof_eps[id(pr)] = cfg.epsilon
    seqs = rand_seqs(rng, 70, 20, 260)
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    scanner.scan(multi, False, 10.0, kernel=kern)
    gn, ga = scanner.scores()
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, multi, False, True)
    assert same_bits(gn, on) and same_bits(ga, oa)
    if kern != dcp.KERNEL_ROWSWEEP:
        # every pair of the three flagged profiles went through the redo lists
        assert scanner.last_scan_redo_pairs >= 3 * len(seqs)
    # the delete states really decide E for the flagged profiles: dropping them would change the score
    assert np.isfinite(oa).all()


@pytest.mark.parametrize("stage,waves,prefetch2", [(0, 4, 0), (20, 1, 0), (20, 3, 0), (20, 4, 0), (20, 16, 0), (84, 1, 0),
                                                  (84, 5, 0), (84, 8, 0), (84, 16, 0), (20, 1, 1), (20, 3, 1), (20, 16, 1),
                                                  (84, 1, 1), (84, 5, 1), (84, 16, 1)])
def test_rowsweep_variants_bit_exact(dcp, oracle32, hooks_scanner, stage, waves, prefetch2):
    """Every grid-mode variant of the row sweep -- rows of the emission table a block keeps in LDS (none, the
    one- and two-base words, the three-base words as well) x wavefronts per block (one profile per block: blocks
    with spare wavefronts, a last block that is not full, more chunks than one block takes) x global rows fetched one
    or two DP rows ahead (sequences of 1..150 nt: every length of the ten-row unrolling's tail) -- against the
    oracle's float32 recursion on the product's tables, bit for bit, over every one-wavefront size class, a
    flagged (positive MD / DD) profile and multi-wavefront classes (which have one variant).  Forced through the
    tests' own -DDCP_TEST_HOOKS build; the shipped library picks among the same kernels by batch size.  (Since round 4
    the two smallest classes run K profiles per wavefront at these batch sizes, whatever variant is forced --
    test_profiles_sharing_a_wavefront_bit_exact -- and their one-profile kernels, which read a member's columns of the
    shared table, are forced in test_one_profile_kernels_on_shared_tables.)"""
    rng = np.random.default_rng(4242)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    sizes = (1, 5, 64, 65, 127, 128, 150, 192, 250, 256, 257, 320, 384, 448, 449, 512, 600)
    params = [pfam_like_params(rng, M) for M in sizes]
    null, match, trans = pfam_like_params(rng, 90)
    trans = trans.copy()
    trans[1:90, 2] = np.float32(0.7)  # MD > 0: the delete states decide E(j)
    trans[1:90, 6] = np.float32(0.4)
    params.append((null, match, trans))
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    for pr in profiles:
        prof_eps[id(pr)] = cfg.epsilon
    hooks_scanner.upload_db(profiles, expand_on_host=True)
    try:
        hooks_scanner.test_set_rowsweep_variant(stage, waves | (prefetch2 << 16))
        for nseq in (1, 3, 21):
            # 21 sequences: lengths 1..21 (every tail of the five- and ten-row unrollings); else random up to 150 nt
            seqs = ([rng.integers(0, 4, L, dtype=np.uint8) for L in range(1, 22)] if nseq == 21
                    else rand_seqs(rng, nseq, 1, 150))
            hooks_scanner.upload_seqs(seqs)
            for multi in (True, False):
                hooks_scanner.scan(multi, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
                gn, ga = hooks_scanner.scores()
                on, oa = oracle_dp_on_product_tables(dcp, oracle32, hooks_scanner, profiles, seqs, multi, False, True)
                assert same_bits(gn, on) and same_bits(ga, oa), (stage, waves, prefetch2, nseq, multi)
    finally:
        hooks_scanner.test_set_rowsweep_variant(-1, 0)


@pytest.mark.parametrize("nsmall", [1, 2, 3, 4, 5, 9])
def test_profiles_sharing_a_wavefront_bit_exact(dcp, oracle32, scanner, nsmall):
    """Profiles of at most 128 nodes in grid mode (viterbi_mp_kernel): four (at most 64 nodes) or two (65 .. 128) of them
    share one wavefront -- 16 / 32 lanes of four nodes each, their tables side by side in one table, E(j) a per-part
    maximum, insert / background emissions per lane -- and are scored against the same query together.  Group sizes
    that do not fill a wavefront (1, 2, 3, 5, 9 profiles of a class: absent members' lanes run on -inf), core sizes at
    both ends of each class and 1 .. 5 nodes, a flagged (positive MD / DD) profile in each class (never grouped: the
    one-profile kernel on its own columns), planted hits, 1 .. 21 queries incl. every tail of the five-row unrolling:
    bit for bit against the oracle's float32 recursion on the product's tables, multi- and uni-hit; the traceback
    and the table read-back work on a member's column view."""
    rng = np.random.default_rng(700 + nsmall)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    small = [1, 2, 3, 5, 17, 33, 47, 63, 64][:nsmall] if nsmall < 9 else [1, 2, 3, 5, 17, 33, 47, 63, 64]
    mid = [65, 66, 90, 127, 128][:max(1, min(5, nsmall))]
    sizes = small + mid + [300]
    params = [pfam_like_params(rng, M) for M in sizes]
    for M in (40, 100):  # one flagged profile per class
        null, match, trans = pfam_like_params(rng, M)
        trans = trans.copy()
        trans[1:M, 2] = np.float32(0.7)
        trans[1:M, 6] = np.float32(0.4)
        params.append((null, match, trans))
        sizes.append(M)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    oprofs = [oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01) for prm in params]
    for pr in profiles:
        prof_eps[id(pr)] = cfg.epsilon
    scanner.upload_db(profiles, expand_on_host=True)
    hit_p = len(small) - 1                  # the largest of the small ones
    hit_q = len(small)                      # 65 nodes
    for nseq in (1, 6, 21):
        seqs = ([rng.integers(0, 4, L, dtype=np.uint8) for L in range(1, 22)] if nseq == 21 else rand_seqs(rng, nseq, 1, 300))
        if nseq == 6:
            seqs[1] = planted_query(rng, oprofs[hit_p], sizes[hit_p], flank=7)
            seqs[4] = planted_query(rng, oprofs[hit_q], sizes[hit_q], flank=11)
        scanner.upload_seqs(seqs)
        for multi in (True, False):
            on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, multi, False, True)
            scanner.scan(multi, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
            gn, ga = scanner.scores()
            assert same_bits(gn, on) and same_bits(ga, oa), (nsmall, nseq, multi)
            hits = scanner.hits()
            lrt = np.float32(-2) * (on - oa)
            want = {(int(q), int(p)) for q, p in zip(*np.nonzero(np.isfinite(lrt) & ~(lrt < np.float32(10.0))))}
            assert {(int(h["seq_idx"]), int(h["profile_idx"])) for h in hits} == want
            if nseq == 6 and sizes[hit_p] >= 17:
                assert (1, hit_p) in want and (4, hit_q) in want
                # the traceback reads a member's columns of the shared table
                sel = np.array([h for h in hits if (int(h["seq_idx"]), int(h["profile_idx"])) in ((1, hit_p), (4, hit_q))], dcp.HIT_DTYPE)
                paths, alt = scanner.trace_paths(sel, multi, False)
                for h, path, a in zip(sel, paths, alt):
                    assert len(path) > 0 and np.float32(a) == h["alt_loglik"]
                    assert int(path["seqlen"].sum()) == len(seqs[int(h["seq_idx"])])
    # the query-lane kernels' redo pairs of these classes go through the one-profile kernel on the same views
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE2)
    qn, qa = scanner.scores()
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
    rn, ra = scanner.scores()
    assert same_bits(qn, rn) and same_bits(qa, ra)


@pytest.mark.parametrize("on_host", [False, True])
def test_one_table_layout_bit_exact(dcp, oracle32, scanner, on_host):
    """DCP_DB_ONE_LAYOUT (include/dcp_gpu.h): only the row-sweep tables [1364][ldk] are resident and the query-lane
    kernels gather each 8-node tile's LDS image from them (stage_tile_image<G, true>) instead of copying a stored
    image.  Core sizes at both ends of every tile (8 k - 1, 8 k, 8 k + 1), of the row classes (63 / 64 / 65 R nodes: a
    row with no, one, several padding columns), profiles sharing their rows four / two at a time (column views with a
    row length that is not their own), a flagged profile, one of several wavefronts; 1 .. 300 queries so that the
    three query-lane kernels all run (64-query wavefronts, one stage, two stages): every kernel gives the bits of
    the two-layout DB and of the oracle's float32 recursion on the product's tables, and the same hit list; the
    tables take less than 0.6 of the two layouts' bytes."""
    rng = np.random.default_rng(4100 + int(on_host))
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    sizes = [1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 33, 63, 64, 65, 71, 72, 73, 127, 128, 129, 191, 192, 193, 255, 256, 257,
             319, 320, 383, 384, 385, 512, 513, 700]
    params = [pfam_like_params(rng, M) for M in sizes]
    null, match, trans = pfam_like_params(rng, 44)  # flagged: positive MD / DD (never grouped, pairs go to the row sweep)
    trans = trans.copy()
    trans[1:44, 2] = np.float32(0.7)
    trans[1:44, 6] = np.float32(0.4)
    params.append((null, match, trans))
    sizes.append(44)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    oprofs = {p: oracle32.new(*params[p], ENTRY_DIST_OCCUPANCY, 0.01) for p in (11, 20, 30)}
    for pr in profiles:
        prof_eps[id(pr)] = cfg.epsilon
    for nseq in (1, 70, 300):
        seqs = rand_seqs(rng, nseq, 1, 260)
        seqs[0] = planted_query(rng, oprofs[11], sizes[11], flank=9)
        if nseq > 1:
            seqs[5] = planted_query(rng, oprofs[20], sizes[20], flank=3)
            seqs[nseq - 1] = planted_query(rng, oprofs[30], sizes[30], flank=12)
        results = {}
        for one in (False, True):
            scanner.upload_db(profiles, expand_on_host=on_host, one_layout=one)
            assert scanner.one_layout == one
            scanner.upload_seqs(seqs)
            for name, k in (("auto", dcp.KERNEL_AUTO), ("rowsweep", dcp.KERNEL_ROWSWEEP), ("qlane", dcp.KERNEL_QLANE),
                            ("qlane2", dcp.KERNEL_QLANE2)):
                scanner.scan(True, False, 10.0, kernel=k)
                n, a = scanner.scores()
                results[(one, name)] = (n.copy(), a.copy(), np.sort(scanner.hits(), order=["seq_idx", "profile_idx"]))
            if one:
                one_bytes = scanner.table_bytes
            else:
                two_bytes = scanner.table_bytes  # both layouts are there by now: the row sweep has run
        assert one_bytes < 0.6 * two_bytes, (one_bytes, two_bytes)
        # the oracle on the product's tables (read back from the one-layout DB, which is the resident one now)
        on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, on_host)
        rn, ra, rh = results[(False, "rowsweep")]
        assert same_bits(rn, on) and same_bits(ra, oa), nseq
        assert {(0, 11)} <= {(int(h["seq_idx"]), int(h["profile_idx"])) for h in rh}
        for key, (n, a, h) in results.items():
            assert same_bits(n, rn) and same_bits(a, ra), (key, nseq)
            assert np.array_equal(h, rh), (key, nseq)
    # the traceback works on the same tables (the flagged profile's gains on its delete transitions leave it no bounded
    # best path: its hits are refused alike on either DB)
    sel = rh[rh["profile_idx"] != len(sizes) - 1][:4]
    assert len(sel) > 0
    paths, alt = scanner.trace_paths(sel, True, False)
    for h, path, a in zip(sel, paths, alt):
        assert len(path) > 0 and np.float32(a) == h["alt_loglik"]


@pytest.mark.parametrize("stage,waves,prefetch2", [(0, 4, 0), (20, 4, 0), (84, 8, 0), (20, 16, 1)])
def test_one_profile_kernels_on_shared_tables(dcp, oracle32, hooks_scanner, stage, waves, prefetch2):
    """The one-profile kernels of the two smallest classes (what the 65 .. 128-node class runs from 96 queries on) read
    a profile's COLUMNS of a table it shares with its neighbours: the staged image is a strided copy, a lane past the
    last node reads the padding behind the profile's own columns.  Forced through the tests' build (bits 26..27: never K
    profiles per wavefront) in every staging variant, bit for bit against the oracle."""
    rng = np.random.default_rng(31337)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    sizes = (1, 4, 31, 64, 65, 99, 121, 125, 128, 64, 3, 128, 70)
    params = [pfam_like_params(rng, M) for M in sizes]
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    for pr in profiles:
        prof_eps[id(pr)] = cfg.epsilon
    hooks_scanner.upload_db(profiles, expand_on_host=True)
    try:
        hooks_scanner.test_set_rowsweep_variant(stage, waves | (prefetch2 << 16) | (1 << 26))
        for nseq in (2, 21):
            seqs = ([rng.integers(0, 4, L, dtype=np.uint8) for L in range(1, 22)] if nseq == 21 else rand_seqs(rng, nseq, 30, 200))
            hooks_scanner.upload_seqs(seqs)
            hooks_scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
            gn, ga = hooks_scanner.scores()
            on, oa = oracle_dp_on_product_tables(dcp, oracle32, hooks_scanner, profiles, seqs, True, False, True)
            assert same_bits(gn, on) and same_bits(ga, oa), (stage, waves, prefetch2, nseq)
    finally:
        hooks_scanner.test_set_rowsweep_variant(-1, 0)


@pytest.mark.parametrize("multi", [True, False])
def test_segmented_sweep_bit_exact(dcp, oracle32, hooks_scanner, multi):
    """Profiles of more than 512 nodes in grid mode: one wavefront per pair, the profile cut into segments of 384 nodes
    (the R = 3 multi-wavefront classes) or 512 nodes (the R = 4 ones), ONE SEGMENT PER LAUNCH with the pair's boundary
    column parked in HBM between launches (viterbi_segment_kernel), B(j) = N(j) + NB, and the pairs whose E -> B /
    J -> B feedback beat that B -- planted hits here, in a 384- and in a 512-node-segment class -- finished by the exact
    multi-wavefront kernel behind it.  Forced on for every batch size through the test-hooks build (the library uses it
    once a class has 4 096 pairs), against the oracle's float32 recursion on the product's tables, bit for bit; every multi-
    wavefront class at both ends of its range (two to eight segments; profiles of one class with different segment
    counts side by side); a flagged (positive MD / DD) profile goes to the exact kernel whole; with a column budget of a
    few hundred KB a class's queries are swept chunk by chunk."""
    rng = np.random.default_rng(9090 + int(multi))
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    sizes = (513, 600, 767, 768, 769, 900, 1024, 1025, 1100, 1536, 1537, 1700, 2048, 2049, 2500, 3072, 3073, 3600, 4096, 300)
    params = [pfam_like_params(rng, M) for M in sizes]
    null, match, trans = pfam_like_params(rng, 640)
    trans = trans.copy()
    trans[1:640, 2] = np.float32(0.7)
    trans[1:640, 6] = np.float32(0.4)
    params.append((null, match, trans))
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    oprofs = [oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01) for prm in params]
    for pr in profiles:
        prof_eps[id(pr)] = cfg.epsilon
    hooks_scanner.upload_db(profiles, expand_on_host=True)
    try:
        for nseq, col_bytes in ((1, 0), (5, 0), (37, 0), (37, 300 << 10)):
            seqs = rand_seqs(rng, nseq, 1, 260)
            if nseq > 1:  # homologous queries: the multi-hit feedback path
                seqs[0] = planted_query(rng, oprofs[1], sizes[1], flank=15)   # 600 nodes: two 384-node segments
                seqs[-1] = planted_query(rng, oprofs[8], sizes[8], flank=9)   # 1100 nodes: three
                seqs[2] = planted_query(rng, oprofs[5], sizes[5], flank=11)   # 900 nodes: two 512-node segments
            hooks_scanner.upload_seqs(seqs)
            hooks_scanner.test_set_seg_col_bytes(col_bytes)
            hooks_scanner.test_set_rowsweep_variant(20, 4 | (2 << 24))  # segmented sweep: always
            hooks_scanner.scan(multi, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
            gn, ga = hooks_scanner.scores()
            hits = hooks_scanner.hits()
            on, oa = oracle_dp_on_product_tables(dcp, oracle32, hooks_scanner, profiles, seqs, multi, False, True)
            assert same_bits(gn, on) and same_bits(ga, oa), (nseq, multi, col_bytes)
            hooks_scanner.test_set_rowsweep_variant(20, 4 | (1 << 24))  # the same batch without it
            hooks_scanner.scan(multi, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
            en, ea = hooks_scanner.scores()
            assert same_bits(gn, en) and same_bits(ga, ea)
            assert np.array_equal(hits, hooks_scanner.hits())
            if nseq > 1:
                got = {(int(h["seq_idx"]), int(h["profile_idx"])) for h in hits}
                assert {(0, 1), (nseq - 1, 8), (2, 5)} <= got
    finally:
        hooks_scanner.test_set_rowsweep_variant(-1, 0)
        hooks_scanner.test_set_seg_col_bytes(0)


def test_qlane_at_block_scale(dcp, oracle32, scanner):
    """The throughput kernel with every lane in use: 700 queries (2 full 256-query blocks + a partial
    one, lengths 1..400 so the length sort matters) x 45 profiles of mixed sizes (more tasks than a
    few blocks take in one go).  Both device kernels must agree bit for bit on all 31 500 pairs,
    the hit lists must be identical, and a sample of pairs is checked against the oracle."""
    rng = np.random.default_rng(2025)
    sizes = [int(m) for m in rng.integers(2, 330, 40)] + [600, 1100, 8, 9, 16]
    params = [pfam_like_params(rng, M) for M in sizes]
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    oprofs = [oracle32.new(*prm, ENTRY_DIST_OCCUPANCY, 0.01) for prm in params]
    seqs = rand_seqs(rng, 700, 1, 400)
    for q, p in ((5, 3), (300, 17), (699, 40), (256, 0)):  # planted hits, also at block boundaries
        seqs[q] = planted_query(rng, oprofs[p], sizes[p], flank=12)
    seqs[511] = np.concatenate([planted_query(rng, oprofs[7], sizes[7], 5), planted_query(rng, oprofs[7], sizes[7], 5)])
    scanner.upload_db(profiles)
    scanner.upload_seqs(seqs)
    out = {}
    for name, k in (("rowsweep", dcp.KERNEL_ROWSWEEP), ("qlane", dcp.KERNEL_QLANE), ("qlane2", dcp.KERNEL_QLANE2),
                    ("auto", dcp.KERNEL_AUTO)):
        scanner.scan(True, False, 10.0, kernel=k)
        out[name] = scanner.scores() + (scanner.hits(),)
    for name in ("qlane", "auto"):
        assert same_bits(out[name][0], out["rowsweep"][0])
        assert same_bits(out[name][1], out["rowsweep"][1])
        assert np.array_equal(out[name][2], out["rowsweep"][2])
    hits = out["qlane"][2]
    got = {(int(h["seq_idx"]), int(h["profile_idx"])) for h in hits}
    assert {(5, 3), (300, 17), (699, 40), (256, 0), (511, 7)} <= got
    gn, ga = out["qlane"][:2]
    assert np.isfinite(gn).all() and np.isfinite(ga).all()
    # oracle (own model build) on a sample of pairs incl. every hit
    sample = list(got) + [(int(rng.integers(0, 700)), int(rng.integers(0, len(sizes)))) for _ in range(60)]
    for q, p in sample:
        rc, nl, al = (oprofs[p].setup(len(seqs[q]), True, False), *oprofs[p].viterbi_fast(bytes(seqs[q]))[1:])
        assert rc == 0
        assert abs(gn[q, p] - nl) <= REL * abs(nl) and abs(ga[q, p] - al) <= REL * abs(al)
    # a sub-range scan keeps batch indices and equals the full scan there
    scanner.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE, q_range=(200, 600))
    n2, a2 = scanner.scores()
    assert same_bits(n2[200:600], gn[200:600]) and same_bits(a2[200:600], ga[200:600])
    h2 = scanner.hits()
    assert {(int(h["seq_idx"]), int(h["profile_idx"])) for h in h2} == {x for x in got if 200 <= x[0] < 600}


def test_long_sequences(dcp, oracle32, scanner, kern):
    """Queries up to 10 kbp (BASELINE config C5's upper end) and one far beyond: offsets, scratch planes
    and the sequence look-ahead hold; scores equal the oracle's bit for bit on the product's tables."""
    rng = np.random.default_rng(77)
    profiles = make_profiles(dcp, [(201, 50, ENTRY_DIST_OCCUPANCY, 0.01), (202, 300, ENTRY_DIST_OCCUPANCY, 0.01),
                                   (203, 700, ENTRY_DIST_UNIFORM, 0.01)])
    seqs = [rng.integers(0, 4, L, dtype=np.uint8) for L in (4095, 7000, 10000, 15, 16, 17, 31, 32, 33, 25000)]
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    scanner.scan(True, False, 10.0, kernel=kern)
    gn, ga = scanner.scores()
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, True)
    assert same_bits(gn, on)
    assert same_bits(ga, oa)
    assert np.isfinite(ga).all()


def test_more_than_65536_profiles(dcp, oracle32, scanner, kern):
    """Profile indices beyond 16 bits (the reader allows 2^20 profiles per DB, src/db/reader.c): three distinct
    profiles repeated 22 000 times; every copy's scores equal the oracle's for its original."""
    rng = np.random.default_rng(65536)
    base = make_profiles(dcp, [(401, 2, ENTRY_DIST_OCCUPANCY, 0.01), (402, 70, ENTRY_DIST_OCCUPANCY, 0.01),
                               (403, 300, ENTRY_DIST_UNIFORM, 0.01)])
    seqs = rand_seqs(rng, 5, 20, 90)
    scanner.upload_db(base, expand_on_host=True)
    scanner.upload_seqs(seqs)
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, base, seqs, True, False, True)
    reps = 22000
    scanner.upload_db(base * reps)
    scanner.upload_seqs(seqs)
    scanner.scan(True, False, 10.0, kernel=kern)
    gn, ga = scanner.scores()
    assert gn.shape == (5, 3 * reps)
    assert same_bits(gn, np.tile(on, (1, reps))) and same_bits(ga, np.tile(oa, (1, reps)))
    hits = scanner.hits()
    want = {(q, p) for q in range(5) for p in range(3) if np.isfinite(-2 * (on[q, p] - oa[q, p])) and not (np.float32(-2) * (on[q, p] - oa[q, p]) < 10.0)}
    assert {(int(h["seq_idx"]), int(h["profile_idx"]) % 3) for h in hits} == want
    assert len(hits) == len(want) * reps


def test_more_than_65536_queries(dcp, oracle32, scanner, kern):
    """Sequence indices beyond 16 bits in one resident batch: 70 000 short queries (1..40 nt), three profiles."""
    rng = np.random.default_rng(70000)
    profiles = make_profiles(dcp, [(411, 5, ENTRY_DIST_OCCUPANCY, 0.01), (412, 70, ENTRY_DIST_OCCUPANCY, 0.01),
                                   (413, 260, ENTRY_DIST_UNIFORM, 0.01)])
    seqs = rand_seqs(rng, 70000, 1, 40)
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    scanner.scan(True, False, 10.0, kernel=kern)
    gn, ga = scanner.scores()
    # the oracle on a sample that includes both ends of the batch and the 16-bit boundary
    idx = sorted(set([0, 1, 65535, 65536, 65537, 69999] + [int(i) for i in rng.integers(0, 70000, 600)]))
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, [seqs[i] for i in idx], True, False, True)
    assert same_bits(gn[idx], on) and same_bits(ga[idx], oa)
    # ranged scan of the tail: same bits, hit records carry batch-relative sequence indices
    scanner.scan(True, False, 10.0, kernel=kern, q_range=(65000, 70000))
    n2, a2 = scanner.scores()
    assert same_bits(n2[65000:], gn[65000:]) and same_bits(a2[65000:], ga[65000:])
    assert all(65000 <= int(h["seq_idx"]) < 70000 for h in scanner.hits())


def test_sequence_near_the_scheduler_limit(dcp, oracle32, scanner):
    """The reference accepts sequences up to SCHED_SEQ_SIZE = 1 MiB (src/server/scan.c:227-229 reads them into a
    buffer of that size).  A 300 000-nt query: the automatic choice (the row sweep -- the query-lane kernels would
    keep one lane busy for minutes; asserted in tests/test_zz_kernel_choice.py), and every kernel forced: the
    query-lane kernels hold too -- 32-bit row offsets into scratch
    planes of 300 MB, fewer resident blocks.  All equal the oracle bit for bit."""
    rng = np.random.default_rng(1 << 20)
    profiles = make_profiles(dcp, [(301, 40, ENTRY_DIST_OCCUPANCY, 0.01), (302, 120, ENTRY_DIST_OCCUPANCY, 0.01)])
    seqs = [rng.integers(0, 4, L, dtype=np.uint8) for L in (300000, 33, 2000)]
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, True)
    for k in (dcp.KERNEL_AUTO, dcp.KERNEL_ROWSWEEP, dcp.KERNEL_QLANE, dcp.KERNEL_QLANE2):
        scanner.scan(True, False, 10.0, kernel=k)
        gn, ga = scanner.scores()
        assert same_bits(gn, on) and same_bits(ga, oa), k
    assert np.isfinite(ga).all()
    # the limit itself (1 MiB - 1 bases), automatic choice only
    seqs = [rng.integers(0, 4, (1 << 20) - 1, dtype=np.uint8), rng.integers(0, 4, 77, dtype=np.uint8)]
    scanner.upload_seqs(seqs)
    on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, True, False, True)
    scanner.scan(True, False, 10.0)
    gn, ga = scanner.scores()
    assert same_bits(gn, on) and same_bits(ga, oa)


def test_full_size_c3_step_both_kernels_agree(dcp, oracle32, c3_profiles, bench_mod):
    """BASELINE.json's headline size: one bench step = 20 000 profiles (sum M = 3.57e6) x 1 000 queries of
    1 000 nt = 2e7 pairs, 3.6e12 cells.  Too big for the oracle, so a size-independent property: the two
    independent device implementations (row sweep, query lane + redo) agree bit for bit on every one of the
    2e7 null and alt scores and on the hit list, and a few sampled pairs are checked against the oracle."""
    bench = bench_mod
    sizes, profiles = c3_profiles
    queries = bench.make_queries(0, 1000, 1000)
    sc = dcp.Scanner(0)
    try:
        sc.upload_db(profiles)
        sc.upload_seqs_flat(queries.reshape(-1), (np.arange(1001, dtype=np.uint64) * 1000).astype(np.uint32))
        out = {}
        for name, k in (("qlane", dcp.KERNEL_QLANE), ("qlane2", dcp.KERNEL_QLANE2), ("rowsweep", dcp.KERNEL_ROWSWEEP)):
            sc.scan(True, False, 10.0, kernel=k)
            n, a = sc.scores()
            out[name] = (n.view(np.uint32).copy(), a.view(np.uint32).copy(), sc.hits())
            if name == "qlane":
                redo = sc.last_scan_redo_pairs
        assert 0 < redo < 0.05 * 2e7
        # (what kernel = 0 picks on this DB at which batch size is a tuning matter: tests/test_zz_kernel_choice.py)
        sc.scan(True, False, 10.0, kernel=dcp.KERNEL_AUTO)
        n, a = sc.scores()
        out["auto"] = (n.view(np.uint32).copy(), a.view(np.uint32).copy(), sc.hits())
        for other in ("qlane2", "rowsweep", "auto"):
            assert np.array_equal(out["qlane"][0], out[other][0]), other
            assert np.array_equal(out["qlane"][1], out[other][1]), other
            assert np.array_equal(out["qlane"][2], out[other][2]), other
        assert np.isfinite(out["qlane"][1].view(np.float32)).all()
        rng = np.random.default_rng(7)
        alt = out["qlane"][1].view(np.float32)
        nul = out["qlane"][0].view(np.float32)
        for _ in range(6):
            q, p = int(rng.integers(0, 1000)), int(rng.integers(0, 20000))
            op = oracle32.sample(0xDEC1F0 + p, int(sizes[p]))
            assert op.setup(1000, True, False) == 0
            _, on, oa = op.viterbi_fast(bytes(queries[q]))
            assert abs(nul[q, p] - on) <= REL * abs(on) and abs(alt[q, p] - oa) <= REL * abs(oa)
    finally:
        sc.close()


def test_packed_slots_skewed_lengths_bit_exact(dcp, oracle32, scanner):
    """Dynamic batching (plan_query_groups, dcp_gpu.hip): with skewed lengths a wavefront slot of a query-lane block
    sweeps SEVERAL 64-query groups one after the other per tile -- plane rows offset by the groups before it, the
    two-stage kernel's ring counting plane rows across groups, null / alt scores parked in the planes' spare rows
    between the first and the last tile.  700 queries of 1 .. 3 000 nt (11 groups in one block of four slots: one slot
    holds the longest group, the others two to five), profiles of 1, 2, 3 and 12 tiles (odd and even tile counts: either
    stage sweeps the last tile), planted hits in a short and in a long group: every score and the hit list equal the
    oracle's float32 recursion on the product's tables bit for bit, on both query-lane kernels; multi- and uni-hit."""
    import ctypes as C
    rng = np.random.default_rng(4242)
    sizes = (5, 13, 23, 92)
    params = [pfam_like_params(rng, M) for M in sizes]
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = [dcp.ProteinProfile.from_params(*prm, cfg) for prm in params]
    for p in profiles:
        prof_eps[id(p)] = cfg.epsilon
    lens = np.round(np.exp(rng.uniform(np.log(1.0), np.log(3000.0), 700))).astype(int)
    seqs = [rng.integers(0, 4, int(n), dtype=np.uint8) for n in lens]
    order = np.argsort(lens, kind="stable")
    short_q, long_q = int(order[100]), int(order[690])
    seqs[short_q] = planted_query(rng, oracle32.new(*params[1], ENTRY_DIST_OCCUPANCY, 0.01), sizes[1], flank=3)
    seqs[long_q] = np.concatenate([rng.integers(0, 4, 1500, dtype=np.uint8),
                                   planted_query(rng, oracle32.new(*params[3], ENTRY_DIST_OCCUPANCY, 0.01), sizes[3], flank=700)])
    # the plan this batch gets: one block, some slot with several groups
    ls = np.sort(np.array([len(s) for s in seqs], np.uint32))
    nb, cost = C.c_uint(0), C.c_ulonglong(0)
    f = dcp.lib.dcp_plan_query_slots
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong), C.c_void_p, C.c_void_p,
                  C.c_uint, C.c_void_p, C.c_uint]
    sf = np.zeros(64, np.uint32)
    assert f(ls.ctypes.data, len(ls), 4, C.byref(nb), C.byref(cost), None, None, 0, sf.ctypes.data, 64) == 0
    assert nb.value == 1 and np.diff(sf[:5].astype(int)).max() >= 3 and np.diff(sf[:5].astype(int)).min() >= 1
    scanner.upload_db(profiles, expand_on_host=True)
    scanner.upload_seqs(seqs)
    for multi in (True, False):
        on, oa = oracle_dp_on_product_tables(dcp, oracle32, scanner, profiles, seqs, multi, False, True)
        lrt = np.float32(-2) * (on - oa)  # xmath_lrt_f32, scan_thread.c:121-123
        want_hits = {(int(q), int(p)) for q, p in zip(*np.nonzero(np.isfinite(lrt) & ~(lrt < np.float32(10.0))))}
        for k in (dcp.KERNEL_QLANE, dcp.KERNEL_QLANE2):
            scanner.scan(multi, False, 10.0, kernel=k)
            gn, ga = scanner.scores()
            assert same_bits(gn, on) and same_bits(ga, oa), (multi, k)
            got = {(int(h["seq_idx"]), int(h["profile_idx"])) for h in scanner.hits()}
            assert got == want_hits and {(short_q, 1), (long_q, 3)} <= got
        # a ranged scan re-plans for its own queries
        scanner.scan(multi, False, 10.0, kernel=dcp.KERNEL_QLANE2, q_range=(130, 570))
        gn2, ga2 = scanner.scores()
        assert same_bits(gn2[130:570], on[130:570]) and same_bits(ga2[130:570], oa[130:570])


def test_ring_hand_shake_is_bounded(dcp, hooks_scanner):
    """the two-stage kernel's LDS ring hand-shake polls a bounded number of times.  Through the
    tests' own -DDCP_TEST_HOOKS build one stage of the first task is made to sit out a step: its partner must run into
    the bound, set the scan's error word and drain -- dcp_gpu_sync returns RC_EFAIL (no hang, no scores taken for
    good) -- and the next scan on the same context is correct again."""
    import time
    sc = hooks_scanner
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    profiles = [dcp.ProteinProfile.sample(77 + i, m, cfg) for i, m in enumerate((40, 17, 9, 64))]
    rng = np.random.default_rng(3)
    seqs = rand_seqs(rng, 300, 50, 400)
    sc.upload_db(profiles)
    sc.upload_seqs(seqs)
    sc.scan(True, False, 10.0, kernel=dcp.KERNEL_ROWSWEEP)
    want_n, want_a = sc.scores()
    sc.test_set_ring_stall(True)
    try:
        t0 = time.time()
        with pytest.raises(dcp.DcpError) as ei:
            sc.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE2)  # sync=True: dcp_gpu_sync reports it
        took = time.time() - t0
        assert ei.value.rc == dcp.RC_EFAIL and "hand-shake" in str(ei.value)
        assert took < 60.0, took  # bounded: a few seconds of polling, not the lease
        with pytest.raises(dcp.DcpError):
            sc.scores()  # nothing of the failed scan is handed out
    finally:
        sc.test_set_ring_stall(False)
    sc.scan(True, False, 10.0, kernel=dcp.KERNEL_QLANE2)
    gn, ga = sc.scores()
    assert same_bits(gn, want_n) and same_bits(ga, want_a)
    # a uni-hit scan (no redo lists to check) reports the error word too
    sc.test_set_ring_stall(True)
    try:
        with pytest.raises(dcp.DcpError) as ei:
            sc.scan(False, False, 10.0, kernel=dcp.KERNEL_QLANE2)
        assert ei.value.rc == dcp.RC_EFAIL
    finally:
        sc.test_set_ring_stall(False)


def test_mixed_length_stress_both_kernels_agree(dcp):
    """BASELINE.json configs[4] shape at a size both kernels finish in seconds: 2 000 profiles with
    M log-uniform on 50..2000 x 600 queries log-uniform on 100..10 000 nt (length-sorted blocks, a partial
    last block, every size class up to W = 8).  Row sweep and query lane + redo must agree bit for bit on
    all 1.2e6 pairs and on the hits."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from concurrent.futures import ThreadPoolExecutor

    sizes = bench.core_sizes_for("c5", 2000)
    cfg = dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)
    with ThreadPoolExecutor(16) as ex:
        profiles = list(ex.map(lambda p: dcp.ProteinProfile.sample(0xDEC1F0 + p, int(sizes[p]), cfg), range(len(sizes))))
    queries = bench.make_queries(0, 600, 0)
    assert min(len(q) for q in queries) < 150 and max(len(q) for q in queries) > 8000
    sc = dcp.Scanner(0)
    try:
        sc.upload_db(profiles)
        del profiles
        sc.upload_seqs(queries)
        out = {}
        for name, k in (("qlane", dcp.KERNEL_QLANE), ("qlane2", dcp.KERNEL_QLANE2), ("rowsweep", dcp.KERNEL_ROWSWEEP)):
            sc.scan(True, False, 10.0, kernel=k)
            n, a = sc.scores()
            out[name] = (n.view(np.uint32).copy(), a.view(np.uint32).copy(), sc.hits())
        for other in ("qlane2", "rowsweep"):
            assert np.array_equal(out["qlane"][0], out[other][0]), other
            assert np.array_equal(out["qlane"][1], out[other][1]), other
            assert np.array_equal(out["qlane"][2], out[other][2]), other
        assert np.isfinite(out["qlane"][1].view(np.float32)).all()
    finally:
        sc.close()
