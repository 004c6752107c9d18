"""HMMER3 ASCII -> profiles.  The reference parses with the absent `hmr` library and
its only golden for this path needs PF02545.hmm (a network download): parity is UNPINNED by the
reference here.  These tests pin the parser to the published text format with synthetic files, and
the resulting profiles to protein_profile_from_params on the same numbers."""
import numpy as np
import pytest

AMINO = "ACDEFGHIKLMNPQRSTVWY"


def fmt(v):
    return "      *" if np.isneginf(v) else "%9.5f" % (-v)


def write_hmm(path, profiles):
    """profiles: list of (name, acc, match lprobs [M,20], trans lprobs [M+1,7], consensus)."""
    with open(path, "w") as f:
        for name, acc, match, trans, cons in profiles:
            M = len(match)
            f.write("HMMER3/f [3.1b2 | February 2015]\n")
            f.write(f"NAME  {name}\n")
            if acc:
                f.write(f"ACC   {acc}\n")
            f.write(f"DESC  synthetic test profile\nLENG  {M}\nALPH  amino\nRF    no\nMM    no\nCONS  yes\nCS    no\nMAP   yes\n")
            f.write("NSEQ  10\nEFFN  1.5\nCKSUM 1\nSTATS LOCAL MSV       -9.0  0.7\n")
            f.write("HMM     " + "".join("%9s" % a for a in AMINO) + "\n")
            f.write("        " + "".join("%9s" % t for t in ("m->m", "m->i", "m->d", "i->m", "i->i", "d->m", "d->d")) + "\n")
            f.write("  COMPO " + "".join(fmt(np.log(0.05)) for _ in range(20)) + "\n")
            f.write("        " + "".join(fmt(np.log(0.05)) for _ in range(20)) + "\n")
            f.write("        " + "".join(fmt(v) for v in trans[0]) + "\n")
            for k in range(M):
                f.write("%7d " % (k + 1) + "".join(fmt(v) for v in match[k]) + "%7d %s - - -\n" % (k + 1, cons[k]))
                f.write("        " + "".join(fmt(np.log(0.05)) for _ in range(20)) + "\n")
                f.write("        " + "".join(fmt(v) for v in trans[k + 1]) + "\n")
            f.write("//\n")


def random_model(rng, M):
    def norm(x):
        return x - np.logaddexp.reduce(x, axis=-1, keepdims=True)

    match = norm(np.log(rng.random((M, 20)) + 0.01))
    trans = np.zeros((M + 1, 7))
    trans[:, 0:3] = norm(np.log(rng.random((M + 1, 3)) * np.array([10, 1, 1])))
    trans[:, 3:5] = norm(np.log(rng.random((M + 1, 2))))
    trans[:, 5:7] = norm(np.log(rng.random((M + 1, 2))))
    trans[0, 5], trans[0, 6] = 0.0, -np.inf  # HMMER writes d->m 0.0 and d->d * at the begin node
    trans[M, 2], trans[M, 6], trans[M, 5] = -np.inf, -np.inf, 0.0
    trans[M, 0:2] = norm(trans[M, 0:2])
    # what a reader sees: values rounded to 5 decimals of -ln p
    rd = lambda a: np.where(np.isneginf(a), a, -np.round(-a, 5))
    cons = "".join(AMINO[i].lower() for i in match.argmax(1))
    return rd(match), rd(trans), cons


def test_swissprot_background(dcp):
    lp = dcp.swissprot_null_lprobs()
    assert abs(np.exp(lp.astype(np.float64)).sum() - 1.0) < 1e-5
    assert abs(np.exp(lp[9]) - 0.0963728) < 1e-7 and abs(np.exp(lp[18]) - 0.0114135) < 1e-7  # L, W


def test_reads_profiles_like_from_params(dcp, tmp_path):
    rng = np.random.default_rng(8)
    models = [("fn3", "PF00041.13", *random_model(rng, 7)), ("noacc", "", *random_model(rng, 1)),
              ("big", "PF99999.1", *random_model(rng, 130))]
    path = tmp_path / "synthetic.hmm"
    write_hmm(path, models)
    cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
    profs = dcp.read_hmmer3(path, cfg)
    assert [p.accession for p in profs] == ["PF00041.13", "noacc", "PF99999.1"]
    assert [p.core_size for p in profs] == [7, 1, 130]
    null = dcp.swissprot_null_lprobs()
    for p, (_, _, match, trans, cons) in zip(profs, models):
        want = dcp.ProteinProfile.from_params(null, match.astype(np.float32), trans.astype(np.float32), cfg)
        assert np.array_equal(p.trans8, want.trans8, equal_nan=True)
        assert np.array_equal(p.match_dist, want.match_dist)
        assert np.array_equal(p.null_dist, want.null_dist) and np.array_equal(p.insert_dist, want.insert_dist)
        assert p.consensus == cons
    # '*' became -inf: D1 has no incoming edge, the last node has no MD/DD
    assert np.isneginf(profs[0].trans8[4:6, 0]).all()


def test_pfam_style_file_features(dcp, tmp_path):
    """What a Pfam-A.hmm written by HMMER 3.1b2 carries beyond the minimum: DATE/GA/TC/NC/BM/SM lines,
    three STATS lines, and five annotation columns (MAP CONS RF MM CS) after each match line.  No real
    Pfam file is available here (a network download in the reference's tests), so the features are
    added to a synthetic model and must not change what is read."""
    import re
    rng = np.random.default_rng(31)
    match, trans, cons = random_model(rng, 9)
    plain, pfam = str(tmp_path / "plain.hmm"), str(tmp_path / "pfam.hmm")
    write_hmm(plain, [("1-cysPrx_C", "PF10417.8", match, trans, cons)])
    s = open(plain).read()
    s = s.replace("MAP   yes\n", "MAP   yes\nDATE  Wed Sep 26 13:22:24 2018\n")
    s = s.replace("CKSUM 1\n", "CKSUM 2692542929\nGA    21.10 21.10;\nTC    21.10 21.10;\nNC    21.00 21.00;\n"
                  "BM    hmmbuild HMM.ann SEED.ann\nSM    hmmsearch -Z 45638612 -E 1000 --cpu 4 HMM pfamseq\n")
    s = s.replace("STATS LOCAL MSV       -9.0  0.7\n", "STATS LOCAL MSV       -7.4458  0.71858\n"
                  "STATS LOCAL VITERBI   -7.6857  0.71858\nSTATS LOCAL FORWARD   -3.8142  0.71858\n")
    s = re.sub(r"(\d+) (\w) - - -\n", r"\1 \2 - - H\n", s)
    open(pfam, "w").write(s)
    a, b = dcp.read_hmmer3(plain)[0], dcp.read_hmmer3(pfam)[0]
    assert b.accession == "PF10417.8" and b.core_size == 9 and b.consensus == a.consensus
    assert np.array_equal(a.match_dist, b.match_dist) and np.array_equal(a.trans8, b.trans8)
    assert np.array_equal(a.null_dist, b.null_dist)


@pytest.mark.parametrize("breakage,rc_name", [
    (lambda s: s.replace("ALPH  amino", "ALPH  DNA"), "RC_EPARSE"),
    (lambda s: s.replace("HMMER3/f", "HMMER2.0"), "RC_EPARSE"),
    (lambda s: s[: s.index("//")], "RC_EPARSE"),                       # truncated
    (lambda s: s.replace("LENG  4", "LENG  5"), "RC_EPARSE"),         # fewer nodes than LENG
    (lambda s: s.replace("      2 ", "      3 ", 1), "RC_EPARSE"),     # node index out of sequence
    (lambda s: s.replace("LENG  4", "LENG  0"), "RC_EINVAL"),         # protein_model_setup(0)
    (lambda s: s.replace("LENG  4", "LENG  5000"), "RC_EINVAL"),
])
def test_malformed_files_are_rejected(dcp, tmp_path, breakage, rc_name):
    rng = np.random.default_rng(1)
    good = tmp_path / "good.hmm"
    write_hmm(good, [("x", "PF1", *random_model(rng, 4))])
    assert len(dcp.read_hmmer3(good)) == 1
    bad = tmp_path / "bad.hmm"
    bad.write_text(breakage(good.read_text()))
    with pytest.raises(dcp.DcpError) as e:
        dcp.read_hmmer3(bad)
    assert dcp.RC_NAMES[e.value.rc] == rc_name
    with pytest.raises(dcp.DcpError):
        dcp.read_hmmer3(tmp_path / "missing.hmm")


@pytest.mark.gpu
def test_pressed_profiles_scan_like_from_params(dcp, tmp_path):
    rng = np.random.default_rng(3)
    models = [(f"m{i}", f"PF{i:05d}.1", *random_model(rng, M)) for i, M in enumerate((5, 60, 200, 300))]
    path = tmp_path / "db.hmm"
    write_hmm(path, models)
    cfg = dcp.PROTEIN_CFG_DEFAULT
    pressed = dcp.read_hmmer3(path, cfg)
    null = dcp.swissprot_null_lprobs()
    direct = [dcp.ProteinProfile.from_params(null, m.astype(np.float32), t.astype(np.float32), cfg) for _, _, m, t, _ in models]
    seqs = [rng.integers(0, 4, n, dtype=np.uint8) for n in (40, 333, 1053)]
    res = []
    for profs in (pressed, direct):
        sc = dcp.Scanner(0)
        sc.upload_db(profs)
        sc.upload_seqs(seqs)
        sc.scan(True, False, 10.0)
        res.append(sc.scores())
        sc.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.isfinite(res[0][1]).all()
