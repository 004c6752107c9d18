"""SURVEY 8f N2 / N3 with an oracle on the other side: the files either side of the scan
path are read by the oracle's OWN parsers (oracle/oracle_io.c: a MessagePack walker and a HMMER3 text
parser written separately from the product's) and the HIP path is scored against the oracle reading the
SAME BYTES -- no longer the product against itself.

Both stay "parity unpinned by the reference": the reference tree holds neither a .hmm nor a .dcp fixture
(its tests download them), its HMMER3 parser (hmmer-reader) and MessagePack layer (lite-pack) are absent,
and the two imm_dp values inside a profile are the product's own encoding because imm's is not in the tree.
What these tests pin is product == oracle on identical input bytes."""
import os
import struct
import subprocess

import numpy as np
import pytest

import test_h3reader as h3
from test_c_host import build_c_test, build_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, tol=5e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= tol * np.abs(b)))


# ---- N3: HMMER3 ASCII ------------------------------------------------------------------------------------
def hmm_models(rng, sizes):
    return [(f"m{i}", f"PF{i:05d}.{i % 9}" if i != 1 else "", *h3.random_model(rng, M)) for i, M in enumerate(sizes)]


def test_oracle_h3_parser_reads_the_numbers_that_were_written(oracle32, tmp_path):
    """The oracle's parser against the generator's own arrays: a profile read from the text equals a
    profile built from the same numbers (text -> double -> float, '*' -> -inf, CONS, ACC / NAME)."""
    rng = np.random.default_rng(8)
    models = hmm_models(rng, (7, 1, 130))
    path = tmp_path / "o.hmm"
    h3.write_hmm(path, models)
    got = oracle32.read_hmmer3(path)
    assert [acc for _, acc, _ in got] == ["PF00000.0", "m1", "PF00002.2"]
    null = oracle32.swissprot_null()
    assert abs(np.exp(null.astype(np.float64)).sum() - 1) < 1e-5
    for (prof, _, cons), (_, _, match, trans, want_cons) in zip(got, models):
        want = oracle32.new(null, match.astype(np.float32), trans.astype(np.float32))
        for a, b in zip(prof.export(), want.export()):
            assert np.array_equal(a, b, equal_nan=True)
        assert cons == want_cons
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.hmm"
        bad.write_text(path.read_text().replace("ALPH  amino", "ALPH  DNA"))
        oracle32.read_hmmer3(bad)
    with pytest.raises(ValueError):
        bad.write_text(path.read_text()[:-40])
        oracle32.read_hmmer3(bad)


def test_product_and_oracle_parse_the_same_hmm_alike(dcp, oracle32, tmp_path):
    """CPU: both parsers on the same file -> the same transitions, and distributions within float32
    rounding of two different builds (product: probability domain in float64; oracle: log domain, imm-style)."""
    rng = np.random.default_rng(21)
    path = tmp_path / "both.hmm"
    h3.write_hmm(path, hmm_models(rng, (3, 40, 77)))
    cfg = dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)
    prod = dcp.read_hmmer3(path, cfg)
    orc = oracle32.read_hmmer3(path)
    assert [p.accession for p in prod] == [acc for _, acc, _ in orc]
    assert [p.consensus for p in prod] == [c for _, _, c in orc]
    for p, (o, _, _) in zip(prod, orc):
        t8, _, _, _, _ = o.export()
        fin = np.isfinite(t8)
        assert np.array_equal(fin, np.isfinite(p.trans8))
        assert close(p.trans8[fin], t8[fin], 2e-6)
        nd, idd, md = o.dists()
        for a, b in ((p.null_dist, nd), (p.insert_dist, idd), (p.match_dist, md)):
            f = np.isfinite(b)
            assert np.array_equal(f, np.isfinite(a)) and np.allclose(a[f], b[f], rtol=2e-5, atol=1e-6)  # log 1 ~ 0


@pytest.mark.gpu
def test_hmm_through_the_hip_path_against_the_oracle_reading_the_same_file(dcp, oracle32, tmp_path):
    """N3 on the device: a synthetic .hmm scored by the product (own parser -> profiles -> HIP scan, and
    through the reference-named C route protein_h3reader_next + protein_profile_absorb + imm_dp_viterbi)
    against the oracle's parser + the oracle's Viterbi on the same bytes."""
    rng = np.random.default_rng(3)
    sizes = (5, 60, 200, 300)
    path = tmp_path / "db.hmm"
    h3.write_hmm(path, hmm_models(rng, sizes))
    seqs = [rng.integers(0, 4, n, dtype=np.uint8) for n in (40, 333, 1053)]
    orc = oracle32.read_hmmer3(path)
    for multi, compat in ((True, False), (False, False), (True, True)):
        _, on, oa = oracle32.scan([o for o, _, _ in orc], [bytes(s) for s in seqs], multi, compat, 10.0, 1, 1)
        sc = dcp.Scanner(0)
        sc.upload_db(dcp.read_hmmer3(path, dcp.ProteinCfg(dcp.ENTRY_DIST_OCCUPANCY, 0.01)))
        sc.upload_seqs(seqs)
        for kernel in (dcp.KERNEL_ROWSWEEP, dcp.KERNEL_QLANE2):
            sc.scan(multi, compat, 10.0, kernel=kernel)
            gn, ga = sc.scores()
            assert close(gn, on) and close(ga, oa)
        sc.close()
        if (multi, compat) == (True, False):
            # the reference's own call sequence, in C
            build_host()
            tool = build_c_test(tmp_path, "dcp_tool")
            (tmp_path / "seqs.txt").write_text("\n".join("".join("ACGT"[b] for b in s) for s in seqs) + "\n")
            out = tmp_path / "hmm_scores.bin"
            subprocess.run([tool, "hmm", str(path), str(tmp_path / "seqs.txt"), str(out), "1", "0", "2", "0.01"],
                           check=True, timeout=300)
            got = np.fromfile(out, np.float32).reshape(len(sizes), len(seqs), 2)
            assert close(got[:, :, 0].T, on) and close(got[:, :, 1].T, oa)


# ---- N2: MessagePack .dcp ----------------------------------------------------------------------------------
def press(tmp_path, n=9, seed0=100):
    build_host()
    tool = build_c_test(tmp_path, "dcp_tool")
    dcp_path, side = tmp_path / "pressed.dcp", tmp_path / "pressed.side"
    subprocess.run([tool, "press", str(dcp_path), str(side), str(n), str(seed0)], check=True, timeout=120)
    return tool, dcp_path, side


def read_sidecar(path):
    raw = path.read_bytes()
    at, out = 0, []
    while at < len(raw):
        (M,) = struct.unpack_from("<I", raw, at)
        at += 4
        f = lambda n: np.frombuffer(raw, np.float32, n, at)
        t8 = f(8 * M).reshape(8, M)
        at += 32 * M
        nd = f(129)
        at += 516
        idd = f(129)
        at += 516
        md = f(129 * M).reshape(M, 129)
        at += 516 * M
        out.append((M, t8, nd, idd, md))
    return out


def test_oracle_reads_a_dcp_the_product_pressed(dcp, oracle32, tmp_path):
    """CPU: a database pressed through the reference-named writer API (protein_db_writer_open /
    _pack_profile / db_writer_close), parsed by the oracle's MessagePack walker: every stored value equals
    what the product held in memory, bit for bit; sizes, keys and framing are as the oracle expects."""
    _, dcp_path, side = press(tmp_path)
    db = oracle32.open_dcp(dcp_path)
    want = read_sidecar(side)
    assert db.nprofiles == len(want) == 9
    assert db.entry_dist == dcp.ENTRY_DIST_OCCUPANCY and abs(db.epsilon - 0.01) < 1e-9
    for i, (M, t8, nd, idd, md) in enumerate(want):
        p = db.profile(i)
        assert p["core_size"] == M and p["accession"] == f"PF{i:05d}.{i % 7}" and len(p["consensus"]) == M
        assert np.array_equal(p["trans8"], t8) and np.array_equal(p["null"], nd)
        assert np.array_equal(p["insert"], idd) and np.array_equal(p["match"], md)
        assert np.all(p["xtrans"] == 0)  # a pressed profile carries the LOG1 defaults (protein_model.c:322-340)
    db.close()
    # a damaged file is refused, not half-read
    raw = dcp_path.read_bytes()
    for cut in (len(raw) - 7, len(raw) // 2, 40):
        (tmp_path / "cut.dcp").write_bytes(raw[:cut])
        with pytest.raises(ValueError):
            oracle32.open_dcp(tmp_path / "cut.dcp")


@pytest.mark.gpu
def test_dcp_through_the_hip_path_against_the_oracle_reading_the_same_file(dcp, oracle32, tmp_path):
    """N2 on the device: the pressed file goes through protein_db_reader_open -> profile_reader_next ->
    protein_profile_setup -> imm_dp_viterbi (HIP) and, independently, through the oracle's reader +
    orc_frame_table + the oracle's DP; same bytes in, scores within the float32 tolerance."""
    tool, dcp_path, _ = press(tmp_path)
    rng = np.random.default_rng(17)
    seqs = [rng.integers(0, 4, n, dtype=np.uint8) for n in (33, 150, 700)]
    (tmp_path / "seqs.txt").write_text("\n".join("".join("ACGT"[b] for b in s) for s in seqs) + "\n")
    db = oracle32.open_dcp(dcp_path)
    for multi, compat in ((1, 0), (0, 0)):
        out = tmp_path / f"scores{multi}.bin"
        subprocess.run([tool, "scan", str(dcp_path), str(tmp_path / "seqs.txt"), str(out), str(multi), str(compat)],
                       check=True, timeout=600)
        got = np.fromfile(out, np.float32).reshape(db.nprofiles, len(seqs), 2)
        want = np.array([[db.score(i, bytes(s), bool(multi), bool(compat)) for s in seqs] for i in range(db.nprofiles)])
        assert np.isfinite(want).all()
        assert close(got, want)
    db.close()
