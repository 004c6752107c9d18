"""Host-side model builder of the product vs the oracle (CPU only).

The product computes in double / probability domain and rounds once to float32;
the oracle mimics imm's float32 log-domain chains.  Agreement is therefore to a
few float32 ulps, well inside the reference's own float32 bar of 5e-5 relative
(test/hope_support.h:26)."""
import numpy as np
import pytest

from oracle_py import ENTRY_DIST_OCCUPANCY, ENTRY_DIST_UNIFORM

RTOL = 5e-6


def close(a, b, rtol=RTOL, atol=5e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    inf = np.isneginf(a)
    assert np.array_equal(inf, np.isneginf(b))
    assert not np.isnan(a).any() and not np.isnan(b).any()
    np.testing.assert_allclose(a[~inf], b[~inf], rtol=rtol, atol=atol)


@pytest.mark.parametrize("entry", [ENTRY_DIST_UNIFORM, ENTRY_DIST_OCCUPANCY])
@pytest.mark.parametrize("seed,M,eps", [(1, 2, 0.1), (2, 2, 0.01), (7, 5, 0.01), (11, 37, 0.01)])
def test_sample_matches_oracle(dcp, oracle32, oracle64, entry, seed, M, eps):
    prof = dcp.ProteinProfile.sample(seed, M, dcp.ProteinCfg(entry, eps))
    assert prof.core_size == M
    for orc in (oracle32, oracle64):
        op = orc.sample(seed, M, entry, eps)
        t8, em, ei, en, _ = op.export()
        nd, idd, md = op.dists()
        close(prof.trans8, t8)
        close(prof.null_dist, nd)
        close(prof.insert_dist, idd)
        close(prof.match_dist, md)
        # frame-state emission tables
        e32 = float(np.float32(eps))
        close(dcp.frame_table_host(prof.null_dist, e32), en)
        close(dcp.frame_table_host(prof.insert_dist, e32), ei)
        for k in range(0, M, max(1, M // 3)):
            close(dcp.frame_table_host(prof.match_dist[k], e32), em[:, k])


def test_frame_table_is_a_distribution(dcp):
    """Self-check that needs no imm (SURVEY §8c ii): total mass over all 1364 words is 1."""
    prof = dcp.ProteinProfile.sample(3, 4)
    for eps in (0.0, 0.01, 0.1, 0.5):
        for d in (prof.null_dist, prof.insert_dist, prof.match_dist[2]):
            t = dcp.frame_table_host(d, eps).astype(np.float64)
            assert abs(np.exp(t).sum() - 1.0) < 1e-5


def test_from_params_matches_oracle(dcp, oracle32):
    rng = np.random.default_rng(5)

    def norm(x):
        return x - np.logaddexp.reduce(x, axis=-1, keepdims=True)

    M = 9
    null = norm(np.log(rng.random(20))).astype(np.float32)
    match = norm(np.log(rng.random((M, 20)))).astype(np.float32)
    trans = np.log(rng.random((M + 1, 7)))
    trans[0, 6] = -np.inf
    trans[M, 2] = trans[M, 6] = -np.inf
    trans = norm(trans).astype(np.float32)
    prof = dcp.ProteinProfile.from_params(null, match, trans, dcp.ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01))
    op = oracle32.new(null, match, trans, ENTRY_DIST_OCCUPANCY, 0.01)
    t8 = op.export()[0]
    close(prof.trans8, t8)
    # the edges that do not exist are -inf: D1 has no incoming edge, I_M has no edges
    assert np.isneginf(prof.trans8[1:6, 0]).all() and np.isneginf(prof.trans8[6:8, M - 1]).all()


@pytest.mark.parametrize("L", [1, 2, 5, 32, 300, 1000, 1053, 10000])
@pytest.mark.parametrize("multi,h3", [(True, False), (False, False), (True, True)])
def test_xtrans_matches_oracle(dcp, oracle32, L, multi, h3):
    op = oracle32.sample(1, 2)
    assert op.setup(L, multi, h3) == 0
    close(dcp.xtrans(L, multi, h3), op.export()[4], rtol=2e-6, atol=2e-7)


def test_setup_rejects_empty_sequence(dcp):
    # EQ(protein_profile_setup(&prof, 0, true, false), RC_EINVAL)  test/protein_profile.c:31
    with pytest.raises(dcp.DcpError) as e:
        dcp.xtrans(0)
    assert e.value.rc == dcp.RC_EINVAL


def test_model_limits(dcp):
    cfg = dcp.ProteinCfg()
    z20 = np.zeros(20, np.float32)
    with pytest.raises(dcp.DcpError) as e:  # protein_model.c:157
        dcp.ProteinProfile.from_params(z20, np.zeros((0, 20), np.float32), np.zeros((1, 7), np.float32), cfg)
    assert e.value.rc == dcp.RC_EINVAL
    with pytest.raises(dcp.DcpError):  # protein_model.c:159
        dcp.ProteinProfile.sample(1, dcp.CORE_SIZE_MAX + 1)
    with pytest.raises(dcp.DcpError):  # assert(core_size >= 2) protein_profile.c:262
        dcp.ProteinProfile.sample(1, 1)
    with pytest.raises(dcp.DcpError):
        dcp.ProteinCfg(ENTRY_DIST_UNIFORM, 1.5)


def test_lrt(dcp):
    # xmath_lrt(-48.927f, -54.355f) = -10.856 (compiled from xmath.h)
    assert abs(dcp.lrt(-48.927, -54.355) - (-10.856)) < 1e-3
    a, b = np.float32(-1430.5), np.float32(-1420.25)
    assert dcp.lrt(a, b) == np.float32(-2) * (a - b)


def ref_partition_sizes(n, nparts):
    """partition_it + xmath_partition_size restated literally (profile_reader.c:54-72)."""
    nparts = min(nparts, n)
    size_of = lambda i: min(-(-n // nparts), n - (-(-n // nparts)) * i)
    out, i, size = [0] * nparts, 0, 0
    for _ in range(n):
        size += 1
        if size >= size_of(i):
            out[i] = size
            i += 1
            size = 0
    return out


@pytest.mark.parametrize("n,parts", [(2, 1), (20000, 8), (10, 4), (9, 4), (5, 4), (3, 64), (64, 64),
                                     (1000, 7), (1, 1), (20000, 64)])
def test_partition_by_count_is_the_reference_split(dcp, n, parts):
    got = dcp.partition_by_count(n, parts)
    assert got == ref_partition_sizes(n, parts)
    assert sum(got) == n
    # xmath_partition_size(20000, 8, 7) = 2500
    if (n, parts) == (20000, 8):
        assert got[7] == 2500


def test_partition_rejects_bad_counts(dcp):
    for bad in (0, 65):
        with pytest.raises(dcp.DcpError) as e:  # profile_reader.c:77-79
            dcp.partition_by_count(10, bad)
        assert e.value.rc == dcp.RC_EINVAL


def test_partition_by_cells_balances_work(dcp):
    rng = np.random.default_rng(0)
    cs = np.clip(np.round(np.exp(rng.normal(np.log(150), 0.6, 5000))), 30, 2000).astype(np.uint32)
    for g in (1, 2, 4, 8):
        b = dcp.partition_by_cells(cs, g)
        assert b[0] == 0 and b[-1] == len(cs) and all(x <= y for x, y in zip(b, b[1:]))
        loads = [int(cs[b[i]:b[i + 1]].sum()) for i in range(g)]
        assert max(loads) - min(loads) <= 2 * int(cs.max())
