"""dcpx profile DB: round trip, header checks (the reference's: src/db/reader.c:25-79,
src/db/protein_reader.c:40-82) and profile_reader's partition table (src/db/profile_reader.c:45-72),
whose integer arithmetic must be bit-exact."""
import struct

import numpy as np
import pytest


def make(dcp, n=7, cfg=None):
    cfg = cfg or dcp.PROTEIN_CFG_DEFAULT
    return [dcp.ProteinProfile.sample(10 + i, 2 + 7 * i, cfg, f"PF{i:05d}.{i}") for i in range(n)]


def test_round_trip(dcp, tmp_path):
    profs = make(dcp)
    path = tmp_path / "db.dcpx"
    dcp.write_db(path, profs)
    db = dcp.ProfileDB(path)
    assert db.nprofiles == len(profs)
    assert db.cfg.entry_dist == dcp.ENTRY_DIST_OCCUPANCY and db.cfg.epsilon == dcp.PROTEIN_CFG_DEFAULT.epsilon
    back = db.read()
    for a, b in zip(profs, back):
        assert a.accession == b.accession and a.core_size == b.core_size and a.consensus == b.consensus
        assert np.array_equal(a.trans8, b.trans8) and np.array_equal(a.match_dist, b.match_dist)
        assert np.array_equal(a.null_dist, b.null_dist) and np.array_equal(a.insert_dist, b.insert_dist)
    part = db.read(2, 5)
    assert [p.accession for p in part] == [p.accession for p in profs[2:5]]
    assert int(db.profile_sizes.sum()) + 60 + 4 * db.nprofiles == path.stat().st_size
    db.close()


def ref_partitions(sizes, nparts, start):
    """partition_init + partition_it restated literally (profile_reader.c:45-72)."""
    n = len(sizes)
    nparts = min(nparts, n)
    ceil = -(-n // nparts)
    psize_of = lambda i: min(ceil, n - ceil * i)
    off = [0] * 65
    psize = [0] * 64
    off[0] = start
    i = size = 0
    for j in range(n):
        off[i + 1] += int(sizes[j])
        size += 1
        if size >= psize_of(i):
            psize[i] = size
            off[i + 1] += off[i]
            i += 1
            size = 0
    return psize[:nparts], off[:nparts + 1]


@pytest.mark.parametrize("nparts", [1, 2, 3, 4, 7, 64])
def test_partition_table_is_the_reference_arithmetic(dcp, tmp_path, nparts):
    profs = make(dcp, 11)
    path = tmp_path / "db.dcpx"
    dcp.write_db(path, profs)
    db = dcp.ProfileDB(path)
    sizes, offs = db.partitions(nparts)
    start = 60 + 4 * db.nprofiles
    want_sizes, want_offs = ref_partitions(db.profile_sizes, nparts, start)
    assert sizes == want_sizes and offs == want_offs
    # ceil(n / nparts)-sized partitions can leave trailing EMPTY ones (11 profiles, 7 partitions ->
    # 2,2,2,2,2,1,0); the reference never writes their end offset (it stays 0) -- reproduced as is
    used = sum(1 for s in sizes if s)
    assert offs[used] == path.stat().st_size and sum(sizes) == db.nprofiles
    assert all(o == 0 for o in offs[used + 1:])
    # reading partition by partition yields every profile once, in order
    names, b = [], 0
    for s in sizes:
        names += [p.accession for p in db.read(b, b + s)]
        b += s
    assert names == [p.accession for p in profs]
    for bad in (0, 65):
        with pytest.raises(dcp.DcpError):
            db.partitions(bad)


def patch(path, offset, fmt, value):
    raw = bytearray(path.read_bytes())
    struct.pack_into(fmt, raw, offset, value)
    path.write_bytes(bytes(raw))


@pytest.mark.parametrize("offset,fmt,value,what", [
    (4, "<H", 0xBEEF, "invalid magic number"), (8, "<I", 1, "invalid typeid"), (12, "<I", 8, "invalid float size"),
    (16, "<I", 0, "invalid entry dist"), (16, "<I", 3, "invalid entry dist"), (20, "<f", 1.5, "invalid epsilon"),
    (20, "<f", -0.1, "invalid epsilon"), (56, "<I", 0, "no profiles"), (56, "<I", (1 << 20) + 1, "too many profiles"),
])
def test_header_checks(dcp, tmp_path, offset, fmt, value, what):
    path = tmp_path / "db.dcpx"
    dcp.write_db(path, make(dcp, 3))
    patch(path, offset, fmt, value)
    with pytest.raises(dcp.DcpError) as e:
        dcp.ProfileDB(path)
    assert e.value.rc == dcp.RC_EINVAL, what


def test_io_errors(dcp, tmp_path):
    path = tmp_path / "db.dcpx"
    dcp.write_db(path, make(dcp, 3))
    raw = path.read_bytes()
    path.write_bytes(raw[:-10])  # truncated
    with pytest.raises(dcp.DcpError) as e:
        dcp.ProfileDB(path)
    assert e.value.rc == dcp.RC_EIO
    with pytest.raises(dcp.DcpError) as e:
        dcp.ProfileDB(tmp_path / "missing.dcpx")
    assert e.value.rc == dcp.RC_EIO
    # one protein_cfg per DB (protein_db_writer_open fixes it for all profiles)
    mixed = make(dcp, 2) + make(dcp, 1, dcp.ProteinCfg(dcp.ENTRY_DIST_UNIFORM, 0.01))
    with pytest.raises(dcp.DcpError) as e:
        dcp.write_db(tmp_path / "mixed.dcpx", mixed)
    assert e.value.rc == dcp.RC_EINVAL


@pytest.mark.gpu
def test_scan_from_file_equals_scan_from_memory(dcp, tmp_path):
    rng = np.random.default_rng(6)
    profs = [dcp.ProteinProfile.sample(40 + i, int(m)) for i, m in enumerate(rng.integers(2, 300, 12))]
    path = tmp_path / "db.dcpx"
    dcp.write_db(path, profs)
    back = dcp.ProfileDB(path).read()
    seqs = [rng.integers(0, 4, n, dtype=np.uint8) for n in (7, 100, 512)]
    res = []
    for ps in (profs, back):
        sc = dcp.Scanner(0)
        sc.upload_db(ps)
        sc.upload_seqs(seqs)
        sc.scan()
        res.append(sc.scores())
        sc.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
