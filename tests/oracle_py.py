"""ctypes binding of the CPU oracle (oracle/liboracle_f{32,64}.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

ENTRY_DIST_UNIFORM = 1
ENTRY_DIST_OCCUPANCY = 2
NCODES = 1364

R_STATE, S_STATE, N_STATE, B_STATE, E_STATE, J_STATE, C_STATE, T_STATE = (
    (3 << 14) | i for i in range(8))


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle_f32.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("oracle.c", "oracle_io.c", "oracle.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(so) < os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])


def encode(seq: str) -> bytes:
    """ACGT text -> symbol ids 0..3 (imm_dna_iupac order)."""
    return bytes("ACGT".index(c) for c in seq)


class Profile:
    def __init__(self, orc, handle):
        self.orc = orc
        self.h = handle

    def __del__(self):
        try:
            self.orc.lib.orc_profile_del(self.h)
        except Exception:
            pass

    @property
    def core_size(self):
        return self.orc.lib.orc_profile_core_size(self.h)

    def setup(self, L, multi_hits=True, hmmer3_compat=False):
        return self.orc.lib.orc_profile_setup(self.h, L, int(multi_hits), int(hmmer3_compat))

    def viterbi(self, alt, seq: bytes, want_path=True):
        o = self.orc
        ll = o.fl()
        cap = 2 * len(seq) + 3 * self.core_size + 16
        n = C.c_uint(cap)
        if want_path:
            st = np.zeros(cap, np.uint16)
            ln = np.zeros(cap, np.uint8)
            rc = o.lib.orc_viterbi(self.h, int(alt), seq, len(seq), C.byref(ll),
                                   st.ctypes.data, ln.ctypes.data, C.byref(n))
            return rc, ll.value, list(zip(st[:n.value].tolist(), ln[:n.value].tolist()))
        rc = o.lib.orc_viterbi(self.h, int(alt), seq, len(seq), C.byref(ll), None, None, C.byref(n))
        return rc, ll.value, None

    def viterbi_fast(self, seq: bytes):
        o = self.orc
        nl, al = o.fl(), o.fl()
        rc = o.lib.orc_viterbi_fast(self.h, seq, len(seq), C.byref(nl), C.byref(al))
        return rc, nl.value, al.value

    def export(self):
        """(trans8 [8,M], emis_match [1364,M], emis_insert, emis_null, xtrans[13])"""
        o, M = self.orc, self.core_size
        t8 = np.zeros((8, M), o.np)
        em = np.zeros((NCODES, M), o.np)
        ei = np.zeros(NCODES, o.np)
        en = np.zeros(NCODES, o.np)
        xt = np.zeros(13, o.np)
        o.lib.orc_profile_export(self.h, t8.ctypes.data, em.ctypes.data, ei.ctypes.data,
                                 en.ctypes.data, xt.ctypes.data)
        return t8, em, ei, en, xt

    def dists(self):
        """(null [129], insert [129], match [M,129]) as nucltp[4] + codonm[125]."""
        o, M = self.orc, self.core_size
        nd = np.zeros(129, o.np)
        idd = np.zeros(129, o.np)
        md = np.zeros((M, 129), o.np)
        o.lib.orc_profile_dists(self.h, nd.ctypes.data, idd.ctypes.data, md.ctypes.data)
        return nd, idd, md

    def decode(self, frag: bytes, state_id: int):
        cod = C.create_string_buffer(3)
        lp = self.orc.lib.orc_profile_decode(self.h, frag, len(frag), state_id, cod)
        return lp, "".join("ACGT"[b] if b < 4 else "X" for b in cod.raw[:3])


    def path_score(self, alt, seq: bytes, path):
        """score of a given path [(state_id, seqlen), ...] in this model; NaN if it is not a path"""
        st = np.array([s for s, _ in path], np.uint16)
        ln = np.array([l for _, l in path], np.uint8)
        return self.orc.lib.orc_path_score(self.h, int(alt), seq, len(seq), st.ctypes.data, ln.ctypes.data, len(path))

    def codon_lprob(self, frag: bytes, state_id: int, codon: str):
        """joint log p(fragment, codon) -- what protein_profile_decode maximises over codons"""
        return self.orc.lib.orc_profile_codon_lprob(self.h, frag, len(frag), state_id,
                                                   bytes("ACGT".index(c) for c in codon))


class Oracle:
    def __init__(self, bits=32, lib_path=None):
        """lib_path: another build of oracle.c (bench.py's -O3 -march=native CPU-baseline build)."""
        if lib_path is None:
            build_oracle()
        self.bits = bits
        self.fl = C.c_double if bits == 64 else C.c_float
        self.np = np.float64 if bits == 64 else np.float32
        lib = C.CDLL(lib_path or os.path.join(ORACLE_DIR, f"liboracle_f{bits}.so"))
        fl = self.fl
        lib.orc_profile_sample.restype = C.c_void_p
        lib.orc_profile_sample.argtypes = [C.c_uint, C.c_uint, C.c_int, fl]
        lib.orc_profile_new.restype = C.c_void_p
        lib.orc_profile_new.argtypes = [C.c_uint, C.c_int, fl, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.orc_profile_del.argtypes = [C.c_void_p]
        lib.orc_profile_core_size.argtypes = [C.c_void_p]
        lib.orc_profile_core_size.restype = C.c_uint
        lib.orc_profile_nstates.argtypes = [C.c_void_p, C.c_int]
        lib.orc_profile_nstates.restype = C.c_uint
        lib.orc_profile_setup.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_int]
        lib.orc_profile_export.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        lib.orc_profile_dists.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        lib.orc_viterbi.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint, C.POINTER(fl),
                                    C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
        lib.orc_viterbi_fast.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.POINTER(fl), C.POINTER(fl)]
        lib.orc_dp_tables.argtypes = [C.c_uint, C.c_uint] + [C.c_void_p] * 5 + [
            C.c_char_p, C.c_uint, C.POINTER(fl), C.POINTER(fl)]
        lib.orc_profile_decode.restype = fl
        lib.orc_profile_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_uint, C.c_char_p]
        lib.orc_path_score.restype = fl
        lib.orc_path_score.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint]
        lib.orc_profile_codon_lprob.restype = fl
        lib.orc_profile_codon_lprob.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_uint, C.c_char_p]
        lib.orc_state_name.argtypes = [C.c_uint, C.c_char_p]
        lib.orc_lrt.restype = fl
        lib.orc_lrt.argtypes = [fl, fl]
        lib.orc_logaddexp.restype = fl
        lib.orc_logaddexp.argtypes = [fl, fl]
        lib.orc_frame_table.argtypes = [C.c_void_p, fl, C.c_void_p]
        lib.orc_setup_nuclt_dist.argtypes = [C.c_void_p, C.c_void_p]
        lib.orc_scan.restype = C.c_long
        lib.orc_scan.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_void_p, C.c_uint, C.c_int,
                                 C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.orc_scan_resident.restype = C.c_long
        lib.orc_scan_resident.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_void_p, C.c_uint, C.c_int,
                                          C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.orc_xtrans.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p]
        lib.orc_rnd_seed.argtypes = [C.c_void_p, C.c_uint64]
        lib.orc_rnd_dbl.restype = C.c_double
        lib.orc_rnd_dbl.argtypes = [C.c_void_p]
        self.lib = lib

    def sample(self, seed, core_size, entry_dist=ENTRY_DIST_OCCUPANCY, epsilon=0.01):
        eps = float(np.float32(epsilon))  # cfg literals are floats: protein_cfg(…, 0.1f)
        h = self.lib.orc_profile_sample(seed, core_size, entry_dist, eps)
        assert h
        return Profile(self, h)

    def new(self, null_lprobs, match_lprobs, trans, entry_dist=ENTRY_DIST_OCCUPANCY, epsilon=0.01):
        nl = np.ascontiguousarray(null_lprobs, self.np)
        ml = np.ascontiguousarray(match_lprobs, self.np)
        tr = np.ascontiguousarray(trans, self.np)
        M = ml.shape[0]
        assert ml.shape == (M, 20) and tr.shape == (M + 1, 7) and nl.shape == (20,)
        h = self.lib.orc_profile_new(M, entry_dist, float(np.float32(epsilon)), nl.ctypes.data,
                                     ml.ctypes.data, tr.ctypes.data)
        assert h
        return Profile(self, h)

    def state_name(self, sid):
        b = C.create_string_buffer(8)
        self.lib.orc_state_name(sid, b)
        return b.value.decode()

    def rnd_doubles(self, seed, n):
        st = (C.c_uint64 * 4)()
        self.lib.orc_rnd_seed(st, seed)
        return [self.lib.orc_rnd_dbl(st) for _ in range(n)]

    def frame_table(self, dist129, eps):
        d = np.ascontiguousarray(dist129, self.np)
        out = np.zeros(NCODES, self.np)
        self.lib.orc_frame_table(d.ctypes.data, float(eps), out.ctypes.data)
        return out

    def dp_tables(self, t8, em, ei, en, xt, seq: bytes):
        t8 = np.ascontiguousarray(t8, self.np)
        em = np.ascontiguousarray(em, self.np)
        ei = np.ascontiguousarray(ei, self.np)
        en = np.ascontiguousarray(en, self.np)
        xt = np.ascontiguousarray(xt, self.np)
        M = t8.shape[1]
        nl, al = self.fl(), self.fl()
        rc = self.lib.orc_dp_tables(M, M, t8.ctypes.data, em.ctypes.data, ei.ctypes.data,
                                    en.ctypes.data, xt.ctypes.data, seq, len(seq),
                                    C.byref(nl), C.byref(al))
        return rc, nl.value, al.value

    def scan(self, profiles, seqs, multi_hits=True, hmmer3_compat=False, lrt_thr=10.0,
             nthreads=1, mode=0):
        """thread_run restatement over all (seq, profile) pairs.
        returns (hits, null[nseq, nprof], alt[nseq, nprof])."""
        n = len(profiles)
        arr = (C.c_void_p * n)(*[p.h for p in profiles])
        off = np.zeros(len(seqs) + 1, np.uint32)
        off[1:] = np.cumsum([len(s) for s in seqs])
        cat = b"".join(seqs)
        on = np.zeros((len(seqs), n), self.np)
        oa = np.zeros((len(seqs), n), self.np)
        hits = self.lib.orc_scan(arr, n, cat, off.ctypes.data, len(seqs), int(multi_hits),
                                 int(hmmer3_compat), float(lrt_thr), nthreads, mode,
                                 on.ctypes.data, oa.ctypes.data)
        return hits, on, oa

    def scan_resident(self, profiles, seqs, multi_hits=True, hmmer3_compat=False, lrt_thr=10.0,
                      nthreads=1, want_scores=True):
        """Optimised CPU variant (DB resident, null once per sequence).
        returns (hits, null, alt, prepare_seconds, dp_seconds)."""
        n = len(profiles)
        arr = (C.c_void_p * n)(*[p.h for p in profiles])
        off = np.zeros(len(seqs) + 1, np.uint32)
        off[1:] = np.cumsum([len(s) for s in seqs])
        cat = b"".join(seqs)
        on = np.zeros((len(seqs), n), self.np) if want_scores else None
        oa = np.zeros((len(seqs), n), self.np) if want_scores else None
        tp, td = C.c_double(0), C.c_double(0)
        hits = self.lib.orc_scan_resident(arr, n, cat, off.ctypes.data, len(seqs), int(multi_hits),
                                          int(hmmer3_compat), float(lrt_thr), nthreads,
                                          on.ctypes.data if want_scores else None,
                                          oa.ctypes.data if want_scores else None, C.byref(tp), C.byref(td))
        if hits < 0:
            raise ValueError("orc_scan_resident: bad input")
        return hits, on, oa, tp.value, td.value

    # ---- the oracle's own file readers (oracle/oracle_io.c) ------------------------------------------
    def _bind_io(self):
        lib, fl = self.lib, self.fl
        if getattr(lib, "_io_bound", False):
            return
        lib.orc_swissprot_null.argtypes = [C.c_void_p]
        lib.orc_h3_open.restype = C.c_void_p
        lib.orc_h3_open.argtypes = [C.c_char_p, C.c_int, fl]
        lib.orc_h3_next.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        for f in ("orc_h3_error", "orc_h3_acc", "orc_h3_consensus"):
            getattr(lib, f).restype = C.c_char_p
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.orc_h3_close.argtypes = [C.c_void_p]
        lib.orc_dcp_open.restype = C.c_void_p
        lib.orc_dcp_open.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        lib.orc_dcp_close.argtypes = [C.c_void_p]
        lib.orc_dcp_nprofiles.argtypes = [C.c_void_p]
        lib.orc_dcp_nprofiles.restype = C.c_uint
        lib.orc_dcp_entry_dist.argtypes = [C.c_void_p]
        lib.orc_dcp_epsilon.argtypes = [C.c_void_p]
        lib.orc_dcp_epsilon.restype = fl
        lib.orc_dcp_profile.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_uint), C.c_char_p] + [C.c_void_p] * 5 + [C.c_char_p]
        lib.orc_dcp_score.argtypes = [C.c_void_p, C.c_uint, C.c_char_p, C.c_uint, C.c_int, C.c_int, C.POINTER(fl),
                                      C.POINTER(fl)]
        lib._io_bound = True

    def swissprot_null(self):
        self._bind_io()
        out = np.zeros(20, self.np)
        self.lib.orc_swissprot_null(out.ctypes.data)
        return out

    def read_hmmer3(self, path, entry_dist=ENTRY_DIST_OCCUPANCY, epsilon=0.01):
        """Every profile of a HMMER3 ASCII file through the oracle's own parser:
        [(Profile, accession-or-name, consensus)].  Raises ValueError(rc, message) on a bad file."""
        self._bind_io()
        r = self.lib.orc_h3_open(os.fsencode(str(path)), entry_dist, float(np.float32(epsilon)))
        if not r:
            raise ValueError(4, "cannot open")
        out = []
        try:
            while True:
                h = C.c_void_p()
                rc = self.lib.orc_h3_next(r, C.byref(h))
                if rc == 1:
                    return out
                if rc:
                    raise ValueError(rc, self.lib.orc_h3_error(r).decode())
                out.append((Profile(self, h.value), self.lib.orc_h3_acc(r).decode(), self.lib.orc_h3_consensus(r).decode()))
        finally:
            self.lib.orc_h3_close(r)

    def open_dcp(self, path):
        self._bind_io()
        return DcpFile(self, path)

    def xtrans(self, L, multi_hits=True, hmmer3_compat=False):
        out = np.zeros(13, self.np)
        rc = self.lib.orc_xtrans(L, int(multi_hits), int(hmmer3_compat), out.ctypes.data)
        return rc, out


class DcpFile:
    """A MessagePack .dcp database through the oracle's own reader (oracle/oracle_io.c)."""

    def __init__(self, orc, path):
        self.orc = orc
        err = C.create_string_buffer(160)
        self.h = orc.lib.orc_dcp_open(os.fsencode(str(path)), err, len(err))
        if not self.h:
            raise ValueError(err.value.decode())
        self.nprofiles = orc.lib.orc_dcp_nprofiles(self.h)
        self.entry_dist = orc.lib.orc_dcp_entry_dist(self.h)
        self.epsilon = orc.lib.orc_dcp_epsilon(self.h)

    def close(self):
        h, self.h = self.h, None
        if h:
            self.orc.lib.orc_dcp_close(h)

    __del__ = close

    def profile(self, i):
        """dict(core_size, accession, consensus, trans8 [8,M], xtrans [13], null [129], insert [129], match [M,129])"""
        o = self.orc
        m = C.c_uint(0)
        rc = o.lib.orc_dcp_profile(self.h, i, C.byref(m), None, None, None, None, None, None, None)
        if rc:
            raise ValueError(f"profile {i}: rc {rc}")
        M = m.value
        acc, cons = C.create_string_buffer(32), C.create_string_buffer(M + 1)
        t8, xt = np.zeros((8, M), o.np), np.zeros(13, o.np)
        nd, idd, md = np.zeros(129, o.np), np.zeros(129, o.np), np.zeros((M, 129), o.np)
        rc = o.lib.orc_dcp_profile(self.h, i, C.byref(m), acc, t8.ctypes.data, xt.ctypes.data, nd.ctypes.data,
                                   idd.ctypes.data, md.ctypes.data, cons)
        if rc:
            raise ValueError(f"profile {i}: rc {rc}")
        return dict(core_size=M, accession=acc.value.decode(), consensus=cons.value.decode(), trans8=t8, xtrans=xt,
                    null=nd, insert=idd, match=md)

    def score(self, i, seq: bytes, multi_hits=True, hmmer3_compat=False):
        o = self.orc
        nl, al = o.fl(), o.fl()
        rc = o.lib.orc_dcp_score(self.h, i, seq, len(seq), int(multi_hits), int(hmmer3_compat), C.byref(nl), C.byref(al))
        if rc:
            raise ValueError(f"score: rc {rc}")
        return nl.value, al.value
