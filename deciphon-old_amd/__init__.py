"""deciphon-old_amd -- MI355X-native profile-HMM scan engine (host binding).

Thin ctypes layer over the C-ABI of ``include/dcp_gpu.h`` (``libdcp_hip.so``,
built from ``csrc/`` by hipcc for gfx950).  It mirrors the reference's interface
for the scan hot path so tests read like the reference's own:

=====================================  =========================================
reference (deciphon-old)               here
=====================================  =========================================
protein_profile_init + _sample         ``ProteinProfile.sample(seed, core_size, cfg)``
protein_profile_init + _absorb(model)  ``ProteinProfile.from_params(...)``
protein_profile_setup(prof, L, ...)    ``xtrans(L, multi_hits, hmmer3_compat)`` (EINVAL on L=0)
profile_reader_setup/rewind/next       ``Scanner.upload_db(profiles)`` (resident, once)
imm_seq + imm_task_setup               ``Scanner.upload_seqs(seqs)``
thread_run (per (seq, profile) pair)   ``Scanner.scan(multi_hits, hmmer3_compat, lrt_threshold)``
xmath_lrt                              ``lrt(null, alt)``
=====================================  =========================================

There is NO CPU fallback: importing works without a GPU (so the host logic and
the symbol table can be tested), but creating a ``Scanner`` without a HIP device
raises, and a missing ``libdcp_hip.so`` raises at import.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libdcp_hip.so")
CSRC = os.path.join(_HERE, "csrc")

# enum rc (include/deciphon/core/rc.h:4-15)
RC_OK, RC_END, RC_EFAIL, RC_EINVAL, RC_EIO, RC_ENOMEM, RC_EPARSE, RC_EAPI, RC_EHTTP = range(9)
RC_NAMES = ["RC_OK", "RC_END", "RC_EFAIL", "RC_EINVAL", "RC_EIO", "RC_ENOMEM", "RC_EPARSE",
            "RC_EAPI", "RC_EHTTP"]
# enum entry_dist (include/deciphon/model/entry_dist.h)
ENTRY_DIST_NULL, ENTRY_DIST_UNIFORM, ENTRY_DIST_OCCUPANCY = 0, 1, 2
NCODES = 1364
NDIST = 129
NXTRANS = 13
CORE_SIZE_MAX = 4096
NUM_THREADS = 64

# every symbol include/dcp_gpu.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "dcp_profile_new", "dcp_profile_sample", "dcp_profile_del", "dcp_profile_core_size",
    "dcp_profile_from_parts", "dcp_profile_entry_dist", "dcp_profile_epsilon", "dcp_rnd_seed", "dcp_rnd_next",
    "dcp_dist_unique_id", "dcp_dist_init", "dcp_dist_init_from_file", "dcp_dist_free", "dcp_dist_rank",
    "dcp_dist_nranks", "dcp_dist_last_error", "dcp_dist_shard", "dcp_dist_gather_hits", "dcp_dist_free_hits",
    "dcp_dist_merge_hits", "dcp_dist_gather_scan_hits", "dcp_dist_gather_plan", "dcp_dist_init_from_file_run",
    "dcp_dist_comm_count", "dcp_dist_last_gather_ms", "dcp_plan_query_slots",
    "dcp_lprob_normalize", "dcp_h3reader_open_fp", "dcp_h3reader_next_params", "dcp_gpu_seqs_set_xtrans",
    "dcp_profile_accession", "dcp_profile_trans8", "dcp_profile_null_dist",
    "dcp_profile_insert_dist", "dcp_profile_match_dist", "dcp_frame_table_host", "dcp_xtrans",
    "dcp_lrt", "dcp_partition_by_count", "dcp_partition_by_cells", "dcp_gpu_device_count",
    "dcp_gpu_ctx_new", "dcp_gpu_ctx_del", "dcp_gpu_last_error", "dcp_gpu_stream",
    "dcp_gpu_db_upload", "dcp_gpu_db_nprofiles", "dcp_gpu_db_fetch_match_table",
    "dcp_gpu_db_one_layout", "dcp_gpu_db_table_bytes",
    "dcp_gpu_seqs_upload", "dcp_gpu_seqs_upload_text", "dcp_gpu_nseqs", "dcp_gpu_scan",
    "dcp_gpu_sync", "dcp_gpu_last_scan_kernel", "dcp_gpu_hit_buffer", "dcp_gpu_last_scan_redo_pairs", "dcp_gpu_last_scan_ms", "dcp_gpu_last_scan_launches", "dcp_gpu_fetch_scores",
    "dcp_gpu_fetch_hits", "dcp_gpu_scan_range", "dcp_gpu_set_hit_buffer",
    "dcp_gpu_last_scan_launch_info", "dcp_gpu_scan_cells", "dcp_gpu_scan_algorithmic_bytes",
    "dcp_gpu_trace_paths", "dcp_state_name", "dcp_profile_decode", "dcp_gc_decode",
    "dcp_prod_format_row", "dcp_prod_header", "dcp_h3reader_open", "dcp_h3reader_next",
    "dcp_h3reader_error", "dcp_h3reader_close", "dcp_swissprot_null_lprobs", "dcp_profile_consensus",
    "dcp_db_write", "dcp_db_open", "dcp_db_close", "dcp_db_nprofiles", "dcp_db_entry_dist", "dcp_db_epsilon",
    "dcp_db_profile_sizes", "dcp_db_partitions", "dcp_db_read",
]


class DcpError(RuntimeError):
    def __init__(self, rc, msg=""):
        self.rc = rc
        name = RC_NAMES[rc] if 0 <= rc < len(RC_NAMES) else str(rc)
        super().__init__(f"{name}: {msg}" if msg else name)


def build(verbose=False):
    """Compile csrc/ for gfx950 into libdcp_hip.so (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", CSRC] + ([] if verbose else ["-s"]))
    return LIB_PATH


class ScanParams(C.Structure):
    _fields_ = [("multi_hits", C.c_int), ("hmmer3_compat", C.c_int),
                ("lrt_threshold", C.c_float), ("keep_scores", C.c_int), ("kernel", C.c_int)]


KERNEL_AUTO, KERNEL_ROWSWEEP, KERNEL_QLANE, KERNEL_QLANE2 = 0, 1, 2, 3
DB_EXPAND_ON_HOST, DB_ONE_LAYOUT = 1, 2  # dcp_gpu_db_upload flags (include/dcp_gpu.h)


class Hit(C.Structure):
    _fields_ = [("seq_idx", C.c_uint32), ("profile_idx", C.c_uint32),
                ("null_loglik", C.c_float), ("alt_loglik", C.c_float)]


class LaunchInfo(C.Structure):
    _fields_ = [("nodes_per_lane", C.c_int), ("waves_per_pair", C.c_int), ("nprofiles", C.c_uint),
                ("ms", C.c_float), ("cells", C.c_uint64), ("algorithmic_bytes", C.c_uint64)]


STEP_DTYPE = np.dtype([("state_id", np.uint16), ("seqlen", np.uint8), ("reserved", np.uint8)])
HIT_DTYPE = np.dtype([("seq_idx", np.uint32), ("profile_idx", np.uint32),
                      ("null_loglik", np.float32), ("alt_loglik", np.float32)])


def _load(path=None, hooks=False):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C deciphon-old_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(path)
    P, F, U, I = C.c_void_p, C.c_float, C.c_uint, C.c_int
    sig = {
        "dcp_profile_new": (P, [C.c_char_p, U, I, F, P, P, P, C.c_char_p, C.POINTER(I)]),
        "dcp_profile_sample": (P, [C.c_char_p, U, U, I, F, C.POINTER(I)]),
        "dcp_profile_del": (None, [P]),
        "dcp_profile_core_size": (U, [P]),
        "dcp_profile_accession": (C.c_char_p, [P]),
        "dcp_profile_trans8": (P, [P]),
        "dcp_profile_null_dist": (P, [P]),
        "dcp_profile_insert_dist": (P, [P]),
        "dcp_profile_match_dist": (P, [P]),
        "dcp_frame_table_host": (None, [P, F, P]),
        "dcp_xtrans": (I, [U, I, I, P]),
        "dcp_lrt": (F, [F, F]),
        "dcp_partition_by_count": (U, [U, U, P]),
        "dcp_partition_by_cells": (None, [P, U, U, P]),
        "dcp_gpu_device_count": (I, []),
        "dcp_gpu_ctx_new": (P, [I]),
        "dcp_gpu_ctx_del": (None, [P]),
        "dcp_gpu_last_error": (C.c_char_p, [P]),
        "dcp_gpu_stream": (P, [P]),
        "dcp_gpu_db_upload": (I, [P, P, U, I]),
        "dcp_gpu_db_nprofiles": (U, [P]),
        "dcp_gpu_db_one_layout": (I, [P]),
        "dcp_gpu_db_table_bytes": (C.c_uint64, [P]),
        "dcp_gpu_db_fetch_match_table": (I, [P, U, P]),
        "dcp_gpu_seqs_upload": (I, [P, P, P, U]),
        "dcp_gpu_seqs_upload_text": (I, [P, C.c_char_p, P, U]),
        "dcp_gpu_nseqs": (U, [P]),
        "dcp_gpu_seqs_set_xtrans": (I, [P, P, U]),
        "dcp_gpu_scan": (I, [P, C.POINTER(ScanParams)]),
        "dcp_gpu_sync": (I, [P]),
        "dcp_gpu_last_scan_redo_pairs": (I, [P, C.POINTER(U)]),
        "dcp_gpu_last_scan_ms": (F, [P]),
        "dcp_gpu_last_scan_launches": (U, [P]),
        "dcp_gpu_last_scan_kernel": (I, [P]),
        "dcp_gpu_fetch_scores": (I, [P, P, P]),
        "dcp_gpu_fetch_hits": (I, [P, P, U, C.POINTER(U)]),
        "dcp_gpu_scan_range": (I, [P, C.POINTER(ScanParams), U, U]),
        "dcp_gpu_set_hit_buffer": (I, [P, P, U, P]),
        "dcp_gpu_last_scan_launch_info": (I, [P, U, C.POINTER(LaunchInfo)]),
        "dcp_gpu_trace_paths": (I, [P, P, U, I, I, I, P, U, P, P]),
        "dcp_state_name": (U, [U, C.c_char_p]),
        "dcp_profile_decode": (I, [P, P, U, U, P]),
        "dcp_gc_decode": (C.c_char, [P]),
        "dcp_prod_format_row": (C.c_long, [C.c_char_p, C.c_size_t, C.c_int64, C.c_int64, C.c_char_p, C.c_char_p,
                                           C.c_double, C.c_double, C.c_char_p, C.c_char_p, P, P, U, P, U]),
        "dcp_prod_header": (C.c_char_p, []),
        "dcp_h3reader_open": (P, [C.c_char_p, I, F]),
        "dcp_h3reader_next": (I, [P, C.POINTER(P)]),
        "dcp_h3reader_error": (C.c_char_p, [P]),
        "dcp_h3reader_close": (None, [P]),
        "dcp_swissprot_null_lprobs": (None, [P]),
        "dcp_profile_consensus": (C.c_char_p, [P]),
        "dcp_db_write": (I, [C.c_char_p, P, U]),
        "dcp_db_open": (P, [C.c_char_p, C.POINTER(I)]),
        "dcp_db_close": (None, [P]),
        "dcp_db_nprofiles": (U, [P]),
        "dcp_db_entry_dist": (I, [P]),
        "dcp_db_epsilon": (F, [P]),
        "dcp_db_profile_sizes": (C.POINTER(C.c_uint32), [P]),
        "dcp_db_partitions": (U, [P, U, P, P]),
        "dcp_db_read": (I, [P, U, U, P]),
        "dcp_gpu_scan_cells": (C.c_uint64, [P]),
        "dcp_gpu_scan_algorithmic_bytes": (C.c_uint64, [P]),
    }
    if hooks:
        sig["dcp_gpu_test_set_redo_cap"] = (I, [P, U])
        sig["dcp_gpu_test_set_rowsweep_variant"] = (I, [P, I, U])
        sig["dcp_gpu_test_set_ring_stall"] = (I, [P, I])
        sig["dcp_gpu_test_set_seg_col_bytes"] = (I, [P, C.c_ulonglong])
        sig["dcp_gpu_test_set_trace_mode"] = (I, [P, I, C.c_ulonglong])
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
TESTHOOKS_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libdcp_hip_testhooks.so")


def load_testhooks():
    """TESTS ONLY: the -DDCP_TEST_HOOKS build of the same sources (csrc/Makefile), which adds
    dcp_gpu_test_set_redo_cap.  `Scanner(device, lib=load_testhooks())` runs a context of that build."""
    return _load(TESTHOOKS_LIB_PATH, hooks=True)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class ProteinCfg:
    """struct protein_cfg {entry_dist, epsilon} (include/deciphon/model/protein_cfg.h)."""

    def __init__(self, entry_dist=ENTRY_DIST_OCCUPANCY, epsilon=0.01):
        if not (0.0 <= epsilon <= 1.0):  # assert in protein_cfg():18
            raise DcpError(RC_EINVAL, "epsilon out of [0, 1]")
        self.entry_dist = entry_dist
        self.epsilon = float(np.float32(epsilon))


PROTEIN_CFG_DEFAULT = ProteinCfg(ENTRY_DIST_OCCUPANCY, 0.01)  # protein_cfg.h:22-23


class ProteinProfile:
    """The scan-time profile object (struct protein_profile, protein_profile.h:12-43) in its
    compact device-facing form: 8 transition rows + one 129-float nuclt_dist per node."""

    def __init__(self, handle):
        if not handle:
            raise DcpError(RC_EINVAL, "profile construction failed")
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.dcp_profile_del(h)

    @classmethod
    def sample(cls, seed, core_size, cfg=PROTEIN_CFG_DEFAULT, accession="accession"):
        """protein_profile_sample (src/model/protein_profile.c:259-304)."""
        rc = C.c_int(0)
        h = lib.dcp_profile_sample(accession.encode(), seed, core_size, cfg.entry_dist,
                                   cfg.epsilon, C.byref(rc))
        if not h:
            raise DcpError(rc.value, "protein_profile_sample")
        return cls(h)

    @classmethod
    def from_params(cls, null_lprobs, match_lprobs, trans, cfg=PROTEIN_CFG_DEFAULT,
                    accession="accession", consensus=None):
        """protein_model_init/setup/add_node/add_trans + protein_profile_absorb."""
        nl, ml, tr = _f32(null_lprobs), _f32(match_lprobs), _f32(trans)
        M = ml.shape[0] if ml.ndim == 2 else 0
        if nl.shape != (20,) or ml.shape != (M, 20) or tr.shape != (M + 1, 7):
            raise DcpError(RC_EINVAL, "bad parameter shapes")
        rc = C.c_int(0)
        cons = consensus.encode() if consensus else None
        h = lib.dcp_profile_new(accession.encode(), M, cfg.entry_dist, cfg.epsilon,
                                nl.ctypes.data, ml.ctypes.data, tr.ctypes.data, cons, C.byref(rc))
        if not h:
            raise DcpError(rc.value, "protein_model_setup")
        return cls(h)

    @property
    def core_size(self):
        return lib.dcp_profile_core_size(self._h)

    def decode(self, frag, state_id):
        """protein_profile_decode (src/model/protein_profile.c:306-331): most likely codon of a
        1..5-nt fragment emitted by `state_id`, as an ACGT string. RC_EINVAL for mute states."""
        f = encode_seq(frag) if isinstance(frag, str) else np.ascontiguousarray(frag, np.uint8)
        out = np.zeros(3, np.uint8)
        rc = lib.dcp_profile_decode(self._h, f.ctypes.data, len(f), int(state_id), out.ctypes.data)
        if rc:
            raise DcpError(rc, "failed to decode sequence")
        return "".join("ACGT"[b] for b in out)

    def prod_row(self, seq, steps, scan_id=0, seq_id=0, alt_loglik=0.0, null_loglik=0.0,
                 abc_name="dna", profile_typeid="protein", version="0.0.0", profile_name=None):
        """One product TSV row as prod_fwrite + protein_match_write_func write it
        (src/server/prod.c:153-181, src/server/protein_match.c:21-56)."""
        s = encode_seq(seq) if isinstance(seq, str) else np.ascontiguousarray(seq, np.uint8)
        st = np.ascontiguousarray(steps, STEP_DTYPE)
        cap = 256 + 64 * (len(st) + 1) + 2 * len(s)
        buf = C.create_string_buffer(cap)
        n = lib.dcp_prod_format_row(buf, cap, scan_id, seq_id, (profile_name or self.accession).encode(),
                                    abc_name.encode(), float(alt_loglik), float(null_loglik),
                                    profile_typeid.encode(), version.encode(), self._h, s.ctypes.data,
                                    len(s), st.ctypes.data, len(st))
        if n < 0:
            raise DcpError(RC_EFAIL, "failed to write prod")
        return buf.raw[:n].decode()

    @property
    def accession(self):
        return lib.dcp_profile_accession(self._h).decode()

    def _view(self, ptr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(n,)).reshape(shape).copy()

    @property
    def consensus(self):
        return lib.dcp_profile_consensus(self._h).decode()

    @property
    def trans8(self):
        return self._view(lib.dcp_profile_trans8(self._h), (8, self.core_size))

    @property
    def null_dist(self):
        return self._view(lib.dcp_profile_null_dist(self._h), (NDIST,))

    @property
    def insert_dist(self):
        return self._view(lib.dcp_profile_insert_dist(self._h), (NDIST,))

    @property
    def match_dist(self):
        return self._view(lib.dcp_profile_match_dist(self._h), (self.core_size, NDIST))


def state_name(state_id):
    """protein_state_name (src/model/protein_state.c:5-39)."""
    b = C.create_string_buffer(8)
    lib.dcp_state_name(int(state_id), b)
    return b.value.decode()


def gc_decode(codon):
    """imm_gc_decode(1, codon): amino acid letter of an ACGT codon string."""
    c = encode_seq(codon)
    return lib.dcp_gc_decode(c.ctypes.data).decode()


PROD_HEADER = lib.dcp_prod_header().decode()


def swissprot_null_lprobs():
    """Swiss-Prot 50.8 background (src/model/protein_h3reader.c:79-103), log-probabilities."""
    out = np.zeros(20, np.float32)
    lib.dcp_swissprot_null_lprobs(out.ctypes.data)
    return out


def read_hmmer3(path, cfg=PROTEIN_CFG_DEFAULT):
    """All profiles of a HMMER3 ASCII file: protein_h3reader_next + protein_profile_absorb per
    profile, as hmm_press does (src/server/hmm.c:120-178)."""
    r = lib.dcp_h3reader_open(str(path).encode(), cfg.entry_dist, cfg.epsilon)
    if not r:
        raise DcpError(RC_EIO, f"failed to open {path}")
    out = []
    try:
        while True:
            h = C.c_void_p()
            rc = lib.dcp_h3reader_next(r, C.byref(h))
            if rc == RC_END:
                return out
            if rc:
                raise DcpError(rc, lib.dcp_h3reader_error(r).decode())
            out.append(ProteinProfile(h.value))
    finally:
        lib.dcp_h3reader_close(r)


def write_db(path, profiles):
    """Write a dcpx profile DB (the press output: protein_db_writer_pack_profile per profile)."""
    arr = (C.c_void_p * len(profiles))(*[p._h for p in profiles])
    rc = lib.dcp_db_write(str(path).encode(), arr, len(profiles))
    if rc:
        raise DcpError(rc, f"failed to write {path}")


class ProfileDB:
    """protein_db_reader + profile_reader over a dcpx file: header fields, per-profile byte sizes,
    the reference's partition table, and reading a range of profiles."""

    def __init__(self, path):
        rc = C.c_int(0)
        self._h = lib.dcp_db_open(str(path).encode(), C.byref(rc))
        if not self._h:
            raise DcpError(rc.value, f"failed to open {path}")

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.dcp_db_close(h)

    __del__ = close

    @property
    def nprofiles(self):
        return lib.dcp_db_nprofiles(self._h)

    @property
    def cfg(self):
        return ProteinCfg(lib.dcp_db_entry_dist(self._h), lib.dcp_db_epsilon(self._h))

    @property
    def profile_sizes(self):
        return np.ctypeslib.as_array(lib.dcp_db_profile_sizes(self._h), shape=(self.nprofiles,)).copy()

    def partitions(self, npartitions):
        """(partition_size[npart], partition_offset[npart + 1]) as profile_reader_setup computes them."""
        sizes = np.zeros(NUM_THREADS, np.uint32)
        offs = np.zeros(NUM_THREADS + 1, np.int64)
        n = lib.dcp_db_partitions(self._h, npartitions, sizes.ctypes.data, offs.ctypes.data)
        if n == 0:
            raise DcpError(RC_EINVAL, "can't have zero partitions / too many partitions")
        return sizes[:n].tolist(), offs[:n + 1].tolist()

    def read(self, begin=0, end=None):
        end = self.nprofiles if end is None else end
        out = (C.c_void_p * max(end - begin, 1))()
        rc = lib.dcp_db_read(self._h, begin, end, out)
        profs = [ProteinProfile(h) for h in out[:end - begin] if h]
        if rc:
            raise DcpError(rc, "failed to read profiles")
        return profs


def frame_table_host(dist, epsilon):
    d = _f32(dist)
    out = np.zeros(NCODES, np.float32)
    lib.dcp_frame_table_host(d.ctypes.data, float(np.float32(epsilon)), out.ctypes.data)
    return out


def xtrans(seq_size, multi_hits=True, hmmer3_compat=False):
    """protein_profile_setup's 13 special transitions (protein_profile.c:155-216).
    Raises RC_EINVAL for an empty sequence, as the reference returns."""
    out = np.zeros(NXTRANS, np.float32)
    rc = lib.dcp_xtrans(seq_size, int(multi_hits), int(hmmer3_compat), out.ctypes.data)
    if rc:
        raise DcpError(rc, "sequence cannot be empty")
    return out


def lrt(null_loglik, alt_loglik):
    """xmath_lrt_f32 (include/deciphon/core/xmath.h:32-35)."""
    return lib.dcp_lrt(float(null_loglik), float(alt_loglik))


def partition_by_count(nprofiles, npartitions):
    """profile_reader partition sizes (src/db/profile_reader.c:54-72)."""
    sizes = np.zeros(NUM_THREADS, np.uint32)
    n = lib.dcp_partition_by_count(nprofiles, npartitions, sizes.ctypes.data)
    if n == 0:
        raise DcpError(RC_EINVAL, "can't have zero partitions / too many partitions")
    return sizes[:n].tolist()


def partition_by_cells(core_sizes, npartitions):
    """Contiguous shards balanced by sum of core sizes: one per GPU."""
    cs = np.ascontiguousarray(core_sizes, np.uint32)
    out = np.zeros(npartitions + 1, np.uint32)
    lib.dcp_partition_by_cells(cs.ctypes.data, len(cs), npartitions, out.ctypes.data)
    return out.tolist()


def encode_seq(text):
    """ACGT text -> symbol ids (imm_dna_iupac order); anything else -> 255."""
    lut = np.full(256, 255, np.uint8)
    for i, ch in enumerate(b"ACGT"):
        lut[ch] = i
    return lut[np.frombuffer(text.encode() if isinstance(text, str) else text, np.uint8)]


def device_count():
    return lib.dcp_gpu_device_count()


class Scanner:
    """One device context = one reference "partition" (thread_run's unit, scan.c:239-249):
    holds a profile shard resident in HBM and scans sequence batches against it."""

    def __init__(self, device=0, lib=None):
        self._lib = lib or globals()["lib"]
        self._c = self._lib.dcp_gpu_ctx_new(device)
        if not self._c:
            raise DcpError(RC_EFAIL, f"no HIP device {device}: this engine has no CPU fallback")
        self._profiles = None

    def close(self):
        c, self._c = getattr(self, "_c", None), None
        if c:
            self._lib.dcp_gpu_ctx_del(c)

    __del__ = close

    def _check(self, rc):
        if rc:
            raise DcpError(rc, self._lib.dcp_gpu_last_error(self._c).decode())

    @property
    def stream(self):
        return self._lib.dcp_gpu_stream(self._c)

    def upload_db(self, profiles, expand_on_host=False, one_layout=False):
        """one_layout: DCP_DB_ONE_LAYOUT (include/dcp_gpu.h) -- no tile images, the query-lane kernels gather them."""
        arr = (C.c_void_p * len(profiles))(*[p._h for p in profiles])
        flags = (DB_EXPAND_ON_HOST if expand_on_host else 0) | (DB_ONE_LAYOUT if one_layout else 0)
        self._check(self._lib.dcp_gpu_db_upload(self._c, arr, len(profiles), flags))
        self._profiles = list(profiles)

    @property
    def one_layout(self):
        return bool(self._lib.dcp_gpu_db_one_layout(self._c))

    @property
    def table_bytes(self):
        return int(self._lib.dcp_gpu_db_table_bytes(self._c))

    @property
    def nprofiles(self):
        return self._lib.dcp_gpu_db_nprofiles(self._c)

    @property
    def nseqs(self):
        return self._lib.dcp_gpu_nseqs(self._c)

    def match_table(self, p):
        M = self._profiles[p].core_size
        out = np.zeros((NCODES, M), np.float32)
        self._check(self._lib.dcp_gpu_db_fetch_match_table(self._c, p, out.ctypes.data))
        return out

    def upload_seqs(self, seqs):
        """seqs: list of ACGT strings, or of uint8 arrays / bytes of symbol ids 0..3."""
        if len(seqs) and isinstance(seqs[0], str):
            seqs = [encode_seq(s) for s in seqs]
        arrs = [np.frombuffer(s, np.uint8) if isinstance(s, (bytes, bytearray)) else
                np.ascontiguousarray(s, np.uint8) for s in seqs]
        off = np.zeros(len(arrs) + 1, np.uint32)
        if arrs:
            off[1:] = np.cumsum([len(a) for a in arrs])
        cat = np.ascontiguousarray(np.concatenate(arrs)) if arrs else np.zeros(0, np.uint8)
        self.upload_seqs_flat(cat, off)

    def upload_seqs_flat(self, cat, off):
        cat = np.ascontiguousarray(cat, np.uint8)
        off = np.ascontiguousarray(off, np.uint32)
        self._check(self._lib.dcp_gpu_seqs_upload(self._c, cat.ctypes.data, off.ctypes.data, len(off) - 1))
        self._seq_lens = np.diff(off.astype(np.int64))

    def set_xtrans(self, xt):
        """Explicit special transitions [nseqs, 13] for the resident sequences (dcp_gpu_seqs_set_xtrans):
        what imm_dp_viterbi uses for a profile whose transitions were not set from the sequence length."""
        xt = np.ascontiguousarray(xt, np.float32).reshape(-1, NXTRANS)
        self._check(self._lib.dcp_gpu_seqs_set_xtrans(self._c, xt.ctypes.data, len(xt)))

    def scan(self, multi_hits=True, hmmer3_compat=False, lrt_threshold=10.0, keep_scores=True,
             sync=True, q_range=None, kernel=KERNEL_AUTO):
        prm = ScanParams(int(multi_hits), int(hmmer3_compat), float(lrt_threshold), int(keep_scores),
                         int(kernel))
        if q_range is None:
            self._check(self._lib.dcp_gpu_scan(self._c, C.byref(prm)))
        else:
            self._check(self._lib.dcp_gpu_scan_range(self._c, C.byref(prm), q_range[0], q_range[1]))
        if sync:
            self.sync()

    def set_hit_buffer(self, hits_dev_ptr, cap, nhits_dev_ptr):
        """Route hit records into caller-owned device memory (e.g. a torch tensor for RCCL)."""
        self._check(self._lib.dcp_gpu_set_hit_buffer(self._c, hits_dev_ptr, cap, nhits_dev_ptr))

    def launch_infos(self):
        out = []
        for i in range(self.last_scan_launches):
            li = LaunchInfo()
            self._check(self._lib.dcp_gpu_last_scan_launch_info(self._c, i, C.byref(li)))
            out.append(dict(R=li.nodes_per_lane, W=li.waves_per_pair, nprofiles=li.nprofiles,
                            ms=li.ms, cells=li.cells, algorithmic_bytes=li.algorithmic_bytes))
        return out

    def sync(self):
        self._check(self._lib.dcp_gpu_sync(self._c))

    def test_set_redo_cap(self, cap):
        """TEST-ONLY, and only on a Scanner of the test-hooks build (load_testhooks): shrink the redo lists
        (0 restores 2^26) to reach the overflow path."""
        self._check(self._lib.dcp_gpu_test_set_redo_cap(self._c, int(cap)))

    def test_set_ring_stall(self, on):
        """TEST-ONLY (test-hooks build): the next two-stage query-lane scans stall one stage of the first task, so its
        partner runs into the ring hand-shake's poll bound (the scan must fail with RC_EFAIL, not hang)."""
        self._check(self._lib.dcp_gpu_test_set_ring_stall(self._c, int(bool(on))))

    def test_set_seg_col_bytes(self, nbytes):
        """TEST-ONLY (test-hooks build): cap on a size class's boundary columns in the segmented row sweep (0: default)."""
        self._check(self._lib.dcp_gpu_test_set_seg_col_bytes(self._c, int(nbytes)))

    def test_set_trace_mode(self, own_forward, budget_floats=0):
        """TEST-ONLY (test-hooks build): trace_paths' forward pass by the trace kernel's own loop (1) or the row-sweep
        kernels (0, the default); budget_floats: work area per round of launches (0: default)."""
        self._check(self._lib.dcp_gpu_test_set_trace_mode(self._c, int(bool(own_forward)), int(budget_floats)))

    def test_set_rowsweep_variant(self, stage_rows, block_waves=0):
        """TEST-ONLY (test-hooks build): force the grid-mode row-sweep kernel variant; stage_rows < 0: automatic."""
        self._check(self._lib.dcp_gpu_test_set_rowsweep_variant(self._c, int(stage_rows), int(block_waves)))

    @property
    def last_scan_ms(self):
        return self._lib.dcp_gpu_last_scan_ms(self._c)

    @property
    def last_scan_kernel(self):
        """KERNEL_ROWSWEEP / KERNEL_QLANE / KERNEL_QLANE2: what the last scan ran with."""
        return self._lib.dcp_gpu_last_scan_kernel(self._c)

    @property
    def last_scan_launches(self):
        return self._lib.dcp_gpu_last_scan_launches(self._c)

    @property
    def last_scan_redo_pairs(self):
        """Pairs of the last query-lane scan that the row-sweep kernel re-scored (synchronises)."""
        n = C.c_uint(0)
        self._check(self._lib.dcp_gpu_last_scan_redo_pairs(self._c, C.byref(n)))
        return n.value

    @property
    def cells(self):
        return self._lib.dcp_gpu_scan_cells(self._c)

    @property
    def algorithmic_bytes(self):
        return self._lib.dcp_gpu_scan_algorithmic_bytes(self._c)

    def scores(self):
        """(null[nseqs, nprofiles], alt[nseqs, nprofiles]) of the last scan."""
        nl = np.zeros((self.nseqs, self.nprofiles), np.float32)
        al = np.zeros_like(nl)
        self._check(self._lib.dcp_gpu_fetch_scores(self._c, nl.ctypes.data, al.ctypes.data))
        return nl, al

    def trace_paths(self, hits, multi_hits=True, hmmer3_compat=False, null_model=False):
        """Viterbi paths (alt model, or null model if null_model) of the given hit records, computed
        on the device: a list of STEP_DTYPE arrays, plus the log-likelihoods the trace recomputed."""
        h = np.ascontiguousarray(hits, HIT_DTYPE)
        n = len(h)
        off = np.zeros(n + 1, np.uint32)
        alt = np.zeros(n, np.float32)
        cap = int(sum(2 * int(self._seq_lens[q]) + 2 * self._profiles[p].core_size + 16
                      for q, p in zip(h["seq_idx"], h["profile_idx"]))) if n else 0
        steps = np.zeros(max(cap, 1), STEP_DTYPE)
        self._check(self._lib.dcp_gpu_trace_paths(self._c, h.ctypes.data, n, int(multi_hits), int(hmmer3_compat),
                                            int(null_model), steps.ctypes.data, cap, off.ctypes.data,
                                            alt.ctypes.data))
        return [steps[off[i]:off[i + 1]].copy() for i in range(n)], alt

    def hits(self, cap=1 << 20):
        buf = np.zeros(cap, HIT_DTYPE)
        n = C.c_uint(0)
        rc = self._lib.dcp_gpu_fetch_hits(self._c, buf.ctypes.data, cap, C.byref(n))
        if rc == RC_ENOMEM and n.value > cap:
            return self.hits(n.value)
        self._check(rc)
        return buf[:n.value]
