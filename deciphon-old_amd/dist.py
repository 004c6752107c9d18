"""Multi-GPU layer: profiles shard across ranks, hits are gathered.

One process per GPU.  Pairs (profile, query) are independent, so the data path needs NO collective:
every rank keeps its contiguous profile shard resident (balanced by sum of core sizes = DP cells,
where the reference's profile_reader balances by count, src/db/profile_reader.c:54-72) and scans ALL
queries.  The only exchange is the final hit gather: tiny records over xGMI.

The gather lives in C (csrc/dcp_dist.cpp, include/dcp_gpu.h "One process per GPU"): counts
all-gather + grouped ncclSend/ncclRecv over RCCL, and the bookkeeping (counts -> displacements ->
global profile indices -> (seq, profile) order) in dcp_dist_merge_hits.  `CDist` binds it.
`gather_hits` keeps a torch.distributed transport (gloo in the CPU tests, the fallback of bench.py)
around the SAME C bookkeeping.
"""
import ctypes as C

import numpy as np

from . import HIT_DTYPE, DcpError, RC_EFAIL, lib, partition_by_cells

HIT_WORDS = 4  # struct dcp_hit = 4 x 32-bit words
ID_BYTES = 128

def bind(lib):
    """ctypes prototypes of the dcp_dist_* entry points on a loaded library (the shipped one below; the tests' own
    -DDCP_TEST_HOOKS build through CDist.create(..., lib=...))."""
    lib.dcp_dist_unique_id.restype = C.c_int
    lib.dcp_dist_unique_id.argtypes = [C.c_void_p]
    lib.dcp_dist_init.restype = C.c_void_p
    lib.dcp_dist_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.dcp_dist_init_from_file.restype = C.c_void_p
    lib.dcp_dist_init_from_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.dcp_dist_free.restype = None
    lib.dcp_dist_free.argtypes = [C.c_void_p]
    lib.dcp_dist_last_error.restype = C.c_char_p
    lib.dcp_dist_last_error.argtypes = [C.c_void_p]
    lib.dcp_dist_shard.restype = None
    lib.dcp_dist_shard.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.dcp_dist_gather_hits.restype = C.c_int
    lib.dcp_dist_gather_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.c_void_p,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_uint)]
    lib.dcp_dist_gather_scan_hits.restype = C.c_int
    lib.dcp_dist_gather_scan_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_uint)]
    lib.dcp_dist_gather_plan.restype = C.c_int
    lib.dcp_dist_gather_plan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int),
                                         C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    lib.dcp_dist_init_from_file_run.restype = C.c_void_p
    lib.dcp_dist_init_from_file_run.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_double]
    lib.dcp_dist_comm_count.restype = C.c_int
    lib.dcp_dist_comm_count.argtypes = [C.c_void_p]
    lib.dcp_dist_last_gather_ms.restype = C.c_double
    lib.dcp_dist_last_gather_ms.argtypes = [C.c_void_p]
    lib.dcp_dist_free_hits.restype = None
    lib.dcp_dist_free_hits.argtypes = [C.c_void_p]
    lib.dcp_dist_merge_hits.restype = C.c_long
    lib.dcp_dist_merge_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint]

    return lib


bind(lib)

def shard_range(core_sizes, world_size, rank):
    """[begin, end) of rank's contiguous profile shard (dcp_dist_shard)."""
    cs = np.ascontiguousarray(core_sizes, np.uint32)
    b, e = C.c_uint(0), C.c_uint(0)
    lib.dcp_dist_shard(cs.ctypes.data, len(cs), world_size, rank, C.byref(b), C.byref(e))
    assert [b.value, e.value] == partition_by_cells(cs, world_size)[rank:rank + 2]
    return int(b.value), int(e.value)


def hits_from_words(words):
    """[n, 4] int32 words -> structured hit records."""
    w = np.ascontiguousarray(words, dtype=np.int32)
    return w.view(HIT_DTYPE).reshape(-1)


def merge_hits(counts, profile_offsets, records):
    """dcp_dist_merge_hits: counts[r] records of rank r lie back to back in `records` (shard-local
    profile indices) -> one list with global indices, ordered by (seq_idx, profile_idx)."""
    counts = np.ascontiguousarray(counts, np.uint32)
    offs = np.ascontiguousarray(profile_offsets, np.uint32)
    rec = np.ascontiguousarray(records, HIT_DTYPE)
    total = int(counts.sum())
    if len(rec) != total or len(offs) != len(counts):
        raise DcpError(RC_EFAIL, "inconsistent counts")
    out = np.zeros(max(total, 1), HIT_DTYPE)
    n = lib.dcp_dist_merge_hits(counts.ctypes.data, offs.ctypes.data, len(counts), rec.ctypes.data,
                                out.ctypes.data, total)
    if n != total:
        raise DcpError(RC_EFAIL, "dcp_dist_merge_hits failed")
    return out[:total]


META_WORDS = 3  # DCP_DIST_META_WORDS: records held, profile offset, records found
FOUND_FAILED = 0xFFFFFFFF  # DCP_DIST_FOUND_FAILED: "records found" of a rank whose scan failed


def gather_plan(meta, with_failed=False):
    """dcp_dist_gather_plan: meta [nranks, 3] uint32 -> (counts, offsets, displ[nranks+1] uint64, any_overflow, total)
    (+ any_failed as a sixth item when with_failed).  Raises DcpError(EINVAL) when the total exceeds 2^32 - 1 or a
    rank holds more than it found."""
    m = np.ascontiguousarray(meta, np.uint32).reshape(-1, META_WORDS)
    n = len(m)
    counts, offs, displ = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n + 1, np.uint64)
    ovf, failed, total = C.c_int(0), C.c_int(0), C.c_uint64(0)
    rc = lib.dcp_dist_gather_plan(m.ctypes.data, n, counts.ctypes.data, offs.ctypes.data, displ.ctypes.data,
                                  C.byref(ovf), C.byref(failed), C.byref(total))
    if rc:
        raise DcpError(rc, "dcp_dist_gather_plan")
    if with_failed:
        return counts, offs, displ, bool(ovf.value), int(total.value), bool(failed.value)
    if failed.value:
        raise DcpError(RC_EFAIL, "a rank's scan failed: its shard's hits are missing from the gathered list")
    return counts, offs, displ, bool(ovf.value), int(total.value)


class CDist:
    """The C host's RCCL communicator (dcp_dist_*): one per process / GPU."""

    def __init__(self, handle, rank, world, lib_=None):
        self._h, self.rank, self.world, self._lib = handle, rank, world, lib_ or lib

    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * ID_BYTES)()
        rc = lib.dcp_dist_unique_id(buf)
        if rc:
            raise DcpError(rc, "ncclGetUniqueId failed (is librccl.so loadable?)")
        return bytes(buf)

    @classmethod
    def create(cls, id_bytes, rank, world, device, lib_=None):
        """lib_: another build of the library (tests: load_testhooks()) whose contexts this communicator serves."""
        use = bind(lib_) if lib_ is not None else lib
        buf = (C.c_ubyte * ID_BYTES).from_buffer_copy(id_bytes)
        h = use.dcp_dist_init(buf, rank, world, device)
        if not h:
            raise DcpError(RC_EFAIL, "ncclCommInitRank failed")
        return cls(h, rank, world, use)

    def gather_scan_hits(self, scanner, profile_offset, root=-1):
        """dcp_dist_gather_scan_hits: completes `scanner`'s scan (dcp_gpu_sync: redo lists checked), then
        gathers the buffer it wrote.  Returns (records or None, global total)."""
        out, n = C.c_void_p(), C.c_uint(0)
        rc = self._lib.dcp_dist_gather_scan_hits(self._h, scanner._c, profile_offset, root, C.byref(out), C.byref(n))
        return self._take(rc, out, n)

    def gather_hits(self, hits_dev_ptr, count_dev_ptr, cap, profile_offset, scan_stream, root=-1):
        """All ranks call it after their scan is complete (Scanner.sync); returns (records or None, global total)."""
        out, n = C.c_void_p(), C.c_uint(0)
        rc = self._lib.dcp_dist_gather_hits(self._h, hits_dev_ptr, count_dev_ptr, cap, profile_offset, root, scan_stream,
                                      C.byref(out), C.byref(n))
        return self._take(rc, out, n)

    def _take(self, rc, out, n):
        if rc:
            if out.value:  # an overflow is reported after the exchange, with the (truncated) list allocated
                self._lib.dcp_dist_free_hits(out)
            raise DcpError(rc, self._lib.dcp_dist_last_error(self._h).decode())
        if not out.value:
            return None, n.value
        try:
            arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), shape=(max(n.value, 1) * HIT_WORDS,))
            return arr[:n.value * HIT_WORDS].copy().view(HIT_DTYPE), n.value
        finally:
            self._lib.dcp_dist_free_hits(out)

    @property
    def comm_count(self):
        """Ranks RCCL itself reports for the communicator (ncclCommCount)."""
        return self._lib.dcp_dist_comm_count(self._h)

    @property
    def last_gather_ms(self):
        return self._lib.dcp_dist_last_gather_ms(self._h)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.dcp_dist_free(h)

    __del__ = close


def gather_hits(hit_words, hit_count, profile_offset, slab=4096, group=None):
    """All-gather every rank's hit records over torch.distributed (gloo on CPU, RCCL as "nccl").

    hit_words: int32 tensor [cap, 4] on this rank's device (what the scan kernels wrote through
    dcp_gpu_set_hit_buffer); hit_count: int32 tensor [1]; profile_offset: first global profile
    index of this rank's shard (records carry shard-local indices).
    Returns all ranks' hits as a HIT_DTYPE array sorted by (seq_idx, profile_idx), identical on
    every rank.  Two small collectives -- {count, offset} of every rank, then fixed-size slabs
    (re-sized if a rank holds more than `slab` hits) -- and the C bookkeeping of merge_hits."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    meta = torch.stack([hit_count.reshape(()).to(torch.int32),
                        torch.tensor(profile_offset, dtype=torch.int32, device=hit_count.device)])
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = torch.stack(metas).cpu().numpy().astype(np.int64)
    cap = hit_words.shape[0]
    # the same decision as the C path (dcp_dist_gather_plan): every rank learns of any rank's overflow
    found = metas[:, 0]
    held = np.minimum(found, cap)  # all ranks use the same capacity in this transport
    ns, offs, _, overflow, _ = gather_plan(np.stack([held, metas[:, 1], found], axis=1))
    ns, offs = ns.astype(np.int64), offs.astype(np.int64)
    if overflow:
        raise RuntimeError(f"hit buffer overflow on some rank: found {found.max()} > capacity {cap}")
    rows = max(1, min(cap, max(slab, int(ns.max()))))
    mine = hit_words[:rows].contiguous()
    slabs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(slabs, mine, group=group)
    parts = [hits_from_words(s[:n].cpu().numpy()) for s, n in zip(slabs, ns) if n]
    records = np.concatenate(parts) if parts else np.zeros(0, HIT_DTYPE)
    return merge_hits(ns, offs, records)
