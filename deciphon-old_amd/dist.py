"""Multi-GPU layer: profiles shard across ranks, hits are gathered.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests).  Pairs (profile, query) are independent, so the data path needs NO collective: every
rank keeps its contiguous profile shard resident (balanced by sum of core sizes = DP cells, where
the reference's profile_reader balances by count, src/db/profile_reader.c:54-72) and scans ALL
queries.  The only exchange is the final hit gather: tiny records over xGMI.
"""
import numpy as np

from . import HIT_DTYPE, partition_by_cells

HIT_WORDS = 4  # struct dcp_hit = 4 x 32-bit words


def shard_range(core_sizes, world_size, rank):
    """[begin, end) of rank's contiguous profile shard."""
    b = partition_by_cells(np.asarray(core_sizes, np.uint32), world_size)
    return int(b[rank]), int(b[rank + 1])


def hits_from_words(words):
    """[n, 4] int32 words -> structured hit records."""
    w = np.ascontiguousarray(words, dtype=np.int32)
    return w.view(HIT_DTYPE).reshape(-1)


def gather_hits(hit_words, hit_count, profile_offset, slab=4096, group=None):
    """All-gather every rank's hit records.

    hit_words: int32 tensor [cap, 4] on this rank's device (what the scan kernels wrote through
    dcp_gpu_set_hit_buffer); hit_count: int32 tensor [1]; profile_offset: first global profile
    index of this rank's shard (records carry shard-local indices).
    Returns all ranks' hits as a HIT_DTYPE array sorted by (seq_idx, profile_idx), identical on
    every rank.  Two small collectives: counts, then fixed-size slabs (re-sized if a rank holds
    more than `slab` hits)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    counts = [torch.zeros_like(hit_count) for _ in range(world)]
    dist.all_gather(counts, hit_count, group=group)
    ns = [int(c.item()) for c in counts]
    cap = hit_words.shape[0]
    if max(ns) > cap:
        raise RuntimeError(f"hit buffer overflow: {max(ns)} > {cap}")
    rows = max(1, min(cap, max(slab, max(ns))))
    mine = hit_words[:rows].clone()
    n_mine = int(hit_count.item())
    if profile_offset:
        mine[:n_mine, 1] += profile_offset
    slabs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(slabs, mine, group=group)
    parts = [hits_from_words(s[:n].cpu().numpy()) for s, n in zip(slabs, ns) if n]
    if not parts:
        return np.zeros(0, HIT_DTYPE)
    allh = np.concatenate(parts)
    return allh[np.lexsort((allh["profile_idx"], allh["seq_idx"]))]
