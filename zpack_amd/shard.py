"""Static sharding of an archive's entry table across the GPUs of one node.

Entries are independent (own frame, offset, size, hash: lib/zpack.h:71-80 of the reference), so a batch
splits into contiguous ranges of the CDR order, balanced by bytes moved (comp_size + uncomp_size), one
range per rank; there is no data-path collective — only the per-rank status / hash arrays are gathered.
"""
import numpy as np


def shard_ranges(comp_sizes, uncomp_sizes, world):
    """-> [(lo, hi)] * world, contiguous, covering [0, n), balanced by Σ(comp+uncomp) (host prefix sum)."""
    w = np.asarray(comp_sizes, dtype=np.uint64).astype(np.float64) + np.asarray(uncomp_sizes, dtype=np.uint64).astype(np.float64)
    n = len(w)
    if world <= 1 or n == 0:
        return [(0, n)] + [(n, n)] * (max(world, 1) - 1)
    csum = np.concatenate([[0.0], np.cumsum(w)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(csum, target, side="left"))
        k = min(max(k, cuts[-1]), n)
        cuts.append(k)
    cuts.append(n)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def gather_results(local, lo, hi, n, rank, world, dist=None):
    """Concatenate per-rank result arrays (numpy structured or plain) on rank 0; no collective on the data path."""
    if world == 1 or dist is None:
        return local
    import torch
    sizes = [None] * world
    dist.all_gather_object(sizes, (int(lo), int(hi)))
    out = None
    if rank == 0:
        out = np.zeros(n, dtype=local.dtype)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(local.tobytes(), parts, dst=0)
    if rank == 0:
        for (l, h), b in zip(sizes, parts):
            out[l:h] = np.frombuffer(b, dtype=local.dtype)
    return out
