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


def _coll_device(dist):
    """tensor collectives run on the backend's device: the current GPU under nccl (= RCCL), the CPU under gloo"""
    import torch
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_results(local, lo, hi, n, rank, world, dist=None):
    """Concatenate per-rank result arrays (numpy structured or plain: 24-byte results, hashes, sizes — bookkeeping, never entry
    payloads) in CDR order on rank 0; no collective on the data path.  Tensor collectives only — an all_gather of the (lo, hi) ranges
    (16 bytes per rank) and ONE gather to rank 0 of the byte images padded to the longest slice, so only rank 0 holds world x longest
    bytes: the same code under gloo on the CPU and under RCCL on the GPUs, no pickling through object collectives."""
    if world == 1 or dist is None:
        return local
    import torch
    dev = _coll_device(dist)
    mine = torch.tensor([int(lo), int(hi)], dtype=torch.int64, device=dev)
    ranges = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ranges, mine)
    ranges = [(int(r[0]), int(r[1])) for r in ranges]
    item = local.dtype.itemsize
    longest = max(h - l for l, h in ranges) * item
    buf = torch.zeros(max(longest, 1), dtype=torch.uint8, device=dev)
    raw = np.frombuffer(np.ascontiguousarray(local).tobytes(), dtype=np.uint8)
    if raw.size:
        buf[:raw.size] = torch.from_numpy(raw.copy()).to(dev)
    parts = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gather_list=parts, dst=0)
    if rank != 0:
        return None
    out = np.zeros(n, dtype=local.dtype)
    for (l, h), p in zip(ranges, parts):
        if h > l:
            out[l:h] = np.frombuffer(p[:(h - l) * item].cpu().numpy().tobytes(), dtype=local.dtype)
    return out


def archive_bases(segment_totals, data_start=10):
    """Write path over several GPUs: every rank compacts its own entries into one packed segment (the per-GPU size
    scan, zpk_codec_pack_batch_device); the segments follow each other in rank order in the archive's data section, so rank r's base
    is data_start + the totals of the ranks before it — the serial `write_offset += comp_size` of lib/zpack_write.c:338 across
    ranks, from ONE host scan of `world` numbers.  -> (bases[world], end of the data section)"""
    t = np.asarray(segment_totals, dtype=np.uint64)
    bases = np.uint64(data_start) + np.concatenate([[np.uint64(0)], np.cumsum(t)[:-1]]).astype(np.uint64)
    return bases, int(data_start + int(t.sum()))


def gather_segment_totals(total, rank, world, dist=None):
    """every rank learns every rank's packed-segment length (8 bytes per rank; bookkeeping, not data)"""
    if world == 1 or dist is None:
        return [int(total)]
    import torch
    dev = _coll_device(dist)
    out = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(out, torch.tensor([int(total)], dtype=torch.int64, device=dev))
    return [int(x[0]) for x in out]
