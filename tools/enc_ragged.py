#!/usr/bin/env python3
"""Developer: a RAGGED encode batch (entry sizes log-uniform 4 KiB ... 1 MiB, the 70/20/5/5 class mix, device-resident) with the
ticket queue in archive order against largest entries first (ZPK_OPT_ORDER_MIN) — same frames, same hashes either way.
usage: enc_ragged.py [n=30000] [level=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
codec = zpack_amd.Codec(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(12)
sizes = np.exp(rng.uniform(np.log(4096), np.log(1 << 20), n)).astype(np.uint64)
# a pool of 256 MiB per class; an entry is a slice of its class's pool
pools, bases, at = [], [], 0
for cls in range(4):
    p = np.concatenate([dg.fill(cls, 9, i, 1 << 20) for i in range(64)])
    p = np.tile(p, 4)
    pools.append(p); bases.append(at); at += len(p)
src = torch.from_numpy(np.concatenate(pools)).to(dev)
cls_of = rng.choice(4, n, p=[0.7, 0.2, 0.05, 0.05])
offs = np.array([bases[c] + int(rng.integers(0, len(pools[c]) - int(s))) for c, s in zip(cls_of, sizes)], dtype=np.uint64)
for method, mname in ((zpack_amd.METHOD_ZSTD, "zstd-%d" % level), (zpack_amd.METHOD_LZ4, "lz4")):
    bounds = np.array([codec.compress_bound(method, int(s)) for s in sizes], dtype=np.uint64)
    slots = (bounds + 255) & ~np.uint64(255)
    desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
    desc["src_offset"] = offs; desc["size"] = sizes
    desc["dst_offset"] = np.concatenate([[0], np.cumsum(slots)[:-1]]).astype(np.uint64)
    desc["dst_capacity"] = bounds; desc["method"] = method; desc["level"] = level
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dst = torch.empty(int(slots.sum()) + 64, dtype=torch.uint8, device=dev)
    out = {}
    for order_min, label in ((0, "archive order"), (1, "largest first")):
        codec.set_option(zpack_amd.OPT_ORDER_MIN, order_min)
        dres = torch.zeros(n * zpack_amd.ENCODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            codec.encode_batch_device(src, ddesc, n, dst, dres)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        res = dres.cpu().numpy().view(zpack_amd.ENCODE_RESULT).copy()
        assert (res["status"] == 0).all()
        out[label] = (best, res)
        print("%-7s %d entries 4 KiB..1 MiB (%.2f GiB) %s: %.1f ms = %.1f GiB/s, ratio %.4f" % (mname, n, sizes.sum() / 2**30, label, best * 1e3,
              sizes.sum() / 2**30 / best, res["comp_size"].sum() / sizes.sum()), flush=True)
    a, b = out["archive order"][1], out["largest first"][1]
    assert np.array_equal(a["comp_size"], b["comp_size"]) and np.array_equal(a["hash"], b["hash"])
