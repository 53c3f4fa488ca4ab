#!/usr/bin/env python3
"""Developer: ratio and encode speed per (method, level, class) on 1 MiB entries.  tools/enc_levels.py [entries per class]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zpack_amd
from benchdata import datagen as dg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
size = 1 << 20
dev = torch.device("cuda:0")
codec = zpack_amd.Codec(0)
for cls, cname in [(0, "text"), (1, "records")]:
    plain = np.empty(n * size, dtype=np.uint8)
    for i in range(n):
        plain[i * size:(i + 1) * size] = dg.fill(cls, 5, i, size)
    src = torch.from_numpy(plain).to(dev)
    for method, mname in [(dg.ZSTD, "zstd"), (dg.LZ4, "lz4")]:
        for level in ([1, 3, 6, 19] if method == dg.ZSTD else [0, 3, 9]):
            bound = codec.compress_bound(method, size); slot = (bound + 255) & ~255
            desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
            desc["src_offset"] = np.arange(n, dtype=np.uint64) * size; desc["size"] = size
            desc["dst_offset"] = np.arange(n, dtype=np.uint64) * slot; desc["dst_capacity"] = bound
            desc["method"] = method; desc["level"] = level
            ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
            slots = torch.empty(n * slot, dtype=torch.uint8, device=dev)
            dres = torch.zeros(n * zpack_amd.ENCODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.time()
                codec.encode_batch_device(src, ddesc, n, slots, dres)
                torch.cuda.synchronize(); dt = time.time() - t0
            res = dres.cpu().numpy().view(zpack_amd.ENCODE_RESULT)
            print("%-8s %-5s level %2d  ratio %.4f  %.1f GiB/s of source  status ok %s" % (
                cname, mname, level, res["comp_size"].sum() / (n * size), n * size / dt / 2**30, bool((res["status"] == 0).all())), flush=True)
