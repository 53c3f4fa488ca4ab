#!/usr/bin/env python3
"""Developer: per-phase cycles of the general LZ4 decoder k_lz4_wave (needs a -DZPK_DEVELOPER -DZPK_STATS build selected with
ZPACK_AMD_CODEC_SO, and ZPK_DEBUG_TIMING=1).  tools/lw_stats.py [entries] [mix]"""
import os, sys
import numpy as np
os.environ["ZPK_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zpack_amd
from benchdata import datagen as dg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
mix = int(sys.argv[2]) if len(sys.argv) > 2 else 0
b = dg.Batch(n, 65536, 65536, method=dg.LZ4, level=0, seed=1, mix=mix)
desc, total = zpack_amd.decode_descs_from_batch(b)
dev = torch.device("cuda:0")
codec = zpack_amd.Codec(0)
src = torch.from_numpy(b.archive).to(dev); dst = torch.empty(total, dtype=torch.uint8, device=dev)
ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev); dres = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
for _ in range(3):
    codec.decode_batch_device(src, ddesc, n, dst, dres)
torch.cuda.synchronize()
a = np.zeros((n, 8), dtype=np.uint64)
codec._chk(codec.L.zpk_codec_debug_read(codec.h, a.ctypes.data, a.nbytes), "debug_read")
m = a.astype(np.float64).mean(0)
print("mean memtime ticks per entry (mix %d, %d entries): total %.0f" % (mix, n, m[6]))
for k, v in zip(["parse", "literals", "deps", "rounds"], m[:4]):
    print("  %-9s %10.0f  %5.1f %%" % (k, v, 100 * v / m[6]))
print("  other (stores, hash, frame) %.1f %%" % (100 * (m[6] - m[:4].sum()) / m[6]))
print("  batches/entry %.1f rounds/entry %.1f" % ((a[:, 4] >> 32).mean(), (a[:, 4] & 0xFFFFFFFF).mean()))
if os.environ.get("LW_PARSE"):      # a -DZPK_STATS_PARSE build: words 0..4 = stage / first walk / fix-up / emit / token fetch, 5 = hop iterations
    ch = (a[:, 7] & 0xFFFFFFFF).astype(np.float64); fi = (a[:, 7] >> 32).astype(np.float64)
    print("  parse phases (ticks/entry): stage %.0f walk1 %.0f fix %.0f emit %.0f tok %.0f" % tuple(m[:5]))
    print("  chunks/entry %.2f fix rounds/chunk %.2f hop iterations per chunk: first walk %.1f, fix-up %.1f; slow-path iterations/chunk %.2f" % (
        ch.mean(), fi.sum() / ch.sum(), (a[:, 5] >> 32).sum() / ch.sum(), (a[:, 5] & 0xFFFFFFFF).sum() / ch.sum(), a[:, 6].sum() / ch.sum()))
