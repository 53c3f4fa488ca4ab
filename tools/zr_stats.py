#!/usr/bin/env python3
"""Developer: per-phase cycles of the Zstandard RING executor (needs a -DZPK_DEVELOPER -DZSTD_EXEC_RING -DLX_STATS -DLX_STATS_SCAN_ONLY=0
build selected with ZPACK_AMD_CODEC_SO, and ZPK_DEBUG_TIMING=1).  tools/zr_stats.py [entries] [mix]"""
import os, sys
import numpy as np
os.environ["ZPK_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zpack_amd
from benchdata import datagen as dg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mix = int(sys.argv[2]) if len(sys.argv) > 2 else 0
b = dg.Batch(n, 262144, 262144, method=dg.ZSTD, level=3, seed=2, mix=mix)
desc, total = zpack_amd.decode_descs_from_batch(b)
dev = torch.device("cuda:0")
codec = zpack_amd.Codec(0)
src = torch.from_numpy(b.archive).to(dev); dst = torch.empty(total, dtype=torch.uint8, device=dev)
ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev); dres = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
for _ in range(3):
    codec.decode_batch_device(src, ddesc, n, dst, dres)
torch.cuda.synchronize()
a = np.zeros((n, 16), dtype=np.uint64)
codec._chk(codec.L.zpk_codec_debug_read(codec.h, a.ctypes.data, a.nbytes), "debug_read")
names = ["seq load+scan", "literals section", "-", "positions", "deps", "lit copy", "match copy", "rounds", "flush+hash", "headers/other", "finish", "-", "TOTAL"]
m = a[:, :13].astype(np.float64).mean(0)
print("mean cycles per entry (mix %d, %d entries):" % (mix, n))
for k, v in zip(names, m):
    print("  %-16s %10.0f  %5.1f %%" % (k, v, 100 * v / m[12]))
