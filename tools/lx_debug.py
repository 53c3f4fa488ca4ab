import sys, os, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zpack_amd
from benchdata import datagen as dg
b = dg.Batch(12, 1 << 20, 1 << 20, method=dg.LZ4, level=0, seed=23, mix=dg.TEXT)
desc, total = zpack_amd.decode_descs_from_batch(b, flags=1)
dev = torch.device("cuda:0"); codec = zpack_amd.Codec(0)
src = torch.from_numpy(b.archive).to(dev); dst = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev); dres = torch.zeros(b.n * 24, dtype=torch.uint8, device=dev)
codec.decode_batch_device(src, ddesc, b.n, dst, dres); torch.cuda.synchronize()
st = codec.decode_stats(); print(st)
meta = codec.debug_fetch(3, 0, b.n, np.uint32); print("meta", meta)
nu = st["lz4_units"]
units = codec.debug_fetch(4, 0, nu * 4, np.uint32).reshape(nu, 4)
tok = codec.debug_fetch(5, 0, int(b.archive.size * 3 // 4), np.uint8)
def region(off): return ((off * 3) >> 4) << 2
bad = 0
for u in units:
    blk = int(u[0]) | (int(u[1]) << 32); bsz = int(u[2]) & 0xFFFFF; seg = int(u[2]) >> 20; e = int(u[3])
    lo = seg * 8192; hi = min(bsz, lo + 8192)
    r0, r1 = region(blk + lo), region(blk + hi)
    cnt, ex = struct.unpack_from("<II", tok, r0); pc, v = struct.unpack_from("<II", tok, r1 - 8)
    if meta[e] != 1 and bad < 12 and (cnt > 3000 or ex > bsz or ex < hi and ex != bsz):
        print("entry", e, "blk", blk, "bsz", bsz, "seg", seg, "count", cnt, "exit", ex, "patch", pc, "v", v); bad += 1
