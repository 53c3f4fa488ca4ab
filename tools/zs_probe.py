#!/usr/bin/env python3
"""Developer probe: one small Zstandard batch through the two-stage decode, with the counters printed.
   zs_probe.py N SIZE LEVEL [MIX]      (run with ZPK_TRACE=1 to see which kernel a hang is in)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg
from tests._libs import oracle

n, size, level = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mix = int(sys.argv[4]) if len(sys.argv) > 4 else -1
b = dg.Batch(n, size, method=dg.ZSTD, level=level, seed=11, mix=mix)
desc, total = zpack_amd.decode_descs_from_batch(b, flags=zpack_amd.DF_SKIP_HASH if os.environ.get("SKIP_HASH") else 0)
codec = zpack_amd.Codec(0)
dev = torch.device("cuda:0")
src = torch.from_numpy(b.archive).to(dev)
dst = torch.zeros(total, dtype=torch.uint8, device=dev)
ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
print("launch", flush=True)
codec.set_profiling(True)
for it in range(3):
    t0 = time.time()
    codec.decode_batch_device(src, ddesc, n, dst, dres)
    torch.cuda.synchronize()
    print("iter", it, "wall %.2f ms" % ((time.time() - t0) * 1e3), "fse %.3f ms  zstd %.3f ms" %
          (codec.kernel_ms(zpack_amd.K_ZSTD_FSE), codec.kernel_ms(zpack_amd.K_ZSTD)), codec.decode_stats(), flush=True)
res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
out = dst.cpu().numpy()
print("status ok:", int((res["status"] == 0).sum()), "/", n, "hash ok:", bool(np.array_equal(res["hash"], b.hashes)))
o = oracle()
arc = b.archive.tobytes()
badn = 0
for i in range(min(n, 64)):
    d = desc[i]
    rc, want, got, h = o.entry_decode(arc, int(d["src_offset"]), int(d["comp_size"]), int(d["uncomp_size"]),
                                      int(d["expect_hash"]), int(d["method"]), int(d["dst_capacity"]))
    if out[int(d["dst_offset"]):int(d["dst_offset"] + d["uncomp_size"])].tobytes() != want:
        badn += 1
print("byte mismatches in sample:", badn)
tot = float(b.total_uncomp)
print("GiB/s (kernels): %.1f" % (tot / ((codec.kernel_ms(zpack_amd.K_ZSTD_FSE) + codec.kernel_ms(zpack_amd.K_ZSTD)) * 1e-3) / 2**30))

# ---- sequence-level check of the pre-decode kernel against the oracle ----
marks = codec.debug_fetch(1, 0, n, np.uint32)
print("marks:", marks[:32])
for i in range(min(n, 8)):
    d = desc[i]
    frame = arc[int(d["src_offset"]):int(d["src_offset"] + d["comp_size"])]
    rc, seqs = o.zstd_sequences(frame, int(d["uncomp_size"]))
    base = (int(d["dst_offset"]) + 7) & ~7
    got = codec.debug_fetch(0, base, max(1, len(seqs)), np.uint64)
    neq = np.nonzero(got[:len(seqs)] != seqs)[0]
    def unp(v):
        v = int(v); return (v & ((1 << 29) - 1), (v >> 29) & ((1 << 18) - 1), v >> 47)
    print("entry", i, "mark", marks[i], "oracle rc", rc, "nseq", len(seqs), "first mismatch", (int(neq[0]) if len(neq) else None),
          "mismatches", len(neq))
    if len(neq) and os.environ.get("ZPK_ZF_DEBUG"):
        dbg = codec.debug_fetch(2, base, len(seqs), np.uint64)
        k = int(neq[0])
        for j in range(max(1, k - 6), min(len(seqs), k + 3)):
            print("    step", j, "gpu pos", int(dbg[j] >> 32) - (1 << 32 if (int(dbg[j]) >> 63) else 0), "loaded_lo", int(dbg[j]) & 0xFFFF,
                  "tick", (int(dbg[j]) >> 16) & 0xFFFF, " oracle pos before step", int(o.last_trace_bits[j - 1]))
    if len(neq):
        k = int(neq[0])
        for j in range(max(0, k - 2), min(len(seqs), k + 4)):
            print("   ", j, "want", unp(seqs[j]), "got", unp(got[j]))
