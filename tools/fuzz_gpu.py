#!/usr/bin/env python3
"""Developer aid: a larger round of tests/test_gpu_codec.py::test_corrupted_frames_* — thousands of damaged frames per
method / level in one device batch (XXH3 verify off), verdict + bytes against the oracle.  usage: fuzz_gpu.py [per_base] [seed] [lz4|all] [level]
(third argument "lz4": only the LZ4 configurations)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import oracle

per = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
codec = zpack_amd.Codec(0)
if os.environ.get("ZPK_FUZZ_ORDER_MIN"):                       # 1: every batch runs its work lists largest entries first (default: batches of >= 8192 entries)
    codec.set_option(zpack_amd.OPT_ORDER_MIN, int(os.environ["ZPK_FUZZ_ORDER_MIN"]))
    print("work lists ordered from", os.environ["ZPK_FUZZ_ORDER_MIN"], "entries")
only_lz4 = len(sys.argv) > 3 and sys.argv[3] == "lz4"
o = oracle()
dev = torch.device("cuda:0")
only_level = int(sys.argv[4]) if len(sys.argv) > 4 else None
for method, level in (((dg.LZ4, 0), (dg.LZ4, 9)) if only_lz4 else ((dg.ZSTD, 3), (dg.ZSTD, 1), (dg.ZSTD, 19), (dg.LZ4, 0), (dg.LZ4, 9))):
    if only_level is not None and level != only_level:
        continue
    rng = np.random.default_rng(seed * 100 + level + method)
    frames, sizes = [], []
    for cls, size in ((dg.TEXT, 300000), (dg.RECORDS, 70000), (dg.RUNS, 150000), (dg.TEXT, 9000), (dg.RANDOM, 20000), (dg.TEXT, 700)):
        plain = dg.fill(cls, seed, 0, size)
        base = bytearray(dg.compress(method, level, plain))
        for k in range(per):
            f = bytearray(base)
            if k:
                hits = 1 + (k % 5 == 0) + (k % 9 == 0)
                for _ in range(hits):
                    f[int(rng.integers(0, len(f)))] ^= int(rng.integers(1, 256))
                if k % 11 == 0:
                    f = f[:int(rng.integers(1, len(f)))]
                if k % 17 == 0 and method == dg.LZ4:              # a second frame behind the first (the reference's LZ4F loop decodes on)
                    f = f + bytearray(dg.compress(method, level, plain[:1000]))
                if k % 13 == 0:                                   # a burst of zeros
                    a = int(rng.integers(0, len(f))); f[a:a + 8] = bytes(min(8, len(f) - a))
            frames.append(bytes(f)); sizes.append(size)
    n = len(frames)
    offs, off = [], 10
    for f in frames:
        offs.append(off); off += len(f)
    arc = zpk.assemble(frames, [("f%d" % i, offs[i], len(frames[i]), sizes[i], 0, method) for i in range(n)])
    desc = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
    desc["src_offset"] = offs; desc["comp_size"] = [len(f) for f in frames]; desc["uncomp_size"] = sizes
    desc["dst_capacity"] = sizes; desc["method"] = method; desc["flags"] = zpack_amd.DF_SKIP_HASH
    desc["dst_offset"] = np.concatenate([[0], np.cumsum((np.array(sizes, dtype=np.uint64) + 255) & ~np.uint64(255))])[:-1]
    total = int(desc["dst_offset"][-1]) + sizes[-1] + 256
    src = torch.from_numpy(np.frombuffer(arc, dtype=np.uint8).copy()).to(dev)
    dst = torch.zeros(total, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    t0 = time.time()
    codec.decode_batch_device(src, ddesc, n, dst, dres)
    torch.cuda.synchronize()
    t_gpu = time.time() - t0
    st = codec.decode_stats()
    res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
    out = dst.cpu().numpy()
    bad, decoded = 0, 0
    for i in range(n):
        rc, want, got, h = o.entry_decode(arc, offs[i], len(frames[i]), sizes[i], 0, method, sizes[i])
        a = int(desc["dst_offset"][i])
        if rc in (0, 15):
            decoded += 1
            if int(res[i]["status"]) != 0 or out[a:a + sizes[i]].tobytes() != want[:sizes[i]]:
                bad += 1; print("  MISMATCH (decodable) entry", i, res[i], rc)
        elif int(res[i]["status"]) != rc:
            bad += 1; print("  MISMATCH (status) entry", i, res[i], rc)
    print("method %d level %2d: %5d frames, %4d still decodable, mismatches %d, gpu %.1f ms, %s" % (method, level, n, decoded, bad, t_gpu * 1e3, st), flush=True)
    assert bad == 0 and st["fse_watchdog"] == 0 and st["fse_budget"] == 0
print("fuzz ok")
