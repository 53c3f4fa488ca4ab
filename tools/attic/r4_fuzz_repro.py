#!/usr/bin/env python3
"""round 4 debugging aid: re-create single entries of a tools/fuzz_gpu.py round and decode them alone through a chosen LZ4 path.
usage: r4_fuzz_repro.py <per> <seed> <level> <path: one|slot|window> <entry indices...>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import oracle
per, seed, level, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
want_idx = [int(x) for x in sys.argv[5:]]
method = dg.LZ4
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
            if k % 17 == 0 and method == dg.LZ4:
                f = f + bytearray(dg.compress(method, level, plain[:1000]))
            if k % 13 == 0:
                a = int(rng.integers(0, len(f))); f[a:a + 8] = bytes(min(8, len(f) - a))
        frames.append(bytes(f)); sizes.append(size)
codec = zpack_amd.Codec(0)
if path != "one":
    codec.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MIN, 0); codec.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MAX_COMP, 4 << 20)
    codec.set_option(zpack_amd.OPT_LZ4_EXEC_WINDOW, 1 if path == "window" else 0)
o = oracle()
def walk(block):
    p, C, out, seqs = 0, len(block), 0, []
    while p < C:
        tok = block[p]; tp = p; p += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = block[p]; p += 1; lit += b
                if b != 255: break
        p += lit
        if p >= C: seqs.append((tp, out, lit, 0, 0)); out += lit; break
        off = block[p] | (block[p + 1] << 8); p += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = block[p]; p += 1; ml += b
                if b != 255: break
        ml += 4
        seqs.append((tp, out, lit, ml, off)); out += lit + ml
    return seqs
for i in want_idx:
    f, size = frames[i], sizes[i]
    arc = zpk.assemble([f], [("f", 10, len(f), size, 0, 2)])
    d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = 10; d["comp_size"] = len(f); d["uncomp_size"] = size; d["dst_capacity"] = size; d["method"] = 2; d["flags"] = zpack_amd.DF_SKIP_HASH
    dev = torch.device("cuda:0")
    src = torch.from_numpy(np.frombuffer(arc, dtype=np.uint8).copy()).to(dev); dst = torch.zeros(size + 256, dtype=torch.uint8, device=dev)
    dres = torch.zeros(zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, torch.from_numpy(d.view(np.uint8)).to(dev), 1, dst, dres); torch.cuda.synchronize()
    r = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)[0]; out = dst.cpu().numpy()[:size]
    rc, want, got, h = o.entry_decode(arc, 10, len(f), size, 0, 2, size)
    w = np.frombuffer(want, dtype=np.uint8)[:size]
    bad = np.nonzero(out != w)[0]
    print("entry", i, "gpu", r, "oracle rc", rc, "produced", got, "stats", {k: v for k, v in codec.decode_stats().items() if k.startswith("lz4")}, "first bad", int(bad[0]) if bad.size else None, "nbad", bad.size)
    if bad.size:
        b0 = int(bad[0])
        # find the block + sequence that writes position b0
        pos, outbase = 7, 0
        while True:
            bh = int.from_bytes(f[pos:pos + 4], "little"); pos += 4
            if bh == 0: break
            bsz = bh & 0x7FFFFFFF
            if bh >> 31: outbase += bsz; pos += bsz; continue
            try: seqs = walk(f[pos:pos + bsz])
            except IndexError: print("  block walk failed"); break
            end = outbase + (seqs[-1][1] + seqs[-1][2] + seqs[-1][3])
            if b0 < end:
                for k, (tp, op_, lit, ml, off) in enumerate(seqs):
                    if outbase + op_ + lit + ml > b0:
                        print("  block at", pos, "out base", outbase, "seq", k, "of", len(seqs), "tok", tp, "out", outbase + op_, "lit", lit, "ml", ml, "off", off)
                        for kk in range(max(0, k - 3), min(len(seqs), k + 3)): print("     ", kk, seqs[kk])
                        break
                break
            outbase = end; pos += bsz
        print("  gpu ", out[max(0, b0 - 8):b0 + 24].tobytes()); print("  want", w[max(0, b0 - 8):b0 + 24].tobytes())
