#!/usr/bin/env python3
"""Developer: randomized inputs through the device ENCODERS (zpack_write_files on libzpack_amd.so), every archive decoded by the oracle and by
the compiled reference (stock libzstd / liblz4 when oracle/_ref is there).  tools/enc_fuzz.py [rounds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from benchdata import datagen as dg
import zpack_amd
from tests._libs import ZPackAPI
from tests.test_gpu_zpack_api import _decode_all_with_checkers, METHOD_ZSTD, METHOD_LZ4

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Z = ZPackAPI(zpack_amd.ZPACK_SO)
rng = np.random.default_rng(seed)


def make(i):
    kind = int(rng.integers(0, 8))
    n = int(rng.choice([int(rng.integers(1, 5000)), int(rng.integers(60000, 70000)), int(rng.integers(100000, 400000)), int(rng.integers(1, 1 << 20))]))
    if kind == 0:
        return dg.fill(int(rng.integers(0, 4)), seed, i, n)
    if kind == 1:                                                    # small alphabet, skewed
        k = int(rng.integers(2, 257)); p = rng.random(k) ** int(rng.integers(1, 8)); p /= p.sum()
        return rng.choice(k, size=n, p=p).astype(np.uint8)
    if kind == 2:                                                    # records of random width with a few random bytes
        w = int(rng.integers(3, 200)); base = rng.integers(0, 256, w, dtype=np.uint8)
        a = np.tile(base, n // w + 1)[:n].copy(); idx = rng.integers(0, n, max(1, n // int(rng.integers(4, 64)))); a[idx] = rng.integers(0, 256, len(idx), dtype=np.uint8)
        return a
    if kind == 3:                                                    # text with long repeats at long distance
        t = dg.fill(dg.TEXT, seed, i, max(1, n // 3)); return np.concatenate([t, t[::-1].copy(), t])[:n]
    if kind == 4:                                                    # runs
        out = np.empty(n, dtype=np.uint8); p = 0
        while p < n:
            l = int(rng.integers(1, 3000)); out[p:p + l] = int(rng.integers(0, 256)); p += l
        return out
    if kind == 5:                                                    # exactly repeating period
        per = int(rng.integers(1, 70000)); return np.tile(rng.integers(0, 256, per, dtype=np.uint8), n // per + 1)[:n].copy()
    if kind == 6:                                                    # mostly random with embedded text
        a = rng.integers(0, 256, n, dtype=np.uint8); t = dg.fill(dg.TEXT, seed, i, max(1, n // 4)); a[n // 3:n // 3 + len(t)] = t[:max(0, n - n // 3)][:len(a[n // 3:n // 3 + len(t)])]
        return a
    return dg.fill(dg.TEXT, seed + 7, i, n)


t0 = time.time()
for r in range(rounds):
    want = [("f%04d" % i, make(r * 1000 + i).tobytes()) for i in range(48)]
    for method, levels in ((METHOD_ZSTD, (1, 3, 6)), (METHOD_LZ4, (0, 3, 9))):
        for level in levels:
            arc = Z.write_archive(want, method, level)
            _decode_all_with_checkers(arc, want)
            comp = len(arc)
            print("round %d method %d level %d: %d files, %.1f MB -> %.1f MB  ok  (%.0f s)" % (r, method, level, len(want), sum(len(d) for _, d in want) / 1e6, comp / 1e6, time.time() - t0), flush=True)
print("enc fuzz ok")
