#!/usr/bin/env python3
"""Developer: TWO large single-frame entries decoded at the same time by two codecs on one device (two host threads) against one after the
other on one codec: does the second entry's work hide the first one's serial XXH3 chain?  usage: big_frame_two.py [MiB=256] [lz4|zstd]"""
import ctypes as C, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpack_amd
from benchdata import datagen as dg
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
METHOD = {"lz4": zpack_amd.METHOD_LZ4, "zstd": zpack_amd.METHOD_ZSTD}[sys.argv[2] if len(sys.argv) > 2 else "lz4"]
LEVEL = 0 if METHOD == zpack_amd.METHOD_LZ4 else 3
size = mib << 20
tile = np.concatenate([dg.fill(k % 2, 5, k, 1 << 20) for k in range(8)])
src = np.ascontiguousarray(np.resize(tile, size))
frame = np.frombuffer(dg.compress(METHOD, LEVEL, src), dtype=np.uint8)
arc = np.concatenate([frame, np.zeros(64, np.uint8)])
h = dg.xxh3(src)
codecs = [zpack_amd.Codec(0), zpack_amd.Codec(0)]
times = []
def job(codec, reps, out):
    d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = 0; d["comp_size"] = len(frame); d["uncomp_size"] = size; d["expect_hash"] = h; d["dst_capacity"] = size; d["method"] = METHOD
    back = np.empty(size, dtype=np.uint8); back[::4096] = 0
    r = np.zeros(1, dtype=zpack_amd.DECODE_RESULT)
    bp = (C.c_void_p * 1)(back.ctypes.data)
    for _ in range(reps):
        t0 = time.perf_counter()
        rc = codec.L.zpk_codec_decode_batch_host(codec.h, arc.ctypes.data, len(frame) + 1, d.ctypes.data, 1, bp, r.ctypes.data)
        times.append(round((time.perf_counter() - t0) * 1e3, 1))
        assert rc == 0 and r["status"][0] == 0
    out.append(np.array_equal(back, src))
for c in codecs: job(c, 1, [])          # warm (allocations)
times.clear(); t = time.perf_counter(); o = []; job(codecs[0], 4, o); t1 = time.perf_counter() - t; print("one codec, per call ms:", times); times.clear()
t = time.perf_counter(); o = []
th = [threading.Thread(target=job, args=(c, 4, o)) for c in codecs]
for x in th: x.start()
for x in th: x.join()
t2 = time.perf_counter() - t
assert all(o); print("two codecs, per call ms:", times)
print("%s %d MiB frames: one codec, 4 entries one after the other: %.1f ms each = %.2f GiB/s | two codecs on one device, 2 x 4 entries side by side: %.1f ms per entry = %.2f GiB/s" % (
    "LZ4" if METHOD == zpack_amd.METHOD_LZ4 else "Zstandard-3", mib, t1 / 4 * 1e3, 4 * mib / 1024 / t1, t2 / 8 * 1e3, 8 * mib / 1024 / t2))
