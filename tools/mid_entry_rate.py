#!/usr/bin/env python3
"""Developer: ONE mid-size entry (256 KiB ... 8 MiB, one frame of the reference writer) through zpk_codec_decode_batch_host with a batch of one —
what zpack_read_file does for a single file: block-parallel (ZPK_OPT_DEC_SPLIT_MIN lowered to 64 KiB) against one wave."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpack_amd
from benchdata import datagen as dg
codec = zpack_amd.Codec(0)
L = codec.L
for method, level, mname in ((zpack_amd.METHOD_LZ4, 0, "LZ4"), (zpack_amd.METHOD_ZSTD, 3, "Zstandard-3")):
    for kib in (256, 512, 1024, 2048, 4096, 8192):
        size = kib << 10
        src = dg.fill(dg.TEXT, 5, kib, size)
        frame = np.frombuffer(dg.compress(method, level, src), dtype=np.uint8)
        arc = np.concatenate([frame, np.zeros(64, np.uint8)])
        d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
        d["src_offset"] = 0; d["comp_size"] = len(frame); d["uncomp_size"] = size; d["expect_hash"] = dg.xxh3(src); d["dst_capacity"] = size; d["method"] = method
        back = np.zeros(size, dtype=np.uint8)
        r = np.zeros(1, dtype=zpack_amd.DECODE_RESULT)
        bp = (C.c_void_p * 1)(back.ctypes.data)
        res = []
        for split in (64 << 10, 0):
            codec.set_option(zpack_amd.OPT_DEC_SPLIT_MIN, split)
            best = 1e9
            for _ in range(5):
                t = time.perf_counter()
                rc = L.zpk_codec_decode_batch_host(codec.h, arc.ctypes.data, len(frame) + 1, d.ctypes.data, 1, bp, r.ctypes.data)
                best = min(best, time.perf_counter() - t)
                assert rc == 0 and r["status"][0] == 0 and np.array_equal(back, src)
            res.append((best, codec.decode_stats()["frame_parallel_entries"]))
        print("%-12s %5d KiB: block-parallel (taken: %d) %.2f ms = %.2f GiB/s | one wave %.2f ms = %.3f GiB/s" % (
            mname, kib, res[0][1], res[0][0] * 1e3, size / res[0][0] / (1 << 30), res[1][0] * 1e3, size / res[1][0] / (1 << 30)), flush=True)
