#!/usr/bin/env python3
"""Developer: ONE large entry through the host paths (what zpack_write_file / zpack_read_file call with a batch of one) — written as
a sequence of 512 KiB pieces side by side (one frame) and read back one wave per frame (ZPK_OPT_ENC_SPLIT_MIN / ZPK_OPT_DEC_SPLIT_MIN, default
2 MiB) against one wave on one frame.
usage: big_entry_rate.py [MiB=256] [MiB_one_wave=16]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zpack_amd
from benchdata import datagen as dg

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mib1 = int(sys.argv[2]) if len(sys.argv) > 2 else 16
codec = zpack_amd.Codec(0)
L = codec.L
L.zpk_codec_encode_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]


def one(size, method, level, split):
    codec.set_option(zpack_amd.OPT_ENC_SPLIT_MIN, (2 << 20) if split else 0)
    codec.set_option(zpack_amd.OPT_DEC_SPLIT_MIN, (2 << 20) if split else 0)
    tile = np.concatenate([dg.fill(k % 2, 5, k, 1 << 20) for k in range(8)])              # text / records, 8 MiB of it repeated
    src = np.ascontiguousarray(np.resize(tile, size))
    bound = codec.compress_bound(method, size)
    out = np.empty(bound + 64, dtype=np.uint8)
    desc = np.zeros(1, dtype=zpack_amd.ENCODE_DESC)
    desc["size"] = size; desc["dst_capacity"] = bound; desc["method"] = method; desc["level"] = level
    res = np.zeros(1, dtype=zpack_amd.ENCODE_RESULT)
    sp = (C.c_void_p * 1)(src.ctypes.data); dp = (C.c_void_p * 1)(out.ctypes.data)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        rc = L.zpk_codec_encode_batch_host(codec.h, sp, desc.ctypes.data, 1, dp, res.ctypes.data)
        best = min(best, time.perf_counter() - t)
        assert rc == 0 and res["status"][0] == 0, (rc, res)
    assert int(res["hash"][0]) == dg.xxh3(src)
    # read it back
    cs = int(res["comp_size"][0])
    d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = 0; d["comp_size"] = cs; d["uncomp_size"] = size; d["expect_hash"] = res["hash"]; d["dst_capacity"] = size; d["method"] = method
    back = np.empty(size, dtype=np.uint8)
    r = np.zeros(1, dtype=zpack_amd.DECODE_RESULT)
    bp = (C.c_void_p * 1)(back.ctypes.data)
    rbest = 1e9
    for _ in range(3):
        t = time.perf_counter()
        rc = L.zpk_codec_decode_batch_host(codec.h, out.ctypes.data, cs + 1, d.ctypes.data, 1, bp, r.ctypes.data)
        rbest = min(rbest, time.perf_counter() - t)
        assert rc == 0 and r["status"][0] == 0, (rc, r)
    assert np.array_equal(back, src)
    return best, rbest, cs / size


for method, level, name in [(zpack_amd.METHOD_LZ4, 0, "lz4"), (zpack_amd.METHOD_ZSTD, 1, "zstd-1"), (zpack_amd.METHOD_ZSTD, 3, "zstd-3"), (zpack_amd.METHOD_NONE, 0, "stored")]:
    t, tr, r = one(mib << 20, method, level, True)
    t1, tr1, r1 = one(mib1 << 20, method, level, False)
    print("%-7s one %d MiB entry, %d pieces side by side (one frame): write %.1f ms = %.2f GiB/s, read %.1f ms = %.2f GiB/s (host pointers in and out), ratio %.4f | one %d MiB entry, one frame, one wave: write %.1f ms = %.3f GiB/s, read %.1f ms = %.3f GiB/s, ratio %.4f"
          % (name, mib, (mib << 20) // (512 << 10), t * 1e3, mib / 1024 / t, tr * 1e3, mib / 1024 / tr, r, mib1, t1 * 1e3, mib1 / 1024 / t1, tr1 * 1e3, mib1 / 1024 / tr1, r1), flush=True)
