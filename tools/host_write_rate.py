#!/usr/bin/env python3
"""Developer: the PCIe-inclusive rate of the host-pointer WRITE path — zpk_codec_encode_batch_host (what zpack_write_files calls once
per batch): sources in pageable host memory in, compressed payloads in pageable host memory out.
  tools/host_write_rate.py [entries] [entry_bytes] [method: 1 zstd | 2 lz4] [level]"""
import os, sys, time
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zpack_amd
from benchdata import datagen as dg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
method = int(sys.argv[3]) if len(sys.argv) > 3 else 1
level = int(sys.argv[4]) if len(sys.argv) > 4 else 1
codec = zpack_amd.Codec(0)
rng = np.random.default_rng(5)
classes = rng.choice(4, size=64, p=[0.70, 0.20, 0.05, 0.05])
pool = [np.ascontiguousarray(dg.fill(int(classes[i]), 5, i, size)) for i in range(64)]
srcs = [pool[i % 64].copy() for i in range(n)]                      # n separate pageable buffers, as a caller's files would be
bound = codec.compress_bound(method, size)
outs = [np.empty(bound, dtype=np.uint8) for _ in range(n)]
desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
desc["size"] = size; desc["dst_capacity"] = bound; desc["method"] = method; desc["level"] = level
res = np.zeros(n, dtype=zpack_amd.ENCODE_RESULT)
sp = (C.c_void_p * n)(*[a.ctypes.data for a in srcs])
dp = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
L = codec.L
L.zpk_codec_encode_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
for it in range(3):
    t0 = time.time()
    rc = L.zpk_codec_encode_batch_host(codec.h, sp, desc.ctypes.data, n, dp, res.ctypes.data)
    dt = time.time() - t0
    assert rc == 0 and (res["status"] == 0).all(), (rc, codec.L.zpk_codec_last_error(codec.h))
    print("pass %d: %d x %d bytes, method %d level %d: %.2f GB of source in, %.2f GB compressed out, %.1f ms -> %.1f GiB/s of source (host pointers, PCIe inclusive)"
          % (it, n, size, method, level, n * size / 1e9, res["comp_size"].sum() / 1e9, dt * 1e3, n * size / dt / 2**30), flush=True)
# spot check: the first payloads decode back (GPU decoder) to their sources
import zlib
d = np.zeros(8, dtype=zpack_amd.DECODE_DESC)
arc = bytearray(); off = 0
for i in range(8):
    cs = int(res["comp_size"][i]); d[i]["src_offset"] = off; d[i]["comp_size"] = cs; d[i]["uncomp_size"] = size; d[i]["expect_hash"] = res["hash"][i]
    d[i]["dst_capacity"] = size; d[i]["method"] = method; arc += outs[i][:cs].tobytes(); off += cs
arc += b"\0" * 64
r2, back = codec.decode_batch_host(bytes(arc), d)
print("round trip of the first 8 payloads:", bool((r2["status"] == 0).all() and all(np.array_equal(back[i], srcs[i]) for i in range(8))))
