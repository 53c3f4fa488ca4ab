#!/usr/bin/env python3
"""Developer: the PCIe-inclusive rate of the host-pointer path — zpack_read_files_packed of libzpack_amd.so on a memory-backed reader
(archive in pageable host memory in, decoded bytes in pageable host memory out).  tools/host_rate.py [entries] [method: 2 lz4 | 1 zstd]"""
import os, sys, time
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zpack_amd
from benchdata import datagen as dg
from tests._libs import ZPackAPI, FileEntry, Reader
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
method = int(sys.argv[2]) if len(sys.argv) > 2 else dg.LZ4
size = 65536 if method == dg.LZ4 else 262144
b = dg.Batch(n, size, size, method=method, level=0 if method == dg.LZ4 else 3, seed=1)
Z = ZPackAPI(zpack_amd.ZPACK_SO)
arc = b.archive.tobytes()
rc, r, keep = Z.open_memory(arc)
assert rc == 0 and r.file_count == n
u8p = C.POINTER(C.c_uint8)
ptrs = (C.POINTER(FileEntry) * n)(*[C.pointer(r.file_entries[i]) for i in range(n)])
total = n * size
big = (C.c_uint8 * total)()
offs = (C.c_uint64 * n)()
results = (C.c_int * n)()
Z.lib.zpack_read_files_packed.argtypes = [C.POINTER(Reader), C.POINTER(C.POINTER(FileEntry)), C.c_uint64, u8p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.c_void_p]
for it in range(3):
    t0 = time.time()
    rc = Z.lib.zpack_read_files_packed(C.byref(r), ptrs, n, C.cast(big, u8p), total, offs, results, None)
    dt = time.time() - t0
    assert rc == 0 and all(x == 0 for x in results)
    print("pass %d: %d entries, %.2f GB compressed in, %.2f GB out, %.1f ms -> %.1f GiB/s decompressed (host pointers, PCIe inclusive)" % (it, n, len(arc) / 1e9, total / 1e9, dt * 1e3, total / dt / 2**30), flush=True)
Z.close_reader(r)

