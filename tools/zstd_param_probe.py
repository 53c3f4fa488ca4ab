#!/usr/bin/env python3
"""Developer (CPU): where does the ratio gap to libzstd come from?  The image's libzstd with its level-1 / level-3 parameters, and with the
window and hash table cut down to what the device encoder has (64 KiB window: 16-bit positions; 4096-entry table: 8 KiB of LDS), on the bench
corpus classes.  tools/zstd_param_probe.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from benchdata import datagen as dg
Z = C.CDLL("libzstd.so.1")
Z.ZSTD_createCCtx.restype = C.c_void_p
Z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
Z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
Z.ZSTD_compress2.restype = C.c_size_t
Z.ZSTD_CCtx_reset.argtypes = [C.c_void_p, C.c_int]
P = dict(level=100, windowLog=101, hashLog=102, chainLog=103, searchLog=104, minMatch=105, targetLength=106, strategy=107)
def comp(data, **kw):
    cctx = Z.ZSTD_createCCtx()
    for k, v in kw.items(): 
        r = Z.ZSTD_CCtx_setParameter(cctx, P[k], v)
    out = C.create_string_buffer(len(data) + 1024)
    n = Z.ZSTD_compress2(cctx, out, len(out), data, len(data))
    Z.ZSTD_freeCCtx.argtypes=[C.c_void_p]; Z.ZSTD_freeCCtx(cctx)
    return n
size = 1 << 20
for cls, name in ((0, "text"), (1, "records")):
    datas = [dg.fill(cls, 4, i, size).tobytes() for i in range(6)]
    def ratio(**kw): return sum(comp(d, **kw) for d in datas) / (len(datas) * size)
    print(name, "level1 default        %.4f" % ratio(level=1))
    print(name, "level1 windowLog 16   %.4f" % ratio(level=1, windowLog=16))
    print(name, "level1 w16 hashLog 12 %.4f" % ratio(level=1, windowLog=16, hashLog=12))
    print(name, "level1 w16 h12 minMatch4 %.4f" % ratio(level=1, windowLog=16, hashLog=12, minMatch=4))
    print(name, "level1 hashLog 12     %.4f" % ratio(level=1, hashLog=12))
    print(name, "level1 w16 hashLog 14 %.4f" % ratio(level=1, windowLog=16, hashLog=14))
    print(name, "level3 default        %.4f" % ratio(level=3))
    print(name, "level3 windowLog 16   %.4f" % ratio(level=3, windowLog=16))
