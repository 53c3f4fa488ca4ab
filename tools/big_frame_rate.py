#!/usr/bin/env python3
"""Developer: ONE large entry as the REFERENCE writes it — one LZ4 frame of linked 64 KiB blocks made by liblz4 (lib/zpack_write.c:204-210),
or one Zstandard frame made by libzstd (:179) — through zpk_codec_decode_batch_host (what zpack_read_file calls with a batch of one):
block-parallel (lz4_pj.h / zstd_pj.h: ZPK_OPT_DEC_SPLIT_MIN, default 2 MiB) against the one-wave decoder.
usage: big_frame_rate.py [MiB=256] [MiB_one_wave=16] [lz4|zstd] [level]"""
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
METHOD = {"lz4": zpack_amd.METHOD_LZ4, "zstd": zpack_amd.METHOD_ZSTD}[sys.argv[3] if len(sys.argv) > 3 else "lz4"]
LEVEL = int(sys.argv[4]) if len(sys.argv) > 4 else (0 if METHOD == zpack_amd.METHOD_LZ4 else 3)
codec = zpack_amd.Codec(0)
L = codec.L


def one(size, cls_mix, parallel):
    codec.set_option(zpack_amd.OPT_DEC_SPLIT_MIN, (2 << 20) if parallel else 0)
    tile = np.concatenate([dg.fill(cls_mix[k % len(cls_mix)], 5, k, 1 << 20) for k in range(8)])
    src = np.ascontiguousarray(np.resize(tile, size))
    frame = np.frombuffer(dg.compress(METHOD, LEVEL, src), dtype=np.uint8)
    arc = np.concatenate([frame, np.zeros(64, np.uint8)])
    d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = 0; d["comp_size"] = len(frame); d["uncomp_size"] = size; d["expect_hash"] = dg.xxh3(src); d["dst_capacity"] = size; d["method"] = METHOD
    back = np.empty(size, dtype=np.uint8)
    r = np.zeros(1, dtype=zpack_amd.DECODE_RESULT)
    bp = (C.c_void_p * 1)(back.ctypes.data)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        rc = L.zpk_codec_decode_batch_host(codec.h, arc.ctypes.data, len(frame) + 1, d.ctypes.data, 1, bp, r.ctypes.data)
        best = min(best, time.perf_counter() - t)
        assert rc == 0 and r["status"][0] == 0, (rc, r)
    assert np.array_equal(back, src)
    st = codec.decode_stats()
    if parallel and not st["frame_parallel_entries"]: print("   (not block-parallel: flags 0x%x)" % st["zstd_blocks_flags"])
    return best, len(frame) / size, st["frame_parallel_entries"], st["frame_parallel_frames"]


def one_device(size, cls_mix):
    """the same entry with its bytes and its output in DEVICE memory (zpk_codec_decode_big_device)"""
    import torch
    codec.set_option(zpack_amd.OPT_DEC_SPLIT_MIN, 2 << 20)
    tile = np.concatenate([dg.fill(cls_mix[k % len(cls_mix)], 5, k, 1 << 20) for k in range(8)])
    src = np.ascontiguousarray(np.resize(tile, size))
    frame = np.frombuffer(dg.compress(METHOD, LEVEL, src), dtype=np.uint8)
    dsrc = torch.zeros(len(frame) + 64, dtype=torch.uint8, device="cuda:0")
    dsrc[:len(frame)] = torch.from_numpy(frame.copy()).to("cuda:0")
    ddst = torch.zeros(size + 64, dtype=torch.uint8, device="cuda:0")
    d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = 0; d["comp_size"] = len(frame); d["uncomp_size"] = size; d["expect_hash"] = dg.xxh3(src); d["dst_capacity"] = size; d["method"] = METHOD
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = codec.decode_big_device(dsrc, d, ddst)
        best = min(best, time.perf_counter() - t)
        assert int(r["status"]) == 0
    assert np.array_equal(ddst[:size].cpu().numpy(), src)
    return best, codec.decode_stats()["frame_parallel_entries"]


for name, mix in (("text+records", (0, 1)), ("text", (0,)), ("byte runs", (3,)), ("random", (2,))):
    t, ratio, par, nb = one(mib << 20, mix, True)
    t1, ratio1, par1, _ = one(mib1 << 20, mix, False)
    print("%-13s one %d MiB %s frame of the reference writer, %d blocks side by side (parallel entries: %d): read %.1f ms = %.2f GiB/s (host pointers in and out), ratio %.3f | "
          "one %d MiB frame, one wave: read %.1f ms = %.3f GiB/s" % (name, mib, "LZ4" if METHOD == zpack_amd.METHOD_LZ4 else "Zstandard-%d" % LEVEL, nb, par, t * 1e3, mib / 1024 / t, ratio, mib1, t1 * 1e3, mib1 / 1024 / t1), flush=True)
    td, pard = one_device(mib << 20, mix)
    print("%-13s   the same with the entry and its output in device memory (zpk_codec_decode_big_device, parallel entries: %d): %.1f ms = %.2f GiB/s" % ("", pard, td * 1e3, mib / 1024 / td), flush=True)
