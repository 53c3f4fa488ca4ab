#!/usr/bin/env python3
"""Developer: rate and memory of zpack_read_file_stream on the reference-made big recipes (tests/golden/recipes_big.json): 128 KiB input window,
1 MiB output window (what the reference's stream sizes suggest), the caller's loop of tests/read_archive.c.  Reports GiB/s of output, the
input offset at the first output byte, the growth of the process's resident set and of the device's used memory while streaming.
usage: stream_rate.py [label,label,...]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import ZPackAPI
from tests.test_gpu_zpack_api import _stream_entry
labels = sys.argv[1].split(",") if len(sys.argv) > 1 else ["lz4_0_64m_text", "lz4_0_512m_text", "zstd_3_64m_text"]
Z = ZPackAPI(zpack_amd.ZPACK_SO)
for label in labels:
    x = [r for r in json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "recipes_big.json"))) if r["label"] == label][0]
    plain = dg.fill(x["cls"], x["seed"], x["index"], x["size"])
    frame = np.frombuffer(dg.compress(x["method"], x["level"], plain), dtype=np.uint8)
    arc = zpk.assemble([frame.tobytes()], [("big", 10, len(frame), x["size"], x["hash"], x["method"])])
    sink = np.full(x["size"], 0xEE, dtype=np.uint8)
    best = 1e9
    for rep in range(2):
        rc, r, keep = Z.open_memory(arc)
        dev = {}
        t = time.perf_counter()
        rc, first, rss, got = _stream_entry(Z, r, 0, 131072, 1 << 20, sink, dev)
        best = min(best, time.perf_counter() - t)
        Z.lib.zpack_close_reader(C.byref(r))
        assert rc == 0 and got == x["size"] and np.array_equal(sink, plain), (label, rc, got)
    print("%-20s %4d MiB  %.3f s = %.2f GiB/s of output   first output after %d input bytes   host growth %.1f MiB   device growth %.1f MiB" % (
        label, x["size"] >> 20, best, x["size"] / best / (1 << 30), first, rss / (1 << 20), dev.get("peak", 0) / (1 << 20)), flush=True)
