#!/usr/bin/env python3
"""Developer check: do two torch streams run concurrently on this box, and does a decode on one stream run beside a spin on another?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = [torch.cuda.Stream() for _ in range(4)]
sp = torch.cuda.Stream(priority=-1)
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for _ in range(3):
    cyc = 100_000_000 / T(lambda: torch.cuda._sleep(100_000_000))
print("spin: %.0f cycles per ms; a 30 ms spin alone: %.1f ms" % (cyc, T(lambda: torch.cuda._sleep(int(30 * cyc)))), flush=True)
def two_sleeps(a, b):
    with torch.cuda.stream(a): torch.cuda._sleep(int(30 * cyc))
    with torch.cuda.stream(b): torch.cuda._sleep(int(30 * cyc))
for i in range(1, 4):
    print("30 ms spin on stream 0 and on stream %d: %.1f ms" % (i, T(lambda: two_sleeps(s[0], s[i]))), flush=True)
print("30 ms spin on stream 0 and the high-priority stream: %.1f ms" % T(lambda: two_sleeps(s[0], sp)), flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
b = dg.Batch(n, 262144, 262144, method=dg.ZSTD, level=3, seed=3, mix=-1, threads=16)
desc, dst_bytes = zpack_amd.decode_descs_from_batch(b)
src = torch.from_numpy(b.archive).to(dev)
dst = torch.empty(dst_bytes, dtype=torch.uint8, device=dev)
dd = torch.from_numpy(desc.view(np.uint8)).to(dev)
rr = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
c = zpack_amd.Codec(0)
def dec(st): c.decode_batch_device(src, dd, n, dst, rr, st.cuda_stream)
dec(s[0]); torch.cuda.synchronize()
print("decode alone: %.1f ms" % T(lambda: dec(s[0])), flush=True)
for i in range(1, 4):
    def both():
        with torch.cuda.stream(s[i]): torch.cuda._sleep(int(30 * cyc))
        dec(s[0])
    print("30 ms spin on stream %d, then decode on stream 0: %.1f ms" % (i, T(both)), flush=True)
def both2():
    dec(s[0])
    with torch.cuda.stream(s[1]): torch.cuda._sleep(int(30 * cyc))
print("decode on stream 0, then 30 ms spin on stream 1: %.1f ms" % T(both2), flush=True)
