#!/usr/bin/env python3
"""Developer: encode rate on data with LONG literal runs — random bytes with a 64-byte marker every `gap` bytes, so that a 64 KiB block holds a
few sequences whose literal runs are tens of KiB.  tools/enc_longlit.py [entries] [gap]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
gap = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
size = 1 << 20
rng = np.random.default_rng(3)
marker = rng.integers(0, 256, 64, dtype=np.uint8)
pool = []
for i in range(16):
    a = rng.integers(0, 256, size, dtype=np.uint8)
    for p in range(gap, size - 64, gap): a[p:p + 64] = marker
    pool.append(a)
plain = np.concatenate([pool[i % 16] for i in range(n)])
codec = zpack_amd.Codec(0); dev = torch.device("cuda:0")
src = torch.from_numpy(plain).to(dev)
for method, mname, level in ((2, "lz4", 0), (1, "zstd", 1)):
    bound = codec.compress_bound(method, size); slot = (bound + 255) & ~255
    desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
    desc["src_offset"] = np.arange(n, dtype=np.uint64) * size; desc["size"] = size
    desc["dst_offset"] = np.arange(n, dtype=np.uint64) * slot; desc["dst_capacity"] = bound; desc["method"] = method; desc["level"] = level
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dst = torch.empty(n * slot, dtype=torch.uint8, device=dev)
    dres = torch.zeros(n * zpack_amd.ENCODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        codec.encode_batch_device(src, ddesc, n, dst, dres)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    res = dres.cpu().numpy().view(zpack_amd.ENCODE_RESULT)
    # round trip through the GPU decoder
    dd = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
    dd["src_offset"] = desc["dst_offset"]; dd["comp_size"] = res["comp_size"]; dd["uncomp_size"] = size; dd["expect_hash"] = res["hash"]
    dd["dst_offset"] = np.arange(n, dtype=np.uint64) * size; dd["dst_capacity"] = size; dd["method"] = method
    back = torch.empty(n * size, dtype=torch.uint8, device=dev); r2 = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(dst, torch.from_numpy(dd.view(np.uint8)).to(dev), n, back, r2); torch.cuda.synchronize()
    ok = bool((res["status"] == 0).all() and (r2.cpu().numpy().view(zpack_amd.DECODE_RESULT)["status"] == 0).all() and torch.equal(back, src))
    print("long literals (%d-byte gaps) %-4s ratio %.4f  %.1f GiB/s of source, round trip %s" % (gap, mname, res["comp_size"].sum() / (n * size), n * size / dt / 2**30, "ok" if ok else "FAILED"), flush=True)
