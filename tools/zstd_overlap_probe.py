#!/usr/bin/env python3
"""Developer probe: what would overlapping the two Zstandard stages buy?  Two codec contexts decode the two halves of a C3-like batch
(n x 256 KiB, zstd-3) on two streams; the second stream is held back by a timed spin so that its pre-decode stage (k_zstd_fse) runs
beside the first half's execute stage (k_zstd_exec).  Prints ms for: whole batch on one stream / halves back to back / halves offset.
  tools/zstd_overlap_probe.py [entries] [size] [level]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    level = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    b = dg.Batch(n, size, size, method=dg.ZSTD, level=level, seed=3, mix=-1, threads=16)
    desc, dst_bytes = zpack_amd.decode_descs_from_batch(b)
    src = torch.from_numpy(b.archive).to(dev)
    dst = torch.empty(dst_bytes, dtype=torch.uint8, device=dev)
    h = n // 2
    parts = [(0, n), (0, h), (h, n)]
    dd = [torch.from_numpy(desc[lo:hi].copy().view(np.uint8)).to(dev) for lo, hi in parts]
    rr = [torch.zeros((hi - lo) * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev) for lo, hi in parts]
    cA, cB = zpack_amd.Codec(0), zpack_amd.Codec(0)
    # streams share a few hardware queues (round robin at creation): take a pair that demonstrably runs side by side
    def T(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
    for _ in range(3):
        cyc_per_ms = 100_000_000 / T(lambda: torch.cuda._sleep(100_000_000))
    pool = [torch.cuda.Stream() for _ in range(8)]
    def two(a, b):
        with torch.cuda.stream(a): torch.cuda._sleep(int(20 * cyc_per_ms))
        with torch.cuda.stream(b): torch.cuda._sleep(int(20 * cyc_per_ms))
    s1 = pool[0]; s2 = None
    for cand in pool[1:]:
        two(s1, cand); t = min(T(lambda: two(s1, cand)) for _ in range(2))
        print("two 20 ms spins on stream 0 and a candidate: %.1f ms" % t, flush=True)
        if t < 23.0: s2 = cand; break
    if s2 is None: print("no concurrent pair of streams found"); return
    out_bytes = float(b.uncomp_sizes.sum())

    def run(fn, reps=4):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best

    def whole():
        cA.decode_batch_device(src, dd[0], n, dst, rr[0], s1.cuda_stream)

    def halves_serial():
        cA.decode_batch_device(src, dd[1], h, dst, rr[1], s1.cuda_stream)
        cA.decode_batch_device(src, dd[2], n - h, dst, rr[2], s1.cuda_stream)

    def halves_parallel():
        cA.decode_batch_device(src, dd[1], h, dst, rr[1], s1.cuda_stream)
        cB.decode_batch_device(src, dd[2], n - h, dst, rr[2], s2.cuda_stream)

    whole(); halves_parallel(); torch.cuda.synchronize()
    t_whole = run(whole)
    cA.set_profiling(True); whole(); torch.cuda.synchronize()
    f_ms, e_ms = cA.kernel_ms(zpack_amd.K_ZSTD_FSE), cA.kernel_ms(zpack_amd.K_ZSTD)
    cA.set_profiling(False)
    print("whole batch, one stream: %.1f ms (%.1f GiB/s); k_zstd_fse %.1f + k_zstd_exec.. %.1f" % (t_whole, out_bytes / t_whole / 1e-3 / 2**30, f_ms, e_ms), flush=True)
    print("halves back to back, one stream: %.1f ms" % run(halves_serial), flush=True)
    print("halves on two streams, no offset: %.1f ms" % run(halves_parallel), flush=True)
    for frac in (0.5, 0.75, 1.0, 1.25):
        delay = f_ms / 2 * frac

        def offset():
            with torch.cuda.stream(s2):
                torch.cuda._sleep(int(delay * cyc_per_ms))
            cA.decode_batch_device(src, dd[1], h, dst, rr[1], s1.cuda_stream)
            cB.decode_batch_device(src, dd[2], n - h, dst, rr[2], s2.cuda_stream)
        t = run(offset)
        print("halves on two streams, second held back %.1f ms (%.2f of a half pre-decode): %.1f ms (%.1f GiB/s)" % (delay, frac, t, out_bytes / t / 1e-3 / 2**30), flush=True)
    r = rr[1].cpu().numpy().view(zpack_amd.DECODE_RESULT)
    r2 = rr[2].cpu().numpy().view(zpack_amd.DECODE_RESULT)
    print("statuses ok:", bool((r["status"] == 0).all() and (r2["status"] == 0).all()),
          "hashes ok:", bool(np.array_equal(np.concatenate([r["hash"], r2["hash"]]), b.hashes)))


if __name__ == "__main__":
    main()
