"""Developer aid: device compress throughput / ratio per class (source bytes resident in HBM).
usage: python tools/enc_bench.py [n_entries] [entry_bytes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
codec = zpack_amd.Codec(0)
dev = torch.device("cuda:0")
for cls, cname in enumerate(["text", "records", "random", "runs"]):
    plain = np.concatenate([dg.fill(cls, 4, i, size) for i in range(64)])
    reps = (n + 63) // 64
    src = torch.from_numpy(np.tile(plain, reps)[: n * size]).to(dev)
    for method, mname in ((2, "lz4"), (1, "zstd")):
        bound = codec.compress_bound(method, size) if hasattr(codec, "compress_bound") else size + size // 128 + 1024
        desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
        desc["src_offset"] = np.arange(n, dtype=np.uint64) * size
        desc["size"] = size
        desc["dst_offset"] = np.arange(n, dtype=np.uint64) * ((bound + 255) & ~255)
        desc["dst_capacity"] = bound
        desc["method"] = method
        desc["level"] = 1
        ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        dst = torch.empty(int(n * ((bound + 255) & ~255)), dtype=torch.uint8, device=dev)
        dres = torch.zeros(n * zpack_amd.ENCODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            codec.encode_batch_device(src, ddesc, n, dst, dres)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
        # K7: scan of the compressed sizes + compaction of the payloads into one packed stream
        offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        packed = torch.empty(int(n * bound), dtype=torch.uint8, device=dev)
        codec.set_profiling(True)
        for it in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            codec.pack_batch_device(dst, ddesc, dres, n, packed, offs, int(bound))
            torch.cuda.synchronize(); dtp = time.perf_counter() - t
        gather_ms = codec.kernel_ms(zpack_amd.K_PACK)
        codec.set_profiling(False)
        total_c = int(offs[-1].item())
        res = dres.cpu().numpy().view(zpack_amd.ENCODE_RESULT)
        assert total_c == int(res["comp_size"][res["status"] == 0].sum())
        print("         pack: %.3f ms wall, k_pack_gather %.3f ms = %.0f GB/s (read + write of %.2f GB packed)" %
              (dtp * 1e3, gather_ms, 2 * total_c / (gather_ms * 1e-3) / 1e9, total_c / 1e9), flush=True)
        ok = int((res["status"] == 0).sum())
        print("%-8s %-5s ok %d/%d ratio %.3f  %.1f GiB/s source" % (cname, mname, ok, n, res["comp_size"].sum() / (n * size), n * size / dt / 2**30), flush=True)
