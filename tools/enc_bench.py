"""Developer aid: device compress throughput / ratio per class (source bytes resident in HBM).
usage: python tools/enc_bench.py [n_entries] [entry_bytes] [level]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
level = int(sys.argv[3]) if len(sys.argv) > 3 else 1
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
        desc["level"] = level
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
        # round trip at full size: the packed stream decodes (GPU) to entries whose bytes and XXH3 equal the sources / the
        # hashes the encoder took of them, every status 0 — encoder + compaction + decoder end to end
        ddesc2 = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
        ho = offs.cpu().numpy().view(np.uint64)
        ddesc2["src_offset"] = ho[:-1]; ddesc2["comp_size"] = res["comp_size"]; ddesc2["uncomp_size"] = size
        ddesc2["expect_hash"] = res["hash"]; ddesc2["dst_offset"] = np.arange(n, dtype=np.uint64) * size
        ddesc2["dst_capacity"] = size; ddesc2["method"] = method
        back = torch.empty(n * size, dtype=torch.uint8, device=dev)
        dres2 = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        pk = torch.cat([packed[:total_c], torch.zeros(64, dtype=torch.uint8, device=dev)])      # (offset + size < archive size guard)
        codec.decode_batch_device(pk, torch.from_numpy(ddesc2.view(np.uint8)).to(dev), n, back, dres2)
        torch.cuda.synchronize()
        r2 = dres2.cpu().numpy().view(zpack_amd.DECODE_RESULT)
        rt_ok = bool((r2["status"] == 0).all() and np.array_equal(r2["hash"], res["hash"]) and torch.equal(back, src[: n * size]))
        del back, pk
        print("         round trip (encode -> pack -> GPU decode; bytes + XXH3): %s" % ("ok" if rt_ok else "FAILED"), flush=True)
        assert rt_ok
        print("         pack: %.3f ms wall, k_pack_gather %.3f ms = %.0f GB/s (read + write of %.2f GB packed)" %
              (dtp * 1e3, gather_ms, 2 * total_c / (gather_ms * 1e-3) / 1e9, total_c / 1e9), flush=True)
        ok = int((res["status"] == 0).sum())
        print("%-8s %-5s ok %d/%d ratio %.3f  %.1f GiB/s source" % (cname, mname, ok, n, res["comp_size"].sum() / (n * size), n * size / dt / 2**30), flush=True)
