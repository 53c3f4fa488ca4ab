"""Multi-GPU path on CPU: static sharding of one archive across ranks (world_size 2, gloo) — every entry is
owned by exactly one rank, the byte balance is even, and the per-rank result arrays concatenate in CDR
order on rank 0.  No data-path collective exists; the only communication is the gather of results."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zpack_amd.shard import shard_ranges, gather_results, archive_bases, gather_segment_totals

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_and_balance():
    rng = np.random.default_rng(0)
    for n, world in ((1000, 8), (7, 8), (0, 4), (100000, 8), (5, 1)):
        us = np.exp(rng.uniform(np.log(4096), np.log(1 << 20), n)).astype(np.uint64)      # log-uniform like config C4
        cs = (us * rng.uniform(0.03, 1.0, n)).astype(np.uint64)
        r = shard_ranges(cs, us, world)
        assert len(r) == world and r[0][0] == 0 and r[-1][1] == n
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1)) and all(lo <= hi for lo, hi in r)
        if n >= 100 * world:
            w = (cs + us).astype(np.float64)
            loads = np.array([w[lo:hi].sum() for lo, hi in r])
            assert loads.max() / loads.mean() < 1.05


def _worker(rank, world, port, q, use_gpu=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from benchdata import datagen as dg
    # ONE archive (seed 9, 48 entries); every rank derives the byte-balanced ranges from the entries' sizes alone and BUILDS ONLY ITS
    # SLICE ("GPU g gets descriptors + its slice of the packed stream") — as bench.py does for c4_mixed
    n_total = 48
    us_all = dg.sizes(n_total, 2000, 60000, 9)
    lo, hi = shard_ranges(np.zeros(n_total, dtype=np.uint64), us_all, world)[rank]
    b = dg.Batch(hi - lo, 2000, 60000, method=dg.COIN, level=3, seed=9, threads=2, first=lo)
    whole = dg.Batch(n_total, 2000, 60000, method=dg.COIN, level=3, seed=9, threads=2) if rank == 0 else None      # (rank 0 only, as the checker's reference)
    local = np.zeros(hi - lo, dtype=np.dtype([("status", "<i4"), ("hash", "<u8")]))
    lo_g, hi_g = lo, hi
    lo, hi = 0, b.n                                                                         # indices into the slice
    if use_gpu:
        # the PRODUCT: this rank's slice through zpk_codec_decode_batch_host of its own codec context (both ranks share the one
        # card of the test box; on a node every rank has its own)
        import zpack_amd
        codec = zpack_amd.Codec(0)
        d = np.zeros(hi - lo, dtype=zpack_amd.DECODE_DESC)
        d["src_offset"] = b.offsets[lo:hi]; d["comp_size"] = b.comp_sizes[lo:hi]; d["uncomp_size"] = b.uncomp_sizes[lo:hi]
        d["expect_hash"] = b.hashes[lo:hi]; d["dst_capacity"] = b.uncomp_sizes[lo:hi]; d["method"] = b.methods[lo:hi]
        res, _ = codec.decode_batch_host(b.archive, d)
        local["status"] = res["status"]; local["hash"] = res["hash"]
        codec.close()
    else:
        # CPU container (no GPU, and the product has no CPU fallback): the checker stands in for the codec so that the
        # partition / gather logic is still exercised; test_two_rank_static_shard_product below runs the real codec
        from tests._libs import oracle
        o = oracle()
        arc = b.archive.tobytes()
        for k, i in enumerate(range(lo, hi)):
            rc, out, got, h = o.entry_decode(arc, int(b.offsets[i]), int(b.comp_sizes[i]), int(b.uncomp_sizes[i]), int(b.hashes[i]),
                                             int(b.methods[i]), int(b.uncomp_sizes[i]))
            local[k] = (rc, h)
    allr = gather_results(local, lo_g, hi_g, n_total, rank, world, dist)
    t = torch.tensor([float(hi - lo), float(len(b.archive))])
    dist.all_reduce(t)                                                                    # bookkeeping only, as bench.py does
    if rank == 0:
        # the slices together hold the data section once: each rank's image is its share of the whole archive, not a copy of it
        slice_share_ok = len(b.archive) < 0.75 * len(whole.archive) and t[1].item() < 1.25 * len(whole.archive)
        q.put((bool((allr["status"] == 0).all()) and slice_share_ok, bool(np.array_equal(allr["hash"], whole.hashes)), int(t[0].item()), n_total))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(use_gpu):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + (2000 if use_gpu else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    ok_status, ok_hash, covered, n = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok_status and ok_hash and covered == n


def test_two_rank_static_shard_gloo():
    _two_ranks(False)


@pytest.mark.gpu
def test_two_rank_static_shard_product():
    """the same two ranks, every slice decoded by the HIP codec (C-ABI), results gathered on rank 0"""
    _two_ranks(True)


@pytest.mark.gpu
def test_bench_gpus2_spawns_its_own_ranks():
    """`python bench.py --gpus 2 --workload c4_mixed` as the driver runs it: the parent starts two ranks itself (a child
    torch.distributed.run), strong scaling over ONE archive, parity on the gathered results"""
    import json
    import subprocess
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4_mixed", "--entries", "3000",
                        "--steps", "2", "--warmup", "1", "--no-cpu"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["parity"]["all_ranks"] is True and d["parity"]["gathered_whole_archive"] is True
    assert d["config"]["entries_total"] == 3000 and 0 < d["config"]["entries_per_gpu"] < 3000
    # every rank generated and uploaded ITS SLICE: rank 0's image is about half of what the two images hold together
    assert 0.3 < d["config"]["archive_image_bytes_this_rank"] / d["config"]["archive_image_bytes_all_ranks"] < 0.7
    assert len(d["roofline"]["stage_ms"]) == 3 and d["roofline"]["stage_names"][0].startswith("k_lz4_wave")


def _write_worker(rank, world, port, q):
    """the write path over two ranks (BASELINE.json configs[4] in miniature): rank r compresses source files [r n, (r + 1) n) of ONE archive,
    packs them into its segment; the archive offsets come from a host scan of the per-rank totals; rank 0 concatenates the segments
    into one .zpk and every entry of it decodes (the checker) to its source."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from benchdata import datagen as dg
    from tests import zpk
    from tests._libs import oracle
    n, size = 9, 30000
    first = rank * n
    frames, plains = [], []
    for i in range(n):
        plain = dg.fill((first + i) % 4, 4, first + i, size)
        plains.append(plain)
        frames.append(dg.compress(dg.ZSTD, 1, plain))                                     # (the CPU stands in for the encoder here: no GPU in this test)
    seg = b"".join(frames)
    totals = gather_segment_totals(len(seg), rank, world, dist)
    bases, data_end = archive_bases(totals)
    tab = np.zeros(n, dtype=np.dtype([("offset", "<u8"), ("comp_size", "<u8"), ("hash", "<u8")]))
    tab["comp_size"] = [len(f) for f in frames]
    tab["offset"] = bases[rank] + np.concatenate([[0], np.cumsum(tab["comp_size"])[:-1]]).astype(np.uint64)
    tab["hash"] = [dg.xxh3(p) for p in plains]
    tab_all = gather_results(tab, first, first + n, n * world, rank, world, dist)
    segs = gather_results(np.frombuffer(seg, dtype=np.uint8), int(bases[rank]) - 10, int(bases[rank]) - 10 + len(seg), data_end - 10, rank, world, dist)
    if rank == 0:
        chain = bool(tab_all["offset"][0] == 10 and np.array_equal(tab_all["offset"][1:], tab_all["offset"][:-1] + tab_all["comp_size"][:-1])
                     and int(tab_all["offset"][-1] + tab_all["comp_size"][-1]) == data_end)
        arc = zpk.assemble([segs.tobytes()], [("f%d" % i, int(tab_all["offset"][i]), int(tab_all["comp_size"][i]), size, int(tab_all["hash"][i]), 1)
                                             for i in range(n * world)])
        o = oracle()
        ok = True
        for i in range(n * world):
            rc, out, got, h = o.entry_decode(arc, int(tab_all["offset"][i]), int(tab_all["comp_size"][i]), size, int(tab_all["hash"][i]), 1, size)
            ok = ok and rc == 0 and out == dg.fill(i % 4, 4, i, size).tobytes()
        q.put((chain, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_write_path_archive_offsets_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_write_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    chain, ok = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert chain and ok


# ---- world_size 8 on the CPU: the plans of BASELINE.json configs[3] and [4] as bench.py --gpus 8 would run them (sizes only, no archive) --------

def _plan8_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from benchdata import datagen as dg
    # configs[3]: ONE archive of 125 000 x 8 = 1 000 000 entries, 4 KiB .. 1 MiB, sharded by bytes; every rank derives its range alone
    plan = bench.shard_plan("c4_mixed", world, rank)
    w = bench.WORKLOADS["c4_mixed"]
    us = dg.sizes(plan["n_total"], w["lo"], w["hi"], w["seed"])
    lo, hi = plan["lo"], plan["hi"]
    # what a rank would gather: one record per entry of its slice (here: the entry's global index and size — the stand-in for the 24-byte
    # results; the real codec runs in the 2-rank tests above), in CDR order on rank 0
    local = np.zeros(hi - lo, dtype=np.dtype([("index", "<u8"), ("size", "<u8")]))
    local["index"] = np.arange(lo, hi, dtype=np.uint64)
    local["size"] = us[lo:hi]
    allr = gather_results(local, lo, hi, plan["n_total"], rank, world, dist)
    mine = torch.tensor([float(lo), float(hi), float(us[lo:hi].astype(np.float64).sum())], dtype=torch.float64)
    every = [torch.zeros(3, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(every, mine)
    # configs[4]: one archive of 12 500 x 8 source files; rank r compresses files [r n, (r + 1) n) and the archive offsets come from a host
    # scan of the per-rank segment totals (lib/zpack_write.c:338 across ranks).  Stand-in totals: a seeded "compressed size" per file.
    p5 = bench.shard_plan("c5_zstd1_1m", world, rank)
    rng = np.random.default_rng(1234)
    comp = rng.integers(300_000, 500_000, p5["n_total"], dtype=np.uint64)          # the same table on every rank
    seg_total = int(comp[p5["lo"]:p5["hi"]].sum())
    totals = gather_segment_totals(seg_total, rank, world, dist)
    bases, end = archive_bases(totals)
    local_off = np.uint64(bases[rank]) + np.concatenate([[np.uint64(0)], np.cumsum(comp[p5["lo"]:p5["hi"]])[:-1]]).astype(np.uint64)
    off_all = gather_results(local_off, p5["lo"], p5["hi"], p5["n_total"], rank, world, dist)
    if rank == 0:
        ranges = [(int(e[0]), int(e[1])) for e in every]
        loads = np.array([float(e[2]) for e in every])
        cover = ranges[0][0] == 0 and ranges[-1][1] == plan["n_total"] and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
        order_ok = bool(np.array_equal(allr["index"], np.arange(plan["n_total"], dtype=np.uint64)) and np.array_equal(allr["size"], us))
        want_off = np.uint64(10) + np.concatenate([[np.uint64(0)], np.cumsum(comp)[:-1]]).astype(np.uint64)
        q.put(dict(n_total=plan["n_total"], scaling=plan["scaling"], cover=cover, imbalance=float(loads.max() / loads.mean()), order_ok=order_ok,
                   c5_total=p5["n_total"], c5_chain_ok=bool(np.array_equal(off_all, want_off)) and end == 10 + int(comp.sum()),
                   c5_ranges_ok=all(bench.shard_plan("c5_zstd1_1m", world, r)["lo"] == r * 12500 for r in range(world))))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_plans_of_configs_3_and_4_gloo():
    """BASELINE.json configs[3] = 1 M mixed entries over 8 GPUs, configs[4] = 100 k x 1 MiB sources over 8 GPUs.  No 8-GPU node has been
    available to any round; this runs the PLAN bench.py --gpus 8 executes (bench.shard_plan, zpack_amd/shard.py) on 8 gloo ranks with
    the real size table of the config: the ranges tile [0, 1 000 000), bytes per rank within 5 % of the mean, results gathered in CDR
    order on rank 0, the line would say entries_total = 1 000 000; the write path's offset chain across 8 ranks equals the serial
    `write_offset += comp_size` (lib/zpack_write.c:338)."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_plan8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    r = q.get(timeout=300)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert r["n_total"] == 1_000_000 and r["scaling"] == "weak" and r["cover"] and r["order_ok"], r
    assert r["imbalance"] < 1.05, r
    assert r["c5_total"] == 100_000 and r["c5_chain_ok"] and r["c5_ranges_ok"], r


def test_c4_balance_by_uncompressed_bytes_is_good_enough():
    """bench.py balances configs[3] by UNCOMPRESSED bytes (compressed sizes are not known before compressing).  On a real sample of the
    corpus (4 000 entries, both methods, all classes) the comp+uncomp load of 8 such ranges stays within 8 % of the mean."""
    from benchdata import datagen as dg
    b = dg.Batch(4000, 4096, 1 << 20, method=dg.COIN, level=3, seed=3, threads=4)
    r = shard_ranges(np.zeros(b.n, dtype=np.uint64), b.uncomp_sizes, 8)
    w = (b.comp_sizes + b.uncomp_sizes).astype(np.float64)
    loads = np.array([w[lo:hi].sum() for lo, hi in r])
    assert loads.max() / loads.mean() < 1.08, loads
