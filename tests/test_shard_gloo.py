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

from zpack_amd.shard import shard_ranges, gather_results

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
    b = dg.Batch(48, 2000, 60000, method=dg.COIN, level=3, seed=9, threads=2)             # the same archive on every rank
    lo, hi = shard_ranges(b.comp_sizes, b.uncomp_sizes, world)[rank]
    local = np.zeros(hi - lo, dtype=np.dtype([("status", "<i4"), ("hash", "<u8")]))
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
    allr = gather_results(local, lo, hi, b.n, rank, world, dist)
    t = torch.tensor([float(hi - lo)])
    dist.all_reduce(t)                                                                    # bookkeeping only, as bench.py does
    if rank == 0:
        q.put((bool((allr["status"] == 0).all()), bool(np.array_equal(allr["hash"], b.hashes)), int(t.item()), b.n))
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
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["parity"]["all_ranks"] is True
    assert d["config"]["entries_total"] == 3000 and 0 < d["config"]["entries_per_gpu"] < 3000
    assert len(d["roofline"]["stage_ms"]) == 3 and d["roofline"]["stage_names"][0] == "k_lz4_wave"
