#!/usr/bin/env python3
"""bench.py — batch entry decode on MI355X: decompressed GiB/s over all entries + % of HBM roofline.

A "step" is one pass of the hot path (guards -> per-entry decode -> fused XXH3 verify) over one batch
of synthetic archive entries that is already resident in HBM.  Default workload = BASELINE.json
configs[1]: 100k x 64 KiB LZ4-frame entries (lz4 level 0, seeded 70/20/5/5 text/records/random/runs
mix, frames produced by the real liblz4 with the reference writer's call sequence).

  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  `value` = decompressed bytes of all ranks / max-over-ranks wall time of
the K timed steps (barrier + synchronize on both sides).  `roofline` is for the dominant kernel,
measured with HIP events on the launch stream; `cpu_baseline` is the compiled reference
(oracle/_ref, zpack_read_file per entry, one context per thread) on a bounded sample of the same archive.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (achievable ~6.3 TB/s)

WORKLOADS = {
    # name: (entries, size_lo, size_hi, method, level, seed, dominant kernel)
    "c2_lz4_64k": dict(n=100000, lo=65536, hi=65536, method=2, level=0, seed=1, kernel="lz4"),
    "c3_zstd_256k": dict(n=100000, lo=262144, hi=262144, method=1, level=3, seed=2, kernel="zstd"),
    "c4_mixed": dict(n=125000, lo=4096, hi=1048576, method=-1, level=3, seed=3, kernel="zstd"),
    "stored_64k": dict(n=100000, lo=65536, hi=65536, method=0, level=0, seed=5, kernel="stored"),
}


def cpu_baseline(batch, seconds, threads):
    """The reference's own read path on the host cores (kind 'reference'); oracle port if _ref is absent."""
    import numpy as np
    ref_drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    n = batch.n
    # bounded sample: the first entries up to ~1 GiB decompressed per thread-second budget
    sample = min(n, max(256, int(2e9 * seconds / 10 / max(1, int(batch.uncomp_sizes[:64].mean())))))
    if os.path.exists(ref_drv):
        L = C.CDLL(ref_drv)
        L.ref_baseline_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.POINTER(C.c_double)]
        out = (C.c_double * 4)()
        rc = L.ref_baseline_decode(batch.archive.ctypes.data, batch.archive.size, 0, sample, threads, float(seconds), out)
        if rc == 0 and out[2] == 0:
            return dict(value=out[0] / out[1] / 2**30, unit="GiB/s", cores=threads, kind="reference",
                        sample="zpack_read_file (decode + XXH3 verify) of the compiled reference over the first %d entries of "
                               "the same archive, %d thread(s), looped for %.1f s (%.2f GiB decoded)" %
                               (sample, threads, out[1], out[0] / 2**30))
        why = "reference driver rc=%d errors=%d; " % (rc, int(out[2]))
    else:
        why = "oracle/_ref not built; "
    from tests._libs import oracle
    o = oracle()
    arc = batch.archive.tobytes() if batch.archive.size < (1 << 31) else None
    t0 = time.time()
    done = 0
    i = 0
    while time.time() - t0 < seconds and arc is not None:
        k = i % sample
        o.entry_decode(arc, int(batch.offsets[k]), int(batch.comp_sizes[k]), int(batch.uncomp_sizes[k]), int(batch.hashes[k]),
                       int(batch.methods[k]), int(batch.uncomp_sizes[k]))
        done += int(batch.uncomp_sizes[k])
        i += 1
    dt = time.time() - t0
    return dict(value=done / dt / 2**30, unit="GiB/s", cores=1, kind="port",
                sample=why + "oracle/liboracle.so entry_decode over %d entries, 1 thread, %.1f s" % (i, dt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2_lz4_64k", choices=sorted(WORKLOADS))
    ap.add_argument("--entries", type=int, default=0, help="override the entry count (debug; the line then names it)")
    ap.add_argument("--mix", type=int, default=-1, help="-1 = 70/20/5/5 class mix, 0..3 = single class")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--lz4-ring", action="store_true", help="LZ4 entries through the scan + LDS-ring executor first (zpk_codec_set_option ZPK_OPT_LZ4_RING)")
    ap.add_argument("--skip-hash", action="store_true", help="diagnostic only: decode without the XXH3 verify (the line says so)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import zpack_amd
    from benchdata import datagen as dg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: ZPK_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo (RCCL refuses two ranks on one
    # device); the multi-GPU launch of the driver takes the nccl (= RCCL) path
    rehearsal = os.environ.get("ZPK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    w = dict(WORKLOADS[args.workload])
    if args.entries:
        w["n"] = args.entries
    ncores = len(os.sched_getaffinity(0))
    gen_threads = max(1, ncores // max(1, world))

    # ---- synthetic archive: independent entries; weak scaling = every rank decodes its own batch ----
    t0 = time.time()
    batch = dg.Batch(w["n"], w["lo"], w["hi"], method=w["method"], level=w["level"], seed=w["seed"] + 1000 * rank,
                     mix=args.mix, threads=gen_threads)
    t_gen = time.time() - t0
    desc, dst_bytes = zpack_amd.decode_descs_from_batch(batch, flags=zpack_amd.DF_SKIP_HASH if args.skip_hash else 0)
    n = batch.n

    codec = zpack_amd.Codec(local_rank)
    if args.lz4_ring:
        codec.set_option(zpack_amd.OPT_LZ4_RING, 1)
    src = torch.from_numpy(batch.archive).to(dev)
    dst = torch.empty(dst_bytes, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        codec.decode_batch_device(src, ddesc, n, dst, dres, stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    codec.timer_start(stream)
    for _ in range(args.steps):
        step()
    ev_ms = codec.timer_stop(stream)
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        tot = torch.tensor([float(batch.total_uncomp), float(batch.total_comp)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_uncomp, total_comp = float(tot[0].item()), float(tot[1].item())
    else:
        total_uncomp, total_comp = float(batch.total_uncomp), float(batch.total_comp)

    # ---- parity gate: every status, every XXH3, and bytes of a sample against the oracle ----
    res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
    bad = int((res["status"] != 0).sum())
    hash_ok = bool(np.array_equal(res["hash"], batch.hashes)) or args.skip_hash
    size_ok = bool(np.array_equal(res["produced"], batch.uncomp_sizes))
    bytes_ok = True
    try:
        from tests._libs import oracle
        o = oracle()
        rng = np.random.default_rng(1234)
        for i in rng.choice(n, size=min(n, 24), replace=False):
            d = desc[i]
            off, cs = int(d["src_offset"]), int(d["comp_size"])
            arc = bytes(b"\0" * 10) + batch.archive[off:off + cs].tobytes() + b"\0"
            rc, want, got, h = o.entry_decode(arc, 10, cs, int(d["uncomp_size"]), int(d["expect_hash"]), int(d["method"]),
                                              int(d["dst_capacity"]))
            have = dst[int(d["dst_offset"]):int(d["dst_offset"]) + int(d["uncomp_size"])].cpu().numpy().tobytes()
            if rc != 0 or have != want:
                bytes_ok = False
    except Exception as ex:                                   # the checker is optional at run time, the hash gate is not
        bytes_ok = "oracle unavailable: %s" % ex
    parity = bad == 0 and hash_ok and size_ok and bytes_ok is True
    if not parity:
        print("PARITY FAILURE rank %d: bad_status=%d hash_ok=%s size_ok=%s bytes_ok=%s" % (rank, bad, hash_ok, size_ok, bytes_ok),
              file=sys.stderr)

    # ---- dominant-kernel time, HIP events on the launch stream ----
    # (Zstandard runs as a pipeline: k_zstd_fse pre-decodes the sequence streams, k_zstd_exec + k_zstd do the rest;
    # its figure is the sum of the stages, bracketed on the stream one after the other)
    kids = dict(lz4=[zpack_amd.K_LZ4_SCAN, zpack_amd.K_LZ4], zstd=[zpack_amd.K_ZSTD_FSE, zpack_amd.K_ZSTD], stored=[zpack_amd.K_STORED])[w["kernel"]]
    codec.set_profiling(True)
    kms, stage_ms = [], None
    for _ in range(max(3, min(args.steps, 10))):
        step()
        stage_ms = [codec.kernel_ms(k) for k in kids]
        kms.append(sum(stage_ms))
    codec.set_profiling(False)
    dstats = codec.decode_stats()
    k_ms = float(np.mean(kms))
    alg_bytes = float(batch.total_comp + batch.total_uncomp)            # each byte moved once
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel: hardware counters cannot be read from inside this process, so the
    # figure comes from the committed rocprofv3 --pmc summary of THIS workload at THIS size (tools/pmc.sh,
    # separate FETCH_SIZE / WRITE_SIZE passes); null when no matching measurement is committed.
    traffic, traffic_note = None, "no committed PMC summary for this workload/size"
    knames = {"lz4": ["k_lz4_wave"], "zstd": ["k_zstd_fse", "k_zstd_exec", "k_zstd"], "stored": ["k_stored"]}[w["kernel"]]
    kname = "+".join(knames)
    try:
        for rd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            f = os.path.join(ROOT, "profiles", rd, "pmc_%s.json" % args.workload)
            if os.path.exists(f):
                pm = json.load(open(f))
                ks = [pm["kernels"][x] for x in knames if x in pm.get("kernels", {})]
                if pm.get("entries_per_gpu") == n and ks:
                    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 (x2); WRITE_SIZE is exact
                    fs, ws = sum(k["FETCH_SIZE_bytes"] for k in ks), sum(k["WRITE_SIZE_bytes"] for k in ks)
                    traffic = 2.0 * fs + ws
                    traffic_note = "%s: 2 x FETCH_SIZE (%.3g B raw) + WRITE_SIZE (%.3g B), mean per launch" % (os.path.relpath(f, ROOT), fs, ws)
                break
    except Exception as ex:
        traffic_note = "PMC summary unreadable: %s" % ex

    if rank == 0:
        cpu = None
        if not args.no_cpu and world == 1:                      # the CPU leg is reported at N=1 only
            cpu = cpu_baseline(batch, args.cpu_seconds, ncores)
        out = {
            "metric": "decompressed GiB/s over all entries; % HBM roofline" + (" [DIAGNOSTIC: hash skipped]" if args.skip_hash else ""),
            "value": total_uncomp * args.steps / wall / 2**30,
            "unit": "GiB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": args.workload, "entries_per_gpu": n, "entry_bytes": [w["lo"], w["hi"]],
                       "method": {0: "none", 1: "zstd", 2: "lz4", -1: "lz4+zstd coin"}[w["method"]], "level": w["level"],
                       "class_mix": "70/20/5/5 text/records/random/runs" if args.mix < 0 else ["text", "records", "random", "runs"][args.mix],
                       "comp_ratio": total_comp / total_uncomp, "parallelism": "static shard, no collectives",
                       "frames_by": "liblz4/libzstd of the image, reference writer call sequence", "gen_seconds": round(t_gen, 1)},
            "parity": {"all_status_ok": bad == 0, "xxh3_equal_real_xxhash": hash_ok, "sizes_equal": size_ok,
                       "bytes_equal_oracle_sample": bytes_ok},
            "event_ms_per_step": ev_ms / args.steps,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note, "kernel": kname,
                         "kernel_ms": k_ms, "stage_ms": stage_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "decode_stats": dstats,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if parity else 1


if __name__ == "__main__":
    sys.exit(main())
