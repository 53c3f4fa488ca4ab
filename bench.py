#!/usr/bin/env python3
"""bench.py — the batch entry codec on MI355X: GiB/s over all entries + % of HBM roofline, CPU reference beside it.

A "step" is one pass of the hot path over one batch of synthetic entries that is already resident in HBM:
  decode workloads   guards -> per-entry decode -> XXH3 verify (zpk_codec_decode_batch_device)
  c5_zstd1_1m        compress + XXH3 of the source + size scan + compaction (zpk_codec_encode_batch_device + _pack_batch_device)
Default workload = BASELINE.json configs[1]: 100k x 64 KiB LZ4-frame entries (lz4 level 0, seeded 70/20/5/5
text/records/random/runs mix, frames produced by the real liblz4 with the reference writer's call sequence).

  python bench.py --gpus N --steps K --warmup W [--workload ...] [--scaling weak|strong]
  N > 1: one rank per GPU.  Either launched by torch.distributed.run (WORLD_SIZE set: this process IS a rank), or started
  plainly — then this process, BEFORE it touches torch / HIP, starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
  as a CHILD process, relays its JSON line and exit code and exits (never an exec from a process that has initialised the GPU).
  On a box with fewer than N GPUs (the 1-GPU development box) the ranks share cuda:0 over gloo ("rehearsal": RCCL refuses two
  ranks on one device; at most 4 ranks) and the line says so in `config`.  The only communication is the barrier, the
  max-reduce of the timing and — strong scaling — the gather of the per-entry results on rank 0: entries are independent,
  there is no data-path collective.  c4_mixed is ONE archive sharded statically by bytes (zpack_amd/shard.py), 125 000
  entries per GPU = BASELINE.json configs[3]'s 1 000 000 at --gpus 8; every rank generates, holds and uploads ONLY its slice
  (--entries N or --scaling strong fix the archive's total instead).  c5_zstd1_1m is one archive of 12 500 x N source files, rank r
  compresses files [r n, (r + 1) n) and the archive offsets come from a host scan of the per-rank totals.  C2 / C3: one batch per rank.

Prints ONE JSON line (rank 0).  `value` = bytes of all ranks / max-over-ranks wall time of the K timed steps (barrier +
synchronize on both sides).  `roofline` is for the dominant kernel(s), timed with HIP events on the launch stream inside the last
timed step; `cpu_baseline` is the COMPILED REFERENCE (oracle/_ref) on a bounded sample of the same data, at one thread and at all
cores of the box.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (achievable ~6.3 TB/s)

WORKLOADS = {
    # decode: entries, size range, method (-1 = LZ4/Zstandard coin), level, seed, dominant kernel
    "c2_lz4_64k": dict(kind="decode", n=100000, lo=65536, hi=65536, method=2, level=0, seed=1, kernel="lz4"),
    "c3_zstd_256k": dict(kind="decode", n=100000, lo=262144, hi=262144, method=1, level=3, seed=2, kernel="zstd"),
    "c4_mixed": dict(kind="decode", n=125000, lo=4096, hi=1048576, method=-1, level=3, seed=3, kernel="zstd"),
    "stored_64k": dict(kind="decode", n=100000, lo=65536, hi=65536, method=0, level=0, seed=5, kernel="stored"),
    # encode: BASELINE.json configs[4] per GPU (100k x 1 MiB over 8 GPUs = 12 500 per GPU), Zstandard level 1 + XXH3
    "c5_zstd1_1m": dict(kind="encode", n=12500, lo=1 << 20, hi=1 << 20, method=1, level=1, seed=4, kernel="encode"),
}
KNAMES = {"lz4": ["k_lz4_wave", "k_lz4_left"],      # the one-wave decoder; its build for entries of long runs
          "zstd": ["k_zstd_fse", "k_zstd_exec", "k_zstd"], "stored": ["k_stored"], "encode": ["k_encode"]}


def csrc_sha1():
    """identity of the kernel sources: a committed PMC summary describes THIS code only if it carries the same value"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "zpack_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def _ref_driver():
    p = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    return C.CDLL(p) if os.path.exists(p) else None


def cpu_baseline_decode(batch, seconds, ncores):
    """zpack_read_file of the compiled reference (decode + XXH3 verify), one context per thread, at 1 thread and at all cores"""
    n = batch.n
    sample = min(n, max(256, int(2e9 * seconds / 10 / max(1, int(batch.uncomp_sizes[:64].mean())))))
    L = _ref_driver()
    if L is not None:
        L.ref_baseline_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.POINTER(C.c_double)]
        legs = {}
        for threads, secs in ((1, max(2.0, seconds * 0.3)), (ncores, max(3.0, seconds * 0.7))):
            out = (C.c_double * 4)()
            rc = L.ref_baseline_decode(batch.archive.ctypes.data, batch.archive.size, 0, sample if threads > 1 else min(sample, 2048),
                                       threads, float(secs), out)
            if rc != 0 or out[2] != 0:
                legs = None
                break
            legs[threads] = (out[0] / out[1] / 2**30, out[1], out[0] / 2**30)
        if legs:
            v, dt, gib = legs[ncores]
            return dict(value=v, unit="GiB/s", cores=ncores, kind="reference",
                        sample="zpack_read_file (decode + XXH3 verify) of the compiled reference over the first %d entries of the same "
                               "archive, %d threads, looped for %.1f s (%.2f GiB decoded)" % (sample, ncores, dt, gib),
                        one_thread=dict(value=legs[1][0], unit="GiB/s", cores=1,
                                        sample="same call, 1 thread, first %d entries, %.1f s" % (min(sample, 2048), legs[1][1])))
    from tests._libs import oracle
    o = oracle()
    arc = batch.archive.tobytes() if batch.archive.size < (1 << 31) else None
    t0 = time.time()
    done = i = 0
    while time.time() - t0 < seconds and arc is not None:
        k = i % sample
        o.entry_decode(arc, int(batch.offsets[k]), int(batch.comp_sizes[k]), int(batch.uncomp_sizes[k]), int(batch.hashes[k]),
                       int(batch.methods[k]), int(batch.uncomp_sizes[k]))
        done += int(batch.uncomp_sizes[k])
        i += 1
    dt = time.time() - t0
    return dict(value=done / dt / 2**30, unit="GiB/s", cores=1, kind="port",
                sample="oracle/_ref not built; oracle/liboracle.so entry_decode over %d entries, 1 thread, %.1f s" % (i, dt))


def cpu_baseline_encode(plain, entry_size, method, level, seconds, ncores):
    """zpack_write_files of the compiled reference (lib/zpack_write.c:280-343), heap writer per thread, 1 thread and all cores"""
    L = _ref_driver()
    count = plain.size // entry_size
    if L is None or not hasattr(L, "ref_baseline_encode"):
        return dict(value=None, unit="GiB/s", cores=0, kind="port", sample="oracle/_ref not built: no CPU write-path baseline")
    L.ref_baseline_encode.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double)]
    legs = {}
    for threads, secs in ((1, max(2.0, seconds * 0.3)), (ncores, max(3.0, seconds * 0.7))):
        out = (C.c_double * 4)()
        cnt = min(count, 8 * threads if threads > 1 else 16)
        L.ref_baseline_encode(plain.ctypes.data, entry_size, cnt, method, level, threads, float(secs), out)
        legs[threads] = (out[0] / out[1] / 2**30, out[1], out[3] / max(out[0], 1.0), int(out[2]), cnt)
    v, dt, ratio, errs, cnt = legs[ncores]
    return dict(value=v, unit="GiB/s", cores=ncores, kind="reference", ratio=ratio, errors=errs,
                sample="zpack_write_files (compress + XXH3 + append) of the compiled reference over the first %d sources, %d threads, "
                       "looped for %.1f s; compressed/source = %.3f" % (cnt, ncores, dt, ratio),
                one_thread=dict(value=legs[1][0], unit="GiB/s", cores=1, ratio=legs[1][2],
                                sample="same call, 1 thread, first %d sources, %.1f s" % (legs[1][4], legs[1][1])))


def pmc_traffic(workload, n, knames, sha):
    """HBM traffic per launch from the committed rocprofv3 --pmc summary of THIS workload, size and kernel sources (tools/pmc.sh,
    separate FETCH_SIZE / WRITE_SIZE passes); None when no summary matches — never a figure measured on other code."""
    note = "no committed PMC summary for this workload / size / kernel sources (csrc sha1 %s)" % sha[:12]
    try:
        for rd in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True):
            f = os.path.join(ROOT, "profiles", rd, "pmc_%s.json" % workload)
            if not os.path.exists(f):
                continue
            pm = json.load(open(f))
            ks = [pm["kernels"][x] for x in knames if x in pm.get("kernels", {})]
            if pm.get("entries_per_gpu") != n or not ks:
                continue
            if pm.get("csrc_sha1") != sha:
                note = "%s was measured on other kernel sources (csrc sha1 %s, now %s): not quoted" % (
                    os.path.relpath(f, ROOT), str(pm.get("csrc_sha1"))[:12], sha[:12])
                continue
            fs, ws = sum(k["FETCH_SIZE_bytes"] for k in ks), sum(k["WRITE_SIZE_bytes"] for k in ks)
            # MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies 64 B per 128-B request for wide coalesced streaming reads (x2 there); scattered
            # 16-byte gathers are uncalibrated, so both bounds are given; WRITE_SIZE is exact for 16-B-per-lane stores
            return dict(traffic=2.0 * fs + ws, fetch_raw=fs, write=ws, low=fs + ws,
                        note="%s: FETCH_SIZE %.3g B raw (x2 for streaming reads = %.3g) + WRITE_SIZE %.3g B, mean per launch; "
                             "traffic is between %.3g (raw) and %.3g (x2) bytes" % (os.path.relpath(f, ROOT), fs, 2 * fs, ws, fs + ws, 2 * fs + ws))
    except Exception as ex:
        note = "PMC summary unreadable: %s" % ex
    return dict(traffic=None, fetch_raw=None, write=None, low=None, note=note)


def _visible_gpus():
    """number of HIP devices WITHOUT initialising the runtime in this process (the parent only spawns)"""
    n = 0
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        for d in os.listdir(base):
            try:
                props = dict(l.split()[:2] for l in open(os.path.join(base, d, "properties")) if len(l.split()) >= 2)
                if int(props.get("simd_count", "0")) > 0:
                    n += 1
            except OSError:
                pass
    except OSError:
        pass
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if vis is not None and vis.strip() != "":
        n = min(n, len([x for x in vis.split(",") if x.strip() != ""])) if n else len([x for x in vis.split(",") if x.strip() != ""])
    if n == 0:
        # sysfs says nothing (a container without /sys/class/kfd): that is "unknown", not "no GPU" — ask the runtime, in a short-lived
        # CHILD process (this one must not initialise HIP: it only spawns the ranks)
        import subprocess
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, text=True, timeout=300)
            n = int(r.stdout.strip().splitlines()[-1])
        except Exception:
            return None
    return n


def spawn_ranks(n, argv):
    """--gpus N without a launcher: start N ranks as a child torch.distributed.run, relay its output and exit code"""
    import socket
    import subprocess
    have = _visible_gpus()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if have is None:
        have = n                     # unknown: no rehearsal is forced; every rank checks torch.cuda.device_count() itself and fails there
    if have < n:
        if n > 4:
            print("bench.py: --gpus %d asked for, %d GPU(s) visible: a rehearsal on one card is limited to 4 ranks" % (n, have), file=sys.stderr)
            return 2
        env["ZPK_BENCH_REHEARSAL"] = "1"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in p.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            print(l, file=sys.stderr)
    if line is not None:
        print(line)
    return p.returncode if p.returncode != 0 or line is not None else 1


def shard_plan(workload, world, rank, entries=0, scaling=None):
    """What rank `rank` of `world` owns of a workload — derived from sizes alone (a function of the seed and the index), so that every rank
    computes the same plan without building anything, and so that tests/test_shard_gloo.py can check the 8-rank plan of BASELINE.json
    configs[3] / [4] on the CPU.  -> dict(n_total, lo, hi, one_archive, strong, scaling, seed)
      c4_mixed: ONE archive of 125 000 x world entries (1 000 000 at 8 ranks), contiguous CDR ranges balanced by UNCOMPRESSED bytes
                (compressed sizes are not known before compressing; the ratio of a slice of tens of thousands of seeded entries is the
                corpus mean: the comp+uncomp balance of the real archive is checked in tests/test_shard_gloo.py on a sample);
      c2 / c3 / stored: one batch of n entries per rank (weak), seed + 1000 * rank;
      c5_zstd1_1m: one archive of n x world source files, rank r owns files [r n, (r + 1) n)."""
    import numpy as np
    from benchdata import datagen as dg
    from zpack_amd.shard import shard_ranges
    w = dict(WORKLOADS[workload])
    if entries:
        w["n"] = entries
    one_archive = (workload == "c4_mixed" or scaling == "strong") and w["kind"] == "decode"
    fixed_total = scaling == "strong" or (workload == "c4_mixed" and entries > 0)
    if scaling is None:
        scaling = "strong" if (workload == "c4_mixed" and entries > 0) else "weak"
    if one_archive:
        n_total = w["n"] if fixed_total else w["n"] * world
        us_all = dg.sizes(n_total, w["lo"], w["hi"], w["seed"])
        lo, hi = shard_ranges(np.zeros(n_total, dtype=np.uint64), us_all, world)[rank]
        seed = w["seed"]
    elif w["kind"] == "encode":
        n_total = w["n"] * world
        lo, hi = rank * w["n"], (rank + 1) * w["n"]
        seed = w["seed"]
    else:
        n_total = w["n"] * world
        lo, hi = 0, w["n"]
        seed = w["seed"] + 1000 * rank
    return dict(n_total=int(n_total), lo=int(lo), hi=int(hi), one_archive=one_archive, strong=one_archive and fixed_total, scaling=scaling, seed=seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2_lz4_64k", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: every rank decodes its own batch; strong: ONE archive, statically sharded by bytes over the ranks "
                         "(default: strong for c4_mixed, weak otherwise)")
    ap.add_argument("--entries", type=int, default=0, help="override the entry count (debug; the line then names it)")
    ap.add_argument("--mix", type=int, default=-1, help="-1 = 70/20/5/5 class mix, 0..3 = single class")
    ap.add_argument("--cpu-seconds", type=float, default=14.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--skip-hash", action="store_true", help="diagnostic only: status ignores the XXH3 verdict (the line says so)")
    args = ap.parse_args()
    # c4_mixed (BASELINE.json configs[3]) is ONE archive sharded statically over the ranks: 125 000 entries per GPU, i.e. the config's
    # 1 000 000 entries at --gpus 8 (per-GPU work fixed: "weak"); --entries N or --scaling strong fix the archive's TOTAL instead
    scaling_arg = args.scaling
    one_archive = args.workload == "c4_mixed" or args.scaling == "strong"
    fixed_total = args.scaling == "strong" or (args.workload == "c4_mixed" and args.entries > 0)
    if args.scaling is None:
        args.scaling = "strong" if (args.workload == "c4_mixed" and args.entries > 0) else "weak"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])              # nothing above has touched torch or HIP
    if args.gpus > 1 and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
        return 2

    import numpy as np
    import torch
    import zpack_amd
    from benchdata import datagen as dg
    from zpack_amd.shard import shard_ranges, gather_results, archive_bases, gather_segment_totals

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: ZPK_BENCH_REHEARSAL=1 runs every rank on cuda:0 over gloo (RCCL refuses two ranks on one
    # device); the multi-GPU launch of the driver takes the nccl (= RCCL) path
    rehearsal = os.environ.get("ZPK_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not rehearsal and local_rank >= torch.cuda.device_count():
        print("bench.py: rank %d wants cuda:%d, %d device(s) visible" % (rank, local_rank, torch.cuda.device_count()), file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    w = dict(WORKLOADS[args.workload])
    if args.entries:
        w["n"] = args.entries
    one_archive = one_archive and w["kind"] == "decode"
    strong = one_archive and fixed_total
    ncores = len(os.sched_getaffinity(0))
    gen_threads = max(1, ncores // max(1, world))
    codec = zpack_amd.Codec(local_rank)
    if os.environ.get("ZPK_BENCH_ORDER_MIN"):                               # A/B: work lists largest entries first (0 = never)
        codec.set_option(zpack_amd.OPT_ORDER_MIN, int(os.environ["ZPK_BENCH_ORDER_MIN"]))
    if os.environ.get("ZPK_BENCH_ORDER_FAST"):
        codec.set_option(zpack_amd.OPT_ORDER_FAST_LAST, int(os.environ["ZPK_BENCH_ORDER_FAST"]))
    stream = torch.cuda.current_stream().cuda_stream
    sha = csrc_sha1()
    red_dev = torch.device("cpu") if rehearsal else dev

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allsum(vals):
        if world == 1:
            return [float(v) for v in vals]
        t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def allmax(v):
        if world == 1:
            return float(v)
        t = torch.tensor([float(v)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(step):
        for _ in range(args.warmup):
            step()
        barrier()
        codec.set_profiling(True)                               # per-kernel HIP events inside the timed launches themselves
        t0 = time.perf_counter()
        codec.timer_start(stream)
        for _ in range(args.steps):
            step()
        ev_ms = codec.timer_stop(stream)
        barrier()
        wall = time.perf_counter() - t0
        return allmax(wall), ev_ms

    # ---- the copy ceiling of THIS build on THIS box: the stored-entry kernel (copy + XXH3 fused) on a 1 GiB batch ----
    def copy_ceiling():
        sb = dg.Batch(16384, 65536, 65536, method=0, level=0, seed=77, threads=gen_threads)
        sd, stot = zpack_amd.decode_descs_from_batch(sb)
        s_src = torch.from_numpy(sb.archive).to(dev)
        s_dst = torch.empty(stot, dtype=torch.uint8, device=dev)
        s_desc = torch.from_numpy(sd.view(np.uint8)).to(dev)
        s_res = torch.zeros(sb.n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        codec.set_profiling(True)
        ms = []
        for _ in range(4):
            codec.decode_batch_device(s_src, s_desc, sb.n, s_dst, s_res, stream)
            ms.append(codec.kernel_ms(zpack_amd.K_STORED))
        codec.set_profiling(False)
        ok = bool((s_res.cpu().numpy().view(zpack_amd.DECODE_RESULT)["status"] == 0).all())
        return 2.0 * sb.total_uncomp / (min(ms[1:]) * 1e-3) / 1e9 if ok else None

    out = None
    parity = True
    if w["kind"] == "decode":
        # ---- synthetic archive.  One batch per rank (weak scaling of C2 / C3), or ONE archive (c4_mixed, --scaling strong) of which
        # every rank generates, holds and uploads ONLY ITS SLICE: the entries' sizes are a function of the seed and the index, so every
        # rank derives the same byte-balanced contiguous ranges (zpack_amd/shard.py) without building anything, then builds entries
        # [lo, hi) alone ("GPU g gets descriptors + its slice of the packed stream in its own HBM") ----
        t0 = time.time()
        flags = zpack_amd.DF_SKIP_HASH if args.skip_hash else 0
        plan = shard_plan(args.workload, world, rank, args.entries, scaling_arg)
        n_total, lo, hi = plan["n_total"], plan["lo"], plan["hi"]
        assert plan["one_archive"] == one_archive
        if one_archive:
            batch = dg.Batch(hi - lo, w["lo"], w["hi"], method=w["method"], level=w["level"], seed=plan["seed"], mix=args.mix, threads=gen_threads, first=lo)
        else:
            batch = dg.Batch(w["n"], w["lo"], w["hi"], method=w["method"], level=w["level"], seed=plan["seed"], mix=args.mix, threads=gen_threads)
        t_gen = time.time() - t0
        desc, dst_bytes = zpack_amd.decode_descs_from_batch(batch, flags=flags)
        n = batch.n
        src = torch.from_numpy(batch.archive).to(dev)
        dst = torch.empty(max(dst_bytes, 1), dtype=torch.uint8, device=dev)
        ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        dres = torch.zeros(max(n, 1) * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)

        def step():
            codec.decode_batch_device(src, ddesc, n, dst, dres, stream)

        wall, ev_ms = timed(step)
        kern = w["kernel"]
        kids = dict(lz4=[zpack_amd.K_LZ4], zstd=[zpack_amd.K_ZSTD_FSE, zpack_amd.K_ZSTD], stored=[zpack_amd.K_STORED])[kern]
        if w["method"] < 0:                                     # mixed batch: the LZ4 kernel runs too, its time explains ms_per_step
            kids = [zpack_amd.K_LZ4] + kids
        stage_ms = [codec.kernel_ms(k) for k in kids] if n else [0.0]      # the LAST timed launch's kernels
        dstats = codec.decode_stats()
        codec.set_profiling(False)
        my_uncomp = float(batch.uncomp_sizes.sum()) if n else 0.0
        my_comp = float(batch.comp_sizes.sum()) if n else 0.0
        total_uncomp, total_comp, image_bytes_all = allsum([my_uncomp, my_comp, float(len(batch.archive))])

        # ---- parity gate: every status, every XXH3, sizes, and the bytes of a sample against the oracle ----
        res_local = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)[:n]
        # every rank judges its own slice against what ITS generator recorded; one archive: rank 0 also holds the whole archive's
        # results and reference hashes, concatenated in CDR order (the only thing that ever crosses ranks: 24 + 8 bytes per entry)
        bad = int((res_local["status"] != 0).sum())
        hash_ok = bool(np.array_equal(res_local["hash"], batch.hashes))
        size_ok = bool(np.array_equal(res_local["produced"], batch.uncomp_sizes))
        gathered_ok = None
        if one_archive and world > 1:
            res_all = gather_results(res_local, lo, hi, n_total, rank, world, dist)
            ref_all = gather_results(np.ascontiguousarray(batch.hashes), lo, hi, n_total, rank, world, dist)
            if rank == 0:
                us_all = dg.sizes(n_total, w["lo"], w["hi"], plan["seed"])
                gathered_ok = bool((res_all["status"] == 0).all() and np.array_equal(res_all["hash"], ref_all) and
                                   np.array_equal(res_all["produced"], us_all))
        bytes_ok = True
        try:
            from tests._libs import oracle
            o = oracle()
            rng = np.random.default_rng(1234)
            for i in (rng.choice(n, size=min(n, 24), replace=False) if n else []):
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
        parity = bad == 0 and hash_ok and size_ok and bytes_ok is True and gathered_ok is not False
        parity = allsum([0.0 if parity else 1.0])[0] == 0.0
        if not parity:
            print("PARITY FAILURE rank %d: bad_status=%d hash_ok=%s size_ok=%s bytes_ok=%s" % (rank, bad, hash_ok, size_ok, bytes_ok),
                  file=sys.stderr)

        # mixed batches: the LZ4 kernels run on a side stream BESIDE the Zstandard stages (zpk_codec.hip), the stage times overlap —
        # the kernel time of a launch is then the device time of the whole batch (HIP events on the launch stream, all timed steps)
        overlapped = w["method"] < 0
        k_ms = float(ev_ms / args.steps) if overlapped else float(sum(stage_ms))
        alg_bytes = my_comp + my_uncomp                                    # each byte moved once, this rank's launch
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        lz4_names = "k_lz4_wave+k_lz4_left" if dstats.get("lz4_long_runs") else "k_lz4_wave"
        if rank == 0:
            ceiling = copy_ceiling()
            tr = pmc_traffic(args.workload, n, KNAMES[kern] + (KNAMES["lz4"] if w["method"] < 0 else []), sha)      # (a mixed batch: all its decode kernels)
            cpu = None
            if not args.no_cpu and world == 1:                      # the CPU leg is reported at N=1 only
                cpu = cpu_baseline_decode(batch, args.cpu_seconds, ncores)
            out = {
                "metric": "decompressed GiB/s over all entries; % HBM roofline" + (" [DIAGNOSTIC: hash verdict ignored]" if args.skip_hash else ""),
                "value": total_uncomp * args.steps / wall / 2**30, "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
                "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": args.workload, "entries_per_gpu": n, "entries_total": n_total,
                           "archive_image_bytes_this_rank": int(len(batch.archive)), "archive_image_bytes_all_ranks": int(image_bytes_all),
                           "entry_bytes": [w["lo"], w["hi"]],
                           "method": {0: "none", 1: "zstd", 2: "lz4", -1: "lz4+zstd coin"}[w["method"]], "level": w["level"],
                           "class_mix": "70/20/5/5 text/records/random/runs" if args.mix < 0 else ["text", "records", "random", "runs"][args.mix],
                           "comp_ratio": total_comp / max(total_uncomp, 1.0),
                           "parallelism": ("ONE archive of entries_total entries, static contiguous shard by bytes (zpack_amd/shard.py): every rank generates, "
                                            "holds and uploads only its slice; results gathered on rank 0" if one_archive
                                           else "one batch per rank") + ", no data-path collective"
                                          + ("; REHEARSAL: all %d ranks share cuda:0 over gloo (fewer GPUs than ranks on this box)" % world if rehearsal else ""),
                           "rehearsal_one_card": rehearsal,
                           "frames_by": "liblz4/libzstd of the image, reference writer call sequence", "gen_seconds": round(t_gen, 1)},
                "parity": {"all_status_ok": bad == 0, "xxh3_equal_real_xxhash": hash_ok, "sizes_equal": size_ok,
                           "bytes_equal_oracle_sample": bytes_ok, "gathered_whole_archive": gathered_ok, "all_ranks": parity},
                "event_ms_per_step": ev_ms / args.steps,
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "copy_ceiling_gbs": ceiling, "frac_of_copy_ceiling": (achieved / ceiling) if ceiling else None,
                             "copy_ceiling_source": "k_stored (copy + XXH3 fused) of this build on a 1 GiB stored batch, read + write bytes / "
                                                    "HIP-event time, same process",
                             "traffic": tr["traffic"], "traffic_fetch_raw": tr["fetch_raw"], "traffic_write": tr["write"],
                             "traffic_source": tr["note"], "csrc_sha1": sha,
                             "kernel": "+".join(([lz4_names] if w["method"] < 0 else []) + ([lz4_names] if kern == "lz4" else KNAMES[kern])),
                             "kernel_ms": k_ms, "stage_ms": stage_ms, "lz4_entries_of_long_runs": dstats.get("lz4_long_runs"),
                             "stage_names": ([lz4_names] if w["method"] < 0 else []) + {"lz4": [lz4_names],
                                             "zstd": ["k_zstd_fse", "k_zstd_exec+k_zstd"], "stored": ["k_stored"]}[kern],
                             "kernel_ms_source": ("HIP events around the whole batch on the launch stream, mean over the timed steps: the LZ4 stage runs beside the "
                                                  "Zstandard stages, stage_ms overlap") if overlapped else "HIP events around the kernels of the last timed step",
                             "algorithmic_bytes_per_launch": alg_bytes},
                "decode_stats": dstats,
                "cpu_baseline": cpu,
            }
    else:
        # ---- C5: the write path.  Sources resident in HBM; timed = encode batch + size scan + compaction (lib/zpack_write.c:280-343) ----
        n, size, method, level = w["n"], w["lo"], w["method"], w["level"]
        t0 = time.time()
        # ONE archive of n x world source files (BASELINE.json configs[4]: 100 000 x 1 MiB at 8 GPUs); rank r owns files [r n, (r + 1) n):
        # class and bytes are functions of the seed and the GLOBAL index, every rank builds only its own
        first = rank * n
        classes = (np.random.default_rng(w["seed"]).choice(4, size=n * world, p=[0.70, 0.20, 0.05, 0.05])[first:first + n]
                   if args.mix < 0 else np.full(n, args.mix))
        plain = np.empty(n * size, dtype=np.uint8)
        for i in range(n):
            plain[i * size:(i + 1) * size] = dg.fill(int(classes[i]), w["seed"], first + i, size)
        t_gen = time.time() - t0
        bound = codec.compress_bound(method, size)
        slot = (bound + 255) & ~255
        desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
        desc["src_offset"] = np.arange(n, dtype=np.uint64) * size
        desc["size"] = size
        desc["dst_offset"] = np.arange(n, dtype=np.uint64) * slot
        desc["dst_capacity"] = bound
        desc["method"] = method
        desc["level"] = level
        src = torch.from_numpy(plain).to(dev)
        ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        slots = torch.empty(n * slot, dtype=torch.uint8, device=dev)
        dres = torch.zeros(n * zpack_amd.ENCODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        packed = torch.empty(n * bound + 64, dtype=torch.uint8, device=dev)

        def step():
            codec.encode_batch_device(src, ddesc, n, slots, dres, stream)
            codec.pack_batch_device(slots, ddesc, dres, n, packed, offs, int(bound), stream)

        wall, ev_ms = timed(step)
        stage_ms = [codec.kernel_ms(zpack_amd.K_ENCODE), codec.kernel_ms(zpack_amd.K_PACK)]
        codec.set_profiling(False)
        res = dres.cpu().numpy().view(zpack_amd.ENCODE_RESULT)
        ho = offs.cpu().numpy().view(np.uint64)
        total_c = int(ho[-1])
        ok_status = bool((res["status"] == 0).all())
        # ---- parity gate: every frame decoded back by the GPU decoder (bytes + XXH3), a sample by the oracle, hashes vs real xxHash ----
        dd = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
        dd["src_offset"] = ho[:-1]; dd["comp_size"] = res["comp_size"]; dd["uncomp_size"] = size; dd["expect_hash"] = res["hash"]
        dd["dst_offset"] = np.arange(n, dtype=np.uint64) * size; dd["dst_capacity"] = size; dd["method"] = method
        back = torch.empty(n * size, dtype=torch.uint8, device=dev)
        r2 = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        codec.decode_batch_device(packed[:total_c + 64], torch.from_numpy(dd.view(np.uint8)).to(dev), n, back, r2, stream)
        torch.cuda.synchronize()
        rr = r2.cpu().numpy().view(zpack_amd.DECODE_RESULT)
        rt_ok = bool((rr["status"] == 0).all() and np.array_equal(rr["hash"], res["hash"]) and torch.equal(back, src))
        hash_ok = all(int(res["hash"][i]) == dg.xxh3(plain[i * size:(i + 1) * size]) for i in range(0, n, max(1, n // 24)))
        orc_ok = True
        try:
            from tests._libs import oracle
            o = oracle()
            hp = packed[:total_c].cpu().numpy()
            for i in range(0, n, max(1, n // 8)):
                fr = hp[int(ho[i]):int(ho[i]) + int(res["comp_size"][i])].tobytes()
                rc, outb = (o.zstd_decode if method == 1 else o.lz4f_decode)(fr, size)
                if rc != 0 or outb != plain[i * size:(i + 1) * size].tobytes():
                    orc_ok = False
        except Exception as ex:
            orc_ok = "oracle unavailable: %s" % ex
        # ---- the archive's offsets across ranks: rank r's packed segment starts where the segments of ranks < r end (ONE host scan of
        # `world` totals, zpack_amd/shard.py); rank 0 gathers the entry table (offset, comp_size: 16 bytes per entry) and checks the
        # chain lib/zpack_write.c:338 builds serially: first payload at 10, every payload where its predecessor ends ----
        totals = gather_segment_totals(total_c, rank, world, dist)
        bases, data_end = archive_bases(totals)
        arch_off = (ho[:-1] + bases[rank]).astype(np.uint64)
        chain_ok = True
        if world > 1:
            tab = np.zeros(n, dtype=np.dtype([("offset", "<u8"), ("comp_size", "<u8")]))
            tab["offset"] = arch_off; tab["comp_size"] = res["comp_size"]
            tab_all = gather_results(tab, first, first + n, n * world, rank, world, dist)
            if rank == 0:
                chain_ok = bool(tab_all["offset"][0] == 10 and np.array_equal(tab_all["offset"][1:], tab_all["offset"][:-1] + tab_all["comp_size"][:-1])
                                and int(tab_all["offset"][-1] + tab_all["comp_size"][-1]) == data_end)
        else:
            chain_ok = bool(arch_off[0] == 10 and np.array_equal(arch_off[1:], arch_off[:-1] + res["comp_size"][:-1].astype(np.uint64)))
        parity = ok_status and rt_ok and hash_ok and orc_ok is True and chain_ok
        parity = allsum([0.0 if parity else 1.0])[0] == 0.0
        total_src, total_comp = allsum([float(n) * size, float(total_c)])
        k_ms = float(stage_ms[0])
        alg_bytes = float(n) * size + float(total_c)                        # source read once + compressed written once
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        if rank == 0:
            ceiling = copy_ceiling()
            tr = pmc_traffic(args.workload, n, KNAMES["encode"], sha)
            cpu = None
            if not args.no_cpu and world == 1:
                cpu = cpu_baseline_encode(plain, size, method, level, args.cpu_seconds, ncores)
            out = {
                "metric": "compressed-source GiB/s (zpack_write path: Zstd-1 + XXH3); % HBM roofline",
                "value": total_src * args.steps / wall / 2**30, "unit": "GiB/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u8", "data": "synthetic",
                "config": {"workload": args.workload, "entries_per_gpu": n, "entries_total": n * world, "archive_data_section_bytes": data_end - 10,
                           "entry_bytes": [size, size], "method": "zstd" if method == 1 else "lz4",
                           "level": level, "class_mix": "70/20/5/5 text/records/random/runs" if args.mix < 0 else ["text", "records", "random", "runs"][args.mix],
                           "comp_ratio": total_comp / total_src, "reference_ratio": (cpu or {}).get("ratio"),
                           "timed": "zpk_codec_encode_batch_device + zpk_codec_pack_batch_device (size scan + compaction = write_offset += comp_size)",
                           "parallelism": "ONE archive of entries_total source files, rank r compresses and compacts files [r n, (r + 1) n); archive offsets from a "
                                          "host scan of the per-rank segment totals (zpack_amd/shard.py); no data-path collective"
                                          + ("; REHEARSAL: all %d ranks share cuda:0 over gloo" % world if rehearsal else ""),
                           "rehearsal_one_card": rehearsal, "gen_seconds": round(t_gen, 1)},
                "parity": {"all_status_ok": ok_status, "gpu_decoder_round_trip_bytes_and_xxh3": rt_ok, "xxh3_equal_real_xxhash_sample": hash_ok,
                           "oracle_decodes_sample": orc_ok, "archive_offset_chain_across_ranks": chain_ok, "all_ranks": parity},
                "event_ms_per_step": ev_ms / args.steps,
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "copy_ceiling_gbs": ceiling, "frac_of_copy_ceiling": (achieved / ceiling) if ceiling else None,
                             "copy_ceiling_source": "k_stored (copy + XXH3 fused) of this build on a 1 GiB stored batch, same process",
                             "traffic": tr["traffic"], "traffic_fetch_raw": tr["fetch_raw"], "traffic_write": tr["write"],
                             "traffic_source": tr["note"], "csrc_sha1": sha, "kernel": "k_encode", "kernel_ms": k_ms,
                             "stage_ms": stage_ms, "stage_names": ["k_encode", "k_pack_gather"],
                             "kernel_ms_source": "HIP events around the kernels of the last timed step",
                             "algorithmic_bytes_per_launch": alg_bytes},
                "cpu_baseline": cpu,
            }
    if rank == 0 and out is not None:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if parity else 1


if __name__ == "__main__":
    sys.exit(main())
