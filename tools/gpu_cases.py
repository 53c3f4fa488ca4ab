#!/usr/bin/env python3
"""Run GPU parity cases one per subprocess with a hard timeout each (a hung kernel must not eat the
whole GPU allocation).  Usage: gpu_cases.py            -> driver, runs every case
                                gpu_cases.py CASE       -> run one case in-process"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


def run_case(name):
    import numpy as np
    import zpack_amd
    from benchdata import datagen as dg
    from tests import zpk
    from tests.test_gpu_codec import _desc
    codec = zpack_amd.Codec(0)
    kind, _, arg = name.partition(":")
    if kind == "ref":
        a = open(os.path.join(G, "ref_workdir", arg), "rb").read()
        ents = zpk.parse(a)
        res, outs = codec.decode_batch_host(a, _desc(ents, [350, 350]))
        for e, r, out in zip(ents, res, outs):
            plain = open(os.path.join(G, "ref_workdir", e["filename"]), "rb").read()
            assert r["status"] == 0 and int(r["hash"]) == e["hash"], (r, e)
            assert out[:len(plain)].tobytes() == plain
    elif kind == "small":
        corpus, label = arg.split("/")
        for case in json.load(open(os.path.join(G, "small_archives.json"))):
            if case["corpus"] != corpus or case["label"] != label:
                continue
            a = bytes.fromhex(case["archive"])
            ents = zpk.parse(a)
            res, outs = codec.decode_batch_host(a, _desc(ents, [e["uncomp_size"] for e in ents]))
            for e, size, r, out in zip(ents, case["sizes"], res, outs):
                plain = dg.fill(case["cls"], case["seed"], size, size).tobytes()
                assert r["status"] == 0, (size, r)
                assert out[:size].tobytes() == plain, size
    elif kind == "status":
        sc = json.load(open(os.path.join(G, "status_cases.json")))
        for c in sc["cases"]:
            if not c["label"].startswith(arg):
                continue
            a = bytearray(bytes.fromhex(sc["bases"][c["base"]]))
            for p, x in c["flips"]:
                a[p] ^= x
            e = zpk.parse(a)[c["index"]]
            for k, v in c["tamper"].items():
                e[{"comp_method": "method"}.get(k, k)] = v
            res, outs = codec.decode_batch_host(bytes(a), _desc([e], [c["max_size"]]))
            assert int(res[0]["status"]) == c["rc"], (c["label"], res[0], c["rc"])
    elif kind == "foreign":
        for c in json.load(open(os.path.join(G, "foreign_frames.json"))):
            if not c["label"].startswith(arg):
                continue
            fr = bytes.fromhex(c["frame"])
            e = dict(offset=10, comp_size=len(fr), uncomp_size=c["uncomp_size"], hash=c["hash"], method=c["method"])
            arc = zpk.assemble([fr], [("f", 10, len(fr), c["uncomp_size"], c["hash"], c["method"])])
            t = time.time()
            res, outs = codec.decode_batch_host(arc, _desc([e], [c["max_size"]]))
            print("   ", c["label"], "status", int(res[0]["status"]), "detail", hex(int(res[0]["detail"])), "%.3fs" % (time.time() - t), flush=True)
            assert int(res[0]["status"]) == c["rc"], (c["label"], res[0])
            if c["rc"] == 0:
                assert dg.xxh3(outs[0][:c["uncomp_size"]]) == c["plain_xxh3"], c["label"]
    elif kind == "batch":
        import torch
        method, level, size, n, mix = [int(x) for x in arg.split(",")]
        b = dg.Batch(n, size, method=method, level=level, seed=7, mix=mix)
        desc, total = zpack_amd.decode_descs_from_batch(b)
        dev = torch.device("cuda:0")
        src = torch.from_numpy(b.archive).to(dev)
        dst = torch.zeros(total, dtype=torch.uint8, device=dev)
        ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
        dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        codec.set_profiling(True)
        for it in range(3):
            t = time.time()
            codec.decode_batch_device(src, ddesc, n, dst, dres)
            torch.cuda.synchronize()
            print("    iter %d: %.3f ms wall, k_lz4 %.3f ms k_zstd %.3f ms" % (it, (time.time() - t) * 1e3, codec.kernel_ms(zpack_amd.K_LZ4), codec.kernel_ms(zpack_amd.K_ZSTD)), flush=True)
        if os.environ.get("ZPK_DEBUG_TIMING"):
            raw = codec.debug_read(n)
            m = lambda k: float(np.median(raw[:, k]))
            if method == 1:
                print("    zstd cycles (median/entry): literals %.0f tables %.0f fse-parse %.0f exec %.0f total %.0f | sequences %.0f blocks %.0f"
                      % (m(0), m(1), m(2), m(3), m(6), m(4), m(5)), flush=True)
            elif os.environ.get("ZPK_DEBUG_TIMING") == "2":
                print("    parse split (median cycles/entry): stage %.0f walk1 %.0f fix %.0f emit+scan %.0f token-fetch %.0f | fix iters %.0f chunks %.0f"
                      % (m(0), m(1), m(2), m(3), m(4), np.median(raw[:, 7] >> 32), np.median(raw[:, 7] & 0xffffffff)), flush=True)
            else:
                print("    wave kernel cycles (median/entry): parse %.0f lit %.0f dep %.0f rounds %.0f total %.0f | batches %.0f rounds %.0f coops %.0f lds-assembled %.0f"
                      % (m(0), m(1), m(2), m(3), m(6), np.median(raw[:, 4] >> 32), np.median(raw[:, 4] & 0xffffffff),
                         np.median(raw[:, 5] >> 32), np.median(raw[:, 5] & 0xffffffff)), flush=True)
            t = raw.astype(np.float64)
            print("    phase cycles (median over entries): stage %.0f walk %.0f scan %.0f emit+lit %.0f match %.0f hash/flush %.0f | walk iters %.1f match iters %.0f"
                  % tuple(np.median(t[:, k]) for k in (0, 1, 2, 3, 4, 7, 5, 6)), flush=True)
            print("    max: ", t.max(axis=0)[:7], flush=True)
        res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
        out = dst.cpu().numpy()
        nbad = int((res["status"] != 0).sum())
        import collections
        print("    bad status:", nbad, collections.Counter(res["detail"][res["status"] != 0].tolist()), res[res["status"] != 0][:3], flush=True)
        assert nbad == 0
        assert np.array_equal(res["hash"], b.hashes), "hash mismatch at %s" % np.nonzero(res["hash"] != b.hashes)[0][:5]
        for i in range(0, n, max(1, n // 50)):
            d = desc[i]
            got = out[int(d["dst_offset"]):int(d["dst_offset"] + d["uncomp_size"])]
            assert np.array_equal(got, b.plaintext(i)), "bytes differ entry %d" % i
    else:
        raise SystemExit("unknown case " + name)
    print("PASS", name, flush=True)


CASES_LZ4 = ["ref:archive_lz4.zpk", "small:text/lz4_0", "small:runs/lz4_9", "small:random/lz4_0", "small:records/lz4_0", "status:lz4", "foreign:lz4f",
             "batch:2,0,65536,64,0", "batch:2,0,65536,64,1", "batch:2,0,65536,64,2", "batch:2,0,65536,64,3", "batch:2,0,65536,2000,-1",
             "batch:2,0,300000,64,-1", "batch:2,9,5000,300,-1"]
CASES = ["ref:archive_none.zpk", "ref:archive_lz4.zpk", "ref:archive_zstd.zpk",
         "small:text/none", "small:text/lz4_0", "small:runs/lz4_9", "small:random/lz4_0",
         "small:text/zstd_3", "small:runs/zstd_19", "small:random/zstd_1", "small:records/zstd_1",
         "status:none", "status:lz4", "status:zstd", "foreign:lz4f", "foreign:zstd"]

if __name__ == "__main__":
    if len(sys.argv) > 1 and not sys.argv[1].startswith("--"):
        run_case(sys.argv[1])
        sys.exit(0)
    per = 40
    t_all = time.time()
    os.environ.pop("ZPK_TRACE", None)
    cases = CASES
    if "--lz4" in sys.argv:
        cases = CASES_LZ4
    if "--bisect" in sys.argv:
        cases = []
        os.environ["ZPK_TRACE"] = "3"
        cases = [("ref:archive_none.zpk", 0), ("ref:archive_lz4.zpk", 0), ("ref:archive_zstd.zpk", 0), ("ref:archive_none.zpk", 6), ("ref:archive_none.zpk", 4)]
    for c in cases:
        if isinstance(c, tuple):
            os.environ["ZPK_SKIP"] = str(c[1]); label = "%s skip=%d" % c; c = c[0]
            print("----", label, flush=True)
        if time.time() - t_all > 600:
            print("budget exhausted, stopping", flush=True)
            break
        t = time.time()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=per)
            tail = (p.stdout + p.stderr).strip().splitlines()[-6:]
            print("%-28s rc=%d %.1fs | %s" % (c, p.returncode, time.time() - t, " / ".join(tail)[-600:]), flush=True)
        except subprocess.TimeoutExpired as ex:
            so = ex.stdout.decode(errors="replace") if isinstance(ex.stdout, bytes) else (ex.stdout or "")
            se = ex.stderr.decode(errors="replace") if isinstance(ex.stderr, bytes) else (ex.stderr or "")
            print("%-28s TIMEOUT after %ds | %s" % (c, per, " / ".join((so + se).strip().splitlines()[-8:])[-700:]), flush=True)
