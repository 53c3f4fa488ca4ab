"""GPU parity tests of the two-stage LZ4 path (zpack_amd/csrc/lz4_two.h: k_lz4_parse, one LANE per entry -> k_lz4_exec, one wave per
entry from the records -> k_lz4_left, the general decoder for whatever the two did not finish).  Large batches take it by default;
here ZPK_OPT_LZ4_TWO_STAGE_MIN = 0 sends every batch through it, so the fixtures and shapes the one-kernel decoder is tested on
(reference-written frames, foreign frames, the reference's verdicts on damaged input) pin this path as well.  XXH3 verify OFF where
bytes are compared: nothing but the bytes decides."""
import json
import os

import numpy as np
import pytest

import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[0, 1], ids=["slot", "window"])
def codec2(request):
    """every batch through the two-stage path; stage 2 over the output slot (k_lz4_exec_g) or with the LDS window (k_lz4_exec)"""
    c = zpack_amd.Codec(0)
    c.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MIN, 0)
    c.set_option(zpack_amd.OPT_LZ4_EXEC_WINDOW, request.param)
    if request.param:
        c.set_option(zpack_amd.OPT_ORDER_MIN, 1)          # ... and, for one of the two, every batch's work lists largest entries first
    return c


def _device_batch(codec, arc, desc, total, n, fill=0xA5):
    import torch
    dev = torch.device("cuda:0")
    src = torch.from_numpy(np.frombuffer(arc, dtype=np.uint8).copy() if not isinstance(arc, np.ndarray) else arc).to(dev)
    dst = torch.full((total + 64,), fill, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, ddesc, n, dst, dres)
    torch.cuda.synchronize()
    return dres.cpu().numpy().view(zpack_amd.DECODE_RESULT).copy(), dst.cpu().numpy(), codec.decode_stats()


@pytest.mark.parametrize("mix,size,n,max_comp", [(-1, 65536, 384, 0), (dg.TEXT, 65536, 128, 0), (dg.RECORDS, 65536, 128, 0), (dg.RUNS, 65536, 64, 0),
                                                 (dg.RANDOM, 65536, 64, 0), (-1, (1000, 300000), 300, 0), (dg.TEXT, 1 << 20, 12, 4 << 20),
                                                 (dg.TEXT, (1, 400), 300, 0), (-1, (1000, 300000), 300, 4 << 20)])
def test_two_stage_shapes_vs_oracle(codec2, mix, size, n, max_comp):
    """The eight batch shapes of test_lz4_decoder_shapes_vs_oracle (+ the ragged one with the per-lane size limit lifted, so that
    multi-block entries of up to 300 KB are walked by a lane): status, size, hash and EVERY byte of every entry equal to the
    oracle's, no byte outside an entry's own range touched, and the counters show which way the entries went."""
    o = oracle()
    lo, hi = size if isinstance(size, tuple) else (size, size)
    codec2.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MAX_COMP, max_comp if max_comp else 96 << 10)
    b = dg.Batch(n, lo, hi, method=dg.LZ4, level=0, seed=23, mix=mix)
    desc, total = zpack_amd.decode_descs_from_batch(b, flags=zpack_amd.DF_SKIP_HASH)
    r, out, st = _device_batch(codec2, b.archive, desc, total, n)
    assert st["lz4"] == n and st["lz4_two_stage_taken"], st
    assert st["lz4_two_stage"] + st["lz4_two_stage_left"] == n, st
    within = int((b.comp_sizes <= (max_comp if max_comp else 96 << 10)).sum())
    assert st["lz4_two_stage"] == within, (st, within)             # every regular entry within the size limit finishes on the two-stage path
    assert (r["status"] == 0).all(), r[r["status"] != 0][:3]
    assert np.array_equal(r["produced"], b.uncomp_sizes)
    assert np.array_equal(r["hash"], b.hashes), np.nonzero(r["hash"] != b.hashes)[0][:10]
    arc = b.archive.tobytes()
    for i in range(n):
        d = desc[i]
        a, k = int(d["dst_offset"]), int(d["uncomp_size"])
        rc, want, _, _ = o.entry_decode(arc, int(d["src_offset"]), int(d["comp_size"]), k, int(d["expect_hash"]), 2, k)
        w = np.frombuffer(want, dtype=np.uint8)[:k]
        bad = np.nonzero(out[a:a + k] != w)[0]
        assert rc == 0 and bad.size == 0, (i, "class", int(b.classes[i]), "first bad byte", int(bad[0]) if bad.size else -1, "of", k)
        nxt = int(desc[i + 1]["dst_offset"]) if i + 1 < n else len(out) - 64
        assert (out[a + k:nxt] == 0xA5).all(), ("bytes past the entry were written", i)
    codec2.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MAX_COMP, 96 << 10)


@pytest.mark.parametrize("level", [0, 9])
def test_two_stage_corrupted_frames_same_verdict_and_bytes_as_oracle(codec2, level):
    """Damaged LZ4 frames in ONE device batch through the two-stage path: the oracle's verdict for every entry, the oracle's bytes
    where the frame still decodes.  (The parser lane marks what it does not like, the executor gives up at the first irregularity,
    the general decoder gives every verdict other than OK: nothing may be finished wrongly on the way.)"""
    o = oracle()
    rng = np.random.default_rng(1234 + level)
    frames, sizes = [], []
    for cls, size in ((dg.TEXT, 70000), (dg.RECORDS, 33000), (dg.RUNS, 50000), (dg.TEXT, 3000), (dg.RANDOM, 9000), (dg.TEXT, 700), (dg.TEXT, 200000)):
        plain = dg.fill(cls, 21, 0, size)
        base = bytearray(dg.compress(dg.LZ4, level, plain))
        for k in range(60):
            f = bytearray(base)
            if k:
                pos = int(rng.integers(0, len(f)))
                f[pos] ^= int(rng.integers(1, 256))
                if k % 7 == 0:
                    f[int(rng.integers(0, len(f)))] ^= 0x80
                if k % 11 == 0:
                    f = f[:int(rng.integers(1, len(f)))]
                if k % 13 == 0:                           # a length nibble forced to 15: extension bytes out of whatever follows
                    f[int(rng.integers(11, len(f)))] |= 0xF0
                if k % 17 == 0:                           # a second frame behind the (damaged) first: the reference's loop decodes on
                    f = f + bytearray(dg.compress(dg.LZ4, level, plain[:1000]))       # (tools/fuzz_gpu.py found the window path stopping behind frame 1)
            frames.append(bytes(f)); sizes.append(size)
    n = len(frames)
    offs, off = [], 10
    for f in frames:
        offs.append(off); off += len(f)
    arc = zpk.assemble(frames, [("f%d" % i, offs[i], len(frames[i]), sizes[i], 0, 2) for i in range(n)])
    desc = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
    desc["src_offset"] = offs; desc["comp_size"] = [len(f) for f in frames]; desc["uncomp_size"] = sizes
    desc["dst_capacity"] = sizes; desc["method"] = 2; desc["flags"] = zpack_amd.DF_SKIP_HASH
    desc["dst_offset"] = np.concatenate([[0], np.cumsum((np.array(sizes, dtype=np.uint64) + 255) & ~np.uint64(255))])[:-1]
    total = int(desc["dst_offset"][-1]) + sizes[-1] + 256
    codec2.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MAX_COMP, 4 << 20)
    r, out, st = _device_batch(codec2, arc, desc, total, n, fill=0)      # (the checker's buffer is zeroed: a damaged frame may decode to fewer bytes)
    codec2.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MAX_COMP, 96 << 10)
    assert st["lz4_two_stage_taken"] and st["lz4_two_stage"] + st["lz4_two_stage_left"] == n, st
    decoded = 0
    for i in range(n):
        rc, want, got, h = o.entry_decode(arc, offs[i], len(frames[i]), sizes[i], 0, 2, sizes[i])
        if rc in (0, 15):
            assert int(r[i]["status"]) == 0, (i, r[i], rc)
            a = int(desc["dst_offset"][i])
            assert out[a:a + sizes[i]].tobytes() == want[:sizes[i]], i
            decoded += 1
        else:
            assert int(r[i]["status"]) == rc, (i, r[i], rc)
    assert decoded >= 7 and st["lz4_two_stage"] >= 7, (decoded, st)


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


def _desc(entries, caps):
    d = np.zeros(len(entries), dtype=zpack_amd.DECODE_DESC)
    for i, (e, cap) in enumerate(zip(entries, caps)):
        d[i]["src_offset"] = e["offset"]; d[i]["comp_size"] = e["comp_size"]; d[i]["uncomp_size"] = e["uncomp_size"]
        d[i]["expect_hash"] = e["hash"]; d[i]["dst_capacity"] = cap; d[i]["method"] = e["method"]
    return d


def test_two_stage_reference_fixtures(codec2, golden_dir):
    """The reference's own LZ4 archive, the frames its writer produced (small_archives.json), its verdicts on guard / malformed cases
    and the foreign-format frames — all through the two-stage path (batches of one: a lane, a wave, the general decoder behind)."""
    taken = 0          # entries the two-stage path itself finished (the rest went through it to the general decoder)
    wd = os.path.join(golden_dir, "ref_workdir")
    a = open(os.path.join(wd, "archive_lz4.zpk"), "rb").read()
    ents = zpk.parse(a)
    res, outs = codec2.decode_batch_host(a, _desc(ents, [350, 350]))
    taken += codec2.decode_stats()["lz4_two_stage"]
    assert taken == 2                                           # the reference's own two entries: regular frames
    for e, r, out in zip(ents, res, outs):
        plain = open(os.path.join(wd, e["filename"]), "rb").read()
        assert r["status"] == 0 and int(r["hash"]) == e["hash"] and int(r["produced"]) == len(plain)
        assert out[:len(plain)].tobytes() == plain
    for case in _load(golden_dir, "small_archives.json"):
        a = bytes.fromhex(case["archive"])
        ents = zpk.parse(a)
        if not any(e["method"] == 2 for e in ents):
            continue
        res, outs = codec2.decode_batch_host(a, _desc(ents, [e["uncomp_size"] for e in ents]))
        st = codec2.decode_stats()
        assert st["lz4_two_stage"] == sum(1 for e in ents if e["comp_size"] >= 11), (case["label"], st)      # every non-empty entry of the reference writer
        for e, size, r, out in zip(ents, case["sizes"], res, outs):
            plain = dg.fill(case["cls"], case["seed"], size, size).tobytes()
            assert r["status"] == 0, (case["label"], case["corpus"], size, r)
            assert out[:size].tobytes() == plain, (case["label"], case["corpus"], size)
    sc = _load(golden_dir, "status_cases.json")
    for c in sc["cases"]:
        a = bytearray(bytes.fromhex(sc["bases"][c["base"]]))
        for p, x in c["flips"]:
            a[p] ^= x
        e = zpk.parse(a)[c["index"]]
        for k, v in c["tamper"].items():
            e[{"comp_method": "method"}.get(k, k)] = v
        res, outs = codec2.decode_batch_host(bytes(a), _desc([e], [c["max_size"]]))
        assert int(res[0]["status"]) == c["rc"], (c["label"], res[0], c["rc"])
        if c["rc"] == 0:
            assert dg.xxh3(outs[0]) == c["out_xxh3"], c["label"]
    for c in _load(golden_dir, "foreign_frames.json"):
        if c["method"] != 2:
            continue
        fr = bytes.fromhex(c["frame"])
        e = dict(offset=10, comp_size=len(fr), uncomp_size=c["uncomp_size"], hash=c["hash"], method=c["method"])
        arc = zpk.assemble([fr], [("f", 10, len(fr), c["uncomp_size"], c["hash"], c["method"])])
        res, outs = codec2.decode_batch_host(arc, _desc([e], [c["max_size"]]))
        assert int(res[0]["status"]) == c["rc"], (c["label"], res[0])
        if c["rc"] == 0:
            assert dg.xxh3(outs[0][:c["uncomp_size"]]) == c["plain_xxh3"], c["label"]
        # only a plain single frame is the two-stage path's to finish (the slot executor also takes checksummed / sized frames);
        # several frames, skippable frames, larger blocks go on to the general decoder
        if any(t in c["label"] for t in ("concatenated", "three_frames", "skippable_between", "skippable_then", "skippable_only", "256k", "1m_blocks")):
            assert codec2.decode_stats()["lz4_two_stage"] == 0, c["label"]


def test_two_stage_entry_at_the_end_of_the_image(codec2):
    """The parser lane reads 16 bytes at a time; an entry that ends one byte in front of the end of the source image (the closest the
    offset guard of lib/zpack_read.c:331 allows) has its last windows pulled back inside the image."""
    o = oracle()
    for size in (1, 5, 17, 100, 5000, 65536):
        b = dg.Batch(40, size, size, method=dg.LZ4, level=0, seed=5, mix=-1)
        desc, total = zpack_amd.decode_descs_from_batch(b, flags=zpack_amd.DF_SKIP_HASH)
        end = int(desc["src_offset"][-1] + desc["comp_size"][-1])
        arc = np.ascontiguousarray(b.archive[:end + 1])
        r, out, st = _device_batch(codec2, arc, desc, total, b.n)
        assert (r["status"] == 0).all() and np.array_equal(r["hash"], b.hashes), (size, r[r["status"] != 0][:3])
        # (an entry whose last 64-byte group of compressed bytes crosses the end of the image goes the general way: the parser lane
        # fetches whole groups)
        so, cs = desc["src_offset"].astype(np.int64), desc["comp_size"].astype(np.int64)
        whole = ((so & ~63) + (((so & 63) + cs + 63) & ~63) <= len(arc)) & (cs >= 11)
        assert st["lz4_two_stage"] == int(whole.sum()) and not whole[-1], (size, st, int(whole.sum()))
        i = b.n - 1
        d = desc[i]
        rc, want, _, _ = o.entry_decode(arc.tobytes(), int(d["src_offset"]), int(d["comp_size"]), size, int(d["expect_hash"]), 2, size)
        assert rc == 0 and out[int(d["dst_offset"]):int(d["dst_offset"]) + size].tobytes() == want[:size]


@pytest.mark.parametrize("window", [0, 1])
def test_two_stage_same_results_as_one_kernel_path_large_batch(window):
    """A large batch through the two-stage path (both stage-2 executors) and through the one-kernel decoder (the default): same
    statuses, sizes, hashes and output bytes, and the counters say which path ran."""
    import torch
    n = 40000
    b = dg.Batch(n, 4096, 4096, method=dg.LZ4, level=0, seed=77, mix=-1)
    desc, total = zpack_amd.decode_descs_from_batch(b)
    outs = []
    for never in (False, True):
        c = zpack_amd.Codec(0)
        if not never:
            c.set_option(zpack_amd.OPT_LZ4_TWO_STAGE_MIN, 32768)
            c.set_option(zpack_amd.OPT_LZ4_EXEC_WINDOW, window)
        r, out, st = _device_batch(c, b.archive, desc, total, n)
        assert st["lz4_two_stage_taken"] == (not never), st
        assert (r["status"] == 0).all() and np.array_equal(r["hash"], b.hashes) and np.array_equal(r["produced"], b.uncomp_sizes)
        outs.append(out)
        c.close()
    assert np.array_equal(outs[0], outs[1])
