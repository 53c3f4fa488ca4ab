"""GPU parity tests: the HIP codec (through the C-ABI of include/zpack_codec.h) against the oracle and
the golden fixtures written by the compiled reference.  Bit-exact: bytes, XXH3-64 and zpack_result."""
import json
import os

import numpy as np
import pytest

import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    return zpack_amd.Codec(0)


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


def _desc(entries, caps):
    d = np.zeros(len(entries), dtype=zpack_amd.DECODE_DESC)
    for i, (e, cap) in enumerate(zip(entries, caps)):
        d[i]["src_offset"] = e["offset"]
        d[i]["comp_size"] = e["comp_size"]
        d[i]["uncomp_size"] = e["uncomp_size"]
        d[i]["expect_hash"] = e["hash"]
        d[i]["dst_capacity"] = cap
        d[i]["method"] = e["method"]
    return d


def test_hash_all_length_classes(codec):
    o = oracle()
    lens = list(range(0, 300)) + [511, 512, 513, 1023, 1024, 1025, 1087, 1088, 1089, 2047, 2048, 2049, 4095, 4096,
                                   4097, 65535, 65536, 65537, 100003, 262144, 1048576 + 7]
    for n in lens:
        d = dg.fill(dg.RANDOM, 5, n, n)
        assert codec.hash_host(d.tobytes()) == o.xxh3(d), n


def test_hash_batch_device_unaligned(codec):
    import torch
    o = oracle()
    blob = dg.fill(dg.RANDOM, 6, 0, 3 << 20)
    rng = np.random.default_rng(3)
    n = 500
    sizes = rng.integers(0, 5000, n).astype(np.uint64)
    sizes[::7] = rng.integers(240, 70000, len(sizes[::7]))
    offs = rng.integers(0, (3 << 20) - 70000, n).astype(np.uint64)
    dev = torch.device("cuda:0")
    src = torch.from_numpy(blob).to(dev)
    out = torch.zeros(n, dtype=torch.int64, device=dev)
    codec.hash_batch_device(src, torch.from_numpy(offs.view(np.int64)).to(dev), torch.from_numpy(sizes.view(np.int64)).to(dev), n, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for i in range(n):
        assert int(got[i]) == o.xxh3(blob[int(offs[i]):int(offs[i] + sizes[i])]), i


@pytest.mark.parametrize("arc", ["archive_none.zpk", "archive_zstd.zpk", "archive_lz4.zpk"])
def test_reference_bundled_archives(codec, golden_dir, arc):
    """tests/read_archive.c:21-35 of the reference on its own archives, through the codec ABI."""
    wd = os.path.join(golden_dir, "ref_workdir")
    a = open(os.path.join(wd, arc), "rb").read()
    ents = zpk.parse(a)
    res, outs = codec.decode_batch_host(a, _desc(ents, [350, 350]))
    for e, r, out in zip(ents, res, outs):
        plain = open(os.path.join(wd, e["filename"]), "rb").read()
        assert r["status"] == 0 and int(r["hash"]) == e["hash"] and int(r["produced"]) == len(plain)
        assert out[:len(plain)].tobytes() == plain


def test_small_archives_written_by_reference(codec, golden_dir):
    for case in _load(golden_dir, "small_archives.json"):
        a = bytes.fromhex(case["archive"])
        ents = zpk.parse(a)
        res, outs = codec.decode_batch_host(a, _desc(ents, [e["uncomp_size"] for e in ents]))
        for e, size, r, out in zip(ents, case["sizes"], res, outs):
            plain = dg.fill(case["cls"], case["seed"], size, size).tobytes()
            assert r["status"] == 0, (case["label"], case["corpus"], size, r)
            assert out[:size].tobytes() == plain, (case["label"], case["corpus"], size)
            if size:
                assert int(r["hash"]) == e["hash"]


def test_recipes_large_frames(codec, golden_dir):
    recs = _load(golden_dir, "recipes.json")
    frames, ents, plains = [], [], []
    off = 10
    for r in recs:
        plain = dg.fill(r["cls"], r["seed"], r["index"], r["size"])
        frame = dg.compress(r["method"], r["level"], plain)
        assert len(frame) == r["comp_size"] and dg.xxh3(frame) == r["frame_xxh3"]
        frames.append(frame)
        ents.append(dict(offset=off, comp_size=len(frame), uncomp_size=r["size"], hash=r["hash"], method=r["method"]))
        plains.append(plain)
        off += len(frame)
    arc = zpk.assemble(frames, [("f%d" % i, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"])
                               for i, e in enumerate(ents)])
    res, outs = codec.decode_batch_host(arc, _desc(ents, [e["uncomp_size"] for e in ents]))
    for r, rr, out, plain in zip(recs, res, outs, plains):
        assert rr["status"] == 0, (r, rr)
        assert int(rr["hash"]) == r["hash"] and int(rr["produced"]) == r["size"]
        assert np.array_equal(out, plain), r


def test_status_codes_match_reference(codec, golden_dir):
    sc = _load(golden_dir, "status_cases.json")
    for c in sc["cases"]:
        a = bytearray(bytes.fromhex(sc["bases"][c["base"]]))
        for p, x in c["flips"]:
            a[p] ^= x
        e = zpk.parse(a)[c["index"]]
        for k, v in c["tamper"].items():
            e[{"comp_method": "method"}.get(k, k)] = v
        res, outs = codec.decode_batch_host(bytes(a), _desc([e], [c["max_size"]]))
        assert int(res[0]["status"]) == c["rc"], (c["label"], res[0], c["rc"])
        if c["rc"] == 0:
            assert dg.xxh3(outs[0]) == c["out_xxh3"], c["label"]


@pytest.mark.parametrize("split_min", [2 << 20, 1])
def test_foreign_frames_match_reference(codec, golden_dir, split_min):
    """split_min = 1: every entry that IS a sequence of frames with content sizes on 256-byte output boundaries is decoded one wave
    per frame (ZPK_OPT_DEC_SPLIT_MIN) — the reference's verdicts and bytes must not depend on that."""
    codec.set_option(zpack_amd.OPT_DEC_SPLIT_MIN, split_min)
    parallel = 0
    for c in _load(golden_dir, "foreign_frames.json"):
        fr = bytes.fromhex(c["frame"])
        e = dict(offset=10, comp_size=len(fr), uncomp_size=c["uncomp_size"], hash=c["hash"], method=c["method"])
        arc = zpk.assemble([fr], [("f", 10, len(fr), c["uncomp_size"], c["hash"], c["method"])])
        res, outs = codec.decode_batch_host(arc, _desc([e], [c["max_size"]]))
        parallel += codec.decode_stats()["frame_parallel_entries"]
        assert int(res[0]["status"]) == c["rc"], (c["label"], res[0])
        if c["rc"] == 0:
            assert dg.xxh3(outs[0][:c["uncomp_size"]]) == c["plain_xxh3"], c["label"]
    codec.set_option(zpack_amd.OPT_DEC_SPLIT_MIN, 2 << 20)
    assert split_min != 2 << 20 or parallel == 0


@pytest.mark.parametrize("method,level,size,n", [(dg.LZ4, 0, 65536, 512), (dg.ZSTD, 3, 262144, 96), (dg.NONE, 0, 65536, 256),
                                                 (dg.COIN, 3, None, 600)])
def test_device_batch_vs_oracle(codec, method, level, size, n):
    """Device-resident batch (the bench path) on a seeded mixed-class archive: every byte vs the oracle,
    every hash vs the real xxHash values the generator recorded."""
    import torch
    o = oracle()
    if size is None:
        b = dg.Batch(n, 1000, 300000, method=method, level=level, seed=7)
    else:
        b = dg.Batch(n, size, method=method, level=level, seed=7)
    desc, total = zpack_amd.decode_descs_from_batch(b)
    dev = torch.device("cuda:0")
    src = torch.from_numpy(b.archive).to(dev)
    dst = torch.zeros(total, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, ddesc, n, dst, dres)
    torch.cuda.synchronize()
    res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
    out = dst.cpu().numpy()
    assert (res["status"] == 0).all(), res[res["status"] != 0][:5]
    assert np.array_equal(res["hash"], b.hashes)
    assert np.array_equal(res["produced"], b.uncomp_sizes)
    arc = b.archive.tobytes()
    for i in range(0, n, max(1, n // 64)):
        d = desc[i]
        rc, want, got, h = o.entry_decode(arc, int(d["src_offset"]), int(d["comp_size"]), int(d["uncomp_size"]),
                                          int(d["expect_hash"]), int(d["method"]), int(d["dst_capacity"]))
        assert rc == 0
        assert out[int(d["dst_offset"]):int(d["dst_offset"] + d["uncomp_size"])].tobytes() == want, i


@pytest.mark.parametrize("mix,size,n", [(dg.RUNS, 65536, 300), (dg.RUNS, 1 << 20, 12), (dg.TEXT, 65536, 300), (dg.MIX, 40000, 400)])
def test_lz4_entries_that_are_mostly_runs_go_to_their_own_kernel(codec, mix, size, n):
    """round 5: an LZ4 entry that is mostly RUNS (compressed to less than an eighth: LZ4 writes a byte run as literal + match at distance
    1 of ~165 bytes, a repeated block as one long match) goes on the list of k_lz4_left, the build of the one-wave decoder with grouped
    cooperative copies (seq_exec.h COOP = 2); k_lz4_wave is the round-4 code.  With the XXH3 verify OFF every byte of every entry
    must equal the oracle's either way, and the counters say which kernel had what: byte-run entries all the second kernel, text none."""
    import torch
    o = oracle()
    b = dg.Batch(n, size, method=dg.LZ4, level=0, seed=23, mix=mix)
    desc, total = zpack_amd.decode_descs_from_batch(b, flags=zpack_amd.DF_SKIP_HASH)
    dev = torch.device("cuda:0")
    src = torch.from_numpy(b.archive).to(dev)
    dst = torch.full((total,), 0xA5, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, ddesc, n, dst, dres)
    torch.cuda.synchronize()
    res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
    out = dst.cpu().numpy()
    assert (res["status"] == 0).all(), res[res["status"] != 0][:5]
    assert np.array_equal(res["produced"], b.uncomp_sizes) and np.array_equal(res["hash"], b.hashes)
    arc = b.archive.tobytes()
    for i in range(n):
        d = desc[i]
        rc, want, got, h = o.entry_decode(arc, int(d["src_offset"]), int(d["comp_size"]), int(d["uncomp_size"]),
                                          int(d["expect_hash"]), int(d["method"]), int(d["dst_capacity"]))
        assert rc == 0 and out[int(d["dst_offset"]):int(d["dst_offset"] + d["uncomp_size"])].tobytes() == want, i
    handed = codec.decode_stats()["lz4_long_runs"]
    runs_entries = int((b.classes == dg.RUNS).sum())
    if mix == dg.RUNS:
        assert handed == n, (handed, n)
    elif mix == dg.TEXT:
        assert handed == 0, handed
    else:
        assert runs_entries > 0 and handed == runs_entries, (handed, runs_entries)
    assert codec.decode_stats()["lz4"] == n


@pytest.mark.parametrize("size,level,n", [(262144, 3, 64), (1 << 20, 1, 24), (5000, 3, 200)])
def test_zstd_two_stage_path(codec, size, level, n):
    """The Zstandard batch goes FSE pre-decode (k_zstd_fse, four streams per wave) -> literals + execution.  With the
    XXH3 verify switched off nothing can paper over a wrong pre-decoded sequence: every byte of every entry must
    equal the oracle's, and the counters must show that the entries really finished on the two-stage path."""
    import torch
    o = oracle()
    b = dg.Batch(n, size, method=dg.ZSTD, level=level, seed=11)
    desc, total = zpack_amd.decode_descs_from_batch(b, flags=zpack_amd.DF_SKIP_HASH)
    dev = torch.device("cuda:0")
    src = torch.from_numpy(b.archive).to(dev)
    dst = torch.zeros(total, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, ddesc, n, dst, dres)
    torch.cuda.synchronize()
    st = codec.decode_stats()
    res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
    out = dst.cpu().numpy()
    assert (res["status"] == 0).all(), res[res["status"] != 0][:5]
    assert np.array_equal(res["produced"], b.uncomp_sizes)
    arc = b.archive.tobytes()
    for i in range(n):
        d = desc[i]
        rc, want, got, h = o.entry_decode(arc, int(d["src_offset"]), int(d["comp_size"]), int(d["uncomp_size"]),
                                          int(d["expect_hash"]), int(d["method"]), int(d["dst_capacity"]))
        assert rc == 0
        assert out[int(d["dst_offset"]):int(d["dst_offset"] + d["uncomp_size"])].tobytes() == want, i
    assert st["zstd"] == n and st["zstd_two_stage"] + st["zstd_fused"] == n, st
    assert st["zstd_two_stage"] >= (3 * n) // 4, st


def test_pack_batch_scan_and_compaction(codec):
    """K7: after an encode batch the payload offsets are the exclusive prefix sum of the compressed sizes (failed
    entries take no room) and the packed stream is the concatenation of the slot payloads — what the serial
    `write_offset += comp_size` loop of lib/zpack_write.c:287-339 lays out."""
    import torch
    rng = np.random.default_rng(5)
    n = 3000
    sizes = rng.integers(0, 9000, n).astype(np.uint64)
    sizes[::97] = rng.integers(60000, 200000, len(sizes[::97]))
    methods = rng.integers(0, 3, n).astype(np.uint32)
    src_off = np.concatenate([[0], np.cumsum((sizes + 31) & ~np.uint64(15))]).astype(np.uint64)
    blob = dg.fill(dg.TEXT, 9, 0, int(src_off[-1]) + 64)
    desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
    caps = np.array([zpack_amd.lib().zpk_codec_compress_bound(int(m), int(s)) for m, s in zip(methods, sizes)], dtype=np.uint64)
    caps[5::211] = 3                                        # too small on purpose: those entries fail and take no room
    desc["src_offset"] = src_off[:-1]; desc["size"] = sizes; desc["method"] = methods; desc["level"] = 1
    desc["dst_offset"] = np.concatenate([[0], np.cumsum((caps + 63) & ~np.uint64(15))])[:-1]; desc["dst_capacity"] = caps
    slots_bytes = int(desc["dst_offset"][-1] + caps[-1]) + 64
    dev = torch.device("cuda:0")
    src = torch.from_numpy(blob).to(dev)
    slots = torch.zeros(slots_bytes, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.ENCODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.encode_batch_device(src, ddesc, n, slots, dres)
    offs = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    packed = torch.full((slots_bytes,), 0xEE, dtype=torch.uint8, device=dev)
    codec.pack_batch_device(slots, ddesc, dres, n, packed, offs, int(caps.max()))
    torch.cuda.synchronize()
    res = dres.cpu().numpy().view(zpack_amd.ENCODE_RESULT)
    ok = res["status"] == 0
    assert (~ok).sum() >= 5 and ok.sum() > n - 40
    want_sizes = np.where(ok, res["comp_size"], 0).astype(np.uint64)
    want_off = np.concatenate([[0], np.cumsum(want_sizes)]).astype(np.uint64)
    got_off = offs.cpu().numpy().view(np.uint64)
    assert np.array_equal(got_off, want_off)
    hs, hp = slots.cpu().numpy(), packed.cpu().numpy()
    for i in range(n):
        a = int(desc["dst_offset"][i]); k = int(want_sizes[i])
        assert np.array_equal(hp[int(want_off[i]):int(want_off[i]) + k], hs[a:a + k]), i
    assert (hp[int(want_off[-1]):int(want_off[-1]) + 64] == 0xEE).all()       # nothing past the end
    # sizes only
    offs2 = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    codec.pack_batch_device(slots, ddesc, dres, n, None, offs2, int(caps.max()))
    torch.cuda.synchronize()
    assert np.array_equal(offs2.cpu().numpy().view(np.uint64), want_off)


@pytest.mark.parametrize("method,level", [(dg.ZSTD, 3), (dg.ZSTD, 1), (dg.LZ4, 0)])
def test_corrupted_frames_same_verdict_and_bytes_as_oracle(codec, method, level):
    """Byte flips in real frames, a few hundred entries in ONE device batch, XXH3 verify off: every entry must end with
    the oracle's verdict, and where the damaged frame still decodes, with the oracle's bytes (a damaged Zstandard frame
    that stays structurally valid goes through the FSE pre-decode + execute path like any other; nothing may hang)."""
    import torch
    o = oracle()
    rng = np.random.default_rng(99 + level)
    frames, sizes = [], []
    # (the 700-byte base: damaged lengths there usually overflow the output slot AND leave a malformed tail — the verdict
    # must be that of whichever a serial decoder meets first, seq_exec.h)
    for cls, size in ((dg.TEXT, 70000), (dg.RECORDS, 33000), (dg.RUNS, 50000), (dg.TEXT, 3000), (dg.RANDOM, 9000), (dg.TEXT, 700)):
        plain = dg.fill(cls, 21, 0, size)
        base = bytearray(dg.compress(method, level, plain))
        for k in range(50):
            f = bytearray(base)
            if k:                                         # k == 0 keeps the intact frame
                pos = int(rng.integers(0, len(f)))
                f[pos] ^= int(rng.integers(1, 256))
                if k % 7 == 0:                            # and sometimes a second hit, or a truncation
                    f[int(rng.integers(0, len(f)))] ^= 0x80
                if k % 11 == 0:
                    f = f[:int(rng.integers(1, len(f)))]
            frames.append(bytes(f)); sizes.append(size)
    n = len(frames)
    offs, off = [], 10
    for f in frames:
        offs.append(off); off += len(f)
    arc = zpk.assemble(frames, [("f%d" % i, offs[i], len(frames[i]), sizes[i], 0, method) for i in range(n)])
    desc = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
    desc["src_offset"] = offs; desc["comp_size"] = [len(f) for f in frames]; desc["uncomp_size"] = sizes
    desc["dst_capacity"] = sizes; desc["method"] = method; desc["flags"] = zpack_amd.DF_SKIP_HASH
    desc["dst_offset"] = np.concatenate([[0], np.cumsum((np.array(sizes, dtype=np.uint64) + 255) & ~np.uint64(255))])[:-1]
    total = int(desc["dst_offset"][-1]) + sizes[-1] + 256
    dev = torch.device("cuda:0")
    src = torch.from_numpy(np.frombuffer(arc, dtype=np.uint8).copy()).to(dev)
    dst = torch.zeros(total, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, ddesc, n, dst, dres)
    torch.cuda.synchronize()
    st = codec.decode_stats()
    res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
    out = dst.cpu().numpy()
    decoded = 0
    for i in range(n):
        rc, want, got, h = o.entry_decode(arc, offs[i], len(frames[i]), sizes[i], 0, method, sizes[i])
        if rc in (0, 15):                                 # the oracle decoded it (15 = only the hash, which is off here)
            assert int(res[i]["status"]) == 0, (i, res[i], rc)
            a = int(desc["dst_offset"][i])
            assert out[a:a + sizes[i]].tobytes() == want[:sizes[i]], i
            decoded += 1
        else:
            assert int(res[i]["status"]) == rc, (i, res[i], rc)
    assert decoded >= 5                                   # the intact frames at least
    assert st["fse_watchdog"] == 0 and st["fse_budget"] == 0
    if method == dg.ZSTD:
        assert st["zstd_two_stage"] + st["zstd_fused"] == n


def test_huffman_12_bit_code_frame(codec):
    """12-bit Huffman literals (tests/huf12.py): the table layout the GPU keeps for codes longer than 11 bits."""
    from tests.huf12 import make_frame
    o = oracle()
    for seed, n in ((1, 200), (2, 1000), (3, 17)):
        frame, lits = make_frame(seed, n)
        e = dict(offset=10, comp_size=len(frame), uncomp_size=n, hash=o.xxh3(lits), method=1)
        arc = zpk.assemble([frame], [("f", 10, len(frame), n, e["hash"], 1)])
        res, outs = codec.decode_batch_host(arc, _desc([e], [n]))
        assert int(res[0]["status"]) == 0 and int(res[0]["produced"]) == n, res[0]
        assert outs[0][:n].tobytes() == lits


def _run_device_batch(codec, b, flags):
    import torch
    desc, total = zpack_amd.decode_descs_from_batch(b, flags=flags)
    dev = torch.device("cuda:0")
    src = torch.from_numpy(b.archive).to(dev)
    dst = torch.full((total + 64,), 0xA5, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(b.n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, ddesc, b.n, dst, dres)
    torch.cuda.synchronize()
    return desc, dres.cpu().numpy().view(zpack_amd.DECODE_RESULT).copy(), dst.cpu().numpy(), codec.decode_stats()


@pytest.mark.parametrize("mix,size,n", [(-1, 65536, 384), (dg.TEXT, 65536, 128), (dg.RECORDS, 65536, 128), (dg.RUNS, 65536, 64),
                                        (dg.RANDOM, 65536, 64), (-1, (1000, 300000), 300), (dg.TEXT, 1 << 20, 12), (dg.TEXT, (1, 400), 300)])
def test_lz4_decoder_shapes_vs_oracle(codec, mix, size, n):
    """k_lz4_wave on eight batch shapes (every corpus class at 64 KiB, ragged sizes from 1 byte to 1 MiB), XXH3 verify OFF so that
    nothing but the bytes decides: status, size and hash of every entry, every byte of every entry equal to the oracle's, and no
    byte outside an entry's own slot range [dst_offset, dst_offset + produced) may have been touched."""
    o = oracle()
    lo, hi = size if isinstance(size, tuple) else (size, size)
    b = dg.Batch(n, lo, hi, method=dg.LZ4, level=0, seed=23, mix=mix)
    desc, r, out, st = _run_device_batch(codec, b, zpack_amd.DF_SKIP_HASH)
    assert st["lz4"] == n, st
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


_RETRY_SCRIPT = r"""
import json, os, sys
import numpy as np
import torch
import zpack_amd
from benchdata import datagen as dg
codec = zpack_amd.Codec(0)
dev = torch.device("cuda:0")
out = {}
for name, method, level in (("lz4", dg.LZ4, 0), ("zstd", dg.ZSTD, 3)):
    b = dg.Batch(10, 6 << 20, 8 << 20, method=method, level=level, seed=31, mix=dg.TEXT)      # large: the budget is polled every 256 steps
    desc, total = zpack_amd.decode_descs_from_batch(b)
    src = torch.from_numpy(b.archive).to(dev)
    dst = torch.zeros(total, dtype=torch.uint8, device=dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    dres = torch.zeros(b.n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
    codec.decode_batch_device(src, ddesc, b.n, dst, dres)
    torch.cuda.synchronize()
    r = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)
    st = codec.decode_stats()
    out[name] = dict(all_ok=bool((r["status"] == 0).all()), hashes=bool(np.array_equal(r["hash"], b.hashes)), details=[int(x) for x in set(r["detail"].tolist())],
                     retried=int(st["retried_" + name]), n=b.n)
print(json.dumps(out))
"""


def test_watchdog_expiry_is_retried_not_reported():
    """A wave that runs out of its time budget does not give a verdict (the reference has no notion of `too slow`): the entry is
    decoded again with a 64 x larger budget behind the batch.  A -DZPK_DEVELOPER build with ZPK_WD_SCALE=0 spends every budget at the
    first poll: all large entries take the retry path, and every status, hash and detail equals the normal run's."""
    import subprocess
    import sys
    so = os.path.join(os.path.dirname(zpack_amd.CODEC_SO), "dev", "libzpk_codec_dev.so")
    assert os.path.exists(so), "zpack_amd/dev/libzpk_codec_dev.so is built by zpack_amd.build.build_all()"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for scale in ("0", "1"):
        # (ZPK_ZSTD_FUSED: the Zstandard entries go straight to the full decoder k_zstd, whose budget the hook scales; on the two-stage
        # path a stage that runs out of budget hands its entry to k_zstd anyway)
        env = dict(os.environ, ZPACK_AMD_CODEC_SO=so, ZPK_WD_SCALE=scale, ZPK_ZSTD_FUSED="1", PYTHONPATH=root)
        p = subprocess.run([sys.executable, "-c", _RETRY_SCRIPT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=250, cwd=root)
        assert p.returncode == 0, p.stderr[-2000:]
        res[scale] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    for name in ("lz4", "zstd"):
        tiny, normal = res["0"][name], res["1"][name]
        assert normal["all_ok"] and normal["hashes"] and normal["retried"] == 0, normal
        assert tiny["all_ok"] and tiny["hashes"] and tiny["details"] == [0], tiny           # no 0xDEAD anywhere: same verdicts as the normal run
        assert tiny["retried"] >= tiny["n"] // 2, tiny                                       # and the retry launches really did the work


@pytest.mark.parametrize("method,level,n,lo,hi", [(dg.LZ4, 0, 13000, 14000, 30000), (dg.ZSTD, 3, 37500, 5000, 11000), (dg.COIN, 3, 37500, 4000, 12000)])
def test_host_batch_takes_the_upload_decode_download_pipeline(codec, method, level, n, lo, hi):
    """zpk_codec_decode_batch_host on a batch large enough for the three-stage pipeline (pieces uploaded by one thread, decoded as they
    arrive, the output streamed back through the pinned buffers): every status, size and hash, every returned byte (by its XXH3), a
    byte-level comparison with the oracle on a sample, and entries with a wrong expected hash keep their bytes (lib/zpack_read.c:466-468)."""
    o = oracle()
    b = dg.Batch(n, lo, hi, method=method, level=level, seed=41, mix=-1)
    d = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = b.offsets; d["comp_size"] = b.comp_sizes; d["uncomp_size"] = b.uncomp_sizes
    d["expect_hash"] = b.hashes; d["dst_capacity"] = b.uncomp_sizes; d["method"] = b.methods
    bad = np.arange(7, n, 997)
    d["expect_hash"][bad] ^= np.uint64(1)
    res, outs = codec.decode_batch_host(b.archive, d)
    want_status = np.zeros(n, dtype=np.int32); want_status[bad] = 15
    assert np.array_equal(res["status"], want_status), np.nonzero(res["status"] != want_status)[0][:10]
    assert np.array_equal(res["produced"], b.uncomp_sizes) and np.array_equal(res["hash"], b.hashes)
    for i in range(n):
        assert dg.xxh3(outs[i]) == int(b.hashes[i]), i
    arc = b.archive.tobytes()
    for i in list(range(0, n, 1501)) + [n - 1]:
        k = int(b.uncomp_sizes[i])
        rc, want, _, _ = o.entry_decode(arc, int(b.offsets[i]), int(b.comp_sizes[i]), k, int(b.hashes[i]), int(b.methods[i]), k)
        assert rc == 0 and outs[i][:k].tobytes() == want


def _stream_decode(codec, frame, method, uncomp_size, want_hash, chunk, out_chunk=4096):
    """zpk_dstream_* directly (include/zpack_codec.h): feed `chunk` bytes per step -> (final status, bytes, input offset at the first output)"""
    import ctypes as C
    L = codec.L
    L.zpk_dstream_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.zpk_dstream_destroy.argtypes = [C.c_void_p]; L.zpk_dstream_destroy.restype = None
    L.zpk_dstream_step.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t),
                                   C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
    L.zpk_dstream_replay.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_int]
    L.zpk_dstream_wants_input.argtypes = [C.c_void_p]
    L.zpk_dstream_device_bytes.argtypes = [C.c_void_p]; L.zpk_dstream_device_bytes.restype = C.c_uint64
    s = C.c_void_p()
    assert L.zpk_dstream_create(codec.h, C.byref(s)) == 0
    src = np.frombuffer(frame, dtype=np.uint8)
    ob = np.zeros(out_chunk, dtype=np.uint8)
    out = bytearray()
    pos, first, status = 0, None, 0
    consumed, produced, done = C.c_size_t(0), C.c_size_t(0), C.c_int(0)
    stats = _stream_decode.stats = dict(restarts=0, device_peak=0, steps=0)
    for _ in range(1000000):
        take = min(chunk, len(src) - pos) if L.zpk_dstream_wants_input(s) else 0
        rc = L.zpk_dstream_step(s, method, len(src), uncomp_size, want_hash, src[pos:].ctypes.data if take else None, take, C.byref(consumed),
                                ob.ctypes.data, out_chunk, C.byref(produced), C.byref(done))
        if rc == 1001:                           # ZPK_DS_RESTART (include/zpack_codec.h): the bytes given so far once more, then on
            stats["restarts"] += 1
            rc = L.zpk_dstream_replay(s, method, len(src), uncomp_size, want_hash, src.ctypes.data if pos else None, pos, 1)
            if rc == 0:
                continue
        stats["device_peak"] = max(stats["device_peak"], int(L.zpk_dstream_device_bytes(s)))
        pos += consumed.value
        if produced.value:
            if first is None:
                first = pos
            out += ob[:produced.value].tobytes()
        status = rc
        if rc not in (0,) or done.value:
            break
        assert consumed.value or produced.value, "a step that neither consumes nor produces"
    la, ca = C.c_uint64(0), C.c_uint64(0)
    L.zpk_dstream_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.zpk_dstream_counters(s, C.byref(la), C.byref(ca))
    stats["steps"] = la.value
    L.zpk_dstream_destroy(s)
    return status, bytes(out), first


@pytest.mark.parametrize("level", [1, 3, 9, 15])
def test_stream_large_zstd_entry_goes_in_bounded_block_parallel_steps(codec, level):
    """The same for one plain Zstandard frame (lib/zpack_write.c:179): steps of up to 64 blocks behind the frame's window; the repeat
    offsets, the running XXH3, the last block with a Huffman tree and the table descriptions in force (levels >= 9: Repeat_Mode blocks)
    travel from step to step."""
    size = (40 << 20) if level <= 3 else (20 << 20)
    tile = np.concatenate([dg.fill(k % 2, 43, k, 1 << 20) for k in range(8)])
    plain = np.ascontiguousarray(np.resize(tile, size))
    frame = dg.compress(dg.ZSTD, level, plain)
    h = dg.xxh3(plain)
    st, out, first = _stream_decode(codec, frame, dg.ZSTD, size, h, 131072, 1 << 20)
    stats = _stream_decode.stats
    assert st == 0 and out == plain.tobytes(), (st, stats)
    assert stats["restarts"] == 0 and first is not None and first <= 4 * 131072, (stats, first)
    assert stats["device_peak"] < (64 << 20) and 4 <= stats["steps"] <= 64, stats
    st, out, _ = _stream_decode(codec, frame, dg.ZSTD, size, h ^ 4, 131072, 1 << 20)
    assert st == 15 and out == plain.tobytes() and _stream_decode.stats["restarts"] == 0
    for cls in (dg.TEXT, dg.RUNS, dg.RANDOM):                   # Treeless chains, RLE / raw blocks
        p1 = dg.fill(cls, 44, 0, 6 << 20)
        f1 = dg.compress(dg.ZSTD, level, p1)
        st, out, _ = _stream_decode(codec, f1, dg.ZSTD, len(p1), dg.xxh3(p1), 100000, 300000)
        assert st == 0 and out == p1.tobytes(), (cls, st, _stream_decode.stats)
    # ---- what the steps do not decide ----
    small = 6 << 20
    p2 = plain[:small]
    f2 = bytearray(dg.compress(dg.ZSTD, level, p2))
    h2 = dg.xxh3(p2)
    rng = np.random.default_rng(10)
    variants = [("bytes behind the last block", bytes(f2) + b"\0\0\0", small, h2),
                ("two frames", bytes(f2) + bytes(f2), 2 * small, dg.xxh3(np.concatenate([p2, p2]))),
                ("truncated", bytes(f2[:-9]), small, h2),
                ("claims less", bytes(f2), small - 1000, h2)]
    for k in range(8):
        b = bytearray(f2); at = int(rng.integers(len(b) // 3, len(b))); b[at] ^= 1 << int(rng.integers(0, 8))
        variants.append(("flip@%d" % at, bytes(b), small, h2))
    for label, payload, usize, hh in variants:
        arc = np.concatenate([np.zeros(10, np.uint8), np.frombuffer(payload, dtype=np.uint8), np.zeros(64, np.uint8)])
        d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
        d["src_offset"] = 10; d["comp_size"] = len(payload); d["uncomp_size"] = usize; d["expect_hash"] = hh; d["dst_capacity"] = usize; d["method"] = dg.ZSTD
        r, o1 = codec.decode_batch_host(arc, d)
        st, out, _ = _stream_decode(codec, payload, dg.ZSTD, usize, hh, 131072, 1 << 20)
        one = int(r["status"][0])
        assert st == one, (label, st, one, _stream_decode.stats)
        if one in (0, 15):
            assert out == o1[0][:usize].tobytes(), label


def test_stream_large_lz4_entry_goes_in_bounded_block_parallel_steps(codec):
    """A large entry that is one plain LZ4 frame (what the reference writer produces, lib/zpack_write.c:204-210) streams in steps of up
    to 64 blocks decoded side by side; the stream holds a window of the entry, not the entry: ~40 MiB of device memory for 48 MiB of
    output as for any other size, first output after a few blocks, bytes and verdict right.  Anything the steps cannot decide —
    a flipped byte, bytes behind the EndMark, a second frame, a truncated entry, a wrong size — restarts the stream in its windowless
    form (ZPK_DS_RESTART + zpk_dstream_replay): the verdict and the bytes are the one-shot decode's."""
    size = 48 << 20
    tile = np.concatenate([dg.fill(k % 2, 41, k, 1 << 20) for k in range(8)])
    plain = np.ascontiguousarray(np.resize(tile, size))
    frame = dg.compress(dg.LZ4, 0, plain)
    h = dg.xxh3(plain)
    st, out, first = _stream_decode(codec, frame, dg.LZ4, size, h, 131072, 1 << 20)
    stats = _stream_decode.stats
    assert st == 0 and out == plain.tobytes()
    assert stats["restarts"] == 0 and first is not None and first <= 6 * 131072, (stats, first)
    assert stats["device_peak"] < (64 << 20) and 8 <= stats["steps"] <= 64, stats
    st, out, _ = _stream_decode(codec, frame, dg.LZ4, size, h ^ 4, 131072, 1 << 20)          # a wrong hash: the verdict with the last byte, the bytes stay
    assert st == 15 and out == plain.tobytes() and _stream_decode.stats["restarts"] == 0
    # ---- what the steps do not decide ----
    small = 5 << 20
    p2 = plain[:small]
    f2 = bytearray(dg.compress(dg.LZ4, 0, p2))
    h2 = dg.xxh3(p2)
    rng = np.random.default_rng(9)
    variants = [("bytes behind the EndMark", bytes(f2) + b"\0\0\0", small, h2),
                ("two frames", bytes(f2) + bytes(f2), 2 * small, dg.xxh3(np.concatenate([p2, p2]))),
                ("truncated", bytes(f2[:-9]), small, h2),
                ("claims less", bytes(f2), small - 1000, h2),
                ("claims more", bytes(f2), small + 1000, h2)]
    for k in range(6):
        b = bytearray(f2); at = int(rng.integers(len(b) // 3, len(b))); b[at] ^= 0x55
        variants.append(("flip@%d" % at, bytes(b), small, h2))
    for label, payload, usize, hh in variants:
        arc = np.concatenate([np.zeros(10, np.uint8), np.frombuffer(payload, dtype=np.uint8), np.zeros(64, np.uint8)])
        d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
        d["src_offset"] = 10; d["comp_size"] = len(payload); d["uncomp_size"] = usize; d["expect_hash"] = hh; d["dst_capacity"] = usize; d["method"] = dg.LZ4
        r, o1 = codec.decode_batch_host(arc, d)
        st, out, _ = _stream_decode(codec, payload, dg.LZ4, usize, hh, 131072, 1 << 20)
        one = int(r["status"][0])
        if label == "claims more":
            # the streaming reader hashes the bytes it PRODUCED (lib/zpack_read.c:556-609): the frame ends short of the claim, its bytes
            # have the expected hash -> OK with the frame's last byte (the one-shot reader hashes its whole buffer: FILE_INCOMPLETE / mismatch)
            assert st == 0 and out == p2.tobytes() and _stream_decode.stats["restarts"] == 0, (label, st, one, _stream_decode.stats)
            continue
        assert st == one, (label, st, one, _stream_decode.stats)
        if one == 0:
            assert out == o1[0][:usize].tobytes(), label
        assert _stream_decode.stats["restarts"] == 1 or label.startswith("flip"), (label, _stream_decode.stats)


@pytest.mark.parametrize("chunk", [7, 1000, 70000])
def test_stream_steps_agree_with_one_shot_on_foreign_and_damaged_frames(codec, golden_dir, chunk):
    """The resumable decode steps (block boundaries, frame headers, checksums, skippable and concatenated frames — whatever the chunk
    size cuts through) against the one-shot decode of the same bytes: every foreign-format fixture the compiled reference produced,
    and byte-flipped / truncated frames of both codecs — the same final verdict, and the same bytes wherever the frame decodes."""
    cases = []
    for c in _load(golden_dir, "foreign_frames.json"):
        if c["max_size"] == c["uncomp_size"] and c["uncomp_size"] > 0:
            cases.append((c["label"], bytes.fromhex(c["frame"]), c["method"], c["uncomp_size"], c["hash"]))
    rng = np.random.default_rng(5)
    for method, level in ((dg.ZSTD, 3), (dg.LZ4, 0)):
        for cls, size in ((dg.TEXT, 200000), (dg.RECORDS, 70000), (dg.RUNS, 90000), (dg.TEXT, 900)):
            plain = dg.fill(cls, 33, 0, size)
            base = bytearray(dg.compress(method, level, plain))
            for k in range(10 if chunk == 7 else 24):
                f = bytearray(base)
                if k:
                    f[int(rng.integers(0, len(f)))] ^= int(rng.integers(1, 256))
                    if k % 5 == 0:
                        f = f[:int(rng.integers(1, len(f)))]
                cases.append(("m%d c%d %d #%d" % (method, cls, size, k), bytes(f), method, size, dg.xxh3(plain)))
    for label, fr, method, usize, h in cases:
        e = dict(offset=10, comp_size=len(fr), uncomp_size=usize, hash=h, method=method)
        arc = zpk.assemble([fr], [("f", 10, len(fr), usize, h, method)])
        res, outs = codec.decode_batch_host(arc, _desc([e], [usize]))
        want_rc = int(res[0]["status"])
        status, got, first = _stream_decode(codec, fr, method, usize, h, chunk)
        produced = int(res[0]["produced"])
        if want_rc == 0 and produced < usize:
            want_rc = 15          # a frame that ends early: the one-shot reader hashes its whole buffer (lib/zpack_read.c:466), the streaming
                                  # reader the bytes it produced (:556-609) — here the expected hash is that of `usize` bytes, so: mismatch
        assert status == want_rc, (label, "one-shot", want_rc, "stream", status)
        if want_rc in (0, 15):
            assert got == outs[0][:min(produced, usize)].tobytes(), label


@pytest.mark.parametrize("lo,hi", [(200, 300000), (3000, 3000)])
def test_work_lists_largest_first_give_identical_results(codec, lo, hi):
    """A ragged batch (200 B ... 300 KiB; and a batch of ONE size, which is ordered by "did it compress" instead: both methods, all four
    classes, + damaged and refused entries) decoded with its work lists in archive order and
    largest entries first (ZPK_OPT_ORDER_MIN: the device counting sort behind the classification): identical statuses, hashes, produced
    counts and bytes; and every entry is on exactly one list position (nothing lost, nothing twice — the output of a skipped entry
    would stay zero)."""
    import torch
    n = 6000
    b = dg.Batch(n, lo, hi, method=dg.COIN, level=3, seed=21)
    desc, total = zpack_amd.decode_descs_from_batch(b)
    desc = desc.copy()
    arc = b.archive.copy()
    rng = np.random.default_rng(4)
    for i in rng.choice(n, 40, replace=False):                                          # damaged payloads
        arc[int(desc["src_offset"][i]) + int(rng.integers(0, int(desc["comp_size"][i])))] ^= 0x5A
    for i in rng.choice(n, 20, replace=False):                                          # refused by the guards
        desc["dst_capacity"][i] = desc["uncomp_size"][i] - 1
    dev = torch.device("cuda:0")
    src = torch.from_numpy(arc).to(dev)
    ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev)
    got = []
    for order_min in (1, 0):
        codec.set_option(zpack_amd.OPT_ORDER_MIN, order_min)
        dst = torch.zeros(total, dtype=torch.uint8, device=dev)
        dres = torch.zeros(n * zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()                     # (the fills above run on torch's stream, the batch on the codec's own)
        codec.decode_batch_device(src, ddesc, n, dst, dres)
        torch.cuda.synchronize()
        got.append((dres.cpu().numpy().view(zpack_amd.DECODE_RESULT).copy(), dst.cpu().numpy().copy()))
    codec.set_option(zpack_amd.OPT_ORDER_MIN, 8192)
    (r1, o1), (r0, o0) = got
    assert np.array_equal(r1["status"], r0["status"]) and np.array_equal(r1["produced"], r0["produced"])
    ok = r0["status"] == 0
    assert ok.sum() >= n - 60 and np.array_equal(r1["hash"][ok], r0["hash"][ok]) and np.array_equal(r1["hash"][ok], b.hashes[ok])
    for i in np.flatnonzero(ok):
        lo = int(desc["dst_offset"][i]); hi = lo + int(desc["uncomp_size"][i])
        assert np.array_equal(o1[lo:hi], o0[lo:hi]), i
