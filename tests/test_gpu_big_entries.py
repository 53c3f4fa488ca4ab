"""Large entries (SURVEY.md §8 a4 / a5 / a8 in depth): one wave works on one FRAME, so an entry of hundreds of MiB in one frame is one
wave's work.  The host write path cuts entries of >= 2 MiB into 512 KiB pieces — one frame each, side by side, XXH3 by the whole chip —
and the host read path decodes an entry that is such a sequence with one wave per frame.  Checked here: the bytes, hashes and
verdicts are those of the one-wave paths, of the oracle and of the compiled reference (whose readers continue with the next frame:
lib/zpack_read.c:380, :414-439)."""
import ctypes as C
import os

import numpy as np
import pytest

import zpack_amd
from benchdata import datagen as dg
from tests._libs import oracle, have_ref, ref
from zpack_amd import METHOD_NONE, METHOD_ZSTD, METHOD_LZ4, OPT_ENC_SPLIT_MIN, OPT_DEC_SPLIT_MIN

pytestmark = pytest.mark.gpu
M = 1 << 20
PIECE = 512 << 10


@pytest.fixture(scope="module")
def codec():
    c = zpack_amd.Codec(0)
    yield c
    c.close()


def _encode(codec, plains, methods):
    n = len(plains)
    bounds = [codec.compress_bound(m, len(p)) for p, (m, _) in zip(plains, methods)]
    outs = [np.zeros(max(b, 1), dtype=np.uint8) for b in bounds]
    desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
    desc["size"] = [len(p) for p in plains]; desc["dst_capacity"] = bounds
    desc["method"] = [m for m, _ in methods]; desc["level"] = [lv for _, lv in methods]
    res = np.zeros(n, dtype=zpack_amd.ENCODE_RESULT)
    sp = (C.c_void_p * n)(*[p.ctypes.data for p in plains])
    dp = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    L = codec.L
    L.zpk_codec_encode_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    rc = L.zpk_codec_encode_batch_host(codec.h, sp, desc.ctypes.data, n, dp, res.ctypes.data)
    assert rc == 0 and (res["status"] == 0).all(), (rc, res)
    return res, [o[:int(c)] for o, c in zip(outs, res["comp_size"])]


def _image(payloads):
    cs = np.array([len(p) for p in payloads], dtype=np.int64)
    offs = np.concatenate([[10], 10 + np.cumsum(cs)])
    arc = np.zeros(int(offs[-1]) + 64, dtype=np.uint8)
    for i, p in enumerate(payloads):
        arc[offs[i]:offs[i + 1]] = p
    return arc, offs[:-1], cs


def _descs(offs, cs, sizes, hashes, methods, flags=0):
    d = np.zeros(len(cs), dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = offs; d["comp_size"] = cs; d["uncomp_size"] = sizes; d["expect_hash"] = hashes
    d["dst_capacity"] = sizes; d["method"] = [m for m, _ in methods]; d["flags"] = flags
    return d


CASES = [(METHOD_LZ4, 0, 2 * M), (METHOD_LZ4, 0, 3 * M + 17), (METHOD_ZSTD, 3, 2 * M + 1), (METHOD_ZSTD, 1, 7 * M - 5), (METHOD_NONE, 0, 4 * M + 3),
         (METHOD_LZ4, 0, 2 * M - 1), (METHOD_ZSTD, 1, 300000), (METHOD_NONE, 0, 70000), (METHOD_LZ4, 9, 5 * M), (METHOD_ZSTD, 1, 33 * M + 777)]


def test_big_entries_written_in_pieces_read_block_parallel_equal_one_wave(codec):
    """Written in pieces (one frame per entry, its blocks compressed 512 KiB piece by piece), read block-parallel; the same archive read by
    one wave per entry; an unsplit writing of the same plaintexts: every variant gives the same bytes, hashes (the real xxHash's) and
    produced counts, and the counters say which path ran.  Then a SEQUENCE of frames as one entry (what round 4 wrote): one wave per frame."""
    plains = [dg.fill(i % 4, 321, i, n) for i, (_, _, n) in enumerate(CASES)]
    methods = [(m, lv) for m, lv, _ in CASES]
    sizes = [n for _, _, n in CASES]
    want = [dg.xxh3(p) for p in plains]
    codec.set_option(OPT_ENC_SPLIT_MIN, 2 * M)
    res, pay = _encode(codec, plains, methods)
    assert [int(h) for h in res["hash"]] == want
    arc, offs, cs = _image(pay)
    d = _descs(offs, cs, sizes, res["hash"], methods)
    nbig = sum(1 for n in sizes if n >= 2 * M)
    # stored: slices of 512 KiB; LZ4: its 64 KiB blocks; Zstandard: its 64 KiB blocks + the empty block that closes a frame written in pieces
    nunits = sum(((n + PIECE - 1) // PIECE) if m == METHOD_NONE else ((n + 65535) // 65536 + (1 if m == METHOD_ZSTD else 0)) for (m, _, n) in CASES if n >= 2 * M)
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    r1, out1 = codec.decode_batch_host(arc, d)
    st = codec.decode_stats()
    # (the Zstandard entry of 2 MiB + 1 holds random bytes: stored blocks, which one wave copies faster than the block-parallel reader's fixed
    # 4 ms — the batch's cost estimate may leave it in the usual batch: 33 blocks + the closing one)
    assert (st["frame_parallel_entries"], st["frame_parallel_frames"]) in ((nbig, nunits), (nbig - 1, nunits - 34)), (st, nbig, nunits)
    codec.set_option(OPT_DEC_SPLIT_MIN, 0)
    r0, out0 = codec.decode_batch_host(arc, d)
    st = codec.decode_stats()
    assert st["frame_parallel_entries"] == 0
    for r in (r0, r1):
        assert (r["status"] == 0).all() and [int(h) for h in r["hash"]] == want and [int(x) for x in r["produced"]] == sizes
    for i, p in enumerate(plains):
        assert np.array_equal(out1[i], p) and np.array_equal(out0[i], p), i
    # the checkers on the payloads written in pieces
    o = oracle()
    arc_b = arc.tobytes()
    for i, p in enumerate(plains):
        if sizes[i] > 8 * M:
            continue
        rc, out, got, h = o.entry_decode(arc_b, int(offs[i]), int(cs[i]), sizes[i], want[i], methods[i][0], sizes[i])
        assert rc == 0 and out == p.tobytes(), (i, rc)
    # unsplit writing: other bytes, same plaintext behind them, read the same way
    codec.set_option(OPT_ENC_SPLIT_MIN, 0)
    res2, pay2 = _encode(codec, plains[:5], methods[:5])
    assert [int(h) for h in res2["hash"]] == want[:5]
    arc2, offs2, cs2 = _image(pay2)
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    r2, out2 = codec.decode_batch_host(arc2, _descs(offs2, cs2, sizes[:5], res2["hash"], methods[:5]))
    st = codec.decode_stats()
    assert (r2["status"] == 0).all() and st["frame_parallel_entries"] in (4, 5), st          # (the same entry of random bytes)
    for i in range(5):
        assert np.array_equal(out2[i], plains[i])
    # ---- a sequence of frames with content sizes as ONE entry (archives of round 4; lib/zpack_read.c:380 continues with the next frame) ----
    big = dg.fill(dg.TEXT, 322, 0, 5 * PIECE + 777)
    parts = [big[k:k + PIECE] for k in range(0, len(big), PIECE)]
    resp, payp = _encode(codec, parts, [(METHOD_ZSTD, 3)] * len(parts))
    seq = np.concatenate(payp)
    arc3, offs3, cs3 = _image([seq])
    d3 = _descs(offs3, cs3, [len(big)], [dg.xxh3(big)], [(METHOD_ZSTD, 3)])
    r3, out3 = codec.decode_batch_host(arc3, d3)
    st = codec.decode_stats()
    assert int(r3["status"][0]) == 0 and np.array_equal(out3[0], big) and st["frame_parallel_entries"] == 1 and st["frame_parallel_frames"] == len(parts), st
    codec.set_option(OPT_ENC_SPLIT_MIN, 2 * M)


@pytest.mark.parametrize("cls", [dg.TEXT, dg.RECORDS, dg.RANDOM, dg.RUNS])
def test_one_large_lz4_frame_is_decoded_block_parallel(codec, cls):
    """What the reference writer produces for a large LZ4 entry is ONE frame of linked 64 KiB blocks (lib/zpack_write.c:204-210).  The
    host read path parses its blocks side by side and resolves every output byte to the literal it copies by pointer doubling
    (lz4_pj.h): frames made by liblz4 (the reference's call sequence) of every corpus class — bytes equal the plaintext and the
    one-wave decoder's, the hash is the real xxHash's, the counters say that the block-parallel path ran."""
    sizes = [2 * M, 5 * M + 12345, 9 * M + 1]
    plains = [dg.fill(cls, 77, i, n) for i, n in enumerate(sizes)]
    pay = [np.frombuffer(dg.compress(METHOD_LZ4, 0, p), dtype=np.uint8) for p in plains]
    want = [dg.xxh3(p) for p in plains]
    arc, offs, cs = _image(pay)
    d = _descs(offs, cs, sizes, want, [(METHOD_LZ4, 0)] * len(sizes))
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    r1, out1 = codec.decode_batch_host(arc, d)
    st = codec.decode_stats()
    assert st["frame_parallel_entries"] == len(sizes), st
    codec.set_option(OPT_DEC_SPLIT_MIN, 0)
    r0, out0 = codec.decode_batch_host(arc, d)
    assert codec.decode_stats()["frame_parallel_entries"] == 0
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    for r in (r0, r1):
        assert (r["status"] == 0).all() and [int(h) for h in r["hash"]] == want and [int(x) for x in r["produced"]] == sizes, r
    for i, p in enumerate(plains):
        assert np.array_equal(out1[i], p) and np.array_equal(out0[i], p), i


@pytest.mark.parametrize("level", [1, 3])
@pytest.mark.parametrize("cls", [dg.TEXT, dg.RECORDS, dg.RANDOM, dg.RUNS])
def test_one_large_zstd_frame_is_decoded_block_parallel(codec, cls, level):
    """What the reference writer produces for a large Zstandard entry is ONE frame of blocks of up to 128 KiB (lib/zpack_write.c:179,
    ZSTD_compressCCtx).  The host read path decodes the blocks' sequences and literals side by side — repeat offsets against symbols
    that a scan over the blocks resolves, Treeless literals with the tree of the block they inherit from — and resolves every
    output byte by pointer doubling (zstd_pj.h): frames made by libzstd of every corpus class and two levels — bytes equal the
    plaintext and the one-wave decoder's, the hash is the real xxHash's, the counters say that the block-parallel path ran."""
    sizes = [2 * M, 5 * M + 12345, 9 * M + 1]
    plains = [dg.fill(cls, 78, i, n) for i, n in enumerate(sizes)]
    pay = [np.frombuffer(dg.compress(METHOD_ZSTD, level, p), dtype=np.uint8) for p in plains]
    want = [dg.xxh3(p) for p in plains]
    arc, offs, cs = _image(pay)
    d = _descs(offs, cs, sizes, want, [(METHOD_ZSTD, level)] * len(sizes))
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    r1, out1 = codec.decode_batch_host(arc, d)
    st = codec.decode_stats()
    # (random bytes = stored blocks: one wave copies them at ~1 GiB/s, so by the batch's cost estimate only the largest such entry is
    # worth the block-parallel reader's fixed 4 ms)
    assert st["frame_parallel_entries"] == len(sizes) or (cls == dg.RANDOM and st["frame_parallel_entries"] >= 1), st
    codec.set_option(OPT_DEC_SPLIT_MIN, 0)
    r0, out0 = codec.decode_batch_host(arc, d)
    assert codec.decode_stats()["frame_parallel_entries"] == 0
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    for r in (r0, r1):
        assert (r["status"] == 0).all() and [int(h) for h in r["hash"]] == want and [int(x) for x in r["produced"]] == sizes, r
    for i, p in enumerate(plains):
        assert np.array_equal(out1[i], p) and np.array_equal(out0[i], p), i


@pytest.mark.parametrize("level", [9, 15, 19])
def test_large_zstd_frame_with_repeat_mode_tables_block_parallel(codec, level):
    """From level 9 on libzstd writes blocks whose sequence tables REPEAT an earlier block's (Repeat_Mode: 32 of 128 text blocks at level 9,
    112 at 15).  The host walk measures every table description, so a repeating block is told where the table it inherits is described
    and builds it again itself: such frames are block-parallel too.  Level 19: a window of 8 MiB, matches reaching megabytes back."""
    sizes = [6 * M + 321, 12 * M]
    plains = [dg.fill(dg.TEXT if i == 0 else dg.RECORDS, 79, i, n) for i, n in enumerate(sizes)]
    plains[1][: 6 * M] = dg.fill(dg.TEXT, 80, 1, 6 * M)                      # text, then records: the tables change kind in the middle
    pay = [np.frombuffer(dg.compress(METHOD_ZSTD, level, p), dtype=np.uint8) for p in plains]
    want = [dg.xxh3(p) for p in plains]
    arc, offs, cs = _image(pay)
    d = _descs(offs, cs, sizes, want, [(METHOD_ZSTD, level)] * len(sizes))
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    r1, out1 = codec.decode_batch_host(arc, d)
    st = codec.decode_stats()
    assert st["frame_parallel_entries"] == len(sizes), st
    assert (r1["status"] == 0).all() and [int(h) for h in r1["hash"]] == want, r1
    for i, p in enumerate(plains):
        assert np.array_equal(out1[i], p), i


def test_many_medium_entries_stay_one_wave_each_while_a_huge_one_goes_block_parallel(codec):
    """The block-parallel reader takes one entry at a time (each fills the chip); the usual batch runs all its entries side by side.  A
    batch of many 2-3 MiB entries is done in the time of one of them there, so they stay one wave each; a 40 MiB entry among them is the
    long pole and goes block-parallel.  Same bytes and verdicts either way."""
    sizes = [2 * M + 7 * i for i in range(24)] + [40 * M + 1]
    plains = [dg.fill(i % 2, 97, i, n) for i, n in enumerate(sizes)]
    pay = [np.frombuffer(dg.compress(METHOD_LZ4, 0, p), dtype=np.uint8) for p in plains]
    want = [dg.xxh3(p) for p in plains]
    arc, offs, cs = _image(pay)
    d = _descs(offs, cs, sizes, want, [(METHOD_LZ4, 0)] * len(sizes))
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    r1, out1 = codec.decode_batch_host(arc, d)
    st = codec.decode_stats()
    assert 1 <= st["frame_parallel_entries"] <= 3, st                             # the 40 MiB entry (and at most a couple within the estimate's slack)
    assert (r1["status"] == 0).all() and [int(h) for h in r1["hash"]] == want
    for i, p in enumerate(plains):
        assert np.array_equal(out1[i], p), i
    # the medium entries alone: none is worth a turn of its own
    r2, out2 = codec.decode_batch_host(arc, d[:24])
    st = codec.decode_stats()
    assert st["frame_parallel_entries"] <= 2 and (r2["status"] == 0).all(), st


@pytest.mark.parametrize("method", [METHOD_LZ4, METHOD_ZSTD])
def test_reference_made_large_recipes_block_parallel(codec, golden_dir, method):
    """The LZ4 and Zstandard entries of tests/golden/recipes_big.json (64 MiB text, 64 MiB records, 512 MiB text; sizes, frame checksums
    and content hashes recorded from the compiled reference by tests/golden/make_golden_big.py) through the host read path: every one
    takes the block-parallel path (several chunks of 512 blocks: references that reach behind a chunk read finished output), verdict,
    size, XXH3 and bytes are the reference's — and the 512 MiB entry comes home at more than 2 GiB/s, host pointer to host pointer."""
    import json, time
    recs = [r for r in json.load(open(os.path.join(golden_dir, "recipes_big.json"))) if r["method"] == method]
    assert len(recs) == 3
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    for r in recs:
        plain = dg.fill(r["cls"], r["seed"], r["index"], r["size"])
        frame = np.frombuffer(dg.compress(r["method"], r["level"], plain), dtype=np.uint8)
        assert len(frame) == r["comp_size"] and dg.xxh3(frame) == r["frame_xxh3"], r["label"]
        arc, offs, cs = _image([frame])
        d = _descs(offs, cs, [r["size"]], [r["hash"]], [(method, 0)])
        back = np.full(r["size"], 0xEE, dtype=np.uint8)                        # (touched now: the timed calls do not pay for its page faults)
        bp = (C.c_void_p * 1)(back.ctypes.data)
        res = np.zeros(1, dtype=zpack_amd.DECODE_RESULT)
        best = 1e9
        for _ in range(2):
            back[::4096] = 0xEE
            t = time.perf_counter()
            rc = codec.L.zpk_codec_decode_batch_host(codec.h, arc.ctypes.data, arc.size, d.ctypes.data, 1, bp, res.ctypes.data)
            best = min(best, time.perf_counter() - t)
            assert rc == 0
            st = codec.decode_stats()
            blk = 65536 if method == METHOD_LZ4 else 131072
            assert st["frame_parallel_entries"] == 1 and st["frame_parallel_frames"] == (r["size"] + blk - 1) // blk, (r["label"], st)
            assert int(res["status"][0]) == 0 and int(res["produced"][0]) == r["size"] and int(res["hash"][0]) == r["hash"], (r["label"], res)
            assert np.array_equal(back, plain), r["label"]
        if r["size"] >= 512 * M:
            assert r["size"] / best > 2 * (1 << 30), "512 MiB entry: %.2f GiB/s" % (r["size"] / best / (1 << 30))
        # a wrong expected hash: the reference's verdict for it (lib/zpack_read.c:467), bytes delivered all the same
        d2 = _descs(offs, cs, [r["size"]], [r["hash"] ^ 2], [(method, 0)])
        back[:] = 0
        rc = codec.L.zpk_codec_decode_batch_host(codec.h, arc.ctypes.data, arc.size, d2.ctypes.data, 1, bp, res.ctypes.data)
        assert rc == 0 and int(res["status"][0]) == 15 and int(res["hash"][0]) == r["hash"] and np.array_equal(back, plain), (r["label"], res)
        del plain, frame, arc, back


def test_one_large_lz4_frame_damaged_gets_the_one_wave_verdict(codec):
    """The block-parallel reader finishes an entry only when everything about it was regular; a flipped byte, a short or long comp_size,
    a wrong hash, a small capacity, a frame with checksums or a content size that disagrees: status, produced and bytes are those of
    the one-wave decoder (= the oracle's = the compiled reference's)."""
    o = oracle()
    rng = np.random.default_rng(17)
    plain = dg.fill(dg.TEXT, 91, 0, 3 * M + 999)
    good = np.frombuffer(dg.compress(METHOD_LZ4, 0, plain), dtype=np.uint8).copy()
    h = dg.xxh3(plain)
    variants = []
    for k in range(int(os.environ.get("ZPK_BIG_FUZZ_ITERS", "12"))):
        b = good.copy(); at = int(rng.integers(0, len(b))); b[at] ^= 0x41
        variants.append(("flip@%d" % at, b, len(b), h, len(plain)))
    b = good.copy(); b[4] ^= 0x20                                            # independent blocks claimed (the header checksum then fails)
    variants.append(("flg", b, len(b), h, len(plain)))
    variants.append(("comp_size - 5", good, len(good) - 5, h, len(plain)))
    variants.append(("comp_size + 3", np.concatenate([good, np.zeros(8, np.uint8)]), len(good) + 3, h, len(plain)))
    variants.append(("hash", good, len(good), h ^ 1, len(plain)))
    variants.append(("capacity", good, len(good), h, len(plain) - 1))
    variants.append(("intact", good, len(good), h, len(plain)))
    for label, payload, csize, eh, cap in variants:
        arc, offs, _ = _image([payload])
        d = _descs(offs, [csize], [len(plain)], [eh], [(METHOD_LZ4, 0)])
        d["dst_capacity"] = cap
        codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
        r1, out1 = codec.decode_batch_host(arc, d)
        par = codec.decode_stats()["frame_parallel_entries"]
        codec.set_option(OPT_DEC_SPLIT_MIN, 0)
        r0, out0 = codec.decode_batch_host(arc, d)
        assert int(r1["status"][0]) == int(r0["status"][0]), (label, r1, r0)
        if r0["status"][0] == 0:
            assert r1["hash"][0] == r0["hash"][0] and r1["produced"][0] == r0["produced"][0], label
            assert np.array_equal(out1[0], out0[0]), label
        if label == "intact":
            assert par == 1 and r1["status"][0] == 0 and np.array_equal(out1[0], plain)
        if label == "hash":
            assert par == 1 and r1["status"][0] == 15 and np.array_equal(out1[0][:len(plain)], plain), label
        rc, out, got, hh = o.entry_decode(arc.tobytes(), int(offs[0]), csize, len(plain), eh, METHOD_LZ4, cap)
        assert rc == int(r1["status"][0]), (label, rc, r1)
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)


def test_one_large_zstd_frame_damaged_gets_the_one_wave_verdict(codec):
    """The same for one large Zstandard frame.  Its blocks carry no checksum, so a flipped byte often still decodes — to other bytes:
    the block-parallel reader then sees a wrong XXH3 and leaves the entry, like everything irregular, to the one-wave decoder.
    Status, hash, produced and bytes equal the one-wave path's and the oracle's for every variant."""
    o = oracle()
    rng = np.random.default_rng(23)
    for cls, level in ((dg.TEXT, 3), (dg.RECORDS, 1)):
        plain = dg.fill(cls, 92, 0, 3 * M + 777)
        good = np.frombuffer(dg.compress(METHOD_ZSTD, level, plain), dtype=np.uint8).copy()
        h = dg.xxh3(plain)
        variants = []
        for k in range(int(os.environ.get("ZPK_BIG_FUZZ_ITERS", "16"))):
            b = good.copy(); at = int(rng.integers(0, len(b))); b[at] ^= 1 << int(rng.integers(0, 8))
            variants.append(("flip@%d" % at, b, len(b), h, len(plain)))
        for at in (4, 5, 6, 9, 10, 11, 12, 13):                              # frame header, first block header, first literals header
            b = good.copy(); b[at] ^= 0x10
            variants.append(("head@%d" % at, b, len(b), h, len(plain)))
        variants.append(("comp_size - 5", good, len(good) - 5, h, len(plain)))
        variants.append(("comp_size + 3", np.concatenate([good, np.zeros(8, np.uint8)]), len(good) + 3, h, len(plain)))
        variants.append(("hash", good, len(good), h ^ 1, len(plain)))
        variants.append(("capacity", good, len(good), h, len(plain) - 1))
        variants.append(("intact", good, len(good), h, len(plain)))
        for label, payload, csize, eh, cap in variants:
            arc, offs, _ = _image([payload])
            d = _descs(offs, [csize], [len(plain)], [eh], [(METHOD_ZSTD, level)])
            d["dst_capacity"] = cap
            codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
            r1, out1 = codec.decode_batch_host(arc, d)
            par = codec.decode_stats()["frame_parallel_entries"]
            codec.set_option(OPT_DEC_SPLIT_MIN, 0)
            r0, out0 = codec.decode_batch_host(arc, d)
            assert int(r1["status"][0]) == int(r0["status"][0]), (label, r1, r0)
            if r0["status"][0] in (0, 15):
                assert r1["hash"][0] == r0["hash"][0] and r1["produced"][0] == r0["produced"][0], label
                assert np.array_equal(out1[0], out0[0]), label
            if label == "intact":
                assert par == 1 and r1["status"][0] == 0 and np.array_equal(out1[0], plain)
            if label == "hash":
                assert par == 0 and r1["status"][0] == 15 and np.array_equal(out1[0][:len(plain)], plain), label
            rc, out, got, hh = o.entry_decode(arc.tobytes(), int(offs[0]), csize, len(plain), eh, METHOD_ZSTD, cap)
            assert rc == int(r1["status"][0]), (label, rc, r1)
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 3)])
def test_large_entry_in_device_memory_is_decoded_block_parallel(codec, method, level):
    """zpk_codec_decode_big_device: the entry's compressed bytes are on the device, the output stays on the device (GPU pipelines).  A large
    frame of the reference writer goes block-parallel (its bytes visit the host once, for the walk over the block headers); a damaged
    one, a small one and a stored one get exactly what zpk_codec_decode_batch_device gives them."""
    import torch
    dev = torch.device("cuda:0")
    size = 24 * M + 4321
    plain = dg.fill(dg.TEXT, 95, 0, size)
    frame = np.frombuffer(dg.compress(method, level, plain), dtype=np.uint8)
    h = dg.xxh3(plain)

    def run(payload, usize, want_hash, meth):
        src = torch.zeros(10 + len(payload) + 64, dtype=torch.uint8, device=dev)
        src[10:10 + len(payload)] = torch.from_numpy(np.array(payload, dtype=np.uint8, copy=True)).to(dev)
        dst = torch.full((256 + usize + 64,), 0xEE, dtype=torch.uint8, device=dev)
        d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
        d["src_offset"] = 10; d["comp_size"] = len(payload); d["uncomp_size"] = usize; d["expect_hash"] = want_hash
        d["dst_offset"] = 256; d["dst_capacity"] = usize; d["method"] = meth
        r = codec.decode_big_device(src, d, dst)
        par = codec.decode_stats()["frame_parallel_entries"]
        # the same through the batch call
        dst2 = torch.full((256 + usize + 64,), 0xEE, dtype=torch.uint8, device=dev)
        ddesc = torch.from_numpy(d.view(np.uint8)).to(dev)
        dres = torch.zeros(zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        codec.decode_batch_device(src, ddesc, 1, dst2, dres)
        torch.cuda.synchronize()
        r2 = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)[0]
        return r, par, dst.cpu().numpy(), r2, dst2.cpu().numpy()

    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
    r, par, out, r2, out2 = run(frame, size, h, method)
    assert par == 1 and int(r["status"]) == 0 and int(r["hash"]) == h and int(r["produced"]) == size, (r, par)
    assert np.array_equal(out[256:256 + size], plain) and (out[:256] == 0xEE).all() and (out[256 + size:] == 0xEE).all()
    assert int(r2["status"]) == 0 and np.array_equal(out2[256:256 + size], plain)
    bad = frame.copy(); bad[len(bad) // 2] ^= 0x10
    r, par, out, r2, out2 = run(bad, size, h, method)
    assert int(r["status"]) == int(r2["status"]) != 0 and int(r["hash"]) == int(r2["hash"]), (r, r2)
    r, par, out, r2, out2 = run(frame, size, h ^ 1, method)
    assert int(r["status"]) == 15 == int(r2["status"]) and np.array_equal(out[256:256 + size], plain)
    small = dg.fill(dg.RECORDS, 96, 0, 100000)
    sf = np.frombuffer(dg.compress(method, level, small), dtype=np.uint8)
    r, par, out, r2, out2 = run(sf, len(small), dg.xxh3(small), method)
    assert par == 0 and int(r["status"]) == 0 and np.array_equal(out[256:256 + len(small)], small)
    r, par, out, r2, out2 = run(plain[:3 * M], 3 * M, dg.xxh3(plain[:3 * M]), METHOD_NONE)
    assert int(r["status"]) == 0 and np.array_equal(out[256:256 + 3 * M], plain[:3 * M])


def test_big_entry_damage_gets_the_one_wave_verdict(codec):
    """A frame sequence with a damaged frame, a damaged frame header, a short or long comp_size, a wrong hash, a capacity below the size:
    the frame-parallel reader gives exactly what the one-wave reader gives (status, detail, produced, hash), which is what the oracle
    and the compiled reference say."""
    codec.set_option(OPT_ENC_SPLIT_MIN, 2 * M)
    o = oracle()
    rng = np.random.default_rng(9)
    for method, level in [(METHOD_LZ4, 0), (METHOD_ZSTD, 1), (METHOD_NONE, 0)]:
        plain = dg.fill(dg.TEXT, 55, method, 3 * M + 4321)
        res, pay = _encode(codec, [plain], [(method, level)])
        good = pay[0].copy()
        h = int(res["hash"][0])
        variants = []
        for k in range(int(os.environ.get("ZPK_BIG_FUZZ_ITERS", "10"))):                   # a flipped byte somewhere in the payload (a soak: tools/evidence.sh f)
            b = good.copy(); at = int(rng.integers(0, len(b))); b[at] ^= 0x41
            variants.append(("flip@%d" % at, b, len(b), h, len(plain)))
        b = good.copy(); b[4] ^= 0x08
        variants.append(("first header", b, len(b), h, len(plain)))
        variants.append(("comp_size - 5", good, len(good) - 5, h, len(plain)))
        variants.append(("comp_size + 3", np.concatenate([good, np.zeros(8, np.uint8)]), len(good) + 3, h, len(plain)))
        variants.append(("hash", good, len(good), h ^ 1, len(plain)))
        variants.append(("capacity", good, len(good), h, len(plain) - 1))
        variants.append(("intact", good, len(good), h, len(plain)))
        for label, payload, csize, eh, cap in variants:
            arc, offs, _ = _image([payload])
            d = _descs(offs, [csize], [len(plain)], [eh], [(method, level)])
            d["dst_capacity"] = cap
            codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)
            r1, out1 = codec.decode_batch_host(arc, d)
            par = codec.decode_stats()["frame_parallel_entries"]
            codec.set_option(OPT_DEC_SPLIT_MIN, 0)
            r0, out0 = codec.decode_batch_host(arc, d)
            key = (method, label)
            assert int(r1["status"][0]) == int(r0["status"][0]), (key, r1, r0)
            if r0["status"][0] == 0:
                assert r1["hash"][0] == r0["hash"][0] and r1["produced"][0] == r0["produced"][0], key
                assert np.array_equal(out1[0], out0[0]), key
            if label == "intact":
                assert par == 1 and r1["status"][0] == 0
            if label == "hash":
                # the bytes stay, like the reference's (a Zstandard frame with a wrong XXH3 gets its verdict from the one-wave decoder: its
                # blocks carry no checksum, the block-parallel reader cannot tell a wrong expectation from a damaged block)
                assert par == (0 if method == METHOD_ZSTD else 1) and r1["status"][0] == 15 and np.array_equal(out1[0][:len(plain)], plain), key
            if not label.startswith("flip") or int(label[5:]) % 7 == 0 or len(variants) < 40:
                rc, out, got, hh = o.entry_decode(arc.tobytes(), int(offs[0]), csize, len(plain), eh, method, cap)
                assert rc == int(r1["status"][0]), (key, rc, r1)
    codec.set_option(OPT_DEC_SPLIT_MIN, 2 * M)


def test_big_entry_through_zpack_h_and_the_compiled_reference(tmp_path):
    """zpack_write_file of a 24 MiB source (one call, one entry, 48 frames) and the streaming writer's entry of the same source: both
    are read back by this library (frame-parallel) and by the compiled reference."""
    from tests._libs import ZPackAPI
    Z = ZPackAPI(zpack_amd.ZPACK_SO)
    plain = dg.fill(dg.RECORDS, 8, 1, 24 * M + 99)
    for method, level in [(METHOD_LZ4, 0), (METHOD_ZSTD, 3)]:
        arc = Z.write_archive([("one", plain.tobytes())], method, level)
        rc, r, keep = Z.open_memory(arc)
        assert rc == 0
        rc, out = Z.read_file(r, 0, len(plain))
        assert rc == 0 and out == plain.tobytes()
        Z.close_reader(r)
        if have_ref():
            R = ref()
            rc, r, keep = R.open_memory(arc)
            assert rc == 0
            rc, out = R.read_file(r, 0, len(plain))
            assert rc == 0 and out == plain.tobytes(), "the reference rejects the frame sequence"
            R.close_reader(r)


def test_split_options_through_the_environment_of_zpack_h(monkeypatch):
    """zpack.h has no place for codec options: a context reads ZPACK_AMD_ENC_SPLIT_MIN / _DEC_SPLIT_MIN when it is created.  0 = one
    wave per entry whatever its size (byte-identical archives across library versions), the default cuts at 2 MiB.  Either way an entry
    is ONE frame; an entry written in pieces differs in a few bytes (a piece's first block has no match into the piece before it)."""
    from tests._libs import ZPackAPI
    from tests import zpk
    from tests.test_gpu_zpack_api import _count_frames
    Z = ZPackAPI(zpack_amd.ZPACK_SO)
    plain = dg.fill(dg.TEXT, 8, 2, 3 * M + 5).tobytes()
    payloads = {}
    for setting in ("0", None, str(1 << 20)):
        if setting is None:
            monkeypatch.delenv("ZPACK_AMD_ENC_SPLIT_MIN", raising=False)
        else:
            monkeypatch.setenv("ZPACK_AMD_ENC_SPLIT_MIN", setting)
        arc = Z.write_archive([("f", plain), ("g", plain[:(3 * M) // 2])], METHOD_LZ4, 0)       # (a writer makes its context on first use)
        ents = zpk.parse(arc)
        payloads[setting] = [arc[e["offset"]:e["offset"] + e["comp_size"]] for e in ents]
        assert [_count_frames(x, METHOD_LZ4) for x in payloads[setting]] == [1, 1]
        rc, r, keep = Z.open_memory(arc)
        assert rc == 0
        for i, want in enumerate((plain, plain[:(3 * M) // 2])):
            rc, out = Z.read_file(r, i, len(want))
            assert rc == 0 and out == want
        Z.close_reader(r)
    # f (3 MiB): in pieces by default and at 1 MiB, not at 0; g (1.5 MiB): in pieces only at 1 MiB
    assert payloads[None][0] == payloads[str(1 << 20)][0] != payloads["0"][0]
    assert payloads[None][1] == payloads["0"][1] != payloads[str(1 << 20)][1]
