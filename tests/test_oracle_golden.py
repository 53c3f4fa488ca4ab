"""Pin the oracle (oracle/liboracle.so) — CPU only.

Every function of the CPU restatement is checked against
  (a) the reference's own test data (tests/golden/ref_workdir = /root/reference/tests/workdir, hashes
      of /root/reference/tests/archive.h:112-115; this is what tests/read_archive.c:21-35 asserts), and
  (b) fixtures produced by running the compiled reference (tests/golden/make_golden.py).
"""
import json
import os

import numpy as np
import pytest

from benchdata import datagen as dg
from tests import zpk
from tests._libs import oracle

REF_HASHES = {"file1.txt": 0x7874cba47d02b07d, "file2.txt": 0x15f25c0f24dd8e52}


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as fh:
        return json.load(fh)


@pytest.mark.parametrize("arc", ["archive_none.zpk", "archive_zstd.zpk", "archive_lz4.zpk"])
def test_reference_bundled_archives(golden_dir, arc):
    """tests/read_archive.c:21-35 + tests/open_archive.c:21-25, re-stated on the oracle."""
    o = oracle()
    wd = os.path.join(golden_dir, "ref_workdir")
    a = open(os.path.join(wd, arc), "rb").read()
    ents = zpk.parse(a)
    assert [e["filename"] for e in ents] == ["file1.txt", "file2.txt"]
    for e in ents:
        plain = open(os.path.join(wd, e["filename"]), "rb").read()
        assert e["uncomp_size"] == len(plain) and e["hash"] == REF_HASHES[e["filename"]]
        assert o.xxh3(plain) == REF_HASHES[e["filename"]]
        rc, out, got, h = o.entry_decode(a, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"], 350)
        assert rc == 0 and out[:len(plain)] == plain and h == e["hash"]


def test_known_answers(golden_dir):
    o = oracle()
    ka = _load(golden_dir, "known_answers.json")
    assert o.xxh3(b"") == ka["xxh3_empty"] == 0x2d06800538d394c2
    assert o.xxh3(b"r") == ka["zstd_r_hash"] == 0xbd7b38f7b4af0b35
    for m, dec in (("zstd", o.zstd_decode), ("lz4", o.lz4f_decode)):
        rc, out = dec(bytes.fromhex(ka[m + "_empty_frame"]), 16)
        assert rc == 0 and out == b""
        rc, out = dec(bytes.fromhex(ka[m + "_r_frame"]), 16)
        assert rc == 0 and out == b"r"
    # the empty LZ4 frame the reference writes is byte-identical to what the oracle encoder emits
    assert o.lz4f_encode(b"").hex() == ka["lz4_empty_frame"]


def test_small_archives_written_by_reference(golden_dir):
    o = oracle()
    n = 0
    for case in _load(golden_dir, "small_archives.json"):
        a = bytes.fromhex(case["archive"])
        ents = zpk.parse(a)
        assert ents == case["entries"]
        for e, size in zip(ents, case["sizes"]):
            plain = dg.fill(case["cls"], case["seed"], size, size).tobytes()
            assert o.xxh3(plain) == e["hash"]
            rc, out, got, h = o.entry_decode(a, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"], size)
            assert rc == 0, (case["label"], case["corpus"], size, rc)
            assert out == plain and (size == 0 or h == e["hash"])
            n += 1
    assert n == 24 * 28


def test_recipes_large_frames(golden_dir):
    o = oracle()
    for r in _load(golden_dir, "recipes.json"):
        plain = dg.fill(r["cls"], r["seed"], r["index"], r["size"])
        frame = dg.compress(r["method"], r["level"], plain)
        # same libraries + same call sequence as the reference writer => the very frame it wrote
        assert len(frame) == r["comp_size"] and dg.xxh3(frame) == r["frame_xxh3"], r
        dec = o.zstd_decode if r["method"] == 1 else o.lz4f_decode
        rc, out = dec(frame, r["size"])
        assert rc == 0, r
        assert out == plain.tobytes()
        assert o.xxh3(out) == r["hash"]


def test_recipes_big_entries(golden_dir):
    """the 64 MiB recipes of recipes_big.json (made with the compiled reference): the oracle decodes the same frames to the
    same bytes and XXH3 (the 512 MiB ones are left to the GPU suite: the CPU suite has to stay short)"""
    o = oracle()
    for r in _load(golden_dir, "recipes_big.json"):
        if r["size"] > (64 << 20) or r["method"] == 0:
            continue
        plain = dg.fill(r["cls"], r["seed"], r["index"], r["size"])
        frame = dg.compress(r["method"], r["level"], plain)
        assert len(frame) == r["comp_size"] and dg.xxh3(frame) == r["frame_xxh3"], r["label"]
        dec = o.zstd_decode if r["method"] == 1 else o.lz4f_decode
        rc, out = dec(frame, r["size"])
        assert rc == 0 and o.xxh3(out) == r["hash"] and out == plain.tobytes(), r["label"]


def test_status_codes_match_reference(golden_dir):
    o = oracle()
    sc = _load(golden_dir, "status_cases.json")
    for c in sc["cases"]:
        a = bytearray(bytes.fromhex(sc["bases"][c["base"]]))
        for p, x in c["flips"]:
            a[p] ^= x
        e = zpk.parse(a)[c["index"]]
        for k, v in c["tamper"].items():
            e[{"comp_method": "method"}.get(k, k)] = v
        rc, out, got, h = o.entry_decode(bytes(a), e["offset"], e["comp_size"], e["uncomp_size"], e["hash"],
                                         e["method"], c["max_size"])
        assert rc == c["rc"], (c["label"], rc, c["rc"])
        if rc in (0, 15):       # bytes are defined when decode succeeded (hash mismatch leaves data in place)
            assert dg.xxh3(out) == c["out_xxh3"], c["label"]


def test_foreign_frames_match_reference(golden_dir):
    o = oracle()
    for c in _load(golden_dir, "foreign_frames.json"):
        fr = bytes.fromhex(c["frame"])
        arc = zpk.assemble([fr], [("f", 10, len(fr), c["uncomp_size"], c["hash"], c["method"])])
        rc, out, got, h = o.entry_decode(arc, 10, len(fr), c["uncomp_size"], c["hash"], c["method"], c["max_size"])
        assert rc == c["rc"], c["label"]
        if rc == 0:
            assert dg.xxh3(out[:c["uncomp_size"]]) == c["plain_xxh3"], c["label"]


def test_xxh3_all_length_classes_and_streaming():
    """XXH3 vs the real xxHash header (benchdata links it) across every length class."""
    o = oracle()
    lens = list(range(0, 260)) + [511, 512, 513, 1023, 1024, 1025, 1087, 1088, 1089, 2047, 2048, 2049, 4095,
                                   4096, 4097, 65535, 65536, 65537, 100003, 262144]
    for n in lens:
        d = dg.fill(dg.RANDOM, 5, n, n)
        want = dg.xxh3(d)
        assert o.xxh3(d) == want, n
        b = d.tobytes()
        assert o.xxh3_stream([b[:n // 3], b[n // 3:n // 3 + 1], b[n // 3 + 1:]]) == want, n
        assert o.xxh3_stream([b[i:i + 16] for i in range(0, n, 16)]) == want, n


def test_oracle_encoders_decode_with_real_libraries():
    """Compressed bytes are unpinned by the reference; validity = the real decoders reproduce the input."""
    import ctypes as C
    o = oracle()
    lz4 = C.CDLL("/opt/conda/lib/liblz4.so.1") if os.path.exists("/opt/conda/lib/liblz4.so.1") else C.CDLL("liblz4.so.1")
    zstd = C.CDLL("/opt/conda/lib/libzstd.so.1") if os.path.exists("/opt/conda/lib/libzstd.so.1") else C.CDLL("libzstd.so.1")
    zstd.ZSTD_decompress.restype = C.c_size_t
    zstd.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lz4.LZ4F_createDecompressionContext.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    lz4.LZ4F_decompress.restype = C.c_size_t
    lz4.LZ4F_decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p]
    lz4.LZ4F_freeDecompressionContext.argtypes = [C.c_void_p]
    for cls in (dg.TEXT, dg.RECORDS, dg.RANDOM, dg.RUNS):
        for n in (0, 1, 12, 13, 100, 4096, 65535, 65536, 65537, 200000):
            plain = dg.fill(cls, 9, n, n).tobytes()
            f = o.zstd_encode(plain)
            out = C.create_string_buffer(max(n, 1))
            got = zstd.ZSTD_decompress(out, n, f, len(f))
            assert got == n and out.raw[:n] == plain
            f = o.lz4f_encode(plain)
            assert len(f) <= o.lib.orc_lz4f_bound(n)
            ctx = C.c_void_p()
            assert lz4.LZ4F_createDecompressionContext(C.byref(ctx), 100) == 0
            dn, sn = C.c_size_t(n), C.c_size_t(len(f))
            r = lz4.LZ4F_decompress(ctx, out, C.byref(dn), f, C.byref(sn), None)
            lz4.LZ4F_freeDecompressionContext(ctx)
            assert r == 0 and dn.value == n and sn.value == len(f) and out.raw[:n] == plain, (cls, n, r)
            if cls != dg.RANDOM and n >= 4096:
                assert len(f) < n


def test_huffman_12_bit_code_frame():
    """The hand-built frame of tests/huf12.py (12-bit Huffman literals): the oracle decodes it, and so does the real
    libzstd of the image when it is there."""
    import ctypes as C
    from tests.huf12 import make_frame
    o = oracle()
    for seed, n in ((1, 200), (2, 1000), (3, 17)):
        frame, lits = make_frame(seed, n)
        rc, out = o.zstd_decode(frame, n)
        assert rc == 0 and out == lits
        for so in ("/opt/conda/lib/libzstd.so.1", "libzstd.so.1"):
            try:
                z = C.CDLL(so)
            except OSError:
                continue
            z.ZSTD_decompress.restype = C.c_size_t
            buf = (C.c_uint8 * n)()
            assert z.ZSTD_decompress(buf, n, frame, len(frame)) == n and bytes(buf) == lits
            break
