#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE in the build container.

Needs oracle/_ref/libzpack_ref.so (`make -C oracle ref`: the reference's lib/*.c compiled in place
against this image's lz4 1.9.3 / zstd 1.4.9 / xxHash 0.8.x) and /root/reference/tests/workdir.  It is
never run on the GPU box; what it writes is committed:

  ref_workdir/            the reference's own test data files (archives + plaintexts), verbatim data
  small_archives.json     archives written by the reference writer (zpack_write_archive) for many
                          sizes/classes/methods, inline (hex), with the entry table the reference
                          reader parsed and the bytes/hash its zpack_read_file returned
  recipes.json            larger cases as (class, seed, index, size, method, level) recipes plus the
                          XXH3 of the reference-written frame and of the plaintext
  status_cases.json       guard / malformed-input cases: tampered entries or frames and the
                          zpack_result the reference's zpack_read_file returned for each
  foreign_frames.json     frames with format features the reference writer never emits (checksums,
                          content size, independent / larger LZ4 blocks, multi-frame + skippable zstd,
                          zstd RLE blocks ...), made with the real libraries, verdicts by the reference
"""
import ctypes as C
import json
import os
import shutil
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from benchdata import datagen as dg            # noqa: E402
from tests._libs import ref, oracle, METHOD_NONE, METHOD_ZSTD, METHOD_LZ4   # noqa: E402
import tests._libs as L                        # noqa: E402

REFDIR = "/root/reference/tests/workdir"
R = ref()

METHODS = [("none", METHOD_NONE, 0), ("lz4_0", METHOD_LZ4, 0), ("lz4_9", METHOD_LZ4, 9),
           ("zstd_1", METHOD_ZSTD, 1), ("zstd_3", METHOD_ZSTD, 3), ("zstd_19", METHOD_ZSTD, 19)]


def assemble(payloads, entries):
    """Hand-assemble a .zpk (docs/specs.md): entries = [(name, offset, comp, uncomp, hash, method)]"""
    out = bytearray(b"ZPK\x15" + struct.pack("<H", 1) + b"ZPK\x14")
    for p in payloads:
        out += p
    cdr_off = len(out)
    body = bytearray()
    for (name, off, cs, us, h, m) in entries:
        nb = name.encode()
        body += struct.pack("<H", len(nb)) + nb + struct.pack("<QQQQB", off, cs, us, h, m)
    out += b"ZPK\x13" + struct.pack("<QQ", len(entries), len(body)) + body
    out += b"ZPK\x12" + struct.pack("<Q", cdr_off)
    return bytes(out)


def ref_read(archive, idx, max_size, tamper=None):
    """zpack_read_file of the REFERENCE on entry idx; tamper = dict of entry fields to overwrite first."""
    rc, r, keep = R.open_memory(archive)
    assert rc == 0, rc
    e = r.file_entries[idx]
    if tamper:
        for k, v in tamper.items():
            setattr(e, k, v)
    rc, out = R.read_file(r, idx, max_size)
    R.close_reader(r)
    return rc, out


def small_archives():
    sizes = [0, 1, 2, 3, 4, 5, 8, 9, 16, 17, 32, 33, 64, 65, 96, 97, 128, 129, 160, 240, 241, 256, 511, 1023,
             1024, 1025, 2048, 4096]
    cases = []
    for cls_name, cls in (("text", dg.TEXT), ("records", dg.RECORDS), ("random", dg.RANDOM), ("runs", dg.RUNS)):
        for mname, method, level in METHODS:
            files = [("%s_%d" % (cls_name, n), dg.fill(cls, 11, n, n).tobytes()) for n in sizes]
            arc = R.write_archive(files, method, level)
            rc, r, keep = R.open_memory(arc)
            assert rc == 0
            ents = R.entries(r)
            for i, (name, data) in enumerate(files):
                rc, out = R.read_file(r, i, len(data))
                assert rc == 0 and out == data, (cls_name, mname, name, rc)
                assert ents[i]["hash"] == dg.xxh3(data)
            R.close_reader(r)
            cases.append(dict(corpus=cls_name, cls=cls, seed=11, method=method, level=level, label=mname,
                              sizes=sizes, archive=arc.hex(), entries=ents))
    return cases


def recipes():
    out = []
    sizes = [65536, 65537, 100000, 131072, 262144, 300000, 1048576]
    idx = 0
    for cls_name, cls in (("text", dg.TEXT), ("records", dg.RECORDS), ("random", dg.RANDOM), ("runs", dg.RUNS)):
        for mname, method, level in METHODS[1:5]:
            for n in sizes:
                idx += 1
                data = dg.fill(cls, 21, idx, n).tobytes()
                arc = R.write_archive([("f", data)], method, level)
                rc, r, keep = R.open_memory(arc)
                e = R.entries(r)[0]
                rc, got = R.read_file(r, 0, n)
                R.close_reader(r)
                assert rc == 0 and got == data
                frame = arc[e["offset"]:e["offset"] + e["comp_size"]]
                # the same frame must come out of the datagen path (same libraries, same call sequence)
                assert dg.compress(method, level, data) == frame, (cls_name, mname, n)
                out.append(dict(corpus=cls_name, cls=cls, seed=21, index=idx, size=n, method=method, level=level,
                                label=mname, comp_size=e["comp_size"], frame_xxh3=dg.xxh3(frame), hash=e["hash"]))
    return out


def status_cases():
    """Guards and malformed inputs: what the reference's zpack_read_file returns (lib/zpack_read.c:326-471)."""
    cases = []
    bases = []
    text = dg.fill(dg.TEXT, 31, 0, 3000).tobytes()
    big = dg.fill(dg.RECORDS, 31, 1, 200000).tobytes()

    def add(label, archive, idx, max_size, tamper=None, flips=()):
        if archive.hex() not in bases:
            bases.append(archive.hex())
        b = bytearray(archive)
        for pos, x in flips:
            b[pos] ^= x
        rc, out = ref_read(bytes(b), idx, max_size, tamper)
        cases.append(dict(label=label, base=bases.index(archive.hex()), flips=[list(f) for f in flips], index=idx,
                          max_size=max_size, tamper=tamper or {}, rc=rc, out_xxh3=dg.xxh3(out), out_len=len(out)))
        return rc

    for mname, method, level in (("none", 0, 0), ("lz4", 2, 0), ("zstd", 1, 3)):
        arc = R.write_archive([("a", text), ("b", big[:70000] if method != 0 else text[:100])], method, level)
        rc, r, keep = R.open_memory(arc)
        ents = R.entries(r)
        R.close_reader(r)
        e0 = ents[0]
        n0 = e0["uncomp_size"]
        add(mname + ":ok_exact", arc, 0, n0)
        add(mname + ":ok_roomy", arc, 0, n0 + 100)
        add(mname + ":buffer_too_small", arc, 0, n0 - 1)
        add(mname + ":bad_hash", arc, 0, n0, dict(hash=e0["hash"] ^ 1))
        add(mname + ":offset_past_end", arc, 0, n0, dict(offset=len(arc)))
        add(mname + ":offset_plus_size_eq_filesize", arc, 0, n0 + 100,
            dict(offset=len(arc) - e0["comp_size"]))
        add(mname + ":bad_method", arc, 0, n0, dict(comp_method=7))
        add(mname + ":comp_size_zero", arc, 0, n0, dict(comp_size=0))
        add(mname + ":uncomp_smaller_than_real", arc, 0, n0, dict(uncomp_size=n0 - 10))
        add(mname + ":uncomp_larger_than_real", arc, 0, n0 + 50, dict(uncomp_size=n0 + 10))
        if method != 0:
            add(mname + ":truncated_1", arc, 0, n0, dict(comp_size=e0["comp_size"] - 1))
            add(mname + ":truncated_4", arc, 0, n0, dict(comp_size=e0["comp_size"] - 4))
            add(mname + ":truncated_half", arc, 0, n0, dict(comp_size=e0["comp_size"] // 2))
            add(mname + ":truncated_to_5", arc, 0, n0, dict(comp_size=5))
            add(mname + ":truncated_to_3", arc, 0, n0, dict(comp_size=3))
            add(mname + ":shifted_start", arc, 0, n0, dict(offset=e0["offset"] + 1, comp_size=e0["comp_size"] - 1))
            add(mname + ":trailing_garbage", arc, 0, n0, dict(comp_size=e0["comp_size"] + 8))
            # second (multi-block for lz4) entry
            e1 = ents[1]
            add(mname + ":multi_ok", arc, 1, e1["uncomp_size"])
            add(mname + ":multi_small_buf", arc, 1, e1["uncomp_size"] - 1)
            add(mname + ":multi_trunc", arc, 1, e1["uncomp_size"], dict(comp_size=e1["comp_size"] - 7))
            # corrupt bytes inside the frame
            for pos_name, pos in (("magic", 0), ("hdr", 4), ("early", 9), ("mid", e0["comp_size"] // 2),
                                  ("late", e0["comp_size"] - 6)):
                for bit in (0x01, 0x80):
                    add("%s:flip_%s_%02x" % (mname, pos_name, bit), arc, 0, n0, flips=[(e0["offset"] + pos, bit)])
        else:
            add(mname + ":uncomp_gt_comp", arc, 0, n0 + 10, dict(uncomp_size=n0 + 1))
    # round 4, late: an entry whose size field and hash agree with each other but not with the frame — :466 hashes buffer[0, uncomp_size),
    # whatever the decoder produced: a prefix of a longer output, or the output followed by what the caller's buffer held (zeros here)
    for mname, method, level in (("none", 0, 0), ("lz4", 2, 0), ("zstd", 1, 3)):
        arc = R.write_archive([("a", text), ("b", big[:70000] if method != 0 else text[:100])], method, level)
        n0 = len(text)
        add(mname + ":size_and_hash_of_a_prefix", arc, 0, n0, dict(uncomp_size=n0 - 10, hash=dg.xxh3(text[:n0 - 10])))
        add(mname + ":size_and_hash_of_a_prefix_exact_buffer", arc, 0, n0 - 10, dict(uncomp_size=n0 - 10, hash=dg.xxh3(text[:n0 - 10])))
        add(mname + ":size_and_hash_with_zero_tail", arc, 0, n0 + 50, dict(uncomp_size=n0 + 10, hash=dg.xxh3(text + bytes(10))))
        if method != 0:
            add(mname + ":multi_size_and_hash_of_a_prefix", arc, 1, 70000, dict(uncomp_size=66000, hash=dg.xxh3(big[:66000])))
            # (no zero-tail case for the large entry: libzstd's wide copies leave bytes of their own just behind a long output — the
            # reference returns FILE_HASH_MISMATCH there through an accident of the library, not through anything zpack does)
    return dict(bases=bases, cases=cases)


# ---- frames made directly with the real libraries (features the reference writer never emits) ----

def _lz4f_custom(data, block_id=0, block_indep=0, content_cksum=0, block_cksum=0, content_size=0, level=0):
    lz4 = C.CDLL("/opt/conda/lib/liblz4.so.1")

    class FrameInfo(C.Structure):
        _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int),
                    ("frameType", C.c_int), ("contentSize", C.c_ulonglong), ("dictID", C.c_uint),
                    ("blockChecksumFlag", C.c_int)]

    class Prefs(C.Structure):
        _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                    ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]
    p = Prefs()
    p.frameInfo.blockSizeID = block_id
    p.frameInfo.blockMode = block_indep
    p.frameInfo.contentChecksumFlag = content_cksum
    p.frameInfo.blockChecksumFlag = block_cksum
    p.frameInfo.contentSize = len(data) if content_size else 0
    p.compressionLevel = level
    lz4.LZ4F_compressFrameBound.restype = C.c_size_t
    lz4.LZ4F_compressFrameBound.argtypes = [C.c_size_t, C.POINTER(Prefs)]
    lz4.LZ4F_compressFrame.restype = C.c_size_t
    lz4.LZ4F_compressFrame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(Prefs)]
    cap = lz4.LZ4F_compressFrameBound(len(data), C.byref(p))
    out = C.create_string_buffer(cap)
    n = lz4.LZ4F_compressFrame(out, cap, data, len(data), C.byref(p))
    assert n < (1 << 62), "LZ4F_compressFrame failed"
    return out.raw[:n]


def _zstd_custom(data, level=3, checksum=0, content_size=1, window_log=0, flush_every=0, pledged=False):
    z = C.CDLL("/opt/conda/lib/libzstd.so.1")
    z.ZSTD_createCCtx.restype = C.c_void_p
    z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    z.ZSTD_CCtx_setParameter.restype = C.c_size_t
    z.ZSTD_compressStream2.restype = C.c_size_t
    z.ZSTD_freeCCtx.argtypes = [C.c_void_p]

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
    z.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(Buf), C.POINTER(Buf), C.c_int]
    cctx = z.ZSTD_createCCtx()
    z.ZSTD_CCtx_setParameter(cctx, 100, level)           # ZSTD_c_compressionLevel
    z.ZSTD_CCtx_setParameter(cctx, 201, checksum)        # ZSTD_c_checksumFlag
    z.ZSTD_CCtx_setParameter(cctx, 200, content_size)    # ZSTD_c_contentSizeFlag
    if window_log:
        z.ZSTD_CCtx_setParameter(cctx, 101, window_log)  # ZSTD_c_windowLog
    if pledged:                                          # the frame header then carries Frame_Content_Size
        z.ZSTD_CCtx_setPledgedSrcSize.argtypes = [C.c_void_p, C.c_ulonglong]
        z.ZSTD_CCtx_setPledgedSrcSize.restype = C.c_size_t
        assert z.ZSTD_CCtx_setPledgedSrcSize(cctx, len(data)) < (1 << 62)
    cap = len(data) + len(data) // 8 + 4096
    out = C.create_string_buffer(cap)
    ob = Buf(C.cast(out, C.c_void_p), cap, 0)
    src = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
    step = flush_every or len(data) or 1
    pos = 0
    while pos < len(data):
        n = min(step, len(data) - pos)
        ib = Buf(C.cast(src, C.c_void_p).value + pos, n, 0)
        while ib.pos < ib.size:
            r = z.ZSTD_compressStream2(cctx, C.byref(ob), C.byref(ib), 1 if flush_every else 0)   # flush / continue
            assert r < (1 << 62)
        pos += n
    ib = Buf(C.cast(src, C.c_void_p), 0, 0)
    while True:
        r = z.ZSTD_compressStream2(cctx, C.byref(ob), C.byref(ib), 2)                              # ZSTD_e_end
        assert r < (1 << 62)
        if r == 0:
            break
    z.ZSTD_freeCCtx(cctx)
    return out.raw[:ob.pos]


def foreign_frames():
    cases = []
    text = dg.fill(dg.TEXT, 41, 0, 12000).tobytes()
    recs = dg.fill(dg.RECORDS, 41, 1, 9000).tobytes()
    runs = dg.fill(dg.RUNS, 41, 2, 300000).tobytes()
    rnd = dg.fill(dg.RANDOM, 41, 3, 3000).tobytes()
    big_rnd = dg.fill(dg.RANDOM, 41, 4, 66000).tobytes()
    const = b"\x5a" * 200000
    few = bytes((i * 7 + (i >> 3)) % 5 + 65 for i in range(5000))     # 5-symbol alphabet
    tiny_alpha = bytes((i % 3) + 48 for i in range(40)) + b"xyz" * 3

    def add(label, method, frame, plain, max_size=None):
        h = dg.xxh3(plain)
        arc = assemble([frame], [("f", 10, len(frame), len(plain), h, method)])
        ms = len(plain) if max_size is None else max_size
        rc, out = ref_read(arc, 0, ms)
        cases.append(dict(label=label, method=method, frame=frame.hex(), uncomp_size=len(plain), hash=h,
                          max_size=ms, rc=rc, out_xxh3=dg.xxh3(out[:len(plain)]), plain_xxh3=h))
        return rc

    # LZ4F variants
    add("lz4f:content_checksum", 2, _lz4f_custom(text, content_cksum=1), text)
    add("lz4f:block_checksum", 2, _lz4f_custom(text, block_cksum=1), text)
    add("lz4f:content_size", 2, _lz4f_custom(text, content_size=1), text)
    add("lz4f:independent_blocks", 2, _lz4f_custom(text, block_indep=1), text)
    add("lz4f:256k_blocks", 2, _lz4f_custom(runs, block_id=5), runs)
    add("lz4f:1m_blocks_hc", 2, _lz4f_custom(runs + text, block_id=6, level=9), runs + text)
    add("lz4f:4m_blocks_all_flags", 2, _lz4f_custom(recs, block_id=7, content_cksum=1, block_cksum=1, content_size=1), recs)
    add("lz4f:stored_blocks_with_checksums", 2, _lz4f_custom(big_rnd, content_cksum=1, block_cksum=1), big_rnd)
    f = bytearray(_lz4f_custom(text, content_cksum=1)); f[-1] ^= 0x10
    add("lz4f:bad_content_checksum", 2, bytes(f), text)
    f = bytearray(_lz4f_custom(text, block_cksum=1)); f[20] ^= 0x01
    add("lz4f:bad_block_checksum", 2, bytes(f), text)
    f = bytearray(_lz4f_custom(text)); f[6] ^= 0x01
    add("lz4f:bad_header_checksum", 2, bytes(f), text)
    skip = struct.pack("<II", 0x184D2A53, 5) + b"hello"
    add("lz4f:skippable_then_frame", 2, skip + _lz4f_custom(recs), recs)

    # zstd variants
    add("zstd:content_checksum", 1, _zstd_custom(text, checksum=1), text)
    f = bytearray(_zstd_custom(text, checksum=1)); f[-2] ^= 0x04
    add("zstd:bad_content_checksum", 1, bytes(f), text)
    add("zstd:no_content_size_streamed", 1, _zstd_custom(text, content_size=0), text)
    add("zstd:flushed_blocks_repeat_modes", 1, _zstd_custom(text, level=3, flush_every=3000), text)
    add("zstd:flushed_blocks_lvl1_records", 1, _zstd_custom(recs, level=1, flush_every=2000), recs)
    add("zstd:rle_blocks", 1, _zstd_custom(const, level=3), const)
    add("zstd:runs_level19", 1, _zstd_custom(runs, level=19), runs)
    add("zstd:small_alphabet", 1, _zstd_custom(few, level=3), few)
    add("zstd:tiny_alphabet_direct_weights", 1, _zstd_custom(tiny_alpha, level=19), tiny_alpha)
    add("zstd:window_log_10", 1, _zstd_custom(text + runs[:40000] + text, level=3, window_log=10), text + runs[:40000] + text)
    add("zstd:negative_level", 1, _zstd_custom(text, level=-7), text)
    two = _zstd_custom(text[:6000], level=3) + _zstd_custom(text[6000:], level=5)
    add("zstd:two_concatenated_frames", 1, two, text)
    skipz = struct.pack("<II", 0x184D2A5E, 7) + b"skip me"
    add("zstd:skippable_between_frames", 1, _zstd_custom(recs[:1000]) + skipz + _zstd_custom(recs[1000:]), recs)
    add("zstd:skippable_only_then_frame", 1, skipz + _zstd_custom(rnd), rnd)
    add("zstd:frame_fcs_smaller_than_capacity", 1, _zstd_custom(text[:5000]), text[:5000], max_size=9000)

    # round 4: several LZ4 frames in one entry.  The reference's one-shot reader calls LZ4F_decompress in a loop
    # `while (avail_out > 0 && avail_in > 0)` (lib/zpack_read.c:414-439): a completed frame returns 0 and the loop goes on, so
    # frames may follow each other; what is left when the loop stops decides between OK, FILE_INCOMPLETE and BUFFER_TOO_SMALL;
    # fewer than 7 bytes are never looked at (LZ4F's minimal header size).  Also what a chunked streaming writer may emit.
    f1, f2 = _lz4f_custom(text[:6000]), _lz4f_custom(text[6000:], level=9)
    add("lz4f:two_concatenated_frames", 2, f1 + f2, text)
    add("lz4f:two_concatenated_frames_loose_capacity", 2, f1 + f2, text, max_size=len(text) + 1000)
    add("lz4f:three_frames_middle_empty", 2, f1 + _lz4f_custom(b"") + f2, text)
    add("lz4f:skippable_between_frames", 2, _lz4f_custom(recs[:1000]) + skip + _lz4f_custom(recs[1000:]), recs)
    add("lz4f:frame_then_skippable", 2, f1 + skip, text[:6000], max_size=7000)
    add("lz4f:skippable_only", 2, skip, b"")
    add("lz4f:frame_then_5_bytes_capacity_full", 2, f1 + b"\x01\x02\x03\x04\x05", text[:6000])
    add("lz4f:frame_then_5_bytes", 2, f1 + b"\x01\x02\x03\x04\x05", text[:6000], max_size=7000)
    add("lz4f:frame_then_7_bytes", 2, f1 + b"\x01\x02\x03\x04\x05\x06\x07", text[:6000], max_size=7000)
    add("lz4f:frame_then_magic_only", 2, f1 + f1[:4], text[:6000], max_size=7000)
    add("lz4f:frame_then_truncated_frame", 2, f1 + f2[:-9], text)
    add("lz4f:frame_then_truncated_frame_loose", 2, f1 + f2[:-9], text, max_size=len(text) + 100)
    add("lz4f:six_bytes_of_nothing", 2, b"\x01\x02\x03\x04\x05\x06", b"x" * 100)
    add("lz4f:seven_bytes_of_nothing", 2, b"\x01\x02\x03\x04\x05\x06\x07", b"x" * 100)
    f = bytearray(_lz4f_custom(text, content_size=1)); f[5] |= 0x01      # a reserved BD bit AND (below) a header cut short: which is judged first
    add("lz4f:bad_bd_full_header", 2, bytes(f), text)
    add("lz4f:bad_bd_header_cut_short", 2, bytes(f[:10]), text)
    g = bytearray(_lz4f_custom(text, content_size=1)); g[4] |= 0x02      # the reserved FLG bit
    add("lz4f:bad_flg_header_cut_short", 2, bytes(g[:10]), text)

    # round 4, late: sequences of frames that each state their content size, outputs on 256-byte boundaries — what this library's writers
    # emit for large / streamed entries and what its host reader decodes one wave per FRAME.  The verdicts below are the reference's.
    import xxhash
    cuts = [0, 4096, 4096 + 6144, 12000]
    zf = [_zstd_custom(text[a:b], level=(3, 1, 5)[k], checksum=(k == 2), pledged=True) for k, (a, b) in enumerate(zip(cuts, cuts[1:]))]
    lf = [_lz4f_custom(text[a:b], content_size=1, content_cksum=(k == 1), block_cksum=(k == 2)) for k, (a, b) in enumerate(zip(cuts, cuts[1:]))]
    add("zstd:three_frames_with_sizes", 1, b"".join(zf), text)
    add("lz4f:three_frames_with_sizes", 2, b"".join(lf), text)
    for name, fs, m in (("zstd", zf, 1), ("lz4f", lf, 2)):
        mid = bytearray(fs[1]); mid[len(mid) // 2] ^= 0x10
        add(name + ":three_frames_middle_damaged", m, fs[0] + bytes(mid) + fs[2], text)
        add(name + ":three_frames_last_cut", m, fs[0] + fs[1] + fs[2][:-6], text)
        add(name + ":three_frames_then_garbage", m, b"".join(fs) + b"\x01\x02\x03\x04\x05\x06\x07\x08\x09", text, max_size=len(text) + 64)
        add(name + ":three_frames_capacity_short", m, b"".join(fs), text, max_size=len(text) - 1)
        add(name + ":three_frames_sizes_add_up_to_less", m, fs[0] + fs[1], text)
        add(name + ":three_frames_wrong_order", m, fs[1] + fs[0] + fs[2], text)
    lie = bytearray(lf[1]); lie[6:14] = struct.pack("<Q", 6144 + 256)                 # the content size field lies; header checksum left as it was
    add("lz4f:three_frames_size_field_changed", 2, lf[0] + bytes(lie) + lf[2], text)
    lie[14] = (xxhash.xxh32(bytes(lie[4:14]), seed=0).intdigest() >> 8) & 0xFF        # ... and with a header checksum that agrees with the lie
    add("lz4f:three_frames_size_field_lies_consistently", 2, lf[0] + bytes(lie) + lf[2], text, max_size=len(text) + 256)
    zl = bytearray(zf[1])                                                             # zstd: the 2-byte FCS (6144 - 256) changed to claim 256 bytes more
    fcs_at = 5 + (0 if zl[4] & 0x20 else 1)
    if zl[4] >> 6 == 1: zl[fcs_at:fcs_at + 2] = struct.pack("<H", 6144 - 256 + 256)
    elif zl[4] >> 6 == 2: zl[fcs_at:fcs_at + 4] = struct.pack("<I", 6144 + 256)
    else: raise RuntimeError("unexpected frame header %02x" % zl[4])
    add("zstd:three_frames_size_field_lies", 1, zf[0] + bytes(zl) + zf[2], text, max_size=len(text) + 256)
    return cases


def main():
    # 1. the reference's own test data, verbatim
    dst = os.path.join(HERE, "ref_workdir")
    os.makedirs(dst, exist_ok=True)
    for f in ("archive_none.zpk", "archive_zstd.zpk", "archive_lz4.zpk", "file1.txt", "file2.txt"):
        shutil.copyfile(os.path.join(REFDIR, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)

    def dump(name, obj):
        with open(os.path.join(HERE, name), "w") as fh:
            json.dump(obj, fh, indent=0, separators=(",", ":"))
        print(name, os.path.getsize(os.path.join(HERE, name)), "bytes")

    dump("small_archives.json", small_archives())
    dump("recipes.json", recipes())
    dump("status_cases.json", status_cases())
    dump("foreign_frames.json", foreign_frames())
    # known answers captured while surveying the reference, re-derived here from the reference build
    ka = {}
    arc = R.write_archive([("e", b"")], METHOD_ZSTD, 3)
    ka["xxh3_empty"] = R.entries(R.open_memory(arc)[1])[0]["hash"]
    for mname, method, level in (("zstd", 1, 3), ("lz4", 2, 0)):
        for label, data in (("empty", b""), ("r", b"r")):
            a = R.write_archive([("e", data)], method, level)
            rc, r, keep = R.open_memory(a)
            e = R.entries(r)[0]
            ka["%s_%s_frame" % (mname, label)] = a[e["offset"]:e["offset"] + e["comp_size"]].hex()
            ka["%s_%s_hash" % (mname, label)] = e["hash"]
    ka["ref_hashes"] = {"file1.txt": 0x7874cba47d02b07d, "file2.txt": 0x15f25c0f24dd8e52}   # tests/archive.h:112-115
    for m in (0, 1, 2):
        ka["dstream_in_%d" % m] = R.lib.zpack_get_dstream_in_size(m)
        ka["dstream_out_%d" % m] = R.lib.zpack_get_dstream_out_size(m)
        ka["cstream_in_%d" % m] = R.lib.zpack_get_cstream_in_size(m)
        ka["cstream_out_%d" % m] = R.lib.zpack_get_cstream_out_size(m)
    dump("known_answers.json", ka)


if __name__ == "__main__":
    main()
