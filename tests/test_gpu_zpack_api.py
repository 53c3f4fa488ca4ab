"""The reference's own three test programs (tests/open_archive.c, tests/read_archive.c, tests/write_archive.c),
re-stated against libzpack_amd.so through the zpack.h ABI — same flow, same buffer sizes (350-byte output
buffer, 16-byte streaming input window), same golden archives.  write_archive is checked harder than the
reference does (it only looks at return codes): every archive written here must read back bit-exactly
through the oracle decoders (and through the compiled reference where it is present)."""
import ctypes as C
import os

import numpy as np
import pytest

import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import (ZPackAPI, Reader, Writer, Stream, File, CompressOptions, FileEntry, u8p, oracle, have_ref, ref,
                         METHOD_NONE, METHOD_ZSTD, METHOD_LZ4)

pytestmark = pytest.mark.gpu

ARCHIVES = ["archive_none.zpk", "archive_zstd.zpk", "archive_lz4.zpk"]
FILES = ["file1.txt", "file2.txt"]
HASHES = [0x7874cba47d02b07d, 0x15f25c0f24dd8e52]          # /root/reference/tests/archive.h:112-115
BUFFER_SIZE, STREAM_IN_SIZE = 350, 16                       # tests/read_archive.c:11-13


@pytest.fixture(scope="module")
def Z():
    return ZPackAPI(zpack_amd.ZPACK_SO)


def _wd(golden_dir, name):
    return os.path.join(golden_dir, "ref_workdir", name)


def _open_three_ways(Z, golden_dir, arc):
    """file reader, copied-buffer reader, shared-buffer reader (tests/open_archive.c:59,72,85)"""
    raw = open(_wd(golden_dir, arc), "rb").read()
    keep = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
    for how in ("file", "copy", "shared"):
        r = Reader()
        if how == "file":
            rc = Z.lib.zpack_init_reader(C.byref(r), _wd(golden_dir, arc).encode())
        elif how == "copy":
            rc = Z.lib.zpack_init_reader_memory(C.byref(r), C.cast(keep, u8p), len(raw))
        else:
            rc = Z.lib.zpack_init_reader_memory_shared(C.byref(r), C.cast(keep, u8p), len(raw))
        assert rc == 0, (arc, how, rc)
        yield how, r
        Z.lib.zpack_close_reader(C.byref(r))
        assert bytes(r) == bytes(C.sizeof(Reader))          # close re-zeroes the struct (lib/zpack_read.c:692-717)


@pytest.mark.parametrize("arc", ARCHIVES)
def test_open_archive(Z, golden_dir, arc):
    for how, r in _open_three_ways(Z, golden_dir, arc):
        assert r.file_count == 2 and r.version == 1
        for i, e in enumerate(Z.entries(r)):
            plain = open(_wd(golden_dir, FILES[i]), "rb").read()
            assert e["filename"] == FILES[i] and e["uncomp_size"] == len(plain) and e["hash"] == HASHES[i]


@pytest.mark.parametrize("arc", ARCHIVES)
def test_read_archive_oneshot_and_streaming(Z, golden_dir, arc):
    for how, r in _open_three_ways(Z, golden_dir, arc):
        # ---- one-shot (tests/read_archive.c:21-35)
        for i in range(2):
            plain = open(_wd(golden_dir, FILES[i]), "rb").read()
            rc, out = Z.read_file(r, i, BUFFER_SIZE)
            assert rc == 0, (arc, how, i, rc, r.last_return)
            assert out[:len(plain)] == plain
        # ---- streaming with a 16-byte input window (tests/read_archive.c:38-82)
        st = Stream()
        assert Z.lib.zpack_init_stream(C.byref(st)) == 0
        in_buf = (C.c_uint8 * STREAM_IN_SIZE)()
        for i in range(2):
            plain = open(_wd(golden_dir, FILES[i]), "rb").read()
            out = (C.c_uint8 * BUFFER_SIZE)()
            Z.lib.zpack_reset_stream(C.byref(st))
            st.next_out = C.cast(out, u8p)
            st.avail_out = BUFFER_SIZE
            e = r.file_entries[i]
            for passes in range(10000):
                if st.read_back:
                    tail = C.string_at(C.addressof(st.next_in.contents) - st.read_back, st.read_back)
                    C.memmove(in_buf, tail, st.read_back)
                st.next_in = C.cast(in_buf, u8p)
                st.avail_in = STREAM_IN_SIZE
                st.avail_out = BUFFER_SIZE
                rc = Z.lib.zpack_read_file_stream(C.byref(r), C.byref(e), C.byref(st), None)
                assert rc == 0, (arc, how, i, passes, rc)
                if st.total_in == e.comp_size and st.read_back == 0:
                    break
            else:
                raise AssertionError("stream never finished")
            assert bytes(out[:len(plain)]) == plain and st.total_out == len(plain)
        Z.lib.zpack_close_stream(C.byref(st))


def _decode_all_with_checkers(arc_bytes, want):
    """every entry of an archive we wrote must decode bit-exactly with the oracle, and with the compiled reference if present"""
    o = oracle()
    ents = zpk.parse(arc_bytes)
    assert [e["filename"] for e in ents] == [n for n, _ in want]
    for e, (name, data) in zip(ents, want):
        assert e["uncomp_size"] == len(data) and e["hash"] == dg.xxh3(data)
        rc, out, got, h = o.entry_decode(arc_bytes, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"], len(data))
        assert rc == 0 and out == data, (name, rc)
    if have_ref():
        R = ref()
        rc, r, keep = R.open_memory(arc_bytes)
        assert rc == 0
        for i, (name, data) in enumerate(want):
            rc, out = R.read_file(r, i, len(data))
            assert rc == 0 and out == data, ("reference rejects our archive", name, rc)
        R.close_reader(r)


@pytest.mark.parametrize("method,level", [(METHOD_ZSTD, 3), (METHOD_LZ4, 1), (METHOD_NONE, 0)])   # tests/write_archive.c:31-41
@pytest.mark.parametrize("sink", ["file", "heap"])
def test_write_archive_oneshot_and_streaming(Z, golden_dir, tmp_path, method, level, sink):
    want = [(n, open(_wd(golden_dir, n), "rb").read()) for n in FILES]
    # ---- one-shot (tests/write_archive.c:112-191 -> zpack_write_archive)
    w = Writer()
    path = str(tmp_path / "out.zpk")
    rc = Z.lib.zpack_init_writer(C.byref(w), path.encode()) if sink == "file" else Z.lib.zpack_init_writer_heap(C.byref(w), 0)
    assert rc == 0
    opts = CompressOptions(method, level)
    arr = (File * 2)()
    keep = []
    for i, (n, data) in enumerate(want):
        b = (C.c_uint8 * len(data)).from_buffer_copy(data)
        keep.append(b)
        arr[i].filename = n.encode(); arr[i].buffer = C.cast(b, u8p); arr[i].size = len(data); arr[i].options = C.pointer(opts)
    rc = Z.lib.zpack_write_archive(C.byref(w), arr, 2)
    assert rc == 0, (rc, w.last_return)
    if sink == "heap":
        arc = bytes(C.cast(w.buffer, C.POINTER(C.c_uint8 * w.file_size)).contents)
    Z.lib.zpack_close_writer(C.byref(w))
    if sink == "file":
        arc = open(path, "rb").read()
    _decode_all_with_checkers(arc, want)

    # ---- streaming in 16-byte chunks (tests/write_archive.c:45-110)
    w = Writer()
    path = str(tmp_path / "out_s.zpk")
    rc = Z.lib.zpack_init_writer(C.byref(w), path.encode()) if sink == "file" else Z.lib.zpack_init_writer_heap(C.byref(w), 0)
    assert rc == 0
    assert Z.lib.zpack_write_header(C.byref(w)) == 0 and Z.lib.zpack_write_data_header(C.byref(w)) == 0
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    out_size = Z.lib.zpack_get_cstream_out_size(METHOD_NONE)
    out_buf = (C.c_uint8 * out_size)()
    st.next_out = C.cast(out_buf, u8p); st.avail_out = out_size
    for i, (n, data) in enumerate(want):
        Z.lib.zpack_reset_stream(C.byref(st))
        st.next_in = arr[i].buffer
        while st.total_in < len(data):
            st.avail_in = min(STREAM_IN_SIZE, len(data) - st.total_in)
            assert Z.lib.zpack_write_file_stream(C.byref(w), C.byref(opts), C.byref(st), None) == 0
        assert Z.lib.zpack_write_file_stream_end(C.byref(w), n.encode(), C.byref(opts), C.byref(st), None) == 0
    Z.lib.zpack_close_stream(C.byref(st))
    assert Z.lib.zpack_write_cdr(C.byref(w)) == 0 and Z.lib.zpack_write_eocdr(C.byref(w)) == 0
    if sink == "heap":
        arc2 = bytes(C.cast(w.buffer, C.POINTER(C.c_uint8 * w.file_size)).contents)
    Z.lib.zpack_close_writer(C.byref(w))
    if sink == "file":
        arc2 = open(path, "rb").read()
    _decode_all_with_checkers(arc2, want)
    assert arc2 == arc                                           # same container either way


def test_c5_config_zstd1_one_mib_sources(Z):
    """BASELINE.json configs[4] in small: zpack_write_files of 1 MiB sources at Zstandard level 1 (16 linked 64 KiB blocks per entry:
    k_encode<12> with matches across blocks), every corpus class; frames decoded by the oracle, the compiled reference and the GPU."""
    want = [("c5_%d_%d" % (cls, k), dg.fill(cls, 4, 100 * cls + k, 1 << 20).tobytes()) for cls in range(4) for k in range(3)]
    arc = Z.write_archive(want, METHOD_ZSTD, 1)
    _decode_all_with_checkers(arc, want)
    ents = zpk.parse(arc)
    by_class = {}
    for e, (n, d) in zip(ents, want):
        by_class.setdefault(int(n.split("_")[1]), []).append(e["comp_size"] / len(d))
    # records / runs: the repeat-offset search (round 3) — without it 0.559 / 0.107; libzstd-1: 0.506 / 0.014
    assert max(by_class[dg.TEXT]) < 0.40 and max(by_class[dg.RECORDS]) < 0.53 and max(by_class[dg.RUNS]) < 0.05, by_class
    assert max(by_class[dg.RANDOM]) < 1.001                          # raw blocks: never larger than the frame overhead
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    for i, (n, d) in enumerate(want):
        rc, out = Z.read_file(r, i, len(d))
        assert rc == 0 and out[:len(d)] == d, (n, rc)
    Z.lib.zpack_close_reader(C.byref(r))


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 3), (METHOD_ZSTD, 1), (METHOD_NONE, 0)])
def test_batch_write_then_batch_read_roundtrip(Z, method, level):
    """zpack_write_files (n files, one device batch) then the additive zpack_read_files / _packed."""
    rng = np.random.default_rng(5)
    sizes = [0, 1, 11, 12, 13, 200, 4095, 65535, 65536, 65537, 200000, 1 << 20] + [int(x) for x in rng.integers(1, 150000, 40)]
    want = [("f%03d" % i, dg.fill(i % 4, 77, i, n).tobytes()) for i, n in enumerate(sizes)]
    arc = Z.write_archive(want, method, level)
    _decode_all_with_checkers(arc, want)
    if method in (METHOD_LZ4, METHOD_ZSTD):                      # the device compressors must actually compress text / records / runs
        ents = zpk.parse(arc)
        for e, (n, d) in zip(ents, want):
            if len(d) >= 4095 and int(n[1:]) % 4 != dg.RANDOM:
                assert e["comp_size"] < 0.92 * len(d), (n, e["comp_size"], len(d))
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    n = r.file_count
    ptrs = (C.POINTER(FileEntry) * n)(*[C.pointer(r.file_entries[i]) for i in range(n)])
    Z.lib.zpack_read_files.argtypes = [C.POINTER(Reader), C.POINTER(C.POINTER(FileEntry)), C.c_uint64, C.POINTER(u8p),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_void_p]
    outs = [(C.c_uint8 * max(1, len(d)))() for _, d in want]
    bufs = (u8p * n)(*[C.cast(o, u8p) for o in outs])
    caps = (C.c_size_t * n)(*[len(d) for _, d in want])
    results = (C.c_int * n)()
    assert Z.lib.zpack_read_files(C.byref(r), ptrs, n, bufs, caps, results, None) == 0
    for i, (name, d) in enumerate(want):
        assert results[i] == 0, (name, results[i])
        assert bytes(outs[i][:len(d)]) == d
    # packed form
    Z.lib.zpack_read_files_packed.argtypes = [C.POINTER(Reader), C.POINTER(C.POINTER(FileEntry)), C.c_uint64, u8p, C.c_size_t,
                                              C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.c_void_p]
    total = sum(len(d) for _, d in want)
    big = (C.c_uint8 * total)()
    offs = (C.c_uint64 * n)()
    assert Z.lib.zpack_read_files_packed(C.byref(r), ptrs, n, C.cast(big, u8p), total, offs, results, None) == 0
    blob = bytes(big)
    for i, (name, d) in enumerate(want):
        assert results[i] == 0 and blob[offs[i]:offs[i] + len(d)] == d
    # one bad entry must not poison the batch: corrupt entry 5's hash
    r.file_entries[5].hash ^= 1
    assert Z.lib.zpack_read_files(C.byref(r), ptrs, n, bufs, caps, results, None) == 0
    assert results[5] == 15 and all(results[i] == 0 for i in range(n) if i != 5 and len(want[i][1]))
    Z.close_reader(r)


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 1), (METHOD_ZSTD, 3)])
def test_encoder_long_literal_runs(Z, method, level):
    """Blocks that hold a few sequences with literal runs of tens of KiB (random bytes, a 64-byte marker every `gap` bytes): the runs beyond
    256 bytes are copied by the whole wave in the LZ4 block writer and in the Zstandard literal gather; lengths around the switch too."""
    rng = np.random.default_rng(11)
    want = []
    for i, gap in enumerate([30000, 3000, 700, 300, 257, 256, 255, 97]):
        a = rng.integers(0, 256, (1 << 18) + 13 * i, dtype=np.uint8)
        mk = rng.integers(0, 256, 64, dtype=np.uint8)
        for p in range(gap, len(a) - 64, gap + 64):
            a[p:p + 64] = mk
        want.append(("ll%02d" % i, a.tobytes()))
    arc = Z.write_archive(want, method, level)
    _decode_all_with_checkers(arc, want)


def test_batch_write_spans_staging_pieces(Z):
    """zpack_write_files with entries larger than, and straddling, the 32 MiB pinned staging pieces the sources are gathered into on
    their way up (h2d_gather) and the payloads come back through (d2h_scatter): every entry decodes back to its source."""
    sizes = [40 << 20, 7, 0, (30 << 20) + 13, 1 << 20, 65537, (33 << 20) + 1, 12345]
    want = [("g%02d" % i, dg.fill(i % 2, 91, i, n).tobytes()) for i, n in enumerate(sizes)]
    arc = Z.write_archive(want, METHOD_LZ4, 0)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0 and r.file_count == len(want)
    for i, (name, d) in enumerate(want):
        rc, out = Z.read_file(r, i, max(1, len(d)))
        assert rc == 0 and out[:len(d)] == d, (name, rc)
    Z.close_reader(r)


@pytest.mark.parametrize("method,level", [(METHOD_ZSTD, 1), (METHOD_LZ4, 0)])
def test_encoder_stress_alphabets_and_boundaries(Z, method, level):
    """Inputs aimed at the device encoders' corner cases — literal-heavy data over alphabets of 2..129 symbols with
    geometric / Fibonacci-like frequencies (Huffman depth beyond 11 bits -> the length limiter, 1-bit codes, the
    128-weight limit of the direct tree description -> FSE-compressed weights), short periods, constant data, and sizes around the
    64 KiB block and the 2 KiB Huffman threshold.  Every archive must decode bit-exactly with the oracle and with stock
    libzstd / liblz4 (compiled reference)."""
    rng = np.random.default_rng(12345)
    want = []

    def add(name, arr):
        want.append(("%s_%03d" % (name, len(want)), np.asarray(arr, dtype=np.uint8).tobytes()))

    for nsym in (2, 3, 17, 64, 127, 128, 129, 200, 256):       # > 128 symbols: FSE-compressed Huffman weights (RFC 8878 4.2.1.1)
        base = 0 if nsym <= 128 or nsym == 256 else 40
        for kind in ("geom", "fib", "flat"):
            if kind == "geom":
                p = 0.5 ** np.arange(1, nsym + 1, dtype=np.float64); p[-1] += 1 - p.sum()
            elif kind == "fib":
                f = [1.0, 1.0]
                while len(f) < nsym:
                    f.append(f[-1] + f[-2] if f[-1] < 1e12 else f[-1])
                p = np.array(f[:nsym][::-1]); p /= p.sum()
            else:
                p = np.full(nsym, 1.0 / nsym)
            p = np.maximum(p, 1e-9); p /= p.sum()
            add("%s%d" % (kind, nsym), base + rng.choice(nsym, size=70000, p=p))
    for n in (2047, 2048, 2049, 4096, 65535, 65536, 65537, 131071, 131072, 131073):
        add("sz", 97 + rng.choice(26, size=n, p=np.array([2.0 ** -(i // 3 + 1) for i in range(26)]) / sum(2.0 ** -(i // 3 + 1) for i in range(26))))
    add("const", np.full(100000, 7))
    add("period3", np.tile([1, 2, 3], 40000))
    add("period257", np.tile(np.arange(257) % 251, 300))
    add("zeros_then_text", np.concatenate([np.zeros(70000, dtype=np.uint8), dg.fill(dg.TEXT, 3, 0, 70000)]))
    arc = Z.write_archive(want, method, level)
    _decode_all_with_checkers(arc, want)
    ents = {e["filename"]: e for e in zpk.parse(arc)}          # skewed data must shrink, flat 200-symbol data must not blow up
    for name, data in want:
        if name.startswith("geom") or name.startswith("const") or name.startswith("period"):
            assert ents[name]["comp_size"] < 0.8 * len(data), (name, ents[name]["comp_size"])
        assert ents[name]["comp_size"] <= len(data) + len(data) // 200 + 64, (name, ents[name]["comp_size"])


@pytest.mark.parametrize("method", [METHOD_LZ4, METHOD_ZSTD])
def test_compression_level_is_honoured(Z, method):
    """zpack_compress_options.level reaches the codec (the reference hands it to ZSTD_compressCCtx / LZ4F preferences,
    lib/zpack_write.c:179,199-201): more effort never makes the archive larger on compressible data and makes it smaller somewhere,
    and every level's archive decodes bit-exactly with the oracle and with stock liblz4 / libzstd (the compiled reference)."""
    want = [("t%d" % i, dg.fill(dg.TEXT, 51, i, 300000).tobytes()) for i in range(3)] + \
           [("r%d" % i, dg.fill(dg.RECORDS, 51, i, 300000).tobytes()) for i in range(3)] + \
           [("u%d" % i, dg.fill(dg.RUNS, 51, i, 100000).tobytes()) for i in range(2)]
    sizes = {}
    for level in (1, 3, 9):
        arc = Z.write_archive(want, method, level)
        _decode_all_with_checkers(arc, want)
        sizes[level] = {e["filename"]: e["comp_size"] for e in zpk.parse(arc)}
    tot = {l: sum(v.values()) for l, v in sizes.items()}
    assert tot[3] < tot[1] and tot[9] < tot[3], tot
    for name in sizes[1]:
        if name[0] in "tr":
            assert sizes[9][name] <= sizes[3][name] * 1.002 and sizes[3][name] <= sizes[1][name] * 1.002, (name, [sizes[l][name] for l in (1, 3, 9)])
    print("method %d: total compressed bytes by level %s" % (method, tot))


def _rss_bytes():
    with open("/proc/self/statm") as fh:
        return int(fh.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def _device_used():
    import torch
    free, total = torch.cuda.mem_get_info()
    return total - free


def _stream_entry(Z, r, i, in_window, out_window, sink, dev=None):
    """zpack_read_file_stream with the caller's loop of tests/read_archive.c:38-82 / programs/commands.c:326-400.
    -> (rc of the last call, total_in at the first output byte, peak RSS growth while streaming); dev (a dict): its "peak" becomes the
    growth of the device's used memory while streaming (sampled every 16 calls)"""
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    in_buf = (C.c_uint8 * in_window)()
    out_buf = (C.c_uint8 * out_window)()
    e = r.file_entries[i]
    Z.lib.zpack_reset_stream(C.byref(st))
    rss0, rss_peak, first_out_in, pos, rc = _rss_bytes(), 0, None, 0, 0
    idle = 0
    dev0 = _device_used() if dev is not None else 0
    for it in range(10_000_000):
        if dev is not None and it % 16 == 8:
            dev["peak"] = max(dev.get("peak", 0), _device_used() - dev0)
        before_in = st.total_in
        if st.read_back:
            tail = C.string_at(C.addressof(st.next_in.contents) - st.read_back, st.read_back)
            C.memmove(in_buf, tail, st.read_back)
        st.next_in = C.cast(in_buf, u8p); st.avail_in = in_window
        st.next_out = C.cast(out_buf, u8p); st.avail_out = out_window
        rc = Z.lib.zpack_read_file_stream(C.byref(r), C.byref(e), C.byref(st), None)
        got = out_window - st.avail_out
        if got:
            if first_out_in is None:
                first_out_in = st.total_in
            sink[pos:pos + got] = np.frombuffer(out_buf, dtype=np.uint8, count=got)
            pos += got
        rss_peak = max(rss_peak, _rss_bytes() - rss0)
        if rc not in (0,) or (st.total_in == e.comp_size and st.total_out == e.uncomp_size and st.read_back == 0):
            break
        idle = idle + 1 if (got == 0 and st.total_in == before_in) else 0
        if idle >= 3:                            # (an entry whose header claims more output than its frame holds: nothing more will come)
            break
    Z.lib.zpack_close_stream(C.byref(st))
    return rc, first_out_in, rss_peak, pos


@pytest.mark.parametrize("label", ["lz4_64M", "zstd_64M", "lz4_512M", "zstd_512M", "none_96M"])
def test_stream_read_is_bounded_and_incremental(Z, golden_dir, label):
    """lib/zpack_read.c:515-640 decodes chunk by chunk.  Here: the reference-made 64 MiB and 512 MiB recipes (and a stored entry)
    through zpack_read_file_stream with a 128 KiB input window and a 1 MiB output window — every byte right, the hash verdict OK,
    the FIRST output byte leaves long before the last input byte arrives, and the process holds no copy of the entry on the host
    (resident-set growth while streaming stays far below the compressed size; the stream's buffers live on the device)."""
    import json
    if label.startswith("none"):
        size = 96 << 20
        plain = dg.fill(dg.TEXT, 3, 0, size)
        frame, method, want_hash = plain, METHOD_NONE, dg.xxh3(plain)
    else:
        recs = {("%s_%dM" % ("lz4" if x["method"] == 2 else "zstd", x["size"] >> 20)): x for x in json.load(open(os.path.join(golden_dir, "recipes_big.json"))) if x["method"] in (1, 2)}
        x = recs[label]
        plain = dg.fill(x["cls"], x["seed"], x["index"], x["size"])
        frame = np.frombuffer(dg.compress(x["method"], x["level"], plain), dtype=np.uint8)
        assert len(frame) == x["comp_size"] and dg.xxh3(frame) == x["frame_xxh3"]
        size, method, want_hash = x["size"], x["method"], x["hash"]
    arc = zpk.assemble([frame.tobytes()], [("big", 10, len(frame), size, want_hash, method)])
    sink = np.full(size, 0xEE, dtype=np.uint8)                                             # (touched now: its pages are not growth later)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    dev = {}
    rc, first_out_in, rss_peak, got = _stream_entry(Z, r, 0, 131075, 1 << 20, sink, dev)
    Z.lib.zpack_close_reader(C.byref(r))
    assert rc == 0 and got == size and np.array_equal(sink, plain)
    assert first_out_in is not None and first_out_in <= 4 * 131075, first_out_in          # output after the first block(s), not after the last input byte
    if not label.startswith("none"):                                                      # (round 5) bounded on the DEVICE too: a window of the entry, in block-parallel steps
        assert dev["peak"] < (64 << 20), dev
    assert first_out_in < len(frame) // 8
    if len(frame) >= (128 << 20):                                                         # (smaller entries drown in allocator noise)
        assert rss_peak < (64 << 20), (rss_peak, len(frame))                              # no host copy of the compressed entry, let alone the output


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 3)])
@pytest.mark.parametrize("in_window,out_window", [(131075, 4096), (3000, 70000), (1 << 20, 1 << 16), (70001, 1 << 22)])
def test_stream_read_in_bounded_steps_with_odd_windows(Z, method, level, in_window, out_window):
    """The bounded block-parallel steps behind zpack_read_file_stream with windows that do not fit their grain: an output window far
    smaller than a step's output (the reader pulls no input while output waits), an input window smaller than a block (the host keeps
    the bytes until blocks are complete), windows larger than a step: every byte, the verdict with the last one."""
    size = (5 << 20) + 12345
    plain = dg.fill(dg.TEXT, 31, 7, size)
    frame = np.frombuffer(dg.compress(method, level, plain), dtype=np.uint8)
    arc = zpk.assemble([frame.tobytes()], [("odd", 10, len(frame), size, dg.xxh3(plain), method)])
    sink = np.zeros(size, dtype=np.uint8)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    rc, first_out_in, rss_peak, got = _stream_entry(Z, r, 0, in_window, out_window, sink)
    Z.lib.zpack_close_reader(C.byref(r))
    assert rc == 0 and got == size and np.array_equal(sink, plain), (rc, got)


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 3), (METHOD_NONE, 0)])
@pytest.mark.parametrize("claimed", [(1 << 64) - 1, 0xAAAAAAAAAAAAA000, (1 << 64) - 64, 1 << 47])
def test_stream_read_survives_lying_entry_sizes(Z, method, level, claimed):
    """comp_size / uncomp_size reach zpack_read_file_stream from the CDR, unverified (lib/zpack_read.c:515-640 never trusts them for a
    size computation of its own either: its library streams into the caller's window).  A stream must never size a device buffer
    from them: an entry that CLAIMS 2^64 - 1 output bytes (or a value whose 1.5 x wraps to a few KiB) decodes what its frame
    really holds — every byte right — and ends with the verdict of a frame that stops short of the claim; nothing is written or
    read out of bounds, nothing allocates the claim."""
    size = 300000
    plain = dg.fill(dg.TEXT, 17, 0, size)
    frame = plain.tobytes() if method == METHOD_NONE else bytes(dg.compress(method, level, plain))
    arc = zpk.assemble([frame], [("liar", 10, len(frame), size, dg.xxh3(plain), method)])
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    r.file_entries[0].uncomp_size = claimed
    sink = np.zeros(size + (1 << 20), dtype=np.uint8)
    rc, first_out_in, rss_peak, got = _stream_entry(Z, r, 0, 131075, 1 << 20, sink)
    Z.lib.zpack_close_reader(C.byref(r))
    if method == METHOD_NONE:
        assert rc == 18, rc                         # ZPACK_ERROR_FILE_SIZE_INVALID: lib/zpack_read.c:539 (uncomp_size > comp_size)
    else:
        # the frame is complete after `size` bytes: the hash of the PRODUCED bytes is right, total_out never reaches the claim —
        # the caller's loop ends on the verdict or on input exhaustion; what matters here: the bytes, and no fault
        assert rc in (0, 15, 17, 13), rc
        assert got == size and np.array_equal(sink[:size], plain)
    # and a comp_size beyond anything a device holds is refused before any arithmetic on it
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    r.file_entries[0].comp_size = claimed
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    in_buf = (C.c_uint8 * 4096)(); out_buf = (C.c_uint8 * 4096)()
    st.next_in = C.cast(in_buf, u8p); st.avail_in = 4096
    st.next_out = C.cast(out_buf, u8p); st.avail_out = 4096
    rc = Z.lib.zpack_read_file_stream(C.byref(r), C.byref(r.file_entries[0]), C.byref(st), None)
    assert rc != 0, rc                              # FILE_OFFSET_INVALID (the reader's own guard) or MALLOC_FAILED — never a decode
    Z.lib.zpack_close_stream(C.byref(st))
    Z.lib.zpack_close_reader(C.byref(r))


@pytest.mark.parametrize("method,level,size", [(METHOD_LZ4, 0, 512 << 20), (METHOD_ZSTD, 1, 512 << 20), (METHOD_ZSTD, 3, 96 << 20), (METHOD_NONE, 0, 96 << 20),
                                               (METHOD_LZ4, 0, (4 << 20) + 1), (METHOD_ZSTD, 3, 512 << 10), (METHOD_ZSTD, 3, (512 << 10) + 1)])
def test_stream_write_is_bounded_and_incremental(Z, tmp_path, method, level, size):
    """lib/zpack_write.c:461-685 compresses and emits per call.  Here: a large source through zpack_write_file_stream with a 128 KiB
    input window — compressed bytes reach the archive long before the last input byte (first output within the first MiB), the host
    holds no copy of the entry (resident-set growth far below the source), the device holds a few MiB of it (free device memory
    hardly moves), and the archive — an entry made of 512 KiB frames — decodes bit-exactly with the checker, with the GPU reader
    (one-shot and streaming) and with the compiled reference where it is present.  Sizes around the piece boundary included."""
    import torch
    plain = dg.fill(dg.TEXT if method != METHOD_NONE else dg.RECORDS, 29, 0, size)
    w = Writer()
    path = str(tmp_path / "stream.zpk")
    assert Z.lib.zpack_init_writer(C.byref(w), path.encode()) == 0
    assert Z.lib.zpack_write_header(C.byref(w)) == 0 and Z.lib.zpack_write_data_header(C.byref(w)) == 0
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    opts = CompressOptions(method, level)
    out_size = Z.lib.zpack_get_cstream_out_size(method)
    out_buf = (C.c_uint8 * out_size)()
    st.next_out = C.cast(out_buf, u8p); st.avail_out = out_size
    Z.lib.zpack_reset_stream(C.byref(st))
    window = 131072
    base = plain.ctypes.data
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rss0, rss_peak, dev_peak, first_out_at = _rss_bytes(), 0, 0, None
    pos = 0
    while pos < size:
        n = min(window, size - pos)
        st.next_in = C.cast(base + pos, u8p); st.avail_in = n
        assert Z.lib.zpack_write_file_stream(C.byref(w), C.byref(opts), C.byref(st), None) == 0
        pos += n
        if first_out_at is None and st.total_out > 0:
            first_out_at = pos
        if (pos >> 17) % 64 == 0:
            rss_peak = max(rss_peak, _rss_bytes() - rss0)
            dev_peak = max(dev_peak, free0 - torch.cuda.mem_get_info()[0])
    assert st.total_in == size
    assert Z.lib.zpack_write_file_stream_end(C.byref(w), b"big", C.byref(opts), C.byref(st), None) == 0
    Z.lib.zpack_close_stream(C.byref(st))
    assert Z.lib.zpack_write_cdr(C.byref(w)) == 0 and Z.lib.zpack_write_eocdr(C.byref(w)) == 0
    Z.lib.zpack_close_writer(C.byref(w))
    if size > (1 << 20):
        assert first_out_at is not None and first_out_at <= (1 << 20), first_out_at       # output long before the last input byte
    if size >= (96 << 20):
        assert rss_peak < (64 << 20), rss_peak                                             # no host copy of the entry
        assert dev_peak < (192 << 20), dev_peak                                            # a few MiB of plaintext + frames (+ the codec's scratch) on the device, not the entry
    arc = open(path, "rb").read()
    ents = zpk.parse(arc)
    assert len(ents) == 1 and ents[0]["uncomp_size"] == size and ents[0]["hash"] == dg.xxh3(plain) and ents[0]["method"] == method
    e = ents[0]
    o = oracle()
    rc, out, got, h = o.entry_decode(arc, e["offset"], e["comp_size"], size, e["hash"], method, size)
    assert rc == 0 and np.array_equal(np.frombuffer(out, dtype=np.uint8, count=size), plain)
    if have_ref():
        R = ref()
        rc, r, keep = R.open_memory(arc)
        assert rc == 0
        rc, out = R.read_file(r, 0, size)
        R.close_reader(r)
        assert rc == 0 and np.array_equal(np.frombuffer(out, dtype=np.uint8, count=size), plain), ("reference rejects the streamed entry", rc)
    # the GPU reader: one-shot, and streamed back with a 128 KiB window
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    rc, out = Z.read_file(r, 0, size)
    assert rc == 0 and np.array_equal(np.frombuffer(out, dtype=np.uint8, count=size), plain)
    if size <= (96 << 20):
        sink = np.zeros(size, dtype=np.uint8)
        rc, _, _, got = _stream_entry(Z, r, 0, 131075, 1 << 20, sink)
        assert rc == 0 and got == size and np.array_equal(sink, plain)
    Z.lib.zpack_close_reader(C.byref(r))


def test_stream_read_small_windows_do_not_launch_per_call(Z):
    """tests/read_archive.c:38-82 feeds 16-byte input windows.  The stream gathers small chunks on the host (256 KiB) and steps the
    device once per gathered buffer (or at the entry's last byte): a 1.2 MiB LZ4 entry read through 16-byte windows is decoded in a
    handful of launches, not in 40 000 — counters from zpk_dstream_counters."""
    size = 3 << 20
    plain = dg.fill(dg.TEXT, 31, 0, size)
    frame = bytes(dg.compress(METHOD_LZ4, 0, plain))
    arc = zpk.assemble([frame], [("small-windows", 10, len(frame), size, dg.xxh3(plain), METHOD_LZ4)])
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    Z.lib.zpack_reset_stream(C.byref(st))
    in_buf = (C.c_uint8 * 16)(); out_buf = (C.c_uint8 * 65536)()
    e = r.file_entries[0]
    sink = np.zeros(size, dtype=np.uint8)
    pos = calls = 0
    while not (st.total_in == e.comp_size and st.total_out == e.uncomp_size and st.read_back == 0):
        if st.read_back:
            tail = C.string_at(C.addressof(st.next_in.contents) - st.read_back, st.read_back)
            C.memmove(in_buf, tail, st.read_back)
        st.next_in = C.cast(in_buf, u8p); st.avail_in = 16
        st.next_out = C.cast(out_buf, u8p); st.avail_out = 65536
        rc = Z.lib.zpack_read_file_stream(C.byref(r), C.byref(e), C.byref(st), None)
        assert rc == 0, rc
        got = 65536 - st.avail_out
        sink[pos:pos + got] = np.frombuffer(out_buf, dtype=np.uint8, count=got); pos += got
        calls += 1
        assert calls < 400000
    assert pos == size and np.array_equal(sink, plain)
    # the stream's device-side counters: xxh3_state -> zi_stream_state { zpk_dstream* d; ... }
    d = C.cast(st.xxh3_state, C.POINTER(C.c_void_p))[0]
    L = zpack_amd.lib()
    L.zpk_dstream_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    launches, served = C.c_uint64(0), C.c_uint64(0)
    L.zpk_dstream_counters(d, C.byref(launches), C.byref(served))
    assert served.value >= len(frame) // 16 and launches.value <= len(frame) // (256 << 10) + 2, (launches.value, served.value, len(frame))
    Z.lib.zpack_close_stream(C.byref(st))
    Z.lib.zpack_close_reader(C.byref(r))


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 3)])
def test_stream_read_with_window_sizes_that_change_between_calls(Z, method, level):
    """zpack_stream lets avail_in differ from call to call (lib/zpack_read.c:515-640 reads whatever window it is handed).  Two entries
    in a row on ONE stream, each fed a 16-byte window first (gathered on the host, no launch) and 1 MiB windows afterwards (the
    entry's first DEVICE step then starts with bytes already counted): the resume record of the first entry must not leak into the
    second (round-4 advisor finding: freshness was decided after the gather branch had moved the bytes)."""
    sizes = (3 << 20, (2 << 20) + 12345)
    plains = [dg.fill(dg.TEXT if k == 0 else dg.RECORDS, 77 + k, k, n) for k, n in enumerate(sizes)]
    frames = [bytes(dg.compress(method, level, p_)) for p_ in plains]
    ents, off = [], 10
    for k, f in enumerate(frames):
        ents.append(("e%d" % k, off, len(f), sizes[k], dg.xxh3(plains[k]), method)); off += len(f)
    arc = zpk.assemble(frames, ents)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    big = 1 << 20
    in_buf = (C.c_uint8 * big)(); out_buf = (C.c_uint8 * big)()
    for k in range(2):
        e = r.file_entries[k]
        Z.lib.zpack_reset_stream(C.byref(st))
        sink = np.zeros(sizes[k], dtype=np.uint8)
        pos = calls = 0
        while not (st.total_in == e.comp_size and st.total_out == e.uncomp_size and st.read_back == 0):
            win = 16 if calls == 0 else big
            if st.read_back:
                tail = C.string_at(C.addressof(st.next_in.contents) - st.read_back, st.read_back)
                C.memmove(in_buf, tail, st.read_back)
            st.next_in = C.cast(in_buf, u8p); st.avail_in = win
            st.next_out = C.cast(out_buf, u8p); st.avail_out = big
            rc = Z.lib.zpack_read_file_stream(C.byref(r), C.byref(e), C.byref(st), None)
            assert rc == 0, (k, calls, rc)
            got = big - st.avail_out
            sink[pos:pos + got] = np.frombuffer(out_buf, dtype=np.uint8, count=got); pos += got
            calls += 1
            assert calls < 10000
        assert pos == sizes[k] and np.array_equal(sink, plains[k]), k
    Z.lib.zpack_close_stream(C.byref(st))
    Z.lib.zpack_close_reader(C.byref(r))


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 1)])
def test_host_write_batch_of_many_ragged_entries(method, level):
    """zpk_codec_encode_batch_host (what zpack_write_files calls once per batch) on 15 000 entries of ragged sizes (~750 MB: the sources
    go up through two dozen staging pieces that cut through entries anywhere, the payloads come back as one packed stream): every
    result OK, every XXH3 that of its source, every payload decodes back (the read pipeline) to its source."""
    codec = zpack_amd.Codec(0)
    n = 15000
    rng = np.random.default_rng(77)
    sizes = rng.integers(30000, 70000, n)
    pool = [dg.fill(int(i % 4), 77, i, 70000) for i in range(48)]
    srcs = [np.ascontiguousarray(pool[i % 48][:int(sizes[i])]).copy() for i in range(n)]
    bounds = [codec.compress_bound(method, int(s)) for s in sizes]
    outs = [np.empty(b, dtype=np.uint8) for b in bounds]
    desc = np.zeros(n, dtype=zpack_amd.ENCODE_DESC)
    desc["size"] = sizes; desc["dst_capacity"] = bounds; desc["method"] = method; desc["level"] = level
    res = np.zeros(n, dtype=zpack_amd.ENCODE_RESULT)
    sp = (C.c_void_p * n)(*[a.ctypes.data for a in srcs])
    dp = (C.c_void_p * n)(*[a.ctypes.data for a in outs])
    L = codec.L
    L.zpk_codec_encode_batch_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    assert int(sizes.sum()) > 3 * (192 << 20)
    rc = L.zpk_codec_encode_batch_host(codec.h, sp, desc.ctypes.data, n, dp, res.ctypes.data)
    assert rc == 0 and (res["status"] == 0).all(), (rc, L.zpk_codec_last_error(codec.h))
    hashes = {}
    for i in range(n):
        key = (i % 48, int(sizes[i]))
        if key not in hashes:
            hashes[key] = dg.xxh3(srcs[i])
        assert int(res["hash"][i]) == hashes[key], i
    # every payload back through the decoder (one image, entries back to back)
    cs = res["comp_size"].astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(cs)])
    arc = np.empty(int(offs[-1]) + 64, dtype=np.uint8)
    for i in range(n):
        arc[offs[i]:offs[i + 1]] = outs[i][:cs[i]]
    d = np.zeros(n, dtype=zpack_amd.DECODE_DESC)
    d["src_offset"] = offs[:-1]; d["comp_size"] = cs; d["uncomp_size"] = sizes; d["expect_hash"] = res["hash"]; d["dst_capacity"] = sizes; d["method"] = method
    r2, back = codec.decode_batch_host(arc, d)
    assert (r2["status"] == 0).all(), r2[r2["status"] != 0][:3]
    for i in range(0, n, 7):
        assert np.array_equal(back[i][:int(sizes[i])], srcs[i]), i
    codec.close()


def _stream_roundtrip(Z, tmp_path, plain, method, level, in_chunks, read_in, read_out, tag, empty_update=False):
    """write `plain` through zpack_write_file_stream in chunks of the given sizes (cycled), read it back through
    zpack_read_file_stream with the given windows; -> the archive bytes"""
    w = Writer()
    path = str(tmp_path / ("s_%s.zpk" % tag))
    assert Z.lib.zpack_init_writer(C.byref(w), path.encode()) == 0
    assert Z.lib.zpack_write_header(C.byref(w)) == 0 and Z.lib.zpack_write_data_header(C.byref(w)) == 0
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    opts = CompressOptions(method, level)
    out_size = Z.lib.zpack_get_cstream_out_size(method)
    out_buf = (C.c_uint8 * out_size)()
    st.next_out = C.cast(out_buf, u8p); st.avail_out = out_size
    Z.lib.zpack_reset_stream(C.byref(st))
    size = len(plain)
    base = plain.ctypes.data if size else 0
    dummy = (C.c_uint8 * 1)()
    pos = k = 0
    while pos < size:
        n = min(in_chunks[k % len(in_chunks)], size - pos); k += 1
        st.next_in = C.cast(base + pos, u8p); st.avail_in = n
        assert Z.lib.zpack_write_file_stream(C.byref(w), C.byref(opts), C.byref(st), None) == 0
        pos += n
    if size == 0:
        st.next_in = C.cast(dummy, u8p); st.avail_in = 0
        if empty_update:                                                               # an empty entry with, and without, a (0-byte) update
            assert Z.lib.zpack_write_file_stream(C.byref(w), C.byref(opts), C.byref(st), None) == 0
    assert Z.lib.zpack_write_file_stream_end(C.byref(w), b"e", C.byref(opts), C.byref(st), None) == 0
    Z.lib.zpack_close_stream(C.byref(st))
    assert Z.lib.zpack_write_cdr(C.byref(w)) == 0 and Z.lib.zpack_write_eocdr(C.byref(w)) == 0
    Z.lib.zpack_close_writer(C.byref(w))
    arc = open(path, "rb").read()
    ents = zpk.parse(arc)
    assert len(ents) == 1 and ents[0]["uncomp_size"] == size and ents[0]["hash"] == dg.xxh3(plain), (tag, ents)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    if size:
        sink = np.zeros(size, dtype=np.uint8)
        rc, _, _, got = _stream_entry(Z, r, 0, read_in, read_out, sink)
        assert rc == 0 and got == size and np.array_equal(sink, plain), (tag, rc, got)
    rc, out = Z.read_file(r, 0, max(size, 1))
    assert rc == 0 and np.array_equal(np.frombuffer(out, dtype=np.uint8, count=size), plain), (tag, rc)
    Z.lib.zpack_close_reader(C.byref(r))
    return arc, ents[0]


def test_stream_roundtrips_random_sizes_and_windows(Z, tmp_path):
    """Randomized: entries of 0 bytes ... 6 MiB (around the streaming writer's 512 KiB pieces, its 4 MiB steps and the reader's 256 KiB
    gather buffer on purpose), every method, written in chunks of 1 byte ... 700 KiB (mixed within one entry) and read back through
    windows of 16 bytes ... 1 MiB: every byte, the XXH3, and the checker's decode of the written entry."""
    rng = np.random.default_rng(int(os.environ.get("ZPK_STREAM_FUZZ_SEED", "20261004")))
    o = oracle()
    edges = [0, 0, 0, 0, 1, 240, 241, (512 << 10) - 1, 512 << 10, (512 << 10) + 1, (1 << 20) + 3, (4 << 20), (4 << 20) + 1, (4 << 20) + (512 << 10) + 1]
    for it in range(int(os.environ.get("ZPK_STREAM_FUZZ_ITERS", "30"))):               # (a longer soak: tools/evidence.sh e)
        size = edges[it] if it < len(edges) else int(rng.integers(1, 6 << 20))
        method, level = [(METHOD_LZ4, 0), (METHOD_ZSTD, 1), (METHOD_ZSTD, 3), (METHOD_NONE, 0), (METHOD_LZ4, 9)][it % 5]
        plain = dg.fill(int(rng.integers(0, 4)), 99, it, size)
        style = it % 4
        in_chunks = ([int(rng.integers(1, 700 << 10)) for _ in range(7)] if style == 0 else [131072] if style == 1
                     else [int(rng.integers(1, 5000)), int(rng.integers(200 << 10, 700 << 10))] if style == 2 else [512 << 10, 1, (512 << 10) - 1])
        if size > (2 << 20) and min(in_chunks) < 64:
            in_chunks = [c if c >= 64 else 4096 for c in in_chunks]                       # (keeps the call count of large entries in bounds)
        read_in = [16, 1000, 131075, 1 << 20][int(rng.integers(0, 4))] if size < (1 << 20) else [131075, 1 << 20, 70000][int(rng.integers(0, 3))]
        read_out = [64, 4096, 1 << 20][int(rng.integers(0, 3))] if size < (1 << 18) else [65536, 1 << 20][int(rng.integers(0, 2))]
        arc, e = _stream_roundtrip(Z, tmp_path, plain, method, level, in_chunks, read_in, read_out, "r%d" % it, empty_update=bool(it & 1))
        rc, out, got, h = o.entry_decode(arc, e["offset"], e["comp_size"], size, e["hash"], method, max(size, 1))
        assert rc == 0 and np.array_equal(np.frombuffer(out, dtype=np.uint8, count=size), plain), (it, size, method, rc)


def _count_frames(payload, method, blocks=None):
    """frames in an entry's payload, walked by their own block headers (LZ4F: 64 KiB blocks, no checksums; Zstandard: frames without
    checksum or dictionary — the flavours the device encoder writes); blocks (a list): gets the number of blocks of every frame"""
    p = n = 0
    b = payload
    while p < len(b):
        nb = 0
        if method == METHOD_LZ4:
            assert b[p:p + 4] == bytes([0x04, 0x22, 0x4D, 0x18]), p
            p += 15 if b[p + 4] & 0x08 else 7
            while True:
                w = int.from_bytes(b[p:p + 4], "little"); p += 4
                if w == 0:
                    break
                p += w & 0x7FFFFFFF; nb += 1
        else:
            assert b[p:p + 4] == bytes([0x28, 0xB5, 0x2F, 0xFD]), p
            fhd = b[p + 4]
            assert not fhd & 0x0F
            single, fcs = (fhd >> 5) & 1, fhd >> 6
            p += 5 + (0 if single else 1) + ([1, 2, 4, 8][fcs] if (single or fcs) else 0)
            while True:
                w = int.from_bytes(b[p:p + 3], "little"); p += 3
                p += 1 if (w >> 1) & 3 == 1 else w >> 3
                nb += 1
                if w & 1:
                    break
        n += 1
        if blocks is not None:
            blocks.append(nb)
    assert p == len(b)
    return n


def test_large_entries_are_written_as_one_frame_in_pieces(Z):
    """zpack_write_files: an entry of >= 2 MiB goes to the device as 512 KiB pieces, one wave each, and comes out as ONE frame — the layout
    of the reference writer (lib/zpack_write.c:179, :204-210; round 4 wrote a frame per piece) — its XXH3 computed by the whole chip
    (per-block partial sums + one chain): sizes around the switch, around the piece and around XXH3's 1 KiB block / 64-byte stripe
    edges, every method; checked by the oracle, the compiled reference and this library's own reader."""
    M = 1 << 20
    cases = [(METHOD_LZ4, 0, 2 * M - 1), (METHOD_LZ4, 0, 2 * M), (METHOD_LZ4, 0, 3 * M + 17), (METHOD_LZ4, 9, 2 * M + 1),
             (METHOD_ZSTD, 3, 2 * M), (METHOD_ZSTD, 1, 5 * M - 1), (METHOD_ZSTD, 1, 2 * M + (512 << 10) + 1), (METHOD_ZSTD, 3, 100000),
             (METHOD_NONE, 0, 4 * M), (METHOD_NONE, 0, 2 * M + 1023), (METHOD_NONE, 0, 2 * M + 1024), (METHOD_NONE, 0, 2 * M + 1025),
             (METHOD_NONE, 0, 2 * M + 63), (METHOD_NONE, 0, 2 * M + 64), (METHOD_NONE, 0, 2 * M + 65), (METHOD_NONE, 0, 2 * M + 1),
             (METHOD_LZ4, 0, 70000), (METHOD_ZSTD, 1, 9 * M + 12345)]
    want, files = [], []
    for i, (m, lv, n) in enumerate(cases):
        want.append(("big%02d" % i, dg.fill([dg.TEXT, dg.RECORDS, dg.RANDOM, dg.RUNS][i % 4], 123, i, n).tobytes()))
    # one zpack_write_files call with per-file options
    w = Writer()
    assert Z.lib.zpack_init_writer_heap(C.byref(w), 0) == 0
    assert Z.lib.zpack_write_header(C.byref(w)) == 0 and Z.lib.zpack_write_data_header(C.byref(w)) == 0
    opts = [CompressOptions(m, lv) for m, lv, _ in cases]
    keep = [np.frombuffer(d, dtype=np.uint8) for _, d in want]
    fl = (File * len(cases))()
    for i, (name, d) in enumerate(want):
        fl[i].filename = name.encode(); fl[i].buffer = C.cast(keep[i].ctypes.data, u8p); fl[i].size = len(d)
        fl[i].options = C.pointer(opts[i]); fl[i].cctx = None
    assert Z.lib.zpack_write_files(C.byref(w), fl, len(cases)) == 0
    assert Z.lib.zpack_write_cdr(C.byref(w)) == 0 and Z.lib.zpack_write_eocdr(C.byref(w)) == 0
    arc = bytes(C.cast(w.buffer, C.POINTER(C.c_uint8 * w.file_size)).contents)
    Z.lib.zpack_close_writer(C.byref(w))
    _decode_all_with_checkers(arc, want)
    ents = zpk.parse(arc)
    for e, (m, lv, n) in zip(ents, cases):
        assert e["method"] == m
        if m != METHOD_NONE:
            nblocks = []
            frames = _count_frames(arc[e["offset"]:e["offset"] + e["comp_size"]], m, nblocks)
            assert frames == 1, (m, n, frames)
            # 64 KiB blocks; a Zstandard frame written in pieces is closed by an empty last block
            assert nblocks[0] == (n + 65535) // 65536 + (1 if (m == METHOD_ZSTD and n >= 2 * M) else 0), (m, n, nblocks)
        else:
            assert e["comp_size"] == n
    rc, r, keepr = Z.open_memory(arc)
    assert rc == 0 and r.file_count == len(want)
    for i, (name, d) in enumerate(want):
        rc, out = Z.read_file(r, i, max(1, len(d)))
        assert rc == 0 and out[:len(d)] == d, (name, rc)
    Z.close_reader(r)
