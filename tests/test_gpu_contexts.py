"""Contexts, threads and the entry points the reference documents but its own tests never call.

  * zpack_create_dctx / zpack_free_dctx: N threads, each with its OWN context, on ONE buffer-backed reader — the
    documented-safe pattern of the reference (lib/zpack.h:337-340).
  * dctx == NULL: every reader creates its own context on first use (lib/zpack_read.c:17-31), so two readers on two
    threads are independent; stored entries from several threads with dctx == NULL (the reference needs no context
    for them at all) are serialised by the codec and still come out right.
  * zpack_reset_reader_dctx after an abandoned stream (lib/zpack_read.c:679-690).
  * one zpack_stream reused across readers that are closed in between (the stream must not keep a dead context).
  * zpack_write_files_from_archive (lib/zpack_write.c:345-428): raw entry copy, checked by the oracle AND the
    compiled reference.
  * explicit cctx for zpack_write_files / the streaming writer.
  * a context over several codecs (ZPACK_AMD_DEVICES, rehearsed as "0,0" on a one-GPU box): the batch calls shard
    their entries statically, results are those of the one-codec call.
  * the host batch path on sparse picks out of a large archive and with generous max_size values.
  * entries of 64 MiB and 512 MiB against the reference-made recipes (tests/golden/recipes_big.json).
"""
import ctypes as C
import json
import os
import threading

import numpy as np
import pytest

import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import (ZPackAPI, Reader, Writer, Stream, File, CompressOptions, FileEntry, u8p, oracle, have_ref, ref,
                         METHOD_NONE, METHOD_ZSTD, METHOD_LZ4)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Z():
    z = ZPackAPI(zpack_amd.ZPACK_SO)
    L = z.lib
    for fn in (L.zpack_create_dctx, L.zpack_create_cctx):
        fn.restype = C.c_void_p
        fn.argtypes = [C.c_int]
    for fn in (L.zpack_free_dctx, L.zpack_free_cctx):
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_void_p]
    L.zpack_reset_reader_dctx.restype = None
    L.zpack_reset_reader_dctx.argtypes = [C.POINTER(Reader)]
    L.zpack_write_files_from_archive.argtypes = [C.POINTER(Writer), C.POINTER(Reader), C.POINTER(FileEntry), C.c_uint64]
    L.zpack_read_files.argtypes = [C.POINTER(Reader), C.POINTER(C.POINTER(FileEntry)), C.c_uint64, C.POINTER(u8p),
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.c_void_p]
    return z


def _mixed_files(n, seed, lo=1, hi=90000):
    rng = np.random.default_rng(seed)
    sizes = [int(x) for x in rng.integers(lo, hi, n)]
    return [("s%d_m%04d" % (seed, i), dg.fill(i % 4, seed, i, s).tobytes()) for i, s in enumerate(sizes)]


def _archive(Z, files, method, level):
    return Z.write_archive(files, method, level)


def _read_all(Z, r, files, dctx, order=None, errors=None, tag=""):
    idx = list(range(len(files))) if order is None else order
    for i in idx:
        name, data = files[i]
        out = (C.c_uint8 * max(1, len(data)))()
        rc = Z.lib.zpack_read_file(C.byref(r), C.byref(r.file_entries[i]), C.cast(out, u8p), len(data), dctx)
        if rc != 0 or bytes(out[:len(data)]) != data:
            if errors is not None:
                errors.append((tag, name, rc))
            else:
                raise AssertionError((tag, name, rc))


@pytest.mark.parametrize("method,level", [(METHOD_LZ4, 0), (METHOD_ZSTD, 3), (METHOD_NONE, 0)])
def test_threads_with_own_dctx_on_one_reader(Z, method, level):
    """lib/zpack.h:337-340: a buffer-backed reader is thread-safe when every thread passes its own dctx."""
    files = _mixed_files(48, 7)
    arc = _archive(Z, files, method, level)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    nthreads = 4
    ctxs = [Z.lib.zpack_create_dctx(method) for _ in range(nthreads)]
    assert all(ctxs), "zpack_create_dctx returned NULL on a GPU box"
    errors = []
    rng = np.random.default_rng(1)
    ths = [threading.Thread(target=_read_all, args=(Z, r, files, C.c_void_p(ctxs[t]), [int(x) for x in rng.permutation(len(files))],
                                                    errors, "t%d" % t)) for t in range(nthreads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors[:5]
    assert not r.zstd_dctx, "explicit contexts were passed: the reader must not have created its own"
    for c in ctxs:
        Z.lib.zpack_free_dctx(method, c)
    Z.close_reader(r)


def test_two_readers_two_threads_null_dctx(Z):
    """dctx == NULL: each reader lazily creates ITS OWN context (lib/zpack_read.c:17-31) — nothing is shared."""
    fa, fb = _mixed_files(40, 11), _mixed_files(40, 12)
    arcs = [_archive(Z, fa, METHOD_LZ4, 0), _archive(Z, fb, METHOD_ZSTD, 1)]
    readers = []
    for a in arcs:
        rc, r, keep = Z.open_memory(a)
        assert rc == 0
        readers.append((r, keep))
    errors = []
    ths = [threading.Thread(target=_read_all, args=(Z, readers[k][0], (fa, fb)[k], None, None, errors, "r%d" % k)) for k in range(2)]
    for _ in range(2):                                    # twice: the second pass runs on already-created contexts
        ths = [threading.Thread(target=_read_all, args=(Z, readers[k][0], (fa, fb)[k], None, None, errors, "r%d" % k)) for k in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
    assert not errors, errors[:5]
    c0, c1 = readers[0][0].zstd_dctx, readers[1][0].zstd_dctx
    assert c0 and c1 and c0 != c1, "every reader owns its context"
    for r, _ in readers:
        Z.close_reader(r)
        assert bytes(r) == bytes(C.sizeof(Reader))


def test_stored_entries_many_threads_null_dctx(Z):
    """Method NONE needs no context in the reference, so N threads with dctx == NULL are safe there; here they share the
    reader's one context, which serialises them — the bytes must still be right."""
    files = _mixed_files(64, 13, 1, 20000)
    arc = _archive(Z, files, METHOD_NONE, 0)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    errors = []
    ths = [threading.Thread(target=_read_all, args=(Z, r, files, None, None, errors, "t%d" % t)) for t in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors[:5]
    Z.close_reader(r)


def _stream_read(Z, r, i, st, dctx, in_size, out_size, stop_after=None):
    """the streaming loop of tests/read_archive.c:38-82; returns the bytes received (None when abandoned)"""
    e = r.file_entries[i]
    in_buf = (C.c_uint8 * in_size)()
    out = bytearray()
    ob = (C.c_uint8 * out_size)()
    Z.lib.zpack_reset_stream(C.byref(st))
    for passes in range(1 << 20):
        if st.read_back:
            tail = C.string_at(C.addressof(st.next_in.contents) - st.read_back, st.read_back)
            C.memmove(in_buf, tail, st.read_back)
        st.next_in = C.cast(in_buf, u8p)
        st.avail_in = in_size
        st.next_out = C.cast(ob, u8p)
        st.avail_out = out_size
        rc = Z.lib.zpack_read_file_stream(C.byref(r), C.byref(e), C.byref(st), dctx)
        assert rc == 0, (i, passes, rc)
        out += bytes(ob[:out_size - st.avail_out])
        if stop_after is not None and passes >= stop_after:
            return None
        if st.total_in == e.comp_size and st.read_back == 0:
            return bytes(out)
    raise AssertionError("stream never finished")


@pytest.mark.parametrize("method,level", [(METHOD_ZSTD, 3), (METHOD_LZ4, 0)])
def test_reset_reader_dctx_after_abandoned_stream(Z, method, level):
    files = _mixed_files(6, 17, 30000, 90000)
    arc = _archive(Z, files, method, level)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    assert _stream_read(Z, r, 0, st, None, 4096, 8192, stop_after=2) is None          # walk away in the middle of entry 0
    Z.lib.zpack_reset_reader_dctx(C.byref(r))                                             # lib/zpack_read.c:679-690
    _read_all(Z, r, files, None)                                                          # one-shot reads are unaffected
    for i in (3, 0):                                                                      # and so is a fresh stream, also of entry 0
        assert _stream_read(Z, r, i, st, None, 4096, 8192) == files[i][1]
    # the same with an explicit context that is reset by freeing / recreating it
    ctx = Z.lib.zpack_create_dctx(method)
    assert _stream_read(Z, r, 1, st, C.c_void_p(ctx), 1000, 3000, stop_after=1) is None
    Z.lib.zpack_free_dctx(method, ctx)
    ctx = Z.lib.zpack_create_dctx(method)
    assert _stream_read(Z, r, 1, st, C.c_void_p(ctx), 1000, 3000) == files[1][1]
    Z.lib.zpack_free_dctx(method, ctx)
    Z.lib.zpack_close_stream(C.byref(st))
    Z.close_reader(r)


def test_stream_reused_across_closed_readers(Z):
    """A zpack_stream belongs to the caller and outlives readers: it must decode with the context of the CURRENT call,
    never with one that died with an earlier reader."""
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    for round_ in range(3):
        files = _mixed_files(3, 20 + round_, 5000, 40000)
        arc = _archive(Z, files, (METHOD_LZ4, METHOD_ZSTD, METHOD_NONE)[round_], 1)
        rc, r, keep = Z.open_memory(arc)
        assert rc == 0
        for i in range(3):
            assert _stream_read(Z, r, i, st, None, 777, 5000) == files[i][1]
        Z.close_reader(r)                                                                 # frees the reader's context
    Z.lib.zpack_close_stream(C.byref(st))


def _heap_archive(w):
    return bytes(C.cast(w.buffer, C.POINTER(C.c_uint8 * w.file_size)).contents)


@pytest.mark.parametrize("backing", ["memory", "file"])
def test_write_files_from_archive_roundtrip(Z, tmp_path, backing):
    """lib/zpack_write.c:345-428: raw copy of still-compressed entries from two source archives (different methods) into
    a third; the result must read back bit-exactly through the oracle, the compiled reference and this library."""
    fa, fb = _mixed_files(9, 31), _mixed_files(7, 32)
    arc_a, arc_b = _archive(Z, fa, METHOD_ZSTD, 3), _archive(Z, fb, METHOD_LZ4, 0)
    w = Writer()
    assert Z.lib.zpack_init_writer_heap(C.byref(w), 0) == 0
    assert Z.lib.zpack_write_header(C.byref(w)) == 0 and Z.lib.zpack_write_data_header(C.byref(w)) == 0
    want = []
    keepalive = []
    for k, (arc, files) in enumerate(((arc_a, fa), (arc_b, fb))):
        r = Reader()
        if backing == "file":
            p = str(tmp_path / ("src%d.zpk" % k))
            open(p, "wb").write(arc)
            assert Z.lib.zpack_init_reader(C.byref(r), p.encode()) == 0
        else:
            buf = (C.c_uint8 * len(arc)).from_buffer_copy(arc)
            keepalive.append(buf)
            assert Z.lib.zpack_init_reader_memory_shared(C.byref(r), C.cast(buf, u8p), len(arc)) == 0
        # a contiguous run of the table (entries 2..n-2), as `zpack` does when it repacks
        first, cnt = 2, r.file_count - 3
        rc = Z.lib.zpack_write_files_from_archive(C.byref(w), C.byref(r), C.byref(r.file_entries[first]), cnt)
        assert rc == 0, rc
        want += files[first:first + cnt]
        Z.close_reader(r)
    # plus one freshly compressed file behind the copied ones
    extra = ("fresh", dg.fill(dg.TEXT, 33, 0, 50000).tobytes())
    opts = CompressOptions(METHOD_ZSTD, 1)
    b = (C.c_uint8 * len(extra[1])).from_buffer_copy(extra[1])
    f = (File * 1)()
    f[0].filename = b"fresh"; f[0].buffer = C.cast(b, u8p); f[0].size = len(extra[1]); f[0].options = C.pointer(opts)
    assert Z.lib.zpack_write_files(C.byref(w), f, 1) == 0
    want.append(extra)
    assert Z.lib.zpack_write_cdr(C.byref(w)) == 0 and Z.lib.zpack_write_eocdr(C.byref(w)) == 0
    out = _heap_archive(w)
    Z.lib.zpack_close_writer(C.byref(w))

    ents = zpk.parse(out)
    assert [e["filename"] for e in ents] == [n for n, _ in want]
    src_ents = {e["filename"]: e for e in zpk.parse(arc_a)[2:-1] + zpk.parse(arc_b)[2:-1]}
    o = oracle()
    pos = 10
    for e, (name, data) in zip(ents, want):
        assert e["offset"] == pos, "payloads are appended back to back"
        pos += e["comp_size"]
        if name in src_ents:                                  # copied raw: same sizes, hash, method as the source entry
            s = src_ents[name]
            assert (e["comp_size"], e["uncomp_size"], e["hash"], e["method"]) == (s["comp_size"], s["uncomp_size"], s["hash"], s["method"])
        rc, got, n, h = o.entry_decode(out, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"], len(data))
        assert rc == 0 and got == data, (name, rc)
    if have_ref():
        R = ref()
        rc, r, keep = R.open_memory(out)
        assert rc == 0
        for i, (name, data) in enumerate(want):
            rc, got = R.read_file(r, i, len(data))
            assert rc == 0 and got == data, ("reference rejects the repacked archive", name, rc)
        R.close_reader(r)
    rc, r, keep = Z.open_memory(out)
    assert rc == 0
    _read_all(Z, r, want, None)
    Z.close_reader(r)


def test_write_files_from_archive_bad_offset(Z):
    files = _mixed_files(3, 35)
    arc = _archive(Z, files, METHOD_LZ4, 0)
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    r.file_entries[1].offset = len(arc) + 5
    w = Writer()
    assert Z.lib.zpack_init_writer_heap(C.byref(w), 0) == 0
    assert Z.lib.zpack_write_files_from_archive(C.byref(w), C.byref(r), r.file_entries, 3) == 16   # FILE_OFFSET_INVALID (lib/zpack_write.c:372)
    assert w.file_count == 1                                   # the entry before the bad one was appended, like the reference loop
    Z.lib.zpack_close_writer(C.byref(w))
    Z.close_reader(r)


@pytest.mark.parametrize("method,level", [(METHOD_ZSTD, 1), (METHOD_LZ4, 0)])
def test_explicit_cctx_oneshot_and_stream(Z, method, level):
    files = _mixed_files(12, 41)
    cctx = Z.lib.zpack_create_cctx(method)
    assert cctx
    w = Writer()
    assert Z.lib.zpack_init_writer_heap(C.byref(w), 0) == 0
    opts = CompressOptions(method, level)
    arr = (File * len(files))()
    keep = []
    for i, (n, d) in enumerate(files):
        b = (C.c_uint8 * len(d)).from_buffer_copy(d)
        keep.append(b)
        arr[i].filename = n.encode(); arr[i].buffer = C.cast(b, u8p); arr[i].size = len(d); arr[i].options = C.pointer(opts)
        arr[i].cctx = cctx
    assert Z.lib.zpack_write_header(C.byref(w)) == 0 and Z.lib.zpack_write_data_header(C.byref(w)) == 0
    assert Z.lib.zpack_write_files(C.byref(w), arr, len(files)) == 0
    assert not w.zstd_cctx, "an explicit cctx was passed for every file: the writer must not create its own"
    # one more entry through the streaming writer with the same explicit context
    st = Stream()
    assert Z.lib.zpack_init_stream(C.byref(st)) == 0
    osz = Z.lib.zpack_get_cstream_out_size(method)
    ob = (C.c_uint8 * osz)()
    st.next_out = C.cast(ob, u8p); st.avail_out = osz
    tail = dg.fill(dg.RECORDS, 42, 0, 70001).tobytes()
    tb = (C.c_uint8 * len(tail)).from_buffer_copy(tail)
    Z.lib.zpack_reset_stream(C.byref(st))
    st.next_in = C.cast(tb, u8p)
    while st.total_in < len(tail):
        st.avail_in = min(5000, len(tail) - st.total_in)
        assert Z.lib.zpack_write_file_stream(C.byref(w), C.byref(opts), C.byref(st), cctx) == 0
    assert Z.lib.zpack_write_file_stream_end(C.byref(w), b"tail", C.byref(opts), C.byref(st), cctx) == 0
    Z.lib.zpack_close_stream(C.byref(st))
    assert Z.lib.zpack_write_cdr(C.byref(w)) == 0 and Z.lib.zpack_write_eocdr(C.byref(w)) == 0
    out = _heap_archive(w)
    Z.lib.zpack_close_writer(C.byref(w))
    Z.lib.zpack_free_cctx(method, cctx)
    want = files + [("tail", tail)]
    o = oracle()
    for e, (name, data) in zip(zpk.parse(out), want):
        rc, got, n, h = o.entry_decode(out, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"], len(data))
        assert rc == 0 and got == data and e["hash"] == dg.xxh3(data), name
    if have_ref():
        R = ref()
        rc, r, k = R.open_memory(out)
        for i, (name, data) in enumerate(want):
            rc, got = R.read_file(r, i, len(data))
            assert rc == 0 and got == data, name
        R.close_reader(r)


def _read_files(Z, r, idx, caps, dctx):
    n = len(idx)
    ptrs = (C.POINTER(FileEntry) * n)(*[C.pointer(r.file_entries[i]) for i in idx])
    outs = [(C.c_uint8 * max(1, c))() for c in caps]
    bufs = (u8p * n)(*[C.cast(o, u8p) for o in outs])
    capv = (C.c_size_t * n)(*caps)
    results = (C.c_int * n)()
    rc = Z.lib.zpack_read_files(C.byref(r), ptrs, n, bufs, capv, results, dctx)
    return rc, list(results), outs


def test_context_over_several_codecs_shards_the_batch(Z):
    """ZPACK_AMD_DEVICES: a context that spans several devices splits batch calls into contiguous ranges balanced by bytes,
    one host thread + one codec per device, no exchange.  Rehearsed on a one-GPU box with the same device
    listed three times; the 8-GPU form is ZPACK_AMD_DEVICES=all."""
    files = _mixed_files(150, 51, 1, 300000)
    old = os.environ.get("ZPACK_AMD_DEVICES")
    os.environ["ZPACK_AMD_DEVICES"] = "0,0,0"
    try:
        for method, level in ((METHOD_LZ4, 0), (METHOD_ZSTD, 3)):
            arc = _archive(Z, files, method, level)              # the writer's own context spans three codecs: write shards too
            o = oracle()
            ents = zpk.parse(arc)
            pos = 10
            for e, (name, data) in zip(ents, files):
                assert e["offset"] == pos
                pos += e["comp_size"]
                rc, got, n, h = o.entry_decode(arc, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"], len(data))
                assert rc == 0 and got == data, name
            rc, r, keep = Z.open_memory(arc)
            assert rc == 0
            r.file_entries[77].hash ^= 1                          # one bad entry in the middle shard
            rc, results, outs = _read_files(Z, r, list(range(len(files))), [len(d) for _, d in files], None)
            assert rc == 0
            for i, (name, data) in enumerate(files):
                assert results[i] == (15 if i == 77 else 0), (name, results[i])
                assert bytes(outs[i][:len(data)]) == data
            Z.close_reader(r)
    finally:
        if old is None:
            del os.environ["ZPACK_AMD_DEVICES"]
        else:
            os.environ["ZPACK_AMD_DEVICES"] = old


def test_host_batch_sparse_picks_and_generous_buffers(Z):
    """zpack_read_files on two small entries at opposite ends of a large archive must not need the whole archive on the
    device, and max_size values far above uncomp_size must not multiply the transfers."""
    files = _mixed_files(400, 61, 20000, 120000)
    arc = _archive(Z, files, METHOD_LZ4, 0)
    assert len(arc) > 8 << 20
    rc, r, keep = Z.open_memory(arc)
    assert rc == 0
    idx = [0, len(files) - 1, 200]
    caps = [len(files[i][1]) + (64 << 20) for i in idx]           # 64 MiB of slack per entry
    rc, results, outs = _read_files(Z, r, idx, caps, None)
    assert rc == 0 and results == [0, 0, 0]
    for k, i in enumerate(idx):
        assert bytes(outs[k][:len(files[i][1])]) == files[i][1]
    # guard order is unchanged in the gathered form: a too-small buffer, a bad offset, a good entry
    r.file_entries[5].offset = len(arc)
    rc, results, outs = _read_files(Z, r, [3, 5, 399], [10, len(files[5][1]), len(files[399][1])], None)
    assert rc == 0 and results == [12, 16, 0], results
    Z.close_reader(r)


def test_large_entries_match_reference_recipes(golden_dir):
    """Entries of 64 MiB and 512 MiB: same verdict, size and XXH3 as the reference produced for the same frames
    (tests/golden/recipes_big.json, made by tests/golden/make_golden_big.py with the compiled reference).  One wave
    decodes an entry, so these run for seconds — the watchdog budget is proportional to the entry size."""
    import torch
    recs = json.load(open(os.path.join(golden_dir, "recipes_big.json")))
    codec = zpack_amd.Codec(0)
    dev = torch.device("cuda:0")
    for r in recs:
        plain = dg.fill(r["cls"], r["seed"], r["index"], r["size"])
        frame = np.frombuffer(dg.compress(r["method"], r["level"], plain), dtype=np.uint8)
        assert len(frame) == r["comp_size"] and dg.xxh3(frame) == r["frame_xxh3"], r["label"]
        src = torch.zeros(10 + len(frame) + 1, dtype=torch.uint8, device=dev)
        src[10:10 + len(frame)] = torch.from_numpy(frame.copy()).to(dev)
        d = np.zeros(1, dtype=zpack_amd.DECODE_DESC)
        d[0]["src_offset"] = 10; d[0]["comp_size"] = len(frame); d[0]["uncomp_size"] = r["size"]
        d[0]["expect_hash"] = r["hash"]; d[0]["dst_capacity"] = r["size"]; d[0]["method"] = r["method"]
        dst = torch.zeros(r["size"], dtype=torch.uint8, device=dev)
        ddesc = torch.from_numpy(d.view(np.uint8)).to(dev)
        dres = torch.zeros(zpack_amd.DECODE_RESULT.itemsize, dtype=torch.uint8, device=dev)
        codec.decode_batch_device(src, ddesc, 1, dst, dres)
        torch.cuda.synchronize()
        res = dres.cpu().numpy().view(zpack_amd.DECODE_RESULT)[0]
        assert int(res["status"]) == 0, (r["label"], res)
        assert int(res["produced"]) == r["size"] and int(res["hash"]) == r["hash"], (r["label"], res)
        assert torch.equal(dst.cpu(), torch.from_numpy(plain)), r["label"]
        del src, dst
    codec.close()
