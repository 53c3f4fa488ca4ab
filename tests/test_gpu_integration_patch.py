"""INTEGRATION.md section B, demonstrated: the REFERENCE's own lib/*.c with its per-entry codec calls replaced by this repository's
C-ABI (oracle/integration/patch_reference.py applies the patch to a scratch copy in the build container; the result,
oracle/_ref/libzpack_patched.so, links libzpk_codec.so).  The reference's three test flows (tests/open_archive.c, read_archive.c,
write_archive.c) run through it: container code = the reference's, codec = the GPU's."""
import ctypes as C
import os

import numpy as np
import pytest

from benchdata import datagen as dg
from tests import zpk
from tests._libs import ROOT, ZPackAPI, Reader, Stream, u8p, oracle, have_ref, ref

pytestmark = pytest.mark.gpu
PATCHED = os.path.join(ROOT, "oracle", "_ref", "libzpack_patched.so")
FILES = ["file1.txt", "file2.txt"]
HASHES = [0x7874cba47d02b07d, 0x15f25c0f24dd8e52]          # /root/reference/tests/archive.h:112-115


@pytest.fixture(scope="module")
def P():
    if not os.path.exists(PATCHED):
        pytest.skip("oracle/_ref/libzpack_patched.so was not built (needs /root/reference at build time)")
    return ZPackAPI(PATCHED)


@pytest.mark.parametrize("arc", ["archive_none.zpk", "archive_zstd.zpk", "archive_lz4.zpk"])
def test_open_and_read_archive_through_the_patched_reference(P, golden_dir, arc):
    wd = os.path.join(golden_dir, "ref_workdir")
    raw = open(os.path.join(wd, arc), "rb").read()
    for how in ("file", "memory"):
        r = Reader()
        if how == "file":
            rc = P.lib.zpack_init_reader(C.byref(r), os.path.join(wd, arc).encode())
        else:
            keep = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
            rc = P.lib.zpack_init_reader_memory_shared(C.byref(r), C.cast(keep, u8p), len(raw))
        assert rc == 0 and r.file_count == 2
        for i, e in enumerate(P.entries(r)):
            plain = open(os.path.join(wd, FILES[i]), "rb").read()
            assert e["filename"] == FILES[i] and e["hash"] == HASHES[i]
            rc, out = P.read_file(r, i, 350)                       # tests/read_archive.c:21-35, its 350-byte buffer
            assert rc == 0 and out[:len(plain)] == plain, (arc, how, i, rc)
        # a wrong hash is still the reference's verdict, produced by the device
        r.file_entries[0].hash ^= 1
        rc, out = P.read_file(r, 0, 350)
        assert rc == 15
        r.file_entries[0].hash ^= 1
        # the streaming reader of the patched library is the reference's own, untouched code (16-byte input window)
        st = Stream()
        assert P.lib.zpack_init_stream(C.byref(st)) == 0
        in_buf = (C.c_uint8 * 16)(); ob = (C.c_uint8 * 350)()
        e = r.file_entries[1]
        plain = open(os.path.join(wd, FILES[1]), "rb").read()
        P.lib.zpack_reset_stream(C.byref(st))
        st.next_out = C.cast(ob, u8p); st.avail_out = 350
        for _ in range(10000):
            if st.read_back:
                C.memmove(in_buf, C.string_at(C.addressof(st.next_in.contents) - st.read_back, st.read_back), st.read_back)
            st.next_in = C.cast(in_buf, u8p); st.avail_in = 16
            assert P.lib.zpack_read_file_stream(C.byref(r), C.byref(e), C.byref(st), None) == 0
            if st.total_in == e.comp_size and st.read_back == 0:
                break
        assert bytes(ob[:len(plain)]) == plain
        P.lib.zpack_close_stream(C.byref(st))
        P.close_reader(r)
        assert bytes(r) == bytes(C.sizeof(Reader))


@pytest.mark.parametrize("method,level", [(1, 3), (2, 1), (0, 0)])               # tests/write_archive.c:31-41
def test_write_archive_through_the_patched_reference(P, golden_dir, method, level):
    wd = os.path.join(golden_dir, "ref_workdir")
    want = [(n, open(os.path.join(wd, n), "rb").read()) for n in FILES]
    want += [("big%d" % i, dg.fill(i % 4, 71, i, 100000 + 7777 * i).tobytes()) for i in range(6)]
    arc = P.write_archive(want, method, level)                   # the reference's zpack_write_archive loop, codec on the GPU
    o = oracle()
    ents = zpk.parse(arc)
    assert [e["filename"] for e in ents] == [n for n, _ in want]
    for e, (name, data) in zip(ents, want):
        assert e["uncomp_size"] == len(data) and e["hash"] == dg.xxh3(data), name       # the codec's XXH3 went into the CDR
        rc, out, got, h = o.entry_decode(arc, e["offset"], e["comp_size"], e["uncomp_size"], e["hash"], e["method"], len(data))
        assert rc == 0 and out == data, name
    if have_ref():                                               # and the UNPATCHED reference (stock liblz4 / libzstd) reads it back
        R = ref()
        rc, r, keep = R.open_memory(arc)
        assert rc == 0
        for i, (name, data) in enumerate(want):
            rc, got = R.read_file(r, i, len(data))
            assert rc == 0 and got == data, name
        R.close_reader(r)
    rc, r, keep = P.open_memory(arc)                             # and the patched one itself
    assert rc == 0
    for i, (name, data) in enumerate(want):
        rc, got = P.read_file(r, i, len(data))
        assert rc == 0 and got == data, name
    P.close_reader(r)
