"""CDR parse at scale, CPU only: the reader of libzpack_amd.so opens a large synthetic
archive — every filename in one arena — and must produce exactly the table the compiled reference produces
(lib/zpack_read.c:109-166), for memory- and file-backed readers, and reject the same malformed CDRs."""
import ctypes as C
import os
import struct
import time

import pytest

import zpack_amd
from tests import _libs as L
from tests import zpk


def _big_archive(n):
    ents = []
    off = 10
    for i in range(n):
        name = "dir%04d/file_%07d%s" % (i % 977, i, ".bin" if i % 3 else "")
        cs = (i * 7919) % 5000
        ents.append((name, off, cs, cs * 3, (i * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF, i % 3))
        off += cs
    return zpk.assemble([b"\0" * (off - 10)], ents), ents


def _table(api, r):
    return [(r.file_entries[i].filename, r.file_entries[i].offset, r.file_entries[i].comp_size, r.file_entries[i].uncomp_size,
             r.file_entries[i].hash, r.file_entries[i].comp_method) for i in range(r.file_count)]


def test_cdr_parse_matches_reference_at_scale(tmp_path):
    n = 60000
    arc, ents = _big_archive(n)
    mine = L.ZPackAPI(zpack_amd.ZPACK_SO)
    t0 = time.time()
    rc, r, keep = mine.open_memory(arc)
    t_mine = time.time() - t0
    assert rc == 0 and r.file_count == n
    assert r.comp_size == sum(e[2] for e in ents) and r.uncomp_size == sum(e[3] for e in ents)
    got = _table(mine, r)
    assert got == [(e[0].encode(), e[1], e[2], e[3], e[4], e[5]) for e in ents]
    if L.have_ref():
        ref = L.ref()
        rc2, r2, keep2 = ref.open_memory(arc)
        assert rc2 == 0 and _table(ref, r2) == got
        ref.close_reader(r2)
    mine.close_reader(r)
    assert r.file_count == 0 and not r.file_entries           # re-zeroed, like the reference (lib/zpack_read.c:692-717)
    # file-backed reader: same table
    p = tmp_path / "big.zpk"
    p.write_bytes(arc)
    r3 = L.Reader()
    mine.lib.zpack_init_reader.argtypes = [C.c_void_p, C.c_char_p]
    assert mine.lib.zpack_init_reader(C.byref(r3), str(p).encode()) == 0
    assert _table(mine, r3) == got
    mine.close_reader(r3)
    print("parse of %d entries: %.3f s" % (n, t_mine))


@pytest.mark.parametrize("what", ["count_too_big", "name_overruns_block", "block_overruns_file"])
def test_cdr_malformed_same_verdict_as_reference(what):
    arc, ents = _big_archive(50)
    a = bytearray(arc)
    cdr = struct.unpack_from("<Q", a, len(a) - 8)[0]
    if what == "count_too_big":
        struct.pack_into("<Q", a, cdr + 4, 10 ** 9)
    elif what == "name_overruns_block":
        struct.pack_into("<H", a, cdr + 20, 60000)
    else:
        struct.pack_into("<Q", a, cdr + 12, len(a))
    mine = L.ZPackAPI(zpack_amd.ZPACK_SO)
    rc, r, keep = mine.open_memory(bytes(a))
    assert rc != 0
    if L.have_ref():
        ref = L.ref()
        rc2, r2, keep2 = ref.open_memory(bytes(a))
        assert rc2 == rc, (what, rc, rc2)
        ref.close_reader(r2)
    mine.close_reader(r)
