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


ENTRY_DT = None


def _np_table(r):
    """the reader's entry table as a numpy view (48-byte zpack_file_entry records) + the names"""
    import numpy as np
    dt = np.dtype([("name", "<u8"), ("offset", "<u8"), ("comp", "<u8"), ("uncomp", "<u8"), ("hash", "<u8"), ("method", "u1"), ("pad", "V7")])
    assert dt.itemsize == 48
    n = int(r.file_count)
    a = np.ctypeslib.as_array(C.cast(r.file_entries, C.POINTER(C.c_uint8)), shape=(n * 48,)).view(dt)
    return a


def test_cdr_one_million_entries_and_name_lookup():
    """SURVEY.md §8f rank 1 at the size it names: 1 M entries.  Same table as the compiled reference (numbers compared as arrays, every
    name compared), open/close times printed; zpack_get_file_entry answers like the reference's linear scan (lib/zpack_read.c:760-769),
    first occurrence of a duplicate name included, through the reader-owned name index."""
    import numpy as np
    n = 1000000
    names = [("d%03d/f%07d" % (i % 311, i)).encode() for i in range(n)]
    names[777777] = names[123]                                  # a duplicate: the first occurrence must win
    nl = np.fromiter((len(x) for x in names), dtype=np.uint16, count=n)
    comp = (np.arange(n, dtype=np.uint64) * 7919) % 64
    offs = 10 + np.concatenate([[0], np.cumsum(comp[:-1])]).astype(np.uint64)
    body = bytearray()
    for i in range(n):
        body += struct.pack("<H", int(nl[i])) + names[i] + struct.pack("<QQQQB", int(offs[i]), int(comp[i]), int(comp[i]) * 2, (i * 0x9E3779B97F4A7C15) & (2**64 - 1), i % 3)
    data_len = int(offs[-1] + comp[-1]) - 10
    arc = bytearray(b"ZPK\x15" + struct.pack("<H", 1) + b"ZPK\x14") + bytes(data_len)
    cdr_off = len(arc)
    arc += b"ZPK\x13" + struct.pack("<QQ", n, len(body)) + body + b"ZPK\x12" + struct.pack("<Q", cdr_off)
    arc = bytes(arc)
    mine = L.ZPackAPI(zpack_amd.ZPACK_SO)
    t0 = time.time(); rc, r, keep = mine.open_memory(arc); t_open = time.time() - t0
    assert rc == 0 and r.file_count == n
    a = _np_table(r)
    assert np.array_equal(a["offset"], offs) and np.array_equal(a["comp"], comp) and np.array_equal(a["uncomp"], comp * 2)
    assert np.array_equal(a["method"], (np.arange(n) % 3).astype(np.uint8))
    for i in range(0, n, 997):
        assert C.string_at(int(a["name"][i])) == names[i]
    fn = mine.lib.zpack_get_file_entry
    fn.restype = C.c_void_p
    fn.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
    base = C.addressof(r.file_entries.contents)
    rng = np.random.default_rng(5)
    probes = [names[int(i)] for i in rng.integers(0, n, 3000)] + [b"nope", b"d000/f9999999", names[777777]]
    t0 = time.time()
    got = [fn(p, base, n) for p in probes]
    t_lookup = time.time() - t0
    want = {}
    for i in range(n - 1, -1, -1):
        want[names[i]] = i                                       # first occurrence wins
    for p, g in zip(probes, got):
        assert (g is None) == (p not in want) and (g is None or (g - base) // 48 == want[p]), p
    # a sub-range of the table is not the registered table: the scan answers, same rule
    assert fn(names[5], base + 48 * 3, 100) == base + 48 * 5 and fn(names[2], base + 48 * 3, 100) is None
    if L.have_ref():
        ref = L.ref()
        t0 = time.time(); rc2, r2, keep2 = ref.open_memory(arc); t_ref = time.time() - t0
        assert rc2 == 0
        b = _np_table(r2)
        for f in ("offset", "comp", "uncomp", "hash", "method"):
            assert np.array_equal(a[f], b[f]), f
        for i in range(0, n, 1009):
            assert C.string_at(int(b["name"][i])) == names[i]
        rfn = ref.lib.zpack_get_file_entry
        rfn.restype = C.c_void_p
        rfn.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
        rbase = C.addressof(r2.file_entries.contents)
        for p, g in list(zip(probes, got))[::100] + list(zip(probes, got))[-3:]:
            rg = rfn(p, rbase, n)
            assert (rg is None) == (g is None) and (g is None or (rg - rbase) == (g - base)), p
        t0 = time.time(); ref.close_reader(r2); t_refc = time.time() - t0
        print("reference: open %.3f s close %.3f s" % (t_ref, t_refc))
    t0 = time.time(); mine.close_reader(r); t_close = time.time() - t0
    print("1 M entries: open %.3f s, %d lookups %.3f s, close %.3f s" % (t_open, len(probes), t_lookup, t_close))
    # a second parse on the same reader struct starts from a clean table (the arena must not leak or be double-freed)
    rc, r, keep = mine.open_memory(arc[:0] + arc)
    assert rc == 0
    assert mine.lib.zpack_read_archive_memory(C.byref(r)) == 0 and r.file_count == n
    mine.close_reader(r)


@pytest.mark.parametrize("back", [1, 4, 11, 19])
def test_cdr_offset_in_the_last_bytes_is_not_read_out_of_bounds(back):
    """the reference reads the 20-byte CDR header after checking only cdr_offset < file_size (lib/zpack_read.c:249-250);
    here a CDR that cannot hold its header is rejected without touching bytes past the buffer."""
    arc, ents = _big_archive(3)
    a = bytearray(arc)
    struct.pack_into("<Q", a, len(a) - 8, len(a) - back)
    mine = L.ZPackAPI(zpack_amd.ZPACK_SO)
    rc, r, keep = mine.open_memory(bytes(a))
    assert rc in (6, 8), rc                                      # SIGNATURE_INVALID or BLOCK_SIZE_INVALID, never a crash
    mine.close_reader(r)
