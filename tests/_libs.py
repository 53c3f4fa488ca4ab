"""ctypes loaders shared by the tests (and by bench.py's cpu_baseline leg / smoke()).

  oracle()   -> oracle/liboracle.so        this repo's CPU restatement (checker)
  ref()      -> oracle/_ref/libzpack_ref.so the compiled REFERENCE (present only where it was built)
  ZPackAPI   -> ctypes view of the zpack.h C API; works for the reference library and for the
                product library (zpack_amd/libzpack_amd.so) alike, because the ABI is the same.
"""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libzpack_ref.so")

u8p = C.POINTER(C.c_uint8)


def _buf(b):
    """bytes/bytearray/memoryview/numpy -> (ctypes pointer, keepalive)"""
    import numpy as np
    if isinstance(b, np.ndarray):
        a = np.ascontiguousarray(b, dtype=np.uint8)
        return a.ctypes.data_as(u8p), a
    if isinstance(b, (bytes, bytearray, memoryview)):
        a = (C.c_uint8 * max(1, len(b))).from_buffer_copy(bytes(b) if len(b) else b"\0")
        return C.cast(a, u8p), a
    raise TypeError(type(b))


class Oracle:
    def __init__(self, path=ORACLE_SO):
        self.lib = L = C.CDLL(path)
        L.orc_xxh3_64.restype = C.c_uint64
        L.orc_xxh3_64.argtypes = [u8p, C.c_size_t]
        L.orc_xxh32.restype = C.c_uint32
        L.orc_xxh32.argtypes = [u8p, C.c_size_t, C.c_uint32]
        L.orc_xxh64.restype = C.c_uint64
        L.orc_xxh64.argtypes = [u8p, C.c_size_t, C.c_uint64]
        for fn in (L.orc_lz4f_decode, L.orc_zstd_decode):
            fn.restype = C.c_int
            fn.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        for fn in (L.orc_lz4f_encode, L.orc_zstd_encode):
            fn.restype = C.c_size_t
            fn.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t]
        for fn in (L.orc_lz4f_bound, L.orc_zstd_bound):
            fn.restype = C.c_size_t
            fn.argtypes = [C.c_size_t]
        L.orc_entry_decode.restype = C.c_int
        L.orc_entry_decode.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                       u8p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_entry_encode.restype = C.c_int
        L.orc_entry_encode.argtypes = [u8p, C.c_uint64, C.c_int, C.c_int, u8p, C.c_size_t,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_xxh3_reset.argtypes = [C.c_void_p]
        L.orc_xxh3_update.argtypes = [C.c_void_p, u8p, C.c_size_t]
        L.orc_xxh3_digest.restype = C.c_uint64
        L.orc_xxh3_digest.argtypes = [C.c_void_p]
        L.orc_zstd_last_stats.restype = C.POINTER(ZstdStats)

    def xxh3(self, data):
        p, k = _buf(data)
        return self.lib.orc_xxh3_64(p, len(data))

    def xxh3_stream(self, chunks):
        st = C.create_string_buffer(512)
        self.lib.orc_xxh3_reset(st)
        for c in chunks:
            p, k = _buf(c)
            self.lib.orc_xxh3_update(st, p, len(c))
        return self.lib.orc_xxh3_digest(st)

    def xxh32(self, data, seed=0):
        p, k = _buf(data)
        return self.lib.orc_xxh32(p, len(data), seed)

    def xxh64(self, data, seed=0):
        p, k = _buf(data)
        return self.lib.orc_xxh64(p, len(data), seed)

    def _dec(self, fn, data, cap):
        p, k = _buf(data)
        out = (C.c_uint8 * max(1, cap))()
        got = C.c_size_t(0)
        rc = fn(p, len(data), C.cast(out, u8p), cap, C.byref(got))
        return rc, bytes(out[:got.value]) if got.value else b""

    def lz4f_decode(self, data, cap):
        return self._dec(self.lib.orc_lz4f_decode, data, cap)

    def zstd_decode(self, data, cap):
        return self._dec(self.lib.orc_zstd_decode, data, cap)

    def zstd_sequences(self, data, cap, max_seq=1 << 22):
        """Sequences orc_zstd_decode executes for this frame stream, packed like zpack_amd/csrc/zstd_fse4.h."""
        import numpy as np
        buf = np.zeros(max_seq, dtype=np.uint64)
        self.last_trace_bits = np.zeros(max_seq, dtype=np.int64)
        self.lib.orc_zstd_trace_bits.argtypes = [C.c_void_p]
        self.lib.orc_zstd_trace_bits(self.last_trace_bits.ctypes.data)
        self.lib.orc_zstd_trace.argtypes = [C.c_void_p, C.c_size_t]
        self.lib.orc_zstd_trace_count.restype = C.c_size_t
        self.lib.orc_zstd_trace(buf.ctypes.data, max_seq)
        try:
            rc, out = self.zstd_decode(data, cap)
            n = int(self.lib.orc_zstd_trace_count())
        finally:
            self.lib.orc_zstd_trace(None, 0)
            self.lib.orc_zstd_trace_bits(None)
        return rc, buf[:min(n, max_seq)].copy()

    def zstd_stats(self):
        return self.lib.orc_zstd_last_stats().contents

    def _enc(self, fn, bound, data):
        p, k = _buf(data)
        cap = bound(len(data)) + 64
        out = (C.c_uint8 * cap)()
        n = fn(p, len(data), C.cast(out, u8p), cap)
        return bytes(out[:n])

    def lz4f_encode(self, data):
        return self._enc(self.lib.orc_lz4f_encode, self.lib.orc_lz4f_bound, data)

    def zstd_encode(self, data):
        return self._enc(self.lib.orc_zstd_encode, self.lib.orc_zstd_bound, data)

    def entry_decode(self, archive, offset, comp_size, uncomp_size, expect_hash, method, max_size):
        p, k = _buf(archive)
        out = (C.c_uint8 * max(1, max_size))()
        got = C.c_uint64(0)
        h = C.c_uint64(0)
        rc = self.lib.orc_entry_decode(p, len(archive), offset, comp_size, uncomp_size, expect_hash, method,
                                       C.cast(out, u8p), max_size, C.byref(got), C.byref(h))
        return rc, bytes(out[:max_size]), got.value, h.value


class ZstdStats(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("frames", "blocks", "raw_blocks", "rle_blocks", "comp_blocks",
                                          "lit_raw", "lit_rle", "lit_huf", "lit_treeless", "lit_huf_1stream",
                                          "lit_huf_4stream", "huf_fse_weights", "huf_direct_weights")] + \
               [("seq_mode", (C.c_uint32 * 4) * 3), ("sequences", C.c_uint64), ("repcode_uses", C.c_uint32),
                ("window_size", C.c_uint64), ("single_segment", C.c_uint32), ("has_fcs", C.c_uint32),
                ("has_checksum", C.c_uint32)]


# ----------------------------------------------------------------------------- zpack.h ABI view

class FileEntry(C.Structure):                # lib/zpack.h:71-80 (48 bytes on x86-64)
    _fields_ = [("filename", C.c_char_p), ("offset", C.c_uint64), ("comp_size", C.c_uint64),
                ("uncomp_size", C.c_uint64), ("hash", C.c_uint64), ("comp_method", C.c_uint8)]


class Reader(C.Structure):                   # lib/zpack.h:85-110 (112 bytes)
    _fields_ = [("version", C.c_uint16), ("file_entries", C.POINTER(FileEntry)), ("file_count", C.c_uint64),
                ("comp_size", C.c_uint64), ("uncomp_size", C.c_uint64), ("file_size", C.c_size_t),
                ("zstd_dctx", C.c_void_p), ("lz4f_dctx", C.c_void_p), ("last_return", C.c_size_t),
                ("cdr_offset", C.c_uint64), ("eocdr_offset", C.c_uint64), ("buffer", u8p),
                ("buffer_shared", C.c_uint8), ("file", C.c_void_p)]


class CompressOptions(C.Structure):          # lib/zpack.h:115-120
    _fields_ = [("method", C.c_int), ("level", C.c_int)]


class File(C.Structure):                     # lib/zpack.h:125-134 (40 bytes)
    _fields_ = [("filename", C.c_char_p), ("buffer", u8p), ("size", C.c_uint64),
                ("options", C.POINTER(CompressOptions)), ("cctx", C.c_void_p)]


class Writer(C.Structure):                   # lib/zpack.h:139-164 (104 bytes)
    _fields_ = [("buffer", u8p), ("buffer_capacity", C.c_size_t), ("file", C.c_void_p), ("file_size", C.c_size_t),
                ("write_offset", C.c_size_t), ("file_entries", C.POINTER(FileEntry)), ("fe_capacity", C.c_uint64),
                ("file_count", C.c_uint64), ("zstd_cctx", C.c_void_p), ("lz4f_cctx", C.c_void_p),
                ("last_return", C.c_size_t), ("cdr_offset", C.c_uint64), ("eocdr_offset", C.c_uint64)]


class Stream(C.Structure):                   # lib/zpack.h:169-184 (64 bytes)
    _fields_ = [("next_in", u8p), ("avail_in", C.c_size_t), ("total_in", C.c_size_t), ("next_out", u8p),
                ("avail_out", C.c_size_t), ("total_out", C.c_size_t), ("read_back", C.c_size_t),
                ("xxh3_state", C.c_void_p)]


METHOD_NONE, METHOD_ZSTD, METHOD_LZ4 = 0, 1, 2


class ZPackAPI:
    """The zpack.h C API through ctypes; `path` may be the reference build or the product library."""

    def __init__(self, path):
        self.lib = L = C.CDLL(path)
        L.zpack_init_reader_memory.argtypes = [C.POINTER(Reader), u8p, C.c_size_t]
        L.zpack_init_reader_memory_shared.argtypes = [C.POINTER(Reader), u8p, C.c_size_t]
        L.zpack_init_reader.argtypes = [C.POINTER(Reader), C.c_char_p]
        L.zpack_read_file.argtypes = [C.POINTER(Reader), C.POINTER(FileEntry), u8p, C.c_size_t, C.c_void_p]
        L.zpack_read_file_stream.argtypes = [C.POINTER(Reader), C.POINTER(FileEntry), C.POINTER(Stream), C.c_void_p]
        L.zpack_close_reader.argtypes = [C.POINTER(Reader)]
        L.zpack_close_reader.restype = None
        L.zpack_init_writer_heap.argtypes = [C.POINTER(Writer), C.c_size_t]
        L.zpack_init_writer.argtypes = [C.POINTER(Writer), C.c_char_p]
        L.zpack_write_archive.argtypes = [C.POINTER(Writer), C.POINTER(File), C.c_uint64]
        L.zpack_write_header.argtypes = [C.POINTER(Writer)]
        L.zpack_write_data_header.argtypes = [C.POINTER(Writer)]
        L.zpack_write_files.argtypes = [C.POINTER(Writer), C.POINTER(File), C.c_uint64]
        L.zpack_write_file_stream.argtypes = [C.POINTER(Writer), C.POINTER(CompressOptions), C.POINTER(Stream), C.c_void_p]
        L.zpack_write_file_stream_end.argtypes = [C.POINTER(Writer), C.c_char_p, C.POINTER(CompressOptions),
                                                  C.POINTER(Stream), C.c_void_p]
        L.zpack_write_cdr.argtypes = [C.POINTER(Writer)]
        L.zpack_write_eocdr.argtypes = [C.POINTER(Writer)]
        L.zpack_close_writer.argtypes = [C.POINTER(Writer)]
        L.zpack_close_writer.restype = None
        L.zpack_init_stream.argtypes = [C.POINTER(Stream)]
        L.zpack_reset_stream.argtypes = [C.POINTER(Stream)]
        L.zpack_reset_stream.restype = None
        L.zpack_close_stream.argtypes = [C.POINTER(Stream)]
        L.zpack_close_stream.restype = None
        for n in ("zpack_get_dstream_in_size", "zpack_get_dstream_out_size", "zpack_get_cstream_in_size",
                  "zpack_get_cstream_out_size"):
            getattr(L, n).restype = C.c_size_t
            getattr(L, n).argtypes = [C.c_int]

    # ---- convenience: build an archive on the heap from [(name, bytes)] ----
    def write_archive(self, files, method, level):
        w = Writer()
        rc = self.lib.zpack_init_writer_heap(C.byref(w), 0)
        assert rc == 0, rc
        opts = CompressOptions(method, level)
        arr = (File * len(files))()
        keep = []
        for i, (name, data) in enumerate(files):
            p, k = _buf(data)
            keep.append(k)
            arr[i].filename = name.encode()
            arr[i].buffer = p
            arr[i].size = len(data)
            arr[i].options = C.pointer(opts)
            arr[i].cctx = None
        rc = self.lib.zpack_write_archive(C.byref(w), arr, len(files))
        if rc != 0:
            self.lib.zpack_close_writer(C.byref(w))
            raise RuntimeError("zpack_write_archive -> %d" % rc)
        out = bytes(C.cast(w.buffer, C.POINTER(C.c_uint8 * w.file_size)).contents)
        self.lib.zpack_close_writer(C.byref(w))
        return out

    def open_memory(self, archive):
        r = Reader()
        p, k = _buf(archive)
        rc = self.lib.zpack_init_reader_memory_shared(C.byref(r), p, len(archive))
        return rc, r, k

    def entries(self, r):
        return [dict(filename=r.file_entries[i].filename.decode(), offset=r.file_entries[i].offset,
                     comp_size=r.file_entries[i].comp_size, uncomp_size=r.file_entries[i].uncomp_size,
                     hash=r.file_entries[i].hash, method=r.file_entries[i].comp_method)
                for i in range(r.file_count)]

    def read_file(self, r, i, max_size):
        out = (C.c_uint8 * max(1, max_size))()
        rc = self.lib.zpack_read_file(C.byref(r), C.byref(r.file_entries[i]), C.cast(out, u8p), max_size, None)
        return rc, bytes(out[:max_size])

    def close_reader(self, r):
        self.lib.zpack_close_reader(C.byref(r))


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        _ref = ZPackAPI(REF_SO)
    return _ref
