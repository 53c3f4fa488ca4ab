"""ctypes view of benchdata/libzpkgen.so — deterministic synthetic .zpk archives (see datagen.c)."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libzpkgen.so")

TEXT, RECORDS, RANDOM, RUNS, MIX = 0, 1, 2, 3, -1
NONE, ZSTD, LZ4, COIN = 0, 1, 2, -1


class _Batch(C.Structure):
    _fields_ = [("n", C.c_uint64), ("archive_size", C.c_uint64), ("archive", C.POINTER(C.c_uint8)),
                ("offsets", C.POINTER(C.c_uint64)), ("comp_sizes", C.POINTER(C.c_uint64)),
                ("uncomp_sizes", C.POINTER(C.c_uint64)), ("hashes", C.POINTER(C.c_uint64)),
                ("methods", C.POINTER(C.c_uint8)), ("classes", C.POINTER(C.c_uint8)),
                ("total_comp", C.c_uint64), ("total_uncomp", C.c_uint64), ("cdr_offset", C.c_uint64),
                ("error", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise RuntimeError("benchdata/libzpkgen.so missing: run `make -C benchdata` (or __graft_entry__.build())")
        L = C.CDLL(SO)
        L.zpkgen_make.restype = C.POINTER(_Batch)
        L.zpkgen_make.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.zpkgen_make_range.restype = C.POINTER(_Batch)
        L.zpkgen_make_range.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int]
        L.zpkgen_sizes.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.zpkgen_free.argtypes = [C.POINTER(_Batch)]
        L.zpkgen_fill.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64]
        L.zpkgen_compress.restype = C.c_size_t
        L.zpkgen_compress.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.zpkgen_bound.restype = C.c_size_t
        L.zpkgen_bound.argtypes = [C.c_int, C.c_size_t]
        L.zpkgen_xxh3.restype = C.c_uint64
        L.zpkgen_xxh3.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


class Batch:
    """A complete in-memory .zpk plus its entry table as numpy arrays (views into C memory).
    first > 0: the SLICE [first, first + n) of the archive with that seed as a self-contained image (one rank's share of a static
    shard): entry k here is entry first + k of the whole archive, byte for byte; offsets are relative to this image."""

    def __init__(self, n, size_lo, size_hi=None, method=LZ4, level=0, seed=1, mix=MIX, threads=None, first=0):
        if size_hi is None:
            size_hi = size_lo
        if threads is None:
            threads = max(1, len(os.sched_getaffinity(0)))
        self.first = first
        self._p = lib().zpkgen_make_range(first, n, size_lo, size_hi, method, level, seed, mix, threads)
        b = self._p.contents
        if b.error:
            raise RuntimeError("zpkgen_make failed: %d" % b.error)
        self.n = int(b.n)
        self.seed, self.mix = seed, mix
        as_np = lambda ptr, cnt, dt: np.ctypeslib.as_array(ptr, shape=(max(cnt, 1),))[:cnt].view(dt)
        self.archive = as_np(b.archive, int(b.archive_size), np.uint8)
        self.offsets = as_np(b.offsets, self.n, np.uint64)
        self.comp_sizes = as_np(b.comp_sizes, self.n, np.uint64)
        self.uncomp_sizes = as_np(b.uncomp_sizes, self.n, np.uint64)
        self.hashes = as_np(b.hashes, self.n, np.uint64)
        self.methods = as_np(b.methods, self.n, np.uint8)
        self.classes = as_np(b.classes, self.n, np.uint8)
        self.total_comp = int(b.total_comp)
        self.total_uncomp = int(b.total_uncomp)

    def plaintext(self, i):
        n = int(self.uncomp_sizes[i])
        out = np.empty(max(n, 1), dtype=np.uint8)
        lib().zpkgen_fill(int(self.classes[i]), self.seed, self.first + i, out.ctypes.data, n)
        return out[:n]

    def close(self):
        if self._p:
            lib().zpkgen_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sizes(n, size_lo, size_hi, seed):
    """uncompressed sizes of entries [0, n) of the archive with that seed — without building it"""
    out = np.zeros(max(n, 1), dtype=np.uint64)
    lib().zpkgen_sizes(n, size_lo, size_hi, seed, out.ctypes.data)
    return out[:n]


def fill(cls, seed, index, n):
    out = np.empty(max(n, 1), dtype=np.uint8)
    lib().zpkgen_fill(cls, seed, index, out.ctypes.data, n)
    return out[:n]


def compress(method, level, data):
    data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    cap = lib().zpkgen_bound(method, len(data)) + 64
    out = np.empty(cap, dtype=np.uint8)
    n = lib().zpkgen_compress(method, level, data.ctypes.data if len(data) else None, len(data), out.ctypes.data, cap)
    if n == 0 and not (method == NONE and len(data) == 0):
        raise RuntimeError("compress failed")
    return out[:n].tobytes()


def xxh3(data):
    a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data)
    return int(lib().zpkgen_xxh3(a.ctypes.data if len(a) else None, len(a)))
