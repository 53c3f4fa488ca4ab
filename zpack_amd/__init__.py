"""zpack_amd — MI355X-native batch entry codec behind the ZPack C API.

The product is native: `libzpk_codec.so` (hand-written HIP kernels for gfx950 + the C-ABI of
include/zpack_codec.h) and `libzpack_amd.so` (the zpack.h reader/writer/stream API in C on top of it).
This package is a thin ctypes view for tests and bench.py.  There is no CPU fallback: if the HIP
library is missing or no device is usable, construction raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CODEC_SO = os.environ.get("ZPACK_AMD_CODEC_SO") or os.path.join(HERE, "libzpk_codec.so")     # override: A/B builds of the kernels
ZPACK_SO = os.path.join(HERE, "libzpack_amd.so")

METHOD_NONE, METHOD_ZSTD, METHOD_LZ4 = 0, 1, 2
DF_SKIP_HASH = 1
DF_GENERAL = 2
K_CLASSIFY, K_STORED, K_LZ4, K_ZSTD, K_ZSTD_FSE, K_PACK, K_ENCODE = 0, 1, 2, 3, 4, 5, 7
OPT_ENC_SPLIT_MIN, OPT_DEC_SPLIT_MIN, OPT_ORDER_MIN, OPT_ORDER_FAST_LAST = 6, 7, 8, 9      # zpk_codec_set_option (include/zpack_codec.h)

# zpk_decode_desc / zpk_decode_result / zpk_encode_desc / zpk_encode_result (include/zpack_codec.h)
DECODE_DESC = np.dtype([("src_offset", "<u8"), ("comp_size", "<u8"), ("uncomp_size", "<u8"), ("expect_hash", "<u8"),
                        ("dst_offset", "<u8"), ("dst_capacity", "<u8"), ("method", "<u4"), ("flags", "<u4")])
DECODE_RESULT = np.dtype([("status", "<i4"), ("detail", "<u4"), ("produced", "<u8"), ("hash", "<u8")])
ENCODE_DESC = np.dtype([("src_offset", "<u8"), ("size", "<u8"), ("dst_offset", "<u8"), ("dst_capacity", "<u8"),
                        ("method", "<u4"), ("level", "<i4")])
ENCODE_RESULT = np.dtype([("status", "<i4"), ("detail", "<u4"), ("comp_size", "<u8"), ("hash", "<u8")])
assert DECODE_DESC.itemsize == 56 and DECODE_RESULT.itemsize == 24
assert ENCODE_DESC.itemsize == 40 and ENCODE_RESULT.itemsize == 24

_lib = None


class CodecUnavailable(RuntimeError):
    pass


def lib():
    """Load libzpk_codec.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(CODEC_SO):
            raise CodecUnavailable("%s is missing: run `python -m zpack_amd.build` (or __graft_entry__.build()); "
                                   "there is no CPU fallback" % CODEC_SO)
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7; importing it first makes
        # the dynamic loader resolve this library's libamdhip64.so.7 dependency to the same copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(CODEC_SO)
        vp, u64, u8p = C.c_void_p, C.c_uint64, C.c_void_p
        L.zpk_codec_create.argtypes = [C.POINTER(vp), C.c_int]
        L.zpk_codec_destroy.argtypes = [vp]
        L.zpk_codec_destroy.restype = None
        L.zpk_codec_last_error.argtypes = [vp]
        L.zpk_codec_last_error.restype = C.c_char_p
        L.zpk_codec_device.argtypes = [vp]
        L.zpk_codec_decode_batch_device.argtypes = [vp, u8p, u64, vp, u64, u8p, u64, vp, vp]
        L.zpk_codec_decode_batch_host.argtypes = [vp, u8p, u64, vp, u64, vp, vp]
        L.zpk_codec_encode_batch_device.argtypes = [vp, u8p, u64, vp, u64, u8p, u64, vp, vp]
        L.zpk_codec_encode_batch_host.argtypes = [vp, vp, vp, u64, vp, vp]
        L.zpk_codec_pack_batch_device.argtypes = [vp, u8p, vp, vp, u64, u8p, u64, vp, u64, vp]
        L.zpk_codec_compress_bound.argtypes = [C.c_uint32, C.c_size_t]
        L.zpk_codec_compress_bound.restype = C.c_size_t
        L.zpk_codec_hash_batch_device.argtypes = [vp, u8p, vp, vp, u64, vp, vp]
        L.zpk_codec_hash_host.argtypes = [vp, u8p, u64, C.POINTER(u64)]
        L.zpk_codec_set_profiling.argtypes = [vp, C.c_int]
        L.zpk_codec_kernel_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
        L.zpk_codec_debug_read.argtypes = [vp, vp, u64]
        L.zpk_codec_debug_fetch.argtypes = [vp, C.c_int, u64, vp, u64]
        L.zpk_codec_decode_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.zpk_codec_timer_start.argtypes = [vp, vp]
        L.zpk_codec_timer_stop.argtypes = [vp, vp, C.POINTER(C.c_float)]
        _lib = L
    return _lib


class Codec:
    """One codec context = one device (one process per GPU under torch.distributed)."""

    def __init__(self, device=-1):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.zpk_codec_create(C.byref(h), device)
        if rc != 0:
            raise CodecUnavailable("zpk_codec_create(device=%d) failed with %d: no usable HIP device; "
                                   "there is no CPU fallback" % (device, rc))
        self.h = h
        self.device = self.L.zpk_codec_device(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.zpk_codec_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.L.zpk_codec_last_error(self.h).decode()))

    # ---- device-resident batch (torch uint8 CUDA tensors) ----
    # stream=None runs the batch on the codec's OWN stream, which is not ordered with torch's: fills / copies torch has enqueued on its
    # current stream for these tensors (a torch.zeros of the result array ...) must have finished before the codec touches them.
    @staticmethod
    def _settle(stream):
        if not stream:
            import sys
            t = sys.modules.get("torch")
            if t is not None and t.cuda.is_available() and t.cuda.is_initialized():
                t.cuda.current_stream().synchronize()

    def decode_batch_device(self, src, desc_dev, n, dst, results_dev, stream=None):
        self._settle(stream)
        st = C.c_void_p(stream) if stream else None
        self._chk(self.L.zpk_codec_decode_batch_device(self.h, src.data_ptr(), src.numel(), desc_dev.data_ptr(), n,
                                                       dst.data_ptr(), dst.numel(), results_dev.data_ptr(), st),
                  "zpk_codec_decode_batch_device")

    def decode_big_device(self, src, desc, dst):
        """ONE entry: src / dst are uint8 CUDA tensors, desc a one-element np array of DECODE_DESC (host) -> DECODE_RESULT (host, after the
        entry is decoded and verified)."""
        self._settle(None)
        desc = np.ascontiguousarray(desc, dtype=DECODE_DESC)
        res = np.zeros(1, dtype=DECODE_RESULT)
        self.L.zpk_codec_decode_big_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        self._chk(self.L.zpk_codec_decode_big_device(self.h, src.data_ptr(), src.numel(), desc.ctypes.data, dst.data_ptr(), dst.numel(), res.ctypes.data),
                  "zpk_codec_decode_big_device")
        return res[0]

    def encode_batch_device(self, src, desc_dev, n, dst, results_dev, stream=None):
        self._settle(stream)
        st = C.c_void_p(stream) if stream else None
        self._chk(self.L.zpk_codec_encode_batch_device(self.h, src.data_ptr(), src.numel(), desc_dev.data_ptr(), n,
                                                       dst.data_ptr(), dst.numel(), results_dev.data_ptr(), st),
                  "zpk_codec_encode_batch_device")

    def pack_batch_device(self, slots, desc_dev, results_dev, n, packed, offsets_dev, max_entry_size, stream=None):
        """offsets_dev: int64 CUDA tensor of n + 1; packed: uint8 CUDA tensor or None (sizes only)."""
        self._settle(stream)
        st = C.c_void_p(stream) if stream else None
        self._chk(self.L.zpk_codec_pack_batch_device(self.h, slots.data_ptr(), desc_dev.data_ptr(), results_dev.data_ptr(), n,
                                                     packed.data_ptr() if packed is not None else None,
                                                     packed.numel() if packed is not None else 0, offsets_dev.data_ptr(),
                                                     max_entry_size, st), "zpk_codec_pack_batch_device")

    def hash_batch_device(self, src, offsets_dev, sizes_dev, n, hashes_dev, stream=None):
        self._settle(stream)
        st = C.c_void_p(stream) if stream else None
        self._chk(self.L.zpk_codec_hash_batch_device(self.h, src.data_ptr(), offsets_dev.data_ptr(), sizes_dev.data_ptr(),
                                                     n, hashes_dev.data_ptr(), st), "zpk_codec_hash_batch_device")

    def set_option(self, option, value):
        self.L.zpk_codec_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._chk(self.L.zpk_codec_set_option(self.h, option, value), "set_option")

    def set_profiling(self, on):
        self._chk(self.L.zpk_codec_set_profiling(self.h, 1 if on else 0), "set_profiling")

    def kernel_ms(self, which):
        ms = C.c_float(0)
        self._chk(self.L.zpk_codec_kernel_ms(self.h, which, C.byref(ms)), "kernel_ms")
        return ms.value

    def decode_stats(self):
        """Counters of the last decode batch: entries per method and how the Zstandard ones were finished."""
        a = (C.c_uint32 * 8)()
        self._chk(self.L.zpk_codec_decode_stats(self.h, a), "decode_stats")
        b = (C.c_uint32 * 16)()
        self.L.zpk_codec_decode_stats2.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        self._chk(self.L.zpk_codec_decode_stats2(self.h, b), "decode_stats2")
        return dict(stored=a[0], zstd=a[1], lz4=a[2], zstd_two_stage=a[3], zstd_fused=a[4], fse_watchdog=a[5], fse_budget=a[6],
                    zstd_arena_refused=bool(a[7] >> 31), retried_lz4=b[0], retried_zstd=b[1],
                    lz4_long_runs=b[3],
                    frame_parallel_entries=b[5], frame_parallel_frames=b[6], zstd_blocks_flags=b[7])

    def debug_fetch(self, what, offset, count, dtype):
        a = np.zeros(count, dtype=dtype)
        self._chk(self.L.zpk_codec_debug_fetch(self.h, what, offset, a.ctypes.data, a.nbytes), "debug_fetch")
        return a

    def debug_read(self, n):
        a = np.zeros((n, 8), dtype=np.uint64)
        self._chk(self.L.zpk_codec_debug_read(self.h, a.ctypes.data, a.nbytes), "debug_read")
        return a

    def timer_start(self, stream=None):
        self._chk(self.L.zpk_codec_timer_start(self.h, C.c_void_p(stream) if stream else None), "timer_start")

    def timer_stop(self, stream=None):
        ms = C.c_float(0)
        self._chk(self.L.zpk_codec_timer_stop(self.h, C.c_void_p(stream) if stream else None, C.byref(ms)), "timer_stop")
        return ms.value

    # ---- host-pointer forms (numpy) ----
    def decode_batch_host(self, archive, desc, caps=None):
        """archive: bytes/np.uint8; desc: np array of DECODE_DESC.  Returns (results, [output bytes])."""
        arc = np.ascontiguousarray(np.frombuffer(archive, dtype=np.uint8) if not isinstance(archive, np.ndarray) else archive)
        desc = np.ascontiguousarray(desc, dtype=DECODE_DESC)
        n = len(desc)
        outs = [np.zeros(max(1, int(d["dst_capacity"])), dtype=np.uint8) for d in desc]
        ptrs = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs])
        res = np.zeros(n, dtype=DECODE_RESULT)
        self._chk(self.L.zpk_codec_decode_batch_host(self.h, arc.ctypes.data, arc.size, desc.ctypes.data, n, ptrs,
                                                     res.ctypes.data), "zpk_codec_decode_batch_host")
        return res, [o[:int(d["dst_capacity"])] for o, d in zip(outs, desc)]

    def hash_host(self, data):
        a = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
        h = C.c_uint64(0)
        self._chk(self.L.zpk_codec_hash_host(self.h, a.ctypes.data if a.size else None, a.size, C.byref(h)), "zpk_codec_hash_host")
        return h.value

    def compress_bound(self, method, n):
        return self.L.zpk_codec_compress_bound(method, n)


def decode_descs_from_batch(batch, dst_align=256, flags=0):
    """Descriptors for every entry of a benchdata Batch (or any object with the same arrays), output
    slots laid out back to back (aligned); returns (desc, total_dst_bytes)."""
    n = batch.n
    d = np.zeros(n, dtype=DECODE_DESC)
    d["src_offset"] = batch.offsets
    d["comp_size"] = batch.comp_sizes
    d["uncomp_size"] = batch.uncomp_sizes
    d["expect_hash"] = batch.hashes
    d["dst_capacity"] = batch.uncomp_sizes
    d["method"] = batch.methods
    d["flags"] = flags
    sizes = (batch.uncomp_sizes.astype(np.uint64) + np.uint64(dst_align - 1)) & ~np.uint64(dst_align - 1)
    offs = np.zeros(n, dtype=np.uint64)
    if n > 1:
        offs[1:] = np.cumsum(sizes[:-1])
    d["dst_offset"] = offs
    total = int(offs[-1] + sizes[-1]) if n else 0
    return d, total
