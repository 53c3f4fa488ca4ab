"""CPU-side checks of the drop-in boundary: both native libraries load, export every symbol their headers
declare, keep the reference's struct layouts, and — without a GPU — FAIL LOUDLY instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import pytest

import zpack_amd
from tests import _libs as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exports(so):
    return subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()


def test_codec_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "zpack_codec.h")).read()
    names = sorted(set(re.findall(r"\b(zpk_(?:codec|dstream|cstream)_\w+)\(", hdr)))
    syms = _exports(zpack_amd.CODEC_SO)
    assert len(names) >= 25
    assert [n for n in names if " T " + n not in syms] == []
    lib = C.CDLL(zpack_amd.CODEC_SO)
    assert lib.zpk_codec_abi_version() == 3


def test_zpack_library_exports_the_reference_api():
    hdr = open(os.path.join(ROOT, "include", "zpack.h")).read()
    names = re.findall(r"ZPACK_EXPORT [\w\s\*]+?\b(zpack_\w+)\(", hdr)
    # the 52 portable functions of the reference header (lib/zpack.h:237-742) + the 2 additive batch reads
    assert len(names) == 54 and "zpack_read_files" in names and "zpack_read_files_packed" in names
    syms = _exports(zpack_amd.ZPACK_SO)
    assert [n for n in names if " T " + n not in syms] == []


def test_struct_layouts_match_reference_abi():
    """SURVEY.md §8b [probe, x86-64]: sizes and key field offsets of the reference structs."""
    assert C.sizeof(L.FileEntry) == 48 and L.FileEntry.offset.offset == 8 and L.FileEntry.hash.offset == 32 and L.FileEntry.comp_method.offset == 40
    assert C.sizeof(L.Reader) == 112 and L.Reader.zstd_dctx.offset == 48 and L.Reader.last_return.offset == 64 and L.Reader.buffer.offset == 88 and L.Reader.file.offset == 104
    assert C.sizeof(L.File) == 40 and C.sizeof(L.CompressOptions) == 8
    assert C.sizeof(L.Writer) == 104 and L.Writer.zstd_cctx.offset == 64 and L.Writer.last_return.offset == 80
    assert C.sizeof(L.Stream) == 64 and L.Stream.read_back.offset == 48 and L.Stream.xxh3_state.offset == 56
    # the C side agrees (a tiny probe compiled against include/zpack.h)
    src = r'''#include <stdio.h>
#include <stddef.h>
#include "zpack.h"
int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(zpack_file_entry), sizeof(zpack_reader), sizeof(zpack_file),
 sizeof(zpack_compress_options), sizeof(zpack_writer), sizeof(zpack_stream), offsetof(zpack_reader, file), offsetof(zpack_writer, last_return));return 0;}'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), "-o", os.path.join(d, "p"), os.path.join(d, "p.c")])
        out = subprocess.check_output([os.path.join(d, "p")]).decode().split()
    assert out == ["48", "112", "40", "8", "104", "64", "104", "80"]


def test_stream_buffer_sizes_and_bounds_match_reference(golden_dir):
    import json
    ka = json.load(open(os.path.join(golden_dir, "known_answers.json")))
    Z = L.ZPackAPI(zpack_amd.ZPACK_SO)
    for m in (0, 1, 2):
        assert Z.lib.zpack_get_dstream_in_size(m) == ka["dstream_in_%d" % m]
        assert Z.lib.zpack_get_dstream_out_size(m) == ka["dstream_out_%d" % m]
        assert Z.lib.zpack_get_cstream_in_size(m) == ka["cstream_in_%d" % m]
        assert Z.lib.zpack_get_cstream_out_size(m) == ka["cstream_out_%d" % m]
    lib = zpack_amd.lib()
    # ZSTD_COMPRESSBOUND / LZ4F_compressBound(n, NULL) values measured on the reference
    assert lib.zpk_codec_compress_bound(1, 65536) == 65824 and lib.zpk_codec_compress_bound(1, 262144) == 263168
    assert lib.zpk_codec_compress_bound(1, 1048576) == 1052672
    assert lib.zpk_codec_compress_bound(2, 65536) == 65552 and lib.zpk_codec_compress_bound(2, 0) == 65551
    assert lib.zpk_codec_compress_bound(2, 1048576) == 1048712 and lib.zpk_codec_compress_bound(0, 12345) == 12345
    o = L.oracle()
    for n in (0, 1, 65535, 65536, 65537, 1 << 20):
        assert o.lib.orc_lz4f_bound(n) == lib.zpk_codec_compress_bound(2, n)


def _no_gpu():
    try:
        return zpack_amd.lib().zpk_codec_device_count() == 0
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine WITHOUT a GPU")
def test_no_gpu_fails_loudly_never_falls_back(golden_dir):
    with pytest.raises(zpack_amd.CodecUnavailable):
        zpack_amd.Codec(0)
    # container parsing is host code and works; the hot path must refuse, not silently run on the CPU
    Z = L.ZPackAPI(zpack_amd.ZPACK_SO)
    for arc in ("archive_none.zpk", "archive_zstd.zpk", "archive_lz4.zpk"):
        a = open(os.path.join(golden_dir, "ref_workdir", arc), "rb").read()
        rc, r, keep = Z.open_memory(a)
        assert rc == 0 and r.file_count == 2
        ents = Z.entries(r)
        assert ents[0]["hash"] == 0x7874cba47d02b07d and ents[1]["hash"] == 0x15f25c0f24dd8e52    # tests/open_archive.c:21-25
        rc, out = Z.read_file(r, 0, 350)
        assert rc == 24, "zpack_read_file must return ZPACK_ERROR_NOT_AVAILABLE without a device, got %d" % rc
        Z.close_reader(r)
    w = L.Writer()
    assert Z.lib.zpack_init_writer_heap(C.byref(w), 0) == 0
    with pytest.raises(RuntimeError, match="-> 24"):
        Z.write_archive([("a", b"hello")], 2, 0)
    Z.lib.zpack_close_writer(C.byref(w))


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under zpack_amd/ or include/ may reference it."""
    bad = []
    for base in ("zpack_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            for f in fn:
                if f.endswith((".py", ".c", ".h", ".hip", ".inc", ".cpp")):
                    t = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"liboracle|oracle/|orc_|libzpack_ref|from tests|import tests", t) and f != "build.py":
                        bad.append(os.path.join(dp, f))
    assert bad == []
    so_deps = subprocess.check_output(["ldd", zpack_amd.ZPACK_SO]).decode()
    assert "oracle" not in so_deps and "zpack_ref" not in so_deps


def test_option_numbers_of_the_python_view_match_the_header():
    """zpk_codec_set_option's option numbers (include/zpack_codec.h) and the flag bits the descriptors carry are what the Python
    view passes; an unknown option must stay an error (ZPK_E_INVALID) on a library without a device as well."""
    import re
    hdr = open(os.path.join(ROOT, "include", "zpack_codec.h")).read()
    nums = {m.group(1): int(m.group(2)) for m in re.finditer(r"(ZPK_OPT_[A-Z0-9_]+)\s*=\s*(\d+)", hdr)}
    view = {"ZPK_OPT_ENC_SPLIT_MIN": zpack_amd.OPT_ENC_SPLIT_MIN, "ZPK_OPT_DEC_SPLIT_MIN": zpack_amd.OPT_DEC_SPLIT_MIN,
            "ZPK_OPT_ORDER_MIN": zpack_amd.OPT_ORDER_MIN, "ZPK_OPT_ORDER_FAST_LAST": zpack_amd.OPT_ORDER_FAST_LAST}
    assert nums == view, (nums, view)
    assert len(set(nums.values())) == len(nums)
    assert re.search(r"#define\s+ZPK_EF_PIECE\s+0x80000000u", hdr) and re.search(r"#define\s+ZPK_DF_SKIP_HASH\s+1u", hdr)
    assert zpack_amd.DF_SKIP_HASH == 1
