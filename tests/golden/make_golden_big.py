#!/usr/bin/env python3
"""Large-entry fixtures (64 MiB and 512 MiB per method), produced by RUNNING THE REFERENCE in the build container.

Same idea as recipes.json of make_golden.py, for entries far beyond a device slot of the batch benchmarks: the
reference writer (oracle/_ref = /root/reference/lib/*.c compiled in place) compresses seeded synthetic data, the
reference reader decodes it back (rc 0, bytes equal), and what is committed is the recipe — (class, seed, index,
size, method, level) — plus the size and XXH3 of the reference-written frame and the XXH3 of the plaintext.  The GPU
tests regenerate the identical frame with the same libraries (benchdata/libzpkgen.so: same call sequence) and check
the codec's verdict, size and hash against these values.  Never run on the GPU box.

    python tests/golden/make_golden_big.py        -> tests/golden/recipes_big.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from benchdata import datagen as dg            # noqa: E402
from tests._libs import ref, Writer, Reader, File, CompressOptions, u8p   # noqa: E402

R = ref()
CASES = [  # (label, class, size, method, level)
    ("lz4_0_64m_text", dg.TEXT, 64 << 20, dg.LZ4, 0),
    ("zstd_3_64m_text", dg.TEXT, 64 << 20, dg.ZSTD, 3),
    ("lz4_0_64m_records", dg.RECORDS, 64 << 20, dg.LZ4, 0),
    ("zstd_3_64m_records", dg.RECORDS, 64 << 20, dg.ZSTD, 3),
    ("none_64m_random", dg.RANDOM, 64 << 20, dg.NONE, 0),
    ("lz4_0_512m_text", dg.TEXT, 512 << 20, dg.LZ4, 0),
    ("zstd_3_512m_text", dg.TEXT, 512 << 20, dg.ZSTD, 3),
]


def ref_write_read(data, method, level):
    """zpack_write_archive of one entry with the reference, then zpack_read_file of it; returns (frame view, entry dict)"""
    w = Writer()
    assert R.lib.zpack_init_writer_heap(C.byref(w), 0) == 0
    opts = CompressOptions(method, level)
    f = (File * 1)()
    f[0].filename = b"big"
    f[0].buffer = data.ctypes.data_as(u8p)
    f[0].size = len(data)
    f[0].options = C.pointer(opts)
    rc = R.lib.zpack_write_archive(C.byref(w), f, 1)
    assert rc == 0, rc
    arc = np.ctypeslib.as_array(w.buffer, shape=(w.file_size,)).copy()
    R.lib.zpack_close_writer(C.byref(w))
    r = Reader()
    assert R.lib.zpack_init_reader_memory_shared(C.byref(r), arc.ctypes.data_as(u8p), arc.size) == 0
    e = R.entries(r)[0]
    out = np.empty(len(data), dtype=np.uint8)
    rc = R.lib.zpack_read_file(C.byref(r), C.byref(r.file_entries[0]), out.ctypes.data_as(u8p), len(data), None)
    assert rc == 0 and np.array_equal(out, data), rc
    R.close_reader(r)
    return arc[e["offset"]:e["offset"] + e["comp_size"]], e


def main():
    out = []
    for i, (label, cls, size, method, level) in enumerate(CASES):
        data = dg.fill(cls, 61, i, size)
        frame, e = ref_write_read(data, method, level)
        mine = np.frombuffer(dg.compress(method, level, data), dtype=np.uint8)
        assert np.array_equal(mine, frame), label          # the datagen path reproduces the reference's frame
        out.append(dict(label=label, cls=cls, seed=61, index=i, size=size, method=method, level=level,
                        comp_size=int(e["comp_size"]), frame_xxh3=dg.xxh3(frame), hash=int(e["hash"])))
        assert e["hash"] == dg.xxh3(data)
        print(out[-1], flush=True)
    with open(os.path.join(HERE, "recipes_big.json"), "w") as fh:
        json.dump(out, fh, indent=0, separators=(",", ":"))


if __name__ == "__main__":
    main()
