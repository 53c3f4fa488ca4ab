"""What the large-entry / streaming writers of this library rely on: an entry that is a SEQUENCE of 512 KiB frames,
each stating its content size, is read by the compiled reference — one-shot (lib/zpack_read.c:380, :414-439) and through its streaming
reader with the window sizes it recommends (zpack_get_dstream_in_size / _out_size, lib/zpack_read.c:515-640) — and by the oracle.
The frames here are made with the real liblz4 / libzstd (the helpers of tests/golden/make_golden.py), no GPU involved.  With other
window sizes the reference's streaming reader fails on ONE-frame entries in exactly the same way (its window protocol: a call that
consumes nothing leaves no room to read), so a frame sequence adds no incompatibility of its own — both facts are asserted."""
import os
import sys

import numpy as np
import pytest

from benchdata import datagen as dg
from tests import zpk
from tests._libs import have_ref, oracle, ref

pytestmark = pytest.mark.skipif(not have_ref() or not os.path.exists("/opt/conda/lib/liblz4.so.1") or not os.path.exists("/opt/conda/lib/libzstd.so.1"),
                                reason="needs oracle/_ref (the compiled reference) and the image's liblz4 / libzstd")
PIECE = 512 << 10


@pytest.mark.parametrize("method", [1, 2])
def test_reference_reads_a_sequence_of_frames_one_shot_and_streaming(method):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    from tests.test_gpu_zpack_api import _stream_entry
    R = ref()
    plain = dg.fill(dg.TEXT, 5, 1, 3 * PIECE + 777).tobytes()
    h = dg.xxh3(np.frombuffer(plain, dtype=np.uint8))

    def entry(pieces):
        fr = b"".join(mg._zstd_custom(p, level=1, pledged=True) if method == 1 else mg._lz4f_custom(p, content_size=1) for p in pieces)
        return fr, zpk.assemble([fr], [("f", 10, len(fr), len(plain), h, method)])
    seq, arc_seq = entry([plain[a:a + PIECE] for a in range(0, len(plain), PIECE)])
    one, arc_one = entry([plain])
    ins, outs = R.lib.zpack_get_dstream_in_size(method), R.lib.zpack_get_dstream_out_size(method)
    verdicts = {}
    for name, arc in (("sequence", arc_seq), ("one frame", arc_one)):
        rc, r, keep = R.open_memory(arc)
        assert rc == 0
        rc, out = R.read_file(r, 0, len(plain))
        assert rc == 0 and out == plain, (name, "one-shot", rc)
        for win in ((ins, outs), (4096, 4096), (1 << 20, 1 << 20)):
            sink = np.zeros(len(plain), dtype=np.uint8)
            rc, _, _, got = _stream_entry(R, r, 0, win[0], win[1], sink)
            verdicts[(name, win)] = (rc, got == len(plain) and sink.tobytes() == plain)
        R.close_reader(r)
    assert verdicts[("sequence", (ins, outs))] == (0, True), verdicts
    for win in ((ins, outs), (4096, 4096), (1 << 20, 1 << 20)):
        assert verdicts[("sequence", win)] == verdicts[("one frame", win)], (win, verdicts)       # nothing a frame sequence adds
    o = oracle()
    fr_off = 10
    rc, out, got, hh = o.entry_decode(arc_seq, fr_off, len(seq), len(plain), h, method, len(plain))
    assert rc == 0 and out == plain
