#!/usr/bin/env python3
"""Developer: stream one big recipe through zpack_read_file_stream and report where the delivered bytes differ from the plaintext."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zpack_amd
from benchdata import datagen as dg
from tests import zpk
from tests._libs import ZPackAPI, Stream, u8p
from tests.test_gpu_zpack_api import _stream_entry
labels = sys.argv[1].split(",") if len(sys.argv) > 1 else ["zstd_3_64m_records"]
inw = int(sys.argv[2]) if len(sys.argv) > 2 else 131075
Z = ZPackAPI(zpack_amd.ZPACK_SO)
for label in labels:
  x = [r for r in json.load(open("tests/golden/recipes_big.json")) if r["label"] == label][0]
  plain = dg.fill(x["cls"], x["seed"], x["index"], x["size"])
  frame = np.frombuffer(dg.compress(x["method"], x["level"], plain), dtype=np.uint8)
  arc = zpk.assemble([frame.tobytes()], [("big", 10, len(frame), x["size"], x["hash"], x["method"])])
  sink = np.zeros(x["size"], dtype=np.uint8)
  rc, r, keep = Z.open_memory(arc)
  rc, first, rss, got = _stream_entry(Z, r, 0, inw, 1 << 20, sink)
  bad = np.nonzero(sink != plain)[0]
  print(label, "rc", rc, "got", got, "first_out_in", first, "mismatches", bad.size)
  if bad.size:
      runs = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
      print("runs", len(runs), [(int(q[0]), len(q)) for q in runs[:12]])
      a = int(bad[0]); print("at", a, "got", sink[a:a + 16].tolist(), "want", plain[a:a + 16].tolist())
