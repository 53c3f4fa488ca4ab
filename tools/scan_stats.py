import os, sys
import numpy as np
os.environ["ZPK_DEBUG_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, zpack_amd
from benchdata import datagen as dg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
mix = int(sys.argv[2]) if len(sys.argv) > 2 else -1
b = dg.Batch(n, 65536, 65536, method=dg.LZ4, level=0, seed=1, mix=mix)
desc, total = zpack_amd.decode_descs_from_batch(b)
dev = torch.device("cuda:0"); codec = zpack_amd.Codec(0); codec.set_option(zpack_amd.OPT_LZ4_RING, 1)
src = torch.from_numpy(b.archive).to(dev); dst = torch.empty(total, dtype=torch.uint8, device=dev)
ddesc = torch.from_numpy(desc.view(np.uint8)).to(dev); dres = torch.zeros(n * 24, dtype=torch.uint8, device=dev)
for _ in range(2):
    codec.decode_batch_device(src, ddesc, n, dst, dres)
torch.cuda.synchronize()
a = np.zeros((1024, 8), dtype=np.uint64)
codec._chk(codec.L.zpk_codec_debug_read(codec.h, a.ctypes.data, a.nbytes), "debug_read")
a = a[a[:, 4] > 0].astype(np.float64)
m = a.mean(0)
print("scan waves sampled %d: total cycles %.0f, in events %.0f (%.0f %%), events %.1f, second passes %.1f, steps %.0f, special steps %.0f" % (len(a), m[0], m[1], 100 * m[1] / m[0], m[2], m[3], m[4], m[5]))
print("cycles per step %.0f, per event %.0f, steps per event %.1f" % (m[0] / m[4], m[1] / max(m[2], 1), m[4] / max(m[2], 1)))
