"""Match-offset distribution of the C2 corpus (CPU): which share of the match loads of an LZ4 entry reaches further back than X bytes —
i.e. what a recent-output window of X bytes (LDS, or the entry's share of L2) can serve.  Walks real liblz4 frames of benchdata."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchdata import datagen as dg

def walk(block):
    p, C, out = 0, len(block), 0
    offs, mls, lls = [], [], []
    while p < C:
        tok = block[p]; p += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = block[p]; p += 1; lit += b
                if b != 255: break
        p += lit; out += lit
        if p >= C:
            lls.append(lit); break
        off = block[p] | (block[p + 1] << 8); p += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = block[p]; p += 1; ml += b
                if b != 255: break
        ml += 4
        offs.append(off); mls.append(ml); lls.append(lit); out += ml
    return np.array(offs), np.array(mls), np.array(lls)

for cls, name in ((dg.TEXT, "text"), (dg.RECORDS, "records"), (dg.RUNS, "runs")):
    allo, allm, alll = [], [], []
    for i in range(24):
        plain = dg.fill(cls, 1, i, 65536)
        f = dg.compress(dg.LZ4, 0, plain)
        f = bytes(f)
        bh = int.from_bytes(f[7:11], "little")
        if bh >> 31: continue
        o, m, l = walk(f[11:11 + (bh & 0x7FFFFFFF)])
        allo.append(o); allm.append(m); alll.append(l)
    o = np.concatenate(allo); m = np.concatenate(allm); l = np.concatenate(alll)
    print("%-8s sequences/entry %.0f  mean lit %.2f  mean match %.2f  lit<=16 %.3f  ml<=16 %.3f ml<=32 %.3f" % (name, len(o) / len(allo), l.mean(), m.mean(), (l <= 16).mean(), (m <= 16).mean(), (m <= 32).mean()))
    print("   share of matches with offset > X:", "  ".join("%dK %.3f" % (x >> 10, (o > x).mean()) for x in (1024, 2048, 4096, 8192, 16384, 24576, 32768, 49152)))
