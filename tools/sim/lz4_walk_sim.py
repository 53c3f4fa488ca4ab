#!/usr/bin/env python3
"""CPU model of k_lz4_wave's chunk parse (lz4_wave.h): speculative per-lane walks with run-in, then fix-up rounds.  Counts wave
iterations (a wave iterates max-over-lanes times) under variants of the fix-up scheme.  Developer tool, no GPU.
  tools/sim/lz4_walk_sim.py [mix] [entries] [runin]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchdata import datagen as dg

SEG, NL = int(os.environ.get("SIM_SEG", "60")), 64
CHUNK = SEG * NL

def blocks_of(frame):
    assert frame[:4].tobytes() == b"\x04\x22\x4d\x18"
    flg = int(frame[4]); p = 6
    if flg & 8: p += 8
    if flg & 1: p += 4
    p += 1
    out = []
    while True:
        bs = int.from_bytes(frame[p:p+4].tobytes(), "little"); p += 4
        if bs == 0: break
        n = bs & 0x7FFFFFFF
        if not (bs >> 31): out.append(frame[p:p+n])
        p += n
        if flg & 0x10: p += 4
    return out

def hop(d, C, p):
    """-> (next, flag)  flag: 0 ok, 1 malformed, 2 last"""
    tok = d[p]; lit = tok >> 4; q = p + 1
    if lit == 15:
        while True:
            if q >= C: return C, 1
            b = d[q]; q += 1; lit += b
            if b != 255: break
    if lit > C - q: return C, 1
    q += lit
    if q == C: return C, 2
    if C - q < 2: return C, 1
    q += 2
    if (tok & 15) == 15:
        while True:
            if q >= C: return C, 1
            b = d[q]; q += 1
            if b != 255: break
    return q, 0

def walk(d, C, frm, s, e, old):
    """returns (visited set, exit, iterations, entry); old = (set, exit) or None"""
    p = frm; vis = set(); it = 0; entry = None
    while p < e:
        if p >= s:
            if entry is None: entry = p
            if old is not None and p in old[0]:
                return vis | {x for x in old[0] if x >= p}, old[1], it, entry
            vis.add(p)
        it += 1
        nx, fl = hop(d, C, p)
        if fl and p < s: nx = s
        elif fl: p = nx; break
        p = nx
    if entry is None: entry = p
    return vis, p, it, entry

def run(mix, n, runin, variant):
    b = dg.Batch(n, 65536, 65536, method=dg.LZ4, level=0, seed=1, mix=mix)
    tot = dict(chunks=0, it_first=0, it_fix=0, rounds=0, missync=0, lanes=0)
    for i in range(n):
        fr = b.archive[int(b.offsets[i]):int(b.offsets[i]) + int(b.comp_sizes[i])]
        for blk in blocks_of(fr):
            d = blk.tolist(); C = len(d); cpos = 0
            while cpos < C:
                tok_end = min(C, cpos + CHUNK)
                lanes = []
                mx = 0
                for l in range(NL):
                    s = cpos + l * SEG
                    if s >= tok_end: break
                    e = min(s + SEG, tok_end)
                    frm = s if l == 0 else max(cpos, s - runin)
                    vis, ex, it, entry = walk(d, C, frm, s, e, None)
                    lanes.append([s, e, vis, ex, entry]); mx = max(mx, it)
                tot["it_first"] += mx; tot["chunks"] += 1; tot["lanes"] += len(lanes)
                first = True
                while True:
                    tot["rounds"] += 1
                    # candidate entries
                    if variant == "neigh":
                        es = [cpos] + [lanes[k - 1][3] for k in range(1, len(lanes))]
                    elif variant == "skip":          # a chain that jumps over a whole lane enters the next one at the same position
                        es = [cpos]
                        for k in range(1, len(lanes)):
                            v = es[k - 1]
                            es.append(v if v >= lanes[k][0] else lanes[k - 1][3])
                    elif variant == "max":
                        es = [cpos]; m = cpos
                        for k in range(1, len(lanes)):
                            m = max(m, lanes[k - 1][3]); es.append(m)
                    ch = [k for k in range(len(lanes)) if es[k] != lanes[k][4]]
                    if first: tot["missync"] += len(ch); first = False
                    if not ch: break
                    mx = 0
                    for k in ch:
                        s, e, vis, ex, _ = lanes[k]
                        nv, nex, it, _ = walk(d, C, es[k], s, e, (vis, ex))
                        lanes[k] = [s, e, nv, nex, es[k]]; mx = max(mx, it)
                    tot["it_fix"] += mx
                ncp = lanes[-1][3]
                assert ncp > cpos
                cpos = ncp
    c = tot["chunks"]
    print("mix %d runin %3d %-5s: chunks %d  first-walk its/chunk %.1f  fix its/chunk %.1f  rounds/chunk %.2f  missynced lanes %.1f %%" % (
        mix, runin, variant, c, tot["it_first"] / c, tot["it_fix"] / c, tot["rounds"] / c, 100.0 * tot["missync"] / tot["lanes"]))

if __name__ == "__main__":
    mix = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    for runin in ([int(sys.argv[3])] if len(sys.argv) > 3 else [60, 120]):
        for v in ("neigh", "skip", "max"):
            run(mix, n, runin, v)


def run_table(mix, n, K):
    """Alternative scheme (not built): no run-in; every lane walks the chains from EACH of its first K positions (a walk stops where it
    meets a position already visited), so that it knows the exit for any entry among its visited positions; the true chain is then
    resolved lane by lane with table look-ups.  Reports the wave iterations (max over lanes of the hops) and how often the true entry
    is not in a lane's table (those lanes still need a walk)."""
    b = dg.Batch(n, 65536, 65536, method=dg.LZ4, level=0, seed=1, mix=mix)
    chunks = its = miss = lanes_n = extra = 0
    for i in range(n):
        fr = b.archive[int(b.offsets[i]):int(b.offsets[i]) + int(b.comp_sizes[i])]
        for blk in blocks_of(fr):
            d = blk.tolist(); C = len(d); cpos = 0
            while cpos < C:
                tok_end = min(C, cpos + CHUNK)
                tabs = []; mx = 0
                for l in range(NL):
                    s = cpos + l * SEG
                    if s >= tok_end: break
                    e = min(s + SEG, tok_end)
                    exit_of = {}; hops = 0
                    for st in range(s, min(s + K, e)):
                        path = []; p = st
                        while p < e and p not in exit_of:
                            path.append(p); hops += 1
                            nx, fl = hop(d, C, p)
                            p = nx
                            if fl: break
                        ex = exit_of[p] if p in exit_of else p
                        for q in path: exit_of[q] = ex
                    tabs.append((s, e, exit_of)); mx = max(mx, hops)
                its += mx; chunks += 1; lanes_n += len(tabs)
                # resolve the true chain
                p = cpos; fix = 0
                for (s, e, exit_of) in tabs:
                    if p >= e: continue
                    if p in exit_of: p = exit_of[p]
                    else:
                        miss += 1
                        while p < e and p not in exit_of:
                            fix += 1
                            nx, fl = hop(d, C, p); p = nx
                            if fl: break
                        if p in exit_of: p = exit_of[p]
                extra += fix
                assert p > cpos
                cpos = p
    print("mix %d table K=%2d: chunks %d  walk its/chunk %.1f  lanes whose true entry is not in the table %.2f %%  serial fix hops/chunk %.1f" % (
        mix, K, chunks, its / chunks, 100.0 * miss / lanes_n, extra / chunks))


if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[4] == "table":
    for K in (8, 16, 24):
        run_table(int(sys.argv[1]), int(sys.argv[2]), K)
