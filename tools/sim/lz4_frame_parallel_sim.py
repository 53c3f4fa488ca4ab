#!/usr/bin/env python3
"""CPU model of a BLOCK-PARALLEL LZ4 frame executor: all sequences of a (linked-block) frame at once, sources re-pointed by pointer jumping,
then executed in dependency rounds.  Answers: how many re-pointing rounds and how many execution rounds does real data need?
  tools/sim/lz4_frame_parallel_sim.py [class] [MiB] [jump_rounds]
Sequence k: literals L_k = [o_k, o_k + ll_k), match M_k = [ms_k, ms_k + ml_k) with source a_k = ms_k - off_k; only the first
need_k = min(ml_k, off_k) source bytes are external (the rest repeats them).  A source that lies wholly inside the external part of ONE
earlier match j is re-pointed at j's own source (a_k -= ms_j - a_j); afterwards a match is READY when every byte of its source is final:
literal bytes (all written up front) or bytes of matches already executed."""
import bisect, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchdata import datagen as dg
from tools.sim.lz4_walk_sim import blocks_of
from tools.sim.lz4_lockstep_sim import tokens


def frame_sequences(frame):
    """-> list of (o, ll, ml, off) with ABSOLUTE output positions, over all compressed blocks (stored blocks become one literal run)"""
    seqs = []; base = 0
    assert frame[:4].tobytes() == b"\x04\x22\x4d\x18"
    flg = int(frame[4]); p = 7 + (8 if flg & 8 else 0) + (4 if flg & 1 else 0)
    while True:
        bs = int.from_bytes(frame[p:p + 4].tobytes(), "little"); p += 4
        if bs == 0: break
        n = bs & 0x7FFFFFFF
        if bs >> 31:
            seqs.append((base, n, 0, 0)); base += n
        else:
            toks, total = tokens(frame[p:p + n].tolist())
            for (t0, ll, ml, off, o) in toks: seqs.append((base + o, ll, ml, off))
            base += total
        p += n
        if flg & 0x10: p += 4
    return seqs, base


def run(cls, mib, jump_rounds):
    size = mib << 20
    plain = dg.fill(cls, 5, 0, size)
    frame = np.frombuffer(dg.compress(dg.LZ4, 0, plain), dtype=np.uint8)
    seqs, total = frame_sequences(frame)
    assert total == size
    n = len(seqs)
    o = [s[0] for s in seqs]; ll = [s[1] for s in seqs]; ml = [s[2] for s in seqs]; off = [s[3] for s in seqs]
    ms = [o[k] + ll[k] for k in range(n)]
    need = [min(ml[k], off[k]) for k in range(n)]
    a = [ms[k] - off[k] for k in range(n)]
    has = [ml[k] > 0 for k in range(n)]

    def piece_at(pos):
        """sequence index whose output range [o_k, o_{k+1}) holds pos"""
        return bisect.bisect_right(o, pos) - 1

    # ---- re-pointing rounds (pointer jumping: all k read the state of the round before) ----
    for r in range(jump_rounds):
        a2 = list(a); moved = 0
        for k in range(n):
            if not has[k]: continue
            j = piece_at(a[k])
            if not has[j] or j >= k: continue
            # wholly inside the external part of match j?
            if a[k] >= ms[j] and a[k] + need[k] <= ms[j] + need[j]:
                a2[k] = a[k] - (ms[j] - a[j]); moved += 1
        a = a2
        print("  jump round %d: %d sources re-pointed (%.1f %% of the matches)" % (r + 1, moved, 100.0 * moved / max(1, sum(has))))
        if moved == 0: break
    # ---- dependency sets: matches overlapped by [a_k, a_k + need_k) ----
    deps = [None] * n
    straddle = lit_only = 0
    for k in range(n):
        if not has[k]: continue
        j0 = piece_at(a[k]); j1 = piece_at(a[k] + need[k] - 1)
        d = []
        for j in range(j0, j1 + 1):
            if has[j] and j < k and a[k] < ms[j] + ml[j] and a[k] + need[k] > ms[j]: d.append(j)
        deps[k] = d
        if d: straddle += 1
        else: lit_only += 1
    nm = sum(has)
    print("  %d sequences, %d matches: %.1f %% read only final bytes after re-pointing, %.1f %% depend on earlier matches; max deps %d" % (
        n, nm, 100.0 * lit_only / nm, 100.0 * straddle / nm, max(len(d) for d in deps if d is not None)))
    # ---- execution rounds ----
    done = [not has[k] for k in range(n)]
    level = [0] * n
    for k in range(n):                      # (sequence order is a topological order: level = 1 + max level of the deps)
        if has[k]: level[k] = 1 + max([level[j] for j in deps[k]] or [0])
    hist = np.bincount(np.array([level[k] for k in range(n) if has[k]]))
    print("  execution rounds needed: %d; matches per round: %s" % (len(hist) - 1, " ".join(str(x) for x in hist[1:17]) + (" ..." if len(hist) > 17 else "")))


if __name__ == "__main__":
    cls = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    jr = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    print("class %d, %d MiB, one LZ4 frame of linked 64 KiB blocks" % (cls, mib))
    run(cls, mib, jr)


def run_bytes(cls, mib):
    """BYTE-level pointer doubling (what k_pj_* do): S[i] = literal reference or the output position the byte copies; S[i] = S[S[i]] until
    every byte points at a literal.  Rounds needed = ceil(log2(longest chain)) + 1: no dependency analysis, no levels."""
    size = mib << 20
    plain = dg.fill(cls, 5, 0, size)
    frame = np.frombuffer(dg.compress(dg.LZ4, 0, plain), dtype=np.uint8)
    # literal references are (1 << 40) + position in `frame`; match references are output positions
    S = np.zeros(size, dtype=np.int64)
    LIT = 1 << 40
    flg = int(frame[4]); p = 7 + (8 if flg & 8 else 0) + (4 if flg & 1 else 0)
    base = 0
    while True:
        bs = int.from_bytes(frame[p:p + 4].tobytes(), "little"); p += 4
        if bs == 0: break
        n = bs & 0x7FFFFFFF
        if bs >> 31:
            S[base:base + n] = LIT + p + np.arange(n); base += n
        else:
            toks, total = tokens(frame[p:p + n].tolist())
            d = frame[p:p + n]
            for (t0, ll, ml, off, o) in toks:
                tok = int(d[t0]); q = t0 + 1
                if (tok >> 4) == 15:
                    while True:
                        b = int(d[q]); q += 1
                        if b != 255: break
                S[base + o: base + o + ll] = LIT + p + q + np.arange(ll)
                if ml: S[base + o + ll: base + o + ll + ml] = base + o + ll - off + np.arange(ml)
            base += total
        p += n
        if flg & 0x10: p += 4
    assert base == size
    rounds = 0
    while True:
        m = S < LIT
        left = int(m.sum())
        if left == 0: break
        rounds += 1
        S[m] = S[S[m]]
        print("  round %2d: %9d bytes still point at output (%.2f %%)" % (rounds, left, 100.0 * left / size))
    out = frame[S - LIT]
    assert np.array_equal(out, plain)
    print("  class %d, %d MiB: %d rounds, bytes equal the plaintext" % (cls, mib, rounds))


if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[4] == "bytes":
    run_bytes(int(sys.argv[1]), int(sys.argv[2]))
