#!/usr/bin/env python3
"""CPU model of the batch executor's in-batch dependency structure (seq_exec.h): for batches of B consecutive LZ4 sequences, how many
matches read bytes produced inside their own batch, and how many copy rounds different scheduling policies need.  Developer tool.
  tools/sim/lz4_exec_sim.py [mix] [entries] [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchdata import datagen as dg
from tools.sim.lz4_walk_sim import blocks_of


def sequences(d):
    """-> list of (ll, ml, off) of one block (the last sequence has ml = 0)"""
    C = len(d); p = 0; out = []
    while p < C:
        tok = int(d[p]); p += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = int(d[p]); p += 1; ll += b
                if b != 255: break
        p += ll
        if p >= C:
            out.append((ll, 0, 0)); break
        off = int(d[p]) | (int(d[p + 1]) << 8); p += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = int(d[p]); p += 1; ml += b
                if b != 255: break
        out.append((ll, ml + 4, off))
    return out


def batch_stats(seqs, B, st):
    n = len(seqs)
    for b0 in range(0, n, B):
        bs = seqs[b0:b0 + B]
        pos = 0
        recs = []       # (lit_start, ms, me, src_start, need_len, off)
        for ll, ml, off in bs:
            ms = pos + ll; me = ms + ml
            recs.append((pos, ms, me, ms - off, min(ml, off), off, ml))
            pos = me
        st["batches"] += 1; st["seqs"] += len(bs); st["bytes"] += pos
        # classify
        pend = []
        for k, (ls, ms, me, s, nl, off, ml) in enumerate(recs):
            if ml == 0: continue
            st["matches"] += 1
            if ml > 32 or ml > nl: st["coop"] += 1
            if s + nl <= 0: st["early"] += 1
            else: pend.append(k)
        st["pending"] += len(pend)
        st["pend_hist"][min(len(pend), 64)] += 1
        # exact dependency sets among pending matches: match outputs of pending lanes intersecting the source
        pset = set(pend)
        deps = {}
        inside_lit = 0
        for k in pend:
            ls, ms, me, s, nl, off, ml = recs[k]
            e = s + nl
            dk = [j for j in pend if j < k and recs[j][1] < e and recs[j][2] > s]
            deps[k] = dk
            if not dk: inside_lit += 1
        st["pend_nodep"] += inside_lit
        # rounds with exact deps, no re-pointing: level = 1 + max(level of deps)
        lvl = {}
        for k in pend:
            lvl[k] = 1 + max([lvl[j] for j in deps[k]], default=0)
        r_exact = max(lvl.values(), default=0)
        st["rounds_exact"] += r_exact
        st["rounds_exact_hist"][min(r_exact, 15)] += 1
        # full re-pointing (sequential chain collapse): a source wholly inside the externally-sourced part of ONE earlier pending match
        # is re-pointed at that match's (already re-pointed) source, inheriting its dependency set
        src = {k: recs[k][3] for k in pend}
        dep2 = {}
        lvl2 = {}
        early2 = 0
        for k in pend:
            ls, ms, me, s, nl, off, ml = recs[k]
            cur_s = s
            d = None
            # walk containment repeatedly in lane order (sequential processing gives full collapse)
            changed = True
            while changed:
                changed = False
                e = cur_s + nl
                if e <= 0: break
                cand = [j for j in pend if j < k and recs[j][1] < e and recs[j][2] > cur_s]
                if len(cand) == 1:
                    j = cand[0]
                    jms = recs[j][1]; jnl = recs[j][4]
                    if jms <= cur_s and e <= jms + jnl:
                        cur_s = cur_s - (jms - src[j]); changed = True
            src[k] = cur_s
            e = cur_s + nl
            if e <= 0:
                early2 += 1; dep2[k] = []; lvl2[k] = 0 if True else 1
                lvl2[k] = 1        # it still executes in the first dependent round at the latest; it could go with the early ones
                lvl2[k] = 0
            else:
                dk = [j for j in pend if j < k and recs[j][1] < e and recs[j][2] > cur_s]
                dep2[k] = dk
                lvl2[k] = 1 + max([lvl2[j] for j in dk], default=0)
        r_jump = max(lvl2.values(), default=0)
        st["rounds_jump"] += r_jump
        st["jump_early"] += early2
        st["rounds_jump_hist"][min(r_jump, 15)] += 1
        # watermark policy: a lane is ready when every pending lane whose output starts before its source end is done
        done = set(); r = 0
        rem = list(pend)
        while rem:
            r += 1
            w = recs[rem[0]][1]            # output start of the first undone pending match: everything before is final
            go = [k for k in rem if recs[k][3] + recs[k][4] <= w or k == rem[0]]
            rem = [k for k in rem if k not in go]
        st["rounds_water"] += r


def main():
    mix = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    b = dg.Batch(n, 65536, method=dg.LZ4, level=0, seed=1, mix=mix, threads=4)
    st = dict(batches=0, seqs=0, bytes=0, matches=0, early=0, pending=0, coop=0, pend_nodep=0, rounds_exact=0, rounds_jump=0, rounds_water=0,
              jump_early=0, pend_hist=np.zeros(65, int), rounds_exact_hist=np.zeros(16, int), rounds_jump_hist=np.zeros(16, int))
    lls = []; mls = []; offs = []
    for i in range(b.n):
        fr = b.archive[int(b.offsets[i]):int(b.offsets[i] + b.comp_sizes[i])]
        for blk in blocks_of(fr):
            s = sequences(blk)
            batch_stats(s, B, st)
            lls += [x[0] for x in s]; mls += [x[1] for x in s if x[1]]; offs += [x[2] for x in s if x[1]]
    nb = st["batches"]
    lls = np.array(lls); mls = np.array(mls); offs = np.array(offs)
    print("mix %d, %d entries, batch %d: %d batches, %.1f seqs/batch, %.0f out bytes/batch" % (mix, b.n, B, nb, st["seqs"] / nb, st["bytes"] / nb))
    print("  ll: mean %.2f  P(0) %.2f P(<=4) %.2f P(<=8) %.2f P(<=16) %.2f P(>32) %.3f" % (lls.mean(), (lls == 0).mean(), (lls <= 4).mean(), (lls <= 8).mean(), (lls <= 16).mean(), (lls > 32).mean()))
    print("  ml: mean %.2f  P(<=8) %.2f P(<=16) %.2f P(<=18) %.2f P(<=32) %.2f" % (mls.mean(), (mls <= 8).mean(), (mls <= 16).mean(), (mls <= 18).mean(), (mls <= 32).mean()))
    print("  ll+ml<=16 %.2f  <=32 %.2f ; off: P(<16) %.3f P(<64) %.2f P(<1024) %.2f P(<4096) %.2f" % (
        0, 0, (offs < 16).mean(), (offs < 64).mean(), (offs < 1024).mean(), (offs < 4096).mean()))
    print("  matches/batch %.1f: early %.1f, pending (source inside the batch) %.1f [of these %.1f touch no pending match], long/self-overlapping %.2f" % (
        st["matches"] / nb, st["early"] / nb, st["pending"] / nb, st["pend_nodep"] / nb, st["coop"] / nb))
    print("  rounds per batch: exact deps %.2f | exact + full re-pointing %.2f (re-pointed to early: %.1f/batch) | watermark %.2f" % (
        st["rounds_exact"] / nb, st["rounds_jump"] / nb, st["jump_early"] / nb, st["rounds_water"] / nb))
    print("  pending histogram (0..64, by 8):", [int(st["pend_hist"][i:i + 8].sum()) for i in range(0, 65, 8)])
    print("  rounds exact hist:", st["rounds_exact_hist"].tolist())
    print("  rounds jump  hist:", st["rounds_jump_hist"].tolist())


if __name__ == "__main__":
    main()
