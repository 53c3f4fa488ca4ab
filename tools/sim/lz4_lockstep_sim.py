#!/usr/bin/env python3
"""CPU model of a SEGMENT-MAJOR lock-step LZ4 executor: the block is taken in chunks of 64 segments of SEG compressed bytes, lane l
decodes the tokens that START in segment l (entry positions and output positions known from a pre-pass), all 64 lanes advance together one
token step per wave iteration; a match waits until its source bytes are there (ready map).  Counts wave iterations per chunk under
scheduling variants.  Developer tool, no GPU.
  tools/sim/lz4_lockstep_sim.py [mix] [entries] [seg]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from benchdata import datagen as dg
from tools.sim.lz4_walk_sim import blocks_of

NL = 64
OWN = int(os.environ.get("SIM_OWN", "32"))      # bytes one lane copies per step


def tokens(d):
    """-> list of (in_pos, ll, ml, off, out_pos)"""
    C = len(d); p = 0; out = []; o = 0
    while p < C:
        t0 = p
        tok = int(d[p]); p += 1
        ll = tok >> 4
        if ll == 15:
            while True:
                b = int(d[p]); p += 1; ll += b
                if b != 255: break
        p += ll
        if p >= C:
            out.append((t0, ll, 0, 0, o)); o += ll; break
        off = int(d[p]) | (int(d[p + 1]) << 8); p += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = int(d[p]); p += 1; ml += b
                if b != 255: break
        ml += 4
        out.append((t0, ll, ml, off, o)); o += ll + ml
    return out, o


def sim_chunk(lanes, out_lo, total_out, ready, variant, st):
    """lanes: list of token lists.  ready: bytearray over the block's output (1 = written)."""
    idx = [0] * len(lanes); ph = [0] * len(lanes); rem = [0] * len(lanes)
    lit_idx = [0] * len(lanes)
    its = 0
    ntok = sum(len(x) for x in lanes)
    done_tok = 0
    while done_tok < ntok:
        its += 1
        marks = []
        progressed = False
        # (1) literals
        for l, tl in enumerate(lanes):
            if variant == "litahead":
                if lit_idx[l] < len(tl):
                    t0, ll, ml, off, o = tl[lit_idx[l]]
                    # one step copies up to OWN literal bytes; longer runs are counted as cooperative events
                    if ll > OWN: st["coop_lit"] += 1
                    for b in range(o, o + ll): ready[b] = 1
                    lit_idx[l] += 1
                    progressed = True
            else:
                if idx[l] < len(tl) and ph[l] == 0:
                    t0, ll, ml, off, o = tl[idx[l]]
                    if ll > OWN: st["coop_lit"] += 1
                    for b in range(o, o + ll): ready[b] = 1
                    ph[l] = 1
                    progressed = True
        # (2) matches
        for l, tl in enumerate(lanes):
            if idx[l] >= len(tl): continue
            if variant == "litahead" and lit_idx[l] <= idx[l]: continue
            if variant != "litahead" and ph[l] != 1: continue
            t0, ll, ml, off, o = tl[idx[l]]
            if ml == 0:
                idx[l] += 1; ph[l] = 0; done_tok += 1; progressed = True
                continue
            ms = o + ll
            need = min(ml, off)
            s = ms - off
            ok = True
            for b in range(max(s, out_lo), s + need):
                if not ready[b]: ok = False; break
            if ok:
                if ml > OWN or off < ml: st["coop_match"] += 1
                marks.append((ms, ms + ml))
                idx[l] += 1; ph[l] = 0; done_tok += 1; progressed = True
            else:
                st["stalls"] += 1
        for a, b in marks:
            for x in range(a, b): ready[x] = 1
        assert progressed
    return its


def run(mix, n, seg, variant):
    b = dg.Batch(n, 65536, 65536, method=dg.LZ4, level=0, seed=1, mix=mix)
    st = dict(chunks=0, its=0, tokens=0, stalls=0, coop_lit=0, coop_match=0, maxlane=0)
    for i in range(n):
        fr = b.archive[int(b.offsets[i]):int(b.offsets[i]) + int(b.comp_sizes[i])]
        for blk in blocks_of(fr):
            d = blk.tolist(); C = len(d)
            toks, total = tokens(d)
            ready = bytearray(total + 64)
            k = 0; cpos = 0
            while k < len(toks):
                cpos = toks[k][0]
                lanes = [[] for _ in range(NL)]
                while k < len(toks) and toks[k][0] < cpos + NL * seg:
                    lanes[(toks[k][0] - cpos) // seg].append(toks[k]); k += 1
                out_lo = lanes[0][0][4]
                st["chunks"] += 1; st["tokens"] += sum(len(x) for x in lanes); st["maxlane"] += max(len(x) for x in lanes)
                st["its"] += sim_chunk(lanes, out_lo, total, ready, variant, st)
    c = st["chunks"]
    print("mix %d seg %3d %-9s: chunks %d  tokens/chunk %.0f  max tokens in a lane %.1f  iterations/chunk %.1f  (ideal %.1f)  stalls/chunk %.0f  coop lit %.1f match %.1f per chunk" % (
        mix, seg, variant, c, st["tokens"] / c, st["maxlane"] / c, st["its"] / c, st["tokens"] / c / 64, st["stalls"] / c, st["coop_lit"] / c, st["coop_match"] / c))


if __name__ == "__main__":
    mix = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    seg = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    for v in ("inorder", "litahead"):
        run(mix, n, seg, v)
