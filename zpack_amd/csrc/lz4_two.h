// lz4_two.h — stage 1 of the two-stage LZ4 path: the token chains of a whole batch, ONE LANE PER ENTRY.
//
// Replaces the parse half of the LZ4F_decompress loop (lib/zpack_read.c:414-439) for large batches.
//
// Why: the token chain of an LZ4 block is serial.  k_lz4_wave (lz4_wave.h) parallelises it INSIDE an entry — 64 lanes walk 64
// segments speculatively and re-walk to a fixed point — which costs ~350 wave instructions per 64 tokens (46 % of that kernel's
// vector instructions).  A batch holds tens of thousands of independent entries: walking 64 chains side by side,
// one per lane, is the same serial code a CPU runs, ~1.5 wave instructions per token, no speculation, no fix-up rounds.  The price
// is latency (a lane takes one memory round trip per token, so a 64 KiB text entry takes ~1 ms whatever the batch size), which is
// why the path is taken only by batches large enough to fill the chip with lanes, and why stage 2 (k_lz4_exec: lz4_block_records)
// stays one WAVE per entry — the copies of an entry need the whole wave and its output window should stay cache-resident.
//
// A lane leaves 8-byte records (lz4_wave.h: lz4_block_records) in the entry's part of a record arena.  It is a PARSER ONLY: it
// checks exactly what lz4_token_at checks about the token stream (lengths that run past the block, a block that ends inside a
// sequence), follows the frame structure without verifying checksums, and on anything it does not like — skippable frames,
// blocks larger than 64 KiB (16-bit fields), dictionaries, damaged framing, entries too long for one lane — it marks the entry
// LZ2_FALLBACK: the general decoder decodes that entry from scratch and alone gives verdicts.  Offsets, output bounds, checksums
// and the XXH3 are stage 2's business.
#pragma once
#include "lz4_wave.h"
#include "lx_ring.h"

namespace zpk {

#define LZ2_FALLBACK 0x80000000u
struct Lz2Info { u32 rec_base; u32 nrec; };          // per work-list slot: first record (arena index), count | LZ2_FALLBACK

enum { LZ2_S_HDR = 0, LZ2_S_TOKEN = 1, LZ2_S_OFFSET = 2, LZ2_S_LITEXT = 3, LZ2_S_MLEXT = 4, LZ2_S_DONE = 5 };

// The compressed bytes reach a lane through its own LDS RING.  (First version: one 16-byte global load at the token per step.  With
// 6 waves x 64 lanes per CU each on its own 128-byte line the 32 KiB L1 holds nothing: every step fetched a line from L2, 63 GB for
// 3.5 GB of input, 13 ms.)  A lane owns LZ2_RING contiguous bytes (+ a 16-byte mirror of its start, so that the three bytes behind any
// position are contiguous); the stream is fetched in 64-byte groups, each byte once: every LZ2_EVERY steps the wave waits for the
// groups requested at the PREVIOUS refill, writes them to LDS and requests the next ones — a memory round trip is covered by
// LZ2_EVERY steps of parsing.  A lane whose next bytes have not landed yet skips the step.
#define LZ2_RING 256u
#define LZ2_LANE_STRIDE (LZ2_RING + 16u)
#define LZ2_EVERY 4u
struct alignas(16) Lz2ParseShared { u8 ring[64 * LZ2_LANE_STRIDE]; };

// slot = this lane's work-list slot (64 consecutive slots per wave)
__device__ inline void lz4_parse_lanes(Lz2ParseShared& sh, const u8* __restrict__ src, const u8* read_lo, const u8* read_hi,
                                       const zpk_decode_desc* __restrict__ desc, const u32* __restrict__ list, u32 nslots,
                                       u64* __restrict__ arena, u64 arena_recs, unsigned long long* __restrict__ bump,
                                       Lz2Info* __restrict__ info, u32 max_comp, u32 slot, int lane)
{
    const bool have = slot < nslots;
    u64 src_off = 0, csz = 0;
    if (have) { const u32 e = list[slot]; src_off = desc[e].src_offset; csz = desc[e].comp_size; }
    const u8* const in = src + src_off;
    // the stream is addressed from the 64-byte boundary at or below the entry: position s = address - org.  Every 64-byte group of
    // [org, end of the entry) has to be readable as a whole: the (at most one) entry of an image that is not goes the general way.
    const u8* const org = (const u8*)((u64)in & ~(u64)63);
    const u32 s0 = (u32)(in - org);
    // ---- frame header: the checks of lz4f_decode_wave; whatever is unusual goes to the general decoder ----
    bool ok = have && csz >= 11 && csz <= (u64)max_comp && org >= read_lo && in + csz <= read_hi &&
              (u64)(read_hi - org) >= (((u64)s0 + csz + 63) & ~(u64)63);
    u32 hdr = 0, bck_bytes = 0;
    if (ok) {
        const u32 magic = ld32(in);
        const u32 fb = ld16(in + 4), flg = fb & 0xFFu, bd = fb >> 8;
        ok = magic == 0x184D2204u && (flg >> 6) == 1 && !(flg & 2) && !(bd & 0x8F) && ((bd >> 4) & 7) == 4;      // 64 KiB blocks only
        bck_bytes = (flg >> 4) & 1 ? 4u : 0u;
        hdr = 7 + ((flg >> 3) & 1 ? 8u : 0u) + (flg & 1 ? 4u : 0u);
        ok = ok && csz >= hdr + 4;
    }
    // ---- record space.  A sequence with a match takes at least 3 bytes of the block, a block's last sequence at least 1 byte
    // behind a 4-byte block header, the frame header 7: fewer than comp_size / 3 records.  One atomic per wave. ----
    const u32 cap = ok ? (u32)(csz / 3) + 2 : 0u;
    const u32 incl = wave_scan_add(cap);
    const u32 wave_total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
    unsigned long long wave_base = 0;
    lane0_guard();
    if (lane == 0 && wave_total) wave_base = atomicAdd(bump, (unsigned long long)wave_total);
    wave_base = uni64(wave_base);
    lane0_guard();
    const u64 my_base = wave_base + (incl - cap);
    if (ok && my_base + cap > arena_recs) ok = false;                      // (overlapping entries of a crafted archive can ask for more than the arena)
    u64* const rp = arena + my_base;

    // ---- the walk.  All positions are stream positions (relative to org). ----
    ZPK_LDS u8* const ring = (ZPK_LDS u8*)sh.ring + (u32)lane * LZ2_LANE_STRIDE;
    u32 state = ok ? (u32)LZ2_S_HDR : (u32)LZ2_S_DONE;
    bool fb = have && !ok;
    const u32 s_end = s0 + (u32)csz;         // end of the entry
    const u32 rq_max = (s_end + 63u) & ~63u;
    u32 s = s0 + hdr;                        // S_HDR: the next block header; otherwise: the position the state reads at
    u32 bend = 0;                            // end of the current block
    u32 nrec = 0;
    u32 lit_pos = 0, lit = 0, mlc = 0, moff = 0, acc = 0, bstart = 0;
    u32 rq = s & ~63u, rdy = rq;             // groups requested / landed up to here (bytes below rq - LZ2_RING are overwritten)
    v4u32 pend0 = {0, 0, 0, 0}, pend1 = pend0, pend2 = pend0, pend3 = pend0;      // the group in flight
    u32 pend_at = 0xFFFFFFFFu;               // its stream position (none)
    const u32 max_steps = 8 * max_comp + 4096u;      // (a lane makes progress at least every second refill: not reached)
    for (u32 step = 0; ; step++) {
        if (__builtin_amdgcn_ballot_w64(state != LZ2_S_DONE) == 0) break;
        if (step > max_steps) { if (state != LZ2_S_DONE) { fb = true; state = LZ2_S_DONE; } break; }
        if (step % LZ2_EVERY == 0) {
            // ---- refill: land the group in flight, request the next ----
            if (pend_at != 0xFFFFFFFFu) {
                ZPK_LDS u8* const d = ring + (pend_at & (LZ2_RING - 1));
                *(ZPK_LDS v4u32*)(d) = pend0; *(ZPK_LDS v4u32*)(d + 16) = pend1; *(ZPK_LDS v4u32*)(d + 32) = pend2; *(ZPK_LDS v4u32*)(d + 48) = pend3;
                if ((pend_at & (LZ2_RING - 1)) == 0) *(ZPK_LDS v4u32*)(ring + LZ2_RING) = pend0;         // the mirror
                rdy = pend_at + 64;
                pend_at = 0xFFFFFFFFu;
            }
            const bool live = state != LZ2_S_DONE;
            if (live && s >= rq + 64) { rq = s & ~63u; rdy = rq; }          // a long literal run was skipped: nothing in between is wanted
            if (live && rq < rq_max && rq + 64 <= (s & ~63u) + LZ2_RING) {
                const u8* const g = org + rq;
                pend0 = *(const ZPK_GLOBAL v4u32*)(g); pend1 = *(const ZPK_GLOBAL v4u32*)(g + 16);
                pend2 = *(const ZPK_GLOBAL v4u32*)(g + 32); pend3 = *(const ZPK_GLOBAL v4u32*)(g + 48);
                pend_at = rq; rq += 64;
            }
        }
        // bytes the state is going to look at: [s, s + need)
        const u32 st0 = state;
        const u32 need = st0 == LZ2_S_HDR ? 4u : st0 == LZ2_S_TOKEN ? 2u : st0 == LZ2_S_OFFSET ? 3u : 1u;
        const u32 lim = rdy >= rq_max ? 0xFFFFFFF0u : rdy;                    // everything there is has landed
        const bool go = st0 != LZ2_S_DONE && s + need <= lim;
        if (go) {
            ZPK_LDS const u8* const at = ring + (s & (LZ2_RING - 1));
            const u32 B0 = at[0], B1 = at[1], B2 = at[2];
            if (st0 == LZ2_S_HDR) {              // block header (rare, divergent)
                if (s_end - s < 4) { fb = true; state = LZ2_S_DONE; }
                else {
                    const u32 bh = B0 | (B1 << 8) | (B2 << 16) | ((u32)at[3] << 24);
                    if (bh == 0) state = LZ2_S_DONE;                                    // EndMark: what follows is stage 2's to check
                    else {
                        const u32 bsz = bh & 0x7FFFFFFFu;
                        if (bsz > 65536u || bsz == 0 || (u64)bsz + bck_bytes > (u64)(s_end - s - 4)) { fb = true; state = LZ2_S_DONE; }
                        else if (bh >> 31) s += 4 + bsz + bck_bytes;                    // stored block: no records
                        else { s += 4; bstart = s; bend = s + bsz; state = LZ2_S_TOKEN; }
                    }
                }
            } else {
                bool t1 = false, t2 = false, emit = false, fail = false;
                u32 q = 0, x = 0, ml = 0;
                if (st0 == LZ2_S_TOKEN) {
                    mlc = B0 & 15u; lit = B0 >> 4; q = s + 1;
                    if (s >= bend) fail = true;                      // the chain ran off the block without a last literal run
                    else if (lit == 15) {
                        if (q >= bend) fail = true;
                        else { lit += B1; q++; if (B1 == 255) { state = LZ2_S_LITEXT; acc = lit; s = q; } else t1 = true; }
                    } else t1 = true;
                } else if (st0 == LZ2_S_LITEXT) {          // one more length byte of a literal run >= 270
                    if (s >= bend) fail = true;
                    else { acc += B0; s++; if (B0 != 255) { lit = acc; q = s; t1 = true; } }
                } else if (st0 == LZ2_S_OFFSET) {          // the offset field had not landed (or lies behind a long literal run)
                    q = s; x = B0 | (B1 << 8) | (B2 << 16); t2 = true;
                } else {                                    // LZ2_S_MLEXT: one more length byte of a match >= 274
                    if (s >= bend) fail = true;
                    else { acc += B0; s++; if (B0 != 255) { ml = acc; q = s; emit = true; } }
                }
                if (t1) {                                   // literal run [q, q + lit)
                    lit_pos = q - bstart;
                    if (lit > bend - q) fail = true;
                    else {
                        q += lit;
                        if (q == bend) {                    // the block's last sequence
                            if (nrec >= cap || lit_pos > 0xFFFFu) fail = true;
                            else { rp[nrec++] = ((u64)lit << 32) | ((u64)lit_pos << 16); s = q + bck_bytes; state = LZ2_S_HDR; }
                        }
                        else if (bend - q < 2) fail = true;
                        else if (q + 3 <= lim && q + 3 <= s + LZ2_RING / 2) {           // the offset field is there already
                            ZPK_LDS const u8* const aq = ring + (q & (LZ2_RING - 1));
                            x = (u32)aq[0] | ((u32)aq[1] << 8) | ((u32)aq[2] << 16); t2 = true;
                        }
                        else { state = LZ2_S_OFFSET; s = q; }
                    }
                }
                if (t2) {                                   // x = offset (2 bytes) + the first match-length extension byte, if any
                    moff = x & 0xFFFFu; q += 2; ml = mlc;
                    if (mlc == 15) {
                        if (q >= bend) fail = true;
                        else { const u32 b2 = (x >> 16) & 0xFFu; ml += b2; q++; if (b2 == 255) { state = LZ2_S_MLEXT; acc = ml; s = q; } else emit = true; }
                    } else emit = true;
                }
                if (emit && !fail) {
                    ml += 4;
                    if (nrec >= cap || ml > 0xFFFFu) fail = true;
                    else { rp[nrec++] = ((u64)ml << 48) | ((u64)lit << 32) | ((u64)lit_pos << 16) | moff; s = q; state = LZ2_S_TOKEN; }
                }
                if (fail) { fb = true; state = LZ2_S_DONE; }
            }
        }
    }
    if (have) { Lz2Info inf; inf.rec_base = (u32)my_base; inf.nrec = fb ? LZ2_FALLBACK : nrec; info[slot] = inf; }
}


// ================================================================================================================================
// Stage 2 over the entry's OUTPUT SLOT (k_lz4_exec_g): the frames the reference writer produces and their plain variations (any block
// count, stored blocks, linked or independent blocks), compressed blocks executed from the records by lz4_block_records (lz4_wave.h:
// seq_exec_batch, window = the slot in memory).  Checksummed, sized or dictionary frames, skippable frames, several frames in one
// entry and anything malformed are the general decoder's (rc != D_OK).  Its own small frame walker rather than lz4f_decode_wave's:
// the general walker's resume / multi-frame state cost this kernel its registers (92 bytes of scratch in the batch loop: 3.5 x slower).
__device__ inline DecodeOut lz4f_slot_decode_wave(Lz4WaveShared& sh, Watchdog& wd, SeqStats& stt, const u8* src, u64 src_size, const u8* src_hi,
                                                  u8* dst, u64 dst_cap, int lane, Lz2Cursor& cur)
{
    DecodeOut r; r.rc = D_MALFORMED; r.produced = 0;
    if (src_size < 11) return r;
    const u8* ip = src;
    const u8* const iend = src + src_size;
    u8* op = dst;
    u8* const oend = dst + dst_cap;
    if (uld32(ip) != 0x184D2204u) return r;
    const u32 flg = uld8(ip + 4), bd = uld8(ip + 5);
    if ((flg != 0x40u && flg != 0x60u) || bd != 0x40u) return r;
    {
        u32 h = 0;
        lane0_guard();
        if (lane == 0) h = (xxh32_serial(ip + 4, 2, 0) >> 8) & 0xFF;
        if (uni(h) != uld8(ip + 6)) return r;
    }
    const bool indep = (flg >> 5) & 1;
    ip += 7;
    for (;;) {
        if (wd.expired()) return r;
        if (iend - ip < 4) return r;
        const u32 bh = uld32(ip);
        ip += 4;
        if (bh == 0) break;
        const u64 bsz = bh & 0x7FFFFFFFu;
        if (bsz > 65536u || (u64)(iend - ip) < bsz) return r;
        if (bh >> 31) {
            if (bsz > (u64)(oend - op)) return r;
            for (u64 i = (u64)lane * 16; i < bsz; i += WAVE * 16) gcopy_upto16(op + i, ip + i, (u32)(bsz - i < 16 ? bsz - i : 16));
            op += bsz;
        } else {
            u8* hist_lo = indep ? op : dst;
            if ((u64)(op - hist_lo) > 65536) hist_lo = op - 65536;
            u8* bend = oend;
            if ((u64)(oend - op) > 65536u) bend = op + 65536u;
            if (lz4_block_records(sh, wd, stt, ip, (u32)bsz, src_hi, hist_lo, op, bend, lane, cur) != D_OK) return r;
        }
        wave_mem_fence();
        ip += bsz;
    }
    // (more frames behind this one while output space is left: the reference's loop decodes on — the general decoder's)
    if (ip != iend && op != oend) return r;
    r.rc = D_OK; r.produced = (u64)(op - dst);
    return r;
}

// ================================================================================================================================
// Stage 2 with an LDS OUTPUT WINDOW (k_lz4_exec).
//
// What bounds the execution of LZ4 sequences on this part is not instruction issue but LINE FETCHES (profiles/r04: the record-driven
// executor over the entry's output slot runs 39 % fewer vector instructions than k_lz4_wave and is 11 % faster; its 3.2e8 L2 misses per
// 100 000 entries — one 128-byte line per far match source for <= 16 wanted bytes, 2.2e8 of them — move 5.8 TB/s through the fabric,
// the practical HBM ceiling; halving the entries in flight costs 29 %, not 100 %).  A resident entry gets ~4 KiB of its XCD's L2,
// shared with the streams, so only sources within ~1.5 KiB hit; but 71 % of a text entry's matches lie within 4 KiB and 85 % within
// 8 KiB (tools/sim/lz4_offset_dist.py).  So the last LZ2_WIN bytes of an entry's output live in the wave's LDS (the ring of
// lx_ring.h: assembled there, whole 1 KiB lines leave with one 16-byte store per lane and feed the XXH3 on the way out — no hash
// pass, no re-read), match sources inside it never touch memory, and the compressed bytes need no stage: a literal run is one
// 16-byte global load, 64 lanes of a batch read ~450 consecutive bytes.
// The copies are seq_exec_batch's LDS-assembly form (one or two 16-byte accesses per piece at any byte address, exact-length stores)
// rather than lx_exec_batch's aligned OR-merge: ~2/3 of the vector instructions, more LDS cycles — the right trade for a kernel that
// no longer parses.  Ring bytes at and beyond the write position stay ZERO (lx_slide), so lx_append_raw / lx_append_match — used
// for stored blocks and for the rare piece of more than LZ2_OWN_MAX bytes or a self-overlapping match — work on the same ring.
#ifndef LZ2_WIN
#define LZ2_WIN 6144u
#endif
#ifndef LZ2_WIN_HIST
#define LZ2_WIN_HIST 3072u
#endif
#define LZ2_OWN_MAX SEQ_OWN_MAX
typedef LxOutT<LZ2_WIN, LZ2_WIN_HIST> Lz2Out;
struct alignas(16) Lz2ExecShared { u8 ring[LZ2_WIN + 32]; u8 secret[192]; };

// 16 / 32 bytes at any byte address of the ring, or of the entry's output slot in memory for what has left the ring
__device__ __forceinline__ Copy32 lz2_src_load(const Lz2Out& O, u32 sabs, u32 n, u64 dst_cap)
{
    Copy32 c; c.lo.lo = c.lo.hi = c.hi.lo = c.hi.hi = 0;
    if (n == 0) return c;
    if (sabs >= O.rb) {
        const lds_cp8 sp = (lds_cp8)(O.ring + (sabs - O.rb));
        c.lo = lds_ld128(sp);
        if (n > 16) c.hi = lds_ld128(sp + (n - 16));
    } else c = gload_wide32(O.dst + sabs, n, (u64)sabs + 16 <= dst_cap);
    return c;
}

// `cnt` sequences (lane k < cnt: ll literal bytes at `lit`, then ml bytes from distance off; all <= LZ2_OWN_MAX, no match overlaps
// itself) at the ring's write position.  cnt may come back smaller (a batch that does not fit behind a slide is cut).
__device__ __forceinline__ int lz2_exec_batch(Lz2Out& O, u32& cnt, u32 ll, u32 ml, u32 off, const u8* lit, const u8* rd_hi, u32 hist_lo,
                                              u64 dst_cap, int lane, SeqStats& stt)
{
    bool act = (u32)lane < cnt;
    if (!act) { ll = 0; ml = 0; }
    u32 x = wave_scan_add(ll + ml);
    u32 total = (u32)__builtin_amdgcn_readlane((int)x, 63);
    if (O.wp + total > O.rb + Lz2Out::RING) {
        lx_slide(O, lane);
        const u32 free_ = O.rb + Lz2Out::RING - O.wp;
        if (total > free_) {
            const u32 c2 = (u32)__popcll(__ballot(act && x <= free_));
            if (c2 == 0) return LX_E_FIT;
            cnt = c2; act = (u32)lane < cnt;
            if (!act) { ll = 0; ml = 0; }
            x = wave_scan_add(ll + ml);
            total = (u32)__builtin_amdgcn_readlane((int)x, 63);
        }
    }
    if ((u64)O.wp + total > dst_cap) return LX_E_CAPACITY;
    const u32 o = O.wp + (x - ll - ml), ms = o + ll;
    const bool has_match = act && ml != 0;
    if (__ballot(has_match && (off == 0 || off > ms - hist_lo)) != 0) return LX_E_OFFSET;
    // ---- in-batch dependencies (positions relative to wp) ----
    u64 pending = __ballot(has_match);
    const u32 r_ms = ms - O.wp, r_me = r_ms + ml;
    i64 srel;
    const u64 need = seq_dependencies<i32>(has_match, r_ms, r_me, off, ml, pending, lane, srel, stt);
    const u32 sabs = (u32)((i64)O.wp + srel);                  // (possibly re-pointed) source
    const bool early = has_match && srel + (i64)ml <= 0;
    // ---- literals + matches whose source is older than the batch: all loads before the first store ----
    Copy32 ca; ca.lo.lo = ca.lo.hi = ca.hi.lo = ca.hi.hi = 0;
    if (ll) {
        if (lit + 16 <= rd_hi) { ca.lo = ld128(lit); if (ll > 16) ca.hi = ld128(lit + (ll - 16)); }
        else ca = gload_wide32(lit, ll, false);
    }
    const Copy32 cb = lz2_src_load(O, sabs, early ? ml : 0u, dst_cap);
    const lds_p8 ring = O.ring;
    lds_store_wide32(ring + (o - O.rb), ca, ll);
    lds_store_wide32(ring + (ms - O.rb), cb, early ? ml : 0u);
    wave_mem_fence();
    // ---- rounds: matches that read this batch's own output ----
    u64 done = ~pending | __ballot(early);
    pending &= ~done;
    u32 guard = 0;
    while (pending) {
        const bool ready = has_match && ((pending >> lane) & 1) && (need & ~done) == 0;
        const u64 rmask = __ballot(ready);
        if (rmask == 0 || ++guard > 70) return LX_E_ROUNDS;
        const Copy32 c = lz2_src_load(O, sabs, ready ? ml : 0u, dst_cap);
        lds_store_wide32(ring + (ms - O.rb), c, ready ? ml : 0u);
        wave_mem_fence();
        done |= rmask;
        pending &= ~rmask;
    }
    O.wp += total;
    lx_flush_blocks(O, lane);
    return LX_OK;
}

// one compressed block from its records (see lz4_block_records): returns LX_OK or an anomaly code (the caller falls back)
__device__ inline int lz2_block_window(Lz2Out& O, Watchdog& wd, SeqStats& stt, const u8* ip, u32 C, const u8* rd_hi, u32 hist_lo,
                                       u64 dst_cap, int lane, Lz2Cursor& cur)
{
    if (C == 0 || C > 65536u) return LX_E_FRAME;
    u64 r = (u32)lane < cur.left ? *(const ZPK_GLOBAL u64*)(cur.rec + lane) : 0ull;
    for (;;) {
        if (wd.expired()) return LX_E_FRAME;
        if (cur.left == 0) return LX_E_LIST;
        const u32 avail = cur.left < (u32)WAVE ? cur.left : (u32)WAVE;
        const u32 off = (u32)r & 0xFFFFu, lp = ((u32)r >> 16), ll = (u32)(r >> 32) & 0xFFFFu, ml = (u32)(r >> 48);
        const bool in_list = (u32)lane < avail;
        const u64 endm = __ballot(in_list && ml == 0);
        u32 cnt = endm ? (u32)__ffsll((long long)endm) : avail;              // up to and including the block's last sequence
        if (__ballot((u32)lane < cnt && lp + ll > C) != 0) return LX_E_TOKEN;
        // pieces the batch form does not take: the sequences in front of the first such one go as a batch, then it goes alone
        const u64 longm = __ballot((u32)lane < cnt && (ll > LZ2_OWN_MAX || ml > LZ2_OWN_MAX || (ml != 0 && off < ml)));
        const u32 first_long = longm ? (u32)__ffsll((long long)longm) - 1u : cnt;
        u32 taken;
        bool last_done;
        if (first_long > 0) {
            taken = first_long;
            const int rc = lz2_exec_batch(O, taken, ll, ml, off, ip + lp, rd_hi, hist_lo, dst_cap, lane, stt);
            if (rc != LX_OK) return rc;
            last_done = endm != 0 && taken == cnt;
        } else {
            const u32 ll0 = (u32)__builtin_amdgcn_readfirstlane((int)ll), ml0 = (u32)__builtin_amdgcn_readfirstlane((int)ml);
            const u32 off0 = (u32)__builtin_amdgcn_readfirstlane((int)off), lp0 = (u32)__builtin_amdgcn_readfirstlane((int)lp);
            int rc = lx_append_raw(O, ip + lp0, ll0, rd_hi, dst_cap, lane);
            if (rc != LX_OK) return rc;
            if (ml0) { rc = lx_append_match(O, off0, ml0, hist_lo, dst_cap, lane); if (rc != LX_OK) return rc; }
            taken = 1;
            last_done = ml0 == 0;
        }
        const u32 last_end = (u32)__builtin_amdgcn_readlane((int)(lp + ll), (int)taken - 1);
        cur.rec += taken; cur.left -= taken;
        if (last_done) { if (last_end != C) return LX_E_TOKEN; break; }
        r = (u32)lane < cur.left ? *(const ZPK_GLOBAL u64*)(cur.rec + lane) : 0ull;
    }
    return LX_OK;
}

// whole frame through the window.  Takes the frames the reference writer produces and their plain variations (any block count,
// stored blocks, linked or independent blocks); checksummed, sized or dictionary frames, skippable frames and anything malformed
// are left to the general decoder (rc != LX_OK).
__device__ inline LxResult lz4f_window_decode_wave(Lz2ExecShared& sh, Watchdog& wd, SeqStats& stt, const u8* src, u64 src_size, const u8* src_hi,
                                                   u8* dst, u64 dst_cap, u64 uncomp_size, int lane, Lz2Cursor& cur)
{
    LxResult R; R.rc = LX_E_FRAME; R.produced = 0; R.hash = 0;
    if (src_size < 11 || dst_cap >= (1ull << 31) || uncomp_size >= (1ull << 31)) return R;
    const u8* ip = src;
    const u8* const iend = src + src_size;
    if (uld32(ip) != 0x184D2204u) return R;
    const u32 flg = uld8(ip + 4), bd = uld8(ip + 5);
    if (flg != 0x40u && flg != 0x60u) return R;                 // version 01, linked or independent blocks, nothing optional
    if (bd != 0x40u) return R;                                  // 64 KiB blocks
    {
        u32 h = 0;
        lane0_guard();
        if (lane == 0) h = (xxh32_serial(ip + 4, 2, 0) >> 8) & 0xFF;
        if (uni(h) != uld8(ip + 6)) return R;
    }
    const bool indep = (flg >> 5) & 1;
    ip += 7;
    Lz2Out O;
    lx_begin(O, to_lds_rw(sh.ring), dst, uncomp_size, lane, to_lds_rw(sh.secret));
    for (;;) {
        if (wd.expired()) return R;
        if (iend - ip < 4) return R;
        const u32 bh = uld32(ip);
        ip += 4;
        if (bh == 0) break;
        const u32 bsz = bh & 0x7FFFFFFFu;
        if (bsz > 65536u || (u64)(iend - ip) < bsz) return R;
        const u32 block_out = O.wp;
        int rc;
        if (bh >> 31) rc = lx_append_raw(O, ip, bsz, src_hi, dst_cap, lane);
        else {
            u32 hist_lo = indep ? block_out : 0u;
            if (block_out - hist_lo > 65536u) hist_lo = block_out - 65536u;
            rc = lz2_block_window(O, wd, stt, ip, bsz, src_hi, hist_lo, dst_cap, lane, cur);
        }
        if (rc != LX_OK) { R.rc = rc; return R; }
        if (O.wp - block_out > 65536u) { R.rc = LX_E_BLOCKMAX; return R; }
        ip += bsz;
    }
    if (cur.left != 0) { R.rc = LX_E_LIST; return R; }
    // the reference's LZ4F_decompress loop goes on with the next frame while input AND output space are left (lib/zpack_read.c:414):
    // an entry of several frames is the general decoder's (found by tools/fuzz_gpu.py: this path used to stop behind the first frame)
    if (ip != iend && (u64)O.wp != dst_cap) { R.rc = LX_E_FRAME; return R; }
    lx_finish(O, dst, uncomp_size, R, lane);
    R.rc = LX_OK;
    return R;
}

}  // namespace zpk
