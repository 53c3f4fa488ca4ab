// lz4_ring.h — the fast LZ4 frame path: a lane-per-entry TOKEN SCAN, then a wave-per-entry executor that assembles the
// whole output in an LDS ring.
//
// Replaces the LZ4F_decompress loop of the reference (lib/zpack_read.c:414-439) for the frames the reference's own writer
// emits (lib/zpack_write.c:199-211: 64 KiB blocks, no checksums, no content size) with sequences of at most 1 KiB per
// literal run / match; every other frame, and every entry on which this path meets ANYTHING it does not like (a malformed
// token, a bad offset, an output slot that is too small, a list that does not verify), is left to the general decoder
// (lz4_wave.h), which then gives the verdict — so the status, detail and bytes of anything that is not a clean decode are
// always the general decoder's.
//
// Why this shape (measured, MI355X, round 2): the general decoder was bound by the SCALAR unit — one per CU, ~1 instruction
// per cycle shared by 28 resident waves — with 47.6 k SALU + 57.6 k VALU instructions per 64 KiB entry (exec-mask juggling of
// divergent copies, speculative token walks with fix-up rounds, ballot loops).  Here
//   * the serial token chain is walked by ONE LANE PER ENTRY (k_lz4_scan: 64 entries per wave instruction, ~0.6
//     instructions per token instead of ~6): it leaves the block-relative position of every token, 2 bytes each, in a scratch
//     region laid out like the source (3/4 of the compressed size bounds it);
//   * the executor (k_lz4_exec) takes 64 consecutive sequences at a time: positions -> tokens are decoded from an LDS stage
//     of just the batch's compressed span, and the list is VERIFIED as it is used (token k must end exactly where token k+1
//     starts, the last one at the end of the block) — a list that passes is the true chain, whatever wrote it;
//   * all output is assembled in an LDS ring of the last 4 KiB (pieces of < 16 bytes by four unconditional sub-stores, the
//     lanes that do not need one aim it at a dump slot: no exec-mask flips; longer pieces by 16-byte stores with an overlapped
//     tail), so in-batch dependencies and every match within the ring's reach (56 % of text matches, 81 % of records) are
//     LDS reads, and memory sees only aligned 1 KiB blocks, 16 bytes per lane — each hashed (XXH3) from the registers that
//     store it.
#pragma once
#include "zpk_device.h"
#include "xxh3_device.h"
#include "seq_exec.h"

namespace zpk {

// LDS per wave: 4 KiB ring + 2 x 1.1 KiB stage = 6.3 KiB -> 25 waves per CU (measured: 20, 28 and 32 waves per CU run the same
// speed — the executor is bound by vector-ALU issue, not by latency).
#ifndef LX_MAX_LL
#define LX_MAX_LL 1024u                      // longest literal run / match of a "friendly" entry (the first run of a text is several hundred bytes)
#endif
#define LX_MAX_ML LX_MAX_LL
#ifndef LX_RING
#define LX_RING 4096u                        // bytes of output kept in LDS: abs positions [rb, rb + LX_RING), rb a multiple of 1 KiB
#endif
#define LX_HIST 1024u                        // a slide keeps at least this much flushed history (>= LX_MAX_ML: a source is all-ring or all-memory)
#ifndef LX_STAGE
#define LX_STAGE 1088u                       // compressed span of one batch (>= 1 + 3 + LX_MAX_LL + 2 + 3)
#endif
static_assert(LX_STAGE >= LX_MAX_LL + 16u, "one sequence always fits the stage");
static_assert(LX_RING - LX_HIST - 1023u >= LX_MAX_LL + LX_MAX_ML, "one sequence always fits behind a slide");
static_assert(LX_HIST >= LX_MAX_ML, "an overlapping match's source is always in the ring");

// byte offset of the token-list region that belongs to source offset `off` (regions of disjoint source ranges are disjoint)
__device__ __forceinline__ u64 lx_region(u64 off) { return ((off * 3) >> 4) << 2; }

// ------------------------------------------------------------------------------------------------ token scan

// The scan works in UNITS: a unit is one LX_SEG-byte segment of one compressed block.  k_lz4_frames (one lane per entry) checks the
// frame header, walks the block headers and appends a unit record per segment; k_lz4_scan (one lane per unit) lists the tokens that
// START inside its segment and notes where its chain leaves the segment (the unit's EXIT).  Segment 0 starts on the true chain (block
// position 0); every other segment starts its walk LX_RUNIN bytes early at an arbitrary byte — LZ4 token chains re-synchronise within
// a few dozen bytes on text — and records nothing before its segment begins.  k_lz4_seam (one lane per seam) then makes the chain
// EXACT: from unit j's exit it walks on until it steps on a token that unit j+1 listed; the tokens walked are unit j+1's PATCH and the
// listed token is where unit j+1's own list becomes valid.  If unit j's tail is true so is its exit, hence the patch, hence unit j+1
// from the joining token on, hence its tail: by induction from unit 0 the whole chain is the true one whenever every seam joins inside
// the next segment (else the entry is left to the general decoder; periodic data — the records class — needs 1 KiB and more to join in
// 1 % of the seams).  And nothing trusts even that: the executor verifies the chain as it uses it.  One lane per block (64 KiB = ~5 k
// serial token steps at 1.5 waves per SIMD) took 4 ms for the C2 batch; units give 8x the lanes and 1/8 the chain.
#ifndef LX_SEG
#define LX_SEG 8192u
#endif
#define LX_RUNIN 384u
#ifndef LXS_WIN
#define LXS_WIN 256u                         // per-lane window of the compressed stream in LDS (measured: 128 B windows = 17 waves per CU run the
                                             // scan 45 % SLOWER — the wave-wide refills, one memory round trip each, set the pace, not occupancy)
#endif
#define LXS_NCH (LXS_WIN / 16u)                // chunks of the ring (a power of two)
#define LXS_STRIDE (LXS_WIN + 16u)           // 16-byte aligned slots, banks staggered
struct alignas(16) Lz4ScanShared { u8 win[64 * LXS_STRIDE + 32]; };
struct Lz4Unit { u64 blk; u32 bsz_seg; u32 entry; };      // block payload offset in src; bsz | segment index << 20; entry index

// segments of a block of bsz bytes: LX_SEG each, the last one takes a remainder of less than a quarter segment along
__device__ __forceinline__ u32 lx_nsub(u32 bsz) { const u32 n = (bsz + (LX_SEG * 3) / 4) / LX_SEG; return n ? n : 1u; }
__device__ __forceinline__ u32 lx_seg_hi(u32 bsz, u32 j) { return j + 1 >= lx_nsub(bsz) ? bsz : (j + 1) * LX_SEG; }
// sub-list of the segment that starts at source offset `seg_abs`: [u32 count][u16 block-relative token position x count]
__device__ __forceinline__ u64 lx_sublist(u64 seg_abs) { return lx_region(seg_abs); }

// One lane per entry: frame header + block walk.  Returns 1 when the frame is of the plain kind (v1, 64 KiB blocks, no checksums /
// content size / dictionary) and its blocks end exactly at the end of the entry; appends the units of its compressed blocks.
// All 64 lanes call (lanes without an entry pass e_size = 0).
__device__ inline u32 lz4_frames_wave(const u8* src, u64 read_hi, u64 e_off, u64 e_size, u32 entry, Lz4Unit* units, u32* unit_count,
                                      u32 unit_cap, int lane)
{
    const u64 iend = e_off + e_size;
    u64 ip = e_off;
    bool ok = e_size >= 11 && iend <= read_hi;
    for (int guard = 0; ok; guard++) {          // skippable frames in front
        if (iend - ip < 4 || guard > 16) { ok = false; break; }
        const u32 magic = ld32(src + ip);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (iend - ip < 8) { ok = false; break; }
            const u64 sz = ld32(src + ip + 4);
            if ((u64)(iend - ip) - 8 < sz) { ok = false; break; }
            ip += 8 + sz;
            continue;
        }
        if (magic != 0x184D2204u) ok = false;
        break;
    }
    if (ok) {
        if (iend - ip < 7) ok = false;
        else {
            const u32 flg = ld8(src + ip + 4), bd = ld8(src + ip + 5), hc = ld8(src + ip + 6);
            u32 h = ZPK_P32_5 + 2u;                                          // header checksum = second byte of XXH32(FLG, BD)
            h = rotl32(h + flg * ZPK_P32_5, 11) * ZPK_P32_1;
            h = rotl32(h + bd * ZPK_P32_5, 11) * ZPK_P32_1;
            h ^= h >> 15; h *= ZPK_P32_2; h ^= h >> 13; h *= ZPK_P32_3; h ^= h >> 16;
            if ((flg & 0xDFu) != 0x40u || bd != 0x40u || ((h >> 8) & 0xFFu) != hc) ok = false;
            ip += 7;
        }
    }
    bool walking = ok;
    for (u32 step = 0; __ballot(walking) != 0; step++) {
        u32 nu = 0, bsz = 0;
        u64 blk = 0;
        if (walking) {
            if (iend - ip < 4) { ok = false; walking = false; }
            else {
                const u32 bh = ld32(src + ip);
                ip += 4;
                if (bh == 0) { walking = false; ok = ip == iend; }
                else {
                    bsz = bh & 0x7FFFFFFFu;
                    if (bsz > 65536u || (u64)(iend - ip) < bsz || (bsz == 0 && !(bh >> 31))) { ok = false; walking = false; }
                    else {
                        if (!(bh >> 31)) { nu = lx_nsub(bsz); blk = ip; }
                        ip += bsz;
                    }
                }
            }
        }
        // append this step's units: one atomic per wave
        const u32 x = wave_scan_add(nu);
        const u32 tot = (u32)__builtin_amdgcn_readlane((int)x, 63);
        if (tot) {
            u32 base = 0;
            lane0_guard();
            if (lane == 0) base = atomicAdd(unit_count, tot);
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            lane0_guard();
            const u32 mine = base + (x - nu);
            if (nu && mine + nu > unit_cap) { ok = false; walking = false; nu = 0; }
            for (u32 j = 0; j < nu; j++) { Lz4Unit u; u.blk = blk; u.bsz_seg = bsz | (j << 20); u.entry = entry; units[mine + j] = u; }
        }
        if (step > (1u << 20)) { ok = false; break; }
    }
    return ok ? 1u : 0u;
}

// One lane per unit.  Returns false when the lane's unit cannot be listed (malformed token, a literal run / match beyond the friendly
// limits, list region full): the caller clears the entry's verdict.  All 64 lanes call (lanes without a unit pass bsz = 0).
__device__ inline bool lz4_scan_units(Lz4ScanShared& sh, const u8* src, u64 read_hi, u64 blk, u32 bsz, u32 seg, u8* tok, int lane, u64* sdbg = nullptr)
{
#ifdef LX_STATS
    u64 st_ev = 0, st_t0 = __builtin_amdgcn_s_memtime(); u32 st_nev = 0, st_again = 0, st_steps = 0, st_special = 0;
#endif
    (void)sdbg;
    const u32 seg_lo = seg * LX_SEG, seg_hi = lx_seg_hi(bsz, seg);
    u64 w = lx_sublist(blk + seg_lo);
    const u64 wend = lx_sublist(blk + seg_hi);
    const u64 cw = w;                          // header: [u32 count][u32 exit]; positions follow; [u32 patch count][u32 valid from] end the region
    w += 8;
    // the lane's window: a ring of LXS_NCH 16-byte chunks of the block's byte stream, chunk c (bytes [16c, 16c + 16)) in ring slot
    // c mod LXS_NCH, i.e. byte x at slot[x mod LXS_WIN]; chunks [lo, hi) are in the ring, chunks [hi, hi + npend) are IN FLIGHT in
    // registers.  Refills are events for the whole wave (lanes run in lockstep: a private refill would stall all 64 on every step):
    // when any lane is about to run dry, every lane stores the chunks that arrived since the last event and asks for as many new
    // ones as its ring has room for — so a load has a whole event interval (~15 steps) to land and nobody waits for memory.
    const u64 max_chunk = read_hi > blk ? (read_hi - blk) >> 4 : 0;            // 16-byte loads stay inside the source
    bool ok = bsz != 0 && w + 8 <= wend && ((u64)bsz + 15) >> 4 <= max_chunk;
    u32 exit_pos = bsz;
    bool live = ok;
    u32 p = seg_lo > LX_RUNIN ? seg_lo - LX_RUNIN : 0u;        // (segment 0, or a run-in that reaches position 0, is the true chain)
    u32 n = 0;
    const lds_p8 slot = to_lds_rw(sh.win) + LXS_STRIDE * (u32)lane;
    u32 lo = p >> 4, hi = lo, npend = 0;
    u128 pend[LXS_NCH];
    #pragma unroll
    for (u32 k = 0; k < LXS_NCH; k++) { pend[k].lo = 0; pend[k].hi = 0; }
    const u8* const sb = src + blk;
    for (u32 step = 0;; step++) {
        if (__ballot(live) == 0) break;
        if (step > 70000u) { ok = false; break; }                              // cannot happen: every step consumes input
        // ---- refill event: some lane has less than two chunks ahead of it ----
#ifdef LX_STATS
        const u64 st_e0 = __builtin_amdgcn_s_memtime(); st_steps++;
#endif
        for (int again = 0; again < 2; again++) {
            const u32 cur = p >> 4;
            if (__ballot(live && (cur + 2 > hi || cur < lo)) == 0) break;
#ifdef LX_STATS
            st_nev++; st_again += again;
#endif
            if (live) {
                #pragma unroll
                for (u32 k = 0; k < LXS_NCH; k++) if (k < npend) lds_st128(slot + (((hi + k) & (LXS_NCH - 1)) << 4), pend[k]);   // what has arrived
                hi += npend; npend = 0;
                if (hi - lo > LXS_NCH) lo = hi - LXS_NCH;
                if (cur >= hi || cur < lo) { lo = cur; hi = cur; }              // a jump past everything loaded: start over at the lane's chunk
                if (lo < cur) lo = cur;                                         // chunks behind the lane are dead
                u32 room = LXS_NCH - (hi - lo);
                const u32 left = (bsz + 15) / 16 > hi ? (bsz + 15) / 16 - hi : 0u;
                if (room > left) room = left;
                #pragma unroll
                for (u32 k = 0; k < LXS_NCH; k++) if (k < room) pend[k] = ld128(sb + 16ull * (hi + k));
                npend = room;
            }
            wave_mem_fence();
            // (second pass only for a lane that is STILL dry — it jumped, or the wave has just started: its loads are waited for at once)
        }
#ifdef LX_STATS
        st_ev += __builtin_amdgcn_s_memtime() - st_e0;
#endif
        // ---- one token per live lane.  The common shapes (at most one extension byte per length, everything inside the window and
        // the block) run straight-line: the loop is executed ~1.3 k times by ~8 k waves, every instruction in it counts ----
        const u32 t = lds_ld8((lds_cp8)(slot + (p & (LXS_WIN - 1))));
        const u32 e1 = lds_ld8((lds_cp8)(slot + ((p + 1) & (LXS_WIN - 1))));
        const bool lx = (t >> 4) == 15;
        u32 lit = (t >> 4) + (lx ? e1 : 0u);
        const u32 mo = p + 1 + (lx ? 1u : 0u) + lit;                           // offset field
        const bool far = ((mo + 2) >> 4) >= hi;                                // the match-length byte is not in the ring
        const u32 e2 = lds_ld8((lds_cp8)(slot + ((mo + 2) & (LXS_WIN - 1))));
        const u32 mlc = t & 15;
        const bool mx = mlc == 15;
        u32 nxt = mo + 2 + (mx ? 1u : 0u);
        bool last = mo == bsz;                                                 // literals run to the end of the block: the last sequence
        bool reject = mo > bsz || (!last && (bsz - mo < 2 || nxt >= bsz));
        const bool special = (lx && e1 == 255) || (mx && !last && (far || e2 == 255)) || reject;
        if (__ballot(live && special) != 0) {                                  // rare: long extension chains, malformed tokens, run-in misses
#ifdef LX_STATS
            st_special++;
#endif
            if (live && special) {
                const u8* b = src + blk;
                bool bad = false;
                u32 q = p + 1, ml = mlc;
                lit = t >> 4;
                if (lit == 15) { u32 x; do { if (q >= bsz) { bad = true; break; } x = ld8(b + q); q++; lit += x; } while (x == 255 && lit <= LX_MAX_LL); }
                if (!bad && (lit > LX_MAX_LL || lit > bsz - q)) bad = true;
                if (!bad) {
                    const u32 m2 = q + lit;
                    last = m2 == bsz;
                    nxt = m2 + 2;
                    if (!last) {
                        if (bsz - m2 < 2) bad = true;
                        else if (mx) { u32 x; do { if (nxt >= bsz) { bad = true; break; } x = ld8(b + nxt); nxt++; ml += x; } while (x == 255 && ml <= LX_MAX_ML); }
                        if (!bad && (ml + 4 > LX_MAX_ML || nxt >= bsz)) bad = true;
                    }
                }
                reject = bad;
            }
        }
        {   // the step's outcome, by selects (one exec-masked store is the only divergent piece)
            const bool miss = live && reject && p < seg_lo;                    // still in the run-in: not a token after all, try the next byte
            const bool fail = live && reject && p >= seg_lo;
            const bool good = live && !reject;
            const bool rec = good && p >= seg_lo;                              // the token starts inside this segment: list it
            const bool full = rec && w + 10 > wend;
            if (rec && !full) st16(tok + w, (u16)p);
            w += rec && !full ? 2u : 0u;
            n += rec && !full ? 1u : 0u;
            ok = ok && !fail && !full;
            const bool leave = good && !last && nxt >= seg_hi;                 // the chain leaves the segment: the unit's exit
            exit_pos = leave ? nxt : exit_pos;                                 // (a chain that ENDS inside the unit keeps exit = bsz)
            p = miss ? p + 1 : (good && !last ? nxt : p);
            live = live && !fail && !full && !(good && last) && !leave;
        }
    }
#ifdef LX_STATS
    if (sdbg && lane == 0) { sdbg[0] = __builtin_amdgcn_s_memtime() - st_t0; sdbg[1] = st_ev; sdbg[2] = st_nev; sdbg[3] = st_again; sdbg[4] = st_steps; sdbg[5] = st_special; }
#endif
    if (ok) {
        { st32(tok + cw, n); st32(tok + cw + 4, exit_pos); st32(tok + wend - 8, 0u); st32(tok + wend - 4, 0u); }
    }
    return ok || bsz == 0;
}

// One lane per seam (unit j -> unit j + 1 of the same block): see the comment above.  Returns false when the seam does not join.
__device__ inline bool lz4_seam(const u8* src, u64 blk, u32 bsz, u32 seg, u8* tok)
{
    const u32 lo1 = (seg + 1) * LX_SEG;
    if (bsz == 0 || seg + 1 >= lx_nsub(bsz)) return true;                        // the block's last unit: no seam behind it
    const u32 hi1 = lx_seg_hi(bsz, seg + 1);
    const u64 r0 = lx_sublist(blk + seg * LX_SEG), r1 = lx_sublist(blk + lo1), r1e = lx_sublist(blk + hi1);
    const u32 X = ld32(tok + r0 + 4);
    const u32 c1 = ld32(tok + r1);
    const u8* L = tok + r1 + 8;
    u32 pc = 0, v = 0;
    if (X >= bsz) v = c1;                                                        // the chain ended before this seam: nothing of unit j + 1 is real
    else if (c1 != 0 && (u32)ld16(L) == X) v = 0;                                 // the common case: unit j + 1 was on the chain from its first token
    else {
        if (X < lo1) return false;
        const u8* b = src + blk;
        const u64 floor_ = r1 + 8 + 2ull * ((c1 + 1) & ~1u);
        u64 pw = 0;
        // two passes over the same short walk: count the patch, then write it in ascending order just below its header
        for (int pass = 0; pass < 2; pass++) {
            u32 p = X, idx = 0, k = 0;
            for (;;) {
                while (idx < c1 && (u32)ld16(L + 2ull * idx) < p) idx++;
                if (idx < c1 && (u32)ld16(L + 2ull * idx) == p) { v = idx; break; }   // joined unit j + 1's own chain
                if (p >= hi1) return false;                                          // left the segment without joining: give the entry up
                if (pass) st16(tok + pw + 2ull * k, (u16)p);
                k++;
                // next token (bytes straight from memory: seams that need a patch are few and their walks short)
                const u32 t = ld8(b + p);
                u32 q = p + 1, lit = t >> 4;
                if (lit == 15) { u32 x; do { if (q >= bsz) return false; x = ld8(b + q); q++; lit += x; } while (x == 255 && lit <= LX_MAX_LL); }
                if (lit > LX_MAX_LL || lit > bsz - q) return false;
                q += lit;
                if (q == bsz) { v = c1; break; }                                     // the block's last sequence was in the patch
                if (bsz - q < 2) return false;
                q += 2;
                u32 ml = t & 15;
                if (ml == 15) { u32 x; do { if (q >= bsz) return false; x = ld8(b + q); q++; ml += x; } while (x == 255 && ml <= LX_MAX_ML); }
                if (ml + 4 > LX_MAX_ML || q >= bsz) return false;
                p = q;
            }
            if (!pass) {
                pc = k;
                const u64 need = 2ull * ((pc + 1) & ~1u);
                if (r1e - 8 < floor_ + need) return false;                           // no room between unit j + 1's list and its patch header
                pw = r1e - 8 - need;
            }
        }
    }
    st32(tok + r1e - 8, pc); st32(tok + r1e - 4, v);
    return true;
}

// ------------------------------------------------------------------------------------------------ executor

struct alignas(16) Lz4ExecShared {
    u8 ring[LX_RING + 32];                   // + 32: 16-byte accesses start anywhere below LX_RING + 16
    u8 stage[2][LX_STAGE + 48];              // two: batch b+1 streams in (LDS-DMA) while batch b is decoded
};

// anomaly codes (diagnostics only: whatever the code, the entry goes to the general decoder)
enum { LX_OK = 0, LX_E_FRAME = 1, LX_E_LIST = 2, LX_E_TOKEN = 3, LX_E_OFFSET = 4, LX_E_CAPACITY = 5, LX_E_BLOCKMAX = 6, LX_E_FIT = 7, LX_E_ROUNDS = 8 };

#ifdef LX_STATS
#define LXT(slot) do { const u64 t_ = __builtin_amdgcn_s_memtime(); O.tm[slot] += t_ - O.t_last; O.t_last = t_; } while (0)
#else
#define LXT(slot) do { } while (0)
#endif
struct LxOut {
#ifdef LX_STATS
    u64 tm[12]; u64 t_last;      // developer: cycles per phase (wait, prefetch, decode, scan, deps, lit, match, rounds, flush, slide, other)
#endif
    lds_p8 ring;         // LDS ring base
    u8* dst;             // the entry's output slot in memory
    u32 wp, rb, fp;      // abs output positions: next byte, ring[0], flushed up to (rb, fp multiples of 1 KiB; rb <= fp <= wp)
    u32 hash_blocks;     // 1 KiB blocks the fused XXH3 takes as whole blocks
    lds_cp8 sec;         // the XXH3 secret in LDS (192 bytes), or null: constant memory
    Xxh3Lite xs;
};

// LDS accesses here are 8-byte ALIGNED only.  Measured (rocprofv3, round 2): with 16-byte reads and 2/4/8-byte stores at
// arbitrary byte addresses the kernel was bound by the LDS itself — SQ_LDS_UNALIGNED_STALL 1.2e9 and SQ_LDS_IDX_ACTIVE 1.9e9 of
// 2.8e9 CU-cycles, 11.7 LDS cycles per instruction — so
//   * 16 bytes at any address = three aligned 8-byte reads + byte funnel shifts in registers;
//   * n bytes TO any address = the piece, masked to n bytes and shifted to its place, OR-ed into three aligned 8-byte words
//     (ds_or_b64): ring bytes at and beyond the write position are kept ZERO, so neighbouring pieces that share a word merge
//     without a read-modify-write race, lanes with nothing to store OR zeros (no exec-mask flips, no dump slot), and the
//     overlapped tail of a long piece ORs the same bytes twice.
__device__ __forceinline__ u128 lds_ld16_any(lds_cp8 base, u32 a)
{
    const ZPK_LDS u32* w = (const ZPK_LDS u32*)(base + (a & ~7u));
    const u32 d0 = w[0], d1 = w[1], d2 = w[2], d3 = w[3], d4 = w[4], d5 = w[5];
    const bool k = a & 4;
    const u32 r = a & 3;
    const u32 e0 = k ? d1 : d0, e1 = k ? d2 : d1, e2 = k ? d3 : d2, e3 = k ? d4 : d3, e4 = k ? d5 : d4;
    const u32 o0 = __builtin_amdgcn_alignbyte(e1, e0, r), o1 = __builtin_amdgcn_alignbyte(e2, e1, r),
              o2 = __builtin_amdgcn_alignbyte(e3, e2, r), o3 = __builtin_amdgcn_alignbyte(e4, e3, r);
    u128 v; v.lo = ((u64)o1 << 32) | o0; v.hi = ((u64)o3 << 32) | o2;
    return v;
}

// the first n (0..16) bytes of v OR-ed into LDS at byte offset pos of `base` (8-byte aligned object); target bytes must be zero
__device__ __forceinline__ void lds_or_piece(lds_p8 base, u32 pos, u128 v, u32 n)
{
    const u32 nlo = n < 8 ? n : 8, nhi = n < 8 ? 0 : n - 8;
    const u64 lo = nlo >= 8 ? v.lo : v.lo & ((1ull << (8 * nlo)) - 1);
    const u64 hi = nhi >= 8 ? v.hi : v.hi & ((1ull << (8 * nhi)) - 1);
    const u32 s = (pos & 7u) * 8u;
    const u64 q0 = lo << s;
    const u64 q1 = s ? (hi << s) | (lo >> (64 - s)) : hi;
    const u64 q2 = s ? hi >> (64 - s) : 0ull;
    ZPK_LDS u64* t = (ZPK_LDS u64*)(base + (pos & ~7u));
    __hip_atomic_fetch_or(t, q0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(t + 1, q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_fetch_or(t + 2, q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 16 bytes of OUTPUT at abs position s: from the ring when they are there, else from memory (flushed long ago)
__device__ __forceinline__ u128 lx_load16(const LxOut& O, u32 s)
{
    u128 v;
#ifdef LX_ABL_NOGATHER
    return lds_ld16_any((lds_cp8)O.ring, (s - O.rb) & (LX_RING - 1));
#endif
    if (s >= O.rb) v = lds_ld16_any((lds_cp8)O.ring, s - O.rb);
    else v = ld128(O.dst + s);
    return v;
}

__device__ __forceinline__ void lx_flush_blocks(LxOut& O, int lane)
{
    while (O.fp + 1024u <= O.wp) {
        const u128 v = lds_ld128((lds_cp8)(O.ring + (O.fp - O.rb) + 16u * (u32)lane));
#ifndef LX_ABL_NOFLUSH
        st128(O.dst + O.fp + 16u * (u32)lane, v);
#endif
#ifndef LX_ABL_NOHASH
        if ((O.fp >> 10) < O.hash_blocks) O.xs.block(v, lane, O.sec);
#endif
        O.fp += 1024u;
    }
}

// make the ring hold [rb', wp) with rb' = fp - LX_HIST: afterwards at least LX_RING - LX_HIST - 1023 bytes are free
__device__ __forceinline__ void lx_slide(LxOut& O, int lane)
{
    lx_flush_blocks(O, lane);
    const u32 nrb = O.fp >= LX_HIST ? O.fp - LX_HIST : 0u;
    if (nrb > O.rb) {
        const u32 shift = nrb - O.rb, n = O.wp - nrb;
        for (u32 c = 16u * (u32)lane; c < n; c += 1024u) {        // rounds in order: a round only overwrites what earlier rounds have read
            const u128 v = lds_ld128((lds_cp8)(O.ring + shift + c));
            lds_st128(O.ring + c, v);
        }
        wave_mem_fence();
        for (u32 c = ((n + 15u) & ~15u) + 16u * (u32)lane; c < n + shift; c += 1024u) { u128 z; z.lo = 0; z.hi = 0; lds_st128(O.ring + c, z); }   // ring bytes >= wp stay zero
        wave_mem_fence();
        O.rb = nrb;
    }
}

// `n` raw bytes from memory appended to the output (stored LZ4 blocks)
__device__ inline int lx_append_raw(LxOut& O, const u8* s, u64 n, const u8* read_hi, u64 dst_cap, int lane)
{
    if ((u64)O.wp + n > dst_cap) return LX_E_CAPACITY;
    while (n) {
        u32 m = n < 1024u ? (u32)n : 1024u;
        if (O.wp + m > O.rb + LX_RING) lx_slide(O, lane);
        const u32 c = 16u * (u32)lane;
        {
            u128 v; v.lo = 0; v.hi = 0;
            if (c < m) {
                if (s + c + 16 <= read_hi) v = ld128(s + c);
                else for (u32 i = 0; i < 16 && s + c + i < read_hi; i++) { const u64 b = (u64)ld8(s + c + i) << (8 * (i & 7)); if (i < 8) v.lo |= b; else v.hi |= b; }
            }
            lds_or_piece(O.ring, (O.wp - O.rb) + (c < m ? c : 0u), v, c < m ? (m - c < 16u ? m - c : 16u) : 0u);
        }
        wave_mem_fence();
        O.wp += m; s += m; n -= m;
        lx_flush_blocks(O, lane);
    }
    return LX_OK;
}

// `n` copies of one byte appended to the output (Zstandard RLE blocks / RLE literals)
__device__ inline int lx_append_fill(LxOut& O, u32 byte, u64 n, u64 dst_cap, int lane)
{
    if ((u64)O.wp + n > dst_cap) return LX_E_CAPACITY;
    u128 pat; pat.lo = 0x0101010101010101ull * (u64)(byte & 0xFFu); pat.hi = pat.lo;
    while (n) {
        const u32 m = n < 1024u ? (u32)n : 1024u;
        if (O.wp + m > O.rb + LX_RING) lx_slide(O, lane);
        const u32 c = 16u * (u32)lane;
        lds_or_piece(O.ring, (O.wp - O.rb) + (c < m ? c : 0u), pat, c < m ? (m - c < 16u ? m - c : 16u) : 0u);
        wave_mem_fence();
        O.wp += m; n -= m;
        lx_flush_blocks(O, lane);
    }
    return LX_OK;
}

// a match of ANY length appended to the output: out[i] = out[i - off].  Everything from (match start - off) on is periodic with period
// off, so every step copies from the largest multiple of off that is already there (<= 1 KiB): the step size doubles until it is 1 KiB
__device__ inline int lx_append_match(LxOut& O, u32 off, u64 n, u32 hist_lo, u64 dst_cap, int lane)
{
    if (off == 0 || off > O.wp - hist_lo) return LX_E_OFFSET;
    if ((u64)O.wp + n > dst_cap) return LX_E_CAPACITY;
    u64 done = 0;
    while (n) {
        u32 m, D;
        if (off >= 1024u) { D = off; m = n < 1024u ? (u32)n : 1024u; }
        else {
            const u64 avail = (u64)off + done;                                   // periodic bytes behind the write position
            const u32 cap = avail < 1024u ? (u32)avail : 1024u;
            D = cap / off * off;                                                 // >= off
            m = n < D ? (u32)n : D;
        }
        if (O.wp + m > O.rb + LX_RING) lx_slide(O, lane);
        const u32 s = O.wp - D;
        if (m >= 16u) {
            const u32 c = 16u * (u32)lane;
            if (c < m) {
                const u32 oc = c + 16u <= m ? c : m - 16u;                        // the last chunk overlaps the one before: same bytes twice
                lds_or_piece(O.ring, (O.wp - O.rb) + oc, lx_load16(O, s + oc), 16u);
            }
        } else if (lane == 0) lds_or_piece(O.ring, O.wp - O.rb, lx_load16(O, s), m);
        wave_mem_fence();
        O.wp += m; n -= m; done += m;
        lx_flush_blocks(O, lane);
    }
    return LX_OK;
}

// literals that lie in memory (Zstandard: the block's raw literals, or the Huffman-decoded ones in the workgroup's scratch)
struct LxLitGlobal {
    const u8* p; const u8* rd_hi;
    __device__ __forceinline__ u128 load16(u32 oc) const
    {
        const u8* g = p + oc;
        if (g + 16 <= rd_hi) return ld128(g);
        u128 v; v.lo = 0; v.hi = 0;
        for (u32 i = 0; i < 16 && g + i < rd_hi; i++) { const u64 b = (u64)ld8(g + i) << (8 * (i & 7)); if (i < 8) v.lo |= b; else v.hi |= b; }
        return v;
    }
};
// literals that are all one byte (Zstandard RLE literals)
struct LxLitFill {
    u32 byte;
    __device__ __forceinline__ u128 load16(u32) const { u128 v; v.lo = 0x0101010101010101ull * (u64)(byte & 0xFFu); v.hi = v.lo; return v; }
};

// literal source of a batch lane: LZ4 reads its literals out of the LDS stage of the compressed span
struct LxLitStage {
    lds_cp8 S; u32 lit_a;
    __device__ __forceinline__ u128 load16(u32 oc) const { return lds_ld16_any(S, lit_a + oc); }
};

// Execute `cnt` sequences (lane k < cnt: literal length ll from L, then a match of ml bytes at distance off; ml = 0: none) at the
// ring's write position: output positions (prefix sum), in-batch dependencies, literals and matches OR-ed into the ring, whole
// 1 KiB blocks flushed (and hashed).  cnt may come back SMALLER: a batch whose output does not fit the ring is cut (the caller goes
// on behind the sequences taken).  Lengths must be <= LX_MAX_LL / LX_MAX_ML.  hist_lo = lowest abs output position a match may reach.
template <class LitSrc>
__device__ __forceinline__ int lx_exec_batch(LxOut& O, u32& cnt, u32 ll, u32 ml, u32 off, const LitSrc& L, u32 hist_lo, u64 dst_cap,
                                             int lane, SeqStats& stt)
{
    bool act = (u32)lane < cnt;
    if (!act) { ll = 0; ml = 0; }
    // ---- output positions ----
    u32 x = wave_scan_add(ll + ml);
    u32 total = (u32)__builtin_amdgcn_readlane((int)x, 63);
    if (O.wp + total > O.rb + LX_RING) {                 // (only a batch of more than 2 KiB of output gets here)
        lx_slide(O, lane);
        const u32 free_ = O.rb + LX_RING - O.wp;
        if (total > free_) {                         // take the sequences that fit; the rest next time round
            const u32 c2 = (u32)__popcll(__ballot(act && x <= free_));
            if (c2 == 0) return LX_E_FIT;
            cnt = c2; act = (u32)lane < cnt;
            if (!act) { ll = 0; ml = 0; }
            x = wave_scan_add(ll + ml);
            total = (u32)__builtin_amdgcn_readlane((int)x, 63);
        }
    }
    if ((u64)O.wp + total > dst_cap) return LX_E_CAPACITY;
    const u32 o = O.wp + (x - ll - ml);              // abs position of this sequence's literals
    const u32 ms = o + ll;                           //               ... of its match
    const bool has_match = act && ml != 0;
    if (__ballot(has_match && (off == 0 || off > ms - hist_lo)) != 0) return LX_E_OFFSET;
    LXT(3);
    // ---- in-batch dependencies (positions relative to wp) ----
    u64 pending = __ballot(has_match);
    const u32 r_ms = ms - O.wp, r_me = r_ms + ml;
    const u32 need_len = ml < off ? ml : off;
    i64 srel;
    u64 need = 0;
    {
        // Few lanes read this batch's own output on text (1-2 of 64): for those, one sweep per reader — its source range is
        // broadcast, every earlier lane answers with one compare, the ballot is the reader's dependency set — costs a dozen
        // plain instructions; the sorted search + pointer jumping of seq_dependencies (a dozen dependent LDS round trips)
        // is for batches where most lanes do (records: every match reads the record before it).
        const i32 s0 = (i32)r_ms - (i32)off;
        const u64 rd = __ballot(has_match && s0 + (i32)need_len > 0);
        if (__popcll(rd) <= 6) {
            srel = s0;
            u64 m = rd;
            while (m) {
                const int k = __ffsll((long long)m) - 1;
                m &= m - 1;
                const i32 ks = __builtin_amdgcn_readlane(s0, k);
                const i32 ke = ks + (i32)__builtin_amdgcn_readlane((int)need_len, k);
                const u64 ov = __ballot(has_match && lane < k && (i32)r_me > ks && (i32)r_ms < ke);
                if (lane == k) need = ov;
            }
        } else need = seq_dependencies<i32>(has_match, r_ms, r_me, off, need_len, pending, lane, srel, stt);
    }
    const u32 sabs = (u32)((i64)O.wp + srel);        // abs position of the (possibly re-pointed) source
    const bool overlap = ml > need_len;              // offset < length: the match feeds itself
    const bool early = has_match && !overlap && srel + (i64)need_len <= 0;
    const lds_p8 ring = O.ring;
    LXT(4);
    // ---- literals: stage -> ring ----
    // (both first loads — the literal run's and the early match's — are issued before either is used: when they come from memory
    // (Zstandard literals, far matches) that is one round trip instead of two)
    u128 mv0; mv0.lo = 0; mv0.hi = 0;
    const u128 lv0 = L.load16(0u);
#ifndef LX_ABL_NOMATCH
    if (early) mv0 = lx_load16(O, sabs);
#endif
#ifndef LX_ABL_NOLIT
    {
        // every lane: its first 16 literal bytes (or fewer); the few longer runs go on in 16-byte steps, the last one overlapped
        lds_or_piece(ring, o - O.rb, lv0, ll < 16 ? ll : 16u);
        u64 bm = __ballot(ll > 16);
        for (u32 c = 16; bm; c += 16) {
            const bool on = c < ll;
            const u32 oc = !on ? 0u : (c + 16 <= ll ? c : ll - 16);
            lds_or_piece(ring, o - O.rb + oc, L.load16(oc), on ? 16u : 0u);
            bm = __ballot(c + 16 < ll);
        }
    }
#endif
    LXT(5);
    // ---- matches whose whole source is older than this batch ----
#ifndef LX_ABL_NOMATCH
    {
        lds_or_piece(ring, ms - O.rb, mv0, !early ? 0u : (ml < 16 ? ml : 16u));
        u64 bm = __ballot(early && ml > 16);
        for (u32 c = 16; bm; c += 16) {
            const bool on = early && c < ml;
            const u32 oc = !on ? 0u : (c + 16 <= ml ? c : ml - 16);
            u128 vv; vv.lo = 0; vv.hi = 0;
            if (on) vv = lx_load16(O, sabs + oc);
            lds_or_piece(ring, ms - O.rb + oc, vv, on ? 16u : 0u);
            bm = __ballot(early && c + 16 < ml);
        }
    }
#endif
    wave_mem_fence();
    LXT(6);
    // ---- rounds: matches that read this batch's own output ----
    u64 done = ~pending | __ballot(early);
    pending &= ~done;
    u32 guard = 0;
    while (pending) {
        const bool ready = has_match && ((pending >> lane) & 1) && (need & ~done) == 0;
        const u64 rmask = __ballot(ready);
        if (rmask == 0 || ++guard > 70) return LX_E_ROUNDS;
        if (ready && !overlap) {
            for (u32 c = 0; c < ml; c += 16) {
                const u32 oc = c + 16 <= ml || ml < 16 ? c : ml - 16;
                lds_or_piece(ring, ms - O.rb + oc, lx_load16(O, sabs + oc), ml < 16 ? ml : 16u);
            }
        }
        u64 cm = __ballot(ready && overlap);
        while (cm) {                                 // self-overlapping matches, one at a time, whole wave; source and target are in the ring
            const int k = __ffsll((long long)cm) - 1;
            cm &= cm - 1;
            const u32 n = (u32)__builtin_amdgcn_readlane((int)ml, k);
            const u32 ko = (u32)__builtin_amdgcn_readlane((int)off, k);
            const u32 D = (u32)__builtin_amdgcn_readlane((int)ms, k) - O.rb;
            wave_mem_fence();
            if (ko < 16) {
                for (u32 c = (u32)lane; c < n; c += 64) lds_st8(ring + D + c, (u8)lds_ld8((lds_cp8)(ring + D - ko + c % ko)));
            } else {
                for (u32 base = 0; base < n; base += ko) {       // period by period: no lane reads what the same round writes
                    const u32 m = n - base < ko ? n - base : ko;
                    if (m >= 16) {
                        for (u32 c = 16u * (u32)lane; c < m; c += 1024u) {
                            const u32 oc = c + 16 <= m ? c : m - 16;
                            lds_or_piece(ring, D + base + oc, lds_ld16_any((lds_cp8)ring, D + base - ko + oc), 16u);
                        }
                    } else if (lane == 0) lds_or_piece(ring, D + base, lds_ld16_any((lds_cp8)ring, D + base - ko), m);
                    wave_mem_fence();
                }
            }
            wave_mem_fence();
        }
        wave_mem_fence();
        done |= rmask;
        pending &= ~rmask;
    }
    LXT(7);
    O.wp += total;
    lx_flush_blocks(O, lane);
    return LX_OK;
}

struct LxResult { int rc; u64 produced; u64 hash; };

// init / finish of an entry's ring
__device__ __forceinline__ void lx_begin(LxOut& O, lds_p8 ring, u8* dst, u64 uncomp_size, int lane, lds_p8 sec = nullptr)
{
    O.ring = ring; O.dst = dst; O.sec = (lds_cp8)sec;
    if (sec && lane < 12) lds_st128(sec + 16 * lane, ld128(XXH3_SECRET + 16 * lane));
    for (u32 c = 16u * (u32)lane; c < LX_RING + 32u; c += 1024u) { u128 z; z.lo = 0; z.hi = 0; lds_st128(O.ring + c, z); }     // ring bytes >= wp are zero, always
    wave_mem_fence();
    O.wp = 0; O.rb = 0; O.fp = 0;
    O.hash_blocks = uncomp_size > 240 ? (u32)((uncomp_size - 1) >> 10) : 0u;
    O.xs.init(lane);
}
// the tail (what is left of the last 1 KiB block, exact to the byte), then XXH3 of dst[0, uncomp_size) (lib/zpack_read.c:466): fused
// when the entry produced exactly that many bytes, by re-reading otherwise
__device__ __forceinline__ void lx_finish(LxOut& O, u8* dst, u64 uncomp_size, LxResult& R, int lane)
{
    lx_flush_blocks(O, lane);
    {
        const u32 tail = O.wp - O.fp, c = 16u * (u32)lane;
        if (c < tail) {
            const u128 v = lds_ld128((lds_cp8)(O.ring + (O.fp - O.rb) + c));
            if (c + 16 <= tail) st128(dst + O.fp + c, v);
            else gstore_upto16(dst + O.fp + c, v, tail - c);
        }
    }
    wave_mem_fence();
    R.produced = O.wp;
    if (uncomp_size > 240 && R.produced == uncomp_size) {
        const u64 nb = O.hash_blocks;
        const u32 nstripes = (u32)(((uncomp_size - 1) - (nb << 10)) >> 6);
        R.hash = uni64(O.xs.finish(dst + (nb << 10), nstripes, dst + uncomp_size, uncomp_size, lane));
    } else R.hash = xxh3_64_wave(dst, uncomp_size, lane);
}

// One sub-list of a compressed block: `nseq` listed tokens at lst[] (positions >= min_pos), block bytes blk[0, C); the last of them must
// end exactly at end_pos (where the next sub-list's first token starts; C and `final` for the block's last sub-list, whose last sequence
// is the literal-only one).  hist_lo = lowest abs output position a match may reach, block_out = where the block's output began.
__device__ inline int lx_block(Lz4ExecShared& sh, LxOut& O, const u8* blk, u32 C, const u8* read_hi, const u8* lst, u32 nseq,
                               u32 min_pos, u32 end_pos, bool final, u32 block_out, u32 hist_lo, u64 dst_cap, int lane)
{
    SeqStats stt = {};
    // The batch loop is software-pipelined over memory latency (the kernel is latency-bound, not issue-bound): while batch
    // b executes, the compressed span of batch b+1 streams into the OTHER stage buffer by LDS-DMA (global_load_lds: no
    // registers held), and the token positions of batch b+2 load into two registers.  A batch that has to be cut short
    // (ring or stage full) breaks the rhythm: the next one then loads and stages synchronously.
    auto load_pos = [&](u32 b, u32& p, u32& n) {
        const u32 i = b + (u32)lane;
        p = 0; n = 0;
        if (i < nseq) { const u32 pn = ld32(lst + 2ull * i); p = pn & 0xFFFFu; n = i + 1 < nseq ? pn >> 16 : end_pos; }   // (lists are padded: 2 bytes past the last position are readable)
    };
    // the batch that starts at sequence b with positions (p, n): how many sequences the stage holds, and their compressed span.
    // false: the list is not a strictly increasing chain inside the block (not a list this path wrote)
    auto plan = [&](u32 b, u32 p, u32 n, u32& cnt, u32& pf, u32& span) -> bool {
        const u32 want = nseq - b < 64u ? nseq - b : 64u;
        const bool a = (u32)lane < want;
        pf = (u32)__builtin_amdgcn_readfirstlane((int)p);
        if (__ballot(a && !(p < n && n <= end_pos && p >= pf && p >= min_pos)) != 0) return false;
        const u64 fm = __ballot(a && n - pf <= LX_STAGE);                    // n increases with the lane
        const u32 c2 = (u32)__popcll(fm);
        if (c2 == 0 || fm != (c2 >= 64 ? ~0ull : ((1ull << c2) - 1))) return false;
        cnt = c2;
        span = (u32)__builtin_amdgcn_readlane((int)n, (int)c2 - 1) - pf;
        return true;
    };
    u32 b0 = 0, P, Nx, Pn = 0, Nxn = 0;
    u32 c_cnt = 0, c_pf = 0, c_span = 0;   // the current batch's plan when it was already made for the prefetch (c_cnt != 0)
    int buf = 0;
    bool staged = false;          // stage[buf] is receiving (LDS-DMA) the span of the batch at b0
    bool have_next = false;       // (Pn, Nxn) are the positions of the batch at b0 + 64
    load_pos(0, P, Nx);
    while (b0 < nseq) {
        u32 cnt, p_first, span;
        LXT(10);
        if (c_cnt) { cnt = c_cnt; p_first = c_pf; span = c_span; }
        else if (!plan(b0, P, Nx, cnt, p_first, span)) return LX_E_LIST;
        LXT(11);
        if (O.rb + LX_RING - O.wp < LX_MAX_LL + LX_MAX_ML + 1) lx_slide(O, lane);      // room for any ordinary batch up front
        LXT(9);
        const lds_cp8 S = to_lds(sh.stage[buf]);
        if (staged) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the DMA issued one batch ago (and everything older)
        } else {
            const lds_p8 Sw = to_lds_rw(sh.stage[buf]);
            wave_mem_fence();
            for (u32 c = 16u * (u32)lane; c < span + 8u; c += 1024u) {     // (+ a few bytes the decode may look at past a sequence's end)
                const u8* g = blk + p_first + c;
                u128 v; v.lo = 0; v.hi = 0;
                if (g + 16 <= read_hi) v = ld128(g);
                else for (u32 k = 0; k < 16 && g + k < read_hi; k++) { const u64 bb = (u64)ld8(g + k) << (8 * (k & 7)); if (k < 8) v.lo |= bb; else v.hi |= bb; }
                lds_st128(Sw + c, v);
            }
        }
        wave_mem_fence();
        LXT(0);
        // ---- prefetch: batch b0 + 64's span into the other stage buffer, batch b0 + 128's positions into registers ----
        bool next_staged = false;
        u32 Pf = 0, Nxf = 0;
        u32 cn = 0, pfn = 0, spn = 0;
        const bool rhythm = cnt == 64u;
        if (rhythm && b0 + 64u < nseq) {
            if (have_next) {
                if (!plan(b0 + 64u, Pn, Nxn, cn, pfn, spn)) return LX_E_LIST;
                if (blk + pfn + spn + 8u + 16u <= read_hi) {
                    const u8* g = blk + pfn + 16u * (u32)lane;
                    ZPK_LDS u8* const d0 = (ZPK_LDS u8*)sh.stage[buf ^ 1];
                    if (16u * (u32)lane < spn + 8u) __builtin_amdgcn_global_load_lds((const ZPK_GLOBAL u32*)g, (ZPK_LDS u32*)d0, 16, 0, 0);
                    if (1024u + 16u * (u32)lane < spn + 8u) __builtin_amdgcn_global_load_lds((const ZPK_GLOBAL u32*)(g + 1024), (ZPK_LDS u32*)(d0 + 1024), 16, 0, 0);
                    next_staged = true;
                }
            } else load_pos(b0 + 64u, Pn, Nxn);
            if (b0 + 128u < nseq) load_pos(b0 + 128u, Pf, Nxf);
        }
        LXT(1);
        const u32 planned = cnt;
        bool act = (u32)lane < cnt;
        const u32 i = b0 + (u32)lane;
        // ---- decode this lane's token ----
        const u32 a = act ? P - p_first : 0u;
        const u32 t = act ? lds_ld8(S + a) : 0u;
        const bool is_last = act && i + 1 == nseq && final;
        u32 ll = t >> 4, q = a + 1;
        {
            const u32 e1 = lds_ld8(S + q);
            const bool ext = ll == 15;
            bool more = ext && e1 == 255;
            ll += ext ? e1 : 0u; q += ext ? 1u : 0u;
            while (more) {                               // 270+ literals: more extension bytes (rare)
                if (q >= span) { more = false; ll = LX_MAX_LL + 1; break; }
                const u32 b = lds_ld8(S + q); q++; ll += b;
                more = b == 255 && ll <= LX_MAX_LL;
            }
        }
        const u32 lit_a = q;
        u32 ml = 0, off = 1, seq_end = q + ll;
        {
            const u32 mo = q + ll < LX_STAGE + 32u ? q + ll : LX_STAGE + 32u;      // (a wild ll stays inside the LDS object; the check below rejects it)
            const u32 o16 = lds_ld8(S + mo) | (lds_ld8(S + mo + 1) << 8), e2 = lds_ld8(S + mo + 2);
            if (!is_last) {
                const u32 mlc = t & 15;
                const bool ext = mlc == 15;
                bool more = ext && e2 == 255;
                off = o16; ml = 4 + mlc + (ext ? e2 : 0u);
                u32 q2 = mo + 2 + (ext ? 1u : 0u);
                while (more) {
                    if (q2 >= span) { more = false; ml = LX_MAX_ML + 1; break; }
                    const u32 b = lds_ld8(S + q2); q2++; ml += b;
                    more = b == 255 && ml <= LX_MAX_ML;
                }
                seq_end = q2;
            }
        }
        if (!act) { ll = 0; ml = 0; }
        // the list verifies: every token ends where the next one starts (the last one at the end of the block), lengths in range
        if (__ballot(act && (seq_end != Nx - p_first || ll > LX_MAX_LL || ml > LX_MAX_ML)) != 0) return LX_E_TOKEN;
        LXT(2);
        // ---- execute: output positions, dependencies, copies into the ring, flush (lx_exec_batch) ----
        {
            const LxLitStage lits = { S, lit_a };
            const int rcb = lx_exec_batch(O, cnt, ll, ml, off, lits, hist_lo, dst_cap, lane, stt);
            if (rcb != LX_OK) return rcb;
            if (O.wp - block_out > 65536u) return LX_E_BLOCKMAX;
        }
        LXT(8);
        b0 += cnt;
        if (cnt == planned && rhythm && b0 < nseq) {                          // in rhythm: what was "next" is current now
            P = Pn; Nx = Nxn; Pn = Pf; Nxn = Nxf;
            staged = have_next && next_staged;
            c_cnt = have_next ? cn : 0u; c_pf = pfn; c_span = spn;
            buf ^= 1;
            have_next = b0 + 64u < nseq;
        } else if (b0 < nseq) {
            if (next_staged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // let the orphaned DMA land before its buffer is reused
            load_pos(b0, P, Nx); have_next = false; staged = false; c_cnt = 0;
        }
    }
    return LX_OK;
}

// whole frame; all arguments uniform.  rc != LX_OK: nothing about the entry is decided (partial output may have been written)
__device__ inline LxResult lz4f_exec_wave(Lz4ExecShared& sh, const u8* src, const u8* read_hi, u64 e_off, u64 e_size, const u8* tok,
                                          u8* dst, u64 dst_cap, u64 uncomp_size, int lane, u64* lx_dbg = nullptr)
{
    (void)lx_dbg;
    LxResult R; R.rc = LX_E_FRAME; R.produced = 0; R.hash = 0;
    if (dst_cap >= (1ull << 31) || uncomp_size >= (1ull << 31)) return R;       // positions are 32-bit here
    const u8* ip = src + e_off;
    const u8* const iend = ip + e_size;
    for (int guard = 0;; guard++) {
        if (iend - ip < 4 || guard > 16) return R;
        const u32 magic = uld32(ip);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (iend - ip < 8) return R;
            const u64 sz = uld32(ip + 4);
            if ((u64)(iend - ip) - 8 < sz) return R;
            ip += 8 + sz;
            continue;
        }
        if (magic != 0x184D2204u) return R;
        break;
    }
    if (iend - ip < 7) return R;
    const bool indep = (uld8(ip + 4) >> 5) & 1;
    ip += 7;
    LxOut O;
    lx_begin(O, to_lds_rw(sh.ring), dst, uncomp_size, lane);
#ifdef LX_STATS
    for (int k = 0; k < 12; k++) O.tm[k] = 0;
    O.t_last = __builtin_amdgcn_s_memtime();
    const u64 t_begin = O.t_last;
#endif
    for (;;) {
        if (iend - ip < 4) return R;
        const u32 bh = uld32(ip);
        ip += 4;
        if (bh == 0) break;
        const u32 bsz = bh & 0x7FFFFFFFu;
        if (bsz > 65536u || (u64)(iend - ip) < bsz) return R;
        int rc;
        if (bh >> 31) rc = lx_append_raw(O, ip, bsz, read_hi, dst_cap, lane);
        else {
            const u64 blk_abs = (u64)(ip - src);
            const u32 nsub = lx_nsub(bsz);
            u32 hist_lo = indep ? O.wp : 0u;
            if (O.wp - hist_lo > 65536u) hist_lo = O.wp - 65536u;
            const u32 block_out = O.wp;
            rc = LX_OK;
            // The block's chain = for every unit j: its PATCH (tokens between unit j - 1's exit and the token where unit j's own list
            // joins the chain; usually none), then its list from that token on.  part k = 2j: patch of unit j, k = 2j + 1: its list.
            auto part = [&](u32 k, const u8*& ptr, u32& cnt) -> bool {
                const u32 j = k >> 1, seg_lo = j * LX_SEG, seg_hi = lx_seg_hi(bsz, j);
                const u64 r0 = lx_sublist(blk_abs + seg_lo), r1 = lx_sublist(blk_abs + seg_hi);
                if (r0 + 16 > r1) return false;
                const u32 c = uld32(tok + r0), pc = uld32(tok + r1 - 8), v = uld32(tok + r1 - 4);
                const u64 lend = r0 + 8 + 2ull * ((c + 1) & ~1u), pbeg = r1 - 8 - 2ull * ((pc + 1) & ~1u);
                if (c > 21846u || pc > 21846u || v > c || lend > pbeg || pbeg < r0 + 8) return false;     // everything stays inside the unit's own region
                if (k & 1) { ptr = tok + r0 + 8 + 2ull * v; cnt = c - v; } else { ptr = tok + pbeg; cnt = pc; }
                return true;
            };
            bool first = true;
            for (u32 k = 0; k < 2 * nsub && rc == LX_OK; k++) {
                const u8* lst; u32 nseq;
                if (!part(k, lst, nseq)) { rc = LX_E_LIST; break; }
                if (nseq == 0) continue;
                // where this part's last token must end: the first token of the next part that has one, else the end of the block
                u32 end_pos = bsz;
                bool final = true;
                for (u32 k2 = k + 1; k2 < 2 * nsub; k2++) {
                    const u8* l2; u32 n2;
                    if (!part(k2, l2, n2)) { rc = LX_E_LIST; break; }
                    if (n2 != 0) { end_pos = uld16(l2); final = false; break; }
                }
                if (rc != LX_OK) break;
                if (first && uld16(lst) != 0) { rc = LX_E_LIST; break; }                             // a block's chain starts at its first byte
                first = false;
                rc = lx_block(sh, O, ip, bsz, read_hi, lst, nseq, (k >> 1) * LX_SEG, end_pos, final, block_out, hist_lo, dst_cap, lane);
            }
            if (rc == LX_OK && first) rc = LX_E_LIST;                                                // no token at all
        }
        if (rc != LX_OK) { R.rc = rc; return R; }
        ip += bsz;
    }
    if (ip != iend) return R;
    lx_finish(O, dst, uncomp_size, R, lane);
#ifdef LX_STATS
#ifndef LX_STATS_SCAN_ONLY
    if (lx_dbg && lane == 0) { for (int k = 0; k < 12; k++) lx_dbg[k] = O.tm[k]; lx_dbg[12] = __builtin_amdgcn_s_memtime() - t_begin; }
#endif
#endif
    R.rc = LX_OK;
    return R;
}

}  // namespace zpk
