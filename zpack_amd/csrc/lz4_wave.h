// lz4_wave.h — LZ4 frame + block decode: ONE WAVE PER ENTRY, output window = the entry's own output slot in
// global memory.  Handles the whole LZ4 Frame format (any block size, linked or independent blocks, stored
// blocks, optional header / block / content checksums, content size, skippable frames).
//
// Replaces the LZ4F_decompress loop of the reference (lib/zpack_read.c:414-439).
//
// Frame and block headers are read wave-uniformly; a compressed block is parsed lane-parallel out of an LDS
// stage (speculative token-chain walks to a fixed point, see below) and executed 64 sequences at a time by
// seq_exec_batch() (seq_exec.h).  The ByteWindow register window below serves the Zstandard bit readers.
#pragma once
#include "zpk_device.h"
#include "xxh3_device.h"
#include "seq_exec.h"

namespace zpk {

// the window word of one lane: bytes [a, a+4) clipped to [lo, hi).  Out of line on purpose: the bit readers
// inline into dozens of call sites, and this rarely-taken body (once per ~240 stream bytes) was most of the
// Zstandard kernel's code size.
__device__ __noinline__ u32 window_word(const u8* a, const u8* lo, const u8* hi)
{
    if (a >= lo && a + 4 <= hi) return ld32(a);
    u32 v = 0;
    #pragma unroll
    for (int i = 0; i < 4; i++) if (a + i >= lo && a + i < hi) v |= (u32)ld8(a + i) << (8 * i);
    return v;
}

// 256-byte sliding register window over a read-only byte stream
struct ByteWindow {
    u32 w;              // lane l holds bytes [base + 4l, base + 4l + 4)
    const u8* base;     // 4-byte aligned, uniform
    const u8* lo;       // readable range of the underlying buffer (uniform)
    const u8* hi;

    __device__ __forceinline__ void load(const u8* p, int lane)
    {
        base = (const u8*)((u64)p & ~(u64)3);
        w = window_word(base + 4 * lane, lo, hi);
    }
    // uniform byte at uniform address p (refills when p leaves the window)
    __device__ __forceinline__ u32 byte(const u8* p, int lane)
    {
        u64 d = (u64)(p - base);
        if (d >= 256) { load(p, lane); d = (u64)(p - base); }
        u32 word = (u32)__builtin_amdgcn_readlane((int)w, (int)(d >> 2));
        return (word >> (8 * (d & 3))) & 0xFF;
    }
};

struct DecodeOut { int rc; u64 produced; };

// ---- LZ4 block decode: lane-parallel parse out of a per-wave LDS chunk, batched execution ----------
//
// The token chain is serial, and walking it wave-uniformly costs ~1000 cycles per sequence (a long
// dependent scalar chain; measured).  Instead the block is taken in chunks of 64 segments x LZ4W_SEG
// bytes staged in LDS: every lane walks its own segment from a guessed entry, then re-walks from its
// predecessor's exit until no entry changes (LZ4 chains re-synchronise within a few tokens; lane 0's
// entry is the true chain position, so the fixed point is the true chain).  A DPP prefix sum numbers
// the chunk's sequences, every lane lists its token positions (the set bits of its visited mask) in a u16
// LDS list, and the chunk is then executed 64 consecutive sequences at a time: lane k re-reads token k's
// fields from LDS and seq_exec_batch() (seq_exec.h) does the copies, assembling the batch in the part of
// the stage the chunk has already consumed when it fits.  LDS per wave: a 3.9 KiB stage (60-byte segments;
// 68 measured equal, 84/108 slower) + the 1.1 KiB list.
#ifndef LZ4W_SEG
#define LZ4W_SEG 60u                            // 15 dwords: odd, so 64 lanes spread over all LDS banks; <= 64 for the visited mask
#endif
static_assert(LZ4W_SEG <= 64u, "the visited-position mask of a segment is one 64-bit word");
#define LZ4W_CHUNK (64u * LZ4W_SEG)
#define LZ4W_SLACK 64u
#ifndef LZ4W_RUNIN
#define LZ4W_RUNIN LZ4W_SEG                       // bytes of run-in before a segment's speculative walk
#endif

#ifndef LZ4W_NREC
#define LZ4W_NREC 576u                           // token-position list: 9 batches of 64 sequences.  Measured: total LDS <= 5 KiB
                                                 // per wave keeps 28 waves per CU (482 GiB/s); 5.5 KiB loses waves (451 GiB/s)
#endif
struct alignas(16) Lz4WaveShared {
    u8  stage[LZ4W_CHUNK + LZ4W_SLACK + 16];     // + 16: literal runs are read 16 bytes at a time
    u16 rec[LZ4W_NREC];                          // chunk-relative token position of sequence (b0 window + i)
};

// byte `pos` of the block: from the staged chunk when it is there, else from memory (only a sequence
// whose literal run crosses the end of the chunk gets there)
struct Lz4Bytes {
    lds_cp8 S;            // staged chunk (LDS, typed: ds_read, never a flat load)
    const u8* g;          // block base in memory
    u32 cbase, cend;      // the chunk holds block bytes [cbase, cend)
    __device__ __forceinline__ u32 at(u32 pos) const
    {
        if (pos < cend) return lds_ld8(S + (pos - cbase));
        return (u32)ld8(g + pos);
    }
    __device__ __forceinline__ u32 at16(u32 pos) const
    {
        if (pos + 1 < cend) return lds_ld16(S + (pos - cbase));
        return at(pos) | (at(pos + 1) << 8);
    }
};

// flags: 1 = malformed (| 4: found AFTER the literal run, i.e. the reference copies the literals first), 2 = this was the
// block's last (literal-only) sequence
struct Lz4Tok { u32 next, lit_pos, lit, ml, off, flags; };

// full decode of the sequence whose token is at block position p (p < C)
__device__ __forceinline__ Lz4Tok lz4_token_at(const Lz4Bytes& B, u32 p, u32 C, bool want_offset)
{
    Lz4Tok t; t.flags = 0; t.ml = 0; t.off = 1;
    const u32 tok = B.at(p);
    u32 q = p + 1, lit = tok >> 4;
    if (lit == 15) {
        u32 b;
        do {
            if (q >= C) { t.flags = 1; t.next = C; t.lit = 0; t.lit_pos = q; return t; }
            b = B.at(q++); lit += b;
        } while (b == 255 && lit < 0x7F000000u);
        if (b == 255) { t.flags = 1; t.next = C; t.lit = 0; t.lit_pos = q; return t; }
    }
    t.lit_pos = q; t.lit = lit;
    if (lit > C - q) { t.flags = 1; t.next = C; return t; }
    q += lit;
    if (q == C) { t.flags = 2; t.next = C; return t; }
    if (C - q < 2) { t.flags = 5; t.next = C; return t; }
    if (want_offset) t.off = B.at16(q);
    q += 2;
    u32 ml = tok & 15;
    if (ml == 15) {
        u32 b;
        do {
            if (q >= C) { t.flags = 5; t.next = C; return t; }
            b = B.at(q++); ml += b;
        } while (b == 255 && ml < 0x7F000000u);
        if (b == 255) { t.flags = 5; t.next = C; return t; }
    }
    t.ml = ml + 4;
    t.next = q;
    return t;
}

// The common shapes of a sequence — at most one length-extension byte on either side, everything inside
// the staged chunk and strictly inside the block — evaluated from LDS without loops.  Returns false when
// the general decoder (lz4_token_at) has to look at it; a divergent slow path costs every lane of the
// wave, so the point is to keep ordinary long matches (ml >= 19) out of it.
template <bool FULL>
__device__ __forceinline__ bool lz4_token_fast(const Lz4Bytes& B, u32 p, u32 C, u32 tok, Lz4Tok& t)
{
    const u32 lim = B.cend < C ? B.cend : C;       // bytes [p, lim) are staged and inside the block
    u32 q = p + 1, lit = tok >> 4, ml = tok & 15;
    bool ok = true;
    if (lit == 15) {
        if (q >= lim) return false;
        const u32 b = lds_ld8(B.S + (q - B.cbase));
        ok = b != 255; lit += b; q++;
    }
    t.lit_pos = q; t.lit = lit;
    if (lit >= lim - q) return false;              // also keeps q + lit below any wrap
    q += lit;
    if (q + 3 > lim) return false;                 // room for the offset and one extension byte
    if (FULL) t.off = lds_ld16(B.S + (q - B.cbase));
    q += 2;
    if (ml == 15) {
        const u32 b = lds_ld8(B.S + (q - B.cbase));
        ok = ok && b != 255; ml += b; q++;
    }
    t.ml = ml + 4; t.next = q; t.flags = 0;
    return ok;
}

// Walk from `entry` while the token lies before seg_end.  The set of token positions visited (relative
// to seg_start, < LZ4W_SEG <= 64) is kept as a bit mask; a re-walk from a new entry stops as soon as it
// lands on a position the previous walk already visited — from there on the two chains are identical —
// and inherits the tail.  flags (1 malformed, 2 last sequence of the block) always belong to a walk's
// final hop, so they travel with the tail.  The number of sequences is the mask's population count.
//
// INTERIOR = the block continues for at least LZ4W_SLACK staged bytes past the last token of the chunk:
// a sequence with a short literal run (< 15) then needs no bounds check at all, and the hop is straight
// -line code (two LDS reads, no exec-mask juggling) — the kernel is VALU/SALU-issue bound, and the hop is
// its most executed piece.
struct Lz4Walk { u32 exit, flags; u64 m; };
template <bool INTERIOR>
__device__ __forceinline__ Lz4Walk lz4_walk(const Lz4Bytes& B, u32 entry, u32 seg_start, u32 seg_end, u32 C, const Lz4Walk& old)
{
    Lz4Walk w; w.flags = 0; w.m = 0;
    u32 p = entry;
    bool merged = false;
    while (p < seg_end) {
        const u32 r = p - seg_start;
        if ((old.m >> r) & 1ull) { merged = true; break; }
        w.m |= 1ull << r;
        // tokens of a chunk always lie inside the staged range: one ds_read, no fallback
        const lds_cp8 at = B.S + (p - B.cbase);
        const u32 tok = lds_ld8(at);
        u32 fl = 0, nx;
        bool slow;
        Lz4Tok t;
        if (INTERIOR) {
            const u32 lit = tok >> 4, ext = (tok & 15) == 15 ? 1u : 0u;
            const u32 b = lds_ld8(at + 3 + lit);            // first match-length extension byte, if there is one
            nx = p + 3 + lit + ext;
            slow = lit == 15 || (ext && b == 255);
        } else {
            slow = !lz4_token_fast<false>(B, p, C, tok, t);
            nx = t.next;
        }
        if (slow) {
            if (!INTERIOR || !lz4_token_fast<false>(B, p, C, tok, t)) t = lz4_token_at(B, p, C, false);
            fl = t.flags; nx = t.next;
        }
        w.flags |= fl;
        p = nx;
        if (fl) break;
    }
    w.exit = p;
    if (merged) {
        const u64 tail = old.m & ~((1ull << (p - seg_start)) - 1);
        w.m |= tail;
        w.flags |= old.flags;
        w.exit = old.exit;
    }
    return w;
}

// Run-in of a speculative walk: hop from `p` (somewhere before seg_start) until the chain position reaches
// seg_start, without recording anything.  A chain started at an arbitrary byte has usually joined the true
// chain after a segment's worth of hops, so the position it enters the segment at is the true entry for ~96 %
// of the segments (text) instead of ~0 % for "start at the segment boundary" — which turns the first, always
// needed, all-lanes fix-up round into a check.  Anything odd (end of block, malformed parse) just gives up
// and returns seg_start.
template <bool INTERIOR>
__device__ __forceinline__ u32 lz4_run_in(const Lz4Bytes& B, u32 p, u32 seg_start, u32 C)
{
    while (p < seg_start) {
        const lds_cp8 at = B.S + (p - B.cbase);
        const u32 tok = lds_ld8(at);
        u32 nx; bool slow; Lz4Tok t;
        if (INTERIOR) {
            const u32 lit = tok >> 4, ext = (tok & 15) == 15 ? 1u : 0u;
            const u32 b = lds_ld8(at + 3 + lit);
            nx = p + 3 + lit + ext;
            slow = lit == 15 || (ext && b == 255);
        } else { slow = true; nx = p; }
        if (slow) {
            if (!lz4_token_fast<false>(B, p, C, tok, t)) t = lz4_token_at(B, p, C, false);
            if (t.flags) return seg_start;
            nx = t.next;
        }
        p = nx;
    }
    return p;
}

// one LZ4 block: src [ip, ip+C) -> dst [op, ...), cap = oend; dst_lo = lowest output address a match may reach
__device__ inline int lz4_block_wave(Lz4WaveShared& sh, Watchdog& wd, SeqStats& stt, const u8* ip, u32 C, const u8* rd_hi,
                                     u8* dst_lo, u8*& op_io, u8* oend, int lane)
{
    u8* op = op_io;
    if (C == 0) return D_MALFORMED;
    u32 cpos = 0;                                   // block position of the next token of the true chain
    bool finished = false;
    while (!finished) {
        if (wd.expired()) return D_MALFORMED;
        const u64 tp0 = SEQ_T(); (void)tp0;
        if (cpos >= C) return D_MALFORMED;          // the chain ran off the block without a final literal run
        // ---- stage [cpos, cpos + nst) ----
        const u32 nst = C - cpos < LZ4W_CHUNK + LZ4W_SLACK ? C - cpos : LZ4W_CHUNK + LZ4W_SLACK;
        const u32 tok_end = C - cpos < LZ4W_CHUNK ? C : cpos + LZ4W_CHUNK;      // tokens of this chunk lie before tok_end
        wave_mem_fence();
        if (ip + cpos + ((nst + 15u) & ~15u) <= rd_hi) {
            // the whole chunk by LDS-DMA (global_load_lds, 16 bytes per lane, no registers): up to four loads in flight, ONE wait — the
            // register form below waits for every 1 KiB before it asks for the next
            const u8* const g = ip + cpos + 16u * (u32)lane;
            ZPK_LDS u8* const d0 = (ZPK_LDS u8*)sh.stage;
            #pragma unroll
            for (u32 i = 0; i < LZ4W_CHUNK + LZ4W_SLACK; i += WAVE * 16)
                if (i + 16u * (u32)lane < nst) __builtin_amdgcn_global_load_lds((const ZPK_GLOBAL u32*)(g + i), (ZPK_LDS u32*)(d0 + i), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (u32 i = (u32)lane * 16; i < nst; i += WAVE * 16) {
                const u8* g = ip + cpos + i;
                if (i + 16 <= nst && g + 16 <= rd_hi) { u128 v = ld128(g); __builtin_memcpy(sh.stage + i, &v, 16); }
                else for (u32 k = i; k < nst && k < i + 16; k++) sh.stage[k] = ld8(ip + cpos + k);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        u64 ts = SEQ_T(); (void)ts; SEQ_STAT(stt.t_stage += ts - tp0; stt.chunks++);
        Lz4Bytes B; B.S = to_lds(sh.stage); B.g = ip; B.cbase = cpos; B.cend = cpos + nst;
        // ---- walks to the fixed point ----
        const u32 my_start = cpos + (u32)lane * LZ4W_SEG;
        const u32 my_end = my_start + LZ4W_SEG < tok_end ? my_start + LZ4W_SEG : tok_end;
        const bool active = my_start < tok_end;
        u32 my_entry = my_start;
        const bool interior = nst == LZ4W_CHUNK + LZ4W_SLACK;      // uniform: every chunk but the last of a block
        Lz4Walk w; w.exit = my_start; w.flags = 0; w.m = 0;
        if (active && lane != 0) {                                 // lane 0 starts on the true chain
            const u32 from = my_start - LZ4W_RUNIN;                // inside the previous lane's segment (lane >= 1)
            my_entry = interior ? lz4_run_in<true>(B, from, my_start, C) : lz4_run_in<false>(B, from, my_start, C);
        }
        if (active) w = interior ? lz4_walk<true>(B, my_entry, my_start, my_end, C, w) : lz4_walk<false>(B, my_entry, my_start, my_end, C, w);
        SEQ_STAT({ u64 t2 = SEQ_T(); stt.t_walk1 += t2 - ts; ts = t2; });
        for (int iter = 0; iter < 66; iter++) {
            SEQ_STAT(stt.fix_iters++);
            u32 e = (u32)__shfl_up((int)w.exit, 1, 64);
            if (lane == 0) e = cpos;
            const bool changed = active && e != my_entry;
            if (__ballot(changed) == 0) break;
            if (changed) {
                my_entry = e;
                w = interior ? lz4_walk<true>(B, my_entry, my_start, my_end, C, w) : lz4_walk<false>(B, my_entry, my_start, my_end, C, w);
            }
        }
        SEQ_STAT({ u64 t2 = SEQ_T(); stt.t_fix += t2 - ts; ts = t2; });
        // inactive lanes forward the chain position
        if (!active) { w.exit = (u32)__shfl((int)w.exit, 63, 64); }
        const u64 lastmask = __ballot(active);
        const int last_lane = 63 - __clzll((long long)lastmask);
        const u32 chain_exit = (u32)__builtin_amdgcn_readlane((int)w.exit, last_lane);
        const u32 fl = (u32)__ballot(active && (w.flags & 1)) != 0 || (__ballot(active && (w.flags & 1)) >> 32) != 0 ? 1u : 0u;
        // (a malformed token ends the chain; it is the chunk's last listed sequence, and the executor below gives the
        // verdict of whichever sequence offends FIRST, in stream order, like the reference's serial decoder)
        if (__ballot(active && (w.flags & 2)) != 0) finished = true;          // the block's last sequence is in this chunk
        // ---- sequence numbers + token records ----
        const u32 my_nseq = active ? (u32)__popcll(w.m) : 0u;
        const u32 x = wave_scan_add(my_nseq);
        const u32 nseq = (u32)__builtin_amdgcn_readlane((int)x, 63);
        SEQ_STAT({ u64 t2 = SEQ_T(); stt.t_emit += t2 - ts; stt.t_parse += t2 - tp0; });
        // ---- execute, 64 consecutive sequences at a time ----
        // Every segment lists the token positions of its sequences (chunk-relative, 16 bit) at their sequence
        // numbers in LDS — a short loop over the set bits of its mask — so that a batch lane finds its token with
        // one ds_read.  The list holds LZ4W_NREC entries; a chunk with more sequences is listed in windows.
        ZPK_LDS u16* const rec = (ZPK_LDS u16*)sh.rec;
        for (u32 b0 = 0; b0 < nseq; b0 += WAVE) {
            if (b0 % LZ4W_NREC == 0) {
                wave_mem_fence();
                u64 m = w.m;
                u32 k = x - my_nseq;                                           // number of this lane's first sequence
                const u32 rel = my_start - cpos;
                while (m) {
                    const u32 b = (u32)__ffsll((long long)m) - 1u;
                    m &= m - 1;
                    if (k - b0 < LZ4W_NREC) rec[k - b0] = (u16)(rel + b);      // k < b0 wraps to a huge value
                    k++;
                }
                wave_mem_fence();                                              // LDS is in order within a wave
            }
            const int cnt = (int)(nseq - b0 < WAVE ? nseq - b0 : WAVE);
            const u64 tq0 = SEQ_T(); (void)tq0;
            SeqBatch q; q.lit = ip; q.lit_lds = SEQ_NO_LDS; q.ll = 0; q.ml = 0; q.off = 1; q.bad = 0;
            const u32 sq = b0 + (u32)lane;
            u32 tok_pos = cpos;
            if (lane < cnt) {
                const u32 p = cpos + (u32)rec[sq % LZ4W_NREC];
                tok_pos = p;
                const lds_cp8 at = B.S + (p - B.cbase);
                const u32 tok = lds_ld8(at);
                const u32 lit = tok >> 4, mlc = tok & 15;
                u32 o16 = 0, eb = 255;
                if (interior) { o16 = lds_ld16(at + 1 + lit); eb = lds_ld8(at + 3 + lit); }   // unconditional: in range, maybe unused
                Lz4Tok t;
                if (interior && lit != 15 && !(mlc == 15 && eb == 255)) {            // straight-line common case
                    q.lit = ip + p + 1; q.ll = lit; q.ml = mlc + 4 + (mlc == 15 ? eb : 0u); q.off = o16;
                } else if (lz4_token_fast<true>(B, p, C, tok, t)) {                         // common case: 2-4 LDS reads in all
                    q.lit = ip + t.lit_pos; q.ll = t.lit; q.ml = t.ml; q.off = t.off;
                } else {
                    t = lz4_token_at(B, p, C, true);
                    q.lit = ip + t.lit_pos; q.ll = t.lit; q.ml = t.ml; q.off = t.off;
                    if (t.flags & 1) q.bad = (t.flags & 4) ? 2u : 1u;
                }
                // (a block holds at most 4 MiB of output: longer lengths only have to stay longer than that, and 64 of them
                // must not wrap the 32-bit prefix sums)
                if (q.ll > (1u << 23)) q.ll = 1u << 23;            // 64 x 2 x 2^23 = 2^30: the executor's 32-bit arithmetic holds
                if (q.ml > (1u << 23)) q.ml = 1u << 23;
                const u32 lp = (u32)(q.lit - ip);                                      // literals that sit in the staged chunk
                if (q.ll <= SEQ_OWN_MAX && lp + q.ll <= B.cend) q.lit_lds = lp - B.cbase;
            }
            SEQ_STAT({ u64 t2 = SEQ_T(); stt.t_parse += t2 - tq0; stt.t_tok += t2 - tq0; });
            // Input this chunk has already consumed (everything before the batch's first token) is dead: the
            // executor assembles the batch there.  The first batch or two of a chunk find too little room and
            // take the direct path.
            const u32 dead = (u32)__builtin_amdgcn_readfirstlane((int)tok_pos) - cpos;
            const int rc = seq_exec_batch<true>(q, cnt, op, oend, dst_lo, -1, lane, stt, B.S, to_lds_rw(sh.stage), dead);
            if (rc != D_OK) { op_io = op; return rc; }
        }
        if (fl) return D_MALFORMED;                                            // (not reached: the executor saw the malformed token)
        if (!finished && chain_exit <= cpos) return D_MALFORMED;               // no progress: cannot happen
        cpos = chain_exit;
    }
    if (cpos != C) return D_MALFORMED;
    op_io = op;
    return D_OK;
}

// whole frame.  src_lo/src_hi bound what may be READ (the archive image); all values uniform.
__device__ inline DecodeOut lz4f_decode_wave(Lz4WaveShared& sh, Watchdog& wd, SeqStats& stt, const u8* src, u64 src_size, const u8* src_lo, const u8* src_hi,
                                             u8* dst, u64 dst_cap, int lane)
{
    DecodeOut r; r.rc = D_OK; r.produced = 0;
    const u8* ip = src;
    const u8* iend = src + src_size;
    u8* op = dst;
    u8* oend = dst + dst_cap;

    for (;;) {      // skippable frames
        if (iend - ip < 4) { r.rc = D_TRUNCATED; return r; }
        u32 magic = uld32(ip);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (iend - ip < 8) { r.rc = D_TRUNCATED; return r; }
            u64 sz = uld32(ip + 4);
            if ((u64)(iend - ip) - 8 < sz) { r.rc = D_TRUNCATED; return r; }
            ip += 8 + sz;
            continue;
        }
        if (magic != 0x184D2204u) { r.rc = D_MALFORMED; return r; }
        break;
    }
    if (iend - ip < 7) { r.rc = D_TRUNCATED; return r; }
    const u32 flg = uld8(ip + 4), bd = uld8(ip + 5);
    const bool indep = (flg >> 5) & 1, bck = (flg >> 4) & 1, has_cs = (flg >> 3) & 1, cck = (flg >> 2) & 1, has_dict = flg & 1;
    if ((flg >> 6) != 1 || (flg & 2) || (bd & 0x8F)) { r.rc = D_MALFORMED; return r; }
    const u32 bcode = (bd >> 4) & 7;
    if (bcode < 4) { r.rc = D_MALFORMED; return r; }
    const u64 bmax = (u64)1 << (8 + 2 * bcode);            // 4:64K 5:256K 6:1M 7:4M
    const u32 hdr = 7 + (has_cs ? 8 : 0) + (has_dict ? 4 : 0);
    if ((u64)(iend - ip) < hdr) { r.rc = D_TRUNCATED; return r; }
    u64 content_size = 0;
    if (has_cs) content_size = uni64(ld64(ip + 6));
    {
        u32 h = 0;
        lane0_guard();
        if (lane == 0) h = (xxh32_serial(ip + 4, hdr - 5, 0) >> 8) & 0xFF;
        if (uni(h) != uld8(ip + hdr - 1)) { r.rc = D_MALFORMED; return r; }
    }
    ip += hdr;

    u8* frame_out = op;
    for (;;) {
        if (wd.expired()) { r.rc = D_MALFORMED; return r; }
        if (iend - ip < 4) { r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r; }
        const u32 bh = uld32(ip);
        ip += 4;
        if (bh == 0) break;
        const bool raw = bh >> 31;
        const u64 bsz = bh & 0x7FFFFFFFu;
        if (bsz > bmax) { r.rc = D_MALFORMED; return r; }
        if ((u64)(iend - ip) < bsz + (bck ? 4u : 0u)) {
            if (raw) {      // what is there of a stored block still streams out before the input starves
                u64 n = (u64)(iend - ip); if (n > bsz) n = bsz;
                bool full = n > (u64)(oend - op);
                if (full) n = (u64)(oend - op);
                for (u64 i = (u64)lane * 16; i < n; i += WAVE * 16) gcopy_upto16(op + i, ip + i, (u32)(n - i < 16 ? n - i : 16));
                op += n;
                if (full) { r.rc = D_DST_FULL; r.produced = (u64)(op - dst); return r; }
            }
            r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r;
        }
        if (bck) {
            u32 h = 0;
            lane0_guard();
            if (lane == 0) h = xxh32_serial(ip, bsz, 0);
            if (uni(h) != uld32(ip + bsz)) { r.rc = D_MALFORMED; return r; }
        }
        if (raw) {
            u64 n = bsz;
            bool full = n > (u64)(oend - op);
            if (full) n = (u64)(oend - op);
            for (u64 i = (u64)lane * 16; i < n; i += WAVE * 16) gcopy_upto16(op + i, ip + i, (u32)(n - i < 16 ? n - i : 16));
            op += n;
            if (full) { r.rc = D_DST_FULL; r.produced = (u64)(op - dst); return r; }
        } else {
            u8* hist_lo = indep ? op : frame_out;
            if ((u64)(op - hist_lo) > 65536) hist_lo = op - 65536;
            u8* bend = oend;
            bool limited = false;
            if ((u64)(oend - op) > bmax) { bend = op + bmax; limited = true; }
            int rc = lz4_block_wave(sh, wd, stt, ip, (u32)bsz, src_hi, hist_lo, op, bend, lane);
            if (rc == D_DST_FULL) {
                if (limited) { r.rc = D_MALFORMED; return r; }
                r.rc = D_DST_FULL; r.produced = (u64)(op - dst); return r;
            }
            if (rc != D_OK) { r.rc = D_MALFORMED; return r; }
        }
        wave_mem_fence();
        ip += bsz + (bck ? 4u : 0u);
    }
    if (cck) {
        if (iend - ip < 4) { r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r; }
        u32 h = 0;
        lane0_guard();
        if (lane == 0) h = xxh32_serial(frame_out, (u64)(op - frame_out), 0);
        if (uni(h) != uld32(ip)) { r.rc = D_MALFORMED; return r; }
        ip += 4;
    }
    if (has_cs && content_size != (u64)(op - frame_out)) { r.rc = D_MALFORMED; return r; }
    r.produced = (u64)(op - dst);
    return r;
}

}  // namespace zpk
