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

// Streaming (zpk_stream.inc): a frame decode that runs out of INPUT at a block boundary can be picked up again when more bytes have
// arrived.  LZ4 keeps no state between blocks but the positions (the frame header is parsed again, it is a few bytes).
struct Lz4Resume { u64 ip_off, op_off, frame_ip_off, frame_op_off; u32 in_blocks, nframes; };      // (zero-initialised = the start of the entry)

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
                                                 // per wave keeps the waves the register budget allows (32 per CU at 64 VGPRs; 5.5 KiB loses waves: 451 vs 482 GiB/s in round 1)
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

// The common shapes of a sequence — at most one length-extension byte on either side (literal runs below 270, matches
// below 274), offset and extension byte before `lim` (the end of the staged bytes or of the block, whichever comes
// first) — decoded WITHOUT a branch: three or four LDS reads and selects.  `slow` says the general decoder
// (lz4_token_at) has to look at the sequence; then nothing else of the result means anything.  The kernel is bound by
// instruction issue — by the total number of instructions, scalar and vector: a divergent `if` costs
// three scalar instructions per level whether or not any lane takes it, and with 64 lanes some lane nearly always took
// the old slow path of a 15-byte literal run.  Reads reach at most 275 bytes past the token: inside Lz4WaveShared for
// every token of a chunk.
struct Lz4Quick { u32 lit_pos, lit, ml, next, off, slow; };
// (here the predicates are 0/1 INTEGERS computed with shifts: a comparison result is a lane mask in scalar registers and
// every `&&` / `||` of two of them is a scalar instruction.  lz4_walk below keeps lane masks instead — there the mask
// arithmetic replaces about as many vector instructions, measured equal; all values are far below 2^31)
__device__ __forceinline__ u32 lt31(u32 a, u32 b) { return (a - b) >> 31; }          // a < b
__device__ __forceinline__ Lz4Quick lz4_quick(lds_cp8 S, u32 cbase, u32 p, u32 lim)
{
    const lds_cp8 at = S + (p - cbase);
    const u32 tok = lds_ld8(at), b1 = lds_ld8(at + 1);
    const u32 lit4 = tok >> 4, mlc = tok & 15;
    const u32 le = (lit4 + 1) >> 4, me = (mlc + 1) >> 4;          // 1 = a length-extension byte follows
    const u32 lit = lit4 + (b1 & (0u - le));
    const u32 q = 1 + le + lit;                                    // the offset field, relative to p
    const u32 b2 = lds_ld8(at + q + 2);                            // first match-length extension byte, if there is one
    Lz4Quick r;
    r.off = lds_ld16(at + q);
    r.lit_pos = p + 1 + le; r.lit = lit;
    r.ml = mlc + 4 + (b2 & (0u - me));
    r.next = p + q + 2 + me;
    r.slow = (le & ((b1 + 1) >> 8)) | (me & ((b2 + 1) >> 8)) | lt31(lim, p + q + 3);
    return r;
}

// Walk the token chain from `from` while the token lies before seg_end.  The set of token positions visited (relative
// to seg_start, < LZ4W_SEG <= 64) is kept as a bit mask; a re-walk from a new entry stops as soon as it lands on a
// position the previous walk already visited — from there on the two chains are identical — and inherits the tail.
// flags (1 malformed, 2 last sequence of the block) always belong to a walk's final hop, so they travel with the tail.
// The number of sequences is the mask's population count.
//
// RUN-IN: a speculative first walk starts at `from` < seg_start (inside the previous lane's segment) and records
// nothing before seg_start.  A chain started at an arbitrary byte has usually joined the true chain after a segment's
// worth of hops, so the position it enters the segment at (`entry`) is the true entry for ~96 % of the segments
// (text) instead of ~0 % for "start at the segment boundary".  Anything odd on the way (end of block, malformed
// parse) just restarts at seg_start.
//
// The loop is WAVE-UNIFORM: every lane runs the same instructions until no lane is live, per-lane state is selected,
// not branched on (`on` = the lane takes part at all; the others get `old` back).  Only the rare general-decoder case
// is a divergent branch.
struct Lz4Walk { u32 exit, flags; u64 m; };
template <bool FIRST>
__device__ __forceinline__ Lz4Walk lz4_walk(const Lz4Bytes& B, bool on, u32 from, u32 seg_start, u32 seg_end, u32 C,
                                            const Lz4Walk& old, u32& entry_out, SeqStats& stt)
{
    (void)stt;
    // positions relative to seg_start (negative = run-in); a lane that is not live keeps a position inside the staged
    // chunk, so its (unused) LDS reads stay inside Lz4WaveShared; an exit far outside the chunk travels in exit_far —
    // also that of a chain which enters beyond the segment (a long literal run or match jumped over it): nothing to walk
    const u32 lim = B.cend < C ? B.cend : C;
    const i32 seg_len = (i32)(seg_end - seg_start), lim_rel = (i32)(lim - seg_start);
    const lds_cp8 base = B.S + (seg_start - B.cbase);
    const bool start_in = on && (i32)(from - seg_start) < seg_len;
    i32 rp = start_in ? (i32)(from - seg_start) : seg_len;
    bool live = start_in, merged = false;
    u32 flags = 0, exit_far = from, far = on && !start_in ? 1u : 0u;
    u64 m = 0;
    for (;;) {
        live = live && rp < seg_len;
        const bool inseg = FIRST ? rp >= 0 : true;
        if (!FIRST) {                                               // (a first walk has nothing to merge with)
            const bool hit = live && ((u32)(old.m >> (rp & 63)) & 1u);
            merged = merged || hit;
            live = live && !hit;
        }
        if (__builtin_amdgcn_ballot_w64(live) == 0) break;
        SEQ_STAT(if (FIRST) stt.hops_first++; else stt.hops_fix++);
        m |= (u64)((live && inseg) ? 1u : 0u) << (rp & 63);
        // the common shapes, without a branch (see lz4_quick)
        const lds_cp8 at = base + rp;
        const u32 tok = lds_ld8(at), b1 = lds_ld8(at + 1);
        const u32 lit4 = tok >> 4, mlc = tok & 15;
        const bool l15 = lit4 == 15, m15 = mlc == 15;
        const u32 q = lit4 + 1 + (l15 ? b1 + 1 : 0u);               // the offset field, relative to the token
        const u32 b2 = lds_ld8(at + q + 2);
        const i32 n2 = rp + (i32)q + 2;
        i32 nx = n2 + (m15 ? 1 : 0);
        const bool slow = (l15 && b1 == 255) || (m15 && b2 == 255) || n2 >= lim_rel;
        if (live && slow) {                                         // rare, divergent
            SEQ_STAT(stt.slow_hops++);
            const Lz4Tok tt = lz4_token_at(B, seg_start + (u32)rp, C, false);
            nx = (i32)(tt.next - seg_start);
            if (tt.flags && !inseg) nx = 0;                         // an odd run-in gives up: walk from the segment boundary
            else if (tt.flags || nx >= seg_len) { flags |= tt.flags; far = 1; exit_far = tt.next; nx = seg_len; }
        }
        rp = live ? nx : rp;
    }
    Lz4Walk w; w.m = m; w.flags = flags; w.exit = far ? exit_far : seg_start + (u32)rp;
    if (merged) {
        const u64 tail = old.m & ~((1ull << (rp & 63)) - 1);
        w.m |= tail;
        w.flags |= old.flags;
        w.exit = old.exit;
    }
    if (!on) w = old;
    entry_out = !FIRST ? from : m ? seg_start + (u32)__ffsll((long long)m) - 1u : w.exit;   // where the chain enters the segment
    return w;
}

// EMIT (lz4_pj.h, the block-parallel decoder of large frames): the block is only PARSED — every batch of up to 64 sequences goes to
// `emit(q, cnt, literal position in the block)` instead of the executor, nothing is read from or written to the output.
struct Lz4NoEmit { __device__ __forceinline__ int operator()(const SeqBatch&, int, u32) const { return D_OK; } };

// one LZ4 block: src [ip, ip+C) -> dst [op, ...), cap = oend; dst_lo = lowest output address a match may reach
// COOP: seq_exec.h (0 = k_lz4_wave's build, 2 = the build with grouped cooperative copies)
template <int COOP = 2, bool EMIT = false, class Emit = Lz4NoEmit>
__device__ inline int lz4_block_wave(Lz4WaveShared& sh, Watchdog& wd, SeqStats& stt, const u8* ip, u32 C, const u8* rd_hi,
                                     u8* dst_lo, u8*& op_io, u8* oend, int lane, Emit emit = Emit())
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
        const u32 lim = B.cend < C ? B.cend : C;                   // sequences wholly before lim need no bounds checks
        // ---- walks to the fixed point ----
        const u32 my_start = cpos + (u32)lane * LZ4W_SEG;
        const u32 my_end = my_start + LZ4W_SEG < tok_end ? my_start + LZ4W_SEG : tok_end;
        const bool active = my_start < tok_end;
        u32 my_entry = my_start;
        Lz4Walk w; w.exit = my_start; w.flags = 0; w.m = 0;
        // lane 0 starts on the true chain, the others speculate from inside the previous lane's segment
        w = lz4_walk<true>(B, active, lane == 0 ? my_start : (my_start - cpos > LZ4W_RUNIN ? my_start - LZ4W_RUNIN : cpos), my_start, my_end, C, w, my_entry, stt);
        SEQ_STAT({ u64 t2 = SEQ_T(); stt.t_walk1 += t2 - ts; ts = t2; });
        for (int iter = 0; iter < 66; iter++) {
            SEQ_STAT(stt.fix_iters++);
            u32 e = (u32)__shfl_up((int)w.exit, 1, 64);
            if (lane == 0) e = cpos;
            const bool changed = active && e != my_entry;
            if (__ballot(changed) == 0) break;
            u32 unused;
            w = lz4_walk<false>(B, changed, e, my_start, my_end, C, w, unused, stt);       // e >= my_start: a predecessor's exit
            my_entry = changed ? e : my_entry;
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
                const Lz4Quick t = lz4_quick(B.S, B.cbase, p, lim);
                if (!t.slow) {                                                               // straight-line common case
                    q.lit = ip + t.lit_pos; q.ll = t.lit; q.ml = t.ml; q.off = t.off;
                } else {
                    const Lz4Tok tt = lz4_token_at(B, p, C, true);
                    q.lit = ip + tt.lit_pos; q.ll = tt.lit; q.ml = tt.ml; q.off = tt.off;
                    if (tt.flags & 1) q.bad = (tt.flags & 4) ? 2u : 1u;
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
#ifdef LZ4W_ABL_NOEXEC      // developer ablation (instruction counters only; the output is wrong): parse without the executor
            (void)dead; const int rc = D_OK; { const u32 xx = wave_scan_add(q.ll + q.ml); op += (u32)__builtin_amdgcn_readlane((int)xx, 63); }
#else
            int rc;
            if constexpr (EMIT) { (void)dead; rc = emit(q, cnt, (u32)(q.lit - ip)); }
            else rc = seq_exec_batch<true, COOP>(q, cnt, op, oend, dst_lo, -1, lane, stt, B.S, to_lds_rw(sh.stage), dead);
#endif
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
// An entry may hold SEVERAL frames back to back: the reference calls LZ4F_decompress in a loop `while (avail_out > 0 && avail_in > 0)`
// (lib/zpack_read.c:414-439) — a frame or skippable frame that completes returns 0 and the loop goes on with what is left of the
// input; it stops when the input or the output space is used up, and the LAST return value decides (0 = a frame boundary: OK;
// otherwise FILE_INCOMPLETE / BUFFER_TOO_SMALL).  LZ4F looks at no header before it holds 7 bytes (fewer = "need more input"), then
// checks the magic and FLG, waits for the header in full, then checks BD and the header checksum — in that order
// (tests/golden/foreign_frames.json "lz4f:*" round-4 cases hold the reference's verdicts).
template <int COOP = 2>
__device__ inline DecodeOut lz4f_decode_wave(Lz4WaveShared& sh, Watchdog& wd, SeqStats& stt, const u8* src, u64 src_size, const u8* src_lo, const u8* src_hi,
                                             u8* dst, u64 dst_cap, int lane, Lz4Resume* rs = nullptr)
{
    DecodeOut r; r.rc = D_OK; r.produced = 0;
    const u8* ip = src;
    const u8* const iend = src + src_size;
    u8* op = dst;
    u8* const oend = dst + dst_cap;
    bool resume_in_blocks = false, first = true;
    if (rs && (rs->in_blocks || rs->nframes)) {      // streaming: pick up at the frame the last call stopped in (or in front of)
        ip = src + rs->frame_ip_off; op = dst + rs->frame_op_off;
        resume_in_blocks = rs->in_blocks != 0;
        first = rs->nframes == 0;
    }
    for (;; first = false) {
        if (wd.expired()) { r.rc = D_MALFORMED; return r; }
        if (!first && (ip == iend || op == oend)) break;               // the reference's loop condition, between frames
        if (rs) { rs->frame_ip_off = (u64)(ip - src); rs->frame_op_off = (u64)(op - dst); if (!resume_in_blocks) rs->in_blocks = 0; }
        if (iend - ip < 7) { r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r; }
        const u32 magic = uld32(ip);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (iend - ip < 8) { r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r; }
            const u64 sz = uld32(ip + 4);
            if ((u64)(iend - ip) - 8 < sz) { r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r; }
            ip += 8 + sz;
            if (rs) rs->nframes++;
            continue;
        }
        if (magic != 0x184D2204u) { r.rc = D_MALFORMED; return r; }
        const u32 flg = uld8(ip + 4), bd = uld8(ip + 5);
        const bool indep = (flg >> 5) & 1, bck = (flg >> 4) & 1, has_cs = (flg >> 3) & 1, cck = (flg >> 2) & 1, has_dict = flg & 1;
        if ((flg >> 6) != 1 || (flg & 2)) { r.rc = D_MALFORMED; return r; }
        const u32 hdr = 7 + (has_cs ? 8 : 0) + (has_dict ? 4 : 0);
        if ((u64)(iend - ip) < hdr) { r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r; }
        if (bd & 0x8F) { r.rc = D_MALFORMED; return r; }
        const u32 bcode = (bd >> 4) & 7;
        if (bcode < 4) { r.rc = D_MALFORMED; return r; }
        const u64 bmax = (u64)1 << (8 + 2 * bcode);            // 4:64K 5:256K 6:1M 7:4M
        u64 content_size = 0;
        if (has_cs) content_size = uni64(ld64(ip + 6));
        {
            u32 h = 0;
            lane0_guard();
            if (lane == 0) h = (xxh32_serial(ip + 4, hdr - 5, 0) >> 8) & 0xFF;
            if (uni(h) != uld8(ip + hdr - 1)) { r.rc = D_MALFORMED; return r; }
        }
        ip += hdr;

        u8* const frame_out = op;
        if (resume_in_blocks) { ip = src + rs->ip_off; op = dst + rs->op_off; resume_in_blocks = false; }     // the block the last call stopped in front of
        for (;;) {
            if (wd.expired()) { r.rc = D_MALFORMED; return r; }
            if (rs) { rs->in_blocks = 1; rs->ip_off = (u64)(ip - src); rs->op_off = (u64)(op - dst); }     // everything before this block is done
            if (iend - ip < 4) { r.rc = D_TRUNCATED; r.produced = (u64)(op - dst); return r; }
            const u32 bh = uld32(ip);
            ip += 4;
            if (bh == 0) break;
            const bool raw = bh >> 31;
            const u64 bsz = bh & 0x7FFFFFFFu;
            if (bsz > bmax) { r.rc = D_MALFORMED; return r; }
            if ((u64)(iend - ip) < bsz + (bck ? 4u : 0u)) {
                if (raw && !rs) {      // what is there of a stored block still streams out before the input starves
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
                const int rc = lz4_block_wave<COOP>(sh, wd, stt, ip, (u32)bsz, src_hi, hist_lo, op, bend, lane);
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
        if (rs) { rs->nframes++; rs->in_blocks = 0; rs->frame_ip_off = (u64)(ip - src); rs->frame_op_off = (u64)(op - dst); }
    }
    r.produced = (u64)(op - dst);
    return r;
}

}  // namespace zpk
