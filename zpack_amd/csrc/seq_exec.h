// seq_exec.h — execute a batch of up to 64 LZ sequences with one wave, lane-parallel.
//
// Shared by the LZ4 (lz4_wave.h) and Zstandard (zstd_wg.h) decoders: both parse their token /
// FSE streams wave-uniformly into lane registers — lane k holds sequence k = (literal source, literal
// length, match length, offset) — and hand the batch to seq_exec_batch().
//
//   1. wave prefix sum of (literal length + match length): every sequence's output position
//      (the "wavefront prefix-sum for output offsets")
//   2. literal runs are copied by their own lanes (16 B per lane per instruction, exact tails); runs
//      longer than 64 B are copied by the whole wave, 1 KiB per instruction.  Literals depend on nothing.
//   3. matches: a match may read bytes written by earlier matches OF THE SAME BATCH (everything older is
//      already in memory).  Each lane builds the 64-bit mask of batch lanes whose match output its
//      source range touches — sequences are sorted by output position, so this is one uniform sweep with
//      v_readlane, no memory — and, where its whole source lies inside ONE earlier match, it re-points
//      the source at THAT match's source (window[x] == window[x - offset] holds for every byte of a
//      match), which collapses copy-of-a-copy chains without waiting for them.
//   4. rounds: all lanes whose dependencies are done copy at once (short, non-overlapping matches by
//      their own lane; long or self-overlapping ones by the whole wave, period handled with a modulo);
//      the done mask is a ballot.  The lowest unfinished sequence is always ready, so the loop ends.
//
// One round costs about one HBM/L2 round trip for up to 64 sequences, where the serial decoder paid one
// or two round trips per sequence.
#pragma once
#include "zpk_device.h"

namespace zpk {

// copy exactly n (0..16) bytes, global -> global, any alignment, never touching byte n or beyond
__device__ __forceinline__ void gcopy_upto16(u8* d, const u8* s, u32 n)
{
    if (n >= 16) { st128(d, ld128(s)); return; }
    if (n & 8) { st64(d, ld64(s)); d += 8; s += 8; }
    if (n & 4) { st32(d, ld32(s)); d += 4; s += 4; }
    if (n & 2) { st16(d, ld16(s)); d += 2; s += 2; }
    if (n & 1) st8(d, ld8(s));
}

// Two-halves form (loads first, stores later) for 0..32 bytes with the small-memcpy trick: any length is
// covered by at most TWO accesses of one width, the second overlapping the first ([0,w) and [n-w,n)).
// Scattered per-lane accesses are bound by the texture-addresser / L1 tag rate (~1 lane-address per cycle),
// so halving the number of vector-memory instructions per copy is worth more than anything else here.
struct Copy32 { u128 lo, hi; };           // lo = bytes [0,w) (zero-extended), hi = bytes [n-w, n)

__device__ __forceinline__ Copy32 gload_upto32(const u8* s, u32 n)
{
    Copy32 c; c.lo.lo = c.lo.hi = c.hi.lo = c.hi.hi = 0;
    if (n >= 16)      { c.lo = ld128(s); c.hi = ld128(s + n - 16); }
    else if (n >= 8)  { c.lo.lo = ld64(s); c.hi.lo = ld64(s + n - 8); }
    else if (n >= 4)  { c.lo.lo = ld32(s); c.hi.lo = ld32(s + n - 4); }
    else if (n >= 2)  { c.lo.lo = ld16(s); c.hi.lo = ld16(s + n - 2); }
    else if (n == 1)  { c.lo.lo = ld8(s); }
    return c;
}
__device__ __forceinline__ void gstore_upto32(u8* d, const Copy32& c, u32 n)
{
    if (n >= 16)      { st128(d, c.lo); st128(d + n - 16, c.hi); }
    else if (n >= 8)  { st64(d, c.lo.lo); st64(d + n - 8, c.hi.lo); }
    else if (n >= 4)  { st32(d, (u32)c.lo.lo); st32(d + n - 4, (u32)c.hi.lo); }
    else if (n >= 2)  { st16(d, (u16)c.lo.lo); st16(d + n - 2, (u16)c.hi.lo); }
    else if (n == 1)  { st8(d, (u8)c.lo.lo); }
}

// Wide form: ONE 16-byte load covers every length up to 16 when the 16 bytes are readable (`wide`: the
// caller knows the buffer extends that far; what lies past n is never stored), two overlapping ones cover
// 17..32.  The store side then needs bytes [0,n) of c.lo exactly: gstore_wide32 splits n < 16 by its bits
// (at most 4 stores per wave instead of 7 for the two-overlap form, and no second load).
__device__ __forceinline__ Copy32 gload_wide32(const u8* s, u32 n, bool wide)
{
    Copy32 c; c.lo.lo = c.lo.hi = c.hi.lo = c.hi.hi = 0;
    if (n > 16)      { c.lo = ld128(s); c.hi = ld128(s + n - 16); }
    else if (n == 0) { }
    else if (wide)   { c.lo = ld128(s); }
    else {
        u64 lo = 0, t = 0; u32 sh = 0; const u8* p = s;
        if (n & 8) { lo = ld64(p); p += 8; }
        if (n & 4) { t = (u64)ld32(p); sh = 32; p += 4; }
        if (n & 2) { t |= (u64)ld16(p) << sh; sh += 16; p += 2; }
        if (n & 1) { t |= (u64)ld8(p) << sh; }
        if (n & 8) { c.lo.lo = lo; c.lo.hi = t; } else c.lo.lo = t;
        if (n == 16) c.lo = ld128(s);
    }
    return c;
}
__device__ __forceinline__ void gstore_wide32(u8* d, const Copy32& c, u32 n)
{
    if (n >= 16) { st128(d, c.lo); if (n > 16) st128(d + n - 16, c.hi); return; }
    u64 t = c.lo.lo;
    if (n & 8) { st64(d, t); d += 8; t = c.lo.hi; }
    if (n & 4) { st32(d, (u32)t); d += 4; t >>= 32; }
    if (n & 2) { st16(d, (u16)t); d += 2; t >>= 16; }
    if (n & 1) st8(d, (u8)t);
}

// the same split into an LDS buffer (batch assembly, see seq_exec_batch)
__device__ __forceinline__ void lds_store_wide32(lds_p8 d, const Copy32& c, u32 n)
{
    if (n >= 16) { lds_st128(d, c.lo); if (n > 16) lds_st128(d + n - 16, c.hi); return; }
    u64 t = c.lo.lo;
    if (n & 8) { lds_st64(d, t); d += 8; t = c.lo.hi; }
    if (n & 4) { lds_st32(d, (u32)t); d += 4; t >>= 32; }
    if (n & 2) { lds_st16(d, (u16)t); d += 2; t >>= 16; }
    if (n & 1) lds_st8(d, (u8)t);
}

// legacy exact-tail helpers (cooperative paths)
__device__ __forceinline__ u128 gload_upto16(const u8* s, u32 n)
{
    if (n >= 16) return ld128(s);
    u64 lo = 0, t = 0;
    u32 pos = 0, sh = 0;
    if (n & 8) { lo = ld64(s); pos = 8; }
    const u8* p = s + pos;
    if (n & 4) { t = (u64)ld32(p); sh = 32; p += 4; }
    if (n & 2) { t |= (u64)ld16(p) << sh; sh += 16; p += 2; }
    if (n & 1) { t |= (u64)ld8(p) << sh; }
    u128 v;
    if (pos) { v.lo = lo; v.hi = t; } else { v.lo = t; v.hi = 0; }
    return v;
}
__device__ __forceinline__ void gstore_upto16(u8* d, u128 v, u32 n)
{
    if (n >= 16) { st128(d, v); return; }
    u64 t = v.lo;
    if (n & 8) { st64(d, v.lo); d += 8; t = v.hi; }
    if (n & 4) { st32(d, (u32)t); d += 4; t >>= 32; }
    if (n & 2) { st16(d, (u16)t); d += 2; t >>= 16; }
    if (n & 1) st8(d, (u8)t);
}
// up to 64 bytes: every load is issued before the first store (one memory round trip)
__device__ __forceinline__ void gcopy_upto64(u8* d, const u8* s, u32 n)
{
    u128 v0 = {0, 0}, v1 = {0, 0}, v2 = {0, 0}, v3 = {0, 0};
    if (n > 0) v0 = gload_upto16(s, n);
    if (n > 16) v1 = gload_upto16(s + 16, n - 16);
    if (n > 32) v2 = gload_upto16(s + 32, n - 32);
    if (n > 48) v3 = gload_upto16(s + 48, n - 48);
    if (n > 0) gstore_upto16(d, v0, n);
    if (n > 16) gstore_upto16(d + 16, v1, n - 16);
    if (n > 32) gstore_upto16(d + 32, v2, n - 32);
    if (n > 48) gstore_upto16(d + 48, v3, n - 48);
}

__device__ __forceinline__ void gcopy_upto32(u8* d, const u8* s, u32 n)
{
    const Copy32 c = gload_upto32(s, n);
    gstore_upto32(d, c, n);
}

__device__ __forceinline__ const u8* readlane_ptr(const u8* p, int k)
{
    return (const u8*)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)p >> 32), k) << 32) |
                       (u32)__builtin_amdgcn_readlane((int)(u32)(u64)p, k));
}

__device__ __forceinline__ u8* shfl_ptr(const u8* p, int k)
{
    return (u8*)(((u64)(u32)__shfl((int)(u32)((u64)p >> 32), k, 64) << 32) | (u32)__shfl((int)(u32)(u64)p, k, 64));
}

// Cooperative copies FOUR AT A TIME (round 5): every 16-lane row of the wave takes one run — lane k in `mask` holds a run of n bytes
// to d from s (both global; no run's source overlaps any run's target of the same call), or, with `pat`, a PERIODIC run: its first
// `period` bytes (1, 2, 4 or 8: a divisor of 16, so every 16-byte piece of the run is the same vector) are the low bytes of `pv` and
// nothing is loaded at all.  A run used to be copied by the whole wave, one run after the other, each behind its own memory round
// trip: an entry of byte runs (LZ4: literal + match at distance 1, ~400 sequences of ~165 bytes per 64 KiB) spent 0.8 ms in ~360
// dependent round trips — as long as a text entry with 12 x the sequences (profiles/r04/r04_c2_per_class.txt: 614 vs 589 GiB/s).
__device__ __forceinline__ void coop_copy_rows(u64 mask, u32 n, u8* d, const u8* s, bool pat, u32 period, u64 pv, int lane)
{
    const int g = lane >> 4;
    const u32 sub16 = 16u * (u32)(lane & 15);
    while (mask) {
        int k[4];
        #pragma unroll
        for (int i = 0; i < 4; i++) { k[i] = mask ? __ffsll((long long)mask) - 1 : -1; if (mask) mask &= mask - 1; }
        const int kk = g == 0 ? k[0] : (g == 1 ? k[1] : (g == 2 ? k[2] : k[3]));
        const int from = kk < 0 ? lane : kk;
        const u32 rn_any = (u32)__shfl((int)n, from, 64);          // (cross-lane reads stay outside divergent control flow: an inactive source lane reads as garbage)
        const u32 rn = kk < 0 ? 0u : rn_any;
        u8* const rd = shfl_ptr(d, from);
        const u8* const rs = shfl_ptr(s, from);
        const bool rpat = __shfl((int)(pat ? 1 : 0), from, 64) != 0;
        const u32 rper = (u32)__shfl((int)period, from, 64);
        u64 x = ((u64)(u32)__shfl((int)(u32)(pv >> 32), from, 64) << 32) | (u32)__shfl((int)(u32)pv, from, 64);
        if (rper < 8u) { x &= 0xFFFFFFFFull; x |= x << 32; }
        if (rper < 4u) { x &= 0x0000FFFF0000FFFFull; x |= x << 16; }
        if (rper < 2u) { x &= 0x00FF00FF00FF00FFull; x |= x << 8; }
        for (u32 c = sub16; __ballot(c < rn) != 0; c += 256u) {
            if (c < rn) {
                u128 v; v.lo = x; v.hi = x;
                if (!rpat) v = gload_upto16(rs + c, rn - c);
                gstore_upto16(rd + c, v, rn - c);
            }
        }
    }
}

// The cooperative matches of one round (lanes in `cm`: ml > SEQ_OWN_MAX or offset < ml; all their sources are final).  Everything
// here sits behind ONE uniform test of the round loop (`if (cm)`): on text a batch in eight has such a match, and a first version
// that mixed the period loads into the round's common loads cost the common case 9 % (round 5, profiles/r05); as a real function
// call (__noinline__) it cost 31 % — the call's register save area turned the kernel's 12 bytes of scratch into 96.
//   * offset >= ml (a long plain copy) and offset < ml with a period of 1, 2, 4 or 8 bytes (byte runs, 16-bit patterns: the period
//     is read by the match's own lane, one round trip for all of them, and the run is written from registers): FOUR AT A TIME,
//     coop_copy_rows;
//   * other periods: one match at a time by the whole wave — period >= 16 slab by slab (no lane reads what the same instruction
//     writes), shorter periods byte by byte with a modulo.
__device__ __forceinline__ void seq_coop_round(u64 cm, u32 ml, u32 off, u8* ms, const u8* srcp, const u8* oend, int lane)
{
    const bool mine = (cm >> lane) & 1;
    const bool self_overlap = ml > off;
    const bool pat = mine && self_overlap && off <= 8u && (off & (off - 1u)) == 0u;
    Copy32 cb; cb.lo.lo = cb.lo.hi = cb.hi.lo = cb.hi.hi = 0;
    if (pat) cb = gload_wide32(srcp, off, srcp + 16 <= oend);
    const u64 gm = __ballot(mine && (pat || !self_overlap));
    if (gm) coop_copy_rows(gm, ml, ms, srcp, pat, off, cb.lo.lo, lane);
    u64 rest = cm & ~gm;
    while (rest) {
        const int k = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        const u32 n = (u32)__builtin_amdgcn_readlane((int)ml, k);
        const u32 koff = (u32)__builtin_amdgcn_readlane((int)off, k);
        u8* p = (u8*)readlane_ptr(ms, k);
        const u8* m = readlane_ptr(srcp, k);                       // first period of match k
        if (koff >= 16) {
            for (u32 c = (u32)lane * 16; c < koff; c += WAVE * 16) gcopy_upto16(p + c, m + c, koff - c);
            wave_mem_fence();
            for (u32 base = koff; base < n; base += koff) {
                const u32 slab = n - base < koff ? n - base : koff;
                for (u32 c = (u32)lane * 16; c < slab; c += WAVE * 16) gcopy_upto16(p + base + c, p + base - koff + c, slab - c);
                wave_mem_fence();
            }
        } else {
            for (u32 c = lane; c < n; c += WAVE) st8(p + c, ld8(m + c % koff));
        }
    }
}

// developer aid: cycle accounting of the executor (kept in registers; written out only when asked)
// compiled in only with -DZPK_STATS (the counters cost ~10 registers, i.e. a wave of occupancy per SIMD)
#ifdef ZPK_STATS
struct SeqStats { u64 t_parse, t_lit, t_dep, t_rounds; u32 rounds, batches, coops, redirects; u64 t_stage, t_walk1, t_fix, t_emit, t_tok; u32 fix_iters, chunks, asm_batches, hops_first, hops_fix, slow_hops; };
#define SEQ_T() __builtin_amdgcn_s_memtime()
#define SEQ_STAT(x) do { x; } while (0)
#else
struct SeqStats { };
#define SEQ_T() 0ull
#define SEQ_STAT(x) do { } while (0)
#endif

// a lane copies its own literal run / match only up to this many bytes (register budget: 2 x 16 B per kind);
// longer ones are copied by the whole wave
#define SEQ_OWN_MAX 32u
// pointer-jumping rounds of the dependency analysis (chain depth halves per round)
#ifndef SEQ_DEP_ROUNDS
#define SEQ_DEP_ROUNDS 4
#endif

#define SEQ_NO_LDS 0xFFFFFFFFu
// Batch assembly buffer (optional, LDS, per wave): SEQ_ASM_PRE bytes of history, the batch output, 16 bytes of
// read slack.
#define SEQ_ASM_PRE 32u
#define SEQ_ASM_SLACK 16u
struct SeqBatch {
    const u8* lit;     // literal source of this lane's sequence (ignored when lit_rle)
    u32 lit_lds;       // byte offset of the same literals in the caller's LDS staging buffer, or SEQ_NO_LDS
    u32 ll, ml;        // literal length, match length (0 = no match)
    u32 off;           // match offset (>= 1 when ml != 0)
    u32 bad;           // 0; 1 = the sequence is malformed before its literal run, 2 = after it (the reference copies the literals first)
};

// In-batch dependency analysis: pure cross-lane arithmetic (ds_bpermute + ALU, no memory).
// In: this lane's match output [r_ms, r_me) and offset (positions relative to the batch start), `pending` =
// ballot of lanes with a match.  Out: src = start of the (possibly re-pointed) source of the first need_len
// bytes, relative to the batch start (negative = older data); returns a mask of earlier batch lanes whose
// match output must be final before that source may be read.
//
// One search gives, per lane, the exact set of earlier matches its source touches.  A source that sits
// entirely inside the externally-sourced part of ONE earlier match k ("inside" = k) can be read from k's own
// source instead: window[x] == window[x - delta_k] holds there.  Such lanes then form a forest, and pointer
// jumping over it needs no further search: after a jump, lane j's source is a sub-range of k's source, so
// k's dependency set is a valid (superset) set for j, and if k's source sits inside m's, so does j's.
// (delta, need, inside) of k are read before k itself jumps in the same round, which keeps the triple
// consistent; the depth of every chain halves per round.
// S = i32 when every position and offset of the batch fits 31 bits (always, for LZ4): one-instruction compares and
// single-dword shuffles instead of 64-bit pairs; S = i64 otherwise.
template <typename S>
__device__ __forceinline__ u64 seq_dependencies(bool has_match, u32 r_ms, u32 r_me, u32 off, u32 need_len, u64 pending, int lane,
                                                i64& src_out, SeqStats& stt)
{
    (void)stt;
    S src = (S)r_ms - (S)off;
    u64 need = 0;
    const S send0 = src + (S)need_len;
    const bool reads_batch = has_match && send0 > 0;
    if (pending && __ballot(reads_batch) != 0) {
        // which earlier matches of the batch does [src, send) touch?  Output ranges are sorted by lane, so two
        // binary searches over the wave give the lane interval.
        int inside = -1;                                       // lane whose match holds the whole source
        int klo = 0, khi = 0;                                  // klo = first lane with r_me > src; khi = first lane with r_ms >= send
        #pragma unroll
        for (int step = 32; step >= 1; step >>= 1) {
            const u32 a = (u32)__shfl((int)r_me, klo + step - 1, 64);
            const u32 b = (u32)__shfl((int)r_ms, khi + step - 1, 64);
            if ((S)a <= src) klo += step;
            if ((S)b < send0) khi += step;
        }
        if (reads_batch && klo < khi) {
            if (khi > lane) khi = lane;
            const u64 span = (khi >= 64 ? ~0ull : ((1ull << khi) - 1)) & ~((1ull << klo) - 1);
            need = span & pending;                             // only lanes that actually have a match
            if (need && (need & (need - 1)) == 0) inside = __ffsll((long long)need) - 1;   // exactly one candidate
        }
        {   // containment in the candidate's externally-sourced part; cross-lane reads stay outside divergent control flow
            const int probe = inside < 0 ? lane : inside;
            const u32 kms = (u32)__shfl((int)r_ms, probe, 64);
            const u32 knl = (u32)__shfl((int)need_len, probe, 64);
            if (inside >= 0 && !((S)kms <= src && send0 <= (S)kms + (S)knl)) inside = -1;
        }
        #pragma unroll 1
        for (int round = 0; round < SEQ_DEP_ROUNDS; round++) {
            if (__ballot(inside >= 0) == 0) break;
            SEQ_STAT(stt.redirects += (u32)__popcll(__ballot(inside >= 0)));
            const int probe = inside < 0 ? lane : inside;
            const S my_delta = (S)r_ms - src;
            const u32 dlo = (u32)__shfl((int)(u32)(u64)(i64)my_delta, probe, 64);
            const u32 dhi = sizeof(S) == 8 ? (u32)__shfl((int)(u32)((u64)(i64)my_delta >> 32), probe, 64) : 0u;
            const u32 nlo = (u32)__shfl((int)(u32)need, probe, 64), nhi = (u32)__shfl((int)(u32)(need >> 32), probe, 64);
            const int kin = __shfl(inside, probe, 64);
            if (inside >= 0) { src -= sizeof(S) == 8 ? (S)(i64)(((u64)dhi << 32) | dlo) : (S)(i32)dlo; need = ((u64)nhi << 32) | nlo; inside = kin; }
        }
    }
    src_out = (i64)src;
    return need;
}

// Execute `cnt` sequences (lane k < cnt holds sequence k).  op = output cursor (uniform, advanced),
// oend = end of the output slot, dst_lo = lowest address a match may read.  lit_rle >= 0: every literal
// byte equals that value (Zstandard RLE literals).  Returns D_OK / D_MALFORMED / D_DST_FULL.
// lit_stage: LDS buffer q.lit_lds indexes (16 readable bytes past every literal run it is used for), or null.
// asm_buf/asm_cap: per-wave LDS scratch of asm_cap bytes (0 = none) that does not overlap this batch's
// literals.  A batch whose output fits (SEQ_ASM_PRE + total + SEQ_ASM_SLACK <= asm_cap) and that has no long or
// self-overlapping piece is ASSEMBLED IN LDS: literals and matches are written there, in-batch sources are
// read back from there at LDS latency instead of a store->load round trip through L2, and the finished batch
// goes to memory as one contiguous 16-bytes-per-lane stream — one or two full store instructions instead of
// ~17 partially filled ones.
// NARROW_ONLY: the caller guarantees 31-bit positions and offsets (LZ4: 16-bit offsets, lengths clamped to 2^23), so only
// the 32-bit dependency analysis is instantiated.
// COOP (round 5): what a batch does with its cooperative pieces (matches longer than SEQ_OWN_MAX or feeding themselves, literal runs
// longer than SEQ_OWN_MAX).  2 = four at a time, periodic runs from registers (coop_copy_rows, seq_coop_round): entries of byte runs
// decode ~10 x faster.  0 = one at a time by the whole wave: the round-4 code, kept for k_lz4_wave ONLY — the mere presence of the
// grouped code in that kernel, behind one uniform branch that text never takes, cost text 6 % (the compiler's code for the common
// path changes: +3.7 % vector instructions, +18 % wait cycles; instruction-cache misses are nil), as a real call (__noinline__) 31 %:
// profiles/r05/r05_coop_variants_ab.txt.  So k_classify sends the LZ4 entries that are mostly runs (compressed to less than an eighth)
// to k_lz4_left, built with COOP = 2, and k_lz4_wave is the code it was.
template <bool NARROW_ONLY = false, int COOP = 2>
__device__ __forceinline__ int seq_exec_batch(const SeqBatch& q, int cnt, u8*& op, u8* oend, const u8* dst_lo, int lit_rle, int lane,
                                              SeqStats& stt, lds_cp8 lit_stage = nullptr, lds_p8 asm_buf = nullptr, u32 asm_cap = 0)
{
    u64 t0 = SEQ_T(); (void)t0; (void)stt;
    SEQ_STAT(stt.batches++);
    const bool act = lane < cnt;
    const u32 ll = act ? q.ll : 0u, ml = act ? q.ml : 0u;
    // ---- 1. output positions ----
    const u32 x = wave_scan_add(ll + ml);
    const u64 total = (u32)__builtin_amdgcn_readlane((int)x, 63);
    u8* const o = op + (x - (ll + ml));          // literal start of this lane's sequence
    u8* const ms = o + ll;                        // match start
    const bool has_match = act && ml != 0;
    {   // The verdict is that of the FIRST offending sequence in stream order, with the checks of one sequence in the order
        // of a serial decoder (malformed token / literals do not fit / bad offset or truncated match / match does not fit):
        // a batch is 64 sequences examined at once, but `output too small` and `malformed` are different results upstream.
        const u64 cap = (u64)(oend - op);
        const bool off_bad = has_match && (q.off == 0 || (u64)q.off > (u64)(ms - dst_lo));
        u32 v = 0;                                // 1 malformed, 2 does not fit
        if (act) {
            if (q.bad == 1) v = 1;
            else if ((u64)(x - ml) > cap) v = 2;
            else if (q.bad == 2 || off_bad) v = 1;
            else if ((u64)x > cap) v = 2;
        }
        const u64 vm = __ballot(v != 0);
        if (vm != 0) return (u32)__builtin_amdgcn_readlane((int)v, __ffsll((long long)vm) - 1) == 1u ? D_MALFORMED : D_DST_FULL;
    }
    u64 pending = __ballot(has_match);

    // ---- 2. in-batch dependencies (positions relative to op) ----
    const u32 r_ms = (u32)(ms - op), r_me = r_ms + ml;           // this lane's match output [r_ms, r_me)
    const u32 need_len = ml < q.off ? ml : q.off;                 // bytes not produced by the match itself
    i64 src;
    u64 need;
    if (NARROW_ONLY) need = seq_dependencies<i32>(has_match, r_ms, r_me, q.off, need_len, pending, lane, src, stt);
    else {
        const bool narrow = total < (1u << 30) && __ballot(has_match && q.off >= (1u << 30)) == 0;     // uniform
        need = narrow ? seq_dependencies<i32>(has_match, r_ms, r_me, q.off, need_len, pending, lane, src, stt)
                      : seq_dependencies<i64>(has_match, r_ms, r_me, q.off, need_len, pending, lane, src, stt);
    }
    const u8* const srcp = op + src;                               // (possibly re-pointed) source of the first need_len bytes
    const bool self_overlap = ml > need_len;                       // offset < length: the match feeds itself
    const bool coop = has_match && (ml > SEQ_OWN_MAX || self_overlap);
    // matches whose whole source is older than this batch go out together with the literals
    const bool early = has_match && !coop && src + (i64)need_len <= 0;
    SEQ_STAT({ u64 t1 = SEQ_T(); stt.t_dep += t1 - t0; t0 = t1; });
#ifdef SEQ_ABL_NOCOPY       // developer ablation (instruction counters only; the output is wrong): positions + dependencies, no copies
    if (__ballot(early && srcp == nullptr) == 0) { op += total; return D_OK; }
#endif

    // ---- 3a. assembly in LDS ----
#ifdef SEQ_ABL_NOASM         // developer ablation: every batch takes the direct path
    asm_cap = 0;
#endif
    if (lit_rle < 0 && total + (SEQ_ASM_PRE + SEQ_ASM_SLACK) <= (u64)asm_cap) {
        const bool straddle = has_match && !early && src < 0;                       // source begins before the batch, ends inside
        const u64 sm = __ballot(straddle);
        if (__ballot(coop || ll > SEQ_OWN_MAX) == 0 && (sm == 0 || (u64)(op - dst_lo) >= SEQ_ASM_PRE)) {
            SEQ_STAT(stt.asm_batches++);
            const lds_p8 ob = asm_buf + SEQ_ASM_PRE;                                // batch position 0
            if (sm != 0 && lane < 2) lds_st128(asm_buf + 16 * lane, ld128(op - SEQ_ASM_PRE + 16 * lane));
            Copy32 ca;
            if (q.lit_lds != SEQ_NO_LDS) {
                ca.lo.lo = ca.lo.hi = ca.hi.lo = ca.hi.hi = 0;
                if (ll) { ca.lo = lds_ld128(lit_stage + q.lit_lds); if (ll > 16) ca.hi = lds_ld128(lit_stage + q.lit_lds + (ll - 16)); }
            } else ca = gload_wide32(q.lit, ll, false);
            const u32 mn = early ? ml : 0u;
            const Copy32 cb = gload_wide32(srcp, mn, srcp + 16 <= oend);
            lds_store_wide32(ob + (r_ms - ll), ca, ll);
            lds_store_wide32(ob + r_ms, cb, mn);
            wave_mem_fence();
            u64 done = ~pending | __ballot(early);
            pending &= ~done;
            u32 guard = 0;
            while (pending) {
                const bool ready = has_match && ((pending >> lane) & 1) && (need & ~done) == 0;
                const u64 rmask = __ballot(ready);
                if (rmask == 0 || ++guard > 70) return D_MALFORMED;
                SEQ_STAT(stt.rounds++);
                if (ready) {
                    const lds_cp8 sp = (lds_cp8)(ob + (i32)src);                     // src >= -SEQ_ASM_PRE here
                    Copy32 c; c.hi.lo = c.hi.hi = 0;
                    c.lo = lds_ld128(sp);
                    if (ml > 16) c.hi = lds_ld128(sp + (ml - 16));
                    lds_store_wide32(ob + r_ms, c, ml);
                }
                wave_mem_fence();
                done |= rmask;
                pending &= ~rmask;
            }
            // flush: contiguous, 16 bytes per lane, exact end (bytes past the batch are never touched)
            const u32 tot = (u32)total;
            for (u32 c = (u32)lane * 16; c < tot; c += WAVE * 16) {
                const u128 v = lds_ld128((lds_cp8)(ob + c));
                if (c + 16 <= tot) st128(op + c, v);
                else gstore_upto16(op + c, v, tot - c);
            }
            wave_mem_fence();
            op += total;
            return D_OK;
        }
    }

    // ---- 3. literals + early matches: every load is issued before the first store ----
    if (lit_rle >= 0) {
        if (__ballot(ll > SEQ_OWN_MAX) == 0) { for (u32 c = 0; c < ll; c++) st8(o + c, (u8)lit_rle); }
        else {
            for (int k = 0; k < cnt; k++) {
                const u32 n = (u32)__builtin_amdgcn_readlane((int)ll, k);
                u8* p = (u8*)readlane_ptr(o, k);
                for (u32 c = lane; c < n; c += WAVE) st8(p + c, (u8)lit_rle);
            }
        }
        if (early) { const Copy32 cb = gload_wide32(srcp, ml, srcp + 16 <= oend); gstore_wide32(ms, cb, ml); }
    } else {
        const bool long_lit = ll > SEQ_OWN_MAX;
        const u32 ln = long_lit ? 0u : ll, mn = early ? ml : 0u;
        // literals straight from the staged input when the caller has it in LDS (no vector-memory load at all);
        // match sources may be over-read up to the end of this entry's own output slot
        Copy32 ca;
        const bool from_lds = q.lit_lds != SEQ_NO_LDS;
        if (from_lds) {
            ca.lo.lo = ca.lo.hi = ca.hi.lo = ca.hi.hi = 0;
            if (ln) { ca.lo = lds_ld128(lit_stage + q.lit_lds); if (ln > 16) ca.hi = lds_ld128(lit_stage + q.lit_lds + (ln - 16)); }
        } else ca = gload_wide32(q.lit, ln, false);
        const Copy32 cb = gload_wide32(srcp, mn, srcp + 16 <= oend);
        gstore_wide32(o, ca, ln);
        gstore_wide32(ms, cb, mn);
        if constexpr (COOP >= 2) {
            const u64 lm = __ballot(long_lit);     // long runs: four at a time, a 16-lane row each (coop_copy_rows)
            if (lm) coop_copy_rows(lm, ll, o, q.lit, false, 16u, 0ull, lane);
        } else {
        u64 lm = __ballot(long_lit);
        while (lm) {                              // long runs: whole wave, 16 B per lane
            const int k = __ffsll((long long)lm) - 1;
            lm &= lm - 1;
            const u32 n = (u32)__builtin_amdgcn_readlane((int)ll, k);
            u8* p = (u8*)readlane_ptr(o, k);
            const u8* s = readlane_ptr(q.lit, k);
            for (u32 c = (u32)lane * 16; c < n; c += WAVE * 16) gcopy_upto16(p + c, s + c, n - c);
        }
        }
    }
    wave_mem_fence();
    u64 done = ~pending | __ballot(early);                         // lanes without a match count as done
    pending &= ~done;
    SEQ_STAT({ u64 t1 = SEQ_T(); stt.t_lit += t1 - t0; t0 = t1; });

    // ---- 4. rounds ----
    u32 guard = 0;
    while (pending) {
        const bool ready = has_match && ((pending >> lane) & 1) && (need & ~done) == 0;
        const u64 rmask = __ballot(ready);
        if (rmask == 0 || ++guard > 70) return D_MALFORMED;        // cannot happen: the lowest pending lane is always ready
        SEQ_STAT(stt.rounds++);
        if (ready && !coop) {                                      // whole source final and not produced by this match
            const Copy32 cb = gload_wide32(srcp, ml, srcp + 16 <= oend);
            gstore_wide32(ms, cb, ml);
        }
        if constexpr (COOP >= 1) {
            const u64 cm = __ballot(ready && coop);                // long or self-overlapping matches (seq_coop_round)
            SEQ_STAT(stt.coops += (u32)__popcll(cm));
            if (cm) seq_coop_round(cm, ml, q.off, ms, srcp, oend, lane);
        } else {
        u64 cm = __ballot(ready && coop);
        SEQ_STAT(stt.coops += (u32)__popcll(cm));
        while (cm) {
            const int k = __ffsll((long long)cm) - 1;
            cm &= cm - 1;
            const u32 n = (u32)__builtin_amdgcn_readlane((int)ml, k);
            const u32 koff = (u32)__builtin_amdgcn_readlane((int)q.off, k);
            u8* p = (u8*)readlane_ptr(ms, k);
            const u8* m = readlane_ptr(srcp, k);                   // first period (or whole source) of match k
            if (koff >= n) { for (u32 c = (u32)lane * 16; c < n; c += WAVE * 16) gcopy_upto16(p + c, m + c, n - c); }
            else if (koff >= 16) {
                for (u32 c = (u32)lane * 16; c < koff; c += WAVE * 16) gcopy_upto16(p + c, m + c, koff - c);
                wave_mem_fence();
                for (u32 base = koff; base < n; base += koff) {
                    const u32 slab = n - base < koff ? n - base : koff;
                    for (u32 c = (u32)lane * 16; c < slab; c += WAVE * 16) gcopy_upto16(p + base + c, p + base - koff + c, slab - c);
                    wave_mem_fence();
                }
            } else {
                for (u32 c = lane; c < n; c += WAVE) st8(p + c, ld8(m + c % koff));
            }
        }
        }
        wave_mem_fence();
        done |= rmask;
        pending &= ~rmask;
    }
    SEQ_STAT({ u64 t1 = SEQ_T(); stt.t_rounds += t1 - t0; });
    op += total;
    return D_OK;
}

}  // namespace zpk
