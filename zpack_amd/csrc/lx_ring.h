// lx_ring.h — an LDS OUTPUT RING for the sequence executors: the last 4 KiB of an entry's output live in LDS, literals and
// matches are OR-merged into it with 8-byte ALIGNED accesses only, matches that reach < 1 KiB back never touch memory,
// whole 1 KiB lines leave with one 16-byte store per lane and are fed to the XXH3 accumulators on the way out (no re-read).
// Used by the Zstandard execute stage (zstd_ring.h).
//
// History: round 2 built this ring for an opt-in LZ4 path (lane-per-unit token scan + seam repair + ring executor in front
// of the general decoder).  Measured on the headline workload it lost to the general decoder k_lz4_wave on every corpus
// class (424 vs 620 GiB/s on the 70/20/5/5 mix, profiles/r02/r02_c2_lz4_ring_bench.json), so round 3 removed the four LZ4
// kernels and the option; the ring itself earns its keep in k_zstd_exec (-34 % bytes in, -27 % out against the direct executor).
#pragma once
#include "zpk_device.h"
#include "xxh3_device.h"
#include "seq_exec.h"

namespace zpk {

#ifndef LX_MAX_LL
#define LX_MAX_LL 1024u                      // longest literal run / match a batch takes; longer ones go on their own (lx_append_*)
#endif
#define LX_MAX_ML LX_MAX_LL
#ifndef LX_RING
#define LX_RING 4096u                        // bytes of output kept in LDS: abs positions [rb, rb + LX_RING), rb a multiple of 1 KiB
#endif
#define LX_HIST 1024u                        // a slide keeps at least this much flushed history (>= LX_MAX_ML: a source is all-ring or all-memory)



// anomaly codes (diagnostics only: whatever the code, the entry goes to the general decoder)
enum { LX_OK = 0, LX_E_FRAME = 1, LX_E_LIST = 2, LX_E_TOKEN = 3, LX_E_OFFSET = 4, LX_E_CAPACITY = 5, LX_E_BLOCKMAX = 6, LX_E_FIT = 7, LX_E_ROUNDS = 8 };

#ifdef LX_STATS
#define LXT(slot) do { const u64 t_ = __builtin_amdgcn_s_memtime(); O.tm[slot] += t_ - O.t_last; O.t_last = t_; } while (0)
#else
#define LXT(slot) do { } while (0)
#endif
// RING_ bytes of output kept in LDS; a slide keeps at least HIST_ bytes of flushed history.  The Zstandard execute stage uses
// LxOut = <LX_RING, LX_HIST>; the LZ4 window executor (lz4_two.h) a larger ring.
template <u32 RING_, u32 HIST_>
struct LxOutT {
    static constexpr u32 RING = RING_, HIST = HIST_;
    static_assert(RING_ - HIST_ - 1023u >= LX_MAX_LL + LX_MAX_ML, "one sequence always fits behind a slide");
    static_assert(HIST_ >= LX_MAX_ML, "an overlapping match's source is always in the ring");
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
typedef LxOutT<LX_RING, LX_HIST> LxOut;

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
template <class OT>
__device__ __forceinline__ u128 lx_load16(const OT& O, u32 s)
{
    u128 v;
#ifdef LX_ABL_NOGATHER
    return lds_ld16_any((lds_cp8)O.ring, (s - O.rb) & (OT::RING - 1));
#endif
    if (s >= O.rb) v = lds_ld16_any((lds_cp8)O.ring, s - O.rb);
    else v = ld128(O.dst + s);
    return v;
}

template <class OT>
__device__ __forceinline__ void lx_flush_blocks(OT& O, int lane)
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

// make the ring hold [rb', wp) with rb' = fp - OT::HIST: afterwards at least OT::RING - OT::HIST - 1023 bytes are free
template <class OT>
__device__ __forceinline__ void lx_slide(OT& O, int lane)
{
    lx_flush_blocks(O, lane);
    const u32 nrb = O.fp >= OT::HIST ? O.fp - OT::HIST : 0u;
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
template <class OT>
__device__ inline int lx_append_raw(OT& O, const u8* s, u64 n, const u8* read_hi, u64 dst_cap, int lane)
{
    if ((u64)O.wp + n > dst_cap) return LX_E_CAPACITY;
    while (n) {
        u32 m = n < 1024u ? (u32)n : 1024u;
        if (O.wp + m > O.rb + OT::RING) lx_slide(O, lane);
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
template <class OT>
__device__ inline int lx_append_fill(OT& O, u32 byte, u64 n, u64 dst_cap, int lane)
{
    if ((u64)O.wp + n > dst_cap) return LX_E_CAPACITY;
    u128 pat; pat.lo = 0x0101010101010101ull * (u64)(byte & 0xFFu); pat.hi = pat.lo;
    while (n) {
        const u32 m = n < 1024u ? (u32)n : 1024u;
        if (O.wp + m > O.rb + OT::RING) lx_slide(O, lane);
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
template <class OT>
__device__ inline int lx_append_match(OT& O, u32 off, u64 n, u32 hist_lo, u64 dst_cap, int lane)
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
        if (O.wp + m > O.rb + OT::RING) lx_slide(O, lane);
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

// Execute `cnt` sequences (lane k < cnt: literal length ll from L, then a match of ml bytes at distance off; ml = 0: none) at the
// ring's write position: output positions (prefix sum), in-batch dependencies, literals and matches OR-ed into the ring, whole
// 1 KiB blocks flushed (and hashed).  cnt may come back SMALLER: a batch whose output does not fit the ring is cut (the caller goes
// on behind the sequences taken).  Lengths must be <= LX_MAX_LL / LX_MAX_ML.  hist_lo = lowest abs output position a match may reach.
template <class OT, class LitSrc>
__device__ __forceinline__ int lx_exec_batch(OT& O, u32& cnt, u32 ll, u32 ml, u32 off, const LitSrc& L, u32 hist_lo, u64 dst_cap,
                                             int lane, SeqStats& stt)
{
    bool act = (u32)lane < cnt;
    if (!act) { ll = 0; ml = 0; }
    // ---- output positions ----
    u32 x = wave_scan_add(ll + ml);
    u32 total = (u32)__builtin_amdgcn_readlane((int)x, 63);
    if (O.wp + total > O.rb + OT::RING) {                 // (only a batch of more than 2 KiB of output gets here)
        lx_slide(O, lane);
        const u32 free_ = O.rb + OT::RING - O.wp;
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
template <class OT>
__device__ __forceinline__ void lx_begin(OT& O, lds_p8 ring, u8* dst, u64 uncomp_size, int lane, lds_p8 sec = nullptr)
{
    O.ring = ring; O.dst = dst; O.sec = (lds_cp8)sec;
    if (sec && lane < 12) lds_st128(sec + 16 * lane, ld128(XXH3_SECRET + 16 * lane));
    for (u32 c = 16u * (u32)lane; c < OT::RING + 32u; c += 1024u) { u128 z; z.lo = 0; z.hi = 0; lds_st128(O.ring + c, z); }     // ring bytes >= wp are zero, always
    wave_mem_fence();
    O.wp = 0; O.rb = 0; O.fp = 0;
    O.hash_blocks = uncomp_size > 240 ? (u32)((uncomp_size - 1) >> 10) : 0u;
    O.xs.init(lane);
}
// the tail (what is left of the last 1 KiB block, exact to the byte), then XXH3 of dst[0, uncomp_size) (lib/zpack_read.c:466): fused
// when the entry produced exactly that many bytes, by re-reading otherwise
template <class OT>
__device__ __forceinline__ void lx_finish(OT& O, u8* dst, u64 uncomp_size, LxResult& R, int lane)
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

}  // namespace zpk
