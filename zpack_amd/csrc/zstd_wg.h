// zstd_wg.h — Zstandard frame decode (RFC 8878) for gfx950: one wave64 workgroup per entry.
//
// Replaces ZSTD_decompressDCtx as the reference calls it (lib/zpack_read.c:380): every concatenated
// frame is decoded, skippable frames are skipped, no dictionary.
//
// Data layout per workgroup:
//   LDS   FSE decode tables (LL/ML 512 x 8 B, OF 256 x 8 B), the three predefined tables, the Huffman
//         table (up to 2^12 x 2 B; persists across blocks for treeless literals), small build scratch
//   HBM   the entry's output slot doubles as the match window (offsets reach 256 KiB - 2 MiB, larger
//         than LDS; recently written lines are L2 hits), a 128 KiB literal scratch per workgroup
// Per block:
//   literals   Raw: used in place.  RLE: one byte.  Huffman: table built in LDS, the 4 streams are
//              decoded by 4 lanes, each with a 128-bit register bit container refilled one load ahead
//   sequences  the backward FSE bitstream is parsed wave-uniformly out of a 256-byte register window
//              (v_readlane, no memory wait per field), 64 sequences at a time into lane registers,
//              then executed: literal run + match copy by all 64 lanes, one sequence after another
#pragma once
#include "zpk_device.h"
#include "xxh3_device.h"
#include "lz4_wave.h"      // ByteWindow, DecodeOut
#include "seq_exec.h"

namespace zpk {

#define ZSTD_WG_THREADS 64
#define ZSTD_BLOCK_MAX (128u << 10)
#define ZSTD_LIT_SCRATCH ((128u << 10) + 64)
#define ZSEQ_LWIN 1024u               // literal window of the pre-decoded executor (LDS)
#define ZSTD_GRID_MAX 3072            // 12 workgroups per CU (LDS-limited: 13 KiB each)

// One FSE decode-table cell, packed into 32 bits (LDS is what limits how many entries a CU decodes at once):
//   bits 0..9 next_base (new_state = next_base + read(nb_bits)), 10..13 nb_bits, 14..18 add_bits (extra bits
//   of the symbol's value), 19..24 the symbol.  The value baseline comes from a per-symbol table.
typedef u32 FseCell;
__device__ __forceinline__ FseCell fse_cell(u32 next_base, u32 nb, u32 add_bits, u32 sym) { return next_base | (nb << 10) | (add_bits << 14) | (sym << 19); }
__device__ __forceinline__ u32 cell_next(FseCell c) { return c & 1023u; }
__device__ __forceinline__ u32 cell_nb(FseCell c)   { return (c >> 10) & 15u; }
__device__ __forceinline__ u32 cell_add(FseCell c)  { return (c >> 14) & 31u; }
__device__ __forceinline__ u32 cell_sym(FseCell c)  { return (c >> 19) & 63u; }

// The members the FSE sequence decoder alone needs come LAST: the execute-only kernel (k_zstd_exec, entries whose
// sequences k_zstd_fse has already decoded) allocates just ZSTD_SHARED_EXEC_BYTES of this layout.
struct alignas(16) ZstdShared {
    u8  huf[4096];                               // symbol of every max_bits-bit prefix (the code length follows from huf_rank)
    u32 huf_rank[16];                            // [w] first table index of weight class w (1..max_bits), ~0 above: nbits = max_bits + 1 - w
    union {
        struct {                                 // table descriptions and construction scratch (block headers)
            FseCell wt[64];                      // FSE table of the Huffman weights (accuracy log <= 6)
            i16 ncount[3][64];
            u8  spread[3][512];
            u16 nextc[3][64];
        };
        u32 seqbuf[64 * 3];                      // one batch of decoded sequences: offset, match length, literal length
        struct { u16 sym_start[256]; u32 huf_cnt[16]; };   // Huffman table construction (after the weights are known)
    };
    u8  weights[256];
    u32 huf_max_bits;
    u32 huf_valid;
    u32 defaults_built;
    u32 pad_;
    // ---- FSE decode (full decoder); its first bytes double as the executor's assembly buffer + literal window ----
    FseCell ll[512], ml[512], of[256];
    FseCell dll[64], dml[64], dof[32];           // predefined distributions (built once per workgroup)
    u32 symtab_ll[36], symtab_ml[54];            // literal-length / match-length codes: baseline | extra bits << 24
    u32 ofbase[32];                              // offset codes: 1 << code
};
#define ZSTD_EXEC_WORK 2592u                     // execute-only kernel: 1536 B of batch assembly + the literal window
#define ZSTD_SHARED_EXEC_BYTES (__builtin_offsetof(ZstdShared, ll) + ZSTD_EXEC_WORK)

__device__ __constant__ const u32 Z_LL_BASE[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,
                                                     0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
__device__ __constant__ const u8 Z_LL_BITS[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
__device__ __constant__ const u32 Z_ML_BASE[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                                     35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };
__device__ __constant__ const u8 Z_ML_BITS[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                                    1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
__device__ __constant__ const i16 Z_LL_DEF[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
__device__ __constant__ const i16 Z_ML_DEF[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                                    -1,-1,-1,-1,-1,-1,-1 };
__device__ __constant__ const i16 Z_OF_DEF[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

enum { T_LL = 0, T_OF = 1, T_ML = 2 };

// ---- wave-uniform bit readers over the register window -----------------------------------------

// 64 stream bits starting at bit `bitpos` (may be negative) of the byte stream at `start`, out of the
// register window (three v_readlane, no memory access unless the window has to move).  Bytes outside
// [win.lo, win.hi) read as zero.
__device__ __forceinline__ u64 win_bits64(ByteWindow& win, const u8* start, i64 bitpos, int lane, bool backward)
{
    const u8* p = start + (bitpos >> 3);         // floor
    i64 d = (i64)(p - win.base);
    if (d < 0 || d > 244) {
        const u8* nb = backward ? (const u8*)(((u64)p & ~(u64)3) - 240) : p;
        win.load(nb, lane);
        d = (i64)(p - win.base);
    }
    int i = (int)(d >> 2), t = (int)(d & 3) * 8 + (int)(bitpos & 7);      // t <= 31
    u32 w0 = (u32)__builtin_amdgcn_readlane((int)win.w, i);
    u32 w1 = (u32)__builtin_amdgcn_readlane((int)win.w, i + 1);
    u32 w2 = (u32)__builtin_amdgcn_readlane((int)win.w, i + 2);
    u64 lo = ((u64)w1 << 32) | w0;
    return t ? (lo >> t) | ((u64)w2 << (64 - t)) : lo;
}

// 128 stream bits starting at bit `bitpos` (may be negative): lo = [bitpos, bitpos+64), hi = the next 64.
// Backward streams only (the window is re-based so that it extends downwards).  Five v_readlane.
__device__ __forceinline__ void win_bits128(ByteWindow& win, const u8* start, i64 bitpos, int lane, u64& lo, u64& hi)
{
    const u8* p = uni_ptr(start + (bitpos >> 3));         // floor
    i64 d = (i64)(p - win.base);
    if (d < 0 || d > 236) {
        win.load((const u8*)(((u64)p & ~(u64)3) - 232), lane);
        d = (i64)(p - win.base);
    }
    const int i = (int)uni((u32)(d >> 2)), t = (int)uni((u32)((d & 3) * 8 + (bitpos & 7)));      // t <= 31
    const u32 w0 = (u32)__builtin_amdgcn_readlane((int)win.w, i);
    const u32 w1 = (u32)__builtin_amdgcn_readlane((int)win.w, i + 1);
    const u32 w2 = (u32)__builtin_amdgcn_readlane((int)win.w, i + 2);
    const u32 w3 = (u32)__builtin_amdgcn_readlane((int)win.w, i + 3);
    const u32 w4 = (u32)__builtin_amdgcn_readlane((int)win.w, i + 4);
    const u64 a = ((u64)w1 << 32) | w0, b = ((u64)w3 << 32) | w2;
    lo = t ? (a >> t) | (b << (64 - t)) : a;
    hi = t ? (b >> t) | ((u64)w4 << (64 - t)) : b;
}

// backward bitstream (FSE payloads): bits are consumed from the top down; below bit 0 reads zero
struct RevBits {
    const u8* start;
    i64 pos;           // bits not yet consumed
    u64 c;             // container: top `avail` bits are stream bits [pos-avail, pos)
    int avail;

    __device__ __forceinline__ bool init(ByteWindow& win, const u8* p, u64 size, int lane)
    {
        if (size == 0) return false;
        u32 last = uld8(p + size - 1);
        if (last == 0) return false;
        start = p;
        pos = (i64)(size - 1) * 8 + highbit32(last);
        win.lo = p; win.hi = p + size;
        win.base = (const u8*)~(u64)0xFFFF;      // force a (re)load on first use
        avail = 0; c = 0;
        refill(win, lane);
        return true;
    }
    __device__ __forceinline__ void refill(ByteWindow& win, int lane)
    {
        c = win_bits64(win, start, pos - 64, lane, true);     // stream bits [pos-64, pos), top-aligned
        avail = 64;
    }
    // n <= 32
    __device__ __forceinline__ u32 read(ByteWindow& win, int n, int lane)
    {
        if (n > avail) refill(win, lane);
        u32 v = n ? (u32)(c >> (64 - n)) : 0u;
        c = n ? c << n : c;          // n <= 32 < 64
        avail -= n; pos -= n;
        return v;
    }
};

// forward LSB-first reader (FSE table descriptions); bits past the end read zero
struct FwdBits {
    const u8* start;
    u64 bit;
    __device__ __forceinline__ u32 read(ByteWindow& win, int n, int lane)
    {
        u64 w = win_bits64(win, start, (i64)bit, lane, false);
        u32 v = (u32)w & ((1u << n) - 1u);                      // n <= 16
        bit += (u64)n;
        return v;
    }
};

// ---- FSE tables ---------------------------------------------------------------------------------

// RFC 8878 4.1.1: normalized counts.  Uniform.  Returns bytes consumed or -1.
__device__ inline int fse_read_ncount(ByteWindow& win, const u8* src, u64 size, int max_sym, int max_al,
                                      i16* ncount /*LDS*/, int& nsym, int& al_out, int lane)
{
    win.lo = src; win.hi = src + size;
    win.base = (const u8*)~(u64)0xFFFF;
    FwdBits b; b.start = src; b.bit = 0;
    int al = 5 + (int)b.read(win, 4, lane);
    if (al > max_al) return -1;
    int remaining = 1 << al;
    int s = 0;
    while (remaining > 0 && s <= max_sym) {
        int nb = highbit32((u32)remaining + 1) + 1;
        u32 val = b.read(win, nb, lane);
        u32 lower_mask = (1u << (nb - 1)) - 1;
        u32 threshold = (1u << nb) - 1 - ((u32)remaining + 1);
        if ((val & lower_mask) < threshold) { b.bit -= 1; val &= lower_mask; }
        else if (val > lower_mask) val -= threshold;
        int proba = (int)val - 1;
        remaining -= proba < 0 ? 1 : proba;
        lane0_guard();
        if (lane == 0) ncount[s] = (i16)proba;
        s++;
        if (proba == 0) {
            u32 rep = b.read(win, 2, lane);
            for (;;) {
                for (u32 i = 0; i < rep; i++) {
                    if (s > max_sym) return -1;
                    lane0_guard();
                    if (lane == 0) ncount[s] = 0;
                    s++;
                }
                if (rep != 3) break;
                rep = b.read(win, 2, lane);
            }
        }
    }
    if (remaining != 0) return -1;
    u64 used = (b.bit + 7) >> 3;
    if (used > size) return -1;
    nsym = s; al_out = al;
    return (int)used;
}

// Build one decode table from normalized counts — executed by ONE lane (three tables are built by
// three lanes side by side).  symtab: per-symbol baseline | extra bits << 24 in LDS (null: offset codes or
// Huffman weights, whose extra bits are the symbol itself resp. zero).  Returns false on a malformed
// distribution.
__device__ __noinline__ bool fse_build_lane(ZPK_LDS FseCell* tab, const ZPK_LDS i16* ncount, int nsym, int al, int kind, const ZPK_LDS u32* symtab,
                                            ZPK_LDS u8* spread, ZPK_LDS u16* nextc)
{
    const int size = 1 << al;
    int high = size;
    #pragma unroll 1
    for (int s = 0; s < nsym; s++)
        if (ncount[s] == -1) { spread[--high] = (u8)s; nextc[s] = 1; }
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    #pragma unroll 1
    for (int s = 0; s < nsym; s++) {
        int f = ncount[s];
        if (f <= 0) continue;
        nextc[s] = (u16)f;
        #pragma unroll 1
        for (int i = 0; i < f; i++) {
            spread[pos] = (u8)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return false;
    #pragma unroll 1
    for (int i = 0; i < size; i++) {
        const u32 s = spread[i];
        const u32 n = nextc[s]++;
        const u32 nb = (u32)al - (u32)highbit32(n);
        const u32 add = symtab ? symtab[s] >> 24 : (kind == T_OF ? s : 0u);
        tab[i] = fse_cell((n << nb) - (u32)size, nb, add, s);
    }
    return true;
}
#define LDSP(T, p) ((ZPK_LDS T*)(p))

__device__ __forceinline__ FseCell fse_rle_cell(int kind, u32 s)
{
    const u32 add = kind == T_LL ? Z_LL_BITS[s] : (kind == T_ML ? Z_ML_BITS[s] : s);
    return fse_cell(0, 0, add, s);
}

// uniform read of one cell (Huffman weight stream)
__device__ __forceinline__ FseCell lds_cell(const FseCell* tab, u32 state) { return uni(tab[state]); }

// ---- Huffman ------------------------------------------------------------------------------------

// weights[0..n) are in LDS (n includes the implied last weight).  Whole wave; returns false if malformed.
// Lane l holds symbols l, l + 64, l + 128, l + 192; the per-weight counts, every symbol's place among the symbols of its weight
// (natural order: RFC 8878 4.2.1) and the table fill are all lane-parallel (a lane-0 loop over up to 256 symbols, three times, was
// 30 % of the literals phase on text and a quarter of it on 256-symbol alphabets: tools/zx_stats.py, round 2).
template <class SH>
__device__ inline bool huf_build(SH& sh, int n, int lane)
{
    u32 w[4];
    #pragma unroll
    for (int p = 0; p < 4; p++) { const int i = lane + 64 * p; w[p] = i < n ? (u32)sh.weights[i] : 0u; }
    if (__ballot(w[0] > 12 || w[1] > 12 || w[2] > 12 || w[3] > 12) != 0) return false;
    u32 part = 0;
    #pragma unroll
    for (int p = 0; p < 4; p++) part += w[p] ? 1u << (w[p] - 1) : 0u;
    const u32 sum = (u32)__builtin_amdgcn_readlane((int)wave_scan_add(part), 63);
    if (sum == 0 || (sum & (sum - 1))) return false;
    const int mb = highbit32(sum);
    if (mb < 1 || mb > 12) return false;
    // counts per weight, and for every symbol the number of symbols of the same weight before it
    const u64 lt = (1ull << lane) - 1ull;
    u32 before[4] = {0, 0, 0, 0};
    u32 rank_w[4] = {0, 0, 0, 0};                      // first table index of the symbol's weight class
    u32 pos = 0, cnt1 = 0;
    #pragma unroll
    for (int wv = 1; wv <= 12; wv++) {
        u32 run = 0;
        #pragma unroll
        for (int p = 0; p < 4; p++) {
            const u64 m = __ballot(w[p] == (u32)wv);
            if (w[p] == (u32)wv) { before[p] = run + (u32)__popcll(m & lt); rank_w[p] = pos; }
            run += (u32)__popcll(m);
        }
        if (wv == 1) cnt1 = run;
        if (lane == 0) sh.huf_rank[wv] = wv <= mb ? pos : 0xFFFFFFFFu;
        pos += run << (wv - 1);
    }
    if (cnt1 < 2 || (cnt1 & 1)) return false;                          // libzstd HUF_readStats
    if (lane == 0) { sh.huf_rank[0] = 0xFFFFFFFFu; sh.huf_rank[13] = sh.huf_rank[14] = sh.huf_rank[15] = 0xFFFFFFFFu; sh.huf_max_bits = (u32)mb; }
    // fill: symbols of one weight take consecutive ranges in natural order.  max_bits <= 11 (what libzstd emits for literals): 2^11 u16
    // entries, symbol | code length << 8, fit the same 4 KiB, and the decoder gets the length with the symbol; a 12-bit code keeps the
    // byte table + the rank thresholds.  Ranges of < 64 entries are written by the symbol's own lane, longer ones by the whole wave.
    const bool wide = mb <= 11;
    #pragma unroll
    for (int p = 0; p < 4; p++) {
        const u32 i = (u32)(lane + 64 * p);
        const u32 len = w[p] ? 1u << (w[p] - 1) : 0u;
        const u32 base = rank_w[p] + before[p] * len;
        const u32 ent = i | (((u32)mb + 1u - w[p]) << 8);
        if (len && len < 64u) {
            if (wide) for (u32 k = 0; k < len; k++) ((ZPK_LDS u16*)sh.huf)[base + k] = (u16)ent;
            else for (u32 k = 0; k < len; k++) sh.huf[base + k] = (u8)i;
        }
        u64 big = __ballot(len >= 64u);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const u32 blen = (u32)__builtin_amdgcn_readlane((int)len, src), bbase = (u32)__builtin_amdgcn_readlane((int)base, src);
            const u32 bent = (u32)__builtin_amdgcn_readlane((int)ent, src);
            if (wide) for (u32 k = (u32)lane; k < blen; k += WAVE) ((ZPK_LDS u16*)sh.huf)[bbase + k] = (u16)bent;
            else for (u32 k = (u32)lane; k < blen; k += WAVE) sh.huf[bbase + k] = (u8)bent;
        }
    }
    __syncthreads();
    if (lane == 0) sh.huf_valid = 1;
    return true;
}

// RFC 8878 4.2.1 Huffman tree description.  Returns bytes consumed or -1.  Uniform.
template <class SH>
__device__ inline int huf_read_tree(SH& sh, ByteWindow& win, const u8* src, u64 size, int lane)
{
    if (size < 1) return -1;
    const u32 hb = uld8(src);
    int n = 0;
    u64 used;
    if (hb >= 128) {
        n = (int)hb - 127;
        u64 bytes = ((u64)n + 1) / 2;
        if (1 + bytes > size) return -1;
        for (int i = lane; i < n; i += WAVE) {
            u8 b = src[1 + i / 2];
            sh.weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        used = 1 + bytes;
    } else {
        const u64 csize = hb;
        if (csize == 0 || 1 + csize > size) return -1;
        int nsym = 0, al = 0;
        int tb = fse_read_ncount(win, src + 1, csize, 12, 6, sh.ncount[0], nsym, al, lane);
        if (tb < 0) return -1;
        __syncthreads();
        __shared__ u32 okb;
        lane0_guard();
        if (lane == 0) okb = fse_build_lane(LDSP(FseCell, sh.wt), LDSP(i16, sh.ncount[0]), nsym, al, 3, nullptr, LDSP(u8, sh.spread[0]), LDSP(u16, sh.nextc[0])) ? 1u : 0u;
        __syncthreads();
        if (!okb) return -1;
        RevBits b;
        if (!b.init(win, src + 1 + tb, csize - (u64)tb, lane)) return -1;
        u32 s1 = b.read(win, al, lane), s2 = b.read(win, al, lane);
        // two interleaved states; ends when an update over-reads (libzstd FSE_decompress tail)
        for (;;) {
            if (n > 253) return -1;
            const FseCell e1 = lds_cell(sh.wt, s1);
            lane0_guard();
            if (lane == 0) sh.weights[n] = (u8)cell_sym(e1);
            n++;
            s1 = cell_next(e1) + b.read(win, (int)cell_nb(e1), lane);
            if (b.pos < 0) { const FseCell e2 = lds_cell(sh.wt, s2); if (lane == 0) sh.weights[n] = (u8)cell_sym(e2); n++; break; }
            if (n > 253) return -1;
            const FseCell e2 = lds_cell(sh.wt, s2);
            lane0_guard();
            if (lane == 0) sh.weights[n] = (u8)cell_sym(e2);
            n++;
            s2 = cell_next(e2) + b.read(win, (int)cell_nb(e2), lane);
            if (b.pos < 0) { const FseCell e3 = lds_cell(sh.wt, s1); if (lane == 0) sh.weights[n] = (u8)cell_sym(e3); n++; break; }
        }
        used = 1 + csize;
    }
    __syncthreads();
    // implied last weight
    int lastw = -1;
    {
        u32 part = 0; bool ok = true;
        for (int i = lane; i < n; i += WAVE) { const u32 w = sh.weights[i]; if (w > 12) ok = false; else if (w) part += 1u << (w - 1); }
        const u32 sum = (u32)__builtin_amdgcn_readlane((int)wave_scan_add(part), 63);
        if (__ballot(!ok) == 0 && sum != 0) {
            const int mb = highbit32(sum) + 1;
            if (mb <= 12) {
                const u32 left = (1u << mb) - sum;
                if (!(left & (left - 1))) lastw = highbit32(left) + 1;
            }
        }
        lane0_guard();
        if (lastw >= 0 && lane == 0) sh.weights[n] = (u8)lastw;
    }
    __syncthreads();
    if (lastw < 0) return -1;
    if (!huf_build(sh, n + 1, lane)) return -1;
    return (int)used;
}

// per-lane backward bit container for the Huffman streams (divergent: each lane its own position).
// Bit positions are 32-bit: a stream is at most 128 KiB.
struct HufBits {
    const u8* start; const u8* rd_hi;
    i32 cb;               // container covers stream bits [cb*8, cb*8 + 128)
    u64 c_lo, c_hi, pre, pre2;  // pre / pre2 = the 8 + 8 bytes below c_lo, loaded two steps ahead: with 8-bit codes a step is
                                // only 8 symbols, less than a memory round trip under load
#ifdef HUF_PRE4
    u64 pre3, pre4;
#define HUF_SHIFT_DOWN() do { c_hi = c_lo; c_lo = pre; pre = pre2; pre2 = pre3; pre3 = pre4; cb -= 8; pre4 = fetch(cb - 32); } while (0)
#define HUF_SEEK_PRE() do { pre = fetch(cb - 8); pre2 = fetch(cb - 16); pre3 = fetch(cb - 24); pre4 = fetch(cb - 32); } while (0)
#else
#define HUF_SHIFT_DOWN() do { c_hi = c_lo; c_lo = pre; pre = pre2; cb -= 8; pre2 = fetch(cb - 16); } while (0)
#define HUF_SEEK_PRE() do { pre = fetch(cb - 8); pre2 = fetch(cb - 16); } while (0)
#endif

    __device__ __forceinline__ u64 fetch(i32 byte) const
    {
        const u8* a = start + byte;
        if (byte >= 0 && a + 8 <= rd_hi) return ld64(a);
        u64 v = 0;
        #pragma unroll 1
        for (int i = 0; i < 8; i++) { const i32 bb = byte + i; if (bb >= 0 && start + bb < rd_hi) v |= (u64)ld8(start + bb) << (8 * i); }
        return v;
    }
    __device__ __forceinline__ void seek(i32 pos)             // the next reads lie just below bit `pos`
    {
        cb = ((pos + 7) >> 3) - 16;
        c_lo = fetch(cb); c_hi = fetch(cb + 8); HUF_SEEK_PRE();
    }
    // the same with the top of the container just above `pos` (at most 127 bits above its bottom): what window() expects
    __device__ __forceinline__ void seek_w(i32 pos)
    {
        cb = ((pos + 7) >> 3) - 15;
        c_lo = fetch(cb); c_hi = fetch(cb + 8); HUF_SEEK_PRE();
    }
    // the 64 stream bits [pos - 64, pos), top-aligned (hi bit 31 = stream bit pos - 1): four codes of <= 11 bits can be decoded out of
    // it with 32-bit shifts and no further container arithmetic.  pos <= cb*8 + 127 (seek_w; positions only go down).
    __device__ __forceinline__ void window(i32 pos, u32& whi, u32& wlo)
    {
        while (pos - 64 < cb * 8) HUF_SHIFT_DOWN();
        const u32 rel = (u32)(pos - 64 - cb * 8);                 // 0..63
        const u32 d0 = (u32)c_lo, d1 = (u32)(c_lo >> 32), d2 = (u32)c_hi, d3 = (u32)(c_hi >> 32);
        const bool up = rel >= 32u;
        const u32 a0 = up ? d1 : d0, a1 = up ? d2 : d1, a2 = up ? d3 : d2;
        wlo = __builtin_amdgcn_alignbit(a1, a0, rel & 31u);
        whi = __builtin_amdgcn_alignbit(a2, a1, rel & 31u);
    }
    // bits [bp, bp+n), n <= 12; bp may be negative (zeros below 0)
    __device__ __forceinline__ u32 peek(i32 bp, int n)
    {
        while (bp < cb * 8) HUF_SHIFT_DOWN();
        const int rel = bp - cb * 8;
        const u64 v = rel >= 64 ? c_hi >> (rel - 64) : (c_lo >> rel) | ((c_hi << 1) << (63 - rel));
        return (u32)v & ((1u << n) - 1u);
    }
};

// One pass of one lane over its piece of a stream: decode from bit `entry` while the position is above `lo`.
// MODE 0: count symbols, remember the positions visited in the first 128 bits below `top` (bit masks) and six
//         CHECKPOINTS further down (the position before the 32nd, 64th, ... 1024th symbol, with the number of symbols
//         from there to the end of the piece);
// MODE 1: the same, but stop as soon as the position is one the previous pass visited — a mask bit or a checkpoint:
//         from there on the two passes are identical, so the previous count and exit are inherited.  The checkpoints
//         matter for codes that re-synchronise slowly (almost fixed-length codes of near-random literals): without them
//         a re-walk that merges after the mask window decodes the whole piece again, round after round;
// MODE 2: decode and store.
#define HUF_NCP 6
#define HUF_CP_NONE ((i32)0x80000000)
struct HufRun { i32 exit; u32 n; u64 m0, m1; i32 cp[HUF_NCP]; u32 cr[HUF_NCP]; };
template <int MODE, bool WIDE>
__device__ __forceinline__ HufRun huf_run(HufBits& b, const ZPK_LDS u8* huf, const u32 (&rk)[11], int mb, i32 entry, i32 lo, i32 top,
                                          const HufRun& old, u8* out, u64 deadline, bool& bad)
{
    HufRun r; r.n = 0; r.m0 = 0; r.m1 = 0;
    #pragma unroll
    for (int k = 0; k < HUF_NCP; k++) { r.cp[k] = HUF_CP_NONE; r.cr[k] = 0; }
    u32 wacc = 0;
    i32 pos = entry;
    b.seek_w(pos);
    bool merged = false, by_cp = false;
    // the highest checkpoint of the previous pass at or below the current position (positions only go down)
    i32 nxt = HUF_CP_NONE;
    if (MODE == 1) {
        #pragma unroll
        for (int k = 0; k < HUF_NCP; k++) if (old.cp[k] <= pos && old.cp[k] > nxt) nxt = old.cp[k];
    }
    if (MODE == 2 && WIDE) {
        // ---- store pass, codes of <= 11 bits: the piece's symbol count is known (old.n), four symbols per window, one dword store ----
        const ZPK_LDS u16* const tab = (const ZPK_LDS u16*)huf;
        const u32 n = old.n, sh = 32u - (u32)mb;
        u32 k = 0;
        #define HUF_ROUND4(acc_) do { u32 whi, wlo, used = 0; acc_ = 0; b.window(pos, whi, wlo);                                  \
            _Pragma("unroll") for (int t = 0; t < 4; t++) {                                                                       \
                const u32 ent = tab[whi >> sh]; const u32 nb = ent >> 8;                                                          \
                acc_ |= (ent & 0xFFu) << (8 * t);                                                                                 \
                whi = __builtin_amdgcn_alignbit(whi, wlo, 32u - nb); wlo <<= nb; used += nb;                                      \
            } pos -= (i32)used; } while (0)
        // sixteen symbols per store: every store of this pass is one more operation the next bitstream refill has to wait for (vmcnt
        // counts loads and stores together)
        for (; k + 16u <= n; k += 16u) {
            u32 a0, a1, a2, a3;
            HUF_ROUND4(a0); HUF_ROUND4(a1); HUF_ROUND4(a2); HUF_ROUND4(a3);
            u128 v; v.lo = (u64)a0 | ((u64)a1 << 32); v.hi = (u64)a2 | ((u64)a3 << 32);
            st128(out + k, v);
        }
        for (; k + 4u <= n; k += 4u) { u32 acc; HUF_ROUND4(acc); st32(out + k, acc); }
        #undef HUF_ROUND4
        if (k < n) {
            u32 whi, wlo;
            b.window(pos, whi, wlo);
            for (; k < n; k++) {
                const u32 ent = tab[whi >> sh];
                const u32 nb = ent >> 8;
                st8(out + k, (u8)ent);
                whi = __builtin_amdgcn_alignbit(whi, wlo, 32u - nb);
                wlo <<= nb;
                pos -= (i32)nb;
            }
        }
        r.n = n; r.exit = pos;
        return r;
    }
    if constexpr (WIDE && MODE != 2) {
        // ---- counting walks, codes of <= 11 bits: ROUNDS of four symbols out of one 64-bit window.  The positions a round visits are a
        // small bit mask L (bit u = the position `u` bits below the round's start); the 128-bit visited mask, the merge test against
        // the previous walk's mask and the checkpoint test are then a handful of operations per ROUND, not per symbol ----
        const ZPK_LDS u16* const tab = (const ZPK_LDS u16*)huf;
        const u32 sh = 32u - (u32)mb;
        const i32 need = 4 * mb;
        u32 rounds = 0;
        while (pos > lo) {
            if (r.n >= 32u && (r.n & (r.n - 1u)) == 0u && r.n <= (32u << (HUF_NCP - 1))) {
                #pragma unroll
                for (int k = 0; k < HUF_NCP; k++) if (r.n == (32u << k)) { r.cp[k] = pos; r.cr[k] = r.n; }       // cr: index for now
            }
            if ((++rounds & 255u) == 0u && __builtin_amdgcn_s_memrealtime() > deadline) { bad = true; break; }
            u32 whi, wlo, used = 0, live_n = 4u;
            u32 u1, u2, u3;                                       // the round's 2nd..4th symbol start this many bits below its start
            u64 Ltail = 0;
            b.window(pos, whi, wlo);
            const bool whole = pos - need > lo;                   // every symbol of the round starts above `lo`
            if (whole) {
                u32 nb = (u32)tab[whi >> sh] >> 8;
                whi = __builtin_amdgcn_alignbit(whi, wlo, 32u - nb); wlo <<= nb; u1 = nb;
                nb = (u32)tab[whi >> sh] >> 8;
                whi = __builtin_amdgcn_alignbit(whi, wlo, 32u - nb); wlo <<= nb; u2 = u1 + nb;
                nb = (u32)tab[whi >> sh] >> 8;
                whi = __builtin_amdgcn_alignbit(whi, wlo, 32u - nb); wlo <<= nb; u3 = u2 + nb;
                nb = (u32)tab[whi >> sh] >> 8;
                used = u3 + nb;
            } else {                                              // the end of the piece: symbol by symbol
                live_n = 0; u1 = u2 = u3 = 0;
                #pragma unroll
                for (int t = 0; t < 4; t++) {
                    const bool live = pos - (i32)used > lo;
                    const u32 nb = live ? (u32)tab[whi >> sh] >> 8 : 0u;
                    Ltail |= (live ? 1ull : 0ull) << used;
                    whi = nb ? __builtin_amdgcn_alignbit(whi, wlo, 32u - nb) : whi;
                    wlo <<= nb;
                    used += nb;
                    live_n += live ? 1u : 0u;
                }
            }
            // positions visited by this round: bit u = `u` bits below its start (u <= 33)
            #define HUF_ROUND_MASK() (whole ? ((u64)(1u | (1u << u1) | (1u << u2)) | (1ull << u3)) : Ltail)
            const u32 rel = (u32)(top - pos);
            if (MODE == 1) {
                const u64 L = HUF_ROUND_MASK();
                u64 slice = 0;                                    // the previous walk's visited positions, as seen from this round's start
                if (rel < 64u) slice = (old.m0 >> rel) | (rel ? old.m1 << (64u - rel) : 0ull);
                else if (rel < 128u) slice = old.m1 >> (rel - 64u);
                u64 hit = L & slice;
                const u32 cpoff = (u32)(pos - nxt);               // nxt <= pos (or NONE: huge)
                const bool cphit = nxt != HUF_CP_NONE && cpoff < 64u && ((L >> cpoff) & 1ull);
                if (cphit) hit |= 1ull << cpoff;
                if (hit) {
                    const u32 off = (u32)__ffsll((long long)hit) - 1u;
                    const u64 below = L & ((1ull << off) - 1ull);
                    by_cp = cphit && off == cpoff && !((L & slice) >> off & 1ull);
                    if (rel < 64u) { r.m0 |= below << rel; r.m1 |= rel ? below >> (64u - rel) : 0ull; }
                    else if (rel < 128u) r.m1 |= below << (rel - 64u);
                    r.n += (u32)__popcll(below);
                    pos -= (i32)off;
                    merged = true;
                    break;
                }
            }
            if (rel < 128u) {
                const u64 L = HUF_ROUND_MASK();
                if (rel < 64u) { r.m0 |= L << rel; r.m1 |= rel ? L >> (64u - rel) : 0ull; }
                else r.m1 |= L << (rel - 64u);
            }
            r.n += live_n;
            pos -= (i32)used;
            if (MODE == 1 && pos < nxt) {                         // passed it: the next checkpoint further down
                nxt = HUF_CP_NONE;
                #pragma unroll
                for (int k = 0; k < HUF_NCP; k++) if (old.cp[k] <= pos && old.cp[k] > nxt) nxt = old.cp[k];
            }
            #undef HUF_ROUND_MASK
        }
    } else {
    while (pos > lo) {
        const u32 rel = (u32)(top - pos);
        if (MODE != 2 && rel < 128) {
            const u64 bit = 1ull << (rel & 63);
            if (MODE == 1 && ((rel < 64 ? old.m0 : old.m1) & bit)) { merged = true; break; }
            if (rel < 64) r.m0 |= bit; else r.m1 |= bit;
        }
        if (MODE == 1 && pos <= nxt) {
            if (pos == nxt) { merged = true; by_cp = true; break; }
            nxt = HUF_CP_NONE;                                // passed it: the next one further down
            #pragma unroll
            for (int k = 0; k < HUF_NCP; k++) if (old.cp[k] <= pos && old.cp[k] > nxt) nxt = old.cp[k];
            if (pos == nxt) { merged = true; by_cp = true; break; }
        }
        if (MODE != 2 && r.n >= 32u && (r.n & (r.n - 1u)) == 0u && r.n <= (32u << (HUF_NCP - 1))) {
            #pragma unroll
            for (int k = 0; k < HUF_NCP; k++) if (r.n == (32u << k)) { r.cp[k] = pos; r.cr[k] = r.n; }       // cr: index for now
        }
        if ((r.n & 4095u) == 4095u && __builtin_amdgcn_s_memrealtime() > deadline) { bad = true; break; }
        const u32 idx = b.peek(pos - mb, mb);              // zeros below bit 0
        u32 nb, sym;
        if (WIDE) { const u32 ent = ((const ZPK_LDS u16*)huf)[idx]; nb = ent >> 8; sym = ent & 0xFFu; }
        else {
            nb = (u32)mb;                                    // code length: max_bits + 1 - weight class, the class from the class starts
            #pragma unroll
            for (int w = 0; w < 11; w++) nb -= idx >= rk[w] ? 1u : 0u;
            sym = MODE == 2 ? (u32)huf[idx] : 0u;
        }
        if (MODE == 2) {                                     // four symbols per store: byte stores cost a TA pass and a partial line each
            wacc |= sym << (8u * (r.n & 3u));
            if ((r.n & 3u) == 3u) { st32(out + (r.n - 3u), wacc); wacc = 0; }
        }
        pos -= (i32)nb;
        r.n++;
    }
    }
    if (MODE == 2) for (u32 k = r.n & ~3u; k < r.n; k++) { st8(out + k, (u8)wacc); wacc >>= 8; }
    r.exit = pos;
    if (MODE == 1 && merged) {
        const u32 rel = (u32)(top - pos);
        const u64 below0 = rel < 64 ? old.m0 & ((1ull << rel) - 1) : old.m0;
        const u64 below1 = rel < 64 ? 0ull : (rel < 128 ? old.m1 & ((1ull << (rel - 64)) - 1) : old.m1);
        u32 rest = old.n - (u32)(__popcll(below0) + __popcll(below1));        // mask merge: every earlier symbol of the old pass is in the masks
        if (by_cp) {
            #pragma unroll
            for (int k = 0; k < HUF_NCP; k++) if (old.cp[k] == pos) rest = old.cr[k];
        }
        // this pass's own checkpoints lie above the merge point and keep (index -> count to the end) after the sum below;
        // the old pass's checkpoints at or below it stay valid as they are, those above it belonged to the abandoned prefix
        #pragma unroll
        for (int k = 0; k < HUF_NCP; k++) {
            if (r.cp[k] != HUF_CP_NONE) r.cr[k] = r.n + rest - r.cr[k];
            else if (old.cp[k] != HUF_CP_NONE && old.cp[k] <= pos) { r.cp[k] = old.cp[k]; r.cr[k] = old.cr[k]; }
        }
        r.n += rest;
        r.m0 |= old.m0 & ~below0; r.m1 |= old.m1 & ~below1;
        r.exit = old.exit;
    } else if (MODE != 2) {
        #pragma unroll
        for (int k = 0; k < HUF_NCP; k++) if (r.cp[k] != HUF_CP_NONE) r.cr[k] = r.n - r.cr[k];
    }
    return r;
}

// Decode `nstreams` (1 or 4) Huffman streams into lit[0..regen) with all 64 lanes: every stream is cut into 16
// pieces by bit position and each lane decodes one.  Only the first piece starts on a code boundary; the others
// start wherever the cut fell, which is wrong — but prefix codes re-synchronise within a few symbols, so a lane
// re-decodes from its predecessor's exit only until it meets its own earlier trajectory (MODE 1), and the
// exits converge to the true chain after a few cheap rounds (lane 0 of a stream is right from the start, so
// the fixed point is the true decode).  A row prefix sum of the counts then places every piece, and one more
// pass stores the symbols.  Two full passes with 64 lanes instead of one with 4.
struct HufArgs { const u8* p; u64 size; int nstreams; u8* lit; u64 regen; const u8* rd_hi; u64 deadline; u64* dbg; };
__device__ __noinline__ bool huf_decode_streams(const ZPK_LDS u8* huf, const ZPK_LDS u32* rank, int mb, const HufArgs* a, int lane)
{
    const u8* const p = uni_ptr(a->p); const u64 size = uni64(a->size); const int nstreams = (int)uni((u32)a->nstreams);
    u8* const lit = uni_ptr(a->lit); const u64 regen = uni64(a->regen); const u8* const rd_hi = uni_ptr(a->rd_hi);
    const u64 deadline = uni64(a->deadline);
    const int grp = lane >> 4, j = lane & 15;
    const u8* sp = p; u64 ssz = size; u8* out = lit; u64 cnt = regen;
    if (nstreams == 4) {
        if (size < 10) return false;
        const u64 s1 = uld16(p), s2 = uld16(p + 2), s3 = uld16(p + 4);
        if (6 + s1 + s2 + s3 > size) return false;
        const u64 s4 = size - 6 - s1 - s2 - s3;
        const u64 seg = (regen + 3) / 4;
        if (seg * 3 > regen) return false;
        const u8* q = p + 6;
        if (grp == 0) { sp = q; ssz = s1; out = lit; cnt = seg; }
        else if (grp == 1) { sp = q + s1; ssz = s2; out = lit + seg; cnt = seg; }
        else if (grp == 2) { sp = q + s1 + s2; ssz = s3; out = lit + 2 * seg; cnt = seg; }
        else { sp = q + s1 + s2 + s3; ssz = s4; out = lit + 3 * seg; cnt = regen - 3 * seg; }
    }
    const bool act = grp < nstreams;
    bool bad = false;
    i32 P = 0;
    if (act) {
        const u32 last = ssz ? (u32)ld8(sp + ssz - 1) : 0u;
        if (last == 0 || ssz > (1u << 20)) bad = true;
        else P = (i32)(ssz - 1) * 8 + highbit32(last);
    }
    if (__ballot(bad) != 0) return false;
    u32 rk[11];
    #pragma unroll
    for (int w = 0; w < 11; w++) rk[w] = uni(rank[w + 2]);
    const i32 top = P - (i32)(((i64)P * j) >> 4), lo = P - (i32)(((i64)P * (j + 1)) >> 4);    // this lane's piece: positions (lo, top]
    HufBits b; b.start = sp; b.rd_hi = rd_hi;
    HufRun r; r.exit = top; r.n = 0; r.m0 = r.m1 = 0;
    #pragma unroll
    for (int k = 0; k < HUF_NCP; k++) { r.cp[k] = HUF_CP_NONE; r.cr[k] = 0; }
    i32 entry = top;
    const bool wide = mb <= 11;                                        // uniform: the table holds symbol | length (huf_build)
    if (act) r = wide ? huf_run<0, true>(b, huf, rk, mb, entry, lo, top, r, nullptr, deadline, bad)
                      : huf_run<0, false>(b, huf, rk, mb, entry, lo, top, r, nullptr, deadline, bad);
#ifdef ZPK_STATS
    const u64 t_fix0 = SEQ_T(); u32 n_iter = 0;
#endif
    for (int iter = 0; iter < 17; iter++) {
        i32 e = __shfl_up(r.exit, 1, 16);
        if (j == 0) e = P;
        const bool changed = act && e != entry;
        if (__ballot(changed) == 0) break;
#ifdef ZPK_STATS
        n_iter++;
#endif
        if (changed) {
            entry = e;
            r = wide ? huf_run<1, true>(b, huf, rk, mb, entry, lo, top, r, nullptr, deadline, bad)
                     : huf_run<1, false>(b, huf, rk, mb, entry, lo, top, r, nullptr, deadline, bad);
        }
    }
#ifdef ZPK_STATS
    if (a->dbg && lane == 0) { a->dbg[0] += SEQ_T() - t_fix0; a->dbg[1] += n_iter; }
#endif
    // place the pieces: prefix sum of the symbol counts inside each 16-lane row
    u32 x = act ? r.n : 0u;
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    const u32 total = (u32)__shfl((int)x, lane | 15, 64);
    const i32 last_exit = __shfl(r.exit, lane | 15, 64);
    if (act && ((u64)total != cnt || last_exit != 0)) bad = true;      // libzstd: exact symbol count and BIT_endOfDStream
    if (__ballot(bad) != 0) return false;
    if (act) {
        if (wide) (void)huf_run<2, true>(b, huf, rk, mb, entry, lo, top, r, out + (x - r.n), deadline, bad);
        else (void)huf_run<2, false>(b, huf, rk, mb, entry, lo, top, r, out + (x - r.n), deadline, bad);
    }
    return __ballot(bad) == 0;
}

// ---- one compressed block ------------------------------------------------------------------------

#ifdef ZPK_STATS
struct ZstdStats { u64 t_lit, t_tab, t_fse, t_exec, nseq, nblk; };
#define ZST(x) do { x; } while (0)
#else
struct ZstdStats { };
#define ZST(x) do { } while (0)
#endif
struct ZFrameState {
    ZstdStats* zs;
    Watchdog* wd;
    u64 rep0, rep1, rep2;
    bool seq_tables_valid;
    int al_ll, al_of, al_ml;
    const u64* pre;          // sequences of this entry already decoded by k_zstd_fse (zstd_fse4.h), or null
    u64 pre_idx;             // sequences of the entry consumed so far
};

// sequences table for one of LL/OF/ML; returns bytes consumed or -1
__device__ inline int read_seq_table(ZstdShared& sh, ByteWindow& win, int kind, int mode, const u8* src, u64 size,
                                     bool have_prev, int& al, int& pending_build, int& nsym_out, int lane)
{
    FseCell* tab = kind == T_LL ? sh.ll : (kind == T_OF ? sh.of : sh.ml);
    const int max_sym = kind == T_LL ? 35 : (kind == T_OF ? 31 : 52);
    const int max_al = kind == T_OF ? 8 : 9;
    if (mode == 0) {
        const FseCell* def = kind == T_LL ? sh.dll : (kind == T_OF ? sh.dof : sh.dml);
        const int n = kind == T_OF ? 32 : 64;
        for (int i = lane; i < n; i += WAVE) tab[i] = def[i];
        al = kind == T_OF ? 5 : 6;
        return 0;
    }
    if (mode == 1) {
        if (size < 1) return -1;
        u32 s = uld8(src);
        if ((int)s > max_sym) return -1;
        lane0_guard();
        if (lane == 0) tab[0] = fse_rle_cell(kind, s);
        al = 0;
        return 1;
    }
    if (mode == 2) {
        int nsym = 0, a = 0;
        int used = fse_read_ncount(win, src, size, max_sym, max_al, sh.ncount[kind], nsym, a, lane);
        if (used < 0) return -1;
        al = a; nsym_out = nsym;
        pending_build |= 1 << kind;
        return used;
    }
    return have_prev ? 0 : -1;
}

// The sequences section of one block: FSE decode + execution.  Out of line, with its own register allocation
// (inlined into the block parser the hot loop shared ~210 live VGPRs with header-parsing state); everything it
// needs comes in through a small argument block, the tables through an LDS-typed pointer.
struct ZSeqArgs {
    const u8* p; u64 left; u64 nseq;
    int al_ll, al_of, al_ml;
    u32 rep0, rep1, rep2;                 // in/out
    const u8* lit; u64 lit_size; int lit_rle; u32 lit_rle_byte;
    u8* op; u8* oend; u8* frame_lo;       // op: in/out
    u64 lit_pos;                          // out: literals consumed
    u64 deadline; int timed_out;
    ZstdStats* zs;
    const u64* pre;                       // this block's sequences, packed (zstd_fse4.h), when pre-decoded
    u32 work_bytes;                       // pre-decoded: bytes of LDS at sh->ll the executor may use
};

// The same section when k_zstd_fse has already decoded it: 64 packed sequences per load, then execution.
__device__ __noinline__ int zstd_sequences_pre(ZPK_LDS ZstdShared* sh, ZSeqArgs* a, int lane)
{
    // the FSE decode tables are idle in this mode: their 5 KiB serve as the executor's batch assembly buffer (seq_exec.h)
    // and as a 1 KiB window over the literal stream, refilled with one coalesced load when a batch leaves it, so that
    // literal runs are read with ds_read_b128 instead of up to four exact-tail vector loads per lane
    const lds_p8 asm_buf = (lds_p8)sh->ll;
    const u32 work = (u32)__builtin_amdgcn_readfirstlane((int)a->work_bytes);          // >= ZSEQ_LWIN + 16 + 64 + a useful assembly size
    const u32 asm_cap = work - (ZSEQ_LWIN + 16u) - 16u;
    const lds_p8 lwin = asm_buf + (work - (ZSEQ_LWIN + 16u));                          // ZSEQ_LWIN bytes + 16 of read slack
    u64 win_lo = ~0ull;
    const u64 nseq = uni64(a->nseq);
    const u64* const pre = uni_ptr(a->pre);
    const u8* const lit = uni_ptr(a->lit); const u64 lit_size = uni64(a->lit_size);
    const bool lit_rle = a->lit_rle != 0; const u32 lit_rle_byte = a->lit_rle_byte;
    u8* op = uni_ptr(a->op); u8* const oend = uni_ptr(a->oend); u8* const frame_lo = uni_ptr(a->frame_lo);
    u64 lit_pos = 0;
    u64 zt0 = SEQ_T(); (void)zt0;
    u64 nxt = 0;                                          // the next batch's packed sequences, loaded one batch ahead
    if ((u64)lane < nseq) nxt = *(const ZPK_GLOBAL u64*)(pre + (u64)lane);
    for (u64 base = 0; base < nseq; base += WAVE) {
        const int cnt = (int)(nseq - base < WAVE ? nseq - base : WAVE);
        if (__builtin_amdgcn_s_memrealtime() > a->deadline) { a->timed_out = 1; return D_MALFORMED; }
        u32 my_ll = 0, my_ml = 0, my_off = 1;
        const u64 v = nxt;
        if (base + WAVE + (u64)lane < nseq) nxt = *(const ZPK_GLOBAL u64*)(pre + base + WAVE + (u64)lane);
        if (lane < cnt) { my_off = (u32)v & ((1u << 29) - 1u); my_ml = (u32)(v >> 29) & ((1u << 18) - 1u); my_ll = (u32)(v >> 47); }
        u32 xl = wave_scan_add(lane < cnt ? my_ll : 0u);
        const u64 lit_total = (u32)__builtin_amdgcn_readlane((int)xl, 63);
        if (lit_total > lit_size - lit_pos) return D_MALFORMED;
        if (__ballot(lane < cnt && my_off == 0) != 0) return D_MALFORMED;
        SeqBatch q;
        const u64 my_lit = lit_pos + (xl - (lane < cnt ? my_ll : 0u));
        q.lit = lit + my_lit;
        q.lit_lds = SEQ_NO_LDS; q.ll = my_ll; q.ml = my_ml; q.off = my_off; q.bad = 0;
        if (!lit_rle && lit_total != 0) {
            if (lit_total <= ZSEQ_LWIN && (lit_pos < win_lo || lit_pos + lit_total > win_lo + ZSEQ_LWIN)) {
                wave_mem_fence();
                win_lo = lit_pos;
                const u64 o = win_lo + 16u * (u64)lane;
                u128 w; w.lo = 0; w.hi = 0;
                if (o + 16 <= lit_size) w = ld128(lit + o);
                else for (u64 k = o; k < lit_size && k < o + 16; k++) { const u64 x = (u64)ld8(lit + k); if (k - o < 8) w.lo |= x << (8 * (k - o)); else w.hi |= x << (8 * (k - o - 8)); }
                lds_st128(lwin + 16u * (u32)lane, w);
                wave_mem_fence();
            }
            if (lane < cnt && my_ll <= SEQ_OWN_MAX && my_lit >= win_lo && my_lit + my_ll <= win_lo + ZSEQ_LWIN) q.lit_lds = (u32)(my_lit - win_lo);
        }
        SeqStats stt = {};
        (void)stt;
        const int rc = seq_exec_batch(q, cnt, op, oend, frame_lo, lit_rle ? (int)lit_rle_byte : -1, lane, stt, (lds_cp8)lwin, asm_buf, asm_cap);
        if (rc != D_OK) { a->op = op; return rc; }
        lit_pos += lit_total;
    }
    ZST({ u64 t = SEQ_T(); a->zs->t_exec += t - zt0; zt0 = t; });
    a->op = op; a->lit_pos = lit_pos;
    return D_OK;
}

__device__ __noinline__ int zstd_sequences(ZPK_LDS ZstdShared* sh, ZSeqArgs* a, int lane)
{
    ByteWindow win;
    const u8* const p = uni_ptr(a->p); const u64 left = uni64(a->left), nseq = uni64(a->nseq);
    const u8* const lit = uni_ptr(a->lit); const u64 lit_size = uni64(a->lit_size);
    const bool lit_rle = a->lit_rle != 0; const u32 lit_rle_byte = a->lit_rle_byte;
    u8* op = uni_ptr(a->op); u8* const oend = uni_ptr(a->oend); u8* const frame_lo = uni_ptr(a->frame_lo);
    u64 lit_pos = 0;
    u64 zt0 = SEQ_T(); (void)zt0;
    {
        RevBits b;
        if (!b.init(win, p, left, lane)) return D_MALFORMED;
        const u32 sll = b.read(win, a->al_ll, lane);
        const u32 sof = b.read(win, a->al_of, lane);
        const u32 sml = b.read(win, a->al_ml, lane);
        // ---- the three FSE chains run side by side in lanes ----
        // Lane t in {0,1,2} (OF, ML, LL) extracts its symbol's value bits, lane 7-t the bits of the same chain's
        // state update: the six fields of a sequence sit in the stream in exactly that lane order (OF, ML, LL
        // values, then LL, ML, OF states), so their bit offsets are one DPP prefix sum over lanes 0..7, every
        // field is cut out of a uniform 128-bit container with per-lane shifts, and row_half_mirror hands the new
        // state to the partner lane.  One ds_read_b64 fetches all three next table entries.  The critical path
        // per sequence is table read -> ~15 dependent vector instructions -> table read, instead of ~7 serial
        // scalar bit-reader calls.
        const int role = lane < 3 ? lane : 7 - lane;                      // 0 OF, 1 ML, 2 LL (lanes 0..2 and 7..5)
        const bool chain = lane < 3 || (lane >= 5 && lane < 8);
        const ZPK_LDS u32* const tab = role == 0 ? sh->of : (role == 1 ? sh->ml : sh->ll);   // other lanes: harmless reads of ll[]
        // value baselines: one table per chain, no branch (offset codes: 1 << code, full 32 bits)
        const ZPK_LDS u32* const symt = role == 0 ? sh->ofbase : (role == 2 ? sh->symtab_ll : sh->symtab_ml);
        const u32 symmask = role == 0 ? 0xFFFFFFFFu : 0xFFFFFFu;
        FseCell cell = tab[role == 0 ? sof : (role == 1 ? sml : (chain ? sll : 0u))];
        u32 rep0 = a->rep0, rep1 = a->rep1, rep2 = a->rep2;    // live in lane 0
        ZPK_LDS u32* const seqbuf = sh->seqbuf;
        // uniform stream position, 32-bit (a block is at most 128 KiB): bits not yet consumed; may go negative
        const u8* const start = b.start;
        i32 pos = (i32)uni((u32)b.pos);
        i32 wb = (i32)uni((u32)(i64)(win.base - start));       // window base as a byte offset from the stream start

        for (u64 base = 0; base < nseq; base += WAVE) {
            const int cnt = (int)(nseq - base < WAVE ? nseq - base : WAVE);
            if (__builtin_amdgcn_s_memrealtime() > a->deadline) { a->timed_out = 1; return D_MALFORMED; }
            wave_mem_fence();
            for (int k = 0; k < cnt; k++) {
                // ---- 128-bit container: stream bits [pos-128, pos), five v_readlane out of the register window ----
                const i32 lowbit = pos - 128;
                const i32 byte = lowbit >> 3;                                   // floor
                i32 d = byte - wb;
                if ((u32)d > 236u) {
                    win.load((const u8*)(((u64)(start + byte) & ~(u64)3) - 232), lane);
                    wb = (i32)uni((u32)(i64)(win.base - start));
                    d = byte - wb;
                }
                const int wi = d >> 2;
                const u32 t = (u32)(d & 3) * 8u + (u32)(lowbit & 7);            // <= 31
                const u32 w0 = (u32)__builtin_amdgcn_readlane((int)win.w, wi), w1 = (u32)__builtin_amdgcn_readlane((int)win.w, wi + 1);
                const u32 w2 = (u32)__builtin_amdgcn_readlane((int)win.w, wi + 2), w3 = (u32)__builtin_amdgcn_readlane((int)win.w, wi + 3);
                const u32 w4 = (u32)__builtin_amdgcn_readlane((int)win.w, wi + 4);
                const u64 qa = ((u64)w1 << 32) | w0, qb = ((u64)w3 << 32) | w2, qc = ((u64)w4 << 32) | w3;
                const u64 clo = (qa >> t) | ((qb << 1) << (63u - t));           // [pos-128, pos-64)
                const u64 chi = (u64)(u32)(qb >> t) | ((qc >> t) << 32);        // [pos-64, pos)
                // ---- field widths and offsets ----
                const u32 sym = cell_sym(cell);
                const u32 e_base = symt[sym];                                    // consumed last
                u32 n = lane < 3 ? cell_add(cell) : cell_nb(cell);
                n = chain ? n : 0u;
                u32 s = n;
                s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xf, 0xf, false);
                s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xf, 0xf, false);
                s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xf, false);
                const u32 off = s - n;                                         // bits of the fields before this lane's
                const u32 total = (u32)__builtin_amdgcn_readlane((int)s, 7);
                const u32 o6 = off & 63u;
                const u64 xa = (chi << o6) | ((clo >> 1) >> (63u - o6)), xb = clo << o6;
                const u64 x = off < 64 ? xa : xb;
                const u32 bits = (u32)((x >> 1) >> (63u - n));                  // n = 0 -> 0
                // libzstd 1.4.9 updates all three states after every sequence, the last included
                const u32 nst = cell_next(cell) + bits;                        // lanes 5..7: next state
                const u32 nst_m = (u32)__builtin_amdgcn_update_dpp(0, (int)nst, 0x141, 0xf, 0xf, false);   // row_half_mirror
                cell = tab[lane < 3 ? nst_m : (chain ? nst : 0u)];
                const u32 val = (e_base & symmask) + bits;                     // lanes 0..2: offset value, match length, literal length
                // repeat offsets (RFC 8878 3.1.1.5), on lane 0
                const u32 llv = (u32)__builtin_amdgcn_update_dpp(0, (int)val, 0xE6, 0xf, 0xf, false);      // quad_perm [2,1,2,3]
                u32 offset;
                {
                    const bool big = val > 3;
                    const u32 idx = val - 1 + (llv == 0 ? 1u : 0u);           // 0..3 when !big
                    u32 tt = idx == 3 ? rep0 - 1 : (idx == 1 ? rep1 : rep2);
                    if (tt == 0) tt = 1;                                       // libzstd: forced to 1 on corrupt input
                    offset = big ? val - 3 : (idx == 0 ? rep0 : tt);
                    const bool shift = big || idx != 0;
                    const u32 n2 = (big || idx != 1) ? rep1 : rep2;
                    if (shift) { rep2 = n2; rep1 = rep0; rep0 = offset; }
                }
                if (lane < 3) seqbuf[k * 3 + lane] = lane == 0 ? offset : val;
                pos = (i32)uni((u32)(pos - (i32)total));
            }
            wave_mem_fence();
            u32 my_ll = 0, my_ml = 0; u64 my_off = 0;
            if (lane < cnt) { my_off = seqbuf[lane * 3]; my_ml = seqbuf[lane * 3 + 1]; my_ll = seqbuf[lane * 3 + 2]; }
            ZST({ u64 t = SEQ_T(); a->zs->t_fse += t - zt0; zt0 = t; });
            // ---- execute the batch lane-parallel (seq_exec.h): literal sources are a prefix sum over the literal buffer
            {
                u32 xl = lane < cnt ? my_ll : 0u;
                #pragma unroll
                for (int d = 1; d < 64; d <<= 1) { u32 y = (u32)__shfl_up((int)xl, d, 64); if (lane >= d) xl += y; }
                const u64 lit_total = (u32)__builtin_amdgcn_readlane((int)xl, 63);
                if (lit_total > lit_size - lit_pos) return D_MALFORMED;
                if (__ballot(lane < cnt && (my_off == 0 || my_off > 0xFFFFFFFFull)) != 0) return D_MALFORMED;
                SeqBatch q;
                q.lit = lit + lit_pos + (xl - (lane < cnt ? my_ll : 0u));
                q.lit_lds = SEQ_NO_LDS; q.ll = my_ll; q.ml = my_ml; q.off = (u32)my_off; q.bad = 0;
                SeqStats stt = {};
                (void)stt;
                const int rc = seq_exec_batch(q, cnt, op, oend, frame_lo, lit_rle ? (int)lit_rle_byte : -1, lane, stt);
                if (rc != D_OK) { a->op = op; return rc; }
                lit_pos += lit_total;
                ZST({ u64 t = SEQ_T(); a->zs->t_exec += t - zt0; zt0 = t; });
            }
        }
        a->rep0 = (u32)__builtin_amdgcn_readfirstlane((int)rep0); a->rep1 = (u32)__builtin_amdgcn_readfirstlane((int)rep1);
        a->rep2 = (u32)__builtin_amdgcn_readfirstlane((int)rep2);
        if (pos > 0) return D_MALFORMED;             // libzstd 1.4.9: the stream must not be under-consumed
    
    }
    a->op = op; a->lit_pos = lit_pos;
    return D_OK;
}

// The literals section of a compressed block (RFC 8878 3.1.1.3.1): raw literals are used where they lie, RLE literals are one byte,
// Huffman-coded ones are decoded into `lit_buf` (the workgroup's scratch).  `used` = bytes of the block the section takes.
struct ZLiterals { const u8* lit; u64 lit_size, used; bool rle; u32 rle_byte; };
template <class SH>
__device__ inline int zstd_literals(SH& sh, ZFrameState& fs, const u8* src, u64 size, const u8* rd_hi, u8* lit_buf, ZLiterals& L, int lane)
{
    ByteWindow win;
    const u32 b0 = uld8(src);
    const u32 type = b0 & 3, fmt = (b0 >> 2) & 3;
    const u8* lit = lit_buf; u64 lit_size = 0, used = 0;
    bool lit_rle = false; u32 lit_rle_byte = 0;
    if (type < 2) {
        u64 hl, n;
        if ((fmt & 1) == 0) { hl = 1; n = b0 >> 3; }
        else if (fmt == 1) { hl = 2; n = (b0 >> 4) | ((u64)uld8(src + 1) << 4); }
        else { hl = 3; n = (b0 >> 4) | ((u64)uld8(src + 1) << 4) | ((u64)uld8(src + 2) << 12); }
        if (n > ZSTD_BLOCK_MAX) return D_MALFORMED;
        if (type == 0) { if (hl + n > size) return D_MALFORMED; lit = src + hl; used = hl + n; }
        else { if (hl + 1 > size) return D_MALFORMED; lit_rle = true; lit_rle_byte = uld8(src + hl); used = hl + 1; }
        lit_size = n;
    } else {
        if (size < 5) return D_MALFORMED;
        u64 hl, regen, csize; int streams;
        const u64 v = uld32(src);
        if (fmt == 0) { hl = 3; streams = 1; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; }
        else if (fmt == 1) { hl = 3; streams = 4; regen = (v >> 4) & 0x3FF; csize = (v >> 14) & 0x3FF; }
        else if (fmt == 2) { hl = 4; streams = 4; regen = (v >> 4) & 0x3FFF; csize = v >> 18; }
        else { hl = 5; streams = 4; regen = (v >> 4) & 0x3FFFF; csize = (v >> 22) | ((u64)uld8(src + 4) << 10); }
        if (regen > ZSTD_BLOCK_MAX || hl + csize > size) return D_MALFORMED;
        const u8* p = src + hl; u64 left = csize;
        if (type == 2) {
            const u64 zt_tree = SEQ_T(); (void)zt_tree;
            int t = huf_read_tree(sh, win, p, left, lane);
            if (t < 0) return D_MALFORMED;
            p += t; left -= (u64)t;
            ZST(fs.zs->nseq += SEQ_T() - zt_tree);               // (stats builds of the execute-only kernel: nseq = cycles in the tree description)
        } else if (!sh.huf_valid) return D_MALFORMED;
        __syncthreads();
        HufArgs ha; ha.p = p; ha.size = left; ha.nstreams = streams; ha.lit = lit_buf; ha.regen = regen; ha.rd_hi = rd_hi; ha.deadline = fs.wd->deadline;
        ha.dbg = nullptr;
        ZST(ha.dbg = (u64*)&fs.zs->t_tab);                   // (stats builds: t_tab = cycles in the fix-up rounds, nseq = their number)
        if (!huf_decode_streams(LDSP(u8, sh.huf), LDSP(u32, sh.huf_rank), (int)sh.huf_max_bits, &ha, lane)) return D_MALFORMED;
        wave_mem_fence();
        lit_size = regen; used = hl + csize;
    }
    L.lit = lit; L.lit_size = lit_size; L.used = used; L.rle = lit_rle; L.rle_byte = lit_rle_byte;
    return D_OK;
}

template <bool EXEC_ONLY>
__device__ inline int zstd_block(ZstdShared& sh, ZFrameState& fs, const u8* src, u64 size, const u8* rd_hi,
                                 u8* dst, u64 dst_cap, u8* frame_lo, u8* lit_buf, u64& produced, int lane)
{
    ByteWindow win;
    if (size < 3) return D_MALFORMED;
    u64 zt0 = SEQ_T(); (void)zt0;
    // ---- literals section ----
    ZLiterals zl;
    { const int lrc = zstd_literals(sh, fs, src, size, rd_hi, lit_buf, zl, lane); if (lrc != D_OK) return lrc; }
    const u8* const lit = zl.lit; const u64 lit_size = zl.lit_size, used = zl.used;
    const bool lit_rle = zl.rle; const u32 lit_rle_byte = zl.rle_byte;
    ZST({ u64 t = SEQ_T(); fs.zs->t_lit += t - zt0; zt0 = t; fs.zs->nblk++; });
    // ---- sequences header ----
    const u8* p = src + used;
    u64 left = size - used;
    if (left < 1) return D_MALFORMED;
    u64 nseq = uld8(p);
    if (nseq == 0) { if (left != 1) return D_MALFORMED; p += 1; left -= 1; }
    else if (nseq < 128) { p += 1; left -= 1; }
    else if (nseq < 255) { if (left < 2) return D_MALFORMED; nseq = ((nseq - 128) << 8) + uld8(p + 1); p += 2; left -= 2; }
    else { if (left < 3) return D_MALFORMED; nseq = (u64)uld8(p + 1) + ((u64)uld8(p + 2) << 8) + 0x7F00; p += 3; left -= 3; }

    u8* op = dst; u8* oend = dst + dst_cap;
    u64 lit_pos = 0;
    if (nseq > 0) {
        if (left < 1) return D_MALFORMED;
        ZSeqArgs sa;
        sa.pre = nullptr;
        sa.work_bytes = EXEC_ONLY ? ZSTD_EXEC_WORK : 5120u;
        if (EXEC_ONLY || fs.pre) {
            // k_zstd_fse has validated the tables and the bitstream of this block and decoded it
            sa.pre = fs.pre + fs.pre_idx;
            fs.pre_idx += nseq;
        }
        if constexpr (!EXEC_ONLY) if (!fs.pre) {
        const u32 modes = uld8(p);
        p += 1; left -= 1;
        int pending = 0, ns[3] = {0, 0, 0};
        int r;
        r = read_seq_table(sh, win, T_LL, (modes >> 6) & 3, p, left, fs.seq_tables_valid, fs.al_ll, pending, ns[T_LL], lane);
        if (r < 0) return D_MALFORMED;
        p += r; left -= (u64)r;
        r = read_seq_table(sh, win, T_OF, (modes >> 4) & 3, p, left, fs.seq_tables_valid, fs.al_of, pending, ns[T_OF], lane);
        if (r < 0) return D_MALFORMED;
        p += r; left -= (u64)r;
        r = read_seq_table(sh, win, T_ML, (modes >> 2) & 3, p, left, fs.seq_tables_valid, fs.al_ml, pending, ns[T_ML], lane);
        if (r < 0) return D_MALFORMED;
        p += r; left -= (u64)r;
        __syncthreads();
        {   // the FSE-described tables are built side by side, one lane each
            bool ok = true;
            lane0_guard();
            // one call, three lanes: the per-lane arguments select the table (a call per table would serialise them)
            const int kd = lane < 3 ? lane : 0;
            const int kal = kd == T_LL ? fs.al_ll : (kd == T_OF ? fs.al_of : fs.al_ml);
            const int kns = kd == T_LL ? ns[T_LL] : (kd == T_OF ? ns[T_OF] : ns[T_ML]);
            FseCell* const ktab = kd == T_LL ? sh.ll : (kd == T_OF ? sh.of : sh.ml);
            const u32* const ksym = kd == T_LL ? sh.symtab_ll : sh.symtab_ml;
            if (lane < 3 && (pending & (1 << lane)))
                ok = fse_build_lane(LDSP(FseCell, ktab), LDSP(i16, sh.ncount[kd]), kns, kal, kd, kd == T_OF ? (const ZPK_LDS u32*)nullptr : LDSP(u32, ksym),
                                    LDSP(u8, sh.spread[kd]), LDSP(u16, sh.nextc[kd]));
            if (__ballot(!ok) != 0) return D_MALFORMED;
        }
        __syncthreads();
        fs.seq_tables_valid = true;
        ZST({ u64 t = SEQ_T(); fs.zs->t_tab += t - zt0; zt0 = t; fs.zs->nseq += nseq; });
        }

        sa.p = p; sa.left = left; sa.nseq = nseq;
        sa.al_ll = fs.al_ll; sa.al_of = fs.al_of; sa.al_ml = fs.al_ml;
        sa.rep0 = (u32)fs.rep0; sa.rep1 = (u32)fs.rep1; sa.rep2 = (u32)fs.rep2;
        sa.lit = lit; sa.lit_size = lit_size; sa.lit_rle = lit_rle ? 1 : 0; sa.lit_rle_byte = lit_rle_byte;
        sa.op = op; sa.oend = oend; sa.frame_lo = frame_lo; sa.lit_pos = 0;
        sa.deadline = fs.wd->deadline; sa.timed_out = 0; sa.zs = fs.zs;
        int src_rc;
        if constexpr (EXEC_ONLY) src_rc = zstd_sequences_pre((ZPK_LDS ZstdShared*)&sh, &sa, lane);
        else src_rc = sa.pre ? zstd_sequences_pre((ZPK_LDS ZstdShared*)&sh, &sa, lane) : zstd_sequences((ZPK_LDS ZstdShared*)&sh, &sa, lane);
        if (sa.timed_out) fs.wd->fired = true;
        op = sa.op;
        if (src_rc != D_OK) { produced = (u64)(op - dst); return src_rc; }
        if (!sa.pre) { fs.rep0 = sa.rep0; fs.rep1 = sa.rep1; fs.rep2 = sa.rep2; }
        lit_pos = sa.lit_pos;
    }
    const u64 rest = lit_size - lit_pos;
    if (rest > (u64)(oend - op)) { produced = (u64)(op - dst); return D_DST_FULL; }
    if (lit_rle) {
        u128 pat; pat.lo = 0x0101010101010101ull * (u64)(lit_rle_byte & 0xFFu); pat.hi = pat.lo;
        for (u64 i = (u64)lane * 16; i < rest; i += WAVE * 16) gstore_upto16(op + i, pat, (u32)(rest - i < 16 ? rest - i : 16));
    } else for (u64 i = (u64)lane * 16; i < rest; i += WAVE * 16) gcopy_upto16(op + i, lit + lit_pos + i, (u32)(rest - i < 16 ? rest - i : 16));
    op += rest;
    wave_mem_fence();
    if ((u64)(op - dst) > ZSTD_BLOCK_MAX) return D_MALFORMED;
    produced = (u64)(op - dst);
    return D_OK;
}

// ---- frames --------------------------------------------------------------------------------------

__device__ inline void zstd_build_defaults(ZstdShared& sh, int lane)
{
    for (int i = lane; i < 36; i += WAVE) { sh.ncount[T_LL][i] = Z_LL_DEF[i]; sh.symtab_ll[i] = Z_LL_BASE[i] | ((u32)Z_LL_BITS[i] << 24); }
    for (int i = lane; i < 29; i += WAVE) sh.ncount[T_OF][i] = Z_OF_DEF[i];
    for (int i = lane; i < 32; i += WAVE) sh.ofbase[i] = 1u << i;
    for (int i = lane; i < 53; i += WAVE) { sh.ncount[T_ML][i] = Z_ML_DEF[i]; sh.symtab_ml[i] = Z_ML_BASE[i] | ((u32)Z_ML_BITS[i] << 24); }
    __syncthreads();
    lane0_guard();
    {
        const int kd = lane < 3 ? lane : 0;
        FseCell* const ktab = kd == T_LL ? sh.dll : (kd == T_OF ? sh.dof : sh.dml);
        const u32* const ksym = kd == T_LL ? sh.symtab_ll : sh.symtab_ml;
        if (lane < 3)
            fse_build_lane(LDSP(FseCell, ktab), LDSP(i16, sh.ncount[kd]), kd == T_LL ? 36 : (kd == T_OF ? 29 : 53), kd == T_OF ? 5 : 6, kd,
                           kd == T_OF ? (const ZPK_LDS u32*)nullptr : LDSP(u32, ksym), LDSP(u8, sh.spread[kd]), LDSP(u16, sh.nextc[kd]));
    }
    __syncthreads();
}

// Streaming (zpk_stream.inc): a decode that runs out of INPUT in front of a block (or a frame header, or a frame's checksum) leaves
// everything the next block depends on in a ZstdResume record in memory — positions, frame header fields, repeat offsets, and the
// whole LDS image (Huffman table for treeless literals, the three sequence tables for Repeat_Mode, the predefined tables) — and is
// picked up there when more bytes have arrived.  `final`: all of the entry's bytes are present, so a shortage is a malformed frame
// (the one-shot verdict, lib/zpack_read.c:380-388), not a reason to wait.
struct alignas(16) ZstdResume {
    u64 ip_off, op_off, frame_lo_off, fcs, pre_idx;
    u64 rep0, rep1, rep2;
    u32 phase;                   // 0 = in front of a frame, 1 = in front of a block, 2 = in front of the frame's checksum
    u32 fn, cksum, seq_tables_valid;
    i32 al_ll, al_of, al_ml, pad_;
    u8  lds[(sizeof(ZstdShared) + 15) & ~15u];
};

__device__ inline void zstd_lds_image(ZstdShared& sh, u8* image, bool save, int lane)
{
    __syncthreads();
    const u32 n16 = (u32)(sizeof(ZstdShared) / 16);
    lds_p8 L = to_lds_rw((u8*)&sh);
    for (u32 i = (u32)lane; i < n16; i += WAVE) {
        if (save) st128(image + 16 * i, lds_ld128((lds_cp8)(L + 16 * i)));
        else lds_st128(L + 16 * i, ld128(image + 16 * i));
    }
    __syncthreads();
}

template <bool EXEC_ONLY = false>
__device__ inline DecodeOut zstd_decode_wave(ZstdShared& sh, Watchdog& wd, const u8* src, u64 src_size, u8* dst, u64 dst_cap, u8* lit_buf, int lane,
                                             ZstdStats* zs = nullptr, const u64* pre = nullptr, ZstdResume* rs = nullptr, bool final = true)
{
    u64 pre_idx = 0;
    DecodeOut r; r.rc = D_OK; r.produced = 0;
    const u8* ip = src; const u8* iend = src + src_size;
    u8* op = dst; u8* oend = dst + dst_cap;
    bool resuming = rs && uni(rs->phase) != 0;
    if constexpr (!EXEC_ONLY) if (!resuming && !sh.defaults_built) { zstd_build_defaults(sh, lane); sh.defaults_built = 1; }
    __syncthreads();
    if (rs && !resuming && uni(rs->ip_off | rs->op_off) != 0) { ip = src + uni64(rs->ip_off); op = dst + uni64(rs->op_off); }    // in front of a later frame
    // a shortage of input: wait (streaming, more to come) or malformed (everything is here)
    const int SHORT = rs && !final ? D_TRUNCATED : D_MALFORMED;
    ZFrameState fs;
    fs.wd = &wd; fs.zs = zs;
    u32 fn = 0, cksum = 0; u64 fcs = 0; u8* frame_lo = op;
    // what a resumed call needs: written (lane 0) whenever the decode stops short of input
    auto park = [&](u32 phase, const u8* at) {
        if (!rs) return;
        if (phase != 0) zstd_lds_image(sh, rs->lds, true, lane);
        lane0_guard();
        if (lane == 0) {
            rs->phase = phase; rs->ip_off = (u64)(at - src); rs->op_off = (u64)(op - dst); rs->frame_lo_off = (u64)(frame_lo - dst);
            rs->fcs = fcs; rs->fn = fn; rs->cksum = cksum; rs->pre_idx = fs.pre_idx;
            rs->rep0 = fs.rep0; rs->rep1 = fs.rep1; rs->rep2 = fs.rep2; rs->seq_tables_valid = fs.seq_tables_valid ? 1u : 0u;
            rs->al_ll = fs.al_ll; rs->al_of = fs.al_of; rs->al_ml = fs.al_ml;
        }
    };

    while (ip < iend || resuming) {
        if (wd.expired()) { r.rc = D_MALFORMED; break; }
        u32 phase = 1;
        if (resuming) {
            resuming = false;
            phase = uni(rs->phase);
            ip = src + uni64(rs->ip_off); op = dst + uni64(rs->op_off); frame_lo = dst + uni64(rs->frame_lo_off);
            fcs = uni64(rs->fcs); fn = uni(rs->fn); cksum = uni(rs->cksum);
            fs.rep0 = uni64(rs->rep0); fs.rep1 = uni64(rs->rep1); fs.rep2 = uni64(rs->rep2); fs.seq_tables_valid = uni(rs->seq_tables_valid) != 0;
            fs.al_ll = (int)uni((u32)rs->al_ll); fs.al_of = (int)uni((u32)rs->al_of); fs.al_ml = (int)uni((u32)rs->al_ml);
            fs.pre = pre; fs.pre_idx = uni64(rs->pre_idx);
            zstd_lds_image(sh, rs->lds, false, lane);
        } else {
            const u8* const frame_at = ip;
            frame_lo = op; fn = 0; cksum = 0; fcs = 0;
            fs.rep0 = 1; fs.rep1 = 4; fs.rep2 = 8; fs.seq_tables_valid = false; fs.al_ll = fs.al_of = fs.al_ml = 0; fs.pre = pre; fs.pre_idx = pre_idx;
            if (iend - ip < 4) { r.rc = SHORT; park(0, frame_at); break; }
            const u32 magic = uld32(ip);
            if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
                if (iend - ip < 8) { r.rc = SHORT; park(0, frame_at); break; }
                u64 sz = uld32(ip + 4);
                if ((u64)(iend - ip) - 8 < sz) { r.rc = SHORT; park(0, frame_at); break; }
                ip += 8 + sz;
                continue;
            }
            if (magic != 0xFD2FB528u) { r.rc = D_MALFORMED; break; }
            // ---- frame header ----
            if (iend - ip < 6) { r.rc = SHORT; park(0, frame_at); break; }
            ip += 4;
            const u32 fhd = uld8(ip++);
            const u32 fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
            cksum = (fhd >> 2) & 1;
            if (fhd & 0x08) { r.rc = D_MALFORMED; break; }
            if (!single) {
                if (iend - ip < 1) { r.rc = SHORT; park(0, frame_at); break; }
                const u32 wdesc = uld8(ip++);
                if (10 + (wdesc >> 3) > 31) { r.rc = D_MALFORMED; break; }
            }
            const u32 dn = did_flag == 3 ? 4 : did_flag;
            if ((u64)(iend - ip) < dn) { r.rc = SHORT; park(0, frame_at); break; }
            u32 dict_id = 0;
            for (u32 i = 0; i < dn; i++) dict_id |= uld8(ip + i) << (8 * i);
            ip += dn;
            if (dict_id != 0) { r.rc = D_MALFORMED; break; }
            fn = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
            if ((u64)(iend - ip) < fn) { r.rc = SHORT; park(0, frame_at); break; }
            for (u32 i = 0; i < fn; i++) fcs |= (u64)uld8(ip + i) << (8 * i);
            if (fn == 2) fcs += 256;
            ip += fn;
            lane0_guard();
            if (lane == 0) sh.huf_valid = 0;
            __syncthreads();
        }
        bool fail = false;
        while (phase == 1) {
            if (wd.expired()) { r.rc = D_MALFORMED; fail = true; break; }
            const u8* const block_at = ip;
            if (iend - ip < 3) { r.rc = SHORT; park(1, block_at); fail = true; break; }
            const u32 bh = uld8(ip) | (uld8(ip + 1) << 8) | (uld8(ip + 2) << 16);
            ip += 3;
            const bool last = bh & 1; const u32 type = (bh >> 1) & 3; const u64 bsize = bh >> 3;
            if (type == 3) { r.rc = D_MALFORMED; fail = true; break; }
            if (type == 0) {
                if (bsize > (u64)(iend - ip)) { r.rc = SHORT; park(1, block_at); fail = true; break; }
                if (bsize > (u64)(oend - op)) { r.rc = D_DST_FULL; fail = true; break; }
                for (u64 i = (u64)lane * 16; i < bsize; i += WAVE * 16) gcopy_upto16(op + i, ip + i, (u32)(bsize - i < 16 ? bsize - i : 16));
                ip += bsize; op += bsize;
            } else if (type == 1) {
                if (iend - ip < 1) { r.rc = SHORT; park(1, block_at); fail = true; break; }
                if (bsize > (u64)(oend - op)) { r.rc = D_DST_FULL; fail = true; break; }
                const u8 v = (u8)uld8(ip);
                {
                    u128 pat; pat.lo = 0x0101010101010101ull * v; pat.hi = pat.lo;
                    for (u64 i = (u64)lane * 16; i < bsize; i += WAVE * 16) gstore_upto16(op + i, pat, (u32)(bsize - i < 16 ? bsize - i : 16));
                }
                ip += 1; op += bsize;
            } else {
                if (bsize >= ZSTD_BLOCK_MAX) { r.rc = D_MALFORMED; fail = true; break; }
                if (bsize > (u64)(iend - ip)) { r.rc = SHORT; park(1, block_at); fail = true; break; }
                u64 got = 0;
                int rc = zstd_block<EXEC_ONLY>(sh, fs, ip, bsize, iend, op, (u64)(oend - op), frame_lo, lit_buf, got, lane);
                if (rc != D_OK) { r.rc = rc; fail = true; break; }
                ip += bsize; op += got;
            }
            wave_mem_fence();
            if (last) break;
        }
        if (fail) break;
        pre_idx = fs.pre_idx;
        if (fn != 0 && (u64)(op - frame_lo) != fcs) { r.rc = D_MALFORMED; break; }
        if (cksum) {
            if (iend - ip < 4) { r.rc = SHORT; park(2, ip); break; }
            u32 h = 0;
            lane0_guard();
            if (lane == 0) h = (u32)xxh64_serial(frame_lo, (u64)(op - frame_lo), 0);
            if (uni(h) != uld32(ip)) { r.rc = D_MALFORMED; break; }
            ip += 4;
        }
        // a frame is complete: a later call starts in front of the next one
        if (rs) { lane0_guard(); if (lane == 0) { rs->phase = 0; rs->ip_off = (u64)(ip - src); rs->op_off = (u64)(op - dst); } }
    }
    r.produced = (u64)(op - dst);
    return r;
}

}  // namespace zpk
