// zstd_fse4.h — Zstandard sequences pre-decode: FOUR FSE streams per wave, one per 16-lane DPP row.
//
// Why.  The sequences section of a block (RFC 8878 3.1.1.3.2) is a serial state machine: ~700 cycles of dependent
// work per sequence for a wave, and the number of streams a CU can have in flight is bounded by the LDS their
// three decode tables take.  The fused kernel (zstd_wg.h) spends a whole wave, 168 VGPRs and 13 KiB of LDS on ONE
// stream (12 per CU) and 77 % of a text entry's time in that chain.  Here a stream costs a quarter of a wave and
// 4.9 KiB (3-byte table cells + a 512-byte bitstream ring + construction scratch): 32 streams per CU, and a wave
// instruction advances four chains at once.
//
// What.  k_zstd_fse walks every frame and block of an entry exactly like zstd_decode_wave, skips the literals
// sections, builds the block's LL/OF/ML tables, decodes the sequences (repeat offsets resolved) and writes them
// packed into 8 bytes each to the batch's sequence arena (HBM), in stream order.  k_zstd then runs the entry with
// the sequences already decoded (literals + execution + checksum only).  ANYTHING unusual — a malformed header,
// a length that does not fit the packing, an arena region that is too small — just leaves the entry unmarked and
// k_zstd decodes it in full; and when a pre-decoded run fails or its XXH3 does not match, k_zstd repeats the
// entry in full as well, so the verdict of every entry is the fused decoder's (the reference's, zpack_read.c:380).
//
// Rows are independent state machines inside one wave: a persistent loop in which every row that is in the
// middle of a block advances by one sequence, and rows that are between blocks run the (divergent, rare)
// header / table-construction code first.  All per-row values are row-uniform VGPRs — nothing here may use
// readfirstlane.
#pragma once
#include "zstd_wg.h"

namespace zpk {

#define ZF_ROWS 4
#define ZF_CHUNK 64u                               // the backward bitstream moves through the ring in 64-byte chunks
#define ZF_RING (2u * ZF_CHUNK)
#define ZF_SEQBUF 8u                               // packed sequences a row collects (in registers, one per lane) before it stores them: one 64-byte store
#define ZF_SEQ_OFF_BITS 29                         // packed sequence: offset | match length << 29 | literal length << 47
#define ZF_SEQ_ML_BITS 18
#define ZF_SEQ_LL_BITS 17
#ifndef ZF_WG_PER_CU
#define ZF_WG_PER_CU 12u                           // LDS: 12 452 B per workgroup = ten 1280-byte allocation units; 12 x 10 of the CU's 128
#endif
#define ZF_GRID_MAX (256u * ZF_WG_PER_CU)
#define ZF_HEAD 8                                  // counters[] word used as this kernel's dequeue head
#define ZF_WATCHDOG_WORD 11                        // counters[11]: entries given up by the row watchdog, [12]: header-loop budget hits
#define ZF_COUNT_WORD 1                            // counters[] word holding the length of the Zstandard work list (L_ZSTD)

// LDS is what bounds the number of streams in flight, and the stage's throughput is proportional to that number (measured, round 2:
// 4 / 6 / 8 workgroups per CU -> 34.7 / 25.7 / 19.2 ms on 16 384 text entries).  So a decode-table cell is TWO bytes here:
//   cell = next-state counter n (10 bits: a symbol of count c owns the counters c .. 2c-1 <= 1023) | symbol << 10
// and what the RFC 8878 4.1.1 table holds besides is recomputed from it: nb_bits = accuracy_log - highbit(n),
// next_base = (n << nb_bits) - table size; the number of extra value bits comes with the symbol's baseline (one LDS read).  A stream
// then takes 2.5 KiB of tables + a 136-byte bitstream ring + 248 B of counts = 2 944 B.
struct ZfTab { u16 c_ll[512], c_ml[512], c_of[256]; };
#define ZF_NC_LL 0                                 // normalized counts: LL 36 symbols, OF 32, ML 53
#define ZF_NC_OF 36
#define ZF_NC_ML 68
struct alignas(8) ZfRow {
    ZfTab t;
    u8  ring[ZF_RING + 8];                         // + mirror of the first 8 bytes (the dword pair of a read may straddle the wrap)
    i16 ncount[124];                               // normalized counts, then (in place) the per-symbol next-state counters
};
struct alignas(16) ZfShared {
    ZfRow row[ZF_ROWS];
    u16 d_ll[64], d_ml[64], d_of[32];              // predefined distributions
    u32 base_ll[36], base_ml[53];                  // LL / ML codes: value baseline | extra bits << 24 (offset codes: 1 << code, `code` bits)
};
static_assert(sizeof(ZfShared) <= 12800, "k_zstd_fse: 12 workgroups per CU need <= 10 LDS allocation units each");
__device__ __forceinline__ int zf_nc_base(int kind) { return kind == T_LL ? ZF_NC_LL : (kind == T_OF ? ZF_NC_OF : ZF_NC_ML); }

// Decode table from normalized counts, by ONE lane, in place: the spread symbols are parked in the table itself
// and the next-state counters overwrite the counts.  Same construction as fse_build_lane (zstd_wg.h).
__device__ __noinline__ bool fse_build_inplace(ZPK_LDS u16* t, ZPK_LDS i16* nc, int nsym, int al)
{
    const int size = 1 << al;
    int high = size;
    #pragma unroll 1
    for (int s = 0; s < nsym; s++)
        if (nc[s] == -1) t[--high] = (u16)s;
    const int step = (size >> 1) + (size >> 3) + 3, mask = size - 1;
    int pos = 0;
    #pragma unroll 1
    for (int s = 0; s < nsym; s++) {
        const int f = nc[s];
        if (f <= 0) continue;
        #pragma unroll 1
        for (int i = 0; i < f; i++) {
            if (pos < high) t[pos] = (u16)s;                    // positions >= high belong to the -1 symbols: skipped below
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return false;
    #pragma unroll 1
    for (int s = 0; s < nsym; s++) { const int f = nc[s]; nc[s] = (i16)(f == -1 ? 1 : (f > 0 ? f : 0)); }
    #pragma unroll 1
    for (int i = 0; i < size; i++) {
        const u32 s = t[i];
        const u32 n = (u32)(u16)nc[s];                          // count <= n < 2 x count <= 2 x size: 10 bits, and highbit(n) <= al
        nc[s] = (i16)(n + 1);
        t[i] = (u16)(n | (s << 10));
    }
    return true;
}

// 16 bytes at stream offset `byte` of [bs, bs+size), zero outside
__device__ __forceinline__ u128 zf_load16(const u8* bs, i32 size, i32 byte)
{
    u128 v; v.lo = 0; v.hi = 0;
    if (byte >= 0 && byte + 16 <= size) return ld128(bs + byte);
    if (byte + 16 <= 0 || byte >= size) return v;
    #pragma unroll 1
    for (int i = 0; i < 16; i++) {
        const i32 b = byte + i;
        if (b >= 0 && b < size) { const u64 x = (u64)ld8(bs + b); if (i < 8) v.lo |= x << (8 * i); else v.hi |= x << (8 * (i - 8)); }
    }
    return v;
}
// 4 bytes at stream offset `byte` of [bs, bs+size), zero outside
__device__ __forceinline__ u32 zf_load4(const u8* bs, i32 size, i32 byte)
{
    if (byte >= 0 && byte + 4 <= size) return ld32(bs + byte);
    u32 v = 0;
    #pragma unroll
    for (int i = 0; i < 4; i++) { const i32 b = byte + i; if (b >= 0 && b < size) v |= (u32)ld8(bs + b) << (8 * i); }
    return v;
}
// chunk `chunk` of the stream (64 bytes: 4 per lane of the row) into its half of the ring
__device__ __forceinline__ void zf_ring_put(lds_p8 ring, i32 chunk, int sub, u32 v)
{
    const u32 slot = ((u32)chunk & 1u) * ZF_CHUNK;
    *(ZPK_LDS u32*)(ring + slot + 4u * (u32)sub) = v;
    if (slot == 0 && sub < 2) *(ZPK_LDS u32*)(ring + ZF_RING + 4u * (u32)sub) = v;
}
// The table descriptions of a block (up to 256 bytes) are staged in REGISTERS, 16 bytes per lane of the row, and read with
// ds_bpermute: dword W (row-uniform) is component W & 3 of lane W >> 2; zero beyond the 256 bytes
__device__ __forceinline__ u32 zf_stage_word(const u128& d, u32 W, int lane)
{
    const u32 c = W & 3u;
    const u32 mine = c == 0 ? (u32)d.lo : (c == 1 ? (u32)(d.lo >> 32) : (c == 2 ? (u32)d.hi : (u32)(d.hi >> 32)));
    const u32 v = (u32)__shfl((int)mine, (lane & ~15) | (int)((W >> 2) & 15u), 64);
    return W < 64u ? v : 0u;
}
__device__ __forceinline__ u32 zf_stage_bits(const u128& d, u32 bit, int lane)          // 32 bits from bit offset `bit`
{
    const u32 W = bit >> 5;
    return __builtin_amdgcn_alignbit(zf_stage_word(d, W + 1, lane), zf_stage_word(d, W, lane), bit & 31u);
}
// 32 stream bits starting at bit `bit`, out of the ring: TWO ALIGNED dword reads + one v_alignbit.  (The one 8-byte read at any byte
// offset this replaces was the kernel's bottleneck: rocprofv3 round 2, SQ_LDS_UNALIGNED_STALL 2.9e9 + SQ_LDS_IDX_ACTIVE 3.9e9 of 6.9e9
// CU-cycles — an LDS access that is not naturally aligned costs ~26 extra LDS cycles on gfx950.)
__device__ __forceinline__ u32 zf_ring_bits(lds_cp8 ring, i32 bit)
{
    const u32 byte = (u32)(bit >> 3) & (ZF_RING - 1u);
    const ZPK_LDS u32* w = (const ZPK_LDS u32*)(ring + (byte & ~3u));          // the 8-byte mirror behind the ring covers w[1] at the wrap
    const u32 d0 = w[0], d1 = w[1];
    return __builtin_amdgcn_alignbit(d1, d0, ((byte & 3u) << 3) | ((u32)bit & 7u));
}
// n (<= 31) stream bits starting at bit `bit` (may be negative: zeros), out of the ring
__device__ __forceinline__ u32 zf_bits(lds_cp8 ring, i32 bit, u32 n)
{
    return zf_ring_bits(ring, bit) & ((1u << n) - 1u);
}
__device__ __forceinline__ u32 lds_ld32u(lds_cp8 p) { return ((const ZPK_LDS pk32*)p)->v; }

// RFC 8878 4.1.1 out of the register stage, from its byte `o` on (row-uniform).  Returns bytes consumed or -1.
__device__ __forceinline__ int zf_read_ncount(const u128& stage, u32 o, u32 size, u32 room, int max_sym, int max_al, ZPK_LDS i16* nc, int& nsym, int& al_out,
                                              int lane)
{
    u32 bit = 0;
    #define ZF_RD(n) ({ const u32 w_ = zf_stage_bits(stage, 8u * o + bit, lane); const u32 v_ = w_ & ((1u << (n)) - 1u); bit += (u32)(n); v_; })
    const int al = 5 + (int)ZF_RD(4);
    if (al > max_al) return -1;
    int remaining = 1 << al;
    int s = 0;
    while (remaining > 0 && s <= max_sym) {
        if ((bit >> 3) + 4u > room) return -1;                 // `room` readable (staged or zeroed) bytes follow buf
        const int nb = highbit32((u32)remaining + 1) + 1;
        u32 val = ZF_RD(nb);
        const u32 lower_mask = (1u << (nb - 1)) - 1;
        const u32 threshold = (1u << nb) - 1 - ((u32)remaining + 1);
        if ((val & lower_mask) < threshold) { bit -= 1; val &= lower_mask; }
        else if (val > lower_mask) val -= threshold;
        const int proba = (int)val - 1;
        remaining -= proba < 0 ? 1 : proba;
        nc[s] = (i16)proba;
        s++;
        if (proba == 0) {
            u32 rep = ZF_RD(2);
            for (;;) {
                for (u32 i = 0; i < rep; i++) {
                    if (s > max_sym) return -1;
                    nc[s] = 0;
                    s++;
                }
                if (rep != 3) break;
                if ((bit >> 3) + 4u > room) return -1;
                rep = ZF_RD(2);
            }
        }
    }
    #undef ZF_RD
    if (remaining != 0) return -1;
    const u32 used = (bit + 7) >> 3;
    if (used > size) return -1;
    nsym = s; al_out = al;
    return (int)used;
}

enum { ZF_NEED_ENTRY = 0, ZF_NEED_FRAME = 1, ZF_NEED_BLOCK = 2, ZF_DECODING = 3, ZF_DONE = 4 };

// state[e] = 1: entry e's sequences are in the arena (region of its output slot, 8 bytes per sequence); 0: not.
//
// BLOCKS (zstd_pj.h: the blocks of ONE large frame side by side): a work item is one compressed BLOCK — desc[e].src_offset = its
// 3-byte block header, comp_size = 3 + Block_Size, its arena region = desc[e].dst_offset / 8, dst_capacity / 8 sequences — not an
// entry: no frame header, the item ends with the block.  What a block inherits from the blocks before it are the three repeat
// offsets, unknown here: the row starts with three SYMBOLS instead (ZF_SYM(1..3): values >= 2^27, where no real offset of this
// path lies; "rep0 - 1" counts down inside the symbol), offsets that come from the inherited history leave as symbols, and the
// block's final history goes to rep_out[3 e ..] for the scan over the blocks that gives every symbol its value (k_zpj_reps).
#define ZF_SYM_SHIFT 27
#define ZF_SYM(j) (((u32)(j) << ZF_SYM_SHIFT) | ((1u << ZF_SYM_SHIFT) - 1u))
template <bool BLOCKS>
__device__ __forceinline__ void zstd_fse_rows(ZfShared& sh, const u8* __restrict__ src, const zpk_decode_desc* __restrict__ desc,
                                              const u32* __restrict__ list, u32* __restrict__ counters,
                                              u64* __restrict__ arena, u32* __restrict__ state, u32* __restrict__ rep_out, u64 src_size)
{
    const int lane = lane_id();
    const int row = lane >> 4, sub = lane & 15;
    ZPK_LDS ZfRow* const R = (ZPK_LDS ZfRow*)&sh.row[row];
    const lds_p8 ring = (lds_p8)R->ring;

    // ---- predefined tables + symbol tables, once per workgroup ----
    for (int i = lane; i < 36; i += WAVE) { sh.row[0].ncount[i] = Z_LL_DEF[i]; sh.base_ll[i] = Z_LL_BASE[i] | ((u32)Z_LL_BITS[i] << 24); }
    for (int i = lane; i < 29; i += WAVE) sh.row[1].ncount[i] = Z_OF_DEF[i];
    for (int i = lane; i < 53; i += WAVE) { sh.row[2].ncount[i] = Z_ML_DEF[i]; sh.base_ml[i] = Z_ML_BASE[i] | ((u32)Z_ML_BITS[i] << 24); }
    __syncthreads();
    if (lane < 3) {
        ZPK_LDS u16* const t = lane == T_LL ? (ZPK_LDS u16*)sh.d_ll : (lane == T_OF ? (ZPK_LDS u16*)sh.d_of : (ZPK_LDS u16*)sh.d_ml);
        fse_build_inplace(t, (ZPK_LDS i16*)sh.row[lane].ncount, lane == T_LL ? 36 : (lane == T_OF ? 29 : 53), lane == T_OF ? 5 : 6);
    }
    __syncthreads();

    // ---- per-row state (row-uniform unless noted) ----
    int phase = ZF_NEED_ENTRY;
    u32 e = 0;
    const u8* ip = nullptr; const u8* iend = nullptr;
    u64 a_base = 0; u32 seq_cap = 0, seq_n = 0;
    u32 rep0 = 1, rep1 = 4, rep2 = 8;
    u32 tab_ll = 0, tab_of = 0, tab_ml = 0, tab_modes = 0x3F;      // BLOCKS: where the table a Repeat_Mode kind inherits is described (offset in src), and how (2 bits per kind; 3 = nowhere)
    bool tables_valid = false, last_block = false, cksum = false, bad = false;
    int al_ll = 0, al_of = 0, al_ml = 0;
    u32 seq_end = 0;                                     // seq_n at the end of the row's current block
    const u8* bs = nullptr; i32 bs_size = 0, pos = 0, loaded_lo = 0, refill_below = 0;
    u32 pf = 0;                                          // per lane: its 4 bytes of the chunk below the ring
    u32 cell = 0;                                        // per lane (chain lanes): next-state counter | symbol << 10
    u32 my_al = 0, al_m31 = 0;                           // per lane: accuracy log of the lane's chain in the current block (and that minus 31)
    // roles inside a row (zstd_wg.h): lanes 0,1,2 cut the OF, ML, LL value bits, lanes 7,6,5 the state bits of the same chains
    const int role = sub < 3 ? sub : 7 - sub;            // 0 OF, 1 ML, 2 LL for chain lanes
    const bool chain = sub < 3 || (sub >= 5 && sub < 8);
    const ZPK_LDS u16* const tab = role == 0 ? (const ZPK_LDS u16*)R->t.c_of : (role == 1 ? (const ZPK_LDS u16*)R->t.c_ml : (const ZPK_LDS u16*)R->t.c_ll);
    const ZPK_LDS u32* const symt = role == 2 ? (const ZPK_LDS u32*)sh.base_ll : (const ZPK_LDS u32*)sh.base_ml;     // (offset lanes: unused)
    // value lanes (0..2 of a row) cut the extra bits of a code, state lanes (5..7) the bits of the state update.  The role of a lane is
    // DATA (bit masks in VGPRs, opaque to the compiler), not control flow: per-role exec masks in the hot loop were SGPR spills + branches
    u32 vmask = sub < 3 ? ~0u : 0u, smask = (chain && sub >= 3) ? ~0u : 0u, omask = role == 0 ? ~0u : 0u;
    asm volatile("" : "+v"(vmask), "+v"(smask), "+v"(omask));
    const u32 tmask = chain ? 511u : 0u;                 // idle lanes read cell 0
    u32 badv = 0;
    u32 acc_lo = 0, acc_hi = 0;                          // per lane: lane j of a row holds packed sequence (seq_n - 1 - j) of the row's entry
    const u32 nz = counters[ZF_COUNT_WORD];
    // A row must never hold the GPU: an entry gets the size-proportional budget of zpk_device.h (like the fused decoder), and the header
    // loop a fixed budget of steps per wave; either limit just hands the entry (or the rest of the list) to k_zstd.
    u64 row_deadline = 0;
    u32 setup_steps = 0;

    for (;;) {
        if (phase != ZF_DONE && phase != ZF_NEED_ENTRY && __builtin_amdgcn_s_memrealtime() > row_deadline) {
            bad = true; ip = iend; phase = ZF_NEED_FRAME;
            lane0_guard();
            if (sub == 0) atomicAdd(&counters[ZF_WATCHDOG_WORD], 1u);
            lane0_guard();
        }
        // =================== between blocks: headers, tables, stream start (divergent, rare) ===================
        if (phase < ZF_DECODING) {
            for (;;) {
                if (++setup_steps > (1u << 24)) {                      // cannot happen: every step consumes input or finishes an entry
                    lane0_guard();
                    if (sub == 0) atomicAdd(&counters[ZF_WATCHDOG_WORD + 1], 1u);
                    lane0_guard();
                    phase = ZF_DONE; break;
                }
                if (phase == ZF_NEED_ENTRY) {
                    lane0_guard();
                    u32 v = 0;
                    if (sub == 0) v = atomicAdd(&counters[ZF_HEAD], 1u);
                    lane0_guard();
                    v = (u32)__shfl((int)v, lane & ~15, 64);
                    if (v >= nz) { phase = ZF_DONE; break; }
                    e = list[v];
                    const zpk_decode_desc d = desc[e];
                    ip = src + d.src_offset; iend = ip + d.comp_size;
                    const u64 lo8 = (d.dst_offset + 7) & ~7ull, hi8 = (d.dst_offset + d.dst_capacity) & ~7ull;
                    a_base = lo8 >> 3;
                    const u64 cap = hi8 > lo8 ? (hi8 - lo8) >> 3 : 0;
                    seq_cap = cap > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (u32)cap;
                    seq_n = 0; bad = false; badv = 0;
                    row_deadline = __builtin_amdgcn_s_memrealtime() + watchdog_budget(d.comp_size + d.dst_capacity);
                    phase = ZF_NEED_FRAME;
                    if constexpr (BLOCKS) {                              // the item IS a block: no frame header, inherited history as symbols
                        rep0 = ZF_SYM(1); rep1 = ZF_SYM(2); rep2 = ZF_SYM(3); tables_valid = false; cksum = false;
                        tab_ll = (u32)d.uncomp_size; tab_of = (u32)(d.uncomp_size >> 32); tab_ml = (u32)d.expect_hash; tab_modes = (u32)(d.expect_hash >> 32) & 0x3Fu;
                        phase = ZF_NEED_BLOCK;
                    }
                }
                if (phase == ZF_NEED_FRAME) {
                    // frame header, as zstd_decode_wave
                    if (ip >= iend) {                                  // every frame of the entry walked: flush and publish
                        const u32 rem = seq_n & (ZF_SEQBUF - 1u);
                        if (!bad && rem && (u32)sub < rem) arena[a_base + (seq_n - 1u) - (u32)sub] = ((u64)acc_hi << 32) | acc_lo;
                        lane0_guard();
                        if (sub == 0) {
                            if constexpr (BLOCKS) { rep_out[3u * e] = rep0; rep_out[3u * e + 1u] = rep1; rep_out[3u * e + 2u] = rep2; }
                            state[e] = bad ? 0u : 1u; if (!bad) atomicAdd(&counters[ZF_WATCHDOG_WORD + 2], 1u);
                        }
                        lane0_guard();
                        phase = ZF_NEED_ENTRY;
                        continue;
                    }
                    bool ok = iend - ip >= 4;
                    u32 magic = ok ? ld32(ip) : 0u;
                    if (ok && (magic & 0xFFFFFFF0u) == 0x184D2A50u) {
                        ok = iend - ip >= 8;
                        const u64 sz = ok ? (u64)ld32(ip + 4) : 0;
                        if (ok && (u64)(iend - ip) - 8 >= sz) { ip += 8 + sz; continue; }
                        ok = false;
                    }
                    if (ok && magic != 0xFD2FB528u) ok = false;
                    if (ok && iend - ip < 6) ok = false;
                    if (ok) {
                        ip += 4;
                        const u32 fhd = ld8(ip++);
                        const u32 fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, did_flag = fhd & 3;
                        cksum = (fhd >> 2) & 1;
                        if (fhd & 0x08) ok = false;
                        if (ok && !single) {
                            if (iend - ip < 1) ok = false;
                            else { const u32 wdesc = ld8(ip++); if (10 + (wdesc >> 3) > 31) ok = false; }
                        }
                        const u32 dn = did_flag == 3 ? 4 : did_flag;
                        if (ok && (u64)(iend - ip) < dn) ok = false;
                        if (ok) { for (u32 i = 0; i < dn; i++) if (ld8(ip + i) != 0) ok = false; ip += dn; }
                        const u32 fn = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
                        if (ok && (u64)(iend - ip) < fn) ok = false;
                        if (ok) ip += fn;
                    }
                    if (!ok) { bad = true; ip = iend; continue; }       // leave the entry to the fused decoder
                    rep0 = 1; rep1 = 4; rep2 = 8; tables_valid = false;
                    phase = ZF_NEED_BLOCK;
                }
                // ---- ZF_NEED_BLOCK ----
                if (iend - ip < 3) { bad = true; ip = iend; phase = ZF_NEED_FRAME; continue; }
                const u32 bh = (u32)ld8(ip) | ((u32)ld8(ip + 1) << 8) | ((u32)ld8(ip + 2) << 16);
                ip += 3;
                last_block = BLOCKS ? true : (bh & 1) != 0;
                const u32 btype = (bh >> 1) & 3; const u64 bsize = bh >> 3;
                bool ok = btype != 3;
                u64 adv = btype == 1 ? 1 : bsize;
                if (ok && adv > (u64)(iend - ip)) ok = false;
                if (ok && btype == 2 && bsize >= ZSTD_BLOCK_MAX) ok = false;
                if (!ok) { bad = true; ip = iend; phase = ZF_NEED_FRAME; continue; }
                const u8* const blk = ip;
                ip += adv;
                bool to_decode = false;
                if (btype == 2) {
                    // literals section: only its size matters here (zstd_block)
                    const u64 size = bsize;
                    u64 used = 0;
                    ok = size >= 3;
                    if (ok) {
                        const u32 b0 = ld8(blk);
                        const u32 ltype = b0 & 3, fmt = (b0 >> 2) & 3;
                        if (ltype < 2) {
                            u64 hl, n;
                            if ((fmt & 1) == 0) { hl = 1; n = b0 >> 3; }
                            else if (fmt == 1) { hl = 2; n = (b0 >> 4) | ((u64)ld8(blk + 1) << 4); }
                            else { hl = 3; n = (b0 >> 4) | ((u64)ld8(blk + 1) << 4) | ((u64)ld8(blk + 2) << 12); }
                            used = ltype == 0 ? hl + n : hl + 1;
                            if (n > ZSTD_BLOCK_MAX || used > size) ok = false;
                        } else {
                            if (size < 5) ok = false;
                            else {
                                const u64 v = ld32(blk);
                                u64 hl, csize;
                                if (fmt == 0 || fmt == 1) { hl = 3; csize = (v >> 14) & 0x3FF; }
                                else if (fmt == 2) { hl = 4; csize = v >> 18; }
                                else { hl = 5; csize = (v >> 22) | ((u64)ld8(blk + 4) << 10); }
                                used = hl + csize;
                                if (used > size) ok = false;
                            }
                        }
                    }
                    const u8* p = blk + used;
                    u64 left = size - used;
                    u64 nseq = 0;
                    if (ok && left < 1) ok = false;
                    if (ok) {
                        nseq = ld8(p);
                        if (nseq == 0) { if (left != 1) ok = false; p += 1; left -= 1; }
                        else if (nseq < 128) { p += 1; left -= 1; }
                        else if (nseq < 255) { if (left < 2) ok = false; else { nseq = ((nseq - 128) << 8) + ld8(p + 1); p += 2; left -= 2; } }
                        else { if (left < 3) ok = false; else { nseq = (u64)ld8(p + 1) + ((u64)ld8(p + 2) << 8) + 0x7F00; p += 3; left -= 3; } }
                    }
                    if (ok && nseq > 0) {
                        if (left < 1 || nseq > (u64)(seq_cap - seq_n)) ok = false;
                        u32 modes = 0;
                        if (ok) { modes = ld8(p); p += 1; left -= 1; }
                        // up to 256 bytes of table descriptions, staged in registers (zf_stage_bits)
                        const u32 avail = left < 256 ? (u32)left : 256u;
                        u128 stage; stage.lo = 0; stage.hi = 0;
                        if (ok) stage = zf_load16(p, (i32)avail, 16 * sub);
                        u32 o = 0;                                       // bytes of descriptions consumed
                        int pending = 0, ns[3] = {0, 0, 0};
                        #pragma unroll 1
                        for (int kind = 0; kind < 3 && ok; kind++) {      // T_LL, T_OF, T_ML: the order in the stream
                            int mode = (int)((modes >> (6 - 2 * kind)) & 3);
                            ZPK_LDS u16* const ts = kind == T_LL ? (ZPK_LDS u16*)R->t.c_ll : (kind == T_OF ? (ZPK_LDS u16*)R->t.c_of : (ZPK_LDS u16*)R->t.c_ml);
                            const int max_sym = kind == T_LL ? 35 : (kind == T_OF ? 31 : 52);
                            int al = 0;
                            // where this kind's description is read: the block's own header, or (BLOCKS, Repeat_Mode) the header of the earlier
                            // block whose table it inherits — the host walk knows which (zstd_pj.h)
                            u128 stg = stage; u32 oo = o, av = avail; bool inherited = false;
                            if constexpr (BLOCKS) if (mode == 3) {
                                const u32 toff = kind == T_LL ? tab_ll : (kind == T_OF ? tab_of : tab_ml);
                                mode = (int)((tab_modes >> (2 * kind)) & 3u);
                                if (mode == 3 || (u64)toff >= src_size) { ok = false; mode = 3; }
                                else {
                                    const u64 rem = src_size - toff;
                                    av = rem < 256 ? (u32)rem : 256u;
                                    stg = zf_load16(src + toff, (i32)av, 16 * sub);
                                    oo = 0; inherited = true;
                                }
                            }
                            if (mode == 0) {
                                const ZPK_LDS u16* const dfs = kind == T_LL ? (const ZPK_LDS u16*)sh.d_ll : (kind == T_OF ? (const ZPK_LDS u16*)sh.d_of : (const ZPK_LDS u16*)sh.d_ml);
                                const int n = kind == T_OF ? 32 : 64;
                                for (int i = sub; i < n; i += 16) ts[i] = dfs[i];
                                al = kind == T_OF ? 5 : 6;
                            } else if (mode == 1) {
                                if (oo >= av) ok = false;
                                else {
                                    const u32 s = zf_stage_bits(stg, 8u * oo, lane) & 0xFFu;
                                    if ((int)s > max_sym) ok = false;
                                    else { ts[0] = (u16)(1u | (s << 10)); al = 0; oo += 1; }      // one cell: counter 1 -> no state bits, next state 0
                                }
                            } else if (mode == 2) {
                                int nsym = 0;
                                const int used2 = oo < av ? zf_read_ncount(stg, oo, av - oo, 272u - oo, max_sym, kind == T_OF ? 8 : 9,
                                                                           (ZPK_LDS i16*)R->ncount + zf_nc_base(kind), nsym, al, lane) : -1;
                                if (used2 < 0) ok = false;
                                else { oo += (u32)used2; ns[kind] = nsym; pending |= 1 << kind; }
                            } else {
                                if (BLOCKS || !tables_valid) ok = false;
                                al = kind == T_LL ? al_ll : (kind == T_OF ? al_of : al_ml);
                            }
                            if (!inherited) o = oo;
                            if (kind == T_LL) al_ll = al; else if (kind == T_OF) al_of = al; else al_ml = al;
                        }
                        if (ok) {
                            wave_mem_fence();
                            bool bok = true;
                            const int kd = sub < 3 ? sub : 0;
                            if (sub < 3 && ((pending >> kd) & 1)) {
                                ZPK_LDS u16* const ts = kd == T_LL ? (ZPK_LDS u16*)R->t.c_ll : (kd == T_OF ? (ZPK_LDS u16*)R->t.c_of : (ZPK_LDS u16*)R->t.c_ml);
                                bok = fse_build_inplace(ts, (ZPK_LDS i16*)R->ncount + zf_nc_base(kd), kd == T_LL ? ns[T_LL] : (kd == T_OF ? ns[T_OF] : ns[T_ML]),
                                                        kd == T_LL ? al_ll : (kd == T_OF ? al_of : al_ml));
                            }
                            wave_mem_fence();
                            if (((__ballot(!bok) >> (16 * row)) & 0xFFFFull) != 0) ok = false;
                        }
                        if (ok) {
                            tables_valid = true;
                            // ---- backward bitstream [bs, bs + bs_size) ----
                            bs = p + o; const u64 bsz = left - o;
                            const u32 lastb = bsz ? (u32)ld8(bs + bsz - 1) : 0u;
                            if (lastb == 0) ok = false;
                            else {
                                bs_size = (i32)bsz;
                                pos = (i32)(bsz - 1) * 8 + highbit32(lastb);
                                const i32 kt = (pos - 1) >> 9;                          // chunk of the stream's top bit
                                const u32 c0 = zf_load4(bs, bs_size, (i32)ZF_CHUNK * kt + 4 * sub);
                                const u32 c1 = zf_load4(bs, bs_size, (i32)ZF_CHUNK * (kt - 1) + 4 * sub);
                                pf = zf_load4(bs, bs_size, (i32)ZF_CHUNK * (kt - 2) + 4 * sub);
                                wave_mem_fence();
                                zf_ring_put(ring, kt, sub, c0);
                                zf_ring_put(ring, kt - 1, sub, c1);
                                wave_mem_fence();
                                loaded_lo = kt - 1; refill_below = (kt - 1) * 512 + 160;    // reads reach at most 160 bits below `pos`
                                // initial states: LL, OF, ML
                                pos -= al_ll; const u32 sll = zf_bits((lds_cp8)ring, pos, (u32)al_ll);
                                pos -= al_of; const u32 sof = zf_bits((lds_cp8)ring, pos, (u32)al_of);
                                pos -= al_ml; const u32 sml = zf_bits((lds_cp8)ring, pos, (u32)al_ml);
                                const u32 st0 = role == 0 ? sof : (role == 1 ? sml : (chain ? sll : 0u));
                                cell = tab[st0];
                                my_al = role == 0 ? (u32)al_of : (role == 1 ? (u32)al_ml : (u32)al_ll);
                                al_m31 = my_al - 31u;
                                seq_end = seq_n + (u32)nseq;
                                to_decode = true;
                            }
                        }
                    }
                    if (!ok) { bad = true; ip = iend; phase = ZF_NEED_FRAME; continue; }
                }
                if (to_decode) { phase = ZF_DECODING; break; }
                if (last_block) {
                    if (cksum) { if (iend - ip < 4) { bad = true; ip = iend; } else ip += 4; }
                    phase = ZF_NEED_FRAME;
                } else phase = ZF_NEED_BLOCK;
            }
        }
        if (__ballot(phase != ZF_DONE) == 0) break;

        // =================== rows inside a block: one sequence per trip, until a row's block ends ===================
        // The loop runs under the exec mask of the decoding rows, its only exit test is wave-uniform, and everything a
        // lane needs to know about its role is a VGPR constant (no per-role exec masks: they were SGPR spills).
        if (phase == ZF_DECODING) {
            do {
                const u32 cnt = cell & 1023u, sym = cell >> 10;        // next-state counter, symbol
                const u32 nbs = al_m31 + (u32)__builtin_clz(cnt);     // state lanes: bits of the state update = al - highbit(counter); counter >= 1
                const u32 tv = symt[sym];                             // LL / ML lanes: value baseline | extra bits << 24
                const u32 nv = ((tv >> 24) & ~omask) | (sym & omask); // value lanes: extra bits of the code (an offset code IS its bit count)
                const u32 n = (nv & vmask) | (nbs & smask);           // this lane's field width
                u32 s = n;                                           // inclusive prefix over the row: fields are consumed in lane order
                s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x111, 0xf, 0xf, false);
                s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x112, 0xf, 0xf, false);
                s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xf, false);
                const u32 total = (u32)__builtin_amdgcn_ds_swizzle((int)s, 0xF0);       // lane 7 of the row -> all 16 lanes
                const i32 b = pos - (i32)s;
                const u32 bits = zf_ring_bits((lds_cp8)ring, b) & ((1u << n) - 1u);
                const u32 nst = (cnt << nbs) - (1u << my_al) + bits;  // lanes 5..7: next state (libzstd updates after the last sequence too)
                // lanes 0..3 take the mirrored lane's state (bank 0 of the row), lanes 4..7 keep their own
                const u32 idx = (u32)__builtin_amdgcn_update_dpp((int)nst, (int)nst, 0x141, 0xf, 0x5, false);
                cell = tab[idx & tmask];
                const u32 base = (tv & 0xFFFFFFu & ~omask) | ((1u << (sym & 31u)) & omask);
                const u32 val = base + bits;                         // lanes 0..2: offset value, match length, literal length
                const u32 llv = (u32)__builtin_amdgcn_mov_dpp((int)val, 0xE6, 0xf, 0xf, true);            // quad_perm [2,1,2,3]
                const u32 mlv = (u32)__builtin_amdgcn_mov_dpp((int)val, 0xE5, 0xf, 0xf, true);            // quad_perm [1,1,2,3]
                u32 offset;
                {   // repeat offsets (RFC 8878 3.1.1.5), meaningful in lane 0 of the row
                    const bool big = val > 3;
                    const u32 ridx = val - 1 + (llv == 0 ? 1u : 0u);
                    u32 tt = ridx == 3 ? rep0 - 1 : (ridx == 1 ? rep1 : rep2);
                    if (tt == 0) tt = 1;
                    offset = big ? val - 3 : (ridx == 0 ? rep0 : tt);
                    const bool shift = big || ridx != 0;
                    const u32 n2 = (big || ridx != 1) ? rep1 : rep2;
                    if (shift) { rep2 = n2; rep1 = rep0; rep0 = offset; }
                }
                // what does not fit the packing (or is corrupt) sends the entry to the fused decoder at the end of the block
                // (an offset is >= 1 by construction and repeat offsets were checked when they were new; match and literal lengths cannot
                // reach their 18 / 17 bits: codes <= 52 / 35 give at most 131 074 / 131 071)
                badv |= val >> (BLOCKS ? ZF_SYM_SHIFT : ZF_SEQ_OFF_BITS);   // (offset values of 2^29 .. 2^29 + 2 would still fit: they go to the fused decoder too; BLOCKS: real offsets stay below the symbols)
                {   // The row keeps its last sequences in REGISTERS, as a shift register along its lanes: every step the packed values move
                    // one lane up (DPP row_shr:1) and lane 0 — which has no lane below it and therefore keeps the `old` operand — takes
                    // the new one.  After eight steps lane j holds sequence (seq_n - 1 - j): no LDS, no exec mask, two instructions.
                    const u64 packed = (u64)offset | ((u64)mlv << ZF_SEQ_OFF_BITS) | ((u64)llv << (ZF_SEQ_OFF_BITS + ZF_SEQ_ML_BITS));
                    acc_lo = (u32)__builtin_amdgcn_update_dpp((int)(u32)packed, (int)acc_lo, 0x111, 0xf, 0xf, false);
                    acc_hi = (u32)__builtin_amdgcn_update_dpp((int)(u32)(packed >> 32), (int)acc_hi, 0x111, 0xf, 0xf, false);
                }
                pos -= (i32)total;
                seq_n += 1;
                if ((seq_n & (ZF_SEQBUF - 1u)) == 0) {               // 8 sequences = one 64-byte store
                    if ((u32)sub < ZF_SEQBUF) arena[a_base + (seq_n - 1u) - (u32)sub] = ((u64)acc_hi << 32) | acc_lo;
                }
                if (pos < refill_below) {                          // the next reads reach below the ring: bring in the prefetched chunk
                    wave_mem_fence();
                    zf_ring_put(ring, loaded_lo - 1, sub, pf);
                    wave_mem_fence();
                    loaded_lo -= 1; refill_below -= 512;
                    pf = zf_load4(bs, bs_size, (i32)ZF_CHUNK * (loaded_lo - 1) + 4 * sub);
                }
            } while (__ballot(seq_n == seq_end) == 0);
            if (seq_n == seq_end) {                                  // this row's block is done
                // lane 0 of the row saw every sequence; libzstd 1.4.9: the stream must not be under-consumed
                const u64 bm = __ballot(sub == 0 && badv != 0);
                if (((bm >> (16 * row)) & 1ull) != 0 || pos > 0) bad = true;
                if (bad) { ip = iend; phase = ZF_NEED_FRAME; }
                else if (last_block) {
                    if (cksum) { if (iend - ip < 4) { bad = true; ip = iend; } else ip += 4; }
                    phase = ZF_NEED_FRAME;
                } else phase = ZF_NEED_BLOCK;
            }
        }
    }
}

__global__ __launch_bounds__(64, 3) void k_zstd_fse(const u8* __restrict__ src, const zpk_decode_desc* __restrict__ desc,
                                                 const u32* __restrict__ list, u32* __restrict__ counters,
                                                 u64* __restrict__ arena, u32* __restrict__ state)
{
    if (counters[ZF_COUNT_WORD] == 0) return;        // no Zstandard entry in the batch
    __shared__ ZfShared sh;
    zstd_fse_rows<false>(sh, src, desc, list, counters, arena, state, nullptr, 0);
}
__global__ __launch_bounds__(64, 3) void k_zstd_fse_blocks(const u8* __restrict__ src, const zpk_decode_desc* __restrict__ desc,
                                                        const u32* __restrict__ list, u32* __restrict__ counters,
                                                        u64* __restrict__ arena, u32* __restrict__ state, u32* __restrict__ rep_out, u64 src_size)
{
    if (counters[ZF_COUNT_WORD] == 0) return;
    __shared__ ZfShared sh;
    zstd_fse_rows<true>(sh, src, desc, list, counters, arena, state, rep_out, src_size);
}

}  // namespace zpk
