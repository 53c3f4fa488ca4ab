// zstd_ring.h — the Zstandard execute stage through the LDS output ring of lx_ring.h.
//
// What.  k_zstd_fse (zstd_fse4.h) has left an entry's sequences, packed, in the arena; what remains (lib/zpack_read.c:376-411:
// ZSTD_decompressStream until the frame ends, then the XXH3 of the result, :466) is literals + execution + hash.  The first
// execute stage (zstd_sequences_pre + seq_exec_batch) wrote every sequence straight to HBM with exact-tail stores and
// gathered every match from HBM: 4.3x the algorithmic read traffic and 2x the write traffic (profiles/r02), and the hash
// re-read the whole output afterwards.  Here the output goes through the same 4 KiB LDS ring the LZ4 executor uses:
// 64 sequences per step land in the ring with aligned LDS accesses, matches that reach less than ~1 KiB back (most, on
// text) are served from LDS, the ring is flushed in whole 1 KiB lines that the XXH3 accumulators consume on the way out.
//
// Exactness.  Same contract as k_lz4_exec: anything this path does not take (frame checksums, an inconsistency of any
// kind) leaves the entry to the general decoder k_zstd, whose verdict is the reference's.
#pragma once
#include "lx_ring.h"
#include "zstd_wg.h"

namespace zpk {

// LDS of the ring executor's workgroup: the Huffman decode table and just the construction scratch the literals section needs (one
// FSE table for the weights — the three sequence tables of ZstdShared are k_zstd_fse's business), then the output ring.
// 9 776 B: 16 workgroups per CU, like the direct executor.
struct alignas(16) ZstdRingShared {
    u8  huf[4096];
    u32 huf_rank[16];
    union {
        struct { FseCell wt[64]; i16 ncount[1][64]; u8 spread[1][512]; u16 nextc[1][64]; };
        struct { u16 sym_start[256]; u32 huf_cnt[16]; };
    };
    u8  weights[256];
    u32 huf_max_bits, huf_valid, pad0_, pad1_;
    u8  ring[LX_RING + 32];
    u8  secret[192];                             // XXH3 secret for the flush (xxh3_device.h, Xxh3Lite::block)
};
#define ZSTD_RING_SHARED_BYTES sizeof(ZstdRingShared)

// the sequences of one block, 64 at a time: packed (zstd_fse4.h: offset 29 bits | match length 18 | literal length 17)
__device__ inline int zr_sequences(LxOut& O, const u64* pre, u64 nseq, const ZLiterals& zl, const u8* lit_hi, u32 hist_lo, u64 dst_cap,
                                   Watchdog& wd, u64& lit_pos_out, int lane)
{
    u64 lit_pos = 0, base = 0;
    u64 nxt = 0, nxt_base = 0;
    if ((u64)lane < nseq) nxt = *(const ZPK_GLOBAL u64*)(pre + (u64)lane);
    SeqStats stt = {};
    (void)stt;
    while (base < nseq) {
        if (__builtin_amdgcn_s_memrealtime() > wd.deadline) { wd.fired = true; return LX_E_FRAME; }
        u32 cnt = (u32)(nseq - base < WAVE ? nseq - base : WAVE);
        u64 v = 0;
        if (nxt_base == base) v = nxt;
        else if (base + (u64)lane < nseq) v = *(const ZPK_GLOBAL u64*)(pre + base + (u64)lane);
        nxt_base = base + WAVE;                              // one batch ahead, for when this one is taken whole
        nxt = 0;
        if (nxt_base + (u64)lane < nseq) nxt = *(const ZPK_GLOBAL u64*)(pre + nxt_base + (u64)lane);
        u32 ll = 0, ml = 0, off = 1;
        if ((u32)lane < cnt) { off = (u32)v & ((1u << 29) - 1u); ml = (u32)(v >> 29) & ((1u << 18) - 1u); ll = (u32)(v >> 47); }
        if (__ballot(off == 0) != 0) return LX_E_OFFSET;
        const u64 longm = __ballot(ll > LX_MAX_LL || ml > LX_MAX_ML);
        if (longm & 1) {
            // a long sequence goes on its own, piece by piece
            const u32 ll0 = (u32)__builtin_amdgcn_readfirstlane((int)ll), ml0 = (u32)__builtin_amdgcn_readfirstlane((int)ml);
            const u32 off0 = (u32)__builtin_amdgcn_readfirstlane((int)off);
            if (ll0 > zl.lit_size - lit_pos) return LX_E_FRAME;
            int rc = zl.rle ? lx_append_fill(O, zl.rle_byte, ll0, dst_cap, lane) : lx_append_raw(O, zl.lit + lit_pos, ll0, lit_hi, dst_cap, lane);
            if (rc != LX_OK) return rc;
            lit_pos += ll0;
            rc = lx_append_match(O, off0, ml0, hist_lo, dst_cap, lane);
            if (rc != LX_OK) return rc;
            base += 1;
            continue;
        }
        if (longm) cnt = (u32)__ffsll((long long)longm) - 1u;                  // the sequences before it, as a batch
        const u32 llx = (u32)lane < cnt ? ll : 0u;
        const u32 xl = wave_scan_add(llx);
        if ((u64)(u32)__builtin_amdgcn_readlane((int)xl, (int)cnt - 1) > zl.lit_size - lit_pos) return LX_E_FRAME;
        const u64 my_lit = lit_pos + (xl - llx);
        int rc;
        LXT(0);
        if (zl.rle) { LxLitFill F; F.byte = zl.rle_byte; rc = lx_exec_batch(O, cnt, ll, ml, off, F, hist_lo, dst_cap, lane, stt); }
        else { LxLitGlobal G; G.p = zl.lit + my_lit; G.rd_hi = lit_hi; rc = lx_exec_batch(O, cnt, ll, ml, off, G, hist_lo, dst_cap, lane, stt); }
        if (rc != LX_OK) return rc;
        LXT(8);
        lit_pos += (u32)__builtin_amdgcn_readlane((int)xl, (int)cnt - 1);       // (the batch may have been cut to what fits the ring)
        base += cnt;
    }
    lit_pos_out = lit_pos;
    return LX_OK;
}

// one compressed block (zstd_block<true> with the ring as its output)
__device__ inline int zr_block(ZstdRingShared& sh, ZFrameState& fs, LxOut& O, const u8* src, u64 size, const u8* rd_hi, u32 hist_lo, u64 dst_cap,
                               u8* lit_buf, int lane)
{
    if (size < 3) return LX_E_FRAME;
    ZLiterals zl;
    LXT(9);
    if (zstd_literals(sh, fs, src, size, rd_hi, lit_buf, zl, lane) != D_OK) return LX_E_FRAME;
    LXT(1);
    const u8* const lit_hi = zl.lit == lit_buf ? lit_buf + ZSTD_LIT_SCRATCH : rd_hi;
    const u8* p = src + zl.used;
    u64 left = size - zl.used;
    if (left < 1) return LX_E_FRAME;
    u64 nseq = uld8(p);
    if (nseq == 0) { if (left != 1) return LX_E_FRAME; }
    else if (nseq < 128) { left -= 1; }
    else if (nseq < 255) { if (left < 2) return LX_E_FRAME; nseq = ((nseq - 128) << 8) + uld8(p + 1); left -= 2; }
    else { if (left < 3) return LX_E_FRAME; nseq = (u64)uld8(p + 1) + ((u64)uld8(p + 2) << 8) + 0x7F00; left -= 3; }
    const u32 block_out = O.wp;
    u64 lit_pos = 0;
#ifdef ZR_ABL_NOSEQ         // developer ablation (instruction counters only; the output is wrong): literals alone
    if (nseq > 0) { fs.pre_idx += nseq; return LX_OK; }
#endif
    if (nseq > 0) {
        if (left < 1) return LX_E_FRAME;
        const u64* const pre = fs.pre + fs.pre_idx;          // k_zstd_fse validated the tables and the bitstream of this block and decoded it
        fs.pre_idx += nseq;
        const int rc = zr_sequences(O, pre, nseq, zl, lit_hi, hist_lo, dst_cap, *fs.wd, lit_pos, lane);
        if (rc != LX_OK) return rc;
    }
    const u64 rest = zl.lit_size - lit_pos;
    const int rc = zl.rle ? lx_append_fill(O, zl.rle_byte, rest, dst_cap, lane) : lx_append_raw(O, zl.lit + lit_pos, rest, lit_hi, dst_cap, lane);
    if (rc != LX_OK) return rc;
    if (O.wp - block_out > ZSTD_BLOCK_MAX) return LX_E_BLOCKMAX;
    return LX_OK;
}

// every frame of an entry; the hash of the output comes out with it
__device__ inline LxResult zstd_ring_decode_wave(ZstdRingShared& sh, Watchdog& wd, const u8* src, u64 src_size, u8* dst, u64 dst_cap, u64 uncomp_size,
                                                 u8* lit_buf, const u64* pre, int lane, u64* lx_dbg = nullptr)
{
    (void)lx_dbg;
    LxResult R; R.rc = LX_E_FRAME; R.produced = 0; R.hash = 0;
    if (dst_cap >= (1ull << 31) || uncomp_size >= (1ull << 31)) return R;       // positions are 32-bit here
    const u8* ip = src; const u8* const iend = src + src_size;
    LxOut O;
    lx_begin(O, to_lds_rw(sh.ring), dst, uncomp_size, lane, to_lds_rw(sh.secret));
#ifdef LX_STATS
    for (int k = 0; k < 12; k++) O.tm[k] = 0;
    O.t_last = __builtin_amdgcn_s_memtime();
    const u64 t_begin = O.t_last;
#endif
    u64 pre_idx = 0;
    __syncthreads();
    while (ip < iend) {
        if (wd.expired()) return R;
        if (iend - ip < 4) return R;
        const u32 magic = uld32(ip);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (iend - ip < 8) return R;
            const u64 sz = uld32(ip + 4);
            if ((u64)(iend - ip) - 8 < sz) return R;
            ip += 8 + sz;
            continue;
        }
        if (magic != 0xFD2FB528u) return R;
        // ---- frame header (as zstd_decode_wave) ----
        if (iend - ip < 6) return R;
        ip += 4;
        const u32 fhd = uld8(ip++);
        const u32 fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, cksum = (fhd >> 2) & 1, did_flag = fhd & 3;
        if ((fhd & 0x08) || cksum) return R;                 // frames with a content checksum: the general decoder's
        if (!single) {
            if (iend - ip < 1) return R;
            const u32 wdesc = uld8(ip++);
            if (10 + (wdesc >> 3) > 31) return R;
        }
        const u32 dn = did_flag == 3 ? 4 : did_flag;
        if ((u64)(iend - ip) < dn) return R;
        u32 dict_id = 0;
        for (u32 i = 0; i < dn; i++) dict_id |= uld8(ip + i) << (8 * i);
        ip += dn;
        if (dict_id != 0) return R;
        const u32 fn = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
        if ((u64)(iend - ip) < fn) return R;
        u64 fcs = 0;
        for (u32 i = 0; i < fn; i++) fcs |= (u64)uld8(ip + i) << (8 * i);
        if (fn == 2) fcs += 256;
        ip += fn;

        ZFrameState fs;
        fs.wd = &wd; fs.zs = nullptr;
        fs.rep0 = 1; fs.rep1 = 4; fs.rep2 = 8; fs.seq_tables_valid = false; fs.al_ll = fs.al_of = fs.al_ml = 0;
        fs.pre = pre; fs.pre_idx = pre_idx;
        lane0_guard();
        if (lane == 0) sh.huf_valid = 0;
        __syncthreads();
        const u32 frame_lo = O.wp;
        for (;;) {
            if (wd.expired()) return R;
            if (iend - ip < 3) return R;
            const u32 bh = uld8(ip) | (uld8(ip + 1) << 8) | (uld8(ip + 2) << 16);
            ip += 3;
            const bool last = bh & 1; const u32 type = (bh >> 1) & 3; const u64 bsize = bh >> 3;
            int rc;
            if (type == 3) return R;
            if (type == 0) {
                if (bsize > (u64)(iend - ip)) return R;
                rc = lx_append_raw(O, ip, bsize, iend, dst_cap, lane);
                ip += bsize;
            } else if (type == 1) {
                if (iend - ip < 1) return R;
                rc = lx_append_fill(O, uld8(ip), bsize, dst_cap, lane);
                ip += 1;
            } else {
                if (bsize > (u64)(iend - ip) || bsize >= ZSTD_BLOCK_MAX) return R;
                rc = zr_block(sh, fs, O, ip, bsize, iend, frame_lo, dst_cap, lit_buf, lane);
                ip += bsize;
            }
            if (rc != LX_OK) { R.rc = rc; return R; }
            wave_mem_fence();
            if (last) break;
        }
        pre_idx = fs.pre_idx;
        if (fn != 0 && (u64)(O.wp - frame_lo) != fcs) return R;
    }
    LXT(9);
    lx_finish(O, dst, uncomp_size, R, lane);
    LXT(10);
#ifdef LX_STATS
    if (lx_dbg && lane == 0) { for (int k = 0; k < 12; k++) lx_dbg[k] = O.tm[k]; lx_dbg[12] = __builtin_amdgcn_s_memtime() - t_begin; }
#endif
    R.rc = LX_OK;
    return R;
}

}  // namespace zpk
