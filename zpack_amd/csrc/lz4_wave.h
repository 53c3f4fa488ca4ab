// lz4_wave.h — LZ4 frame + block decode, generic path: ONE WAVE PER ENTRY, output window in global
// memory (L2-resident while hot).  Handles the whole LZ4 Frame format (any block size, linked or
// independent blocks, stored blocks, optional checksums, skippable frames) and is the fallback for
// frames the LDS-window fast path (lz4_lds.h) does not take.
//
// Replaces the LZ4F_decompress loop of the reference (lib/zpack_read.c:414-439).
//
// Parsing is wave-uniform: the compressed stream is held as a 256-byte register window (one dword per
// lane, one coalesced load per refill) and token / length / offset bytes are pulled out of it with
// v_readlane, so the serial token chain never waits on memory.  Literal and match copies are done by
// all 64 lanes (byte per lane, 64 bytes per instruction); overlapping matches (offset < length) read
// the period [op-offset, op) with a modulo so that no lane depends on a byte written by the same copy.
#pragma once
#include "zpk_device.h"
#include "xxh3_device.h"

namespace zpk {

// 256-byte sliding register window over a read-only byte stream
struct ByteWindow {
    u32 w;              // lane l holds bytes [base + 4l, base + 4l + 4)
    const u8* base;     // 4-byte aligned, uniform
    const u8* lo;       // readable range of the underlying buffer (uniform)
    const u8* hi;

    __device__ __forceinline__ void load(const u8* p, int lane)
    {
        base = (const u8*)((u64)p & ~(u64)3);
        const u8* a = base + 4 * lane;
        u32 v = 0;
        if (a >= lo && a + 4 <= hi) v = *(const u32*)a;
        else {
            #pragma unroll
            for (int i = 0; i < 4; i++) if (a + i >= lo && a + i < hi) v |= (u32)a[i] << (8 * i);
        }
        w = v;
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

// one LZ4 block: src [ip, iend) -> dst [op, ...), cap = oend; hist_lo = lowest output address a match may reach
__device__ inline int lz4_block_wave(ByteWindow& win, Watchdog& wd, const u8* ip, const u8* iend, u8* dst_lo, u8*& op_io, u8* oend, int lane)
{
    u8* op = op_io;
    if (ip >= iend) return D_MALFORMED;
    for (;;) {
        if (ip >= iend || wd.expired()) return D_MALFORMED;
        u32 tok = win.byte(ip++, lane);
        u64 lit = tok >> 4;
        if (lit == 15) {
            u32 b;
            do {
                if (ip >= iend) return D_MALFORMED;
                b = win.byte(ip++, lane);
                lit += b;
            } while (b == 255);
        }
        if (lit > (u64)(iend - ip)) return D_MALFORMED;
        if (lit > (u64)(oend - op)) { op_io = op; return D_DST_FULL; }
        for (u64 i = lane; i < lit; i += WAVE) op[i] = ip[i];
        ip += lit; op += lit;
        if (ip == iend) break;
        if (iend - ip < 2) return D_MALFORMED;
        u32 off = win.byte(ip, lane) | (win.byte(ip + 1, lane) << 8);
        ip += 2;
        if (off == 0 || (u64)off > (u64)(op - dst_lo)) return D_MALFORMED;
        u64 ml = tok & 15;
        if (ml == 15) {
            u32 b;
            do {
                if (ip >= iend) return D_MALFORMED;
                b = win.byte(ip++, lane);
                ml += b;
            } while (b == 255);
        }
        ml += 4;
        if (ml > (u64)(oend - op)) { op_io = op; return D_DST_FULL; }
        wave_mem_fence();                                  // literals above are match sources
        const u8* m = op - off;
        if ((u64)off >= ml) {
            for (u64 i = lane; i < ml; i += WAVE) op[i] = m[i];
        } else {
            for (u64 i = lane; i < ml; i += WAVE) op[i] = m[(u32)i % off];
        }
        wave_mem_fence();
        op += ml;
    }
    op_io = op;
    return D_OK;
}

// whole frame.  src_lo/src_hi bound what may be READ (the archive image); all values uniform.
__device__ inline DecodeOut lz4f_decode_wave(Watchdog& wd, const u8* src, u64 src_size, const u8* src_lo, const u8* src_hi,
                                             u8* dst, u64 dst_cap, int lane)
{
    DecodeOut r; r.rc = D_OK; r.produced = 0;
    const u8* ip = src;
    const u8* iend = src + src_size;
    u8* op = dst;
    u8* oend = dst + dst_cap;
    ByteWindow win; win.lo = src_lo; win.hi = src_hi;
    win.load(ip, lane);

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
                for (u64 i = lane; i < n; i += WAVE) op[i] = ip[i];
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
            for (u64 i = lane; i < n; i += WAVE) op[i] = ip[i];
            op += n;
            if (full) { r.rc = D_DST_FULL; r.produced = (u64)(op - dst); return r; }
        } else {
            u8* hist_lo = indep ? op : frame_out;
            if ((u64)(op - hist_lo) > 65536) hist_lo = op - 65536;
            u8* bend = oend;
            bool limited = false;
            if ((u64)(oend - op) > bmax) { bend = op + bmax; limited = true; }
            int rc = lz4_block_wave(win, wd, ip, ip + bsz, hist_lo, op, bend, lane);
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
