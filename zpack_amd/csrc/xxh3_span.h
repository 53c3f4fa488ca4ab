// XXH3-64 of long spans by the whole chip: per-block partial sums side by side, then one short chain per span.
// Used where ONE entry is large: the split entries of the host write path (zpk_encode.inc) and the frame-parallel decode of entries
// that are sequences of frames (zpk_codec.hip, decode_big_entries).  Replaces the serial XXH3_64bits over the whole buffer of
// lib/zpack_write.c:256 / lib/zpack_read.c:466 for such entries.
#pragma once
#include "zpk_device.h"
#include "xxh3_device.h"

namespace zpk {

// A 1 KiB block enters the state as  acc = scramble(acc + S_b)  where S_b — the sum of the block's 16 stripe products — does not
// depend on the state: the S_b of all blocks are computed side by side (k_xxh3_partials, 64 bytes per block), the chain over them is
// 8 additions + scrambles per block for ONE wave (k_xxh3_chain): 256 MiB in a few ms instead of ~190 ms of one wave's load latency.
__device__ __forceinline__ u64 shfl64(u64 v, int from)
{
    const u32 lo = (u32)__shfl((int)(u32)v, from, 64), hi = (u32)__shfl((int)(u32)(v >> 32), from, 64);
    return ((u64)hi << 32) | lo;
}
struct zpk_span { u64 off, len, part_base; };    // part_base: index of the span's first block among the partial sums (a multiple of 64)
#define XS_GROUP 64u                             // blocks per wave of the partial pass
__global__ __launch_bounds__(256) void k_xxh3_partials(const u8* __restrict__ src, const zpk_span* __restrict__ spans, u32 nspans,
                                                       u64 ngroups, u64* __restrict__ partial)
{
    const int lane = lane_id();
    const u64 g = uni64((u64)blockIdx.x * 4 + (threadIdx.x >> 6));
    if (g >= ngroups) return;
    u32 lo = 0, hi = nspans;                                                          // last span with part_base / 64 <= g
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (spans[mid].part_base / XS_GROUP <= g) lo = mid; else hi = mid; }
    const zpk_span sp = spans[lo];
    const u64 nblocks = (sp.len - 1) >> 10;                                           // the block with the last byte belongs to the chain
    const u64 b0 = (g - sp.part_base / XS_GROUP) * XS_GROUP;
    if (b0 >= nblocks) return;
    const u32 nb = (u32)(nblocks - b0 < XS_GROUP ? nblocks - b0 : XS_GROUP);
    Xxh3Wave w; w.init(lane);
    const u8* q = src + sp.off + (b0 << 10) + 16 * lane;
    u64* out = partial + (sp.part_base + b0) * 8 + 2 * (lane & 3);
    for (u32 j = 0; j < nb; j += 4) {                                                 // four blocks' loads in flight (unconditional: past the end the last block again)
        u128 d[4];
        #pragma unroll
        for (u32 t = 0; t < 4; t++) d[t] = ld128(q + ((u64)(j + t < nb ? j + t : nb - 1) << 10));
        #pragma unroll
        for (u32 t = 0; t < 4; t++) {
            u64 c0, c1;
            Xxh3Wave::slot(d[t].lo, d[t].hi, w.k0, w.k1, c0, c1);
            Xxh3Wave::reduce16<true>(c0, c1);
            if (lane < 4 && j + t < nb) { out[(u64)(j + t) * 8] = c0; out[(u64)(j + t) * 8 + 1] = c1; }
        }
    }
}
__global__ __launch_bounds__(64) void k_xxh3_chain(const u8* __restrict__ src, const zpk_span* __restrict__ spans, const u64* __restrict__ partial,
                                                   u64* __restrict__ hash_out)
{
    const int lane = lane_id(), q = lane & 3, j = lane >> 2;
    const zpk_span sp = spans[blockIdx.x];
    const u8* p = uni_ptr(src + sp.off);
    const u64 len = uni64(sp.len), nblocks = (len - 1) >> 10;
    Xxh3Wave w; w.init(lane);
    const u64* part = partial + sp.part_base * 8 + 2 * q;
    // sixteen blocks' sums per load (lane group j holds block b + j), the next sixteen in flight meanwhile
    #define XS_LD(b) ((b) + (u64)j < nblocks ? ld128((const u8*)(part + ((b) + (u64)j) * 8)) : u128{0, 0})
    u128 cur = XS_LD(0);
    for (u64 b = 0; b < nblocks; b += 16) {
        const u128 nxt = XS_LD(b + 16);
        const u32 m = (u32)(nblocks - b < 16 ? nblocks - b : 16);
        for (u32 k = 0; k < m; k++) {
            const int from = (int)(4 * k) + q;
            w.a0 += shfl64(cur.lo, from); w.a1 += shfl64(cur.hi, from);
            w.scramble();
        }
        cur = nxt;
    }
    #undef XS_LD
    const u32 nstripes = (u32)(((len - 1) - (nblocks << 10)) >> 6);
    const u64 h = uni64(w.finish(p + (nblocks << 10), nstripes, p + len, len, lane));
    lane0_guard();
    if (lane == 0) hash_out[blockIdx.x] = h;
}

}  // namespace zpk
