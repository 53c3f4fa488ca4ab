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
// groups [g_first, ngroups) of 64 blocks (a caller whose span becomes final piece by piece launches a range of groups per piece)
__global__ __launch_bounds__(256) void k_xxh3_partials(const u8* __restrict__ src, const zpk_span* __restrict__ spans, u32 nspans,
                                                       u64 g_first, u64 ngroups, u64* __restrict__ partial)
{
    const int lane = lane_id();
    const u64 g = uni64(g_first + (u64)blockIdx.x * 4 + (threadIdx.x >> 6));
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
    u64* out = partial + (sp.part_base + b0) * 8 + 128 * (lane & 3);                  // a group's sums lie [accumulator][block]: the chain's lane reads ITS 64 sums as 512 contiguous bytes
    for (u32 j = 0; j < nb; j += 4) {                                                 // four blocks' loads in flight (unconditional: past the end the last block again)
        u128 d[4];
        #pragma unroll
        for (u32 t = 0; t < 4; t++) d[t] = ld128(q + ((u64)(j + t < nb ? j + t : nb - 1) << 10));
        #pragma unroll
        for (u32 t = 0; t < 4; t++) {
            u64 c0, c1;
            Xxh3Wave::slot(d[t].lo, d[t].hi, w.k0, w.k1, c0, c1);
            Xxh3Wave::reduce16<true>(c0, c1);
            if (lane < 4 && j + t < nb) { out[j + t] = c0; out[64 + j + t] = c1; }
        }
    }
}
// One step of the chain: (lo, hi) = the accumulator with its block's sum already added; scrambled, and the NEXT block's sum added (the
// compiler folds that addition into the multiply-add).  Three formulations of this step — 7 to 9 instructions, one or two dependent
// multiplies — differ by 7 % (profiles/r05/r05_xxh3_chain_variants.txt): this is the shortest.
__device__ __forceinline__ void chain_step(u32& lo, u32& hi, u64 sum_next, u64 key)
{
    u64 acc = ((u64)hi << 32) | lo;
    acc = ((acc ^ (acc >> 47)) ^ key) * ZPK_P32_1 + sum_next;
    lo = (u32)acc; hi = (u32)(acc >> 32);
}
// One wave per span.  The chain may be run in SECTIONS (a span that becomes final piece by piece): blocks [b_from, b_to) of the span,
// the eight accumulators carried in state[8 * span ..] between the launches; the launch with last != 0 finishes the span (every block
// still open, the tail, the avalanche).  state == nullptr: the whole span at once.
__global__ __launch_bounds__(64) void k_xxh3_chain(const u8* __restrict__ src, const zpk_span* __restrict__ spans, const u64* __restrict__ partial,
                                                   u64* __restrict__ hash_out, u64* __restrict__ state, u64 b_from, u64 b_to, int last)
{
    const int lane = lane_id(), q = lane & 3, j = lane >> 2;
    const zpk_span sp = spans[blockIdx.x];
    const u8* p = uni_ptr(src + sp.off);
    const u64 len = uni64(sp.len), nblocks = (len - 1) >> 10;
    const u64 bend = (state == nullptr || last || b_to > nblocks) ? nblocks : b_to;
    Xxh3Wave w; w.init(lane);
    // (round 5) The chain is ONE wave's instruction stream, so what it costs is instructions per block.  Lane l carries accumulator
    // l & 7 alone (Xxh3Wave's pair per lane would be twice the instructions) and reads its sum itself — the eight lanes of a group read
    // one 64-byte line per block, sixteen blocks per trip, the next sixteen in flight meanwhile: no cross-lane traffic.  Before, a
    // block's sums came through four ds_bpermute in front of two scrambles per lane: ~330 cycles per block, 36 ms for the 262 144
    // blocks of a 256 MiB span — as long as the block-parallel decode of the entry in front of it.
    (void)q; (void)j;
    const int ai = lane & 7;
    const u64 init[8] = { ZPK_P32_3, ZPK_P64_1, ZPK_P64_2, ZPK_P64_3, ZPK_P64_4, ZPK_P32_2, ZPK_P64_5, ZPK_P32_1 };
    u64 acc = init[0];
    #pragma unroll
    for (int t = 1; t < 8; t++) acc = ai == t ? init[t] : acc;
    u64 b = 0;
    if (state != nullptr && b_from != 0) { acc = state[8 * (u64)blockIdx.x + ai]; b = uni64(b_from); }
    const u64 sk = sec64(128 + 8 * ai);
    // The sums of a group of 64 blocks lie [accumulator][block] (k_xxh3_partials): a lane's 64 sums are 32 loads of 16 bytes, all in
    // flight while the group before is chained.  (With 16 sums of 8 bytes per trip the wave waited ~0.9 us of load latency per trip:
    // 59 ns per block whatever the arithmetic; now 46 ns, 12.1 ms per 256 MiB.)
    const u8* const part = (const u8*)(partial + sp.part_base * 8 + 64 * ai);          // + 4096 bytes per group
    constexpr u32 NL = XS_GROUP / 2;
    u128 cur[NL], nxt[NL];
    const u64 g_end = (bend + XS_GROUP - 1) / XS_GROUP;                                 // groups [b / 64, g_end); the last may be short
    u64 g = b / XS_GROUP;
    if (g < g_end) {
        #pragma unroll
        for (u32 t = 0; t < NL; t++) cur[t] = ld128(part + g * 4096 + 16 * t);
    } else {
        #pragma unroll
        for (u32 t = 0; t < NL; t++) cur[t] = u128{0, 0};
    }
    acc += b < bend ? cur[0].lo : 0ull;                                                 // (a step adds the sum of the block BEHIND its own)
    u32 lo = (u32)acc, hi = (u32)(acc >> 32);
    for (; g < g_end; g++) {
        const bool more = g + 1 < g_end;
        #pragma unroll
        for (u32 t = 0; t < NL; t++) nxt[t] = more ? ld128(part + (g + 1) * 4096 + 16 * t) : u128{0, 0};
        const u64 left = bend - g * XS_GROUP;                                           // blocks of this group that exist (>= 1)
        if (left >= XS_GROUP) {
            #pragma unroll
            for (u32 t = 0; t < NL; t++) {
                chain_step(lo, hi, cur[t].hi, sk);
                chain_step(lo, hi, t + 1 < NL ? cur[t + 1].lo : (left > XS_GROUP ? nxt[0].lo : 0ull), sk);
            }
        } else {
            #pragma unroll
            for (u32 t = 0; t < NL; t++) {
                if (2 * t < left) chain_step(lo, hi, 2 * t + 1 < left ? cur[t].hi : 0ull, sk);
                if (2 * t + 1 < left) chain_step(lo, hi, (t + 1 < NL && 2 * t + 2 < left) ? cur[t + 1].lo : 0ull, sk);
            }
        }
        #pragma unroll
        for (u32 t = 0; t < NL; t++) cur[t] = nxt[t];
    }
    acc = ((u64)hi << 32) | lo;
    if (state != nullptr && !last) { if (lane < 8) state[8 * (u64)blockIdx.x + lane] = acc; return; }
    w.a0 = shfl64(acc, 2 * q); w.a1 = shfl64(acc, 2 * q + 1);                         // back to the pair per lane the tail works on
    const u32 nstripes = (u32)(((len - 1) - (nblocks << 10)) >> 6);
    const u64 h = uni64(w.finish(p + (nblocks << 10), nstripes, p + len, len, lane));
    lane0_guard();
    if (lane == 0) hash_out[blockIdx.x] = h;
}

}  // namespace zpk
