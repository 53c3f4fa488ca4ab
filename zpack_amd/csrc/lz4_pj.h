// lz4_pj.h — ONE LARGE LZ4 FRAME decoded by the whole chip: blocks parsed side by side, every output byte resolved to the literal it
// is a copy of by POINTER DOUBLING.
//
// Replaces the LZ4F_decompress loop of lib/zpack_read.c:414-439 for the entries the reference writer produces above a few MiB: ONE
// frame (lib/zpack_write.c:204-210: Begin, one Update, End) of linked 64 KiB blocks.  Its blocks depend on each other — a match may
// reach 64 KiB back into the previous block's output — so the one-wave decoder takes them in order: 0.1 GiB/s, forty times below one
// CPU core (profiles/r04/r04_big_entry_rate.txt).  What does NOT depend on anything is the TOKEN STREAM of a block, and what a match
// copies is, in the end, literal bytes:
//   1. k_pj_parse   one wave per compressed block: the lane-parallel parse of lz4_wave.h, nothing executed — per sequence an 8-byte
//                   record (output position in the block, literal length, literal position, offset), a bit mask of the output
//                   positions sequences start at, the block's output size;
//   2. k_pj_scan    the blocks' output offsets (a block decodes to 64 KiB only by convention);
//   3. k_pj_init    one workgroup per block: every output byte finds its sequence (rank in the bit mask) and becomes a 32-bit
//                   REFERENCE: "literal byte at offset x of the compressed entry" (final) or "the output byte at position y < mine";
//   4. k_pj_jump    S[i] = S[S[i]] for every byte that still refers to an output byte: the chain of copies a byte hangs on halves
//                   per round — 11 rounds for 4 MiB of text, byte runs or records (tools/sim/lz4_frame_parallel_sim.py), at most
//                   log2(size) ever; no dependency analysis, no levels, no order;
//   5. k_pj_gather  every byte is a literal reference now: the output is one gather from the compressed entry.
// Memory: 4 bytes of scratch per output byte.  Verdicts: this path finishes an entry only when everything about it was regular (the
// host then compares the XXH3 of the assembled output, xxh3_span.h); any irregularity — a token the parser rejects, an offset that
// reaches in front of the frame, a block that decodes to more than 64 KiB, sizes that do not add up — sets the error word and the
// entry is decoded from scratch by the one-wave decoder, which alone gives verdicts.
#pragma once
#include "lz4_wave.h"

namespace zpk {

#define PJ_BLOCK 65536u                          // LZ4F block size of the frames this path takes (BD = 0x40: what the reference writes)
#define PJ_LIT 0x80000000u                       // reference: a literal — low 31 bits = byte offset in the compressed entry
struct PjBlock { u32 comp_off, comp_size /* bit 31: stored */, rec_base, out_size, out_off, nrec; };
enum { PJ_ERR = 0, PJ_CHANGED = 1, PJ_TOTAL = 2 };      // words of the flags array (PJ_TOTAL: two words)

// record of one sequence: output position in the block | literal length << 16 | literal position in the block << 32 | offset << 48
struct PjEmit {
    u64* recs; ZPK_LDS u32* mask; u32 op, nrec; bool err; int lane;
    __device__ __forceinline__ int operator()(const SeqBatch& q, int cnt, u32 lit_pos)
    {
        const bool act = lane < cnt;
        const u32 ll = act ? q.ll : 0u, ml = act ? q.ml : 0u;
        const u32 x = wave_scan_add(ll + ml);
        const u32 total = (u32)__builtin_amdgcn_readlane((int)x, 63);
        const u32 o = op + (x - (ll + ml));
        if (__ballot(act && (q.bad != 0 || ll >= PJ_BLOCK || (ml != 0 && q.off == 0))) != 0 || op + total > PJ_BLOCK) { err = true; return D_MALFORMED; }
        const bool some = act && (ll + ml) != 0;                 // (a block's last sequence may be empty: no bytes, no record)
        const u64 sm = __ballot(some);
        const u32 slot = __builtin_amdgcn_mbcnt_hi((u32)(sm >> 32), __builtin_amdgcn_mbcnt_lo((u32)sm, 0u));
        if (some) {
            recs[nrec + slot] = (u64)o | ((u64)ll << 16) | ((u64)(lit_pos & 0xFFFFu) << 32) | ((u64)(q.off & 0xFFFFu) << 48);
            __hip_atomic_fetch_or(mask + (o >> 5), 1u << (o & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        op += total; nrec += (u32)__popcll(sm);
        return D_OK;
    }
};

struct alignas(16) PjParseShared { Lz4WaveShared w; u32 mask[PJ_BLOCK / 32]; };

// one wave per block
__global__ __launch_bounds__(64) void k_pj_parse(const u8* __restrict__ src, u64 src_size, PjBlock* __restrict__ blocks, u32 nblocks,
                                                 u64* __restrict__ recs, u32* __restrict__ masks, u32* __restrict__ flags)
{
    __shared__ PjParseShared sh;
    const int lane = lane_id();
    const u32 b = uni((u32)blockIdx.x);
    if (b >= nblocks) return;
    const u32 coff = uni(blocks[b].comp_off), csz = uni(blocks[b].comp_size);
    if (csz >> 31) {                                             // stored: its bytes are literals as they stand
        if (lane == 0) { blocks[b].out_size = csz & 0x7FFFFFFFu; blocks[b].nrec = 0; }
        return;
    }
    for (u32 i = (u32)lane; i < PJ_BLOCK / 32; i += WAVE) sh.mask[i] = 0;
    wave_mem_fence();
    Watchdog wd; wd.arm((u64)csz + PJ_BLOCK);
    SeqStats stt = {};
    PjEmit E; E.recs = recs + uni(blocks[b].rec_base); E.mask = (ZPK_LDS u32*)sh.mask; E.op = 0; E.nrec = 0; E.err = false; E.lane = lane;
    u8* op = nullptr;
    struct Ref { PjEmit* e; __device__ __forceinline__ int operator()(const SeqBatch& q, int cnt, u32 lp) const { return (*e)(q, cnt, lp); } } ref{&E};
    const int rc = lz4_block_wave<2, true, Ref>(sh.w, wd, stt, src + coff, csz, src + src_size, nullptr, op, nullptr, lane, ref);
    wave_mem_fence();
    if (rc != D_OK || E.err || wd.fired) { if (lane == 0) atomicOr(&flags[PJ_ERR], 1u); return; }
    u32* const m = masks + (u64)b * (PJ_BLOCK / 32);
    for (u32 i = (u32)lane; i < PJ_BLOCK / 32; i += WAVE) m[i] = sh.mask[i];
    if (lane == 0) { blocks[b].out_size = E.op; blocks[b].nrec = E.nrec; }
}

// output offsets of the blocks (one wave; a frame of 1 GiB has 16 384 blocks)
// (first: the output offset of block 0 — a stream's step has the bytes of its history in front, zpk_stream.inc)
__global__ __launch_bounds__(64) void k_pj_scan(PjBlock* __restrict__ blocks, u32 nblocks, u32* __restrict__ flags, u32 first = 0)
{
    const int lane = lane_id();
    u64 base = first;
    for (u32 b0 = 0; b0 < nblocks; b0 += WAVE) {
        const u32 b = b0 + (u32)lane;
        const u32 sz = b < nblocks ? blocks[b].out_size : 0u;
        const u32 x = wave_scan_add(sz);
        const u64 off = base + (x - sz);
        if (b < nblocks) { if (off + sz > 0x7FFFFFF0ull) atomicOr(&flags[PJ_ERR], 2u); blocks[b].out_off = (u32)off; }
        base += (u32)__builtin_amdgcn_readlane((int)x, 63);
    }
    if (lane == 0) { flags[PJ_TOTAL] = (u32)base; flags[PJ_TOTAL + 1] = (u32)(base >> 32); }
}

// one workgroup of 256 threads per block (blocks b0 + blockIdx.x): references of its output bytes.  n = bytes of S (nothing beyond is touched)
__global__ __launch_bounds__(256) void k_pj_init(const PjBlock* __restrict__ blocks, u32 b0, u32 nblocks, const u64* __restrict__ recs,
                                                 const u32* __restrict__ masks, u32* __restrict__ S, u64 n, u32* __restrict__ flags, int independent)
{
    __shared__ u32 m[PJ_BLOCK / 32];
    __shared__ u32 pre[PJ_BLOCK / 32];             // set bits in front of word w
    __shared__ u32 part[256];
    const u32 b = b0 + blockIdx.x, tid = threadIdx.x;
    if (b >= nblocks) return;
    const PjBlock B = blocks[b];
    if ((u64)B.out_off + B.out_size > n) { if (tid == 0) atomicOr(&flags[PJ_ERR], 16u); return; }
    u32* const out = S + B.out_off;
    if (B.comp_size >> 31) {
        for (u32 i = tid; i < B.out_size; i += 256) out[i] = PJ_LIT | (B.comp_off + i);
        return;
    }
    const u32* const gm = masks + (u64)b * (PJ_BLOCK / 32);
    u32 cnt = 0;
    for (u32 k = 0; k < 8; k++) { const u32 w = tid * 8 + k; const u32 v = gm[w]; m[w] = v; pre[w] = cnt; cnt += (u32)__popc(v); }
    part[tid] = cnt;
    __syncthreads();
    u32 before = 0;                                // (256 partial counts: every thread sums what lies before it)
    for (u32 t = 0; t < tid; t++) before += part[t];
    __syncthreads();
    for (u32 k = 0; k < 8; k++) pre[tid * 8 + k] += before;
    __syncthreads();
    const u64* const R = recs + B.rec_base;
    bool bad = false;
    for (u32 pos = tid; pos < B.out_size; pos += 256) {
        const u32 w = pos >> 5;
        const u32 rank = pre[w] + (u32)__popc(m[w] & (0xFFFFFFFFu >> (31u - (pos & 31u))));      // sequences starting at or before pos
        if (rank == 0 || rank > B.nrec) { bad = true; continue; }
        const u64 r = R[rank - 1];
        const u32 o = (u32)r & 0xFFFFu, ll = (u32)(r >> 16) & 0xFFFFu, lp = (u32)(r >> 32) & 0xFFFFu, off = (u32)(r >> 48);
        const u32 rel = pos - o;
        if (rel < ll) out[pos] = PJ_LIT | (B.comp_off + lp + rel);
        else {
            const u32 here = independent ? pos : B.out_off + pos;       // bytes of history a match at this byte may use
            if (off == 0 || off > here) { bad = true; continue; }
            out[pos] = B.out_off + pos - off;
        }
    }
    if (bad) atomicOr(&flags[PJ_ERR], 4u);
}

// The frame is resolved in CHUNKS of consecutive blocks [b0, b1) — a chunk's references (4 bytes per output byte) should stay in the
// Infinity Cache — one after the other: a reference to a byte in front of the chunk is FINAL (that byte is in the output already).
#define PJ_ROUND0 8                              // flags[PJ_ROUND0 + r]: round r of the current chunk left a byte unresolved
#define PJ_MAX_ROUNDS 28
__device__ __forceinline__ void pj_range(const PjBlock* blocks, u32 b0, u32 b1, u32 nblocks, const u32* flags, u32& lo, u32& hi)
{
    lo = blocks[b0].out_off;
    hi = b1 < nblocks ? blocks[b1].out_off : flags[PJ_TOTAL];
}
// round r: S[i] = S[S[i]] for the chunk's bytes that still refer to a byte of the chunk (in place: a value read early or late is a valid
// reference either way).  Launched PJ_MAX_ROUNDS times per chunk without a host round trip: a round behind the last one that changed
// anything returns at once.
__global__ __launch_bounds__(256) void k_pj_jump(u32* __restrict__ S, const PjBlock* __restrict__ blocks, u32 b0, u32 b1, u32 nblocks,
                                                 u32* __restrict__ flags, u32 r)
{
    if (r > 0 && flags[PJ_ROUND0 + r - 1] == 0) return;
    u32 lo, hi;
    pj_range(blocks, b0, b1, nblocks, flags, lo, hi);
    // a workgroup owns 1024 consecutive references; thread t those at t, t + 256, t + 512, t + 768: every load of a wave is 256 contiguous
    // bytes, and the references it then follows are consecutive wherever the bytes belong to one match
    const u64 base = (u64)lo + (u64)blockIdx.x * 1024 + threadIdx.x;
    u32 v[4]; bool mine[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const u64 i = base + 256u * k; v[k] = i < hi ? S[i] : PJ_LIT; mine[k] = !(v[k] >> 31) && v[k] >= lo; }
    u32 w[4];
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = mine[k] ? S[v[k]] : v[k];
    bool open = false;
#pragma unroll
    for (int k = 0; k < 4; k++) if (mine[k]) { S[base + 256u * k] = w[k]; open |= !(w[k] >> 31) && w[k] >= lo; }
    // one word says "this round left something open": thousands of atomics on it would take longer than the round (12 ns each), so a
    // wave looks first and only the first few write
    if (__ballot(open) != 0 && lane_id() == 0 && __hip_atomic_load(&flags[PJ_ROUND0 + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
        __hip_atomic_store(&flags[PJ_ROUND0 + r], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the chunk's bytes: a literal of the compressed entry, or a byte of the output in front of the chunk
__global__ __launch_bounds__(256) void k_pj_gather(const u32* __restrict__ S, const PjBlock* __restrict__ blocks, u32 b0, u32 b1, u32 nblocks,
                                                   const u8* __restrict__ src, u64 src_size, u8* __restrict__ dst, u32* __restrict__ flags)
{
    u32 lo, hi;
    pj_range(blocks, b0, b1, nblocks, flags, lo, hi);
    const u64 i = (u64)(lo & ~3u) + ((u64)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= hi) return;
    bool bad = false;
    u32 word = 0;
    const u64 k0 = i < lo ? lo : i, k1 = i + 4 < hi ? i + 4 : hi;
    for (u64 k = k0; k < k1; k++) {
        const u32 v = S[k];
        u32 byte = 0;
        if (v >> 31) { const u32 a = v & 0x7FFFFFFFu; if (a >= src_size) bad = true; else byte = ld8(src + a); }
        else if (v < lo) byte = ld8(dst + v);
        else bad = true;                                         // (not resolved: cannot happen after PJ_MAX_ROUNDS)
        word |= byte << (8 * (u32)(k - i));
    }
    if (k0 == i && k1 == i + 4) *(ZPK_GLOBAL u32*)(dst + i) = word;
    else for (u64 k = k0; k < k1; k++) dst[k] = (u8)(word >> (8 * (u32)(k - i)));
    if (bad) atomicOr(&flags[PJ_ERR], 8u);
}

}  // namespace zpk
