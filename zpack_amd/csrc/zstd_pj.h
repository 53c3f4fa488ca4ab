// zstd_pj.h — ONE large Zstandard frame, its blocks side by side (the host read path; what the reference writer produces for a large entry:
// lib/zpack_write.c:179, ZSTD_compressCCtx = one frame of blocks of up to 128 KiB; read back by lib/zpack_read.c:380).
//
// One workgroup decodes one frame block after block at 30-40 MB/s (the FSE chain of a block is serial); 256 MiB are 2 048 blocks.
// What a block needs from the blocks before it:
//   * the bytes its matches copy          -> resolved for all blocks at once by pointer doubling over byte references (lz4_pj.h);
//   * the three repeat offsets            -> the sequences are decoded against SYMBOLS (zstd_fse4.h, BLOCKS), a scan over the blocks'
//                                            final histories gives every symbol its value (k_zpj_reps);
//   * a Huffman table (Treeless literals) -> the block reads the tree description of the block it inherits from itself (the host walk
//                                            knows which: nearly every block of a text frame is treeless);
//   * FSE tables (Repeat_Mode)            -> the block builds the table again from the description in the header of the block it
//                                            inherits from (the host walk knows where each description starts: it measures them);
//                                            libzstd 1.4.9 writes such blocks from level 9 on (text: 32 of 128 blocks at level 9, 112 at 15).
// Steps (all blocks side by side in each):
//   host        walks the block headers, literals-section headers and sequence-section headers (a few bytes per block): ZpjBlock table;
//   k_zstd_fse_blocks   sequences of every block -> 8-byte records (offset | match length | literal length), offsets symbolic where inherited;
//   k_zpj_lit   Huffman literals of every block -> the literal arena behind the compressed entry (raw / RLE literals stay where they are);
//   k_zpj_reps  one thread: the repeat offsets at every block's start;
//   k_zpj_pos   per block: prefix sums of the sequence lengths -> where every sequence starts in the block's output and in its literals,
//               the block's output size, a bit mask of the sequence starts;
//   k_pj_scan   output offsets of the blocks;  then per chunk of blocks  k_zpj_init (a reference per output byte), k_pj_jump x rounds,
//               k_pj_gather  — lz4_pj.h, unchanged.
// Verdicts: as in lz4_pj.h this path finishes an entry only when everything was regular AND the XXH3 of the assembled output is the
// expected one; anything else — and any hash mismatch — goes to the one-wave decoder, which alone gives verdicts.
#pragma once
#include "lz4_pj.h"
#include "zstd_fse4.h"

namespace zpk {

#define ZPJ_BLOCK (128u << 10)
#define ZPJ_NONE 0xFFFFFFFFu
#define ZPJ_TREE_ONLY 3u                           // ZpjBlock::type of a stream step's first table entry: a block of an earlier step, here only for its Huffman tree
struct ZpjBlock {
    u32 hdr_off;             // offset of the 3-byte block header in the compressed entry
    u32 size;                // Block_Size (Compressed / Raw: bytes of content; RLE: the regenerated size)
    u32 type;                // 0 Raw, 1 RLE, 2 Compressed
    u32 lit_type;            // Compressed: 0 Raw, 1 RLE, 2 Compressed, 3 Treeless
    u32 lit_size;            // ... regenerated literal bytes
    u32 lit_used;            // ... bytes of the block the literals section takes
    u32 lit_ref;             // ... reference of literal byte 0: offset in [compressed entry | literal arena]; RLE literals: the byte's offset
    u32 lit_base;            // ... decoded literals: their offset in the literal arena
    u32 tree_src;            // ... Treeless: the block whose tree description it uses
    u32 nseq;                // ... sequences
    u32 seq_base;            // ... index of its first sequence record (a block owns nseq + 1 slots: the last one stands for the trailing literals)
    u32 rep_in[3];           // device (k_zpj_reps): repeat offsets at the block's start
    u32 tab_off[3];          // ... a Repeat_Mode table (LL, OF, ML): where the table it inherits is described (offset in the compressed entry)
    u32 tab_modes;           // ... and how: 2 bits per kind (0 predefined, 1 RLE, 2 FSE description, 3 nowhere)
};
// words of the flags array this path adds (lz4_pj.h: PJ_ERR, PJ_TOTAL, PJ_ROUND0 ..): the counters k_zstd_fse_blocks works with
#define ZPJ_CNT 40                                 // flags + ZPJ_CNT = its `counters` (ZF_COUNT_WORD 1, ZF_HEAD 8, ZF_WATCHDOG_WORD 11..13)

struct ZpjLitHdr { u32 type, hl, streams, regen, csize; bool ok; };
__device__ __forceinline__ ZpjLitHdr zpj_lit_hdr(const u8* p, u64 size)
{
    ZpjLitHdr h; h.ok = false; h.type = 0; h.hl = 0; h.streams = 1; h.regen = 0; h.csize = 0;
    if (size < 1) return h;
    const u32 b0 = uld8(p);
    h.type = b0 & 3;
    const u32 fmt = (b0 >> 2) & 3;
    if (h.type < 2) {
        if ((fmt & 1) == 0) { h.hl = 1; h.regen = b0 >> 3; }
        else if (fmt == 1) { if (size < 2) return h; h.hl = 2; h.regen = (b0 >> 4) | ((u32)uld8(p + 1) << 4); }
        else { if (size < 3) return h; h.hl = 3; h.regen = (b0 >> 4) | ((u32)uld8(p + 1) << 4) | ((u32)uld8(p + 2) << 12); }
        h.csize = h.type == 0 ? h.regen : 1u;
    } else {
        if (size < 5) return h;
        const u64 v = uld32(p);
        if (fmt == 0) { h.hl = 3; h.streams = 1; h.regen = (u32)(v >> 4) & 0x3FF; h.csize = (u32)(v >> 14) & 0x3FF; }
        else if (fmt == 1) { h.hl = 3; h.streams = 4; h.regen = (u32)(v >> 4) & 0x3FF; h.csize = (u32)(v >> 14) & 0x3FF; }
        else if (fmt == 2) { h.hl = 4; h.streams = 4; h.regen = (u32)(v >> 4) & 0x3FFF; h.csize = (u32)(v >> 18); }
        else { h.hl = 5; h.streams = 4; h.regen = (u32)(v >> 4) & 0x3FFFF; h.csize = (u32)(v >> 22) | ((u32)uld8(p + 4) << 10); }
    }
    h.ok = h.regen <= ZPJ_BLOCK && (u64)h.hl + h.csize <= size;
    return h;
}

// one wave per block: Huffman-coded literals -> the literal arena (src + arena_off + lit_base).  (The launch bounds are k_zstd_exec's: the
// out-of-line Huffman decoder both call is compiled for the loosest bound among its callers — with a plain (64) here k_zstd_exec lost a
// wave per SIMD.)
#ifndef ZSTD_EXEC_WAVES_FOR_CALLEES
#define ZSTD_EXEC_WAVES_FOR_CALLEES 4
#endif
__global__ __launch_bounds__(64, ZSTD_EXEC_WAVES_FOR_CALLEES) void k_zpj_lit(u8* __restrict__ src, u64 src_size, u64 arena_off, const ZpjBlock* __restrict__ blocks, u32 nblocks,
                                                u32* __restrict__ flags)
{
    __shared__ __attribute__((aligned(16))) u8 sh_raw[__builtin_offsetof(ZstdShared, ll)];
    ZstdShared& sh = *(ZstdShared*)sh_raw;
    const int lane = lane_id();
    const u32 b = uni((u32)blockIdx.x);
    if (b >= nblocks) return;
    const u32 type = uni(blocks[b].type), lt = uni(blocks[b].lit_type);
    if (type != 2 || lt < 2) return;
    if (threadIdx.x == 0) { sh.huf_valid = 0; sh.defaults_built = 0; }
    __syncthreads();
    const u32 hdr = uni(blocks[b].hdr_off), size = uni(blocks[b].size), lit_size = uni(blocks[b].lit_size), lit_used = uni(blocks[b].lit_used);
    const u8* const rd_hi = src + src_size;
    Watchdog wd; wd.arm((u64)size + lit_size + 65536);
    bool ok = (u64)hdr + 3 + size <= src_size;
    if (ok && lt == 3) {                                         // the tree of the block this one inherits from
        const u32 t = uni(blocks[b].tree_src);
        ok = t < b && (uni(blocks[t].type) == 2 || uni(blocks[t].type) == ZPJ_TREE_ONLY) && uni(blocks[t].lit_type) == 2;
        if (ok) {
            const u8* const tp = src + uni(blocks[t].hdr_off) + 3;
            const ZpjLitHdr th = zpj_lit_hdr(tp, uni(blocks[t].size));
            ok = th.ok && th.type == 2;
            if (ok) { ByteWindow win; ok = huf_read_tree(sh, win, tp + th.hl, th.csize, lane) >= 0; }
        }
        __syncthreads();
    }
    if (ok) {
        ZFrameState fs; fs.zs = nullptr; fs.wd = &wd; fs.rep0 = 1; fs.rep1 = 4; fs.rep2 = 8; fs.seq_tables_valid = false;
        fs.al_ll = fs.al_of = fs.al_ml = 0; fs.pre = nullptr; fs.pre_idx = 0;
        ZLiterals L;
        const int rc = zstd_literals(sh, fs, src + hdr + 3, size, rd_hi, src + arena_off + uni(blocks[b].lit_base), L, lane);
        ok = rc == D_OK && !wd.fired && L.lit_size == lit_size && L.used == lit_used && !L.rle;
    }
    wave_mem_fence();
    if (!ok && lane == 0) atomicOr(&flags[PJ_ERR], 32u);
}

// one wave: the repeat offsets at every block's start (RFC 8878 3.1.1.5: 1, 4, 8 at the frame's start; Raw and RLE blocks and blocks
// without sequences pass them on); a block the sequence stage did not finish makes the entry irregular.  64 blocks are loaded at a
// time, one per lane; the walk over them is a scalar loop over the lanes (a thread that loaded block after block took 1 ms for
// 2 048 blocks: four dependent loads each).
// (r0, r1, r2: the history in front of block 0 — 1, 4, 8 at a frame's start; a stream's step continues with what the step before left
// in flags[ZPJ_REPS ..])
#define ZPJ_REPS 36
__global__ __launch_bounds__(64) void k_zpj_reps(ZpjBlock* __restrict__ blocks, u32 nblocks, const u32* __restrict__ state, const u32* __restrict__ rep_out, u32* __restrict__ flags,
                                                 u32 r0 = 1u, u32 r1 = 4u, u32 r2 = 8u)
{
    if (blockIdx.x != 0) return;
    const int lane = lane_id();
    u32 c0 = r0, c1 = r1, c2 = r2;                               // (uniform)
    u32 err = 0;
    for (u32 base = 0; base < nblocks; base += WAVE) {
        const u32 b = base + (u32)lane;
        const bool in = b < nblocks;
        const bool has = in && blocks[b].type == 2 && blocks[b].nseq != 0;
        const u32 st = has ? state[b] : 1u;
        const u32 o0 = has ? rep_out[3u * b] : 0u, o1 = has ? rep_out[3u * b + 1u] : 0u, o2 = has ? rep_out[3u * b + 2u] : 0u;
        const u64 hm = __ballot(has);
        if (__ballot(has && st != 1u) != 0) err |= 64u;
        u32 i0 = 0, i1 = 0, i2 = 0;                              // this lane's block: the history at its start
        const int cnt = (int)(nblocks - base < WAVE ? nblocks - base : WAVE);
        for (int i = 0; i < cnt; i++) {
            if (lane == i) { i0 = c0; i1 = c1; i2 = c2; }
            if (!((hm >> i) & 1)) continue;
            const u32 v[3] = { (u32)__builtin_amdgcn_readlane((int)o0, i), (u32)__builtin_amdgcn_readlane((int)o1, i), (u32)__builtin_amdgcn_readlane((int)o2, i) };
            u32 nx[3];
            #pragma unroll
            for (int j = 0; j < 3; j++) {
                if (v[j] >> ZF_SYM_SHIFT) {
                    const u32 k = (v[j] >> ZF_SYM_SHIFT) - 1u, dec = ((1u << ZF_SYM_SHIFT) - 1u) - (v[j] & ((1u << ZF_SYM_SHIFT) - 1u));
                    const u32 r = k == 0 ? c0 : (k == 1 ? c1 : c2);
                    if (k > 2u || dec >= r) { err |= 128u; nx[j] = 1u; } else nx[j] = r - dec;
                } else nx[j] = v[j];
            }
            c0 = nx[0]; c1 = nx[1]; c2 = nx[2];
        }
        if (in) { blocks[b].rep_in[0] = i0; blocks[b].rep_in[1] = i1; blocks[b].rep_in[2] = i2; }
    }
    if (lane == 0) { flags[ZPJ_REPS] = c0; flags[ZPJ_REPS + 1] = c1; flags[ZPJ_REPS + 2] = c2; }
    if (err && lane == 0) atomicOr(&flags[PJ_ERR], err);
}

struct alignas(16) ZpjPosShared { u32 mask[ZPJ_BLOCK / 32]; u32 wsum[2][4]; };
// one workgroup of 256 threads per block: where every sequence starts (output position | literal position << 32), the block's output
// size, the bit mask of the sequence starts
__global__ __launch_bounds__(256) void k_zpj_pos(const ZpjBlock* __restrict__ blocks, PjBlock* __restrict__ pj, u32 nblocks, const u64* __restrict__ recs,
                                                 u64* __restrict__ pos, u32* __restrict__ masks, u32* __restrict__ flags)
{
    __shared__ ZpjPosShared sh;
    const u32 b = blockIdx.x, tid = threadIdx.x;
    if (b >= nblocks) return;
    const ZpjBlock B = blocks[b];
    if (B.type != 2) {
        if (tid == 0) { pj[b].out_size = B.type == ZPJ_TREE_ONLY ? 0u : B.size; pj[b].nrec = 0; }
        return;
    }
    for (u32 i = tid; i < ZPJ_BLOCK / 32; i += 256) sh.mask[i] = 0;
    __syncthreads();
    const int lane = (int)(tid & 63u), wave = (int)(tid >> 6);
    const u64* const R = recs + B.seq_base;
    u64* const P = pos + B.seq_base;
    u32 base_len = 0, base_ll = 0;
    bool bad = false;
    for (u32 i0 = 0; i0 < B.nseq; i0 += 256) {
        const u32 i = i0 + tid;
        u32 ll = 0, ml = 0;
        if (i < B.nseq) { const u64 v = R[i]; ml = (u32)(v >> ZF_SEQ_OFF_BITS) & ((1u << ZF_SEQ_ML_BITS) - 1u); ll = (u32)(v >> (ZF_SEQ_OFF_BITS + ZF_SEQ_ML_BITS)); }
        const u32 len = ll + ml;
        const u32 xl = wave_scan_add(len), xq = wave_scan_add(ll);               // inclusive, inside the wave
        if (lane == 63) { sh.wsum[0][wave] = xl; sh.wsum[1][wave] = xq; }
        __syncthreads();
        u32 wl = 0, wq = 0, tl = 0, tq = 0;
        #pragma unroll
        for (int w = 0; w < 4; w++) { const u32 a = sh.wsum[0][w], c = sh.wsum[1][w]; if (w < wave) { wl += a; wq += c; } tl += a; tq += c; }
        const u32 o = base_len + wl + (xl - len), lp = base_ll + wq + (xq - ll);
        if (i < B.nseq) {
            if (o >= ZPJ_BLOCK || len > ZPJ_BLOCK) bad = true;
            else { P[i] = (u64)o | ((u64)lp << 32); __hip_atomic_fetch_or(&sh.mask[o >> 5], 1u << (o & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
        }
        __syncthreads();
        if (tl > ZPJ_BLOCK || base_len + tl > ZPJ_BLOCK) { bad = true; break; }     // (uniform)
        base_len += tl; base_ll += tq;
    }
    // the trailing literals: what the literals section holds beyond the sequences' literal lengths
    const bool short_lits = base_ll > B.lit_size;
    const u32 rest = short_lits ? 0u : B.lit_size - base_ll;
    const u32 out = base_len + rest;
    if (short_lits || out > ZPJ_BLOCK) bad = true;
    if (!bad && tid == 0) {
        P[B.nseq] = (u64)base_len | ((u64)base_ll << 32);
        if (rest) sh.mask[base_len >> 5] |= 1u << (base_len & 31u);
    }
    __syncthreads();
    if (__syncthreads_or(bad ? 1 : 0)) { if (tid == 0) { atomicOr(&flags[PJ_ERR], 256u); pj[b].out_size = 0; pj[b].nrec = 0; } return; }
    u32* const gm = masks + (u64)b * (ZPJ_BLOCK / 32);
    for (u32 i = tid; i < ZPJ_BLOCK / 32; i += 256) gm[i] = sh.mask[i];
    if (tid == 0) { pj[b].out_size = out; pj[b].nrec = B.nseq + (rest ? 1u : 0u); }
}

// one workgroup of 256 threads per block (blocks b0 + blockIdx.x): the references of its output bytes (lz4_pj.h: PJ_LIT | offset in
// [compressed entry | literal arena], or the output position the byte copies).  n = bytes of S.
__global__ __launch_bounds__(256) void k_zpj_init(const ZpjBlock* __restrict__ blocks, const PjBlock* __restrict__ pj, u32 b0, u32 nblocks,
                                                  const u64* __restrict__ recs, const u64* __restrict__ pos, const u32* __restrict__ masks,
                                                  u32* __restrict__ S, u64 n, u32* __restrict__ flags)
{
    __shared__ u32 m[ZPJ_BLOCK / 32];
    __shared__ u32 pre[ZPJ_BLOCK / 32];             // set bits in front of word w
    __shared__ u32 part[256];
    const u32 b = b0 + blockIdx.x, tid = threadIdx.x;
    if (b >= nblocks) return;
    const ZpjBlock B = blocks[b];
    const u32 out_off = pj[b].out_off, out_size = pj[b].out_size;
    if ((u64)out_off + out_size > n) { if (tid == 0) atomicOr(&flags[PJ_ERR], 16u); return; }
    u32* const out = S + out_off;
    if (B.type != 2) {                                           // Raw: its bytes are literals where they lie; RLE: one byte, every time
        const u32 step = B.type == 0 ? 1u : 0u;
        for (u32 i = tid; i < out_size; i += 256) out[i] = PJ_LIT | (B.hdr_off + 3u + step * i);
        return;
    }
    const u32* const gm = masks + (u64)b * (ZPJ_BLOCK / 32);
    u32 cnt = 0;
    for (u32 k = 0; k < 16; k++) { const u32 w = tid * 16 + k; const u32 v = gm[w]; m[w] = v; pre[w] = cnt; cnt += (u32)__popc(v); }
    part[tid] = cnt;
    __syncthreads();
    u32 before = 0;
    for (u32 t = 0; t < tid; t++) before += part[t];
    __syncthreads();
    for (u32 k = 0; k < 16; k++) pre[tid * 16 + k] += before;
    __syncthreads();
    const u64* const R = recs + B.seq_base;
    const u64* const P = pos + B.seq_base;
    const u32 nrec = pj[b].nrec;
    const bool lit_rle = B.lit_type == 1;
    bool bad = false;
    // four positions per trip: their table reads are in flight together (one at a time the loop waited ~1 us per position)
    for (u32 p0 = tid; p0 < out_size; p0 += 1024) {
        u32 idx[4]; u64 at[4], rec[4]; bool in[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 p = p0 + 256u * (u32)k;
            in[k] = p < out_size;
            const u32 w = (in[k] ? p : 0u) >> 5;
            const u32 rank = pre[w] + (u32)__popc(m[w] & (0xFFFFFFFFu >> (31u - (p & 31u))));      // sequences starting at or before p
            if (in[k] && (rank == 0 || rank > nrec)) { bad = true; in[k] = false; }
            idx[k] = in[k] ? rank - 1 : 0u;
        }
        #pragma unroll
        for (int k = 0; k < 4; k++) { at[k] = P[idx[k]]; rec[k] = idx[k] < B.nseq ? R[idx[k]] : 0ull; }
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            if (!in[k]) continue;
            const u32 p = p0 + 256u * (u32)k;
            const u32 o = (u32)at[k], lp = (u32)(at[k] >> 32);
            const u32 rel = p - o;
            u32 ll, off;
            if (idx[k] < B.nseq) { off = (u32)rec[k] & ((1u << ZF_SEQ_OFF_BITS) - 1u); ll = (u32)(rec[k] >> (ZF_SEQ_OFF_BITS + ZF_SEQ_ML_BITS)); }
            else { ll = out_size - o; off = 0; }                 // the trailing literals
            if (rel < ll) out[p] = PJ_LIT | (B.lit_ref + (lit_rle ? 0u : lp + rel));
            else {
                if (off >> ZF_SYM_SHIFT) {                       // inherited: the block's starting history, counted down
                    const u32 kk = (off >> ZF_SYM_SHIFT) - 1u, dec = ((1u << ZF_SYM_SHIFT) - 1u) - (off & ((1u << ZF_SYM_SHIFT) - 1u));
                    const u32 r = kk == 0 ? B.rep_in[0] : (kk == 1 ? B.rep_in[1] : B.rep_in[2]);
                    if (kk > 2u || dec >= r) { bad = true; continue; }
                    off = r - dec;
                }
                const u64 here = (u64)out_off + p;               // bytes of the frame in front of this one
                if (off == 0 || off > here) { bad = true; continue; }
                out[p] = (u32)(here - off);
            }
        }
    }
    if (bad) atomicOr(&flags[PJ_ERR], 4u);
}

}  // namespace zpk
