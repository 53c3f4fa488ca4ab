// zpk_codec.hip — kernels + C-ABI of the MI355X entry codec (include/zpack_codec.h).
//
// Launch structure of one decode batch (all on one HIP stream, no host synchronisation):
//   1. k_classify   one thread per entry: the guards of zpack_read_file (lib/zpack_read.c:328-332,
//                   :354, :459) in the reference's order, then the entry index is appended to the
//                   work list of its method (wave-aggregated atomics).
//   2. k_stored / k_lz4_wave   one wave per work-list slot.
//   3. k_zstd_fse -> k_zstd_exec -> k_zstd   Zstandard: persistent grids that pull entries from the work list with an
//                   atomic dequeue (FSE sequence pre-decode four streams per wave; literals + execution + XXH3 of the
//                   pre-decoded entries; the full decoder for whatever is left over) — zstd_fse4.h, zstd_wg.h.
// Entries are independent, so there is no inter-workgroup communication besides the
// dequeue counters.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <time.h>
#include <pthread.h>
#include <thread>
#include <atomic>
#include <vector>
#include <algorithm>


#include "zpk_device.h"
#include "xxh3_device.h"
#include "xxh3_span.h"
#include "lz4_wave.h"
#include "lx_ring.h"
#ifndef LX_WAVES_PER_SIMD
#define LX_WAVES_PER_SIMD 6
#endif
#include "zstd_wg.h"
#include "zstd_fse4.h"
#include "zstd_ring.h"
#include "lz4_pj.h"
#include "zstd_pj.h"

using namespace zpk;

// counters layout (u32): [0..3] count per work list, [4..7] dequeue head per list
enum { L_NONE = 0, L_ZSTD = 1, L_LZ4 = 2, L_COUNT = 4, N_LISTS = 3, L_LZ4_RUNS = 100 /* k_classify only: the LZ4 entries k_lz4_left takes (its list is slot N_LISTS + 2) */ };
// [8] dequeue head of k_zstd_fse, [9] Zstandard entries finished on pre-decoded sequences, [10] finished by the fused decoder
enum { C_ZSTD_TWO_STAGE = 9, C_ZSTD_FUSED = 10, C_EXEC_HEAD = 16, C_LEFT_COUNT = 17,
       C_RETRY_LZ4 = 18, C_RETRY_ZSTD = 19,     // entries whose decoder ran out of its time budget: decoded again by the retry launches
       C_RETRY_HEAD = 20,                       // dequeue head of the Zstandard retry launch
       C_LZ4_LEFT = 21, C_LZ4_LEFT_HEAD = 22,   // LZ4 entries that are mostly runs (k_classify: compressed to less than 1/8): k_lz4_left's list, its dequeue head
       C_ENC_CLASS = 28,                        // (three words) encode batches: does the batch hold entries for k_encode<12> / <13> / <14> at all
       C_ORDER_SPAN = 26,                       // (two words) largest size class and largest 15 - class among the Zstandard / LZ4 entries
       C_ORDER = 32,                            // k_order_*: [2 lists][16 classes] entry counts, then the same again as fill cursors
       N_COUNTERS = 32 + 64 };
enum { N_LISTS_ALLOC = N_LISTS + 5 };           // + the two retry lists + the LZ4 entries of long runs (k_lz4_left) + the two ordered lists

// ------------------------------------------------------------------------------------ kernels

#define ORD_CLASSES 16
__device__ __forceinline__ int order_class(u64 size)
{
    const int lg = 63 - __clzll((long long)(size | 1));
    const int b = 26 - lg;                                                            // >= 64 MiB: class 0 ... < 4 KiB: class 15
    return b < 0 ? 0 : (b > ORD_CLASSES - 1 ? ORD_CLASSES - 1 : b);
}
// A batch of ONE size class (the uniform workloads) has nothing to order by size; there the entries that did not compress — payload not
// smaller than ~15/16 of the size: stored LZ4 blocks, raw Zstandard blocks, a copy that decodes several times faster than anything
// compressed — go LAST: they are what is left to fill the final round with (C2: +0.75 % over six A/B pairs; on the ragged c4_mixed the
// same key inside every size class measured -4 % +- noise, so it is not used there).
__device__ __forceinline__ int order_fast(u64 size, u64 comp) { return comp * 16 >= size * 15 ? 1 : 0; }
__global__ __launch_bounds__(256) void k_classify(const zpk_decode_desc* __restrict__ desc, u64 n, u64 src_size, u64 dst_size,
                                                  zpk_decode_result* __restrict__ res, u32* __restrict__ lists, u64 list_stride,
                                                  u32* __restrict__ counters)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    zpk_decode_desc d; memset(&d, 0, sizeof(d));
    if (live) d = desc[i];
    zpk_decode_result r; r.status = R_OK; r.detail = 0; r.produced = 0; r.hash = 0;
    int list = -1;
    // lib/zpack_read.c:328-332, in this order
    if (!live) list = -1;
    else if (d.comp_size == 0) r.status = R_OK;
    else if (d.dst_capacity < d.uncomp_size) r.status = R_BUFFER_TOO_SMALL;
    else if (d.src_offset + d.comp_size >= src_size || d.src_offset > src_size || d.comp_size > src_size - d.src_offset)
        r.status = R_FILE_OFFSET_INVALID;
    else if (d.dst_offset > dst_size || d.dst_capacity > dst_size - d.dst_offset) { r.status = R_BUFFER_TOO_SMALL; r.detail = 0xBAD0D57u; }
    else if (d.method == ZPK_METHOD_NONE) {
        if (d.uncomp_size > d.comp_size) r.status = R_FILE_SIZE_INVALID;      // :354
        else list = L_NONE;
    }
    else if (d.method == ZPK_METHOD_ZSTD) list = L_ZSTD;
    else if (d.method == ZPK_METHOD_LZ4) list = d.comp_size < (d.uncomp_size >> 3) ? L_LZ4_RUNS : L_LZ4;      // mostly runs: k_lz4_left's
    else r.status = R_COMP_METHOD_INVALID;                                     // :459
    // an entry that goes on a work list is not decoded yet: until its decoder writes the verdict the slot says so
    // (a decoder that never ran must not read as R_OK)
    if (list >= 0) { r.status = R_DECOMPRESS_FAILED; r.detail = 0xFFFFFFFFu; }
    if (live) res[i] = r;
    // wave-aggregated append: one atomic per list per wave (a per-lane atomicAdd on three hot words cost 1.1 ms / 100k entries)
    const int lane = lane_id();
    {   // the span of size classes among the entries that go to a decoder (k_order_*)
        const bool dec = list == L_ZSTD || list == L_LZ4;
        u32 hi = dec ? (u32)order_class(d.uncomp_size) + 1u : 0u, inv = dec ? (u32)(ORD_CLASSES - order_class(d.uncomp_size)) : 0u;   // (+1: 0 = none)
        #pragma unroll
        for (int m = 1; m < 64; m <<= 1) { const u32 a = (u32)__shfl_xor((int)hi, m, 64), b2 = (u32)__shfl_xor((int)inv, m, 64); hi = a > hi ? a : hi; inv = b2 > inv ? b2 : inv; }
        if (lane == 0 && hi) { atomicMax(&counters[C_ORDER_SPAN], hi - 1u); atomicMax(&counters[C_ORDER_SPAN + 1], inv - 1u); }
    }
    #pragma unroll
    for (int L = 0; L < N_LISTS; L++) {
        const u64 m = __ballot(list == L);
        if (m == 0) continue;
        const int leader = __ffsll((long long)m) - 1;
        u32 base = 0;
        if (lane == leader) base = atomicAdd(&counters[L], (u32)__popcll(m));
        base = (u32)__shfl((int)base, leader, 64);
        if (list == L) lists[(u64)L * list_stride + base + (u32)__popcll(m & ((1ull << lane) - 1))] = (u32)i;
    }
    {   // k_lz4_left's list: storage slot N_LISTS + 2, length in counters[C_LZ4_LEFT]
        const u64 m = __ballot(list == L_LZ4_RUNS);
        if (m != 0) {
            const int leader = __ffsll((long long)m) - 1;
            u32 base = 0;
            if (lane == leader) base = atomicAdd(&counters[C_LZ4_LEFT], (u32)__popcll(m));
            base = (u32)__shfl((int)base, leader, 64);
            if (list == L_LZ4_RUNS) lists[(u64)(N_LISTS + 2) * list_stride + base + (u32)__popcll(m & ((1ull << lane) - 1))] = (u32)i;
        }
    }
}

// ---- largest entries first -----------------------------------------------------------------------------------------------------
// One wave (or one FSE row) works on one entry, and a batch is only a few rounds of the resident waves: with entries of 4 KiB ... 1 MiB
// in archive order, a 1 MiB entry that starts in the last round runs on alone for its whole length while the rest of the chip idles.
// The Zstandard and LZ4 work lists are therefore re-ordered by size class (floor(log2 uncomp_size), largest first: longest
// processing time first) with a two-kernel counting sort — per-workgroup LDS histograms, one global atomic per class and
// workgroup; entries of one class keep their neighbourhood.  (Uniform batches come out in nearly the order they went in.)
// (k_classify leaves the largest class and the largest 15 - class it saw in counters[C_ORDER_SPAN], [C_ORDER_SPAN + 1]: a batch of ONE
// class — the uniform workloads — is copied through in its own order.)
__device__ __forceinline__ bool order_single_class(const u32* counters) { return counters[C_ORDER_SPAN] + counters[C_ORDER_SPAN + 1] == ORD_CLASSES - 1; }
// rank of this lane among the lanes of its wave with the same class (in lane order), and how many there are
__device__ __forceinline__ void order_wave_rank(int b, int lane, u32& rank, u32& count)
{
    rank = 0; count = 0;
    u64 todo = __ballot(b >= 0);
    while (todo) {
        const int cls = __shfl(b, __ffsll((long long)todo) - 1, 64);
        const u64 m = __ballot(b == cls);
        if (b == cls) { rank = (u32)__popcll(m & ((1ull << lane) - 1)); count = (u32)__popcll(m); }
        todo &= ~m;
    }
}
__global__ __launch_bounds__(256) void k_order_count(const zpk_decode_desc* __restrict__ desc, const u32* __restrict__ lists, u64 list_stride,
                                                     u32* __restrict__ counters, int fast_last)
{
    __shared__ u32 h[ORD_CLASSES];
    const int L = blockIdx.y == 0 ? L_ZSTD : L_LZ4;
    const u32 cnt = counters[L];
    const bool uniform = order_single_class(counters);
    if ((u64)blockIdx.x * 256 >= cnt || (uniform && !fast_last)) return;
    if (threadIdx.x < ORD_CLASSES) h[threadIdx.x] = 0;
    __syncthreads();
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    const int lane = lane_id();
    int b = -1;
    if (k < cnt) { const zpk_decode_desc& d = desc[lists[(u64)L * list_stride + k]]; b = uniform ? order_fast(d.uncomp_size, d.comp_size) : order_class(d.uncomp_size); }
    u32 rank, count;
    order_wave_rank(b, lane, rank, count);
    if (b >= 0 && rank == 0) atomicAdd(&h[b], count);
    __syncthreads();
    if (threadIdx.x < ORD_CLASSES && h[threadIdx.x]) atomicAdd(&counters[C_ORDER + blockIdx.y * ORD_CLASSES + threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_order_fill(const zpk_decode_desc* __restrict__ desc, const u32* __restrict__ lists, u64 list_stride,
                                                    u32* __restrict__ ordered /* two lists of list_stride */, u32* __restrict__ counters, int fast_last)
{
    __shared__ u32 wcount[4][ORD_CLASSES], base[ORD_CLASSES];
    const int L = blockIdx.y == 0 ? L_ZSTD : L_LZ4;
    const u32 cnt = counters[L];
    if ((u64)blockIdx.x * 256 >= cnt) return;
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    const bool uniform = order_single_class(counters);
    if (uniform && !fast_last) { if (k < cnt) ordered[(u64)blockIdx.y * list_stride + k] = lists[(u64)L * list_stride + k]; return; }
    if (threadIdx.x < 4 * ORD_CLASSES) (&wcount[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int lane = lane_id(), w = threadIdx.x >> 6;
    u32 e = 0; int b = -1;
    if (k < cnt) { e = lists[(u64)L * list_stride + k]; b = uniform ? order_fast(desc[e].uncomp_size, desc[e].comp_size) : order_class(desc[e].uncomp_size); }
    u32 rank, count;
    order_wave_rank(b, lane, rank, count);
    if (b >= 0 && rank == 0) wcount[w][b] = count;
    __syncthreads();
    if (threadIdx.x < ORD_CLASSES) {                                                   // this workgroup's range of the class: entries stay in list order inside it
        const int c = threadIdx.x;
        const u32 total = wcount[0][c] + wcount[1][c] + wcount[2][c] + wcount[3][c];
        const u32* const hist = counters + C_ORDER + blockIdx.y * ORD_CLASSES;
        u32 before = 0;
        for (int j = 0; j < c; j++) before += hist[j];
        base[c] = total ? before + atomicAdd(&counters[C_ORDER + 2 * ORD_CLASSES + blockIdx.y * ORD_CLASSES + c], total) : 0;
    }
    __syncthreads();
    if (b >= 0) {
        u32 at = base[b] + rank;
        for (int j = 0; j < w; j++) at += wcount[j][b];
        ordered[(u64)blockIdx.y * list_stride + at] = e;
    }
}

// one wave per work-list slot: the hardware dispatcher is the load balancer
__device__ __forceinline__ bool my_slot(const u32* counters, int list, u32& idx)
{
    idx = uni((u32)(((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6));           // 64-bit: n blocks x 64 threads passes 2^32 at n = 2^26
    return idx < uni(counters[list]);
}

__device__ __forceinline__ void finish_entry(const zpk_decode_desc& d, zpk_decode_result* res, u32 e, int status, u32 detail,
                                             u64 produced, const u8* out, int lane)
{
    u64 h = 0;
    lane0_guard();
    if (status == R_OK) {                                                      // (ZPK_DF_SKIP_HASH: the hash is still produced, the status ignores it)
        wave_mem_fence();
#ifndef ZPK_ABL_NOHASH       // (developer ablation: instruction counters without the hash pass)
        h = xxh3_64_wave(out, d.uncomp_size, lane);                            // lib/zpack_read.c:466
#endif
        if (h != d.expect_hash && !(d.flags & ZPK_DF_SKIP_HASH)) status = R_FILE_HASH_MISMATCH;   // :467-468
    }
    lane0_guard();
    if (lane == 0) {
        zpk_decode_result r; r.status = status; r.detail = detail; r.produced = produced; r.hash = h;
        res[e] = r;
    }
}

// method 0: copy + hash fused — each 1 KiB block is loaded once, stored, and folded into the hash
__global__ __launch_bounds__(256) void k_stored(const u8* __restrict__ src, const zpk_decode_desc* __restrict__ desc,
                                                u8* __restrict__ dst, zpk_decode_result* __restrict__ res,
                                                const u32* __restrict__ list, const u32* __restrict__ counters)
{
    const int lane = lane_id();
    u32 idx;
    if (my_slot(counters, L_NONE, idx)) {
        const u32 e = uni(list[idx]);
        const zpk_decode_desc d = desc[e];
        const u8* in = uni_ptr(src + d.src_offset);
        u8* out = uni_ptr(dst + d.dst_offset);
        const u64 len = uni64(d.uncomp_size);
        u64 h;
        if (len <= 240) {
            for (u64 i = lane; i < len; i += WAVE) st8(out + i, ld8(in + i));
            h = 0;
            lane0_guard();
            if (lane == 0) h = xxh3_short(in, (u32)len);
            h = uni64(h);
        } else {
            Xxh3Wave st; st.init(lane);
            const u64 nblocks = (len - 1) >> 10;
            const u8* q = in + 16 * lane;
            u8* o = out + 16 * lane;
            u128 cur = {0, 0};
            if (nblocks) cur = ld128(q);
            for (u64 b = 0; b < nblocks; b++) {
                u128 nxt = {0, 0};
                if (b + 1 < nblocks) nxt = ld128(q + ((b + 1) << 10));
                st128(o + (b << 10), cur);
                st.block(cur);
                cur = nxt;
            }
            const u64 done = nblocks << 10;
            for (u64 i = done + lane; i < len; i += WAVE) st8(out + i, ld8(in + i));
            const u32 nstripes = (u32)(((len - 1) - done) >> 6);
            h = st.finish(in + done, nstripes, in + len, len, lane);
        }
        int status = R_OK;
        if (!(d.flags & ZPK_DF_SKIP_HASH) && h != d.expect_hash) status = R_FILE_HASH_MISMATCH;
        lane0_guard();
        if (lane == 0) {
            zpk_decode_result r; r.status = status; r.detail = 0; r.produced = len; r.hash = h;
            res[e] = r;
        }
    }
}

// one LZ4 entry, one wave.  retry_list != nullptr: a decode that ran out of its time budget is not reported — the entry goes on that
// list (counters[C_RETRY_LZ4]) and k_lz4_retry decodes it again behind the batch with ZPK_WATCHDOG_RETRY_SCALE times the budget.
// COOP: seq_exec.h — 0 in k_lz4_wave (the round-4 code, one cooperative piece at a time), 2 everywhere else (grouped cooperative copies).
template <int COOP>
__device__ __forceinline__ void lz4_entry_wave(Lz4WaveShared& shw, const u8* __restrict__ src, const u8* read_lo, const u8* read_hi,
                                               const zpk_decode_desc* __restrict__ desc, u8* dst, zpk_decode_result* __restrict__ res,
                                               u32 e, u32* __restrict__ counters, u64* __restrict__ dbg, u32* __restrict__ retry_list,
                                               u32 wd_scale, int lane)
{
    const zpk_decode_desc d = desc[e];
    const u8* in = uni_ptr(src + d.src_offset);
    u8* out = uni_ptr(dst + d.dst_offset);
    Watchdog wd; wd.arm(uni64(d.comp_size) + uni64(d.dst_capacity), wd_scale);
    SeqStats stt = {};
    const u64 t_all = SEQ_T(); (void)t_all;
    DecodeOut o = lz4f_decode_wave<COOP>(shw, wd, stt, in, uni64(d.comp_size), read_lo, read_hi, out, uni64(d.dst_capacity), lane);
#ifdef ZPK_STATS
    if (dbg && lane == 0) {
        u64* g = dbg + (u64)e * 8;
        g[0] = stt.t_parse; g[1] = stt.t_lit; g[2] = stt.t_dep; g[3] = stt.t_rounds;
        g[4] = ((u64)stt.batches << 32) | stt.rounds; g[5] = ((u64)stt.coops << 32) | stt.asm_batches; g[6] = SEQ_T() - t_all;
        g[7] = ((u64)stt.fix_iters << 32) | stt.chunks;
#ifdef ZPK_STATS_PARSE
        g[0] = stt.t_stage; g[1] = stt.t_walk1; g[2] = stt.t_fix; g[3] = stt.t_emit; g[4] = stt.t_tok;
        g[5] = ((u64)stt.hops_first << 32) | stt.hops_fix; g[6] = stt.slow_hops;
#endif
    }
#else
    (void)dbg; (void)t_all;
#endif
    if (wd.fired && retry_list) {                               // slow is not a verdict: again, later, with the large budget
        lane0_guard();
        if (lane == 0) retry_list[atomicAdd(&counters[C_RETRY_LZ4], 1u)] = e;
        return;
    }
    // lib/zpack_read.c:421-450
    int status = R_OK;
    if (o.rc == D_MALFORMED) status = R_DECOMPRESS_FAILED;
    else if (o.rc == D_TRUNCATED) status = o.produced < d.dst_capacity ? R_FILE_INCOMPLETE : R_BUFFER_TOO_SMALL;
    else if (o.rc == D_DST_FULL) status = R_BUFFER_TOO_SMALL;
    finish_entry(d, res, e, status, wd.fired ? 0xDEADu : (u32)(-o.rc), o.produced, out, lane);
}

// one wave per slot of the LZ4 work list: the hardware dispatcher is the load balancer
__global__ __launch_bounds__(64, 8) void k_lz4_wave(const u8* __restrict__ src, const u8* read_lo, const u8* read_hi,
                                                  const zpk_decode_desc* __restrict__ desc, u8* dst,
                                                  zpk_decode_result* __restrict__ res, const u32* __restrict__ list,
                                                  u32* __restrict__ counters, u64* __restrict__ dbg, u32* __restrict__ retry_list
#ifdef ZPK_DEVELOPER
                                                  , u32 wd_scale
#endif
                                                  )
{
#ifndef ZPK_DEVELOPER
    const u32 wd_scale = 1u;
#endif
    const int lane = lane_id();
    __shared__ Lz4WaveShared shw;
    u32 idx;
    if (my_slot(counters, L_LZ4, idx))
        lz4_entry_wave<0>(shw, src, read_lo, read_hi, desc, dst, res, uni(list[idx]), counters, dbg, retry_list, wd_scale, lane);
}

// the entries k_lz4_wave gave up on (normally none): a small grid, ZPK_WATCHDOG_RETRY_SCALE times the budget, and now the verdict counts
__global__ __launch_bounds__(64, 8) void k_lz4_retry(const u8* __restrict__ src, const u8* read_lo, const u8* read_hi,
                                                   const zpk_decode_desc* __restrict__ desc, u8* dst,
                                                   zpk_decode_result* __restrict__ res, const u32* __restrict__ list,
                                                   u32* __restrict__ counters, u64* __restrict__ dbg)
{
    const int lane = lane_id();
    __shared__ Lz4WaveShared shw;
    const u32 n_slots = uni(counters[C_RETRY_LZ4]);
    for (u32 idx = uni((u32)blockIdx.x); idx < n_slots; idx += gridDim.x)
        lz4_entry_wave<2>(shw, src, read_lo, read_hi, desc, dst, res, uni(list[idx]), counters, dbg, nullptr, (u32)ZPK_WATCHDOG_RETRY_SCALE, lane);
}

// k_lz4_left: the LZ4 entries that are mostly RUNS — k_classify puts an entry compressed to less than an eighth of its size on this
// list instead of k_lz4_wave's (byte runs, repeated blocks: matches longer than 32 bytes or feeding themselves; on the benchmark
// corpus exactly the `runs` class) — decoded by the same one-wave decoder built with the grouped cooperative copies (seq_exec.h
// COOP = 2): 614 -> ~2 000 GiB/s on such entries.  A persistent grid with an atomic dequeue, launched in FRONT of k_lz4_wave (a
// launch behind it would be a serial tail); it leaves at once when the list is empty (text, records: always).
#define LZ4_LEFT_GRID_MAX 8192u
#ifndef LZ4_LEFT_WAVES
#define LZ4_LEFT_WAVES 8
#endif
__global__ __launch_bounds__(64, LZ4_LEFT_WAVES) void k_lz4_left(const u8* __restrict__ src, const u8* read_lo, const u8* read_hi,
                                                  const zpk_decode_desc* __restrict__ desc, u8* dst,
                                                  zpk_decode_result* __restrict__ res, const u32* __restrict__ list,
                                                  u32* __restrict__ counters, u64* __restrict__ dbg, u32* __restrict__ retry_list, u32 wd_scale)
{
    const int lane = lane_id();
    __shared__ Lz4WaveShared shw;
    const u32 n_slots = uni(counters[C_LZ4_LEFT]);
    if (n_slots == 0) return;                      // (text, records: always — 8192 dequeues on one word are 0.1 ms by themselves)
    for (;;) {
        lane0_guard();
        u32 v = 0;
        if (lane == 0) v = atomicAdd(&counters[C_LZ4_LEFT_HEAD], 1u);
        const u32 idx = uni(v);
        lane0_guard();
        if (idx >= n_slots) break;
        lz4_entry_wave<2>(shw, src, read_lo, read_hi, desc, dst, res, uni(list[idx]), counters, dbg, retry_list, wd_scale, lane);
    }
}

// Stage 2 of the two-stage Zstandard path: entries whose sequences k_zstd_fse left in the arena (zstate == 1) are run
// here — Huffman literals, execution, XXH3 — by a kernel that carries neither the FSE decoder's code nor its tables:
// 9.4 KiB of LDS and <= 128 VGPRs, 16 workgroups per CU.  Only a COMPLETE entry (every frame decoded, exactly uncomp_size bytes) is finished here
// (result written, zstate = 2: status OK, or FILE_HASH_MISMATCH when its XXH3 differs — decoding it again could only find the same);
// everything else is left to k_zstd, so every verdict other than those two is always the full decoder's.
#define ZSTD_EXEC_GRID_MAX 4096
#ifndef ZSTD_EXEC_WAVES
#define ZSTD_EXEC_WAVES 4
#endif
// The execute stage runs through the LDS output ring (zstd_ring.h): aligned LDS accesses, near matches served from LDS, whole 1 KiB
// lines flushed with the XXH3 accumulators fed on the way out.  Against the direct executor (zstd_sequences_pre + seq_exec_batch,
// every sequence written to and gathered from HBM with exact-tail accesses, hash by re-reading; -DZSTD_EXEC_DIRECT keeps it) it moves
// 34 % fewer bytes in and 27 % fewer out of HBM at the same speed (profiles/r02: text, 8192 x 256 KiB: FETCH 13.8 -> 9.1 GB raw,
// WRITE 4.5 -> 3.3 GB; C3 62.4 vs 61.3 ms, C4 equal).
#ifndef ZSTD_EXEC_DIRECT
#define ZSTD_EXEC_RING 1
#endif
__global__ __launch_bounds__(ZSTD_WG_THREADS, ZSTD_EXEC_WAVES) void k_zstd_exec(const u8* __restrict__ src, const zpk_decode_desc* __restrict__ desc,
                                                               u8* dst, zpk_decode_result* __restrict__ res,
                                                               const u32* __restrict__ list, u32* __restrict__ counters,
                                                               u8* __restrict__ lit_scratch, const u64* __restrict__ arena,
                                                               u32* __restrict__ zstate, u32* __restrict__ leftover, u64* __restrict__ dbg)
{
    const int lane = lane_id();
    const u32 nz = uni(counters[L_ZSTD]);
    if (nz == 0) return;
#ifndef ZSTD_EXEC_RING
    __shared__ __attribute__((aligned(16))) u8 sh_raw[ZSTD_SHARED_EXEC_BYTES];
    ZstdShared& sh = *(ZstdShared*)sh_raw;
    if (threadIdx.x == 0) { sh.defaults_built = 0; sh.huf_valid = 0; }
#else
    __shared__ ZstdRingShared sh;
    if (threadIdx.x == 0) sh.huf_valid = 0;
#endif
    __syncthreads();
    u8* lit = lit_scratch + (u64)blockIdx.x * ZSTD_LIT_SCRATCH;
    for (;;) {
        lane0_guard();
        u32 v = 0;
        if (lane == 0) v = atomicAdd(&counters[C_EXEC_HEAD], 1u);
        const u32 idx = uni(v);
        lane0_guard();
        if (idx >= nz) break;
        const u32 e = uni(list[idx]);
        if (uni(zstate[e]) != 1u) {                     // not pre-decoded: straight to the full decoder's list
            lane0_guard();
            if (lane == 0) leftover[atomicAdd(&counters[C_LEFT_COUNT], 1u)] = e;
            lane0_guard();
            continue;
        }
        const zpk_decode_desc d = desc[e];
        const u8* in = uni_ptr(src + d.src_offset);
        u8* out = uni_ptr(dst + d.dst_offset);
        const u64* const pre = arena + (((u64)d.dst_offset + 7) >> 3);
        Watchdog wd; wd.arm(uni64(d.comp_size) + uni64(d.dst_capacity));
#ifndef ZSTD_EXEC_RING
#ifdef ZPK_STATS
        ZstdStats zs = {};
        const u64 t_all = SEQ_T();
        DecodeOut o = zstd_decode_wave<true>(sh, wd, in, uni64(d.comp_size), out, uni64(d.dst_capacity), lit, lane, &zs, pre);
        const u64 t_dec = SEQ_T();
#else
        (void)dbg;
        DecodeOut o = zstd_decode_wave<true>(sh, wd, in, uni64(d.comp_size), out, uni64(d.dst_capacity), lit, lane, nullptr, pre);
#endif
        bool ok = o.rc == D_OK;
        u64 h = 0;
        lane0_guard();
        if (ok) {
            wave_mem_fence();
    #ifndef ZPK_ABL_NOHASH       // (developer ablation: instruction counters without the hash pass)
        h = xxh3_64_wave(out, d.uncomp_size, lane);                            // lib/zpack_read.c:466
#endif
            ok = h == d.expect_hash || (d.flags & ZPK_DF_SKIP_HASH);
        }
        const int ok_status = R_OK;
#ifdef ZPK_STATS
        if (dbg && lane == 0) {
            u64* g = dbg + (u64)e * 16;
            g[0] = zs.t_lit; g[1] = zs.t_tab; g[2] = zs.t_fse; g[3] = zs.t_exec; g[4] = zs.nseq; g[5] = zs.nblk; g[6] = t_dec - t_all; g[7] = SEQ_T() - t_dec;
        }
#endif
#else
        (void)dbg;
        // through the LDS output ring (zstd_ring.h): the hash comes out of the flushes
        struct { int rc; u64 produced; } o;
        const LxResult xr = zstd_ring_decode_wave(sh, wd, in, uni64(d.comp_size), out, uni64(d.dst_capacity), uni64(d.uncomp_size), lit, pre, lane,
                                                   dbg ? dbg + (u64)e * 16 : nullptr);
        o.rc = xr.rc == LX_OK ? D_OK : -(0x100 + xr.rc); o.produced = xr.produced;
        const u64 h = xr.hash;
        // A complete, byte-exact decode is finished here whatever its checksum says: FILE_HASH_MISMATCH is the verdict of
        // lib/zpack_read.c:466-468 for exactly this case, and decoding the entry a second time in k_zstd would only reach it again
        // (the set of frames this path accepts equals the oracle's on 18 000 damaged frames: profiles/r02/r02_fuzz_ring_executor.log)
        const bool ok = xr.rc == LX_OK && xr.produced == uni64(d.uncomp_size);
        const int ok_status = (h == d.expect_hash || (d.flags & ZPK_DF_SKIP_HASH)) ? R_OK : R_FILE_HASH_MISMATCH;
#endif
        lane0_guard();
        if (lane == 0) {
            if (ok) {
                zpk_decode_result r; r.status = ok_status; r.detail = 0; r.produced = o.produced; r.hash = h;
                res[e] = r;
                zstate[e] = 2u;
                atomicAdd(&counters[C_ZSTD_TWO_STAGE], 1u);
            } else {
                leftover[atomicAdd(&counters[C_LEFT_COUNT], 1u)] = e;
                atomicAdd(&counters[14], 1u); counters[15] = ((u32)(-o.rc) & 0xFFFFu);
            }
        }
    }
}

// the full decoder: every Zstandard entry the two-stage path did not finish (all of them when it is off)
__global__ __launch_bounds__(ZSTD_WG_THREADS, 3) void k_zstd(const u8* __restrict__ src, const zpk_decode_desc* __restrict__ desc,
                                                          u8* dst, zpk_decode_result* __restrict__ res,
                                                          const u32* __restrict__ list, u32* __restrict__ counters,
                                                          u8* __restrict__ lit_scratch, u64* __restrict__ dbg, int count_word,
                                                          int head_word, u32* __restrict__ retry_list, int retry_word, u32 wd_scale)
{
    // `list` / counters[count_word]: the Zstandard work list itself, what k_zstd_exec left over, or (retry_list == nullptr) the
    // entries the first launch gave up on; counters[head_word]: the dequeue head of this launch
    const int lane = lane_id();
    const u32 nz = uni(counters[count_word]);
    if (nz == 0) return;
    __shared__ ZstdShared sh;
    if (threadIdx.x == 0) { sh.defaults_built = 0; sh.huf_valid = 0; }
    __syncthreads();
    u8* lit = lit_scratch + (u64)blockIdx.x * ZSTD_LIT_SCRATCH;
    for (;;) {
        lane0_guard();
        u32 v = 0;
        if (lane == 0) v = atomicAdd(&counters[head_word], 1u);
        const u32 idx = uni(v);
        lane0_guard();
        if (idx >= nz) break;
        const u32 e = uni(list[idx]);
        const zpk_decode_desc d = desc[e];
        const u8* in = uni_ptr(src + d.src_offset);
        u8* out = uni_ptr(dst + d.dst_offset);
        Watchdog wd; wd.arm(uni64(d.comp_size) + uni64(d.dst_capacity), wd_scale);
#ifdef ZPK_STATS
        ZstdStats zs = {};
        const u64 t_all = SEQ_T();
        DecodeOut o = zstd_decode_wave<false>(sh, wd, in, uni64(d.comp_size), out, uni64(d.dst_capacity), lit, lane, &zs);
        if (dbg && lane == 0) {
            u64* g = dbg + (u64)e * 8;
            g[0] = zs.t_lit; g[1] = zs.t_tab; g[2] = zs.t_fse; g[3] = zs.t_exec; g[4] = zs.nseq; g[5] = zs.nblk; g[6] = SEQ_T() - t_all; g[7] = 0;
        }
#else
        (void)dbg;
        DecodeOut o = zstd_decode_wave<false>(sh, wd, in, uni64(d.comp_size), out, uni64(d.dst_capacity), lit, lane);
#endif
        int status = o.rc == D_OK ? R_OK : R_DECOMPRESS_FAILED;               // lib/zpack_read.c:384-388
        if (wd.fired && retry_list) {                               // slow is not a verdict: again, later, with the large budget
            lane0_guard();
            if (lane == 0) retry_list[atomicAdd(&counters[retry_word], 1u)] = e;
            lane0_guard();
            continue;
        }
        finish_entry(d, res, e, status, wd.fired ? 0xDEADu : (u32)(-o.rc), o.produced, out, lane);
        lane0_guard();
        if (lane == 0) atomicAdd(&counters[C_ZSTD_FUSED], 1u);
    }
}

__global__ __launch_bounds__(256) void k_hash(const u8* __restrict__ src, const u64* __restrict__ offsets,
                                              const u64* __restrict__ sizes, u64 n, u64* __restrict__ hashes)
{
    const int lane = lane_id();
    const u64 w = uni64(((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w < n) {
        u64 h = xxh3_64_wave(uni_ptr(src + offsets[w]), uni64(sizes[w]), lane);
        lane0_guard();
        if (lane == 0) hashes[w] = h;
    }
}

// ------------------------------------------------------------------------------------ host side

#define ZPK_DEC_SPLIT_MIN_DEFAULT (256ull << 10)     // (round 5: a single 512 KiB LZ4 entry is 0.66 ms block-parallel against 3.3 ms by one wave, 1 MiB of Zstandard 4.3 against 31.7: tools/mid_entry_rate.py)
#define ZPK_ENC_SPLIT_MIN_DEFAULT (2ull << 20)
#define ZPK_ENC_PIECE (512u << 10)                 // = ZPK_CS_PIECE of the streaming writer
#ifndef ZPK_PJ_CHUNK_BLOCKS
#define ZPK_PJ_CHUNK_BLOCKS 512u                   // lz4_pj.h: blocks per chunk = 32 MiB of output, 128 MiB of byte references (the Infinity Cache holds 256)
#endif
#define ZPK_PJ_MAX_CHUNKS 64u
struct zpk_codec {
    int device = 0;
    hipStream_t stream = nullptr;
    // One codec = one in-flight batch: the work lists, counters and staging buffers below are shared by every call.
    // The host-pointer entry points (decode/encode_batch_host, hash_host, the streaming triple) take `mu` for their
    // whole duration, so threads that share a codec are serialised, never corrupted.  The device-pointer entry points
    // take it only while they enqueue: batches on ONE stream are ordered by the stream; a codec must not be driven
    // from two streams at once (create one codec per stream — contexts are cheap).
    pthread_mutex_t mu;
    u32* d_counters = nullptr;
    u32* d_lists = nullptr;      u64 list_cap = 0;
    u8*  d_lit = nullptr;        u64 lit_cap = 0;
    // host-API staging
    u8*  d_src = nullptr;        u64 src_cap = 0;
    u8*  h_pin[2] = {nullptr, nullptr};          // pinned staging of the host-pointer paths (ZPK_PIN_CHUNK bytes each), created on first use
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    hipStream_t s_up = nullptr, s_dn = nullptr;  // host-pointer decode pipeline: upload / download streams beside `stream` (created on first use)
    hipStream_t s_side = nullptr;                // decode batches: the LZ4 kernel beside the Zstandard stages (low priority, created on first use)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void* d_pj_blocks = nullptr; u64 pj_blocks_cap = 0; // large single LZ4 frames (lz4_pj.h): block table, sequence records, start masks, byte references, flags
    void* d_pj_recs = nullptr;   u64 pj_recs_cap = 0;
    void* d_pj_masks = nullptr;  u64 pj_masks_cap = 0;
    void* d_pj_S = nullptr;      u64 pj_S_cap = 0;
    u32*  d_pj_flags = nullptr;
    u8*   h_bigsrc = nullptr; u64 h_bigsrc_cap = 0;  // pinned: the compressed bytes of one large device-resident entry, for the host's block walk (zpk_codec_decode_big_device)
    u8*   d_big1 = nullptr;                          // device: one descriptor + one result (the same call's one-wave fallback)
    u32   zpj_last_err = 0;                          // developer: the flag word of the most recent large Zstandard frame (why it went to the one-wave decoder)
    void* d_zpj_blocks = nullptr; u64 zpj_blocks_cap = 0; // large single Zstandard frames (zstd_pj.h): block table; work items, states, final histories; sequence positions
    void* d_zpj_aux = nullptr;    u64 zpj_aux_cap = 0;
    void* d_zpj_pos = nullptr;    u64 zpj_pos_cap = 0;
    hipStream_t s_left = nullptr;                // decode batches: k_lz4_left (the LZ4 entries that are mostly runs) beside k_lz4_wave
    hipEvent_t ev_lfork = nullptr, ev_ljoin = nullptr;
    volatile u32* h_seen = nullptr;              // pinned: the work-list counts of an earlier device batch (what the next one probably holds)
    hipEvent_t pipe_ev[2 * 64] = {};             // per piece: uploaded, decoded
    u8*  d_dst = nullptr;        u64 dst_cap = 0;
    void* d_desc = nullptr;      u64 desc_cap = 0;
    void* d_res = nullptr;       u64 res_cap = 0;
    u64* d_dbg = nullptr;        u64 dbg_cap = 0;
    u64* d_seq = nullptr;        u64 seq_cap = 0;      // encoder: sequence lists, one per workgroup
    u8*  d_pack = nullptr;       u64 pack_cap = 0;     // K7: block sums + span index of the compaction
    u8*  d_packed = nullptr;     u64 packed_cap = 0;   // host encode path: packed payload stream
    u8*  d_packoff = nullptr;    u64 packoff_cap = 0;  // host encode path: payload offsets
    u8*  d_xpart = nullptr;      u64 xpart_cap = 0;    // host encode path, split entries: span list | 64 bytes of XXH3 partial sums per 1 KiB block | hashes
    u64  enc_order_min = 4608;                         // ... and encode batches their ticket queue (the encoder's resident waves: 18 per CU)
    int  order_fast_last = 1;                          // ZPK_OPT_ORDER_FAST_LAST: a batch of one size class runs its incompressible entries last
    u64  order_min = 8192;                             // ZPK_OPT_ORDER_MIN: decode batches of at least this many entries run their work lists largest entries first
    u64  dec_split_min = ZPK_DEC_SPLIT_MIN_DEFAULT;    // ZPK_OPT_DEC_SPLIT_MIN: entries of at least this many bytes that ARE sequences of frames are decoded frame-parallel
    hipEvent_t pj_ev[ZPK_PJ_MAX_CHUNKS] = {};           // lz4_pj.h: one event behind every chunk of a large LZ4 frame (its bytes may go home)
    u32  big_last[2] = {0, 0};                         // host decode path, most recent call: entries decoded frame-parallel, their frames
    u64  enc_split_min = ZPK_ENC_SPLIT_MIN_DEFAULT;    // ZPK_OPT_ENC_SPLIT_MIN: entries of at least this many bytes are written as a sequence of frames
    u64* d_zarena = nullptr;     u64 zarena_cap = 0;   // decoder: pre-decoded Zstandard sequences, laid out like dst (zstd_fse4.h)
    u32* d_zstate = nullptr;     u64 zstate_cap = 0;   // decoder: per entry, 1 = its sequences are in the arena
    int lz4_hint = -1;           // host path: does the batch hold an LZ4 entry?  -1 = unknown (device path)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t kev[ZPK_K_COUNT][2] = {};
    int profiling = 0;
    int zstd_hint = -1;          // host path: does the batch hold a Zstandard entry (1 / 0)?  -1 = unknown (device path)
    int fell_back_fused = 0;     // the last decode batch could not get its sequence arena and ran the fused decoder only
    // the host-path pipeline decodes one call in several launches: their counters are brought back piece by piece and summed, so that
    // decode_stats / decode_stats2 describe the whole call (a retry or watchdog event in an early piece is not lost)
    u64* h_pj = nullptr;                               // 8 pinned words (pin_ready): hash and flags of a large LZ4 frame come home without blocking the launcher
    u32 (*piece_counters)[N_COUNTERS] = nullptr;     // [64], pinned (pin_ready): a D2H copy into pageable memory would block the launcher thread per piece
    u32 host_totals[N_COUNTERS] = {};
    int totals_valid = 0;
    char err[256] = {0};
};

struct CodecLock {
    pthread_mutex_t* m;
    explicit CodecLock(zpk_codec* c) : m(&c->mu) { pthread_mutex_lock(m); }
    ~CodecLock() { pthread_mutex_unlock(m); }
    CodecLock(const CodecLock&) = delete;
    CodecLock& operator=(const CodecLock&) = delete;
};

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    snprintf((c)->err, sizeof((c)->err), "%s: %s", #call, hipGetErrorString(e_)); return ZPK_E_LAUNCH; } } while (0)

static int grow(zpk_codec* c, void** p, u64* cap, u64 need)
{
    if (need <= *cap) return ZPK_OK;
    // the buffer may still be in use by work enqueued earlier on the codec's stream or the caller's
    if (*p) { (void)hipDeviceSynchronize(); (void)hipFree(*p); *p = nullptr; *cap = 0; }
    u64 want = need + need / 4 + 4096;
    if (hipMalloc(p, want) != hipSuccess) {
        (void)hipGetLastError();
        want = need + 256;                                                          // the slack was a convenience, not a need
        if (hipMalloc(p, want) != hipSuccess) {
            (void)hipGetLastError();
            snprintf(c->err, sizeof(c->err), "hipMalloc(%llu) failed", (unsigned long long)want); return ZPK_E_NOMEM;
        }
    }
    *cap = want;
    return ZPK_OK;
}

extern "C" {

int zpk_codec_abi_version(void) { return ZPK_CODEC_ABI_VERSION; }

int zpk_codec_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int zpk_codec_create(zpk_codec** out, int device)
{
    if (!out) return ZPK_E_INVALID;
    *out = nullptr;
    int n = zpk_codec_device_count();
    if (n <= 0) return ZPK_E_NO_DEVICE;
    if (device < 0) {
        const char* e = getenv("ZPACK_AMD_DEVICE");
        if (!e) e = getenv("LOCAL_RANK");
        device = e ? atoi(e) : 0;
        if (device < 0 || device >= n) device = 0;
    }
    if (device >= n) return ZPK_E_INVALID;
    zpk_codec* c = new (std::nothrow) zpk_codec();
    if (!c) return ZPK_E_NOMEM;
    c->device = device;
    {   // recursive: the host entry points call the device ones underneath
        pthread_mutexattr_t at; pthread_mutexattr_init(&at); pthread_mutexattr_settype(&at, PTHREAD_MUTEX_RECURSIVE);
        pthread_mutex_init(&c->mu, &at); pthread_mutexattr_destroy(&at);
    }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&c->d_counters, N_COUNTERS * sizeof(u32)) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        zpk_codec_destroy(c);
        return ZPK_E_NO_DEVICE;
    }
    *out = c;
    return ZPK_OK;
}

void zpk_codec_destroy(zpk_codec* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    (void)hipFree(c->d_counters); (void)hipFree(c->d_lists); (void)hipFree(c->d_lit);
    (void)hipFree(c->d_src); (void)hipFree(c->d_dst); (void)hipFree(c->d_desc); (void)hipFree(c->d_res);
    for (int k = 0; k < 2; k++) { if (c->h_pin[k]) (void)hipHostFree(c->h_pin[k]); if (c->pin_ev[k]) (void)hipEventDestroy(c->pin_ev[k]); }
    if (c->piece_counters) (void)hipHostFree(c->piece_counters);
    if (c->h_pj) (void)hipHostFree(c->h_pj);
    if (c->h_seen) (void)hipHostFree((void*)c->h_seen);
    if (c->s_side) (void)hipStreamDestroy(c->s_side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (u32 k = 0; k < ZPK_PJ_MAX_CHUNKS; k++) if (c->pj_ev[k]) (void)hipEventDestroy(c->pj_ev[k]);
    (void)hipFree(c->d_zpj_blocks); (void)hipFree(c->d_zpj_aux); (void)hipFree(c->d_zpj_pos); (void)hipFree(c->d_big1);
    if (c->h_bigsrc) (void)hipHostFree(c->h_bigsrc);
    (void)hipFree(c->d_pj_blocks); (void)hipFree(c->d_pj_recs); (void)hipFree(c->d_pj_masks); (void)hipFree(c->d_pj_S); (void)hipFree(c->d_pj_flags);
    if (c->s_left) (void)hipStreamDestroy(c->s_left);
    if (c->ev_lfork) (void)hipEventDestroy(c->ev_lfork);
    if (c->ev_ljoin) (void)hipEventDestroy(c->ev_ljoin);
    if (c->s_up) (void)hipStreamDestroy(c->s_up);
    if (c->s_dn) (void)hipStreamDestroy(c->s_dn);
    for (int k = 0; k < 2 * 64; k++) if (c->pipe_ev[k]) (void)hipEventDestroy(c->pipe_ev[k]);
    (void)hipFree(c->d_dbg); (void)hipFree(c->d_seq); (void)hipFree(c->d_zarena); (void)hipFree(c->d_zstate); (void)hipFree(c->d_pack); (void)hipFree(c->d_packed); (void)hipFree(c->d_packoff); (void)hipFree(c->d_xpart);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (int i = 0; i < ZPK_K_COUNT; i++) for (int j = 0; j < 2; j++) if (c->kev[i][j]) (void)hipEventDestroy(c->kev[i][j]);
    pthread_mutex_destroy(&c->mu);
    delete c;
}

// after an abandoned stream / a failed call (the reference resets its library context: lib/zpack_read.c:679-690): wait for
// whatever the codec still has in flight and forget the last error; every later call starts from a clean context
void zpk_codec_reset(zpk_codec* c)
{
    if (!c) return;
    CodecLock lk(c);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
    c->err[0] = 0;
    // a context keeps its grown staging between batches (the next batch of that size starts at once); a reset gives the large pieces
    // back — a process that holds many readers can bound what each one retains (zpack_reset_reader_dctx / zpack_reset_writer_cctx)
    const u64 keep = 64ull << 20;
    void** bufs[] = { (void**)&c->d_src, (void**)&c->d_dst, (void**)&c->d_zarena, (void**)&c->d_lit };
    u64* caps[] = { &c->src_cap, &c->dst_cap, &c->zarena_cap, &c->lit_cap };
    for (int i = 0; i < 5; i++) if (*bufs[i] && *caps[i] > keep) { (void)hipFree(*bufs[i]); *bufs[i] = nullptr; *caps[i] = 0; }
}
const char* zpk_codec_last_error(const zpk_codec* c) { return c ? c->err : "no codec"; }
int zpk_codec_device(const zpk_codec* c) { return c ? c->device : -1; }

// Developer hooks (ZPK_TRACE / ZPK_SKIP / ZPK_DEBUG_TIMING environment switches) exist only in a -DZPK_DEVELOPER build:
// the product launch path reads no environment and can neither drop a kernel nor end the host process.
#ifdef ZPK_DEVELOPER
#define ZPK_DEV(x) x
#define ZPK_WD_ARG , wd_scale
#else
#define ZPK_DEV(x)
#define ZPK_WD_ARG
#endif

// XXH3-64 of `nspans` long spans of `base` (xxh3_span.h): enqueues the two kernels and the copy of the hashes to `h_hash`
static int xxh3_spans_launch(zpk_codec* c, const u8* base, const zpk_span* h_spans, u64 nspans, u64 part_blocks, u64* h_hash, hipStream_t st)
{
    if (nspans == 0) return ZPK_OK;
    if (nspans > 0x7FFFFFFFull) return ZPK_E_INVALID;
    const u64 span_bytes = (nspans * sizeof(zpk_span) + 255) & ~255ull, part_bytes = part_blocks * 64;
    int rc;
    if ((rc = grow(c, (void**)&c->d_xpart, &c->xpart_cap, span_bytes + part_bytes + nspans * 8 + 64))) return rc;
    zpk_span* d_spans = (zpk_span*)c->d_xpart;
    u64* d_part = (u64*)(c->d_xpart + span_bytes);
    u64* d_hash = (u64*)(c->d_xpart + span_bytes + part_bytes);
    HIPCHK(c, hipMemcpyAsync(d_spans, h_spans, nspans * sizeof(zpk_span), hipMemcpyHostToDevice, st));
    const u64 ngroups = part_blocks / XS_GROUP;
    if (ngroups) hipLaunchKernelGGL(k_xxh3_partials, dim3((u32)((ngroups + 3) / 4)), dim3(256), 0, st, base, (const zpk_span*)d_spans, (u32)nspans, (u64)0, ngroups, d_part);
    hipLaunchKernelGGL(k_xxh3_chain, dim3((u32)nspans), dim3(64), 0, st, base, (const zpk_span*)d_spans, (const u64*)d_part, d_hash, (u64*)nullptr, (u64)0, ~(u64)0, 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h_hash, d_hash, nspans * 8, hipMemcpyDeviceToHost, st));
    return ZPK_OK;
}
static inline u64 xxh3_span_blocks(u64 len) { return (((len - 1) >> 10) + XS_GROUP - 1) / XS_GROUP * XS_GROUP; }   // partial-sum slots of one span

static int decode_launch(zpk_codec* c, const u8* src, u64 src_size, const u8* read_lo, const u8* read_hi,
                         const zpk_decode_desc* desc, u64 n, u8* dst, u64 dst_size, zpk_decode_result* res, hipStream_t st)
{
    if (n == 0) return ZPK_OK;
    if (n > 0x7FFFFFF0ull) return ZPK_E_INVALID;               // one workgroup per LZ4 entry: the grid's x limit
    c->totals_valid = 0;                                        // (the pipeline sets it again once it has summed its pieces)
    int rc;
    if ((rc = grow(c, (void**)&c->d_lists, &c->list_cap, N_LISTS_ALLOC * n * sizeof(u32)))) return rc;
    const u64 stride = c->list_cap / (N_LISTS_ALLOC * sizeof(u32));
    u32* const retry_lz4 = c->d_lists + (u64)N_LISTS * stride;
    u32* const retry_zstd = c->d_lists + (u64)(N_LISTS + 1) * stride;
    u32* const left_lz4 = c->d_lists + (u64)(N_LISTS + 2) * stride;
    u32 wd_scale = 1; (void)wd_scale;
    ZPK_DEV(static const int wd_env = getenv("ZPK_WD_SCALE") ? atoi(getenv("ZPK_WD_SCALE")) : 1; wd_scale = (u32)wd_env;)
    int skip = 0; (void)skip;
#ifdef ZPK_DEVELOPER
    static const int want_dbg = getenv("ZPK_DEBUG_TIMING") ? atoi(getenv("ZPK_DEBUG_TIMING")) : 0;
    if (want_dbg) { if ((rc = grow(c, (void**)&c->d_dbg, &c->dbg_cap, n * 128))) return rc; }
    static const int trace = getenv("ZPK_TRACE") ? atoi(getenv("ZPK_TRACE")) : 0;
    static const int skip_env = getenv("ZPK_SKIP") ? atoi(getenv("ZPK_SKIP")) : 0;      // bitmask: 1 stored, 2 lz4, 4 zstd
    skip = skip_env;
#define ZPK_TRACE_STEP(name) do { if (trace == 1) { hipError_t te_ = hipStreamSynchronize(st); \
        fprintf(stderr, "[zpk] %s done: %s\n", name, hipGetErrorString(te_)); fflush(stderr); } } while (0)
#else
#define ZPK_TRACE_STEP(name) do { } while (0)
#endif
    const u32 zstd_grid = (u32)(n < ZSTD_GRID_MAX ? n : ZSTD_GRID_MAX);
    const u32 exec_grid = (u32)(n < ZSTD_EXEC_GRID_MAX ? n : ZSTD_EXEC_GRID_MAX);
    const bool maybe_zstd = c->zstd_hint != 0;                  // the host path knows its methods; device batches may hold any
    if (maybe_zstd && (rc = grow(c, (void**)&c->d_lit, &c->lit_cap, (u64)(exec_grid > zstd_grid ? exec_grid : zstd_grid) * ZSTD_LIT_SCRATCH))) return rc;
    HIPCHK(c, hipMemsetAsync(c->d_counters, 0, N_COUNTERS * sizeof(u32), st));
    ZPK_TRACE_STEP("memset");
#define ZPK_KEV(k, j) do { if (c->profiling) (void)hipEventRecord(c->kev[k][j], st); } while (0)
    ZPK_KEV(ZPK_K_CLASSIFY, 0);
    hipLaunchKernelGGL(k_classify, dim3((u32)((n + 255) / 256)), dim3(256), 0, st, desc, n, src_size, dst_size, res,
                       c->d_lists, stride, c->d_counters);
    // largest entries first (see k_order_count); batches that fit the resident waves in one round have nothing to order
    const u32* zstd_list = c->d_lists + L_ZSTD * stride;
    const u32* lz4_list = c->d_lists + L_LZ4 * stride;
    if (n >= c->order_min) {
        u32* const ordered = c->d_lists + (u64)(N_LISTS + 3) * stride;
        const dim3 og((u32)((n + 255) / 256), 2);
        hipLaunchKernelGGL(k_order_count, og, dim3(256), 0, st, desc, (const u32*)c->d_lists, stride, c->d_counters, c->order_fast_last);
        hipLaunchKernelGGL(k_order_fill, og, dim3(256), 0, st, desc, (const u32*)c->d_lists, stride, ordered, c->d_counters, c->order_fast_last);
        zstd_list = ordered; lz4_list = ordered + stride;
    }
    ZPK_KEV(ZPK_K_CLASSIFY, 1);
    ZPK_TRACE_STEP("k_classify");
    const u32 wgrid = (u32)((n + 3) / 4);          // one wave per list slot
    ZPK_KEV(ZPK_K_STORED, 0);
    if (!(skip & 1)) hipLaunchKernelGGL(k_stored, dim3(wgrid), dim3(256), 0, st, src, desc, dst, res, c->d_lists + L_NONE * stride, c->d_counters);
    ZPK_KEV(ZPK_K_STORED, 1);
    ZPK_TRACE_STEP("k_stored");
    // LZ4: one wave per work-list slot (lz4_wave.h).  A batch that may hold both methods runs the LZ4 kernels on a SIDE stream of
    // low priority, beside the Zstandard stages: those are bound by the latency of their serial chains and by LDS capacity (12 or
    // 16 workgroups per CU leave 3-8 KiB of LDS and most of the vector issue slots idle), so LZ4 waves fill what they leave.
    const bool maybe_lz4 = c->lz4_hint != 0;
    hipStream_t sl = st;
#ifndef ZPK_NO_SIDE_STREAM
    // (measured, 125 000 mixed entries: 112.0 -> 90.9 ms per batch, the LZ4 kernel's 25 ms disappear inside the Zstandard stages, which
    // get 1-3 ms longer; a pure LZ4 batch — the Zstandard kernels find empty lists — is unchanged within noise: 618.4 vs 618.2 GiB/s.)
    // A device batch does not say which methods it holds, and a pure Zstandard batch must NOT take the side stream: k_lz4_wave is one
    // workgroup per entry, and 100 000 EMPTY workgroups trickling through at low priority beside the pre-decode stage cost it 15 ms
    // (71.9 -> 87.6 ms).  So the codec looks at the work-list counts of the batch BEFORE (copied to pinned memory behind every batch,
    // no synchronisation): both methods there, or nothing known yet -> side stream; a codec fed batches of one method stays on one stream.
    if (!c->h_seen && hipHostMalloc((void**)&c->h_seen, 64, hipHostMallocDefault) == hipSuccess) { c->h_seen[L_NONE] = 0; c->h_seen[L_ZSTD] = 1; c->h_seen[L_LZ4] = 1; c->h_seen[8] = 1; }
    const bool both_seen = c->zstd_hint >= 0 /* the host path knows */ || (c->h_seen && c->h_seen[L_ZSTD] != 0 && c->h_seen[L_LZ4] != 0);
    if (maybe_lz4 && maybe_zstd && both_seen && !(skip & 6)) {
        int lo_prio = 0, hi_prio = 0;
        if (!c->s_side) { (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
                          if (hipStreamCreateWithPriority(&c->s_side, hipStreamNonBlocking, lo_prio) != hipSuccess) c->s_side = nullptr; }
        if (!c->ev_fork && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) c->ev_fork = nullptr;
        if (!c->ev_join && hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) c->ev_join = nullptr;
        if (c->s_side && c->ev_fork && c->ev_join && hipEventRecord(c->ev_fork, st) == hipSuccess &&
            hipStreamWaitEvent(c->s_side, c->ev_fork, 0) == hipSuccess) sl = c->s_side;
    }
#endif
    auto launch_lz4 = [&]() {
        if (c->profiling) (void)hipEventRecord(c->kev[ZPK_K_LZ4][0], sl);
        if (!(skip & 2) && maybe_lz4) {
            // The entries that are mostly runs (k_classify's second LZ4 list; none on text) by the build with the grouped cooperative copies,
            // BESIDE k_lz4_wave on a stream of its own: in front of it or behind it the few thousand of them were a serial stretch of one
            // entry's latency (0.4 ms of a 10 ms batch) with the chip nearly idle.
            hipStream_t sx = sl;
            // (a small batch is latency, not throughput: no second stream; neither when the batch BEFORE had no such entry — then the launch
            // is an empty grid in front of k_lz4_wave, and a batch that does have some pays the serial stretch once)
            if (n >= 4096 && c->h_seen && c->h_seen[8] != 0) {
            if (!c->s_left && hipStreamCreateWithFlags(&c->s_left, hipStreamNonBlocking) != hipSuccess) c->s_left = nullptr;
            if (!c->ev_lfork && hipEventCreateWithFlags(&c->ev_lfork, hipEventDisableTiming) != hipSuccess) c->ev_lfork = nullptr;
            if (!c->ev_ljoin && hipEventCreateWithFlags(&c->ev_ljoin, hipEventDisableTiming) != hipSuccess) c->ev_ljoin = nullptr;
            if (c->s_left && c->ev_lfork && c->ev_ljoin && hipEventRecord(c->ev_lfork, sl) == hipSuccess &&
                hipStreamWaitEvent(c->s_left, c->ev_lfork, 0) == hipSuccess) sx = c->s_left;
            }
            hipLaunchKernelGGL(k_lz4_left, dim3((u32)(n < LZ4_LEFT_GRID_MAX ? n : LZ4_LEFT_GRID_MAX)), dim3(64), 0, sx, src, read_lo, read_hi, desc, dst, res,
                               (const u32*)left_lz4, c->d_counters, c->d_dbg, retry_lz4, wd_scale);
            hipLaunchKernelGGL(k_lz4_wave, dim3((u32)n), dim3(64), 0, sl, src, read_lo, read_hi, desc, dst, res,
                               lz4_list, c->d_counters, c->d_dbg, retry_lz4 ZPK_WD_ARG);
            if (sx != sl && (hipEventRecord(c->ev_ljoin, sx) != hipSuccess || hipStreamWaitEvent(sl, c->ev_ljoin, 0) != hipSuccess))
                (void)hipStreamSynchronize(sx);
            // entries whose decoder ran out of its time budget: again, with ZPK_WATCHDOG_RETRY_SCALE times the budget (a small grid
            // that leaves at once when the list is empty — the normal case)
            hipLaunchKernelGGL(k_lz4_retry, dim3((u32)(n < 256 ? n : 256)), dim3(64), 0, sl, src, read_lo, read_hi, desc, dst, res,
                               (const u32*)retry_lz4, c->d_counters, c->d_dbg);
        }
        if (c->profiling) (void)hipEventRecord(c->kev[ZPK_K_LZ4][1], sl);
    };
    if (sl == st) launch_lz4();                    // (side stream: enqueued BEHIND the Zstandard stages below, so that those are dispatched first)
    ZPK_TRACE_STEP("k_lz4_wave");
    // Zstandard in two stages: the FSE sequence streams four per wave into an arena laid out like dst (8 bytes per
    // sequence: room for one sequence per 8 output bytes; entries that need more stay with the fused decoder), then
    // literals + execution + checksum.  Without the arena (allocation refused) k_zstd does it all, and the codec says so
    // (zpk_codec_decode_stats out[7] bit 31, zpk_codec_last_error).
    bool two_stage = maybe_zstd && dst_size >= 64;
    ZPK_DEV(static const int fused_only = getenv("ZPK_ZSTD_FUSED") ? atoi(getenv("ZPK_ZSTD_FUSED")) : 0; if (fused_only) two_stage = false;)
    c->fell_back_fused = 0;
    if (two_stage && (grow(c, (void**)&c->d_zarena, &c->zarena_cap, dst_size + 64) != ZPK_OK ||
                      grow(c, (void**)&c->d_zstate, &c->zstate_cap, 2 * n * sizeof(u32)) != ZPK_OK)) {
        two_stage = false; c->fell_back_fused = 1;
        snprintf(c->err, sizeof(c->err), "note: no memory for the %llu-byte Zstandard sequence arena; this batch ran the fused decoder",
                 (unsigned long long)dst_size + 64);
    }
    u32* const leftover = two_stage ? c->d_zstate + n : nullptr;          // entries k_zstd_exec hands to the full decoder
    ZPK_KEV(ZPK_K_ZSTD_FSE, 0);
    if (!(skip & 4) && two_stage) {
        HIPCHK(c, hipMemsetAsync(c->d_zstate, 0, n * sizeof(u32), st));          // entries k_zstd_fse never reaches stay unmarked
        const u64 zwaves = (n + ZF_ROWS - 1) / ZF_ROWS;
        hipLaunchKernelGGL(k_zstd_fse, dim3((u32)(zwaves < ZF_GRID_MAX ? zwaves : ZF_GRID_MAX)), dim3(64), 0, st, src, desc,
                           zstd_list, c->d_counters, c->d_zarena, c->d_zstate);
    }
    ZPK_KEV(ZPK_K_ZSTD_FSE, 1);
    ZPK_TRACE_STEP("k_zstd_fse");
    ZPK_KEV(ZPK_K_ZSTD, 0);
    if (!(skip & 4) && two_stage)
        hipLaunchKernelGGL(k_zstd_exec, dim3(exec_grid), dim3(ZSTD_WG_THREADS), 0, st, src, desc, dst, res,
                           zstd_list, c->d_counters, c->d_lit, c->d_zarena, c->d_zstate, leftover, c->d_dbg);
    if (!(skip & 4) && maybe_zstd)
        hipLaunchKernelGGL(k_zstd, dim3(zstd_grid), dim3(ZSTD_WG_THREADS), 0, st, src, desc, dst, res,
                       two_stage ? (const u32*)leftover : zstd_list, c->d_counters, c->d_lit, c->d_dbg,
                       two_stage ? (int)C_LEFT_COUNT : (int)L_ZSTD, (int)(L_COUNT + L_ZSTD), retry_zstd, (int)C_RETRY_ZSTD, wd_scale);
    ZPK_KEV(ZPK_K_ZSTD, 1);
    ZPK_TRACE_STEP("k_zstd");
    // Entries whose decoder ran out of its time budget (a preempted or contended GPU, not the entry's fault) are decoded again
    // here, behind the stages of their method, with ZPK_WATCHDOG_RETRY_SCALE times the budget: small grids that leave at once
    // when their list is empty (the normal case).
    if (!(skip & 4) && maybe_zstd)
        hipLaunchKernelGGL(k_zstd, dim3(zstd_grid < 128 ? zstd_grid : 128), dim3(ZSTD_WG_THREADS), 0, st, src, desc, dst, res,
                           (const u32*)retry_zstd, c->d_counters, c->d_lit, c->d_dbg, (int)C_RETRY_ZSTD, (int)C_RETRY_HEAD,
                           (u32*)nullptr, 0, (u32)ZPK_WATCHDOG_RETRY_SCALE);
    if (sl != st) {                                 // the LZ4 kernels, beside the above; the batch is done when both streams are
        launch_lz4();
        if (hipEventRecord(c->ev_join, sl) != hipSuccess || hipStreamWaitEvent(st, c->ev_join, 0) != hipSuccess) {
            (void)hipStreamSynchronize(sl);
        }
    }
    ZPK_TRACE_STEP("retry");
    if (c->h_seen && c->zstd_hint < 0) (void)hipMemcpyAsync((void*)c->h_seen, c->d_counters, N_LISTS * sizeof(u32), hipMemcpyDeviceToHost, st);
    if (c->h_seen) (void)hipMemcpyAsync((void*)(c->h_seen + 8), c->d_counters + C_LZ4_LEFT, sizeof(u32), hipMemcpyDeviceToHost, st);      // did this batch have LZ4 entries of runs?
    HIPCHK(c, hipGetLastError());
    return ZPK_OK;
}

int zpk_codec_decode_batch_device(zpk_codec* c, const uint8_t* src, uint64_t src_size, const zpk_decode_desc* desc, uint64_t n,
                                  uint8_t* dst, uint64_t dst_size, zpk_decode_result* results, void* stream)
{
    if (!c || (n && (!desc || !results))) return ZPK_E_INVALID;
    CodecLock lk(c);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    c->zstd_hint = -1; c->lz4_hint = -1;
    return decode_launch(c, src, src_size, src, src + src_size, desc, n, dst, dst_size, results, st);
}

// An entry whose frames end before uncomp_size bytes exist (a short decode is not an error of the libraries): lib/zpack_read.c:466
// hashes buffer[0, uncomp_size) all the same — the decoded bytes followed by whatever the CALLER'S buffer held.  The device slot holds
// something else there (an earlier batch's output), so for these rare entries the caller's bytes are brought up behind the decoded
// ones and the slot is hashed again: the verdict is the one the reference reaches on this caller's buffer.
static int rehash_short_entries(zpk_codec* c, const zpk_decode_desc* hd, const zpk_decode_desc* desc, u64 n, uint8_t* const* dst_ptrs,
                                zpk_decode_result* results)
{
    for (u64 i = 0; i < n; i++) {
        zpk_decode_result& r = results[i];
        if ((r.status != 0 && r.status != 15) || r.produced >= desc[i].uncomp_size || desc[i].uncomp_size > desc[i].dst_capacity ||
            desc[i].comp_size == 0 /* :328: OK before anything is read */ || desc[i].method == ZPK_METHOD_NONE) continue;
        int rc;
        if ((rc = grow(c, (void**)&c->d_xpart, &c->xpart_cap, 64))) return rc;
        const u64 tail = desc[i].uncomp_size - r.produced;
        u64 meta[3] = { hd[i].dst_offset, desc[i].uncomp_size, 0 };
        HIPCHK(c, hipMemcpyAsync(c->d_dst + hd[i].dst_offset + r.produced, dst_ptrs[i] + r.produced, tail, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_xpart, meta, sizeof(meta), hipMemcpyHostToDevice, c->stream));
        u64* m = (u64*)c->d_xpart;
        hipLaunchKernelGGL(k_hash, dim3(1), dim3(64), 0, c->stream, (const u8*)c->d_dst, (const u64*)m, (const u64*)(m + 1), (u64)1, m + 2);
        u64 h = 0;
        HIPCHK(c, hipMemcpyAsync(&h, m + 2, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        r.hash = h;
        r.status = (h == desc[i].expect_hash || (desc[i].flags & ZPK_DF_SKIP_HASH)) ? 0 : 15;
    }
    return ZPK_OK;
}

// One sub-batch of the host path: entries [0, n) of hd/desc, whose slots (hd[i].dst_offset, already laid out) total
// out_total bytes.  `image` is what gets staged: either the archive itself (span mode, src_offset = archive offsets,
// staged range [lo, hi)) or a packed copy of just these payloads (gather mode: hd[i].src_offset already rewritten).
#ifndef ZPK_PIN_CHUNK
#define ZPK_PIN_CHUNK (32ull << 20)
#endif
// The pinned staging buffers are touched by copy engines and host threads only, never by kernels: non-coherent host memory
// (measured, tools/micro/pinned_copy.hip: H2D 47.8 vs 40.5 GB/s, D2H 55.9 vs 47.4 GB/s against the default, coherent kind)
#ifndef ZPK_PIN_FLAGS
#define ZPK_PIN_FLAGS hipHostMallocNonCoherent
#endif
#ifndef ZPK_SCATTER_THREADS
#define ZPK_SCATTER_THREADS 4u
#endif
static int pin_ready(zpk_codec* c)
{
    if (!c->h_pj && hipHostMalloc((void**)&c->h_pj, 64, hipHostMallocDefault) != hipSuccess) { c->h_pj = nullptr; return ZPK_E_NOMEM; }
    if (!c->piece_counters && hipHostMalloc((void**)&c->piece_counters, 64 * N_COUNTERS * sizeof(u32), hipHostMallocDefault) != hipSuccess) { c->piece_counters = nullptr; return ZPK_E_NOMEM; }
    for (int k = 0; k < 2; k++) {
        if (!c->h_pin[k] && hipHostMalloc((void**)&c->h_pin[k], ZPK_PIN_CHUNK, ZPK_PIN_FLAGS) != hipSuccess) { c->h_pin[k] = nullptr; snprintf(c->err, sizeof(c->err), "pinned staging: out of memory"); return ZPK_E_NOMEM; }
        if (!c->pin_ev[k] && hipEventCreateWithFlags(&c->pin_ev[k], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) { c->pin_ev[k] = nullptr; return ZPK_E_LAUNCH; }
    }
    return ZPK_OK;
}

// Device range [d_base, d_base + total) back to the host in pieces of ZPK_PIN_CHUNK bytes through the two pinned buffers; piece j + 1 is
// on the bus while piece j is scattered: entry i owns bytes [off(i), off(i) + len(i)) of the range (ascending in i) and goes to dst_ptrs[i].
extern "C++" {
struct NoPre { hipError_t operator()(u64) const { return hipSuccess; } };
// pre(q1): called before bytes below q1 of the range are copied (the pipeline makes the download stream wait for their decode there)
template <class OffFn, class LenFn, class PreFn = NoPre>
static int d2h_scatter(zpk_codec* c, const u8* d_base, u64 total, u64 n, uint8_t* const* dst_ptrs, OffFn off, LenFn len, hipError_t& e,
                       hipStream_t st = nullptr, PreFn pre = PreFn())
{
    int rc = pin_ready(c);
    if (rc) return rc;
    if (total == 0) return ZPK_OK;
    if (!st) st = c->stream;
    const u64 npieces = (total + ZPK_PIN_CHUNK - 1) / ZPK_PIN_CHUNK;
    u64 ei = 0;                                                                       // first entry that may still reach into the current piece
    e = pre(total < ZPK_PIN_CHUNK ? total : ZPK_PIN_CHUNK);
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_pin[0], d_base, total < ZPK_PIN_CHUNK ? total : ZPK_PIN_CHUNK, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipEventRecord(c->pin_ev[0], st);
    for (u64 j = 0; j < npieces && e == hipSuccess; j++) {
        const int k = (int)(j & 1);
        const u64 p0 = j * ZPK_PIN_CHUNK, p1 = p0 + ZPK_PIN_CHUNK < total ? p0 + ZPK_PIN_CHUNK : total;
        if (j + 1 < npieces) {
            const u64 q0 = p1, q1 = q0 + ZPK_PIN_CHUNK < total ? q0 + ZPK_PIN_CHUNK : total;
            e = pre(q1);
            if (e == hipSuccess) e = hipMemcpyAsync(c->h_pin[k ^ 1], d_base + q0, q1 - q0, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipEventRecord(c->pin_ev[k ^ 1], st);
            if (e != hipSuccess) break;
        }
        e = hipEventSynchronize(c->pin_ev[k]);
        if (e != hipSuccess) break;
        while (ei < n && off(ei) + len(ei) <= p0) ei++;
        u64 ej = ei;
        while (ej < n && off(ej) < p1) ej++;
        // the scatter of one piece, split over a few host threads by entry count (one thread copies at ~22 GB/s, the bus brings ~50)
        auto part = [&](u64 lo_i, u64 hi_i) {
            for (u64 i = lo_i; i < hi_i; i++) {
                const u64 o = off(i), l = len(i);
                const u64 a = o > p0 ? o : p0, z = o + l < p1 ? o + l : p1;
                if (z > a) memcpy(dst_ptrs[i] + (a - o), c->h_pin[k] + (a - p0), z - a);
            }
        };
        const u64 cnt = ej - ei;
        const unsigned T = p1 - p0 >= (4u << 20) ? ZPK_SCATTER_THREADS : 1u;
        if (T <= 1) part(ei, ej);
        else if (cnt < 8) {
            // a few large entries: the piece's BYTES are split over the threads (one entry of 256 MiB came home at one thread's rate)
            auto bytes = [&](u64 lo_b, u64 hi_b) {
                for (u64 i = ei; i < ej; i++) {
                    const u64 o = off(i), l = len(i);
                    const u64 a = o > lo_b ? o : lo_b, z = o + l < hi_b ? o + l : hi_b;
                    if (z > a) memcpy(dst_ptrs[i] + (a - o), c->h_pin[k] + (a - p0), z - a);
                }
            };
            const u64 span = p1 - p0;
            auto cut = [&](unsigned t) { return t >= T ? p1 : p0 + ((span * t / T) & ~(u64)4095); };
            std::thread th[ZPK_SCATTER_THREADS - 1];
            bool started[ZPK_SCATTER_THREADS - 1] = {};
            for (unsigned t = 1; t < T; t++) {
                try { th[t - 1] = std::thread(bytes, cut(t), cut(t + 1)); started[t - 1] = true; }
                catch (...) { started[t - 1] = false; }
            }
            bytes(p0, cut(1));
            for (unsigned t = 1; t < T; t++) { if (started[t - 1]) th[t - 1].join(); else bytes(cut(t), cut(t + 1)); }
        } else {
            // (a thread that cannot be started must not unwind through the C ABI: its share is copied inline instead)
            std::thread th[ZPK_SCATTER_THREADS - 1];
            bool started[ZPK_SCATTER_THREADS - 1] = {};
            for (unsigned t = 1; t < T; t++) {
                try { th[t - 1] = std::thread(part, ei + cnt * t / T, ei + cnt * (t + 1) / T); started[t - 1] = true; }
                catch (...) { started[t - 1] = false; }
            }
            part(ei, ei + cnt / T);
            for (unsigned t = 1; t < T; t++) { if (started[t - 1]) th[t - 1].join(); else part(ei + cnt * t / T, ei + cnt * (t + 1) / T); }
        }
    }
    return ZPK_OK;
}

// The other direction (round 3): entries that lie in the caller's (pageable, separate) buffers go up to the device range [d_base,
// d_base + total) in pieces of ZPK_PIN_CHUNK bytes through the two pinned buffers — piece j + 1 is gathered by a few host threads
// while piece j is on the bus.  Entry i owns bytes [off(i), off(i) + len(i)) of the range (ascending in i); the gaps between entries
// carry whatever the staging buffer held.  (One hipMemcpyAsync per entry out of pageable memory: 40 000 x 64 KiB took 0.5 s.)
template <class OffFn, class LenFn>
static int h2d_gather(zpk_codec* c, u8* d_base, u64 total, u64 n, const uint8_t* const* src_ptrs, OffFn off, LenFn len, hipError_t& e, hipStream_t st,
                      u8* const* pins = nullptr, hipEvent_t* evs = nullptr)
{
    int rc = pins ? ZPK_OK : pin_ready(c);
    if (rc) return rc;
    if (!pins) { pins = c->h_pin; evs = c->pin_ev; }
    e = hipSuccess;
    if (total == 0) return ZPK_OK;
    const u64 npieces = (total + ZPK_PIN_CHUNK - 1) / ZPK_PIN_CHUNK;
    u64 ei = 0;
    bool used[2] = { false, false };
    for (u64 j = 0; j < npieces && e == hipSuccess; j++) {
        const int k = (int)(j & 1);
        const u64 p0 = j * ZPK_PIN_CHUNK, p1 = p0 + ZPK_PIN_CHUNK < total ? p0 + ZPK_PIN_CHUNK : total;
        if (used[k]) { e = hipEventSynchronize(evs[k]); if (e != hipSuccess) break; }      // the buffer's previous piece has left
        while (ei < n && off(ei) + len(ei) <= p0) ei++;
        u64 ej = ei;
        while (ej < n && off(ej) < p1) ej++;
        auto part = [&](u64 lo_i, u64 hi_i) {
            for (u64 i = lo_i; i < hi_i; i++) {
                const u64 o = off(i), l = len(i);
                const u64 a = o > p0 ? o : p0, z = o + l < p1 ? o + l : p1;
                if (z > a) memcpy(pins[k] + (a - p0), src_ptrs[i] + (a - o), z - a);
            }
        };
        const u64 cnt = ej - ei;
        const unsigned T = p1 - p0 >= (4u << 20) ? ZPK_SCATTER_THREADS : 1u;
        if (T <= 1) part(ei, ej);
        else if (cnt < 8) {
            // a few large entries: the piece's BYTES are split over the threads (as in d2h_scatter)
            auto bytes = [&](u64 lo_b, u64 hi_b) {
                for (u64 i = ei; i < ej; i++) {
                    const u64 o = off(i), l = len(i);
                    const u64 a = o > lo_b ? o : lo_b, z = o + l < hi_b ? o + l : hi_b;
                    if (z > a) memcpy(pins[k] + (a - p0), src_ptrs[i] + (a - o), z - a);
                }
            };
            const u64 span = p1 - p0;
            auto cut = [&](unsigned t) { return t >= T ? p1 : p0 + ((span * t / T) & ~(u64)4095); };
            std::thread th[ZPK_SCATTER_THREADS - 1];
            bool started[ZPK_SCATTER_THREADS - 1] = {};
            for (unsigned t = 1; t < T; t++) {
                try { th[t - 1] = std::thread(bytes, cut(t), cut(t + 1)); started[t - 1] = true; }
                catch (...) { started[t - 1] = false; }
            }
            bytes(p0, cut(1));
            for (unsigned t = 1; t < T; t++) { if (started[t - 1]) th[t - 1].join(); else bytes(cut(t), cut(t + 1)); }
        } else {
            std::thread th[ZPK_SCATTER_THREADS - 1];
            bool started[ZPK_SCATTER_THREADS - 1] = {};
            for (unsigned t = 1; t < T; t++) {
                try { th[t - 1] = std::thread(part, ei + cnt * t / T, ei + cnt * (t + 1) / T); started[t - 1] = true; }
                catch (...) { started[t - 1] = false; }
            }
            part(ei, ei + cnt / T);
            for (unsigned t = 1; t < T; t++) { if (started[t - 1]) th[t - 1].join(); else part(ei + cnt * t / T, ei + cnt * (t + 1) / T); }
        }
        e = hipMemcpyAsync(d_base + p0, pins[k], p1 - p0, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipEventRecord(evs[k], st);
        used[k] = true;
    }
    // the pinned buffers serve the download next: both uploads have to be off them
    for (int k = 0; k < 2 && e == hipSuccess; k++) if (used[k]) e = hipEventSynchronize(evs[k]);
    return ZPK_OK;
}
}  // extern "C++"

// the staged image is followed by ZPK_SRC_SLACK bytes of the codec's own: the kernels may READ ZPK_SRC_READ_SLACK bytes past its
// logical end (wide loads next to an entry's last byte; the two-stage LZ4 parser fetches whole 64-byte groups) — never interpret them
#define ZPK_SRC_SLACK 192u
#define ZPK_SRC_READ_SLACK 128u
static int decode_host_chunk(zpk_codec* c, const u8* image, u64 image_size, u64 lo, u64 hi, zpk_decode_desc* hd,
                             const zpk_decode_desc* desc, u64 n, u64 out_total, uint8_t* const* dst_ptrs, zpk_decode_result* results)
{
    int rc;
    if ((rc = grow(c, (void**)&c->d_src, &c->src_cap, hi - lo + ZPK_SRC_SLACK)) || (rc = grow(c, (void**)&c->d_dst, &c->dst_cap, out_total + 16)) ||
        (rc = grow(c, &c->d_desc, &c->desc_cap, n * sizeof(zpk_decode_desc))) ||
        (rc = grow(c, &c->d_res, &c->res_cap, n * sizeof(zpk_decode_result)))) return rc;
    hipError_t e = hipSuccess;
    if (hi > lo) e = hipMemcpyAsync(c->d_src, image + lo, hi - lo, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_desc, hd, n * sizeof(zpk_decode_desc), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "H2D: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    // base pointer such that base + src_offset lands in the staged range; reads are clamped to it
    const u8* base = c->d_src - lo;
    rc = decode_launch(c, base, image_size, c->d_src, c->d_src + (hi - lo) + ZPK_SRC_READ_SLACK, (const zpk_decode_desc*)c->d_desc, n,
                       c->d_dst, out_total, (zpk_decode_result*)c->d_res, c->stream);
    if (rc) return rc;
    e = hipMemcpyAsync(results, c->d_res, n * sizeof(zpk_decode_result), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "decode: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    // hand the bytes back: everything the codec produced (a hash mismatch leaves the data in the buffer, like the
    // reference), and only that — a generous max_size costs device address space, not PCIe time
    u64 produced_total = 0;
    for (u64 i = 0; i < n; i++) {
        if (results[i].produced > desc[i].dst_capacity) results[i].produced = desc[i].dst_capacity;     // (cannot happen)
        produced_total += results[i].produced;
    }
    if (n > 1 && produced_total && produced_total * 2 >= out_total) {
        // dense: the slots come back in pieces of ZPK_PIN_CHUNK bytes through two PINNED staging buffers — piece j + 1 is on the bus
        // while piece j is scattered into the caller's (pageable) buffers.  (One hipMemcpy of everything into a fresh malloc, then
        // the scatter, ran at 6.7 GB/s: page faults + the driver's own staging of pageable memory.)
        rc = d2h_scatter(c, c->d_dst, out_total, n, dst_ptrs, [&](u64 i) { return (u64)hd[i].dst_offset; }, [&](u64 i) { return (u64)results[i].produced; }, e);
        if (rc) return rc;
    } else {
        for (u64 i = 0; i < n && e == hipSuccess; i++)
            if (results[i].produced) e = hipMemcpy(dst_ptrs[i], c->d_dst + hd[i].dst_offset, results[i].produced, hipMemcpyDeviceToHost);
    }
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "D2H: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    return rehash_short_entries(c, hd, desc, n, dst_ptrs, results);
}

// ---- the same chunk as a three-stage pipeline (round 3) ------------------------------------------------------------------------
// decode_host_chunk runs upload -> decode -> download one after the other: 0.71 GB up, 2 ms of kernels and 1.31 GB down took 40 ms
// for 20 000 x 64 KiB LZ4 entries, i.e. 30 GiB/s of decoded bytes where the bus alone allows ~47 (PCIe is full duplex).  Here the
// chunk is cut into pieces (at least ZPK_PIPE_PIECE output bytes and enough entries to fill the device: one wave per LZ4 entry,
// 48 Zstandard streams per CU): an UPLOADER thread sends the compressed span of piece after piece (pageable memory, its own
// stream), a LAUNCHER thread starts the decode of piece k as soon as its upload has been enqueued (the codec's stream waits for
// the upload's event), and this thread brings the output range back as ONE continuous double-buffered stream through the pinned
// buffers (download stream; before a range is copied that stream is told to wait for the decode of the pieces it covers) and
// scatters it.  Entries must lie in the archive roughly in batch order (each piece uploads the span of its own entries);
// anything else, or a chunk too small to be worth it, takes decode_host_chunk.
#ifndef ZPK_PIPE_PIECE
#define ZPK_PIPE_PIECE (64ull << 20)
#endif
struct HostPiece { u64 e0, e1, clo, chi, olo, ohi; };

static int decode_host_pipelined(zpk_codec* c, const u8* image, u64 image_size, u64 lo, u64 hi, zpk_decode_desc* hd,
                                 const zpk_decode_desc* desc, u64 n, u64 out_total, uint8_t* const* dst_ptrs, zpk_decode_result* results,
                                 bool& taken)
{
    taken = false;
    const u64 min_entries = c->zstd_hint ? 12288 : 4096;
    if (out_total < 3 * ZPK_PIPE_PIECE || n < 3 * min_entries) return ZPK_OK;
    // ---- pieces ----
    HostPiece pc[64];
    int np = 0;
    u64 span_sum = 0;
    for (u64 i = 0; i < n; ) {
        if (np == 64) return ZPK_OK;
        HostPiece& P = pc[np];
        P.e0 = i; P.olo = hd[i].dst_offset; P.clo = ~0ull; P.chi = 0;
        u64 out = 0;
        while (i < n && (out < ZPK_PIPE_PIECE || i - P.e0 < min_entries || n - i < min_entries / 2)) {
            const zpk_decode_desc& d = hd[i];
            const bool ok = d.comp_size && d.src_offset <= image_size && d.comp_size <= image_size - d.src_offset;
            if (ok) { if (d.src_offset < P.clo) P.clo = d.src_offset; if (d.src_offset + d.comp_size > P.chi) P.chi = d.src_offset + d.comp_size; }
            const u64 next = i + 1 < n ? hd[i + 1].dst_offset : out_total;
            out += next - d.dst_offset;
            i++;
        }
        P.e1 = i; P.ohi = i < n ? hd[i].dst_offset : out_total;
        if (P.clo > P.chi) { P.clo = lo; P.chi = lo; }
        if (P.clo < lo || P.chi > hi) return ZPK_OK;
        span_sum += P.chi - P.clo;
        np++;
    }
    if (np < 3 || span_sum > (hi - lo) + (hi - lo) / 4 + (1u << 20)) return ZPK_OK;      // entries not in archive order: the spans would be uploaded many times over
    int rc;
    if ((rc = grow(c, (void**)&c->d_src, &c->src_cap, hi - lo + ZPK_SRC_SLACK)) || (rc = grow(c, (void**)&c->d_dst, &c->dst_cap, out_total + 16)) ||
        (rc = grow(c, &c->d_desc, &c->desc_cap, n * sizeof(zpk_decode_desc))) ||
        (rc = grow(c, &c->d_res, &c->res_cap, n * sizeof(zpk_decode_result))) || (rc = pin_ready(c))) return rc;
    if (!c->s_up && hipStreamCreateWithFlags(&c->s_up, hipStreamNonBlocking) != hipSuccess) { c->s_up = nullptr; return ZPK_OK; }
    if (!c->s_dn && hipStreamCreateWithFlags(&c->s_dn, hipStreamNonBlocking) != hipSuccess) { c->s_dn = nullptr; return ZPK_OK; }
    for (int k = 0; k < 2 * np; k++)
        if (!c->pipe_ev[k] && hipEventCreateWithFlags(&c->pipe_ev[k], hipEventDisableTiming | hipEventReleaseToSystem) != hipSuccess) { c->pipe_ev[k] = nullptr; return ZPK_OK; }
    taken = true;
    hipError_t e = hipMemcpyAsync(c->d_desc, hd, n * sizeof(zpk_decode_desc), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "H2D: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    std::atomic<int> uploaded(0), launched(0);   // pieces whose upload / decode has been ENQUEUED with its event recorded (-1: failed)
    std::atomic<int> launch_rc(ZPK_OK);
    const int device = c->device;
    const u8* base = c->d_src - lo;
    auto uploader = [&]() {
        if (hipSetDevice(device) != hipSuccess) { uploaded.store(-1); return; }
        for (int k = 0; k < np; k++) {
            hipError_t ue = hipSuccess;
            if (pc[k].chi > pc[k].clo) ue = hipMemcpyAsync(c->d_src + (pc[k].clo - lo), image + pc[k].clo, pc[k].chi - pc[k].clo, hipMemcpyHostToDevice, c->s_up);
            if (ue == hipSuccess) ue = hipEventRecord(c->pipe_ev[2 * k], c->s_up);
            if (ue != hipSuccess) { uploaded.store(-1); return; }
            uploaded.store(k + 1);
        }
    };
    auto launcher = [&]() {
        if (hipSetDevice(device) != hipSuccess) { launch_rc.store(ZPK_E_LAUNCH); launched.store(-1); return; }
        for (int k = 0; k < np; k++) {
            int u;
            while ((u = uploaded.load()) >= 0 && u <= k) std::this_thread::yield();
            int lrc = u < 0 ? (int)ZPK_E_LAUNCH : (int)ZPK_OK;
            if (lrc == ZPK_OK && hipStreamWaitEvent(c->stream, c->pipe_ev[2 * k], 0) != hipSuccess) lrc = ZPK_E_LAUNCH;
            const HostPiece& P = pc[k];
            if (lrc == ZPK_OK)
                lrc = decode_launch(c, base, image_size, c->d_src + (P.clo - lo), c->d_src + (P.chi - lo) + ZPK_SRC_READ_SLACK, (const zpk_decode_desc*)c->d_desc + P.e0,
                                    P.e1 - P.e0, c->d_dst, out_total, (zpk_decode_result*)c->d_res + P.e0, c->stream);
            if (lrc == ZPK_OK && hipMemcpyAsync(c->piece_counters[k], c->d_counters, N_COUNTERS * sizeof(u32), hipMemcpyDeviceToHost, c->stream) != hipSuccess) lrc = ZPK_E_LAUNCH;
            if (lrc == ZPK_OK && hipEventRecord(c->pipe_ev[2 * k + 1], c->stream) != hipSuccess) lrc = ZPK_E_LAUNCH;
            if (lrc != ZPK_OK) { launch_rc.store(lrc); launched.store(-1); return; }
            launched.store(k + 1);
        }
    };
    // (a thread that cannot be started must not unwind through the C ABI: its stage then runs here, in order — still correct, no overlap)
    std::thread t_up, t_launch;
    bool up_threaded = true, launch_threaded = true;
    try { t_up = std::thread(uploader); } catch (...) { up_threaded = false; }
    if (!up_threaded) uploader();
    try { t_launch = std::thread(launcher); } catch (...) { launch_threaded = false; }
    if (!launch_threaded) launcher();
    // ---- download: one continuous stream over the chunk's output range ----
    int next = 0;                                // first piece the download stream has not been told to wait for
    auto pre = [&](u64 q1) -> hipError_t {
        while (next < np && pc[next].olo < q1) {
            int l;
            while ((l = launched.load()) >= 0 && l <= next) std::this_thread::yield();
            if (l < 0) return hipErrorUnknown;
            hipError_t pe = hipStreamWaitEvent(c->s_dn, c->pipe_ev[2 * next + 1], 0);
            const HostPiece& P = pc[next];
            if (pe == hipSuccess) pe = hipMemcpyAsync(results + P.e0, (const zpk_decode_result*)c->d_res + P.e0, (P.e1 - P.e0) * sizeof(zpk_decode_result),
                                                      hipMemcpyDeviceToHost, c->s_dn);      // in stream order before the bytes: there when they are scattered
            if (pe != hipSuccess) return pe;
            next++;
        }
        return hipSuccess;
    };
    rc = d2h_scatter(c, c->d_dst, out_total, n, dst_ptrs, [&](u64 i) { return (u64)hd[i].dst_offset; },
                     [&](u64 i) { const u64 p = results[i].produced; return p > desc[i].dst_capacity ? (u64)desc[i].dst_capacity : p; }, e, c->s_dn, pre);
    if (up_threaded) t_up.join();
    if (launch_threaded) t_launch.join();
    if (rc == ZPK_OK && e == hipSuccess) e = pre(out_total + 1);       // (pieces with no output bytes: their results still come back)
    if (rc == ZPK_OK && e == hipSuccess) e = hipStreamSynchronize(c->s_dn);
    if (launch_rc.load() != ZPK_OK) rc = launch_rc.load();
    else if (rc == ZPK_OK && e != hipSuccess) { snprintf(c->err, sizeof(c->err), "host decode pipeline: %s", hipGetErrorString(e)); rc = ZPK_E_LAUNCH; }
    if (rc != ZPK_OK) { (void)hipStreamSynchronize(c->s_up); (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(c->s_dn); }
    for (u64 i = 0; i < n && rc == ZPK_OK; i++) if (results[i].produced > desc[i].dst_capacity) results[i].produced = desc[i].dst_capacity;     // (cannot happen)
    if (rc == ZPK_OK && hipStreamSynchronize(c->stream) == hipSuccess) {
        memset(c->host_totals, 0, sizeof(c->host_totals));
        for (int k = 0; k < np; k++) for (int w = 0; w < N_COUNTERS; w++) c->host_totals[w] += c->piece_counters[k][w];
        c->totals_valid = 1;
    }
    if (rc == ZPK_OK) rc = rehash_short_entries(c, hd, desc, n, dst_ptrs, results);
    return rc;
}

#ifndef ZPK_HOST_CHUNK_BYTES
#define ZPK_HOST_CHUNK_BYTES (4ull << 30)        // output slots of one device sub-batch of the host path (an entry larger than this goes alone)
#endif

// ---- entries that are SEQUENCES OF FRAMES, decoded frame-parallel (host path) -----------------------------------------------------
// One wave decodes one frame; an entry of hundreds of MiB in ONE frame is therefore one wave's work (~0.1 GB/s).  Entries written by
// this library's own writer above 2 MiB (zpk_encode.inc, zpk_stream.inc) are sequences of 512 KiB frames that each state their
// content size: the host walks the frames' block headers (4 / 3 bytes per block, in the caller's archive image), and when the frames
// tile the entry exactly — >= 2 of them, every one with its content size, the sizes summing to uncomp_size — they go to the device
// as a batch of their own, every frame a sub-entry with its own output range; the entry's XXH3 is computed over the assembled
// output by the whole chip (xxh3_span.h).  Large stored entries are cut into 512 KiB slices the same way.  Anything else — a single
// frame, a frame without content size, skippable frames, trailing bytes, a guard of lib/zpack_read.c:328-348 that would fire — stays
// with the one-wave decoders, and so does every entry one of whose frames fails here: verdicts come from one place only.
struct BigSub { u64 src_off, comp, out_off, size; };                  // a frame: byte ranges relative to its entry
struct BigEntry { u64 idx, first_sub, nsub; };
static inline u32 hrd32(const u8* p) { u32 v; memcpy(&v, p, 4); return v; }
static inline u64 hrd64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }

static bool walk_lz4_frames(const u8* p, u64 comp, u64 uncomp, std::vector<BigSub>& subs)
{
    const size_t start = subs.size();
    u64 ip = 0, out = 0;
    while (ip < comp) {
        if (comp - ip < 15 + 4 || hrd32(p + ip) != 0x184D2204u) goto other;
        {
            const u8 flg = p[ip + 4];
            if ((flg >> 6) != 1 || (flg & 0x03) || !(flg & 0x08)) goto other;        // version 01, no reserved bit, no dictionary, content size present
            const u64 csz = hrd64(p + ip + 6);
            u64 q = ip + 15;
            for (;;) {
                if (comp - q < 4) goto other;
                const u32 w = hrd32(p + q); q += 4;
                if (w == 0) break;
                const u64 nb = (u64)(w & 0x7FFFFFFFu) + ((flg & 0x10) ? 4 : 0);
                if (nb > comp - q) goto other;
                q += nb;
            }
            if (flg & 0x04) { if (comp - q < 4) goto other; q += 4; }
            if (csz == 0 || csz > uncomp - out || (out & 255)) goto other;           // (output ranges start on 256-byte boundaries, like the slots of any batch)
            subs.push_back(BigSub{ ip, q - ip, out, csz });
            out += csz; ip = q;
        }
    }
    if (out == uncomp && subs.size() - start >= 2) return true;
other:
    subs.resize(start);
    return false;
}

static bool walk_zstd_frames(const u8* p, u64 comp, u64 uncomp, std::vector<BigSub>& subs)
{
    const size_t start = subs.size();
    u64 ip = 0, out = 0;
    while (ip < comp) {
        if (comp - ip < 4 + 1 + 1 + 3 || hrd32(p + ip) != 0xFD2FB528u) goto other;
        {
            const u8 fhd = p[ip + 4];
            const u32 fcs_flag = fhd >> 6, ss = (fhd >> 5) & 1, did = fhd & 3;
            if (fhd & 0x08) goto other;                                               // reserved bit
            const u32 fcs_bytes = fcs_flag == 0 ? ss : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
            if (!fcs_bytes) goto other;                                               // no content size: its output cannot be placed
            u64 q = ip + 5 + (ss ? 0 : 1) + (did == 3 ? 4 : did);
            if (q > comp || comp - q < fcs_bytes) goto other;
            u64 fcs = 0;
            for (u32 i = 0; i < fcs_bytes; i++) fcs |= (u64)p[q + i] << (8 * i);
            if (fcs_bytes == 2) fcs += 256;
            q += fcs_bytes;
            for (;;) {
                if (comp - q < 3) goto other;
                const u32 w = (u32)p[q] | ((u32)p[q + 1] << 8) | ((u32)p[q + 2] << 16); q += 3;
                const u32 type = (w >> 1) & 3;
                if (type == 3) goto other;
                const u64 nb = type == 1 ? 1 : (w >> 3);
                if (nb > comp - q) goto other;
                q += nb;
                if (w & 1) break;
            }
            if (fhd & 0x04) { if (comp - q < 4) goto other; q += 4; }
            if (fcs == 0 || fcs > uncomp - out || (out & 255)) goto other;
            subs.push_back(BigSub{ ip, q - ip, out, fcs });
            out += fcs; ip = q;
        }
    }
    if (out == uncomp && subs.size() - start >= 2) return true;
other:
    subs.resize(start);
    return false;
}

// ---- ONE LARGE LZ4 FRAME (what the reference writer produces for any large entry: lib/zpack_write.c:204-210), block-parallel: lz4_pj.h ----
// XXH32 of a frame descriptor (2 .. 14 bytes; xxHash specification, inputs shorter than 16 bytes): the header checksum byte is (h >> 8) & 0xFF
static u32 host_xxh32_small(const u8* p, u32 len)
{
    const u32 P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    (void)P1; (void)P2;
    u32 h = P5 + len;
    u32 i = 0;
    for (; i + 4 <= len; i += 4) { h += hrd32(p + i) * P3; h = ((h << 17) | (h >> 15)) * P4; }
    for (; i < len; i++) { h += (u32)p[i] * P5; h = ((h << 11) | (h >> 21)) * P1; }
    h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3; h ^= h >> 16;
    return h;
}
#ifndef ZPK_PJ_MIN_BLOCKS
#define ZPK_PJ_MIN_BLOCKS 4u                          // LZ4: 256 KiB (fewer blocks: the fixed ~0.6 ms is not earned back)
#define ZPK_ZPJ_MIN_BLOCKS 2u                         // Zstandard: 2 blocks (the serial FSE chain of ONE block, ~3.5 ms, is the fixed cost either way)
#endif
// The entry is ONE frame of 64 KiB blocks, nothing optional but a content size that agrees with the entry, nothing behind its EndMark:
// its block table (offsets relative to the entry).  Anything else: false (the one-wave decoder's).
static bool walk_lz4_single(const u8* p, u64 comp, u64 uncomp, std::vector<PjBlock>& blocks, int& independent)
{
    blocks.clear();
    if (comp < 7 + 4 || comp >= 0x7FFF0000ull || uncomp >= 0x7FFF0000ull || hrd32(p) != 0x184D2204u) return false;
    const u8 flg = p[4], bd = p[5];
    if ((flg >> 6) != 1 || (flg & 0x03) || (flg & 0x10) || (flg & 0x04) || bd != 0x40) return false;      // version 01, no dictionary, no block / content checksums, 64 KiB blocks
    const u32 hdr = 7 + ((flg & 0x08) ? 8u : 0u);
    if (comp < hdr + 4) return false;
    if ((flg & 0x08) && hrd64(p + 6) != uncomp) return false;
    if (((host_xxh32_small(p + 4, hdr - 5) >> 8) & 0xFF) != p[hdr - 1]) return false;
    independent = (flg >> 5) & 1;
    u64 q = hdr, recs = 0;
    for (;;) {
        if (comp - q < 4) return false;
        const u32 w = hrd32(p + q); q += 4;
        if (w == 0) break;
        const u32 n = w & 0x7FFFFFFFu;
        if (n == 0 || n > PJ_BLOCK || n > comp - q) return false;
        PjBlock B; B.comp_off = (u32)q; B.comp_size = w; B.rec_base = (u32)recs; B.out_size = 0; B.out_off = 0; B.nrec = 0;
        if (!(w >> 31)) recs += n / 3 + 2;
        if (recs > 0xFFFFFF00ull) return false;
        blocks.push_back(B);
        q += n;
    }
    return q == comp && blocks.size() >= ZPK_PJ_MIN_BLOCKS;
}

// The common second half of the block-parallel readers (lz4_pj.h, zstd_pj.h): the blocks' references exist per chunk through `init`, the
// block table `hb` (output offsets) is on the host.  Chunks of ZPK_PJ_CHUNK_BLOCKS blocks are resolved one after the other, hashed and
// downloaded beside that.  accept_mismatch: a wrong XXH3 is this path's verdict (LZ4: everything about the frame was checked); otherwise
// the one-wave decoder decides.
extern "C++" {
template <class InitFn>
static int pj_finish(zpk_codec* c, const zpk_decode_desc& d, const std::vector<PjBlock>& hb, u64 chunk_blocks, u64 gather_src_size, InitFn init, bool accept_mismatch,
                     u8* d_out, uint8_t* dst_ptr, zpk_decode_result& result, u8& redo)
{
    const u64 nb = hb.size(), n = d.uncomp_size;
    hipStream_t st = c->stream;
    PjBlock* const B = (PjBlock*)c->d_pj_blocks;
    u32* const S = (u32*)c->d_pj_S;
    hipError_t e;
    int rc;
    if (chunk_blocks * ZPK_PJ_MAX_CHUNKS < nb) chunk_blocks = (nb + ZPK_PJ_MAX_CHUNKS - 1) / ZPK_PJ_MAX_CHUNKS;      // (small blocks, a very large entry: larger chunks)
    const u64 nchunks = (nb + chunk_blocks - 1) / chunk_blocks;
    if (nchunks > ZPK_PJ_MAX_CHUNKS) return ZPK_OK;
    for (u64 k = 0; k < nchunks; k++) if (!c->pj_ev[k] && hipEventCreateWithFlags(&c->pj_ev[k], hipEventDisableTiming) != hipSuccess) { c->pj_ev[k] = nullptr; return ZPK_OK; }
    if (!c->s_dn && hipStreamCreateWithFlags(&c->s_dn, hipStreamNonBlocking) != hipSuccess) { c->s_dn = nullptr; return ZPK_OK; }
    if ((rc = pin_ready(c))) return rc;
    // the XXH3 of the output runs BESIDE all this on its own stream, section by section as the chunks become final (xxh3_span.h: the
    // partial sums of a section's blocks side by side, then the one-wave chain over them — 14 ms for 256 MiB, as long as everything else
    // together, which is why it must not come behind)
    if (!c->s_left && hipStreamCreateWithFlags(&c->s_left, hipStreamNonBlocking) != hipSuccess) { c->s_left = nullptr; return ZPK_OK; }
    hipStream_t sh = c->s_left;
    const u64 part_blocks = xxh3_span_blocks(n), ngroups = part_blocks / XS_GROUP;
    if ((rc = grow(c, (void**)&c->d_xpart, &c->xpart_cap, 256 + part_blocks * 64 + 256))) return rc;
    zpk_span* const d_span = (zpk_span*)c->d_xpart;
    u64* const d_part = (u64*)(c->d_xpart + 256);
    u64* const d_hash = (u64*)(c->d_xpart + 256 + part_blocks * 64);
    u64* const d_state = d_hash + 8;
    zpk_span span; span.off = 0; span.len = n; span.part_base = 0;
    e = hipMemcpyAsync(d_span, &span, sizeof(span), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "large frame: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    // ---- the output to the caller's buffer starts while the chunks are still being resolved: d2h_scatter (pinned staging, the copy of piece
    // j + 1 on the bus while piece j is copied out by a few threads) runs on a helper thread and takes a piece as soon as the chunks under it
    // have been ENQUEUED (their events recorded).  The bytes are the entry's whatever the verdict (lib/zpack_read.c:466-468 leaves
    // them); if the path turns out not to have been regular the one-wave decoder overwrites them. ----
    std::vector<u64> chunk_hi(nchunks);
    for (u64 k = 0; k < nchunks; k++) { const u64 b1 = (k + 1) * chunk_blocks; chunk_hi[k] = b1 < nb ? hb[b1].out_off : n; }
    std::atomic<u64> enqueued{0};
    std::atomic<int> give_up{0};
    int dn_rc = ZPK_OK; hipError_t dn_e = hipSuccess;
    auto download = [&]() {
        (void)hipSetDevice(c->device);
        u64 waited = 0;
        auto pre = [&](u64 q1) -> hipError_t {
            while (waited < nchunks && (waited == 0 || chunk_hi[waited - 1] < q1)) {
                while (enqueued.load(std::memory_order_acquire) <= waited) { if (give_up.load(std::memory_order_acquire)) return hipErrorUnknown; std::this_thread::yield(); }
                const hipError_t we = hipStreamWaitEvent(c->s_dn, c->pj_ev[waited], 0);
                if (we != hipSuccess) return we;
                waited++;
            }
            return hipSuccess;
        };
        uint8_t* optr[1] = { dst_ptr };
        dn_rc = d2h_scatter(c, d_out, n, 1, optr, [&](u64) { return (u64)0; }, [&](u64) { return n; }, dn_e, c->s_dn, pre);
    };
    std::thread dn_thread;
    bool dn_started = false;
    if (dst_ptr) { try { dn_thread = std::thread(download); dn_started = true; } catch (...) { dn_started = false; } }      // (no thread: the download follows the loop; no host destination: the output stays where it is)
    // ---- every chunk: references, PJ_MAX_ROUNDS rounds of pointer doubling (a round behind the last one that changed anything returns at
    // once: no host round trip), the gather; an event behind each chunk lets its bytes be hashed and go home while the next is resolved ----
    bool launch_failed = false;
    u64 g_lo = 0;
    for (u64 k = 0; k < nchunks; k++) {
        const u32 b0 = (u32)(k * chunk_blocks), b1 = (u32)(b0 + chunk_blocks < nb ? b0 + chunk_blocks : nb);
        const u64 lo = hb[b0].out_off, hi = chunk_hi[k];
        const u32 grid = (u32)((hi - (lo & ~3ull) + 1023) / 1024), jgrid = (u32)((hi - lo + 1023) / 1024);
        (void)hipMemsetAsync(c->d_pj_flags + PJ_ROUND0, 0, PJ_MAX_ROUNDS * 4, st);
        init(b0, b1, st);
        if (grid) {
            for (u32 r = 0; r < PJ_MAX_ROUNDS; r++)
                hipLaunchKernelGGL(k_pj_jump, dim3(jgrid), dim3(256), 0, st, S, (const PjBlock*)B, b0, b1, (u32)nb, c->d_pj_flags, r);
            hipLaunchKernelGGL(k_pj_gather, dim3(grid), dim3(256), 0, st, (const u32*)S, (const PjBlock*)B, b0, b1, (u32)nb, (const u8*)c->d_src, gather_src_size, d_out, c->d_pj_flags);
        }
        if (hipEventRecord(c->pj_ev[k], st) != hipSuccess || hipStreamWaitEvent(sh, c->pj_ev[k], 0) != hipSuccess) { launch_failed = true; break; }
        enqueued.store(k + 1, std::memory_order_release);
        const bool last = k + 1 == nchunks;
        const u64 g_hi = last ? ngroups : (hi >> 10) / XS_GROUP;                       // groups of 64 blocks that are final now
        if (g_hi > g_lo) hipLaunchKernelGGL(k_xxh3_partials, dim3((u32)((g_hi - g_lo + 3) / 4)), dim3(256), 0, sh, (const u8*)d_out, (const zpk_span*)d_span, 1u, g_lo, g_hi, d_part);
        if (g_hi > g_lo || last)
            hipLaunchKernelGGL(k_xxh3_chain, dim3(1), dim3(64), 0, sh, (const u8*)d_out, (const zpk_span*)d_span, (const u64*)d_part, d_hash, d_state, g_lo * XS_GROUP, g_hi * XS_GROUP, last ? 1 : 0);
        if (g_hi > g_lo) g_lo = g_hi;
    }
    if (launch_failed) give_up.store(1, std::memory_order_release);
    e = launch_failed ? hipErrorUnknown : hipMemcpyAsync(&c->h_pj[0], d_hash, 8, hipMemcpyDeviceToHost, sh);
    if (e == hipSuccess) e = hipMemcpyAsync(&c->h_pj[1], c->d_pj_flags, 4, hipMemcpyDeviceToHost, sh);
    if (dn_started) dn_thread.join(); else if (!launch_failed && dst_ptr) download();
    if (e != hipSuccess || dn_rc != ZPK_OK || dn_e != hipSuccess) {
        (void)hipDeviceSynchronize();
        if (dn_rc != ZPK_OK && !launch_failed) return dn_rc;
        snprintf(c->err, sizeof(c->err), "large frame: %s", hipGetErrorString(e != hipSuccess ? e : dn_e));
        return ZPK_E_LAUNCH;
    }
    e = dst_ptr ? hipStreamSynchronize(c->s_dn) : hipSuccess;
    const hipError_t e2 = hipStreamSynchronize(sh), e3 = hipStreamSynchronize(st);
    if (e == hipSuccess) e = e2 != hipSuccess ? e2 : e3;
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "large frame: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    if ((u32)c->h_pj[1] != 0) return ZPK_OK;                                                       // PJ_ERR: something was irregular after all
    const u64 h = c->h_pj[0];
    result.hash = h; result.produced = n; result.detail = 0;
    const bool hash_ok = (d.flags & ZPK_DF_SKIP_HASH) || h == d.expect_hash;
    if (!hash_ok && !accept_mismatch) return ZPK_OK;                                               // (the one-wave decoder gives this entry's verdict)
    result.status = hash_ok ? 0 : 15;                                                              // ZPACK_ERROR_FILE_HASH_MISMATCH, lib/zpack_read.c:467
    c->big_last[0]++; c->big_last[1] += (u32)nb;
    redo = 0;
    return ZPK_OK;
}

}   // extern "C++"

// -> ZPK_OK with redo = 0: the entry is decoded, hashed and delivered; redo = 1: not this path's (the one-wave decoder decides)
// (d_archive != nullptr: the entry's bytes are on the device already; d_out != nullptr: that is where the output goes, on the device)
static int decode_big_lz4_single(zpk_codec* c, const u8* archive, const zpk_decode_desc& d, const std::vector<PjBlock>& blocks, int independent,
                                 uint8_t* dst_ptr, zpk_decode_result& result, u8& redo, const u8* d_archive = nullptr, u8* d_out = nullptr)
{
    redo = 1;
    const u64 nb = blocks.size(), n = d.uncomp_size;
    const u64 total_recs = (u64)blocks.back().rec_base + ((blocks.back().comp_size >> 31) ? 0 : (blocks.back().comp_size / 3 + 2));
    int rc;
    if ((rc = grow(c, (void**)&c->d_src, &c->src_cap, d.comp_size + ZPK_SRC_SLACK)) || (!d_out && (rc = grow(c, (void**)&c->d_dst, &c->dst_cap, n + 16))) ||
        (rc = grow(c, &c->d_pj_blocks, &c->pj_blocks_cap, nb * sizeof(PjBlock))) || (rc = grow(c, &c->d_pj_recs, &c->pj_recs_cap, (total_recs + 64) * 8)) ||
        (rc = grow(c, &c->d_pj_masks, &c->pj_masks_cap, nb * (PJ_BLOCK / 8))) || (rc = grow(c, &c->d_pj_S, &c->pj_S_cap, n * 4 + 64))) { c->err[0] = 0; return ZPK_OK; }     // no memory for the scratch: the one-wave decoder
    if (!c->d_pj_flags && hipMalloc((void**)&c->d_pj_flags, 256) != hipSuccess) { c->d_pj_flags = nullptr; (void)hipGetLastError(); return ZPK_OK; }
    hipStream_t st = c->stream;
    hipError_t e = d_archive ? hipMemcpyAsync(c->d_src, d_archive + d.src_offset, d.comp_size, hipMemcpyDeviceToDevice, st)
                             : hipMemcpyAsync(c->d_src, archive + d.src_offset, d.comp_size, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_pj_blocks, blocks.data(), nb * sizeof(PjBlock), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_pj_flags, 0, 256, st);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "H2D: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    PjBlock* const B = (PjBlock*)c->d_pj_blocks;
    u32* const S = (u32*)c->d_pj_S;
    hipLaunchKernelGGL(k_pj_parse, dim3((u32)nb), dim3(64), 0, st, (const u8*)c->d_src, d.comp_size, B, (u32)nb, (u64*)c->d_pj_recs, (u32*)c->d_pj_masks, c->d_pj_flags);
    hipLaunchKernelGGL(k_pj_scan, dim3(1), dim3(64), 0, st, B, (u32)nb, c->d_pj_flags);
    u32 hf[4] = {0, 0, 0, 0};
    // the verdict of the parse and the block table with its output offsets, back on the host (the one round trip of this path):
    // chunk k = blocks [k * ZPK_PJ_CHUNK_BLOCKS, ...) = output bytes [lo_k, hi_k)
    std::vector<PjBlock> hb(nb);
    e = hipMemcpyAsync(hf, c->d_pj_flags, sizeof(hf), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(hb.data(), B, nb * sizeof(PjBlock), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "large LZ4 frame: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    if (hf[PJ_ERR] || (((u64)hf[PJ_TOTAL + 1] << 32) | hf[PJ_TOTAL]) != n) return ZPK_OK;          // irregular, or the sizes do not add up
    auto init = [&](u32 b0, u32 b1, hipStream_t s2) {
        hipLaunchKernelGGL(k_pj_init, dim3(b1 - b0), dim3(256), 0, s2, (const PjBlock*)B, b0, (u32)nb, (const u64*)c->d_pj_recs, (const u32*)c->d_pj_masks, S, n, c->d_pj_flags, independent);
    };
    return pj_finish(c, d, hb, ZPK_PJ_CHUNK_BLOCKS, d.comp_size, init, true, d_out ? d_out : c->d_dst, dst_ptr, result, redo);
}

// Bytes an FSE table description (RFC 8878 4.1.1) takes, or -1 (malformed / beyond `avail` / more symbols or accuracy than its kind allows)
static int zpj_ncount_len(const u8* p, u64 avail, int max_sym, int max_al)
{
    u64 bit = 0;
    auto rd = [&](u32 n) -> i64 {                                   // n <= 16 bits from the LSB-first stream; -1 beyond the bytes
        if (((bit + n + 7) >> 3) > avail) return -1;
        u32 v = 0;
        for (u32 i = 0; i < 4 && (bit >> 3) + i < avail; i++) v |= (u32)p[(bit >> 3) + i] << (8 * i);
        v = (v >> (bit & 7)) & ((1u << n) - 1u);
        bit += n;
        return (i64)v;
    };
    i64 x = rd(4);
    if (x < 0) return -1;
    const int al = 5 + (int)x;
    if (al > max_al) return -1;
    int remaining = 1 << al, s = 0;
    while (remaining > 0 && s <= max_sym) {
        int nb = 0; for (u32 t = (u32)remaining + 1; t; t >>= 1) nb++;      // highbit(remaining + 1) + 1
        x = rd((u32)nb);
        if (x < 0) return -1;
        u32 val = (u32)x;
        const u32 lower_mask = (1u << (nb - 1)) - 1, threshold = (1u << nb) - 1 - ((u32)remaining + 1);
        if ((val & lower_mask) < threshold) { bit -= 1; val &= lower_mask; }
        else if (val > lower_mask) val -= threshold;
        const int proba = (int)val - 1;
        remaining -= proba < 0 ? 1 : proba;
        s++;
        if (proba == 0) {
            for (;;) {
                x = rd(2);
                if (x < 0) return -1;
                s += (int)x;
                if (s > max_sym + 1) return -1;
                if (x != 3) break;
            }
        }
    }
    if (remaining != 0) return -1;
    return (int)((bit + 7) >> 3);
}
// what governs the three sequence tables at some point of a frame: mode (0 predefined, 1 RLE, 2 FSE description, 3 nothing yet) and
// where the description starts (offset in the compressed entry)
struct ZpjTabs { u32 mode[3], off[3]; };

// One block of a Zstandard frame at p (avail bytes follow): 1 = parsed into B (hdr_off = at; sizes, literals and sequence headers; nothing
// about trees or slots), 0 = the bytes end inside it, -1 = not a block this path takes (reserved type, Repeat_Mode table, sizes that
// disagree).  *last = its Last_Block bit, *total = 3 + the bytes of its body.
// tabs != nullptr: Repeat_Mode tables are taken — resolved against *tabs, which is updated with what this block defines.
static int zpj_parse_block(const u8* p, u64 avail, u64 at, ZpjBlock& B, u32* last, u64* total, ZpjTabs* tabs = nullptr)
{
    if (avail < 3) return 0;
    const u32 bh = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16);
    const u32 bt = (bh >> 1) & 3, bs = bh >> 3;
    *last = bh & 1;
    memset(&B, 0, sizeof(B));
    B.hdr_off = (u32)at; B.type = bt; B.size = bs; B.tree_src = ZPJ_NONE;
    if (bt == 3 || bs > ZPJ_BLOCK || at > 0x7FFFFF00ull) return -1;
    const u64 body = bt == 1 ? 1 : bs;
    *total = 3 + body;
    if (avail - 3 < body) return 0;
    if (bt != 2) return 1;
    const u8* const b = p + 3;
    if (bs < 3) return -1;
    const u32 b0 = b[0], lt = b0 & 3, fmt = (b0 >> 2) & 3;
    u32 hl, regen, csize;
    if (lt < 2) {
        if ((fmt & 1) == 0) { hl = 1; regen = b0 >> 3; }
        else if (fmt == 1) { hl = 2; regen = (b0 >> 4) | ((u32)b[1] << 4); }
        else { hl = 3; regen = (b0 >> 4) | ((u32)b[1] << 4) | ((u32)b[2] << 12); }
        csize = lt == 0 ? regen : 1u;
    } else {
        if (bs < 5) return -1;
        const u64 v = hrd32(b);
        if (fmt < 2) { hl = 3; regen = (u32)(v >> 4) & 0x3FF; csize = (u32)(v >> 14) & 0x3FF; }
        else if (fmt == 2) { hl = 4; regen = (u32)(v >> 4) & 0x3FFF; csize = (u32)(v >> 18); }
        else { hl = 5; regen = (u32)(v >> 4) & 0x3FFFF; csize = (u32)(v >> 22) | ((u32)b[4] << 10); }
    }
    if (regen > ZPJ_BLOCK || (u64)hl + csize > bs) return -1;
    B.lit_type = lt; B.lit_size = regen; B.lit_used = hl + csize;
    if (lt < 2) B.lit_ref = (u32)(at + 3 + hl);
    u64 o = B.lit_used;
    if (bs - o < 1) return -1;
    u64 nseq = b[o];
    if (nseq == 0) { if (bs - o != 1) return -1; }
    else {
        if (nseq < 128) o += 1;
        else if (nseq < 255) { if (bs - o < 2) return -1; nseq = ((nseq - 128) << 8) + b[o + 1]; o += 2; }
        else { if (bs - o < 3) return -1; nseq = (u64)b[o + 1] + ((u64)b[o + 2] << 8) + 0x7F00; o += 3; }
        if (bs - o < 1) return -1;
        const u32 modes = b[o];
        const bool any_repeat = ((modes >> 6) & 3) == 3 || ((modes >> 4) & 3) == 3 || ((modes >> 2) & 3) == 3;
        if ((modes & 3) || (any_repeat && !tabs)) return -1;
        B.tab_modes = 0x3F;
        if (tabs) {
            // the three descriptions follow the modes byte in the order LL, OF, ML; each is measured so that the next one's start — and
            // what a later Repeat_Mode block inherits — is known
            u64 q = o + 1;
            for (int kind = 0; kind < 3; kind++) {                     // (T_LL, T_OF, T_ML of zstd_wg.h)
                const u32 mode = (modes >> (6 - 2 * kind)) & 3;
                if (mode == 3) {
                    if (tabs->mode[kind] == 3) return -1;              // nothing to repeat
                } else {
                    tabs->mode[kind] = mode; tabs->off[kind] = (u32)(at + 3 + q);
                    if (mode == 1) { if (bs - q < 1) return -1; q += 1; }
                    else if (mode == 2) { const int n = zpj_ncount_len(b + q, bs - q, kind == 0 ? 35 : (kind == 1 ? 31 : 52), kind == 1 ? 8 : 9); if (n < 0) return -1; q += (u64)n; }
                }
                B.tab_off[kind] = tabs->off[kind];
                B.tab_modes = (B.tab_modes & ~(3u << (2 * kind))) | (tabs->mode[kind] << (2 * kind));
            }
        }
    }
    B.nseq = (u32)nseq;
    return 1;
}

// The header of a Zstandard frame as ZSTD_compressCCtx writes it (lib/zpack_write.c:179): no dictionary, no checksum, a window of at
// most 2^max_wlog bytes.  -> its size (0: the bytes end inside it, -1: not this path's); *window = Window_Size, *fcs = content size or ~0.
static int zpj_parse_frame_header(const u8* p, u64 avail, u32 max_wlog, u64* window, u64* fcs)
{
    if (avail < 6) return 0;
    if (hrd32(p) != 0xFD2FB528u) return -1;
    u64 q = 4;
    const u32 fhd = p[q++];
    const u32 fcs_flag = fhd >> 6, single = (fhd >> 5) & 1;
    if (fhd & 0x0F) return -1;                                      // reserved bit, content checksum, dictionary: the one-wave decoder's
    *window = 0;
    if (!single) {
        const u32 wdesc = p[q++], wlog = 10 + (wdesc >> 3);
        if (wlog > max_wlog) return -1;
        *window = (1ull << wlog) + ((1ull << wlog) >> 3) * (wdesc & 7);
    }
    const u32 fn = fcs_flag == 0 ? (single ? 1u : 0u) : (fcs_flag == 1 ? 2u : (fcs_flag == 2 ? 4u : 8u));
    if (avail - q < fn) return 0;
    *fcs = ~0ull;
    if (fn) { u64 v = 0; for (u32 i = 0; i < fn; i++) v |= (u64)p[q + i] << (8 * i); if (fn == 2) v += 256; *fcs = v; q += fn; }
    if (single) *window = *fcs;
    return (int)q;
}

// ONE Zstandard frame as ZSTD_compressCCtx writes it (lib/zpack_write.c:179): no dictionary, no checksum, its content size (if stated) the
// entry's, a window of at most 128 MiB, no Repeat_Mode table, every Treeless block behind a block with a tree, >= ZPK_PJ_MIN_BLOCKS
// blocks, nothing behind the last block -> the block table of zstd_pj.h.  slots = sequence slots (a block owns nseq + 1), lit_total =
// bytes of the literal arena.
static bool walk_zstd_single(const u8* p, u64 comp, u64 uncomp, std::vector<ZpjBlock>& blocks, u64& slots, u64& lit_total)
{
    blocks.clear(); slots = 0; lit_total = 0;
    u64 window = 0, fcs = ~0ull;
    const int hdr = zpj_parse_frame_header(p, comp, 27, &window, &fcs);
    if (hdr <= 0 || (fcs != ~0ull && fcs != uncomp)) return false;
    u64 q = (u64)hdr;
    u32 tree = ZPJ_NONE;
    ZpjTabs tabs; for (int k = 0; k < 3; k++) { tabs.mode[k] = 3; tabs.off[k] = 0; }
    for (;;) {
        ZpjBlock B; u32 last = 0; u64 total = 0;
        if (zpj_parse_block(p + q, comp - q, q, B, &last, &total, &tabs) != 1) return false;
        if (B.type == 2) {
            if (B.lit_type == 2) tree = (u32)blocks.size();
            if (B.lit_type == 3) { if (tree == ZPJ_NONE) return false; B.tree_src = tree; }
            if (B.lit_type >= 2) { B.lit_base = (u32)lit_total; lit_total += ((u64)B.lit_size + 15) / 16 * 16 + 64; }
            B.seq_base = (u32)slots;
            slots += (u64)B.nseq + 1;
        }
        blocks.push_back(B);
        q += total;
        if (last) break;
        if (slots > 0x7FFFFF00ull || lit_total > 0x70000000ull) return false;
    }
    if (q != comp || blocks.size() < ZPK_ZPJ_MIN_BLOCKS) return false;
    return comp + lit_total + 1024 < 0x7FFFFF00ull;
}

// -> ZPK_OK with redo = 0: the entry is decoded, its XXH3 is the expected one, the bytes are delivered; redo = 1: not this path's
static int decode_big_zstd_single(zpk_codec* c, const u8* archive, const zpk_decode_desc& d, std::vector<ZpjBlock>& blocks, u64 slots, u64 lit_total,
                                  uint8_t* dst_ptr, zpk_decode_result& result, u8& redo, const u8* d_archive = nullptr, u8* d_out = nullptr)
{
    redo = 1;
    const u64 nb = blocks.size(), n = d.uncomp_size;
    const u64 arena_off = (d.comp_size + 64 + 255) & ~255ull;
    for (u64 b = 0; b < nb; b++) if (blocks[b].type == 2 && blocks[b].lit_type >= 2) blocks[b].lit_ref = (u32)(arena_off + blocks[b].lit_base);
    // work items of the sequence stage: the compressed blocks that have sequences
    std::vector<zpk_decode_desc> items(nb);
    std::vector<u32> list;
    memset(items.data(), 0, nb * sizeof(zpk_decode_desc));
    for (u64 b = 0; b < nb; b++) {
        if (blocks[b].type != 2 || blocks[b].nseq == 0) continue;
        items[b].src_offset = blocks[b].hdr_off; items[b].comp_size = 3ull + blocks[b].size;
        items[b].dst_offset = 8ull * blocks[b].seq_base; items[b].dst_capacity = 8ull * blocks[b].nseq; items[b].method = ZPK_METHOD_ZSTD;
        items[b].uncomp_size = (u64)blocks[b].tab_off[0] | ((u64)blocks[b].tab_off[1] << 32);         // (k_zstd_fse_blocks: inherited table descriptions)
        items[b].expect_hash = (u64)blocks[b].tab_off[2] | ((u64)blocks[b].tab_modes << 32);
        list.push_back((u32)b);
    }
    const u64 aux_desc = 0, aux_list = (nb * sizeof(zpk_decode_desc) + 255) & ~255ull, aux_state = aux_list + ((nb * 4 + 255) & ~255ull),
              aux_rep = aux_state + ((nb * 4 + 255) & ~255ull), aux_size = aux_rep + nb * 12 + 256;
    int rc;
    if ((rc = grow(c, (void**)&c->d_src, &c->src_cap, arena_off + lit_total + ZPK_SRC_SLACK)) || (!d_out && (rc = grow(c, (void**)&c->d_dst, &c->dst_cap, n + 16))) ||
        (rc = grow(c, &c->d_pj_blocks, &c->pj_blocks_cap, nb * sizeof(PjBlock))) || (rc = grow(c, &c->d_zpj_blocks, &c->zpj_blocks_cap, nb * sizeof(ZpjBlock))) ||
        (rc = grow(c, &c->d_zpj_aux, &c->zpj_aux_cap, aux_size)) || (rc = grow(c, &c->d_pj_recs, &c->pj_recs_cap, (slots + 64) * 8)) ||
        (rc = grow(c, &c->d_zpj_pos, &c->zpj_pos_cap, (slots + 64) * 8)) || (rc = grow(c, &c->d_pj_masks, &c->pj_masks_cap, nb * (ZPJ_BLOCK / 8))) ||
        (rc = grow(c, &c->d_pj_S, &c->pj_S_cap, n * 4 + 64))) { c->err[0] = 0; return ZPK_OK; }        // no memory for the scratch: the one-wave decoder
    if (!c->d_pj_flags && hipMalloc((void**)&c->d_pj_flags, 256) != hipSuccess) { c->d_pj_flags = nullptr; (void)hipGetLastError(); return ZPK_OK; }
    hipStream_t st = c->stream;
    u8* const aux = (u8*)c->d_zpj_aux;
    u32 hflags[64]; memset(hflags, 0, sizeof(hflags));
    hflags[ZPJ_CNT + ZF_COUNT_WORD] = (u32)list.size();
    std::vector<PjBlock> hb(nb);
    memset(hb.data(), 0, nb * sizeof(PjBlock));
    hipError_t e = d_archive ? hipMemcpyAsync(c->d_src, d_archive + d.src_offset, d.comp_size, hipMemcpyDeviceToDevice, st)
                             : hipMemcpyAsync(c->d_src, archive + d.src_offset, d.comp_size, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_zpj_blocks, blocks.data(), nb * sizeof(ZpjBlock), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_pj_blocks, hb.data(), nb * sizeof(PjBlock), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(aux + aux_desc, items.data(), nb * sizeof(zpk_decode_desc), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && !list.empty()) e = hipMemcpyAsync(aux + aux_list, list.data(), list.size() * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(aux + aux_state, 0, aux_size - aux_state, st);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_pj_flags, hflags, sizeof(hflags), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "H2D: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    PjBlock* const B = (PjBlock*)c->d_pj_blocks;
    ZpjBlock* const ZB = (ZpjBlock*)c->d_zpj_blocks;
    u32* const S = (u32*)c->d_pj_S;
    u32* const state = (u32*)(aux + aux_state);
    u32* const rep_out = (u32*)(aux + aux_rep);
    if (!list.empty()) {
        const u32 rows = (u32)list.size();
        const u32 grid = (rows + ZF_ROWS - 1) / ZF_ROWS < ZF_GRID_MAX ? (rows + ZF_ROWS - 1) / ZF_ROWS : ZF_GRID_MAX;
        hipLaunchKernelGGL(k_zstd_fse_blocks, dim3(grid), dim3(64), 0, st, (const u8*)c->d_src, (const zpk_decode_desc*)(aux + aux_desc), (const u32*)(aux + aux_list),
                           c->d_pj_flags + ZPJ_CNT, (u64*)c->d_pj_recs, state, rep_out, d.comp_size);
    }
    hipLaunchKernelGGL(k_zpj_lit, dim3((u32)nb), dim3(64), 0, st, c->d_src, arena_off + lit_total, arena_off, (const ZpjBlock*)ZB, (u32)nb, c->d_pj_flags);
    hipLaunchKernelGGL(k_zpj_reps, dim3(1), dim3(64), 0, st, ZB, (u32)nb, (const u32*)state, (const u32*)rep_out, c->d_pj_flags);
    hipLaunchKernelGGL(k_zpj_pos, dim3((u32)nb), dim3(256), 0, st, (const ZpjBlock*)ZB, B, (u32)nb, (const u64*)c->d_pj_recs, (u64*)c->d_zpj_pos, (u32*)c->d_pj_masks, c->d_pj_flags);
    hipLaunchKernelGGL(k_pj_scan, dim3(1), dim3(64), 0, st, B, (u32)nb, c->d_pj_flags);
    u32 hf[4] = {0, 0, 0, 0};
    e = hipMemcpyAsync(hf, c->d_pj_flags, sizeof(hf), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(hb.data(), B, nb * sizeof(PjBlock), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "large Zstandard frame: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    c->zpj_last_err = hf[PJ_ERR] | ((((u64)hf[PJ_TOTAL + 1] << 32) | hf[PJ_TOTAL]) != n ? 0x40000000u : 0u);
    if (hf[PJ_ERR] || (((u64)hf[PJ_TOTAL + 1] << 32) | hf[PJ_TOTAL]) != n) return ZPK_OK;          // irregular, or the sizes do not add up
    auto init = [&](u32 b0, u32 b1, hipStream_t s2) {
        hipLaunchKernelGGL(k_zpj_init, dim3(b1 - b0), dim3(256), 0, s2, (const ZpjBlock*)ZB, (const PjBlock*)B, b0, (u32)nb, (const u64*)c->d_pj_recs, (const u64*)c->d_zpj_pos,
                           (const u32*)c->d_pj_masks, S, n, c->d_pj_flags);
    };
    return pj_finish(c, d, hb, ZPK_PJ_CHUNK_BLOCKS * PJ_BLOCK / ZPJ_BLOCK, arena_off + lit_total, init, false, d_out ? d_out : c->d_dst, dst_ptr, result, redo);
}

// the frames of entries [g0, g1) of `be` as one device batch; redo[k] = 1: entry k takes the serial path after all
static int decode_big_group(zpk_codec* c, const u8* archive, const zpk_decode_desc* desc, const BigEntry* be, u64 g0, u64 g1,
                            const std::vector<BigSub>& subs, uint8_t* const* dst_ptrs, zpk_decode_result* results, u8* redo)
{
    const u64 ng = g1 - g0;
    u64 nsub = 0;
    for (u64 k = g0; k < g1; k++) nsub += be[k].nsub;
    std::vector<u64> coff(ng + 1), ooff(ng + 1);
    std::vector<zpk_span> spans(ng);
    std::vector<u64> h_hash(ng);
    std::vector<zpk_decode_desc> hd(nsub);
    std::vector<zpk_decode_result> hr(nsub);
    std::vector<const uint8_t*> cptr(ng);
    u64 ct = 0, ot = 0, part_blocks = 0, j = 0;
    int has_zstd = 0, has_lz4 = 0;
    for (u64 k = 0; k < ng; k++) {
        const BigEntry& E = be[g0 + k];
        const zpk_decode_desc& d = desc[E.idx];
        coff[k] = ct; ooff[k] = ot; cptr[k] = archive + d.src_offset;
        for (u64 f = 0; f < E.nsub; f++, j++) {
            const BigSub& S = subs[E.first_sub + f];
            zpk_decode_desc& x = hd[j];
            x.src_offset = ct + S.src_off; x.comp_size = S.comp; x.uncomp_size = S.size; x.expect_hash = 0;
            x.dst_offset = ot + S.out_off; x.dst_capacity = S.size; x.method = d.method; x.flags = ZPK_DF_SKIP_HASH;
        }
        spans[k].off = ot; spans[k].len = d.uncomp_size; spans[k].part_base = part_blocks;
        part_blocks += xxh3_span_blocks(d.uncomp_size);
        ct += (d.comp_size + 255) & ~255ull; ot += (d.uncomp_size + 255) & ~255ull;
        if (d.method == ZPK_METHOD_ZSTD) has_zstd = 1;
        if (d.method == ZPK_METHOD_LZ4) has_lz4 = 1;
    }
    coff[ng] = ct; ooff[ng] = ot;
    int rc;
    if ((rc = grow(c, (void**)&c->d_src, &c->src_cap, ct + ZPK_SRC_SLACK)) || (rc = grow(c, (void**)&c->d_dst, &c->dst_cap, ot + 16)) ||
        (rc = grow(c, &c->d_desc, &c->desc_cap, nsub * sizeof(zpk_decode_desc))) ||
        (rc = grow(c, &c->d_res, &c->res_cap, nsub * sizeof(zpk_decode_result)))) return rc;
    hipError_t e = hipSuccess;
    if (ng == 1) e = hipMemcpyAsync(c->d_src, cptr[0], desc[be[g0].idx].comp_size, hipMemcpyHostToDevice, c->stream);
    else {
        const int grc = h2d_gather(c, c->d_src, ct, ng, cptr.data(), [&](u64 k) { return coff[k]; }, [&](u64 k) { return (u64)desc[be[g0 + k].idx].comp_size; }, e, c->stream);
        if (grc) return grc;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_desc, hd.data(), nsub * sizeof(zpk_decode_desc), hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "H2D: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    c->zstd_hint = has_zstd; c->lz4_hint = has_lz4;
    // (image size = the staged bytes + 1: the last frame still passes the `offset + comp_size < file_size` guard of :331)
    rc = decode_launch(c, c->d_src, ct + 1, c->d_src, c->d_src + ct + ZPK_SRC_READ_SLACK, (const zpk_decode_desc*)c->d_desc, nsub,
                       c->d_dst, ot, (zpk_decode_result*)c->d_res, c->stream);
    if (rc) return rc;
    if ((rc = xxh3_spans_launch(c, c->d_dst, spans.data(), ng, part_blocks, h_hash.data(), c->stream))) return rc;
    e = hipMemcpyAsync(hr.data(), c->d_res, nsub * sizeof(zpk_decode_result), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "decode: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    j = 0;
    for (u64 k = 0; k < ng; k++) {
        const BigEntry& E = be[g0 + k];
        const zpk_decode_desc& d = desc[E.idx];
        bool ok = true;
        for (u64 f = 0; f < E.nsub; f++, j++) if (hr[j].status != 0 || hr[j].produced != subs[E.first_sub + f].size) ok = false;
        redo[g0 + k] = ok ? 0 : 1;
        if (!ok) continue;
        zpk_decode_result& r = results[E.idx];
        r.hash = h_hash[k]; r.produced = d.uncomp_size; r.detail = 0;
        r.status = ((d.flags & ZPK_DF_SKIP_HASH) || r.hash == d.expect_hash) ? 0 : 15;          // ZPACK_ERROR_FILE_HASH_MISMATCH, lib/zpack_read.c:467
        c->big_last[0]++; c->big_last[1] += (u32)E.nsub;
    }
    std::vector<uint8_t*> optr(ng);
    for (u64 k = 0; k < ng; k++) optr[k] = dst_ptrs[be[g0 + k].idx];
    rc = d2h_scatter(c, c->d_dst, ot, ng, optr.data(), [&](u64 k) { return ooff[k]; }, [&](u64 k) { return redo[g0 + k] ? 0ull : (u64)desc[be[g0 + k].idx].uncomp_size; }, e);
    if (rc) return rc;
    if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "D2H: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
    return ZPK_OK;
}

static int decode_batch_host_plain(zpk_codec* c, const uint8_t* archive, uint64_t archive_size, const zpk_decode_desc* desc, uint64_t n,
                                   uint8_t* const* dst_ptrs, zpk_decode_result* results);

int zpk_codec_decode_batch_host(zpk_codec* c, const uint8_t* archive, uint64_t archive_size, const zpk_decode_desc* desc, uint64_t n,
                                uint8_t* const* dst_ptrs, zpk_decode_result* results)
{
    if (!c || (n && (!desc || !results || !dst_ptrs))) return ZPK_E_INVALID;
    if (n == 0) return ZPK_OK;
    CodecLock lk(c);
    HIPCHK(c, hipSetDevice(c->device));
    c->big_last[0] = c->big_last[1] = 0;
    // ---- which entries are sequences of frames worth decoding frame-parallel ----
    std::vector<BigEntry> be;
    std::vector<BigSub> subs;
    struct PjEntry { u64 idx; std::vector<PjBlock> blocks; int independent; std::vector<ZpjBlock> zblocks; u64 slots, lit_total; };
    std::vector<PjEntry> pj;                                                          // large single frames (lz4_pj.h, zstd_pj.h)
    if (archive && c->dec_split_min != ~0ull) {
        try {
            for (u64 i = 0; i < n; i++) {
                const zpk_decode_desc& d = desc[i];
                if (d.uncomp_size < c->dec_split_min || d.uncomp_size > ZPK_HOST_CHUNK_BYTES || d.method > ZPK_METHOD_LZ4) continue;
                // every guard of lib/zpack_read.c:328-348 must pass: an entry that earns a verdict there gets it from the usual path
                if (!d.comp_size || d.src_offset > archive_size || d.comp_size >= archive_size - d.src_offset || d.dst_capacity < d.uncomp_size) continue;
                const size_t s0 = subs.size();
                bool ok = false;
                if (d.method == ZPK_METHOD_NONE) {
                    if (d.comp_size == d.uncomp_size && d.uncomp_size > ZPK_ENC_PIECE) {
                        for (u64 o = 0; o < d.uncomp_size; o += ZPK_ENC_PIECE) {
                            const u64 len = d.uncomp_size - o < ZPK_ENC_PIECE ? d.uncomp_size - o : (u64)ZPK_ENC_PIECE;
                            subs.push_back(BigSub{ o, len, o, len });
                        }
                        ok = true;
                    }
                } else if (d.method == ZPK_METHOD_LZ4) {
                    ok = walk_lz4_frames(archive + d.src_offset, d.comp_size, d.uncomp_size, subs);
                    if (!ok) {                                                        // ONE frame (what the reference writes): block-parallel
                        PjEntry P; P.idx = i; P.independent = 0; P.slots = P.lit_total = 0;
                        if (walk_lz4_single(archive + d.src_offset, d.comp_size, d.uncomp_size, P.blocks, P.independent)) pj.push_back(std::move(P));
                    }
                }
                else {
                    ok = walk_zstd_frames(archive + d.src_offset, d.comp_size, d.uncomp_size, subs);
                    if (!ok) {                                                        // ONE frame (what the reference writes): block-parallel
                        PjEntry P; P.idx = i; P.independent = 0; P.slots = P.lit_total = 0;
                        if (walk_zstd_single(archive + d.src_offset, d.comp_size, d.uncomp_size, P.zblocks, P.slots, P.lit_total)) pj.push_back(std::move(P));
                    }
                }
                if (ok) be.push_back(BigEntry{ i, (u64)s0, (u64)(subs.size() - s0) });
            }
        } catch (...) { be.clear(); subs.clear(); pj.clear(); }                       // out of host memory for the plan: the usual path
    }
    if (be.empty() && pj.empty()) {
        const int rc = decode_batch_host_plain(c, archive, archive_size, desc, n, dst_ptrs, results);
        c->zstd_hint = -1; c->lz4_hint = -1;
        return rc;
    }
    int rc = ZPK_OK;
    try {
        std::vector<u8> redo(be.size(), 0), is_big(n, 0);
        for (u64 g0 = 0; g0 < be.size() && rc == ZPK_OK; ) {                          // groups by the size of their output
            u64 g1 = g0, out = 0;
            while (g1 < be.size() && (g1 == g0 || out + desc[be[g1].idx].uncomp_size <= ZPK_HOST_CHUNK_BYTES)) { out += (desc[be[g1].idx].uncomp_size + 255) & ~255ull; g1++; }
            rc = decode_big_group(c, archive, desc, be.data(), g0, g1, subs, dst_ptrs, results, redo.data());
            g0 = g1;
        }
        // ---- everything else, and the entries a frame of which did not decode, through the usual path ----
        for (u64 k = 0; k < be.size(); k++) if (!redo[k]) is_big[be[k].idx] = 1;
        // Which single frames go block-parallel.  One at a time, each fills the chip: a fixed cost + its bytes at ~12 GiB/s — while the
        // entries of the usual batch all run side by side, one wave each: a batch of a hundred 3 MiB entries is done in the time of ONE
        // of them there.  The batch's time is (the block-parallel entries, one after the
        // other) + (the longest one-wave entry left): the largest entries go block-parallel as long as that sum shrinks.
        if (!pj.empty()) {
            // (measured, tools/mid_entry_rate.py + big_frame_rate.py: one wave 0.15 GiB/s LZ4 — ~1 GiB/s when the entry did not compress —,
            // 0.031 GiB/s Zstandard; block-parallel 0.6 ms + 12 GiB/s LZ4, 4.2 ms + 12 GiB/s Zstandard)
            auto wave_ms = [&](u64 i) {
                const double mib = (double)desc[i].uncomp_size / (1 << 20);
                const bool stored_like = desc[i].comp_size >= desc[i].uncomp_size - desc[i].uncomp_size / 16;
                return mib / 1.024 / (desc[i].method == ZPK_METHOD_LZ4 ? (stored_like ? 1.0 : 0.15) : (stored_like ? 0.9 : 0.031));
            };
            auto pj_ms = [&](u64 i) { return (desc[i].method == ZPK_METHOD_LZ4 ? 0.6 : 4.2) + (double)desc[i].uncomp_size / (1 << 20) / 12.0 / 1.024; };
            std::sort(pj.begin(), pj.end(), [&](const PjEntry& a, const PjEntry& b) { return wave_ms(a.idx) > wave_ms(b.idx); });
            double other = 0;                                                         // the longest entry that is not a candidate at all
            { std::vector<u8> cand(n, 0); for (auto& P : pj) cand[P.idx] = 1;
              for (u64 i = 0; i < n; i++) if (!cand[i] && desc[i].method != ZPK_METHOD_NONE && desc[i].uncomp_size >= (64u << 10)) { const double t = wave_ms(i); if (t > other) other = t; } }
            std::vector<double> tk(pj.size() + 1);
            double best = 1e300, acc = 0;
            for (size_t k = 0; k <= pj.size(); k++) {                                 // the first k block-parallel
                const double rest_ms = k < pj.size() ? wave_ms(pj[k].idx) : 0.0;
                tk[k] = acc + (rest_ms > other ? rest_ms : other);
                if (tk[k] < best) best = tk[k];
                if (k < pj.size()) acc += pj_ms(pj[k].idx);
            }
            size_t keep = pj.size();                                                  // (the estimates are rough: as many as come within 10 % of the best)
            while (keep > 0 && tk[keep] > 1.1 * best) keep--;
            pj.resize(keep);
        }
        for (u64 k = 0; k < pj.size() && rc == ZPK_OK; k++) {                         // one large frame at a time: each fills the chip
            u8 again = 1;
            if (desc[pj[k].idx].method == ZPK_METHOD_LZ4)
                rc = decode_big_lz4_single(c, archive, desc[pj[k].idx], pj[k].blocks, pj[k].independent, dst_ptrs[pj[k].idx], results[pj[k].idx], again);
            else
                rc = decode_big_zstd_single(c, archive, desc[pj[k].idx], pj[k].zblocks, pj[k].slots, pj[k].lit_total, dst_ptrs[pj[k].idx], results[pj[k].idx], again);
            if (rc == ZPK_OK && !again) is_big[pj[k].idx] = 1;
        }
        std::vector<u64> rest;
        for (u64 i = 0; i < n; i++) if (!is_big[i]) rest.push_back(i);
        if (rc == ZPK_OK && !rest.empty()) {
            std::vector<zpk_decode_desc> rd(rest.size());
            std::vector<uint8_t*> rp(rest.size());
            std::vector<zpk_decode_result> rr(rest.size());
            for (u64 k = 0; k < rest.size(); k++) { rd[k] = desc[rest[k]]; rp[k] = dst_ptrs[rest[k]]; }
            const u32 keep0 = c->big_last[0], keep1 = c->big_last[1];
            rc = decode_batch_host_plain(c, archive, archive_size, rd.data(), rest.size(), rp.data(), rr.data());
            c->big_last[0] = keep0; c->big_last[1] = keep1;
            if (rc == ZPK_OK) for (u64 k = 0; k < rest.size(); k++) results[rest[k]] = rr[k];
        }
    } catch (...) { rc = ZPK_E_NOMEM; }
    c->zstd_hint = -1; c->lz4_hint = -1;
    return rc;
}

// ONE entry whose compressed bytes are ON THE DEVICE, decoded into device memory (round 5; the device-pointer form of what
// zpk_codec_decode_batch_host does for a large single frame).  desc and result are HOST memory; the call returns when the entry is
// decoded and verified.  A large entry that is one frame of the reference writer is decoded block-parallel (lz4_pj.h / zstd_pj.h): its
// compressed bytes come to the host once, into a pinned buffer, for the walk over the block headers (2.6 ms for 130 MiB; a walk on
// the device is a chain of dependent loads of about the same length), everything else stays on the device.  Any other entry — and any
// entry the block-parallel path does not finish — is decoded by the one-wave kernels, exactly as zpk_codec_decode_batch_device would.
int zpk_codec_decode_big_device(zpk_codec* c, const uint8_t* d_archive, uint64_t archive_size, const zpk_decode_desc* desc,
                                uint8_t* d_dst, uint64_t dst_size, zpk_decode_result* result)
{
    if (!c || !desc || !result || !d_archive) return ZPK_E_INVALID;
    CodecLock lk(c);
    HIPCHK(c, hipSetDevice(c->device));
    c->big_last[0] = c->big_last[1] = 0;
    const zpk_decode_desc d = *desc;
    u8 redo = 1;
    int rc = ZPK_OK;
    const bool guards = d.comp_size && d.src_offset <= archive_size && d.comp_size < archive_size - d.src_offset && d.dst_capacity >= d.uncomp_size &&
                        d.dst_offset <= dst_size && d.uncomp_size <= dst_size - d.dst_offset;
    if (guards && c->dec_split_min != ~0ull && d.uncomp_size >= c->dec_split_min && d.uncomp_size <= ZPK_HOST_CHUNK_BYTES &&
        (d.method == ZPK_METHOD_LZ4 || d.method == ZPK_METHOD_ZSTD) && d_dst) {
        bool have = c->h_bigsrc_cap >= d.comp_size;
        if (!have) {
            if (c->h_bigsrc) { (void)hipHostFree(c->h_bigsrc); c->h_bigsrc = nullptr; c->h_bigsrc_cap = 0; }
            const u64 want = d.comp_size + d.comp_size / 4 + 4096;
            if (hipHostMalloc((void**)&c->h_bigsrc, want, hipHostMallocDefault) == hipSuccess) { c->h_bigsrc_cap = want; have = true; }
            else { (void)hipGetLastError(); c->h_bigsrc = nullptr; }
        }
        if (have) {
            hipError_t e = hipMemcpyAsync(c->h_bigsrc, d_archive + d.src_offset, d.comp_size, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { snprintf(c->err, sizeof(c->err), "large entry: %s", hipGetErrorString(e)); return ZPK_E_LAUNCH; }
            try {
                zpk_decode_result r; memset(&r, 0, sizeof(r));
                if (d.method == ZPK_METHOD_LZ4) {
                    std::vector<PjBlock> blocks; int independent = 0;
                    if (walk_lz4_single(c->h_bigsrc, d.comp_size, d.uncomp_size, blocks, independent))
                        rc = decode_big_lz4_single(c, nullptr, d, blocks, independent, nullptr, r, redo, d_archive, d_dst + d.dst_offset);
                } else {
                    std::vector<ZpjBlock> zb; u64 slots = 0, lit_total = 0;
                    if (walk_zstd_single(c->h_bigsrc, d.comp_size, d.uncomp_size, zb, slots, lit_total))
                        rc = decode_big_zstd_single(c, nullptr, d, zb, slots, lit_total, nullptr, r, redo, d_archive, d_dst + d.dst_offset);
                }
                if (rc != ZPK_OK) return rc;
                if (!redo) { *result = r; return ZPK_OK; }
            } catch (...) { redo = 1; }
        }
    }
    // ---- the one-wave kernels (every verdict is theirs) ----
    if (!c->d_big1 && hipMalloc((void**)&c->d_big1, 512) != hipSuccess) { c->d_big1 = nullptr; (void)hipGetLastError(); return ZPK_E_NOMEM; }
    zpk_decode_desc* const dd = (zpk_decode_desc*)c->d_big1;
    zpk_decode_result* const dr = (zpk_decode_result*)(c->d_big1 + 256);
    HIPCHK(c, hipMemcpyAsync(dd, &d, sizeof(d), hipMemcpyHostToDevice, c->stream));
    c->zstd_hint = -1; c->lz4_hint = -1;
    const u32 keep0 = c->big_last[0], keep1 = c->big_last[1];
    if ((rc = decode_launch(c, d_archive, archive_size, d_archive, d_archive + archive_size, dd, 1, d_dst, dst_size, dr, c->stream))) return rc;
    c->big_last[0] = keep0; c->big_last[1] = keep1;
    HIPCHK(c, hipMemcpyAsync(result, dr, sizeof(*result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZPK_OK;
}

static int decode_batch_host_plain(zpk_codec* c, const uint8_t* archive, uint64_t archive_size, const zpk_decode_desc* desc, uint64_t n,
                                   uint8_t* const* dst_ptrs, zpk_decode_result* results)
{
    if (n == 0) return ZPK_OK;
    zpk_decode_desc* hd = (zpk_decode_desc*)malloc(n * sizeof(zpk_decode_desc));
    if (!hd) return ZPK_E_NOMEM;
    int rc = ZPK_OK;
    u8* gathered = nullptr; u64 gathered_cap = 0;
    for (u64 first = 0; first < n && rc == ZPK_OK; ) {
        // ---- cut a sub-batch by the size of its output slots ----
        u64 out_total = 0, cnt = 0, lo = ~0ull, hi = 0, comp_sum = 0;
        int has_zstd = 0, has_lz4 = 0;
        while (first + cnt < n) {
            const zpk_decode_desc& d = desc[first + cnt];
            // stored entries copy exactly uncomp_size bytes (lib/zpack_read.c:366); the decoders may fill all of max_size
            const u64 slot_bytes = d.method == ZPK_METHOD_NONE && d.uncomp_size < d.dst_capacity ? d.uncomp_size : d.dst_capacity;
            const u64 slot = (slot_bytes + 255) & ~255ull;
            if (cnt && (out_total + slot > ZPK_HOST_CHUNK_BYTES || cnt >= 0x7FFFFFF0ull)) break;
            hd[first + cnt] = d;
            hd[first + cnt].dst_offset = out_total;
            // (the device still judges `max_size < uncomp_size` on the caller's capacity: dst_capacity stays as given
            // except for stored entries, whose slot is the smaller of the two and whose guard passed or failed on the host values)
            if (d.method == ZPK_METHOD_NONE && d.dst_capacity >= d.uncomp_size) hd[first + cnt].dst_capacity = slot_bytes;
            out_total += slot;
            const bool ok = d.comp_size && d.src_offset <= archive_size && d.comp_size <= archive_size - d.src_offset;
            if (ok) { if (d.src_offset < lo) lo = d.src_offset; if (d.src_offset + d.comp_size > hi) hi = d.src_offset + d.comp_size; comp_sum += d.comp_size; }
            if (d.method == ZPK_METHOD_ZSTD) has_zstd = 1;
            if (d.method == ZPK_METHOD_LZ4) has_lz4 = 1;
            cnt++;
        }
        if (lo > hi) { lo = 0; hi = 0; }
        c->zstd_hint = has_zstd; c->lz4_hint = has_lz4;
        const u8* image = archive; u64 image_size = archive_size;
        if (hi - lo > 2 * comp_sum + (1u << 20)) {
            // sparse picks out of a large archive: stage only the payloads.  Entries that pass the reference's offset guard
            // (lib/zpack_read.c:331) are packed behind each other, followed by one pad byte so that the guard still passes;
            // the others get an offset that still fails it — the device evaluates the same guards in the same order.
            const u64 total = comp_sum + 1;
            if (total + 1 > gathered_cap) { free(gathered); gathered = (u8*)malloc(total + 1); gathered_cap = gathered ? total + 1 : 0; }
            if (!gathered) { rc = ZPK_E_NOMEM; break; }
            u64 pos = 0;
            for (u64 i = first; i < first + cnt; i++) {
                const zpk_decode_desc& d = desc[i];
                const bool in_image = d.comp_size && d.src_offset <= archive_size && d.comp_size <= archive_size - d.src_offset;
                const bool passes = in_image && d.src_offset + d.comp_size < archive_size;
                if (passes) { memcpy(gathered + pos, archive + d.src_offset, d.comp_size); hd[i].src_offset = pos; pos += d.comp_size; }
                else hd[i].src_offset = total + 1;
            }
            gathered[pos] = 0;
            image = gathered; image_size = total; lo = 0; hi = pos;
        }
        bool piped = false;
        rc = decode_host_pipelined(c, image, image_size, lo, hi, hd + first, desc + first, cnt, out_total, dst_ptrs + first, results + first, piped);
        if (rc == ZPK_OK && !piped)
            rc = decode_host_chunk(c, image, image_size, lo, hi, hd + first, desc + first, cnt, out_total, dst_ptrs + first, results + first);
        first += cnt;
    }
    c->zstd_hint = -1; c->lz4_hint = -1;
    free(gathered);
    free(hd);
    return rc;
}

int zpk_codec_hash_batch_device(zpk_codec* c, const uint8_t* src, const uint64_t* offsets, const uint64_t* sizes, uint64_t n,
                                uint64_t* hashes, void* stream)
{
    if (!c) return ZPK_E_INVALID;
    if (n == 0) return ZPK_OK;
    CodecLock lk(c);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    hipLaunchKernelGGL(k_hash, dim3((u32)((n + 3) / 4)), dim3(256), 0, st, src, offsets, sizes, n, hashes);
    HIPCHK(c, hipGetLastError());
    return ZPK_OK;
}

int zpk_codec_hash_host(zpk_codec* c, const uint8_t* data, uint64_t size, uint64_t* hash)
{
    if (!c || !hash) return ZPK_E_INVALID;
    CodecLock lk(c);
    HIPCHK(c, hipSetDevice(c->device));
    int rc;
    if ((rc = grow(c, (void**)&c->d_src, &c->src_cap, size + 16)) || (rc = grow(c, &c->d_res, &c->res_cap, 64))) return rc;
    u64 meta[3] = { 0, size, 0 };
    if (size) HIPCHK(c, hipMemcpyAsync(c->d_src, data, size, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_res, meta, sizeof(meta), hipMemcpyHostToDevice, c->stream));
    u64* m = (u64*)c->d_res;
    hipLaunchKernelGGL(k_hash, dim3(1), dim3(64), 0, c->stream, c->d_src, m, m + 1, (u64)1, m + 2);
    HIPCHK(c, hipMemcpyAsync(hash, m + 2, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZPK_OK;
}

// debugging: copy the per-entry phase timing words (8 x u64 per entry; needs ZPK_DEBUG_TIMING=1) to the host
int zpk_codec_debug_read(zpk_codec* c, void* host, uint64_t bytes)
{
    if (!c || !c->d_dbg || bytes > c->dbg_cap) return ZPK_E_INVALID;      // d_dbg exists only in a -DZPK_DEVELOPER build
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(host, c->d_dbg, bytes, hipMemcpyDeviceToHost));
    return ZPK_OK;
}

// counters of the most recent decode batch (synchronises the device): out[0..2] = entries per work list
// (none, zstd, lz4), out[3] = Zstandard entries finished on pre-decoded sequences, out[4] = by the fused decoder
int zpk_codec_decode_stats(zpk_codec* c, uint32_t out[8])
{
    if (!c || !out) return ZPK_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    u32 h[N_COUNTERS];
    HIPCHK(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    if (c->totals_valid) memcpy(h, c->host_totals, sizeof(h));          // a pipelined host batch: the sum over its launches
    out[0] = h[L_NONE]; out[1] = h[L_ZSTD]; out[2] = h[L_LZ4] + h[C_LZ4_LEFT] /* both LZ4 lists */; out[3] = h[C_ZSTD_TWO_STAGE]; out[4] = h[C_ZSTD_FUSED];
    out[5] = h[ZF_WATCHDOG_WORD]; out[6] = h[ZF_WATCHDOG_WORD + 1]; out[7] = h[13];
    if (c->fell_back_fused) out[7] |= 0x80000000u;          // the batch could not get its sequence arena: fused decoder only
    ZPK_DEV(if (getenv("ZPK_TRACE")) fprintf(stderr, "[zpk] fse marked %u, pass-0 failures %u, last failure status/rc %08x\n", h[13], h[14], h[15]);)
    return ZPK_OK;
}

// out[0], out[1] = LZ4 / Zstandard entries of the most recent decode batch whose first decode ran out of its time budget and that
// were decoded again by the retry launches (expected 0 on an idle GPU)
int zpk_codec_decode_stats2(zpk_codec* c, uint32_t out[16])
{
    if (!c || !out) return ZPK_E_INVALID;
    CodecLock lk(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    u32 h[N_COUNTERS];
    HIPCHK(c, hipMemcpy(h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    if (c->totals_valid) memcpy(h, c->host_totals, sizeof(h));
    memset(out, 0, 16 * sizeof(uint32_t));
    out[0] = h[C_RETRY_LZ4]; out[1] = h[C_RETRY_ZSTD];
    out[2] = 0; out[3] = h[C_LZ4_LEFT]; out[4] = 0;      // [3]: LZ4 entries that are mostly runs, decoded by k_lz4_left (the two-stage path of round 4 is gone: [2], [4] read 0)
    out[5] = c->big_last[0]; out[6] = c->big_last[1];
    out[7] = c->zpj_last_err;                              // why the most recent large Zstandard frame was NOT finished block-parallel (0: it was, or none came)
    return ZPK_OK;
}

// developer aid: read back part of the Zstandard sequence arena (what = 0, byte offset = the entry's dst_offset
// rounded up to 8) or of the per-entry marks (what = 1, u32 per entry) of the most recent decode batch
int zpk_codec_debug_fetch(zpk_codec* c, int what, uint64_t offset, void* host, uint64_t bytes)
{
    if (!c || !host) return ZPK_E_INVALID;
    const u8* base = what == 0 ? (const u8*)c->d_zarena : (const u8*)c->d_zstate;
    const u64 cap = what == 0 ? c->zarena_cap : c->zstate_cap;
    if (!base || offset > cap || bytes > cap - offset) return ZPK_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    HIPCHK(c, hipMemcpy(host, base + offset, bytes, hipMemcpyDeviceToHost));
    return ZPK_OK;
}

int zpk_codec_set_option(zpk_codec* c, int option, int value)
{
    if (!c) return ZPK_E_INVALID;
    CodecLock lk(c);
    if (option == ZPK_OPT_ORDER_FAST_LAST) { c->order_fast_last = value ? 1 : 0; return ZPK_OK; }
    if (option == ZPK_OPT_ORDER_MIN) { if (value < 0) return ZPK_E_INVALID; c->order_min = c->enc_order_min = value == 0 ? ~0ull : (u64)value; return ZPK_OK; }
    if (option == ZPK_OPT_DEC_SPLIT_MIN) { if (value < 0) return ZPK_E_INVALID; c->dec_split_min = value == 0 ? ~0ull : (u64)value; return ZPK_OK; }
    if (option == ZPK_OPT_ENC_SPLIT_MIN) { if (value < 0) return ZPK_E_INVALID; c->enc_split_min = value == 0 ? ~0ull : (u64)value; return ZPK_OK; }
    return ZPK_E_INVALID;
}

int zpk_codec_set_profiling(zpk_codec* c, int enabled)
{
    if (!c) return ZPK_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (enabled)
        for (int i = 0; i < ZPK_K_COUNT; i++) for (int j = 0; j < 2; j++)
            if (!c->kev[i][j]) HIPCHK(c, hipEventCreate(&c->kev[i][j]));
    c->profiling = enabled ? 1 : 0;
    return ZPK_OK;
}

int zpk_codec_kernel_ms(zpk_codec* c, int which, float* ms)
{
    if (!c || !ms || which < 0 || which >= ZPK_K_COUNT || !c->kev[which][1]) return ZPK_E_INVALID;
    HIPCHK(c, hipEventSynchronize(c->kev[which][1]));
    HIPCHK(c, hipEventElapsedTime(ms, c->kev[which][0], c->kev[which][1]));
    return ZPK_OK;
}

int zpk_codec_timer_start(zpk_codec* c, void* stream)
{
    if (!c) return ZPK_E_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventRecord(c->ev0, stream ? (hipStream_t)stream : c->stream));
    return ZPK_OK;
}

int zpk_codec_timer_stop(zpk_codec* c, void* stream, float* elapsed_ms)
{
    if (!c || !elapsed_ms) return ZPK_E_INVALID;
    HIPCHK(c, hipEventRecord(c->ev1, stream ? (hipStream_t)stream : c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    HIPCHK(c, hipEventElapsedTime(elapsed_ms, c->ev0, c->ev1));
    return ZPK_OK;
}

}  // extern "C"

#include "zpk_encode.inc"
#include "zpk_stream.inc"
