// zpk_device.h — shared device-side helpers for the MI355X (gfx950) entry codec kernels.
// wave = 64 lanes everywhere; every helper that says "uniform" expects the same value in all lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/zpack_codec.h"

namespace zpk {

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t  i16;
typedef int32_t  i32;
typedef int64_t  i64;

// enum zpack_result values (include/zpack.h; reference lib/zpack.h:189-218)
enum : int {
    R_OK = 0,
    R_BUFFER_TOO_SMALL = 12,
    R_DECOMPRESS_FAILED = 13,
    R_COMPRESS_FAILED = 14,
    R_FILE_HASH_MISMATCH = 15,
    R_FILE_OFFSET_INVALID = 16,
    R_FILE_INCOMPLETE = 17,
    R_FILE_SIZE_INVALID = 18,
    R_COMP_METHOD_INVALID = 19,
};

// codec-internal decode verdicts (mapped to zpack_result by the entry kernels)
enum : int { D_OK = 0, D_MALFORMED = -1, D_TRUNCATED = -2, D_DST_FULL = -3 };

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// make a wave-uniform value live in an SGPR so the compiler emits scalar control flow
__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uni64(u64 v)
{
    u32 lo = uni((u32)v), hi = uni((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
template <typename T> __device__ __forceinline__ T* uni_ptr(T* p) { return (T*)uni64((u64)p); }

// CONVERGENCE HAZARD (hit on hardware, ROCm 7.2): two `if (lane == 0)` regions with no convergent
// operation between them get jump-threaded together by LLVM, also across a loop back edge; the loop
// then runs per-lane-class and a following readfirstlane no longer sees lane 0 (the wave re-processes
// stale data forever).  Every lane-0 region whose result is broadcast starts behind lane0_guard(): the
// wave barrier is a convergent no-op, and LLVM never duplicates a block that holds one.
__device__ __forceinline__ void lane0_guard() { __builtin_amdgcn_wave_barrier(); }

// Address spaces.  A `const u8*` that went through readfirstlane (uni_ptr) or a pointer select is a
// GENERIC pointer to the compiler, and every access through it becomes a FLAT instruction — an order of
// magnitude slower than ds_read for LDS and tied to both vmcnt and lgkmcnt (measured: the LZ4 walk ran
// ~1000 cycles per token on flat loads).  So: every helper below that takes a plain pointer means HBM
// (global) and says so to the compiler; LDS is only ever touched through array indexing on __shared__
// objects or through the lds_* helpers.
#define ZPK_GLOBAL __attribute__((address_space(1)))
#define ZPK_LDS __attribute__((address_space(3)))
struct __attribute__((packed, aligned(1))) pk16 { u16 v; };
struct __attribute__((packed, aligned(1))) pk32 { u32 v; };
struct __attribute__((packed, aligned(1))) pk64 { u64 v; };
struct __attribute__((packed, aligned(1))) u128 { u64 lo, hi; };

// unaligned global loads/stores: gfx950 runs in unaligned-access mode, these are single
// global_load/store_{ubyte,ushort,dword,dwordx2,dwordx4} instructions (checked in the ISA)
__device__ __forceinline__ u8   ld8(const u8* p)   { return *(const ZPK_GLOBAL u8*)p; }
__device__ __forceinline__ u16  ld16(const u8* p)  { return ((const ZPK_GLOBAL pk16*)p)->v; }
__device__ __forceinline__ u32  ld32(const u8* p)  { return ((const ZPK_GLOBAL pk32*)p)->v; }
__device__ __forceinline__ u64  ld64(const u8* p)  { return ((const ZPK_GLOBAL pk64*)p)->v; }
typedef u32 v4u32 __attribute__((ext_vector_type(4)));
typedef v4u32 __attribute__((aligned(1))) v4u32_u;
__device__ __forceinline__ u128 ld128(const u8* p)
{
    const v4u32 x = *(const ZPK_GLOBAL v4u32_u*)p;
    u128 r; r.lo = ((u64)x.y << 32) | x.x; r.hi = ((u64)x.w << 32) | x.z;
    return r;
}
__device__ __forceinline__ void st8(u8* p, u8 v)     { *(ZPK_GLOBAL u8*)p = v; }
__device__ __forceinline__ void st16(u8* p, u16 v)   { ((ZPK_GLOBAL pk16*)p)->v = v; }
__device__ __forceinline__ void st32(u8* p, u32 v)   { ((ZPK_GLOBAL pk32*)p)->v = v; }
__device__ __forceinline__ void st64(u8* p, u64 v)   { ((ZPK_GLOBAL pk64*)p)->v = v; }
__device__ __forceinline__ void st128(u8* p, u128 v)
{
    v4u32 x; x.x = (u32)v.lo; x.y = (u32)(v.lo >> 32); x.z = (u32)v.hi; x.w = (u32)(v.hi >> 32);
    *(ZPK_GLOBAL v4u32_u*)p = x;
}

// LDS accessors for code that only has a generic pointer to a __shared__ object
typedef const ZPK_LDS u8* lds_cp8;
__device__ __forceinline__ lds_cp8 to_lds(const u8* p) { return (lds_cp8)p; }
__device__ __forceinline__ u32 lds_ld8(lds_cp8 p)  { return (u32)*p; }
__device__ __forceinline__ u32 lds_ld16(lds_cp8 p) { return (u32)((const ZPK_LDS pk16*)p)->v; }
typedef ZPK_LDS u8* lds_p8;
__device__ __forceinline__ lds_p8 to_lds_rw(u8* p) { return (lds_p8)p; }
__device__ __forceinline__ void lds_st8(lds_p8 p, u8 v)    { *p = v; }
__device__ __forceinline__ void lds_st16(lds_p8 p, u16 v)  { ((ZPK_LDS pk16*)p)->v = v; }
__device__ __forceinline__ void lds_st32(lds_p8 p, u32 v)  { ((ZPK_LDS pk32*)p)->v = v; }
__device__ __forceinline__ void lds_st64(lds_p8 p, u64 v)  { ((ZPK_LDS pk64*)p)->v = v; }
__device__ __forceinline__ void lds_st128(lds_p8 p, u128 v)
{
    v4u32 x; x.x = (u32)v.lo; x.y = (u32)(v.lo >> 32); x.z = (u32)v.hi; x.w = (u32)(v.hi >> 32);
    *(ZPK_LDS v4u32_u*)p = x;
}
__device__ __forceinline__ u128 lds_ld128(lds_cp8 p)      // one ds_read_b128 at any byte offset
{
    const v4u32 x = *(const ZPK_LDS v4u32_u*)p;
    u128 r; r.lo = ((u64)x.y << 32) | x.x; r.hi = ((u64)x.w << 32) | x.z;
    return r;
}

// uniform byte / halfword / word reads: one lane-0 style load broadcast through an SGPR
__device__ __forceinline__ u32 uld8(const u8* p) { return uni((u32)ld8(p)); }
__device__ __forceinline__ u32 uld16(const u8* p) { return uni((u32)ld16(p)); }
__device__ __forceinline__ u32 uld32(const u8* p) { return uni(ld32(p)); }

__device__ __forceinline__ u64 shfl_xor64(u64 v, int mask)
{
    u32 lo = (u32)__shfl_xor((int)(u32)v, mask, 64);
    u32 hi = (u32)__shfl_xor((int)(u32)(v >> 32), mask, 64);
    return ((u64)hi << 32) | lo;
}

// inclusive prefix sum over the wave with DPP (row shifts + the gfx9 row broadcasts): six v_add_u32_dpp,
// no LDS traffic — a ds_bpermute ladder costs ~5x the instructions
__device__ __forceinline__ u32 wave_scan_add(u32 v)
{
#ifdef ZPK_NO_DPP
    const int lane_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u32 y = (u32)__shfl_up((int)v, d, 64); if (lane_ >= d) v += y; }
    return v;
#endif
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// lanes of one wave exchange data through global memory / LDS: the hardware keeps a wave's vector
// memory operations in order, so a wavefront-scope fence (a compiler barrier, no s_waitcnt) suffices.
__device__ __forceinline__ void wave_mem_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// Every parse loop polls this: a wave whose entry has used up its time budget gives the entry up (status
// DECOMPRESS_FAILED, detail 0xDEAD) instead of holding the GPU — malformed input or a decoder bug must never leave
// a wave spinning.  The budget is PROPORTIONAL TO THE ENTRY: ZPK_WATCHDOG_SECONDS of grace plus
// ZPK_WATCHDOG_TICKS_PER_BYTE ticks of the 100 MHz s_memrealtime clock per byte of (compressed + output capacity),
// i.e. a floor of 2 MB/s per wave — an order of magnitude below the slowest rate measured for one wave under full
// load (~20 MB/s, Zstandard FSE chain) — so a valid entry of ANY size finishes with the reference's verdict
// (lib/zpack_read.c:380,414-439 decode any size), and the verdict does not depend on how busy the GPU is.
// Polled every 256 steps.
#ifndef ZPK_WATCHDOG_SECONDS
#define ZPK_WATCHDOG_SECONDS 4
#endif
#ifndef ZPK_WATCHDOG_TICKS_PER_BYTE
#define ZPK_WATCHDOG_TICKS_PER_BYTE 50
#endif
__device__ __forceinline__ u64 watchdog_budget(u64 bytes)
{
    const u64 cap = ~0ull / (4 * ZPK_WATCHDOG_TICKS_PER_BYTE);                       // sizes are untrusted: no wrap
    return (u64)ZPK_WATCHDOG_SECONDS * 100000000ull + (bytes < cap ? bytes : cap) * ZPK_WATCHDOG_TICKS_PER_BYTE;
}
// A budget that runs out is NOT a verdict (round 3): the entry goes on a retry list and is decoded again, after the batch's
// kernels have drained, with ZPK_WATCHDOG_RETRY_SCALE times the budget — a floor of ~30 KB/s per wave, minutes for a 64 KiB
// entry: only a decoder that really does not terminate reports DECOMPRESS_FAILED / 0xDEAD.  scale 0 (developer builds,
// ZPK_WD_SCALE=0): the budget is spent at the first poll, every entry that polls takes the retry path.
#ifndef ZPK_WATCHDOG_RETRY_SCALE
#define ZPK_WATCHDOG_RETRY_SCALE 64u
#endif
struct Watchdog {
    u64 deadline; u32 tick; bool fired;
    __device__ __forceinline__ void arm(u64 bytes, u32 scale = 1u)
    {
        const u64 b = watchdog_budget(bytes), lim = scale ? (~0ull >> 2) / (u64)scale : ~0ull;        // (sizes are untrusted: the product saturates)
        deadline = __builtin_amdgcn_s_memrealtime() + (b < lim ? b : lim) * (u64)scale; tick = 0; fired = false;
    }
    __device__ __forceinline__ bool expired()
    {
        if (((++tick) & 255u) == 0 && __builtin_amdgcn_s_memrealtime() > deadline) fired = true;
        return fired;
    }
};

__device__ __forceinline__ u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ int highbit32(u32 v) { return 31 - __clz((int)v); }      // v != 0

}  // namespace zpk
