// xxh3_device.h — XXH3-64 (seed 0, default secret) as a wave64 primitive for gfx950, plus serial
// XXH32 / XXH64 for the frame-internal checksums of foreign LZ4F / Zstandard frames.
//
// Replaces XXH3_64bits of the reference (lib/zpack_read.c:466, lib/zpack_write.c:256).
//
// Long inputs (> 240 B): the 8 u64 accumulators are spread over the wave.  Lane l owns the 16-byte
// slot l of a 1 KiB block = stripe (l >> 2), accumulator pair q = (l & 3).  Within a block the 16
// stripe contributions to one accumulator only ADD, so they are summed across the 16 lanes that share
// q with DPP row rotations + v_permlane16/32_swap (no LDS); the per-block scramble then runs redundantly in every lane.  One block
// costs one 16 B/lane load (1 KiB per wave instruction, fully coalesced) and ~70 VALU/DPP instructions.
#pragma once
#include "zpk_device.h"

namespace zpk {

#define ZPK_P32_1 0x9E3779B1U
#define ZPK_P32_2 0x85EBCA77U
#define ZPK_P32_3 0xC2B2AE3DU
#define ZPK_P32_4 0x27D4EB2FU
#define ZPK_P32_5 0x165667B1U
#define ZPK_P64_1 0x9E3779B185EBCA87ULL
#define ZPK_P64_2 0xC2B2AE3D27D4EB4FULL
#define ZPK_P64_3 0x165667B19E3779F9ULL
#define ZPK_P64_4 0x85EBCA77C2B2AE63ULL
#define ZPK_P64_5 0x27D4EB2F165667C5ULL
#define ZPK_PMX_1 0x165667919E3779F9ULL
#define ZPK_PMX_2 0x9FB21C651E98DF25ULL

// default XXH3 secret (xxHash specification, "kSecret"), padded to a multiple of 8
__device__ __constant__ const u8 XXH3_SECRET[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};

__device__ __forceinline__ u64 sec64(int off) { return ld64(XXH3_SECRET + off); }
__device__ __forceinline__ u32 sec32(int off) { return ld32(XXH3_SECRET + off); }

__device__ __forceinline__ u64 mul128_fold64(u64 a, u64 b) { return (a * b) ^ __umul64hi(a, b); }
__device__ __forceinline__ u64 xxh64_avalanche(u64 h)
{
    h ^= h >> 33; h *= ZPK_P64_2; h ^= h >> 29; h *= ZPK_P64_3; h ^= h >> 32; return h;
}
__device__ __forceinline__ u64 xxh3_avalanche(u64 h) { h ^= h >> 37; h *= ZPK_PMX_1; h ^= h >> 32; return h; }
__device__ __forceinline__ u64 bswap64(u64 x) { return __builtin_bswap64(x); }

__device__ __forceinline__ u64 xxh3_mix16(const u8* in, int soff)
{
    return mul128_fold64(ld64(in) ^ sec64(soff), ld64(in + 8) ^ sec64(soff + 8));
}

// 0..240 bytes, serial (executed by whichever lanes call it; callers use one lane or the whole wave
// redundantly — every load is in bounds of [in, in+len))
__device__ inline u64 xxh3_short(const u8* in, u32 len)
{
    if (len == 0) return xxh64_avalanche(sec64(56) ^ sec64(64));
    if (len <= 3) {
        u32 c1 = ld8(in), c2 = ld8(in + (len >> 1)), c3 = ld8(in + len - 1);
        u32 combined = (c1 << 16) | (c2 << 24) | c3 | (len << 8);
        u64 flip = (u64)(sec32(0) ^ sec32(4));
        return xxh64_avalanche((u64)combined ^ flip);
    }
    if (len <= 8) {
        u32 a = ld32(in), b = ld32(in + len - 4);
        u64 flip = sec64(8) ^ sec64(16);
        u64 in64 = (u64)b + ((u64)a << 32);
        u64 h = in64 ^ flip;
        h ^= rotl64(h, 49) ^ rotl64(h, 24);
        h *= ZPK_PMX_2;
        h ^= (h >> 35) + len;
        h *= ZPK_PMX_2;
        return h ^ (h >> 28);
    }
    if (len <= 16) {
        u64 lo = ld64(in) ^ (sec64(24) ^ sec64(32));
        u64 hi = ld64(in + len - 8) ^ (sec64(40) ^ sec64(48));
        u64 acc = len + bswap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    if (len <= 128) {
        u64 acc = (u64)len * ZPK_P64_1;
        if (len > 32) {
            if (len > 64) {
                if (len > 96) {
                    acc += xxh3_mix16(in + 48, 96);
                    acc += xxh3_mix16(in + len - 64, 112);
                }
                acc += xxh3_mix16(in + 32, 64);
                acc += xxh3_mix16(in + len - 48, 80);
            }
            acc += xxh3_mix16(in + 16, 32);
            acc += xxh3_mix16(in + len - 32, 48);
        }
        acc += xxh3_mix16(in, 0);
        acc += xxh3_mix16(in + len - 16, 16);
        return xxh3_avalanche(acc);
    }
    u64 acc = (u64)len * ZPK_P64_1;
    u32 rounds = len / 16;
    for (u32 i = 0; i < 8; i++) acc += xxh3_mix16(in + 16 * i, 16 * (int)i);
    acc = xxh3_avalanche(acc);
    for (u32 i = 8; i < rounds; i++) acc += xxh3_mix16(in + 16 * i, 16 * (int)(i - 8) + 3);
    acc += xxh3_mix16(in + len - 16, 136 - 17);
    return xxh3_avalanche(acc);
}

// ---- wave-parallel long hash -------------------------------------------------------------------

struct Xxh3Wave {
    u64 a0, a1;        // acc[2q], acc[2q+1], q = lane & 3 (identical in the 16 lanes sharing q)
    u64 k0, k1;        // secret words of this lane's slot in a full block: secret[8*(l>>2) + 16q (+8)]
    u64 s0, s1;        // scramble keys secret[128 + 16q (+8)]

    __device__ __forceinline__ void init(int lane)
    {
        const int q = lane & 3, s = lane >> 2;
        const u64 init[8] = { ZPK_P32_3, ZPK_P64_1, ZPK_P64_2, ZPK_P64_3, ZPK_P64_4, ZPK_P32_2, ZPK_P64_5, ZPK_P32_1 };
        a0 = q == 0 ? init[0] : q == 1 ? init[2] : q == 2 ? init[4] : init[6];
        a1 = q == 0 ? init[1] : q == 1 ? init[3] : q == 2 ? init[5] : init[7];
        k0 = sec64(8 * s + 16 * q); k1 = sec64(8 * s + 16 * q + 8);
        s0 = sec64(128 + 16 * q);   s1 = sec64(128 + 16 * q + 8);
    }

    // contribution of one 16-byte slot (d0 = u64 index 2q of its stripe, d1 = index 2q+1)
    static __device__ __forceinline__ void slot(u64 d0, u64 d1, u64 key0, u64 key1, u64& c0, u64& c1)
    {
        u64 x0 = d0 ^ key0, x1 = d1 ^ key1;
        c0 = (u64)(u32)x0 * (u64)(u32)(x0 >> 32) + d1;      // acc[2q]   += product(2q)   + data(2q+1)
        c1 = (u64)(u32)x1 * (u64)(u32)(x1 >> 32) + d0;      // acc[2q+1] += product(2q+1) + data(2q)
    }

    // Sum over the 16 lanes that share q = lane & 3 (lanes q, q + 4, ... q + 60), result in all of them.  No LDS: inside a DPP row
    // two rotations (row_ror:4, row_ror:8) add up the four lanes q + 4i; across the four rows v_permlane16_swap / v_permlane32_swap
    // (gfx950) exchange whole rows / halves between a value and its copy, so that value + copy is the sum of the pair.  (Round 2 did
    // this with 16 ds_bpermute per 1 KiB block: ~100 LDS cycles per block on a CU whose LDS pipe the decoders keep 80 % busy.)
    template <int CTRL>
    static __device__ __forceinline__ u64 add_rotated(u64 c)
    {
        const u32 rl = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)c, CTRL, 0xf, 0xf, false);
        const u32 rh = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(c >> 32), CTRL, 0xf, 0xf, false);
        return c + (((u64)rh << 32) | rl);
    }
    static __device__ __forceinline__ u64 add_rows(u64 c)
    {
        const u32 lo = (u32)c, hi = (u32)(c >> 32);
        auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        const u64 s = (((u64)h16[0] << 32) | l16[0]) + (((u64)h16[1] << 32) | l16[1]);        // rows {0,1} and {2,3} summed pairwise
        const u32 slo = (u32)s, shi = (u32)(s >> 32);
        auto l32 = __builtin_amdgcn_permlane32_swap(slo, slo, false, false);
        auto h32 = __builtin_amdgcn_permlane32_swap(shi, shi, false, false);
        return (((u64)h32[0] << 32) | l32[0]) + (((u64)h32[1] << 32) | l32[1]);
    }
    // Which form is faster depends on what the kernel around it is short of (measured, round 3): k_lz4_wave is bound by vector
    // issue with LDS cycles to spare — there the 16 ds_bpermute are the cheaper ones (508 vs 498 GiB/s on text); the Zstandard execute
    // stage is short of LDS cycles and gains with the DPP / permlane form (13.1 -> 12.7 ms on C3).
    template <bool NO_LDS>
    static __device__ __forceinline__ void reduce16(u64& c0, u64& c1)
    {
        if (!NO_LDS) {
            #pragma unroll
            for (int m = 4; m < 64; m <<= 1) { c0 += shfl_xor64(c0, m); c1 += shfl_xor64(c1, m); }
        } else {
            c0 = add_rotated<0x124>(c0); c1 = add_rotated<0x124>(c1);         // row_ror:4
            c0 = add_rotated<0x128>(c0); c1 = add_rotated<0x128>(c1);         // row_ror:8
            c0 = add_rows(c0); c1 = add_rows(c1);
        }
    }

    __device__ __forceinline__ void scramble()
    {
        a0 = ((a0 ^ (a0 >> 47)) ^ s0) * ZPK_P32_1;
        a1 = ((a1 ^ (a1 >> 47)) ^ s1) * ZPK_P32_1;
    }

    // one full 1 KiB block whose 16 bytes for this lane are already in registers
    __device__ __forceinline__ void block(u128 d)
    {
        u64 c0, c1;
        slot(d.lo, d.hi, k0, k1, c0, c1);
        reduce16<false>(c0, c1);
        a0 += c0; a1 += c1;
        scramble();
    }

    // tail: `nstripes` (0..15) whole stripes at `p`, then the last stripe [end-64, end).
    // Lane group 15 is never used by a partial block, so it carries the last stripe.
    __device__ __forceinline__ u64 finish(const u8* p, u32 nstripes, const u8* end, u64 total_len, int lane)
    {
        const int q = lane & 3, s = lane >> 2;
        u64 c0 = 0, c1 = 0;
        if ((u32)s < nstripes) {
            u128 d = ld128(p + 16 * lane);
            slot(d.lo, d.hi, k0, k1, c0, c1);
        } else if (s == 15) {
            u128 d = ld128(end - 64 + 16 * q);
            slot(d.lo, d.hi, sec64(121 + 16 * q), sec64(121 + 16 * q + 8), c0, c1);
        }
        reduce16<false>(c0, c1);
        a0 += c0; a1 += c1;
        u64 t = mul128_fold64(a0 ^ sec64(11 + 16 * q), a1 ^ sec64(11 + 16 * q + 8));
        t += shfl_xor64(t, 1);
        t += shfl_xor64(t, 2);
        return xxh3_avalanche(total_len * ZPK_P64_1 + t);
    }
};

// The same state in 4 registers instead of 12: the per-lane secret words are re-read from the constant table for every block (for
// kernels whose occupancy is worth more than four cached loads per KiB)
struct Xxh3Lite {
    u64 a0, a1;
    __device__ __forceinline__ void init(int lane)
    {
        Xxh3Wave w; w.init(lane); a0 = w.a0; a1 = w.a1;
    }
    // `sec` = a copy of the 192-byte secret in LDS, or null (constant memory).  The constant-memory reads are vector loads: their
    // s_waitcnt vmcnt(0) also waits for the 1 KiB flush store issued just before — a full store round trip per flushed block.
    __device__ __forceinline__ void block(u128 d, int lane, lds_cp8 sec = nullptr)
    {
        const int q = lane & 3, s = lane >> 2;
        u64 k0, k1, s0, s1;
        if (sec) {
            k0 = ((const ZPK_LDS u64*)(sec + 8 * s + 16 * q))[0]; k1 = ((const ZPK_LDS u64*)(sec + 8 * s + 16 * q))[1];
            s0 = ((const ZPK_LDS u64*)(sec + 128 + 16 * q))[0];   s1 = ((const ZPK_LDS u64*)(sec + 128 + 16 * q))[1];
        } else { k0 = sec64(8 * s + 16 * q); k1 = sec64(8 * s + 16 * q + 8); s0 = sec64(128 + 16 * q); s1 = sec64(128 + 16 * q + 8); }
        u64 c0, c1;
        Xxh3Wave::slot(d.lo, d.hi, k0, k1, c0, c1);
        Xxh3Wave::reduce16<true>(c0, c1);
        a0 += c0; a1 += c1;
        a0 = ((a0 ^ (a0 >> 47)) ^ s0) * ZPK_P32_1;
        a1 = ((a1 ^ (a1 >> 47)) ^ s1) * ZPK_P32_1;
    }
    __device__ __forceinline__ u64 finish(const u8* p, u32 nstripes, const u8* end, u64 total_len, int lane)
    {
        Xxh3Wave w; w.init(lane); w.a0 = a0; w.a1 = a1;
        return w.finish(p, nstripes, end, total_len, lane);
    }
};

// XXH3_64bits(p, len) computed by one full wave (all 64 lanes must call; result uniform).
// `p` may have any alignment.
__device__ __forceinline__ u64 xxh3_64_wave(const u8* p, u64 len, int lane)
{
    if (len <= 240) {
        u64 h = 0;
        lane0_guard();
        if (lane == 0) h = xxh3_short(p, (u32)len);
        return uni64(h);
    }
    Xxh3Wave st;
    st.init(lane);
    const u64 nblocks = (len - 1) >> 10;
    const u8* q = p + 16 * lane;
    // Software prefetch, two pairs of blocks in turn: while one pair is reduced (a block costs ~60 vector instructions) the
    // loads of the other are in flight.  The loads are UNCONDITIONAL (past the end they re-read the last block) and no loaded
    // register is ever moved: the compiler counts outstanding loads exactly only then (s_waitcnt vmcnt(2)); with a guarded
    // load or a register rotation it waits for everything, i.e. for the loads it has just issued — a full memory round trip
    // per pair, which is what this loop used to cost (18 % of k_lz4_wave).
    if (nblocks > 0) {
        const u64 last = nblocks - 1;
        #define XXH3_LD(i) ld128(q + (((i) < last ? (i) : last) << 10))
        u128 a0 = XXH3_LD(0), a1 = XXH3_LD(1);
        for (u64 b = 0; b < nblocks; b += 4) {
            const u128 c0 = XXH3_LD(b + 2), c1 = XXH3_LD(b + 3);
            st.block(a0);
            if (b + 1 < nblocks) st.block(a1);
            a0 = XXH3_LD(b + 4); a1 = XXH3_LD(b + 5);
            if (b + 2 < nblocks) st.block(c0);
            if (b + 3 < nblocks) st.block(c1);
        }
        #undef XXH3_LD
    }
    const u32 nstripes = (u32)(((len - 1) - (nblocks << 10)) >> 6);
    return st.finish(p + (nblocks << 10), nstripes, p + len, len, lane);
}

// ---- serial XXH32 / XXH64 (foreign-frame checksums; single lane) --------------------------------

__device__ inline u32 xxh32_serial(const u8* p, u64 len, u32 seed)
{
    const u8* end = p + len;
    u32 h;
    if (len >= 16) {
        u32 v1 = seed + ZPK_P32_1 + ZPK_P32_2, v2 = seed + ZPK_P32_2, v3 = seed, v4 = seed - ZPK_P32_1;
        do {
            v1 = rotl32(v1 + ld32(p) * ZPK_P32_2, 13) * ZPK_P32_1;
            v2 = rotl32(v2 + ld32(p + 4) * ZPK_P32_2, 13) * ZPK_P32_1;
            v3 = rotl32(v3 + ld32(p + 8) * ZPK_P32_2, 13) * ZPK_P32_1;
            v4 = rotl32(v4 + ld32(p + 12) * ZPK_P32_2, 13) * ZPK_P32_1;
            p += 16;
        } while (p + 16 <= end);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + ZPK_P32_5;
    }
    h += (u32)len;
    while (p + 4 <= end) { h = rotl32(h + ld32(p) * ZPK_P32_3, 17) * ZPK_P32_4; p += 4; }
    while (p < end) { h = rotl32(h + (u32)ld8(p) * ZPK_P32_5, 11) * ZPK_P32_1; p++; }
    h ^= h >> 15; h *= ZPK_P32_2; h ^= h >> 13; h *= ZPK_P32_3; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ u64 xxh64_round(u64 acc, u64 in) { return rotl64(acc + in * ZPK_P64_2, 31) * ZPK_P64_1; }

__device__ inline u64 xxh64_serial(const u8* p, u64 len, u64 seed)
{
    const u8* end = p + len;
    u64 h;
    if (len >= 32) {
        u64 v1 = seed + ZPK_P64_1 + ZPK_P64_2, v2 = seed + ZPK_P64_2, v3 = seed, v4 = seed - ZPK_P64_1;
        do {
            v1 = xxh64_round(v1, ld64(p)); v2 = xxh64_round(v2, ld64(p + 8));
            v3 = xxh64_round(v3, ld64(p + 16)); v4 = xxh64_round(v4, ld64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = (h ^ xxh64_round(0, v1)) * ZPK_P64_1 + ZPK_P64_4;
        h = (h ^ xxh64_round(0, v2)) * ZPK_P64_1 + ZPK_P64_4;
        h = (h ^ xxh64_round(0, v3)) * ZPK_P64_1 + ZPK_P64_4;
        h = (h ^ xxh64_round(0, v4)) * ZPK_P64_1 + ZPK_P64_4;
    } else {
        h = seed + ZPK_P64_5;
    }
    h += len;
    while (p + 8 <= end) { h ^= xxh64_round(0, ld64(p)); h = rotl64(h, 27) * ZPK_P64_1 + ZPK_P64_4; p += 8; }
    if (p + 4 <= end) { h ^= (u64)ld32(p) * ZPK_P64_1; h = rotl64(h, 23) * ZPK_P64_2 + ZPK_P64_3; p += 4; }
    while (p < end) { h ^= (u64)ld8(p) * ZPK_P64_5; h = rotl64(h, 11) * ZPK_P64_1; p++; }
    return xxh64_avalanche(h);
}

}  // namespace zpk
