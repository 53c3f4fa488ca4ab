/*
 * lz4f_enc_oracle.c — LZ4 frame encoder with the frame layout the reference writer produces.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Stands in for LZ4F_compressBegin/Update/End as called at lib/zpack_write.c:199-211 (zeroed
 * LZ4F_preferences_t + level): header `04 22 4D 18 40 40 C0` (v1, block-linked, 64 KiB max block,
 * no checksums, no content size), per block `u32le size | bit31 = stored`, EndMark `00 00 00 00`.
 * Compressed BYTES are not pinned by any reference test (tests/write_archive.c checks return codes
 * only; liblz4 output is version dependent) — "parity" for this function is: the reference decoder
 * (oracle/_ref) reproduces the input, and the block layout rules of the LZ4 block format hold
 * (last 5 bytes literal, last match starts >= 12 bytes before the block end).
 */
#include "oracle.h"
#include <string.h>

#define BLOCK      65536u
#define HASH_LOG   13
#define MINMATCH   4
#define MFLIMIT    12
#define LASTLIT    5

/* LZ4F_compressBound(n, NULL) of lz4 1.9.x: worst-case flags (block+content checksum), 64 KiB
 * blocks, a full 64 KiB-1 of previously buffered input.  lib/zpack_write.c:141 */
size_t orc_lz4f_bound(size_t n)
{
    size_t max_src = n + (BLOCK - 1);
    size_t full = max_src / BLOCK;
    size_t part = max_src & (BLOCK - 1);
    size_t last = (n == 0) ? part : 0;
    size_t nblocks = full + (last > 0);
    return (4 + 4) * nblocks + (size_t)BLOCK * full + last + (4 + 4);
}

static void wr32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }
static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint32_t hash4(uint32_t v) { return (v * 2654435761u) >> (32 - HASH_LOG); }

static size_t put_len(uint8_t* dst, size_t n)        /* 255-run length extension */
{
    size_t k = 0;
    while (n >= 255) { dst[k++] = 255; n -= 255; }
    dst[k++] = (uint8_t)n;
    return k;
}

/* compress src[start, start+len) where src[0,start) is history (<= 64 KiB reachable).
 * Returns compressed size, or 0 if it does not fit in cap. */
static size_t lz4_block_encode(const uint8_t* src, size_t start, size_t len, uint8_t* dst, size_t cap, int32_t* table)
{
    size_t ip = start, anchor = start, end = start + len, op = 0;
    size_t mflimit = len >= MFLIMIT ? end - MFLIMIT : start;   /* last position a match may start */
    if (len >= MFLIMIT + 1) {
        while (ip <= mflimit) {
            uint32_t h = hash4(rd32(src + ip));
            int32_t cand = table[h];
            table[h] = (int32_t)ip;
            if (cand >= 0 && ip - (size_t)cand <= 65535 && rd32(src + cand) == rd32(src + ip)) {
                size_t m = (size_t)cand;
                size_t mlen = MINMATCH;
                size_t limit = end - LASTLIT;
                while (ip + mlen < limit && src[m + mlen] == src[ip + mlen]) mlen++;
                size_t lit = ip - anchor;
                size_t need = 1 + lit / 255 + 1 + lit + 2 + (mlen - MINMATCH) / 255 + 1;
                if (op + need > cap) return 0;
                uint8_t* tok = dst + op++;
                if (lit >= 15) { *tok = 0xF0; op += put_len(dst + op, lit - 15); } else *tok = (uint8_t)(lit << 4);
                memcpy(dst + op, src + anchor, lit); op += lit;
                size_t off = ip - m;
                dst[op++] = (uint8_t)off; dst[op++] = (uint8_t)(off >> 8);
                size_t ml = mlen - MINMATCH;
                if (ml >= 15) { *tok |= 15; op += put_len(dst + op, ml - 15); } else *tok |= (uint8_t)ml;
                ip += mlen;
                anchor = ip;
                continue;
            }
            ip++;
        }
    }
    /* last literals */
    size_t lit = end - anchor;
    size_t need = 1 + lit / 255 + 1 + lit;
    if (op + need > cap) return 0;
    uint8_t* tok = dst + op++;
    if (lit >= 15) { *tok = 0xF0; op += put_len(dst + op, lit - 15); } else *tok = (uint8_t)(lit << 4);
    memcpy(dst + op, src + anchor, lit); op += lit;
    return op;
}

size_t orc_lz4f_encode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap)
{
    static const uint8_t header[7] = { 0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0xC0 };
    size_t op = 0;
    int32_t table[1 << HASH_LOG];
    if (dst_cap < 7) return 0;
    memcpy(dst, header, 7); op = 7;
    for (size_t i = 0; i < (1u << HASH_LOG); i++) table[i] = -1;
    for (size_t pos = 0; pos < src_size; pos += BLOCK) {
        size_t len = src_size - pos < BLOCK ? src_size - pos : BLOCK;
        if (dst_cap - op < 4 + len) return 0;
        /* liblz4 LZ4F_makeBlock: compress into at most len-1 bytes, else store */
        size_t c = lz4_block_encode(src, pos, len, dst + op + 4, len - 1, table);
        if (c == 0) {
            wr32(dst + op, (uint32_t)len | 0x80000000u);
            memcpy(dst + op + 4, src + pos, len);
            op += 4 + len;
            /* positions of a stored block were still entered into the table by the attempt above;
             * they stay valid history because the block is linked */
        } else {
            wr32(dst + op, (uint32_t)c);
            op += 4 + c;
        }
    }
    if (dst_cap - op < 4) return 0;
    wr32(dst + op, 0); op += 4;
    return op;
}
