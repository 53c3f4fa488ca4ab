/*
 * xxh3_oracle.c — XXH3-64 (seed 0, default secret), XXH32 and XXH64, restated from the xxHash
 * specification.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Reference call sites this stands in for:
 *   XXH3_64bits            lib/zpack_read.c:466, lib/zpack_write.c:256
 *   XXH3_64bits_reset/update/digest   lib/zpack_stream.c:12; lib/zpack_read.c:525,556,579,609,634;
 *                                     lib/zpack_write.c:471,503,535,609,627,680,683
 * XXH32 / XXH64 are what LZ4F (header, block, content checksums) and Zstandard (content checksum)
 * use inside the frames; the reference never sets those flags when writing but its decoders
 * verify them when present.
 */
#include "oracle.h"
#include <string.h>

#define P32_1 0x9E3779B1U
#define P32_2 0x85EBCA77U
#define P32_3 0xC2B2AE3DU
#define P32_4 0x27D4EB2FU
#define P32_5 0x165667B1U
#define P64_1 0x9E3779B185EBCA87ULL
#define P64_2 0xC2B2AE3D27D4EB4FULL
#define P64_3 0x165667B19E3779F9ULL
#define P64_4 0x85EBCA77C2B2AE63ULL
#define P64_5 0x27D4EB2F165667C5ULL
#define PMX_1 0x165667919E3779F9ULL
#define PMX_2 0x9FB21C651E98DF25ULL

#define SECRET_SIZE 192
static const uint8_t k_secret[SECRET_SIZE] = {
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

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
static uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static uint32_t bswap32(uint32_t x) { return ((x & 0xFF) << 24) | ((x & 0xFF00) << 8) | ((x >> 8) & 0xFF00) | (x >> 24); }
static uint64_t bswap64(uint64_t x) { return ((uint64_t)bswap32((uint32_t)x) << 32) | bswap32((uint32_t)(x >> 32)); }

static uint64_t mul128_fold64(uint64_t a, uint64_t b)
{
    unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}

/* ------------------------------------------------------------------ XXH32 */
static uint32_t xxh32_round(uint32_t acc, uint32_t in) { acc += in * P32_2; acc = rotl32(acc, 13); return acc * P32_1; }

uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint32_t h;
    if (len >= 16) {
        uint32_t v1 = seed + P32_1 + P32_2, v2 = seed + P32_2, v3 = seed, v4 = seed - P32_1;
        do {
            v1 = xxh32_round(v1, rd32(p));
            v2 = xxh32_round(v2, rd32(p + 4));
            v3 = xxh32_round(v3, rd32(p + 8));
            v4 = xxh32_round(v4, rd32(p + 12));
            p += 16;
        } while (p + 16 <= end);
        h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
    } else {
        h = seed + P32_5;
    }
    h += (uint32_t)len;
    while (p + 4 <= end) { h += rd32(p) * P32_3; h = rotl32(h, 17) * P32_4; p += 4; }
    while (p < end) { h += (*p) * P32_5; h = rotl32(h, 11) * P32_1; p++; }
    h ^= h >> 15; h *= P32_2; h ^= h >> 13; h *= P32_3; h ^= h >> 16;
    return h;
}

/* ------------------------------------------------------------------ XXH64 */
static uint64_t xxh64_round(uint64_t acc, uint64_t in) { acc += in * P64_2; acc = rotl64(acc, 31); return acc * P64_1; }
static uint64_t xxh64_merge(uint64_t h, uint64_t v) { v = xxh64_round(0, v); h ^= v; return h * P64_1 + P64_4; }
static uint64_t xxh64_avalanche(uint64_t h) { h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32; return h; }

uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed)
{
    const uint8_t* p = (const uint8_t*)data;
    const uint8_t* end = p + len;
    uint64_t h;
    if (len >= 32) {
        uint64_t v1 = seed + P64_1 + P64_2, v2 = seed + P64_2, v3 = seed, v4 = seed - P64_1;
        do {
            v1 = xxh64_round(v1, rd64(p));
            v2 = xxh64_round(v2, rd64(p + 8));
            v3 = xxh64_round(v3, rd64(p + 16));
            v4 = xxh64_round(v4, rd64(p + 24));
            p += 32;
        } while (p + 32 <= end);
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh64_merge(h, v1); h = xxh64_merge(h, v2); h = xxh64_merge(h, v3); h = xxh64_merge(h, v4);
    } else {
        h = seed + P64_5;
    }
    h += (uint64_t)len;
    while (p + 8 <= end) { h ^= xxh64_round(0, rd64(p)); h = rotl64(h, 27) * P64_1 + P64_4; p += 8; }
    if (p + 4 <= end) { h ^= (uint64_t)rd32(p) * P64_1; h = rotl64(h, 23) * P64_2 + P64_3; p += 4; }
    while (p < end) { h ^= (*p) * P64_5; h = rotl64(h, 11) * P64_1; p++; }
    return xxh64_avalanche(h);
}

/* ------------------------------------------------------------------ XXH3-64 */
static uint64_t xxh3_avalanche(uint64_t h) { h ^= h >> 37; h *= PMX_1; h ^= h >> 32; return h; }

static uint64_t rrmxmx(uint64_t h, uint64_t len)
{
    h ^= rotl64(h, 49) ^ rotl64(h, 24);
    h *= PMX_2;
    h ^= (h >> 35) + len;
    h *= PMX_2;
    return h ^ (h >> 28);
}

static uint64_t mix16(const uint8_t* in, const uint8_t* sec)
{
    return mul128_fold64(rd64(in) ^ rd64(sec), rd64(in + 8) ^ rd64(sec + 8));
}

static void accumulate_stripe(uint64_t acc[8], const uint8_t* in, const uint8_t* sec)
{
    for (int i = 0; i < 8; i++) {
        uint64_t v = rd64(in + 8 * i);
        uint64_t k = v ^ rd64(sec + 8 * i);
        acc[i ^ 1] += v;
        acc[i] += (k & 0xFFFFFFFFULL) * (k >> 32);
    }
}

static void scramble(uint64_t acc[8])
{
    const uint8_t* sec = k_secret + SECRET_SIZE - 64;
    for (int i = 0; i < 8; i++) {
        uint64_t a = acc[i];
        a ^= a >> 47;
        a ^= rd64(sec + 8 * i);
        a *= P32_1;
        acc[i] = a;
    }
}

static void acc_init(uint64_t acc[8])
{
    acc[0] = P32_3; acc[1] = P64_1; acc[2] = P64_2; acc[3] = P64_3;
    acc[4] = P64_4; acc[5] = P32_2; acc[6] = P64_5; acc[7] = P32_1;
}

static uint64_t merge_accs(const uint64_t acc[8], uint64_t len)
{
    const uint8_t* sec = k_secret + 11;
    uint64_t r = len * P64_1;
    for (int i = 0; i < 4; i++)
        r += mul128_fold64(acc[2 * i] ^ rd64(sec + 16 * i), acc[2 * i + 1] ^ rd64(sec + 16 * i + 8));
    return xxh3_avalanche(r);
}

static uint64_t xxh3_short(const uint8_t* in, size_t len)
{
    if (len == 0)
        return xxh64_avalanche(rd64(k_secret + 56) ^ rd64(k_secret + 64));
    if (len <= 3) {
        uint32_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
        uint32_t combined = (c1 << 16) | (c2 << 24) | c3 | ((uint32_t)len << 8);
        uint64_t flip = (uint64_t)(rd32(k_secret) ^ rd32(k_secret + 4));
        return xxh64_avalanche((uint64_t)combined ^ flip);
    }
    if (len <= 8) {
        uint32_t a = rd32(in), b = rd32(in + len - 4);
        uint64_t flip = rd64(k_secret + 8) ^ rd64(k_secret + 16);
        uint64_t in64 = (uint64_t)b + ((uint64_t)a << 32);
        return rrmxmx(in64 ^ flip, len);
    }
    if (len <= 16) {
        uint64_t f1 = rd64(k_secret + 24) ^ rd64(k_secret + 32);
        uint64_t f2 = rd64(k_secret + 40) ^ rd64(k_secret + 48);
        uint64_t lo = rd64(in) ^ f1, hi = rd64(in + len - 8) ^ f2;
        uint64_t acc = len + bswap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    if (len <= 128) {
        uint64_t acc = len * P64_1;
        if (len > 32) {
            if (len > 64) {
                if (len > 96) {
                    acc += mix16(in + 48, k_secret + 96);
                    acc += mix16(in + len - 64, k_secret + 112);
                }
                acc += mix16(in + 32, k_secret + 64);
                acc += mix16(in + len - 48, k_secret + 80);
            }
            acc += mix16(in + 16, k_secret + 32);
            acc += mix16(in + len - 32, k_secret + 48);
        }
        acc += mix16(in, k_secret);
        acc += mix16(in + len - 16, k_secret + 16);
        return xxh3_avalanche(acc);
    }
    /* 129..240 */
    {
        uint64_t acc = len * P64_1;
        size_t rounds = len / 16;
        for (size_t i = 0; i < 8; i++) acc += mix16(in + 16 * i, k_secret + 16 * i);
        acc = xxh3_avalanche(acc);
        for (size_t i = 8; i < rounds; i++) acc += mix16(in + 16 * i, k_secret + 16 * (i - 8) + 3);
        acc += mix16(in + len - 16, k_secret + 136 - 17);
        return xxh3_avalanche(acc);
    }
}

uint64_t orc_xxh3_64(const void* data, size_t len)
{
    const uint8_t* in = (const uint8_t*)data;
    if (len <= 240) return xxh3_short(in, len);

    uint64_t acc[8];
    acc_init(acc);
    const size_t block_len = 1024;
    size_t nb_blocks = (len - 1) / block_len;
    for (size_t n = 0; n < nb_blocks; n++) {
        for (int s = 0; s < 16; s++) accumulate_stripe(acc, in + n * block_len + 64 * (size_t)s, k_secret + 8 * s);
        scramble(acc);
    }
    size_t nb_stripes = ((len - 1) - block_len * nb_blocks) / 64;
    for (size_t s = 0; s < nb_stripes; s++) accumulate_stripe(acc, in + nb_blocks * block_len + 64 * s, k_secret + 8 * s);
    accumulate_stripe(acc, in + len - 64, k_secret + SECRET_SIZE - 64 - 7);
    return merge_accs(acc, (uint64_t)len);
}

/* ------------------------------------------------------------------ streaming XXH3-64 */
void orc_xxh3_reset(orc_xxh3_state* st)
{
    memset(st, 0, sizeof(*st));
    acc_init(st->acc);
}

static void stream_consume(uint64_t acc[8], uint32_t* stripes_in_block, const uint8_t* p, size_t nstripes)
{
    for (size_t i = 0; i < nstripes; i++) {
        accumulate_stripe(acc, p + 64 * i, k_secret + 8 * (*stripes_in_block));
        if (++(*stripes_in_block) == 16) { scramble(acc); *stripes_in_block = 0; }
    }
}

void orc_xxh3_update(orc_xxh3_state* st, const void* data, size_t len)
{
    const uint8_t* p = (const uint8_t*)data;
    st->total += len;
    while (len) {
        if (st->buffered == sizeof(st->buf)) {
            /* more input follows, so the first 3 buffered stripes each have a successor byte */
            stream_consume(st->acc, &st->stripes_in_block, st->buf, 3);
            memmove(st->buf, st->buf + 192, 64);
            st->buffered = 64;
        }
        size_t room = sizeof(st->buf) - st->buffered;
        size_t n = len < room ? len : room;
        memcpy(st->buf + st->buffered, p, n);
        st->buffered += (uint32_t)n;
        p += n; len -= n;
    }
}

uint64_t orc_xxh3_digest(const orc_xxh3_state* st)
{
    if (st->total <= 240) return xxh3_short(st->buf, (size_t)st->total);
    uint64_t acc[8];
    memcpy(acc, st->acc, sizeof(acc));
    uint32_t sib = st->stripes_in_block;
    size_t n = (st->buffered - 1) / 64;
    stream_consume(acc, &sib, st->buf, n);
    /* a scramble done by the line above is only legal if a byte follows the block — true, since
     * stripe i < (buffered-1)/64 always has a successor byte. */
    accumulate_stripe(acc, st->buf + st->buffered - 64, k_secret + SECRET_SIZE - 64 - 7);
    return merge_accs(acc, st->total);
}
