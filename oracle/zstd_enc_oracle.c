/*
 * zstd_enc_oracle.c — minimal VALID Zstandard frame writer.  TEST INFRASTRUCTURE ONLY.
 *
 * Stands in for ZSTD_compressCCtx (lib/zpack_write.c:179) only as far as the container goes:
 * frame header flavour of the one-shot API (single-segment + Frame_Content_Size, no checksum, no
 * dictID — `28 B5 2F FD 20 <n>` / `.. A0 <u32>`), then Raw_Block / RLE_Block bodies.  Compressed
 * bytes are NOT pinned by any reference test (tests/write_archive.c checks return codes only) and
 * libzstd's output is version- and level-dependent, so the oracle's role for the write path is the
 * frame grammar + "the reference decoder reproduces the input"; ratio is reported against the real
 * libzstd by bench.py, not against this file.
 */
#include "oracle.h"
#include <string.h>

#define BLOCK_MAX (128u << 10)

/* ZSTD_COMPRESSBOUND(n)  (zstd.h; lib/zpack_write.c:134) */
size_t orc_zstd_bound(size_t n)
{
    return n + (n >> 8) + (n < BLOCK_MAX ? ((BLOCK_MAX - n) >> 11) : 0);
}

size_t orc_zstd_encode(const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    size_t op = 0;
    if (cap < 18) return 0;
    dst[op++] = 0x28; dst[op++] = 0xB5; dst[op++] = 0x2F; dst[op++] = 0xFD;
    int single = n <= (2u << 20);
    unsigned fcs_flag, fcs_bytes;
    if (single && n < 256) { fcs_flag = 0; fcs_bytes = 1; }
    else if (n < 65536 + 256) { fcs_flag = 1; fcs_bytes = 2; }
    else if (n <= 0xFFFFFFFFull) { fcs_flag = 2; fcs_bytes = 4; }
    else { fcs_flag = 3; fcs_bytes = 8; }
    dst[op++] = (uint8_t)((fcs_flag << 6) | (single ? 0x20 : 0));
    if (!single) dst[op++] = 0x58;                      /* windowLog 21 = 2 MiB */
    uint64_t f = fcs_bytes == 2 ? (uint64_t)n - 256 : (uint64_t)n;
    for (unsigned i = 0; i < fcs_bytes; i++) dst[op++] = (uint8_t)(f >> (8 * i));

    size_t pos = 0;
    do {
        size_t len = n - pos < BLOCK_MAX ? n - pos : BLOCK_MAX;
        int last = pos + len == n;
        int rle = len > 1;
        for (size_t i = 1; rle && i < len; i++) rle = src[pos + i] == src[pos];
        if (rle) {
            if (cap - op < 4) return 0;
            uint32_t bh = (uint32_t)last | (1u << 1) | ((uint32_t)len << 3);
            dst[op++] = (uint8_t)bh; dst[op++] = (uint8_t)(bh >> 8); dst[op++] = (uint8_t)(bh >> 16);
            dst[op++] = src[pos];
        } else {
            if (cap - op < 3 + len) return 0;
            uint32_t bh = (uint32_t)last | ((uint32_t)len << 3);
            dst[op++] = (uint8_t)bh; dst[op++] = (uint8_t)(bh >> 8); dst[op++] = (uint8_t)(bh >> 16);
            memcpy(dst + op, src + pos, len); op += len;
        }
        pos += len;
    } while (pos < n);
    return op;
}
