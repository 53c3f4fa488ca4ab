/*
 * entry_oracle.c — the per-entry hot path of ZPack restated end to end.  TEST INFRASTRUCTURE ONLY.
 *
 *   orc_entry_decode  = zpack_read_file            lib/zpack_read.c:326-471 (memory-backed reader)
 *   orc_entry_encode  = zpack_compress_file        lib/zpack_write.c:161-224
 *                     + the hash of zpack_add_written_file_entry   lib/zpack_write.c:256
 */
#include "oracle.h"
#include <string.h>

int orc_entry_decode(const uint8_t* archive, uint64_t archive_size,
                     uint64_t offset, uint64_t comp_size, uint64_t uncomp_size,
                     uint64_t expect_hash, int method,
                     uint8_t* dst, size_t max_size, uint64_t* produced, uint64_t* hash)
{
    if (produced) *produced = 0;
    if (hash) *hash = 0;
    /* guards, in the reference's order (zpack_read.c:328-332); note the `>=` of :331 */
    if (comp_size == 0) return ORC_OK;
    if (max_size < uncomp_size) return ORC_ERROR_BUFFER_TOO_SMALL;
    if (offset + comp_size >= archive_size) return ORC_ERROR_FILE_OFFSET_INVALID;
    const uint8_t* comp = archive + offset;
    size_t got = 0;

    switch (method) {
    case ORC_METHOD_NONE:
        if (uncomp_size > comp_size) return ORC_ERROR_FILE_SIZE_INVALID;      /* :354 */
        memcpy(dst, comp, (size_t)uncomp_size);                               /* :366 */
        got = (size_t)uncomp_size;
        break;
    case ORC_METHOD_ZSTD: {
        int r = orc_zstd_decode(comp, (size_t)comp_size, dst, max_size, &got); /* :380 */
        if (produced) *produced = got;
        if (r != 0) return ORC_ERROR_DECOMPRESS_FAILED;                        /* :384-388 */
        break;
    }
    case ORC_METHOD_LZ4: {
        /* :414-450.  The reference loops LZ4F_decompress while input and output space remain; with
         * the whole frame in memory one pass either finishes the frame (last_return 0), runs out of
         * input (hint != 0, output space left -> FILE_INCOMPLETE), runs out of output space
         * (-> BUFFER_TOO_SMALL) or hits malformed data (-> DECOMPRESS_FAILED). */
        int r = orc_lz4f_decode(comp, (size_t)comp_size, dst, max_size, &got);
        if (produced) *produced = got;
        if (r == -1) return ORC_ERROR_DECOMPRESS_FAILED;
        if (r == -2) return got < max_size ? ORC_ERROR_FILE_INCOMPLETE : ORC_ERROR_BUFFER_TOO_SMALL;
        if (r == -3) return ORC_ERROR_BUFFER_TOO_SMALL;
        break;
    }
    default:
        return ORC_ERROR_COMP_METHOD_INVALID;                                  /* :459-461 */
    }
    if (produced) *produced = got;
    /* :466-468 — the hash covers uncomp_size bytes of the caller's buffer, whatever the codec produced */
    uint64_t h = orc_xxh3_64(dst, (size_t)uncomp_size);
    if (hash) *hash = h;
    return h == expect_hash ? ORC_OK : ORC_ERROR_FILE_HASH_MISMATCH;
}

int orc_entry_encode(const uint8_t* src, uint64_t size, int method, int level,
                     uint8_t* dst, size_t dst_cap, uint64_t* comp_size, uint64_t* hash)
{
    (void)level;
    size_t c;
    switch (method) {
    case ORC_METHOD_NONE:
        if (dst_cap < size) return ORC_ERROR_BUFFER_TOO_SMALL;                 /* zpack_write.c:167 */
        memcpy(dst, src, (size_t)size);
        c = (size_t)size;
        break;
    case ORC_METHOD_ZSTD:
        c = orc_zstd_encode(src, (size_t)size, dst, dst_cap);
        if (c == 0) return ORC_ERROR_COMPRESS_FAILED;                          /* :183-184 */
        break;
    case ORC_METHOD_LZ4:
        c = orc_lz4f_encode(src, (size_t)size, dst, dst_cap);
        if (c == 0) return ORC_ERROR_COMPRESS_FAILED;                          /* :152-159 */
        break;
    default:
        return ORC_ERROR_COMP_METHOD_INVALID;                                  /* :219-220 */
    }
    *comp_size = c;
    *hash = orc_xxh3_64(src, (size_t)size);                                    /* :256 */
    return ORC_OK;
}
