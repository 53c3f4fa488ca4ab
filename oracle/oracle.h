/*
 * oracle.h — CPU restatement of the ZPack entry-codec hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the checker for the HIP product in zpack_amd/.  It is imported ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the product path links,
 * loads or calls it; the product fails loudly when its HIP library is missing.
 *
 * What it restates (reference = /root/reference, ZPack v2.0.3):
 *   - lib/zpack_read.c:326-471  zpack_read_file      -> orc_entry_decode()
 *   - lib/zpack_write.c:161-260 zpack_compress_file + zpack_add_written_file_entry -> orc_entry_encode()
 * The codec arithmetic the reference reaches through its un-vendored submodules
 * (externals/lz4, externals/zstd, externals/xxHash — empty directories, pinned commits
 * unrecoverable; versions present in this image: lz4 1.9.3, zstd 1.4.9, xxHash 0.8.x) is restated
 * from the published formats: LZ4 Frame format v1.6.x + LZ4 Block format, RFC 8878 (Zstandard),
 * and the xxHash specification (XXH32, XXH64, XXH3-64).
 *
 * PARITY PINNING: pinned.  tests/test_oracle_golden.py checks every function here against
 *   (a) the reference's own golden archives + plaintexts + hashes (tests/golden/ref_workdir/,
 *       = /root/reference/tests/workdir, values of tests/archive.h:93-115), and
 *   (b) fixtures produced by running the compiled reference (oracle/_ref/libzpack_ref.so, built
 *       by `make ref` from the sources in place) in the build container
 *       (tests/golden/make_golden.py -> tests/golden/ fixtures).
 */
#ifndef ZPK_ORACLE_H
#define ZPK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- result codes: numerically identical to enum zpack_result (lib/zpack.h:189-218) ---- */
enum {
    ORC_OK = 0,
    ORC_ERROR_BUFFER_TOO_SMALL = 12,
    ORC_ERROR_DECOMPRESS_FAILED = 13,
    ORC_ERROR_COMPRESS_FAILED = 14,
    ORC_ERROR_FILE_HASH_MISMATCH = 15,
    ORC_ERROR_FILE_OFFSET_INVALID = 16,
    ORC_ERROR_FILE_INCOMPLETE = 17,
    ORC_ERROR_FILE_SIZE_INVALID = 18,
    ORC_ERROR_COMP_METHOD_INVALID = 19
};

/* compression methods (lib/zpack.h:60-65) */
enum { ORC_METHOD_NONE = 0, ORC_METHOD_ZSTD = 1, ORC_METHOD_LZ4 = 2 };

/* ---- xxHash family (xxHash spec; call sites lib/zpack_read.c:466, lib/zpack_write.c:256) ---- */
uint64_t orc_xxh3_64(const void* data, size_t len);                 /* XXH3_64bits, seed 0, default secret */
uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed);    /* LZ4F header/block/content checksums */
uint64_t orc_xxh64(const void* data, size_t len, uint64_t seed);    /* zstd content checksum (low 32 bits) */

/* streaming XXH3-64 (lib/zpack_stream.c:4-28; zpack_read.c:525,556,579,609,634) */
typedef struct orc_xxh3_state_s {
    uint64_t acc[8];
    uint8_t  buf[256];
    uint32_t buffered;
    uint64_t total;
    uint32_t stripes_in_block;
} orc_xxh3_state;
void     orc_xxh3_reset(orc_xxh3_state* st);
void     orc_xxh3_update(orc_xxh3_state* st, const void* data, size_t len);
uint64_t orc_xxh3_digest(const orc_xxh3_state* st);

/* ---- LZ4 frame decode (replaces the LZ4F_decompress loop, lib/zpack_read.c:414-439) ----
 * Returns 0 when a complete frame was decoded, else a negative code:
 *   -1 malformed (-> DECOMPRESS_FAILED), -2 input truncated (-> FILE_INCOMPLETE),
 *   -3 output capacity exhausted (-> BUFFER_TOO_SMALL).
 * *produced = bytes written to dst (also on error: bytes produced before the failure). */
int orc_lz4f_decode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap, size_t* produced);

/* bare LZ4 block decode with `hist` bytes of already-decoded history directly before dst */
int orc_lz4_block_decode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap,
                         size_t hist, size_t* produced);

/* ---- Zstandard frame decode (replaces ZSTD_decompressDCtx, lib/zpack_read.c:380) ----
 * Decodes all concatenated frames (and skips skippable frames) like ZSTD_decompressDCtx.
 * Returns 0 on success, -1 malformed/truncated, -3 output capacity exhausted. */
int orc_zstd_decode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap, size_t* produced);

/* statistics of the last frame decoded by orc_zstd_decode (test introspection: which format
 * features a fixture exercises).  Not thread safe. */
typedef struct orc_zstd_stats_s {
    uint32_t frames, blocks, raw_blocks, rle_blocks, comp_blocks;
    uint32_t lit_raw, lit_rle, lit_huf, lit_treeless, lit_huf_1stream, lit_huf_4stream;
    uint32_t huf_fse_weights, huf_direct_weights;
    uint32_t seq_mode[3][4];    /* [LL,OF,ML][predefined,rle,fse,repeat] */
    uint64_t sequences;
    uint32_t repcode_uses;
    uint64_t window_size;
    uint32_t single_segment, has_fcs, has_checksum;
} orc_zstd_stats;
const orc_zstd_stats* orc_zstd_last_stats(void);
/* trace of the sequences orc_zstd_decode executes, packed offset | match length << 29 | literal length << 47 (the
 * layout of zpack_amd/csrc/zstd_fse4.h), into buf[0..cap); NULL switches it off.  Count = sequences seen. */
void orc_zstd_trace(uint64_t* buf, size_t cap);
size_t orc_zstd_trace_count(void);

/* ---- encoders (replace LZ4F_compressBegin/Update/End, zpack_write.c:204-210, and
 *      ZSTD_compressCCtx, zpack_write.c:179).  Compressed bytes are NOT pinned by any reference
 *      test (tests/write_archive.c checks return codes only); validity is: the reference decoder
 *      reproduces the input.  Return compressed size, or 0 when dst_cap is too small. ---- */
size_t orc_lz4f_bound(size_t src_size);   /* = LZ4F_compressBound(n, NULL) + header (zpack_write.c:141) */
size_t orc_zstd_bound(size_t src_size);   /* = ZSTD_COMPRESSBOUND(n)              (zpack_write.c:134) */
size_t orc_lz4f_encode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap);
size_t orc_zstd_encode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap);

/* ---- the per-entry hot path, restated end to end ---- */

/* zpack_read_file (lib/zpack_read.c:326-471) for a memory-backed reader:
 *   archive/archive_size = reader->buffer / reader->file_size.
 * Returns a zpack_result code.  *produced and *hash are diagnostics (may be NULL). */
int orc_entry_decode(const uint8_t* archive, uint64_t archive_size,
                     uint64_t offset, uint64_t comp_size, uint64_t uncomp_size,
                     uint64_t expect_hash, int method,
                     uint8_t* dst, size_t max_size, uint64_t* produced, uint64_t* hash);

/* zpack_compress_file + hash (lib/zpack_write.c:161-260).  Returns a zpack_result code. */
int orc_entry_encode(const uint8_t* src, uint64_t size, int method, int level,
                     uint8_t* dst, size_t dst_cap, uint64_t* comp_size, uint64_t* hash);

#ifdef __cplusplus
}
#endif
#endif
