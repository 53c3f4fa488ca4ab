/*
 * zpack.h — public C API of the ZPack archive library, MI355X edition.
 *
 * Source- and ABI-compatible with the reference header (/root/reference/lib/zpack.h, v2.0.3): same
 * function names and signatures (lib/zpack.h:237-742), same struct layouts (zpack_file_entry 48 B,
 * zpack_reader 112 B, zpack_file 40 B, zpack_writer 104 B, zpack_stream 64 B on x86-64), same result
 * codes (lib/zpack.h:189-218), same on-disk format (docs/specs.md).  A program written against the
 * reference recompiles or relinks against libzpack_amd.so unchanged.
 *
 * What differs is underneath: every entry's decompress / compress + XXH3-64 goes through the codec
 * C-ABI (zpack_codec.h) into HIP kernels on the GPU.  Consequences a caller can observe:
 *   - the `void* dctx / cctx` handles are OPAQUE codec contexts obtained from zpack_create_dctx /
 *     zpack_create_cctx (the reference let a caller pass a raw ZSTD_DCtx* / LZ4F_dctx* here — that was
 *     never documented and is not supported);
 *   - contexts: a reader / writer called with dctx / cctx == NULL creates its own context on first use (as the
 *     reference does, lib/zpack_read.c:17-31) — nothing is shared between two readers, so two readers on two threads
 *     are independent; buffer-backed reads from N threads on ONE reader are safe when each thread passes its own
 *     zpack_create_dctx context (lib/zpack.h:337-340); a context shared by threads is serialised, not corrupted.
 *     With ZPACK_AMD_DEVICES=all (or a list of HIP ordinals) a context spans several GPUs and the batch calls
 *     (zpack_read_files*, zpack_write_files) shard their entries statically over them;
 *   - if no HIP device is usable, calls that need the codec return ZPACK_ERROR_NOT_AVAILABLE — there is
 *     no CPU fallback;
 *   - two ADDITIVE batch entry points, zpack_read_files / zpack_read_files_packed, mirror
 *     zpack_write_files on the read side (the reference reads one entry per call, lib/zpack.h:383).
 */
#ifndef __ZPACK_H__
#define __ZPACK_H__

#include <stdio.h>
#include <stddef.h>
#include <stdint.h>

#if defined(_WIN32) && defined(zpack_EXPORTS)
#   define ZPACK_EXPORT __declspec(dllexport)
#elif defined(__GNUC__)
#   define ZPACK_EXPORT __attribute__((visibility("default")))
#else
#   define ZPACK_EXPORT
#endif

/* library version this API is compatible with */
#define ZPACK_VERSION_MAJOR 2
#define ZPACK_VERSION_MINOR 0
#define ZPACK_VERSION_PATCH 3
#define ZPACK_VERSION (ZPACK_VERSION_MAJOR * 100000 + ZPACK_VERSION_MINOR * 1000 + ZPACK_VERSION_PATCH * 10)
#define ZPACK_VERSION_STRING "2.0.3"

typedef uint8_t  zpack_u8;
typedef uint16_t zpack_u16;
typedef uint32_t zpack_u32;
typedef uint64_t zpack_u64;
typedef zpack_u8 zpack_bool;
#define ZPACK_FALSE 0
#define ZPACK_TRUE  1

/* ---- on-disk format constants (docs/specs.md; lib/zpack.h:36-52) ---- */
#define ZPACK_HEADER_SIGNATURE 0x154b505a   /* "ZPK\x15" */
#define ZPACK_DATA_SIGNATURE   0x144b505a   /* "ZPK\x14" */
#define ZPACK_CDR_SIGNATURE    0x134b505a   /* "ZPK\x13" */
#define ZPACK_EOCDR_SIGNATURE  0x124b505a   /* "ZPK\x12" */
#define ZPACK_SIGNATURE_SIZE 4
#define ZPACK_HEADER_SIZE 6
#define ZPACK_CDR_HEADER_SIZE 20
#define ZPACK_FILE_ENTRY_FIXED_SIZE 35
#define ZPACK_EOCDR_SIZE 12
#define ZPACK_MINIMUM_ARCHIVE_SIZE (ZPACK_HEADER_SIZE + ZPACK_SIGNATURE_SIZE + ZPACK_CDR_HEADER_SIZE + ZPACK_EOCDR_SIZE)
#define ZPACK_MAX_FILENAME_LENGTH 65535
#define ZPACK_ARCHIVE_VERSION_MIN 1
#define ZPACK_ARCHIVE_VERSION_MAX 1

typedef enum zpack_compression_method_e {
    ZPACK_COMPRESSION_NONE = 0,
    ZPACK_COMPRESSION_ZSTD = 1,
    ZPACK_COMPRESSION_LZ4  = 2
} zpack_compression_method;

/* one archive entry (lib/zpack.h:71-80) */
typedef struct zpack_file_entry_s {
    char*     filename;
    zpack_u64 offset;        /* of the compressed payload inside the archive */
    zpack_u64 comp_size;
    zpack_u64 uncomp_size;
    zpack_u64 hash;          /* XXH3-64 of the uncompressed bytes */
    zpack_u8  comp_method;
} zpack_file_entry;

/* archive reader (lib/zpack.h:85-110); zero-initialise before use */
typedef struct zpack_reader_s {
    zpack_u16 version;
    zpack_file_entry* file_entries;
    zpack_u64 file_count;
    zpack_u64 comp_size;
    zpack_u64 uncomp_size;
    size_t file_size;
    void* zstd_dctx;         /* this reader's own codec context, created on the first decode with dctx == NULL (every method) */
    void* lz4f_dctx;         /* PRIVATE: name arena of a reader-owned entry table.  While set, every file_entries[i].filename
                                points into it: the names are read-only and must not be freed or replaced by the caller;
                                zpack_close_reader / a second zpack_read_archive* free it.  Do not touch. */
    size_t last_return;      /* codec detail of the last decode (reference: library return value) */
    zpack_u64 cdr_offset;
    zpack_u64 eocdr_offset;
    zpack_u8* buffer;
    zpack_bool buffer_shared;
    FILE* file;
} zpack_reader;

typedef struct zpack_compress_options_s {
    zpack_compression_method method;
    int level;
} zpack_compress_options;

/* one file handed to the writer (lib/zpack.h:125-134) */
typedef struct zpack_file_s {
    char*     filename;
    zpack_u8* buffer;
    zpack_u64 size;
    zpack_compress_options* options;
    void* cctx;              /* optional codec context from zpack_create_cctx, else NULL */
} zpack_file;

/* archive writer (lib/zpack.h:139-164); zero-initialise before use */
typedef struct zpack_writer_s {
    zpack_u8* buffer;
    size_t buffer_capacity;
    FILE* file;
    size_t file_size;
    size_t write_offset;
    zpack_file_entry* file_entries;
    zpack_u64 fe_capacity;
    zpack_u64 file_count;
    void* zstd_cctx;         /* this writer's own codec context, created on the first write with cctx == NULL (every method) */
    void* lz4f_cctx;         /* streaming aggregation state */
    size_t last_return;
    zpack_u64 cdr_offset;
    zpack_u64 eocdr_offset;
} zpack_writer;

/* streaming cursor (lib/zpack.h:169-184) */
typedef struct zpack_stream_s {
    zpack_u8* next_in;
    size_t avail_in;
    size_t total_in;
    zpack_u8* next_out;
    size_t avail_out;
    size_t total_out;
    size_t read_back;        /* bytes at the END of the current input buffer that must be presented again first */
    void* xxh3_state;        /* here: the stream's aggregation state (the reference keeps an XXH3 state) */
} zpack_stream;

/* return codes (lib/zpack.h:189-218) */
enum zpack_result {
    ZPACK_OK,
    ZPACK_ERROR_ARCHIVE_NOT_LOADED,
    ZPACK_ERROR_WRITER_NOT_OPENED,
    ZPACK_ERROR_OPEN_FAILED,
    ZPACK_ERROR_SEEK_FAILED,
    ZPACK_ERROR_FILE_TOO_SMALL,
    ZPACK_ERROR_SIGNATURE_INVALID,
    ZPACK_ERROR_READ_FAILED,
    ZPACK_ERROR_BLOCK_SIZE_INVALID,
    ZPACK_ERROR_VERSION_INCOMPATIBLE,
    ZPACK_ERROR_MALLOC_FAILED,
    ZPACK_ERROR_FILE_NOT_FOUND,
    ZPACK_ERROR_BUFFER_TOO_SMALL,
    ZPACK_ERROR_DECOMPRESS_FAILED,
    ZPACK_ERROR_COMPRESS_FAILED,
    ZPACK_ERROR_FILE_HASH_MISMATCH,
    ZPACK_ERROR_FILE_OFFSET_INVALID,
    ZPACK_ERROR_FILE_INCOMPLETE,
    ZPACK_ERROR_FILE_SIZE_INVALID,
    ZPACK_ERROR_COMP_METHOD_INVALID,
    ZPACK_ERROR_WRITE_FAILED,
    ZPACK_ERROR_STREAM_INVALID,
    ZPACK_ERROR_HASH_FAILED,
    ZPACK_ERROR_FILENAME_TOO_LONG,
    ZPACK_ERROR_NOT_AVAILABLE
};

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- container parsing (CPU) */
ZPACK_EXPORT int zpack_read_header_memory(const zpack_u8* buffer, zpack_u16* version);
ZPACK_EXPORT int zpack_read_header(FILE* fp, zpack_u16* version);
ZPACK_EXPORT int zpack_read_data_header_memory(const zpack_u8* buffer);
ZPACK_EXPORT int zpack_read_data_header(FILE* fp);
ZPACK_EXPORT int zpack_read_eocdr_memory(const zpack_u8* buffer, zpack_u64* cdr_offset);
ZPACK_EXPORT int zpack_read_eocdr(FILE* fp, zpack_u64 eocdr_offset, zpack_u64* cdr_offset);
ZPACK_EXPORT int zpack_read_cdr_header_memory(const zpack_u8* buffer, zpack_u64* count, zpack_u64* block_size);
ZPACK_EXPORT int zpack_read_file_entry_memory(const zpack_u8* buffer, zpack_u64* size_left, zpack_file_entry* entry, size_t* entry_size);
ZPACK_EXPORT int zpack_read_file_entries_memory(const zpack_u8* buffer, zpack_file_entry** entries, zpack_u64 header_count,
                                                zpack_u64 block_size, zpack_u64* count, zpack_u64* total_cs, zpack_u64* total_us);
ZPACK_EXPORT int zpack_read_cdr_memory(const zpack_u8* buffer, size_t size_left, zpack_file_entry** entries, zpack_u64* count,
                                       zpack_u64* total_cs, zpack_u64* total_us);
ZPACK_EXPORT int zpack_read_cdr(FILE* fp, zpack_u64 cdr_offset, zpack_file_entry** entries, zpack_u64* count,
                                zpack_u64* total_cs, zpack_u64* total_us);
ZPACK_EXPORT int zpack_read_archive_memory(zpack_reader* reader);
ZPACK_EXPORT int zpack_read_archive(zpack_reader* reader);

/* ---------------------------------------------------------------- entry reads (GPU codec) */
ZPACK_EXPORT int zpack_read_raw_file(zpack_reader* reader, zpack_file_entry* entry, zpack_u8* buffer, size_t max_size);
/* decompress one entry into buffer and verify its hash; dctx from zpack_create_dctx or NULL */
ZPACK_EXPORT int zpack_read_file(zpack_reader* reader, zpack_file_entry* entry, zpack_u8* buffer, size_t max_size, void* dctx);
ZPACK_EXPORT int zpack_read_raw_file_stream(zpack_reader* reader, zpack_file_entry* entry, zpack_stream* stream, size_t* in_size);
ZPACK_EXPORT int zpack_read_file_stream(zpack_reader* reader, zpack_file_entry* entry, zpack_stream* stream, void* dctx);

/* ADDITIVE (not in the reference): decompress + verify `count` entries in one device batch.
 * buffers[i] / max_sizes[i] play zpack_read_file's (buffer, max_size) for entries[i]; results[i] receives
 * the zpack_result zpack_read_file would have returned for that entry (one bad entry does not stop the
 * others).  Returns ZPACK_OK when the batch ran (inspect results[]), or the first infrastructure error. */
ZPACK_EXPORT int zpack_read_files(zpack_reader* reader, zpack_file_entry* const* entries, zpack_u64 count,
                                  zpack_u8* const* buffers, const size_t* max_sizes, int* results, void* dctx);
/* Same, into one packed buffer: entry i lands at out_offsets[i] (filled in; sizes = uncomp_size, back to back) */
ZPACK_EXPORT int zpack_read_files_packed(zpack_reader* reader, zpack_file_entry* const* entries, zpack_u64 count,
                                         zpack_u8* buffer, size_t buffer_size, zpack_u64* out_offsets, int* results, void* dctx);

ZPACK_EXPORT int zpack_init_reader(zpack_reader* reader, const char* path);
ZPACK_EXPORT int zpack_init_reader_cfile(zpack_reader* reader, FILE* fp);
ZPACK_EXPORT int zpack_init_reader_memory(zpack_reader* reader, const zpack_u8* buffer, size_t size);
ZPACK_EXPORT int zpack_init_reader_memory_shared(zpack_reader* reader, zpack_u8* buffer, size_t size);
ZPACK_EXPORT void zpack_reset_reader_dctx(zpack_reader* reader);
ZPACK_EXPORT void zpack_close_reader(zpack_reader* reader);

/* ---------------------------------------------------------------- writing */
ZPACK_EXPORT int zpack_init_writer(zpack_writer* writer, const char* path);
ZPACK_EXPORT int zpack_init_writer_cfile(zpack_writer* writer, FILE* fp);
ZPACK_EXPORT int zpack_init_writer_heap(zpack_writer* writer, size_t initial_size);
ZPACK_EXPORT int zpack_write_header(zpack_writer* writer);
ZPACK_EXPORT int zpack_write_header_ex(zpack_writer* writer, zpack_u16 version);
ZPACK_EXPORT int zpack_write_data_header(zpack_writer* writer);
/* compress + hash `file_count` files in one device batch and append them (lib/zpack_write.c:280-343) */
ZPACK_EXPORT int zpack_write_files(zpack_writer* writer, zpack_file* files, zpack_u64 file_count);
ZPACK_EXPORT int zpack_write_files_from_archive(zpack_writer* writer, zpack_reader* reader, zpack_file_entry* entries, zpack_u64 file_count);
ZPACK_EXPORT int zpack_write_file_stream(zpack_writer* writer, zpack_compress_options* options, zpack_stream* stream, void* cctx);
ZPACK_EXPORT int zpack_write_file_stream_end(zpack_writer* writer, char* filename, zpack_compress_options* options, zpack_stream* stream, void* cctx);
ZPACK_EXPORT int zpack_write_cdr(zpack_writer* writer);
ZPACK_EXPORT int zpack_write_cdr_ex(zpack_writer* writer, zpack_file_entry* entries, zpack_u64 file_count);
ZPACK_EXPORT int zpack_write_eocdr(zpack_writer* writer);
ZPACK_EXPORT int zpack_write_eocdr_ex(zpack_writer* writer, zpack_u64 cdr_offset);
ZPACK_EXPORT int zpack_write_archive(zpack_writer* writer, zpack_file* files, zpack_u64 file_count);
ZPACK_EXPORT void zpack_close_writer(zpack_writer* writer);

/* ---------------------------------------------------------------- streams */
ZPACK_EXPORT int zpack_init_stream(zpack_stream* stream);
ZPACK_EXPORT void zpack_reset_stream(zpack_stream* stream);
ZPACK_EXPORT void zpack_close_stream(zpack_stream* stream);

/* ---------------------------------------------------------------- utilities */
ZPACK_EXPORT size_t zpack_get_dstream_in_size(zpack_compression_method method);
ZPACK_EXPORT size_t zpack_get_dstream_out_size(zpack_compression_method method);
ZPACK_EXPORT size_t zpack_get_cstream_in_size(zpack_compression_method method);
ZPACK_EXPORT size_t zpack_get_cstream_out_size(zpack_compression_method method);
ZPACK_EXPORT zpack_file_entry* zpack_get_file_entry(const char* filename, zpack_file_entry* file_entries, zpack_u64 file_count);
ZPACK_EXPORT zpack_bool zpack_read_stream_done(zpack_stream* stream, zpack_file_entry* entry);
#define ZPACK_READ_STREAM_DONE(stream, entry) \
    ((stream)->total_in == (entry)->comp_size && (stream)->read_back == 0)

/* opaque codec contexts (one per thread for concurrent buffer-backed reads, lib/zpack.h:337-340) */
ZPACK_EXPORT void* zpack_create_cctx(zpack_compression_method method);
ZPACK_EXPORT void* zpack_create_dctx(zpack_compression_method method);
ZPACK_EXPORT void zpack_free_cctx(zpack_compression_method method, void* cctx);
ZPACK_EXPORT void zpack_free_dctx(zpack_compression_method method, void* dctx);

#ifdef __cplusplus
}
#endif

#endif /* __ZPACK_H__ */
