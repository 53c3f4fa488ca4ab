/* util.c — codec context plumbing and small helpers for the host library. */
#include "internal.h"
#include <pthread.h>

static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static zpk_codec* g_codec = NULL;

static void make_default(void)
{
    zpk_codec* c = NULL;
    if (zpk_codec_create(&c, -1) == ZPK_OK) g_codec = c;
}

zpk_codec* zi_default_codec(void)
{
    pthread_once(&g_once, make_default);
    return g_codec;
}

zpk_codec* zi_pick_codec(void* explicit_ctx, void** owner_slot)
{
    if (explicit_ctx) return (zpk_codec*)explicit_ctx;
    if (owner_slot && *owner_slot) return (zpk_codec*)*owner_slot;
    return zi_default_codec();
}

/* contexts are opaque codec handles; one per thread for concurrent buffer-backed reads
 * (the threading contract of lib/zpack.h:337-340).  Stands where the reference hands out
 * ZSTD_DCtx / LZ4F_dctx / ZSTD_CCtx / LZ4F_cctx objects (lib/zpack_read.c:776-812, zpack_write.c:899-935). */
static void* make_ctx(zpack_compression_method method)
{
    if (method != ZPACK_COMPRESSION_ZSTD && method != ZPACK_COMPRESSION_LZ4) return NULL;
    zpk_codec* c = NULL;
    if (zpk_codec_create(&c, -1) != ZPK_OK) return NULL;
    return c;
}

void* zpack_create_dctx(zpack_compression_method method) { return make_ctx(method); }
void* zpack_create_cctx(zpack_compression_method method) { return make_ctx(method); }
void zpack_free_dctx(zpack_compression_method method, void* dctx) { (void)method; zpk_codec_destroy((zpk_codec*)dctx); }
void zpack_free_cctx(zpack_compression_method method, void* cctx) { (void)method; zpk_codec_destroy((zpk_codec*)cctx); }

/* advisory stream buffer sizes — the values the reference reports with lz4 1.9.3 / zstd 1.4.9
 * (lib/zpack_read.c:719-758, lib/zpack_write.c:858-897); method NONE takes the largest */
size_t zpack_get_dstream_in_size(zpack_compression_method m)
{
    switch (m) { case ZPACK_COMPRESSION_NONE: case ZPACK_COMPRESSION_ZSTD: return 131075; case ZPACK_COMPRESSION_LZ4: return 65551; default: return 0; }
}
size_t zpack_get_dstream_out_size(zpack_compression_method m)
{
    switch (m) { case ZPACK_COMPRESSION_NONE: case ZPACK_COMPRESSION_ZSTD: return 131072; case ZPACK_COMPRESSION_LZ4: return 65536; default: return 0; }
}
size_t zpack_get_cstream_in_size(zpack_compression_method m)
{
    switch (m) { case ZPACK_COMPRESSION_NONE: case ZPACK_COMPRESSION_ZSTD: return 131072; case ZPACK_COMPRESSION_LZ4: return 65536; default: return 0; }
}
size_t zpack_get_cstream_out_size(zpack_compression_method m)
{
    switch (m) { case ZPACK_COMPRESSION_NONE: case ZPACK_COMPRESSION_ZSTD: return 131591; case ZPACK_COMPRESSION_LZ4: return 65551; default: return 0; }
}

zpack_file_entry* zpack_get_file_entry(const char* filename, zpack_file_entry* file_entries, zpack_u64 file_count)
{
    for (zpack_u64 i = 0; i < file_count; i++)
        if (strcmp(file_entries[i].filename, filename) == 0) return file_entries + i;
    return NULL;
}

zpack_bool zpack_read_stream_done(zpack_stream* stream, zpack_file_entry* entry)
{
    return ZPACK_READ_STREAM_DONE(stream, entry);
}
