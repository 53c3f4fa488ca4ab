/* ---- INTEGRATION.md section B, as code: the body that replaces zpack_compress_file (lib/zpack_write.c:161-224); the XXH3 of
 * zpack_add_written_file_entry (:256) becomes the hash the codec took of the source in the same pass. ---- */
#include "zpack_codec.h"
#include <pthread.h>

static pthread_mutex_t zpk_patch_wmu = PTHREAD_MUTEX_INITIALIZER;
static struct { const void* owner; zpk_codec* codec; } zpk_patch_wtab[64];
static __thread zpack_u64 zpk_patch_hash;

static zpk_codec* zpk_patch_wcodec(const void* owner)
{
    zpk_codec* c = NULL;
    int free_slot = -1;
    pthread_mutex_lock(&zpk_patch_wmu);
    for (int i = 0; i < 64; i++) {
        if (zpk_patch_wtab[i].owner == owner) { c = zpk_patch_wtab[i].codec; break; }
        if (!zpk_patch_wtab[i].owner && free_slot < 0) free_slot = i;
    }
    if (!c && free_slot >= 0 && zpk_codec_create(&c, -1) == ZPK_OK) { zpk_patch_wtab[free_slot].owner = owner; zpk_patch_wtab[free_slot].codec = c; }
    pthread_mutex_unlock(&zpk_patch_wmu);
    return c;
}

static void zpk_patch_wdrop(const void* owner)
{
    pthread_mutex_lock(&zpk_patch_wmu);
    for (int i = 0; i < 64; i++)
        if (zpk_patch_wtab[i].owner == owner) { zpk_codec_destroy(zpk_patch_wtab[i].codec); zpk_patch_wtab[i].owner = NULL; zpk_patch_wtab[i].codec = NULL; }
    pthread_mutex_unlock(&zpk_patch_wmu);
}

static int zpack_compress_file(zpack_writer* writer, zpack_u8* buffer, size_t capacity,
                               const zpack_file* file, zpack_u64* comp_size, void* cctx)
{
    (void)cctx;
    const zpack_compression_method m = file->options->method;
    if (m != ZPACK_COMPRESSION_NONE && m != ZPACK_COMPRESSION_ZSTD && m != ZPACK_COMPRESSION_LZ4)
        return ZPACK_ERROR_COMP_METHOD_INVALID;
    zpk_codec* codec = zpk_patch_wcodec(writer);
    if (!codec) return ZPACK_ERROR_NOT_AVAILABLE;
    zpk_encode_desc d;
    memset(&d, 0, sizeof(d));
    d.size = file->size; d.method = (uint32_t)m; d.level = file->options->level; d.dst_capacity = capacity;
    zpk_encode_result r;
    memset(&r, 0, sizeof(r));
    const uint8_t* src = file->buffer;
    uint8_t* dst = buffer;
    if (zpk_codec_encode_batch_host(codec, &src, &d, 1, &dst, &r) != ZPK_OK) return ZPACK_ERROR_NOT_AVAILABLE;
    writer->last_return = r.status ? (size_t)0 - r.detail : (size_t)r.comp_size;
    if (r.status) return r.status;
    *comp_size = r.comp_size;
    zpk_patch_hash = r.hash;          /* XXH3-64 of file->buffer, taken on the device while compressing */
    return ZPACK_OK;
}
