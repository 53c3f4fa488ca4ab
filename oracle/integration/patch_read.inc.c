/* ---- INTEGRATION.md section B, as code: the body that replaces zpack_read_file (lib/zpack_read.c:326-471) in a patched copy
 * of the reference.  The container code around it stays the reference's own; only the per-entry codec + XXH3 go to the GPU. ---- */
#include "zpack_codec.h"
#include <pthread.h>

/* one codec per reader, kept beside the reader (its two context fields stay what the streaming path expects them to be) */
static pthread_mutex_t zpk_patch_mu = PTHREAD_MUTEX_INITIALIZER;
static struct { const void* owner; zpk_codec* codec; } zpk_patch_tab[64];

static zpk_codec* zpk_patch_codec(const void* owner)
{
    zpk_codec* c = NULL;
    int free_slot = -1;
    pthread_mutex_lock(&zpk_patch_mu);
    for (int i = 0; i < 64; i++) {
        if (zpk_patch_tab[i].owner == owner) { c = zpk_patch_tab[i].codec; break; }
        if (!zpk_patch_tab[i].owner && free_slot < 0) free_slot = i;
    }
    if (!c && free_slot >= 0 && zpk_codec_create(&c, -1) == ZPK_OK) { zpk_patch_tab[free_slot].owner = owner; zpk_patch_tab[free_slot].codec = c; }
    pthread_mutex_unlock(&zpk_patch_mu);
    return c;
}

static void zpk_patch_drop(const void* owner)
{
    pthread_mutex_lock(&zpk_patch_mu);
    for (int i = 0; i < 64; i++)
        if (zpk_patch_tab[i].owner == owner) { zpk_codec_destroy(zpk_patch_tab[i].codec); zpk_patch_tab[i].owner = NULL; zpk_patch_tab[i].codec = NULL; }
    pthread_mutex_unlock(&zpk_patch_mu);
}

int zpack_read_file(zpack_reader* reader, zpack_file_entry* entry, zpack_u8* buffer, size_t max_size, void* dctx)
{
    (void)dctx;             /* library-native contexts have no meaning to the codec: the reader's own codec serves every call */
    if (entry->comp_size == 0) return ZPACK_OK;
    if (max_size < entry->uncomp_size) return ZPACK_ERROR_BUFFER_TOO_SMALL;
    if (entry->offset + entry->comp_size >= reader->file_size)
        return ZPACK_ERROR_FILE_OFFSET_INVALID;

    zpack_u8* comp_data;
    if (reader->file)
    {
        if (entry->comp_size > SIZE_MAX - 1) return ZPACK_ERROR_MALLOC_FAILED;
        comp_data = (zpack_u8*)malloc(sizeof(zpack_u8) * (entry->comp_size + 1));
        if (comp_data == NULL) return ZPACK_ERROR_MALLOC_FAILED;
        int ret;
        if ((ret = zpack_read_raw_file(reader, entry, comp_data, entry->comp_size)))
        {
            free(comp_data);
            return ret;
        }
        comp_data[entry->comp_size] = 0;
    }
    else if (reader->buffer)
        comp_data = reader->buffer + entry->offset;
    else
        return ZPACK_ERROR_ARCHIVE_NOT_LOADED;

    zpk_codec* codec = zpk_patch_codec(reader);
    int result = ZPACK_ERROR_NOT_AVAILABLE;
    if (codec)
    {
        zpk_decode_desc d;
        memset(&d, 0, sizeof(d));
        d.comp_size = entry->comp_size; d.uncomp_size = entry->uncomp_size; d.expect_hash = entry->hash;
        d.dst_capacity = max_size; d.method = entry->comp_method;
        zpk_decode_result r;
        memset(&r, 0, sizeof(r));
        /* image = the frame + one byte behind it (the pad byte of the file case, the next archive byte of the memory case):
         * the `>=` guard of :331 was evaluated above on the real offsets and passes here by construction */
        if (zpk_codec_decode_batch_host(codec, comp_data, entry->comp_size + 1, &d, 1, &buffer, &r) == ZPK_OK)
        {
            reader->last_return = r.status ? (size_t)0 - r.detail : 0;
            result = r.status;
        }
    }
    if (reader->file) free(comp_data);
    return result;
}
