/* reader.c — archive reader of the zpack.h API.
 *
 * Container parsing (header / data signature / EOCDR / CDR -> file_entries[]) is plain host code and
 * follows the on-disk format of docs/specs.md exactly as the reference reader does
 * (lib/zpack_read.c:33-296).  The per-entry hot path — decompress + XXH3 verify, the
 * `switch (entry->comp_method)` of lib/zpack_read.c:350-468 — is ONE call into the GPU codec
 * (zpk_codec_decode_batch_host); zpack_read_file is a batch of one, zpack_read_files a batch of n.
 */
#include "internal.h"

/* ------------------------------------------------------------------ container parsing */

static int sig_is(const zpack_u8* p, zpack_u32 sig) { return zi_get32(p) == sig; }

int zpack_read_header_memory(const zpack_u8* buffer, zpack_u16* version)
{
    if (!sig_is(buffer, ZPACK_HEADER_SIGNATURE)) return ZPACK_ERROR_SIGNATURE_INVALID;
    *version = zi_get16(buffer + 4);
    if (*version < ZPACK_ARCHIVE_VERSION_MIN || *version > ZPACK_ARCHIVE_VERSION_MAX) return ZPACK_ERROR_VERSION_INCOMPATIBLE;
    return ZPACK_OK;
}

static int read_at(FILE* fp, zpack_u64 off, zpack_u8* dst, size_t n)
{
    if (zi_fseek(fp, off, SEEK_SET) != 0) return ZPACK_ERROR_SEEK_FAILED;
    if (n && fread(dst, n, 1, fp) != 1) return ZPACK_ERROR_READ_FAILED;
    return ZPACK_OK;
}

int zpack_read_header(FILE* fp, zpack_u16* version)
{
    zpack_u8 b[ZPACK_HEADER_SIZE];
    int rc = read_at(fp, 0, b, sizeof(b));
    return rc ? rc : zpack_read_header_memory(b, version);
}

int zpack_read_data_header_memory(const zpack_u8* buffer)
{
    return sig_is(buffer, ZPACK_DATA_SIGNATURE) ? ZPACK_OK : ZPACK_ERROR_SIGNATURE_INVALID;
}

int zpack_read_data_header(FILE* fp)
{
    zpack_u8 b[ZPACK_SIGNATURE_SIZE];
    int rc = read_at(fp, ZPACK_HEADER_SIZE, b, sizeof(b));          /* entry data starts right behind the header */
    return rc ? rc : zpack_read_data_header_memory(b);
}

int zpack_read_eocdr_memory(const zpack_u8* buffer, zpack_u64* cdr_offset)
{
    if (!sig_is(buffer, ZPACK_EOCDR_SIGNATURE)) return ZPACK_ERROR_SIGNATURE_INVALID;
    *cdr_offset = zi_get64(buffer + 4);
    return ZPACK_OK;
}

int zpack_read_eocdr(FILE* fp, zpack_u64 eocdr_offset, zpack_u64* cdr_offset)
{
    zpack_u8 b[ZPACK_EOCDR_SIZE];
    int rc = read_at(fp, eocdr_offset, b, sizeof(b));
    return rc ? rc : zpack_read_eocdr_memory(b, cdr_offset);
}

int zpack_read_cdr_header_memory(const zpack_u8* buffer, zpack_u64* count, zpack_u64* block_size)
{
    if (!sig_is(buffer, ZPACK_CDR_SIGNATURE)) return ZPACK_ERROR_SIGNATURE_INVALID;
    *count = zi_get64(buffer + 4);
    *block_size = zi_get64(buffer + 12);
    return ZPACK_OK;
}

/* The reference reads the 20-byte CDR header at cdr_offset after checking only cdr_offset < file_size
 * (lib/zpack_read.c:249-250), i.e. up to 19 bytes past the buffer.  Same verdicts where the reference's are defined
 * (a readable, wrong signature is SIGNATURE_INVALID), no out-of-bounds read where they are not. */
static int cdr_header_bounded(const zpack_u8* buffer, size_t size_left, zpack_u64* count, zpack_u64* block_size)
{
    if (size_left < ZPACK_SIGNATURE_SIZE) return ZPACK_ERROR_SIGNATURE_INVALID;
    if (!sig_is(buffer, ZPACK_CDR_SIGNATURE)) return ZPACK_ERROR_SIGNATURE_INVALID;
    if (size_left < ZPACK_CDR_HEADER_SIZE) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
    return zpack_read_cdr_header_memory(buffer, count, block_size);
}

/* one CDR record: u16 name length, name (no NUL), offset, comp_size, uncomp_size, hash, method */
int zpack_read_file_entry_memory(const zpack_u8* buffer, zpack_u64* size_left, zpack_file_entry* entry, size_t* entry_size)
{
    const zpack_u16 nlen = zi_get16(buffer);
    *entry_size = (size_t)ZPACK_FILE_ENTRY_FIXED_SIZE + nlen;
    if (*entry_size > *size_left) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
    *size_left -= *entry_size;
    entry->filename = (char*)malloc((size_t)nlen + 1);
    if (!entry->filename) return ZPACK_ERROR_MALLOC_FAILED;
    memcpy(entry->filename, buffer + 2, nlen);
    entry->filename[nlen] = '\0';
    const zpack_u8* f = buffer + 2 + nlen;
    entry->offset = zi_get64(f);
    entry->comp_size = zi_get64(f + 8);
    entry->uncomp_size = zi_get64(f + 16);
    entry->hash = zi_get64(f + 24);
    entry->comp_method = f[32];
    return ZPACK_OK;
}

int zpack_read_file_entries_memory(const zpack_u8* buffer, zpack_file_entry** entries, zpack_u64 header_count, zpack_u64 block_size,
                                   zpack_u64* count, zpack_u64* total_cs, zpack_u64* total_us)
{
    if (header_count > block_size / ZPACK_FILE_ENTRY_FIXED_SIZE) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
    if (header_count > SIZE_MAX / sizeof(zpack_file_entry)) return ZPACK_ERROR_MALLOC_FAILED;
    const size_t bytes = sizeof(zpack_file_entry) * (size_t)header_count;
    zpack_file_entry* table = (zpack_file_entry*)realloc(*entries, bytes ? bytes : 1);
    if (!table) return ZPACK_ERROR_MALLOC_FAILED;
    memset(table, 0, bytes);
    *entries = table;
    for (zpack_u64 i = 0; i < header_count; i++) {
        size_t used = 0;
        int rc = zpack_read_file_entry_memory(buffer, &block_size, table + i, &used);
        if (rc) return rc;
        buffer += used;
        ++*count;
        *total_cs += table[i].comp_size;
        *total_us += table[i].uncomp_size;
    }
    return ZPACK_OK;
}

int zpack_read_cdr_memory(const zpack_u8* buffer, size_t size_left, zpack_file_entry** entries, zpack_u64* count,
                          zpack_u64* total_cs, zpack_u64* total_us)
{
    zpack_u64 n = 0, block = 0;
    int rc = cdr_header_bounded(buffer, size_left, &n, &block);
    if (rc) return rc;
    if (block > (zpack_u64)size_left || ZPACK_CDR_HEADER_SIZE + block > (zpack_u64)size_left) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
    if (n == 0) return ZPACK_OK;
    return zpack_read_file_entries_memory(buffer + ZPACK_CDR_HEADER_SIZE, entries, n, block, count, total_cs, total_us);
}

int zpack_read_cdr(FILE* fp, zpack_u64 cdr_offset, zpack_file_entry** entries, zpack_u64* count, zpack_u64* total_cs, zpack_u64* total_us)
{
    zpack_u8 hdr[ZPACK_CDR_HEADER_SIZE];
    int rc = read_at(fp, cdr_offset, hdr, sizeof(hdr));
    if (rc) return rc;
    zpack_u64 n = 0, block = 0;
    if ((rc = zpack_read_cdr_header_memory(hdr, &n, &block))) return rc;
    if (block > SIZE_MAX) return ZPACK_ERROR_MALLOC_FAILED;
    if (n == 0) return ZPACK_OK;
    zpack_u8* body = (zpack_u8*)malloc((size_t)block ? (size_t)block : 1);
    if (!body) return ZPACK_ERROR_MALLOC_FAILED;
    if (block && fread(body, (size_t)block, 1, fp) != 1) { free(body); return ZPACK_ERROR_READ_FAILED; }
    rc = zpack_read_file_entries_memory(body, entries, n, block, count, total_cs, total_us);
    free(body);
    return rc;
}

/* CDR -> reader-owned entry table, with ALL filenames in one allocation (at 1 M entries the
 * reference's malloc + memcpy per name, lib/zpack_read.c:109-134, is the visible serial cost once decode is fast).
 * The arena hangs off reader->lz4f_dctx (documented in zpack.h: "name arena of a reader-owned table"); while it is set,
 * every filename of reader->file_entries points into it, the names are read-only, and zpack_close_reader / a re-parse
 * free the arena instead of the individual names.  The public zpack_read_cdr* / zpack_read_file_entries_memory keep the
 * reference's one-malloc-per-name contract for callers that own their tables. */
static void reader_drop_table(zpack_reader* reader)
{
    if (reader->file_entries) {
        zi_index_drop(reader->file_entries);
        if (!reader->lz4f_dctx)
            for (zpack_u64 i = 0; i < reader->file_count; i++) free(reader->file_entries[i].filename);
        free(reader->file_entries);
    }
    free(reader->lz4f_dctx);
    reader->file_entries = NULL; reader->lz4f_dctx = NULL;
    reader->file_count = 0; reader->comp_size = 0; reader->uncomp_size = 0;
}

static int reader_load_entries(zpack_reader* reader, const zpack_u8* cdr, size_t size_left)
{
    zpack_u64 n = 0, block = 0;
    int rc = cdr_header_bounded(cdr, size_left, &n, &block);
    if (rc) return rc;
    if (block > (zpack_u64)size_left || ZPACK_CDR_HEADER_SIZE + block > (zpack_u64)size_left) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
    if (n == 0) return ZPACK_OK;
    if (n > block / ZPACK_FILE_ENTRY_FIXED_SIZE) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
    if (n > SIZE_MAX / sizeof(zpack_file_entry)) return ZPACK_ERROR_MALLOC_FAILED;
    const zpack_u8* const body = cdr + ZPACK_CDR_HEADER_SIZE;
    /* pass 1: record sizes (the same checks, in the same order, as the record-by-record parser) */
    zpack_u64 left = block, names = 0;
    const zpack_u8* p = body;
    for (zpack_u64 i = 0; i < n; i++) {
        if (left < 2) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
        const zpack_u64 rec = (zpack_u64)ZPACK_FILE_ENTRY_FIXED_SIZE + zi_get16(p);
        if (rec > left) return ZPACK_ERROR_BLOCK_SIZE_INVALID;
        names += (zpack_u64)zi_get16(p) + 1;
        left -= rec; p += rec;
    }
    zpack_file_entry* table = (zpack_file_entry*)realloc(reader->file_entries, sizeof(zpack_file_entry) * (size_t)n);
    char* arena = (char*)malloc((size_t)names);
    if (!table || !arena) { free(arena); if (table) reader->file_entries = table; return ZPACK_ERROR_MALLOC_FAILED; }
    reader->file_entries = table;
    reader->lz4f_dctx = arena;
    zi_index_register(table, n);
    /* pass 2 */
    p = body;
    char* w = arena;
    for (zpack_u64 i = 0; i < n; i++) {
        const zpack_u16 nlen = zi_get16(p);
        zpack_file_entry* e = table + i;
        e->filename = w;
        memcpy(w, p + 2, nlen); w[nlen] = '\0'; w += (size_t)nlen + 1;
        const zpack_u8* f = p + 2 + nlen;
        e->offset = zi_get64(f);
        e->comp_size = zi_get64(f + 8);
        e->uncomp_size = zi_get64(f + 16);
        e->hash = zi_get64(f + 24);
        e->comp_method = f[32];
        reader->comp_size += e->comp_size;
        reader->uncomp_size += e->uncomp_size;
        p = f + 33;
    }
    reader->file_count += n;
    return ZPACK_OK;
}

int zpack_read_archive_memory(zpack_reader* reader)
{
    if (!reader->buffer) return ZPACK_ERROR_ARCHIVE_NOT_LOADED;
    if (reader->file_size < ZPACK_MINIMUM_ARCHIVE_SIZE) return ZPACK_ERROR_FILE_TOO_SMALL;
    const zpack_u8* a = reader->buffer;
    int rc;
    if ((rc = zpack_read_header_memory(a, &reader->version))) return rc;
    if ((rc = zpack_read_data_header_memory(a + ZPACK_HEADER_SIZE))) return rc;
    reader->eocdr_offset = reader->file_size - ZPACK_EOCDR_SIZE;
    if ((rc = zpack_read_eocdr_memory(a + reader->eocdr_offset, &reader->cdr_offset))) return rc;
    if (reader->cdr_offset >= reader->file_size) return ZPACK_ERROR_READ_FAILED;
    if (reader->lz4f_dctx) reader_drop_table(reader);     /* an arena table from an earlier parse of this reader: start over */
    if (reader->file_entries)                             /* a table the caller put there (public-API parse): keep its contract */
        return zpack_read_cdr_memory(a + reader->cdr_offset, reader->file_size - (size_t)reader->cdr_offset, &reader->file_entries,
                                     &reader->file_count, &reader->comp_size, &reader->uncomp_size);
    return reader_load_entries(reader, a + reader->cdr_offset, reader->file_size - (size_t)reader->cdr_offset);
}

int zpack_read_archive(zpack_reader* reader)
{
    if (!reader->file) return ZPACK_ERROR_ARCHIVE_NOT_LOADED;
    if (zi_fseek(reader->file, 0, SEEK_END) != 0) return ZPACK_ERROR_SEEK_FAILED;
    if (!reader->file_size) reader->file_size = (size_t)zi_ftell(reader->file);
    if (reader->file_size < ZPACK_MINIMUM_ARCHIVE_SIZE) return ZPACK_ERROR_FILE_TOO_SMALL;
    int rc;
    if ((rc = zpack_read_header(reader->file, &reader->version))) return rc;
    if ((rc = zpack_read_data_header(reader->file))) return rc;
    reader->eocdr_offset = reader->file_size - ZPACK_EOCDR_SIZE;
    if ((rc = zpack_read_eocdr(reader->file, reader->eocdr_offset, &reader->cdr_offset))) return rc;
    if (reader->lz4f_dctx) reader_drop_table(reader);
    if (reader->file_entries)
        return zpack_read_cdr(reader->file, reader->cdr_offset, &reader->file_entries, &reader->file_count,
                              &reader->comp_size, &reader->uncomp_size);
    {   /* the whole CDR in one read, then the arena parse */
        zpack_u8 hdr[ZPACK_CDR_HEADER_SIZE];
        if ((rc = read_at(reader->file, reader->cdr_offset, hdr, sizeof(hdr)))) return rc;
        zpack_u64 n = 0, block = 0;
        if ((rc = zpack_read_cdr_header_memory(hdr, &n, &block))) return rc;
        if (n == 0) return ZPACK_OK;
        if (block > SIZE_MAX - ZPACK_CDR_HEADER_SIZE) return ZPACK_ERROR_MALLOC_FAILED;
        zpack_u8* cdr = (zpack_u8*)malloc(ZPACK_CDR_HEADER_SIZE + (size_t)block);
        if (!cdr) return ZPACK_ERROR_MALLOC_FAILED;
        memcpy(cdr, hdr, sizeof(hdr));
        if (block && fread(cdr + ZPACK_CDR_HEADER_SIZE, (size_t)block, 1, reader->file) != 1) { free(cdr); return ZPACK_ERROR_READ_FAILED; }
        rc = reader_load_entries(reader, cdr, ZPACK_CDR_HEADER_SIZE + (size_t)block);
        free(cdr);
        return rc;
    }
}

/* ------------------------------------------------------------------ raw (still compressed) access */

int zpack_read_raw_file(zpack_reader* reader, zpack_file_entry* entry, zpack_u8* buffer, size_t max_size)
{
    if (entry->offset + entry->comp_size > reader->file_size) return ZPACK_ERROR_FILE_OFFSET_INVALID;       /* `>`: lib/zpack_read.c:301 */
    const size_t n = (zpack_u64)max_size < entry->comp_size ? max_size : (size_t)entry->comp_size;
    if (reader->file) return read_at(reader->file, entry->offset, buffer, n);
    if (reader->buffer) { memcpy(buffer, reader->buffer + entry->offset, n); return ZPACK_OK; }
    return ZPACK_ERROR_ARCHIVE_NOT_LOADED;
}

int zpack_read_raw_file_stream(zpack_reader* reader, zpack_file_entry* entry, zpack_stream* stream, size_t* in_size)
{
    if (entry->comp_size == 0) return ZPACK_OK;
    if (entry->offset + entry->comp_size > reader->file_size) return ZPACK_ERROR_FILE_OFFSET_INVALID;
    if (!stream->next_in || !stream->avail_in || stream->total_in > entry->comp_size) return ZPACK_ERROR_STREAM_INVALID;
    zpack_u64 left = entry->comp_size - stream->total_in;
    size_t n = (zpack_u64)stream->avail_in < left ? stream->avail_in : (size_t)left;
    if (n == 0) return ZPACK_OK;
    const zpack_u64 at = entry->offset + stream->total_in;
    if (reader->file) { int rc = read_at(reader->file, at, stream->next_in, n); if (rc) return rc; }
    else if (reader->buffer) memcpy(stream->next_in, reader->buffer + at, n);
    else return ZPACK_ERROR_ARCHIVE_NOT_LOADED;
    stream->next_in += n; stream->avail_in -= n; stream->total_in += n;
    *in_size = n;
    return ZPACK_OK;
}

/* ------------------------------------------------------------------ the hot path: decode + verify */

static void fill_desc(zpk_decode_desc* d, const zpack_file_entry* e, zpack_u64 src_offset, size_t max_size)
{
    memset(d, 0, sizeof(*d));
    d->src_offset = src_offset;
    d->comp_size = e->comp_size;
    d->uncomp_size = e->uncomp_size;
    d->expect_hash = e->hash;
    d->dst_capacity = max_size;
    d->method = e->comp_method;
}

typedef struct {
    zi_ctx* ctx; const zpack_u8* image; zpack_u64 image_size; const zpk_decode_desc* desc; zpack_u8* const* buffers;
    zpk_decode_result* res; zpack_u64 cut[ZI_MAX_DEVICES + 1]; int rc[ZI_MAX_DEVICES];
} read_job;

static void read_part(void* arg, int k)
{
    read_job* j = (read_job*)arg;
    const zpack_u64 lo = j->cut[k], n = j->cut[k + 1] - lo;
    j->rc[k] = n ? zpk_codec_decode_batch_host(j->ctx->dev[k], j->image, j->image_size, j->desc + lo, n, j->buffers + lo, j->res + lo) : ZPK_OK;
}

/* n entries in ONE device batch.  Memory-backed readers hand the codec the archive image itself
 * (zero-copy on the host side, as the reference does at lib/zpack_read.c:345-346); file-backed readers
 * gather the payloads first (the reference mallocs + freads per entry, :336-344). */
static int read_batch(zpack_reader* reader, zpack_file_entry* const* entries, zpack_u64 count,
                      zpack_u8* const* buffers, const size_t* max_sizes, int* results, void* dctx)
{
    if (count == 0) return ZPACK_OK;
    if (!reader->file && !reader->buffer) return ZPACK_ERROR_ARCHIVE_NOT_LOADED;
    /* the caller's context, else this reader's own, created on first use (lib/zpack_read.c:17-31) and freed by zpack_close_reader */
    zi_ctx* ctx = zi_pick_ctx(dctx, &reader->zstd_dctx);
    if (!ctx) return ZPACK_ERROR_NOT_AVAILABLE;              /* no HIP device: there is no CPU fallback */

    zpk_decode_desc* desc = (zpk_decode_desc*)calloc((size_t)count, sizeof(*desc));
    zpk_decode_result* res = (zpk_decode_result*)calloc((size_t)count, sizeof(*res));
    if (!desc || !res) { free(desc); free(res); return ZPACK_ERROR_MALLOC_FAILED; }
    const zpack_u8* image = reader->buffer;
    zpack_u64 image_size = reader->file_size;
    zpack_u8* gathered = NULL;
    int rc = ZPACK_OK;

    if (reader->file) {
        /* pack the payloads of the entries that pass the reference's guards behind a 1-byte pad, so that
         * the device sees the same `offset + comp_size >= file_size` verdicts (lib/zpack_read.c:331) */
        zpack_u64 total = 1;
        for (zpack_u64 i = 0; i < count; i++)
            if (entries[i]->comp_size && entries[i]->offset + entries[i]->comp_size < reader->file_size) total += entries[i]->comp_size;
        gathered = (zpack_u8*)malloc((size_t)total + 1);
        if (!gathered) { free(desc); free(res); return ZPACK_ERROR_MALLOC_FAILED; }
        zpack_u64 pos = 0;
        for (zpack_u64 i = 0; i < count && rc == ZPACK_OK; i++) {
            const zpack_file_entry* e = entries[i];
            if (e->comp_size && e->offset + e->comp_size < reader->file_size) {
                rc = read_at(reader->file, e->offset, gathered + pos, (size_t)e->comp_size);
                fill_desc(&desc[i], e, pos, max_sizes[i]);
                pos += e->comp_size;
            } else {
                fill_desc(&desc[i], e, total + 1, max_sizes[i]);                 /* fails the offset guard on the device */
            }
        }
        gathered[pos] = 0;
        image = gathered; image_size = total + 1;
        /* entries that the reference would reject with BUFFER_TOO_SMALL before looking at the offset keep
         * that order: the device checks comp_size==0, then max_size, then the offset — same as :328-332 */
    } else {
        for (zpack_u64 i = 0; i < count; i++) fill_desc(&desc[i], entries[i], entries[i]->offset, max_sizes[i]);
    }
    if (rc == ZPACK_OK) {
        read_job job = { ctx, image, image_size, desc, buffers, res, {0}, {0} };
        int parts = ctx->n;
        if ((zpack_u64)parts > count) parts = (int)count;
        if (parts > 1) {
            /* static shard: contiguous ranges of the batch balanced by comp + uncomp bytes, one host thread
             * and one codec per device, nothing exchanged; results land in their own slices of res[] */
            zpack_u64* w = (zpack_u64*)malloc(sizeof(zpack_u64) * (size_t)count);
            if (!w) parts = 1;
            else {
                for (zpack_u64 i = 0; i < count; i++) w[i] = entries[i]->comp_size + entries[i]->uncomp_size;
                zi_split(w, count, parts, job.cut);
                free(w);
            }
        }
        if (parts <= 1) { job.cut[0] = 0; job.cut[1] = count; parts = 1; }
        zi_parallel(parts, read_part, &job);
        for (int k = 0; k < parts; k++) if (job.rc[k] != ZPK_OK) rc = ZPACK_ERROR_NOT_AVAILABLE;
    }
    if (rc == ZPACK_OK)
        for (zpack_u64 i = 0; i < count; i++) {
            results[i] = res[i].status;
            if (res[i].status != ZPACK_OK || i + 1 == count) reader->last_return = res[i].status ? (size_t)0 - res[i].detail : 0;
        }
    free(gathered); free(desc); free(res);
    return rc;
}

int zpack_read_file(zpack_reader* reader, zpack_file_entry* entry, zpack_u8* buffer, size_t max_size, void* dctx)
{
    /* the cheap guards are answered on the host without a device round trip, in the reference's order */
    if (entry->comp_size == 0) return ZPACK_OK;
    if ((zpack_u64)max_size < entry->uncomp_size) return ZPACK_ERROR_BUFFER_TOO_SMALL;
    if (entry->offset + entry->comp_size >= reader->file_size) return ZPACK_ERROR_FILE_OFFSET_INVALID;
    if (!reader->file && !reader->buffer) return ZPACK_ERROR_ARCHIVE_NOT_LOADED;
    int result = ZPACK_OK;
    zpack_file_entry* one = entry;
    zpack_u8* out = buffer;
    int rc = read_batch(reader, &one, 1, &out, &max_size, &result, dctx);
    return rc ? rc : result;
}

int zpack_read_files(zpack_reader* reader, zpack_file_entry* const* entries, zpack_u64 count,
                     zpack_u8* const* buffers, const size_t* max_sizes, int* results, void* dctx)
{
    if (!reader || (count && (!entries || !buffers || !max_sizes || !results))) return ZPACK_ERROR_STREAM_INVALID;
    return read_batch(reader, entries, count, buffers, max_sizes, results, dctx);
}

int zpack_read_files_packed(zpack_reader* reader, zpack_file_entry* const* entries, zpack_u64 count,
                            zpack_u8* buffer, size_t buffer_size, zpack_u64* out_offsets, int* results, void* dctx)
{
    if (!reader || (count && (!entries || !buffer || !out_offsets || !results))) return ZPACK_ERROR_STREAM_INVALID;
    zpack_u8** bufs = (zpack_u8**)malloc(sizeof(zpack_u8*) * (size_t)(count ? count : 1));
    size_t* caps = (size_t*)malloc(sizeof(size_t) * (size_t)(count ? count : 1));
    if (!bufs || !caps) { free(bufs); free(caps); return ZPACK_ERROR_MALLOC_FAILED; }
    zpack_u64 pos = 0;
    for (zpack_u64 i = 0; i < count; i++) {
        out_offsets[i] = pos;
        zpack_u64 room = pos <= buffer_size ? buffer_size - pos : 0;
        caps[i] = (size_t)(entries[i]->uncomp_size <= room ? entries[i]->uncomp_size : room);   /* too little room => BUFFER_TOO_SMALL */
        bufs[i] = buffer + (pos <= buffer_size ? pos : buffer_size);
        pos += entries[i]->uncomp_size;
    }
    int rc = read_batch(reader, entries, count, bufs, caps, results, dctx);
    free(bufs); free(caps);
    return rc;
}

static int zi_stream_replay(zpack_reader* reader, const zpack_file_entry* entry, zpk_dstream* d, zpack_u64 upto)
{
    const size_t piece = (size_t)1 << 20;
    zpack_u8* tmp = (zpack_u8*)malloc(piece);
    if (!tmp) return ZPACK_ERROR_MALLOC_FAILED;
    int rc = ZPACK_OK, first = 1;
    for (zpack_u64 off = 0; (off < upto || first) && rc == ZPACK_OK; ) {
        const size_t n = (size_t)(upto - off < piece ? upto - off : piece);
        const zpack_u64 at = entry->offset + off;
        if (n) {
            if (reader->file) rc = read_at(reader->file, at, tmp, n);
            else if (reader->buffer) memcpy(tmp, reader->buffer + at, n);
            else rc = ZPACK_ERROR_ARCHIVE_NOT_LOADED;
        }
        if (rc == ZPACK_OK) rc = zpk_dstream_replay(d, entry->comp_method, entry->comp_size, entry->uncomp_size, entry->hash, tmp, n, first);
        first = 0; off += n;
        if (n == 0) break;
    }
    free(tmp);
    return rc;
}

/* streaming read (lib/zpack_read.c:515-640): the same observable protocol — the library pulls the
 * compressed bytes into the caller's input window, output arrives in avail_out pieces, read_back != 0
 * means "call again" — served by aggregating the entry and decoding it in one device batch. */
int zpack_read_file_stream(zpack_reader* reader, zpack_file_entry* entry, zpack_stream* stream, void* dctx)
{
    if (entry->comp_size == 0 || ZPACK_READ_STREAM_DONE(stream, entry)) return ZPACK_OK;
    if (!stream->next_out || !stream->avail_out) return ZPACK_ERROR_STREAM_INVALID;
    zi_stream_state* st = (zi_stream_state*)stream->xxh3_state;
    if (!st) return ZPACK_ERROR_STREAM_INVALID;
    zi_ctx* ctx = zi_pick_ctx(dctx, &reader->zstd_dctx);
    if (!ctx) return ZPACK_ERROR_NOT_AVAILABLE;
    if (!st->d && zpk_dstream_create(ctx->dev[0], &st->d) != ZPK_OK) return ZPACK_ERROR_MALLOC_FAILED;
    zpk_dstream_bind(st->d, ctx->dev[0]);                     /* a stream outlives readers: it decodes with THIS call's context */
    if (stream->total_in == 0 && stream->read_back == 0) { zpk_dstream_reset(st->d); st->d_active = 1; }

    zpack_u8* src = stream->next_in;
    size_t have = 0;
    if (stream->read_back) {           /* bytes the caller re-presented: already aggregated, just step over them */
        stream->next_in += stream->read_back;
        stream->avail_in -= stream->read_back;
        src = stream->next_in;
        stream->read_back = 0;
    }
    /* (a stream that decodes in bounded memory takes no input while output is waiting: nothing is pulled then) */
    if (stream->total_in < entry->comp_size && zpk_dstream_wants_input(st->d)) {
        size_t got = 0;
        int rc = zpack_read_raw_file_stream(reader, entry, stream, &got);
        if (rc) return rc;
        have = got;
    }
    size_t consumed = 0, produced = 0;
    int done = 0;
    int rc = zpk_dstream_step(st->d, entry->comp_method, entry->comp_size, entry->uncomp_size, entry->hash,
                              src, have, &consumed, stream->next_out, stream->avail_out, &produced, &done);
    if (rc == ZPK_DS_RESTART) {
        /* not an entry the bounded steps can decide: the bytes read so far go through the stream again (1 MiB at a time out of
         * the archive), it continues in its windowless form — where every verdict is — and hands out nothing twice */
        rc = zi_stream_replay(reader, entry, st->d, stream->total_in);
        if (rc == ZPACK_OK)
            rc = zpk_dstream_step(st->d, entry->comp_method, entry->comp_size, entry->uncomp_size, entry->hash,
                                  src, 0, &consumed, stream->next_out, stream->avail_out, &produced, &done);
    }
    stream->next_out += produced; stream->avail_out -= produced; stream->total_out += produced;
    if (rc != ZPACK_OK && rc != ZPACK_ERROR_FILE_HASH_MISMATCH) { reader->last_return = (size_t)-1; return rc; }
    if (stream->total_in == entry->comp_size && !done) stream->read_back = 1;   /* output still pending: not DONE yet */
    if (ZPACK_READ_STREAM_DONE(stream, entry)) return rc;                        /* hash verdict arrives with the last byte */
    return ZPACK_OK;
}

/* ------------------------------------------------------------------ lifecycle */

int zpack_init_reader(zpack_reader* reader, const char* path)
{
    FILE* fp = fopen(path, "rb");
    if (!fp) return ZPACK_ERROR_OPEN_FAILED;
    if (reader->file) fclose(reader->file);
    reader->file = fp;
    return zpack_read_archive(reader);
}

int zpack_init_reader_cfile(zpack_reader* reader, FILE* fp)
{
    reader->file = fp;
    return zpack_read_archive(reader);
}

int zpack_init_reader_memory(zpack_reader* reader, const zpack_u8* buffer, size_t size)
{
    reader->buffer = (zpack_u8*)malloc(size ? size : 1);
    if (!reader->buffer) return ZPACK_ERROR_MALLOC_FAILED;
    memcpy(reader->buffer, buffer, size);
    reader->file_size = size;
    reader->buffer_shared = ZPACK_FALSE;
    return zpack_read_archive_memory(reader);
}

int zpack_init_reader_memory_shared(zpack_reader* reader, zpack_u8* buffer, size_t size)
{
    reader->buffer = buffer;
    reader->file_size = size;
    reader->buffer_shared = ZPACK_TRUE;
    return zpack_read_archive_memory(reader);
}

void zpack_reset_reader_dctx(zpack_reader* reader)
{
    zi_ctx_reset((zi_ctx*)reader->zstd_dctx);                   /* lib/zpack_read.c:679-690: after an abandoned stream / error */
}

void zpack_close_reader(zpack_reader* reader)
{
    if (reader->file) fclose(reader->file);
    if (!reader->buffer_shared) free(reader->buffer);
    reader_drop_table(reader);                                 /* names in one arena (reader_load_entries) or one malloc each */
    zi_ctx_destroy((zi_ctx*)reader->zstd_dctx);
    memset(reader, 0, sizeof(*reader));
}
