/* writer.c — archive writer of the zpack.h API.
 *
 * Container emission (header, data signature, CDR, EOCDR) is host code following docs/specs.md like the
 * reference writer (lib/zpack_write.c:60-123, :687-829).  The hot path — zpack_compress_file +
 * the XXH3 of zpack_add_written_file_entry (lib/zpack_write.c:161-260), looped per file at :287-339 —
 * is ONE batched call into the GPU codec (zpk_codec_encode_batch_host); the serial
 * `write_offset += comp_size` of :338 becomes a prefix sum over the batch's compressed sizes.
 */
#include "internal.h"

/* ------------------------------------------------------------------ sink + entry table */

static zpack_u64 pow2_at_least(zpack_u64 n) { zpack_u64 b = 1; while (b < n) b <<= 1; return b; }

/* append `size` bytes at the writer's cursor: file (seek + write) or heap (power-of-two growth,
 * like lib/zpack_common.c:83-104) */
int zi_writer_put(zpack_writer* w, const zpack_u8* data, size_t size)
{
    if (w->file) {
        if (zi_fseek(w->file, w->write_offset, SEEK_SET) != 0) return ZPACK_ERROR_SEEK_FAILED;
        if (size && fwrite(data, 1, size, w->file) != size) return ZPACK_ERROR_WRITE_FAILED;
    } else if (w->buffer) {
        const zpack_u64 need = (zpack_u64)w->file_size + size;
        if ((zpack_u64)w->buffer_capacity < need) {
            const zpack_u64 cap = pow2_at_least(need);
            if (cap > SIZE_MAX) return ZPACK_ERROR_MALLOC_FAILED;
            zpack_u8* nb = (zpack_u8*)realloc(w->buffer, (size_t)cap);
            if (!nb) return ZPACK_ERROR_MALLOC_FAILED;
            w->buffer = nb; w->buffer_capacity = (size_t)cap;
        }
        if (size) memcpy(w->buffer + w->write_offset, data, size);
    } else {
        return ZPACK_ERROR_WRITER_NOT_OPENED;
    }
    w->write_offset += size;
    w->file_size += size;
    return ZPACK_OK;
}

zpack_file_entry* zi_writer_push_entry(zpack_writer* w)
{
    if (w->file_count + 1 > w->fe_capacity) {
        const zpack_u64 cap = pow2_at_least(w->file_count + 1);
        if (cap > SIZE_MAX / sizeof(zpack_file_entry)) return NULL;
        zpack_file_entry* t = (zpack_file_entry*)realloc(w->file_entries, sizeof(zpack_file_entry) * (size_t)cap);
        if (!t) return NULL;
        w->file_entries = t; w->fe_capacity = cap;
    }
    zpack_file_entry* e = w->file_entries + w->file_count++;
    memset(e, 0, sizeof(*e));
    return e;
}

static char* dup_name(const char* s)
{
    const size_t n = strlen(s) + 1;
    char* d = (char*)malloc(n);
    if (d) memcpy(d, s, n);
    return d;
}

/* ------------------------------------------------------------------ open */

int zpack_init_writer(zpack_writer* writer, const char* path)
{
    writer->file = fopen(path, "wb");
    return writer->file ? ZPACK_OK : ZPACK_ERROR_OPEN_FAILED;
}

int zpack_init_writer_cfile(zpack_writer* writer, FILE* fp)
{
    if (!fp) return ZPACK_ERROR_OPEN_FAILED;
    writer->file = fp;
    return ZPACK_OK;
}

int zpack_init_writer_heap(zpack_writer* writer, size_t initial_size)
{
    const size_t least = ZPACK_HEADER_SIZE + ZPACK_SIGNATURE_SIZE;
    writer->buffer_capacity = initial_size > least ? initial_size : least;
    writer->buffer = (zpack_u8*)malloc(writer->buffer_capacity);
    return writer->buffer ? ZPACK_OK : ZPACK_ERROR_MALLOC_FAILED;
}

/* ------------------------------------------------------------------ fixed blocks */

int zpack_write_header_ex(zpack_writer* writer, zpack_u16 version)
{
    zpack_u8 b[ZPACK_HEADER_SIZE];
    zi_put32(b, ZPACK_HEADER_SIGNATURE);
    zi_put16(b + 4, version);
    return zi_writer_put(writer, b, sizeof(b));
}

int zpack_write_header(zpack_writer* writer) { return zpack_write_header_ex(writer, ZPACK_ARCHIVE_VERSION_MAX); }

int zpack_write_data_header(zpack_writer* writer)
{
    zpack_u8 b[ZPACK_SIGNATURE_SIZE];
    zi_put32(b, ZPACK_DATA_SIGNATURE);
    return zi_writer_put(writer, b, sizeof(b));
}

int zpack_write_cdr_ex(zpack_writer* writer, zpack_file_entry* entries, zpack_u64 file_count)
{
    if (!writer->file && !writer->buffer) return ZPACK_ERROR_WRITER_NOT_OPENED;
    zpack_u64 block = 0;
    for (zpack_u64 i = 0; i < file_count; i++) {
        const size_t n = strlen(entries[i].filename);
        if (n > ZPACK_MAX_FILENAME_LENGTH) return ZPACK_ERROR_FILENAME_TOO_LONG;
        block += ZPACK_FILE_ENTRY_FIXED_SIZE + n;
    }
    const zpack_u64 total = ZPACK_CDR_HEADER_SIZE + block;
    if (total > SIZE_MAX) return ZPACK_ERROR_MALLOC_FAILED;
    zpack_u8* img = (zpack_u8*)malloc((size_t)total);
    if (!img) return ZPACK_ERROR_MALLOC_FAILED;
    zi_put32(img, ZPACK_CDR_SIGNATURE);
    zi_put64(img + 4, file_count);
    zi_put64(img + 12, block);
    zpack_u8* p = img + ZPACK_CDR_HEADER_SIZE;
    for (zpack_u64 i = 0; i < file_count; i++) {
        const zpack_file_entry* e = entries + i;
        const zpack_u16 n = (zpack_u16)strlen(e->filename);
        zi_put16(p, n); memcpy(p + 2, e->filename, n); p += 2 + n;
        zi_put64(p, e->offset); zi_put64(p + 8, e->comp_size); zi_put64(p + 16, e->uncomp_size); zi_put64(p + 24, e->hash);
        p[32] = e->comp_method;
        p += 33;
    }
    const zpack_u64 at = writer->write_offset;
    int rc = zi_writer_put(writer, img, (size_t)total);
    free(img);
    if (rc == ZPACK_OK) writer->cdr_offset = at;
    return rc;
}

int zpack_write_cdr(zpack_writer* writer) { return zpack_write_cdr_ex(writer, writer->file_entries, writer->file_count); }

int zpack_write_eocdr_ex(zpack_writer* writer, zpack_u64 cdr_offset)
{
    zpack_u8 b[ZPACK_EOCDR_SIZE];
    zi_put32(b, ZPACK_EOCDR_SIGNATURE);
    zi_put64(b + 4, cdr_offset);
    const zpack_u64 at = writer->write_offset;
    int rc = zi_writer_put(writer, b, sizeof(b));
    if (rc == ZPACK_OK) writer->eocdr_offset = at;
    return rc;
}

int zpack_write_eocdr(zpack_writer* writer) { return zpack_write_eocdr_ex(writer, writer->cdr_offset); }

/* ------------------------------------------------------------------ the hot path: compress + hash */

typedef struct {
    zi_ctx* ctx; const zpack_u8* const* srcs; const zpk_encode_desc* desc; zpack_u8* const* dsts; zpk_encode_result* res;
    zpack_u64 cut[ZI_MAX_DEVICES + 1]; int rc[ZI_MAX_DEVICES];
} write_job;

static void write_part(void* arg, int k)
{
    write_job* j = (write_job*)arg;
    const zpack_u64 lo = j->cut[k], n = j->cut[k + 1] - lo;
    j->rc[k] = n ? zpk_codec_encode_batch_host(j->ctx->dev[k], j->srcs + lo, j->desc + lo, n, j->dsts + lo, j->res + lo) : ZPK_OK;
}

int zpack_write_files(zpack_writer* writer, zpack_file* files, zpack_u64 file_count)
{
    if (file_count == 0) return ZPACK_OK;
    if (!writer->file && !writer->buffer) return ZPACK_ERROR_WRITER_NOT_OPENED;
    /* contexts: a per-file cctx is honoured when every file names the same one, else the writer's own */
    void* explicit_ctx = files[0].cctx;
    for (zpack_u64 i = 1; i < file_count; i++) if (files[i].cctx != explicit_ctx) explicit_ctx = NULL;
    zi_ctx* ctx = zi_pick_ctx(explicit_ctx, &writer->zstd_cctx);     /* lib/zpack_write.c:20-34: the writer's own, created on first use */
    if (!ctx) return ZPACK_ERROR_NOT_AVAILABLE;              /* no HIP device: there is no CPU fallback */

    zpk_encode_desc* desc = (zpk_encode_desc*)calloc((size_t)file_count, sizeof(*desc));
    zpk_encode_result* res = (zpk_encode_result*)calloc((size_t)file_count, sizeof(*res));
    const zpack_u8** srcs = (const zpack_u8**)calloc((size_t)file_count, sizeof(*srcs));
    zpack_u8** dsts = (zpack_u8**)calloc((size_t)file_count, sizeof(*dsts));
    int rc = (desc && res && srcs && dsts) ? ZPACK_OK : ZPACK_ERROR_MALLOC_FAILED;
    zpack_u64 slots = 0;
    for (zpack_u64 i = 0; i < file_count && rc == ZPACK_OK; i++) {
        const zpack_compression_method m = files[i].options->method;
        if (m != ZPACK_COMPRESSION_NONE && m != ZPACK_COMPRESSION_ZSTD && m != ZPACK_COMPRESSION_LZ4) rc = ZPACK_ERROR_COMP_METHOD_INVALID;
        desc[i].size = files[i].size;
        desc[i].method = (uint32_t)m;
        desc[i].level = files[i].options->level;
        desc[i].dst_capacity = zpk_codec_compress_bound((uint32_t)m, (size_t)files[i].size);     /* lib/zpack_write.c:125-150 */
        slots += desc[i].dst_capacity + 16;
        srcs[i] = files[i].buffer;
    }
    zpack_u8* scratch = NULL;
    if (rc == ZPACK_OK) {
        scratch = (zpack_u8*)malloc((size_t)slots ? (size_t)slots : 1);
        if (!scratch) rc = ZPACK_ERROR_MALLOC_FAILED;
    }
    if (rc == ZPACK_OK) {
        zpack_u64 pos = 0;
        for (zpack_u64 i = 0; i < file_count; i++) { dsts[i] = scratch + pos; pos += desc[i].dst_capacity + 16; }
        write_job job = { ctx, srcs, desc, dsts, res, {0}, {0} };
        int parts = ctx->n;
        if ((zpack_u64)parts > file_count) parts = (int)file_count;
        if (parts > 1) {           /* static shard by source bytes: one host thread + one codec per device */
            zpack_u64* w = (zpack_u64*)malloc(sizeof(zpack_u64) * (size_t)file_count);
            if (!w) parts = 1;
            else { for (zpack_u64 i = 0; i < file_count; i++) w[i] = files[i].size; zi_split(w, file_count, parts, job.cut); free(w); }
        }
        if (parts <= 1) { job.cut[0] = 0; job.cut[1] = file_count; parts = 1; }
        zi_parallel(parts, write_part, &job);
        for (int k = 0; k < parts; k++) if (job.rc[k] != ZPK_OK) rc = ZPACK_ERROR_NOT_AVAILABLE;
    }
    /* append in order; the first failing file stops the call like the reference loop does (:299-303) */
    for (zpack_u64 i = 0; i < file_count && rc == ZPACK_OK; i++) {
        writer->last_return = res[i].status ? (size_t)0 - res[i].detail : (size_t)res[i].comp_size;
        if (res[i].status != ZPACK_OK) { rc = res[i].status; break; }
        const zpack_u64 at = writer->write_offset;
        if ((rc = zi_writer_put(writer, dsts[i], (size_t)res[i].comp_size))) break;
        zpack_file_entry* e = zi_writer_push_entry(writer);
        if (!e || !(e->filename = dup_name(files[i].filename))) { rc = ZPACK_ERROR_MALLOC_FAILED; break; }
        e->offset = at;
        e->comp_size = res[i].comp_size;
        e->uncomp_size = files[i].size;
        e->hash = res[i].hash;                                                  /* XXH3-64 of the source, :256 */
        e->comp_method = (zpack_u8)files[i].options->method;
    }
    free(scratch); free(desc); free(res); free(srcs); free(dsts);
    return rc;
}

/* raw entry copy between archives: no codec involved (lib/zpack_write.c:345-428) */
int zpack_write_files_from_archive(zpack_writer* writer, zpack_reader* reader, zpack_file_entry* entries, zpack_u64 file_count)
{
    if (!writer->file && !writer->buffer) return ZPACK_ERROR_WRITER_NOT_OPENED;
    zpack_u8* tmp = NULL; size_t tmp_cap = 0;
    int rc = ZPACK_OK;
    for (zpack_u64 i = 0; i < file_count && rc == ZPACK_OK; i++) {
        const zpack_file_entry* src = entries + i;
        const zpack_u8* payload = NULL;
        if (reader->file) {
            if (src->comp_size > SIZE_MAX) { rc = ZPACK_ERROR_MALLOC_FAILED; break; }
            if (tmp_cap < src->comp_size) {
                zpack_u8* nb = (zpack_u8*)realloc(tmp, (size_t)src->comp_size);
                if (!nb) { rc = ZPACK_ERROR_MALLOC_FAILED; break; }
                tmp = nb; tmp_cap = (size_t)src->comp_size;
            }
            if ((rc = zpack_read_raw_file(reader, (zpack_file_entry*)src, tmp, tmp_cap))) break;
            payload = tmp;
        } else if (reader->buffer) {
            if (src->offset >= reader->file_size || src->comp_size > reader->file_size - src->offset) { rc = ZPACK_ERROR_FILE_OFFSET_INVALID; break; }
            payload = reader->buffer + src->offset;
        } else { rc = ZPACK_ERROR_ARCHIVE_NOT_LOADED; break; }
        const zpack_u64 at = writer->write_offset;
        if ((rc = zi_writer_put(writer, payload, (size_t)src->comp_size))) break;
        zpack_file_entry* e = zi_writer_push_entry(writer);
        if (!e) { rc = ZPACK_ERROR_MALLOC_FAILED; break; }
        *e = *src;
        e->offset = at;
        if (!(e->filename = dup_name(src->filename))) { rc = ZPACK_ERROR_MALLOC_FAILED; break; }
    }
    free(tmp);
    return rc;
}

/* streaming write (lib/zpack_write.c:461-685): the plaintext goes to the device chunk by chunk and is compressed there as it comes —
 * every call hands what has been compressed so far to the archive through the caller's output window, as the reference does with
 * its library buffers (:541-571); the entry is one frame, its blocks compressed piece by piece (zpk_stream.inc). */
static int stream_drain(zpack_writer* writer, zpack_stream* stream, zi_stream_state* st)
{
    for (;;) {
        size_t n = zpk_cstream_drain(st->c, stream->next_out, stream->avail_out);
        if (n == 0) return ZPACK_OK;
        int rc = zi_writer_put(writer, stream->next_out, n);
        if (rc) return rc;
        stream->total_out += n;
    }
}

int zpack_write_file_stream(zpack_writer* writer, zpack_compress_options* options, zpack_stream* stream, void* cctx)
{
    if (!stream->next_in || !stream->next_out || !stream->avail_out) return ZPACK_ERROR_STREAM_INVALID;
    const zpack_compression_method m = options->method;
    if (m != ZPACK_COMPRESSION_NONE && m != ZPACK_COMPRESSION_ZSTD && m != ZPACK_COMPRESSION_LZ4) return ZPACK_ERROR_COMP_METHOD_INVALID;
    if (!writer->file && !writer->buffer) return ZPACK_ERROR_WRITER_NOT_OPENED;
    zi_stream_state* st = (zi_stream_state*)stream->xxh3_state;
    if (!st) return ZPACK_ERROR_STREAM_INVALID;
    zi_ctx* ctx = zi_pick_ctx(cctx, &writer->zstd_cctx);
    if (!ctx) return ZPACK_ERROR_NOT_AVAILABLE;
    if (!st->c && zpk_cstream_create(ctx->dev[0], &st->c) != ZPK_OK) return ZPACK_ERROR_MALLOC_FAILED;
    zpk_cstream_bind(st->c, ctx->dev[0]);
    if (stream->total_in == 0) {
        zpk_cstream_reset(st->c); st->c_active = 1;
        st->entry_offset = writer->write_offset;                  /* where this entry's first byte goes */
        int rc0 = zpk_cstream_configure(st->c, (uint32_t)m, options->level);
        if (rc0) return rc0;
    }
    int rc = zpk_cstream_update(st->c, stream->next_in, stream->avail_in);
    if (rc) return rc;
    stream->next_in += stream->avail_in;
    stream->total_in += stream->avail_in;
    stream->avail_in = 0;
    return stream_drain(writer, stream, st);
}

int zpack_write_file_stream_end(zpack_writer* writer, char* filename, zpack_compress_options* options, zpack_stream* stream, void* cctx)
{
    if (!stream->next_out || !stream->avail_out) return ZPACK_ERROR_STREAM_INVALID;
    const zpack_compression_method m = options->method;
    if (m != ZPACK_COMPRESSION_NONE && m != ZPACK_COMPRESSION_ZSTD && m != ZPACK_COMPRESSION_LZ4) return ZPACK_ERROR_COMP_METHOD_INVALID;
    if (!writer->file && !writer->buffer) return ZPACK_ERROR_WRITER_NOT_OPENED;
    zi_stream_state* st = (zi_stream_state*)stream->xxh3_state;
    if (!st) return ZPACK_ERROR_STREAM_INVALID;
    zi_ctx* ctx = zi_pick_ctx(cctx, &writer->zstd_cctx);
    if (!ctx) return ZPACK_ERROR_NOT_AVAILABLE;
    if (!st->c && zpk_cstream_create(ctx->dev[0], &st->c) != ZPK_OK) return ZPACK_ERROR_MALLOC_FAILED;   /* an empty entry: end without update */
    zpk_cstream_bind(st->c, ctx->dev[0]);
    if (stream->total_in == 0 && !st->c_active) { zpk_cstream_reset(st->c); st->entry_offset = writer->write_offset; }
    uint64_t csize = 0, usize = 0, hash = 0;
    int rc = zpk_cstream_finish(st->c, (uint32_t)m, options->level, &csize, &usize, &hash);
    if (rc) return rc;
    if ((rc = stream_drain(writer, stream, st))) return rc;
    zpack_file_entry* e = zi_writer_push_entry(writer);
    if (!e || !(e->filename = dup_name(filename))) return ZPACK_ERROR_MALLOC_FAILED;
    e->offset = st->entry_offset;
    e->comp_size = csize;
    e->uncomp_size = usize;
    e->hash = hash;
    e->comp_method = (zpack_u8)m;
    zpk_cstream_reset(st->c);
    st->c_active = 0;
    return ZPACK_OK;
}

int zpack_write_archive(zpack_writer* writer, zpack_file* files, zpack_u64 file_count)
{
    int rc;
    if ((rc = zpack_write_header(writer))) return rc;
    if ((rc = zpack_write_data_header(writer))) return rc;
    if ((rc = zpack_write_files(writer, files, file_count))) return rc;
    if ((rc = zpack_write_cdr(writer))) return rc;
    return zpack_write_eocdr(writer);
}

void zpack_close_writer(zpack_writer* writer)
{
    if (writer->file) fclose(writer->file);
    free(writer->buffer);
    if (writer->file_entries) {
        for (zpack_u64 i = 0; i < writer->file_count; i++) free(writer->file_entries[i].filename);
        free(writer->file_entries);
    }
    zi_ctx_destroy((zi_ctx*)writer->zstd_cctx);
    memset(writer, 0, sizeof(*writer));
}
