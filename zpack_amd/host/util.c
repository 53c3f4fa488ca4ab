/* util.c — codec context plumbing and small helpers for the host library. */
#include "internal.h"
#include <pthread.h>

/* ------------------------------------------------------------------ contexts
 *
 * A context (the `void* dctx / cctx` of zpack.h, and what a reader / writer creates lazily for itself when the
 * caller passes NULL — lib/zpack_read.c:17-31, lib/zpack_write.c:20-34) is a set of device codecs: one, on the
 * default device, unless ZPACK_AMD_DEVICES asks for more ("all", or a comma list of HIP ordinals such as "0,1,2,3";
 * an ordinal may repeat, which gives that device two codecs).  A batch call on a context with several codecs is
 * split into contiguous ranges balanced by bytes, one host thread per codec, no exchange between devices
 * (SURVEY.md §8e).  There is NO process-wide context: like the reference, two readers (or two explicit contexts)
 * share nothing, so they may be used from two threads at once; one context used from two threads is serialised
 * by the codec's own lock (the reference leaves that case undefined). */

static pthread_mutex_t g_lazy = PTHREAD_MUTEX_INITIALIZER;      /* guards the lazy creation inside an owner slot */

static int parse_devices(int* out, int cap)
{
    const char* e = getenv("ZPACK_AMD_DEVICES");
    const int have = zpk_codec_device_count();
    if (have <= 0) return 0;
    if (!e || !*e) { out[0] = -1; return 1; }                     /* -1: ZPACK_AMD_DEVICE / LOCAL_RANK / 0, resolved by the codec */
    int n = 0;
    if (strcmp(e, "all") == 0) {
        for (int d = 0; d < have && n < cap; d++) out[n++] = d;
        return n;
    }
    while (*e && n < cap) {
        char* end = NULL;
        long v = strtol(e, &end, 10);
        if (end == e) break;
        if (v >= 0 && v < have) out[n++] = (int)v;
        e = *end == ',' ? end + 1 : end;
        if (*end && *end != ',') break;
    }
    if (n == 0) { out[0] = -1; n = 1; }
    return n;
}

zi_ctx* zi_ctx_create(void)
{
    int devs[ZI_MAX_DEVICES];
    const int n = parse_devices(devs, ZI_MAX_DEVICES);
    if (n <= 0) return NULL;
    zi_ctx* x = (zi_ctx*)calloc(1, sizeof(*x));
    if (!x) return NULL;
    for (int i = 0; i < n; i++) {
        if (zpk_codec_create(&x->dev[x->n], devs[i]) == ZPK_OK) {
            /* zpack.h has no place for codec options: a deployment sets them through the environment (bytes; 0 = never).  Entries of at
             * least ZPACK_AMD_ENC_SPLIT_MIN bytes are written as sequences of 512 KiB frames (other bytes than one frame, same
             * plaintext; default 2 MiB), entries of at least ZPACK_AMD_DEC_SPLIT_MIN that are such sequences are read one wave per frame. */
            static const struct { const char* env; int opt; } knobs[] = {
                { "ZPACK_AMD_ENC_SPLIT_MIN", ZPK_OPT_ENC_SPLIT_MIN }, { "ZPACK_AMD_DEC_SPLIT_MIN", ZPK_OPT_DEC_SPLIT_MIN }, { "ZPACK_AMD_ORDER_MIN", ZPK_OPT_ORDER_MIN } };
            for (size_t k = 0; k < sizeof(knobs) / sizeof(knobs[0]); k++) {
                const char* v = getenv(knobs[k].env);
                if (v && *v) { char* end = NULL; long long q = strtoll(v, &end, 10); if (end && *end == 0 && q >= 0 && q <= 0x7FFFFFFF) (void)zpk_codec_set_option(x->dev[x->n], knobs[k].opt, (int)q); }
            }
            x->n++;
        }
    }
    if (x->n == 0) { free(x); return NULL; }
    return x;
}

void zi_ctx_destroy(zi_ctx* x)
{
    if (!x) return;
    for (int i = 0; i < x->n; i++) zpk_codec_destroy(x->dev[i]);
    free(x);
}

void zi_ctx_reset(zi_ctx* x)
{
    if (!x) return;
    for (int i = 0; i < x->n; i++) zpk_codec_reset(x->dev[i]);
}

zi_ctx* zi_pick_ctx(void* explicit_ctx, void** owner_slot)
{
    if (explicit_ctx) return (zi_ctx*)explicit_ctx;
    if (!owner_slot) return NULL;
    pthread_mutex_lock(&g_lazy);
    if (!*owner_slot) *owner_slot = zi_ctx_create();              /* NULL when no HIP device is usable: there is no CPU fallback */
    zi_ctx* x = (zi_ctx*)*owner_slot;
    pthread_mutex_unlock(&g_lazy);
    return x;
}

/* split `count` items with weights w[] into x->n contiguous ranges of about equal weight: cut[k] .. cut[k+1] */
void zi_split(const zpack_u64* w, zpack_u64 count, int parts, zpack_u64* cut)
{
    zpack_u64 total = 0;
    for (zpack_u64 i = 0; i < count; i++) total += w[i] + 1;
    zpack_u64 acc = 0, i = 0;
    cut[0] = 0;
    for (int k = 1; k < parts; k++) {
        const zpack_u64 goal = total / (zpack_u64)parts * (zpack_u64)k;
        while (i < count && acc + (w[i] + 1) / 2 < goal) { acc += w[i] + 1; i++; }
        cut[k] = i;
    }
    cut[parts] = count;
}

typedef struct { void (*fn)(void*, int); void* arg; int part; } zi_job;
static void* job_main(void* p) { zi_job* j = (zi_job*)p; j->fn(j->arg, j->part); return NULL; }

/* run fn(arg, k) for k in [0, parts): parts-1 helper threads + the caller */
void zi_parallel(int parts, void (*fn)(void*, int), void* arg)
{
    pthread_t th[ZI_MAX_DEVICES];
    zi_job jobs[ZI_MAX_DEVICES];
    int started[ZI_MAX_DEVICES] = {0};
    for (int k = 1; k < parts; k++) {
        jobs[k].fn = fn; jobs[k].arg = arg; jobs[k].part = k;
        started[k] = pthread_create(&th[k], NULL, job_main, &jobs[k]) == 0;
    }
    fn(arg, 0);
    for (int k = 1; k < parts; k++) {
        if (started[k]) pthread_join(th[k], NULL);
        else fn(arg, k);                                          /* no thread to be had: do it here */
    }
}

/* contexts are opaque; one per thread for concurrent buffer-backed reads (the threading contract of
 * lib/zpack.h:337-340).  Stands where the reference hands out ZSTD_DCtx / LZ4F_dctx / ZSTD_CCtx / LZ4F_cctx objects
 * (lib/zpack_read.c:776-812, zpack_write.c:899-935).  One context serves every method, also ZPACK_COMPRESSION_NONE
 * (where the reference returns NULL: stored entries need no library context there, but they do run on the device here). */
static void* make_ctx(zpack_compression_method method)
{
    if (method != ZPACK_COMPRESSION_NONE && method != ZPACK_COMPRESSION_ZSTD && method != ZPACK_COMPRESSION_LZ4) return NULL;
    return zi_ctx_create();
}

void* zpack_create_dctx(zpack_compression_method method) { return make_ctx(method); }
void* zpack_create_cctx(zpack_compression_method method) { return make_ctx(method); }
void zpack_free_dctx(zpack_compression_method method, void* dctx) { (void)method; zi_ctx_destroy((zi_ctx*)dctx); }
void zpack_free_cctx(zpack_compression_method method, void* cctx) { (void)method; zi_ctx_destroy((zi_ctx*)cctx); }

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

/* lookup by name (lib/zpack_read.c:760-769: a linear strcmp scan, first match wins) — SURVEY.md §8f rank 1: at 1 M
 * entries the scan is the visible cost of every by-name access.  Tables OWNED BY A READER (reader_load_entries: names in
 * one arena, never touched by the caller — the reference frees those strings itself, so they are read-only by contract
 * there too) are registered here; the first lookup in a registered table builds an open-addressing index over FNV-1a
 * hashes of the names (first occurrence of a name wins, like the scan), later lookups are O(1).  Any other table —
 * one the caller built or parsed through the public zpack_read_cdr* calls — is scanned exactly like the reference. */
typedef struct zi_index_s { const zpack_file_entry* table; zpack_u64 count; zpack_u32* slot; zpack_u64 mask; struct zi_index_s* next; } zi_index;
static pthread_mutex_t g_idx_mu = PTHREAD_MUTEX_INITIALIZER;
static zi_index* g_idx = NULL;

static zpack_u64 name_hash(const char* s)
{
    zpack_u64 h = 1469598103934665603ull;
    for (; *s; s++) { h ^= (zpack_u8)*s; h *= 1099511628211ull; }
    return h;
}

void zi_index_register(const zpack_file_entry* table, zpack_u64 count)
{
    if (count < 64 || count >= 0xFFFFFFFFull) return;               /* small tables: the scan is as fast */
    zi_index* x = (zi_index*)calloc(1, sizeof(*x));
    if (!x) return;
    x->table = table; x->count = count;
    pthread_mutex_lock(&g_idx_mu);
    x->next = g_idx; g_idx = x;
    pthread_mutex_unlock(&g_idx_mu);
}

void zi_index_drop(const zpack_file_entry* table)
{
    pthread_mutex_lock(&g_idx_mu);
    for (zi_index** pp = &g_idx; *pp; pp = &(*pp)->next)
        if ((*pp)->table == table) { zi_index* x = *pp; *pp = x->next; free(x->slot); free(x); break; }
    pthread_mutex_unlock(&g_idx_mu);
}

zpack_file_entry* zpack_get_file_entry(const char* filename, zpack_file_entry* file_entries, zpack_u64 file_count)
{
    if (file_count >= 64) {
        zpack_file_entry* hit = NULL;
        int answered = 0;
        pthread_mutex_lock(&g_idx_mu);
        zi_index* x = g_idx;
        while (x && !(x->table == file_entries && x->count == file_count)) x = x->next;
        if (x && !x->slot) {
            zpack_u64 cap = 1; while (cap < file_count * 2) cap <<= 1;
            x->slot = (zpack_u32*)malloc(sizeof(zpack_u32) * (size_t)cap);
            if (x->slot) {
                memset(x->slot, 0xFF, sizeof(zpack_u32) * (size_t)cap);
                x->mask = cap - 1;
                for (zpack_u64 i = 0; i < file_count; i++) {
                    zpack_u64 p = name_hash(file_entries[i].filename) & x->mask;
                    for (;;) {
                        const zpack_u32 s = x->slot[p];
                        if (s == 0xFFFFFFFFu) { x->slot[p] = (zpack_u32)i; break; }
                        if (strcmp(file_entries[s].filename, file_entries[i].filename) == 0) break;   /* first occurrence wins */
                        p = (p + 1) & x->mask;
                    }
                }
            }
        }
        if (x && x->slot) {
            zpack_u64 p = name_hash(filename) & x->mask;
            answered = 1;
            for (;;) {
                const zpack_u32 s = x->slot[p];
                if (s == 0xFFFFFFFFu) break;
                if (strcmp(file_entries[s].filename, filename) == 0) { hit = file_entries + s; break; }
                p = (p + 1) & x->mask;
            }
        }
        pthread_mutex_unlock(&g_idx_mu);
        if (answered) return hit;
    }
    for (zpack_u64 i = 0; i < file_count; i++)
        if (strcmp(file_entries[i].filename, filename) == 0) return file_entries + i;
    return NULL;
}

zpack_bool zpack_read_stream_done(zpack_stream* stream, zpack_file_entry* entry)
{
    return ZPACK_READ_STREAM_DONE(stream, entry);
}
