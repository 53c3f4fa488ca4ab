/*
 * datagen.c — deterministic synthetic .zpk archives for tests and bench.py (NOT product code, NOT
 * the oracle).
 *
 * Builds, in host memory, a complete ZPack archive (docs/specs.md of the reference: header,
 * data signature, entry payloads back to back, CDR, EOCDR) whose entries are compressed with the
 * REAL lz4 / zstd libraries of this image using the same call sequence as the reference writer
 * (lib/zpack_write.c:179 ZSTD_compressCCtx; :199-211 LZ4F_compressBegin/Update/End with zeroed
 * preferences + level) and whose entry hashes come from the real xxHash header (XXH3_64bits,
 * lib/zpack_write.c:256).  So the frames are exactly what a .zpk written by the reference holds,
 * and the expected hashes are independent of both the product and the oracle.
 *
 * Corpus classes (Silesia is not available offline, this seeded mixture replaces it):
 *   0 text    Zipf(s=1.1) over 8192 random lowercase words of 2-12 letters, newline every 12 words
 *   1 records 32-byte records {u32 counter, 2 low-cardinality bytes, u32 slow timestamp, 8 random, 14 zero}
 *   2 random  uniform bytes (stored LZ4 blocks / raw zstd blocks)
 *   3 runs    byte runs, geometric length, mean 200 (overlapping matches, RLE blocks)
 * PRNG: SplitMix64-seeded xoshiro256**.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

#include <lz4frame.h>
#include <zstd.h>
#define XXH_INLINE_ALL
#include <xxhash.h>

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix(uint64_t* x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t* r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r)
{
    uint64_t* s = r->s;
    uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static inline double rng_unit(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* ---- text dictionary + Walker alias table for Zipf(1.1) over 8192 ranks (built once) ---- */
#define NWORDS 8192
static char     g_words[NWORDS][13];
static uint8_t  g_wlen[NWORDS];
static double   g_alias_p[NWORDS];
static uint32_t g_alias_i[NWORDS];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static void build_dictionary(void)
{
    rng_t r; rng_seed(&r, 0x5A50414B44494354ULL);   /* fixed: the dictionary is part of the corpus definition */
    for (int w = 0; w < NWORDS; w++) {
        int len = 2 + (int)(rng_next(&r) % 11);
        for (int i = 0; i < len; i++) g_words[w][i] = (char)('a' + rng_next(&r) % 26);
        g_wlen[w] = (uint8_t)len;
    }
    static double p[NWORDS];
    double sum = 0;
    for (int k = 0; k < NWORDS; k++) { p[k] = 1.0 / pow((double)(k + 1), 1.1); sum += p[k]; }
    static uint32_t small[NWORDS], large[NWORDS];
    int ns = 0, nl = 0;
    for (int k = 0; k < NWORDS; k++) {
        p[k] = p[k] / sum * NWORDS;
        if (p[k] < 1.0) small[ns++] = (uint32_t)k; else large[nl++] = (uint32_t)k;
    }
    while (ns && nl) {
        uint32_t s = small[--ns], l = large[--nl];
        g_alias_p[s] = p[s]; g_alias_i[s] = l;
        p[l] = (p[l] + p[s]) - 1.0;
        if (p[l] < 1.0) small[ns++] = l; else large[nl++] = l;
    }
    while (nl) { uint32_t l = large[--nl]; g_alias_p[l] = 1.0; g_alias_i[l] = l; }
    while (ns) { uint32_t s = small[--ns]; g_alias_p[s] = 1.0; g_alias_i[s] = s; }
}

static void fill_text(rng_t* r, uint8_t* dst, uint64_t n)
{
    uint64_t op = 0; int col = 0;
    while (op < n) {
        uint64_t u = rng_next(r);
        uint32_t k = (uint32_t)(u & (NWORDS - 1));
        double f = (double)(u >> 11) * (1.0 / 9007199254740992.0);
        if (f >= g_alias_p[k]) k = g_alias_i[k];
        unsigned len = g_wlen[k];
        for (unsigned i = 0; i < len && op < n; i++) dst[op++] = (uint8_t)g_words[k][i];
        if (op < n) dst[op++] = (++col == 12) ? (col = 0, '\n') : ' ';
    }
}

static void fill_records(rng_t* r, uint8_t* dst, uint64_t n)
{
    uint32_t counter = (uint32_t)rng_next(r), ts = (uint32_t)rng_next(r);
    uint8_t rec[32];
    uint64_t op = 0;
    while (op < n) {
        memset(rec, 0, sizeof(rec));
        memcpy(rec, &counter, 4); counter++;
        uint64_t u = rng_next(r);
        rec[4] = (uint8_t)(u & 3); rec[5] = (uint8_t)((u >> 8) % 7);
        ts += (uint32_t)((u >> 16) & 15);
        memcpy(rec + 6, &ts, 4);
        uint64_t v = rng_next(r);
        memcpy(rec + 10, &v, 8);
        uint64_t k = n - op < 32 ? n - op : 32;
        memcpy(dst + op, rec, k); op += k;
    }
}

static void fill_random(rng_t* r, uint8_t* dst, uint64_t n)
{
    uint64_t op = 0;
    while (op + 8 <= n) { uint64_t v = rng_next(r); memcpy(dst + op, &v, 8); op += 8; }
    if (op < n) { uint64_t v = rng_next(r); memcpy(dst + op, &v, n - op); }
}

static void fill_runs(rng_t* r, uint8_t* dst, uint64_t n)
{
    uint64_t op = 0;
    while (op < n) {
        double u = rng_unit(r);
        uint64_t len = 1 + (uint64_t)(-log(1.0 - u) * 200.0);
        uint8_t b = (uint8_t)rng_next(r);
        if (len > n - op) len = n - op;
        memset(dst + op, b, len); op += len;
    }
}

void zpkgen_fill(int cls, uint64_t seed, uint64_t index, uint8_t* dst, uint64_t n)
{
    pthread_once(&g_once, build_dictionary);
    rng_t r; rng_seed(&r, seed * 0x9E3779B97F4A7C15ULL + index * 0xD1B54A32D192ED03ULL + (uint64_t)cls);
    switch (cls) {
    case 0: fill_text(&r, dst, n); break;
    case 1: fill_records(&r, dst, n); break;
    case 2: fill_random(&r, dst, n); break;
    default: fill_runs(&r, dst, n); break;
    }
}

/* ------------------------------------------------------------------ batch -> archive image */

typedef struct zpkgen_batch_s {
    uint64_t  n;
    uint64_t  archive_size;       /* bytes of the complete .zpk image */
    uint8_t*  archive;            /* header | data | CDR | EOCDR */
    uint64_t* offsets;            /* entry payload offsets inside archive */
    uint64_t* comp_sizes;
    uint64_t* uncomp_sizes;
    uint64_t* hashes;             /* XXH3_64bits of the plaintext, by the real xxHash */
    uint8_t*  methods;            /* 0 none, 1 zstd, 2 lz4 */
    uint8_t*  classes;
    uint64_t  total_comp, total_uncomp;
    uint64_t  cdr_offset;
    int       error;
} zpkgen_batch;

typedef struct {
    zpkgen_batch* b;
    uint64_t lo, hi;
    uint64_t seed; int method, level, mix;
    uint64_t size_lo, size_hi;
    uint8_t* arena; uint64_t arena_cap, arena_used;
    uint64_t* local_off;
    uint64_t first;                 /* global index of the slice's entry 0: seeds, sizes and classes are functions of the GLOBAL index */
    int error;
} worker_t;

static uint64_t entry_size(uint64_t seed, uint64_t i, uint64_t lo, uint64_t hi)
{
    if (lo == hi) return lo;
    uint64_t x = seed ^ (i * 0xA24BAED4963EE407ULL);
    double u = (double)(splitmix(&x) >> 11) * (1.0 / 9007199254740992.0);
    double s = exp(log((double)lo) + u * (log((double)hi) - log((double)lo)));     /* log-uniform */
    uint64_t v = (uint64_t)s;
    return v < lo ? lo : (v > hi ? hi : v);
}

static int entry_class(uint64_t seed, uint64_t i, int mix)
{
    if (mix >= 0) return mix;
    uint64_t x = seed ^ (i * 0x9FB21C651E98DF25ULL);
    unsigned p = (unsigned)(splitmix(&x) % 100);
    return p < 70 ? 0 : (p < 90 ? 1 : (p < 95 ? 2 : 3));                           /* 70/20/5/5 */
}

static int entry_method(uint64_t seed, uint64_t i, int method)
{
    if (method >= 0) return method;
    uint64_t x = seed ^ (i * 0xC2B2AE3D27D4EB4FULL);
    return (splitmix(&x) & 1) ? 2 : 1;                                              /* 50/50 lz4 / zstd */
}

static void* worker(void* arg)
{
    worker_t* w = (worker_t*)arg;
    zpkgen_batch* b = w->b;
    uint8_t* plain = (uint8_t*)malloc(w->size_hi ? w->size_hi : 1);
    ZSTD_CCtx* zc = ZSTD_createCCtx();
    LZ4F_cctx* lc = NULL;
    LZ4F_createCompressionContext(&lc, LZ4F_VERSION);
    if (!plain || !zc || !lc) { w->error = 1; return NULL; }
    for (uint64_t i = w->lo; i < w->hi; i++) {
        const uint64_t gi = w->first + i;
        uint64_t n = entry_size(w->seed, gi, w->size_lo, w->size_hi);
        int cls = entry_class(w->seed, gi, w->mix);
        int method = entry_method(w->seed, gi, w->method);
        zpkgen_fill(cls, w->seed, gi, plain, n);
        size_t bound = method == 1 ? ZSTD_COMPRESSBOUND(n) : (method == 2 ? LZ4F_compressBound(n, NULL) : n);
        if (w->arena_used + bound + 64 > w->arena_cap) {
            uint64_t nc = (w->arena_cap + bound + 64) * 3 / 2;
            uint8_t* na = (uint8_t*)realloc(w->arena, nc);
            if (!na) { w->error = 1; break; }
            w->arena = na; w->arena_cap = nc;
        }
        uint8_t* dst = w->arena + w->arena_used;
        size_t c = 0;
        if (method == 0) { memcpy(dst, plain, n); c = n; }
        else if (method == 1) {
            c = ZSTD_compressCCtx(zc, dst, bound, plain, n, w->level);                /* zpack_write.c:179 */
            if (ZSTD_isError(c)) { w->error = 2; break; }
        } else {
            LZ4F_preferences_t prefs; memset(&prefs, 0, sizeof(prefs));               /* zpack_write.c:199-201 */
            prefs.compressionLevel = w->level;
            size_t r = LZ4F_compressBegin(lc, dst, bound, &prefs);
            if (LZ4F_isError(r)) { w->error = 3; break; }
            c = r;
            r = LZ4F_compressUpdate(lc, dst + c, bound - c, plain, n, NULL);
            if (LZ4F_isError(r)) { w->error = 3; break; }
            c += r;
            r = LZ4F_compressEnd(lc, dst + c, bound - c, NULL);
            if (LZ4F_isError(r)) { w->error = 3; break; }
            c += r;
        }
        w->local_off[i - w->lo] = w->arena_used;
        w->arena_used += c;
        b->comp_sizes[i] = c;
        b->uncomp_sizes[i] = n;
        b->hashes[i] = XXH3_64bits(plain, n);                                          /* zpack_write.c:256 */
        b->methods[i] = (uint8_t)method;
        b->classes[i] = (uint8_t)cls;
    }
    free(plain);
    ZSTD_freeCCtx(zc);
    LZ4F_freeCompressionContext(lc);
    return NULL;
}

static void wr16(uint8_t* p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void wr32(uint8_t* p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static void wr64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }

void zpkgen_free(zpkgen_batch* b)
{
    if (!b) return;
    free(b->archive); free(b->offsets); free(b->comp_sizes); free(b->uncomp_sizes);
    free(b->hashes); free(b->methods); free(b->classes); free(b);
}

/*
 * n entries; sizes log-uniform in [size_lo, size_hi] (equal => fixed); method 0/1/2 or -1 for a
 * seeded 50/50 lz4/zstd coin; level = compression level handed to the library; mix = class 0..3 or
 * -1 for 70/20/5/5 text/records/random/runs; threads = worker threads.
 */
zpkgen_batch* zpkgen_make_range(uint64_t first, uint64_t n, uint64_t size_lo, uint64_t size_hi, int method, int level,
                                uint64_t seed, int mix, int threads);
zpkgen_batch* zpkgen_make(uint64_t n, uint64_t size_lo, uint64_t size_hi, int method, int level,
                          uint64_t seed, int mix, int threads)
{
    return zpkgen_make_range(0, n, size_lo, size_hi, method, level, seed, mix, threads);
}

/* uncompressed sizes of entries [0, n) of the archive (seed, size range): what a rank needs to pick its byte-balanced slice
 * WITHOUT building anything (sizes are a function of the seed and the global index) */
void zpkgen_sizes(uint64_t n, uint64_t size_lo, uint64_t size_hi, uint64_t seed, uint64_t* out)
{
    for (uint64_t i = 0; i < n; i++) out[i] = entry_size(seed, i, size_lo, size_hi);
}

/* Entries [first, first + n) of that archive as a self-contained .zpk image (its own header, data section and CDR; offsets relative to
 * THIS image): the slice one rank of a static shard holds.  Entry k of the slice is entry first + k of the whole archive, byte for byte. */
zpkgen_batch* zpkgen_make_range(uint64_t first, uint64_t n, uint64_t size_lo, uint64_t size_hi, int method, int level,
                                uint64_t seed, int mix, int threads)
{
    pthread_once(&g_once, build_dictionary);
    zpkgen_batch* b = (zpkgen_batch*)calloc(1, sizeof(*b));
    if (!b) return NULL;
    b->n = n;
    uint64_t na = n ? n : 1;
    b->offsets = (uint64_t*)calloc(na, 8); b->comp_sizes = (uint64_t*)calloc(na, 8);
    b->uncomp_sizes = (uint64_t*)calloc(na, 8); b->hashes = (uint64_t*)calloc(na, 8);
    b->methods = (uint8_t*)calloc(na, 1); b->classes = (uint8_t*)calloc(na, 1);
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > na) threads = (int)na;
    worker_t* ws = (worker_t*)calloc((size_t)threads, sizeof(worker_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        worker_t* w = &ws[t];
        w->b = b; w->lo = n * (uint64_t)t / (uint64_t)threads; w->hi = n * (uint64_t)(t + 1) / (uint64_t)threads;
        w->seed = seed; w->method = method; w->level = level; w->mix = mix; w->first = first;
        w->size_lo = size_lo; w->size_hi = size_hi;
        w->arena_cap = (w->hi - w->lo) * (size_lo + size_hi) / 4 + (1u << 20);
        w->arena = (uint8_t*)malloc(w->arena_cap);
        w->local_off = (uint64_t*)calloc(w->hi - w->lo + 1, 8);
        pthread_create(&th[t], NULL, worker, w);
    }
    uint64_t data_bytes = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(th[t], NULL);
        if (ws[t].error) b->error = ws[t].error;
        data_bytes += ws[t].arena_used;
    }
    /* CDR size: 20-byte header + per entry 2 + 8 ("e%07llu") + 33 */
    uint64_t name_len = 8;
    uint64_t cdr_body = n * (2 + name_len + 33);
    b->cdr_offset = 10 + data_bytes;
    b->archive_size = 10 + data_bytes + 20 + cdr_body + 12;
    b->archive = (uint8_t*)malloc(b->archive_size);
    if (!b->archive) { b->error = 1; return b; }
    uint8_t* a = b->archive;
    wr32(a, 0x154b505a); wr16(a + 4, 1); wr32(a + 6, 0x144b505a);
    uint64_t pos = 10;
    for (int t = 0; t < threads; t++) {
        worker_t* w = &ws[t];
        memcpy(a + pos, w->arena, w->arena_used);
        for (uint64_t i = w->lo; i < w->hi; i++) b->offsets[i] = pos + w->local_off[i - w->lo];
        pos += w->arena_used;
        free(w->arena); free(w->local_off);
    }
    uint8_t* c = a + pos;
    wr32(c, 0x134b505a); wr64(c + 4, n); wr64(c + 12, cdr_body);
    c += 20;
    for (uint64_t i = 0; i < n; i++) {
        char name[16];
        snprintf(name, sizeof(name), "e%07llu", (unsigned long long)((first + i) % 10000000ULL));
        wr16(c, (uint16_t)name_len); memcpy(c + 2, name, name_len); c += 2 + name_len;
        wr64(c, b->offsets[i]); wr64(c + 8, b->comp_sizes[i]); wr64(c + 16, b->uncomp_sizes[i]);
        wr64(c + 24, b->hashes[i]); c[32] = b->methods[i]; c += 33;
        b->total_comp += b->comp_sizes[i]; b->total_uncomp += b->uncomp_sizes[i];
    }
    wr32(c, 0x124b505a); wr64(c + 4, b->cdr_offset);
    free(ws); free(th);
    return b;
}

/* one-entry helpers (tests): compress `src` exactly as the reference writer would */
size_t zpkgen_compress(int method, int level, const uint8_t* src, size_t n, uint8_t* dst, size_t cap)
{
    if (method == 0) { if (cap < n) return 0; memcpy(dst, src, n); return n; }
    if (method == 1) { size_t c = ZSTD_compress(dst, cap, src, n, level); return ZSTD_isError(c) ? 0 : c; }
    LZ4F_preferences_t prefs; memset(&prefs, 0, sizeof(prefs));
    prefs.compressionLevel = level;
    LZ4F_cctx* lc = NULL;
    if (LZ4F_isError(LZ4F_createCompressionContext(&lc, LZ4F_VERSION))) return 0;
    size_t c = 0, r = LZ4F_compressBegin(lc, dst, cap, &prefs);
    if (!LZ4F_isError(r)) { c = r; r = LZ4F_compressUpdate(lc, dst + c, cap - c, src, n, NULL); }
    if (!LZ4F_isError(r)) { c += r; r = LZ4F_compressEnd(lc, dst + c, cap - c, NULL); }
    LZ4F_freeCompressionContext(lc);
    return LZ4F_isError(r) ? 0 : c + r;
}

size_t zpkgen_bound(int method, size_t n)
{
    return method == 1 ? ZSTD_COMPRESSBOUND(n) : (method == 2 ? LZ4F_compressBound(n, NULL) + 32 : n);
}

uint64_t zpkgen_xxh3(const uint8_t* p, size_t n) { return XXH3_64bits(p, n); }
