/*
 * ref_driver.c — multi-threaded CPU baseline driver around the COMPILED REFERENCE.  TEST INFRASTRUCTURE.
 *
 * Runs the reference's own per-entry read path — zpack_read_file on a memory-shared reader with one
 * decompression context per thread, which lib/zpack.h:337-340 permits — over a bounded sample of an
 * archive, and reports decompressed bytes per second.  Linked against oracle/_ref/libzpack_ref.so.
 * Only bench.py's cpu_baseline leg and tests call this.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "zpack.h"      /* the reference's own header, found via -I/root/reference/lib at build time */

typedef struct {
    zpack_reader* reader;
    uint64_t lo, hi;
    double seconds;         /* keep looping over [lo,hi) until this much time has passed */
    uint64_t bytes, entries, errors;
    double elapsed;
} job_t;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void* run(void* arg)
{
    job_t* j = (job_t*)arg;
    /* every thread reads through its own shallow copy of the (memory-shared) reader: zpack_read_file stores the codec's
     * return value in reader->last_return and the LZ4 path reads it back (lib/zpack_read.c:419-450), so one shared
     * struct makes concurrent readers see each other's values (spurious FILE_INCOMPLETE on mixed archives) */
    zpack_reader local = *j->reader;
    void* zd = zpack_create_dctx(ZPACK_COMPRESSION_ZSTD);
    void* ld = zpack_create_dctx(ZPACK_COMPRESSION_LZ4);
    uint64_t cap = 0;
    for (uint64_t i = j->lo; i < j->hi; i++) if (j->reader->file_entries[i].uncomp_size > cap) cap = j->reader->file_entries[i].uncomp_size;
    uint8_t* buf = (uint8_t*)malloc(cap ? cap : 1);
    double t0 = now();
    do {
        for (uint64_t i = j->lo; i < j->hi; i++) {
            zpack_file_entry* e = &local.file_entries[i];
            void* d = e->comp_method == ZPACK_COMPRESSION_ZSTD ? zd : (e->comp_method == ZPACK_COMPRESSION_LZ4 ? ld : NULL);
            int rc = zpack_read_file(&local, e, buf, cap, d);      /* decode + XXH3 verify, lib/zpack_read.c:326-471 */
            if (rc) j->errors++;
            j->bytes += e->uncomp_size; j->entries++;
        }
    } while (now() - t0 < j->seconds);
    j->elapsed = now() - t0;
    free(buf);
    zpack_free_dctx(ZPACK_COMPRESSION_ZSTD, zd);
    zpack_free_dctx(ZPACK_COMPRESSION_LZ4, ld);
    return NULL;
}

/* decode entries [first, first+count) of the archive with `threads` threads for at least `seconds`;
 * out[0] = decompressed bytes, out[1] = wall seconds, out[2] = errors, out[3] = entries decoded.
 * Returns 0, or the zpack error of opening the archive. */
int ref_baseline_decode(uint8_t* archive, size_t size, uint64_t first, uint64_t count, int threads, double seconds, double* out)
{
    zpack_reader r; memset(&r, 0, sizeof(r));
    int rc = zpack_init_reader_memory_shared(&r, archive, size);
    if (rc) return rc;
    if (first > r.file_count) first = r.file_count;
    if (count > r.file_count - first) count = r.file_count - first;
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > count && count) threads = (int)count;
    job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    double t0 = now();
    for (int t = 0; t < threads; t++) {
        jobs[t].reader = &r; jobs[t].seconds = seconds;
        jobs[t].lo = first + count * (uint64_t)t / (uint64_t)threads;
        jobs[t].hi = first + count * (uint64_t)(t + 1) / (uint64_t)threads;
        pthread_create(&th[t], NULL, run, &jobs[t]);
    }
    double bytes = 0, errors = 0, entries = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); bytes += (double)jobs[t].bytes; errors += (double)jobs[t].errors; entries += (double)jobs[t].entries; }
    out[0] = bytes; out[1] = now() - t0; out[2] = errors; out[3] = entries;
    free(jobs); free(th);
    zpack_close_reader(&r);
    return 0;
}

/* ---- write path: zpack_write_files of the reference (lib/zpack_write.c:280-343: bound -> compress -> XXH3 -> append) ---- */
typedef struct {
    uint8_t* plain; uint64_t entry_size, lo, hi; int method, level; double seconds;
    uint64_t src_bytes, comp_bytes, errors, entries; double elapsed;
} wjob_t;

static void* wrun(void* arg)
{
    wjob_t* j = (wjob_t*)arg;
    zpack_compress_options opt; opt.method = (zpack_compression_method)j->method; opt.level = j->level;
    double t0 = now();
    do {
        zpack_writer w; memset(&w, 0, sizeof(w));
        if (zpack_init_writer_heap(&w, 0)) { j->errors++; break; }
        zpack_write_header(&w); zpack_write_data_header(&w);
        for (uint64_t i = j->lo; i < j->hi; i++) {
            zpack_file f; memset(&f, 0, sizeof(f));
            f.filename = (char*)"e"; f.buffer = j->plain + i * j->entry_size; f.size = j->entry_size; f.options = &opt; f.cctx = NULL;
            size_t before = w.write_offset;
            if (zpack_write_files(&w, &f, 1)) j->errors++;
            j->src_bytes += j->entry_size; j->comp_bytes += w.write_offset - before; j->entries++;
            if (now() - t0 >= j->seconds && j->entries >= 2) break;
        }
        zpack_close_writer(&w);
    } while (now() - t0 < j->seconds);
    j->elapsed = now() - t0;
    return NULL;
}

/* compress entries [0, count) of `plain` (count x entry_size bytes) with `threads` threads for at least `seconds`;
 * out[0] = source bytes compressed, out[1] = wall seconds, out[2] = errors, out[3] = compressed bytes produced */
int ref_baseline_encode(uint8_t* plain, uint64_t entry_size, uint64_t count, int method, int level, int threads, double seconds, double* out)
{
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > count && count) threads = (int)count;
    wjob_t* jobs = (wjob_t*)calloc((size_t)threads, sizeof(wjob_t));
    pthread_t* th = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    double t0 = now();
    for (int t = 0; t < threads; t++) {
        jobs[t].plain = plain; jobs[t].entry_size = entry_size; jobs[t].method = method; jobs[t].level = level; jobs[t].seconds = seconds;
        jobs[t].lo = count * (uint64_t)t / (uint64_t)threads;
        jobs[t].hi = count * (uint64_t)(t + 1) / (uint64_t)threads;
        pthread_create(&th[t], NULL, wrun, &jobs[t]);
    }
    double sb = 0, cb = 0, er = 0;
    for (int t = 0; t < threads; t++) { pthread_join(th[t], NULL); sb += (double)jobs[t].src_bytes; cb += (double)jobs[t].comp_bytes; er += (double)jobs[t].errors; }
    out[0] = sb; out[1] = now() - t0; out[2] = er; out[3] = cb;
    free(jobs); free(th);
    return 0;
}
