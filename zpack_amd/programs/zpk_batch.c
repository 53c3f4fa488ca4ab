/* zpk_batch.c — batch-aware `t` (test) and `x` (extract) over the additive batch read API.
 *
 * The reference's CLI tests and extracts one entry at a time through the streaming reader
 * (programs/commands.c:706-773 command_test, :413-487 extract_files_i / :330-409 extract_file).  Here the same
 * two commands hand the codec whole batches — zpack_read_files_packed, up to ZPK_BATCH_BYTES of output per call —
 * and print what the reference prints: the same lines, in the same entry order, with the same verdict per entry
 * (a hash mismatch is "corrupted" and the run goes on; for `t` any other error ends the run with status 1 at that
 * entry, for `x` it counts as an error and the run goes on).
 *
 *   zpk-batch t <archive.zpk>
 *   zpk-batch x <archive.zpk> [-o <output dir>] [-j]      (-j: junk paths, like the reference's `e`)
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include "zpack.h"

#ifndef ZPK_BATCH_BYTES
#define ZPK_BATCH_BYTES ((zpack_u64)1 << 30)
#endif

static int mkdir_p_for(char* path)
{
    for (char* p = path + 1; *p; p++)
        if (*p == '/') {
            *p = 0;
            if (mkdir(path, 0777) != 0 && errno != EEXIST) { *p = '/'; return 0; }
            *p = '/';
        }
    return 1;
}

/* a path that stays below the output directory: no leading separators, no "." / ".." components */
static void safe_path(const char* in, char* out)
{
    char* w = out;
    while (*in) {
        while (*in == '/' || *in == '\\') in++;
        const char* e = in;
        while (*e && *e != '/' && *e != '\\') e++;
        const size_t n = (size_t)(e - in);
        if (n && !(n == 1 && in[0] == '.') && !(n == 2 && in[0] == '.' && in[1] == '.')) {
            if (w != out) *w++ = '/';
            memcpy(w, in, n); w += n;
        }
        in = e;
    }
    *w = 0;
}

static const char* base_name(const char* s)
{
    const char* b = s;
    for (; *s; s++) if (*s == '/' || *s == '\\') b = s + 1;
    return b;
}

int main(int argc, char** argv)
{
    if (argc < 3 || (strcmp(argv[1], "t") && strcmp(argv[1], "x"))) {
        fprintf(stderr, "usage: %s t <archive>\n       %s x <archive> [-o dir] [-j]\n", argv[0], argv[0]);
        return 2;
    }
    const int extract = argv[1][0] == 'x';
    const char* archive = argv[2];
    const char* outdir = NULL;
    int junk = 0;
    for (int i = 3; i < argc; i++) {
        if (!strcmp(argv[i], "-o") && i + 1 < argc) outdir = argv[++i];
        else if (!strcmp(argv[i], "-j")) junk = 1;
    }
    printf("-- Reading archive: %s\n", archive);
    zpack_reader reader;
    memset(&reader, 0, sizeof(reader));
    int ret = zpack_init_reader(&reader, archive);
    if (ret) {
        printf("Error: Failed to open \"%s\" for reading (error %d)\n", archive, ret);
        zpack_close_reader(&reader);
        return 1;
    }
    printf("-- Found %" PRIu64 " files\n", reader.file_count);
    printf(extract ? "-- Extracting files...\n" : "-- Testing files...\n");

    const zpack_u64 n = reader.file_count;
    zpack_file_entry** ptrs = (zpack_file_entry**)malloc(sizeof(*ptrs) * (size_t)(n ? n : 1));
    zpack_u64* offs = (zpack_u64*)malloc(sizeof(*offs) * (size_t)(n ? n : 1));
    int* results = (int*)malloc(sizeof(*results) * (size_t)(n ? n : 1));
    zpack_u8* buf = NULL;
    size_t buf_cap = 0;
    zpack_u64 corrupt = 0;
    int errors = 0, rc = 0;
    for (zpack_u64 first = 0; first < n && rc == 0;) {
        /* one device batch: consecutive entries up to ZPK_BATCH_BYTES of output (an entry larger than that goes alone) */
        zpack_u64 cnt = 0, bytes = 0;
        while (first + cnt < n) {
            const zpack_u64 u = reader.file_entries[first + cnt].uncomp_size;          /* untrusted: no wrap of the sum, no malloc beyond size_t */
            if (cnt && (u > ZPK_BATCH_BYTES || bytes + u > ZPK_BATCH_BYTES)) break;
            if (u > (zpack_u64)(SIZE_MAX / 2)) { printf("Error: \"%s\" claims %" PRIu64 " bytes\n", reader.file_entries[first + cnt].filename, u); rc = 1; break; }
            ptrs[cnt] = reader.file_entries + first + cnt;
            bytes += u; cnt++;
        }
        if (rc) break;
        if (bytes > buf_cap) {
            free(buf);
            buf = (zpack_u8*)malloc((size_t)bytes);
            buf_cap = buf ? (size_t)bytes : 0;
            if (!buf) { printf("Error: out of memory for a %" PRIu64 "-byte batch\n", bytes); rc = 1; break; }
        }
        ret = zpack_read_files_packed(&reader, ptrs, cnt, buf, (size_t)bytes, offs, results, NULL);
        if (ret) { printf("Error: Failed to decompress the batch at \"%s\" (error %d)\n", ptrs[0]->filename, ret); rc = 1; break; }
        for (zpack_u64 k = 0; k < cnt; k++) {
            const zpack_file_entry* e = ptrs[k];
            printf("  %s\n", e->filename);
            if (!extract) {
                if (results[k] == ZPACK_ERROR_FILE_HASH_MISMATCH) { printf("-- File is corrupted!\n"); corrupt++; }
                else if (results[k]) {
                    printf("Error: Failed to decompress \"%s\" (error %d)\n", e->filename, results[k]);
                    rc = 1;
                    break;
                }
                continue;
            }
            if (results[k] == ZPACK_ERROR_FILE_HASH_MISMATCH) printf("Warning: File is corrupted (file hash mismatch)\n");
            else if (results[k]) { printf("Error: Failed to extract \"%s\" (error %d)\n", e->filename, results[k]); errors++; continue; }
            const size_t ol = outdir ? strlen(outdir) : 0, fl = strlen(e->filename);
            char* path = (char*)malloc(ol + fl + 3);
            char* name = (char*)malloc(fl + 1);
            if (!path || !name) { printf("Error: out of memory\n"); free(path); free(name); errors++; continue; }
            if (junk) strcpy(name, base_name(e->filename)); else safe_path(e->filename, name);
            if (outdir) { memcpy(path, outdir, ol); path[ol] = '/'; strcpy(path + ol + 1, name); } else strcpy(path, name);
            FILE* fp = NULL;
            int failed = 0;
            if (!*name || !mkdir_p_for(path)) { printf("Error: Failed to create output directory for \"%s\" %s\n", path, strerror(errno)); failed = 1; }
            else if (!(fp = fopen(path, "wb"))) { printf("Failed to open \"%s\" for writing\n", path); failed = 1; }
            else if (e->uncomp_size && fwrite(buf + offs[k], 1, (size_t)e->uncomp_size, fp) != e->uncomp_size) { printf("Error: Failed to write data to \"%s\"\n", path); failed = 1; }
            if (fp && fclose(fp) != 0 && !failed) { printf("Error: Failed to write data to \"%s\"\n", path); failed = 1; }
            errors += failed;
            free(path); free(name);
        }
        first += cnt;
    }
    if (rc == 0) {
        /* (the reference's extract counts failed files and still exits 0: programs/commands.c:472-487) */
        if (extract) { if (errors) printf("-- Errors: %d\n", errors); printf("-- Done.\n"); }
        else printf("-- Done.\n-- Corrupted files: %" PRIu64 "/%" PRIu64 "\n", corrupt, n);
    }
    free(buf); free(ptrs); free(offs); free(results);
    zpack_close_reader(&reader);
    return rc;
}
