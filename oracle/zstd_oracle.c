/*
 * zstd_oracle.c — Zstandard frame decoder restated from RFC 8878.  TEST INFRASTRUCTURE ONLY.
 *
 * Stands in for ZSTD_decompressDCtx as the reference calls it (lib/zpack_read.c:380): decodes every
 * concatenated frame, skips skippable frames, no dictionary.  Where RFC 8878 leaves a malformed
 * input's fate open, the behaviour of libzstd 1.4.9 (the version in this image, the one the
 * compiled reference in oracle/_ref links) is followed and noted inline.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define ZSTD_MAGIC        0xFD2FB528U
#define SKIP_MAGIC_MASK   0xFFFFFFF0U
#define SKIP_MAGIC        0x184D2A50U
#define BLOCK_MAX         (128u << 10)

enum { E_OK = 0, E_CORRUPT = -1, E_DST_FULL = -3 };

static orc_zstd_stats g_stats;
const orc_zstd_stats* orc_zstd_last_stats(void) { return &g_stats; }

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static int highbit(uint32_t v) { int r = -1; while (v) { v >>= 1; r++; } return r; }

/* ------------------------------------------------------------------ bit readers */

/* forward, LSB-first (FSE table descriptions).  Bits past the end read as zero; the caller checks
 * the consumed byte count afterwards (libzstd FSE_readNCount does the same via a zero-padded copy). */
typedef struct { const uint8_t* p; size_t size; size_t bit; } fwd_bits;

static uint32_t fwd_read(fwd_bits* b, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
        size_t pos = b->bit + (size_t)i;
        size_t byte = pos >> 3;
        uint32_t bitv = byte < b->size ? (b->p[byte] >> (pos & 7)) & 1u : 0u;
        v |= bitv << i;
    }
    b->bit += (size_t)n;
    return v;
}

/* backward (FSE / Huffman payloads): the last byte carries a 1-bit end mark above the payload;
 * fields are read from the top down, each field little-endian.  Bits below position 0 are zero. */
typedef struct { const uint8_t* p; int64_t bits; } bwd_bits;

static int bwd_init(bwd_bits* b, const uint8_t* p, size_t size)
{
    if (size == 0 || p[size - 1] == 0) return -1;
    b->p = p;
    b->bits = (int64_t)(size - 1) * 8 + highbit(p[size - 1]);
    return 0;
}

static uint64_t bwd_peek_at(const bwd_bits* b, int64_t pos, int n)   /* bits [pos, pos+n) */
{
    uint64_t v = 0;
    for (int i = 0; i < n; i++) {
        int64_t q = pos + i;
        if (q >= 0) v |= (uint64_t)((b->p[q >> 3] >> (q & 7)) & 1u) << i;
    }
    return v;
}

static uint64_t bwd_read(bwd_bits* b, int n)
{
    b->bits -= n;
    return bwd_peek_at(b, b->bits, n);
}

/* ------------------------------------------------------------------ FSE */

typedef struct {
    uint8_t  sym[512];
    uint8_t  nbits[512];
    uint16_t base[512];
    int      al;             /* accuracy log; 0 => single-entry RLE table */
} fse_table;

/* RFC 8878 §4.1.1: read normalized counts.  Returns bytes consumed (>0) or -1. */
static int fse_read_ncount(const uint8_t* src, size_t size, int max_sym, int max_al, int16_t* freq, int* nsym, int* al_out)
{
    fwd_bits b = { src, size, 0 };
    int al = 5 + (int)fwd_read(&b, 4);
    if (al > max_al) return -1;
    int remaining = 1 << al;
    int s = 0;
    while (remaining > 0 && s <= max_sym) {
        int nb = highbit((uint32_t)remaining + 1) + 1;
        uint32_t val = fwd_read(&b, nb);
        uint32_t lower_mask = (1u << (nb - 1)) - 1;
        uint32_t threshold = (1u << nb) - 1 - ((uint32_t)remaining + 1);
        if ((val & lower_mask) < threshold) {
            b.bit -= 1;
            val &= lower_mask;
        } else if (val > lower_mask) {
            val -= threshold;
        }
        int proba = (int)val - 1;
        remaining -= proba < 0 ? 1 : proba;
        freq[s++] = (int16_t)proba;
        if (proba == 0) {
            uint32_t rep = fwd_read(&b, 2);
            for (;;) {
                for (uint32_t i = 0; i < rep; i++) {
                    if (s > max_sym) return -1;
                    freq[s++] = 0;
                }
                if (rep != 3) break;
                rep = fwd_read(&b, 2);
            }
        }
    }
    if (remaining != 0) return -1;
    size_t used = (b.bit + 7) >> 3;
    if (used > size) return -1;
    *nsym = s;
    *al_out = al;
    return (int)used;
}

/* RFC 8878 §4.1.1 "from normalized distribution to decoding tables" */
static int fse_build(fse_table* t, const int16_t* freq, int nsym, int al)
{
    int size = 1 << al;
    int high = size;
    uint16_t next[64];
    if (nsym > 64) return -1;
    t->al = al;
    for (int s = 0; s < nsym; s++) {
        if (freq[s] == -1) { t->sym[--high] = (uint8_t)s; next[s] = 1; }
    }
    int step = (size >> 1) + (size >> 3) + 3, mask = size - 1, pos = 0;
    for (int s = 0; s < nsym; s++) {
        if (freq[s] <= 0) continue;
        next[s] = (uint16_t)freq[s];
        for (int i = 0; i < freq[s]; i++) {
            t->sym[pos] = (uint8_t)s;
            do { pos = (pos + step) & mask; } while (pos >= high);
        }
    }
    if (pos != 0) return -1;
    for (int i = 0; i < size; i++) {
        uint16_t n = next[t->sym[i]]++;
        int nb = al - highbit(n);
        t->nbits[i] = (uint8_t)nb;
        t->base[i] = (uint16_t)(((uint32_t)n << nb) - (uint32_t)size);
    }
    return 0;
}

static void fse_build_rle(fse_table* t, uint8_t sym)
{
    t->al = 0; t->sym[0] = sym; t->nbits[0] = 0; t->base[0] = 0;
}

/* ------------------------------------------------------------------ Huffman */

typedef struct {
    uint8_t sym[4096];
    uint8_t nbits[4096];
    int     max_bits;
    int     valid;
} huf_table;

static int huf_build(huf_table* h, const uint8_t* weights, int n)     /* n includes the implied last one */
{
    uint32_t rank_count[14] = {0};
    uint32_t sum = 0;
    for (int i = 0; i < n; i++) {
        if (weights[i] > 12) return -1;
        rank_count[weights[i]]++;
        if (weights[i]) sum += 1u << (weights[i] - 1);
    }
    int max_bits = highbit(sum);          /* sum is an exact power of two here */
    if (max_bits < 1 || max_bits > 12 || sum != (1u << max_bits)) return -1;
    /* libzstd HUF_readStats: "by construction: at least 2 elts of rank 1, must be even" */
    if (rank_count[1] < 2 || (rank_count[1] & 1)) return -1;
    h->max_bits = max_bits;
    /* weight w => code length max_bits+1-w, 2^(w-1) table slots; lowest weights first, natural order inside a weight */
    uint32_t start[14];
    uint32_t pos = 0;
    for (int w = 1; w <= max_bits; w++) { start[w] = pos; pos += rank_count[w] << (w - 1); }
    for (int i = 0; i < n; i++) {
        int w = weights[i];
        if (!w) continue;
        uint32_t len = 1u << (w - 1);
        for (uint32_t k = 0; k < len; k++) {
            h->sym[start[w] + k] = (uint8_t)i;
            h->nbits[start[w] + k] = (uint8_t)(max_bits + 1 - w);
        }
        start[w] += len;
    }
    h->valid = 1;
    return 0;
}

/* RFC 8878 §4.2.1: Huffman tree description.  Returns bytes consumed or -1. */
static int huf_read_tree(huf_table* h, const uint8_t* src, size_t size)
{
    uint8_t weights[256];
    int n = 0;
    if (size < 1) return -1;
    unsigned hb = src[0];
    size_t used;
    if (hb >= 128) {
        n = (int)hb - 127;
        size_t bytes = ((size_t)n + 1) / 2;
        if (1 + bytes > size) return -1;
        for (int i = 0; i < n; i++) {
            uint8_t b = src[1 + i / 2];
            weights[i] = (i & 1) ? (b & 15) : (b >> 4);
        }
        used = 1 + bytes;
        g_stats.huf_direct_weights++;
    } else {
        size_t csize = hb;
        if (csize == 0 || 1 + csize > size) return -1;
        int16_t freq[16]; int nsym, al;
        int tb = fse_read_ncount(src + 1, csize, 12, 6, freq, &nsym, &al);
        if (tb < 0) return -1;
        fse_table t;
        if (fse_build(&t, freq, nsym, al) < 0) return -1;
        bwd_bits b;
        if (bwd_init(&b, src + 1 + tb, csize - (size_t)tb) < 0) return -1;
        uint32_t s1 = (uint32_t)bwd_read(&b, al), s2 = (uint32_t)bwd_read(&b, al);
        /* two interleaved states; the stream ends when an update over-reads (libzstd FSE_decompress tail) */
        for (;;) {
            if (n > 253) return -1;
            weights[n++] = t.sym[s1];
            s1 = t.base[s1] + (uint32_t)bwd_read(&b, t.nbits[s1]);
            if (b.bits < 0) { weights[n++] = t.sym[s2]; break; }
            if (n > 253) return -1;
            weights[n++] = t.sym[s2];
            s2 = t.base[s2] + (uint32_t)bwd_read(&b, t.nbits[s2]);
            if (b.bits < 0) { weights[n++] = t.sym[s1]; break; }
        }
        used = 1 + csize;
        g_stats.huf_fse_weights++;
    }
    /* the last weight is implied: it completes the sum of 2^(w-1) to a power of two */
    uint32_t sum = 0;
    for (int i = 0; i < n; i++) {
        if (weights[i] > 12) return -1;
        if (weights[i]) sum += 1u << (weights[i] - 1);
    }
    if (sum == 0) return -1;
    int max_bits = highbit(sum) + 1;
    if (max_bits > 12) return -1;
    uint32_t left = (1u << max_bits) - sum;
    if (left & (left - 1)) return -1;
    weights[n++] = (uint8_t)(highbit(left) + 1);
    if (huf_build(h, weights, n) < 0) return -1;
    return (int)used;
}

static int huf_decode_stream(const huf_table* h, const uint8_t* src, size_t size, uint8_t* out, size_t n)
{
    bwd_bits b;
    if (bwd_init(&b, src, size) < 0) return -1;
    int mb = h->max_bits;
    for (size_t i = 0; i < n; i++) {
        uint32_t idx = (uint32_t)bwd_peek_at(&b, b.bits - mb, mb);
        out[i] = h->sym[idx];
        b.bits -= h->nbits[idx];
    }
    return b.bits == 0 ? 0 : -1;           /* libzstd: BIT_endOfDStream required */
}

/* ------------------------------------------------------------------ sequences */

static const uint32_t LL_BASE[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,
                                      0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
static const uint8_t  LL_BITS[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const uint32_t ML_BASE[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                      35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };
static const uint8_t  ML_BITS[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                      1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static const int16_t LL_DEFAULT[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static const int16_t ML_DEFAULT[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                        -1,-1,-1,-1,-1,-1,-1 };
static const int16_t OF_DEFAULT[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

typedef struct {
    huf_table huf;
    fse_table ll, of, ml;
    int       seq_tables_valid;      /* libzstd dctx->fseEntropy */
    uint64_t  rep[3];
    uint8_t*  lit;                   /* BLOCK_MAX bytes */
} frame_ctx;

/* one of the three symbol-compression tables; returns bytes consumed or -1 */
static int read_seq_table(fse_table* t, int mode, int which, const uint8_t* src, size_t size,
                          int max_sym, int max_al, const int16_t* def, int def_n, int def_al, int have_prev)
{
    g_stats.seq_mode[which][mode]++;
    switch (mode) {
    case 0:
        if (fse_build(t, def, def_n, def_al) < 0) return -1;
        return 0;
    case 1:
        if (size < 1 || src[0] > max_sym) return -1;
        fse_build_rle(t, src[0]);
        return 1;
    case 2: {
        int16_t freq[64]; int nsym, al;
        int used = fse_read_ncount(src, size, max_sym, max_al, freq, &nsym, &al);
        if (used < 0) return -1;
        if (fse_build(t, freq, nsym, al) < 0) return -1;
        return used;
    }
    default:
        return have_prev ? 0 : -1;
    }
}

static int decode_literals(frame_ctx* c, const uint8_t* src, size_t size, size_t* lit_size, size_t* consumed)
{
    if (size < 1) return -1;
    unsigned type = src[0] & 3, fmt = (src[0] >> 2) & 3;
    if (type == 0 || type == 1) {
        size_t hl, n;
        if ((fmt & 1) == 0) { hl = 1; n = src[0] >> 3; }
        else if (fmt == 1)  { hl = 2; if (size < 2) return -1; n = (src[0] >> 4) | ((size_t)src[1] << 4); }
        else                { hl = 3; if (size < 3) return -1; n = (src[0] >> 4) | ((size_t)src[1] << 4) | ((size_t)src[2] << 12); }
        if (n > BLOCK_MAX) return -1;
        if (type == 0) {
            if (hl + n > size) return -1;
            memcpy(c->lit, src + hl, n);
            *consumed = hl + n;
            g_stats.lit_raw++;
        } else {
            if (hl + 1 > size) return -1;
            memset(c->lit, src[hl], n);
            *consumed = hl + 1;
            g_stats.lit_rle++;
        }
        *lit_size = n;
        return 0;
    }
    /* Huffman-compressed (2) or treeless (3) */
    if (size < 5) return -1;                      /* libzstd: "srcSize >= MIN_CBLOCK_SIZE == 3; here we need up to 5" */
    size_t hl, regen, csize; int streams;
    uint32_t v = rd32(src);
    switch (fmt) {
    case 0: hl = 3; streams = 1; regen = (v >> 4) & 0x3FF;  csize = (v >> 14) & 0x3FF; break;
    case 1: hl = 3; streams = 4; regen = (v >> 4) & 0x3FF;  csize = (v >> 14) & 0x3FF; break;
    case 2: hl = 4; streams = 4; regen = (v >> 4) & 0x3FFF; csize = v >> 18; break;
    default: hl = 5; streams = 4; regen = (v >> 4) & 0x3FFFF; csize = (v >> 22) | ((size_t)src[4] << 10); break;
    }
    if (regen > BLOCK_MAX || hl + csize > size) return -1;
    const uint8_t* p = src + hl;
    size_t left = csize;
    if (type == 2) {
        int used = huf_read_tree(&c->huf, p, left);
        if (used < 0) return -1;
        p += used; left -= (size_t)used;
        g_stats.lit_huf++;
    } else {
        if (!c->huf.valid) return -1;
        g_stats.lit_treeless++;
    }
    if (streams == 1) {
        if (huf_decode_stream(&c->huf, p, left, c->lit, regen) < 0) return -1;
        g_stats.lit_huf_1stream++;
    } else {
        if (left < 10) return -1;                 /* libzstd: jump table + 1 byte per stream */
        size_t s1 = p[0] | ((size_t)p[1] << 8), s2 = p[2] | ((size_t)p[3] << 8), s3 = p[4] | ((size_t)p[5] << 8);
        if (6 + s1 + s2 + s3 > left) return -1;
        size_t s4 = left - 6 - s1 - s2 - s3;
        size_t seg = (regen + 3) / 4;
        if (seg * 3 > regen) return -1;
        const uint8_t* q = p + 6;
        if (huf_decode_stream(&c->huf, q, s1, c->lit, seg) < 0) return -1;
        if (huf_decode_stream(&c->huf, q + s1, s2, c->lit + seg, seg) < 0) return -1;
        if (huf_decode_stream(&c->huf, q + s1 + s2, s3, c->lit + 2 * seg, seg) < 0) return -1;
        if (huf_decode_stream(&c->huf, q + s1 + s2 + s3, s4, c->lit + 3 * seg, regen - 3 * seg) < 0) return -1;
        g_stats.lit_huf_4stream++;
    }
    *lit_size = regen;
    *consumed = hl + csize;
    return 0;
}

/* decode + execute the sequences of one compressed block; `frame_out` = bytes of this frame already in dst */
/* optional trace of the decoded sequences (offset | match length << 29 | literal length << 47), for the tests that
 * check the GPU's FSE pre-decode kernel (zpack_amd/csrc/zstd_fse4.h) sequence by sequence */
static uint64_t* g_trace; static size_t g_trace_cap, g_trace_n; static int64_t* g_trace_bits;
void orc_zstd_trace_bits(int64_t* buf) { g_trace_bits = buf; }
void orc_zstd_trace(uint64_t* buf, size_t cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
size_t orc_zstd_trace_count(void) { return g_trace_n; }

static int decode_block(frame_ctx* c, const uint8_t* src, size_t size, uint8_t* dst, size_t dst_cap,
                        size_t frame_out, size_t* produced)
{
    size_t lit_size = 0, used = 0;
    if (size < 3) return E_CORRUPT;               /* libzstd MIN_CBLOCK_SIZE */
    if (decode_literals(c, src, size, &lit_size, &used) < 0) return E_CORRUPT;
    const uint8_t* p = src + used;
    size_t left = size - used;

    if (left < 1) return E_CORRUPT;
    size_t nseq = p[0];
    if (nseq == 0) {
        if (left != 1) return E_CORRUPT;          /* libzstd ZSTD_decodeSeqHeaders: srcSize_wrong */
        p += 1; left -= 1;
    } else if (nseq < 128) {
        p += 1; left -= 1;
    } else if (nseq < 255) {
        if (left < 2) return E_CORRUPT;
        nseq = ((nseq - 128) << 8) + p[1];
        p += 2; left -= 2;
    } else {
        if (left < 3) return E_CORRUPT;
        nseq = (size_t)p[1] + ((size_t)p[2] << 8) + 0x7F00;
        p += 3; left -= 3;
    }

    size_t op = 0, lit_pos = 0;
    if (nseq > 0) {
        if (left < 1) return E_CORRUPT;
        unsigned modes = p[0];
        p += 1; left -= 1;
        int r;
        r = read_seq_table(&c->ll, (modes >> 6) & 3, 0, p, left, 35, 9, LL_DEFAULT, 36, 6, c->seq_tables_valid);
        if (r < 0) return E_CORRUPT;
        p += r; left -= (size_t)r;
        r = read_seq_table(&c->of, (modes >> 4) & 3, 1, p, left, 31, 8, OF_DEFAULT, 29, 5, c->seq_tables_valid);
        if (r < 0) return E_CORRUPT;
        p += r; left -= (size_t)r;
        r = read_seq_table(&c->ml, (modes >> 2) & 3, 2, p, left, 52, 9, ML_DEFAULT, 53, 6, c->seq_tables_valid);
        if (r < 0) return E_CORRUPT;
        p += r; left -= (size_t)r;
        c->seq_tables_valid = 1;

        bwd_bits b;
        if (bwd_init(&b, p, left) < 0) return E_CORRUPT;
        uint32_t sll = (uint32_t)bwd_read(&b, c->ll.al);
        uint32_t sof = (uint32_t)bwd_read(&b, c->of.al);
        uint32_t sml = (uint32_t)bwd_read(&b, c->ml.al);
        g_stats.sequences += nseq;

        for (size_t i = 0; i < nseq; i++) {
            unsigned of_code = c->of.sym[sof], ll_code = c->ll.sym[sll], ml_code = c->ml.sym[sml];
            if (of_code > 31 || ll_code > 35 || ml_code > 52) return E_CORRUPT;
            uint64_t of_val = ((uint64_t)1 << of_code) + bwd_read(&b, (int)of_code);
            uint64_t mlen = ML_BASE[ml_code] + bwd_read(&b, ML_BITS[ml_code]);
            uint64_t llen = LL_BASE[ll_code] + bwd_read(&b, LL_BITS[ll_code]);
            uint64_t offset;
            if (of_val > 3) {
                offset = of_val - 3;
                c->rep[2] = c->rep[1]; c->rep[1] = c->rep[0]; c->rep[0] = offset;
            } else {
                /* repeat offsets (RFC 8878 §3.1.1.5); the comparison is on the LL *code* like libzstd (llBase == 0) */
                uint64_t idx = of_val - 1 + (ll_code == 0 ? 1 : 0);
                g_stats.repcode_uses++;
                if (idx == 0) {
                    offset = c->rep[0];
                } else {
                    uint64_t t = idx == 3 ? c->rep[0] - 1 : c->rep[idx];
                    if (t == 0) t = 1;                       /* libzstd: "0 is not valid; input is corrupted; force offset to 1" */
                    if (idx != 1) c->rep[2] = c->rep[1];
                    c->rep[1] = c->rep[0];
                    c->rep[0] = offset = t;
                }
            }
            /* libzstd 1.4.9 updates all three states after every sequence, the last included */
            sll = c->ll.base[sll] + (uint32_t)bwd_read(&b, c->ll.nbits[sll]);
            sml = c->ml.base[sml] + (uint32_t)bwd_read(&b, c->ml.nbits[sml]);
            sof = c->of.base[sof] + (uint32_t)bwd_read(&b, c->of.nbits[sof]);

            if (g_trace) {
                if (g_trace_n < g_trace_cap) { g_trace[g_trace_n] = offset | (mlen << 29) | (llen << 47); if (g_trace_bits) g_trace_bits[g_trace_n] = (int64_t)b.bits; }
                g_trace_n++;
            }
            if (llen + mlen > dst_cap - op) return E_DST_FULL;
            if (llen > lit_size - lit_pos) return E_CORRUPT;
            memcpy(dst + op, c->lit + lit_pos, (size_t)llen);
            op += (size_t)llen; lit_pos += (size_t)llen;
            if (offset > frame_out + op) return E_CORRUPT;
            const uint8_t* m = dst + op - offset;
            for (uint64_t k = 0; k < mlen; k++) dst[op + k] = m[k];
            op += (size_t)mlen;
        }
        if (b.bits > 0) return E_CORRUPT;          /* libzstd 1.4.9: stream must not be under-consumed */
    }
    size_t rest = lit_size - lit_pos;
    if (rest > dst_cap - op) return E_DST_FULL;
    memcpy(dst + op, c->lit + lit_pos, rest);
    op += rest;
    if (op > BLOCK_MAX) return E_CORRUPT;
    *produced = op;
    return E_OK;
}

/* one frame; returns E_* and advances *ip / *op */
static int decode_frame(frame_ctx* c, const uint8_t* src, size_t size, size_t* ip_io, uint8_t* dst, size_t dst_cap, size_t* op_io)
{
    size_t ip = *ip_io, op = *op_io;
    size_t frame_start = op;
    if (size - ip < 6) return E_CORRUPT;          /* magic + FHD + at least one more byte */
    ip += 4;
    unsigned fhd = src[ip++];
    unsigned fcs_flag = fhd >> 6, single = (fhd >> 5) & 1, cksum = (fhd >> 2) & 1, did_flag = fhd & 3;
    if (fhd & 0x08) return E_CORRUPT;
    uint64_t window = 0;
    if (!single) {
        if (size - ip < 1) return E_CORRUPT;
        unsigned wd = src[ip++];
        unsigned wlog = 10 + (wd >> 3);
        if (wlog > 31) return E_CORRUPT;           /* libzstd ZSTD_WINDOWLOG_MAX (64-bit) */
        window = (uint64_t)1 << wlog;
        window += (window >> 3) * (wd & 7);
    }
    static const unsigned did_bytes[4] = { 0, 1, 2, 4 };
    unsigned dn = did_bytes[did_flag];
    if (size - ip < dn) return E_CORRUPT;
    uint32_t dict_id = 0;
    for (unsigned i = 0; i < dn; i++) dict_id |= (uint32_t)src[ip + i] << (8 * i);
    ip += dn;
    if (dict_id != 0) return E_CORRUPT;           /* libzstd: dictionary_wrong (no dictionary is ever loaded) */
    unsigned fn = fcs_flag == 0 ? (single ? 1 : 0) : (fcs_flag == 1 ? 2 : (fcs_flag == 2 ? 4 : 8));
    if (size - ip < fn) return E_CORRUPT;
    uint64_t fcs = 0;
    for (unsigned i = 0; i < fn; i++) fcs |= (uint64_t)src[ip + i] << (8 * i);
    if (fn == 2) fcs += 256;
    ip += fn;
    if (single) window = fcs;
    g_stats.frames++; g_stats.window_size = window; g_stats.single_segment = single;
    g_stats.has_fcs = fn != 0; g_stats.has_checksum = cksum;

    c->huf.valid = 0;
    c->seq_tables_valid = 0;
    c->rep[0] = 1; c->rep[1] = 4; c->rep[2] = 8;

    for (;;) {
        if (size - ip < 3) return E_CORRUPT;
        uint32_t bh = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16);
        ip += 3;
        int last = bh & 1; unsigned type = (bh >> 1) & 3; size_t bsize = bh >> 3;
        g_stats.blocks++;
        if (type == 3) return E_CORRUPT;
        if (type == 0) {
            if (bsize > size - ip) return E_CORRUPT;
            if (bsize > dst_cap - op) return E_DST_FULL;
            memcpy(dst + op, src + ip, bsize);
            ip += bsize; op += bsize;
            g_stats.raw_blocks++;
        } else if (type == 1) {
            if (size - ip < 1) return E_CORRUPT;
            if (bsize > dst_cap - op) return E_DST_FULL;
            memset(dst + op, src[ip], bsize);
            ip += 1; op += bsize;
            g_stats.rle_blocks++;
        } else {
            if (bsize > size - ip) return E_CORRUPT;
            if (bsize >= BLOCK_MAX) return E_CORRUPT;          /* libzstd: srcSize >= ZSTD_BLOCKSIZE_MAX */
            size_t got = 0;
            int r = decode_block(c, src + ip, bsize, dst + op, dst_cap - op, op - frame_start, &got);
            if (r != E_OK) return r;
            ip += bsize; op += got;
            g_stats.comp_blocks++;
        }
        if (last) break;
    }
    if (fn != 0 && (uint64_t)(op - frame_start) != fcs) return E_CORRUPT;
    if (cksum) {
        if (size - ip < 4) return E_CORRUPT;
        if ((uint32_t)orc_xxh64(dst + frame_start, op - frame_start, 0) != rd32(src + ip)) return E_CORRUPT;
        ip += 4;
    }
    *ip_io = ip; *op_io = op;
    return E_OK;
}

int orc_zstd_decode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap, size_t* produced)
{
    size_t ip = 0, op = 0;
    int rc = E_OK;
    *produced = 0;
    memset(&g_stats, 0, sizeof(g_stats));
    frame_ctx* c = (frame_ctx*)malloc(sizeof(frame_ctx));
    if (!c) return E_CORRUPT;
    c->lit = (uint8_t*)malloc(BLOCK_MAX);
    if (!c->lit) { free(c); return E_CORRUPT; }

    int frames = 0;
    while (ip < src_size) {
        if (src_size - ip < 4) { rc = E_CORRUPT; break; }       /* libzstd: srcSize_wrong */
        uint32_t magic = rd32(src + ip);
        if ((magic & SKIP_MAGIC_MASK) == SKIP_MAGIC) {
            if (src_size - ip < 8) { rc = E_CORRUPT; break; }
            uint32_t sz = rd32(src + ip + 4);
            if (src_size - ip - 8 < sz) { rc = E_CORRUPT; break; }
            ip += 8 + (size_t)sz;
            continue;
        }
        if (magic != ZSTD_MAGIC) { rc = E_CORRUPT; break; }      /* libzstd: prefix_unknown */
        rc = decode_frame(c, src, src_size, &ip, dst, dst_cap, &op);
        if (rc != E_OK) break;
        frames++;
    }
    free(c->lit);
    free(c);
    *produced = op;
    (void)frames;
    return rc;
}
