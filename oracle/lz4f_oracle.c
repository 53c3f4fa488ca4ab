/*
 * lz4f_oracle.c — LZ4 Frame (v1.6.x) + LZ4 Block decode, restated from the published format
 * documents.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Stands in for the LZ4F_decompress loop of the reference (lib/zpack_read.c:414-439).  The frames
 * the reference writer emits (lib/zpack_write.c:199-211: zeroed LZ4F_preferences_t + level) are
 * `04 22 4D 18 | FLG 0x40 | BD 0x40 | HC | blocks... | 00 00 00 00`: block-linked, 64 KiB blocks,
 * no checksums, no content size.  The decoder below accepts the whole format (block-independent
 * frames, content size, dictID field, block and content checksums, larger block sizes, skippable
 * frames) because LZ4F_decompress does.
 */
#include "oracle.h"
#include <string.h>

#define LZ4F_MAGIC          0x184D2204U
#define LZ4F_SKIP_MAGIC_LO  0x184D2A50U
#define LZ4F_SKIP_MAGIC_HI  0x184D2A5FU

enum { E_OK = 0, E_MALFORMED = -1, E_TRUNCATED = -2, E_DST_FULL = -3 };

static uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/*
 * LZ4 block: sequences of [token][litlen ext*][literals][offset LE16][matchlen ext*]; the last
 * sequence stops after its literals.  `hist` = number of valid bytes directly before dst that
 * matches may reach into (linked blocks: up to 64 KiB of previous output).
 */
int orc_lz4_block_decode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap,
                         size_t hist, size_t* produced)
{
    size_t ip = 0, op = 0;
    *produced = 0;
    if (src_size == 0) return E_MALFORMED;
    for (;;) {
        if (ip >= src_size) return E_MALFORMED;
        unsigned token = src[ip++];
        size_t lit = token >> 4;
        if (lit == 15) {
            unsigned b;
            do {
                if (ip >= src_size) return E_MALFORMED;
                b = src[ip++];
                lit += b;
            } while (b == 255);
        }
        if (lit > src_size - ip) return E_MALFORMED;
        if (lit > dst_cap - op) { *produced = op; return E_DST_FULL; }
        memcpy(dst + op, src + ip, lit);
        ip += lit; op += lit;
        if (ip == src_size) break;                     /* last sequence: literals only */
        if (src_size - ip < 2) return E_MALFORMED;
        size_t offset = (size_t)src[ip] | ((size_t)src[ip + 1] << 8);
        ip += 2;
        if (offset == 0 || offset > op + hist) return E_MALFORMED;
        size_t mlen = token & 15;
        if (mlen == 15) {
            unsigned b;
            do {
                if (ip >= src_size) return E_MALFORMED;
                b = src[ip++];
                mlen += b;
            } while (b == 255);
        }
        mlen += 4;
        if (mlen > dst_cap - op) { *produced = op; return E_DST_FULL; }
        /* byte-wise so that overlapping matches (offset < length) replicate correctly */
        const uint8_t* m = dst + op - offset;
        for (size_t i = 0; i < mlen; i++) dst[op + i] = m[i];
        op += mlen;
    }
    *produced = op;
    return E_OK;
}

static size_t block_max_size(unsigned bd_code)
{
    switch (bd_code) {
    case 4: return 64u << 10;
    case 5: return 256u << 10;
    case 6: return 1u << 20;
    case 7: return 4u << 20;
    default: return 0;
    }
}

/* one frame (or one skippable frame) at *ipp, output at *opp; E_OK = the frame is complete (LZ4F_decompress returned 0) */
static int lz4f_one_frame(const uint8_t* src, size_t src_size, size_t* ipp, uint8_t* dst, size_t dst_cap, size_t* opp, size_t* produced)
{
    size_t ip = *ipp, op = *opp;
    /* LZ4F_decompress decodes no header before it holds minFHSize = 7 bytes: fewer is "need more input", whatever they are */
    if (src_size - ip < 7) { *produced = op; return E_TRUNCATED; }
    uint32_t magic = rd32(src + ip);
    if (magic >= LZ4F_SKIP_MAGIC_LO && magic <= LZ4F_SKIP_MAGIC_HI) {
        if (src_size - ip < 8) { *produced = op; return E_TRUNCATED; }
        uint32_t sz = rd32(src + ip + 4);
        if (src_size - ip - 8 < sz) { *produced = op; return E_TRUNCATED; }
        *ipp = ip + 8 + (size_t)sz;
        return E_OK;
    }
    if (magic != LZ4F_MAGIC) return E_MALFORMED;
    size_t desc_start = ip + 4;
    unsigned flg = src[ip + 4], bd = src[ip + 5];
    unsigned version = flg >> 6;
    int block_indep = (flg >> 5) & 1, block_cksum = (flg >> 4) & 1, has_csize = (flg >> 3) & 1;
    int content_cksum = (flg >> 2) & 1, has_dict = flg & 1;
    /* LZ4F_decodeHeader's order: FLG is validated, then the header has to be there in full, then BD and the header checksum */
    if (version != 1 || (flg & 2)) return E_MALFORMED;
    size_t hdr = 7 + (has_csize ? 8 : 0) + (has_dict ? 4 : 0);
    if (src_size - ip < hdr) { *produced = op; return E_TRUNCATED; }
    if (bd & 0x8F) return E_MALFORMED;
    size_t bmax = block_max_size((bd >> 4) & 7);
    if (!bmax) return E_MALFORMED;
    uint64_t content_size = 0;
    if (has_csize) content_size = (uint64_t)rd32(src + ip + 6) | ((uint64_t)rd32(src + ip + 10) << 32);
    unsigned hc = src[ip + hdr - 1];
    if (((orc_xxh32(src + desc_start, hdr - 5, 0) >> 8) & 0xFF) != hc) return E_MALFORMED;
    ip += hdr;

    size_t frame_start_op = op;
    for (;;) {
        if (src_size - ip < 4) { *produced = op; return E_TRUNCATED; }
        uint32_t bh = rd32(src + ip);
        ip += 4;
        if (bh == 0) break;                                    /* EndMark */
        int raw = (bh >> 31) & 1;
        size_t bsz = bh & 0x7FFFFFFFU;
        if (bsz > bmax) return E_MALFORMED;
        if (src_size - ip < bsz + (block_cksum ? 4u : 0u)) {
            /* what is there of a raw block still streams out before the loop starves */
            if (raw) {
                size_t n = src_size - ip; if (n > bsz) n = bsz;
                if (n > dst_cap - op) { n = dst_cap - op; memcpy(dst + op, src + ip, n); *produced = op + n; return E_DST_FULL; }
                memcpy(dst + op, src + ip, n); op += n;
            }
            *produced = op; return E_TRUNCATED;
        }
        if (block_cksum && orc_xxh32(src + ip, bsz, 0) != rd32(src + ip + bsz)) return E_MALFORMED;
        if (raw) {
            if (bsz > dst_cap - op) {
                size_t n = dst_cap - op; memcpy(dst + op, src + ip, n); *produced = op + n; return E_DST_FULL;
            }
            memcpy(dst + op, src + ip, bsz);
            op += bsz;
        } else {
            size_t hist = block_indep ? 0 : (op - frame_start_op);
            if (hist > 65536) hist = 65536;
            size_t cap = dst_cap - op, got = 0;
            int limited = 0;
            if (cap > bmax) { cap = bmax; limited = 1; }
            int r = orc_lz4_block_decode(src + ip, bsz, dst + op, cap, hist, &got);
            if (r == E_DST_FULL) {
                if (limited) return E_MALFORMED;               /* block decodes to more than its max */
                *produced = op + got; return E_DST_FULL;
            }
            if (r != E_OK) return E_MALFORMED;
            op += got;
        }
        ip += bsz + (block_cksum ? 4u : 0u);
    }
    if (content_cksum) {
        if (src_size - ip < 4) { *produced = op; return E_TRUNCATED; }
        if (orc_xxh32(dst + frame_start_op, op - frame_start_op, 0) != rd32(src + ip)) return E_MALFORMED;
        ip += 4;
    }
    if (has_csize && content_size != (uint64_t)(op - frame_start_op)) return E_MALFORMED;
    *ipp = ip; *opp = op;
    return E_OK;
}

/*
 * The caller's loop of the reference (lib/zpack_read.c:414-439): `while (avail_out > 0 && avail_in > 0) LZ4F_decompress(...)`.
 * A frame (or skippable frame) that completes returns 0; the loop goes on with whatever input is left, so an entry may hold
 * several frames back to back; it stops when the input or the output space is used up, and what decides the verdict is the
 * LAST return value (0 = a frame boundary: OK; otherwise FILE_INCOMPLETE / BUFFER_TOO_SMALL).
 */
int orc_lz4f_decode(const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap, size_t* produced)
{
    size_t ip = 0, op = 0;
    *produced = 0;
    for (int first = 1; ; first = 0) {
        if (!first && (ip == src_size || op == dst_cap)) break;
        int r = lz4f_one_frame(src, src_size, &ip, dst, dst_cap, &op, produced);
        if (r != E_OK) return r;
    }
    *produced = op;
    return E_OK;
}
