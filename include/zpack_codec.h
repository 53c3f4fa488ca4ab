/*
 * zpack_codec.h — the thin C-ABI between the ZPack host library and the MI355X entry codec.
 *
 * This is the INWARD drop-in boundary: every third-party codec call the reference
 * makes on the per-entry hot path is replaced by one of the entry points below.  Plain C, plain
 * pointers and sizes, no torch / HIP types in any signature (streams travel as void*).
 *
 *   reference call site (file:line under /root/reference)              replaced by
 *   ------------------------------------------------------------------ ---------------------------
 *   lib/zpack_read.c:350-468   switch(comp_method){memcpy |            zpk_codec_decode_batch_device
 *        ZSTD_decompressDCtx :380 | LZ4F_decompress loop :414-439}      zpk_codec_decode_batch_host
 *        + XXH3_64bits verify :466-468, guards :328-332                 (one descriptor per entry)
 *   lib/zpack_write.c:161-224  zpack_compress_file {memcpy |            zpk_codec_encode_batch_device
 *        ZSTD_compressCCtx :179 | LZ4F_compressBegin/Update/End         zpk_codec_encode_batch_host
 *        :204-210} + XXH3_64bits :256
 *   lib/zpack_write.c:125-150  zpack_get_compress_bound                 zpk_codec_compress_bound
 *   lib/zpack_write.c:338      write_offset += comp_size (per file)     zpk_codec_pack_batch_device (scan + compaction)
 *   lib/zpack_read.c:466, lib/zpack_write.c:256  XXH3_64bits            zpk_codec_hash_batch_device / _host
 *   lib/zpack_read.c:17-31,776-812; lib/zpack_write.c:20-34,899-935     zpk_codec_create / _destroy / _reset
 *        ZSTD_createDCtx / LZ4F_createDecompressionContext / ...        (one opaque context for every method)
 *   lib/zpack_stream.c:4-28 XXH3 state; zpack_read.c:515-640;           zpk_dstream_* / zpk_cstream_*
 *        zpack_write.c:461-685 streaming calls                          (chunk aggregation over the batch codec)
 *
 * Status values are `enum zpack_result` codes (zpack.h): the device evaluates the same guards in the
 * same order as zpack_read_file, so `results[i].status` is exactly what zpack_read_file would have
 * returned for entry i, and one bad entry never poisons the batch.
 *
 * There is NO CPU fallback behind this interface: if no HIP device is usable, zpk_codec_create fails
 * with ZPK_E_NO_DEVICE and every zpack.h call that needs the codec returns ZPACK_ERROR_NOT_AVAILABLE.
 */
#ifndef ZPACK_CODEC_H
#define ZPACK_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZPK_CODEC_ABI_VERSION 3      /* 3: the two-stage LZ4 path of version 2 is gone (options 2..5 are ZPK_E_INVALID again, decode_stats2 out[2] = out[4] = 0,
                                        out[3] = LZ4 entries that are mostly runs, decoded by k_lz4_left); 2: set_option has options again */

/* return codes of the zpk_* entry points themselves (not per-entry statuses) */
enum {
    ZPK_OK = 0,
    ZPK_E_NO_DEVICE = -1,      /* no usable HIP device / runtime */
    ZPK_E_INVALID = -2,        /* bad argument */
    ZPK_E_NOMEM = -3,          /* host or device allocation failed */
    ZPK_E_LAUNCH = -4          /* HIP launch / copy error (hipGetLastError text via zpk_codec_last_error) */
};

/* compression methods — values of zpack_compression_method (zpack.h) */
enum { ZPK_METHOD_NONE = 0, ZPK_METHOD_ZSTD = 1, ZPK_METHOD_LZ4 = 2 };

/* One codec = one device context with ONE in-flight batch: its work lists, counters and staging buffers are shared by
 * every call.  Threading contract: the host-pointer entry points (*_host, the streaming triple) hold the codec's lock for
 * their whole duration — threads sharing a codec are serialised, never corrupted; the device-pointer entry points hold it
 * while they enqueue, and successive batches on ONE stream are ordered by that stream.  Do not drive one codec from two
 * streams at once: create one codec per stream / per thread (zpack.h contexts do exactly that). */
typedef struct zpk_codec zpk_codec;

/* One entry to decode.  Mirrors zpack_file_entry (lib/zpack.h:71-80) + the (buffer, max_size) pair of
 * zpack_read_file (lib/zpack.h:383).  `src_offset` is entry->offset: a byte offset into the archive
 * image `src` handed to the batch call (whose size plays reader->file_size in the :331 guard). */
typedef struct zpk_decode_desc {
    uint64_t src_offset;
    uint64_t comp_size;
    uint64_t uncomp_size;
    uint64_t expect_hash;
    uint64_t dst_offset;       /* byte offset of this entry's output slot inside `dst` */
    uint64_t dst_capacity;     /* max_size */
    uint32_t method;
    uint32_t flags;            /* ZPK_DF_* */
} zpk_decode_desc;

#define ZPK_DF_SKIP_HASH 1u    /* decode only; results[i].hash is still produced, status ignores it */
#define ZPK_DF_GENERAL   2u    /* accepted and ignored (round 2 used it to keep an entry off an opt-in LZ4 path that no longer exists) */

typedef struct zpk_decode_result {
    int32_t  status;           /* enum zpack_result */
    uint32_t detail;           /* codec-specific error detail (what the reference keeps in last_return) */
    uint64_t produced;         /* bytes written to the slot */
    uint64_t hash;             /* XXH3-64 of dst[0, uncomp_size) — what zpack_read.c:466 computes */
} zpk_decode_result;

typedef struct zpk_encode_desc {
    uint64_t src_offset;       /* plaintext offset inside `src` */
    uint64_t size;
    uint64_t dst_offset;       /* output slot offset inside `dst` */
    uint64_t dst_capacity;
    uint32_t method;           /* ZPK_METHOD_*, optionally | ZPK_EF_PIECE */
    int32_t  level;
} zpk_encode_desc;

/* The "entry" is a PIECE of a larger archive entry (large entries of the host write path, the streaming writer): its output is a run of
 * BLOCKS — no frame header, no EndMark / Last_Block — that whoever assembles the entry puts between a frame header and the end of the
 * frame (LZ4: the EndMark; Zstandard: an empty last block), so that the entry is ONE frame as the reference writer produces it
 * (lib/zpack_write.c:179, :204-210; round 4 made every piece a frame of its own: ABI 3 changed the meaning).  A piece's blocks do not
 * refer to the piece before it.  result.hash is left 0 (the entry's hash covers all pieces). */
#define ZPK_EF_PIECE 0x80000000u

typedef struct zpk_encode_result {
    int32_t  status;
    uint32_t detail;
    uint64_t comp_size;
    uint64_t hash;             /* XXH3-64 of the plaintext (lib/zpack_write.c:256) */
} zpk_encode_result;

/* ---- context lifecycle -------------------------------------------------------------------- */
int         zpk_codec_abi_version(void);
int         zpk_codec_device_count(void);
/* device: HIP ordinal, or -1 for "env ZPACK_AMD_DEVICE, else LOCAL_RANK, else 0" */
int         zpk_codec_create(zpk_codec** out, int device);
void        zpk_codec_destroy(zpk_codec* c);
void        zpk_codec_reset(zpk_codec* c);                  /* after an abandoned stream / error */
const char* zpk_codec_last_error(const zpk_codec* c);
int         zpk_codec_device(const zpk_codec* c);
/* options (ZPK_E_INVALID for anything else):
 *   ZPK_OPT_ENC_SPLIT_MIN           zpk_codec_encode_batch_host: an entry of at least `value` bytes is compressed as a SEQUENCE OF
 *                                   FRAMES, one per 512 KiB of plaintext, all of them side by side (one wave encodes one frame: a
 *                                   256 MiB entry is 512 waves instead of one), its XXH3 by the whole chip (per-block partial sums,
 *                                   then one short chain).  Both readers of the reference continue with the next frame
 *                                   (lib/zpack_read.c:380, :414-439), and the sequence fits the reference's own bound for the entry.
 *                                   Default 2 MiB; 0 = never split.
 *   ZPK_OPT_DEC_SPLIT_MIN           zpk_codec_decode_batch_host: an entry of at least `value` bytes that IS such a sequence (>= 2 frames
 *                                   that tile it exactly, each stating its content size; or a stored entry) is decoded with one wave
 *                                   per FRAME and hashed by the whole chip; any other entry, and any entry a frame of which fails,
 *                                   is decoded by one wave as before (verdicts come from there only).  Default 2 MiB; 0 = never.
 *   ZPK_OPT_ORDER_MIN               decode batches of at least `value` entries run their Zstandard and LZ4 work lists LARGEST ENTRIES
 *                                   FIRST (size classes by powers of two; a device counting sort behind the classification): one
 *                                   wave works on one entry, so a large entry that starts last runs on alone; encode batches order
 *                                   their ticket queue the same way.  Default 8192 (decode), 4608 (encode); 0 = never. */
/* (2..5 were the opt-in two-stage LZ4 path of round 4 and its measurement aids: measured slower than the one-kernel decoder, removed in round 5) */
enum { ZPK_OPT_ENC_SPLIT_MIN = 6, ZPK_OPT_DEC_SPLIT_MIN = 7, ZPK_OPT_ORDER_MIN = 8, ZPK_OPT_ORDER_FAST_LAST = 9 /* a batch of ONE size class runs the entries that did not compress (a copy to decode) last: 1 (default) / 0 */ };
int         zpk_codec_set_option(zpk_codec* c, int option, int value);

/* ---- batch decode + verify ----------------------------------------------------------------
 * Device-resident form: every pointer is a DEVICE pointer on the codec's device; work is enqueued on
 * `stream` (a hipStream_t passed as void*, NULL = the codec's own stream) and the call returns without
 * synchronising.  `src` is the archive image (or any packed frame stream), `n` descriptors. */
int zpk_codec_decode_batch_device(zpk_codec* c, const uint8_t* src, uint64_t src_size,
                                  const zpk_decode_desc* desc, uint64_t n,
                                  uint8_t* dst, uint64_t dst_size,
                                  zpk_decode_result* results, void* stream);

/* ONE entry whose compressed bytes are in DEVICE memory, decoded into device memory (d_dst + desc->dst_offset); desc and result are host
 * memory; returns when the entry is decoded and verified.  A large entry that is one frame of the reference writer
 * (lib/zpack_write.c:179, :204-210) is decoded block-parallel — 13-17 GiB/s instead of one wave's 0.03-0.14; anything else runs through the
 * kernels of zpk_codec_decode_batch_device.  Same verdicts either way (what zpack_read_file would return, lib/zpack_read.c:326-471). */
int  zpk_codec_decode_big_device(zpk_codec* c, const uint8_t* d_archive, uint64_t archive_size, const zpk_decode_desc* desc,
                                 uint8_t* d_dst, uint64_t dst_size, zpk_decode_result* result);

/* Host form: host pointers, synchronous.  Stages [min src_offset, max src_offset+comp_size) of
 * `archive` to the device, decodes, and copies each slot back to dst_ptrs[i] (desc[i].dst_offset is
 * ignored; desc[i].dst_capacity is the size of dst_ptrs[i]).  zpack_read_file is a batch of one. */
int zpk_codec_decode_batch_host(zpk_codec* c, const uint8_t* archive, uint64_t archive_size,
                                const zpk_decode_desc* desc, uint64_t n,
                                uint8_t* const* dst_ptrs, zpk_decode_result* results);

/* ---- batch encode + hash ------------------------------------------------------------------ */
size_t zpk_codec_compress_bound(uint32_t method, size_t src_size);
int zpk_codec_encode_batch_device(zpk_codec* c, const uint8_t* src, uint64_t src_size,
                                  const zpk_encode_desc* desc, uint64_t n,
                                  uint8_t* dst, uint64_t dst_size,
                                  zpk_encode_result* results, void* stream);
int zpk_codec_encode_batch_host(zpk_codec* c, const uint8_t* const* src_ptrs,
                                const zpk_encode_desc* desc, uint64_t n,
                                uint8_t* const* dst_ptrs, zpk_encode_result* results);

/* ---- compressed-size scan + compaction (the serial `write_offset += comp_size` of lib/zpack_write.c:338, for a batch)
 * After an encode batch: offsets[i] = start of entry i's payload in the packed stream (exclusive prefix sum of the
 * comp_size of the entries with status 0; failed entries take no room), offsets[n] = its total length; when `packed`
 * is not NULL the payloads are gathered there back-to-back from their slots (`slots` + desc[i].dst_offset), i.e. the
 * archive's data section in CDR order.  Device pointers, asynchronous on `stream`.  `max_entry_size` = an upper bound
 * of any comp_size (e.g. the largest dst_capacity); it only sizes the launch.  If packed_cap < offsets[n] the bytes
 * that do not fit are not written (check offsets[n]). */
int zpk_codec_pack_batch_device(zpk_codec* c, const uint8_t* slots, const zpk_encode_desc* desc,
                                const zpk_encode_result* results, uint64_t n,
                                uint8_t* packed, uint64_t packed_cap, uint64_t* offsets,
                                uint64_t max_entry_size, void* stream);

/* ---- hash only ---------------------------------------------------------------------------- */
/* hashes[i] = XXH3_64bits(src + offsets[i], sizes[i]); device pointers, async on stream */
int zpk_codec_hash_batch_device(zpk_codec* c, const uint8_t* src, const uint64_t* offsets,
                                const uint64_t* sizes, uint64_t n, uint64_t* hashes, void* stream);
int zpk_codec_hash_host(zpk_codec* c, const uint8_t* data, uint64_t size, uint64_t* hash);

/* ---- timing helper for benchmarks: elapsed device time of everything enqueued between the two
 * marks on `stream`, measured with HIP events on that stream ------------------------------- */
/* per-kernel timing of decode batches: when enabled, every decode batch brackets each of its kernels
 * with HIP events on the launch stream; zpk_codec_kernel_ms then returns the duration of kernel
 * `which` (ZPK_K_*) in the most recent batch (synchronises on that batch). */
enum { ZPK_K_CLASSIFY = 0, ZPK_K_STORED = 1, ZPK_K_LZ4 = 2 /* k_lz4_wave + k_lz4_left + k_lz4_retry */, ZPK_K_ZSTD = 3 /* k_zstd_exec + k_zstd */, ZPK_K_ZSTD_FSE = 4,
       ZPK_K_PACK = 5, ZPK_K_RESERVED6 = 6 /* (was k_lz4_parse) */, ZPK_K_ENCODE = 7, ZPK_K_COUNT = 8 };
int zpk_codec_set_profiling(zpk_codec* c, int enabled);
/* out[0], out[1] = LZ4 / Zstandard entries of the most recent decode batch whose first decode ran out of its time budget (a
 * contended or preempted GPU) and that were decoded again, behind the batch, with a 64 x larger one — a slow wave is not a
 * verdict; expected 0 on an idle GPU.  out[3] = LZ4 entries that are mostly runs (compressed to less than 1/8: the classification puts
 * them on the list of k_lz4_left, the build of the one-wave decoder with grouped cooperative copies); out[2] = out[4] = 0; out[5], out[6] = entries of the most recent
 * zpk_codec_decode_batch_host call that were decoded frame-parallel (ZPK_OPT_DEC_SPLIT_MIN) and the frames they had */
int zpk_codec_decode_stats2(zpk_codec* c, uint32_t out[16]);
/* counters of the most recent decode batch (synchronises): out[0..2] = entries on the stored / zstd / lz4 work
 * lists, out[3] = Zstandard entries finished on pre-decoded sequences (two-stage path), out[4] = by the fused decoder,
 * out[5], out[6] = entries / waves the pre-decode kernel gave up on (watchdog; expected 0) */
int zpk_codec_decode_stats(zpk_codec* c, uint32_t out[8]);
int zpk_codec_debug_read(zpk_codec* c, void* host, uint64_t bytes);   /* developer aid: phase timing words */
/* developer aid: read back the Zstandard pre-decode arena (what 0) / per-entry marks (what 1) of the last batch */
int zpk_codec_debug_fetch(zpk_codec* c, int what, uint64_t offset, void* host, uint64_t bytes);
int zpk_codec_kernel_ms(zpk_codec* c, int which, float* ms);
int zpk_codec_timer_start(zpk_codec* c, void* stream);
int zpk_codec_timer_stop(zpk_codec* c, void* stream, float* elapsed_ms);   /* synchronises the stop event */

/* ---- streaming triple (lib/zpack_read.c:515-640, lib/zpack_write.c:461-685) ------------------
 * Chunk granularity is hostile to the GPU, so a stream GATHERS small chunks on the host (256 KiB) before it steps the device:
 * every step decodes all complete blocks the bytes so far hold (resumable) and the output is handed out in avail_out-sized
 * pieces while more input arrives.  Same observable protocol as the reference (total_in/total_out/read_back). */
typedef struct zpk_dstream zpk_dstream;
int  zpk_dstream_create(zpk_codec* c, zpk_dstream** out);
void zpk_dstream_bind(zpk_dstream* s, zpk_codec* c);     /* the codec the NEXT step decodes with (a stream may outlive codecs) */
void zpk_dstream_reset(zpk_dstream* s);
void zpk_dstream_destroy(zpk_dstream* s);
/* feed in_size bytes, receive up to out_cap bytes; *consumed / *produced report progress;
 * entry_* describe the entry being streamed; *done = 1 once every output byte has been delivered and
 * the hash verified.  Returns an enum zpack_result code. */
int  zpk_dstream_step(zpk_dstream* s, uint32_t method, uint64_t entry_comp_size, uint64_t entry_uncomp_size,
                      uint64_t entry_hash, const uint8_t* in, size_t in_size, size_t* consumed,
                      uint8_t* out, size_t out_cap, size_t* produced, int* done);

/* Bounded memory (round 5): a large entry that is one plain LZ4 frame is decoded in block-parallel steps and the stream holds only a window
 * of it.  zpk_dstream_wants_input() == 0: the stream has output waiting and takes no input — call zpk_dstream_step with in_size 0.
 * zpk_dstream_step may answer ZPK_DS_RESTART (not a zpack_result): the entry is not what the steps can decide; feed the entry's bytes
 * [0, bytes given so far) again through zpk_dstream_replay (first = 1 with the first piece) and go on with zpk_dstream_step: the
 * stream continues in its windowless form, where the verdicts are; nothing is handed out twice. */
#define ZPK_DS_RESTART 1001
int  zpk_dstream_wants_input(const zpk_dstream* s);
int  zpk_dstream_replay(zpk_dstream* s, uint32_t method, uint64_t entry_comp_size, uint64_t entry_uncomp_size, uint64_t entry_hash,
                        const uint8_t* in, size_t in_size, int first);

/* diagnostics: device decode steps launched / calls served since the stream's last reset (small chunks are gathered on the host: at
 * most one launch per 256 KiB of input) */
void zpk_dstream_counters(const zpk_dstream* s, uint64_t* launches, uint64_t* calls);
uint64_t zpk_dstream_device_bytes(const zpk_dstream* s);   /* diagnostics: device memory the stream holds right now */

typedef struct zpk_cstream zpk_cstream;
int  zpk_cstream_create(zpk_codec* c, zpk_cstream** out);
void zpk_cstream_bind(zpk_cstream* s, zpk_codec* c);
void zpk_cstream_reset(zpk_cstream* s);
void zpk_cstream_destroy(zpk_cstream* s);
/* method + level of the entry about to be streamed (before its first update): from then on zpk_cstream_update compresses as the
 * plaintext arrives — every complete 512 KiB piece becomes a frame of its own, handed out by zpk_cstream_drain — and the device
 * holds a few MiB of plaintext at most.  Without it update only collects and zpk_cstream_finish compresses everything. */
int  zpk_cstream_configure(zpk_cstream* s, uint32_t method, int32_t level);
int  zpk_cstream_update(zpk_cstream* s, const uint8_t* in, size_t in_size);       /* plaintext in; may make output available */
/* compress what is left; *comp_size = all compressed bytes of the entry (handed out already or still to drain), *hash = XXH3-64 of the
 * whole plaintext; the remaining bytes are then drained with zpk_cstream_drain */
int  zpk_cstream_finish(zpk_cstream* s, uint32_t method, int32_t level, uint64_t* comp_size,
                        uint64_t* uncomp_size, uint64_t* hash);
size_t zpk_cstream_drain(zpk_cstream* s, uint8_t* out, size_t out_cap);           /* returns bytes copied */

#ifdef __cplusplus
}
#endif
#endif /* ZPACK_CODEC_H */
