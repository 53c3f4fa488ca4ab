/* internal.h — shared by the host-side zpack.h implementation (reader.c, writer.c, stream.c, util.c).
 * Nothing here is part of the public API. */
#ifndef ZPACK_AMD_INTERNAL_H
#define ZPACK_AMD_INTERNAL_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include "zpack.h"
#include "zpack_codec.h"

/* little-endian field access (the format is LE everywhere: docs/specs.md) */
static inline zpack_u16 zi_get16(const zpack_u8* p) { return (zpack_u16)(p[0] | (p[1] << 8)); }
static inline zpack_u32 zi_get32(const zpack_u8* p) { return (zpack_u32)p[0] | ((zpack_u32)p[1] << 8) | ((zpack_u32)p[2] << 16) | ((zpack_u32)p[3] << 24); }
static inline zpack_u64 zi_get64(const zpack_u8* p) { return (zpack_u64)zi_get32(p) | ((zpack_u64)zi_get32(p + 4) << 32); }
static inline void zi_put16(zpack_u8* p, zpack_u16 v) { p[0] = (zpack_u8)v; p[1] = (zpack_u8)(v >> 8); }
static inline void zi_put32(zpack_u8* p, zpack_u32 v) { for (int i = 0; i < 4; i++) p[i] = (zpack_u8)(v >> (8 * i)); }
static inline void zi_put64(zpack_u8* p, zpack_u64 v) { for (int i = 0; i < 8; i++) p[i] = (zpack_u8)(v >> (8 * i)); }

/* 64-bit clean file positioning (the reference maps these per platform, lib/zpack_common.h:37-78) */
#define zi_fseek(fp, off, whence) fseeko((fp), (off_t)(off), (whence))
#define zi_ftell(fp) ((zpack_u64)ftello(fp))

/* a context = the device codecs a reader / writer / explicit dctx / cctx works with (util.c) */
#define ZI_MAX_DEVICES 16
typedef struct zi_ctx_s { int n; zpk_codec* dev[ZI_MAX_DEVICES]; } zi_ctx;
zi_ctx* zi_ctx_create(void);                 /* NULL when no HIP device is usable: there is no CPU fallback */
void    zi_ctx_destroy(zi_ctx* x);
void    zi_ctx_reset(zi_ctx* x);
/* resolve the context of a call: the explicit one, else the owner's (created on first use, freed by zpack_close_*) */
zi_ctx* zi_pick_ctx(void* explicit_ctx, void** owner_slot);
/* contiguous split of `count` weighted items into `parts` ranges: cut[0..parts] */
void zi_split(const zpack_u64* w, zpack_u64 count, int parts, zpack_u64* cut);
void zi_parallel(int parts, void (*fn)(void*, int), void* arg);
/* name index of reader-owned entry tables (zpack_get_file_entry) */
void zi_index_register(const zpack_file_entry* table, zpack_u64 count);
void zi_index_drop(const zpack_file_entry* table);

/* per-stream aggregation state hung off zpack_stream.xxh3_state */
typedef struct zi_stream_state_s {
    zpk_dstream* d;
    zpk_cstream* c;
    int          d_active, c_active;
    zpack_u64    entry_offset;       /* streaming write: archive offset of the entry being written */
} zi_stream_state;

/* the writer's unified sink: file or growing heap buffer, at the writer's cursor */
int zi_writer_put(zpack_writer* w, const zpack_u8* data, size_t size);
zpack_file_entry* zi_writer_push_entry(zpack_writer* w);

#endif
