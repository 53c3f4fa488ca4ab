/* stream.c — zpack_stream lifecycle.  The reference hangs an XXH3 streaming state off the stream
 * (lib/zpack_stream.c:4-28); here the slot holds the aggregation state of the GPU-backed streaming
 * calls (see zpk_dstream / zpk_cstream in zpack_codec.h). */
#include "internal.h"

int zpack_init_stream(zpack_stream* stream)
{
    if (!stream->xxh3_state) {
        zi_stream_state* st = (zi_stream_state*)calloc(1, sizeof(*st));
        if (!st) return ZPACK_ERROR_MALLOC_FAILED;
        stream->xxh3_state = st;
    }
    return ZPACK_OK;
}

void zpack_reset_stream(zpack_stream* stream)
{
    stream->total_in = 0;
    stream->total_out = 0;
    stream->read_back = 0;
    zi_stream_state* st = (zi_stream_state*)stream->xxh3_state;
    if (st) {
        if (st->d) zpk_dstream_reset(st->d);
        if (st->c) zpk_cstream_reset(st->c);
        st->d_active = st->c_active = 0;
    }
}

void zpack_close_stream(zpack_stream* stream)
{
    zi_stream_state* st = (zi_stream_state*)stream->xxh3_state;
    if (st) {
        zpk_dstream_destroy(st->d);
        zpk_cstream_destroy(st->c);
        free(st);
    }
    stream->xxh3_state = NULL;
}
