/* stream_copy.c — the reference's `io::copy` protocol on the stream ABI, in C (bench / test infrastructure, not product).
 *
 * The canonical libflate caller (examples/flate.rs:52,96-97; flate_bench/src/main.rs:86-100) is
 *     io::copy(&mut input, &mut Encoder::new(sink))  /  io::copy(&mut Decoder::new(input), &mut sink)
 * i.e. `write()` calls of 8192 bytes into the encoder and `read()` calls of 8192 bytes out of the decoder.  Driving 2 x 32768
 * such calls through Python callbacks would time the interpreter; this file is the same loop in C over include/lfx.h:
 * lfx_encoder_write in `chunk`-byte calls with a sink that appends to a memory buffer (like BenchWriter,
 * flate_bench/src/main.rs:102-125, but keeping the bytes), then lfx_decoder_read in `chunk`-byte calls over a reader that
 * serves the compressed bytes from memory (a `Cursor`: as many bytes as asked for).
 *
 * build: gcc -O2 -shared -fPIC stream_copy.c -L../libflate_amd -llfx  (tools/stream_copy.py does it) */
#include <stdint.h>
#include <string.h>
#include <time.h>

#include "../include/lfx.h"

typedef struct { uint8_t *p; size_t cap, n; int overflow; } sink_t;
static int64_t sink_write(void *user, const uint8_t *p, size_t n) {
    sink_t *s = (sink_t *)user;
    if (s->n + n > s->cap) { s->overflow = 1; return -5; }
    memcpy(s->p + s->n, p, n);
    s->n += n;
    return (int64_t)n;
}
static int sink_flush(void *user) { (void)user; return 0; }

typedef struct { const uint8_t *p; size_t n, pos; } cursor_t;
static int64_t cursor_read(void *user, uint8_t *out, size_t cap) {
    cursor_t *c = (cursor_t *)user;
    size_t k = c->n - c->pos;
    if (k > cap) k = cap;
    memcpy(out, c->p + c->pos, k);
    c->pos += k;
    return (int64_t)k;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* seconds of the last lfx_sc_encode / lfx_sc_decode spent in calls longer than 50 us — the write() that closes a batch and
 * runs it on the GPU, the read() that decodes the next window — and the number of such calls: the rest of the time is the
 * protocol's own copying (8 KiB memcpy calls into the encoder / out of the decoder, the sink, the cursor) */
static double g_slow_s;
static long g_slow_calls;
void lfx_sc_last_split(double *slow_s, long *slow_calls) { *slow_s = g_slow_s; *slow_calls = g_slow_calls; }

/* encode `in` with `chunk`-byte writes → enc[0, *enc_len); seconds in *t_enc.  → LFX status */
int lfx_sc_encode(lfx_ctx *c, int format, const lfx_encode_opts *o, const uint8_t *in, size_t n, size_t chunk, uint8_t *enc,
                  size_t enc_cap, size_t *enc_len, double *t_enc) {
    sink_t s = {enc, enc_cap, 0, 0};
    int st = 0;
    g_slow_s = 0;
    g_slow_calls = 0;
    const double t0 = now_s();
    lfx_encoder *e = lfx_encoder_new(c, format, o, sink_write, sink_flush, &s, &st);
    if (!e) return st ? st : LFX_E_DEVICE;
    for (size_t off = 0; off < n; off += chunk) {
        const size_t k = n - off < chunk ? n - off : chunk;
        const double a = now_s();
        const int64_t r = lfx_encoder_write(e, in + off, k);
        const double b = now_s() - a;
        if (b > 50e-6) { g_slow_s += b; g_slow_calls++; }
        if (r != (int64_t)k) { lfx_encoder_free(e); return r < 0 ? (int)-r : LFX_E_IO; }
    }
    {
        const double a = now_s();
        st = lfx_encoder_finish(e);
        g_slow_s += now_s() - a;
        g_slow_calls++;
    }
    lfx_encoder_free(e);
    *t_enc = now_s() - t0;
    *enc_len = s.n;
    return st;
}

/* decode enc with `chunk`-byte reads → dec[0, *dec_len); seconds in *t_dec.  → LFX status */
int lfx_sc_decode(lfx_ctx *c, int format, const uint8_t *enc, size_t enc_len, size_t chunk, uint8_t *dec, size_t dec_cap,
                  size_t *dec_len, double *t_dec) {
    cursor_t cur = {enc, enc_len, 0};
    int st = 0;
    g_slow_s = 0;
    g_slow_calls = 0;
    const double t0 = now_s();
    lfx_decoder *d = lfx_decoder_new(c, format, 0, cursor_read, &cur, &st);
    if (!d) return st ? st : LFX_E_DEVICE;
    size_t got = 0;
    for (;;) {
        const size_t k = dec_cap - got < chunk ? dec_cap - got : chunk;
        if (!k) break;
        const double a = now_s();
        const int64_t r = lfx_decoder_read(d, dec + got, k);
        const double b = now_s() - a;
        if (b > 50e-6) { g_slow_s += b; g_slow_calls++; }
        if (r < 0) { lfx_decoder_free(d); return (int)-r; }
        if (r == 0) break;
        got += (size_t)r;
    }
    lfx_decoder_free(d);
    *t_dec = now_s() - t0;
    *dec_len = got;
    return LFX_OK;
}

/* plain memcpy of n bytes in `chunk`-byte calls (what the protocol's copying alone costs on this host) → seconds */
double lfx_sc_memcpy(uint8_t *dst, const uint8_t *src, size_t n, size_t chunk) {
    const double t0 = now_s();
    for (size_t off = 0; off < n; off += chunk) memcpy(dst + off, src + off, n - off < chunk ? n - off : chunk);
    return now_s() - t0;
}
