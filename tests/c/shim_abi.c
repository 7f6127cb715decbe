/* shim_abi.c — drives, from plain C, every extern "C" function the Rust shim crate (rust/libflate-amd/src/ffi.rs)
 * declares, with the argument shapes the crate uses: callbacks over user pointers, header-first decoder construction,
 * header getters, surplus bytes, non-blocking reads, the Lz77Encode plug-in.  The expected bytes are the reference's
 * own known-answer vectors (src/deflate/encode.rs:152-154, src/zlib.rs:547-549, src/gzip.rs:800-802,1072-1083,
 * src/lz77.rs:16-32).  tests/test_abi.py keeps the two lists of functions in step.
 *
 * Without a GPU: checks that construction fails loudly with LFX_E_DEVICE (exit 0, prints "no device").
 * With a GPU: runs the vectors (exit 0 and "shim abi ok", or exit 1 with the failing check). */
#define _GNU_SOURCE   /* RTLD_DEFAULT */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/lfx.h"

#define CHECK(cond)                                                               \
    do {                                                                          \
        if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } \
    } while (0)

typedef struct { unsigned char b[1 << 16]; size_t n; int flushes; } sink_t;
static int64_t on_write(void *user, const uint8_t *p, size_t n) {
    sink_t *s = (sink_t *)user;
    if (s->n + n > sizeof s->b) return -5;
    memcpy(s->b + s->n, p, n);
    s->n += n;
    return (int64_t)n;
}
static int on_flush(void *user) { ((sink_t *)user)->flushes++; return 0; }

typedef struct { const unsigned char *p; size_t n, pos, step; int block_every; int calls; } src_t;
static int64_t on_read(void *user, uint8_t *out, size_t cap) {
    src_t *s = (src_t *)user;
    s->calls++;
    if (s->block_every && (s->calls % s->block_every) == 1) return -(int64_t)LFX_E_WOULD_BLOCK;
    size_t k = s->n - s->pos;
    if (k > cap) k = cap;
    if (s->step && k > s->step) k = s->step;
    memcpy(out, s->p + s->pos, k);
    s->pos += k;
    return (int64_t)k;
}

typedef struct { uint32_t w[64]; size_t n; } codes_t;
static void on_codes(void *user, const uint32_t *codes, size_t n) {
    codes_t *c = (codes_t *)user;
    for (size_t i = 0; i < n && c->n < 64; i++) c->w[c->n++] = codes[i];
}

static int read_all(lfx_decoder *d, unsigned char *out, size_t cap, size_t *got, int *blocks) {
    *got = 0;
    for (;;) {
        int64_t r = lfx_decoder_read(d, out + *got, cap - *got);
        if (r == -(int64_t)LFX_E_WOULD_BLOCK) { if (blocks) (*blocks)++; continue; }
        if (r < 0) return (int)-r;
        if (r == 0) return 0;
        *got += (size_t)r;
    }
}

/* device buffers for the sharded drivers: the HIP runtime liblfx.so already brought into the process */
typedef int (*hip_malloc_t)(void **, size_t);
typedef int (*hip_free_t)(void *);
typedef int (*hip_memcpy_t)(void *, const void *, size_t, int);
static hip_malloc_t p_hipMalloc; static hip_free_t p_hipFree; static hip_memcpy_t p_hipMemcpy;
static int hip_bind(void) {
    p_hipMalloc = (hip_malloc_t)dlsym(RTLD_DEFAULT, "hipMalloc");
    p_hipFree = (hip_free_t)dlsym(RTLD_DEFAULT, "hipFree");
    p_hipMemcpy = (hip_memcpy_t)dlsym(RTLD_DEFAULT, "hipMemcpy");
    return p_hipMalloc && p_hipFree && p_hipMemcpy;
}

static const unsigned char HELLO[] = "Hello World!";
static const unsigned char DEFLATE_HELLO[] = {5, 192, 49, 13, 0, 0, 8, 3, 65, 43, 224, 6, 7, 24, 128, 237, 147, 38, 245, 63, 244, 230, 65, 181, 50, 215, 1};
static const unsigned char ZLIB_HELLO[] = {120, 156, 5, 192, 49, 13, 0, 0, 8, 3, 65, 43, 224, 6, 7, 24, 128, 237, 147, 38, 245, 63, 244, 230, 65, 181, 50, 215, 1, 28, 73, 4, 62};
static const unsigned char GZIP_STORED[] = {31, 139, 8, 0, 123, 0, 0, 0, 0, 3, 1, 12, 0, 243, 255, 72, 101, 108, 108, 111, 32, 87, 111, 114, 108, 100, 33, 163, 28, 41, 28, 12, 0, 0, 0};
static const unsigned char MEMBER_A[] = {31, 139, 8, 0, 51, 206, 75, 90, 0, 3, 5, 128, 49, 9, 0, 0, 0, 194, 170, 24, 199, 34, 126, 3, 251, 127, 163, 131, 71, 192, 252, 45, 234, 6, 0, 0, 0};
static const unsigned char MEMBER_B[] = {31, 139, 8, 0, 227, 207, 75, 90, 0, 3, 5, 128, 49, 9, 0, 0, 0, 194, 178, 152, 202, 2, 158, 130, 96, 255, 99, 120, 111, 4, 222, 157, 40, 118, 6, 0, 0, 0};

int main(void) {
    CHECK(lfx_version() == LFX_VERSION);
    int st = -1;
    lfx_ctx *c = lfx_ctx_new(0, &st);
    if (!c) {
        CHECK(st == LFX_E_DEVICE);
        CHECK(lfx_device_count() == 0);
        printf("no device: lfx_ctx_new fails with LFX_E_DEVICE (no CPU fallback)\n");
        return 0;
    }
    CHECK(st == LFX_OK && lfx_device_count() >= 1);
    CHECK(lfx_ctx_last_error(c) != NULL);
    lfx_encode_opts o;

    /* ---- encoders: deflate / zlib / gzip (stored, mtime 123), write + flush + finish */
    {
        sink_t s = {{0}, 0, 0};
        lfx_encode_opts_default(&o);
        lfx_encoder *e = lfx_encoder_new(c, LFX_DEFLATE, &o, on_write, on_flush, &s, &st);
        CHECK(e && st == LFX_OK);
        CHECK(lfx_encoder_write(e, HELLO, 12) == 12);
        CHECK(lfx_encoder_finish(e) == LFX_OK);
        CHECK(lfx_encoder_last_error(e) != NULL);
        lfx_encoder_free(e);
        CHECK(s.n == sizeof DEFLATE_HELLO && !memcmp(s.b, DEFLATE_HELLO, s.n) && s.flushes >= 1);
    }
    {
        sink_t s = {{0}, 0, 0};
        lfx_encode_opts_default(&o);
        lfx_encoder *e = lfx_encoder_new(c, LFX_ZLIB, &o, on_write, NULL, &s, &st);
        CHECK(e && s.n == 2);                                   /* the header is written immediately (zlib.rs:577-585) */
        CHECK(lfx_encoder_write(e, HELLO, 5) == 5 && lfx_encoder_write(e, HELLO + 5, 7) == 7);
        CHECK(lfx_encoder_finish(e) == LFX_OK);
        lfx_encoder_free(e);
        CHECK(s.n == sizeof ZLIB_HELLO && !memcmp(s.b, ZLIB_HELLO, s.n));
    }
    {
        sink_t s = {{0}, 0, 0};
        lfx_encode_opts_default(&o);
        o.no_compression = 1;
        o.mtime = 123;
        lfx_encoder *e = lfx_encoder_new(c, LFX_GZIP, &o, on_write, on_flush, &s, &st);
        CHECK(e && s.n == 10);
        CHECK(lfx_encoder_write(e, HELLO, 12) == 12);
        {   /* the same options without a flush: the reference's bytes (gzip.rs:800-802) */
            sink_t s2 = {{0}, 0, 0};
            lfx_encoder *e2 = lfx_encoder_new(c, LFX_GZIP, &o, on_write, on_flush, &s2, &st);
            CHECK(e2 && lfx_encoder_write(e2, HELLO, 12) == 12 && lfx_encoder_finish(e2) == LFX_OK);
            lfx_encoder_free(e2);
            CHECK(s2.n == sizeof GZIP_STORED && !memcmp(s2.b, GZIP_STORED, s2.n));
        }
        CHECK(lfx_encoder_flush(e) == LFX_OK);                  /* closes a (here: stored) block */
        CHECK(lfx_encoder_finish(e) == LFX_OK);
        lfx_encoder_free(e);
        /* flush() before finish() adds an empty final stored block: decode instead of comparing bytes */
        src_t r = {s.b, s.n, 0, 0, 0, 0};
        lfx_decoder *d = lfx_decoder_new(c, LFX_GZIP, 0, on_read, &r, &st);
        CHECK(d && st == LFX_OK);
        unsigned char out[64]; size_t got;
        CHECK(read_all(d, out, sizeof out, &got, NULL) == 0 && got == 12 && !memcmp(out, HELLO, 12));
        lfx_decoder_free(d);
    }
    /* ---- a caller-side Lz77Encode (EncodeOptions::with_lz77(E), encode.rs:59-65): the shim's Lz77Stage hands E's code words
     *      to lfx_encoder_write_codes.  E = "every byte a literal, CompressionLevel::None": the stream must equal the one the
     *      device pipeline makes for NoCompressionLz77Encoder (lib.rs:111-145) */
    {
        sink_t s = {{0}, 0, 0}, s2 = {{0}, 0, 0};
        lfx_encode_opts_default(&o);
        o.lz77_level = 1 + LFX_LEVEL_NONE;
        lfx_encoder *e = lfx_encoder_new(c, LFX_ZLIB, &o, on_write, on_flush, &s, &st);
        CHECK(e && st == LFX_OK);
        uint32_t codes[12];
        for (int i = 0; i < 12; i++) codes[i] = (uint32_t)HELLO[i] << 16;
        CHECK(lfx_encoder_write_codes(e, codes, 5, HELLO, 5, 0) == LFX_OK);          /* write("Hello"): E::encode's codes */
        CHECK(lfx_encoder_write_codes(e, codes + 5, 7, HELLO + 5, 7, 0) == LFX_OK);
        CHECK(lfx_encoder_write_codes(e, NULL, 0, NULL, 0, 2) == LFX_OK);            /* finish(): E::flush's codes (none), final block */
        CHECK(lfx_encoder_finish(e) == LFX_OK);
        lfx_encoder_free(e);
        lfx_encode_opts_default(&o);
        o.lz77_kind = LFX_LZ77_NOCOMPRESSION;
        lfx_encoder *e2 = lfx_encoder_new(c, LFX_ZLIB, &o, on_write, on_flush, &s2, &st);
        CHECK(e2 && lfx_encoder_write(e2, HELLO, 5) == 5 && lfx_encoder_write(e2, HELLO + 5, 7) == 7 && lfx_encoder_finish(e2) == LFX_OK);
        lfx_encoder_free(e2);
        CHECK(s.n == s2.n && s.n > 6 && !memcmp(s.b, s2.b, s.n));
    }
    /* ---- decoder: header first, getters, one member of two, surplus, consumed */
    {
        unsigned char both[sizeof MEMBER_A + sizeof MEMBER_B];
        memcpy(both, MEMBER_A, sizeof MEMBER_A);
        memcpy(both + sizeof MEMBER_A, MEMBER_B, sizeof MEMBER_B);
        src_t r = {both, sizeof both, 0, 0, 0, 0};
        lfx_decoder *d = lfx_decoder_new(c, LFX_GZIP, 0, on_read, &r, &st);
        CHECK(d);
        lfx_header h;
        CHECK(lfx_decoder_header(d, &h) == LFX_OK && h.format == LFX_GZIP && h.mtime == 0x5A4BCE33u && h.os == 3 && !h.filename);
        unsigned char out[64]; size_t got;
        CHECK(lfx_decoder_read(d, out, 0) == 0);                /* a zero-capacity read never latches EOS (gzip.rs:1025) */
        CHECK(read_all(d, out, sizeof out, &got, NULL) == 0 && got == 6 && !memcmp(out, "Hello ", 6));
        CHECK(lfx_decoder_consumed(d) == sizeof MEMBER_A);
        CHECK(lfx_decoder_buffered(d) >= sizeof MEMBER_B && lfx_decoder_buffered(d) < (1u << 20));   /* the surplus, a read chunk, history */
        const uint8_t *sp; size_t sn;
        CHECK(lfx_decoder_surplus(d, &sp, &sn) == LFX_OK && sn == sizeof MEMBER_B && !memcmp(sp, MEMBER_B, sn));
        CHECK(lfx_decoder_unread(d, &sp, &sn) == LFX_OK && sn == 0);
        lfx_decoder_free(d);
        /* MultiDecoder over a reader that hands out 5 bytes at a time */
        src_t r2 = {both, sizeof both, 0, 5, 0, 0};
        d = lfx_decoder_new(c, LFX_GZIP, LFX_DEC_MULTI, on_read, &r2, &st);
        CHECK(d);
        CHECK(read_all(d, out, sizeof out, &got, NULL) == 0 && got == 12 && !memcmp(out, HELLO, 12));
        CHECK(lfx_decoder_consumed(d) == sizeof both);
        lfx_decoder_free(d);
        /* non-blocking: WouldBlock on every other call, nothing read by the constructor */
        src_t r3 = {both, sizeof both, 0, 7, 2, 0};
        d = lfx_decoder_new(c, LFX_GZIP, LFX_DEC_MULTI | LFX_DEC_NONBLOCKING, on_read, &r3, &st);
        CHECK(d && r3.calls == 0);
        int blocks = 0;
        CHECK(read_all(d, out, sizeof out, &got, &blocks) == 0 && got == 12 && blocks > 0);
        lfx_decoder_free(d);
        /* a header error surfaces in the constructor (gzip.rs:941-944), with the reference's message */
        both[0] = 30;
        src_t r4 = {both, sizeof both, 0, 0, 0, 0};
        d = lfx_decoder_new(c, LFX_GZIP, 0, on_read, &r4, &st);
        CHECK(!d && st == LFX_E_INVALID_DATA && strstr(lfx_ctx_last_error(c), "Unexpected GZIP ID"));
    }
    /* ---- decode error text after the bytes in front of it (zlib.rs:916-934 style) */
    {
        src_t r = {ZLIB_HELLO, sizeof ZLIB_HELLO - 1, 0, 0, 0, 0};
        lfx_decoder *d = lfx_decoder_new(c, LFX_ZLIB, 0, on_read, &r, &st);
        CHECK(d);
        lfx_header h;
        CHECK(lfx_decoder_header(d, &h) == LFX_OK && h.zlib_window_size == 32768 && h.zlib_level == 2);
        unsigned char out[64]; size_t got;
        CHECK(read_all(d, out, sizeof out, &got, NULL) == LFX_E_UNEXPECTED_EOF);
        CHECK(lfx_decoder_last_error(d)[0] != 0);
        lfx_decoder_free(d);
    }
    /* ---- Lz77Encode plug-in (src/lz77.rs:16-32) */
    {
        lfx_lz77 *z = lfx_lz77_new(c, 32768, 258, &st);
        CHECK(z && st == LFX_OK);
        CHECK(lfx_lz77_window_size(z) == 32768 && lfx_lz77_compression_level(z) == LFX_LEVEL_BALANCE);
        codes_t cs = {{0}, 0};
        CHECK(lfx_lz77_encode(z, (const uint8_t *)"aaaaa", 5, on_codes, &cs) == LFX_OK && cs.n == 0);   /* buffers */
        CHECK(lfx_lz77_flush(z, on_codes, &cs) == LFX_OK);
        CHECK(cs.n == 2 && cs.w[0] == (97u << 16) && cs.w[1] == ((4u << 16) | 1u));
        lfx_lz77_free(z);
    }
    /* ---- the N-GPU drivers at world size 1 (rust/libflate-amd/src/sharded.rs): the sharded encode must give the bytes ONE
     *      encoder gives (src/gzip.rs:858-868: one trailer from one checksum), the member decode by byte ranges must give the
     *      input back with the trailer's CRC-32 */
    {
        CHECK(hip_bind());
        const size_t n = 3u << 20;
        lfx_encode_opts o;
        lfx_encode_opts_default(&o);
        o.mtime = 7;
        lfx_schedule sc = {LFX_SCHED_FIXED, 8192, NULL, 0};
        const uint64_t bound = lfx_encode_bound(n, &o, &sc) & ~3ull;
        CHECK(bound > n / 2);
        unsigned char *in = malloc(n), *one = malloc(bound), *mem = malloc(bound), *back = malloc(n);
        CHECK(in && one && mem && back);
        unsigned x = 12345;
        for (size_t i = 0; i < n; i++) {                          /* text-like: words of a small vocabulary */
            x = x * 1664525u + 1013904223u;
            in[i] = (i % 7 == 6) ? ' ' : (unsigned char)('a' + ((x >> 24) % 9) + ((i / 4096) % 3));
        }
        uint64_t one_len = 0;
        CHECK(lfx_encode_host(c, LFX_GZIP, &o, &sc, in, n, one, bound, &one_len) == LFX_OK);
        /* page-locked buffers (lfx_host_alloc; rust: HostBuf) through the same call: the same bytes (3 MiB is above the size
         * from which pageable memory is staged by the copy threads, so `one` above took that path and this one plain DMA) */
        {
            unsigned char *pin_in = lfx_host_alloc(n), *pin_out = lfx_host_alloc(bound);
            CHECK(pin_in && pin_out);
            memcpy(pin_in, in, n);
            uint64_t pl = 0;
            CHECK(lfx_encode_host(c, LFX_GZIP, &o, &sc, pin_in, n, pin_out, bound, &pl) == LFX_OK);
            CHECK(pl == one_len && !memcmp(pin_out, one, one_len));
            lfx_host_free(pin_in);
            lfx_host_free(pin_out);
            lfx_host_free(NULL);
            CHECK(lfx_ctx_match_fallbacks(c) == 0);
        }
        void *d_in, *d_part, *d_member, *d_out;
        CHECK(!p_hipMalloc(&d_in, n) && !p_hipMalloc(&d_part, bound) && !p_hipMalloc(&d_member, bound) && !p_hipMalloc(&d_out, n));
        CHECK(!p_hipMemcpy(d_in, in, n, 1 /* hipMemcpyHostToDevice */));
        lfx_comm cm = {NULL, 0, 1, NULL, NULL, NULL, NULL, NULL};
        lfx_sharded_enc *stt = NULL;
        lfx_sharded_part part;
        CHECK(lfx_sharded_encode_begin(c, &cm, LFX_GZIP, &o, &sc, d_in, n, d_part, bound, d_member, bound, NULL, 0, &stt, &part) == LFX_OK);
        CHECK(stt && part.start_bit == 8 * lfx_container_header_len(LFX_GZIP, &o) && part.total_n == n && part.member_len == one_len);
        uint64_t mlen = 0;
        CHECK(lfx_sharded_encode_finish(c, &cm, stt, &mlen) == LFX_OK && mlen == one_len);
        CHECK(!p_hipMemcpy(mem, d_member, mlen, 2 /* hipMemcpyDeviceToHost */) && !memcmp(mem, one, mlen));
        uint64_t lo, hi, hold;
        lfx_sharded_byte_range(10, mlen, 0, 1, &lo, &hi, &hold);
        CHECK(lo == 10 && hi == mlen && hold == mlen);
        lfx_sharded_slice sl;
        CHECK(lfx_sharded_decode(c, &cm, d_member, mlen, 0, mlen, part.start_bit, mlen, d_out, n, &sl) == LFX_OK);
        CHECK(sl.out_len == n && sl.out_base == 0 && sl.total_out == n && sl.crc32 == part.check);
        CHECK(!p_hipMemcpy(back, d_out, n, 2) && !memcmp(back, in, n));
        uint32_t tr;
        memcpy(&tr, mem + mlen - 8, 4);
        CHECK(tr == sl.crc32);
        /* the RCCL binding refuses a null communicator; freeing an unbound comm is a no-op */
        lfx_comm rc;
        memset(&rc, 0, sizeof rc);
        CHECK(lfx_comm_rccl(NULL, NULL, 0, 1, &rc) == LFX_E_ARG);
        lfx_comm_rccl_free(&rc);
        /* ---- the RCCL binding EXECUTED: a communicator of one rank on this GPU.  Every callback runs
         *      on real RCCL: the all-gather; a self send + receive inside ONE group (legal in NCCL), started by `start` and
         *      completed by `wait`; an all-gather issued while a group is still open (it has to close the group first — inside
         *      it the collective would only be queued and `recv` read too early); then both drivers over this comm (at one
         *      rank their all-gathers go through RCCL since round 6; the member must not change). */
        void *rccl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!rccl) rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!rccl) rccl = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (rccl && !getenv("LFX_SHIM_NO_RCCL")) {
            typedef struct { char internal[128]; } nccl_uid;
            int (*p_uid)(nccl_uid *) = (int (*)(nccl_uid *))dlsym(rccl, "ncclGetUniqueId");
            int (*p_init)(void **, int, nccl_uid, int) = (int (*)(void **, int, nccl_uid, int))dlsym(rccl, "ncclCommInitRank");
            int (*p_destroy)(void *) = (int (*)(void *))dlsym(rccl, "ncclCommDestroy");
            CHECK(p_uid && p_init && p_destroy);
            nccl_uid uid;
            void *ncomm = NULL;
            CHECK(p_uid(&uid) == 0);
            CHECK(p_init(&ncomm, 1, uid, 0) == 0 && ncomm);
            CHECK(lfx_comm_rccl(ncomm, NULL /* the null stream */, 0, 1, &rc) == LFX_OK);
            CHECK(rc.allgather && rc.isend && rc.irecv && rc.wait && rc.start && rc.world == 1 && rc.rank == 0);
            unsigned char sendb[40], recvb[40];
            for (int i = 0; i < 40; i++) sendb[i] = (unsigned char)(3 * i + 1);
            memset(recvb, 0, sizeof recvb);
            CHECK(rc.allgather(rc.user, sendb, recvb, 40) == 0 && !memcmp(sendb, recvb, 40));
            const size_t tn = 3u << 20;                                   /* a shard-sized transfer to self */
            void *d_a, *d_b;
            CHECK(!p_hipMalloc(&d_a, tn) && !p_hipMalloc(&d_b, tn));
            CHECK(!p_hipMemcpy(d_a, in, tn, 1) && !p_hipMemcpy(d_b, back, tn, 1));
            CHECK(rc.irecv(rc.user, d_b, tn, 0) == 0 && rc.isend(rc.user, d_a, tn, 0) == 0);
            CHECK(rc.start(rc.user) == 0);
            CHECK(rc.wait(rc.user) == 0);
            unsigned char *chk = malloc(tn);
            CHECK(chk && !p_hipMemcpy(chk, d_b, tn, 2) && !memcmp(chk, in, tn));
            /* posted but not started, then an all-gather: the group is closed first, the gather's result is real, and the
             * posted transfer completes by wait() */
            CHECK(!p_hipMemcpy(d_b, one, tn < one_len ? tn : one_len, 1));
            CHECK(rc.irecv(rc.user, d_b, tn, 0) == 0 && rc.isend(rc.user, d_a, tn, 0) == 0);
            memset(recvb, 0, sizeof recvb);
            CHECK(rc.allgather(rc.user, sendb, recvb, 40) == 0 && !memcmp(sendb, recvb, 40));
            CHECK(rc.wait(rc.user) == 0);
            CHECK(!p_hipMemcpy(chk, d_b, tn, 2) && !memcmp(chk, in, tn));
            free(chk);
            p_hipFree(d_a); p_hipFree(d_b);
            /* both drivers over the RCCL comm */
            lfx_sharded_enc *st2 = NULL;
            lfx_sharded_part part2;
            CHECK(!p_hipMemcpy(d_member, back, 16, 1));
            CHECK(lfx_sharded_encode_begin(c, &rc, LFX_GZIP, &o, &sc, d_in, n, d_part, bound, d_member, bound, NULL, 0, &st2, &part2) == LFX_OK);
            CHECK(st2 && part2.member_len == one_len && part2.check == part.check);
            uint64_t mlen2 = 0;
            CHECK(lfx_sharded_encode_finish(c, &rc, st2, &mlen2) == LFX_OK && mlen2 == one_len);
            CHECK(!p_hipMemcpy(mem, d_member, mlen2, 2) && !memcmp(mem, one, mlen2));
            memset(&sl, 0, sizeof sl);
            CHECK(lfx_sharded_decode(c, &rc, d_member, mlen2, 0, mlen2, part2.start_bit, mlen2, d_out, n, &sl) == LFX_OK);
            CHECK(sl.out_len == n && sl.total_out == n && sl.crc32 == part.check);
            CHECK(!p_hipMemcpy(back, d_out, n, 2) && !memcmp(back, in, n));
            /* a member buffer that is too small is refused by the exchange itself (every rank the same verdict) */
            st2 = NULL;
            CHECK(lfx_sharded_encode_begin(c, &rc, LFX_GZIP, &o, &sc, d_in, n, d_part, bound, d_member, one_len - 1, NULL, 0, &st2, &part2) == LFX_E_NOSPACE);
            CHECK(st2 == NULL);
            lfx_comm_rccl_free(&rc);
            CHECK(p_destroy(ncomm) == 0);
            printf("rccl binding ok (1 rank: all-gather, grouped self send/recv, start/wait, both drivers)\n");
        } else {
            printf("rccl binding skipped (librccl not loadable)\n");
        }
        p_hipFree(d_in); p_hipFree(d_part); p_hipFree(d_member); p_hipFree(d_out);
        free(in); free(one); free(mem); free(back);
    }
    lfx_ctx_free(c);
    printf("shim abi ok\n");
    return 0;
}
