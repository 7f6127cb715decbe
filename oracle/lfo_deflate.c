/*
 * lfo_deflate.c — ORACLE (test infrastructure only; see lfo.h).
 * DEFLATE symbol coding, block loops, bit I/O and the gzip/zlib containers,
 * restated from the reference (citations inline).
 */
#include "lfo.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ buffers */
static void buf_reserve(lfo_buf *b, size_t extra) {
    if (b->n + extra > b->cap) {
        size_t c = b->cap ? b->cap * 2 : 256;
        while (c < b->n + extra) c *= 2;
        b->p = (uint8_t *)realloc(b->p, c);
        b->cap = c;
    }
}
static void buf_put(lfo_buf *b, const void *p, size_t n) {
    buf_reserve(b, n);
    if (n) memcpy(b->p + b->n, p, n);
    b->n += n;
}
static void buf_put1(lfo_buf *b, uint8_t v) { buf_put(b, &v, 1); }

/* ------------------------------------------------------------------ tables
 * src/deflate/symbol.rs:16-18, 22-52, 56-87 */
static const uint8_t CLEN_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
static const uint16_t LEN_BASE[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                      31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                      2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,
                                       33,  49,  65,  97,  129, 193,  257,  385,  513,  769,
                                       1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                       6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

/* Symbol::code / extra_lengh (symbol.rs:95-125) */
static inline uint16_t sym_len_code(uint16_t length, uint8_t *ebits, uint16_t *extra) {
    if (length <= 10) { *ebits = 0; *extra = 0; return (uint16_t)(257 + length - 3); }
    if (length <= 18) { *ebits = 1; *extra = (length - 11) % 2; return (uint16_t)(265 + (length - 11) / 2); }
    if (length <= 34) { *ebits = 2; *extra = (length - 19) % 4; return (uint16_t)(269 + (length - 19) / 4); }
    if (length <= 66) { *ebits = 3; *extra = (length - 35) % 8; return (uint16_t)(273 + (length - 35) / 8); }
    if (length <= 130) { *ebits = 4; *extra = (length - 67) % 16; return (uint16_t)(277 + (length - 67) / 16); }
    if (length <= 257) { *ebits = 5; *extra = (length - 131) % 32; return (uint16_t)(281 + (length - 131) / 32); }
    *ebits = 0; *extra = 0; return 285;
}
/* Symbol::distance (symbol.rs:126-154) */
static inline uint8_t sym_dist_code(uint16_t distance, uint8_t *ebits, uint16_t *extra) {
    if (distance <= 4) { *ebits = 0; *extra = 0; return (uint8_t)(distance - 1); }
    uint32_t extra_bits = 1, code = 4, base = 4;
    while (base * 2 < distance) { extra_bits++; code += 2; base *= 2; }
    uint32_t half = base / 2, delta = distance - base - 1;
    *ebits = (uint8_t)extra_bits;
    *extra = (uint16_t)(delta % half);
    return (uint8_t)(distance <= base + half ? code : code + 1);
}

/* ------------------------------------------------------------------ encoder */
#define SYM_EOB 0x01000000u /* val 256, dist 0 */

struct lfo_encoder {
    int format;
    lfo_opts o;
    lfo_buf out;
    uint32_t bw_buf; /* BitWriter bit.rs:4-8 */
    uint32_t bw_end;
    int block_type; /* deflate/mod.rs:34-39: Raw 0, Fixed 1, Dynamic 2 */
    size_t block_size;
    lfo_buf raw;   /* RawBuf encode.rs:354-383 */
    lfo_buf lzbuf; /* DefaultLz77Encoder.buf default.rs:18 */
    uint32_t *syms; /* CompressBuf.buf encode.rs:389 */
    size_t nsyms, symcap;
    size_t original_size;
    uint32_t crc, adler, isize;
    int finished;
};

static void bw_write_bits(lfo_encoder *e, uint32_t w, uint32_t bits) {
    /* bit.rs:25-31,42-49 */
    e->bw_buf |= bits << e->bw_end;
    e->bw_end += w;
    if (e->bw_end >= 16) {
        uint8_t b[2] = {(uint8_t)e->bw_buf, (uint8_t)(e->bw_buf >> 8)};
        buf_put(&e->out, b, 2);
        e->bw_end -= 16;
        e->bw_buf >>= 16;
    }
}
static void bw_flush(lfo_encoder *e) {
    /* bit.rs:32-40 */
    while (e->bw_end > 0) {
        buf_put1(&e->out, (uint8_t)e->bw_buf);
        e->bw_buf >>= 8;
        e->bw_end = e->bw_end > 8 ? e->bw_end - 8 : 0;
    }
}

static void syms_reserve(lfo_encoder *e, size_t extra) {
    if (e->nsyms + extra > e->symcap) {
        size_t c = e->symcap ? e->symcap * 2 : 1024;
        while (c < e->nsyms + extra) c *= 2;
        e->syms = (uint32_t *)realloc(e->syms, c * sizeof(uint32_t));
        e->symcap = c;
    }
}

/* DefaultLz77Encoder::flush → sink = Vec<Symbol> */
static void lz_custom(lfo_encoder *e, const uint8_t *p, size_t n) {
    /* a generic E: whatever it hands to the sink is appended to CompressBuf.buf (encode.rs:405-408,416) */
    const uint32_t *codes = NULL;
    size_t k = e->o.custom_cb(e->o.custom_user, p, n, &codes);
    syms_reserve(e, k);
    memcpy(e->syms + e->nsyms, codes, k * sizeof(uint32_t));
    e->nsyms += k;
}
static void lz_flush(lfo_encoder *e) {
    if (e->o.lz77_kind == LFO_LZ77_CUSTOM) { lz_custom(e, NULL, 0); return; }
    if (e->o.lz77_kind != LFO_LZ77_DEFAULT) return; /* lib.rs:136-141: no-op */
    syms_reserve(e, e->lzbuf.n + 1);
    e->nsyms += lfo_lz77_chunk(e->lzbuf.p, e->lzbuf.n, e->o.window_size, e->o.max_length,
                               e->syms + e->nsyms);
    e->lzbuf.n = 0; /* default.rs:108 */
}
/* DefaultLz77Encoder::encode default.rs:60-68 / NoCompressionLz77Encoder lib.rs:127-135 */
static void lz_encode(lfo_encoder *e, const uint8_t *p, size_t n) {
    if (e->o.lz77_kind == LFO_LZ77_CUSTOM) { lz_custom(e, p ? p : (const uint8_t *)"", n); return; }
    if (e->o.lz77_kind == LFO_LZ77_NOCOMPRESSION) {
        syms_reserve(e, n);
        for (size_t i = 0; i < n; i++) e->syms[e->nsyms++] = (uint32_t)p[i] << 16;
        return;
    }
    buf_put(&e->lzbuf, p, n);
    if (e->lzbuf.n >= (size_t)e->o.window_size * 8) lz_flush(e);
}

typedef struct {
    uint8_t lw[288];
    uint16_t lb[288];
    int ln; /* table length = last used symbol + 1 (huffman.rs:193-198) */
    uint8_t dw[30];
    uint16_t db[30];
    int dn;
} symenc;

static int table_len(const uint8_t *w, int n) {
    int last = 0; /* rfind(width>0).map_or(0, idx) + 1 */
    for (int i = n - 1; i >= 0; i--)
        if (w[i] > 0) { last = i; break; }
    return last + 1;
}

/* DynamicHuffmanCodec::build symbol.rs:321-342 */
static void dyn_build(const uint32_t *syms, size_t n, symenc *se) {
    size_t lc[286], dc[30];
    memset(lc, 0, sizeof lc);
    memset(dc, 0, sizeof dc);
    int empty = 1;
    for (size_t i = 0; i < n; i++) {
        uint32_t s = syms[i];
        uint16_t dist = (uint16_t)s, val = (uint16_t)(s >> 16);
        uint8_t eb;
        uint16_t ex;
        if (dist == 0) {
            lc[val]++; /* literal 0..255 or EOB 256 */
        } else {
            lc[sym_len_code(val, &eb, &ex)]++;
            empty = 0;
            dc[sym_dist_code(dist, &eb, &ex)]++;
        }
    }
    if (empty) dc[0] = 1; /* symbol.rs:332-337 */
    memset(se, 0, sizeof *se);
    lfo_huff_widths(lc, 286, 15, se->lw);
    lfo_huff_widths(dc, 30, 15, se->dw);
    lfo_huff_codes(se->lw, 286, se->lb);
    lfo_huff_codes(se->dw, 30, se->db);
    se->ln = table_len(se->lw, 286);
    se->dn = table_len(se->dw, 30);
}

/* FixedHuffmanCodec::build symbol.rs:258-280 */
static uint16_t rev_bits(uint16_t v, int w) {
    uint16_t t = 0;
    for (int i = 0; i < w; i++) { t = (uint16_t)((t << 1) | (v & 1)); v >>= 1; }
    return t;
}
static void fixed_build(symenc *se) {
    memset(se, 0, sizeof *se);
    for (int s = 0; s < 288; s++) {
        int w; uint16_t c;
        if (s < 144) { w = 8; c = (uint16_t)(0x30 + s); }
        else if (s < 256) { w = 9; c = (uint16_t)(0x190 + (s - 144)); }
        else if (s < 280) { w = 7; c = (uint16_t)(s - 256); }
        else { w = 8; c = (uint16_t)(0xC0 + (s - 280)); }
        se->lw[s] = (uint8_t)w;
        se->lb[s] = rev_bits(c, w);
    }
    for (int i = 0; i < 30; i++) { se->dw[i] = 5; se->db[i] = rev_bits((uint16_t)i, 5); }
    se->ln = 288;
    se->dn = 30;
}

/* build_bitwidth_codes symbol.rs:486-540 → triples (code, extra bits, extra) */
typedef struct { uint8_t c, b, x; } bwcode;
static size_t build_bitwidth_codes(const symenc *se, int nl, int nd, bwcode *codes) {
    struct { uint8_t v; size_t c; } runs[400];
    int nr = 0;
    for (int t = 0; t < 2; t++) {
        const uint8_t *w = t ? se->dw : se->lw;
        int size = t ? nd : nl;
        for (int i = 0; i < size; i++) {
            uint8_t c = w[i];
            if (i > 0 && nr > 0 && runs[nr - 1].v == c) runs[nr - 1].c++; /* :501-503 */
            else { runs[nr].v = c; runs[nr].c = 1; nr++; }
        }
    }
    size_t n = 0;
    for (int r = 0; r < nr; r++) {
        size_t c = runs[r].c;
        if (runs[r].v == 0) {
            while (c >= 11) {
                uint8_t k = (uint8_t)(c < 138 ? c : 138);
                codes[n++] = (bwcode){18, 7, (uint8_t)(k - 11)};
                c -= k;
            }
            if (c >= 3) { codes[n++] = (bwcode){17, 3, (uint8_t)(c - 3)}; c = 0; }
            for (; c > 0; c--) codes[n++] = (bwcode){0, 0, 0};
        } else {
            codes[n++] = (bwcode){runs[r].v, 0, 0};
            c -= 1;
            while (c >= 3) {
                uint8_t k = (uint8_t)(c < 6 ? c : 6);
                codes[n++] = (bwcode){16, 2, (uint8_t)(k - 3)};
                c -= k;
            }
            for (; c > 0; c--) codes[n++] = (bwcode){runs[r].v, 0, 0};
        }
    }
    return n;
}

/* DynamicHuffmanCodec::save symbol.rs:343-386 */
static void dyn_save(lfo_encoder *e, const symenc *se) {
    /* used_max_symbol (huffman.rs:247-253) = table_len-1 if any width>0 */
    int lit_used = -1, dist_used = -1;
    for (int i = se->ln - 1; i >= 0; i--) if (se->lw[i]) { lit_used = i; break; }
    for (int i = se->dn - 1; i >= 0; i--) if (se->dw[i]) { dist_used = i; break; }
    int nl = (lit_used < 0 ? 0 : lit_used) + 1; if (nl < 257) nl = 257;
    int nd = (dist_used < 0 ? 0 : dist_used) + 1; if (nd < 1) nd = 1;
    bwcode codes[400];
    size_t nc = build_bitwidth_codes(se, nl, nd, codes);
    size_t cc[19];
    memset(cc, 0, sizeof cc);
    for (size_t i = 0; i < nc; i++) cc[codes[i].c]++;
    uint8_t cw[19];
    uint16_t cb[19];
    lfo_huff_widths(cc, 19, 7, cw);
    lfo_huff_codes(cw, 19, cb);
    int bcc = 0; /* :357-364 */
    for (int k = 18; k >= 0; k--) {
        int i = CLEN_ORDER[k];
        if (cc[i] != 0 && cw[i] > 0) { bcc = k + 1; break; }
    }
    if (bcc < 4) bcc = 4;
    bw_write_bits(e, 5, (uint32_t)(nl - 257));
    bw_write_bits(e, 5, (uint32_t)(nd - 1));
    bw_write_bits(e, 4, (uint32_t)(bcc - 4));
    for (int k = 0; k < bcc; k++) {
        int i = CLEN_ORDER[k];
        bw_write_bits(e, 3, cc[i] == 0 ? 0 : cw[i]);
    }
    for (size_t i = 0; i < nc; i++) {
        bw_write_bits(e, cw[codes[i].c], cb[codes[i].c]);
        if (codes[i].b > 0) bw_write_bits(e, codes[i].b, codes[i].x);
    }
}

/* symbol::Encoder::encode symbol.rs:168-183 */
static void sym_encode(lfo_encoder *e, const symenc *se, uint32_t s) {
    uint16_t dist = (uint16_t)s, val = (uint16_t)(s >> 16);
    if (dist == 0) {
        bw_write_bits(e, se->lw[val], se->lb[val]);
        return;
    }
    uint8_t eb;
    uint16_t ex;
    uint16_t lc = sym_len_code(val, &eb, &ex);
    bw_write_bits(e, se->lw[lc], se->lb[lc]);
    if (eb) bw_write_bits(e, eb, ex);
    uint8_t dc = sym_dist_code(dist, &eb, &ex);
    bw_write_bits(e, se->dw[dc], se->db[dc]);
    if (eb > 0) bw_write_bits(e, eb, ex);
}

/* CompressBuf::flush encode.rs:412-425 */
static void compress_flush(lfo_encoder *e) {
    lz_flush(e);
    syms_reserve(e, 1);
    e->syms[e->nsyms++] = SYM_EOB;
    symenc se;
    if (e->block_type == 2) {
        dyn_build(e->syms, e->nsyms, &se);
        dyn_save(e, &se);
    } else {
        fixed_build(&se);
    }
    for (size_t i = 0; i < e->nsyms; i++) sym_encode(e, &se, e->syms[i]);
    e->nsyms = 0;
    e->original_size = 0;
}
/* RawBuf::flush encode.rs:364-382 */
static void raw_flush(lfo_encoder *e) {
    size_t size = e->raw.n < 0xFFFF ? e->raw.n : 0xFFFF;
    bw_flush(e);
    uint8_t h[4] = {(uint8_t)size, (uint8_t)(size >> 8), (uint8_t)~size, (uint8_t)(~size >> 8)};
    buf_put(&e->out, h, 4);
    buf_put(&e->out, e->raw.p, size);
    memmove(e->raw.p, e->raw.p + size, e->raw.n - size);
    e->raw.n -= size;
}
/* Block::flush encode.rs:287-295 */
static void block_flush(lfo_encoder *e, int is_final) {
    bw_write_bits(e, 1, (uint32_t)is_final);
    bw_write_bits(e, 2, (uint32_t)e->block_type);
    if (e->block_type == 0) raw_flush(e);
    else compress_flush(e);
}
static size_t block_len(lfo_encoder *e) { return e->block_type == 0 ? e->raw.n : e->original_size; }

void lfo_opts_default(lfo_opts *o) {
    memset(o, 0, sizeof *o);
    o->block_size = 1024 * 1024; /* encode.rs:11 */
    o->dynamic_huffman = 1;
    o->window_size = 32768;
    o->max_length = 258;
    o->os = 3; /* gzip.rs:158 */
}

/* gzip Header::write_to gzip.rs:368-389 (+flags 343-355, crc16 356-367) */
static void gzip_header_bytes(const lfo_opts *o, int with_hcrc, lfo_buf *b) {
    uint8_t flg = (uint8_t)((o->is_text ? 1 : 0) | (with_hcrc ? 2 : 0) | (o->extra ? 4 : 0) |
                            (o->filename ? 8 : 0) | (o->comment ? 16 : 0));
    /* XFL: lz77 CompressionLevel Balance/None → Unknown → 0, Fast → Fastest → 4, Best → Slowest → 2 (gzip.rs:69-92,684);
     * no_compression() resets it to Unknown (gzip.rs:703) */
    uint8_t xfl = 0;
    if (o->lz77_kind == LFO_LZ77_CUSTOM && !o->no_compression)
        xfl = o->custom_level == LFO_LEVEL_FAST ? 4 : o->custom_level == LFO_LEVEL_BEST ? 2 : 0;
    uint8_t h[10] = {31, 139, 8, flg, (uint8_t)o->mtime, (uint8_t)(o->mtime >> 8),
                     (uint8_t)(o->mtime >> 16), (uint8_t)(o->mtime >> 24), xfl, o->os};
    buf_put(b, h, 10);
    if (o->extra) {
        uint8_t l[2] = {(uint8_t)o->extra_len, (uint8_t)(o->extra_len >> 8)};
        buf_put(b, l, 2);
        buf_put(b, o->extra, o->extra_len);
    }
    if (o->filename) buf_put(b, o->filename, strlen(o->filename) + 1);
    if (o->comment) buf_put(b, o->comment, strlen(o->comment) + 1);
    if (with_hcrc) {
        /* crc16 = CRC-32 of the header serialised with is_verified=false (FLG.HCRC clear!) */
        lfo_buf t = {0, 0, 0};
        gzip_header_bytes(o, 0, &t);
        uint32_t c = lfo_crc32(0, t.p, t.n);
        lfo_buf_free(&t);
        uint8_t cb[2] = {(uint8_t)c, (uint8_t)(c >> 8)};
        buf_put(b, cb, 2);
    }
}

lfo_encoder *lfo_encoder_new(int format, const lfo_opts *o) {
    lfo_encoder *e = (lfo_encoder *)calloc(1, sizeof *e);
    e->format = format;
    e->o = *o;
    if (e->o.window_size > 32768) e->o.window_size = 32768;
    if (e->o.max_length > 258) e->o.max_length = 258;
    e->adler = 1;
    /* EncodeOptions::get_block_type / get_block_size encode.rs:112-127 */
    if (o->no_compression) {
        e->block_type = 0;
        e->block_size = o->block_size < 0xFFFF ? o->block_size : 0xFFFF;
    } else {
        e->block_type = o->dynamic_huffman ? 2 : 1;
        e->block_size = o->block_size;
    }
    if (format == LFO_GZIP) {
        gzip_header_bytes(&e->o, e->o.hcrc, &e->out);
    } else if (format == LFO_ZLIB) {
        /* zlib Header::write_to zlib.rs:267-279; from_lz77 :212-220; from_u16 :132-151 */
        uint32_t ws = e->o.window_size;
        uint8_t cinfo = ws > 16384 ? 7 : ws > 8192 ? 6 : ws > 4096 ? 5 : ws > 2048 ? 4 : ws > 1024 ? 3
                        : ws > 512 ? 2 : ws > 256 ? 1 : 0;
        /* level: lz77 Balance→Default(2); NoCompression encoder → None→Fastest(0);
         * no_compression() option → Fastest(0) (zlib.rs:471-475) */
        uint8_t level = (o->no_compression || o->lz77_kind == LFO_LZ77_NOCOMPRESSION) ? 0 : 2;
        if (o->lz77_kind == LFO_LZ77_CUSTOM && !o->no_compression) level = (uint8_t)(o->custom_level & 3); /* zlib.rs:59-68 */
        uint8_t cmf = (uint8_t)((cinfo << 4) | 8);
        uint8_t flg = (uint8_t)(level << 6);
        uint32_t check = ((uint32_t)cmf << 8) + flg;
        if (check % 31 != 0) flg = (uint8_t)(flg + (31 - check % 31));
        uint8_t h[2] = {cmf, flg};
        buf_put(&e->out, h, 2);
    }
    return e;
}

void lfo_encoder_write(lfo_encoder *e, const uint8_t *p, size_t n) {
    /* Block::write encode.rs:277-286 */
    if (e->block_type == 0) buf_put(&e->raw, p, n);
    else {
        e->original_size += n; /* CompressBuf::append encode.rs:405-408 */
        lz_encode(e, p, n);
    }
    while (block_len(e) >= e->block_size) block_flush(e, 0);
    /* container bookkeeping gzip.rs:890-895, zlib.rs:661-665 */
    if (e->format == LFO_GZIP) {
        e->crc = lfo_crc32(e->crc, p, n);
        e->isize += (uint32_t)n;
    } else if (e->format == LFO_ZLIB) {
        e->adler = lfo_adler32(e->adler, p, n);
    }
}

void lfo_encoder_flush(lfo_encoder *e) {
    /* deflate Encoder::flush encode.rs:245-248 ; zlib Sync → zlib_sync_flush :225-234 */
    block_flush(e, 0);
    if (e->format == LFO_ZLIB && e->o.zlib_sync_flush) {
        bw_write_bits(e, 1, 0);
        bw_write_bits(e, 2, 0);
        bw_flush(e);
        uint8_t m[4] = {0, 0, 255, 255};
        buf_put(&e->out, m, 4);
    }
}

const uint8_t *lfo_encoder_finish(lfo_encoder *e, size_t *out_len) {
    if (!e->finished) {
        block_flush(e, 1); /* Block::finish encode.rs:296-303 */
        bw_flush(e);
        if (e->format == LFO_GZIP) { /* Trailer::write_to gzip.rs:114-121 */
            uint8_t t[8] = {(uint8_t)e->crc, (uint8_t)(e->crc >> 8), (uint8_t)(e->crc >> 16),
                            (uint8_t)(e->crc >> 24), (uint8_t)e->isize, (uint8_t)(e->isize >> 8),
                            (uint8_t)(e->isize >> 16), (uint8_t)(e->isize >> 24)};
            buf_put(&e->out, t, 8);
        } else if (e->format == LFO_ZLIB) { /* zlib.rs:630-639 big-endian */
            uint8_t t[4] = {(uint8_t)(e->adler >> 24), (uint8_t)(e->adler >> 16),
                            (uint8_t)(e->adler >> 8), (uint8_t)e->adler};
            buf_put(&e->out, t, 4);
        }
        e->finished = 1;
    }
    *out_len = e->out.n;
    return e->out.p;
}
const uint8_t *lfo_encoder_output(lfo_encoder *e, size_t *out_len) {
    *out_len = e->out.n;
    return e->out.p;
}
void lfo_encoder_free(lfo_encoder *e) {
    if (!e) return;
    lfo_buf_free(&e->out);
    lfo_buf_free(&e->raw);
    lfo_buf_free(&e->lzbuf);
    free(e->syms);
    free(e);
}

int lfo_encode_buffer(int format, const lfo_opts *o, const uint8_t *in, size_t n,
                      size_t fixed_write, lfo_buf *out) {
    lfo_encoder *e = lfo_encoder_new(format, o);
    if (fixed_write == 0) {
        lfo_encoder_write(e, in, n); /* write_all of one slice = one write() (encode.rs:243) */
    } else {
        for (size_t off = 0; off < n; off += fixed_write)
            lfo_encoder_write(e, in + off, n - off < fixed_write ? n - off : fixed_write);
    }
    size_t len;
    const uint8_t *p = lfo_encoder_finish(e, &len);
    out->n = 0;
    buf_put(out, p, len);
    lfo_encoder_free(e);
    return LFO_OK;
}

/* ------------------------------------------------------------------ decoder */
typedef struct {
    const uint8_t *in;
    size_t n, pos; /* inner reader cursor */
    uint32_t last_read;
    uint32_t offset; /* bit.rs:62-68 */
    int err;         /* latched last_error kind (0 none) */
    char msg[160];
} breader;

static void br_set_err(breader *r, int kind, const char *msg) {
    r->err = kind; /* set_last_error overwrites (bit.rs:84-86) */
    snprintf(r->msg, sizeof r->msg, "%s", msg);
}
static uint16_t br_peek(breader *r, uint32_t w) {
    /* peek_bits_unchecked bit.rs:111-125 (+fill_next_u8 132-141) */
    while (32 < r->offset + w) {
        if (r->err) return 0;
        r->offset -= 8;
        r->last_read >>= 8;
        if (r->pos >= r->n) {
            r->err = LFO_UNEXPECTED_EOF;
            snprintf(r->msg, sizeof r->msg, "failed to fill whole buffer");
            return 0;
        }
        r->last_read |= (uint32_t)r->in[r->pos++] << 24;
    }
    uint32_t bits = r->last_read >> (r->offset & 31); /* wrapping_shr */
    return (uint16_t)(bits & ((1u << w) - 1));
}
static inline void br_skip(breader *r, uint32_t w) { r->offset = (r->offset + w) & 0xFF; }
static uint16_t br_read_unchecked(breader *r, uint32_t w) {
    uint16_t v = br_peek(r, w);
    br_skip(r, w);
    return v;
}
/* read_bits (checked): returns -1 on error (error taken) */
static int br_read(breader *r, uint32_t w) {
    uint16_t v = br_read_unchecked(r, w);
    if (r->err) return -1;
    return v;
}

/* huffman::DecoderBuilder / Decoder huffman.rs:58-179 */
typedef struct {
    uint16_t *table;
    uint32_t max_bw, safe_bw;
} hdec;
static void hdec_free(hdec *d) { free(d->table); d->table = NULL; }

/* from_bitwidthes + restore_canonical_huffman_codes; returns 0 or error (msg in r) */
static int hdec_build(hdec *d, const uint8_t *bw, int n, int safe_some, uint32_t safe, int eob,
                      breader *r) {
    uint32_t max_bw = 0;
    for (int i = 0; i < n; i++) if (bw[i] > max_bw) max_bw = bw[i];
    d->max_bw = max_bw;
    d->table = (uint16_t *)malloc(sizeof(uint16_t) << max_bw);
    for (uint32_t i = 0; i < (1u << max_bw); i++) d->table[i] = 16;
    uint16_t code = 0;
    uint32_t prev = 0;
    for (uint32_t w = 1; w <= 15; w++)
        for (int s = 0; s < n; s++) {
            if (bw[s] != w) continue;
            code = (uint16_t)(code << (w - prev));
            /* set_mapping huffman.rs:96-122 */
            if (s == eob) { safe_some = 1; safe = w; }
            uint16_t value = (uint16_t)((s << 5) | w);
            uint16_t f = code, be = 0;
            for (uint32_t k = 0; k < w; k++) { be = (uint16_t)((be << 1) | (f & 1)); f >>= 1; }
            for (uint32_t pad = 0; pad < (1u << (max_bw - w)); pad++) {
                uint32_t i = (pad << w) | be;
                if (d->table[i] != 16) {
                    char m[160];
                    snprintf(m, sizeof m,
                             "Bit region conflict: i=%u, old_value=%u, new_value=%u, symbol=%d",
                             i, d->table[i], value, s);
                    br_set_err(r, LFO_INVALID_DATA, m);
                    return -1;
                }
                d->table[i] = value;
            }
            code++;
            prev = w;
        }
    uint32_t sp = safe_some ? safe : 1; /* finish huffman.rs:123-132 */
    d->safe_bw = max_bw < sp ? max_bw : sp;
    return 0;
}
static uint16_t hdec_decode_unchecked(const hdec *d, breader *r) {
    /* huffman.rs:157-179 */
    uint32_t peek = d->safe_bw, bw;
    uint16_t value;
    for (;;) {
        uint16_t code = br_peek(r, peek);
        value = d->table[code];
        bw = value & 31;
        if (bw <= peek) break;
        if (bw > d->max_bw) {
            br_set_err(r, LFO_INVALID_DATA, "Invalid huffman coded stream");
            break;
        }
        peek = bw;
    }
    br_skip(r, bw);
    return (uint16_t)(value >> 5);
}

typedef struct { hdec lit, dist; } symdec;

static int fixed_load(symdec *sd, breader *r) {
    /* FixedHuffmanCodec::load symbol.rs:290-315 */
    uint8_t lw[288], dw[32];
    for (int s = 0; s < 288; s++) lw[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
    /* the fixed table is built by explicit set_mapping calls with max_bitwidth 9; the
     * canonical construction over these widths yields the same codes */
    if (hdec_build(&sd->lit, lw, 288, 0, 0, 256, r)) return -1;
    /* distance: DecoderBuilder::new(5, literal.safely_peek_bitwidth, None); 30 codes of width 5
     * with code == symbol: canonical over 30 five-bit codes gives the same mapping; 30/31 unfilled */
    for (int i = 0; i < 30; i++) dw[i] = 5;
    if (hdec_build(&sd->dist, dw, 30, 1, sd->lit.safe_bw, -1, r)) return -1;
    return 0;
}

static int dyn_load(symdec *sd, breader *r) {
    /* DynamicHuffmanCodec::load symbol.rs:387-456 */
    int v;
    if ((v = br_read(r, 5)) < 0) return -1;
    int nl = v + 257;
    if ((v = br_read(r, 5)) < 0) return -1;
    int nd = v + 1;
    if ((v = br_read(r, 4)) < 0) return -1;
    int nb = v + 4;
    if (nd > 30) {
        char m[160];
        snprintf(m, sizeof m, "The value of HDIST is too big: max=30, actual=%d", nd);
        br_set_err(r, LFO_INVALID_DATA, m);
        return -1;
    }
    uint8_t cbw[19];
    memset(cbw, 0, sizeof cbw);
    for (int k = 0; k < nb; k++) {
        if ((v = br_read(r, 3)) < 0) return -1;
        cbw[CLEN_ORDER[k]] = (uint8_t)v;
    }
    hdec bd;
    if (hdec_build(&bd, cbw, 19, 1, 1, -1, r)) { hdec_free(&bd); return -1; }
    uint8_t lens[640];
    int have = 0;
    int phase_target = nl;
    int dist_start = -1;
    /* literal list, then distance list (spill handled by one contiguous array) */
    for (int phase = 0; phase < 2; phase++) {
        if (phase == 1) {
            dist_start = nl; /* drain(nl..) */
            phase_target = nl + nd;
        }
        while (have < phase_target) {
            uint16_t c = hdec_decode_unchecked(&bd, r);
            if (r->err) { hdec_free(&bd); return -1; }
            /* load_bitwidthes symbol.rs:459-484 */
            if (c <= 15) lens[have++] = (uint8_t)c;
            else if (c == 16) {
                if ((v = br_read(r, 2)) < 0) { hdec_free(&bd); return -1; }
                if (have == 0) { /* last = None (distance falls back to literal's last) */
                    br_set_err(r, LFO_INVALID_DATA, "No preceding value");
                    hdec_free(&bd);
                    return -1;
                }
                uint8_t last = lens[have - 1];
                for (int k = 0; k < v + 3; k++) lens[have++] = last;
            } else if (c == 17) {
                if ((v = br_read(r, 3)) < 0) { hdec_free(&bd); return -1; }
                for (int k = 0; k < v + 3; k++) lens[have++] = 0;
            } else {
                if ((v = br_read(r, 7)) < 0) { hdec_free(&bd); return -1; }
                for (int k = 0; k < v + 11; k++) lens[have++] = 0;
            }
        }
    }
    hdec_free(&bd);
    if (have - dist_start > nd) {
        char m[160];
        snprintf(m, sizeof m,
                 "The length of `distance_code_bitwidthes` is too large: actual=%d, expected=%d",
                 have - dist_start, nd);
        br_set_err(r, LFO_INVALID_DATA, m);
        return -1;
    }
    if (hdec_build(&sd->lit, lens, nl, 0, 0, 256, r)) return -1;
    if (hdec_build(&sd->dist, lens + nl, nd, 1, sd->lit.safe_bw, -1, r)) return -1;
    return 0;
}

typedef struct {
    breader r;
    lfo_buf *out;
    size_t member_start; /* Lz77Decoder buffer start (cleared per gzip member) */
    lfo_blockinfo *info;  /* optional recorder (test helper) */
    size_t info_max, info_n;
} inflater;
static uint64_t br_bitpos(const breader *r) { return (uint64_t)r->pos * 8 - (32 - r->offset); }

/* deflate::Decoder::read loop (decode.rs:136-164) run to completion.
 * Returns LFO_OK or an error kind (message in r.msg). */
static int inflate_stream(inflater *z) {
    breader *r = &z->r;
    for (;;) {
        uint64_t start_bit = br_bitpos(r);
        size_t out_start = z->out->n;
        int bfinal = br_read(r, 1);
        if (bfinal < 0) return r->err;
        int btype = br_read(r, 2);
        if (btype < 0) return r->err;
        if (btype == 0) {
            /* read_non_compressed_block decode.rs:81-111 */
            r->offset = 32; /* bit_reader.reset() */
            if (r->n - r->pos < 2) { r->pos = r->n; br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
            uint16_t len = (uint16_t)(r->in[r->pos] | r->in[r->pos + 1] << 8);
            r->pos += 2;
            if (r->n - r->pos < 2) { r->pos = r->n; br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
            uint16_t nlen = (uint16_t)(r->in[r->pos] | r->in[r->pos + 1] << 8);
            r->pos += 2;
            if ((uint16_t)~len != nlen) {
                char m[160];
                snprintf(m, sizeof m, "LEN=%u is not the one's complement of NLEN=%u", len, nlen);
                br_set_err(r, LFO_INVALID_DATA, m);
                return r->err;
            }
            size_t avail = r->n - r->pos, take = len < avail ? len : avail;
            buf_put(z->out, r->in + r->pos, take); /* extend_from_reader keeps what it read */
            r->pos += take;
            if (take != len) {
                char m[160];
                snprintf(m, sizeof m, "The reader has incorrect length: expected %u, read %zu", len, take);
                br_set_err(r, LFO_UNEXPECTED_EOF, m);
                return r->err;
            }
        } else if (btype == 3) {
            br_set_err(r, LFO_INVALID_DATA, "btype 0x11 of DEFLATE is reserved(error) value");
            return r->err;
        } else {
            /* read_compressed_block decode.rs:112-130 */
            symdec sd;
            memset(&sd, 0, sizeof sd);
            int rc = btype == 1 ? fixed_load(&sd, r) : dyn_load(&sd, r);
            if (rc) { hdec_free(&sd.lit); hdec_free(&sd.dist); return r->err; }
            for (;;) {
                /* symbol::Decoder::decode_unchecked symbol.rs:193-244 */
                uint16_t d = hdec_decode_unchecked(&sd.lit, r);
                int is_ptr = 0, is_eob = 0;
                uint16_t length = 0, distance = 0;
                if (d <= 255) {
                } else if (d == 256) {
                    is_eob = 1;
                } else if (d == 286 || d == 287) {
                    char m[160];
                    snprintf(m, sizeof m, "The value %u must not occur in compressed data", d);
                    br_set_err(r, LFO_INVALID_DATA, m);
                    is_eob = 1;
                } else {
                    length = (uint16_t)(LEN_BASE[d - 257] + br_read_unchecked(r, LEN_EXTRA[d - 257]));
                    uint16_t dc = hdec_decode_unchecked(&sd.dist, r);
                    distance = (uint16_t)(DIST_BASE[dc] + br_read_unchecked(r, DIST_EXTRA[dc]));
                    is_ptr = 1;
                }
                if (r->err) { hdec_free(&sd.lit); hdec_free(&sd.dist); return r->err; }
                if (is_eob) break;
                if (!is_ptr) {
                    buf_put1(z->out, (uint8_t)d);
                } else {
                    /* Lz77Decoder::decode lib.rs:164-194 */
                    size_t blen = z->out->n - z->member_start;
                    uint32_t dist = distance;
                    if (blen < dist) {
                        char m[160];
                        snprintf(m, sizeof m, "Too long backword reference: buffer.len=%zu, distance=%u", blen, dist);
                        br_set_err(r, LFO_INVALID_DATA, m);
                        hdec_free(&sd.lit); hdec_free(&sd.dist);
                        return r->err;
                    }
                    buf_reserve(z->out, length);
                    uint8_t *o = z->out->p + z->out->n;
                    for (uint32_t k = 0; k < length; k++) o[k] = o[(ptrdiff_t)k - (ptrdiff_t)dist];
                    z->out->n += length;
                }
            }
            hdec_free(&sd.lit);
            hdec_free(&sd.dist);
        }
        if (z->info && z->info_n < z->info_max) {
            lfo_blockinfo *bi = &z->info[z->info_n];
            bi->start_bit = start_bit;
            bi->end_bit = btype == 0 ? (uint64_t)r->pos * 8 : br_bitpos(r);
            bi->btype = (uint32_t)btype;
            bi->bfinal = (uint32_t)bfinal;
            bi->out_len = z->out->n - out_start;
        }
        z->info_n++;
        if (bfinal) return LFO_OK;
    }
}

/* gzip Header::read_from gzip.rs:390-446.  Returns 0, or error kind */
static int gzip_read_header(breader *r) {
    if (r->n - r->pos < 10) { r->pos = r->n; br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
    const uint8_t *h = r->in + r->pos;
    size_t start = r->pos;
    r->pos += 10;
    if (h[0] != 31 || h[1] != 139) { br_set_err(r, LFO_INVALID_DATA, "Unexpected GZIP ID"); return r->err; }
    if (h[2] != 8) {
        char m[160];
        snprintf(m, sizeof m, "Compression methods other than DEFLATE(8) are unsupported: method=%u", h[2]);
        br_set_err(r, LFO_INVALID_DATA, m);
        return r->err;
    }
    uint8_t flags = h[3];
    size_t extra_off = 0, extra_len = 0, name_off = 0, name_len = 0, com_off = 0, com_len = 0;
    if (flags & 4) {
        if (r->n - r->pos < 2) { r->pos = r->n; br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
        size_t xl = (size_t)(r->in[r->pos] | r->in[r->pos + 1] << 8);
        r->pos += 2;
        extra_off = r->pos;
        extra_len = xl;
        /* ExtraField::read_from gzip.rs:470-485: subfields parsed inside take(xl) */
        size_t lim = xl, p = r->pos;
        while (lim > 0) {
            if (lim < 4 || r->n - p < 4) { r->pos = r->n; br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
            size_t dl = (size_t)(r->in[p + 2] | r->in[p + 3] << 8);
            p += 4; lim -= 4;
            if (lim < dl || r->n - p < dl) { r->pos = r->n; br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
            p += dl; lim -= dl;
        }
        r->pos = p;
    }
    for (int k = 0; k < 2; k++) {
        if (!(flags & (k ? 16 : 8))) continue;
        size_t s = r->pos;
        for (;;) { /* read_cstring gzip.rs:449-462 */
            if (r->pos >= r->n) { br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
            if (r->in[r->pos++] == 0) break;
        }
        if (k) { com_off = s; com_len = r->pos - s; } else { name_off = s; name_len = r->pos - s; }
    }
    if (flags & 2) {
        if (r->n - r->pos < 2) { r->pos = r->n; br_set_err(r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); return r->err; }
        uint16_t crc = (uint16_t)(r->in[r->pos] | r->in[r->pos + 1] << 8);
        r->pos += 2;
        /* expected = crc16 of the RE-SERIALISED header with only the five known flag bits and
         * HCRC cleared (gzip.rs:343-367) */
        lfo_buf t = {0, 0, 0};
        uint8_t hh[10];
        memcpy(hh, r->in + start, 10);
        hh[3] = (uint8_t)(flags & (1 | 4 | 8 | 16));
        /* compression level byte is re-serialised through from_u8/to_u8 (gzip.rs:69-82) */
        hh[8] = (hh[8] == 4 || hh[8] == 2) ? hh[8] : 0;
        buf_put(&t, hh, 10);
        if (flags & 4) {
            uint8_t l[2] = {(uint8_t)extra_len, (uint8_t)(extra_len >> 8)};
            buf_put(&t, l, 2);
            buf_put(&t, r->in + extra_off, extra_len);
        }
        if (flags & 8) buf_put(&t, r->in + name_off, name_len);
        if (flags & 16) buf_put(&t, r->in + com_off, com_len);
        uint16_t expect = (uint16_t)lfo_crc32(0, t.p, t.n);
        lfo_buf_free(&t);
        if (crc != expect) {
            char m[160];
            snprintf(m, sizeof m, "CRC16 of GZIP header mismatched: value=%u, expected=%u", crc, expect);
            br_set_err(r, LFO_INVALID_DATA, m);
            return r->err;
        }
    }
    return 0;
}

int lfo_decode(int format, int multi, const uint8_t *in, size_t n, lfo_buf *out,
               size_t *consumed, char *err) {
    inflater z;
    memset(&z, 0, sizeof z);
    z.r.in = in;
    z.r.n = n;
    z.r.offset = 32;
    z.out = out;
    out->n = 0;
    int rc = LFO_OK;
    if (err) err[0] = 0;
    if (format == LFO_ZLIB) {
        /* zlib Header::read_from zlib.rs:221-266 */
        if (n < 2) { z.r.pos = n; br_set_err(&z.r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); rc = z.r.err; goto done; }
        uint8_t cmf = in[0], flg = in[1];
        z.r.pos = 2;
        char m[160];
        if ((((uint32_t)cmf << 8) + flg) % 31 != 0) {
            snprintf(m, sizeof m, "Inconsistent ZLIB check bits: `CMF(%u) * 256 + FLG(%u)` must be a multiple of 31", cmf, flg);
            br_set_err(&z.r, LFO_INVALID_DATA, m); rc = z.r.err; goto done;
        }
        if ((cmf & 15) != 8) {
            snprintf(m, sizeof m, "Compression methods other than DEFLATE(8) are unsupported: method=%u", cmf & 15);
            br_set_err(&z.r, LFO_INVALID_DATA, m); rc = z.r.err; goto done;
        }
        if ((cmf >> 4) > 7) {
            snprintf(m, sizeof m, "CINFO above 7 are not allowed: value=%u", cmf >> 4);
            br_set_err(&z.r, LFO_INVALID_DATA, m); rc = z.r.err; goto done;
        }
        if (flg & 0x20) {
            if (n - z.r.pos < 4) { z.r.pos = n; br_set_err(&z.r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); rc = z.r.err; goto done; }
            uint32_t id = (uint32_t)in[2] << 24 | (uint32_t)in[3] << 16 | (uint32_t)in[4] << 8 | in[5];
            z.r.pos += 4;
            snprintf(m, sizeof m, "Preset dictionaries are not supported: dictionary_id=0x%X", id);
            br_set_err(&z.r, LFO_INVALID_DATA, m); rc = z.r.err; goto done;
        }
        rc = inflate_stream(&z);
        if (rc) goto done;
        /* zlib Decoder::read zlib.rs:377-409 */
        if (n - z.r.pos < 4) { z.r.pos = n; br_set_err(&z.r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); rc = z.r.err; goto done; }
        uint32_t ad = (uint32_t)in[z.r.pos] << 24 | (uint32_t)in[z.r.pos + 1] << 16 | (uint32_t)in[z.r.pos + 2] << 8 | in[z.r.pos + 3];
        z.r.pos += 4;
        uint32_t got = lfo_adler32(1, out->p, out->n);
        if (ad != got) {
            snprintf(m, sizeof m, "Adler32 checksum mismatched: value=%u, expected=%u", got, ad);
            br_set_err(&z.r, LFO_INVALID_DATA, m); rc = z.r.err; goto done;
        }
    } else if (format == LFO_GZIP) {
        int first = 1;
        for (;;) {
            size_t save = z.r.pos;
            int hr = gzip_read_header(&z.r);
            if (hr) {
                if (!first && hr == LFO_UNEXPECTED_EOF) { /* MultiDecoder gzip.rs:1150-1156 */
                    z.r.err = 0;
                    z.r.msg[0] = 0;
                    (void)save;
                    rc = LFO_OK;
                } else rc = hr;
                break;
            }
            z.member_start = out->n; /* Decoder::reset → lz77 clear */
            z.r.offset = 32;
            rc = inflate_stream(&z);
            if (rc) break;
            /* Trailer::read_from gzip.rs:103-113; CRC verified, ISIZE ignored gzip.rs:1030-1042 */
            if (n - z.r.pos < 8) { z.r.pos = n; br_set_err(&z.r, LFO_UNEXPECTED_EOF, "failed to fill whole buffer"); rc = z.r.err; break; }
            const uint8_t *t = in + z.r.pos;
            uint32_t crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
            z.r.pos += 8;
            uint32_t got = lfo_crc32(0, out->p + z.member_start, out->n - z.member_start);
            if (crc != got) {
                char m[160];
                snprintf(m, sizeof m, "CRC32 mismatched: value=%u, expected=%u", got, crc);
                br_set_err(&z.r, LFO_INVALID_DATA, m); rc = z.r.err; break;
            }
            if (!multi) break;
            first = 0;
        }
    } else {
        rc = inflate_stream(&z);
    }
done:
    if (consumed) *consumed = z.r.pos;
    if (err) snprintf(err, 160, "%s", z.r.msg);
    return rc;
}

long lfo_scan_blocks(const uint8_t *in, size_t n, lfo_blockinfo *info, size_t max_info) {
    /* test helper: decode a raw DEFLATE stream recording every block's bit range */
    lfo_buf out = {0, 0, 0};
    inflater z;
    memset(&z, 0, sizeof z);
    z.r.in = in;
    z.r.n = n;
    z.r.offset = 32;
    z.out = &out;
    z.info = info;
    z.info_max = max_info;
    int rc = inflate_stream(&z);
    lfo_buf_free(&out);
    return rc ? -1 : (long)z.info_n;
}
