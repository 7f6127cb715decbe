/*
 * lfo_core.c — ORACLE (test infrastructure only; see lfo.h).
 * Checksums, the default LZ77 encoder and the length-limited Huffman builder,
 * restated from the reference line by line (citations inline).
 */
#include "lfo.h"
#include <stdlib.h>
#include <string.h>

void lfo_buf_free(lfo_buf *b) {
    free(b->p);
    b->p = NULL;
    b->n = b->cap = 0;
}

/* ------------------------------------------------------------------ checksums
 * src/checksum.rs:22-33 wraps crc32fast::Hasher (crate "crc32fast" ^1.1.1, not in
 * the tree): the published algorithm is CRC-32/ISO-HDLC (reflected 0xEDB88320,
 * init/xorout 0xFFFFFFFF).  Pinned by checksum.rs:44-49 ("abcde" → 0x8587D865). */
static uint32_t crc_tab[8][256];
static int crc_ready;
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
        crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++)
            crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xFF];
    crc_ready = 1;
}
uint32_t lfo_crc32(uint32_t crc, const uint8_t *p, size_t n) {
    if (!crc_ready) crc_init();
    uint32_t c = ~crc;
    while (n >= 8) { /* slice-by-8, as crc32fast's baseline path does */
        uint32_t a = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
        a ^= c;
        c = crc_tab[7][a & 0xFF] ^ crc_tab[6][(a >> 8) & 0xFF] ^ crc_tab[5][(a >> 16) & 0xFF] ^
            crc_tab[4][a >> 24] ^ crc_tab[3][p[4]] ^ crc_tab[2][p[5]] ^ crc_tab[1][p[6]] ^
            crc_tab[0][p[7]];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

/* src/checksum.rs:4-15 wraps adler32::RollingAdler32 (crate "adler32" 1.x): RFC 1950
 * Adler-32, modulus 65521.  Pinned by checksum.rs:51-56 ("abcde" → 0x05C801F0). */
uint32_t lfo_adler32(uint32_t adler, const uint8_t *p, size_t n) {
    uint32_t a = adler & 0xFFFF, b = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        while (k--) {
            a += *p++;
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
    }
    return (b << 16) | a;
}

/* ------------------------------------------------------------------ LZ77
 * libflate_lz77/src/default.rs.  PrefixTable (132-152): exact 3-byte key → last
 * position.  Small = HashMap when the chunk is < 32768 bytes, Large = 65536 buckets
 * of (third byte, position) vectors (155-183).  Both are "exact key, last writer
 * wins"; both are restated so the CPU baseline pays the same costs. */
typedef struct {
    uint8_t k;
    uint32_t v;
} bent;
typedef struct {
    bent *e;
    uint32_t n, cap;
} bucket;

typedef struct {
    int large;
    bucket *b;      /* large: 65536 buckets */
    uint32_t *keys; /* small: open addressing, key+1 (0 = empty) */
    uint32_t *vals;
    uint32_t mask;
} ptable;

static void pt_new(ptable *t, size_t bytes) {
    memset(t, 0, sizeof *t);
    if (bytes < 32768) { /* default.rs:138 */
        uint32_t cap = 64;
        while (cap < bytes * 2 + 8) cap <<= 1;
        t->keys = (uint32_t *)calloc(cap, sizeof(uint32_t));
        t->vals = (uint32_t *)malloc(cap * sizeof(uint32_t));
        t->mask = cap - 1;
    } else {
        t->large = 1;
        t->b = (bucket *)calloc(65536, sizeof(bucket)); /* default.rs:161-165 */
    }
}
static void pt_free(ptable *t) {
    if (t->large) {
        for (int i = 0; i < 65536; i++) free(t->b[i].e);
        free(t->b);
    } else {
        free(t->keys);
        free(t->vals);
    }
}
/* returns old position or -1 (default.rs:146-151,166-182) */
static inline int64_t pt_insert(ptable *t, const uint8_t *p, uint32_t pos) {
    if (t->large) {
        bucket *bk = &t->b[((uint32_t)p[0] << 8) + p[1]];
        uint8_t k = p[2];
        for (uint32_t i = 0; i < bk->n; i++)
            if (bk->e[i].k == k) {
                uint32_t old = bk->e[i].v;
                bk->e[i].v = pos;
                return old;
            }
        if (bk->n == bk->cap) {
            bk->cap = bk->cap ? bk->cap * 2 : 4;
            bk->e = (bent *)realloc(bk->e, bk->cap * sizeof(bent));
        }
        bk->e[bk->n].k = k;
        bk->e[bk->n].v = pos;
        bk->n++;
        return -1;
    }
    uint32_t key = ((uint32_t)p[0] << 16 | (uint32_t)p[1] << 8 | p[2]) + 1;
    uint32_t h = (key * 2654435761u) >> 7;
    for (;;) {
        h &= t->mask;
        if (t->keys[h] == key) {
            uint32_t old = t->vals[h];
            t->vals[h] = pos;
            return old;
        }
        if (t->keys[h] == 0) {
            t->keys[h] = key;
            t->vals[h] = pos;
            return -1;
        }
        h++;
    }
}

size_t lfo_lz77_chunk(const uint8_t *buf, size_t n, uint32_t window, uint32_t max_len,
                      uint32_t *out) {
    /* default.rs:69-109 */
    ptable t;
    pt_new(&t, n); /* :73 fresh table per flush */
    size_t nc = 0, i = 0;
    size_t end = (n > 3 ? n : 3) - 3; /* :75 */
    while (i < end) {
        int64_t j = pt_insert(&t, buf + i, (uint32_t)i); /* :77-78 */
        if (j >= 0) {
            size_t dist = i - (size_t)j;
            if (dist <= window) { /* :81 inclusive */
                /* longest_common_prefix(buf, i+3, j+3, max) :122-129:
                 * buf[i+3..].iter().take(max-3).zip(&buf[j+3..]).take_while(eq).count() */
                size_t lim = n - (i + 3);
                if (lim > max_len - 3) lim = max_len - 3;
                size_t l = 0;
                const uint8_t *a = buf + i + 3, *b = buf + (size_t)j + 3;
                while (l < lim && a[l] == b[l]) l++;
                size_t length = 3 + l;
                out[nc++] = (uint32_t)length << 16 | (uint32_t)dist; /* :88-91 */
                for (size_t k = i + 1; k < i + length; k++) {         /* :92-97 */
                    if (k >= end) break;
                    pt_insert(&t, buf + k, (uint32_t)k);
                }
                i += length;
                continue;
            }
        }
        out[nc++] = (uint32_t)buf[i] << 16; /* :102 */
        i++;
    }
    for (; i < n; i++) out[nc++] = (uint32_t)buf[i] << 16; /* :105-107 */
    pt_free(&t);
    return nc;
}

/* ------------------------------------------------------------------ Huffman
 * src/huffman.rs */

/* ordinary_huffman_codes::calc_optimal_max_bitwidth (huffman.rs:261-274): max-heap of
 * (−freq, width) tuples; pop two, push (w1+w2, 1+max(width)); result max(1, root width).
 * dary_heap 0.3.5 is not in the tree; only the pop order of a totally ordered key
 * matters, so any correct max-heap reproduces it. */
typedef struct {
    int64_t w;
    uint8_t d;
} hitem;
static inline int h_less(hitem a, hitem b) { return a.w < b.w || (a.w == b.w && a.d < b.d); }
static void h_push(hitem *h, int *n, hitem x) {
    int i = (*n)++;
    h[i] = x;
    while (i > 0) {
        int p = (i - 1) / 2;
        if (!h_less(h[p], h[i])) break;
        hitem t = h[p];
        h[p] = h[i];
        h[i] = t;
        i = p;
    }
}
static hitem h_pop(hitem *h, int *n) {
    hitem top = h[0];
    h[0] = h[--(*n)];
    int i = 0;
    for (;;) {
        int l = 2 * i + 1, r = l + 1, m = i;
        if (l < *n && h_less(h[m], h[l])) m = l;
        if (r < *n && h_less(h[m], h[r])) m = r;
        if (m == i) break;
        hitem t = h[m];
        h[m] = h[i];
        h[i] = t;
        i = m;
    }
    return top;
}
static int optimal_max_bitwidth(const size_t *freq, int nsym) {
    hitem *h = (hitem *)malloc(sizeof(hitem) * (size_t)(nsym + 1));
    int n = 0;
    for (int i = 0; i < nsym; i++)
        if (freq[i] > 0) {
            hitem x = {-(int64_t)freq[i], 0};
            h_push(h, &n, x);
        }
    while (n > 1) {
        hitem a = h_pop(h, &n), b = h_pop(h, &n);
        hitem c = {a.w + b.w, (uint8_t)(1 + (a.d > b.d ? a.d : b.d))};
        h_push(h, &n, c);
    }
    int d = n ? h[0].d : 0;
    free(h);
    return d > 1 ? d : 1;
}

/* length_limited_huffman_codes (huffman.rs:276-363), restated literally: nodes carry
 * their symbol multiset. */
typedef struct {
    uint16_t *sym;
    uint32_t ns;
    size_t weight;
} node;
typedef struct {
    node *v;
    int n;
} nlist;

static node node_clone(const node *a) {
    node r;
    r.ns = a->ns;
    r.weight = a->weight;
    r.sym = (uint16_t *)malloc(sizeof(uint16_t) * (a->ns ? a->ns : 1));
    memcpy(r.sym, a->sym, sizeof(uint16_t) * a->ns);
    return r;
}
static nlist list_clone(const nlist *s) {
    nlist r;
    r.n = s->n;
    r.v = (node *)malloc(sizeof(node) * (size_t)(s->n ? s->n : 1));
    for (int i = 0; i < s->n; i++) r.v[i] = node_clone(&s->v[i]);
    return r;
}
static void list_free(nlist *l) {
    for (int i = 0; i < l->n; i++) free(l->v[i].sym);
    free(l->v);
}
/* package (huffman.rs:350-362): pair (2i,2i+1); odd tail dropped; len<2 unchanged */
static nlist package(nlist in) {
    if (in.n >= 2) {
        int nl = in.n / 2;
        for (int i = 0; i < nl; i++) {
            node a = in.v[2 * i], b = in.v[2 * i + 1];
            node m;
            m.ns = a.ns + b.ns;
            m.weight = a.weight + b.weight;
            m.sym = (uint16_t *)malloc(sizeof(uint16_t) * m.ns);
            memcpy(m.sym, a.sym, sizeof(uint16_t) * a.ns);
            memcpy(m.sym + a.ns, b.sym, sizeof(uint16_t) * b.ns);
            free(a.sym);
            free(b.sym);
            in.v[i] = m;
        }
        if (in.n & 1) free(in.v[in.n - 1].sym);
        in.n = nl;
    }
    return in;
}
/* merge (huffman.rs:330-349): take x only if x.weight < y.weight (ties → y first) */
static nlist merge(nlist x, nlist y) {
    nlist z;
    z.v = (node *)malloc(sizeof(node) * (size_t)(x.n + y.n + 1));
    z.n = 0;
    int i = 0, j = 0;
    while (i < x.n || j < y.n) {
        if (i >= x.n)
            z.v[z.n++] = y.v[j++];
        else if (j >= y.n)
            z.v[z.n++] = x.v[i++];
        else if (x.v[i].weight < y.v[j].weight)
            z.v[z.n++] = x.v[i++];
        else
            z.v[z.n++] = y.v[j++];
    }
    free(x.v);
    free(y.v);
    return z;
}
static void ll_calc(int max_bitwidth, const size_t *freq, int nsym, uint8_t *width) {
    /* huffman.rs:307-328 */
    nlist source;
    source.v = (node *)malloc(sizeof(node) * (size_t)(nsym + 1));
    source.n = 0;
    for (int s = 0; s < nsym; s++)
        if (freq[s] > 0) {
            node nd;
            nd.ns = 1;
            nd.weight = freq[s];
            nd.sym = (uint16_t *)malloc(sizeof(uint16_t));
            nd.sym[0] = (uint16_t)s;
            source.v[source.n++] = nd;
        }
    /* source.sort_by_key(|o| o.weight) — stable: insertion sort keeps symbol order on ties */
    for (int i = 1; i < source.n; i++) {
        node k = source.v[i];
        int j = i - 1;
        while (j >= 0 && source.v[j].weight > k.weight) {
            source.v[j + 1] = source.v[j];
            j--;
        }
        source.v[j + 1] = k;
    }
    nlist weighted = list_clone(&source);
    for (int r = 0; r < max_bitwidth - 1; r++) weighted = merge(package(weighted), list_clone(&source));
    memset(width, 0, (size_t)nsym);
    weighted = package(weighted);
    for (int i = 0; i < weighted.n; i++)
        for (uint32_t k = 0; k < weighted.v[i].ns; k++) width[weighted.v[i].sym[k]]++;
    list_free(&weighted);
    list_free(&source);
}

void lfo_huff_widths(const size_t *freq, int nsym, int limit, uint8_t *width_out) {
    /* EncoderBuilder::from_frequencies huffman.rs:202-209 */
    int opt = optimal_max_bitwidth(freq, nsym);
    int mb = limit < opt ? limit : opt;
    ll_calc(mb, freq, nsym, width_out);
}

void lfo_huff_codes(const uint8_t *width, int nsym, uint16_t *bits_out) {
    /* restore_canonical_huffman_codes huffman.rs:35-55 + inverse_endian 19-28 */
    uint16_t code = 0;
    int prev = 0;
    memset(bits_out, 0, sizeof(uint16_t) * (size_t)nsym);
    for (int w = 1; w <= 15; w++) /* stable sort by width == (width, symbol) order */
        for (int s = 0; s < nsym; s++)
            if (width[s] == w) {
                code = (uint16_t)(code << (w - prev));
                uint16_t f = code, t = 0;
                for (int k = 0; k < w; k++) {
                    t = (uint16_t)((t << 1) | (f & 1));
                    f >>= 1;
                }
                bits_out[s] = t;
                code++;
                prev = w;
            }
}
