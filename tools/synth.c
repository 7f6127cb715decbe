/*
 * synth.c — deterministic synthetic inputs for tests and bench.
 * Not part of the product library and not part of the oracle: plain data generators.
 *   TEXT(N, seed):   enwiki-titles-like lines (Zipf-distributed pseudo-words)
 *   LOWENT(N, seed): run / fixed-record mix (low entropy binary)
 * PRNG = splitmix64.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t sm64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline double u01(uint64_t *s) { return (double)(sm64(s) >> 11) * (1.0 / 9007199254740992.0); }

#define VOCAB 50000
static const char LETTERS[] = "etaoinshrdlcumwfgypbvkjxqz";
/* cumulative English letter frequencies (per 1000) in LETTERS order */
static const int LETCUM[26] = {127, 218, 300, 375, 445, 512, 575, 636, 696, 739, 779, 807, 835,
                               859, 883, 905, 925, 945, 964, 979, 989, 997, 998, 999, 1000, 1000};

void lfx_synth_text(uint8_t *out, size_t n, uint64_t seed) {
    uint64_t s = seed;
    char (*words)[16] = malloc((size_t)VOCAB * 16);
    uint8_t *wlen = malloc(VOCAB);
    double *cdf = malloc(sizeof(double) * VOCAB);
    double tot = 0;
    for (int i = 0; i < VOCAB; i++) {
        int len = 2 + (int)(sm64(&s) % 13);
        for (int k = 0; k < len; k++) {
            int r = (int)(sm64(&s) % 1000), c = 0;
            while (LETCUM[c] <= r) c++;
            words[i][k] = LETTERS[c];
        }
        if (sm64(&s) % 100 < 35) words[i][0] = (char)(words[i][0] - 32);
        wlen[i] = (uint8_t)len;
        tot += 1.0 / (double)(i + 1);
        cdf[i] = tot;
    }
    size_t p = 0;
    while (p < n) {
        char line[128];
        int ll = 0;
        int nw = 1 + (int)(sm64(&s) % 5);
        for (int w = 0; w < nw; w++) {
            double u = u01(&s) * tot;
            int lo = 0, hi = VOCAB - 1;
            while (lo < hi) {
                int mid = (lo + hi) / 2;
                if (cdf[mid] < u) lo = mid + 1; else hi = mid;
            }
            if (w) line[ll++] = '_';
            memcpy(line + ll, words[lo], wlen[lo]);
            ll += wlen[lo];
        }
        if (sm64(&s) % 10 == 0) {
            int y = 1800 + (int)(sm64(&s) % 225);
            line[ll++] = '_'; line[ll++] = '(';
            line[ll++] = (char)('0' + y / 1000); line[ll++] = (char)('0' + y / 100 % 10);
            line[ll++] = (char)('0' + y / 10 % 10); line[ll++] = (char)('0' + y % 10);
            line[ll++] = ')';
        }
        line[ll++] = '\n';
        size_t take = (size_t)ll < n - p ? (size_t)ll : n - p;
        memcpy(out + p, line, take);
        p += take;
    }
    free(words); free(wlen); free(cdf);
}

void lfx_synth_lowent(uint8_t *out, size_t n, uint64_t seed) {
    uint64_t s = seed;
    uint8_t pool[256][64];
    for (int i = 0; i < 256; i++)
        for (int k = 0; k < 64; k += 8) {
            uint64_t v = sm64(&s);
            memcpy(&pool[i][k], &v, 8);
        }
    size_t p = 0;
    while (p < n) {
        if (sm64(&s) & 1) {
            int r = (int)(sm64(&s) % 10);
            uint8_t b = r < 7 ? 0x00 : r < 9 ? 0xFF : (uint8_t)sm64(&s);
            size_t len = 64 + (size_t)(sm64(&s) % 4033);
            if (len > n - p) len = n - p;
            memset(out + p, b, len);
            p += len;
        } else {
            const uint8_t *rec = pool[sm64(&s) & 255];
            size_t len = 64 < n - p ? 64 : n - p;
            memcpy(out + p, rec, len);
            p += len;
        }
    }
}
