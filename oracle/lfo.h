/*
 * lfo.h — ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C, single-threaded CPU restatement of the libflate hot path
 * (reference = sile/libflate v2.3.0; citations are file:line relative to the
 * reference tree).  It exists to CHECK the HIP product path; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (libflate_amd/csrc) never links or calls anything here.
 *
 * Parity status: PINNED against every known-answer vector the reference's own
 * tests/doctests hold for this path (tests/test_oracle_kat.py lists them with
 * their reference file:line).  No reference-produced vector larger than 48
 * input bytes exists, so large-input parity rests on this
 * restatement + python-zlib round trips.  The real reference cannot be built
 * here (Rust toolchain absent), so there is no oracle/_ref.
 */
#ifndef LFO_H
#define LFO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { LFO_DEFLATE = 0, LFO_ZLIB = 1, LFO_GZIP = 2 };
enum { LFO_OK = 0, LFO_INVALID_DATA = 1, LFO_UNEXPECTED_EOF = 2 };
enum { LFO_LZ77_DEFAULT = 0, LFO_LZ77_NOCOMPRESSION = 1, LFO_LZ77_CUSTOM = 2 };

/* growable byte buffer */
typedef struct lfo_buf {
    uint8_t *p;
    size_t n, cap;
} lfo_buf;
void lfo_buf_free(lfo_buf *b);

/* ---- checksums (src/checksum.rs:4-33; arithmetic = crates crc32fast/adler32) */
uint32_t lfo_crc32(uint32_t crc, const uint8_t *p, size_t n);     /* start with 0 */
uint32_t lfo_adler32(uint32_t adler, const uint8_t *p, size_t n); /* start with 1 */

/* ---- LZ77 (libflate_lz77/src/default.rs:69-109): one flush unit ("chunk").
 * code word = (val << 16) | dist ; dist == 0 → literal val, else pointer length val.
 * out must hold n words.  Returns the number of codes. */
size_t lfo_lz77_chunk(const uint8_t *buf, size_t n, uint32_t window, uint32_t max_len,
                      uint32_t *out);

/* ---- Huffman (src/huffman.rs:202-209,261-362): code lengths from frequencies. */
void lfo_huff_widths(const size_t *freq, int nsym, int limit, uint8_t *width_out);
/* canonical codes, bit-reversed for LSB-first emission (huffman.rs:35-55,19-28). */
void lfo_huff_codes(const uint8_t *width, int nsym, uint16_t *bits_out);

/* ---- encoder options: one-to-one with deflate/gzip/zlib EncodeOptions */
typedef struct lfo_opts {
    size_t block_size;       /* deflate/encode.rs:11  default 1 MiB */
    int dynamic_huffman;     /* encode.rs:107-110     default 1 */
    int no_compression;      /* encode.rs:77-80       stored blocks */
    int lz77_kind;           /* LFO_LZ77_* (lib.rs:111-145 for NOCOMPRESSION) */
    uint32_t window_size;    /* default.rs:226-231    default 32768 */
    uint32_t max_length;     /* default.rs:238-243    default 258 */
    int zlib_sync_flush;     /* zlib.rs:184-195 FlushMode::Sync */
    /* gzip header (gzip.rs:292-301) */
    uint32_t mtime;
    uint8_t os;              /* default 3 (Unix) */
    int is_text, hcrc;
    const uint8_t *extra;    /* serialized subfields (id[2] len[2] data)*, or NULL */
    size_t extra_len;
    const char *filename;    /* NUL-terminated or NULL */
    const char *comment;
    /* lz77_kind == LFO_LZ77_CUSTOM: the generic parameter `E: Lz77Encode` of EncodeOptions::with_lz77(E)
     * (src/deflate/encode.rs:59-65).  custom_cb(user, p, n, &codes) is E::encode(p[0..n], sink) when p != NULL and
     * E::flush(sink) when p == NULL (libflate_lz77/src/lib.rs:83-95); it returns the number of code words it emitted
     * and points *codes at them ((val << 16) | dist, valid until the next call).  custom_level = E::compression_level()
     * as LFO_LEVEL_* (lib.rs:96-99), window_size above = E::window_size() (lib.rs:103-106): both only reach the
     * container header (zlib.rs:212-220, gzip.rs:684). */
    size_t (*custom_cb)(void *user, const uint8_t *p, size_t n, const uint32_t **codes);
    void *custom_user;
    int custom_level;
} lfo_opts;
enum { LFO_LEVEL_NONE = 0, LFO_LEVEL_FAST = 1, LFO_LEVEL_BALANCE = 2, LFO_LEVEL_BEST = 3 };
void lfo_opts_default(lfo_opts *o);

typedef struct lfo_encoder lfo_encoder;
lfo_encoder *lfo_encoder_new(int format, const lfo_opts *o);
void lfo_encoder_write(lfo_encoder *e, const uint8_t *p, size_t n); /* ONE write() call */
void lfo_encoder_flush(lfo_encoder *e);                             /* Write::flush */
/* finish; returns pointer to the complete output (owned by e) */
const uint8_t *lfo_encoder_finish(lfo_encoder *e, size_t *out_len);
const uint8_t *lfo_encoder_output(lfo_encoder *e, size_t *out_len);
void lfo_encoder_free(lfo_encoder *e);

/* one-shot helper: schedule = write sizes w[0..nw) (sum may be < n: remainder is
 * one more write; nw==0 → single write_all); fixed_write>0 → every write that size */
int lfo_encode_buffer(int format, const lfo_opts *o, const uint8_t *in, size_t n,
                      size_t fixed_write, lfo_buf *out);

/* ---- decoder.  Decodes ONE stream (gzip: one member unless multi!=0) from memory.
 * out receives every byte decoded before success or failure (read_to_end +
 * unread_decoded_data, deflate/decode.rs:68-73).  *consumed = input bytes consumed
 * (incl. trailer).  Returns LFO_* ; err (>=160 bytes) gets the message text. */
int lfo_decode(int format, int multi, const uint8_t *in, size_t n, lfo_buf *out,
               size_t *consumed, char *err);

/* ---- block inspector used by tests: raw-deflate stream → per block
 * (start_bit, type, out_len); returns number of blocks or -1 */
typedef struct lfo_blockinfo {
    uint64_t start_bit, end_bit;
    uint32_t btype, bfinal;
    uint64_t out_len;
} lfo_blockinfo;
long lfo_scan_blocks(const uint8_t *in, size_t n, lfo_blockinfo *info, size_t max_info);

#ifdef __cplusplus
}
#endif
#endif
