/*
 * lfx.h — C ABI of the MI355X-native DEFLATE hot path (drop-in boundary for sile/libflate).
 *
 * Every entry point names the reference interface it replaces (file:line in libflate v2.3.0).
 * Plain pointers and sizes only; no torch / HIP types in the signatures (a HIP stream is
 * passed as an opaque void*).  All compression / decompression work runs in hand-written HIP
 * kernels for gfx950; there is NO CPU fallback: without a usable device every compute call
 * returns LFX_E_DEVICE.
 *
 * Bit-exactness contract: compressed bytes are a function of (input bytes, sequence of write()
 * sizes, options) exactly as in the reference (SURVEY.md "fact 2").  One lfx_encoder_write()
 * call == one `Write::write` call; the one-shot calls take an explicit lfx_schedule.
 */
#ifndef LFX_H
#define LFX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFX_VERSION 0x000100

/* container formats: libflate::{deflate,zlib,gzip} */
enum { LFX_DEFLATE = 0, LFX_ZLIB = 1, LFX_GZIP = 2 };

/* status codes ~ io::ErrorKind used by the reference (src/lib.rs:10-29, src/bit.rs:132-141) */
enum {
    LFX_OK = 0,
    LFX_E_INVALID_DATA = 1,   /* io::ErrorKind::InvalidData  */
    LFX_E_UNEXPECTED_EOF = 2, /* io::ErrorKind::UnexpectedEof */
    LFX_E_IO = 3,             /* error returned by a user callback */
    LFX_E_OOM = 4,
    LFX_E_DEVICE = 5,         /* no device / HIP failure (never a silent CPU fallback) */
    LFX_E_ARG = 6,            /* out-of-domain option */
    LFX_E_NOSPACE = 7,        /* output capacity too small */
    LFX_E_UNSUPPORTED = 8,
    LFX_E_WOULD_BLOCK = 9     /* io::ErrorKind::WouldBlock: a read callback of a non-blocking decoder has nothing yet */
};

/* libflate_lz77::CompressionLevel (libflate_lz77/src/lib.rs:44-58) */
enum { LFX_LEVEL_NONE = 0, LFX_LEVEL_FAST = 1, LFX_LEVEL_BALANCE = 2, LFX_LEVEL_BEST = 3 };
/* which Lz77Encode implementation (lib.rs:83-145, default.rs:14-51) */
enum { LFX_LZ77_DEFAULT = 0, LFX_LZ77_NOCOMPRESSION = 1 };
/* zlib::FlushMode (src/zlib.rs:184-195) */
enum { LFX_FLUSH_NONE = 0, LFX_FLUSH_SYNC = 2 };

/* Options: deflate::EncodeOptions (src/deflate/encode.rs:17-128) + DefaultLz77EncoderBuilder
 * (libflate_lz77/src/default.rs:202-249) + gzip::HeaderBuilder (src/gzip.rs:126-288) +
 * zlib::EncodeOptions.flush_mode (src/zlib.rs:414-518). */
typedef struct lfx_encode_opts {
    uint64_t block_size;       /* encode.rs:11   default 1 MiB (hint; a block holds whole writes) */
    int32_t dynamic_huffman;   /* encode.rs:107  default 1; 0 = fixed_huffman_codes() */
    int32_t no_compression;    /* encode.rs:77   stored blocks */
    int32_t lz77_kind;         /* LFX_LZ77_*     with_lz77(...) */
    uint32_t window_size;      /* default.rs:226 default 32768 (clamped) */
    uint32_t max_length;       /* default.rs:238 default 258 (clamped; must be >= 3) */
    int32_t zlib_flush_mode;   /* zlib.rs:504    LFX_FLUSH_* */
    uint32_t mtime;            /* gzip.rs:167    (reference default = now; callers set it) */
    uint8_t os;                /* gzip.rs:173    default 3 (Unix) */
    uint8_t is_text;           /* gzip.rs:179 */
    uint8_t hcrc;              /* gzip.rs:185    verify() */
    uint8_t lz77_level;        /* 0: the level of lz77_kind's encoder; 1 + LFX_LEVEL_*: E::compression_level() of a caller-supplied
                                  Lz77Encode (lfx_encoder_write_codes) — it decides zlib's FLEVEL (zlib.rs:59-68,212-220) and gzip's XFL
                                  (gzip.rs:84-92,684) */
    const uint8_t *extra;      /* gzip.rs:191    serialized subfields (id[2] len[2] data)* */
    uint32_t extra_len;
    const char *filename;      /* gzip.rs:197    NUL-terminated */
    const char *comment;       /* gzip.rs:203 */
} lfx_encode_opts;
void lfx_encode_opts_default(lfx_encode_opts *o);

/* How the caller slices its writes (what decides LZ77 chunk and DEFLATE block boundaries:
 * default.rs:60-68, encode.rs:277-286). */
enum { LFX_SCHED_SINGLE = 0, /* one write_all(buf)            — every in-tree reference test */
       LFX_SCHED_FIXED = 1,  /* writes of `fixed_write` bytes — examples/flate.rs:52 uses 8192 */
       LFX_SCHED_LIST = 2 }; /* explicit sizes; a size of UINT64_MAX means Write::flush() */
#define LFX_SCHED_FLUSH UINT64_MAX
typedef struct lfx_schedule {
    int32_t kind;
    uint64_t fixed_write;
    const uint64_t *writes;
    size_t n_writes;
} lfx_schedule;

/* ---- device context (one per GPU; owns a HIP stream + cached scratch in HBM) ------------- */
typedef struct lfx_ctx lfx_ctx;
lfx_ctx *lfx_ctx_new(int device, int *status);
/* Every handle made from a context (lfx_encoder, lfx_decoder, lfx_lz77, a sharded encode in flight) uses it until that handle is
 * freed — free the handles first.  (The Rust crate's handles hold an Arc<Context>; the Python wrappers count themselves in and out
 * of their Context, because a garbage collector runs the finalizers of one unreachable group in an undefined order.) */
void lfx_ctx_free(lfx_ctx *c);
const char *lfx_ctx_last_error(const lfx_ctx *c);
/* use the caller's HIP stream (hipStream_t as void*) instead of the context's own */
void lfx_ctx_set_stream(lfx_ctx *c, void *hip_stream);
int lfx_device_count(void);

/* ---- one-shot, device-resident buffers (what io::copy over in-memory buffers amounts to) --
 * Replaces {deflate,zlib,gzip}::Encoder::{with_options, write*, finish}
 * (encode.rs:182-249, zlib.rs:577-681, gzip.rs:804-908) for a whole buffer. */
uint64_t lfx_encode_bound(uint64_t n, const lfx_encode_opts *o, const lfx_schedule *s);
int lfx_encode_device(lfx_ctx *c, int format, const lfx_encode_opts *o, const lfx_schedule *s,
                      const void *d_in, uint64_t n, void *d_out, uint64_t cap, uint64_t *out_len);
/* `count` independent streams in one call (BASELINE.json configs[2] needs thousands of 64 KiB streams): stream i is what
 * lfx_encode_device makes of d_in[in_off[i] .. +in_len[i]) with the same options and the schedule applied to it alone, and
 * lies at d_out[out_off[i] .. +out_len[i]) (out_off 4-byte aligned, out_cap[i] >= lfx_encode_bound(in_len[i])).  Offsets and
 * lengths are HOST arrays.  The output ranges [out_off[i], out_off[i] + out_cap[i]) must not overlap (LFX_E_ARG); the whole
 * span from the lowest out_off to the highest out_off + out_cap is zero-filled first — gaps between the ranges included, and
 * also when the call then fails.  A stream that does not fit its capacity voids the call: LFX_E_NOSPACE, status[i] =
 * LFX_E_NOSPACE for exactly the streams that were too small (0 for those that would have fit), every out_len[i] = 0. */
int lfx_encode_batch_device(lfx_ctx *c, int format, const lfx_encode_opts *o, const lfx_schedule *s, uint32_t count,
                            const void *d_in, const uint64_t *in_off, const uint64_t *in_len, void *d_out,
                            const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len, int32_t *status);
/* same, host buffers (H2D + device path + D2H; PCIe-inclusive).  Round 6: page-locked buffers — lfx_host_alloc below, or the
 * caller's own hipHostMalloc / hipHostRegister — are handed to the DMA engine as they are; pageable ones are staged through
 * page-locked slabs by four copy threads (one thread's memcpy is slower than the link).  lfx_decode_host likewise. */
void *lfx_host_alloc(size_t bytes);   /* page-locked host memory (NULL: none to be had); for buffers that cross PCIe often */
void lfx_host_free(void *p);
int lfx_encode_host(lfx_ctx *c, int format, const lfx_encode_opts *o, const lfx_schedule *s,
                    const void *in, uint64_t n, void *out, uint64_t cap, uint64_t *out_len);

/* Replaces {deflate,zlib,gzip}::Decoder::{new, read_to_end} (decode.rs:40-164,
 * zlib.rs:312-409, gzip.rs:941-1047) and gzip::MultiDecoder (gzip.rs:1100-1166, flag below).
 * *out_len = bytes produced (also on failure: read_to_end + unread_decoded_data,
 * decode.rs:68-73); *consumed = input bytes consumed including the trailer. */
#define LFX_DEC_MULTI 1u
int lfx_decode_device(lfx_ctx *c, int format, uint32_t flags, const void *d_in, uint64_t n,
                      void *d_out, uint64_t cap, uint64_t *out_len, uint64_t *consumed);
int lfx_decode_host(lfx_ctx *c, int format, uint32_t flags, const void *in, uint64_t n, void *out,
                    uint64_t cap, uint64_t *out_len, uint64_t *consumed);
/* `count` independent streams, one wavefront each (BASELINE.json configs[2]).
 * offsets/lengths are HOST arrays; status[i] gets LFX_* per stream. */
int lfx_decode_batch_device(lfx_ctx *c, int format, uint32_t count, const void *d_in,
                            const uint64_t *in_off, const uint64_t *in_len, void *d_out,
                            const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len,
                            int32_t *status);

/* ---- sharded encode: independent block ranges per rank, one gzip/zlib/deflate member
 * (SURVEY §8e).  prepare() runs match finding → Huffman build and the checksums and reports the
 * shard's bit length; after the ranks exchanged lfx_shard_info (RCCL all-gather), emit() packs the
 * shard at its global bit offset.  The byte at each shard boundary is shared (OR the halves). */
typedef struct lfx_shard_info {
    uint64_t total_bits; /* DEFLATE bits of this shard (incl. final byte alignment on the last) */
    uint64_t n_bytes;    /* uncompressed bytes */
    uint32_t crc32;      /* of the shard's input */
    uint32_t adler32;    /* of the shard's input, as if it started a stream (A0 = 1) */
} lfx_shard_info;
int lfx_encode_shard_prepare(lfx_ctx *c, int format, const lfx_encode_opts *o,
                             const lfx_schedule *s, const void *d_in, uint64_t n, int is_first,
                             int is_last, lfx_shard_info *info);
/* start_bit = bit offset of this shard's first DEFLATE bit inside the member (container header
 * included).  Writes bytes [start_bit/8, ceil((start_bit+total_bits)/8)) of the member to d_out
 * (d_out[0] is byte start_bit/8); the first shard also writes the container header, the last
 * shard the trailer built from the combined checksum / size given here. */
int lfx_encode_shard_emit(lfx_ctx *c, uint64_t start_bit, uint32_t combined_check,
                          uint64_t total_n, void *d_out, uint64_t cap, uint64_t *out_len);
/* Optional, in front of lfx_encode_shard_prepare: names the buffer lfx_encode_shard_emit will write, which is then zero-filled on
 * the context's side stream beside the prepare call's kernels instead of in front of the pack kernels (what lfx_encode_device
 * does with its own output).  (NULL, 0) waits for a fill in flight and forgets it (no emit call will follow). */
int lfx_encode_shard_prezero(lfx_ctx *c, void *d_out, uint64_t cap);
/* inverse of the sharded encode: decodes the blocks of ONE shard of a member.  d_in[0] is the byte
 * that holds bit `start_bit` (0..7) of the shard; a non-last shard ends when a block ends exactly
 * `total_bits` later (it holds no BFINAL block).  Reference-made blocks never reference earlier
 * blocks (fresh PrefixTable per flush, default.rs:73), which is what makes shards decodable alone. */
int lfx_decode_shard_device(lfx_ctx *c, const void *d_in, uint64_t n, uint64_t start_bit,
                            uint64_t total_bits, int is_last, void *d_out, uint64_t cap,
                            uint64_t *out_len);
/* ---- N-GPU decode of ONE member WITHOUT the encoder's layout (north_star: "independent DEFLATE blocks
 * partition across the GPUs").  The member is cut by compressed bytes; rank r holds the bytes [lo_byte, lo_byte + n_part)
 * of it in d_part — its own range [lo_byte, hi_byte) plus a tail of a few MiB from its right neighbour, so that a block
 * that starts inside the range can be scanned to its end.
 *   1. lfx_decode_range_scan: block finder + speculative scan of every candidate that STARTS inside the range →
 *      one tuple per candidate (bit offsets relative to the member).  first_bit: the known start of the member's first
 *      block, on the rank whose range holds it (right behind the container header), ~0 elsewhere.  final_from_bit:
 *      headers with BFINAL set are reported only from this bit of the member on (0: everywhere) — a member's last block
 *      is the only one that carries the flag, so looking for it near the end halves the finder's and the scan's work;
 *      when the chain of step 3 then breaks at the last block, scan again with 0;
 *   2. the ranks all-gather their tuples (the only collective: 56 bytes per candidate, a few hundred per rank);
 *   3. lfx_decode_chain (host only, deterministic): the true block list from the known first block; candidates that
 *      are not block starts are never reached.  LFX_E_UNSUPPORTED when the chain breaks (a stored / fixed block the
 *      finder cannot see, damage): decode on one GPU then;
 *   4. lfx_decode_range_emit: materialises the chain blocks owned by `rank` (those that start in its range) into
 *      d_out — the rank's slice of the output, which starts *out_base bytes into the member's output.  Must follow the
 *      same rank's range_scan on the same context (the scan's tables live in its scratch).  *state = 0: the slice holds
 *      its bytes.  Reference-made blocks never read earlier blocks (default.rs:73), so their slices always do.
 *      *state = 1: the slice's blocks read output in front of them (another encoder's member — the reference decodes any
 *      valid stream, src/deflate/decode.rs:112-164, libflate_lz77/src/lib.rs:164-194), possibly bytes another rank
 *      produces: the slice is held as 16-bit symbols and waits for the 32 KiB window in front of it;
 *   5. window hand-over, only when some rank reported state 1: every rank writes its slice's action on the 32 KiB in
 *      front of it as ONE index map (lfx_decode_range_map: 32768 uint16 on the device — a byte value, or 256 + j = "byte
 *      j of the window in front of my slice"), the ranks all-gather the maps (64 KiB each, rank order);
 *   6. lfx_decode_range_finish: composes the window in front of the slice from the maps of the ranks before it (d_maps:
 *      world x 32768 uint16 on the device, 16-byte aligned; NULL when no rank needed a window), replaces the slice's markers, and returns
 *      the slice's CRC-32 / Adler-32 (fold them with lfx_crc32_combine / lfx_adler32_combine and compare with the
 *      trailer). */
typedef struct lfx_blk_tuple {
    uint64_t start_bit, end_bit;  /* header bit, bit behind EndOfBlock (relative to the member's first byte) */
    uint64_t n_out;               /* bytes the block produces */
    uint64_t data_bit;            /* first symbol bit */
    uint32_t n_codes, nlanes;
    uint8_t btype, bfinal, status /* 0 = scanned to its EndOfBlock */, _pad;
    uint16_t rank;                /* owner: the rank whose range holds start_bit */
    uint16_t _pad2;
    uint32_t slot;                /* the owner's scan slot */
    uint32_t _pad3;
} lfx_blk_tuple;                  /* 56 bytes */
int lfx_decode_range_scan(lfx_ctx *c, const void *d_part, uint64_t n_part, uint64_t lo_byte, uint64_t hi_byte,
                          uint64_t first_bit, uint64_t final_from_bit, uint32_t rank, lfx_blk_tuple *tuples, uint32_t cap,
                          uint32_t *count);
int lfx_decode_chain(const lfx_blk_tuple *all, uint32_t n_all, uint64_t first_bit, uint32_t *chain, uint32_t cap,
                     uint32_t *n_chain, uint64_t *total_out);
int lfx_decode_range_emit(lfx_ctx *c, const void *d_part, uint64_t n_part, uint64_t lo_byte, const lfx_blk_tuple *all,
                          const uint32_t *chain, uint32_t n_chain, uint32_t rank, void *d_out, uint64_t cap,
                          uint64_t *out_len, uint64_t *out_base, uint32_t *state);
int lfx_decode_range_map(lfx_ctx *c, void *d_map);
int lfx_decode_range_finish(lfx_ctx *c, const void *d_maps, uint32_t rank, uint32_t *crc32, uint32_t *adler32);
/* stream concatenation on the writer rank: places one shard's bytes (as written by emit(), already
 * on this device — e.g. received over RCCL / xGMI) at byte start_bit/8 of the member buffer; when the shard starts
 * inside a byte (start_bit % 8 != 0) that byte is shared with the shard in front and is OR-ed.  Place the shards in
 * rank order. */
int lfx_shard_place_device(lfx_ctx *c, void *d_member, uint64_t cap, const void *d_part, uint64_t part_len,
                           uint64_t start_bit, int is_first);
/* ---- the N-GPU drivers (round 5): the sequencing of the steps above, with the caller's collectives -----------------------
 * One rank per GPU, one lfx_ctx per rank.  The reference has ONE encoder with ONE running checksum (gzip::Encoder::finish,
 * src/gzip.rs:858-868; src/checksum.rs:22-33); here the ranks' block ranges, bit offsets and partial checksums are exchanged
 * through an lfx_comm — five callbacks the caller implements over whatever moves bytes between its ranks (torch.distributed
 * in libflate_amd/sharded.py, MPI, ...), or lfx_comm_rccl() for RCCL over xGMI.  Every callback returns 0 on success.
 *   allgather: every rank contributes `bytes` HOST bytes; recv (world * bytes, rank order) is complete on return.
 *   isend / irecv: post a transfer of a DEVICE buffer (a binding may only collect them: RCCL groups them, torch batches them);
 *   start (round 6; may be NULL when isend / irecv start their transfers themselves): everything posted so far begins to
 *     move NOW and the call returns without waiting for it — lfx_sharded_encode_begin calls it last, so the shards travel
 *     while the caller works between begin and finish;
 *   wait: everything this rank posted is complete (transfers that were never started are started first).
 * A failure on one rank travels with the next collective and comes back from the same call on EVERY rank. */
typedef struct lfx_comm {
    void *user;
    uint32_t rank, world;
    int (*allgather)(void *user, const void *send, void *recv, uint64_t bytes);
    int (*isend)(void *user, const void *d_buf, uint64_t bytes, uint32_t to_rank);
    int (*irecv)(void *user, void *d_buf, uint64_t bytes, uint32_t from_rank);
    int (*wait)(void *user);
    int (*start)(void *user);
} lfx_comm;
/* RCCL binding: `nccl_comm` is an ncclComm_t, `hip_stream` the hipStream_t its collectives run on.  librccl is loaded at run
 * time (LFX_E_UNSUPPORTED when it is not there): liblfx.so does not link it.  isend / irecv open ONE ncclGroup, start closes
 * it (all transfers of the group begin together, one per xGMI link), wait synchronises the stream; an all-gather that finds a
 * group open closes it first (a collective issued inside an open group would only be queued, and its result read too early).
 * RCCL serialises the operations of one communicator: a caller that wants its small all-gathers (lfx_sharded_decode) not to
 * queue behind shards in flight gives the two drivers two communicators. */
int lfx_comm_rccl(void *nccl_comm, void *hip_stream, uint32_t rank, uint32_t world, lfx_comm *out);
void lfx_comm_rccl_free(lfx_comm *cm);

/* Sharded encode of ONE member: rank r holds the r-th slice of the input (whole blocks: n a multiple of the block size on
 * every rank but the last).  begin(): prepare → all-gather of the 32-byte shard infos → emit at the rank's bit offset →
 * the shards start travelling to rank 0 (all transfers posted at once; d_staging on rank 0 holds the shards of ranks 1.. side
 * by side, 256-byte aligned).  Between begin and finish the caller may use its own shard (d_part).  finish(): waits, places the
 * shards in d_member (rank 0; the byte two shards share is OR-ed) and frees the state.  → the member equals what ONE
 * encoder emits for the concatenated input. */
typedef struct lfx_sharded_enc lfx_sharded_enc;
typedef struct lfx_sharded_part {
    uint64_t start_bit;   /* of this rank's first DEFLATE bit inside the member */
    uint64_t end_bit;     /* the bit behind this rank's last one = the next rank's start_bit */
    uint64_t part_len;    /* bytes emitted into d_part (d_part[0] is byte start_bit / 8 of the member) */
    uint64_t member_len;  /* bytes of the whole member */
    uint64_t total_n;     /* uncompressed bytes of the whole member */
    uint32_t check;       /* combined CRC-32 (gzip) / Adler-32 (zlib) of the whole input */
    uint32_t _pad;
} lfx_sharded_part;
int lfx_sharded_encode_begin(lfx_ctx *c, const lfx_comm *cm, int format, const lfx_encode_opts *o, const lfx_schedule *s,
                             const void *d_in, uint64_t n, void *d_part, uint64_t part_cap, void *d_member /* rank 0 */,
                             uint64_t member_cap, void *d_staging /* rank 0 */, uint64_t staging_cap, lfx_sharded_enc **state,
                             lfx_sharded_part *out);
int lfx_sharded_encode_finish(lfx_ctx *c, const lfx_comm *cm, lfx_sharded_enc *state, uint64_t *member_len /* rank 0 */);

/* N-GPU decode of ONE member cut by compressed bytes (steps 1-6 above): rank r holds d_part = member bytes [lo_byte, hold_hi)
 * (lfx_sharded_byte_range: an equal share plus a tail of one maximal block) and gets its slice of the output in d_out.
 * first_bit: the member's first DEFLATE bit (behind the container header); member_len: 0 if unknown.  Collectives: the
 * candidate tuples, the slices' (status, length, checksums), and — only for members whose blocks read earlier blocks — the
 * ranks' 64 KiB index maps. */
typedef struct lfx_sharded_slice {
    uint64_t out_len, out_base;   /* this rank's slice: bytes, and where it lies in the member's output */
    uint64_t total_out;           /* bytes of the whole member's output */
    uint32_t crc32, adler32;      /* of the whole member's output (compare with the trailer) */
} lfx_sharded_slice;
void lfx_sharded_byte_range(uint64_t first_byte, uint64_t member_len, uint32_t rank, uint32_t world, uint64_t *lo_byte,
                            uint64_t *hi_byte, uint64_t *hold_hi);
int lfx_sharded_decode(lfx_ctx *c, const lfx_comm *cm, const void *d_part, uint64_t n_part, uint64_t lo_byte, uint64_t hi_byte,
                       uint64_t first_bit, uint64_t member_len, void *d_out, uint64_t cap, lfx_sharded_slice *out);
/* the exchange steps of the two drivers on their own (host memory + the comm; no device): the layout of the ranks' bit ranges
 * and the combined trailer checksum; the all-gather of candidate tuples (*all: lfx_sharded_free); the fold of the slices'
 * checksums.  `status`: this rank's error so far — a non-zero status of ANY rank is returned on EVERY rank. */
int lfx_sharded_layout(const lfx_comm *cm, const lfx_shard_info *mine, uint64_t header_len, int format,
                       uint64_t *start_bits /* world + 1 entries */, uint32_t *check, uint64_t *total_n);
int lfx_sharded_gather_tuples(const lfx_comm *cm, const lfx_blk_tuple *mine, uint32_t count, int status, lfx_blk_tuple **all,
                              uint32_t *n_all, uint32_t *failed_rank);
int lfx_sharded_fold(const lfx_comm *cm, int status, uint32_t state, uint64_t len, uint32_t crc32, uint32_t adler32,
                     uint32_t *any_state, uint32_t *crc_all, uint32_t *adler_all, uint64_t *total, uint32_t *failed_rank);
void lfx_sharded_free(void *p);
uint32_t lfx_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);
uint32_t lfx_adler32_combine(uint32_t ad1, uint32_t ad2, uint64_t len2);
uint64_t lfx_container_header_len(int format, const lfx_encode_opts *o);

/* ---- stream API: io::Write / io::Read shaped (host buffers, callbacks) --------------------
 * write_cb must consume all n bytes and return n, or a negative errno. */
typedef int64_t (*lfx_write_cb)(void *user, const uint8_t *p, size_t n);
typedef int (*lfx_flush_cb)(void *user);
/* read_cb fills up to cap bytes, returns bytes read (0 = EOF) or a negative errno */
typedef int64_t (*lfx_read_cb)(void *user, uint8_t *p, size_t cap);

typedef struct lfx_encoder lfx_encoder;
/* {deflate,zlib,gzip}::Encoder::with_options — gzip/zlib write their header immediately and
 * can fail (gzip.rs:804-812, zlib.rs:577-585). */
lfx_encoder *lfx_encoder_new(lfx_ctx *c, int format, const lfx_encode_opts *o, lfx_write_cb w,
                             lfx_flush_cb f, void *user, int *status);
/* io::Write::write — always consumes everything (encode.rs:241-244); one call = one write() */
int64_t lfx_encoder_write(lfx_encoder *e, const uint8_t *p, size_t n);
/* EncodeOptions::with_lz77(E) for a caller-supplied `E: Lz77Encode` (src/deflate/encode.rs:59-65; call sites
 * CompressBuf::{append, flush}, encode.rs:405-425: the reference Huffman-codes whatever E emits).  E runs on the caller's
 * side; this call hands over what it produced and the GPU does the rest (histogram, Huffman build, bit packing,
 * container checksum) exactly as for the built-in encoders:
 *   write(buf):  E::encode(buf, sink)  → lfx_encoder_write_codes(e, sink's codes, n, buf, len, 0)      (CompressBuf::append)
 *                then, while the bytes since the last close reach block_size (Block::write, encode.rs:282):
 *                E::flush(sink)        → lfx_encoder_write_codes(e, codes, n, NULL, 0, 1)              (CompressBuf::flush)
 *   flush():     E::flush(sink)        → lfx_encoder_write_codes(e, codes, n, NULL, 0, 1); lfx_encoder_flush(e)
 *   finish():    E::flush(sink)        → lfx_encoder_write_codes(e, codes, n, NULL, 0, 2); lfx_encoder_finish(e)
 * codes: (val << 16) | dist as in lfx_sink_cb below; end_block: 0 = the block stays open, 1 = EndOfBlock follows the
 * codes and the block closes, 2 = the same for the stream's final block.  raw: the bytes the codes stand for — only the
 * container checksum (and gzip's ISIZE) looks at them.  An encoder takes bytes (lfx_encoder_write) or codes, never
 * both; code words outside Code's domain (lib.rs:27-42) are LFX_E_ARG. */
int lfx_encoder_write_codes(lfx_encoder *e, const uint32_t *codes, size_t n_codes, const uint8_t *raw, size_t n_raw,
                            int end_block);
/* io::Write::flush — closes the current block (encode.rs:245-248); zlib Sync adds 00 00 FF FF */
int lfx_encoder_flush(lfx_encoder *e);
/* Encoder::finish (encode.rs:203-208, gzip.rs:858-868, zlib.rs:630-639). Everything already
 * handed to write_cb stays with the sink even on error (finish.rs:46-68). */
int lfx_encoder_finish(lfx_encoder *e);
const char *lfx_encoder_last_error(const lfx_encoder *e);
void lfx_encoder_free(lfx_encoder *e);

typedef struct lfx_decoder lfx_decoder;
/* {deflate,zlib,gzip}::Decoder::new / gzip::MultiDecoder::new (flag LFX_DEC_MULTI).  gzip / zlib read the container
 * header — and nothing that is not needed for it beyond the chunk the reader hands over — and can fail
 * (gzip.rs:941-944, zlib.rs:312-320); the body is decoded by the first read().
 * LFX_DEC_NONBLOCKING = src/non_blocking/{deflate,zlib,gzip}::Decoder: the read callback may return
 * -LFX_E_WOULD_BLOCK at any point; read() / header() then return LFX_E_WOULD_BLOCK and can be called again
 * (the header is read lazily, non_blocking/gzip.rs:64-113).
 *
 * The body is decoded a WINDOW at a time, as the reference decodes a block at a time (src/deflate/decode.rs:136-164,
 * 32 KiB of history: libflate_lz77/src/lib.rs:219-231): up to 16 MiB of input are pulled, the blocks that are complete
 * in them (and fit 96 MiB of output) are decoded and served, their input is dropped, the last 32 KiB of output stay as
 * history — what the decoder buffers (lfx_decoder_buffered) is bounded by one window of input plus one of output,
 * whatever the member's size, and the first byte is served as soon as the first window is decoded.  Only a block
 * larger than a window (members written by ONE huge write) makes the window grow.  A window is attempted when it is
 * full, the reader ends, hands over a short read, or would block.  Bytes pulled beyond the member's trailer are not
 * decoded: lfx_decoder_surplus() hands them back (what into_inner() means for a reader that cannot be rewound,
 * gzip.rs:987,1216-1226); a MultiDecoder continues with them. */
#define LFX_DEC_NONBLOCKING 2u
lfx_decoder *lfx_decoder_new(lfx_ctx *c, int format, uint32_t flags, lfx_read_cb r, void *user,
                             int *status);
/* io::Read::read: >0 bytes, 0 = end of stream, <0 = -(LFX_E_*) (reported once, after the bytes decoded in front of
 * the error).  A zero-capacity read returns 0 without latching end-of-stream (gzip.rs:1025-1027, zlib.rs:383-385). */
int64_t lfx_decoder_read(lfx_decoder *d, uint8_t *out, size_t cap);
/* Decoder::unread_decoded_data (decode.rs:68-73) */
int lfx_decoder_unread(lfx_decoder *d, const uint8_t **p, size_t *n);
/* bytes of the inner reader that belong to the members decoded so far (Decoder::into_inner position;
 * gzip.rs:1216-1226) */
uint64_t lfx_decoder_consumed(const lfx_decoder *d);
/* bytes the decoder holds right now: pulled input not yet decoded + decoded output not yet read + 32 KiB of history */
uint64_t lfx_decoder_buffered(const lfx_decoder *d);
/* bytes pulled from the reader behind the last finished member (valid until the next read()) */
int lfx_decoder_surplus(lfx_decoder *d, const uint8_t **p, size_t *n);
/* gzip::Header (gzip.rs:292-341: modification_time, compression_level (XFL), os, is_text, is_verified, extra_field,
 * filename, comment) / zlib::Header (zlib.rs:197-220: window_size, compression_level) of the current member.
 * Pointers stay valid until the next member starts or the decoder is freed. */
typedef struct lfx_header {
    int32_t format;
    uint32_t mtime;
    uint8_t xfl, os, is_text, is_verified, has_extra;
    uint8_t _pad[3];
    const uint8_t *extra;     /* serialized subfields (id[2] len[2] data)*; NULL when absent */
    uint32_t extra_len;
    const char *filename;     /* NUL-terminated; NULL when absent */
    const char *comment;
    uint32_t zlib_window_size; /* 256 << CINFO */
    uint32_t zlib_level;       /* FLEVEL 0..3 */
} lfx_header;
int lfx_decoder_header(lfx_decoder *d, lfx_header *h);
const char *lfx_decoder_last_error(const lfx_decoder *d);
void lfx_decoder_free(lfx_decoder *d);

/* ---- plug-in: libflate_lz77::Lz77Encode (libflate_lz77/src/lib.rs:83-107) -----------------
 * Codes are delivered in batches: word = (val << 16) | dist; dist == 0 → Code::Literal(val),
 * else Code::Pointer{length: val, backward_distance: dist} (lib.rs:27-42). */
typedef void (*lfx_sink_cb)(void *user, const uint32_t *codes, size_t n);
typedef struct lfx_lz77 lfx_lz77;
lfx_lz77 *lfx_lz77_new(lfx_ctx *c, uint32_t window_size, uint32_t max_length, int *status);
/* Lz77Encode::encode — buffers; flushes when >= window*8 bytes are held (default.rs:60-68) */
int lfx_lz77_encode(lfx_lz77 *z, const uint8_t *buf, size_t len, lfx_sink_cb sink, void *user);
/* Lz77Encode::flush (default.rs:69-109) */
int lfx_lz77_flush(lfx_lz77 *z, lfx_sink_cb sink, void *user);
uint32_t lfx_lz77_window_size(const lfx_lz77 *z); /* lib.rs:103-106 */
int lfx_lz77_compression_level(const lfx_lz77 *z); /* lib.rs:96-99 → LFX_LEVEL_BALANCE */
void lfx_lz77_free(lfx_lz77 *z);

/* ---- introspection used by tests / bench -------------------------------------------------- */
/* per-phase GPU milliseconds of the last encode/decode on this context (hipEvent-timed) */
typedef struct lfx_timing {
    float total_ms;
    float phase_ms[16];
    char phase_name[16][24];
    int n_phases;
} lfx_timing;
/* encode passes this context ran on the fallback match kernel because the ordered-LDS-exchange assumption of the default one
 *  was seen violated — 0 on every part measured so far; a non-zero value explains a 3x slower match stage */
uint64_t lfx_ctx_match_fallbacks(const lfx_ctx *c);
int lfx_ctx_last_timing(lfx_ctx *c, lfx_timing *t);
void lfx_ctx_enable_timing(lfx_ctx *c, int on);   /* 0: off; 1: a HIP event behind every phase of a call (lfx_ctx_last_timing);
                                                      2..5: only the two events around ONE kernel's phase — an event record between
                                                      two kernels costs ~6 us of idle GPU — 2: lz77_walk (parse_walk_kernel),
                                                      3: blk_scan, 4: lz77_cand (lz77_match7_kernel), 5: lz77_copy;
                                                      6: as 1, with the encode's match and parse phases split by kernel
                                                      (lz77_cand + lz77_resolve, lz77_walk + lz77_chain) */
uint32_t lfx_version(void);

#ifdef __cplusplus
}
#endif
#endif
