// lfx_device.h — device-side descriptors and the kernel launchers (lfx_*_kernels.hip) used by
// lfx_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfx_common.h"

namespace lfx {

// one workgroup's share of the match search: positions [start, start+len) of a chunk
struct SegDesc {
    uint32_t chunk;
    uint32_t start;
    uint32_t len;
    uint32_t lnk_base;   // lfx_match7 / lfx_match5: first entry of the segment's private link region, in units of 64 entries (128 bytes)
};
constexpr uint32_t SEG_POSITIONS = 256 * 1024;

struct EncodeResult {
    uint64_t end_bit;    // bit after the last DEFLATE bit (absolute, container header included)
    uint64_t out_bytes;  // bytes of output relative to the output base
    uint32_t status;     // 0 ok, 1 capacity exceeded
    uint32_t crc32;
    uint32_t adler32;
    uint32_t match_flags;  // bit 0: the second-generation match kernel saw an LDS lane-order violation (results void);
                           // bit 1: one of its segments held runs of equal bytes (its sample) — the parse walk takes its instance for such data
};

// one stream of a batch encode: where its bytes lie, where its output goes, which blocks of the merged plan are its own
struct BatchStream {
    uint64_t in_off, in_len;
    uint64_t out_off, out_cap;
    uint32_t first_block, n_blocks;
};

// one workgroup of the parse walk: PARSE_WG_SEGS consecutive parse segments of one chunk, from segment seg0 on
struct ParseWg {
    uint32_t chunk;
    uint32_t seg0;
};

int launch_match(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks,
                 const SegDesc *segs, uint32_t nsegs, uint32_t window, uint32_t max_len, uint32_t *md,
                 uint64_t *dbg = nullptr);
// candidate stage: per position the distance to the most recent earlier occurrence of its 3-byte prefix (0 = none) → cd.
// flags[0] |= 1 on a lane-order violation.
// lfx_match7.hip (round 5, the default): two-level bucket LRU with exact tags + the resolver kernel for the few positions
// it leaves open.  glnk: scratch for the duplicate-collapsed link of every position of every segment (warm-up included),
// regions by SegDesc::lnk_base
int launch_match7(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint16_t *cd, uint32_t *glnk /* link | tag << 16 per link entry */,
                  uint64_t *umask /* 8 bytes per 64 link entries */, uint32_t *flags, uint64_t *dbg = nullptr);
// ... and the positions it leaves open (ballot words → a list per segment → chain walks through glnk): may run on another
// stream, behind the candidate kernel of the same segments.  ucount: zeroed by the caller.
int launch_resolve7(hipStream_t st, const ChunkDesc *chunks, const SegDesc *segs, uint32_t nsegs, uint32_t window, uint16_t *cd,
                    const uint32_t *glnk, const uint64_t *umask, uint32_t *ulist /* 4 bytes per input byte */,
                    uint32_t *ucount /* 4 bytes per segment */,
                    uint32_t hop_cap = 0 /* diagnostics (LFX_R7_CAP): end every walk after so many hops — wrong answers, timing only */);
// lfx_match5.hip (round 4; LFX_MATCH_V5=1): hash heads + window ring + link ring, deep chain walks handed over to wave 0
int launch_match5(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint16_t *cd, uint16_t *glnk, uint32_t *flags, uint64_t *dbg = nullptr);
// the first-generation kernel's answers (length << 16 | distance) → cd
int launch_md_to_cd(hipStream_t st, const uint32_t *md, uint64_t n, uint16_t *cd);
// the greedy walk with lazy match lengths (lfx_parse2.hip) → code words per chunk
int launch_parse(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, uint32_t nchunks,
                 uint32_t nsegs, const ParseWg *wgs, uint32_t nwgs, const uint16_t *cd, uint32_t max_len, uint64_t *vis,
                 uint32_t *seg_tmp, uint32_t *codes, uint32_t *ncodes, uint32_t *stage /* 4 bytes per input byte */,
                 const uint32_t *seg_map, int stop_after = 0 /* diagnostics: 1 = behind the walk, 2 = behind fixseg */,
                 uint64_t *dbg = nullptr /* LFX_DEBUG: cycle stamps of one walk workgroup */,
                 uint32_t *hist = nullptr /* 320 zeroed counters per block: the code words are counted as they are emitted
                                             (launch_histogram is then not needed) */,
                 uint32_t emit_per = 0 /* with hist: segments per emit workgroup ... */,
                 uint32_t emit_parts = 0 /* ... and workgroups for the chunk of most segments (grid = nchunks x emit_parts) */,
                 hipEvent_t ev_walked = nullptr /* recorded behind the walk kernel (in front of the chaining kernels) */,
                 const uint32_t *mflags = nullptr /* EncodeResult::match_flags of this call (bit 1 picks the walk's instance) */,
                 int start_at = 0 /* 1: the walk kernel has been launched (by a call with stop_after = 1): the chaining kernels only */);
struct ZeroSpan { uint32_t *p; uint32_t n; };      // n words at p to be cleared (by the kernel that runs first anyway)
int launch_chunk_maps(hipStream_t st, const ChunkDesc *chunks, uint32_t nchunks, uint64_t ntiles, uint32_t nsegs,
                      uint32_t *tile_map, uint32_t *seg_map, ZeroSpan z0 = ZeroSpan{nullptr, 0}, ZeroSpan z1 = ZeroSpan{nullptr, 0},
                      ZeroSpan z2 = ZeroSpan{nullptr, 0});
int launch_histogram(hipStream_t st, const ChunkDesc *chunks, uint32_t nchunks, uint32_t split,
                     const uint32_t *codes, const uint32_t *ncodes, uint32_t *hist);
int launch_huffman(hipStream_t st, const BlockDesc *blocks, uint32_t nblocks, const uint32_t *hist,
                   BlockCodes *bc, uint64_t *dbg = nullptr);
int launch_offsets(hipStream_t st, const BlockDesc *blocks, uint32_t nblocks, const BlockCodes *bc,
                   uint64_t start_bit, uint64_t cap_bits, uint64_t *block_start, EncodeResult *res);
int launch_offsets_batch(hipStream_t st, const BatchStream *streams, uint32_t count, const BlockDesc *blocks, const BlockCodes *bc,
                         uint32_t hdr_len, uint32_t trailer_len, uint64_t *block_start, uint64_t *stream_end, int32_t *status,
                         EncodeResult *res);
int launch_frame_batch(hipStream_t st, int format, const BatchStream *streams, uint32_t count, const uint8_t *hdr, uint32_t hdr_len,
                       const uint64_t *stream_end, const uint32_t *crc, const uint32_t *adler, const EncodeResult *res, uint32_t *out,
                       uint64_t *out_len);
int launch_pack(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks,
                uint32_t nchunks, const BlockDesc *blocks, uint32_t nblocks, uint64_t ntiles,
                const uint32_t *codes, const uint32_t *ncodes, const BlockCodes *bc,
                const uint64_t *block_start, uint32_t *tile_bits, uint64_t *tile_start,
                const EncodeResult *res, uint64_t out_base_bit, uint32_t *out, const uint32_t *tile_map);
// span of the checksum kernels = the bytes one wavefront folds serially: 64 KiB; 8 KiB up to 64 MiB (a 1 MiB buffer is
// sixteen 64 KiB spans — sixteen wavefronts on the whole GPU, each walking sixteen dependent pieces: 0.12 ms; 16 MiB are 256)
inline uint32_t ck_span(uint64_t n) { return n <= (64ull << 20) ? 8192u : 65536u; }
inline uint64_t ck_nspans(uint64_t n) { return div_up(n ? n : 1, ck_span(n)); }   // partials needed: 3 x 4 bytes each
int launch_checksum(hipStream_t st, const uint8_t *in, uint64_t n, uint32_t *crc_part,
                    uint32_t *a_part, uint32_t *b_part, EncodeResult *res,
                    int mode = 3)   /* bit 0: CRC-32, bit 1: Adler-32 (the other result is then 0) */;
// the same sweep in `nparts` launches (the caller orders them; the last one folds all spans)
int launch_checksum_part(hipStream_t st, const uint8_t *in, uint64_t n, uint32_t *crc_part, uint32_t *a_part, uint32_t *b_part,
                         EncodeResult *res, int mode, uint32_t part, uint32_t nparts);
// CRC-32 / Adler-32 of count byte ranges data[off[i*off_stride] .. +len[i*len_stride]) (strides in 8-byte units)
int launch_checksum_ranges(hipStream_t st, const uint8_t *data, uint32_t count, const uint64_t *off,
                           uint32_t off_stride, const uint64_t *len, uint32_t len_stride, uint32_t *crc, uint32_t *adler);
int launch_or_byte(hipStream_t st, uint8_t *dst, const uint8_t *src);
int launch_put_bytes(hipStream_t st, const uint8_t *d_bytes, uint32_t n, uint64_t at_byte, uint32_t *out);
int launch_trailer(hipStream_t st, int format, uint32_t isize, uint64_t out_base_bit,
                   EncodeResult *res, uint32_t *out);

}  // namespace lfx
