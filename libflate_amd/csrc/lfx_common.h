// lfx_common.h — descriptors shared by the host planner and the HIP kernels.
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LFX_HD __host__ __device__
#else
#define LFX_HD
#endif

namespace lfx {

// DEFLATE block types (reference src/deflate/mod.rs:34-39)
enum : uint32_t { BT_RAW = 0, BT_FIXED = 1, BT_DYNAMIC = 2 };

constexpr uint32_t MAX_WINDOW = 32768;  // libflate_lz77/src/lib.rs:21-24
constexpr uint32_t MAX_LENGTH = 258;    // lib.rs:18
constexpr uint32_t CODE_EOB = 0x01000000u;  // (256 << 16) | 0 : Symbol::EndOfBlock as a code word

// One LZ77 flush unit (DefaultLz77Encoder::flush, default.rs:69-109): no match crosses it.
struct ChunkDesc {
    uint64_t in_off;    // first input byte
    uint64_t len;       // bytes (< 4 GiB: positions are u32 in the reference, default.rs:78)
    uint64_t code_off;  // first slot of this chunk in the code array
    uint32_t block;     // owning block
    uint32_t flags;     // CH_*
    uint64_t tile_base; // first pack tile of this chunk (prefix of ceil((len+1)/TILE))
    uint64_t vis_base;  // first 64-position visited-mask word of this chunk
    uint32_t seg_base;  // first parse segment of this chunk
    uint32_t n_seg;     // parse segments (PARSE_SEG positions each)
};
enum : uint32_t {
    CH_LAST_IN_BLOCK = 1,  // parse appends Symbol::EndOfBlock (encode.rs:417)
    CH_LITERALS = 2,       // NoCompressionLz77Encoder: every byte a literal (lib.rs:127-135)
};

// One DEFLATE block (Block::flush, encode.rs:287-295)
struct BlockDesc {
    uint64_t in_off, in_len;  // RAW: the bytes stored; compressed: bytes covered (informative)
    uint32_t first_chunk, n_chunks;
    uint32_t type;   // BT_*
    uint32_t final;  // BFINAL
    uint32_t align_after;  // 1: byte-align after this block (Block::finish → BitWriter::flush)
    uint32_t _pad;
};

// per-block result of the Huffman stage
struct BlockCodes {
    uint32_t lit[288];   // (bits | width << 16), bits already reversed for LSB-first emission
    uint32_t dist[32];
    uint64_t body_bits;  // 3 + header bits + symbol bits (compressed blocks)
    uint32_t hdr_bits;   // dynamic header bits (after the 3 block bits)
    uint32_t _pad;
    uint32_t hdr[160];   // header bit string, LSB-first, up to 5120 bits
};

constexpr uint32_t PACK_TILE = 2048;  // codes per pack tile
// Speculative parse segments: one wavefront per segment, 64 lanes ("groups") of PARSE_GROUP positions each; a group's
// visited mask is one 64-bit word of vis[].  52 positions = 13 dwords per lane: an odd dword stride, so that 32 lanes
// reading at the same offset inside their groups hit 32 different LDS banks (lfx_parse2.hip).
#ifndef LFX_PARSE_GROUP
#define LFX_PARSE_GROUP 52
#endif
constexpr uint32_t PARSE_GROUP = LFX_PARSE_GROUP;
constexpr uint32_t PARSE_SEG = 64 * PARSE_GROUP;   // 3328 positions
// Segments (wavefronts) per workgroup of the walk kernel.  Rounds 3-5: 4 (72 KB of LDS, two workgroups per CU = two
// wavefronts per SIMD; 8 — one workgroup per CU, the same occupancy — measured 1.14 ms against 1.10).  Round 6: 12 — the 32 KiB
// look-back is paid once for twelve segments, 153 KB of LDS, ONE workgroup per CU but THREE wavefronts per SIMD: the walk is
// VALU-bound with its wavefronts parked 39 % of the time, a third one fills the gaps (parse 0.954 → 0.896 ms; 10: 0.974).
#ifndef LFX_PARSE_WG_SEGS
#define LFX_PARSE_WG_SEGS 12
#endif
constexpr uint32_t PARSE_WG_SEGS = LFX_PARSE_WG_SEGS;
// parse_emit_hist_kernel (round 6): wavefronts per workgroup, and workgroups per CU the host aims at when it sizes a
// workgroup's share of segments (every workgroup ends with a global atomic per non-zero symbol counter)
#ifndef LFX_EMIT_WAVES
#define LFX_EMIT_WAVES 4
#endif
#ifndef LFX_EMIT_WG_PER_CU
#define LFX_EMIT_WG_PER_CU 16
#endif
constexpr uint32_t PARSE_EMIT_WAVES = LFX_EMIT_WAVES, PARSE_EMIT_WG_PER_CU = LFX_EMIT_WG_PER_CU;

#ifdef __HIPCC__
// pointers that are known to address global memory (HBM): keeps loads on the global_load path —
// a pointer rebuilt from an integer becomes `flat`, and flat loads also tick lgkmcnt, which makes
// every LDS wait drain outstanding memory prefetches
typedef const __attribute__((address_space(1))) uint32_t *gptr_u32;
typedef const __attribute__((address_space(1))) uint8_t *gptr_u8;
#endif

LFX_HD inline uint64_t div_up(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

}  // namespace lfx
