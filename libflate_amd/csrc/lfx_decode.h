// lfx_decode.h — descriptors of the inflate kernels (lfx_decode_kernels.hip) and their launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfx_common.h"

namespace lfx {

// error codes → reference message texts (formatted on the host, lfx_decode.cpp)
enum : uint32_t {
    ERR_NONE = 0,
    ERR_EOF,           // UnexpectedEof: "failed to fill whole buffer"
    ERR_HUFF,          // "Invalid huffman coded stream"                      huffman.rs:171-174
    ERR_CONFLICT,      // "Bit region conflict"                               huffman.rs:107-119
    ERR_HDIST,         // "The value of HDIST is too big: max=30, actual=N"   symbol.rs:395-403
    ERR_NO_PREV,       // "No preceding value"                                symbol.rs:470
    ERR_DIST_LIST,     // "The length of `distance_code_bitwidthes` is too large"  symbol.rs:433-443
    ERR_286,           // "The value N must not occur in compressed data"     symbol.rs:216-223
    ERR_BACKREF,       // "Too long backword reference: buffer.len=, distance="   lz77 lib.rs:173-185
    ERR_BTYPE3,        // "btype 0x11 of DEFLATE is reserved(error) value"    decode.rs:158-160
    ERR_LEN_NLEN,      // "LEN=.. is not the one's complement of NLEN=.."     decode.rs:88-93
    ERR_STORED_SHORT,  // "The reader has incorrect length: expected, read"   decode.rs:98-105
    ERR_NOSPACE,       // output capacity exhausted (ours)
    ERR_ZLIB_CHECK, ERR_METHOD, ERR_CINFO, ERR_FDICT,   // zlib.rs:229-259
    ERR_GZIP_ID, ERR_HCRC,                              // gzip.rs:398-441
    ERR_CRC32, ERR_ADLER32,                             // gzip.rs:1035-1040, zlib.rs:396-401
};

struct DecStream {
    uint64_t in_off, in_len;
    uint64_t out_off, out_cap;
};
struct DecHeader {
    uint64_t deflate_off;  // first DEFLATE byte relative to the stream start
    uint32_t status;       // 0 ok, 1 InvalidData, 2 UnexpectedEof
    uint32_t err, a0, a1;
    uint32_t flags;
    uint32_t _pad;
};

enum : uint32_t { JOB_SINGLE_BLOCK = 1, JOB_COUNT_ONLY = 2 };
struct InflateJob {
    uint64_t in_off;      // stream base
    uint64_t in_len;      // bytes available from in_off
    uint64_t start_bit;   // first bit to decode, relative to in_off
    uint64_t out_off, out_cap;
    uint64_t hist_avail;  // bytes of this member already produced before this job
    uint64_t stop_bit;    // != 0: the walk ends cleanly when a block ends exactly at this bit (a shard without BFINAL)
    uint32_t flags;
    uint32_t _pad;
};
struct InflateResult {
    uint64_t end_bit;  // bit after the last consumed bit (stored blocks: after the data)
    uint64_t out_len;
    uint32_t status;   // 0 ok, 1 InvalidData, 2 UnexpectedEof, 3 output capacity
    uint32_t final_seen;
    uint32_t err, a0, a1;
    uint32_t needs_hist;
    uint32_t nblocks;
    uint32_t _pad;
    uint64_t blk_out_start;  // output bytes before the block that was being decoded last
    uint64_t blk_start_bit;  // ... and the bit its header started at (a windowed decode resumes there)
};

// ---- lane-parallel single-stream path (lfx_inflate_fast.hip)
enum : uint32_t { BLK_OK = 0, BLK_BAD = 1, BLK_NO_EOB = 2 };
struct BlkJob {
    uint64_t start_bit;  // block header bit
    uint64_t end_bit;    // range end guess: the next candidate's start (or the end of the input)
    // PIECE of a huge block (piece != 0): the job scans [first symbol boundary >= lo_bit, end_bit) with the tables
    // of the block whose header is at start_bit.  That boundary is found by a warm-up decode from warm_bit
    // (a few Kbit earlier, a speculative start that is in step long before lo_bit); the host accepts the piece
    // only if the boundary equals the exit of the piece before it.  A piece without EndOfBlock is "open":
    // status BLK_NO_EOB, but lanes / counts / end_bit (exit of its last lane) are valid.
    // Piece 0 starts behind the header like any job (warm_bit = 0).
    uint64_t lo_bit, warm_bit;
    uint32_t piece, _pad;
    // the storing scan (round 6, launch_blk_scan_store): this job's lanes write their code words to temp + temp_off +
    // lane * cap (dwords); 0 = none
    uint64_t temp_off;
    uint32_t cap, _pad2;
};
struct BlkInfo {
    uint64_t end_bit;    // bit after EndOfBlock (stored: after the data)
    uint64_t n_out;      // bytes the block produces
    uint64_t data_bit;   // first symbol bit (stored: first data bit)
    uint32_t n_codes;
    uint32_t status, btype, bfinal, nlanes, rounds;
    uint32_t _pad;
    uint32_t cyc_hdr, cyc_total;   // shader-clock stamps (diagnostics)
};
struct BlkLanes {
    uint64_t start[1024];     // validated first bit of every lane's slice
    uint64_t out_off[1024];   // bytes of the block produced before the slice
    uint32_t code_off[1024];  // codes of the block before the slice
};
// what a storing scan leaves per lane for blk_place_kernel (indexed like BlkLanes)
struct BlkLanesX {
    uint32_t n_head[1024];    // the slice's codes: n_head of them from 0 of the lane's region ...
    uint32_t n_rest[1024];    // ... and n_rest from rest_at on
    uint32_t rest_at[1024];
    int32_t reach[1024];      // smallest (bytes of the slice produced before a match - its distance); INT32_MAX: no match
    uint32_t cut_code[1024];  // earliest cut of the slice no later code of the slice reads across: code index and byte offset,
    uint32_t cut_out[1024];   //   relative to the slice (cut_code 0xFFFFFFFF: none)
};
struct BlkEmit {
    uint64_t start_bit, data_bit;
    uint64_t code_off;   // first code slot of the block
    uint64_t out_off;    // first output byte of the block
    uint64_t n_out;
    uint32_t n_codes, nlanes, btype, cand;
    uint64_t hist;       // output bytes of the same member in front of the block (bounds its back-references)
    uint64_t end_limit;  // 0: the last lane decodes up to EndOfBlock; else (an open piece) up to this bit
    uint32_t preload;    // materialise: those bytes are already final in `out` — load up to 32 KiB of them as history
    uint32_t placed;     // round 6: 1 = the scan stored this block's codes (blk_place_kernel moves them; blk_emit_kernel skips the block)
    uint64_t temp_off;   // ... at temp + temp_off + lane * cap
    uint32_t cap, _pad;
};
constexpr uint32_t MAX_FREE_UNITS = 64;   // marker units per block (a schedule-S1 stream is ONE block)
struct BlkUnits {
    uint32_t n;          // independent units of the block (no back-reference crosses a cut)
    uint32_t code0[9];   // unit u covers codes [code0[u], code0[u+1]) of the block
    uint64_t out0[9];    // ... and bytes [out0[u], out0[u+1]) of the block's output
    uint32_t cyc[4];     // diagnostics: header, decode, cut search, unit selection (clock64 ticks)
    // the same block cut at slice boundaries WITHOUT regard to back-references (marker-based materialisation)
    uint32_t fn;
    uint32_t fcode0[MAX_FREE_UNITS + 1];
    uint64_t fout0[MAX_FREE_UNITS + 1];
};
// one unit of the marker-based path, in stream order
struct SymUnit {
    uint64_t start;      // first output byte
    uint64_t len;        // output bytes
};
// tabs: njobs * blk_tabs_bytes() bytes: the decode tables of every scanned block, reused by launch_blk_emit
size_t blk_tabs_bytes();
int launch_blk_scan(hipStream_t st, const uint8_t *in, uint64_t nbytes, const BlkJob *jobs, uint32_t njobs,
                    BlkInfo *infos, BlkLanes *lanes, void *tabs = nullptr,
                    bool small_blocks = false);      // round 6: 256 slices (and threads) a block instead of 1024: blocks of a few tens of KB
int launch_blk_scan_store(hipStream_t st, const uint8_t *in, uint64_t nbytes, const BlkJob *jobs, uint32_t njobs,
                          BlkInfo *infos, BlkLanes *lanes, void *tabs, uint32_t *temp, BlkLanesX *lanesx);
int launch_blk_place(hipStream_t st, const BlkEmit *jobs, uint32_t njobs, const BlkLanes *lanes, const BlkLanesX *lanesx,
                     const uint32_t *temp, uint32_t *codes, uint32_t *flags, BlkUnits *units, uint32_t unit_target,
                     uint32_t *job_flags, uint32_t free_shift);
int launch_blk_emit(hipStream_t st, const uint8_t *in, uint64_t nbytes, const BlkEmit *jobs, uint32_t njobs,
                    const BlkLanes *lanes, uint32_t *codes, uint32_t *flags, BlkUnits *units, uint32_t unit_target,
                    uint32_t *job_flags = nullptr,   // job_flags[j] = 1: block j reads bytes in front of itself
                    const void *tabs = nullptr,      // tables from launch_blk_scan, indexed by BlkEmit::cand
                    uint32_t free_shift = 17,        // marker units: 2^free_shift output bytes each (>= 15)
                    bool large_blocks = false,       // the kernel instance whose lanes read their bits through LDS rings (round 5): one
                                                     // workgroup per CU, faster per symbol — for blocks of tens of thousands of codes
                    bool small_blocks = false);      // the 256-lane instance: exactly the blocks that launch_blk_scan(small_blocks) scanned
int launch_blk_materialize(hipStream_t st, const uint8_t *in, const BlkEmit *jobs, uint32_t njobs,
                           const BlkLanes *lanes, const BlkUnits *units, const uint32_t *codes, uint8_t *out,
                           uint64_t *dbg);

// marker-based materialisation (streams whose blocks read earlier blocks):
//  sym: one 16-bit symbol per output byte — a byte value, or 256 + j = byte j of the 32 KiB in front of the unit
int launch_blk_materialize_sym(hipStream_t st, const uint8_t *in, const BlkEmit *jobs, uint32_t njobs,
                               const BlkUnits *units, const uint32_t *codes, uint16_t *sym,
                               bool few_units = false /* at most one unit per CU: the 1024-lane variant */);
//  windows[u] = the final 32 KiB of output up to the end of unit u (units in stream order, one workgroup walks them)
// init_win: the 32 KiB of output in front of the first unit (a later window of a member), or null (start of a member)
int launch_window_chain(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits, uint8_t *windows,
                        const uint8_t *init_win = nullptr);
//  the same windows by a blocked parallel prefix over the units' index maps (long streams)
size_t window_prefix_scratch_bytes(uint32_t nunits);
int launch_window_prefix(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits, void *scratch,
                         uint8_t *windows, const uint8_t *init_win = nullptr);
//  N-GPU decode, window hand-over: a rank's slice as ONE index map on the 32 KiB in front of it (from its symbol units, or
//  from its bytes when the slice was materialised directly), and the window in front of rank `nranks` from the maps of
//  the ranks before it (maps: nranks x 32768 entries in rank order)
int launch_window_rank_map(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits, void *scratch, uint16_t *out_map);
int launch_bytes_to_map(hipStream_t st, const uint8_t *out, uint64_t len, uint16_t *map);
int launch_window_ranks(hipStream_t st, const uint16_t *maps, uint32_t nranks, uint8_t *win);
//  out = sym with every marker replaced through the window in front of its unit
int launch_sym_substitute(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits,
                          const uint8_t *windows, uint8_t *out, uint64_t max_len, const uint8_t *init_win = nullptr);   // max_len: longest unit

int launch_container(hipStream_t st, int format, uint32_t count, const uint8_t *in,
                     const DecStream *streams, DecHeader *hdrs);
int launch_inflate(hipStream_t st, const uint8_t *in, uint8_t *out, const InflateJob *jobs,
                   InflateResult *results, uint32_t njobs);
// The finder's survivor lists: FIND_SHARDS lists of shard_cap entries — since round 6 a list holds ONE class of survivors
// (lfx_decode_kernels.hip: FIND_CLASSES classes x FIND_SUB lists each; a device-scope counter serialises at ~11 ns per atomic,
// and a workgroup of stage 1 now adds to about ten of them instead of one).  The buffer starts with FIND_HDR_WORDS 32-bit
// words: [0, FIND_SHARDS) the lists' counts, [FIND_SHARDS] a workgroup overflowed, FIND_HDR_FINAL the number of headers stage 2
// found, FIND_HDR_WORK.. its batch counters, FIND_HDR_DBG.. seven 64-bit LFX_DEBUG counters.
constexpr uint32_t FIND_SHARDS = 256;
constexpr uint32_t FIND_HDR_WORDS = 1024, FIND_HDR_FINAL = 320, FIND_HDR_DBG = 360, FIND_HDR_READ = 512 /* words the host reads back */;
// stage 2's batch counters: FIND2_GROUPS of them, 128 bytes apart.  ONE counter for all batches was the kernel's floor: eight
// thousand device-scope atomics on one address at ~20 ns each are 0.19 ms whatever the batches cost (measured with batches
// that do nothing, LFX_FIND2_EXP=3) — the wavefronts of group k deal the batches k, k + GROUPS, ... among themselves.
constexpr uint32_t FIND2_GROUPS = 16, FIND_HDR_WORK = 512, FIND_HDR_WORK_STRIDE = 32;
inline uint32_t find_shard_cap(uint64_t comp_bytes) {      // survivors are ~0.4 % of the bytes: room for 16 x a list's expected share
    const uint64_t want = comp_bytes / 4000;
    return (uint32_t)(want < 4096 ? 4096 : want > (1u << 23) ? (1u << 23) : want);
}
// count: FIND_SHARDS + 1 words (the last one marks a workgroup overflow)
int launch_find_stage1(hipStream_t st, const uint8_t *in, uint64_t nbytes, uint64_t first_byte,
                       uint32_t *count, uint64_t *cand, uint32_t shard_cap, uint64_t final_from_bit, uint32_t n_cu);
// survivors of the full header check are appended to final[] (final_count = number appended)
int launch_find_stage2(hipStream_t st, const uint8_t *in, uint64_t nbytes, const uint64_t *cand,
                       uint32_t shard_cap, const uint32_t *count /* device: stage 1's FIND_SHARDS + 1 words */,
                       uint32_t *work /* device, zero: the batch counter of the persistent grid */,
                       uint32_t *final_count, uint64_t *final_list, uint32_t final_cap, uint32_t n_cu,
                       uint64_t *dbg = nullptr /* device, zero: seven counters (LFX_DEBUG) */,
                       int exp = 0 /* timing experiments: a cut-down kernel that finds nothing (LFX_FIND2_EXP) */);
int launch_verify_trailers(hipStream_t st, int format, uint32_t count, const uint8_t *in,
                           const DecStream *streams, const DecHeader *hdrs, InflateResult *results,
                           const uint32_t *crc, const uint32_t *adler, uint64_t *consumed);
}  // namespace lfx
