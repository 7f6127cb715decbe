// lfx_plan.h — host-side planner: turns (options, sequence of write()/flush() events) into the
// LZ77 chunk list and DEFLATE block list the reference would produce for the same calls.
// Pure index arithmetic — no data is touched on the host.
//
// Mirrors: DefaultLz77Encoder::encode (libflate_lz77/src/default.rs:60-68),
//          Block::{write,flush,finish} (src/deflate/encode.rs:277-303),
//          CompressBuf::{append,flush} (encode.rs:405-425), RawBuf::flush (encode.rs:364-382),
//          Encoder::{flush,zlib_sync_flush} (encode.rs:225-249).
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "lfx_common.h"

namespace lfx {

struct PlanOpts {
    uint64_t block_size = 1u << 20;
    bool dynamic_huffman = true;
    bool no_compression = false;
    int lz77_kind = 0;  // 0 default, 1 NoCompressionLz77Encoder
    uint32_t window_size = MAX_WINDOW;
    uint32_t max_length = MAX_LENGTH;
    bool zlib_sync = false;  // zlib FlushMode::Sync (only meaningful for the zlib container)
};

struct Plan {
    std::vector<ChunkDesc> chunks;
    std::vector<BlockDesc> blocks;
    uint64_t n_codes_cap = 0;  // slots needed in the code array
    uint64_t n_tiles = 0;      // pack tiles (upper bound)
    uint64_t n_vis = 0;        // visited-mask words
    uint32_t n_segs = 0;       // parse segments
};

class Planner {
  public:
    explicit Planner(const PlanOpts &o) : o_(o) {
        raw_ = o.no_compression;
        block_type_ = raw_ ? BT_RAW : (o.dynamic_huffman ? BT_DYNAMIC : BT_FIXED);
        // EncodeOptions::get_block_size encode.rs:121-127
        block_size_ = raw_ ? (o.block_size < 0xFFFF ? o.block_size : 0xFFFF) : o.block_size;
    }
    // one Write::write(buf) of n bytes
    void write(uint64_t n) {
        cursor_ += n;
        if (raw_) {
            raw_len_ += n;
            while (raw_len_ >= block_size_) raw_block(false);  // encode.rs:282
            return;
        }
        original_size_ += n;  // CompressBuf::append encode.rs:405-408
        lz_len_ += n;
        if (o_.lz77_kind == 0 && lz_len_ >= (uint64_t)o_.window_size * 8) lz_flush();  // default.rs:65
        while (original_size_ >= block_size_) block_flush(false);
    }
    // `count` consecutive write() calls of n bytes each — the reference's io::copy protocol is 32768 writes of 8 KiB per
    // 256 MiB, 157 us of this loop per encode call when each goes through write().  Between two events (an LZ77 flush at
    // window * 8 buffered bytes, a block at block_size) a write only moves counters: those writes are taken in one step, the
    // write that triggers an event goes through write() itself.
    void write_repeat(uint64_t n, uint64_t count) {
        if (n == 0) return;  // (a write of nothing moves no counter: the thresholds were settled by the write before)
        while (count) {
            // writes that can pass without reaching a threshold: the state after k of them is (x + k n) for every counter
            uint64_t quiet;
            if (raw_) quiet = raw_len_ < block_size_ ? (block_size_ - raw_len_ - 1) / n : 0;      // raw_len_ + k n < block_size_
            else {
                quiet = original_size_ < block_size_ ? (block_size_ - original_size_ - 1) / n : 0;
                if (o_.lz77_kind == 0) {
                    const uint64_t w8 = (uint64_t)o_.window_size * 8;
                    const uint64_t q2 = lz_len_ < w8 ? (w8 - lz_len_ - 1) / n : 0;
                    quiet = quiet < q2 ? quiet : q2;
                }
            }
            if (quiet > count) quiet = count;
            if (quiet) {
                const uint64_t bytes = quiet * n;
                cursor_ += bytes;
                if (raw_) raw_len_ += bytes;
                else { original_size_ += bytes; lz_len_ += bytes; }
                count -= quiet;
            }
            if (count) { write(n); --count; }
        }
    }
    // Write::flush
    void flush() {
        if (raw_) raw_block(false); else block_flush(false);
        if (o_.zlib_sync) {  // zlib_sync_flush encode.rs:225-234 == an empty stored block
            BlockDesc b{};
            b.type = BT_RAW;
            b.in_off = cursor_;
            b.in_len = 0;
            b.first_chunk = (uint32_t)plan_.chunks.size();
            plan_.blocks.push_back(b);
        }
    }
    // Encoder::finish
    Plan &finish() {
        if (raw_) raw_block(true); else block_flush(true);
        plan_.blocks.back().align_after = 1;  // BitWriter::flush encode.rs:301
        plan_.n_codes_cap = code_cursor_;
        plan_.n_tiles = tile_cursor_;
        return plan_;
    }
    Plan &plan() { return plan_; }
    uint64_t cursor() const { return cursor_; }

    // ---- incremental use (stream encoder): blocks that are already closed can be encoded while later writes are
    // still being collected.  closed_bytes() = input bytes covered by closed blocks; take_closed() moves those
    // blocks (and their chunks) out and rebases what remains — the open block's finished chunks, the partial chunk
    // and all counters — to start at byte 0, exactly as if the stream had begun there in the same state.
    uint64_t closed_bytes() const {
        if (raw_) return cursor_ - raw_len_;
        return plan_.chunks.size() > first_chunk_ ? plan_.chunks[first_chunk_].in_off : cursor_ - lz_len_;
    }
    size_t closed_blocks() const { return plan_.blocks.size(); }
    Plan take_closed() {
        Plan out;
        const uint64_t cut = closed_bytes();
        out.blocks = std::move(plan_.blocks);
        plan_.blocks.clear();
        out.chunks.assign(plan_.chunks.begin(), plan_.chunks.begin() + (std::ptrdiff_t)first_chunk_);
        plan_.chunks.erase(plan_.chunks.begin(), plan_.chunks.begin() + (std::ptrdiff_t)first_chunk_);
        // what the closed part used of the code / tile / mask / segment spaces
        uint64_t code0 = code_cursor_, tile0 = tile_cursor_, vis0 = plan_.n_vis;
        uint32_t seg0 = plan_.n_segs;
        if (!plan_.chunks.empty()) {
            code0 = plan_.chunks[0].code_off; tile0 = plan_.chunks[0].tile_base;
            vis0 = plan_.chunks[0].vis_base; seg0 = plan_.chunks[0].seg_base;
        }
        out.n_codes_cap = code0; out.n_tiles = tile0; out.n_vis = vis0; out.n_segs = seg0;
        for (ChunkDesc &c : plan_.chunks) {
            c.in_off -= cut; c.code_off -= code0; c.tile_base -= tile0; c.vis_base -= vis0; c.seg_base -= seg0;
            c.block = 0;    // (the open block becomes block 0)
        }
        code_cursor_ -= code0; tile_cursor_ -= tile0; plan_.n_vis -= vis0; plan_.n_segs -= seg0;
        cursor_ -= cut;
        first_chunk_ = 0;
        return out;
    }

  private:
    void emit_chunk(uint64_t off, uint64_t len, uint32_t flags) {
        ChunkDesc c{};
        c.in_off = off;
        c.len = len;
        c.code_off = code_cursor_;
        c.block = (uint32_t)plan_.blocks.size();
        c.flags = flags;
        c.tile_base = tile_cursor_;
        c.vis_base = plan_.n_vis;
        c.seg_base = plan_.n_segs;
        c.n_seg = (uint32_t)div_up(len, PARSE_SEG);
        plan_.n_vis += (uint64_t)c.n_seg * 64;   // one mask word per group of PARSE_GROUP positions
        plan_.n_segs += c.n_seg;
        code_cursor_ += len + 1;  // worst case all literals + a possible EndOfBlock
        tile_cursor_ += div_up(len + 1, PACK_TILE);
        plan_.chunks.push_back(c);
    }
    void lz_flush() {
        // DefaultLz77Encoder::flush; an empty buffer produces no codes (default.rs:75: end = 0)
        if (lz_len_ > 0) emit_chunk(cursor_ - lz_len_, lz_len_, o_.lz77_kind == 1 ? CH_LITERALS : 0);
        lz_len_ = 0;
    }
    void block_flush(bool final) {
        lz_flush();  // CompressBuf::flush encode.rs:416
        if (plan_.chunks.size() == first_chunk_) emit_chunk(cursor_, 0, 0);  // block with only EOB
        plan_.chunks.back().flags |= CH_LAST_IN_BLOCK;  // encode.rs:417
        BlockDesc b{};
        b.type = block_type_;
        b.final = final;
        b.first_chunk = (uint32_t)first_chunk_;
        b.n_chunks = (uint32_t)(plan_.chunks.size() - first_chunk_);
        b.in_off = plan_.chunks[first_chunk_].in_off;
        b.in_len = cursor_ - b.in_off;
        plan_.blocks.push_back(b);
        first_chunk_ = plan_.chunks.size();
        original_size_ = 0;  // encode.rs:423
    }
    void raw_block(bool final) {
        uint64_t size = raw_len_ < 0xFFFF ? raw_len_ : 0xFFFF;  // encode.rs:368
        BlockDesc b{};
        b.type = BT_RAW;
        b.final = final;
        b.in_off = cursor_ - raw_len_;
        b.in_len = size;
        b.first_chunk = (uint32_t)plan_.chunks.size();
        plan_.blocks.push_back(b);
        raw_len_ -= size;
    }

    PlanOpts o_;
    Plan plan_;
    bool raw_ = false;
    uint32_t block_type_ = BT_DYNAMIC;
    uint64_t block_size_ = 0;
    uint64_t cursor_ = 0, raw_len_ = 0, original_size_ = 0, lz_len_ = 0;
    uint64_t code_cursor_ = 0, tile_cursor_ = 0;
    size_t first_chunk_ = 0;
};

}  // namespace lfx
