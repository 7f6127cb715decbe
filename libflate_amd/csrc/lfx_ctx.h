// lfx_ctx.h — per-GPU context: HIP stream, grow-only scratch in HBM, last error, phase timers.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "lfx_common.h"
#include "lfx_device.h"
#include <string.h>

#include "lfx_hostio.h"

namespace lfx {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    uint32_t gen = 0;           // bumped by every (re)allocation: a cached "this table is already there" must not survive one
    int reserve(size_t bytes);  // grow-only; contents are NOT preserved
    void release();
};

// Diagnostic switches, read from the environment ONCE when the context is created — never on the
// encode / decode paths.
struct Diag {
    bool debug = false;          // LFX_DEBUG: per-stage counters and cycle stamps on stderr
    bool match_v1 = false;       // LFX_MATCH_V1: first-generation match kernel (+ md → cd)
    bool match_v5 = false;       // LFX_MATCH_V5: lfx_match5.hip (round 4: hash heads, window ring, link ring)
    int r7_cap = 0;              // LFX_R7_CAP: lfx_match7's resolver ends every walk after so many hops (WRONG answers: timing experiments only)
    int match_parts = 1;         // LFX_MATCH_PARTS: lfx_match7 in up to four launches, each part's resolver on the side stream
    bool no_serial = false;      // LFX_NO_SERIAL: the serial fallback of the single-stream decoder is an error
    bool batch_serial = false;   // LFX_BATCH_SERIAL: every stream of a batch through the serial kernel
    bool no_markers = false;     // LFX_NO_MARKERS
    bool no_pieces = false;      // LFX_NO_PIECES
    bool no_final_cand = false;  // LFX_NO_FINAL_CAND: the finder reports no BFINAL header at all (the chain walk scans the last block on demand)
    bool window_chain = false;   // LFX_WINDOW_CHAIN
    int free_shift = -1;         // LFX_FREE_SHIFT
    bool no_small_scan = false;  // LFX_NO_SMALL_SCAN: 1024 slices a block also for small blocks (round 5's geometry)
    bool two_pass = false;       // LFX_TWO_PASS: every block through blk_emit_kernel (no storing scan)
    bool no_pin_slots = false;   // LFX_NO_PIN_SLOTS: the decode's small transfers as plain pageable copies (the path of a full arena)
    int find2_exp = 0;           // LFX_FIND2_EXP=1..4: a cut-down finder stage 2 runs in front of the real one (phase "find2x"): timing only
    bool hist_separate = false;  // LFX_HIST_SEPARATE: the blocks' symbol counts by histogram_kernel (round 5) instead of inside parse_emit
    bool store_tight = false;    // LFX_STORE_TIGHT: the storing scan's regions sized for 16 bits a code (tests: lanes overflow, blocks fall back)
    int enc_batch_mb = 0;        // LFX_ENC_BATCH_MB: the stream encoder encodes closed blocks once so many MiB wait (0: the default, 8)
    int pocr_max = 100;          // LFX_POCR_MAX: most candidate ranges the decoder scans in pieces at once
    void read();
};

// N-GPU member decode: what lfx_decode_range_emit leaves for lfx_decode_range_map / lfx_decode_range_finish
struct RangeState {
    int state = -1;            // -1 none, 0 the slice holds bytes, 1 the slice is held as 16-bit symbols (d_dec_sym) and waits for its window
    uint64_t total = 0, before = 0, max_len = 0;   // slice bytes, member bytes in front of it, longest symbol unit
    uint8_t *d_out = nullptr;
    uint32_t nsu = 0;          // symbol units (their table lives in d_dec_win behind the windows)
};

struct Ctx {
    int device = 0;
    RangeState range;
    // lfx_decode_range_scan → lfx_decode_range_emit: which scan jobs stored their code words (the storing scan), and where
    std::vector<uint8_t> range_stored;
    std::vector<uint64_t> range_temp_off;
    std::vector<uint32_t> range_cap;
    Diag diag;
    int n_cu = 0;   // compute units of the device
    hipStream_t own_stream = nullptr, stream = nullptr;
    // work that only depends on the input (the container checksum) runs beside the thin kernels of the
    // main stream: fork / join through these two events
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_zero = nullptr;
    hipEvent_t ev_part[4] = {}, ev_res = nullptr;   // lfx_match7: the resolver of one part of the segments runs beside the next part's kernel
    // one-shot encode: the output buffer is zero-filled on the side stream while the match kernel runs
    const void *prezero_ptr = nullptr;
    uint64_t prezero_bytes = 0;
    // Every entry point that touches the context's scratch holds this lock: handles (encoders, decoders, LZ77 plug-ins)
    // of one context may live on different threads, they simply take turns on the GPU (SURVEY §8b threading row).
    std::recursive_mutex mu;
    std::string err;
    void set_error(const std::string &e) { err = e; }

    // encode scratch
    DevBuf d_chunks, d_blocks, d_segs, d_pwgs, d_cd, d_md, d_codes, d_ncodes, d_hist, d_bc, d_block_start, d_tile_bits,
        d_tile_start, d_ck, d_res, d_small, d_hdr, d_io_in, d_io_out, d_vis, d_segtmp, d_stage, d_chunkmap, d_glnk, d_ucount;
    // decode scratch
    // host shadows of the plan tables last uploaded (chunks, blocks, segments, parse workgroups) and the device buffers they
    // went to: an encode with the same plan (same size, schedule and options — every step of a loop) uploads nothing, and a
    // different plan is copied from the shadow, which outlives the asynchronous copy (no synchronisation either way)
    std::vector<uint8_t> up_shadow[4];
    uint64_t up_dev[4] = {0, 0, 0, 0};      // (buffer address ^ allocation generation << 48)
    DevBuf d_dec_streams, d_dec_state, d_dec_tmp, d_dec_cand, d_dec_blocks, d_dec_tabs, d_dec_sym, d_dec_win, d_dec_maps,
        d_dec_temp, d_dec_lanesx;     // round 6: the storing scan's per-lane code regions and BlkLanesX records
    std::vector<DevBuf *> all_bufs() {
        return {&d_chunks, &d_blocks, &d_segs, &d_pwgs, &d_cd, &d_md, &d_codes, &d_ncodes, &d_hist, &d_bc, &d_block_start,
                &d_tile_bits, &d_tile_start, &d_ck, &d_res, &d_small, &d_hdr, &d_io_in, &d_io_out, &d_vis, &d_segtmp, &d_stage, &d_chunkmap, &d_glnk, &d_ucount,
                &d_dec_streams, &d_dec_state, &d_dec_tmp, &d_dec_cand, &d_dec_blocks, &d_dec_tabs, &d_dec_sym, &d_dec_win, &d_dec_maps,
                &d_dec_temp, &d_dec_lanesx};
    }
    void *h_res = nullptr;  // pinned, 4 KiB (+ the arena below)
    // Small transfers of the decode paths (job lists up, scan results and counters down) go through page-locked memory: a
    // pageable hipMemcpyAsync is staged by the runtime on the calling thread, 20-45 us of idle GPU per transfer in the kernel
    // trace (round 6: 170 us of a 2.4 ms decode).  An arena behind h_res, handed out in slots, reset per member; a transfer
    // that does not fit goes the old way.
    static constexpr size_t PIN_ARENA = 1u << 20;
    size_t pin_used = 0;
    struct PinDown { void *slot, *dst; size_t n; };
    std::vector<PinDown> pin_pending;
    void pin_reset() { pin_used = 0; pin_pending.clear(); }
    void *pin_take(size_t n) {
        const size_t at = (pin_used + 63) & ~(size_t)63;
        if (!h_res || diag.no_pin_slots || at + n > PIN_ARENA) return nullptr;
        pin_used = at + n;
        return (uint8_t *)h_res + 4096 + at;
    }
    hipError_t small_up(void *dev, const void *host, size_t n, hipStream_t st) {
        if (!n) return hipSuccess;
        void *slot = pin_take(n);
        if (!slot) return hipMemcpyAsync(dev, host, n, hipMemcpyHostToDevice, st);
        memcpy(slot, host, n);
        return hipMemcpyAsync(dev, slot, n, hipMemcpyHostToDevice, st);
    }
    // host[0, n) <- dev[0, n): complete behind small_sync()
    hipError_t small_down(void *host, const void *dev, size_t n, hipStream_t st) {
        if (!n) return hipSuccess;
        void *slot = pin_take(n);
        if (!slot) return hipMemcpyAsync(host, dev, n, hipMemcpyDeviceToHost, st);
        pin_pending.push_back({slot, host, n});
        return hipMemcpyAsync(slot, dev, n, hipMemcpyDeviceToHost, st);
    }
    hipError_t small_sync(hipStream_t st) {
        const hipError_t e = hipStreamSynchronize(st);
        for (const PinDown &d : pin_pending) memcpy(d.dst, d.slot, d.n);
        pin_pending.clear();
        return e;
    }

    // state between encode_prepare and encode_emit
    uint32_t cur_nchunks = 0, cur_nblocks = 0;
    uint64_t cur_ntiles = 0, cur_n = 0;
    const uint32_t *cur_tile_map = nullptr;   // tile → chunk table of the prepared encode (lives in d_chunkmap)
    const uint8_t *cur_in = nullptr;
    bool force_match_v1 = false;   // sticky: the second-generation match kernel reported a lane-order violation
    uint64_t match_fallbacks = 0;  // encode passes this context ran on the fallback kernel because of it (lfx_ctx_match_fallbacks)
    HostIo hostio;                 // copy streams and page-locked slabs for callers' pageable buffers (lfx_hostio.h), made on first use
    // Buffers of stream encoders / decoders that were freed, kept for the next handle of this context: page-locking 16 .. 64 MiB
    // and a hipMalloc per handle cost more than encoding a small stream (an encoder per file is the reference's own usage,
    // examples/flate.rs:89-110).  Taken and given back under `mu`; at most POOL_MAX of each kind are kept.
    static constexpr size_t POOL_MAX = 8;
    std::vector<PinVec> pin_pool;
    std::vector<DevBuf> dev_pool;
    PinVec take_pin() {            // the roomiest one (a handle's first buffer is its largest: pending input, window output)
        std::lock_guard<std::recursive_mutex> lock(mu);
        if (pin_pool.empty()) return PinVec();
        size_t best = 0;
        for (size_t i = 1; i < pin_pool.size(); i++)
            if (pin_pool[i].capacity() > pin_pool[best].capacity()) best = i;
        PinVec v = std::move(pin_pool[best]);
        pin_pool.erase(pin_pool.begin() + (std::ptrdiff_t)best);
        v.clear();
        return v;
    }
    void give_pin(PinVec &&v) {
        std::lock_guard<std::recursive_mutex> lock(mu);
        if (v.capacity() && pin_pool.size() < POOL_MAX) { v.clear(); pin_pool.push_back(std::move(v)); }
    }
    DevBuf take_dev() {
        std::lock_guard<std::recursive_mutex> lock(mu);
        if (dev_pool.empty()) return DevBuf();
        DevBuf b = dev_pool.back();
        dev_pool.pop_back();
        return b;
    }
    void give_dev(DevBuf &b) {
        std::lock_guard<std::recursive_mutex> lock(mu);
        if (b.p && dev_pool.size() < POOL_MAX) dev_pool.push_back(b);
        else b.release();
        b = DevBuf();
    }
    std::vector<uint8_t> shard_hdr;
    int shard_format = 0;
    bool shard_last = false;

    // phase timers
    int timing_on = 0;      // 0: no events; 1: an event behind every phase; 2..5: only the two events around ONE kernel's phase (the bench's
                            // timed steps: an event record between two kernels is ~6 us of idle GPU, two dozen of them 3 % of a 256 MiB
                            // round trip) — 2: lz77_walk, 3: blk_scan, 4: lz77_cand, 5: lz77_copy; 6: every phase, the encode's match and
                            // parse phases split by kernel (lz77_cand + lz77_resolve for lz77_match, lz77_walk + lz77_chain for lz77_parse)
    bool timing_fine() const { return timing_on >= 2; }
    hipEvent_t ev[17] = {};
    char ev_name[17][24] = {};
    int n_ev = 0;
    void phase(const char *name);
};

// code words made on the host by a caller-supplied Lz77Encode (lfx_encoder_write_codes): contiguous, chunk after chunk
struct HostCodes {
    const uint32_t *codes;        // n_codes words, (val << 16) | dist, every block's EndOfBlock included
    uint64_t n_codes;
    const uint32_t *chunk_codes;  // codes per chunk of the plan
};
int encode_prepare(Ctx *c, const struct Plan &plan, const struct PlanOpts &po, const uint8_t *d_in,
                   uint64_t n, int ck_mode, const HostCodes *hc = nullptr);

}  // namespace lfx
