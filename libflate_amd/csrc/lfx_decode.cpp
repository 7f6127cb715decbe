// lfx_decode.cpp — host orchestration of the inflate path (C ABI: lfx_decode_*, lfx_decoder_*).
// Every compressed bit is decoded on the GPU; the host only sequences kernels, chains block
// boundaries and formats error messages.
#include "../../include/lfx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "lfx_ctx.h"
#include "lfx_container.h"
#include "lfx_decode.h"
#include <thread>
#include <chrono>

static_assert(offsetof(lfx::DecStream, out_off) == 16 && sizeof(lfx::DecStream) % 8 == 0, "checksum_ranges stride");
static_assert(offsetof(lfx::InflateResult, out_len) == 8 && sizeof(lfx::InflateResult) % 8 == 0, "checksum_ranges stride");
#include "lfx_device.h"
#include "lfx_abi_guard.h"

using namespace lfx;

#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            c->set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
            return LFX_E_DEVICE;                                                      \
        }                                                                             \
    } while (0)
#define LAUNCH_TRY(call)                                                              \
    do {                                                                              \
        int e_ = (call);                                                              \
        if (e_) {                                                                     \
            c->set_error(std::string(#call) + ": " + hipGetErrorString((hipError_t)e_)); \
            return LFX_E_DEVICE;                                                      \
        }                                                                             \
    } while (0)

namespace {

// messages: prefixes match the reference texts quoted in SURVEY.md §4
std::string format_error(uint32_t err, uint32_t a0, uint32_t a1) {
    char m[200];
    switch (err) {
        case ERR_EOF: return "failed to fill whole buffer";
        case ERR_HUFF: return "Invalid huffman coded stream";
        case ERR_CONFLICT: snprintf(m, sizeof m, "Bit region conflict: symbol=%u", a0); return m;
        case ERR_HDIST: snprintf(m, sizeof m, "The value of HDIST is too big: max=30, actual=%u", a0); return m;
        case ERR_NO_PREV: return "No preceding value";
        case ERR_DIST_LIST:
            snprintf(m, sizeof m, "The length of `distance_code_bitwidthes` is too large: actual=%u, expected=%u", a0, a1);
            return m;
        case ERR_286: snprintf(m, sizeof m, "The value %u must not occur in compressed data", a0); return m;
        case ERR_BACKREF: snprintf(m, sizeof m, "Too long backword reference: buffer.len=%u, distance=%u", a0, a1); return m;
        case ERR_BTYPE3: return "btype 0x11 of DEFLATE is reserved(error) value";
        case ERR_LEN_NLEN: snprintf(m, sizeof m, "LEN=%u is not the one's complement of NLEN=%u", a0, a1); return m;
        case ERR_STORED_SHORT: snprintf(m, sizeof m, "The reader has incorrect length: expected %u, read %u", a0, a1); return m;
        case ERR_NOSPACE: return "output capacity too small";
        case ERR_ZLIB_CHECK:
            snprintf(m, sizeof m, "Inconsistent ZLIB check bits: `CMF(%u) * 256 + FLG(%u)` must be a multiple of 31", a0, a1);
            return m;
        case ERR_METHOD: snprintf(m, sizeof m, "Compression methods other than DEFLATE(8) are unsupported: method=%u", a0); return m;
        case ERR_CINFO: snprintf(m, sizeof m, "CINFO above 7 are not allowed: value=%u", a0); return m;
        case ERR_FDICT: snprintf(m, sizeof m, "Preset dictionaries are not supported: dictionary_id=0x%X", a0); return m;
        case ERR_GZIP_ID: return "Unexpected GZIP ID";
        case ERR_HCRC: snprintf(m, sizeof m, "CRC16 of GZIP header mismatched: value=%u, expected=%u", a0, a1); return m;
        case ERR_CRC32: snprintf(m, sizeof m, "CRC32 mismatched: value=%u, expected=%u", a0, a1); return m;
        case ERR_ADLER32: snprintf(m, sizeof m, "Adler32 checksum mismatched: value=%u, expected=%u", a0, a1); return m;
        default: return "";
    }
}
int map_status(uint32_t st) {
    return st == 0 ? LFX_OK : st == 1 ? LFX_E_INVALID_DATA : st == 2 ? LFX_E_UNEXPECTED_EOF : LFX_E_NOSPACE;
}

struct MemberResult {
    int status = LFX_OK;
    uint64_t out_len = 0;        // bytes produced (also on failure)
    uint64_t blk_out_start = 0;  // bytes of completed blocks
    uint64_t end_byte = 0;       // input byte after the last DEFLATE byte (relative to member base)
    std::string msg;
    // windowed (partial) decode: where the decoded part ends and whether the member's last block is behind it
    uint64_t end_bit = 0;
    bool final_seen = false;
    bool need_cap = false;       // nothing decoded because the first block does not fit the output capacity
    // in: the container checksum the caller will need (launch_checksum mode: 1 CRC-32, 2 Adler-32, 0 none) and how many
    // trailer bytes follow the member; out (ck_done): checksum of the output and the trailer bytes, fetched in the SAME
    // host round trip as the materialisation's verdict (the checksum kernels are queued behind it before that verdict is
    // known: on the clean path one synchronisation less; a failed path simply ignores them)
    int ck_mode = 0;
    uint32_t trailer_len = 0;
    bool ck_done = false;
    uint32_t crc32 = 0, adler32 = 1;
    uint8_t trailer[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

// run `njobs` inflate jobs and fetch their results
int run_jobs(Ctx *c, const uint8_t *d_in, uint8_t *d_out, const std::vector<InflateJob> &jobs,
             std::vector<InflateResult> &res) {
    const size_t n = jobs.size();
    res.resize(n);
    if (!n) return LFX_OK;
    int rc;
    if ((rc = c->d_dec_streams.reserve(sizeof(InflateJob) * n))) return rc;
    if ((rc = c->d_dec_state.reserve(sizeof(InflateResult) * n))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_dec_streams.p, jobs.data(), sizeof(InflateJob) * n, hipMemcpyHostToDevice, c->stream));
    LAUNCH_TRY(launch_inflate(c->stream, d_in, d_out, (const InflateJob *)c->d_dec_streams.p,
                              (InflateResult *)c->d_dec_state.p, (uint32_t)n));
    HIP_TRY(hipMemcpyAsync(res.data(), c->d_dec_state.p, sizeof(InflateResult) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LFX_OK;
}

// Decode the DEFLATE stream that starts at byte `off0` of d_in[0..n) into d_out[0..cap).
// hist0 = 0 (a member starts with an empty Lz77Decoder buffer, gzip.rs:1000-1005).
// stop_bit != ~0: the walk ends cleanly when a block ends exactly at stop_bit (a shard of a member that
// does not hold the BFINAL block); start_bit0 may be any bit of the first byte.
// partial: a WINDOW of a member (the stream decoders): decode the blocks that are complete in d_in[0..n) and fit into
// `cap`, stop cleanly in front of the first one that is not (mr.end_bit = its header bit, mr.final_seen = false); `hist`
// = bytes of the member produced by earlier windows — the last 32 KiB of them lie right in front of d_out.
int inflate_member(Ctx *c, const uint8_t *d_in, uint64_t n, uint64_t off0, uint8_t *d_out, uint64_t cap,
                   MemberResult &mr, uint64_t start_bit0 = ~0ull, uint64_t stop_bit = ~0ull, bool partial = false,
                   uint64_t hist = 0) {
    const uint64_t first_bit = start_bit0 == ~0ull ? off0 * 8 : start_bit0;
    mr.end_bit = first_bit;
    hipStream_t st = c->stream;
    c->pin_reset();        // (the page-locked slots of the small transfers, lfx_ctx.h: nothing of an earlier member is in flight)
    std::vector<InflateJob> jobs;
    std::vector<InflateResult> res;
    bool parallel_done = false;
    const uint64_t comp = n > off0 ? n - off0 : 0;
    // (below a few KiB the exact serial kernel is faster than the parallel path's fixed cost of about 2 ms; measured:
    //  a 32 KiB stream takes 17 ms on the serial kernel)
    if (comp >= (4u << 10)) {
        // ---- speculative block-start search
        // survivors of stage 1 are ~0.1 % of the bit offsets (more on incompressible data): room for 0.4 % of them, so
        // that a gibibyte-sized stream does not overflow the lists and fall back to the serial walk
        int rc;
        bool overflow = false;
        uint32_t n1 = 0;
        std::vector<uint64_t> starts;
        auto find_candidates = [&]() -> int {
            const uint32_t shard_cap = find_shard_cap(comp);
            const uint32_t final_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 16, comp / 4096), 1u << 24);
            int rc;
            if ((rc = c->d_dec_cand.reserve(8ull * shard_cap * FIND_SHARDS + 8ull * final_cap + 4 * FIND_HDR_WORDS))) return rc;
            uint32_t *d_count = (uint32_t *)c->d_dec_cand.p;                       // the header (lfx_decode.h), the results, the lists
            uint32_t *d_final_count = d_count + FIND_HDR_FINAL;
            uint64_t *d_final = (uint64_t *)((uint8_t *)c->d_dec_cand.p + 4 * FIND_HDR_WORDS);   // (behind the header: both come back in ONE transfer)
            uint64_t *d_cand = d_final + final_cap;
            HIP_TRY(hipMemsetAsync(d_count, 0, 4 * FIND_HDR_WORDS, st));
            // (the member's last block is looked for in the final eighth of the input, at least 8 MiB of it: one that starts
            //  earlier — a last block of more than that — is scanned on demand by the chain walk below)
            const uint64_t tail_bytes = std::max<uint64_t>(comp / 8, 8ull << 20);
            const uint64_t final_from = c->diag.no_final_cand ? ~0ull >> 1 : comp > tail_bytes ? (n - tail_bytes) * 8 : 0;
            LAUNCH_TRY(launch_find_stage1(st, d_in, n, off0, d_count, d_cand, shard_cap, final_from, (uint32_t)std::max(c->n_cu, 1)));
            c->phase("find1");
            // stage 2 takes the survivor counts from the device (persistent grid): no host round trip between the stages; the
            // counts, the overflow marker, the number of results and the first results come back in ONE round trip
            const int find2_exp = c->diag.find2_exp >= 1 && c->diag.find2_exp <= 4 ? c->diag.find2_exp : 0;
            if (find2_exp) {       // timing experiment: a cut-down stage 2 first (phase "find2x"), then the real one
                LAUNCH_TRY(launch_find_stage2(st, d_in, n, d_cand, shard_cap, d_count, d_count + FIND_HDR_WORK, d_final_count, d_final, final_cap,
                                              (uint32_t)std::max(c->n_cu, 1), nullptr, find2_exp));
                HIP_TRY(hipMemsetAsync(d_count + FIND_HDR_FINAL, 0, 4, st));
                HIP_TRY(hipMemsetAsync(d_count + FIND_HDR_WORK, 0, 4 * (FIND_HDR_WORDS - FIND_HDR_WORK), st));
                c->phase("find2x");
            }
            LAUNCH_TRY(launch_find_stage2(st, d_in, n, d_cand, shard_cap, d_count, d_count + FIND_HDR_WORK, d_final_count, d_final, final_cap,
                                          (uint32_t)std::max(c->n_cu, 1), c->diag.debug ? (uint64_t *)(d_count + FIND_HDR_DBG) : nullptr));
            constexpr uint32_t HEAD_N = 1024;      // (results that come back with the counts; a stream has a few hundred)
            const uint32_t head_n = std::min<uint32_t>(HEAD_N, final_cap);
            std::vector<uint64_t> back(FIND_HDR_WORDS / 2 + head_n);      // header words, then the first results
            HIP_TRY(c->small_down(back.data(), d_count, 8ull * back.size(), st));
            HIP_TRY(c->small_sync(st));
            const uint32_t *hc = (const uint32_t *)back.data();
            std::vector<uint64_t> cand(back.begin() + FIND_HDR_WORDS / 2, back.end());
            overflow = hc[FIND_SHARDS] != 0;
            n1 = 0;
            for (uint32_t k = 0; k < FIND_SHARDS; k++) { if (hc[k] > shard_cap) overflow = true; n1 += hc[k]; }
            if (c->diag.debug) {
                uint64_t d[7];
                memcpy(d, hc + FIND_HDR_DBG, sizeof d);
                fprintf(stderr, "[lfx] finder stage 2: batches=%llu cycles per batch: stage+fields=%llu table=%llu walk=%llu; steps per batch=%.1f "
                        "restagings=%llu; wavefront lives (sum)=%llu\n", (unsigned long long)d[5], (unsigned long long)(d[0] / (d[5] ? d[5] : 1)),
                        (unsigned long long)(d[1] / (d[5] ? d[5] : 1)), (unsigned long long)(d[2] / (d[5] ? d[5] : 1)),
                        (double)d[3] / (double)(d[5] ? d[5] : 1), (unsigned long long)d[4], (unsigned long long)d[6]);
            }
            starts.clear();
            starts.push_back(first_bit);  // the first block's start is known
            if (overflow) return LFX_OK;
            uint32_t nf = hc[FIND_HDR_FINAL];
            if (nf > final_cap) nf = final_cap;
            cand.resize(nf);
            if (nf > head_n) HIP_TRY(hipMemcpy(cand.data() + head_n, d_final + head_n, 8ull * (nf - head_n), hipMemcpyDeviceToHost));
            c->phase("find2");
            for (uint32_t i = 0; i < nf; i++) if (cand[i] != first_bit) starts.push_back(cand[i]);
            std::sort(starts.begin(), starts.end());
            return LFX_OK;
        };
        // A small stream (round 4) does not start with the finder (0.23 ms of fixed cost): reference-made streams of this
        // size are one block plus the empty final one, which the piece scan below walks from the known first block in two
        // steps.  A stream that turns out to have many blocks (another encoder's) gets the finder after all.
        // (measured, profiles/r04_small_sizes.json: every dependent block step costs about 0.2 ms — header parse and table
        //  build of one workgroup — so the walk pays for up to three blocks: streams below 1.5 MiB)
        const bool small_first = comp < (1536u << 10) && stop_bit == ~0ull && !partial && !c->diag.no_pieces;
        if (small_first) starts.push_back(first_bit);
        else if ((rc = find_candidates())) return rc;
        if (!overflow) {
            // ---- K1: every candidate block is scanned by a 1024-lane workgroup (speculative slices,
            //      chained exits) for its end bit, byte and code counts
            uint32_t nc = (uint32_t)starts.size();
            auto start_at = [&](uint32_t i) { return i < nc ? starts[i] : n * 8; };
            std::vector<BlkEmit> emit;
            uint32_t n_placed = 0;          // blocks whose codes the scan stored (blk_place_kernel instead of blk_emit_kernel)
            bool scan_small = false;        // the 256-lane instances of scan and emit (small blocks)
            uint64_t pos = first_bit, total = 0, total_codes = 0;
            bool ok_chain = false, chain_final = false;
            bool front_bad = false;      // the window's FIRST block does not scan: damaged rather than incomplete
            uint64_t last_end = 0;   // end bit of the last block of the chain
            bool pieces_mode = false;
            uint32_t n_on_demand = 0;       // blocks of the chain the finder did not report (scanned one by one)
            bool pieces_multi = false;   // some block was scanned in more than one piece (its pieces read each other's output)
            bool units_per_piece = false; // (pieces over candidate ranges that fill the GPU: one symbol unit per piece)
            const size_t tab_bytes = blk_tabs_bytes();
            // ---- few candidates in a long stream = few, huge blocks (schedule S1: ONE block for the whole input).
            // One workgroup per block would scan it alone; instead the block is scanned in PIECES of 4 Mbit, one
            // workgroup each, all with the block's tables.  A piece finds its first symbol boundary by a warm-up
            // decode that starts a few Kbit early; it is accepted iff that boundary equals the exit of the piece in
            // front of it (piece 0 starts exactly behind the header), so the chain of pieces is proven, not assumed.
            // Pieces behave like blocks from here on (their back-references cross pieces: marker path).
            // Piece size (round 4): a 1 MiB stream is ONE block too — one workgroup in K1 and K2, four units in K3 (2.1 ms).
            // Pieces adapt to the stream: enough of them to give every CU two, between 256 Kbit and 4 Mbit each.
            // (few candidates in a long stream = few, huge blocks — at least 2 MiB of stream per candidate; a 4 MiB stream of four
            //  ordinary blocks is not that case: walking it block by block costs a dependent step per block)
            // ---- a stream of a FEW ordinary blocks (round 4: 2 … 32 candidates, at least 128 KiB of stream each — a 16 MiB
            // member is 16 blocks for 256 CUs: K1 and K2 ran on sixteen of them, K3 on sixty-four chunk units, 2.3 ms): every
            // candidate's range [start, next candidate) is scanned in pieces at once, with the tables of the block that
            // starts there.  Accepted only if every block's pieces chain, the piece that holds its EndOfBlock ends exactly
            // where the next candidate starts, and the last block is final; anything else (a false candidate, a stored or
            // fixed block in between, damage) leaves the stream to the one-workgroup-per-block path below.
            // (Measured, profiles/r04_small_sizes.json: 16 MiB 2.25 -> 1.75 ms; at 64 blocks the marker path's fixed costs — window
            //  resolution 0.6 ms, symbol units — outweigh what K1 / K2 gain: 2.87 against 2.40 ms.  Hence up to 32 candidates.)
            const bool giant_blocks = nc <= 8 && comp / nc >= (2u << 20);     // (schedule S1: the block-by-block piece walk below)
            if (!small_first && !giant_blocks && nc >= 2 && nc <= (uint32_t)c->diag.pocr_max && comp / nc >= (128u << 10) && stop_bit == ~0ull &&
                !partial && !c->diag.no_pieces) {
                const uint64_t end_bits = n * 8;
                // two pieces per CU over the whole stream, every block split EVENLY (a block of 4.7 Mbit in pieces of 4 Mbit is
                // a long piece and a short one: the symbol kernel's time is that of its largest unit) — the blocks' rounding
                // taken off the target, so that the pieces, one symbol unit each, are all resident at once
                const uint64_t ptarget = (uint64_t)std::max<int64_t>(2ll * std::max(c->n_cu, 1) - (int64_t)nc, (int64_t)nc);
                const uint64_t PIECE_BITS = std::min<uint64_t>(4ull << 20, std::max<uint64_t>(256ull << 10,
                                            ((end_bits - first_bit) / ptarget + 63) & ~63ull));
                // warm-up in front of a piece: ONE lane decodes it, 0.2 us per symbol on an otherwise idle CU — 8 Kbit are 550
                // symbols, 115 us, most of a small stream's scan step (round 4, profiles/r04_small_sizes.json).  A speculative
                // decode is in step within a few dozen symbols; small pieces get 2 Kbit.  (A piece whose warm-up did not get in
                // step is rejected by the chain check below, and the stream takes the one-workgroup-per-block path.)
                const uint64_t OVERLAP = PIECE_BITS <= (1ull << 20) ? 2048 : 8192;
                std::vector<BlkJob> pj;
                std::vector<uint32_t> first_piece(nc + 1, 0);
                // the pieces of one range [s0, s1), scanned with the tables of the block whose header is at s0
                auto add_pieces = [&](std::vector<BlkJob> &jobs, uint64_t s0, uint64_t s1) {
                    const uint64_t len = s1 - s0;
                    const uint64_t np0 = std::max<uint64_t>((len + PIECE_BITS - 1) / PIECE_BITS, 1);
                    const uint64_t pb = std::max<uint64_t>(((len + np0 - 1) / np0 + 63) & ~63ull, 64);     // this block's piece
                    const uint32_t np = (uint32_t)std::max<uint64_t>((len + pb - 1) / pb, 1);
                    for (uint32_t q = 0; q < np; q++) {
                        const uint64_t lo = s0 + q * pb;
                        jobs.push_back(BlkJob{s0, std::min(lo + pb, s1), q ? lo : 0, q ? lo - OVERLAP : 0, 1u, 0u});
                    }
                };
                for (uint32_t i = 0; i < nc; i++) {
                    first_piece[i] = (uint32_t)pj.size();
                    add_pieces(pj, starts[i], start_at(i + 1));
                }
                first_piece[nc] = (uint32_t)pj.size();
                const uint32_t npj = (uint32_t)pj.size();
                units_per_piece = npj >= (uint32_t)std::max(c->n_cu, 1);
                // (+ slots for the repair launches below: a range cut in two by a false candidate is scanned again as one)
                constexpr uint32_t REPAIRS = 2;
                const uint32_t repair_slots = REPAIRS * (2 * (uint32_t)((end_bits - first_bit) / nc / PIECE_BITS + 2) + 4);
                const uint32_t nslots = npj + repair_slots;
                int rc2;
                if ((rc2 = c->d_dec_streams.reserve(sizeof(BlkJob) * nslots))) return rc2;
                if ((rc2 = c->d_dec_state.reserve(sizeof(BlkInfo) * nslots))) return rc2;
                if ((rc2 = c->d_dec_blocks.reserve(sizeof(BlkLanes) * (size_t)nslots))) return rc2;
                if ((rc2 = c->d_dec_tabs.reserve(tab_bytes * nslots))) return rc2;
                HIP_TRY(hipMemcpyAsync(c->d_dec_streams.p, pj.data(), sizeof(BlkJob) * npj, hipMemcpyHostToDevice, st));
                LAUNCH_TRY(launch_blk_scan(st, d_in, n, (const BlkJob *)c->d_dec_streams.p, npj, (BlkInfo *)c->d_dec_state.p,
                                           (BlkLanes *)c->d_dec_blocks.p, c->d_dec_tabs.p));
                std::vector<BlkInfo> pi(npj);
                HIP_TRY(hipMemcpyAsync(pi.data(), c->d_dec_state.p, sizeof(BlkInfo) * npj, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                bool fail = false;
                // the pieces of one range in order → emit entries.  0: the block closed (pos / ok_chain updated); 1: every piece is
                // open (no EndOfBlock in the range); 2: damaged, or the pieces do not chain
                auto walk_range = [&](const BlkInfo *infos, uint32_t cnt, uint32_t slot0) -> int {
                    uint64_t prev_end = 0;
                    for (uint32_t q = 0; q < cnt; q++) {
                        const BlkInfo &r = infos[q];
                        if (r.status == BLK_BAD || r.btype == 0 || (q && r.data_bit != prev_end) || r.end_bit <= pos || r.end_bit > end_bits) return 2;
                        BlkEmit e{};
                        e.start_bit = pos; e.data_bit = r.data_bit; e.code_off = total_codes; e.out_off = total;
                        e.n_out = r.n_out; e.n_codes = r.n_codes; e.nlanes = r.nlanes; e.btype = r.btype; e.cand = slot0 + q;
                        e.hist = hist + total;
                        e.end_limit = r.status == BLK_NO_EOB ? r.end_bit : 0;
                        emit.push_back(e);
                        total += r.n_out;
                        total_codes += r.n_codes;
                        prev_end = r.end_bit;
                        if (r.status == BLK_OK) {
                            last_end = r.end_bit;
                            pieces_multi |= q > 0;
                            if (r.bfinal) { ok_chain = true; chain_final = true; } else pos = r.end_bit;
                            return 0;
                        }
                    }
                    return 1;
                };
                uint32_t repairs = 0, rslot = npj;
                for (uint32_t i = 0; i < nc && !fail && !ok_chain; i++) {
                    if (starts[i] != pos) { fail = true; break; }             // (the chain must pass through every candidate)
                    const size_t emit_mark = emit.size();
                    const uint64_t total_mark = total, codes_mark = total_codes;
                    int res = walk_range(&pi[first_piece[i]], first_piece[i + 1] - first_piece[i], first_piece[i]);
                    if (res == 1 && i + 1 < nc && repairs < REPAIRS) {
                        // No EndOfBlock up to the next candidate: that candidate is a false one inside this block (about one
                        // per 30 MB of stream survives the finder), and the pieces behind it were scanned with tables read from
                        // data.  The block's true range — up to the candidate after it — is scanned again, alone.
                        emit.resize(emit_mark); total = total_mark; total_codes = codes_mark;
                        std::vector<BlkJob> rj;
                        add_pieces(rj, starts[i], start_at(i + 2));
                        const uint32_t nr = (uint32_t)rj.size();
                        if (rslot + nr > nslots) { fail = true; break; }
                        HIP_TRY(hipMemcpyAsync((BlkJob *)c->d_dec_streams.p + rslot, rj.data(), sizeof(BlkJob) * nr, hipMemcpyHostToDevice, st));
                        LAUNCH_TRY(launch_blk_scan(st, d_in, n, (const BlkJob *)c->d_dec_streams.p + rslot, nr, (BlkInfo *)c->d_dec_state.p + rslot,
                                                   (BlkLanes *)c->d_dec_blocks.p + rslot, (uint8_t *)c->d_dec_tabs.p + tab_bytes * rslot));
                        std::vector<BlkInfo> ri(nr);
                        HIP_TRY(hipMemcpyAsync(ri.data(), (BlkInfo *)c->d_dec_state.p + rslot, sizeof(BlkInfo) * nr, hipMemcpyDeviceToHost, st));
                        HIP_TRY(hipStreamSynchronize(st));
                        res = walk_range(ri.data(), nr, rslot);
                        if (c->diag.debug) fprintf(stderr, "[lfx]  pieces: candidate %u (bit %llu) is a false one, its block scanned again in %u pieces: %d\n", i + 1,
                                                   (unsigned long long)starts[i + 1], nr, res);
                        rslot += nr;
                        repairs++;
                        i++;                      // (the false candidate is not a block)
                    }
                    if (res != 0) fail = true;
                }
                if (c->diag.debug) fprintf(stderr, "[lfx]  pieces over %u candidate ranges: ok=%d fail=%d pieces=%zu of %u total=%llu\n", nc, (int)ok_chain, (int)fail,
                                           emit.size(), npj, (unsigned long long)total);
                if (fail || !ok_chain) { emit.clear(); pos = first_bit; total = 0; total_codes = 0; ok_chain = false; chain_final = false; last_end = 0; pieces_multi = false; units_per_piece = false; }
                else pieces_mode = true;
                c->phase("pieces");
            }
            if (!pieces_mode && (small_first || (nc <= 8 && comp / nc >= (2u << 20))) && stop_bit == ~0ull && !partial && !c->diag.no_pieces) {
                const uint64_t end_bits = n * 8;
                const uint64_t PIECE_BITS = std::min<uint64_t>(4ull << 20, std::max<uint64_t>(256ull << 10,
                                            ((end_bits - first_bit) / (2ull * (uint64_t)std::max(c->n_cu, 1)) + 63) & ~63ull));
                // warm-up in front of a piece: ONE lane decodes it, 0.2 us per symbol on an otherwise idle CU — 8 Kbit are 550
                // symbols, 115 us, most of a small stream's scan step (round 4, profiles/r04_small_sizes.json).  A speculative
                // decode is in step within a few dozen symbols; small pieces get 2 Kbit.  (A piece whose warm-up did not get in
                // step is rejected by the chain check below, and the stream takes the one-workgroup-per-block path.)
                const uint64_t OVERLAP = PIECE_BITS <= (1ull << 20) ? 2048 : 8192;
                const uint32_t cap_slots = (uint32_t)std::min<uint64_t>((end_bits - first_bit) / PIECE_BITS * 2 + 64, 1u << 20);
                int rc2;
                if ((rc2 = c->d_dec_streams.reserve(sizeof(BlkJob) * cap_slots))) return rc2;
                if ((rc2 = c->d_dec_state.reserve(sizeof(BlkInfo) * cap_slots))) return rc2;
                if ((rc2 = c->d_dec_blocks.reserve(sizeof(BlkLanes) * (size_t)cap_slots))) return rc2;
                if ((rc2 = c->d_dec_tabs.reserve(tab_bytes * cap_slots))) return rc2;
                uint32_t base = 0;
                bool fail = false;
                for (uint32_t iter = 0; iter < (small_first ? 4u : 64u) && !fail && !ok_chain; iter++) {
                    const uint32_t np = (uint32_t)((end_bits - pos + PIECE_BITS - 1) / PIECE_BITS);
                    if (np == 0 || base + np > cap_slots) { fail = true; break; }
                    std::vector<BlkJob> pj(np);
                    for (uint32_t q = 0; q < np; q++) {
                        const uint64_t lo = pos + q * PIECE_BITS;
                        pj[q] = BlkJob{pos, std::min(lo + PIECE_BITS, end_bits), q ? lo : 0, q ? lo - OVERLAP : 0, 1u, 0u};
                    }
                    HIP_TRY(hipMemcpyAsync((BlkJob *)c->d_dec_streams.p + base, pj.data(), sizeof(BlkJob) * np, hipMemcpyHostToDevice, st));
                    LAUNCH_TRY(launch_blk_scan(st, d_in, n, (const BlkJob *)c->d_dec_streams.p + base, np, (BlkInfo *)c->d_dec_state.p + base,
                                               (BlkLanes *)c->d_dec_blocks.p + base, (uint8_t *)c->d_dec_tabs.p + tab_bytes * base));
                    std::vector<BlkInfo> pi(np);
                    HIP_TRY(hipMemcpyAsync(pi.data(), (BlkInfo *)c->d_dec_state.p + base, sizeof(BlkInfo) * np, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    uint32_t used = 0;
                    bool closed = false;
                    uint64_t prev_end = 0;
                    for (uint32_t q = 0; q < np && !closed; q++) {
                        const BlkInfo &r = pi[q];
                        if (c->diag.debug && (q < 3 || r.status != BLK_NO_EOB))
                            fprintf(stderr, "[lfx]   piece %u/%u: status=%u btype=%u data=%llu end=%llu prev_end=%llu lanes=%u codes=%u out=%llu rounds=%u cyc_hdr=%u cyc_total=%u\n", q, np,
                                    r.status, r.btype, (unsigned long long)r.data_bit, (unsigned long long)r.end_bit,
                                    (unsigned long long)prev_end, r.nlanes, r.n_codes, (unsigned long long)r.n_out, r.rounds, r.cyc_hdr, r.cyc_total);
                        if (r.status == BLK_BAD || (q && r.data_bit != prev_end) || r.end_bit <= pos || r.end_bit > end_bits) { fail = true; break; }
                        if (q == 0 && r.btype == 0 && r.status != BLK_OK) { fail = true; break; }
                        BlkEmit e{};
                        e.start_bit = pos; e.data_bit = r.data_bit; e.code_off = total_codes; e.out_off = total;
                        e.n_out = r.n_out; e.n_codes = r.n_codes; e.nlanes = r.nlanes; e.btype = r.btype; e.cand = base + q;
                        e.hist = hist + total;   // (a stream decoder's last window arrives with history)
                        e.end_limit = r.status == BLK_NO_EOB ? r.end_bit : 0;   // an open piece ends where its last lane stopped
                        emit.push_back(e);
                        total += r.n_out;
                        total_codes += r.n_codes;
                        used = q + 1;
                        prev_end = r.end_bit;
                        if (r.status == BLK_OK) {          // the piece that holds EndOfBlock (or a whole stored block)
                            closed = true;
                            last_end = r.end_bit;
                            pieces_multi |= q > 0;
                            if (r.bfinal) { ok_chain = true; chain_final = true; } else pos = r.end_bit;
                        }
                    }
                    if (!closed) fail = true;
                    base += used;
                }
                if (c->diag.debug) fprintf(stderr, "[lfx]  pieces: ok=%d fail=%d pieces=%zu total=%llu\n", (int)ok_chain, (int)fail, emit.size(), (unsigned long long)total);
                if (fail || !ok_chain) { emit.clear(); pos = first_bit; total = 0; total_codes = 0; ok_chain = false; last_end = 0; pieces_multi = false; }
                else pieces_mode = true;
                c->phase("pieces");
                if (small_first && !pieces_mode) {     // many blocks after all: the finder, then one workgroup per block
                    if ((rc = find_candidates())) return rc;
                    nc = (uint32_t)starts.size();
                }
            }
            if (!pieces_mode && !overflow) {
            std::vector<BlkJob> bj(nc);
            for (uint32_t i = 0; i < nc; i++) bj[i] = BlkJob{starts[i], start_at(i + 1)};
            // A false candidate inside a block cuts that block's range in two, and the first part then has
            // no end-of-block.  Ranges much shorter than the median are the suspects: the candidate in
            // front of each (and the suspect itself) also gets a job that ignores one candidate, in the
            // same launch; what is still unresolved afterwards goes through the repair rescans below.
            std::vector<int32_t> alt(nc, -1);
            {
                std::vector<uint64_t> len(nc);
                for (uint32_t i = 0; i < nc; i++) len[i] = bj[i].end_bit - bj[i].start_bit;
                std::vector<uint64_t> sorted = len;
                std::nth_element(sorted.begin(), sorted.begin() + nc / 2, sorted.end());
                const uint64_t median = sorted[nc / 2], thresh = median / 5 * 3;
                const uint32_t max_extra = nc / 4 + 4;
                for (uint32_t i = 0; i + 1 < nc && bj.size() - nc < max_extra; i++) {
                    if (len[i] >= thresh && len[i + 1] >= thresh) continue;
                    // (the LAST range is short because the stream ends there — a member's final block is often tiny or empty —
                    //  not because a false candidate cut it: a full-size alternative job for it is a second workgroup on one
                    //  CU, and that CU decides the kernel's duration: 0.74 against 0.62 ms at 256 blocks on 256 CUs)
                    if (i + 2 == nc && len[i] >= thresh) continue;
                    // a block cut in two is about one block long when put together; anything much longer
                    // would only be a slow job that decides the kernel's duration
                    if (len[i] + len[i + 1] > median + median / 4) continue;
                    alt[i] = (int32_t)bj.size();
                    bj.push_back(BlkJob{starts[i], start_at(i + 2)});
                }
            }
            // few candidates = few, huge blocks (schedule S1: one): a false candidate would cost a full rescan of
            // such a block, so the known first block also gets a job that runs to the end of the stream
            if (nc > 1 && nc <= 8 && alt[0] < 0 && comp / nc >= (2u << 20)) {
                alt[0] = (int32_t)bj.size();
                bj.push_back(BlkJob{starts[0], n * 8});
            }
            const uint32_t nj = (uint32_t)bj.size();
            // (+ EXTRA slots for blocks the finder cannot see — fixed-Huffman and stored blocks of other encoders —
            // which the chain walk below scans on demand)
            constexpr uint32_t EXTRA = 64, MAX_ON_DEMAND = 1u << 16;
            if ((rc = c->d_dec_streams.reserve(sizeof(BlkJob) * (nj + 1)))) return rc;
            if ((rc = c->d_dec_state.reserve(sizeof(BlkInfo) * (nj + EXTRA + 1)))) return rc;
            if ((rc = c->d_dec_blocks.reserve(sizeof(BlkLanes) * (size_t)(nj + EXTRA + 1)))) return rc;
            if ((rc = c->d_dec_tabs.reserve(tab_bytes * (nj + EXTRA + 1)))) return rc;
            // ---- ONE Huffman pass for a stream's own large blocks (round 6): the scan stores every lane's code words in a
            //      region of its own (cap = half a code per bit of the slice + a head's worth + slack: a slice whose codes
            //      average less than two bits overflows, is flagged, and takes the emit kernel as before), blk_place_kernel
            //      moves them.  Only the first batch of jobs stores; rescans and on-demand scans are the classic ones.
            //      Not for thousands of small blocks (the regions' fixed part would dominate) or when the regions would
            //      not fit 16 GiB; LFX_TWO_PASS=1 keeps the emit kernel for everything.
            std::vector<uint8_t> stored(nj + EXTRA + 1, 0);
            bool store_mode = !c->diag.two_pass && nj && (n * 8) / nj >= (1ull << 20);
            // another encoder's blocks of a few tens of KB: the 256-lane instances of the scan and the emit kernel — for EVERY scan
            // of this call (rescans and on-demand scans too: the emit launch takes all blocks in one geometry)
            scan_small = !c->diag.no_small_scan && nj && (n * 8) / nj < (512ull << 10);
            if (store_mode) {
                uint64_t off = 0;
                for (uint32_t j = 0; j < nj; j++) {
                    const uint64_t bits = bj[j].end_bit > bj[j].start_bit ? bj[j].end_bit - bj[j].start_bit : 0;
                    const uint64_t slice = std::max<uint64_t>((bits + 1023) / 1024, 128);
                    const uint64_t cap = (slice / (c->diag.store_tight ? 16 : 2) + 448 + 64 + 3) & ~3ull;   // (448 = SCAN_HEADCAP, lfx_inflate_fast.hip)
                    bj[j].temp_off = off;
                    bj[j].cap = (uint32_t)cap;
                    off += 1024 * cap;
                }
                if (off * 4 > (16ull << 30) || c->d_dec_temp.reserve(off * 4) || c->d_dec_lanesx.reserve(sizeof(BlkLanesX) * (size_t)(nj + 1))) {
                    store_mode = false;
                    for (uint32_t j = 0; j < nj; j++) { bj[j].temp_off = 0; bj[j].cap = 0; }
                }
            }
            HIP_TRY(c->small_up(c->d_dec_streams.p, bj.data(), sizeof(BlkJob) * nj, st));
            if (store_mode)
                LAUNCH_TRY(launch_blk_scan_store(st, d_in, n, (const BlkJob *)c->d_dec_streams.p, nj, (BlkInfo *)c->d_dec_state.p,
                                                 (BlkLanes *)c->d_dec_blocks.p, c->d_dec_tabs.p, (uint32_t *)c->d_dec_temp.p,
                                                 (BlkLanesX *)c->d_dec_lanesx.p));
            else
                LAUNCH_TRY(launch_blk_scan(st, d_in, n, (const BlkJob *)c->d_dec_streams.p, nj, (BlkInfo *)c->d_dec_state.p,
                                           (BlkLanes *)c->d_dec_blocks.p, c->d_dec_tabs.p, scan_small));
            std::vector<BlkInfo> bi(nj);
            HIP_TRY(c->small_down(bi.data(), c->d_dec_state.p, sizeof(BlkInfo) * nj, st));
            HIP_TRY(c->small_sync(st));
            for (BlkInfo &b : bi) if (b.status == BLK_OK && b.end_bit > n * 8) b.status = BLK_NO_EOB;   // (cut by the input's end)
            if (store_mode) for (uint32_t j = 0; j < nj; j++) stored[j] = bi[j].status == BLK_OK && bi[j].btype != 0 && bi[j]._pad == 0;
            c->phase("blk_scan");
            // slot[i]: where candidate i's scan result and lanes live (its own slot or the wider job's)
            std::vector<uint32_t> slot(nc);
            for (uint32_t i = 0; i < nc; i++)
                slot[i] = (bi[i].status == BLK_NO_EOB && alt[i] >= 0 && bi[alt[i]].status == BLK_OK) ? (uint32_t)alt[i] : i;
            if (c->diag.debug) {
                fprintf(stderr, "[lfx] finder: stage1=%u candidates=%u scan jobs=%u (alternatives: %u)\n", n1, nc, nj, nj - nc);
                for (uint32_t i = 0; i < nc && i < 12; i++)
                    fprintf(stderr, "[lfx]  cand %u start=%llu status=%u btype=%u final=%u end=%llu n_out=%llu n_codes=%u lanes=%u rounds=%u cyc_hdr=%u cyc_total=%u\n",
                            i, (unsigned long long)starts[i], bi[i].status, bi[i].btype, bi[i].bfinal,
                            (unsigned long long)bi[i].end_bit, (unsigned long long)bi[i].n_out, bi[i].n_codes,
                            bi[i].nlanes, bi[i].rounds, bi[i].cyc_hdr, bi[i].cyc_total);
                uint32_t nbad = 0, maxr = 0;
                for (uint32_t i = 0; i < nc; i++) { nbad += bi[i].status != BLK_OK; maxr = std::max(maxr, bi[i].rounds); }
                fprintf(stderr, "[lfx]  not-ok=%u max_rounds=%u\n", nbad, maxr);
            }
            // still without an end-of-block: rescan those together with wider and wider ranges (one round
            // trip per widening step; a rescanned block's lanes land in its own slot)
            for (uint32_t widen = 2; widen <= 6; widen++) {
                std::vector<uint32_t> redo;
                for (uint32_t i = 0; i < nc; i++) if (bi[slot[i]].status == BLK_NO_EOB && i + widen <= nc) redo.push_back(i);
                if (redo.empty()) break;
                std::vector<BlkJob> rj(redo.size());
                for (size_t q = 0; q < redo.size(); q++) rj[q] = BlkJob{starts[redo[q]], start_at(redo[q] + widen)};
                if ((rc = c->d_dec_tmp.reserve(sizeof(BlkJob) * redo.size()))) return rc;
                BlkJob *d_rj = (BlkJob *)c->d_dec_tmp.p;
                HIP_TRY(hipMemcpyAsync(d_rj, rj.data(), sizeof(BlkJob) * redo.size(), hipMemcpyHostToDevice, st));
                for (size_t q = 0; q < redo.size(); q++) {
                    slot[redo[q]] = redo[q];
                    stored[redo[q]] = 0;            // (a classic scan takes the slot over: what its lanes stored before is stale)
                    LAUNCH_TRY(launch_blk_scan(st, d_in, n, d_rj + q, 1, (BlkInfo *)c->d_dec_state.p + redo[q],
                                               (BlkLanes *)c->d_dec_blocks.p + redo[q],
                                               (uint8_t *)c->d_dec_tabs.p + tab_bytes * redo[q], scan_small));
                }
                for (size_t q = 0; q < redo.size(); q++)
                    HIP_TRY(hipMemcpyAsync(&bi[redo[q]], (BlkInfo *)c->d_dec_state.p + redo[q], sizeof(BlkInfo), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
            }
            c->phase("repair");
            // ---- chain from the known first block
            uint32_t n_extra = 0;
            n_on_demand = 0;
            for (;;) {
                if (pos == stop_bit && !emit.empty()) { ok_chain = true; break; }
                auto it = std::lower_bound(starts.begin(), starts.end(), pos);
                uint32_t k;
                BlkInfo r;
                if (it != starts.end() && *it == pos) { k = slot[(uint32_t)(it - starts.begin())]; r = bi[k]; }
                else {
                    // not a dynamic-block start the finder knows: scan the block that starts here on demand
                    // (stored blocks need no slot; fixed / late dynamic ones keep one of the EXTRA slots)
                    if (pos >= n * 8 || n_on_demand++ >= MAX_ON_DEMAND || n_extra >= EXTRA) break;
                    k = nj + n_extra;
                    const BlkJob one{pos, it != starts.end() ? *it : n * 8};
                    BlkJob *d_one = (BlkJob *)c->d_dec_streams.p + nj;
                    HIP_TRY(hipMemcpyAsync(d_one, &one, sizeof one, hipMemcpyHostToDevice, st));
                    LAUNCH_TRY(launch_blk_scan(st, d_in, n, d_one, 1, (BlkInfo *)c->d_dec_state.p + k,
                                               (BlkLanes *)c->d_dec_blocks.p + k, (uint8_t *)c->d_dec_tabs.p + tab_bytes * k, scan_small));
                    HIP_TRY(hipMemcpyAsync(&r, (BlkInfo *)c->d_dec_state.p + k, sizeof r, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    if (r.status == BLK_OK && r.btype != 0) n_extra++;   // the slot stays in use
                }
                //  a block the end of the input cuts in half is INCOMPLETE even when its scan reports an EndOfBlock:
                // the last lane decodes a few symbols past the input (the bit source repeats the last dword there) and one
                // of them may read as EndOfBlock — end_bit then lies behind the input and n_out counts garbage symbols
                if (r.status == BLK_OK && r.end_bit > n * 8) r.status = BLK_NO_EOB;
                if (r.status != BLK_OK || r.end_bit <= pos) { front_bad = emit.empty() && r.status == BLK_BAD; break; }
                if (partial && total + r.n_out > cap) { mr.need_cap = emit.empty(); break; }   // (the next window takes it)
                BlkEmit e{};
                e.start_bit = pos; e.data_bit = r.data_bit; e.code_off = total_codes; e.out_off = total;
                e.n_out = r.n_out; e.n_codes = r.n_codes; e.nlanes = r.nlanes; e.btype = r.btype; e.cand = k;
                e.hist = hist + total;      // (hist = 0: a member starts with an empty window)
                if (k < nj && stored[k]) { e.placed = 1; e.temp_off = bj[k].temp_off; e.cap = bj[k].cap; n_placed++; }
                emit.push_back(e);
                total += r.n_out;
                total_codes += r.n_codes;
                last_end = r.end_bit;
                if (r.bfinal) { ok_chain = true; chain_final = true; break; }
                pos = r.end_bit;
            }
            // a window: the blocks in front of the first incomplete one are what this call delivers
            if (partial && !ok_chain && !emit.empty()) ok_chain = true;
            }   // !pieces_mode
            if (c->diag.debug) fprintf(stderr, "[lfx]  chain ok=%d blocks=%zu pos=%llu total=%llu on_demand=%u\n", (int)ok_chain, emit.size(), (unsigned long long)pos, (unsigned long long)total, n_on_demand);
            if (partial && emit.empty() && !front_bad) {
                // a window without one complete block (or whose first block does not fit `cap`): nothing to deliver — the
                // caller widens the window (the exact serial walk of the whole window would only find out the same, slowly).
                // A first block that does not even scan is different: the serial walk below says at once whether the
                // stream is damaged there (a verdict) or merely cut (nothing to deliver).
                mr.status = LFX_OK; mr.out_len = 0; mr.blk_out_start = 0; mr.final_seen = false;
                mr.end_bit = first_bit; mr.end_byte = first_bit / 8;
                return LFX_OK;
            }
            if (ok_chain && total <= cap) {
                // ---- K2 + K3: validated lanes emit codes, one wavefront per block materialises them
                const uint32_t ne = (uint32_t)emit.size();
                // (room for a second set: the ordered runs of the cross-block case below)
                if ((rc = c->d_dec_tmp.reserve(sizeof(BlkEmit) * 2 * (size_t)ne + 64))) return rc;
                if ((rc = c->d_hist.reserve(sizeof(BlkUnits) * 2 * (size_t)ne + 64))) return rc;
                if ((rc = c->d_codes.reserve(4 * std::max<uint64_t>(total_codes, 1)))) return rc;
                uint32_t *d_flags = (uint32_t *)c->d_dec_tmp.p;
                BlkEmit *d_emit = (BlkEmit *)((uint8_t *)c->d_dec_tmp.p + 64);
                uint64_t *dbgbuf = nullptr;
                if (c->diag.debug) {
                    if ((rc = c->d_ck.reserve(64ull * 8 * ne + 64))) return rc;
                    dbgbuf = (uint64_t *)c->d_ck.p;
                    HIP_TRY(hipMemsetAsync(dbgbuf, 0, 64ull * 8 * ne, st));
                }
                {
                    // the flags (64 zero bytes) and the emit jobs behind them in ONE transfer (a fill and a copy of their own were
                    // two launches with ten microseconds of idle GPU in front of each, between the scan and the place kernel)
                    std::vector<uint8_t> upl(64 + sizeof(BlkEmit) * (size_t)ne, 0);
                    memcpy(upl.data() + 64, emit.data(), sizeof(BlkEmit) * (size_t)ne);
                    HIP_TRY(c->small_up(d_flags, upl.data(), upl.size(), st));
                }
                // K3 keeps four units resident per CU (LDS): size the units so that all of them are resident at once
                const uint64_t slots = 4ull * (uint64_t)std::max(c->n_cu, 1);
                const uint32_t unit_target = (uint32_t)std::min<uint64_t>((total_codes + slots - 1) / slots + 1, 0x7FFFFFFFu);
                // marker units (used only when blocks read earlier blocks): two symbol units are resident per CU and the
                // symbol kernel's time does not depend on the unit size as long as every slot has a unit, while every
                // unit costs the window resolution 32 Ki lookups (256 MiB: 128 KiB units 1.31 + 0.74 ms, 512 KiB units
                // 0.59 + 0.64 ms for window resolution + substitution)
                uint32_t free_shift = 15;
                while (free_shift < 20 && (total >> (free_shift + 1)) >= 2ull * (uint64_t)std::max(c->n_cu, 1)) free_shift++;
                if (pieces_mode && units_per_piece) free_shift = 20;
                if (c->diag.free_shift >= 0) free_shift = (uint32_t)c->diag.free_shift;
                // (blocks whose codes the scan stored are moved into place; the others are decoded a second time)
                if (n_placed)
                    LAUNCH_TRY(launch_blk_place(st, d_emit, ne, (const BlkLanes *)c->d_dec_blocks.p, (const BlkLanesX *)c->d_dec_lanesx.p,
                                                (const uint32_t *)c->d_dec_temp.p, (uint32_t *)c->d_codes.p, d_flags,
                                                (BlkUnits *)c->d_hist.p, unit_target, nullptr, free_shift));
                if (n_placed < ne)
                    LAUNCH_TRY(launch_blk_emit(st, d_in, n, d_emit, ne, (const BlkLanes *)c->d_dec_blocks.p,
                                               (uint32_t *)c->d_codes.p, d_flags, (BlkUnits *)c->d_hist.p, unit_target, nullptr,
                                               c->d_dec_tabs.p, free_shift, total_codes >= 32768ull * ne, scan_small && !pieces_mode));
                c->phase("blk_emit");
                // a huge block (a schedule-S1 stream is ONE block) rarely has enough legal cuts: it goes straight to
                // the marker path, which may cut anywhere
                bool giant = pieces_multi;   // (pieces of one block read each other's output; blocks scanned in ONE piece each are ordinary blocks)
                for (const BlkEmit &e : emit) giant |= e.n_out >= (8ull << 20);   // (a block that big has too few legal cuts for K3's resident units)
                // small blocks smell of another encoder (zlib cuts every ~50-100 KiB of output; the reference at
                // block_size = 1 MiB): look at the emit flags BEFORE materialising, so that a stream which needs
                // the marker path does not pay for a discarded direct pass (costs one round trip otherwise saved)
                const bool probe = !giant && total / ne < (256u << 10);
                uint32_t fl = 0;
                if (probe) {
                    HIP_TRY(c->small_down(&fl, d_flags, 4, st));
                    HIP_TRY(c->small_sync(st));
                }
                bool ck_spec = false;
                if (!giant && !(probe && fl == 2 && !c->diag.no_markers)) {
                    LAUNCH_TRY(launch_blk_materialize(st, d_in, d_emit, ne, (const BlkLanes *)c->d_dec_blocks.p,
                                                      (const BlkUnits *)c->d_hist.p, (const uint32_t *)c->d_codes.p, d_out, dbgbuf));
                    const uint64_t tb = (pos == stop_bit ? pos : last_end), tpos = tb / 8 + ((tb & 7) ? 1 : 0);
                    if (mr.ck_mode && total && !partial && !c->diag.debug && tpos + mr.trailer_len <= n) {
                        // the checksum of what is being materialised, and the trailer bytes, behind the same synchronisation
                        c->phase("lz77_copy");
                        const uint64_t nspans = ck_nspans(total);
                        if ((rc = c->d_ck.reserve(12 * nspans))) return rc;
                        if ((rc = c->d_res.reserve(256))) return rc;
                        uint32_t *ck = (uint32_t *)c->d_ck.p;
                        LAUNCH_TRY(launch_checksum(st, d_out, total, ck, ck + nspans, ck + 2 * nspans, (EncodeResult *)c->d_res.p, mr.ck_mode));
                        HIP_TRY(hipMemcpyAsync(c->h_res, c->d_res.p, sizeof(EncodeResult), hipMemcpyDeviceToHost, st));
                        if (mr.trailer_len) HIP_TRY(hipMemcpyAsync((uint8_t *)c->h_res + 256, d_in + tpos, mr.trailer_len, hipMemcpyDeviceToHost, st));
                        ck_spec = true;
                    }
                }
                HIP_TRY(c->small_down(&fl, d_flags, 4, st));
                HIP_TRY(c->small_sync(st));
                if (ck_spec && fl == 0) {
                    const EncodeResult er = *(EncodeResult *)c->h_res;
                    mr.ck_done = true; mr.crc32 = er.crc32; mr.adler32 = er.adler32;
                    memcpy(mr.trailer, (uint8_t *)c->h_res + 256, mr.trailer_len);
                    c->phase("checksum");
                }
                if (giant && !(fl & 1) && !c->diag.no_markers) fl = 2;
                else if (giant) {   // (markers switched off, or an invalid reference: materialise normally / fall back)
                    LAUNCH_TRY(launch_blk_materialize(st, d_in, d_emit, ne, (const BlkLanes *)c->d_dec_blocks.p,
                                                      (const BlkUnits *)c->d_hist.p, (const uint32_t *)c->d_codes.p, d_out, dbgbuf));
                    HIP_TRY(hipStreamSynchronize(st));
                }
                if (!ck_spec) c->phase("lz77_copy");
                if (c->diag.debug) {
                    fprintf(stderr, "[lfx]  emit flags=%u\n", fl);
                    std::vector<BlkUnits> uv(ne);
                    (void)hipMemcpy(uv.data(), c->d_hist.p, sizeof(BlkUnits) * ne, hipMemcpyDeviceToHost);
                    for (uint32_t u = 0; u < 3 && u < ne; u++)
                        fprintf(stderr, "[lfx]  K2 block %u: units=%u hdr=%u decode=%u cut=%u select=%u\n", u, uv[u].n, uv[u].cyc[0],
                                uv[u].cyc[1], uv[u].cyc[2], uv[u].cyc[3]);
                    std::vector<uint64_t> dv(64ull * ne);
                    (void)hipMemcpy(dv.data(), dbgbuf, 64ull * 8 * ne, hipMemcpyDeviceToHost);
                    uint32_t nunits = 0;
                    uint64_t maxcyc = 0, tmin = ~0ull, tmax = 0;
                    for (uint32_t u = 0; u < 8 * ne; u++) {
                        if (!dv[u * 8 + 1]) continue;
                        nunits++;
                        maxcyc = std::max(maxcyc, dv[u * 8]);
                        tmin = std::min(tmin, dv[u * 8 + 6]); tmax = std::max(tmax, dv[u * 8 + 6]);
                    }
                    fprintf(stderr, "[lfx]  K3: units=%u max_cycles=%llu end-time spread=%.1f us (100 MHz clock)\n", nunits,
                            (unsigned long long)maxcyc, (double)(tmax - tmin) / 100.0);
                    for (uint32_t u = 0; u < 8 && u < 8 * ne; u++)
                        fprintf(stderr, "[lfx]  K3 unit %u: cycles=%llu batches=%llu nseq=%llu seq_cycles=%llu codes=%llu bytes=%llu\n", u,
                                (unsigned long long)dv[u * 8], (unsigned long long)dv[u * 8 + 1], (unsigned long long)dv[u * 8 + 2],
                                (unsigned long long)dv[u * 8 + 3], (unsigned long long)dv[u * 8 + 4], (unsigned long long)dv[u * 8 + 5]);
                }
                // (a later window of a member: the 32 KiB in front of d_out hold the member's earlier output — the markers of
                //  the first unit resolve through them; bytes in front of what `hist` covers are never looked up: a reference
                //  that far back raised flag 1 above)
                const uint8_t *init_win = hist ? d_out - MAX_WINDOW : nullptr;
                if (fl == 2 && !c->diag.no_markers) {
                    // Blocks read the output of earlier blocks (streams of other encoders; the reference's own
                    // blocks never do).  Marker-based materialisation: every block, cut into units at slice
                    // boundaries, is materialised at once into 16-bit symbols (a byte, or a reference into the
                    // unknown 32 KiB in front of the unit); one workgroup then walks the units in order resolving
                    // only each unit's last 32 KiB; a last pass replaces every marker.
                    std::vector<BlkUnits> uv(ne);
                    HIP_TRY(hipMemcpyAsync(uv.data(), c->d_hist.p, sizeof(BlkUnits) * ne, hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    std::vector<SymUnit> su;
                    uint64_t max_len = 0;
                    for (uint32_t q = 0; q < ne; q++)
                        for (uint32_t b = 0; b < uv[q].fn && b < MAX_FREE_UNITS; b++) {
                            const uint64_t len = uv[q].fout0[b + 1] - uv[q].fout0[b];
                            if (!len) continue;
                            su.push_back(SymUnit{emit[q].out_off + uv[q].fout0[b], len});
                            max_len = std::max(max_len, len);
                        }
                    const uint32_t nsu = (uint32_t)su.size();
                    if ((rc = c->d_dec_sym.reserve(2 * std::max<uint64_t>(total, 1)))) return rc;
                    if ((rc = c->d_dec_win.reserve(32768ull * std::max<uint32_t>(nsu, 1) + sizeof(SymUnit) * (size_t)nsu + 64))) return rc;
                    uint8_t *d_win = (uint8_t *)c->d_dec_win.p;
                    SymUnit *d_su = (SymUnit *)(d_win + 32768ull * std::max<uint32_t>(nsu, 1));
                    HIP_TRY(hipMemcpyAsync(d_su, su.data(), sizeof(SymUnit) * nsu, hipMemcpyHostToDevice, st));
                    LAUNCH_TRY(launch_blk_materialize_sym(st, d_in, d_emit, ne, (const BlkUnits *)c->d_hist.p,
                                                          (const uint32_t *)c->d_codes.p, (uint16_t *)c->d_dec_sym.p,
                                                          nsu <= (uint32_t)std::max(c->n_cu, 1) && !c->diag.window_chain));
                    c->phase("lz77_sym");
                    if (nsu >= 16 && !c->diag.window_chain) {   // blocked parallel prefix over the units (groups of about sqrt(nsu))
                        if ((rc = c->d_dec_maps.reserve(window_prefix_scratch_bytes(nsu)))) return rc;
                        LAUNCH_TRY(launch_window_prefix(st, (const uint16_t *)c->d_dec_sym.p, d_su, nsu, c->d_dec_maps.p, d_win, init_win));
                    } else LAUNCH_TRY(launch_window_chain(st, (const uint16_t *)c->d_dec_sym.p, d_su, nsu, d_win, init_win));
                    c->phase("win_chain");
                    LAUNCH_TRY(launch_sym_substitute(st, (const uint16_t *)c->d_dec_sym.p, d_su, nsu, d_win, d_out, max_len, init_win));
                    HIP_TRY(hipStreamSynchronize(st));
                    c->phase("substitute");
                    if (c->diag.debug) fprintf(stderr, "[lfx]  cross-block references: %u blocks, %u units through markers\n", ne, nsu);
                    fl = 0;
                }
                if (fl == 2) {
                    // (LFX_NO_MARKERS) the same blocks materialised IN ORDER instead:
                    // Blocks read the output of earlier blocks (streams of other encoders; the reference's own
                    // blocks never do): every block cannot be materialised at once.  The codes are all there, so
                    // the blocks are materialised again IN ORDER — each run of consecutive compressed blocks as
                    // one unit on one wavefront, its 32 KiB of history loaded from the output in front of it;
                    // stored blocks in between are plain copies.  Slow (one wavefront) but ~40x the serial kernel.
                    std::vector<BlkEmit> runs;
                    for (uint32_t q = 0; q < ne; ) {
                        BlkEmit r = emit[q];
                        uint32_t q2 = q + 1;
                        if (r.btype != 0) {
                            r.btype = 2;
                            uint64_t nco = r.n_codes;
                            while (q2 < ne && emit[q2].btype != 0 && nco + emit[q2].n_codes < 0xFFFFFFFFull) {
                                nco += emit[q2].n_codes; r.n_out += emit[q2].n_out; q2++;
                            }
                            r.n_codes = (uint32_t)nco;
                        }
                        r.preload = r.hist != 0;
                        runs.push_back(r);
                        q = q2;
                    }
                    const uint32_t nr = (uint32_t)runs.size();
                    std::vector<BlkUnits> ru(nr);
                    for (uint32_t q = 0; q < nr; q++) {
                        ru[q] = BlkUnits{};
                        ru[q].n = 1; ru[q].code0[0] = 0; ru[q].code0[1] = runs[q].n_codes; ru[q].out0[0] = 0; ru[q].out0[1] = runs[q].n_out;
                    }
                    BlkEmit *d_runs = (BlkEmit *)((uint8_t *)c->d_dec_tmp.p + 64) + ne;
                    BlkUnits *d_ru = (BlkUnits *)c->d_hist.p + ne;
                    HIP_TRY(hipMemcpyAsync(d_runs, runs.data(), sizeof(BlkEmit) * nr, hipMemcpyHostToDevice, st));
                    HIP_TRY(hipMemcpyAsync(d_ru, ru.data(), sizeof(BlkUnits) * nr, hipMemcpyHostToDevice, st));
                    for (uint32_t q = 0; q < nr; q++)
                        LAUNCH_TRY(launch_blk_materialize(st, d_in, d_runs + q, 1, (const BlkLanes *)c->d_dec_blocks.p, d_ru + q,
                                                          (const uint32_t *)c->d_codes.p, d_out, nullptr));
                    HIP_TRY(hipStreamSynchronize(st));
                    c->phase("lz77_chain");
                    if (c->diag.debug) fprintf(stderr, "[lfx]  cross-block references: %u blocks re-materialised in %u ordered runs\n", ne, nr);
                    fl = 0;
                }
                if (fl == 0) {   // no back-reference reached before its block's first byte
                    mr.status = LFX_OK;
                    mr.out_len = total;
                    mr.blk_out_start = total;
                    const uint64_t eb = pos == stop_bit ? pos : last_end;
                    mr.end_byte = eb / 8 + ((eb & 7) ? 1 : 0);
                    mr.end_bit = eb;
                    mr.final_seen = chain_final;
                    parallel_done = true;
                }
            }
        }
    }
    if (!parallel_done && c->diag.no_serial) { c->set_error("serial fallback disabled (LFX_NO_SERIAL)"); return LFX_E_UNSUPPORTED; }
    if (!parallel_done) {
        // ---- serial walk of the whole stream by one wavefront (exact error / partial-output semantics)
        jobs.clear();
        InflateJob j{};
        j.in_off = 0; j.in_len = n; j.start_bit = first_bit;
        j.out_off = 0; j.out_cap = cap; j.hist_avail = hist; j.flags = 0;
        j.stop_bit = stop_bit == ~0ull ? 0 : stop_bit;   // (small or irregular shards: the exact walk, ended at the shard's last bit)
        jobs.push_back(j);
        int rc;
        if ((rc = run_jobs(c, d_in, d_out, jobs, res))) return rc;
        c->phase("serial");
        InflateResult &r = res[0];
        if (stop_bit != ~0ull && r.status == 0 && (r.final_seen || r.end_bit != stop_bit)) {
            // a shard without the BFINAL block must end exactly where the next shard starts
            r.status = 1; r.err = ERR_HUFF; r.a0 = r.a1 = 0;
        }
        mr.status = map_status(r.status);
        mr.out_len = r.out_len;
        mr.blk_out_start = r.status ? r.blk_out_start : r.out_len;
        mr.end_byte = std::min<uint64_t>((r.end_bit + 7) / 8, n);
        mr.end_bit = r.end_bit;
        mr.final_seen = r.status == 0 && r.final_seen;
        mr.msg = format_error(r.err, r.a0, r.a1);
        // a window: running out of input (or of output capacity) inside a block is not a verdict — deliver the blocks in
        // front of it and let the caller come back with more.  An error in the very last bits of the window may be an
        // artefact of the cut (the reference's BitReader reads zeros past the end before it reports UnexpectedEof) and
        // is treated the same way; the caller repeats it without `partial` once the reader has ended.
        if (partial && r.status != 0 && (r.status == 2 || r.status == 3 || r.end_bit + 64 >= n * 8)) {
            mr.status = LFX_OK;
            mr.out_len = mr.blk_out_start = r.blk_out_start;
            mr.end_bit = r.blk_start_bit;
            mr.end_byte = r.blk_start_bit / 8;
            mr.final_seen = false;
            mr.need_cap = r.status == 3 && r.blk_out_start == 0;
            mr.msg.clear();
        }
    }
    return LFX_OK;
}

struct DecodeOutcome {
    int status = LFX_OK;
    uint64_t out_len = 0, delivered_len = 0, consumed = 0;
    bool header_failed = false;  // the FIRST member's container header was rejected
    std::string msg;
};

int decode_stream(Ctx *c, int format, uint32_t flags, const uint8_t *d_in, uint64_t n, uint8_t *d_out,
                  uint64_t cap, DecodeOutcome &oc) {
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    c->n_ev = 0;
    c->phase("start");
    uint64_t base = 0, out_at = 0;
    bool first = true;
    int rc;
    if ((rc = c->d_res.reserve(256))) return rc;
    if ((rc = c->d_small.reserve(70000))) return rc;
    for (;;) {
        // ---- container header (device parse, one lane)
        uint64_t off0 = 0;
        if (format != LFX_DEFLATE) {
            DecStream ds{base, n - base, 0, 0};
            DecHeader dh{};
            c->pin_reset();       // (page-locked slots for the two small transfers, lfx_ctx.h)
            HIP_TRY(c->small_up(c->d_small.p, &ds, sizeof ds, st));
            LAUNCH_TRY(launch_container(st, format, 1, d_in, (const DecStream *)c->d_small.p,
                                        (DecHeader *)((uint8_t *)c->d_small.p + 256)));
            HIP_TRY(c->small_down(&dh, (uint8_t *)c->d_small.p + 256, sizeof dh, st));
            HIP_TRY(c->small_sync(st));
            if (dh.status != 0) {
                if (!first && dh.status == 2) {  // MultiDecoder: UnexpectedEof on the next header = clean end
                    oc.consumed = n;             // (gzip.rs:1150-1156; the partial header bytes were read)
                    break;
                }
                oc.status = map_status(dh.status);
                oc.header_failed = first;
                oc.msg = format_error(dh.err, dh.a0, dh.a1);
                oc.consumed = base + dh.deflate_off;
                oc.out_len = oc.delivered_len = out_at;
                return LFX_OK;
            }
            off0 = dh.deflate_off;
        }
        MemberResult mr;
        mr.ck_mode = format == LFX_GZIP ? 1 : format == LFX_ZLIB ? 2 : 0;
        mr.trailer_len = format == LFX_GZIP ? 8 : format == LFX_ZLIB ? 4 : 0;
        if ((rc = inflate_member(c, d_in + base, n - base, off0, d_out + out_at, cap - out_at, mr))) return rc;
        oc.out_len = out_at + mr.out_len;
        oc.delivered_len = out_at + mr.blk_out_start;
        oc.consumed = base + mr.end_byte;
        if (mr.status != LFX_OK) { oc.status = mr.status; oc.msg = mr.msg; return LFX_OK; }
        // ---- trailer
        if (format != LFX_DEFLATE) {
            const uint64_t need = format == LFX_GZIP ? 8 : 4;
            const uint64_t tpos = base + mr.end_byte;
            if (n - tpos < need) {
                oc.status = LFX_E_UNEXPECTED_EOF;
                oc.msg = "failed to fill whole buffer";
                oc.consumed = n;
                return LFX_OK;
            }
            uint8_t t[8];
            EncodeResult er{};
            if (mr.ck_done) {                   // (came back with the materialisation's verdict)
                er.crc32 = mr.crc32; er.adler32 = mr.adler32;
                memcpy(t, mr.trailer, need);
            } else {
                const uint64_t nspans = ck_nspans(mr.out_len);
                if ((rc = c->d_ck.reserve(12 * nspans))) return rc;
                uint32_t *ck = (uint32_t *)c->d_ck.p;
                LAUNCH_TRY(launch_checksum(st, d_out + out_at, mr.out_len, ck, ck + nspans, ck + 2 * nspans, (EncodeResult *)c->d_res.p,
                                           format == LFX_GZIP ? 1 : format == LFX_ZLIB ? 2 : 3));
                HIP_TRY(hipMemcpyAsync(c->h_res, c->d_res.p, sizeof(EncodeResult), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipMemcpyAsync(t, d_in + tpos, need, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                c->phase("checksum");
                er = *(EncodeResult *)c->h_res;
            }
            oc.consumed = tpos + need;
            if (format == LFX_GZIP) {
                const uint32_t crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                if (crc != er.crc32) {  // gzip.rs:1035-1040 (ISIZE is read but never verified)
                    oc.status = LFX_E_INVALID_DATA;
                    oc.msg = format_error(ERR_CRC32, er.crc32, crc);
                    return LFX_OK;
                }
            } else {
                const uint32_t ad = (uint32_t)t[0] << 24 | (uint32_t)t[1] << 16 | (uint32_t)t[2] << 8 | t[3];
                if (ad != er.adler32) {
                    oc.status = LFX_E_INVALID_DATA;
                    oc.msg = format_error(ERR_ADLER32, er.adler32, ad);
                    return LFX_OK;
                }
            }
        }
        out_at = oc.out_len;
        if (!(format == LFX_GZIP && (flags & LFX_DEC_MULTI))) break;
        base = oc.consumed;
        first = false;
    }
    oc.out_len = oc.delivered_len = out_at;
    c->phase("done");
    return LFX_OK;
}

}  // namespace

extern "C" int lfx_decode_device(lfx_ctx *cc, int format, uint32_t flags, const void *d_in, uint64_t n,
                                 void *d_out, uint64_t cap, uint64_t *out_len, uint64_t *consumed) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (format < 0 || format > 2) return LFX_E_ARG;
    DecodeOutcome oc;
    int rc = decode_stream(c, format, flags, (const uint8_t *)d_in, n, (uint8_t *)d_out, cap, oc);
    if (rc) return rc;
    if (oc.out_len > cap) oc.out_len = cap;  // defensive: never report more than the buffer holds
    if (out_len) *out_len = oc.out_len;
    if (consumed) *consumed = oc.consumed;
    if (oc.status != LFX_OK) c->set_error(oc.msg);
    return oc.status;
} LFX_ABI_CATCH

extern "C" int lfx_decode_shard_device(lfx_ctx *cc, const void *d_in, uint64_t n, uint64_t start_bit,
                                       uint64_t total_bits, int is_last, void *d_out, uint64_t cap,
                                       uint64_t *out_len) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    c->n_ev = 0;
    c->phase("start");
    MemberResult mr;
    int rc = inflate_member(c, (const uint8_t *)d_in, n, start_bit >> 3, (uint8_t *)d_out, cap, mr, start_bit,
                            is_last ? ~0ull : start_bit + total_bits);
    if (rc) return rc;
    if (out_len) *out_len = mr.out_len;
    if (mr.status != LFX_OK) c->set_error(mr.msg);
    return mr.status;
} LFX_ABI_CATCH

// ------------------------------------------------------------------------------------------------
// N-GPU decode of ONE member without the encoder's help: the member is cut by compressed BYTES;
// every rank finds and scans the blocks that START in its byte range (lfx_decode_range_scan), the ranks exchange one
// tuple per candidate (one all-gather), every rank walks the same chain over the gathered table (lfx_decode_chain) and
// materialises the blocks it owns into its slice of the output (lfx_decode_range_emit).  The container checksum is
// folded from the ranks' slice checksums (lfx_crc32_combine / lfx_adler32_combine).
static_assert(sizeof(lfx_blk_tuple) == 56, "tuple layout (all-gathered as raw bytes)");

extern "C" int lfx_decode_range_scan(lfx_ctx *cc, const void *d_part_, uint64_t n_part, uint64_t lo_byte, uint64_t hi_byte,
                                     uint64_t first_bit, uint64_t final_from_bit, uint32_t rank, lfx_blk_tuple *tuples, uint32_t cap,
                                     uint32_t *count) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    const uint8_t *d_in = (const uint8_t *)d_part_;
    if (!count || hi_byte < lo_byte || n_part < hi_byte - lo_byte) return LFX_E_ARG;
    *count = 0;
    c->n_ev = 0;
    c->phase("start");
    const uint64_t n = n_part, range_bits = (hi_byte - lo_byte) * 8, base_bit = lo_byte * 8;
    // ---- block-start candidates in the local bytes (the tail behind hi_byte is searched too: a false candidate there
    //      still ends the range guess of the last real block early, exactly as on one GPU)
    const uint32_t shard_cap = find_shard_cap(n);
    const uint32_t final_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1u << 16, n / 4096), 1u << 24);
    int rc;
    if ((rc = c->d_dec_cand.reserve(8ull * shard_cap * FIND_SHARDS + 8ull * final_cap + 4 * FIND_HDR_WORDS))) return rc;
    uint32_t *d_count = (uint32_t *)c->d_dec_cand.p;
    uint32_t *d_final_count = d_count + FIND_HDR_FINAL;
    uint64_t *d_cand = (uint64_t *)((uint8_t *)c->d_dec_cand.p + 4 * FIND_HDR_WORDS);
    uint64_t *d_final = d_cand + (uint64_t)shard_cap * FIND_SHARDS;
    HIP_TRY(hipMemsetAsync(d_count, 0, 4 * FIND_HDR_WORDS, st));
    std::vector<uint64_t> starts;
    if (n >= 16) {
        // headers with BFINAL set are reported from member bit `final_from_bit` on (the finder's tail rule, inflate_member: a
        // member's last block is the only one that carries the flag — everywhere else the offsets whose bit 0 is set are
        // dropped, which halves the candidates of both stages and the scan jobs).  The chain is walked on tuples, so a last
        // block that starts in front of that bit breaks the chain: the caller then scans again with final_from_bit = 0.
        const uint64_t base0 = lo_byte * 8;
        const uint64_t final_local = final_from_bit <= base0 ? 0 : std::min<uint64_t>(final_from_bit - base0, ~0ull >> 1);
        LAUNCH_TRY(launch_find_stage1(st, d_in, n, 0, d_count, d_cand, shard_cap, final_local, (uint32_t)std::max(c->n_cu, 1)));
        LAUNCH_TRY(launch_find_stage2(st, d_in, n, d_cand, shard_cap, d_count, d_count + FIND_HDR_WORK, d_final_count, d_final, final_cap,
                                      (uint32_t)std::max(c->n_cu, 1)));
        uint32_t hc[FIND_HDR_FINAL + 1];
        HIP_TRY(hipMemcpyAsync(hc, d_count, sizeof hc, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        bool overflow = hc[FIND_SHARDS] != 0;
        for (uint32_t k = 0; k < FIND_SHARDS; k++) if (hc[k] > shard_cap) overflow = true;
        if (overflow) { c->set_error("block finder overflow"); return LFX_E_UNSUPPORTED; }
        uint32_t nf = hc[FIND_HDR_FINAL];
        if (nf > final_cap) nf = final_cap;
        starts.resize(nf);
        if (nf) HIP_TRY(hipMemcpy(starts.data(), d_final, 8ull * nf, hipMemcpyDeviceToHost));
    }
    c->phase("find");
    if (first_bit != ~0ull) {            // the member's first block (its start is known: right behind the container header)
        if (first_bit < base_bit || first_bit - base_bit >= n * 8) return LFX_E_ARG;
        starts.push_back(first_bit - base_bit);
    }
    std::sort(starts.begin(), starts.end());
    starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
    // jobs: the candidates that start inside the range; a job's range guess ends at the next candidate (inside or behind it)
    uint32_t nc = 0;
    while (nc < starts.size() && starts[nc] < range_bits) nc++;
    if (nc > cap) { c->set_error("more candidates than the caller's tuple buffer holds"); return LFX_E_NOSPACE; }
    const size_t tab_bytes = blk_tabs_bytes();
    c->range_stored.clear();
    if (nc) {
        auto start_at = [&](uint32_t i) { return i < starts.size() ? starts[i] : n * 8; };
        std::vector<BlkJob> bj(nc);
        for (uint32_t i = 0; i < nc; i++) bj[i] = BlkJob{starts[i], start_at(i + 1)};
        if ((rc = c->d_dec_streams.reserve(sizeof(BlkJob) * (nc + 1)))) return rc;
        if ((rc = c->d_dec_state.reserve(sizeof(BlkInfo) * (nc + 1)))) return rc;
        if ((rc = c->d_dec_blocks.reserve(sizeof(BlkLanes) * (size_t)(nc + 1)))) return rc;
        if ((rc = c->d_dec_tabs.reserve(tab_bytes * (nc + 1)))) return rc;
        // ONE Huffman pass for large blocks, as in inflate_member (round 6): the scan stores every lane's code words in a region of
        // its own, lfx_decode_range_emit moves the owned blocks' codes into place (blk_place_kernel) instead of decoding them again
        bool store_mode = !c->diag.two_pass && (n * 8) / nc >= (1ull << 20);
        if (store_mode) {
            uint64_t off = 0;
            for (uint32_t j = 0; j < nc; j++) {
                const uint64_t bits = bj[j].end_bit > bj[j].start_bit ? bj[j].end_bit - bj[j].start_bit : 0;
                const uint64_t slice = std::max<uint64_t>((bits + 1023) / 1024, 128);
                const uint64_t lcap = (slice / (c->diag.store_tight ? 16 : 2) + 448 + 64 + 3) & ~3ull;   // (448 = SCAN_HEADCAP, lfx_inflate_fast.hip)
                bj[j].temp_off = off;
                bj[j].cap = (uint32_t)lcap;
                off += 1024 * lcap;
            }
            if (off * 4 > (16ull << 30) || c->d_dec_temp.reserve(off * 4) || c->d_dec_lanesx.reserve(sizeof(BlkLanesX) * (size_t)(nc + 1))) {
                store_mode = false;
                for (uint32_t j = 0; j < nc; j++) { bj[j].temp_off = 0; bj[j].cap = 0; }
            }
        }
        HIP_TRY(hipMemcpyAsync(c->d_dec_streams.p, bj.data(), sizeof(BlkJob) * nc, hipMemcpyHostToDevice, st));
        if (store_mode)
            LAUNCH_TRY(launch_blk_scan_store(st, d_in, n, (const BlkJob *)c->d_dec_streams.p, nc, (BlkInfo *)c->d_dec_state.p,
                                             (BlkLanes *)c->d_dec_blocks.p, c->d_dec_tabs.p, (uint32_t *)c->d_dec_temp.p,
                                             (BlkLanesX *)c->d_dec_lanesx.p));
        else
            LAUNCH_TRY(launch_blk_scan(st, d_in, n, (const BlkJob *)c->d_dec_streams.p, nc, (BlkInfo *)c->d_dec_state.p,
                                       (BlkLanes *)c->d_dec_blocks.p, c->d_dec_tabs.p));
        std::vector<BlkInfo> bi(nc);
        HIP_TRY(hipMemcpyAsync(bi.data(), c->d_dec_state.p, sizeof(BlkInfo) * nc, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (BlkInfo &b : bi) if (b.status == BLK_OK && b.end_bit > n * 8) b.status = BLK_NO_EOB;
        c->range_stored.assign(nc, 0);
        c->range_temp_off.assign(nc, 0);
        c->range_cap.assign(nc, 0);
        if (store_mode)
            for (uint32_t j = 0; j < nc; j++) {
                // (a compressed block that scanned to its EndOfBlock and whose lanes' codes all fitted their regions; a job that is
                //  scanned again below loses the mark: the plain scan rewrites its lanes)
                c->range_stored[j] = bi[j].status == BLK_OK && bi[j].btype != 0 && bi[j]._pad == 0;
                c->range_temp_off[j] = bj[j].temp_off;
                c->range_cap[j] = bj[j].cap;
            }
        // a false candidate inside a block cuts that block's range guess short (no EndOfBlock): rescan with wider ranges
        for (uint32_t widen = 2; widen <= 6; widen++) {
            std::vector<uint32_t> redo;
            for (uint32_t i = 0; i < nc; i++) if (bi[i].status == BLK_NO_EOB && i + widen <= starts.size()) redo.push_back(i);
            if (redo.empty()) break;
            std::vector<BlkJob> rj(redo.size());
            for (size_t q = 0; q < redo.size(); q++) { rj[q] = BlkJob{starts[redo[q]], start_at(redo[q] + widen)}; c->range_stored[redo[q]] = 0; }
            if ((rc = c->d_dec_tmp.reserve(sizeof(BlkJob) * redo.size()))) return rc;
            BlkJob *d_rj = (BlkJob *)c->d_dec_tmp.p;
            HIP_TRY(hipMemcpyAsync(d_rj, rj.data(), sizeof(BlkJob) * redo.size(), hipMemcpyHostToDevice, st));
            for (size_t q = 0; q < redo.size(); q++)
                LAUNCH_TRY(launch_blk_scan(st, d_in, n, d_rj + q, 1, (BlkInfo *)c->d_dec_state.p + redo[q],
                                           (BlkLanes *)c->d_dec_blocks.p + redo[q], (uint8_t *)c->d_dec_tabs.p + tab_bytes * redo[q]));
            for (size_t q = 0; q < redo.size(); q++)
                HIP_TRY(hipMemcpyAsync(&bi[redo[q]], (BlkInfo *)c->d_dec_state.p + redo[q], sizeof(BlkInfo), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        for (uint32_t i = 0; i < nc; i++) {
            lfx_blk_tuple &t = tuples[i];
            t = lfx_blk_tuple{};
            t.start_bit = base_bit + starts[i];
            t.end_bit = base_bit + bi[i].end_bit;
            t.n_out = bi[i].n_out;
            t.n_codes = bi[i].n_codes;
            t.btype = (uint8_t)bi[i].btype;
            t.bfinal = (uint8_t)bi[i].bfinal;
            // (a block cut by the end of the local bytes is incomplete whatever its scan says — see inflate_member)
            t.status = (uint8_t)(bi[i].status == BLK_OK && bi[i].end_bit > n * 8 ? (uint32_t)BLK_NO_EOB : bi[i].status);
            t._pad = 0;
            t.slot = i;
            t.rank = (uint16_t)rank;
            t._pad2 = 0;
            t.nlanes = bi[i].nlanes;
            t.data_bit = base_bit + bi[i].data_bit;
        }
    }
    c->phase("blk_scan");
    *count = nc;
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_decode_chain(const lfx_blk_tuple *all, uint32_t n_all, uint64_t first_bit, uint32_t *chain, uint32_t cap,
                                uint32_t *n_chain, uint64_t *total_out) try {
    if (!all || !chain || !n_chain) return LFX_E_ARG;
    // the true block list: from the known first block, every block starts where the one before it ended; false candidates
    // (inside a block) are never reached.  Deterministic: every rank computes the same list from the same table.
    std::vector<uint32_t> order(n_all);
    for (uint32_t i = 0; i < n_all; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return all[a].start_bit != all[b].start_bit ? all[a].start_bit < all[b].start_bit : all[a].rank < all[b].rank;
    });
    uint64_t pos = first_bit, total = 0;
    uint32_t k = 0;
    for (;;) {
        auto it = std::lower_bound(order.begin(), order.end(), pos, [&](uint32_t a, uint64_t v) { return all[a].start_bit < v; });
        if (it == order.end() || all[*it].start_bit != pos) return LFX_E_UNSUPPORTED;     // a block the finder cannot see (stored / fixed), or damage
        const lfx_blk_tuple &t = all[*it];
        if (t.status != BLK_OK || t.end_bit <= pos) return LFX_E_UNSUPPORTED;
        if (k >= cap) return LFX_E_NOSPACE;
        chain[k++] = *it;
        total += t.n_out;
        if (t.bfinal) break;
        pos = t.end_bit;
    }
    *n_chain = k;
    if (total_out) *total_out = total;
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_decode_range_emit(lfx_ctx *cc, const void *d_part_, uint64_t n_part, uint64_t lo_byte, const lfx_blk_tuple *all,
                                     const uint32_t *chain, uint32_t n_chain, uint32_t rank, void *d_out_, uint64_t cap,
                                     uint64_t *out_len, uint64_t *out_base, uint32_t *state) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    const uint8_t *d_in = (const uint8_t *)d_part_;
    uint8_t *d_out = (uint8_t *)d_out_;
    const uint64_t base_bit = lo_byte * 8;
    c->range = RangeState{};
    // the chain blocks this rank owns (consecutive in stream order: ownership goes by start position)
    std::vector<BlkEmit> emit;
    uint64_t before = 0, total = 0, total_codes = 0;
    uint32_t n_placed = 0;
    bool seen = false;
    for (uint32_t q = 0; q < n_chain; q++) {
        const lfx_blk_tuple &t = all[chain[q]];
        if (t.rank != rank) { if (!seen) before += t.n_out; continue; }
        seen = true;
        BlkEmit e{};
        e.start_bit = t.start_bit - base_bit; e.data_bit = t.data_bit - base_bit; e.code_off = total_codes; e.out_off = total;
        e.n_out = t.n_out; e.n_codes = t.n_codes; e.nlanes = t.nlanes; e.btype = t.btype; e.cand = t.slot;
        e.hist = before + total;             // bytes of the member in front of the block: bounds its back-references
        if (t.slot < c->range_stored.size() && c->range_stored[t.slot]) {      // (this rank's scan kept the block's code words)
            e.placed = 1; e.temp_off = c->range_temp_off[t.slot]; e.cap = c->range_cap[t.slot]; n_placed++;
        }
        emit.push_back(e);
        total += t.n_out;
        total_codes += t.n_codes;
    }
    if (out_base) *out_base = before;
    if (out_len) *out_len = total;
    if (state) *state = 0;
    if (total > cap) { c->set_error("output capacity too small"); return LFX_E_NOSPACE; }
    c->range.state = 0; c->range.total = total; c->range.d_out = d_out; c->range.before = before;
    if (emit.empty()) return LFX_OK;
    const uint32_t ne = (uint32_t)emit.size();
    int rc;
    if ((rc = c->d_dec_tmp.reserve(sizeof(BlkEmit) * (size_t)ne + 64))) return rc;
    if ((rc = c->d_hist.reserve(sizeof(BlkUnits) * (size_t)ne + 64))) return rc;
    if ((rc = c->d_codes.reserve(4 * std::max<uint64_t>(total_codes, 1)))) return rc;
    uint32_t *d_flags = (uint32_t *)c->d_dec_tmp.p;
    BlkEmit *d_emit = (BlkEmit *)((uint8_t *)c->d_dec_tmp.p + 64);
    HIP_TRY(hipMemsetAsync(d_flags, 0, 64, st));
    HIP_TRY(hipMemcpyAsync(d_emit, emit.data(), sizeof(BlkEmit) * ne, hipMemcpyHostToDevice, st));
    const uint64_t slots = 4ull * (uint64_t)std::max(c->n_cu, 1);
    const uint32_t unit_target = (uint32_t)std::min<uint64_t>((total_codes + slots - 1) / slots + 1, 0x7FFFFFFFu);
    uint32_t free_shift = 15;   // marker units as on one GPU: two resident per CU, as large as that allows
    while (free_shift < 20 && (total >> (free_shift + 1)) >= 2ull * (uint64_t)std::max(c->n_cu, 1)) free_shift++;
    if (n_placed)
        LAUNCH_TRY(launch_blk_place(st, d_emit, ne, (const BlkLanes *)c->d_dec_blocks.p, (const BlkLanesX *)c->d_dec_lanesx.p,
                                    (const uint32_t *)c->d_dec_temp.p, (uint32_t *)c->d_codes.p, d_flags, (BlkUnits *)c->d_hist.p,
                                    unit_target, nullptr, free_shift));
    if (n_placed < ne)
        LAUNCH_TRY(launch_blk_emit(st, d_in, n_part, d_emit, ne, (const BlkLanes *)c->d_dec_blocks.p, (uint32_t *)c->d_codes.p, d_flags,
                                   (BlkUnits *)c->d_hist.p, unit_target, nullptr, c->d_dec_tabs.p, free_shift, total_codes >= 32768ull * ne));
    c->phase("blk_emit");
    // small blocks smell of another encoder: look at the flags before materialising (as inflate_member does); the reference's
    // 1 MiB blocks are materialised at once and the flags read afterwards
    const bool probe = total / ne < (256u << 10);
    uint32_t fl = 0;
    if (probe) {
        HIP_TRY(hipMemcpyAsync(&fl, d_flags, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (!(probe && fl)) {
        LAUNCH_TRY(launch_blk_materialize(st, d_in, d_emit, ne, (const BlkLanes *)c->d_dec_blocks.p, (const BlkUnits *)c->d_hist.p,
                                          (const uint32_t *)c->d_codes.p, d_out, nullptr));
        HIP_TRY(hipMemcpyAsync(&fl, d_flags, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    c->phase("lz77_copy");
    if (fl & 1) {
        c->range.state = -1;
        c->set_error("a back-reference reaches in front of the member's first byte: decode it with lfx_decode_device for the exact error");
        return LFX_E_INVALID_DATA;
    }
    if (fl == 2) {
        // blocks that read the output of earlier blocks (another encoder's member), possibly of blocks another rank owns: the
        // slice is materialised as 16-bit symbols (a byte, or "byte j of the 32 KiB in front of my unit"); the window in front
        // of the slice arrives with lfx_decode_range_finish
        std::vector<BlkUnits> uv(ne);
        HIP_TRY(hipMemcpyAsync(uv.data(), c->d_hist.p, sizeof(BlkUnits) * ne, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<SymUnit> su;
        uint64_t max_len = 0;
        for (uint32_t q = 0; q < ne; q++)
            for (uint32_t b = 0; b < uv[q].fn && b < MAX_FREE_UNITS; b++) {
                const uint64_t len = uv[q].fout0[b + 1] - uv[q].fout0[b];
                if (!len) continue;
                su.push_back(SymUnit{emit[q].out_off + uv[q].fout0[b], len});
                max_len = std::max(max_len, len);
            }
        const uint32_t nsu = (uint32_t)su.size();
        if ((rc = c->d_dec_sym.reserve(2 * std::max<uint64_t>(total, 1)))) return rc;
        if ((rc = c->d_dec_win.reserve(32768ull * (std::max<uint32_t>(nsu, 1) + 1) + sizeof(SymUnit) * (size_t)nsu + 64))) return rc;
        if ((rc = c->d_dec_maps.reserve(window_prefix_scratch_bytes(std::max<uint32_t>(nsu, 1))))) return rc;
        SymUnit *d_su = (SymUnit *)((uint8_t *)c->d_dec_win.p + 32768ull * (std::max<uint32_t>(nsu, 1) + 1));
        HIP_TRY(hipMemcpyAsync(d_su, su.data(), sizeof(SymUnit) * nsu, hipMemcpyHostToDevice, st));
        LAUNCH_TRY(launch_blk_materialize_sym(st, d_in, d_emit, ne, (const BlkUnits *)c->d_hist.p, (const uint32_t *)c->d_codes.p,
                                              (uint16_t *)c->d_dec_sym.p, nsu <= (uint32_t)std::max(c->n_cu, 1)));
        HIP_TRY(hipStreamSynchronize(st));     // (su must outlive its copy)
        c->phase("lz77_sym");
        c->range.state = 1; c->range.nsu = nsu; c->range.max_len = max_len;
        if (state) *state = 1;
    }
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_decode_range_map(lfx_ctx *cc, void *d_map) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    if (c->range.state < 0 || !d_map) { c->set_error("lfx_decode_range_map: no range_emit on this context"); return LFX_E_ARG; }
    if (c->range.state == 1) {
        const uint32_t nsu = c->range.nsu;
        const SymUnit *d_su = (const SymUnit *)((uint8_t *)c->d_dec_win.p + 32768ull * (std::max<uint32_t>(nsu, 1) + 1));
        LAUNCH_TRY(launch_window_rank_map(c->stream, (const uint16_t *)c->d_dec_sym.p, d_su, nsu, c->d_dec_maps.p, (uint16_t *)d_map));
    } else LAUNCH_TRY(launch_bytes_to_map(c->stream, c->range.d_out, c->range.total, (uint16_t *)d_map));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_decode_range_finish(lfx_ctx *cc, const void *d_maps, uint32_t rank, uint32_t *crc32, uint32_t *adler32) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    if (crc32) *crc32 = 0;
    if (adler32) *adler32 = 1;
    if (c->range.state < 0) { c->set_error("lfx_decode_range_finish: no range_emit on this context"); return LFX_E_ARG; }
    const uint64_t total = c->range.total;
    uint8_t *d_out = c->range.d_out;
    int rc;
    if (c->range.state == 1) {
        const uint32_t nsu = c->range.nsu;
        uint8_t *d_win = (uint8_t *)c->d_dec_win.p;
        uint8_t *d_init = d_win + 32768ull * std::max<uint32_t>(nsu, 1);        // the window in front of the slice
        const SymUnit *d_su = (const SymUnit *)(d_init + 32768);
        const uint8_t *init_win = nullptr;
        if (c->range.before) {
            if (!d_maps || !rank) { c->set_error("this slice reads the output of the ranks in front of it: their maps are needed"); return LFX_E_ARG; }
            if ((uintptr_t)d_maps & 15) { c->set_error("the gathered maps must be 16-byte aligned"); return LFX_E_ARG; }
            LAUNCH_TRY(launch_window_ranks(st, (const uint16_t *)d_maps, rank, d_init));
            init_win = d_init;
        }
        if (nsu >= 16 && !c->diag.window_chain)
            LAUNCH_TRY(launch_window_prefix(st, (const uint16_t *)c->d_dec_sym.p, d_su, nsu, c->d_dec_maps.p, d_win, init_win));
        else LAUNCH_TRY(launch_window_chain(st, (const uint16_t *)c->d_dec_sym.p, d_su, nsu, d_win, init_win));
        c->phase("win_chain");
        LAUNCH_TRY(launch_sym_substitute(st, (const uint16_t *)c->d_dec_sym.p, d_su, nsu, d_win, d_out, c->range.max_len, init_win));
        c->phase("substitute");
        c->range.state = 0;        // the slice holds bytes now
    }
    if (total) {
        if ((rc = c->d_res.reserve(256))) return rc;
        const uint64_t nspans = ck_nspans(total);
        if ((rc = c->d_ck.reserve(12 * nspans))) return rc;
        uint32_t *ck = (uint32_t *)c->d_ck.p;
        LAUNCH_TRY(launch_checksum(st, d_out, total, ck, ck + nspans, ck + 2 * nspans, (EncodeResult *)c->d_res.p, 3));
        HIP_TRY(hipMemcpyAsync(c->h_res, c->d_res.p, sizeof(EncodeResult), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const EncodeResult er = *(EncodeResult *)c->h_res;
        if (crc32) *crc32 = er.crc32;
        if (adler32) *adler32 = er.adler32;
    } else HIP_TRY(hipStreamSynchronize(st));
    c->phase("checksum");
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_decode_host(lfx_ctx *cc, int format, uint32_t flags, const void *in, uint64_t n, void *out,
                               uint64_t cap, uint64_t *out_len, uint64_t *consumed) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    int rc;
    if ((rc = c->d_io_in.reserve(std::max<uint64_t>(n, 4)))) return rc;
    if ((rc = c->d_io_out.reserve(std::max<uint64_t>(cap, 4)))) return rc;
    // (lfx_hostio.h: page-locked buffers — lfx_host_alloc — go to the DMA engine as they are, pageable ones through slabs)
    if (int hr = host_to_device(c, c->d_io_in.p, in, n, c->stream)) { c->set_error("host to device copy failed"); return hr; }
    uint64_t ol = 0;
    rc = lfx_decode_device(cc, format, flags, c->d_io_in.p, n, c->d_io_out.p, cap, &ol, consumed);
    // (a page-locked `in` was only queued for DMA: no return before the stream has passed the copy, on any path)
    if (rc == LFX_E_DEVICE || rc == LFX_E_OOM || rc == LFX_E_ARG) { (void)hipStreamSynchronize(c->stream); return rc; }
    if (ol) { if (int hr = device_to_host(c, out, c->d_io_out.p, ol, c->stream)) { c->set_error("device to host copy failed"); return hr; } }
    if (out_len) *out_len = ol;
    return rc;
} LFX_ABI_CATCH

// Batch fast path: every stream's blocks go through the lane-parallel kernels, one block per stream and
// round (a block's start is only known once the block before it has been scanned; reference-made streams
// have two blocks).  A stream leaves the fast path — and is decoded again, exactly, by the serial kernel —
// on any anomaly: undecodable block, a block that runs past the stream's end or its output capacity, a
// back-reference that reaches in front of its block, more than MAX_ROUNDS blocks.
// fast[i] = 1: d_out holds the stream's bytes and res[i] is filled in.
namespace lfx {
static int batch_fast(Ctx *c, const uint8_t *d_in, uint64_t n_in, uint8_t *d_out, uint32_t count,
                      const std::vector<InflateJob> &jobs, std::vector<uint8_t> &fast, std::vector<InflateResult> &res) {
    constexpr uint32_t MAX_ROUNDS = 4;
    hipStream_t st = c->stream;
    struct Live { uint32_t stream; uint64_t bit, produced; uint32_t nblocks; };
    std::vector<Live> live;
    fast.assign(count, 0);
    res.assign(count, InflateResult{});
    for (uint32_t i = 0; i < count; i++)
        if (jobs[i].in_len >= 64 && jobs[i].in_off + jobs[i].in_len <= n_in)
            live.push_back(Live{i, jobs[i].in_off * 8 + jobs[i].start_bit, 0, 0});
    int rc;
    for (uint32_t round = 0; round < MAX_ROUNDS && !live.empty(); round++) {
        const uint32_t nj = (uint32_t)live.size();
        std::vector<BlkJob> bj(nj);
        for (uint32_t k = 0; k < nj; k++) {
            const InflateJob &j = jobs[live[k].stream];
            bj[k] = BlkJob{live[k].bit, (j.in_off + j.in_len) * 8};
        }
        if ((rc = c->d_dec_streams.reserve(sizeof(BlkJob) * nj))) return rc;
        if ((rc = c->d_dec_state.reserve(sizeof(BlkInfo) * nj))) return rc;
        if ((rc = c->d_dec_cand.reserve(sizeof(BlkLanes) * (size_t)nj))) return rc;
        if ((rc = c->d_dec_tabs.reserve(blk_tabs_bytes() * nj))) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_dec_streams.p, bj.data(), sizeof(BlkJob) * nj, hipMemcpyHostToDevice, st));
        // (streams of a few tens of KB: the 256-lane instances of the scan and the emit kernel — a 32 KB block in 1024 slices is
        //  17 symbols a lane; LFX_NO_SMALL_SCAN=1 keeps 1024)
        uint64_t range_bits = 0;
        for (uint32_t k = 0; k < nj; k++) range_bits += bj[k].end_bit - bj[k].start_bit;
        const bool small = !c->diag.no_small_scan && range_bits / nj < (512ull << 10);
        LAUNCH_TRY(launch_blk_scan(st, d_in, n_in, (const BlkJob *)c->d_dec_streams.p, nj, (BlkInfo *)c->d_dec_state.p,
                                   (BlkLanes *)c->d_dec_cand.p, c->d_dec_tabs.p, small));
        // (phase brackets of the first rounds only: the timer holds sixteen, and "fast" / "inflate" / "verify" close the call)
        const bool stamp = c->n_ev + 6 < 17;
        if (stamp) c->phase("blk_scan");
        std::vector<BlkInfo> bi(nj);
        HIP_TRY(hipMemcpyAsync(bi.data(), c->d_dec_state.p, sizeof(BlkInfo) * nj, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // blocks that scanned cleanly and fit are emitted; the others drop their stream out of the fast path
        std::vector<BlkEmit> emit;
        std::vector<uint32_t> owner;
        uint64_t total_codes = 0;
        for (uint32_t k = 0; k < nj; k++) {
            const InflateJob &j = jobs[live[k].stream];
            const BlkInfo &r = bi[k];
            if (r.status != BLK_OK || r.end_bit <= live[k].bit || r.end_bit > (j.in_off + j.in_len) * 8 ||
                live[k].produced + r.n_out > j.out_cap)
                continue;
            BlkEmit e{};
            e.start_bit = live[k].bit; e.data_bit = r.data_bit; e.code_off = total_codes;
            e.out_off = j.out_off + live[k].produced;
            e.n_out = r.n_out; e.n_codes = r.n_codes; e.nlanes = r.nlanes; e.btype = r.btype; e.cand = k;
            e.hist = live[k].produced;
            e.preload = live[k].produced != 0;
            emit.push_back(e);
            owner.push_back(k);
            total_codes += r.n_codes;
        }
        const uint32_t ne = (uint32_t)emit.size();
        std::vector<uint32_t> jf(ne, 0);
        if (ne) {
            if ((rc = c->d_dec_tmp.reserve(sizeof(BlkEmit) * ne + 4ull * ne + 128))) return rc;
            if ((rc = c->d_hist.reserve(sizeof(BlkUnits) * (size_t)ne + 64))) return rc;
            if ((rc = c->d_codes.reserve(4 * std::max<uint64_t>(total_codes, 1)))) return rc;
            uint32_t *d_flags = (uint32_t *)c->d_dec_tmp.p;
            uint32_t *d_jf = d_flags + 16;
            BlkEmit *d_emit = (BlkEmit *)((uint8_t *)c->d_dec_tmp.p + 64 + 4ull * ne + (8 - (4ull * ne) % 8) % 8);
            HIP_TRY(hipMemsetAsync(d_flags, 0, 64 + 4ull * ne, st));
            HIP_TRY(hipMemcpyAsync(d_emit, emit.data(), sizeof(BlkEmit) * ne, hipMemcpyHostToDevice, st));
            const uint64_t slots = 4ull * (uint64_t)std::max(c->n_cu, 1);
            const uint32_t unit_target = (uint32_t)std::min<uint64_t>((total_codes + slots - 1) / slots + 1, 0x7FFFFFFFu);
            LAUNCH_TRY(launch_blk_emit(st, d_in, n_in, d_emit, ne, (const BlkLanes *)c->d_dec_cand.p, (uint32_t *)c->d_codes.p,
                                       d_flags, (BlkUnits *)c->d_hist.p, unit_target, d_jf, c->d_dec_tabs.p, 17, false, small));
            if (stamp) c->phase("blk_emit");
            LAUNCH_TRY(launch_blk_materialize(st, d_in, d_emit, ne, (const BlkLanes *)c->d_dec_cand.p,
                                              (const BlkUnits *)c->d_hist.p, (const uint32_t *)c->d_codes.p, d_out, nullptr));
            if (stamp) c->phase("lz77_copy");
            HIP_TRY(hipMemcpyAsync(jf.data(), d_jf, 4ull * ne, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        std::vector<Live> next;
        for (uint32_t q = 0; q < ne; q++) {
            if (jf[q]) continue;                                 // reads in front of the block: serial kernel
            Live l = live[owner[q]];
            const BlkInfo &r = bi[owner[q]];
            l.produced += r.n_out;
            l.nblocks++;
            if (r.bfinal) {
                const InflateJob &j = jobs[l.stream];
                InflateResult &o = res[l.stream];
                o.end_bit = r.end_bit - j.in_off * 8;
                o.out_len = l.produced;
                o.blk_out_start = l.produced;
                o.final_seen = 1;
                o.nblocks = l.nblocks;
                fast[l.stream] = 1;
            } else { l.bit = r.end_bit; next.push_back(l); }
        }
        if (c->diag.debug) {
            uint32_t bad = 0, nfl = 0;
            for (uint32_t k = 0; k < nj; k++) bad += bi[k].status != BLK_OK;
            for (uint32_t q = 0; q < ne; q++) nfl += jf[q];
            fprintf(stderr, "[lfx] batch round %u: jobs=%u scan-not-ok=%u emitted=%u cross-block=%u continuing=%zu\n", round, nj, bad,
                    ne, nfl, next.size());
        }
        live.swap(next);
    }
    return LFX_OK;
}
}  // namespace lfx

extern "C" int lfx_decode_batch_device(lfx_ctx *cc, int format, uint32_t count, const void *d_in,
                                       const uint64_t *in_off, const uint64_t *in_len, void *d_out,
                                       const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len,
                                       int32_t *status) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    c->n_ev = 0;
    c->phase("start");
    if (!count) return LFX_OK;
    std::vector<DecStream> streams(count);
    for (uint32_t i = 0; i < count; i++) streams[i] = DecStream{in_off[i], in_len[i], out_off[i], out_cap[i]};
    int rc;
    const size_t sz_streams = sizeof(DecStream) * count, sz_hdr = sizeof(DecHeader) * count;
    if ((rc = c->d_dec_blocks.reserve(sz_streams + sz_hdr + 16ull * count + 64))) return rc;
    DecStream *d_streams = (DecStream *)c->d_dec_blocks.p;
    DecHeader *d_hdrs = (DecHeader *)((uint8_t *)c->d_dec_blocks.p + sz_streams);
    uint32_t *d_crc = (uint32_t *)((uint8_t *)d_hdrs + sz_hdr);
    uint32_t *d_adler = d_crc + count;
    uint64_t *d_consumed = (uint64_t *)(d_adler + count);
    HIP_TRY(hipMemcpyAsync(d_streams, streams.data(), sz_streams, hipMemcpyHostToDevice, st));
    LAUNCH_TRY(launch_container(st, format, count, (const uint8_t *)d_in, d_streams, d_hdrs));
    std::vector<DecHeader> hdrs(count);
    if (format != LFX_DEFLATE) {
        HIP_TRY(hipMemcpyAsync(hdrs.data(), d_hdrs, sz_hdr, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        HIP_TRY(hipMemsetAsync(d_hdrs, 0, sz_hdr, st));
    }
    c->phase("headers");
    std::vector<InflateJob> jobs(count);
    for (uint32_t i = 0; i < count; i++) {
        InflateJob j{};
        j.in_off = in_off[i];
        j.in_len = in_len[i];
        j.start_bit = (format == LFX_DEFLATE ? 0 : hdrs[i].deflate_off) * 8;
        if (format != LFX_DEFLATE && hdrs[i].status != 0) j.in_len = 0;  // header failed: nothing to decode
        j.out_off = out_off[i];
        j.out_cap = out_cap[i];
        j.flags = 0;
        jobs[i] = j;
    }
    // ---- lane-parallel path first; whatever it could not take goes through the exact serial kernel
    uint64_t n_in = 0;
    for (uint32_t i = 0; i < count; i++) n_in = std::max(n_in, in_off[i] + in_len[i]);
    std::vector<uint8_t> fast;
    std::vector<InflateResult> fres;
    if (c->diag.batch_serial) { fast.assign(count, 0); fres.assign(count, InflateResult{}); }
    else if ((rc = batch_fast(c, (const uint8_t *)d_in, n_in, (uint8_t *)d_out, count, jobs, fast, fres))) return rc;
    c->phase("fast");
    std::vector<InflateJob> slow_jobs;
    std::vector<uint32_t> slow_idx;
    for (uint32_t i = 0; i < count; i++)
        if (!fast[i]) { slow_jobs.push_back(jobs[i]); slow_idx.push_back(i); }
    const uint32_t nslow = (uint32_t)slow_jobs.size();
    // d_dec_state holds the per-stream results the checksum / trailer kernels read: [count] then [nslow] scratch
    if ((rc = c->d_dec_streams.reserve(sizeof(InflateJob) * std::max<uint32_t>(nslow, 1)))) return rc;
    if ((rc = c->d_dec_state.reserve(sizeof(InflateResult) * ((size_t)count + nslow)))) return rc;
    InflateResult *d_res = (InflateResult *)c->d_dec_state.p, *d_slow = d_res + count;
    HIP_TRY(hipMemcpyAsync(d_res, fres.data(), sizeof(InflateResult) * count, hipMemcpyHostToDevice, st));
    if (nslow) {
        HIP_TRY(hipMemcpyAsync(c->d_dec_streams.p, slow_jobs.data(), sizeof(InflateJob) * nslow, hipMemcpyHostToDevice, st));
        LAUNCH_TRY(launch_inflate(st, (const uint8_t *)d_in, (uint8_t *)d_out, (const InflateJob *)c->d_dec_streams.p, d_slow, nslow));
        if (nslow == count) HIP_TRY(hipMemcpyAsync(d_res, d_slow, sizeof(InflateResult) * count, hipMemcpyDeviceToDevice, st));
        else for (uint32_t q = 0; q < nslow; q++)
            HIP_TRY(hipMemcpyAsync(d_res + slow_idx[q], d_slow + q, sizeof(InflateResult), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipStreamSynchronize(st));   // (host vectors above must outlive the copies)
    c->phase("inflate");
    if (format != LFX_DEFLATE)
        LAUNCH_TRY(launch_checksum_ranges(st, (const uint8_t *)d_out, count, (const uint64_t *)d_streams + 2,
                                          sizeof(DecStream) / 8, (const uint64_t *)c->d_dec_state.p + 1,
                                          sizeof(InflateResult) / 8, d_crc, d_adler));   // out_off / out_len fields
    LAUNCH_TRY(launch_verify_trailers(st, format, count, (const uint8_t *)d_in, d_streams, d_hdrs,
                                      (InflateResult *)c->d_dec_state.p, d_crc, d_adler, d_consumed));
    std::vector<InflateResult> res(count);
    HIP_TRY(hipMemcpyAsync(res.data(), c->d_dec_state.p, sizeof(InflateResult) * count, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    c->phase("verify");
    int worst = LFX_OK;
    for (uint32_t i = 0; i < count; i++) {
        if (out_len) out_len[i] = res[i].out_len;
        const int s = map_status(res[i].status);
        if (status) status[i] = s;
        if (s != LFX_OK && worst == LFX_OK) { worst = s; c->set_error(format_error(res[i].err, res[i].a0, res[i].a1)); }
    }
    return LFX_OK;  // per-stream results are in status[]
} LFX_ABI_CATCH

// ------------------------------------------------------------------------------------------------
// stream decoder: io::Read shaped ({deflate,zlib,gzip}::Decoder, gzip::MultiDecoder, src/non_blocking/*).
//
// Header first: the constructor pulls only what the container header needs (plus the rest of the chunk the reader
// handed over) and reports header errors (gzip.rs:941-944, zlib.rs:312-320); the body is decoded by the first
// read().  The GPU inflates whole members, so input is pulled in growing batches and a decode is ATTEMPTED whenever
// a batch is complete, the reader ends, hands over a short read, or (non-blocking mode) would block; an attempt
// that runs out of input (UnexpectedEof while the reader has not ended) simply waits for more.  Bytes pulled
// beyond the member's trailer are never decoded: they stay in the decoder (lfx_decoder_surplus) for the caller to
// hand to whatever reads next — what `into_inner()` amounts to for a reader that cannot be rewound
// (gzip.rs:987,1216-1226) — and MultiDecoder continues with them.
struct lfx_decoder {
    Ctx *c;
    int format;
    uint32_t flags;
    lfx_read_cb r;
    void *user;
    // (page-locked vectors, lfx_hostio.h: a window's H2D / D2H copies are DMA transfers straight out of / into them, and
    //  resize() does not zero-fill what a copy overwrites)
    PinVec in;                      // pulled from the reader and not yet consumed by a decoded window
    bool reader_eof = false;
    enum State { ST_HEADER, ST_BODY, ST_SERVE, ST_DONE, ST_FAILED } state = ST_HEADER;
    bool first_member = true;
    // current member
    ContainerFields hf{};
    std::vector<uint8_t> hdr_bytes;
    bool have_header = false;
    PinVec out;
    uint64_t cursor = 0, serve_limit = 0;
    int pending_status = LFX_OK;    // reported once the bytes in front of it have been served
    uint64_t consumed_total = 0;    // reader bytes that belong to finished members
    uint64_t target = 0;            // input size at which the next attempt is due
    uint64_t tried_at = 0;          // input size of the last attempt that ran out of input
    std::string err;
    // windowed body decode: the member is decoded a window of complete blocks at a time
    bool body_started = false;      // the header bytes have been dropped from `in`; in[0] holds the next block's first bit
    uint32_t bit_off = 0;           // ... at this bit of in[0]
    uint64_t member_out = 0;        // bytes of the member decoded by earlier windows
    std::vector<uint8_t> hist;      // the last (at most 32 KiB) of them: the LZ77 window the next blocks may reach into
    uint32_t run_crc = 0, run_adler = 1;   // container checksum over member_out bytes
    bool member_final = false;      // the BFINAL block has been decoded: the trailer is next
    bool more_windows = false;      // ST_SERVE: another window follows the bytes being served
    uint64_t out_cap = 0;           // output capacity of one window
    // ---- the next window decoded AHEAD (round 6): while the caller copies window k out of `out` (io::copy: 8192 bytes a
    // read), a worker thread runs window k + 1 on the GPU into `next.out`.  The input was pulled by the CALLER's thread before
    // the worker started (the read callback never runs on another thread); the worker touches `in` (read only), `hist` (read
    // only), `next` and the context (under its lock) — nothing the serving path reads.
    struct Window {
        PinVec out;
        MemberResult mr;
        uint32_t crc = 0, adler = 1;
        uint64_t n = 0;             // bytes of `in` the window was given
        int rc = LFX_OK;            // a device-level failure of the attempt
    } next;
    std::thread worker;
    bool ahead = false;             // `worker` is running (or has finished) window k + 1
};

namespace {

enum { PULL_OK = 0, PULL_EOF = 1, PULL_BLOCK = 2, PULL_ERR = 3 };
// one read() of up to `want` bytes from the inner reader
int dec_pull(lfx_decoder *d, size_t want, size_t *got) {
    *got = 0;
    if (d->reader_eof) return PULL_EOF;
    // (straight into the spare room behind `in` — page-locked, no zero fill, no second copy; the vector grows geometrically,
    //  so a reader that hands over a few bytes at a time does not pay for `want` bytes each time)
    const size_t old = d->in.size();
    d->in.resize(old + want);
    const int64_t k = d->r(d->user, d->in.data() + old, want);
    d->in.resize(old + (k > 0 ? (size_t)std::min<int64_t>(k, (int64_t)want) : 0));
    if (k == -(int64_t)LFX_E_WOULD_BLOCK) return PULL_BLOCK;
    if (k < 0) return PULL_ERR;
    if (k == 0) { d->reader_eof = true; return PULL_EOF; }
    *got = (size_t)k;
    return PULL_OK;
}

// parse the container header at the front of d->in, pulling what is missing.
// → LFX_OK (have_header set), LFX_E_WOULD_BLOCK, LFX_E_IO, or the header's own failure (err set)
int dec_header(lfx_decoder *d) {
    if (d->format == LFX_DEFLATE) { d->have_header = true; d->hf = ContainerFields{}; d->hdr_bytes.clear(); return LFX_OK; }
    for (;;) {
        ContainerFields cf;
        const DecHeader h = parse_container(d->format, d->in.data(), d->in.size(), &cf);
        if (h.status == 0) {
            d->hf = cf;                    // (a header that fails to parse leaves the previous member's in place)
            d->hdr_bytes.assign(d->in.begin(), d->in.begin() + (size_t)h.deflate_off);
            d->have_header = true;
            return LFX_OK;
        }
        if (h.status == 2 && !d->reader_eof) {          // the header may simply not be complete yet
            size_t got;
            const int pr = dec_pull(d, 1 << 16, &got);
            if (pr == PULL_BLOCK) return LFX_E_WOULD_BLOCK;
            if (pr == PULL_ERR) { d->err = "read callback failed"; return LFX_E_IO; }
            continue;
        }
        d->err = format_error(h.err, h.a0, h.a1);
        d->consumed_total += h.deflate_off;
        return map_status(h.status);
    }
}

// The body is decoded a WINDOW at a time (the reference decodes one block per read, src/deflate/decode.rs:136-164, and keeps
// 32 KiB of history, libflate_lz77/src/lib.rs:219-231): pull up to WINDOW_IN compressed bytes, decode the blocks that are
// complete in them and fit WINDOW_OUT (inflate_member, partial), serve those bytes, drop the input they used, keep the
// last 32 KiB of output as history — what is buffered never exceeds one window of input plus one window of output,
// whatever the member's size, and the first byte is served as soon as the first window is decoded.  The container
// checksum is folded window by window (CRC-32 / Adler-32 combine).  A window without one complete block (a block larger
// than the window: schedule-S1 members) doubles the window.
// WINDOW_IN_MAX bounds the doubling (a damaged stream whose block never reaches an EndOfBlock, on a reader that
// never ends, must not pull the rest of the input into memory): at the limit the window is decoded WITHOUT `partial` — the
// exact walk gives a verdict for a damaged stream; a valid block larger than the limit is refused.
// Round 6: a window's kernels are bound by per-block latency, not by its size — 33 one-MiB blocks (16 MiB of text stream) take
// 2.0 ms of kernels where the whole 256-block stream takes 2.6 (LFX_DEBUG window lines) — so after a first window of WINDOW_IN
// bytes (first bytes early) the later ones take WINDOW_IN_LATER: half as many windows for 16 MiB more of page-locked memory.
constexpr uint64_t WINDOW_IN = 16ull << 20, WINDOW_IN_LATER = 32ull << 20, WINDOW_OUT = 96ull << 20, WINDOW_IN_MAX = 4ull << 30;

// → LFX_OK when bytes or a verdict are ready (state ST_SERVE), LFX_E_WOULD_BLOCK / LFX_E_IO from the reader, or a device error
// ---- one window of the member's body in three steps: input (caller's thread) → GPU (caller's thread, or a worker thread
//      that decodes AHEAD while the caller drains the window before) → state update (dec_body)
// input up to the window size; a short read or the end of the reader also triggers an attempt.  → LFX_OK: attempt now
int dec_fill(lfx_decoder *d) {
    bool attempt = d->reader_eof;
    while (!attempt) {
        if (d->in.size() >= d->target) { attempt = true; break; }
        size_t got;
        const size_t want = std::min<uint64_t>(d->target - d->in.size(), 4u << 20);
        const int pr = dec_pull(d, want, &got);
        if (pr == PULL_ERR) { d->err = "read callback failed"; return LFX_E_IO; }
        if (pr == PULL_EOF) { attempt = true; break; }
        if (pr == PULL_BLOCK) {
            // everything the peer has sent is here: decode it if enough new bytes arrived since an attempt that found
            // no complete block (an eighth more: the retries of one long block stay linear in its size), else WouldBlock
            if (d->in.size() > d->tried_at + d->tried_at / 8) { attempt = true; break; }
            return LFX_E_WOULD_BLOCK;
        }
        // a short read hints that the reader has no more right now (pipes, sockets): worth an attempt once the
        // input has grown by a quarter since the last one (keeps the total work linear)
        if (got < want && d->in.size() >= d->tried_at + d->tried_at / 4 + 1) { attempt = true; break; }
    }
    return LFX_OK;
}

// one window on the GPU (the context's scratch is shared: one decode at a time per context): in[0, n) from bit bit_off with
// the history → W.out, W.mr, the window's checksums.  Reads d->in, d->hist, d->bit_off, d->member_out, d->reader_eof,
// d->out_cap (grows it when the tail of the member does not fit); writes only W and d->out_cap.
void dec_gpu(lfx_decoder *d, lfx_decoder::Window &W) {
    Ctx *c = d->c;
    const uint64_t trailer = d->format == LFX_GZIP ? 8 : d->format == LFX_ZLIB ? 4 : 0;
    const uint64_t n = d->in.size();
    W.n = n;
    W.crc = 0;
    W.adler = 1;
    W.rc = LFX_OK;
    MemberResult &mr = W.mr;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    for (;;) {
        mr = MemberResult();
        c->n_ev = 0;
        c->phase("start");
        int rc;
        const uint64_t H = d->hist.size();
        if ((rc = c->d_io_in.reserve(std::max<uint64_t>(n, 4)))) { W.rc = rc; return; }
        if ((rc = c->d_io_out.reserve(MAX_WINDOW + d->out_cap))) { W.rc = rc; return; }
        if ((rc = c->d_res.reserve(256))) { W.rc = rc; return; }
        uint8_t *d_out = (uint8_t *)c->d_io_out.p + MAX_WINDOW;          // the history lies right in front of it
        if (n && hipMemcpyAsync(c->d_io_in.p, d->in.data(), n, hipMemcpyHostToDevice, c->stream) != hipSuccess) { W.rc = LFX_E_DEVICE; return; }
        if (H && hipMemcpyAsync(d_out - H, d->hist.data(), H, hipMemcpyHostToDevice, c->stream) != hipSuccess) { W.rc = LFX_E_DEVICE; return; }
        // (once the reader has ended nothing more can arrive: the exact walk gives the member's verdict)
        const bool at_limit = !d->reader_eof && n >= WINDOW_IN_MAX;
        const bool partial = !d->reader_eof && !at_limit;
        const auto tw0 = std::chrono::steady_clock::now();
        rc = inflate_member(c, (const uint8_t *)c->d_io_in.p, n, 0, d_out, d->out_cap, mr, d->bit_off, ~0ull, partial, d->member_out);
        if (rc) { W.rc = rc; return; }
        const auto tw1 = std::chrono::steady_clock::now();
        if (at_limit && mr.status == LFX_E_UNEXPECTED_EOF) {
            mr.status = LFX_E_UNSUPPORTED;
            mr.msg = "a DEFLATE block exceeds the stream decoder's window limit (4 GiB of compressed bytes)";
        }
        if (mr.status == LFX_E_NOSPACE && !partial) {     // (the tail of the member does not fit one window: grow and retry)
            d->out_cap *= 2;
            continue;
        }
        const uint64_t keep = mr.out_len;                 // bytes produced (also on failure)
        if (keep && mr.status == LFX_OK && trailer) {
            const uint64_t nspans = ck_nspans(keep);
            if ((rc = c->d_ck.reserve(12 * nspans))) { W.rc = rc; return; }
            uint32_t *ck = (uint32_t *)c->d_ck.p;
            if (int e_ = launch_checksum(c->stream, d_out, keep, ck, ck + nspans, ck + 2 * nspans, (EncodeResult *)c->d_res.p,
                                         d->format == LFX_GZIP ? 1 : 2)) {
                c->set_error(hipGetErrorString((hipError_t)e_));
                W.rc = LFX_E_DEVICE;
                return;
            }
            if (hipMemcpyAsync(c->h_res, c->d_res.p, sizeof(EncodeResult), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { W.rc = LFX_E_DEVICE; return; }
        }
        W.out.resize(keep);
        if (keep && hipMemcpyAsync(W.out.data(), d_out, keep, hipMemcpyDeviceToHost, c->stream) != hipSuccess) { W.rc = LFX_E_DEVICE; return; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { W.rc = LFX_E_DEVICE; return; }
        if (keep && mr.status == LFX_OK && trailer) {
            const EncodeResult er = *(EncodeResult *)c->h_res;
            W.crc = er.crc32;
            W.adler = er.adler32;
        }
        if (c->diag.debug) {
            fprintf(stderr, "[lfx] window gpu: in=%llu out=%llu inflate_member %.3f ms, checksum + D2H %.3f ms", (unsigned long long)n,
                    (unsigned long long)keep, std::chrono::duration<double, std::milli>(tw1 - tw0).count(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count());
            if (c->timing_on)          // (the kernels' own brackets of this window; the first one holds the H2D copy too)
                for (int i = 0; i + 1 < c->n_ev; i++) {
                    float ms = 0;
                    (void)hipEventSynchronize(c->ev[i + 1]);
                    if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) fprintf(stderr, " %s=%.3f", c->ev_name[i + 1], ms);
                }
            fprintf(stderr, "\n");
        }
        return;
    }
}

// ST_SERVE, first read of a window that is worth the thread (a few MiB to copy out) and has a successor: pull the successor's
// input NOW, on the caller's thread, and let a worker decode it while the caller drains this window.  Not for non-blocking
// readers (a WouldBlock in the middle of a drain has no call to come back from).
void dec_start_ahead(lfx_decoder *d) {
    if (d->ahead || !d->more_windows || d->member_final || d->pending_status != LFX_OK || (d->flags & LFX_DEC_NONBLOCKING) ||
        !d->body_started || d->serve_limit < (4u << 20)) return;
    if (dec_fill(d) != LFX_OK) return;                 // (an I/O error is found again, and reported, by the window's own turn)
    if (d->in.empty()) return;
    try {
        d->worker = std::thread([d] {
            try { dec_gpu(d, d->next); }
            catch (const std::bad_alloc &) { d->next.rc = LFX_E_OOM; }
            catch (...) { d->next.rc = LFX_E_DEVICE; }
        });
        d->ahead = true;
    } catch (...) {
        d->ahead = false;                              // (no thread to be had: the window is decoded when its turn comes)
    }
}
// accessors and the destructor look at state the worker may be writing: let it finish first
void dec_settle(lfx_decoder *d) {
    if (d->ahead && d->worker.joinable()) d->worker.join();
}

int dec_body(lfx_decoder *d) {
    Ctx *c = d->c;
    if (!d->body_started) {
        // the header has been parsed on the host (dec_header): drop its bytes, the first block starts at bit 0
        const size_t hl = d->hdr_bytes.size();
        d->in.erase(d->in.begin(), d->in.begin() + (std::ptrdiff_t)std::min(hl, d->in.size()));
        d->consumed_total += hl;
        d->body_started = true;
        d->bit_off = 0; d->member_out = 0; d->hist.clear(); d->run_crc = 0; d->run_adler = 1; d->member_final = false;
        d->target = 0; d->tried_at = 0;
        d->out_cap = WINDOW_OUT;
    }
    // (small members: do not wait for a whole window.  1 MiB since round 6 — a reader that has less hands over a short read or
    //  its end, either of which triggers an attempt; at 64 KiB a member of 1 MiB blocks made three attempts of 0.45 ms each
    //  before its first block was complete)
    if (d->target == 0) d->target = 1 << 20;
    const uint64_t trailer = d->format == LFX_GZIP ? 8 : d->format == LFX_ZLIB ? 4 : 0;
    for (;;) {
        // ---- the member's blocks are done: the trailer (gzip.rs:1030-1042, zlib.rs:387-401)
        if (d->member_final) {
            const uint64_t tpos = d->bit_off ? 1 : 0;       // (the last block ends inside in[0]: the trailer is byte aligned)
            while (d->in.size() < tpos + trailer && !d->reader_eof) {
                size_t got;
                const int pr = dec_pull(d, (size_t)(tpos + trailer - d->in.size()), &got);
                if (pr == PULL_ERR) { d->err = "read callback failed"; return LFX_E_IO; }
                if (pr == PULL_BLOCK) return LFX_E_WOULD_BLOCK;
            }
            d->out.clear(); d->cursor = 0; d->serve_limit = 0; d->more_windows = false;
            d->pending_status = LFX_OK; d->err.clear();
            if (d->in.size() < tpos + trailer) {
                d->pending_status = LFX_E_UNEXPECTED_EOF;
                d->err = "failed to fill whole buffer";
                d->consumed_total += d->in.size();
                d->in.clear();
            } else {
                const uint8_t *t = d->in.data() + tpos;
                if (d->format == LFX_GZIP) {
                    const uint32_t crc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                    if (crc != d->run_crc) {   // gzip.rs:1035-1040 (ISIZE is read but never verified)
                        d->pending_status = LFX_E_INVALID_DATA;
                        d->err = format_error(ERR_CRC32, d->run_crc, crc);
                    }
                } else if (d->format == LFX_ZLIB) {
                    const uint32_t ad = (uint32_t)t[0] << 24 | (uint32_t)t[1] << 16 | (uint32_t)t[2] << 8 | t[3];
                    if (ad != d->run_adler) {
                        d->pending_status = LFX_E_INVALID_DATA;
                        d->err = format_error(ERR_ADLER32, d->run_adler, ad);
                    }
                }
                d->consumed_total += tpos + trailer;
                d->in.erase(d->in.begin(), d->in.begin() + (std::ptrdiff_t)(tpos + trailer));   // what is left is the surplus
            }
            d->body_started = false;
            d->state = lfx_decoder::ST_SERVE;
            return LFX_OK;
        }
        // ---- input (dec_fill) and one window on the GPU (dec_gpu) — or the window the worker thread decoded ahead while the
        //      caller was copying the one before out (dec_start_ahead)
        lfx_decoder::Window &W = d->next;
        if (d->ahead) {
            if (d->worker.joinable()) d->worker.join();      // (an accessor may have waited for it already)
            d->ahead = false;
        } else {
            const int fr = dec_fill(d);
            if (fr != LFX_OK) return fr;
            dec_gpu(d, W);
        }
        if (W.rc) return W.rc;
        const uint64_t n = W.n;
        MemberResult &mr = W.mr;
        const uint32_t w_crc = W.crc, w_adler = W.adler;
        d->out.swap(W.out);                          // (the window's bytes; W.out keeps the old buffer's room for the next one)
        bool verdict = false;
        (void)verdict;
        if (c->diag.debug)
            fprintf(stderr, "[lfx] window: n=%llu bit_off=%u eof=%d hist=%llu → status=%d out=%llu blk_out=%llu end_bit=%llu final=%d need_cap=%d target=%llu\n",
                    (unsigned long long)n, d->bit_off, (int)d->reader_eof, (unsigned long long)d->member_out, mr.status,
                    (unsigned long long)mr.out_len, (unsigned long long)mr.blk_out_start, (unsigned long long)mr.end_bit,
                    (int)mr.final_seen, (int)mr.need_cap, (unsigned long long)d->target);
        if (mr.status == LFX_OK && mr.out_len == 0 && !mr.final_seen) {
            // no complete block in this window: more input (or, for a block that does not fit, more room)
            if (mr.need_cap) { d->out_cap *= 2; continue; }
            if (d->reader_eof || n >= WINDOW_IN_MAX) {
                // (cannot happen with !partial; kept as the way out of a reader that never ends on a broken stream)
                d->pending_status = LFX_E_UNEXPECTED_EOF; d->err = "failed to fill whole buffer";
                d->out.clear(); d->cursor = 0; d->serve_limit = 0; d->more_windows = false;
                d->state = lfx_decoder::ST_SERVE;
                return LFX_OK;
            }
            d->tried_at = n;
            d->target = std::max<uint64_t>(d->target, std::max<uint64_t>(2 * n, 1 << 16));   // the next try: twice what did not suffice
            continue;
        }
        // ---- bytes (and maybe a verdict) to serve
        d->cursor = 0;
        d->pending_status = LFX_OK;
        d->err.clear();
        if (mr.status != LFX_OK) {
            // the reference hands out what completed blocks produced, then the error (decode.rs:136-164); everything
            // produced stays available through unread_decoded_data (decode.rs:68-73)
            d->serve_limit = mr.blk_out_start;
            d->pending_status = mr.status;
            d->err = mr.msg;
            d->more_windows = false;
            const uint64_t used = std::min<uint64_t>(mr.end_byte, n);
            d->consumed_total += used;
            d->in.erase(d->in.begin(), d->in.begin() + (std::ptrdiff_t)used);
            d->body_started = false;
            d->state = lfx_decoder::ST_SERVE;
            (void)verdict;
            return LFX_OK;
        }
        d->serve_limit = mr.out_len;
        if (trailer && mr.out_len) {
            d->run_crc = lfx_crc32_combine(d->run_crc, w_crc, mr.out_len);
            d->run_adler = lfx_adler32_combine(d->run_adler, w_adler, mr.out_len);
        }
        d->member_out += mr.out_len;
        // history: the last 32 KiB of (history ++ this window's output)
        if (mr.out_len >= MAX_WINDOW) d->hist.assign(d->out.end() - MAX_WINDOW, d->out.end());
        else {
            d->hist.insert(d->hist.end(), d->out.begin(), d->out.end());
            if (d->hist.size() > MAX_WINDOW) d->hist.erase(d->hist.begin(), d->hist.end() - MAX_WINDOW);
        }
        // input: everything in front of the byte that holds the next unread bit is done with
        const uint64_t used = std::min<uint64_t>(mr.end_bit / 8, n);
        d->consumed_total += used;
        d->in.erase(d->in.begin(), d->in.begin() + (std::ptrdiff_t)used);
        d->bit_off = (uint32_t)(mr.end_bit & 7);
        d->member_final = mr.final_seen;
        d->more_windows = true;                        // (the trailer check, at least, follows)
        d->target = std::max<uint64_t>(d->member_out > mr.out_len ? WINDOW_IN_LATER : WINDOW_IN, d->target > (1 << 20) ? d->target : 0);
        d->tried_at = 0;
        d->state = lfx_decoder::ST_SERVE;
        return LFX_OK;
    }
}

}  // namespace

extern "C" void lfx_decoder_free(lfx_decoder *d);
extern "C" lfx_decoder *lfx_decoder_new(lfx_ctx *cc, int format, uint32_t flags, lfx_read_cb r, void *user, int *status) try {
    if (!cc || !r || format < 0 || format > 2) { if (status) *status = cc ? LFX_E_ARG : LFX_E_DEVICE; return nullptr; }
    lfx_decoder *d = new lfx_decoder();
    d->c = reinterpret_cast<Ctx *>(cc);
    d->out = d->c->take_pin();         // (page-locked buffers of an earlier decoder of this context, when there are any)
    d->next.out = d->c->take_pin();
    d->in = d->c->take_pin();
    d->format = format;
    d->flags = flags;
    d->r = r;
    d->user = user;
    if (!(flags & LFX_DEC_NONBLOCKING)) {
        // gzip / zlib constructors read the header and can fail (gzip.rs:941-944, zlib.rs:312-320); the non-blocking
        // decoders read it lazily (src/non_blocking/gzip.rs:64-88)
        const int rc = dec_header(d);
        if (rc) {
            if (status) *status = rc;
            d->c->set_error(d->err);
            lfx_decoder_free(d);
            return nullptr;
        }
        d->state = lfx_decoder::ST_BODY;
    }
    if (status) *status = LFX_OK;
    return d;
} LFX_ABI_CATCH_NEW

extern "C" int64_t lfx_decoder_read(lfx_decoder *d, uint8_t *out, size_t cap) try {
    if (!d) return -(int64_t)LFX_E_ARG;
    if (cap == 0) return 0;  // never latches end-of-stream (gzip.rs:1025-1027, zlib.rs:383-385)
    for (;;) {
        switch (d->state) {
            case lfx_decoder::ST_DONE: return 0;
            case lfx_decoder::ST_FAILED: return 0;   // (the error was reported once, like a latched io::Error)
            case lfx_decoder::ST_SERVE: {
                if (d->cursor < d->serve_limit) {
                    if (d->cursor == 0) dec_start_ahead(d);       // (the next window on the GPU while this one is copied out)
                    const uint64_t k = std::min<uint64_t>(cap, d->serve_limit - d->cursor);
                    memcpy(out, d->out.data() + d->cursor, k);
                    d->cursor += k;
                    return (int64_t)k;
                }
                if (d->pending_status != LFX_OK) { d->state = lfx_decoder::ST_FAILED; return -(int64_t)d->pending_status; }
                if (d->more_windows) { d->state = lfx_decoder::ST_BODY; continue; }      // the next window (or the trailer)
                if (d->format == LFX_GZIP && (d->flags & LFX_DEC_MULTI)) {   // MultiDecoder::read gzip.rs:1142-1166
                    d->first_member = false;
                    d->state = lfx_decoder::ST_HEADER;
                    continue;
                }
                d->state = lfx_decoder::ST_DONE;
                return 0;
            }
            case lfx_decoder::ST_HEADER: {
                const uint64_t before = d->consumed_total;
                const int rc = dec_header(d);
                if (rc == LFX_E_WOULD_BLOCK) return -(int64_t)LFX_E_WOULD_BLOCK;
                if (rc == LFX_E_UNEXPECTED_EOF && !d->first_member) {
                    // a following member's header that ends early = clean end of the stream (gzip.rs:1150-1156);
                    // the partial header bytes were read
                    d->consumed_total = before + d->in.size();
                    d->in.clear();
                    d->err.clear();
                    d->state = lfx_decoder::ST_DONE;
                    return 0;
                }
                if (rc) { d->state = lfx_decoder::ST_FAILED; return -(int64_t)rc; }
                d->state = lfx_decoder::ST_BODY;
                continue;
            }
            case lfx_decoder::ST_BODY: {
                const int rc = dec_body(d);
                if (rc == LFX_E_WOULD_BLOCK) return -(int64_t)LFX_E_WOULD_BLOCK;
                if (rc) { d->err = d->err.empty() ? d->c->err : d->err; d->state = lfx_decoder::ST_FAILED; return -(int64_t)rc; }
                continue;
            }
        }
    }
} LFX_ABI_CATCH_NEG
extern "C" int lfx_decoder_unread(lfx_decoder *d, const uint8_t **p, size_t *n) try {
    if (!d) return LFX_E_ARG;
    // data decoded but not handed out: the rest of completed blocks + the partial block (decode.rs:68-73)
    const uint64_t start = std::min<uint64_t>(d->cursor, d->out.size());
    *p = d->out.data() + start;
    *n = d->out.size() - start;
    return LFX_OK;
} LFX_ABI_CATCH
extern "C" int lfx_decoder_surplus(lfx_decoder *d, const uint8_t **p, size_t *n) try {
    if (!d) return LFX_E_ARG;
    // input pulled from the reader that lies behind the last finished member (only meaningful between members /
    // at the end: while a member is being collected `in` holds that member's bytes)
    const bool settled = (d->state == lfx_decoder::ST_SERVE && !d->more_windows && !d->body_started) ||
                         d->state == lfx_decoder::ST_DONE || d->state == lfx_decoder::ST_FAILED;
    *p = d->in.data();
    *n = settled ? d->in.size() : 0;
    return LFX_OK;
} LFX_ABI_CATCH
extern "C" int lfx_decoder_header(lfx_decoder *d, lfx_header *h) try {
    if (!d || !h) return LFX_E_ARG;
    memset(h, 0, sizeof *h);
    if (!d->have_header) {
        if (!(d->flags & LFX_DEC_NONBLOCKING) || d->state != lfx_decoder::ST_HEADER) return LFX_E_ARG;
        const int rc = dec_header(d);          // non-blocking decoders read the header on demand (non_blocking/gzip.rs:98-113)
        if (rc) { if (rc != LFX_E_WOULD_BLOCK) d->state = lfx_decoder::ST_FAILED; return rc; }
        d->state = lfx_decoder::ST_BODY;
    }
    const ContainerFields &f = d->hf;
    const uint8_t *b = d->hdr_bytes.data();
    h->format = d->format;
    if (d->format == LFX_GZIP) {
        h->mtime = f.mtime;
        h->xfl = f.xfl;
        h->os = f.os;
        h->is_text = (f.flg & 1) != 0;
        h->is_verified = (f.flg & 2) != 0;
        if (f.flg & 4) { h->extra = b + f.extra_off; h->extra_len = (uint32_t)f.extra_len; h->has_extra = 1; }
        if (f.name_len) h->filename = (const char *)(b + f.name_off);
        if (f.comment_len) h->comment = (const char *)(b + f.comment_off);
    } else if (d->format == LFX_ZLIB) {
        h->zlib_window_size = 1u << (((uint32_t)f.cmf >> 4) + 8);   // Lz77WindowSize (zlib.rs:99-173)
        h->zlib_level = (uint32_t)f.flg >> 6;                        // CompressionLevel (zlib.rs:28-58)
    }
    return LFX_OK;
} LFX_ABI_CATCH
extern "C" uint64_t lfx_decoder_consumed(const lfx_decoder *d) { return d ? d->consumed_total : 0; }
extern "C" uint64_t lfx_decoder_buffered(const lfx_decoder *d) {
    if (!d) return 0;
    dec_settle(const_cast<lfx_decoder *>(d));      // (a window decoded ahead counts, and its buffer is not read while it grows)
    return (uint64_t)(d->in.size() + d->out.size() + d->hist.size() + d->next.out.size());
}
extern "C" const char *lfx_decoder_last_error(const lfx_decoder *d) { return d ? d->err.c_str() : "null"; }
extern "C" void lfx_decoder_free(lfx_decoder *d) {
    if (!d) return;
    dec_settle(d);
    d->c->give_pin(std::move(d->in));
    d->c->give_pin(std::move(d->out));
    d->c->give_pin(std::move(d->next.out));
    delete d;
}
