// lfx_api.cpp — C ABI (include/lfx.h) over the HIP kernels.  Host side = framing bytes, the write
// schedule planner and kernel orchestration; every byte of compression work runs on the GPU.
#include "../../include/lfx.h"
#include "../../include/lfx_testhooks.h"

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
#include "lfx_device.h"
#include "lfx_huff.h"
#include "lfx_plan.h"
#include "lfx_abi_guard.h"

using namespace lfx;

// ------------------------------------------------------------------------------------------------
// small host helpers (framing only)
static uint32_t host_crc32(const uint8_t *p, size_t n) {
    static uint32_t tab[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
            tab[i] = c;
        }
    });
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = (c >> 8) ^ tab[(c ^ p[i]) & 0xFF];
    return ~c;
}
static uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1)));
    }
    return p;
}
static uint32_t gf2_xpow8n(uint64_t nbytes) {
    uint32_t r = 0x80000000u, sq = 0x00800000u;
    while (nbytes) {
        if (nbytes & 1) r = gf2_mulmod(r, sq);
        sq = gf2_mulmod(sq, sq);
        nbytes >>= 1;
    }
    return r;
}

extern "C" uint32_t lfx_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) {
    // CRC(A||B) from CRC(A), CRC(B), |B| (the reference never combines)
    return gf2_mulmod(crc1, gf2_xpow8n(len2)) ^ crc2;
}
extern "C" uint32_t lfx_adler32_combine(uint32_t ad1, uint32_t ad2, uint64_t len2) {
    const uint32_t M = 65521u;
    uint32_t a1 = ad1 & 0xFFFF, b1 = ad1 >> 16, a2 = ad2 & 0xFFFF, b2 = ad2 >> 16;
    uint32_t rem = (uint32_t)(len2 % M);
    uint32_t a = (a1 + a2 + M - 1) % M;
    uint32_t b = (uint32_t)((b1 + b2 + (uint64_t)rem * ((a1 + M - 1) % M)) % M);
    return (b << 16) | a;
}

extern "C" void lfx_encode_opts_default(lfx_encode_opts *o) {
    memset(o, 0, sizeof *o);
    o->block_size = 1u << 20;
    o->dynamic_huffman = 1;
    o->window_size = MAX_WINDOW;
    o->max_length = MAX_LENGTH;
    o->os = 3;
}
extern "C" uint32_t lfx_version(void) { return LFX_VERSION; }

static lfx_encode_opts norm_opts(const lfx_encode_opts *o) {
    lfx_encode_opts d;
    if (o) d = *o; else lfx_encode_opts_default(&d);
    if (d.window_size > MAX_WINDOW) d.window_size = MAX_WINDOW;  // default.rs:228
    if (d.max_length > MAX_LENGTH) d.max_length = MAX_LENGTH;    // default.rs:240
    return d;
}
static int check_opts(const lfx_encode_opts &o) {
    if (o.block_size == 0) return LFX_E_ARG;  // `while len >= 0` never terminates in the reference
    if (o.max_length < 3) return LFX_E_ARG;   // default.rs:125 underflows
    return LFX_OK;
}
static PlanOpts plan_opts(int format, const lfx_encode_opts &o) {
    PlanOpts p;
    p.block_size = o.block_size;
    p.dynamic_huffman = o.dynamic_huffman != 0;
    p.no_compression = o.no_compression != 0;
    p.lz77_kind = o.lz77_kind;
    p.window_size = o.window_size;
    p.max_length = o.max_length;
    p.zlib_sync = format == LFX_ZLIB && o.zlib_flush_mode == LFX_FLUSH_SYNC;
    return p;
}

// gzip Header::write_to gzip.rs:368-389 (+flags 343-355, crc16 356-367)
static void gzip_header(const lfx_encode_opts &o, bool with_hcrc, std::vector<uint8_t> &b) {
    uint8_t flg = (uint8_t)((o.is_text ? 1 : 0) | (with_hcrc ? 2 : 0) | (o.extra ? 4 : 0) |
                            (o.filename ? 8 : 0) | (o.comment ? 16 : 0));
    // XFL is the level of the header the options hold when the encoder is made (gzip.rs:368-389): DefaultLz77Encoder →
    // Balance → Unknown(0); NoCompression → None → Unknown(0); a caller's E: Fast → Fastest(4), Best → Slowest(2)
    // (gzip.rs:84-92,684).  no_compression() resets it to Unknown WHEN IT IS CALLED (gzip.rs:703) and a later header(h)
    // replaces it again (gzip.rs:717-720): the option builders above this ABI apply both in call order and pass the result in
    // lz77_level — it is not overridden here (no_compression().header(h with Fastest) writes XFL 4).
    const uint8_t xfl = o.lz77_level == 1 + LFX_LEVEL_FAST ? 4 : o.lz77_level == 1 + LFX_LEVEL_BEST ? 2 : 0;
    const uint8_t h[10] = {31, 139, 8, flg, (uint8_t)o.mtime, (uint8_t)(o.mtime >> 8),
                           (uint8_t)(o.mtime >> 16), (uint8_t)(o.mtime >> 24), xfl, o.os};
    b.insert(b.end(), h, h + 10);
    if (o.extra) {
        b.push_back((uint8_t)o.extra_len);
        b.push_back((uint8_t)(o.extra_len >> 8));
        b.insert(b.end(), o.extra, o.extra + o.extra_len);
    }
    if (o.filename) b.insert(b.end(), o.filename, o.filename + strlen(o.filename) + 1);
    if (o.comment) b.insert(b.end(), o.comment, o.comment + strlen(o.comment) + 1);
    if (with_hcrc) {
        // the reference checksums the header serialised with is_verified = false (FLG.HCRC clear)
        std::vector<uint8_t> t;
        gzip_header(o, false, t);
        uint32_t c = host_crc32(t.data(), t.size());
        b.push_back((uint8_t)c);
        b.push_back((uint8_t)(c >> 8));
    }
}
static int container_header(int format, const lfx_encode_opts &o, std::vector<uint8_t> &b) {
    if (format == LFX_GZIP) {
        if (o.extra && o.extra_len > 0xFFFF) return LFX_E_INVALID_DATA;  // gzip.rs:490-492
        gzip_header(o, o.hcrc != 0, b);
    } else if (format == LFX_ZLIB) {
        // zlib Header::write_to zlib.rs:267-279; from_lz77 212-220; Lz77WindowSize::from_u16 132-151
        uint32_t ws = o.window_size;
        uint8_t cinfo = ws > 16384 ? 7 : ws > 8192 ? 6 : ws > 4096 ? 5 : ws > 2048 ? 4 : ws > 1024 ? 3
                        : ws > 512 ? 2 : ws > 256 ? 1 : 0;
        uint8_t level = (o.no_compression || o.lz77_kind == LFX_LZ77_NOCOMPRESSION) ? 0 : 2;
        if (o.lz77_level && !o.no_compression) level = (uint8_t)((o.lz77_level - 1) & 3);   // lz77 None/Fast/Balance/Best → 0..3 (zlib.rs:59-68)
        uint8_t cmf = (uint8_t)((cinfo << 4) | 8), flg = (uint8_t)(level << 6);
        uint32_t check = ((uint32_t)cmf << 8) + flg;
        if (check % 31 != 0) flg = (uint8_t)(flg + (31 - check % 31));
        b.push_back(cmf);
        b.push_back(flg);
    } else if (format != LFX_DEFLATE) {
        return LFX_E_ARG;
    }
    return LFX_OK;
}
extern "C" uint64_t lfx_container_header_len(int format, const lfx_encode_opts *o) {
    std::vector<uint8_t> b;
    lfx_encode_opts d = norm_opts(o);
    container_header(format, d, b);
    return b.size();
}

static void apply_schedule(Planner &pl, const lfx_schedule *s, uint64_t n) {
    if (!s || s->kind == LFX_SCHED_SINGLE) {
        pl.write(n);  // write_all of one slice == one write() (encode.rs:243 consumes everything)
    } else if (s->kind == LFX_SCHED_FIXED) {
        uint64_t w = s->fixed_write ? s->fixed_write : n;
        if (w) {
            pl.write_repeat(w, n / w);
            if (n % w) pl.write(n % w);
        }
    } else {
        uint64_t used = 0;
        for (size_t i = 0; i < s->n_writes; i++) {
            if (s->writes[i] == LFX_SCHED_FLUSH) { pl.flush(); continue; }
            uint64_t w = std::min(s->writes[i], n - used);
            pl.write(w);
            used += w;
        }
        if (used < n) pl.write(n - used);
    }
}

extern "C" uint64_t lfx_encode_bound(uint64_t n, const lfx_encode_opts *o, const lfx_schedule *s) {
    lfx_encode_opts d = norm_opts(o);
    if (check_opts(d)) return 0;
    Planner pl(plan_opts(LFX_ZLIB, d));
    apply_schedule(pl, s, n);
    Plan &p = pl.finish();
    // a dynamic block never costs more than ~9 bits per literal + its header; stored: 5 bytes per block
    uint64_t hdr = 64 + (d.extra ? d.extra_len + 2 : 0) + (d.filename ? strlen(d.filename) + 1 : 0) +
                   (d.comment ? strlen(d.comment) + 1 : 0);
    return n + n / 4 + 1024 * (uint64_t)p.blocks.size() + hdr + 64;
}

// ------------------------------------------------------------------------------------------------
// context
namespace lfx {

#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            c->set_error(std::string(#expr) + ": " + hipGetErrorString(e_));          \
            return LFX_E_DEVICE;                                                      \
        }                                                                             \
    } while (0)

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    gen++;
    size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc(&p, want) != hipSuccess) {
        if (hipMalloc(&p, bytes) != hipSuccess) { p = nullptr; return LFX_E_OOM; }
        want = bytes;
    }
    cap = want;
    return 0;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

void Diag::read() {
    auto on = [](const char *k) { return getenv(k) != nullptr; };
    debug = on("LFX_DEBUG");
    match_v1 = on("LFX_MATCH_V1");
    match_v5 = on("LFX_MATCH_V5");
    if (const char *mp = getenv("LFX_MATCH_PARTS")) match_parts = atoi(mp);
    if (const char *rc7 = getenv("LFX_R7_CAP")) r7_cap = atoi(rc7);
    no_serial = on("LFX_NO_SERIAL");
    batch_serial = on("LFX_BATCH_SERIAL");
    no_markers = on("LFX_NO_MARKERS");
    no_pieces = on("LFX_NO_PIECES");
    no_final_cand = on("LFX_NO_FINAL_CAND");
    window_chain = on("LFX_WINDOW_CHAIN");
    if (const char *fs = getenv("LFX_FREE_SHIFT")) free_shift = atoi(fs);
    if (const char *pm = getenv("LFX_POCR_MAX")) pocr_max = atoi(pm);
    if (const char *eb = getenv("LFX_ENC_BATCH_MB")) enc_batch_mb = atoi(eb);
    two_pass = on("LFX_TWO_PASS");
    hist_separate = on("LFX_HIST_SEPARATE");
    find2_exp = getenv("LFX_FIND2_EXP") ? atoi(getenv("LFX_FIND2_EXP")) : 0;
    no_pin_slots = on("LFX_NO_PIN_SLOTS");
    no_small_scan = on("LFX_NO_SMALL_SCAN");
    store_tight = on("LFX_STORE_TIGHT");
}

void Ctx::phase(const char *name) {
    if (!timing_on) return;
    if (timing_on >= 2 && timing_on <= 5) {
        // only the event in front of ONE kernel's phase and the one behind it
        static const char *const pair[4][2] = {{"lz77_resolve", "lz77_walk"}, {"find2", "blk_scan"}, {"upload", "lz77_cand"}, {"blk_emit", "lz77_copy"}};
        const char *const *pr = pair[timing_on - 2];
        if (strcmp(name, pr[0]) != 0 && strcmp(name, pr[1]) != 0) return;
    }
    if (n_ev >= 17) return;
    if (!ev[n_ev]) (void)hipEventCreate(&ev[n_ev]);
    (void)hipEventRecord(ev[n_ev], stream);
    snprintf(ev_name[n_ev], sizeof ev_name[n_ev], "%s", name);
    n_ev++;
}

}  // namespace lfx

extern "C" int lfx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" lfx_ctx *lfx_ctx_new(int device, int *status) try {
    int st = LFX_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        if (status) *status = LFX_E_DEVICE;  // fail loudly: there is no CPU path
        return nullptr;
    }
    Ctx *c = new Ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->own_stream) != hipSuccess) {
        delete c;
        if (status) *status = LFX_E_DEVICE;
        return nullptr;
    }
    c->stream = c->own_stream;
    c->diag.read();     // diagnostic environment switches are read once, here
    (void)hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device);
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_zero, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_res, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_part[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_part[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_part[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_part[3], hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        if (status) *status = LFX_E_DEVICE;
        return nullptr;
    }
    if (hipHostMalloc((void **)&c->h_res, 4096 + Ctx::PIN_ARENA, hipHostMallocDefault) != hipSuccess) {
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        if (status) *status = LFX_E_OOM;
        return nullptr;
    }
    if (status) *status = st;
    return reinterpret_cast<lfx_ctx *>(c);
} LFX_ABI_CATCH_NEW
extern "C" void lfx_ctx_free(lfx_ctx *cc) {
    if (!cc) return;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : c->all_bufs()) b->release();
    for (DevBuf &b : c->dev_pool) b.release();
    c->dev_pool.clear();
    c->pin_pool.clear();
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->h_res) (void)hipHostFree(c->h_res);
    c->hostio.release();
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_zero) (void)hipEventDestroy(c->ev_zero);
    if (c->ev_res) (void)hipEventDestroy(c->ev_res);
    for (hipEvent_t e : c->ev_part) if (e) (void)hipEventDestroy(e);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}
extern "C" const char *lfx_ctx_last_error(const lfx_ctx *cc) {
    return cc ? reinterpret_cast<const Ctx *>(cc)->err.c_str() : "no context";
}
extern "C" void lfx_ctx_set_stream(lfx_ctx *cc, void *s) {
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    c->stream = s ? (hipStream_t)s : c->own_stream;
}
extern "C" uint64_t lfx_ctx_match_fallbacks(const lfx_ctx *cc) {
    return cc ? reinterpret_cast<const Ctx *>(cc)->match_fallbacks : 0;
}
extern "C" void lfx_ctx_enable_timing(lfx_ctx *cc, int on) { reinterpret_cast<Ctx *>(cc)->timing_on = on >= 2 && on <= 6 ? on : on != 0; }
extern "C" int lfx_ctx_last_timing(lfx_ctx *cc, lfx_timing *t) try {
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    memset(t, 0, sizeof *t);
    if (c->n_ev < 2) return LFX_E_ARG;
    (void)hipEventSynchronize(c->ev[c->n_ev - 1]);
    (void)hipEventElapsedTime(&t->total_ms, c->ev[0], c->ev[c->n_ev - 1]);
    t->n_phases = c->n_ev - 1;
    for (int i = 0; i + 1 < c->n_ev && i < 16; i++) {
        (void)hipEventElapsedTime(&t->phase_ms[i], c->ev[i], c->ev[i + 1]);
        snprintf(t->phase_name[i], sizeof t->phase_name[i], "%s", c->ev_name[i + 1]);
    }
    return LFX_OK;
} LFX_ABI_CATCH

// ------------------------------------------------------------------------------------------------
// encode core
namespace lfx {

// Stage A: plan upload → match → parse → histogram → Huffman (+ checksum).  Leaves everything the
// emit stage needs in the context.
int encode_prepare(Ctx *c, const Plan &plan, const PlanOpts &po, const uint8_t *d_in, uint64_t n,
                   int ck_mode, const HostCodes *hc) {
    const bool want_checksum = ck_mode != 0;   // 1: CRC-32 (gzip), 2: Adler-32 (zlib), 3: both (a shard: the caller folds either)
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    c->n_ev = 0;
    c->phase("start");
    for (const ChunkDesc &ch : plan.chunks)
        if (ch.len >= (1ull << 32) - 4) {
            c->set_error("an LZ77 chunk of 4 GiB or more is outside the reference's domain (u32 positions, default.rs:78)");
            return LFX_E_ARG;
        }
    const uint32_t nchunks = (uint32_t)plan.chunks.size(), nblocks = (uint32_t)plan.blocks.size();
    // ---- which candidate stage: lfx_match3.hip unless it once reported a lane-order violation (or LFX_MATCH_V1 is set)
    const bool match_v1 = c->force_match_v1 || c->diag.match_v1;
    if (c->force_match_v1 && !hc) c->match_fallbacks++;      // (a silent fallback must be visible)
    // A segment is one workgroup's serial walk (plus a 32 KiB warm-up when it does not start a chunk).  Small
    // inputs are cut finer so that the GPU still fills: halve the segment length until there are >= 512 of them
    // (never below 32 Ki positions: the warm-up would dominate).
    std::vector<SegDesc> segs;
    if (!hc) {
        uint64_t seg_len = SEG_POSITIONS;
        for (;;) {
            uint64_t cnt = 0;
            for (const ChunkDesc &ch : plan.chunks)
                if (!(ch.flags & CH_LITERALS) && ch.len > 3) cnt += div_up(ch.len - 3, seg_len);
            if (cnt >= 512 || seg_len <= 32768) break;
            seg_len /= 2;
        }
        for (uint32_t ci = 0; ci < nchunks; ci++) {
            const ChunkDesc &ch = plan.chunks[ci];
            if (ch.flags & CH_LITERALS) continue;
            for (uint64_t s = 0; s + 3 < ch.len; s += seg_len)
                segs.push_back(SegDesc{ci, (uint32_t)s, (uint32_t)std::min<uint64_t>(seg_len, ch.len - s), 0});
        }
    }
    // lfx_match5: every segment keeps the final links of its positions — its warm-up included — in a region of its own
    uint64_t lnk_units = 0;
    for (SegDesc &sg : segs) {
        if (lnk_units > 0xFFFFFFFFull) { c->set_error("input too large for the link scratch"); return LFX_E_ARG; }
        sg.lnk_base = (uint32_t)lnk_units;
        lnk_units += div_up((uint64_t)sg.len + std::min<uint64_t>(sg.start, MAX_WINDOW) + 4, 64);
        lnk_units = (lnk_units + 1) & ~1ull;   // (even: lfx_match7 stores two ballot words, one per unit, as 16 bytes)
    }
    // workgroups of the parse walk: PARSE_WG_SEGS consecutive segments of one chunk each
    std::vector<ParseWg> pwgs;
    for (uint32_t ci = 0; ci < nchunks && !hc; ci++) {
        const ChunkDesc &ch = plan.chunks[ci];
        if (ch.flags & CH_LITERALS) continue;
        for (uint32_t s = 0; s < ch.n_seg; s += PARSE_WG_SEGS) pwgs.push_back(ParseWg{ci, s});
    }
    // XCD-aware launch order: workgroup i runs on XCD i mod 8, and every XCD has an L2 of its own.  A workgroup stages the
    // 32 KiB window in front of its 13 KiB of positions — the positions of its two or three left neighbours — so each XCD
    // takes one contiguous eighth of the list and finds those bytes (and its own `cd` lines) in ITS L2 instead of
    // fetching them over the fabric again (slots behind the end of an eighth are marked empty).
    if (pwgs.size() > 8) {
        const size_t nl = pwgs.size(), per = (nl + 7) / 8;
        std::vector<ParseWg> phys(per * 8);
        for (size_t i = 0; i < phys.size(); i++) {
            const size_t l = (i % 8) * per + i / 8;
            phys[i] = (i / 8 < per && l < nl && l / per == i % 8) ? pwgs[l] : ParseWg{0xFFFFFFFFu, 0u};
        }
        pwgs.swap(phys);
    }
    c->cur_nchunks = nchunks;
    c->cur_nblocks = nblocks;
    c->cur_ntiles = plan.n_tiles;
    c->cur_n = n;
    c->cur_in = d_in;
    int rc;
    if ((rc = c->d_chunks.reserve(sizeof(ChunkDesc) * std::max<size_t>(nchunks, 1)))) return rc;
    if ((rc = c->d_blocks.reserve(sizeof(BlockDesc) * std::max<size_t>(nblocks, 1)))) return rc;
    if ((rc = c->d_segs.reserve(sizeof(SegDesc) * std::max<size_t>(segs.size(), 1)))) return rc;
    if ((rc = c->d_pwgs.reserve(sizeof(ParseWg) * std::max<size_t>(pwgs.size(), 1)))) return rc;
    if (!hc && (rc = c->d_cd.reserve(2 * n + 64))) return rc;                   // candidate distances, 16 bits per position
    if (!hc && !match_v1 && (rc = c->d_glnk.reserve(264 * std::max<uint64_t>(lnk_units, 1)))) return rc;   // lfx_match7: link records (4 bytes) + ballot words (8 bytes per 64); lfx_match5: links (2 bytes)
    if (!hc && !match_v1 && (rc = c->d_ucount.reserve(4 * std::max<size_t>(segs.size(), 1)))) return rc;   // lfx_match7: unresolved positions per segment
    if (!hc && match_v1 && (rc = c->d_md.reserve(4 * std::max<uint64_t>(n, 1)))) return rc;   // first-generation kernel: (length, distance) words
    if ((rc = c->d_codes.reserve(4 * std::max<uint64_t>(plan.n_codes_cap, 1)))) return rc;
    if ((rc = c->d_ncodes.reserve(4 * std::max<size_t>(nchunks, 1)))) return rc;
    if ((rc = c->d_vis.reserve(8 * std::max<uint64_t>(plan.n_vis, 1)))) return rc;
    if ((rc = c->d_segtmp.reserve(24ull * std::max<uint32_t>(plan.n_segs, 1)))) return rc;
    if (!hc && (rc = c->d_stage.reserve(4ull * (n + 64)))) return rc;   // code words staged by the speculative parse walk
    if ((rc = c->d_chunkmap.reserve(4ull * (plan.n_tiles + plan.n_segs + 2)))) return rc;   // tile → chunk, segment → chunk
    if ((rc = c->d_hist.reserve(4ull * 320 * std::max<size_t>(nblocks, 1)))) return rc;
    if ((rc = c->d_bc.reserve(sizeof(BlockCodes) * std::max<size_t>(nblocks, 1)))) return rc;
    if ((rc = c->d_block_start.reserve(8 * std::max<size_t>(nblocks, 1)))) return rc;
    if ((rc = c->d_tile_bits.reserve(4 * std::max<uint64_t>(plan.n_tiles, 1)))) return rc;
    if ((rc = c->d_tile_start.reserve(8 * std::max<uint64_t>(plan.n_tiles, 1)))) return rc;
    const uint64_t nspans = ck_nspans(n);   // partial results of the checksum kernels
    if ((rc = c->d_ck.reserve(12 * nspans))) return rc;
    if ((rc = c->d_res.reserve(256))) return rc;
    if ((rc = c->d_small.reserve(70000))) return rc;

    // plan tables: uploaded from shadows kept on the context (they outlive the asynchronous copies: no synchronisation),
    // and not at all when the same table already sits in the same device buffer (an encode loop over equal-sized inputs)
    auto upload = [&](int slot, DevBuf &b, const void *src, size_t bytes) -> hipError_t {
        std::vector<uint8_t> &sh = c->up_shadow[slot];
        const uint64_t tag = (uint64_t)(uintptr_t)b.p ^ ((uint64_t)b.gen << 48) ^ (1ull << 63);
        if (c->up_dev[slot] == tag && sh.size() == bytes && (bytes == 0 || memcmp(sh.data(), src, bytes) == 0)) return hipSuccess;
        sh.assign((const uint8_t *)src, (const uint8_t *)src + bytes);
        c->up_dev[slot] = tag;
        return bytes ? hipMemcpyAsync(b.p, sh.data(), bytes, hipMemcpyHostToDevice, st) : hipSuccess;
    };
    HIP_TRY(upload(0, c->d_chunks, plan.chunks.data(), sizeof(ChunkDesc) * nchunks));
    HIP_TRY(upload(1, c->d_blocks, plan.blocks.data(), sizeof(BlockDesc) * nblocks));
    HIP_TRY(upload(2, c->d_segs, segs.data(), sizeof(SegDesc) * segs.size()));
    HIP_TRY(upload(3, c->d_pwgs, pwgs.data(), sizeof(ParseWg) * pwgs.size()));
    // the blocks' symbol counters, the result record and the match stage's per-segment counts are cleared by the call's first
    // kernel (three fill operations in front of the match kernel before: 22 us of a 4.6 ms step)
    const bool ucount_here = !hc && !match_v1 && !c->diag.match_v5;
    const ZeroSpan z_hist{(uint32_t *)c->d_hist.p, (uint32_t)(320 * std::max<size_t>(nblocks, 1))}, z_res{(uint32_t *)c->d_res.p, 64u},
        z_ucount{ucount_here ? (uint32_t *)c->d_ucount.p : nullptr, ucount_here ? (uint32_t)std::max<size_t>(segs.size(), 1) : 0u};
    c->phase("upload");
    uint32_t *tile_map = (uint32_t *)c->d_chunkmap.p, *seg_map = tile_map + plan.n_tiles;
    c->cur_tile_map = tile_map;
    if (int e_ = launch_chunk_maps(st, (const ChunkDesc *)c->d_chunks.p, nchunks, plan.n_tiles, plan.n_segs, tile_map, seg_map, z_hist,
                                   z_res, z_ucount)) {
        c->set_error(hipGetErrorString((hipError_t)e_));
        return LFX_E_DEVICE;
    }

#define LAUNCH_TRY(call)                                                              \
    do {                                                                              \
        int e_ = (call);                                                              \
        if (e_) {                                                                     \
            c->set_error(std::string(#call) + ": " + hipGetErrorString((hipError_t)e_)); \
            return LFX_E_DEVICE;                                                      \
        }                                                                             \
    } while (0)

    uint64_t *mdbg = nullptr;
    if (c->diag.debug) mdbg = (uint64_t *)((uint8_t *)c->d_small.p + 32768);   // per-wavefront cycle counters of workgroup 0
    bool fused_hist = false;      // the parse counted the blocks' symbols (no histogram_kernel)
    bool forked = false;          // the first half of the checksum's sweep is on the side stream already
    uint32_t emit_per = 0, emit_parts = 0;
    if (hc) {
        // the caller's own Lz77Encode produced the code words (EncodeOptions::with_lz77(E), encode.rs:59-65): they take the
        // place of the match + parse stages' output — one chunk per block, EndOfBlock included; everything from the
        // histogram on (CompressBuf::flush, encode.rs:416-425) runs as for the built-in encoders
        if (hc->n_codes) HIP_TRY(hipMemcpyAsync(c->d_codes.p, hc->codes, 4ull * hc->n_codes, hipMemcpyHostToDevice, st));
        if (nchunks) HIP_TRY(hipMemcpyAsync(c->d_ncodes.p, hc->chunk_codes, 4ull * nchunks, hipMemcpyHostToDevice, st));
        c->phase("codes_upload");
    } else {
    uint32_t *d_match_flags = (uint32_t *)((uint8_t *)c->d_res.p + offsetof(EncodeResult, match_flags));
    uint16_t *d_cd = (uint16_t *)c->d_cd.p;
    if (match_v1) {
        LAUNCH_TRY(launch_match(st, d_in, n, (const ChunkDesc *)c->d_chunks.p, (const SegDesc *)c->d_segs.p,
                                (uint32_t)segs.size(), po.window_size, po.max_length, (uint32_t *)c->d_md.p, mdbg));
        LAUNCH_TRY(launch_md_to_cd(st, (const uint32_t *)c->d_md.p, n, d_cd));
    } else {
        if (!c->diag.match_v5) {
            // lfx_match7: the candidate kernel, then the positions it leaves open (0.5 % of a text): ballot words → lists → walks.
            // (Measured, round 5: the segments in four parts, the resolver of a part on the side stream beside the next part's
            //  kernel — `LFX_MATCH_PARTS=4` — 1.10 ms against 1.15 in one piece at 256 MiB: a part of 256 workgroups ends with
            //  its slowest segment, and the compaction waits for slots behind the kernel.  The resolver with four walks in
            //  flight per lane made the overlap pointless.)
            const uint32_t ns = (uint32_t)segs.size();
            const uint32_t ncu = (uint32_t)std::max(c->n_cu, 1);
            const uint32_t want_parts = (uint32_t)std::max(c->diag.match_parts, 1);
            const uint32_t parts = ns >= 2 * ncu ? std::min<uint32_t>(std::min<uint32_t>(4, want_parts), ns / ncu) : 1;
            const SegDesc *dsegs = (const SegDesc *)c->d_segs.p;
            uint32_t *d_glnk = (uint32_t *)c->d_glnk.p;
            uint64_t *d_umask = (uint64_t *)((uint8_t *)c->d_glnk.p + 256 * std::max<uint64_t>(lnk_units, 1));
            uint32_t *d_ucount = (uint32_t *)c->d_ucount.p;
            // (d_ucount: cleared by the call's first kernel, launch_chunk_maps above)
            for (uint32_t k = 0; k < parts; k++) {
                const uint32_t s0 = (uint32_t)((uint64_t)ns * k / parts), s1 = (uint32_t)((uint64_t)ns * (k + 1) / parts);
                LAUNCH_TRY(launch_match7(st, d_in, n, (const ChunkDesc *)c->d_chunks.p, dsegs + s0, s1 - s0, po.window_size, d_cd,
                                         d_glnk, d_umask, d_match_flags, k == 0 ? mdbg : nullptr));
                if (parts == 1 && c->timing_fine()) c->phase("lz77_cand");
                hipStream_t rs = parts > 1 ? c->side_stream : st;
                if (parts > 1) {
                    HIP_TRY(hipEventRecord(c->ev_part[k], st));
                    HIP_TRY(hipStreamWaitEvent(rs, c->ev_part[k], 0));
                }
                LAUNCH_TRY(launch_resolve7(rs, (const ChunkDesc *)c->d_chunks.p, dsegs + s0, s1 - s0, po.window_size, d_cd,
                                           d_glnk, d_umask, (uint32_t *)c->d_stage.p, d_ucount + s0, (uint32_t)std::max(c->diag.r7_cap, 0)));
            }
            if (parts > 1) {
                HIP_TRY(hipEventRecord(c->ev_res, c->side_stream));
                HIP_TRY(hipStreamWaitEvent(st, c->ev_res, 0));
            }
        }
        else
            LAUNCH_TRY(launch_match5(st, d_in, n, (const ChunkDesc *)c->d_chunks.p, (const SegDesc *)c->d_segs.p,
                                     (uint32_t)segs.size(), po.window_size, d_cd, (uint16_t *)c->d_glnk.p, d_match_flags, mdbg));
    }
    if (mdbg) {
        uint64_t hv[256];
        (void)hipMemcpy(hv, mdbg, sizeof hv, hipMemcpyDeviceToHost);
        const bool v5 = !match_v1 && c->diag.match_v5, v7 = !match_v1 && !c->diag.match_v5;
        if (v7) {
            // lfx_match7: wave 0 = exchange on head, wave 1 = exchange on second, waves 2..15 = helpers
            for (int w = 0; w < 16; w++)
                fprintf(stderr, "[lfx] match7 wave%d: work=%llu barrier-wait=%llu tiles=%llu\n", w, (unsigned long long)hv[w * 8],
                        (unsigned long long)hv[w * 8 + 1], (unsigned long long)hv[w * 8 + 5]);
        } else {
            for (int w = 0; w < 16; w++)
                fprintf(stderr, "[lfx] match%s wave%d: %s=%llu %s=%llu wait=%llu tiles=%llu\n", match_v1 ? "1" : "5", w,
                        match_v1 ? "load" : "phaseA", (unsigned long long)hv[w * 8], match_v1 ? "work" : "phaseB",
                        (unsigned long long)hv[w * 8 + 1], (unsigned long long)hv[w * 8 + 2], (unsigned long long)hv[w * 8 + 5]);
        }
        if (v5) {
            uint64_t handed = 0;
            for (int w = 1; w < 16; w++) handed += hv[128 + w * 8 + 5];
            fprintf(stderr, "[lfx] match5 workgroup 0: %llu walks handed over to wave 0 of %llu positions; wave 0 waited %llu cycles for their loads\n",
                    (unsigned long long)handed, (unsigned long long)hv[5] * 960ull, (unsigned long long)hv[6]);
            for (int w = 1; w < 16; w++)
                fprintf(stderr, "[lfx] match5 wave%d loop trips: sum=%u max=%u tiles>4=%u tiles>8=%u\n", w, (unsigned)hv[w * 8 + 3],
                        (unsigned)(hv[w * 8 + 3] >> 32), (unsigned)hv[w * 8 + 4], (unsigned)(hv[w * 8 + 4] >> 32));
        }
    }
    c->phase(c->timing_fine() ? "lz77_resolve" : "lz77_match");
    if (c->diag.debug && getenv("LFX_DUMP_SEG")) {
        // diagnostics: the parse state of one segment behind the walk, behind fixseg and at the end
        const uint32_t sg = (uint32_t)atoi(getenv("LFX_DUMP_SEG"));
        for (int stage_no = 1; stage_no <= 3 && sg < plan.n_segs; stage_no++) {
            LAUNCH_TRY(launch_parse(st, d_in, n, (const ChunkDesc *)c->d_chunks.p, nchunks, plan.n_segs, (const ParseWg *)c->d_pwgs.p,
                                    (uint32_t)pwgs.size(), d_cd, po.max_length, (uint64_t *)c->d_vis.p, (uint32_t *)c->d_segtmp.p,
                                    (uint32_t *)c->d_codes.p, (uint32_t *)c->d_ncodes.p, (uint32_t *)c->d_stage.p, seg_map, stage_no % 3));
            (void)hipStreamSynchronize(st);
            uint64_t v[4];
            uint32_t t[6];
            uint16_t cdv[64];
            (void)hipMemcpy(v, (uint64_t *)c->d_vis.p + (uint64_t)sg * 64, sizeof v, hipMemcpyDeviceToHost);
            for (int q = 0; q < 6; q++) (void)hipMemcpy(&t[q], (uint32_t *)c->d_segtmp.p + (size_t)q * plan.n_segs + sg, 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(cdv, d_cd + (uint64_t)sg * PARSE_SEG, sizeof cdv, hipMemcpyDeviceToHost);
            fprintf(stderr, "[lfx] seg %u stage %d: vis %016llx %016llx %016llx %016llx exit=%u count=%u off=%u exit2=%u mpos=%u kspec=%u\n", sg,
                    stage_no, (unsigned long long)v[0], (unsigned long long)v[1], (unsigned long long)v[2], (unsigned long long)v[3], t[0],
                    t[1], t[2], t[3], t[4], t[5]);
            if (stage_no == 1) { fprintf(stderr, "[lfx]  cd:"); for (int q = 0; q < 64; q++) fprintf(stderr, " %u", cdv[q]); fprintf(stderr, "\n"); }
        }
    }
    // The blocks' symbol counts are taken by the kernel that writes the code words (parse_emit_hist_kernel): a grid of
    // nchunks x (workgroups of the longest chunk), about eight workgroups per CU in all.  A chunk list of very unequal chunks
    // (a few huge ones among thousands of small ones) would launch mostly empty workgroups: histogram_kernel counts then.
    {
        const uint32_t target = PARSE_EMIT_WG_PER_CU * (uint32_t)std::max(c->n_cu, 1);
        uint32_t per = (uint32_t)div_up(std::max<uint32_t>(plan.n_segs, 1), target);
        per = (per + PARSE_EMIT_WAVES - 1) / PARSE_EMIT_WAVES * PARSE_EMIT_WAVES;
        uint32_t max_segs = 0;
        uint64_t useful = 0;
        for (uint32_t ci = 0; ci < nchunks; ci++) {
            max_segs = std::max(max_segs, plan.chunks[ci].n_seg);
            useful += div_up(plan.chunks[ci].n_seg, per);
        }
        const uint64_t parts = div_up(max_segs, per);
        fused_hist = !c->diag.hist_separate && parts <= 65535 && (uint64_t)nchunks * parts <= 4 * useful + 4096;
        emit_per = per;
        emit_parts = (uint32_t)parts;
    }
    // (fine timing: the walk kernel in a bracket of its own — the same launches in two calls)
    for (int part = c->timing_fine() ? 1 : 0; part <= (c->timing_fine() ? 2 : 0); part++) {
        LAUNCH_TRY(launch_parse(st, d_in, n, (const ChunkDesc *)c->d_chunks.p, nchunks, plan.n_segs, (const ParseWg *)c->d_pwgs.p,
                                (uint32_t)pwgs.size(), d_cd, po.max_length, (uint64_t *)c->d_vis.p, (uint32_t *)c->d_segtmp.p,
                                (uint32_t *)c->d_codes.p, (uint32_t *)c->d_ncodes.p, (uint32_t *)c->d_stage.p, seg_map, part == 1 ? 1 : 0,
                                mdbg ? mdbg + 256 : nullptr, fused_hist ? (uint32_t *)c->d_hist.p : nullptr, emit_per, emit_parts,
                                want_checksum ? c->ev_fork : nullptr, d_match_flags, part == 2 ? 1 : 0));
        if (part == 1) c->phase("lz77_walk");
    }
    if (want_checksum) {
        // the first half of the container checksum's sweep: on the side stream from behind the walk kernel on, beside the
        // chaining kernels (one wavefront per segment and a handful of steps each: they leave most of the GPU idle); the
        // second half beside the Huffman kernel, below.  (Measured, round 6: all of it here ran into parse_emit — both
        // want the memory system — parse + Huffman 1.13 ms; all of it behind the parse no longer fits under the Huffman
        // kernel now that the histogram kernel is gone.)
        uint32_t *ck = (uint32_t *)c->d_ck.p;
        HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        LAUNCH_TRY(launch_checksum_part(c->side_stream, d_in, n, ck, ck + nspans, ck + 2 * nspans, (EncodeResult *)c->d_res.p, ck_mode, 0, 2));
        forked = true;
    }
    if (mdbg) {
        uint64_t hv[32];
        (void)hipMemcpy(hv, mdbg + 256, sizeof hv, hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; w++)
            fprintf(stderr, "[lfx] walk wave%d: fill=%llu spec=%llu resolve+chain=%llu emit=%llu cycles\n", w, (unsigned long long)hv[w * 8],
                    (unsigned long long)hv[w * 8 + 1], (unsigned long long)hv[w * 8 + 2], (unsigned long long)hv[w * 8 + 3]);
    }
    c->phase(c->timing_fine() ? "lz77_chain" : "lz77_parse");
    }   // !hc
    if (want_checksum) {
        // the container checksum reads only the input: it runs on the side stream, beside the parse's chaining kernels
        // (one wavefront per segment, a handful of steps each) and the one-workgroup-per-block Huffman kernel, which leave
        // most of the GPU idle (the fork: behind the walk kernel, launch_parse)
        uint32_t *ck = (uint32_t *)c->d_ck.p;
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        if (forked) LAUNCH_TRY(launch_checksum_part(c->side_stream, d_in, n, ck, ck + nspans, ck + 2 * nspans, (EncodeResult *)c->d_res.p, ck_mode, 1, 2));
        else LAUNCH_TRY(launch_checksum(c->side_stream, d_in, n, ck, ck + nspans, ck + 2 * nspans, (EncodeResult *)c->d_res.p, ck_mode));
        HIP_TRY(hipEventRecord(c->ev_join, c->side_stream));
    }
    // enough workgroups to fill the GPU even when there are few chunks (schedule S1: one)
    // (more workgroups per chunk were measured slower at 1024 chunks: 8 per chunk 0.23 ms against 0.20 ms for one —
    //  every workgroup ends with a global atomic per non-zero counter)
    uint32_t split = nchunks && nchunks < 1024 ? std::min<uint32_t>(1024, 2048 / nchunks + 1) : 1;
    if (!fused_hist)       // (a caller's own code words, LFX_HIST_SEPARATE, very unequal chunks)
        LAUNCH_TRY(launch_histogram(st, (const ChunkDesc *)c->d_chunks.p, nchunks, split, (const uint32_t *)c->d_codes.p,
                                    (const uint32_t *)c->d_ncodes.p, (uint32_t *)c->d_hist.p));
    c->phase("histogram");
    LAUNCH_TRY(launch_huffman(st, (const BlockDesc *)c->d_blocks.p, nblocks, (const uint32_t *)c->d_hist.p,
                              (BlockCodes *)c->d_bc.p, mdbg ? mdbg + 512 : nullptr));
    c->phase("huffman");
    if (mdbg) {
        uint64_t hv[24];
        (void)hipMemcpy(hv, mdbg + 512, sizeof hv, hipMemcpyDeviceToHost);
        static const char *nm[10] = {"lit: rank sort", "lit: depth", "lit: package-merge", "lit: widths", "lit: codes", "dist tree", "run lengths",
                                     "code-length tree", "header bits", "body size"};
        fprintf(stderr, "[lfx] huffman block 0 (cycles):");
        for (int k = 0; k < 10; k++) fprintf(stderr, " %s=%llu", nm[k], (unsigned long long)(hv[k + 1] - hv[k]));
        fprintf(stderr, "\n");
    }
    if (want_checksum) {
        HIP_TRY(hipStreamWaitEvent(st, c->ev_join, 0));
        c->phase("checksum");
    }
    return LFX_OK;
}

int encode_emit(Ctx *c, int format, bool with_trailer, uint32_t trailer_check, bool use_device_check, uint64_t total_n,
                const uint8_t *prefix, uint32_t prefix_len, uint64_t start_bit, uint8_t *d_out, uint64_t cap,
                EncodeResult *host_res, EncodeResult *async_slot = nullptr);
// Stage B: offsets → pack → framing.  `prefix` bytes are placed at the start of d_out; the DEFLATE
// bits start at bit `start_bit` of d_out (prefix may end with a partial byte).
int encode_emit(Ctx *c, int format, bool with_trailer, uint32_t trailer_check, bool use_device_check,
                uint64_t total_n, const uint8_t *prefix, uint32_t prefix_len, uint64_t start_bit,
                uint8_t *d_out, uint64_t cap, EncodeResult *host_res, EncodeResult *async_slot) {
    // async_slot (page-locked, the caller's own): the result is copied there and the call returns WITHOUT waiting — the
    // stream encoder's batch in flight; the caller synchronises and reads the slot itself.  host_res is not touched then.
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    if (((uintptr_t)d_out & 3) != 0) { c->set_error("output buffer must be 4-byte aligned"); return LFX_E_ARG; }
    const uint64_t trailer = with_trailer ? (format == LFX_GZIP ? 8 : format == LFX_ZLIB ? 4 : 0) : 0;
    if (cap < prefix_len + trailer + 8) { c->set_error("output capacity too small"); return LFX_E_NOSPACE; }
    const uint64_t cap_words = cap / 4;  // whole dwords only (kernels write dwords)
    const uint64_t cap_bits = (cap_words * 4 - trailer) * 8;
    if (c->prezero_ptr == d_out && c->prezero_bytes == cap_words * 4) {
        HIP_TRY(hipStreamWaitEvent(st, c->ev_zero, 0));      // zero-filled on the side stream, under the match kernel
    } else {
        HIP_TRY(hipMemsetAsync(d_out, 0, cap_words * 4, st));
    }
    c->prezero_ptr = nullptr;
    c->phase("memset_out");
    EncodeResult *dres = (EncodeResult *)c->d_res.p;
    LAUNCH_TRY(launch_offsets(st, (const BlockDesc *)c->d_blocks.p, c->cur_nblocks, (const BlockCodes *)c->d_bc.p,
                              start_bit, cap_bits, (uint64_t *)c->d_block_start.p, dres));
    LAUNCH_TRY(launch_pack(st, c->cur_in, c->cur_n, (const ChunkDesc *)c->d_chunks.p, c->cur_nchunks,
                           (const BlockDesc *)c->d_blocks.p, c->cur_nblocks, c->cur_ntiles,
                           (const uint32_t *)c->d_codes.p, (const uint32_t *)c->d_ncodes.p,
                           (const BlockCodes *)c->d_bc.p, (const uint64_t *)c->d_block_start.p,
                           (uint32_t *)c->d_tile_bits.p, (uint64_t *)c->d_tile_start.p, dres, 0,
                           (uint32_t *)d_out, c->cur_tile_map));
    c->phase("pack");
    if (prefix_len) {
        // (a gzip header holds an unbounded file name / comment: its own buffer, sized to fit)
        int rcp = c->d_hdr.reserve(prefix_len);
        if (rcp) return rcp;
        HIP_TRY(hipMemcpyAsync(c->d_hdr.p, prefix, prefix_len, hipMemcpyHostToDevice, st));
        LAUNCH_TRY(launch_put_bytes(st, (const uint8_t *)c->d_hdr.p, prefix_len, 0, (uint32_t *)d_out));
    }
    if (!use_device_check) {
        // combined checksum supplied by the caller (sharded encode): patch the device result
        HIP_TRY(hipMemcpyAsync((uint8_t *)dres + offsetof(EncodeResult, crc32), &trailer_check, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync((uint8_t *)dres + offsetof(EncodeResult, adler32), &trailer_check, 4, hipMemcpyHostToDevice, st));
    }
    LAUNCH_TRY(launch_trailer(st, with_trailer ? format : LFX_DEFLATE, (uint32_t)total_n, 0, dres, (uint32_t *)d_out));
    HIP_TRY(hipMemcpyAsync(async_slot ? (void *)async_slot : c->h_res, dres, sizeof(EncodeResult), hipMemcpyDeviceToHost, st));
    c->phase("frame");
    if (async_slot) return LFX_OK;
    HIP_TRY(hipStreamSynchronize(st));
    *host_res = *(EncodeResult *)c->h_res;
    if (host_res->status != 0) { c->set_error("output capacity too small"); return LFX_E_NOSPACE; }
    return LFX_OK;
}

}  // namespace lfx

// The second-generation match kernel proves its one hardware assumption at run time; a violation voids the results
// and makes the context fall back to the first-generation kernel for good.
static bool match_violation(Ctx *c, const EncodeResult &res) {
    if (!(res.match_flags & 1) || c->force_match_v1) return false;
    c->force_match_v1 = true;
    return true;
}

extern "C" int lfx_encode_device(lfx_ctx *cc, int format, const lfx_encode_opts *o, const lfx_schedule *s,
                                 const void *d_in, uint64_t n, void *d_out, uint64_t cap, uint64_t *out_len) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    lfx_encode_opts d = norm_opts(o);
    int rc = check_opts(d);
    if (rc) { c->set_error("option outside the reference's domain"); return rc; }
    std::vector<uint8_t> hdr;
    if ((rc = container_header(format, d, hdr))) { c->set_error("bad container options"); return rc; }
    PlanOpts po = plan_opts(format, d);
    Planner pl(po);
    apply_schedule(pl, s, n);
    Plan &plan = pl.finish();
    EncodeResult res{};
    // the pack kernels OR into a zero-filled output: fill it now, on the side stream, instead of between the Huffman
    // and the pack kernels (every entry point is synchronous, so nothing else is using the buffer)
    c->prezero_ptr = nullptr;
    if (((uintptr_t)d_out & 3) == 0 && cap >= 16) {
        (void)hipSetDevice(c->device);
        const uint64_t bytes = cap / 4 * 4;
        // ordered behind whatever the caller's stream still has queued on d_out (a consumer of the previous encode, an
        // allocator-reused block): fork from c->stream exactly as the checksum does
        if (hipEventRecord(c->ev_fork, c->stream) == hipSuccess && hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) == hipSuccess &&
            hipMemsetAsync(d_out, 0, bytes, c->side_stream) == hipSuccess && hipEventRecord(c->ev_zero, c->side_stream) == hipSuccess) {
            c->prezero_ptr = d_out;
            c->prezero_bytes = bytes;
        }
    }
    for (;;) {
        if ((rc = encode_prepare(c, plan, po, (const uint8_t *)d_in, n, format == LFX_GZIP ? 1 : format == LFX_ZLIB ? 2 : 0))) {
            // the fill may still be running on the caller's buffer: do not return before it has finished
            if (c->prezero_ptr) (void)hipEventSynchronize(c->ev_zero);
            c->prezero_ptr = nullptr;
            return rc;
        }
        rc = encode_emit(c, format, true, 0, true, n, hdr.data(), (uint32_t)hdr.size(), 8 * (uint64_t)hdr.size(),
                         (uint8_t *)d_out, cap, &res);
        if (match_violation(c, res)) continue;   // (never observed: see lfx_match2.hip) redo with the first-generation kernel
        break;
    }
    if (c->prezero_ptr) (void)hipEventSynchronize(c->ev_zero);   // (emit failed before it waited for the fill)
    c->prezero_ptr = nullptr;
    if (rc) return rc;
    if (out_len) *out_len = res.out_bytes;
    return LFX_OK;
} LFX_ABI_CATCH

// ---- batch encode: `count` independent streams in ONE launch set (thousands of small streams).
// Every stream is what lfx_encode_device would make of its bytes alone (same options, the schedule applied to each
// stream); the streams' chunks and blocks form one merged plan, so that the match / parse / Huffman / pack kernels see
// thousands of chunks at once instead of one launch set per 64 KiB.
extern "C" int lfx_encode_batch_device(lfx_ctx *cc, int format, const lfx_encode_opts *o, const lfx_schedule *s, uint32_t count,
                                       const void *d_in, const uint64_t *in_off, const uint64_t *in_len, void *d_out,
                                       const uint64_t *out_off, const uint64_t *out_cap, uint64_t *out_len, int32_t *status) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (format < 0 || format > 2 || (count && (!in_off || !in_len || !out_off || !out_cap))) return LFX_E_ARG;
    lfx_encode_opts d = norm_opts(o);
    int rc = check_opts(d);
    if (rc) { c->set_error("option outside the reference's domain"); return rc; }
    std::vector<uint8_t> hdr;
    if ((rc = container_header(format, d, hdr))) { c->set_error("bad container options"); return rc; }
    if (!count) return LFX_OK;
    (void)hipSetDevice(c->device);
    hipStream_t st = c->stream;
    const PlanOpts po = plan_opts(format, d);
    const uint32_t trailer = format == LFX_GZIP ? 8 : format == LFX_ZLIB ? 4 : 0;
    // ---- the merged plan: every stream planned alone, its descriptors shifted into the shared index spaces
    Plan plan;
    std::vector<BatchStream> streams(count);
    uint64_t in_extent = 0, out_lo = ~0ull, out_hi = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (out_off[i] & 3) { c->set_error("output offsets must be 4-byte aligned"); return LFX_E_ARG; }
        Planner pl(po);
        apply_schedule(pl, s, in_len[i]);
        Plan &p = pl.finish();
        const uint32_t c0 = (uint32_t)plan.chunks.size(), b0 = (uint32_t)plan.blocks.size();
        for (ChunkDesc ch : p.chunks) {
            ch.in_off += in_off[i]; ch.code_off += plan.n_codes_cap; ch.block += b0; ch.tile_base += plan.n_tiles;
            ch.vis_base += plan.n_vis; ch.seg_base += plan.n_segs;
            plan.chunks.push_back(ch);
        }
        for (BlockDesc b : p.blocks) { b.in_off += in_off[i]; b.first_chunk += c0; plan.blocks.push_back(b); }
        plan.n_codes_cap += p.n_codes_cap; plan.n_tiles += p.n_tiles; plan.n_vis += p.n_vis; plan.n_segs += p.n_segs;
        streams[i] = BatchStream{in_off[i], in_len[i], out_off[i], out_cap[i], b0, (uint32_t)p.blocks.size()};
        in_extent = std::max(in_extent, in_off[i] + in_len[i]);
        out_lo = std::min(out_lo, out_off[i]);
        out_hi = std::max(out_hi, out_off[i] + out_cap[i]);
    }
    {
        // the streams are OR-ed into their ranges (pack kernel): two streams in one range would garble each other silently
        std::vector<uint32_t> order(count);
        for (uint32_t i = 0; i < count; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return out_off[a] < out_off[b]; });
        for (uint32_t k = 0; k + 1 < count; k++)
            if (out_off[order[k]] + out_cap[order[k]] > out_off[order[k + 1]]) {
                c->set_error("the output ranges of two streams overlap");
                return LFX_E_ARG;
            }
    }
    const uint32_t nblocks = (uint32_t)plan.blocks.size();
    // per-stream scratch: descriptors, checksums, end bits, status, lengths (+ the header bytes)
    const size_t sz_streams = sizeof(BatchStream) * count;
    DevBuf &sb = c->d_dec_streams;     // (decode scratch: free during an encode)
    if ((rc = sb.reserve(sz_streams + (4 + 4 + 8 + 4 + 8) * (size_t)count + hdr.size() + 64))) return rc;
    BatchStream *d_streams = (BatchStream *)sb.p;
    uint64_t *d_end = (uint64_t *)((uint8_t *)sb.p + sz_streams);
    uint64_t *d_len = d_end + count;
    uint32_t *d_crc = (uint32_t *)(d_len + count), *d_adler = d_crc + count;
    int32_t *d_status = (int32_t *)(d_adler + count);
    uint8_t *d_hdr = (uint8_t *)(d_status + count);
    std::vector<uint64_t> h_len(count);
    std::vector<int32_t> h_status(count);
    EncodeResult res{};
    for (;;) {
        if ((rc = encode_prepare(c, plan, po, (const uint8_t *)d_in, in_extent, 0))) return rc;
        HIP_TRY(hipMemcpyAsync(d_streams, streams.data(), sz_streams, hipMemcpyHostToDevice, st));
        if (!hdr.empty()) HIP_TRY(hipMemcpyAsync(d_hdr, hdr.data(), hdr.size(), hipMemcpyHostToDevice, st));
        if (trailer)
            LAUNCH_TRY(launch_checksum_ranges(st, (const uint8_t *)d_in, count, (const uint64_t *)d_streams, sizeof(BatchStream) / 8,
                                              (const uint64_t *)d_streams + 1, sizeof(BatchStream) / 8, d_crc, d_adler));
        c->phase("checksum");
        HIP_TRY(hipMemsetAsync((uint8_t *)d_out + out_lo, 0, out_hi - out_lo, st));
        EncodeResult *dres = (EncodeResult *)c->d_res.p;
        LAUNCH_TRY(launch_offsets_batch(st, d_streams, count, (const BlockDesc *)c->d_blocks.p, (const BlockCodes *)c->d_bc.p,
                                        (uint32_t)hdr.size(), trailer, (uint64_t *)c->d_block_start.p, d_end, d_status, dres));
        LAUNCH_TRY(launch_pack(st, (const uint8_t *)d_in, in_extent, (const ChunkDesc *)c->d_chunks.p, c->cur_nchunks,
                               (const BlockDesc *)c->d_blocks.p, nblocks, c->cur_ntiles, (const uint32_t *)c->d_codes.p,
                               (const uint32_t *)c->d_ncodes.p, (const BlockCodes *)c->d_bc.p, (const uint64_t *)c->d_block_start.p,
                               (uint32_t *)c->d_tile_bits.p, (uint64_t *)c->d_tile_start.p, dres, 0, (uint32_t *)d_out, c->cur_tile_map));
        c->phase("pack");
        LAUNCH_TRY(launch_frame_batch(st, format, d_streams, count, d_hdr, (uint32_t)hdr.size(), d_end, d_crc, d_adler, dres,
                                      (uint32_t *)d_out, d_len));
        HIP_TRY(hipMemcpyAsync(c->h_res, dres, sizeof(EncodeResult), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_len.data(), d_len, 8ull * count, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_status.data(), d_status, 4ull * count, hipMemcpyDeviceToHost, st));
        c->phase("frame");
        HIP_TRY(hipStreamSynchronize(st));
        res = *(EncodeResult *)c->h_res;
        if (match_violation(c, res)) continue;
        break;
    }
    for (uint32_t i = 0; i < count; i++) {
        if (status) status[i] = res.status ? h_status[i] : LFX_OK;     // (a voided call: non-zero for exactly the streams that were too small)
        if (out_len) out_len[i] = res.status ? 0 : h_len[i];
    }
    if (res.status) {
        c->set_error("output capacity of a stream too small (status[] says which): the call is void, the output span holds no stream");
        return LFX_E_NOSPACE;
    }
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_encode_host(lfx_ctx *cc, int format, const lfx_encode_opts *o, const lfx_schedule *s,
                               const void *in, uint64_t n, void *out, uint64_t cap, uint64_t *out_len) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    uint64_t bound = lfx_encode_bound(n, o, s);
    if (bound == 0) { c->set_error("option outside the reference's domain"); return LFX_E_ARG; }
    int rc;
    if ((rc = c->d_io_in.reserve(std::max<uint64_t>(n, 4)))) return rc;
    if ((rc = c->d_io_out.reserve(bound))) return rc;
    // H2D / D2H at link rate (lfx_hostio.h): page-locked buffers (lfx_host_alloc) go to the DMA engine as they are, pageable
    // ones through page-locked slabs filled by a few threads
    if ((rc = host_to_device(c, c->d_io_in.p, in, n, c->stream))) { c->set_error("host to device copy failed"); return rc; }
    uint64_t len = 0;
    rc = lfx_encode_device(cc, format, o, s, c->d_io_in.p, n, c->d_io_out.p, bound & ~3ull, &len);
    // (a page-locked `in` was only queued for DMA: no return before the stream has passed the copy, on any path)
    if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
    if (len > cap) { c->set_error("output capacity too small"); return LFX_E_NOSPACE; }
    if ((rc = device_to_host(c, out, c->d_io_out.p, len, c->stream))) { c->set_error("device to host copy failed"); return rc; }
    if (out_len) *out_len = len;
    return LFX_OK;
} LFX_ABI_CATCH

// ---- sharded encode -------------------------------------------------------------------------
// Where lfx_encode_shard_emit will write: zero-filled on the side stream NOW, beside the match kernel of the prepare call, as
// lfx_encode_device does for its own output (the pack kernels OR into zeros; on the main stream the fill of a 130 MB shard was
// 0.07 ms in front of them).  (NULL, 0): wait for a fill in flight and forget it — the emit call is not going to come.
extern "C" int lfx_encode_shard_prezero(lfx_ctx *cc, void *d_out, uint64_t cap) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    if (c->prezero_ptr) (void)hipEventSynchronize(c->ev_zero);
    c->prezero_ptr = nullptr;
    if (!d_out || ((uintptr_t)d_out & 3) != 0 || cap < 16) return LFX_OK;
    (void)hipSetDevice(c->device);
    const uint64_t bytes = cap / 4 * 4;
    // (ordered behind whatever the caller's stream still has queued on d_out: forked from c->stream as in lfx_encode_device)
    if (hipEventRecord(c->ev_fork, c->stream) == hipSuccess && hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) == hipSuccess &&
        hipMemsetAsync(d_out, 0, bytes, c->side_stream) == hipSuccess && hipEventRecord(c->ev_zero, c->side_stream) == hipSuccess) {
        c->prezero_ptr = d_out;
        c->prezero_bytes = bytes;
    }
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_encode_shard_prepare(lfx_ctx *cc, int format, const lfx_encode_opts *o, const lfx_schedule *s,
                                        const void *d_in, uint64_t n, int is_first, int is_last,
                                        lfx_shard_info *info) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    lfx_encode_opts d = norm_opts(o);
    int rc = check_opts(d);
    if (rc) return rc;
    if (d.no_compression) { c->set_error("sharded encode of stored blocks is not supported (their size depends on the bit phase)"); return LFX_E_UNSUPPORTED; }
    PlanOpts po = plan_opts(format, d);
    Planner pl(po);
    apply_schedule(pl, s, n);
    Plan *plan;
    if (is_last) {
        plan = &pl.finish();
    } else {
        // a non-last shard must end on a block boundary with nothing buffered: close the block
        // exactly as the stream would have (block_size reached) — otherwise the shards would not
        // concatenate to what one encoder emits
        plan = &pl.plan();
        Planner probe = pl;
        Plan &fin = probe.finish();
        if (fin.blocks.size() != plan->blocks.size() + 1 || fin.blocks.back().in_len != 0) {
            c->set_error("shard boundary is not a block boundary under this write schedule");
            return LFX_E_ARG;
        }
        plan->n_codes_cap = fin.n_codes_cap;
        plan->n_tiles = plan->chunks.empty() ? 0 : plan->chunks.back().tile_base + div_up(plan->chunks.back().len + 1, PACK_TILE);
    }
    c->shard_hdr.clear();
    if (is_first && (rc = container_header(format, d, c->shard_hdr))) return rc;
    c->shard_format = format;
    c->shard_last = is_last != 0;
    hipStream_t st = c->stream;
    EncodeResult r{};
    for (;;) {
        if ((rc = encode_prepare(c, *plan, po, (const uint8_t *)d_in, n, 3))) return rc;
        // total bits at bit phase 0 (compressed blocks only → independent of the phase)
        LAUNCH_TRY(launch_offsets(st, (const BlockDesc *)c->d_blocks.p, c->cur_nblocks, (const BlockCodes *)c->d_bc.p,
                                  0, ~0ull, (uint64_t *)c->d_block_start.p, (EncodeResult *)c->d_res.p));
        HIP_TRY(hipMemcpyAsync(c->h_res, c->d_res.p, sizeof(EncodeResult), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        r = *(EncodeResult *)c->h_res;
        if (match_violation(c, r)) continue;
        break;
    }
    // For the LAST shard the figure includes the final byte alignment as seen from phase 0; its true value depends on
    // the shard's start phase and is settled by emit().  Layouts only ever sum the bits of the shards in front of a
    // shard, so the last shard's own figure is informative.
    info->total_bits = r.end_bit;
    info->n_bytes = n;
    info->crc32 = r.crc32;
    info->adler32 = r.adler32;
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_encode_shard_emit(lfx_ctx *cc, uint64_t start_bit, uint32_t combined_check, uint64_t total_n,
                                     void *d_out, uint64_t cap, uint64_t *out_len) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    // d_out[0] is byte start_bit/8 of the member; a first shard carries the container header
    const uint64_t rel_start = c->shard_hdr.empty() ? (start_bit & 7) : start_bit;
    EncodeResult res;
    int rc = encode_emit(c, c->shard_format, c->shard_last, combined_check, false, total_n,
                         c->shard_hdr.data(), (uint32_t)c->shard_hdr.size(), rel_start, (uint8_t *)d_out, cap, &res);
    if (rc) return rc;
    if (out_len) *out_len = c->shard_last ? res.out_bytes : (res.end_bit + 7) / 8;
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_shard_place_device(lfx_ctx *cc, void *d_member, uint64_t cap, const void *d_part, uint64_t part_len,
                                      uint64_t start_bit, int is_first) try {
    if (!cc) return LFX_E_DEVICE;
    Ctx *c = reinterpret_cast<Ctx *>(cc);
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    const uint64_t at = is_first ? 0 : start_bit / 8;
    if (at + part_len > cap) { c->set_error("member buffer too small"); return LFX_E_NOSPACE; }
    if (part_len == 0) return LFX_OK;
    uint8_t *dst = (uint8_t *)d_member + at;
    const uint8_t *src = (const uint8_t *)d_part;
    const bool shared = !is_first && (start_bit & 7) != 0;      // the first byte also holds the last bits of the shard in front
    if (shared) {
        LAUNCH_TRY(launch_or_byte(c->stream, dst, src));
        if (part_len > 1) HIP_TRY(hipMemcpyAsync(dst + 1, src + 1, part_len - 1, hipMemcpyDeviceToDevice, c->stream));
    } else {
        HIP_TRY(hipMemcpyAsync(dst, src, part_len, hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LFX_OK;
} LFX_ABI_CATCH

// ------------------------------------------------------------------------------------------------
// stream encoder: io::Write shaped
struct lfx_encoder {
    Ctx *c;
    int format;
    lfx_encode_opts o;
    std::string filename, comment;
    std::vector<uint8_t> extra;
    PlanOpts po;
    lfx_write_cb w;
    lfx_flush_cb f;
    void *user;
    PinVec pending;                     // input bytes not yet encoded: the planner's byte 0 is pending[0].  Page-locked (lfx_hostio.h):
                                        // the batch's H2D copy is one DMA transfer straight out of it
    PinVec h_out;                       // a batch's output lands here (page-locked) and goes to the sink from here: no copy between
    // one batch in flight on the GPU (bytes mode; enc_launch / enc_collect)
    PinVec inflight_buf, h_res;         // the batch's input bytes (DMA source until collected); result slot + the shared byte
    bool inflight = false, inflight_final = false;
    uint64_t inflight_n = 0, inflight_bound = 0;
    Plan inflight_plan;
    hipEvent_t ev_out = nullptr, ev_small = nullptr;
    Planner *pl = nullptr;              // incremental write-schedule state (chunks / blocks of `pending`)
    uint64_t total_in = 0, encoded_in = 0;
    uint32_t crc = 0, adler = 1;        // running container checksum (combined per batch)
    uint8_t carry = 0;                  // partial last byte of the bitstream so far
    uint32_t carry_bits = 0;
    bool finished = false, failed = false;
    std::string err;
    DevBuf d_in, d_out;
    // ---- codes mode (lfx_encoder_write_codes): the caller runs its own Lz77Encode, the GPU Huffman-codes what it emitted.
    // `pending` then holds the raw bytes whose checksum is still due (they are never matched).
    int mode = 0;                        // 0 undecided, 1 bytes (lfx_encoder_write), 2 codes
    struct CodeBlock { uint64_t n_codes; uint32_t type, final; };   // n_codes includes the EndOfBlock; BT_RAW = zlib sync marker
    std::vector<uint32_t> codes;         // code words of the closed blocks, then of the open one
    std::vector<CodeBlock> cblocks;      // closed blocks not yet encoded
    uint64_t closed_codes = 0, open_codes = 0;
    bool final_closed = false;
};

static int enc_emit_bytes(lfx_encoder *e, const uint8_t *p, size_t n) {
    if (!n) return LFX_OK;
    int64_t r = e->w(e->user, p, n);
    if (r != (int64_t)n) { e->err = "write callback failed"; e->failed = true; return LFX_E_IO; }
    return LFX_OK;
}

// ---- bytes mode: the closed blocks of `pending` as one GPU batch, ONE BATCH IN FLIGHT (round 6).
// The write() that closes a batch starts it — H2D straight out of the page-locked pending buffer, match … pack — and returns;
// the GPU works while the caller copies the next batch's bytes in (io::copy: 8192 at a time).  The NEXT batch's start (or
// flush / finish) collects it: result, the partial last byte (the next batch's first bits share it), the bytes to the sink.
// A failure of a batch therefore surfaces at the call that collects it — like an io::BufWriter's; the bytes and their
// order are those of the synchronous form.
struct EncCollected { uint64_t whole = 0; bool have = false; };

// start the batch; nothing is waited for.  The batch's bytes move to `inflight_buf`, the open block's bytes stay pending.
static int enc_launch(lfx_encoder *e, Plan &&plan, uint64_t n, bool final) {
    Ctx *c = e->c;
    int rc;
    if ((rc = e->d_in.reserve(std::max<uint64_t>(n, 4)))) return rc;
    const uint64_t bound = n + n / 4 + 1024 * (uint64_t)plan.blocks.size() + 128;
    if ((rc = e->d_out.reserve(bound))) return rc;
    if (e->h_res.size() < 256) e->h_res.resize(256);
    // ping-pong: appends of later writes must not move (or free) memory a DMA transfer is still reading
    e->inflight_buf.assign(e->pending.begin() + (std::ptrdiff_t)n, e->pending.end());
    e->inflight_buf.swap(e->pending);              // pending = the open block's bytes, inflight_buf = the batch (its first n bytes)
    if (n && hipMemcpyAsync(e->d_in.p, e->inflight_buf.data(), n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return LFX_E_DEVICE;
    const uint8_t prefix[1] = {e->carry};
    if ((rc = encode_prepare(c, plan, e->po, (const uint8_t *)e->d_in.p, n, e->format == LFX_GZIP ? 1 : e->format == LFX_ZLIB ? 2 : 0))) { e->err = c->err; return rc; }
    // the container trailer is written by the host (the checksum spans batches)
    rc = encode_emit(c, LFX_DEFLATE, false, 0, true, 0, prefix, e->carry_bits ? 1 : 0, e->carry_bits, (uint8_t *)e->d_out.p,
                     bound & ~3ull, nullptr, (EncodeResult *)e->h_res.data());
    if (rc) { e->err = c->err; return rc; }
    e->inflight = true;
    e->inflight_final = final;
    e->inflight_n = n;
    e->inflight_bound = bound;
    e->inflight_plan = std::move(plan);
    return LFX_OK;
}

// wait for the batch in flight, take its result and the byte the next batch shares with it, queue its bytes' way back
// (into h_out; `*done` tells when they have arrived)
static int enc_collect(lfx_encoder *e, EncCollected *out) {
    Ctx *c = e->c;
    out->have = false;
    if (!e->inflight) return LFX_OK;
    e->inflight = false;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return LFX_E_DEVICE;
    EncodeResult res = *(const EncodeResult *)e->h_res.data();
    const uint64_t n = e->inflight_n;
    int rc = LFX_OK;
    if (match_violation(c, res)) {
        // (never observed) the batch's bytes are still in d_in: once more, on the fallback kernel, synchronously
        const uint8_t prefix[1] = {e->carry};
        if ((rc = encode_prepare(c, e->inflight_plan, e->po, (const uint8_t *)e->d_in.p, n, e->format == LFX_GZIP ? 1 : e->format == LFX_ZLIB ? 2 : 0))) { e->err = c->err; return rc; }
        rc = encode_emit(c, LFX_DEFLATE, false, 0, true, 0, prefix, e->carry_bits ? 1 : 0, e->carry_bits, (uint8_t *)e->d_out.p,
                         e->inflight_bound & ~3ull, &res);
        if (rc) { e->err = c->err; return rc; }
    }
    if (res.status != 0) { c->set_error("output capacity too small"); e->err = c->err; return LFX_E_NOSPACE; }
    if (e->format == LFX_GZIP) e->crc = e->encoded_in == 0 ? res.crc32 : lfx_crc32_combine(e->crc, res.crc32, n);
    if (e->format == LFX_ZLIB) e->adler = e->encoded_in == 0 ? res.adler32 : lfx_adler32_combine(e->adler, res.adler32, n);
    e->encoded_in += n;
    const uint64_t whole = e->inflight_final ? (res.end_bit + 7) / 8 : res.end_bit / 8;
    e->h_out.resize(whole + 1);
    // the shared byte first (the next batch's launch needs it), then the bulk — which travels while that launch is prepared
    uint8_t *slot = e->h_res.data() + 128;
    if (hipMemcpyAsync(slot, (const uint8_t *)e->d_out.p + whole, 1, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipEventRecord(e->ev_small, c->stream) != hipSuccess ||
        (whole && hipMemcpyAsync(e->h_out.data(), e->d_out.p, whole, hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
        hipEventRecord(e->ev_out, c->stream) != hipSuccess || hipEventSynchronize(e->ev_small) != hipSuccess) return LFX_E_DEVICE;
    e->carry_bits = e->inflight_final ? 0 : (uint32_t)(res.end_bit & 7);
    e->carry = e->carry_bits ? *slot : 0;
    out->whole = whole;
    out->have = true;
    return LFX_OK;
}

// Start the closed blocks of `pending` (all of it when `final`, which first closes the stream) as a batch; collect the batch
// before it; with `drain` also the one just started.  The open block's bytes stay pending.
static int enc_run(lfx_encoder *e, bool final, bool drain = true) {
    Ctx *c = e->c;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    if (!e->ev_out && (hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&e->ev_small, hipEventDisableTiming) != hipSuccess)) return LFX_E_DEVICE;
    if (final) e->pl->finish();
    const uint64_t n = final ? e->pending.size() : e->pl->closed_bytes();
    Plan plan = final ? std::move(e->pl->plan()) : e->pl->take_closed();
    int rc;
    EncCollected prev;
    if ((rc = enc_collect(e, &prev))) return rc;
    const bool launched = !plan.blocks.empty();
    if (launched && (rc = enc_launch(e, std::move(plan), n, final))) return rc;
    if (prev.have) {
        // the earlier batch's bytes go to the sink while the GPU works on the batch just started
        if (hipEventSynchronize(e->ev_out) != hipSuccess) return LFX_E_DEVICE;
        if ((rc = enc_emit_bytes(e, e->h_out.data(), prev.whole))) return rc;
    }
    if (drain && launched) {
        EncCollected cur;
        if ((rc = enc_collect(e, &cur))) return rc;
        if (cur.have) {
            if (hipEventSynchronize(e->ev_out) != hipSuccess) return LFX_E_DEVICE;
            if ((rc = enc_emit_bytes(e, e->h_out.data(), cur.whole))) return rc;
        }
    }
    return LFX_OK;
}

// codes mode: encode every closed block (CompressBuf::flush from the histogram on, encode.rs:416-425)
static int enc_run_codes(lfx_encoder *e) {
    Ctx *c = e->c;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    if (e->cblocks.empty()) return LFX_OK;
    Plan plan;
    std::vector<uint32_t> chunk_codes;
    uint64_t code_cursor = 0, tile_cursor = 0;
    for (const lfx_encoder::CodeBlock &cb : e->cblocks) {
        BlockDesc b{};
        b.type = cb.type;
        b.final = cb.final;
        b.first_chunk = (uint32_t)plan.chunks.size();
        if (cb.type != BT_RAW) {
            ChunkDesc ch{};
            ch.len = cb.n_codes - 1;            // (slots = len + 1, as for a chunk of len bytes that is all literals + EndOfBlock)
            ch.code_off = code_cursor;
            ch.block = (uint32_t)plan.blocks.size();
            ch.flags = CH_LAST_IN_BLOCK;
            ch.tile_base = tile_cursor;
            code_cursor += cb.n_codes;
            tile_cursor += div_up(cb.n_codes, PACK_TILE);
            plan.chunks.push_back(ch);
            chunk_codes.push_back((uint32_t)cb.n_codes);
            b.n_chunks = 1;
        }
        plan.blocks.push_back(b);
    }
    if (e->final_closed) plan.blocks.back().align_after = 1;     // Block::finish → BitWriter::flush encode.rs:301
    plan.n_codes_cap = code_cursor;
    plan.n_tiles = tile_cursor;
    const uint64_t n = e->pending.size();
    int rc;
    if ((rc = e->d_in.reserve(std::max<uint64_t>(n, 4)))) return rc;
    // a code costs at most 15 + 5 + 15 + 13 bits
    const uint64_t bound = 6 * code_cursor + 1024 * (uint64_t)plan.blocks.size() + 128;
    if ((rc = e->d_out.reserve(bound))) return rc;
    if (n && hipMemcpyAsync(e->d_in.p, e->pending.data(), n, hipMemcpyHostToDevice, c->stream) != hipSuccess) return LFX_E_DEVICE;
    EncodeResult res{};
    uint8_t prefix[1] = {e->carry};
    const HostCodes hc{e->codes.data(), code_cursor, chunk_codes.data()};
    if ((rc = encode_prepare(c, plan, e->po, (const uint8_t *)e->d_in.p, n, e->format == LFX_GZIP ? 1 : e->format == LFX_ZLIB ? 2 : 0, &hc))) { e->err = c->err; return rc; }
    rc = encode_emit(c, LFX_DEFLATE, false, 0, true, 0, prefix, e->carry_bits ? 1 : 0, e->carry_bits, (uint8_t *)e->d_out.p, bound & ~3ull, &res);
    if (rc) { e->err = c->err; return rc; }
    if (n) {
        if (e->format == LFX_GZIP) e->crc = e->encoded_in == 0 ? res.crc32 : lfx_crc32_combine(e->crc, res.crc32, n);
        if (e->format == LFX_ZLIB) e->adler = e->encoded_in == 0 ? res.adler32 : lfx_adler32_combine(e->adler, res.adler32, n);
        e->encoded_in += n;
    }
    const uint64_t whole = e->final_closed ? (res.end_bit + 7) / 8 : res.end_bit / 8;
    e->h_out.resize(whole + 1);
    if (hipMemcpyAsync(e->h_out.data(), e->d_out.p, whole + 1, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) return LFX_E_DEVICE;
    e->carry_bits = e->final_closed ? 0 : (uint32_t)(res.end_bit & 7);
    e->carry = e->carry_bits ? e->h_out[whole] : 0;
    e->pending.clear();
    e->codes.erase(e->codes.begin(), e->codes.begin() + (std::ptrdiff_t)e->closed_codes);
    e->closed_codes = 0;
    e->cblocks.clear();
    return enc_emit_bytes(e, e->h_out.data(), whole);
}

// closed blocks of the codes mode are encoded once this many code words wait (about 8 MiB of text)
static const uint64_t ENC_BATCH_CODES = 2ull << 20;
// closed blocks (either mode) are encoded once this many raw bytes wait
static const uint64_t ENC_BATCH_BYTES_DEFAULT = 8ull << 20;
static uint64_t enc_batch_bytes(const lfx_encoder *e) { return e->c->diag.enc_batch_mb > 0 ? (uint64_t)e->c->diag.enc_batch_mb << 20 : ENC_BATCH_BYTES_DEFAULT; }

extern "C" int lfx_encoder_write_codes(lfx_encoder *e, const uint32_t *codes, size_t n_codes, const uint8_t *raw, size_t n_raw,
                                       int end_block) try {
    if (!e || e->finished || e->final_closed || end_block < 0 || end_block > 2 || (n_codes && !codes) || (n_raw && !raw)) return LFX_E_ARG;
    if (e->failed) return LFX_E_IO;
    // stored blocks never run an Lz77Encode (RawBuf, encode.rs:348-383); bytes and codes cannot be mixed on one encoder
    if (e->mode == 1 || e->po.no_compression) { e->err = "lfx_encoder_write_codes on an encoder that takes bytes"; return LFX_E_ARG; }
    for (size_t i = 0; i < n_codes; i++) {       // Code::Literal(u8) / Code::Pointer{3..=258, 1..=32768} (lib.rs:27-42)
        const uint32_t dist = codes[i] & 0xFFFFu, val = codes[i] >> 16;
        if (dist == 0 ? val > 255 : (val < 3 || val > MAX_LENGTH || dist > MAX_WINDOW)) { e->err = "code word outside Code's domain"; return LFX_E_ARG; }
    }
    e->mode = 2;
    e->codes.insert(e->codes.end(), codes, codes + n_codes);
    e->open_codes += n_codes;
    e->pending.insert(e->pending.end(), raw, raw + n_raw);
    e->total_in += n_raw;
    if (end_block) {
        e->codes.push_back(CODE_EOB);            // encode.rs:417
        e->cblocks.push_back(lfx_encoder::CodeBlock{e->open_codes + 1, e->po.dynamic_huffman ? (uint32_t)BT_DYNAMIC : (uint32_t)BT_FIXED,
                                                    end_block == 2 ? 1u : 0u});
        e->closed_codes += e->open_codes + 1;
        e->open_codes = 0;
        if (end_block == 2) e->final_closed = true;
        else if (e->closed_codes >= ENC_BATCH_CODES || e->pending.size() >= enc_batch_bytes(e)) {
            // (the byte threshold: a well-compressing Lz77Encode — 258-byte matches — closes 2 M code words only after half a
            //  gigabyte of raw bytes; the reference emits every block as it closes)
            int rc = enc_run_codes(e);
            if (rc) { e->failed = true; return rc; }
        }
    }
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" void lfx_encoder_free(lfx_encoder *e);
extern "C" lfx_encoder *lfx_encoder_new(lfx_ctx *cc, int format, const lfx_encode_opts *o, lfx_write_cb w,
                                        lfx_flush_cb f, void *user, int *status) try {
    if (!cc || !w) { if (status) *status = cc ? LFX_E_ARG : LFX_E_DEVICE; return nullptr; }
    lfx_encode_opts d = norm_opts(o);
    int rc = check_opts(d);
    if (rc) { if (status) *status = rc; return nullptr; }
    lfx_encoder *e = new lfx_encoder();
    e->c = reinterpret_cast<Ctx *>(cc);
    e->format = format;
    e->o = d;
    if (d.filename) { e->filename = d.filename; e->o.filename = e->filename.c_str(); }
    if (d.comment) { e->comment = d.comment; e->o.comment = e->comment.c_str(); }
    if (d.extra) { e->extra.assign(d.extra, d.extra + d.extra_len); e->o.extra = e->extra.data(); }
    e->po = plan_opts(format, e->o);
    e->pl = new Planner(e->po);
    // (page-locked and device buffers of an earlier encoder of this context, when there are any)
    e->pending = e->c->take_pin();
    e->h_out = e->c->take_pin();
    e->inflight_buf = e->c->take_pin();
    e->d_in = e->c->take_dev();
    e->d_out = e->c->take_dev();
    e->w = w;
    e->f = f;
    e->user = user;
    // gzip/zlib write their header immediately (gzip.rs:805, zlib.rs:578)
    std::vector<uint8_t> hdr;
    rc = container_header(format, e->o, hdr);
    if (!rc) rc = enc_emit_bytes(e, hdr.data(), hdr.size());
    if (rc) { if (status) *status = rc; lfx_encoder_free(e); return nullptr; }
    if (status) *status = LFX_OK;
    return e;
} LFX_ABI_CATCH_NEW

// closed blocks are encoded — and handed to the sink — once this much input is waiting; the open block's bytes — all the
// reference itself would be holding (encode.rs:386-426) — stay in `pending`.  8 MiB keeps what the encoder buffers within
// eight default blocks (the reference emits per block, encode.rs:277-286) at a third of the one-shot call's throughput:
// a batch is one GPU pass with ~0.4 ms of fixed latency.

extern "C" int64_t lfx_encoder_write(lfx_encoder *e, const uint8_t *p, size_t n) try {
    if (!e || e->finished) return -(int64_t)LFX_E_ARG;
    if (e->failed) return -(int64_t)LFX_E_IO;
    if (e->mode == 2) { e->err = "lfx_encoder_write on an encoder that takes code words"; return -(int64_t)LFX_E_ARG; }
    e->mode = 1;
    e->pending.insert(e->pending.end(), p, p + n);
    e->pl->write(n);
    e->total_in += n;
    if (e->pl->closed_bytes() >= enc_batch_bytes(e) || (e->po.no_compression && e->pl->closed_blocks() >= 1024)) {
        // the batch is started; the next one — or flush / finish — collects it.  Stored blocks (a batch is 64 MiB of them,
        // copied, not compressed) leave as they are closed.
        int rc = enc_run(e, false, /*drain=*/e->po.no_compression);
        if (rc) { e->failed = true; return -(int64_t)rc; }
    }
    return (int64_t)n;  // encode.rs:243: always consumes everything
} LFX_ABI_CATCH_NEG

extern "C" int lfx_encoder_flush(lfx_encoder *e) try {
    if (!e || e->finished) return LFX_E_ARG;
    if (e->failed) return LFX_E_IO;
    int rc;
    if (e->mode == 2) {
        // the caller closed the block with E::flush()'s codes (end_block = 1) — Encoder::flush is Block::flush (encode.rs:245-248)
        if (e->open_codes) { e->err = "close the open block first (end_block = 1)"; return LFX_E_ARG; }
        if (e->po.zlib_sync) { e->cblocks.push_back(lfx_encoder::CodeBlock{0, (uint32_t)BT_RAW, 0u}); }   // zlib_sync_flush encode.rs:225-234
        rc = enc_run_codes(e);
        if (rc) { e->failed = true; return rc; }
        if (e->f && e->f(e->user) != 0) { e->failed = true; e->err = "flush callback failed"; return LFX_E_IO; }
        return LFX_OK;
    }
    e->pl->flush();
    rc = enc_run(e, false);  // io::Write::flush pushes everything to the inner writer
    if (rc) { e->failed = true; return rc; }
    if (e->f && e->f(e->user) != 0) { e->failed = true; e->err = "flush callback failed"; return LFX_E_IO; }
    return LFX_OK;
} LFX_ABI_CATCH

extern "C" int lfx_encoder_finish(lfx_encoder *e) try {
    if (!e || e->finished) return LFX_E_ARG;
    if (e->failed) return LFX_E_IO;
    if (e->mode == 2 && !e->final_closed) { e->err = "close the final block first (end_block = 2)"; return LFX_E_ARG; }
    e->finished = true;
    int rc = e->mode == 2 ? enc_run_codes(e) : enc_run(e, true);
    if (rc) return rc;
    uint8_t t[8];
    size_t nt = 0;
    if (e->format == LFX_GZIP) {  // Trailer::write_to gzip.rs:114-121; ISIZE wraps (gzip.rs:893)
        uint32_t c = e->total_in ? e->crc : 0, sz = (uint32_t)e->total_in;
        t[0] = c; t[1] = c >> 8; t[2] = c >> 16; t[3] = c >> 24;
        t[4] = sz; t[5] = sz >> 8; t[6] = sz >> 16; t[7] = sz >> 24;
        nt = 8;
    } else if (e->format == LFX_ZLIB) {  // zlib.rs:630-639
        uint32_t a = e->total_in ? e->adler : 1;
        t[0] = a >> 24; t[1] = a >> 16; t[2] = a >> 8; t[3] = a;
        nt = 4;
    }
    rc = enc_emit_bytes(e, t, nt);
    if (rc) return rc;
    if (e->f && e->f(e->user) != 0) { e->err = "flush callback failed"; return LFX_E_IO; }
    return LFX_OK;
} LFX_ABI_CATCH
extern "C" const char *lfx_encoder_last_error(const lfx_encoder *e) { return e ? e->err.c_str() : "null"; }
extern "C" void lfx_encoder_free(lfx_encoder *e) {
    if (!e) return;
    (void)hipSetDevice(e->c->device);
    e->c->give_dev(e->d_in);
    e->c->give_dev(e->d_out);
    (void)hipStreamSynchronize(e->c->stream);      // (freed without finish, or after a failed launch: the DMA engine may still read the buffers)
    if (e->ev_out) (void)hipEventDestroy(e->ev_out);
    if (e->ev_small) (void)hipEventDestroy(e->ev_small);
    e->c->give_pin(std::move(e->pending));
    e->c->give_pin(std::move(e->h_out));
    e->c->give_pin(std::move(e->inflight_buf));
    delete e->pl;
    delete e;
}

// ------------------------------------------------------------------------------------------------
// Lz77Encode plug-in
struct lfx_lz77 {
    Ctx *c;
    uint32_t window, max_len;
    std::vector<uint8_t> buf;
    DevBuf d_in;
    std::vector<uint32_t> host_codes;
};
extern "C" lfx_lz77 *lfx_lz77_new(lfx_ctx *cc, uint32_t window_size, uint32_t max_length, int *status) try {
    if (!cc) { if (status) *status = LFX_E_DEVICE; return nullptr; }
    if (max_length < 3) { if (status) *status = LFX_E_ARG; return nullptr; }
    lfx_lz77 *z = new lfx_lz77();
    z->c = reinterpret_cast<Ctx *>(cc);
    z->window = std::min(window_size, MAX_WINDOW);
    z->max_len = std::min(max_length, MAX_LENGTH);
    if (status) *status = LFX_OK;
    return z;
} LFX_ABI_CATCH_NEW
extern "C" int lfx_lz77_flush(lfx_lz77 *z, lfx_sink_cb sink, void *user) try {
    // DefaultLz77Encoder::flush default.rs:69-109 — one chunk through match + parse
    Ctx *c = z->c;
    std::lock_guard<std::recursive_mutex> lock(c->mu);
    (void)hipSetDevice(c->device);
    const uint64_t n = z->buf.size();
    if (n == 0) return LFX_OK;
    PlanOpts po;
    po.window_size = z->window;
    po.max_length = z->max_len;
    Plan plan;
    ChunkDesc ch{};
    ch.len = n;
    ch.n_seg = (uint32_t)div_up(n, PARSE_SEG);
    plan.n_segs = ch.n_seg;
    plan.n_vis = (uint64_t)ch.n_seg * 64;
    plan.chunks.push_back(ch);
    BlockDesc bd{};
    bd.type = BT_DYNAMIC;
    bd.n_chunks = 1;
    plan.blocks.push_back(bd);
    plan.n_codes_cap = n + 1;
    plan.n_tiles = div_up(n + 1, PACK_TILE);
    int rc;
    if ((rc = z->d_in.reserve(std::max<uint64_t>(n, 4)))) return rc;
    HIP_TRY(hipMemcpyAsync(z->d_in.p, z->buf.data(), n, hipMemcpyHostToDevice, c->stream));
    uint32_t nc = 0;
    for (;;) {
        if ((rc = encode_prepare(c, plan, po, (const uint8_t *)z->d_in.p, n, 0))) return rc;
        EncodeResult r{};
        HIP_TRY(hipMemcpy(&r, c->d_res.p, sizeof r, hipMemcpyDeviceToHost));
        if (match_violation(c, r)) continue;
        HIP_TRY(hipMemcpy(&nc, c->d_ncodes.p, 4, hipMemcpyDeviceToHost));
        break;
    }
    z->host_codes.resize(nc);
    if (nc) HIP_TRY(hipMemcpy(z->host_codes.data(), c->d_codes.p, 4ull * nc, hipMemcpyDeviceToHost));
    z->buf.clear();  // default.rs:108
    if (sink && nc) sink(user, z->host_codes.data(), nc);
    return LFX_OK;
} LFX_ABI_CATCH
extern "C" int lfx_lz77_encode(lfx_lz77 *z, const uint8_t *buf, size_t len, lfx_sink_cb sink, void *user) try {
    z->buf.insert(z->buf.end(), buf, buf + len);                     // default.rs:64
    if (z->buf.size() >= (size_t)z->window * 8) return lfx_lz77_flush(z, sink, user);  // default.rs:65-67
    return LFX_OK;
} LFX_ABI_CATCH
extern "C" uint32_t lfx_lz77_window_size(const lfx_lz77 *z) { return z->window; }
extern "C" int lfx_lz77_compression_level(const lfx_lz77 *) { return LFX_LEVEL_BALANCE; }
extern "C" void lfx_lz77_free(lfx_lz77 *z) {
    if (!z) return;
    (void)hipSetDevice(z->c->device);
    z->d_in.release();
    delete z;
}

// ------------------------------------------------------------------------------------------------
// test hooks (include/lfx_testhooks.h) for the CPU test-suite: run the SAME host/device-shared code on the host.
// Not a product path (no compression work can be reached through them).
extern "C" int lfx_debug_huff_block(const uint32_t *hist320, uint32_t type, uint32_t *lit288, uint32_t *dist32,
                                    uint32_t *hdr160, uint32_t *hdr_bits, uint64_t *body_bits) try {
    static HuffScratch S;
    static BlockCodes bc;
    memset(&bc, 0, sizeof bc);
    huff_block_build(hist320, type, &bc, S, 0, 1);
    memcpy(lit288, bc.lit, sizeof bc.lit);
    memcpy(dist32, bc.dist, sizeof bc.dist);
    memcpy(hdr160, bc.hdr, sizeof bc.hdr);
    *hdr_bits = bc.hdr_bits;
    *body_bits = bc.body_bits;
    return 0;
} LFX_ABI_CATCH
extern "C" int lfx_debug_plan(int format, const lfx_encode_opts *o, const lfx_schedule *s, uint64_t n,
                              uint64_t *chunk_out /* in_off,len,block,flags per chunk */, size_t max_chunks,
                              size_t *n_chunks, uint64_t *block_out /* type,final,first,n,in_off,in_len */,
                              size_t max_blocks, size_t *n_blocks) try {
    lfx_encode_opts d = norm_opts(o);
    if (check_opts(d)) return LFX_E_ARG;
    Planner pl(plan_opts(format, d));
    apply_schedule(pl, s, n);
    Plan &p = pl.finish();
    *n_chunks = p.chunks.size();
    *n_blocks = p.blocks.size();
    for (size_t i = 0; i < p.chunks.size() && i < max_chunks; i++) {
        chunk_out[4 * i] = p.chunks[i].in_off; chunk_out[4 * i + 1] = p.chunks[i].len;
        chunk_out[4 * i + 2] = p.chunks[i].block; chunk_out[4 * i + 3] = p.chunks[i].flags;
    }
    for (size_t i = 0; i < p.blocks.size() && i < max_blocks; i++) {
        block_out[6 * i] = p.blocks[i].type; block_out[6 * i + 1] = p.blocks[i].final;
        block_out[6 * i + 2] = p.blocks[i].first_chunk; block_out[6 * i + 3] = p.blocks[i].n_chunks;
        block_out[6 * i + 4] = p.blocks[i].in_off; block_out[6 * i + 5] = p.blocks[i].in_len;
    }
    return 0;
} LFX_ABI_CATCH
// the same plan, but collected the way the stream encoder does: closed blocks are taken out after every
// `take_every`-th event and the remainder is rebased; the pieces are stitched back together here
extern "C" int lfx_debug_plan_incremental(int format, const lfx_encode_opts *o, const lfx_schedule *s, uint64_t n,
                                          uint32_t take_every, uint64_t *chunk_out, size_t max_chunks, size_t *n_chunks,
                                          uint64_t *block_out, size_t max_blocks, size_t *n_blocks) try {
    lfx_encode_opts d = norm_opts(o);
    if (check_opts(d) || !s || s->kind != LFX_SCHED_LIST) return LFX_E_ARG;
    Planner pl(plan_opts(format, d));
    std::vector<ChunkDesc> chunks;
    std::vector<BlockDesc> blocks;
    uint64_t byte0 = 0;
    auto stitch = [&](Plan &p, uint64_t bytes) {
        const uint32_t b0 = (uint32_t)blocks.size(), c0 = (uint32_t)chunks.size();
        for (ChunkDesc c : p.chunks) { c.in_off += byte0; c.block += b0; chunks.push_back(c); }
        for (BlockDesc b : p.blocks) { b.in_off += byte0; b.first_chunk += c0; blocks.push_back(b); }
        byte0 += bytes;
    };
    uint64_t used = 0;
    for (size_t i = 0; i < s->n_writes; i++) {
        if (s->writes[i] == LFX_SCHED_FLUSH) pl.flush();
        else { const uint64_t w = std::min(s->writes[i], n - used); pl.write(w); used += w; }
        if (take_every && (i + 1) % take_every == 0) {
            const uint64_t cut = pl.closed_bytes();
            Plan p = pl.take_closed();
            stitch(p, cut);
        }
    }
    if (used < n) pl.write(n - used);
    Plan &fin = pl.finish();
    stitch(fin, 0);
    *n_chunks = chunks.size();
    *n_blocks = blocks.size();
    for (size_t i = 0; i < chunks.size() && i < max_chunks; i++) {
        chunk_out[4 * i] = chunks[i].in_off; chunk_out[4 * i + 1] = chunks[i].len;
        chunk_out[4 * i + 2] = chunks[i].block; chunk_out[4 * i + 3] = chunks[i].flags;
    }
    for (size_t i = 0; i < blocks.size() && i < max_blocks; i++) {
        block_out[6 * i] = blocks[i].type; block_out[6 * i + 1] = blocks[i].final;
        block_out[6 * i + 2] = blocks[i].first_chunk; block_out[6 * i + 3] = blocks[i].n_chunks;
        block_out[6 * i + 4] = blocks[i].in_off; block_out[6 * i + 5] = blocks[i].in_len;
    }
    return 0;
} LFX_ABI_CATCH
extern "C" void lfx_debug_symbols(uint32_t length, uint32_t distance, uint32_t *out6) {
    out6[0] = len_symbol(length, out6[1], out6[2]);
    out6[3] = dist_symbol(distance, out6[4], out6[5]);
}
