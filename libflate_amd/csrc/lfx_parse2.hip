// lfx_parse2.hip — the greedy LZ77 walk i += length / i += 1 of DefaultLz77Encoder::flush
// (libflate_lz77/src/default.rs:76-107) with LAZY match lengths, for gfx950.
//
// The candidate stage (lfx_match3.hip) leaves cd[p] = distance to the most recent earlier occurrence of p's 3-byte
// prefix (0 = none).  longest_common_prefix (default.rs:122-129) is only ever needed at the positions the walk visits —
// about a quarter of all positions on text — so it is computed HERE, by the walk, from the input bytes.
//
// The walk is a serial chain per chunk; it runs speculatively in parallel at two levels:
//   * a LANE owns a group of PARSE_GROUP (52) consecutive positions and walks it from its first position; greedy
//     parses that start at different positions merge after a few steps, so almost all of each group's speculative
//     walk is the true walk;
//   * a WAVEFRONT owns a segment of 64 groups (3328 positions).  Inside the wavefront the true entry of group L is the
//     exit of group L-1: every lane re-walks from there until it lands on a position its speculative walk visited (from
//     there on both coincide), then the chain of entries is verified lane by lane (one ballot when every lane merged —
//     the common case on text) and repaired serially where it does not hold (long matches that jump over whole
//     groups).  Segments are chained the same way by the three small kernels behind the walk (parse_fixseg / parse_fix
//     / parse_emit): they re-walk from a segment's true entry to the merge point, through global memory.
// A workgroup is PARSE_WG_SEGS (12 since round 6; 4 before) wavefronts = 39936 consecutive positions of one chunk; it stages the bytes
// [first position - 32 KiB, last position + 258 + slack) and the cd[] values of its positions in LDS (153 KB: one
// workgroup per CU, three wavefronts per SIMD), so that a walk step — cd[p] together with 16 bytes at p+3, then 16 bytes at p+3-d per compare step — is two LDS
// round trips and no HBM access.
//
// A code word is derivable from the visited set alone: the step at p is the distance to the next visited position, and
// it is a match exactly when cd[p] != 0 (a match is at least 3 long, a literal step is 1).  The walk STAGES its code
// words (stage[], a segment's codes compacted from its first position on); everything behind the point where the true
// walk merges into a segment's speculative one is final, so parse_emit copies it and rebuilds only the few codes in
// front of the merge point from the visit bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "lfx_common.h"
#include "lfx_device.h"
#include "lfx_huff.h"

namespace lfx {

namespace p2 {

constexpr uint32_t U = PARSE_GROUP;                 // positions per lane
constexpr uint32_t WAVES = PARSE_WG_SEGS;           // segments per workgroup
constexpr uint32_t THREADS = 64 * WAVES;
constexpr uint32_t WG_POS = WAVES * PARSE_SEG;      // 13312 positions per workgroup
constexpr uint32_t TAIL = 288;                      // bytes staged behind the last position: 3 + 240 + 16 + 4 (dword reads) + alignment
constexpr uint32_t WIN_BYTES = MAX_WINDOW + WG_POS + TAIL + 16;  // (+ the byte phase of the window start on the 16-byte grid)
constexpr uint32_t OFF_CD = (WIN_BYTES + 15) & ~15u;
constexpr uint32_t CD_BYTES = (2 * WG_POS + 16 + 15) & ~15u;    // (+ up to seven entries of alignment shift)
constexpr uint32_t LDS_BYTES = OFF_CD + CD_BYTES;
constexpr uint32_t WQ = (OFF_CD / 16 + THREADS - 1) / THREADS;   // 16-byte units of the window a lane copies (6)
constexpr uint32_t CQ = (CD_BYTES / 16 + THREADS - 1) / THREADS; // ... of the candidates (7)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) u32x4 *gptr_x4;
static_assert(U <= 64 && (U / 4) * 4 == U && ((U / 4) & 1) == 1, "a group is an odd number of dwords: bank-conflict-free lane stride");
static_assert(PARSE_WG_SEGS > 4 ? LDS_BYTES <= 160 * 1024 : 2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU (one, of more wavefronts, in the experiments)");

struct ByteSrcG {
    gptr_u32 w;
    uint64_t shift, nbytes;
    __device__ __forceinline__ uint32_t load4(uint64_t off) const {   // bytes [off, off+4), zeros past the buffer
        const uint64_t a = off + shift, idx = a >> 2;
        const uint32_t sh = (uint32_t)a & 3;
        const uint64_t last = (nbytes + shift + 3) >> 2;
        const uint32_t w0 = idx < last ? w[idx] : 0;
        const uint32_t w1 = (sh != 0 && idx + 1 < last) ? w[idx + 1] : 0;
        return __builtin_amdgcn_alignbyte(w1, w0, sh);
    }
    __device__ __forceinline__ uint32_t load1(uint64_t off) const {
        const uint64_t a = off + shift;
        return (w[a >> 2] >> (((uint32_t)a & 3) * 8)) & 0xFF;
    }
};
__device__ __forceinline__ ByteSrcG make_src(const uint8_t *p, uint64_t n) {
    ByteSrcG s;
    const uint64_t a = (uint64_t)p;
    s.w = (gptr_u32)(a & ~3ull);
    s.shift = a & 3;
    s.nbytes = n;
    return s;
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    const uint32_t lane = __lane_id();
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}
__device__ __forceinline__ uint64_t bits_from(uint32_t r) { return ~0ull << r; }          // r < 64

// 8 bytes at LDS byte offset off (aligned dword reads: an unaligned ds_read costs 21-26 cycles, tools/exp/lds_tput)
__device__ __forceinline__ uint64_t lds8(const uint32_t *win32, uint32_t off) {
    const uint32_t i = off >> 2;
    const uint32_t w0 = win32[i], w1 = win32[i + 1], w2 = win32[i + 2];
    return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, off & 3) | (uint64_t)__builtin_amdgcn_alignbyte(w2, w1, off & 3) << 32;
}

// what one wavefront needs to take walk steps out of LDS
struct WalkCtx {
    const uint32_t *win32;     // staged bytes; byte offset of chunk position p is p - w0
    const uint16_t *cd16;      // staged candidates; index of chunk position p is p - c0
    uint32_t w0, c0;
    uint32_t n;                // chunk length
    uint32_t max_len;
};

// 16 bytes at LDS byte offset off as four dwords (five aligned dword reads)
struct B16 { uint32_t v[4]; };
__device__ __forceinline__ B16 lds16(const uint32_t *win32, uint32_t off) {
    const uint32_t i = off >> 2, sh = off & 3;
    const uint32_t w0 = win32[i], w1 = win32[i + 1], w2 = win32[i + 2], w3 = win32[i + 3], w4 = win32[i + 4];
    B16 r;
    r.v[0] = __builtin_amdgcn_alignbyte(w1, w0, sh);
    r.v[1] = __builtin_amdgcn_alignbyte(w2, w1, sh);
    r.v[2] = __builtin_amdgcn_alignbyte(w3, w2, sh);
    r.v[3] = __builtin_amdgcn_alignbyte(w4, w3, sh);
    return r;
}
// number of equal leading bytes of two 16-byte strings (16 when all are equal)
__device__ __forceinline__ uint32_t eq_bytes16(const B16 &a, const B16 &b) {
    const uint32_t x0 = a.v[0] ^ b.v[0], x1 = a.v[1] ^ b.v[1], x2 = a.v[2] ^ b.v[2], x3 = a.v[3] ^ b.v[3];
    const uint64_t lo = (uint64_t)x0 | (uint64_t)x1 << 32, hi = (uint64_t)x2 | (uint64_t)x3 << 32;
    const uint32_t nlo = lo ? (uint32_t)__builtin_ctzll(lo) >> 3 : 8u;
    const uint32_t nhi = hi ? (uint32_t)__builtin_ctzll(hi) >> 3 : 8u;
    return lo ? nlo : 8u + nhi;
}

// One walk step at position pos (default.rs:79-103): 1 for a literal, the match length otherwise.  `act` lanes take
// part; the others return 0 and read harmless addresses.  The compare runs 16 bytes per iteration and the wavefront
// iterates until its longest match is settled: with 8 bytes per iteration nearly every step of a wavefront held some
// lane with a match of 12 or more, and every lane paid its extra iterations (1400 cycles per step, measured; the LDS
// round trips of one iteration are the same for 8 and for 16 bytes).  The position's own bytes do not depend on the
// candidate: they are loaded together with it.
__device__ __forceinline__ uint32_t walk_step(const WalkCtx &w, uint32_t pos, bool act) {
    uint32_t oa = act ? pos + 3 - w.w0 : 0u;                        // (idle lanes read offset 0)
    // loads that do not depend on the candidate
    const uint32_t d = act ? w.cd16[pos - w.c0] : 0u;
    B16 a = lds16(w.win32, oa);
    uint32_t lim = w.n - (pos + 3);                                 // default.rs:125 (bounded by the end of the chunk)
    lim = lim > w.max_len - 3 ? w.max_len - 3 : lim;
    lim = d ? lim : 0u;
    uint32_t ob = oa - d;
    uint32_t l = 0;
    bool cmp = lim != 0;
    // the first 16 bytes (nearly every match of a text ends inside them) ...
    if (__ballot(cmp)) {
        const B16 b = lds16(w.win32, ob);
        const uint32_t adv = eq_bytes16(a, b);
        l += cmp ? adv : 0u;
        cmp = cmp && adv == 16 && l < lim;
        oa += cmp ? 16u : 0u;
        ob += cmp ? 16u : 0u;
    }
    // ... then 64 bytes per round trip (round 4: on low-entropy data most matches are 258 long, and sixteen dependent rounds of
    // 16 bytes made the walk of BASELINE cfg5 three times as slow as a text's; a lane whose compare has ended re-reads its
    // last offsets)
    for (int round = 0; round < 4 && __ballot(cmp); ++round) {       // (255 bytes at most)
        B16 aa[4], bb[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) { aa[k] = lds16(w.win32, oa + 16 * k); bb[k] = lds16(w.win32, ob + 16 * k); }
        uint32_t adv = eq_bytes16(aa[0], bb[0]);
        adv += adv == 16 ? eq_bytes16(aa[1], bb[1]) : 0u;
        adv += adv == 32 ? eq_bytes16(aa[2], bb[2]) : 0u;
        adv += adv == 48 ? eq_bytes16(aa[3], bb[3]) : 0u;
        l += cmp ? adv : 0u;
        cmp = cmp && adv == 64 && l < lim;
        oa += cmp ? 64u : 0u;
        ob += cmp ? 64u : 0u;
    }
    l = l > lim ? lim : l;
    return act ? (d ? 3u + l : 1u) : 0u;
}

// One walk step at a UNIFORM position by the whole wavefront: lane k compares bytes [4k, 4k + 4) behind the prefix, so one
// LDS round trip settles a match of any length (the repair of the entry chain walks ONE group at a time: with a single
// lane at work a 258-byte match cost six dependent round trips — that chain is what made the walk of low-entropy data,
// BASELINE cfg5, three times as slow as a text's).
__device__ __forceinline__ uint32_t walk_step_coop_d(const WalkCtx &w, uint32_t pos, uint32_t d, uint32_t lane) {
    if (d == 0) return 1;
    uint32_t lim = w.n - (pos + 3);                                 // default.rs:125
    lim = lim > w.max_len - 3 ? w.max_len - 3 : lim;
    const uint32_t off = 4 * lane;
    uint32_t x = 0;
    if (off < lim) {
        const uint32_t oa = pos + 3 - w.w0 + off, ob = oa - d;
        const uint32_t a0 = w.win32[oa >> 2], a1 = w.win32[(oa >> 2) + 1], b0 = w.win32[ob >> 2], b1 = w.win32[(ob >> 2) + 1];
        x = __builtin_amdgcn_alignbyte(a1, a0, oa & 3) ^ __builtin_amdgcn_alignbyte(b1, b0, ob & 3);
    }
    const uint64_t mis = __ballot(x != 0);
    uint32_t l = lim;
    if (mis) {
        const uint32_t fl = (uint32_t)__builtin_ctzll(mis);
        const uint32_t cand = off + ((uint32_t)__builtin_ctz(x | 0x80000000u) >> 3);
        l = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)fl);
        l = l > lim ? lim : l;
    }
    return 3 + l;
}
__device__ __forceinline__ uint32_t walk_step_coop(const WalkCtx &w, uint32_t pos, uint32_t lane) {
    return walk_step_coop_d(w, pos, w.cd16[pos - w.c0], lane);    // (one address: a broadcast read)
}
// Few lanes left at work (round 6): a wavefront's loop over walk steps costs its ~100 instructions per trip whatever the number of lanes
// that still walk — on data with long literal stretches among long matches (BASELINE cfg5: a record seen for the first time is 64
// literals, everything around it 258-byte matches) one lane took 52 trips and 63 lanes one or two.  When at most COOP_K lanes are left,
// the wavefront finishes their groups one after the other, all lanes on ONE group: a stretch of literals is one ballot over the
// group's candidates, a match one LDS round trip.
// This way of finishing is a serial chain of LDS round trips — slower than the common loop for lanes that have a few steps left (a
// text: any COOP_K made the walk slower, 8 by 20 %) — so it is taken only when a lane left has FAR to go (COOP_FAR positions, or
// COOP_CODES code words in the emit loop), and the test is made every fourth trip (measured: cfg5 parse 6.23 -> 4.7 ms per GiB).
#ifndef LFX_WALK_COOP_K
#define LFX_WALK_COOP_K 16
#endif
#ifndef LFX_WALK_COOP_FAR
#define LFX_WALK_COOP_FAR 16
#endif
constexpr uint32_t COOP_K = LFX_WALK_COOP_K, COOP_FAR = LFX_WALK_COOP_FAR, COOP_CODES = 12;
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t j) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)j) |
           (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)j) << 32;
}
// the speculative walk of ONE group [ja, jstop) from position p on (everything uniform) → the positions visited, the exit
__device__ __forceinline__ void spec_one(const WalkCtx &w, uint32_t lane, uint32_t p, uint32_t ja, uint32_t jstop, uint64_t &m_out,
                                         uint32_t &x_out) {
    const uint32_t width = jstop - ja;                                       // <= U < 64
    const uint32_t D = lane < width ? w.cd16[ja + lane - w.c0] : 0u;         // the group's candidates, a lane each
    const uint64_t nzall = __ballot(D != 0);
    uint64_t mj = 0;
    for (uint32_t guard = 0; guard <= U && p < jstop; ++guard) {
        const uint32_t r = p - ja;
        const uint64_t nz = nzall & bits_from(r);
        if (!nz) { mj |= bits_from(r) & ((1ull << width) - 1); p = jstop; break; }     // literals up to the group's end
        const uint32_t f = (uint32_t)__builtin_ctzll(nz);
        mj |= bits_from(r) & ((2ull << f) - 1);                               // literals [r, f) and the match at f
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)D, (int)f);
        p = ja + f + walk_step_coop_d(w, ja + f, d, lane);
    }
    m_out = mj; x_out = p;
}
// resolve() for ONE group with everything uniform (the group's first position a, its end, its speculative mask and exit):
// the true walk enters at `in` → the visited mask and the exit for that entry
__device__ __forceinline__ void resolve_one(const WalkCtx &w, uint32_t lane, uint32_t in, uint32_t a, uint32_t stop, uint64_t mask,
                                            uint32_t exit_spec, uint64_t &m_out, uint32_t &x_out, uint64_t walked = 0) {
    uint32_t pos = in;
    for (uint32_t guard = 0; guard <= U + 1; ++guard) {
        if (pos >= stop) { m_out = walked; x_out = pos; return; }
        if ((mask >> (pos - a)) & 1) { m_out = walked | (mask & bits_from(pos - a)); x_out = exit_spec; return; }
        walked |= 1ull << (pos - a);
        pos += walk_step_coop(w, pos, lane);
    }
    m_out = walked; x_out = pos;
}

// The true walk enters this lane's group at `in`.  Re-walk from there until it lands on a position the speculative walk
// visited (mask, exit) — from there on both coincide — or leaves the group.  → the visited mask and the exit for THAT
// entry.  `act` lanes take part (wave-uniform loops inside).
template <bool COOP>
__device__ __forceinline__ void resolve(const WalkCtx &w, bool act, uint32_t in, uint32_t a, uint32_t stop, uint64_t mask,
                                        uint32_t exit_spec, uint64_t &m_out, uint32_t &x_out) {
    uint64_t walked = 0;
    uint32_t pos = in;
    bool run = act;
    uint64_t m = 0;
    uint32_t x = in;                                   // (passed over: nothing visited, the walk goes on where it was)
    for (uint32_t guard = 0; guard <= U + 1; ++guard) {               // (a group holds U positions: U + 1 rounds settle it)
        if (run) {
            if (pos >= stop) { m = walked; x = pos; run = false; }
            else if ((mask >> (pos - a)) & 1) { m = walked | (mask & bits_from(pos - a)); x = exit_spec; run = false; }
        }
        const uint64_t left = __ballot(run);
        if (!left) break;
        if (COOP && (guard & 3) == 3 && (uint32_t)__popcll(left) <= COOP_K && __ballot(run && stop - pos >= COOP_FAR)) {
            for (uint64_t todo = left; todo; todo &= todo - 1) {
                const uint32_t j = (uint32_t)__builtin_ctzll(todo);
                uint64_t mj = 0;
                uint32_t xj = 0;
                resolve_one(w, __lane_id(), (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)j), (uint32_t)__builtin_amdgcn_readlane((int)a, (int)j),
                            (uint32_t)__builtin_amdgcn_readlane((int)stop, (int)j), readlane64(mask, j),
                            (uint32_t)__builtin_amdgcn_readlane((int)exit_spec, (int)j), mj, xj, readlane64(walked, j));
                if (__lane_id() == j) { m = mj; x = xj; }
            }
            break;
        }
        const uint32_t st = walk_step(w, run ? pos : a, run);
        walked |= run ? 1ull << (pos - a) : 0ull;
        pos += st;
    }
    if (act) { m_out = m; x_out = x; }
}



}  // namespace p2

// What a workgroup's slot of the launch order asks for (uniform).
struct WalkSlot {
    uint32_t kind;            // 0: empty slot / nothing to do, 1: behind the chunk's last walked position (pass through), 2: walk
    ParseWg wg;
    ChunkDesc ch;
    uint32_t n, end, g0, w0, sh, csh;
    p2::gptr_x4 gw, gc;       // first 16-byte unit of the bytes / of the candidates
    uint32_t lastw, lastc;    // last unit that may be read
};
__device__ __forceinline__ WalkSlot walk_slot(uint32_t slot, uint32_t nwgs, const uint8_t *in, uint64_t in_bytes,
                                              const ChunkDesc *chunks, const ParseWg *wgs, const uint16_t *cd) {
    using namespace p2;
    WalkSlot q;
    q.kind = 0;
    if (slot >= nwgs) return q;
    q.wg = wgs[slot];
    if (q.wg.chunk == 0xFFFFFFFFu) return q;          // an empty slot of the XCD-aware order (lfx_api.cpp)
    q.ch = chunks[q.wg.chunk];
    if (q.ch.flags & CH_LITERALS) return q;           // no walk: every byte is a literal
    q.n = (uint32_t)q.ch.len;
    q.end = (q.n > 3 ? q.n : 3) - 3;                  // default.rs:75
    q.g0 = q.wg.seg0 * PARSE_SEG;                     // first position of the workgroup
    if (q.g0 >= q.end) { q.kind = 1; return q; }
    q.kind = 2;
    // The bytes [w0, hi) and the candidates of [g0, g0 + WG_POS) are copied on their own 16-byte grid (position p sits at LDS
    // byte p - w0 + sh, its candidate at entry p - g0 + csh): aligned 16-byte loads and LDS stores.
    q.w0 = q.g0 > MAX_WINDOW ? q.g0 - MAX_WINDOW : 0u;
    const uint64_t abs0 = (uint64_t)(in + q.ch.in_off) + q.w0;
    q.sh = (uint32_t)abs0 & 15;
    const uint64_t e0 = q.ch.in_off + q.g0;           // cd[] index of position g0
    q.csh = (uint32_t)e0 & 7;
    q.gw = (gptr_x4)(abs0 & ~15ull);
    const uint64_t left = (uint64_t)in + in_bytes - (abs0 & ~15ull);                     // bytes up to the end of the input
    const uint32_t hi = min(q.g0 + WG_POS + TAIL, q.n);                                  // (the compare never reads past the chunk)
    const uint64_t want = ((uint64_t)(hi - q.w0) + q.sh + 15) >> 4, have = (left + 15) >> 4;
    q.lastw = (uint32_t)(want < have ? want : have) - 1;                                 // (>= 1 unit here)
    q.gc = (gptr_x4)(cd + (e0 - q.csh));
    q.lastc = ((min(WG_POS, q.end - q.g0) + q.csh + 7) >> 3) - 1;
    return q;
}

// K1: speculative walk, in-wavefront chaining, staging.  seg_exit / seg_count / vis / stage describe the segment as
// walked from its FIRST position.
// Round 6 (second half): the kernel is PERSISTENT — one workgroup per CU takes every gridDim.x-th slot of the launch order
// (gridDim.x a multiple of 8: a slot stays on the XCD the order gave it to) — and the 16-byte loads of the NEXT slot's bytes
// and candidates are in flight while the wavefronts walk the current one (52 registers per lane); they reach LDS behind the
// barrier that ends the walk.  Before, a workgroup's first 9 K of its 39 K cycles were the fill, with nothing else on the CU.
template <bool DBG, bool COOP>
__device__ __forceinline__ void walk_segment(const WalkSlot &q, uint32_t max_len, uint64_t *__restrict__ vis,
                                             uint32_t *__restrict__ seg_exit, uint32_t *__restrict__ seg_count,
                                             uint32_t *__restrict__ stage, const uint32_t *win32, const uint32_t *cd32,
                                             uint32_t lane, uint32_t wave, uint64_t *stamps) {
    using namespace p2;
    const ParseWg &wg = q.wg;
    const ChunkDesc &ch = q.ch;
    const uint32_t n = q.n, end = q.end, g0 = q.g0, w0 = q.w0, sh = q.sh, csh = q.csh;
    const uint64_t t1 = DBG ? clock64() : 0;
    const uint32_t sidx = wg.seg0 + wave;
    if (sidx >= ch.n_seg) return;
    const uint32_t s0 = sidx * PARSE_SEG;
    const uint32_t seg = ch.seg_base + sidx;
    uint64_t *vw = vis + ch.vis_base + (uint64_t)sidx * 64;
    if (s0 >= end) {
        vw[lane] = 0;
        if (lane == 0) { seg_exit[seg] = s0; seg_count[seg] = 0; }
        return;
    }
    const uint32_t s1 = min(s0 + PARSE_SEG, end);
    WalkCtx w;
    w.win32 = win32;
    w.cd16 = (const uint16_t *)cd32;
    w.w0 = w0 - sh;          // (LDS byte offset of position p: p - w.w0)
    w.c0 = g0 - csh;
    w.n = n;
    w.max_len = max_len;

    const uint32_t a = s0 + lane * U;                 // first position of this lane's group
    const uint32_t stop = min(a + U, s1);
    const bool have = a < s1;                         // the group holds positions (lanes behind the end stay idle)
    const uint32_t nact = (s1 - s0 + U - 1) / U;      // groups with positions: lanes [0, nact)
    // ---- speculative walk of every group from its first position
    uint64_t mask = 0;
    uint32_t pos = a;
    for (uint32_t guard = 0; guard <= U; ++guard) {
        const bool run = have && pos < stop;
        const uint64_t left = __ballot(run);
        if (!left) break;
        if (COOP && (guard & 3) == 3 && (uint32_t)__popcll(left) <= COOP_K && __ballot(run && stop - pos >= COOP_FAR)) {
            for (uint64_t todo = left; todo; todo &= todo - 1) {
                const uint32_t j = (uint32_t)__builtin_ctzll(todo);
                const uint32_t ja = s0 + j * U;
                uint64_t mj = 0;
                uint32_t xj = 0;
                spec_one(w, lane, (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)j), ja, min(ja + U, s1), mj, xj);
                if (lane == j) { mask |= mj; pos = xj; }
            }
            break;
        }
        const uint32_t st = walk_step(w, run ? pos : a, run);
        mask |= run ? 1ull << (pos - a) : 0ull;
        pos += st;
    }
    const uint32_t exit_spec = pos;
    const uint64_t t2 = DBG ? clock64() : 0;
    // ---- the walk enters group L where it left group L-1: assume that is the speculative exit (it is when the walk
    //      through L-1 merged), resolve every group for that entry ...
    uint32_t used_in = __shfl_up(exit_spec, 1);
    if (lane == 0) used_in = a;
    uint64_t m_fin = mask;
    uint32_t x_fin = exit_spec;
    resolve<COOP>(w, have && lane != 0, used_in, a, stop, mask, exit_spec, m_fin, x_fin);
    // ---- ... and verify the chain: lane 0's entry is the segment's first position by definition, so every lane in front
    //      of the first one whose assumed entry is not its predecessor's exit is exact.  Repair that lane with its true
    //      entry and look again (strictly increasing: at most nact rounds; none on text, one per jumped-over group on
    //      runs).
    for (uint32_t guard = 0; guard < 64; ++guard) {
        const uint32_t prev_x = __shfl_up(x_fin, 1);
        const uint64_t bad = __ballot(lane != 0 && lane < nact && used_in != prev_x);
        if (!bad) break;
        const uint32_t j = (uint32_t)__builtin_ctzll(bad);
        const uint32_t tin = __builtin_amdgcn_readlane(x_fin, j - 1);
        const bool me = lane == j;
        used_in = me ? tin : used_in;
        {
            // lane j's group, walked by the whole wavefront (its parameters made uniform)
            const uint32_t ja = s0 + j * U, jstop = min(ja + U, s1);
            const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mask, (int)j);
            const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mask >> 32), (int)j);
            const uint32_t jx = (uint32_t)__builtin_amdgcn_readlane((int)exit_spec, (int)j);
            uint64_t mj = 0;
            uint32_t xj0 = tin;
            resolve_one(w, lane, tin, ja, jstop, (uint64_t)mlo | (uint64_t)mhi << 32, jx, mj, xj0);
            if (me) { m_fin = mj; x_fin = xj0; }
        }
        // the groups this lane's walk jumps over entirely are settled with it (a 258-byte match passes four of them)
        const uint32_t xj = __builtin_amdgcn_readlane(x_fin, j);
        if (lane > j && lane < nact && stop <= xj) { used_in = xj; m_fin = 0; x_fin = xj; }
    }
    if (!have) m_fin = 0;
    const uint64_t t3 = DBG ? clock64() : 0;
    // ---- results of the segment as walked from s0
    const uint32_t cnt = (uint32_t)__popcll(m_fin);
    uint32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(incl, o); if ((int)lane >= o) incl += y; }
    const uint32_t total = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t seg_x = __builtin_amdgcn_readlane(x_fin, nact - 1);
    vw[lane] = m_fin;
    if (lane == 0) { seg_exit[seg] = seg_x; seg_count[seg] = total; }
    // ---- stage the code words, in position order: a visited position's step is the distance to the next visited one
    uint32_t *st = stage + ch.in_off + s0 + (incl - cnt);
    const uint8_t *win8 = (const uint8_t *)win32;
    // Four code words per store (a lane's run is contiguous): a wavefront's store touches 64 different lines whatever its
    // width, so a quarter of the store instructions is a quarter of the line transactions.
    uint64_t m = m_fin;
    uint32_t k = 0;
    for (uint32_t guard = 0; guard < 16; ++guard) {
        const uint64_t left = __ballot(m != 0);
        if (!left) break;
        if (COOP && (guard & 1) && (uint32_t)__popcll(left) <= COOP_K && __ballot((uint32_t)__popcll(m) >= COOP_CODES)) {
            // few lanes left with code words: their groups one after the other, a lane per POSITION
            for (uint64_t todo = left; todo; todo &= todo - 1) {
                const uint32_t j = (uint32_t)__builtin_ctzll(todo);
                const uint64_t mj = readlane64(m, j);
                const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)(incl - cnt + k), (int)j);
                const uint32_t xj = (uint32_t)__builtin_amdgcn_readlane((int)x_fin, (int)j);
                if ((mj >> lane) & 1) {
                    const uint32_t i = s0 + j * U + lane;
                    const uint64_t above = lane < 63 ? mj >> (lane + 1) : 0ull;
                    const uint32_t nxt = above ? i + 1 + (uint32_t)__builtin_ctzll(above) : xj;
                    const uint32_t d = w.cd16[i - w.c0];
                    const uint32_t code = d ? ((nxt - i) << 16) | d : (uint32_t)win8[i - w.w0] << 16;
                    stage[ch.in_off + s0 + kj + __popcll(mj & lanemask_lt())] = code;
                }
            }
            break;
        }
        // the (up to) four positions and their successors come from the mask alone; all eight LDS loads are issued before
        // the first use (one round trip per four code words: issued one by one they cost 2000-3500 cycles per round)
        uint32_t pp[4], nx[4], dd[4], bb[4];
        uint32_t nc4 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool on = m != 0;
            const uint32_t b = on ? (uint32_t)__builtin_ctzll(m) : 0u;
            m &= m - 1;                                    // (0 stays 0)
            pp[q] = a + b;
            nx[q] = m ? a + (uint32_t)__builtin_ctzll(m) : x_fin;
            nc4 += on ? 1u : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {                      // (a slot without a position reads the group's first one)
            dd[q] = w.cd16[pp[q] - w.c0];
            bb[q] = win8[pp[q] - w.w0];
        }
        uint32_t c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = dd[q] ? ((nx[q] - pp[q]) << 16) | dd[q] : bb[q] << 16;
        if (nc4 == 4) {
            struct __attribute__((packed, aligned(4))) Q { uint32_t v[4]; } qv{{c[0], c[1], c[2], c[3]}};   // 16-byte store at a dword address
            *(Q *)(st + k) = qv;
        } else {
            if (nc4 > 0) st[k] = c[0];
            if (nc4 > 1) st[k + 1] = c[1];
            if (nc4 > 2) st[k + 2] = c[2];
        }
        k += nc4;
    }
    if (DBG && stamps && lane == 0) { stamps[1] = t2 - t1; stamps[2] = t3 - t2; stamps[3] = clock64() - t3; }
}

template <bool DBG>
__global__ __launch_bounds__(p2::THREADS) void parse_walk_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const ParseWg *__restrict__ wgs, uint32_t nwgs, const uint16_t *__restrict__ cd, uint32_t max_len,
    uint64_t *__restrict__ vis, uint32_t *__restrict__ seg_exit, uint32_t *__restrict__ seg_count,
    uint32_t *__restrict__ stage, const uint32_t *__restrict__ mflags, uint64_t *__restrict__ dbg) {
    using namespace p2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    // (bit 1 of the match stage's flags: a segment of this call held runs of equal bytes — lfx_match7.hip's sample)
    const bool coop = mflags && (*mflags & 2u);
    uint32_t *win32 = (uint32_t *)smem;
    uint32_t *cd32 = (uint32_t *)(smem + OFF_CD);
    u32x4 *winx = (u32x4 *)smem, *cdx = (u32x4 *)(smem + OFF_CD);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t dbg_slot = nwgs > 1000 ? 1000u : 0u;          // (a workgroup in the middle of the launch)

    uint32_t slot = blockIdx.x;
    WalkSlot q = walk_slot(slot, nwgs, in, in_bytes, chunks, wgs, cd);
    u32x4 wv[WQ], cv[CQ];
    // (branch-free: indices past the end are clamped to the last unit — a few redundant loads and stores of the same value)
    if (q.kind == 2) {
#pragma unroll
        for (uint32_t k = 0; k < CQ; ++k) cv[k] = q.gc[min(k * THREADS + tid, q.lastc)];
#pragma unroll
        for (uint32_t k = 0; k < WQ; ++k) wv[k] = q.gw[min(k * THREADS + tid, q.lastw)];
    }
    while (slot < nwgs) {
        const uint64_t t0 = DBG ? clock64() : 0;
        // (the next slot's descriptors — two dependent scalar loads — are fetched beside the LDS stores)
        const uint32_t nslot = slot + gridDim.x;
        const WalkSlot qn = walk_slot(nslot, nwgs, in, in_bytes, chunks, wgs, cd);
        if (q.kind == 2) {
#pragma unroll
            for (uint32_t k = 0; k < CQ; ++k) cdx[min(k * THREADS + tid, q.lastc)] = cv[k];
#pragma unroll
            for (uint32_t k = 0; k < WQ; ++k) winx[min(k * THREADS + tid, q.lastw)] = wv[k];
        }
        __syncthreads();
        // the next slot's loads: in flight while this one is walked
        if (qn.kind == 2) {
#pragma unroll
            for (uint32_t k = 0; k < CQ; ++k) cv[k] = qn.gc[min(k * THREADS + tid, qn.lastc)];
#pragma unroll
            for (uint32_t k = 0; k < WQ; ++k) wv[k] = qn.gw[min(k * THREADS + tid, qn.lastw)];
        }
        uint64_t *stamps = (DBG && dbg && slot == dbg_slot) ? dbg + wave * 8 : nullptr;
        if (DBG && stamps && lane == 0) stamps[0] = clock64() - t0;
        if (q.kind == 1) {
            // nothing to walk (the chunk's last three bytes, or an empty chunk): the segments pass the walk through
            const uint32_t sidx = q.wg.seg0 + wave;
            if (sidx < q.ch.n_seg) {
                vis[q.ch.vis_base + (uint64_t)sidx * 64 + lane] = 0;
                if (lane == 0) { seg_exit[q.ch.seg_base + sidx] = sidx * PARSE_SEG; seg_count[q.ch.seg_base + sidx] = 0; }
            }
        } else if (q.kind == 2) {
            // two instances: the tests of the cooperative finish cost a text's loops 2 to 10 % and never fire there
            if (coop) walk_segment<DBG, true>(q, max_len, vis, seg_exit, seg_count, stage, win32, cd32, lane, wave, stamps);
            else walk_segment<DBG, false>(q, max_len, vis, seg_exit, seg_count, stage, win32, cd32, lane, wave, stamps);
        }
        __syncthreads();          // every wavefront is done with this slot's LDS
        q = qn;
        slot = nslot;
    }
}

constexpr uint32_t PARSE_HIST_STRIDE = 320;      // a block's symbol counters: [0,288) literal/length, [288,320) distance (= HIST_STRIDE, lfx_encode_kernels.hip)

// ------------------------------------------------------------------------------------------------
// Walk steps out of global memory, one at a time, by a whole wavefront (the chaining kernels: a handful of steps per
// segment).  Lane k compares bytes [4k, 4k+4) behind the prefix: one round settles a whole match.
struct GWalk {
    p2::ByteSrcG src;
    const uint16_t *cd;      // this chunk's candidates
    uint32_t n, max_len;
};
__device__ __forceinline__ uint32_t gwalk_step(const GWalk &g, uint32_t pos, uint32_t lane) {
    const uint32_t d = g.cd[pos];                                    // (uniform address)
    if (d == 0) return 1;
    uint32_t lim = g.n - (pos + 3);
    lim = lim > g.max_len - 3 ? g.max_len - 3 : lim;
    const uint32_t off = 4 * lane;
    uint32_t x = 0;
    if (off < lim) x = g.src.load4((uint64_t)pos + 3 + off) ^ g.src.load4((uint64_t)pos + 3 - d + off);
    const uint64_t mis = __ballot(x != 0);
    uint32_t l = lim;
    if (mis) {
        const uint32_t fl = (uint32_t)__builtin_ctzll(mis);
        const uint32_t cand = off + ((uint32_t)__builtin_ctz(x | 0x80000000u) >> 3);
        l = __builtin_amdgcn_readlane(cand, fl);
        l = l > lim ? lim : l;
    }
    return 3 + l;
}

// P2a: one wavefront per segment.  The true walk enters segment s where the walk of segment s-1 left it;
// that is the speculative exit of s-1 unless s-1 itself never merged (rare: P2b repairs those).  Re-walk
// from the entry until the walk lands on a position the speculative walk visited — from there on both
// coincide — and rewrite the visited masks, the count and the exit of the segment accordingly.
struct SegFix { uint32_t cnt, ex, mpos, kspec; };   // codes, exit; merge position, staged codes in front of it
__device__ __forceinline__ SegFix parse_rewalk(const GWalk &gw, uint64_t *__restrict__ vw, uint32_t s0, uint32_t s1,
                                               uint32_t e, uint32_t cnt, uint32_t ex, uint32_t lane) {
    using p2::U;
    uint32_t pos = e, walked = 0, spec_below = 0, merge_pos = s1;
    bool merged = false;
    // (the 64 groups' masks in ONE load, lane g holds group g's: a load per trip of the loop was 64 dependent memory round trips
    //  per segment on data whose walk jumps over whole groups — BASELINE cfg5: parse_fixseg 1.3 ms per GiB.  A group's mask is read
    //  before the loop rewrites it.)
    const uint64_t Vmine = vw[lane];
    for (uint32_t g = 0; g < 64 && !merged; ++g) {
        const uint32_t base = s0 + g * U;
        if (base >= s1) break;
        const uint32_t stop = min(base + U, s1);
        const uint64_t V = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)Vmine, (int)g) |
                           (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(Vmine >> 32), (int)g) << 32;
        if (pos >= stop) {               // wholly before the true entry: nothing visited here
            spec_below += __popcll(V);
            if (lane == 0 && V) vw[g] = 0;
            continue;
        }
        uint64_t T = 0;
        uint32_t mr = 64;
        while (pos < stop) {
            const uint32_t r = pos - base;
            if ((V >> r) & 1) { merged = true; mr = r; merge_pos = pos; break; }
            T |= 1ull << r;
            walked++;
            pos += gwalk_step(gw, pos, lane);
        }
        const uint64_t keep = mr < 64 ? (V & ~((1ull << mr) - 1)) : 0;   // speculative bits from the merge on
        spec_below += __popcll(V & ~keep);
        if (lane == 0) vw[g] = T | keep;
    }
    SegFix f;
    if (merged) { f.cnt = cnt - spec_below + walked; f.ex = ex; f.mpos = merge_pos; f.kspec = spec_below; }   // the exit stays the one already known
    else { f.cnt = walked; f.ex = pos; f.mpos = s1; f.kspec = cnt; }     // never merged inside this segment: nothing staged survives
    return f;
}

__global__ __launch_bounds__(64) void parse_fixseg_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                          const ChunkDesc *__restrict__ chunks,
                                                          const uint16_t *__restrict__ cd, uint32_t max_len,
                                                          uint64_t *__restrict__ vis,
                                                          const uint32_t *__restrict__ seg_exit,
                                                          uint32_t *__restrict__ seg_count,
                                                          uint32_t *__restrict__ seg_exit2,
                                                          uint32_t *__restrict__ seg_mpos,
                                                          uint32_t *__restrict__ seg_kspec,
                                                          const uint32_t *__restrict__ seg_map) {
    const uint32_t seg = blockIdx.x;
    const ChunkDesc ch = chunks[seg_map[seg]];
    if (ch.flags & CH_LITERALS) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t s = seg - ch.seg_base;
    const uint32_t n = (uint32_t)ch.len;
    const uint32_t end = (n > 3 ? n : 3) - 3;
    const uint32_t s0 = s * PARSE_SEG, s1 = min(s0 + PARSE_SEG, end);
    const uint32_t e = s ? seg_exit[seg - 1] : 0;                   // assumed entry
    SegFix f{seg_count[seg], seg_exit[seg], s0, 0u};                 // (entered where assumed: every staged code is final)
    if (s0 >= end) { f.cnt = 0; f.ex = e; }                          // behind the last walked position: pass through
    else if (e != s0) {
        GWalk gw{p2::make_src(in + ch.in_off, in_bytes - ch.in_off), cd + ch.in_off, n, max_len};
        f = parse_rewalk(gw, vis + ch.vis_base + (uint64_t)s * 64, s0, s1, e, f.cnt, f.ex, lane);
    }
    if (lane == 0) { seg_count[seg] = f.cnt; seg_exit2[seg] = f.ex; seg_mpos[seg] = f.mpos; seg_kspec[seg] = f.kspec; }
}

// P2b: one workgroup per chunk: checks the assumption of P2a for a batch of segments at a time (segment s was
// entered correctly iff the exit of s-1 did not change), repairs the rare segment that was not, and turns
// the counts into offsets; then the chunk's tail and the EndOfBlock marker.
// 64 lanes when a chunk has a few dozen segments (the reference's 256 KiB chunks), 1024 when one chunk is the whole input
// (schedule S1: 80 K segments per 256 MiB).
__global__ __launch_bounds__(1024) void parse_fix_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                         const ChunkDesc *__restrict__ chunks,
                                                         const uint16_t *__restrict__ cd, uint32_t max_len,
                                                         uint64_t *__restrict__ vis,
                                                         const uint32_t *__restrict__ seg_exit,
                                                         uint32_t *__restrict__ seg_count,
                                                         uint32_t *__restrict__ seg_exit2,
                                                         uint32_t *__restrict__ seg_off, uint32_t *__restrict__ codes,
                                                         uint32_t *__restrict__ ncodes, uint32_t *__restrict__ seg_mpos,
                                                         uint32_t *__restrict__ hist) {
    __shared__ uint32_t s_first_bad, s_wsum[16], s_redo[2];
    const ChunkDesc ch = chunks[blockIdx.x];
    const uint32_t tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    const uint32_t n = (uint32_t)ch.len;
    const p2::ByteSrcG src = p2::make_src(in + ch.in_off, in_bytes - ch.in_off);
    uint32_t *out = codes + ch.code_off;
    uint32_t total = 0;
    if (ch.flags & CH_LITERALS) {
        for (uint32_t s = tid; s < ch.n_seg; s += T) seg_off[ch.seg_base + s] = s * PARSE_SEG;
        total = n;
    } else {
        const uint32_t end = (n > 3 ? n : 3) - 3;
        const uint32_t *sx = seg_exit + ch.seg_base;
        uint32_t *sx2 = seg_exit2 + ch.seg_base, *sc = seg_count + ch.seg_base;
        uint32_t e_last = 0;   // true exit of the segment before the current batch
        for (uint32_t b0 = 0; b0 < ch.n_seg; ) {
            if (tid == 0) s_first_bad = 0xFFFFFFFFu;
            __syncthreads();
            const uint32_t s = b0 + tid;
            const bool have = s < ch.n_seg;
            // entry assumed by P2a vs the true exit of the predecessor
            const uint32_t assumed = have ? (s ? sx[s - 1] : 0) : 0;
            const uint32_t actual = have ? (s == b0 ? e_last : sx2[s - 1]) : 0;
            const uint64_t bad = __ballot(have && assumed != actual);
            if (bad && lane == 0) atomicMin(&s_first_bad, wave * 64 + (uint32_t)__builtin_ctzll(bad));
            __syncthreads();
            const uint32_t fb = s_first_bad;
            const uint32_t nok = fb != 0xFFFFFFFFu ? fb : min(T, ch.n_seg - b0);   // leading good segments
            // offsets of the good prefix
            const uint32_t c = (have && tid < nok) ? sc[s] : 0;
            uint32_t x = c;
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
            if (lane == 63) s_wsum[wave] = x;
            __syncthreads();
            uint32_t pre = 0, all = 0;
            for (uint32_t w = 0; w < nw; ++w) { const uint32_t v = s_wsum[w]; pre += w < wave ? v : 0u; all += v; }
            if (have && tid < nok) seg_off[ch.seg_base + s] = total + pre + x - c;
            if (nok) {
                total += all;
                e_last = sx2[b0 + nok - 1];
            }
            b0 += nok;
            if (fb != 0xFFFFFFFFu) {
                // segment b0 was entered at the wrong position: redo it from the true entry (one wavefront)
                const uint32_t sb = b0;
                if (wave == 0) {
                    const uint32_t s0 = sb * PARSE_SEG, s1 = min(s0 + PARSE_SEG, end);
                    SegFix f{sc[sb], sx2[sb], 0u, 0u};
                    if (s0 >= end) { f.cnt = 0; f.ex = e_last; }
                    else {
                        GWalk gw{src, cd + ch.in_off, n, max_len};
                        f = parse_rewalk(gw, vis + ch.vis_base + (uint64_t)sb * 64, s0, s1, e_last, f.cnt, f.ex, lane);
                    }
                    // (walked a second time: the staged codes no longer line up with the visit bits — emit all of this
                    //  segment from the bits)
                    if (lane == 0) {
                        sc[sb] = f.cnt; sx2[sb] = f.ex; seg_off[ch.seg_base + sb] = total; seg_mpos[ch.seg_base + sb] = 0xFFFFFFFFu;
                        s_redo[0] = f.cnt; s_redo[1] = f.ex;
                    }
                }
                __syncthreads();
                total += s_redo[0];
                e_last = s_redo[1];
                b0 += 1;
            }
            __syncthreads();
        }
        // default.rs:105-107: the rest are literals (at most 3 bytes)
        const uint32_t pos = ch.n_seg ? e_last : 0;
        // (hist: the block's symbol counts are taken where the code words are written — parse_emit_hist_kernel — so these few are too)
        uint32_t *hb = hist ? hist + (uint64_t)ch.block * PARSE_HIST_STRIDE : nullptr;
        for (uint32_t i = pos + tid; i < n; i += T) {
            const uint32_t byte = src.load1(i);
            out[total + (i - pos)] = byte << 16;
            if (hb) atomicAdd(&hb[byte], 1u);
        }
        if (n > pos) total += n - pos;
    }
    if (ch.flags & CH_LAST_IN_BLOCK) {
        if (tid == 0) {
            out[total] = CODE_EOB;  // encode.rs:417
            if (hist) atomicAdd(&hist[(uint64_t)ch.block * PARSE_HIST_STRIDE + 256], 1u);
        }
        total += 1;
    }
    if (tid == 0) ncodes[blockIdx.x] = total;
}

// P3: every segment emits the codes of its visited positions.
// HIST (round 6): the segment's code words are counted into `h`, this lane's replica of the workgroup's symbol counters in
// LDS — the block's histogram (DynamicHuffmanCodec::build, symbol.rs:320-341) is taken where the code words pass through
// registers anyway instead of by a kernel of its own that reads all of them again.
__device__ __forceinline__ void hist_count(uint32_t *h, uint32_t code) {
    const uint32_t dist = code & 0xFFFFu, val = code >> 16;
    uint32_t eb, ex;
    const uint32_t s1 = dist ? len_symbol(val, eb, ex) : val;
    atomicAdd(&h[s1], 1u);
    if (dist) atomicAdd(&h[288 + dist_symbol(dist, eb, ex)], 1u);
}
template <bool HIST>
__device__ __forceinline__ void emit_segment(const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc &ch, uint32_t seg,
                                             const uint16_t *__restrict__ cd, const uint64_t *__restrict__ vis,
                                             const uint32_t *__restrict__ seg_off, uint32_t *__restrict__ codes,
                                             const uint32_t *__restrict__ stage, const uint32_t *__restrict__ seg_count,
                                             const uint32_t *__restrict__ seg_exit2, const uint32_t *__restrict__ seg_mpos,
                                             const uint32_t *__restrict__ seg_kspec, uint32_t lane, uint32_t *h) {
    using p2::U;
    const uint32_t s = seg - ch.seg_base;
    const uint32_t n = (uint32_t)ch.len;
    const uint32_t s0 = s * PARSE_SEG;
    const p2::ByteSrcG src = p2::make_src(in + ch.in_off, in_bytes - ch.in_off);
    uint32_t *out = codes + ch.code_off + seg_off[seg];
    if (ch.flags & CH_LITERALS) {
        const uint32_t s1 = min(s0 + PARSE_SEG, n);
        for (uint32_t i = s0 + lane; i < s1; i += 64) {
            const uint32_t byte = src.load1(i);
            out[i - s0] = byte << 16;
            if (HIST) atomicAdd(&h[byte], 1u);
        }
        return;
    }
    const uint32_t end = (n > 3 ? n : 3) - 3;
    const uint32_t s1 = min(s0 + PARSE_SEG, end);
    const uint32_t total = seg_count[seg];
    uint32_t nout = 0;
    // positions in front of the merge point: rebuilt from the (repaired) visit bits — usually less than one group, none
    // at all when the segment was entered where the speculative walk assumed.  A visited position's step is the
    // distance to the next visited position (the segment's exit behind the last one); a match iff cd != 0.
    const uint32_t mpos = s0 < s1 ? min(seg_mpos[seg], s1) : s0;
    if (mpos > s0) {
        const uint64_t *vw = vis + ch.vis_base + (uint64_t)s * 64;
        const uint16_t *cdc = cd + ch.in_off;
        const uint64_t W = vw[lane];                                           // lane g: the mask of group g
        const uint32_t first = W ? s0 + lane * U + (uint32_t)__builtin_ctzll(W) : 0xFFFFFFFFu;
        // next visited position behind group g: the first one of the groups behind it, else the segment's true exit
        uint32_t sfx = first;                                                  // inclusive suffix minimum
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_down(sfx, o); if ((int)lane + o < 64) sfx = min(sfx, y); }
        uint32_t nxt_g = __shfl_down(sfx, 1);
        if (lane == 63) nxt_g = 0xFFFFFFFFu;
        nxt_g = min(nxt_g, seg_exit2[seg]);
        const uint64_t lt = p2::lanemask_lt();
        // (only the groups that hold a visited position in front of the merge point: on data whose matches jump over whole groups
        //  most of the 64 trips found nothing — cfg5)
        uint64_t todo;
        {
            const uint32_t mybase = s0 + lane * U;
            uint64_t mine = mybase < mpos ? W : 0ull;
            if (mybase < mpos && mpos - mybase < 64) mine &= (1ull << (mpos - mybase)) - 1;
            todo = __ballot(mine != 0);
        }
        while (todo) {
            const uint32_t g = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t base = s0 + g * U;
            // (readlane returns int: through uint32_t, or a mask whose bit 31 is set is sign-extended into all of bits 32-63)
            const uint64_t V = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)W, (int)g) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(W >> 32), (int)g) << 32;
            uint64_t m = V;
            if (mpos - base < 64) m &= (1ull << (mpos - base)) - 1;
            const uint32_t ng = __builtin_amdgcn_readlane(nxt_g, g);
            if ((m >> lane) & 1) {
                const uint32_t i = base + lane;
                const uint64_t above = lane < 63 ? V >> (lane + 1) : 0ull;
                const uint32_t nxt = above ? i + 1 + (uint32_t)__builtin_ctzll(above) : ng;
                const uint32_t d = cdc[i];
                const uint32_t code = d ? ((nxt - i) << 16) | d : src.load1(i) << 16;
                out[nout + __popcll(m & lt)] = code;
                if (HIST) hist_count(h, code);
            }
            nout += __popcll(m);
        }
    }
    // everything behind it: the codes the speculative walk staged, behind the kspec it visited in front of the merge
    const uint32_t n2 = total - nout;
    const uint32_t *st = stage + ch.in_off + s0 + seg_kspec[seg];
    // (eight loads in flight per lane: a load → store pair per trip made the copy ~20 dependent HBM round trips per segment)
    for (uint32_t j0 = 0; j0 < n2; j0 += 8 * 64) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) v[k] = st[min(j0 + 64 * k + lane, n2 - 1)];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t j = j0 + 64 * k + lane;
            if (j < n2) {
                out[nout + j] = v[k];
                if (HIST) hist_count(h, v[k]);
            }
        }
    }
}

__global__ __launch_bounds__(64) void parse_emit_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                        const ChunkDesc *__restrict__ chunks,
                                                        const uint16_t *__restrict__ cd,
                                                        const uint64_t *__restrict__ vis,
                                                        const uint32_t *__restrict__ seg_off,
                                                        uint32_t *__restrict__ codes,
                                                        const uint32_t *__restrict__ stage,
                                                        const uint32_t *__restrict__ seg_count,
                                                        const uint32_t *__restrict__ seg_exit2,
                                                        const uint32_t *__restrict__ seg_mpos,
                                                        const uint32_t *__restrict__ seg_kspec,
                                                        const uint32_t *__restrict__ seg_map) {
    const uint32_t seg = blockIdx.x;
    const ChunkDesc ch = chunks[seg_map[seg]];
    emit_segment<false>(in, in_bytes, ch, seg, cd, vis, seg_off, codes, stage, seg_count, seg_exit2, seg_mpos, seg_kspec, threadIdx.x,
                        nullptr);
}

// P3 + histogram: a workgroup of EMIT_WAVES wavefronts takes `segs_per_wg` consecutive segments of ONE chunk (blockIdx.x =
// the chunk, blockIdx.y = the part of it) — a chunk lies in one block, so the workgroup's counters are one block's — a
// wavefront every EMIT_WAVES-th of them; the counters (EMIT_HREP replicas, chosen by the lane: the lanes of one LDS atomic
// that count the same symbol are served one after the other) go to the block's with one global atomic per non-zero counter.
constexpr uint32_t EMIT_WAVES = PARSE_EMIT_WAVES, EMIT_HREP = 4;
__global__ __launch_bounds__(64 * EMIT_WAVES) void parse_emit_hist_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                                         const ChunkDesc *__restrict__ chunks,
                                                                         const uint16_t *__restrict__ cd,
                                                                         const uint64_t *__restrict__ vis,
                                                                         const uint32_t *__restrict__ seg_off,
                                                                         uint32_t *__restrict__ codes,
                                                                         const uint32_t *__restrict__ stage,
                                                                         const uint32_t *__restrict__ seg_count,
                                                                         const uint32_t *__restrict__ seg_exit2,
                                                                         const uint32_t *__restrict__ seg_mpos,
                                                                         const uint32_t *__restrict__ seg_kspec,
                                                                         uint32_t segs_per_wg, uint32_t *__restrict__ hist) {
    __shared__ uint32_t hh[EMIT_HREP * PARSE_HIST_STRIDE];
    const ChunkDesc ch = chunks[blockIdx.x];
    const uint32_t sA = blockIdx.y * segs_per_wg;
    if (sA >= ch.n_seg) return;                                   // (uniform)
    const uint32_t sB = min(ch.n_seg, sA + segs_per_wg);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t i = tid; i < EMIT_HREP * PARSE_HIST_STRIDE; i += 64 * EMIT_WAVES) hh[i] = 0;
    __syncthreads();
    uint32_t *h = hh + (lane & (EMIT_HREP - 1)) * PARSE_HIST_STRIDE;
    for (uint32_t s = sA + wave; s < sB; s += EMIT_WAVES)
        emit_segment<true>(in, in_bytes, ch, ch.seg_base + s, cd, vis, seg_off, codes, stage, seg_count, seg_exit2, seg_mpos, seg_kspec,
                           lane, h);
    __syncthreads();
    uint32_t *g = hist + (uint64_t)ch.block * PARSE_HIST_STRIDE;
    for (uint32_t i = tid; i < PARSE_HIST_STRIDE; i += 64 * EMIT_WAVES) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t r = 0; r < EMIT_HREP; ++r) v += hh[r * PARSE_HIST_STRIDE + i];
        if (v) atomicAdd(&g[i], v);
    }
}

// first-generation match kernel (the fallback behind a lane-order violation of lfx_match3.hip, LFX_MATCH_V1=1): its
// md[] words (length << 16 | distance, or byte << 16) → cd[]
__global__ __launch_bounds__(256) void md_to_cd_kernel(const uint32_t *__restrict__ md, uint64_t n, uint16_t *__restrict__ cd) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) cd[i] = (uint16_t)(md[i] & 0xFFFFu);
}

#define LFX_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)

int launch_md_to_cd(hipStream_t st, const uint32_t *md, uint64_t n, uint16_t *cd) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(md_to_cd_kernel, dim3((uint32_t)(div_up(n, 256) < 16384 ? div_up(n, 256) : 16384)), dim3(256), 0, st, md, n, cd);
    LFX_LAUNCH_CHECK();
    return 0;
}

// workgroups of the persistent walk kernel: one per CU of the current device (read once)
static uint32_t walk_grid() {
    static uint32_t g = 0;
    if (!g) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        const char *e = getenv("LFX_WALK_GRID");     // (experiments)
        if (e && atoi(e) > 0) cus = atoi(e);
        g = ((uint32_t)cus + 7) & ~7u;
    }
    return g;
}

int launch_parse(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, uint32_t nchunks,
                 uint32_t nsegs, const ParseWg *wgs, uint32_t nwgs, const uint16_t *cd, uint32_t max_len, uint64_t *vis,
                 uint32_t *seg_tmp, uint32_t *codes, uint32_t *ncodes, uint32_t *stage, const uint32_t *seg_map, int stop_after,
                 uint64_t *dbg, uint32_t *hist, uint32_t emit_per, uint32_t emit_parts, hipEvent_t ev_walked, const uint32_t *mflags,
                 int start_at) {
    if (nchunks == 0) return 0;
    // seg_tmp: six arrays of nsegs words
    uint32_t *seg_exit = seg_tmp, *seg_count = seg_tmp + nsegs, *seg_off = seg_tmp + 2 * (size_t)nsegs;
    uint32_t *seg_exit2 = seg_tmp + 3 * (size_t)nsegs, *seg_mpos = seg_tmp + 4 * (size_t)nsegs;
    uint32_t *seg_kspec = seg_tmp + 5 * (size_t)nsegs;
    if (nwgs && start_at < 1) {
        // persistent: one workgroup per CU (153 KB of LDS), every grid-th slot; the grid a multiple of 8 so that a slot keeps its XCD
        const uint32_t grid = std::min<uint32_t>((nwgs + 7) & ~7u, walk_grid());
        if (dbg)
            hipLaunchKernelGGL(parse_walk_kernel<true>, dim3(grid), dim3(p2::THREADS), 0, st, in, in_bytes, chunks, wgs, nwgs, cd, max_len,
                               vis, seg_exit, seg_count, stage, mflags, dbg);
        else
            hipLaunchKernelGGL(parse_walk_kernel<false>, dim3(grid), dim3(p2::THREADS), 0, st, in, in_bytes, chunks, wgs, nwgs, cd, max_len,
                               vis, seg_exit, seg_count, stage, mflags, dbg);
        LFX_LAUNCH_CHECK();
    }
    // (the caller's side stream — the container checksum — starts here: beside the chaining kernels, which leave the GPU
    //  mostly idle, instead of behind all of the parse)
    if (start_at < 1 && ev_walked && hipEventRecord(ev_walked, st) != hipSuccess) return (int)hipGetLastError();
    if (stop_after == 1) return 0;      // (LFX_DEBUG dumps; the walk's own bracket of the fine phase timing)
    if (nsegs) {
        hipLaunchKernelGGL(parse_fixseg_kernel, dim3(nsegs), dim3(64), 0, st, in, in_bytes, chunks, cd, max_len, vis, seg_exit,
                           seg_count, seg_exit2, seg_mpos, seg_kspec, seg_map);
        LFX_LAUNCH_CHECK();
    }
    if (stop_after == 2) return 0;
    // (workgroup size by the segments per chunk: the fold over a chunk's segments is serial in batches of that size)
    const uint32_t fix_threads = nsegs / nchunks > 128 ? 1024u : 64u;
    hipLaunchKernelGGL(parse_fix_kernel, dim3(nchunks), dim3(fix_threads), 0, st, in, in_bytes, chunks, cd, max_len, vis,
                       seg_exit, seg_count, seg_exit2, seg_off, codes, ncodes, seg_mpos, hist);
    LFX_LAUNCH_CHECK();
    if (nsegs && hist) {
        // (the caller sizes the grid: emit_per segments per workgroup, emit_parts = the longest chunk's workgroups)
        hipLaunchKernelGGL(parse_emit_hist_kernel, dim3(nchunks, emit_parts ? emit_parts : 1), dim3(64 * EMIT_WAVES), 0, st, in, in_bytes,
                           chunks, cd, vis, seg_off, codes, stage, seg_count, seg_exit2, seg_mpos, seg_kspec, emit_per ? emit_per : 1, hist);
        LFX_LAUNCH_CHECK();
    } else if (nsegs) {
        hipLaunchKernelGGL(parse_emit_kernel, dim3(nsegs), dim3(64), 0, st, in, in_bytes, chunks, cd, vis, seg_off, codes,
                           stage, seg_count, seg_exit2, seg_mpos, seg_kspec, seg_map);
        LFX_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace lfx
