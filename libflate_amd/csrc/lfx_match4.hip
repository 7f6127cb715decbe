// lfx_match4.hip — LZ77 candidate stage for gfx950, ONE barrier per tile: per position, the DISTANCE to the most recent
// earlier occurrence of its 3-byte prefix inside the chunk (0 = none inside the window), written to cd[] as 16 bits per
// position.  Same answers, same data structures and the same exactness argument as lfx_match3.hip (ordered head pass by
// ds_mskor_rtn_b32, duplicate-collapsed links, chain walk with exact 3-byte verification); what changes is the SHAPE of
// the pipeline.
//
// Replaces, bit for bit, the table probe of DefaultLz77Encoder::flush (libflate_lz77/src/default.rs:76-87,146-182).
//
// Why: lfx_match3.hip runs two LDS-only barriers per tile of 960 positions, and its waves move in lock step — all of them
// issue their LDS gathers at once, wait for the burst to drain (≈ 450 cycles per dependent round trip under that load),
// then all compute while the LDS idles.  Counters (profiles/r03_*): waves parked 60 % of their life, VALU 52 % busy, LDS
// array 25 % busy; ≈ 7 dependent round trips per tile on the critical path, because the stages of one tile
// (F1 → barrier → F2 → barrier → F1 of the next tile) are chained through the barriers.
//
// The chain F2(t) → F1(t+1) exists because a position whose same-prefix predecessor q lies in the previous tile inherits
// q's final link, and that link is final only behind F2 of q's tile.  Here F1 DEFERS that case: it stores a pointer
// "inherit the link of position k of the previous tile" (PTR_PREV + k), and F2 — one iteration later, when the previous
// tile IS final — resolves it with one read of the link ring.  With that, the five stages of five different tiles run in
// ONE phase, their loads issued together and waited for together:
//
//   iteration i   resolvers: P(i+3)  3-byte prefix, hash, request word                                  (1 round trip)
//                            F1(i+1) what the head pass returned → raw predecessor → known answer /
//                                    first link state: a distance, an in-tile pointer, or a deferred one (2 round trips)
//                            F2(i)   pointer jumping → final link → link ring                           (2 + tail)
//                            R1(i-1) chain walk → cd[]                                                   (2 + tail)
//                 wave 0:    H(i+2)  ordered head pass (15 exchanges), then the incremental sweep of stale head fields
//   barrier
//
// i.e. ≈ 4-5 dependent round trips and one barrier per tile instead of 7 and two.  Request / result words of the head
// pass and the link states are double buffered (the buffers alternate with the tile's parity); the head pass writes
// what the exchanges returned over the request words it has just consumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "lfx_common.h"
#include "lfx_device.h"

namespace lfx {

namespace m4 {

constexpr int THREADS = 1024;
constexpr uint32_t RW = 15;                   // resolver wavefronts (waves 1..15; wave 0: head pass + sweep)
constexpr uint32_t TILE = RW * 64;            // 960 positions
constexpr uint32_t NSUB = RW;                 // 64-position sub-tiles per tile (= exchanges of the head pass)
constexpr int HASH_BITS = 14;
// ONE ring modulus for the window bytes and the link distances: a position's ring offset indexes both.  A multiple of
// the tile size, so that a tile never straddles the end of the ring.  It holds [R1's tile - 32 KiB, P's tile + fill in
// flight): window + 6 tiles + 4.
constexpr uint32_t RING = 41 * TILE;          // 39360
constexpr uint32_t HEAD_FAR = 33000;          // distance marker of an empty / swept head field
constexpr uint32_t SWEEP_SLICES = 32;         // the whole table is swept every 32 tiles (30720 positions)
constexpr uint32_t HB = 5;                    // exchanges per batch of the head pass (one wait per batch)
constexpr uint32_t FUTURE = 65536 - 64;       // a distance this large can only come from a lane-order violation
// link values (16 bits, in lk[] and prevd[]): 1..32768 a distance; NONE..LK_PTR-1 no link (every sum of a distance and a
// link is clamped to NONE with one v_min — no compare, no select); lk[] only: LK_PTR + j inherit the link of in-tile
// index j; PTR_PREV + k inherit the link of index k of the PREVIOUS tile (final one iteration later)
constexpr uint32_t NONE = MAX_WINDOW + 1;
constexpr uint32_t LK_PTR = 0xC000;
constexpr uint32_t PTR_PREV = 0xD000;

// LDS layout (bytes)
constexpr uint32_t OFF_WIN = 0;                                    // RING + 8 bytes (+ pad)
constexpr uint32_t OFF_LK = OFF_WIN + RING + 16;                   // 2 x TILE u16: link states, by tile parity
constexpr uint32_t OFF_PREVD = OFF_LK + 2 * TILE * 2;              // RING u16
constexpr uint32_t OFF_REQ = OFF_PREVD + RING * 2;                 // 2 x TILE u32: head-pass requests / results, by tile parity
constexpr uint32_t OFF_HEAD = OFF_REQ + 2 * TILE * 4;              // 8192 dwords
constexpr uint32_t LDS_BYTES = OFF_HEAD + (2u << HASH_BITS);
static_assert(NSUB % HB == 0, "the head pass issues whole batches");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(RING % 4 == 0 && 6 * TILE + 4 + 8 <= RING - 32768, "ring slack");
static_assert(SWEEP_SLICES * TILE + 32768 + TILE + 64 < FUTURE, "head ages must stay below the violation zone");
static_assert(HEAD_FAR + SWEEP_SLICES * TILE + TILE < FUTURE && HEAD_FAR > 32768, "far marker range");
static_assert(NONE < LK_PTR && LK_PTR + TILE <= PTR_PREV && PTR_PREV + TILE <= 65536, "link states are 16 bits");
static_assert(((1u << (HASH_BITS - 1)) / SWEEP_SLICES) % 64 == 0, "sweep slice per lane");

struct ByteSrc2 {
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
    // the two dwords load4() would combine (with alignbyte(w1, w0, shift)), for a dword-aligned `off`
    __device__ __forceinline__ void load_raw(uint64_t off, uint32_t &w0, uint32_t &w1) const {
        const uint64_t a = off + shift, idx = a >> 2;
        const uint64_t last = (nbytes + shift + 3) >> 2;
        w0 = idx < last ? w[idx] : 0;
        w1 = (shift != 0 && idx + 1 < last) ? w[idx + 1] : 0;
    }
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t hash3(uint32_t key) { return (key * 2654435761u) >> (32 - HASH_BITS); }
__device__ __forceinline__ uint32_t ring_wrap(uint32_t x) { return min(x, x - RING); }          // x in [0, 2 RING)
__device__ __forceinline__ uint32_t ring_back(uint32_t off, uint32_t sub) {                      // off, sub < RING
    const uint32_t a = off - sub;
    return min(a, a + RING);
}
__device__ __forceinline__ uint32_t ring_next(uint32_t o) { return o + TILE == RING ? 0u : o + TILE; }
__device__ __forceinline__ void pin(uint32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ uint32_t win4(const uint32_t *win32, uint32_t off) {             // 4 bytes at ring offset
    const uint32_t w0 = win32[off >> 2], w1 = win32[(off >> 2) + 1];
    return __builtin_amdgcn_alignbyte(w1, w0, off & 3);
}

// HB 16-bit exchanges, in order, one wait.  old[i] = the dword that held the field before.
__device__ __forceinline__ void mskor_batch(uint32_t (&old)[HB], const uint32_t (&addr)[HB], const uint32_t (&mask)[HB],
                                            const uint32_t (&val)[HB]) {
    static_assert(HB == 5, "operand list below");
    asm volatile(
        "ds_mskor_rtn_b32 %0, %5, %10, %15\n\t"
        "ds_mskor_rtn_b32 %1, %6, %11, %16\n\t"
        "ds_mskor_rtn_b32 %2, %7, %12, %17\n\t"
        "ds_mskor_rtn_b32 %3, %8, %13, %18\n\t"
        "ds_mskor_rtn_b32 %4, %9, %14, %19\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3]), "=&v"(old[4])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]),
          "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]),
          "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4])
        : "memory");
}

// F2: the LDS u16 a link state has to look at (byte offset into smem) — the slot of the in-tile predecessor, the link
// ring entry of the previous tile's position, or (state already final) the lane's own slot: the update below is then the
// identity
__device__ __forceinline__ uint32_t f2_addr(uint32_t e, uint32_t idx, uint32_t lk_off, uint32_t o_prev) {
    const uint32_t j = min(e - LK_PTR, idx);                 // final state: e - LK_PTR wraps to a huge value → own slot
    const uint32_t a_lk = lk_off + 2 * j;
    const uint32_t a_pv = OFF_PREVD + 2 * (o_prev + (e - PTR_PREV));
    return e >= PTR_PREV ? a_pv : a_lk;
}
// F2: one jump.  x = what f2_addr() pointed at.
__device__ __forceinline__ uint32_t f2_step(uint32_t e, uint32_t x, uint32_t idx) {
    const uint32_t j = min(e - LK_PTR, idx);
    const uint32_t via_lk = x < LK_PTR ? min((idx - j) + x, NONE) : x;                 // the predecessor's final link made ours — or jump on
    const uint32_t via_pv = min((idx + TILE - (e - PTR_PREV)) + x, NONE);              // previous tile: final by construction
    const uint32_t nxt = e >= PTR_PREV ? via_pv : via_lk;
    return e < LK_PTR ? e : nxt;                                                        // a final state stays what it is, whatever its slot holds
}

}  // namespace m4

// flags[0] |= 1 when the head pass observed a lane-order violation (results are then discarded by the host).
// DBG: per-wavefront cycle stamps of workgroup 0 (LFX_DEBUG); the production instance carries none of it.
template <bool DBG>
__global__ __launch_bounds__(m4::THREADS) void lz77_match4_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint16_t *__restrict__ cd,
    uint32_t *__restrict__ flags, uint64_t *__restrict__ dbg) {
    using namespace m4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    uint32_t *head32 = (uint32_t *)(smem + OFF_HEAD);
    uint16_t *prevd = (uint16_t *)(smem + OFF_PREVD);
    uint32_t *win32 = (uint32_t *)(smem + OFF_WIN);
    // LDS byte address of head[] for the asm exchanges (taking it from the pointer also makes the array escape)
    const uint32_t head_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)head32;

    const SegDesc sg = segs[blockIdx.x];
    const ChunkDesc ch = chunks[sg.chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = (uint32_t)ch.len;
    ByteSrc2 src;
    {
        const uint64_t a = (uint64_t)(in + ch.in_off);
        src.w = (gptr_u32)(a & ~3ull);
        src.shift = a & 3;
        src.nbytes = in_bytes - ch.in_off;
    }
    uint16_t *cd_c = cd + ch.in_off;              // this chunk's answers
    if (ch.flags & CH_LITERALS) return;           // NoCompressionLz77Encoder chunks never come here
    const uint32_t end = (n > 3 ? n : 3) - 3;     // default.rs:75
    const uint32_t q0 = sg.start;                 // first position answered by this segment
    const uint32_t q1 = min(sg.start + sg.len, end);
    if (q0 >= q1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;   // warm-up: link only
    const uint32_t base = l0 & ~3u;                               // tile origin (dword aligned)
    const int ntiles = (int)((q1 - base + TILE - 1) / TILE);
    const uint32_t n_pad = (n + 3) & ~3u;

    // ---- prologue: empty head table, window bytes of the first two tiles
    for (uint32_t i = tid; i < (1u << (HASH_BITS - 1)); i += THREADS) {
        const uint32_t f = (base - HEAD_FAR) & 0xFFFFu;
        head32[i] = f | f << 16;
    }
    uint32_t loaded_to = base;                                    // the ring holds [.., loaded_to) (uniform)
    {
        const uint32_t need = min(base + 2 * TILE + 4, n_pad);
        for (uint32_t p = loaded_to + 4 * tid; p < need; p += 4 * THREADS) {
            const uint32_t v = src.load4(p), o = p - base;        // (first pass: no wrap)
            win32[o >> 2] = v;
            if (o < 8) win32[(RING + o) >> 2] = v;
        }
        loaded_to = max(loaded_to, need);
    }
    lds_barrier();

    // resolver lane state: the same lane (index idx inside the tile) carries a tile's position from stage to stage
    const uint32_t idx = (wave - 1) * 64 + lane;
    uint32_t key_p = 0, key_h = 0, key_f = 0, key_g = 0, key_r = 0;     // 3-byte prefix
    uint32_t hh_p = 0, hh_h = 0, hh_f = 0;                              // hash (selects the half of the exchanged dword)
    bool val_p = false, val_h = false, val_f = false, val_g = false, val_r = false;   // takes part in the chain structure
    uint32_t cd_f = 0, cd_g = 0, cd_r = 0;         // known answer distance (0 = walk)
    uint32_t lk_f = NONE, lk_g = NONE;             // F1 → F2: first link state
    uint32_t e_g = NONE, e_r = NONE;               // own final link distance
    bool viol = false;
    // ring offsets of the five tiles in flight (tile 0 = position `base` at ring offset 0; the loop starts three tiles early)
    uint32_t o_r = RING - 4 * TILE;                // tile it-1
    uint32_t fill_off = loaded_to - base;          // ring offset of position loaded_to
    uint32_t fill_w0 = 0, fill_w1 = 0;             // window dword in flight (resolver lanes [256, 496))
    uint32_t pend_lo = loaded_to, pend_hi = loaded_to, pend_off = fill_off;
    uint64_t cy_a = 0, cy_w = 0;
    constexpr uint32_t FILL_LANE0 = 4 * 64;
    static_assert(FILL_LANE0 + TILE / 4 <= TILE, "lanes of the window fill");

    if (wave == 0) __builtin_amdgcn_s_setprio(3);

    for (int it = -3; it <= ntiles; ++it) {
        const uint64_t c0 = DBG ? clock64() : 0;
        const uint32_t o_g = ring_next(o_r), o_f = ring_next(o_g), o_h = ring_next(o_f), o_p = ring_next(o_h);
        const bool do_r = it - 1 >= 0 && it - 1 < ntiles;
        const bool do_g = it >= 0 && it < ntiles;
        const bool do_f = it + 1 >= 0 && it + 1 < ntiles;
        const bool do_h = it + 2 >= 0 && it + 2 < ntiles;
        const bool do_p = it + 3 >= 0 && it + 3 < ntiles;
        const uint32_t t_g = base + (uint32_t)it * TILE;            // first position of tile it (wraps harmlessly when unused)
        const uint32_t t_r = t_g - TILE, t_f = t_g + TILE, t_h = t_f + TILE, t_p = t_h + TILE;
        uint32_t *rb_p = (uint32_t *)(smem + OFF_REQ) + ((uint32_t)(it + 3) & 1) * TILE;      // requests of tile it+3
        uint32_t *rb_h = (uint32_t *)(smem + OFF_REQ) + ((uint32_t)(it + 2) & 1) * TILE;      // requests → results of tile it+2
        const uint32_t *rb_f = (const uint32_t *)(smem + OFF_REQ) + ((uint32_t)(it + 1) & 1) * TILE;   // results of tile it+1
        uint16_t *lk_fb = (uint16_t *)(smem + OFF_LK) + ((uint32_t)(it + 1) & 1) * TILE;      // link states of tile it+1 (F1 writes)
        const uint32_t lk_g_off = OFF_LK + ((uint32_t)it & 1) * TILE * 2;                     // link states of tile it (F2), byte offset
        uint16_t *lk_gb = (uint16_t *)(smem + lk_g_off);
        // window fill (uniform): six tiles ahead of R — the stores land one iteration later, P reads the prefixes of tile
        // it+3 (three bytes past its last position).  At most one tile (240 dwords) per iteration.
        const uint32_t fill_need = max(loaded_to, min(t_r + 7 * TILE + 4, n_pad));

        if (wave == 0) {
            if (do_h) {
                // ---- H(it+2): the ordered head pass — NSUB exchanges in position order, in batches of HB; the dwords the
                //      exchanges returned go where the request words were
#pragma unroll
                for (uint32_t h = 0; h < NSUB / HB; ++h) {
                    uint32_t old[HB], addr[HB], mask[HB], val[HB];
#pragma unroll
                    for (uint32_t s = 0; s < HB; ++s) {
                        const uint32_t rq = rb_h[(h * HB + s) * 64 + lane];
                        const uint32_t sh = (rq >> 13) & 16u;                 // (hash & 1) * 16
                        addr[s] = head_lds + ((rq >> 18) << 2);               // dword of field hash
                        mask[s] = (0u - ((rq >> 16) & 1u)) & (0xFFFFu << sh);
                        val[s] = (rq & 0xFFFFu) << sh;
                    }
                    mskor_batch(old, addr, mask, val);
#pragma unroll
                    for (uint32_t s = 0; s < HB; ++s) rb_h[(h * HB + s) * 64 + lane] = old[s];
                }
            }
            if (do_p) {
                // ---- incremental sweep: stale fields (older than the window, seen from the next tile to be inserted) →
                //      "far".  By this wavefront, behind its exchanges: a wavefront's LDS operations execute in order, so
                //      the read-modify-write below cannot interleave with an exchange.
                const uint32_t slice = (uint32_t)(it + 3) % SWEEP_SLICES;
                const uint32_t far = (t_p - HEAD_FAR) & 0xFFFFu;
                constexpr uint32_t PER = (1u << (HASH_BITS - 1)) / SWEEP_SLICES / 64;
                uint32_t hw[PER];
#pragma unroll
                for (uint32_t q = 0; q < PER; ++q) hw[q] = head32[slice * (PER * 64) + q * 64 + lane];
#pragma unroll
                for (uint32_t q = 0; q < PER; ++q) {
                    uint32_t lo = hw[q] & 0xFFFFu, hi = hw[q] >> 16;
                    const uint32_t dlo = (t_p - lo) & 0xFFFFu, dhi = (t_p - hi) & 0xFFFFu;
                    if (dlo == 0 || dlo > MAX_WINDOW) lo = far;
                    if (dhi == 0 || dhi > MAX_WINDOW) hi = far;
                    head32[slice * (PER * 64) + q * 64 + lane] = lo | hi << 16;
                }
            }
        } else if (wave <= RW) {
            const uint32_t p_r = t_r + idx;
            const uint32_t or_i = o_r + idx, og_i = o_g + idx, of_i = o_f + idx, op_i = o_p + idx;   // ring offsets (no wrap)
            const bool act_r = do_r && val_r && p_r >= q0;       // (val_r implies p_r < q1)
            const bool act_f = do_f && val_f;
            // ---- R1(it-1), setup: chain walk only where the answer is not already known (cd) — and then starting at the
            //      LINK of the raw predecessor, which is known to carry another prefix.  (A link never reaches in front of
            //      the first inserted position, so the distance needs no check against the position itself.)
            const bool known = act_r && cd_r != 0;
            const bool walk = act_r && cd_r == 0 && e_r <= window;          // (NONE > every window)
            uint32_t dist = known ? cd_r : (walk ? e_r : 0u);
            uint32_t found = (known && dist <= window) ? 1u : 0u;
            // ---- F2(it), setup
            uint32_t e = do_g ? lk_g : NONE;                                // (NONE where the position takes no part)
            // ================= round 1: loads
            const uint32_t d0 = prevd[walk ? ring_back(or_i, dist) : or_i];
            const uint32_t ow = rb_f[idx];
            const uint32_t kp_raw = win4(win32, op_i);
            const uint32_t x1 = *(const uint16_t *)(smem + f2_addr(e, idx, lk_g_off, o_r));
            // ================= round 1: uses
            uint32_t d = walk ? d0 : 0u;
            const uint32_t of = (hh_f & 1) ? ow >> 16 : ow & 0xFFFFu;   // what the exchange returned for this field
            uint32_t d_f = act_f ? (t_f + idx - of) & 0xFFFFu : NONE;   // (never 0: the sweep retires a field long before)
            viol |= d_f >= FUTURE;
            d_f = min(d_f, NONE);
            const bool has_f = d_f < NONE;
            e = f2_step(e, x1, idx);
            lk_gb[idx] = (uint16_t)e;
            // ================= round 2: loads (R1 hop 1, F1 predecessor, F2 second jump)
            dist += d;
            d = dist > window ? 0u : d;                          // default.rs:81 (inclusive window)
            const uint32_t a1 = d ? ring_back(or_i, dist) : or_i;
            const uint32_t kq1 = win4(win32, a1) & 0xFFFFFFu;
            const uint32_t dn1 = prevd[a1];
            const uint32_t af = has_f ? ring_back(of_i, d_f) : of_i;
            const uint32_t kqf = win4(win32, af) & 0xFFFFFFu;
            uint32_t pqf = prevd[af];                            // (final when the predecessor lies two tiles back or more)
            pin(pqf);
            const uint32_t x2 = *(const uint16_t *)(smem + f2_addr(e, idx, lk_g_off, o_r));
            // ================= round 2: uses
            {
                const bool hit = d != 0 && kq1 == key_r;
                found = hit ? 1u : found;
                d = (d == 0 || hit) ? 0u : dn1;
            }
            e = f2_step(e, x2, idx);
            lk_gb[idx] = (uint16_t)e;
            // F1(it+1): raw predecessor → known answer / first link state
            {
                const bool same = has_f && kqf == key_f;
                cd_f = same ? d_f : 0u;
                uint32_t e_old = min(d_f + pqf, NONE);                // predecessor two tiles back or more: inherit its final link
                pin(e_old);
                // ... in this tile: by pointer jumping; in the previous tile (final one iteration from now): deferred
                const uint32_t e_same = d_f <= idx ? LK_PTR + (idx - d_f) : (d_f <= idx + TILE ? PTR_PREV + (idx + TILE - d_f) : e_old);
                const uint32_t ef = same ? e_same : d_f;              // another prefix: plain link (or none)
                lk_f = ef;
                lk_fb[idx] = (uint16_t)ef;   // (a slot of a position outside the chain structure is never read)
            }
            // P(it+3): request word of the head pass: hash << 17 | valid << 16 | low 16 bits of the position
            {
                const uint32_t p_p = t_p + idx;
                val_p = do_p && p_p >= l0 && p_p < q1;
                key_p = kp_raw & 0xFFFFFFu;
                hh_p = hash3(key_p);
                if (do_p) rb_p[idx] = val_p ? (hh_p << 17) | 0x10000u | (p_p & 0xFFFFu) : 0u;
            }
            // ---- window bytes: the previous iteration's dword → LDS ring, then this iteration's load (lanes [256, 496))
            if (idx >= FILL_LANE0 && idx < FILL_LANE0 + TILE / 4) {
                const uint32_t f4 = 4 * (idx - FILL_LANE0);
                if (pend_lo + f4 < pend_hi) {
                    const uint32_t v = __builtin_amdgcn_alignbyte(fill_w1, fill_w0, (uint32_t)src.shift);
                    const uint32_t o = ring_wrap(pend_off + f4);
                    win32[o >> 2] = v;
                    if (o < 8) win32[(RING + o) >> 2] = v;
                }
                fill_w0 = fill_w1 = 0;
                if (loaded_to + f4 < fill_need) src.load_raw(loaded_to + f4, fill_w0, fill_w1);
            }
            // ================= tail: further jumps of F2(it) and further hops of R1(it-1), ONE loop for both chains (its
            //                   trip count is the longer of the two, not their sum)
            for (uint32_t guard = 0; __ballot(e >= LK_PTR || d != 0); ++guard) {
                if (guard > 40000u) { viol = true; break; }      // (cannot happen: a chain has at most one hop per window position)
                const uint32_t x = *(const uint16_t *)(smem + f2_addr(e, idx, lk_g_off, o_r));
                dist += d;
                d = dist > window ? 0u : d;
                const uint32_t a = d ? ring_back(or_i, dist) : or_i;
                const uint32_t kq = win4(win32, a) & 0xFFFFFFu;
                const uint32_t dn = prevd[a];
                e = f2_step(e, x, idx);
                lk_gb[idx] = (uint16_t)e;
                const bool hit = d != 0 && kq == key_r;
                found = hit ? 1u : found;
                d = (d == 0 || hit) ? 0u : dn;
            }
            e_g = e;
            if (do_g) prevd[og_i] = (uint16_t)e;        // (positions outside the chain structure: their slot is never read)
            if (act_r) cd_c[p_r] = (uint16_t)(found ? dist : 0u);
        }
        pend_lo = loaded_to; pend_hi = fill_need; pend_off = fill_off;
        fill_off = ring_wrap(fill_off + (fill_need - loaded_to));
        loaded_to = fill_need;
        const uint64_t c1 = DBG ? clock64() : 0;
        // ---- rotate the stage registers
        key_r = key_g; key_g = key_f; key_f = key_h; key_h = key_p;
        hh_f = hh_h; hh_h = hh_p;
        val_r = val_g; val_g = val_f; val_f = val_h; val_h = val_p;
        cd_r = cd_g; cd_g = cd_f;
        e_r = e_g; lk_g = lk_f;
        o_r = o_g;
        lds_barrier();
        const uint64_t c2 = DBG ? clock64() : 0;
        if (DBG) { cy_a += c1 - c0; cy_w += c2 - c1; }
    }
    if (__ballot(viol) && lane == 0) atomicOr(flags, 1u);             // lane-order violation (never observed)
    if (DBG && dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_a; d[1] = 0; d[2] = cy_w; d[3] = 0; d[4] = 0; d[5] = (uint64_t)ntiles;
    }
}

int launch_match4(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint16_t *cd, uint32_t *flags, uint64_t *dbg) {
    if (nsegs == 0) return 0;
    if (dbg)
        hipLaunchKernelGGL(lz77_match4_kernel<true>, dim3(nsegs), dim3(m4::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, flags, dbg);
    else
        hipLaunchKernelGGL(lz77_match4_kernel<false>, dim3(nsegs), dim3(m4::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, flags, dbg);
    const hipError_t e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

}  // namespace lfx
