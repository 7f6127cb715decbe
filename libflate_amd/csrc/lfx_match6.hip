// lfx_match6.hip — LZ77 candidate stage for gfx950, round 4, second step: lfx_match3.hip (see there for the stages P / H /
// F1 / F2 / R1, the ordered head pass by ds_mskor_rtn_b32 and the exactness argument) with its DEEP CHAIN WALKS TAKEN OUT OF
// THE KERNEL.
//
// Replaces, bit for bit, the table probe of DefaultLz77Encoder::flush (libflate_lz77/src/default.rs:76-87,146-182): per
// position the distance to the most recent earlier occurrence of its 3-byte prefix inside the window (0 = none) → cd[].
//
// Measured on lfx_match3 (profiles/r04_match_experiments.txt): 0.6 % of the positions walk more than two links of their
// bucket's chain inside the phase-B loop, 0.07 % more than eight — and a tile waits for its slowest wavefront: ending every
// walk after three trips (timing only) takes 30 % off the kernel.  Nothing in the kernel consumes a walk's answer.
// lfx_match5 handed such walks to wave 0 (one link per tile through global memory); its service call, ≈ 1600 cycles per
// tile, then bounded phase A.  Here a walk that is still running when the loop has done DEFER_TRIPS trips and the tile's
// pointer jumps are settled is WRITTEN OUT — {position, distance so far | next link << 16}, 8 bytes, to a list that
// belongs to the wavefront alone (no atomic: the count is a scalar register) — and the loop ends.  F2 writes every final
// link to `glnk` (16 bits per position, per segment).  A second kernel, lz77_walk_finish_kernel, walks the listed
// positions to their end through glnk and the input itself: one lane per walk, a memory round trip per link, but
// thousands of wavefronts at once and nobody waiting for anybody — 1.4 M walks of a 256 MiB text in well under 0.1 ms.
// A wavefront whose list is full keeps its walks in the loop, as lfx_match3 does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "lfx_common.h"
#include "lfx_device.h"

namespace lfx {

namespace m6 {

constexpr int THREADS = 1024;
constexpr uint32_t RW = 15;                   // resolver wavefronts (waves 1..15; wave 0: head pass + window)
constexpr uint32_t TILE = RW * 64;            // 960 positions
constexpr uint32_t NSUB = RW;                 // 64-position sub-tiles per tile (= exchanges of the head pass)
constexpr int HASH_BITS = 14;
// ONE ring modulus for the window bytes and the link distances: a position's ring offset indexes both.  A multiple
// of the tile size, so that a tile never straddles the end of the ring; >= window + 5 tiles + 4 (the fill runs five
// tiles ahead of R and must not touch what R(k) reads).
constexpr uint32_t RING = 40 * TILE;          // 38400
constexpr uint32_t HEAD_FAR = 33000;          // distance marker of an empty / swept head field
constexpr uint32_t SWEEP_SLICES = 32;         // the whole table is swept every 32 tiles (30720 positions)
constexpr uint32_t HB = 5;                    // exchanges per batch of the head pass (one wait per batch)
constexpr uint32_t FUTURE = 65536 - 64;       // a distance this large can only come from a lane-order violation
// link values (16 bits, in lk[] and prevd[]): 1..32768 a distance; NONE..LK_PTR-1 no link (every sum of a distance and a
// link is clamped to NONE with one v_min — no compare, no select); >= LK_PTR (lk[] only) inherit the link of in-tile
// index (v - LK_PTR)
constexpr uint32_t NONE = MAX_WINDOW + 1;
constexpr uint32_t LK_PTR = 0xC000;

// LDS layout (bytes), static so that the offsets fold into the ds instructions: the window at 0 (ds_read2_b32 offsets
// are dword indices below 256), the arrays addressed by computed indices below 64 KiB (16-bit offset field)
constexpr uint32_t OFF_WIN = 0;                                    // RING + 8 bytes (+ pad)
constexpr uint32_t OFF_LK = OFF_WIN + RING + 16;                   // TILE u16: link states of the tile being finalized
constexpr uint32_t OFF_PREVD = OFF_LK + TILE * 2;                  // RING u16
constexpr uint32_t OFF_REQ = OFF_PREVD + RING * 2;                 // TILE u32: head-pass requests (hash, valid, position); the head pass
                                                                   // puts the dword each exchange returned into the SAME slot
constexpr uint32_t OFF_HEAD = OFF_REQ + TILE * 4;                  // 8192 dwords
constexpr uint32_t LDS_BYTES = OFF_HEAD + (2u << HASH_BITS);
constexpr uint32_t DEFER_TRIPS = 2;           // trips of the phase-B loop after which a wavefront hands its walks over
static_assert(OFF_PREVD < 65536 && OFF_LK < 65536, "offset field");
static_assert(NSUB % HB == 0, "the head pass issues whole batches");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(RING % 4 == 0 && 5 * TILE + 4 + 258 <= RING - 32768, "ring slack");
static_assert(SWEEP_SLICES * TILE + 32768 + TILE + 64 < FUTURE, "head ages must stay below the violation zone");
static_assert(HEAD_FAR + SWEEP_SLICES * TILE + TILE < FUTURE && HEAD_FAR > 32768, "far marker range");
static_assert(LK_PTR + TILE <= 65536 && NONE < LK_PTR, "link states are 16 bits; every stored non-pointer is clamped to NONE");
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
    // the two dwords load4() would combine (with alignbyte(w1, w0, shift)), for a dword-aligned `off`: the combination
    // can then wait until the data is needed
    __device__ __forceinline__ void load_raw(uint64_t off, uint32_t &w0, uint32_t &w1) const {
        const uint64_t a = off + shift, idx = a >> 2;
        const uint64_t last = (nbytes + shift + 3) >> 2;
        w0 = idx < last ? w[idx] : 0;
        w1 = (shift != 0 && idx + 1 < last) ? w[idx + 1] : 0;
    }
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t hash3(uint32_t key) { return (key * 2654435761u) >> (32 - HASH_BITS); }
// ring offsets without compares: for x in [0, 2 RING) the wrapped value is the smaller of x and x - RING (unsigned);
// for a difference a - b of an offset and a distance (both below RING) it is the smaller of a - b and a - b + RING
__device__ __forceinline__ uint32_t ring_wrap(uint32_t x) { return min(x, x - RING); }
__device__ __forceinline__ uint32_t ring_back(uint32_t off, uint32_t sub) {
    const uint32_t a = off - sub;
    return min(a, a + RING);
}
// keeps a value (a load's result) materialised where it stands: the compiler otherwise sinks a load into the one branch
// that uses it, which turns an interleaved load into a dependent round trip of its own
__device__ __forceinline__ void pin(uint32_t &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ uint32_t win4(const uint32_t *win32, uint32_t off) {             // 4 bytes at ring offset
    const uint32_t w0 = win32[off >> 2], w1 = win32[(off >> 2) + 1];
    return __builtin_amdgcn_alignbyte(w1, w0, off & 3);
}

// HB 16-bit exchanges, in order, one wait.  old[i] = the dword that held the field before.
// A lane with mask 0 / value 0 leaves its dword untouched.
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

}  // namespace m6

// flags[0] |= 1 when the head pass observed a lane-order violation (results are then discarded by the host).
// DBG: per-wavefront cycle stamps of workgroup 0 (LFX_DEBUG); the production instance carries none of it.
template <bool DBG>
__global__ __launch_bounds__(m6::THREADS) void lz77_match6_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint16_t *__restrict__ cd,
    uint16_t *__restrict__ glnk, uint2 *__restrict__ wlist, uint32_t *__restrict__ wcount, uint32_t wcap,
    uint32_t *__restrict__ flags, uint64_t *__restrict__ dbg) {
    using namespace m6;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    uint32_t *head32 = (uint32_t *)(smem + OFF_HEAD);
    uint16_t *prevd = (uint16_t *)(smem + OFF_PREVD);
    uint32_t *win32 = (uint32_t *)(smem + OFF_WIN);
    uint32_t *reqb = (uint32_t *)(smem + OFF_REQ);
    uint16_t *lk = (uint16_t *)(smem + OFF_LK);
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
    // ... and this SEGMENT's final links, for the walks wave 0 takes over: a region of its own (entry 0 = the tile origin
    // `base`, so that a wavefront's 64 links are one aligned 128-byte line) — the warm-up positions of a segment are another
    // segment's own positions, and the two link structures differ there
    uint16_t *glnk_s = glnk + (uint64_t)sg.lnk_base * 64u;
    if (ch.flags & CH_LITERALS) return;           // NoCompressionLz77Encoder chunks never come here
    const uint32_t end = (n > 3 ? n : 3) - 3;     // default.rs:75
    const uint32_t q0 = sg.start;                 // first position answered by this segment
    const uint32_t q1 = min(sg.start + sg.len, end);
    if (q0 >= q1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;   // warm-up: link only
    const uint32_t base = l0 & ~3u;                               // tile origin (dword aligned)
    const int ntiles = (int)((q1 - base + TILE - 1) / TILE);
    const uint32_t n_pad = (n + 3) & ~3u;

    // ---- prologue: empty head table, window bytes for P(0)
    for (uint32_t i = tid; i < (1u << (HASH_BITS - 1)); i += THREADS) {
        const uint32_t f = (base - HEAD_FAR) & 0xFFFFu;
        head32[i] = f | f << 16;
    }
    uint32_t loaded_to = base;                                    // window holds [.., loaded_to) (wave 0 keeps it)
    {
        const uint32_t need = min(base + 2 * TILE + 4, n_pad);    // (P(0) and P(1) read it before the first fill lands)
        for (uint32_t p = loaded_to + 4 * tid; p < need; p += 4 * THREADS) {
            const uint32_t v = src.load4(p), o = p - base;        // (first pass: no wrap)
            win32[o >> 2] = v;
            if (o < 8) win32[(RING + o) >> 2] = v;
        }
        loaded_to = max(loaded_to, need);
    }
    lds_barrier();

    // resolver lane state carried from stage to stage: stage X of tile k and stage X+1 of the same tile run on the
    // same lane (index idx inside the tile), a phase or two later
    const uint32_t idx = (wave - 1) * 64 + lane;
    uint32_t key_p = 0, key_f = 0, key_r = 0;      // 3-byte prefix of the P / F / R tile position
    uint32_t hh_p = 0, hh_f = 0;                   // its hash (selects the half of the exchanged dword)
    bool val_p = false, val_f = false, val_r = false;   // position takes part in the chain structure (l0 <= p < q1)
    uint32_t cd_f = 0, cd_r = 0;                   // known answer distance (0 = walk)
    uint32_t e_f = NONE, e_r = NONE;               // own final link distance
    uint32_t lk_f = NONE;                          // F1 → F2: first link state
    bool viol = false;                             // a lane-order violation seen by this lane (reported once, at the end)
    // the walks this wavefront writes out: its own list (wcap entries), the count in a scalar register
    uint2 *my_list = wlist + ((uint64_t)blockIdx.x * RW + (wave ? wave - 1 : 0)) * wcap;
    uint32_t n_listed = 0;
    // (tile 0 = position `base` sits at ring offset 0; the loop starts two tiles early)
    uint32_t ok = RING - 2 * TILE;                 // ring offset of tile `it`
    uint32_t fill_off = loaded_to - base;          // wave 0: ring offset of position loaded_to
    // window bytes in flight — loaded (global → registers) in phase A of one iteration, stored to the LDS ring in phase A
    // of the NEXT one, by 240 resolver lanes (one dword each): wave 0 takes no part in phase A, so that phase A is as
    // long as the resolvers' own two LDS round trips
    uint32_t fill_w0 = 0, fill_w1 = 0;
    uint32_t pend_lo = loaded_to, pend_hi = loaded_to, pend_off = fill_off;
    uint32_t r_dist = 0, r_d = 0, r_found = 0;     // R1: phase A (first hop) → phase B (further hops)
    uint64_t cy_a = 0, cy_b = 0, cy_w = 0;
    uint32_t tr_sum = 0, tr_max = 0, tr_gt4 = 0, tr_gt8 = 0, tr_thin = 0;   // DBG: trips of the phase-B loop
    uint32_t hop_n[6] = {0, 0, 0, 0, 0, 0};
    constexpr uint32_t FILL_LANE0 = 4 * 64;        // resolver lanes [256, 496) move the window bytes
    constexpr uint32_t SWEEP_DW = (1u << (HASH_BITS - 1)) / SWEEP_SLICES;   // 256 head dwords per tile: resolver lanes [0, 256)
    static_assert(FILL_LANE0 + TILE / 4 <= TILE && SWEEP_DW <= FILL_LANE0, "lane assignment of the fill and the sweep");

    if (wave == 0) __builtin_amdgcn_s_setprio(3);

    for (int it = -2; it < ntiles; ++it) {
        const uint64_t c0 = DBG ? clock64() : 0;
        const uint32_t o1 = ok + TILE == RING ? 0u : ok + TILE;        // tile it+1 (RING is a multiple of TILE)
        const uint32_t o2 = o1 + TILE == RING ? 0u : o1 + TILE;        // tile it+2
        const bool do_r = it >= 0;
        const bool do_f = it + 1 >= 0 && it + 1 < ntiles;
        const bool do_p = it + 2 < ntiles;
        const uint32_t t_r = base + (uint32_t)it * TILE;            // only used when do_r
        const uint32_t t_f = t_r + TILE;                            // only used when do_f
        const uint32_t t_p = t_f + TILE;                            // only used when do_p
        // window fill bookkeeping (uniform): five tiles ahead of R — the stores land one iteration later, and P reads the
        // prefixes of tile it+2 (three bytes past its last position).  At most one tile (240 dwords) per iteration.
        const uint32_t fill_need = max(loaded_to, min(t_r + 5 * TILE + 4, n_pad));

        // =================================================== phase A
        if (wave != 0 && wave <= RW) {
            const uint32_t p_r = t_r + idx;
            const uint32_t o_r = ok + idx, o_f = o1 + idx, o_p = o2 + idx;   // ring offsets of the three positions (no wrap)
            const bool act_r = do_r && val_r && p_r >= q0;       // (val_r implies p_r < q1)
            const bool act_f = do_f && val_f;
            // ---- R1(it), first hop: chain walk, only where the answer is not already known (cd) — and then starting at
            //      the LINK of the raw predecessor, which is known to carry another prefix.  (A link never reaches in front
            //      of the first inserted position, so the distance needs no check against the position itself.)
            const bool known = act_r && cd_r != 0;
            const bool walk = act_r && cd_r == 0 && e_r <= window;          // (NONE > every window)
            uint32_t dist = known ? cd_r : (walk ? e_r : 0u);
            uint32_t found = (known && dist <= window) ? 1u : 0u;
            // -- step 0: loads
            const uint32_t d0 = prevd[walk ? ring_back(o_r, dist) : o_r];
            const uint32_t ow = reqb[idx];                      // (what the head pass left in the request's slot)
            const uint32_t kp_raw = win4(win32, o_p);
            // (the incremental sweep of stale head fields — older than the window → "far" — rides along: lanes [0, 256))
            const uint32_t slice = (uint32_t)(it + 2) % SWEEP_SLICES;
            const bool sweeper = do_p && idx < SWEEP_DW;
            uint32_t hw = 0;
            if (sweeper) hw = head32[slice * SWEEP_DW + idx];
            // -- step 0: uses
            uint32_t d = walk ? d0 : 0u;
            const uint32_t of = (hh_f & 1) ? ow >> 16 : ow & 0xFFFFu;   // what the exchange returned for this field
            uint32_t d_f = act_f ? (t_f + idx - of) & 0xFFFFu : NONE;   // (never 0: the sweep retires a field long before)
            viol |= d_f >= FUTURE;
            d_f = min(d_f, NONE);
            const bool has_f = d_f < NONE;
            // -- step 1: loads (R1 hop 1, F1 predecessor)
            dist += d;
            d = dist > window ? 0u : d;                          // default.rs:81 (inclusive window)
            const uint32_t a1 = d ? ring_back(o_r, dist) : o_r;
            const uint32_t kq1 = win4(win32, a1) & 0xFFFFFFu;
            const uint32_t dn1 = prevd[a1];
            const uint32_t af = has_f ? ring_back(o_f, d_f) : o_f;
            const uint32_t kqf = win4(win32, af) & 0xFFFFFFu;
            uint32_t pqf = prevd[af];                            // (final when the predecessor lies in an older tile)
            pin(pqf);
            if (sweeper) {
                const uint32_t far = (t_p - HEAD_FAR) & 0xFFFFu;
                uint32_t lo = hw & 0xFFFFu, hi = hw >> 16;
                const uint32_t dlo = (t_p - lo) & 0xFFFFu, dhi = (t_p - hi) & 0xFFFFu;
                if (dlo == 0 || dlo > MAX_WINDOW) lo = far;
                if (dhi == 0 || dhi > MAX_WINDOW) hi = far;
                head32[slice * SWEEP_DW + idx] = lo | hi << 16;
            }
            // -- step 1: uses
            {
                const bool hit = d != 0 && kq1 == key_r;
                found = hit ? 1u : found;
                d = (d == 0 || hit) ? 0u : dn1;
            }
            r_dist = dist; r_d = d; r_found = found;
            // F1(it+1): raw predecessor → known answer / first link state
            {
                const bool same = has_f && kqf == key_f;
                cd_f = same ? d_f : 0u;
                uint32_t e_old = min(d_f + pqf, NONE);                // predecessor in an older tile: inherit its final link
                pin(e_old);
                const uint32_t e_same = d_f > idx ? e_old : LK_PTR + (idx - d_f);   // in this tile: by pointer jumping
                const uint32_t e = same ? e_same : d_f;               // another prefix: plain link (or none)
                lk_f = e;
                lk[idx] = (uint16_t)e;       // (a slot of a position outside the chain structure is never read)
            }
            // P(it+2): request word of the head pass: hash << 17 | valid << 16 | low 16 bits of the position
            {
                const uint32_t p_p = t_p + idx;
                val_p = do_p && p_p >= l0 && p_p < q1;
                key_p = kp_raw & 0xFFFFFFu;
                hh_p = hash3(key_p);
                if (do_p) reqb[idx] = val_p ? (hh_p << 17) | 0x10000u | (p_p & 0xFFFFu) : 0u;
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
        }
        pend_lo = loaded_to; pend_hi = fill_need; pend_off = fill_off;
        fill_off = ring_wrap(fill_off + (fill_need - loaded_to));
        loaded_to = fill_need;
        const uint64_t c1 = DBG ? clock64() : 0;
        lds_barrier();
        // =================================================== phase B
        if (wave == 0) {
            if (do_p) {
                // ---- H(it+2): the ordered head pass — NSUB exchanges in position order, in batches of HB
#pragma unroll
                for (uint32_t h = 0; h < NSUB / HB; ++h) {
                    uint32_t old[HB], addr[HB], mask[HB], val[HB];
#pragma unroll
                    for (uint32_t s = 0; s < HB; ++s) {
                        const uint32_t rq = reqb[(h * HB + s) * 64 + lane];
                        const uint32_t sh = (rq >> 13) & 16u;                 // (hash & 1) * 16
                        addr[s] = head_lds + ((rq >> 18) << 2);               // dword of field hash
                        mask[s] = (0u - ((rq >> 16) & 1u)) & (0xFFFFu << sh);
                        val[s] = (rq & 0xFFFFu) << sh;
                    }
                    mskor_batch(old, addr, mask, val);
#pragma unroll
                    for (uint32_t s = 0; s < HB; ++s) reqb[(h * HB + s) * 64 + lane] = old[s];
                }
            }
        } else if (wave <= RW) {
            const uint32_t p_r = t_r + idx;
            const uint32_t o_r = ok + idx, o_f = o1 + idx;
            const bool act_r = do_r && val_r && p_r >= q0;
            // the final links of tile `it` (settled at the end of the previous iteration), for lz77_walk_finish_kernel
            if (do_r && val_r) glnk_s[p_r - base] = (uint16_t)e_r;
            // ---- F2(it+1): inherit the link of the same-prefix predecessor (pointer jumping, no ordering needed: every
            //      state a reader can observe is valid and the oldest member of a run is final from the start)
            // ---- R1(it), further hops (a few percent of the positions, but nearly every wavefront holds one).  They read
            //      link-ring entries of tiles <= it only: final since F2(it) of the previous iteration, and disjoint from the
            //      slots F2(it+1) stores at the end.
            //      ONE loop advances both chains: its trip count is the longer of the two, not their sum — every trip is an
            //      LDS round trip on the workgroup's critical path.
            //      (Measured, round 3: 3.2 trips per wavefront and tile on average, more than 4 in one of seven, more than
            //      8 in one of 23 — a rare prefix in a bucket that two frequent ones alternate in walks dozens of links —
            //      so most tiles see a wavefront with 8-9 trips.  Carrying unfinished walks over into the next two tiles
            //      (nothing in the kernel consumes their answers) cut the tiles with more than 4 trips from 36 to 4 per
            //      wavefront and made the kernel SLOWER, 3.04 ms against 2.65: the second walk's loads in every trip cost
            //      more than the waiting they remove — the tile is bound by the sum of the LDS and vector work of its
            //      fifteen wavefronts, not by the slowest of them.)
            uint32_t e = lk_f;                                         // (NONE where the position takes no part)
            // A RUN (every position repeats the prefix of the position right in front of it: zero-filled and constant regions,
            // BASELINE cfg5) is a pointer chain as long as the tile — ten rounds of pointer jumping.  Inside a wavefront it
            // collapses at once: every lane of a run takes the state of the run's first lane (a final link grows by the
            // lane distance, a pointer is shared).  Here, not in phase A: this phase waits for LDS round trips, not for
            // issue slots.  (Other wavefronts may read lk[] before or after the update: both states are valid.)
            {
                const uint64_t rm = __ballot(e == LK_PTR + idx - 1 && lane != 0);
                if (__popcll(rm) >= 8) {
                    const uint64_t below = ~rm & ((2ull << lane) - 1ull);            // lanes at or below me that start a run (or stand alone)
                    const uint32_t h = 63u - (uint32_t)__builtin_clzll(below);
                    const uint32_t eh = (uint32_t)__shfl((int)e, (int)h);
                    if ((rm >> lane) & 1) {
                        e = eh >= LK_PTR ? eh : min(eh + (lane - h), NONE);
                        lk[idx] = (uint16_t)e;
                    }
                }
            }
            uint32_t dist = r_dist, d = r_d, found = r_found;
            uint32_t trips = 0;
            bool deferred = false;
            for (;;) {
                // loads (a lane whose link state is final reads its own slot, which holds that state: the update is the
                // identity; a lane whose walk has ended reads its own position)
                const uint32_t j = min(e - LK_PTR, idx);
                const uint32_t eq = lk[j];
                dist += d;
                d = dist > window ? 0u : d;
                const uint32_t a = d ? ring_back(o_r, dist) : o_r;
                const uint32_t kq = win4(win32, a) & 0xFFFFFFu;
                const uint32_t dn = prevd[a];
                // uses
                {
                    const uint32_t en = min((idx - j) + eq, NONE);     // the predecessor's link is final: make it ours
                    e = eq < LK_PTR ? en : eq;                         // ... or it still points on: jump
                    lk[idx] = (uint16_t)e;
                }
                {
                    const bool hit = d != 0 && kq == key_r;
                    found = hit ? 1u : found;
                    d = (d == 0 || hit) ? 0u : dn;
                }
                const uint64_t ptrs = __ballot(e >= LK_PTR), walks = __ballot(d != 0);
                ++trips;
                if (!(ptrs | walks)) break;
                // ---- only walks are left (the pointer jumps — the links the next tile needs — are settled) and the loop has
                //      run its share: hand the walks over to wave 0 instead of keeping fifteen wavefronts waiting for them
                if (!ptrs && trips >= DEFER_TRIPS && n_listed + 64 <= wcap) {
                    // the walks that are left go to this wavefront's list (a later kernel finishes them); a full list keeps
                    // them here
                    if (d != 0) {
                        my_list[n_listed + (uint32_t)__popcll(walks & ((1ull << lane) - 1ull))] = make_uint2(p_r, dist | d << 16);
                        deferred = true;
                    }
                    n_listed += (uint32_t)__popcll(walks);
                    break;
                }
            }
            if (DBG) {
                tr_sum += trips; tr_max = max(tr_max, trips); tr_gt4 += trips > 4; tr_gt8 += trips > 8;
                hop_n[5] += __popcll(__ballot(deferred));      // (the hop statistics are lfx_match3's: LFX_MATCH_V3 + LFX_DEBUG)
            }
            e_f = e;
            prevd[o_f] = (uint16_t)e;        // (positions outside the chain structure: their slot is never read)
            if (act_r && !deferred) cd_c[p_r] = (uint16_t)(found ? dist : 0u);
        }
        const uint64_t c2 = DBG ? clock64() : 0;
        // ---- rotate the stage registers
        key_r = key_f; key_f = key_p; hh_f = hh_p;
        val_r = val_f; val_f = val_p;
        cd_r = cd_f; e_r = e_f;
        ok = o1;
        lds_barrier();
        const uint64_t c3 = DBG ? clock64() : 0;
        if (DBG) { cy_a += c1 - c0; cy_b += c2 - c1; cy_w += c3 - c2; }
    }
    if (wave != 0 && wave <= RW && lane == 0) wcount[(uint64_t)blockIdx.x * RW + (wave - 1)] = n_listed;
    if (__ballot(viol) && lane == 0) atomicOr(flags, 1u);             // lane-order violation (never observed)
    if (DBG && dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_a; d[1] = cy_b; d[2] = cy_w; d[3] = (uint64_t)tr_sum | (uint64_t)tr_max << 32; d[4] = (uint64_t)tr_gt4 | (uint64_t)tr_gt8 << 32;
        d[5] = (uint64_t)ntiles; d[6] = n_listed; d[7] = tr_thin;
        uint64_t *h = dbg + 128 + wave * 8;       // (behind the sixteen 8-word rows)
        for (int k = 0; k < 6; ++k) h[k] = hop_n[k];
    }
}

// The walks lz77_match6_kernel wrote out, to their end: one wavefront per list (segment x resolver wavefront), one lane per
// walk — a lane that finishes takes the list's next entry, so no lane waits for another's chain.  A link is two dependent
// loads (the prefix at the position it leads to, that position's own link): latency bound, and run by thousands of
// wavefronts at once.
__global__ __launch_bounds__(64) void lz77_walk_finish_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                              const ChunkDesc *__restrict__ chunks, const SegDesc *__restrict__ segs,
                                                              uint32_t window, uint16_t *__restrict__ cd,
                                                              const uint16_t *__restrict__ glnk, const uint2 *__restrict__ wlist,
                                                              const uint32_t *__restrict__ wcount, uint32_t wcap) {
    using namespace m6;
    const uint32_t r = blockIdx.x, lane = threadIdx.x;
    const uint32_t n = wcount[r];
    if (n == 0) return;
    const SegDesc sg = segs[r / RW];
    const ChunkDesc ch = chunks[sg.chunk];
    ByteSrc2 src;
    {
        const uint64_t a = (uint64_t)(in + ch.in_off);
        src.w = (gptr_u32)(a & ~3ull);
        src.shift = a & 3;
        src.nbytes = in_bytes - ch.in_off;
    }
    const uint32_t q0 = sg.start;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0, base = l0 & ~3u;   // (as in the match kernel)
    const uint16_t *glnk_s = glnk + (uint64_t)sg.lnk_base * 64u;
    uint16_t *cd_c = cd + ch.in_off;
    const uint2 *list = wlist + (uint64_t)r * wcap;
    uint32_t next = lane, p = 0, key = 0, dist = 0, d = 0;
    bool have = false;
    for (;;) {
        if (!have && next < n) {
            const uint2 e = list[next];
            p = e.x; dist = e.y & 0xFFFFu; d = e.y >> 16;
            key = src.load4(p) & 0xFFFFFFu;
            have = true;
            next += 64;
        }
        if (!__ballot(have)) break;
        if (have) {
            dist += d;
            // default.rs:81 (inclusive window); NONE (no link) ends here as well; a link never reaches in front of l0
            if (dist > window || dist > p - l0) { cd_c[p] = 0; have = false; }
            else {
                const uint32_t a = p - dist;
                const uint32_t kq = src.load4(a) & 0xFFFFFFu;
                const uint32_t ln = glnk_s[a - base];
                if (kq == key) { cd_c[p] = (uint16_t)dist; have = false; }      // the most recent occurrence of the prefix
                else d = ln;
            }
        }
    }
}

// scratch of the hand-over: per (segment, resolver wavefront) a list of wcap entries and a count
uint32_t match6_list_cap(uint32_t max_seg_len) { return ((max_seg_len / m6::RW) / 8 + 64 + 63) & ~63u; }
size_t match6_list_bytes(uint32_t nsegs, uint32_t wcap) { return (size_t)nsegs * m6::RW * ((size_t)wcap * 8 + 4); }

int launch_match6(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint16_t *cd, uint16_t *glnk, void *lists, uint32_t wcap, uint32_t *flags,
                  uint64_t *dbg) {
    if (nsegs == 0) return 0;
    uint2 *wlist = (uint2 *)lists;
    uint32_t *wcount = (uint32_t *)((uint8_t *)lists + (size_t)nsegs * m6::RW * wcap * 8);
    hipError_t e_ = hipMemsetAsync(wcount, 0, (size_t)nsegs * m6::RW * 4, st);
    if (e_ != hipSuccess) return (int)e_;
    if (dbg)
        hipLaunchKernelGGL(lz77_match6_kernel<true>, dim3(nsegs), dim3(m6::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, glnk, wlist, wcount, wcap, flags, dbg);
    else
        hipLaunchKernelGGL(lz77_match6_kernel<false>, dim3(nsegs), dim3(m6::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, glnk, wlist, wcount, wcap, flags, dbg);
    e_ = hipGetLastError();
    if (e_ != hipSuccess) return (int)e_;
    hipLaunchKernelGGL(lz77_walk_finish_kernel, dim3(nsegs * m6::RW), dim3(64), 0, st, in, in_bytes, chunks, segs, window, cd, glnk,
                       wlist, wcount, wcap);
    e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

}  // namespace lfx
