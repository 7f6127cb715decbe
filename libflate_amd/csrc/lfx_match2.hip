// lfx_match2.hip — second-generation LZ77 match stage for gfx950: per position, the most recent earlier occurrence
// of its 3-byte prefix inside the chunk and the match length, written to md[] for the parse kernels.
//
// Replaces, bit for bit, the table probe and longest_common_prefix of DefaultLz77Encoder::flush
// (libflate_lz77/src/default.rs:76-87,122-129,146-182).  Parse independence (tests/test_host_pipeline.py): the
// reference inserts EVERY position < end exactly once and in order (default.rs:78,92-97), so
// cand(i) = max{ j < i : buf[j..j+3] == buf[i..i+3] } does not depend on which positions the walk visits.
//
// One workgroup of 16 wavefronts per segment, a software pipeline over tiles of 832 positions (13 resolver wavefronts
// x 64 lanes), two LDS-only barriers per tile.  Stages of iteration k:
//
//   phase A   resolvers (waves 1..13): R1(k)   first hop of the chain walk for the positions whose answer is not known;
//                                              walks that go on are handed to a helper
//                                      F1(k+1) what the head pass returned → raw predecessor → same prefix? (answer
//                                              known) : plain link ; first link state lk[]
//             wave 0:                  window loads (global → registers), sweep of stale head fields, H(k+2) part 1
//   phase B   resolvers:               F2(k+1) duplicate-collapsed link by pointer jumping → prevd[]
//                                      R2(k)   match length → md[]
//                                      P(k+3)  3-byte prefix, hash, request word of the head pass
//             wave 0:                  H(k+2) part 2, window stores
//   helpers (waves 14, 15; tiles alternate): the handed-over walks of a tile, one per lane, over three phases, then
//                                      match length and md[] for those positions in a fourth
//
// H, the ordered head pass: head[] holds 2^14 16-bit fields (low 16 bits of the most recent position per hash) packed
// two per dword; ONE ds_mskor_rtn_b32 per 64 positions exchanges the field and returns the old dword.  The LDS serves
// the lanes of one instruction that hit the same field in ascending lane order and a wavefront's instructions in issue
// order (measured: tools/exp/mskor_test.hip, 0 violations in 1.3 M conflicting operations), so every lane receives
// exactly its raw predecessor ph(p) = most recent earlier position with the same hash.  A lane that observes a value
// "from the future" (distance >= 65536-64) proves a violation: the kernel raises a flag and the host re-runs the
// first-generation kernel.
//
// Why helpers: 2 % of the positions walk more than one hop (crowded buckets; mostly keys that do not occur in the
// window at all), but nearly every wavefront holds one, every further hop is a full LDS round trip for the whole
// wavefront, and the deepest walk of a tile takes ten and more.  Measured by switching stages off (LFX_ABLATE): 1.7 of
// 4.4 ms.  Handed to a helper the walks are dense (one per lane), run beside the next tiles' stages, and the
// resolvers never loop.
//
// Duplicate collapsing (exactness argument as in the first-generation kernel): link(p) = ph(p) if the
// prefixes differ, else link(ph(p)); the chain from p therefore visits the most recent member of every run of equal
// prefixes in its bucket, in decreasing position order, and the walk stops at the first exact 3-byte match (the most
// recent occurrence) or when the distance exceeds the window (default.rs:81).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "lfx_common.h"
#include "lfx_device.h"

namespace lfx {

namespace m2 {

constexpr int THREADS = 1024;
constexpr uint32_t RW = 13;                   // resolver wavefronts (waves 1..13)
constexpr uint32_t TILE = RW * 64;            // 832 positions
constexpr uint32_t NSUB = RW;                 // 64-position sub-tiles per tile (= exchanges of the head pass)
constexpr uint32_t HA = 4;                    // exchanges of a head pass issued in phase A (the rest in phase B)
constexpr int HASH_BITS = 14;
constexpr uint32_t PRING = 32768 + 2560;      // prevd ring (entries)  >= window + 3 TILE (helpers walk tile k-1 while F2
                                              // writes tile k+1)
constexpr uint32_t WRING = 37776;             // window ring (bytes)   >= window + 6 TILE + 4 (P reads tile k+3, the fill runs
                                              // two tiles ahead of it, helpers finish tile k-2)
constexpr uint32_t HEAD_FAR = 33000;          // distance marker of an empty / swept head field
constexpr uint32_t SWEEP_SLICES = 32;         // the whole table is swept every 32 tiles (26624 positions)
constexpr uint32_t FUTURE = 65536 - 64;       // a distance this large can only come from a lane-order violation
constexpr uint32_t FILL_LOADS = (TILE + 255) / 256;   // dword loads per lane of wave 0 and tile
constexpr uint32_t LK_PTR = 32769;            // lk value >= LK_PTR: inherit the link of in-tile index (v - LK_PTR)
constexpr uint32_t QCAP = 128;                // walks a helper takes per tile (more: their owners walk them)
constexpr uint32_t HE = QCAP / 64;            // ... = entries per helper lane

// LDS layout (bytes)
constexpr uint32_t OFF_HEAD = 0;                                   // 8192 dwords
constexpr uint32_t OFF_PREVD = OFF_HEAD + (2u << HASH_BITS);       // PRING u16
constexpr uint32_t OFF_WIN = OFF_PREVD + PRING * 2;                // WRING + 8 bytes (+ pad)
constexpr uint32_t OFF_REQ = OFF_WIN + WRING + 16;                 // TILE u32: head-pass requests (hash, valid, position)
constexpr uint32_t OFF_OLD = OFF_REQ + TILE * 4;                   // 2 x TILE u32 (tile parity): the dwords the exchanges returned
constexpr uint32_t OFF_LK = OFF_OLD + 2 * TILE * 4;                // TILE u16: link states of the tile being finalized
constexpr uint32_t OFF_Q = OFF_LK + TILE * 2;                      // 2 x (QCAP x 2 u32 + counter): walks handed to the helpers
constexpr uint32_t QWORDS = QCAP * 2 + 4;
constexpr uint32_t LDS_BYTES = OFF_Q + 2 * QWORDS * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(WRING % 4 == 0 && 6 * TILE + 4 <= WRING - 32768, "window ring slack");
static_assert(3 * TILE <= PRING - 32768, "prevd ring slack");
static_assert(SWEEP_SLICES * TILE + 32768 + TILE + 64 < FUTURE, "head ages must stay below the violation zone");
static_assert(HEAD_FAR + SWEEP_SLICES * TILE + TILE < FUTURE && HEAD_FAR > 32768, "far marker range");
static_assert(LK_PTR + TILE <= 65536, "link states are 16 bits");
static_assert(((1u << (HASH_BITS - 1)) / SWEEP_SLICES) % 64 == 0, "sweep slice per lane");
static_assert(NSUB - HA == 9, "phase B issues a batch of five and a batch of four exchanges");

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
__device__ __forceinline__ uint32_t ring_fwd(uint32_t off, uint32_t add, uint32_t size) {   // off, add < size
    const uint32_t r = off + add;
    return r >= size ? r - size : r;
}
__device__ __forceinline__ uint32_t ring_back(uint32_t off, uint32_t sub, uint32_t size) {  // off, sub < size
    return off >= sub ? off - sub : off + size - sub;
}
__device__ __forceinline__ uint32_t win4(const uint32_t *win32, uint32_t off) {             // 4 bytes at ring offset
    const uint32_t w0 = win32[off >> 2], w1 = win32[(off >> 2) + 1];
    return __builtin_amdgcn_alignbyte(w1, w0, off & 3);
}
__device__ __forceinline__ uint64_t win8(const uint32_t *win32, uint32_t off) {             // 8 bytes at ring offset
    const uint32_t i = off >> 2;
    const uint32_t w0 = win32[i], w1 = win32[i + 1], w2 = win32[i + 2];
    return (uint64_t)__builtin_amdgcn_alignbyte(w1, w0, off & 3) | (uint64_t)__builtin_amdgcn_alignbyte(w2, w1, off & 3) << 32;
}

// 16-bit exchanges, in order, one wait.  old[i] = the dword that held the field before.  A lane with mask 0 /
// value 0 leaves its dword untouched.
__device__ __forceinline__ void mskor4(uint32_t *old, const uint32_t *addr, const uint32_t *mask, const uint32_t *val) {
    asm volatile(
        "ds_mskor_rtn_b32 %0, %4, %8, %12\n\t"
        "ds_mskor_rtn_b32 %1, %5, %9, %13\n\t"
        "ds_mskor_rtn_b32 %2, %6, %10, %14\n\t"
        "ds_mskor_rtn_b32 %3, %7, %11, %15\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]),
          "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3])
        : "memory");
}
__device__ __forceinline__ void mskor5(uint32_t *old, const uint32_t *addr, const uint32_t *mask, const uint32_t *val) {
    asm volatile(
        "ds_mskor_rtn_b32 %0, %5, %10, %15\n\t"
        "ds_mskor_rtn_b32 %1, %6, %11, %16\n\t"
        "ds_mskor_rtn_b32 %2, %7, %12, %17\n\t"
        "ds_mskor_rtn_b32 %3, %8, %13, %18\n\t"
        "ds_mskor_rtn_b32 %4, %9, %14, %19\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3]), "=&v"(old[4])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(mask[0]), "v"(mask[1]), "v"(mask[2]),
          "v"(mask[3]), "v"(mask[4]), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4])
        : "memory");
}

}  // namespace m2

// flags[0] |= 1 when the head pass observed a lane-order violation (results are then discarded by the host).
// ablate: timing experiments only (LFX_ABLATE) — switches stages off, the results are then wrong.
__global__ __launch_bounds__(m2::THREADS) void lz77_match2_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint32_t max_len, uint32_t *__restrict__ md,
    uint32_t *__restrict__ flags, uint64_t *__restrict__ dbg, uint32_t ablate, uint32_t *__restrict__ deep,
    uint32_t deep_cap) {
    using namespace m2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *head32 = (uint32_t *)(smem + OFF_HEAD);
    uint16_t *prevd = (uint16_t *)(smem + OFF_PREVD);
    uint32_t *win32 = (uint32_t *)(smem + OFF_WIN);
    uint32_t *reqb = (uint32_t *)(smem + OFF_REQ);
    uint32_t *oldb = (uint32_t *)(smem + OFF_OLD);
    uint16_t *lk = (uint16_t *)(smem + OFF_LK);
    uint32_t *qb = (uint32_t *)(smem + OFF_Q);     // [helper][QCAP x {index, dist << 16 | next link}, counter]
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
    if (ch.flags & CH_LITERALS) return;           // NoCompressionLz77Encoder chunks never come here
    const uint32_t end = (n > 3 ? n : 3) - 3;     // default.rs:75
    const uint32_t q0 = sg.start;                 // first position answered by this segment
    const uint32_t q1 = min(sg.start + sg.len, end);
    if (q0 >= q1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;   // warm-up: link only
    const uint32_t base = l0 & ~3u;                               // tile origin (dword aligned)
    const int ntiles = (int)((q1 - base + TILE - 1) / TILE);
    const uint32_t n_pad = (n + 3) & ~3u;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    // ---- prologue: empty head table, empty queues, window bytes for P(0)
    for (uint32_t i = tid; i < (1u << (HASH_BITS - 1)); i += THREADS) {
        const uint32_t f = (base - HEAD_FAR) & 0xFFFFu;
        head32[i] = f | f << 16;
    }
    if (tid < 2) qb[tid * QWORDS + QCAP * 2] = 0;
    uint32_t loaded_to = base;                                    // window holds [.., loaded_to) (wave 0 keeps it)
    {
        const uint32_t need = min(base + TILE + 4, n_pad);
        for (uint32_t p = loaded_to + 4 * tid; p < need; p += 4 * THREADS) {
            const uint32_t v = src.load4(p), o = p - base;        // (first pass: no wrap)
            win32[o >> 2] = v;
            if (o < 8) win32[(WRING + o) >> 2] = v;
        }
        loaded_to = max(loaded_to, need);
    }
    lds_barrier();

    // resolver lane state carried from stage to stage: stage X of tile k and stage X+1 of the same tile run on the
    // same lane (index idx inside the tile), a phase or two later
    const uint32_t idx = (wave - 1) * 64 + lane;
    uint32_t key_p = 0, key_f = 0, key_r = 0;      // 3-byte prefix of the P / F / R tile position
    uint32_t key_q = 0, hh_q = 0;                  // (one tile of delay between P and F1: the head pass sits between)
    bool val_q = false;
    uint32_t hh_p = 0, hh_f = 0;                   // its hash (selects the half of the exchanged dword)
    bool val_p = false, val_f = false, val_r = false;   // position takes part in the chain structure (l0 <= p < q1)
    uint32_t cd_f = 0, cd_r = 0;                   // known answer distance (0 = walk)
    uint32_t e_f = 0, e_r = 0;                     // own final link distance (0 = none)
    uint32_t lk_f = 0;                             // F1 → F2: first link state
    uint32_t r_dist = 0;                           // R1 → R2
    bool r_found = false, r_deleg = false;
    // helper state (waves 14, 15): the walks of one tile, HE per lane
    uint32_t h_idx[HE], h_dist[HE], h_d[HE], h_key[HE];
    bool h_found[HE];
#pragma unroll
    for (uint32_t e = 0; e < HE; ++e) { h_idx[e] = 0xFFFFFFFFu; h_dist[e] = h_d[e] = h_key[e] = 0; h_found[e] = false; }
    // head wave: the request words of the pass in flight
    uint32_t rq[NSUB];
#pragma unroll
    for (uint32_t s = 0; s < NSUB; ++s) rq[s] = 0;
    // (tile 0 = position `base` sits at ring offset 0; the loop starts three tiles early)
    uint32_t wk = WRING - 3 * TILE, sk = PRING - 3 * TILE;   // window / prevd ring offset of tile `it`
    uint32_t fill_off = loaded_to - base;          // wave 0: ring offset of position loaded_to
    uint64_t cy_a = 0, cy_b = 0, cy_w = 0, last_a = 0, last_b = 0, cy_w1 = 0;

    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    else if (wave > RW) __builtin_amdgcn_s_setprio(2);   // (the helpers' dependent hops are the longest chain of a tile)

    // one hop of HE chain walks (helpers): returns false when none of the wavefront's walks is still going
    auto helper_hop = [&](uint32_t t_tile, uint32_t w_tile, uint32_t s_tile) -> bool {
        bool any = false;
#pragma unroll
        for (uint32_t e = 0; e < HE; ++e) any |= h_d[e] != 0;
        if (!__ballot(any)) return false;
        uint32_t kq[HE], dn[HE];
        bool ok[HE];
#pragma unroll
        for (uint32_t e = 0; e < HE; ++e) {
            const uint32_t w_pos = ring_fwd(w_tile, h_idx[e] & 1023u, WRING), s_pos = ring_fwd(s_tile, h_idx[e] & 1023u, PRING);
            ok[e] = false;
            uint32_t aw = w_pos, as = s_pos;
            if (h_d[e] != 0) {
                h_dist[e] += h_d[e];
                if (h_dist[e] > window || h_dist[e] > t_tile + h_idx[e]) h_d[e] = 0;         // default.rs:81
                else { ok[e] = true; aw = ring_back(w_pos, h_dist[e], WRING); as = ring_back(s_pos, h_dist[e], PRING); }
            }
            kq[e] = win4(win32, aw) & 0xFFFFFFu;
            dn[e] = prevd[as];
        }
#pragma unroll
        for (uint32_t e = 0; e < HE; ++e)
            if (ok[e]) { if (kq[e] == h_key[e]) { h_found[e] = true; h_d[e] = 0; } else h_d[e] = dn[e]; }
        return true;
    };

    for (int it = -3; it < ntiles + 2; ++it) {     // (+2: the helpers finish the last tiles' walks up to two iterations later)
        const uint64_t c0 = dbg ? clock64() : 0;
        const uint32_t w1 = ring_fwd(wk, TILE, WRING), s1 = ring_fwd(sk, TILE, PRING);   // tile it+1
        const uint32_t w2 = ring_fwd(w1, TILE, WRING);                                   // tile it+2
        const uint32_t w3 = ring_fwd(w2, TILE, WRING);                                   // tile it+3
        const bool do_r = it >= 0 && it < ntiles;
        const bool do_f = it + 1 >= 0 && it + 1 < ntiles;
        const bool do_h = it + 2 >= 0 && it + 2 < ntiles;
        const bool do_p = it + 3 >= 0 && it + 3 < ntiles;
        const uint32_t t_r = base + (uint32_t)it * TILE;            // only used when do_r
        const uint32_t t_f = base + (uint32_t)(it + 1) * TILE;      // only used when do_f
        const uint32_t t_h = base + (uint32_t)(it + 2) * TILE;      // only used when do_h
        const uint32_t t_p = base + (uint32_t)(it + 3) * TILE;      // only used when do_p
        uint32_t fill_w0[FILL_LOADS], fill_w1[FILL_LOADS];
        uint32_t fill_need = loaded_to;
        uint32_t *oldh = oldb + ((uint32_t)(it + 2) & 1) * TILE;    // what H(it+2) writes
        const uint32_t *oldf = oldb + ((uint32_t)(it + 1) & 1) * TILE;   // what F1(it+1) reads

        // =================================================== phase A
        if (wave == 0) {
            // ---- window bytes: global loads now, LDS stores in phase B behind the head pass.  Five tiles ahead of R: P reads
            //      tile it+3 in phase B of the NEXT iteration's predecessor, match lengths read 258 bytes past the segment.
            fill_need = max(loaded_to, min(base + (uint32_t)(it + 5) * TILE + 4, n_pad));
#pragma unroll
            for (uint32_t q = 0; q < FILL_LOADS; ++q) {
                const uint32_t p = loaded_to + 4 * lane + 256 * q;
                fill_w0[q] = fill_w1[q] = 0;
                if (p < fill_need) src.load_raw(p, fill_w0[q], fill_w1[q]);
            }
            if (do_h) {
                // ---- the request words of H(it+2) (written by P in the last phase B): all of them now, the buffer is reused
#pragma unroll
                for (uint32_t s = 0; s < NSUB; ++s) rq[s] = reqb[s * 64 + lane];
                // ---- incremental sweep: stale fields (older than the window) → "far"
                const uint32_t slice = (uint32_t)(it + 2) % SWEEP_SLICES;
                const uint32_t far = (t_h - HEAD_FAR) & 0xFFFFu;
                constexpr uint32_t PER = (1u << (HASH_BITS - 1)) / SWEEP_SLICES / 64;
                uint32_t hw[PER];
#pragma unroll
                for (uint32_t q = 0; q < PER; ++q) hw[q] = head32[slice * (PER * 64) + q * 64 + lane];
#pragma unroll
                for (uint32_t q = 0; q < PER; ++q) {
                    uint32_t lo = hw[q] & 0xFFFFu, hi = hw[q] >> 16;
                    const uint32_t dlo = (t_h - lo) & 0xFFFFu, dhi = (t_h - hi) & 0xFFFFu;
                    if (dlo == 0 || dlo > MAX_WINDOW) lo = far;
                    if (dhi == 0 || dhi > MAX_WINDOW) hi = far;
                    head32[slice * (PER * 64) + q * 64 + lane] = lo | hi << 16;
                }
                // ---- H(it+2), part 1: the first HA exchanges, in position order
                uint32_t old[HA], addr[HA], mask[HA], val[HA];
#pragma unroll
                for (uint32_t s = 0; s < HA; ++s) {
                    const uint32_t sh = (rq[s] >> 13) & 16u;                 // (hash & 1) * 16
                    addr[s] = head_lds + ((rq[s] >> 18) << 2);               // dword of field hash
                    mask[s] = (0u - ((rq[s] >> 16) & 1u)) & (0xFFFFu << sh);
                    val[s] = (rq[s] & 0xFFFFu) << sh;
                }
                mskor4(old, addr, mask, val);
#pragma unroll
                for (uint32_t s = 0; s < HA; ++s) oldh[s * 64 + lane] = old[s];
            }
        } else if (wave <= RW) {
            const uint32_t p_r = t_r + idx;
            const uint32_t w_pos_r = ring_fwd(wk, idx, WRING), s_pos_r = ring_fwd(sk, idx, PRING);
            const uint32_t w_pos_f = ring_fwd(w1, idx, WRING), s_pos_f = ring_fwd(s1, idx, PRING);
            const bool act_r = do_r && val_r && p_r >= q0;       // (val_r implies p_r < q1)
            const bool act_f = do_f && val_f;
            // ---- R1(it): chain walk, only where the answer is not already known (cd) — and then starting at the LINK of
            //      the raw predecessor, which is known to carry another prefix
            uint32_t dist = 0, d = 0;
            bool found = false, walk = false;
            if (act_r) {
                if (cd_r) { dist = cd_r; found = dist <= window; }
                else if (e_r != 0 && e_r <= window && !(ablate & 1)) { dist = e_r; walk = true; }
            }
            // -- step 0: loads
            const uint32_t d0 = prevd[walk ? ring_back(s_pos_r, dist, PRING) : s_pos_r];
            const uint32_t ow = oldf[idx];
            // -- step 0: uses
            d = walk ? d0 : 0u;
            uint32_t d_f = 0;
            if (act_f) {
                const uint32_t of = (hh_f & 1) ? ow >> 16 : ow & 0xFFFFu;   // what the exchange returned for this field
                d_f = (t_f + idx - of) & 0xFFFFu;
            }
            if (__ballot(d_f >= FUTURE) && lane == 0) atomicOr(flags, 1u);   // lane-order violation (never observed)
            if (d_f > MAX_WINDOW || (ablate & 32)) d_f = 0;
            // -- step 1: loads (R1's one in-line hop, F1's predecessor)
            bool ok = false;
            uint32_t aw = w_pos_r, as = s_pos_r;
            if (d != 0) {
                dist += d;
                if (dist > window || dist > p_r) d = 0;              // default.rs:81 (inclusive window)
                else { ok = true; aw = ring_back(w_pos_r, dist, WRING); as = ring_back(s_pos_r, dist, PRING); }
            }
            const uint32_t kq = win4(win32, aw) & 0xFFFFFFu;
            const uint32_t dn = prevd[as];
            const uint32_t kqf = win4(win32, d_f ? ring_back(w_pos_f, d_f, WRING) : w_pos_f) & 0xFFFFFFu;
            const uint32_t pqf = prevd[d_f > idx ? ring_back(s_pos_f, d_f, PRING) : s_pos_f];   // older tile: final
            // -- step 1: uses
            if (ok) { if (kq == key_r) { found = true; d = 0; } else d = dn; }
            // F1(it+1): raw predecessor → known answer / first link state
            cd_f = 0;
            {
                uint32_t e = d_f;                                    // plain link (or none)
                if (d_f != 0 && kqf == key_f) {
                    cd_f = d_f;
                    if (d_f > idx) { e = pqf ? d_f + pqf : 0; if (e > MAX_WINDOW) e = 0; }
                    else e = LK_PTR + (idx - d_f);                   // in this tile: inherit by pointer jumping
                }
                lk_f = e;
                if (act_f) lk[idx] = (uint16_t)e;
            }
            // -- walks that are still going: hand them to this tile's helper (queue full: the owner walks on below)
            bool deleg = false;
            {
                uint32_t *q = qb + ((uint32_t)it & 1) * QWORDS;
                const bool still = d != 0;
                const uint64_t bm = __ballot(still);
                if (bm) {
                    uint32_t slot0 = 0;
                    if (lane == 0) slot0 = atomicAdd(&q[QCAP * 2], (uint32_t)__popcll(bm));
                    slot0 = __builtin_amdgcn_readfirstlane(slot0);
                    const uint32_t slot = slot0 + __popcll(bm & lt_mask);
                    if (still && slot < QCAP) {
                        q[slot * 2] = idx;
                        q[slot * 2 + 1] = dist << 16 | d;
                        deleg = true;
                        d = 0;
                    }
                }
            }
            while (__ballot(d != 0)) {
                bool ok2 = false;
                uint32_t aw2 = w_pos_r, as2 = s_pos_r;
                if (d != 0) {
                    dist += d;
                    if (dist > window || dist > p_r) d = 0;
                    else { ok2 = true; aw2 = ring_back(w_pos_r, dist, WRING); as2 = ring_back(s_pos_r, dist, PRING); }
                }
                const uint32_t kq2 = win4(win32, aw2) & 0xFFFFFFu;
                const uint32_t dn2 = prevd[as2];
                if (ok2) { if (kq2 == key_r) { found = true; d = 0; } else d = dn2; }
            }
            r_dist = dist; r_found = found; r_deleg = deleg;
        } else {
            // ---- helpers, phase A: the tile whose walks this helper started in the last phase B goes on (at most six
            //      hops now, the rest in phase B); the one it finished walking in the last phase B gets its match lengths
            const uint32_t hid = wave - (RW + 1);
            if ((((uint32_t)(it - 1)) & 1) == hid) {
                const uint32_t t_t = base + (uint32_t)(it - 1) * TILE;
                const uint32_t w_t = ring_back(wk, TILE, WRING), s_t = ring_back(sk, TILE, PRING);
                for (int k = 0; k < 4; ++k) if (!helper_hop(t_t, w_t, s_t)) break;
            } else {
                const uint32_t t_t = base + (uint32_t)(it - 2) * TILE;
                const uint32_t w_t = ring_back(ring_back(wk, TILE, WRING), TILE, WRING);
                uint32_t l[HE], lim[HE], oa[HE], ob[HE];
                bool cmp[HE], any_found = false;
#pragma unroll
                for (uint32_t e = 0; e < HE; ++e) {
                    l[e] = 0; lim[e] = 0;
                    oa[e] = ring_fwd(ring_fwd(w_t, h_idx[e] & 1023u, WRING), 3, WRING);
                    ob[e] = oa[e];
                    if (h_found[e]) {
                        lim[e] = n - (t_t + h_idx[e] + 3);
                        if (lim[e] > max_len - 3) lim[e] = max_len - 3;
                        ob[e] = ring_back(oa[e], h_dist[e], WRING);
                    }
                    cmp[e] = h_found[e] && lim[e] != 0;
                    any_found |= h_found[e];
                }
                if (__ballot(any_found)) {
#pragma unroll
                    for (int step = 0; step < 2; ++step) {
                        uint64_t xa[HE], xb[HE];
#pragma unroll
                        for (uint32_t e = 0; e < HE; ++e) { xa[e] = win8(win32, oa[e]); xb[e] = win8(win32, ob[e]); }
#pragma unroll
                        for (uint32_t e = 0; e < HE; ++e) {
                            if (cmp[e]) {
                                const uint64_t x = xa[e] ^ xb[e];
                                if (x) { l[e] += (uint32_t)__builtin_ctzll(x) >> 3; cmp[e] = false; }
                                else {
                                    l[e] += 8;
                                    oa[e] += 8; if (oa[e] >= WRING) oa[e] -= WRING;
                                    ob[e] += 8; if (ob[e] >= WRING) ob[e] -= WRING;
                                    if (l[e] >= lim[e]) cmp[e] = false;
                                }
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t e = 0; e < HE; ++e) {
                        uint64_t lm = __ballot(cmp[e]);
                        while (lm) {
                            const uint32_t sl = (uint32_t)__builtin_ctzll(lm);
                            lm &= lm - 1;
                            const uint32_t boa = __builtin_amdgcn_readlane(oa[e], sl), bob = __builtin_amdgcn_readlane(ob[e], sl);
                            const uint32_t blim = __builtin_amdgcn_readlane(lim[e], sl);
                            const uint32_t off = 4 * lane;
                            uint32_t x = 0;
                            if (16 + off < blim) {
                                uint32_t a = boa + off, b = bob + off;
                                if (a >= WRING) a -= WRING;
                                if (b >= WRING) b -= WRING;
                                x = win4(win32, a) ^ win4(win32, b);
                            }
                            const uint64_t mis = __ballot(x != 0);
                            uint32_t res = blim;
                            if (mis) {
                                const uint32_t fl = (uint32_t)__builtin_ctzll(mis);
                                const uint32_t cand = 16 + off + ((uint32_t)__builtin_ctz(x | 0x80000000u) >> 3);
                                res = __builtin_amdgcn_readlane(cand, fl);
                            }
                            if (lane == sl) l[e] = res;
                        }
                    }
                }
#pragma unroll
                for (uint32_t e = 0; e < HE; ++e) {
                    if (h_idx[e] != 0xFFFFFFFFu) {                   // a queue entry: this lane owns the position's md word
                        uint32_t word = 0;
                        if (h_found[e]) {
                            if (l[e] > lim[e]) l[e] = lim[e];
                            word = ((3 + l[e]) << 16) | h_dist[e];
                        }
                        md[ch.in_off + t_t + h_idx[e]] = word;
                    }
                    h_idx[e] = 0xFFFFFFFFu; h_found[e] = false; h_d[e] = 0;
                }
            }
        }
        const uint64_t c1 = dbg ? clock64() : 0;
        lds_barrier();
        const uint64_t c1b = dbg ? clock64() : 0;
        // =================================================== phase B
        if (wave == 0) {
            if (do_h) {
                // ---- H(it+2), part 2: the remaining exchanges — a batch of five, a batch of four
                {
                    uint32_t old[5], addr[5], mask[5], val[5];
#pragma unroll
                    for (uint32_t s = 0; s < 5; ++s) {
                        const uint32_t r = rq[HA + s];
                        const uint32_t sh = (r >> 13) & 16u;
                        addr[s] = head_lds + ((r >> 18) << 2);
                        mask[s] = (0u - ((r >> 16) & 1u)) & (0xFFFFu << sh);
                        val[s] = (r & 0xFFFFu) << sh;
                    }
                    mskor5(old, addr, mask, val);
#pragma unroll
                    for (uint32_t s = 0; s < 5; ++s) oldh[(HA + s) * 64 + lane] = old[s];
                }
                {
                    uint32_t old[4], addr[4], mask[4], val[4];
#pragma unroll
                    for (uint32_t s = 0; s < 4; ++s) {
                        const uint32_t r = rq[HA + 5 + s];
                        const uint32_t sh = (r >> 13) & 16u;
                        addr[s] = head_lds + ((r >> 18) << 2);
                        mask[s] = (0u - ((r >> 16) & 1u)) & (0xFFFFu << sh);
                        val[s] = (r & 0xFFFFu) << sh;
                    }
                    mskor4(old, addr, mask, val);
#pragma unroll
                    for (uint32_t s = 0; s < 4; ++s) oldh[(HA + 5 + s) * 64 + lane] = old[s];
                }
            }
            // ---- window stores (their loads were issued in phase A)
            const uint32_t sh = (uint32_t)src.shift;
#pragma unroll
            for (uint32_t q = 0; q < FILL_LOADS; ++q) {
                const uint32_t p = loaded_to + 4 * lane + 256 * q;
                if (p < fill_need) {
                    const uint32_t v = __builtin_amdgcn_alignbyte(fill_w1[q], fill_w0[q], sh);
                    const uint32_t o = ring_fwd(fill_off, 4 * lane + 256 * q, WRING);
                    win32[o >> 2] = v;
                    if (o < 8) win32[(WRING + o) >> 2] = v;
                }
            }
            fill_off = ring_fwd(fill_off, fill_need - loaded_to, WRING);
            loaded_to = fill_need;
        } else if (wave <= RW) {
            const uint32_t p_r = t_r + idx;
            const uint32_t w_pos_r = ring_fwd(wk, idx, WRING);
            const bool act_r = do_r && val_r && p_r >= q0;
            const bool act_f = do_f && val_f;
            // ---- F2(it+1): inherit the link of the same-prefix predecessor (pointer jumping, no ordering needed: every
            //      state a reader can observe is valid and the oldest member of a run is final from the start)
            // ---- R2(it): longest_common_prefix (default.rs:122-129): 8 bytes per step for the first 16
            // ---- P(it+3): request word of the head pass: hash << 17 | valid << 16 | low 16 bits of the position
            uint32_t e = act_f ? lk_f : 0u;
            if ((ablate & 4) && e >= LK_PTR) e = 0;
            const bool found = act_r && r_found;
            const uint32_t dist = r_dist;
            uint32_t l = 0, lim = 0;
            uint32_t oa = ring_fwd(w_pos_r, 3, WRING), ob = oa;
            if (found) {
                lim = n - (p_r + 3);                                   // bounded by the end of the chunk
                if (lim > max_len - 3) lim = max_len - 3;
                ob = ring_back(oa, dist, WRING);
            }
            bool cmp = found && lim != 0 && !(ablate & 2);             // still comparing
            const uint32_t kp_raw = win4(win32, ring_fwd(w3, idx, WRING));
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                // loads
                const bool ptr = e >= LK_PTR;
                const uint32_t j = ptr ? e - LK_PTR : idx;
                const uint32_t eq = lk[j];
                const uint64_t xa = win8(win32, oa), xb = win8(win32, ob);
                // uses
                if (ptr) {
                    if (eq < LK_PTR) { e = eq ? (idx - j) + eq : 0; if (e > MAX_WINDOW) e = 0; }
                    else e = eq;
                    lk[idx] = (uint16_t)e;
                }
                if (cmp) {
                    const uint64_t x = xa ^ xb;
                    if (x) { l += (uint32_t)__builtin_ctzll(x) >> 3; cmp = false; }
                    else {
                        l += 8;
                        oa += 8; if (oa >= WRING) oa -= WRING;
                        ob += 8; if (ob >= WRING) ob -= WRING;
                        if (l >= lim) cmp = false;
                    }
                }
                if (step == 0) {
                    const uint32_t p_p = t_p + idx;
                    val_p = do_p && p_p >= l0 && p_p < q1;
                    key_p = kp_raw & 0xFFFFFFu;
                    hh_p = hash3(key_p);
                    if (do_p) reqb[idx] = val_p ? (hh_p << 17) | 0x10000u | (p_p & 0xFFFFu) : 0u;
                }
            }
            while (__ballot(e >= LK_PTR)) {
                if (e >= LK_PTR) {
                    const uint32_t j = e - LK_PTR;
                    const uint32_t eq = lk[j];
                    if (eq < LK_PTR) { e = eq ? (idx - j) + eq : 0; if (e > MAX_WINDOW) e = 0; }
                    else e = eq;
                    lk[idx] = (uint16_t)e;
                }
            }
            e_f = e;
            if (act_f) prevd[ring_fwd(s1, idx, PRING)] = (uint16_t)e;
            // a lane still matching after 16 bytes gets the whole wavefront: lane j compares bytes
            // [16+4j, 16+4j+4) — one step settles up to 256 more bytes
            uint64_t lm = __ballot(cmp);                               // (cmp here ⇒ l == 16 < lim)
            while (lm) {
                const uint32_t sl = (uint32_t)__builtin_ctzll(lm);
                lm &= lm - 1;
                const uint32_t boa = __builtin_amdgcn_readlane(oa, sl), bob = __builtin_amdgcn_readlane(ob, sl);
                const uint32_t blim = __builtin_amdgcn_readlane(lim, sl);
                const uint32_t off = 4 * lane;
                uint32_t x = 0;
                if (16 + off < blim) {
                    uint32_t a = boa + off, b = bob + off;
                    if (a >= WRING) a -= WRING;
                    if (b >= WRING) b -= WRING;
                    x = win4(win32, a) ^ win4(win32, b);
                }
                const uint64_t mis = __ballot(x != 0);
                uint32_t res = blim;
                if (mis) {
                    const uint32_t fl = (uint32_t)__builtin_ctzll(mis);
                    const uint32_t cand = 16 + off + ((uint32_t)__builtin_ctz(x | 0x80000000u) >> 3);
                    res = __builtin_amdgcn_readlane(cand, fl);
                }
                if (lane == sl) l = res;
            }
            uint32_t word = 0;
            if (found) {
                if (l > lim) l = lim;
                word = ((3 + l) << 16) | dist;
            }
            if (act_r && !r_deleg && !(ablate & 16)) md[ch.in_off + p_r] = word;
        } else {
            // ---- helpers, phase B: this tile's helper takes the walks the resolvers just handed over (one per lane and
            //      entry) and starts on them; the other helper walks its tile (it-1) to the end
            const uint32_t hid = wave - (RW + 1);
            if ((((uint32_t)it) & 1) == hid) {
                uint32_t *q = qb + hid * QWORDS;
                const uint32_t cnt = do_r ? min(q[QCAP * 2], QCAP) : 0u;
#pragma unroll
                for (uint32_t e = 0; e < HE; ++e) {
                    const uint32_t slot = e * 64 + lane;
                    const bool v = slot < cnt;
                    const uint32_t qi = v ? q[slot * 2] : 0u, qs = v ? q[slot * 2 + 1] : 0u;
                    h_idx[e] = v ? qi : 0xFFFFFFFFu;
                    h_key[e] = win4(win32, ring_fwd(wk, qi, WRING)) & 0xFFFFFFu;
                    h_dist[e] = qs >> 16; h_d[e] = qs & 0xFFFFu;
                    h_found[e] = false;
                }
                if (cnt && lane == 0) q[QCAP * 2] = 0;
                if (cnt) for (int k = 0; k < 3; ++k) if (!helper_hop(t_r, wk, sk)) break;
            } else {
                const uint32_t t_t = base + (uint32_t)(it - 1) * TILE;
                const uint32_t w_t = ring_back(wk, TILE, WRING), s_t = ring_back(sk, TILE, PRING);
                bool more = true;
                for (int k = 0; k < 4 && more; ++k) more = helper_hop(t_t, w_t, s_t);
                if (more) {
                    // walks deeper than twelve hops (0.03 % of the positions of text, but one in eight tiles holds one): off
                    // to the deep list — lz77_deep_kernel settles them by a backward scan of the input, one wavefront each
                    // — so that no tile ever waits for the deepest chain of a crowded bucket
#pragma unroll
                    for (uint32_t e = 0; e < HE; ++e) {
                        const bool sp = h_d[e] != 0;
                        const uint64_t bm = __ballot(sp);
                        if (bm) {
                            uint32_t slot0 = 0;
                            if (lane == 0) slot0 = atomicAdd(&deep[0], (uint32_t)__popcll(bm));
                            slot0 = __builtin_amdgcn_readfirstlane(slot0);
                            const uint32_t slot = slot0 + __popcll(bm & lt_mask);
                            if (sp && slot < deep_cap) {
                                deep[2 + 2 * slot] = sg.chunk;
                                deep[3 + 2 * slot] = t_t + h_idx[e];
                                h_idx[e] = 0xFFFFFFFFu; h_d[e] = 0; h_found[e] = false;
                            }
                        }
                    }
                    while (helper_hop(t_t, w_t, s_t)) {}    // (list full: walk on)
                }
            }
        }
        const uint64_t c2 = dbg ? clock64() : 0;
        // ---- rotate the stage registers (P → [head pass] → F → R)
        key_r = key_f; key_f = key_q; key_q = key_p;
        hh_f = hh_q; hh_q = hh_p;
        val_r = val_f; val_f = val_q; val_q = val_p;
        cd_r = cd_f; e_r = e_f;
        wk = w1; sk = s1;
        lds_barrier();
        const uint64_t c3 = dbg ? clock64() : 0;
        cy_a += c1 - c0; cy_b += c2 - c1b; cy_w += c3 - c2; cy_w1 += c1b - c1;
        if (c1b - c1 < 150) last_a++;
        if (c3 - c2 < 150) last_b++;
    }
    if (dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_a; d[1] = cy_b; d[2] = cy_w; d[3] = last_a; d[4] = last_b; d[5] = (uint64_t)ntiles; d[6] = cy_w1;
    }
}

// The walks the helpers gave up on: deep[0] = count, then {chunk, position} pairs.  One wavefront per entry scans the
// input backwards from the position, 64 candidates per step, for the most recent earlier occurrence of the position's
// 3-byte prefix inside the window (default.rs:78-81: the table holds the most recent occurrence in the chunk; one that
// lies further back than the window is rejected, so nothing beyond the window needs to be looked at), then settles the
// match length like the main kernel does for a long match (256 bytes per step).
__global__ __launch_bounds__(256) void lz77_deep_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes,
                                                        const ChunkDesc *__restrict__ chunks, uint32_t window,
                                                        uint32_t max_len, uint32_t *__restrict__ md,
                                                        const uint32_t *__restrict__ deep, uint32_t deep_cap) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t count = min(deep[0], deep_cap);
    const gptr_u32 w = (gptr_u32)((uint64_t)in & ~3ull);
    const int64_t shift0 = (int64_t)((uint64_t)in & 3), last = (int64_t)((in_bytes + (uint64_t)shift0 + 3) >> 2);
    for (uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6); i < count; i += gridDim.x * 4) {
        const ChunkDesc ch = chunks[deep[2 + 2 * i]];
        const uint32_t p = deep[3 + 2 * i];
        const uint8_t *b = in + ch.in_off;
        const uint32_t n = (uint32_t)ch.len;
        const uint32_t key = (uint32_t)b[p] | (uint32_t)b[p + 1] << 8 | (uint32_t)b[p + 2] << 16;
        const uint32_t reach = min(p, window);                // candidates p-1 ... p-reach
        // byte addresses relative to the 4-byte aligned base of `in`: the position itself, and the aligned end of the scan
        const int64_t P = (int64_t)ch.in_off + (int64_t)p + shift0;
        const int64_t A = (P + 3) & ~3ll;
        uint32_t dist = 0;
        // 4096 candidate bytes per step, backwards: lane j takes 16 aligned dwords (64 positions; higher lane = closer)
        for (int64_t hi = A; dist == 0 && P - hi < (int64_t)reach; hi -= 4096) {
            const int64_t d0 = ((hi - 4096) >> 2) + 16 * (int64_t)lane;      // first dword of this lane
            uint32_t ww[17];
#pragma unroll
            for (int k = 0; k < 17; ++k) {
                const int64_t di = d0 + k;
                ww[k] = (di >= 0 && di < last) ? w[di] : 0u;
            }
            int64_t best = -1;                                  // highest matching position (address form), -1 = none
#pragma unroll
            for (int k = 0; k < 16; ++k) {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const uint32_t tri = __builtin_amdgcn_alignbyte(ww[k + 1], ww[k], (uint32_t)o) & 0xFFFFFFu;
                    const int64_t pos = (d0 + k) * 4 + o;
                    const int64_t dd = P - pos;
                    if (tri == key && dd >= 1 && dd <= (int64_t)reach) best = pos;   // ascending positions: keeps the closest
                }
            }
            const uint64_t m = __ballot(best >= 0);
            if (m) {
                const uint32_t src_lane = 63u - (uint32_t)__builtin_clzll(m);       // highest lane = closest group
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)(P - best), src_lane);
                dist = lo;
            }
        }
        uint32_t word = 0;
        if (dist) {
            uint32_t lim = n - (p + 3);
            if (lim > max_len - 3) lim = max_len - 3;
            uint32_t l = lim;
            for (uint32_t o = 0; o < lim; o += 64) {
                const uint32_t k = o + lane;
                const bool mis = k < lim && b[p + 3 + k] != b[p + 3 + k - dist];
                const uint64_t m = __ballot(mis);
                if (m) { l = o + (uint32_t)__builtin_ctzll(m); break; }
            }
            word = ((3 + l) << 16) | dist;
        }
        if (lane == 0) md[ch.in_off + p] = word;
    }
}

int launch_match2(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint32_t max_len, uint32_t *md, uint32_t *flags, uint32_t *deep,
                  uint32_t deep_cap, uint64_t *dbg) {
    if (nsegs == 0) return 0;
    // (LFX_ABLATE: timing experiments only — switches stages off, the results are then wrong)
    static const uint32_t ablate = getenv("LFX_ABLATE") ? (uint32_t)strtoul(getenv("LFX_ABLATE"), nullptr, 0) : 0u;
    const size_t lds = m2::LDS_BYTES;
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)lz77_match2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev_ & 63] = true;
    }
    hipLaunchKernelGGL(lz77_match2_kernel, dim3(nsegs), dim3(m2::THREADS), lds, st, in, in_bytes, chunks, segs, window, max_len,
                       md, flags, dbg, ablate, deep, deep_cap);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return (int)e_;
    // the walks the helpers gave up on (deep[0] = how many; more than deep_cap never happens: the helpers then walk on)
    hipLaunchKernelGGL(lz77_deep_kernel, dim3(2048), dim3(256), 0, st, in, in_bytes, chunks, window, max_len, md, deep, deep_cap);
    e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

}  // namespace lfx
