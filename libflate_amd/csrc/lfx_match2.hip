// lfx_match2.hip — second-generation LZ77 stage for gfx950: match finding AND (when a workgroup owns a
// whole chunk) the greedy parse, without the per-position intermediate ever leaving the chip.
//
// Replaces, bit for bit, DefaultLz77Encoder::flush (libflate_lz77/src/default.rs:69-109):
//   PrefixTable::insert            default.rs:146-182   → hash buckets with exact 3-byte verification
//   longest_common_prefix          default.rs:122-129   → 8 bytes per step per lane, 256 per step per wavefront
//   the walk `i += length / i += 1` default.rs:76-107   → one walker wavefront per workgroup (fused mode)
//
// Parse independence (tests/test_host_pipeline.py): the reference inserts EVERY position < end exactly once
// and in order (default.rs:78,92-97), so cand(i) = max{ j < i : buf[j..j+3] == buf[i..i+3] } does not depend
// on which positions the walk visits.
//
// One workgroup of 16 wavefronts per segment, a four-stage software pipeline over tiles of 896 positions
// (14 resolver wavefronts x 64 lanes), two LDS-only barriers per tile:
//
//   wave 0      H(k+2)  ordered head pass.  head[] holds 2^14 16-bit fields (low 16 bits of the most recent
//                       position per hash) packed two per dword; ONE ds_mskor_rtn_b32 per 64 positions
//                       exchanges the field and returns the old word.  The LDS serves the lanes of one
//                       instruction that hit the same field in ascending lane order and a wavefront's
//                       instructions in issue order (measured: tools/exp/mskor_test.hip, 0 violations in
//                       1.3 M conflicting operations), so every lane receives exactly its raw predecessor
//                       ph(p) = most recent earlier position with the same hash.  A lane that observes a
//                       value "from the future" (distance >= 65536-64) proves a violation: the kernel raises
//                       a flag and the host re-runs the first-generation kernel.  Also: window fill (global
//                       loads issued one phase before their LDS stores) and an incremental sweep of stale
//                       head fields (a sixteenth of the table per tile) so 16-bit distances never alias.
//   waves 1..14 phase A: F1(k+1) raw predecessor → same prefix? (answer known, cd) : plain link ; first link
//                                state lk[] ; R1(k) chain walk for the positions whose answer is not known
//               phase B: F2(k+1) duplicate-collapsed link by pointer jumping → prevd[] ; R2(k) match length
//                                → code word (fused) or md[] (segments of a larger chunk)
//   wave 15     W(k-1)  fused mode: the greedy walk over the code words of the previous tile (64 answers in a
//                       VGPR, two steps per readlane chain) + compaction of the visited ones into codes[].
//
// Duplicate collapsing (exactness argument as in the first-generation kernel): link(p) = ph(p)
// if the prefixes differ, else link(ph(p)); the chain from p therefore visits the most recent member of every
// run of equal prefixes in its bucket, in decreasing position order, and the walk stops at the first exact
// 3-byte match (the most recent occurrence) or when the distance exceeds the window (default.rs:81).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfx_common.h"
#include "lfx_device.h"

namespace lfx {

namespace m2 {

constexpr int THREADS = 1024;
constexpr uint32_t RW = 14;                   // resolver wavefronts
constexpr uint32_t TILE = RW * 64;            // 896 positions
constexpr uint32_t NSUB = RW;                 // 64-position sub-tiles per tile
constexpr int HASH_BITS = 14;
constexpr uint32_t PRING = 32768 + 2048;      // prevd ring (entries)  >= window + TILE
constexpr uint32_t WRING = 36864;             // window ring (bytes)   >= window + 4 TILE + 4, and the fill of
                                              // tile k+4 must not touch what R(k) reads: 4 TILE + 4 <= WRING - 32768
constexpr uint32_t HEAD_FAR = 33000;          // distance marker of an empty / swept head field
constexpr uint32_t SWEEP_SLICES = 32;         // the whole table is swept every 32 tiles (28672 positions)
constexpr uint32_t FUTURE = 65536 - 64;       // a distance this large can only come from a lane-order violation

constexpr uint32_t LK_PTR = 32769;            // lk value >= LK_PTR: inherit the link of in-tile index (v - LK_PTR)

// LDS layout (bytes)
constexpr uint32_t OFF_HEAD = 0;                                   // 8192 dwords
constexpr uint32_t OFF_PREVD = OFF_HEAD + (2u << HASH_BITS);       // PRING u16
constexpr uint32_t OFF_WIN = OFF_PREVD + PRING * 2;                // WRING + 8 bytes (+ pad)
constexpr uint32_t OFF_REQ = OFF_WIN + WRING + 16;                 // TILE u32: head-pass requests (hash, valid, position)
constexpr uint32_t OFF_OLD = OFF_REQ + TILE * 4;                   // TILE u32: the dwords the exchanges returned
constexpr uint32_t OFF_LK = OFF_OLD + TILE * 4;                    // TILE u16: link states of the tile being finalized
constexpr uint32_t OFF_TABE = OFF_LK + TILE * 2;                   // fused: per group and entry lane, where the walk leaves the group
constexpr uint32_t OFF_TABM = OFF_TABE + TILE * 2;                 // fused: ... and the 64-bit set of positions it visits
constexpr uint32_t OFF_RES = OFF_TABM + TILE * 8;                  // fused: per group {visited mask, codes before the group}
constexpr uint32_t LDS_BYTES = OFF_RES + NSUB * 16;
static_assert(NSUB == 14, "the head pass issues two batches of seven exchanges");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(4 * TILE + 4 <= WRING - 32768, "window ring slack");
static_assert(TILE <= PRING - 32768, "prevd ring slack");
static_assert(SWEEP_SLICES * TILE + 32768 + TILE + 64 < FUTURE, "head ages must stay below the violation zone");
static_assert(HEAD_FAR + SWEEP_SLICES * TILE + TILE < FUTURE && HEAD_FAR > 32768, "far marker range");

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
    __device__ __forceinline__ uint32_t load1(uint64_t off) const {
        const uint64_t a = off + shift;
        return (w[a >> 2] >> (((uint32_t)a & 3) * 8)) & 0xFF;
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
__device__ __forceinline__ uint64_t lanemask_lt() {
    const uint32_t lane = __lane_id();
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// seven 16-bit exchanges, in order, one wait.  old[i] = the dword that held the field before.
// A lane with mask 0 / value 0 leaves its dword untouched.
__device__ __forceinline__ void mskor7(uint32_t (&old)[7], const uint32_t (&addr)[7], const uint32_t (&mask)[7],
                                       const uint32_t (&val)[7]) {
    asm volatile(
        "ds_mskor_rtn_b32 %0, %7, %14, %21\n\t"
        "ds_mskor_rtn_b32 %1, %8, %15, %22\n\t"
        "ds_mskor_rtn_b32 %2, %9, %16, %23\n\t"
        "ds_mskor_rtn_b32 %3, %10, %17, %24\n\t"
        "ds_mskor_rtn_b32 %4, %11, %18, %25\n\t"
        "ds_mskor_rtn_b32 %5, %12, %19, %26\n\t"
        "ds_mskor_rtn_b32 %6, %13, %20, %27\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3]), "=&v"(old[4]), "=&v"(old[5]), "=&v"(old[6])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]),
          "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]),
          "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6])
        : "memory");
}

}  // namespace m2

// FUSED: the segment is a whole chunk; wave 15 walks and emits the chunk's code words (codes, ncodes).
// !FUSED: per-position answers go to md[] (the parse kernels of lfx_encode_kernels.hip take over).
// flags[0] |= 1 when the head pass observed a lane-order violation (results are then discarded by the host).
template <bool FUSED>
__global__ __launch_bounds__(m2::THREADS) void lz77_match2_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint32_t max_len, uint32_t *__restrict__ md,
    uint32_t *__restrict__ codes, uint32_t *__restrict__ ncodes, uint32_t *__restrict__ flags, uint64_t *__restrict__ dbg) {
    using namespace m2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *head32 = (uint32_t *)(smem + OFF_HEAD);
    uint16_t *prevd = (uint16_t *)(smem + OFF_PREVD);
    uint32_t *win32 = (uint32_t *)(smem + OFF_WIN);
    uint32_t *reqb = (uint32_t *)(smem + OFF_REQ);
    uint32_t *oldb = (uint32_t *)(smem + OFF_OLD);
    uint16_t *lk = (uint16_t *)(smem + OFF_LK);
    uint16_t *tabE = (uint16_t *)(smem + OFF_TABE);
    uint2 *tabM = (uint2 *)(smem + OFF_TABM);
    uint32_t *resw = (uint32_t *)(smem + OFF_RES);   // per group: mask lo, mask hi, codes before, unused
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
    uint32_t *out = FUSED ? codes + ch.code_off : nullptr;
    if (ch.flags & CH_LITERALS) return;           // NoCompressionLz77Encoder chunks never come here
    const uint32_t end = (n > 3 ? n : 3) - 3;     // default.rs:75
    const uint32_t q0 = FUSED ? 0u : sg.start;    // first position answered by this segment
    const uint32_t q1 = FUSED ? end : min(sg.start + sg.len, end);
    if (q0 >= q1) {
        if (FUSED && wave == 0) {
            // n <= 3: every byte is a literal (default.rs:105-107), then the block's EndOfBlock (encode.rs:417)
            if (lane < n) out[lane] = src.load1(lane) << 16;
            uint32_t total = n;
            if (ch.flags & CH_LAST_IN_BLOCK) { if (lane == 0) out[total] = CODE_EOB; total += 1; }
            if (lane == 0) ncodes[sg.chunk] = total;
        }
        return;
    }
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;   // warm-up: link only
    const uint32_t base = l0 & ~3u;                               // tile origin (dword aligned)
    const uint32_t cover = FUSED ? n : q1;                        // the tiles cover positions [base, cover)
    const int ntiles = (int)((cover - base + TILE - 1) / TILE);
    const uint32_t n_pad = (n + 3) & ~3u;

    // ---- prologue: empty head table, window bytes for H(0)
    for (uint32_t i = tid; i < (1u << (HASH_BITS - 1)); i += THREADS) {
        const uint32_t f = (base - HEAD_FAR) & 0xFFFFu;
        head32[i] = f | f << 16;
    }
    uint32_t loaded_to = base;                                    // window holds [.., loaded_to) (all waves keep it)
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

    // resolver lane state carried from stage to stage (a lane keeps its index inside the tile: stage X of tile k and
    // stage X+1 of the same tile run on the same lane one or two phases later)
    const uint32_t idx = (wave - 1) * 64 + lane;   // resolver's index inside a tile (waves 1..14)
    uint32_t key_p = 0, key_f = 0, key_r = 0;      // 3-byte prefix of the P / F / R tile position
    uint32_t hh_p = 0, hh_f = 0;                   // its hash (selects the half of the exchanged dword)
    bool val_p = false, val_f = false, val_r = false;   // position takes part in the chain structure (l0 <= p < q1)
    uint32_t cd_f = 0, cd_r = 0;                   // known answer distance (0 = walk)
    uint32_t e_f = 0, e_r = 0;                     // own final link distance (0 = none)
    uint32_t lk_f = 0;                             // F1 → F2: first link state
    uint32_t r_dist = 0;                           // R1 → R2
    bool r_found = false;
    uint32_t code_r = 0, code_t = 0, code_e = 0;   // fused: code word of the R2 / T / E tile position
    uint32_t w_pos = 0, w_cnt = 0;                 // walker (wave 15): next position to visit, codes emitted
    // (tile 0 = position `base` sits at ring offset 0; the loop starts two tiles early)
    uint32_t wk = WRING - 2 * TILE, sk = PRING - 2 * TILE;   // window / prevd ring offset of tile `it`
    uint32_t fill_off = loaded_to - base;          // wave 15: ring offset of position loaded_to
    uint64_t cy_a = 0, cy_b = 0, cy_w = 0;
    const uint64_t lt_mask = lanemask_lt();

    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    else if (wave == 15) __builtin_amdgcn_s_setprio(2);

    // Stages of iteration `it` (tile numbers in parentheses):
    //   phase A   resolvers: R1(it) F1(it+1) P(it+2) [E(it-2) T(it-1)]   wave 15: window loads (global → registers), sweep
    //   phase B   resolvers: F2(it+1) R2(it)      wave 0: H(it+2)         wave 15: [W(it-1)], window stores
    // Inside a phase the stages of a resolver are independent chains of LDS round trips; they are written
    // interleaved — all loads of a step first (dummy addresses for lanes that do not need them), then their uses —
    // so that several round trips are in flight per wavefront.
    const int it_end = FUSED ? ntiles + 2 : ntiles;
    for (int it = -2; it < it_end; ++it) {
        const uint64_t c0 = dbg ? clock64() : 0;
        const uint32_t w1 = ring_fwd(wk, TILE, WRING), s1 = ring_fwd(sk, TILE, PRING);   // tile it+1
        const uint32_t w2 = ring_fwd(w1, TILE, WRING);                                   // tile it+2
        const bool do_r = it >= 0 && it < ntiles;
        const bool do_f = it + 1 >= 0 && it + 1 < ntiles;
        const bool do_p = it + 2 >= 0 && it + 2 < ntiles;
        const bool do_t = FUSED && it - 1 >= 0 && it - 1 < ntiles;
        const bool do_e = FUSED && it - 2 >= 0 && it - 2 < ntiles;
        const uint32_t t_r = base + (uint32_t)it * TILE;            // only used when do_r
        const uint32_t t_f = base + (uint32_t)(it + 1) * TILE;      // only used when do_f
        const uint32_t t_p = base + (uint32_t)(it + 2) * TILE;      // only used when do_p
        uint32_t fill_w0[4] = {0, 0, 0, 0}, fill_w1[4] = {0, 0, 0, 0};
        uint32_t fill_need = loaded_to;

        // =================================================== phase A
        if (wave == 15) {
            // ---- window bytes: global loads now, LDS stores at the end of phase B.  Always four tiles ahead of R:
            //      the last tiles' match lengths read up to 258 bytes past the segment.
            fill_need = max(loaded_to, min(base + (uint32_t)(it + 4) * TILE + 4, n_pad));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t p = loaded_to + 4 * lane + 256 * q;
                if (p < fill_need) src.load_raw(p, fill_w0[q], fill_w1[q]);
            }
            if (do_p) {
                // ---- incremental sweep: stale fields (older than the window) → "far"
                const uint32_t slice = (uint32_t)(it + 2) % SWEEP_SLICES;
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
        } else if (wave >= 1 && wave <= RW) {
            const uint32_t p_r = t_r + idx, p_f = t_f + idx, p_p = t_p + idx;
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
                else if (e_r != 0 && e_r <= window) { dist = e_r; walk = true; }
            }
            struct Hop { uint32_t kq, dn; bool ok; };
            auto hop_issue = [&]() -> Hop {
                Hop h;
                h.ok = false;
                uint32_t aw = w_pos_r, as = s_pos_r;
                if (d != 0) {
                    dist += d;
                    if (dist > window || dist > p_r) d = 0;            // default.rs:81 (inclusive window)
                    else { h.ok = true; aw = ring_back(w_pos_r, dist, WRING); as = ring_back(s_pos_r, dist, PRING); }
                }
                h.kq = win4(win32, aw) & 0xFFFFFFu;
                h.dn = prevd[as];
                return h;
            };
            auto hop_finish = [&](const Hop &h) {
                if (h.ok) { if (h.kq == key_r) { found = true; d = 0; } else d = h.dn; }
            };
            // -- step 0: loads
            const uint32_t d0 = prevd[walk ? ring_back(s_pos_r, dist, PRING) : s_pos_r];
            const uint32_t ow = oldb[idx];
            const uint32_t kp_raw = win4(win32, ring_fwd(w2, idx, WRING));
            uint32_t e_ml = 0, e_mh = 0, e_b = 0;
            if (FUSED) { e_ml = resw[(wave - 1) * 4]; e_mh = resw[(wave - 1) * 4 + 1]; e_b = resw[(wave - 1) * 4 + 2]; }
            // -- step 0: uses
            d = walk ? d0 : 0u;
            uint32_t d_f = 0;
            if (act_f) {
                const uint32_t of = (hh_f & 1) ? ow >> 16 : ow & 0xFFFFu;   // what the exchange returned for this field
                d_f = (p_f - of) & 0xFFFFu;
            }
            if (__ballot(d_f >= FUTURE) && lane == 0) atomicOr(flags, 1u);  // lane-order violation (never observed)
            if (d_f > MAX_WINDOW) d_f = 0;
            // -- step 1: loads (R1 hop 1, F1 predecessor)
            Hop h1 = hop_issue();
            const uint32_t kqf = win4(win32, d_f ? ring_back(w_pos_f, d_f, WRING) : w_pos_f) & 0xFFFFFFu;
            const uint32_t pqf = prevd[d_f > idx ? ring_back(s_pos_f, d_f, PRING) : s_pos_f];   // older tile: final
            // -- step 1: uses
            hop_finish(h1);
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
            // P(it+2): request word of the head pass: hash << 17 | valid << 16 | low 16 bits of the position
            val_p = do_p && p_p >= l0 && p_p < q1;
            key_p = kp_raw & 0xFFFFFFu;
            hh_p = hash3(key_p);
            if (do_p) reqb[idx] = val_p ? (hh_p << 17) | 0x10000u | (p_p & 0xFFFFu) : 0u;
            // E(it-2): the walker's verdict for this wavefront's group → compact the visited code words
            if (do_e) {
                const uint64_t M = (uint64_t)e_ml | (uint64_t)e_mh << 32;
                if ((M >> lane) & 1) out[e_b + __popcll(M & lt_mask)] = code_e;
            }
            // -- further R1 hops, interleaved with T(it-1): the transition table of this wavefront's group by pointer
            //    doubling — for EVERY entry lane, where the greedy walk (default.rs:76-107) leaves the group and which
            //    positions it visits
            uint32_t cur = 64, mlo = 0, mhi = 0, stop_r = 0;
            if (do_t) {
                const uint32_t gb = base + (uint32_t)(it - 1) * TILE + (wave - 1) * 64;
                stop_r = gb < n ? min(64u, n - gb) : 0u;
                cur = lane + ((code_t & 0xFFFFu) ? (code_t >> 16) : 1u);
                mlo = lane < 32 ? 1u << lane : 0u;
                mhi = lane >= 32 ? 1u << (lane - 32) : 0u;
            }
            for (int round = 0;; ++round) {
                const bool any_r = __ballot(d != 0) != 0;
                const bool ta = cur < stop_r;
                const bool any_t = FUSED && do_t && round < 6 && __ballot(ta) != 0;
                if (!any_r && !any_t) break;
                Hop h{0, 0, false};
                if (any_r) h = hop_issue();
                uint32_t c2 = 0, l2 = 0, h2 = 0;
                if (any_t) {
                    const int sl = (int)((ta ? cur : lane) << 2);
                    c2 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)cur);
                    l2 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)mlo);
                    h2 = (uint32_t)__builtin_amdgcn_ds_bpermute(sl, (int)mhi);
                }
                if (any_r) hop_finish(h);
                if (any_t && ta) { cur = c2; mlo |= l2; mhi |= h2; }
            }
            if (do_t) {
                tabE[(wave - 1) * 64 + lane] = (uint16_t)cur;
                tabM[(wave - 1) * 64 + lane] = make_uint2(mlo, mhi);
            }
            r_dist = dist;
            r_found = found;
        }
        const uint64_t c1 = dbg ? clock64() : 0;
        lds_barrier();
        // =================================================== phase B
        if (wave == 0) {
            if (do_p) {
                // ---- H(it+2): the ordered head pass — fourteen exchanges in position order
                // (two batches of seven: the second batch's requests are unpacked while the first is in flight)
#pragma unroll
                for (uint32_t h = 0; h < 2; ++h) {
                    uint32_t old[7], addr[7], mask[7], val[7];
#pragma unroll
                    for (uint32_t s = 0; s < 7; ++s) {
                        const uint32_t rq = reqb[(h * 7 + s) * 64 + lane];
                        const uint32_t sh = (rq >> 13) & 16u;                 // (hash & 1) * 16
                        addr[s] = head_lds + ((rq >> 18) << 2);               // dword of field hash
                        mask[s] = (0u - ((rq >> 16) & 1u)) & (0xFFFFu << sh);
                        val[s] = (rq & 0xFFFFu) << sh;
                    }
                    mskor7(old, addr, mask, val);
#pragma unroll
                    for (uint32_t s = 0; s < 7; ++s) oldb[(h * 7 + s) * 64 + lane] = old[s];
                }
            }
        } else if (wave <= RW) {
            const uint32_t p_r = t_r + idx;
            const uint32_t w_pos_r = ring_fwd(wk, idx, WRING);
            const bool act_r = do_r && val_r && p_r >= q0;
            const bool act_f = do_f && val_f;
            // ---- F2(it+1): inherit the link of the same-prefix predecessor (pointer jumping, no ordering needed: every
            //      state a reader can observe is valid and the oldest member of a run is final from the start)
            // ---- R2(it): longest_common_prefix (default.rs:122-129): 8 bytes per step for the first 16
            uint32_t e = act_f ? lk_f : 0u;
            const bool found = act_r && r_found;
            const uint32_t dist = r_dist;
            uint32_t l = 0, lim = 0;
            uint32_t oa = ring_fwd(w_pos_r, 3, WRING), ob = oa;
            if (found) {
                lim = n - (p_r + 3);                                   // bounded by the end of the chunk
                if (lim > max_len - 3) lim = max_len - 3;
                ob = ring_back(oa, dist, WRING);
            }
            bool cmp = found && lim != 0;                              // still comparing
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
            if (FUSED) {
                // every position of the chunk gets a code word: a pointer, or its own byte as a literal
                // (positions >= end are never hashed and are always literals, default.rs:75,105-107)
                code_r = word ? word : (key_r & 0xFFu) << 16;
            } else if (act_r) {
                md[ch.in_off + p_r] = word;
            }
        } else {
            if (FUSED && do_t) {
                // ---- W(it-1): the serial part of the greedy walk — one table look-up per group of 64 positions.  The
                // tables of seven groups are fetched at a time; the chain then runs on readlanes only and leaves the
                // groups' verdicts {visited mask, codes before} in lanes 0..13 (one store at the end).
                const uint32_t t_w = base + (uint32_t)(it - 1) * TILE;
                uint32_t res_l = 0, res_h = 0, res_c = 0;
#pragma unroll
                for (uint32_t g0 = 0; g0 < NSUB; g0 += NSUB / 2) {
                    uint32_t tE[NSUB / 2], tL[NSUB / 2], tH[NSUB / 2];
#pragma unroll
                    for (uint32_t q = 0; q < NSUB / 2; ++q) {
                        tE[q] = tabE[(g0 + q) * 64 + lane];
                        const uint2 m = tabM[(g0 + q) * 64 + lane];
                        tL[q] = m.x; tH[q] = m.y;
                    }
#pragma unroll
                    for (uint32_t q = 0; q < NSUB / 2; ++q) {
                        const uint32_t g = g0 + q;
                        const uint32_t gb = t_w + g * 64;
                        const uint32_t lim_g = min(gb + 64, n);    // (gb may lie past the chunk: then nothing is visited)
                        uint32_t ml = 0, mh = 0;
                        const uint32_t wp = __builtin_amdgcn_readfirstlane(w_pos);
                        if (wp >= gb && wp < lim_g) {
                            const uint32_t en = wp - gb;
                            ml = (uint32_t)__builtin_amdgcn_readlane((int)tL[q], en);
                            mh = (uint32_t)__builtin_amdgcn_readlane((int)tH[q], en);
                            w_pos = gb + (uint32_t)__builtin_amdgcn_readlane((int)tE[q], en);
                        }
                        if (lane == g) { res_l = ml; res_h = mh; res_c = w_cnt; }
                        w_cnt += __popc(ml) + __popc(mh);
                    }
                }
                if (lane < NSUB) { resw[lane * 4] = res_l; resw[lane * 4 + 1] = res_h; resw[lane * 4 + 2] = res_c; }
            }
            // ---- window stores (their loads were issued in phase A)
            const uint32_t sh = (uint32_t)src.shift;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
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
        }
        const uint64_t c2 = dbg ? clock64() : 0;
        // ---- rotate the stage registers
        key_r = key_f; key_f = key_p; hh_f = hh_p;
        val_r = val_f; val_f = val_p;
        cd_r = cd_f; e_r = e_f;
        code_e = code_t; code_t = code_r;
        wk = w1; sk = s1;
        lds_barrier();
        const uint64_t c3 = dbg ? clock64() : 0;
        cy_a += c1 - c0; cy_b += c2 - c1; cy_w += c3 - c2;
    }
    if (FUSED && wave == 15) {
        uint32_t total = w_cnt;
        if (ch.flags & CH_LAST_IN_BLOCK) { if (lane == 0) out[total] = CODE_EOB; total += 1; }   // encode.rs:417
        if (lane == 0) ncodes[sg.chunk] = total;
    }
    if (dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_a; d[1] = cy_b; d[2] = cy_w; d[3] = 0; d[4] = 0; d[5] = (uint64_t)ntiles;
    }
}

int launch_match2(hipStream_t st, bool fused, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks,
                  const SegDesc *segs, uint32_t nsegs, uint32_t window, uint32_t max_len, uint32_t *md, uint32_t *codes,
                  uint32_t *ncodes, uint32_t *flags, uint64_t *dbg) {
    if (nsegs == 0) return 0;
    const size_t lds = m2::LDS_BYTES;
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)lz77_match2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)lz77_match2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev_ & 63] = true;
    }
    if (fused)
        hipLaunchKernelGGL(lz77_match2_kernel<true>, dim3(nsegs), dim3(m2::THREADS), lds, st, in, in_bytes, chunks, segs, window,
                           max_len, md, codes, ncodes, flags, dbg);
    else
        hipLaunchKernelGGL(lz77_match2_kernel<false>, dim3(nsegs), dim3(m2::THREADS), lds, st, in, in_bytes, chunks, segs, window,
                           max_len, md, codes, ncodes, flags, dbg);
    const hipError_t e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

}  // namespace lfx
