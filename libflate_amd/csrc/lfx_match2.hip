// lfx_match2.hip — second-generation LZ77 match stage for gfx950: per position, the most recent earlier occurrence
// of its 3-byte prefix inside the chunk and the match length, written to md[] for the parse kernels.
//
// Replaces, bit for bit, the table probe and longest_common_prefix of DefaultLz77Encoder::flush
// (libflate_lz77/src/default.rs:76-87,122-129,146-182).  Parse independence (tests/test_host_pipeline.py): the
// reference inserts EVERY position < end exactly once and in order (default.rs:78,92-97), so
// cand(i) = max{ j < i : buf[j..j+3] == buf[i..i+3] } does not depend on which positions the walk visits.
//
// One workgroup of 16 wavefronts per segment, a software pipeline over tiles of U x 896 positions (14 resolver
// wavefronts x 64 lanes x U positions per lane), two LDS-only barriers per tile:
//
//   phase A   resolvers: R1(k)   chain walk for the positions whose answer is not known yet
//                        F1(k+1) what the head pass returned → raw predecessor → same prefix? (answer known) :
//                                plain link ; first link state lk[]
//                        P(k+2)  3-byte prefix, hash, request word of the head pass
//             wave 15:   window loads (global → registers), incremental sweep of stale head fields
//   phase B   wave 0:    H(k+2)  ordered head pass.  head[] holds 2^14 16-bit fields (low 16 bits of the most recent
//                                position per hash) packed two per dword; ONE ds_mskor_rtn_b32 per 64 positions
//                                exchanges the field and returns the old dword.  The LDS serves the lanes of one
//                                instruction that hit the same field in ascending lane order and a wavefront's
//                                instructions in issue order (measured: tools/exp/mskor_test.hip, 0 violations in
//                                1.3 M conflicting operations), so every lane receives exactly its raw predecessor
//                                ph(p) = most recent earlier position with the same hash.  A lane that observes a
//                                value "from the future" (distance >= 65536-64) proves a violation: the kernel
//                                raises a flag and the host re-runs the first-generation kernel.
//             resolvers: F2(k+1) duplicate-collapsed link by pointer jumping → prevd[]
//                        R2(k)   match length → md[]
//             wave 15:   window stores
//
// The kernel is bound by the latency of dependent LDS round trips (waves parked at s_waitcnt 60 % of the time with
// U = 1, rocprofv3), not by LDS or VALU throughput: every lane therefore carries U = 2 independent positions through
// every stage, and inside a phase the stages are written interleaved — all loads of a step first (dummy addresses for
// lanes that do not need them), then their uses — so that several round trips are in flight per wavefront.
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
constexpr uint32_t RW = 14;                   // resolver wavefronts (waves 1..14; wave 0: head pass, wave 15: window)
constexpr uint32_t U = 1;                     // positions per resolver lane (2 was measured: no gain — the kernel is not
                                              // bound by the latency a second position would hide)
constexpr uint32_t SUBT = RW * 64;            // positions covered by one pass of the resolvers
constexpr uint32_t TILE = SUBT * U;           // 896 positions
constexpr uint32_t NSUB = RW * U;             // 64-position sub-tiles per tile (= exchanges of the head pass)
constexpr int HASH_BITS = 14;
constexpr uint32_t PRING = 32768 + 2048;      // prevd ring (entries)  >= window + 2 TILE (the helper still walks tile k
                                              // while F2 writes tile k+1)
constexpr uint32_t WRING = 36864;             // window ring (bytes)   >= window + 4 TILE + 4: the fill of tile k+4
                                              // must not touch what R(k) reads
constexpr uint32_t HEAD_FAR = 33000;          // distance marker of an empty / swept head field
constexpr uint32_t SWEEP_SLICES = 32;         // the whole table is swept every 32 tiles (28672 positions)
constexpr uint32_t FUTURE = 65536 - 64;       // a distance this large can only come from a lane-order violation
constexpr uint32_t FILL_LOADS = (TILE + 255) / 256;   // dword loads per lane of wave 15 and tile

constexpr uint32_t LK_PTR = 32769;            // lk value >= LK_PTR: inherit the link of in-tile index (v - LK_PTR)

// LDS layout (bytes)
constexpr uint32_t OFF_HEAD = 0;                                   // 8192 dwords
constexpr uint32_t OFF_PREVD = OFF_HEAD + (2u << HASH_BITS);       // PRING u16
constexpr uint32_t OFF_WIN = OFF_PREVD + PRING * 2;                // WRING + 8 bytes (+ pad)
constexpr uint32_t OFF_REQ = OFF_WIN + WRING + 16;                 // TILE u32: head-pass requests (hash, valid, position)
constexpr uint32_t OFF_OLD = OFF_REQ + TILE * 4;                   // TILE u32: the dwords the exchanges returned
constexpr uint32_t OFF_LK = OFF_OLD + TILE * 4;                    // TILE u16: link states of the tile being finalized
constexpr uint32_t OFF_Q = OFF_LK + TILE * 2;                      // 2 x (QCAP x 2 u32 + counter): walks handed to the helper
constexpr uint32_t QCAP = 128;
constexpr uint32_t QBYTES = QCAP * 8 + 16;
constexpr uint32_t LDS_BYTES = OFF_Q + 2 * QBYTES;
static_assert(NSUB % 7 == 0, "the head pass issues batches of seven exchanges");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(WRING % 4 == 0 && 4 * TILE + 4 <= WRING - 32768, "window ring slack");
static_assert(2 * TILE <= PRING - 32768, "prevd ring slack");
static_assert(SWEEP_SLICES * TILE + 32768 + TILE + 64 < FUTURE, "head ages must stay below the violation zone");
static_assert(HEAD_FAR + SWEEP_SLICES * TILE + TILE < FUTURE && HEAD_FAR > 32768, "far marker range");
static_assert(LK_PTR + TILE <= 65536, "link states are 16 bits");
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

// flags[0] |= 1 when the head pass observed a lane-order violation (results are then discarded by the host).
__global__ __launch_bounds__(m2::THREADS) void lz77_match2_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint32_t max_len, uint32_t *__restrict__ md,
    uint32_t *__restrict__ flags, uint64_t *__restrict__ dbg, uint32_t ablate) {
    using namespace m2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *head32 = (uint32_t *)(smem + OFF_HEAD);
    uint16_t *prevd = (uint16_t *)(smem + OFF_PREVD);
    uint32_t *win32 = (uint32_t *)(smem + OFF_WIN);
    uint32_t *reqb = (uint32_t *)(smem + OFF_REQ);
    uint32_t *oldb = (uint32_t *)(smem + OFF_OLD);
    uint16_t *lk = (uint16_t *)(smem + OFF_LK);
    uint32_t *qb = (uint32_t *)(smem + OFF_Q);     // [parity][QCAP x {index, dist << 16 | next link}] then the counter
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

    // ---- prologue: empty head table, window bytes for P(0)
    for (uint32_t i = tid; i < (1u << (HASH_BITS - 1)); i += THREADS) {
        const uint32_t f = (base - HEAD_FAR) & 0xFFFFu;
        head32[i] = f | f << 16;
    }
    if (tid < 2) qb[tid * (QBYTES / 4) + QCAP * 2] = 0;
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

    // resolver lane state carried from stage to stage.  A lane owns U positions of every tile: in-tile indices
    // idx[u] = u * SUBT + (wave - 1) * 64 + lane; stage X of tile k and stage X+1 of the same tile run on the same lane.
    uint32_t idx[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) idx[u] = u * SUBT + (wave - 1) * 64 + lane;
    uint32_t key_p[U], key_f[U], key_r[U];         // 3-byte prefix of the P / F / R tile position
    uint32_t hh_p[U], hh_f[U];                     // its hash (selects the half of the exchanged dword)
    bool val_p[U], val_f[U], val_r[U];             // position takes part in the chain structure (l0 <= p < q1)
    uint32_t cd_f[U], cd_r[U];                     // known answer distance (0 = walk)
    uint32_t e_f[U], e_r[U];                       // own final link distance (0 = none)
    uint32_t lk_f[U];                              // F1 → F2: first link state
    uint32_t r_dist[U];                            // R1 → R2
    bool r_found[U], r_deleg[U];
    // helper (wave 15): the walks it took over in phase B, finished (length + store) in the next phase A
    constexpr uint32_t HE = QCAP / 64;
    uint32_t h_idx[HE], h_dist[HE];
    bool h_found[HE];
#pragma unroll
    for (uint32_t e = 0; e < HE; ++e) { h_idx[e] = 0xFFFFFFFFu; h_dist[e] = 0; h_found[e] = false; }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
        key_p[u] = key_f[u] = key_r[u] = hh_p[u] = hh_f[u] = cd_f[u] = cd_r[u] = e_f[u] = e_r[u] = lk_f[u] = r_dist[u] = 0;
        val_p[u] = val_f[u] = val_r[u] = r_found[u] = r_deleg[u] = false;
    }
    // (tile 0 = position `base` sits at ring offset 0; the loop starts two tiles early)
    uint32_t wk = WRING - 2 * TILE, sk = PRING - 2 * TILE;   // window / prevd ring offset of tile `it`
    uint32_t fill_off = loaded_to - base;          // wave 0: ring offset of position loaded_to
    uint64_t cy_a = 0, cy_b = 0, cy_w = 0;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));

    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    else if (wave == 15) __builtin_amdgcn_s_setprio(2);

    for (int it = -2; it < ntiles + 1; ++it) {     // (+1: the helper finishes the last tile's walks one phase later)
        const uint64_t c0 = dbg ? clock64() : 0;
        const uint32_t w1 = ring_fwd(wk, TILE, WRING), s1 = ring_fwd(sk, TILE, PRING);   // tile it+1
        const uint32_t w2 = ring_fwd(w1, TILE, WRING);                                   // tile it+2
        const bool do_r = it >= 0 && it < ntiles;
        const bool do_f = it + 1 >= 0 && it + 1 < ntiles;
        const bool do_p = it + 2 >= 0 && it + 2 < ntiles;
        const uint32_t t_r = base + (uint32_t)it * TILE;            // only used when do_r
        const uint32_t t_f = base + (uint32_t)(it + 1) * TILE;      // only used when do_f
        const uint32_t t_p = base + (uint32_t)(it + 2) * TILE;      // only used when do_p
        uint32_t fill_w0[FILL_LOADS], fill_w1[FILL_LOADS];
        uint32_t fill_need = loaded_to;

        // =================================================== phase A
        if (wave == 0) {
            // ---- window bytes: global loads now, LDS stores in phase B behind the head pass.  Always four tiles ahead of R:
            //      the last tiles' match lengths read up to 258 bytes past the segment.
            fill_need = max(loaded_to, min(base + (uint32_t)(it + 4) * TILE + 4, n_pad));
#pragma unroll
            for (uint32_t q = 0; q < FILL_LOADS; ++q) {
                const uint32_t p = loaded_to + 4 * lane + 256 * q;
                fill_w0[q] = fill_w1[q] = 0;
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
            uint32_t p_r[U], w_pos_r[U], s_pos_r[U], w_pos_f[U], s_pos_f[U];
            bool act_r[U], act_f[U];
            // ---- R1(it): chain walk, only where the answer is not already known (cd) — and then starting at the LINK of
            //      the raw predecessor, which is known to carry another prefix
            uint32_t dist[U], d[U];
            bool found[U], walk[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                p_r[u] = t_r + idx[u];
                w_pos_r[u] = ring_fwd(wk, idx[u], WRING); s_pos_r[u] = ring_fwd(sk, idx[u], PRING);
                w_pos_f[u] = ring_fwd(w1, idx[u], WRING); s_pos_f[u] = ring_fwd(s1, idx[u], PRING);
                act_r[u] = do_r && val_r[u] && p_r[u] >= q0;       // (val_r implies p_r < q1)
                act_f[u] = do_f && val_f[u];
                dist[u] = 0; d[u] = 0; found[u] = false; walk[u] = false;
                if (act_r[u]) {
                    if (cd_r[u]) { dist[u] = cd_r[u]; found[u] = dist[u] <= window; }
                    else if (e_r[u] != 0 && e_r[u] <= window && !(ablate & 1)) { dist[u] = e_r[u]; walk[u] = true; }
                }
            }
            struct Hop { uint32_t kq, dn; bool ok; };
            auto hop_issue = [&](uint32_t u) -> Hop {
                Hop h;
                h.ok = false;
                uint32_t aw = w_pos_r[u], as = s_pos_r[u];
                if (d[u] != 0) {
                    dist[u] += d[u];
                    if (dist[u] > window || dist[u] > p_r[u]) d[u] = 0;            // default.rs:81 (inclusive window)
                    else { h.ok = true; aw = ring_back(w_pos_r[u], dist[u], WRING); as = ring_back(s_pos_r[u], dist[u], PRING); }
                }
                h.kq = win4(win32, aw) & 0xFFFFFFu;
                h.dn = prevd[as];
                return h;
            };
            auto hop_finish = [&](uint32_t u, const Hop &h) {
                if (h.ok) { if (h.kq == key_r[u]) { found[u] = true; d[u] = 0; } else d[u] = h.dn; }
            };
            // -- step 0: loads
            uint32_t d0[U], ow[U], kp_raw[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                d0[u] = prevd[walk[u] ? ring_back(s_pos_r[u], dist[u], PRING) : s_pos_r[u]];
                ow[u] = oldb[idx[u]];
                kp_raw[u] = win4(win32, ring_fwd(w2, idx[u], WRING));
            }
            // -- step 0: uses
            uint32_t d_f[U];
            bool viol = false;
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                d[u] = walk[u] ? d0[u] : 0u;
                d_f[u] = 0;
                if (act_f[u]) {
                    const uint32_t of = (hh_f[u] & 1) ? ow[u] >> 16 : ow[u] & 0xFFFFu;   // what the exchange returned for this field
                    d_f[u] = (t_f + idx[u] - of) & 0xFFFFu;
                }
                viol |= d_f[u] >= FUTURE;
                if (d_f[u] > MAX_WINDOW || (ablate & 32)) d_f[u] = 0;
            }
            if (__ballot(viol) && lane == 0) atomicOr(flags, 1u);                        // lane-order violation (never observed)
            // -- step 1: loads (R1 hop 1, F1 predecessor)
            Hop h1[U];
            uint32_t kqf[U], pqf[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                h1[u] = hop_issue(u);
                kqf[u] = win4(win32, d_f[u] ? ring_back(w_pos_f[u], d_f[u], WRING) : w_pos_f[u]) & 0xFFFFFFu;
                pqf[u] = prevd[d_f[u] > idx[u] ? ring_back(s_pos_f[u], d_f[u], PRING) : s_pos_f[u]];   // older tile: final
            }
            // -- step 1: uses
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                hop_finish(u, h1[u]);
                // F1(it+1): raw predecessor → known answer / first link state
                cd_f[u] = 0;
                uint32_t e = d_f[u];                                 // plain link (or none)
                if (d_f[u] != 0 && kqf[u] == key_f[u]) {
                    cd_f[u] = d_f[u];
                    if (d_f[u] > idx[u]) { e = pqf[u] ? d_f[u] + pqf[u] : 0; if (e > MAX_WINDOW) e = 0; }
                    else e = LK_PTR + (idx[u] - d_f[u]);             // in this tile: inherit by pointer jumping
                }
                lk_f[u] = e;
                if (act_f[u]) lk[idx[u]] = (uint16_t)e;
                // P(it+2): request word of the head pass: hash << 17 | valid << 16 | low 16 bits of the position
                const uint32_t p_p = t_p + idx[u];
                val_p[u] = do_p && p_p >= l0 && p_p < q1;
                key_p[u] = kp_raw[u] & 0xFFFFFFu;
                hh_p[u] = hash3(key_p[u]);
                if (do_p) reqb[idx[u]] = val_p[u] ? (hh_p[u] << 17) | 0x10000u | (p_p & 0xFFFFu) : 0u;
            }
            // -- walks that are still going after the in-line hop (2 % of the positions, but nearly every wavefront has
            //    one, and every further hop is a full LDS round trip for the whole wavefront): hand them to the helper
            //    wavefront, which walks them densely during the next two phases and finishes those positions itself
            bool deleg[U];
            {
                uint32_t *q = qb + ((uint32_t)it & 1) * (QBYTES / 4);
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {
                    deleg[u] = false;
                    const bool still = d[u] != 0;
                    const uint64_t bm = __ballot(still);
                    if (bm) {
                        uint32_t slot0 = 0;
                        if (lane == 0) slot0 = atomicAdd(&q[QCAP * 2], (uint32_t)__popcll(bm));
                        slot0 = __builtin_amdgcn_readfirstlane(slot0);
                        const uint32_t slot = slot0 + __popcll(bm & lt_mask);
                        if (still && slot < QCAP) {
                            q[slot * 2] = idx[u];
                            q[slot * 2 + 1] = dist[u] << 16 | d[u];
                            deleg[u] = true;
                            d[u] = 0;
                        }
                    }
                }
            }
            // -- (queue full: the owner keeps walking)
            for (;;) {
                bool any = false;
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) any |= d[u] != 0;
                if (!__ballot(any)) break;
                Hop h[U];
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) h[u] = hop_issue(u);
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) hop_finish(u, h[u]);
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) { r_dist[u] = dist[u]; r_found[u] = found[u]; r_deleg[u] = deleg[u]; }
        }
        else if (wave == 15) {
            // ---- helper, second half: match length and md[] for the walks of tile it-1 it resolved in the last phase B
            const uint32_t t_h = base + (uint32_t)(it - 1) * TILE;
            const uint32_t wkh = ring_back(wk, TILE, WRING);
            uint32_t l[HE], lim[HE], oa[HE], ob[HE];
            bool cmp[HE];
#pragma unroll
            for (uint32_t e = 0; e < HE; ++e) {
                l[e] = 0; lim[e] = 0;
                oa[e] = ring_fwd(ring_fwd(wkh, h_idx[e], WRING), 3, WRING);
                ob[e] = oa[e];
                if (h_found[e]) {
                    lim[e] = n - (t_h + h_idx[e] + 3);
                    if (lim[e] > max_len - 3) lim[e] = max_len - 3;
                    ob[e] = ring_back(oa[e], h_dist[e], WRING);
                }
                cmp[e] = h_found[e] && lim[e] != 0;
            }
            bool any_found = false;
#pragma unroll
            for (uint32_t e = 0; e < HE; ++e) any_found |= h_found[e];
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
                if (h_idx[e] != 0xFFFFFFFFu) {                       // a queue entry: this lane owns the position's md word
                    uint32_t word = 0;
                    if (h_found[e]) {
                        if (l[e] > lim[e]) l[e] = lim[e];
                        word = ((3 + l[e]) << 16) | h_dist[e];
                    }
                    md[ch.in_off + t_h + h_idx[e]] = word;
                }
                h_idx[e] = 0xFFFFFFFFu; h_found[e] = false;
            }
        }
        const uint64_t c1 = dbg ? clock64() : 0;
        lds_barrier();
        // =================================================== phase B
        if (wave == 0) {
            if (do_p) {
                // ---- H(it+2): the ordered head pass — NSUB exchanges in position order, in batches of seven
#pragma unroll
                for (uint32_t h = 0; h < NSUB / 7; ++h) {
                    uint32_t old[7], addr[7], mask[7], val[7];
#pragma unroll
                    for (uint32_t s = 0; s < 7; ++s) {
                        const uint32_t rq = reqb[(h * 7 + s) * 64 + lane];
                        const uint32_t sh = (rq >> 13) & 16u;                 // (hash & 1) * 16
                        addr[s] = head_lds + ((rq >> 18) << 2);               // dword of field hash
                        mask[s] = (0u - ((rq >> 16) & 1u)) & (0xFFFFu << sh);
                        val[s] = (rq & 0xFFFFu) << sh;
                    }
                    if (!(ablate & 8)) mskor7(old, addr, mask, val);
                    else { for (uint32_t s = 0; s < 7; ++s) old[s] = 0; }
#pragma unroll
                    for (uint32_t s = 0; s < 7; ++s) oldb[(h * 7 + s) * 64 + lane] = old[s];
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
            // ---- F2(it+1): inherit the link of the same-prefix predecessor (pointer jumping, no ordering needed: every
            //      state a reader can observe is valid and the oldest member of a run is final from the start)
            // ---- R2(it): longest_common_prefix (default.rs:122-129): 8 bytes per step for the first 16
            uint32_t p_r[U], e[U], dist[U], l[U], lim[U], oa[U], ob[U];
            bool act_r[U], act_f[U], found[U], cmp[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                p_r[u] = t_r + idx[u];
                act_r[u] = do_r && val_r[u] && p_r[u] >= q0;
                act_f[u] = do_f && val_f[u];
                e[u] = act_f[u] ? lk_f[u] : 0u;
                if ((ablate & 4) && e[u] >= LK_PTR) e[u] = 0;
                found[u] = act_r[u] && r_found[u];
                dist[u] = r_dist[u];
                l[u] = 0; lim[u] = 0;
                oa[u] = ring_fwd(ring_fwd(wk, idx[u], WRING), 3, WRING);
                ob[u] = oa[u];
                if (found[u]) {
                    lim[u] = n - (p_r[u] + 3);                         // bounded by the end of the chunk
                    if (lim[u] > max_len - 3) lim[u] = max_len - 3;
                    ob[u] = ring_back(oa[u], dist[u], WRING);
                }
                cmp[u] = found[u] && lim[u] != 0 && !(ablate & 2);     // still comparing
            }
#pragma unroll
            for (int step = 0; step < 2; ++step) {
                uint32_t j[U], eq[U];
                uint64_t xa[U], xb[U];
                bool ptr[U];
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {                     // loads
                    ptr[u] = e[u] >= LK_PTR;
                    j[u] = ptr[u] ? e[u] - LK_PTR : idx[u];
                    eq[u] = lk[j[u]];
                    xa[u] = win8(win32, oa[u]); xb[u] = win8(win32, ob[u]);
                }
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {                     // uses
                    if (ptr[u]) {
                        if (eq[u] < LK_PTR) { e[u] = eq[u] ? (idx[u] - j[u]) + eq[u] : 0; if (e[u] > MAX_WINDOW) e[u] = 0; }
                        else e[u] = eq[u];
                        lk[idx[u]] = (uint16_t)e[u];
                    }
                    if (cmp[u]) {
                        const uint64_t x = xa[u] ^ xb[u];
                        if (x) { l[u] += (uint32_t)__builtin_ctzll(x) >> 3; cmp[u] = false; }
                        else {
                            l[u] += 8;
                            oa[u] += 8; if (oa[u] >= WRING) oa[u] -= WRING;
                            ob[u] += 8; if (ob[u] >= WRING) ob[u] -= WRING;
                            if (l[u] >= lim[u]) cmp[u] = false;
                        }
                    }
                }
            }
            for (;;) {
                bool any = false;
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) any |= e[u] >= LK_PTR;
                if (!__ballot(any)) break;
#pragma unroll
                for (uint32_t u = 0; u < U; ++u) {
                    if (e[u] >= LK_PTR) {
                        const uint32_t j = e[u] - LK_PTR;
                        const uint32_t eq = lk[j];
                        if (eq < LK_PTR) { e[u] = eq ? (idx[u] - j) + eq : 0; if (e[u] > MAX_WINDOW) e[u] = 0; }
                        else e[u] = eq;
                        lk[idx[u]] = (uint16_t)e[u];
                    }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                e_f[u] = e[u];
                if (act_f[u]) prevd[ring_fwd(s1, idx[u], PRING)] = (uint16_t)e[u];
            }
            // a lane still matching after 16 bytes gets the whole wavefront: lane j compares bytes
            // [16+4j, 16+4j+4) — one step settles up to 256 more bytes
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                uint64_t lm = __ballot(cmp[u]);                        // (cmp here ⇒ l == 16 < lim)
                while (lm) {
                    const uint32_t sl = (uint32_t)__builtin_ctzll(lm);
                    lm &= lm - 1;
                    const uint32_t boa = __builtin_amdgcn_readlane(oa[u], sl), bob = __builtin_amdgcn_readlane(ob[u], sl);
                    const uint32_t blim = __builtin_amdgcn_readlane(lim[u], sl);
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
                    if (lane == sl) l[u] = res;
                }
                uint32_t word = 0;
                if (found[u]) {
                    if (l[u] > lim[u]) l[u] = lim[u];
                    word = ((3 + l[u]) << 16) | dist[u];
                }
                if (act_r[u] && !r_deleg[u] && !(ablate & 16)) md[ch.in_off + p_r[u]] = word;
            }
        } else {
            // ---- helper, first half: the walks of tile `it` the resolvers handed over, one per lane, to their end
            uint32_t *q = qb + ((uint32_t)it & 1) * (QBYTES / 4);
            const uint32_t cnt = do_r ? min(q[QCAP * 2], QCAP) : 0u;
            if (cnt) {
                uint32_t p_h[HE], w_pos_h[HE], s_pos_h[HE], key_h[HE], dist[HE], d[HE];
                bool found[HE];
#pragma unroll
                for (uint32_t e = 0; e < HE; ++e) {
                    const uint32_t slot = e * 64 + lane;
                    const bool v = slot < cnt;
                    const uint32_t qi = v ? q[slot * 2] : 0u, qs = v ? q[slot * 2 + 1] : 0u;
                    h_idx[e] = v ? qi : 0xFFFFFFFFu;
                    p_h[e] = t_r + qi;
                    w_pos_h[e] = ring_fwd(wk, qi, WRING); s_pos_h[e] = ring_fwd(sk, qi, PRING);
                    key_h[e] = win4(win32, w_pos_h[e]) & 0xFFFFFFu;
                    dist[e] = qs >> 16; d[e] = qs & 0xFFFFu;
                    found[e] = false;
                }
                for (;;) {
                    bool any = false;
#pragma unroll
                    for (uint32_t e = 0; e < HE; ++e) any |= d[e] != 0;
                    if (!__ballot(any)) break;
                    uint32_t kq[HE], dn[HE];
                    bool ok[HE];
#pragma unroll
                    for (uint32_t e = 0; e < HE; ++e) {
                        ok[e] = false;
                        uint32_t aw = w_pos_h[e], as = s_pos_h[e];
                        if (d[e] != 0) {
                            dist[e] += d[e];
                            if (dist[e] > window || dist[e] > p_h[e]) d[e] = 0;        // default.rs:81
                            else { ok[e] = true; aw = ring_back(w_pos_h[e], dist[e], WRING); as = ring_back(s_pos_h[e], dist[e], PRING); }
                        }
                        kq[e] = win4(win32, aw) & 0xFFFFFFu;
                        dn[e] = prevd[as];
                    }
#pragma unroll
                    for (uint32_t e = 0; e < HE; ++e)
                        if (ok[e]) { if (kq[e] == key_h[e]) { found[e] = true; d[e] = 0; } else d[e] = dn[e]; }
                }
#pragma unroll
                for (uint32_t e = 0; e < HE; ++e) { h_found[e] = found[e]; h_dist[e] = dist[e]; }
                if (lane == 0) q[QCAP * 2] = 0;
            }
        }
        const uint64_t c2 = dbg ? clock64() : 0;
        // ---- rotate the stage registers
#pragma unroll
        for (uint32_t u = 0; u < U; ++u) {
            key_r[u] = key_f[u]; key_f[u] = key_p[u]; hh_f[u] = hh_p[u];
            val_r[u] = val_f[u]; val_f[u] = val_p[u];
            cd_r[u] = cd_f[u]; e_r[u] = e_f[u];
        }
        wk = w1; sk = s1;
        lds_barrier();
        const uint64_t c3 = dbg ? clock64() : 0;
        cy_a += c1 - c0; cy_b += c2 - c1; cy_w += c3 - c2;
    }
    if (dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_a; d[1] = cy_b; d[2] = cy_w; d[3] = 0; d[4] = 0; d[5] = (uint64_t)ntiles;
    }
}

int launch_match2(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint32_t max_len, uint32_t *md, uint32_t *flags, uint64_t *dbg) {
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
                       md, flags, dbg, ablate);
    const hipError_t e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

}  // namespace lfx
