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
constexpr uint32_t HEAD_FAR = 40000;          // distance marker of an empty / swept head field
constexpr uint32_t SWEEP_SLICES = 16;         // the whole table is swept every 16 tiles (14336 positions)
constexpr uint32_t FUTURE = 65536 - 64;       // a distance this large can only come from a lane-order violation

constexpr uint32_t LK_PTR = 32769;            // lk value >= LK_PTR: inherit the link of in-tile index (v - LK_PTR)

// LDS layout (bytes)
constexpr uint32_t OFF_HEAD = 0;                                   // 8192 dwords
constexpr uint32_t OFF_PREVD = OFF_HEAD + (2u << HASH_BITS);       // PRING u16
constexpr uint32_t OFF_WIN = OFF_PREVD + PRING * 2;                // WRING + 8 bytes (+ pad)
constexpr uint32_t OFF_LK = OFF_WIN + WRING + 16;                  // 2 x TILE u16 (raw predecessor, then link state)
constexpr uint32_t OFF_CODE = OFF_LK + 2 * TILE * 2;               // 2 x TILE u32 (fused mode)
constexpr uint32_t LDS_BYTES = OFF_CODE + 2 * TILE * 4;
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

// fourteen 16-bit exchanges, in order, one wait.  old[i] = the dword that held the field before.
// A lane with mask 0 / value 0 leaves its dword untouched.
__device__ __forceinline__ void mskor14(uint32_t (&old)[14], const uint32_t (&addr)[14], const uint32_t (&mask)[14],
                                        const uint32_t (&val)[14]) {
    asm volatile(
        "ds_mskor_rtn_b32 %0, %14, %28, %42\n\t"
        "ds_mskor_rtn_b32 %1, %15, %29, %43\n\t"
        "ds_mskor_rtn_b32 %2, %16, %30, %44\n\t"
        "ds_mskor_rtn_b32 %3, %17, %31, %45\n\t"
        "ds_mskor_rtn_b32 %4, %18, %32, %46\n\t"
        "ds_mskor_rtn_b32 %5, %19, %33, %47\n\t"
        "ds_mskor_rtn_b32 %6, %20, %34, %48\n\t"
        "ds_mskor_rtn_b32 %7, %21, %35, %49\n\t"
        "ds_mskor_rtn_b32 %8, %22, %36, %50\n\t"
        "ds_mskor_rtn_b32 %9, %23, %37, %51\n\t"
        "ds_mskor_rtn_b32 %10, %24, %38, %52\n\t"
        "ds_mskor_rtn_b32 %11, %25, %39, %53\n\t"
        "ds_mskor_rtn_b32 %12, %26, %40, %54\n\t"
        "ds_mskor_rtn_b32 %13, %27, %41, %55\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3]), "=&v"(old[4]), "=&v"(old[5]), "=&v"(old[6]),
          "=&v"(old[7]), "=&v"(old[8]), "=&v"(old[9]), "=&v"(old[10]), "=&v"(old[11]), "=&v"(old[12]), "=&v"(old[13])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
          "v"(addr[8]), "v"(addr[9]), "v"(addr[10]), "v"(addr[11]), "v"(addr[12]), "v"(addr[13]),
          "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]),
          "v"(mask[8]), "v"(mask[9]), "v"(mask[10]), "v"(mask[11]), "v"(mask[12]), "v"(mask[13]),
          "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]),
          "v"(val[8]), "v"(val[9]), "v"(val[10]), "v"(val[11]), "v"(val[12]), "v"(val[13])
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
    uint16_t *lkb = (uint16_t *)(smem + OFF_LK);
    uint32_t *codeb = (uint32_t *)(smem + OFF_CODE);
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

    // resolver lane state carried from stage to stage (same lane index in consecutive tiles)
    const uint32_t idx = (wave - 1) * 64 + lane;   // resolver's index inside a tile (waves 1..14)
    uint32_t key_f = 0, key_r = 0;                 // 3-byte prefix of the F / R tile position
    uint32_t cd_f = 0, cd_r = 0;                   // known answer distance (0 = walk)
    uint32_t e_f = 0, e_r = 0;                     // own final link distance (0 = none)
    bool val_f = false, val_r = false;             // position takes part in the chain structure
    uint32_t lk_f = 0;                             // F1 → F2: first link state
    uint32_t r_dist = 0;                           // R1 → R2
    bool r_found = false;
    // walker state (wave 15)
    uint32_t w_pos = 0, w_cnt = 0;
    // (tile 0 = position `base` sits at ring offset 0; the loop starts two tiles early)
    uint32_t wk = WRING - 2 * TILE, sk = PRING - 2 * TILE;   // window / prevd ring offset of tile `it`
    uint32_t fill_off = loaded_to - base;          // wave 0: ring offset of position loaded_to
    uint64_t cy_a = 0, cy_b = 0, cy_w = 0;

    // ---- W: the greedy walk (default.rs:76-107) over one tile's code words, groups [g_lo, g_hi).  64 answers sit in
    // a VGPR; the vector side precomputes where TWO steps lead from every position (one ds_bpermute) and the two
    // bits they visit, so the serial chain is three readlanes per two steps; the visited code words are compacted
    // into the chunk's code array.
    auto walk_groups = [&](uint32_t tile, uint32_t g_lo, uint32_t g_hi) {
        const uint32_t t_w = base + tile * TILE;
        const uint32_t *cw = codeb + (tile & 1) * TILE;
        const uint64_t lt = lanemask_lt();
        for (uint32_t g = g_lo; g < g_hi; ++g) {
            const uint32_t gb = t_w + g * 64;
            if (gb >= n) break;
            const uint32_t v = cw[g * 64 + lane];
            const uint32_t stepv = (v & 0xFFFFu) ? (v >> 16) : 1u;
            const uint32_t stop_r = min(gb + 64, n) - gb;
            const uint32_t j1 = lane + stepv;
            const bool in1 = j1 < stop_r;
            const uint32_t j1n = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((in1 ? j1 : lane) << 2), (int)j1);
            const uint32_t j2 = in1 ? j1n : j1;
            uint32_t klo = lane < 32 ? 1u << lane : 0u, khi = lane >= 32 ? 1u << (lane - 32) : 0u;
            if (in1) { if (j1 < 32) klo |= 1u << j1; else khi |= 1u << (j1 - 32); }
            uint64_t m = 0;
            uint32_t r = __builtin_amdgcn_readfirstlane(w_pos - gb);   // w_pos >= gb: steps only go forward
            while (r < stop_r) {
                m |= (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)klo, r) |
                     (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)khi, r) << 32;
                r = (uint32_t)__builtin_amdgcn_readlane((int)j2, r);
            }
            w_pos = gb + r;
            if ((m >> lane) & 1) out[w_cnt + __popcll(m & lt)] = v;
            w_cnt += __popcll(m);
        }
    };

    if (wave == 0) __builtin_amdgcn_s_setprio(3);
    else if (FUSED && wave == 15) __builtin_amdgcn_s_setprio(2);

    const int it_end = FUSED ? ntiles + 1 : ntiles;
    for (int it = -2; it < it_end; ++it) {
        const uint64_t c0 = dbg ? clock64() : 0;
        // ring offsets of the tiles in flight (tile it may be "negative" during the ramp-up: only it+1 / it+2 are used then)
        const uint32_t w1 = ring_fwd(wk, TILE, WRING), s1 = ring_fwd(sk, TILE, PRING);
        const uint32_t w2 = ring_fwd(w1, TILE, WRING);
        const bool do_r = it >= 0 && it < ntiles;
        const bool do_f = it + 1 >= 0 && it + 1 < ntiles;
        const bool do_h = it + 2 >= 0 && it + 2 < ntiles;
        // (during the ramp-up wk/sk are kept at the offsets tile `it` WOULD have: tile -2 → ring offset -2*TILE mod ring)
        const uint32_t t_r = base + (uint32_t)it * TILE;            // only used when do_r
        const uint32_t t_h = base + (uint32_t)(it + 2) * TILE;      // only used when do_h
        uint32_t fill_v[4] = {0, 0, 0, 0};
        uint32_t fill_need = loaded_to;

        // =================================================== phase A
        if (wave == 0) {
            // ---- window bytes for H(it+3): global loads now, LDS stores in phase B
            // (always four tiles ahead of R: the last tiles' match lengths read up to 258 bytes past the segment)
            fill_need = max(loaded_to, min(base + (uint32_t)(it + 4) * TILE + 4, n_pad));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t p = loaded_to + 4 * lane + 256 * q;
                if (p < fill_need) fill_v[q] = src.load4(p);
            }
            if (do_h) {
                // ---- incremental sweep: stale fields (older than the window) → "far"
                {
                    const uint32_t slice = (uint32_t)(it + 2) % SWEEP_SLICES;
                    const uint32_t far = (t_h - HEAD_FAR) & 0xFFFFu;
#pragma unroll
                    for (uint32_t q = 0; q < (1u << (HASH_BITS - 1)) / SWEEP_SLICES / 64; ++q) {
                        const uint32_t wi = slice * ((1u << (HASH_BITS - 1)) / SWEEP_SLICES) + q * 64 + lane;
                        const uint32_t v = head32[wi];
                        uint32_t lo = v & 0xFFFFu, hi = v >> 16;
                        const uint32_t dlo = (t_h - lo) & 0xFFFFu, dhi = (t_h - hi) & 0xFFFFu;
                        if (dlo == 0 || dlo > MAX_WINDOW) lo = far;
                        if (dhi == 0 || dhi > MAX_WINDOW) hi = far;
                        head32[wi] = lo | hi << 16;
                    }
                }
                // ---- ordered head pass of tile it+2
                uint16_t *hv = lkb + ((uint32_t)(it + 2) & 1) * TILE;
                uint32_t old[NSUB], addr[NSUB], mask[NSUB], val[NSUB];
#pragma unroll
                for (uint32_t s = 0; s < NSUB; ++s) {
                    const uint32_t p = t_h + s * 64 + lane;
                    const bool v = p >= l0 && p < q1;
                    const uint32_t key = win4(win32, ring_fwd(w2, s * 64 + lane, WRING)) & 0xFFFFFFu;
                    const uint32_t hh = hash3(key);
                    const uint32_t sh = (hh & 1) * 16;
                    addr[s] = head_lds + (hh >> 1) * 4;
                    mask[s] = v ? 0xFFFFu << sh : 0u;
                    val[s] = v ? (p & 0xFFFFu) << sh : 0u;
                }
                mskor14(old, addr, mask, val);
                bool viol = false;
#pragma unroll
                for (uint32_t s = 0; s < NSUB; ++s) {
                    const uint32_t p = t_h + s * 64 + lane;
                    const bool v = mask[s] != 0;
                    const uint32_t of = (mask[s] >> 16) ? old[s] >> 16 : old[s] & 0xFFFFu;
                    uint32_t d = (p - of) & 0xFFFFu;
                    viol |= v && d >= FUTURE;
                    if (d > MAX_WINDOW) d = 0;
                    hv[s * 64 + lane] = v ? (uint16_t)d : (uint16_t)0xFFFFu;
                }
                if (__ballot(viol) && lane == 0) atomicOr(flags, 1u);
            }
        } else if (wave <= RW) {
            // ---- R1(it): chain walk (only where the answer is not already known)
            const uint32_t p_r = t_r + idx;
            const bool act_r = do_r && val_r && p_r >= q0;       // (val_r implies p_r < q1)
            uint32_t dist = 0;
            bool found = false;
            if (act_r) {
                const uint32_t w_pos_r = ring_fwd(wk, idx, WRING), s_pos_r = ring_fwd(sk, idx, PRING);
                uint32_t d = 0;
                if (cd_r) { dist = cd_r; found = dist <= window; }
                else if (e_r) {
                    // the raw predecessor is known to carry another prefix: start at ITS link
                    dist = e_r;
                    if (dist <= window) d = prevd[ring_back(s_pos_r, dist, PRING)];
                }
                while (d != 0) {
                    dist += d;
                    if (dist > window || dist > p_r) break;        // default.rs:81 (inclusive window)
                    const uint32_t kq = win4(win32, ring_back(w_pos_r, dist, WRING)) & 0xFFFFFFu;
                    const uint32_t dn = prevd[ring_back(s_pos_r, dist, PRING)];
                    if (kq == key_r) { found = true; break; }
                    d = dn;
                }
            }
            r_dist = dist;
            r_found = found;
            // ---- F1(it+1): raw predecessor → known answer / first link state
            val_f = false; cd_f = 0; lk_f = 0; key_f = 0;
            if (do_f) {
                uint16_t *lk = lkb + ((uint32_t)(it + 1) & 1) * TILE;
                const uint32_t hvv = lk[idx];
                const uint32_t w_pos_f = ring_fwd(w1, idx, WRING), s_pos_f = ring_fwd(s1, idx, PRING);
                key_f = win4(win32, w_pos_f) & 0xFFFFFFu;
                if (hvv != 0xFFFFu) {
                    val_f = true;
                    const uint32_t d = hvv;
                    uint32_t e = d;                                  // plain link (or none)
                    if (d) {
                        const uint32_t kq = win4(win32, ring_back(w_pos_f, d, WRING)) & 0xFFFFFFu;
                        const uint32_t pq = d > idx ? prevd[ring_back(s_pos_f, d, PRING)] : 0u;   // older tile: final
                        if (kq == key_f) {
                            cd_f = d;
                            if (d > idx) { e = pq ? d + pq : 0; if (e > MAX_WINDOW) e = 0; }
                            else e = LK_PTR + (idx - d);             // in this tile: inherit by pointer jumping
                        }
                    }
                    lk_f = e;
                    lk[idx] = (uint16_t)e;
                }
            }
        } else if (FUSED && it >= 1) {
            walk_groups((uint32_t)(it - 1), 0, NSUB / 2);   // W(it-1), first half of the groups
        }
        const uint64_t c1 = dbg ? clock64() : 0;
        lds_barrier();
        // =================================================== phase B
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t p = loaded_to + 4 * lane + 256 * q;
                if (p < fill_need) {
                    const uint32_t o = ring_fwd(fill_off, 4 * lane + 256 * q, WRING);
                    win32[o >> 2] = fill_v[q];
                    if (o < 8) win32[(WRING + o) >> 2] = fill_v[q];
                }
            }
            fill_off = ring_fwd(fill_off, fill_need - loaded_to, WRING);
            loaded_to = fill_need;
        } else if (wave <= RW) {
            // ---- F2(it+1): inherit the link of the same-prefix predecessor (pointer jumping, no ordering needed:
            //      every state a reader can observe is valid and the oldest member of a run is final from the start)
            if (do_f) {
                uint16_t *lk = lkb + ((uint32_t)(it + 1) & 1) * TILE;
                uint32_t e = lk_f;
                while (__ballot(val_f && e >= LK_PTR)) {
                    if (val_f && e >= LK_PTR) {
                        const uint32_t j = e - LK_PTR;
                        const uint32_t eq = lk[j];
                        if (eq < LK_PTR) {
                            e = eq ? (idx - j) + eq : 0;
                            if (e > MAX_WINDOW) e = 0;
                        } else e = eq;
                        lk[idx] = (uint16_t)e;
                    }
                }
                e_f = e;
                if (val_f) prevd[ring_fwd(s1, idx, PRING)] = (uint16_t)e;
            }
            // ---- R2(it): longest_common_prefix (default.rs:122-129) → code word
            if (do_r) {
                const uint32_t p_r = t_r + idx;
                const bool act_r = val_r && p_r >= q0;
                const bool found = r_found;
                const uint32_t dist = r_dist;
                uint32_t l = 0, lim = 0, oa = 0, ob = 0;
                if (act_r && found) {
                    lim = n - (p_r + 3);                               // bounded by the end of the chunk
                    if (lim > max_len - 3) lim = max_len - 3;
                    oa = ring_fwd(ring_fwd(wk, idx, WRING), 3, WRING);
                    ob = ring_back(oa, dist, WRING);
                    // the first 16 bytes, every lane on its own, 8 bytes per step
                    while (l < lim && l < 16) {
                        const uint64_t x = win8(win32, oa) ^ win8(win32, ob);
                        if (x) { l += (uint32_t)__builtin_ctzll(x) >> 3; break; }
                        l += 8;
                        oa += 8; if (oa >= WRING) oa -= WRING;
                        ob += 8; if (ob >= WRING) ob -= WRING;
                    }
                }
                // a lane still matching after 16 bytes gets the whole wavefront: lane j compares bytes
                // [16+4j, 16+4j+4) — one step settles up to 256 more bytes
                uint64_t lm = __ballot(act_r && found && l == 16 && l < lim);
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
                if (act_r && found) {
                    if (l > lim) l = lim;
                    word = ((3 + l) << 16) | dist;
                }
                if (FUSED) {
                    // every position of the chunk gets a code word: a pointer, or its own byte as a literal
                    // (positions >= end are never hashed and are always literals, default.rs:75,105-107)
                    if (p_r < n) {
                        if (!word) word = (win4(win32, ring_fwd(wk, idx, WRING)) & 0xFFu) << 16;
                        codeb[((uint32_t)it & 1) * TILE + idx] = word;
                    }
                } else if (act_r) {
                    md[ch.in_off + p_r] = word;
                }
            }
        }
        if (FUSED && wave == 15 && it >= 1) walk_groups((uint32_t)(it - 1), NSUB / 2, NSUB);
        const uint64_t c2 = dbg ? clock64() : 0;
        // ---- rotate the stage registers
        key_r = key_f; cd_r = cd_f; e_r = e_f; val_r = val_f;
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
