// lfx_encode_kernels.hip — hand-written gfx950 kernels of the DEFLATE encode hot path.
//
//   lz77_match   : per position, most recent earlier occurrence of the same 3-byte prefix inside the
//                  chunk (hash chains in LDS, exact verification) + match length.  Replaces the
//                  PrefixTable probe + longest_common_prefix of DefaultLz77Encoder::flush
//                  (libflate_lz77/src/default.rs:76-87,122-129,146-182).  Parse-independent: every
//                  position < end is inserted exactly once and in order by the reference
//                  (default.rs:78,92-97), so cand(i) = max{ j < i : buf[j..j+3] == buf[i..i+3] }.
//   lz77_parse   : the greedy walk i += length / i += 1 (default.rs:76-107) → Code stream.
//   histogram    : DynamicHuffmanCodec::build counting (src/deflate/symbol.rs:322-337).
//   huffman      : lfx_huff.h (one wavefront per block).
//   offsets/pack : BitWriter (src/bit.rs:25-49) as size → scan → LSB-first scatter.
//   checksum     : CRC-32 / Adler-32 of the input (src/checksum.rs:4-33).
//
// No MFMA (no dense contraction on this path); the bound is HBM / LDS bandwidth.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfx_common.h"
#include "lfx_device.h"
#include "lfx_huff.h"

namespace lfx {

// ------------------------------------------------------------------------------------------------
// byte access through aligned dword loads (input pointers are arbitrary byte addresses)
struct ByteSrc {
    gptr_u32 w;         // 4-byte aligned base (global address space)
    uint64_t shift;     // byte offset of logical byte 0 inside w
    uint64_t nbytes;    // logical size
    __device__ __forceinline__ uint32_t load4(uint64_t off) const {
        // bytes [off, off+4) little-endian; bytes beyond the buffer read as 0
        uint64_t a = off + shift;
        uint64_t idx = a >> 2;
        uint32_t sh = (uint32_t)a & 3;
        uint64_t last = (nbytes + shift + 3) >> 2;  // number of dwords covering the buffer
        uint32_t w0 = idx < last ? w[idx] : 0;
        uint32_t w1 = (sh != 0 && idx + 1 < last) ? w[idx + 1] : 0;
        return __builtin_amdgcn_alignbyte(w1, w0, sh);
    }
    __device__ __forceinline__ uint32_t load1(uint64_t off) const {
        uint64_t a = off + shift;
        return (w[a >> 2] >> (((uint32_t)a & 3) * 8)) & 0xFF;
    }
};
__device__ __forceinline__ ByteSrc make_src(const uint8_t *p, uint64_t n) {
    ByteSrc s;
    uint64_t a = (uint64_t)p;
    s.w = (gptr_u32)(a & ~3ull);
    s.shift = a & 3;
    s.nbytes = n;
    return s;
}

// workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a release fence for
// global memory (s_waitcnt vmcnt(0)): inside a streaming loop that makes every iteration wait for
// its HBM store acknowledgements.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    uint32_t lane = __lane_id();
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// ------------------------------------------------------------------------------------------------
// lz77_match: one workgroup (16 wavefronts) per segment, software pipelined over tiles of 960
// positions: wavefront 0 LINKS tile i+1 (hash + insert, in position order) while wavefronts 1..15
// RESOLVE tile i (chain walk with exact 3-byte verification + match length).  Everything the inner
// loops touch lives in LDS:
//   head[16384] u16 : low 16 bits of the most recent inserted position per hash (stale entries are
//                     swept to a "far" marker every <= 16 Ki positions so they can never alias)
//   prevd[PRING] u16: distance to the previous position with the same hash (0 = none in window)
//   win[WRING+8] u8 : the input bytes [tile - 32 KiB, tile + lookahead) as a ring
constexpr int MATCH_THREADS = 1024;
constexpr uint32_t MTILE = 960;                 // 15 resolver wavefronts x 64 lanes
constexpr int HASH_BITS = 14;
constexpr uint32_t PRING = 32768 + 2048;        // prevd ring (entries)
constexpr uint32_t WRING = 36864;               // window ring (bytes), multiple of 4096
constexpr uint32_t HEAD_FAR = 40000;            // distance marker of a swept head entry
constexpr size_t MATCH_LDS = (2u << HASH_BITS) + PRING * 2 + WRING + 8 + 3 * MTILE * 4 + MTILE * 4 + MTILE * 4 + MTILE * 4;

__device__ __forceinline__ uint32_t hash3(uint32_t key) { return (key * 2654435761u) >> (32 - HASH_BITS); }
__device__ __forceinline__ uint32_t wring_off(uint32_t pos) {
    // pos % 36864 = ((pos >> 12) % 9) << 12 | (pos & 4095)
    const uint32_t x = pos >> 12;
    const uint32_t q = (uint32_t)(((uint64_t)x * 954437177ull) >> 33);  // x / 9
    return ((x - q * 9) << 12) | (pos & 4095);
}
// ring arithmetic without divisions (v_mul_hi is quarter rate): offsets are derived from one per-tile
// offset by adding a lane index or subtracting a distance (<= 32768, smaller than either ring)
__device__ __forceinline__ uint32_t ring_fwd(uint32_t off, uint32_t add, uint32_t size) {   // add < size
    const uint32_t r = off + add;
    return r >= size ? r - size : r;
}
__device__ __forceinline__ uint32_t ring_back(uint32_t off, uint32_t sub, uint32_t size) {  // sub < size
    return off >= sub ? off - sub : off + size - sub;
}
__device__ __forceinline__ uint32_t win_at(const uint32_t *win32, uint32_t off) {
    // 4 bytes at ring byte offset `off` (the 8 mirror bytes after the ring make the wrap seamless)
    const uint32_t w0 = win32[off >> 2], w1 = win32[(off >> 2) + 1];
    return __builtin_amdgcn_alignbyte(w1, w0, off & 3);
}

__global__ __launch_bounds__(MATCH_THREADS) void lz77_match_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint32_t max_len, uint32_t *__restrict__ md,
    uint64_t *__restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *head = (uint16_t *)smem;
    uint16_t *prevd = (uint16_t *)(smem + (2u << HASH_BITS));
    uint32_t *win32 = (uint32_t *)(smem + (2u << HASH_BITS) + PRING * 2);
    // staging of the link tile: 24-bit prefix | (lane delta to the nearest lower same-hash lane, bit 6:
    // that lane has the same prefix, bit 7: last lane with this hash) << 24 ; 0xFFFFFFFF = no position
    uint32_t *st_k = (uint32_t *)(smem + (2u << HASH_BITS) + PRING * 2 + WRING + 8);
    // cd[parity][idx]: distance to the most recent same-prefix position when the insertion already
    // knows it (0 = the resolver has to walk the chain)
    uint16_t *cd = (uint16_t *)(st_k + 3 * MTILE);
    // hvs[parity][idx]: the bucket head every position of a tile saw in the ordered head pass
    uint16_t *hvs = cd + 2 * MTILE;
    // lk[idx]: link state of the tile being finalized: LK_DONE | distance to the link target (0 = none), or
    // the in-tile index of the same-prefix predecessor whose link this position inherits
    uint32_t *lk = (uint32_t *)(hvs + 2 * MTILE);
    constexpr uint32_t LK_DONE = 0x80000000u;

    const SegDesc sg = segs[blockIdx.x];
    const ChunkDesc ch = chunks[sg.chunk];
    if (ch.flags & CH_LITERALS) return;
    const uint32_t n = (uint32_t)ch.len;
    const uint32_t end = (n > 3 ? n : 3) - 3;  // default.rs:75
    const uint32_t q0 = sg.start;              // first position answered by this segment
    const uint32_t q1 = min(sg.start + sg.len, end);
    if (q0 >= q1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;  // warm-up: link only
    const uint32_t base = l0 & ~3u;                              // tile origin (dword aligned)
    const ByteSrc src = make_src(in + ch.in_off, in_bytes - ch.in_off);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t ntiles = (q1 - base + MTILE - 1) / MTILE;

    for (uint32_t i = tid; i < (1u << HASH_BITS); i += MATCH_THREADS) head[i] = (uint16_t)(base - HEAD_FAR);
    uint32_t loaded_to = base;      // window holds [.., loaded_to)
    uint32_t swept_at = base;
    uint64_t cy_load = 0, cy_work = 0, cy_wait = 0;

    // Pipeline, four stages one tile apart, two barriers per iteration `it`:
    //   before the first barrier : window extension, head sweep, finalize-init(tile it+1)
    //   wave 0                   : ordered head pass of tile it+2
    //   waves 1..15              : finalize-jump(tile it+1) ; resolve(tile it) ; pre-digest(tile it+3)
    // pre-digest : 3-byte prefix, nearest lower same-hash lane by 14 ballots, flags           (parallel)
    // head pass  : head read → head write per 64-position sub-tile, in order — the only serial part of the
    //              insertion.  A head write does not depend on the read before it and the LDS executes a
    //              wavefront's operations in order: the 15 read/write pairs are issued back to back.
    // finalize   : every position now knows its raw predecessor ph (most recent earlier position with the
    //              same hash).  Duplicate collapsing: when ph carries the SAME prefix it is this position's
    //              answer (cd, no walk later) and the position's chain link is ph's own link — an older
    //              occurrence of the same prefix can never be an answer again, so chains hold one entry
    //              per run of equal prefixes and a rare prefix that shares a bucket with a frequent one
    //              does not walk through hundreds of useless entries.  link(p) = link(ph) is resolved
    //              without any ordering: predecessors in older tiles are final (one read); inside the tile
    //              by pointer jumping over lk[] (log rounds, no barrier: every state a reader can see is
    //              valid, and the oldest member of a run is final from the start).
    // resolve    : chain walk + match length                                                  (parallel)
    constexpr uint32_t NSUB = MTILE / 64;
    auto predigest = [&](uint32_t tile_idx) {
        const uint32_t idx = (wave - 1) * 64 + lane;
        const uint32_t p = base + tile_idx * MTILE + idx;
        const bool v = tile_idx < ntiles && p >= l0 && p < q1;
        uint32_t key = 0, hh = 0;
        if (v) { key = win_at(win32, wring_off(p)) & 0xFFFFFFu; hh = hash3(key); }
        // match-any over the 14 hash bits: a lane differs from lane j in bit b iff ballot_b[j] != its own bit;
        // per bit one sign-extending bit extract, one compare and one or-of-xor per 32-lane half
        uint32_t dlo = 0, dhi = 0;
#pragma unroll
        for (int b = 0; b < HASH_BITS; ++b) {
            const int32_t nb = ((int32_t)(hh << (31 - b))) >> 31;          // 0 or -1
            const uint64_t m = __ballot(nb != 0);
            dlo |= (uint32_t)m ^ (uint32_t)nb;
            dhi |= (uint32_t)(m >> 32) ^ (uint32_t)nb;
        }
        const uint64_t same = __ballot(v) & ~((uint64_t)dhi << 32 | dlo);
        const uint64_t lower = same & lanemask_lt();
        const uint32_t hb = lower ? 63 - __clzll(lower) : lane;
        const uint32_t kprev = __shfl(key, hb);
        uint32_t f = lower ? lane - hb : 0;                 // 1..63, 0 = none in this sub-tile
        if (lower && kprev == key) f |= 0x40;               // ... and it has the same 3-byte prefix
        if (((same >> lane) >> 1) == 0) f |= 0x80;          // last lane with this hash
        st_k[(tile_idx % 3) * MTILE + idx] = v ? (key | (f << 24)) : 0xFFFFFFFFu;
    };
    if (wave == 0) __builtin_amdgcn_s_setprio(3);

    for (int it = -3; it < (int)ntiles; ++it) {
        const uint64_t c0 = dbg ? clock64() : 0;
        const uint32_t t_res = base + (uint32_t)it * MTILE;              // tile being resolved (it >= 0)
        const uint32_t fin_idx = (uint32_t)(it + 1);                     // tile being finalized (it >= -1)
        const uint32_t t_fin = base + fin_idx * MTILE;
        const bool do_fin = it >= -1 && fin_idx < ntiles;
        const uint32_t head_idx = (uint32_t)(it + 2);                    // tile of the head pass (it >= -2)
        const uint32_t t_head = base + head_idx * MTILE;
        const bool do_head = it >= -2 && head_idx < ntiles;
        const uint32_t pre_idx = (uint32_t)(it + 3);                     // tile being pre-digested
        // ---- A: everyone extends the window to cover the pre-digest tile (+3 bytes)
        const uint32_t need = min(base + pre_idx * MTILE + MTILE + 4, (n + 3) & ~3u);
        for (uint32_t p = loaded_to + 4 * tid; p < need; p += 4 * MATCH_THREADS) {
            const uint32_t v = src.load4(p);
            const uint32_t o = wring_off(p);
            win32[o >> 2] = v;
            if (o < 8) win32[(WRING + o) >> 2] = v;   // mirror
        }
        if (need > loaded_to) loaded_to = need;
        if (do_head && t_head + MTILE - swept_at > 16384) {
            // stale heads (older than the window) → "far", so that 16-bit distances never alias
            for (uint32_t i = tid; i < (1u << HASH_BITS); i += MATCH_THREADS) {
                const uint32_t d = (t_head - head[i]) & 0xFFFFu;
                if (d == 0 || d > MAX_WINDOW) head[i] = (uint16_t)(t_head - HEAD_FAR);
            }
            swept_at = t_head;
        }
        // finalize-init: raw predecessor, known answer, first link state
        const uint32_t fidx = (wave - 1) * 64 + lane;
        bool fin_valid = false;
        uint32_t fin_e = LK_DONE;
        if (wave > 0 && do_fin) {
            const uint32_t sk = st_k[(fin_idx % 3) * MTILE + fidx];
            if (sk != 0xFFFFFFFFu) {
                fin_valid = true;
                const uint32_t key = sk & 0xFFFFFFu, f = sk >> 24;
                const uint32_t w_p = ring_fwd(wring_off(t_fin), fidx, WRING), s_p = ring_fwd(t_fin % PRING, fidx, PRING);
                uint32_t d = f & 0x3F;                       // nearest lower same-hash lane of the sub-tile
                bool same = (f & 0x40) != 0;
                if (d == 0) {
                    d = (t_fin + fidx - hvs[(fin_idx & 1) * MTILE + fidx]) & 0xFFFFu;
                    if (d > MAX_WINDOW) d = 0;
                    if (d) same = (win_at(win32, ring_back(w_p, d, WRING)) & 0xFFFFFFu) == key;
                }
                uint32_t e = LK_DONE | d;                   // a predecessor with another prefix: plain link
                if (d && same) {
                    if (d > fidx) {                          // the predecessor lies in an older tile: its link is final
                        const uint32_t pq = prevd[ring_back(s_p, d, PRING)];
                        e = LK_DONE | (pq ? d + pq : 0);
                    } else e = fidx - d;                     // ... in this tile: inherit through pointer jumping
                }
                fin_e = e;
                lk[fidx] = e;
                cd[(fin_idx & 1) * MTILE + fidx] = (uint16_t)((d && same) ? d : 0);
            }
        }
        lds_barrier();
        const uint64_t c1 = dbg ? clock64() : 0;
        if (wave == 0) {
            if (do_head) {
                const uint32_t *stl = st_k + (head_idx % 3) * MTILE;
                uint16_t *hv_out = hvs + (head_idx & 1) * MTILE;
                uint32_t hvv[NSUB];
#pragma unroll
                for (uint32_t sub = 0; sub < NSUB; ++sub) {
                    const uint32_t sk = stl[sub * 64 + lane];
                    const uint32_t hh = hash3(sk & 0xFFFFFFu) & ((1u << HASH_BITS) - 1);
                    hvv[sub] = head[hh];
                    if (sk != 0xFFFFFFFFu && ((sk >> 24) & 0x80)) head[hh] = (uint16_t)(t_head + sub * 64 + lane);
                }
#pragma unroll
                for (uint32_t sub = 0; sub < NSUB; ++sub) hv_out[sub * 64 + lane] = (uint16_t)hvv[sub];
            }
        } else {
            if (do_fin) {
                // finalize-jump: inherit the link of the same-prefix predecessor
                uint32_t e = fin_e;
                while (__ballot(fin_valid && !(e & LK_DONE))) {
                    if (fin_valid && !(e & LK_DONE)) {
                        const uint32_t eq = lk[e];          // e = in-tile index of the predecessor
                        if (eq & LK_DONE) {
                            const uint32_t dq = eq & ~LK_DONE;
                            e = LK_DONE | (dq ? (fidx - e) + dq : 0);
                        } else e = eq;
                        lk[fidx] = e;
                    }
                }
                if (fin_valid) {
                    const uint32_t dist = e & ~LK_DONE;
                    prevd[ring_fwd(t_fin % PRING, fidx, PRING)] = (uint16_t)(dist <= MAX_WINDOW ? dist : 0);
                }
            }
            if (it >= 0) {
            const uint32_t w_res = wring_off(t_res), s_res = t_res % PRING;   // wave-uniform
            const uint32_t pos = t_res + (wave - 1) * 64 + lane;
            const bool act = pos >= q0 && pos < q1;
            uint32_t dist = 0, l = 0, lim = 0, oa = 0, ob = 0, lit_byte = 0;
            bool found = false;
            if (act) {
                // resolve: walk the hash chain until the exact 3-byte prefix matches (the most recent
                // occurrence) or the chain leaves the window
                const uint32_t w_pos = ring_fwd(w_res, (wave - 1) * 64 + lane, WRING);
                const uint32_t s_pos = ring_fwd(s_res, (wave - 1) * 64 + lane, PRING);
                // (the prefix is still in this wavefront's own staging slots: the pre-digest of tile it+3, which
                // reuses them, comes later in this wavefront's program order)
                const uint32_t key = st_k[((uint32_t)it % 3) * MTILE + (wave - 1) * 64 + lane] & 0xFFFFFFu;
                const uint32_t known = cd[((uint32_t)it & 1) * MTILE + (wave - 1) * 64 + lane];
                lit_byte = key & 0xFFu;
                uint32_t d = prevd[s_pos];
                if (known) { dist = known; found = dist <= window; d = 0; }
                while (d != 0) {
                    dist += d;
                    if (dist > window || dist > pos) break;  // default.rs:81 (inclusive window)
                    if ((win_at(win32, ring_back(w_pos, dist, WRING)) & 0xFFFFFFu) == key) { found = true; break; }
                    d = prevd[ring_back(s_pos, dist, PRING)];
                }
                if (found) {
                    // longest_common_prefix default.rs:122-129: up to max_len-3 more bytes, bounded by
                    // the end of the chunk.  Phase 1: the first 16 bytes, every lane on its own.
                    lim = n - (pos + 3);
                    if (lim > max_len - 3) lim = max_len - 3;
                    oa = ring_fwd(w_pos, 3, WRING); ob = ring_back(oa, dist, WRING);
                    while (l < lim && l < 16) {
                        const uint32_t x = win_at(win32, oa) ^ win_at(win32, ob);
                        if (x) { l += (uint32_t)__builtin_ctz(x) >> 3; break; }
                        l += 4;
                        oa += 4; if (oa >= WRING) oa -= WRING;
                        ob += 4; if (ob >= WRING) ob -= WRING;
                    }
                }
            }
            // Phase 2: a lane still matching after 16 bytes gets the whole wavefront: lane k compares
            // bytes [16+4k, 16+4k+4) — one step settles up to 256 more bytes
            uint64_t lm = __ballot(found && l == 16 && l < lim);
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
                    x = win_at(win32, a) ^ win_at(win32, b);
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
            if (act) {
                uint32_t out = lit_byte << 16;      // (no match: the literal's code word)
                if (found) {
                    if (l > lim) l = lim;
                    out = ((3 + l) << 16) | dist;
                }
                md[ch.in_off + pos] = out;
            }
            }
            predigest(pre_idx);
        }
        const uint64_t c2 = dbg ? clock64() : 0;
        lds_barrier();
        const uint64_t c3 = dbg ? clock64() : 0;
        cy_load += c1 - c0; cy_work += c2 - c1; cy_wait += c3 - c2;
    }
    if (dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_load; d[1] = cy_work; d[2] = cy_wait; d[3] = 0; d[4] = 0; d[5] = ntiles;
    }
}

// ------------------------------------------------------------------------------------------------
// lz77_parse: lfx_parse2.hip.
// tile → chunk and segment → chunk tables (one workgroup per chunk): the tile / segment kernels start with one load
// instead of a binary search over the chunk list (ten dependent L2 round trips per workgroup, also in the many
// workgroups of tiles that hold no code)
// (round 6: the call's first kernel also clears its small accumulators — the blocks' symbol counters, the result record, the
//  match stage's per-segment counts: three fill operations of 5 to 12 us each in front of the match kernel before)
__global__ __launch_bounds__(256) void chunk_maps_kernel(const ChunkDesc *__restrict__ chunks, uint32_t nchunks,
                                                         uint64_t ntiles, uint32_t nsegs,
                                                         uint32_t *__restrict__ tile_map, uint32_t *__restrict__ seg_map,
                                                         ZeroSpan z0, ZeroSpan z1, ZeroSpan z2) {
    {
        const uint32_t nthreads = 256u * gridDim.x * gridDim.y, me = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
        for (uint32_t i = me; i < z0.n; i += nthreads) z0.p[i] = 0;
        for (uint32_t i = me; i < z1.n; i += nthreads) z1.p[i] = 0;
        for (uint32_t i = me; i < z2.n; i += nthreads) z2.p[i] = 0;
    }
    const uint32_t c = blockIdx.x;
    const ChunkDesc ch = chunks[c];
    const uint64_t t1 = c + 1 < nchunks ? chunks[c + 1].tile_base : ntiles;
    const uint32_t s1 = c + 1 < nchunks ? chunks[c + 1].seg_base : nsegs;
    for (uint64_t t = ch.tile_base + blockIdx.y * 256 + threadIdx.x; t < t1; t += 256ull * gridDim.y) tile_map[t] = c;
    for (uint32_t q = ch.seg_base + blockIdx.y * 256 + threadIdx.x; q < s1; q += 256u * gridDim.y) seg_map[q] = c;
}

// ------------------------------------------------------------------------------------------------
// histogram of a chunk's codes into its block's counters (LDS privatised per workgroup)
constexpr int HIST_STRIDE = 320;  // [0,288) literal/length, [288,320) distance

__global__ __launch_bounds__(256) void histogram_kernel(const ChunkDesc *__restrict__ chunks,
                                                        const uint32_t *__restrict__ codes,
                                                        const uint32_t *__restrict__ ncodes,
                                                        uint32_t *__restrict__ hist) {
    // HREP replicas of the counters, chosen by the lane: the lanes of one atomic instruction that count the same symbol
    // (a frequent literal, length 3) are served one after the other — a quarter as many per address
    constexpr uint32_t HREP = 4;       // (measured: 1 replica 0.197 ms, 4: 0.142, 16: 0.140 at 256 MiB)
    __shared__ uint32_t hh[HREP * HIST_STRIDE];
    uint32_t *h = hh + (threadIdx.x & (HREP - 1)) * HIST_STRIDE;
    const uint32_t c = blockIdx.x;
    const ChunkDesc ch = chunks[c];
    for (uint32_t i = threadIdx.x; i < HREP * HIST_STRIDE; i += 256) hh[i] = 0;
    __syncthreads();
    const uint32_t n = ncodes[c];
    const uint32_t per = (uint32_t)div_up(n, gridDim.y);
    const uint32_t lo = blockIdx.y * per, hi = min(n, lo + per);
    const uint32_t *p = codes + ch.code_off;
    // HLOADS loads in flight per lane: the kernel is one workgroup per chunk (more workgroups cost more global atomics at
    // the end), so a lane walks ~400 codes and every trip is an HBM round trip — with four loads per trip the kernel's
    // time WAS those ~100 round trips (round 3: 4 → 16 loads per trip)
    constexpr uint32_t HLOADS = 16;
    for (uint32_t i0 = lo + threadIdx.x; i0 < hi; i0 += HLOADS * 256) {
        uint32_t v[HLOADS];
#pragma unroll
        for (uint32_t q = 0; q < HLOADS; ++q) v[q] = p[min(i0 + 256 * q, hi - 1)];
#pragma unroll
        for (uint32_t q = 0; q < HLOADS; ++q) {
            if (i0 + 256 * q >= hi) break;
            const uint32_t dist = v[q] & 0xFFFFu, val = v[q] >> 16;
            if (dist == 0) {
                atomicAdd(&h[val], 1u);
            } else {
                uint32_t eb, ex;
                atomicAdd(&h[len_symbol(val, eb, ex)], 1u);
                atomicAdd(&h[288 + dist_symbol(dist, eb, ex)], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *g = hist + (uint64_t)ch.block * HIST_STRIDE;
    for (uint32_t i = threadIdx.x; i < HIST_STRIDE; i += 256) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t r = 0; r < HREP; ++r) v += hh[r * HIST_STRIDE + i];
        if (v) atomicAdd(&g[i], v);
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int HUFF_THREADS = 320;   // five wavefronts per block: the rank / merge / code passes stride over the lanes, and 286 symbols
                                    // on 256 lanes made every such pass two trips for thirty symbols (round 3)
__global__ __launch_bounds__(HUFF_THREADS) void huffman_kernel(const BlockDesc *__restrict__ blocks,
                                                               const uint32_t *__restrict__ hist,
                                                               BlockCodes *__restrict__ bc, uint64_t *__restrict__ dbg) {
    __shared__ HuffScratch S;
    const uint32_t b = blockIdx.x;
    const uint32_t type = blocks[b].type;
    if (type == BT_RAW) return;
    huff_block_build(hist + (uint64_t)b * HIST_STRIDE, type, &bc[b], S, (int)threadIdx.x, HUFF_THREADS);
    if (dbg && b == 0 && threadIdx.x < 24) dbg[threadIdx.x] = S.stamp[threadIdx.x];   // (LFX_DEBUG)
}

// ------------------------------------------------------------------------------------------------
// block start bits: a serial fold (stored blocks byte-align, the final block byte-aligns)
__global__ __launch_bounds__(256) void offsets_kernel(const BlockDesc *__restrict__ blocks, uint32_t nblocks,
                                                      const BlockCodes *__restrict__ bc, uint64_t start_bit,
                                                      uint64_t cap_bits, uint64_t *__restrict__ block_start,
                                                      EncodeResult *__restrict__ res) {
    // the fold is serial (stored blocks and the final block byte-align), its inputs are gathered in
    // parallel: 256 blocks per round into LDS
    __shared__ uint64_t s_bits[256];
    __shared__ uint32_t s_kind[256];   // bit 0: stored block, bit 1: align after
    __shared__ uint64_t s_bit;
    if (threadIdx.x == 0) s_bit = start_bit;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += 256) {
        const uint32_t b = b0 + threadIdx.x;
        if (b < nblocks) {
            const BlockDesc bd = blocks[b];
            s_kind[threadIdx.x] = (bd.type == BT_RAW ? 1u : 0u) | (bd.align_after ? 2u : 0u);
            s_bits[threadIdx.x] = bd.type == BT_RAW ? 32 + 8 * bd.in_len : bc[b].body_bits;
        }
        const uint32_t cnt = min(256u, nblocks - b0);
        // Compressed blocks without an alignment behind them simply add up: when the round holds nothing else — except,
        // possibly, an alignment behind its LAST block — the starts are an exclusive prefix sum (round 6: the serial fold was
        // 21 us for the 257 blocks of 256 MiB, two dependent LDS reads per block)
        const uint32_t my_kind = b < nblocks ? s_kind[threadIdx.x] : 0u;
        const bool odd = b < nblocks && (my_kind & 1u || ((my_kind & 2u) && threadIdx.x + 1 != cnt));
        const int serial = __syncthreads_or(odd ? 1 : 0);
        if (!serial) {
            const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            uint64_t x = b < nblocks ? s_bits[threadIdx.x] : 0ull;
            const uint64_t mine = x;
            for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(x, o); if ((int)lane >= o) x += y; }
            __shared__ uint64_t s_wsum[4];
            if (lane == 63) s_wsum[wave] = x;
            __syncthreads();
            uint64_t pre = s_bit;
            for (uint32_t w = 0; w < wave; ++w) pre += s_wsum[w];
            if (b < nblocks) block_start[b] = pre + x - mine;
            __syncthreads();
            if (threadIdx.x == cnt - 1) {
                uint64_t bit = pre + x;
                if (my_kind & 2u) bit = (bit + 7) & ~7ull;
                s_bit = bit;
            }
        } else if (threadIdx.x == 0) {
            uint64_t bit = s_bit;
            for (uint32_t k = 0; k < cnt; ++k) {
                block_start[b0 + k] = bit;
                if (s_kind[k] & 1) { bit += 3; bit = (bit + 7) & ~7ull; }   // RawBuf::flush → BitWriter::flush (encode.rs:372)
                bit += s_bits[k];
                if (s_kind[k] & 2) bit = (bit + 7) & ~7ull;
            }
            s_bit = bit;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        res->end_bit = s_bit;
        res->status = s_bit > cap_bits ? 1u : 0u;
    }
}

// ------------------------------------------------------------------------------------------------
// batch encode (lfx_encode_batch_device: many independent streams in one launch set): a stream's blocks start right
// behind its own container header; one lane per stream folds its few blocks (the same fold as offsets_kernel) and checks
// the stream's capacity — one stream that does not fit voids the launch (res->status), its status word says which.
__global__ __launch_bounds__(256) void offsets_batch_kernel(const BatchStream *__restrict__ streams, uint32_t count,
                                                            const BlockDesc *__restrict__ blocks, const BlockCodes *__restrict__ bc,
                                                            uint32_t hdr_len, uint32_t trailer_len, uint64_t *__restrict__ block_start,
                                                            uint64_t *__restrict__ stream_end, int32_t *__restrict__ status,
                                                            EncodeResult *__restrict__ res) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= count) return;
    const BatchStream bs = streams[s];
    uint64_t bit = (bs.out_off + hdr_len) * 8;
    for (uint32_t b = bs.first_block; b < bs.first_block + bs.n_blocks; ++b) {
        const BlockDesc bd = blocks[b];
        block_start[b] = bit;
        if (bd.type == BT_RAW) { bit += 3; bit = (bit + 7) & ~7ull; bit += 32 + 8 * bd.in_len; }
        else bit += bc[b].body_bits;
        if (bd.align_after) bit = (bit + 7) & ~7ull;
    }
    stream_end[s] = bit;
    const bool over = bs.out_cap < hdr_len + trailer_len || bit > (bs.out_off + bs.out_cap - trailer_len) * 8;
    status[s] = over ? 7 : 0;                 // LFX_E_NOSPACE
    if (over) atomicOr(&res->status, 1u);
}
// container header (the same bytes for every stream: shared options) and trailer of every stream
__global__ __launch_bounds__(256) void frame_batch_kernel(int format, const BatchStream *__restrict__ streams, uint32_t count,
                                                          const uint8_t *__restrict__ hdr, uint32_t hdr_len,
                                                          const uint64_t *__restrict__ stream_end, const uint32_t *__restrict__ crc,
                                                          const uint32_t *__restrict__ adler, const EncodeResult *__restrict__ res,
                                                          uint32_t *__restrict__ out, uint64_t *__restrict__ out_len) {
    const uint32_t s = blockIdx.x * 256 + threadIdx.x;
    if (s >= count || res->status != 0) return;
    const BatchStream bs = streams[s];
    auto put = [&](uint64_t ob, uint32_t byte) { atomicOr(&out[ob >> 2], byte << (8 * (ob & 3))); };
    for (uint32_t i = 0; i < hdr_len; ++i) put(bs.out_off + i, hdr[i]);
    const uint64_t at = stream_end[s] >> 3;          // (the final block byte-aligns)
    uint32_t nt = 0;
    if (format == 2) {                               // gzip.rs:114-121: CRC-32 LE + ISIZE LE
        const uint32_t c = bs.in_len ? crc[s] : 0u, sz = (uint32_t)bs.in_len;
        for (uint32_t i = 0; i < 4; ++i) { put(at + i, (c >> (8 * i)) & 255u); put(at + 4 + i, (sz >> (8 * i)) & 255u); }
        nt = 8;
    } else if (format == 1) {                        // zlib.rs:630-639: Adler-32 BE
        const uint32_t a = bs.in_len ? adler[s] : 1u;
        for (uint32_t i = 0; i < 4; ++i) put(at + i, (a >> (8 * (3 - i))) & 255u);
        nt = 4;
    }
    out_len[s] = at + nt - bs.out_off;
}

// ------------------------------------------------------------------------------------------------
// pack: bits of one code word
__device__ __forceinline__ uint32_t code_bits(uint32_t v, const uint32_t *lit, const uint32_t *dst,
                                              uint64_t &bits) {
    const uint32_t dist = v & 0xFFFFu, val = v >> 16;
    if (dist == 0) {
        const uint32_t e = lit[val];
        bits = e & 0xFFFFu;
        return e >> 16;
    }
    uint32_t eb, ex, db, dx;
    const uint32_t le = lit[len_symbol(val, eb, ex)];
    const uint32_t de = dst[dist_symbol(dist, db, dx)];
    uint32_t n = le >> 16;
    uint64_t acc = le & 0xFFFFu;
    acc |= (uint64_t)ex << n;
    n += eb;
    acc |= (uint64_t)(de & 0xFFFFu) << n;
    n += de >> 16;
    acc |= (uint64_t)dx << n;
    n += db;
    bits = acc;
    return n;
}

constexpr int PACK_THREADS = 256;
constexpr int PACK_PER_THREAD = PACK_TILE / PACK_THREADS;  // 8

// Workgroups take PACK_TPW consecutive tiles (round 3).  With one tile per workgroup every workgroup started with a chain
// of dependent loads — tile → chunk, chunk descriptor, code count and block type, code tables — before its codes were
// even requested, and the tiles behind a chunk's last code (tiles are laid out for the worst case, 129 per chunk; a text
// fills 52) each paid three of those round trips only to find out that they are empty.  Now the chain is paid once per
// PACK_TPW tiles (descriptor and tables are kept while the chunk / block stays the same), and the next tile's codes are
// requested before the current tile's are used.
constexpr uint32_t PACK_TPW = 4;       // (measured: 1 tile per workgroup 0.372 ms for tile_bits + pack, 4: 0.342, 8: 0.363)

__global__ __launch_bounds__(PACK_THREADS) void tile_bits_kernel(
    const ChunkDesc *__restrict__ chunks, uint32_t nchunks, const BlockDesc *__restrict__ blocks,
    const uint32_t *__restrict__ codes, const uint32_t *__restrict__ ncodes,
    const BlockCodes *__restrict__ bc, uint32_t *__restrict__ tile_bits, const uint32_t *__restrict__ tile_map,
    uint64_t ntiles) {
    __shared__ uint32_t lit[288], dst[32], red[PACK_THREADS / 64];
    constexpr uint32_t PER = PACK_TILE / PACK_THREADS;
    const uint64_t g0 = (uint64_t)blockIdx.x * PACK_TPW;
    uint32_t cm[PACK_TPW];
#pragma unroll
    for (uint32_t q = 0; q < PACK_TPW; ++q) cm[q] = tile_map[min(g0 + q, ntiles - 1)];
    uint32_t cur_c = 0xFFFFFFFFu, cur_block = 0xFFFFFFFFu, n = 0;
    ChunkDesc ch{};
    bool raw = false;
    uint32_t v[PER], vn[PER];
    bool have_next = false;          // vn holds the codes of this tile (requested while the previous one was summed)
#pragma unroll
    for (uint32_t q = 0; q < PACK_TPW; ++q) {
        const uint64_t gt = g0 + q;
        if (gt >= ntiles) break;
        const uint32_t c = cm[q];
        if (c != cur_c) {
            cur_c = c;
            ch = chunks[c];
            n = ncodes[c];
            raw = blocks[ch.block].type == BT_RAW;
            have_next = false;
        }
        const uint32_t lo = (uint32_t)(gt - ch.tile_base) * PACK_TILE;
        if (lo >= n || raw) {
            if (threadIdx.x == 0) tile_bits[gt] = 0;
            have_next = false;
            continue;
        }
        const uint32_t hi = min(n, lo + PACK_TILE);
        const uint32_t *p = codes + ch.code_off;
        // all of the lane's loads are issued before the first use (clamped addresses, no branch around a load)
        if (have_next) {
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) v[k] = vn[k];
        } else {
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) v[k] = p[min(lo + threadIdx.x + k * PACK_THREADS, hi - 1)];
        }
        // the next tile of the same chunk: its codes are requested now (an empty one re-reads the chunk's last code)
        have_next = q + 1 < PACK_TPW && gt + 1 < ntiles && cm[q + 1 < PACK_TPW ? q + 1 : q] == c;
        if (have_next) {
#pragma unroll
            for (uint32_t k = 0; k < PER; ++k) vn[k] = p[min(lo + PACK_TILE + threadIdx.x + k * PACK_THREADS, n - 1)];
        }
        if (ch.block != cur_block) {
            __syncthreads();             // (the previous tile's sums have read the old tables)
            cur_block = ch.block;
            const BlockCodes *B = &bc[ch.block];
            for (uint32_t i = threadIdx.x; i < 288; i += PACK_THREADS) lit[i] = B->lit[i];
            if (threadIdx.x < 32) dst[threadIdx.x] = B->dist[threadIdx.x];
        }
        __syncthreads();
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; ++k) {
            uint64_t bits;
            const uint32_t nb = code_bits(v[k], lit, dst, bits);
            sum += lo + threadIdx.x + k * PACK_THREADS < hi ? nb : 0u;
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) tile_bits[gt] = red[0] + red[1] + red[2] + red[3];
    }
}

// per block: exclusive scan of its tiles' bit counts → absolute start bit of every tile
__global__ __launch_bounds__(1024) void tile_scan_kernel(const ChunkDesc *__restrict__ chunks,
                                                        const BlockDesc *__restrict__ blocks,
                                                        const BlockCodes *__restrict__ bc,
                                                        const uint64_t *__restrict__ block_start,
                                                        const uint32_t *__restrict__ tile_bits,
                                                        uint64_t total_tiles,
                                                        uint32_t nchunks,
                                                        uint64_t *__restrict__ tile_start) {
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    const uint32_t b = blockIdx.x;
    const BlockDesc bd = blocks[b];
    if (bd.type == BT_RAW || bd.n_chunks == 0) return;
    const uint64_t t0 = chunks[bd.first_chunk].tile_base;
    const uint32_t lastc = bd.first_chunk + bd.n_chunks;
    const uint64_t t1 = lastc < nchunks ? chunks[lastc].tile_base : total_tiles;
    if (threadIdx.x == 0) carry = block_start[b] + 3 + bc[b].hdr_bits;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t T = blockDim.x;   // (256, or 1024 when a block has many tiles: the scan is serial in batches of T)
    for (uint64_t base = t0; base < t1; base += T) {
        const uint64_t i = base + threadIdx.x;
        const uint64_t v = i < t1 ? tile_bits[i] : 0;
        uint64_t x = v;  // inclusive wave scan
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t y = __shfl_up(x, o);
            if ((int)lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint64_t pre = carry;
        for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
        if (i < t1) tile_start[i] = pre + x - v;
        __syncthreads();
        if (threadIdx.x == T - 1) carry = pre + x;
        __syncthreads();
    }
}

constexpr int STAGE_WORDS64 = (PACK_TILE * 48) / 64 + 4;  // worst case + phase + spill

__global__ __launch_bounds__(PACK_THREADS) void pack_kernel(
    const ChunkDesc *__restrict__ chunks, uint32_t nchunks, const BlockDesc *__restrict__ blocks,
    const uint32_t *__restrict__ codes, const uint32_t *__restrict__ ncodes,
    const BlockCodes *__restrict__ bc, const uint64_t *__restrict__ tile_start,
    const EncodeResult *__restrict__ res, uint64_t out_base_bit, uint32_t *__restrict__ out,
    const uint32_t *__restrict__ tile_map, uint64_t ntiles) {
    __shared__ uint32_t lit[288], dst[32];
    __shared__ uint32_t wsum[PACK_THREADS / 64];
    __shared__ unsigned long long stage[STAGE_WORDS64];
    if (res->status != 0) return;
    // PACK_TPW consecutive tiles per workgroup (see tile_bits_kernel)
    const uint64_t g0 = (uint64_t)blockIdx.x * PACK_TPW;
    uint32_t cm[PACK_TPW];
    uint64_t ts[PACK_TPW];
#pragma unroll
    for (uint32_t q = 0; q < PACK_TPW; ++q) {
        cm[q] = tile_map[min(g0 + q, ntiles - 1)];
        ts[q] = tile_start[min(g0 + q, ntiles - 1)];       // (an empty tile's entry is never used)
    }
    uint32_t cur_c = 0xFFFFFFFFu, cur_block = 0xFFFFFFFFu, n = 0;
    ChunkDesc ch{};
    bool raw = false;
    uint32_t cv[PACK_PER_THREAD], cvn[PACK_PER_THREAD];
    bool have_next = false;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (uint32_t q = 0; q < PACK_TPW; ++q) {
        const uint64_t gt = g0 + q;
        if (gt >= ntiles) break;
        const uint32_t c = cm[q];
        if (c != cur_c) {
            cur_c = c;
            ch = chunks[c];
            n = ncodes[c];
            raw = blocks[ch.block].type == BT_RAW;
            have_next = false;
        }
        const uint32_t lo = (uint32_t)(gt - ch.tile_base) * PACK_TILE;
        if (lo >= n || raw) { have_next = false; continue; }
        const uint32_t hi = min(n, lo + PACK_TILE);
        const uint32_t *p = codes + ch.code_off;
        // each lane owns PACK_PER_THREAD consecutive codes (loads first, branch-free)
        const uint32_t first = lo + threadIdx.x * PACK_PER_THREAD;
        if (have_next) {
#pragma unroll
            for (int k = 0; k < PACK_PER_THREAD; ++k) cv[k] = cvn[k];
        } else {
#pragma unroll
            for (int k = 0; k < PACK_PER_THREAD; ++k) cv[k] = p[min(first + k, hi - 1)];
        }
        have_next = q + 1 < PACK_TPW && gt + 1 < ntiles && cm[q + 1 < PACK_TPW ? q + 1 : q] == c;
        if (have_next) {
#pragma unroll
            for (int k = 0; k < PACK_PER_THREAD; ++k) cvn[k] = p[min(first + PACK_TILE + k, n - 1)];
        }
        __syncthreads();                 // (the previous tile's flush has read the staging buffer, its sums the tables)
        if (ch.block != cur_block) {
            cur_block = ch.block;
            const BlockCodes *B = &bc[ch.block];
            for (uint32_t i = threadIdx.x; i < 288; i += PACK_THREADS) lit[i] = B->lit[i];
            if (threadIdx.x < 32) dst[threadIdx.x] = B->dist[threadIdx.x];
        }
        for (uint32_t i = threadIdx.x; i < STAGE_WORDS64; i += PACK_THREADS) stage[i] = 0;
        __syncthreads();
        uint64_t cb[PACK_PER_THREAD];
        uint32_t cn[PACK_PER_THREAD];
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < PACK_PER_THREAD; ++k) {
            const uint32_t i = first + k;
            const uint32_t nb = code_bits(cv[k], lit, dst, cb[k]);
            cn[k] = i < hi ? nb : 0u;
            cb[k] = i < hi ? cb[k] : 0ull;
            mine += cn[k];
        }
        // exclusive scan of `mine` over the workgroup
        uint32_t x = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(x, o);
            if ((int)lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint32_t pre = 0;
        for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
        const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const uint64_t start = ts[q] - out_base_bit;  // bit offset inside `out`
        const uint32_t phase = (uint32_t)start & 31;
        uint32_t bit = phase + pre + x - mine;
#pragma unroll
        for (int k = 0; k < PACK_PER_THREAD; ++k) {
            if (cn[k]) {
                const uint32_t w = bit >> 6, sh = bit & 63;
                atomicOr(&stage[w], cb[k] << sh);
                if (sh + cn[k] > 64) atomicOr(&stage[w + 1], cb[k] >> (64 - sh));
                bit += cn[k];
            }
        }
        __syncthreads();
        // staging → output words; the first and last word may be shared with neighbours
        const uint32_t nwords = (phase + total + 31) >> 5;
        const uint32_t *s32 = (const uint32_t *)stage;
        uint32_t *o = out + (start >> 5);
        for (uint32_t i = threadIdx.x; i < nwords; i += PACK_THREADS) {
            const uint32_t v = s32[i];
            if (i == 0 || i == nwords - 1) { if (v) atomicOr(&o[i], v); }
            else o[i] = v;
        }
    }
}

// block header bits (BFINAL, BTYPE, dynamic table) and stored blocks
__global__ __launch_bounds__(256) void block_header_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const BlockDesc *__restrict__ blocks,
    const BlockCodes *__restrict__ bc, const uint64_t *__restrict__ block_start,
    const EncodeResult *__restrict__ res, uint64_t out_base_bit, uint32_t *__restrict__ out) {
    if (res->status != 0) return;
    const uint32_t b = blockIdx.x;
    const BlockDesc bd = blocks[b];
    const uint64_t start = block_start[b] - out_base_bit;
    // Block::flush encode.rs:291-292: 1 bit BFINAL then 2 bits BTYPE, LSB-first
    const uint32_t first3 = (bd.final & 1) | (bd.type << 1);
    if (threadIdx.x == 0) {
        const uint32_t sh = (uint32_t)start & 31;
        atomicOr(&out[start >> 5], first3 << sh);
        if (sh > 29) atomicOr(&out[(start >> 5) + 1], first3 >> (32 - sh));
    }
    if (bd.type == BT_DYNAMIC) {
        const BlockCodes *B = &bc[b];
        const uint64_t hb = start + 3;
        const uint32_t nw = (B->hdr_bits + 31) >> 5;
        const uint32_t sh = (uint32_t)hb & 31;
        for (uint32_t i = threadIdx.x; i < nw; i += 256) {
            const uint32_t v = B->hdr[i];
            if (v == 0) continue;
            atomicOr(&out[(hb >> 5) + i], v << sh);
            if (sh) atomicOr(&out[(hb >> 5) + i + 1], v >> (32 - sh));
        }
    } else if (bd.type == BT_RAW) {
        // RawBuf::flush encode.rs:364-382: byte-align, LEN, NLEN, bytes
        const uint64_t byte0 = ((start + 3 + 7) >> 3);
        const uint32_t len = (uint32_t)bd.in_len;
        const ByteSrc src = make_src(in + bd.in_off, in_bytes - bd.in_off);
        const uint64_t total = 4 + (uint64_t)len;  // bytes to place starting at byte0
        // each lane assembles output dwords [w0, w1]
        const uint64_t wfirst = byte0 >> 2, wlast = (byte0 + total - 1) >> 2;
        for (uint64_t w = wfirst + threadIdx.x; w <= wlast; w += 256) {
            uint32_t v = 0;
            for (int k = 0; k < 4; ++k) {
                const uint64_t ob = w * 4 + k;
                if (ob < byte0 || ob >= byte0 + total) continue;
                const uint64_t r = ob - byte0;
                uint32_t byte;
                if (r == 0) byte = len & 0xFF;
                else if (r == 1) byte = (len >> 8) & 0xFF;
                else if (r == 2) byte = (~len) & 0xFF;
                else if (r == 3) byte = ((~len) >> 8) & 0xFF;
                else byte = src.load1(r - 4);
                v |= byte << (8 * k);
            }
            if (v) atomicOr(&out[w], v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// checksums.  A wavefront owns a 64 KiB region and sweeps it 4 KiB at a time: lane l digests bytes
// [64 l, 64 l + 64) of every 4 KiB slice (four 16-byte loads, well coalesced), then advances its CRC
// register over the 4032 bytes it skips with one multiplication by x^(8*4032) mod P.  At the end every
// lane advances to the region end and — CRC being linear — the 64 registers simply XOR together.
// Regions are combined by a second kernel (x^n mod P shifts for CRC-32, closed form for Adler-32).
constexpr uint32_t CK_SPAN = 65536;
constexpr uint32_t CRC_POLY = 0xEDB88320u;

__device__ __forceinline__ uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
    // product of two polynomials in the reflected CRC-32 domain, mod P
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;   // reflected: bit 31 is x^0
        a <<= 1;
        b = (b >> 1) ^ (CRC_POLY & (0u - (b & 1)));
    }
    return p;
}
// x^(8 * 2^k) mod P for k = 0..47, reflected representation (x^0 = 0x80000000), built at compile time:
// x^(8n) is then one multiplication per set bit of n instead of two per bit (a multiplication is a
// 32-step shift/xor loop, ~1000 cycles for a lone lane).
struct Xp8Table { uint32_t v[48]; };
constexpr uint32_t cx_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; ++i) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b >> 1) ^ (0xEDB88320u & (0u - (b & 1)));
    }
    return p;
}
constexpr Xp8Table cx_xp8_table() {
    Xp8Table t{};
    uint32_t sq = 0x00800000u;  // x^8
    for (int k = 0; k < 48; ++k) { t.v[k] = sq; sq = cx_mulmod(sq, sq); }
    return t;
}
__constant__ Xp8Table XP8 = cx_xp8_table();

// multiplication by the constant x^(8*4032) (a lane's step over the other lanes' bytes of a 4 KiB row) as
// four byte-indexed lookups: the product is linear in the register, T[k][b] = (b << 8k) * x^(8*4032)
struct AdvTable { uint32_t t[4][256]; };
constexpr uint32_t cx_xpow8n(uint64_t n) {
    uint32_t r = 0x80000000u, sq = 0x00800000u;
    while (n) { if (n & 1) r = cx_mulmod(r, sq); sq = cx_mulmod(sq, sq); n >>= 1; }
    return r;
}
constexpr AdvTable cx_adv_table() {
    AdvTable a{};
    const uint32_t adv = cx_xpow8n(4096 - 64);
    for (int k = 0; k < 4; ++k)
        for (uint32_t b = 0; b < 256; ++b) a.t[k][b] = cx_mulmod(b << (8 * k), adv);
    return a;
}
__constant__ AdvTable ADV = cx_adv_table();

__device__ __forceinline__ uint32_t gf2_xpow8n(uint64_t nbytes) {
    // x^(8*nbytes) mod P
    uint32_t r = 0x80000000u;
    bool first = true;
    for (uint32_t k = 0; nbytes; ++k, nbytes >>= 1)
        if (nbytes & 1) {
            const uint32_t f = XP8.v[k < 47 ? k : 47];
            r = first ? f : gf2_mulmod(r, f);
            first = false;
        }
    return r;
}

// checksum tables of a 256-thread workgroup: slice-by-4 CRC tables and the multiply-by-x^(8*4032) tables
__device__ __forceinline__ void ck_tables(uint32_t (*tab)[256], uint32_t (*advt)[256]) {
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (CRC_POLY & (0u - (c & 1)));
        tab[0][threadIdx.x] = c;
    }
    for (int k = 0; k < 4; ++k) advt[k][threadIdx.x] = ADV.t[k][threadIdx.x];
    __syncthreads();
    for (int t = 1; t < 4; ++t) {
        const uint32_t pv = tab[t - 1][threadIdx.x];
        tab[t][threadIdx.x] = (pv >> 8) ^ tab[0][pv & 0xFF];
        __syncthreads();
    }
}

// One wavefront, one region of at most CK_SPAN bytes: raw CRC register (init 0, no xorout — linear in the
// data), sum of bytes mod 65521 and sum of (rlen - i) * byte_i mod 65521.  The wavefront sweeps the region
// 4 KiB at a time (coalesced); a lane advances its register over the 4032 bytes of the other lanes with
// one multiplication by x^(8*4032) mod P.  Results are valid in every lane.
struct CkPartial { uint32_t crc, a, b; };
// MODE bit 0: CRC-32, bit 1: Adler-32 (a caller that needs one of them does not pay for the other: the kernel is
// bound by its instruction count, not by the 64 KiB it reads)
template <int MODE = 3>
__device__ __forceinline__ CkPartial ck_span_partial(const uint8_t *p, uint32_t rlen, uint32_t lane,
                                                     const uint32_t (*tab)[256], const uint32_t (*advt)[256]) {
    constexpr bool CRC = (MODE & 1) != 0, ADL = (MODE & 2) != 0;
    const ByteSrc src = make_src(p, rlen);
    // Invariant: `crc` is the raw register of this lane's bytes with zeros everywhere else, standing at
    // region offset `stand`.
    uint32_t crc = 0, stand = 0;
    uint32_t s1 = 0, s1_before = 0;   // sum of bytes
    uint64_t s2 = 0;                  // sum of (offset in region) * byte
    const uint32_t first = 64 * lane;
    // the loads of piece k+1 are issued before the (serially dependent) table walk over piece k starts: a wavefront's
    // 64 KiB are sixteen pieces, and without the look-ahead every piece waited for its own HBM round trip.  The loads are
    // branch-free (clamped addresses, the bytes behind the region masked off afterwards): a load under a lane-dependent
    // branch is followed by its own s_waitcnt, and sixteen of those in a row cost sixteen HBM round trips per piece
    // (0.73 us per load, measured: the kernel ran at 1.4 TB/s).
    const uint32_t shb = (uint32_t)src.shift;
    const bool wide = shb == 0 && ((uint64_t)src.w & 15) == 0;      // (uniform)
    auto load_piece = [&](uint32_t (&w)[16], uint32_t o) {
        if (wide) {
            const uint32_t last16 = (rlen - 1) >> 4;                 // the aligned 16 bytes that hold the last byte
            const uint4 *w16 = (const uint4 *)src.w;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                const uint32_t i = min((o >> 4) + q, last16);
                const uint4 v = w16[i];
                w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
            }
        } else {
            const uint32_t lastd = (rlen + shb - 1) >> 2;            // the aligned dword that holds the last byte
            const uint32_t i0 = (o + shb) >> 2;
            uint32_t raw[17];
#pragma unroll
            for (uint32_t q = 0; q < 17; ++q) raw[q] = src.w[min(i0 + q, lastd)];
#pragma unroll
            for (uint32_t q = 0; q < 16; ++q) w[q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], shb);
        }
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) {                           // bytes behind the region read as 0
            const int rem = (int)rlen - (int)(o + 4 * q);
            w[q] = rem >= 4 ? w[q] : rem <= 0 ? 0u : w[q] & ((1u << (8 * rem)) - 1u);
        }
    };
    auto walk_piece = [&](const uint32_t (&w)[16], uint32_t o) {
        if (CRC && o != first)                           // over the other lanes' 4032 bytes
            crc = advt[0][crc & 0xFF] ^ advt[1][(crc >> 8) & 0xFF] ^ advt[2][(crc >> 16) & 0xFF] ^ advt[3][crc >> 24];
        const uint32_t len = min(64u, rlen - o);         // only the last piece can be short
        uint32_t wsum = 0;       // sum of (offset in piece) * byte, fits 32 bits
#pragma unroll
        for (uint32_t q = 0; q < 16; ++q) {
            const uint32_t m = len > 4 * q ? min(4u, len - 4 * q) : 0u;
            if (m == 4) {
                if (CRC) {
                    const uint32_t x = crc ^ w[q];
                    crc = tab[3][x & 0xFF] ^ tab[2][(x >> 8) & 0xFF] ^ tab[1][(x >> 16) & 0xFF] ^ tab[0][x >> 24];
                }
                if (ADL) {
                    const uint32_t b0 = w[q] & 0xFF, b1 = (w[q] >> 8) & 0xFF, b2 = (w[q] >> 16) & 0xFF, b3 = w[q] >> 24;
                    s1 += b0 + b1 + b2 + b3;
                    wsum += 4 * q * (b0 + b1 + b2 + b3) + b1 + 2 * b2 + 3 * b3;
                }
            } else {
                for (uint32_t k = 0; k < m; ++k) {
                    const uint32_t byte = (w[q] >> (8 * k)) & 0xFF;
                    if (CRC) crc = (crc >> 8) ^ tab[0][(crc ^ byte) & 0xFF];
                    if (ADL) { s1 += byte; wsum += (4 * q + k) * byte; }
                }
            }
        }
        s2 += (uint64_t)o * (s1 - s1_before) + wsum;
        s1_before = s1;
        stand = o + len;
    };
    uint32_t wa[16], wb[16];
    load_piece(wa, first);
    for (uint32_t o = first; o < rlen; o += 2 * 4096) {
        load_piece(wb, o + 4096);
        walk_piece(wa, o);
        load_piece(wa, o + 2 * 4096);
        if (o + 4096 < rlen) walk_piece(wb, o + 4096);
    }
    CkPartial r;
    r.crc = 0; r.a = 0; r.b = 0;
    if (CRC) {
        uint32_t reg = first < rlen ? gf2_mulmod(crc, gf2_xpow8n(rlen - stand)) : 0u;
        for (int o = 32; o > 0; o >>= 1) reg ^= __shfl_xor(reg, o);
        r.crc = reg;
    }
    if (ADL) {
        uint64_t t1 = s1, t2 = s2 % 65521u;
        for (int o = 32; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o); t2 += __shfl_xor(t2, o); }
        r.a = (uint32_t)(t1 % 65521u);
        // sum over i of (rlen - i) * byte_i = rlen * S1 - S2
        r.b = (uint32_t)(((uint64_t)(rlen % 65521u) * r.a + 65521ull * 65521ull - (t2 % 65521u)) % 65521u);
    }
    return r;
}

template <int MODE>
__global__ __launch_bounds__(256) void checksum_span_kernel(const uint8_t *__restrict__ in,
                                                            uint64_t n, uint32_t span, uint32_t *__restrict__ crc_part,
                                                            uint32_t *__restrict__ a_part,
                                                            uint32_t *__restrict__ b_part, uint64_t region0) {
    __shared__ uint32_t tab[4][256];   // slice-by-4: four independent lookups per input dword
    __shared__ uint32_t advt[4][256];
    ck_tables(tab, advt);
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t region = region0 + (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (region0: the spans of a later part, launch_checksum_part)
    const uint64_t r0 = region * span;      // (span: 64 KiB, or 8 KiB for small inputs — ck_span(): a wavefront's serial share)
    if (r0 >= n) return;
    const uint32_t rlen = (uint32_t)min((uint64_t)span, n - r0);
    const CkPartial r = ck_span_partial<MODE>(in + r0, rlen, lane, tab, advt);
    if (lane == 0) { crc_part[region] = r.crc; a_part[region] = r.a; b_part[region] = r.b; }
}

// CRC-32 and Adler-32 of many independent byte ranges (batch decode): one wavefront per range, which folds
// its CK_SPAN pieces in order.  off[i] / len[i] are read with the given strides (in 8-byte units) so that the
// caller's own descriptor arrays can be used.
__global__ __launch_bounds__(256) void checksum_ranges_kernel(const uint8_t *__restrict__ data, uint32_t count,
                                                              const uint64_t *__restrict__ off, uint32_t off_stride,
                                                              const uint64_t *__restrict__ len, uint32_t len_stride,
                                                              uint32_t *__restrict__ crc_out,
                                                              uint32_t *__restrict__ adler_out) {
    __shared__ uint32_t tab[4][256];
    __shared__ uint32_t advt[4][256];
    ck_tables(tab, advt);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= count) return;
    const uint8_t *p = data + off[(uint64_t)i * off_stride];
    const uint64_t n = len[(uint64_t)i * len_stride];
    uint32_t crc = 0, a = 0, b = 0;
    const uint32_t xs = gf2_xpow8n(CK_SPAN);
    for (uint64_t o = 0; o < n; o += CK_SPAN) {
        const uint32_t sl = (uint32_t)min((uint64_t)CK_SPAN, n - o);
        const CkPartial r = ck_span_partial(p + o, sl, lane, tab, advt);
        crc = gf2_mulmod(crc, sl == CK_SPAN ? xs : gf2_xpow8n(sl)) ^ r.crc;
        b = (uint32_t)((b + r.b + (sl % 65521u) * (uint64_t)a) % 65521u);   // B += b2 + len2 * A_before
        a = (a + r.a) % 65521u;
    }
    if (lane == 0) {
        // raw register of the data; apply init 0xFFFFFFFF over n bytes and the final xor; Adler with A0 = 1
        crc_out[i] = crc ^ gf2_mulmod(0xFFFFFFFFu, gf2_xpow8n(n)) ^ 0xFFFFFFFFu;
        const uint32_t A = (1u + a) % 65521u;
        const uint32_t B = (uint32_t)((n % 65521u + b) % 65521u);
        adler_out[i] = (B << 16) | A;
    }
}

// one workgroup folds all span partials.  Round 4: no tree.  A lane folds its consecutive spans (Horner), moves the result
// to the END of the data with one x^(8 * bytes behind it) — CRC being linear, the lanes' registers then simply XOR
// together — and the Adler sums likewise become plain sums: a byte at offset i weighs (n - i) in B, which is its weight
// inside the lane's piece plus the bytes behind the piece.  Lane 1023 holds the init term 0xFFFFFFFF * x^(8n).
// (The tree combined pairs level by level: ten levels of dependent multiplications and barriers, 44 us for 4096 spans.)
__global__ __launch_bounds__(1024) void checksum_combine_kernel(const uint32_t *__restrict__ crc_part,
                                                                const uint32_t *__restrict__ a_part,
                                                                const uint32_t *__restrict__ b_part,
                                                                uint64_t n, uint32_t span,
                                                                EncodeResult *__restrict__ res) {
    __shared__ uint32_t s_crc[16], s_a[16], s_b[16];
    const uint64_t nspans = div_up(n, span);
    const uint64_t per = div_up(nspans ? nspans : 1, 1023);
    const uint32_t t = threadIdx.x;
    uint32_t crc = 0, a = 0, b = 0;
    if (t < 1023) {
        const uint64_t lo = (uint64_t)t * per, hi = min(nspans, lo + per);
        if (lo < hi) {
            uint64_t len = 0;
            const uint32_t xs = gf2_xpow8n(span);
            for (uint64_t s = lo; s < hi; ++s) {
                const uint64_t sl = min((uint64_t)span, n - s * span);
                const uint32_t sh = sl == span ? xs : gf2_xpow8n(sl);
                crc = gf2_mulmod(crc, sh) ^ crc_part[s];
                // Adler: A += a2 ; B += b2 + len2 * A_before
                b = (uint32_t)((b + b_part[s] + (sl % 65521u) * (uint64_t)a) % 65521u);
                a = (a + a_part[s]) % 65521u;
                len += sl;
            }
            const uint64_t after = n - (lo * span + len);
            if (after) crc = gf2_mulmod(crc, gf2_xpow8n(after));
            b = (uint32_t)((b + (after % 65521u) * (uint64_t)a) % 65521u);
        }
    } else {
        crc = gf2_mulmod(0xFFFFFFFFu, gf2_xpow8n(n));   // init 0xFFFFFFFF carried over n bytes
    }
    for (int o = 32; o > 0; o >>= 1) {                  // (a, b < 65521: 64 of them stay far below 2^32)
        crc ^= __shfl_xor(crc, o);
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    if ((t & 63) == 0) { s_crc[t >> 6] = crc; s_a[t >> 6] = a; s_b[t >> 6] = b; }
    __syncthreads();
    if (t == 0) {
        uint32_t c = 0;
        uint64_t A = 0, B = 0;
        for (int w = 0; w < 16; ++w) { c ^= s_crc[w]; A += s_a[w]; B += s_b[w]; }
        res->crc32 = c ^ 0xFFFFFFFFu;                   // the final xor
        // Adler with A0 = 1: A = 1 + sum ; B = n*1 + sum_b
        const uint32_t Af = (uint32_t)((1u + A) % 65521u);
        const uint32_t Bv = (uint32_t)((n % 65521u + B) % 65521u);
        res->adler32 = (Bv << 16) | Af;
    }
}

// container header / trailer bytes (host-built, a few bytes) → output, via atomicOr on zeroed words
__global__ void put_bytes_kernel(const uint8_t *__restrict__ bytes, uint32_t n, uint64_t at_byte,
                                 uint32_t *__restrict__ out) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint64_t ob = at_byte + i;
        atomicOr(&out[ob >> 2], (uint32_t)bytes[i] << (8 * (ob & 3)));
    }
}
// shard concatenation: the byte a shard shares with the shard in front of it
__global__ void or_byte_kernel(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src) {
    if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = (uint8_t)(dst[0] | src[0]);
}
int launch_or_byte(hipStream_t st, uint8_t *dst, const uint8_t *src) {
    hipLaunchKernelGGL(or_byte_kernel, dim3(1), dim3(64), 0, st, dst, src);
    const hipError_t e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}
// trailer from the device-side checksum result (gzip.rs:114-121 CRC-32 LE + ISIZE LE;
// zlib.rs:630-639 Adler-32 BE), placed at the byte after the last DEFLATE bit
__global__ void trailer_kernel(int format, uint32_t isize, uint64_t out_base_bit,
                               EncodeResult *__restrict__ res, uint32_t *__restrict__ out) {
    if (threadIdx.x != 0 || res->status != 0) return;
    const uint64_t at = (res->end_bit - out_base_bit + 7) >> 3;
    uint8_t t[8];
    uint32_t nt = 0;
    if (format == 2) {
        const uint32_t c = res->crc32;
        t[0] = c; t[1] = c >> 8; t[2] = c >> 16; t[3] = c >> 24;
        t[4] = isize; t[5] = isize >> 8; t[6] = isize >> 16; t[7] = isize >> 24;
        nt = 8;
    } else if (format == 1) {
        const uint32_t a = res->adler32;
        t[0] = a >> 24; t[1] = a >> 16; t[2] = a >> 8; t[3] = a;
        nt = 4;
    }
    for (uint32_t i = 0; i < nt; ++i) {
        const uint64_t ob = at + i;
        atomicOr(&out[ob >> 2], (uint32_t)t[i] << (8 * (ob & 3)));
    }
    res->out_bytes = at + nt;
}

// ------------------------------------------------------------------------------------------------
// launchers (called from lfx_api.cpp)
#define LFX_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)

int launch_match(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks,
                 const SegDesc *segs, uint32_t nsegs, uint32_t window, uint32_t max_len, uint32_t *md, uint64_t *dbg) {
    if (nsegs == 0) return 0;
    const size_t lds = MATCH_LDS;
    // (a function attribute is per device: one flag per device ordinal)
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)lz77_match_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[dev_ & 63] = true;
    }
    hipLaunchKernelGGL(lz77_match_kernel, dim3(nsegs), dim3(MATCH_THREADS), lds, st, in, in_bytes,
                       chunks, segs, window, max_len, md, dbg);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_chunk_maps(hipStream_t st, const ChunkDesc *chunks, uint32_t nchunks, uint64_t ntiles, uint32_t nsegs,
                      uint32_t *tile_map, uint32_t *seg_map, ZeroSpan z0, ZeroSpan z1, ZeroSpan z2) {
    if (nchunks == 0) {
        // (no kernel: plain fills)
        for (const ZeroSpan &z : {z0, z1, z2})
            if (z.n && hipMemsetAsync(z.p, 0, 4ull * z.n, st) != hipSuccess) return (int)hipGetLastError();
        return 0;
    }
    // (few chunks = long chunks: several workgroups share one)
    uint32_t split = nchunks >= 256 ? 1 : 256 / nchunks + 1;
    if (split > 64) split = 64;
    hipLaunchKernelGGL(chunk_maps_kernel, dim3(nchunks, split), dim3(256), 0, st, chunks, nchunks, ntiles, nsegs, tile_map, seg_map, z0, z1, z2);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_histogram(hipStream_t st, const ChunkDesc *chunks, uint32_t nchunks, uint32_t split,
                     const uint32_t *codes, const uint32_t *ncodes, uint32_t *hist) {
    if (nchunks == 0) return 0;
    hipLaunchKernelGGL(histogram_kernel, dim3(nchunks, split), dim3(256), 0, st, chunks, codes, ncodes, hist);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_huffman(hipStream_t st, const BlockDesc *blocks, uint32_t nblocks, const uint32_t *hist,
                   BlockCodes *bc, uint64_t *dbg) {
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(huffman_kernel, dim3(nblocks), dim3(HUFF_THREADS), 0, st, blocks, hist, bc, dbg);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_offsets(hipStream_t st, const BlockDesc *blocks, uint32_t nblocks, const BlockCodes *bc,
                   uint64_t start_bit, uint64_t cap_bits, uint64_t *block_start, EncodeResult *res) {
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(256), 0, st, blocks, nblocks, bc, start_bit,
                       cap_bits, block_start, res);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_offsets_batch(hipStream_t st, const BatchStream *streams, uint32_t count, const BlockDesc *blocks, const BlockCodes *bc,
                         uint32_t hdr_len, uint32_t trailer_len, uint64_t *block_start, uint64_t *stream_end, int32_t *status,
                         EncodeResult *res) {
    if (!count) return 0;
    hipLaunchKernelGGL(offsets_batch_kernel, dim3((count + 255) / 256), dim3(256), 0, st, streams, count, blocks, bc, hdr_len,
                       trailer_len, block_start, stream_end, status, res);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_frame_batch(hipStream_t st, int format, const BatchStream *streams, uint32_t count, const uint8_t *hdr, uint32_t hdr_len,
                       const uint64_t *stream_end, const uint32_t *crc, const uint32_t *adler, const EncodeResult *res, uint32_t *out,
                       uint64_t *out_len) {
    if (!count) return 0;
    hipLaunchKernelGGL(frame_batch_kernel, dim3((count + 255) / 256), dim3(256), 0, st, format, streams, count, hdr, hdr_len,
                       stream_end, crc, adler, res, out, out_len);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_pack(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks,
                uint32_t nchunks, const BlockDesc *blocks, uint32_t nblocks, uint64_t ntiles,
                const uint32_t *codes, const uint32_t *ncodes, const BlockCodes *bc,
                const uint64_t *block_start, uint32_t *tile_bits, uint64_t *tile_start,
                const EncodeResult *res, uint64_t out_base_bit, uint32_t *out, const uint32_t *tile_map) {
    if (ntiles) {
        const uint32_t ngroups = (uint32_t)div_up(ntiles, PACK_TPW);
        hipLaunchKernelGGL(tile_bits_kernel, dim3(ngroups), dim3(PACK_THREADS), 0, st, chunks,
                           nchunks, blocks, codes, ncodes, bc, tile_bits, tile_map, ntiles);
        LFX_LAUNCH_CHECK();
        hipLaunchKernelGGL(tile_scan_kernel, dim3(nblocks), dim3(ntiles / (nblocks ? nblocks : 1) > 2048 ? 1024 : 256), 0, st, chunks, blocks, bc,
                           block_start, tile_bits, ntiles, nchunks, tile_start);
        LFX_LAUNCH_CHECK();
        hipLaunchKernelGGL(pack_kernel, dim3(ngroups), dim3(PACK_THREADS), 0, st, chunks,
                           nchunks, blocks, codes, ncodes, bc, tile_start, res, out_base_bit, out, tile_map, ntiles);
        LFX_LAUNCH_CHECK();
    }
    if (nblocks) {
        hipLaunchKernelGGL(block_header_kernel, dim3(nblocks), dim3(256), 0, st, in, in_bytes, blocks,
                           bc, block_start, res, out_base_bit, out);
        LFX_LAUNCH_CHECK();
    }
    return 0;
}
// the spans of part `part` of `nparts` (whole groups of four spans); part == nparts - 1 also folds ALL spans (it must be
// ordered behind the other parts).  launch_checksum = one part.  The encoder runs the first part beside the parse's
// chaining kernels and the last beside the Huffman kernel (lfx_api.cpp): each leaves most of the GPU idle for about as
// long as half of the sweep takes.
int launch_checksum_part(hipStream_t st, const uint8_t *in, uint64_t n, uint32_t *crc_part, uint32_t *a_part, uint32_t *b_part,
                         EncodeResult *res, int mode, uint32_t part, uint32_t nparts) {
    const uint32_t span = ck_span(n);
    const uint64_t nspans = div_up(n, span);
    const uint64_t ngroups = div_up(nspans, 4);
    const uint64_t g0 = ngroups * part / nparts, g1 = ngroups * (part + 1) / nparts;
    if (g1 > g0) {
        const dim3 grid((uint32_t)(g1 - g0));
        if (mode == 1) hipLaunchKernelGGL(checksum_span_kernel<1>, grid, dim3(256), 0, st, in, n, span, crc_part, a_part, b_part, 4 * g0);
        else if (mode == 2) hipLaunchKernelGGL(checksum_span_kernel<2>, grid, dim3(256), 0, st, in, n, span, crc_part, a_part, b_part, 4 * g0);
        else hipLaunchKernelGGL(checksum_span_kernel<3>, grid, dim3(256), 0, st, in, n, span, crc_part, a_part, b_part, 4 * g0);
        LFX_LAUNCH_CHECK();
    }
    if (part + 1 == nparts) {
        hipLaunchKernelGGL(checksum_combine_kernel, dim3(1), dim3(1024), 0, st, crc_part, a_part, b_part, n, span, res);
        LFX_LAUNCH_CHECK();
    }
    return 0;
}
int launch_checksum(hipStream_t st, const uint8_t *in, uint64_t n, uint32_t *crc_part,
                    uint32_t *a_part, uint32_t *b_part, EncodeResult *res, int mode) {
    return launch_checksum_part(st, in, n, crc_part, a_part, b_part, res, mode, 0, 1);
}
int launch_checksum_ranges(hipStream_t st, const uint8_t *data, uint32_t count, const uint64_t *off,
                           uint32_t off_stride, const uint64_t *len, uint32_t len_stride, uint32_t *crc, uint32_t *adler) {
    if (!count) return 0;
    hipLaunchKernelGGL(checksum_ranges_kernel, dim3((count + 3) / 4), dim3(256), 0, st, data, count, off, off_stride, len,
                       len_stride, crc, adler);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_put_bytes(hipStream_t st, const uint8_t *d_bytes, uint32_t n, uint64_t at_byte, uint32_t *out) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(put_bytes_kernel, dim3(1), dim3(256), 0, st, d_bytes, n, at_byte, out);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_trailer(hipStream_t st, int format, uint32_t isize, uint64_t out_base_bit,
                   EncodeResult *res, uint32_t *out) {
    hipLaunchKernelGGL(trailer_kernel, dim3(1), dim3(64), 0, st, format, isize, out_base_bit, res, out);
    LFX_LAUNCH_CHECK();
    return 0;
}

}  // namespace lfx
