// lfx_inflate_fast.hip — lane-parallel inflate of ONE large DEFLATE stream (gfx950).
//
// The reference decodes a stream strictly serially (src/deflate/decode.rs:136-164, one symbol at a
// time through symbol.rs:193-244).  Here every candidate block (found by find_blocks_*) is scanned by
// a 1024-lane workgroup (K1): the block's bit range is cut into 1024 slices, every lane starts decoding
// at its slice start *speculatively* (Huffman streams self-synchronise within a few symbols), and the
// exits are chained from lane 0 — whose start is exact — until nothing changes.  Validated lanes then
// re-decode their slices into a code stream (K2; same word format as the encoder's: (val << 16) | dist)
// and the block is cut into units no back-reference crosses; one wavefront per unit materialises
// literals and back-references through a 36 KiB LDS ring (K3; Lz77Decoder::decode,
// libflate_lz77/src/lib.rs:164-194).  Any anomaly (error, cross-block reference, unchained block)
// makes the host fall back to the exact serial kernel, which reproduces the reference's error kinds
// and partial output.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "lfx_common.h"
#include "lfx_decode.h"

namespace lfx {

__constant__ uint16_t f_len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                        31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t f_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                        2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t f_dist_base[30] = {1,    2,    3,    4,    5,    7,    9,    13,    17,    25,
                                         33,   49,   65,   97,   129,  193,  257,  385,   513,   769,
                                         1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t f_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                         6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t f_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
constexpr uint32_t f_clen_order_c(uint32_t k) {
    constexpr uint8_t o[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return o[k];
}
// BITWIDTH_CODE_ORDER (symbol.rs:16-18) without a memory lookup: 5 bits per entry
__device__ __forceinline__ uint32_t clen_order(uint32_t k) {
    // entries 0..11 in lo, 12..18 in hi
    const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                        10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    return k < 12 ? (uint32_t)(lo >> (5 * k)) & 31 : (uint32_t)(hi >> (5 * (k - 12))) & 31;
}

// table entry: bits 0-3 code width (0 = no direct code), bits 4-5 kind (0 literal, 1 length, 2 EOB,
// 3 long-code prefix), bits 6-10 extra-bit count, bits 16-31 value (literal byte / length base /
// distance base).  0 = unassigned.
constexpr uint32_t K_LIT = 0, K_LEN = 1, K_EOB = 2, K_LONG = 3;
constexpr uint32_t E_LONG = (K_LONG << 4);

// per-lane bit source over global memory: 64-bit window + a prefetched 4-dword FIFO
struct LaneBits {
    gptr_u32 w;          // 4-byte aligned base (global address space)
    uint64_t wlast;      // last readable dword index
    uint64_t widx;       // next dword to fetch
    uint64_t pos;        // stream bit position of buf bit 0 (relative to the stream's first byte)
    uint64_t buf;
    uint32_t nb;
    uint32_t c0, c1, c2, c3, cn;
    uint32_t n0, n1, n2, n3;
    // clamped, branch-free: words past the end repeat the last word (callers stop by bit position)
    __device__ __forceinline__ uint32_t ld(uint64_t i) const { return w[i < wlast ? i : wlast]; }
    __device__ __forceinline__ void init(const uint8_t *base, uint64_t nbytes, uint64_t bitpos) {
        const uint64_t a = (uint64_t)base;
        w = (gptr_u32)(a & ~3ull);
        const uint64_t sh = (a & 3) * 8;
        wlast = ((a & 3) + nbytes + 3) / 4;
        wlast = wlast ? wlast - 1 : 0;
        const uint64_t abs = bitpos + sh;
        widx = abs >> 5;
        const uint32_t off = (uint32_t)abs & 31;
        c0 = ld(widx); c1 = ld(widx + 1); c2 = ld(widx + 2); c3 = ld(widx + 3);
        n0 = ld(widx + 4); n1 = ld(widx + 5); n2 = ld(widx + 6); n3 = ld(widx + 7);
        widx += 8;
        buf = (uint64_t)(c0 >> off);
        nb = 32 - off;
        c0 = c1; c1 = c2; c2 = c3; cn = 3;
        pos = bitpos;
    }
    __device__ __forceinline__ void refill() {
        if (nb <= 32) {
            buf |= (uint64_t)c0 << nb;
            nb += 32;
            c0 = c1; c1 = c2; c2 = c3;
            if (--cn == 0) {
                c0 = n0; c1 = n1; c2 = n2; c3 = n3; cn = 4;
                n0 = ld(widx); n1 = ld(widx + 1); n2 = ld(widx + 2); n3 = ld(widx + 3);
                widx += 4;
            }
        }
    }
    __device__ __forceinline__ void skip(uint32_t k) { buf >>= k; nb -= k; pos += k; }
};

constexpr uint32_t LIT_BITS = 12, DIST_BITS = 10;
struct FastTabs {
    uint32_t lit[1u << LIT_BITS];
    uint32_t dist[1u << DIST_BITS];
    uint32_t lit_info[288];   // entry of every literal/length symbol without its width
    uint32_t dist_info[32];
    uint16_t lit_sorted[288];
    uint16_t dist_sorted[32];
    uint16_t lit_count[16];
    uint16_t dist_count[16];
};

// canonical walk for codes longer than the primary table (rare)
__device__ __forceinline__ uint32_t long_decode(const uint16_t *count, const uint16_t *sorted, uint64_t bits,
                                                uint32_t &width) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t w = 1; w <= 15; ++w) {
        code |= (uint32_t)(bits >> (w - 1)) & 1;
        const uint32_t cnt = count[w];
        if (code < first + cnt) { width = w; return sorted[index + (code - first)]; }
        index += cnt;
        first = (first + cnt) << 1;
        code <<= 1;
    }
    width = 0;
    return 0xFFFF;
}

__device__ __forceinline__ uint32_t lit_entry_of(uint32_t s, uint32_t w) {
    if (s < 256) return w | (K_LIT << 4) | (s << 16);
    if (s == 256) return w | (K_EOB << 4);
    if (s < 286) return w | (K_LEN << 4) | ((uint32_t)f_len_extra[s - 257] << 6) | ((uint32_t)f_len_base[s - 257] << 16);
    return 0;  // 286 / 287 must not occur (symbol.rs:216-223): treated as undecodable here
}
__device__ __forceinline__ uint32_t dist_entry_of(uint32_t s, uint32_t w) {
    if (s >= 30) return 0;
    return w | ((uint32_t)f_dist_extra[s] << 6) | ((uint32_t)f_dist_base[s] << 16);
}

// canonical walk limited to `maxw` bits (stream bit k of the code at bit k of `bits`)
__device__ __forceinline__ uint32_t short_decode(const uint16_t *count, const uint16_t *sorted, uint32_t bits,
                                                 uint32_t maxw, uint32_t &width) {
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t w = 1; w <= maxw; ++w) {
        code |= (bits >> (w - 1)) & 1;
        const uint32_t cnt = count[w];
        if (code < first + cnt) { width = w; return sorted[index + (code - first)]; }
        index += cnt;
        first = (first + cnt) << 1;
        code <<= 1;
    }
    width = 0;
    return 0xFFFF;
}

// workgroup-wide: canonical tables from code widths.  Returns false when over-subscribed.
// Every step is spread over the workgroup: width histogram by LDS atomics, symbol order by rank,
// and the primary tables entry by entry (each entry walks the canonical code of its own index).
__device__ bool build_fast(FastTabs &T, const uint8_t *lw, uint32_t nl, const uint8_t *dw, uint32_t nd,
                           uint32_t tid, uint32_t nthreads) {
    __shared__ uint32_t s_cnt[2][16], s_off[2][16], s_bad, s_long[2];
    if (tid < 32) s_cnt[tid >> 4][tid & 15] = 0;
    for (uint32_t i = tid; i < 288; i += nthreads) T.lit_info[i] = lit_entry_of(i, 0);
    for (uint32_t i = tid; i < 32; i += nthreads) T.dist_info[i] = dist_entry_of(i, 0);
    __syncthreads();
    for (uint32_t s = tid; s < nl + nd; s += nthreads) {
        const uint32_t t = s >= nl;
        const uint32_t w = t ? dw[s - nl] : lw[s];
        if (w) atomicAdd(&s_cnt[t][w], 1u);
    }
    __syncthreads();
    if (tid < 2) {
        const uint32_t t = tid;
        uint16_t *cnt = t ? T.dist_count : T.lit_count;
        const uint32_t pri = t ? DIST_BITS : LIT_BITS;
        uint32_t off = 0, nlong = 0;
        int left = 1;
        bool bad = false;
        cnt[0] = 0;
        for (uint32_t w = 1; w <= 15; ++w) {
            const uint32_t c = s_cnt[t][w];
            cnt[w] = (uint16_t)c;
            s_off[t][w] = off;
            left = (left << 1) - (int)c;
            if (left < 0) bad = true;
            off += c;
            if (w > pri) nlong += c;
        }
        s_long[t] = nlong;
        if (t == 0) s_bad = 0;
        __builtin_amdgcn_wave_barrier();
        if (bad) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) return false;
    for (uint32_t s = tid; s < nl + nd; s += nthreads) {
        const uint32_t t = s >= nl;
        const uint32_t sym = t ? s - nl : s;
        const uint8_t *bw = t ? dw : lw;
        const uint32_t w = bw[sym];
        if (w == 0) continue;
        // symbols of the same width in front of this one.  Literal/length widths (the array is dword-aligned): four at a
        // time (round 4 — the byte loop was up to 285 LDS reads per lane)
        uint32_t rank = 0;
        if (t == 0 && ((uint32_t)(uintptr_t)lw & 3) == 0) {
            const uint32_t *w32 = (const uint32_t *)lw;
            const uint32_t pat = w * 0x01010101u, full = sym >> 2;
            auto zero_bytes = [](uint32_t x) { return (uint32_t)__builtin_popcount(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu)); };
#pragma unroll 8
            for (uint32_t q = 0; q < full; ++q) rank += zero_bytes(w32[q] ^ pat);
            if (sym & 3) rank += zero_bytes((w32[full] ^ pat) | (0xFFFFFFFFu << (8 * (sym & 3))));
        } else {
            for (uint32_t q = 0; q < sym; ++q) rank += bw[q] == w;
        }
        (t ? T.dist_sorted : T.lit_sorted)[s_off[t][w] + rank] = (uint16_t)sym;
    }
    __syncthreads();
    // the primary tables, entry by entry: each entry walks the canonical code of its own index.  The per-width counts are
    // taken into registers once (round 4: the walk read them from LDS, a dependent round trip per width and entry)
    {
        uint32_t cr[2][8];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            cr[0][k] = ((const uint32_t *)T.lit_count)[k];
            cr[1][k] = ((const uint32_t *)T.dist_count)[k];
        }
        for (uint32_t i = tid; i < (1u << LIT_BITS) + (1u << DIST_BITS); i += nthreads) {
            const uint32_t t = i >= (1u << LIT_BITS);
            const uint32_t idx = t ? i - (1u << LIT_BITS) : i;
            const uint32_t maxw = t ? DIST_BITS : LIT_BITS;
            uint32_t w = 0, pos = 0, code = 0, first = 0, index = 0;
#pragma unroll
            for (uint32_t ww = 1; ww <= LIT_BITS; ++ww) {
                const uint32_t cnt = ((t ? cr[1][ww >> 1] : cr[0][ww >> 1]) >> (16 * (ww & 1))) & 0xFFFFu;
                code |= (idx >> (ww - 1)) & 1;
                const bool hit = w == 0 && ww <= maxw && code < first + cnt;
                w = hit ? ww : w;
                pos = hit ? index + (code - first) : pos;
                index += cnt;
                first = (first + cnt) << 1;
                code <<= 1;
            }
            uint32_t e;
            if (w) {
                const uint32_t sym = (t ? T.dist_sorted : T.lit_sorted)[pos];
                e = t ? (sym < 30 ? (T.dist_info[sym] | w) : 0) : (sym < 286 ? (T.lit_info[sym] | w) : 0);
            } else e = s_long[t] ? E_LONG : 0;             // a longer code may start with these bits
            (t ? T.dist : T.lit)[idx] = e;
        }
    }
    __syncthreads();
    return true;
}

// rare path, kept out of line so the hot loop stays small: a code longer than the primary table
__device__ __noinline__ uint32_t long_lookup(const uint16_t *count, const uint16_t *sorted, const uint32_t *info,
                                            uint32_t nsym, uint64_t bits) {
    uint32_t w = 0;
    const uint32_t s = long_decode(count, sorted, bits, w);
    if (w == 0 || s >= nsym) return 0;
    return info[s] | w;
}

// Bit source of lane_decode_fifo (rounds 1-4; still the scan kernel's and the batch path's: 64 vector registers, two
// workgroups per CU): 64-bit window `buf` (nb valid bits), a FIFO of up to five dwords in
// registers (q0 first) and four more dwords in flight (x0..x3).  The in-flight dwords are merged into
// the FIFO only between symbols, *before* the next four loads are issued, so the loads write straight
// into dead registers and the only wait for them sits one reload period (~8 symbols) later.
struct FastBits {
    gptr_u32 w;
    uint64_t wlast, widx;
    uint64_t buf;
    uint32_t nb, used;              // used = bits consumed since init
    uint32_t q0, q1, q2, q3, q4, qn;
    uint32_t x0, x1, x2, x3;
    __device__ __forceinline__ uint32_t ld(uint64_t i) const { return w[i < wlast ? i : wlast]; }
    __device__ __forceinline__ void init(const uint8_t *base, uint64_t nbytes, uint64_t bitpos) {
        const uint64_t a = (uint64_t)base;
        w = (gptr_u32)(a & ~3ull);
        wlast = ((a & 3) + nbytes + 3) / 4;
        wlast = wlast ? wlast - 1 : 0;
        const uint64_t abs = bitpos + (a & 3) * 8;
        widx = abs >> 5;
        const uint32_t off = (uint32_t)abs & 31;
        const uint32_t f = ld(widx);
        q0 = ld(widx + 1); q1 = ld(widx + 2); q2 = ld(widx + 3); q3 = ld(widx + 4); q4 = 0; qn = 4;
        // The FIFO dwords are pinned (an empty asm that reads them) before the in-flight ones are loaded: without it the
        // compiler issues the q loads LAST, the first use of the FIFO inside the symbol loop then needs vmcnt(0), and —
        // the wait being one instruction for both loop edges — every symbol waited for the four prefetch loads (and the
        // code stores) just issued: the look-ahead was dead and every reload period paid a full memory round trip.
        uint32_t f_ = f;
        asm volatile("" : "+v"(f_), "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        x0 = ld(widx + 5); x1 = ld(widx + 6); x2 = ld(widx + 7); x3 = ld(widx + 8);
        widx += 9;
        buf = (uint64_t)(f_ >> off);
        nb = 32 - off;
        used = 0;
    }
    __device__ __forceinline__ void append() {   // needs qn >= 1
        // branch-free: a divergent branch here costs an exec-mask round trip and a register copy per FIFO slot
        const bool need = nb <= 32;
        const uint64_t add = (uint64_t)q0 << (nb & 63);
        buf |= need ? add : 0ull;
        nb += need ? 32u : 0u;
        q0 = need ? q1 : q0; q1 = need ? q2 : q1; q2 = need ? q3 : q2; q3 = need ? q4 : q3;
        qn -= need ? 1u : 0u;
    }
    __device__ __forceinline__ void skip(uint32_t k) { buf >>= k; nb -= k; used += k; }
    // reload in two halves, so that the caller's code stores go between them: the wait for the in-flight dwords then
    // sits in FRONT of the stores (vmcnt counts loads and stores alike, in order: behind them it waits for their
    // acknowledgements too)
    __device__ __forceinline__ void take() {     // qn is 0 or 1 here
        if (qn == 0) { q0 = x0; q1 = x1; q2 = x2; q3 = x3; qn = 4; }
        else { q1 = x0; q2 = x1; q3 = x2; q4 = x3; qn = 5; }
        __builtin_amdgcn_sched_barrier(0);       // keep the caller's stores and the loads of issue() below the moves above
    }
    __device__ __forceinline__ void issue() {
        x0 = ld(widx); x1 = ld(widx + 1); x2 = ld(widx + 2); x3 = ld(widx + 3);
        widx += 4;
    }
    __device__ __forceinline__ void reload() { take(); issue(); }
};

// Hot bit source of lane_decode (round 5): the lane's next dwords of the stream in a RING IN LDS — eight dwords per lane plus
// copies of slots 0..2 behind slot 7, so that FOUR consecutive dwords are two ds_read2_b32 at any position — and four more
// dwords in flight in registers (x0..x3).  A symbol's bits are `v_alignbit(hi, lo, position)` on a register window of four
// dwords: no 64-bit shift (a quarter of the vector rate), no register FIFO to shuffle (five v_cndmask per appended dword).
// The literal/length code with its extra bits (<= 20 bits) and the distance code with its extra bits (<= 28) each fit one
// 32-bit window, and a symbol's two windows lie inside the four dwords read at the PREVIOUS symbol's distance position: the
// ring read for the next symbol is issued one symbol ahead, off the critical path.  What that path is made of — measured,
// profiles/r05_pmc_counters.csv: a wavefront of the two Huffman kernels waits 68 % of its cycles, an LDS round trip costs it
// about 240 cycles with sixteen wavefronts on the CU, the vector unit is 32 % busy — is the number of DEPENDENT LDS round trips
// per symbol: two (the literal/length table, the distance table).  And a 128-byte line of the stream is fetched once per
// sixteen-byte piece by the ONE lane that decodes it, not reloaded around the L2 (blk_emit: 1.8 GB of fabric traffic for a
// 130 MB stream until round 4).
#ifndef LFX_RING_DW
#define LFX_RING_DW 16        // dwords of a lane's ring (16 / 8 is the geometry the suite runs; 12 / 4 compiles and HANGS the storing scan — HISTORY.md, round 6)
#endif
#ifndef LFX_RING_FILL
#define LFX_RING_FILL 8       // dwords per refill (in flight in registers meanwhile)
#endif
#ifndef LFX_RING_PIPE
#define LFX_RING_PIPE 1       // 1: a four-dword register window, the ring read issued a symbol ahead; 0: two ring reads per symbol
#endif
constexpr uint32_t RING_DW = LFX_RING_DW, RING_FILL = LFX_RING_FILL, RING_DUP = LFX_RING_PIPE ? 3 : 1;
constexpr uint32_t RING_STRIDE = RING_DW + RING_DUP + ((RING_DW + RING_DUP) % 2 == 0 ? 1 : 0);   // (odd stride: the lanes' slots k fall into different banks)
constexpr uint32_t RING_AHEAD = LFX_RING_PIPE ? 5 : 3;        // dwords from rel / 32 on that a symbol may read
static_assert(RING_FILL == 4 || RING_FILL == 8, "register sets below");
static_assert(RING_DW % RING_FILL == 0 && RING_DW >= RING_FILL + RING_AHEAD && RING_DUP <= RING_FILL, "ring geometry");
struct RingBits {
    gptr_u32 w;
    uint64_t wlast, widx;           // last readable dword, next dword to fetch
    uint32_t *ring;                 // this lane's RING_STRIDE dwords of LDS
    uint32_t rel, off0;             // bit position since the first ring dword; its value at init (used = rel - off0)
    uint32_t filled;                // dwords written to the ring since init (a multiple of RING_FILL)
    uint64_t r01, r23;              // (LFX_RING_PIPE) the register window: ring dwords rb .. rb + 3, loaded by inline asm
    uint32_t rb;
    uint32_t x[RING_FILL];
    __device__ __forceinline__ uint32_t ld(uint64_t i) const { return w[i < wlast ? i : wlast]; }   // (clamped: callers stop by bit position)
    __device__ __forceinline__ void init(const uint8_t *base, uint64_t nbytes, uint64_t bitpos, uint32_t *lds_ring) {
        const uint64_t a = (uint64_t)base;
        w = (gptr_u32)(a & ~3ull);
        wlast = ((a & 3) + nbytes + 3) / 4;
        wlast = wlast ? wlast - 1 : 0;
        const uint64_t abs = bitpos + (a & 3) * 8;
        widx = abs >> 5;
        ring = lds_ring;
        uint32_t f[RING_DW];
#pragma unroll
        for (uint32_t k = 0; k < RING_DW; ++k) f[k] = ld(widx + k);
        // (the ring's dwords are pinned before the in-flight loads are issued: the compiler otherwise issues THESE loads last,
        //  and their first use waits for the prefetch as well)
#pragma unroll
        for (uint32_t k = 0; k < RING_DW; ++k) asm volatile("" : "+v"(f[k]));
#pragma unroll
        for (uint32_t k = 0; k < RING_FILL; ++k) x[k] = ld(widx + RING_DW + k);
        widx += RING_DW + RING_FILL;
#pragma unroll
        for (uint32_t k = 0; k < RING_DW; ++k) ring[k] = f[k];
#pragma unroll
        for (uint32_t k = 0; k < RING_DUP; ++k) ring[RING_DW + k] = f[k];
        filled = RING_DW;
        rel = off0 = (uint32_t)abs & 31;
        r01 = (uint64_t)f[0] | (uint64_t)f[1] << 32; r23 = (uint64_t)f[2] | (uint64_t)f[3] << 32; rb = 0;
    }
    __device__ __forceinline__ uint32_t used() const { return rel - off0; }
    // a whole symbol can be decoded at `rel`: every dword it may read is in the ring
    __device__ __forceinline__ bool room() const { return (rel >> 5) + RING_AHEAD <= filled; }
    // the 32 bits at position `at`
    // (LFX_RING_PIPE: the compiler must not see the window's LDS loads — it sinks a load it sees to its first use, an iteration
    //  later, and the look-ahead is gone — so they are inline asm, and an explicit wait that names the registers sits in front
    //  of their first use; by then the distance-table read issued behind them has been waited for: the wait is free)
    __device__ __forceinline__ uint32_t window_lit(uint32_t at) {
        if (LFX_RING_PIPE) {                                                  // at / 32 - rb in {0, 1}
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r01), "+v"(r23));
            const uint32_t r0 = (uint32_t)r01, r1 = (uint32_t)(r01 >> 32), r2 = (uint32_t)r23;
            const bool o = (at >> 5) != rb;
            return __builtin_amdgcn_alignbit(o ? r2 : r1, o ? r1 : r0, at);  // (the shift is at mod 32)
        }
        const uint32_t *p = ring + ((at >> 5) & (RING_DW - 1));
        return __builtin_amdgcn_alignbit(p[1], p[0], at);
    }
    __device__ __forceinline__ uint32_t window_dist(uint32_t at) {
        if (LFX_RING_PIPE) {                                                  // at / 32 - rb in {0, 1, 2}
            const uint32_t r0 = (uint32_t)r01, r1 = (uint32_t)(r01 >> 32), r2 = (uint32_t)r23, r3 = (uint32_t)(r23 >> 32);
            const uint32_t o = (at >> 5) - rb;
            const uint32_t lo = o == 0 ? r0 : o == 1 ? r1 : r2, hi = o == 0 ? r1 : o == 1 ? r2 : r3;
            const uint32_t v = __builtin_amdgcn_alignbit(hi, lo, at);
            // the register window of the NEXT symbol (it starts at at .. at + 28): issued here, used an iteration later
            rb = at >> 5;
            const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)(ring + (rb & (RING_DW - 1)));
            asm volatile("ds_read2_b32 %0, %2 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3"
                         : "=&v"(r01), "=&v"(r23) : "v"(a) : "memory");
            return v;
        }
        const uint32_t *p = ring + ((at >> 5) & (RING_DW - 1));
        return __builtin_amdgcn_alignbit(p[1], p[0], at);
    }
    // reload in two halves, so that the caller's code stores go between them: the wait for the in-flight dwords then sits in
    // FRONT of the stores (vmcnt counts loads and stores alike, in order: behind them it waits for their acknowledgements too).
    // take(): only when !room() — every later read of the ring starts at rel / 32 >= filled - RING_AHEAD + 1, so the RING_FILL
    // oldest dwords are free
    __device__ __forceinline__ void take() {
        const uint32_t slot = filled & (RING_DW - 1);                   // a multiple of RING_FILL
        uint32_t *p = ring + slot;
#pragma unroll
        for (uint32_t k = 0; k < RING_FILL; ++k) p[k] = x[k];
        if (slot == 0) {
#pragma unroll
            for (uint32_t k = 0; k < RING_DUP; ++k) ring[RING_DW + k] = x[k];
        }
        filled += RING_FILL;
        __builtin_amdgcn_sched_barrier(0);       // keep the caller's stores and the loads of issue() below the writes above
    }
    __device__ __forceinline__ void issue() {
#pragma unroll
        for (uint32_t k = 0; k < RING_FILL; ++k) x[k] = ld(widx + k);
        widx += RING_FILL;
    }
};

constexpr uint32_t EMIT_STAGE = 8;             // code words staged per lane (two workgroups' staging must fit one CU)
constexpr uint32_t EMIT_STRIDE = EMIT_STAGE + 1;   // row stride in dwords (odd: conflict-free across lanes)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(4))) U32x4 { u32x4 v; };   // 16-byte store at any dword address

// One lane decodes symbols from bit `start` until a symbol would start at or after `limit`, or
// EndOfBlock.  EMIT: write code words.  Returns 0 ok / 1 EOB / 2 undecodable; `endpos` = bit reached.
// EMIT also tracks the earliest cut of the slice: (cut_code, cut_out) = first code / byte such that no later
// code OF THIS SLICE reads a byte produced before it.  A back-reference that reads below the current
// candidate kills it and every later one up to itself, so the candidate moves to just behind it.
// the symbol loop over the register FIFO (FastBits)
#ifndef LFX_DEC_PRIO
#define LFX_DEC_PRIO 1
#endif
#ifndef LFX_SCAN_CP_BITS
#define LFX_SCAN_CP_BITS 768      // head of a slice: what a corrected start decodes again (blk_scan_kernel)
#endif
// Wavefront priority that rotates with a trip count (round 5).  The wavefronts of a workgroup — and the workgroups of a CU —
// that do equal work from the same start were served oldest first: in the symbol kernels the first wavefront left its decode
// after 783 K cycles and waited 297 K at the barrier for the last one (LFX_DEBUG, K2 lines), a SIMD running with three, two,
// one wavefront towards the end.  The four wavefronts that share a SIMD (w, w+4, w+8, w+12 or 4s .. 4s+3, either way) hold
// four different priorities at any trip, each one every priority in turn.
__device__ __forceinline__ void rotate_prio(uint32_t trip) {
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    switch ((trip + w + (w >> 2)) & 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// Writes a lane's staged code words (stage[0, staged), LDS) to dst[0, ...) with ALIGNED 16-byte stores only (round 6: the stores cost
// the storing scan 0.16 of its 0.80 ms, HISTORY.md): the words in front of the first 16-byte boundary leave as dwords, whole groups
// of four as one aligned store each, and the (up to three) words behind the last boundary STAY staged (moved to the front of the
// row) until the next pause — at the call's last one (`more` false) they leave as dwords.  A 16-byte store at any dword address
// touched two 32-byte sectors more often than not.  → the number of words that stay staged.
__device__ __forceinline__ uint32_t flush_staged(uint32_t *stage, uint32_t staged, uint32_t *dst, bool fits, bool more) {
    const uint32_t a0 = (uint32_t)((uintptr_t)dst >> 2) & 3u;
    const uint32_t head = min(staged, (4u - a0) & 3u);              // dwords up to the first boundary (<= 3)
    const uint32_t body = (staged - head) & ~3u;                    // 0, 4, or 8 (8: head = tail = 0)
    const uint32_t tail = staged - head - body;                     // <= 3
    const uint32_t tt = head + body;                                // where the tail starts (<= 8)
    {
        const uint32_t s0 = stage[0], s1 = stage[1], s2 = stage[2];
        const uint32_t g0 = stage[head], g1 = stage[head + 1], g2 = stage[head + 2], g3 = stage[head + 3];   // (head + 3 <= 6)
        if (fits) {
            if (head > 0) dst[0] = s0;
            if (head > 1) dst[1] = s1;
            if (head > 2) dst[2] = s2;
            if (body >= 4) *(u32x4 *)(dst + head) = u32x4{g0, g1, g2, g3};
        }
    }
    if (__ballot(body == 8)) {
        const uint32_t h0 = stage[4], h1 = stage[5], h2 = stage[6], h3 = stage[7];
        if (fits && body == 8) *(u32x4 *)(dst + 4) = u32x4{h0, h1, h2, h3};
    }
    if (__ballot(tail != 0)) {
        const uint32_t e0 = stage[min(tt, 8u)], e1 = stage[min(tt + 1, 8u)], e2 = stage[min(tt + 2, 8u)];
        if (!more) {
            if (fits) {
                if (tail > 0) dst[tt] = e0;
                if (tail > 1) dst[tt + 1] = e1;
                if (tail > 2) dst[tt + 2] = e2;
            }
        } else {
            if (tail > 0) stage[0] = e0;
            if (tail > 1) stage[1] = e1;
            if (tail > 2) stage[2] = e2;
        }
    }
    return more ? tail : 0u;
}

template <bool EMIT>
__device__ __forceinline__ int lane_decode_fifo(const FastTabs &T, const uint8_t *in, uint64_t nbytes, uint64_t start,
                                           uint64_t limit, uint32_t &ncodes, uint64_t &nout, uint32_t *codes,
                                           int64_t &reach, uint64_t &endpos, uint32_t &cut_code, uint32_t &cut_out,
                                           uint32_t *stage = nullptr, uint32_t store_cap = 0xFFFFFFFFu, uint32_t *ovf = nullptr) {
    // EMIT: code words are staged in this lane's LDS row (EMIT_STAGE entries) and written out as runs of
    // 16-byte stores whenever the wavefront pauses to refill its bit FIFOs — a lane's 4-byte stores, each
    // to a cache line of its own, cost ~6x their size in HBM traffic (partial-line evictions).
    // store_cap (the storing scan, round 6): `codes` has room for so many code words; what does not fit is counted, not
    // stored, and *ovf is raised (the block then takes the two-pass path).
    FastBits b;
    uint32_t staged = 0;
    b.init(in, nbytes, start);
    const uint64_t span = limit > start ? limit - start : 0;
    const uint32_t lim = span > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)span;
    int ret = 0;
    uint32_t no = 0;   // bytes produced by this call (added to nout at the end)
    int32_t reach_rel = INT32_MAX;   // EMIT: smallest (bytes produced by this call) - distance over the matches
    uint32_t trip = 0;
    while (b.used < lim && ret == 0) {
        if (LFX_DEC_PRIO) rotate_prio(trip++);
        // each symbol takes at most two dwords from the FIFO.
        // ONE divergent region per symbol: the loop runs on the lanes that still take a symbol, and EndOfBlock or an
        // undecodable code end a lane by predication (zero-length skips, nothing staged, nothing counted) instead of by
        // breaks — every break cost the loop an exec-mask save / merge of its own, about 30 scalar instructions per
        // symbol on top of 95 vector ones.
        bool act = b.qn >= 2 && (!EMIT || staged < EMIT_STAGE);      // (b.used < lim holds here)
        while (act) {
            b.append();
            uint32_t e = T.lit[(uint32_t)b.buf & ((1u << LIT_BITS) - 1)];
            if (__builtin_expect((e & 15) == 0, 0))
                e = e == E_LONG ? long_lookup(T.lit_count, T.lit_sorted, T.lit_info, 286, b.buf) : 0;
            const bool bad1 = e == 0;
            const uint32_t kind = (e >> 4) & 3;
            const bool eob = kind == K_EOB;              // (an entry 0 has kind 0: never EndOfBlock)
            // Literals and matches share one straight-line path (a literal is a symbol without extra bits and
            // without a distance): nearly every wavefront iteration holds both kinds, and a divergent branch
            // costs exec-mask round trips and a register copy per live value.  A literal lane also tops up
            // its bit window and looks up a distance entry; both are harmless and ignored.
            const bool is_match = kind == K_LEN;
            const uint32_t w = e & 15, eb = (e >> 6) & 31;      // (EndOfBlock carries no extra bits; entry 0: w = eb = 0)
            const uint32_t val = (e >> 16) + (((uint32_t)(b.buf >> w)) & ((1u << eb) - 1));   // byte, or length
            b.skip(w + eb);
            b.append();
            uint32_t d = T.dist[(uint32_t)b.buf & ((1u << DIST_BITS) - 1)];
            if (__builtin_expect(is_match && (d & 15) == 0, 0))
                d = d == E_LONG ? long_lookup(T.dist_count, T.dist_sorted, T.dist_info, 30, b.buf) : 0;
            const bool bad2 = is_match && d == 0;
            const bool ok = !(bad1 || eob || bad2);
            const bool okm = ok && is_match;
            const uint32_t dw = d & 15, db = (d >> 6) & 31;
            const uint32_t distance = (d >> 16) + (((uint32_t)(b.buf >> dw)) & ((1u << db) - 1));
            b.skip(okm ? dw + db : 0u);
            ret = (bad1 || bad2) ? 2 : eob ? 1 : 0;
            if (EMIT) {
                stage[staged] = (val << 16) | (is_match ? distance : 0u);    // (a slot behind the last code is never flushed)
                staged += ok ? 1u : 0u;
                const int32_t rel = (int32_t)no - (int32_t)distance;          // first byte a match reads
                reach_rel = okm && rel < reach_rel ? rel : reach_rel;
                const bool cut = okm && rel < (int32_t)cut_out;
                cut_code = cut ? ncodes + 1 : cut_code;
                cut_out = cut ? no + val : cut_out;
            }
            ncodes += ok ? 1u : 0u;
            no += ok ? (is_match ? val : 1u) : 0u;
            act = ok && b.qn >= 2 && b.used < lim && (!EMIT || staged < EMIT_STAGE);
        }
        const bool refill = ret == 0 && b.used < lim && b.qn < 2;
        if (refill) b.take();
        if (EMIT) {
            const bool fits = ncodes <= store_cap;
            if (!fits && ovf) *ovf = 1u;
            staged = flush_staged(stage, staged, codes + (ncodes - staged), fits, ret == 0 && b.used < lim);
        }
        if (refill) b.issue();
    }
    if (EMIT && reach_rel != INT32_MAX) {
        const int64_t r = (int64_t)nout + (int64_t)reach_rel;
        reach = r < reach ? r : reach;
    }
    nout += no;
    endpos = start + b.used;
    return ret;
}

// the symbol loop over the LDS ring (RingBits)
template <bool EMIT>
__device__ __forceinline__ int lane_decode(const FastTabs &T, const uint8_t *in, uint64_t nbytes, uint64_t start,
                                           uint64_t limit, uint32_t &ncodes, uint64_t &nout, uint32_t *codes,
                                           int64_t &reach, uint64_t &endpos, uint32_t &cut_code, uint32_t &cut_out,
                                           uint32_t *ring, uint32_t *stage = nullptr, uint32_t store_cap = 0xFFFFFFFFu,
                                           uint32_t *ovf = nullptr) {
    // EMIT: code words are staged in this lane's LDS row (EMIT_STAGE entries) and written out as runs of
    // 16-byte stores whenever the wavefront pauses to refill its bit rings — a lane's 4-byte stores, each
    // to a cache line of its own, cost ~6x their size in HBM traffic (partial-line evictions).
    RingBits b;
    uint32_t staged = 0;
    b.init(in, nbytes, start, ring);
    const uint64_t span = limit > start ? limit - start : 0;
    const uint32_t lim = span > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)span;
    const uint32_t rlim = lim + b.off0;          // used < lim  <=>  rel < rlim
    int ret = 0;
    uint32_t no = 0;   // bytes produced by this call (added to nout at the end)
    int32_t reach_rel = INT32_MAX;   // EMIT: smallest (bytes produced by this call) - distance over the matches
    uint32_t trip = 0;
    while (b.rel < rlim && ret == 0) {
        if (LFX_DEC_PRIO) rotate_prio(trip++);
        // ONE divergent region per symbol: the loop runs on the lanes that still take a symbol, and EndOfBlock or an
        // undecodable code end a lane by predication (zero-length skips, nothing staged, nothing counted) instead of by
        // breaks — every break cost the loop an exec-mask save / merge of its own.
        bool act = b.room() && (!EMIT || staged < EMIT_STAGE);      // (b.rel < rlim holds here)
        while (act) {
            const uint32_t w1 = b.window_lit(b.rel);
            uint32_t e = T.lit[w1 & ((1u << LIT_BITS) - 1)];
            if (__builtin_expect((e & 15) == 0, 0))
                e = e == E_LONG ? long_lookup(T.lit_count, T.lit_sorted, T.lit_info, 286, (uint64_t)w1) : 0;
            const bool bad1 = e == 0;
            const uint32_t kind = (e >> 4) & 3;
            const bool eob = kind == K_EOB;              // (an entry 0 has kind 0: never EndOfBlock)
            // Literals and matches share one straight-line path (a literal is a symbol without extra bits and
            // without a distance): nearly every wavefront iteration holds both kinds, and a divergent branch
            // costs exec-mask round trips and a register copy per live value.  A literal lane also looks up a
            // distance entry; it is harmless and ignored.
            const bool is_match = kind == K_LEN;
            const uint32_t wl = e & 15, eb = (e >> 6) & 31;      // (EndOfBlock carries no extra bits; entry 0: wl = eb = 0)
            const uint32_t val = (e >> 16) + __builtin_amdgcn_ubfe(w1, wl, eb);   // byte, or length
            const uint32_t at2 = b.rel + wl + eb;
            const uint32_t w2 = b.window_dist(at2);
            uint32_t d = T.dist[w2 & ((1u << DIST_BITS) - 1)];
            if (__builtin_expect(is_match && (d & 15) == 0, 0))
                d = d == E_LONG ? long_lookup(T.dist_count, T.dist_sorted, T.dist_info, 30, (uint64_t)w2) : 0;
            const bool bad2 = is_match && d == 0;
            const bool ok = !(bad1 || eob || bad2);
            const bool okm = ok && is_match;
            const uint32_t dw = d & 15, db = (d >> 6) & 31;
            const uint32_t distance = (d >> 16) + __builtin_amdgcn_ubfe(w2, dw, db);
            b.rel = at2 + (okm ? dw + db : 0u);
            ret = (bad1 || bad2) ? 2 : eob ? 1 : 0;
            if (EMIT) {
                stage[staged] = (val << 16) | (is_match ? distance : 0u);    // (a slot behind the last code is never flushed)
                staged += ok ? 1u : 0u;
                const int32_t rel = (int32_t)no - (int32_t)distance;          // first byte a match reads
                reach_rel = okm && rel < reach_rel ? rel : reach_rel;
                const bool cut = okm && rel < (int32_t)cut_out;
                cut_code = cut ? ncodes + 1 : cut_code;
                cut_out = cut ? no + val : cut_out;
            }
            ncodes += ok ? 1u : 0u;
            no += ok ? (is_match ? val : 1u) : 0u;
            act = ok && b.room() && b.rel < rlim && (!EMIT || staged < EMIT_STAGE);
        }
        const bool refill = ret == 0 && b.rel < rlim && !b.room();
        if (refill) b.take();
        if (EMIT) {
            const bool fits = ncodes <= store_cap;       // (the storing scan: what does not fit its region is counted, not stored)
            if (!fits && ovf) *ovf = 1u;
            staged = flush_staged(stage, staged, codes + (ncodes - staged), fits, ret == 0 && b.rel < rlim);
        }
        if (refill) b.issue();
    }
    if (EMIT && reach_rel != INT32_MAX) {
        const int64_t r = (int64_t)nout + (int64_t)reach_rel;
        reach = r < reach ? r : reach;
    }
    nout += no;
    endpos = start + b.used();
    return ret;
}

// ------------------------------------------------------------------------------------------------
// inclusive prefix sum over the wavefront: four row shifts and two row broadcasts, no LDS traffic
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast:15 → rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast:31 → rows 2, 3
    return x;
}

// ------------------------------------------------------------------------------------------------
// header parse (lane 0, plain byte loads — ~300 short steps per block) + tables
struct HdrBits {
    LaneBits b;       // register bit buffer + prefetched dwords (no per-call memory latency)
    uint64_t nbits;
    bool bad;
    __device__ __forceinline__ void init(const uint8_t *g, uint64_t nbytes, uint64_t pos) {
        b.init(g, nbytes, pos);
        nbits = nbytes * 8;
        bad = false;
    }
    __device__ __forceinline__ uint32_t get(uint32_t w) {
        if (b.pos + w > nbits) { bad = true; return 0; }
        b.refill();
        const uint32_t r = (uint32_t)b.buf & ((1u << w) - 1);
        b.skip(w);
        return r;
    }
};

// The code-length code (at most 19 symbols, widths <= 7) kept entirely in registers: per-lane arrays
// with dynamic indexing would live in scratch memory (~1 us per access).
struct ClenCode {
    uint64_t cnt;      // 5 bits per width 1..7 at bit 5*w
    uint64_t sym_lo;   // sorted symbols 0..11, 5 bits each
    uint64_t sym_hi;   // sorted symbols 12..18
    __device__ __forceinline__ uint32_t count(uint32_t w) const { return (uint32_t)(cnt >> (5 * w)) & 31; }
    __device__ __forceinline__ uint32_t sorted(uint32_t i) const {
        return i < 12 ? (uint32_t)(sym_lo >> (5 * i)) & 31 : (uint32_t)(sym_hi >> (5 * (i - 12))) & 31;
    }
    // clw: 3-bit widths of symbols 0..18 packed at bit 3*s
    __device__ __forceinline__ void build(uint64_t clw) {
        cnt = 0; sym_lo = 0; sym_hi = 0;
        for (uint32_t s = 0; s < 19; ++s) { const uint32_t w = (uint32_t)(clw >> (3 * s)) & 7; if (w) cnt += 1ull << (5 * w); }
        uint32_t k = 0;
        for (uint32_t w = 1; w <= 7; ++w)
            for (uint32_t s = 0; s < 19; ++s)
                if (((uint32_t)(clw >> (3 * s)) & 7) == w) {
                    if (k < 12) sym_lo |= (uint64_t)s << (5 * k); else sym_hi |= (uint64_t)s << (5 * (k - 12));
                    k++;
                }
    }
    // canonical walk over the next (at most 7) bits; returns the symbol or 99, *used = bits consumed
    __device__ __forceinline__ uint32_t decode(uint32_t bits, uint32_t &used) const {
        uint32_t code = 0, first = 0, index = 0;
        for (uint32_t w = 1; w <= 7; ++w) {
            code |= (bits >> (w - 1)) & 1;
            const uint32_t c = count(w);
            if (code < first + c) { used = w; return sorted(index + (code - first)); }
            index += c;
            first = (first + c) << 1;
            code <<= 1;
        }
        used = 0;
        return 99;
    }
};

constexpr int SCAN_THREADS = 1024;   // lanes per block: 16 wavefronts, 4 per SIMD hide each other's latency

// parse the block header at job.start_bit and build T.  → btype in hdr[0], bfinal hdr[1],
// data start bit hdr64[0]; status in hdr[2] (0 ok, 1 undecodable header).
// The code-length sequence itself is a serial Huffman stream (one lane), everything around it is spread
// over the workgroup: a 128-entry lookup table for the code-length code, pre-zeroed widths (zero runs
// only advance the cursor) and the table build.
template <int NT = SCAN_THREADS>      // NT: threads of the workgroup (1024; 256 in the kernels' instances for small blocks)
__device__ void parse_header(const uint8_t *in, uint64_t nbytes, uint64_t start_bit, FastTabs &T,
                             uint8_t *lens, uint32_t *hdr, uint64_t *hdr64, uint32_t tid) {
    __shared__ uint8_t cl_tab[128];   // symbol | width << 5 ; 0xFF = no code
    for (uint32_t i = tid; i < 640 / 4; i += NT) ((uint32_t *)lens)[i] = 0;
    if (tid == 0) {
        HdrBits hb;
        hb.init(in, nbytes, start_bit);
        uint32_t bad = 0;
        const uint32_t bfinal = hb.get(1), btype = hb.get(2);
        hdr[0] = btype; hdr[1] = bfinal;
        hdr64[1] = 0;
        if (hb.bad || btype == 3) bad = 1;
        else if (btype == 2) {
            const uint32_t nl = hb.get(5) + 257, nd = hb.get(5) + 1, nc = hb.get(4) + 4;
            hdr[3] = nl; hdr[4] = nd;
            uint64_t clw = 0;
            for (uint32_t k = 0; k < nc; ++k) clw |= (uint64_t)hb.get(3) << (3 * clen_order(k));
            hdr64[1] = clw;
            if (nd > 30 || hb.bad) bad = 1;
        }
        hdr[2] = bad;
        hdr64[0] = hb.b.pos;
    }
    __syncthreads();
    const uint32_t btype = hdr[0];
    if (hdr[2] || btype == 0) return;
    if (btype == 1) {
        for (uint32_t s = tid; s < 288 + 30; s += NT)
            lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5;
        if (tid == 0) { hdr[3] = 288; hdr[4] = 30; }
    } else {
        if (tid < 128) {
            ClenCode cc;
            cc.build(hdr64[1]);
            uint32_t used = 0;
            const uint32_t sym = cc.decode(tid, used);
            cl_tab[tid] = sym == 99 ? 0xFF : (uint8_t)(sym | used << 5);
        }
        // Round 5: the code-length sequence (HLIT + 257 + HDIST + 1 widths, up to 320 symbols of 1 .. 14 bits) is decoded by
        // wavefront 0 a WINDOW of 64 bit offsets at a time: lane i decodes the symbol that would start at offset i, a scalar walk
        // (one v_readlane per symbol) marks the offsets the true chain visits, and everything else — output positions (prefix sum
        // of the repeat counts), "previous width" for symbol 16 (the last marked lane in front that is not a 16), the
        // checks, the stores — is data-parallel over the marked lanes: about eleven symbols per window.  Rounds 1-4 walked the
        // sequence on ONE lane, a dependent LDS lookup and a 64-bit window update per symbol: 57 us per block, which is most of a
        // small block's scan (cfg3: 4096 streams of 64 KiB) and all of a small stream's.
        // (a header is at most 4554 bits long: far from the end of the stream no step needs a bounds check — the window form;
        //  near the end: lane 0, every step checked)
        __shared__ uint32_t hbits[160];
        const uint64_t hpos = hdr64[0];
        const uint64_t a = (uint64_t)in;
        const uint64_t habs = hpos + (a & 3) * 8;
        const bool lean = hpos + 6000 <= nbytes * 8;
        if (lean && tid < 160) hbits[tid] = ((gptr_u32)(a & ~3ull))[(habs >> 5) + tid];      // (6000 bits = 187 dwords are there)
        __syncthreads();
        if (lean && tid < 64) {
            const uint32_t lane = tid;
            const uint32_t total = hdr[3] + hdr[4];
            uint32_t rel = (uint32_t)habs & 31, have = 0, last = 0, bad = 0;
            bool done = false;
            while (!done && !bad) {
                const uint32_t b = rel + lane;
                const uint32_t w = __builtin_amdgcn_alignbit(hbits[(b >> 5) + 1], hbits[b >> 5], b & 31);
                const uint32_t e = cl_tab[w & 127];
                const uint32_t sym = e & 31, used = e >> 5;
                const uint32_t k4 = (sym - 16u < 3u ? sym - 15 : 0) * 4;   // repeat codes 16 / 17 / 18
                const uint32_t nbx = (0x7320u >> k4) & 15, basex = (0xB331u >> k4) & 15;
                const uint32_t rep = basex + __builtin_amdgcn_ubfe(w, used, nbx);
                const uint32_t nxt = lane + used + nbx;
                // the offsets of this window the chain visits
                uint64_t M = 0;
                uint32_t i = 0;
                while (i < 64) {
                    M |= 1ull << i;
                    i = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)i);
                }
                const bool on = (M >> lane) & 1ull;
                const uint32_t repm = on ? rep : 0u;
                const uint32_t incl = wave_inclusive_sum(repm);
                const uint32_t at = have + incl - repm;                    // widths in front of this symbol
                const bool real = on && at < total;                        // (a marked lane behind the last width is data already)
                const uint64_t R = __ballot(real);
                if (__ballot(real && (e == 0xFF || (sym == 16 && at == 0) || at + rep > total))) { bad = 1; break; }
                // symbol 16 repeats the previous width: that of the last real lane in front that is not a 16 (17 / 18 leave 0)
                const uint64_t D = __ballot(real && sym != 16);
                const uint64_t Dlt = D & ((1ull << lane) - 1ull);
                const uint32_t own = sym < 16 ? sym : 0u;
                const uint32_t from = Dlt ? 63u - (uint32_t)__builtin_clzll(Dlt) : 0u;
                const uint32_t prev = (uint32_t)__shfl((int)own, (int)from);
                const uint32_t val = sym == 16 ? (Dlt ? prev : last) : own;
                if (real && val)
                    for (uint32_t k = 0; k < rep; ++k) lens[at + k] = (uint8_t)val;      // (rep <= 6 here: zeros are not written)
                if (R) {
                    const uint32_t lr = 63u - (uint32_t)__builtin_clzll(R);            // the window's last real symbol
                    have = (uint32_t)__builtin_amdgcn_readlane((int)(at + rep), (int)lr);
                    last = (uint32_t)__builtin_amdgcn_readlane((int)val, (int)lr);
                }
                // the first marked lane that is not real any more is where the header ends; else the chain's exit from the window
                const uint64_t E = M & ~R;
                if (have >= total) {
                    done = true;
                    rel += E ? (uint32_t)__builtin_ctzll(E) : i;
                } else rel += i;
            }
            if (lane == 0) {
                hdr[2] = bad;
                hdr64[0] = ((habs >> 5) << 5) + rel - (a & 3) * 8;
            }
        }
        if (!lean && tid == 0) {
            HdrBits hb;
            hb.init(in, nbytes, hdr64[0]);
            const uint32_t nl = hdr[3], nd = hdr[4], total = nl + nd;
            uint32_t have = 0, last = 0, bad = 0;
            while (!bad && have < total) {
                if (hb.b.pos >= hb.nbits) { bad = 1; break; }
                hb.b.refill();
                const uint32_t e = cl_tab[(uint32_t)hb.b.buf & 127];
                const uint32_t sym = e & 31, used = e >> 5;
                if (e == 0xFF || hb.b.pos + used > hb.nbits) { bad = 1; break; }
                hb.b.skip(used);
                uint32_t rep = 1, val = sym;
                if (sym == 16) { if (have == 0) { bad = 1; break; } rep = 3 + hb.get(2); val = last; }
                else if (sym == 17) { rep = 3 + hb.get(3); val = 0; }
                else if (sym == 18) { rep = 11 + hb.get(7); val = 0; }
                if (have + rep > total || hb.bad) { bad = 1; break; }
                if (val) for (uint32_t k = 0; k < rep; ++k) lens[have + k] = (uint8_t)val;
                have += rep;
                last = val;
            }
            hdr[2] = bad;
            hdr64[0] = hb.b.pos;
        }
    }
    __syncthreads();
    if (hdr[2]) return;
    const uint32_t nl = hdr[3], nd = hdr[4];
    if (!build_fast(T, lens, nl, lens + nl, nd, tid, NT)) {
        if (tid == 0) hdr[2] = 1;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// K1: speculative scan of one candidate block: validated per-lane starts, code and byte counts
// (two workgroups per CU: a stream has a few more candidate blocks than the GPU has CUs, and a second
// round of workgroups would double the kernel's time — 8 waves per SIMD = at most 64 VGPRs)
// STORE (round 6, the single-pass decode of a stream's own large blocks): the scan KEEPS what it decodes — every lane writes
// its slice's code words into a region of its own in `temp` (BlkJob::temp_off, `cap` code words per lane) and tracks what
// blk_emit_kernel tracks (how far back its matches reach, its earliest legal cut), so that the second Huffman pass of the
// block is replaced by blk_place_kernel: a copy of the code words to their final places.  A slice is decoded in two
// segments — the head [start, checkpoint), which a corrected start decodes again, and the rest, which is reused — so the
// lane's region holds the head's codes from 0 on and the rest's from SCAN_HEADCAP on (or right behind the head's when the
// rest was decoded behind a head that missed the checkpoint); BlkLanesX says which.
constexpr uint32_t SCAN_HEADCAP = 448;        // code words a head may take: (768 + 48) bits at two bits a code, rounded up to a multiple of 4
// NT (round 6): threads of the workgroup = slices of the block.  1024 for a stream's own blocks; 256 for SMALL blocks (the batch
// path's 64 KiB streams, another encoder's 30 KB blocks): a 32 KB block cut into 1024 slices is 17 symbols a lane — a speculative
// start is not in step before its slice ends, and five rounds of head decodes follow — and two workgroups of sixteen wavefronts
// per CU spend most of their time in the header's serial parts; with 256 slices a lane has 68 symbols and five blocks share a CU.
template <bool STORE, int NT = SCAN_THREADS>
__global__ __launch_bounds__(NT, STORE ? 4 : 8) void blk_scan_kernel(const uint8_t *__restrict__ in, uint64_t nbytes,
                                                                const BlkJob *__restrict__ jobs,
                                                                BlkInfo *__restrict__ infos,
                                                                BlkLanes *__restrict__ lanes,
                                                                FastTabs *__restrict__ tabs,
                                                                uint32_t *__restrict__ temp, BlkLanesX *__restrict__ lanesx) {
    extern __shared__ uint32_t scan_stage[];   // STORE: NT rows of EMIT_STRIDE dwords
    __shared__ FastTabs T;
    // STORE: the lanes' bits through LDS rings, as in blk_emit_kernel<true> — a wavefront that stores its codes AND loads its
    // bits through the register FIFO waits for its own stores at every refill (loads and stores share vmcnt): measured,
    // 0.945 ms for the storing scan on the FIFO against 0.57 for the counting one
    __shared__ uint32_t s_ring[STORE ? NT * RING_STRIDE : 1];
    __shared__ __attribute__((aligned(4))) uint8_t lens[640];
    __shared__ uint32_t hdr[8];
    __shared__ uint64_t hdr64[2];
    __shared__ uint64_t s_start[NT + 1];
    __shared__ uint32_t s_flag[NT];
    __shared__ uint64_t s_exit[NT];
    __shared__ uint32_t s_scan[NT / 64];
    __shared__ uint64_t s_scan64[NT / 64];
    const uint32_t tid = threadIdx.x;
    const BlkJob job = jobs[blockIdx.x];
    BlkInfo bi;
    bi.status = BLK_OK; bi.btype = 0; bi.bfinal = 0; bi.nlanes = 0; bi.end_bit = 0; bi.n_codes = 0; bi.n_out = 0;
    bi.data_bit = 0; bi.rounds = 0; bi._pad = 0; bi.cyc_hdr = 0; bi.cyc_total = 0;
    const uint64_t t_begin = clock64();
    parse_header<NT>(in, nbytes, job.start_bit, T, lens, hdr, hdr64, tid);
    const uint64_t t_hdr = clock64();
    bi.btype = hdr[0]; bi.bfinal = hdr[1]; bi.data_bit = hdr64[0];
    if (hdr[2]) { bi.status = BLK_BAD; if (tid == 0) infos[blockIdx.x] = bi; return; }
    if (hdr[0] == 0) {
        // stored block (decode.rs:81-111): LEN / NLEN after byte alignment
        uint64_t byte = (hdr64[0] + 7) >> 3;
        if (byte + 4 > nbytes) bi.status = BLK_BAD;
        else {
            const uint32_t len = in[byte] | in[byte + 1] << 8, nlen = in[byte + 2] | in[byte + 3] << 8;
            if (((~len) & 0xFFFF) != nlen || byte + 4 + len > nbytes) bi.status = BLK_BAD;
            else { bi.n_out = len; bi.data_bit = (byte + 4) * 8; bi.end_bit = (byte + 4 + len) * 8; }
        }
        if (tid == 0) infos[blockIdx.x] = bi;
        return;
    }
    if (tabs) {   // the emit kernel takes the block's tables from here instead of parsing the header again
        uint32_t *dst = (uint32_t *)&tabs[blockIdx.x];
        const uint32_t *srcw = (const uint32_t *)&T;
        for (uint32_t i = tid; i < sizeof(FastTabs) / 4; i += NT) dst[i] = srcw[i];
    }
    const bool piece = job.piece != 0;
    if (piece && job.warm_bit) {
        // warm-up: the first symbol boundary at or behind lo_bit (one lane; ~500 symbols)
        if (tid == 0) {
            uint32_t wn = 0, wcc = 0, wco = 0;
            uint64_t wo = 0, at = 0;
            int64_t wr = 0;
            const int r = lane_decode_fifo<false>(T, in, nbytes, job.warm_bit, job.lo_bit, wn, wo, nullptr, wr, at, wcc, wco);
            hdr64[0] = at;
            hdr[2] = r != 0;          // EndOfBlock or an undecodable code before the piece: the block ended earlier
        }
        __syncthreads();
        bi.data_bit = hdr64[0];
        if (hdr[2]) { bi.status = BLK_BAD; if (tid == 0) infos[blockIdx.x] = bi; return; }
    }
    const uint64_t d0 = hdr64[0];
    uint64_t e = job.end_bit;
    if (e > nbytes * 8) e = nbytes * 8;
    if (e < d0 + 1) e = d0 + 1;
    // slices of >= 128 bits so that one symbol (<= 48 bits) never skips a whole slice
    uint64_t slice = (e - d0 + NT - 1) / NT;
    if (slice < 128) slice = 128;
    const uint32_t nl = (uint32_t)((e - d0 + slice - 1) / slice);  // lanes in use (<= NT)
    const uint64_t my_bound = d0 + (uint64_t)(tid + 1) * slice;     // end of my slice
    s_start[tid] = tid == 0 ? d0 : (tid < nl ? d0 + (uint64_t)tid * slice : ~0ull);
    if (tid == 0) { s_start[NT] = ~0ull; hdr[7] = 0; }      // (hdr[7]: STORE's "a lane's codes did not fit" flag)
    __syncthreads();
    uint32_t rounds = 0;
    // Round 1 decodes every slice from its guessed start: a short head [start, checkpoint) and the rest
    // [checkpoint, bound).  A corrected start (round 2+) only re-decodes the head: when it lands exactly on
    // the checkpoint — a speculative decode is in step after a few symbols — the rest is bit for bit the
    // decode already done, so its counts are reused.
    constexpr uint64_t CP_BITS = LFX_SCAN_CP_BITS;
    uint32_t nc = 0, flag = 4;
    uint64_t no = 0, exitpos = ~0ull;
    uint64_t decoded_from = ~0ull, cp_pos = 0, rest_exit = 0;
    uint32_t rest_flag = 0;
    bool have_cp = false;
    // STORE: the lane's region of `temp`, and per segment (A = head, B = rest) what the emit kernel's decode tracks: the
    // smallest (bytes of the segment produced so far - distance) over its matches and its earliest legal cut, both relative
    // to the segment's own start
    uint32_t *my_temp = nullptr;
    uint32_t lane_cap = 0, ovf = 0;
    if constexpr (STORE) {
        lane_cap = job.cap;
        my_temp = temp + job.temp_off + (uint64_t)tid * lane_cap;
    }
    struct Seg { uint32_t n; uint64_t no; int64_t reach; uint32_t cc, co; };
    Seg A{0, 0, INT64_MAX, 0, 0}, B{0, 0, INT64_MAX, 0, 0};
    bool b_split = false;                      // B's codes lie at SCAN_HEADCAP (else right behind A's)
    // one segment [from, to): counts (and, STORE, codes at dst with room for `room` of them + tracking) → decoder's return
    auto seg = [&](uint64_t from, uint64_t to, Seg &g, uint64_t &at, uint32_t *dst, uint32_t room) -> int {
        g = Seg{0, 0, INT64_MAX, 0, 0};
        if constexpr (STORE)
            return lane_decode<true>(T, in, nbytes, from, to, g.n, g.no, dst, g.reach, at, g.cc, g.co, s_ring + tid * RING_STRIDE,
                                     scan_stage + tid * EMIT_STRIDE, room, &ovf);
        else {
            int64_t dummy = 0;
            return lane_decode_fifo<false>(T, in, nbytes, from, to, g.n, g.no, nullptr, dummy, at, g.cc, g.co);
        }
    };
    for (;;) {
        const uint64_t st = s_start[tid];
        if (tid < nl && st != ~0ull) {
            if (st != decoded_from) {
                // the last lane keeps going to the end of the stream range (the block may end exactly at e)
                // (a piece's last lane stops at the first symbol boundary >= e: the next piece starts exactly there)
                const uint64_t lim = tid + 1 == nl ? (piece ? e : e + 64) : my_bound;
                // Two segments at most — the head A, then the rest B — through ONE call site of the symbol loop (three
                // inlined copies of it were 4 400 instructions of kernel): pass 0 decodes A; what it finds decides whether
                // pass 1 decodes a B, and which kind.
                const bool retry = have_cp && st < cp_pos;           // a corrected start in front of a cached rest
                uint64_t at = st, from = st, to = retry ? cp_pos : (st + CP_BITS < lim ? st + CP_BITS : lim);
                uint32_t *dst = my_temp;
                uint32_t room = lane_cap < SCAN_HEADCAP ? lane_cap : SCAN_HEADCAP;    // (a cached rest lies at SCAN_HEADCAP)
                int r = 0;
                bool reuse = false, cache_b = false;
                if (!retry) have_cp = false;
                for (int pass = 0; pass < 2; ++pass) {
                    Seg G;
                    uint64_t ex = from;
                    const int rr = seg(from, to, G, ex, dst, room);
                    if (pass == 0) {
                        A = G; r = rr; at = ex;
                        if (retry) {
                            reuse = r == 0 && at == cp_pos;             // the cached rest stands
                            if (reuse) break;
                            have_cp = false;
                        }
                        B = Seg{0, 0, INT64_MAX, 0, 0};
                        b_split = false;
                        if (!(r == 0 && at < lim)) break;               // the head was all of the slice (or ended it)
                        // a B behind a fresh head is cached at SCAN_HEADCAP (a later corrected start reuses it when it
                        // lands on the checkpoint); behind a head that missed the checkpoint it follows A's codes
                        cache_b = !retry;
                        from = at; to = lim;
                        dst = my_temp + (cache_b ? SCAN_HEADCAP : A.n);
                        room = cache_b ? (lane_cap > SCAN_HEADCAP ? lane_cap - SCAN_HEADCAP : 0u) : (lane_cap > A.n ? lane_cap - A.n : 0u);
                    } else {
                        B = G;
                        if (cache_b) {
                            have_cp = true; cp_pos = at; b_split = true; reuse = true;
                            rest_exit = ex;
                            rest_flag = rr == 1 ? 1 : rr == 2 ? 2 : 0;
                        } else { r = rr; at = ex; }
                    }
                }
                if (reuse) { exitpos = rest_exit; flag = rest_flag; }
                else { exitpos = at; flag = r == 1 ? 1 : r == 2 ? 2 : 0; }
                nc = A.n + B.n;
                no = A.no + B.no;
                decoded_from = st;
            }
        } else flag = 4;
        s_flag[tid] = flag; s_exit[tid] = exitpos;      // (the counts stay in registers: only their owner reads them)
        __syncthreads();
        // lane k+1 must start where lane k stopped (only while no EOB / failure)
        bool changed = false;
        // A lane that stopped on EndOfBlock / an undecodable code may simply have been mis-aligned:
        // it must not disturb its successor's (speculative) start.  Validity is settled after
        // convergence: lanes before the first flagged lane are chained from the exact lane 0.
        if (tid + 1 < nl && flag == 0 && s_start[tid + 1] != exitpos) { s_start[tid + 1] = exitpos; changed = true; }
        rounds++;
        const int any = __syncthreads_or(changed ? 1 : 0);
        if (!any || rounds >= 64) break;
    }
    // the EOB lane: first lane whose (validated) decode hit EndOfBlock; every lane before it must be clean
    __shared__ uint32_t s_eob;
    if (tid == 0) s_eob = 0xFFFFFFFFu;
    __syncthreads();
    if (tid < nl && s_flag[tid] != 0) atomicMin(&s_eob, tid);   // first flagged lane of the chain
    __syncthreads();
    uint32_t eobl = s_eob;
    bi.rounds = rounds;
    if (rounds >= 64) { bi.status = BLK_BAD; if (tid == 0) infos[blockIdx.x] = bi; return; }
    if (eobl == 0xFFFFFFFFu) {
        // no EndOfBlock inside the range.  An ordinary job reports just that (its range was cut short by a false
        // candidate); a piece is "open": all its lanes count and its end is the exit of the last lane
        bi.status = BLK_NO_EOB;
        if (!piece || nl == 0) { if (tid == 0) infos[blockIdx.x] = bi; return; }
        eobl = nl - 1;
    } else if (s_flag[eobl] != 1) { bi.status = BLK_BAD; if (tid == 0) infos[blockIdx.x] = bi; return; }
    // exclusive scans of code / byte counts over lanes <= eobl
    const uint32_t mync = tid <= eobl ? nc : 0;
    const uint64_t myno = tid <= eobl ? no : 0;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    uint32_t x = mync;
    uint64_t y = myno;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t a = __shfl_up(x, o);
        const uint64_t c = __shfl_up(y, o);
        if ((int)lane >= o) { x += a; y += c; }
    }
    if (lane == 63) { s_scan[wave] = x; s_scan64[wave] = y; }
    __syncthreads();
    uint32_t px = 0;
    uint64_t py = 0;
    for (uint32_t w = 0; w < wave; ++w) { px += s_scan[w]; py += s_scan64[w]; }
    BlkLanes *L = &lanes[blockIdx.x];
    L->start[tid] = tid <= eobl ? s_start[tid] : ~0ull;
    L->code_off[tid] = px + x - mync;
    L->out_off[tid] = py + y - myno;
    if constexpr (STORE) {
        // the slice as ONE record: where its codes lie, how far back it reads, its earliest legal cut (slice-relative).
        // A's cut candidate stands if nothing in B reads in front of it; otherwise B's own (B's codes all come later).
        BlkLanesX *X = &lanesx[blockIdx.x];
        const bool live = tid <= eobl;
        const int64_t rb = B.reach == INT64_MAX ? INT64_MAX : (int64_t)A.no + B.reach;
        const int64_t reach = A.reach < rb ? A.reach : rb;
        const bool a_stands = rb >= (int64_t)A.co;
        uint32_t cc = a_stands ? A.cc : A.n + B.cc, co = a_stands ? A.co : (uint32_t)A.no + B.co;
        if (cc >= A.n + B.n) cc = 0xFFFFFFFFu;          // (a cut behind the last code belongs to the next lane)
        X->n_head[tid] = live ? A.n : 0u;
        X->n_rest[tid] = live ? B.n : 0u;
        X->rest_at[tid] = b_split ? SCAN_HEADCAP : A.n;
        X->reach[tid] = live && reach != INT64_MAX ? (reach < INT32_MIN ? INT32_MIN : (int32_t)reach) : INT32_MAX;
        X->cut_code[tid] = live ? cc : 0xFFFFFFFFu;
        X->cut_out[tid] = co;
        if (live && ovf) atomicOr(&hdr[7], 1u);
    }
    if (tid == NT - 1) {
        bi.n_codes = px + x;
        bi.n_out = py + y;
    }
    // broadcast totals through LDS
    __shared__ uint32_t s_tot;
    __shared__ uint64_t s_tot64;
    if (tid == NT - 1) { s_tot = px + x; s_tot64 = py + y; }
    __syncthreads();
    if (tid == 0) {
        if constexpr (STORE) bi._pad = hdr[7] & 1u;      // 1: some lane's codes did not fit its region (the block takes the two-pass path)
        bi.n_codes = s_tot; bi.n_out = s_tot64; bi.end_bit = s_exit[eobl]; bi.nlanes = eobl + 1; bi.rounds = rounds;
        bi.cyc_hdr = (uint32_t)(t_hdr - t_begin); bi.cyc_total = (uint32_t)(clock64() - t_begin);
        infos[blockIdx.x] = bi;
    }
}

// ------------------------------------------------------------------------------------------------
// K2: validated lanes decode their slices into the code array; the block is then cut into units
// that no back-reference crosses (reference-made blocks fall apart at every LZ77 chunk boundary,
// default.rs:73) so that K3 can materialise them concurrently.
constexpr uint32_t MAX_UNITS = 8;

// The part of K2 behind the decode, shared by blk_emit_kernel and blk_place_kernel: from every lane's `reach` (smallest byte of the
// block a match of its slice reads: absolute, may be negative) and cut candidate → the block's units.
template <int NT = SCAN_THREADS>
__device__ __forceinline__ void blk_units_tail(const uint32_t tid, const BlkEmit &job, const BlkLanes *L, const int64_t reach,
                                               uint32_t cut_code, const uint64_t cut_pos, BlkUnits *U, const uint32_t unit_target,
                                               const uint32_t free_shift, const uint64_t t_begin, const uint64_t t_hdr) {
    const uint64_t t_dec = clock64();
    // suffix minimum of `reach` over LATER lanes: within the wavefront by shuffles, then across wavefronts
    __shared__ int64_t s_wmin[NT / 64];
    const uint32_t lane = tid & 63, wave = tid >> 6;
    int64_t sfx = reach;   // inclusive suffix min
    for (int o = 1; o < 64; o <<= 1) {
        const int64_t y = __shfl_down(sfx, o);
        if (lane + o < 64 && y < sfx) sfx = y;
    }
    if (lane == 0) s_wmin[wave] = sfx;
    __syncthreads();
    int64_t later = __shfl_down(sfx, 1);            // min over the later lanes of my wavefront
    if (lane == 63) later = INT64_MAX;
    for (uint32_t w = wave + 1; w < NT / 64; ++w) { const int64_t y = s_wmin[w]; if (y < later) later = y; }
    // the lane's cut is legal iff no later lane reads a byte in front of it either (a later candidate of
    // this lane lies even further right, so it cannot be legal when this one is not)
    if (cut_code != 0xFFFFFFFFu && later < (int64_t)cut_pos) cut_code = 0xFFFFFFFFu;
    __shared__ uint32_t s_cut_code[NT];
    __shared__ uint64_t s_cut_pos[NT];
    s_cut_code[tid] = cut_code;
    s_cut_pos[tid] = cut_pos;
    __shared__ uint64_t s_legal[NT / 64];
    const uint64_t legal = __ballot(cut_code != 0xFFFFFFFFu);
    if (lane == 0) s_legal[wave] = legal;
    __syncthreads();
    const uint64_t t_cut = clock64();
    if (tid == 0) {
        // The block is cut into about n_codes / unit_target units (the host sizes unit_target so that all
        // units of the stream are resident in K3 at once): for every ideal boundary take the nearest legal
        // cut — legal cuts are few (reference-made blocks: the LZ77 chunk boundaries), lanes hold about the
        // same number of codes, so the search starts at the lane the boundary falls into.
        uint32_t nu = 0, last = 0;
        uint32_t want_units = (job.n_codes + unit_target / 2) / unit_target;
        if (want_units < 1) want_units = 1;
        if (want_units > MAX_UNITS) want_units = MAX_UNITS;
        const uint32_t tol = job.n_codes / (2 * want_units);
        U->code0[0] = 0; U->out0[0] = 0;
        for (uint32_t b = 1; b < want_units; ++b) {
            const uint32_t ideal = (uint32_t)((uint64_t)job.n_codes * b / want_units);
            uint32_t l0 = (uint32_t)((uint64_t)job.nlanes * b / want_units);
            if (l0 >= NT) l0 = NT - 1;
            // nearest legal lane at or above l0, and below l0
            int up = -1, dn = -1;
            for (uint32_t w = l0 >> 6; w < NT / 64 && up < 0; ++w) {
                uint64_t m = s_legal[w];
                if (w == (l0 >> 6)) m &= ~0ull << (l0 & 63);
                if (m) up = (int)(w * 64 + (uint32_t)__builtin_ctzll(m));
            }
            for (int w = (int)(l0 >> 6); w >= 0 && dn < 0; --w) {
                uint64_t m = s_legal[w];
                if (w == (int)(l0 >> 6)) m &= (l0 & 63) ? ~0ull >> (64 - (l0 & 63)) : 0ull;
                if (m) dn = w * 64 + 63 - __builtin_clzll(m);
            }
            int pick = -1;
            uint32_t best = 0xFFFFFFFFu;
            if (up >= 0) { const uint32_t cc = s_cut_code[up]; const uint32_t d = cc > ideal ? cc - ideal : ideal - cc; if (d < best) { best = d; pick = up; } }
            if (dn >= 0) { const uint32_t cc = s_cut_code[dn]; const uint32_t d = cc > ideal ? cc - ideal : ideal - cc; if (d < best) { best = d; pick = dn; } }
            if (pick < 0 || best > tol) continue;
            const uint32_t cc = s_cut_code[pick];
            if (cc == 0 || cc <= last || cc >= job.n_codes) continue;
            nu++;
            U->code0[nu] = cc; U->out0[nu] = s_cut_pos[pick];
            last = cc;
        }
        nu++;
        U->code0[nu] = job.n_codes; U->out0[nu] = job.n_out;
        U->n = nu;
        // ... and cut at slice boundaries without regard to back-references (marker-based materialisation)
        // (units of 2^free_shift bytes, 32 KiB and up: every unit costs the window resolution 32 Ki lookups; the host
        // sizes them so that the stream still has a few units per resident slot)
        uint32_t fn = 0;
        uint32_t fwant = (uint32_t)(job.n_out >> free_shift);
        fwant = fwant < 1 ? 1 : fwant > MAX_FREE_UNITS ? MAX_FREE_UNITS : fwant;
        U->fcode0[0] = 0; U->fout0[0] = 0;
        for (uint32_t b = 1; b < fwant; ++b) {
            const uint32_t l = (uint32_t)((uint64_t)job.nlanes * b / fwant);
            if (l == 0 || l >= job.nlanes) continue;
            const uint32_t cc = L->code_off[l];
            if (cc <= U->fcode0[fn] || cc >= job.n_codes) continue;
            fn++;
            U->fcode0[fn] = cc; U->fout0[fn] = L->out_off[l];
        }
        fn++;
        U->fcode0[fn] = job.n_codes; U->fout0[fn] = job.n_out;
        U->fn = fn;
        U->cyc[0] = (uint32_t)(t_hdr - t_begin); U->cyc[1] = (uint32_t)(t_dec - t_hdr);
        U->cyc[2] = (uint32_t)(t_cut - t_dec); U->cyc[3] = (uint32_t)(clock64() - t_cut);
    }
}

// A block's verdict on its back-references → the call's flags (bit 0: a reference in front of the member's first byte, bit 1: in
// front of the block — it needs the earlier output): ONE device-scope atomic per block, and none once the bit stands.  (One per
// wavefront with a reference that leaves its block — another encoder's stream of thousands of small blocks — was tens of
// thousands of atomics on one address, ~12 ns each one after the other: tools/exp/atomic_lat.hip.)  Every thread calls it.
__device__ __forceinline__ void blk_flag_reach(uint32_t tid, int64_t reach, uint64_t hist, uint32_t *__restrict__ flags,
                                               uint32_t *__restrict__ job_flags) {
    __shared__ uint32_t s_fl;
    if (tid == 0) s_fl = 0;
    __syncthreads();
    if (reach < -(int64_t)hist) atomicOr(&s_fl, 1u);
    else if (reach < 0) atomicOr(&s_fl, 2u);
    __syncthreads();
    if (tid == 0 && s_fl) {
        const uint32_t f = s_fl;
        if ((f & 1u) && job_flags) job_flags[blockIdx.x] = 1u;
        if ((__atomic_load_n(&flags[0], __ATOMIC_RELAXED) & f) != f) atomicOr(&flags[0], f);
    }
}

// RING: the lanes' bits come through LDS rings (RingBits: one workgroup per CU — the single-stream path, whose blocks are
// large); else through the register FIFO (FastBits: two workgroups per CU — the batch path's thousands of small blocks)
template <bool RING, int NT = SCAN_THREADS>
__global__ __launch_bounds__(NT, RING ? 4 : 8) void blk_emit_kernel(const uint8_t *__restrict__ in, uint64_t nbytes,
                                                                const BlkEmit *__restrict__ jobs,
                                                                const BlkLanes *__restrict__ lanes,
                                                                uint32_t *__restrict__ codes,
                                                                uint32_t *__restrict__ flags,
                                                                BlkUnits *__restrict__ units, uint32_t unit_target, uint32_t free_shift,
                                                                uint32_t *__restrict__ job_flags,
                                                                const FastTabs *__restrict__ tabs) {
    __shared__ FastTabs T;
    __shared__ __attribute__((aligned(4))) uint8_t lens[640];
    __shared__ uint32_t hdr[8];
    __shared__ uint64_t hdr64[2];
    extern __shared__ uint32_t emit_stage[];   // NT rows of EMIT_STRIDE dwords
    __shared__ uint32_t s_ring[RING ? NT * RING_STRIDE : 1];      // the lanes' bit rings (lane_decode)
    const uint32_t tid = threadIdx.x;
    const BlkEmit job = jobs[blockIdx.x];
    BlkUnits *U = &units[blockIdx.x];
    if (job.placed) return;                  // (its codes were stored by the scan: blk_place_kernel's block)
    if (job.btype == 0) {
        if (tid == 0) {
            U->n = 1; U->code0[0] = 0; U->code0[1] = 0; U->out0[0] = 0; U->out0[1] = job.n_out;
            U->fn = 1; U->fcode0[0] = 0; U->fcode0[1] = 0; U->fout0[0] = 0; U->fout0[1] = job.n_out;
        }
        return;
    }
    const uint64_t t_begin = clock64();
    if (tabs) {   // the tables the scan kernel built for this block
        const uint32_t *srcw = (const uint32_t *)&tabs[job.cand];
        uint32_t *dst = (uint32_t *)&T;
        constexpr uint32_t TW = sizeof(FastTabs) / 4, TPER = (TW + NT - 1) / NT;
        uint32_t tv[TPER];                      // (all of a lane's loads in flight, then the LDS stores)
#pragma unroll
        for (uint32_t k = 0; k < TPER; ++k) tv[k] = srcw[min(tid + k * NT, TW - 1)];
#pragma unroll
        for (uint32_t k = 0; k < TPER; ++k) if (tid + k * NT < TW) dst[tid + k * NT] = tv[k];
        __syncthreads();
    } else parse_header<NT>(in, nbytes, job.start_bit, T, lens, hdr, hdr64, tid);
    const uint64_t t_hdr = clock64();
    const BlkLanes *L = &lanes[job.cand];
    int64_t reach = INT64_MAX;
    uint32_t cut_code = 0xFFFFFFFFu;
    uint64_t cut_pos = 0;
    if (tid < job.nlanes) {
        const uint64_t st = L->start[tid];
        const uint64_t lim = tid + 1 < job.nlanes ? L->start[tid + 1] : (job.end_limit ? job.end_limit : ~0ull >> 1);
        uint32_t nc = 0, cc = 0, co = 0;
        const uint64_t out0 = L->out_off[tid];   // bytes of this block produced before my slice
        uint64_t no = out0, endpos;
        if (RING)
            lane_decode<true>(T, in, nbytes, st, lim, nc, no, codes + job.code_off + L->code_off[tid], reach, endpos, cc, co,
                              s_ring + tid * RING_STRIDE, emit_stage + tid * EMIT_STRIDE);
        else
            lane_decode_fifo<true>(T, in, nbytes, st, lim, nc, no, codes + job.code_off + L->code_off[tid], reach, endpos, cc, co,
                                   emit_stage + tid * EMIT_STRIDE);
        if (cc < nc) { cut_code = L->code_off[tid] + cc; cut_pos = out0 + co; }   // a cut behind the last code belongs to the next lane
    }
    blk_flag_reach(tid, reach, job.hist, flags, job_flags);
    blk_units_tail<NT>(tid, job, L, reach, cut_code, cut_pos, U, unit_target, free_shift, t_begin, t_hdr);
}



// ------------------------------------------------------------------------------------------------
// K2' (round 6): the block's codes were STORED by the scan (blk_scan_kernel<true>), one region per lane; this kernel moves
// them to their final places — code_off of the block + code_off of the lane, known since the scan's prefix sums — and does
// what blk_emit_kernel does behind its decode (flags of back-references that leave the block, the block's units).  A copy
// at memory speed in place of a second Huffman pass over the block.
// A wavefront moves the slices of its own 64 lanes one after the other, all 64 lanes on one slice (coalesced both ways); a
// slice's loads are all issued before its first store (one memory round trip per slice, not per 64 codes).
__global__ __launch_bounds__(SCAN_THREADS) void blk_place_kernel(const BlkEmit *__restrict__ jobs, const BlkLanes *__restrict__ lanes,
                                                                  const BlkLanesX *__restrict__ lanesx, const uint32_t *__restrict__ temp,
                                                                  uint32_t *__restrict__ codes, uint32_t *__restrict__ flags,
                                                                  BlkUnits *__restrict__ units, uint32_t unit_target, uint32_t free_shift,
                                                                  uint32_t *__restrict__ job_flags) {
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const BlkEmit job = jobs[blockIdx.x];
    if (!job.placed) return;
    const uint64_t t_begin = clock64();
    BlkUnits *U = &units[blockIdx.x];
    const BlkLanes *L = &lanes[job.cand];
    const BlkLanesX *X = &lanesx[job.cand];
    const bool live = tid < job.nlanes;
    const uint32_t my_head = live ? X->n_head[tid] : 0u, my_rest = live ? X->n_rest[tid] : 0u, my_at = X->rest_at[tid];
    const uint32_t my_off = L->code_off[tid];
    const uint32_t *src0 = temp + job.temp_off + (uint64_t)(tid & ~63u) * job.cap;      // region of the wavefront's first lane
    uint32_t *dst0 = codes + job.code_off;
    constexpr uint32_t DEEP = 8;                       // 512 codes per trip (a 1 MiB block's slice holds about 270)
    // (two workgroups per block — blockIdx.y — each moves half of every wavefront's 64 slices: a wavefront's slices are a
    //  chain of dependent round trips, and 256 blocks alone leave the copy latency-bound; the units are workgroup 0's)
    const uint32_t j0 = blockIdx.y * 32u;
    // two slices per trip: the loads of both are in flight before the first store (a trip is one memory round trip)
    for (uint32_t j = j0; j < j0 + 32u; j += 2) {
        uint32_t nh[2], n[2], at[2];
        const uint32_t *src[2];
        uint32_t *dst[2];
#pragma unroll
        for (uint32_t q = 0; q < 2; ++q) {
            nh[q] = (uint32_t)__shfl((int)my_head, (int)(j + q));
            n[q] = nh[q] + (uint32_t)__shfl((int)my_rest, (int)(j + q));
            at[q] = (uint32_t)__shfl((int)my_at, (int)(j + q));
            src[q] = src0 + (uint64_t)(j + q) * job.cap;
            dst[q] = dst0 + (uint32_t)__shfl((int)my_off, (int)(j + q));
        }
        const uint32_t nmax = n[0] > n[1] ? n[0] : n[1];
        for (uint32_t base = 0; base < nmax; base += 64 * DEEP) {      // (uniform; one trip for slices of up to 512 codes)
            uint32_t v[2][DEEP];
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
#pragma unroll
                for (uint32_t k = 0; k < DEEP; ++k) {
                    const uint32_t i = base + k * 64 + lane;               // index among the slice's codes
                    const uint32_t si = i < nh[q] ? i : at[q] + (i - nh[q]);   // ... and in the lane's region
                    v[q][k] = i < n[q] ? src[q][si] : 0u;
                }
#pragma unroll
            for (uint32_t q = 0; q < 2; ++q)
#pragma unroll
                for (uint32_t k = 0; k < DEEP; ++k) {
                    const uint32_t i = base + k * 64 + lane;
                    if (i < n[q]) dst[q][i] = v[q][k];
                }
        }
    }
    if (blockIdx.y != 0) return;
    const uint64_t t_hdr = clock64();
    // ---- what the emit kernel derives from its decode
    int64_t reach = INT64_MAX;
    uint32_t cut_code = 0xFFFFFFFFu;
    uint64_t cut_pos = 0;
    if (live) {
        const uint64_t out0 = L->out_off[tid];
        const int32_t rr = X->reach[tid];
        if (rr != INT32_MAX) reach = (int64_t)out0 + (int64_t)rr;
        const uint32_t cc = X->cut_code[tid];
        if (cc != 0xFFFFFFFFu) { cut_code = my_off + cc; cut_pos = out0 + X->cut_out[tid]; }
    }
    blk_flag_reach(tid, reach, job.hist, flags, job_flags);
    blk_units_tail(tid, job, L, reach, cut_code, cut_pos, U, unit_target, free_shift, t_begin, t_hdr);
}

// ------------------------------------------------------------------------------------------------
// K3: a 256-lane workgroup per unit, the copy itself data-parallel over BYTES.
//
// (The first-generation kernel — removed in round 3 — gave a unit to ONE wavefront: lanes = codes, back-references that
// read bytes of their own batch executed one after the other, and with a single wavefront per SIMD every instruction's
// latency was exposed: 3360 cycles per batch of 64 codes = 234 bytes.)  Here a tile — a window of 512 codes, of which as
// many are taken as give four bytes per lane (round 5; rounds 2-4: 256 codes and up to six bytes per lane) — is
// expanded to bytes: every lane owns four output bytes, finds the code that covers them (the wavefronts' totals in registers
// pick the wavefront's 128 codes, a binary search in their end offsets the code), and
//   * a literal, or a byte whose source lies in front of the tile (final, in the ring), is written at once;
//   * a byte whose source lies inside the tile gets a POINTER to it: P[i] = i - distance — or, in tiles of long matches, to
//     the byte of the match's first period it repeats (a match that overlaps itself is then one step deep).  Pointer
//     jumping (P[i] = P[P[i]] until the target is resolved) settles these in log2(chain depth) steps — the overlapping
//     forward copy of rle_decode (libflate_lz77/src/lib.rs:186-190) is just a chain of depth length / distance.
// In-place jumping is safe without a second buffer and without holding the wavefronts in step (round 5: each wavefront jumps
// in its own loop): every value a reader can observe in P[j] is either "resolved" — and then the byte is already in the
// ring, because the LDS executes a wavefront's instructions in order — or an earlier byte with the same content.
// Four units per CU (40.1 KB of LDS each), 16 wavefronts per CU instead of 4; their priorities rotate (M2_PRIO_TILES).
// Geometry by workgroup size.  Measured on the 256 MiB corpus (round 3, units = the 1024 LZ77 chunks): 256 lanes, four
// units per CU: 1.03 ms; 512 lanes (tiles of 512 codes, three units per CU): 1.11 ms; 1024 lanes (two units): 1.47 ms.
// Larger tiles amortise the per-tile latencies (barriers, the owner search, the pointer rounds) and still lose: the
// kernel is bound by its VALU instruction count per byte slot and by its barriers (four per tile since round 5), not by
// those latencies.  Round 5: 1.00 -> 0.68 ms.
template <uint32_t THREADS>
struct M2 {
    static constexpr uint32_t WAVES = THREADS / 64;
    static constexpr uint32_t NC = 2 * THREADS;                             // codes a tile looks at (two per lane)
    static constexpr uint32_t TILE = 4 * THREADS;                           // bytes one tile may produce (four per lane)
    static constexpr uint32_t RING = 32768 + TILE;                          // the DEFLATE window + the tile in flight
    static constexpr uint32_t PASSES = (TILE + 4 * THREADS - 1) / (4 * THREADS);
    static_assert(RING % 4 == 0 && TILE >= 258 && TILE < 0xFFFFu && RING + 64 < 65536, "tile");
};
constexpr uint32_t M2_THREADS = 256;                      // direct path
constexpr uint32_t M2_WIDE_THREADS = 1024;                // direct path of a stream of at most M2_FEW_JOBS blocks
constexpr uint32_t M2_FEW_JOBS = 4;
constexpr uint32_t M2_SYM_THREADS = 256;                  // marker path
constexpr uint32_t M2_DONE = 0xFFFFu;
#ifndef LFX_M2_PRIO_TILES
#define LFX_M2_PRIO_TILES 8
#endif
constexpr uint32_t M2_PRIO_TILES = LFX_M2_PRIO_TILES;   // tiles per priority step (a power of two)

template <uint32_t RING> __device__ __forceinline__ uint32_t m2_wrap(uint32_t x) { return min(x, x - RING); }   // x in [0, 2 RING)
template <uint32_t RING> __device__ __forceinline__ uint32_t m2_back(uint32_t idx, uint32_t d) {                 // idx, d < RING
    const uint32_t a = idx - d;
    return min(a, a + RING);
}

// One body for both materialisations: SYM = false writes BYTES (direct path: the unit's history is known or empty),
// SYM = true writes 16-bit SYMBOLS for the marker path (below): the 32 Ki entries in front of the unit start out as the
// markers 256 + j, and the units are the ones cut without regard to back-references (fcode0 / fout0).
// LDS of one unit: ring, pointer states, per-code offsets, two small exchange arrays.  DYN: carved out of the dynamic
// allocation (the 1024-lane symbol variant needs 103 KB, more than a static allocation may hold)
template <bool SYM, uint32_t THREADS>
struct M2Lds {
    using elem_t = typename std::conditional<SYM, uint16_t, uint8_t>::type;
    using G = M2<THREADS>;
    static constexpr uint32_t RING_BYTES = ((G::RING + 64) * (uint32_t)sizeof(elem_t) + 15u) & ~15u;
    static constexpr uint32_t P_BYTES = G::PASSES * 4 * THREADS * 2;
    static constexpr uint32_t XC_BYTES = (G::NC + 4) * 8;
    static constexpr uint32_t SW_BYTES = 2 * G::WAVES * 4;
    static constexpr uint32_t BYTES = RING_BYTES + P_BYTES + XC_BYTES + SW_BYTES;
};
template <bool SYM, uint32_t THREADS, bool DYN = false>
__device__ __forceinline__ void materialize2_body(const uint8_t *__restrict__ in, const BlkEmit *__restrict__ jobs,
                                                  const BlkUnits *__restrict__ units,
                                                  const uint32_t *__restrict__ codes,
                                                  typename std::conditional<SYM, uint16_t, uint8_t>::type *__restrict__ out,
                                                  uint32_t njobs, uint64_t *__restrict__ dbg) {
    using elem_t = typename std::conditional<SYM, uint16_t, uint8_t>::type;
    using G = M2<THREADS>;
    using LD = M2Lds<SYM, THREADS>;
    constexpr uint32_t WAVES = G::WAVES, TILE = G::TILE, RING = G::RING, PASSES = G::PASSES, NC = G::NC;
    constexpr uint32_t EPD = 4 / sizeof(elem_t);                                // elements per dword (flush granule)
    elem_t *ring;             // RING + 64 (+ a dump for the stores of idle bytes)
    uint16_t *P;              // PASSES * 4 * THREADS
    // per code of the window (the NC codes from `base` on): x = its end offset (inclusive prefix sum of the lengths), y = the
    // code word; four sentinels behind the last (never passed)
    uint2 *XC;                // NC + 4
    uint32_t *s_w;            // 2 * WAVES
    if constexpr (DYN) {
        extern __shared__ __attribute__((aligned(16))) uint8_t m2_dyn[];
        ring = (elem_t *)m2_dyn;
        P = (uint16_t *)(m2_dyn + LD::RING_BYTES);
        XC = (uint2 *)(m2_dyn + LD::RING_BYTES + LD::P_BYTES);
        s_w = (uint32_t *)(m2_dyn + LD::RING_BYTES + LD::P_BYTES + LD::XC_BYTES);
    } else {
        __shared__ __attribute__((aligned(16))) elem_t ring_s[RING + 64];
        __shared__ __attribute__((aligned(8))) uint16_t P_s[PASSES * 4 * THREADS];
        __shared__ __attribute__((aligned(16))) uint2 XC_s[G::NC + 4];
        __shared__ uint32_t s_w_s[2 * WAVES];
        ring = ring_s; P = P_s; XC = XC_s; s_w = s_w_s;
    }
    const uint32_t bidx = blockIdx.x % njobs, u = blockIdx.x / njobs;   // unit-major (XCD balance, see K3)
    const BlkEmit job = jobs[bidx];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (job.btype == 0) {
        if (u != 0) return;
        elem_t *o = out + job.out_off;
        const uint8_t *src = in + (job.data_bit >> 3);
        for (uint64_t k = tid; k < job.n_out; k += THREADS) o[k] = src[k];
        return;
    }
    const BlkUnits *U = &units[bidx];
    if (u >= (SYM ? U->fn : U->n)) return;
    const uint32_t c0 = SYM ? U->fcode0[u] : U->code0[u], c1 = SYM ? U->fcode0[u + 1] : U->code0[u + 1];
    const uint64_t ob = SYM ? U->fout0[u] : U->out0[u];
    const uint64_t gbase = job.out_off + ob;
    elem_t *o = out + gbase;
    const uint32_t *cp = codes + job.code_off + c0;
    const uint32_t n = c1 - c0;
    // unit element p lives at ring index (p + shift) mod RING; ring and output share their dword alignment.
    // History in front of the block (batch rounds / ordered runs: already final in `out`): up to 32 KiB preloaded;
    // SYM: the 32 Ki markers "entry j of the window in front of this unit".
    const uint32_t hist = SYM ? 32768u : (u == 0 && job.preload) ? (uint32_t)(job.hist < 32768 ? job.hist : 32768) : 0;
    const uint32_t shift = (uint32_t)((gbase - hist) & (EPD - 1)) + hist;       // < RING
    for (uint32_t k = tid; k < hist; k += THREADS)
        ring[shift - hist + k] = SYM ? (elem_t)(256 + k) : (elem_t)o[(int64_t)k - (int64_t)hist];
    if (tid < 4) XC[NC + tid] = make_uint2(0xFFFFFFFFu, 0u);
    uint64_t produced = 0, flushed = 0;
    uint32_t tr = shift;                       // ring index of the tile's first byte
    uint32_t fr = shift;                       // ring index of byte `flushed`
    uint32_t base = 0, ntiles = 0, nrounds = 0;
    // Round 5: a tile looks at TWO codes per lane (window positions 2 tid, 2 tid + 1) and takes as many as give four bytes per
    // lane.  (One code per lane and six bytes: a text's 256 codes are 640 bytes, the byte-level work below ran with three
    // lanes of eight idle.)
    uint32_t cw0 = 2 * tid < n ? cp[2 * tid] : 0u, cw1 = 2 * tid + 1 < n ? cp[2 * tid + 1] : 0u;
    const uint64_t t0 = dbg ? clock64() : 0;
    while (base < n) {
        // Wavefront priority rotates with the tile count, a quarter turn per unit index: the units that share a CU — four,
        // one of each launch quarter, of equal work and all resident from the start — were served oldest first, the oldest
        // left after 1.49 M cycles, the youngest after 2.02 M, and the CU ran with three, two, one unit for the last quarter
        // of the kernel (LFX_DEBUG K3 lines, round 5).
        if ((ntiles & (M2_PRIO_TILES - 1)) == 0) {
            switch (((ntiles / M2_PRIO_TILES) + u) & 3u) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        ntiles++;
        const uint32_t i = base + 2 * tid;
        const uint32_t len0 = i < n ? ((cw0 & 0xFFFFu) ? cw0 >> 16 : 1u) : 0u;
        const uint32_t len1 = i + 1 < n ? ((cw1 & 0xFFFFu) ? cw1 >> 16 : 1u) : 0u;
        uint32_t x = wave_inclusive_sum(len0 + len1);
        if (lane == 63) s_w[wave] = x;
        __syncthreads();       // (also: the previous tile's flush has read the ring before this tile writes it)
        uint32_t tot = 0, cum[WAVES];          // cum[j]: end offset of wavefront j's last code = X[128 j + 127]
#pragma unroll
        for (uint32_t j = 0; j < WAVES; ++j) {
            const uint32_t wj = s_w[j];
            x += j < wave ? wj : 0u;
            tot += wj;
            cum[j] = tot;
        }
        *(uint4 *)&XC[2 * tid] = make_uint4(x - len1, cw0, x, cw1);
        if (tot > TILE) {   // (uniform) only the codes whose output fits are taken; a code is at most 258 bytes
            const uint32_t cnt = (uint32_t)__popcll(__ballot(x - len1 <= TILE)) + (uint32_t)__popcll(__ballot(x <= TILE));
            if (lane == 0) s_w[WAVES + wave] = cnt;
        }
        __syncthreads();
        uint32_t take = NC, total = tot;
        if (tot > TILE) {
            take = 0;
#pragma unroll
            for (uint32_t j = 0; j < WAVES; ++j) take += s_w[WAVES + j];
            total = XC[take - 1].x;
        }
        // The window moves by `take`: this lane's next two codes are the ones `take` further on — still in XC, or behind the
        // window: those are loaded now, by the lane that will hold them (the round trip hides behind the tile's work; an
        // exchange through LDS and its barrier stood here).
        // (predicated loads, picked against XC only at the tile's end: with the choice up here the wait for them stood in
        //  front of round 0)
        const uint32_t j0 = 2 * tid + take, j1 = j0 + 1;
        const uint32_t rf0 = (j0 >= NC && base + j0 < n) ? cp[base + j0] : 0u;
        const uint32_t rf1 = (j1 >= NC && base + j1 < n) ? cp[base + j1] : 0u;
        // ---- round 0: bytes -> owner code -> literal / final source / pointer.  Branch-free: every byte loads from the
        //      ring (its own slot when there is nothing to fetch) and stores (to the dump when it lies behind the tile).
        uint32_t pp[PASSES][4];
        bool pend = false;
        // (two instances: a tile whose codes are more than eight bytes long on average is made of long matches, and long
        //  matches at short distances are what runs turn into: cfg5's LOWENT is half runs of 64 .. 4096 equal bytes, a chain
        //  as deep as the tile for the jumping below, eleven rounds per tile.  A tile of text — 2.5 bytes per code — never
        //  comes here.)
        auto round0 = [&](auto periodic_tag) {
        constexpr bool PERIODIC = decltype(periodic_tag)::value;
#pragma unroll
        for (uint32_t ps = 0; ps < PASSES; ++ps) {
            const uint32_t b = ps * 4 * THREADS + 4 * tid;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) pp[ps][q] = M2_DONE;
            if (ps == 0 || __ballot(b < total)) {   // (pass 1 and later: whole wavefronts skip)
                uint32_t k = 0;   // smallest k with X[k] > b  (zero-length slots behind the last code are never chosen)
                // the search's first levels — which wavefront's 128 codes — from the totals every lane holds in registers
                // (two dependent LDS round trips less per tile)
#pragma unroll
                for (uint32_t j = 0; j + 1 < WAVES; ++j) k += cum[j] <= b ? 128u : 0u;
#pragma unroll
                for (uint32_t step = 64; step; step >>= 1) k += XC[k + step - 1].x <= b ? step : 0u;
                uint32_t src[4], dst[4], cwq[4];
                bool lit[4], fin[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) {
                    const uint32_t bi = b + q;
                    if (q) k += bi >= cwq[q - 1] ? 1u : 0u;        // (cwq[q-1] still holds the previous owner's end offset)
                    const uint2 xc = XC[k];
                    const uint32_t d = xc.y & 0xFFFFu;
                    const uint32_t t = m2_wrap<RING>(tr + bi);
                    const bool in = bi < total;
                    lit[q] = d == 0;
                    // the byte this one copies: bi - d — or, PERIODIC, the byte of the match's first period it repeats
                    // (offset mod distance): a match that overlaps itself is then one step deep instead of length /
                    // distance.  off < 258 and d <= off here: the float quotient is exact or one too small.
                    uint32_t tgt = bi, tt = t;
                    if constexpr (PERIODIC) {
                        const uint32_t s0 = xc.x - (xc.y >> 16);                // the match's first byte (tile offset)
                        const uint32_t off = bi - s0;
                        if (d != 0 && off >= d) {
                            uint32_t r = off - d * (uint32_t)((float)off * __builtin_amdgcn_rcpf((float)d));
                            r = r >= d ? r - d : r;
                            tgt = s0 + r;
                            tt = m2_wrap<RING>(tr + tgt);
                        }
                    }
                    fin[q] = lit[q] || d > tgt;
                    src[q] = (d > tgt && in) ? m2_back<RING>(tt, d) : t;
                    dst[q] = in ? t : RING + (tid & 63u);
                    pp[ps][q] = (fin[q] || !in) ? M2_DONE : tgt - d;
                    cwq[q] = xc.x;                                  // end offset now, code word's literal below
                    src[q] |= xc.y & 0xFFFF0000u;                   // (ring indices are below 2^16: the value rides along)
                }
                uint32_t hv[4];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) hv[q] = ring[src[q] & 0xFFFFu];
#pragma unroll
                for (uint32_t q = 0; q < 4; ++q) ring[dst[q]] = (elem_t)(lit[q] ? src[q] >> 16 : hv[q]);
                pend |= (pp[ps][0] & pp[ps][1] & pp[ps][2] & pp[ps][3]) != M2_DONE;
            }
            *(uint64_t *)&P[b] = (uint64_t)pp[ps][0] | (uint64_t)pp[ps][1] << 16 | (uint64_t)pp[ps][2] << 32 | (uint64_t)pp[ps][3] << 48;
        }
        };
        if (take * 8 < total) round0(std::true_type{}); else round0(std::false_type{});
        // ---- rounds of pointer jumping.  Per byte: load the target's state, then the target's byte (in this order: a
        //      target seen resolved has its byte in the ring), store the byte, then the state.  A byte fetched from an
        //      unresolved target is garbage in a slot nobody reads yet.
        //      Round 5: the rounds are a WAVEFRONT's own loop.  The order above is all that in-place jumping needs between
        //      wavefronts — a reader that sees a state resolved finds the byte, one that sees a pointer follows it — so
        //      nothing has to hold the wavefronts in step: one barrier behind round 0 (P is rewritten whole there: no state of
        //      the tile before survives it), one when a wavefront's own bytes are all final (the flush reads across
        //      wavefronts).  Rounds 2-4 put a barrier and an exchange of "anyone pending" between all rounds: 3.9 per tile.
        __syncthreads();
        while (__ballot(pend)) {
            nrounds++;
            if (pend) {
#pragma unroll
                for (uint32_t ps = 0; ps < PASSES; ++ps) {
                    const uint32_t b = ps * 4 * THREADS + 4 * tid;
                    if (ps && (pp[ps][0] & pp[ps][1] & pp[ps][2] & pp[ps][3]) == M2_DONE) continue;
                    uint32_t nx[4], hv[4], jj[4];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) {
                        jj[q] = pp[ps][q] != M2_DONE ? pp[ps][q] : b + q;
                        nx[q] = P[jj[q]];
                    }
                    asm volatile("" ::: "memory");   // the states are loaded BEFORE the bytes (the LDS keeps a wavefront's order)
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) hv[q] = ring[m2_wrap<RING>(tr + jj[q])];
#pragma unroll
                    for (uint32_t q = 0; q < 4; ++q) {
                        // (a byte that was final before reads itself — also one behind the tile — and writes the same
                        //  byte and the same state back)
                        ring[m2_wrap<RING>(tr + b + q)] = (elem_t)hv[q];
                        pp[ps][q] = nx[q];
                    }
                    asm volatile("" ::: "memory");   // ... and the bytes are stored BEFORE the states
                    *(uint64_t *)&P[b] = (uint64_t)nx[0] | (uint64_t)nx[1] << 16 | (uint64_t)nx[2] << 32 | (uint64_t)nx[3] << 48;
                }
                pend = false;
#pragma unroll
                for (uint32_t ps = 0; ps < PASSES; ++ps) pend |= (pp[ps][0] & pp[ps][1] & pp[ps][2] & pp[ps][3]) != M2_DONE;
            }
        }
        __syncthreads();
        // ---- flush: whole dwords; what is left over waits for the next tile
        const uint64_t upto = produced + total;
        base += take;
        const bool last = base >= n;
        if ((gbase + flushed) & (EPD - 1)) {   // (only in front of the first aligned dword)
            uint32_t hb = EPD - (uint32_t)((gbase + flushed) & (EPD - 1));
            if (hb > upto - flushed) hb = (uint32_t)(upto - flushed);
            if (tid < hb) o[flushed + tid] = ring[m2_wrap<RING>(fr + tid)];
            flushed += hb;
            fr = m2_wrap<RING>(fr + hb);
        }
        const uint32_t ndw = (uint32_t)((upto - flushed) / EPD);
        uint32_t *o32 = (uint32_t *)(o + flushed);
        for (uint32_t k = tid; k < ndw; k += THREADS) o32[k] = *(const uint32_t *)&ring[m2_wrap<RING>(fr + EPD * k)];
        flushed += (uint64_t)EPD * ndw;
        fr = m2_wrap<RING>(fr + EPD * ndw);
        if (last) {
            const uint32_t rest = (uint32_t)(upto - flushed);
            if (tid < rest) o[flushed + tid] = ring[m2_wrap<RING>(fr + tid)];
            flushed = upto;
        }
        produced = upto;
        tr = m2_wrap<RING>(tr + total);
        cw0 = j0 < NC ? XC[j0].y : rf0;
        cw1 = j1 < NC ? XC[j1].y : rf1;
    }
    if (!SYM && dbg && tid == 0) {
        uint64_t *d = dbg + ((uint64_t)bidx * MAX_UNITS + u) * 8;
        d[0] = clock64() - t0; d[1] = ntiles; d[2] = nrounds; d[3] = 0; d[4] = n; d[5] = produced; d[6] = wall_clock64();
    }
}


template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void blk_materialize2_kernel(const uint8_t *__restrict__ in,
                                                                   const BlkEmit *__restrict__ jobs,
                                                                   const BlkUnits *__restrict__ units,
                                                                   const uint32_t *__restrict__ codes,
                                                                   uint8_t *__restrict__ out, uint32_t njobs,
                                                                   uint64_t *__restrict__ dbg) {
    materialize2_body<false, THREADS>(in, jobs, units, codes, out, njobs, dbg);
}

// ------------------------------------------------------------------------------------------------
// Marker-based materialisation (pugz / rapidgzip style) for streams whose blocks read the output of earlier
// blocks.  Pass 1 — this kernel, every unit at once: the 32 KiB in front of a unit are unknown, so the unit
// works on 16-bit SYMBOLS: a byte value, or 256 + j = "byte j of the 32 KiB in front of me".  The ring starts
// out holding the markers 256 + 0 .. 256 + 32767; back-references then copy symbols exactly as K3 copies
// bytes, and the symbols go to `sym` (one per output byte).  Pass 2 (window_chain_kernel) walks the units in
// order and resolves only each unit's LAST 32 KiB; pass 3 (sym_substitute_kernel) replaces every marker.
// Pass 2: one workgroup walks the units in stream order.  win[u] = the final 32 KiB of output that end where
// unit u ends; byte i of it is a symbol of the unit's own tail resolved through win[u-1], or (for a unit
// shorter than 32 KiB) byte i + len of win[u-1].  The two windows in flight live in LDS.
// init_win: the 32 KiB of output in front of unit 0 (a later window of a stream decoder: the member's earlier output), or
// null at the start of a member (an empty Lz77Decoder buffer: no marker can survive to be looked up there).
__global__ __launch_bounds__(1024) void window_chain_kernel(const uint16_t *__restrict__ sym,
                                                            const SymUnit *__restrict__ units, uint32_t nunits,
                                                            uint8_t *__restrict__ windows, const uint8_t *__restrict__ init_win) {
    extern __shared__ uint8_t wbuf[];   // 2 x 32 KiB
    uint8_t *prev = wbuf, *cur = wbuf + 32768;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) prev[i] = init_win ? init_win[i] : (uint8_t)0;
    // Thread t owns the four window bytes 4 (t + 1024 k) .. +3 of row k = 0..7: one 8-byte symbol load, four LDS
    // gathers, one dword LDS store and one dword global store per row.  The symbols of unit u+1's tail do not
    // depend on the chain: they are loaded while unit u is being resolved, so the walk itself only touches LDS.
    uint32_t sn[16];                    // two symbols per register
    // (round 4) the loads carry no branch: a symbol that lies in front of the unit — the window entry is then a byte of
    // the window before it — is read from index 0 and ignored.  Under a lane-dependent branch every load was followed
    // by its own s_waitcnt: thirty-two HBM round trips per unit, 24 us of a step that computes for one.
    auto fetch = [&](uint32_t u) {
        if (u >= nunits) return;
        const SymUnit su = units[u];
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t i = 4 * (threadIdx.x + 1024 * k);
            const uint64_t at = su.start + su.len - 32768 + i;   // (wraps when the entry lies in front of the unit: not used then)
            const uint32_t s0 = sym[su.len + i >= 32768 ? at : 0];
            const uint32_t s1 = sym[su.len + i + 1 >= 32768 ? at + 1 : 0];
            const uint32_t s2 = sym[su.len + i + 2 >= 32768 ? at + 2 : 0];
            const uint32_t s3 = sym[su.len + i + 3 >= 32768 ? at + 3 : 0];
            sn[2 * k] = s0 | s1 << 16;
            sn[2 * k + 1] = s2 | s3 << 16;
        }
    };
    fetch(0);
    __syncthreads();
    for (uint32_t u = 0; u < nunits; ++u) {
        const uint64_t len = units[u].len;
        uint32_t sc[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) sc[k] = sn[k];
        fetch(u + 1);
        uint32_t *wout = (uint32_t *)(windows + (uint64_t)u * 32768);
#pragma unroll
        for (uint32_t k = 0; k < 8; ++k) {
            const uint32_t i = 4 * (threadIdx.x + 1024 * k);
            uint32_t lb[4], sv[4];
            bool own[4];
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) {
                sv[q] = (sc[2 * k + (q >> 1)] >> (16 * (q & 1))) & 0xFFFFu;
                own[q] = len + i + q >= 32768;                                  // a byte of this unit
                // a marker looks into the window in front; an entry in front of the unit IS a byte of that window
                const uint32_t at = own[q] ? (sv[q] >= 256 ? sv[q] - 256 : 0u) : i + q + (uint32_t)len;
                lb[q] = prev[at];
            }
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4; ++q) packed |= ((own[q] && sv[q] < 256) ? sv[q] : lb[q]) << (8 * q);
            ((uint32_t *)cur)[i >> 2] = packed;
            wout[i >> 2] = packed;
        }
        __syncthreads();
        uint8_t *t = prev; prev = cur; cur = t;
    }
}

// Pass 2 as a blocked parallel prefix (long streams).  A unit acts on the 32 KiB window in front of it as an
// index map F_u: window byte i behind the unit is a literal, or byte j of the window in front — exactly the
// unit's tail symbols (for a unit shorter than 32 KiB the head of the map is the shift i -> i + len).  Maps
// compose by a gather: (F_v o F_u)[i] = F_v[i] if literal, else F_u[F_v[i] - 256].
//  (a) window_compose_kernel: one workgroup per group of WC_GROUP units composes the group's maps in order,
//      C_k = F_k o ... o F_first, and stores every C_k (64 KiB each);
//  (b) window_groups_kernel: one workgroup walks the GROUPS in order: window after group g = C_last(window
//      after group g-1);
//  (c) window_apply_kernel: every unit's window = its C_k applied to the window in front of its group.
// Group size (round 4: chosen per call, about the square root of the number of units — compose costs G dependent steps, the
// walk over the groups nunits / G: 128 units of a 4 MiB stream took 127 serial steps of the chain kernel, 0.95 ms)
__host__ __device__ inline uint32_t wc_group(uint32_t nunits) {
    uint32_t g = 4;
    while (g * g < nunits && g < 64) ++g;
    return g;
}
// entry i of a unit's map: its tail symbol, or — in front of a unit shorter than 32 KiB — the shift i -> i + len.  The load
// carries no branch (an entry in front of the unit reads index 0 and ignores it): the thirty-two loads of a step are
// then in flight together (round 4; one load and one s_waitcnt per entry made every step of the chains below thirty-two
// dependent HBM round trips — the window resolution of a 256 MiB single-block stream took 0.59 ms).
__device__ __forceinline__ uint32_t unit_map_load(const uint16_t *__restrict__ sym, const SymUnit &su, uint32_t i) {
    return sym[su.len + i >= 32768 ? su.start + su.len - 32768 + i : 0];
}
__device__ __forceinline__ uint32_t unit_map_value(uint32_t loaded, const SymUnit &su, uint32_t i) {
    return su.len + i >= 32768 ? loaded : 256u + i + (uint32_t)su.len;
}
__global__ __launch_bounds__(1024) void window_compose_kernel(const uint16_t *__restrict__ sym,
                                                              const SymUnit *__restrict__ units, uint32_t nunits,
                                                              uint16_t *__restrict__ maps, uint32_t WC_GROUP) {
    extern __shared__ uint16_t mbuf[];   // 2 x 32 Ki entries
    uint16_t *prev = mbuf, *cur = mbuf + 32768;
    const uint32_t u0 = blockIdx.x * WC_GROUP, u1 = u0 + WC_GROUP < nunits ? u0 + WC_GROUP : nunits;
    if (u0 >= u1) return;
    // thread t owns the entries t + 1024 k (two per register); the symbols of unit u+1 are loaded while unit u is composed
    uint32_t nx[16];
    auto load_unit = [&](const SymUnit &su) {
        if (su.len >= 32768) {                                   // (uniform) every entry is a symbol of the unit's tail
            const uint16_t *p = sym + (su.start + su.len - 32768) + threadIdx.x;
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k) nx[k] = (uint32_t)p[1024 * (2 * k)] | (uint32_t)p[1024 * (2 * k + 1)] << 16;
        } else {
#pragma unroll
            for (uint32_t k = 0; k < 16; ++k)
                nx[k] = unit_map_load(sym, su, threadIdx.x + 1024 * (2 * k)) | unit_map_load(sym, su, threadIdx.x + 1024 * (2 * k + 1)) << 16;
        }
    };
    SymUnit sun = units[u0];
    load_unit(sun);
    for (uint32_t u = u0; u < u1; ++u) {
        const SymUnit su = sun;
        uint32_t sc[16];
#pragma unroll
        for (uint32_t k = 0; k < 16; ++k) sc[k] = nx[k];
        if (u + 1 < u1) {
            sun = units[u + 1];
            load_unit(sun);
        }
        uint16_t *mout = maps + (uint64_t)u * 32768;
        const bool first = u == u0;
#pragma unroll
        for (uint32_t h = 0; h < 4; ++h) {                       // eight entries at a time (register pressure)
            uint32_t sv[8], g[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t k = 8 * h + j;
                sv[j] = unit_map_value((sc[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, su, threadIdx.x + 1024 * k);
                g[j] = prev[sv[j] >= 256 ? sv[j] - 256 : 0u];
            }
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                const uint32_t k = 8 * h + j;
                const uint32_t v = (!first && sv[j] >= 256) ? g[j] : sv[j];
                cur[threadIdx.x + 1024 * k] = (uint16_t)v;
                mout[threadIdx.x + 1024 * k] = (uint16_t)v;
            }
        }
        __syncthreads();
        uint16_t *t = prev; prev = cur; cur = t;
    }
}
// eight map entries (one 16-byte load) as eight symbols
__device__ __forceinline__ void map_octet(const uint4 &q, uint32_t (&s8)[8]) {
    s8[0] = q.x & 0xFFFFu; s8[1] = q.x >> 16; s8[2] = q.y & 0xFFFFu; s8[3] = q.y >> 16;
    s8[4] = q.z & 0xFFFFu; s8[5] = q.z >> 16; s8[6] = q.w & 0xFFFFu; s8[7] = q.w >> 16;
}
// one step of a chain over BYTE windows in LDS: cur = map(prev); thread t owns the octets t + 1024 k of the 4096
__device__ __forceinline__ void window_step_bytes(const uint4 (&mq)[4], const uint8_t *prev, uint8_t *cur, uint8_t *gout) {
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t o = threadIdx.x + 1024 * k;
        uint32_t s8[8], b8[8];
        map_octet(mq[k], s8);
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) b8[j] = prev[s8[j] >= 256 ? s8[j] - 256 : 0u];
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            lo |= ((s8[j] < 256 ? s8[j] : b8[j]) & 0xFFu) << (8 * j);
            hi |= ((s8[j + 4] < 256 ? s8[j + 4] : b8[j + 4]) & 0xFFu) << (8 * j);
        }
        *(uint2 *)(cur + 8 * o) = make_uint2(lo, hi);
        if (gout) *(uint2 *)(gout + 8 * o) = make_uint2(lo, hi);
    }
}
__global__ __launch_bounds__(1024) void window_groups_kernel(const uint16_t *__restrict__ maps, uint32_t nunits,
                                                             uint8_t *__restrict__ gwin, const uint8_t *__restrict__ init_win,
                                                             uint32_t WC_GROUP) {
    extern __shared__ __attribute__((aligned(16))) uint8_t wbuf[];   // 2 x 32 KiB
    uint8_t *prev = wbuf, *cur = wbuf + 32768;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) prev[i] = init_win ? init_win[i] : (uint8_t)0;
    const uint32_t ngroups = (nunits + WC_GROUP - 1) / WC_GROUP;
    auto last_of = [&](uint32_t g) { return (g + 1) * WC_GROUP < nunits ? (g + 1) * WC_GROUP - 1 : nunits - 1; };
    // (the maps do not depend on the chain: group g+1's is loaded while group g is resolved)
    uint4 nx[4];
    if (ngroups) {
        const uint4 *m = (const uint4 *)(maps + (uint64_t)last_of(0) * 32768);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) nx[k] = m[threadIdx.x + 1024 * k];
    }
    __syncthreads();
    for (uint32_t g = 0; g < ngroups; ++g) {
        uint4 mq[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) mq[k] = nx[k];
        if (g + 1 < ngroups) {
            const uint4 *m = (const uint4 *)(maps + (uint64_t)last_of(g + 1) * 32768);
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) nx[k] = m[threadIdx.x + 1024 * k];
        }
        window_step_bytes(mq, prev, cur, gwin + (uint64_t)g * 32768);
        __syncthreads();
        uint8_t *t = prev; prev = cur; cur = t;
    }
}
// every unit's window = its composed map applied to the window in front of its group (staged in LDS: the gathers are
// random bytes of 32 KiB)
__global__ __launch_bounds__(1024) void window_apply_kernel(const uint16_t *__restrict__ maps, const uint8_t *__restrict__ gwin,
                                                            uint8_t *__restrict__ windows, const uint8_t *__restrict__ init_win,
                                                            uint32_t WC_GROUP) {
    __shared__ __attribute__((aligned(16))) uint8_t wl[32768];
    const uint32_t u = blockIdx.x, g = u / WC_GROUP;
    // (group 0: the member's earlier output, or nothing in front — then no markers are left)
    const uint8_t *w = g ? gwin + (uint64_t)(g - 1) * 32768 : init_win;
    const uint4 *m = (const uint4 *)(maps + (uint64_t)u * 32768);
    uint4 mq[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) mq[k] = m[threadIdx.x + 1024 * k];
    {
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4 *w16 = (const uint4 *)w;            // (gwin rows and init_win: 16-byte aligned? gwin yes; init_win by bytes)
        if (w && ((uint64_t)w & 15) == 0) {
            ((uint4 *)wl)[threadIdx.x] = w16[threadIdx.x];
            ((uint4 *)wl)[threadIdx.x + 1024] = w16[threadIdx.x + 1024];
        } else if (w) {
            for (uint32_t i = threadIdx.x; i < 32768; i += 1024) wl[i] = w[i];
        } else {
            ((uint4 *)wl)[threadIdx.x] = z;
            ((uint4 *)wl)[threadIdx.x + 1024] = z;
        }
    }
    __syncthreads();
    uint8_t *wout = windows + (uint64_t)u * 32768;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t o = threadIdx.x + 1024 * k;
        uint32_t s8[8], b8[8];
        map_octet(mq[k], s8);
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) b8[j] = wl[s8[j] >= 256 ? s8[j] - 256 : 0u];
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            lo |= ((s8[j] < 256 ? s8[j] : b8[j]) & 0xFFu) << (8 * j);
            hi |= ((s8[j + 4] < 256 ? s8[j + 4] : b8[j + 4]) & 0xFFu) << (8 * j);
        }
        *(uint2 *)(wout + 8 * o) = make_uint2(lo, hi);
    }
}

// ---- N-GPU decode, window hand-over between ranks (round 4).  A rank's slice acts on the 32 KiB of output in
// front of it as ONE index map (the composition of its units' maps): the ranks all-gather these maps (64 KiB each) and
// every rank composes the maps of the ranks in front of it into the bytes of its own initial window.
//  window_rank_map_kernel: the groups' composed maps (window_compose_kernel) folded in order, symbolically → the rank's map
__global__ __launch_bounds__(1024) void window_rank_map_kernel(const uint16_t *__restrict__ maps, uint32_t nunits,
                                                               uint16_t *__restrict__ out_map, uint32_t WC_GROUP) {
    extern __shared__ __attribute__((aligned(16))) uint16_t mbuf2[];   // 2 x 32 Ki entries
    uint16_t *prev = mbuf2, *cur = mbuf2 + 32768;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) prev[i] = (uint16_t)(256u + i);   // identity: "byte i of the window in front"
    const uint32_t ngroups = (nunits + WC_GROUP - 1) / WC_GROUP;
    auto last_of = [&](uint32_t g) { return (g + 1) * WC_GROUP < nunits ? (g + 1) * WC_GROUP - 1 : nunits - 1; };
    uint4 nx[4];
    if (ngroups) {
        const uint4 *m = (const uint4 *)(maps + (uint64_t)last_of(0) * 32768);
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) nx[k] = m[threadIdx.x + 1024 * k];
    }
    __syncthreads();
    for (uint32_t g = 0; g < ngroups; ++g) {
        uint4 mq[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) mq[k] = nx[k];
        if (g + 1 < ngroups) {
            const uint4 *m = (const uint4 *)(maps + (uint64_t)last_of(g + 1) * 32768);
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) nx[k] = m[threadIdx.x + 1024 * k];
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t o = threadIdx.x + 1024 * k;
            uint32_t s8[8], g8[8];
            map_octet(mq[k], s8);
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) g8[j] = prev[s8[j] >= 256 ? s8[j] - 256 : 0u];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) s8[j] = s8[j] < 256 ? s8[j] : g8[j];
            *(uint4 *)(cur + 8 * o) = make_uint4(s8[0] | s8[1] << 16, s8[2] | s8[3] << 16, s8[4] | s8[5] << 16, s8[6] | s8[7] << 16);
        }
        __syncthreads();
        uint16_t *t = prev; prev = cur; cur = t;
    }
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) out_map[i] = prev[i];
}
//  a slice that was materialised directly (bytes): its map is its last 32 KiB as literals (a slice shorter than 32 KiB:
//  the head of the map is the shift i -> i + len; an empty slice: the identity)
__global__ __launch_bounds__(256) void bytes_to_map_kernel(const uint8_t *__restrict__ out, uint64_t len, uint16_t *__restrict__ map) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 32768) return;
    map[i] = len + i >= 32768 ? (uint16_t)out[len + i - 32768] : (uint16_t)(256u + i + (uint32_t)len);
}
//  the window in front of rank `nranks`'s slice: the maps of ranks 0 .. nranks-1 applied in order to "nothing" (a member
//  starts with an empty Lz77Decoder buffer: a marker that survives to the start is never looked up — K2 flagged it)
__global__ __launch_bounds__(1024) void window_ranks_kernel(const uint16_t *__restrict__ maps, uint32_t nranks, uint8_t *__restrict__ win) {
    extern __shared__ __attribute__((aligned(16))) uint8_t wbuf3[];   // 2 x 32 KiB
    uint8_t *prev = wbuf3, *cur = wbuf3 + 32768;
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) prev[i] = 0;
    uint4 nx[4];
    if (nranks) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) nx[k] = ((const uint4 *)maps)[threadIdx.x + 1024 * k];
    }
    __syncthreads();
    for (uint32_t r = 0; r < nranks; ++r) {
        uint4 mq[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) mq[k] = nx[k];
        if (r + 1 < nranks) {
            const uint4 *m = (const uint4 *)(maps + (uint64_t)(r + 1) * 32768);
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) nx[k] = m[threadIdx.x + 1024 * k];
        }
        window_step_bytes(mq, prev, cur, nullptr);
        __syncthreads();
        uint8_t *t = prev; prev = cur; cur = t;
    }
    for (uint32_t i = threadIdx.x; i < 32768; i += 1024) win[i] = prev[i];
}

// Pass 3: every marker is replaced through the window in front of its unit.
__global__ __launch_bounds__(256) void sym_substitute_kernel(const uint16_t *__restrict__ sym,
                                                             const SymUnit *__restrict__ units,
                                                             const uint8_t *__restrict__ windows,
                                                             uint8_t *__restrict__ out, const uint8_t *__restrict__ init_win) {
    const uint32_t u = blockIdx.x;
    const SymUnit su = units[u];
    // (unit 0: the member's earlier output; at the start of a member it holds no markers)
    const uint8_t *w = u ? windows + (uint64_t)(u - 1) * 32768 : (init_win ? init_win : windows);
    const uint64_t lo = (uint64_t)blockIdx.y * 16384, hi = lo + 16384 < su.len ? lo + 16384 : su.len;
    if (lo >= hi) return;
    // Eight symbols per lane and trip (round 3): one 16-byte load, the (up to eight) window bytes gathered together, one
    // 8-byte store.  One symbol per lane and trip was 64 dependent round trips per lane for 64 bytes of output.
    // Octets are aligned on the ABSOLUTE symbol index (the arrays are 16-byte aligned); the elements of the first and last
    // octet that belong to a neighbouring range are masked.
    const uint64_t A = su.start + lo, B = su.start + hi;
    const bool out_aligned = ((uint64_t)out & 7) == 0;        // (the caller's buffer: usually, not necessarily)
    for (uint64_t p = (A & ~7ull) + 8ull * threadIdx.x; p < B; p += 8ull * 256) {
        const uint4 q = *(const uint4 *)(sym + p);
        const uint32_t s8[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
        uint32_t b8[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) b8[j] = w[s8[j] >= 256 ? min(s8[j] - 256, 32767u) : 0];   // (a plain byte reads window entry 0: ignored; an
                                                                                                 //  element outside [A, B) may be anything: clamped)
        uint64_t v = 0;
#pragma unroll
        for (uint32_t j = 0; j < 8; ++j) v |= (uint64_t)((s8[j] < 256 ? s8[j] : b8[j]) & 0xFFu) << (8 * j);
        if (p >= A && p + 8 <= B && out_aligned) *(uint64_t *)(out + p) = v;
        else {
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) if (p + j >= A && p + j < B) out[p + j] = (uint8_t)(v >> (8 * j));
        }
    }
}

// Per-lane decode table of a COMPLETE code-length code (stage 1 of the finder guarantees completeness):
// entry e of lane l lives at tab[e * 64 + l] (consecutive lanes in consecutive bytes) and holds
// symbol | width << 5 for the 7 stream bits e.  Canonical codes (symbol.rs:354-369) are assigned per width
// in symbol order; a code of width w owns the 2^(7-w) entries whose low w bits are its reversed bits.
__device__ __forceinline__ void clen_table_build(uint8_t *tab, uint32_t lane, uint64_t clw) {
    uint64_t cnt = 0;   // symbols per width, 5 bits at 5*w
    for (uint32_t s = 0; s < 19; ++s) { const uint32_t w = (uint32_t)(clw >> (3 * s)) & 7; if (w) cnt += 1ull << (5 * w); }
    uint64_t next = 0;  // next code per width, 8 bits at 8*w
    uint32_t code = 0;
    for (uint32_t w = 1; w <= 7; ++w) {
        code <<= 1;
        next |= (uint64_t)code << (8 * w);
        code += (uint32_t)(cnt >> (5 * w)) & 31;
    }
    // base entries: a code of width w sits at entry r = its reversed bits (r < 2^w) ...
    for (uint32_t s = 0; s < 19; ++s) {
        const uint32_t w = (uint32_t)(clw >> (3 * s)) & 7;
        const uint32_t c = (uint32_t)(next >> (8 * w)) & 255;
        next += w ? 1ull << (8 * w) : 0ull;
        const uint32_t r = __brev(c) >> (32 - w);
        if (w) tab[r * 64 + lane] = (uint8_t)(s | w << 5);
    }
    // ... and is replicated upward: entry i + 2^k (i < 2^k) decodes like entry i iff that code is at most k bits wide
    // (otherwise bit k belongs to the code and i + 2^k is a base entry of its own); the code is complete (stage 1), so every
    // entry ends up written.  Uniform steps instead of one divergent fill loop per symbol — and (round 5) a level's reads in
    // batches of sixteen before its writes: one read and one conditional write per step were 127 dependent LDS round trips.
#pragma unroll
    for (uint32_t k = 0; k < 7; ++k) {
#pragma unroll
        for (uint32_t i0 = 0; i0 < (1u << k); i0 += 16) {
            uint8_t e[16];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j)
                if (i0 + j < (1u << k)) e[j] = tab[(i0 + j) * 64 + lane];
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j)
                if (i0 + j < (1u << k) && (uint32_t)(e[j] >> 5) <= k) tab[(i0 + j + (1u << k)) * 64 + lane] = e[j];
        }
    }
}

// block finder, stage 2: full header parse of each stage-1 survivor, one lane each (`valid` lanes; the others walk along
// frozen): the code-length sequence must decode to exactly HLIT+257+HDIST+1 lengths, EOB must have a code, the literal/length
// code must be complete and the distance code complete, single or empty, and the header must end inside the stream.
// Round 5: the lanes' bits come from LDS.  Every lane stages FIND2_HB dwords of the stream from its candidate on with all loads
// in flight (hbuf: dword k of lane l at k * 64 + l), takes the header's fixed fields from them by funnel shifts, and walks on
// them; a lane that uses its dwords up makes every lane still walking stage again from where it stands (a few times per
// batch).  Rounds 3-4 topped the window up from global memory inside the walk: one dependent, uncoalesced load and its wait
// per step of the wavefront — the stage's time was that latency, about 200 times per batch, and the generic bit reader's in
// front of it.
// Round 6: EVERY candidate takes this path — the loads are clamped to the stream's last dword and a header that ends behind
// the stream's last bit fails at the end (the bits a walk reads behind the end are the last dword's, over and over: whatever
// it makes of them is discarded).  Until then a wavefront that held a candidate within 6000 bits of the stream's end took a
// second, byte-wise walk with every step bounds-checked: three or four wavefronts per stream, ~100 us each — two thirds of
// the kernel's time whenever one of them started late (found with batches that do nothing, LFX_FIND2_EXP=4: 102 us).
#ifndef LFX_FIND2_HB
#define LFX_FIND2_HB 16
#endif
constexpr uint32_t FIND2_HB = LFX_FIND2_HB;   // dwords of the stream a lane stages at a time (stage2_check_staged)
template <int EXP>      // EXP: timing experiments (LFX_FIND2_EXP: a cut-down kernel runs in front of the real one): 1 = no walk (every candidate
                        // fails), 2 = no table either, 3 = no staging either, 4 = no batch counter either (batches by stride)
__device__ __forceinline__ bool stage2_check_staged(const uint8_t *__restrict__ in, uint64_t nbytes, uint64_t cand_bit, bool valid,
                                                    uint8_t *cl_tab, uint32_t *hbuf, uint64_t *dbg, uint32_t *work, uint32_t &next_raw) {
    const uint32_t lane = threadIdx.x;
    const uint64_t t0 = dbg ? clock64() : 0;
    const uint64_t a = (uint64_t)in;
    const uint64_t abs0 = cand_bit + (a & 3) * 8;
    gptr_u32 w = (gptr_u32)(a & ~3ull);
    uint64_t gd = abs0 >> 5;                         // dword of the stream that hbuf[0] holds
    const uint32_t off = (uint32_t)abs0 & 31;
    const uint64_t wlast = ((a & 3) + nbytes + 3) / 4 - 1;       // the stream's last (aligned) dword: loads are clamped to it
    auto stage = [&](bool need) {
        uint32_t v[FIND2_HB];
#pragma unroll
        for (uint32_t k = 0; k < FIND2_HB; ++k) { const uint64_t idx = gd + k; v[k] = need ? w[idx < wlast ? idx : wlast] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < FIND2_HB; ++k)
            if (need) hbuf[k * 64 + lane] = v[k];
    };
    stage(valid && EXP < 3);
    const uint32_t e0 = hbuf[lane], e1 = hbuf[64 + lane], e2 = hbuf[128 + lane], e3 = hbuf[192 + lane];
    const uint32_t x0 = __builtin_amdgcn_alignbit(e1, e0, off), x1 = __builtin_amdgcn_alignbit(e2, e1, off),
                   x2 = __builtin_amdgcn_alignbit(e3, e2, off);
    const uint32_t nl = ((x0 >> 3) & 31) + 257, nd = ((x0 >> 8) & 31) + 1, nc = ((x0 >> 13) & 15) + 4;
    const uint32_t f0 = __builtin_amdgcn_alignbit(x1, x0, 17), f1 = __builtin_amdgcn_alignbit(x2, x1, 17);   // the 3-bit widths
    uint64_t clw = 0;
#pragma unroll
    for (uint32_t k = 0; k < 19; ++k) {
        const uint32_t fw = 3 * k + 3 <= 32 ? f0 >> (3 * k) : 3 * k >= 32 ? f1 >> (3 * k - 32) : __builtin_amdgcn_alignbit(f1, f0, 3 * k);
        clw |= (uint64_t)(k < nc ? fw & 7u : 0u) << (3 * f_clen_order_c(k));
    }
    const uint64_t t1 = dbg ? clock64() : 0;
    if (EXP < 2) clen_table_build(cl_tab, lane, clw);
    // the wavefront's next batch is reserved HERE — behind the staging loads and the table, in front of the walk, which only
    // touches LDS: the counter's answer (a device-scope atomic, thousands on one address) arrives while the wavefront walks.
    // (In front of the staging loads — rounds 3-5 — the loads' wait was the atomic's wait too: memory operations return in order.)
    // (the address through an opaque zero: on a uniform address the compiler's atomic optimizer rewrites the call into "one lane
    //  adds for all, broadcast, every lane derives its value" and waits for the answer on the spot)
    {
        uint32_t zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        if (lane == 0 && EXP < 4) next_raw = atomicAdd(work + zero, 1u);
    }
    const uint64_t t2 = dbg ? clock64() : 0;
    uint32_t steps = 0, restaged = 0;
    uint32_t have = 0, kl = 0, kd = 0, nlit = 0, ndist = 0, eob_len = 0, last = 0;
    const uint32_t total = nl + nd;
    bool good = valid;
    // the window behind the fixed fields: 32 bits from bit `o` of the dword pair (cur, nxt) by ONE funnel shift — a step takes at
    // most 7 + 7 bits.  (A 64-bit register window cost three 64-bit shifts per step, quarter-rate instructions: a sixth of the
    // loop's issue time.)
    const uint32_t bp = off + 17 + 3 * nc;             // (at most 31 + 17 + 57: dwords 0 .. 3)
    uint32_t di = (bp >> 5) + 2;                       // next dword of hbuf to take
    uint32_t cur = hbuf[(bp >> 5) * 64 + lane], nxt = hbuf[((bp >> 5) + 1) * 64 + lane];
    uint32_t o = bp & 31;
    // Round 6: the literal / length widths first, in a loop that carries only what they need — nearly every false candidate
    // walks until those widths are complete (their Kraft sum only shows then) and the wavefront's time is its instruction
    // count per step: no split of a run at the literal / distance boundary, no distance sums, no bound on the total here.
    // A lane stops IN FRONT of the step that reaches HLIT + 257 (its state untouched: the general loop below takes that
    // step, the completeness check and the distance widths).  acc = sum of rep * (weight << 9 | 1): Kraft sum and the
    // number of codes in one multiply-add; `last` starts as 31, which only a repeat-previous code at the very start can
    // copy into val.
#if !defined(LFX_FIND2_NO_LEAN)
    {
        uint32_t acc = 0;
        last = 31u;
        if (EXP >= 1) good = false;
        bool runA = good;
        uint32_t badv = 0;                 // (the only verdict carried through the loop besides runA: `good` is settled behind it)
        // (the re-staging stands OUTSIDE the step loop: with the branch inside it the compiler merged the two paths with a copy of
        //  every loop-carried value per step — ten v_mov; 59 → 48 vector instructions per step, stage 2 0.322 → 0.294 ms)
        while (__ballot(runA)) {
            if (__ballot(runA && di >= FIND2_HB)) {
                ++restaged;
                gd += di;
                di = runA ? 0u : di;
                gd -= di;
                stage(runA);
            }
            do {
                ++steps;
                const uint32_t win = __builtin_amdgcn_alignbit(nxt, cur, o);
                const uint32_t e = cl_tab[(win & 127) * 64 + lane];
                const uint32_t ahead = hbuf[min(di, FIND2_HB - 1) * 64 + lane];
                const uint32_t sym = e & 31, used = e >> 5;
                const uint32_t k4 = (sym > 15 ? sym - 15 : 0) * 4;
                const uint32_t nbx = (0x7320u >> k4) & 15, basex = (0xB331u >> k4) & 15;
                uint32_t rep = basex + __builtin_amdgcn_ubfe(win, used, nbx);
                const uint32_t val = sym < 16 ? sym : (sym == 16 ? last : 0);
                const bool take = runA && have + rep < nl;           // (else: the step that completes the widths — the loop below)
                rep = take ? rep : 0u;
                const uint32_t adv = take ? used + nbx : 0u;
                o += adv;
                const bool pass = o >= 32;
                cur = pass ? nxt : cur;
                nxt = pass ? ahead : nxt;
                di += pass ? 1u : 0u;
                o &= 31u;
                const uint32_t w9 = val ? (0x1000000u >> val) | 1u : 0u;
                acc += __umul24(rep, w9);
                eob_len = (256u - have) < rep ? val : eob_len;       // (have <= 256 < have + rep, once; unsigned: false behind 256)
                have += rep;
                last = take ? val : last;
                const bool bad = val == 31u || acc > (32768u << 9 | 511u);             // repeat-previous at the start; over-subscribed
                badv = take && bad ? 1u : badv;
                runA = take && !bad;
            } while (__ballot(runA) && !__ballot(runA && di >= FIND2_HB));
        }
        good = good && badv == 0;
        kl = acc >> 9;
        nlit = acc & 511u;
        last = last == 31u ? 0u : last;                          // (nothing walked: the general loop checks sym 16 itself)
    }
#endif
    bool run = good && have < total;
    while (__ballot(run)) {
        ++steps;
        if (__ballot(run && di >= FIND2_HB)) {         // (uniform, rare) a lane's dwords are used up
            ++restaged;
            gd += di;
            di = run ? 0u : di;
            gd -= di;                                  // (a lane that has stopped keeps its place: it is not staged)
            stage(run);
        }
        const uint32_t win = __builtin_amdgcn_alignbit(nxt, cur, o);
        const uint32_t e = cl_tab[(win & 127) * 64 + lane];
        const uint32_t ahead = hbuf[min(di, FIND2_HB - 1) * 64 + lane];      // (the dword behind nxt, taken when o passes 32)
        const uint32_t sym = e & 31, used = e >> 5;
        // repeat codes 16 / 17 / 18: extra bits 2 / 3 / 7, base count 3 / 3 / 11 (packed nibble tables)
        const uint32_t k4 = (sym > 15 ? sym - 15 : 0) * 4;
        const uint32_t nbx = (0x7320u >> k4) & 15, basex = (0xB331u >> k4) & 15;
        uint32_t rep = basex + __builtin_amdgcn_ubfe(win, used, nbx);
        uint32_t val = sym < 16 ? sym : (sym == 16 ? last : 0);
        uint32_t adv = used + nbx;
        bool bad = run && ((sym == 16 && have == 0) || have + rep > total);
        rep = run ? rep : 0u;
        val = run ? val : 0u;
        adv = run ? adv : 0u;
        o += adv;
        const bool pass = o >= 32;
        cur = pass ? nxt : cur;
        nxt = pass ? ahead : nxt;
        di += pass ? 1u : 0u;
        o &= 31u;
        // [have, have+rep) split at the literal / distance boundary
        const uint32_t nlit_part = have < nl ? min(rep, nl - have) : 0u;
        const uint32_t ndist_part = rep - nlit_part;
        const uint32_t wgt = val ? 32768u >> val : 0u;
        kl += __umul24(nlit_part, wgt);
        kd += __umul24(ndist_part, wgt);
        nlit += val ? nlit_part : 0u;
        ndist += val ? ndist_part : 0u;
        eob_len = (val && have <= 256 && 256 < have + rep) ? val : eob_len;
        bad |= kl > 32768u || kd > 32768u;                       // over-subscribed
        // the literal / length widths are complete once `have` passes HLIT+257
        bad |= have + rep >= nl && have < nl && !(kl == 32768u || (nlit == 1 && kl == 16384u));
        have += rep;
        last = val;
        good = good && !bad;
        run = good && have < total;
    }
    if (good) {
        if (eob_len == 0) good = false;
        if (!(kl == 32768u || (nlit == 1 && kl == 16384u))) good = false;
        if (!(kd == 32768u || (ndist == 1 && kd == 16384u) || ndist == 0)) good = false;
        // the header's last bit lies inside the stream (cur is dword gd + di - 2 of the aligned grid, o the offset in it)
        if (((gd + di - 2) << 5) + o > nbytes * 8 + (a & 3) * 8) good = false;
    }
    if (dbg && lane == 0) {   // LFX_DEBUG: cycles of staging + fields, table, walk; steps, restagings, batches
        const uint64_t t3 = clock64();
        atomicAdd((unsigned long long *)&dbg[0], (unsigned long long)(t1 - t0));
        atomicAdd((unsigned long long *)&dbg[1], (unsigned long long)(t2 - t1));
        atomicAdd((unsigned long long *)&dbg[2], (unsigned long long)(t3 - t2));
        atomicAdd((unsigned long long *)&dbg[3], (unsigned long long)steps);
        atomicAdd((unsigned long long *)&dbg[4], (unsigned long long)restaged);
        atomicAdd((unsigned long long *)&dbg[5], 1ull);
    }
    return good;
}

// Persistent grid (round 3): the number of survivors is only known on the device, so a fixed number of one-wavefront
// workgroups fetch batches of 64 survivors from a device counter until the lists are exhausted — stage 1 and stage 2 run
// back to back without the host reading the counts in between (it reads them, the overflow marker and the result list
// in ONE round trip afterwards).
template <int EXP>
__global__ __launch_bounds__(64) void find_blocks_stage2(const uint8_t *__restrict__ in, uint64_t nbytes,
                                                         const uint64_t *__restrict__ cand, uint32_t shard_cap,
                                                         const uint32_t *__restrict__ count, uint32_t *__restrict__ work,
                                                         uint32_t *__restrict__ final_count,
                                                         uint64_t *__restrict__ final_list, uint32_t final_cap,
                                                         uint64_t *__restrict__ dbg) {
    __shared__ uint8_t cl_tab[128 * 64];
    __shared__ uint32_t hbuf[FIND2_HB * 64];
    const uint64_t tk0 = dbg ? clock64() : 0;
    __shared__ uint32_t s_pre[FIND_SHARDS + 1];
    {
        // exclusive prefix of the lists' counts (FIND_SHARDS / 64 lists per lane)
        constexpr uint32_t PER = FIND_SHARDS / 64;
        static_assert(PER * 64 == FIND_SHARDS, "lists per lane");
        uint32_t cnt[PER], sum = 0;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) { cnt[q] = min(count[threadIdx.x * PER + q], shard_cap); sum += cnt[q]; }   // (an overflow is the host's to report)
        const uint32_t incl = wave_inclusive_sum(sum);
        uint32_t at = incl - sum;
#pragma unroll
        for (uint32_t q = 0; q < PER; ++q) { s_pre[threadIdx.x * PER + q] = at; at += cnt[q]; }
        if (threadIdx.x == 63) s_pre[FIND_SHARDS] = incl;
    }
    __syncthreads();
    const uint32_t n1 = s_pre[FIND_SHARDS];
    // Batches of 64 survivors: a wavefront's first one by its index, the later ones from a device counter — asked for while
    // the wavefront walks.  (Rounds 3-4 fetched every batch, the first included, from the counter and waited for it: the grid's first atomics,
    // thousands on one address at about 11 ns each, arrived together — LFX_DEBUG: the wavefronts' lives summed to 1.6 times their
    // batches' cycles.  A fixed deal by stride alone loses a third to rounding: 2.4 batches per wavefront are three rounds.)
    // Round 6: FIND2_GROUPS counters.  Wavefront w belongs to group k = w mod GROUPS and is its r-th member; the group's batches
    // are k, k + GROUPS, k + 2 GROUPS, ... (the lists are sorted by class, longest walks first: every group gets the same mix),
    // its first members' by rank, the later ones from the group's counter.
    const uint32_t grp = blockIdx.x % FIND2_GROUPS, members = (gridDim.x - grp + FIND2_GROUPS - 1) / FIND2_GROUPS;
    uint32_t *my_work = work + grp * FIND_HDR_WORK_STRIDE;
    uint32_t base = (grp + FIND2_GROUPS * (blockIdx.x / FIND2_GROUPS)) * 64u;      // (= blockIdx.x * 64)
    while (base < n1) {
        uint32_t next_raw = 0;                            // lane 0: the work counter's answer (the batch behind this one)
        const uint32_t gi = base + threadIdx.x;
        const bool valid = gi < n1;
        uint32_t shard = 0;                               // the last list whose first index is <= gi (empty lists share theirs)
#pragma unroll
        for (uint32_t step = FIND_SHARDS / 2; step; step >>= 1) shard += s_pre[shard + step] <= gi ? step : 0u;
        const uint64_t i = (uint64_t)shard * shard_cap + (gi - s_pre[shard]);
        const uint64_t cand_bit = valid ? cand[i] : 0;
        const bool good = stage2_check_staged<EXP>(in, nbytes, cand_bit, valid, cl_tab, hbuf, dbg, my_work, next_raw);
        if (good) {
            const uint32_t slot = atomicAdd(final_count, 1u);   // a few hundred per stream
            if (slot < final_cap) final_list[slot] = cand_bit;
        }
        __syncthreads();      // (the per-lane tables are reused)
        if (EXP == 4) base += gridDim.x * 64u;                    // (timing: a fixed deal by stride, no counter)
        else base = (grp + FIND2_GROUPS * (members + (uint32_t)__builtin_amdgcn_readfirstlane((int)next_raw))) * 64u;
    }
    if (dbg && threadIdx.x == 0) atomicAdd((unsigned long long *)&dbg[6], (unsigned long long)(clock64() - tk0));   // a wavefront's life
}

// marker path, pass 1, second generation: the byte kernel's tiles on 16-bit symbols (75 KB of LDS: two units per CU,
// eight wavefronts per CU where the first-generation kernel above runs two)
__global__ __launch_bounds__(M2_SYM_THREADS) void blk_materialize2_sym_kernel(const uint8_t *__restrict__ in,
                                                                          const BlkEmit *__restrict__ jobs,
                                                                          const BlkUnits *__restrict__ units,
                                                                          const uint32_t *__restrict__ codes,
                                                                          uint16_t *__restrict__ sym, uint32_t njobs) {
    materialize2_body<true, M2_SYM_THREADS>(in, jobs, units, codes, sym, njobs, nullptr);
}
// the same on 1024 lanes, one unit per CU (103 KB of LDS): for a stream whose units do not fill the GPU anyway — a unit's
// own time is what counts then, and a tile of 1024 codes pays the per-tile latencies once for four times the codes
__global__ __launch_bounds__(M2_WIDE_THREADS) void blk_materialize2_sym_wide_kernel(const uint8_t *__restrict__ in,
                                                                                const BlkEmit *__restrict__ jobs,
                                                                                const BlkUnits *__restrict__ units,
                                                                                const uint32_t *__restrict__ codes,
                                                                                uint16_t *__restrict__ sym, uint32_t njobs) {
    materialize2_body<true, M2_WIDE_THREADS, true>(in, jobs, units, codes, sym, njobs, nullptr);
}

// ------------------------------------------------------------------------------------------------
#define LFX_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)

size_t blk_tabs_bytes() { return sizeof(FastTabs); }
constexpr int SCAN_SMALL = 256;      // threads of the kernels' instances for small blocks
int launch_blk_scan(hipStream_t st, const uint8_t *in, uint64_t nbytes, const BlkJob *jobs, uint32_t njobs,
                    BlkInfo *infos, BlkLanes *lanes, void *tabs, bool small_blocks) {
    if (!njobs) return 0;
    if (small_blocks)
        hipLaunchKernelGGL((blk_scan_kernel<false, SCAN_SMALL>), dim3(njobs), dim3(SCAN_SMALL), 0, st, in, nbytes, jobs, infos, lanes,
                           (FastTabs *)tabs, (uint32_t *)nullptr, (BlkLanesX *)nullptr);
    else
        hipLaunchKernelGGL((blk_scan_kernel<false, SCAN_THREADS>), dim3(njobs), dim3(SCAN_THREADS), 0, st, in, nbytes, jobs, infos, lanes,
                           (FastTabs *)tabs, (uint32_t *)nullptr, (BlkLanesX *)nullptr);
    LFX_LAUNCH_CHECK();
    return 0;
}
// the storing scan (round 6): every job's lanes write their code words to temp (BlkJob::temp_off, ::cap) and leave a
// BlkLanesX record; BlkInfo::_pad = 1 when a lane's codes did not fit (the block then needs launch_blk_emit)
int launch_blk_scan_store(hipStream_t st, const uint8_t *in, uint64_t nbytes, const BlkJob *jobs, uint32_t njobs,
                          BlkInfo *infos, BlkLanes *lanes, void *tabs, uint32_t *temp, BlkLanesX *lanesx) {
    if (!njobs) return 0;
    constexpr size_t stage_bytes = (size_t)SCAN_THREADS * EMIT_STRIDE * 4;
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)blk_scan_kernel<true, SCAN_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes);
        attr_set[dev_ & 63] = true;
    }
    hipLaunchKernelGGL((blk_scan_kernel<true, SCAN_THREADS>), dim3(njobs), dim3(SCAN_THREADS), stage_bytes, st, in, nbytes, jobs, infos, lanes,
                       (FastTabs *)tabs, temp, lanesx);
    LFX_LAUNCH_CHECK();
    return 0;
}
// ... and the move of the stored codes to their places, for the jobs with BlkEmit::placed = 1 (launch_blk_emit skips those)
int launch_blk_place(hipStream_t st, const BlkEmit *jobs, uint32_t njobs, const BlkLanes *lanes, const BlkLanesX *lanesx,
                     const uint32_t *temp, uint32_t *codes, uint32_t *flags, BlkUnits *units, uint32_t unit_target, uint32_t *job_flags,
                     uint32_t free_shift) {
    if (!njobs) return 0;
    hipLaunchKernelGGL(blk_place_kernel, dim3(njobs, 2), dim3(SCAN_THREADS), 0, st, jobs, lanes, lanesx, temp, codes, flags, units,
                       unit_target ? unit_target : 1u, free_shift < 15 ? 15u : free_shift, job_flags);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_blk_emit(hipStream_t st, const uint8_t *in, uint64_t nbytes, const BlkEmit *jobs, uint32_t njobs,
                    const BlkLanes *lanes, uint32_t *codes, uint32_t *flags, BlkUnits *units, uint32_t unit_target,
                    uint32_t *job_flags, const void *tabs, uint32_t free_shift, bool large_blocks, bool small_blocks) {
    if (!njobs) return 0;
    constexpr size_t stage_bytes = (size_t)SCAN_THREADS * EMIT_STRIDE * 4, stage_small = (size_t)SCAN_SMALL * EMIT_STRIDE * 4;
    // (a function attribute is per device: one flag per device ordinal)
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)blk_emit_kernel<false, SCAN_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes);
        (void)hipFuncSetAttribute((const void *)blk_emit_kernel<true, SCAN_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes);
        attr_set[dev_ & 63] = true;
    }
    const uint32_t ut = unit_target ? unit_target : 1u, fs = free_shift < 15 ? 15u : free_shift;
    // large blocks (a stream's own 1 MiB blocks, pieces): the LDS-ring bit source, one workgroup per CU; many small blocks
    // (the batch path, another encoder's 30 KB blocks): the register FIFO — 256 lanes a block when the scan was (small_blocks:
    // the two kernels' instances go together, BlkEmit::nlanes <= 256 then), else 1024 and two workgroups per CU
    if (small_blocks)
        hipLaunchKernelGGL((blk_emit_kernel<false, SCAN_SMALL>), dim3(njobs), dim3(SCAN_SMALL), stage_small, st, in, nbytes, jobs, lanes, codes,
                           flags, units, ut, fs, job_flags, (const FastTabs *)tabs);
    else if (large_blocks)
        hipLaunchKernelGGL((blk_emit_kernel<true, SCAN_THREADS>), dim3(njobs), dim3(SCAN_THREADS), stage_bytes, st, in, nbytes, jobs, lanes,
                           codes, flags, units, ut, fs, job_flags, (const FastTabs *)tabs);
    else
        hipLaunchKernelGGL((blk_emit_kernel<false, SCAN_THREADS>), dim3(njobs), dim3(SCAN_THREADS), stage_bytes, st, in, nbytes, jobs, lanes, codes,
                           flags, units, ut, fs, job_flags, (const FastTabs *)tabs);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_blk_materialize(hipStream_t st, const uint8_t *in, const BlkEmit *jobs, uint32_t njobs,
                           const BlkLanes *lanes, const BlkUnits *units, const uint32_t *codes, uint8_t *out,
                           uint64_t *dbg) {
    if (!njobs) return 0;
    // A stream of a few blocks fills a handful of CUs whatever the geometry: what counts then is a unit's own time, and a
    // tile of 1024 codes on sixteen wavefronts pays the per-tile latencies (barriers, owner search, pointer rounds) once
    // for four times the codes (round 4; a 64 KiB stream is ONE unit: 68 tiles of 256 codes, 0.22 of its 0.75 ms).
    if (njobs <= M2_FEW_JOBS)
        hipLaunchKernelGGL(blk_materialize2_kernel<M2_WIDE_THREADS>, dim3(njobs * MAX_UNITS), dim3(M2_WIDE_THREADS), 0, st, in, jobs, units, codes, out, njobs, dbg);
    else
        hipLaunchKernelGGL(blk_materialize2_kernel<M2_THREADS>, dim3(njobs * MAX_UNITS), dim3(M2_THREADS), 0, st, in, jobs, units, codes, out, njobs, dbg);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_blk_materialize_sym(hipStream_t st, const uint8_t *in, const BlkEmit *jobs, uint32_t njobs,
                               const BlkUnits *units, const uint32_t *codes, uint16_t *sym, bool few_units) {
    if (!njobs) return 0;
    if (few_units) {
        constexpr int lds = (int)M2Lds<true, M2_WIDE_THREADS>::BYTES;
        static bool attr_set[64] = {};
        int dev_ = 0;
        (void)hipGetDevice(&dev_);
        if (!attr_set[dev_ & 63]) {
            (void)hipFuncSetAttribute((const void *)blk_materialize2_sym_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_set[dev_ & 63] = true;
        }
        hipLaunchKernelGGL(blk_materialize2_sym_wide_kernel, dim3(njobs * MAX_FREE_UNITS), dim3(M2_WIDE_THREADS), lds, st, in, jobs, units, codes, sym, njobs);
    } else
        hipLaunchKernelGGL(blk_materialize2_sym_kernel, dim3(njobs * MAX_FREE_UNITS), dim3(M2_SYM_THREADS), 0, st, in, jobs, units, codes, sym, njobs);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_window_chain(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits, uint8_t *windows,
                        const uint8_t *init_win) {
    if (!nunits) return 0;
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)window_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set[dev_ & 63] = true;
    }
    hipLaunchKernelGGL(window_chain_kernel, dim3(1), dim3(1024), 65536, st, sym, units, nunits, windows, init_win);
    LFX_LAUNCH_CHECK();
    return 0;
}
// blocked parallel prefix of the window chain (see window_compose_kernel); scratch: nunits * 64 KiB of maps and
// ceil(nunits / WC_GROUP) * 32 KiB of group windows
size_t window_prefix_scratch_bytes(uint32_t nunits) {
    const uint32_t g = wc_group(nunits);
    return (size_t)nunits * 65536 + (size_t)((nunits + g - 1) / g) * 32768;
}
int launch_window_prefix(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits, void *scratch,
                         uint8_t *windows, const uint8_t *init_win) {
    if (!nunits) return 0;
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)window_compose_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void *)window_groups_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set[dev_ & 63] = true;
    }
    uint16_t *maps = (uint16_t *)scratch;
    uint8_t *gwin = (uint8_t *)scratch + (size_t)nunits * 65536;
    const uint32_t G = wc_group(nunits);
    const uint32_t ngroups = (nunits + G - 1) / G;
    hipLaunchKernelGGL(window_compose_kernel, dim3(ngroups), dim3(1024), 131072, st, sym, units, nunits, maps, G);
    LFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(window_groups_kernel, dim3(1), dim3(1024), 65536, st, maps, nunits, gwin, init_win, G);
    LFX_LAUNCH_CHECK();
    hipLaunchKernelGGL(window_apply_kernel, dim3(nunits), dim3(1024), 0, st, maps, gwin, windows, init_win, G);
    LFX_LAUNCH_CHECK();
    return 0;
}
// the rank's map from its symbol units (scratch as for launch_window_prefix) / from its bytes; the window in front of a rank
int launch_window_rank_map(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits, void *scratch, uint16_t *out_map) {
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)window_compose_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute((const void *)window_rank_map_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr_set[dev_ & 63] = true;
    }
    uint16_t *maps = (uint16_t *)scratch;
    const uint32_t G = wc_group(nunits);
    const uint32_t ngroups = (nunits + G - 1) / G;
    if (nunits) {
        hipLaunchKernelGGL(window_compose_kernel, dim3(ngroups), dim3(1024), 131072, st, sym, units, nunits, maps, G);
        LFX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(window_rank_map_kernel, dim3(1), dim3(1024), 131072, st, maps, nunits, out_map, G);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_bytes_to_map(hipStream_t st, const uint8_t *out, uint64_t len, uint16_t *map) {
    hipLaunchKernelGGL(bytes_to_map_kernel, dim3(128), dim3(256), 0, st, out, len, map);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_window_ranks(hipStream_t st, const uint16_t *maps, uint32_t nranks, uint8_t *win) {
    static bool attr_set[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    if (!attr_set[dev_ & 63]) {
        (void)hipFuncSetAttribute((const void *)window_ranks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        attr_set[dev_ & 63] = true;
    }
    hipLaunchKernelGGL(window_ranks_kernel, dim3(1), dim3(1024), 65536, st, maps, nranks, win);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_sym_substitute(hipStream_t st, const uint16_t *sym, const SymUnit *units, uint32_t nunits,
                          const uint8_t *windows, uint8_t *out, uint64_t max_len, const uint8_t *init_win) {
    if (!nunits) return 0;
    const uint32_t gy = (uint32_t)((max_len + 16383) / 16384);
    hipLaunchKernelGGL(sym_substitute_kernel, dim3(nunits, gy ? gy : 1), dim3(256), 0, st, sym, units, windows, out, init_win);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_find_stage2(hipStream_t st, const uint8_t *in, uint64_t nbytes, const uint64_t *cand,
                       uint32_t shard_cap, const uint32_t *count, uint32_t *work, uint32_t *final_count, uint64_t *final_list,
                       uint32_t final_cap, uint32_t n_cu, uint64_t *dbg, int exp) {
    // One-wavefront workgroups, 8 KB + 256 bytes per staged dword + the lists' prefix of LDS each: EXACTLY as many as are
    // resident at once, by the runtime's own count.  (Rounds 3-5 divided 160 KB by the arrays' bytes: 13 — but the allocation
    // is rounded up and twelve fit; the thirteenth workgroup of every CU started when the first ones LEFT, at the end, and the
    // batch it owns by its index ran behind everything else: one batch's time on top of the kernel's.)
    static int per_cu_dev[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    int &per_cu = per_cu_dev[dev_ & 63];
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)find_blocks_stage2<0>, 64, 0) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 8;
        }
        per_cu = nb;
    }
    const dim3 grid((uint32_t)per_cu * (n_cu ? n_cu : 256u));
    if (exp == 1) hipLaunchKernelGGL(find_blocks_stage2<1>, grid, dim3(64), 0, st, in, nbytes, cand, shard_cap, count, work, final_count, final_list, final_cap, dbg);
    else if (exp == 2) hipLaunchKernelGGL(find_blocks_stage2<2>, grid, dim3(64), 0, st, in, nbytes, cand, shard_cap, count, work, final_count, final_list, final_cap, dbg);
    else if (exp == 3) hipLaunchKernelGGL(find_blocks_stage2<3>, grid, dim3(64), 0, st, in, nbytes, cand, shard_cap, count, work, final_count, final_list, final_cap, dbg);
    else if (exp == 4) hipLaunchKernelGGL(find_blocks_stage2<4>, grid, dim3(64), 0, st, in, nbytes, cand, shard_cap, count, work, final_count, final_list, final_cap, dbg);
    else hipLaunchKernelGGL(find_blocks_stage2<0>, grid, dim3(64), 0, st, in, nbytes, cand, shard_cap, count, work, final_count, final_list, final_cap, dbg);
    LFX_LAUNCH_CHECK();
    return 0;
}

}  // namespace lfx
