// lfx_huff.h — DEFLATE symbol maps, length-limited Huffman code construction and dynamic block
// header generation.  ONE code path for the HIP kernel (one wavefront per block, lane-strided
// loops + workgroup barriers) and for the host unit test (lane 0 of 1, barriers are no-ops).
//
// Reference behaviour reproduced (sile/libflate v2.3.0):
//   src/deflate/symbol.rs:95-154      Symbol::{code,extra_lengh,distance}
//   src/huffman.rs:202-209            EncoderBuilder::from_frequencies
//   src/huffman.rs:261-274            calc_optimal_max_bitwidth (heap order: weight asc, depth desc)
//   src/huffman.rs:307-362            package-merge (stable sort, leaf first on ties, odd tail dropped)
//   src/huffman.rs:35-55,19-28        canonical codes in (width, symbol) order, bit-reversed
//   src/deflate/symbol.rs:343-386     DynamicHuffmanCodec::save
//   src/deflate/symbol.rs:486-540     build_bitwidth_codes (run restarts at dist[0])
#pragma once
#include "lfx_common.h"

namespace lfx {

// ---- Symbol::code / extra_lengh (symbol.rs:95-125), closed form -----------------------------
LFX_HD inline uint32_t len_symbol(uint32_t length, uint32_t &ebits, uint32_t &extra) {
    if (length <= 10) { ebits = 0; extra = 0; return 254 + length; }
    if (length == 258) { ebits = 0; extra = 0; return 285; }
    uint32_t l = length - 3;                 // 8..254
#ifdef __HIP_DEVICE_COMPILE__
    uint32_t e = 29 - __builtin_clz(l);      // floor(log2 l) - 2
#else
    uint32_t e = 29 - (uint32_t)__builtin_clz(l);
#endif
    ebits = e;
    extra = l & ((1u << e) - 1);
    return 261 + 4 * e + ((l >> e) & 3);
}
// ---- Symbol::distance (symbol.rs:126-154), closed form --------------------------------------
LFX_HD inline uint32_t dist_symbol(uint32_t distance, uint32_t &ebits, uint32_t &extra) {
    uint32_t x = distance - 1;
    if (x < 4) { ebits = 0; extra = 0; return x; }
    uint32_t e = 30 - (uint32_t)__builtin_clz(x);  // floor(log2 x) - 1
    ebits = e;
    extra = x & ((1u << e) - 1);
    return 2 * e + 2 + ((x >> e) & 1);
}
LFX_HD inline uint32_t len_extra_bits_of_symbol(uint32_t sym) {  // sym 257..285
    return (sym < 265 || sym == 285) ? 0 : (sym - 261) >> 2;
}
LFX_HD inline uint32_t dist_extra_bits_of_symbol(uint32_t sym) {  // sym 0..29
    return sym < 4 ? 0 : (sym - 2) >> 1;
}

LFX_HD inline uint32_t bitrev(uint32_t v, uint32_t w) {
    uint32_t t = 0;
    for (uint32_t i = 0; i < w; i++) { t = (t << 1) | (v & 1); v >>= 1; }
    return t;
}

// ---- scratch (lives in LDS on the device) ---------------------------------------------------
constexpr int HMAX = 288;
struct HuffScratch {
    uint64_t cur[2 * HMAX];      // current weighted list (level k-1)
    uint64_t nxt[2 * HMAX];      // next weighted list (level k)
    uint64_t pk[HMAX];           // package weights of the current level
    uint64_t sw[HMAX];           // sorted leaf weights
    uint64_t qw[HMAX];           // depth calc: internal-node queue weights
    uint16_t ssym[HMAX];         // sorted leaf symbols
    uint16_t leafpos[15][HMAX];  // position of leaf i in the level-k list
    uint16_t listlen[16];
    uint16_t acnt[16];
    uint8_t qd[HMAX];            // depth calc: internal-node depths
    alignas(4) uint8_t width[HMAX];   // result: code width per symbol (read as dwords by the canonical-code pass)
    uint16_t code[HMAX];         // result: bit-reversed code per symbol
    uint32_t freq[HMAX];         // input frequencies (with the dist[0] dummy applied)
    int32_t n;                   // used symbols
    int32_t L;                   // max bitwidth in force
    alignas(16) uint32_t key[HMAX + 4];   // rank sort: (frequency << 9) | symbol when every frequency is below 2^23
    uint32_t big;                // some frequency is 2^23 or more: the rank sort compares (frequency, symbol) pairs instead
    uint32_t wcnt[16];           // canonical codes: symbols per width
    // run-length pass (build_bitwidth_codes): run starts as a bit set over the nl + nd widths (+ a sentinel behind
    // them), entries per run, last used symbols
    uint32_t rbits[12];
    alignas(4) uint16_t rcnt[HMAX + 34];
    int32_t lit_used, dist_used;
    // header builder
    uint8_t rl_code[HMAX + 32], rl_bits[HMAX + 32], rl_extra[HMAX + 32];
    alignas(4) uint8_t rl_w[HMAX + 32];     // header assembly: bits of entry i (its code-length code + extra bits; read as dwords)
    int32_t rl_n;
    uint8_t clw[19];
    uint16_t clc[19];
    alignas(4) uint8_t lw[HMAX]; // saved literal widths while the clen code is built (read as dwords by the run-length pass)
    alignas(4) uint8_t dw[32];
    uint32_t hdrw[160];          // the header bits are assembled here (not by read-modify-write on global memory)
    uint64_t stamp[24];          // device diagnostics (LFX_STAMP)
};

#ifdef __HIP_DEVICE_COMPILE__
#define LFX_SYNC() __syncthreads()
#define LFX_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define LFX_ATOMIC_OR(p, v) atomicOr((p), (v))
#define LFX_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define LFX_STAMP(S, k) do { if (lane == 0) (S).stamp[k] = clock64(); } while (0)   /* LFX_DEBUG: where the kernel's time goes */
#define LFX_UNROLL8 _Pragma("unroll 8")   /* independent LDS reads of a counting loop: issue them in batches, not one round trip each */
#else
#define LFX_STAMP(S, k) ((void)0)
#define LFX_UNROLL8
#define LFX_SYNC() ((void)0)
#define LFX_ATOMIC_ADD(p, v) (*(p) += (v))       /* (the host runs the shared code with one lane) */
#define LFX_ATOMIC_OR(p, v) (*(p) |= (v))
#define LFX_ATOMIC_MAX(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#endif

// number of zero bytes of x (exact: no borrow runs from one byte into the next)
LFX_HD inline uint32_t zero_bytes(uint32_t x) {
    const uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // 0x80 in every zero byte
    return (uint32_t)__builtin_popcount(t);
}

// Code widths for S.freq[0..nsym) with limit `limit` → S.width[], S.code[] (bit-reversed).
// All lanes of the (single-wave) workgroup must call it.
LFX_HD inline void huff_build(HuffScratch &S, int nsym, int limit, int lane, int nlanes, int stamp0 = -1) {
    (void)stamp0;
    // 1. used symbols, stable order by weight (huffman.rs:309-315)
    //    rank = #{used t : f_t < f or (f_t == f and t < s)}.  Round 4: with every frequency below 2^23 (any block of
    //    less than 8 Mi symbols) the pair (f, s) is ONE 32-bit key (f << 9) | s, and the rank is a count of smaller
    //    keys — a compare and an add per element instead of seven instructions; the unused symbols (keys below 512,
    //    smaller than every used key) are counted too and taken off again.  Keys behind the alphabet are all-ones.
    for (int i = lane; i < HMAX; i += nlanes) { S.width[i] = 0; S.code[i] = 0; }
    for (int i = lane; i < 16; i += nlanes) S.wcnt[i] = 0;
    if (lane == 0) { S.n = 0; S.big = 0; }
    LFX_SYNC();
    const int nsym4 = (nsym + 3) & ~3;
    for (int s = lane; s < nsym4; s += nlanes) {
        const uint32_t f = s < nsym ? S.freq[s] : 0u;
        if (f) LFX_ATOMIC_ADD(&S.n, 1);
        if (f >> 23) LFX_ATOMIC_OR(&S.big, 1u);
        S.key[s] = s < nsym ? (f << 9) | (uint32_t)s : 0xFFFFFFFFu;
    }
    LFX_SYNC();
    const int n = S.n;
    const bool big = S.big != 0;
    for (int s = lane; s < nsym; s += nlanes) {
        uint32_t f = S.freq[s];
        if (f == 0) continue;
        int r = 0;
        if (!big) {
            const uint32_t key = (f << 9) | (uint32_t)s;
            uint32_t cnt = 0;
            for (int t = 0; t < nsym4; t += 4) {
                const uint32_t k0 = S.key[t], k1 = S.key[t + 1], k2 = S.key[t + 2], k3 = S.key[t + 3];
                cnt += (uint32_t)(k0 < key) + (uint32_t)(k1 < key) + (uint32_t)(k2 < key) + (uint32_t)(k3 < key);
            }
            r = (int)cnt - (nsym - n);
        } else {
            LFX_UNROLL8
            for (int t = 0; t < nsym; t++) {
                uint32_t g = S.freq[t];
                r += (g != 0) & ((g < f) | ((g == f) & (t < s)));
            }
        }
        S.sw[r] = f;
        S.ssym[r] = (uint16_t)s;
    }
    LFX_SYNC();
    if (stamp0 >= 0) LFX_STAMP(S, stamp0 + 0);       // rank sort done
    if (n == 0) return;                       // every width 0
    if (n == 1) {                             // package() leaves a 1-element list untouched → width 1
        if (lane == 0) { S.width[S.ssym[0]] = 1; S.code[S.ssym[0]] = 0; }
        LFX_SYNC();
        return;
    }
    // 2. calc_optimal_max_bitwidth (huffman.rs:261-274): pop order = (weight asc, depth desc);
    //    leaves have depth 0, so on equal weight an internal node goes first, and among internal
    //    nodes of equal weight (a contiguous run at the queue head, weights are created in
    //    non-decreasing order) the deepest goes first.
    //    Serial on lane 0.  Round 4: no pick waits for an LDS round trip — the next three leaf weights and the first
    //    three queue entries (weight << 8 | depth, one 64-bit word) are held in registers, refilled two picks ahead of
    //    their use, and a new node that lands inside the window goes into its register as well.  Only a run of equal
    //    weights at the queue head walks the queue in LDS (round 3: four dependent reads per merge, 62 K cycles).
    if (lane == 0) {
        uint64_t *Q = S.qw;
        int li = 0, qh = 0, qt = 0;
        uint32_t depth = 0;
        uint64_t l0 = S.sw[0], l1 = S.sw[1], l2 = n > 2 ? S.sw[2] : 0;     // (n >= 2 here)
        uint64_t q0 = 0, q1 = 0, q2 = 0;
        for (int m = 0; m < n - 1; m++) {
            uint64_t w2[2];
            uint32_t d2[2];
            for (int k = 0; k < 2; k++) {
                const bool takeq = qh < qt && (li >= n || (q0 >> 8) <= l0);
                if (takeq) {
                    const uint64_t w = q0 >> 8;
                    uint32_t d = (uint32_t)(q0 & 0xFF);
                    if (qh + 1 < qt && (q1 >> 8) == w) {
                        int best = qh;
                        uint32_t bd = d;
                        for (int g = qh + 1; g < qt; g++) {
                            const uint64_t e = Q[g];
                            if ((e >> 8) != w) break;
                            if ((uint32_t)(e & 0xFF) > bd) { bd = (uint32_t)(e & 0xFF); best = g; }
                        }
                        if (best != qh) {   // weights in the run are equal: only depths move (the head's goes to the slot taken)
                            const uint64_t moved = (w << 8) | d;
                            Q[best] = moved;
                            if (best == qh + 1) q1 = moved;
                            if (best == qh + 2) q2 = moved;
                            d = bd;
                        }
                    }
                    w2[k] = w;
                    d2[k] = d;
                    qh++;
                    q0 = q1; q1 = q2;
                    if (qh + 2 < qt) q2 = Q[qh + 2];
                } else {
                    w2[k] = l0;
                    d2[k] = 0;
                    li++;
                    l0 = l1; l1 = l2;
                    if (li + 2 < n) l2 = S.sw[li + 2];
                }
            }
            uint32_t d = 1 + (d2[0] > d2[1] ? d2[0] : d2[1]);
            if (d > 255) d = 255;       // (only compared with a limit of at most 15)
            const uint64_t e = ((w2[0] + w2[1]) << 8) | d;
            Q[qt] = e;
            if (qt == qh) q0 = e; else if (qt == qh + 1) q1 = e; else if (qt == qh + 2) q2 = e;
            qt++;
            depth = d;  // the last node created is the root
        }
        const int opt = depth > 1 ? (int)depth : 1;
        S.L = limit < opt ? limit : opt;
    }
    LFX_SYNC();
    if (stamp0 >= 0) LFX_STAMP(S, stamp0 + 1);       // depth done
    const int L = S.L;
    // 3. package-merge forward (huffman.rs:317-318): level 0 = source.
    //    Round 4: the two weighted lists swap roles instead of being copied (one barrier less per level), and a lane's two
    //    rank searches — its leaf among the packages, its package among the leaves — run in lockstep, branch-free, so that
    //    their LDS reads overlap (they were eighteen dependent round trips per level).
    uint64_t *cur = S.cur, *nxt = S.nxt;
    for (int i = lane; i < n; i += nlanes) { cur[i] = S.sw[i]; S.leafpos[0][i] = (uint16_t)i; }
    if (lane == 0) S.listlen[0] = (uint16_t)n;
    LFX_SYNC();
    for (int k = 1; k < L; k++) {
        const int len = S.listlen[k - 1];
        const int np = len / 2;  // package(): pairs (2p, 2p+1), odd tail dropped (n >= 2 here)
        for (int p = lane; p < np; p += nlanes) S.pk[p] = cur[2 * p] + cur[2 * p + 1];
        if (lane == 0) S.listlen[k] = (uint16_t)(np + n);
        LFX_SYNC();
        // merge(packages, source): a package goes first only if strictly lighter (huffman.rs:342-346)
        //   leaf i lands at i + #{p : P[p] < S[i]}, package p at p + #{i : S[i] <= P[p]}
        const int top = n > np ? n : np;                                   // (np >= 1)
        const int step0 = 1 << (31 - __builtin_clz((unsigned)top));      // counts up to 2 * step0 - 1 >= top
        for (int i = lane; i < top; i += nlanes) {
            const bool isleaf = i < n, ispk = i < np;
            const uint64_t wl = S.sw[isleaf ? i : 0], wp = S.pk[ispk ? i : 0];
            int cl = 0, cp = 0;
            for (int step = step0; step; step >>= 1) {
                const int il = cl + step, ip = cp + step;
                const uint64_t xl = S.pk[(il <= np ? il : np) - 1];       // (np >= 1, n >= 2: the clamped indices are valid)
                const uint64_t xp = S.sw[(ip <= n ? ip : n) - 1];
                cl = (il <= np && xl < wl) ? il : cl;
                cp = (ip <= n && xp <= wp) ? ip : cp;
            }
            if (isleaf) { nxt[i + cl] = wl; S.leafpos[k][i] = (uint16_t)(i + cl); }
            if (ispk) nxt[i + cp] = wp;
        }
        LFX_SYNC();
        uint64_t *t = cur; cur = nxt; nxt = t;
    }
    if (stamp0 >= 0) LFX_STAMP(S, stamp0 + 2);       // forward passes done
    // 4. backward: the final package() keeps the first 2*floor(len/2) items of the last list; a
    //    selected package at level k expands to two items of level k-1 (a prefix, merge is stable).
    //    a_k = #{i : leafpos[k][i] < m}: the positions grow strictly with i, so a_k is found by the ONE leaf that is
    //    below m while its successor is not — every lane looks at its own leaf (round 4; a binary search on lane 0 was
    //    nine dependent LDS round trips per level).
    for (int k = lane; k < L; k += nlanes) S.acnt[k] = 0;
    LFX_SYNC();
    {
        int m = 2 * (S.listlen[L - 1] / 2);
        for (int k = L - 1; k >= 0; k--) {
            for (int i = lane; i < n; i += nlanes)
                if (S.leafpos[k][i] < m && (i + 1 == n || S.leafpos[k][i + 1] >= m)) S.acnt[k] = (uint16_t)(i + 1);
            LFX_SYNC();
            m = 2 * (m - (int)S.acnt[k]);
        }
    }
    for (int i = lane; i < n; i += nlanes) {
        int w = 0;
        LFX_UNROLL8
        for (int k = 0; k < L; k++) w += i < S.acnt[k];
        S.width[S.ssym[i]] = (uint8_t)w;
        LFX_ATOMIC_ADD(&S.wcnt[w], 1u);
    }
    LFX_SYNC();
    if (stamp0 >= 0) LFX_STAMP(S, stamp0 + 3);       // widths done
    // 5. canonical codes (huffman.rs:35-55): symbols in (width, symbol) order.  The code of symbol s of width w =
    //    (codes of the shorter widths, each scaled to w) + (symbols of width w in front of s): the first from the
    //    per-width counts, the second by counting equal bytes in the width array a dword at a time (round 4: the
    //    loop over all symbols was ~4000 instructions per lane).
    {
        const uint32_t *w32 = (const uint32_t *)S.width;
        for (int s = lane; s < nsym; s += nlanes) {
            const uint32_t w = S.width[s];
            if (w == 0) continue;
            uint32_t c = 0;
            for (uint32_t v = 1; v < w; v++) c += S.wcnt[v] << (w - v);
            const uint32_t pat = w * 0x01010101u;
            const int full = s >> 2;
            LFX_UNROLL8
            for (int j = 0; j < full; j++) c += zero_bytes(w32[j] ^ pat);
            if (s & 3) c += zero_bytes((w32[full] ^ pat) | (0xFFFFFFFFu << (8 * (s & 3))));
            S.code[s] = (uint16_t)bitrev(c & 0xFFFF, w);
        }
    }
    LFX_SYNC();
}

// bit writer into a uint32 array (LSB-first, like BitWriter bit.rs:25-49)
struct HdrWriter {
    uint32_t *w;
    uint32_t nbits;
    LFX_HD void put(uint32_t width, uint32_t bits) {
        uint32_t word = nbits >> 5, sh = nbits & 31;
        w[word] |= bits << sh;
        if (sh + width > 32) w[word + 1] |= bits >> (32 - sh);
        nbits += width;
    }
};

constexpr int CLEN_ORDER_N = 19;
LFX_HD inline int clen_order(int k) {
    // BITWIDTH_CODE_ORDER symbol.rs:16-18 — 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 — five bits per entry in two
    // constants: a local array is a table in global memory on the device, one dependent load per call (tools/isa_scan.py)
    const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                        10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    return (int)((k < 12 ? lo >> (5 * k) : hi >> (5 * (k - 12))) & 31);
}

// Whole per-block Huffman stage.  hist[0..286) literal/length counts (EOB included),
// hist[288..318) distance counts.  All lanes of the single-wave workgroup call it.
LFX_HD inline void huff_block_build(const uint32_t *hist, uint32_t type, BlockCodes *out,
                                    HuffScratch &S, int lane, int nlanes) {
    if (type == BT_FIXED) {
        // FixedHuffmanCodec::build symbol.rs:258-280
        for (int s = lane; s < 288; s += nlanes) {
            uint32_t w, c;
            if (s < 144) { w = 8; c = 0x30 + s; }
            else if (s < 256) { w = 9; c = 0x190 + (s - 144); }
            else if (s < 280) { w = 7; c = s - 256; }
            else { w = 8; c = 0xC0 + (s - 280); }
            out->lit[s] = bitrev(c, w) | (w << 16);
        }
        for (int s = lane; s < 32; s += nlanes) out->dist[s] = s < 30 ? (bitrev((uint32_t)s, 5) | (5u << 16)) : 0;
        LFX_SYNC();
        if (lane == 0) {
            uint64_t bits = 3;
            for (int s = 0; s < 286; s++) {
                uint32_t w = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                bits += (uint64_t)hist[s] * (w + (s > 256 ? len_extra_bits_of_symbol((uint32_t)s) : 0));
            }
            for (int d = 0; d < 30; d++) bits += (uint64_t)hist[288 + d] * (5 + dist_extra_bits_of_symbol((uint32_t)d));
            out->body_bits = bits;
            out->hdr_bits = 0;
        }
        LFX_SYNC();
        return;
    }
    // DynamicHuffmanCodec::build symbol.rs:321-342 — literal/length alphabet
    LFX_STAMP(S, 0);
    for (int s = lane; s < HMAX; s += nlanes) S.freq[s] = s < 286 ? hist[s] : 0;
    for (int i = lane; i < 12; i += nlanes) S.rbits[i] = 0;
    if (lane == 0) { S.lit_used = 0; S.dist_used = 0; S.rl_n = 0; }
    LFX_SYNC();
    huff_build(S, 286, 15, lane, nlanes, 1);
    LFX_STAMP(S, 5);
    for (int s = lane; s < 288; s += nlanes) {
        out->lit[s] = (uint32_t)S.code[s] | ((uint32_t)S.width[s] << 16);
        S.lw[s] = S.width[s];
        if (s < 286 && S.width[s]) LFX_ATOMIC_MAX(&S.lit_used, s);      // used_max_symbol().unwrap_or(0), huffman.rs:246-253
    }
    LFX_SYNC();
    // distance alphabet, with the dist[0] = 1 dummy when the block has no pointer (symbol.rs:332-337)
    for (int d = lane; d < HMAX; d += nlanes) S.freq[d] = d < 30 ? hist[288 + d] : 0;   // (lanes in parallel: on lane 0 these were sixty dependent global loads)
    LFX_SYNC();
    if (lane == 0) {
        uint32_t any = 0;
        LFX_UNROLL8
        for (int d = 0; d < 30; d++) any |= S.freq[d];
        if (!any) S.freq[0] = 1;
    }
    LFX_SYNC();
    huff_build(S, 30, 15, lane, nlanes);
    LFX_STAMP(S, 6);
    for (int s = lane; s < 32; s += nlanes) {
        out->dist[s] = s < 30 ? ((uint32_t)S.code[s] | ((uint32_t)S.width[s] << 16)) : 0;
        S.dw[s] = s < 30 ? S.width[s] : 0;
        if (s < 30 && S.width[s]) LFX_ATOMIC_MAX(&S.dist_used, s);
    }
    LFX_SYNC();
    // DynamicHuffmanCodec::save symbol.rs:343-386; build_bitwidth_codes symbol.rs:486-540.
    // Round 4: by all lanes (on lane 0 the pass was ~300 dependent steps, a sixth of the kernel).  The nl literal widths
    // and the nd distance widths are one sequence of nl + nd positions in which a run never crosses into the second table
    // (symbol.rs:501-507).  (A) every position decides whether a run starts at it and sets its bit; (B) a run's first
    // position finds the next start (its length), and from value and length the number of entries the reference's loops
    // emit for it; (C) the entries of the runs in front of it give the run its place, and it writes its entries.
    {
        const int nl = S.lit_used + 1 < 257 ? 257 : S.lit_used + 1;
        const int nd = S.dist_used + 1 < 1 ? 1 : S.dist_used + 1;
        const int tot = nl + nd;
        auto width_at = [&](int i) -> uint32_t { return i < nl ? S.lw[i] : S.dw[i - nl]; };
        for (int i = lane; i <= tot; i += nlanes) {
            const bool st = i == tot || i == 0 || i == nl || width_at(i - 1) != width_at(i);    // (i == tot: the sentinel)
            if (st) LFX_ATOMIC_OR(&S.rbits[i >> 5], 1u << (i & 31));
        }
        LFX_SYNC();
        auto run_len = [&](int i) -> int {                   // i: a run's first position
            int p = i + 1, wd = p >> 5;
            uint32_t x = S.rbits[wd] >> (p & 31);
            if (x) return p + __builtin_ctz(x) - i;
            for (;;) {                                       // (ends at the sentinel's word at the latest)
                x = S.rbits[++wd];
                if (x) return wd * 32 + __builtin_ctz(x) - i;
            }
        };
        for (int i = lane; i < tot; i += nlanes) {
            uint32_t e = 0;
            if ((S.rbits[i >> 5] >> (i & 31)) & 1) {
                const uint32_t c = (uint32_t)run_len(i);
                if (width_at(i) == 0) {
                    // while c >= 11 { 18: min(c, 138) }; then c >= 3: one 17; else c plain zeros
                    const uint32_t q = c / 138, r = c % 138;
                    e = q + (r >= 3 ? 1u : r);
                } else {
                    // the value once; then while c >= 3 { 16: min(c, 6) }; the rest as plain values
                    const uint32_t c1 = c - 1, q = c1 / 6, r = c1 % 6;
                    e = 1 + q + (r >= 3 ? 1u : r);
                }
            }
            S.rcnt[i] = (uint16_t)e;
        }
        LFX_SYNC();
        const uint32_t *r32 = (const uint32_t *)S.rcnt;
        for (int i = lane; i < tot; i += nlanes) {
            const uint32_t e = S.rcnt[i];
            if (e == 0) continue;
            uint32_t at = 0;
            LFX_UNROLL8
            for (int j = 0; j < (i >> 1); j++) { const uint32_t x = r32[j]; at += (x & 0xFFFFu) + (x >> 16); }
            if (i & 1) at += S.rcnt[i - 1];
            uint32_t c = (uint32_t)run_len(i);
            const uint32_t v = width_at(i);
            if (i + (int)c == tot) S.rl_n = (int32_t)(at + e);
            uint32_t rn = at;
            auto put = [&](uint32_t code, uint32_t bits, uint32_t extra) {
                S.rl_code[rn] = (uint8_t)code; S.rl_bits[rn] = (uint8_t)bits; S.rl_extra[rn] = (uint8_t)extra; rn++;
            };
            if (v == 0) {
                while (c >= 11) { const uint32_t k = c < 138 ? c : 138; put(18, 7, k - 11); c -= k; }
                if (c >= 3) { put(17, 3, c - 3); c = 0; }
                for (; c > 0; c--) put(0, 0, 0);
            } else {
                put(v, 0, 0);
                c -= 1;
                while (c >= 3) { const uint32_t k = c < 6 ? c : 6; put(16, 2, k - 3); c -= k; }
                for (; c > 0; c--) put(v, 0, 0);
            }
        }
        if (lane == 0) {
            // stash nl/nd for the emit step
            S.listlen[15] = (uint16_t)nl;
            S.acnt[15] = (uint16_t)nd;
        }
    }
    LFX_STAMP(S, 7);                                   // run-length pass done
    for (int s = lane; s < HMAX; s += nlanes) S.freq[s] = 0;
    LFX_SYNC();
    // code-length-symbol counts and the header's bit assembly are spread over the lanes (round 3: on lane 0 they were
    // ~500 dependent LDS read-modify-writes, a third of the kernel's time)
    for (int i = lane; i < S.rl_n; i += nlanes) LFX_ATOMIC_ADD(&S.freq[S.rl_code[i]], 1u);
    LFX_SYNC();
    // keep the code-length counts: huff_build leaves S.freq intact
    huff_build(S, 19, 7, lane, nlanes);
    LFX_STAMP(S, 8);
    for (int i = lane; i < 19; i += nlanes) { S.clw[i] = S.width[i]; S.clc[i] = S.code[i]; }
    for (int i = lane; i < 160; i += nlanes) S.hdrw[i] = 0;
    LFX_SYNC();
    for (int i = lane; i < S.rl_n; i += nlanes) S.rl_w[i] = (uint8_t)(S.clw[S.rl_code[i]] + S.rl_bits[i]);
    if (lane == 0) {
        int nl = S.listlen[15], nd = S.acnt[15];
        int bcc = 0;  // symbol.rs:357-364
        for (int k = 18; k >= 0; k--) {
            int i = clen_order(k);
            if (S.freq[i] != 0 && S.clw[i] > 0) { bcc = k + 1; break; }
        }
        if (bcc < 4) bcc = 4;
        HdrWriter hw{S.hdrw, 0};
        hw.put(5, (uint32_t)(nl - 257));
        hw.put(5, (uint32_t)(nd - 1));
        hw.put(4, (uint32_t)(bcc - 4));
        for (int k = 0; k < bcc; k++) {
            int i = clen_order(k);
            hw.put(3, S.freq[i] == 0 ? 0u : (uint32_t)S.clw[i]);
        }
        S.acnt[14] = (uint16_t)hw.nbits;   // first bit of the code-length sequence
    }
    LFX_SYNC();
    {
        // entry i = its code-length code, then its extra bits (LSB first, bit.rs:25-49): at most 7 + 7 bits at the
        // offset Σ widths of the entries in front of it
        const uint32_t base = S.acnt[14];
        const int rn = S.rl_n;
        for (int i = lane; i < rn; i += nlanes) {
            uint32_t at = base;
            {   // Σ widths of the entries in front: four at a time (each at most 14 bits wide)
                const uint32_t *w32 = (const uint32_t *)S.rl_w;
                uint32_t acc = 0;
                LFX_UNROLL8
                for (int j = 0; j < (i >> 2); j++) { const uint32_t x = w32[j]; acc += (x & 0x00FF00FFu) + ((x >> 8) & 0x00FF00FFu); }
                at += (acc & 0xFFFFu) + (acc >> 16);
                for (int j = i & ~3; j < i; j++) at += S.rl_w[j];
            }
            const uint32_t cw = S.clw[S.rl_code[i]];
            const uint32_t v = (uint32_t)S.clc[S.rl_code[i]] | ((uint32_t)S.rl_extra[i] << cw);
            const uint32_t word = at >> 5, sh = at & 31, wd = S.rl_w[i];
            if (wd) {
                LFX_ATOMIC_OR(&S.hdrw[word], v << sh);
                if (sh + wd > 32) LFX_ATOMIC_OR(&S.hdrw[word + 1], v >> (32 - sh));
            }
            if (i == rn - 1) { out->hdr_bits = at + wd; S.L = (int32_t)(at + wd); }   // (S.L: free by now; read by the sum below.  rn is never 0: at least 258 widths)
        }
    }
    LFX_SYNC();
    LFX_STAMP(S, 9);                                   // header bits assembled
    // header words out; body size = 3 + header + Σ count · (width + extra bits), partial sums per lane
    for (int i = lane; i < 160; i += nlanes) out->hdr[i] = S.hdrw[i];
    {
        uint64_t part = 0;
        for (int s = lane; s < 286; s += nlanes)
            part += (uint64_t)hist[s] * (S.lw[s] + (s > 256 ? len_extra_bits_of_symbol((uint32_t)s) : 0));
        for (int d = lane; d < 30; d += nlanes)
            part += (uint64_t)hist[288 + d] * (S.dw[d] + dist_extra_bits_of_symbol((uint32_t)d));
        S.cur[lane] = part;           // (the weighted lists are no longer needed)
    }
    LFX_SYNC();
    if (lane == 0) {
        uint64_t bits = 3 + (uint64_t)(uint32_t)S.L;
        for (int l = 0; l < nlanes; l++) bits += S.cur[l];
        out->body_bits = bits;
    }
    LFX_STAMP(S, 10);
    LFX_SYNC();
}

}  // namespace lfx
