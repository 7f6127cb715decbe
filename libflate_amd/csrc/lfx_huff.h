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
    uint8_t width[HMAX];         // result: code width per symbol
    uint16_t code[HMAX];         // result: bit-reversed code per symbol
    uint32_t freq[HMAX];         // input frequencies (with the dist[0] dummy applied)
    int32_t n;                   // used symbols
    int32_t L;                   // max bitwidth in force
    // header builder
    uint8_t rl_code[HMAX + 32], rl_bits[HMAX + 32], rl_extra[HMAX + 32];
    uint8_t rl_w[HMAX + 32];     // header assembly: bits of entry i (its code-length code + extra bits)
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
#define LFX_STAMP(S, k) do { if (lane == 0) (S).stamp[k] = clock64(); } while (0)   /* LFX_DEBUG: where the kernel's time goes */
#define LFX_UNROLL8 _Pragma("unroll 8")   /* independent LDS reads of a counting loop: issue them in batches, not one round trip each */
#else
#define LFX_STAMP(S, k) ((void)0)
#define LFX_UNROLL8
#define LFX_SYNC() ((void)0)
#define LFX_ATOMIC_ADD(p, v) (*(p) += (v))       /* (the host runs the shared code with one lane) */
#define LFX_ATOMIC_OR(p, v) (*(p) |= (v))
#endif

// Code widths for S.freq[0..nsym) with limit `limit` → S.width[], S.code[] (bit-reversed).
// All lanes of the (single-wave) workgroup must call it.
LFX_HD inline void huff_build(HuffScratch &S, int nsym, int limit, int lane, int nlanes, int stamp0 = -1) {
    (void)stamp0;
    // 1. used symbols, stable order by weight (huffman.rs:309-315)
    for (int i = lane; i < HMAX; i += nlanes) { S.width[i] = 0; S.code[i] = 0; }
    LFX_SYNC();
    if (lane == 0) {
        int n = 0;
        LFX_UNROLL8
        for (int s = 0; s < nsym; s++) n += S.freq[s] > 0;
        S.n = n;
    }
    LFX_SYNC();
    const int n = S.n;
    for (int s = lane; s < nsym; s += nlanes) {
        uint32_t f = S.freq[s];
        if (f == 0) continue;
        int r = 0;  // rank = #{used t : f_t < f or (f_t == f and t < s)}
        LFX_UNROLL8
        for (int t = 0; t < nsym; t++) {
            uint32_t g = S.freq[t];
            r += (g != 0) & ((g < f) | ((g == f) & (t < s)));
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
    //    Serial on lane 0, one dependent LDS round trip after the other: the two queue heads (next leaf weight, weight
    //    at the head of the internal-node queue) are kept in registers and reloaded only when consumed (round 3: 7n → 4n
    //    dependent reads).
    if (lane == 0) {
        int li = 0, qh = 0, qt = 0, depth = 0;
        uint64_t lw = S.sw[0];     // weight of the leaf at li (while li < n)
        uint64_t qw0 = 0;          // weight at the queue head (while qh < qt)
        for (int m = 0; m < n - 1; m++) {
            uint64_t w2[2];
            int d2[2];
            for (int k = 0; k < 2; k++) {
                const bool takeq = qh < qt && (li >= n || qw0 <= lw);
                if (takeq) {
                    const uint64_t w = qw0;
                    int best = qh;
                    for (int g = qh + 1; g < qt && S.qw[g] == w; g++)
                        if (S.qd[g] > S.qd[best]) best = g;
                    d2[k] = S.qd[best];
                    if (best != qh) S.qd[best] = S.qd[qh];  // weights in the run are equal: only depths move
                    w2[k] = w;
                    qh++;
                    if (qh < qt) qw0 = S.qw[qh];
                } else {
                    w2[k] = lw;
                    d2[k] = 0;
                    li++;
                    if (li < n) lw = S.sw[li];
                }
            }
            int d = 1 + (d2[0] > d2[1] ? d2[0] : d2[1]);
            const uint64_t ws = w2[0] + w2[1];
            S.qw[qt] = ws;
            S.qd[qt] = (uint8_t)(d > 255 ? 255 : d);
            if (qh == qt) qw0 = ws;   // the queue was empty: the new node is its head
            qt++;
            depth = d;  // the last node created is the root
        }
        int opt = depth > 1 ? depth : 1;
        S.L = limit < opt ? limit : opt;
    }
    LFX_SYNC();
    if (stamp0 >= 0) LFX_STAMP(S, stamp0 + 1);       // depth done
    const int L = S.L;
    // 3. package-merge forward (huffman.rs:317-318): level 0 = source
    for (int i = lane; i < n; i += nlanes) { S.cur[i] = S.sw[i]; S.leafpos[0][i] = (uint16_t)i; }
    if (lane == 0) S.listlen[0] = (uint16_t)n;
    LFX_SYNC();
    for (int k = 1; k < L; k++) {
        const int len = S.listlen[k - 1];
        const int np = len / 2;  // package(): pairs (2p, 2p+1), odd tail dropped (n >= 2 here)
        for (int p = lane; p < np; p += nlanes) S.pk[p] = S.cur[2 * p] + S.cur[2 * p + 1];
        LFX_SYNC();
        // merge(packages, source): a package goes first only if strictly lighter (huffman.rs:342-346)
        for (int i = lane; i < n; i += nlanes) {
            uint64_t w = S.sw[i];
            int lo = 0, hi = np;  // #{p : P[p] < w}
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (S.pk[mid] < w) lo = mid + 1; else hi = mid;
            }
            S.nxt[i + lo] = w;
            S.leafpos[k][i] = (uint16_t)(i + lo);
        }
        for (int p = lane; p < np; p += nlanes) {
            uint64_t w = S.pk[p];
            int lo = 0, hi = n;  // #{i : S[i] <= w}
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (S.sw[mid] <= w) lo = mid + 1; else hi = mid;
            }
            S.nxt[p + lo] = w;
        }
        LFX_SYNC();
        for (int i = lane; i < np + n; i += nlanes) S.cur[i] = S.nxt[i];
        if (lane == 0) S.listlen[k] = (uint16_t)(np + n);
        LFX_SYNC();
    }
    if (stamp0 >= 0) LFX_STAMP(S, stamp0 + 2);       // forward passes done
    // 4. backward: the final package() keeps the first 2*floor(len/2) items of the last list; a
    //    selected package at level k expands to two items of level k-1 (a prefix, merge is stable)
    if (lane == 0) {
        int m = 2 * (S.listlen[L - 1] / 2);
        for (int k = L - 1; k >= 0; k--) {
            int lo = 0, hi = n;  // a = #{i : leafpos[k][i] < m}
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (S.leafpos[k][mid] < m) lo = mid + 1; else hi = mid;
            }
            S.acnt[k] = (uint16_t)lo;
            m = 2 * (m - lo);
        }
    }
    LFX_SYNC();
    for (int i = lane; i < n; i += nlanes) {
        int w = 0;
        LFX_UNROLL8
        for (int k = 0; k < L; k++) w += i < S.acnt[k];
        S.width[S.ssym[i]] = (uint8_t)w;
    }
    LFX_SYNC();
    if (stamp0 >= 0) LFX_STAMP(S, stamp0 + 3);       // widths done
    // 5. canonical codes (huffman.rs:35-55): symbols in (width, symbol) order
    for (int s = lane; s < nsym; s += nlanes) {
        int w = S.width[s];
        if (w == 0) continue;
        // code = (number of codes before me, each scaled to my width)
        uint32_t c = 0;
        LFX_UNROLL8
        for (int t = 0; t < nsym; t++) {
            int wt = S.width[t];
            if (wt == 0) continue;
            if (wt < w) c += 1u << (w - wt);
            else if (wt == w && t < s) c += 1;
        }
        S.code[s] = (uint16_t)bitrev(c & 0xFFFF, (uint32_t)w);
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
    // BITWIDTH_CODE_ORDER symbol.rs:16-18
    const uint8_t o[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    return o[k];
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
    LFX_SYNC();
    huff_build(S, 286, 15, lane, nlanes, 1);
    LFX_STAMP(S, 5);
    for (int s = lane; s < 288; s += nlanes) {
        out->lit[s] = (uint32_t)S.code[s] | ((uint32_t)S.width[s] << 16);
        S.lw[s] = S.width[s];
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
    }
    LFX_SYNC();
    // DynamicHuffmanCodec::save symbol.rs:343-386 (serial: ~300 short steps)
    if (lane == 0) {
        int lit_used = 0, dist_used = 0;  // used_max_symbol().unwrap_or(0)
        for (int s = 285; s >= 0; s--) if (S.lw[s]) { lit_used = s; break; }
        for (int s = 29; s >= 0; s--) if (S.dw[s]) { dist_used = s; break; }
        int nl = lit_used + 1 < 257 ? 257 : lit_used + 1;
        int nd = dist_used + 1 < 1 ? 1 : dist_used + 1;
        // build_bitwidth_codes symbol.rs:486-540
        int rn = 0;
        for (int t = 0; t < 2; t++) {
            // (the widths are read a dword at a time: every read is a dependent LDS round trip on this one lane)
            const uint32_t *w32 = (const uint32_t *)(t ? S.dw : S.lw);
            int size = t ? nd : nl;
            int i = 0;
            int cidx = -1;
            uint32_t cw = 0;
            auto wat = [&](int q) -> uint8_t {
                if ((q >> 2) != cidx) { cidx = q >> 2; cw = w32[cidx]; }
                return (uint8_t)(cw >> (8 * (q & 3)));
            };
            while (i < size) {
                uint8_t v = wat(i);
                int c = 1;
                while (i + c < size && wat(i + c) == v) c++;  // a run never crosses into the next table
                i += c;
                if (v == 0) {
                    while (c >= 11) {
                        int k = c < 138 ? c : 138;
                        S.rl_code[rn] = 18; S.rl_bits[rn] = 7; S.rl_extra[rn] = (uint8_t)(k - 11); rn++;
                        c -= k;
                    }
                    if (c >= 3) { S.rl_code[rn] = 17; S.rl_bits[rn] = 3; S.rl_extra[rn] = (uint8_t)(c - 3); rn++; c = 0; }
                    for (; c > 0; c--) { S.rl_code[rn] = 0; S.rl_bits[rn] = 0; S.rl_extra[rn] = 0; rn++; }
                } else {
                    S.rl_code[rn] = v; S.rl_bits[rn] = 0; S.rl_extra[rn] = 0; rn++;
                    c -= 1;
                    while (c >= 3) {
                        int k = c < 6 ? c : 6;
                        S.rl_code[rn] = 16; S.rl_bits[rn] = 2; S.rl_extra[rn] = (uint8_t)(k - 3); rn++;
                        c -= k;
                    }
                    for (; c > 0; c--) { S.rl_code[rn] = v; S.rl_bits[rn] = 0; S.rl_extra[rn] = 0; rn++; }
                }
            }
        }
        S.rl_n = rn;
        // stash nl/nd for the emit step
        S.listlen[15] = (uint16_t)nl;
        S.acnt[15] = (uint16_t)nd;
    }
    LFX_STAMP(S, 7);                                   // run-length pass done (lane 0)
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
            for (int j = 0; j < i; j++) at += S.rl_w[j];
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
