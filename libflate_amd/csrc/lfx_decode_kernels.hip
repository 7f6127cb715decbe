// lfx_decode_kernels.hip — hand-written gfx950 kernels of the inflate hot path.
//
//   container_kernel : gzip / zlib header parse per stream (src/gzip.rs:390-446, src/zlib.rs:221-266)
//   find_blocks_*    : speculative search for dynamic-block headers at every bit offset of ONE stream
//                      (the reference decodes strictly serially, src/deflate/decode.rs:136-164; block
//                      start bits are not recorded in the format, so they are rediscovered and then
//                      validated by chaining end bits from the known first block)
//   inflate_kernel   : one wavefront per job (a whole stream, or one block of a stream): lane 0 walks
//                      the Huffman symbols (symbol.rs:193-244 / huffman.rs:157-179 semantics, including
//                      the deferred-error behaviour of BitReader, bit.rs:84-141) into a 64-entry queue
//                      in LDS, the 64 lanes then materialise literals and back-references
//                      (Lz77Decoder::decode, libflate_lz77/src/lib.rs:164-194) together.
//   stream_checksum  : CRC-32 / Adler-32 of each stream's output for trailer verification.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lfx_common.h"
#include "lfx_decode.h"
#include "lfx_container.h"

namespace lfx {

__device__ __forceinline__ uint32_t ld1(gptr_u8 p) { return *p; }

// ------------------------------------------------------------------------------------------------
// container headers.  One lane per stream.
__global__ void container_kernel(int format, uint32_t count, const uint8_t *__restrict__ in,
                                 const DecStream *__restrict__ streams, DecHeader *__restrict__ hdrs) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    hdrs[i] = parse_container(format, in + streams[i].in_off, streams[i].in_len, nullptr);
}

// ------------------------------------------------------------------------------------------------
// bit reader for lane 0 over an LDS window of the input
constexpr uint32_t WIN_BYTES = 2048;  // LDS window (refilled by the whole wave)

struct BitIn {
    gptr_u8 g;           // stream base (global address space)
    uint64_t nbits;      // bits available in the stream
    uint64_t pos;        // next bit
    uint64_t win_base;   // byte offset of the window start (multiple of 4)
    uint32_t *win;       // LDS window
    int err;             // latched error: 0 none, 1 InvalidData, 2 UnexpectedEof  (bit.rs:84-94)
    uint32_t ecode, ea0, ea1;
};

// all lanes: make the window cover [pos/8, pos/8 + need) bytes
__device__ __forceinline__ void win_ensure(BitIn &b, uint32_t need, uint32_t lane) {
    const uint64_t byte = b.pos >> 3;
    if (byte >= b.win_base && byte + need <= b.win_base + WIN_BYTES) return;
    const uint64_t nb = (b.nbits + 7) >> 3;
    const uint64_t base = byte & ~3ull;
    // the stream base may be unaligned: load bytes (coalesced enough: 64 lanes x 32 bytes)
    for (uint32_t k = lane; k < WIN_BYTES / 4; k += 64) {
        const uint64_t o = base + 4ull * k;
        uint32_t v = 0;
        if (o + 4 <= nb) {
            gptr_u8 q = b.g + o;
            if ((((uint64_t)q) & 3) == 0) v = *(gptr_u32)q;
            else v = (uint32_t)q[0] | (uint32_t)q[1] << 8 | (uint32_t)q[2] << 16 | (uint32_t)q[3] << 24;
        } else {
            for (uint32_t j = 0; j < 4; ++j) if (o + j < nb) v |= (uint32_t)b.g[o + j] << (8 * j);
        }
        b.win[k] = v;
    }
    b.win_base = base;
    __syncthreads();
}

// lane 0: peek_bits_unchecked(w) (bit.rs:111-125): fails (latches EOF, returns 0) iff the stream has
// fewer than w bits left; once an error is latched a peek that needs more input returns 0.
__device__ __forceinline__ uint32_t bi_peek(BitIn &b, uint32_t w) {
    if (b.pos + w > b.nbits) {
        if (!b.err) { b.err = 2; b.ecode = ERR_EOF; }
        return 0;
    }
    const uint64_t rel = b.pos - (b.win_base << 3);
    const uint32_t wi = (uint32_t)(rel >> 5), sh = (uint32_t)rel & 31;
    const uint64_t two = (uint64_t)b.win[wi] | ((uint64_t)b.win[wi + 1] << 32);
    return (uint32_t)(two >> sh) & ((1u << w) - 1);
}
__device__ __forceinline__ void bi_skip(BitIn &b, uint32_t w) { b.pos += w; }
__device__ __forceinline__ uint32_t bi_read_unchecked(BitIn &b, uint32_t w) {
    const uint32_t v = bi_peek(b, w);
    bi_skip(b, w);
    return v;
}

// decode tables of one alphabet (LDS): primary table of PRI bits, entry = (symbol << 4) | width,
// 0 = unassigned; codes longer than PRI bits resolved by canonical walk (count/first/sorted).
constexpr uint32_t LIT_PRI = 10, DIST_PRI = 9;
struct HuffTab {
    uint16_t *pri;       // 1 << PRI entries
    uint16_t *sorted;    // symbols in (width, symbol) order
    uint16_t count[16];  // codes per width
    uint32_t pri_bits;
    uint32_t max_bw;     // huffman.rs:72: max code width (0 = empty table)
    uint32_t safe_bw;    // huffman.rs:123-132
};

// all lanes: build tables from code widths bw[0..n) (LDS).  Returns 0, or 1 = "Bit region conflict"
// (over-subscribed, huffman.rs:107-119).  eob = symbol whose width seeds safely_peek_bitwidth.
__device__ int tab_build(HuffTab &t, const uint8_t *bw, uint32_t n, int safe_some, uint32_t safe,
                         int eob, uint32_t lane, uint32_t *conflict_sym) {
    __shared__ uint32_t s_first[16], s_off[16], s_bad;
    for (uint32_t i = lane; i < (1u << t.pri_bits); i += 64) t.pri[i] = 0;
    if (lane == 0) {
        for (int w = 0; w < 16; ++w) t.count[w] = 0;
        uint32_t mx = 0;
        for (uint32_t s = 0; s < n; ++s) { t.count[bw[s]]++; if (bw[s] > mx) mx = bw[s]; }
        t.count[0] = 0;
        t.max_bw = mx;
        // canonical first codes; detect the first symbol whose code overflows its width
        uint32_t code = 0, off = 0;
        s_bad = 0xFFFFFFFFu;
        int left = 1;
        for (uint32_t w = 1; w <= 15; ++w) {
            code <<= 1;
            left <<= 1;
            s_first[w] = code;
            s_off[w] = off;
            if (s_bad == 0xFFFFFFFFu && (int)t.count[w] > left) {
                // the (left+1)-th symbol of this width is the first to collide
                s_bad = (w << 16) | (uint32_t)left;
            }
            left -= (int)t.count[w];
            if (left < 0) left = 0;  // keep scanning harmlessly
            code += t.count[w];
            off += t.count[w];
        }
        if (eob >= 0 && (uint32_t)eob < n && bw[eob] > 0) { safe_some = 1; safe = bw[eob]; }
        const uint32_t sp = safe_some ? safe : 1;
        t.safe_bw = mx < sp ? mx : sp;
    }
    __syncthreads();
    if (s_bad != 0xFFFFFFFFu) {
        // find the symbol: the (k+1)-th symbol (0-based k) of width w in symbol order
        if (lane == 0) {
            const uint32_t w = s_bad >> 16;
            uint32_t k = s_bad & 0xFFFF, found = 0;
            for (uint32_t s = 0; s < n; ++s)
                if (bw[s] == w) { if (k == 0) { found = s; break; } k--; }
            *conflict_sym = found;
        }
        __syncthreads();
        return 1;
    }
    for (uint32_t s = lane; s < n; s += 64) {
        const uint32_t w = bw[s];
        if (w == 0) continue;
        uint32_t rank = 0;
        for (uint32_t q = 0; q < s; ++q) rank += bw[q] == w;
        const uint32_t code = s_first[w] + rank;
        t.sorted[s_off[w] + rank] = (uint16_t)s;
        if (w <= t.pri_bits) {
            uint32_t r = __brev(code) >> (32 - w);  // LSB-first
            for (uint32_t i = r; i < (1u << t.pri_bits); i += (1u << w)) t.pri[i] = (uint16_t)((s << 4) | w);
        }
    }
    __syncthreads();
    return 0;
}

// lane 0: huffman::Decoder::decode_unchecked (huffman.rs:157-179).  Unassigned → InvalidData latched,
// returns symbol 0 and skips 16 bits like the reference (value 16: width 16, symbol 0).
__device__ __forceinline__ uint32_t tab_decode(const HuffTab &t, BitIn &b) {
    uint32_t peek = t.safe_bw;
    for (;;) {
        const uint32_t bits = bi_peek(b, peek);
        // canonical lookup of the code whose LSB-first bits are `bits` zero-extended
        uint32_t width = 16, sym = 0;  // unassigned
        if (t.max_bw == 0) {
            width = 16;
        } else {
            const uint32_t idx = bits & ((1u << t.pri_bits) - 1);
            const uint32_t e = t.pri[idx];
            if (e) { width = e & 15; sym = e >> 4; }
            else if (t.max_bw > t.pri_bits) {
                // long code: walk the canonical code bit by bit over the zero-extended bits
                uint32_t code = 0, first = 0, index = 0;
                const uint32_t full = bits;  // bits beyond `peek` are zero
                for (uint32_t w = 1; w <= t.max_bw; ++w) {
                    code |= (full >> (w - 1)) & 1;
                    const uint32_t cnt = t.count[w];
                    if (code < first + cnt) { width = w; sym = t.sorted[index + (code - first)]; break; }
                    index += cnt;
                    first = (first + cnt) << 1;
                    code <<= 1;
                }
            }
        }
        if (width <= peek) { bi_skip(b, width); return sym; }
        if (width > t.max_bw) {  // unassigned slot
            b.err = 1; b.ecode = ERR_HUFF;  // set_last_error overwrites (bit.rs:84-86)
            bi_skip(b, 16);
            return 0;
        }
        peek = width;
    }
}

__constant__ uint16_t c_len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                        31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t c_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                        2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t c_dist_base[30] = {1,    2,    3,    4,    5,    7,    9,    13,    17,    25,
                                         33,   49,   65,   97,   129,  193,  257,  385,   513,   769,
                                         1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t c_dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2,  3,  3,  4,  4,  5,  5,  6,
                                         6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t c_clen_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ------------------------------------------------------------------------------------------------
// inflate: one wavefront per job
constexpr uint32_t QN = 64;

__global__ __launch_bounds__(64) void inflate_kernel(const uint8_t *__restrict__ in,
                                                     uint8_t *__restrict__ out,
                                                     const InflateJob *__restrict__ jobs,
                                                     InflateResult *__restrict__ results) {
    __shared__ uint32_t win[WIN_BYTES / 4 + 2];
    __shared__ uint16_t lit_pri[1u << LIT_PRI], dist_pri[1u << DIST_PRI], cl_pri[128];
    __shared__ uint16_t lit_sorted[288], dist_sorted[32], cl_sorted[19];
    __shared__ uint8_t lens[640];
    __shared__ uint8_t clw[19];
    __shared__ uint32_t q[QN];
    __shared__ HuffTab T_lit, T_dist, T_cl;
    __shared__ uint32_t s_ctl[8];  // 0: queue count, 1: block done, 2: stop, 3: btype, 4: stored len
    __shared__ uint16_t l_len_base[29], l_dist_base[30];
    __shared__ uint8_t l_len_extra[29], l_dist_extra[30];
    if (threadIdx.x < 29) { l_len_base[threadIdx.x] = c_len_base[threadIdx.x]; l_len_extra[threadIdx.x] = c_len_extra[threadIdx.x]; }
    if (threadIdx.x < 30) { l_dist_base[threadIdx.x] = c_dist_base[threadIdx.x]; l_dist_extra[threadIdx.x] = c_dist_extra[threadIdx.x]; }

    const uint32_t lane = threadIdx.x;
    const InflateJob job = jobs[blockIdx.x];
    BitIn b;
    b.g = (gptr_u8)(in + job.in_off);
    b.nbits = job.in_len * 8;
    b.pos = job.start_bit;
    b.win = win;
    b.win_base = ~0ull >> 1;  // force the first fill
    b.err = 0; b.ecode = 0; b.ea0 = 0; b.ea1 = 0;
    uint8_t *o = out + job.out_off;
    const bool do_write = !(job.flags & JOB_COUNT_ONLY);
    uint64_t produced = 0;        // bytes produced by this job
    uint32_t status = 0, final_seen = 0, needs_hist = 0, nblocks = 0;
    uint64_t blk_out_start = 0, blk_start_bit = job.start_bit;
    uint64_t hist_avail = job.hist_avail;  // bytes of the member already produced before this job

    if (lane == 0) {
        T_lit.pri = lit_pri; T_lit.sorted = lit_sorted; T_lit.pri_bits = LIT_PRI;
        T_dist.pri = dist_pri; T_dist.sorted = dist_sorted; T_dist.pri_bits = DIST_PRI;
        T_cl.pri = cl_pri; T_cl.sorted = cl_sorted; T_cl.pri_bits = 7;
    }
    __syncthreads();

    for (;;) {
        // ---- block header: deflate::Decoder::read decode.rs:146-162
        blk_out_start = produced;
        blk_start_bit = b.pos;
        win_ensure(b, 700, lane);
        if (lane == 0) {
            s_ctl[2] = 0;
            const uint32_t bfinal = bi_read_unchecked(b, 1);
            uint32_t btype = 0;
            if (!b.err) btype = bi_read_unchecked(b, 2);
            if (b.err) s_ctl[2] = 1;
            s_ctl[3] = btype;
            s_ctl[5] = bfinal;
        }
        __syncthreads();
        b.pos = __shfl(b.pos, 0);
        if (s_ctl[2]) break;
        const uint32_t btype = s_ctl[3];
        final_seen = s_ctl[5];
        nblocks++;
        if (btype == 3) {
            if (lane == 0) { b.err = 1; b.ecode = ERR_BTYPE3; }
            break;
        }
        if (btype == 0) {
            // read_non_compressed_block decode.rs:81-111
            uint64_t byte = (b.pos + 7) >> 3;  // bit_reader.reset(): drop the partial byte
            const uint64_t nb = job.in_len;
            uint32_t len = 0;
            int bad = 0;
            if (nb < byte || nb - byte < 2) { bad = 2; byte = nb; }
            else {
                len = ld1(b.g + byte) | ld1(b.g + byte + 1) << 8;
                byte += 2;
                if (nb - byte < 2) { bad = 2; byte = nb; }
                else {
                    const uint32_t nlen = ld1(b.g + byte) | ld1(b.g + byte + 1) << 8;
                    byte += 2;
                    if (((~len) & 0xFFFF) != nlen) { bad = 1; b.ea0 = len; b.ea1 = nlen; }
                }
            }
            if (bad) {
                if (lane == 0) { b.err = bad; b.ecode = bad == 2 ? ERR_EOF : ERR_LEN_NLEN; }
                b.pos = byte << 3;
                break;
            }
            const uint64_t avail = nb - byte;
            const uint32_t take = len < avail ? len : (uint32_t)avail;
            if (do_write) {
                if (produced + take > job.out_cap) { if (lane == 0) { b.err = 3; b.ecode = ERR_NOSPACE; } break; }
                for (uint32_t k = lane; k < take; k += 64) o[produced + k] = b.g[byte + k];
                __threadfence_block();
            }
            produced += take;
            b.pos = (byte + take) << 3;
            if (take != len) {
                if (lane == 0) { b.err = 2; b.ecode = ERR_STORED_SHORT; b.ea0 = len; b.ea1 = take; }
                break;
            }
        } else {
            // ---- code tables
            int rc = 0;
            uint32_t csym = 0;
            uint32_t nl = 288, nd = 30;
            if (btype == 1) {
                // FixedHuffmanCodec::load symbol.rs:290-315
                for (uint32_t s = lane; s < 288; s += 64) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                for (uint32_t s = lane; s < 30; s += 64) lens[288 + s] = 5;
                __syncthreads();
            } else {
                // DynamicHuffmanCodec::load symbol.rs:387-456 (checked reads)
                if (lane == 0) {
                    s_ctl[2] = 0;
                    uint32_t hl = bi_read_unchecked(b, 5);
                    uint32_t hd = b.err ? 0 : bi_read_unchecked(b, 5);
                    uint32_t hc = b.err ? 0 : bi_read_unchecked(b, 4);
                    if (b.err) s_ctl[2] = 1;
                    else if (hd + 1 > 30) { b.err = 1; b.ecode = ERR_HDIST; b.ea0 = hd + 1; s_ctl[2] = 1; }
                    else {
                        for (int k = 0; k < 19; ++k) clw[k] = 0;
                        for (uint32_t k = 0; k < hc + 4 && !b.err; ++k) clw[c_clen_order[k]] = (uint8_t)bi_read_unchecked(b, 3);
                        if (b.err) s_ctl[2] = 1;
                    }
                    s_ctl[6] = hl + 257;
                    s_ctl[7] = hd + 1;
                }
                __syncthreads();
                if (s_ctl[2]) break;
                nl = s_ctl[6]; nd = s_ctl[7];
                rc = tab_build(T_cl, clw, 19, 1, 1, -1, lane, &csym);
                if (rc) { if (lane == 0) { b.err = 1; b.ecode = ERR_CONFLICT; b.ea0 = csym; } break; }
                if (lane == 0) {
                    // code length sequences (load_bitwidthes symbol.rs:459-484); one contiguous array:
                    // literal overflow spills into the distance list (symbol.rs:422-424)
                    uint32_t have = 0;
                    s_ctl[2] = 0;
                    for (int phase = 0; phase < 2 && !s_ctl[2]; ++phase) {
                        const uint32_t target = phase ? nl + nd : nl;
                        while (have < target) {
                            const uint32_t c = tab_decode(T_cl, b);
                            if (b.err) { s_ctl[2] = 1; break; }
                            if (c <= 15) lens[have++] = (uint8_t)c;
                            else if (c == 16) {
                                const uint32_t r = bi_read_unchecked(b, 2);
                                if (b.err) { s_ctl[2] = 1; break; }
                                if (have == 0) { b.err = 1; b.ecode = ERR_NO_PREV; s_ctl[2] = 1; break; }
                                const uint8_t last = lens[have - 1];
                                for (uint32_t k = 0; k < r + 3; ++k) lens[have++] = last;
                            } else if (c == 17) {
                                const uint32_t r = bi_read_unchecked(b, 3);
                                if (b.err) { s_ctl[2] = 1; break; }
                                for (uint32_t k = 0; k < r + 3; ++k) lens[have++] = 0;
                            } else {
                                const uint32_t r = bi_read_unchecked(b, 7);
                                if (b.err) { s_ctl[2] = 1; break; }
                                for (uint32_t k = 0; k < r + 11; ++k) lens[have++] = 0;
                            }
                        }
                    }
                    if (!s_ctl[2] && have - nl > nd) {
                        b.err = 1; b.ecode = ERR_DIST_LIST; b.ea0 = have - nl; b.ea1 = nd; s_ctl[2] = 1;
                    }
                }
                __syncthreads();
                if (s_ctl[2]) break;
            }
            const uint32_t dist_at = btype == 1 ? 288 : nl;
            rc = tab_build(T_lit, lens, nl, 0, 0, 256, lane, &csym);
            if (rc) { if (lane == 0) { b.err = 1; b.ecode = ERR_CONFLICT; b.ea0 = csym; } break; }
            rc = tab_build(T_dist, lens + dist_at, nd, 1, T_lit.safe_bw, -1, lane, &csym);
            if (rc) { if (lane == 0) { b.err = 1; b.ecode = ERR_CONFLICT; b.ea0 = csym; } break; }
            // ---- symbols: read_compressed_block decode.rs:112-130
            bool stop = false, nospace = false;
            for (;;) {
                win_ensure(b, QN * 6 + 16, lane);
                if (lane == 0) {
                    uint32_t n = 0, done = 0;
                    uint64_t prod = produced;
                    while (n < QN) {
                        // symbol::Decoder::decode_unchecked symbol.rs:193-244
                        const uint32_t d = tab_decode(T_lit, b);
                        uint32_t entry;
                        bool eob = false;
                        if (d <= 255) entry = d;
                        else if (d == 256) { eob = true; entry = 0; }
                        else if (d >= 286) { b.err = 1; b.ecode = ERR_286; b.ea0 = d; eob = true; entry = 0; }
                        else {
                            const uint32_t length = l_len_base[d - 257] + bi_read_unchecked(b, l_len_extra[d - 257]);
                            const uint32_t dc = tab_decode(T_dist, b);
                            const uint32_t distance = l_dist_base[dc % 30] + bi_read_unchecked(b, l_dist_extra[dc % 30]);
                            entry = 0x80000000u | (length << 16) | distance;  // distance <= 32768 fits 16 bits
                            if (!b.err) {
                                // Lz77Decoder::decode lib.rs:173-185
                                const uint64_t blen = hist_avail + prod;
                                if (blen < distance) {
                                    if (job.flags & JOB_SINGLE_BLOCK && (job.flags & JOB_COUNT_ONLY)) {
                                        needs_hist = 1;  // history unknown in the speculative pass
                                    } else {
                                        b.err = 1; b.ecode = ERR_BACKREF; b.ea0 = (uint32_t)blen; b.ea1 = distance;
                                    }
                                }
                            }
                            if (!b.err) prod += length;
                        }
                        if (b.err) { done = 2; break; }  // check_last_error after every symbol
                        if (eob) { done = 1; break; }
                        if (d <= 255) prod += 1;
                        q[n++] = entry;
                    }
                    s_ctl[0] = n;
                    s_ctl[1] = done;
                }
                __syncthreads();
                b.pos = __shfl(b.pos, 0);
                needs_hist = __shfl(needs_hist, 0);
                const uint32_t n = s_ctl[0], done = s_ctl[1];
                // ---- materialise the queue
                const uint32_t e = lane < n ? q[lane] : 0;
                const bool is_match = lane < n && (e >> 31);
                const uint32_t mylen = lane < n ? (is_match ? ((e >> 16) & 0x1FF) : 1) : 0;
                uint32_t x = mylen;  // inclusive scan
                for (int ofs = 1; ofs < 64; ofs <<= 1) {
                    const uint32_t y = __shfl_up(x, ofs);
                    if ((int)lane >= ofs) x += y;
                }
                const uint32_t total = __shfl(x, 63);
                const uint64_t at = produced + x - mylen;
                if (do_write && n) {
                    if (produced + total > job.out_cap) { if (lane == 0) { b.err = 3; b.ecode = ERR_NOSPACE; } stop = true; nospace = true; }
                    else {
                        if (lane < n && !is_match) o[at] = (uint8_t)e;
                        __threadfence_block();
                        uint64_t mm = __ballot(is_match);
                        while (mm) {
                            const uint32_t src_lane = (uint32_t)__builtin_ctzll(mm);
                            mm &= mm - 1;
                            const uint32_t me = __shfl(e, src_lane);
                            const uint64_t mat = __shfl(at, src_lane);
                            const uint32_t len = (me >> 16) & 0x1FF;
                            const uint32_t dist = me & 0xFFFF;
                            const uint8_t *srcp = o + mat - dist;
                            // out[k] = src[k mod dist] reproduces the overlapping forward copy (rle_decode)
                            if (dist >= len) { for (uint32_t k = lane; k < len; k += 64) o[mat + k] = srcp[k]; }
                            else { for (uint32_t k = lane; k < len; k += 64) o[mat + k] = srcp[k % dist]; }
                            __threadfence_block();
                        }
                    }
                }
                if (!nospace) produced += total;  // never report bytes that were not written
                __syncthreads();
                if (stop || done) { if (done == 2) stop = true; break; }
            }
            if (stop) break;
        }
        if (final_seen || (job.flags & JOB_SINGLE_BLOCK)) break;
        if (job.stop_bit != 0 && b.pos == job.stop_bit) break;   // end of a shard that holds no BFINAL block
    }
    // ---- result
    b.err = __shfl(b.err, 0);
    if (lane == 0) {
        status = b.err;
        InflateResult r;
        r.end_bit = b.pos;
        r.out_len = produced;
        r.status = status;
        r.final_seen = final_seen;
        r.err = b.ecode; r.a0 = b.ea0; r.a1 = b.ea1;
        r.needs_hist = needs_hist;
        r.nblocks = nblocks;
        r._pad = 0;
        r.blk_out_start = blk_out_start;
        r.blk_start_bit = blk_start_bit;
        results[blockIdx.x] = r;
    }
}

// ------------------------------------------------------------------------------------------------
// block finder, stage 1: every bit offset of the stream is tested for "dynamic block header with a
// complete code-length code".  A workgroup stages 4 KiB (+16 bytes) of the stream in LDS with coalesced
// dword loads.  Round 5: two steps per 256 dwords.  (a) A lane takes one dword, a 3-operation mask gives its offsets whose
// BTYPE reads 2 (and whose BFINAL is clear, outside the stream's tail) — an eighth of them, four per dword on average and
// nine in the worst lane of a wavefront — and the lane appends them to a list in LDS.  (b) The list is tested DENSELY, one
// entry per lane: HLIT / HDIST ranges, then the Kraft sum of the (at most 19) 3-bit code-length-code widths from a
// 4096-entry table of 4-field sums.  (Rounds 1-4 ran the test inside the per-lane loop over the mask: the loop's trip count
// is the wavefront's maximum, so more than half of the lanes idled through a 70-instruction body.)
// Survivors are collected per workgroup in LDS and appended to the lists of their CLASSES (below), one global atomic per
// list the workgroup has entries for (a single device-scope counter serialises at ~11 ns per atomic: FIND_SUB lists per class).
constexpr uint32_t FIND_DWORDS = 1024;   // dwords (4 KiB of stream) per workgroup
#ifndef LFX_FIND_FLUSH
#define LFX_FIND_FLUSH 4
#endif
constexpr uint32_t FIND_FLUSH = LFX_FIND_FLUSH;   // tiles of a workgroup per append of its survivors (stage 1 is a persistent grid)
constexpr uint32_t FIND_WL = 64 * FIND_FLUSH < 256 ? 256 : 64 * FIND_FLUSH;   // survivors a workgroup can hold between two appends (expected: ~14 per tile)
constexpr uint32_t FIND_LIST = 4096;     // offsets of 256 dwords that pass the mask: no two adjacent ones can (bit 1 clear, bit 2 set)
// Kraft contribution (128 >> l, 0 for l = 0) of four 3-bit fields | number of nonzero fields << 12
struct FindLut { uint16_t v[4096]; };
constexpr FindLut make_find_lut() {
    FindLut t{};
    for (uint32_t i = 0; i < 4096; ++i) {
        uint32_t k = 0, u = 0;
        for (int f = 0; f < 4; ++f) { const uint32_t l = (i >> (3 * f)) & 7; if (l) { k += 128u >> l; u++; } }
        t.v[i] = (uint16_t)(k | (u << 12));
    }
    return t;
}
__device__ const FindLut g_find_lut = make_find_lut();

// Round 6: stage 1 SORTS its survivors for stage 2.  A wavefront of stage 2 walks 64 candidates until its longest walk ends
// (201 steps for a mean of 43, tools/exp/find2_model.py on a real stream); how long a false candidate walks is mostly a matter
// of how many widths one symbol of ITS code-length code yields — the repeat codes 16 / 17 / 18 (3-6 / 3-10 / 11-138 widths
// each) and how short their codes are — and their three widths are the first three 3-bit fields of the header.  Class = the
// expected widths per symbol from those fields alone, E2 = 2 * 128 * E[widths per symbol] with every other symbol counted
// as one width, cut at the sixteenths of its distribution over a stream's false candidates: class 0 = the longest walks.
// Candidates of a class go to that class's lists; stage 2 takes its batches from the lists in order (longest first), so a
// batch holds candidates of one class: 108 steps per batch in the model.
constexpr uint32_t FIND_CLASSES = 16, FIND_SUB = FIND_SHARDS / FIND_CLASSES;     // lists = classes x sub-lists (atomics spread)
static_assert(FIND_CLASSES * FIND_SUB == FIND_SHARDS, "lists");
struct FindCls { uint8_t v[512]; };
constexpr FindCls make_find_cls() {
    constexpr uint32_t th[15] = {432, 568, 652, 832, 944, 1076, 1420, 1604, 2008, 2710, 3016, 4978, 5184, 5664, 9820};
    FindCls t{};
    for (uint32_t i = 0; i < 512; ++i) {
        const uint32_t l16 = i & 7, l17 = (i >> 3) & 7, l18 = i >> 6;
        const uint32_t p16 = l16 ? 128u >> l16 : 0u, p17 = l17 ? 128u >> l17 : 0u, p18 = l18 ? 128u >> l18 : 0u;
        const uint32_t rest = p16 + p17 + p18 <= 128 ? 128 - p16 - p17 - p18 : 0;          // (an over-subscribed triple never passes the Kraft test)
        const uint32_t e2 = p18 * 149 + p17 * 13 + p16 * 9 + 2 * rest;
        uint32_t c = 0;
        for (uint32_t k = 0; k < 15; ++k) c += e2 >= th[k] ? 1u : 0u;
        t.v[i] = (uint8_t)c;
    }
    return t;
}
__device__ const FindCls g_find_cls = make_find_cls();

__device__ __forceinline__ uint32_t find_wave_inclusive_sum(uint32_t x) {        // (row shifts and broadcasts, no LDS traffic)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

// Round 6: a PERSISTENT grid — a workgroup takes the 4 KiB tiles blockIdx.x, blockIdx.x + gridDim.x, ... and appends its
// survivors every FIND_FLUSH tiles: the lists are per class now, a workgroup's append is one device-scope atomic per class it
// holds, and those atomics (about half a nanosecond each in aggregate, whatever their addresses) are what the append costs —
// one workgroup per tile made 320 K of them where 32 K had been (+0.15 ms).  The next tile's dwords are loaded while the
// current one is tested.
__global__ __launch_bounds__(256) void find_blocks_stage1(const uint8_t *__restrict__ in, uint64_t nbytes,
                                                          uint64_t first_byte, uint32_t *__restrict__ count,
                                                          uint64_t *__restrict__ cand, uint32_t shard_cap,
                                                          uint64_t final_from_bit, uint32_t ntiles) {
    __shared__ __attribute__((aligned(16))) uint32_t sd[FIND_DWORDS + 4];
    __shared__ __attribute__((aligned(16))) uint16_t lut[4096];
    __shared__ uint16_t list[FIND_LIST];
    __shared__ uint64_t wl[FIND_WL];
    __shared__ uint32_t wn, s_wtot[4], ccnt[FIND_CLASSES], cbase[FIND_CLASSES];
    if (threadIdx.x == 0) wn = 0;
    if (threadIdx.x < FIND_CLASSES) ccnt[threadIdx.x] = 0;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t a = (uint64_t)in;
    gptr_u32 w = (gptr_u32)(a & ~3ull);
    const uint64_t shift = a & 3;                      // stream byte b lives at aligned byte b + shift
    const uint64_t wlast = (shift + nbytes + 3) / 4 - 1;
    const uint64_t stream_bits = nbytes * 8, lo_bit = first_byte * 8;
    const uint32_t sub = blockIdx.x % FIND_SUB;
    {
        const uint4 *gl = (const uint4 *)g_find_lut.v;
        const uint4 l0 = gl[tid], l1 = gl[tid + 256];
        ((uint4 *)lut)[tid] = l0;
        ((uint4 *)lut)[tid + 256] = l1;
    }
    // the staging loads of a tile are issued together (clamped addresses): a loop of dependent load → store pairs made every
    // workgroup wait five HBM round trips before its first test
    uint32_t v[5];
    auto tile_w0 = [&](uint32_t tile) { return (first_byte + (uint64_t)tile * (4 * FIND_DWORDS) + shift) >> 2; };
    auto fetch = [&](uint32_t tile) {
        const uint64_t tw0 = tile_w0(tile);
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) { const uint64_t idx = tw0 + tid + 256 * k; v[k] = w[idx < wlast ? idx : wlast]; }
    };
    uint32_t done = 0;                                 // tiles since the last append
    if (blockIdx.x < ntiles) fetch(blockIdx.x);
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t w0 = tile_w0(tile);
        __syncthreads();                               // (the tile before has been read; the append's counters are settled)
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) { const uint32_t i = tid + 256 * k; if (i < FIND_DWORDS + 4) sd[i] = v[k]; }
        __syncthreads();
        if (tile + gridDim.x < ntiles) fetch(tile + gridDim.x);
        for (uint32_t t0 = 0; t0 < FIND_DWORDS; t0 += 256) {
            // ---- (a) my dword = aligned dword w0 + t; its bit 0 is stream bit (4*(w0+t) - shift) * 8
            const uint32_t t = t0 + tid;
            const uint32_t d0 = sd[t], d1 = sd[t + 1];
            // offsets whose BTYPE field (bits 1..2) reads 2: bit 1 clear, bit 2 set — a quarter of them.  A header with
            // BFINAL set is wanted only near the end of the stream (final_from_bit, see launch_find_stage1): elsewhere the
            // offsets with bit 0 set are dropped too, which halves the candidates of both stages.
            const uint64_t dword_bit0 = (4 * (w0 + t) - shift) * 8;    // "negative" only for the first dword when shift > 0
            const uint32_t allow_final = dword_bit0 + 32 > final_from_bit ? ~0u : 0u;
            const uint64_t lo = (uint64_t)d0 | (uint64_t)d1 << 32;
            uint32_t pm = (uint32_t)(~(lo >> 1) & (lo >> 2)) & (~d0 | allow_final);
            const uint32_t cnt = (uint32_t)__popc(pm);
            const uint32_t incl = find_wave_inclusive_sum(cnt);
            if (lane == 63) s_wtot[wave] = incl;
            __syncthreads();                                     // (also: the previous trip's list has been read)
            uint32_t pos = incl - cnt, total = 0;
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) { const uint32_t wj = s_wtot[j]; pos += j < wave ? wj : 0u; total += wj; }
            while (pm) {
                const uint32_t ph = (uint32_t)__builtin_ctz(pm);
                pm &= pm - 1;
                list[pos++] = (uint16_t)(t << 5 | ph);
            }
            __syncthreads();
            // ---- (b) one list entry per lane: 96 bits of the stream from the offset on
            for (uint32_t i = tid; i < total; i += 256) {
                const uint32_t e = list[i], te = e >> 5, ph = e & 31;
                const uint32_t e0 = sd[te], e1 = sd[te + 1], e2 = sd[te + 2], e3 = sd[te + 3];
                const uint32_t x0 = __builtin_amdgcn_alignbit(e1, e0, ph), x1 = __builtin_amdgcn_alignbit(e2, e1, ph),
                               x2 = __builtin_amdgcn_alignbit(e3, e2, ph);
                const uint32_t hlit = (x0 >> 3) & 31, hdist = (x0 >> 8) & 31, hclen = (x0 >> 13) & 15;
                if (hlit > 29 || hdist > 29) continue;
                // the (hclen + 4) 3-bit fields start at bit 17: 57 bits at most
                const uint32_t nb = 3 * (hclen + 4);
                uint32_t f0 = __builtin_amdgcn_alignbit(x1, x0, 17), f1 = __builtin_amdgcn_alignbit(x2, x1, 17);
                const uint32_t m0 = nb >= 32 ? ~0u : (1u << nb) - 1u, m1 = nb > 32 ? (1u << (nb - 32)) - 1u : 0u;
                f0 &= m0;
                f1 &= m1;
                const uint32_t acc = (uint32_t)lut[f0 & 4095] + lut[(f0 >> 12) & 4095] + lut[__builtin_amdgcn_alignbit(f1, f0, 24) & 4095] +
                                     lut[(f1 >> 4) & 4095] + lut[(f1 >> 16) & 4095];
                if ((acc & 0xFFF) != 128 || (acc >> 12) < 2) continue;           // complete code-length code
                const uint64_t bit = (4 * (w0 + te) - shift) * 8 + ph;
                if ((int64_t)bit < (int64_t)lo_bit || bit + 96 > stream_bits) continue;
                const uint32_t slot = atomicAdd(&wn, 1u);
                if (slot < FIND_WL) wl[slot] = bit | (uint64_t)g_find_cls.v[f0 & 511u] << 56;      // (fields 0..2: the widths of 16, 17, 18)
            }
        }
        // ---- every FIND_FLUSH tiles and behind the last one: the survivors go to the lists of their classes (this workgroup's
        //      sub-list of each) — ranks inside the workgroup by LDS atomics, one global atomic per class it holds
        const bool last = tile + gridDim.x >= ntiles;
        if (++done < FIND_FLUSH && !last) continue;
        done = 0;
        __syncthreads();
        const uint32_t mine = wn;   // > FIND_WL would be a pathological input: counted, reported as overflow
        if (mine > FIND_WL) {
            if (threadIdx.x == 0) atomicAdd(&count[FIND_SHARDS], 1u);   // overflow marker
            return;
        }
        uint32_t cls[FIND_WL / 256], rnk[FIND_WL / 256];
#pragma unroll
        for (uint32_t q = 0; q < FIND_WL / 256; ++q) {
            const uint32_t k = threadIdx.x + 256 * q;
            cls[q] = k < mine ? (uint32_t)(wl[k] >> 56) : 0u;
            rnk[q] = k < mine ? atomicAdd(&ccnt[cls[q]], 1u) : 0u;
        }
        __syncthreads();
        if (threadIdx.x < FIND_CLASSES && ccnt[threadIdx.x]) cbase[threadIdx.x] = atomicAdd(&count[threadIdx.x * FIND_SUB + sub], ccnt[threadIdx.x]);
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < FIND_WL / 256; ++q) {
            const uint32_t k = threadIdx.x + 256 * q;
            if (k < mine) {
                const uint32_t at = cbase[cls[q]] + rnk[q];
                if (at < shard_cap) cand[(uint64_t)(cls[q] * FIND_SUB + sub) * shard_cap + at] = wl[k] & ((1ull << 56) - 1);
            }
        }
        __syncthreads();                               // (wl and the counters are free again)
        if (threadIdx.x == 0) wn = 0;
        if (threadIdx.x < FIND_CLASSES) ccnt[threadIdx.x] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// trailer verification per stream (gzip.rs:1030-1042: CRC-32 checked, ISIZE ignored;
// zlib.rs:387-401: Adler-32 big-endian).  One lane per stream.
__global__ void verify_trailers_kernel(int format, uint32_t count, const uint8_t *__restrict__ in,
                                       const DecStream *__restrict__ streams,
                                       const DecHeader *__restrict__ hdrs,
                                       InflateResult *__restrict__ results,
                                       const uint32_t *__restrict__ crc, const uint32_t *__restrict__ adler,
                                       uint64_t *__restrict__ consumed) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    InflateResult r = results[i];
    const DecHeader h = hdrs[i];
    const uint64_t n = streams[i].in_len;
    uint64_t used = h.deflate_off;
    if (h.status != 0) {
        r.status = h.status; r.err = h.err; r.a0 = h.a0; r.a1 = h.a1; r.out_len = 0;
    } else {
        used = (r.end_bit + 7) >> 3;  // end_bit is relative to the stream start
        if (used > n) used = n;
        if (r.status == 0 && format != 0) {
            const uint8_t *t = in + streams[i].in_off + used;
            const uint32_t need = format == 2 ? 8 : 4;
            if (n - used < need) { r.status = 2; r.err = ERR_EOF; used = n; }
            else if (format == 2) {
                const uint32_t c = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
                used += 8;
                if (c != crc[i]) { r.status = 1; r.err = ERR_CRC32; r.a0 = crc[i]; r.a1 = c; }
            } else {
                const uint32_t a = (uint32_t)t[0] << 24 | (uint32_t)t[1] << 16 | (uint32_t)t[2] << 8 | t[3];
                used += 4;
                if (a != adler[i]) { r.status = 1; r.err = ERR_ADLER32; r.a0 = adler[i]; r.a1 = a; }
            }
        }
    }
    results[i] = r;
    consumed[i] = used;
}

// ------------------------------------------------------------------------------------------------
#define LFX_LAUNCH_CHECK()                          \
    do {                                            \
        hipError_t e_ = hipGetLastError();          \
        if (e_ != hipSuccess) return (int)e_;       \
    } while (0)

int launch_container(hipStream_t st, int format, uint32_t count, const uint8_t *in,
                     const DecStream *streams, DecHeader *hdrs) {
    if (!count) return 0;
    hipLaunchKernelGGL(container_kernel, dim3((count + 63) / 64), dim3(64), 0, st, format, count, in, streams, hdrs);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_inflate(hipStream_t st, const uint8_t *in, uint8_t *out, const InflateJob *jobs,
                   InflateResult *results, uint32_t njobs) {
    if (!njobs) return 0;
    hipLaunchKernelGGL(inflate_kernel, dim3(njobs), dim3(64), 0, st, in, out, jobs, results);
    LFX_LAUNCH_CHECK();
    return 0;
}
// final_from_bit: headers with BFINAL set are reported only from this stream bit on.  The last block of a member is the
// only one that carries the flag, and the chain walk scans a block the finder did not report on demand — so a caller that
// walks the chain itself may pass the start of the stream's tail (where a last block of ordinary size begins) instead
// of 0 and save half of the finder's work; a caller that depends on every start being reported passes 0.
int launch_find_stage1(hipStream_t st, const uint8_t *in, uint64_t nbytes, uint64_t first_byte,
                       uint32_t *count, uint64_t *cand, uint32_t shard_cap, uint64_t final_from_bit, uint32_t n_cu) {
    if (nbytes <= first_byte) return 0;
    const uint64_t n = nbytes - first_byte;
    const uint64_t ntiles = div_up(n + 4, 4 * FIND_DWORDS);
    if (ntiles > 0xFFFFFFFFull) return (int)hipErrorInvalidValue;
    // persistent: as many workgroups as are resident at once (seven of 22.8 KB with FIND_FLUSH = 4), fewer when the stream has fewer tiles
    // (the runtime's own count of resident workgroups: a workgroup that has to wait for a slot would do its tiles behind all others)
    static int per_cu_dev[64] = {};
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    int &per_cu = per_cu_dev[dev_ & 63];
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)find_blocks_stage1, 256, 0) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 4;
        }
        per_cu = nb;
    }
    const uint64_t resident = (uint64_t)per_cu * (n_cu ? n_cu : 256u);
    hipLaunchKernelGGL(find_blocks_stage1, dim3((uint32_t)(ntiles < resident ? ntiles : resident)), dim3(256), 0, st, in, nbytes,
                       first_byte, count, cand, shard_cap, final_from_bit, (uint32_t)ntiles);
    LFX_LAUNCH_CHECK();
    return 0;
}
int launch_verify_trailers(hipStream_t st, int format, uint32_t count, const uint8_t *in,
                           const DecStream *streams, const DecHeader *hdrs, InflateResult *results,
                           const uint32_t *crc, const uint32_t *adler, uint64_t *consumed) {
    if (!count) return 0;
    hipLaunchKernelGGL(verify_trailers_kernel, dim3((count + 63) / 64), dim3(64), 0, st, format, count, in,
                       streams, hdrs, results, crc, adler, consumed);
    LFX_LAUNCH_CHECK();
    return 0;
}
}  // namespace lfx
