// lfx_match7.hip — LZ77 candidate stage for gfx950, round 5: a TWO-LEVEL BUCKET LRU WITH EXACT TAGS.
//
// Replaces, bit for bit, the table probe of DefaultLz77Encoder::flush (libflate_lz77/src/default.rs:76-87,146-182): per
// position the distance to the most recent earlier occurrence of its 3-byte prefix inside the window (0 = none) → cd[].
//
// What the earlier generations (lfx_match5.hip) paid for: a 14-bit hash table answers "most recent position with this
// HASH"; exactness came from a 38 KB window ring (3-byte compares), a 75 KB link ring (duplicate-collapsed chains), pointer
// jumping and chain walks between two workgroup barriers per 960 positions — 385 wave-instructions per 64 positions.
//
// This formulation keeps no bytes and no links in LDS.  The 24-bit prefix is mapped by a BIJECTION (multiplication by an odd
// constant mod 2^24) to (bucket: 14 bits, tag: 10 bits), so bucket + tag IS the prefix.  Per bucket two entries
// {position, tag}:
//     head   = the most recent position of the bucket,
//     second = the most recent position of the bucket whose tag differs from head's tag.
// Inserting (p, t):  o1 = exchange(head, (p, t));  tag(o1) == t ?  second is only read  :  o2 = exchange(second, o1).
//   * tag(o1) == t: o1 is the most recent occurrence of the prefix — the answer.
//   * else tag(o2) == t: o2 is (every position of the bucket between o2 and p carries tag(o1)).
//   * an entry older than the window ends the search: nothing more recent carries the prefix.
//   * otherwise three prefixes alternate in the bucket inside one window (1 % of a text's positions): UNRESOLVED, left to
//     lz77_resolve7_kernel below.
// Both exchanges are ordered LDS read-modify-writes (ds_wrxchg_rtn_b32 / ds_mskor_rtn_b32: lanes of one instruction that hit
// the same dword are served in ascending lane order, a wavefront's instructions in issue order — measured,
// tools/exp/lds_lru.hip and tools/exp/mskor_test.hip, and checked at run time: a lane that receives a position from its own
// future raises flags[0] and the host falls back to the first-generation kernel).  So wave 0 issues ONE exchange per 64
// positions on `head`, wave 1 ONE masked exchange per 64 positions on `second` (a tile behind), and fourteen helper
// wavefronts turn bytes into requests (a tile ahead) and results into cd[] (two tiles behind).  One LDS-only barrier per
// 896 positions; nobody waits for a chain.
//
// The same two values give every position its DUPLICATE-COLLAPSED LINK for free — the most recent position of the bucket
// with ANOTHER prefix: o1 when the tags differ, the value read from `second` when they are equal — written to glnk[] (2 bytes
// per position, as lfx_match5 did).  lz77_resolve7_kernel walks those links for the unresolved positions through global
// memory, compacted to dense lanes: link, compare the 3 bytes, stop at the first exact hit or beyond the window
// (default.rs:81, inclusive).  Exactness argument as in lfx_match3.hip: the chain visits the most recent member of every run
// of equal prefixes of the bucket in decreasing position order.
//
// Positions are stored in full (segment-relative, 19 bits): nothing aliases, no sweep of stale fields.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "lfx_common.h"
#include "lfx_device.h"

namespace lfx {
namespace m7 {

constexpr int THREADS = 1024;
constexpr uint32_t HW = 14;                    // helper wavefronts (waves 2..15); wave 0: head, wave 1: second
constexpr uint32_t TILE = HW * 64;             // 896 positions per barrier
constexpr uint32_t BUCKET_BITS = 14, TAG_BITS = 24 - BUCKET_BITS;
constexpr uint32_t TAG_MASK = (1u << TAG_BITS) - 1;
constexpr uint32_t KEY_MULT = 0x00C5A3B5u;     // odd: k → k·M mod 2^24 is a bijection; chosen on text (tools/parse... see DESIGN §3.1b)
// entry = (spos << TAG_BITS) | tag, spos = position − base + SPOS0: 0 (an empty slot) is further than any window
constexpr uint32_t SPOS0 = MAX_WINDOW + 1;
constexpr uint32_t UNRES = 0x8000u;            // cd value UNRES + (d2 − 1), d2 in [2, 32768]: unresolved, the walk continues at p − d2
constexpr uint32_t RQ_INVALID = 0x80000000u;

// LDS layout (bytes)
constexpr uint32_t OFF_HEAD = 0;
constexpr uint32_t OFF_SEC = OFF_HEAD + (4u << BUCKET_BITS);
constexpr uint32_t OFF_DUMMY = OFF_SEC + (4u << BUCKET_BITS);     // 64 dwords: where the lanes without a position exchange
constexpr uint32_t OFF_RQ = OFF_DUMMY + 256;                      // 4 tiles of requests (bucket << TAG_BITS | tag, or RQ_INVALID)
constexpr uint32_t OFF_R1 = OFF_RQ + 4 * TILE * 4;                // 3 tiles: what the exchange on head returned
constexpr uint32_t OFF_R2 = OFF_R1 + 3 * TILE * 4;                // 2 tiles: what the exchange on second returned
constexpr uint32_t LDS_BYTES = OFF_R2 + 2 * TILE * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(SEG_POSITIONS + MAX_WINDOW + SPOS0 + 8 * TILE < (1u << (32 - TAG_BITS)), "segment-relative positions fit the entry");
static_assert(HW == 14, "the exchange waves issue two batches of seven; seven loading and seven storing helpers");

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// seven exchanges, in order; no wait (xchg7_wait below orders the uses of the results)
__device__ __forceinline__ void xchg7(uint32_t (&old)[7], const uint32_t (&addr)[7], const uint32_t (&val)[7]) {
    asm volatile(
        "ds_wrxchg_rtn_b32 %0, %7, %14\n\t"
        "ds_wrxchg_rtn_b32 %1, %8, %15\n\t"
        "ds_wrxchg_rtn_b32 %2, %9, %16\n\t"
        "ds_wrxchg_rtn_b32 %3, %10, %17\n\t"
        "ds_wrxchg_rtn_b32 %4, %11, %18\n\t"
        "ds_wrxchg_rtn_b32 %5, %12, %19\n\t"
        "ds_wrxchg_rtn_b32 %6, %13, %20"
        : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3]), "=&v"(old[4]), "=&v"(old[5]), "=&v"(old[6])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]),
          "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6])
        : "memory");
}
// seven more, then ONE wait for all fourteen: `prev` (the results of the batch before) is an in/out operand, so that no use
// of it can be scheduled in front of the wait
__device__ __forceinline__ void xchg7_wait(uint32_t (&old)[7], const uint32_t (&addr)[7], const uint32_t (&val)[7], uint32_t (&prev)[7]) {
    asm volatile(
        "ds_wrxchg_rtn_b32 %0, %14, %21\n\t"
        "ds_wrxchg_rtn_b32 %1, %15, %22\n\t"
        "ds_wrxchg_rtn_b32 %2, %16, %23\n\t"
        "ds_wrxchg_rtn_b32 %3, %17, %24\n\t"
        "ds_wrxchg_rtn_b32 %4, %18, %25\n\t"
        "ds_wrxchg_rtn_b32 %5, %19, %26\n\t"
        "ds_wrxchg_rtn_b32 %6, %20, %27\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3]), "=&v"(old[4]), "=&v"(old[5]), "=&v"(old[6]),
          "+v"(prev[0]), "+v"(prev[1]), "+v"(prev[2]), "+v"(prev[3]), "+v"(prev[4]), "+v"(prev[5]), "+v"(prev[6])
        : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]),
          "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6])
        : "memory");
}
// seven masked exchanges (mem = (mem & ~mask) | val, returns the old dword; mask 0 / val 0: an ORDERED READ), one wait
__device__ __forceinline__ void mskor7(uint32_t (&old)[7], const uint32_t (&addr)[7], const uint32_t (&mask)[7], const uint32_t (&val)[7]) {
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

struct __attribute__((packed, aligned(4))) U32x2 { uint32_t a, b; };   // an 8-byte load at any dword address

}  // namespace m7

// flags[0] |= 1 when an exchange returned a position from the lane's own future (results are then discarded by the host).
// DBG: per-wavefront cycle stamps of workgroup 0 (LFX_DEBUG): work / barrier wait.
template <bool DBG>
__global__ __launch_bounds__(m7::THREADS) void lz77_match7_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint16_t *__restrict__ cd,
    uint16_t *__restrict__ glnk, uint32_t *__restrict__ flags, uint64_t *__restrict__ dbg) {
    using namespace m7;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    uint32_t *rqb = (uint32_t *)(smem + OFF_RQ);
    uint32_t *r1b = (uint32_t *)(smem + OFF_R1);
    uint32_t *r2b = (uint32_t *)(smem + OFF_R2);
    // LDS byte address of the tables for the asm exchanges (taking it from the pointer also makes the array escape)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const SegDesc sg = segs[blockIdx.x];
    const ChunkDesc ch = chunks[sg.chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = (uint32_t)ch.len;
    if (ch.flags & CH_LITERALS) return;           // NoCompressionLz77Encoder chunks never come here
    const uint32_t end = (n > 3 ? n : 3) - 3;     // default.rs:75
    const uint32_t q0 = sg.start;                 // first position answered by this segment
    const uint32_t q1 = min(sg.start + sg.len, end);
    if (q0 >= q1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;   // warm-up: inserted and linked, not answered
    const uint32_t base = l0 & ~3u;                               // tile origin
    const int ntiles = (int)((q1 - base + TILE - 1) / TILE);
    uint16_t *cd_c = cd + ch.in_off;
    uint16_t *glnk_s = glnk + (uint64_t)sg.lnk_base * 64u;
    // the chunk's bytes as dwords of the (4-byte aligned) allocation in front of them
    const uint64_t a0 = (uint64_t)(in + ch.in_off);
    const gptr_u32 srcw = (gptr_u32)(a0 & ~3ull);
    const uint32_t shift = (uint32_t)(a0 & 3);
    const uint64_t lastm1 = ((in_bytes - ch.in_off + shift + 3) >> 2) - 1;   // last dword that holds input bytes

    // ---- prologue: empty tables
    {
        uint4 *t4 = (uint4 *)smem;
        for (uint32_t i = tid; i < (OFF_RQ >> 4); i += THREADS) t4[i] = make_uint4(0, 0, 0, 0);
    }
    lds_barrier();

    // helpers: waves 2..8 turn bytes into requests (loads only), waves 9..15 turn results into answers (stores only), two
    // 64-position groups of the tile each.  (A wavefront that both loads and stores gets `s_waitcnt vmcnt(0)` in front of
    // every use of a loaded value — loads and stores share the counter and may complete out of order with respect to each
    // other — and a look-ahead of two tiles would be worth nothing.)
    const bool is_p = wave >= 2 && wave < 2 + HW / 2, is_c = wave >= 2 + HW / 2;
    const uint32_t hidx = ((is_c ? wave - (2 + HW / 2) : wave - 2) * 2) * 64 + lane;   // index of the first group's position inside the tile
    uint64_t cy_work = 0, cy_wait = 0;
    bool viol = false;

    if (wave < 2) __builtin_amdgcn_s_setprio(3);

    // Tiles in flight in iteration i: loads of tile i+3 (consumed two iterations later, so that no wavefront ever waits for
    // HBM), P(i+1) requests, X1(i) exchange on head, X2(i-1) exchange on second, C(i-2) answers.  Every role runs its OWN
    // loop (same trip count, one barrier per trip): inside one loop the role test would be a branch per iteration, and the
    // compiler's wait insertion would have to assume that the other register set's loads were never issued.
    const int i_first = -3, i_end = ntiles + 2;     // i_first .. i_end-1 (+1 when the count is odd: stages are predicated)
    auto sync = [&](uint64_t c0) {
        const uint64_t c1 = DBG ? clock64() : 0;
        lds_barrier();
        if (DBG) { cy_work += c1 - c0; cy_wait += clock64() - c1; }
    };
    if (is_p) {
        // the loads alternate between two register sets (an iteration consumes the set that was loaded two iterations ago and
        // reloads it): tiles 0 and 1 are loaded by the first two iterations
        uint32_t ldA[4] = {0, 0, 0, 0}, ldB[4] = {0, 0, 0, 0};
        auto p_iter = [&](int i, uint32_t (&ld)[4]) {
            const uint64_t c0 = DBG ? clock64() : 0;
            // ---- P(i+1): bytes → prefix → (bucket, tag)
            const int tp = i + 1;
            if (tp >= 0 && tp < ntiles) {
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    const uint32_t p = base + (uint32_t)tp * TILE + hidx + g * 64;
                    const uint32_t key = __builtin_amdgcn_alignbyte(ld[2 * g + 1], ld[2 * g], (p + shift) & 3u) & 0xFFFFFFu;
                    const bool val = p >= l0 && p < q1;
                    rqb[(uint32_t)(tp & 3) * TILE + hidx + g * 64] = val ? (key * KEY_MULT) & 0xFFFFFFu : RQ_INVALID;
                }
            }
            // ---- loads of tile i+3, into the registers P just consumed (unconditional: clamped addresses)
#pragma unroll
            for (uint32_t g = 0; g < 2; ++g) {
                const uint32_t p = base + (uint32_t)(i + 3) * TILE + hidx + g * 64;
                const uint64_t wi = ((uint64_t)p + shift) >> 2;
                ld[2 * g] = srcw[min(wi, lastm1)];
                ld[2 * g + 1] = srcw[min(wi + 1, lastm1)];
            }
            sync(c0);
        };
        for (int i = i_first; i < i_end; i += 2) { p_iter(i, ldA); p_iter(i + 1, ldB); }
    } else if (is_c) {
        auto c_iter = [&](int i) {
            const uint64_t c0 = DBG ? clock64() : 0;
            // ---- C(i-2): the two exchanged values → answer, collapsed link
            const int tc = i - 2;
            if (tc >= 0 && tc < ntiles) {
                uint32_t kk[2], o1[2], o2[2];
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    kk[g] = rqb[(uint32_t)(tc & 3) * TILE + hidx + g * 64];
                    o1[g] = r1b[(uint32_t)(tc % 3) * TILE + hidx + g * 64];
                    o2[g] = r2b[(uint32_t)(tc & 1) * TILE + hidx + g * 64];
                }
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    const uint32_t p = base + (uint32_t)tc * TILE + hidx + g * 64;
                    const uint32_t sp = p - base + SPOS0;
                    const uint32_t p1 = o1[g] >> TAG_BITS, p2 = o2[g] >> TAG_BITS;
                    const bool valid = (int32_t)kk[g] >= 0;
                    viol |= valid && (p1 >= sp || p2 >= sp);
                    const uint32_t d1 = sp - p1, d2 = sp - p2;
                    const bool same1 = ((o1[g] ^ kk[g]) & TAG_MASK) == 0, same2 = ((o2[g] ^ kk[g]) & TAG_MASK) == 0;
                    // the link: the most recent position of the bucket with another prefix
                    const uint32_t dl = same1 ? d2 : d1;
                    const uint32_t lnk = dl <= MAX_WINDOW ? dl : 0u;
                    // the answer (default.rs:81: inclusive window).  d1 > window: the bucket's most recent position is out of
                    // reach, so is everything; d2 >= 2 always (second lies in front of head)
                    const uint32_t deep = d2 > window ? 0u : (same2 ? d2 : UNRES + (d2 - 1));
                    const uint32_t ans = d1 > window ? 0u : (same1 ? d1 : deep);
                    if (valid) glnk_s[p - base] = (uint16_t)lnk;
                    if (valid && p >= q0) cd_c[p] = (uint16_t)ans;
                }
            }
            sync(c0);
        };
        for (int i = i_first; i < i_end; i += 2) { c_iter(i); c_iter(i + 1); }
    } else if (wave == 0) {
        auto x1_iter = [&](int i) {
            const uint64_t c0 = DBG ? clock64() : 0;
            // ---- X1(i): head ← (position, tag), in position order; the old entries → r1
            if (i >= 0 && i < ntiles) {
                const uint32_t *rq = rqb + (uint32_t)(i & 3) * TILE;
                uint32_t *r1 = r1b + (uint32_t)(i % 3) * TILE;
                const uint32_t ent0 = ((uint32_t)i * TILE + lane + SPOS0) << TAG_BITS;
                uint32_t q[HW];
#pragma unroll
                for (uint32_t g = 0; g < HW; ++g) q[g] = rq[g * 64 + lane];
                uint32_t aa[7], va[7], oa[7], ab[7], vb[7], ob[7];
#pragma unroll
                for (uint32_t s = 0; s < 7; ++s) {
                    const uint32_t ra = q[s], rb = q[7 + s];
                    aa[s] = lds0 + ((int32_t)ra < 0 ? OFF_DUMMY + lane * 4 : OFF_HEAD + ((ra >> (TAG_BITS - 2)) & ~3u));
                    ab[s] = lds0 + ((int32_t)rb < 0 ? OFF_DUMMY + lane * 4 : OFF_HEAD + ((rb >> (TAG_BITS - 2)) & ~3u));
                    va[s] = (ent0 + ((s * 64u) << TAG_BITS)) | (ra & TAG_MASK);
                    vb[s] = (ent0 + (((7 + s) * 64u) << TAG_BITS)) | (rb & TAG_MASK);
                }
                xchg7(oa, aa, va);
                xchg7_wait(ob, ab, vb, oa);
#pragma unroll
                for (uint32_t s = 0; s < 7; ++s) {
                    r1[s * 64 + lane] = oa[s];
                    r1[(7 + s) * 64 + lane] = ob[s];
                }
            }
            sync(c0);
        };
        for (int i = i_first; i < i_end; i += 2) { x1_iter(i); x1_iter(i + 1); }
    } else {
        auto x2_iter = [&](int i) {
            const uint64_t c0 = DBG ? clock64() : 0;
            // ---- X2(i-1): second ← old head where the tags differ (an ordered read where they are equal) → r2
            const int t2 = i - 1;
            if (t2 >= 0 && t2 < ntiles) {
                const uint32_t *rq = rqb + (uint32_t)(t2 & 3) * TILE;
                const uint32_t *r1 = r1b + (uint32_t)(t2 % 3) * TILE;
                uint32_t *r2 = r2b + (uint32_t)(t2 & 1) * TILE;
                uint32_t q[HW], o[HW];
#pragma unroll
                for (uint32_t g = 0; g < HW; ++g) { q[g] = rq[g * 64 + lane]; o[g] = r1[g * 64 + lane]; }
#pragma unroll
                for (uint32_t h = 0; h < HW; h += 7) {
                    uint32_t ad[7], mk[7], vl[7], od[7];
#pragma unroll
                    for (uint32_t s = 0; s < 7; ++s) {
                        const uint32_t r = q[h + s], o1 = o[h + s];
                        const bool inval = (int32_t)r < 0;
                        const bool differ = !inval && ((r ^ o1) & TAG_MASK) != 0;
                        ad[s] = lds0 + (inval ? OFF_DUMMY + lane * 4 : OFF_SEC + ((r >> (TAG_BITS - 2)) & ~3u));
                        mk[s] = differ ? 0xFFFFFFFFu : 0u;
                        vl[s] = differ ? o1 : 0u;
                    }
                    mskor7(od, ad, mk, vl);
#pragma unroll
                    for (uint32_t s = 0; s < 7; ++s) r2[(h + s) * 64 + lane] = od[s];
                }
            }
            sync(c0);
        };
        for (int i = i_first; i < i_end; i += 2) { x2_iter(i); x2_iter(i + 1); }
    }
    if (__ballot(viol) && lane == 0) atomicOr(flags, 1u);
    if (DBG && dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_work; d[1] = cy_wait; d[2] = 0; d[3] = 0; d[4] = 0; d[5] = (uint64_t)ntiles; d[6] = 0; d[7] = 0;
    }
}

// The unresolved positions (cd >= UNRES + 1: three or more prefixes alternate in the bucket inside one window): follow the
// duplicate-collapsed links from p - d2 on until the prefix is found or the window ends.  One workgroup per slab of a
// segment; the slab's unresolved positions are compacted into an LDS list so that the walks run on dense lanes, a walk's two
// loads per hop (the link of the position reached, the dwords that hold its prefix) issued together.
namespace r7 {
constexpr uint32_t SLAB = 8192;
constexpr uint32_t THREADS = 256;
constexpr uint32_t SLABS_PER_SEG = SEG_POSITIONS / SLAB;
}  // namespace r7

__global__ __launch_bounds__(r7::THREADS) void lz77_resolve7_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint16_t *__restrict__ cd, const uint16_t *__restrict__ glnk) {
    using namespace r7;
    __shared__ uint32_t list[SLAB];
    __shared__ uint32_t cnt;
    const SegDesc sg = segs[blockIdx.x / SLABS_PER_SEG];
    const uint32_t slab = blockIdx.x % SLABS_PER_SEG;
    if (slab * SLAB >= sg.len) return;
    const ChunkDesc ch = chunks[sg.chunk];
    if (ch.flags & CH_LITERALS) return;
    const uint32_t n = (uint32_t)ch.len;
    const uint32_t end = (n > 3 ? n : 3) - 3;
    const uint32_t q0 = sg.start, q1 = min(sg.start + sg.len, end);
    const uint32_t s0 = q0 + slab * SLAB, s1 = min(s0 + SLAB, q1);
    if (s0 >= s1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;
    const uint32_t base = l0 & ~3u;
    uint16_t *cd_c = cd + ch.in_off;
    const uint16_t *glnk_s = glnk + (uint64_t)sg.lnk_base * 64u;
    const uint64_t a0 = (uint64_t)(in + ch.in_off);
    const gptr_u32 srcw = (gptr_u32)(a0 & ~3ull);
    const uint32_t shift = (uint32_t)(a0 & 3);
    const uint64_t lastm1 = ((in_bytes - ch.in_off + shift + 3) >> 2) - 1;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) cnt = 0;
    __syncthreads();
    // ---- the slab's answers, sixteen bytes per lane and round (cd_c + s0 is 2-byte aligned only: the head up to the next
    //      16-byte boundary is taken by single loads)
    {
        const uint64_t addr = (uint64_t)(cd_c + s0);
        const uint32_t head = min((uint32_t)(((16 - (addr & 15)) & 15) >> 1), s1 - s0);
        if (tid < head) {
            const uint32_t v = cd_c[s0 + tid];
            if (v > m7::UNRES) list[atomicAdd(&cnt, 1u)] = s0 + tid;
        }
        const uint32_t b0 = s0 + head;
        const uint32_t nvec = (s1 - b0) >> 3;
        const uint4 *v4 = (const uint4 *)(cd_c + b0);
        for (uint32_t i = tid; i < nvec; i += THREADS) {
            const uint4 q = v4[i];
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if ((w[k] & 0xFFFFu) > m7::UNRES) list[atomicAdd(&cnt, 1u)] = b0 + i * 8 + 2 * k;
                if ((w[k] >> 16) > m7::UNRES) list[atomicAdd(&cnt, 1u)] = b0 + i * 8 + 2 * k + 1;
            }
        }
        const uint32_t t0 = b0 + nvec * 8;
        if (t0 + tid < s1) {
            const uint32_t v = cd_c[t0 + tid];
            if (v > m7::UNRES) list[atomicAdd(&cnt, 1u)] = t0 + tid;
        }
    }
    __syncthreads();
    const uint32_t total = cnt;
    auto key_at = [&](uint32_t p, uint32_t w0, uint32_t w1) { return __builtin_amdgcn_alignbyte(w1, w0, (p + shift) & 3u) & 0xFFFFFFu; };
    for (uint32_t i = tid; i < total; i += THREADS) {
        const uint32_t p = list[i];
        uint32_t dist = (uint32_t)cd_c[p] - m7::UNRES + 1;             // d2: the position of `second`, known to carry another prefix
        const uint64_t wp = ((uint64_t)p + shift) >> 2;
        const uint32_t key = key_at(p, srcw[min(wp, lastm1)], srcw[min(wp + 1, lastm1)]);
        uint32_t r = p - dist;
        uint32_t l = glnk_s[r - base];
        uint32_t ans = 0;
        for (;;) {
            // (a link never reaches in front of l0, the first inserted position: l <= r - l0 always)
            dist += l;
            if (l == 0 || dist > window) break;                         // default.rs:81 (inclusive window)
            r -= l;
            const uint64_t wi = ((uint64_t)r + shift) >> 2;
            const uint32_t w0 = srcw[min(wi, lastm1)], w1 = srcw[min(wi + 1, lastm1)];
            l = glnk_s[r - base];
            if (key_at(r, w0, w1) == key) { ans = dist; break; }
        }
        cd_c[p] = (uint16_t)ans;
    }
}

int launch_match7(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint16_t *cd, uint16_t *glnk, uint32_t *flags, uint64_t *dbg) {
    if (nsegs == 0) return 0;
    if (dbg)
        hipLaunchKernelGGL(lz77_match7_kernel<true>, dim3(nsegs), dim3(m7::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, glnk, flags, dbg);
    else
        hipLaunchKernelGGL(lz77_match7_kernel<false>, dim3(nsegs), dim3(m7::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, glnk, flags, dbg);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return (int)e_;
    hipLaunchKernelGGL(lz77_resolve7_kernel, dim3(nsegs * r7::SLABS_PER_SEG), dim3(r7::THREADS), 0, st, in, in_bytes, chunks, segs,
                       window, cd, glnk);
    e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

}  // namespace lfx
