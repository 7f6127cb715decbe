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
// future raises flags[0] and the host falls back to the first-generation kernel).  So ONE exchange per 64 positions on `head`
// (wave 0) and ONE masked exchange per 64 positions on `second` (wave 1, a tile behind) are all the ordered work there is —
// fourteen in a row per tile, one wait; fourteen helper wavefronts turn bytes into requests (a tile ahead) and results into
// cd[] (two tiles behind).  One LDS-only barrier per 896 positions; nobody waits for a chain.  (Measured and dropped: each
// exchange stage shared by two wavefronts that take turns through a counter in LDS — the hand-over costs an LDS round trip,
// as much as the second batch of exchanges in the same wavefront.)
//
// The same two values give every position its DUPLICATE-COLLAPSED LINK for free — the most recent position of the bucket
// with ANOTHER prefix: o1 when the tags differ, the value read from `second` when they are equal — written to glnk[] together
// with the position's own tag (4 bytes per position: link | tag << 16).  A chain never leaves its bucket, so along a chain the
// tag alone IS the prefix: the walk below compares tags and never reads the input (a hop is ONE scattered 4-byte load, not a
// link and the bytes it leads to — the resolver's time is its scattered 64-byte sectors from HBM).  The unresolved positions are marked in one 64-bit word per 64 positions (a ballot);
// lz77_compact7_kernel turns the words into a list per segment (in the parse stage's staging buffer, idle until then) and
// lz77_resolve7_kernel walks the links for them through global memory: link, compare the 3 bytes, stop at the first exact hit
// or beyond the window
// (default.rs:81, inclusive).  Exactness argument as in lfx_match3.hip: the chain visits the most recent member of every run
// of equal prefixes of the bucket in decreasing position order.
//
// Positions are stored in full (segment-relative, 19 bits): nothing aliases, no sweep of stale fields.
//
// What bounds the kernel: the workgroup's vector instructions (a SIMD issues one wave-instruction per four cycles: 4 SIMDs
// = one per cycle), so every stage below is written for instruction count — interior tiles take paths without per-lane
// validity tests, addresses are one 32-bit offset per wavefront and tile plus immediate offsets, the multiplication is the
// full-rate 24-bit one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "lfx_common.h"
#include "lfx_device.h"

namespace lfx {
namespace m7 {

constexpr int THREADS = 1024;
constexpr uint32_t NG = 14;                    // 64-position groups per tile
constexpr uint32_t TILE = NG * 64;             // 896 positions per barrier
// wavefront roles: 0 exchange on head (X1), 1 exchange on second (X2), seven "P" wavefronts bytes → requests (loads only),
// seven "C" wavefronts results → answers (stores only), two groups each.  (A wavefront that both loads and stores gets
// `s_waitcnt vmcnt(0)` in front of every use of a loaded value — loads and stores share the counter and may complete out of
// order with respect to each other — and a look-ahead of several tiles would be worth nothing.)
// The kernel is bound by its vector instructions per SIMD (wavefront w runs on SIMD w mod 4), so the roles are dealt by
// their instruction counts per tile — X1 70, X2 125, P 35, C 55: SIMD 0 = X1 P C C, SIMD 1 = X2 P P P, SIMD 2 = P P C C,
// SIMD 3 = P C C C.
//                              wave:  15 14 13 12 11 10  9  8  7  6  5  4  3  2  1  0
constexpr uint64_t ROLE_IDX = 0x6564325104321000ull;     // index of the wavefront among those of its role (a nibble each)
constexpr uint32_t ROLE_IS_P = 0x227Cu;                   // waves 2 3 4 5 6 9 13
constexpr uint32_t ROLE_IS_C = 0xDD80u;                   // waves 7 8 10 11 12 14 15
// Wavefront priorities (s_setprio, two bits per wavefront): the exchange wavefronts 3, the C wavefronts 2, the P wavefronts 0.
// Every wavefront has to reach the tile's barrier; the C wavefronts are the ones that arrive last (LFX_DEBUG: work 250 .. 300 K
// cycles of 411 K against 120 .. 200 K for a P wavefront), and on a SIMD they share with P wavefronts they now issue first:
// the kernel 0.66 -> 0.60 ms.  (Measured, tools/exp/r5_prio.sh: C at 3 or 1, P at 1, X1 at 0..2 — all between the two.)
#ifndef LFX_M7_PRIO
#define LFX_M7_PRIO 0xa2a2800fu
#endif
constexpr int AHEAD = 4;                       // register sets of the loading wavefronts = iterations between a load and its use
// How the stages of the pipeline wait for each other.  1 (the default): one workgroup barrier per tile — every wavefront waits
// for the slowest of the sixteen, every tile: a quarter of the kernel's time is that wait.  0 (measured, round 5, not kept): every
// stage publishes how far it is in an LDS counter of its own and waits only for what it reads or overwrites (the rings are
// three to four tiles deep, so a slow tile of one stage is absorbed) — bit-exact, and SLOWER: 1.00 against 0.89 ms for the
// match stage at 256 MiB.  A wait that is already satisfied still costs the LDS round trip of its poll (about 250 cycles with
// sixteen wavefronts on the CU), one or two per stage and tile on the two exchange wavefronts that ARE the critical path;
// the hardware barrier costs no LDS access at all.
#ifndef LFX_M7_BARRIER
#define LFX_M7_BARRIER 1
#endif
constexpr uint32_t NP = 7, NC = 7;             // loading / storing helper wavefronts
// timing experiments only (WRONG answers): bit 0 — no link-record stores, bit 1 — no cd stores (tools/exp/r6_m7_stores.sh)
#ifndef LFX_M7_EXP
#define LFX_M7_EXP 0
#endif
// RUNS (zero-filled and constant regions, BASELINE cfg5's LOWENT: every position repeats the prefix of the one in front of it).
// The lanes of an exchange that hit the same dword are served one after the other — a run is sixty-four of them, on both
// tables (cfg5's match stage took 8.3 ms per GiB against 2.6 for a text).  But a position that repeats its predecessor's prefix
// needs no table at all: its head answer IS the predecessor, and `second` is for it what it was for the run's first lane.  So
// the P stage marks such lanes in the request (bit 30: same prefix as the lane below, bit 31: the lane above continues), only
// a run's first lane (the true exchange) and last lane (its entry must stay in `head`) touch the table, the others exchange
// on dummies, and X2 hands the run's first lane's `second` to the rest by one ds_bpermute.  A tile without a marked lane (nine
// in ten of a text's) takes the paths without any of this: the test is the OR of the tile's fourteen requests.
constexpr uint32_t RQ_SAME_PREV = 1u << 30, RQ_NEXT_SAME = 1u << 31, RQ_ADDR_BITS = 18;
constexpr uint32_t BUCKET_BITS = 14, TAG_BITS = 24 - BUCKET_BITS;
constexpr uint32_t TAG_MASK = (1u << TAG_BITS) - 1;
constexpr uint32_t KEY_MULT = 0x00374ADDu;     // odd, 24 bits: k → k·M mod 2^24 is a bijection (chosen on text)
// entry = (spos << TAG_BITS) | tag, spos = position − base + SPOS0: 0 (an empty slot) is further than any window
constexpr uint32_t SPOS0 = MAX_WINDOW + 1;
constexpr uint32_t UNRES = 0x8000u;            // cd value UNRES + (d2 − 1), d2 in [2, 32768]: unresolved, the walk continues at p − d2

// LDS layout (bytes).  A request is (LDS address of the bucket's head entry) << TAG_BITS | tag; its entry in `second` lies
// SEC_DELTA further (two tables, not pairs: an exchange's 64 random buckets then spread over all 32 banks).  A lane without a
// position exchanges on a slot of its own in dummy 1 (and, SEC_DELTA further, in dummy 2): no exchange wavefront ever tests
// for validity.
constexpr uint32_t OFF_DUMMY = 0;                                   // 32 dwords (lanes l and l + 32 share one: ordered like any bucket)
constexpr uint32_t OFF_TAB = 128;                                   // head: 16 Ki dwords
constexpr uint32_t OFF_DUMMY2 = OFF_TAB + (4u << BUCKET_BITS);      // 32 dwords
constexpr uint32_t OFF_SEC = OFF_DUMMY2 + 128;                      // second: 16 Ki dwords
constexpr uint32_t SEC_DELTA = OFF_SEC - OFF_TAB;
constexpr uint32_t OFF_CTL = OFF_SEC + (4u << BUCKET_BITS);         // progress counters of the four stages (LFX_M7_BARRIER=0)
constexpr uint32_t OFF_RQ = OFF_CTL + 64;                           // 4 tiles of requests
constexpr uint32_t OFF_R1 = OFF_RQ + 4 * TILE * 4;                  // 3 tiles: what the exchange on head returned
constexpr uint32_t OFF_R2 = OFF_R1 + 3 * TILE * 4;                  // 2 tiles: what the exchange on second returned
constexpr uint32_t LDS_BYTES = OFF_R2 + 2 * TILE * 4;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(OFF_DUMMY + SEC_DELTA == OFF_DUMMY2, "a dummy request's second entry is a dummy too");
static_assert(LDS_BYTES < (1u << RQ_ADDR_BITS) && TAG_BITS + RQ_ADDR_BITS <= 30, "LDS addresses fit the request, below its two run bits");
static_assert(SEG_POSITIONS + MAX_WINDOW + SPOS0 + 8 * TILE < (1u << (32 - TAG_BITS)), "segment-relative positions fit the entry");
static_assert(KEY_MULT < (1u << 24) && (KEY_MULT & 1), "24-bit multiplication, bijective");
static_assert(NG == 14 && NP + NC == NG, "operand lists below: two batches of seven; two groups per helper wavefront");
static_assert(TILE % 4 == 0, "a lane's byte phase is the same in every tile");

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// progress counters in LDS (byte address `a`).  A wavefront's LDS instructions execute in order, so a counter written behind a
// stage's data is seen only when the data is; a reader that has seen the counter reads the data behind it.
__device__ __forceinline__ void ctr_set(uint32_t a, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
// until the `n` counters from `a` on (one per wavefront of a stage: each is written by its own wavefront only — a sum would not
// say that the slowest of them has arrived) are all >= v: lane l reads counter l, one LDS read per poll
__device__ __forceinline__ void ctr_wait(uint32_t a, uint32_t n, int v) {
    const uint32_t l = threadIdx.x & 63;
    const uint32_t mine = a + (l < n ? l : 0u) * 4;
    for (;;) {
        int got;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(mine) : "memory");
        if (__ballot(got - v < 0) == 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

// seven exchanges on the head entries, in order; NO wait (lds_wait14 below orders every use of the results)
__device__ __forceinline__ void xchg7(uint32_t (&old)[14], const uint32_t (&addr)[14], const uint32_t (&val)[14], const int o) {
    asm volatile(
        "ds_wrxchg_rtn_b32 %0, %7, %14\n\t"
        "ds_wrxchg_rtn_b32 %1, %8, %15\n\t"
        "ds_wrxchg_rtn_b32 %2, %9, %16\n\t"
        "ds_wrxchg_rtn_b32 %3, %10, %17\n\t"
        "ds_wrxchg_rtn_b32 %4, %11, %18\n\t"
        "ds_wrxchg_rtn_b32 %5, %12, %19\n\t"
        "ds_wrxchg_rtn_b32 %6, %13, %20"
        : "=&v"(old[o]), "=&v"(old[o + 1]), "=&v"(old[o + 2]), "=&v"(old[o + 3]), "=&v"(old[o + 4]), "=&v"(old[o + 5]), "=&v"(old[o + 6])
        : "v"(addr[o]), "v"(addr[o + 1]), "v"(addr[o + 2]), "v"(addr[o + 3]), "v"(addr[o + 4]), "v"(addr[o + 5]), "v"(addr[o + 6]),
          "v"(val[o]), "v"(val[o + 1]), "v"(val[o + 2]), "v"(val[o + 3]), "v"(val[o + 4]), "v"(val[o + 5]), "v"(val[o + 6])
        : "memory");
}
// seven masked exchanges on the second entries (mem = (mem & ~mask) | val, returns the old dword; mask 0 /
// val 0: an ORDERED READ); no wait
__device__ __forceinline__ void mskor7(uint32_t (&old)[14], const uint32_t (&addr)[14], const uint32_t (&mask)[14],
                                       const uint32_t (&val)[14], const int o) {
    asm volatile(
        "ds_mskor_rtn_b32 %0, %7, %14, %21 \n\t"
        "ds_mskor_rtn_b32 %1, %8, %15, %22 \n\t"
        "ds_mskor_rtn_b32 %2, %9, %16, %23 \n\t"
        "ds_mskor_rtn_b32 %3, %10, %17, %24 \n\t"
        "ds_mskor_rtn_b32 %4, %11, %18, %25 \n\t"
        "ds_mskor_rtn_b32 %5, %12, %19, %26 \n\t"
        "ds_mskor_rtn_b32 %6, %13, %20, %27"
        : "=&v"(old[o]), "=&v"(old[o + 1]), "=&v"(old[o + 2]), "=&v"(old[o + 3]), "=&v"(old[o + 4]), "=&v"(old[o + 5]), "=&v"(old[o + 6])
        : "v"(addr[o]), "v"(addr[o + 1]), "v"(addr[o + 2]), "v"(addr[o + 3]), "v"(addr[o + 4]), "v"(addr[o + 5]), "v"(addr[o + 6]),
          "v"(mask[o]), "v"(mask[o + 1]), "v"(mask[o + 2]), "v"(mask[o + 3]), "v"(mask[o + 4]), "v"(mask[o + 5]), "v"(mask[o + 6]),
          "v"(val[o]), "v"(val[o + 1]), "v"(val[o + 2]), "v"(val[o + 3]), "v"(val[o + 4]), "v"(val[o + 5]), "v"(val[o + 6])
        : "memory");
}
// ONE wait for the fourteen results in flight: they are in/out operands, so that no use of them can be scheduled in front
__device__ __forceinline__ void lds_wait14(uint32_t (&v)[14]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                   "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13])
                 :
                 : "memory");
}

}  // namespace m7

// flags[0] |= 1 when an exchange returned a position from the lane's own future (results are then discarded by the host).
// umask: one 64-bit word per 64 positions of every segment (index SegDesc::lnk_base + (p − base) / 64): the unresolved ones.
// DBG: per-wavefront cycle stamps of workgroup 0 (LFX_DEBUG): work / barrier wait.
template <bool DBG>
__global__ __launch_bounds__(m7::THREADS) void lz77_match7_kernel(
    const uint8_t *__restrict__ in, uint64_t in_bytes, const ChunkDesc *__restrict__ chunks,
    const SegDesc *__restrict__ segs, uint32_t window, uint16_t *__restrict__ cd,
    uint32_t *__restrict__ glnk, uint64_t *__restrict__ umask, uint32_t *__restrict__ flags, uint64_t *__restrict__ dbg) {
    using namespace m7;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    uint32_t *rqb = (uint32_t *)(smem + OFF_RQ);
    uint32_t *r1b = (uint32_t *)(smem + OFF_R1);
    uint32_t *r2b = (uint32_t *)(smem + OFF_R2);
    // LDS byte address of the arena for the asm exchanges (taking it from the pointer also makes the array escape)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;

    const SegDesc sg = segs[blockIdx.x];
    const ChunkDesc ch = chunks[sg.chunk];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t n = (uint32_t)ch.len;
    if (ch.flags & CH_LITERALS) return;           // NoCompressionLz77Encoder chunks never come here
    const uint32_t end = (n > 3 ? n : 3) - 3;     // default.rs:75
    const uint32_t q0 = sg.start;                 // first position answered by this segment
    const uint32_t q1 = min(sg.start + sg.len, end);
    if (q0 >= q1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;   // warm-up: inserted and linked, not answered
    const uint32_t base = l0 & ~3u;                               // tile origin
    const int ntiles = (int)((q1 - base + TILE - 1) / TILE);

    // Tiles in flight in iteration i: loads of tile i+1+AHEAD (consumed AHEAD iterations later, so that no wavefront ever
    // waits for HBM), P(i+1) requests, X1(i) exchange on head, X2(i-1) exchange on second, C(i-2) answers.  Every role runs its OWN
    // loop (same trip count, one barrier per trip): inside one loop the role test would be a branch per iteration, and the
    // compiler's wait insertion would have to assume that the other register set's loads were never issued.
    const int i_first = -1 - AHEAD;                                       // tile 0 is loaded by the first iteration
    const int n_iter = (ntiles + 2 - i_first + AHEAD - 1) / AHEAD * AHEAD;   // (stages are predicated: extra iterations do nothing)
    const int i_end = i_first + n_iter;

    // ---- prologue: empty tables; and a sample of the segment — 4 KiB from its first answered position on: a dword of four
    //      equal bytes in one of sixteen or more means constant regions (zero-filled pages, BASELINE cfg5's LOWENT: 95 % of the
    //      positions repeat their predecessor's prefix; a text: 1 in 6000).  A wrong guess costs time, never the answer.
    {
        uint4 *t4 = (uint4 *)smem;
        for (uint32_t i = tid; i < (OFF_RQ >> 4); i += THREADS) t4[i] = make_uint4(0, 0, 0, 0);
    }
    lds_barrier();
    bool runs_mode;
    {
        const uint64_t a0 = (uint64_t)(in + ch.in_off);
        const uint32_t shift = (uint32_t)(a0 & 3);
        const uint32_t last_off = (uint32_t)min((((uint64_t)n + shift + 3) >> 2 << 2) - 4, (uint64_t)0xFFFFFFFCu);
        const uint32_t off = min(((q0 + shift) & ~3u) + tid * 4, last_off);      // (chunk-relative, as the P wavefronts' loads)
        const uint32_t v = *(gptr_u32)((gptr_u8)(a0 & ~3ull) + off);
        const uint64_t eq = __ballot(((v ^ (v >> 8)) & 0xFFFFFFu) == 0);
        uint32_t *cnt = (uint32_t *)(smem + OFF_CTL) + 15;                      // (no room for __syncthreads_count's own LDS)
        if (lane == 0) atomicAdd(cnt, (uint32_t)__popcll(eq));
        lds_barrier();
        runs_mode = *cnt >= THREADS / 16;
        lds_barrier();
        if (tid == 0) *cnt = 0;
        if (tid == 0 && runs_mode) atomicOr(flags, 2u);      // (tells the parse walk which of its instances this call's data wants)
    }
    lds_barrier();

    uint64_t cy_work = 0, cy_wait = 0;
    // barrier mode: one barrier per iteration.  counter mode: wait_for() in front of a stage, done() behind it
    // counters: [0, 7) the P wavefronts, [7] X1, [8, 15) the C wavefronts, [15] X2 — each the number of tiles its wavefront has finished
    const uint32_t c_p = lds0 + OFF_CTL, c_x1 = c_p + 4 * NP, c_c = c_p + 32, c_x2 = c_c + 4 * NC;

    auto sync = [&](uint64_t c0) {
        if (!LFX_M7_BARRIER) { if (DBG) cy_work += clock64() - c0; return; }
        const uint64_t c1 = DBG ? clock64() : 0;
        lds_barrier();
        if (DBG) { cy_work += c1 - c0; cy_wait += clock64() - c1; }
    };
    auto wait_for = [&](uint32_t ctr, uint32_t nctr, int v) {
        if (LFX_M7_BARRIER || v <= 0) return;
        const uint64_t c0 = DBG ? clock64() : 0;
        ctr_wait(ctr, nctr, v);
        if (DBG) { const uint64_t d = clock64() - c0; cy_wait += d; cy_work -= d; }
    };
    if (wave < 2) __builtin_amdgcn_s_setprio(3);

    // The whole pipeline, in two instances: RUNS marks and short-cuts repeated prefixes (see RQ_SAME_PREV), the other knows nothing
    // of them — and is correct on any data, only slower on runs.  A workgroup picks one for its segment from a sample (below).
    auto pipeline = [&](auto runs_tag) {
    constexpr bool RUNS = decltype(runs_tag)::value;
    switch ((LFX_M7_PRIO >> (2 * wave)) & 3u) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
    const uint32_t ridx = (uint32_t)(ROLE_IDX >> (4 * wave)) & 15u;   // which pair of groups a P / C wavefront takes
    if ((ROLE_IS_P >> wave) & 1) {
        // ================================================== P: bytes → prefix → (bucket, tag) → request
        // The chunk's bytes as dwords of the 4-byte aligned allocation in front of them; a lane's dword pair lies at byte
        // offset off_lane + tile * TILE (+ 64 for its second group): one 32-bit add per tile, the rest immediate offsets.
        const uint64_t a0 = (uint64_t)(in + ch.in_off);
        const gptr_u8 srcb = (gptr_u8)(a0 & ~3ull);
        const uint32_t shift = (uint32_t)(a0 & 3);
        const uint32_t hidx = ridx * 128 + lane;              // the first group's position inside the tile
        const uint32_t off_lane = (base + hidx + shift) & ~3u;
        const uint32_t sh8 = (base + hidx + shift) & 3u;      // byte phase inside the dword pair (TILE and 64 are multiples of 4)
        // last dword of the CHUNK (every valid position's pair ends at or in front of it: p + 2 < n)
        const uint32_t last_off = (uint32_t)min((((uint64_t)n + shift + 3) >> 2 << 2) - 4, (uint64_t)0xFFFFFFFCu);
        const uint32_t dummy = (lds0 + OFF_DUMMY + (lane & 31) * 4) << TAG_BITS;
        const uint32_t tab = lds0 + OFF_TAB;
        auto ldw = [&](uint32_t off) { return *(gptr_u32)(srcb + off); };
        // the loads rotate through AHEAD register sets (an iteration consumes the set that was loaded AHEAD iterations ago and
        // reloads it)
        uint32_t ldA[4] = {0, 0, 0, 0}, ldB[4] = {0, 0, 0, 0}, ldC[4] = {0, 0, 0, 0}, ldD[4] = {0, 0, 0, 0};
        auto p_iter = [&](int i, uint32_t (&ld)[4]) {
            const uint64_t c0 = DBG ? clock64() : 0;
            const int tp = i + 1;
            if (tp >= 0 && tp < ntiles) {
                wait_for(c_c, NC, tp - 3);                                // the slot's last readers: C(tp - 4), all of its wavefronts
                const uint32_t t0 = base + (uint32_t)tp * TILE;           // first position of the tile
                const bool interior = t0 >= l0 && t0 + TILE <= q1;        // every position takes part
                uint32_t *rq = rqb + (uint32_t)(tp & 3) * TILE + hidx;
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    const uint32_t key = __builtin_amdgcn_alignbyte(ld[2 * g + 1], ld[2 * g], sh8) & 0xFFFFFFu;
                    uint32_t kk;                                          // low 24 bits: bucket << TAG_BITS | tag
                    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(kk) : "v"(key), "s"(KEY_MULT));   // (full rate; v_mul_lo_u32 is a quarter of it)
                    // (address of the head entry) << TAG_BITS | tag
                    uint32_t r = ((((kk >> TAG_BITS) & ((1u << BUCKET_BITS) - 1)) * 4u + tab) << TAG_BITS) | (kk & TAG_MASK);
                    if (!interior) {
                        const uint32_t p = t0 + hidx + g * 64;
                        r = (p >= l0 && p < q1) ? r : dummy;
                    }
                    // runs: the lane below / above asks for the same bucket and tag — four equal bytes in a row, so a prefix of
                    // three equal bytes comes first (a text: one wavefront in sixteen holds such a lane; the others skip the shuffles)
                    if (RUNS && __ballot(((key ^ (key >> 8)) & 0xFFFFu) == 0)) {
                        const uint32_t below = (uint32_t)__shfl_up((int)r, 1), above = (uint32_t)__shfl_down((int)r, 1);
                        // (a dummy request never equals its neighbour's: lanes l and l + 32 share a dummy)
                        const bool sp = lane > 0 && below == r && r != dummy, ns = lane < 63 && above == r && r != dummy;
                        r |= (sp ? RQ_SAME_PREV : 0u) | (ns ? RQ_NEXT_SAME : 0u);
                    }
                    rq[g * 64] = r;
                }
                if (!LFX_M7_BARRIER) ctr_set(c_p + 4 * ridx, tp + 1);
            }
            // ---- loads of tile i+1+AHEAD, into the registers P just consumed (unconditional; clamped where the chunk ends)
            {
                const uint32_t toff = (uint32_t)(i + 1 + AHEAD) * TILE;
                const uint32_t off = off_lane + toff;
                if (((base + shift) & ~3u) + toff + TILE + 72 <= last_off && last_off >= TILE + 72) {   // (uniform)
                    ld[0] = ldw(off); ld[1] = ldw(off + 4); ld[2] = ldw(off + 64); ld[3] = ldw(off + 68);
                } else {
                    ld[0] = ldw(min(off, last_off)); ld[1] = ldw(min(off + 4, last_off));
                    ld[2] = ldw(min(off + 64, last_off)); ld[3] = ldw(min(off + 68, last_off));
                }
            }
            sync(c0);
        };
        static_assert(AHEAD == 4, "register sets below");
        for (int i = i_first; i < i_end; i += 4) { p_iter(i, ldA); p_iter(i + 1, ldB); p_iter(i + 2, ldC); p_iter(i + 3, ldD); }
    } else if ((ROLE_IS_C >> wave) & 1) {
        // ================================================== C: the two exchanged values → answer, collapsed link
        const uint32_t hidx = ridx * 128 + lane;
        // answers and links by one 32-bit byte offset per tile: cd_c[p] = (cd_c + base)[p - base] at boff, glnk_s[p - base] at 2 boff
        uint8_t *glnk_b = (uint8_t *)(glnk + (uint64_t)sg.lnk_base * 64u);
        uint8_t *cd_b = (uint8_t *)(cd + ch.in_off + base);
        uint64_t *um_s = umask + sg.lnk_base;
        int32_t dmin = 1;                                  // smallest distance seen: <= 0 means a position from the future
        auto c_iter = [&](int i) {
            const uint64_t c0 = DBG ? clock64() : 0;
            const int tc = i - 2;
            if (tc >= 0 && tc < ntiles) {
                wait_for(c_x2, 1, tc + 1);                                // X2(tc) (and with it X1(tc), P(tc))
                const uint32_t t0 = base + (uint32_t)tc * TILE;
                const bool interior = t0 >= q0 && t0 + TILE <= q1;        // every position is answered
                uint32_t rq[2], o1[2], o2[2];
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    rq[g] = rqb[(uint32_t)(tc & 3) * TILE + hidx + g * 64];
                    o1[g] = r1b[(uint32_t)(tc % 3) * TILE + hidx + g * 64];
                    o2[g] = r2b[(uint32_t)(tc & 1) * TILE + hidx + g * 64];
                }
                if (!LFX_M7_BARRIER) {
                    // (the three slots are free once the values are in registers: the wait names them, the add sits behind it)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rq[0]), "+v"(rq[1]), "+v"(o1[0]), "+v"(o1[1]), "+v"(o2[0]), "+v"(o2[1]));
                    ctr_set(c_c + 4 * ridx, tc + 1);
                }
                const uint32_t rel = (uint32_t)tc * TILE + hidx;          // p - base of the first group
                const uint32_t boff = rel * 2;
                uint64_t um[2] = {0, 0};
                bool has_act[2] = {true, true};
#pragma unroll
                for (uint32_t g = 0; g < 2; ++g) {
                    const uint32_t sp = rel + g * 64 + SPOS0;
                    const uint32_t d1 = sp - (o1[g] >> TAG_BITS), d2 = sp - (o2[g] >> TAG_BITS);
                    const bool same1 = ((o1[g] ^ rq[g]) & TAG_MASK) == 0, same2 = ((o2[g] ^ rq[g]) & TAG_MASK) == 0;
                    // the link: the most recent position of the bucket with another prefix
                    const uint32_t dl = same1 ? d2 : d1;
                    const uint32_t lnk = dl <= MAX_WINDOW ? dl : 0u;
                    // the answer (default.rs:81: inclusive window).  d1 > window: the bucket's most recent position is out of
                    // reach, so is everything; d2 >= 2 always (second lies in front of head)
                    uint32_t a2 = same2 ? d2 : d2 + (UNRES - 1);
                    a2 = d2 > window ? 0u : a2;
                    const uint32_t a1 = same1 ? d1 : a2;
                    const uint32_t ans = d1 > window ? 0u : a1;
                    if (interior) {
                        dmin = min(dmin, min((int32_t)d1, (int32_t)d2));
                        if (!(LFX_M7_EXP & 1)) *(uint32_t *)(glnk_b + 2 * boff + g * 256) = lnk | (rq[g] << 16);   // (the request's low 16 bits: the tag and six bits of the bucket — inside a bucket as good as the tag)
                        if (!(LFX_M7_EXP & 2)) *(uint16_t *)(cd_b + boff + g * 128) = (uint16_t)ans;
                        um[g] = __ballot(ans > UNRES);
                    } else {
                        const uint32_t p = t0 + hidx + g * 64;
                        const bool valid = p >= l0 && p < q1, act = valid && p >= q0;
                        if (valid) dmin = min(dmin, min((int32_t)d1, (int32_t)d2));
                        if (valid) *(uint32_t *)(glnk_b + 2 * boff + g * 256) = lnk | (rq[g] << 16);
                        if (act) *(uint16_t *)(cd_b + boff + g * 128) = (uint16_t)ans;
                        um[g] = __ballot(act && ans > UNRES);
                        has_act[g] = __ballot(act) != 0;
                    }
                }
                // the two ballot words of the wavefront's groups are neighbours (and 16-byte aligned: the host keeps lnk_base
                // even): one store by one lane.  A word without an answered position is never read — and never written: behind
                // the segment's last position it may be another segment's.
                if (lane == 0) {
                    if (has_act[0] && has_act[1]) *(ulonglong2 *)(um_s + (rel >> 6)) = make_ulonglong2(um[0], um[1]);
                    else if (has_act[0]) um_s[rel >> 6] = um[0];
                    else if (has_act[1]) um_s[(rel >> 6) + 1] = um[1];
                }
            }
            sync(c0);
        };
        for (int i = i_first; i < i_end; ++i) c_iter(i);
        if (__ballot(dmin <= 0) && lane == 0) atomicOr(flags, 1u);
    } else if (wave == 0) {
        // ================================================== X1(i): head ← (position, tag), in position order; the old entries → r1
        for (int i = i_first; i < i_end; ++i) {
            const uint64_t c0 = DBG ? clock64() : 0;
            if (i >= 0 && i < ntiles) {
                wait_for(c_p, NP, i + 1);                                 // P(i), all of its wavefronts
                wait_for(c_c, NC, i - 2);                                 // the r1 slot's last readers: C(i - 3)
                const uint32_t *rq = rqb + (uint32_t)(i & 3) * TILE + lane;
                uint32_t *r1 = r1b + (uint32_t)(i % 3) * TILE + lane;
                const uint32_t ent0 = ((uint32_t)i * TILE + lane + SPOS0) << TAG_BITS;
                uint32_t q[NG], ad[NG], vl[NG], od[NG];
#pragma unroll
                for (uint32_t s = 0; s < NG; ++s) q[s] = rq[s * 64];
                constexpr bool runs = RUNS;
                const uint32_t dummy_a = lds0 + OFF_DUMMY + (lane & 31) * 4;
#pragma unroll
                for (uint32_t s = 0; s < NG; ++s) {
                    ad[s] = __builtin_amdgcn_ubfe(q[s], TAG_BITS, RQ_ADDR_BITS);
                    vl[s] = (q[s] & TAG_MASK) | (ent0 + ((s * 64u) << TAG_BITS));
                }
                if constexpr (runs) {
                    // the middle of a run touches no table: only its first lane (the true exchange) and its last (its entry stays)
#pragma unroll
                    for (uint32_t s = 0; s < NG; ++s) ad[s] = (q[s] >> 30) == 3u ? dummy_a : ad[s];
                }
                xchg7(od, ad, vl, 0);
                xchg7(od, ad, vl, 7);
                lds_wait14(od);
                if constexpr (runs) {
                    // ... and what a lane behind a run's first would have received: the entry of the position in front of it
#pragma unroll
                    for (uint32_t s = 0; s < NG; ++s) od[s] = (q[s] & RQ_SAME_PREV) ? vl[s] - (1u << TAG_BITS) : od[s];
                }
#pragma unroll
                for (uint32_t s = 0; s < NG; ++s) r1[s * 64] = od[s];
                if (!LFX_M7_BARRIER) ctr_set(c_x1, i + 1);
            }
            sync(c0);
        }
    } else {
        // ================================================== X2(i-1): second ← old head where the tags differ (an ordered read
        // where they are equal) → r2
        for (int i = i_first; i < i_end; ++i) {
            const uint64_t c0 = DBG ? clock64() : 0;
            const int t2 = i - 1;
            if (t2 >= 0 && t2 < ntiles) {
                wait_for(c_x1, 1, t2 + 1);                                // X1(t2)
                wait_for(c_c, NC, t2 - 1);                                // the r2 slot's readers: C(t2 - 2)
                const uint32_t *rq = rqb + (uint32_t)(t2 & 3) * TILE + lane;
                const uint32_t *r1 = r1b + (uint32_t)(t2 % 3) * TILE + lane;
                uint32_t *r2 = r2b + (uint32_t)(t2 & 1) * TILE + lane;
                uint32_t q[NG], o[NG], ad[NG], mk[NG], vl[NG], od[NG];
#pragma unroll
                for (uint32_t s = 0; s < NG; ++s) { q[s] = rq[s * 64]; o[s] = r1[s * 64]; }
                constexpr bool runs = RUNS;
                const uint32_t dummy_a = lds0 + OFF_DUMMY2 + (lane & 31) * 4;
#pragma unroll
                for (uint32_t s = 0; s < NG; ++s) {
                    const bool differ = ((q[s] ^ o[s]) & TAG_MASK) != 0;       // (never for a lane behind a run's first)
                    ad[s] = __builtin_amdgcn_ubfe(q[s], TAG_BITS, RQ_ADDR_BITS) + SEC_DELTA;
                    mk[s] = differ ? 0xFFFFFFFFu : 0u;
                    vl[s] = differ ? o[s] : 0u;
                }
                if constexpr (runs) {
#pragma unroll
                    for (uint32_t s = 0; s < NG; ++s) ad[s] = (q[s] & RQ_SAME_PREV) ? dummy_a : ad[s];
                }
                mskor7(od, ad, mk, vl, 0);
                mskor7(od, ad, mk, vl, 7);
                lds_wait14(od);
                if constexpr (runs) {
                    // `second` behind a lane's action: its old head where it wrote, what it read otherwise; a lane behind a run's
                    // first takes the first's (nothing between them touches the bucket)
#pragma unroll
                    for (uint32_t s = 0; s < NG; ++s) {
                        const bool sp = (q[s] & RQ_SAME_PREV) != 0;
                        const uint64_t heads = __ballot(!sp) & ((2ull << lane) - 1ull);      // (lane 0 never is behind a first)
                        const uint32_t h = 63u - (uint32_t)__builtin_clzll(heads);
                        const uint32_t behind = mk[s] ? o[s] : od[s];
                        const uint32_t fromhead = (uint32_t)__shfl((int)behind, (int)h);
                        od[s] = sp ? fromhead : od[s];
                    }
                }
#pragma unroll
                for (uint32_t s = 0; s < NG; ++s) r2[s * 64] = od[s];
                if (!LFX_M7_BARRIER) ctr_set(c_x2, t2 + 1);
            }
            sync(c0);
        }
    }
    };
    if (runs_mode) pipeline(std::true_type{}); else pipeline(std::false_type{});
    if (DBG && dbg && blockIdx.x == 0 && lane == 0) {
        uint64_t *d = dbg + wave * 8;
        d[0] = cy_work; d[1] = cy_wait; d[2] = 0; d[3] = 0; d[4] = 0; d[5] = (uint64_t)ntiles; d[6] = 0; d[7] = 0;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The unresolved positions (three or more prefixes alternate in the bucket inside one window; 1 % of a text, half of random
// bytes).  lz77_compact7_kernel: the ballot words of a slab of 8192 positions → entries of the segment's list (one global
// add per workgroup).  lz77_resolve7_kernel: follow the duplicate-collapsed links from p - d2 on until the prefix is found
// or the window ends.
namespace r7 {
constexpr uint32_t SLAB_WORDS = 128;                       // 8192 positions per compaction workgroup
constexpr uint32_t SLABS = SEG_POSITIONS / 64 / SLAB_WORDS;
constexpr uint32_t THREADS = 256, WGS = 2, NS = 4;         // resolver: 2 x 256 lanes per segment, NS walks each, stride over its list
}  // namespace r7

__global__ __launch_bounds__(r7::SLAB_WORDS) void lz77_compact7_kernel(
    const ChunkDesc *__restrict__ chunks, const SegDesc *__restrict__ segs, const uint64_t *__restrict__ umask,
    uint32_t *__restrict__ ulist, uint32_t *__restrict__ ucount) {
    using namespace r7;
    __shared__ uint32_t wsum[2], gbase;
    const uint32_t seg = blockIdx.x / (SLABS + 1), slab = blockIdx.x % (SLABS + 1);   // (+1: the first word starts up to 3 positions in front of q0)
    const SegDesc sg = segs[seg];
    if (slab * (SLAB_WORDS * 64) >= sg.len + 64) return;
    const ChunkDesc ch = chunks[sg.chunk];
    const uint32_t n = (uint32_t)ch.len;
    const uint32_t end = (n > 3 ? n : 3) - 3;
    const uint32_t q0 = sg.start, q1 = min(sg.start + sg.len, end);
    if (q0 >= q1) return;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;
    const uint32_t base = l0 & ~3u;
    // word w covers positions [base + 64 w, base + 64 w + 64); the slab's words start at the word that holds q0
    const uint32_t w = (q0 - base) / 64 + slab * SLAB_WORDS + threadIdx.x;
    const uint32_t p0 = base + w * 64;
    uint64_t m = 0;
    if (p0 < q1) m = umask[sg.lnk_base + w];               // (every word with an answered position was written)
    // positions in front of q0 / behind q1 are never marked (the kernel above ballots `act`), but a word may straddle the
    // slab's first position only in the segment's first word: nothing to mask
    if (base + (w - threadIdx.x) * 64 >= q1) return;         // the whole slab lies behind the segment (uniform)
    const uint32_t c = (uint32_t)__popcll(m);
    // exclusive scan over the 128 lanes (two wavefronts)
    uint32_t x = c;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up((int)x, o); if ((threadIdx.x & 63) >= (uint32_t)o) x += y; }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    __syncthreads();
    const uint32_t total = wsum[0] + wsum[1];
    if (total == 0) return;
    if (threadIdx.x == 0) gbase = atomicAdd(&ucount[seg], total);
    __syncthreads();
    uint32_t at = gbase + (x - c) + ((threadIdx.x >> 6) ? wsum[0] : 0u);
    uint32_t *ul = ulist + ch.in_off + sg.start;
    while (m) {
        const uint32_t b = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        ul[at++] = p0 + b;
    }
}

// A lane strides over its segment's list with NS walks in flight at once, each on its own: every trip of the loop is ONE memory
// round trip for every walk — the entry's position, or its answer-so-far and its tag, or a hop (the record of the position
// reached: its tag and its link) — so that a long walk holds nobody up, and the kernel's time is the round trips of its longest
// lane, not their sum over the entries.  (What it costs is its scattered sectors from HBM: ending every walk after one hop —
// LFX_R7_CAP=1, wrong answers — saves 0.07 of 0.24 ms.)
__global__ __launch_bounds__(r7::THREADS) void lz77_resolve7_kernel(
    const ChunkDesc *__restrict__ chunks, const SegDesc *__restrict__ segs, uint32_t window, uint16_t *__restrict__ cd,
    const uint32_t *__restrict__ glnk, const uint32_t *__restrict__ ulist, const uint32_t *__restrict__ ucount,
    uint32_t hop_cap /* diagnostics: 0 = none */) {
    using namespace r7;
    const uint32_t seg = blockIdx.x / WGS, part = blockIdx.x % WGS;
    const uint32_t total = ucount[seg];
    if (part * THREADS >= total) return;
    const SegDesc sg = segs[seg];
    const ChunkDesc ch = chunks[sg.chunk];
    const uint32_t q0 = sg.start;
    const uint32_t l0 = q0 > MAX_WINDOW ? q0 - MAX_WINDOW : 0;
    const uint32_t base = l0 & ~3u;
    uint16_t *cd_c = cd + ch.in_off;
    const uint32_t *glnk_s = glnk + (uint64_t)sg.lnk_base * 64u;
    const uint32_t *ulist_s = ulist + ch.in_off + sg.start;
    constexpr uint32_t STRIDE = WGS * THREADS;             // walks j of all lanes: entries j * STRIDE + lane, + NS * STRIDE, ...
    // walk state: 0 fetch the entry, 1 fetch its tag and where its walk starts, 2 hop, 3 no more entries
    uint32_t state[NS], i[NS], p[NS], tag[NS], r[NS], dist[NS], hops[NS];
    bool first[NS];
#pragma unroll
    for (uint32_t j = 0; j < NS; ++j) {
        i[j] = part * THREADS + threadIdx.x + j * STRIDE;
        state[j] = i[j] < total ? 0u : 3u;
        p[j] = tag[j] = r[j] = dist[j] = hops[j] = 0;
        first[j] = false;
    }
    for (;;) {
        bool any = false;
#pragma unroll
        for (uint32_t j = 0; j < NS; ++j) any |= state[j] != 3u;
        if (!__ballot(any)) break;
        // ---- loads of this trip: only what a walk's state needs (a scattered load costs by its active lanes), all of them
        //      issued before the first is used
        uint32_t e[NS], c[NS], rec[NS];
#pragma unroll
        for (uint32_t j = 0; j < NS; ++j) {
            const bool s0 = state[j] == 0, s1 = state[j] == 1, s2 = state[j] == 2;
            e[j] = c[j] = rec[j] = 0;
            if (s0) e[j] = ulist_s[i[j]];
            if (s1) c[j] = cd_c[p[j]];
            if (s1 || s2) rec[j] = glnk_s[(s1 ? p[j] : r[j]) - base];     // (state 2: the position reached)
        }
        // (the loaded values are pinned here: the compiler otherwise sinks a value's first use into the branch that loaded it,
        //  and the trip becomes a round trip per state and walk instead of one)
#pragma unroll
        for (uint32_t j = 0; j < NS; ++j) asm volatile("" : "+v"(e[j]), "+v"(c[j]), "+v"(rec[j]));
        // ---- uses
#pragma unroll
        for (uint32_t j = 0; j < NS; ++j) {
            if (state[j] == 0) { p[j] = e[j]; state[j] = 1; }
            else if (state[j] == 1) {
                tag[j] = rec[j] >> 16;
                dist[j] = c[j] - m7::UNRES + 1;              // d2: the position of `second`, known to carry another prefix
                r[j] = p[j] - dist[j];
                first[j] = true;
                hops[j] = 0;
                state[j] = 2;
            } else if (state[j] == 2) {
                bool done = false;
                uint32_t ans = 0;
                const uint32_t l = rec[j] & 0xFFFFu;
                if (!first[j] && (rec[j] >> 16) == tag[j]) { done = true; ans = dist[j]; }
                else {
                    // (a link never reaches in front of l0, the first inserted position)
                    dist[j] += l;
                    if (l == 0 || dist[j] > window) done = true;       // default.rs:81 (inclusive window)
                    else r[j] -= l;
                    if (hop_cap && ++hops[j] >= hop_cap) done = true;   // (LFX_R7_CAP: wrong answers, timing only)
                }
                first[j] = false;
                if (done) {
                    cd_c[p[j]] = (uint16_t)ans;
                    i[j] += NS * STRIDE;
                    state[j] = i[j] < total ? 0u : 3u;
                }
            }
        }
    }
}

// The candidate kernel for `nsegs` segments (the caller passes `segs` at the first of them)
int launch_match7(hipStream_t st, const uint8_t *in, uint64_t in_bytes, const ChunkDesc *chunks, const SegDesc *segs,
                  uint32_t nsegs, uint32_t window, uint16_t *cd, uint32_t *glnk, uint64_t *umask, uint32_t *flags, uint64_t *dbg) {
    if (nsegs == 0) return 0;
    if (dbg)
        hipLaunchKernelGGL(lz77_match7_kernel<true>, dim3(nsegs), dim3(m7::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, glnk, umask, flags, dbg);
    else
        hipLaunchKernelGGL(lz77_match7_kernel<false>, dim3(nsegs), dim3(m7::THREADS), 0, st, in, in_bytes, chunks, segs, window,
                           cd, glnk, umask, flags, dbg);
    const hipError_t e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

// ... and what it left open: ballot words → lists → walks.  ucount (one counter per segment, at the first of these segments)
// must be zero.
int launch_resolve7(hipStream_t st, const ChunkDesc *chunks, const SegDesc *segs, uint32_t nsegs, uint32_t window, uint16_t *cd,
                    const uint32_t *glnk, const uint64_t *umask, uint32_t *ulist, uint32_t *ucount, uint32_t hop_cap) {
    if (nsegs == 0) return 0;
    hipLaunchKernelGGL(lz77_compact7_kernel, dim3(nsegs * (r7::SLABS + 1)), dim3(r7::SLAB_WORDS), 0, st, chunks, segs, umask, ulist,
                       ucount);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return (int)e_;
    hipLaunchKernelGGL(lz77_resolve7_kernel, dim3(nsegs * r7::WGS), dim3(r7::THREADS), 0, st, chunks, segs, window, cd, glnk, ulist,
                       ucount, hop_cap);
    e_ = hipGetLastError();
    return e_ != hipSuccess ? (int)e_ : 0;
}

}  // namespace lfx
