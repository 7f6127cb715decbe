// CPU: Planner::write_repeat(n, count) leaves exactly the plan and the state of `count` calls of write(n)
// (libflate_amd/csrc/lfx_plan.h; the reference's per-write rules: encode.rs:277-303, 405-425, default.rs:60-68).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

#include "../../libflate_amd/csrc/lfx_plan.h"

using namespace lfx;

static bool same(const Plan &a, const Plan &b) {
    if (a.chunks.size() != b.chunks.size() || a.blocks.size() != b.blocks.size()) return false;
    if (a.n_codes_cap != b.n_codes_cap || a.n_tiles != b.n_tiles || a.n_vis != b.n_vis || a.n_segs != b.n_segs) return false;
    for (size_t i = 0; i < a.chunks.size(); i++) if (memcmp(&a.chunks[i], &b.chunks[i], sizeof(ChunkDesc))) return false;
    for (size_t i = 0; i < a.blocks.size(); i++) if (memcmp(&a.blocks[i], &b.blocks[i], sizeof(BlockDesc))) return false;
    return true;
}

int main() {
    std::mt19937_64 rng(7);
    int cases = 0;
    for (int it = 0; it < 20000; it++) {
        PlanOpts o;
        const uint64_t bs[] = {1, 100, 4096, 65535, 65536, 300 << 10, 1 << 20, 4 << 20};
        o.block_size = bs[rng() % 8];
        o.dynamic_huffman = rng() & 1;
        o.no_compression = (rng() % 4) == 0;
        o.lz77_kind = (rng() % 5) == 0;
        const uint32_t ws[] = {256, 1024, 32768};
        o.window_size = ws[rng() % 3];
        o.zlib_sync = (rng() % 7) == 0;
        Planner a(o), b(o);
        // a few segments: repeated writes, single writes, flushes — the same calls on both, b through write_repeat
        for (int seg = 0; seg < 6; seg++) {
            const int kind = (int)(rng() % 4);
            if (kind == 0) { a.flush(); b.flush(); continue; }
            const uint64_t ns[] = {0, 1, 7, 100, 1000, 8192, 8191, 65536, 262144, 262145, 1 << 20, 3 << 20};
            const uint64_t n = ns[rng() % 12];
            const uint64_t count = kind == 1 ? 1 : rng() % (n > 100000 ? 40 : 3000);
            for (uint64_t k = 0; k < count; k++) a.write(n);
            b.write_repeat(n, count);
            if (a.cursor() != b.cursor() || a.closed_bytes() != b.closed_bytes()) { printf("state differs: it %d seg %d\n", it, seg); return 1; }
        }
        Plan pa = a.finish(), pb = b.finish();
        if (!same(pa, pb)) { printf("plans differ: it %d\n", it); return 1; }
        cases++;
    }
    printf("plan_repeat ok: %d cases\n", cases);
    return 0;
}
