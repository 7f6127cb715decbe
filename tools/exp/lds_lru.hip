// Micro-benchmark (GPU box), round 5: what an LDS read-modify-write WITH return costs as a function of the ACTIVE LANES,
// and what two wavefronts of one workgroup get when both issue them (is the atomic path shared?) — the cost model of
// lfx_match7.hip (two-level bucket LRU: one full-wave exchange + one partial-wave exchange per 64 positions).
//   A: one wavefront, batches of 15 ds_mskor_rtn_b32, k active lanes (exec-masked), random dwords of a 64 KiB table
//   B: two / four wavefronts of one workgroup, each with its own table, each issuing the same batches
//   C: wave 0 with 64 lanes, wave 1 with 10 lanes
//   D: which lane's data a plain ds_write_b32 leaves when several lanes of ONE instruction write the same dword
//   E: ds_wrxchg_rtn_b32, same-dword lanes served in ascending lane order? (as mskor_test.hip for ds_mskor_rtn_b32)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define B15(OP3)                                                                                                      \
    asm volatile(OP3(0, 15, 30) OP3(1, 16, 31) OP3(2, 17, 32) OP3(3, 18, 33) OP3(4, 19, 34) OP3(5, 20, 35) OP3(6, 21, 36) \
                 OP3(7, 22, 37) OP3(8, 23, 38) OP3(9, 24, 39) OP3(10, 25, 40) OP3(11, 26, 41) OP3(12, 27, 42)         \
                 OP3(13, 28, 43) OP3(14, 29, 44) "s_waitcnt lgkmcnt(0)"                                              \
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),   \
                   "=&v"(o[8]), "=&v"(o[9]), "=&v"(o[10]), "=&v"(o[11]), "=&v"(o[12]), "=&v"(o[13]), "=&v"(o[14])    \
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),    \
                   "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), \
                   "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), \
                   "v"(v[13]), "v"(v[14]), "v"(m)                                                                   \
                 : "memory")
#define MSKOR(O, A, V) "ds_mskor_rtn_b32 %" #O ", %" #A ", %45, %" #V "\n\t"
#define XCHG(O, A, V) "ds_wrxchg_rtn_b32 %" #O ", %" #A ", %" #V "\n\t"
#define READ(O, A, V) "ds_read_b32 %" #O ", %" #A "\n\t"

// MODE 0 mskor (mask ~0), 1 wrxchg, 2 plain read, 3 mskor with mask 0 (pure ordered read)
template <int MODE>
__global__ __launch_bounds__(256) void k_cost(uint32_t *out, uint32_t iters, uint64_t *cyc, uint32_t k0, uint32_t k1) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tab[];          // 16 Ki dwords per wavefront
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *mine = tab + wave * 8192;
    for (uint32_t i = lane; i < 8192; i += 64) mine[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)mine;
    uint32_t x = threadIdx.x * 2654435761u + 12345u, acc = 0;
    const uint32_t k = wave == 0 ? k0 : k1;
    const uint32_t m = MODE == 3 ? 0u : 0xFFFFFFFFu;
    const uint64_t c0 = clock64();
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t a[15], v[15], o[15];
#pragma unroll
        for (int j = 0; j < 15; j++) {
            x = x * 1664525u + 1013904223u;
            a[j] = base + (((x >> 8) & 8191u) << 2);
            v[j] = MODE == 3 ? 0u : x;
            o[j] = 0;
        }
        if (lane < k) {
            if (MODE == 0 || MODE == 3) B15(MSKOR);
            if (MODE == 1) B15(XCHG);
            if (MODE == 2) B15(READ);
        }
#pragma unroll
        for (int j = 0; j < 15; j++) acc ^= o[j];
    }
    const uint64_t c1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = c1 - c0;
}

template <int MODE>
void cost(const char *name, uint32_t waves, uint32_t k0, uint32_t k1, uint32_t *d_o, uint64_t *d_c) {
    const uint32_t iters = 400;
    hipLaunchKernelGGL(k_cost<MODE>, dim3(1), dim3(64 * waves), waves * 32768, 0, d_o, iters, d_c, k0, k1);
    uint64_t cyc[4] = {0, 0, 0, 0};
    (void)hipMemcpy(cyc, d_c, 32, hipMemcpyDeviceToHost);
    printf("%-18s waves=%u lanes(w0)=%2u lanes(others)=%2u : cycles per wave-instruction w0=%.1f w1=%.1f\n", name, waves, k0,
           k1, (double)cyc[0] / iters / 15, waves > 1 ? (double)cyc[1] / iters / 15 : 0.0);
}

// D + E: conflict semantics.  bucket[] gives every lane of every round a dword; round r lane l writes r*64+l+1.
__global__ __launch_bounds__(64) void k_order(const uint32_t *bucket, uint32_t nb, uint32_t rounds, uint32_t *got_x,
                                              uint32_t *got_w) {
    __shared__ uint32_t tx[1024], tw[1024];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 1024; i += 64) tx[i] = tw[i] = 0;
    __syncthreads();
    const uint32_t bx = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)tx;
    const uint32_t bw = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)tw;
    for (uint32_t r = 0; r < rounds; r++) {
        const uint32_t b = bucket[r * 64 + lane] % nb, val = r * 64 + lane + 1;
        uint32_t old, after;
        asm volatile("ds_wrxchg_rtn_b32 %0, %2, %3\n\t"
                     "ds_write_b32 %4, %3\n\t"
                     "ds_read_b32 %1, %4\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(old), "=&v"(after)
                     : "v"(bx + b * 4), "v"(val), "v"(bw + b * 4)
                     : "memory");
        got_x[r * 64 + lane] = old;          // E: the previous writer of the dword
        got_w[r * 64 + lane] = after;        // D: what the plain write of this instruction left
    }
}

int main() {
    uint32_t *d_o; uint64_t *d_c;
    (void)hipMalloc(&d_o, 4096 * 4); (void)hipMalloc(&d_c, 64);
    (void)hipFuncSetAttribute((const void *)k_cost<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_cost<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_cost<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_cost<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (uint32_t k : {64u, 48u, 32u, 16u, 8u, 4u, 1u}) cost<0>("A mskor_rtn", 1, k, 0, d_o, d_c);
    for (uint32_t k : {64u, 16u, 1u}) cost<1>("A wrxchg_rtn", 1, k, 0, d_o, d_c);
    for (uint32_t k : {64u, 16u, 1u}) cost<2>("A read_b32", 1, k, 0, d_o, d_c);
    for (uint32_t k : {64u, 16u}) cost<3>("A mskor mask0", 1, k, 0, d_o, d_c);
    cost<0>("B mskor_rtn", 2, 64, 64, d_o, d_c);
    cost<0>("B mskor_rtn", 4, 64, 64, d_o, d_c);
    cost<1>("B wrxchg_rtn", 2, 64, 64, d_o, d_c);
    cost<0>("C mskor_rtn", 2, 64, 10, d_o, d_c);
    cost<0>("C mskor_rtn", 2, 64, 20, d_o, d_c);
    cost<2>("B read_b32", 4, 64, 64, d_o, d_c);

    const uint32_t rounds = 3000;
    std::vector<uint32_t> hb(rounds * 64);
    uint64_t s = 88172645463325252ull;
    for (auto &v : hb) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 20); }
    uint32_t *d_b, *d_x, *d_w;
    (void)hipMalloc(&d_b, hb.size() * 4); (void)hipMalloc(&d_x, hb.size() * 4); (void)hipMalloc(&d_w, hb.size() * 4);
    (void)hipMemcpy(d_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    for (uint32_t nb : {1u, 2u, 3u, 8u, 64u, 1024u}) {
        hipLaunchKernelGGL(k_order, dim3(1), dim3(64), 0, 0, d_b, nb, rounds, d_x, d_w);
        std::vector<uint32_t> gx(hb.size()), gw(hb.size());
        (void)hipMemcpy(gx.data(), d_x, gx.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(gw.data(), d_w, gw.size() * 4, hipMemcpyDeviceToHost);
        std::vector<uint32_t> model(1024, 0);
        uint64_t bad_x = 0, bad_w = 0;
        for (uint32_t r = 0; r < rounds; r++) {
            uint32_t last[1024];
            for (uint32_t l = 0; l < 64; l++) {
                const uint32_t b = hb[r * 64 + l] % nb;
                if (gx[r * 64 + l] != model[b]) bad_x++;
                model[b] = r * 64 + l + 1;
                last[b] = r * 64 + l + 1;           // highest lane of the round that wrote b
            }
            for (uint32_t l = 0; l < 64; l++)
                if (gw[r * 64 + l] != last[hb[r * 64 + l] % nb]) bad_w++;
        }
        printf("E wrxchg order nbuckets=%4u: violations=%llu of %u | D plain write, highest lane wins: violations=%llu\n", nb,
               (unsigned long long)bad_x, rounds * 64, (unsigned long long)bad_w);
    }
    return 0;
}
