// Micro-test (GPU box): semantics of ds_mskor_rtn_b32 as a 16-bit exchange inside a 32-bit LDS word.
//  * lanes of ONE instruction that hit the same 16-bit field must be served in ascending lane order
//    (each lane then receives the value of the nearest lower lane with the same field);
//  * instructions of one wavefront are served in issue order;
//  * the other half of the word is untouched.
// Prints the number of violations per conflict level and the cycles of 15 back-to-back operations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ __launch_bounds__(64) void k_order(const uint32_t *bucket, uint32_t nbuckets, uint32_t rounds,
                                              uint32_t *got, uint64_t *cyc) {
    __shared__ uint32_t tab[8192];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 8192; i += 64) tab[i] = 0;
    __syncthreads();
    // the asm below addresses the array by its LDS byte offset: take it from the pointer (and make the array escape,
    // otherwise the compiler drops an array that is only ever zero-filled and the accesses fall outside the allocation)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)tab;
    uint64_t t = 0;
    for (uint32_t r = 0; r < rounds; r += 15) {
        uint32_t old[15], addr[15], mask[15], val[15];
#pragma unroll
        for (int j = 0; j < 15; j++) {
            const uint32_t b = bucket[(r + j) * 64 + lane] % nbuckets;      // 16-bit field index
            addr[j] = lds_base + (b >> 1) * 4;
            mask[j] = (b & 1) ? 0xFFFF0000u : 0x0000FFFFu;
            const uint32_t v = ((r + j) * 64 + lane + 1) & 0xFFFFu;          // "position + 1" mod 65536
            val[j] = (b & 1) ? v << 16 : v;
        }
        const uint64_t c0 = clock64();
        asm volatile(
            "ds_mskor_rtn_b32 %0, %15, %30, %45\n\t"
            "ds_mskor_rtn_b32 %1, %16, %31, %46\n\t"
            "ds_mskor_rtn_b32 %2, %17, %32, %47\n\t"
            "ds_mskor_rtn_b32 %3, %18, %33, %48\n\t"
            "ds_mskor_rtn_b32 %4, %19, %34, %49\n\t"
            "ds_mskor_rtn_b32 %5, %20, %35, %50\n\t"
            "ds_mskor_rtn_b32 %6, %21, %36, %51\n\t"
            "ds_mskor_rtn_b32 %7, %22, %37, %52\n\t"
            "ds_mskor_rtn_b32 %8, %23, %38, %53\n\t"
            "ds_mskor_rtn_b32 %9, %24, %39, %54\n\t"
            "ds_mskor_rtn_b32 %10, %25, %40, %55\n\t"
            "ds_mskor_rtn_b32 %11, %26, %41, %56\n\t"
            "ds_mskor_rtn_b32 %12, %27, %42, %57\n\t"
            "ds_mskor_rtn_b32 %13, %28, %43, %58\n\t"
            "ds_mskor_rtn_b32 %14, %29, %44, %59\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(old[0]), "=&v"(old[1]), "=&v"(old[2]), "=&v"(old[3]), "=&v"(old[4]), "=&v"(old[5]), "=&v"(old[6]),
              "=&v"(old[7]), "=&v"(old[8]), "=&v"(old[9]), "=&v"(old[10]), "=&v"(old[11]), "=&v"(old[12]), "=&v"(old[13]),
              "=&v"(old[14])
            : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
              "v"(addr[8]), "v"(addr[9]), "v"(addr[10]), "v"(addr[11]), "v"(addr[12]), "v"(addr[13]), "v"(addr[14]),
              "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]),
              "v"(mask[8]), "v"(mask[9]), "v"(mask[10]), "v"(mask[11]), "v"(mask[12]), "v"(mask[13]), "v"(mask[14]),
              "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]),
              "v"(val[8]), "v"(val[9]), "v"(val[10]), "v"(val[11]), "v"(val[12]), "v"(val[13]), "v"(val[14])
            : "memory");
        t += clock64() - c0;
#pragma unroll
        for (int j = 0; j < 15; j++) {
            const uint32_t b = bucket[(r + j) * 64 + lane] % nbuckets;
            got[(r + j) * 64 + lane] = (b & 1) ? old[j] >> 16 : old[j] & 0xFFFFu;
        }
    }
    __syncthreads();
    got[rounds * 64 + lane] = tab[lane];
    if (lane == 0) *cyc = t;
}

// plain LDS read latency (dependent chain) and gather throughput, for the design worksheet
__global__ __launch_bounds__(1024) void k_lat(uint32_t *out, uint32_t iters, uint64_t *cyc) {
    __shared__ uint32_t tab[16384];
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = (i * 2654435761u) >> 18;
    __syncthreads();
    uint32_t x = threadIdx.x;
    const uint64_t c0 = clock64();
    for (uint32_t i = 0; i < iters; i++) x = tab[x & 16383];
    const uint64_t c1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

int main() {
    const uint32_t rounds = 15 * 200;
    std::vector<uint32_t> hb(rounds * 64);
    uint64_t s = 88172645463325252ull;
    for (auto &v : hb) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint32_t)(s >> 20); }
    uint32_t *d_b, *d_got; uint64_t *d_c;
    hipMalloc(&d_b, hb.size() * 4); hipMalloc(&d_got, hb.size() * 4 + 256); hipMalloc(&d_c, 8);
    hipMemcpy(d_b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    const uint32_t levels[] = {1, 2, 3, 8, 64, 1024, 16384};
    for (uint32_t nb : levels) {
        hipLaunchKernelGGL(k_order, dim3(1), dim3(64), 0, 0, d_b, nb, rounds, d_got, d_c);
        std::vector<uint32_t> got(hb.size()); uint64_t cyc = 0;
        hipMemcpy(got.data(), d_got, got.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost);
        std::vector<uint32_t> model(16384, 0);
        uint64_t bad = 0;
        for (uint32_t i = 0; i < rounds * 64; i++) {
            const uint32_t b = hb[i] % nb;
            if (got[i] != model[b]) bad++;
            model[b] = (i + 1) & 0xFFFFu;
        }
        printf("mskor nbuckets=%5u: violations=%llu of %u, cycles per 15 ops=%.0f\n", nb, (unsigned long long)bad, rounds * 64,
               (double)cyc / (rounds / 15));
    }
    uint32_t *d_o; hipMalloc(&d_o, 1024 * 256 * 4);
    for (int wg : {1, 256}) for (int th : {64, 256, 1024}) {
        hipLaunchKernelGGL(k_lat, dim3(wg), dim3(th), 0, 0, d_o, 2000, d_c);
        uint64_t cyc = 0; hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost);
        printf("lds dependent gather: grid=%d threads=%d: %.1f cycles per hop\n", wg, th, (double)cyc / 2000);
    }
    return 0;
}
