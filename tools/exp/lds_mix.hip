// Micro-benchmark (GPU box): how much does one wavefront's stream of returning LDS atomics (ds_mskor_rtn_b32)
// slow the gathers of the other 15 wavefronts of the workgroup?  Plus the cost of gather flavours.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k_mix(uint32_t *out, uint32_t iters, uint64_t *cyc, int atomics_on) {
    __shared__ uint32_t tab[16384];
    __shared__ uint32_t heads[8192];
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = (i * 2654435761u) >> 18;
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) heads[i] = 0;
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t x = threadIdx.x * 77u;
    const uint64_t c0 = clock64();
    if (wave == 0 && atomics_on) {
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)heads;
        for (uint32_t i = 0; i < iters; i++) {
            const uint32_t h = (x * 2654435761u) >> 18;   // 14 bits
            const uint32_t addr = lds_base + (h >> 1) * 4, mask = (h & 1) ? 0xFFFF0000u : 0xFFFFu, val = (h & 1) ? (i << 16) : (i & 0xFFFF);
            uint32_t old;
            asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(old) : "v"(addr), "v"(mask), "v"(val) : "memory");
            x = x * 1664525u + 1013904223u + (old & 1);
        }
    } else if (wave != 0) {
        const uint8_t *tb = (const uint8_t *)tab;
        const uint16_t *th = (const uint16_t *)tab;
        for (uint32_t i = 0; i < iters; i++) {
            if (MODE == 0) x = tab[x & 16383];                                   // aligned b32 gather
            if (MODE == 1) x = th[x & 32767] * 2654435761u >> 7;                 // u16 gather
            if (MODE == 2) { uint32_t a = x & 16382; x = tab[a] ^ tab[a + 1]; }  // read2 gather
            if (MODE == 3) { uint32_t a = x & 65531; x = *(const uint32_t *)(tb + a); }   // unaligned b32 gather
            if (MODE == 4) { uint32_t a = (x & 16380) ; const uint64_t v = *(const uint64_t *)(tab + (a & ~1u)); x = (uint32_t)v ^ (uint32_t)(v >> 32); }  // aligned b64 gather
            if (MODE == 5) x = tab[(i * 64 + lane + (x & 1)) & 16383];          // linear b32
        }
    }
    const uint64_t c1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = x;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = c1 - c0;
}

template <int MODE>
void run(const char *name, uint32_t *d_o, uint64_t *d_c) {
    for (int at = 0; at < 2; at++) {
        hipLaunchKernelGGL(k_mix<MODE>, dim3(256), dim3(1024), 0, 0, d_o, 2000, d_c, at);
        uint64_t cyc[16];
        (void)hipMemcpy(cyc, d_c, sizeof cyc, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int w = 1; w < 16; w++) avg += (double)cyc[w] / 2000 / 15;
        printf("%-22s atomics=%d: wave0 %.1f cyc/op, waves1-15 %.1f cyc/hop\n", name, at, (double)cyc[0] / 2000, avg);
    }
}

int main() {
    uint32_t *d_o; uint64_t *d_c;
    (void)hipMalloc(&d_o, 1024 * 256 * 4); (void)hipMalloc(&d_c, 16 * 8);
    run<0>("b32 gather", d_o, d_c);
    run<1>("u16 gather", d_o, d_c);
    run<2>("read2_b32 gather", d_o, d_c);
    run<3>("unaligned b32 gather", d_o, d_c);
    run<4>("b64 gather", d_o, d_c);
    run<5>("b32 linear", d_o, d_c);
    return 0;
}
