// round 6 microbenchmark: what a persistent grid of one-wavefront workgroups pays for (a) being launched, (b) a dependent
// device-scope atomic with return, alone and under contention, (c) a dependent global load.  Build: hipcc --offload-arch=gfx950 -O3
// Usage: atomic_lat            (prints microseconds)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(64) void k_empty(uint32_t *out) {
    __shared__ uint32_t lds[3200];            // 12.8 KB: as find_blocks_stage2
    lds[threadIdx.x] = blockIdx.x;
    __syncthreads();
    if (lds[63 - threadIdx.x] == 0xFFFFFFFFu) out[0] = 1;
}
__global__ __launch_bounds__(64) void k_atomic_chain(uint32_t *ctr, uint32_t stride_words, uint32_t groups, uint32_t iters, uint32_t *out) {
    uint32_t *c = ctr + (blockIdx.x % groups) * stride_words;
    uint32_t v = 0, acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        if (threadIdx.x == 0) v = atomicAdd(c + (v & 0u), 1u);      // dependent on the previous answer
        v = __shfl(v, 0);
        acc += v;
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(64) void k_load_chain(const uint32_t *buf, uint32_t n, uint32_t iters, uint32_t *out) {
    uint32_t idx = (blockIdx.x * 977u + threadIdx.x * 64u) % n, acc = 0;
    for (uint32_t i = 0; i < iters; ++i) { idx = buf[idx] % n; acc += idx; }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    uint32_t *d_ctr, *d_out, *d_buf;
    const uint32_t nbuf = 64u << 20;
    CK(hipMalloc(&d_ctr, 1 << 20)); CK(hipMalloc(&d_out, 64)); CK(hipMalloc(&d_buf, 4ull * nbuf));
    CK(hipMemset(d_ctr, 0, 1 << 20));
    std::vector<uint32_t> h(nbuf);
    uint32_t x = 12345;
    for (uint32_t i = 0; i < nbuf; i++) { x = x * 1664525u + 1013904223u; h[i] = x; }
    CK(hipMemcpy(d_buf, h.data(), 4ull * nbuf, hipMemcpyHostToDevice));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time_us = [&](auto launch) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipEventRecord(a, 0); launch(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        return best * 1000.f;
    };
    for (uint32_t grid : {256u, 1024u, 3328u, 6656u})
        printf("empty kernel, %u one-wavefront workgroups with 12.8 KB of LDS: %.1f us\n", grid, time_us([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(64), 0, 0, d_out); }));
    printf("one wavefront, 1000 dependent atomics with return: %.3f us each\n", time_us([&] { hipLaunchKernelGGL(k_atomic_chain, dim3(1), dim3(64), 0, 0, d_ctr, 32u, 1u, 1000u, d_out); }) / 1000.f);
    for (uint32_t groups : {1u, 16u, 256u, 3328u})
        for (uint32_t iters : {1u, 3u, 10u})
            printf("3328 wavefronts, %u dependent atomics each, %u counters (128 B apart): %.1f us\n", iters, groups,
                   time_us([&] { hipLaunchKernelGGL(k_atomic_chain, dim3(3328), dim3(64), 0, 0, d_ctr, 32u, groups, iters, d_out); }));
    printf("one wavefront, 1000 dependent scattered loads (256 MiB): %.3f us each\n", time_us([&] { hipLaunchKernelGGL(k_load_chain, dim3(1), dim3(64), 0, 0, d_buf, nbuf, 1000u, d_out); }) / 1000.f);
    for (uint32_t iters : {1u, 3u, 10u})
        printf("3328 wavefronts, %u dependent scattered loads each: %.1f us\n", iters, time_us([&] { hipLaunchKernelGGL(k_load_chain, dim3(3328), dim3(64), 0, 0, d_buf, nbuf, iters, d_out); }));
    return 0;
}
