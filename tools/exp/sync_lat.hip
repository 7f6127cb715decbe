// round 6 microbenchmark: what a host round trip costs behind a kernel — the ways to wait, and the ways to bring a few KB back.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <vector>
__global__ void k_work(uint32_t *out, uint32_t iters) {
    uint32_t x = threadIdx.x;
    for (uint32_t i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    out[threadIdx.x] = x;
}
__global__ void k_publish(const uint32_t *src, uint32_t *host_dst, uint32_t n, volatile uint32_t *flag, uint32_t seq) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) host_dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *flag = seq; }
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    uint32_t *d, *hp, *hflag;
    hipMalloc(&d, 1 << 20);
    hipHostMalloc(&hp, 1 << 20, hipHostMallocDefault);
    hipHostMalloc(&hflag, 64, hipHostMallocDefault);
    std::vector<uint32_t> pageable(1 << 18);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const uint32_t iters = 20000;      // ~ a 100 us kernel, so that the host is waiting when it ends
    const int reps = 200;
    auto run = [&](const char *name, auto body) {
        for (int w = 0; w < 5; w++) body(w);
        // the kernel's own duration: events
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a, st); hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipEventRecord(b, st); hipEventSynchronize(b);
        float kms = 0; hipEventElapsedTime(&kms, a, b);
        const double t0 = now_us();
        for (int r = 0; r < reps; r++) body(r + 100);
        const double per = (now_us() - t0) / reps;
        printf("%-70s %.1f us per round trip on top of the kernel (%.1f)\n", name, per - kms * 1000.0, kms * 1000.0);
    };
    run("kernel; hipStreamSynchronize", [&](int) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipStreamSynchronize(st); });
    run("kernel; spin on hipStreamQuery", [&](int) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); while (hipStreamQuery(st) == hipErrorNotReady) {} });
    run("kernel; 8 KB D2H to pageable; sync", [&](int) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipMemcpyAsync(pageable.data(), d, 8192, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    run("kernel; 8 KB D2H to page-locked; sync", [&](int) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipMemcpyAsync(hp, d, 8192, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    run("kernel; two D2H (2 KB + 8 KB) to pageable; sync", [&](int) { hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipMemcpyAsync(pageable.data(), d, 2048, hipMemcpyDeviceToHost, st); hipMemcpyAsync(pageable.data() + 4096, d + 4096, 8192, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    run("kernel; publish kernel writes 8 KB to page-locked + flag; host spins", [&](int r) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters);
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, st, d, hp, 2048u, (volatile uint32_t *)hflag, (uint32_t)r + 1u);
        while (*(volatile uint32_t *)hflag != (uint32_t)r + 1u) {}
    });
    run("kernel; sync; 16 KB H2D from pageable; kernel; sync (second leg)", [&](int) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipStreamSynchronize(st);
        hipMemcpyAsync(d, pageable.data(), 16384, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, 1u); hipStreamSynchronize(st); });
    run("kernel; sync; 16 KB H2D from page-locked; kernel; sync (second leg)", [&](int) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipStreamSynchronize(st);
        hipMemcpyAsync(d, hp, 16384, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, 1u); hipStreamSynchronize(st); });
    run("kernel; sync; kernel reading page-locked memory directly; sync", [&](int) {
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, iters); hipStreamSynchronize(st);
        hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, hp, 1u); hipStreamSynchronize(st); });
    return 0;
}
