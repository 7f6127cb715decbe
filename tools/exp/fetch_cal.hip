// FETCH_SIZE calibration on gfx950 for three access patterns over the same 1 GiB buffer (larger than the 256 MiB
// Infinity Cache): every kernel reads each byte exactly once, so FETCH_SIZE * 1024 / bytes is the counter's factor.
//   wide16   : 16 B per lane, coalesced (1 KiB per wavefront request)           — the guide's case: counter = 1/2
//   narrow4  : 4 B per lane, coalesced (256 B per wavefront request)
//   strided4 : 4 B per lane, lanes 512 B apart, each lane walks 128 consecutive dwords (the inflate scan's pattern)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__global__ void wide16(const uint4 *p, uint64_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void narrow4(const uint32_t *p, uint64_t n4, uint32_t *sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void strided4(const uint32_t *p, uint64_t n4, uint32_t *sink) {
    // lane l of global thread t owns dwords [t * 128, t * 128 + 128)
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    if ((t + 1) * 128 <= n4)
        for (uint32_t k = 0; k < 128; ++k) {
            acc ^= p[t * 128 + k];
            acc = acc * 2654435761u + 1;          // a dependent chain between the loads, like a decoder
        }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const uint64_t bytes = 1ull << 30;
    void *d; uint32_t *sink;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc((void **)&sink, 4) != hipSuccess) return 1;
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(wide16, dim3(4096), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, sink);
        hipLaunchKernelGGL(narrow4, dim3(4096), dim3(256), 0, 0, (const uint32_t *)d, bytes / 4, sink);
        hipLaunchKernelGGL(strided4, dim3((unsigned)(bytes / 4 / 128 / 256)), dim3(256), 0, 0, (const uint32_t *)d, bytes / 4, sink);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: %llu\n", (unsigned long long)bytes);
    return 0;
}
