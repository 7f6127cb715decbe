// Micro-benchmark (GPU box): LDS THROUGHPUT of gather flavours — 16 wavefronts, 8 independent accesses per
// wavefront and iteration (cycles per wave-instruction on one CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k_tput(uint32_t *out, uint32_t iters, uint64_t *cyc) {
    __shared__ __attribute__((aligned(16))) uint32_t tab[16384 + 8];
    for (uint32_t i = threadIdx.x; i < 16384 + 8; i += blockDim.x) tab[i] = i * 2654435761u;
    __syncthreads();
    const uint8_t *tb = (const uint8_t *)tab;
    const uint16_t *th = (const uint16_t *)tab;
    uint32_t x = threadIdx.x * 2654435761u, acc = 0;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t c0 = clock64();
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t a[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { x = x * 1664525u + 1013904223u; a[k] = x >> 8; }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (MODE == 0) acc += tab[a[k] & 16383];                                     // aligned b32 gather
            if (MODE == 1) acc += th[a[k] & 32767];                                      // u16 gather
            if (MODE == 2) { const uint32_t j = a[k] & 16383; acc += tab[j] ^ tab[j + 1]; }   // read2 (adjacent dwords)
            if (MODE == 3) acc += *(const uint32_t *)(tb + (a[k] & 65535));             // unaligned b32 gather
            if (MODE == 4) { const uint64_t v = *(const uint64_t *)(tb + (a[k] & 65535)); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }   // unaligned b64
            if (MODE == 5) { const uint64_t v = *(const uint64_t *)(tb + (a[k] & 65528)); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }   // aligned b64 gather
            if (MODE == 6) acc += tab[(a[k] & 16320) + lane];                            // linear b32 (random row)
            if (MODE == 7) { const uint32_t j = (a[k] & 16320) + lane; acc += tab[j] ^ tab[j + 1]; }   // linear read2
            if (MODE == 8) acc += *(const uint32_t *)(tb + ((a[k] & 65280) + lane + 1));   // consecutive BYTES per lane, unaligned b32
            if (MODE == 9) { const uint32_t j = ((a[k] & 65280) + lane + 1); acc += __builtin_amdgcn_alignbyte(tab[(j >> 2) + 1], tab[j >> 2], j & 3); }  // same via two aligned + alignbyte
            if (MODE == 10) acc += (uint32_t)__builtin_amdgcn_ds_bpermute((int)(a[k] & 252), (int)x);   // bpermute
            if (MODE == 11) { const uint64_t v = *(const uint64_t *)(tb + ((a[k] & 65280) + lane + 1)); acc += (uint32_t)v ^ (uint32_t)(v >> 32); }  // consecutive bytes per lane, unaligned b64
        }
    }
    const uint64_t c1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

template <int MODE>
void run(const char *name, uint32_t *d_o, uint64_t *d_c) {
    hipLaunchKernelGGL(k_tput<MODE>, dim3(256), dim3(1024), 0, 0, d_o, 500, d_c);
    uint64_t cyc = 0;
    (void)hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.1f cycles per wave-instruction (16 waves x 8 per iteration)\n", name, (double)cyc / 500 / 128);
}

int main() {
    uint32_t *d_o; uint64_t *d_c;
    (void)hipMalloc(&d_o, 1024 * 256 * 4); (void)hipMalloc(&d_c, 8);
    run<0>("aligned b32 gather", d_o, d_c);
    run<1>("u16 gather", d_o, d_c);
    run<2>("two adjacent dwords gather (read2)", d_o, d_c);
    run<3>("unaligned b32 gather", d_o, d_c);
    run<4>("unaligned b64 gather", d_o, d_c);
    run<5>("aligned b64 gather", d_o, d_c);
    run<6>("linear b32", d_o, d_c);
    run<7>("linear two adjacent dwords", d_o, d_c);
    run<8>("consecutive bytes/lane, unaligned b32", d_o, d_c);
    run<9>("consecutive bytes/lane, 2 aligned + alignbyte", d_o, d_c);
    run<10>("ds_bpermute", d_o, d_c);
    run<11>("consecutive bytes/lane, unaligned b64", d_o, d_c);
    return 0;
}
