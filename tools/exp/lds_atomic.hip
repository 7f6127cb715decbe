// Micro-benchmark (GPU box): cost of LDS read-modify-write instructions WITH return on gfx950 — ONE wavefront on a CU
// issuing dependent-free batches of five (the head pass of lfx_match3.hip), random dword addresses in a 32 KiB table.
//   cycles per wave-instruction, 64 lanes each.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(64) void k_atomic(uint32_t *out, uint32_t iters, uint64_t *cyc, uint32_t spread) {
    __shared__ __attribute__((aligned(16))) uint32_t tab[8192];
    for (uint32_t i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = i * 2654435761u;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)tab;
    uint32_t x = threadIdx.x * 2654435761u + 12345u, acc = 0;
    const uint64_t c0 = clock64();
    for (uint32_t i = 0; i < iters; i++) {
        uint32_t a[5], m[5], v[5], o[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            x = x * 1664525u + 1013904223u;
            a[k] = base + (((x >> 8) & spread) << 2);
            const uint32_t sh = (x >> 3) & 16;
            m[k] = 0xFFFFu << sh;
            v[k] = (x >> 16) << sh & m[k];
        }
        if (MODE == 0)
            asm volatile("ds_mskor_rtn_b32 %0, %5, %10, %15\n\tds_mskor_rtn_b32 %1, %6, %11, %16\n\tds_mskor_rtn_b32 %2, %7, %12, %17\n\t"
                         "ds_mskor_rtn_b32 %3, %8, %13, %18\n\tds_mskor_rtn_b32 %4, %9, %14, %19\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]),
                           "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]) : "memory");
        if (MODE == 1)
            asm volatile("ds_wrxchg_rtn_b32 %0, %5, %10\n\tds_wrxchg_rtn_b32 %1, %6, %11\n\tds_wrxchg_rtn_b32 %2, %7, %12\n\t"
                         "ds_wrxchg_rtn_b32 %3, %8, %13\n\tds_wrxchg_rtn_b32 %4, %9, %14\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]) : "memory");
        if (MODE == 2)
            asm volatile("ds_max_rtn_u32 %0, %5, %10\n\tds_max_rtn_u32 %1, %6, %11\n\tds_max_rtn_u32 %2, %7, %12\n\t"
                         "ds_max_rtn_u32 %3, %8, %13\n\tds_max_rtn_u32 %4, %9, %14\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]) : "memory");
        if (MODE == 3)   // plain reads, same addresses
            asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %6\n\tds_read_b32 %2, %7\n\tds_read_b32 %3, %8\n\tds_read_b32 %4, %9\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]) : "memory");
        if (MODE == 4)   // mskor without return
            asm volatile("ds_mskor_b32 %0, %5, %10\n\tds_mskor_b32 %1, %6, %11\n\tds_mskor_b32 %2, %7, %12\n\t"
                         "ds_mskor_b32 %3, %8, %13\n\tds_mskor_b32 %4, %9, %14\n\ts_waitcnt lgkmcnt(0)"
                         :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]),
                           "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]) : "memory");
        if (MODE == 5) { // u16 exchange emulated: read u16 + write u16 (NOT ordered; cost reference)
            asm volatile("ds_read_u16 %0, %5\n\tds_read_u16 %1, %6\n\tds_read_u16 %2, %7\n\tds_read_u16 %3, %8\n\tds_read_u16 %4, %9\n\t"
                         "ds_write_b16 %5, %10\n\tds_write_b16 %6, %11\n\tds_write_b16 %7, %12\n\tds_write_b16 %8, %13\n\tds_write_b16 %9, %14\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4])
                         : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]) : "memory");
        }
        if (MODE != 4) acc += o[0] ^ o[1] ^ o[2] ^ o[3] ^ o[4];
    }
    const uint64_t c1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

template <int MODE>
void run(const char *name, uint32_t *d_o, uint64_t *d_c, uint32_t spread) {
    hipLaunchKernelGGL(k_atomic<MODE>, dim3(256), dim3(64), 0, 0, d_o, 2000, d_c, spread);
    uint64_t cyc = 0;
    (void)hipMemcpy(&cyc, d_c, 8, hipMemcpyDeviceToHost);
    printf("%-34s spread %5u dwords: %.1f cycles per wave-instruction\n", name, spread + 1, (double)cyc / 2000 / 5);
}

int main() {
    uint32_t *d_o; uint64_t *d_c;
    (void)hipMalloc(&d_o, 64 * 256 * 4); (void)hipMalloc(&d_c, 8);
    for (uint32_t spread : {8191u, 31u}) {
        run<3>("ds_read_b32", d_o, d_c, spread);
        run<0>("ds_mskor_rtn_b32", d_o, d_c, spread);
        run<4>("ds_mskor_b32 (no return)", d_o, d_c, spread);
        run<1>("ds_wrxchg_rtn_b32", d_o, d_c, spread);
        run<2>("ds_max_rtn_u32", d_o, d_c, spread);
        run<5>("ds_read_u16 + ds_write_b16", d_o, d_c, spread);
    }
    return 0;
}
