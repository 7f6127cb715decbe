// lfx_hostio.h — host buffers on the drop-in surface (round 6).
//
// What a libflate user calls hands over HOST memory: `io::Write::write(&[u8])` / `io::Read::read(&mut [u8])`
// (src/deflate/encode.rs:241-249, src/gzip.rs:890-895,1018-1047; the canonical caller is `io::copy` with 8 KiB buffers,
// examples/flate.rs:52,96-97) and the one-shot `lfx_encode_host` / `lfx_decode_host`.  Until round 5 those bytes crossed
// PCIe through pageable `hipMemcpy` calls of whole buffers (0.7 GB/s round trip at 256 MiB against 51 GB/s resident).  Here:
//   * PinVec — std::vector over page-locked host memory (hipHostMalloc) the library owns: the stream encoder's pending input,
//     its output staging, the stream decoder's input window and output window.  A DMA engine reads / writes it directly and
//     the copy is asynchronous for real;
//   * host_to_device / device_to_host — a caller's buffer of any kind: page-locked memory (lfx_host_alloc, hipHostMalloc,
//     hipHostRegister) is handed to the DMA engine as it is; pageable memory is staged through page-locked slabs by a few
//     worker threads, each with its own HIP stream, so that the CPU's memcpy into (out of) the slabs and the DMA transfers
//     of several slabs overlap — one thread's memcpy is slower than the link.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <new>
#include <utility>
#include <vector>

namespace lfx {

struct Ctx;

// Allocator of page-locked host memory for the std::vectors that cross PCIe (the stream encoder's pending input and output
// staging, the stream decoder's input and output windows): a DMA engine reads / writes them directly and hipMemcpyAsync is
// asynchronous for real.  Default-initialising construct(): resize() does not zero-fill bytes a copy is about to overwrite
// (a 32 MiB window was 3 ms of memset).  When page-locking fails (no device, limits) plain memory stands in — correct, slower;
// a 64-byte header in front of the block remembers which kind it is.
void *pinned_alloc(size_t bytes);      // nullptr when out of memory
void pinned_free(void *p);
template <class T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U> PinnedAlloc(const PinnedAlloc<U> &) {}
    T *allocate(size_t n) {
        void *p = pinned_alloc(n * sizeof(T));
        if (!p) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t) { pinned_free(p); }
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }                   // default-init: no zero fill
    template <class U, class... A> void construct(U *p, A &&... a) { ::new ((void *)p) U(std::forward<A>(a)...); }
    template <class U> bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
using PinVec = std::vector<uint8_t, PinnedAlloc<uint8_t>>;

constexpr int HOSTIO_WORKERS = 4;              // copy threads for pageable memory (each: one stream, two slabs)
constexpr size_t HOSTIO_SLAB = 4u << 20;       // bytes per slab
constexpr size_t HOSTIO_DIRECT_BELOW = 2u << 20;   // smaller transfers: one plain hipMemcpyAsync

struct HostIo {
    hipStream_t stream[HOSTIO_WORKERS] = {};
    hipEvent_t ev[HOSTIO_WORKERS][2] = {};
    hipEvent_t ev_ready = nullptr;
    uint8_t *slab[HOSTIO_WORKERS][2] = {};
    bool ready = false, broken = false;
    int init();
    void release();
};

// d_dst[0, n) <- h_src[0, n).  On return `st` is ordered behind the transfers: work queued on `st` afterwards sees the data.
// Pageable memory has left h_src by then (it went through the slabs); PAGE-LOCKED memory is only queued — the DMA engine reads
// h_src until `st` has passed the copy, so the caller's buffer must stay untouched until then (lfx_encode_host /
// lfx_decode_host synchronise `st` before they return: their callers never see the difference).  → LFX status
int host_to_device(Ctx *c, void *d_dst, const void *h_src, uint64_t n, hipStream_t st);
// h_dst[0, n) <- d_src[0, n), after everything queued on `st` so far.  Complete on return.
int device_to_host(Ctx *c, void *h_dst, const void *d_src, uint64_t n, hipStream_t st);

}  // namespace lfx
