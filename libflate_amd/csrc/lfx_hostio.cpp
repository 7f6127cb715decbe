// lfx_hostio.cpp — page-locked staging and fast transfers of host buffers (see lfx_hostio.h).
#include "lfx_hostio.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/lfx.h"
#include "lfx_abi_guard.h"
#include "lfx_ctx.h"

namespace lfx {

// ------------------------------------------------------------------------------------------------ page-locked blocks
namespace {
struct PinHdr { uint64_t magic; uint64_t pinned; uint8_t pad[48]; };
static_assert(sizeof(PinHdr) == 64, "header keeps the block's alignment");
constexpr uint64_t PIN_MAGIC = 0x4C46585F50494E21ull;
}  // namespace
void *pinned_alloc(size_t bytes) {
    void *raw = nullptr;
    uint64_t pin = 1;
    if (hipHostMalloc(&raw, bytes + sizeof(PinHdr), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        pin = 0;
        raw = malloc(bytes + sizeof(PinHdr));
        if (!raw) return nullptr;
    }
    PinHdr *h = (PinHdr *)raw;
    h->magic = PIN_MAGIC;
    h->pinned = pin;
    return h + 1;
}
void pinned_free(void *p) {
    if (!p) return;
    PinHdr *h = (PinHdr *)p - 1;
    if (h->magic != PIN_MAGIC) return;       // (not ours: leak rather than corrupt)
    h->magic = 0;
    if (h->pinned) (void)hipHostFree(h);
    else free(h);
}

// ------------------------------------------------------------------------------------------------ HostIo
int HostIo::init() {
    if (ready) return LFX_OK;
    if (broken) return LFX_E_DEVICE;
    bool ok = hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming) == hipSuccess;
    for (int t = 0; ok && t < HOSTIO_WORKERS; t++) {
        ok = hipStreamCreateWithFlags(&stream[t], hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; ok && k < 2; k++) {
            ok = hipEventCreateWithFlags(&ev[t][k], hipEventDisableTiming) == hipSuccess &&
                 hipHostMalloc((void **)&slab[t][k], HOSTIO_SLAB, hipHostMallocDefault) == hipSuccess;
        }
    }
    if (!ok) { (void)hipGetLastError(); release(); broken = true; return LFX_E_DEVICE; }
    ready = true;
    return LFX_OK;
}
void HostIo::release() {
    for (int t = 0; t < HOSTIO_WORKERS; t++) {
        for (int k = 0; k < 2; k++) {
            if (slab[t][k]) (void)hipHostFree(slab[t][k]);
            if (ev[t][k]) (void)hipEventDestroy(ev[t][k]);
            slab[t][k] = nullptr;
            ev[t][k] = nullptr;
        }
        if (stream[t]) (void)hipStreamDestroy(stream[t]);
        stream[t] = nullptr;
    }
    if (ev_ready) (void)hipEventDestroy(ev_ready);
    ev_ready = nullptr;
    ready = false;
}

// page-locked (hipHostMalloc / hipHostRegister) over the whole range?
static bool is_pinned(const void *p, uint64_t n) {
    for (const uint8_t *q : {(const uint8_t *)p, (const uint8_t *)p + (n ? n - 1 : 0)}) {
        hipPointerAttribute_t a;
        memset(&a, 0, sizeof a);
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

int host_to_device(Ctx *c, void *d_dst, const void *h_src, uint64_t n, hipStream_t st) {
    if (!n) return LFX_OK;
    if (n < HOSTIO_DIRECT_BELOW || is_pinned(h_src, n) || c->hostio.init() != LFX_OK) {
        // page-locked memory: one DMA transfer, asynchronous; pageable and small: the runtime stages it (h_src is free on return)
        return hipMemcpyAsync(d_dst, h_src, n, hipMemcpyHostToDevice, st) == hipSuccess ? LFX_OK : LFX_E_DEVICE;
    }
    HostIo &io = c->hostio;
    const uint64_t nchunks = (n + HOSTIO_SLAB - 1) / HOSTIO_SLAB;
    const int T = (int)std::min<uint64_t>(HOSTIO_WORKERS, nchunks);
    if (hipEventRecord(io.ev_ready, st) != hipSuccess) return LFX_E_DEVICE;      // (d_dst may still be read by work queued on st)
    std::atomic<int> fail{0};
    auto work = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess || hipStreamWaitEvent(io.stream[t], io.ev_ready, 0) != hipSuccess) { fail = 1; return; }
        int slot = 0;
        for (uint64_t i = t; i < nchunks; i += T, slot ^= 1) {
            const uint64_t off = i * HOSTIO_SLAB, len = std::min<uint64_t>(HOSTIO_SLAB, n - off);
            if (hipEventSynchronize(io.ev[t][slot]) != hipSuccess) { fail = 1; return; }      // (the slab's previous transfer)
            memcpy(io.slab[t][slot], (const uint8_t *)h_src + off, len);
            if (hipMemcpyAsync((uint8_t *)d_dst + off, io.slab[t][slot], len, hipMemcpyHostToDevice, io.stream[t]) != hipSuccess ||
                hipEventRecord(io.ev[t][slot], io.stream[t]) != hipSuccess) { fail = 1; return; }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    if (fail) return LFX_E_DEVICE;
    for (int t = 0; t < T; t++)
        for (int k = 0; k < 2; k++)
            if (hipStreamWaitEvent(st, io.ev[t][k], 0) != hipSuccess) return LFX_E_DEVICE;
    return LFX_OK;
}

int device_to_host(Ctx *c, void *h_dst, const void *d_src, uint64_t n, hipStream_t st) {
    if (!n) return hipStreamSynchronize(st) == hipSuccess ? LFX_OK : LFX_E_DEVICE;
    if (n < HOSTIO_DIRECT_BELOW || is_pinned(h_dst, n) || c->hostio.init() != LFX_OK) {
        if (hipMemcpyAsync(h_dst, d_src, n, hipMemcpyDeviceToHost, st) != hipSuccess) return LFX_E_DEVICE;
        return hipStreamSynchronize(st) == hipSuccess ? LFX_OK : LFX_E_DEVICE;
    }
    HostIo &io = c->hostio;
    const uint64_t nchunks = (n + HOSTIO_SLAB - 1) / HOSTIO_SLAB;
    const int T = (int)std::min<uint64_t>(HOSTIO_WORKERS, nchunks);
    if (hipEventRecord(io.ev_ready, st) != hipSuccess) return LFX_E_DEVICE;
    std::atomic<int> fail{0};
    auto work = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess || hipStreamWaitEvent(io.stream[t], io.ev_ready, 0) != hipSuccess) { fail = 1; return; }
        const uint64_t cnt = (nchunks - t + T - 1) / T;             // this worker's chunks: t, t + T, ...
        auto issue = [&](uint64_t k) {
            const uint64_t off = (t + k * T) * HOSTIO_SLAB, len = std::min<uint64_t>(HOSTIO_SLAB, n - off);
            return hipMemcpyAsync(io.slab[t][k & 1], (const uint8_t *)d_src + off, len, hipMemcpyDeviceToHost, io.stream[t]) == hipSuccess &&
                   hipEventRecord(io.ev[t][k & 1], io.stream[t]) == hipSuccess;
        };
        if (!issue(0)) { fail = 1; return; }
        for (uint64_t k = 0; k < cnt; k++) {
            if (k + 1 < cnt && !issue(k + 1)) { fail = 1; return; }     // (the other slab: emptied one trip ago)
            if (hipEventSynchronize(io.ev[t][k & 1]) != hipSuccess) { fail = 1; return; }
            const uint64_t off = (t + k * T) * HOSTIO_SLAB, len = std::min<uint64_t>(HOSTIO_SLAB, n - off);
            memcpy((uint8_t *)h_dst + off, io.slab[t][k & 1], len);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    return fail ? LFX_E_DEVICE : LFX_OK;
}

}  // namespace lfx

// ------------------------------------------------------------------------------------------------ C ABI: page-locked memory for callers
extern "C" void *lfx_host_alloc(size_t bytes) try {
    void *p = nullptr;
    if (!bytes) return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
} catch (...) { return nullptr; }
extern "C" void lfx_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
