// Stream-ordered scratch for the entry points whose tables live for one call (smt_asw, smt_ncc): ONE arena per
// library (this translation unit), blocks kept per device.
//
// Why the library does not simply use hipMallocAsync: with the device's default memory pool (release threshold 0:
// the pool trims at every synchronisation) the first smt_asw after an smt_ncc in a fresh process returned wrong
// maps in round 2.  tools/asw_bisect.py (profiles/r3b_asw_bisect_default_pool_zero_fill.json) pins the cause: after
// a trim, the block the pool grows again is ZERO-FILLED BY SOMETHING OUTSIDE THE KERNELS WHILE THEY RUN ON IT --
// right after k_asw_anchor 22 732 of the 291 600 table entries it had just written read back as +0.0 (one
// contiguous run), after k_asw3 241 476 of them plus 4 703 entries of the second table -- with vector loads as with
// scalar loads, on the first call after every trim and never on a block that was not re-grown, never with plain
// hipMalloc and never with a pool that does not trim.  A zero weight table gives 0/0 = NaN costs, the "runs of NaN
// pixels, different on every run" of round 2.  That is a defect below this library (ROCm 7.2 stream-ordered pool
// re-growth on this driver), so the default here avoids that path altogether:
//   arena    (default) blocks from plain hipMalloc, cached per device and handed out stream-ordered: a freed block
//            goes back to the same stream at once and to another stream once the event recorded at its free has
//            completed.  Nothing is returned to the driver until smt_scratch_trim() asks (hipFree, after a device
//            synchronisation) -- the hipMalloc / hipFree path every handle of the library uses anyway.
//   pool     a hipMemPool the library creates with the release threshold at its maximum (round 2's workaround).
//   default  the device's default pool, trimming: the configuration that fails; kept for tools/asw_bisect.py.
//   malloc   hipMalloc per call; smt_scratch_free synchronises the stream and hipFree's.
// SMT_SCRATCH_MODE selects (read once).
#include "smt_common.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {
std::mutex g_mu;
hipMemPool_t g_pools[64] = {};

struct Block {
    void *p;
    size_t bytes;
    int dev;
    bool in_use;
    hipStream_t st;      // stream of the last free
    hipEvent_t ev;       // recorded on st at the last free
};
std::vector<Block> g_blocks;

enum { MODE_ARENA = 0, MODE_DEFAULT = 1, MODE_MALLOC = 2, MODE_POOL = 3 };
int scratch_mode()
{
    static const int mode = [] {
        const char *e = getenv("SMT_SCRATCH_MODE");
        if (e && !strcmp(e, "default")) return (int)MODE_DEFAULT;
        if (e && !strcmp(e, "malloc")) return (int)MODE_MALLOC;
        if (e && !strcmp(e, "pool")) return (int)MODE_POOL;
        return (int)MODE_ARENA;
    }();
    return mode;
}

// caller holds g_mu.  Frees every idle block of `dev` whose free has completed; returns the bytes released.
size_t arena_release_idle(int dev, size_t keep_bytes)
{
    size_t kept = 0, released = 0;
    for (size_t k = 0; k < g_blocks.size();) {
        Block &b = g_blocks[k];
        if (b.dev == dev && !b.in_use && hipEventQuery(b.ev) == hipSuccess) {
            if (kept + b.bytes <= keep_bytes) { kept += b.bytes; k++; continue; }
            (void)hipFree(b.p);
            (void)hipEventDestroy(b.ev);
            released += b.bytes;
            g_blocks.erase(g_blocks.begin() + (long)k);
        } else {
            k++;
        }
    }
    return released;
}

hipError_t arena_alloc(void **p, size_t bytes, hipStream_t st, int dev)
{
    std::lock_guard<std::mutex> lock(g_mu);
    // best fit among the idle blocks this stream may take now (no more than twice the request: a 5 GB table must not
    // be pinned down by a 20 KB one)
    long best = -1;
    for (size_t k = 0; k < g_blocks.size(); k++) {
        Block &b = g_blocks[k];
        if (b.dev != dev || b.in_use || b.bytes < bytes || b.bytes / 2 > bytes + 4096) continue;
        if (b.st != st && hipEventQuery(b.ev) != hipSuccess) continue;
        if (best < 0 || b.bytes < g_blocks[(size_t)best].bytes) best = (long)k;
    }
    if (best >= 0) {
        g_blocks[(size_t)best].in_use = true;
        *p = g_blocks[(size_t)best].p;
        return hipSuccess;
    }
    Block nb = {};
    nb.bytes = (bytes + 255) & ~(size_t)255;
    if (nb.bytes == 0) nb.bytes = 256;
    nb.dev = dev;
    nb.in_use = true;
    hipError_t e = hipEventCreateWithFlags(&nb.ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    e = hipMalloc(&nb.p, nb.bytes);
    if (e != hipSuccess) {
        // out of memory: give the idle blocks back first, once
        (void)hipGetLastError();
        if (arena_release_idle(dev, 0) > 0) e = hipMalloc(&nb.p, nb.bytes);
        if (e != hipSuccess) { (void)hipEventDestroy(nb.ev); return e; }
    }
    g_blocks.push_back(nb);
    *p = nb.p;
    return hipSuccess;
}

bool arena_free(void *p, hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_mu);
    for (Block &b : g_blocks)
        if (b.p == p && b.in_use) {
            b.st = st;
            (void)hipEventRecord(b.ev, st);
            b.in_use = false;
            return true;
        }
    return false;
}
}  // namespace

hipError_t smt_scratch_alloc(void **p, size_t bytes, hipStream_t st)
{
    const int mode = scratch_mode();
    if (mode == MODE_MALLOC) return hipMalloc(p, bytes ? bytes : 1);
    if (mode == MODE_DEFAULT) return hipMallocAsync(p, bytes, st);
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (mode == MODE_ARENA) return arena_alloc(p, bytes, st, dev);
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_pools[dev]) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        e = hipMemPoolCreate(&pool, &props);
        if (e != hipSuccess) return e;
        unsigned long long keep = ~0ull;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        g_pools[dev] = pool;
    }
    return hipMallocFromPoolAsync(p, bytes, g_pools[dev], st);
}

void smt_scratch_free(void *p, hipStream_t st)
{
    if (!p) return;
    const int mode = scratch_mode();
    if (mode == MODE_ARENA) {
        (void)arena_free(p, st);
    } else if (mode == MODE_MALLOC) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(p);
    } else {
        (void)hipFreeAsync(p, st);
    }
}

// Gives the scratch the current device holds but does not use back to the driver, keeping at most `keep_bytes`
// cached.  Synchronises the device first (a block freed on a stream is only idle once the stream has passed the
// free).  The arena grows again on the next call that needs it.
SMT_API int smt_scratch_trim(size_t keep_bytes)
{
    int dev = 0;
    SMT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return SMT_ERR_ARG;
    SMT_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_mu);
    (void)arena_release_idle(dev, keep_bytes);
    if (g_pools[dev]) SMT_HIP(hipMemPoolTrimTo(g_pools[dev], keep_bytes));
    return SMT_OK;
}

// Bytes of scratch the library holds from the driver on the current device / has handed out right now.
SMT_API int smt_scratch_info(size_t *reserved_bytes, size_t *used_bytes)
{
    int dev = 0;
    SMT_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return SMT_ERR_ARG;
    unsigned long long r = 0, u = 0;
    std::lock_guard<std::mutex> lock(g_mu);
    for (const Block &b : g_blocks)
        if (b.dev == dev) { r += b.bytes; if (b.in_use) u += b.bytes; }
    if (g_pools[dev]) {
        unsigned long long pr = 0, pu = 0;
        SMT_HIP(hipMemPoolGetAttribute(g_pools[dev], hipMemPoolAttrReservedMemCurrent, &pr));
        SMT_HIP(hipMemPoolGetAttribute(g_pools[dev], hipMemPoolAttrUsedMemCurrent, &pu));
        r += pr; u += pu;
    }
    if (reserved_bytes) *reserved_bytes = (size_t)r;
    if (used_bytes) *used_bytes = (size_t)u;
    return SMT_OK;
}
