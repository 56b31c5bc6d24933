// Host buffers -> HBM for the host-pointer entry points ("pinned-host double-buffered H2D"). The
// reference's blocks hand over ordinary host memory (Pothos buffers, LoRaDemod.cpp:151-154); a plain hipMemcpy from pageable memory
// stages it through the runtime's own bounce buffer, serially. Here: pieces scattered on the host are gathered into device memory
// back to back through two pinned staging buffers -- several threads copy into one while the DMA engine drains the other -- and
// pieces that already live in pinned memory (lorahip_host_alloc, or anything hipHostMalloc'ed / hipHostRegister'ed) skip the
// staging copy and go straight to the DMA engine.
#include "lorahip_internal.h"
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

namespace lorahip {

//! per staging buffer: 64 MiB (LORAHIP_STAGE_MB: measurements -- profiles/r06/s19_staging_size.txt: from ordinary memory at 8 / 16 / 32 / 64 MiB SF7 43.9 /
//! 42.6 / 36.8-41.9 / 43.1, SF10 37.3 / 40.7 / 46.6 / 52.4, SF12 43.3 / 47.1 / 50.3 / 52.5 GB/s; two buffers per context that uploads, allocated on first use)
static size_t stageBytes()
{
    static const size_t n = []() {
        if (const char *e = std::getenv("LORAHIP_STAGE_MB")) { const long v = std::atol(e); if (v >= 1 && v <= 1024) return size_t(v) << 20; }
        return size_t(64) << 20;
    }();
    return n;
}
#define kStageBytes (stageBytes())
static const size_t kDirectMin = size_t(1) << 20;         // only pieces this large are asked whether they are pinned (a query per piece costs microseconds); smaller ones are packed

static int uploadThreads()
{
    static const int n = []() {
        if (const char *e = std::getenv("LORAHIP_UPLOAD_THREADS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64) return v; }
        const unsigned hw = std::thread::hardware_concurrency();
        return hw >= 32 ? 8 : (hw >= 16 ? 6 : (hw >= 4 ? 3 : 1));
    }();
    return n;
}

static bool isPinned(const void *p)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // ordinary memory: not an error
    return at.type == hipMemoryTypeHost;
}

//! the pinned allocation that holds p, as a host address range ([lo, hi)); false if p is not in pinned memory the runtime knows
static bool pinnedRange(const void *p, const char *&lo, const char *&hi)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (at.type != hipMemoryTypeHost || at.devicePointer == nullptr) return false;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, at.devicePointer) != hipSuccess || size == 0) { (void)hipGetLastError(); return false; }
    const size_t off = size_t(static_cast<const char *>(at.devicePointer) - static_cast<const char *>(base));
    lo = static_cast<const char *>(p) - off;
    hi = lo + size;
    return true;
}

struct Seg { char *dst; const char *src; size_t bytes; };

//! thread k of T copies bytes [k*share, (k+1)*share) of the concatenation of the segments
static void copyShare(const std::vector<Seg> &segs, const size_t share, const int k)
{
    const size_t lo = size_t(k) * share, hi = lo + share;
    size_t at = 0;
    for (const Seg &s : segs)
    {
        const size_t a = at > lo ? at : lo, b = (at + s.bytes) < hi ? (at + s.bytes) : hi;
        if (a < b) std::memcpy(s.dst + (a - at), s.src + (a - at), b - a);
        at += s.bytes;
        if (at >= hi) break;
    }
}

/*! The helpers of a context's uploads, started once and parked between the staging fills. (Until round 6 every 32 MiB fill started and
 * joined its helper threads: five thread creations per 0.8 ms of copying, 68 times per 2 GiB at SF7 -- the CPU side of the gather, which
 * bounds an upload from ordinary memory, ran at 37 GB/s where the DMA engine takes 55.) */
struct CopyPool
{
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable go, done;
    const std::vector<Seg> *segs = nullptr;
    size_t share = 0;
    unsigned gen = 0;
    int running = 0;
    bool quit = false;

    void worker(const int k)
    {
        unsigned seen = 0;
        for (;;)
        {
            const std::vector<Seg> *mine; size_t sh;
            {
                std::unique_lock<std::mutex> lk(m);
                go.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen; mine = segs; sh = share;
            }
            copyShare(*mine, sh, k);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0) done.notify_one();
            }
        }
    }
    //! helpers 1 .. n-1 (as many as could be started); the caller is thread 0
    int start(const int n)
    {
        try { for (int k = int(th.size()) + 1; k < n; k++) th.emplace_back(&CopyPool::worker, this, k); }
        catch (...) {}                                    // (no exception may cross the C ABI: fewer helpers, the caller copies their shares)
        return int(th.size()) + 1;
    }
    void run(const std::vector<Seg> &s, const size_t total, const int T)
    {
        const int have = start(T);
        const size_t sh = (total + size_t(T) - 1) / size_t(T);
        {
            std::lock_guard<std::mutex> lk(m);
            segs = &s; share = sh; running = have - 1; gen++;
        }
        go.notify_all();
        copyShare(s, sh, 0);
        for (int k = have; k < T; k++) copyShare(s, sh, k);   // shares of helpers that could not be started
        // (helpers beyond T, from an earlier larger T, find their share empty)
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return running == 0; });
    }
    ~CopyPool(void)
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        go.notify_all();
        for (auto &t : th) t.join();
    }
};

//! the segments of one staging fill, copied by the calling thread and up to T-1 parked helpers
static void copySegments(Uploader &u, const std::vector<Seg> &segs, const size_t total)
{
    const int T = total >= (size_t(4) << 20) ? uploadThreads() : 1;
    if (T <= 1) { for (const Seg &s : segs) std::memcpy(s.dst, s.src, s.bytes); return; }
    if (u.pool == nullptr) u.pool = new (std::nothrow) CopyPool();
    if (u.pool == nullptr) { for (const Seg &s : segs) std::memcpy(s.dst, s.src, s.bytes); return; }
    static_cast<CopyPool *>(u.pool)->run(segs, total, T);
}

int gatherUpload(lorahip_ctx *ctx, void *dDstV, const void *const *src, const size_t *bytes, const size_t n)
{
    char *dDst = static_cast<char *>(dDstV);
    Uploader &u = ctx->up;
    if (!u.ready)
    {
        // both buffers and both events, or nothing: a transient failure (pinned memory exhausted) must leave the uploader in the
        // state "not initialised", not half of it -- the next call tries again from scratch
        destroyUploader(ctx);
        for (int k = 0; k < 2; k++)
        {
            hipError_t e = hipHostMalloc(&u.buf[k], kStageBytes, hipHostMallocDefault);
            if (e != hipSuccess) u.buf[k] = nullptr;
            else e = hipEventCreateWithFlags(&u.ev[k], hipEventDisableTiming);
            if (e != hipSuccess)
            {
                destroyUploader(ctx);
                return hipFail(e, "upload staging (hipHostMalloc / hipEventCreate)");
            }
            u.busy[k] = false;
        }
        u.ready = true;
    }
    size_t done = 0;                                     // bytes of the destination already handed to the DMA engine
    int k = 0;
    std::vector<Seg> segs;
    size_t fill = 0;
    auto flush = [&]() -> int
    {
        if (fill == 0) return LORAHIP_OK;
        copySegments(u, segs, fill);
        LORAHIP_TRY(hipMemcpyAsync(dDst + done, u.buf[k], fill, hipMemcpyHostToDevice, ctx->stream));
        LORAHIP_TRY(hipEventRecord(u.ev[k], ctx->stream));
        u.busy[k] = true;
        done += fill; fill = 0; segs.clear();
        k ^= 1;
        if (u.busy[k]) { LORAHIP_TRY(hipEventSynchronize(u.ev[k])); u.busy[k] = false; }     // the other buffer's DMA of two fills ago
        return LORAHIP_OK;
    };
    if (u.busy[k]) { LORAHIP_TRY(hipEventSynchronize(u.ev[k])); u.busy[k] = false; }
    size_t askedUntil = 0;
    for (size_t i = 0; i < n; i++)
    {
        const char *p = static_cast<const char *>(src[i]);
        size_t left = bytes[i];
        if (left == 0) continue;
        // A RUN of equally long pieces at a constant distance from each other inside ONE pinned allocation -- the rows of a (channels,
        // samples) array handed over as one pointer per channel, the slabs of a buffer pool -- is one copy: a plain one when the pieces
        // touch, a strided one otherwise; not a copy (and a pinned-memory query) per piece.
        if (i >= askedUntil && i + 1 < n && bytes[i + 1] == left && src[i + 1] > src[i])
        {
            const size_t stride = size_t(static_cast<const char *>(src[i + 1]) - p);
            size_t m = i + 1;
            while (m + 1 < n && bytes[m + 1] == left && static_cast<const char *>(src[m + 1]) == static_cast<const char *>(src[m]) + stride) m++;
            const size_t rows = m - i + 1;
            const char *lo = nullptr, *hi = nullptr;
            if (stride >= left && rows * left >= kDirectMin && pinnedRange(p, lo, hi) && p >= lo && p + (rows - 1) * stride + left <= hi)
            {
                const int rc = flush();
                if (rc != LORAHIP_OK) return rc;
                if (stride == left) LORAHIP_TRY(hipMemcpyAsync(dDst + done, p, rows * left, hipMemcpyHostToDevice, ctx->stream));
                else LORAHIP_TRY(hipMemcpy2DAsync(dDst + done, left, p, stride, left, rows, hipMemcpyHostToDevice, ctx->stream));
                done += rows * left;
                i = m;
                continue;
            }
            askedUntil = m + 1;                           // (ordinary memory, or not one allocation: the run is asked about once, not per piece)
        }
        if (left >= kDirectMin && isPinned(p))
        {
            const int rc = flush();
            if (rc != LORAHIP_OK) return rc;
            LORAHIP_TRY(hipMemcpyAsync(dDst + done, p, left, hipMemcpyHostToDevice, ctx->stream));
            done += left;
            continue;
        }
        while (left)
        {
            const size_t take = left < kStageBytes - fill ? left : kStageBytes - fill;
            segs.push_back(Seg{static_cast<char *>(u.buf[k]) + fill, p, take});
            fill += take; p += take; left -= take;
            if (fill == kStageBytes) { const int rc = flush(); if (rc != LORAHIP_OK) return rc; }
        }
    }
    return flush();
}

void destroyUploader(lorahip_ctx *ctx)
{
    for (int k = 0; k < 2; k++)
    {
        if (ctx->up.buf[k]) (void)hipHostFree(ctx->up.buf[k]);
        if (ctx->up.ev[k]) (void)hipEventDestroy(ctx->up.ev[k]);
        ctx->up.buf[k] = nullptr; ctx->up.ev[k] = nullptr; ctx->up.busy[k] = false;
    }
    ctx->up.ready = false;
    delete static_cast<CopyPool *>(ctx->up.pool);         // (parks no thread beyond the context's life)
    ctx->up.pool = nullptr;
}

} // namespace lorahip

using namespace lorahip;

extern "C" {

void *lorahip_host_alloc(const size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void lorahip_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

} // extern "C"
