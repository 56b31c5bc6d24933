// C ABI of liblorahip.so: context management, the batch entry points and the
// LoRaDetector shim (include/lorahip.h). Host-side only; kernels are in lorahip_kernels.hip.
#include "lorahip_internal.h"
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <new>

namespace lorahip {

static thread_local std::string g_lastError;

void setLastError(const std::string &s) { g_lastError = s; }

int hipFail(const hipError_t e, const char *what)
{
    g_lastError = std::string(what) + ": " + hipGetErrorString(e);
    if (e == hipErrorOutOfMemory) return LORAHIP_E_NOMEM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return LORAHIP_E_NODEVICE;
    return LORAHIP_E_HIP;
}

// The attribute is set ONCE per (kernel, device). Kernels whose LDS size depends on the object (the channeliser's tile, the
// decoder's rows) pass the device maximum, 160 KiB, so that a later, larger instance is covered too.
hipError_t ensureDynamicLds(const void *kernel, const size_t bytes, unsigned long long &doneMask)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&doneMask, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
    if (e != hipSuccess) return e;
    __atomic_fetch_or(&doneMask, bit, __ATOMIC_RELEASE);
    return hipSuccess;
}

int residentWorkgroups(const void *kernel, const int threads, const size_t smem)
{
    int dev = 0, cus = 0, perCu = 0;
    // (a refusal here must not be taken for the failure of the launch that follows: hipGetLastError() is read after it)
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kernel, threads, smem) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return perCu > 0 && cus > 0 ? perCu * cus : 0;
}

int residentWorkgroupsCached(PerDeviceCount &cache, const void *kernel, const int threads, const size_t smem)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int *slot = &cache.v[dev & 63];
    int r = __atomic_load_n(slot, __ATOMIC_RELAXED);
    if (r < 0)
    {
        r = residentWorkgroups(kernel, threads, smem);          // (two threads of one device may both ask: they store the same answer)
        __atomic_store_n(slot, r, __ATOMIC_RELAXED);
    }
    return r;
}

static bool isGfx950(const int device)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
    return std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
}

//! device and pinned-host staging of the host-pointer entry point (the IQ itself goes through gatherUpload: device side only)
static int growStage(lorahip_ctx *ctx, const size_t devBytes, const size_t hostBytes)
{
    if (devBytes <= ctx->dStageBytes && hostBytes <= ctx->hStageBytes) return LORAHIP_OK;
    if (ctx->dStage) { (void)hipFree(ctx->dStage); ctx->dStage = nullptr; }
    if (ctx->hStage) { (void)hipHostFree(ctx->hStage); ctx->hStage = nullptr; }
    ctx->dStageBytes = ctx->hStageBytes = 0;                    // both or neither: a half-grown pair must not look usable
    const size_t dCap = devBytes + devBytes / 4, hCap = hostBytes + hostBytes / 4 + 256;
    LORAHIP_TRY(hipMalloc(&ctx->dStage, dCap));
    const hipError_t e = hipHostMalloc(&ctx->hStage, hCap, hipHostMallocDefault);
    if (e != hipSuccess)
    {
        (void)hipFree(ctx->dStage); ctx->dStage = nullptr; ctx->hStage = nullptr;
        return hipFail(e, "hipHostMalloc(staging)");
    }
    ctx->dStageBytes = dCap; ctx->hStageBytes = hCap;
    return LORAHIP_OK;
}

} // namespace lorahip

using namespace lorahip;

extern "C" {

const char *lorahip_strerror(const int code)
{
    switch (code)
    {
    case LORAHIP_OK: return "ok";
    case LORAHIP_E_INVALID: return "invalid argument";
    case LORAHIP_E_NODEVICE: return "no usable HIP device";
    case LORAHIP_E_HIP: return "HIP runtime error";
    case LORAHIP_E_NOMEM: return "out of memory";
    case LORAHIP_E_ARCH: return "device is not gfx950";
    default: return "unknown error";
    }
}

const char *lorahip_last_error(void) { return g_lastError.c_str(); }

int lorahip_version(void) { return 4; }

int lorahip_selfcheck(void)
{
    if (!fastLayoutsOk()) { setLastError("an LDS exchange layout of lorahip_fast.hip is not injective"); return LORAHIP_E_INVALID; }
    if (!wideLayoutsOk()) { setLastError("an LDS exchange layout of lorahip_wide.hip is not injective"); return LORAHIP_E_INVALID; }
    return LORAHIP_OK;
}

int lorahip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; d++) if (isGfx950(d)) ok++;
    return ok;
}

int lorahip_create(lorahip_ctx **out, const int device, const int sf)
{
    if (out == nullptr) return LORAHIP_E_INVALID;
    *out = nullptr;
    if (sf < LORAHIP_SF_MIN || sf > LORAHIP_SF_MAX) return LORAHIP_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); setLastError("no HIP device"); return LORAHIP_E_NODEVICE; }
    if (device < 0 || device >= n) return LORAHIP_E_NODEVICE;
    if (!isGfx950(device)) return LORAHIP_E_ARCH;
    const DeviceGuard guard(device);                            // the caller's current device is restored on return

    lorahip_ctx *ctx = new (std::nothrow) lorahip_ctx();
    if (ctx == nullptr) return LORAHIP_E_NOMEM;
    std::memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    ctx->sf = sf;
    ctx->N = size_t(1) << sf;
    ctx->powerScale = float(20 * std::log10(double(ctx->N)));   // LoRaDetector.hpp:18

    HostTables t;
    buildHostTables(sf, t, true);
    const size_t nb = ctx->N * sizeof(cf32);
    int rc = LORAHIP_OK;
    do
    {
#define LORAHIP_CK(expr) { hipError_t _e = (expr); if (_e != hipSuccess) { rc = hipFail(_e, #expr); break; } }
        {
            // LORAHIP_PART_PRIORITY=1 (a measurement: profiles/r06): the stream of a context at SF11 / 12 above, at SF7 / 8 below the
            // others -- in a mixed object the long windows' launches then get the device first and their tail starts earliest
            static const bool byPrio = std::getenv("LORAHIP_PART_PRIORITY") != nullptr;
            int least = 0, greatest = 0;
            if (byPrio && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
            {
                const int prio = sf >= 11 ? greatest : (sf <= 8 ? least : (least + greatest) / 2);
                LORAHIP_CK(hipStreamCreateWithPriority(&ctx->ownStream, hipStreamNonBlocking, prio));
            }
            else LORAHIP_CK(hipStreamCreateWithFlags(&ctx->ownStream, hipStreamNonBlocking));
        }
        ctx->stream = ctx->ownStream;
        LORAHIP_CK(hipEventCreate(&ctx->ev0));
        LORAHIP_CK(hipEventCreate(&ctx->ev1));
        LORAHIP_CK(hipMalloc((void **)&ctx->dUp, nb));
        LORAHIP_CK(hipMalloc((void **)&ctx->dDown, nb));
        LORAHIP_CK(hipMalloc((void **)&ctx->dTw, nb));
        LORAHIP_CK(hipMalloc((void **)&ctx->dFine, nb * LORAHIP_FINE_STEPS));
        LORAHIP_CK(hipMemcpy(ctx->dUp, t.up.data(), nb, hipMemcpyHostToDevice));
        LORAHIP_CK(hipMemcpy(ctx->dDown, t.down.data(), nb, hipMemcpyHostToDevice));
        LORAHIP_CK(hipMemcpy(ctx->dTw, t.twiddle.data(), nb, hipMemcpyHostToDevice));
        LORAHIP_CK(hipMemcpy(ctx->dFine, t.fine.data(), nb * LORAHIP_FINE_STEPS, hipMemcpyHostToDevice));
        {
            // fp64 factor tables that reproduce the fine-tune table without a gather (lorahip_fine.h); only when the host check of
            // all 128*N entries passes -- otherwise the kernels keep reading the table itself
            std::vector<double> fa, fb;
            if (buildFineSplit(sf, t.fine, fa, fb))
            {
                LORAHIP_CK(hipMalloc((void **)&ctx->dFineA, fa.size() * sizeof(double)));
                LORAHIP_CK(hipMalloc((void **)&ctx->dFineB, fb.size() * sizeof(double)));
                LORAHIP_CK(hipMemcpy(ctx->dFineA, fa.data(), fa.size() * sizeof(double), hipMemcpyHostToDevice));
                LORAHIP_CK(hipMemcpy(ctx->dFineB, fb.data(), fb.size() * sizeof(double), hipMemcpyHostToDevice));
            }
        }
        const std::vector<cf32> st = buildStageTwiddles(sf, t.twiddle);
        LORAHIP_CK(hipMalloc((void **)&ctx->dTwStage, (st.size() + 1) * sizeof(cf32)));
        LORAHIP_CK(hipMemcpy(ctx->dTwStage, st.data(), st.size() * sizeof(cf32), hipMemcpyHostToDevice));
        hipDeviceProp_t prop;
        LORAHIP_CK(hipGetDeviceProperties(&prop, device));
        ctx->cuCount = prop.multiProcessorCount;
#undef LORAHIP_CK
    } while (false);
    if (rc != LORAHIP_OK) { lorahip_destroy(ctx); return rc; }
    *out = ctx;
    return LORAHIP_OK;
}

void lorahip_destroy(lorahip_ctx *ctx)
{
    if (ctx == nullptr) return;
    const DeviceGuard guard(ctx->device);
    if (ctx->ownStream) (void)hipStreamSynchronize(ctx->ownStream);
    if (ctx->dUp) (void)hipFree(ctx->dUp);
    if (ctx->dDown) (void)hipFree(ctx->dDown);
    if (ctx->dTw) (void)hipFree(ctx->dTw);
    if (ctx->dFine) (void)hipFree(ctx->dFine);
    if (ctx->dTwStage) (void)hipFree(ctx->dTwStage);
    if (ctx->dFineA) (void)hipFree(ctx->dFineA);
    if (ctx->dFineB) (void)hipFree(ctx->dFineB);
    if (ctx->dStage) (void)hipFree(ctx->dStage);
    if (ctx->hStage) (void)hipHostFree(ctx->hStage);
    destroyUploader(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ownStream) (void)hipStreamDestroy(ctx->ownStream);
    delete ctx;
}

int lorahip_sf(const lorahip_ctx *ctx) { return ctx ? ctx->sf : LORAHIP_E_INVALID; }

int lorahip_set_stream(lorahip_ctx *ctx, void *hip_stream)
{
    if (ctx == nullptr) return LORAHIP_E_INVALID;
    // NULL is HIP's null stream (what torch calls its default stream), not "no stream"
    ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return LORAHIP_OK;
}

int lorahip_reset_stream(lorahip_ctx *ctx)
{
    if (ctx == nullptr) return LORAHIP_E_INVALID;
    ctx->stream = ctx->ownStream;
    return LORAHIP_OK;
}

int lorahip_synchronize(lorahip_ctx *ctx)
{
    if (ctx == nullptr) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    LORAHIP_TRY(hipStreamSynchronize(ctx->stream));
    return LORAHIP_OK;
}

int lorahip_set_variant(lorahip_ctx *ctx, const int variant)
{
    if (ctx == nullptr || variant < 0 || variant > LORAHIP_VARIANT_FMA) return LORAHIP_E_INVALID;
    ctx->variant = variant;
    return LORAHIP_OK;
}

int lorahip_set_fine_gather(lorahip_ctx *ctx, const int enable)
{
    if (ctx == nullptr) return LORAHIP_E_INVALID;
    ctx->fineGather = enable != 0;
    return LORAHIP_OK;
}

int lorahip_fine_split_active(const lorahip_ctx *ctx)
{
    if (ctx == nullptr) return LORAHIP_E_INVALID;
    return (ctx->dFineA != nullptr && ctx->dFineB != nullptr && !ctx->fineGather) ? 1 : 0;
}

static int checkBatch(const lorahip_ctx *ctx, const lorahip_batch *b)
{
    if (ctx == nullptr || b == nullptr) return LORAHIP_E_INVALID;
    if (b->struct_size != sizeof(lorahip_batch)) return LORAHIP_E_INVALID;
    if (b->n_windows > 0xffffffffu) return LORAHIP_E_INVALID;
    if (b->n_windows == 0) return LORAHIP_OK;
    if (!b->iq || !b->sym || !b->power || !b->power_avg || !b->f_index) return LORAHIP_E_INVALID;
    if (!b->chirp_sel && (b->chirp_sel_all < 0 || b->chirp_sel_all > LORAHIP_CHIRP_NONE)) return LORAHIP_E_INVALID;
    return LORAHIP_OK;
}

static void fillArgs(const lorahip_ctx *ctx, const lorahip_batch *b, DetectArgs &a)
{
    a.iq = reinterpret_cast<const float2 *>(b->iq);
    a.offsets = reinterpret_cast<const long long *>(b->offsets);
    a.stride = (long long)(b->window_stride ? b->window_stride : ctx->N);
    a.chirpSel = b->chirp_sel;
    a.chirpSelAll = b->chirp_sel_all;
    a.fineIdx0 = b->fine_idx0;
    a.fineErr = b->fine_err;
    a.sym = b->sym;
    a.power = b->power;
    a.powerAvg = b->power_avg;
    a.fIndex = b->f_index;
    a.fineIdxOut = b->fine_idx_out;
    a.fftOut = reinterpret_cast<float2 *>(b->fft_out);
    a.decOut = reinterpret_cast<float2 *>(b->dec_out);
    a.up = ctx->dUp;
    a.down = ctx->dDown;
    a.fine = ctx->dFine;
    a.tw = ctx->dTw;
    a.fineA = ctx->fineGather ? nullptr : ctx->dFineA;
    a.fineB = ctx->fineGather ? nullptr : ctx->dFineB;
    a.nWindows = unsigned(b->n_windows);
    a.powerScale = ctx->powerScale;
}

int lorahip_detect_batch(lorahip_ctx *ctx, const lorahip_batch *b)
{
    const int rc = checkBatch(ctx, b);
    if (rc != LORAHIP_OK || b->n_windows == 0) return rc;
    const DeviceGuard guard(ctx->device);
    DetectArgs a;
    fillArgs(ctx, b, a);
    FastTables ft;
    ft.twStage = ctx->dTwStage;
    ft.nBlocksHint = ctx->cuCount;
    LORAHIP_TRY(launchDetect(ctx->sf, ctx->variant, a, ft, ctx->stream));
    return LORAHIP_OK;
}

// Host-pointer convenience: stage through context-owned pinned + device buffers.
int lorahip_detect_batch_host(lorahip_ctx *ctx, const lorahip_batch *b)
{
    int rc = checkBatch(ctx, b);
    if (rc != LORAHIP_OK || b->n_windows == 0) return rc;
    const DeviceGuard guard(ctx->device);
    const size_t N = ctx->N, W = b->n_windows;
    const size_t stride = b->window_stride ? b->window_stride : N;
    if (b->fine_idx0)                                            // host pointers: the index must lie inside the 128*N-entry table
        for (size_t w = 0; w < W; w++)
            if (b->fine_idx0[w] < 0 || size_t(b->fine_idx0[w]) >= N * LORAHIP_FINE_STEPS) return LORAHIP_E_INVALID;
    size_t iqLen = 0;
    if (b->offsets)
    {
        for (size_t w = 0; w < W; w++)
        {
            if (b->offsets[w] < 0) return LORAHIP_E_INVALID;
            if (size_t(b->offsets[w]) + N > iqLen) iqLen = size_t(b->offsets[w]) + N;
        }
    }
    else
    {
        if (stride > (size_t(1) << 40) / W) return LORAHIP_E_INVALID;
        iqLen = (W - 1) * stride + N;
    }
    if (iqLen > (size_t(1) << 40)) return LORAHIP_E_INVALID;     // 8 TiB of samples: a wrapped or garbage offset, not a batch

    // carve one staging block: inputs first, then outputs (256 B aligned pieces)
    struct Piece { size_t off, bytes; };
    size_t cur = 0;
    auto carve = [&cur](const size_t bytes) { Piece p = { cur, bytes }; cur += (bytes + 255) & ~size_t(255); return p; };
    const Piece pIq = carve(iqLen * sizeof(cf32));
    const Piece pOff = carve(b->offsets ? W * sizeof(int64_t) : 0);
    const Piece pSel = carve(b->chirp_sel ? W * sizeof(int32_t) : 0);
    const Piece pIdx = carve(b->fine_idx0 ? W * sizeof(int32_t) : 0);
    const Piece pErr = carve(b->fine_err ? W * sizeof(float) : 0);
    const size_t inBytes = cur;
    const Piece pSym = carve(W * sizeof(uint16_t));
    const Piece pPow = carve(W * sizeof(float));
    const Piece pAvg = carve(W * sizeof(float));
    const Piece pFi = carve(W * sizeof(float));
    const Piece pIdxOut = carve(b->fine_idx_out ? W * sizeof(int32_t) : 0);
    const Piece pFft = carve(b->fft_out ? W * N * sizeof(cf32) : 0);
    const Piece pDec = carve(b->dec_out ? W * N * sizeof(cf32) : 0);
    // the device holds every piece; the pinned host mirror everything but the IQ, which is first in the layout (its offset there
    // is 0) and travels through the double-buffered upload instead
    const size_t hShift = pOff.off;
    rc = growStage(ctx, cur, cur - hShift);
    if (rc != LORAHIP_OK) return rc;

    char *d = static_cast<char *>(ctx->dStage), *h = static_cast<char *>(ctx->hStage) - hShift;   // h + off is valid for off >= hShift
    {
        const void *src[1] = { b->iq };
        const size_t len[1] = { pIq.bytes };
        rc = gatherUpload(ctx, d + pIq.off, src, len, 1);
        if (rc != LORAHIP_OK) return rc;
    }
    if (b->offsets) std::memcpy(h + pOff.off, b->offsets, pOff.bytes);
    if (b->chirp_sel) std::memcpy(h + pSel.off, b->chirp_sel, pSel.bytes);
    if (b->fine_idx0) std::memcpy(h + pIdx.off, b->fine_idx0, pIdx.bytes);
    if (b->fine_err) std::memcpy(h + pErr.off, b->fine_err, pErr.bytes);
    if (inBytes > hShift) LORAHIP_TRY(hipMemcpyAsync(d + hShift, h + hShift, inBytes - hShift, hipMemcpyHostToDevice, ctx->stream));

    lorahip_batch db = *b;
    db.iq = reinterpret_cast<const float *>(d + pIq.off);
    db.offsets = b->offsets ? reinterpret_cast<const int64_t *>(d + pOff.off) : nullptr;
    db.chirp_sel = b->chirp_sel ? reinterpret_cast<const int32_t *>(d + pSel.off) : nullptr;
    db.fine_idx0 = b->fine_idx0 ? reinterpret_cast<const int32_t *>(d + pIdx.off) : nullptr;
    db.fine_err = b->fine_err ? reinterpret_cast<const float *>(d + pErr.off) : nullptr;
    db.sym = reinterpret_cast<uint16_t *>(d + pSym.off);
    db.power = reinterpret_cast<float *>(d + pPow.off);
    db.power_avg = reinterpret_cast<float *>(d + pAvg.off);
    db.f_index = reinterpret_cast<float *>(d + pFi.off);
    db.fine_idx_out = b->fine_idx_out ? reinterpret_cast<int32_t *>(d + pIdxOut.off) : nullptr;
    db.fft_out = b->fft_out ? reinterpret_cast<float *>(d + pFft.off) : nullptr;
    db.dec_out = b->dec_out ? reinterpret_cast<float *>(d + pDec.off) : nullptr;
    rc = lorahip_detect_batch(ctx, &db);
    if (rc != LORAHIP_OK) return rc;

    LORAHIP_TRY(hipMemcpyAsync(h + inBytes, d + inBytes, cur - inBytes, hipMemcpyDeviceToHost, ctx->stream));
    LORAHIP_TRY(hipStreamSynchronize(ctx->stream));
    std::memcpy(b->sym, h + pSym.off, pSym.bytes);
    std::memcpy(b->power, h + pPow.off, pPow.bytes);
    std::memcpy(b->power_avg, h + pAvg.off, pAvg.bytes);
    std::memcpy(b->f_index, h + pFi.off, pFi.bytes);
    if (b->fine_idx_out) std::memcpy(b->fine_idx_out, h + pIdxOut.off, pIdxOut.bytes);
    if (b->fft_out) std::memcpy(b->fft_out, h + pFft.off, pFft.bytes);
    if (b->dec_out) std::memcpy(b->dec_out, h + pDec.off, pDec.bytes);
    return LORAHIP_OK;
}

int lorahip_timer_start(lorahip_ctx *ctx)
{
    if (ctx == nullptr) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    LORAHIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return LORAHIP_OK;
}

int lorahip_timer_stop(lorahip_ctx *ctx, float *elapsed_ms)
{
    if (ctx == nullptr || elapsed_ms == nullptr) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    LORAHIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    LORAHIP_TRY(hipEventSynchronize(ctx->ev1));
    LORAHIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return LORAHIP_OK;
}

int lorahip_membw_probe(lorahip_ctx *ctx, const float *buf_dev, const size_t n_bytes, const int pattern, const int blocks_per_cu)
{
    if (ctx == nullptr || buf_dev == nullptr || blocks_per_cu < 1) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    LORAHIP_TRY(launchMembw(reinterpret_cast<const float2 *>(buf_dev), n_bytes, pattern, ctx->cuCount * blocks_per_cu,
                            reinterpret_cast<float *>(ctx->dTwStage), ctx->stream));
    return LORAHIP_OK;
}

int lorahip_synth_symbols(lorahip_ctx *ctx, float *iq_dev, const uint16_t *sym_dev, const size_t n_windows,
                          const float ampl, const float noise_sigma, const uint64_t seed)
{
    if (ctx == nullptr || (n_windows && (!iq_dev || !sym_dev))) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    LORAHIP_TRY(launchSynth(ctx->sf, reinterpret_cast<float2 *>(iq_dev), sym_dev, n_windows, ampl, noise_sigma,
                            (unsigned long long)seed, ctx->stream));
    return LORAHIP_OK;
}

size_t lorahip_mod_frame_len(const int sf, const size_t nsyms, const size_t padding)
{
    if (sf < LORAHIP_SF_MIN || sf > LORAHIP_SF_MAX) return 0;
    const size_t N = size_t(1) << sf;
    return N * (10 + 2 + 2 + nsyms + (padding ? padding : 1)) + N / 4;      // LoRaMod.cpp:135-229
}

int lorahip_mod_frames(lorahip_ctx *ctx, float *iq_dev, const size_t frame_stride, const uint16_t *syms_dev,
                       const size_t n_frames, const size_t nsyms, const unsigned char sync, const float ampl, const size_t padding)
{
    if (ctx == nullptr || (n_frames && (!iq_dev || !syms_dev)) || nsyms == 0 || nsyms > 0x7fffffu || padding > 0x7fffffu) return LORAHIP_E_INVALID;
    if (n_frames > 0xffffffffu || frame_stride < lorahip_mod_frame_len(ctx->sf, nsyms, padding)) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    LORAHIP_TRY(launchModFrames(reinterpret_cast<float2 *>(iq_dev), (long long)frame_stride, syms_dev, n_frames, int(nsyms), int(sync), ampl,
                                int(padding), ctx->sf, ctx->stream));
    return LORAHIP_OK;
}

int lorahip_add_awgn(lorahip_ctx *ctx, float *iq_dev, const size_t n_samples, const float sigma, const uint64_t seed)
{
    if (ctx == nullptr || (n_samples && !iq_dev)) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    LORAHIP_TRY(launchAwgn(reinterpret_cast<float2 *>(iq_dev), n_samples, sigma, (unsigned long long)seed, ctx->stream));
    return LORAHIP_OK;
}

int lorahip_decode_packets(lorahip_ctx *ctx, const lorahip_decoder_cfg *cfg, const uint16_t *syms_dev, const size_t sym_stride,
                           const int32_t *nsyms_dev, const size_t n_packets, uint8_t *out_dev, const size_t out_stride,
                           int32_t *out_len_dev, int32_t *dropped_dev)
{
    if (ctx == nullptr || cfg == nullptr || cfg->struct_size != sizeof(lorahip_decoder_cfg)) return LORAHIP_E_INVALID;
    if (n_packets == 0) return LORAHIP_OK;
    if (!syms_dev || !nsyms_dev || !out_dev || !out_len_dev || !dropped_dev || n_packets > 0x7fffffffu) return LORAHIP_E_INVALID;
    if (cfg->sf < 1 || cfg->sf > LORAHIP_SF_MAX || cfg->ppm < 0 || cfg->ppm > cfg->sf || cfg->rdd < 0 || cfg->rdd > 4 || cfg->data_length < 0) return LORAHIP_E_INVALID;
    // An explicit header occupies the first 5 codewords of the first (PPM-codeword) block: with fewer than 5 bits per symbol the
    // reference whitens `PPM - 5` (an unsigned short: ~65535) codewords past its buffer (LoRaDecoder.cpp:235, undefined behaviour).
    // There is nothing to be identical to; refuse the configuration instead of corrupting device memory.
    if (cfg->explicit_hdr && cfg->interleaving && (cfg->ppm ? cfg->ppm : cfg->sf) < 5)
    {
        setLastError("explicit header needs a symbol size (PPM) of at least 5 bits");
        return LORAHIP_E_INVALID;
    }
    if (sym_stride == 0 || sym_stride > size_t(decodeMaxSymbols()) || (out_stride & 1) || out_stride < 2 * (sym_stride + 8)) return LORAHIP_E_INVALID;
    // without a header the length of every packet is the setter's: what the decoder's tables reach bounds it (the reference has no bound;
    // a LoRa length field is one byte). Refused loudly -- never a packet silently not decoded.
    if (!cfg->explicit_hdr && cfg->interleaving && cfg->data_length > decodeMaxDataLength())
    {
        setLastError("data_length beyond what this build decodes (lorahip_decode_max_data_length())");
        return LORAHIP_E_INVALID;
    }
    const DeviceGuard guard(ctx->device);
    DecodeArgs a;
    a.syms = syms_dev; a.nsyms = nsyms_dev; a.out = out_dev; a.outLen = out_len_dev; a.dropped = dropped_dev;
    a.nPackets = unsigned(n_packets); a.symStride = int(sym_stride); a.outStride = int(out_stride);
    a.sf = cfg->sf; a.ppm = cfg->ppm; a.rdd = cfg->rdd; a.crcc = cfg->crcc; a.interleaving = cfg->interleaving;
    a.errorCheck = cfg->error_check; a.explicitHdr = cfg->explicit_hdr; a.hdr = cfg->hdr; a.dataLength = cfg->data_length;
    LORAHIP_TRY(launchDecode(a, ctx->variant, ctx->stream));
    return LORAHIP_OK;
}

/* The same for a caller in HOST memory (a Pothos block: lora_sdr_amd/pothos/LoRaDecoderBatch.cpp): the rows are staged through the
 * context's pinned buffer, decoded, and the three outputs copied back; synchronous. */
int lorahip_decode_packets_host(lorahip_ctx *ctx, const lorahip_decoder_cfg *cfg, const uint16_t *syms, const size_t sym_stride,
                                const int32_t *nsyms, const size_t n_packets, uint8_t *out, const size_t out_stride, int32_t *out_len,
                                int32_t *dropped)
{
    if (ctx == nullptr || cfg == nullptr) return LORAHIP_E_INVALID;
    if (n_packets == 0) return LORAHIP_OK;
    if (!syms || !nsyms || !out || !out_len || !dropped || n_packets > 0x7fffffffu) return LORAHIP_E_INVALID;
    if (sym_stride == 0 || sym_stride > size_t(decodeMaxSymbols()) || (out_stride & 1) || out_stride < 2 * (sym_stride + 8)) return LORAHIP_E_INVALID;
    const DeviceGuard guard(ctx->device);
    struct Piece { size_t off, bytes; };
    size_t cur = 0;
    auto carve = [&cur](const size_t bytes) { Piece p = { cur, bytes }; cur += (bytes + 255) & ~size_t(255); return p; };
    const Piece pSym = carve(n_packets * sym_stride * sizeof(uint16_t));
    const Piece pN = carve(n_packets * sizeof(int32_t));
    const size_t inBytes = cur;
    const Piece pOut = carve(n_packets * out_stride);
    const Piece pLen = carve(n_packets * sizeof(int32_t));
    const Piece pDrop = carve(n_packets * sizeof(int32_t));
    { const int rc = growStage(ctx, cur, cur); if (rc != LORAHIP_OK) return rc; }
    char *d = static_cast<char *>(ctx->dStage), *h = static_cast<char *>(ctx->hStage);
    std::memcpy(h + pSym.off, syms, pSym.bytes);
    std::memcpy(h + pN.off, nsyms, pN.bytes);
    LORAHIP_TRY(hipMemcpyAsync(d, h, inBytes, hipMemcpyHostToDevice, ctx->stream));
    const int rc = lorahip_decode_packets(ctx, cfg, reinterpret_cast<const uint16_t *>(d + pSym.off), sym_stride, reinterpret_cast<const int32_t *>(d + pN.off),
                                          n_packets, reinterpret_cast<uint8_t *>(d + pOut.off), out_stride, reinterpret_cast<int32_t *>(d + pLen.off),
                                          reinterpret_cast<int32_t *>(d + pDrop.off));
    if (rc != LORAHIP_OK) return rc;
    LORAHIP_TRY(hipMemcpyAsync(h + inBytes, d + inBytes, cur - inBytes, hipMemcpyDeviceToHost, ctx->stream));
    LORAHIP_TRY(hipStreamSynchronize(ctx->stream));
    std::memcpy(out, h + pOut.off, pOut.bytes);
    std::memcpy(out_len, h + pLen.off, pLen.bytes);
    std::memcpy(dropped, h + pDrop.off, pDrop.bytes);
    return LORAHIP_OK;
}

int lorahip_decode_max_symbols(void) { return decodeMaxSymbols(); }
int lorahip_decode_max_data_length(void) { return decodeMaxDataLength(); }

/***********************************************************************
 * LoRaDetector<float> shim
 **********************************************************************/
} // extern "C"

struct lorahip_detector
{
    lorahip_ctx *ctx;
    size_t N;
    // One block of pinned host memory that the device addresses directly (hipHostMalloc: mapped, coherent): the window's samples
    // (_fftInput, LoRaDetector.hpp:68), its bins (_fftOutput, :69) and detect()'s four results. A detect() is then ONE kernel launch
    // and one wait -- the kernel reads its 8 N bytes over PCIe and writes 8 N + 14 back -- instead of a staged upload, the kernel and
    // five downloads (28 -> the figure in DESIGN.md section 5 per call; the verbatim CPU detector takes 1-30 us for N = 2^7 ... 2^12).
    char *block;
    cf32 *input, *output;
    uint16_t *sym;
    float *res;                   // power, powerAvg, fIndex
};

extern "C" {

int lorahip_detector_create(lorahip_detector **out, const int device, const size_t N)
{
    if (out == nullptr) return LORAHIP_E_INVALID;
    *out = nullptr;
    int sf = -1;
    for (int s = LORAHIP_SF_MIN; s <= LORAHIP_SF_MAX; s++) if ((size_t(1) << s) == N) sf = s;
    if (sf < 0) return LORAHIP_E_INVALID;
    lorahip_detector *det = new (std::nothrow) lorahip_detector();
    if (det == nullptr) return LORAHIP_E_NOMEM;
    det->ctx = nullptr;
    det->N = N;
    det->block = nullptr;
    const int rc = lorahip_create(&det->ctx, device, sf);
    if (rc != LORAHIP_OK) { delete det; return rc; }
    {
        const DeviceGuard guard(det->ctx->device);
        const size_t bytes = 2 * N * sizeof(cf32) + 256;
        void *p = nullptr;
        const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable);
        if (e != hipSuccess) { (void)hipGetLastError(); lorahip_destroy(det->ctx); delete det; return hipFail(e, "hipHostMalloc (detector staging)"); }
        std::memset(p, 0, bytes);
        det->block = static_cast<char *>(p);
    }
    det->input = reinterpret_cast<cf32 *>(det->block);
    det->output = det->input + N;
    det->res = reinterpret_cast<float *>(det->output + N);
    det->sym = reinterpret_cast<uint16_t *>(det->res + 4);
    *out = det;
    return LORAHIP_OK;
}

void lorahip_detector_destroy(lorahip_detector *det)
{
    if (det == nullptr) return;
    if (det->block)
    {
        const DeviceGuard guard(det->ctx->device);
        (void)hipStreamSynchronize(det->ctx->stream);
        (void)hipHostFree(det->block);
    }
    lorahip_destroy(det->ctx);
    delete det;
}

int lorahip_detector_feed(lorahip_detector *det, const size_t i, const float re, const float im)
{
    if (det == nullptr || i >= det->N) return LORAHIP_E_INVALID;
    det->input[i] = cf32(re, im);
    return LORAHIP_OK;
}

int lorahip_detector_detect(lorahip_detector *det, size_t *index, float *power, float *power_avg,
                            float *f_index, float *fft_out)
{
    if (det == nullptr || !index || !power || !power_avg || !f_index) return LORAHIP_E_INVALID;
    lorahip_batch b;
    std::memset(&b, 0, sizeof(b));
    b.struct_size = sizeof(b);
    b.iq = reinterpret_cast<const float *>(det->input);
    b.n_windows = 1;
    b.chirp_sel_all = LORAHIP_CHIRP_NONE;
    b.sym = det->sym;
    b.power = det->res;
    b.power_avg = det->res + 1;
    b.f_index = det->res + 2;
    b.fft_out = reinterpret_cast<float *>(det->output);
    const int rc = lorahip_detect_batch(det->ctx, &b);          // the device-pointer entry: the block is addressable from the device
    if (rc != LORAHIP_OK) return rc;
    {
        const DeviceGuard guard(det->ctx->device);
        LORAHIP_TRY(hipStreamSynchronize(det->ctx->stream));
    }
    *index = *det->sym;
    *power = det->res[0]; *power_avg = det->res[1]; *f_index = det->res[2];
    if (fft_out) std::memcpy(fft_out, det->output, det->N * sizeof(cf32));
    return LORAHIP_OK;
}

} // extern "C"
