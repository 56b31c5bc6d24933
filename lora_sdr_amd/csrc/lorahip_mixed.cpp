// Mixed-SF scheduler of the batch path (BASELINE configs[3]): channels of different spreading
// factors demodulated in one call. A launch is uniform in N, so the channels are bucketed by SF; every bucket owns a level-2
// context (tables, kernels) with its own HIP stream, the buckets' launches are issued back to back and run concurrently, and a
// step ends when every bucket's stream has passed its event. Nothing here touches the data path's arithmetic: a bucket's windows
// go through lorahip_detect_batch exactly as a single-SF caller's would.
//
// Several devices ("one process, 8 devices, one host thread + stream per device"): lorahip_mixed_create_multi
// splits the channels over the devices with the byte-weighted rule of lora_sdr_amd/shard.py (lorahip_shard_plan, host-only), gives
// every device its own single-device scheduler and a host thread that issues that device's launches; a step ends when every
// device's events have passed (lorahip_mixed_synchronize). No data-path collective: channels are independent units.
#include "lorahip_internal.h"
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>

using namespace lorahip;

namespace {
//! one device's launch thread: sleeps until a step is posted, issues the shard's launches, reports the code
struct Worker
{
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool have = false, done = true, quit = false;
    const float *iq = nullptr; uint16_t *sym = nullptr; float *power = nullptr, *pavg = nullptr, *fidx = nullptr;
    int rc = LORAHIP_OK;
    std::string err;
};
} // namespace

struct lorahip_mixed
{
    int device;
    size_t nChannels, S;
    std::vector<int32_t> sf;                 // per channel
    std::vector<int64_t> row;                // per channel: its row in the [rows][S] result arrays (buckets by ascending SF)
    struct Bucket
    {
        int sf;
        lorahip_ctx *ctx;
        std::vector<uint32_t> channels;      // ascending
        size_t firstRow;
        int64_t *dOffsets;                   // [channels * S] start sample of every window
        hipEvent_t done;
    };
    std::vector<Bucket> buckets;
    bool planned;
    // several devices: the object is a container of single-device schedulers (buckets stays empty)
    std::vector<lorahip_mixed *> shards;     // one per entry of `devices`, nullptr where a shard got no channel
    std::vector<int> devices;
    std::vector<int32_t> shardOf;            // per channel
    std::vector<std::vector<uint32_t>> shardChannels;   // per shard: its channels ordered by SF, then channel number
    std::vector<uint32_t> localOf;           // per channel: its index among its shard's channels
    std::vector<Worker *> workers;
};

static void stopWorkers(lorahip_mixed *m)
{
    for (Worker *w : m->workers)
    {
        if (w == nullptr) continue;
        { std::lock_guard<std::mutex> g(w->mu); w->quit = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    m->workers.clear();
}

static void workerLoop(lorahip_mixed *shard, Worker *w)
{
    std::unique_lock<std::mutex> lk(w->mu);
    while (true)
    {
        w->cv.wait(lk, [w] { return w->have || w->quit; });
        if (w->quit) return;
        w->have = false;
        lk.unlock();
        const int rc = lorahip_mixed_detect(shard, w->iq, w->sym, w->power, w->pavg, w->fidx);
        const std::string err = rc == LORAHIP_OK ? std::string() : std::string(lorahip_last_error());   // last_error is per thread
        lk.lock();
        w->rc = rc; w->err = err; w->done = true;
        w->cv.notify_all();
    }
}

extern "C" {

int lorahip_shard_plan(const int32_t *channel_sf, const size_t n_channels, const size_t n_shards, int32_t *shard_of_channel)
{
    if ((n_channels && (channel_sf == nullptr || shard_of_channel == nullptr)) || n_shards == 0 || n_shards > (size_t(1) << 20)) return LORAHIP_E_INVALID;
    for (size_t c = 0; c < n_channels; c++) if (channel_sf[c] < 1 || channel_sf[c] > 24) return LORAHIP_E_INVALID;
    try
    {
    // lora_sdr_amd/shard.py::shard_channels, statement for statement: SF buckets from the largest windows down, each cut into
    // n_shards contiguous ranges of floor(count / n_shards) channels; the count % n_shards left-overs go to the shards that would
    // hold the fewest bytes (8*2^SF + 14 per symbol window), lowest shard first among equals
    std::vector<long long> load(n_shards, 0), counts(n_shards);
    std::vector<size_t> order(n_shards);
    for (int sf = 24; sf >= 1; sf--)
    {
        std::vector<size_t> chans;
        for (size_t c = 0; c < n_channels; c++) if (channel_sf[c] == sf) chans.push_back(c);
        if (chans.empty()) continue;
        const long long w = (8LL << sf) + 14;
        const size_t base = chans.size() / n_shards, rem = chans.size() % n_shards;
        for (size_t r = 0; r < n_shards; r++) { counts[r] = (long long)base; order[r] = r; }
        if (rem)
        {
            std::stable_sort(order.begin(), order.end(), [&](const size_t a, const size_t b) { return load[a] + counts[a] * w < load[b] + counts[b] * w; });
            for (size_t i = 0; i < rem; i++) counts[order[i]]++;
        }
        size_t start = 0;
        for (size_t r = 0; r < n_shards; r++)
        {
            for (long long i = 0; i < counts[r]; i++) shard_of_channel[chans[start + size_t(i)]] = int32_t(r);
            start += size_t(counts[r]);
            load[r] += counts[r] * w;
        }
    }
    }
    catch (const std::exception &) { return LORAHIP_E_NOMEM; }   // no exception crosses the C ABI
    return LORAHIP_OK;
}

void lorahip_mixed_destroy(lorahip_mixed *m)
{
    if (m == nullptr) return;
    stopWorkers(m);
    for (lorahip_mixed *sh : m->shards) lorahip_mixed_destroy(sh);
    m->shards.clear();
    {
        const DeviceGuard guard(m->device);
        for (auto &b : m->buckets)
        {
            if (b.dOffsets) (void)hipFree(b.dOffsets);
            if (b.done) (void)hipEventDestroy(b.done);
        }
    }
    for (auto &b : m->buckets) lorahip_destroy(b.ctx);
    delete m;
}

int lorahip_mixed_create(lorahip_mixed **out, const int device, const int32_t *channel_sf, const size_t n_channels)
{
    if (out == nullptr || channel_sf == nullptr || n_channels == 0 || n_channels > 0x7fffffffu) return LORAHIP_E_INVALID;
    *out = nullptr;
    for (size_t c = 0; c < n_channels; c++)
        if (channel_sf[c] < LORAHIP_SF_MIN || channel_sf[c] > LORAHIP_SF_MAX) return LORAHIP_E_INVALID;
    lorahip_mixed *m = new (std::nothrow) lorahip_mixed();
    if (m == nullptr) return LORAHIP_E_NOMEM;
    m->device = device; m->nChannels = n_channels; m->S = 0; m->planned = false;
    m->sf.assign(channel_sf, channel_sf + n_channels);
    m->row.assign(n_channels, -1);
    size_t rows = 0;
    for (int sf = LORAHIP_SF_MIN; sf <= LORAHIP_SF_MAX; sf++)
    {
        lorahip_mixed::Bucket b;
        b.sf = sf; b.ctx = nullptr; b.firstRow = rows; b.dOffsets = nullptr; b.done = nullptr;
        for (size_t c = 0; c < n_channels; c++) if (channel_sf[c] == sf) { m->row[c] = int64_t(rows++); b.channels.push_back(uint32_t(c)); }
        if (b.channels.empty()) continue;
        const int rc = lorahip_create(&b.ctx, device, sf);          // private non-blocking stream: the buckets overlap
        if (rc != LORAHIP_OK) { lorahip_mixed_destroy(m); return rc; }
        m->buckets.push_back(b);
        const DeviceGuard guard(device);
        if (hipEventCreateWithFlags(&m->buckets.back().done, hipEventDisableTiming) != hipSuccess) { lorahip_mixed_destroy(m); return LORAHIP_E_HIP; }
    }
    *out = m;
    return LORAHIP_OK;
}

int lorahip_mixed_create_multi(lorahip_mixed **out, const int *devices, const size_t n_devices, const int32_t *channel_sf, const size_t n_channels)
{
    if (out == nullptr || devices == nullptr || n_devices == 0 || n_devices > 1024 || channel_sf == nullptr || n_channels == 0 || n_channels > 0x7fffffffu)
        return LORAHIP_E_INVALID;
    *out = nullptr;
    for (size_t c = 0; c < n_channels; c++)
        if (channel_sf[c] < LORAHIP_SF_MIN || channel_sf[c] > LORAHIP_SF_MAX) return LORAHIP_E_INVALID;
    lorahip_mixed *m = new (std::nothrow) lorahip_mixed();
    if (m == nullptr) return LORAHIP_E_NOMEM;
    int rc = LORAHIP_OK;
    try
    {
        m->device = devices[0]; m->nChannels = n_channels; m->S = 0; m->planned = false;
        m->sf.assign(channel_sf, channel_sf + n_channels);
        m->devices.assign(devices, devices + n_devices);
        m->shardOf.assign(n_channels, 0);
        m->localOf.assign(n_channels, 0);
        m->shardChannels.assign(n_devices, std::vector<uint32_t>());
        rc = lorahip_shard_plan(channel_sf, n_channels, n_devices, m->shardOf.data());
        // inside a shard: by SF, then by channel number (one launch per SF bucket; shard.py returns the same order)
        for (int sf = LORAHIP_SF_MIN; rc == LORAHIP_OK && sf <= LORAHIP_SF_MAX; sf++)
            for (size_t c = 0; c < n_channels; c++)
                if (channel_sf[c] == sf) { auto &v = m->shardChannels[size_t(m->shardOf[c])]; m->localOf[c] = uint32_t(v.size()); v.push_back(uint32_t(c)); }
        m->shards.assign(n_devices, nullptr);
        m->workers.assign(n_devices, nullptr);
        for (size_t s = 0; rc == LORAHIP_OK && s < n_devices; s++)
        {
            const auto &ch = m->shardChannels[s];
            if (ch.empty()) continue;
            std::vector<int32_t> sfs(ch.size());
            for (size_t i = 0; i < ch.size(); i++) sfs[i] = channel_sf[ch[i]];
            rc = lorahip_mixed_create(&m->shards[s], devices[s], sfs.data(), sfs.size());
            if (rc != LORAHIP_OK) break;
            m->workers[s] = new Worker();
            m->workers[s]->th = std::thread(workerLoop, m->shards[s], m->workers[s]);
        }
    }
    catch (const std::exception &e) { setLastError(std::string("lorahip_mixed_create_multi: ") + e.what()); rc = LORAHIP_E_NOMEM; }
    if (rc != LORAHIP_OK) { lorahip_mixed_destroy(m); return rc; }
    *out = m;
    return LORAHIP_OK;
}

size_t lorahip_mixed_num_devices(const lorahip_mixed *m) { return m == nullptr ? 0 : (m->shards.empty() ? 1 : m->shards.size()); }

int lorahip_mixed_device(const lorahip_mixed *m, const size_t shard, int32_t *device, size_t *n_channels)
{
    if (m == nullptr || shard >= lorahip_mixed_num_devices(m)) return LORAHIP_E_INVALID;
    if (device) *device = m->shards.empty() ? m->device : m->devices[shard];
    if (n_channels) *n_channels = m->shards.empty() ? m->nChannels : m->shardChannels[shard].size();
    return LORAHIP_OK;
}

lorahip_mixed *lorahip_mixed_shard(const lorahip_mixed *m, const size_t shard)
{
    if (m == nullptr || m->shards.empty()) return shard == 0 ? const_cast<lorahip_mixed *>(m) : nullptr;
    return shard < m->shards.size() ? m->shards[shard] : nullptr;
}

int lorahip_mixed_shard_of(const lorahip_mixed *m, int32_t *shard_of_channel)
{
    if (m == nullptr || shard_of_channel == nullptr) return LORAHIP_E_INVALID;
    if (m->shards.empty()) std::memset(shard_of_channel, 0, m->nChannels * sizeof(int32_t));
    else std::memcpy(shard_of_channel, m->shardOf.data(), m->nChannels * sizeof(int32_t));
    return LORAHIP_OK;
}

int lorahip_mixed_detect_multi(lorahip_mixed *m, const float *const *iq_dev, uint16_t *const *sym_dev, float *const *power_dev,
                               float *const *power_avg_dev, float *const *f_index_dev)
{
    if (m == nullptr || !m->planned || !iq_dev || !sym_dev || !power_dev || !power_avg_dev || !f_index_dev) return LORAHIP_E_INVALID;
    if (m->shards.empty()) return lorahip_mixed_detect(m, iq_dev[0], sym_dev[0], power_dev[0], power_avg_dev[0], f_index_dev[0]);
    // every device's launches are issued by that device's own host thread, all at once; this call returns when all are ISSUED
    // (asynchronous like lorahip_mixed_detect; lorahip_mixed_synchronize is the join on the devices' events)
    for (size_t s = 0; s < m->shards.size(); s++)
    {
        Worker *w = m->workers[s];
        if (w == nullptr) continue;
        { std::lock_guard<std::mutex> g(w->mu);
          w->iq = iq_dev[s]; w->sym = sym_dev[s]; w->power = power_dev[s]; w->pavg = power_avg_dev[s]; w->fidx = f_index_dev[s];
          w->have = true; w->done = false; }
        w->cv.notify_all();
    }
    int rc = LORAHIP_OK;
    for (size_t s = 0; s < m->shards.size(); s++)
    {
        Worker *w = m->workers[s];
        if (w == nullptr) continue;
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [w] { return w->done; });
        if (w->rc != LORAHIP_OK && rc == LORAHIP_OK) { rc = w->rc; setLastError(w->err); }
    }
    return rc;
}

size_t lorahip_mixed_num_buckets(const lorahip_mixed *m) { return m ? m->buckets.size() : 0; }

int lorahip_mixed_bucket(const lorahip_mixed *m, const size_t i, int32_t *sf, size_t *first_row, size_t *n_channels)
{
    if (m == nullptr || i >= m->buckets.size()) return LORAHIP_E_INVALID;
    if (sf) *sf = m->buckets[i].sf;
    if (first_row) *first_row = m->buckets[i].firstRow;
    if (n_channels) *n_channels = m->buckets[i].channels.size();
    return LORAHIP_OK;
}

lorahip_ctx *lorahip_mixed_context(const lorahip_mixed *m, const size_t i)
{
    return (m == nullptr || i >= m->buckets.size()) ? nullptr : m->buckets[i].ctx;
}

int lorahip_mixed_rows(const lorahip_mixed *m, int64_t *row_of_channel)
{
    if (m == nullptr || row_of_channel == nullptr) return LORAHIP_E_INVALID;
    if (!m->shards.empty())
    {
        // the row inside the channel's own shard (every device has its own result arrays)
        for (size_t c = 0; c < m->nChannels; c++) row_of_channel[c] = m->shards[size_t(m->shardOf[c])]->row[m->localOf[c]];
        return LORAHIP_OK;
    }
    std::memcpy(row_of_channel, m->row.data(), m->nChannels * sizeof(int64_t));
    return LORAHIP_OK;
}

int lorahip_mixed_plan(lorahip_mixed *m, const int64_t *channel_offset, const size_t windows_per_channel)
{
    if (m == nullptr || channel_offset == nullptr || windows_per_channel == 0) return LORAHIP_E_INVALID;
    if (!m->shards.empty())
    {
        // offsets are relative to the IQ buffer of the channel's own device
        std::vector<int64_t> local;
        for (size_t s = 0; s < m->shards.size(); s++)
        {
            if (m->shards[s] == nullptr) continue;
            const auto &ch = m->shardChannels[s];
            local.resize(ch.size());
            for (size_t i = 0; i < ch.size(); i++) local[i] = channel_offset[ch[i]];
            const int rc = lorahip_mixed_plan(m->shards[s], local.data(), windows_per_channel);
            if (rc != LORAHIP_OK) return rc;
        }
        m->S = windows_per_channel;
        m->planned = true;
        return LORAHIP_OK;
    }
    const DeviceGuard guard(m->device);
    std::vector<int64_t> off;
    for (auto &b : m->buckets)
    {
        const size_t N = size_t(1) << b.sf, W = b.channels.size() * windows_per_channel;
        if (W > 0xffffffffu) return LORAHIP_E_INVALID;
        off.resize(W);
        for (size_t i = 0; i < b.channels.size(); i++)
        {
            const int64_t base = channel_offset[b.channels[i]];
            if (base < 0) return LORAHIP_E_INVALID;
            for (size_t k = 0; k < windows_per_channel; k++) off[i * windows_per_channel + k] = base + int64_t(k * N);
        }
        if (b.dOffsets) { (void)hipFree(b.dOffsets); b.dOffsets = nullptr; }
        LORAHIP_TRY(hipMalloc((void **)&b.dOffsets, W * sizeof(int64_t)));
        LORAHIP_TRY(hipMemcpy(b.dOffsets, off.data(), W * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    m->S = windows_per_channel;
    m->planned = true;
    return LORAHIP_OK;
}

int lorahip_mixed_detect(lorahip_mixed *m, const float *iq_dev, uint16_t *sym_dev, float *power_dev, float *power_avg_dev, float *f_index_dev)
{
    if (m == nullptr || !m->planned || iq_dev == nullptr || !sym_dev || !power_dev || !power_avg_dev || !f_index_dev) return LORAHIP_E_INVALID;
    if (!m->shards.empty()) return LORAHIP_E_INVALID;          // several devices: one buffer per device, lorahip_mixed_detect_multi
    const DeviceGuard guard(m->device);
    for (size_t i = m->buckets.size(); i-- > 0;)        // largest windows first: the long launches start early, the short ones fill in
    {
        auto &b = m->buckets[i];
        const size_t r0 = b.firstRow * m->S;
        lorahip_batch q;
        std::memset(&q, 0, sizeof(q));
        q.struct_size = sizeof(q);
        q.iq = iq_dev;
        q.n_windows = b.channels.size() * m->S;
        q.offsets = b.dOffsets;
        q.chirp_sel_all = LORAHIP_CHIRP_UP;
        q.sym = sym_dev + r0; q.power = power_dev + r0; q.power_avg = power_avg_dev + r0; q.f_index = f_index_dev + r0;
        const int rc = lorahip_detect_batch(b.ctx, &q);
        if (rc != LORAHIP_OK) return rc;
        LORAHIP_TRY(hipEventRecord(b.done, b.ctx->stream));
    }
    return LORAHIP_OK;
}

int lorahip_mixed_synchronize(lorahip_mixed *m)
{
    if (m == nullptr) return LORAHIP_E_INVALID;
    if (!m->shards.empty())
    {
        for (lorahip_mixed *sh : m->shards) if (sh) { const int rc = lorahip_mixed_synchronize(sh); if (rc != LORAHIP_OK) return rc; }
        return LORAHIP_OK;
    }
    const DeviceGuard guard(m->device);
    for (auto &b : m->buckets) LORAHIP_TRY(hipEventSynchronize(b.done));
    return LORAHIP_OK;
}

} // extern "C"
