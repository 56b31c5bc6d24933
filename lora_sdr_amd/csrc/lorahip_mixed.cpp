// Mixed-SF scheduler of the batch path (BASELINE configs[3]): channels of different spreading
// factors demodulated in one call. A launch is uniform in N, so the channels are bucketed by SF; every bucket owns a level-2
// context (tables, kernels) with its own HIP stream, the buckets' launches are issued back to back and run concurrently, and a
// step ends when every bucket's stream has passed its event. Nothing here touches the data path's arithmetic: a bucket's windows
// go through lorahip_detect_batch exactly as a single-SF caller's would.
#include "lorahip_internal.h"
#include <cstring>
#include <new>

using namespace lorahip;

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
};

extern "C" {

void lorahip_mixed_destroy(lorahip_mixed *m)
{
    if (m == nullptr) return;
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
    std::memcpy(row_of_channel, m->row.data(), m->nChannels * sizeof(int64_t));
    return LORAHIP_OK;
}

int lorahip_mixed_plan(lorahip_mixed *m, const int64_t *channel_offset, const size_t windows_per_channel)
{
    if (m == nullptr || channel_offset == nullptr || windows_per_channel == 0) return LORAHIP_E_INVALID;
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
    const DeviceGuard guard(m->device);
    for (auto &b : m->buckets) LORAHIP_TRY(hipEventSynchronize(b.done));
    return LORAHIP_OK;
}

} // extern "C"
