// Level 3 over mixed spreading factors and several devices: ONE lorahip_demod handle whose channels each have their own SF, like
// the reference's one LoRaDemod block per channel (LoRaDemod.cpp:119-122: make(sf)), spread over the GPUs of a node (SURVEY.md
// section 8e: channels are independent units, no data-path collective).
//
// A streaming launch is uniform in N, so the object is a container of PARTS: lorahip_shard_plan assigns every channel a device
// (byte-weighted, the rule of lora_sdr_amd/shard.py), and on each device the channels of one SF form one plain lorahip_demod with
// its own level-2 context and HIP stream. Every part has a host thread that issues its runs, so the parts of one device overlap on
// it and the devices run side by side. Nothing here touches the data path: a part's channels go through the very kernels a
// single-SF object's would, and every accessor maps global channel numbers to (part, local channel).
#include "lorahip_internal.h"
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

namespace lorahip {

namespace {
//! a part's host thread: sleeps until a task is posted, runs it, keeps its code and error text (lorahip_last_error is per thread)
struct PartWorker
{
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool have = false, done = true, quit = false;
    std::function<int()> task;
    int rc = LORAHIP_OK;
    std::string err;
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu);
        while (true)
        {
            cv.wait(lk, [this] { return have || quit; });
            if (quit) return;
            have = false;
            lk.unlock();
            const int r = task();
            const std::string e = r == LORAHIP_OK ? std::string() : std::string(lorahip_last_error());
            lk.lock();
            rc = r; err = e; done = true;
            cv.notify_all();
        }
    }
};
} // namespace

struct Composite::Impl
{
    struct Part
    {
        lorahip_demod *d = nullptr;
        int device = 0, deviceSlot = 0, sf = 0;
        std::vector<uint32_t> chan;             // global channel numbers, ascending
        PartWorker *w = nullptr;
    };
    std::vector<Part> parts;
    std::vector<int> devices;
    std::vector<int32_t> sf;                    // per global channel
    std::vector<uint32_t> partOf, localOf;      // per global channel
    size_t B = 0;
    bool uniformSf = true;

    ~Impl()
    {
        for (Part &p : parts)
        {
            if (p.w)
            {
                { std::lock_guard<std::mutex> g(p.w->mu); p.w->quit = true; }
                p.w->cv.notify_all();
                if (p.w->th.joinable()) p.w->th.join();
                delete p.w;
            }
            if (p.d) lorahip_demod_destroy(p.d);
        }
    }

    //! fn(part index) on every part, each on its own host thread; the first failure's code, its text as this thread's last error
    int onAll(const std::function<int(size_t)> &fn)
    {
        if (parts.size() == 1) return fn(0);
        for (size_t i = 0; i < parts.size(); i++)
        {
            PartWorker *w = parts[i].w;
            { std::lock_guard<std::mutex> g(w->mu); w->task = [&fn, i] { return fn(i); }; w->have = true; w->done = false; }
            w->cv.notify_all();
        }
        int rc = LORAHIP_OK;
        for (size_t i = 0; i < parts.size(); i++)
        {
            PartWorker *w = parts[i].w;
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return w->done; });
            if (w->rc != LORAHIP_OK && rc == LORAHIP_OK) { rc = w->rc; setLastError(w->err); }
        }
        return rc;
    }

    //! the same on the calling thread, part after part (setters, accessors)
    int each(const std::function<int(lorahip_demod *)> &fn)
    {
        for (Part &p : parts) { const int rc = fn(p.d); if (rc != LORAHIP_OK) return rc; }
        return LORAHIP_OK;
    }
};

Composite::Composite() : p(nullptr) {}
Composite::~Composite() { delete p; }

int Composite::create(Composite **out, const int *devices, const size_t nDev, const int32_t *channelSf, const size_t n)
{
    if (out == nullptr || devices == nullptr || nDev == 0 || nDev > 4096 || channelSf == nullptr || n == 0 || n > 0x7fffffffu) return LORAHIP_E_INVALID;
    *out = nullptr;
    for (size_t c = 0; c < n; c++) if (channelSf[c] < LORAHIP_SF_MIN || channelSf[c] > LORAHIP_SF_MAX) return LORAHIP_E_INVALID;
    Composite *k = nullptr;
    try
    {
        k = new Composite();
        k->p = new Impl();
        Impl &I = *k->p;
        I.B = n;
        I.devices.assign(devices, devices + nDev);
        I.sf.assign(channelSf, channelSf + n);
        for (size_t c = 1; c < n; c++) I.uniformSf = I.uniformSf && channelSf[c] == channelSf[0];
        std::vector<int32_t> shard(n, 0);
        const int prc = lorahip_shard_plan(channelSf, n, nDev, shard.data());
        if (prc != LORAHIP_OK) { delete k; return prc; }
        I.partOf.assign(n, 0); I.localOf.assign(n, 0);
        for (size_t s = 0; s < nDev; s++)
            for (int sf = LORAHIP_SF_MIN; sf <= LORAHIP_SF_MAX; sf++)
            {
                Impl::Part part;
                part.device = devices[s]; part.deviceSlot = int(s); part.sf = sf;
                for (size_t c = 0; c < n; c++) if (size_t(shard[c]) == s && channelSf[c] == sf) part.chan.push_back(uint32_t(c));
                if (part.chan.empty()) continue;
                // the slot first, the object into it: a part that exists is always owned by Impl (whose destructor destroys it),
                // whatever throws afterwards
                I.parts.push_back(std::move(part));
                Impl::Part &slot = I.parts.back();
                const int rc = lorahip_demod_create(&slot.d, slot.device, sf, slot.chan.size());
                if (rc != LORAHIP_OK) { const std::string e = lorahip_last_error(); delete k; setLastError(e); return rc; }
                for (size_t j = 0; j < slot.chan.size(); j++) { I.partOf[slot.chan[j]] = uint32_t(I.parts.size() - 1); I.localOf[slot.chan[j]] = uint32_t(j); }
            }
        // The parts of one device run side by side (runSegments: a stream and a host thread each), so a part's launch does not have the
        // device to itself: tell each how many wavefronts its siblings bring (at 16 points per lane: 2^(sf - 4) lanes per channel),
        // and its choice of a wider geometry (lorahip_stream_lanes.hip) is made into the slots that are left. BASELINE configs[3]:
        // 2731 channels per SF fill the device several times over -- every part on 16 points per lane, +2 % (profiles/r06/s40_*).
        for (Impl::Part &part : I.parts)
        {
            unsigned long long others = 0;
            for (const Impl::Part &q : I.parts)
                if (&q != &part && q.device == part.device) others += ((unsigned long long)q.chan.size() << (q.sf - 4)) / 64u + 1u;
            demodSetCoResidentWaves(part.d, others > 0xffffffffull ? 0xffffffffu : unsigned(others));
        }
        if (I.parts.size() > 1)
            for (Impl::Part &part : I.parts)
            {
                part.w = new PartWorker();
                part.w->th = std::thread(&PartWorker::loop, part.w);
            }
    }
    catch (const std::bad_alloc &) { delete k; return LORAHIP_E_NOMEM; }
    catch (const std::exception &e) { delete k; setLastError(e.what()); return LORAHIP_E_HIP; }   // a thread could not be started
    *out = k;
    return LORAHIP_OK;
}

size_t Composite::numChannels() const { return p->B; }
size_t Composite::numParts() const { return p->parts.size(); }
int Composite::partInfo(const size_t i, int32_t *device, int32_t *sf, size_t *nChannels, int32_t *deviceSlot) const
{
    if (i >= p->parts.size()) return LORAHIP_E_INVALID;
    const Impl::Part &q = p->parts[i];
    if (device) *device = q.device;
    if (sf) *sf = q.sf;
    if (nChannels) *nChannels = q.chan.size();
    if (deviceSlot) *deviceSlot = q.deviceSlot;
    return LORAHIP_OK;
}
int Composite::partOf(int32_t *part, int32_t *local) const
{
    for (size_t c = 0; c < p->B; c++) { if (part) part[c] = int32_t(p->partOf[c]); if (local) local[c] = int32_t(p->localOf[c]); }
    return LORAHIP_OK;
}
lorahip_demod *Composite::part(const size_t i) const { return i < p->parts.size() ? p->parts[i].d : nullptr; }

int Composite::setSync(const unsigned char v) { return p->each([v](lorahip_demod *d) { return lorahip_demod_set_sync(d, v); }); }
int Composite::setThreshold(const double v) { return p->each([v](lorahip_demod *d) { return lorahip_demod_set_threshold(d, v); }); }
int Composite::setMtu(const size_t v) { return p->each([v](lorahip_demod *d) { return lorahip_demod_set_mtu(d, v); }); }
int Composite::setMode(const int v) { return p->each([v](lorahip_demod *d) { return lorahip_demod_set_mode(d, v); }); }
int Composite::setFineGather(const int v) { return p->each([v](lorahip_demod *d) { return lorahip_demod_set_fine_gather(d, v); }); }
int Composite::setTrace(const int v) { return p->each([v](lorahip_demod *d) { return lorahip_demod_set_trace(d, v); }); }
int Composite::setSignals(const int v) { return p->each([v](lorahip_demod *d) { return lorahip_demod_set_signals(d, v); }); }
int Composite::activate() { return p->each([](lorahip_demod *d) { return lorahip_demod_activate(d); }); }

int Composite::setStream(void *stream)
{
    // a HIP stream belongs to one device: an object that spans several keeps its parts' private streams
    for (const Impl::Part &q : p->parts) if (q.device != p->parts[0].device) { setLastError("lorahip_demod_set_stream: the object spans several devices"); return LORAHIP_E_INVALID; }
    return p->each([stream](lorahip_demod *d) { return lorahip_demod_set_stream(d, stream); });
}
int Composite::streamWait(void *stream)
{
    for (const Impl::Part &q : p->parts) if (q.device != p->parts[0].device) { setLastError("lorahip_demod_stream_wait: the object spans several devices"); return LORAHIP_E_INVALID; }
    return p->each([stream](lorahip_demod *d) { return lorahip_demod_stream_wait(d, stream); });
}
int Composite::streamFollow(void *stream)
{
    for (const Impl::Part &q : p->parts) if (q.device != p->parts[0].device) { setLastError("lorahip_demod_stream_follow: the object spans several devices"); return LORAHIP_E_INVALID; }
    return p->each([stream](lorahip_demod *d) { return lorahip_demod_stream_follow(d, stream); });
}
int Composite::resetStream() { return p->each([](lorahip_demod *d) { return lorahip_demod_reset_stream(d); }); }

int Composite::run(const float *const *streams, const size_t *nSamples, int64_t *rounds)
{
    try
    {
        Impl &I = *p;
        std::vector<std::vector<const float *>> ptr(I.parts.size());
        std::vector<std::vector<size_t>> len(I.parts.size());
        std::vector<int64_t> rd(I.parts.size(), 0);
        for (size_t i = 0; i < I.parts.size(); i++)
        {
            const auto &ch = I.parts[i].chan;
            ptr[i].resize(ch.size()); len[i].resize(ch.size());
            for (size_t j = 0; j < ch.size(); j++) { ptr[i][j] = streams[ch[j]]; len[i][j] = nSamples[ch[j]]; }
        }
        const int rc = I.onAll([&](const size_t i) { return lorahip_demod_run(I.parts[i].d, ptr[i].data(), len[i].data(), &rd[i]); });
        if (rounds) *rounds = *std::max_element(rd.begin(), rd.end());
        return rc;
    }
    catch (const std::bad_alloc &) { setLastError("out of host memory"); return LORAHIP_E_NOMEM; }        // nothing may cross the C ABI
}

int Composite::runSegments(const float *const *iqPerDevice, const size_t nDev, const int64_t *first, const size_t *nSamples, int64_t *rounds)
{
    try
    {
        Impl &I = *p;
        if (nDev != I.devices.size()) { setLastError("one device buffer per entry of the device list"); return LORAHIP_E_INVALID; }
        std::vector<std::vector<int64_t>> fs(I.parts.size());
        std::vector<std::vector<size_t>> len(I.parts.size());
        std::vector<int64_t> rd(I.parts.size(), 0);
        for (size_t i = 0; i < I.parts.size(); i++)
        {
            const auto &ch = I.parts[i].chan;
            fs[i].resize(ch.size()); len[i].resize(ch.size());
            for (size_t j = 0; j < ch.size(); j++) { fs[i][j] = first[ch[j]]; len[i][j] = nSamples[ch[j]]; }
        }
        const int rc = I.onAll([&](const size_t i) {
            return lorahip_demod_run_device_segments(I.parts[i].d, iqPerDevice[I.parts[i].deviceSlot], fs[i].data(), len[i].data(), &rd[i]); });
        if (rounds) *rounds = *std::max_element(rd.begin(), rd.end());
        return rc;
    }
    catch (const std::bad_alloc &) { setLastError("out of host memory"); return LORAHIP_E_NOMEM; }        // nothing may cross the C ABI
}

size_t Composite::numPackets() const { size_t n = 0; for (const auto &q : p->parts) n += lorahip_demod_num_packets(q.d); return n; }
size_t Composite::numPacketSymbols() const { size_t n = 0; for (const auto &q : p->parts) n += lorahip_demod_num_packet_symbols(q.d); return n; }
size_t Composite::numSignals() const { size_t n = 0; for (const auto &q : p->parts) n += lorahip_demod_num_signals(q.d); return n; }

int Composite::getPacket(size_t i, int32_t *channel, int64_t *round, size_t *len, int16_t *out, const size_t cap) const
{
    for (const auto &q : p->parts)
    {
        const size_t n = lorahip_demod_num_packets(q.d);
        if (i < n)
        {
            int32_t ch = 0;
            const int rc = lorahip_demod_get_packet(q.d, i, &ch, round, len, out, cap);
            if (rc == LORAHIP_OK && channel) *channel = int32_t(q.chan[size_t(ch)]);
            return rc;
        }
        i -= n;
    }
    return LORAHIP_E_INVALID;
}

int Composite::getPackets(int32_t *channels, int64_t *rounds, int64_t *lens, const size_t capPackets, int16_t *syms, const size_t capSyms) const
{
    try
    {
        if (capPackets < numPackets() || capSyms < numPacketSymbols()) return LORAHIP_E_INVALID;
        size_t at = 0, sat = 0;
        std::vector<int32_t> ch;
        for (const auto &q : p->parts)
        {
            const size_t n = lorahip_demod_num_packets(q.d), ns = lorahip_demod_num_packet_symbols(q.d);
            ch.resize(n);
            const int rc = lorahip_demod_get_packets(q.d, ch.data(), rounds ? rounds + at : nullptr, lens ? lens + at : nullptr, n, syms ? syms + sat : nullptr, ns);
            if (rc != LORAHIP_OK) return rc;
            if (channels) for (size_t j = 0; j < n; j++) channels[at + j] = int32_t(q.chan[size_t(ch[j])]);
            at += n; sat += ns;
        }
        return LORAHIP_OK;
    }
    catch (const std::bad_alloc &) { setLastError("out of host memory"); return LORAHIP_E_NOMEM; }        // nothing may cross the C ABI
}

int Composite::getSignals(int32_t *channels, int64_t *rounds, int32_t *errors, float *powers, float *snrs, const size_t cap) const
{
    try
    {
        if (cap < numSignals()) return LORAHIP_E_INVALID;
        size_t at = 0;
        std::vector<int32_t> ch;
        for (const auto &q : p->parts)
        {
            const size_t n = lorahip_demod_num_signals(q.d);
            ch.resize(n);
            const int rc = lorahip_demod_get_signals(q.d, ch.data(), rounds ? rounds + at : nullptr, errors ? errors + at : nullptr, powers ? powers + at : nullptr,
                                                     snrs ? snrs + at : nullptr, n);
            if (rc != LORAHIP_OK) return rc;
            if (channels) for (size_t j = 0; j < n; j++) channels[at + j] = int32_t(q.chan[size_t(ch[j])]);
            at += n;
        }
        return LORAHIP_OK;
    }
    catch (const std::bad_alloc &) { setLastError("out of host memory"); return LORAHIP_E_NOMEM; }        // nothing may cross the C ABI
}

void Composite::clearPackets() { for (auto &q : p->parts) lorahip_demod_clear_packets(q.d); }

int64_t Composite::consumed(const size_t c) const
{
    if (c >= p->B) return LORAHIP_E_INVALID;
    return lorahip_demod_consumed(p->parts[p->partOf[c]].d, p->localOf[c]);
}

int Composite::consumedAll(int64_t *out) const
{
    try
    {
        std::vector<int64_t> tmp;
        for (const auto &q : p->parts)
        {
            tmp.resize(q.chan.size());
            const int rc = lorahip_demod_consumed_all(q.d, tmp.data());
            if (rc != LORAHIP_OK) return rc;
            for (size_t j = 0; j < q.chan.size(); j++) out[q.chan[j]] = tmp[j];
        }
        return LORAHIP_OK;
    }
    catch (const std::bad_alloc &) { setLastError("out of host memory"); return LORAHIP_E_NOMEM; }        // nothing may cross the C ABI
}

int64_t Composite::workCalls() const { int64_t n = 0; for (const auto &q : p->parts) n += lorahip_demod_work_calls(q.d); return n; }
double Composite::kernelMs() const { double m = 0; for (const auto &q : p->parts) m = std::max(m, lorahip_demod_kernel_ms(q.d)); return m; }
int Composite::lastLaunches() const { int m = 0; for (const auto &q : p->parts) m = std::max(m, lorahip_demod_last_launches(q.d)); return m; }
int Composite::nearThreshold(int64_t *sq, int64_t *st) const
{
    int64_t a = 0, b = 0;
    for (const auto &q : p->parts)
    {
        int64_t x = 0, y = 0;
        const int rc = lorahip_demod_near_threshold(q.d, &x, &y);
        if (rc != LORAHIP_OK) return rc;
        a += x; b += y;
    }
    if (sq) *sq = a;
    if (st) *st = b;
    return LORAHIP_OK;
}

int Composite::setPorts(const lorahip_demod_ports *ports)
{
    Impl &I = *p;
    if (ports == nullptr) return I.each([](lorahip_demod *d) { return lorahip_demod_set_ports(d, nullptr); });
    // the port arrays are [channel][capacity][N]: one N, host buffers (device arrays cannot span devices), and parts that hold
    // contiguous channel ranges -- which is what lorahip_shard_plan gives an object of one SF
    if (ports->struct_size != sizeof(lorahip_demod_ports) || !I.uniformSf || !ports->host_buffers)
    {
        setLastError("debug ports on an object of several parts need one SF and host buffers");
        return LORAHIP_E_INVALID;
    }
    const size_t N = size_t(1) << I.sf[0];
    for (const auto &q : I.parts)
    {
        for (size_t j = 1; j < q.chan.size(); j++) if (q.chan[j] != q.chan[0] + j) { setLastError("debug ports: a part's channels are not a contiguous range"); return LORAHIP_E_INVALID; }
        lorahip_demod_ports sub = *ports;
        const size_t first = q.chan[0];
        if (sub.fft_dev) sub.fft_dev += 2 * first * sub.fft_cap_frames * N;
        if (sub.dec_dev) sub.dec_dev += 2 * first * sub.dec_cap_samples;
        if (sub.raw_dev) sub.raw_dev += 2 * first * sub.raw_cap_samples;
        const int rc = lorahip_demod_set_ports(q.d, &sub);
        if (rc != LORAHIP_OK) return rc;
    }
    return LORAHIP_OK;
}

#define LORAHIP_RX_CHANNEL(c) if ((c) >= p->B) return LORAHIP_E_INVALID; lorahip_demod *d_ = p->parts[p->partOf[c]].d; const size_t l_ = p->localOf[c]
int Composite::portCounts(const size_t c, size_t *f, size_t *d, size_t *r) const { LORAHIP_RX_CHANNEL(c); return lorahip_demod_port_counts(d_, l_, f, d, r); }
int Composite::getLabels(const size_t c, char *buf, const size_t cap, size_t *n, size_t *bytes) const { LORAHIP_RX_CHANNEL(c); return lorahip_demod_get_labels(d_, l_, buf, cap, n, bytes); }
int Composite::getTrace(const size_t c, lorahip_work_result *out, const size_t cap) const { LORAHIP_RX_CHANNEL(c); return lorahip_demod_get_trace(d_, l_, out, cap); }
size_t Composite::traceLen(const size_t c) const { if (c >= p->B) return 0; return lorahip_demod_trace_len(p->parts[p->partOf[c]].d, p->localOf[c]); }
#undef LORAHIP_RX_CHANNEL

} // namespace lorahip
