// TEST INFRASTRUCTURE ONLY (oracle/) -- never linked or called by the product path.
//
// extern "C" driver of the multi-channel Pothos block lora_sdr_amd/pothos/LoRaDemodBatch.cpp, compiled against the recording
// fake of oracle/stub/Pothos and linked with liblorahip.so into oracle/_ref/libloradrop.so. The same .so carries
// oracle/ref_driver.cpp around the reference's OWN LoRaDemod.cpp with the two-line patch of INTEGRATION.md section 1 applied
// (LoRaDetector<float> -> LoRaDetectorHip<float>; the patched copy is a build product under oracle/_ref/, never committed),
// so the `loraref_demod_*` entry points of this library run the verbatim block on the HIP detector.
// tests/test_gpu_dropin.py compares both with the recorded behaviour of the unpatched reference.
#include <Pothos/Framework.hpp>
#include <chrono>
#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

typedef std::complex<float> cf32;

namespace {

struct ChanLog
{
    std::vector<cf32> raw, dec, fft;
    std::vector<std::pair<size_t, std::string>> rawLabels, decLabels, fftLabels;    // absolute element index, id
    std::vector<std::vector<int16_t>> packets;
    size_t consumed;
};

struct BatchHandle
{
    Pothos::Block *block;
    size_t N, B, maxWindows;
    std::vector<std::vector<cf32>> rawBuf, decBuf, fftBuf;      // the framework's port buffers
    std::vector<ChanLog> log;
    int64_t works;
    bool ports, ready;
    std::vector<size_t> need;                                   // per channel: 2N, what a work() call of that channel wants (LoRaDemod.cpp:148)
    bool slabs;                                                 // inputs through the block's own input buffer managers (getInputBufferManager)
    std::vector<Pothos::BufferManager::Sptr> mgr;               // ... one per input port
};

/*! The framework's side of the input ports, two ways. VIEWS: a port's buffer is a window into the caller's array (ordinary host
 * memory), the unconsumed remainder stays where it is. SLABS: every arrival is written into the next buffer of the port's own
 * manager -- the one the block returned from getInputBufferManager() -- behind the (< 2N) samples the block left unconsumed, which the
 * framework's accumulator carries over; the buffer before goes back to its manager. `sourceSeconds` collects the time spent writing
 * the arriving samples: the upstream block's work, not this block's. */
struct Feeder
{
    BatchHandle *h;
    const float *iq; size_t spc;
    std::vector<Pothos::InputPort *> in;                        // looked up once: the ports live in a map keyed by name
    std::vector<size_t> pos;                                    // samples consumed so far (absolute)
    std::vector<size_t> have;                                   // slabs: samples of the channel that have arrived
    std::vector<Pothos::BufferChunk> chunk;                     // slabs: the buffer being presented (holding it keeps it out of its manager's pool)
    std::vector<char *> cur; std::vector<size_t> curLen, curOff; // ... its memory, its valid samples, the read offset
    double sourceSeconds;
    Feeder(BatchHandle *h_, const float *iq_, const size_t spc_) : h(h_), iq(iq_), spc(spc_), in(h_->B), pos(h_->B, 0), have(h_->B, 0), chunk(h_->B), cur(h_->B, nullptr), curLen(h_->B, 0), curOff(h_->B, 0), sourceSeconds(0.0)
    {
        for (size_t c = 0; c < h->B; c++) in[c] = h->block->input(int(c));
    }
    // (a new stream on every port: the buffers the stream before still held went back to their managers when its Feeder -- the last
    // holder of their chunks -- was destroyed; that is how the framework returns buffers, ManagedBuffer's reference count)
    //! samples [have, w) of every channel arrive
    void arrive(const size_t w)
    {
        const size_t B = h->B;
        if (!h->slabs) { for (size_t c = 0; c < B; c++) have[c] = w; return; }
        for (size_t c = 0; c < B; c++)
        {
            Pothos::BufferManager &m = *h->mgr[c];
            const size_t rem = curLen[c] - curOff[c], add = w - have[c];
            if (m.empty() || m.front().length < (rem + add) * sizeof(cf32)) throw std::runtime_error("input slab too small for an arrival");
            Pothos::BufferChunk nextChunk = m.front();              // taken BEFORE pop(): the reference that keeps it out of the pool
            char *next = nextChunk.as<char *>();
            m.pop((rem + add) * sizeof(cf32));
            if (rem) std::memcpy(next, cur[c] + curOff[c] * sizeof(cf32), rem * sizeof(cf32));      // the accumulator's carry-over (< 2N samples)
            const auto t0 = std::chrono::steady_clock::now();
            std::memcpy(next + rem * sizeof(cf32), iq + 2 * (c * spc + have[c]), add * sizeof(cf32)); // the upstream block produces
            sourceSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            chunk[c] = nextChunk;                                   // the buffer before is free again: its last reference goes here
            cur[c] = next; curLen[c] = rem + add; curOff[c] = 0; have[c] = w;
        }
    }
    //! what the ports show the block now; true if some channel has a call's worth
    bool present(void)
    {
        bool any = false;
        for (size_t c = 0; c < h->B; c++)
        {
            Pothos::InputPort *p = in[c];
            if (h->slabs) { p->_elems = curLen[c] - curOff[c]; p->_buff = Pothos::BufferChunk::view(cur[c] + curOff[c] * sizeof(cf32), p->_elems * sizeof(cf32)); }
            else { p->_elems = have[c] - pos[c]; p->_buff = Pothos::BufferChunk::view(const_cast<float *>(iq) + 2 * (c * spc + pos[c]), p->_elems * sizeof(cf32)); }
            p->consumed = 0;
            any = any || p->_elems >= h->need[c];
        }
        return any;
    }
    size_t consumed(const size_t c)
    {
        const size_t used = in[c]->consumed;
        pos[c] += used; curOff[c] += used;
        return used;
    }
};

//! after the last setter, before the first work(): the framework allocates the port buffers the block's buffer-manager hook asks for
//! (only where the debug ports are on: 3 x B buffers), then activates the block
static void prepare(BatchHandle *h)
{
    if (h->ready) return;
    for (size_t c = 0; c < h->B && h->ports; c++)
    {
        const std::string s = std::to_string(c);
        const size_t nbRaw = h->block->getOutputBufferManager("raw" + s, "")->front().length;      // (the size of the buffers it hands out)
        const size_t nbFft = h->block->getOutputBufferManager("fft" + s, "")->front().length;
        h->rawBuf[c].resize(nbRaw / sizeof(cf32)); h->decBuf[c].resize(nbRaw / sizeof(cf32)); h->fftBuf[c].resize(nbFft / sizeof(cf32));
        h->block->output("raw" + s)->_buff = Pothos::BufferChunk::view(h->rawBuf[c].data(), nbRaw);
        h->block->output("dec" + s)->_buff = Pothos::BufferChunk::view(h->decBuf[c].data(), nbRaw);
        h->block->output("fft" + s)->_buff = Pothos::BufferChunk::view(h->fftBuf[c].data(), nbFft);
    }
    if (h->slabs)
    {
        // the framework asks the block for the buffer manager of every input port (LoRaDemod.cpp:346-357 is the reference's answer)
        h->mgr.resize(h->B);
        for (size_t c = 0; c < h->B; c++)
        {
            h->mgr[c] = h->block->getInputBufferManager(std::to_string(c), "");
            if (!h->mgr[c]) { h->mgr.clear(); h->slabs = false; break; }
        }
    }
    h->block->activate();
    h->ready = true;
}

} // namespace

extern "C" {

void *loradrop_batch_new(const size_t sf, const size_t channels, const size_t maxWindows)
{
    auto it = Pothos::BlockRegistry::table2().find("/lora/lora_demod_batch");
    if (it == Pothos::BlockRegistry::table2().end()) return nullptr;
    auto h = new BatchHandle();
    try { h->block = it->second(sf, channels); }
    catch (const std::exception &) { delete h; return nullptr; }
    h->N = size_t(1) << sf; h->B = channels; h->maxWindows = maxWindows; h->works = 0; h->ports = false; h->ready = false;
    h->block->calls.at("setMaxWindows")(double(maxWindows));
    h->rawBuf.resize(channels); h->decBuf.resize(channels); h->fftBuf.resize(channels); h->log.resize(channels);
    for (size_t c = 0; c < channels; c++) h->log[c].consumed = 0;
    h->need.assign(channels, 2 * h->N);
    h->slabs = false;
    return h;
}

void loradrop_batch_free(void *p)
{
    auto h = reinterpret_cast<BatchHandle *>(p);
    delete h->block;
    delete h;
}

int loradrop_batch_set(void *p, const char *name, const double v)
{
    auto h = reinterpret_cast<BatchHandle *>(p);
    auto it = h->block->calls.find(name);
    if (it == h->block->calls.end()) return -1;
    try { it->second(v); } catch (const std::exception &) { return -2; }
    if (std::string(name) == "setDebugPorts") h->ports = v != 0.0;
    return 0;
}

//! a registered getter of the block (slabRowRuns, workRuns, fftFramesDropped); -1 if there is none of that name
double loradrop_batch_get(void *p, const char *name)
{
    auto h = reinterpret_cast<BatchHandle *>(p);
    auto it = h->block->getters.find(name);
    return it == h->block->getters.end() ? -1.0 : it->second();
}

//! a registered call that takes a string (setDevices("0,1,..."))
int loradrop_batch_set_string(void *p, const char *name, const char *v)
{
    auto h = reinterpret_cast<BatchHandle *>(p);
    auto it = h->block->stringCalls.find(name);
    if (it == h->block->stringCalls.end()) return -1;
    try { it->second(v); } catch (const std::exception &) { return -2; }
    if (std::string(name) == "setSpreadFactors")
    {
        // the same list the block parsed (comma separated, repeating): the scheduler's per-channel reserve
        std::vector<size_t> pat;
        for (const char *q = v; *q; ) { pat.push_back(size_t(std::strtol(q, nullptr, 10))); while (*q && *q != ',') q++; if (*q == ',') q++; }
        for (size_t c = 0; c < h->B && !pat.empty(); c++) h->need[c] = size_t(2) << pat[c % pat.size()];
    }
    return 0;
}

//! every channel's whole stream (iq: [channels][samplesPerChannel] cf32): call work() the way a scheduler would -- each input
//! presents what the block has not consumed yet -- until no channel has 2N samples left. Returns the number of work() calls.
int64_t loradrop_batch_run(void *p, const float *iq, const size_t samplesPerChannel)
{
    auto h = reinterpret_cast<BatchHandle *>(p);
    const size_t B = h->B;
    prepare(h);
    Feeder f(h, iq, samplesPerChannel);
    // (slabs: the stream arrives in pieces a slab can hold; views: all at once)
    const size_t piece = h->slabs ? h->maxWindows * h->N : samplesPerChannel;
    try
    {
    for (size_t w = piece < samplesPerChannel ? piece : samplesPerChannel; ; w = w + piece < samplesPerChannel ? w + piece : samplesPerChannel)
    {
    f.arrive(w);
    while (true)
    {
        const bool any = f.present();
        for (size_t c = 0; c < B; c++)
        {
            const std::string s = std::to_string(c);
            for (const char *port : { "raw", "dec", "fft" }) { auto o = h->block->output(port + s); o->produced = 0; o->labels.clear(); }
        }
        if (!any) break;
        h->block->work();
        h->works++;
        size_t progressed = 0;
        for (size_t c = 0; c < B; c++)
        {
            const std::string s = std::to_string(c);
            ChanLog &L = h->log[c];
            auto raw = h->block->output("raw" + s), dec = h->block->output("dec" + s), fft = h->block->output("fft" + s), out = h->block->output(int(c));
            for (const auto &l : raw->labels) L.rawLabels.emplace_back(L.raw.size() + l.index, l.id);
            for (const auto &l : dec->labels) L.decLabels.emplace_back(L.dec.size() + l.index, l.id);
            for (const auto &l : fft->labels) L.fftLabels.emplace_back(L.fft.size() + l.index, l.id);
            L.raw.insert(L.raw.end(), h->rawBuf[c].begin(), h->rawBuf[c].begin() + long(raw->produced));
            L.dec.insert(L.dec.end(), h->decBuf[c].begin(), h->decBuf[c].begin() + long(dec->produced));
            L.fft.insert(L.fft.end(), h->fftBuf[c].begin(), h->fftBuf[c].begin() + long(fft->produced));
            for (const auto &bytes : out->messages)
            {
                std::vector<int16_t> syms(bytes.size() / sizeof(int16_t));
                if (!syms.empty()) std::memcpy(syms.data(), bytes.data(), syms.size() * sizeof(int16_t));
                L.packets.push_back(syms);
            }
            out->messages.clear();
            const size_t used = f.consumed(c);
            L.consumed += used; progressed += used;
        }
        if (progressed == 0) break;     // cannot happen; guards the loop
    }
    if (w >= samplesPerChannel) break;
    }
    }
    catch (const std::exception &) { return -2; }
    return h->works;
}

//! inputs through the block's own input buffer managers from now on (before the first run): returns 1 if the block supplies them
int loradrop_batch_use_input_slabs(void *p, const int on)
{
    auto h = reinterpret_cast<BatchHandle *>(p);
    if (h->ready) return -1;
    h->slabs = on != 0;
    return 0;
}
int loradrop_batch_input_slabs_active(void *p) { auto h = reinterpret_cast<BatchHandle *>(p); return h->ready && h->slabs ? 1 : 0; }

/*! The block as a receiver, timed: the channels' samples ARRIVE in chunks of `chunk` per channel (a source block upstream); after every
 * arrival the scheduler calls work() with what each input holds -- the unconsumed remainder and the new samples, in ordinary host
 * memory -- until no input has 2N left. Messages and signals are counted and dropped (a sink downstream). Nothing is recorded.
 * out[0] = seconds inside the loop, out[1] = work() calls of the block, out[2] = packets posted, out[3] = samples consumed (all
 * channels), out[4] = signal emissions. */
int loradrop_batch_bench(void *p, const float *iq, const size_t samplesPerChannel, const size_t chunk, double *out)
{
    auto h = reinterpret_cast<BatchHandle *>(p);
    const size_t B = h->B;
    prepare(h);
    Feeder f(h, iq, samplesPerChannel);
    std::vector<Pothos::OutputPort *> msg(B);
    for (size_t c = 0; c < B; c++) msg[c] = h->block->output(int(c));
    double works = 0, packets = 0, consumed = 0;
    const size_t sig0 = h->block->signals.size();
    const auto t0 = std::chrono::steady_clock::now();
    try
    {
        for (size_t w = chunk < samplesPerChannel ? chunk : samplesPerChannel; ; w = w + chunk < samplesPerChannel ? w + chunk : samplesPerChannel)
        {
            f.arrive(w);
            while (true)
            {
                if (!f.present()) break;
                if (h->ports)
                    for (size_t c = 0; c < B; c++)
                    {
                        const std::string s = std::to_string(c);
                        for (const char *port : { "raw", "dec", "fft" }) { auto o = h->block->output(port + s); o->produced = 0; o->labels.clear(); }
                    }
                h->block->work();
                works += 1;
                size_t progressed = 0;
                for (size_t c = 0; c < B; c++)
                {
                    packets += double(msg[c]->messages.size());
                    msg[c]->messages.clear();
                    progressed += f.consumed(c);
                }
                consumed += double(progressed);
                if (progressed == 0) break;
            }
            if (w >= samplesPerChannel) break;
        }
    }
    catch (const std::exception &) { return -2; }
    // (the time the SOURCE spent writing the arriving samples into the input buffers is the upstream block's, not this block's)
    out[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - f.sourceSeconds;
    out[1] = works; out[2] = packets; out[3] = consumed; out[4] = double(h->block->signals.size() - sig0);
    h->block->signals.clear();
    return 0;
}

size_t loradrop_batch_count(void *p, const size_t c, const char *what)
{
    const ChanLog &L = reinterpret_cast<BatchHandle *>(p)->log.at(c);
    const std::string w(what);
    if (w == "raw") return L.raw.size();
    if (w == "dec") return L.dec.size();
    if (w == "fft") return L.fft.size();
    if (w == "rawLabels") return L.rawLabels.size();
    if (w == "decLabels") return L.decLabels.size();
    if (w == "fftLabels") return L.fftLabels.size();
    if (w == "packets") return L.packets.size();
    if (w == "consumed") return L.consumed;
    return 0;
}

void loradrop_batch_get_stream(void *p, const size_t c, const char *what, float *out)
{
    const ChanLog &L = reinterpret_cast<BatchHandle *>(p)->log.at(c);
    const std::string w(what);
    const std::vector<cf32> &v = w == "raw" ? L.raw : (w == "dec" ? L.dec : L.fft);
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(cf32));
}

//! label i of a port: its absolute element index is returned, its id copied into buf
size_t loradrop_batch_get_label(void *p, const size_t c, const char *what, const size_t i, char *buf, const size_t cap)
{
    const ChanLog &L = reinterpret_cast<BatchHandle *>(p)->log.at(c);
    const std::string w(what);
    const auto &v = w == "raw" ? L.rawLabels : (w == "dec" ? L.decLabels : L.fftLabels);
    const auto &l = v.at(i);
    if (cap) { std::strncpy(buf, l.second.c_str(), cap - 1); buf[cap - 1] = 0; }
    return l.first;
}

size_t loradrop_batch_packet_len(void *p, const size_t c, const size_t i) { return reinterpret_cast<BatchHandle *>(p)->log.at(c).packets.at(i).size(); }
void loradrop_batch_get_packet(void *p, const size_t c, const size_t i, int16_t *out)
{
    const auto &s = reinterpret_cast<BatchHandle *>(p)->log.at(c).packets.at(i);
    if (!s.empty()) std::memcpy(out, s.data(), s.size() * sizeof(int16_t));
}

size_t loradrop_batch_num_signals(void *p) { return reinterpret_cast<BatchHandle *>(p)->block->signals.size(); }
double loradrop_batch_get_signal(void *p, const size_t i, char *buf, const size_t cap)
{
    const auto &r = reinterpret_cast<BatchHandle *>(p)->block->signals.at(i);
    if (cap) { std::strncpy(buf, r.name.c_str(), cap - 1); buf[cap - 1] = 0; }
    return r.value;
}


/***********************************************************************
 * lora_sdr_amd/pothos/LoRaDecoderBatch.cpp (/lora/lora_decoder_batch): messages in on B inputs, one work(), messages and the
 * "dropped" signal out. tests/test_gpu_dropin.py compares it, message by message, with the verbatim LoRaDecoder.cpp block.
 **********************************************************************/
void *loradrop_decoder_new(const size_t channels)
{
    auto it = Pothos::BlockRegistry::table().find("/lora/lora_decoder_batch");
    if (it == Pothos::BlockRegistry::table().end()) return nullptr;
    try { return it->second(channels); }
    catch (const std::exception &) { return nullptr; }
}

void loradrop_decoder_free(void *p) { delete reinterpret_cast<Pothos::Block *>(p); }

int loradrop_decoder_set(void *p, const char *name, const double v)
{
    auto b = reinterpret_cast<Pothos::Block *>(p);
    auto it = b->calls.find(name);
    if (it == b->calls.end()) return -1;
    try { it->second(v); } catch (const std::exception &) { return -2; }
    return 0;
}

int loradrop_decoder_set_string(void *p, const char *name, const char *v)
{
    auto b = reinterpret_cast<Pothos::Block *>(p);
    auto it = b->stringCalls.find(name);
    if (it == b->stringCalls.end()) return -1;
    try { it->second(v); } catch (const std::exception &) { return -2; }
    return 0;
}

int loradrop_decoder_activate(void *p)
{
    try { reinterpret_cast<Pothos::Block *>(p)->activate(); } catch (const std::exception &) { return -2; }
    return 0;
}

//! one symbol packet onto input `channel` (what /lora/lora_demod_batch posts there)
void loradrop_decoder_push(void *p, const size_t channel, const uint16_t *syms, const size_t n)
{
    Pothos::Packet pkt;
    pkt.payload = Pothos::BufferChunk(typeid(uint16_t), n ? n : 1);
    pkt.payload.length = n * sizeof(uint16_t);
    if (n) std::memcpy(pkt.payload.as<void *>(), syms, n * sizeof(uint16_t));
    reinterpret_cast<Pothos::Block *>(p)->input(int(channel))->_msgs.push_back(Pothos::Object(pkt));
}

int loradrop_decoder_work(void *p)
{
    try { reinterpret_cast<Pothos::Block *>(p)->work(); } catch (const std::exception &) { return -2; }
    return 0;
}

size_t loradrop_decoder_num_out(void *p, const size_t channel) { return reinterpret_cast<Pothos::Block *>(p)->output(int(channel))->messages.size(); }
size_t loradrop_decoder_out_len(void *p, const size_t channel, const size_t i) { return reinterpret_cast<Pothos::Block *>(p)->output(int(channel))->messages.at(i).size(); }
void loradrop_decoder_get_out(void *p, const size_t channel, const size_t i, void *out)
{
    const auto &m = reinterpret_cast<Pothos::Block *>(p)->output(int(channel))->messages.at(i);
    if (!m.empty()) std::memcpy(out, m.data(), m.size());
}
//! the values the block emitted on "dropped", in order
size_t loradrop_decoder_num_dropped_signals(void *p)
{
    size_t n = 0;
    for (const auto &sg : reinterpret_cast<Pothos::Block *>(p)->signals) n += sg.name == "dropped";
    return n;
}
double loradrop_decoder_dropped_signal(void *p, const size_t i)
{
    size_t n = 0;
    for (const auto &sg : reinterpret_cast<Pothos::Block *>(p)->signals) if (sg.name == "dropped" && n++ == i) return sg.value;
    return -1.0;
}

} // extern "C"
