// TEST INFRASTRUCTURE ONLY (oracle/) -- never linked or called by the product path.
//
// Thin extern "C" driver around the REAL reference code, compiled in place from
// /root/reference (see oracle/Makefile; output goes to oracle/_ref/ only):
//   * LoRaDetector.hpp + kissfft.hh      (framework-free, included below verbatim)
//   * ChirpGenerator.hpp                 (needs only the empty stub Pothos/Config.hpp)
//   * LoRaDemod.cpp, LoRaMod.cpp         (separate TUs, verbatim, against the recording
//                                          fake in oracle/stub/Pothos/Framework.hpp)
// It exists to (1) pin the plain-C restatement in oracle/lora_oracle.c, (2) generate the
// golden vectors in tests/golden/, (3) serve as the "reference" CPU baseline in bench.py.
//
// The reference leaves LoRaDemod::_finefreqError/_freqError/_prevValue uninitialised
// (LoRaDemod.cpp:68-74 vs :160,:183). This library replaces operator new
// with a zero-filling one (linked -Bsymbolic so only this .so is affected) so that the
// block starts from the all-zero state the restatement and the HIP path define.
#include <Pothos/Framework.hpp>
#include <chrono>
#include <cstdlib>
#include <new>
#include <thread>
#include "LoRaDetector.hpp"
#include "ChirpGenerator.hpp"

void *operator new(std::size_t n)
{
    void *p = std::calloc(n ? n : 1, 1);
    if (p == nullptr) throw std::bad_alloc();
    return p;
}
void *operator new[](std::size_t n) { return operator new(n); }
void operator delete(void *p) noexcept { std::free(p); }
void operator delete[](void *p) noexcept { std::free(p); }
void operator delete(void *p, std::size_t) noexcept { std::free(p); }
void operator delete[](void *p, std::size_t) noexcept { std::free(p); }

typedef std::complex<float> cf32;

namespace {

struct DemodHandle
{
    Pothos::Block *block;
    size_t N;
    std::vector<cf32> raw, dec, fft;
    //logs across work() calls
    std::vector<int64_t> consumed;      // per call
    std::vector<std::string> labels;    // per call: label posted on "raw" this call ("" if none)
    std::vector<float> fftLog;          // N floats*2 per call
    std::vector<float> decLog;          // 2N floats*2 per call (full port buffer snapshot)
    std::vector<std::vector<int16_t>> packets;
    std::vector<int64_t> packetCall;    // call index that posted each packet
    std::vector<Pothos::SignalRecord> signals;
    int keepBuffers;                    // 0: count only, 1: consumed + labels + fft/dec snapshots per call, 2: consumed + labels only
    int64_t callsMade;                  // work() calls since the handle was created
};

} // namespace

extern "C" {

/***********************************************************************
 * LoRaDetector<float> (LoRaDetector.hpp:8-72)
 **********************************************************************/
void *loraref_detector_new(const size_t N) { return new LoRaDetector<float>(N); }
void loraref_detector_free(void *d) { delete reinterpret_cast<LoRaDetector<float> *>(d); }

size_t loraref_detector_detect(void *d, const size_t N, const float *iq,
                               float *power, float *powerAvg, float *fIndex, float *fftOut)
{
    auto det = reinterpret_cast<LoRaDetector<float> *>(d);
    auto in = reinterpret_cast<const cf32 *>(iq);
    for (size_t i = 0; i < N; i++) det->feed(i, in[i]);
    return det->detect(*power, *powerAvg, *fIndex, reinterpret_cast<cf32 *>(fftOut));
}

//! n independent windows of already-dechirped samples, split over nthreads
void loraref_detect_windows(const size_t N, const float *iq, const size_t nWindows,
                            uint16_t *sym, float *power, float *powerAvg, float *fIndex,
                            float *fftOut, const int nthreads)
{
    auto body = [=](const size_t lo, const size_t hi)
    {
        LoRaDetector<float> det(N);
        auto in = reinterpret_cast<const cf32 *>(iq);
        for (size_t w = lo; w < hi; w++)
        {
            for (size_t i = 0; i < N; i++) det.feed(i, in[w * N + i]);
            cf32 *out = fftOut ? reinterpret_cast<cf32 *>(fftOut) + w * N : nullptr;
            sym[w] = uint16_t(det.detect(power[w], powerAvg[w], fIndex[w], out));
        }
    };
    const size_t T = nthreads > 1 ? size_t(nthreads) : 1;
    std::vector<std::thread> pool;
    for (size_t t = 0; t < T; t++)
        pool.emplace_back(body, nWindows * t / T, nWindows * (t + 1) / T);
    for (auto &th : pool) th.join();
}

/***********************************************************************
 * genChirp<float> (ChirpGenerator.hpp:22-47)
 **********************************************************************/
int loraref_genchirp(float *samps, const int N, const int ovs, const int NN, const float f0,
                     const int down, const float ampl, float *phaseAccum)
{
    return genChirp(reinterpret_cast<cf32 *>(samps), N, ovs, NN, f0, down != 0, ampl, *phaseAccum);
}

/***********************************************************************
 * LoRaDemod block (LoRaDemod.cpp, through the registry like a topology would)
 **********************************************************************/
void *loraref_demod_new(const size_t sf, const int keepBuffers)
{
    auto it = Pothos::BlockRegistry::table().find("/lora/lora_demod");
    if (it == Pothos::BlockRegistry::table().end()) return nullptr;
    auto h = new DemodHandle();
    h->block = it->second(sf);
    h->N = size_t(1) << sf;
    h->keepBuffers = keepBuffers;
    h->callsMade = 0;
    //the buffer-manager hooks set the reserves (LoRaDemod.cpp:330-358)
    h->block->getInputBufferManager("0", "");
    h->block->getOutputBufferManager("raw", "");
    h->block->getOutputBufferManager("dec", "");
    h->block->getOutputBufferManager("fft", "");
    h->raw.resize(2 * h->N); h->dec.resize(2 * h->N); h->fft.resize(h->N);
    h->block->output("raw")->_buff = Pothos::BufferChunk::view(h->raw.data(), 2 * h->N * sizeof(cf32));
    h->block->output("dec")->_buff = Pothos::BufferChunk::view(h->dec.data(), 2 * h->N * sizeof(cf32));
    h->block->output("fft")->_buff = Pothos::BufferChunk::view(h->fft.data(), h->N * sizeof(cf32));
    h->block->activate();
    return h;
}

void loraref_demod_free(void *p)
{
    auto h = reinterpret_cast<DemodHandle *>(p);
    delete h->block;
    delete h;
}

int loraref_demod_set(void *p, const char *name, const double v)
{
    auto h = reinterpret_cast<DemodHandle *>(p);
    auto it = h->block->calls.find(name);
    if (it == h->block->calls.end()) return -1;
    it->second(v);
    return 0;
}

//! Feed a whole stream: call work() until fewer than 2N elements remain (LoRaDemod.cpp:148).
//! Returns the number of work() calls made; logs are appended to the handle.
int64_t loraref_demod_run(void *p, const float *iq, const size_t nSamples)
{
    auto h = reinterpret_cast<DemodHandle *>(p);
    auto in = h->block->input(0);
    auto raw = h->block->output("raw"), dec = h->block->output("dec"), fft = h->block->output("fft");
    auto out0 = h->block->output(0);
    size_t pos = 0;
    int64_t calls = 0;
    while (true)
    {
        in->_elems = nSamples - pos;
        in->_buff = Pothos::BufferChunk::view(const_cast<float *>(iq) + 2 * pos, (nSamples - pos) * sizeof(cf32));
        in->consumed = 0;
        if (in->_elems < 2 * h->N) break;
        const size_t nLabels = raw->labels.size();
        const size_t nMsgs = out0->messages.size();
        h->block->work();
        calls++;
        h->callsMade++;
        if (h->keepBuffers)
        {
            h->consumed.push_back(int64_t(in->consumed));
            h->labels.push_back(raw->labels.size() > nLabels ? raw->labels.back().id : std::string());
        }
        if (h->keepBuffers == 1)
        {
            auto f = reinterpret_cast<const float *>(h->fft.data());
            h->fftLog.insert(h->fftLog.end(), f, f + 2 * h->N);
            auto d = reinterpret_cast<const float *>(h->dec.data());
            h->decLog.insert(h->decLog.end(), d, d + 4 * h->N);
        }
        for (size_t m = nMsgs; m < out0->messages.size(); m++)
        {
            const auto &bytes = out0->messages[m];
            std::vector<int16_t> syms(bytes.size() / sizeof(int16_t));
            if (!syms.empty()) std::memcpy(syms.data(), bytes.data(), syms.size() * sizeof(int16_t));
            h->packets.push_back(syms);
            h->packetCall.push_back(h->callsMade - 1);
        }
        if (!h->keepBuffers) { out0->messages.clear(); raw->labels.clear(); dec->labels.clear(); fft->labels.clear(); }
        pos += in->consumed;
        if (in->consumed == 0) break; //cannot happen in the reference; guards the loop
    }
    return calls;
}

size_t loraref_demod_num_calls(void *p) { return reinterpret_cast<DemodHandle *>(p)->consumed.size(); }
size_t loraref_demod_num_packets(void *p) { return reinterpret_cast<DemodHandle *>(p)->packets.size(); }
size_t loraref_demod_num_signals(void *p) { return reinterpret_cast<DemodHandle *>(p)->block->signals.size(); }

void loraref_demod_get_calls(void *p, int64_t *consumed, float *fftLog, float *decLog)
{
    auto h = reinterpret_cast<DemodHandle *>(p);
    if (consumed) std::memcpy(consumed, h->consumed.data(), h->consumed.size() * sizeof(int64_t));
    if (fftLog) std::memcpy(fftLog, h->fftLog.data(), h->fftLog.size() * sizeof(float));
    if (decLog) std::memcpy(decLog, h->decLog.data(), h->decLog.size() * sizeof(float));
}

//! copies label i (NUL terminated, truncated to cap) and returns its length
size_t loraref_demod_get_label(void *p, const size_t i, char *buf, const size_t cap)
{
    auto h = reinterpret_cast<DemodHandle *>(p);
    const std::string &s = h->labels.at(i);
    if (cap) { std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
    return s.size();
}

size_t loraref_demod_packet_len(void *p, const size_t i) { return reinterpret_cast<DemodHandle *>(p)->packets.at(i).size(); }
int64_t loraref_demod_packet_call(void *p, const size_t i) { return reinterpret_cast<DemodHandle *>(p)->packetCall.at(i); }
void loraref_demod_get_packet(void *p, const size_t i, int16_t *out)
{
    const auto &s = reinterpret_cast<DemodHandle *>(p)->packets.at(i);
    if (!s.empty()) std::memcpy(out, s.data(), s.size() * sizeof(int16_t));
}

//! signal i: name into buf, value returned
double loraref_demod_get_signal(void *p, const size_t i, char *buf, const size_t cap)
{
    const auto &r = reinterpret_cast<DemodHandle *>(p)->block->signals.at(i);
    if (cap) { std::strncpy(buf, r.name.c_str(), cap - 1); buf[cap - 1] = 0; }
    return r.value;
}

//! CPU baseline: nthreads blocks, each runs its own contiguous stream; returns total work() calls
int64_t loraref_demod_bench(const size_t sf, const float *iq, const size_t samplesPerStream,
                            const int nStreams, const int nthreads, const int repeat)
{
    std::vector<int64_t> calls(size_t(nStreams), 0);
    const std::vector<cf32> ones((size_t(2) << sf), cf32(1.0f, 0.0f));   // exactly one work() call
    auto body = [&](const int lo, const int hi)
    {
        // one block per worker (the constructor builds a 128*N-entry table: LoRaDemod.cpp:108-114),
        // re-activated for every stream
        void *h = loraref_demod_new(sf, 0);
        for (int rep = 0; rep < (repeat > 1 ? repeat : 1); rep++)
        for (int s = lo; s < hi; s++)
        {
            // Bring the reused block back to its start state. _finefreqError/_fineTuneIndex are only
            // reset by the "just noise" branch (LoRaDemod.cpp:229-234); left to random-walk across
            // streams they eventually index past _fineTuneTable (reference UB). One squelched work()
            // on a constant window (threshold raised for that call) performs exactly that reset.
            auto dh = reinterpret_cast<DemodHandle *>(h);
            dh->block->activate();
            dh->block->calls["setThreshold"](1e30);
            loraref_demod_run(h, reinterpret_cast<const float *>(ones.data()), ones.size());
            dh->block->calls["setThreshold"](-30.0);
            dh->block->activate();
            dh->block->signals.clear();
            dh->packets.clear(); dh->packetCall.clear();
            calls[size_t(s)] += loraref_demod_run(h, iq + 2 * size_t(s) * samplesPerStream, samplesPerStream);
        }
        loraref_demod_free(h);
    };
    const int T = nthreads > 1 ? nthreads : 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++) pool.emplace_back(body, nStreams * t / T, nStreams * (t + 1) / T);
    for (auto &th : pool) th.join();
    int64_t total = 0;
    for (auto c : calls) total += c;
    return total;
}

/*! Parity aid at scale: nStreams independent streams, each through a FRESH block (the zero start state the HIP path defines),
 * split over nthreads. Per stream s: nCalls[s] work() calls, nPackets[s] packets whose lengths go to pktLens[s*pktCap + j], the
 * posting call to pktCall[s*pktCap + j] (nullable) and whose symbols go back to back to pktSyms[s*symCap + ...]; optionally, per
 * call, what it consumed (callConsumed[s*callCap + k]) and the kind of label it posted (callClass: 0 none, 1 "SYNC", 2 "P ..",
 * 3 "DC", 4 "QC", 5 "S<n> .."; LoRaDemod.cpp:213,220-224,245,282,302-305). Returns the total number of calls, or -1 if a
 * capacity was too small for some stream (that stream's nPackets is set to -1). */
int64_t loraref_demod_run_many(const size_t sf, const float *iq, const size_t samplesPerStream, const int nStreams, const int nthreads,
                               const int sync, const double thresh, const size_t mtu,
                               int32_t *nCalls, int32_t *nPackets, int16_t *pktSyms, const size_t symCap,
                               int32_t *pktLens, int32_t *pktCall, const size_t pktCap,
                               int32_t *callConsumed, uint8_t *callClass, const size_t callCap)
{
    std::vector<int> bad(size_t(nthreads > 1 ? nthreads : 1), 0);
    auto body = [&](const int tix, const int lo, const int hi)
    {
        for (int s = lo; s < hi; s++)
        {
            void *h = loraref_demod_new(sf, callConsumed || callClass ? 2 : 0);
            auto dh = reinterpret_cast<DemodHandle *>(h);
            dh->block->calls["setSync"](double(sync));
            dh->block->calls["setThreshold"](thresh);
            dh->block->calls["setMTU"](double(mtu));
            const int64_t calls = loraref_demod_run(h, iq + 2 * size_t(s) * samplesPerStream, samplesPerStream);
            nCalls[s] = int32_t(calls);
            bool ok = dh->packets.size() <= pktCap;
            size_t at = 0;
            for (size_t j = 0; ok && j < dh->packets.size(); j++)
            {
                const auto &p = dh->packets[j];
                if (at + p.size() > symCap) { ok = false; break; }
                if (!p.empty()) std::memcpy(pktSyms + size_t(s) * symCap + at, p.data(), p.size() * sizeof(int16_t));
                at += p.size();
                pktLens[size_t(s) * pktCap + j] = int32_t(p.size());
                if (pktCall) pktCall[size_t(s) * pktCap + j] = int32_t(dh->packetCall[j]);
            }
            if (callConsumed || callClass)
            {
                if (dh->consumed.size() > callCap) ok = false;
                for (size_t k = 0; ok && k < dh->consumed.size(); k++)
                {
                    if (callConsumed) callConsumed[size_t(s) * callCap + k] = int32_t(dh->consumed[k]);
                    if (callClass)
                    {
                        const std::string &id = dh->labels[k];
                        uint8_t c = 0;
                        if (id == "SYNC") c = 1;
                        else if (id.size() && id[0] == 'P') c = 2;
                        else if (id == "DC") c = 3;
                        else if (id == "QC") c = 4;
                        else if (id.size() && id[0] == 'S') c = 5;
                        callClass[size_t(s) * callCap + k] = c;
                    }
                }
            }
            nPackets[s] = ok ? int32_t(dh->packets.size()) : -1;
            if (!ok) bad[size_t(tix)] = 1;
            loraref_demod_free(h);
        }
    };
    const int T = nthreads > 1 ? nthreads : 1;
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++) pool.emplace_back(body, t, nStreams * t / T, nStreams * (t + 1) / T);
    for (auto &th : pool) th.join();
    int64_t total = 0;
    for (int s = 0; s < nStreams; s++) total += nCalls[s];
    for (int b : bad) if (b) return -1;
    return total;
}

} // extern "C"

/***********************************************************************
 * LoRaMod block (LoRaMod.cpp:109-238): one packet of symbols -> the samples of its frame
 **********************************************************************/
extern "C" size_t loraref_mod_frame(const size_t sf, const int sync, const float ampl, const size_t padding,
                                    const uint16_t *syms, const size_t nsyms, float *out, const size_t capSamples)
{
    auto it = Pothos::BlockRegistry::table().find("/lora/lora_mod");
    if (it == Pothos::BlockRegistry::table().end()) return 0;
    Pothos::Block *block = it->second(sf);
    const size_t N = size_t(1) << sf;
    block->calls.at("setSync")(double(sync));
    block->calls.at("setPadding")(double(padding));
    block->calls.at("setAmplitude")(double(ampl));
    block->getOutputBufferManager("0", "");
    std::vector<cf32> buf(N);
    block->output(0)->_buff = Pothos::BufferChunk::view(buf.data(), N * sizeof(cf32));
    block->activate();
    Pothos::Packet pkt;
    pkt.payload = Pothos::BufferChunk(typeid(uint16_t), nsyms);
    for (size_t i = 0; i < nsyms; i++) pkt.payload.as<uint16_t *>()[i] = syms[i];
    block->input(0)->_msgs.push_back(Pothos::Object(pkt));
    size_t n = 0;
    for (size_t call = 0; call < nsyms + padding + 64; call++)
    {
        const size_t before = block->output(0)->produced;
        block->work();
        const size_t got = block->output(0)->produced - before;
        if (got == 0 && call > 0) break;           // back in STATE_WAITINPUT with nothing queued (LoRaMod.cpp:124-130)
        if (n + got > capSamples) { n = 0; break; }
        std::memcpy(reinterpret_cast<cf32 *>(out) + n, buf.data(), got * sizeof(cf32));
        n += got;
    }
    delete block;
    return n;
}

/***********************************************************************
 * The codec blocks, verbatim (LoRaEncoder.cpp:161-233, LoRaDecoder.cpp:196-397 on LoRaCodes.hpp)
 **********************************************************************/
static void setCodec(Pothos::Block *b, const char *name, const double v)
{
    auto it = b->calls.find(name);
    if (it != b->calls.end()) it->second(v);
}

//! one packet of demodulated symbols -> bytes. Returns the payload length of the message the block posted
//! (in elements: bytes, or uint16 symbols when interleaving is off), -1 if it posted nothing (too short / dropped).
extern "C" long loraref_decode(const size_t sf, const size_t ppm, const char *cr, const int crcc, const int interleaving,
                               const int errorCheck, const int explicitHdr, const int hdr, const size_t dataLength,
                               const uint16_t *syms, const size_t nsyms, void *out, const size_t capBytes, unsigned long long *dropped)
{
    auto it = Pothos::BlockRegistry::tableVoid().find("/lora/lora_decoder");
    if (it == Pothos::BlockRegistry::tableVoid().end()) return -2;
    Pothos::Block *b = it->second();
    setCodec(b, "setSpreadFactor", double(sf));
    setCodec(b, "setSymbolSize", double(ppm));
    b->stringCalls.at("setCodingRate")(cr);
    setCodec(b, "enableCrcc", crcc);
    setCodec(b, "enableInterleaving", interleaving);
    setCodec(b, "enableErrorCheck", errorCheck);
    setCodec(b, "enableExplicit", explicitHdr);
    setCodec(b, "enableHdr", hdr);
    setCodec(b, "setDataLength", double(dataLength));
    b->activate();
    Pothos::Packet pkt;
    pkt.payload = Pothos::BufferChunk(typeid(uint16_t), nsyms);
    if (nsyms) std::memcpy(pkt.payload.as<void *>(), syms, nsyms * sizeof(uint16_t));
    b->input(0)->_msgs.push_back(Pothos::Object(pkt));
    // the block can throw (Pothos::Exception for PPM > SF, std::bad_alloc / length_error when a header without the crc
    // flag announces fewer than 2 bytes and `dataLength -= 5` wraps around, LoRaDecoder.cpp:377): nothing is posted then
    try { b->work(); } catch (...) { delete b; if (dropped) *dropped = 0; return -1; }
    long n = -1;
    auto &msgs = b->output(0)->messages;
    if (!msgs.empty())
    {
        n = long(msgs[0].size());
        if (size_t(n) <= capBytes && n) std::memcpy(out, msgs[0].data(), size_t(n));
        if (!interleaving) n /= 2;
    }
    if (dropped)
    {
        *dropped = 0;
        for (const auto &sg : b->signals) if (sg.name == "dropped") *dropped = (unsigned long long)(sg.value);
    }
    delete b;
    return n;
}

//! bytes -> the symbols the encoder block posts; returns their number
extern "C" long loraref_encode(const size_t sf, const size_t ppm, const char *cr, const int explicitHdr, const int crc,
                               const int whitening, const uint8_t *bytes, const size_t nbytes, uint16_t *out, const size_t capSyms)
{
    auto it = Pothos::BlockRegistry::tableVoid().find("/lora/lora_encoder");
    if (it == Pothos::BlockRegistry::tableVoid().end()) return -2;
    Pothos::Block *b = it->second();
    setCodec(b, "setSpreadFactor", double(sf));
    setCodec(b, "setSymbolSize", double(ppm));
    b->stringCalls.at("setCodingRate")(cr);
    setCodec(b, "enableExplicit", explicitHdr);
    setCodec(b, "enableCrc", crc);
    setCodec(b, "enableWhitening", whitening);
    b->activate();
    Pothos::Packet pkt;
    pkt.payload = Pothos::BufferChunk(typeid(uint8_t), nbytes);
    if (nbytes) std::memcpy(pkt.payload.as<void *>(), bytes, nbytes);
    b->input(0)->_msgs.push_back(Pothos::Object(pkt));
    b->work();
    long n = -1;
    auto &msgs = b->output(0)->messages;
    if (!msgs.empty())
    {
        n = long(msgs[0].size() / 2);
        if (size_t(n) <= capSyms && n) std::memcpy(out, msgs[0].data(), size_t(n) * 2);
    }
    delete b;
    return n;
}

//! the code primitives themselves, for exhaustive checks (LoRaCodes.hpp is included by the codec blocks' TUs; here again)
#include "LoRaCodes.hpp"
extern "C" int loraref_code_primitive(const int which, const int b)
{
    bool error = false, bad = false;
    int v = 0;
    switch (which)
    {
    case 0: v = decodeHamming84sx((unsigned char)b, error, bad); break;
    case 1: v = decodeHamming74sx((unsigned char)b, error); break;
    case 2: v = checkParity54((unsigned char)b, error); break;
    case 3: v = checkParity64((unsigned char)b, error); break;
    case 4: { uint8_t h[3] = { uint8_t(b & 0xff), uint8_t((b >> 8) & 0xf), 0 }; return headerChecksum(h); }
    case 5: return binaryToGray16((unsigned short)b);
    default: return -1;
    }
    return v | (error ? 0x100 : 0) | (bad ? 0x200 : 0);
}

//! timing aid: the decoder block created once, `reps` messages decoded back to back; returns seconds
extern "C" double loraref_decode_bench(const size_t sf, const size_t ppm, const char *cr, const int crcc, const int errorCheck,
                                       const uint16_t *syms, const size_t nsyms, const size_t reps)
{
    auto it = Pothos::BlockRegistry::tableVoid().find("/lora/lora_decoder");
    if (it == Pothos::BlockRegistry::tableVoid().end()) return -1.0;
    Pothos::Block *b = it->second();
    setCodec(b, "setSpreadFactor", double(sf));
    setCodec(b, "setSymbolSize", double(ppm));
    b->stringCalls.at("setCodingRate")(cr);
    setCodec(b, "enableCrcc", crcc);
    setCodec(b, "enableErrorCheck", errorCheck);
    b->activate();
    Pothos::Packet pkt;
    pkt.payload = Pothos::BufferChunk(typeid(uint16_t), nsyms);
    std::memcpy(pkt.payload.as<void *>(), syms, nsyms * sizeof(uint16_t));
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t r = 0; r < reps; r++)
    {
        b->input(0)->_msgs.push_back(Pothos::Object(pkt));
        b->work();
        b->output(0)->messages.clear();
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    delete b;
    return dt;
}
