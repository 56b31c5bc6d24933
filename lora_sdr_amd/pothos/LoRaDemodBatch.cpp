// LoRaDemodBatch.cpp -- a Pothos block that runs B channels of the reference's /lora/lora_demod on one MI355X through
// level 3 of the C ABI (include/lorahip.h). The reference-side binding of INTEGRATION.md section 2, as a real translation unit:
// it compiles against <Pothos/Framework.hpp> and links liblorahip.so; nothing else.
//
// Same parameters and defaults as LoRaDemod (LoRaDemod.cpp:68-74, setters :124-137), same state machine, packets, signals,
// labels and debug ports per channel -- what changes is the shape: one block instance owns B inputs.
//
//   factory   /lora/lora_demod_batch(sf, channels)
//   inputs    0 .. B-1            complex float streams, reserve 2N each                       (LoRaDemod.cpp:79,90)
//   outputs   "0" .. "B-1"        Pothos::Packet messages of int16 symbols                     (:80,295-298)
//             "raw<c>" "dec<c>"   complex float streams, `total` elements per work() call      (:81-82,163-164,321-322)
//             "fft<c>"            complex float stream, N bins per work() call                 (:83,172,324)
//   signals   "channel" followed by "error", "power", "snr" once per packet at DOWNCHIRP1     (:85-87,267-269)
//   labels    "SYNC", "P x", "DC", "QC", "S<n> x" on raw<c> / dec<c> / fft<c> at the first element each call produced (:314-319)
//
// One work() of this block performs, per channel, as many LoRaDemod::work() calls as the channel's input buffer allows
// (each needs 2N samples, :148) inside ONE device launch. Output buffers must therefore hold what several calls produce:
// setMaxWindows(K) sizes them (raw/dec: the samples consumed; fft: 2K frames per work(), more are dropped and counted).
#include <Pothos/Framework.hpp>
#include <complex>
#include <cstring>
#include <string>
#include <vector>
#include "lorahip.h"

class LoRaDemodBatch : public Pothos::Block
{
    typedef std::complex<float> cf32;

public:
    LoRaDemodBatch(const size_t sf, const size_t channels) :
        N(size_t(1) << sf), B(channels), _d(nullptr), _maxWindows(64), _fftDropped(0)
    {
        const int rc = lorahip_demod_create(&_d, 0, int(sf), channels);
        if (rc != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch", std::string(lorahip_strerror(rc)) + " " + lorahip_last_error());
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setSync));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setThreshold));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setMTU));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setMaxWindows));
        for (size_t c = 0; c < B; c++)
        {
            this->setupInput(int(c), typeid(cf32));
            this->setupOutput(int(c));
            this->setupOutput("raw" + std::to_string(c), typeid(cf32));
            this->setupOutput("dec" + std::to_string(c), typeid(cf32));
            this->setupOutput("fft" + std::to_string(c), typeid(cf32));
            this->input(int(c))->setReserve(N * 2);                         // use at most two input symbols available (:90)
        }
        this->registerSignal("channel");
        this->registerSignal("error");
        this->registerSignal("power");
        this->registerSignal("snr");
        sizeBuffers();
    }

    ~LoRaDemodBatch(void) { lorahip_demod_destroy(_d); }

    static Block *make(const size_t sf, const size_t channels) { return new LoRaDemodBatch(sf, channels); }

    void setSync(const unsigned char sync) { lorahip_demod_set_sync(_d, sync); }
    void setThreshold(const double thresh_dB) { lorahip_demod_set_threshold(_d, thresh_dB); }
    void setMTU(const size_t mtu) { lorahip_demod_set_mtu(_d, mtu); }
    void setMaxWindows(const size_t k) { _maxWindows = k ? k : 1; sizeBuffers(); }
    size_t fftFramesDropped(void) const { return _fftDropped; }

    void activate(void) { lorahip_demod_activate(_d); }

    void work(void)
    {
        std::vector<const float *> streams(B);
        std::vector<size_t> avail(B);
        bool any = false;
        const size_t capSamples = _maxWindows * N;
        for (size_t c = 0; c < B; c++)
        {
            auto in = this->input(int(c));
            streams[c] = reinterpret_cast<const float *>(in->buffer().template as<const cf32 *>());
            avail[c] = in->elements() < capSamples ? in->elements() : capSamples;       // never produce more than the output buffers hold
            any = any || avail[c] >= 2 * N;                                             // :148
        }
        if (!any) return;

        lorahip_demod_set_trace(_d, 1);
        if (lorahip_demod_run(_d, streams.data(), avail.data(), nullptr) != LORAHIP_OK)
            throw Pothos::Exception("LoRaDemodBatch::work()", lorahip_last_error());

        std::vector<lorahip_work_result> tr;
        std::vector<char> labels;
        for (size_t c = 0; c < B; c++)
        {
            const size_t nCalls = lorahip_demod_trace_len(_d, c);
            if (nCalls == 0) continue;
            tr.resize(nCalls);
            lorahip_demod_get_trace(_d, c, tr.data(), nCalls);
            size_t nLab = 0, nBytes = 0;
            lorahip_demod_get_labels(_d, c, nullptr, 0, &nLab, &nBytes);
            labels.resize(nBytes);
            lorahip_demod_get_labels(_d, c, labels.data(), nBytes, &nLab, &nBytes);

            size_t nf = 0, nd = 0, nr = 0;
            lorahip_demod_port_counts(_d, c, &nf, &nd, &nr);
            const size_t frames = nf < _fftCap ? nf : _fftCap;
            _fftDropped += nf - frames;
            auto raw = this->output("raw" + std::to_string(c)), dec = this->output("dec" + std::to_string(c)), fft = this->output("fft" + std::to_string(c));
            std::memcpy(raw->buffer().template as<cf32 *>(), _raw.data() + c * capSamples, nr * sizeof(cf32));
            std::memcpy(dec->buffer().template as<cf32 *>(), _dec.data() + c * capSamples, nd * sizeof(cf32));
            std::memcpy(fft->buffer().template as<cf32 *>(), _fft.data() + c * _fftCap * N, frames * N * sizeof(cf32));

            // labels at the first element each call produced (:314-319); signals (:267-269)
            size_t pos = 0;
            const char *lab = labels.data();
            for (size_t k = 0; k < nCalls; k++)
            {
                const std::string id(lab);
                lab += id.size() + 1;
                if (!id.empty())
                {
                    raw->postLabel(Pothos::Label(id, Pothos::Object(), pos));
                    dec->postLabel(Pothos::Label(id, Pothos::Object(), pos));
                    if (k < frames) fft->postLabel(Pothos::Label(id, Pothos::Object(), k * N));
                }
                if (tr[k].signals)
                {
                    this->emitSignal("channel", int(c));
                    this->emitSignal("error", tr[k].sig_error);
                    this->emitSignal("power", tr[k].sig_power);
                    this->emitSignal("snr", tr[k].sig_snr);
                }
                pos += size_t(tr[k].consumed);
            }
            this->input(int(c))->consume(size_t(lorahip_demod_consumed(_d, c)));        // the sum of consume(total), :320
            raw->produce(nr);
            dec->produce(nd);
            fft->produce(frames * N);
        }
        // packets, in the order the channels posted them (:295-298)
        const size_t nPackets = lorahip_demod_num_packets(_d);
        for (size_t i = 0; i < nPackets; i++)
        {
            int32_t ch = 0;
            size_t len = 0;
            lorahip_demod_get_packet(_d, i, &ch, nullptr, &len, nullptr, 0);
            Pothos::Packet pkt;
            pkt.payload = Pothos::BufferChunk(typeid(int16_t), len ? len : 1);
            pkt.payload.length = len * sizeof(int16_t);
            lorahip_demod_get_packet(_d, i, nullptr, nullptr, nullptr, pkt.payload.template as<int16_t *>(), len);
            this->output(int(ch))->postMessage(pkt);
        }
        lorahip_demod_clear_packets(_d);
        lorahip_demod_set_trace(_d, 0);                                                  // the next work() starts a fresh trace
    }

    //! output buffers large enough for what one work() produces (the reference does the same for its 2N / N, :330-358)
    Pothos::BufferManager::Sptr getOutputBufferManager(const std::string &name, const std::string &domain)
    {
        if (name.compare(0, 3, "raw") == 0 || name.compare(0, 3, "dec") == 0 || name.compare(0, 3, "fft") == 0)
        {
            Pothos::BufferManagerArgs args;
            args.bufferSize = (name.compare(0, 3, "fft") == 0 ? _fftCap : _maxWindows) * N * sizeof(cf32);
            return Pothos::BufferManager::make("generic", args);
        }
        return Pothos::Block::getOutputBufferManager(name, domain);
    }

private:
    void sizeBuffers(void)
    {
        _fftCap = 2 * _maxWindows;
        const size_t capSamples = _maxWindows * N;
        _raw.assign(B * capSamples, cf32());
        _dec.assign(B * capSamples, cf32());
        _fft.assign(B * _fftCap * N, cf32());
        lorahip_demod_ports p;
        std::memset(&p, 0, sizeof(p));
        p.struct_size = sizeof(p);
        p.fft_dev = reinterpret_cast<float *>(_fft.data()); p.fft_cap_frames = _fftCap;
        p.dec_dev = reinterpret_cast<float *>(_dec.data()); p.dec_cap_samples = capSamples;
        p.raw_dev = reinterpret_cast<float *>(_raw.data()); p.raw_cap_samples = capSamples;
        p.host_buffers = 1;
        if (lorahip_demod_set_ports(_d, &p) != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch", lorahip_last_error());
    }

    const size_t N, B;
    lorahip_demod *_d;
    size_t _maxWindows, _fftCap, _fftDropped;
    std::vector<cf32> _raw, _dec, _fft;          // host staging of the three ports, [channel][capacity]
};

static Pothos::BlockRegistry registerLoRaDemodBatch("/lora/lora_demod_batch", &LoRaDemodBatch::make);
