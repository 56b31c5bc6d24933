// LoRaDemodBatch.cpp -- a Pothos block that runs B channels of the reference's /lora/lora_demod on one MI355X through
// level 3 of the C ABI (include/lorahip.h). The reference-side binding of INTEGRATION.md section 2, as a real translation unit:
// it compiles against <Pothos/Framework.hpp> and links liblorahip.so; nothing else.
//
// Same parameters and defaults as LoRaDemod (LoRaDemod.cpp:68-74, setters :124-137), same state machine, packets, signals,
// labels and debug ports per channel -- what changes is the shape: one block instance owns B inputs.
//
//   factory   /lora/lora_demod_batch(sf, channels)
//   setDevices("0,1,2,3,4,5,6,7")  spread the B channels over several GPUs of the node: contiguous ranges
//             from lorahip_shard_plan, one level-3 object per device, each run from its own host thread inside work(); the
//             default is device 0 alone. Channels are independent: per-channel outputs do not depend on the split.
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
#include <thread>
#include <vector>
#include "lorahip.h"

class LoRaDemodBatch : public Pothos::Block
{
    typedef std::complex<float> cf32;

public:
    LoRaDemodBatch(const size_t sf, const size_t channels) :
        N(size_t(1) << sf), B(channels), _sf(sf), _sync(0x12), _thresh(-30.0), _mtu(256), _maxWindows(64), _fftDropped(0)
    {
        createShards(std::vector<int>(1, 0));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setSync));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setDevices));
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

    ~LoRaDemodBatch(void) { destroyShards(); }

    static Block *make(const size_t sf, const size_t channels) { return new LoRaDemodBatch(sf, channels); }

    void setSync(const unsigned char sync) { _sync = sync; for (auto &s : _shards) lorahip_demod_set_sync(s.d, sync); }
    void setThreshold(const double thresh_dB) { _thresh = thresh_dB; for (auto &s : _shards) lorahip_demod_set_threshold(s.d, thresh_dB); }
    void setMTU(const size_t mtu) { _mtu = mtu; for (auto &s : _shards) lorahip_demod_set_mtu(s.d, mtu); }
    void setMaxWindows(const size_t k) { _maxWindows = k ? k : 1; sizeBuffers(); }
    size_t fftFramesDropped(void) const { return _fftDropped; }

    //! comma-separated device indices, e.g. "0,1,2,3,4,5,6,7"; an index may repeat (two shards on one GPU)
    void setDevices(const std::string &list)
    {
        std::vector<int> devs;
        size_t at = 0;
        while (at < list.size())
        {
            size_t end = list.find(',', at);
            if (end == std::string::npos) end = list.size();
            const std::string tok = list.substr(at, end - at);
            if (tok.empty() || tok.find_first_not_of("0123456789 ") != std::string::npos) throw Pothos::InvalidArgumentException("LoRaDemodBatch::setDevices(" + list + ")", "not a list of device indices");
            devs.push_back(std::atoi(tok.c_str()));
            at = end + 1;
        }
        if (devs.empty()) throw Pothos::InvalidArgumentException("LoRaDemodBatch::setDevices()", "empty list");
        createShards(devs);
        sizeBuffers();
    }

    void activate(void) { for (auto &s : _shards) lorahip_demod_activate(s.d); }

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

        // every device's channels in one launch on that device; several devices run side by side, each from its own host thread
        std::vector<int> rcs(_shards.size(), LORAHIP_OK);
        std::vector<std::string> errs(_shards.size());
        auto runShard = [&](const size_t i)
        {
            Shard &s = _shards[i];
            lorahip_demod_set_trace(s.d, 1);
            rcs[i] = lorahip_demod_run(s.d, streams.data() + s.first, avail.data() + s.first, nullptr);
            if (rcs[i] != LORAHIP_OK) errs[i] = lorahip_last_error();                    // the text is per thread
        };
        if (_shards.size() == 1) runShard(0);
        else
        {
            std::vector<std::thread> pool;
            for (size_t i = 0; i < _shards.size(); i++) pool.emplace_back(runShard, i);
            for (auto &t : pool) t.join();
        }
        for (size_t i = 0; i < _shards.size(); i++)
            if (rcs[i] != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch::work()", errs[i]);

        std::vector<lorahip_work_result> tr;
        std::vector<char> labels;
        for (size_t cg = 0; cg < B; cg++)
        {
            // channel cg of the block = channel c of the shard that owns it
            const Shard &sh = _shards[_shardOf[cg]];
            lorahip_demod *_d = sh.d;
            const size_t c = cg - sh.first;
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
            auto raw = this->output("raw" + std::to_string(cg)), dec = this->output("dec" + std::to_string(cg)), fft = this->output("fft" + std::to_string(cg));
            std::memcpy(raw->buffer().template as<cf32 *>(), _raw.data() + cg * capSamples, nr * sizeof(cf32));
            std::memcpy(dec->buffer().template as<cf32 *>(), _dec.data() + cg * capSamples, nd * sizeof(cf32));
            std::memcpy(fft->buffer().template as<cf32 *>(), _fft.data() + cg * _fftCap * N, frames * N * sizeof(cf32));

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
                    this->emitSignal("channel", int(cg));
                    this->emitSignal("error", tr[k].sig_error);
                    this->emitSignal("power", tr[k].sig_power);
                    this->emitSignal("snr", tr[k].sig_snr);
                }
                pos += size_t(tr[k].consumed);
            }
            this->input(int(cg))->consume(size_t(lorahip_demod_consumed(_d, c)));       // the sum of consume(total), :320
            raw->produce(nr);
            dec->produce(nd);
            fft->produce(frames * N);
        }
        // packets, in the order the channels posted them (:295-298); every channel has its own message port
        for (const Shard &sh : _shards)
        {
            lorahip_demod *_d = sh.d;
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
                this->output(int(sh.first + size_t(ch)))->postMessage(pkt);
            }
            lorahip_demod_clear_packets(_d);
            lorahip_demod_set_trace(_d, 0);                                              // the next work() starts a fresh trace
        }
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
        for (const Shard &sh : _shards)
        {
            // a shard's channels are a contiguous range of the block's: its port buffers are that range of the staging arrays
            lorahip_demod_ports p;
            std::memset(&p, 0, sizeof(p));
            p.struct_size = sizeof(p);
            p.fft_dev = reinterpret_cast<float *>(_fft.data() + sh.first * _fftCap * N); p.fft_cap_frames = _fftCap;
            p.dec_dev = reinterpret_cast<float *>(_dec.data() + sh.first * capSamples); p.dec_cap_samples = capSamples;
            p.raw_dev = reinterpret_cast<float *>(_raw.data() + sh.first * capSamples); p.raw_cap_samples = capSamples;
            p.host_buffers = 1;
            if (lorahip_demod_set_ports(sh.d, &p) != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch", lorahip_last_error());
        }
    }

    struct Shard { lorahip_demod *d; size_t first, count; int device; };

    void destroyShards(void)
    {
        for (auto &s : _shards) lorahip_demod_destroy(s.d);
        _shards.clear();
    }

    //! B channels of one SF over the given devices: lorahip_shard_plan (equal weights: contiguous ranges, sizes differing by <= 1)
    void createShards(const std::vector<int> &devices)
    {
        std::vector<int32_t> sfs(B, int32_t(_sf)), plan(B, 0);
        if (lorahip_shard_plan(sfs.data(), B, devices.size(), plan.data()) != LORAHIP_OK) throw Pothos::InvalidArgumentException("LoRaDemodBatch::setDevices()", "bad device list");
        destroyShards();
        _shardOf.assign(B, 0);
        for (size_t i = 0; i < devices.size(); i++)
        {
            size_t first = B, count = 0;
            for (size_t c = 0; c < B; c++) if (size_t(plan[c]) == i) { if (first == B) first = c; count++; }
            if (count == 0) continue;
            Shard s; s.d = nullptr; s.first = first; s.count = count; s.device = devices[i];
            const int rc = lorahip_demod_create(&s.d, devices[i], int(_sf), count);
            if (rc != LORAHIP_OK) { destroyShards(); throw Pothos::Exception("LoRaDemodBatch", std::string(lorahip_strerror(rc)) + " " + lorahip_last_error()); }
            lorahip_demod_set_sync(s.d, _sync); lorahip_demod_set_threshold(s.d, _thresh); lorahip_demod_set_mtu(s.d, _mtu);
            for (size_t c = first; c < first + count; c++) _shardOf[c] = _shards.size();
            _shards.push_back(s);
        }
    }

    const size_t N, B, _sf;
    unsigned char _sync; double _thresh; size_t _mtu;       // the setters' values, re-applied when the device list changes
    std::vector<Shard> _shards;
    std::vector<size_t> _shardOf;                           // per channel: index into _shards
    size_t _maxWindows, _fftCap, _fftDropped;
    std::vector<cf32> _raw, _dec, _fft;          // host staging of the three ports, [channel][capacity]
};

static Pothos::BlockRegistry registerLoRaDemodBatch("/lora/lora_demod_batch", &LoRaDemodBatch::make);
