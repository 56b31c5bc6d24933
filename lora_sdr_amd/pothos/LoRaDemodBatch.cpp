// LoRaDemodBatch.cpp -- a Pothos block that runs B channels of the reference's /lora/lora_demod on the MI355X GPUs of a node through
// level 3 of the C ABI (include/lorahip.h). The reference-side binding of INTEGRATION.md section 2, as a real translation unit:
// it compiles against <Pothos/Framework.hpp> and links liblorahip.so; nothing else.
//
// Same parameters and defaults as LoRaDemod (LoRaDemod.cpp:68-74, setters :124-137), same state machine, packets and signals per
// channel -- what changes is the shape: one block instance owns B inputs, each channel with its own SF if asked.
//
//   factory   /lora/lora_demod_batch(sf, channels)
//   setSpreadFactors("7,8,9,10,11,12")  one SF per channel (a shorter list repeats: this one is SF = 7 + c mod 6); the reference makes
//             one block per channel with its own sf (LoRaDemod.cpp:119-122). Before activate().
//   setDevices("0,1,2,3,4,5,6,7")  spread the channels over several GPUs of the node: lorahip_demod_create_mixed
//             splits them with lorahip_shard_plan, one part per (device, SF) with its own stream and host thread; the default is
//             device 0 alone. Channels are independent: per-channel outputs do not depend on the split. Before activate().
//   setDebugPorts(bool)  the reference's three debug outputs raw / dec / fft (LoRaDemod.cpp:81-83,163-164,172,316-324) and the labels
//             on them. OFF by default: they triple the traffic, need a per-call trace (a full record drain per run) and are for a
//             plotter, not for a receiver. One SF only.
//   setSignals(bool)  the "error" / "power" / "snr" signals (default on)
//   inputs    0 .. B-1            complex float streams, reserve 2N each                       (LoRaDemod.cpp:79,90)
//   outputs   "0" .. "B-1"        Pothos::Packet messages of int16 symbols                     (:80,295-298)
//             "raw<c>" "dec<c>"   complex float streams, `total` elements per work() call      (:81-82,163-164,321-322)   [debug ports]
//             "fft<c>"            complex float stream, N bins per work() call                 (:83,172,324)              [debug ports]
//   signals   "channel" followed by "error", "power", "snr" once per packet at DOWNCHIRP1     (:85-87,267-269)
//   labels    "SYNC", "P x", "DC", "QC", "S<n> x" on raw<c> / dec<c> / fft<c> at the first element each call produced (:314-319) [debug ports]
//
//   input buffers  getInputBufferManager() (the counterpart of LoRaDemod.cpp:346-357, which asks for slabs of >= 2N samples) hands the
//             framework PINNED slabs: all B ports' slabs are carved out of one lorahip_host_alloc block, slab k of port c at
//             (k * B + c) * slabBytes, so that a work() whose inputs all sit in slabs of one generation k uploads them as ONE strided
//             DMA straight from where the upstream blocks wrote them (lorahip_demod_run_host_rows) -- no staging copy, no per-channel
//             call. Inputs anywhere else (another domain's buffers, mixed generations, several parts) take lorahip_demod_run.
//
// One work() of this block performs, per channel, as many LoRaDemod::work() calls as the channel's input buffer allows
// (each needs 2N samples, :148) inside ONE device launch per (device, SF) part. Without the debug ports a work() is: gather the
// input buffers (pinned double-buffered upload), the streaming kernels, 52 B of state per channel back, the packets that
// completed and the signals; nothing per call crosses PCIe. With them, output buffers must hold what several calls produce:
// setMaxWindows(K) sizes them (raw/dec: the samples consumed; fft: 2K frames per work(), more are dropped and counted).
#include <Pothos/Framework.hpp>
#include <algorithm>
#include <complex>
#include <cstdlib>
#include <memory>
#include <cstring>
#include <string>
#include <vector>
#include "lorahip.h"

//! one pinned allocation for the input slabs of all ports: slab k of port c at (k * ports + c) * slabBytes
struct PinnedPool
{
    PinnedPool(const size_t ports, const size_t numBuffers, const size_t slabBytes) :
        ports(ports), numBuffers(numBuffers), slabBytes(slabBytes), base(static_cast<char *>(lorahip_host_alloc(ports * numBuffers * slabBytes))) {}
    ~PinnedPool(void) { if (base) lorahip_host_free(base); }
    size_t address(const size_t k, const size_t port) const { return size_t(base) + (k * ports + port) * slabBytes; }
    const size_t ports, numBuffers, slabBytes;
    char *const base;
};

/*! The buffer manager of ONE input port over its column of the pool, written against Pothos/Framework/BufferManager.hpp as it is:
 * empty() / pop() / push() are the three pure virtuals, the front buffer is published with setFrontBuffer(), and buffers come back
 * through push() when the last reference to their ManagedBuffer is dropped (that is also how init() fills the queue: the buffers it
 * makes go out of scope). Like the framework's "generic" manager the slabs are handed out in slab order -- slab k+1 only after slab k --
 * so that ports fed at the same rate stay in the same generation of the pool, which is what makes a work() one strided DMA. */
class PinnedSlabManager : public Pothos::BufferManager, public std::enable_shared_from_this<PinnedSlabManager>
{
public:
    PinnedSlabManager(const std::shared_ptr<PinnedPool> &pool, const size_t port) : _pool(pool), _port(port), _next(0) {}
    void init(const Pothos::BufferManagerArgs &args)
    {
        Pothos::BufferManager::init(args);
        _slots.assign(_pool->numBuffers, Pothos::ManagedBuffer());
        _next = 0;
        for (size_t k = 0; k < _pool->numBuffers; k++)
        {
            Pothos::ManagedBuffer b;
            b.reset(this->shared_from_this(), Pothos::SharedBuffer(_pool->address(k, _port), _pool->slabBytes, _pool), k);
        }                                                       // (out of scope: returned to this manager through push())
    }
    bool empty(void) const { return _slots.empty() || !_slots[_next]; }
    void pop(const size_t)
    {
        if (this->empty()) return;
        _slots[_next].reset();
        _next = (_next + 1) % _slots.size();
        if (this->empty()) this->setFrontBuffer(Pothos::BufferChunk::null());
        else this->setFrontBuffer(_slots[_next]);
    }
    void push(const Pothos::ManagedBuffer &buff)
    {
        const size_t k = buff.getSlabIndex();
        if (k >= _slots.size()) return;
        _slots[k] = buff;
        if (k == _next) this->setFrontBuffer(buff);
    }
private:
    std::shared_ptr<PinnedPool> _pool;
    const size_t _port;
    std::vector<Pothos::ManagedBuffer> _slots;                  // by slab index; the one at _next is the front
    size_t _next;
};

class LoRaDemodBatch : public Pothos::Block
{
    typedef std::complex<float> cf32;

public:
    LoRaDemodBatch(const size_t sf, const size_t channels) :
        B(channels), _d(nullptr), _sfs(channels, int32_t(sf)), _devices(1, 0), _sync(0x12), _thresh(-30.0), _mtu(256), _maxWindows(64),
        _fftCap(128), _fftDropped(0), _pinnedLimitMiB(12288), _debugPorts(false), _signals(true), _active(false)
    {
        _d = makeDemod(_sfs, _devices);
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setSync));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setDevices));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setSpreadFactors));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setThreshold));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setMTU));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setMaxWindows));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setDebugPorts));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setSignals));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, setPinnedInputLimit));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, slabRowRuns));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, workRuns));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDemodBatch, fftFramesDropped));
        _in.resize(B); _msg.resize(B); _raw.resize(B); _dec.resize(B); _fft.resize(B);
        for (size_t c = 0; c < B; c++)
        {
            const std::string s = std::to_string(c);
            this->setupInput(int(c), typeid(cf32));
            this->setupOutput(int(c));
            this->setupOutput("raw" + s, typeid(cf32));
            this->setupOutput("dec" + s, typeid(cf32));
            this->setupOutput("fft" + s, typeid(cf32));
            // the ports are looked up once: B can be tens of thousands, and work() touches every one of them
            _in[c] = this->input(int(c)); _msg[c] = this->output(int(c));
            _raw[c] = this->output("raw" + s); _dec[c] = this->output("dec" + s); _fft[c] = this->output("fft" + s);
        }
        this->registerSignal("channel");
        this->registerSignal("error");
        this->registerSignal("power");
        this->registerSignal("snr");
        applyReserves();
    }

    ~LoRaDemodBatch(void) { lorahip_demod_destroy(_d); }

    static Block *make(const size_t sf, const size_t channels) { return new LoRaDemodBatch(sf, channels); }

    void setSync(const unsigned char sync) { _sync = sync; lorahip_demod_set_sync(_d, sync); }
    void setThreshold(const double thresh_dB) { _thresh = thresh_dB; lorahip_demod_set_threshold(_d, thresh_dB); }
    void setMTU(const size_t mtu) { _mtu = mtu; lorahip_demod_set_mtu(_d, mtu); }
    void setMaxWindows(const size_t k) { _maxWindows = k ? k : 1; if (_debugPorts) attachPorts(_d); }
    void setSignals(const bool on) { _signals = on; lorahip_demod_set_signals(_d, on ? 1 : 0); }
    size_t fftFramesDropped(void) const { return _fftDropped; }
    //! most pinned host memory (MiB) the input slabs of all ports together may take; above it (or at 0) the ports get the framework's
    //! default buffers and every work() uploads through the library's pinned staging instead. Before the ports are connected.
    void setPinnedInputLimit(const size_t MiB) { _pinnedLimitMiB = MiB; }

    //! the reference block's raw / dec / fft outputs and their labels; off unless asked for (see the head of this file)
    void setDebugPorts(const bool on)
    {
        if (on)
        {
            for (size_t c = 1; c < B; c++)
                if (_sfs[c] != _sfs[0]) throw Pothos::InvalidArgumentException("LoRaDemodBatch::setDebugPorts(true)", "the debug ports need one spreading factor for all channels");
            attachPorts(_d);
        }
        else if (lorahip_demod_set_ports(_d, nullptr) != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch::setDebugPorts(false)", lorahip_last_error());
        _debugPorts = on;
    }

    //! comma-separated device indices, e.g. "0,1,2,3,4,5,6,7"; an index may repeat (two shards on one GPU)
    void setDevices(const std::string &list)
    {
        const std::vector<int> devs = parseList(list, "LoRaDemodBatch::setDevices", 0, 4095);
        rebuild(_sfs, devs, "LoRaDemodBatch::setDevices(" + list + ")");
    }

    //! comma-separated spreading factors, one per channel; a shorter list repeats ("7,8,9,10,11,12" is SF = 7 + c mod 6)
    void setSpreadFactors(const std::string &list)
    {
        const std::vector<int> pat = parseList(list, "LoRaDemodBatch::setSpreadFactors", LORAHIP_SF_MIN, LORAHIP_SF_MAX);
        if (pat.size() > B) throw Pothos::InvalidArgumentException("LoRaDemodBatch::setSpreadFactors(" + list + ")", "more entries than channels");
        std::vector<int32_t> sfs(B);
        for (size_t c = 0; c < B; c++) sfs[c] = int32_t(pat[c % pat.size()]);
        if (_debugPorts) for (size_t c = 1; c < B; c++)
            if (sfs[c] != sfs[0]) throw Pothos::InvalidArgumentException("LoRaDemodBatch::setSpreadFactors(" + list + ")", "the debug ports need one spreading factor for all channels");
        rebuild(sfs, _devices, "LoRaDemodBatch::setSpreadFactors(" + list + ")");
        applyReserves();
    }

    void activate(void) { lorahip_demod_activate(_d); _active = true; }
    void deactivate(void) { _active = false; }

    void work(void)
    {
        _streams.resize(B); _avail.resize(B);
        bool any = false;
        for (size_t c = 0; c < B; c++)
        {
            const size_t N = size_t(1) << _sfs[c], capSamples = _maxWindows * N;
            _streams[c] = reinterpret_cast<const float *>(_in[c]->buffer().template as<const cf32 *>());
            // with the debug ports: never produce more than the output buffers hold
            _avail[c] = (_debugPorts && _in[c]->elements() > capSamples) ? capSamples : _in[c]->elements();
            any = any || _avail[c] >= 2 * N;                                            // :148
        }
        if (!any) return;
        _workRuns++;
        if (_debugPorts) lorahip_demod_set_trace(_d, 1);                                // labels and per-call signals come from the trace

        // every part's channels in one launch on its device; the parts run side by side, each from its own host thread (inside the library)
        if (inputsAreOneSlabGeneration())
        {
            // all inputs sit in pinned slabs of one generation of this block's own pool: rows of one host block, one strided DMA
            if (lorahip_demod_run_host_rows(_d, reinterpret_cast<const float *>(_rowBase), _pool->slabBytes / sizeof(cf32), _first.data(), _avail.data(), nullptr) != LORAHIP_OK)
                throw Pothos::Exception("LoRaDemodBatch::work()", lorahip_last_error());
            _rowRuns++;
        }
        else if (lorahip_demod_run(_d, _streams.data(), _avail.data(), nullptr) != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch::work()", lorahip_last_error());

        _consumed.resize(B);
        if (lorahip_demod_consumed_all(_d, _consumed.data()) != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch::work()", lorahip_last_error());
        for (size_t c = 0; c < B; c++) if (_consumed[c] > 0) _in[c]->consume(size_t(_consumed[c]));   // the sum of consume(total), :320

        if (_debugPorts) producePorts();
        else if (_signals)
        {
            // once per packet at DOWNCHIRP1 (:267-269): the kernels kept a record per emission
            const size_t n = lorahip_demod_num_signals(_d);
            _sigCh.resize(n); _sigErr.resize(n); _sigPow.resize(n); _sigSnr.resize(n);
            if (n && lorahip_demod_get_signals(_d, _sigCh.data(), nullptr, _sigErr.data(), _sigPow.data(), _sigSnr.data(), n) != LORAHIP_OK)
                throw Pothos::Exception("LoRaDemodBatch::work()", lorahip_last_error());
            for (size_t i = 0; i < n; i++)
            {
                this->emitSignal("channel", int(_sigCh[i]));
                this->emitSignal("error", int(_sigErr[i]));
                this->emitSignal("power", _sigPow[i]);
                this->emitSignal("snr", _sigSnr[i]);
            }
        }

        // packets (:295-298): all of them in one call; every channel has its own message port
        const size_t nPackets = lorahip_demod_num_packets(_d), nSyms = lorahip_demod_num_packet_symbols(_d);
        _pkCh.resize(nPackets); _pkLen.resize(nPackets); _pkSyms.resize(nSyms);
        if (nPackets && lorahip_demod_get_packets(_d, _pkCh.data(), nullptr, _pkLen.data(), nPackets, _pkSyms.data(), nSyms) != LORAHIP_OK)
            throw Pothos::Exception("LoRaDemodBatch::work()", lorahip_last_error());
        // ONE buffer for all packets of this work(), every payload a view into it (a BufferChunk is a view with shared ownership: the
        // buffer lives as long as any of the messages) -- not an allocation per packet, of which a work() posts tens of thousands
        if (nPackets)
        {
            Pothos::BufferChunk all(typeid(int16_t), nSyms ? nSyms : 1);
            if (nSyms) std::memcpy(all.template as<int16_t *>(), _pkSyms.data(), nSyms * sizeof(int16_t));
            size_t at = 0;
            for (size_t i = 0; i < nPackets; i++)
            {
                const size_t len = size_t(_pkLen[i]);
                Pothos::Packet pkt;
                pkt.payload = all;
                pkt.payload.address = all.address + at * sizeof(int16_t);
                pkt.payload.length = len * sizeof(int16_t);
                at += len;
                _msg[size_t(_pkCh[i])]->postMessage(pkt);
            }
        }
        lorahip_demod_clear_packets(_d);
        if (_debugPorts) lorahip_demod_set_trace(_d, 0);                                // the next work() starts a fresh trace
    }

    //! work() calls that uploaded their inputs as rows of the pinned pool (one strided DMA), and those that ran at all: a ratio below
    //! one means ports out of step (different generations) or inputs that were not this block's slabs
    size_t slabRowRuns(void) const { return _rowRuns; }
    size_t workRuns(void) const { return _workRuns; }

    /*! Input buffers for the upstream blocks to write into (the reference: "slabs large enough for fft input", LoRaDemod.cpp:346-357).
     * Here: PINNED slabs of (maxWindows + 2) symbols, all ports' slabs in one allocation (PinnedPool), so that what the framework
     * presents to work() can cross PCIe as one strided DMA. Another domain's memory is not ours to manage: the framework's default. */
    Pothos::BufferManager::Sptr getInputBufferManager(const std::string &name, const std::string &domain)
    {
        if (!domain.empty() || name.empty() || name.find_first_not_of("0123456789") != std::string::npos) return Pothos::Block::getInputBufferManager(name, domain);
        const size_t c = size_t(std::atol(name.c_str()));
        if (c >= B) return Pothos::Block::getInputBufferManager(name, domain);
        // one row stride for all ports: the pool is for ONE spreading factor (a mixed-SF block would pin the largest SF's slab for every
        // port -- 32x too much for its SF7 channels); such a block, and one whose pool would not fit the limit, gets the default buffers
        for (size_t k = 1; k < B; k++) if (_sfs[k] != _sfs[0]) return Pothos::Block::getInputBufferManager(name, domain);
        Pothos::BufferManagerArgs args;
        const size_t N = size_t(1) << _sfs[0];
        const size_t slabBytes = std::max(args.bufferSize, (_maxWindows + 2) * N * sizeof(cf32));            // >= 2N: the reference's bound (:352-353)
        // (a pool of another slab size -- setMaxWindows / setSpreadFactors since the ports were last asked for -- is left to the managers
        // that hold it; ports on different pools are uploaded piece by piece, see inputsAreOneSlabGeneration)
        if (!_pool || _pool->slabBytes != slabBytes)
        {
            // generations: one inside work(), one being filled by the upstream blocks, one free so that a producer that runs ahead does
            // not stall (the framework's default is 4); two if three do not fit the limit
            const size_t limit = _pinnedLimitMiB << 20;
            size_t gens = 3;
            if (B * gens * slabBytes > limit) gens = 2;
            if (B * gens * slabBytes > limit) return Pothos::Block::getInputBufferManager(name, domain);
            std::shared_ptr<PinnedPool> pool(new PinnedPool(B, gens, slabBytes));
            if (pool->base == nullptr) return Pothos::Block::getInputBufferManager(name, domain);            // no pinned memory to be had: the default
            _pool = pool;
        }
        args.numBuffers = _pool->numBuffers; args.bufferSize = _pool->slabBytes;
        std::shared_ptr<PinnedSlabManager> m(new PinnedSlabManager(_pool, c));
        m->init(args);
        return m;
    }

    //! output buffers large enough for what one work() produces (the reference does the same for its 2N / N, :330-358)
    Pothos::BufferManager::Sptr getOutputBufferManager(const std::string &name, const std::string &domain)
    {
        if (name.compare(0, 3, "raw") == 0 || name.compare(0, 3, "dec") == 0 || name.compare(0, 3, "fft") == 0)
        {
            const size_t c = size_t(std::atol(name.c_str() + 3));
            const size_t N = size_t(1) << _sfs[c < B ? c : 0];
            Pothos::BufferManagerArgs args;
            args.bufferSize = (name.compare(0, 3, "fft") == 0 ? _fftCap : _maxWindows) * N * sizeof(cf32);
            return Pothos::BufferManager::make("generic", args);
        }
        return Pothos::Block::getOutputBufferManager(name, domain);
    }

private:
    //! do all inputs with samples lie in slabs of ONE generation of the pool, each in its own port's column? (fills _first, _rowBase)
    bool inputsAreOneSlabGeneration(void)
    {
        if (!_pool || _debugPorts) return false;
        const size_t gen = _pool->ports * _pool->slabBytes, lo = size_t(_pool->base), hi = lo + _pool->numBuffers * gen;
        _first.resize(B);
        size_t k = size_t(-1);
        for (size_t c = 0; c < B; c++)
        {
            _first[c] = 0;
            if (_avail[c] == 0) continue;
            const size_t a = size_t(_streams[c]);
            if (a < lo || a >= hi) return false;
            const size_t kc = (a - lo) / gen, off = (a - lo) - kc * gen;
            if (off / _pool->slabBytes != c || (k != size_t(-1) && kc != k)) return false;
            if (off - c * _pool->slabBytes + _avail[c] * sizeof(cf32) > _pool->slabBytes) return false;
            k = kc;
            _first[c] = int64_t((off - c * _pool->slabBytes) / sizeof(cf32));
        }
        if (k == size_t(-1)) return false;
        _rowBase = _pool->base + k * gen;
        return true;
    }

    //! what the reference block does on its three stream outputs, from the per-call trace (one SF: checked by setDebugPorts)
    void producePorts(void)
    {
        const size_t N = size_t(1) << _sfs[0], capSamples = _maxWindows * N;
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
            std::memcpy(_raw[c]->buffer().template as<cf32 *>(), _stRaw.data() + c * capSamples, nr * sizeof(cf32));
            std::memcpy(_dec[c]->buffer().template as<cf32 *>(), _stDec.data() + c * capSamples, nd * sizeof(cf32));
            std::memcpy(_fft[c]->buffer().template as<cf32 *>(), _stFft.data() + c * _fftCap * N, frames * N * sizeof(cf32));

            // labels at the first element each call produced (:314-319); signals (:267-269)
            size_t pos = 0;
            const char *lab = labels.data();
            for (size_t k = 0; k < nCalls; k++)
            {
                const std::string id(lab);
                lab += id.size() + 1;
                if (!id.empty())
                {
                    _raw[c]->postLabel(Pothos::Label(id, Pothos::Object(), pos));
                    _dec[c]->postLabel(Pothos::Label(id, Pothos::Object(), pos));
                    if (k < frames) _fft[c]->postLabel(Pothos::Label(id, Pothos::Object(), k * N));
                }
                if (tr[k].signals && _signals)
                {
                    this->emitSignal("channel", int(c));
                    this->emitSignal("error", tr[k].sig_error);
                    this->emitSignal("power", tr[k].sig_power);
                    this->emitSignal("snr", tr[k].sig_snr);
                }
                pos += size_t(tr[k].consumed);
            }
            _raw[c]->produce(nr);
            _dec[c]->produce(nd);
            _fft[c]->produce(frames * N);
        }
    }

    static std::vector<int> parseList(const std::string &list, const std::string &what, const int lo, const int hi)
    {
        std::vector<int> out;
        size_t at = 0;
        while (at < list.size())
        {
            size_t end = list.find(',', at);
            if (end == std::string::npos) end = list.size();
            const std::string tok = list.substr(at, end - at);
            if (tok.empty() || tok.find_first_not_of("0123456789 ") != std::string::npos || tok.find_first_of("0123456789") == std::string::npos)
                throw Pothos::InvalidArgumentException(what + "(" + list + ")", "not a list of non-negative integers");
            const long v = std::atol(tok.c_str());
            if (v < lo || v > hi) throw Pothos::InvalidArgumentException(what + "(" + list + ")", "value out of range");
            out.push_back(int(v));
            at = end + 1;
        }
        if (out.empty()) throw Pothos::InvalidArgumentException(what + "()", "empty list");
        return out;
    }

    //! a level-3 object with this block's settings applied; throws (and leaves the block as it was) if it cannot be made
    lorahip_demod *makeDemod(const std::vector<int32_t> &sfs, const std::vector<int> &devices)
    {
        lorahip_demod *d = nullptr;
        const int rc = lorahip_demod_create_mixed(&d, devices.data(), devices.size(), sfs.data(), sfs.size());
        if (rc != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch", std::string(lorahip_strerror(rc)) + " " + lorahip_last_error());
        lorahip_demod_set_sync(d, _sync); lorahip_demod_set_threshold(d, _thresh); lorahip_demod_set_mtu(d, _mtu);
        lorahip_demod_set_signals(d, _signals ? 1 : 0);
        return d;
    }

    //! Another device list or SF list: the NEW object is made first, with the settings and ports of the old one; only when all of that
    //! has succeeded does it replace the old one (a failure leaves the block exactly as it was). Refused while the block is active:
    //! the channels' frame machines and open packets live in the object that would be thrown away.
    void rebuild(const std::vector<int32_t> &sfs, const std::vector<int> &devices, const std::string &what)
    {
        if (_active) throw Pothos::Exception(what, "not while the block is active (the channels' receiver state would be lost)");
        lorahip_demod *d = makeDemod(sfs, devices);
        const std::vector<int32_t> oldSfs = _sfs;
        _sfs = sfs;                                                 // attachPorts sizes by the new list
        if (_debugPorts)
        {
            try { attachPorts(d); }
            catch (...) { _sfs = oldSfs; lorahip_demod_destroy(d); if (_debugPorts) attachPorts(_d); throw; }
        }
        lorahip_demod_destroy(_d);
        _d = d;
        _devices = devices;
    }

    void applyReserves(void) { for (size_t c = 0; c < B; c++) _in[c]->setReserve((size_t(1) << _sfs[c]) * 2); }   // at most two input symbols (:90)

    //! host staging of the three debug ports, [channel][capacity], handed to the library as host buffers
    void attachPorts(lorahip_demod *d)
    {
        const size_t N = size_t(1) << _sfs[0], capSamples = _maxWindows * N;
        _fftCap = 2 * _maxWindows;
        _stRaw.assign(B * capSamples, cf32());
        _stDec.assign(B * capSamples, cf32());
        _stFft.assign(B * _fftCap * N, cf32());
        lorahip_demod_ports p;
        std::memset(&p, 0, sizeof(p));
        p.struct_size = sizeof(p);
        p.fft_dev = reinterpret_cast<float *>(_stFft.data()); p.fft_cap_frames = _fftCap;
        p.dec_dev = reinterpret_cast<float *>(_stDec.data()); p.dec_cap_samples = capSamples;
        p.raw_dev = reinterpret_cast<float *>(_stRaw.data()); p.raw_cap_samples = capSamples;
        p.host_buffers = 1;
        if (lorahip_demod_set_ports(d, &p) != LORAHIP_OK) throw Pothos::Exception("LoRaDemodBatch", lorahip_last_error());
    }

    const size_t B;
    lorahip_demod *_d;                                      // one handle: (device, SF) parts inside the library
    std::vector<int32_t> _sfs;                              // per channel
    std::vector<int> _devices;
    unsigned char _sync; double _thresh; size_t _mtu;       // the setters' values, re-applied when the object is rebuilt
    size_t _maxWindows, _fftCap, _fftDropped, _pinnedLimitMiB;
    bool _debugPorts, _signals, _active;
    std::vector<Pothos::InputPort *> _in;
    std::vector<Pothos::OutputPort *> _msg, _raw, _dec, _fft;
    std::vector<const float *> _streams; std::vector<size_t> _avail; std::vector<int64_t> _consumed;
    std::vector<int32_t> _pkCh; std::vector<int64_t> _pkLen; std::vector<int16_t> _pkSyms;
    std::vector<int32_t> _sigCh, _sigErr; std::vector<float> _sigPow, _sigSnr;
    std::vector<cf32> _stRaw, _stDec, _stFft;               // host staging of the three ports, [channel][capacity]
    std::shared_ptr<PinnedPool> _pool;                      // the input slabs of all ports (getInputBufferManager)
    std::vector<int64_t> _first; const char *_rowBase = nullptr; size_t _rowRuns = 0, _workRuns = 0;
};

static Pothos::BlockRegistry registerLoRaDemodBatch("/lora/lora_demod_batch", &LoRaDemodBatch::make);
