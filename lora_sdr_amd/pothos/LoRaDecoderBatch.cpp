// LoRaDecoderBatch.cpp -- a Pothos block that does the work of B instances of the reference's /lora/lora_decoder (LoRaDecoder.cpp) on
// the GPU through lorahip_decode_packets_host (include/lorahip.h): the reference-side binding of the batched decoder (SURVEY.md section
// 8f #2), as a real translation unit. It compiles against <Pothos/Framework.hpp> and links liblorahip.so.
//
// Same parameters, defaults and setters as LoRaDecoder (LoRaDecoder.cpp:97-196), the same decisions in the same order per packet
// (the kernel keeps the block's order: header, Hamming / parity, CRC), the same "dropped" signal; what changes is the shape: one block
// instance has B message inputs -- the B message outputs of /lora/lora_demod_batch -- and one work() decodes EVERY message that is
// waiting on any of them in one device launch.
//
//   factory   /lora/lora_decoder_batch(channels)
//   setters   setSpreadFactor, setSymbolSize, setCodingRate("4/4".."4/8"), enableWhitening, enableCrcc, enableInterleaving,
//             enableExplicit, enableHdr, setDataLength, enableErrorCheck; getDropped                       (LoRaDecoder.cpp:111-121)
//             setDevice(index): the GPU that decodes (default 0)
//   inputs    "0" .. "B-1"   Pothos::Packet messages, payload = uint16 LoRa symbols                      (:24-27, :204-211)
//   outputs   "0" .. "B-1"   Pothos::Packet messages, payload = the decoded bytes (uint16 codewords when interleaving is off)
//   signals   "dropped" (the running count) on activate() and on every drop, as the reference block     (:192, :401-405)
//
// enableWhitening is accepted and stored like the reference's; the reference's work() never reads _whitening (the whitening
// sequence is always applied, LoRaDecoder.cpp:228-249), and neither does this block.
#include <Pothos/Framework.hpp>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "lorahip.h"

class LoRaDecoderBatch : public Pothos::Block
{
public:
    LoRaDecoderBatch(const size_t channels) :
        B(channels), _ctx(nullptr), _device(0), _sf(10), _ppm(0), _rdd(4), _whitening(true), _crcc(false), _interleaving(true), _errorCheck(false),
        _explicit(true), _hdr(false), _dataLength(8), _dropped(0)
    {
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, setSpreadFactor));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, setSymbolSize));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, setCodingRate));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, enableWhitening));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, enableCrcc));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, enableInterleaving));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, enableExplicit));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, enableHdr));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, setDataLength));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, enableErrorCheck));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, setDevice));
        this->registerCall(this, POTHOS_FCN_TUPLE(LoRaDecoderBatch, getDropped));
        this->registerSignal("dropped");
        _in.resize(B); _out.resize(B);
        for (size_t c = 0; c < B; c++)
        {
            this->setupInput(int(c));
            this->setupOutput(int(c));
            _in[c] = this->input(int(c)); _out[c] = this->output(int(c));
        }
    }

    ~LoRaDecoderBatch(void) { if (_ctx) lorahip_destroy(_ctx); }

    static Block *make(const size_t channels) { return new LoRaDecoderBatch(channels); }

    void setSpreadFactor(const size_t sf) { _sf = sf; }
    void setSymbolSize(const size_t ppm) { _ppm = ppm; }
    void setCodingRate(const std::string &cr)
    {
        if (cr == "4/4") _rdd = 0;
        else if (cr == "4/5") _rdd = 1;
        else if (cr == "4/6") _rdd = 2;
        else if (cr == "4/7") _rdd = 3;
        else if (cr == "4/8") _rdd = 4;
        else throw Pothos::InvalidArgumentException("LoRaDecoderBatch::setCodingRate(" + cr + ")", "unknown coding rate");
    }
    void enableWhitening(const bool whitening) { _whitening = whitening; }
    void enableInterleaving(const bool interleaving) { _interleaving = interleaving; }
    void enableExplicit(const bool on) { _explicit = on; }
    void enableHdr(const bool hdr) { _hdr = hdr; }
    void enableErrorCheck(const bool errorCheck) { _errorCheck = errorCheck; }
    void enableCrcc(const bool crcc) { _crcc = crcc; }
    void setDataLength(const size_t dataLength) { _dataLength = dataLength; }
    void setDevice(const size_t device)
    {
        if (_ctx && int(device) != _device) { lorahip_destroy(_ctx); _ctx = nullptr; }
        _device = int(device);
    }
    unsigned long long getDropped(void) const { return _dropped; }

    void activate(void)
    {
        _dropped = 0;
        this->emitSignal("dropped", _dropped);                              // LoRaDecoder.cpp:189-193
        // (the decoder context carries the device and a stream; its SF is not the packets' SF -- that comes with every call)
        if (_ctx == nullptr && lorahip_create(&_ctx, _device, 7) != LORAHIP_OK)
            throw Pothos::Exception("LoRaDecoderBatch::activate()", std::string(lorahip_last_error()));
    }

    void work(void)
    {
        const size_t PPM = (_ppm == 0) ? _sf : _ppm;
        // every message that waits on any input: one row each. Too short for a header: nothing is posted (LoRaDecoder.cpp:208).
        const size_t N_HEADER_SYMBOLS = 8;
        _rowCh.clear(); _rowLen.clear(); _msgs.clear();
        size_t longest = 0;
        for (size_t c = 0; c < B; c++)
            while (_in[c]->hasMessage())
            {
                if (PPM > _sf) throw Pothos::Exception("LoRaDecoderBatch::work()", "failed check: PPM <= SF");     // :200-201, before the message is taken
                const Pothos::Packet pkt = _in[c]->popMessage().template extract<Pothos::Packet>();
                const size_t n = pkt.payload.elements();
                if (n < N_HEADER_SYMBOLS) continue;
                _rowCh.push_back(int32_t(c)); _rowLen.push_back(int32_t(n)); _msgs.push_back(pkt);
                if (n > longest) longest = n;
            }
        const size_t P = _msgs.size();
        if (P == 0) return;
        if (_ctx == nullptr) this->activate();

        // rows as long as the longest message (the reference sizes its vectors from the message, LoRaDecoder.cpp:210-213); one beyond the
        // library's row limit keeps its length, and is an error below unless the symbols its length needs lie inside the row
        const size_t maxSyms = size_t(lorahip_decode_max_symbols());
        const size_t stride = longest < maxSyms ? ((longest + 7) & ~size_t(7)) : maxSyms;
        const size_t outStride = 2 * (stride + 8);
        _syms.assign(P * stride, 0);
        for (size_t p = 0; p < P; p++)
        {
            const size_t n = size_t(_rowLen[p]) < stride ? size_t(_rowLen[p]) : stride;
            std::memcpy(_syms.data() + p * stride, _msgs[p].payload.template as<const void *>(), n * sizeof(uint16_t));
        }
        _outBytes.resize(P * outStride); _outLen.resize(P); _drop.resize(P);

        lorahip_decoder_cfg cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.struct_size = sizeof(cfg);
        cfg.sf = int32_t(_sf); cfg.ppm = int32_t(_ppm); cfg.rdd = int32_t(_rdd); cfg.crcc = _crcc; cfg.interleaving = _interleaving;
        cfg.error_check = _errorCheck; cfg.explicit_hdr = _explicit; cfg.hdr = _hdr; cfg.data_length = int32_t(_dataLength);
        if (lorahip_decode_packets_host(_ctx, &cfg, _syms.data(), stride, _rowLen.data(), P, _outBytes.data(), outStride, _outLen.data(), _drop.data()) != LORAHIP_OK)
            throw Pothos::Exception("LoRaDecoderBatch::work()", std::string(lorahip_last_error()));

        // in arrival order per channel: the message the reference block would have posted (:390-396), or its drop() (:401-405)
        for (size_t p = 0; p < P; p++)
        {
            if (_drop[p]) { _dropped++; this->emitSignal("dropped", _dropped); }
            if (_outLen[p] == -2)                                           // never a silent loss: the message could not be decoded as the reference would
                throw Pothos::Exception("LoRaDecoderBatch::work()", "a message of " + std::to_string(_rowLen[p]) + " symbols is longer than this build decodes (lorahip_decode_max_symbols)");
            if (_outLen[p] < 0) continue;                                   // nothing posted: too short for a header or dropped, as in the reference
            const size_t n = size_t(_outLen[p]);
            Pothos::Packet out;
            if (_interleaving)
            {
                out.payload = Pothos::BufferChunk(typeid(uint8_t), n ? n : 1);
                out.payload.length = n;
                if (n) std::memcpy(out.payload.template as<void *>(), _outBytes.data() + p * outStride, n);
            }
            else
            {
                // interleaving off: the block posts its uint16 symbols as they stand after the Gray step (:252-258 of the reference)
                out.payload = Pothos::BufferChunk(typeid(uint16_t), n ? n : 1);
                out.payload.length = n * sizeof(uint16_t);
                if (n) std::memcpy(out.payload.template as<void *>(), _outBytes.data() + p * outStride, n * sizeof(uint16_t));
            }
            _out[size_t(_rowCh[p])]->postMessage(out);
        }
    }

private:
    const size_t B;
    lorahip_ctx *_ctx;
    int _device;
    size_t _sf, _ppm, _rdd;
    bool _whitening, _crcc, _interleaving, _errorCheck, _explicit, _hdr;
    size_t _dataLength;
    unsigned long long _dropped;
    std::vector<Pothos::InputPort *> _in;
    std::vector<Pothos::OutputPort *> _out;
    std::vector<int32_t> _rowCh, _rowLen, _outLen, _drop;
    std::vector<Pothos::Packet> _msgs;
    std::vector<uint16_t> _syms;
    std::vector<uint8_t> _outBytes;
};

static Pothos::BlockRegistry registerLoRaDecoderBatch("/lora/lora_decoder_batch", &LoRaDecoderBatch::make);
