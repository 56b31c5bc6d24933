// TEST INFRASTRUCTURE ONLY (oracle/): a minimal stand-in for <Pothos/Framework.hpp>.
//
// Pothos is not installed in this image, so the reference's LoRaDemod.cpp cannot be
// built against the real framework. This header provides just enough of the Pothos
// surface that LoRaDemod.cpp touches (SURVEY.md §8b symbol list; LoRaDemod.cpp:76-94,
// 147-154, 295-298, 316-324, 330-358, 395), LoRaMod.cpp (:65-70, 97, 111-132, 226-250) and the codec
// blocks LoRaEncoder.cpp / LoRaDecoder.cpp (string / bool setters, message ports, Pothos::Exception) touch, for the files to compile VERBATIM from /root/reference and be driven by
// oracle/ref_driver.cpp. It is a recording fake:
// ports are plain host buffers owned by the driver, labels / messages / signals are
// appended to per-block logs. Nothing here is product code.
#pragma once
#include <complex>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <new>
#include <deque>
#include <sstream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <typeinfo>
#include <vector>

namespace Pothos {

struct DType
{
    DType(void) : size(1) {}
    DType(const std::type_info &t) : size(1)
    {
        if (t == typeid(std::complex<float>)) size = sizeof(std::complex<float>);
        else if (t == typeid(int16_t)) size = sizeof(int16_t);
        else if (t == typeid(uint16_t)) size = sizeof(uint16_t);
        else if (t == typeid(uint8_t)) size = sizeof(uint8_t);
        else if (t == typeid(float)) size = sizeof(float);
    }
    size_t size;
};

//! memory + the thing that keeps it alive (Pothos/Framework/SharedBuffer.hpp: the container form is how a block wraps memory of its own)
struct SharedBuffer
{
    SharedBuffer(void) : _address(0), _length(0) {}
    SharedBuffer(const size_t address, const size_t length, std::shared_ptr<void> container) : _address(address), _length(length), _container(container) {}
    size_t getAddress(void) const { return _address; }
    size_t getLength(void) const { return _length; }
    size_t _address, _length;
    std::shared_ptr<void> _container;
};

class BufferManager;
/*! A buffer that belongs to a manager's pool (Pothos/Framework/ManagedBuffer.hpp). Reference counted like the real one: when the last
 * copy -- and the last BufferChunk that refers to it -- is gone, the buffer goes back to its manager through BufferManager::push();
 * that is the ONLY way the framework returns buffers, so a custom manager that relies on anything else does not work here either. */
class ManagedBuffer
{
public:
    ManagedBuffer(void) : _impl(nullptr) {}
    ManagedBuffer(const ManagedBuffer &o) : _impl(o._impl) { if (_impl) _impl->counter++; }
    ManagedBuffer &operator=(const ManagedBuffer &o) { if (o._impl) o._impl->counter++; release(); _impl = o._impl; return *this; }
    ~ManagedBuffer(void) { release(); }
    void reset(void) { release(); _impl = nullptr; }
    void reset(std::shared_ptr<BufferManager> manager, const SharedBuffer &buff, const size_t slabIndex = 0)
    {
        release();
        _impl = new Impl();
        _impl->counter = 1; _impl->weakManager = manager; _impl->buffer = buff; _impl->slabIndex = slabIndex;
    }
    explicit operator bool(void) const { return _impl != nullptr; }
    const SharedBuffer &getBuffer(void) const { return _impl->buffer; }
    size_t getSlabIndex(void) const { return _impl->slabIndex; }
    std::shared_ptr<BufferManager> getBufferManager(void) const { return _impl ? _impl->weakManager.lock() : std::shared_ptr<BufferManager>(); }
    size_t useCount(void) const { return _impl ? size_t(_impl->counter) : 0; }
private:
    struct Impl { int counter; std::weak_ptr<BufferManager> weakManager; SharedBuffer buffer; size_t slabIndex; };
    inline void release(void);                                  // (defined behind BufferManager)
    Impl *_impl;
};

//! shared-ownership buffer, copy = alias (same as the real BufferChunk); a chunk made from a ManagedBuffer keeps that buffer out of its
//! manager's pool for as long as the chunk (or a copy) lives
struct BufferChunk
{
    BufferChunk(void) : address(0), length(0), elemSize(1) {}
    BufferChunk(const DType &dtype, const size_t numElems) :
        address(0), length(dtype.size * numElems), elemSize(dtype.size), _mem(alloc(dtype.size, numElems), std::default_delete<char[]>())
    {
        std::memset(_mem.get(), 0, length + 16);
        address = size_t(_mem.get());
    }
    BufferChunk(const ManagedBuffer &b) : address(b ? b.getBuffer().getAddress() : 0), length(b ? b.getBuffer().getLength() : 0), elemSize(1), _managed(b) {}
    static const BufferChunk &null(void) { static const BufferChunk n; return n; }
    //! a request the real framework could never satisfy (a size_t that wrapped around) must fail, not wrap again
    static char *alloc(const size_t elemSize, const size_t numElems)
    {
        if (numElems > (size_t(1) << 40)) throw std::bad_alloc();
        return new char[elemSize * numElems + 16];
    }
    //! view of externally owned memory (driver side)
    static BufferChunk view(void *p, const size_t bytes)
    {
        BufferChunk b;
        b.address = size_t(p);
        b.length = bytes;
        return b;
    }
    template <typename T> T as(void) const { return reinterpret_cast<T>(address); }
    size_t elements(void) const { return length / elemSize; }
    const ManagedBuffer &getManagedBuffer(void) const { return _managed; }
    size_t address;
    size_t length;
    size_t elemSize;
    std::shared_ptr<char> _mem;
    ManagedBuffer _managed;
};

struct Packet
{
    BufferChunk payload;
};

//! the only payload a message carries in these blocks is a Packet
struct Object
{
    Object(void) {}
    explicit Object(const Packet &p) : _pkt(p) {}
    template <typename T> T extract(void) const { return _pkt; }
    Packet _pkt;
};

struct Label
{
    template <typename IdT>
    Label(const IdT &id, const Object &, const size_t index) : id(id), index(index) {}
    std::string id;
    size_t index;
};

struct InvalidArgumentException : public std::runtime_error
{
    InvalidArgumentException(const std::string &what, const std::string &why) : std::runtime_error(what + ": " + why) {}
};

struct Exception : public std::runtime_error
{
    Exception(const std::string &what, const std::string &why) : std::runtime_error(what + ": " + why) {}
};

struct BufferManagerArgs
{
    BufferManagerArgs(void) : numBuffers(4), bufferSize(8192), nodeAffinity(-1) {}
    size_t numBuffers;
    size_t bufferSize;
    long nodeAffinity;
};

/*! Pothos/Framework/BufferManager.hpp with the surface of the real class and no more: empty() / pop() / push() are pure virtual, the
 * front buffer and the initialised flag are private (a manager publishes its front through setFrontBuffer), there are no queues to
 * inherit. make("generic", args) is the framework's own heap manager (GenericBufferManager below). The recording driver plays the
 * framework: it asks the block for its managers, takes front() / pop()s, and lets go of the chunks -- which returns them via push(). */
class BufferManager
{
public:
    typedef std::shared_ptr<BufferManager> Sptr;
    virtual ~BufferManager(void) {}
    static inline Sptr make(const std::string &name, const BufferManagerArgs &args);
    virtual void init(const BufferManagerArgs &) { _initialized = true; }
    virtual bool empty(void) const = 0;
    const BufferChunk &front(void) const { return _frontBuffer; }
    virtual void pop(const size_t numBytes) = 0;
    virtual void push(const ManagedBuffer &buff) = 0;
    bool isInitialized(void) const { return _initialized; }
protected:
    BufferManager(void) : _initialized(false) {}
    void setFrontBuffer(const BufferChunk &buff) { _frontBuffer = buff; }
private:
    bool _initialized;
    BufferChunk _frontBuffer;
};

inline void ManagedBuffer::release(void)
{
    if (_impl == nullptr || --_impl->counter != 0) return;
    Impl *impl = _impl;
    _impl = nullptr;
    std::shared_ptr<BufferManager> manager = impl->weakManager.lock();
    if (!manager) { delete impl; return; }
    // the last reference is gone: back to the manager (which takes a new reference)
    ManagedBuffer again;
    again._impl = impl;
    impl->counter = 1;
    manager->push(again);
    if (--impl->counter == 0) delete impl;                      // (a manager that did not keep it)
    again._impl = nullptr;
}

//! the framework's "generic" manager: numBuffers heap buffers handed out in slab order
class GenericBufferManager : public BufferManager, public std::enable_shared_from_this<GenericBufferManager>
{
public:
    void init(const BufferManagerArgs &a)
    {
        BufferManager::init(a);
        _slots.assign(a.numBuffers, ManagedBuffer()); _next = 0; _count = 0;
        std::shared_ptr<char> mem(new char[(a.bufferSize ? a.bufferSize : 1) * a.numBuffers], std::default_delete<char[]>());
        for (size_t i = 0; i < a.numBuffers; i++)
        {
            ManagedBuffer b;
            b.reset(this->shared_from_this(), SharedBuffer(size_t(mem.get()) + i * a.bufferSize, a.bufferSize, mem), i);
        }                                                       // (leaving scope returns each buffer to this manager: push)
    }
    bool empty(void) const { return _slots.empty() || !_slots[_next]; }
    void pop(const size_t)
    {
        if (empty()) return;
        _slots[_next].reset(); _count--;
        _next = (_next + 1) % _slots.size();
        this->setFrontBuffer(empty() ? BufferChunk::null() : BufferChunk(_slots[_next]));
    }
    void push(const ManagedBuffer &buff)
    {
        _slots.at(buff.getSlabIndex()) = buff; _count++;
        if (buff.getSlabIndex() == _next) this->setFrontBuffer(BufferChunk(buff));
    }
private:
    std::vector<ManagedBuffer> _slots; size_t _next = 0, _count = 0;
};

inline BufferManager::Sptr BufferManager::make(const std::string &, const BufferManagerArgs &args)
{
    Sptr m(new GenericBufferManager());
    m->init(args);
    return m;
}

struct InputPort
{
    InputPort(void) : reserve(0), _elems(0), consumed(0) {}
    void setReserve(const size_t n) { reserve = n; }
    size_t elements(void) const { return _elems; }
    const BufferChunk &buffer(void) const { return _buff; }
    void consume(const size_t n) { consumed += n; }
    bool hasMessage(void) const { return !_msgs.empty(); }
    Object popMessage(void) { Object o = _msgs.front(); _msgs.pop_front(); return o; }
    std::deque<Object> _msgs;
    size_t reserve;
    size_t _elems;
    BufferChunk _buff;
    size_t consumed;
};

struct OutputPort
{
    OutputPort(void) : reserve(0), produced(0) {}
    void setReserve(const size_t n) { reserve = n; }
    const BufferChunk &buffer(void) const { return _buff; }
    void produce(const size_t n) { produced += n; }
    void postLabel(const Label &l) { labels.push_back(l); }
    template <typename T> void postMessage(const T &m) { _post(m); }
    void _post(const Packet &p)
    {
        //deep-copy at post time: the block may keep writing to the shared chunk
        std::vector<char> bytes(p.payload.length);
        if (p.payload.length) std::memcpy(bytes.data(), p.payload.as<const char *>(), p.payload.length);
        messages.push_back(bytes);
    }
    size_t reserve;
    BufferChunk _buff;
    size_t produced;
    std::vector<Label> labels;
    std::vector<std::vector<char>> messages;
};

struct SignalRecord
{
    std::string name;
    double value;
};

class Block
{
public:
    virtual ~Block(void) {}
    virtual void activate(void) {}
    virtual void deactivate(void) {}
    virtual void work(void) {}

    //! setters become name -> callable(double) so the driver can reach non-virtual members
    template <typename C, typename A>
    void registerCall(C *obj, const char *name, void (C::*m)(A))
    {
        calls[name] = [obj, m](const double v) { (obj->*m)(static_cast<A>(v)); };
    }
    template <typename C>
    void registerCall(C *obj, const char *name, void (C::*m)(const std::string &))
    {
        stringCalls[name] = [obj, m](const std::string &v) { (obj->*m)(v); };
    }
    //! getters: the arithmetic ones can be read by the driver (name -> double), the others are not driven
    template <typename C, typename R>
    typename std::enable_if<std::is_arithmetic<R>::value>::type registerCall(C *obj, const char *name, R (C::*m)(void) const)
    {
        getters[name] = [obj, m](void) { return double((obj->*m)()); };
    }
    template <typename C, typename R>
    typename std::enable_if<!std::is_arithmetic<R>::value>::type registerCall(C *, const char *, R (C::*)(void) const) {}
    void registerSignal(const std::string &name) { signalNames.push_back(name); }
    template <typename T>
    void emitSignal(const std::string &name, const T &v)
    {
        SignalRecord r; r.name = name; r.value = double(v);
        signals.push_back(r);
    }

    void setupInput(const int i, const DType & = DType()) { inputs[std::to_string(i)]; }
    void setupInput(const std::string &n, const DType & = DType()) { inputs[n]; }
    void setupOutput(const int i, const DType & = DType()) { outputs[std::to_string(i)]; }
    void setupOutput(const std::string &n, const DType & = DType()) { outputs[n]; }
    void setupOutput(const char *n, const DType &d = DType()) { setupOutput(std::string(n), d); }

    InputPort *input(const int i) { return &inputs.at(std::to_string(i)); }
    InputPort *input(const std::string &n) { return &inputs.at(n); }
    OutputPort *output(const int i) { return &outputs.at(std::to_string(i)); }
    OutputPort *output(const std::string &n) { return &outputs.at(n); }
    OutputPort *output(const char *n) { return &outputs.at(std::string(n)); }

    virtual BufferManager::Sptr getInputBufferManager(const std::string &, const std::string &) { return BufferManager::Sptr(); }
    virtual BufferManager::Sptr getOutputBufferManager(const std::string &, const std::string &) { return BufferManager::Sptr(); }

    //recording state, read by the driver
    std::map<std::string, InputPort> inputs;
    std::map<std::string, OutputPort> outputs;
    std::map<std::string, std::function<void(double)>> calls;
    std::map<std::string, std::function<void(const std::string &)>> stringCalls;
    std::map<std::string, std::function<double(void)>> getters;
    std::vector<std::string> signalNames;
    std::vector<SignalRecord> signals;
};

//! registry fake: remembers factories of signature Block*(size_t) by path
class BlockRegistry
{
public:
    typedef Block *(*FactorySizeT)(const size_t);
    static std::map<std::string, FactorySizeT> &table(void)
    {
        static std::map<std::string, FactorySizeT> t;
        return t;
    }
    BlockRegistry(const std::string &path, FactorySizeT f) { table()[path] = f; }
    typedef Block *(*FactoryVoid)(void);
    static std::map<std::string, FactoryVoid> &tableVoid(void)
    {
        static std::map<std::string, FactoryVoid> t;
        return t;
    }
    BlockRegistry(const std::string &path, FactoryVoid f) { tableVoid()[path] = f; }
    //! two size_t parameters (a multi-channel block: sf, channels)
    typedef Block *(*FactorySizeT2)(const size_t, const size_t);
    static std::map<std::string, FactorySizeT2> &table2(void)
    {
        static std::map<std::string, FactorySizeT2> t;
        return t;
    }
    BlockRegistry(const std::string &path, FactorySizeT2 f) { table2()[path] = f; }
};

} // namespace Pothos

#define POTHOS_FCN_TUPLE(classPath, functionName) \
    #functionName, &classPath::functionName
