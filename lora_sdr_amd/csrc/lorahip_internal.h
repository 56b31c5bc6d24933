// Internal declarations shared by the host-side translation units of liblorahip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <complex>
#include <string>
#include <vector>
#include "../../include/lorahip.h"

namespace lorahip {

typedef std::complex<float> cf32;

//! host tables, built with the reference's own expressions (lorahip_tables.cpp)
struct HostTables
{
    std::vector<cf32> up, down, fine, twiddle;
};
void buildHostTables(int sf, HostTables &t, bool wantFine);
//! the fp64 factor tables of lorahip_fine.h (interleaved re, im) for the given fine-tune table; false (and empty tables) unless
//! EVERY entry of `fine` is reproduced bit for bit by the device's evaluation order (mul, fma, convert)
bool buildFineSplit(int sf, const std::vector<cf32> &fine, std::vector<double> &A, std::vector<double> &B);

//! kernel argument block (device pointers), one launch = nWindows independent windows
struct DetectArgs
{
    const float2 *iq;
    const long long *offsets;   // nullable
    long long stride;           // samples between windows when offsets == nullptr
    const int *chirpSel;        // nullable
    int chirpSelAll;
    const int *fineIdx0;        // nullable
    const float *fineErr;       // nullable
    unsigned short *sym;
    float *power;
    float *powerAvg;
    float *fIndex;
    int *fineIdxOut;            // nullable
    float2 *fftOut;             // nullable
    float2 *decOut;             // nullable
    const float2 *up;
    const float2 *down;
    const float2 *fine;
    const float2 *tw;
    const double2 *fineA;       // split of the fine-tune table (lorahip_fine.h), fp64; nullptr: gather from `fine`
    const double2 *fineB;
    unsigned nWindows;
    float powerScale;           // float(20*log10(double(N)))  LoRaDetector.hpp:18
};

//! tables of the tuned kernels (device pointers, built at context creation)
struct FastTables
{
    const float2 *twStage;      // stage-major twiddles: for each radix-4 stage with remainder m = 2^b,
                                // [q-1][k] = twiddle(k * fstride_b * q), k < m  (values of kissfft's table)
    int nBlocksHint;            // multiprocessor count of the device
};
//! host: stage-major twiddle table from kissfft's table (lorahip_fast.hip)
std::vector<cf32> buildStageTwiddles(int sf, const std::vector<cf32> &tw);

//! per-channel state of the LoRaDemod block as the streaming kernel keeps it (LoRaDemod.cpp:385-392)
struct StreamState
{
    int state;              // 0 FRAMESYNC .. 4 DATASYMBOLS
    int downTable;          // _chirpTable == _downChirpTable
    int prevValue;          // short _prevValue
    int freqError;
    int fineTuneIndex;
    float finefreqError;
    int symCount;
    int callCount;          // work() calls made since the run started (= the lock-step round index of the next call)
    long long pos;          // samples consumed so far
};

/* Decisions that the last-ulp differences between this library's power / powerAvg / fIndex and the CPU build's could flip
 * (include/lorahip.h, lorahip_demod_near_threshold). They are COUNTED, never changed:
 *   squelch: `snr < thresh` (LoRaDemod.cpp:174) consumed by FRAMESYNC / DATASYMBOLS with |snr - thresh| <= 4e-5 dB (twice the
 *            tolerance the tests hold power and powerAvg to);
 *   step:    a window dechirped with a moving index whose step d = _finefreqError * 128 (:160) lies within 6e-5 of an integer,
 *            where ceil(d) / floor(d) -- the closed form's increment, i.e. where the reference's truncation falls -- would change
 *            if the fIndex values accumulated in _finefreqError (:219) differed in their last place (1 ulp of 0.5 bins * 128
 *            steps * up to 16 accumulations). */
#define LORAHIP_NEAR_SNR_DB 4e-5f
#define LORAHIP_NEAR_STEP 6e-5f
__host__ __device__ inline bool nearSquelch(const float snr, const float thresh)
{
    const float m = snr - thresh;
    return (m < 0.0f ? -m : m) <= LORAHIP_NEAR_SNR_DB;
}
__host__ __device__ inline bool nearStep(const float d)
{
    if (d == 0.0f || !(d == d)) return false;
    const float a = d < 0.0f ? -d : d;
    if (a >= 8388608.0f) return true;                      // every float this large is an integer
    const float r = float(int(a + 0.5f));
    const float m = a - r;
    return (m < 0.0f ? -m : m) <= LORAHIP_NEAR_STEP;
}

//! one posted packet: the call (round) it was posted in and its length; its symbols are the next `len` entries
//! of the channel's symbol stream (after what earlier launches carried over)
struct StreamPacket { int callIndex; int len; };
//! what the block emits on its "error" / "power" / "snr" signals at DOWNCHIRP1 (LoRaDemod.cpp:267-269), one record per emission:
//! kept only when the caller asked for signals (lorahip_demod_set_signals) -- a receiver without a per-call trace still gets them
struct StreamSignal { int callIndex; int error; float power; float snr; };

/*! The RESIDENT receiver (lorahip_demod_receive with async = 3; lorahip_streamkernel.h, RES): one launch stays on the device across the
 * receiver's steps. The host describes a step in a ResidentMsg, written with plain stores into a ring in PINNED HOST memory (the
 * "doorbell": no API call, and nothing that needs a place on the device -- a copy that the runtime performs with a blit kernel would
 * wait for ever behind a kernel that fills every SIMD). The wavefronts poll a mirror of the ring in DEVICE memory; a few of them at a
 * time also look at the host's ring over PCIe, and whoever finds a new message first copies it into the mirror for all. They work
 * through what arrived, the last wavefront of every workgroup packs the workgroup's packets / signals into the step's rows, and the
 * last workgroup of the step reports two words to pinned host memory. No launch, no helper kernel, no API call per step. */
struct ResidentMsg
{
    unsigned long long nValid;          // samples per channel that are valid now
    unsigned short *syms; int *nsyms; int *chan;            // this step's packet rows (lorahip_packet_rows)
    int *sigCh; int *sigErr; float *sigPow; float *sigSnr;  // ... and signal rows (null: signals are not kept)
    unsigned symStride, capRows, capSig;
    unsigned flags;                     // bit 0: leave the kernel (no samples with this message)
    unsigned check;                     // residentCheck() of the fields above and seq: a torn copy is not taken for a message
    unsigned seq;                       // the step's number (1, 2, ...): written last in the struct
};
__host__ __device__ inline unsigned residentCheck(const ResidentMsg &m)
{
    unsigned long long h = m.nValid ^ (unsigned long long)(size_t)m.syms ^ ((unsigned long long)(size_t)m.nsyms << 1) ^ ((unsigned long long)(size_t)m.chan << 2) ^
                           ((unsigned long long)(size_t)m.sigCh << 3) ^ ((unsigned long long)(size_t)m.sigErr << 4) ^ ((unsigned long long)(size_t)m.sigPow << 5) ^
                           ((unsigned long long)(size_t)m.sigSnr << 6);
    h ^= (unsigned long long)m.symStride * 0x9E3779B97F4A7C15ull + m.capRows * 0xC2B2AE3D27D4EB4Full + m.capSig * 0x165667B19E3779F9ull + m.flags;
    return unsigned(h ^ (h >> 32)) ^ (m.seq * 0x85EBCA6Bu) ^ 0x5A5A5A5Au;
}
//! device memory shared by the workgroups of a resident launch; the per-step counters exist eight times (step & 7): the host never has
//! more than RES_DEPTH_MAX + 1 = 4 steps outstanding, and the last workgroup of step k clears the set of step k + 4
enum { RES_DEPTH_MAX = 3, RES_RING = 4 };       // steps the host may ring ahead of the last report; LDS / record-array sets per workgroup
struct ResidentCtl
{
    ResidentMsg msg[16][8];             // the ring's mirrors, one per group of workgroups (blockIdx & 15), slot = seq & 7: a thousand
                                        // wavefronts polling ONE line of memory at system scope queue up behind each other for hundreds of
                                        // microseconds (measured: profiles/r06); sixteen lines, one poller per workgroup at a time, do not
    unsigned long long doneCalls[8][16];    // [k][0] (a 128-byte line per step slot): [63:48] workgroups that finished the step, [47:36] of them with a stopped
                                        // channel, [35:0] work() calls
    unsigned long long rowSig[8][16];   // [k][0]: [31:0] packet rows, [63:32] signal rows handed out in the step (may exceed the capacity: the excess was
                                        // dropped and is reported) -- ONE atomic per wavefront and channel set that has either
    unsigned abortDev;                  // the host's abort flag, relayed (only the relay wavefronts read host memory)
    unsigned arrived;                   // workgroups that have started (the census: all of them must be on the device at once)
    unsigned expired;                   // a wavefront gave up waiting for a message (watchdog)
    unsigned long long dbg[8][6];       // LORAHIP_RESIDENT_DEBUG: workgroup 0, wavefront 0, the LAST eight steps (slot = (step - 1) & 7): 100 MHz ticks at wait start, message seen,
                                        // pass loop entered, pass loop left, records carried out, step end
    unsigned long long dbgWave[16384][4];   // ... and EVERY wavefront's stamps of ONE step (LORAHIP_RESIDENT_DEBUG = its number; plain stores, the other steps
                                        // are not disturbed): wait start, message seen, windows done, step end
};
//! pinned host memory the device addresses directly: the host's side of the doorbell and the kernel's reports
struct ResidentHost
{
    ResidentMsg msg[8];                 // the ring the host writes (seq last)
    unsigned long long sum[16];         // [8][2] the steps' reports (below)
    unsigned abort;                     // host: leave now
    unsigned arrivedAll;                // kernel: every workgroup has started
};
//! what the last workgroup of a step writes to pinned host memory: two 64-bit words, each carrying (part of) the step number
//!   w0 = seq << 32 | calls (32 bits)      w1 = (seq & 0xff) << 56 | flags << 48 | signals (24 bits) << 24 | packets (24 bits)
//!   flags: 1 packets dropped (rows too small), 2 signals dropped, 4 more, 8 expired
enum { RES_F_PKT_OVERFLOW = 1, RES_F_SIG_OVERFLOW = 2, RES_F_MORE = 4, RES_F_EXPIRED = 8 };

//! argument block of the streaming demod kernel (lorahip_stream.hip); all pointers are device pointers
struct StreamArgs
{
    const float2 *iq;
    const long long *base;      // [nChannels] first sample of the channel's stream in iq
    const long long *len;       // [nChannels] samples available
    StreamState *state;         // [nChannels] in/out
    lorahip_work_result *calls; // [nChannels][cap] one record per work() call; nullptr unless tracing
    int *nCalls;                // [nChannels] work() calls made by this launch
    short *symOut;              // [nChannels][symStride] the DATASYMBOLS values of this launch, in order; with flag bit 2 the symbols of the
                                // packet the channel was inside when the launch began come first (copied there from `carry`)
    int *nSym;                  // [nChannels] entries of the channel's symOut row that are valid after the launch (carried ones included)
    StreamPacket *pktOut;       // [nChannels][capPkt]
    int *nPkt;                  // [nChannels]
    int capPkt;
    StreamSignal *sigOut;       // [nChannels][capPkt] one record per DOWNCHIRP1 call; nullptr unless signals are kept
    int *nSig;                  // [nChannels]
    int2 *end;                  // [nChannels] where the channel stands after the launch, as the summary needs it: x = symCount of the packet it is
                                // inside (-1: in none), y = callCount -- beside the counts, so that streamSummary reads five dense arrays and
                                // not three words of every 40-byte state
    const float2 *down, *fine, *twStage;
    const double2 *fineA, *fineB;   // split of the fine-tune table (lorahip_fine.h); nullptr: gather from `fine`
    unsigned nChannels;
    int cap;                    // work() calls per channel this launch may make
    int symStride;              // entries per channel in symOut: cap + the longest carried packet
    float powerScale;
    float thresh;
    int sync;
    unsigned mtu;
    long long uniformLen;       // >= 0: channel c's stream is the uniformLen samples at c * uniformStride (base / len are not read)
    long long uniformStride;    // samples between the first samples of consecutive channels' streams (uniformLen >= 0)
    int flags;                  // bit 0: first launch of a run -- every channel starts at sample 0, call 0; bit 1: activate() first;
                                // bit 2: a channel in DATASYMBOLS first copies its open packet's symCount symbols from `carry` to the head of
                                //        its symOut row and appends behind them; bit 3: a channel that ends the launch in DATASYMBOLS leaves
                                //        the last symCount entries of its row in `carry` (the open packets stay on the device, no extra launch)
    short *carry;               // [nChannels][carryCap] symbols of the packets the channels are inside, between launches (bits 2 / 3)
    int carryCap;
    int lanes;                  // log2 of the lanes per channel (SF7-9: lorahip_stream_lanes.hip): 0 = chosen by the channel count, < 0 = the
                                // 16-points-per-lane geometry always, else that instance if the build holds it
    int maxBlocks;              // the grid: 0 = the kernel's default, < 0 = one workgroup per channel set always, > 0 = at most this many workgroups, each looping over channel sets
    unsigned lastRoundFrom;     // set by the launcher: workgroups from this blockIdx on are not followed by another one in their slot (lorahip_device.h::rotatePriority)
    unsigned *near;             // [2] decisions float rounding could flip (lorahip_demod_near_threshold): squelch margins, fine-tune steps.
                                //     Running counters: the kernels only add, the host takes differences
    // the resident receiver (RES instances only)
    ResidentCtl *res = nullptr;
    ResidentHost *resHost = nullptr;           // pinned host memory as the device addresses it
    unsigned long long resWatchdog = 0;        // 100 MHz ticks a wavefront waits for a message before it gives up
    unsigned long long resRecStride = 0;       // bytes between the RES_RING sets of record arrays (symOut / pktOut / sigOut): a step writes set
                                               // step & 3, so that a wavefront already in a later step does not write into the rows the
                                               // workgroup's last wavefront of step k is still packing from (found by the soak: profiles/r06/s15_*)
    unsigned resDebug = 0;                     // LORAHIP_RESIDENT_DEBUG: the step whose stamps every wavefront leaves in ResidentCtl::dbgWave
    int resSleep = 8;                          // a waiting wavefront's nap between looks, in units of s_sleep 8 (512 clocks); 0: it spins
};

//! what the host needs to know after a streaming launch -- reduced on the device (streamSummary), so that 72 bytes cross PCIe per run
//! instead of the per-channel state and counts (52 B per channel), which are fetched only when somebody asks for them
struct StreamSummary
{
    long long calls, packets, syms, signals;    // sums over the channels of this launch's counts
    long long openSyms;                         // sum of symCount over the channels that are inside a packet now
    int more;                                   // some channel filled a record buffer: the launch must be resumed
    int anyOpen;                                // some channel is in DATASYMBOLS
    int maxCallCount;                           // largest callCount (= lock-step rounds so far)
    int maxOpen;                                // largest symCount among the open packets
    int fullest;                                // most calls any channel made in this launch
    unsigned nearSquelch, nearStep;             // the running counters (StreamArgs::near)
    int pad;
};
//! scratch: streamSummaryScratchBytes(nChannels) bytes of device memory (beyond 32768 channels one workgroup per 4096 channels leaves a
//! record there and a second small launch adds them up; up to that, or with nullptr, one workgroup walks every channel)
hipError_t launchStreamSummary(const int2 *end, const int *nCalls, const int *nSym, const int *nPkt, const int *nSig, size_t nChannels,
                               int cap, int capPkt, const unsigned *near, void *scratch, StreamSummary *out, hipStream_t stream);
size_t streamSummaryScratchBytes(size_t nChannels);

//! argument block of the batched decoder (lorahip_codec.hip); device pointers
struct DecodeArgs
{
    const unsigned short *syms;     // [nPackets][symStride]
    const int *nsyms;               // [nPackets]
    unsigned char *out;             // [nPackets][outStride]
    int *outLen;                    // [nPackets] elements posted, -1 nothing posted, -2 packet too long for this build
    int *dropped;                   // [nPackets] the block called drop()
    unsigned nPackets;
    int symStride, outStride;
    int sf, ppm, rdd, crcc, interleaving, errorCheck, explicitHdr, hdr, dataLength;
};
hipError_t launchDecode(const DecodeArgs &a, int variant, hipStream_t stream);   // variant 1: the lane-per-packet checker
int decodeMaxSymbols();
int decodeMaxDataLength();

//! launchers (lorahip_kernels.hip / lorahip_fast.hip)
hipError_t launchDetect(int sf, int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream);
bool fastAvailable(int sf);
hipError_t launchFast(int sf, int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream);
bool wideAvailable(int sf);
bool wideLayoutsOk();
bool fastLayoutsOk();
hipError_t launchWide(int sf, int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream);
bool streamAvailable(int sf);
hipError_t launchStream(int sf, const StreamArgs &s, hipStream_t stream);
hipError_t launchStreamWide(int sf, const StreamArgs &s, hipStream_t stream);
//! the resident receiver's launch (SF7-10, the 16-points-per-lane geometries): hipErrorNotSupported when there is no such instance or
//! the grid would not be resident all at once; *grid = workgroups launched
hipError_t launchStreamResident(int sf, const StreamArgs &s, hipStream_t stream, unsigned *grid);
hipError_t launchStreamResidentWide(int sf, const StreamArgs &s, hipStream_t stream, unsigned *grid);    // SF11 / SF12 (lorahip_wide.hip)
bool streamLanesAvailable(int sf, int log2Lanes);
int streamLanesChosen(int sf, unsigned nChannels, int forced, unsigned otherWaves = 0);
//! wavefronts the other parts of a mixed object put on this part's device at 16 points per lane (lorahip_rx.cpp -> lorahip_demod.cpp)
void demodSetCoResidentWaves(lorahip_demod *dm, unsigned waves);
hipError_t launchStreamLanes(int sf, int log2Lanes, const StreamArgs &s, hipStream_t stream);
//! a lanes code with this bit: the AHEAD instance over windows of 2^(code & 15) lanes (lorahip_stream_pairs.hip) -- a channel takes two such
//! lane groups, the second evaluates the window the NEXT call reads if this one is plain
constexpr int LORAHIP_LANES_AHEAD = 16;
bool streamPairsAvailable(int sf, int log2WindowLanes);
hipError_t launchStreamPairs(int sf, int log2WindowLanes, const StreamArgs &s, hipStream_t stream);
hipError_t launchCompactRows(void *dst, const void *src, size_t rows, size_t srcPitchBytes, size_t rowBytes, hipStream_t stream);
hipError_t launchPackSignals(const StreamSignal *sigOut, const int *nSig, size_t nChannels, int capPkt, int *channel, int *error, float *power, float *snr,
                             size_t firstRow, size_t capRows, hipStream_t stream);
hipError_t launchPackPackets(const StreamPacket *pktOut, const int *nPkt, const short *symOut, int *rowStart, size_t nChannels, int cap, int capPkt,
                             size_t nPackets, long long *srcOff, unsigned short *symsOut, int stride, int *nsymsOut, int *channelOut, hipStream_t stream);
hipError_t launchCopySegments(float2 *dst, const float2 *src, const long long *srcOff, const long long *dstOff, const int *len, size_t nSeg,
                              hipStream_t stream);
hipError_t launchSynth(int sf, float2 *iq, const unsigned short *sym, size_t nWindows,
                       float ampl, float sigma, unsigned long long seed, hipStream_t stream);

hipError_t launchModFrames(float2 *iq, long long frameStride, const unsigned short *syms, size_t nFrames, int nsyms, int sync,
                           float ampl, int padding, int sf, hipStream_t stream);
hipError_t launchAwgn(float2 *iq, size_t n, float sigma, unsigned long long seed, hipStream_t stream);
hipError_t launchMembw(const float2 *iq, size_t nBytes, int pattern, int blocks, float *scratch, hipStream_t stream);

//! hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): function attributes are per device, and one
//! process may drive several contexts on several devices from several threads
hipError_t ensureDynamicLds(const void *kernel, size_t bytes, unsigned long long &doneMask);
//! StreamArgs::lastRoundFrom of a grid of `grid` workgroups on a device that holds `resident` of them. Workgroups of the last resident
//! set are not replaced when they finish: they share the SIMD's priority in turns (lorahip_device.h::rotatePriority) and finish
//! together; the ones before them keep a priority above, oldest first, so that one of two co-resident workgroups finishes early and
//! its slot goes to the next one. A grid of WHOLE resident sets rotates from the first workgroup on: every set then ends at once and
//! nothing runs alone (profiles/r04/s27_*: 2.0 sets +6-9 % over no priorities at all against +3-5 % with the split rule; 1.5 / 2.5
//! sets -8 % against +1-2 %).
inline unsigned lastRoundFrom(const unsigned grid, const int resident)
{
    if (resident <= 0 || grid <= unsigned(resident) || grid % unsigned(resident) == 0) return 0u;
    return grid - unsigned(resident);
}
//! how many workgroups of `kernel` the current device holds at once (occupancy x compute units); 0 if the runtime will not say
int residentWorkgroups(const void *kernel, int threads, size_t smem);
//! the same, asked once per (kernel, device) and kept: one value per device, because the per-device host threads of a mixed object
//! launch the same kernel on different devices at the same time (a single static would be one device's answer for all of them)
struct PerDeviceCount
{
    int v[64];
    PerDeviceCount(void) { for (int i = 0; i < 64; i++) v[i] = -1; }
};
int residentWorkgroupsCached(PerDeviceCount &cache, const void *kernel, int threads, size_t smem);

//! makes a context's device current for the duration of an entry point and restores the caller's afterwards (a process
//! may hold contexts on several devices; torch keeps its own notion of the current device)
struct DeviceGuard
{
    int prev;
    bool switched;
    explicit DeviceGuard(const int device) : prev(-1), switched(false)
    {
        if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipSetDevice(device); }       // nothing to restore
        else if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

void setLastError(const std::string &s);
int hipFail(hipError_t e, const char *what);

#define LORAHIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return ::lorahip::hipFail(_e, #expr); } while (0)

} // namespace lorahip

namespace lorahip {
//! two pinned staging buffers of the host -> device gather (lorahip_upload.cpp)
struct Uploader { void *buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool busy[2] = {false, false}; bool ready = false; void *pool = nullptr; /* CopyPool: lorahip_upload.cpp */ };
}

struct lorahip_ctx
{
    int device;
    int sf;
    size_t N;
    int variant;
    hipStream_t ownStream;
    hipStream_t stream;
    float2 *dUp, *dDown, *dFine, *dTw, *dTwStage;
    double2 *dFineA, *dFineB;   // nullptr when the split did not verify on this host (kernels gather then)
    int fineGather;             // A/B switch (lorahip_set_fine_gather): read the fine-tune table itself even though the split verified
    int cuCount;
    hipEvent_t ev0, ev1;
    float powerScale;
    // staging for the host-pointer entry point (grown on demand)
    void *dStage; size_t dStageBytes;
    void *hStage; size_t hStageBytes;
    lorahip::Uploader up;
};

namespace lorahip {
//! A level-3 object over several (device, SF) parts behind one lorahip_demod handle (lorahip_rx.cpp; lorahip_demod_create_mixed).
//! The entry points of lorahip_demod.cpp hand over to it when the handle carries one; it speaks global channel numbers.
class Composite
{
public:
    static int create(Composite **out, const int *devices, size_t nDev, const int32_t *channelSf, size_t n);
    ~Composite();
    size_t numChannels() const;
    size_t numParts() const;
    int partInfo(size_t i, int32_t *device, int32_t *sf, size_t *nChannels, int32_t *deviceSlot) const;
    int partOf(int32_t *part, int32_t *local) const;
    lorahip_demod *part(size_t i) const;
    int setSync(unsigned char v); int setThreshold(double v); int setMtu(size_t v); int setMode(int v); int setFineGather(int v);
    int setTrace(int v); int setSignals(int v); int activate(); int setStream(void *stream); int resetStream(); int streamWait(void *stream); int streamFollow(void *stream);
    int run(const float *const *streams, const size_t *nSamples, int64_t *rounds);
    int runSegments(const float *const *iqPerDevice, size_t nDev, const int64_t *first, const size_t *nSamples, int64_t *rounds);
    size_t numPackets() const; size_t numPacketSymbols() const; size_t numSignals() const;
    int getPacket(size_t i, int32_t *channel, int64_t *round, size_t *len, int16_t *out, size_t cap) const;
    int getPackets(int32_t *channels, int64_t *rounds, int64_t *lens, size_t capPackets, int16_t *syms, size_t capSyms) const;
    int getSignals(int32_t *channels, int64_t *rounds, int32_t *errors, float *powers, float *snrs, size_t cap) const;
    void clearPackets();
    int64_t consumed(size_t c) const; int consumedAll(int64_t *out) const;
    int64_t workCalls() const; double kernelMs() const; int lastLaunches() const; int nearThreshold(int64_t *sq, int64_t *st) const;
    int setPorts(const lorahip_demod_ports *ports);
    int portCounts(size_t c, size_t *f, size_t *d, size_t *r) const;
    int getLabels(size_t c, char *buf, size_t cap, size_t *n, size_t *bytes) const;
    int getTrace(size_t c, lorahip_work_result *out, size_t cap) const;
    size_t traceLen(size_t c) const;
private:
    Composite();
    struct Impl;
    Impl *p;
};

//! n host pieces -> device memory back to back from dDst, asynchronous on ctx->stream (pinned pieces by DMA straight away, the
//! others through the context's double-buffered pinned staging); the caller synchronises the stream
int gatherUpload(lorahip_ctx *ctx, void *dDst, const void *const *src, const size_t *bytes, size_t n);
void destroyUploader(lorahip_ctx *ctx);
}
