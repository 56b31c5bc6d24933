// Level 3 of the C ABI: B channels of the LoRaDemod block driven in lock-step.
//
// Mirrors LoRaDemod.cpp: parameters and defaults (:68-74), setters (:124-137), activate()
// (:139-143) and work() (:145-327). One work() "round" = every channel that has >= 2N
// samples left performs one work() call. The dechirp+FFT+detect of all those calls is ONE
// batch launch (lorahip_detect_batch); the frame state machine -- a handful of integer
// operations per call whose only coupling is the per-channel window offset -- runs on the
// host between launches. Channels that hit "syncd and match0" (:189) get their second
// window in a second, compacted launch, exactly like the reference's nested loop (:189-206).
//
// The three members the reference leaves uninitialised (_finefreqError, _freqError,
// _prevValue) start at zero here.
#include "lorahip_internal.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace lorahip;

namespace {

enum State { ST_FRAMESYNC = 0, ST_DOWNCHIRP0, ST_DOWNCHIRP1, ST_QUARTERCHIRP, ST_DATASYMBOLS };

struct Channel
{
    int state;
    bool downTable;         // _chirpTable == _downChirpTable
    short prevValue;
    int freqError;
    int fineTuneIndex;
    float finefreqError;
    size_t symCount;
    std::vector<int16_t> outSymbols;
    size_t base, len, pos;  // stream placement in the device buffer, read position
    int64_t callCount;      // work() calls since the stream began (a run that does not continue a stream starts it at 0)
    size_t runStart;        // read position when the last run began (append runs continue; the port replay starts there)
    std::vector<lorahip_work_result> trace;
    size_t traceStart;      // first trace entry of the last run
    size_t traceSymCount0;  // _symCount when the trace began (labels "S<n>" continue a packet that was open then)
    size_t portFft, portDec, portRaw;   // frames / samples the last run produced on the debug ports
};

//! one posted packet: its symbols are pktSyms[off, off+len) of the owning demod (flat storage: tens of thousands of
//! packets per run must not cost an allocation each)
struct Packet
{
    int32_t channel;
    int64_t round;
    size_t off, len;
};

//! one emission of the block's "error" / "power" / "snr" signals (LoRaDemod.cpp:267-269)
struct Signal
{
    int32_t channel;
    int64_t round;
    int32_t error;
    float power, snr;
};

} // namespace

struct lorahip_demod
{
    int streamGrid;                  // lorahip_demod_set_stream_grid: 0 default, < 0 one workgroup per channel set, > 0 at most that many workgroups
    size_t streamCapMax;             // lorahip_demod_set_record_capacity: 0 = no bound beyond the library's own
    int streamLanes;                 // lorahip_demod_set_stream_lanes: 0 by channel count, < 0 always 16 points per lane, else log2 of the lanes per channel
    unsigned coWaves;                // a part of a mixed object: the wavefronts its sibling parts put on the same device (the automatic choice counts them)
    lorahip::Composite *comp;        // non-null: the handle is a container of (device, SF) parts (lorahip_rx.cpp); nothing below is used then
    lorahip_ctx *ctx;
    size_t N, B;
    unsigned char sync;
    float thresh;
    size_t mtu;
    bool tracing;
    int64_t workCalls;
    std::vector<Channel> ch;
    std::vector<Packet> packets;
    std::vector<int16_t> pktSyms;
    bool wantSignals;                // keep a record per DOWNCHIRP1 call (lorahip_demod_set_signals)
    std::vector<Signal> signals;
    lorahip_signal_rows sigRows;     // lorahip_demod_receive_signal_rows: where receiver steps deliver the signals (all pointers null: they are dropped)
    bool sigRowsOn;
    size_t lastSignals;              // ... how many the last receive / receive_flush delivered there
    // per-round staging (host pinned + device), sized for B windows
    char *h, *d;
    size_t stageBytes;
    float *dIq; size_t dIqSamples;   // owned upload buffer for lorahip_demod_run
    int mode;                        // 0 auto, 1 device streaming kernel, 2 host-driven rounds
    // streaming path: device + pinned-host mirrors, grown on demand
    char *sDev, *sHost; size_t sBytes;
    char *dDense, *hDense; size_t denseBytes;   // the used part of the record arrays, packed for the copy back
    hipEvent_t evK0, evK1;           // around the streaming kernel launches of a run (lorahip_demod_kernel_ms)
    hipEvent_t evJoin;               // lorahip_demod_stream_wait
    void *dSumScratch;               // streamSummary's partial records (more than 32768 channels)
    double kernelMs;
    int lastLaunches;                // streaming kernel launches of the last run
    lorahip_demod_ports ports;       // level-3 debug ports (all pointers null: off); DEVICE pointers (the library's own when the caller's are host buffers)
    lorahip_demod_ports hostPorts;   // the caller's host buffers (host_buffers == 1)
    float *ownFft, *ownDec, *ownRaw; // device mirrors owned by the library for host_buffers
    bool portsOn, userTracing;
    char *dPort; size_t dPortBytes;  // scratch of the port replay: window descriptors, replayed fft / dec windows
    int64_t nNearSquelch, nNearStep; // decisions float rounding could flip, since activate() (lorahip_demod_near_threshold)
    // Streaming mode keeps the per-channel frame-machine state ON THE DEVICE between runs (its pinned copy in sHost is what the host
    // reads); the Channel mirrors above are brought up to date only when somebody needs them (host-driven rounds, ports, accessors)
    bool devStateFresh;              // the device holds the current state: the next streaming run need not upload it
    bool mirrorsStale;               // ch[].state .. ch[].pos lag behind the pinned copy of the device state
    bool activatePending;            // activate() since the last run, not yet applied to the device state / the mirrors
    bool uniform; size_t uniSpc;     // the streams of the current run are n_channels x uniSpc samples, uniStride apart (lorahip_demod_run_device[_append])
    size_t uniStride;
    bool append;                     // the current run continues every channel's stream where the last append run left it
    bool appendFresh;                // ... unless nothing has been appended yet (create, rewind, any other kind of run in between)
    size_t appendPrev;               // samples per channel the last append run was given
    bool headStale;                  // the pinned copy of the per-channel state and counts (sHost) lags the device: ensureHead() fetches it
    StreamSummary lastSum;           // of the last streaming launch (valid while devStateFresh)
    unsigned nearSeen[2];            // the kernels' running near-threshold counters as last read
    bool geomApplied;                // ch[].base / len / pos hold the current run's placement
    bool posOnDevice;                // the pinned state copy's `pos` belongs to the CURRENT placement (a streaming run filled it; a new
                                     // lorahip_demod_run[_device] call invalidates it: its streams start at sample 0)
    bool portCountsDirty;            // ch[].portFft / portDec / portRaw may be non-zero
    // The symbols of the packet a channel is INSIDE when a streaming run ends stay on the device too: the kernel leaves them in dCarry
    // and copies them to the head of the channel's symbol row at the start of the next run, then appends behind them -- a packet
    // that spans runs is assembled without the host (the running receiver: lorahip_demod_run_device_segments + packets_to_device).
    short *dCarry; size_t carryCap;  // [B][carryCap]
    bool devCarryValid;              // dCarry holds the open packets of the state on the device
    bool hostCarryStale;             // ch[].outSymbols lack what the runs since the last drain received (implies devCarryValid)
    size_t callsPerWindowQ8;         // streaming runs: work() calls per N samples the record buffers are sized for, in 1/256 (adapts, see runStream)
    void *pipe;                      // Pipe: the two record sets and events of the pipelined receiver (lorahip_demod_receive, async = 2)
    void *pending;                   // PendingLaunch (records of the last streaming launch still on the device)
    std::vector<size_t> carry;       // per channel: symbols of a packet begun before the launch being drained
};

namespace {

struct Round
{
    int64_t *off; int32_t *sel; int32_t *idx0; float *err;                  // inputs
    uint16_t *sym; float *power; float *pavg; float *fidx; int32_t *idxOut; // outputs
    size_t inBytes, total;
};

static size_t align256(const size_t x) { return (x + 255) & ~size_t(255); }

static Round carve(char *p, const size_t B)
{
    // (offsets first, pointers last: the size of the block is asked for with p == nullptr, and arithmetic on a null pointer is
    // undefined -- found by the -fsanitize=undefined pass, profiles/r05)
    Round r;
    size_t cur = 0;
    const size_t oOff = cur; cur += align256(B * sizeof(int64_t));
    const size_t oSel = cur; cur += align256(B * sizeof(int32_t));
    const size_t oIdx0 = cur; cur += align256(B * sizeof(int32_t));
    const size_t oErr = cur; cur += align256(B * sizeof(float));
    r.inBytes = cur;
    const size_t oSym = cur; cur += align256(B * sizeof(uint16_t));
    const size_t oPower = cur; cur += align256(B * sizeof(float));
    const size_t oPavg = cur; cur += align256(B * sizeof(float));
    const size_t oFidx = cur; cur += align256(B * sizeof(float));
    const size_t oIdxOut = cur; cur += align256(B * sizeof(int32_t));
    r.total = cur;
    r.off = nullptr; r.sel = nullptr; r.idx0 = nullptr; r.err = nullptr;
    r.sym = nullptr; r.power = nullptr; r.pavg = nullptr; r.fidx = nullptr; r.idxOut = nullptr;
    if (p == nullptr) return r;
    r.off = reinterpret_cast<int64_t *>(p + oOff); r.sel = reinterpret_cast<int32_t *>(p + oSel);
    r.idx0 = reinterpret_cast<int32_t *>(p + oIdx0); r.err = reinterpret_cast<float *>(p + oErr);
    r.sym = reinterpret_cast<uint16_t *>(p + oSym); r.power = reinterpret_cast<float *>(p + oPower);
    r.pavg = reinterpret_cast<float *>(p + oPavg); r.fidx = reinterpret_cast<float *>(p + oFidx);
    r.idxOut = reinterpret_cast<int32_t *>(p + oIdxOut);
    return r;
}

//! one compacted launch over `n` windows described in the host staging block
static int launchRound(lorahip_demod *dm, const float *iqDev, const size_t n)
{
    lorahip_ctx *ctx = dm->ctx;
    const Round hr = carve(dm->h, dm->B), dr = carve(dm->d, dm->B);
    LORAHIP_TRY(hipMemcpyAsync(dm->d, dm->h, hr.inBytes, hipMemcpyHostToDevice, ctx->stream));
    lorahip_batch b;
    std::memset(&b, 0, sizeof(b));
    b.struct_size = sizeof(b);
    b.iq = iqDev;
    b.n_windows = n;
    b.offsets = dr.off;
    b.chirp_sel = dr.sel;
    b.fine_idx0 = dr.idx0;
    b.fine_err = dr.err;
    b.sym = dr.sym; b.power = dr.power; b.power_avg = dr.pavg; b.f_index = dr.fidx;
    b.fine_idx_out = dr.idxOut;
    const int rc = lorahip_detect_batch(ctx, &b);
    if (rc != LORAHIP_OK) return rc;
    LORAHIP_TRY(hipMemcpyAsync(dm->h + hr.inBytes, dm->d + hr.inBytes, hr.total - hr.inBytes,
                               hipMemcpyDeviceToHost, ctx->stream));
    LORAHIP_TRY(hipStreamSynchronize(ctx->stream));
    return LORAHIP_OK;
}

//! ch[].base / len / pos of a uniform run (filled only for the paths that read them: host-driven rounds, the port replay)
static void applyGeometry(lorahip_demod *dm)
{
    if (dm->geomApplied) return;
    const bool cont = dm->append && !dm->appendFresh;           // an append run continues at the mirrors' read positions (synced by the caller)
    for (size_t c = 0; c < dm->B; c++)
    {
        dm->ch[c].base = c * dm->uniStride; dm->ch[c].len = dm->uniSpc;
        if (!cont) { dm->ch[c].pos = 0; dm->ch[c].callCount = 0; }
    }
    dm->geomApplied = true;
}

static int syncMirrors(lorahip_demod *dm);

static int runRounds(lorahip_demod *dm, const float *iqDev, int64_t *roundsOut)
{
    const size_t N = dm->N, B = dm->B;
    { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; }       // (a failed fetch of the open packets' symbols: nothing is invalidated)
    applyGeometry(dm);
    if (!(dm->append && !dm->appendFresh)) for (auto &k : dm->ch) k.callCount = 0;
    dm->devStateFresh = false;                                  // the mirrors are about to change: the device copy goes stale
    dm->devCarryValid = false;                                  // ... and so do the open packets' symbols it holds (syncMirrors fetched them)
    const DeviceGuard guard(dm->ctx->device);
    const Round hr = carve(dm->h, B);
    std::vector<uint32_t> live, second;
    std::vector<lorahip_work_result> res(B);
    int64_t rounds = 0;
    while (true)
    {
        // ---- who can work this round (LoRaDemod.cpp:148) ----
        live.clear();
        for (size_t c = 0; c < B; c++)
        {
            Channel &k = dm->ch[c];
            if (k.len - k.pos < 2 * N) continue;
            const size_t i = live.size();
            live.push_back(uint32_t(c));
            hr.off[i] = int64_t(k.base + k.pos);
            hr.sel[i] = k.downTable ? LORAHIP_CHIRP_DOWN : LORAHIP_CHIRP_UP;
            hr.idx0[i] = k.fineTuneIndex;
            hr.err[i] = k.finefreqError;
        }
        if (live.empty()) break;
        int rc = launchRound(dm, iqDev, live.size());       // loop A + detect (:157-172)
        if (rc != LORAHIP_OK) return rc;

        // ---- first pass: commit the index, find channels needing window 1 ----
        second.clear();
        for (size_t i = 0; i < live.size(); i++)
        {
            Channel &k = dm->ch[live[i]];
            lorahip_work_result &r = res[live[i]];
            std::memset(&r, 0, sizeof(r));
            r.worked = 1;
            r.state_before = k.state;
            r.value = hr.sym[i];
            r.power = hr.power[i]; r.power_avg = hr.pavg[i]; r.f_index = hr.fidx[i];
            r.snr = r.power - r.power_avg;                                      // :173
            r.fine_idx_before = k.fineTuneIndex; r.fine_err_before = k.finefreqError;
            if ((k.state == ST_FRAMESYNC || k.state == ST_DATASYMBOLS) && nearSquelch(r.snr, dm->thresh)) dm->nNearSquelch++;
            if (nearStep(k.finefreqError * float(LORAHIP_FINE_STEPS))) dm->nNearStep++;
            k.fineTuneIndex = hr.idxOut[i];
            r.fine_idx_after = k.fineTuneIndex;
            if (k.state == ST_FRAMESYNC)
            {
                const bool squelched = r.snr < dm->thresh;                      // :174
                const bool syncd = !squelched && (k.prevValue + 4) / 8 == 0;    // :183
                const bool match0 = (size_t(r.value) + 4) / 8 == unsigned(dm->sync >> 4); // :184
                if (syncd && match0) second.push_back(live[i]);
            }
        }
        std::vector<uint16_t> value1(second.size());
        if (!second.empty())
        {
            for (size_t i = 0; i < second.size(); i++)
            {
                Channel &k = dm->ch[second[i]];
                hr.off[i] = int64_t(k.base + k.pos + N);                         // inBuff[i + N]  :194
                hr.sel[i] = k.downTable ? LORAHIP_CHIRP_DOWN : LORAHIP_CHIRP_UP;
                hr.idx0[i] = k.fineTuneIndex;                                    // int ft = _fineTuneIndex  :191
                hr.err[i] = k.finefreqError;
                if (nearStep(k.finefreqError * float(LORAHIP_FINE_STEPS))) dm->nNearStep++;
            }
            rc = launchRound(dm, iqDev, second.size());
            if (rc != LORAHIP_OK) return rc;
            for (size_t i = 0; i < second.size(); i++)
            {
                lorahip_work_result &r = res[second[i]];
                value1[i] = hr.sym[i];
                // detect(power,powerAvg,fIndex) of window 1 overwrites the locals (:203); snr is not recomputed
                r.power = hr.power[i]; r.power_avg = hr.pavg[i]; r.f_index = hr.fidx[i];
            }
        }

        // ---- second pass: the state machine (:176-312) ----
        size_t si = 0;
        for (size_t i = 0; i < live.size(); i++)
        {
            const uint32_t c = live[i];
            Channel &k = dm->ch[c];
            lorahip_work_result &r = res[c];
            const size_t value = size_t(r.value);
            const bool squelched = r.snr < dm->thresh;
            size_t total = 0;
            switch (k.state)
            {
            case ST_FRAMESYNC:
            {
                const bool syncd = !squelched && (k.prevValue + 4) / 8 == 0;
                const bool match0 = (value + 4) / 8 == unsigned(dm->sync >> 4);
                bool match1 = false;
                if (syncd && match0)
                {
                    match1 = (size_t(value1[si]) + 4) / 8 == unsigned(dm->sync & 0xf);   // :205
                    si++;
                }
                if (syncd && match0 && match1)
                {
                    total = 2 * N;
                    k.state = ST_DOWNCHIRP0;
                    k.downTable = true;
                }
                else if (!squelched)
                {
                    total = N - value;
                    k.finefreqError += r.f_index;
                }
                else
                {
                    total = N;
                    k.finefreqError = 0;
                    k.fineTuneIndex = 0;
                }
            } break;
            case ST_DOWNCHIRP0:
            {
                k.state = ST_DOWNCHIRP1;
                total = N;
                int error = int(value);
                if (value > N / 2) error -= int(N);
                k.freqError = error;
            } break;
            case ST_DOWNCHIRP1:
            {
                k.state = ST_QUARTERCHIRP;
                total = N;
                k.downTable = false;
                k.outSymbols.assign(dm->mtu ? dm->mtu : 1, 0);
                int error = int(value);
                if (value > N / 2) error -= int(N);
                k.freqError = (k.freqError + error) / 2;
                r.signals = 1;
                r.sig_error = k.freqError; r.sig_power = r.power; r.sig_snr = r.snr;
                if (dm->wantSignals)
                {
                    Signal g;
                    g.channel = int32_t(c); g.round = k.callCount; g.error = k.freqError; g.power = r.power; g.snr = r.snr;
                    dm->signals.push_back(g);
                }
            } break;
            case ST_QUARTERCHIRP:
            {
                k.state = ST_DATASYMBOLS;
                total = N / 4 + size_t(k.freqError / 2);
                k.finefreqError += float(k.freqError / 2);
                k.symCount = 0;
            } break;
            case ST_DATASYMBOLS:
            {
                total = N;
                if (k.outSymbols.size() <= k.symCount) k.outSymbols.resize(k.symCount + 1, 0);   // a packet begun on the streaming path
                k.outSymbols[k.symCount++] = int16_t(value);
                if (k.symCount >= dm->mtu || squelched)
                {
                    Packet p;
                    p.channel = int32_t(c);
                    p.round = k.callCount;                                       // = the lock-step round where every stream began with the run
                    p.off = dm->pktSyms.size();
                    p.len = k.symCount;
                    dm->pktSyms.insert(dm->pktSyms.end(), k.outSymbols.begin(), k.outSymbols.begin() + long(k.symCount));
                    dm->packets.push_back(p);
                    r.packet_len = int32_t(k.symCount);
                    k.finefreqError = 0;
                    k.state = ST_FRAMESYNC;
                }
            } break;
            }
            k.prevValue = short(value);                                          // :326
            r.consumed = int64_t(total);
            k.pos += total;                                                      // consume(total)  :320
            k.callCount++;
            dm->workCalls++;
            if (dm->tracing) k.trace.push_back(r);
        }
        rounds++;
    }
    if (roundsOut) *roundsOut = rounds;
    return LORAHIP_OK;
}


/***********************************************************************
 * Streaming path: the device walks every channel (lorahip_stream.hip); the host only drains the
 * per-call records, assembles the packets and keeps the Channel mirrors in step.
 **********************************************************************/
//! device + pinned scratch pair for packed records (grown on demand, both or neither)
static int growDense(lorahip_demod *dm, const size_t bytes)
{
    if (bytes <= dm->denseBytes) return LORAHIP_OK;
    if (dm->dDense) { (void)hipFree(dm->dDense); dm->dDense = nullptr; }
    if (dm->hDense) { (void)hipHostFree(dm->hDense); dm->hDense = nullptr; }
    dm->denseBytes = 0;
    const size_t want = bytes + bytes / 4;
    LORAHIP_TRY(hipMalloc((void **)&dm->dDense, want));
    const hipError_t e = hipHostMalloc((void **)&dm->hDense, want, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipFree(dm->dDense); dm->dDense = nullptr; dm->hDense = nullptr; return hipFail(e, "hipHostMalloc(dense records)"); }
    dm->denseBytes = want;
    return LORAHIP_OK;
}

//! where the pieces of one streaming launch live inside dm->sDev (device) / dm->sHost (host mirror of the head)
struct StreamLayout
{
    size_t B, cap, capPkt, symStride;
    bool tracing, signals;
    size_t oBase, oLen, oState, oN, oNSym, oNPkt, oNSig, oNear, oSum, oEnd, oPkt, oSym, oSig, oCalls, total;
    void make(const size_t B_, const size_t cap_, const size_t capPkt_, const bool tracing_, const size_t carryCap_ = 0, const bool signals_ = false)
    {
        B = B_; cap = cap_; capPkt = capPkt_; tracing = tracing_; signals = signals_;
        symStride = cap_ + carryCap_;                    // a channel's symbol row: what the launch may add behind what it was handed
        size_t cur = 0;
        auto carve = [&cur](const size_t bytes) { const size_t o = cur; cur += align256(bytes); return o; };
        oBase = carve(B * sizeof(long long)); oLen = carve(B * sizeof(long long));
        oState = carve(B * sizeof(StreamState));
        oN = carve(B * sizeof(int)); oNSym = carve(B * sizeof(int)); oNPkt = carve(B * sizeof(int)); oNSig = carve(B * sizeof(int));
        oNear = carve(2 * sizeof(unsigned));
        oSum = carve(sizeof(StreamSummary));
        oEnd = carve(B * sizeof(int2));
        oPkt = carve(B * capPkt * sizeof(StreamPacket));
        oSym = carve(B * symStride * sizeof(short));
        oSig = carve(signals ? B * capPkt * sizeof(StreamSignal) : 0);
        oCalls = carve(tracing ? B * cap * sizeof(lorahip_work_result) : 0);
        total = cur;
    }
};

//! The records of the LAST streaming launch of a run stay on the device until somebody needs them on the host: a receive chain
//! that hands the packets to the batched decoder (lorahip_demod_packets_to_device) never does, and then nothing but the
//! per-channel state and counts (52 B per channel) crosses PCIe per run.
struct PendingLaunch
{
    bool valid;
    StreamLayout lay;
    size_t firstNewPacket;          // index in dm->packets of the first packet of the run this launch belongs to
    int64_t rounds;
    size_t packets, packetSyms;     // what draining will append
    size_t signals;
    bool anyCarryIn, anyOpen;       // a channel entered the launch inside a packet / leaves it inside one
    double drainMs;                 // LORAHIP_DEMOD_TIMING
};

static PendingLaunch &pendingOf(lorahip_demod *dm) { return *static_cast<PendingLaunch *>(dm->pending); }
//! a pipelined receiver step is in flight (lorahip_demod_receive with async = 2): only receive / receive_flush may touch the object
static bool pipeBusy(const lorahip_demod *dm);
static int refuseWhilePiped(const lorahip_demod *dm)
{
    if (!pipeBusy(dm)) return LORAHIP_OK;
    setLastError("a pipelined receiver step is in flight: lorahip_demod_receive_flush first");
    return LORAHIP_E_INVALID;
}
static std::vector<size_t> &carryOf(lorahip_demod *dm) { return dm->carry; }

//! packets of one run in the order the host-driven path posts them: round by round, channels ascending inside a round
static void orderNewPackets(lorahip_demod *dm, const size_t firstNewPacket, const int64_t rounds)
{
    // The new packets were appended channel by channel with rounds ascending inside a channel: a stable counting sort by round
    // restores the order in O(packets + rounds) instead of a comparison sort of tens of thousands of records (which cost more
    // than the kernel). Several launches per run can interleave channels inside a round; that case is detected and sorted.
    const size_t nNew = dm->packets.size() - firstNewPacket;
    if (nNew <= 1) return;
    Packet *first = dm->packets.data() + firstNewPacket;
    bool sorted = true, channelsAscendPerRound = true;
    for (size_t i = 1; i < nNew && sorted; i++)
        sorted = first[i - 1].round < first[i].round || (first[i - 1].round == first[i].round && first[i - 1].channel <= first[i].channel);
    if (!sorted && rounds >= 0 && size_t(rounds) <= 4 * nNew + 1024)
    {
        std::vector<size_t> start(size_t(rounds) + 2, 0);
        for (size_t i = 0; i < nNew; i++) start[size_t(first[i].round) + 1]++;
        for (size_t r = 1; r < start.size(); r++) start[r] += start[r - 1];
        std::vector<Packet> tmp(nNew);
        for (size_t i = 0; i < nNew; i++) tmp[start[size_t(first[i].round)]++] = first[i];
        for (size_t i = 1; i < nNew && channelsAscendPerRound; i++)
            channelsAscendPerRound = tmp[i - 1].round != tmp[i].round || tmp[i - 1].channel <= tmp[i].channel;
        std::copy(tmp.begin(), tmp.end(), first);
        sorted = channelsAscendPerRound;
    }
    if (!sorted)
        std::stable_sort(first, first + nNew,
                         [](const Packet &x, const Packet &y) { return x.round != y.round ? x.round < y.round : x.channel < y.channel; });
}

//! the head of the streaming buffers inside dm->sHost: offsets that depend on the channel count only (StreamLayout::make)
static StreamLayout headLayout(const lorahip_demod *dm)
{
    StreamLayout L;
    L.make(dm->B, 8, 4, false);
    return L;
}

//! The per-channel state and counts of the last streaming launch into their pinned copy (52 B per channel): a run itself reads back
//! only its summary; whoever needs the arrays -- the Channel mirrors, consumed(), a drain of the records -- asks here first.
static int ensureHead(lorahip_demod *dm)
{
    if (!dm->headStale || dm->sHost == nullptr || dm->sDev == nullptr) return LORAHIP_OK;
    const StreamLayout L = headLayout(dm);
    const DeviceGuard guard(dm->ctx->device);
    LORAHIP_TRY(hipMemcpyAsync(dm->sHost + L.oState, dm->sDev + L.oState, L.oPkt - L.oState, hipMemcpyDeviceToHost, dm->ctx->stream));
    LORAHIP_TRY(hipStreamSynchronize(dm->ctx->stream));
    dm->headStale = false;
    return LORAHIP_OK;
}

//! records of one launch (still in dm->sDev; the counts in dm->sHost) -> the host queue, the per-channel traces and open packets
static int drainLaunch(lorahip_demod *dm, const StreamLayout &L)
{
    lorahip_ctx *ctx = dm->ctx;
    const size_t B = L.B;
    { const int rc = ensureHead(dm); if (rc != LORAHIP_OK) return rc; }
    char *h = dm->sHost, *d = dm->sDev;
    const int *hN = reinterpret_cast<int *>(h + L.oN), *hNSym = reinterpret_cast<int *>(h + L.oNSym), *hNPkt = reinterpret_cast<int *>(h + L.oNPkt);
    std::vector<size_t> &carry = carryOf(dm);
    // only as many columns of the [channel][capacity] record arrays as the fullest channel used cross PCIe: the capacities are
    // worst-case bounds, several times what a run fills
    const int *hNSig = reinterpret_cast<int *>(h + L.oNSig);
    size_t maxSym = 0, maxPkt = 0, maxCalls = 0, maxSig = 0;
    for (size_t c = 0; c < B; c++)
    {
        if (size_t(hNSym[c]) > maxSym) maxSym = size_t(hNSym[c]);
        if (size_t(hNPkt[c]) > maxPkt) maxPkt = size_t(hNPkt[c]);
        if (size_t(hN[c]) > maxCalls) maxCalls = size_t(hN[c]);
        if (L.signals && size_t(hNSig[c]) > maxSig) maxSig = size_t(hNSig[c]);
    }
    const size_t nbPkt = align256(B * maxPkt * sizeof(StreamPacket)), nbSym = align256(B * maxSym * sizeof(short));
    const size_t nbCalls = L.tracing ? align256(B * maxCalls * sizeof(lorahip_work_result)) : 0;
    const size_t nbSig = align256(B * maxSig * sizeof(StreamSignal));
    const size_t nbDense = nbPkt + nbSym + nbCalls + nbSig;
    { const int grc = growDense(dm, nbDense); if (grc != LORAHIP_OK) return grc; }
    LORAHIP_TRY(launchCompactRows(dm->dDense, d + L.oPkt, B, L.capPkt * sizeof(StreamPacket), maxPkt * sizeof(StreamPacket), ctx->stream));
    LORAHIP_TRY(launchCompactRows(dm->dDense + nbPkt, d + L.oSym, B, L.symStride * sizeof(short), maxSym * sizeof(short), ctx->stream));
    if (L.tracing)
        LORAHIP_TRY(launchCompactRows(dm->dDense + nbPkt + nbSym, d + L.oCalls, B, L.cap * sizeof(lorahip_work_result),
                                      maxCalls * sizeof(lorahip_work_result), ctx->stream));
    if (maxSig)
        LORAHIP_TRY(launchCompactRows(dm->dDense + nbPkt + nbSym + nbCalls, d + L.oSig, B, L.capPkt * sizeof(StreamSignal), maxSig * sizeof(StreamSignal), ctx->stream));
    if (nbDense) LORAHIP_TRY(hipMemcpyAsync(dm->hDense, dm->dDense, nbDense, hipMemcpyDeviceToHost, ctx->stream));
    LORAHIP_TRY(hipStreamSynchronize(ctx->stream));
    const StreamPacket *hPkt = reinterpret_cast<const StreamPacket *>(dm->hDense);
    const short *hSym = reinterpret_cast<const short *>(dm->hDense + nbPkt);
    const lorahip_work_result *hCalls = reinterpret_cast<const lorahip_work_result *>(dm->hDense + nbPkt + nbSym);
    const StreamSignal *hSig = reinterpret_cast<const StreamSignal *>(dm->hDense + nbPkt + nbSym + nbCalls);
    for (size_t c = 0; c < B; c++)
    {
        Channel &k = dm->ch[c];
        for (int j = 0; L.signals && j < hNSig[c]; j++)
        {
            const StreamSignal &q = hSig[c * maxSig + size_t(j)];
            Signal g;
            g.channel = int32_t(c); g.round = q.callIndex; g.error = q.error; g.power = q.power; g.snr = q.snr;
            dm->signals.push_back(g);
        }
        // symbols of this launch continue the packet the previous launches left open (k.outSymbols[0..carry[c]))
        const short *sy = hSym + c * maxSym;
        size_t p = 0;
        for (int j = 0; j < hNPkt[c]; j++)
        {
            const StreamPacket &q = hPkt[c * maxPkt + size_t(j)];
            Packet pk;
            pk.channel = int32_t(c);
            pk.round = q.callIndex;
            pk.off = dm->pktSyms.size();
            pk.len = size_t(q.len);
            dm->pktSyms.insert(dm->pktSyms.end(), k.outSymbols.begin(), k.outSymbols.begin() + long(carry[c]));
            const size_t fresh = size_t(q.len) - carry[c];
            dm->pktSyms.insert(dm->pktSyms.end(), sy + p, sy + p + fresh);
            p += fresh;
            carry[c] = 0;
            dm->packets.push_back(pk);
        }
        // what is left belongs to a packet still being received
        const size_t left = size_t(hNSym[c]) - p;
        if (left)
        {
            if (k.outSymbols.size() < carry[c] + left) k.outSymbols.resize(carry[c] + left, 0);
            for (size_t i = 0; i < left; i++) k.outSymbols[carry[c] + i] = sy[p + i];
            carry[c] += left;
        }
        if (L.tracing) k.trace.insert(k.trace.end(), hCalls + c * maxCalls, hCalls + c * maxCalls + hN[c]);
    }
    return LORAHIP_OK;
}

//! bring a pending launch's records to the host (any accessor of the queue / traces does this first)
static int drainPending(lorahip_demod *dm)
{
    PendingLaunch &P = pendingOf(dm);
    if (!P.valid) return LORAHIP_OK;
    const DeviceGuard guard(dm->ctx->device);
    typedef std::chrono::steady_clock Clock;
    const Clock::time_point t0 = Clock::now();
    // the records stay pending until they HAVE been drained: after a failed drain (HIP error, no memory for the staging) the
    // packet counts reported so far remain true and a later accessor retries. drainLaunch appends to the queue only after its
    // last fallible step (the copy back), so a retry cannot duplicate packets.
    const int rc = drainLaunch(dm, P.lay);
    if (rc != LORAHIP_OK) return rc;
    P.valid = false;
    dm->hostCarryStale = false;                       // drainLaunch has written the open packets' symbols of this launch into the mirrors
    orderNewPackets(dm, P.firstNewPacket, P.rounds);
    P.drainMs = std::chrono::duration<double>(Clock::now() - t0).count() * 1e3;
    static const bool timing = std::getenv("LORAHIP_DEMOD_TIMING") != nullptr;
    if (timing) std::fprintf(stderr, "lorahip demod: records of the last launch drained to the host in %.3f ms\n", P.drainMs);
    return LORAHIP_OK;
}

static StreamState *hostStates(lorahip_demod *dm)
{
    return reinterpret_cast<StreamState *>(dm->sHost + headLayout(dm).oState);
}

//! the open packets' symbols the device holds (dCarry) into the mirrors' outSymbols; the mirrors' state must be current
static int fetchCarry(lorahip_demod *dm)
{
    if (!dm->hostCarryStale) return LORAHIP_OK;
    const size_t B = dm->B, cap = dm->carryCap;
    bool any = false;
    for (size_t c = 0; c < B && !any; c++) any = dm->ch[c].state == ST_DATASYMBOLS && dm->ch[c].symCount != 0;
    if (any && dm->dCarry && dm->devCarryValid)
    {
        const DeviceGuard guard(dm->ctx->device);
        std::vector<short> rows;
        try { rows.resize(B * cap); } catch (...) { setLastError("no memory for the open packets' symbols"); return LORAHIP_E_NOMEM; }   // nothing may cross the C ABI
        LORAHIP_TRY(hipMemcpyAsync(rows.data(), dm->dCarry, B * cap * sizeof(short), hipMemcpyDeviceToHost, dm->ctx->stream));
        LORAHIP_TRY(hipStreamSynchronize(dm->ctx->stream));
        for (size_t c = 0; c < B; c++)
        {
            Channel &k = dm->ch[c];
            if (k.state != ST_DATASYMBOLS || k.symCount == 0) continue;
            const size_t n = k.symCount < cap ? k.symCount : cap;
            if (k.outSymbols.size() < k.symCount) k.outSymbols.resize(k.symCount, 0);
            std::memcpy(k.outSymbols.data(), rows.data() + c * cap, n * sizeof(short));
        }
    }
    dm->hostCarryStale = false;
    return LORAHIP_OK;
}

//! bring the Channel mirrors up to date with the device's state (its pinned copy): only the paths that read them pay for it
static int syncMirrors(lorahip_demod *dm)
{
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    if (dm->mirrorsStale && dm->sHost)
    {
        { const int rc = ensureHead(dm); if (rc != LORAHIP_OK) return rc; }
        const StreamState *hs = hostStates(dm);
        for (size_t c = 0; c < dm->B; c++)
        {
            Channel &k = dm->ch[c];
            const StreamState &st = hs[c];
            k.state = st.state; k.downTable = st.downTable != 0; k.prevValue = short(st.prevValue); k.freqError = st.freqError;
            k.fineTuneIndex = st.fineTuneIndex; k.finefreqError = st.finefreqError; k.symCount = size_t(st.symCount);
            if (dm->posOnDevice) { k.pos = size_t(st.pos); k.callCount = st.callCount; }
        }
    }
    dm->mirrorsStale = false;
    // before a deferred activate() hides which channels were inside a packet. A failed copy leaves hostCarryStale set and the
    // device's copy valid: the caller must not invalidate it (it returns the error instead)
    { const int rc = fetchCarry(dm); if (rc != LORAHIP_OK) return rc; }
    if (dm->activatePending)
    {
        for (auto &k : dm->ch) { k.state = ST_FRAMESYNC; k.downTable = false; }     // activate() (:139-143), deferred
        dm->activatePending = false;
        dm->devStateFresh = false;                      // the device copy does not have it: the next streaming run uploads
    }
    return LORAHIP_OK;
}

//! StreamArgs::lanes of this object's launches: what was asked for; where nothing was and sibling parts share the device (a mixed
//! object), the automatic choice made HERE with their wavefronts counted -- as an explicit choice for launchStream
static int lanesArg(const lorahip_demod *dm)
{
    if (dm->streamLanes != 0 || dm->coWaves == 0) return dm->streamLanes;
    const int l = streamLanesChosen(dm->ctx->sf, unsigned(dm->B), 0, dm->coWaves);
    return l == dm->ctx->sf - 4 ? -1 : l;
}

static int runStream(lorahip_demod *dm, const float *iqDev, int64_t *roundsOut)
{
    lorahip_ctx *ctx = dm->ctx;
    const size_t N = dm->N, B = dm->B;
    const DeviceGuard guard(ctx->device);
    // the previous run's records live in the buffers this run is about to reuse
    { const int rc = drainPending(dm); if (rc != LORAHIP_OK) return rc; }
    // diagnostic: LORAHIP_DEMOD_TIMING=1 prints where a run's host wall clock goes
    static const bool timing = std::getenv("LORAHIP_DEMOD_TIMING") != nullptr;
    typedef std::chrono::steady_clock Clock;
    const Clock::time_point t0 = Clock::now();
    double tDev = 0, tAsm = 0;
    const bool cont = dm->append && !dm->appendFresh;           // the channels continue where the last append run left them
    // an append run works through what is new since the last one plus what that one left (fewer than 2N samples per channel)
    size_t maxLen = dm->uniform ? (cont && dm->appendPrev <= dm->uniSpc ? dm->uniSpc - dm->appendPrev + 2 * N : dm->uniSpc) : 0;
    if (!dm->uniform) for (size_t c = 0; c < B; c++) if (dm->ch[c].len - dm->ch[c].pos > maxLen) maxLen = dm->ch[c].len - dm->ch[c].pos;
    // work() calls per channel per launch: enough for a clean stream in one launch, bounded so that the
    // per-launch buffers stay moderate (the launch is resumable)
    const size_t perCall = sizeof(short) + (dm->tracing ? sizeof(lorahip_work_result) : 0) + sizeof(StreamPacket) / 4 + 1 + (dm->wantSignals ? sizeof(StreamSignal) / 4 : 0);
    // A call consumes N samples except around a frame's sync (N - value, N/4 + error/2: LoRaDemod.cpp:219, :278) -- a handful of short
    // calls per frame -- and in an unsquelched FRAMESYNC window that does not sync (N - value every call: a receiver idling on noise
    // above its threshold makes about two calls per N samples). A launch whose record buffers fill is resumed, but the records have
    // to be drained in between (measured: 9.7 ms instead of 7.0 ms per run at SF7, 16384 channels x 16 frames), so the capacity
    // starts with an eighth of headroom and follows what the runs of this receiver actually needed (never shrinks).
    size_t cap = (maxLen / N) * dm->callsPerWindowQ8 / 256 + 64;
    if (cap > 65536) cap = 65536;
    const size_t capMem = (size_t(1) << 30) / (B * perCall);
    if (cap > capMem) cap = capMem;
    if (cap < 8) cap = 8;
    // lorahip_demod_set_record_capacity: a bound on the per-launch record capacity (tests force the resume path with it)
    if (dm->streamCapMax != 0 && cap > dm->streamCapMax) cap = dm->streamCapMax;
    const size_t capPkt = cap / 4 + 2;               // a packet costs at least 5 calls (3 sync, quarter, 1 symbol)

    // open packets on the device (dCarry): rows of mtu + 1 symbols; receivers with longer packets than this keep the host path
    // (the carry rows and their share of the symbol rows stay inside the memory bound of the record buffers above)
    const size_t kCarryLimit = 4096;
    bool useDevCarry = dm->mtu <= kCarryLimit && B * (dm->mtu + 1 < 64 ? 64 : dm->mtu + 1) * sizeof(short) <= (size_t(1) << 29);
    if (useDevCarry && dm->mtu + 1 > dm->carryCap)
    {
        { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; }      // what the device holds of open packets goes to the mirrors first
        if (dm->dCarry) { (void)hipFree(dm->dCarry); dm->dCarry = nullptr; }
        dm->carryCap = 0; dm->devCarryValid = false;
        const size_t rows = dm->mtu + 1 < 64 ? 64 : dm->mtu + 1;
        LORAHIP_TRY(hipMalloc((void **)&dm->dCarry, B * rows * sizeof(short)));
        dm->carryCap = rows;
    }

    StreamLayout L;
    long rowPad = 0;
#ifdef LORAHIP_ALL_VARIANTS
    // measurement hook of the profiling build only (tools/row_stride.sh): extra entries per symbol row; never fewer than cap + carryCap
    if (const char *e = std::getenv("LORAHIP_SYM_PAD")) { const long v = std::atol(e); if (v >= 0 && v < (1 << 20)) rowPad = v; }
#endif
    L.make(B, cap, capPkt, dm->tracing, size_t(long(useDevCarry ? dm->carryCap : 0) + rowPad), dm->wantSignals);
    if (L.total > dm->sBytes)
    {
        { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; }      // the pinned copy of the state goes away with the buffers
        if (dm->sDev) { (void)hipFree(dm->sDev); dm->sDev = nullptr; }
        if (dm->sHost) { (void)hipHostFree(dm->sHost); dm->sHost = nullptr; }
        dm->sBytes = 0;
        dm->devStateFresh = false;
        dm->headStale = false;                       // (syncMirrors has read what there was)
        // a quarter more than this run needs: the chunks of a running receiver differ by the remainders they start with, and every
        // growth costs two allocations and an upload of the state
        const size_t want = L.total + L.total / 4;
        LORAHIP_TRY(hipMalloc((void **)&dm->sDev, want));
        LORAHIP_TRY(hipHostMalloc((void **)&dm->sHost, L.oPkt, hipHostMallocDefault));       // the host mirrors only the head: placement, state, counts
        dm->sBytes = want;
        // the kernels' near-threshold counters only ever add: they start at zero with the buffers
        LORAHIP_TRY(hipMemsetAsync(dm->sDev + L.oNear, 0, 2 * sizeof(unsigned), ctx->stream));
        dm->nearSeen[0] = dm->nearSeen[1] = 0;
    }
    char *h = dm->sHost, *d = dm->sDev;
    long long *hBase = reinterpret_cast<long long *>(h + L.oBase), *hLen = reinterpret_cast<long long *>(h + L.oLen);
    StreamState *hState = reinterpret_cast<StreamState *>(h + L.oState);
    const StreamSummary *hSum = reinterpret_cast<const StreamSummary *>(h + L.oSum);
    std::vector<size_t> &carry = carryOf(dm);
    carry.assign(B, 0);
    bool anyCarryIn = false;
    size_t maxCarry = 0;
    // A steady receiver -- run after run in this mode, device buffers, nothing touched in between -- uploads nothing and reads back
    // 72 bytes: the state is where the last run left it on the device, the placement of uniform streams is computed by the kernel,
    // an activate() in between travels as a flag, the open packets' symbols are in the device's carry rows, and what the host needs
    // to know of the last run is its summary (streamSummary). The per-channel state and counts are fetched when somebody asks.
    const bool resident = dm->devStateFresh;
    const bool activate = resident && dm->activatePending;
    if (resident)
    {
        // symbols of packets that are still being received when the run starts: the last summary says whether there are any
        if (!activate && dm->lastSum.anyOpen && dm->lastSum.openSyms > 0) { anyCarryIn = true; maxCarry = size_t(dm->lastSum.maxOpen); }
        // which channels, and how many symbols each, only matters where the HOST hands them on
        const bool perChannel = anyCarryIn && (!useDevCarry || maxCarry >= dm->carryCap || !dm->devCarryValid);
        if (perChannel)
        {
            { const int rc = ensureHead(dm); if (rc != LORAHIP_OK) return rc; }
            for (size_t c = 0; c < B; c++) if (hState[c].state == ST_DATASYMBOLS && hState[c].symCount) carry[c] = size_t(hState[c].symCount);
        }
    }
    else
    {
        { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; }
        for (size_t c = 0; c < B; c++)
        {
            Channel &k = dm->ch[c];
            StreamState &st = hState[c];
            st.state = k.state; st.downTable = k.downTable ? 1 : 0; st.prevValue = k.prevValue; st.freqError = k.freqError;
            st.fineTuneIndex = k.fineTuneIndex; st.finefreqError = k.finefreqError; st.symCount = int(k.symCount);
            st.callCount = cont ? int(k.callCount) : 0;
            st.pos = cont ? (long long)k.pos : 0;
            if (k.outSymbols.size() < k.symCount) k.outSymbols.resize(k.symCount, 0);
            // symbols of a packet that is still being received when the run starts. _symCount itself is only reset at
            // QUARTERCHIRP (:279), so outside DATASYMBOLS it still holds the length of the LAST packet: nothing is carried then
            carry[c] = k.state == ST_DATASYMBOLS ? k.symCount : 0;
            anyCarryIn = anyCarryIn || carry[c] != 0;
            if (carry[c] > maxCarry) maxCarry = carry[c];
        }
        LORAHIP_TRY(hipMemcpyAsync(d + L.oState, h + L.oState, L.oN - L.oState, hipMemcpyHostToDevice, ctx->stream));
        dm->headStale = false;                        // the pinned copy IS what the device is being given
    }
    // Where do the symbols of those packets come from? From the device's own copy (the carry rows: the kernel copies a channel's open
    // packet to the head of its symbol row and appends behind it, so the host sees packets without a past), unless a packet is
    // longer than those rows -- then from the mirrors' outSymbols, as the launches of a resumed run do among themselves.
    if (useDevCarry && maxCarry >= dm->carryCap) useDevCarry = false;
    if (useDevCarry)
    {
        if (anyCarryIn && !dm->devCarryValid)
        {
            // the mirrors hold them (a host-driven run, or a resumed one, came before): up they go
            const size_t cc = dm->carryCap;
            { const int grc = growDense(dm, B * cc * sizeof(short)); if (grc != LORAHIP_OK) return grc; }
            short *stage = reinterpret_cast<short *>(dm->hDense);
            for (size_t c = 0; c < B; c++)
                if (carry[c]) std::memcpy(stage + c * cc, dm->ch[c].outSymbols.data(), (carry[c] < dm->ch[c].outSymbols.size() ? carry[c] : dm->ch[c].outSymbols.size()) * sizeof(short));
            LORAHIP_TRY(hipMemcpyAsync(dm->dCarry, stage, B * cc * sizeof(short), hipMemcpyHostToDevice, ctx->stream));
            LORAHIP_TRY(hipStreamSynchronize(ctx->stream));       // the pinned scratch is reused
        }
        carry.assign(B, 0);
        anyCarryIn = false;                           // as far as the host's assembly of packets is concerned
    }
    else
    {
        if (dm->hostCarryStale) { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; }      // (a deferred activate() then travels with the state upload: see syncMirrors)
        if (anyCarryIn)
            for (size_t c = 0; c < B; c++) if (carry[c] && dm->ch[c].outSymbols.size() < carry[c]) dm->ch[c].outSymbols.resize(carry[c], 0);
        dm->devCarryValid = false;
    }
    if (!dm->uniform)
    {
        for (size_t c = 0; c < B; c++) { hBase[c] = (long long)dm->ch[c].base; hLen[c] = (long long)dm->ch[c].len; }
        LORAHIP_TRY(hipMemcpyAsync(d, h, L.oState, hipMemcpyHostToDevice, ctx->stream));       // base, len
    }

    StreamArgs a;
    a.iq = reinterpret_cast<const float2 *>(iqDev);
    a.base = reinterpret_cast<const long long *>(d + L.oBase);
    a.len = reinterpret_cast<const long long *>(d + L.oLen);
    a.uniformLen = dm->uniform ? (long long)dm->uniSpc : -1;
    a.uniformStride = (long long)dm->uniStride;
    // first launch of the run: every channel starts at sample 0, call 0 (unless the run continues the streams), behind its open packet's
    // symbols, and leaves the packet it is inside at the end in the carry rows
    a.flags = (cont ? 0 : 1) | (activate ? 2 : 0) | (useDevCarry ? 4 | 8 : 0);
    a.carry = dm->dCarry; a.carryCap = int(dm->carryCap);
    // The grid (lorahip_demod_set_stream_grid; 0 = the kernels' own default). One workgroup per channel set, however many there are,
    // at every SF but 11: the dispatcher hands a free slot the next set. A PERSISTENT grid (the resident number of workgroups, each
    // looping over sets) lost there even with the alternating priority (profiles/r04/s22_*: SF7 32768 channels 0.34 against 0.44,
    // SF9 0.36 against 0.40; SF8 / 10 / 12 equal) -- the loop costs the wave-per-channel-set kernels registers -- and is the default
    // only where one-by-one placement leaves slots unusable: SF11 (lorahip_wide.hip::launchStreamWideCfg).
    a.maxBlocks = dm->streamGrid; a.lanes = lanesArg(dm); a.lastRoundFrom = 0;
#if defined(LORAHIP_ALL_VARIANTS) || defined(LORAHIP_STREAM_PERSIST)
    if (const char *e = std::getenv("LORAHIP_STREAM_BLOCKS")) a.maxBlocks = std::atoi(e);        // e.g. 512: two workgroups of 256 threads per CU
#endif
    a.state = reinterpret_cast<StreamState *>(d + L.oState);
    a.nCalls = reinterpret_cast<int *>(d + L.oN);
    a.nSym = reinterpret_cast<int *>(d + L.oNSym);
    a.nPkt = reinterpret_cast<int *>(d + L.oNPkt);
    a.pktOut = reinterpret_cast<StreamPacket *>(d + L.oPkt);
    a.symOut = reinterpret_cast<short *>(d + L.oSym);
    a.sigOut = dm->wantSignals ? reinterpret_cast<StreamSignal *>(d + L.oSig) : nullptr;
    a.nSig = reinterpret_cast<int *>(d + L.oNSig);
    a.end = reinterpret_cast<int2 *>(d + L.oEnd);
    a.calls = dm->tracing ? reinterpret_cast<lorahip_work_result *>(d + L.oCalls) : nullptr;
    a.down = ctx->dDown; a.fine = ctx->dFine; a.twStage = ctx->dTwStage;
    a.fineA = ctx->fineGather ? nullptr : ctx->dFineA;
    a.fineB = ctx->fineGather ? nullptr : ctx->dFineB;
    a.nChannels = unsigned(B);
    a.cap = int(cap);
    a.symStride = int(L.symStride);
    a.capPkt = int(capPkt);
    a.powerScale = ctx->powerScale;
    a.thresh = dm->thresh;
    a.sync = dm->sync;
    a.mtu = dm->mtu > 0xffffffffu ? 0xffffffffu : unsigned(dm->mtu);
    a.near = reinterpret_cast<unsigned *>(d + L.oNear);
    if (activate) dm->activatePending = false;        // applied by the kernel when it loads the state
    dm->devStateFresh = false;                        // until the run has completed
    dm->mirrorsStale = true;

    const size_t firstNewPacket = dm->packets.size();
    const Clock::time_point t1 = Clock::now();
    dm->kernelMs = 0.0;
    if (dm->evK0 == nullptr) { LORAHIP_TRY(hipEventCreate(&dm->evK0)); LORAHIP_TRY(hipEventCreate(&dm->evK1)); }
    bool lastPending = false;
    int launches = 0;
    size_t runCalls = 0;                              // calls of the fullest channel, launch by launch
    StreamSummary sum;
    while (true)
    {
        const Clock::time_point ta = Clock::now();
        LORAHIP_TRY(hipEventRecord(dm->evK0, ctx->stream));
        LORAHIP_TRY(launchStream(ctx->sf, a, ctx->stream));
        LORAHIP_TRY(hipEventRecord(dm->evK1, ctx->stream));
        a.flags = 0;                                  // a resumed launch continues where the state says
        // what the host needs of the launch, reduced on the device: 72 bytes come back, not 52 per channel
        // (written by the kernel straight into the pinned staging block when the device can address it: no copy to enqueue)
        void *sumDev = nullptr;
        const bool direct = hipHostGetDevicePointer(&sumDev, const_cast<StreamSummary *>(hSum), 0) == hipSuccess && sumDev != nullptr;
        if (!direct) (void)hipGetLastError();
        LORAHIP_TRY(launchStreamSummary(a.end, a.nCalls, a.nSym, a.nPkt, dm->wantSignals ? a.nSig : nullptr, B, int(cap), int(capPkt), a.near, dm->dSumScratch,
                                        direct ? static_cast<StreamSummary *>(sumDev) : reinterpret_cast<StreamSummary *>(d + L.oSum), ctx->stream));
        if (!direct) LORAHIP_TRY(hipMemcpyAsync(h + L.oSum, d + L.oSum, sizeof(StreamSummary), hipMemcpyDeviceToHost, ctx->stream));
        LORAHIP_TRY(hipStreamSynchronize(ctx->stream));
        dm->headStale = true;                         // the pinned copy of the per-channel state and counts lags the device now
        { float ms = 0.0f; if (hipEventElapsedTime(&ms, dm->evK0, dm->evK1) == hipSuccess) dm->kernelMs += ms; }
        const Clock::time_point tb = Clock::now();
        tDev += std::chrono::duration<double>(tb - ta).count();
        sum = *hSum;
        // the kernels' counters run on (they only add): what this launch added
        dm->nNearSquelch += int64_t(unsigned(sum.nearSquelch - dm->nearSeen[0]));
        dm->nNearStep += int64_t(unsigned(sum.nearStep - dm->nearSeen[1]));
        dm->nearSeen[0] = sum.nearSquelch; dm->nearSeen[1] = sum.nearStep;
        const bool more = sum.more != 0;
        dm->workCalls += sum.calls;
        runCalls += size_t(sum.fullest);
        launches++;
        // a launch that must be resumed hands its records over now (the next one reuses the buffers); so does a traced run (its
        // callers read the trace next). Otherwise the records wait on the device.
        if (more || dm->tracing)
        {
            const int rc = drainLaunch(dm, L);
            if (rc != LORAHIP_OK) return rc;
            tAsm += std::chrono::duration<double>(Clock::now() - tb).count();
            anyCarryIn = false;
            for (size_t c = 0; c < B; c++) anyCarryIn = anyCarryIn || carry[c] != 0;
        }
        else lastPending = true;
        if (!more) break;
    }
    const Clock::time_point t2 = Clock::now();
    const int64_t rounds = sum.maxCallCount;
    const bool anyOpen = sum.anyOpen != 0;
    const size_t openSyms = size_t(sum.openSyms);
    size_t carriedIn = 0;
    if (anyCarryIn || lastPending) for (size_t c = 0; c < B; c++) carriedIn += carry[c];
    dm->lastSum = sum;
    dm->lastLaunches = launches;
    if (launches > 1 && maxLen >= N)
    {
        // the run had to be resumed: size the record buffers of the runs to come for the fullest channel's rate of calls, a quarter on top
        const size_t q8 = (runCalls * 256 / (maxLen / N)) * 5 / 4 + 16;
        if (q8 > dm->callsPerWindowQ8) dm->callsPerWindowQ8 = q8 > size_t(256) * 64 ? size_t(256) * 64 : q8;
    }
    dm->devStateFresh = true;                         // the device holds the current state; the pinned copy (headStale) and the mirrors lag
    dm->posOnDevice = true;
    if (useDevCarry && launches == 1)
    {
        // the packets the channels are inside now: the kernel left the last symCount entries of their symbol rows in the carry rows
        dm->devCarryValid = true;
        dm->hostCarryStale = lastPending;             // a drain (traced runs) has brought the mirrors' outSymbols up to date already
    }
    else { dm->devCarryValid = false; dm->hostCarryStale = false; }      // a resumed run: its launches handed the symbols on through the mirrors
    PendingLaunch &P = pendingOf(dm);
    if (lastPending)
    {
        P.valid = true;
        P.lay = L;
        P.firstNewPacket = firstNewPacket;
        P.rounds = rounds;
        P.packets = size_t(sum.packets);
        P.packetSyms = carriedIn + size_t(sum.syms) - openSyms;     // symbols of the packets completed by this launch
        P.signals = size_t(sum.signals);
        P.anyCarryIn = anyCarryIn;
        P.anyOpen = anyOpen;
        P.drainMs = 0.0;
    }
    else orderNewPackets(dm, firstNewPacket, rounds);
    if (roundsOut) *roundsOut = rounds;
    if (timing)
        std::fprintf(stderr, "lorahip demod run: setup+H2D %.3f ms%s, kernel+summary D2H+sync %.3f ms (kernel %.3f), record drain %.3f ms%s, rest %.3f ms\n",
                     std::chrono::duration<double>(t1 - t0).count() * 1e3, resident ? " (state resident on the device)" : "", tDev * 1e3, dm->kernelMs, tAsm * 1e3,
                     lastPending ? " (last launch left on the device)" : "", std::chrono::duration<double>(Clock::now() - t2).count() * 1e3);
    return LORAHIP_OK;
}

/***********************************************************************
 * The PIPELINED receiver step (lorahip_demod_receive with async = 2). A receiver step is: streaming kernel, 72-byte summary back,
 * packets packed for the decoder. Run strictly one after the other, the host's share (launch latencies, the wait for the summary,
 * the packing launches: ~60 us) is exposed once per step -- half the step at chunks of 8 windows. Here step k's kernel is launched
 * BEFORE step k-1's summary is read: the host waits for summary k-1 and packs step k-1's packets while kernel k runs, and the
 * caller gets every packet one step later (lorahip_demod_receive_flush delivers the last step's). Nothing the host learns from
 * summary k-1 is needed to launch kernel k: the per-channel state, the read positions and the open packets' symbols are on the
 * device, in stream order; a channel that filled its record buffer in step k-1 is simply continued by kernel k (the kernels are
 * resumable by design). The per-launch record arrays exist twice (kernel k writes one set while step k-1's are packed from the other).
 **********************************************************************/
struct Pipe
{
    bool active;                    // steps are in flight: only receive / flush may touch the object
    unsigned k;                     // steps launched since the pipeline was entered
    char *dev[2]; size_t bytes[2];  // record sets (a StreamLayout each; the state and the carry rows are the object's own)
    StreamLayout lay[2];
    StreamSummary *hSum;            // [2] pinned and mapped: the summary kernel writes here directly (no copy to enqueue)
    hipEvent_t ev[2];
    bool pending[2];                // the set's kernel has been launched, its summary not read yet
    bool held[2];                   // the set's summary has been read, its packets are still in the set (rows too small: nothing is lost)
    size_t nPk[2]; int64_t nCalls[2];   // ... what that summary said
    size_t nSig[2]; bool sigs[2];       // ... and the signals kept in the set (the step ran with lorahip_demod_set_signals on)
    hipStream_t side;               // step k's packets are packed HERE while step k + 1's kernel runs on the launch stream
    hipEvent_t packDone;            // ... which waits for this before anything later (the next kernel reuses the record set, the caller reads the rows)
    hipEvent_t entry;               // ... and the side stream for this: where the launch stream stood when the call began (the caller's
                                    // consumer of the rows handed out by the call before, earlier packing on the launch stream)
    const float *iq; size_t rowStride;  // the rows the steps read (a flush that has to resume a full channel continues on them)
    // ---- the RESIDENT receiver (lorahip_demod_receive, async = 3): one launch across the steps, lorahip_streamkernel.h (RES) ----
    struct Resident
    {
        bool active;                    // the kernel is on the device: only receive (async = 3) / flush may touch the object
        bool unavailable;               // tried and refused for this object (no instance, the grid not resident at once, a step timed out)
        unsigned seq;                   // steps rung
        unsigned reported;              // steps whose report the caller has had
        ResidentCtl *ctl;               // device: the mirror of the ring, the steps' counters
        ResidentHost *host;             // pinned and mapped: the ring the host writes, the steps' reports, the abort flag
        char *rec; size_t recBytes;     // the channels' records of a step (a StreamLayout; the state and the carry rows are the object's own)
        StreamLayout lay;
        hipStream_t run;                // the kernel's stream
        hipEvent_t ev;
        unsigned grid;
        size_t lastValid;
        bool lastMore;                  // the last reported step left a channel with samples it could not record
        bool sigs;                      // the launch keeps signal records
        bool tail;                      // the last step left fewer than 16 valid samples per row untouched (the flush's ordinary step takes them)
        unsigned nRep; size_t repPk[RES_DEPTH_MAX + 1], repSg[RES_DEPTH_MAX + 1];   // the steps the last call reported, oldest first (lorahip_demod_receive_steps)
        unsigned depth;                 // steps the caller lets the receiver run ahead of the last report (1 .. RES_DEPTH_MAX): it has depth + 1 sets of rows
        // LORAHIP_RESIDENT_DEBUG prints these at the flush: where the host's share of a step goes
        uint64_t dbgReports, dbgImmediate, dbgCalls;
        double dbgWaitNs, dbgBetweenNs;
        std::chrono::steady_clock::time_point dbgLast;
    } res;
};
static Pipe &pipeOf(lorahip_demod *dm) { return *static_cast<Pipe *>(dm->pipe); }
static bool pipeBusy(const lorahip_demod *dm)
{
    return dm->pipe != nullptr && (static_cast<const Pipe *>(dm->pipe)->active || static_cast<const Pipe *>(dm->pipe)->res.active);
}

//! per-launch record capacity (work() calls per channel) for streams of at most maxLen samples: see runStream
static void streamCapacity(const lorahip_demod *dm, const size_t maxLen, size_t &cap, size_t &capPkt)
{
    const size_t perCall = sizeof(short) + (dm->tracing ? sizeof(lorahip_work_result) : 0) + sizeof(StreamPacket) / 4 + 1 + (dm->wantSignals ? sizeof(StreamSignal) / 4 : 0);
    cap = (maxLen / dm->N) * dm->callsPerWindowQ8 / 256 + 64;
    if (cap > 65536) cap = 65536;
    const size_t capMem = (size_t(1) << 30) / (dm->B * perCall);
    if (cap > capMem) cap = capMem;
    if (cap < 8) cap = 8;
    if (dm->streamCapMax != 0 && cap > dm->streamCapMax) cap = dm->streamCapMax;      // lorahip_demod_set_record_capacity (tests: resumed launches)
    capPkt = cap / 4 + 2;
}

//! summary of record set `set` into the object's books (waits for that step's kernel); its packets stay in the set until packed
static int pipeRead(lorahip_demod *dm, const int set)
{
    Pipe &P = pipeOf(dm);
    LORAHIP_TRY(hipEventSynchronize(P.ev[set]));
    P.pending[set] = false;
    const StreamSummary sum = P.hSum[set];
    dm->nNearSquelch += int64_t(unsigned(sum.nearSquelch - dm->nearSeen[0]));
    dm->nNearStep += int64_t(unsigned(sum.nearStep - dm->nearSeen[1]));
    dm->nearSeen[0] = sum.nearSquelch; dm->nearSeen[1] = sum.nearStep;
    dm->workCalls += sum.calls;
    dm->lastSum = sum;
    P.nPk[set] = size_t(sum.packets);
    P.nCalls[set] = sum.calls;
    P.nSig[set] = P.sigs[set] ? size_t(sum.signals) : 0;
    P.held[set] = true;
    return LORAHIP_OK;
}

static bool rowsHold(const lorahip_packet_rows *rows, const size_t n)
{
    return n == 0 || (rows != nullptr && rows->syms_dev != nullptr && rows->nsyms_dev != nullptr && rows->sym_stride != 0 && rows->sym_stride <= 0x7fffffffu &&
                      rows->cap_packets >= n);
}

//! do the registered signal rows hold n signals? (nothing registered: the signals are dropped, as the header says)
static bool sigRowsHold(const lorahip_demod *dm, const size_t n) { return !dm->sigRowsOn || n <= dm->sigRows.cap; }

//! the packets held in record set `set` into `rows` from row `firstRow` on, its signals into the signal rows from `firstSig` on
//! (stream-ordered; no wait); the set is free afterwards. The scratch (growDense) has been sized by the caller.
static int pipePack(lorahip_demod *dm, const int set, const lorahip_packet_rows *rows, const size_t firstRow, const size_t firstSig, const bool beside)
{
    Pipe &P = pipeOf(dm);
    lorahip_ctx *ctx = dm->ctx;
    const size_t n = P.nPk[set], ns = dm->sigRowsOn ? P.nSig[set] : 0;
    if (n == 0 && ns == 0) { P.held[set] = false; return LORAHIP_OK; }
    const StreamLayout &L = P.lay[set];
    const size_t nbRow = align256(L.B * sizeof(int));
    char *d = P.dev[set];
    // `beside`: packed on the side stream WHILE the step just launched runs (the host has just waited for this step's summary: its
    // kernel is complete). The side stream first waits for where the launch stream stood when this call began -- whatever the caller
    // queued there to read the rows of the call before (a decoder) and any earlier packing, which shares the scratch -- and the launch
    // stream then waits for the packing before anything later: the next kernel reuses this record set, the caller reads the rows.
    // Worth it for short steps only (profiles/r04/s37_*: 8-window chunks at SF7 +14 %; at 128-window chunks the packing kernels
    // displace workgroups of a streaming grid that exactly fills the device, -5 %).
    hipStream_t packStream = beside ? P.side : ctx->stream;
    if (beside) LORAHIP_TRY(hipStreamWaitEvent(P.side, P.entry, 0));
    if (n)
        LORAHIP_TRY(launchPackPackets(reinterpret_cast<const StreamPacket *>(d + L.oPkt), reinterpret_cast<const int *>(d + L.oNPkt),
                                      reinterpret_cast<const short *>(d + L.oSym), reinterpret_cast<int *>(dm->dDense), L.B, int(L.symStride), int(L.capPkt), n,
                                      reinterpret_cast<long long *>(dm->dDense + nbRow), rows->syms_dev + firstRow * rows->sym_stride, int(rows->sym_stride),
                                      rows->nsyms_dev + firstRow, rows->channel_dev ? rows->channel_dev + firstRow : nullptr, packStream));
    if (ns)
        LORAHIP_TRY(launchPackSignals(reinterpret_cast<const StreamSignal *>(d + L.oSig), reinterpret_cast<const int *>(d + L.oNSig), L.B, int(L.capPkt),
                                      dm->sigRows.channel, dm->sigRows.error, dm->sigRows.power, dm->sigRows.snr, firstSig, dm->sigRows.cap, packStream));
    if (beside)
    {
        LORAHIP_TRY(hipEventRecord(P.packDone, P.side));
        LORAHIP_TRY(hipStreamWaitEvent(ctx->stream, P.packDone, 0));
    }
    P.held[set] = false;                              // only now: a launch that failed above leaves the packets where they are
    return LORAHIP_OK;
}

/*! Every step whose summary can be read (oldest first) into `rows`, or -- if they do not all fit -- NONE of them: *nPackets = the
 * rows needed, LORAHIP_E_INVALID, the packets stay in their record sets and the next call (with rows that hold them) delivers
 * them. `sets`: number of record sets to consider, oldest first (1: only the older one). The same holds for the signals and the
 * rows registered for them (lorahip_demod_receive_signal_rows): all or nothing, lorahip_demod_receive_num_signals() = what is due. */
static int pipeDeliverHeld(lorahip_demod *dm, const int older, const int sets, const lorahip_packet_rows *rows, size_t *nPackets, int64_t *calls, const bool beside)
{
    Pipe &P = pipeOf(dm);
    size_t need = 0, needSig = 0, most = 0;
    for (int i = 0; i < sets; i++)
        if (P.held[older ^ i]) { need += P.nPk[older ^ i]; needSig += P.nSig[older ^ i]; if (P.nPk[older ^ i] > most) most = P.nPk[older ^ i]; }
    if (nPackets) *nPackets = need;
    dm->lastSignals = dm->sigRowsOn ? needSig : 0;
    if (!rowsHold(rows, need))
    {
        setLastError("lorahip_demod_receive (pipelined): the rows cannot hold the packets that are due; they are kept -- call again with rows for *n_packets");
        return LORAHIP_E_INVALID;
    }
    if (!sigRowsHold(dm, needSig))
    {
        setLastError("lorahip_demod_receive (pipelined): the signal rows cannot hold the signals that are due; they are kept -- register rows for lorahip_demod_receive_num_signals()");
        return LORAHIP_E_INVALID;
    }
    // the scratch for the largest set, before any state is touched: a failure here delivers nothing and loses nothing
    if (most) { const int grc = growDense(dm, align256(dm->B * sizeof(int)) + most * sizeof(long long)); if (grc != LORAHIP_OK) return grc; }
    size_t at = 0, atSig = 0;
    int64_t c = 0;
    for (int i = 0; i < sets; i++)
    {
        const int set = older ^ i;
        if (!P.held[set]) continue;
        const size_t n = P.nPk[set], ns = dm->sigRowsOn ? P.nSig[set] : 0;
        const int rc = pipePack(dm, set, rows, at, atSig, beside);
        if (rc != LORAHIP_OK) return rc;
        c += P.nCalls[set];
        at += n; atSig += ns;
    }
    if (calls) *calls = c;
    return LORAHIP_OK;
}

//! leave the pipeline: the packets not delivered yet into `rows` (nullable: they are dropped), the object back in the state a streaming run leaves
static int pipeFlush(lorahip_demod *dm, const lorahip_packet_rows *rows, size_t *nPackets, int64_t *calls)
{
    Pipe &P = pipeOf(dm);
    if (nPackets) *nPackets = 0;
    if (calls) *calls = 0;
    if (!P.active) return LORAHIP_OK;
    const DeviceGuard guard(dm->ctx->device);
    const int last = int((P.k - 1) & 1);
    for (int i = 1; i >= 0; i--)                      // oldest first: the books follow the steps in order
        if (P.k > 0 && P.pending[last ^ i]) { const int rc = pipeRead(dm, last ^ i); if (rc != LORAHIP_OK) return rc; }
    size_t n1 = 0;
    int64_t c1 = 0;
    dm->lastSignals = 0;
    if (rows == nullptr) P.held[0] = P.held[1] = false;                       // dropped on request
    else
    {
        LORAHIP_TRY(hipEventRecord(P.entry, dm->ctx->stream));
        const int rc = pipeDeliverHeld(dm, last ^ 1, 2, rows, &n1, &c1, false);
        if (nPackets) *nPackets = n1;
        if (rc != LORAHIP_OK) return rc;              // (the pipeline stays entered: flush again with rows that hold *n_packets)
    }
    if (calls) *calls = c1;
    LORAHIP_TRY(hipStreamSynchronize(dm->ctx->stream));
    P.active = false;
    dm->kernelMs = 0.0;                               // (the pipelined steps are not timed one by one: an event pair per step is two more calls)
    // what a streaming run leaves: the state (and the open packets' symbols) on the device, the pinned copy and the mirrors behind
    dm->devStateFresh = true; dm->posOnDevice = true; dm->mirrorsStale = true; dm->headStale = true;
    dm->devCarryValid = true; dm->hostCarryStale = true;
    pendingOf(dm).valid = false;
    return LORAHIP_OK;
}

static int pipeStep(lorahip_demod *dm, const float *iqDev, const size_t rowStride, const size_t nValid, const lorahip_packet_rows *rows, size_t *nPackets,
                    int64_t *calls, bool &handled)
{
    handled = false;
    Pipe &P = pipeOf(dm);
    lorahip_ctx *ctx = dm->ctx;
    const size_t N = dm->N, B = dm->B;
    const bool stream = dm->mode == 1 || (dm->mode == 0 && streamAvailable(ctx->sf));
    // What the pipeline needs in place: the streaming mode, no trace / ports, the state and the open packets on the device
    // with carry rows long enough, a continuing append stream. Anything else takes the ordinary step (which establishes exactly that).
    const bool compatible = stream && !dm->tracing && !dm->portsOn && !dm->activatePending && dm->sDev != nullptr &&
                            dm->append && !dm->appendFresh && dm->uniStride == rowStride && nValid >= dm->appendPrev &&
                            dm->dCarry != nullptr && dm->mtu + 1 <= dm->carryCap;
    const bool ready = compatible && dm->devStateFresh && (dm->devCarryValid || !dm->lastSum.anyOpen) && !pendingOf(dm).valid;
    if (!P.active && !ready) return LORAHIP_OK;
    if (P.active && !compatible) { setLastError("lorahip_demod_receive (pipelined): a setting or the rows changed under a running pipeline (lorahip_demod_receive_flush first)"); return LORAHIP_E_INVALID; }
    handled = true;
    const DeviceGuard guard(ctx->device);
    if (!P.active)
    {
        if (P.hSum == nullptr)
        {
            LORAHIP_TRY(hipHostMalloc((void **)&P.hSum, 2 * sizeof(StreamSummary), hipHostMallocMapped));
            for (int i = 0; i < 2; i++) LORAHIP_TRY(hipEventCreateWithFlags(&P.ev[i], hipEventDisableTiming));
            LORAHIP_TRY(hipEventCreateWithFlags(&P.packDone, hipEventDisableTiming));
            LORAHIP_TRY(hipEventCreateWithFlags(&P.entry, hipEventDisableTiming));
            LORAHIP_TRY(hipStreamCreateWithFlags(&P.side, hipStreamNonBlocking));
        }
        P.active = true; P.k = 0; P.pending[0] = P.pending[1] = false; P.held[0] = P.held[1] = false;
        dm->devStateFresh = false;                    // until the pipeline is flushed, only it knows where the state stands
    }
    const int set = int(P.k & 1);
    if (nPackets) *nPackets = 0;
    if (calls) *calls = 0;
    // where the launch stream stands now: behind whatever the caller queued to read the rows of the call before (pipePack)
    LORAHIP_TRY(hipEventRecord(P.entry, ctx->stream));
    bool delivered = false;
    dm->lastSignals = 0;
    if (P.held[set] || P.held[set ^ 1])
    {
        // A call before could not hand over the packets of a step (rows too small). If they sit in THIS record set the kernel about to
        // be launched would overwrite them; if in the other one (a flush that failed on small rows left the last step's there) they are
        // simply older than anything this call produces. Either way they are due now, oldest first, together with the packets of a step
        // launched since -- or nothing is launched and nothing is lost (the caller comes back with rows for *n_packets; the samples
        // wait in its array).
        const int older = P.held[set] ? set : (set ^ 1);
        if (P.pending[older ^ 1]) { const int rc = pipeRead(dm, older ^ 1); if (rc != LORAHIP_OK) return rc; }
        const int rc = pipeDeliverHeld(dm, older, 2, rows, nPackets, calls, false);
        if (rc != LORAHIP_OK) return rc;
        delivered = true;
    }
    const size_t prevValid = dm->appendPrev;
    size_t cap, capPkt;
    streamCapacity(dm, nValid - dm->appendPrev + 2 * N, cap, capPkt);
    StreamLayout L;
    L.make(B, cap, capPkt, false, dm->carryCap, dm->wantSignals);
    if (L.total > P.bytes[set])
    {
        // (this set's last use, step k - 2, was packed during step k - 1's call, stream-ordered before kernel k - 1: freeing waits for it)
        if (P.dev[set]) { (void)hipFree(P.dev[set]); P.dev[set] = nullptr; P.bytes[set] = 0; }
        const size_t want = L.total + L.total / 4;
        LORAHIP_TRY(hipMalloc((void **)&P.dev[set], want));
        P.bytes[set] = want;
    }
    P.lay[set] = L;
    const StreamLayout H = headLayout(dm);
    char *d = P.dev[set];
    StreamArgs a;
    a.iq = reinterpret_cast<const float2 *>(iqDev);
    a.base = nullptr; a.len = nullptr;
    a.uniformLen = (long long)nValid; a.uniformStride = (long long)rowStride;
    a.flags = 4 | 8;                                  // continue the streams; open packets in from / out to the carry rows
    a.carry = dm->dCarry; a.carryCap = int(dm->carryCap); a.maxBlocks = dm->streamGrid; a.lanes = lanesArg(dm); a.lastRoundFrom = 0;
    a.state = reinterpret_cast<StreamState *>(dm->sDev + H.oState);          // the object's own: every launch continues it
    a.nCalls = reinterpret_cast<int *>(d + L.oN); a.nSym = reinterpret_cast<int *>(d + L.oNSym); a.nPkt = reinterpret_cast<int *>(d + L.oNPkt);
    a.nSig = reinterpret_cast<int *>(d + L.oNSig); a.end = reinterpret_cast<int2 *>(d + L.oEnd);
    a.pktOut = reinterpret_cast<StreamPacket *>(d + L.oPkt); a.symOut = reinterpret_cast<short *>(d + L.oSym);
    // the block's signals (:267-269), kept per step like the packets and delivered with them one step late
    a.sigOut = dm->wantSignals ? reinterpret_cast<StreamSignal *>(d + L.oSig) : nullptr; a.calls = nullptr;
    P.sigs[set] = dm->wantSignals;
    a.down = ctx->dDown; a.fine = ctx->dFine; a.twStage = ctx->dTwStage;
    a.fineA = ctx->fineGather ? nullptr : ctx->dFineA; a.fineB = ctx->fineGather ? nullptr : ctx->dFineB;
    a.nChannels = unsigned(B); a.cap = int(cap); a.symStride = int(L.symStride); a.capPkt = int(capPkt);
    a.powerScale = ctx->powerScale; a.thresh = dm->thresh; a.sync = dm->sync;
    a.mtu = dm->mtu > 0xffffffffu ? 0xffffffffu : unsigned(dm->mtu);
    a.near = reinterpret_cast<unsigned *>(dm->sDev + H.oNear);
    // four calls per step: the kernel, its summary (written straight into pinned host memory), the event the next call waits on -- and
    // the previous step's packing below. (The summary on a stream of its own, behind the kernel's event and beside the NEXT step's kernel
    // -- it reads only this record set's counts and end words -- was measured: two more API calls and the hand-over between the streams
    // cost more than the 11 us the next kernel would no longer queue behind, 93 -> 108 us per 8-window step at SF7; profiles/r05/s36_*.)
    LORAHIP_TRY(launchStream(ctx->sf, a, ctx->stream));
    LORAHIP_TRY(launchStreamSummary(a.end, a.nCalls, a.nSym, a.nPkt, dm->wantSignals ? a.nSig : nullptr, B, int(cap), int(capPkt), a.near, dm->dSumScratch, &P.hSum[set], ctx->stream));
    LORAHIP_TRY(hipEventRecord(P.ev[set], ctx->stream));
    P.pending[set] = true;
    P.k++;
    P.iq = iqDev; P.rowStride = rowStride;
    dm->uniform = true; dm->uniSpc = nValid; dm->uniStride = rowStride; dm->appendPrev = nValid; dm->geomApplied = false;
    dm->mirrorsStale = true; dm->headStale = true;
    // ... and while it runs: the step before
    if (delivered || P.k < 2 || !P.pending[set ^ 1]) return LORAHIP_OK;
    { const int rc = pipeRead(dm, set ^ 1); if (rc != LORAHIP_OK) return rc; }
    const bool shortStep = (nValid - prevValid) <= 48 * N;    // (the step launched above)
    return pipeDeliverHeld(dm, set ^ 1, 1, rows, nPackets, calls, shortStep);
}

/***********************************************************************
 * The resident receiver: the host side. A step is a message copied into the ring in device memory (the doorbell) and, one call later,
 * two words read from pinned memory. Packets and signals are written by the kernel itself into the rows that came WITH the step's
 * message, i.e. with the call that rang it; they are complete when the next call (or the flush) returns.
 **********************************************************************/
static const double kResidentTimeoutS = 5.0;                // host: longest wait for a step's report
static const unsigned long long kResidentWatchdog = 800000000ull;   // device: 8 s of 100 MHz ticks without a message, then the wavefront leaves

static int residentRing(lorahip_demod *dm, const size_t nValid, const lorahip_packet_rows *rows, const unsigned flags)
{
    Pipe::Resident &R = pipeOf(dm).res;
    const unsigned seq = R.seq + 1;
    ResidentMsg m;
    std::memset(&m, 0, sizeof(m));
    m.nValid = nValid;
    if (rows)
    {
        m.syms = rows->syms_dev; m.nsyms = rows->nsyms_dev; m.chan = rows->channel_dev;
        m.symStride = unsigned(rows->sym_stride); m.capRows = unsigned(rows->cap_packets > 0xffffffu ? 0xffffffu : rows->cap_packets);
    }
    if (R.sigs && dm->sigRowsOn)
    {
        m.sigCh = dm->sigRows.channel; m.sigErr = dm->sigRows.error; m.sigPow = dm->sigRows.power; m.sigSnr = dm->sigRows.snr;
        m.capSig = unsigned(dm->sigRows.cap > 0xffffffu ? 0xffffffu : dm->sigRows.cap);
    }
    m.flags = flags;
    m.seq = seq;
    m.check = residentCheck(m);
    // plain stores into pinned memory, the step number last (the kernel verifies the check word whenever it sees the number)
    ResidentMsg *slot = &R.host->msg[seq & 7];
    __atomic_store_n(&slot->seq, 0u, __ATOMIC_RELEASE);
    std::memcpy(slot, &m, offsetof(ResidentMsg, seq));
    __atomic_store_n(&slot->seq, seq, __ATOMIC_RELEASE);
    R.seq = seq;
    return LORAHIP_OK;
}

//! wait for step `k`'s report (bounded); *packets / *signals = what the step produced (dropped ones included), *flags = RES_F_*
static int residentReport(lorahip_demod *dm, const unsigned k, size_t *packets, size_t *signals, int64_t *calls, unsigned *flags)
{
    Pipe::Resident &R = pipeOf(dm).res;
    volatile unsigned long long *h = R.host->sum + 2 * (k & 7);
    typedef std::chrono::steady_clock Clock;
    const Clock::time_point t0 = Clock::now();
    unsigned spins = 0;
    for (;;)
    {
        const unsigned long long w0 = __atomic_load_n(const_cast<unsigned long long *>(h), __ATOMIC_ACQUIRE);
        const unsigned long long w1 = __atomic_load_n(const_cast<unsigned long long *>(h + 1), __ATOMIC_ACQUIRE);
        if (unsigned(w0 >> 32) == k && unsigned(w1 >> 56) == (k & 0xffu))
        {
            R.dbgReports++;
            if (spins == 0) R.dbgImmediate++;
            else R.dbgWaitNs += std::chrono::duration<double, std::nano>(Clock::now() - t0).count();
            *calls = int64_t(w0 & 0xffffffffull);
            *packets = size_t(w1 & 0xffffffull);
            *signals = size_t((w1 >> 24) & 0xffffffull);
            *flags = unsigned((w1 >> 48) & 0xffull);
            return LORAHIP_OK;
        }
        if ((++spins & 1023u) == 0 && std::chrono::duration<double>(Clock::now() - t0).count() > kResidentTimeoutS) break;
    }
    setLastError("lorahip_demod_receive (resident): no report for a step within 5 s -- the kernel is told to leave");
    return LORAHIP_E_HIP;
}

//! get the kernel off the device whatever state the steps are in (error paths, destroy)
static void residentAbort(lorahip_demod *dm)
{
    Pipe::Resident &R = pipeOf(dm).res;
    if (!R.active) return;
    __atomic_store_n(&R.host->abort, 1u, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(R.run);
    R.active = false; R.unavailable = true;
    dm->devStateFresh = false;                        // the steps' bookkeeping is incomplete: nothing on the device is trusted
}

static int residentFlush(lorahip_demod *dm, size_t *nPackets, int64_t *calls)
{
    Pipe::Resident &R = pipeOf(dm).res;
    if (nPackets) *nPackets = 0;
    if (calls) *calls = 0;
    dm->lastSignals = 0;
    if (!R.active) return LORAHIP_OK;
    const DeviceGuard guard(dm->ctx->device);
    size_t pk = 0, sg = 0;
    int64_t cl = 0;
    unsigned fl = 0;
    bool lost = false;
    R.nRep = 0;
    while (R.reported < R.seq)
    {
        size_t p1 = 0, s1 = 0; int64_t c1 = 0;
        const int rc = residentReport(dm, R.reported + 1, &p1, &s1, &c1, &fl);
        if (rc != LORAHIP_OK) { residentAbort(dm); return rc; }
        R.reported++;
        if (R.nRep <= RES_DEPTH_MAX) { R.repPk[R.nRep] = p1; R.repSg[R.nRep] = dm->sigRowsOn ? s1 : 0; R.nRep++; }
        pk += p1; sg += s1; cl += c1;
        dm->workCalls += c1;
        lost = lost || (fl & (RES_F_PKT_OVERFLOW | RES_F_SIG_OVERFLOW));
        R.lastMore = (fl & RES_F_MORE) != 0;
    }
    // the quit message, then the kernel's end: the state, the read positions and the open packets are on the device as a streaming run leaves them
    { const int rc = residentRing(dm, R.lastValid, nullptr, 1u); if (rc != LORAHIP_OK) { residentAbort(dm); return rc; } }
    LORAHIP_TRY(hipStreamSynchronize(R.run));
    R.active = false;
    if (nPackets) *nPackets = pk;
    if (calls) *calls = cl;
    dm->lastSignals = dm->sigRowsOn ? sg : 0;
    dm->kernelMs = 0.0;
    dm->devStateFresh = true; dm->posOnDevice = true; dm->mirrorsStale = true; dm->headStale = true;
    dm->devCarryValid = true; dm->hostCarryStale = true;
    pendingOf(dm).valid = false;
    std::memset(&dm->lastSum, 0, sizeof(dm->lastSum));
    dm->lastSum.anyOpen = 1;                          // (not tracked per step: the carry rows are valid, which is all `anyOpen` guards)
    dm->lastSum.more = (R.lastMore || R.tail) ? 1 : 0;
    if (std::getenv("LORAHIP_RESIDENT_DEBUG"))
    {
        std::vector<char> cbuf(sizeof(ResidentCtl));
        ResidentCtl &c = *reinterpret_cast<ResidentCtl *>(cbuf.data());
        if (hipMemcpy(&c, R.ctl, sizeof(c), hipMemcpyDeviceToHost) == hipSuccess)
            for (int k = 0; k < 8; k++)                     // (slot k holds the last step with (step - 1) & 7 == k; the wait stamp of slot k is of the step after)
                std::fprintf(stderr, "resident slot %d (workgroup 0, wave 0; us): waited %.1f, setup %.1f, windows + records %.1f, look-ahead + step end %.1f; since the end of the slot before %.1f\n", k + 1,
                             (c.dbg[k][1] - c.dbg[k][0]) / 100.0, (c.dbg[k][2] - c.dbg[k][1]) / 100.0, (c.dbg[k][4] - c.dbg[k][2]) / 100.0,
                             (c.dbg[k][5] - c.dbg[k][4]) / 100.0, (double(c.dbg[k][5]) - double(c.dbg[(k + 7) & 7][5])) / 100.0);
        {
            // every wavefront's stamps of the chosen step: when the step began (the first wavefront that saw the message), and how the
            // wavefronts' finishing times -- windows done, step end -- are spread behind it
            const unsigned nw = R.grid * 4u < 16384u ? R.grid * 4u : 16384u;
            std::vector<double> seen, done, end, busy;
            unsigned long long first = ~0ull;
            for (unsigned w = 0; w < nw; w++) if (c.dbgWave[w][1] && c.dbgWave[w][1] < first) first = c.dbgWave[w][1];
            for (unsigned w = 0; w < nw; w++)
                if (c.dbgWave[w][1])
                {
                    seen.push_back((c.dbgWave[w][1] - first) / 100.0); done.push_back((c.dbgWave[w][2] - first) / 100.0); end.push_back((c.dbgWave[w][3] - first) / 100.0);
                    busy.push_back((c.dbgWave[w][2] - c.dbgWave[w][1]) / 100.0);
                }
            {
                // who the late ones are: the relay wavefronts (wavefront 0 of workgroups 0..7), and the twelve that ended the step last
                std::vector<std::pair<double, unsigned>> byEnd;
                for (unsigned w = 0; w < nw; w++) if (c.dbgWave[w][1]) byEnd.push_back(std::make_pair((c.dbgWave[w][3] - first) / 100.0, w));
                std::sort(byEnd.begin(), byEnd.end());
                const auto show = [&](const unsigned w, const char *tag)
                {
                    size_t rank = 0;
                    for (; rank < byEnd.size() && byEnd[rank].second != w; rank++) {}
                    std::fprintf(stderr, "resident wavefront %5u (workgroup %4u wave %u) %s: waiting since %.1f, message seen %.1f, windows + records done %.1f, step end %.1f (rank %zu of %zu)\n", w, w / 4u, w % 4u, tag,
                                 (double(c.dbgWave[w][0]) - double(first)) / 100.0, (c.dbgWave[w][1] - first) / 100.0, (c.dbgWave[w][2] - first) / 100.0, (c.dbgWave[w][3] - first) / 100.0, rank + 1, byEnd.size());
                };
                for (unsigned g = 0; g < 8u && g * 4u < nw; g++) show(g * 4u, "relay");
                for (size_t i = byEnd.size() > 12 ? byEnd.size() - 12 : 0; i < byEnd.size(); i++) show(byEnd[i].second, "late ");
                for (size_t i = 0; i < 4 && i < byEnd.size(); i++) show(byEnd[i].second, "early");
            }
            const auto pct = [](std::vector<double> &v, const double p) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[size_t(p * double(v.size() - 1))]; };
            const char *what[4] = {"message seen", "windows + records done", "step end", "(windows + records alone)"};
            std::vector<double> *vs[4] = {&seen, &done, &end, &busy};
            for (int q = 0; q < 4; q++)
                std::fprintf(stderr, "resident step of %zu wavefronts, us after the first one saw the message: %-28s min %.1f  10%% %.1f  50%% %.1f  90%% %.1f  99%% %.1f  max %.1f\n", vs[q]->size(),
                             what[q], pct(*vs[q], 0.0), pct(*vs[q], 0.1), pct(*vs[q], 0.5), pct(*vs[q], 0.9), pct(*vs[q], 0.99), pct(*vs[q], 1.0));
        }
        std::fprintf(stderr, "resident host: %llu reports, %llu there at the first look, %.1f us waited per report on average; %.1f us between two receive calls on average\n",
                     (unsigned long long)R.dbgReports, (unsigned long long)R.dbgImmediate, R.dbgReports ? R.dbgWaitNs / 1e3 / double(R.dbgReports) : 0.0,
                     R.dbgCalls > 1 ? R.dbgBetweenNs / 1e3 / double(R.dbgCalls - 1) : 0.0);
    }
    {
        // the kernels' running near-threshold counters
        const StreamLayout H = headLayout(dm);
        unsigned near[2] = {0, 0};
        LORAHIP_TRY(hipMemcpy(near, dm->sDev + H.oNear, sizeof(near), hipMemcpyDeviceToHost));
        dm->nNearSquelch += int64_t(unsigned(near[0] - dm->nearSeen[0])); dm->nNearStep += int64_t(unsigned(near[1] - dm->nearSeen[1]));
        dm->nearSeen[0] = near[0]; dm->nearSeen[1] = near[1];
    }
    if (lost)
    {
        setLastError("lorahip_demod_receive (resident): rows too small for a step's packets or signals -- the excess was dropped (counted in *n_packets)");
        return LORAHIP_E_INVALID;
    }
    return LORAHIP_OK;
}

/*! One step of the resident receiver. handled = false: the resident mode is not in place (yet) or not to be had for this object -- the
 * caller takes an ordinary step. *nPackets / *calls / lastSignals are those of the PREVIOUS step (0 for the first), whose rows -- the
 * ones that came with the previous call -- are complete now. */
static int residentStep(lorahip_demod *dm, const float *iqDev, const size_t rowStride, const size_t nValid, const lorahip_packet_rows *rows, size_t *nPackets,
                        int64_t *calls, bool &handled)
{
    handled = false;
    Pipe &P = pipeOf(dm);
    Pipe::Resident &R = P.res;
    lorahip_ctx *ctx = dm->ctx;
    const size_t N = dm->N, B = dm->B;
    const bool stream = dm->mode == 1 || (dm->mode == 0 && streamAvailable(ctx->sf));
    const bool compatible = stream && ctx->sf >= 7 && ctx->sf <= 12 && !dm->tracing && !dm->portsOn && !dm->activatePending && dm->sDev != nullptr &&
                            dm->append && !dm->appendFresh && dm->uniStride == rowStride && nValid >= dm->appendPrev && dm->dCarry != nullptr &&
                            dm->mtu + 1 <= dm->carryCap && !P.active && rowsHold(rows, 1) &&
                            // rows of whole 128-byte lines: a step then never reads a line that holds samples which arrive later (no cache to invalidate)
                            (rowStride & 15u) == 0 && (reinterpret_cast<uintptr_t>(iqDev) & 127u) == 0;
    if (R.active && (!compatible || iqDev != P.iq))
    {
        setLastError("lorahip_demod_receive (resident): a setting or the rows changed under the resident kernel (lorahip_demod_receive_flush first)");
        return LORAHIP_E_INVALID;
    }
    if (!R.active)
    {
        const bool ready = compatible && !R.unavailable && dm->devStateFresh && (dm->devCarryValid || !dm->lastSum.anyOpen) && !pendingOf(dm).valid;
        if (!ready) return LORAHIP_OK;
        const DeviceGuard guard(ctx->device);
        if (R.ctl == nullptr)
        {
            LORAHIP_TRY(hipMalloc((void **)&R.ctl, sizeof(ResidentCtl)));
            LORAHIP_TRY(hipHostMalloc((void **)&R.host, sizeof(ResidentHost), hipHostMallocMapped));
            LORAHIP_TRY(hipStreamCreateWithFlags(&R.run, hipStreamNonBlocking));
            LORAHIP_TRY(hipEventCreateWithFlags(&R.ev, hipEventDisableTiming));
        }
        // a step's records: sized for a step as long as this one (a longer one fills them, stops the channel and the next step resumes it)
        size_t cap, capPkt;
        streamCapacity(dm, nValid - dm->appendPrev + 2 * N, cap, capPkt);
        StreamLayout L;
        L.make(B, cap, capPkt, false, dm->carryCap, dm->wantSignals);
        // (RES_RING sets of record arrays: a step writes set step & 3 -- StreamArgs::resRecStride)
        if (RES_RING * L.total > R.recBytes)
        {
            if (R.rec) { (void)hipFree(R.rec); R.rec = nullptr; R.recBytes = 0; }
            LORAHIP_TRY(hipMalloc((void **)&R.rec, RES_RING * L.total + L.total / 4));
            R.recBytes = RES_RING * L.total + L.total / 4;
        }
        R.lay = L;
        R.sigs = dm->wantSignals;
        void *hostDev = nullptr;
        LORAHIP_TRY(hipHostGetDevicePointer(&hostDev, R.host, 0));
        std::memset(R.host, 0, sizeof(ResidentHost));
        LORAHIP_TRY(hipMemsetAsync(R.ctl, 0, std::getenv("LORAHIP_RESIDENT_DEBUG") ? sizeof(ResidentCtl) : offsetof(ResidentCtl, dbgWave), ctx->stream));
        const StreamLayout H = headLayout(dm);
        char *d = R.rec;
        StreamArgs a;
        a.iq = reinterpret_cast<const float2 *>(iqDev);
        a.base = nullptr; a.len = nullptr;
        a.uniformLen = 0; a.uniformStride = (long long)rowStride;       // (the length of a step comes with its message)
        a.flags = 4 | 8;
        a.carry = dm->dCarry; a.carryCap = int(dm->carryCap); a.maxBlocks = 0; a.lanes = -1; a.lastRoundFrom = 0;
        a.state = reinterpret_cast<StreamState *>(dm->sDev + H.oState);
        a.nCalls = reinterpret_cast<int *>(d + L.oN); a.nSym = reinterpret_cast<int *>(d + L.oNSym); a.nPkt = reinterpret_cast<int *>(d + L.oNPkt);
        a.nSig = reinterpret_cast<int *>(d + L.oNSig); a.end = reinterpret_cast<int2 *>(d + L.oEnd);
        a.pktOut = reinterpret_cast<StreamPacket *>(d + L.oPkt); a.symOut = reinterpret_cast<short *>(d + L.oSym);
        a.sigOut = dm->wantSignals ? reinterpret_cast<StreamSignal *>(d + L.oSig) : nullptr; a.calls = nullptr;
        a.down = ctx->dDown; a.fine = ctx->dFine; a.twStage = ctx->dTwStage;
        a.fineA = ctx->fineGather ? nullptr : ctx->dFineA; a.fineB = ctx->fineGather ? nullptr : ctx->dFineB;
        a.nChannels = unsigned(B); a.cap = int(cap); a.symStride = int(L.symStride); a.capPkt = int(capPkt);
        a.powerScale = ctx->powerScale; a.thresh = dm->thresh; a.sync = dm->sync;
        a.mtu = dm->mtu > 0xffffffffu ? 0xffffffffu : unsigned(dm->mtu);
        a.near = reinterpret_cast<unsigned *>(dm->sDev + H.oNear);
        a.res = R.ctl; a.resHost = static_cast<ResidentHost *>(hostDev); a.resWatchdog = kResidentWatchdog; a.resRecStride = L.total;
        if (const char *e = std::getenv("LORAHIP_RESIDENT_DEBUG")) a.resDebug = unsigned(std::atoi(e));     // (the step whose stamps every wavefront leaves)
        if (const char *e = std::getenv("LORAHIP_RESIDENT_SLEEP")) a.resSleep = std::atoi(e);             // (measurements: profiles/r06)
        // behind everything queued on the launch stream (the state of the run before, the cleared control block)
        LORAHIP_TRY(hipEventRecord(R.ev, ctx->stream));
        LORAHIP_TRY(hipStreamWaitEvent(R.run, R.ev, 0));
        const hipError_t le = launchStreamResident(ctx->sf, a, R.run, &R.grid);
        if (le == hipErrorNotSupported) { (void)hipGetLastError(); R.unavailable = true; return LORAHIP_OK; }     // not for this geometry: ordinary steps
        LORAHIP_TRY(le);
        R.active = true; R.seq = 0; R.reported = 0; R.lastMore = false;
        R.dbgReports = R.dbgImmediate = R.dbgCalls = 0; R.dbgWaitNs = R.dbgBetweenNs = 0.0;
        R.depth = rows->reserved >= 1 ? (rows->reserved > RES_DEPTH_MAX ? unsigned(RES_DEPTH_MAX) : unsigned(rows->reserved)) : 1u;
        {
            // the census: every workgroup must be ON the device before a step is rung (bounded wait; otherwise the kernel is told to
            // leave and the object takes ordinary steps from now on)
            typedef std::chrono::steady_clock Clock;
            const Clock::time_point t0 = Clock::now();
            while (__atomic_load_n(&R.host->arrivedAll, __ATOMIC_ACQUIRE) == 0u && std::chrono::duration<double>(Clock::now() - t0).count() < 2.0) {}
            if (__atomic_load_n(&R.host->arrivedAll, __ATOMIC_ACQUIRE) == 0u)
            {
                residentAbort(dm);
                dm->devStateFresh = true;             // (no step was taken: the state on the device is what it was)
                return LORAHIP_OK;
            }
        }
        P.iq = iqDev; P.rowStride = rowStride;
        dm->devStateFresh = false;                    // until the kernel has left, only it knows where the state stands
    }
    handled = true;
    {
        const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
        if (R.dbgCalls++) R.dbgBetweenNs += std::chrono::duration<double, std::nano>(now - R.dbgLast).count();
        R.dbgLast = now;
    }
    if (nPackets) *nPackets = 0;
    if (calls) *calls = 0;
    dm->lastSignals = 0;
    R.nRep = 0;
    const DeviceGuard guard(ctx->device);
    // (up to the last whole line of every row: the < 16 samples behind it wait for the next step, or for the flush's ordinary step)
    { const int rc = residentRing(dm, nValid & ~size_t(15), rows, 0u); if (rc != LORAHIP_OK) { residentAbort(dm); return rc; } }
    R.lastValid = nValid & ~size_t(15);
    R.tail = (nValid & 15u) != 0;
    dm->uniform = true; dm->uniSpc = nValid; dm->uniStride = rowStride; dm->appendPrev = nValid; dm->geomApplied = false;
    dm->mirrorsStale = true; dm->headStale = true;
    if (R.seq <= R.depth) return LORAHIP_OK;
    // ... and while it runs: the step `depth` calls back
    size_t pk = 0, sg = 0;
    int64_t cl = 0;
    unsigned fl = 0;
    const int rc = residentReport(dm, R.seq - R.depth, &pk, &sg, &cl, &fl);
    if (rc != LORAHIP_OK) { residentAbort(dm); return rc; }
    R.reported = R.seq - R.depth;
    R.repPk[0] = pk; R.repSg[0] = dm->sigRowsOn ? sg : 0; R.nRep = 1;
    R.lastMore = (fl & RES_F_MORE) != 0;
    dm->workCalls += cl;
    if (nPackets) *nPackets = pk;
    if (calls) *calls = cl;
    dm->lastSignals = dm->sigRowsOn ? sg : 0;
    if (fl & (RES_F_PKT_OVERFLOW | RES_F_SIG_OVERFLOW))
    {
        setLastError("lorahip_demod_receive (resident): the rows of the call before were too small for its step -- the excess was dropped (counted in *n_packets)");
        return LORAHIP_E_INVALID;
    }
    return LORAHIP_OK;
}

/***********************************************************************
 * Level-3 debug ports: the block's "raw" / "dec" / "fft" outputs (LoRaDemod.cpp:81-83). The run records, per work() call, the
 * dechirp state it started from (lorahip_work_result.fine_*); the ports are then REPLAYED from that trace with the batch
 * kernels -- the same arithmetic on the same inputs, hence the same bits -- and scattered into the caller's per-channel
 * port streams. Off unless asked for: they triple the HBM traffic.
 **********************************************************************/
static int growPort(lorahip_demod *dm, const size_t bytes)
{
    if (bytes <= dm->dPortBytes) return LORAHIP_OK;
    if (dm->dPort) { (void)hipFree(dm->dPort); dm->dPort = nullptr; dm->dPortBytes = 0; }
    LORAHIP_TRY(hipMalloc((void **)&dm->dPort, bytes));
    dm->dPortBytes = bytes;
    return LORAHIP_OK;
}

static int fillPorts(lorahip_demod *dm, const float *iqDev)
{
    lorahip_ctx *ctx = dm->ctx;
    const size_t N = dm->N, B = dm->B;
    const lorahip_demod_ports &P = dm->ports;
    const DeviceGuard guard(ctx->device);
    applyGeometry(dm);                                          // the replay reads ch[].base
    struct Seg { long long src, dst; int len; };
    std::vector<int64_t> wOff; std::vector<int32_t> wSel, wIdx; std::vector<float> wErr;
    std::vector<Seg> segFft, segDec, segRaw;                      // src of fft/dec segments: window number * N (chunk-relative later)
    for (size_t c = 0; c < B; c++)
    {
        Channel &k = dm->ch[c];
        size_t decPos = 0, frame = 0;
        for (size_t i = k.traceStart; i < k.trace.size(); i++, frame++)
        {
            const lorahip_work_result &r = k.trace[i];
            const bool down = r.state_before == ST_DOWNCHIRP0 || r.state_before == ST_DOWNCHIRP1;   // _chirpTable == _downChirpTable
            const size_t w = wOff.size();
            wOff.push_back(int64_t(k.base + k.runStart + decPos));  // from where the run began (the head of the stream unless it continues one)
            wSel.push_back(down ? LORAHIP_CHIRP_DOWN : LORAHIP_CHIRP_UP);
            wIdx.push_back(r.fine_idx_before);
            wErr.push_back(r.fine_err_before);
            const size_t total = size_t(r.consumed);
            if (P.fft_dev && frame < P.fft_cap_frames) segFft.push_back({ (long long)(w * N), (long long)((c * P.fft_cap_frames + frame) * N), int(N) });
            if (P.dec_dev)
            {
                const size_t n0 = total < N ? total : N;
                if (decPos < P.dec_cap_samples)
                    segDec.push_back({ (long long)(w * N), (long long)(c * P.dec_cap_samples + decPos), int(decPos + n0 <= P.dec_cap_samples ? n0 : P.dec_cap_samples - decPos) });
            }
            if (total == 2 * N && r.state_before == ST_FRAMESYNC)
            {
                // the sync check dechirped window 1 too (:189-206): it starts from the committed index and is part of what `dec` produces
                const size_t w1 = wOff.size();
                wOff.push_back(int64_t(k.base + k.runStart + decPos + N));
                wSel.push_back(LORAHIP_CHIRP_UP);
                wIdx.push_back(r.fine_idx_after);
                wErr.push_back(r.fine_err_before);
                if (P.dec_dev && decPos + N < P.dec_cap_samples)
                    segDec.push_back({ (long long)(w1 * N), (long long)(c * P.dec_cap_samples + decPos + N),
                                       int(decPos + 2 * N <= P.dec_cap_samples ? N : P.dec_cap_samples - decPos - N) });
            }
            decPos += total;
        }
        k.portFft = frame; k.portDec = decPos; k.portRaw = decPos;
        if (P.raw_dev && decPos) segRaw.push_back({ (long long)(k.base + k.runStart), (long long)(c * P.raw_cap_samples), int(decPos <= P.raw_cap_samples ? decPos : P.raw_cap_samples) });
    }
    hipStream_t st = ctx->stream;
    auto scatter = [&](float *dst, const float *src, const std::vector<Seg> &segs, const size_t first, const size_t last, const long long srcBias, char *scratch) -> int
    {
        // segments [first, last) of `segs`, their src offsets rebased by srcBias
        const size_t n = last - first;
        if (n == 0) return LORAHIP_OK;
        std::vector<long long> so(n), dof(n); std::vector<int> ln(n);
        for (size_t i = 0; i < n; i++) { so[i] = segs[first + i].src - srcBias; dof[i] = segs[first + i].dst; ln[i] = segs[first + i].len; }
        long long *dSo = reinterpret_cast<long long *>(scratch), *dDo = dSo + n;
        int *dLn = reinterpret_cast<int *>(dDo + n);
        LORAHIP_TRY(hipMemcpyAsync(dSo, so.data(), n * sizeof(long long), hipMemcpyHostToDevice, st));
        LORAHIP_TRY(hipMemcpyAsync(dDo, dof.data(), n * sizeof(long long), hipMemcpyHostToDevice, st));
        LORAHIP_TRY(hipMemcpyAsync(dLn, ln.data(), n * sizeof(int), hipMemcpyHostToDevice, st));
        LORAHIP_TRY(launchCopySegments(reinterpret_cast<float2 *>(dst), reinterpret_cast<const float2 *>(src), dSo, dDo, dLn, n, st));
        LORAHIP_TRY(hipStreamSynchronize(st));                  // the host vectors die with this scope
        return LORAHIP_OK;
    };
    const size_t W = wOff.size();
    // replay in chunks: at most ~256 MiB of replayed windows per port at a time
    size_t chunk = (size_t(256) << 20) / (N * sizeof(cf32));
    if (chunk < 64) chunk = 64;
    if (chunk > W) chunk = W;
    const size_t descBytes = align256(chunk * 2 * (2 * sizeof(long long) + sizeof(int)) + segRaw.size() * (2 * sizeof(long long) + sizeof(int)) + 64);
    const size_t winBytes = align256(chunk * N * sizeof(cf32));
    const size_t inBytes = align256(chunk * sizeof(int64_t)) + 3 * align256(chunk * sizeof(int32_t));
    const size_t outBytes = align256(chunk * sizeof(uint16_t)) + 3 * align256(chunk * sizeof(float));
    { const int rc = growPort(dm, descBytes + 2 * winBytes + inBytes + outBytes); if (rc != LORAHIP_OK) return rc; }
    char *cur = dm->dPort;
    char *dDesc = cur; cur += descBytes;
    float *tFft = reinterpret_cast<float *>(cur); cur += winBytes;
    float *tDec = reinterpret_cast<float *>(cur); cur += winBytes;
    int64_t *dOff = reinterpret_cast<int64_t *>(cur); cur += align256(chunk * sizeof(int64_t));
    int32_t *dSel = reinterpret_cast<int32_t *>(cur); cur += align256(chunk * sizeof(int32_t));
    int32_t *dIdx = reinterpret_cast<int32_t *>(cur); cur += align256(chunk * sizeof(int32_t));
    float *dErr = reinterpret_cast<float *>(cur); cur += align256(chunk * sizeof(int32_t));
    uint16_t *dSym = reinterpret_cast<uint16_t *>(cur); cur += align256(chunk * sizeof(uint16_t));
    float *dPow = reinterpret_cast<float *>(cur); cur += align256(chunk * sizeof(float));
    float *dAvg = reinterpret_cast<float *>(cur); cur += align256(chunk * sizeof(float));
    float *dFi = reinterpret_cast<float *>(cur);
    if (P.raw_dev) { const int rc = scatter(P.raw_dev, iqDev, segRaw, 0, segRaw.size(), 0, dDesc); if (rc != LORAHIP_OK) return rc; }
    struct HostCopy
    {
        static int run(lorahip_demod *dm, hipStream_t st)
        {
            // host port buffers: what each channel produced, out of the library's device mirrors
            const lorahip_demod_ports &H = dm->hostPorts, &D = dm->ports;
            const size_t N = dm->N;
            for (size_t c = 0; c < dm->B; c++)
            {
                const Channel &k = dm->ch[c];
                const size_t nf = k.portFft < D.fft_cap_frames ? k.portFft : D.fft_cap_frames;
                const size_t nd = k.portDec < D.dec_cap_samples ? k.portDec : D.dec_cap_samples;
                const size_t nr = k.portRaw < D.raw_cap_samples ? k.portRaw : D.raw_cap_samples;
                if (H.fft_dev && nf) LORAHIP_TRY(hipMemcpyAsync(H.fft_dev + 2 * c * D.fft_cap_frames * N, D.fft_dev + 2 * c * D.fft_cap_frames * N, nf * N * sizeof(cf32), hipMemcpyDeviceToHost, st));
                if (H.dec_dev && nd) LORAHIP_TRY(hipMemcpyAsync(H.dec_dev + 2 * c * D.dec_cap_samples, D.dec_dev + 2 * c * D.dec_cap_samples, nd * sizeof(cf32), hipMemcpyDeviceToHost, st));
                if (H.raw_dev && nr) LORAHIP_TRY(hipMemcpyAsync(H.raw_dev + 2 * c * D.raw_cap_samples, D.raw_dev + 2 * c * D.raw_cap_samples, nr * sizeof(cf32), hipMemcpyDeviceToHost, st));
            }
            LORAHIP_TRY(hipStreamSynchronize(st));
            return LORAHIP_OK;
        }
    };
    size_t fi = 0, di = 0;
    for (size_t w0 = 0; w0 < W && (P.fft_dev || P.dec_dev); w0 += chunk)
    {
        const size_t n = W - w0 < chunk ? W - w0 : chunk;
        LORAHIP_TRY(hipMemcpyAsync(dOff, wOff.data() + w0, n * sizeof(int64_t), hipMemcpyHostToDevice, st));
        LORAHIP_TRY(hipMemcpyAsync(dSel, wSel.data() + w0, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
        LORAHIP_TRY(hipMemcpyAsync(dIdx, wIdx.data() + w0, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
        LORAHIP_TRY(hipMemcpyAsync(dErr, wErr.data() + w0, n * sizeof(float), hipMemcpyHostToDevice, st));
        lorahip_batch b;
        std::memset(&b, 0, sizeof(b));
        b.struct_size = sizeof(b);
        b.iq = iqDev; b.n_windows = n; b.offsets = dOff; b.chirp_sel = dSel; b.fine_idx0 = dIdx; b.fine_err = dErr;
        b.sym = dSym; b.power = dPow; b.power_avg = dAvg; b.f_index = dFi;
        b.fft_out = P.fft_dev ? tFft : nullptr;
        b.dec_out = P.dec_dev ? tDec : nullptr;
        const int rc = lorahip_detect_batch(ctx, &b);
        if (rc != LORAHIP_OK) return rc;
        const long long lo = (long long)(w0 * N), hi = (long long)((w0 + n) * N);
        size_t f1 = fi; while (f1 < segFft.size() && segFft[f1].src < hi) f1++;
        size_t d1 = di; while (d1 < segDec.size() && segDec[d1].src < hi) d1++;
        int rc2 = scatter(P.fft_dev, tFft, segFft, fi, f1, lo, dDesc);
        if (rc2 == LORAHIP_OK) rc2 = scatter(P.dec_dev, tDec, segDec, di, d1, lo, dDesc);
        if (rc2 != LORAHIP_OK) return rc2;
        fi = f1; di = d1;
    }
    if (dm->hostPorts.host_buffers) return HostCopy::run(dm, st);
    return LORAHIP_OK;
}

static int runAny(lorahip_demod *dm, const float *iqDev, int64_t *roundsOut)
{
    const bool stream = dm->mode == 1 || (dm->mode == 0 && streamAvailable(dm->ctx->sf));
    if (stream && !streamAvailable(dm->ctx->sf)) { setLastError("no streaming kernel for this SF"); return LORAHIP_E_INVALID; }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    { const int rc = drainPending(dm); if (rc != LORAHIP_OK) return rc; }       // the previous run's records, if still on the device
    // the port replay reads the per-call trace; a trace the caller did not ask for lives for this run only (it must neither grow
    // without bound in a long-running receiver nor show up in lorahip_demod_get_trace / _trace_len / _get_labels)
    const bool internalTrace = dm->portsOn && !dm->userTracing;
    const bool cont = dm->append && !dm->appendFresh;
    if (internalTrace || (dm->portsOn && cont)) { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; }
    if (internalTrace) for (auto &k : dm->ch) { k.trace.clear(); k.traceSymCount0 = k.symCount; }
    if (dm->portsOn) for (auto &k : dm->ch) k.runStart = cont ? k.pos : 0;      // where the port replay starts reading
    dm->tracing = dm->userTracing || dm->portsOn;
    if (dm->tracing || dm->portCountsDirty)
    {
        for (auto &k : dm->ch) { k.traceStart = k.trace.size(); k.portFft = k.portDec = k.portRaw = 0; }
        dm->portCountsDirty = dm->portsOn;                       // a run without ports leaves them at zero: nothing to reset next time
    }
    if (!stream) { dm->kernelMs = 0.0; dm->lastLaunches = 0; }       // what the accessors say of a host-driven run
    int rc = stream ? runStream(dm, iqDev, roundsOut) : runRounds(dm, iqDev, roundsOut);
    if (rc == LORAHIP_OK && dm->portsOn) rc = fillPorts(dm, iqDev);
    if (rc == LORAHIP_OK && dm->append) { dm->appendFresh = false; dm->appendPrev = dm->uniSpc; }    // the next append run continues this one
    if (internalTrace) for (auto &k : dm->ch) { std::vector<lorahip_work_result>().swap(k.trace); k.traceStart = 0; }
    return rc;
}

} // namespace

namespace lorahip {
void demodSetCoResidentWaves(lorahip_demod *dm, const unsigned waves) { if (dm != nullptr && dm->comp == nullptr) dm->coWaves = waves; }
}

extern "C" {

int lorahip_demod_create(lorahip_demod **out, const int device, const int sf, const size_t n_channels)
{
    if (out == nullptr || n_channels == 0 || n_channels > 0x7fffffffu) return LORAHIP_E_INVALID;
    *out = nullptr;
    lorahip_demod *dm = new (std::nothrow) lorahip_demod();
    if (dm == nullptr) return LORAHIP_E_NOMEM;
    dm->comp = nullptr;
    dm->ctx = nullptr; dm->h = nullptr; dm->d = nullptr; dm->dIq = nullptr; dm->dIqSamples = 0;
    dm->evK0 = nullptr; dm->evK1 = nullptr; dm->evJoin = nullptr; dm->kernelMs = 0.0; dm->dSumScratch = nullptr;
    dm->pending = new (std::nothrow) PendingLaunch();
    dm->pipe = new (std::nothrow) Pipe();
    if (dm->pending == nullptr || dm->pipe == nullptr) { delete static_cast<PendingLaunch *>(dm->pending); delete static_cast<Pipe *>(dm->pipe); delete dm; return LORAHIP_E_NOMEM; }
    pendingOf(dm).valid = false;
    std::memset(dm->pipe, 0, sizeof(Pipe));
    std::memset(&dm->ports, 0, sizeof(dm->ports)); dm->portsOn = false; dm->userTracing = false; dm->dPort = nullptr; dm->dPortBytes = 0;
    std::memset(&dm->hostPorts, 0, sizeof(dm->hostPorts)); dm->ownFft = dm->ownDec = dm->ownRaw = nullptr;
    dm->mode = 0; dm->sDev = nullptr; dm->sHost = nullptr; dm->sBytes = 0; dm->dDense = nullptr; dm->hDense = nullptr; dm->denseBytes = 0;
    int rc = lorahip_create(&dm->ctx, device, sf);
    if (rc != LORAHIP_OK) { const std::string e = lorahip_last_error(); lorahip_demod_destroy(dm); setLastError(e); return rc; }
    dm->N = size_t(1) << sf;
    dm->B = n_channels;
    dm->sync = 0x12; dm->thresh = -30.0f; dm->mtu = 256;        // LoRaDemod.cpp:71-73
    dm->tracing = false;
    dm->workCalls = 0;
    dm->nNearSquelch = dm->nNearStep = 0;
    dm->devStateFresh = false; dm->mirrorsStale = false; dm->activatePending = false;
    dm->uniform = false; dm->uniSpc = 0; dm->uniStride = 0; dm->geomApplied = true; dm->portCountsDirty = true; dm->posOnDevice = false;
    dm->append = false; dm->appendFresh = true; dm->appendPrev = 0; dm->headStale = false; dm->nearSeen[0] = dm->nearSeen[1] = 0;
    std::memset(&dm->lastSum, 0, sizeof(dm->lastSum));
    dm->wantSignals = false;
    std::memset(&dm->sigRows, 0, sizeof(dm->sigRows)); dm->sigRowsOn = false; dm->lastSignals = 0;
    dm->streamGrid = 0;
    dm->streamCapMax = 0;
    dm->streamLanes = 0;
    dm->coWaves = 0;
    dm->lastLaunches = 0;
    dm->dCarry = nullptr; dm->carryCap = 0; dm->devCarryValid = false; dm->hostCarryStale = false;
    dm->callsPerWindowQ8 = 288;                                 // 1.125 calls per N samples to begin with
    try { dm->ch.resize(n_channels); }
    catch (const std::bad_alloc &) { lorahip_demod_destroy(dm); return LORAHIP_E_NOMEM; }
    for (auto &k : dm->ch) { k.traceStart = 0; k.traceSymCount0 = 0; k.portFft = k.portDec = k.portRaw = 0; k.callCount = 0; k.runStart = 0; }
    dm->stageBytes = carve(nullptr, n_channels).total;
    bool staged;
    {
        const DeviceGuard guard(device);                    // lorahip_create() restored the caller's device: allocate on OURS
        staged = hipMalloc((void **)&dm->d, dm->stageBytes) == hipSuccess &&
                 hipHostMalloc((void **)&dm->h, dm->stageBytes, hipHostMallocDefault) == hipSuccess;
        // (more than 32768 channels: the summary of a streaming launch is reduced by one workgroup per 4096 of them and a second launch)
        if (staged && n_channels > 32768)
            staged = hipMalloc(&dm->dSumScratch, streamSummaryScratchBytes(n_channels)) == hipSuccess &&
                     hipMemset(dm->dSumScratch, 0, streamSummaryScratchBytes(n_channels)) == hipSuccess;
    }
    if (!staged)
    {
        lorahip_demod_destroy(dm);
        return LORAHIP_E_NOMEM;
    }
    lorahip_demod_activate(dm);
    *out = dm;
    return LORAHIP_OK;
}

int lorahip_demod_create_mixed(lorahip_demod **out, const int *devices, const size_t n_devices, const int32_t *channel_sf, const size_t n_channels)
{
    if (out == nullptr) return LORAHIP_E_INVALID;
    *out = nullptr;
    Composite *k = nullptr;
    const int rc = Composite::create(&k, devices, n_devices, channel_sf, n_channels);
    if (rc != LORAHIP_OK) return rc;
    lorahip_demod *dm = new (std::nothrow) lorahip_demod();         // value-initialised: every pointer null
    if (dm == nullptr) { delete k; return LORAHIP_E_NOMEM; }
    dm->comp = k;
    dm->B = n_channels;
    *out = dm;
    return LORAHIP_OK;
}

size_t lorahip_demod_num_channels(const lorahip_demod *dm) { return dm ? dm->B : 0; }
size_t lorahip_demod_num_parts(const lorahip_demod *dm) { return dm == nullptr ? 0 : (dm->comp ? dm->comp->numParts() : 1); }

int lorahip_demod_part(const lorahip_demod *dm, const size_t i, int32_t *device, int32_t *sf, size_t *n_channels, int32_t *device_slot)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->partInfo(i, device, sf, n_channels, device_slot);
    if (i != 0) return LORAHIP_E_INVALID;
    if (device) *device = dm->ctx->device;
    if (sf) *sf = dm->ctx->sf;
    if (n_channels) *n_channels = dm->B;
    if (device_slot) *device_slot = 0;
    return LORAHIP_OK;
}

int lorahip_demod_part_of(const lorahip_demod *dm, int32_t *part_of_channel, int32_t *local_channel)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->partOf(part_of_channel, local_channel);
    for (size_t c = 0; c < dm->B; c++) { if (part_of_channel) part_of_channel[c] = 0; if (local_channel) local_channel[c] = int32_t(c); }
    return LORAHIP_OK;
}

lorahip_demod *lorahip_demod_part_handle(const lorahip_demod *dm, const size_t i)
{
    if (dm == nullptr) return nullptr;
    if (dm->comp) return dm->comp->part(i);
    return i == 0 ? const_cast<lorahip_demod *>(dm) : nullptr;
}

void lorahip_demod_destroy(lorahip_demod *dm)
{
    if (dm == nullptr) return;
    if (dm->comp) { delete dm->comp; delete dm; return; }
    {
    const DeviceGuard guard(dm->ctx ? dm->ctx->device : 0);   // the caller's current device is restored on return
    if (dm->ctx) (void)hipStreamSynchronize(dm->ctx->stream);  // a pipelined step may still be writing into what is freed below
    if (dm->d) (void)hipFree(dm->d);
    if (dm->h) (void)hipHostFree(dm->h);
    if (dm->dIq) (void)hipFree(dm->dIq);
    if (dm->sDev) (void)hipFree(dm->sDev);
    if (dm->sHost) (void)hipHostFree(dm->sHost);
    if (dm->dCarry) (void)hipFree(dm->dCarry);
    if (dm->dDense) (void)hipFree(dm->dDense);
    if (dm->hDense) (void)hipHostFree(dm->hDense);
    if (dm->dPort) (void)hipFree(dm->dPort);
    if (dm->ownFft) (void)hipFree(dm->ownFft);
    if (dm->ownDec) (void)hipFree(dm->ownDec);
    if (dm->ownRaw) (void)hipFree(dm->ownRaw);
    if (dm->evK0) (void)hipEventDestroy(dm->evK0);
    if (dm->evK1) (void)hipEventDestroy(dm->evK1);
    if (dm->evJoin) (void)hipEventDestroy(dm->evJoin);
    if (dm->dSumScratch) (void)hipFree(dm->dSumScratch);
    if (dm->pipe)
    {
        Pipe &P = pipeOf(dm);
        residentAbort(dm);                                        // (a resident kernel must leave before its memory goes)
        if (P.res.ctl) (void)hipFree(P.res.ctl);
        if (P.res.host) (void)hipHostFree(P.res.host);
        if (P.res.rec) (void)hipFree(P.res.rec);
        if (P.res.run) (void)hipStreamDestroy(P.res.run);
        if (P.res.ev) (void)hipEventDestroy(P.res.ev);
        for (int i = 0; i < 2; i++)
        {
            if (P.dev[i]) (void)hipFree(P.dev[i]);
            if (P.hSum) (void)hipEventDestroy(P.ev[i]);
        }
        if (P.side) { (void)hipStreamSynchronize(P.side); (void)hipStreamDestroy(P.side); }
        if (P.packDone) (void)hipEventDestroy(P.packDone);
        if (P.entry) (void)hipEventDestroy(P.entry);
        if (P.hSum) (void)hipHostFree(P.hSum);
    }
    }
    lorahip_destroy(dm->ctx);
    delete static_cast<PendingLaunch *>(dm->pending);
    delete static_cast<Pipe *>(dm->pipe);
    delete dm;
}

int lorahip_demod_set_sync(lorahip_demod *dm, const unsigned char sync)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setSync(sync);
    dm->sync = sync;
    return LORAHIP_OK;
}

int lorahip_demod_set_threshold(lorahip_demod *dm, const double thresh_dB)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setThreshold(thresh_dB);
    dm->thresh = float(thresh_dB);
    return LORAHIP_OK;
}

int lorahip_demod_set_mtu(lorahip_demod *dm, const size_t mtu)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setMtu(mtu);
    dm->mtu = mtu;
    return LORAHIP_OK;
}

int lorahip_demod_set_mode(lorahip_demod *dm, const int mode)
{
    if (dm == nullptr || mode < 0 || mode > 2) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setMode(mode);
    dm->mode = mode;
    return LORAHIP_OK;
}

// (A streaming run leaves nothing in flight: the open packets' symbols are saved by the streaming kernel itself and the run ends with
// its stream drained, so moving the object to another stream needs no ordering. A PIPELINED step is in flight by design: refused.)
int lorahip_demod_set_stream(lorahip_demod *dm, void *hip_stream)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setStream(hip_stream);
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }     // the step in flight is ordered on the stream it was launched on
    return lorahip_set_stream(dm->ctx, hip_stream);
}

// What has been queued on the object's launch stream so far -- packing kernels of an async receive, a pipelined step -- before
// anything queued on `hip_stream` from now on. Allowed while a pipelined step is in flight (it is how a consumer on another stream
// reads that mode's rows).
int lorahip_demod_stream_wait(lorahip_demod *dm, void *hip_stream)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->streamWait(hip_stream);
    const DeviceGuard guard(dm->ctx->device);
    if (dm->evJoin == nullptr) LORAHIP_TRY(hipEventCreateWithFlags(&dm->evJoin, hipEventDisableTiming));
    LORAHIP_TRY(hipEventRecord(dm->evJoin, dm->ctx->stream));
    LORAHIP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(hip_stream), dm->evJoin, 0));
    return LORAHIP_OK;
}

// ... and the other direction: the launch stream behind what `hip_stream` holds now (the producer of the samples, the consumer of rows
// that the next step will overwrite), without a host wait.
int lorahip_demod_stream_follow(lorahip_demod *dm, void *hip_stream)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->streamFollow(hip_stream);
    const DeviceGuard guard(dm->ctx->device);
    if (dm->evJoin == nullptr) LORAHIP_TRY(hipEventCreateWithFlags(&dm->evJoin, hipEventDisableTiming));
    LORAHIP_TRY(hipEventRecord(dm->evJoin, reinterpret_cast<hipStream_t>(hip_stream)));
    LORAHIP_TRY(hipStreamWaitEvent(dm->ctx->stream, dm->evJoin, 0));
    return LORAHIP_OK;
}

int lorahip_demod_reset_stream(lorahip_demod *dm)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->resetStream();
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    return lorahip_reset_stream(dm->ctx);
}

int lorahip_demod_set_variant(lorahip_demod *dm, const int variant)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    // level 3 promises the reference block's packets, traces and ports: only kernels on the reference's operation graph qualify
    if (variant == LORAHIP_VARIANT_FMA) { setLastError("the contracted kernels (LORAHIP_VARIANT_FMA) are not offered at level 3: their bins are not the reference's"); return LORAHIP_E_INVALID; }
    if (dm->comp)
    {
        for (size_t i = 0; i < dm->comp->numParts(); i++) { const int rc = lorahip_demod_set_variant(dm->comp->part(i), variant); if (rc != LORAHIP_OK) return rc; }
        return LORAHIP_OK;
    }
    return lorahip_set_variant(dm->ctx, variant);
}

int lorahip_demod_set_stream_grid(lorahip_demod *dm, const int max_workgroups)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp)
    {
        for (size_t i = 0; i < dm->comp->numParts(); i++) { const int rc = lorahip_demod_set_stream_grid(dm->comp->part(i), max_workgroups); if (rc != LORAHIP_OK) return rc; }
        return LORAHIP_OK;
    }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    dm->streamGrid = max_workgroups;
    return LORAHIP_OK;
}

int lorahip_demod_set_stream_lanes(lorahip_demod *dm, const int log2_lanes)
{
    if (dm == nullptr || (log2_lanes > 6 && (log2_lanes < (LORAHIP_LANES_AHEAD | 3) || log2_lanes > (LORAHIP_LANES_AHEAD | 5)))) return LORAHIP_E_INVALID;
    if (dm->comp)
    {
        for (size_t i = 0; i < dm->comp->numParts(); i++) { const int rc = lorahip_demod_set_stream_lanes(dm->comp->part(i), log2_lanes); if (rc != LORAHIP_OK) return rc; }
        return LORAHIP_OK;
    }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    dm->streamLanes = log2_lanes;
    return LORAHIP_OK;
}

int lorahip_demod_stream_lanes(const lorahip_demod *dm)
{
    if (dm == nullptr || dm->comp) return LORAHIP_E_INVALID;
    const DeviceGuard guard(dm->ctx->device);
    return streamLanesChosen(dm->ctx->sf, unsigned(dm->B), dm->streamLanes, dm->coWaves);
}

int lorahip_demod_set_record_capacity(lorahip_demod *dm, const size_t max_calls_per_launch)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp)
    {
        for (size_t i = 0; i < dm->comp->numParts(); i++) { const int rc = lorahip_demod_set_record_capacity(dm->comp->part(i), max_calls_per_launch); if (rc != LORAHIP_OK) return rc; }
        return LORAHIP_OK;
    }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    dm->streamCapMax = max_calls_per_launch;
    return LORAHIP_OK;
}

int lorahip_demod_set_fine_gather(lorahip_demod *dm, const int enable)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setFineGather(enable);
    return lorahip_set_fine_gather(dm->ctx, enable);
}

int lorahip_demod_activate(lorahip_demod *dm)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->activate();
    // activate() resets only _state and _chirpTable (:139-143); everything else keeps the
    // constructor / zero state, or whatever the previous activation left. While the state lives on the device (streaming mode)
    // the reset travels with the next launch as a flag; the mirrors get it when they are next brought up to date.
    if (dm->devStateFresh) dm->activatePending = true;
    else
    {
        const int rc = syncMirrors(dm);
        if (rc != LORAHIP_OK) return rc;
        for (auto &k : dm->ch) { k.state = ST_FRAMESYNC; k.downTable = false; }
    }
    dm->nNearSquelch = dm->nNearStep = 0;
    return LORAHIP_OK;
}

int lorahip_demod_run_device(lorahip_demod *dm, const float *iq_dev, const size_t samples_per_channel, int64_t *rounds)
{
    if (dm == nullptr || iq_dev == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) { setLastError("lorahip_demod_run_device: an object made by lorahip_demod_create_mixed takes per-channel segments"); return LORAHIP_E_INVALID; }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }         // before anything of the object changes
    // n_channels streams of equal length back to back: the placement is two numbers, not 3 * n_channels (ch[].base / len / pos are
    // filled only for the paths that read them, applyGeometry)
    dm->uniform = true; dm->uniSpc = dm->uniStride = samples_per_channel; dm->geomApplied = false; dm->posOnDevice = false;
    dm->append = false; dm->appendFresh = true;
    return runAny(dm, iq_dev, rounds);
}

int lorahip_demod_run_device_append(lorahip_demod *dm, const float *iq_dev, const size_t row_stride, const size_t n_valid, int64_t *rounds)
{
    if (dm == nullptr || n_valid > row_stride || (n_valid && iq_dev == nullptr)) return LORAHIP_E_INVALID;
    if (dm->comp) { setLastError("append runs are per part: lorahip_demod_part_handle"); return LORAHIP_E_INVALID; }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    if (!dm->appendFresh && n_valid < dm->appendPrev) { setLastError("an append run was given fewer samples than the one before it (lorahip_demod_rewind starts a new stream)"); return LORAHIP_E_INVALID; }
    // every channel's stream is the first n_valid samples of its row; a channel continues at its own read position (the device's
    // copy of the state holds it: nothing is uploaded per run)
    dm->uniform = true; dm->uniSpc = n_valid; dm->uniStride = row_stride; dm->geomApplied = false;
    if (dm->appendFresh) dm->posOnDevice = false;
    dm->append = true;
    return runAny(dm, iq_dev, rounds);
}

int lorahip_demod_rewind(lorahip_demod *dm)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) { for (size_t i = 0; i < dm->comp->numParts(); i++) (void)lorahip_demod_rewind(dm->comp->part(i)); return LORAHIP_OK; }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    dm->appendFresh = true; dm->appendPrev = 0;
    return LORAHIP_OK;
}

int lorahip_demod_run_device_segments(lorahip_demod *dm, const float *iq_dev, const int64_t *first_sample, const size_t *n_samples, int64_t *rounds)
{
    if (dm == nullptr || first_sample == nullptr || n_samples == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->runSegments(&iq_dev, 1, first_sample, n_samples, rounds);
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    bool any = false;
    for (size_t c = 0; c < dm->B; c++)
    {
        if (first_sample[c] < 0 || n_samples[c] > size_t(0x7fffffffffffffffLL) - size_t(first_sample[c])) return LORAHIP_E_INVALID;
        any = any || n_samples[c] != 0;
    }
    if (any && iq_dev == nullptr) return LORAHIP_E_INVALID;
    const DeviceGuard guard(dm->ctx->device);
    dm->uniform = false; dm->geomApplied = true; dm->posOnDevice = false;
    dm->append = false; dm->appendFresh = true;
    for (size_t c = 0; c < dm->B; c++)
    {
        dm->ch[c].base = size_t(first_sample[c]);
        dm->ch[c].len = n_samples[c];
        dm->ch[c].pos = 0; dm->ch[c].callCount = 0;
    }
    return runAny(dm, iq_dev, rounds);
}

int lorahip_demod_run(lorahip_demod *dm, const float *const *streams, const size_t *n_samples, int64_t *rounds)
{
    if (dm == nullptr || streams == nullptr || n_samples == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp)
    {
        for (size_t c = 0; c < dm->B; c++) if (n_samples[c] && streams[c] == nullptr) return LORAHIP_E_INVALID;
        return dm->comp->run(streams, n_samples, rounds);
    }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    const DeviceGuard guard(dm->ctx->device);
    // every argument is checked before anything of the object changes: a refused call leaves consumed() of the last run intact
    for (size_t c = 0; c < dm->B; c++)
        if ((n_samples[c] && streams[c] == nullptr) || n_samples[c] > (size_t(1) << 48)) return LORAHIP_E_INVALID;
    size_t total = 0;
    for (size_t c = 0; c < dm->B; c++) total += n_samples[c];
    if (total > dm->dIqSamples)
    {
        if (dm->dIq) { (void)hipFree(dm->dIq); dm->dIq = nullptr; dm->dIqSamples = 0; }
        LORAHIP_TRY(hipMalloc((void **)&dm->dIq, (total ? total : 1) * sizeof(cf32)));
        dm->dIqSamples = total;
    }
    dm->uniform = false; dm->geomApplied = true; dm->posOnDevice = false;
    dm->append = false; dm->appendFresh = true;
    total = 0;
    for (size_t c = 0; c < dm->B; c++)
    {
        dm->ch[c].base = total;
        dm->ch[c].len = n_samples[c];
        dm->ch[c].pos = 0; dm->ch[c].callCount = 0;
        total += n_samples[c];
    }
    {
        // the channels' buffers gathered into one device array back to back (base = running total): pinned double-buffered upload
        std::vector<const void *> src(dm->B);
        std::vector<size_t> len(dm->B);
        for (size_t c = 0; c < dm->B; c++) { src[c] = streams[c]; len[c] = n_samples[c] * sizeof(cf32); }
        const int rc = gatherUpload(dm->ctx, dm->dIq, src.data(), len.data(), dm->B);
        if (rc != LORAHIP_OK) return rc;
    }
    LORAHIP_TRY(hipStreamSynchronize(dm->ctx->stream));
    return runAny(dm, dm->dIq, rounds);
}

/* Host buffers that are the ROWS of one block of host memory -- channel c's n_samples[c] samples begin at sample first_sample[c] of
 * row c, row_stride samples per row: what a framework hands a block whose input buffer manager carves every port's slabs out of one
 * pinned allocation (lora_sdr_amd/pothos/LoRaDemodBatch.cpp::getInputBufferManager, the counterpart of LoRaDemod.cpp:346-357). The
 * span of the rows that holds samples crosses PCIe as ONE strided copy straight from the caller's memory (pinned: a plain DMA, no
 * staging, no per-channel call), and the run reads its per-channel segments of the device copy. */
int lorahip_demod_run_host_rows(lorahip_demod *dm, const float *rows, const size_t row_stride, const int64_t *first_sample, const size_t *n_samples, int64_t *rounds)
{
    if (dm == nullptr || first_sample == nullptr || n_samples == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp)
    {
        // one part holds every channel in order: its rows are the caller's; several parts take their channels' buffers one by one
        if (dm->comp->numParts() == 1) return lorahip_demod_run_host_rows(dm->comp->part(0), rows, row_stride, first_sample, n_samples, rounds);
        try
        {
            std::vector<const float *> ptr(dm->B);
            for (size_t c = 0; c < dm->B; c++)
            {
                if (first_sample[c] < 0 || size_t(first_sample[c]) > row_stride || n_samples[c] > row_stride - size_t(first_sample[c])) return LORAHIP_E_INVALID;
                ptr[c] = rows ? rows + 2 * (c * row_stride + size_t(first_sample[c])) : nullptr;
            }
            return lorahip_demod_run(dm, ptr.data(), n_samples, rounds);
        }
        catch (const std::bad_alloc &) { setLastError("out of host memory"); return LORAHIP_E_NOMEM; }
    }
    { const int rc = refuseWhilePiped(dm); if (rc != LORAHIP_OK) return rc; }
    size_t lo = row_stride, hi = 0;
    for (size_t c = 0; c < dm->B; c++)
    {
        if (first_sample[c] < 0 || size_t(first_sample[c]) > row_stride || n_samples[c] > row_stride - size_t(first_sample[c])) return LORAHIP_E_INVALID;
        if (n_samples[c] == 0) continue;
        lo = size_t(first_sample[c]) < lo ? size_t(first_sample[c]) : lo;
        hi = size_t(first_sample[c]) + n_samples[c] > hi ? size_t(first_sample[c]) + n_samples[c] : hi;
    }
    const size_t width = hi > lo ? hi - lo : 0;
    if (width && rows == nullptr) return LORAHIP_E_INVALID;
    if (width > (size_t(1) << 40) / (dm->B ? dm->B : 1)) return LORAHIP_E_INVALID;
    const DeviceGuard guard(dm->ctx->device);
    const size_t total = dm->B * width;
    if (total > dm->dIqSamples)
    {
        if (dm->dIq) { (void)hipFree(dm->dIq); dm->dIq = nullptr; dm->dIqSamples = 0; }
        LORAHIP_TRY(hipMalloc((void **)&dm->dIq, (total ? total : 1) * sizeof(cf32)));
        dm->dIqSamples = total;
    }
    if (width)
        LORAHIP_TRY(hipMemcpy2DAsync(dm->dIq, width * sizeof(cf32), rows + 2 * lo, row_stride * sizeof(cf32), width * sizeof(cf32), dm->B, hipMemcpyHostToDevice,
                                     dm->ctx->stream));
    dm->uniform = false; dm->geomApplied = true; dm->posOnDevice = false;
    dm->append = false; dm->appendFresh = true;
    for (size_t c = 0; c < dm->B; c++)
    {
        dm->ch[c].base = n_samples[c] ? c * width + (size_t(first_sample[c]) - lo) : 0;
        dm->ch[c].len = n_samples[c];
        dm->ch[c].pos = 0; dm->ch[c].callCount = 0;
    }
    return runAny(dm, dm->dIq, rounds);               // (in stream order behind the copy; returns with the stream drained: the rows are the caller's again)
}

//! accessors of the host queue first bring over what the last streaming launch left on the device; a failure there is the
//! accessor's failure (the records stay pending, see drainPending)
static int drained(const lorahip_demod *dm)
{
    lorahip_demod *m = const_cast<lorahip_demod *>(dm);
    if (m && m->comp) return LORAHIP_OK;                        // the parts drain their own
    if (m) { const int rc = refuseWhilePiped(m); if (rc != LORAHIP_OK) return rc; }
    return m ? drainPending(m) : LORAHIP_E_INVALID;
}

int lorahip_demod_run_device_segments_multi(lorahip_demod *dm, const float *const *iq_dev, const size_t n_devices, const int64_t *first_sample,
                                            const size_t *n_samples, int64_t *rounds)
{
    if (dm == nullptr || iq_dev == nullptr || first_sample == nullptr || n_samples == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->runSegments(iq_dev, n_devices, first_sample, n_samples, rounds);
    if (n_devices != 1) return LORAHIP_E_INVALID;
    return lorahip_demod_run_device_segments(dm, iq_dev[0], first_sample, n_samples, rounds);
}

size_t lorahip_demod_num_packets(const lorahip_demod *dm)
{
    if (dm == nullptr) return 0;
    if (dm->comp) return dm->comp->numPackets();
    const PendingLaunch &P = pendingOf(const_cast<lorahip_demod *>(dm));
    return dm->packets.size() + (P.valid ? P.packets : 0);      // known from the per-channel counts: no drain needed
}

int lorahip_demod_get_packet(const lorahip_demod *dm, const size_t i, int32_t *channel, int64_t *round,
                             size_t *len, int16_t *out, const size_t cap)
{
    { const int rc = drained(dm); if (rc != LORAHIP_OK) return rc; }
    if (dm->comp) return dm->comp->getPacket(i, channel, round, len, out, cap);
    if (dm == nullptr || i >= dm->packets.size()) return LORAHIP_E_INVALID;
    const Packet &p = dm->packets[i];
    if (channel) *channel = p.channel;
    if (round) *round = p.round;
    if (len) *len = p.len;
    if (out)
    {
        if (cap < p.len) return LORAHIP_E_INVALID;
        std::memcpy(out, dm->pktSyms.data() + p.off, p.len * sizeof(int16_t));
    }
    return LORAHIP_OK;
}

size_t lorahip_demod_num_packet_symbols(const lorahip_demod *dm)
{
    if (dm == nullptr) return 0;
    if (dm->comp) return dm->comp->numPacketSymbols();
    const PendingLaunch &P = pendingOf(const_cast<lorahip_demod *>(dm));
    return dm->pktSyms.size() + (P.valid ? P.packetSyms : 0);
}

int lorahip_demod_get_packets(const lorahip_demod *dm, int32_t *channels, int64_t *rounds, int64_t *lens, const size_t cap_packets,
                              int16_t *syms, const size_t cap_syms)
{
    { const int rc = drained(dm); if (rc != LORAHIP_OK) return rc; }
    if (dm->comp) return dm->comp->getPackets(channels, rounds, lens, cap_packets, syms, cap_syms);
    if (dm == nullptr || cap_packets < dm->packets.size() || cap_syms < dm->pktSyms.size()) return LORAHIP_E_INVALID;
    size_t o = 0;
    for (size_t i = 0; i < dm->packets.size(); i++)
    {
        const Packet &p = dm->packets[i];
        if (channels) channels[i] = p.channel;
        if (rounds) rounds[i] = p.round;
        if (lens) lens[i] = int64_t(p.len);
        if (syms) std::memcpy(syms + o, dm->pktSyms.data() + p.off, p.len * sizeof(int16_t));
        o += p.len;
    }
    return LORAHIP_OK;
}

static int packetsToDevice(lorahip_demod *dm, uint16_t *syms_dev, const size_t sym_stride, int32_t *nsyms_dev, int32_t *channel_dev,
                           const size_t cap_packets, size_t *n_packets, const bool sync)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) { setLastError("packets on the device are per part (one decoder per SF): lorahip_demod_part_handle"); return LORAHIP_E_INVALID; }
    {
        // The records of the last streaming launch are still on the device and nothing else is queued: pack them there
        // (rows: channels ascending, time ascending inside a channel). A channel that entered the launch inside a packet found
        // its symbols at the head of its row (carryLoad); only a run that had to take them from the mirrors (anyCarryIn: packets
        // longer than the carry rows, launches of a resumed run) takes the queue path below.
        PendingLaunch &Q = pendingOf(dm);
        if (Q.valid && dm->packets.empty() && !Q.anyCarryIn)
        {
            const size_t n = Q.packets;
            if (n_packets) *n_packets = n;
            if (n == 0) return LORAHIP_OK;
            if (syms_dev == nullptr || nsyms_dev == nullptr || sym_stride == 0 || sym_stride > 0x7fffffffu || cap_packets < n) return LORAHIP_E_INVALID;
            const DeviceGuard guard(dm->ctx->device);
            const StreamLayout &L = Q.lay;
            const size_t nbRow = align256(L.B * sizeof(int));
            { const int grc = growDense(dm, nbRow + n * sizeof(long long)); if (grc != LORAHIP_OK) return grc; }
            hipStream_t st = dm->ctx->stream;
            // the rows are numbered on the device (scanDescribe: exclusive prefix sum of the per-channel packet counts): nothing is
            // uploaded, and nothing on the host is reused, so the caller decides whether to wait
            LORAHIP_TRY(launchPackPackets(reinterpret_cast<const StreamPacket *>(dm->sDev + L.oPkt), reinterpret_cast<const int *>(dm->sDev + L.oNPkt),
                                          reinterpret_cast<const short *>(dm->sDev + L.oSym), reinterpret_cast<int *>(dm->dDense), L.B, int(L.symStride),
                                          int(L.capPkt), n, reinterpret_cast<long long *>(dm->dDense + nbRow), syms_dev, int(sym_stride), nsyms_dev,
                                          channel_dev, st));
            if (sync) LORAHIP_TRY(hipStreamSynchronize(st));
            return LORAHIP_OK;
        }
    }
    { const int rc = drainPending(dm); if (rc != LORAHIP_OK) return rc; }
    const size_t P = dm->packets.size();
    if (n_packets) *n_packets = P;
    if (P == 0) return LORAHIP_OK;
    if (syms_dev == nullptr || nsyms_dev == nullptr || sym_stride == 0 || cap_packets < P) return LORAHIP_E_INVALID;
    const DeviceGuard guard(dm->ctx->device);
    const size_t nbSym = align256(P * sym_stride * sizeof(uint16_t)), nbInt = align256(P * sizeof(int32_t));
    { const int grc = growDense(dm, nbSym + 2 * nbInt); if (grc != LORAHIP_OK) return grc; }
    uint16_t *hs = reinterpret_cast<uint16_t *>(dm->hDense);
    int32_t *hn = reinterpret_cast<int32_t *>(dm->hDense + nbSym), *hc = reinterpret_cast<int32_t *>(dm->hDense + nbSym + nbInt);
    std::memset(hs, 0, P * sym_stride * sizeof(uint16_t));
    for (size_t i = 0; i < P; i++)
    {
        const Packet &p = dm->packets[i];
        std::memcpy(hs + i * sym_stride, dm->pktSyms.data() + p.off, (p.len < sym_stride ? p.len : sym_stride) * sizeof(uint16_t));
        hn[i] = int32_t(p.len > 0x7fffffffu ? 0x7fffffff : p.len);     // the true length: a packet longer than the stride is flagged by the decoder
        hc[i] = p.channel;
    }
    hipStream_t st = dm->ctx->stream;
    LORAHIP_TRY(hipMemcpyAsync(syms_dev, hs, P * sym_stride * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    LORAHIP_TRY(hipMemcpyAsync(nsyms_dev, hn, P * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (channel_dev) LORAHIP_TRY(hipMemcpyAsync(channel_dev, hc, P * sizeof(int32_t), hipMemcpyHostToDevice, st));
    LORAHIP_TRY(hipStreamSynchronize(st));                              // the pinned scratch is reused by the next run
    return LORAHIP_OK;
}

int lorahip_demod_packets_to_device(lorahip_demod *dm, uint16_t *syms_dev, const size_t sym_stride, int32_t *nsyms_dev, int32_t *channel_dev,
                                    const size_t cap_packets, size_t *n_packets)
{
    return packetsToDevice(dm, syms_dev, sym_stride, nsyms_dev, channel_dev, cap_packets, n_packets, true);
}

/*! The signals of the run just made (an ordinary receiver step) into the registered signal rows from row `first` on: packed on the
 * device from the last launch's records while they are the only ones (the packets' device path), else from the host queue. Stream
 * ordered like the packet rows. *n = signals of the run (also when the rows are too small: LORAHIP_E_INVALID, nothing cleared). */
static int signalsToRows(lorahip_demod *dm, const size_t first, size_t *n, const bool sync)
{
    *n = 0;
    if (!dm->wantSignals || !dm->sigRowsOn) return LORAHIP_OK;
    const lorahip_signal_rows &R = dm->sigRows;
    const DeviceGuard guard(dm->ctx->device);
    hipStream_t st = dm->ctx->stream;
    PendingLaunch &Q = pendingOf(dm);
    if (Q.valid && dm->signals.empty())
    {
        *n = Q.signals;
        if (*n == 0) return LORAHIP_OK;
        if (first + *n > R.cap) { setLastError("lorahip_demod_receive: the signal rows cannot hold the signals that are due"); return LORAHIP_E_INVALID; }
        const StreamLayout &L = Q.lay;
        if (!L.signals) { *n = 0; return LORAHIP_OK; }
        LORAHIP_TRY(launchPackSignals(reinterpret_cast<const StreamSignal *>(dm->sDev + L.oSig), reinterpret_cast<const int *>(dm->sDev + L.oNSig), L.B,
                                      int(L.capPkt), R.channel, R.error, R.power, R.snr, first, R.cap, st));
        if (sync) LORAHIP_TRY(hipStreamSynchronize(st));
        return LORAHIP_OK;
    }
    { const int rc = drainPending(dm); if (rc != LORAHIP_OK) return rc; }
    const size_t S = dm->signals.size();
    *n = S;
    if (S == 0) return LORAHIP_OK;
    if (first + S > R.cap) { setLastError("lorahip_demod_receive: the signal rows cannot hold the signals that are due"); return LORAHIP_E_INVALID; }
    const size_t nb = align256(S * sizeof(int32_t));
    { const int grc = growDense(dm, 4 * nb); if (grc != LORAHIP_OK) return grc; }
    int32_t *hc = reinterpret_cast<int32_t *>(dm->hDense), *he = reinterpret_cast<int32_t *>(dm->hDense + nb);
    float *hp = reinterpret_cast<float *>(dm->hDense + 2 * nb), *hs = reinterpret_cast<float *>(dm->hDense + 3 * nb);
    for (size_t i = 0; i < S; i++) { const Signal &g = dm->signals[i]; hc[i] = g.channel; he[i] = g.error; hp[i] = g.power; hs[i] = g.snr; }
    if (R.channel) LORAHIP_TRY(hipMemcpyAsync(R.channel + first, hc, S * sizeof(int32_t), hipMemcpyDefault, st));
    if (R.error) LORAHIP_TRY(hipMemcpyAsync(R.error + first, he, S * sizeof(int32_t), hipMemcpyDefault, st));
    if (R.power) LORAHIP_TRY(hipMemcpyAsync(R.power + first, hp, S * sizeof(float), hipMemcpyDefault, st));
    if (R.snr) LORAHIP_TRY(hipMemcpyAsync(R.snr + first, hs, S * sizeof(float), hipMemcpyDefault, st));
    LORAHIP_TRY(hipStreamSynchronize(st));                              // the pinned scratch is reused by the next run
    return LORAHIP_OK;
}

int lorahip_demod_receive_signal_rows(lorahip_demod *dm, const lorahip_signal_rows *rows)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) { setLastError("receiver steps are per part: lorahip_demod_part_handle"); return LORAHIP_E_INVALID; }
    if (rows == nullptr) { std::memset(&dm->sigRows, 0, sizeof(dm->sigRows)); dm->sigRowsOn = false; return LORAHIP_OK; }
    if (rows->struct_size != sizeof(lorahip_signal_rows) || rows->cap > 0x7fffffffu) return LORAHIP_E_INVALID;
    dm->sigRows = *rows;
    dm->sigRowsOn = rows->cap != 0 && (rows->channel || rows->error || rows->power || rows->snr);
    return LORAHIP_OK;
}

size_t lorahip_demod_receive_num_signals(const lorahip_demod *dm) { return dm && !dm->comp ? dm->lastSignals : 0; }

size_t lorahip_demod_receive_steps(const lorahip_demod *dm, size_t *packets, size_t *signals, const size_t cap)
{
    if (dm == nullptr || dm->comp || dm->pipe == nullptr) return 0;
    const Pipe::Resident &R = static_cast<const Pipe *>(dm->pipe)->res;
    for (unsigned i = 0; i < R.nRep && i < cap; i++)
    {
        if (packets) packets[i] = R.repPk[i];
        if (signals) signals[i] = R.repSg[i];
    }
    return R.nRep;
}

int lorahip_demod_resident_active(const lorahip_demod *dm)
{
    return dm && !dm->comp && dm->pipe && static_cast<const Pipe *>(dm->pipe)->res.active ? 1 : 0;
}

int lorahip_demod_receive(lorahip_demod *dm, const float *iq_dev, const size_t row_stride, const size_t n_valid, const lorahip_packet_rows *rows,
                          size_t *n_packets, int64_t *work_calls)
{
    if (dm == nullptr || rows == nullptr || rows->struct_size != sizeof(lorahip_packet_rows)) return LORAHIP_E_INVALID;
    if (n_packets) *n_packets = 0;
    if (work_calls) *work_calls = 0;
    if (dm->comp) { setLastError("append runs are per part: lorahip_demod_part_handle"); return LORAHIP_E_INVALID; }
    if (rows->async == 2 || rows->async == 3)
    {
        if (n_valid > row_stride || (n_valid && iq_dev == nullptr)) return LORAHIP_E_INVALID;
        bool handled = false;
        if (rows->async == 3)
        {
            if (pipeOf(dm).active) { setLastError("a pipelined step is in flight: lorahip_demod_receive_flush first"); return LORAHIP_E_INVALID; }
            const int prc = residentStep(dm, iq_dev, row_stride, n_valid, rows, n_packets, work_calls, handled);
            if (handled || prc != LORAHIP_OK) return prc;
        }
        else
        {
            if (pipeOf(dm).res.active) { setLastError("the resident kernel is on the device: lorahip_demod_receive_flush first"); return LORAHIP_E_INVALID; }
            const int prc = pipeStep(dm, iq_dev, row_stride, n_valid, rows, n_packets, work_calls, handled);
            if (handled || prc != LORAHIP_OK) return prc;
        }
        // not in place yet (first step, or something else touched the object): an ordinary step, its packets delivered at once
    }
    else { const int frc = refuseWhilePiped(dm); if (frc != LORAHIP_OK) return frc; }
    const int64_t calls0 = dm->workCalls;
    int rc = lorahip_demod_run_device_append(dm, iq_dev, row_stride, n_valid, nullptr);
    if (rc != LORAHIP_OK) return rc;
    if (work_calls) *work_calls = dm->workCalls - calls0;
    size_t n = 0;
    dm->lastSignals = 0;
    rc = packetsToDevice(dm, rows->syms_dev, rows->sym_stride, rows->nsyms_dev, rows->channel_dev, rows->cap_packets, &n, rows->async == 0);
    if (n_packets) *n_packets = n;
    if (rc != LORAHIP_OK) return rc;                                    // (the packets stay queued: a caller with too few rows can fetch them)
    rc = signalsToRows(dm, 0, &dm->lastSignals, rows->async == 0);      // the block's signals of this step, beside its packets
    if (rc != LORAHIP_OK) return rc;                                    // (... and so do the signals: lorahip_demod_get_signals)
    lorahip_demod_clear_packets(dm);
    return LORAHIP_OK;
}

int lorahip_demod_receive_flush(lorahip_demod *dm, const lorahip_packet_rows *rows, size_t *n_packets, int64_t *work_calls)
{
    if (dm == nullptr || (rows != nullptr && rows->struct_size != sizeof(lorahip_packet_rows))) return LORAHIP_E_INVALID;
    if (n_packets) *n_packets = 0;
    if (work_calls) *work_calls = 0;
    if (dm->comp) return LORAHIP_OK;
    dm->lastSignals = 0;                                    // (a flush with nothing in flight delivers nothing: not the count of the call before)
    pipeOf(dm).res.nRep = 0;
    if (pipeOf(dm).res.active)
    {
        // the resident receiver: every step's rows came with its own call; what is left to report are the counts of the last step(s)
        size_t n0 = 0; int64_t c0 = 0;
        int rc0 = residentFlush(dm, &n0, &c0);
        if (n_packets) *n_packets = n0;
        if (work_calls) *work_calls = c0;
        if (rc0 != LORAHIP_OK || dm->lastSum.more == 0) return rc0;
        // a channel still holds samples its last step could not record: an ordinary step over the same rows takes them (as below)
        const Pipe &P0 = pipeOf(dm);
        const int64_t calls0 = dm->workCalls;
        rc0 = lorahip_demod_run_device_append(dm, P0.iq, P0.rowStride, dm->appendPrev, nullptr);
        if (rc0 != LORAHIP_OK) return rc0;
        if (work_calls) *work_calls = c0 + (dm->workCalls - calls0);
        if (rows == nullptr) { lorahip_demod_clear_packets(dm); return LORAHIP_OK; }
        size_t n2 = 0;
        rc0 = packetsToDevice(dm, rows->syms_dev, rows->sym_stride, rows->nsyms_dev, rows->channel_dev, rows->cap_packets, &n2, true);
        if (n_packets) *n_packets = n0 + n2;                  // (n0 of them in the rows of their own calls, n2 in these)
        if (rc0 != LORAHIP_OK) return rc0;
        {
            const size_t s1 = dm->lastSignals;
            size_t s2 = 0;
            rc0 = signalsToRows(dm, 0, &s2, true);            // (into the rows registered now, from row 0: the steps' signals are in their own)
            dm->lastSignals = s1 + s2;
            if (rc0 != LORAHIP_OK) return rc0;
        }
        lorahip_demod_clear_packets(dm);
        return LORAHIP_OK;
    }
    const bool wasPiped = pipeBusy(dm);
    size_t n1 = 0;
    int64_t c1 = 0;
    int rc = pipeFlush(dm, rows, &n1, &c1);
    if (n_packets) *n_packets = n1;
    if (work_calls) *work_calls = c1;
    if (rc != LORAHIP_OK || !wasPiped || dm->lastSum.more == 0) return rc;
    // The LAST step filled a channel's record or packet capacity: that channel still holds >= 2N samples nobody would look at again.
    // An ordinary step over the same rows resumes it until dry (the pipeline is left: this is lorahip_demod_receive's own path), its
    // packets behind the ones above. If THEY do not fit they stay queued like any ordinary step's (LORAHIP_E_INVALID, *n_packets =
    // all rows needed; the first rows are filled; lorahip_demod_packets_to_device / the next receive delivers the rest).
    const Pipe &P = pipeOf(dm);
    const int64_t calls0 = dm->workCalls;
    rc = lorahip_demod_run_device_append(dm, P.iq, P.rowStride, dm->appendPrev, nullptr);
    if (rc != LORAHIP_OK) return rc;
    if (work_calls) *work_calls = c1 + (dm->workCalls - calls0);
    if (rows == nullptr) { lorahip_demod_clear_packets(dm); return LORAHIP_OK; }
    size_t n2 = 0;
    rc = packetsToDevice(dm, rows->syms_dev ? rows->syms_dev + n1 * rows->sym_stride : nullptr, rows->sym_stride, rows->nsyms_dev ? rows->nsyms_dev + n1 : nullptr,
                         rows->channel_dev ? rows->channel_dev + n1 : nullptr, rows->cap_packets - n1, &n2, true);
    if (n_packets) *n_packets = n1 + n2;
    if (rc != LORAHIP_OK) return rc;
    {
        const size_t s1 = dm->lastSignals;
        size_t s2 = 0;
        rc = signalsToRows(dm, s1, &s2, true);
        dm->lastSignals = s1 + s2;
        if (rc != LORAHIP_OK) return rc;
    }
    lorahip_demod_clear_packets(dm);
    return LORAHIP_OK;
}

int lorahip_demod_set_signals(lorahip_demod *dm, const int enable)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setSignals(enable);
    dm->wantSignals = enable != 0;
    return LORAHIP_OK;
}

size_t lorahip_demod_num_signals(const lorahip_demod *dm)
{
    if (dm == nullptr) return 0;
    if (dm->comp) return dm->comp->numSignals();
    const PendingLaunch &P = pendingOf(const_cast<lorahip_demod *>(dm));
    return dm->signals.size() + (P.valid ? P.signals : 0);
}

int lorahip_demod_get_signals(const lorahip_demod *dm, int32_t *channels, int64_t *rounds, int32_t *errors, float *powers, float *snrs, const size_t cap)
{
    { const int rc = drained(dm); if (rc != LORAHIP_OK) return rc; }
    if (dm->comp) return dm->comp->getSignals(channels, rounds, errors, powers, snrs, cap);
    if (dm == nullptr || cap < dm->signals.size()) return LORAHIP_E_INVALID;
    for (size_t i = 0; i < dm->signals.size(); i++)
    {
        const Signal &g = dm->signals[i];
        if (channels) channels[i] = g.channel;
        if (rounds) rounds[i] = g.round;
        if (errors) errors[i] = g.error;
        if (powers) powers[i] = g.power;
        if (snrs) snrs[i] = g.snr;
    }
    return LORAHIP_OK;
}

void lorahip_demod_clear_packets(lorahip_demod *dm)
{
    if (dm == nullptr) return;
    if (dm->comp) { dm->comp->clearPackets(); return; }
    PendingLaunch &P = pendingOf(dm);
    if (P.valid)
    {
        // records still on the device: they can simply be dropped unless a channel is inside a packet -- the symbols it has
        // received so far open the first packet of the next run
        if (P.anyOpen && !dm->devCarryValid) (void)drainPending(dm);
        else P.valid = false;                                   // (with the open packets kept on the device nothing is lost)
    }
    dm->packets.clear();
    dm->pktSyms.clear();
    dm->signals.clear();
}

int64_t lorahip_demod_work_calls(const lorahip_demod *dm) { return dm ? (dm->comp ? dm->comp->workCalls() : dm->workCalls) : 0; }

double lorahip_demod_kernel_ms(const lorahip_demod *dm) { return dm ? (dm->comp ? dm->comp->kernelMs() : dm->kernelMs) : 0.0; }
int lorahip_demod_last_launches(const lorahip_demod *dm) { return dm ? (dm->comp ? dm->comp->lastLaunches() : dm->lastLaunches) : 0; }

int lorahip_demod_near_threshold(const lorahip_demod *dm, int64_t *near_squelch, int64_t *near_step)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->nearThreshold(near_squelch, near_step);
    if (near_squelch) *near_squelch = dm->nNearSquelch;
    if (near_step) *near_step = dm->nNearStep;
    return LORAHIP_OK;
}

int64_t lorahip_demod_consumed(const lorahip_demod *dm, const size_t channel)
{
    if (dm == nullptr || channel >= dm->B) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->consumed(channel);
    if (dm->mirrorsStale && dm->posOnDevice && dm->sHost)
    {
        lorahip_demod *m = const_cast<lorahip_demod *>(dm);
        if (ensureHead(m) != LORAHIP_OK) return LORAHIP_E_HIP;
        return int64_t(hostStates(m)[channel].pos);
    }
    if (!dm->posOnDevice && dm->uniform && !dm->geomApplied) return 0;     // placement set, nothing run on it yet
    return int64_t(dm->ch[channel].pos);
}

int lorahip_demod_consumed_all(const lorahip_demod *dm, int64_t *out)
{
    if (dm == nullptr || out == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->consumedAll(out);
    // (while the resident kernel is on the device the positions live in its steps; a copy queued behind it could wait for the flush)
    if (dm->pipe && static_cast<const Pipe *>(dm->pipe)->res.active)
    {
        setLastError("the resident kernel is on the device: lorahip_demod_receive_flush first");
        return LORAHIP_E_INVALID;
    }
    // the one read that can fail -- the per-channel state back from the device -- up front: an error is the call's, not a negative
    // entry a caller would take for "nothing consumed"
    if (dm->mirrorsStale && dm->posOnDevice && dm->sHost)
    {
        const int rc = ensureHead(const_cast<lorahip_demod *>(dm));
        if (rc != LORAHIP_OK) return rc;
    }
    for (size_t c = 0; c < dm->B; c++)
    {
        out[c] = lorahip_demod_consumed(dm, c);
        if (out[c] < 0) return int(out[c]);
    }
    return LORAHIP_OK;
}

int lorahip_demod_set_trace(lorahip_demod *dm, const int enable)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setTrace(enable);
    const bool was = dm->tracing;
    { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; }       // traceSymCount0 is read from the mirrors
    dm->userTracing = enable != 0;
    dm->tracing = dm->userTracing || dm->portsOn;
    if (!dm->userTracing) for (auto &k : dm->ch) { k.trace.clear(); k.traceStart = 0; k.traceSymCount0 = k.symCount; }
    else if (!was) for (auto &k : dm->ch) k.traceSymCount0 = k.symCount;
    return LORAHIP_OK;
}

size_t lorahip_demod_trace_len(const lorahip_demod *dm, const size_t channel)
{
    if (drained(dm) != LORAHIP_OK) return 0;                    // lorahip_last_error() says why
    if (dm == nullptr || channel >= dm->B) return 0;
    if (dm->comp) return dm->comp->traceLen(channel);
    return dm->ch[channel].trace.size();
}

int lorahip_demod_set_ports(lorahip_demod *dm, const lorahip_demod_ports *p)
{
    if (dm == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->setPorts(p);
    const DeviceGuard guard(dm->ctx->device);
    if (dm->ownFft) { (void)hipFree(dm->ownFft); dm->ownFft = nullptr; }
    if (dm->ownDec) { (void)hipFree(dm->ownDec); dm->ownDec = nullptr; }
    if (dm->ownRaw) { (void)hipFree(dm->ownRaw); dm->ownRaw = nullptr; }
    std::memset(&dm->ports, 0, sizeof(dm->ports));
    std::memset(&dm->hostPorts, 0, sizeof(dm->hostPorts));
    dm->portsOn = false;
    if (p != nullptr)
    {
        if (p->struct_size != sizeof(lorahip_demod_ports)) return LORAHIP_E_INVALID;
        if ((p->fft_dev && !p->fft_cap_frames) || (p->dec_dev && !p->dec_cap_samples) || (p->raw_dev && !p->raw_cap_samples)) return LORAHIP_E_INVALID;
        dm->ports = *p;
        if (p->host_buffers)
        {
            dm->hostPorts = *p;
            dm->ports.host_buffers = 0;
            dm->ports.fft_dev = dm->ports.dec_dev = dm->ports.raw_dev = nullptr;
            if (p->fft_dev) { LORAHIP_TRY(hipMalloc((void **)&dm->ownFft, dm->B * p->fft_cap_frames * dm->N * sizeof(cf32))); dm->ports.fft_dev = dm->ownFft; }
            if (p->dec_dev) { LORAHIP_TRY(hipMalloc((void **)&dm->ownDec, dm->B * p->dec_cap_samples * sizeof(cf32))); dm->ports.dec_dev = dm->ownDec; }
            if (p->raw_dev) { LORAHIP_TRY(hipMalloc((void **)&dm->ownRaw, dm->B * p->raw_cap_samples * sizeof(cf32))); dm->ports.raw_dev = dm->ownRaw; }
        }
        dm->portsOn = dm->ports.fft_dev || dm->ports.dec_dev || dm->ports.raw_dev;
    }
    if (dm->portsOn && !dm->tracing) { const int rc = syncMirrors(dm); if (rc != LORAHIP_OK) return rc; for (auto &k : dm->ch) k.traceSymCount0 = k.symCount; }
    dm->tracing = dm->userTracing || dm->portsOn;
    return LORAHIP_OK;
}

int lorahip_demod_port_counts(const lorahip_demod *dm, const size_t channel, size_t *fft_frames, size_t *dec_samples, size_t *raw_samples)
{
    if (dm == nullptr || channel >= dm->B) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->portCounts(channel, fft_frames, dec_samples, raw_samples);
    const Channel &k = dm->ch[channel];
    if (fft_frames) *fft_frames = k.portFft;
    if (dec_samples) *dec_samples = k.portDec;
    if (raw_samples) *raw_samples = k.portRaw;
    return LORAHIP_OK;
}

//! the label LoRaDemod::work() posts for one call (LoRaDemod.cpp:213,220-224,232,245,258,282,302-305): "" = none
static std::string labelOf(const lorahip_work_result &r, const size_t N, const float thresh, size_t &symCount)
{
    char buf[64];
    buf[0] = 0;
    switch (r.state_before)
    {
    case ST_FRAMESYNC:
        if (size_t(r.consumed) == 2 * N) return "SYNC";                                           // :213
        if (!(r.snr < thresh)) { std::snprintf(buf, sizeof(buf), "P %.4f", double(r.f_index)); return buf; }   // :220-224 (fixed, precision 4)
        return "";                                                                                // :232
    case ST_DOWNCHIRP0: return "DC";                                                              // :245
    case ST_DOWNCHIRP1: return "";                                                                // :258
    case ST_QUARTERCHIRP: symCount = 0; return "QC";                                              // :281-282
    default:
        symCount++;                                                                               // :290
        std::snprintf(buf, sizeof(buf), "S%zu %.4f", symCount, double(r.f_index));                // :302-305
        return buf;
    }
}

int lorahip_demod_get_labels(const lorahip_demod *dm, const size_t channel, char *buf, const size_t cap, size_t *n_calls, size_t *bytes)
{
    { const int rc = drained(dm); if (rc != LORAHIP_OK) return rc; }
    if (dm == nullptr || channel >= dm->B) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->getLabels(channel, buf, cap, n_calls, bytes);
    const auto &t = dm->ch[channel].trace;
    // _symCount is only reset at QUARTERCHIRP (:279): a trace that starts inside a packet continues the count the channel held then
    size_t symCount = dm->ch[channel].traceSymCount0;
    size_t used = 0;
    for (const auto &r : t)
    {
        const std::string s = labelOf(r, dm->N, dm->thresh, symCount);
        if (buf && used + s.size() + 1 <= cap) std::memcpy(buf + used, s.c_str(), s.size() + 1);
        used += s.size() + 1;
    }
    if (n_calls) *n_calls = t.size();
    if (bytes) *bytes = used;
    return (buf == nullptr || used <= cap) ? LORAHIP_OK : LORAHIP_E_INVALID;
}

int lorahip_demod_get_trace(const lorahip_demod *dm, const size_t channel, lorahip_work_result *out, const size_t cap)
{
    { const int rc = drained(dm); if (rc != LORAHIP_OK) return rc; }
    if (dm == nullptr || channel >= dm->B || out == nullptr) return LORAHIP_E_INVALID;
    if (dm->comp) return dm->comp->getTrace(channel, out, cap);
    const auto &t = dm->ch[channel].trace;
    if (cap < t.size()) return LORAHIP_E_INVALID;
    if (!t.empty()) std::memcpy(out, t.data(), t.size() * sizeof(lorahip_work_result));
    return LORAHIP_OK;
}

} // extern "C"
