// The LoRaDemod frame state machine (LoRaDemod.cpp:176-312) as the streaming kernels run it: per work() call, in the
// registers of every lane that owns the channel (replicated, so no broadcast is needed); the writer lane appends the
// symbol / packet / trace records. Shared by lorahip_stream.hip (a channel per T <= 64 lanes) and lorahip_wide.hip
// (a channel per workgroup). The host-path twin is lorahip_demod.cpp::runRounds; tests pin both against the verbatim
// LoRaDemod.cpp.
#pragma once
#include "lorahip_device.h"

namespace lorahip {

enum { ST_FRAMESYNC = 0, ST_DOWNCHIRP0, ST_DOWNCHIRP1, ST_QUARTERCHIRP, ST_DATASYMBOLS };

//! where one channel's records of this launch go, and how many there are so far
struct StreamOut
{
    lorahip_work_result *out;
    short *symOut;
    StreamPacket *pktOut;
    StreamSignal *sigOut;
    int calls, nSym, nPkt, nSig;
    __device__ __forceinline__ void init(const StreamArgs &s, const unsigned channel)
    {
        out = s.calls ? s.calls + (size_t)channel * s.cap : nullptr;
        symOut = s.symOut + (size_t)channel * s.symStride;
        pktOut = s.pktOut + (size_t)channel * s.capPkt;
        sigOut = s.sigOut ? s.sigOut + (size_t)channel * s.capPkt : nullptr;
        calls = nSym = nPkt = nSig = 0;
    }
    //! the packet the channel is inside continues behind the symbols it has already (flag bit 2; `st` with the launch's flags applied)
    __device__ __forceinline__ void carryIn(const StreamArgs &s, const StreamState &st)
    {
        if ((s.flags & 4) && st.state == ST_DATASYMBOLS) nSym = st.symCount;
    }
};

/*! One work() call after its detect(s): `value` .. `fIndex` are the locals of LoRaDemod::work() as they stand after the
 * optional second detect (:203 overwrites power / powerAvg / fIndex, not snr), `match1` the second sync-word test. */
template <int N>
__device__ __forceinline__ void frameStep(StreamState &st, const StreamArgs &s, StreamOut &o, const bool writer,
                                          const int value, const float power, const float powerAvg, const float snr,
                                          const float fIndex, const bool squelched, const bool syncd, const bool match0,
                                          const bool match1, const int fineIdxBefore, const float fineErrBefore)
{
    const int stateBefore = st.state;
    const int fineIdxAfter = st.fineTuneIndex;      // the caller has committed window 0's steps (:160-162)
    int total = 0, packetLen = 0, signals = 0, sigError = 0;
    switch (st.state)
    {
    case ST_FRAMESYNC:
        if (syncd && match0 && match1) { total = 2 * N; st.state = ST_DOWNCHIRP0; st.downTable = 1; }   // :209-213
        else if (!squelched) { total = N - value; st.finefreqError += fIndex; }                          // :217-221
        else { total = N; st.finefreqError = 0.0f; st.fineTuneIndex = 0; }                               // :228-233
        break;
    case ST_DOWNCHIRP0:
    {
        st.state = ST_DOWNCHIRP1;
        total = N;
        int error = value;
        if (value > N / 2) error -= N;
        st.freqError = error;                                                                            // :246-249
    } break;
    case ST_DOWNCHIRP1:
    {
        st.state = ST_QUARTERCHIRP;
        total = N;
        st.downTable = 0;
        int error = value;
        if (value > N / 2) error -= N;
        st.freqError = (st.freqError + error) / 2;                                                       // :262-265
        signals = 1; sigError = st.freqError;                                                            // :267-269
        if (o.sigOut)
        {
            // emitSignal("error" / "power" / "snr") as a record of its own: the caller evaluated the logarithms for this call
            if (writer) { StreamSignal g; g.callIndex = st.callCount; g.error = sigError; g.power = power; g.snr = snr; o.sigOut[o.nSig] = g; }
            o.nSig++;
        }
    } break;
    case ST_QUARTERCHIRP:
        st.state = ST_DATASYMBOLS;
        total = N / 4 + st.freqError / 2;                                                                // :278
        st.finefreqError += (float)(st.freqError / 2);
        st.symCount = 0;
        break;
    default: // ST_DATASYMBOLS
        total = N;
#ifndef LORAHIP_TIMING_NO_RECORD_STORES     // timing-only build (profiles/r03): what the per-call record stores cost
        if (writer) o.symOut[o.nSym] = (short)value;                                                     // out[_symCount++] = value  :290
#endif
        o.nSym++;
        st.symCount++;
        if ((unsigned)st.symCount >= s.mtu || squelched)                                                 // :291
        {
            packetLen = st.symCount;
            if (writer) { o.pktOut[o.nPkt].callIndex = st.callCount; o.pktOut[o.nPkt].len = packetLen; } // postMessage  :295-298
            o.nPkt++;
            st.finefreqError = 0.0f;
            st.state = ST_FRAMESYNC;
        }
        break;
    }
    st.prevValue = (short)value;                                                                         // :326
    st.pos += total;                                                                                     // consume(total)  :320
    if (writer && o.out)
    {
        lorahip_work_result r;
        r.consumed = total;
        r.state_before = stateBefore;
        r.value = value;
        r.power = power; r.power_avg = powerAvg; r.snr = snr; r.f_index = fIndex;
        r.worked = 1;
        r.packet_len = packetLen;
        r.signals = signals;
        r.sig_error = sigError;
        r.sig_power = signals ? power : 0.0f;
        r.sig_snr = signals ? snr : 0.0f;
        r.fine_idx_before = fineIdxBefore; r.fine_idx_after = fineIdxAfter; r.fine_err_before = fineErrBefore; r.reserved = 0;
        o.out[o.calls] = r;
    }
    o.calls++;
    st.callCount++;
}

} // namespace lorahip
