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
    //! The packet the channel is inside continues behind the symbols it has already (flag bit 2; `st` with the launch's flags
    //! applied): the channel's lanes (lane `t` of `T`) copy them from the carry rows to the head of the symbol row.
    //! `copy` false (the resident receiver): they stay where they are -- the symbol row holds only what the launch adds (nSym counts from
    //! 0), whoever packs the packet takes its first symbols from the carry row (residentPackOwn), and carryOut appends behind them.
    __device__ __forceinline__ void carryIn(const StreamArgs &s, const StreamState &st, const unsigned channel, const int t, const int T, const bool copy = true)
    {
        if (!copy || !(s.flags & 4) || st.state != ST_DATASYMBOLS) return;
        const int k = st.symCount < s.carryCap ? st.symCount : s.carryCap;
        const short *src = s.carry + (size_t)channel * s.carryCap;
        // (read at agent scope: in the resident receiver the row was written by this compute unit a step ago and an older copy of the
        // line may still sit in its L1)
        // (eight loads in flight per lane before the stores -- instead of a round trip to L2 per element, six per step with 8 lanes per
        // channel -- was measured in the resident receiver and LOST: 2.36 -> 1.79 Gsym/s at SF7, the registers it takes are spilled in the
        // window loop; profiles/r06/s26_*)
        for (int i = t; i < k; i += T) symOut[i] = __hip_atomic_load(const_cast<short *>(src) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nSym = st.symCount;
    }
    //! ... and a channel that ends the launch inside a packet leaves the packet's symbols in the carry rows (flag bit 3): the last
    //! k = min(symCount, nSym) entries of its row, written by the writer lane, are the packet's LAST k symbols and go to the entries
    //! symCount - k ... of the carry row. With carryIn's copy k = symCount (the whole packet, from entry 0); without it only what the
    //! launch added is appended -- with short receiver steps one turn of the loop instead of mtu / T, a round trip to L2 each.
    //! `acrossWaves`: the channel's lanes span several wavefronts (demodStreamWide): the other wavefronts must not load before the
    //! writer's wavefront has waited for its stores -- a workgroup barrier between the two (every thread of the workgroup calls this,
    //! the flags and the loop exit are workgroup-uniform).
    __device__ __forceinline__ void carryOut(const StreamArgs &s, const StreamState &st, const unsigned channel, const int t, const int T, const bool mine = true,
                                             const bool acrossWaves = false)
    {
        if (!(s.flags & 8)) return;                                 // uniform over the launch
        // The writer lane's stores have to be seen by its neighbours' loads. They are lanes of ONE wavefront, whose memory operations
        // go through one L1 in order: a fence -- the wait for the stores -- orders them, and the loads are made at agent scope (they
        // read L2, where the stores have landed), so no stale L1 line can be hit.
        // An agent-scope FENCE would do too, but it writes the whole L2 back (buffer_wbl2): 90 us per launch at 2048 wavefronts.
        // (the wait is spelled out: a workgroup-scope fence does not wait for global stores on this target -- the wavefronts of a workgroup
        // share an L1 --, and the loads below go past the L1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (acrossWaves) __syncthreads();                           // ... and the writer's wavefront has got here: its last symbol is in L2
        if (!mine || st.state != ST_DATASYMBOLS) return;
        int k = st.symCount < nSym ? st.symCount : nSym;
        int at = st.symCount - k;                                   // (0 with carryIn's copy: nSym >= symCount there)
        if (at + k > s.carryCap) { at = 0; k = k < s.carryCap ? k : s.carryCap; }     // (a packet longer than the carry row: as ever, its last k symbols from entry 0 -- such receivers keep the host path)
        short *dst = s.carry + (size_t)channel * s.carryCap + at;
        for (int i = t; i < k; i += T) dst[i] = __hip_atomic_load(symOut + (nSym - k + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

/*! One work() call after its detect(s): `value` .. `fIndex` are the locals of LoRaDemod::work() as they stand after the
 * optional second detect (:203 overwrites power / powerAvg / fIndex, not snr), `match1` the second sync-word test.
 *
 * Written WITHOUT branches on the state: where a wavefront carries several channels (demodStream, SF6-9) they are in different
 * states at the same time, and a switch over the five states executes every arm present in the wave one after the other under
 * exec masks (~200 instructions per call, profiles/r04). As selects it is ~40 operations whatever the mix; where the state is
 * wave-uniform (one channel per wavefront / workgroup) the same expressions run on the scalar unit. Only the record stores
 * remain predicated. Each line cites the statement of the reference it stands for. */
template <int N>
__device__ __forceinline__ void frameStep(StreamState &st, const StreamArgs &s, StreamOut &o, const bool writer,
                                          const int value, const float power, const float powerAvg, const float snr,
                                          const float fIndex, const bool squelched, const bool syncd, const bool match0,
                                          const bool match1, const int fineIdxBefore, const float fineErrBefore)
{
    const int stateBefore = st.state;
    const int fineIdxAfter = st.fineTuneIndex;      // the caller has committed window 0's steps (:160-162)
    const bool isFS = stateBefore == ST_FRAMESYNC, isD0 = stateBefore == ST_DOWNCHIRP0, isD1 = stateBefore == ST_DOWNCHIRP1;
    const bool isQC = stateBefore == ST_QUARTERCHIRP, isDA = stateBefore == ST_DATASYMBOLS;
    const bool sync3 = isFS && syncd && match0 && match1;                                                // :209
    const bool fsOpen = isFS && !sync3 && !squelched;                                                    // :217
    const bool fsQuiet = isFS && !sync3 && squelched;                                                    // :228
    const int error = value > N / 2 ? value - N : value;                                                 // :246-248, :262-264
    const int halfError = st.freqError / 2;                                                              // :278 (int division: towards zero)

    int total = N;
    total = sync3 ? 2 * N : total;                                                                       // :210
    total = fsOpen ? N - value : total;                                                                  // :219
    total = isQC ? N / 4 + halfError : total;                                                            // :278

    const int symCount = isQC ? 0 : st.symCount + (isDA ? 1 : 0);                                        // :279, out[_symCount++]  :290
    const bool post = isDA && ((unsigned)symCount >= s.mtu || squelched);                                // :291

    float ffe = st.finefreqError;
    ffe = fsOpen ? ffe + fIndex : ffe;                                                                   // :221
    ffe = isQC ? ffe + (float)halfError : ffe;                                                           // :277
    ffe = (fsQuiet || post) ? 0.0f : ffe;                                                                // :230, :299
    const int fti = fsQuiet ? 0 : st.fineTuneIndex;                                                      // :231
    const int freqError = isD0 ? error : (isD1 ? (st.freqError + error) / 2 : st.freqError);             // :249, :265
    // FRAMESYNC stays unless the sync words matched (:211); the chirp states and QUARTERCHIRP step on (:243, :256, :276); DATASYMBOLS
    // stays until the packet is posted (:300)
    const int next = isFS ? (sync3 ? ST_DOWNCHIRP0 : ST_FRAMESYNC) : (isDA ? (post ? ST_FRAMESYNC : ST_DATASYMBOLS) : stateBefore + 1);
    const int downTable = sync3 ? 1 : (isD1 ? 0 : st.downTable);                                        // :212, :257

#ifndef LORAHIP_TIMING_NO_RECORD_STORES     // timing-only build (profiles/r03): what the per-call record stores cost
    if (writer && isDA) o.symOut[o.nSym] = (short)value;                                                 // out[_symCount++] = value  :290
#endif
    if (writer && post) { StreamPacket q; q.callIndex = st.callCount; q.len = symCount; o.pktOut[o.nPkt] = q; }   // postMessage  :295-298
    if (o.sigOut)                                                                                        // uniform over the launch
    {
        // emitSignal("error" / "power" / "snr") as a record of its own (:267-269): the caller evaluated the logarithms for this call
        if (writer && isD1) { StreamSignal g; g.callIndex = st.callCount; g.error = freqError; g.power = power; g.snr = snr; o.sigOut[o.nSig] = g; }
        o.nSig += isD1 ? 1 : 0;
    }
    if (writer && o.out)
    {
        lorahip_work_result r;
        r.consumed = total;
        r.state_before = stateBefore;
        r.value = value;
        r.power = power; r.power_avg = powerAvg; r.snr = snr; r.f_index = fIndex;
        r.worked = 1;
        r.packet_len = post ? symCount : 0;
        r.signals = isD1 ? 1 : 0;
        r.sig_error = isD1 ? freqError : 0;
        r.sig_power = isD1 ? power : 0.0f;
        r.sig_snr = isD1 ? snr : 0.0f;
        r.fine_idx_before = fineIdxBefore; r.fine_idx_after = fineIdxAfter; r.fine_err_before = fineErrBefore; r.reserved = 0;
        o.out[o.calls] = r;
    }
    o.nSym += isDA ? 1 : 0;
    o.nPkt += post ? 1 : 0;
    o.calls++;
    st.state = next; st.downTable = downTable; st.freqError = freqError; st.fineTuneIndex = fti; st.finefreqError = ffe;
    st.symCount = symCount;
    st.prevValue = (short)value;                                                                         // :326
    st.pos += total;                                                                                     // consume(total)  :320
    st.callCount++;
}

} // namespace lorahip
