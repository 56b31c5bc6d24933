// The resident receiver's protocol on the device: the step messages (host ring -> relay wavefronts -> mirrors -> LDS), a wavefront's own
// packets and signals into the step's rows, the end of a step and its report. Shared by the RES instances of demodStream
// (lorahip_streamkernel.h: SF7-10) and of demodStreamWide (lorahip_wide.hip: SF11 / SF12). ResidentMsg / ResidentCtl / ResidentHost:
// lorahip_internal.h; the host side: lorahip_demod.cpp::residentStep; the protocol in words: INTEGRATION.md section 4.
#pragma once
#include "lorahip_framemachine.h"

namespace lorahip {

/***********************************************************************
 * The resident receiver (RES instances of demodStream; ResidentMsg / ResidentCtl in lorahip_internal.h)
 **********************************************************************/
template <class V> __device__ __forceinline__ void sysStore(V *p, const V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <class V> __device__ __forceinline__ V sysLoad(const V *p) { return __hip_atomic_load(const_cast<V *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <class V> __device__ __forceinline__ V agentLoad(const V *p) { return __hip_atomic_load(const_cast<V *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

//! a step's message as the wavefront holds it (wave-uniform: scalar registers)
struct ResMsgR
{
    unsigned long long nValid;
    unsigned short *syms; int *nsyms, *chan, *sigCh, *sigErr; float *sigPow, *sigSnr;
    unsigned symStride, capRows, capSig, flags;
};

//! what the wavefronts of a workgroup leave for the one that arrives last at the end of a step (RES_RING sets, by step & 3: a fast
//! wavefront may be up to RES_DEPTH_MAX steps ahead of a slow one -- the host rings step k + depth + 1 only after step k has been reported),
//! and the step's message for the workgroup's other wavefronts
struct ResLds
{
    int calls[RES_RING], arrive[RES_RING], more[RES_RING];
    short *carry; int carryCap;         // StreamArgs::carry / carryCap for what runs at the end of a step (read from here there: a kernel argument kept in
                                        // scalar registers across the window loop costs the loop registers)
    int go;                             // demodStreamWide: wavefront 0's verdict on the wait for a step (its workgroup moves in step)
    unsigned msgSeq[RES_RING];          // the step whose message the workgroup holds in msg[step & 3] (whichever wavefront found it first left it there)
    ResMsgR msg[RES_RING];
};

__device__ __forceinline__ unsigned long long uni64(const unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

//! a ring slot read at system scope (from memory); true if it holds step `want` and its check word fits
__device__ __forceinline__ bool residentRead(const ResidentMsg *g, const unsigned want, ResidentMsg &c)
{
    if (sysLoad(&g->seq) != want) return false;
    c.nValid = sysLoad(&g->nValid);
    c.syms = reinterpret_cast<unsigned short *>(sysLoad(reinterpret_cast<const unsigned long long *>(&g->syms)));
    c.nsyms = reinterpret_cast<int *>(sysLoad(reinterpret_cast<const unsigned long long *>(&g->nsyms)));
    c.chan = reinterpret_cast<int *>(sysLoad(reinterpret_cast<const unsigned long long *>(&g->chan)));
    c.sigCh = reinterpret_cast<int *>(sysLoad(reinterpret_cast<const unsigned long long *>(&g->sigCh)));
    c.sigErr = reinterpret_cast<int *>(sysLoad(reinterpret_cast<const unsigned long long *>(&g->sigErr)));
    c.sigPow = reinterpret_cast<float *>(sysLoad(reinterpret_cast<const unsigned long long *>(&g->sigPow)));
    c.sigSnr = reinterpret_cast<float *>(sysLoad(reinterpret_cast<const unsigned long long *>(&g->sigSnr)));
    c.symStride = sysLoad(&g->symStride); c.capRows = sysLoad(&g->capRows); c.capSig = sysLoad(&g->capSig); c.flags = sysLoad(&g->flags);
    c.seq = want;
    c.check = sysLoad(&g->check);
    return c.check == residentCheck(c);
}

//! a message found in the host's ring into all sixteen mirrors, a lane per mirror: the fields, then the check word, then the step number
//! (a reader that sees the number verifies the check)
__device__ __forceinline__ void residentRelay(const StreamArgs &s, const unsigned want, const ResidentMsg &c, const unsigned lane)
{
    if (lane >= 16u) return;
    ResidentMsg *q = &s.res->msg[lane][want & 7];
    sysStore(&q->nValid, c.nValid);
    sysStore(reinterpret_cast<unsigned long long *>(&q->syms), (unsigned long long)(size_t)c.syms);
    sysStore(reinterpret_cast<unsigned long long *>(&q->nsyms), (unsigned long long)(size_t)c.nsyms);
    sysStore(reinterpret_cast<unsigned long long *>(&q->chan), (unsigned long long)(size_t)c.chan);
    sysStore(reinterpret_cast<unsigned long long *>(&q->sigCh), (unsigned long long)(size_t)c.sigCh);
    sysStore(reinterpret_cast<unsigned long long *>(&q->sigErr), (unsigned long long)(size_t)c.sigErr);
    sysStore(reinterpret_cast<unsigned long long *>(&q->sigPow), (unsigned long long)(size_t)c.sigPow);
    sysStore(reinterpret_cast<unsigned long long *>(&q->sigSnr), (unsigned long long)(size_t)c.sigSnr);
    sysStore(&q->symStride, c.symStride); sysStore(&q->capRows, c.capRows); sysStore(&q->capSig, c.capSig); sysStore(&q->flags, c.flags);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sysStore(&q->check, c.check);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sysStore(&q->seq, want);
}

//! A relay wavefront that has finished a step's windows looks for the NEXT step's message before it reports: the host rings one step
//! ahead, so it is usually there by then, and the others find it in the mirrors the moment they finish instead of a PCIe round trip later.
__device__ __forceinline__ void residentLookAhead(const StreamArgs &s, const unsigned next)
{
    if (blockIdx.x >= 8u || (threadIdx.x >> 6) != 0u) return;
    ResidentMsg c;
    if (sysLoad(&s.res->msg[blockIdx.x & 15u][next & 7].seq) == next) return;          // another relay wavefront has been there
    if (residentRead(&s.resHost->msg[next & 7], next, c)) residentRelay(s, next, c, threadIdx.x & 63u);
}

//! the step's message from the workgroup's LDS copy into the wavefront's scalar registers
__device__ __forceinline__ void residentMsgFromLds(const ResLds *sR, const int par, ResMsgR &m)
{
    const ResMsgR &q = sR->msg[par];
    m.nValid = uni64(q.nValid);
    m.syms = reinterpret_cast<unsigned short *>(uni64((unsigned long long)(size_t)q.syms));
    m.nsyms = reinterpret_cast<int *>(uni64((unsigned long long)(size_t)q.nsyms));
    m.chan = reinterpret_cast<int *>(uni64((unsigned long long)(size_t)q.chan));
    m.sigCh = reinterpret_cast<int *>(uni64((unsigned long long)(size_t)q.sigCh));
    m.sigErr = reinterpret_cast<int *>(uni64((unsigned long long)(size_t)q.sigErr));
    m.sigPow = reinterpret_cast<float *>(uni64((unsigned long long)(size_t)q.sigPow));
    m.sigSnr = reinterpret_cast<float *>(uni64((unsigned long long)(size_t)q.sigSnr));
    m.symStride = (unsigned)__builtin_amdgcn_readfirstlane((int)q.symStride); m.capRows = (unsigned)__builtin_amdgcn_readfirstlane((int)q.capRows);
    m.capSig = (unsigned)__builtin_amdgcn_readfirstlane((int)q.capSig); m.flags = (unsigned)__builtin_amdgcn_readfirstlane((int)q.flags);
    // (No acquire fence at agent scope here: it invalidates L1 and L2 (buffer_inv sc1), and 2048 wavefronts doing that once per step cost
    // more than the step. What could be stale is decided by construction instead: a step works up to the last WHOLE 128-byte line of
    // every row (the host passes n_valid rounded down to 16 samples; rows start on line boundaries), so no line the kernel has ever read
    // holds samples that arrive later; the records and carry rows a workgroup reads back are its own and are read at agent scope.)
}

/*! Wait for the message of step `want`. The wavefronts of a workgroup are independent in the loop, so each waits for itself -- on a
 * word in LDS: whichever wavefront of the workgroup finds the message first leaves it there for the others. A waiting wavefront looks
 * at MEMORY only every fourth turn (the four take turns: about one poll per workgroup and turn), at its group's mirror of the ring in
 * device memory (read at system scope, i.e. from memory; sixteen mirrors, see ResidentCtl). The HOST's ring is read by eight wavefronts
 * only -- wavefront 0 of workgroups 0..7, whenever they wait: reads of host memory by hundreds of wavefronts (PCIe round trips, address
 * translation) slowed every memory operation of the kernel by orders of magnitude (profiles/r06) -- and whichever of them finds a
 * new message copies it into all sixteen mirrors, a lane per mirror. A message counts only when its check word fits its fields (neither
 * the host's stores nor the relay are atomic). Between polls the wavefront naps. false: leave the kernel (quit message, abort flag,
 * or nothing for s.resWatchdog ticks: every spin is bounded). */
__device__ __forceinline__ bool residentWait(const StreamArgs &s, const unsigned want, ResMsgR &m, ResLds *sR, const bool solo = false)
{
    const int par = int(want & 3u);
    ResidentMsg *g = &s.res->msg[blockIdx.x & 15u][want & 7];
    const ResidentMsg *h = &s.resHost->msg[want & 7];
    const unsigned long long t0 = wall_clock64();
    const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const bool relay = blockIdx.x < 8u && wave == 0u;
    unsigned it = 0;
    for (;; it++)
    {
        if (__hip_atomic_load(&sR->msgSeq[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == want)
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            break;
        }
        if (relay || solo || ((it + wave) & 3u) == 0u)            // (solo: the only wavefront of its workgroup that waits)
        {
            ResidentMsg c;
            bool got = false;
            if (!relay)
            {
                got = residentRead(g, want, c);
                if (!got && (it & 127u) == 127u && sysLoad(&s.res->abortDev) != 0u) return false;
            }
            else
            {
                got = residentRead(h, want, c);
                if (got) residentRelay(s, want, c, lane);
                else if (sysLoad(&s.resHost->abort) != 0u) { sysStore(&s.res->abortDev, 1u); return false; }
            }
            if (got)
            {
                // for the workgroup's other wavefronts (and this one: it reads it back below like they do)
                if (lane == 0u)
                {
                    ResMsgR &d = sR->msg[par];
                    d.nValid = c.nValid; d.syms = c.syms; d.nsyms = c.nsyms; d.chan = c.chan; d.sigCh = c.sigCh; d.sigErr = c.sigErr; d.sigPow = c.sigPow; d.sigSnr = c.sigSnr;
                    d.symStride = c.symStride; d.capRows = c.capRows; d.capSig = c.capSig; d.flags = c.flags;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0u) __hip_atomic_store(&sR->msgSeq[par], want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                break;
            }
        }
        if ((it & 63u) == 63u && wall_clock64() - t0 > s.resWatchdog) { sysStore(&s.res->expired, 1u); return false; }
        for (int z = 0; z < s.resSleep; z++) __builtin_amdgcn_s_sleep(8);
    }
    residentMsgFromLds(sR, par, m);
    return (m.flags & 1u) == 0u;
}

/*! A wavefront's own packets and signals of a step into the step's rows, right after the windows of a channel set and BEFORE carryOut
 * (which may overwrite the carry row a packet's first symbols are read from) (the records are its
 * own: written through this compute unit's L1 into L2, waited for here, read back at agent scope; rows written at system scope --
 * through to memory: the consumer is another kernel, a copy engine or the host, and no cache has to be written back for them).
 * ONE device-wide atomic per wavefront and channel set that has anything: packet rows in the low word, signal rows in the high word.
 * Rows are handed out in the order the wavefronts finish: a channel's packets of a step are consecutive and in time order, the channels
 * are not sorted (channel_dev says whose a row is).
 * (Until profiles/r06/s24_*: the wavefront that arrived LAST at the end of a step packed for the whole workgroup, one packet after the
 * other -- 20-30 us behind its own windows, every step, and always the same wavefront, because being last made it later still: the step
 * rate of the whole receiver was that wavefront's cycle, the others waited ~20 us of every 64.) */
template <class C>
__device__ __forceinline__ void residentPackOwn(const StreamArgs &s, const ResLds *sR, const unsigned step, const unsigned chan0, const StreamOut &o, const bool mine,
                                                const int lane, const int openSyms = -1)
{
    constexpr int WPW = C::WPW, T = C::T, LOG2T = C::LOG2T;
    const int wsub = lane >> LOG2T, t = lane & (T - 1);
    const int np = mine ? o.nPkt : 0, ns = (mine && o.sigOut) ? o.nSig : 0;         // (replicated in the channel's T lanes)
    int exP = 0, totP = 0, exS = 0, totS = 0, maxP = 0;
    unsigned hasMask = 0;                                           // bit w: channel w of the wavefront has a packet
#pragma unroll
    for (int w = 0; w < WPW; w++)
    {
        const int a = __shfl(np, w * T), b = __shfl(ns, w * T);
        if (w < wsub) { exP += a; exS += b; }
        totP += a; totS += b;
        maxP = a > maxP ? a : maxP;
        hasMask |= a ? (1u << w) : 0u;
    }
    if (totP == 0 && totS == 0) return;                             // (wave-uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the writer lanes' records are in L2 (they are read back past the L1)
    ResMsgR m;                                                      // where the rows are: from the workgroup's copy of the step's message
    residentMsgFromLds(sR, int(step & 3u), m);
    unsigned long long rs = 0;
    if (lane == 0) rs = atomicAdd(&s.res->rowSig[step & 7u][0], (unsigned long long)unsigned(totP) | ((unsigned long long)unsigned(totS) << 32));
    // (the atomic is in flight while the lengths are read)
    const size_t setOff = size_t(step & 3u) * size_t(s.resRecStride);
    int ln0 = 0;
    if (np > 0) ln0 = agentLoad(&o.pktOut[0].len);
    const unsigned row0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rs), sig0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rs >> 32));
    const unsigned stride = m.symStride;
    // openSyms >= 0 (carryIn without the copy): the channel's symbol row holds only what the step ADDED -- the new part of its first packet
    // (the one it was inside when the step began), the later packets, the symbols of the packet it ends the step inside (openSyms of
    // them, if that one began in this step: np > 0). The first packet's first `carried` symbols never left the carry row.
    if (totP && maxP == 1)
    {
        const int carried = (np > 0 && openSyms >= 0) ? ln0 - (o.nSym - openSyms) : 0;
        // at most one packet per channel (every short step): the wavefront's packets as ONE run of totP x stride elements over the 64 lanes
        const unsigned E = unsigned(totP) * stride;
        for (unsigned base = 0; base < E; base += 128u)
        {
            unsigned short v[2];
            unsigned short *dst[2];
            bool put[2];
#pragma unroll
            for (int u = 0; u < 2; u++)
            {
                const unsigned idx = base + unsigned(u) * 64u + unsigned(lane);
                const bool valid = idx < E;
                const unsigned pq = valid ? idx / stride : 0u, i = idx - pq * stride;
                int wp = 0, cnt = 0;                                // the pq-th channel of the wavefront that has a packet
#pragma unroll
                for (int w = 0; w < WPW; w++) { const int a = int((hasMask >> w) & 1u); if (a && cnt == int(pq)) wp = w; cnt += a; }
                const int lnp = __shfl(ln0, wp * T), cap_ = __shfl(carried, wp * T);
                const unsigned g = chan0 + unsigned(wp), r = row0 + pq;
                const short *sy = reinterpret_cast<const short *>(reinterpret_cast<const char *>(s.symOut + (size_t)g * s.symStride) + setOff);
                const short *cy = sR->carry + (size_t)g * sR->carryCap;
                const int keep = lnp < int(stride) ? lnp : int(stride);
                put[u] = valid && r < m.capRows;
                dst[u] = m.syms + (size_t)r * stride + i;
                v[u] = (put[u] && int(i) < keep) ? (unsigned short)agentLoad(int(i) < cap_ ? cy + i : sy + (int(i) - cap_)) : (unsigned short)0;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) if (put[u]) sysStore(dst[u], v[u]);
        }
        if (np > 0 && t == 0)
        {
            const unsigned r = row0 + unsigned(exP);
            if (r < m.capRows) { sysStore(m.nsyms + r, ln0); if (m.chan) sysStore(m.chan + r, int(chan0 + unsigned(wsub))); }   // (the true length, as lorahip_demod_packets_to_device)
        }
    }
    else if (np > 0)
    {
        // a channel with several packets in the step (long steps): its own T lanes copy them one after the other
        unsigned r = row0 + unsigned(exP);
        int carried = 0;
        if (openSyms >= 0)
        {
            int later = 0;
            for (int j = 1; j < np; j++) later += agentLoad(&o.pktOut[j].len);
            carried = ln0 - (o.nSym - openSyms - later);
        }
        int off = -carried;                                 // (where the first packet would begin in the row if its head were there)
        for (int j = 0; j < np; j++, r++)
        {
            const int ln = j == 0 ? ln0 : agentLoad(&o.pktOut[j].len);
            if (r < m.capRows)
            {
                const int keep = ln < int(stride) ? ln : int(stride);
                unsigned short *dst = m.syms + (size_t)r * stride;
                const short *cy = sR->carry + (size_t)(chan0 + unsigned(wsub)) * sR->carryCap;
                for (int i = t; i < int(stride); i += T)
                    sysStore(dst + i, i < keep ? (unsigned short)agentLoad((j == 0 && i < carried) ? cy + i : o.symOut + off + i) : (unsigned short)0);
                if (t == 0) { sysStore(m.nsyms + r, ln); if (m.chan) sysStore(m.chan + r, int(chan0 + unsigned(wsub))); }
            }
            off += ln;
        }
    }
    if (ns > 0 && t == 0)
    {
        unsigned r = sig0 + unsigned(exS);
        for (int j = 0; j < ns; j++, r++)
            if (r < m.capSig)
            {
                if (m.sigCh) sysStore(m.sigCh + r, int(chan0 + unsigned(wsub)));
                if (m.sigErr) sysStore(m.sigErr + r, agentLoad(&o.sigOut[j].error));
                if (m.sigPow) sysStore(m.sigPow + r, agentLoad(&o.sigOut[j].power));
                if (m.sigSnr) sysStore(m.sigSnr + r, agentLoad(&o.sigOut[j].snr));
            }
    }
}

/*! ONE thread of a workgroup whose wavefronts have all finished the step (their rows are in memory): the workgroup is added to the step's
 * count with one device-wide atomic; the workgroup that completes the count reports the step to the host's pinned memory. */
__device__ __forceinline__ void residentWgDone(const StreamArgs &s, const ResLds *sR, const unsigned step, const unsigned wgCalls, const unsigned wgMore)
{
    const unsigned slot = step & 7u;
    // [63:48] workgroups done, [47:36] of them with a channel that stopped for capacity, [35:0] work() calls
    const unsigned long long add = (1ull << 48) | ((unsigned long long)wgMore << 36) | (unsigned long long)wgCalls;
    const unsigned long long prev = atomicAdd(&s.res->doneCalls[slot][0], add);
    if ((prev >> 48) + 1ull != (unsigned long long)gridDim.x) return;
    const unsigned long long tot = prev + add;
    const unsigned long long rs = agentLoad(&s.res->rowSig[slot][0]);
    const unsigned pkAll = unsigned(rs), sgAll = unsigned(rs >> 32);
    const ResMsgR &m = sR->msg[step & 3u];                          // (one thread: the row capacities straight from the workgroup's copy)
    const unsigned flags = (pkAll > m.capRows ? unsigned(RES_F_PKT_OVERFLOW) : 0u) | ((m.capSig != 0u && sgAll > m.capSig) ? unsigned(RES_F_SIG_OVERFLOW) : 0u) |
                           (((tot >> 36) & 0xfffull) ? unsigned(RES_F_MORE) : 0u);
    // the counters of step + 4: nobody is there yet (the host rings step k + depth + 1, depth <= 3, only after it has seen report k)
    const unsigned nx = (step + 4u) & 7u;
    sysStore(&s.res->doneCalls[nx][0], 0ull); sysStore(&s.res->rowSig[nx][0], 0ull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long *h = s.resHost->sum + 2 * slot;
    const unsigned long long w1 = ((unsigned long long)(step & 0xffu) << 56) | ((unsigned long long)flags << 48) | ((unsigned long long)(sgAll & 0xffffffu) << 24) |
                                  (unsigned long long)(pkAll & 0xffffffu);
    const unsigned long long w0 = ((unsigned long long)step << 32) | (tot & 0xffffffffull);
    sysStore(h + 1, w1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    sysStore(h, w0);
}

/*! The end of a receiver step for one wavefront: it waits for its stores (its rows are in memory before it counts as arrived), adds its
 * work() calls to the workgroup's, and the wavefront that arrives LAST (an LDS counter, no barrier: the others go straight back to
 * polling) adds the workgroup to the step's count with one device-wide atomic. The workgroup that completes the count reports the step
 * to the host's pinned memory. */
template <class C>
__device__ __forceinline__ void residentStepEnd(const StreamArgs &s, const unsigned step, ResLds *sR, int calls, const bool stopped, const int lane)
{
    constexpr int WAVES = 4;
    const int par = int(step & 3u);
    for (int d = 32; d >= 1; d >>= 1) calls += __shfl_xor(calls, d);
    const bool anyStopped = __any(stopped);
    if (lane == 0)
    {
        if (calls) atomicAdd(&sR->calls[par], calls);
        if (anyStopped) sR->more[par] = 1;
    }
    // (spelled out: a workgroup-scope release does not wait for global stores on this target -- the wavefronts of a workgroup share an L1,
    // the compiler emits no s_waitcnt vmcnt for it; found by the randomised soak, profiles/r06/s13_*)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // ... and the counts in LDS
    int arrived = 0;
    if (lane == 0) arrived = __hip_atomic_fetch_add(&sR->arrive[par], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (__builtin_amdgcn_readfirstlane(arrived) != WAVES - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0)
    {
        const unsigned wgCalls = unsigned(sR->calls[par]), wgMore = sR->more[par] ? 1u : 0u;
        sR->calls[par] = 0; sR->more[par] = 0; sR->arrive[par] = 0;                     // for step + 4
        residentWgDone(s, sR, step, wgCalls, wgMore);
    }
}

} // namespace lorahip
